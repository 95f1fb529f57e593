#!/usr/bin/env python3
"""Generate the golden fixtures tests/golden/G*.npz by running the REFERENCE.

Runs ONLY in the build container (needs /root/reference); nothing here is
imported by tests, the product or the bench.  Usage:

    python3 -B tests/golden/make_goldens.py            # writes G1..G8 + KAT

What executes verbatim from the reference:
  efa_xray/assimilation/ensrf.py:33-151        EnSRF.update
  efa_xray/assimilation/assimilation.py:15-49,120-171
  efa_xray/observation/observation.py:17-36,59-87,117-146
  efa_xray/state/ensemble.py:254-267           distance_to_point
The reference's top-level imports of packages absent from this image
(xarray, netCDF4, cPickle, xarray.ufuncs) are satisfied by empty module
objects; none of their functionality is reached on the path above
(`xu.*` is applied to ndarrays only, where xarray.ufuncs delegated to the
NumPy ufunc of the same name).  The state container is duck-typed: a
subclass of the reference's EnsembleState overriding the accessors that
need a real xarray Dataset (shape/nstate/nmems/to_vect/from_vect and
['lat'|'lon']); in G1..G8 Observation.estimate is overridden by a linear
operator (row pick or 4-point weights).

G9..G11 (round 3) pin the forward operator itself: the reference's
  efa_xray/state/ensemble.py:152-168           nearest_points
  efa_xray/state/ensemble.py:170-239           interpolate
  efa_xray/state/ensemble.py:241-252           haversine
  efa_xray/observation/observation.py:40-50    Observation.estimate
run verbatim on `InterpState`, a duck state that additionally offers what those
lines touch: `self['lat'|'lon'|'validtime']` with `.values`, `.shape` and
`__getitem__`, and `self.variables[var].values` of shape (nt, ny, nx, nmem).
The stencil WEIGHTS are read off the reference by interpolating a one-hot
probe state (member m is 1 at flat (t, y, x) index m, 0 elsewhere): its
estimate IS the weight vector.  The 2-D branch's "< 1 km" exact-match case
raises IndexError in the reference (ensemble.py:194-196 writes a 2-D index
into a 1-D array) and is therefore kept out of G9; the 1-D branch's
spaceweights are (1, 4), so the same statement works there and G10 holds one.

A fixture is data only: inputs and the reference's outputs.
"""
import os
import sys
import types
import pickle
from copy import deepcopy

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_placeholders():
    xr = types.ModuleType("xarray")
    xr.DataArray = object
    xr.Dataset = object
    xr.open_dataset = None
    xu = types.ModuleType("xarray.ufuncs")
    for name in ("hypot", "sin", "cos", "radians", "arctan2", "sqrt"):
        setattr(xu, name, getattr(np, name))
    xr.ufuncs = xu
    nc = types.ModuleType("netCDF4")
    nc.Dataset = object
    sys.modules["xarray"] = xr
    sys.modules["xarray.ufuncs"] = xu
    sys.modules["netCDF4"] = nc
    sys.modules["cPickle"] = pickle
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)


_install_placeholders()
from efa_xray.assimilation.ensrf import EnSRF                      # noqa: E402
from efa_xray.observation.observation import (Observation,         # noqa: E402
                                               gaspari_cohn, haversine)
from efa_xray.state.ensemble import EnsembleState                  # noqa: E402


class _Coord:
    def __init__(self, v):
        self.values = np.asarray(v, dtype=np.float64)
        self.shape = self.values.shape


class DuckState(EnsembleState):
    """Holds the state as one (nvar, nt, ny, nx, nmem) array."""

    def __init__(self, arr, lat, lon):
        self.arr = np.array(arr, dtype=np.float64)
        self._lat = _Coord(lat)
        self._lon = _Coord(lon)

    def shape(self):
        return self.arr.shape

    def nmems(self):
        return self.arr.shape[-1]

    def nstate(self):
        return int(np.prod(self.arr.shape[:-1]))

    def to_vect(self):
        return np.reshape(self.arr, (self.nstate(), self.nmems()))

    def from_vect(self, v):
        self.arr = np.reshape(np.array(v), self.arr.shape)

    def __getitem__(self, key):
        return {"lat": self._lat, "lon": self._lon}[key]

    def __deepcopy__(self, memo):
        return DuckState(self.arr.copy(), self._lat.values.copy(), self._lon.values.copy())


class LinOb(Observation):
    """Observation whose forward operator is a fixed linear stencil."""

    def __init__(self, idx, wts, **kw):
        Observation.__init__(self, **kw)
        self.idx = np.asarray(idx, dtype=np.int64)
        self.wts = np.asarray(wts, dtype=np.float64)

    def estimate(self, state):
        rows = state.to_vect()[self.idx]            # (npts, M)
        if len(self.idx) == 1 and self.wts[0] == 1.0:
            return rows[0].copy()
        return (self.wts[:, None] * rows).sum(axis=0)


def run_reference(arr, lat, lon, stencils, values, errors, assim, ob_lat, ob_lon,
                  radii, loc):
    state = DuckState(arr, lat, lon)
    obs = []
    for k in range(len(values)):
        obs.append(LinOb(stencils[k][0], stencils[k][1], value=float(values[k]),
                         error=float(errors[k]), lat=float(ob_lat[k]), lon=float(ob_lon[k]),
                         assimilate_this=bool(assim[k]),
                         localize_radius=(None if radii is None else float(radii[k]))))
    captured = {}
    flt = EnSRF(state, obs, verbose=False, loc=loc)
    orig = flt.format_posterior_state

    def spy(xam, Xap):
        captured["xam"] = np.array(xam)
        captured["Xap"] = np.array(Xap)
        return orig(xam, Xap)

    flt.format_posterior_state = spy
    # HX exactly as the reference's compute_ob_priors sees it
    HX = np.array([ob.estimate(state) for ob in obs])
    xbm, Xbp = EnSRF(deepcopy(state), deepcopy(obs), verbose=False, loc=loc).format_prior_state()
    post_state, obs_out = flt.update()
    nanf = lambda v: np.nan if v is None else float(v)   # noqa: E731
    return dict(
        HX=HX, xbm=xbm, Xbp=Xbp, xam=captured["xam"], Xap=captured["Xap"],
        post=post_state.to_vect(),
        prior_mean=np.array([nanf(o.prior_mean) for o in obs_out]),
        prior_var=np.array([nanf(o.prior_var) for o in obs_out]),
        post_mean=np.array([nanf(o.post_mean) for o in obs_out]),
        post_var=np.array([nanf(o.post_var) for o in obs_out]),
        assimilated=np.array([bool(o.assimilated) for o in obs_out]),
    )


def pack_stencils(stencils):
    npt = max(len(s[0]) for s in stencils)
    idx = np.zeros((len(stencils), npt), dtype=np.int64)
    wts = np.zeros((len(stencils), npt))
    for k, (i, w) in enumerate(stencils):
        idx[k, :len(i)] = i
        wts[k, :len(i)] = w
    return idx, wts


def make_case(name, shape, lat, lon, rng, P, loc, radii=None, errors=1.0, assim=None,
              stencil="pick", pick=None, keep=("xbm", "Xbp", "xam", "Xap"), sigma=3.0):
    nvar, nt, ny, nx, M = shape
    N = nvar * nt * ny * nx
    mu = rng.standard_normal((nvar, nt, ny, nx, 1))
    arr = mu + sigma * rng.standard_normal(shape)
    lat = np.asarray(lat, dtype=np.float64)
    lon = np.asarray(lon, dtype=np.float64)
    if pick is None:
        pick = rng.choice(N, P, replace=False)
    pick = np.asarray(pick)
    stencils = []
    for k in range(P):
        if stencil == "pick":
            stencils.append((np.array([pick[k]]), np.array([1.0])))
        else:
            i = rng.choice(N, 4, replace=False)
            w = rng.random(4) + 0.1
            w = w / w.sum()
            stencils.append((i, w))
    # ob location: the (y, x) of the (first) stencil point
    col = np.array([s[0][0] % (ny * nx) for s in stencils])
    if lat.ndim == 2:
        ob_lat = lat.reshape(-1)[col]
        ob_lon = lon.reshape(-1)[col]
    else:
        ob_lat = lat[col % nx]
        ob_lon = lon[col % nx]
    # jitter so obs are off-grid
    ob_lat = ob_lat + 0.3 * rng.standard_normal(P)
    ob_lon = ob_lon + 0.3 * rng.standard_normal(P)
    errors = np.broadcast_to(np.asarray(errors, dtype=np.float64), (P,)).copy()
    vect = arr.reshape(N, M)
    truth = np.array([(s[1][:, None] * vect[s[0]]).sum(axis=0).mean() for s in stencils])
    values = truth + np.sqrt(errors) * rng.standard_normal(P)
    if assim is None:
        assim = np.ones(P, dtype=bool)
    out = run_reference(arr, lat, lon, stencils, values, errors, assim, ob_lat, ob_lon,
                        radii, loc)
    idx, wts = pack_stencils(stencils)
    fix = dict(
        shape=np.array(shape), X=arr, grid_lat=lat, grid_lon=lon,
        ob_value=values, ob_error=errors, ob_assim=np.asarray(assim, dtype=bool),
        ob_lat=ob_lat, ob_lon=ob_lon,
        ob_radius=(np.full(P, np.nan) if radii is None else np.asarray(radii, dtype=np.float64)),
        loc=np.array("GC" if loc == "GC" else ""),
        sten_idx=idx, sten_wts=wts, HX=out["HX"], post=out["post"],
        prior_mean=out["prior_mean"], prior_var=out["prior_var"],
        post_mean=out["post_mean"], post_var=out["post_var"], assimilated=out["assimilated"],
    )
    for k in keep:
        fix[k] = out[k]
    path = os.path.join(OUT, name + ".npz")
    np.savez(path, **fix)
    print("%-4s N=%d M=%d P=%d loc=%r assimilated=%d -> %s (%.1f KB)" % (
        name, N, M, P, loc, int(out["assimilated"].sum()), os.path.basename(path),
        os.path.getsize(path) / 1024))


def main():
    # ---- known-answer tables for the scalar functions --------------------
    d = np.array([0.0, 400.0, 800.0, 1200.0, 1599.0, 1600.0, 2000.0, 1e-9, 799.999999, 800.000001])
    kat = dict(gc_d=d, gc_c=np.array(800.0), gc_w=gaspari_cohn(d, 800.0),
               gc_w_neg=gaspari_cohn(d, -800.0))
    rng = np.random.default_rng(42)
    pa = np.stack([rng.uniform(-89, 89, 64), rng.uniform(-180, 360, 64)], axis=1)
    pb = np.stack([rng.uniform(-89, 89, 64), rng.uniform(-180, 360, 64)], axis=1)
    pa[0] = (47.4489, -122.3094)
    pb[0] = (45.0, -120.0)
    pb[1] = pa[1]                                   # zero distance
    kat["hv_a"] = pa
    kat["hv_b"] = pb
    kat["hv_km"] = np.array([haversine(tuple(a), tuple(b)) for a, b in zip(pa, pb)])
    glat, glon = np.meshgrid(np.linspace(-80, 80, 9), np.linspace(0, 350, 12), indexing="ij")
    st = DuckState(np.zeros((1, 1, 9, 12, 2)), glat, glon)
    kat["dp_lat"] = glat
    kat["dp_lon"] = glon
    kat["dp_pt"] = np.array([33.3, 200.2])
    kat["dp_km"] = st.distance_to_point(33.3, 200.2)
    np.savez(os.path.join(OUT, "KAT.npz"), **kat)
    print("KAT  gaspari_cohn/haversine/distance_to_point known answers")

    # ---- G1: cfg-1, 1-D Lorenz-96 size, no localisation ------------------
    rng = np.random.default_rng(0)
    assim = np.ones(10, dtype=bool)
    assim[3] = False
    make_case("G1", (1, 1, 1, 40, 20), np.zeros((1, 40)), np.linspace(0, 351, 40)[None, :],
              rng, 10, False, assim=assim, pick=np.arange(0, 40, 4))
    # ---- G2: small 2-D grid with GC ---------------------------------------
    rng = np.random.default_rng(1)
    lat, lon = np.meshgrid(np.linspace(30, 50, 6), np.linspace(230, 260, 8), indexing="ij")
    make_case("G2", (2, 3, 6, 8, 10), lat, lon, rng, 5, "GC", radii=np.full(5, 800.0), errors=0.5)
    # ---- G3: 1-D lat/lon branch (ensrf.py:110-111) ------------------------
    rng = np.random.default_rng(2)
    make_case("G3", (1, 1, 1, 40, 20), np.linspace(-40, 40, 40), np.linspace(100, 178, 40),
              rng, 8, "GC", radii=np.full(8, 1500.0))
    # ---- G4: 4-point-weight linear H, no loc ------------------------------
    rng = np.random.default_rng(7)
    make_case("G4", (1, 1, 40, 50, 24), *np.meshgrid(np.linspace(20, 60, 40), np.linspace(200, 280, 50), indexing="ij"),
              rng, 60, False, errors=np.random.default_rng(3).uniform(0.3, 1.3, 60), stencil="w4")
    # ---- G5: mixed per-ob radii incl. one with all-zero taper -------------
    rng = np.random.default_rng(5)
    lat, lon = np.meshgrid(np.linspace(-30, 30, 12), np.linspace(0, 90, 16), indexing="ij")
    radii = np.array([300.0, 2500.0, 1e-3, 900.0, 5000.0, 150.0, 1200.0, 40000.0, 700.0, 60.0, 2000.0, 450.0])
    assim = np.ones(12, dtype=bool)
    assim[[4, 9]] = False
    make_case("G5", (2, 2, 12, 16, 16), lat, lon, rng, 12, "GC", radii=radii, assim=assim,
              errors=np.random.default_rng(6).uniform(0.2, 2.0, 12))
    # ---- G6: mid-size GC, realistic tile counts ---------------------------
    rng = np.random.default_rng(11)
    lat, lon = np.meshgrid(np.linspace(-60, 60, 64), np.linspace(0, 354, 64), indexing="ij")
    make_case("G6", (2, 1, 64, 64, 50), lat, lon, rng, 100, "GC", radii=np.full(100, 1000.0),
              keep=("xam",))
    # ---- G7: many obs (several device batches), no loc --------------------
    rng = np.random.default_rng(12)
    assim = np.random.default_rng(13).random(300) > 0.1
    make_case("G7", (1, 2, 30, 50, 50), *np.meshgrid(np.linspace(10, 70, 30), np.linspace(0, 120, 50), indexing="ij"),
              rng, 300, False, assim=assim, errors=np.random.default_rng(14).uniform(0.5, 2.0, 300),
              keep=("xam",))
    # ---- G8: many obs with GC and the headline member count ---------------
    rng = np.random.default_rng(15)
    lat, lon = np.meshgrid(np.linspace(-85, 85, 24), np.linspace(0, 345, 24), indexing="ij")
    assim = np.random.default_rng(16).random(200) > 0.05
    make_case("G8", (3, 1, 24, 24, 100), lat, lon, rng, 200, "GC",
              radii=np.random.default_rng(17).uniform(500, 3000, 200), assim=assim, keep=("xam",))


# ---------------------------------------------------------------------------
# f1: the reference's own forward operator (nearest_points / interpolate / estimate)
# ---------------------------------------------------------------------------
class _Field:
    """What `self[...]` / `self.variables[...]` must offer on the lines cited above."""

    def __init__(self, v):
        self.values = np.asarray(v)
        self.shape = self.values.shape

    def __getitem__(self, key):
        return _Field(self.values[key])


class InterpState(DuckState):
    """DuckState + validtime + named variables, for ensemble.py:152-239."""

    def __init__(self, arr, lat, lon, valids, names):
        DuckState.__init__(self, arr, lat, lon)
        self._lat = _Field(np.asarray(lat, dtype=np.float64))
        self._lon = _Field(np.asarray(lon, dtype=np.float64))
        self._valid = _Field(np.asarray(valids))
        self.names = list(names)

    @property
    def variables(self):
        return dict((n, _Field(self.arr[i])) for i, n in enumerate(self.names))

    def __getitem__(self, key):
        return {"lat": self._lat, "lon": self._lon, "validtime": self._valid}[key]

    def __deepcopy__(self, memo):
        return InterpState(self.arr.copy(), self._lat.values.copy(), self._lon.values.copy(),
                           self._valid.values.copy(), self.names)


def _interp_case(name, shape, lat, lon, valids, rng, P, lat_rng, lon_rng, exact=None):
    nvar, nt, ny, nx, M = shape
    names = ["var%d" % i for i in range(nvar)]
    arr = rng.standard_normal((nvar, nt, ny, nx, 1)) + 2.0 * rng.standard_normal(shape)
    state = InterpState(arr, lat, lon, valids, names)
    # one-hot probe: member m <-> flat (t, y, x) index m of ONE variable
    eye = np.eye(nt * ny * nx).reshape(1, nt, ny, nx, nt * ny * nx)
    probe = InterpState(eye, lat, lon, valids, ["var0"])
    span = (valids[-1] - valids[0]) / np.timedelta64(1, "s")
    ob_var = rng.integers(0, nvar, P)
    ob_lat = rng.uniform(lat_rng[0], lat_rng[1], P)
    ob_lon = rng.uniform(lon_rng[0], lon_rng[1], P)
    # times: every third ob exactly ON a valid time, the rest strictly between
    ob_time = []
    for k in range(P):
        if k % 3 == 0:
            ob_time.append(valids[k % nt])
        else:
            ob_time.append(valids[0] + np.timedelta64(int(rng.uniform(1, span - 1)), "s"))
    ob_time = np.array(ob_time, dtype="datetime64[s]")
    if exact is not None:                            # 1-D branch only (see the module docstring)
        k, n = exact
        ob_lat[k] = state["lat"].values[n] + 1e-4
        ob_lon[k] = state["lon"].values[n] - 1e-4
    near = np.zeros((P, 4, len(np.shape(state["lat"].values))), dtype=np.int64)
    HX = np.zeros((P, M))
    W = np.zeros((P, nt * ny * nx))
    for k in range(P):
        ob = Observation(value=0.0, obtype=names[ob_var[k]], time=ob_time[k], lat=float(ob_lat[k]),
                         lon=float(ob_lon[k]), error=1.0)
        near[k] = np.stack(state.nearest_points(ob.lat, ob.lon, npt=4), axis=-1)
        HX[k] = np.reshape(ob.estimate(state), (M,))            # the 1-D branch returns (1, M)
        ob.obtype = "var0"
        W[k] = np.reshape(ob.estimate(probe), (-1,))
    # an ob outside the valid times: the reference prints and returns None (ensemble.py:207-209)
    late = Observation(value=0.0, obtype="var0", time=valids[-1] + np.timedelta64(1, "s"), lat=float(ob_lat[0]),
                       lon=float(ob_lon[0]))
    assert late.estimate(state) is None
    fix = dict(shape=np.array(shape), X=arr, grid_lat=np.asarray(lat, dtype=np.float64),
               grid_lon=np.asarray(lon, dtype=np.float64), validtime=np.asarray(valids, dtype="datetime64[s]"),
               var_names=np.array(names), ob_var=ob_var.astype(np.int64), ob_time=ob_time, ob_lat=ob_lat, ob_lon=ob_lon,
               nearest=near, HX=HX, weights=W)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **fix)
    print("%-4s interpolate: shape=%s P=%d nonzero weights/ob=%s -> %s (%.1f KB)" % (
        name, shape, P, sorted(set((W != 0).sum(axis=1).tolist())), os.path.basename(path), os.path.getsize(path) / 1024))
    return state, names


def main_f1():
    t0 = np.datetime64("2020-03-01T00:00:00", "s")
    # ---- G9: 2-D lat/lon, irregular datetime64 valid times, three variables ----
    rng = np.random.default_rng(21)
    valids = t0 + np.array([0, 6 * 3600, 9 * 3600, 18 * 3600]).astype("timedelta64[s]")
    lat, lon = np.meshgrid(np.linspace(30, 50, 14), np.linspace(230, 262, 18), indexing="ij")
    _interp_case("G9", (3, 4, 14, 18, 12), lat, lon, valids, rng, 48, (31, 49), (231, 261))
    # ---- G10: 1-D lat/lon branch (ensemble.py:186-190: y index = x index = n) ----
    rng = np.random.default_rng(22)
    valids = t0 + np.array([0, 3600, 7200]).astype("timedelta64[s]")
    _interp_case("G10", (2, 3, 16, 16, 10), np.linspace(30, 50, 16), np.linspace(230, 262, 16), valids, rng, 36,
                 (31, 49), (231, 261), exact=(7, 5))
    # ---- G11: EnSRF.update() end to end with the reference's own estimate (no LinOb) ----
    rng = np.random.default_rng(23)
    shape = (2, 3, 10, 12, 16)
    nvar, nt, ny, nx, M = shape
    names = ["var%d" % i for i in range(nvar)]
    valids = t0 + np.array([0, 3600, 10800]).astype("timedelta64[s]")
    lat, lon = np.meshgrid(np.linspace(25, 55, ny), np.linspace(220, 270, nx), indexing="ij")
    arr = rng.standard_normal((nvar, nt, ny, nx, 1)) + 3.0 * rng.standard_normal(shape)
    state = InterpState(arr, lat, lon, valids, names)
    P = 30
    ob_var = rng.integers(0, nvar, P)
    ob_lat = rng.uniform(26, 54, P)
    ob_lon = rng.uniform(221, 269, P)
    ob_time = np.array([valids[k % nt] if k % 4 == 0 else valids[0] + np.timedelta64(int(rng.uniform(1, 10799)), "s")
                        for k in range(P)], dtype="datetime64[s]")
    errors = rng.uniform(0.4, 1.6, P)
    radii = rng.uniform(600, 2500, P)
    assim = rng.random(P) > 0.15
    values = rng.standard_normal(P) * 2.0
    obs = [Observation(value=float(values[k]), obtype=names[ob_var[k]], time=ob_time[k], error=float(errors[k]),
                       lat=float(ob_lat[k]), lon=float(ob_lon[k]), assimilate_this=bool(assim[k]),
                       localize_radius=float(radii[k])) for k in range(P)]
    HX = np.array([ob.estimate(state) for ob in obs])
    flt = EnSRF(state, obs, verbose=False, loc="GC")
    captured = {}
    orig = flt.format_posterior_state

    def spy(xam, Xap):
        captured["xam"] = np.array(xam)
        return orig(xam, Xap)

    flt.format_posterior_state = spy
    post_state, obs_out = flt.update()
    nanf = lambda v: np.nan if v is None else float(v)   # noqa: E731
    fix = dict(shape=np.array(shape), X=arr, grid_lat=lat, grid_lon=lon, validtime=valids.astype("datetime64[s]"),
               var_names=np.array(names), ob_var=ob_var.astype(np.int64), ob_time=ob_time, ob_lat=ob_lat, ob_lon=ob_lon,
               ob_value=values, ob_error=errors, ob_assim=assim, ob_radius=radii, loc=np.array("GC"), HX=HX,
               xam=captured["xam"], post=post_state.to_vect(),
               prior_mean=np.array([nanf(o.prior_mean) for o in obs_out]),
               prior_var=np.array([nanf(o.prior_var) for o in obs_out]),
               post_mean=np.array([nanf(o.post_mean) for o in obs_out]),
               post_var=np.array([nanf(o.post_var) for o in obs_out]),
               assimilated=np.array([bool(o.assimilated) for o in obs_out]))
    path = os.path.join(OUT, "G11.npz")
    np.savez_compressed(path, **fix)
    print("G11  EnSRF.update() with Observation.estimate as shipped: shape=%s P=%d assimilated=%d -> %s (%.1f KB)" % (
        shape, P, int(fix["assimilated"].sum()), os.path.basename(path), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    if "--f1-only" not in sys.argv:
        main()
    main_f1()
