"""GPU tests of the Python API either side of the loop: the inflation hook of
`format_prior_state` (assimilation.py:131-134), the per-ob statistics table fed from the kernel's
diagnostics (postprocess.py:8-39), radius validation limited to assimilated obs (ensrf.py:74-76
precedes :101) and the default forward operator built on the device (ensemble.py:152-239)."""
from copy import deepcopy

import numpy as np
import pytest

from oracle import ensrf_oracle as orc
from test_gpu_parity import assert_parity, RTOL

pytestmark = pytest.mark.gpu


def _state_and_obs(seed, loc, P=40, M=20, nvar=2, nt=1, ny=12, nx=16):
    from efa_xray_amd import EnsembleState, Observation
    rng = np.random.default_rng(seed)
    lat, lon = np.meshgrid(np.linspace(20, 60, ny), np.linspace(200, 280, nx), indexing="ij")
    arr = rng.standard_normal((nvar, nt, ny, nx, 1)) + 3.0 * rng.standard_normal((nvar, nt, ny, nx, M))
    state = EnsembleState.from_array(arr, lat, lon)
    N = state.nstate()
    rows = rng.choice(N, P, replace=False)

    class RowOb(Observation):
        def estimate(self, st):
            return st.to_vect()[self.row].copy()

    X = state.to_vect().copy()
    obs = []
    for k in range(P):
        col = rows[k] % (ny * nx)
        ob = RowOb(value=float(X[rows[k]].mean() + rng.standard_normal()), error=float(rng.uniform(0.5, 1.5)),
                   lat=float(lat.reshape(-1)[col]), lon=float(lon.reshape(-1)[col]), obtype="var0", time=0,
                   assimilate_this=(k % 7 != 3), localize_radius=1500.0, description="ob%d" % k)
        ob.row = int(rows[k])
        obs.append(ob)
    kw = {}
    if loc == "GC":
        kw = dict(loc="GC", ob_lat=[o.lat for o in obs], ob_lon=[o.lon for o in obs],
                  ob_halfwidth=[o.localize_radius for o in obs], grid_lat=lat, grid_lon=lon,
                  state_shape=(nvar, nt, ny, nx))
    return state, obs, X, rows, kw


def _oracle_cycle(X, rows, obs, kw):
    return orc.ensrf_cycle(X, X[rows], np.array([o.value for o in obs]), np.array([o.error for o in obs]),
                           np.array([o.assimilate_this for o in obs]), **kw)


@pytest.mark.parametrize("loc", [False, "GC"])
def test_update_applies_float_inflation_like_format_prior_state(loc):
    """`EnSRF(state, obs, inflation=1.3).update()` == the oracle cycle on `inflate_constant(X, 1.3)`
    (assimilation.py:131-134 inside format_prior_state, reached from ensrf.py:44; float branch :62-69,
    in place on the caller's state).  PARITY UNPINNED: the reference's inflation branch needs a real
    xarray Dataset (absent from the image), so no golden covers it; this follows the source text."""
    from efa_xray_amd import EnSRF
    state, obs, X, rows, kw = _state_and_obs(3, loc)
    flt = EnSRF(state, obs, inflation=1.3, verbose=False, loc=loc)
    post_state, obs_out = flt.update()
    Xi = orc.inflate_constant(X, 1.3)
    assert flt.is_inflated
    assert_parity(state.to_vect(), Xi, "the caller's state is inflated in place (assimilation.py:67)")
    ref_post, _, _, diag = _oracle_cycle(Xi, rows, obs, kw)
    assert_parity(post_state.to_vect(), ref_post, "post (inflation=1.3)")
    assert_parity([o.prior_var for o in obs_out], diag["prior_var"], "prior_var")
    got = np.array([o.post_mean for o in obs_out if o.assimilated], dtype=float)
    assert_parity(got, diag["post_mean"][diag["assimilated"]], "post_mean")
    # a second update() on the same filter object does not inflate again (assimilation.py:57-59)
    before = state.to_vect().copy()
    flt.update()
    assert np.array_equal(state.to_vect(), before)
    # and the (xbm, Xbp) helper pair goes through the same hook
    state2, obs2, X2, rows2, kw2 = _state_and_obs(3, loc)
    f2 = EnSRF(state2, obs2, inflation=1.3, verbose=False, loc=loc)
    xbm, Xbp = f2.format_prior_state()
    r_xbm, r_Xbp = orc.format_prior_state(Xi, Xi[rows])
    assert_parity(xbm, r_xbm, "xbm of the inflated prior")
    assert_parity(Xbp, r_Xbp, "Xbp of the inflated prior")


def test_update_with_per_variable_and_per_dimension_inflation():
    from efa_xray_amd import EnSRF
    state, obs, X, rows, kw = _state_and_obs(4, False, nt=3)
    nvar, nt, ny, nx, M = state.shape()
    f = np.linspace(1.0, 1.4, nt)
    mine = deepcopy(state)
    post_state, _ = EnSRF(mine, obs, inflation={"validtime": f}, verbose=False).update()
    assert np.array_equal(mine.to_vect(), X), "per-dimension factors leave the caller's state alone (assimilation.py:96)"
    arr = X.reshape(nvar, nt, ny, nx, M)
    m = arr.mean(axis=-1, keepdims=True)
    Xi = ((arr - m) * f[None, :, None, None, None] + m).reshape(-1, M)
    ref_post, _, _, _ = _oracle_cycle(Xi, rows, obs, kw)
    assert_parity(post_state.to_vect(), ref_post, "post (per-time inflation)")


def test_update_with_inflation_factors_from_a_netcdf_file_and_state_round_trip(tmp_path):
    """f4: `EnSRF(state, obs, inflation='factors.nc').update()` (assimilation.py:71-79: per-variable factors on any
    subset of the state's dimensions, broadcast by dimension name) == the oracle cycle on the prior inflated in
    closed form; and the posterior survives `save_to_disk` / `from_netcdf` bit for bit (ensemble.py:269-273).
    PARITY UNPINNED: the reference's reader and writer need xarray; this follows the source text."""
    from scipy.io import netcdf_file
    from efa_xray_amd import EnSRF, EnsembleState
    state, obs, X, rows, kw = _state_and_obs(6, False, nt=2)
    nvar, nt, ny, nx, M = state.shape()
    rng = np.random.default_rng(11)
    fac = {name: 1.0 + rng.random((ny, nx)) for name in state.vars()}
    fn = str(tmp_path / "factors.nc")
    with netcdf_file(fn, "w", version=2) as f:
        f.createDimension("y", ny)
        f.createDimension("x", nx)
        for name in state.vars():
            f.createVariable(name, "d", ("y", "x"))[:] = fac[name]
    mine = deepcopy(state)
    post_state, _ = EnSRF(mine, obs, inflation=fn, verbose=False).update()
    assert np.array_equal(mine.to_vect(), X), "the file form rebinds the prior and leaves the caller's state alone"
    arr = X.reshape(nvar, nt, ny, nx, M)
    m = arr.mean(axis=-1, keepdims=True)
    F = np.stack([fac[name] for name in state.vars()])[:, None, :, :, None]
    Xi = ((arr - m) * F + m).reshape(-1, M)
    ref_post, _, _, _ = _oracle_cycle(Xi, rows, obs, kw)
    assert_parity(post_state.to_vect(), ref_post, "post (inflation factors from a file)")
    out = str(tmp_path / "post.nc")
    post_state.save_to_disk(out)
    assert np.array_equal(EnsembleState.from_netcdf(out).to_vect(), post_state.to_vect())


@pytest.mark.parametrize("loc", [False, "GC"])
def test_statistics_table_from_kernel_diagnostics(loc):
    """f3: `obs_assimilation_statistics(prior, post, obs, from_diagnostics=True)` after a real `update()` equals
    the table built from the oracle's diagnostics; the reference's re-interpolating form (postprocess.py:24-31)
    is evaluated too and agrees on the prior columns."""
    from efa_xray_amd import EnSRF
    from efa_xray_amd.postprocess.postprocess import obs_assimilation_statistics
    state, obs, X, rows, kw = _state_and_obs(8, loc)
    post_state, obs_out = EnSRF(state, obs, verbose=False, loc=loc).update()
    _, _, _, diag = _oracle_cycle(X, rows, obs, kw)
    df = obs_assimilation_statistics(state, post_state, obs_out, from_diagnostics=True)
    assert len(df) == len(obs)
    assert_parity(np.asarray(df["prior mean"], dtype=float), diag["prior_mean"], "prior mean column")
    assert_parity(np.asarray(df["prior variance"], dtype=float), diag["prior_var"], "prior variance column")
    a = diag["assimilated"]
    assert np.array_equal(np.asarray(df["assimilated"], dtype=bool), a)
    assert_parity(np.asarray(df["post mean"], dtype=float)[a], diag["post_mean"][a], "post mean column")
    assert_parity(np.asarray(df["post variance"], dtype=float)[a], diag["post_var"][a], "post variance column")
    # unassimilated obs report their prior as their posterior
    assert_parity(np.asarray(df["post mean"], dtype=float)[~a], diag["prior_mean"][~a], "post mean of unassimilated obs")
    assert list(df["lat"]) == [o.lat for o in obs] and list(df["description"]) == [o.description for o in obs]
    # the reference's form: re-estimate on prior and posterior states
    df2 = obs_assimilation_statistics(state, post_state, obs_out)
    # prior mean of ob k in the table is the mean of the UNTOUCHED prior (not the running one)
    assert_parity(np.asarray(df2["prior mean"], dtype=float), X[rows].mean(axis=1), "re-interpolated prior mean")
    if not loc:
        # without localisation the last assimilated ob's row is final when it is assimilated
        k = int(np.nonzero(a)[0][-1])
        assert abs(df2["post mean"][k] - diag["post_mean"][k]) <= RTOL * max(1.0, abs(diag["post_mean"][k]))


def test_unassimilated_obs_may_lack_a_localize_radius():
    """ensrf.py:74-76 skips an unassimilated ob before its radius is ever read (:101)."""
    from efa_xray_amd import EnSRF
    state, obs, X, rows, kw = _state_and_obs(12, "GC")
    for k, o in enumerate(obs):
        if not o.assimilate_this:
            o.localize_radius = None if k % 2 else 0.0
    post_state, obs_out = EnSRF(state, obs, verbose=False, loc="GC").update()
    kw["ob_halfwidth"] = [o.localize_radius if o.assimilate_this else 1.0 for o in obs]
    ref_post, _, _, diag = _oracle_cycle(X, rows, obs, kw)
    assert_parity(post_state.to_vect(), ref_post, "post")
    assert_parity([o.prior_var for o in obs_out], diag["prior_var"], "prior_var")
    obs[0].assimilate_this, obs[0].localize_radius = True, None
    with pytest.raises(ValueError):
        EnSRF(state, obs, verbose=False, loc="GC").update()


# ---------------------------------------------------------------------------
# f1: the reference's point-interpolation forward operator on the device
# ---------------------------------------------------------------------------
def _interp_state(seed, ny=14, nx=18, nt=3, nvar=2, M=12, one_d=False):
    from efa_xray_amd import EnsembleState
    rng = np.random.default_rng(seed)
    if one_d:
        ny = nx
        lat, lon = np.linspace(30, 50, nx), np.linspace(230, 262, nx)
    else:
        lat, lon = np.meshgrid(np.linspace(30, 50, ny), np.linspace(230, 262, nx), indexing="ij")
    arr = rng.standard_normal((nvar, nt, ny, nx, 1)) + 2.0 * rng.standard_normal((nvar, nt, ny, nx, M))
    return EnsembleState.from_array(arr, lat, lon, validtime=np.array([0.0, 3600.0, 7200.0])[:nt]), lat, lon


def _point_obs(state, rng, P, lat_rng, lon_rng):
    from efa_xray_amd import Observation
    names = state.vars()
    obs = []
    for k in range(P):
        obs.append(Observation(value=float(rng.standard_normal()), obtype=names[k % len(names)],
                               time=float([0.0, 1800.0, 3600.0, 5000.0, 7200.0][k % 5]), error=1.0,
                               lat=float(rng.uniform(*lat_rng)), lon=float(rng.uniform(*lon_rng)),
                               assimilate_this=(k % 6 != 1), localize_radius=1200.0))
    return obs


@pytest.mark.parametrize("one_d", [False, True])
def test_device_interpolation_stencils_vs_host_and_oracle(one_d):
    """`efa_interp_stencils` against the host restatement `EnsembleState.interp_stencil` (ensemble.py:152-239) and, for
    the space weights, against the oracle's `interp_space_weights` (both pinned to the reference by fixtures G9/G10:
    `test_host_interpolation_stencil_matches_the_reference`, `test_oracle_forward_operator_matches_reference`)."""
    from efa_xray_amd import _lib
    state, lat, lon = _interp_state(31, one_d=one_d)
    rng = np.random.default_rng(32)
    obs = _point_obs(state, rng, 60, (31, 49), (231, 261))
    # an exact match (within 1 km of a grid point): weight 1 there (the reference raises IndexError, ensemble.py:194-196)
    if not one_d:
        obs[7].lat, obs[7].lon = float(lat[5, 6]), float(lon[5, 6])
    ctx = _lib.get_context(0)
    nvar, nt, ny, nx, M = state.shape()
    names = state.vars()
    idx, wts, st = ctx.interp_stencils(nvar, nt, ny, nx, state.coords["lat"], state.coords["lon"], state.ensemble_times(),
                                       [names.index(o.obtype) for o in obs], [o.time for o in obs],
                                       [o.lat for o in obs], [o.lon for o in obs])
    assert not st.any()
    X = state.to_vect()
    for k, o in enumerate(obs):
        rows, w = o.stencil(state)
        got = dict((int(r), float(v)) for r, v in zip(idx[k], wts[k]) if r >= 0)
        assert sorted(got) == sorted(int(r) for r in rows), "ob %d: stencil rows" % k
        assert_parity(np.array([got[int(r)] for r in rows]), w, "ob %d: stencil weights" % k)
        if not one_d:
            iy, ix, sw = orc.interp_space_weights(lat, lon, o.lat, o.lon)
            tot = {}
            for r, v in got.items():
                tot[r % (ny * nx)] = tot.get(r % (ny * nx), 0.0) + v     # summed over the (<= 2) time slots
            assert_parity(np.array([tot[int(y) * nx + int(x)] for y, x in zip(iy, ix)]), sw, "ob %d: space weights" % k)
    if not one_d:
        k7 = dict((int(r), float(v)) for r, v in zip(idx[7], wts[7]) if v != 0.0)
        assert len(k7) <= 2 and abs(sum(k7.values()) - 1.0) < 1e-15
    # the estimates: device gather == host gather
    Xd = ctx.to_device(X)
    HX = ctx.empty((len(obs), M))
    ctx.forward_interp(ny * nx, 0, ny * nx, nvar * nt, M, Xd, HX)
    ref = np.array([o.estimate(state) for o in obs])
    assert_parity(HX.download(), ref, "HX")
    # sharded by columns: partial sums add up (the all-reduce payload)
    acc = np.zeros_like(ref)
    for lo, hi in ((0, 100), (100, ny * nx)):
        rows = (np.arange(nvar * nt)[:, None] * (ny * nx) + np.arange(lo, hi)[None, :]).reshape(-1)
        Xs = ctx.to_device(np.ascontiguousarray(X[rows]))
        ctx.forward_interp(ny * nx, lo, hi, nvar * nt, M, Xs, HX)
        acc += HX.download()
    assert_parity(acc, ref, "sharded HX sum")


@pytest.mark.parametrize("name", ["G9", "G10"])
def test_device_forward_operator_matches_the_reference(name):
    """`efa_interp_stencils` + `efa_forward_interp_dev` against what the reference's own nearest_points / interpolate /
    Observation.estimate (ensemble.py:152-239, observation.py:40-50) returned on the same state: fixtures G9 (2-D lat/lon,
    irregular datetime64 valid times, obs on and between them, three variables) and G10 (the 1-D lat/lon branch, one ob
    within 1 km of a grid point).  Tolerance 1e-10 as everywhere."""
    from conftest import load_golden
    from efa_xray_amd import _lib
    g = load_golden(name)
    nvar, nt, ny, nx, M = [int(v) for v in g["shape"]]
    P = len(g["ob_lat"])
    t0 = g["validtime"][0]
    vt = (g["validtime"] - t0) / np.timedelta64(1, "s")
    ot = (g["ob_time"] - t0) / np.timedelta64(1, "s")
    ctx = _lib.get_context(0)
    idx, wts, st = ctx.interp_stencils(nvar, nt, ny, nx, g["grid_lat"], g["grid_lon"], vt, g["ob_var"], ot,
                                       g["ob_lat"], g["ob_lon"])
    assert not st.any()
    per = nt * ny * nx
    for k in range(P):
        dense = np.zeros(nvar * per)
        keep = idx[k] >= 0
        np.add.at(dense, idx[k][keep], wts[k][keep])
        dense = dense.reshape(nvar, per)
        iv = int(g["ob_var"][k])
        assert not dense[np.arange(nvar) != iv].any(), "ob %d: weight on another variable" % k
        assert_parity(dense[iv], g["weights"][k], "ob %d: stencil weights" % k)
        # the four points are the reference's four (as a set: the order among them carries no meaning downstream)
        cols = sorted(set(int(r) % (ny * nx) for r, w in zip(idx[k], wts[k]) if r >= 0))
        if g["nearest"].shape[-1] == 2:
            ref_cols = sorted(int(y) * nx + int(x) for y, x in g["nearest"][k])
        else:
            ref_cols = sorted(int(n) * nx + int(n) for (n,) in g["nearest"][k])
        assert set(cols) <= set(ref_cols), "ob %d: stencil columns" % k
    Xd = ctx.to_device(np.ascontiguousarray(g["X"].reshape(-1, M)))
    HX = ctx.empty((P, M))
    ctx.forward_interp(ny * nx, 0, ny * nx, nvar * nt, M, Xd, HX)
    assert_parity(HX.download(), g["HX"], "HX")


def test_update_with_plain_observations_matches_the_reference_end_to_end():
    """G11: the reference's `EnSRF(state, obs, loc='GC').update()` with `Observation.estimate` as shipped -- forward
    operator, perturbation formation, localised loop, posterior rebuild -- against this package's `EnSRF.update()`
    on the same state and obs (the stencils are built and applied on the device)."""
    from conftest import load_golden
    from efa_xray_amd import EnsembleState, Observation, EnSRF
    g = load_golden("G11")
    names = [str(n) for n in g["var_names"]]
    state = EnsembleState.from_array(g["X"], g["grid_lat"], g["grid_lon"], varnames=names, validtime=g["validtime"])
    obs = [Observation(value=float(g["ob_value"][k]), obtype=names[g["ob_var"][k]], time=g["ob_time"][k],
                       error=float(g["ob_error"][k]), lat=float(g["ob_lat"][k]), lon=float(g["ob_lon"][k]),
                       assimilate_this=bool(g["ob_assim"][k]), localize_radius=float(g["ob_radius"][k]))
           for k in range(len(g["ob_value"]))]
    flt = EnSRF(state, obs, verbose=False, loc="GC")
    assert flt._default_forward_operator()
    post_state, obs_out = flt.update()
    assert obs_out is obs
    assert_parity(post_state.to_vect(), g["post"], "post")
    for key in ("prior_mean", "prior_var"):
        assert_parity([getattr(o, key) for o in obs_out], g[key], key)
    done = g["assimilated"]
    assert [bool(o.assimilated) for o in obs_out] == done.tolist()
    for key in ("post_mean", "post_var"):
        assert_parity([getattr(o, key) for o, a in zip(obs_out, done) if a], g[key][done], key)
        assert all(getattr(o, key) is None for o, a in zip(obs_out, done) if not a)
    m, p = EnSRF(state, obs, verbose=False).compute_ob_priors()
    rm, rp = orc.compute_ob_priors(g["HX"])
    assert_parity(m, rm, "ob prior means")
    assert_parity(p, rp, "ob prior perturbations")


def test_update_with_the_default_forward_operator_runs_on_device():
    """`EnSRF.update()` with plain `Observation` objects: no per-ob Python loop -- the stencils are built and applied
    on the device -- and the result equals the oracle cycle on the host-interpolated estimates."""
    from efa_xray_amd import EnSRF
    state, lat, lon = _interp_state(41)
    rng = np.random.default_rng(42)
    obs = _point_obs(state, rng, 50, (31, 49), (231, 261))
    calls = []
    orig = type(state).interpolate
    try:
        type(state).interpolate = lambda self, *a: calls.append(1) or orig(self, *a)
        flt = EnSRF(state, obs, verbose=False, loc="GC")
        assert flt._default_forward_operator()
        post_state, obs_out = flt.update()
        assert calls == [], "update() went through the per-ob host interpolation"
    finally:
        type(state).interpolate = orig
    X = state.to_vect()
    HX = np.array([o.estimate(state) for o in obs])
    nvar, nt, ny, nx, M = state.shape()
    ref_post, _, _, diag = orc.ensrf_cycle(X, HX, np.array([o.value for o in obs]), np.array([o.error for o in obs]),
                                           np.array([o.assimilate_this for o in obs]), loc="GC",
                                           ob_lat=[o.lat for o in obs], ob_lon=[o.lon for o in obs],
                                           ob_halfwidth=[o.localize_radius for o in obs], grid_lat=lat, grid_lon=lon,
                                           state_shape=(nvar, nt, ny, nx))
    assert_parity(post_state.to_vect(), ref_post, "post")
    assert_parity([o.prior_mean for o in obs_out], diag["prior_mean"], "prior_mean")
    # compute_ob_priors (assimilation.py:36-49) takes the same route
    m, p = EnSRF(state, obs, verbose=False).compute_ob_priors()
    rm, rp = orc.compute_ob_priors(HX)
    assert_parity(m, rm, "ob prior means")
    assert_parity(p, rp, "ob prior perturbations")
    # an ob outside the state's time range: the reference prints a message and then fails on None.mean()
    obs[3].time = 1e9
    with pytest.raises(ValueError):
        EnSRF(state, obs, verbose=False).update()


def test_nearest_four_on_a_global_grid_with_mirror_ties():
    """On a global regular grid the reference's pseudo-distance hypot(sin(lat_g)-sin(lat), cos(lon_g)-cos(lon))
    (ensemble.py:160-163) cannot tell lon from 360-lon, so its "nearest four" contain mirror points and exact ties;
    the device kernel must return four points none of which is farther (in that pseudo-distance) than the fourth
    smallest value, in ascending order, ties towards the lower index."""
    from efa_xray_amd import _lib
    ctx = _lib.get_context(0)
    ny, nx = 91, 180
    lat2, lon2 = np.meshgrid(np.linspace(-90, 90, ny), np.linspace(0, 358, nx), indexing="ij")
    rng = np.random.default_rng(5)
    P = 200
    olat, olon = rng.uniform(-89, 89, P), rng.uniform(0, 359, P)
    olat[:20] = lat2[rng.integers(0, ny, 20), 0]          # on grid latitudes: more ties
    idx, wts, st = ctx.interp_stencils(1, 1, ny, nx, lat2, lon2, [0.0], np.zeros(P, dtype=int), np.zeros(P), olat, olon)
    assert not st.any()
    for k in range(P):
        d = np.hypot(np.sin(np.radians(lat2)) - np.sin(np.radians(olat[k])),
                     np.cos(np.radians(lon2)) - np.cos(np.radians(olon[k]))).reshape(-1)
        ref = np.argsort(d, kind="stable")[:4]
        got = idx[k][idx[k] >= 0]
        assert len(got) == 4
        # same pseudo-distances in the same order (device sin/cos/hypot differ from NumPy's by a few ulp of values
        # ~1, i.e. ~1e-14 of a distance ~1e-2); identical indices where there is no near-tie
        assert np.allclose(d[got], d[ref], rtol=1e-11, atol=1e-15)
        d5 = d[np.argsort(d, kind="stable")[:5]]
        if np.all(np.diff(d5) > 1e-9 * d5[1:]):
            assert sorted(got.tolist()) == sorted(ref.tolist())
        assert abs(wts[k].sum() - 1.0) < 1e-14
