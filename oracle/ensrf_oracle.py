"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.

A NumPy restatement of the reference's serial EnSRF hot path
(lmadaus/efa_xray, `efa_xray/assimilation`).  It exists so that the HIP
path can be checked against the reference's arithmetic on a box where
`/root/reference` does not exist.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this module; the product package `efa_xray_amd` never does (a test
enforces that).

Parity status: PINNED.  Every function here is checked bit-for-bit
against golden vectors produced by running the reference's own modules in
the build container (`tests/golden/make_goldens.py`, fixtures
`tests/golden/G*.npz`; see `tests/test_oracle_golden.py`).  The reference
itself ships no tests or fixtures.

Each function cites the reference file:line it follows (paths relative to
`/root/reference/`).  The arithmetic keeps the reference's operation
order (same NumPy calls on the same shapes) so that, on the same NumPy /
OpenBLAS install, results are bit-identical to the reference.
"""
import numpy as np

EARTH_RADIUS_KM = 6371.0


# ---------------------------------------------------------------------------
# geometry + localisation weights
# ---------------------------------------------------------------------------
def haversine(loc1, loc2):
    """Great-circle km between two (lat, lon) degree pairs.

    Follows efa_xray/observation/observation.py:135-146 (module-level
    `haversine`; the method at efa_xray/state/ensemble.py:241-252 is the same
    arithmetic).
    """
    p1 = np.radians(loc1[0])
    p2 = np.radians(loc2[0])
    dp = p2 - p1
    dl = np.radians(loc2[1] - loc1[1])
    a = np.sin(dp / 2) ** 2 + np.cos(p1) * np.cos(p2) * np.sin(dl / 2) ** 2
    c = 2 * np.arctan2(np.sqrt(a), np.sqrt(1 - a))
    return EARTH_RADIUS_KM * c


def distance_to_point(grid_lat, grid_lon, lat, lon):
    """Great-circle km from every grid point to (lat, lon).

    Follows efa_xray/state/ensemble.py:254-267.  `grid_lat`/`grid_lon` are
    the state's `lat`/`lon` coordinate arrays, 2-D (y, x) or 1-D.
    """
    plat = np.radians(lat)
    plon = np.radians(lon)
    glat = np.radians(grid_lat)
    dlat = plat - glat
    dlon = plon - np.radians(grid_lon)
    a = np.sin(dlat / 2) ** 2 + np.cos(plat) * np.cos(glat) * np.sin(dlon / 2) ** 2
    c = 2 * np.arctan2(np.sqrt(a), np.sqrt(1.0 - a))
    return EARTH_RADIUS_KM * c


def gaspari_cohn(distances, halfwidth):
    """Gaspari-Cohn 5th-order taper.  Follows observation.py:117-130."""
    r = np.divide(distances, abs(halfwidth))
    w = np.zeros(r.shape)
    inner = r <= 1.0
    outer = (r > 1.0) & (r < 2.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        w[inner] = ((((-0.25 * r + 0.5) * r + 0.625) * r - 5.0 / 3.0) * r ** 2 + 1.0)[inner]
        w[outer] = (((((r / 12.0 - 0.5) * r + 0.625) * r + 5.0 / 3.0) * r - 5.0) * r + 4.0
                    - 2.0 / (3.0 * r))[outer]
    return w


def localize_state(grid_lat, grid_lon, ob_lat, ob_lon, halfwidth):
    """Taper of one ob against the state grid.

    Follows observation.py:59-87 for an `EnsembleState` argument:
    `distance_to_point` then `gaspari_cohn`.  Returns an array shaped like
    the lat/lon grid.
    """
    return gaspari_cohn(distance_to_point(grid_lat, grid_lon, ob_lat, ob_lon), halfwidth)


def localize_obs(all_lat, all_lon, ob_lat, ob_lon, halfwidth):
    """Taper of one ob against the list of all obs.

    Follows observation.py:68-83 (the "list of observations" branch: one
    module-level `haversine` call per ob, then `gaspari_cohn`).
    """
    here = (ob_lat, ob_lon)
    d = np.array([haversine(here, s) for s in zip(all_lat, all_lon)])
    return gaspari_cohn(d, halfwidth)


# ---------------------------------------------------------------------------
# formation of the augmented mean / perturbation arrays
# ---------------------------------------------------------------------------
def compute_ob_priors(HX):
    """Obs-space prior means / perturbations from the (P, M) ensemble
    estimates `HX[k] = ob_k.estimate(prior)`.  Follows
    efa_xray/assimilation/assimilation.py:36-49.
    """
    HX = np.asarray(HX, dtype=np.float64)
    P, M = HX.shape
    means = np.zeros(P)
    perts = np.zeros((P, M))
    for k in range(P):
        ye = HX[k]
        means[k] = ye.mean()
        perts[k, :] = ye - ye.mean()
    return means, perts


def format_prior_state(X, HX):
    """(N, M) state-vector ensemble + (P, M) obs estimates -> augmented
    `xbm (A,)`, `Xbp (A, M)`, A = N + P.  Follows assimilation.py:138-150
    (no inflation).
    """
    obmeans, obperts = compute_ob_priors(HX)
    xbm = X.mean(axis=1)
    Xbp = X - xbm[:, None]
    xbm = np.hstack((xbm, obmeans))
    Xbp = np.vstack((Xbp, obperts))
    return xbm, Xbp


def format_posterior_state(xam, Xap, nstate):
    """Rebuild full member values for the state rows.  Follows
    assimilation.py:167-168."""
    return (xam[:, None] + Xap)[:nstate]


def inflate_constant(X, factor):
    """Constant multiplicative inflation of the perturbations about the
    ensemble mean.  Follows assimilation.py:62-68 (float branch): for every
    variable `perts * factor + mean` with mean/perts taken over `mem`."""
    mean = X.mean(axis=-1, keepdims=True)
    return (X - mean) * factor + mean


# ---------------------------------------------------------------------------
# the serial square-root filter loop
# ---------------------------------------------------------------------------
def ensrf_update(xbm, Xbp, nstate, ob_value, ob_error, ob_assim,
                 loc=None, ob_lat=None, ob_lon=None, ob_halfwidth=None,
                 grid_lat=None, grid_lon=None, state_shape=None,
                 faithful_cost=False):
    """Serial EnSRF loop on the augmented state.

    Follows efa_xray/assimilation/ensrf.py:50-149 statement for statement.

    xbm (A,), Xbp (A, M): augmented prior mean / perturbations (not modified).
    nstate: N; rows N..N+P-1 are the obs-space priors.
    ob_value/ob_error/ob_assim: (P,) observed value, error VARIANCE, flag.
    loc: None/False (no localisation) or 'GC'.
    grid_lat/grid_lon: state lat/lon, 2-D (ny, nx) or 1-D (nx,) (ensrf.py:108-111).
    state_shape: (nvar, nt, ny, nx) -- the state's shape without the member
        axis; the taper is broadcast over it (ensrf.py:37-38,108-111).
    faithful_cost: if True also perform the reference's one-hot `np.dot`
        row picks (ensrf.py:61-64,144-145) instead of a direct row read --
        same bits, but the reference's memory traffic; used when this loop is
        timed as the CPU baseline.

    Returns xam (A,), Xap (A, M) and a dict of the per-ob diagnostics the
    reference writes onto each Observation (ensrf.py:66,70,75,146-149):
    prior_mean, prior_var, post_mean, post_var (NaN where the reference
    leaves the attribute untouched) and assimilated (bool).
    """
    xam = xbm
    Xap = Xbp
    A, M = Xap.shape
    P = len(ob_value)
    assert A == nstate + P
    use_loc = loc not in (None, False)
    if use_loc:
        grid_lat = np.asarray(grid_lat, dtype=np.float64)
        grid_lon = np.asarray(grid_lon, dtype=np.float64)
        dum_localize = np.ones(tuple(state_shape))

    prior_mean = np.full(P, np.nan)
    prior_var = np.full(P, np.nan)
    post_mean = np.full(P, np.nan)
    post_var = np.full(P, np.nan)
    assimilated = np.zeros(P, dtype=bool)

    for k in range(P):
        xb = xam
        Xb = Xap
        row = nstate + k
        if faithful_cost:
            H = np.zeros(xam.shape)
            H[row] = 1.0
            mye = np.dot(H, xb)
            ye = np.dot(H, Xb)
        else:
            mye = xb[row]
            ye = Xb[row]
        prior_mean[k] = mye
        varye = np.var(ye)
        prior_var[k] = varye
        if not ob_assim[k]:
            continue
        obs_err = ob_error[k]
        innov = ob_value[k] - mye
        kdenom = varye + obs_err
        kcov = np.dot(Xb, np.transpose(ye)) / (M - 1)
        if use_loc:
            sl = localize_state(grid_lat, grid_lon, ob_lat[k], ob_lon[k], ob_halfwidth[k])
            if sl.ndim == 2:
                sl = (sl[None, None, :, :] * dum_localize).flatten()
            else:
                sl = (sl[None, None, None, :] * dum_localize).flatten()
            ol = localize_obs(ob_lat, ob_lon, ob_lat[k], ob_lon[k], ob_halfwidth[k])
            kcov = np.multiply(np.hstack((sl, ol)), kcov)
        kmat = np.divide(kcov, kdenom)
        xam = xb + np.multiply(kmat, innov)
        beta = 1.0 / (1.0 + np.sqrt(obs_err / (varye + obs_err)))
        kmat = np.multiply(beta, kmat)
        ye2 = np.array(ye)[np.newaxis]
        kmat2 = np.array(kmat)[np.newaxis]
        Xap = Xb - np.dot(kmat2.T, ye2)
        if faithful_cost:
            post_mean[k] = np.dot(H, xam)
            post_var[k] = np.var(np.dot(H, Xap))
        else:
            post_mean[k] = xam[row]
            post_var[k] = np.var(Xap[row])
        assimilated[k] = True

    diag = dict(prior_mean=prior_mean, prior_var=prior_var, post_mean=post_mean,
                post_var=post_var, assimilated=assimilated)
    return xam, Xap, diag


def ensrf_cycle(X, HX, ob_value, ob_error, ob_assim, **kw):
    """format_prior_state -> serial loop -> format_posterior_state
    (ensrf.py:33-151 end to end, given the forward-operator output HX)."""
    N = X.shape[0]
    xbm, Xbp = format_prior_state(X, HX)
    xam, Xap, diag = ensrf_update(xbm, Xbp, N, ob_value, ob_error, ob_assim, **kw)
    return format_posterior_state(xam, Xap, N), xam, Xap, diag


# ---------------------------------------------------------------------------
# forward operator (next-row f1).  PINNED since round 3: the reference's own
# nearest_points / interpolate / Observation.estimate run verbatim in the build
# container on a duck-typed state (tests/golden/make_goldens.py, fixtures G9
# (2-D lat/lon), G10 (1-D lat/lon) and G11 (EnSRF.update() end to end)).
# ---------------------------------------------------------------------------
def nearest_points(grid_lat, grid_lon, lat, lon, npt=1):
    """Indices of the `npt` nearest grid points in the reference's sin/cos
    pseudo-distance.  Follows efa_xray/state/ensemble.py:152-168."""
    d = np.hypot(np.sin(np.radians(grid_lat)) - np.sin(np.radians(lat)),
                 np.cos(np.radians(grid_lon)) - np.cos(np.radians(lon)))
    raw = d.argsort(axis=None)[:npt]
    return np.unravel_index(raw, np.shape(grid_lat))


def interp_space_weights(grid_lat, grid_lon, lat, lon):
    """4-point inverse-distance stencil of ensemble.py:178-200, both lat/lon
    branches.  Returns (iy, ix, weights); for 1-D lat/lon iy = ix = n as in
    ensemble.py:186-188.  The 2-D branch's exact-match case (`distances < 1 km`,
    ensemble.py:194-196) raises IndexError in the reference (a 2-D index into
    a 1-D array); the restatement gives that point weight 1, which is what the
    same statement does in the 1-D branch, where it works (fixture G10)."""
    grid_lat = np.asarray(grid_lat)
    grid_lon = np.asarray(grid_lon)
    if grid_lat.ndim == 2:
        iy, ix = nearest_points(grid_lat, grid_lon, lat, lon, npt=4)
        d = np.array([haversine((grid_lat[y, x], grid_lon[y, x]), (lat, lon))
                      for y, x in zip(list(iy), list(ix))])
    else:
        (n,) = nearest_points(grid_lat, grid_lon, lat, lon, npt=4)
        iy = ix = n
        d = haversine((grid_lat[n], grid_lon[n]), (lat, lon))
    w = np.zeros(d.shape)
    if (d < 1.0).sum() > 0:
        w[d.argmin()] = 1
    else:
        w = 1.0 / d
        w /= w.sum()
    return iy, ix, w


def interp_time_weights(valids, time):
    """Time weights of ensemble.py:203-224 as coded (the weight of the LATER
    valid time is |t - t_later| / dt: swapped, and kept).  `valids` is a
    datetime64 array; None outside the range (ensemble.py:207-209)."""
    time64 = np.datetime64(time)
    valids = np.asarray(valids)
    timeweights = np.zeros(valids.shape)
    if (time64 < valids[0]) or (time64 > valids[-1]):
        return None
    lastdex = (valids >= time64).argmax()
    if valids[lastdex] == time64:
        timeweights[lastdex] = 1
    else:
        diff = (valids[lastdex] - valids[lastdex - 1])
        totsec = np.abs(diff / np.timedelta64(1, 's'))
        thisdiff = time64 - valids[lastdex]
        thissec = np.abs(thisdiff / np.timedelta64(1, 's'))
        timeweights[lastdex] = float(thissec) / totsec
        timeweights[lastdex - 1] = 1.0 - (float(thissec) / totsec)
    return timeweights


def interpolate(field, grid_lat, grid_lon, valids, time, lat, lon):
    """EnsembleState.interpolate (ensemble.py:170-239) on one variable's
    (nt, ny, nx, nmem) array: gather the four points, weight in time (all nt
    slots, zeros included, ensemble.py:229-232), then in space
    (ensemble.py:234-237).  Returns (nmem,) -- the reference's 1-D branch
    returns the same numbers shaped (1, nmem)."""
    iy, ix, sw = interp_space_weights(grid_lat, grid_lon, lat, lon)
    tw = interp_time_weights(valids, time)
    if tw is None:
        return None
    interp = np.asarray(field)[:, iy, ix, :]
    interp = (tw[:, None, None] * interp).sum(axis=0)
    return (sw[:, None] * interp).sum(axis=0)
