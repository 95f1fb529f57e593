"""GPU parity tests: the HIP path (through the C ABI) against the reference's
golden vectors and against the CPU oracle on seeded inputs.

Tolerance (BASELINE.json north_star / SURVEY.md 8d): float64,
max|hip - ref| <= 1e-10 * max|ref| per array and elementwise rtol 1e-10 with
atol 1e-12*max|ref|.
"""
import numpy as np
import pytest

from conftest import load_golden, GOLDEN_CASES
from oracle import ensrf_oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-10


GRAM_DEFAULT = 2  # library default of the "gram" option (2: band leader, with and without localisation)


def _ctx():
    from efa_xray_amd import _lib
    return _lib.get_context(0)


def assert_parity(got, ref, what):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, what
    if ref.size == 0:
        return
    scale = np.nanmax(np.abs(ref))
    assert np.array_equal(np.isnan(got), np.isnan(ref)), what + ": NaN pattern"
    err = np.nanmax(np.abs(got - ref)) if np.isfinite(scale) else 0.0
    assert err <= RTOL * max(scale, 1e-300), "%s: max abs err %.3e vs scale %.3e" % (what, err, scale)
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=1e-12 * max(scale, 1e-300), equal_nan=True,
                               err_msg=what)


def golden_kwargs(g):
    nvar, nt, ny, nx, M = [int(v) for v in g["shape"]]
    kw = dict(loc_mode=0)
    if g["loc"] == "GC":
        glat, glon = g["grid_lat"], g["grid_lon"]
        if glat.ndim == 1:
            glat, glon = np.tile(glat, ny), np.tile(glon, ny)
        kw = dict(loc_mode=1, ob_lat=g["ob_lat"], ob_lon=g["ob_lon"], ob_halfwidth=g["ob_radius"],
                  grid_lat=glat.reshape(-1), grid_lon=glon.reshape(-1), n_lead=nvar * nt)
    return kw


def oracle_kwargs(g):
    nvar, nt, ny, nx, M = [int(v) for v in g["shape"]]
    if g["loc"] != "GC":
        return {}
    return dict(loc="GC", ob_lat=g["ob_lat"], ob_lon=g["ob_lon"], ob_halfwidth=g["ob_radius"],
                grid_lat=g["grid_lat"], grid_lon=g["grid_lon"], state_shape=(nvar, nt, ny, nx))


def prior_arrays(g):
    nvar, nt, ny, nx, M = [int(v) for v in g["shape"]]
    N = nvar * nt * ny * nx
    if "xbm" in g:
        return N, g["xbm"].copy(), g["Xbp"].copy()
    xbm, Xbp = orc.format_prior_state(g["X"].reshape(N, M), g["HX"])
    return N, xbm, Xbp


@pytest.mark.parametrize("name", GOLDEN_CASES)
@pytest.mark.parametrize("path,batch", [("sweep", 32), ("sweep", 1), ("sweep", 7), ("sweep", 64),
                                        ("auto", 32), ("transform", 16)])
def test_host_abi_matches_reference_goldens(name, path, batch):
    """efa_ensrf_update on the reference's augmented arrays == the reference's (xam, Xap)."""
    from efa_xray_amd import _lib
    g = load_golden(name)
    N, xbm, Xbp = prior_arrays(g)
    ctx = _ctx()
    ctx.set_option("obs_batch", batch)
    ctx.set_option("path", {"auto": 0, "sweep": 1, "transform": 2}[path])
    diag = ctx.ensrf_update_host(xbm, Xbp, N, g["ob_value"], g["ob_error"], g["ob_assim"], **golden_kwargs(g))
    ctx.set_option("path", 0)
    ctx.set_option("obs_batch", 64)
    assert_parity(xbm, g["xam"], name + " xam")
    if "Xap" in g:
        assert_parity(Xbp, g["Xap"], name + " Xap")
    post = orc.format_posterior_state(xbm, Xbp, N)
    assert_parity(post, g["post"], name + " post")
    for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
        assert_parity(diag[key], g[key], name + " " + key)
    assert np.array_equal(diag["assimilated"], g["assimilated"])


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_phase_a_per_batch_kernels_and_pipeline_fallback(name):
    """Phase A has two implementations (persistent pipeline / per-batch kernels); both must
    match the reference, and a pipeline whose bounded polls expire must fall back cleanly."""
    g = load_golden(name)
    ctx = _ctx()
    ctx.set_option("path", 1)
    try:
        for mode in ("pipeline", "gram", "band", "batch", "expired", "gram_expired", "band_expired"):
            N, xbm, Xbp = prior_arrays(g)
            ctx.set_option("pipeline", 0 if mode == "batch" else 1)
            ctx.set_option("gram", 2 if mode.startswith("band") else 1 if mode.startswith("gram") else 0)
            ctx.set_option("spin_limit", 1 if mode.endswith("expired") else 4000000)
            diag = ctx.ensrf_update_host(xbm, Xbp, N, g["ob_value"], g["ob_error"], g["ob_assim"],
                                         **golden_kwargs(g))
            kind = ctx.get_option("phase_a_kind")
            if mode == "pipeline":
                assert kind == 1
            if mode == "gram":
                assert kind == 3
            if mode == "band":
                assert kind == 4   # the band leader also covers Gaspari-Cohn cycles (taper corner in LDS)
            if mode == "batch":
                assert kind == 2
            assert_parity(xbm, g["xam"], "%s %s xam" % (name, mode))
            for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
                assert_parity(diag[key], g[key], "%s %s %s" % (name, mode, key))
            assert np.array_equal(diag["assimilated"], g["assimilated"])
    finally:
        ctx.set_option("pipeline", 1)
        ctx.set_option("gram", GRAM_DEFAULT)
        ctx.set_option("spin_limit", 4000000)
        ctx.set_option("path", 0)


def _make_api_objects(g):
    from efa_xray_amd import EnsembleState, Observation

    class StencilOb(Observation):
        def estimate(self, state):
            rows = state.to_vect()[self.idx]
            if len(self.idx) == 1 and self.wts[0] == 1.0:
                return rows[0].copy()
            return (self.wts[:, None] * rows).sum(axis=0)

    state = EnsembleState.from_array(g["X"], g["grid_lat"], g["grid_lon"])
    obs = []
    for k in range(len(g["ob_value"])):
        nz = g["sten_wts"][k] != 0
        ob = StencilOb(value=float(g["ob_value"][k]), error=float(g["ob_error"][k]),
                       lat=float(g["ob_lat"][k]), lon=float(g["ob_lon"][k]),
                       assimilate_this=bool(g["ob_assim"][k]),
                       localize_radius=(None if np.isnan(g["ob_radius"][k]) else float(g["ob_radius"][k])))
        ob.idx = g["sten_idx"][k][nz]
        ob.wts = g["sten_wts"][k][nz]
        obs.append(ob)
    return state, obs


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_python_api_matches_reference_goldens(name):
    """EnSRF(state, obs, loc=...).update() == the reference's update() outputs."""
    from efa_xray_amd import EnSRF
    g = load_golden(name)
    state, obs = _make_api_objects(g)
    X_before = state.to_vect().copy()
    post_state, obs_out = EnSRF(state, obs, verbose=False, loc=(g["loc"] or False)).update()
    assert obs_out is obs
    assert post_state is not state
    assert np.array_equal(state.to_vect(), X_before), "prior must not be modified"
    assert_parity(post_state.to_vect(), g["post"], name + " post")
    for key in ("prior_mean", "prior_var"):
        assert_parity([getattr(o, key) for o in obs], g[key], name + " " + key)
    for k, o in enumerate(obs):
        assert bool(o.assimilated) == bool(g["assimilated"][k])
        if o.assimilated:
            assert abs(o.post_mean - g["post_mean"][k]) <= RTOL * max(1.0, abs(g["post_mean"][k]))
            assert abs(o.post_var - g["post_var"][k]) <= RTOL * max(1.0, abs(g["post_var"][k]))
        else:
            assert o.post_mean is None and o.post_var is None


def test_format_prior_and_posterior_helpers():
    from efa_xray_amd import EnSRF
    g = load_golden("G2")
    state, obs = _make_api_objects(g)
    flt = EnSRF(state, obs, verbose=False, loc="GC")
    xbm, Xbp = flt.format_prior_state()
    assert_parity(xbm, g["xbm"], "xbm")
    assert_parity(Xbp, g["Xbp"], "Xbp")
    xam, Xap = flt.update_arrays(xbm, Xbp)
    assert_parity(xam, g["xam"], "xam")
    assert_parity(Xap, g["Xap"], "Xap")
    post_state, _ = flt.format_posterior_state(xam, Xap)
    assert_parity(post_state.to_vect(), g["post"], "post")


# ---------------------------------------------------------------------------
# seeded inputs vs the oracle: shapes the goldens do not cover
# ---------------------------------------------------------------------------
def _random_case(seed, N, M, P, loc, frac_assim=0.9, ncol=None):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, 1)) + 3.0 * rng.standard_normal((N, M))
    rows = rng.choice(N, P, replace=(P > N))
    HX = X[rows]
    val = HX.mean(axis=1) + rng.standard_normal(P)
    err = rng.uniform(0.5, 2.0, P)
    asm = rng.random(P) < frac_assim
    case = dict(X=X, HX=HX, val=val, err=err, asm=asm, N=N, M=M, P=P, loc=loc)
    if loc:
        ncol = ncol or N
        assert N % ncol == 0
        ny = int(np.sqrt(ncol))
        while ncol % ny:
            ny -= 1
        nx = ncol // ny
        lat, lon = np.meshgrid(np.linspace(-70, 70, ny), np.linspace(0, 357, nx), indexing="ij")
        case.update(lat=lat, lon=lon, n_lead=N // ncol, ny=ny, nx=nx,
                    ob_lat=lat.reshape(-1)[rows % ncol] + 0.2 * rng.standard_normal(P),
                    ob_lon=lon.reshape(-1)[rows % ncol] + 0.2 * rng.standard_normal(P),
                    hw=rng.uniform(800, 4000, P))
    return case


def _run_oracle(c):
    kw = {}
    if c["loc"]:
        kw = dict(loc="GC", ob_lat=c["ob_lat"], ob_lon=c["ob_lon"], ob_halfwidth=c["hw"], grid_lat=c["lat"],
                  grid_lon=c["lon"], state_shape=(c["n_lead"], 1, c["ny"], c["nx"]))
    xbm, Xbp = orc.format_prior_state(c["X"], c["HX"])
    xam, Xap, diag = orc.ensrf_update(xbm, Xbp, c["N"], c["val"], c["err"], c["asm"], **kw)
    return xam, Xap, diag


def _run_hip(c, path="auto", batch=32, pipeline=None):
    """pipeline: None library default, 0 per-batch kernels, 1 persistent kernel (vector chain),
    2 persistent kernel (Gram leader), 3 persistent kernel (band leader, with and without localisation)"""
    ctx = _ctx()
    ctx.set_option("obs_batch", batch)
    if pipeline is not None:
        ctx.set_option("pipeline", 1 if pipeline else 0)
        ctx.set_option("gram", 2 if pipeline == 3 else 1 if pipeline == 2 else 0)
    ctx.set_option("path", {"auto": 0, "sweep": 1, "transform": 2}[path])
    xbm, Xbp = orc.format_prior_state(c["X"], c["HX"])
    kw = dict(loc_mode=0)
    if c["loc"]:
        kw = dict(loc_mode=1, ob_lat=c["ob_lat"], ob_lon=c["ob_lon"], ob_halfwidth=c["hw"],
                  grid_lat=c["lat"].reshape(-1), grid_lon=c["lon"].reshape(-1), n_lead=c["n_lead"])
    diag = ctx.ensrf_update_host(xbm, Xbp, c["N"], c["val"], c["err"], c["asm"], **kw)
    ctx.set_option("path", 0)
    ctx.set_option("obs_batch", 64)
    ctx.set_option("pipeline", 1)
    ctx.set_option("gram", GRAM_DEFAULT)
    return xbm, Xbp, diag


SHAPES = [
    # N, M, P, loc
    (1, 2, 1, False), (5, 2, 3, False), (63, 3, 9, False), (65, 7, 70, False),
    (1000, 21, 33, False), (257, 100, 150, False), (4096, 50, 64, False), (300, 128, 140, False),
    (513, 130, 20, False), (200, 200, 12, False), (129, 256, 5, False),
    (48, 6, 10, True), (1024, 20, 90, True), (2304, 80, 130, True), (900, 33, 40, True),
]


@pytest.mark.parametrize("N,M,P,loc", SHAPES)
def test_seeded_shapes_vs_oracle(N, M, P, loc):
    c = _random_case(100 + N + M + P, N, M, P, loc, ncol=(N // 4 if loc and N % 4 == 0 and N >= 1024 else None))
    xam, Xap, diag = _run_oracle(c)
    for path, pipe in ((("sweep", 1), ("auto", 1), ("sweep", 0), ("sweep", 2), ("auto", 2), ("sweep", 3), ("auto", 3)) if not loc
                       else (("sweep", 1), ("sweep", 0), ("sweep", 2), ("sweep", 3))):
        h_xam, h_Xap, h_diag = _run_hip(c, path=path, pipeline=pipe)
        assert_parity(h_xam, xam, "xam %s" % path)
        assert_parity(h_Xap, Xap, "Xap %s" % path)
        for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
            assert_parity(h_diag[key], diag[key], key)
        assert np.array_equal(h_diag["assimilated"], diag["assimilated"])


def test_gram_leader_cancellation_guard_falls_back():
    """Near-exact, repeated observations of one quantity collapse the variance of the rows that follow by
    far more than 1e3 inside one 64-ob block: the Gram-space leader must notice (its downdated G_kk has
    lost digits), bail out, and the vector-chain pipeline must deliver the reference's numbers."""
    c = _random_case(77, 300, 24, 90, False, frac_assim=1.0)
    c["HX"][1:40] = c["HX"][0] + 1e-4 * np.random.default_rng(3).standard_normal((39, 24))  # 40 near-copies of ob 0
    c["val"][:40] = c["HX"][0].mean() + 0.1
    c["err"][:40] = 1e-8
    xam, Xap, diag = _run_oracle(c)
    ctx = _ctx()
    for pipeline, gram in ((2, 1), (3, 2)):
        h_xam, h_Xap, h_diag = _run_hip(c, path="sweep", pipeline=pipeline)
        assert_parity(h_xam, xam, "xam")
        assert_parity(h_Xap, Xap, "Xap")
        for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
            assert_parity(h_diag[key], diag[key], key)
        # and the guard did trip: the last Phase A was done by the vector-chain kernel
        ctx.set_option("gram", gram)
        ctx.set_option("path", 1)
        xbm, Xbp = orc.format_prior_state(c["X"], c["HX"])
        ctx.ensrf_update_host(xbm, Xbp, c["N"], c["val"], c["err"], c["asm"], loc_mode=0)
        kind = ctx.get_option("phase_a_kind")
        ctx.set_option("gram", GRAM_DEFAULT)
        ctx.set_option("path", 0)
        assert kind == 1


def test_edge_cases_no_obs_and_none_assimilated():
    c = _random_case(5, 300, 20, 12, False, frac_assim=0.0)
    xbm, Xbp = orc.format_prior_state(c["X"], c["HX"])
    h_xam, h_Xap, h_diag = _run_hip(c)
    assert np.array_equal(h_xam, xbm) and np.array_equal(h_Xap, Xbp)   # untouched, bit for bit
    assert not h_diag["assimilated"].any()
    assert np.all(np.isnan(h_diag["post_mean"]))
    assert_parity(h_diag["prior_var"], np.var(Xbp[300:], axis=1), "prior_var")
    # P = 0
    ctx = _ctx()
    x0, X0 = xbm[:300].copy(), Xbp[:300].copy()
    d = ctx.ensrf_update_host(x0, X0, 300, np.zeros(0), np.zeros(0), np.zeros(0, dtype=bool))
    assert np.array_equal(x0, xbm[:300]) and np.array_equal(X0, Xbp[:300])
    assert d["prior_mean"].shape == (0,)


def test_unassimilated_obs_leave_state_untouched_and_zero_taper_rows_bit_unchanged():
    # F3: rows whose taper is 0 for every ob are returned bit-identical
    c = _random_case(9, 4096, 40, 30, True, ncol=1024)
    c["hw"][:] = 150.0                       # tiny footprints: most rows outside 2*halfwidth
    xbm, Xbp = orc.format_prior_state(c["X"], c["HX"])
    h_xam, h_Xap, _ = _run_hip(c, path="sweep")
    w = np.zeros(1024, dtype=bool)
    for k in range(c["P"]):
        if c["asm"][k]:
            w |= orc.localize_state(c["lat"], c["lon"], c["ob_lat"][k], c["ob_lon"][k], c["hw"][k]).reshape(-1) != 0
    untouched = np.tile(~w, c["n_lead"])
    assert untouched.sum() > 0
    assert np.array_equal(h_Xap[:c["N"]][untouched], Xbp[:c["N"]][untouched])
    assert np.array_equal(h_xam[:c["N"]][untouched], xbm[:c["N"]][untouched])
    xam, Xap, _ = _run_oracle(c)
    assert_parity(h_Xap, Xap, "Xap")


def test_posterior_obs_variance_identity():
    # closed form of the ob's own row: ye' = ye*(1 - beta*kmat) with the reference's mixed
    # conventions (np.var is ddof=0, the covariance divides by M-1: ensrf.py:69,95,135)
    c = _random_case(21, 500, 30, 25, False, frac_assim=1.0)
    _, _, d = _run_hip(c)
    M = c["M"]
    var, R = d["prior_var"], c["err"]
    kmat = (var * M / (M - 1)) / (var + R)
    beta = 1.0 / (1.0 + np.sqrt(R / (var + R)))
    np.testing.assert_allclose(d["post_var"], var * (1.0 - beta * kmat) ** 2, rtol=1e-9)


def test_logical_shards_equal_unsharded_bit_for_bit():
    """Row independence (F1): updating two column shards separately with the
    replicated obs block gives exactly the bits of the unsharded update."""
    c = _random_case(33, 3 * 800, 50, 70, True, ncol=800)
    ctx = _ctx()
    ctx.set_option("path", 1)
    N, M, P, ncol, L = c["N"], c["M"], c["P"], 800, 3
    xbm, Xbp = orc.format_prior_state(c["X"], c["HX"])
    full_x, full_X, _ = _run_hip(c, path="sweep", batch=64)
    lat, lon = c["lat"].reshape(-1), c["lon"].reshape(-1)
    out_x = np.empty(N)
    out_X = np.empty((N, M))
    for lo, hi in ((0, 333), (333, 800)):
        cols = np.arange(lo, hi)
        rows = (np.arange(L)[:, None] * ncol + cols[None, :]).reshape(-1)
        xs = np.ascontiguousarray(np.hstack((xbm[rows], xbm[N:])))
        Xs = np.ascontiguousarray(np.vstack((Xbp[rows], Xbp[N:])))
        ctx.ensrf_update_host(xs, Xs, len(rows), c["val"], c["err"], c["asm"], loc_mode=1, ob_lat=c["ob_lat"],
                              ob_lon=c["ob_lon"], ob_halfwidth=c["hw"], grid_lat=lat[lo:hi], grid_lon=lon[lo:hi],
                              n_lead=L)
        out_x[rows] = xs[:len(rows)]
        out_X[rows] = Xs[:len(rows)]
    ctx.set_option("path", 0)
    assert np.array_equal(out_x, full_x[:N]) and np.array_equal(out_X, full_X[:N])


def test_sweep_and_transform_paths_agree_at_scale():
    """1e6 rows x 100 members x 400 obs: the two Phase-B paths agree to 1e-11
    and an encode/decode style round trip (posterior of posterior-free obs) holds."""
    from efa_xray_amd import _lib
    ctx = _ctx()
    rows, M, P = 1_000_000, 100, 400
    rng = np.random.default_rng(4)
    X = ctx.empty((rows, M))
    ctx.fill_synthetic(rows, 0, M, 1234, 3.0, X)
    pick = np.sort(rng.choice(rows, P, replace=False)).astype(np.int64)
    HX = ctx.empty((P, M))
    ctx.forward_stencil(rows, 0, M, X, pick[:, None], np.ones((P, 1)), HX)
    hx = HX.download()
    val = hx.mean(axis=1) + rng.standard_normal(P)
    err = np.ones(P)
    asm = np.ones(P, dtype=bool)
    outs = {}
    for path in (1, 2):
        ctx.set_option("path", path)
        ym = ctx.empty((P,))
        Yp = ctx.to_device(hx)
        ctx.form_perts(P, M, Yp, ym, Yp)
        d = ctx.obs_phase(M, P, ym, Yp, val, err, asm)
        post = ctx.empty((rows, M))
        ctx.state_cycle(rows, M, X, post)
        assert ctx.last_timing()["path"] == path
        outs[path] = (post.download(), d)
        post.free()
    ctx.set_option("path", 0)
    a, b = outs[1][0], outs[2][0]
    scale = np.abs(a).max()
    assert np.abs(a - b).max() <= 1e-11 * scale
    # posterior at the observed rows reproduces the diagnostics' post_mean / post_var
    d = outs[1][1]
    sub = a[pick[-1]]
    assert abs(sub.mean() - d["post_mean"][-1]) <= 1e-10 * max(1.0, abs(d["post_mean"][-1]))
    assert abs(sub.var() - d["post_var"][-1]) <= 1e-9 * max(1.0, d["post_var"][-1])
    # oracle on a thin slice of rows with the same obs trajectory
    sl = np.concatenate([np.arange(0, 2000), pick])
    Xs = X.download()[sl]
    ref_post, _, _, rd = orc.ensrf_cycle(Xs, hx, val, err, asm)
    assert_parity(a[sl], ref_post, "slice post vs oracle")
    assert_parity(d["post_var"], rd["post_var"], "post_var")


def test_headline_size_properties():
    """BASELINE.json's full size (1e7 state rows x 100 members x 1e4 obs, float64, loc=None): properties that
    do not need an oracle run of that size.
      - the one-pass transform and the per-batch sweeps (157 read+write passes) give the same posterior;
      - no row's ensemble variance grows (each serial EnSRF update shrinks it);
      - the row observed by the LAST ob reproduces that ob's post_mean / post_var;
      - a cycle in which no ob is assimilated returns mean + (prior - mean), as the reference does
        (assimilation.py:146-147,168), i.e. the prior to rounding."""
    ctx = _ctx()
    rows, M, P = 10_000_000, 100, 10_000
    rng = np.random.default_rng(11)
    X = ctx.empty((rows, M))
    ctx.fill_synthetic(rows, 0, M, 4321, 3.0, X)
    pick = np.sort(rng.choice(rows, P, replace=False)).astype(np.int64)
    HX = ctx.empty((P, M))
    ctx.forward_stencil(rows, 0, M, X, pick[:, None], np.ones((P, 1)), HX)
    hx = HX.download()
    val = hx.mean(axis=1) + rng.standard_normal(P)
    err = np.ones(P)
    blocks = [(int(r0), int(r0) + 4096) for r0 in rng.choice(rows - 4096, 12, replace=False)] + [(rows - 4096, rows)]
    post = ctx.empty((rows, M))
    got = {}
    try:
        for path in (2, 1):
            ctx.set_option("path", path)
            ym = ctx.empty((P,))
            Yp = ctx.to_device(hx)
            ctx.form_perts(P, M, Yp, ym, Yp)
            d = ctx.obs_phase(M, P, ym, Yp, val, err, np.ones(P, dtype=bool))
            ctx.state_cycle(rows, M, X, post)
            assert ctx.last_timing()["path"] == path
            got[path] = ([post.download_rows(a, b) for a, b in blocks], post.download_rows(int(pick[-1]), int(pick[-1]) + 1)[0], d)
        for (a, b), pa, pb, in zip(blocks, got[2][0], got[1][0]):
            prior = X.download_rows(a, b)
            assert np.isfinite(pa).all()
            assert np.abs(pa - pb).max() <= 1e-10 * np.abs(pb).max(), "transform vs sweep rows %d..%d" % (a, b)
            assert (pa.var(axis=1) <= prior.var(axis=1) * (1.0 + 1e-9)).all(), "variance grew in rows %d..%d" % (a, b)
        for path in (1, 2):
            last, d = got[path][1], got[path][2]
            assert abs(last.mean() - d["post_mean"][-1]) <= 1e-10 * max(1.0, abs(d["post_mean"][-1]))
            assert abs(last.var() - d["post_var"][-1]) <= 1e-9 * max(1.0, d["post_var"][-1])
        assert_parity(got[2][2]["prior_var"], got[1][2]["prior_var"], "prior_var, transform vs sweep trajectory")
        # nothing assimilated: the posterior members are the prior members (re-centred and restored)
        ctx.set_option("path", 0)
        ym = ctx.empty((P,))
        Yp = ctx.to_device(hx)
        ctx.form_perts(P, M, Yp, ym, Yp)
        d0 = ctx.obs_phase(M, P, ym, Yp, val, err, np.zeros(P, dtype=bool))
        ctx.state_cycle(rows, M, X, post)
        assert not d0["assimilated"].any()
        for a, b in blocks[:4]:
            pr = X.download_rows(a, b)
            assert np.abs(post.download_rows(a, b) - pr).max() <= 4e-15 * np.abs(pr).max()
    finally:
        ctx.set_option("path", 0)
        post.free()
        X.free()


def _headline_obs_sets(M, P):
    """Two obs sets of the headline's shape: "uncorrelated" -- the bench's construction (row picks of a sigma = 3 field) --
    and "correlated": every ob sees the same 12 smooth modes plus 5 % noise, so the 64 x 64 Gram blocks are close to
    rank 12 and each pivot removes a large part of the remaining variance (the downdate's hard case)."""
    rng = np.random.default_rng(77)
    sets = {}
    sets["uncorrelated"] = 3.0 * rng.standard_normal((P, M)) + rng.standard_normal((P, 1))
    t = np.linspace(0.0, 1.0, P)[:, None]
    modes = np.concatenate([np.cos(np.pi * j * t) for j in range(12)], axis=1)          # (P, 12), smooth in k
    sets["correlated"] = modes @ rng.standard_normal((12, M)) + 0.05 * rng.standard_normal((P, M)) + 0.3
    return sets


def test_headline_phase_a_and_transform_vs_oracle():
    """The headline's Phase A (1e4 obs x 100 members, one persistent launch, band leader) and the [T | w] it hands
    to the transform, DIRECTLY against the oracle (ensrf.py:50-149 run on the obs block with the M identity rows and
    2 000 state rows as the "state": then Xap[:M] = T, xam[:M] = w).  Compared: all four diagnostics of every ob, the
    final obs block, T and w as `k_transform` applies them (unfused form on the identity rows), and the fused
    prior-members -> posterior-members form on the 2 000 rows.  Two obs sets, see `_headline_obs_sets`; the oracle runs
    of both (about a minute each) go side by side on two host threads."""
    from concurrent.futures import ThreadPoolExecutor
    ctx = _ctx()
    M, P, R = 100, 10_000, 2_000
    rng = np.random.default_rng(78)
    sets = _headline_obs_sets(M, P)
    Xs = rng.standard_normal((R, 1)) + 3.0 * rng.standard_normal((R, M))
    Xs[:200] = sets["correlated"][:200] + 0.5 * rng.standard_normal((200, M))       # rows that the obs really constrain
    xsm = Xs.mean(axis=1)
    Xsp = Xs - xsm[:, None]
    val, err, asm = {}, {}, {}
    for name, HX in sets.items():
        val[name] = HX.mean(axis=1) + rng.standard_normal(P)
        err[name] = rng.uniform(0.5, 2.0, P) if name == "correlated" else np.ones(P)
        asm[name] = np.ones(P, dtype=bool)
    asm["correlated"][rng.choice(P, 300, replace=False)] = False

    def run_oracle(name):
        ym0, Yp0 = orc.compute_ob_priors(sets[name])
        xbm = np.hstack((np.zeros(M), xsm, ym0))
        Xbp = np.vstack((np.eye(M), Xsp, Yp0))
        return orc.ensrf_update(xbm, Xbp, M + R, val[name], err[name], asm[name])

    with ThreadPoolExecutor(2) as pool:
        futs = dict((name, pool.submit(run_oracle, name)) for name in sets)
        got = {}
        try:
            for name, HX in sets.items():
                ctx.set_option("path", 2)
                Yp = ctx.to_device(HX)
                ym = ctx.empty((P,))
                ctx.form_perts(P, M, Yp, ym, Yp)
                d = ctx.obs_phase(M, P, ym, Yp, val[name], err[name], asm[name])
                kind = ctx.get_option("phase_a_kind")
                # unfused transform on [I ; state perturbations]: rows 0..M-1 come out as T (and w in the mean)
                xm = ctx.to_device(np.hstack((np.zeros(M), xsm)))
                Xp = ctx.to_device(np.vstack((np.eye(M), Xsp)))
                ctx.state_phase(M + R, M, xm, Xp, xm, Xp)
                assert ctx.last_timing()["path"] == 2
                # fused form: prior members in, posterior members out
                Xd = ctx.to_device(Xs)
                post = ctx.empty((R, M))
                ctx.state_cycle(R, M, Xd, post)
                got[name] = (d, kind, Yp.download(), ym.download(), xm.download(), Xp.download(), post.download())
        finally:
            ctx.set_option("path", 0)
        for name in sets:
            xam, Xap, rd = futs[name].result()
            d, kind, Yp_f, ym_f, xm_f, Xp_f, post = got[name]
            print("%s: phase_a_kind %d, prior_var range %.3g..%.3g" % (name, kind, rd["prior_var"].min(), rd["prior_var"].max()))
            assert kind == 4 or name == "correlated", "the band leader did not serve the bench's obs set"
            for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
                assert_parity(d[key], rd[key], "%s %s" % (name, key))
            assert np.array_equal(d["assimilated"], rd["assimilated"])
            assert_parity(Yp_f, Xap[M + R:], name + " final obs perturbations")
            assert_parity(ym_f, xam[M + R:], name + " final obs means")
            assert_parity(Xp_f[:M], Xap[:M], name + " T")
            assert_parity(xm_f[:M], xam[:M], name + " w")
            assert_parity(Xp_f[M:], Xap[M:M + R], name + " state perturbations (transform)")
            assert_parity(xm_f[M:], xam[M:M + R], name + " state means (transform)")
            assert_parity(post, orc.format_posterior_state(xam[M:M + R], Xap[M:M + R], R), name + " posterior members (fused)")


def test_persistent_phase_a_at_its_residency_limit():
    """16 284 obs x 100 members (+100 carried transform rows = 16 384 rows = 256 co-resident workgroups, one per
    CU): the persistent Phase-A launch (both leaders) against the per-batch kernels."""
    ctx = _ctx()
    M, P = 100, 16284
    rng = np.random.default_rng(21)
    HX = 3.0 * rng.standard_normal((P, M))
    val = HX.mean(axis=1) + rng.standard_normal(P)
    err = rng.uniform(0.5, 2.0, P)
    asm = rng.random(P) < 0.95
    res = {}
    try:
        for name, pipe, gram in (("batch", 0, 0), ("chain", 1, 0), ("gram", 1, 1), ("band", 1, 2)):
            ctx.set_option("pipeline", pipe)
            ctx.set_option("gram", gram)
            ctx.set_option("path", 0)
            Yp = ctx.to_device(HX)
            ym = ctx.empty((P,))
            ctx.form_perts(P, M, Yp, ym, Yp)
            d = ctx.obs_phase(M, P, ym, Yp, val, err, asm)
            assert ctx.get_option("phase_a_kind") == {"batch": 2, "chain": 1, "gram": 3, "band": 4}[name]
            res[name] = (Yp.download(), ym.download(), d)
        for name in ("chain", "gram", "band"):
            assert_parity(res[name][0], res["batch"][0], name + " obs perturbations")
            assert_parity(res[name][1], res["batch"][1], name + " obs means")
            for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
                assert_parity(res[name][2][key], res["batch"][2][key], name + " " + key)
            assert np.array_equal(res[name][2]["assimilated"], res["batch"][2]["assimilated"])
    finally:
        ctx.set_option("pipeline", 1)
        ctx.set_option("gram", GRAM_DEFAULT)
        ctx.set_option("path", 0)


@pytest.mark.parametrize("loc", [False, True])
def test_phase_a_in_windows_beyond_one_persistent_launch(loc):
    """More observations than one persistent launch can hold (256 workgroups x 64 rows): Phase A runs window by
    window -- each window one persistent launch, every other row of the obs block taking the window's records through the
    sweep kernel -- and must give what the per-batch kernels give for all P obs (and, on a prefix, what the oracle gives).
    40 000 obs x 100 members without localisation (three windows, the carried transform rows in each), 20 000 x 40 with
    Gaspari-Cohn (two windows, obs-obs taper)."""
    import time
    ctx = _ctx()
    M, P = (100, 40000) if not loc else (40, 20000)
    rng = np.random.default_rng(91)
    HX = 3.0 * rng.standard_normal((P, M)) + rng.standard_normal((P, 1))
    val = HX.mean(axis=1) + rng.standard_normal(P)
    err = rng.uniform(0.5, 2.0, P)
    asm = rng.random(P) < 0.95
    kw = {}
    if loc:
        kw = dict(loc_mode=1, ob_lat=rng.uniform(-60, 60, P), ob_lon=rng.uniform(0, 360, P), ob_halfwidth=rng.uniform(300, 900, P))
    res = {}
    try:
        for name, pipe in (("batch", 0), ("windows", 1)):
            ctx.set_option("pipeline", pipe)
            ctx.set_option("path", 1 if loc else 2)
            Yp = ctx.to_device(HX)
            ym = ctx.empty((P,))
            ctx.form_perts(P, M, Yp, ym, Yp)
            ctx.synchronize()
            t0 = time.perf_counter()
            d = ctx.obs_phase(M, P, ym, Yp, val, err, asm, **kw)
            dt = time.perf_counter() - t0
            kind = ctx.get_option("phase_a_kind")
            assert kind == (2 if pipe == 0 else 4), (name, kind)
            res[name] = (Yp.download(), ym.download(), d)
            print("%s: %.2f us per ob (host wall time of the call, P = %d, M = %d, loc = %s)" % (name, 1e6 * dt / P, P, M, loc))
            if name == "windows" and not loc:     # the transform the windows leave behind serves Phase B
                X = rng.standard_normal((500, M))
                xm = ctx.to_device(X.mean(axis=1))
                Xp = ctx.to_device(X - X.mean(axis=1, keepdims=True))
                ctx.state_phase(500, M, xm, Xp, xm, Xp)
                res["T"] = (xm.download(), Xp.download(), X)
        a, b = res["windows"], res["batch"]
        assert_parity(a[0], b[0], "final obs perturbations, windows vs per-batch")
        assert_parity(a[1], b[1], "final obs means")
        for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
            assert_parity(a[2][key], b[2][key], key)
        assert np.array_equal(a[2]["assimilated"], b[2]["assimilated"])
        # oracle on a prefix (obs k is only influenced by earlier obs): 400 obs, with 500 state rows riding along
        n = 400
        ym0, Yp0 = orc.compute_ob_priors(HX[:n])
        okw = {}
        if loc:
            okw = dict(loc="GC", ob_lat=kw["ob_lat"][:n], ob_lon=kw["ob_lon"][:n], ob_halfwidth=kw["ob_halfwidth"][:n],
                       grid_lat=np.zeros((1, 0)), grid_lon=np.zeros((1, 0)), state_shape=(1, 1, 1, 0))
        _, _, od = orc.ensrf_update(ym0, Yp0, 0, val[:n], err[:n], asm[:n], **okw)
        for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
            assert_parity(a[2][key][:n], od[key], "oracle prefix " + key)
        if not loc:   # the carried transform after three windows == the sweep of all obs over the same rows (per-batch trajectory)
            xm_t, Xp_t, X = res["T"]
            ctx.set_option("pipeline", 0)
            ctx.set_option("path", 1)
            Yp = ctx.to_device(HX)
            ym = ctx.empty((P,))
            ctx.form_perts(P, M, Yp, ym, Yp)
            ctx.obs_phase(M, P, ym, Yp, val, err, asm)
            xm = ctx.to_device(X.mean(axis=1))
            Xp = ctx.to_device(X - X.mean(axis=1, keepdims=True))
            ctx.state_phase(500, M, xm, Xp, xm, Xp)
            assert_parity(Xp_t, Xp.download(), "state rows: transform carried through three windows vs sweep")
            assert_parity(xm_t, xm.download(), "state means")
    finally:
        ctx.set_option("pipeline", 1)
        ctx.set_option("gram", GRAM_DEFAULT)
        ctx.set_option("path", 0)


def test_a_window_that_gives_up_is_redone_by_the_per_batch_kernels():
    """Two windows (17 000 obs x 100 members).  The second holds 40 near-exact copies of one observation with a tiny
    error variance: the band leader's cancellation guard trips there (as in
    `test_gram_leader_cancellation_guard_falls_back`).  The first window has already left band-layout records, so the
    vector-chain kernel (another record layout) is not an option: the window is redone by the per-batch kernels for its
    own obs only, its ye rows are copied into the records' layout, and the whole call -- final obs block, diagnostics, the
    carried transform, the sweep over the recorded trajectory -- must equal the per-batch kernels' result for all P obs."""
    ctx = _ctx()
    M, P = 100, 17000
    rng = np.random.default_rng(123)
    HX = 3.0 * rng.standard_normal((P, M)) + rng.standard_normal((P, 1))
    k0 = 16400                                       # inside the second window (it starts at 16 184)
    HX[k0 + 1:k0 + 40] = HX[k0] + 1e-4 * rng.standard_normal((39, M))
    val = HX.mean(axis=1) + rng.standard_normal(P)
    val[k0:k0 + 40] = HX[k0].mean() + 0.1
    err = np.ones(P)
    err[k0:k0 + 40] = 1e-8
    asm = np.ones(P, dtype=bool)
    X = rng.standard_normal((600, M))
    res = {}
    try:
        for name, pipe in (("batch", 0), ("windows", 1)):
            for path in (2, 1):
                ctx.set_option("pipeline", pipe)
                ctx.set_option("path", path)
                Yp = ctx.to_device(HX)
                ym = ctx.empty((P,))
                ctx.form_perts(P, M, Yp, ym, Yp)
                d = ctx.obs_phase(M, P, ym, Yp, val, err, asm)
                kind = ctx.get_option("phase_a_kind")
                xm = ctx.to_device(X.mean(axis=1))
                Xp = ctx.to_device(X - X.mean(axis=1, keepdims=True))
                ctx.state_phase(600, M, xm, Xp, xm, Xp)
                assert ctx.last_timing()["path"] == path
                res[name, path] = (Yp.download(), ym.download(), d, xm.download(), Xp.download(), kind)
        assert res["batch", 2][5] == 2 and res["windows", 2][5] == 4      # the first window did run as the band leader
        ref = res["batch", 1]
        for key in (("windows", 2), ("windows", 1), ("batch", 2)):
            got = res[key]
            assert_parity(got[0], ref[0], "%s: final obs perturbations" % (key,))
            assert_parity(got[1], ref[1], "%s: final obs means" % (key,))
            for dk in ("prior_mean", "prior_var", "post_mean", "post_var"):
                assert_parity(got[2][dk], ref[2][dk], "%s: %s" % (key, dk))
            assert_parity(got[3], ref[3], "%s: state means" % (key,))
            assert_parity(got[4], ref[4], "%s: state perturbations" % (key,))
        # the guard really tripped in the second window: run its obs alone (one window) and see the fallback kind
        ctx.set_option("pipeline", 1)
        ctx.set_option("path", 1)
        sub = slice(16184, P)
        Yp = ctx.to_device(HX[sub])
        ym = ctx.empty((P - 16184,))
        ctx.form_perts(P - 16184, M, Yp, ym, Yp)
        ctx.obs_phase(M, P - 16184, ym, Yp, val[sub], err[sub], asm[sub])
        assert ctx.get_option("phase_a_kind") == 1
    finally:
        ctx.set_option("pipeline", 1)
        ctx.set_option("gram", GRAM_DEFAULT)
        ctx.set_option("path", 0)


def test_helper_kernels_vs_oracle():
    ctx = _ctx()
    rng = np.random.default_rng(8)
    for rows, M in ((1, 2), (77, 5), (1000, 50), (333, 100), (64, 256)):
        X = rng.standard_normal((rows, M)) * 5 + rng.standard_normal((rows, 1))
        d = ctx.to_device(X)
        xm = ctx.empty((rows,))
        Xp = ctx.empty((rows, M))
        ctx.form_perts(rows, M, d, xm, Xp)
        assert_parity(xm.download(), X.mean(axis=1), "mean")
        assert_parity(Xp.download(), X - X.mean(axis=1)[:, None], "perts")
        ctx.form_perts(rows, M, d, xm, Xp, scale=1.3)
        assert_parity(Xp.download(), (X - X.mean(axis=1)[:, None]) * 1.3, "inflated perts")
        post = ctx.empty((rows, M))
        ctx.posterior(rows, M, xm, Xp, post)
        assert_parity(post.download(), orc.inflate_constant(X, 1.3), "posterior/inflate")
    # forward stencil incl. sharding by row ranges
    rows, M, P = 500, 20, 40
    X = rng.standard_normal((rows, M))
    idx = rng.integers(0, rows, (P, 4))
    wts = rng.random((P, 4))
    wts /= wts.sum(axis=1, keepdims=True)
    ref = np.array([(wts[k][:, None] * X[idx[k]]).sum(axis=0) for k in range(P)])
    d = ctx.to_device(X)
    HX = ctx.empty((P, M))
    ctx.forward_stencil(rows, 0, M, d, idx, wts, HX)
    assert_parity(HX.download(), ref, "stencil")
    acc = np.zeros((P, M))
    for lo, hi in ((0, 123), (123, 500)):
        ds = ctx.to_device(X[lo:hi])
        ctx.forward_stencil(hi - lo, lo, M, ds, idx, wts, HX)
        acc += HX.download()
    assert_parity(acc, ref, "sharded stencil sum")


def test_synthetic_fill_is_shard_invariant_and_sane():
    ctx = _ctx()
    full = ctx.empty((4000, 30))
    ctx.fill_synthetic(4000, 0, 30, 77, 3.0, full)
    a = full.download()
    part = ctx.empty((1500, 30))
    ctx.fill_synthetic(1500, 2500, 30, 77, 3.0, part)
    assert np.array_equal(part.download(), a[2500:])
    z = (a - a.mean(axis=1, keepdims=True)) / 3.0
    assert abs(z.std() - 1.0) < 0.05 and abs(a.mean()) < 0.1
    assert np.isfinite(a).all()


def test_errors_are_loud():
    from efa_xray_amd import _lib, EnSRF, EnsembleState, Observation
    ctx = _ctx()
    with pytest.raises(_lib.EfaError):
        ctx.set_option("obs_batch", 1000)
    with pytest.raises(_lib.EfaError):
        ctx.set_option("nonsense", 1)
    x = np.zeros(4)
    X = np.zeros((4, 1))
    with pytest.raises(_lib.EfaError):     # M = 1: covariance divides by M-1
        ctx.ensrf_update_host(x, X, 3, [0.0], [1.0], [True])
    st = EnsembleState.from_array(np.zeros((1, 1, 2, 2, 4)), np.zeros((2, 2)), np.zeros((2, 2)))
    ob = Observation(value=1.0, error=1.0, lat=0.0, lon=0.0, assimilate_this=True)
    ob.estimate = lambda s: np.zeros(4)
    with pytest.raises(ValueError):
        EnSRF(st, [ob], verbose=False, loc=True).update()
    with pytest.raises(ValueError):
        EnSRF(st, [ob], verbose=False, loc='GC').update()   # localize_radius is None


@pytest.mark.parametrize("N,M,P", [(1, 4, 1), (130, 8, 70), (257, 128, 129), (1000, 100, 300), (4096, 128, 512), (300, 36, 1000), (515, 256, 130)])
def test_dense_contraction_f32_vs_float64(N, M, P):
    """configs[4] path: fp32 MFMA contraction against a float64 evaluation, rtol 1e-4."""
    ctx = _ctx()
    rng = np.random.default_rng(N + M + P)
    X = rng.standard_normal((N, M)).astype(np.float32)
    Ye = rng.standard_normal((P, M)).astype(np.float32)
    dX, dY, dC = ctx.malloc_bytes(X.nbytes), ctx.malloc_bytes(Ye.nbytes), ctx.malloc_bytes(N * P * 4)
    try:
        ctx.h2d(dX, X)
        ctx.h2d(dY, Ye)
        ctx.cov_contract_f32(N, M, P, dX, dY, dC)
        C = np.empty((N, P), dtype=np.float32)
        ctx.d2h(C, dC)
    finally:
        for p in (dX, dY, dC):
            ctx.free_bytes(p)
    ref = X.astype(np.float64) @ Ye.astype(np.float64).T
    scale = np.abs(X.astype(np.float64)) @ np.abs(Ye.astype(np.float64)).T   # sum |a b|
    assert np.all(np.abs(C - ref) <= 1e-4 * np.abs(ref) + 2e-6 * scale)
    # exact f32 FMA chain (the MFMA's documented numerics) for one row, in the kernel's documented member
    # order: M <= 128 interleaves the two halves (s, KH + s with KH = 8 ceil(M / 16)), larger M runs in k order
    if M <= 128:
        KH = 8 * ((M + 15) // 16)
        order = [k for s_ in range(KH) for k in (s_, KH + s_) if k < M]
    else:
        order = list(range(M))
    assert sorted(order) == list(range(M))
    chain = np.zeros(P, dtype=np.float32)
    for m in order:
        chain = (chain.astype(np.float64) + X[0, m].astype(np.float64) * Ye[:, m].astype(np.float64)).astype(np.float32)
    assert np.allclose(C[0], chain, rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("onepass", [1, 0])
def test_localised_sweep_many_obs_vs_oracle(onepass):
    """GC with several hundred obs on a 48 x 64 grid x 3 slabs: the one-pass active-list sweep and the
    per-batch taper-table sweep against the oracle; state_cycle (members in -> members out) too."""
    c = _random_case(71, 3 * 48 * 64, 40, 420, True, ncol=48 * 64)
    c["hw"][:] = np.random.default_rng(5).uniform(300, 2500, c["P"])
    xam, Xap, diag = _run_oracle(c)
    ctx = _ctx()
    ctx.set_option("gc_onepass", onepass)
    try:
        h_xam, h_Xap, h_diag = _run_hip(c, path="sweep")
        assert_parity(h_xam, xam, "xam")
        assert_parity(h_Xap, Xap, "Xap")
        for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
            assert_parity(h_diag[key], diag[key], key)
        # fused members path
        N, M, P = c["N"], c["M"], c["P"]
        X = ctx.to_device(c["X"])
        Yp = ctx.to_device(c["HX"])
        ym = ctx.empty((P,))
        ctx.form_perts(P, M, Yp, ym, Yp)
        ctx.obs_phase(M, P, ym, Yp, c["val"], c["err"], c["asm"], 1, c["ob_lat"], c["ob_lon"], c["hw"])
        post = ctx.empty((N, M))
        ctx.state_cycle(N, M, X, post, c["lat"].reshape(-1), c["lon"].reshape(-1), c["n_lead"])
        assert_parity(post.download(), orc.format_posterior_state(xam, Xap, N), "post (state_cycle)")
    finally:
        ctx.set_option("gc_onepass", 1)


@pytest.mark.parametrize("name", ["G2", "G3", "G5", "G6", "G8"])
def test_gc_goldens_per_batch_table_path(name):
    """The per-batch taper-table sweep (gc_onepass=0) stays reference-exact on the GC goldens."""
    g = load_golden(name)
    N, xbm, Xbp = prior_arrays(g)
    ctx = _ctx()
    ctx.set_option("gc_onepass", 0)
    try:
        ctx.ensrf_update_host(xbm, Xbp, N, g["ob_value"], g["ob_error"], g["ob_assim"], **golden_kwargs(g))
    finally:
        ctx.set_option("gc_onepass", 1)
    assert_parity(xbm, g["xam"], name + " xam")
    assert_parity(orc.format_posterior_state(xbm, Xbp, N), g["post"], name + " post")


@pytest.mark.parametrize("N,M,P,ncol", [(3 * 48 * 64, 40, 300, 48 * 64), (5 * 700, 80, 150, 700), (2 * 1000, 100, 64, 1000),
                                        (7 * 333, 2, 40, 333), (1 * 257, 128, 33, 257), (6 * 64, 6, 20, 64),
                                        (2 * 300, 130, 40, 300), (3 * 200, 200, 30, 200), (1 * 100, 256, 20, 100),
                                        # the row-per-lane kernel: whole groups of 16 slabs plus a partial one of 5, 1, 10, 3, 2 slabs;
                                        # member counts that are not a multiple of 4 (padded lanes) and the largest it takes (104)
                                        (37 * 100, 98, 40, 100), (17 * 50, 102, 30, 50), (26 * 40, 104, 25, 40), (19 * 33, 10, 30, 33),
                                        (18 * 20, 64, 20, 20), (32 * 17, 50, 20, 17)])
def test_one_pass_gc_sweep_vs_oracle_ragged_shapes(N, M, P, ncol):
    """The one-pass localised sweep against the oracle, in perturbation form and as prior members -> posterior
    members, on shapes whose column count is not a multiple of the 16-column block and whose slab count is not a
    multiple of the slab group (4 RPL slabs in the quad form, 16 in the row-per-lane form); zero-taper rows bit-unchanged."""
    c = _random_case(900 + N + M + P, N, M, P, True, ncol=ncol)
    c["hw"][:] = np.random.default_rng(M).uniform(300, 3000, P)
    xam, Xap, diag = _run_oracle(c)
    ctx = _ctx()
    h_xam, h_Xap, h_diag = _run_hip(c, path="sweep")
    assert_parity(h_xam, xam, "xam")
    assert_parity(h_Xap, Xap, "Xap")
    X = ctx.to_device(c["X"])
    Yp = ctx.to_device(c["HX"])
    ym = ctx.empty((P,))
    ctx.form_perts(P, M, Yp, ym, Yp)
    ctx.obs_phase(M, P, ym, Yp, c["val"], c["err"], c["asm"], 1, c["ob_lat"], c["ob_lon"], c["hw"])
    post = ctx.empty((N, M))
    ctx.state_cycle(N, M, X, post, c["lat"].reshape(-1), c["lon"].reshape(-1), c["n_lead"])
    assert_parity(post.download(), orc.format_posterior_state(xam, Xap, N), "post (state_cycle)")
    w = np.zeros(ncol, dtype=bool)
    for k in range(P):
        if c["asm"][k]:
            w |= orc.localize_state(c["lat"], c["lon"], c["ob_lat"][k], c["ob_lon"][k], c["hw"][k]).reshape(-1) != 0
    untouched = np.tile(~w, c["n_lead"])
    xbm, Xbp = orc.format_prior_state(c["X"], c["HX"])
    assert np.array_equal(h_Xap[:N][untouched], Xbp[:N][untouched])


@pytest.mark.parametrize("M", [3, 5, 7, 21, 99, 101, 127, 129, 130, 135, 136, 137, 160, 201, 255, 256])
def test_transform_path_for_odd_and_large_ensembles(M):
    """The one-pass transform serves every M the library accepts (2..256), odd sizes included (8-byte row alignment: the
    member pairs are loaded separately; above 136 members the [T | w] image is applied in column groups), so no ensemble
    size falls back to one read+write pass per 64 obs; perturbation form through the host ABI and prior members ->
    posterior members through efa_state_cycle_dev, both against the oracle."""
    N, P = 1000 + M, 2 * M + 3
    c = _random_case(4000 + M, N, M, P, False, frac_assim=1.0)
    xam, Xap, diag = _run_oracle(c)
    ctx = _ctx()
    h_xam, h_Xap, h_diag = _run_hip(c, path="transform")
    assert_parity(h_xam, xam, "xam")
    assert_parity(h_Xap, Xap, "Xap")
    for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
        assert_parity(h_diag[key], diag[key], key)
    X = ctx.to_device(c["X"])
    Yp = ctx.to_device(c["HX"])
    ym = ctx.empty((P,))
    ctx.form_perts(P, M, Yp, ym, Yp)
    ctx.obs_phase(M, P, ym, Yp, c["val"], c["err"], c["asm"])
    post = ctx.empty((N, M))
    ctx.state_cycle(N, M, X, post)
    assert ctx.last_timing()["path"] == 2, "auto must choose the transform (more than M/2 obs assimilated)"
    assert_parity(post.download(), orc.format_posterior_state(xam, Xap, N), "post (state_cycle)")


def test_persistent_phase_a_falls_back_within_a_bound_when_the_device_is_occupied():
    """The persistent Phase-A kernels need their whole grid resident at once.  With 200 of the 256 CUs held by another
    kernel (a diagnostic occupier: 120 KB of LDS per CU, so no Phase-A workgroup fits beside it) only part of the grid
    starts; its waves must give up after `spin_ms` of wall time -- not after millions of polls -- the host must go
    straight to the per-batch kernels (not to the other persistent kernel, which has the same need), and the numbers
    must be the per-batch kernels' numbers.  Bound checked: 2 x spin_ms + 1 s for the whole call."""
    import time
    ctx = _ctx()
    M, P = 100, 10000
    rng = np.random.default_rng(77)
    HX = 3.0 * rng.standard_normal((P, M))
    val = HX.mean(axis=1) + rng.standard_normal(P)
    err = rng.uniform(0.5, 2.0, P)
    asm = np.ones(P, dtype=bool)

    def run():
        Yp = ctx.to_device(HX)
        ym = ctx.empty((P,))
        ctx.form_perts(P, M, Yp, ym, Yp)
        t0 = time.perf_counter()
        d = ctx.obs_phase(M, P, ym, Yp, val, err, asm)
        return d, Yp.download(), time.perf_counter() - t0, ctx.get_option("phase_a_kind")

    assert ctx.get_option("cu_count") == 256
    try:
        ctx.set_option("path", 1)
        ctx.set_option("pipeline", 0)
        ref, ref_Y, _, kind = run()
        assert kind == 2
        ctx.set_option("pipeline", 1)
        free, free_Y, t_free, kind = run()
        assert kind == 4                                   # free device: the persistent (band) pipeline runs
        ctx.set_option("spin_ms", 40)
        ctx.set_option("debug_occupy_blocks", 200)
        ctx.set_option("debug_occupy_ms", 1500)            # returns at once: the occupier runs on its own stream
        time.sleep(0.05)
        busy, busy_Y, t_busy, kind = run()
        assert kind == 2, "the persistent launch cannot have completed on 56 CUs"
        assert t_busy < 2 * 0.040 + 1.0, "fallback took %.3f s" % t_busy
        for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
            assert np.array_equal(busy[key], ref[key], equal_nan=True), key
        assert np.array_equal(busy_Y, ref_Y)
        assert_parity(free_Y, ref_Y, "pipeline vs per-batch obs rows")
    finally:
        ctx.set_option("debug_occupy_ms", 0)               # wait for the occupier
        ctx.set_option("spin_ms", -1)
        ctx.set_option("pipeline", 1)
        ctx.set_option("path", 0)


@pytest.mark.gpu
def test_gc_sweep_random_shapes_vs_oracle():
    """Sixteen random (members, slabs, columns, obs) shapes through both forms of the one-pass localised sweep -- prior members
    in / posterior members out (efa_state_cycle_dev) and perturbations + means (the array API) -- against the oracle:
    member counts that are and are not a multiple of 4 (the row-per-lane kernel pads), slab counts around the groups of 16,
    column counts that are not a multiple of the 16-column block."""
    rng = np.random.default_rng(2026)
    for it in range(16):
        M = int(rng.choice([2, 4, 6, 10, 18, 34, 50, 66, 80, 98, 100, 102, 104]))
        n_lead, ncol, P = int(rng.integers(1, 41)), int(rng.integers(17, 140)), int(rng.integers(5, 70))
        N = n_lead * ncol
        c = _random_case(5000 + it, N, M, P, True, ncol=ncol)
        c["hw"][:] = rng.uniform(300, 4000, P)
        xam, Xap, diag = _run_oracle(c)
        ctx = _ctx()
        X = ctx.to_device(c["X"])
        Yp = ctx.to_device(c["HX"])
        ym = ctx.empty((P,))
        ctx.form_perts(P, M, Yp, ym, Yp)
        ctx.obs_phase(M, P, ym, Yp, c["val"], c["err"], c["asm"], 1, c["ob_lat"], c["ob_lon"], c["hw"])
        post = ctx.empty((N, M))
        ctx.state_cycle(N, M, X, post, c["lat"].reshape(-1), c["lon"].reshape(-1), c["n_lead"])
        assert_parity(post.download(), orc.format_posterior_state(xam, Xap, N), "post, case %d (M=%d, slabs=%d, columns=%d)" % (it, M, n_lead, ncol))
        h_xam, h_Xap, _ = _run_hip(c, path="sweep")
        assert_parity(h_Xap, Xap, "Xap, case %d" % it)
        assert_parity(h_xam, xam, "xam, case %d" % it)


@pytest.mark.gpu
def test_phase_a_band_leader_random_shapes_vs_oracle():
    """Twelve random (members, obs) shapes, a quarter of the obs not assimilated, through the persistent band leader and the
    transform: block counts with and without a partial last block and a partial last band, member counts on both sides of
    the deferred-Gram limit (104)."""
    rng = np.random.default_rng(77)
    ctx = _ctx()
    for it in range(12):
        M = int(rng.choice([2, 3, 8, 20, 50, 64, 99, 100, 104, 105, 120, 128]))
        P = int(rng.integers(1, 700))
        N = int(rng.integers(1, 300))
        c = _random_case(7000 + it, N, M, P, False, frac_assim=0.75)
        xam, Xap, diag = _run_oracle(c)
        for path in ("auto", "sweep"):
            h_xam, h_Xap, h_diag = _run_hip(c, path=path, pipeline=3)
            # (with a handful of members the variance collapses within a block and the leader's cancellation guard hands the
            #  call to the vector-chain kernel: kind 1 -- legitimately; from 20 members on the band leader must have done it)
            assert ctx.get_option("phase_a_kind") in ((4,) if M >= 20 else (4, 1)), (it, M, P)
            assert_parity(h_xam, xam, "xam, case %d (M=%d, P=%d) %s" % (it, M, P, path))
            assert_parity(h_Xap, Xap, "Xap, case %d" % it)
            for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
                assert_parity(h_diag[key], diag[key], key)
            assert np.array_equal(h_diag["assimilated"], diag["assimilated"])


@pytest.mark.gpu
def test_deferred_timing_sums_over_back_to_back_cycles():
    """`timing` 2 (what bench.py runs its timed steps with): no phase call waits for its own events, efa_last_timing returns the
    sums over the calls since its previous call and clears them; `timing` 1 keeps the per-call meaning.  Results are the same
    either way."""
    ctx = _ctx()
    rng = np.random.default_rng(5)
    M, P, N = 40, 300, 5000
    HX = 3.0 * rng.standard_normal((P, M))
    val = HX.mean(axis=1) + rng.standard_normal(P)
    err, asm = np.ones(P), np.ones(P, dtype=bool)
    X = ctx.to_device(rng.standard_normal((N, M)))
    posts = {}
    try:
        for mode in (1, 2):
            ctx.set_option("timing", mode)
            post = ctx.empty((N, M))
            per_call = []
            for _ in range(3):
                Yp = ctx.to_device(HX)
                ym = ctx.empty((P,))
                ctx.form_perts(P, M, Yp, ym, Yp)
                ctx.obs_phase(M, P, ym, Yp, val, err, asm)
                ctx.state_cycle(N, M, X, post)
                if mode == 1:
                    per_call.append(ctx.last_timing())
            ctx.synchronize()
            t = ctx.last_timing()
            posts[mode] = post.download()
            if mode == 1:
                assert all(c["state_ms"] > 0 and c["obs_ms"] > 0 and c["state_launches"] == 1 for c in per_call)
                assert t["state_ms"] == per_call[-1]["state_ms"]            # still the last call's
            else:
                assert t["state_launches"] == 3 and t["state_ms"] > 0 and t["obs_ms"] > 0
                again = ctx.last_timing()
                assert again["state_ms"] == 0.0 and again["obs_ms"] == 0.0 and again["state_launches"] == 0
        assert np.array_equal(posts[1], posts[2])
        with pytest.raises(Exception):
            ctx.set_option("timing", 3)
    finally:
        ctx.set_option("timing", 0)


def _cycle_two_ways(c, fused_kwargs=None, path="auto"):
    """The same resident cycle through efa_obs_phase_dev + efa_state_cycle_dev and through efa_ensrf_cycle_dev."""
    ctx = _ctx()
    ctx.set_option("path", {"auto": 0, "sweep": 1, "transform": 2}[path])
    N, M, P = c["N"], c["M"], c["P"]
    loc = 1 if c["loc"] else 0
    okw = dict(loc_mode=loc)
    gl = gn = None
    n_lead = 1
    if loc:
        okw.update(ob_lat=c["ob_lat"], ob_lon=c["ob_lon"], ob_halfwidth=c["hw"])
        gl, gn, n_lead = c["lat"].reshape(-1), c["lon"].reshape(-1), c["n_lead"]
    out = {}
    try:
        for how in ("split", "fused"):
            X = ctx.to_device(c["X"])
            post = ctx.empty((N, M))
            Yp = ctx.to_device(c["HX"])
            ym = ctx.empty((P,))
            ctx.form_perts(P, M, Yp, ym, Yp)
            if how == "split":
                d = ctx.obs_phase(M, P, ym, Yp, c["val"], c["err"], c["asm"], **okw)
                ctx.state_cycle(N, M, X, post, gl, gn, n_lead)
            else:
                d = ctx.ensrf_cycle(N, M, P, X, post, ym, Yp, c["val"], c["err"], c["asm"], grid_lat=gl, grid_lon=gn, n_lead=n_lead,
                                    **dict(okw, **(fused_kwargs or {})))
            out[how] = (post.download(), Yp.download(), ym.download(), d, ctx.get_option("phase_a_kind"), ctx.last_timing()["path"])
    finally:
        ctx.set_option("path", 0)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("loc", [False, True])
def test_fused_cycle_equals_obs_phase_plus_state_cycle(loc):
    """efa_ensrf_cycle_dev == efa_obs_phase_dev + efa_state_cycle_dev bit for bit -- unlocalised (Phase B enqueued behind
    Phase A before its status is known), localised (no speculation) -- and against the oracle; obs_block_out."""
    c = _random_case(4242, 2048 if loc else 3000, 50, 500, loc, frac_assim=0.9, ncol=(512 if loc else None))
    xam, Xap, diag = _run_oracle(c)
    ref_post = orc.format_posterior_state(xam, Xap, c["N"])
    ym0, Yp0 = orc.compute_ob_priors(c["HX"])
    for obs_out in (False, True):
        r = _cycle_two_ways(c, dict(obs_block_out=obs_out))
        s, f = r["split"], r["fused"]
        assert np.array_equal(f[0], s[0]), "posterior members"
        assert_parity(f[0], ref_post, "post vs oracle")
        for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
            assert np.array_equal(f[3][key], s[3][key], equal_nan=True), key
            assert_parity(f[3][key], diag[key], key)
        assert np.array_equal(f[3]["assimilated"], s[3]["assimilated"])
        assert f[4] == s[4] == 4 and f[5] == s[5] == (1 if loc else 2)
        if obs_out:      # the final obs block comes back as from efa_obs_phase_dev
            assert np.array_equal(f[1], s[1]) and np.array_equal(f[2], s[2])
        else:            # the caller's obs priors are left alone
            assert_parity(f[1], Yp0, "obs perturbations untouched")
            assert_parity(f[2], ym0, "obs means untouched")


@pytest.mark.gpu
def test_fused_cycle_when_the_speculated_launch_falls_back():
    """The Gram-space cancellation guard trips in the launch behind which Phase B was enqueued: Phase A is redone by the
    vector-chain kernel and Phase B enqueued again -- the wasted transform must leave no trace."""
    c = _random_case(77, 3000, 24, 90, False, frac_assim=1.0)
    c["HX"][1:40] = c["HX"][0] + 1e-4 * np.random.default_rng(3).standard_normal((39, 24))
    c["val"][:40] = c["HX"][0].mean() + 0.1
    c["err"][:40] = 1e-8
    xam, Xap, diag = _run_oracle(c)
    r = _cycle_two_ways(c, path="transform")
    s, f = r["split"], r["fused"]
    assert f[4] == s[4] == 1, "the guard must have handed Phase A to the vector-chain kernel"
    assert np.array_equal(f[0], s[0])
    assert_parity(f[0], orc.format_posterior_state(xam, Xap, c["N"]), "post vs oracle")
    assert_parity(f[3]["post_var"], diag["post_var"], "post_var")


@pytest.mark.gpu
def test_fused_cycle_in_place_and_edge_cases():
    """post_dev == X_dev (no speculation: a wrong guess would cost the prior), no obs, none assimilated, deferred timing."""
    ctx = _ctx()
    c = _random_case(99, 1000, 40, 200, False, frac_assim=0.9)
    xam, Xap, _ = _run_oracle(c)
    ref_post = orc.format_posterior_state(xam, Xap, c["N"])
    N, M, P = c["N"], c["M"], c["P"]
    X = ctx.to_device(c["X"])
    Yp = ctx.to_device(c["HX"])
    ym = ctx.empty((P,))
    ctx.form_perts(P, M, Yp, ym, Yp)
    ctx.ensrf_cycle(N, M, P, X, X, ym, Yp, c["val"], c["err"], c["asm"])
    assert_parity(X.download(), ref_post, "in-place cycle")
    # none assimilated / no obs: the posterior is the prior
    for Pz, asm in ((P, np.zeros(P, dtype=bool)), (0, np.zeros(0, dtype=bool))):
        X = ctx.to_device(c["X"])
        post = ctx.empty((N, M))
        Yp = ctx.to_device(c["HX"][:max(Pz, 1)])
        ym = ctx.empty((max(Pz, 1),))
        ctx.form_perts(max(Pz, 1), M, Yp, ym, Yp)
        d = ctx.ensrf_cycle(N, M, Pz, X, post, ym, Yp, c["val"][:Pz], c["err"][:Pz], asm)
        assert_parity(post.download(), c["X"], "no update")
        assert not d["assimilated"].any()
    # deferred timing across fused cycles: sums, one transform launch per cycle, and the host is not held by the events
    try:
        ctx.set_option("timing", 2)
        X = ctx.to_device(c["X"])
        post = ctx.empty((N, M))
        ym = ctx.empty((P,))
        for _ in range(4):
            Yp = ctx.to_device(c["HX"])
            ctx.form_perts(P, M, Yp, ym, Yp)
            ctx.ensrf_cycle(N, M, P, X, post, ym, Yp, c["val"], c["err"], c["asm"])
        ctx.synchronize()
        t = ctx.last_timing()
        assert t["state_launches"] == 4 and t["state_ms"] > 0 and t["obs_ms"] > 0 and t["path"] == 2
        assert_parity(post.download(), ref_post, "post after four cycles")
    finally:
        ctx.set_option("timing", 0)


@pytest.mark.gpu
def test_geometry_reuse_across_localised_cycles():
    """The obs-obs taper table and the sweep's active lists depend on the geometry only and are kept from cycle to cycle
    (efa_ensrf_cycle_dev, option geometry_reuse): a second cycle with new obs values / error variances / state on the same
    geometry must use them and still match the oracle; a moved ob, a changed assimilate flag, a changed radius and a changed grid
    must each be noticed."""
    ctx = _ctx()
    base = _random_case(808, 4 * 600, 40, 150, True, frac_assim=0.9, ncol=600)
    N, M, P = base["N"], base["M"], base["P"]
    rng = np.random.default_rng(809)

    def run(c, reuse=1):
        ctx.set_option("geometry_reuse", reuse)
        X = ctx.to_device(c["X"])
        post = ctx.empty((N, M))
        Yp = ctx.to_device(c["HX"])
        ym = ctx.empty((P,))
        ctx.form_perts(P, M, Yp, ym, Yp)
        d = ctx.ensrf_cycle(N, M, P, X, post, ym, Yp, c["val"], c["err"], c["asm"], 1, c["ob_lat"], c["ob_lon"], c["hw"],
                            c["lat"].reshape(-1), c["lon"].reshape(-1), c["n_lead"])
        xam, Xap, diag = _run_oracle(c)
        assert_parity(post.download(), orc.format_posterior_state(xam, Xap, N), "post")
        assert_parity(d["post_var"], diag["post_var"], "post_var")
        return post.download(), int(ctx.get_option("gc_active_pairs"))

    def variant(**changes):
        c = dict(base)
        for k in ("X", "HX", "val", "err", "asm", "ob_lat", "ob_lon", "hw", "lat", "lon"):
            c[k] = np.array(base[k], copy=True)
        c.update(changes)
        return c

    try:
        p0, n0 = run(base)
        # same geometry, everything else new
        c1 = variant(X=base["X"] + rng.standard_normal(base["X"].shape), err=rng.uniform(0.3, 3.0, P))
        c1["HX"] = c1["X"][rng.choice(N, P, replace=False)]
        c1["val"] = c1["HX"].mean(axis=1) + rng.standard_normal(P)
        p1, n1 = run(c1)
        assert n1 == n0
        p1b, _ = run(c1, reuse=0)              # rebuilt from scratch: the same bits
        assert np.array_equal(p1, p1b)
        # one ob moved / one flag flipped / one radius changed / the grid shifted: each must be rebuilt for
        lat2 = base["ob_lat"].copy()
        lat2[7] += 3.0
        run(variant(ob_lat=lat2))
        asm2 = base["asm"].copy()
        asm2[11] = not asm2[11]
        run(variant(asm=asm2))
        hw2 = base["hw"].copy()
        hw2[np.flatnonzero(base["asm"])[3]] *= 0.37
        _, n4 = run(variant(hw=hw2))
        assert n4 != n0
        run(variant(lat=base["lat"] + 0.5))
        run(base)
    finally:
        ctx.set_option("geometry_reuse", 1)
