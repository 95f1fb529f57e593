"""Pin the CPU oracle (oracle/ensrf_oracle.py) to the reference's own outputs.

The fixtures were produced by running the reference's modules in the build
container (tests/golden/make_goldens.py).  On the NumPy/OpenBLAS install that
made them the restatement is bit-identical; elsewhere BLAS summation order may
differ, so the portable assertion is rtol 1e-12 and bit-equality is reported.
"""
import numpy as np
import pytest

from conftest import load_golden
from oracle import ensrf_oracle as orc

RTOL = 1e-12


def _close(a, b, what):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, what
    scale = max(np.nanmax(np.abs(b)), 1e-300) if b.size else 1.0
    np.testing.assert_allclose(a, b, rtol=RTOL, atol=1e-14 * scale, equal_nan=True, err_msg=what)


def test_known_answers_scalar_functions():
    k = load_golden("KAT")
    assert np.array_equal(orc.gaspari_cohn(k["gc_d"], float(k["gc_c"])), k["gc_w"])
    assert np.array_equal(orc.gaspari_cohn(k["gc_d"], -float(k["gc_c"])), k["gc_w_neg"])
    # survey probe values
    w = orc.gaspari_cohn(np.array([0, 400, 800, 1200, 1599, 1600, 2000.0]), 800.0)
    np.testing.assert_allclose(w, [1, 0.684895833333, 0.208333333333, 0.0164930555556, 7.632e-13, 0, 0],
                               rtol=2e-4, atol=1e-16)
    hv = np.array([orc.haversine(tuple(a), tuple(b)) for a, b in zip(k["hv_a"], k["hv_b"])])
    assert np.array_equal(hv, k["hv_km"])
    assert abs(hv[0] - 325.1000863921916) < 1e-9
    assert hv[1] == 0.0
    d = orc.distance_to_point(k["dp_lat"], k["dp_lon"], *k["dp_pt"])
    assert np.array_equal(d, k["dp_km"])


def test_oracle_matches_reference(golden):
    g = golden
    nvar, nt, ny, nx, M = [int(v) for v in g["shape"]]
    N = nvar * nt * ny * nx
    X = g["X"].reshape(N, M)
    # forward operator output as the reference's compute_ob_priors saw it
    HX = np.array([(g["sten_wts"][k][:, None] * X[g["sten_idx"][k]]).sum(axis=0)
                   if g["sten_wts"][k][0] != 1.0 else X[g["sten_idx"][k][0]]
                   for k in range(len(g["ob_value"]))])
    assert np.array_equal(HX, g["HX"])
    xbm, Xbp = orc.format_prior_state(X, HX)
    if "xbm" in g:
        assert np.array_equal(xbm, g["xbm"]) and np.array_equal(Xbp, g["Xbp"])
    kw = {}
    if g["loc"] == "GC":
        kw = dict(loc="GC", ob_lat=g["ob_lat"], ob_lon=g["ob_lon"], ob_halfwidth=g["ob_radius"],
                  grid_lat=g["grid_lat"], grid_lon=g["grid_lon"], state_shape=(nvar, nt, ny, nx))
    xam, Xap, diag = orc.ensrf_update(xbm, Xbp, N, g["ob_value"], g["ob_error"], g["ob_assim"], **kw)
    post = orc.format_posterior_state(xam, Xap, N)
    _close(xam, g["xam"], "xam")
    if "Xap" in g:
        _close(Xap, g["Xap"], "Xap")
    _close(post, g["post"], "post")
    for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
        _close(diag[key], g[key], key)
    assert np.array_equal(diag["assimilated"], g["assimilated"])
    bit = np.array_equal(post, g["post"]) and np.array_equal(xam, g["xam"])
    print("%s bit-identical to reference: %s" % (g["name"], bit))


@pytest.mark.parametrize("name", ["G1", "G2", "G5"])
def test_faithful_cost_mode_same_bits(name):
    g = load_golden(name)
    nvar, nt, ny, nx, M = [int(v) for v in g["shape"]]
    N = nvar * nt * ny * nx
    kw = {}
    if g["loc"] == "GC":
        kw = dict(loc="GC", ob_lat=g["ob_lat"], ob_lon=g["ob_lon"], ob_halfwidth=g["ob_radius"],
                  grid_lat=g["grid_lat"], grid_lon=g["grid_lon"], state_shape=(nvar, nt, ny, nx))
    a = orc.ensrf_update(g["xbm"], g["Xbp"], N, g["ob_value"], g["ob_error"], g["ob_assim"], **kw)
    b = orc.ensrf_update(g["xbm"], g["Xbp"], N, g["ob_value"], g["ob_error"], g["ob_assim"],
                         faithful_cost=True, **kw)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for key in a[2]:
        assert np.array_equal(a[2][key], b[2][key], equal_nan=True)


# ---------------------------------------------------------------------------
# f1: forward operator, pinned by the reference's own nearest_points / interpolate /
# Observation.estimate run verbatim (fixtures G9 2-D lat/lon, G10 1-D lat/lon, G11 end to end)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["G9", "G10"])
def test_oracle_forward_operator_matches_reference(name):
    g = load_golden(name)
    nvar, nt, ny, nx, M = [int(v) for v in g["shape"]]
    eye = np.eye(nt * ny * nx).reshape(nt, ny, nx, -1)
    bit = True
    for k in range(len(g["ob_lat"])):
        args = (g["grid_lat"], g["grid_lon"], g["validtime"], g["ob_time"][k], g["ob_lat"][k], g["ob_lon"][k])
        near = orc.nearest_points(g["grid_lat"], g["grid_lon"], g["ob_lat"][k], g["ob_lon"][k], npt=4)
        assert np.array_equal(np.stack(near, axis=-1), g["nearest"][k]), "ob %d: nearest four" % k
        hx = orc.interpolate(g["X"][g["ob_var"][k]], *args)
        w = orc.interpolate(eye, *args)
        _close(hx, g["HX"][k], "ob %d: estimate" % k)
        _close(w, g["weights"][k], "ob %d: stencil weights" % k)
        bit = bit and np.array_equal(hx, g["HX"][k]) and np.array_equal(w, g["weights"][k])
    # outside the valid times the reference returns None (ensemble.py:207-209)
    assert orc.interpolate(g["X"][0], g["grid_lat"], g["grid_lon"], g["validtime"],
                           g["validtime"][-1] + np.timedelta64(1, "s"), g["ob_lat"][0], g["ob_lon"][0]) is None
    print("%s forward operator bit-identical to reference: %s" % (name, bit))


def test_oracle_cycle_with_the_reference_forward_operator_end_to_end():
    """G11: EnSRF.update() with plain Observations (ensrf.py:33-151 incl. assimilation.py:36-49 ->
    observation.py:40-50 -> ensemble.py:170-239)."""
    g = load_golden("G11")
    nvar, nt, ny, nx, M = [int(v) for v in g["shape"]]
    N = nvar * nt * ny * nx
    HX = np.array([orc.interpolate(g["X"][g["ob_var"][k]], g["grid_lat"], g["grid_lon"], g["validtime"],
                                   g["ob_time"][k], g["ob_lat"][k], g["ob_lon"][k]) for k in range(len(g["ob_lat"]))])
    _close(HX, g["HX"], "HX")
    post, xam, _, diag = orc.ensrf_cycle(g["X"].reshape(N, M), HX, g["ob_value"], g["ob_error"], g["ob_assim"],
                                         loc="GC", ob_lat=g["ob_lat"], ob_lon=g["ob_lon"], ob_halfwidth=g["ob_radius"],
                                         grid_lat=g["grid_lat"], grid_lon=g["grid_lon"], state_shape=(nvar, nt, ny, nx))
    _close(xam, g["xam"], "xam")
    _close(post, g["post"], "post")
    for key in ("prior_mean", "prior_var", "post_mean", "post_var"):
        _close(diag[key], g[key], key)
    assert np.array_equal(diag["assimilated"], g["assimilated"])
