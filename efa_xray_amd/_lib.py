"""ctypes binding of libefa_hip.so (C ABI declared in include/efa_hip.h).

There is no CPU fallback anywhere in this package: if the shared library is
missing, or no gfx950 device is usable, the calls below raise.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libefa_hip.so")

EFA_OK = 0
EFA_ERR_INVALID = -1
EFA_ERR_NO_DEVICE = -2
EFA_ERR_HIP = -3
EFA_ERR_UNSUPPORTED = -4

LOC_NONE = 0
LOC_GC = 1

PATH_AUTO = 0
PATH_SWEEP = 1
PATH_TRANSFORM = 2

c_double_p = ctypes.POINTER(ctypes.c_double)
c_uint8_p = ctypes.POINTER(ctypes.c_uint8)
c_int64_p = ctypes.POINTER(ctypes.c_int64)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)

# every exported symbol with its (restype, argtypes); tests check this table
# against include/efa_hip.h and against the built library.
SIGNATURES = {
    "efa_abi_version": (ctypes.c_int, []),
    "efa_last_error": (ctypes.c_char_p, []),
    "efa_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "efa_ctx_create": (ctypes.c_int, [ctypes.c_int, c_void_pp]),
    "efa_ctx_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "efa_ctx_set_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "efa_ctx_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_long]),
    "efa_ctx_get_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_long)]),
    "efa_ctx_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "efa_malloc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, c_void_pp]),
    "efa_free": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "efa_memcpy_h2d": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "efa_memcpy_d2h": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "efa_memcpy_d2d": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "efa_form_perts_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    "efa_posterior_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p]),
    "efa_forward_stencil_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                               ctypes.c_void_p, ctypes.c_long, ctypes.c_int, c_int64_p,
                                               c_double_p, ctypes.c_void_p]),
    "efa_interp_stencils": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_long, c_double_p, c_double_p, c_double_p, ctypes.c_long,
                                           ctypes.POINTER(ctypes.c_int32), c_double_p, c_double_p, c_double_p,
                                           c_int64_p, c_double_p, c_uint8_p]),
    "efa_forward_interp_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long,
                                              ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "efa_ensrf_update_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_long,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            c_double_p, c_double_p, c_uint8_p, ctypes.c_int,
                                            c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                            ctypes.c_long, ctypes.c_long,
                                            c_double_p, c_double_p, c_double_p, c_double_p, c_uint8_p]),
    "efa_obs_phase_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_void_p,
                                         ctypes.c_void_p, c_double_p, c_double_p, c_uint8_p, ctypes.c_int,
                                         c_double_p, c_double_p, c_double_p,
                                         c_double_p, c_double_p, c_double_p, c_double_p, c_uint8_p]),
    "efa_state_phase_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           c_double_p, c_double_p, ctypes.c_long, ctypes.c_long]),
    "efa_state_cycle_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p, c_double_p, c_double_p, ctypes.c_long,
                                           ctypes.c_long]),
    "efa_ensrf_cycle_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_long,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                           c_double_p, c_double_p, c_uint8_p, ctypes.c_int,
                                           c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                           ctypes.c_long, ctypes.c_long,
                                           c_double_p, c_double_p, c_double_p, c_double_p, c_uint8_p]),
    "efa_ensrf_update": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                        ctypes.c_long, c_double_p, c_double_p,
                                        c_double_p, c_double_p, c_uint8_p, ctypes.c_int,
                                        c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                        ctypes.c_long, ctypes.c_long,
                                        c_double_p, c_double_p, c_double_p, c_double_p, c_uint8_p]),
    "efa_cov_contract_f32_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_long,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "efa_last_timing": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p,
                                       ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_int)]),
    "efa_fill_synthetic_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                              ctypes.c_uint64, ctypes.c_double, ctypes.c_void_p]),
    "efa_comm_unique_id": (ctypes.c_int, [c_uint8_p]),
    "efa_comm_init": (ctypes.c_int, [ctypes.c_void_p, c_uint8_p, ctypes.c_int, ctypes.c_int]),
    "efa_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "efa_allreduce_sum_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]),
    "efa_gc_block_counts": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, c_double_p, c_double_p, ctypes.c_long,
                                           c_double_p, c_double_p, c_double_p, c_uint8_p,
                                           ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                           ctypes.POINTER(ctypes.c_uint64)]),
}

COMM_ID_BYTES = 128


class EfaError(RuntimeError):
    """A libefa_hip call failed (status < 0); message from efa_last_error()."""

    def __init__(self, status, message):
        RuntimeError.__init__(self, "libefa_hip error %d: %s" % (status, message))
        self.status = status


_lib = None


def load_library(path=None):
    """dlopen libefa_hip.so (once) and declare every prototype.

    Raises RuntimeError if the library has not been built -- build it with
    `python -c "import __graft_entry__ as g; g.build()"` or
    `make -C efa_xray_amd/csrc`.  Nothing falls back to a CPU path.
    """
    global _lib
    if _lib is not None and path is None:
        return _lib
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64; if libefa_hip.so pulls in
    # /opt/rocm's copy first and torch is imported later, torch's copy finds no GPU ("No HIP GPUs are
    # available").  Where torch is installed (HipEngine / bench.py use it for device memory and RCCL) it is
    # therefore imported first, so that both resolve to the copy already loaded.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    # EFA_HIP_LIB: a diagnostic build of the same library (tools/ only; `make -C efa_xray_amd/csrc diag`)
    p = path or os.environ.get("EFA_HIP_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            "libefa_hip.so not found at %s: the HIP extension is not built "
            "(run `make -C %s`); efa_xray_amd has no CPU fallback" % (p, os.path.join(_HERE, "csrc")))
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.efa_abi_version() != 1:
        raise RuntimeError("libefa_hip ABI version %d, expected 1" % lib.efa_abi_version())
    if path is None:
        _lib = lib
    return lib


def _check(lib, status):
    if status != EFA_OK:
        msg = lib.efa_last_error()
        raise EfaError(status, msg.decode("utf-8", "replace") if msg else "")


def device_count():
    lib = load_library()
    n = ctypes.c_int(0)
    _check(lib, lib.efa_device_count(ctypes.byref(n)))
    return n.value


def _dp(a):
    """double* view of a C-contiguous float64 ndarray (or NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


def _u8p(a):
    if a is None:
        return None
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_uint8_p)


class DeviceArray(object):
    """A float64 array in the context GPU's HBM (hipMalloc via efa_malloc)."""

    def __init__(self, ctx, shape):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * 8
        ptr = ctypes.c_void_p()
        _check(ctx.lib, ctx.lib.efa_malloc(ctx.handle, self.nbytes, ctypes.byref(ptr)))
        self.ptr = ptr

    @property
    def address(self):
        return self.ptr.value

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=np.float64)
        assert host.nbytes == self.nbytes, (host.shape, self.shape)
        _check(self.ctx.lib, self.ctx.lib.efa_memcpy_h2d(self.ctx.handle, self.ptr, host.ctypes.data, self.nbytes))
        return self

    def download(self, out=None):
        if out is None:
            out = np.empty(self.shape, dtype=np.float64)
        assert out.nbytes == self.nbytes and out.flags["C_CONTIGUOUS"]
        _check(self.ctx.lib, self.ctx.lib.efa_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def upload_rows(self, r0, host):
        """Rows [r0, r0 + len(host)) from a C-contiguous float64 host array, straight from the caller's memory."""
        host = np.ascontiguousarray(host, dtype=np.float64)
        width = int(np.prod(self.shape[1:], dtype=np.int64)) if len(self.shape) > 1 else 1
        assert host.size % width == 0 and r0 * width * 8 + host.nbytes <= self.nbytes
        dst = ctypes.c_void_p(self.ptr.value + r0 * width * 8)
        _check(self.ctx.lib, self.ctx.lib.efa_memcpy_h2d(self.ctx.handle, dst, host.ctypes.data, host.nbytes))

    def download_rows_into(self, r0, out):
        """Rows [r0, r0 + out.size / width) into `out` (C-contiguous float64), without an intermediate array."""
        assert out.dtype == np.float64 and out.flags["C_CONTIGUOUS"]
        width = int(np.prod(self.shape[1:], dtype=np.int64)) if len(self.shape) > 1 else 1
        assert out.size % width == 0 and r0 * width * 8 + out.nbytes <= self.nbytes
        src = ctypes.c_void_p(self.ptr.value + r0 * width * 8)
        _check(self.ctx.lib, self.ctx.lib.efa_memcpy_d2h(self.ctx.handle, out.ctypes.data, src, out.nbytes))
        return out

    def download_rows(self, r0, r1):
        """Rows [r0, r1) of a 2-D (or 1-D) array, without copying the rest."""
        width = int(np.prod(self.shape[1:], dtype=np.int64)) if len(self.shape) > 1 else 1
        out = np.empty((r1 - r0,) + self.shape[1:], dtype=np.float64)
        src = ctypes.c_void_p(self.ptr.value + r0 * width * 8)
        _check(self.ctx.lib, self.ctx.lib.efa_memcpy_d2h(self.ctx.handle, out.ctypes.data, src, out.nbytes))
        return out

    def free(self):
        if self.ptr is not None and self.ptr.value and self.ctx.handle is not None:
            self.ctx.lib.efa_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context(object):
    """One efa_ctx: a GPU, its stream and its workspaces."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.handle = None
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.efa_ctx_create(int(device), ctypes.byref(h)))
        self.handle = h
        self.device = int(device)

    def close(self):
        if self.handle is not None:
            self.lib.efa_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- options ------------------------------------------------------------
    def set_option(self, key, value):
        _check(self.lib, self.lib.efa_ctx_set_option(self.handle, key.encode(), int(value)))

    def get_option(self, key):
        v = ctypes.c_long(0)
        _check(self.lib, self.lib.efa_ctx_get_option(self.handle, key.encode(), ctypes.byref(v)))
        return v.value

    def set_stream(self, hip_stream):
        _check(self.lib, self.lib.efa_ctx_set_stream(self.handle, ctypes.c_void_p(hip_stream or 0)))

    def synchronize(self):
        _check(self.lib, self.lib.efa_ctx_synchronize(self.handle))

    # -- memory ---------------------------------------------------------------
    def empty(self, shape):
        return DeviceArray(self, shape)

    def to_device(self, host):
        host = np.ascontiguousarray(host, dtype=np.float64)
        return DeviceArray(self, host.shape).upload(host)

    # -- kernels --------------------------------------------------------------
    @staticmethod
    def _addr(x):
        if x is None:
            return None
        if isinstance(x, DeviceArray):
            return x.ptr
        if isinstance(x, ctypes.c_void_p):
            return x
        return ctypes.c_void_p(int(x))      # raw device address (e.g. torch data_ptr())

    def form_perts(self, rows, M, X, xm, Xp, scale=1.0):
        _check(self.lib, self.lib.efa_form_perts_dev(self.handle, rows, M, self._addr(X), float(scale),
                                                     self._addr(xm), self._addr(Xp)))

    def posterior(self, rows, M, xm, Xp, post):
        _check(self.lib, self.lib.efa_posterior_dev(self.handle, rows, M, self._addr(xm), self._addr(Xp),
                                                    self._addr(post)))

    def forward_stencil(self, rows, row_offset, M, X, idx, wts, HX):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        wts = np.ascontiguousarray(wts, dtype=np.float64)
        assert idx.ndim == 2 and idx.shape == wts.shape
        P, npt = idx.shape
        _check(self.lib, self.lib.efa_forward_stencil_dev(
            self.handle, rows, row_offset, M, self._addr(X), P, npt,
            idx.ctypes.data_as(c_int64_p), _dp(wts), self._addr(HX)))

    def interp_stencils(self, nvar, nt, ny, nx, grid_lat, grid_lon, valid_times, ob_var, ob_time, ob_lat, ob_lon,
                        want_host=True):
        """f1: build the interpolation stencils of P point obs on the device (they stay in the context for
        `forward_interp`).  Returns (idx (P,8) int64 global rows, wts (P,8), status (P,) uint8) or None."""
        glat = np.ascontiguousarray(grid_lat, dtype=np.float64)
        glon = np.ascontiguousarray(grid_lon, dtype=np.float64)
        latlon_1d = 1 if glat.ndim == 1 else 0
        glat, glon = glat.reshape(-1), glon.reshape(-1)
        vt = np.ascontiguousarray(valid_times, dtype=np.float64).reshape(-1)
        P = len(ob_var)
        var = np.ascontiguousarray(ob_var, dtype=np.int32)
        tim = np.ascontiguousarray(ob_time, dtype=np.float64).reshape(P)
        lat = np.ascontiguousarray(ob_lat, dtype=np.float64).reshape(P)
        lon = np.ascontiguousarray(ob_lon, dtype=np.float64).reshape(P)
        idx = np.full((P, 8), -1, dtype=np.int64)
        wts = np.zeros((P, 8))
        st = np.zeros(P, dtype=np.uint8)
        _check(self.lib, self.lib.efa_interp_stencils(
            self.handle, int(nvar), int(nt), int(ny), int(nx), latlon_1d, glat.shape[0], _dp(glat), _dp(glon), _dp(vt), P,
            var.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _dp(tim), _dp(lat), _dp(lon),
            idx.ctypes.data_as(c_int64_p) if want_host else None, _dp(wts) if want_host else None, _u8p(st)))
        return idx, wts, st

    def forward_interp(self, ncol, col_lo, col_hi, n_lead, M, X, HX):
        _check(self.lib, self.lib.efa_forward_interp_dev(self.handle, ncol, col_lo, col_hi, n_lead, M, self._addr(X),
                                                         self._addr(HX)))

    @staticmethod
    def _ob_arrays(P, ob_value, ob_error, ob_assim, loc_mode, ob_lat, ob_lon, ob_halfwidth):
        val = np.ascontiguousarray(ob_value, dtype=np.float64).reshape(P)
        err = np.ascontiguousarray(ob_error, dtype=np.float64).reshape(P)
        asm = np.ascontiguousarray(np.asarray(ob_assim).astype(bool), dtype=np.uint8).reshape(P)
        lat = lon = hw = None
        if loc_mode == LOC_GC:
            lat = np.ascontiguousarray(ob_lat, dtype=np.float64).reshape(P)
            lon = np.ascontiguousarray(ob_lon, dtype=np.float64).reshape(P)
            hw = np.ascontiguousarray(ob_halfwidth, dtype=np.float64).reshape(P)
        return val, err, asm, lat, lon, hw

    @staticmethod
    def _diag_arrays(P):
        return dict(prior_mean=np.full(P, np.nan), prior_var=np.full(P, np.nan),
                    post_mean=np.full(P, np.nan), post_var=np.full(P, np.nan),
                    assimilated=np.zeros(P, dtype=np.uint8))

    @staticmethod
    def _grid(loc_mode, grid_lat, grid_lon):
        if loc_mode != LOC_GC:
            return None, None, 0
        glat = np.ascontiguousarray(grid_lat, dtype=np.float64).reshape(-1)
        glon = np.ascontiguousarray(grid_lon, dtype=np.float64).reshape(-1)
        assert glat.shape == glon.shape
        return glat, glon, glat.shape[0]

    def obs_phase(self, M, P, ym, Yp, ob_value, ob_error, ob_assim, loc_mode=LOC_NONE,
                  ob_lat=None, ob_lon=None, ob_halfwidth=None):
        """Phase A.  Returns the per-ob diagnostics dict."""
        val, err, asm, lat, lon, hw = self._ob_arrays(P, ob_value, ob_error, ob_assim, loc_mode,
                                                      ob_lat, ob_lon, ob_halfwidth)
        d = self._diag_arrays(P)
        _check(self.lib, self.lib.efa_obs_phase_dev(
            self.handle, M, P, self._addr(ym), self._addr(Yp), _dp(val), _dp(err), _u8p(asm), loc_mode,
            _dp(lat), _dp(lon), _dp(hw), _dp(d["prior_mean"]), _dp(d["prior_var"]), _dp(d["post_mean"]),
            _dp(d["post_var"]), _u8p(d["assimilated"])))
        d["assimilated"] = d["assimilated"].astype(bool)
        return d

    def state_phase(self, rows, M, xm_in, Xp_in, xm_out, Xp_out, grid_lat=None, grid_lon=None, n_lead=1):
        loc_mode = LOC_GC if grid_lat is not None else LOC_NONE
        glat, glon, ncol = self._grid(loc_mode, grid_lat, grid_lon)
        if loc_mode == LOC_NONE:
            ncol, n_lead = rows, 1
        _check(self.lib, self.lib.efa_state_phase_dev(
            self.handle, rows, M, self._addr(xm_in), self._addr(Xp_in), self._addr(xm_out),
            self._addr(Xp_out), _dp(glat), _dp(glon), ncol, n_lead))

    def state_cycle(self, rows, M, X, post, grid_lat=None, grid_lon=None, n_lead=1):
        loc_mode = LOC_GC if grid_lat is not None else LOC_NONE
        glat, glon, ncol = self._grid(loc_mode, grid_lat, grid_lon)
        if loc_mode == LOC_NONE:
            ncol, n_lead = rows, 1
        _check(self.lib, self.lib.efa_state_cycle_dev(
            self.handle, rows, M, self._addr(X), self._addr(post), _dp(glat), _dp(glon), ncol, n_lead))

    def ensrf_cycle(self, rows, M, P, X, post, ym, Yp, ob_value, ob_error, ob_assim, loc_mode=LOC_NONE,
                    ob_lat=None, ob_lon=None, ob_halfwidth=None, grid_lat=None, grid_lon=None, n_lead=1, obs_block_out=False):
        """Phase A + the state phase on resident prior members in ONE call (efa_ensrf_cycle_dev): Phase B goes into the stream
        behind Phase A without a host round trip when the cycle is unlocalised and takes the transform.  Returns the diagnostics."""
        val, err, asm, lat, lon, hw = self._ob_arrays(P, ob_value, ob_error, ob_assim, loc_mode,
                                                      ob_lat, ob_lon, ob_halfwidth)
        glat, glon, ncol = self._grid(loc_mode, grid_lat, grid_lon)
        if loc_mode == LOC_NONE:
            ncol, n_lead = rows, 1
        d = self._diag_arrays(P)
        _check(self.lib, self.lib.efa_ensrf_cycle_dev(
            self.handle, rows, M, P, self._addr(X), self._addr(post), self._addr(ym), self._addr(Yp), 1 if obs_block_out else 0,
            _dp(val), _dp(err), _u8p(asm), loc_mode, _dp(lat), _dp(lon), _dp(hw), _dp(glat), _dp(glon), ncol, n_lead,
            _dp(d["prior_mean"]), _dp(d["prior_var"]), _dp(d["post_mean"]), _dp(d["post_var"]), _u8p(d["assimilated"])))
        d["assimilated"] = d["assimilated"].astype(bool)
        return d

    def ensrf_update_dev(self, rows, M, P, xm, Xp, ym, Yp, ob_value, ob_error, ob_assim,
                         loc_mode=LOC_NONE, ob_lat=None, ob_lon=None, ob_halfwidth=None,
                         grid_lat=None, grid_lon=None, n_lead=1):
        val, err, asm, lat, lon, hw = self._ob_arrays(P, ob_value, ob_error, ob_assim, loc_mode,
                                                      ob_lat, ob_lon, ob_halfwidth)
        glat, glon, ncol = self._grid(loc_mode, grid_lat, grid_lon)
        if loc_mode == LOC_NONE:
            ncol, n_lead = rows, 1
        d = self._diag_arrays(P)
        _check(self.lib, self.lib.efa_ensrf_update_dev(
            self.handle, rows, M, P, self._addr(xm), self._addr(Xp), self._addr(ym), self._addr(Yp),
            _dp(val), _dp(err), _u8p(asm), loc_mode, _dp(lat), _dp(lon), _dp(hw), _dp(glat), _dp(glon),
            ncol, n_lead, _dp(d["prior_mean"]), _dp(d["prior_var"]), _dp(d["post_mean"]),
            _dp(d["post_var"]), _u8p(d["assimilated"])))
        d["assimilated"] = d["assimilated"].astype(bool)
        return d

    def ensrf_update_host(self, xbm, Xbp, nstate, ob_value, ob_error, ob_assim, loc_mode=LOC_NONE,
                          ob_lat=None, ob_lon=None, ob_halfwidth=None, grid_lat=None, grid_lon=None,
                          n_lead=1):
        """efa_ensrf_update on the reference's augmented host arrays, in place."""
        assert xbm.dtype == np.float64 and Xbp.dtype == np.float64
        assert xbm.flags["C_CONTIGUOUS"] and Xbp.flags["C_CONTIGUOUS"]
        A, M = Xbp.shape
        P = A - nstate
        val, err, asm, lat, lon, hw = self._ob_arrays(P, ob_value, ob_error, ob_assim, loc_mode,
                                                      ob_lat, ob_lon, ob_halfwidth)
        glat, glon, ncol = self._grid(loc_mode, grid_lat, grid_lon)
        if loc_mode == LOC_NONE:
            ncol, n_lead = nstate, 1
        d = self._diag_arrays(P)
        _check(self.lib, self.lib.efa_ensrf_update(
            self.handle, A, nstate, M, P, _dp(xbm), _dp(Xbp), _dp(val), _dp(err), _u8p(asm), loc_mode,
            _dp(lat), _dp(lon), _dp(hw), _dp(glat), _dp(glon), ncol, n_lead,
            _dp(d["prior_mean"]), _dp(d["prior_var"]), _dp(d["post_mean"]), _dp(d["post_var"]),
            _u8p(d["assimilated"])))
        d["assimilated"] = d["assimilated"].astype(bool)
        return d

    def cov_contract_f32(self, N, M, P, Xbp_f32, Ye_f32, C_f32):
        """C (N x P) = Xbp (N x M) . Ye^T (P x M), float32, device addresses."""
        _check(self.lib, self.lib.efa_cov_contract_f32_dev(self.handle, N, M, P, self._addr(Xbp_f32),
                                                           self._addr(Ye_f32), self._addr(C_f32)))

    def malloc_bytes(self, nbytes):
        ptr = ctypes.c_void_p()
        _check(self.lib, self.lib.efa_malloc(self.handle, int(nbytes), ctypes.byref(ptr)))
        return ptr

    def free_bytes(self, ptr):
        _check(self.lib, self.lib.efa_free(self.handle, ptr))

    def h2d(self, dst_ptr, host):
        host = np.ascontiguousarray(host)
        _check(self.lib, self.lib.efa_memcpy_h2d(self.handle, dst_ptr, host.ctypes.data, host.nbytes))

    def d2h(self, host_out, src_ptr):
        assert host_out.flags["C_CONTIGUOUS"]
        _check(self.lib, self.lib.efa_memcpy_d2h(self.handle, host_out.ctypes.data, src_ptr, host_out.nbytes))

    def last_timing(self):
        s = ctypes.c_double(0)
        o = ctypes.c_double(0)
        n = ctypes.c_long(0)
        p = ctypes.c_int(0)
        _check(self.lib, self.lib.efa_last_timing(self.handle, ctypes.byref(s), ctypes.byref(o),
                                                  ctypes.byref(n), ctypes.byref(p)))
        return dict(state_ms=s.value, obs_ms=o.value, state_launches=n.value, path=p.value)

    # -- multi-GPU: the communicator the context owns (RCCL) -----------------------------------------
    def comm_unique_id(self):
        """A fresh RCCL id (bytes) for rank 0 to hand to the other ranks."""
        buf = np.zeros(COMM_ID_BYTES, dtype=np.uint8)
        _check(self.lib, self.lib.efa_comm_unique_id(_u8p(buf)))
        return buf.tobytes()

    def comm_init(self, comm_id, rank, world):
        buf = np.frombuffer(bytes(comm_id), dtype=np.uint8).copy()
        assert buf.size == COMM_ID_BYTES
        _check(self.lib, self.lib.efa_comm_init(self.handle, _u8p(buf), int(rank), int(world)))

    def comm_destroy(self):
        _check(self.lib, self.lib.efa_comm_destroy(self.handle))

    def allreduce_sum(self, buf, count):
        """buf[count] (device, float64) <- sum over the ranks, in place, on the context's stream."""
        _check(self.lib, self.lib.efa_allreduce_sum_dev(self.handle, self._addr(buf), int(count)))

    def gc_block_counts(self, grid_lat, grid_lon, ob_lat, ob_lon, ob_halfwidth, ob_assim):
        """Per block of 16 (y,x) columns under Gaspari-Cohn localisation: the length of its active list and its
        (column, ob) pairs with a non-zero weight; and the total of the pairs."""
        glat = np.ascontiguousarray(grid_lat, dtype=np.float64).reshape(-1)
        glon = np.ascontiguousarray(grid_lon, dtype=np.float64).reshape(-1)
        ncol = glat.shape[0]
        P = len(ob_lat)
        lat = np.ascontiguousarray(ob_lat, dtype=np.float64).reshape(P)
        lon = np.ascontiguousarray(ob_lon, dtype=np.float64).reshape(P)
        hw = np.ascontiguousarray(ob_halfwidth, dtype=np.float64).reshape(P)
        asm = np.ascontiguousarray(np.asarray(ob_assim).astype(bool), dtype=np.uint8).reshape(P)
        cnt = np.zeros((ncol + 15) // 16, dtype=np.int32)
        bp = np.zeros_like(cnt)
        pairs = ctypes.c_uint64(0)
        i32p = ctypes.POINTER(ctypes.c_int32)
        _check(self.lib, self.lib.efa_gc_block_counts(self.handle, ncol, _dp(glat), _dp(glon), P, _dp(lat), _dp(lon), _dp(hw),
                                                      _u8p(asm), cnt.ctypes.data_as(i32p), bp.ctypes.data_as(i32p),
                                                      ctypes.byref(pairs)))
        return cnt, bp, int(pairs.value)

    def fill_synthetic(self, rows, row_offset, M, seed, sigma, X):
        _check(self.lib, self.lib.efa_fill_synthetic_dev(self.handle, rows, row_offset, M, int(seed),
                                                         float(sigma), self._addr(X)))


_contexts = {}


def get_context(device=0):
    """Process-wide cached context per device."""
    device = int(device)
    ctx = _contexts.get(device)
    if ctx is None or ctx.handle is None:
        ctx = Context(device)
        _contexts[device] = ctx
    return ctx
