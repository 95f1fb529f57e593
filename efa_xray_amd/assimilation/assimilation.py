"""Assimilation base class: the bookkeeping either side of the EnSRF loop.

Same contract as the reference's `Assimilation`
(efa_xray/assimilation/assimilation.py:10-171): constructor
`(state, obs, nproc=1, inflation=None, verbose=False)` and the helpers
`compute_ob_priors`, `inflate_state`, `format_prior_state`,
`format_posterior_state`.  The array work of the two `format_*` helpers runs on
the GPU through libefa_hip (efa_form_perts_dev / efa_posterior_dev).
"""
from collections import OrderedDict
from copy import deepcopy

import numpy as np

from efa_xray_amd import _lib


def _read_inflation_file(filename):
    """{variable: (dims, float64 array)} of a netCDF file of inflation factors (assimilation.py:71-79 opens it with
    xarray.open_dataset).  xarray reads it when importable; otherwise scipy.io.netcdf_file (classic netCDF-3)."""
    try:
        import xarray
    except ImportError:
        xarray = None
    out = {}
    if xarray is not None:
        with xarray.open_dataset(filename) as ds:
            for name in ds.data_vars:
                out[name] = (tuple(ds[name].dims), np.asarray(ds[name].values, dtype=np.float64))
        return out
    from scipy.io import netcdf_file
    with netcdf_file(filename, "r", mmap=False) as f:
        for name, var in f.variables.items():
            if name in f.dimensions and len(var.dimensions) == 1:  # a coordinate variable
                continue
            out[name] = (tuple(var.dimensions), np.array(var[:], dtype=np.float64))
    return out


class Assimilation(object):
    def __init__(self, state, obs, nproc=1, inflation=None, verbose=False, device=0):
        # assimilation.py:15-33.  The reference also deep-copies the state into
        # `self.post` here and never uses it; that copy is not made.
        self.prior = state
        self.obs = obs
        self.verbose = verbose
        self.nproc = nproc          # stored and unused, as in the reference
        self.inflation = inflation
        self.is_inflated = False
        self.device = device

    # ------------------------------------------------------------------
    def _context(self):
        return _lib.get_context(self.device)

    _DIM_AXIS = {"validtime": 0, "y": 1, "x": 2}

    def inflate_state(self):
        """Inflate the prior as specified by `inflation` (assimilation.py:52-118).

        Called by `format_prior_state()` -- and therefore by `EnSRF.update()` -- whenever
        `inflation is not None` (assimilation.py:131-134, reached from ensrf.py:44); callable on
        its own too.  It changes `self.prior`: a float scales the ensemble perturbations of every
        variable IN PLACE, so the caller's state object is inflated as well (assimilation.py:62-69);
        a dict maps variable names to float factors (in place, :103-113) or the dimension names
        'validtime' / 'y' / 'x' to arrays of per-index factors that are broadcast over the other
        dimensions (:82-100; there the reference rebinds `self.prior` to a new object and leaves
        the caller's alone).  A second call is a no-op (:57-59).  A string names a netCDF file of factors
        (:71-79): variables of the same names as the state's, on any subset of its dimensions, broadcast by
        dimension name as xarray does; `self.prior` is rebound.  Read with xarray when importable, otherwise
        with scipy.io.netcdf_file (classic netCDF-3).

        PARITY UNPINNED: the reference's inflation code needs a real xarray Dataset, which the
        build image lacks, so no golden vector covers it; this follows the source text."""
        if self.is_inflated:
            print("State already inflated.  Skipping additional inflation.")
            return
        prior = self.prior

        def scale_var(name, factor, inplace=True):
            v = prior.variables[name]
            mean = v.mean(axis=-1, keepdims=True)
            if self.verbose:
                print(name, "BEFORE stdev:", np.mean(np.std(v, axis=-1), axis=None))
            if inplace:
                v[...] = (v - mean) * factor + mean          # `variables[v][:] = ...`, assimilation.py:67,113
            else:
                prior.variables[name] = np.ascontiguousarray((v - mean) * factor + mean)
            if self.verbose:
                print(name, "AFTER stdev:", np.mean(np.std(prior.variables[name], axis=-1), axis=None))

        if isinstance(self.inflation, (float, np.floating)):
            if self.verbose:
                print("Inflating all variables by factor: {:3.2f}".format(self.inflation))
            for name in prior.vars():
                scale_var(name, float(self.inflation))
        elif isinstance(self.inflation, str):
            # assimilation.py:71-79: `prior = perts * open_dataset(file) + mean` -- xarray multiplies variables of the
            # same name and broadcasts by DIMENSION NAME, and rebinds self.prior (the caller's state stays as it was)
            if self.verbose:
                print("Trying to load inflation from file: {:s}".format(self.inflation))
            factors = _read_inflation_file(self.inflation)
            prior = self.prior = deepcopy(prior)
            shape = prior._first().shape
            for name in prior.vars():
                if name not in factors:
                    # (xarray's Dataset arithmetic would DROP a variable the file does not hold; it is kept, uninflated)
                    print("Inflation file holds no factors for variable {:s}.  Left as it is.".format(name))
                    continue
                dims, fac = factors[name]
                view = [1, 1, 1, 1]
                for d, n in zip(dims, fac.shape):
                    if d not in self._DIM_AXIS and d != "mem":
                        raise ValueError("inflation file: dimension %r of %r is not a state dimension" % (d, name))
                    ax = 3 if d == "mem" else self._DIM_AXIS[d]
                    if n != shape[ax]:
                        raise ValueError("inflation file: %r has %d entries along %r, the state has %d" % (name, n, d, shape[ax]))
                    view[ax] = n
                order = sorted(range(len(dims)), key=lambda q: 3 if dims[q] == "mem" else self._DIM_AXIS[dims[q]])
                scale_var(name, np.transpose(fac, order).reshape(view), inplace=False)
            if self.verbose:
                print("Succeeded inflation from file: {:s}".format(self.inflation))
        else:
            for k, v in self.inflation.items():  # a dictionary, as in the reference
                if k in ("validtime", "lat", "lon", "x", "y"):
                    if k not in self._DIM_AXIS:
                        raise NotImplementedError("inflation along %r: lat/lon are 2-D coordinates here, "
                                                  "use 'y' / 'x'" % k)
                    v = np.asarray(v, dtype=np.float64)
                    axis = self._DIM_AXIS[k]
                    assert v.shape[0] == prior._first().shape[axis]  # assimilation.py:88-89
                    if self.verbose:
                        print("Inflating all variables along {:s} dimension".format(k))
                    # the reference rebinds self.prior to a new object here (assimilation.py:96)
                    prior = self.prior = deepcopy(prior)
                    shape = [1, 1, 1, 1]
                    shape[axis] = v.shape[0]
                    for name in prior.vars():
                        scale_var(name, v.reshape(shape), inplace=False)
                else:
                    assert isinstance(v, (float, np.floating))  # assimilation.py:106
                    if k not in prior.variables:
                        print("Unable to find variable {:s} to inflate.  Skipping...".format(k))
                        continue
                    if self.verbose:
                        print("Inflating variable {:s} by factor: {:3.2f}".format(k, v))
                    scale_var(k, float(v))
        self.is_inflated = True

    # -- the state between host and device: one pass each way, variable by variable ----------------
    def _upload_prior(self, ctx):
        """The prior as an (nstate, nmems) device array in `to_vect()` order (ensemble.py:110-114), copied slab by slab
        from each variable's own (nt, ny, nx, nmem) array: no stacked host copy of the state is made."""
        prior = self.prior
        N, M = prior.nstate(), prior.nmems()
        X = ctx.empty((max(N, 1), M))
        per = prior.ntimes() * prior.ny() * prior.nx()
        for iv, name in enumerate(prior.vars()):
            X.upload_rows(iv * per, prior.variables[name])
        return X

    def _download_posterior(self, X):
        """A NEW state object like the prior (assimilation.py:165 deep-copies it) whose variables are the rows of the device
        array `X`: coordinates are copied, member arrays are downloaded straight into their own fresh arrays."""
        prior = self.prior
        per = prior.ntimes() * prior.ny() * prior.nx()
        shape = prior.shape()[1:]
        variables = OrderedDict()
        for iv, name in enumerate(prior.vars()):
            variables[name] = X.download_rows_into(iv * per, np.empty(shape, dtype=np.float64))
        return type(prior)(variables, deepcopy(prior.coords))

    def _default_forward_operator(self):
        """True when every ob uses `Observation.estimate` as shipped (the reference's point interpolation,
        observation.py:40-50): then all P estimates are computed on the device.  An ob whose class or
        instance overrides `estimate` (any other forward operator) is evaluated on the host, one by one."""
        from efa_xray_amd.observation.observation import Observation
        return len(self.obs) > 0 and all(
            type(ob).estimate is Observation.estimate and "estimate" not in vars(ob) for ob in self.obs)

    def _time_axis(self):
        """The state's valid times and the obs' times on one float axis (seconds for datetime64)."""
        valids = np.asarray(self.prior.ensemble_times())
        if valids.dtype.kind == "M":
            t0 = valids[0]
            vt = (valids - t0) / np.timedelta64(1, "s")
            ot = np.array([(np.datetime64(ob.time) - t0) / np.timedelta64(1, "s") for ob in self.obs], dtype=np.float64)
        else:
            vt = valids.astype(np.float64)
            ot = np.array([float(ob.time) for ob in self.obs], dtype=np.float64)
        return vt, ot

    def device_ob_estimates(self, ctx, X_dev):
        """(P, M) estimates of the reference's default forward operator as a device array, computed from the
        state resident at `X_dev` (rows in to_vect() order): the interpolation stencils of all obs are built by
        `efa_interp_stencils` (nearest-4 search, inverse-distance and time weights: ensemble.py:152-224) and
        applied by `efa_forward_interp_dev`.  No per-ob Python loop, no host pass over the grid."""
        prior = self.prior
        names = prior.vars()
        try:
            ob_var = [names.index(ob.obtype) for ob in self.obs]
        except ValueError as e:
            raise KeyError("observation type not in the state: %s" % e)
        vt, ot = self._time_axis()
        nvar, nt, ny, nx, M = prior.shape()
        _, _, status = ctx.interp_stencils(nvar, nt, ny, nx, prior.coords["lat"], prior.coords["lon"], vt, ob_var, ot,
                                           [float(ob.lat) for ob in self.obs], [float(ob.lon) for ob in self.obs],
                                           want_host=False)
        if status.any():
            k = int(np.nonzero(status)[0][0])
            why = {1: "its time is outside the state's valid times (the reference prints 'Interpolation is outside of "
                      "time range in state!' and then fails on None.mean(), assimilation.py:46-47)",
                   2: "a nearest grid index is out of range for 1-D lat/lon (ensemble.py:186-190,227)",
                   3: "bad variable index"}.get(int(status[k]), "status %d" % status[k])
            raise ValueError("observation %d cannot be interpolated: %s" % (k, why))
        P = len(self.obs)
        HX = ctx.empty((P, M))
        ncol = ny * nx
        ctx.forward_interp(ncol, 0, ncol, nvar * nt, M, X_dev, HX)
        return HX

    def compute_ob_estimates(self, X_dev=None):
        """(P, M) ensemble estimates HX[k] = ob_k.estimate(prior): the forward operator loop of
        assimilation.py:45-46.  With the reference's own point-interpolation operator the whole loop runs on
        the device (`device_ob_estimates`, on the resident copy `X_dev` of the prior if the caller has one);
        user-defined `estimate` methods are called one by one."""
        nobs = len(self.obs)
        if self._default_forward_operator():
            ctx = self._context()
            X = X_dev if X_dev is not None else self._upload_prior(ctx)
            return self.device_ob_estimates(ctx, X).download()
        HX = np.zeros((nobs, self.prior.nmems()))
        for k, ob in enumerate(self.obs):
            HX[k, :] = ob.estimate(self.prior)
        return HX

    def compute_ob_priors(self, X_dev=None):
        """Obs-space prior means and perturbations (assimilation.py:36-49)."""
        HX = self.compute_ob_estimates(X_dev)
        P, M = HX.shape
        if P == 0:
            return np.zeros(0), np.zeros((0, M))
        ctx = self._context()
        d = ctx.to_device(HX)
        m = ctx.empty((P,))
        ctx.form_perts(P, M, d, m, d)
        return m.download(), d.download()

    def format_prior_state(self):
        """Augmented (xbm, Xbp): state rows then one row per ob
        (assimilation.py:120-154).  Inflates the prior first when `inflation` is set
        (assimilation.py:131-134)."""
        if self.inflation is not None:
            if self.verbose:
                print("Inflating Prior State")
            self.inflate_state()
        ctx = self._context()
        d = self._upload_prior(ctx)                    # ONE upload serves the forward operator and the perturbations
        obmeans, obperts = self.compute_ob_priors(d)
        N, M = self.prior.nstate(), self.prior.nmems()
        m = ctx.empty((N,))
        ctx.form_perts(N, M, d, m, d)
        xbm = np.hstack((m.download(), obmeans))
        Xbp = np.vstack((d.download(), obperts))
        return xbm, Xbp

    def format_posterior_state(self, xam, Xap):
        """(xam, Xap) -> new state object + the obs list (assimilation.py:157-171)."""
        N = self.prior.nstate()
        M = self.prior.nmems()
        ctx = self._context()
        d = ctx.to_device(np.ascontiguousarray(Xap[:N]))
        m = ctx.to_device(np.ascontiguousarray(xam[:N]))
        ctx.posterior(N, M, m, d, d)
        return self._download_posterior(d), self.obs
