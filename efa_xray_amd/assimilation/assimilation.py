"""Assimilation base class: the bookkeeping either side of the EnSRF loop.

Same contract as the reference's `Assimilation`
(efa_xray/assimilation/assimilation.py:10-171): constructor
`(state, obs, nproc=1, inflation=None, verbose=False)` and the helpers
`compute_ob_priors`, `inflate_state`, `format_prior_state`,
`format_posterior_state`.  The array work of the two `format_*` helpers runs on
the GPU through libefa_hip (efa_form_perts_dev / efa_posterior_dev).
"""
from copy import deepcopy

import numpy as np

from efa_xray_amd import _lib


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
        the caller's alone).  A second call is a no-op (:57-59).  The file form (:71-79, an
        xarray/netCDF dataset of factors) is not supported: SURVEY.md 8(f4).

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
            raise NotImplementedError("inflation from a file (assimilation.py:71-79) needs an xarray/netCDF dataset "
                                      "of factors; pass a float or a dict instead")
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

    def compute_ob_estimates(self):
        """(P, M) ensemble estimates HX[k] = ob_k.estimate(prior): the forward
        operator loop of assimilation.py:45-46."""
        nobs = len(self.obs)
        HX = np.zeros((nobs, self.prior.nmems()))
        for k, ob in enumerate(self.obs):
            HX[k, :] = ob.estimate(self.prior)
        return HX

    def compute_ob_priors(self):
        """Obs-space prior means and perturbations (assimilation.py:36-49)."""
        HX = self.compute_ob_estimates()
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
        obmeans, obperts = self.compute_ob_priors()
        X = np.ascontiguousarray(self.prior.to_vect(), dtype=np.float64)
        N, M = X.shape
        ctx = self._context()
        d = ctx.to_device(X)
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
        post_state = deepcopy(self.prior)
        post_state.from_vect(d.download())
        return post_state, self.obs
