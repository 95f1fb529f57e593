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

    def _inflation_factor(self):
        """Only the constant-float form of assimilation.py:62-69 is supported;
        it is applied to the perturbations on the GPU and -- unlike the
        reference, which overwrites the caller's state (assimilation.py:67) --
        leaves `self.prior` untouched."""
        if self.inflation is None:
            return 1.0
        if isinstance(self.inflation, (float, np.floating)):
            return float(self.inflation)
        raise NotImplementedError(
            "inflation=%r: only None or a float is supported (dict / file inflation of "
            "assimilation.py:71-114 needs xarray broadcasting and is out of scope)" % (self.inflation,))

    def inflate_state(self):
        """Kept for API compatibility (assimilation.py:52-118): records that the
        constant factor will be applied when the perturbations are formed."""
        if self.is_inflated:
            print("State already inflated.  Skipping additional inflation.")
            return
        self._inflation_factor()
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
        ctx.form_perts(P, M, d, m, d, scale=self._inflation_factor())
        return m.download(), d.download()

    def format_prior_state(self):
        """Augmented (xbm, Xbp): state rows then one row per ob
        (assimilation.py:120-154)."""
        if self.inflation is not None:
            self.inflate_state()
        obmeans, obperts = self.compute_ob_priors()
        X = np.ascontiguousarray(self.prior.to_vect(), dtype=np.float64)
        N, M = X.shape
        ctx = self._context()
        d = ctx.to_device(X)
        m = ctx.empty((N,))
        ctx.form_perts(N, M, d, m, d, scale=self._inflation_factor())
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
