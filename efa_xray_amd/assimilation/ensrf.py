"""EnSRF: serial ensemble square-root filter, computed on an MI355X.

Drop-in for the reference's `EnSRF(Assimilation)`
(efa_xray/assimilation/ensrf.py:8-151): same constructor
`EnSRF(state, obs, nproc=1, inflation=None, verbose=True, loc=False)`, same
`.update() -> (post_state, obs)` with `post_state` a new object and the five
diagnostics (`prior_mean, prior_var, post_mean, post_var, assimilated`,
ensrf.py:66,70,75,146-149) written onto each `Observation` in place.

The per-observation loop (ensrf.py:50-149) does not exist in Python here:
`update()` marshals the state vector and the per-ob scalars into flat arrays
and calls libefa_hip (`efa_obs_phase_dev` + `efa_state_cycle_dev`).  If the
library or a gfx950 GPU is missing the call raises -- there is no NumPy path.
"""
import numpy as np

from efa_xray_amd import _lib
from efa_xray_amd.assimilation.assimilation import Assimilation


class EnSRF(Assimilation):
    def __init__(self, state, obs, nproc=1, inflation=None, verbose=True, loc=False, **kw):
        """Extra keyword-only options (all default to reference behaviour):
        device   -- HIP device ordinal (default 0)
        obs_batch-- observations fused per sweep launch (1..64)
        path     -- 'auto' | 'sweep' | 'transform' (how the state sweep runs)
        """
        device = kw.pop("device", 0)
        self.obs_batch = kw.pop("obs_batch", None)
        self.path = kw.pop("path", None)
        if kw:
            raise TypeError("unexpected keyword arguments %r" % sorted(kw))
        Assimilation.__init__(self, state, obs, nproc, inflation, verbose, device=device)
        self.loc = loc
        self.last_timing = None

    # ------------------------------------------------------------------
    def _loc_mode(self):
        if self.loc in (None, False):
            return _lib.LOC_NONE
        if self.loc == 'GC':
            return _lib.LOC_GC
        # ensrf.py:99-101 passes any other truthy value to Observation.localize,
        # which then hits an unbound local (observation.py:77-87)
        raise ValueError("loc=%r: supported values are None, False and 'GC'" % (self.loc,))

    def _ob_arrays(self, loc_mode):
        obs = self.obs
        P = len(obs)
        value = np.array([float(ob.value) if ob.assimilate_this else
                          (np.nan if ob.value is None else float(ob.value)) for ob in obs], dtype=np.float64)
        error = np.array([float(ob.error) if ob.assimilate_this else
                          (np.nan if ob.error is None else float(ob.error)) for ob in obs], dtype=np.float64)
        assim = np.array([bool(ob.assimilate_this) for ob in obs], dtype=bool)
        lat = lon = hw = None
        if loc_mode == _lib.LOC_GC:
            # the reference reads localize_radius only for obs it assimilates (ensrf.py:74-76 precedes :101)
            for k, ob in enumerate(obs):
                if ob.assimilate_this and ob.localize_radius is None:
                    raise ValueError("observation %d has localize_radius=None but loc='GC' "
                                     "(the reference raises TypeError in abs(None), observation.py:120)" % k)

            # lat/lon of EVERY ob enter the obs-obs taper (ensrf.py:113), assimilated or not
            lat = np.array([float(ob.lat) for ob in obs], dtype=np.float64)
            lon = np.array([float(ob.lon) for ob in obs], dtype=np.float64)
            hw = np.array([float(ob.localize_radius) if ob.localize_radius is not None else np.nan
                           for ob in obs], dtype=np.float64)
        return P, value, error, assim, lat, lon, hw

    def _configure(self, ctx):
        if self.obs_batch is not None:
            ctx.set_option("obs_batch", int(self.obs_batch))
        path = {None: _lib.PATH_AUTO, "auto": _lib.PATH_AUTO, "sweep": _lib.PATH_SWEEP,
                "transform": _lib.PATH_TRANSFORM}[self.path]
        ctx.set_option("path", path)

    # ------------------------------------------------------------------
    def update(self):
        if self.verbose:
            print("Beginning update sequence")
        loc_mode = self._loc_mode()
        P, value, error, assim, lat, lon, hw = self._ob_arrays(loc_mode)
        # ensrf.py:44 -> format_prior_state: the inflation hook comes first (assimilation.py:131-134);
        # it may rebind self.prior (per-dimension factors, assimilation.py:96)
        if self.inflation is not None:
            if self.verbose:
                print("Inflating Prior State")
            self.inflate_state()
        prior = self.prior
        N = prior.nstate()
        M = prior.nmems()

        ctx = self._context()
        self._configure(ctx)
        ctx.set_option("timing", 1)
        if self.verbose:
            print("Converting state to vector")
        X = self._upload_prior(ctx)                            # slab by slab from the variables: no stacked host copy
        # forward operator, once per ob from the prior (assimilation.py:45-48): on the device for the
        # reference's point interpolation, through ob.estimate() for user-defined operators
        if self.verbose:
            print("Computing observation priors")
        ym = ctx.empty((max(P, 1),))
        if P and self._default_forward_operator():
            Yp = self.device_ob_estimates(ctx, X)
        else:
            Yp = ctx.to_device(self.compute_ob_estimates()) if P else ctx.empty((1, M))
        if P:
            ctx.form_perts(P, M, Yp, ym, Yp)                   # assimilation.py:46-48

        grid_lat = grid_lon = None
        n_lead = 1
        if loc_mode == _lib.LOC_GC:
            grid_lat, grid_lon = prior.column_latlon()          # 2-D taper, ensrf.py:108-111
            n_lead = prior.nvars() * prior.ntimes()

        if self.verbose:
            print("Beginning observation loop")
        # Phase A (ensrf.py:50-149 on the obs block) and the state phase in one library call; in place, so nothing is
        # enqueued ahead of Phase A's status (efa_ensrf_cycle_dev speculates only into a separate posterior buffer)
        diag = ctx.ensrf_cycle(N, M, P, X, X, ym, Yp, value, error, assim, loc_mode, lat, lon, hw, grid_lat, grid_lon, n_lead)
        self.last_timing = ctx.last_timing()

        # diagnostics onto the observations, as ensrf.py:66,70,75,146-149
        for k, ob in enumerate(self.obs):
            ob.prior_mean = np.float64(diag["prior_mean"][k])
            ob.prior_var = np.float64(diag["prior_var"][k])
            if diag["assimilated"][k]:
                ob.post_mean = np.float64(diag["post_mean"][k])
                ob.post_var = np.float64(diag["post_var"][k])
                ob.assimilated = True
            else:
                ob.assimilated = False

        if self.verbose:
            print("Formatting posterior")
        # a NEW state object (assimilation.py:165 deep-copies the prior): coordinates copied, member arrays downloaded
        # straight into their own fresh arrays -- the prior's members are not copied only to be overwritten
        post_state = self._download_posterior(X)
        return post_state, self.obs

    # ------------------------------------------------------------------
    def update_arrays(self, xbm, Xbp):
        """Run the loop on the reference's augmented arrays (as returned by
        `format_prior_state`) and return `(xam, Xap)` as handed to
        `format_posterior_state` (ensrf.py:44,151).  Diagnostics are written
        onto the observations."""
        loc_mode = self._loc_mode()
        P, value, error, assim, lat, lon, hw = self._ob_arrays(loc_mode)
        N = self.prior.nstate()
        xam = np.array(xbm, dtype=np.float64, order="C")
        Xap = np.array(Xbp, dtype=np.float64, order="C")
        grid_lat = grid_lon = None
        n_lead = 1
        if loc_mode == _lib.LOC_GC:
            grid_lat, grid_lon = self.prior.column_latlon()
            n_lead = self.prior.nvars() * self.prior.ntimes()
        ctx = self._context()
        self._configure(ctx)
        diag = ctx.ensrf_update_host(xam, Xap, N, value, error, assim, loc_mode, lat, lon, hw,
                                     grid_lat, grid_lon, n_lead)
        for k, ob in enumerate(self.obs):
            ob.prior_mean = np.float64(diag["prior_mean"][k])
            ob.prior_var = np.float64(diag["prior_var"][k])
            if diag["assimilated"][k]:
                ob.post_mean = np.float64(diag["post_mean"][k])
                ob.post_var = np.float64(diag["post_var"][k])
                ob.assimilated = True
            else:
                ob.assimilated = False
        return xam, Xap
