"""Per-observation assimilation statistics table.

Counterpart of the reference's `obs_assimilation_statistics(prior, post, obs)`
(efa_xray/postprocess/postprocess.py:8-39): one row per observation with its
location, error, value, the `assimilated` flag and prior/posterior ensemble
mean and variance in observation space.

The reference re-runs `ob.estimate()` on the prior and on the posterior state
(two more forward-operator sweeps).  When the observations come out of
`EnSRF.update()` the same four numbers are already on them -- the kernel's
diagnostics (ensrf.py:66,70,146-149) -- so `from_diagnostics=True` builds the
table without touching the states.  Note the two are not the same quantity for
the posterior: the reference's table interpolates the FINAL posterior state,
the diagnostics are the ob's posterior at the moment it was assimilated.
"""
import numpy as np

from efa_xray_amd.state.ensemble import EnsembleState


def obs_assimilation_statistics(prior, post, obs, from_diagnostics=False):
    """Returns a pandas DataFrame (or a list of dicts if pandas is missing)."""
    assert isinstance(prior, EnsembleState)
    assert isinstance(post, EnsembleState)
    rows = []
    t0 = prior.ensemble_times()[0]
    for ob in obs:
        d = {}
        d['validtime'] = ob.time
        try:
            lead = ob.time - t0
            d['flead'] = lead.total_seconds() / 3600 if hasattr(lead, "total_seconds") else float(lead) / 3600.0
        except TypeError:
            d['flead'] = None
        d['lat'] = ob.lat
        d['lon'] = ob.lon
        d['obtype'] = ob.obtype
        d['description'] = ob.description
        d['ob error'] = ob.error
        d['value'] = ob.value
        d['assimilated'] = ob.assimilated
        if from_diagnostics:
            d['prior mean'] = ob.prior_mean
            d['post mean'] = ob.post_mean if ob.assimilated else ob.prior_mean
            d['prior variance'] = ob.prior_var
            d['post variance'] = ob.post_var if ob.assimilated else ob.prior_var
        else:
            prior_ye = ob.estimate(prior)
            post_ye = ob.estimate(post)
            d['prior mean'] = prior_ye.mean()
            d['post mean'] = post_ye.mean()
            d['prior variance'] = prior_ye.var()
            d['post variance'] = post_ye.var()
        rows.append(d)
    try:
        import pandas as pd
        return pd.DataFrame(rows)
    except ImportError:
        return rows
