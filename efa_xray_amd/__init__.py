"""efa_xray_amd: MI355X-native serial EnSRF assimilation update.

A from-scratch gfx950 implementation of the hot path of lmadaus/efa_xray
(`efa_xray/assimilation`): hand-written HIP kernels behind a C ABI
(`include/efa_hip.h`, `libefa_hip.so`), with the reference's Python surface
(`EnsembleState`, `Observation`, `EnSRF(...).update()`) on top.
"""
from efa_xray_amd.state.ensemble import EnsembleState
from efa_xray_amd.observation.observation import Observation, gaspari_cohn, haversine
from efa_xray_amd.assimilation.assimilation import Assimilation
from efa_xray_amd.assimilation.ensrf import EnSRF

__all__ = ["EnsembleState", "Observation", "gaspari_cohn", "haversine", "Assimilation", "EnSRF"]
__version__ = "0.1.0"
