"""``beampower.beampower`` sub-module, as accessed by BPMF (``bp.beampower.beamform``)."""
from seismic_bpmf_amd.beampower import beamform  # noqa: F401

__all__ = ["beamform"]
