"""seismic_bpmf_amd -- MI355X-native hot paths of the Seismic_BPMF workflow.

Two data-parallel paths, hand-written in HIP for gfx950 behind the call surface the
reference already uses:

* :func:`matched_filter`  <-> ``fast_matched_filter.matched_filter``
  (BPMF/similarity_search.py:526-533, BPMF/dataset.py:4818-4827)
* :func:`beamform`        <-> ``beampower.beampower.beamform``
  (BPMF/template_search.py:549-569)

The package has no CPU implementation: without ``lib/libbpmf_hip.so`` (built by
``python -m seismic_bpmf_amd.build``) and a HIP device every entry point raises.
"""
from ._lib import (BpmfHipError, device_count, device_info, device_memory_held,  # noqa: F401
                   release_device_memory, set_option, get_option, compat_profile)
from .beampower import BeamformerGPU, beamform  # noqa: F401
from .matched_filter import MatchedFilterGPU, accept_cpu_arch, matched_filter  # noqa: F401

__version__ = "0.1.0"
