"""Import shim: lets ``import fast_matched_filter as fmf`` (BPMF/similarity_search.py:9,
BPMF/dataset.py:4745) resolve to the MI355X implementation without touching BPMF.

Put ``<repo>/shims`` and ``<repo>`` on ``sys.path`` (or install them) *instead of* the upstream
package.  Only the GPU path exists: BPMF must be driven with ``device="gpu"``.
"""
from seismic_bpmf_amd.matched_filter import matched_filter  # noqa: F401

__bpmf_shim__ = True      # tools/diff_upstream.py refuses to diff the build against itself
__all__ = ["matched_filter"]
