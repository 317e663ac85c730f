"""Import shim: ``import beampower as bp`` then ``bp.beampower.beamform(...)``
(BPMF/template_search.py:12, 549-569) resolves to the MI355X implementation."""
from . import beampower  # noqa: F401
from .beampower import beamform  # noqa: F401

__bpmf_shim__ = True      # tools/diff_upstream.py refuses to diff the build against itself
__all__ = ["beampower", "beamform"]
