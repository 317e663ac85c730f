"""CPU: the import shims expose the names BPMF imports (similarity_search.py:9, template_search.py:12)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shims_resolve_to_the_hip_call_surface():
    sys.path.insert(0, os.path.join(ROOT, "shims"))
    try:
        import beampower as bp
        import fast_matched_filter as fmf
        import seismic_bpmf_amd as sb
        assert fmf.matched_filter is sb.matched_filter
        assert bp.beampower.beamform is sb.beamform
    finally:
        sys.path.remove(os.path.join(ROOT, "shims"))
        for m in ("beampower", "beampower.beampower", "fast_matched_filter"):
            sys.modules.pop(m, None)
