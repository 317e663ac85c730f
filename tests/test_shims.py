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


def test_cpu_arch_is_refused_unless_the_user_opts_in(monkeypatch):
    """The reference's defaults are arch / device = "cpu"; the package has no CPU path and says so --
    or, with BPMF_AMD_ACCEPT_CPU_ARCH=1, serves the call on the MI355X (never on a CPU)."""
    import pytest
    import importlib
    mfm = importlib.import_module("seismic_bpmf_amd.matched_filter")
    monkeypatch.delenv("BPMF_AMD_ACCEPT_CPU_ARCH", raising=False)
    for kw, v in (("arch", "cpu"), ("device", "cpu"), ("arch", "precise")):
        with pytest.raises(ValueError, match="no CPU implementation"):
            mfm.require_gpu_arch(v, kw)
    mfm.require_gpu_arch("gpu", "arch")
    monkeypatch.setenv("BPMF_AMD_ACCEPT_CPU_ARCH", "1")
    mfm.require_gpu_arch("cpu", "arch")            # accepted: the caller proceeds to the HIP library
