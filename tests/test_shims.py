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


def test_cpu_arch_is_refused_unless_the_user_opts_in():
    """The reference's defaults are arch / device = "cpu"; the package has no CPU path and says so --
    or, after seismic_bpmf_amd.accept_cpu_arch(), serves the call on the MI355X (never on a CPU).  The
    package reads nothing from the environment."""
    import pytest
    import importlib
    import inspect
    import seismic_bpmf_amd
    mfm = importlib.import_module("seismic_bpmf_amd.matched_filter")
    seismic_bpmf_amd.accept_cpu_arch(False)
    for kw, v in (("arch", "cpu"), ("device", "cpu"), ("arch", "precise")):
        with pytest.raises(ValueError, match="no CPU implementation"):
            mfm.require_gpu_arch(v, kw)
    mfm.require_gpu_arch("gpu", "arch")
    seismic_bpmf_amd.accept_cpu_arch(True)
    try:
        mfm.require_gpu_arch("cpu", "arch")            # accepted: the caller proceeds to the HIP library
    finally:
        seismic_bpmf_amd.accept_cpu_arch(False)
    assert "os.environ" not in inspect.getsource(mfm) and "getenv" not in inspect.getsource(mfm)
