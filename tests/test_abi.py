"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares.
No compute call is made here (there is no GPU in the build container)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from seismic_bpmf_amd import build
    return build.build_lib()


def header_functions():
    txt = open(os.path.join(ROOT, "include", "bpmf_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bpmf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built):
    import ctypes
    names = header_functions()
    assert len(names) >= 15
    handle = ctypes.CDLL(built)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/bpmf_hip.h but not exported"


def test_python_binding_lists_every_declared_symbol(built):
    from seismic_bpmf_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()
    _lib.lib()


def test_code_object_is_gfx950(built):
    blob = open(built, "rb").read()
    assert b"gfx950" in blob
    assert b"gfx942" not in blob and b"sm_" not in blob[:0]  # single-target build


def test_no_cpu_fallback_in_call_surface():
    import seismic_bpmf_amd as sb
    tp = np.zeros((1, 1, 1, 8), np.float32)
    d = np.zeros((1, 1, 64), np.float32)
    with pytest.raises(ValueError, match="no CPU implementation"):
        sb.matched_filter(tp, np.zeros((1, 1, 1), np.int32), np.ones((1, 1, 1), np.float32), d, 1, arch="cpu")
    with pytest.raises(ValueError, match="no CPU implementation"):
        sb.beamform(d, np.zeros((2, 1, 1), np.int64), np.ones((1, 1, 1), np.float32),
                    np.ones((2, 1), np.float32), device="cpu")


def test_argument_validation_happens_before_any_device_work():
    import seismic_bpmf_amd as sb
    with pytest.raises(ValueError):
        sb.matched_filter(np.zeros((2, 3, 3, 8), np.float32), np.zeros((2, 3, 3)), np.zeros((2, 3, 3)),
                          np.zeros((4, 3, 100), np.float32), 1)     # station count mismatch
    with pytest.raises(ValueError):
        sb.beamform(np.zeros((3, 3, 100), np.float32), np.zeros((5, 3, 2)), np.zeros((3, 3, 2)),
                    np.zeros((5, 4)))                                # weights_sources shape
    with pytest.raises(NotImplementedError):
        sb.matched_filter(np.zeros((1, 1, 1, 8), np.float32), np.zeros((1, 1, 1)), np.ones((1, 1, 1)),
                          np.zeros((1, 1, 64), np.float32), 1, normalize="full")


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: no file of the package may reference it."""
    pkg = os.path.join(ROOT, "seismic_bpmf_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, fn
                assert "liboracle" not in txt, fn
