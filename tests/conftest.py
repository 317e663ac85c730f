import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # BPMF_CRASH_BT=1: a process that dies of SIGSEGV / SIGABRT prints the C stack of the faulting thread
    # first (tools/stress/crash_bt.c; stress sessions only -- xdist workers inherit the variable)
    if os.environ.get("BPMF_CRASH_BT"):
        import ctypes
        sys.path.insert(0, os.path.join(ROOT, "tools", "stress"))
        import stress_multi
        handler = ctypes.CDLL(stress_multi.build_crash_lib())
        handler.crash_bt_install(f"pytest-{os.environ.get('PYTEST_XDIST_WORKER', 'main')}".encode())
        config._bpmf_crash_bt = handler
    # BPMF_STRESS_OLD_LIB=path: run the suite against an OLDER build of the library (the round-3 build kept
    # beside the stress tools, to reproduce its crash with native backtraces): symbols that build lacks
    # are dropped from the binding's list
    old = os.environ.get("BPMF_STRESS_OLD_LIB")
    if old:
        import ctypes
        from seismic_bpmf_amd import _lib
        _lib.LIBPATH = os.path.join(ROOT, old)
        have = ctypes.CDLL(_lib.LIBPATH)
        for name in [n for n in _lib.SIGNATURES if not hasattr(have, n)]:
            del _lib.SIGNATURES[name]


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    from oracle import oracle
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        oracle.build()
    return oracle


@pytest.fixture
def hip_opts():
    """Set execution options of libbpmf_hip.so (bpmf_set_option: kernel family, plan sizes, ... --
    never a result); everything touched goes back to its default when the test ends."""
    from seismic_bpmf_amd import _lib
    touched = set()

    def set_(name, value):
        touched.add(name)
        _lib.set_option(name, value)

    def reset(name):
        _lib.set_option(name, _lib.get_option(name)[1])

    set_.reset = reset
    yield set_
    for name in touched:
        reset(name)


@pytest.fixture(autouse=True, scope="session")
def _poisoned_outputs(request):
    """GPU sessions run with option debug.poison_output on: the max-beam / arg-max / CC-sum outputs are
    filled with 0xFF bytes before the kernels run, so a sample that no kernel writes fails its comparison
    whatever the buffer held before (round 4: the last partial tile of a series was left unwritten when
    every used moveout was negative, and passed whenever the allocator handed back a buffer that already
    held the right values)."""
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from seismic_bpmf_amd import _lib
    _lib.set_option("debug.poison_output", 1)
    yield
    _lib.set_option("debug.poison_output", 0)

