import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
