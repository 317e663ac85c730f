"""GPU: the C ABI from plain C.  tests/c_abi/abi_smoke.c is compiled with gcc against include/bpmf_hip.h and
linked to libbpmf_hip.so -- no Python, no torch types at the boundary -- and runs both host-pointer entry points
against a scalar restatement of the conventions written in the C file itself."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_program_links_and_matches(tmp_path):
    import torch  # the library binds to the HIP runtime torch ships; put that directory on the loader path
    libdir = os.path.join(ROOT, "seismic_bpmf_amd", "lib")
    exe = str(tmp_path / "abi_smoke")
    cmd = ["gcc", "-std=c99", "-O1", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", "abi_smoke.c"), "-o", exe, "-L", libdir, "-lbpmf_hip", "-lm",
           "-Wl,--allow-shlib-undefined"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    env = dict(os.environ, LD_LIBRARY_PATH=os.pathsep.join([libdir, torch_lib, "/opt/rocm/lib", os.environ.get("LD_LIBRARY_PATH", "")]))
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert run.stdout.startswith("ok:"), run.stdout
