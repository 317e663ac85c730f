"""Build libbpmf_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

The library is compiled ahead of time into seismic_bpmf_amd/lib/ so that it travels with the
source tree (no JIT cache).  hipcc cross-compiles gfx950 without a GPU present.
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libbpmf_hip.so")
SOURCES = ["mf.hip", "bp.hip", "post.hip", "decimate.hip", "util.hip"]
ARCH = "gfx950"
# -ffp-contract=off: the kernels spell out every fmaf; the compiler must not fuse more.
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-function"]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _stale():
    if not os.path.exists(LIBPATH):
        return True
    t = os.path.getmtime(LIBPATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(_HERE, "..", "include", "bpmf_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force=False, verbose=False):
    """Compile every HIP source into lib/libbpmf_hip.so for gfx950.  Returns the path."""
    if not force and not _stale():
        return LIBPATH
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [find_hipcc(), f"--offload-arch={ARCH}"] + FLAGS + srcs + ["-o", LIBPATH]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose and res.stderr:
        print(res.stderr)
    return LIBPATH


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
