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
OBJDIR = os.path.join(LIBDIR, "obj")
SOURCES = ["mf.hip", "mf_split.hip", "bp.hip", "bp_fast.hip", "bp_direct.hip", "bp_detect.hip", "stats.hip", "intertp.hip", "post.hip", "decimate.hip", "util.hip", "multi.hip", "context.hip"]
ARCH = "gfx950"
# -ffp-contract=off: the kernels spell out every fmaf; the compiler must not fuse more.
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-function"]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(_HERE, "..", "include", "bpmf_hip.h"))
    return [d for d in deps if os.path.exists(d)]


def _obj_of(src):
    return os.path.join(OBJDIR, os.path.basename(src) + ".o")


def _obj_stale(src):
    obj = _obj_of(src)
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + _deps())


def _stale():
    if not os.path.exists(LIBPATH):
        return True
    t = os.path.getmtime(LIBPATH)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    return any(os.path.getmtime(d) > t for d in srcs + _deps() if os.path.exists(d))


def build_lib(force=False, verbose=False):
    """Compile every HIP source for gfx950 (one hipcc per source, in parallel; objects of unchanged
    sources are reused unless `force`) and link lib/libbpmf_hip.so.  Returns the path."""
    if not force and not _stale():
        return LIBPATH
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = find_hipcc()
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cflags = [f for f in FLAGS if f != "-shared"]

    def compile_one(src):
        if not force and not _obj_stale(src):
            return src, 0, ""
        cmd = [hipcc, f"--offload-arch={ARCH}"] + cflags + ["-c", src, "-o", _obj_of(src)]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        return src, res.returncode, res.stdout + res.stderr

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as pool:
        results = list(pool.map(compile_one, srcs))
    for src, rc, log in results:
        if rc != 0:
            raise RuntimeError(f"hipcc failed on {os.path.basename(src)}:\n{log}")
        if verbose and log:
            print(log)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC"] + [_obj_of(s) for s in srcs] + \
          ["-o", LIBPATH]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    return LIBPATH


if __name__ == "__main__":
    import sys
    print(build_lib(force="--incremental" not in sys.argv, verbose=True))
