"""Build a SECOND copy of the library with the in-kernel cycle accounting compiled in
(-DBPMF_PHASE_CYCLES: s_memtime at the phase boundaries of the two hot kernels, csrc/bp_fast.hip and
csrc/mf.hip) as tools/phase/libbpmf_hip_phase.so.  The shipping library carries none of it.

    python tools/phase/build_phase_lib.py        (cross-compiles without a GPU; ~2 min)
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from seismic_bpmf_amd import build as b  # noqa: E402

OUT = os.path.join(ROOT, "tools", "phase", "libbpmf_hip_phase.so")
OBJ = os.path.join(ROOT, "tools", "phase", "obj")


def main():
    os.makedirs(OBJ, exist_ok=True)
    hipcc = b.find_hipcc()
    flags = [f for f in b.FLAGS if f != "-shared"] + ["-DBPMF_PHASE_CYCLES"]
    srcs = [os.path.join(b.CSRC, s) for s in b.SOURCES]

    def one(src):
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        res = subprocess.run([hipcc, f"--offload-arch={b.ARCH}"] + flags + ["-c", src, "-o", obj],
                             capture_output=True, text=True)
        if res.returncode:
            raise RuntimeError(res.stderr[-3000:])
        return obj

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as pool:
        objs = list(pool.map(one, srcs))
    subprocess.check_call([hipcc, f"--offload-arch={b.ARCH}", "-shared", "-fPIC"] + objs + ["-o", OUT])
    print(OUT)


if __name__ == "__main__":
    main()
