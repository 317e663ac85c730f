"""Option mf.fused_prologue (per-template preparation inside the L <= 257 kernel, one launch per call) on the
small problems it is for: BASELINE configs[0] (one hour @ 50 Hz, 4 templates, 8 x 3 channels, L = 128) and
the same hour with more templates.  Kernel time from the library's events, the whole call from the host
clock over back-to-back resident calls."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import seismic_bpmf_amd as sb
from seismic_bpmf_amd import _lib

PEAK = 157.3


def run(T, S, C, L, N, fused, reps=50):
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    d = torch.randn((S, C, N), device="cuda", generator=g)
    tp = torch.randn((T, S, C, L), device="cuda", generator=g)
    mv = torch.randint(0, 1500, (T, S, C), device="cuda", dtype=torch.int32, generator=g)
    w = torch.full((T, S, C), 1.0 / (S * C), device="cuda")
    _lib.set_option("mf.fused_prologue", fused)
    mf = sb.MatchedFilterGPU()
    mf.set_data(d)
    o = mf.run(tp, mv, w, 1)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            mf.run(tp, mv, w, 1, out=o)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    _lib.profile_enable(True)
    for _ in range(reps):
        mf.run(tp, mv, w, 1, out=o)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    kms = float(np.mean(_lib.profile_times_ms(_lib.KERNEL_MF_MAIN)))
    flop = 2.0 * L * S * C * T * (N - L + 1)
    print(f"T={T:4d} S={S} C={C} L={L} N={N} fused={fused}: kernel {kms * 1e3:7.1f} us ({flop / kms / 1e9 / PEAK:.3f} of peak), "
          f"call {best * 1e6:7.1f} us ({flop / best / 1e12 / PEAK:.3f} of peak)", flush=True)
    return o.clone()


if __name__ == "__main__":
    for T in (4, 16, 64):
        a = run(T, 8, 3, 128, 180_000, 0)
        b = run(T, 8, 3, 128, 180_000, 1)
        assert torch.equal(a, b)
    for L in (64, 256):
        a = run(4, 8, 3, L, 180_000, 0)
        b = run(4, 8, 3, L, 180_000, 1)
        assert torch.equal(a, b)
    _lib.set_option("mf.fused_prologue", 1)
