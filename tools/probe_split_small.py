"""mf.split16 on small launches: kernel time of the split-precision matched filter beside the exact kernel from
configs[0] (4 templates x one hour: 88 workgroups of 8192 lags) upwards -- where does the split kernel start to win?
(mf.split16 = 1 hands launches below the crossover to the exact kernel; = 2 forces the split kernel.)
Usage (GPU box): python tools/probe_split_small.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import seismic_bpmf_amd as sb
from seismic_bpmf_amd import _lib

FORCE = int(os.environ.get("BPMF_SPLIT_FORCE", "2"))
shapes = [(f"configs0 x {k}", 4 * k, 8, 3, 128, 180_000) for k in (1, 2, 3, 4, 6, 8, 16)]
shapes += [(f"{T} tmpl, 20x3, L=256, 1 h @ 100 Hz", T, 20, 3, 256, 360_000) for T in (1, 2, 4, 8, 16)]
shapes += [("tutorial", 10, 8, 3, 200, 2_160_000), ("1 tmpl day", 1, 20, 3, 256, 8_640_000)]
for name, T, S, C, L, N in shapes:
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    d = torch.randn((S, C, N), device="cuda", generator=g)
    t = torch.randn((T, S, C, L), device="cuda", generator=g)
    m = torch.randint(0, 1500, (T, S, C), device="cuda", dtype=torch.int32, generator=g)
    w = torch.full((T, S, C), 1.0 / (S * C), device="cuda")
    ms = []
    for split in (0, FORCE):
        _lib.set_option("mf.split16", split)
        mf = sb.MatchedFilterGPU(device=0); mf.set_data(d)
        o = mf.run(t, m, w, 1); torch.cuda.synchronize()
        reps = 50 if N < 1_000_000 else 5
        _lib.profile_enable(True)
        for _ in range(reps): mf.run(t, m, w, 1, out=o)
        torch.cuda.synchronize(); _lib.profile_enable(False)
        ms.append(float(np.mean(_lib.profile_times_ms(_lib.KERNEL_MF_MAIN))))
        del mf, o
    nb = -(-(N - L + 1) // 8192)
    print(f"{name:36s} T x blocks(8192 lags) = {T * nb:5d}: exact {ms[0]:8.4f} ms, split {ms[1]:8.4f} ms, x {ms[0] / ms[1]:.2f}", flush=True)
_lib.set_option("mf.split16", 0)
