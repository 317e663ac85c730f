"""Option mf.split16 against the exact kernels, resident engine, a day of data:  python tools/probe_mf_split16.py [L ...]
Kernel milliseconds from the library's events, max |d cc| / sum|w| over the whole matrix."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import seismic_bpmf_amd as sb
from seismic_bpmf_amd import _lib

S, C, N = 20, 3, 8_640_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
data = torch.randn((S, C, N), device="cuda", generator=g)
for L in [int(x) for x in sys.argv[1:]] or [128, 256, 376, 400, 800]:
    T = max(8, int(64 * 256 / L))
    tp = torch.randn((T, S, C, L), device="cuda", generator=g)
    mv = torch.randint(0, 3000, (T, S, C), device="cuda", dtype=torch.int32, generator=g)
    w = torch.full((T, S, C), 1.0 / (S * C), device="cuda")
    res = {}
    for opt in (0, 1):
        _lib.set_option("mf.split16", opt)
        m = sb.MatchedFilterGPU(); m.set_data(data)
        out = m.run(tp, mv, w, 1)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(2):
            m.run(tp, mv, w, 1, out=out)
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        res[opt] = (float(np.mean(_lib.profile_times_ms(_lib.KERNEL_MF_MAIN))), out.clone())
        del m
    _lib.set_option("mf.split16", 0)
    diff = float((res[1][1] - res[0][1]).abs().max().item())
    flop = 2.0 * L * S * C * T * (N - L + 1)
    print(f"L {L:5d} T {T:3d}: exact {res[0][0]:8.2f} ms ({flop / res[0][0] / 1e9:6.1f} TF)  split16 {res[1][0]:8.2f} ms "
          f"({flop / res[1][0] / 1e9:6.1f} TF)  x {res[0][0] / res[1][0]:.2f}   max |d cc| {diff:.2e}", flush=True)
    del res, tp
    torch.cuda.empty_cache()
