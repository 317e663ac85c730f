"""MF on an hour-long series (BASELINE configs[0]'s shape, T templates): tiles per wave of the L <= 257 kernel
(option mf.tiles_per_wave) x number of templates -- calibrates the automatic choice (python tools/probe_mf_ntile_T.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, seismic_bpmf_amd as sb
from seismic_bpmf_amd import _lib
S2, C2, N2 = 8, 3, 180_000
for L2 in (128, 256):
    for T2 in (4, 8, 16, 32, 64, 128, 256):
        g2 = torch.Generator(device="cuda"); g2.manual_seed(77)
        d2 = torch.randn((S2, C2, N2), device="cuda", generator=g2)
        t2 = torch.randn((T2, S2, C2, L2), device="cuda", generator=g2)
        m2 = torch.randint(0, 1500, (T2, S2, C2), device="cuda", dtype=torch.int32, generator=g2)
        w2 = torch.full((T2, S2, C2), 1.0 / (S2 * C2), device="cuda")
        mf2 = sb.MatchedFilterGPU(); mf2.set_data(d2)
        o2 = mf2.run(t2, m2, w2, 1); torch.cuda.synchronize()
        res = {}
        for rep in range(2):
            for ntile in (0, 1, 2, 4):
                _lib.set_option("mf.tiles_per_wave", ntile)
                mf2.run(t2, m2, w2, 1, out=o2); torch.cuda.synchronize()
                _lib.profile_enable(True)
                for _ in range(10): mf2.run(t2, m2, w2, 1, out=o2)
                torch.cuda.synchronize(); _lib.profile_enable(False)
                res[ntile] = float(np.mean(_lib.profile_times_ms(_lib.KERNEL_MF_MAIN))) * 1e3
        waves4 = T2 * ((N2 - L2 + 1 + 4095) // 4096) * 4
        flop = 2.0 * L2 * S2 * C2 * T2 * (N2 - L2 + 1)
        print(f"L={L2} T={T2:3d} (waves at 4 tiles: {waves4:5d}): " + "  ".join(f"ntile {k}: {v:7.1f} us ({flop / (v * 1e-6) / 157.3e12:.3f})" for k, v in res.items()), flush=True)
_lib.set_option("mf.tiles_per_wave", 0)
