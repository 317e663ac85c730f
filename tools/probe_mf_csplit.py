"""MF on an hour-long series (BASELINE configs[0]'s shape, T templates): the channel-split variant of the one-tile
kernel (option mf.channel_split: four waves per 256 lags, every fourth used channel each) against the plain one;
kernel time from the library's events and whole resident call from the host clock.  Also checks the bits."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, seismic_bpmf_amd as sb
from seismic_bpmf_amd import _lib
S2, C2, N2 = 8, 3, 180_000
for L2 in (64, 128, 200, 256):
    for T2 in (1, 2, 4, 8, 16):
        g2 = torch.Generator(device="cuda"); g2.manual_seed(77)
        d2 = torch.randn((S2, C2, N2), device="cuda", generator=g2)
        t2 = torch.randn((T2, S2, C2, L2), device="cuda", generator=g2)
        m2 = torch.randint(-200, 1500, (T2, S2, C2), device="cuda", dtype=torch.int32, generator=g2)
        w2 = torch.full((T2, S2, C2), 1.0 / (S2 * C2), device="cuda")
        w2[0, 1] = 0.0
        mf2 = sb.MatchedFilterGPU(); mf2.set_data(d2)
        res, outs = {}, {}
        for rep in range(2):
            for cs in (0, 1 << 20):
                _lib.set_option("mf.channel_split", cs)
                o2 = mf2.run(t2, m2, w2, 1); torch.cuda.synchronize()
                outs[cs] = o2.clone()
                wall = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _ in range(50): mf2.run(t2, m2, w2, 1, out=o2)
                    torch.cuda.synchronize()
                    wall = min(wall, (time.perf_counter() - t0) / 50)
                _lib.profile_enable(True)
                for _ in range(20): mf2.run(t2, m2, w2, 1, out=o2)
                torch.cuda.synchronize(); _lib.profile_enable(False)
                res[cs] = (float(np.mean(_lib.profile_times_ms(_lib.KERNEL_MF_MAIN))) * 1e3, wall * 1e6)
        same = torch.equal(outs[0], outs[1 << 20])
        flop = 2.0 * L2 * S2 * C2 * T2 * (N2 - L2 + 1)
        print(f"L={L2} T={T2:3d}: " + "  ".join(f"{'split' if k else 'plain'}: kernel {v[0]:6.1f} us ({flop / (v[0] * 1e-6) / 157.3e12:.3f}) call {v[1]:6.1f} us ({flop / (v[1] * 1e-6) / 157.3e12:.3f})" for k, v in res.items()) + f"  same bits: {same}", flush=True)
_lib.set_option("mf.channel_split", 4096)
