"""MAD threshold on rows whose samples are correlated like a CC series (white noise through a k-tap moving
average): the window medians then scatter k-fold wider around the row's median than those of independent
samples, which is what the one-pass band has to hold.  Time per option stats.bucketed_median, and how many
windows the one-pass kernel left to the general one."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from seismic_bpmf_amd.threshold import ThresholdGPU
from seismic_bpmf_amd import _lib
rows, n, W = 500, 8_640_000 - 255, 180_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
wn = np.random.default_rng(0).standard_normal(10_000).astype(np.float32)
th = ThresholdGPU()
for taps in (1, 5, 25, 100):
    cc = torch.randn((rows, n + taps), device="cuda", generator=g)
    if taps > 1:
        cs = torch.cumsum(cc.double(), dim=1)
        cc = ((cs[:, taps:] - cs[:, :-taps]) / taps).float()[:, :n].contiguous()
        del cs
    else:
        cc = cc[:, :n].contiguous()
    cc *= 0.05
    # a noise level that changes over the day (x 1 .. x 3)
    cc *= torch.linspace(1.0, 3.0, n, device="cuda")[None, :]
    cc[:, -1500:] = 0.0
    for mode in (2, 1):
        _lib.set_option("stats.bucketed_median", mode)
        for ov in (0.25,):
            th.time_dependent_threshold_mad(cc, W, 8.0, overlap=ov, white_noise=wn, expand=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th.time_dependent_threshold_mad(cc, W, 8.0, overlap=ov, white_noise=wn, expand=False)
            torch.cuda.synchronize()
            print(f"taps {taps:3d} mode {mode} overlap {ov}: {(time.perf_counter() - t0) * 1e3:.1f} ms")
    del cc
    torch.cuda.empty_cache()
