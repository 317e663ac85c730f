import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from seismic_bpmf_amd import features, _lib
g = torch.Generator(device="cuda"); g.manual_seed(3)
x = torch.randn((60, 4_320_000), device="cuda", generator=g)
x3 = x.reshape(20, 3, -1)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
env = features.envelope(x3)
rows = env.reshape(60, -1)
for mn in (131072, -1):
    _lib.set_option("stats.row_grid_min_n", mn)
    print(f"stats.row_grid_min_n {mn}: row_median_mad {timed(lambda: features.row_median_mad(rows, True)):.2f} ms, saturated_envelopes {timed(lambda: features.saturated_envelopes(x3)):.2f} ms, envelope {timed(lambda: features.envelope(x3)):.2f} ms")
