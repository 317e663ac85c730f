"""Row median / MAD when rows CANNOT take the two-read path (two-valued rows: their middle spans the whole range):
the radix select by the whole chip that serves them (round 5) against the one-workgroup kernel they used to go to
(option stats.row_grid_min_n -1 sends every row there)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from seismic_bpmf_amd import features, _lib
g = torch.Generator(device="cuda"); g.manual_seed(1)
def t_of(x, skip=True):
    features.row_median_mad(x, skip); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); r = features.row_median_mad(x, skip); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best, r
for rows, n in ((500, 8_639_745), (60, 4_320_000)):
    x = torch.randn((rows, n), device="cuda", generator=g) * 0.05
    base, _ = t_of(x)
    two = torch.where(torch.rand((n,), device="cuda", generator=g) < 0.5, -0.5, 0.75)
    for k in (1, rows // 10, rows):
        y = x.clone(); y[:k] = two[None, :]
        t_new, r_new = t_of(y)
        _lib.set_option("stats.row_grid_min_n", -1)
        t_old, r_old = t_of(y[:k].contiguous())
        _lib.set_option("stats.row_grid_min_n", 131072)
        same = all(torch.equal(a[:k], b) for a, b in zip(r_new, r_old))
        print(f"{rows} rows x {n}: all normal {base:.2f} ms; {k} two-valued row(s) among them {t_new:.2f} ms "
              f"(those {k} alone through the one-workgroup kernel: {t_old:.2f} ms; same bits: {same})")
    del x, y
