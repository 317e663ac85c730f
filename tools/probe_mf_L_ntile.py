import sys, os
sys.path.insert(0, os.getcwd())
import torch, seismic_bpmf_amd as sb
from seismic_bpmf_amd import _lib
T, S, C, N = 32, 20, 3, 8_640_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
data = torch.randn((S, C, N), device="cuda", generator=g)
m = sb.MatchedFilterGPU(); m.set_data(data)
for L in (64, 96, 128, 160, 192, 256):
    tp = torch.randn((T, S, C, L), device="cuda", generator=g)
    mv = torch.randint(0, 3000, (T, S, C), device="cuda", dtype=torch.int32, generator=g)
    w = torch.full((T, S, C), 1.0 / (S * C), device="cuda")
    out = torch.empty((T, N - L + 1), device="cuda")
    for nt in (4, 2, 1):
        with _lib.options(**{"mf.tiles_per_wave": nt}):
            m.run(tp, mv, w, 1, out=out); torch.cuda.synchronize()
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); m.run(tp, mv, w, 1, out=out); b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / 1e3
        fl = 2.0 * L * S * C * T * (N - L + 1)
        print(f"L={L} ntile={nt}: {t*1e3:.1f} ms  ({fl/t/157.3e12*100:.1f}%)")
