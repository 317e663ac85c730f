import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from seismic_bpmf_amd import MatchedFilterGPU, synthetic as syn
c = syn.MF_CONFIGS["cfg1"]
m = syn.make_mf_inputs(c["T"], c["S"], c["C"], c["L"], c["N"], seed=1)
mf = MatchedFilterGPU()
t0 = time.perf_counter(); mf.set_data(m["data"]); torch.cuda.synchronize(); print("set_data", round((time.perf_counter()-t0)*1e3, 3), "ms")
tp = torch.as_tensor(m["templates"], device="cuda"); mv = torch.as_tensor(m["moveouts"], device="cuda"); w = torch.as_tensor(m["weights"], device="cuda")
cc = mf.run(tp, mv, w, 1); torch.cuda.synchronize()
ts = []
for _ in range(20):
    t0 = time.perf_counter(); cc = mf.run(tp, mv, w, 1); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = min(ts); n = cc.numel()
print(f"cfg1 run: {t*1e3:.3f} ms, {n / t / 1e6:.1f} M CC-samples/s, {2*c['L']*c['S']*c['C']*n/t/1e12:.2f} TFLOP/s")
for T in (16, 64, 256):
    tpT = tp.repeat(T // 4, 1, 1, 1); mvT = mv.repeat(T // 4, 1, 1); wT = w.repeat(T // 4, 1, 1)
    cc = mf.run(tpT, mvT, wT, 1); torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); cc = mf.run(tpT, mvT, wT, 1); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = min(ts); n = cc.numel()
    print(f"T={T}: {t*1e3:.3f} ms, {2*c['L']*c['S']*c['C']*n/t/1e12:.2f} TFLOP/s")
