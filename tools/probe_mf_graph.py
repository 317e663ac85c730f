"""BASELINE configs[0] through the resident handle: plain calls vs one hipGraph replay of the same call
(torch.cuda.CUDAGraph around MatchedFilterGPU.run: the three launches of bpmf_mf_run_dev are captured
from the stream they are enqueued on)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from seismic_bpmf_amd import MatchedFilterGPU, synthetic as syn
c = syn.MF_CONFIGS["cfg1"]
m = syn.make_mf_inputs(c["T"], c["S"], c["C"], c["L"], c["N"], seed=1)
mf = MatchedFilterGPU()
mf.set_data(m["data"])
tp = torch.as_tensor(m["templates"], device="cuda"); mv = torch.as_tensor(m["moveouts"], device="cuda"); w = torch.as_tensor(m["weights"], device="cuda")
n_corr = c["N"] - c["L"] + 1
out = torch.empty((c["T"], n_corr), device="cuda")
mf.run(tp, mv, w, 1, out=out); torch.cuda.synchronize()
ref = out.clone()
def timeit(f, n=200):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
t_plain = timeit(lambda: mf.run(tp, mv, w, 1, out=out))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    mf.run(tp, mv, w, 1, out=out)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        mf.run(tp, mv, w, 1, out=out)
out.zero_()
g.replay(); torch.cuda.synchronize()
print("graph result identical:", bool(torch.equal(out, ref)))
t_graph = timeit(lambda: g.replay())
flop = 2.0 * c["L"] * c["S"] * c["C"] * c["T"] * n_corr
print(f"plain call {t_plain*1e3:.4f} ms ({flop/t_plain/157.3e12*100:.1f} % of peak)   graph replay {t_graph*1e3:.4f} ms ({flop/t_graph/157.3e12*100:.1f} %)")
