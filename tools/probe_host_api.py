"""End-to-end time of the NumPy-in / NumPy-out call surfaces (H2D + kernels + D2H)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import seismic_bpmf_amd as sb
from seismic_bpmf_amd import synthetic as syn
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
inp = syn.make_mf_inputs(T, 20, 3, 256, 8_640_000, seed=3, n_events=0)
for rep in range(3):
    t0 = time.perf_counter()
    cc = sb.matched_filter(inp["templates"], inp["moveouts"], inp["weights"], inp["data"], 1, arch="gpu", check_zeros=False)
    t1 = time.perf_counter()
    print(f"matched_filter T={T}: {t1-t0:.3f}s end to end ({T*cc.shape[1]/(t1-t0)/1e6:.0f} M CC/s), out {cc.nbytes/1e9:.2f} GB")
# raw copies
x = torch.empty(int(2e9)//4, dtype=torch.float32, device="cuda")
h = np.empty(int(2e9)//4, np.float32); hp = torch.empty(int(2e9)//4, dtype=torch.float32).pin_memory()
for name, dst in (("pageable", torch.from_numpy(h)), ("pinned", hp)):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dst.copy_(x); torch.cuda.synchronize(); t1 = time.perf_counter()
    t2 = time.perf_counter(); x.copy_(dst); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"{name}: D2H {2/(t1-t0):.1f} GB/s, H2D {2/(t3-t2):.1f} GB/s")
t0 = time.perf_counter(); hp2 = torch.empty(int(4e9)//4, dtype=torch.float32).pin_memory(); t1 = time.perf_counter()
print(f"pinning 4 GB: {t1-t0:.2f}s")
