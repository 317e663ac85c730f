import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from seismic_bpmf_amd.threshold import ThresholdGPU
T, n = int(sys.argv[1]) if len(sys.argv) > 1 else 500, 8_639_745
cc = torch.randn((T, n), device="cuda") * 0.02
th = ThresholdGPU()
wn = np.random.default_rng(0).standard_normal(500).astype(np.float32)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tw, _ = th.time_dependent_threshold(cc, 180_000, 8.0, overlap=0.25, white_noise=wn)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    cand = th.extract_candidates(cc, tw, 180_000, overlap=0.25)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"T={T}: threshold {(t1-t0)*1e3:.2f} ms ({4*T*n*4/(t1-t0)/1e9:.0f} GB/s of 4 passes), candidates {(t2-t1)*1e3:.2f} ms ({cand.size} found)")
