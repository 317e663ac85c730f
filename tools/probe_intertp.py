import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from seismic_bpmf_amd.workflow import intertemplate_cc
T = int(sys.argv[1]) if len(sys.argv) > 1 else 500
S, C, L, max_lag = 20, 3, 200, 10
rng = np.random.default_rng(0)
wf = rng.standard_normal((T, S, C, L)).astype(np.float32)
w1 = np.full((T, S, C), 1.0 / (S * C), np.float32)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = intertemplate_cc(wf, lambda t: w1, max_lag=max_lag)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"T={T}: intertemplate_cc {t1-t0:.3f}s ({T*T/(t1-t0)/1e6:.2f} M pairs/s), diag mean {out.diagonal().mean():.4f}")
