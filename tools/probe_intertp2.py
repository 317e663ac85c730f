"""Times the batched inter-template CC against the per-template loop (T templates, all pairs)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from seismic_bpmf_amd import workflow
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
S, C, L, max_lag = 10, 3, 200, 10
rng = np.random.default_rng(1)
wf = rng.standard_normal((T, S, C, L)).astype(np.float32)
base = np.full((T, S, C), 1.0 / (S * C), np.float32)
mask = np.ones((T, T), bool)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a = workflow.intertemplate_cc(wf, base, max_lag=max_lag, pair_mask=mask)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"batched T={T}: {t1 - t0:.3f} s", flush=True)
if T <= 500:
    full = base[:, None] * mask[:, :, None, None]
    t0 = time.perf_counter(); b = workflow.intertemplate_cc_loop(wf, full, max_lag=max_lag); t1 = time.perf_counter()
    print(f"loop    T={T}: {t1 - t0:.3f} s, identical: {np.array_equal(a, b)}")
