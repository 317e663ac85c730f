import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seismic_bpmf_amd.decimate import find_similar_sources
from oracle import oracle
K = int(sys.argv[1]) if len(sys.argv) > 1 else 35490
S = 8
rng = np.random.default_rng(0)
n = int(round(K ** (1 / 3))) + 1
g = np.stack(np.meshgrid(np.linspace(0, 60, n), np.linspace(0, 60, n), np.linspace(0, 20, n), indexing="ij"), -1).reshape(-1, 3)[:K]
sta = np.c_[rng.uniform(0, 60, S), rng.uniform(0, 60, S), np.zeros(S)]
mv = (np.linalg.norm(g[:, None] - sta[None], axis=2) / 6.0).astype(np.float32)
mv -= mv.min(axis=1, keepdims=True)
lon, lat = g[:, 0].astype(np.float32), g[:, 1].astype(np.float32)
cl = np.linspace(-1, 61, 6).astype(np.float32)
for method in ("closest", "smallest"):
    t0 = time.perf_counter(); red = find_similar_sources(mv, lon, lat, cl, cl, 0.05, num_stations_for_diff=S, method=method); t1 = time.perf_counter()
    msg = f"K={K} {method}: GPU {t1-t0:.2f}s kept {K-red.sum()}"
    if K <= 40000:
        t0 = time.perf_counter(); want = oracle.find_similar_sources(mv, lon, lat, cl, cl, 0.05, S, method); t2 = time.perf_counter()
        msg += f" | CPU oracle (1 thread) {t2-t0:.2f}s equal={np.array_equal(red, want)}"
    print(msg)
