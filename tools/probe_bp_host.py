import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import seismic_bpmf_amd as sb
from seismic_bpmf_amd import synthetic as syn
geo = syn.make_bp_geometry((125, 125, 64), 40, 2, 100.0, n_closest=10)
N = 200_000
feat = np.abs(np.random.default_rng(0).standard_normal((40, 3, N))).astype(np.float32)
wp = syn.phase_weights(40, 3, 2)
for rep in range(3):
    t0 = time.perf_counter(); b, a = sb.beamform(feat, geo["moveouts"], wp, geo["weights_sources"], device="gpu"); t1 = time.perf_counter()
    print(f"beamform() K=1M N={N}: {t1-t0:.3f}s (max {b.max():.3f})")
