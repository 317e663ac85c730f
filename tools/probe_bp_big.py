import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import seismic_bpmf_amd as sb
from seismic_bpmf_amd import synthetic as syn
t0 = time.time(); geo = syn.make_bp_geometry((125, 125, 64), 40, 2, 100.0, n_closest=10); t1 = time.time()
K = geo["moveouts"].shape[0]
print(f"geometry K={K}: {t1-t0:.1f}s, max moveout {geo['moveouts'].max()}")
from seismic_bpmf_amd import _lib; _lib.set_option("bp.verbose", 1)
t0 = time.time(); b = sb.BeamformerGPU(geo["moveouts"], geo["weights_sources"]); t1 = time.time()
print(f"plan: {t1-t0:.1f}s")
N = 500_000
feat = torch.randn((40, 3, N), device="cuda").abs_()
wp = syn.phase_weights(40, 3, 2)
b.run(feat, wp); torch.cuda.synchronize()
t0 = time.time(); beam, arg = b.run(feat, wp); torch.cuda.synchronize(); t = time.time() - t0
print(f"run N={N}: {t:.3f}s -> {K*N/t:.3e} gp*samples/s; argmax range {int(arg.min())}..{int(arg.max())}")
