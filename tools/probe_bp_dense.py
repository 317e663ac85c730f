import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import seismic_bpmf_amd as sb
from seismic_bpmf_amd import synthetic as syn
ncl = int(sys.argv[1]) if len(sys.argv) > 1 else 20
geo = syn.make_bp_geometry((50, 50, 20), 20, 2, 50.0, n_closest=ncl)
N = 4_320_000
g = torch.Generator(device="cuda"); g.manual_seed(2)
feat = torch.randn((20, 3, N), device="cuda", generator=g).abs_()
wp = syn.phase_weights(20, 3, 2)
b = sb.BeamformerGPU(geo["moveouts"], geo["weights_sources"])
b.run(feat, wp); torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
a.record(); b.run(feat, wp); e.record(); torch.cuda.synchronize()
sa = float((geo["weights_sources"] != 0).sum(1).mean())
t = a.elapsed_time(e) / 1e3
print(f"n_closest={ncl} S_a={sa:.1f}: {t:.3f}s gather {4*sa*2*50000*N/t/1e12:.1f} TB/s")
