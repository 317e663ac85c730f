"""rocprofv3 target: a few BP launches on BASELINE configs[2] (python tools/prof_bp.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import seismic_bpmf_amd as sb
from seismic_bpmf_amd import synthetic as syn
cfg = syn.BP_CONFIGS["cfg3"]
geo = syn.make_bp_geometry(cfg["grid"], cfg["S"], cfg["P"], cfg["sr"])
g = torch.Generator(device="cuda"); g.manual_seed(2)
feat = torch.randn((cfg["S"], cfg["C"], cfg["N"]), device="cuda", generator=g).abs_()
wp = syn.phase_weights(cfg["S"], cfg["C"], cfg["P"])
b = sb.BeamformerGPU(geo["moveouts"], geo["weights_sources"])
for _ in range(2):
    b.run(feat, wp, "max", "strict")
torch.cuda.synchronize()
print("done")
