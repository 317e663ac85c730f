import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import seismic_bpmf_amd as sb
from seismic_bpmf_amd import synthetic as syn, _lib
which, ncl = sys.argv[1], int(sys.argv[2])
cfg = syn.BP_CONFIGS[which]
slab = (0, 64) if cfg["grid"] == (125, 125, 8) else None
g = torch.Generator(device="cuda"); g.manual_seed(2)
feat = torch.randn((cfg["S"], cfg["C"], cfg["N"]), device="cuda", generator=g).abs_()
wp = syn.phase_weights(cfg["S"], cfg["C"], cfg["P"])
geo = syn.make_bp_geometry(cfg["grid"], cfg["S"], cfg["P"], cfg["sr"], n_closest=ncl, depth_slab=slab)
b = sb.BeamformerGPU(geo["moveouts"], geo["weights_sources"])
info = b.plan_info()
beam, arg = b.run(feat, wp); torch.cuda.synchronize()
tile = info["class_tile"][0]
a = arg.cpu().numpy()
names = ["barrier1(wait for stragglers)", "desc read + DMA issue", "DMA latency + barrier2", "run/record loads", "gathers", "drain"]
for wv in range(16):
    v = a[3 * tile + 8 * wv: 3 * tile + 8 * wv + 7].astype(np.int64) * 64
    n = max(1, int(v[6] // 64))
    print(f"wave {wv:2d}:", " ".join(f"{float(x) / n:9.0f}" for x in v[:6]), " sum", round(float(v[:6].sum()) / n))
print(info["class_tile"], info["class_groups"])
