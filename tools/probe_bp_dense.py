"""Backprojection with dense station weights (round 3): cfg3's grid with n_closest = 10 / 16 / 17 / 20
(the literal "x 20 stations" of BASELINE configs[2]) and one 17-station source among 10-station ones;
optionally the cfg5 per-GPU share with 40 weighted stations.  Prints the plan's classes, the time of
the beam kernels (HIP events inside the library) and the gathered TB/s against the ds_read_b64 rate."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import seismic_bpmf_amd as sb
from seismic_bpmf_amd import synthetic as syn, _lib

which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cases = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [10, 16, 17, 20]
cfg = syn.BP_CONFIGS[which]
slab = (0, 64) if cfg["grid"] == (125, 125, 8) else None
g = torch.Generator(device="cuda"); g.manual_seed(2)
feat = torch.randn((cfg["S"], cfg["C"], cfg["N"]), device="cuda", generator=g).abs_()
wp = syn.phase_weights(cfg["S"], cfg["C"], cfg["P"])
if os.environ.get("VERBOSE"):
    _lib.set_option("bp.verbose", 1)
for kv in os.environ.get("BP_OPTS", "").split(","):      # e.g. BP_OPTS=bp.fast_uniform=0,bp.halves=0
    if "=" in kv:
        _lib.set_option(kv.split("=")[0], int(kv.split("=")[1]))
if len(sys.argv) > 3:                       # force the tile of every class (256 on 33-64 stations: two residencies)
    _lib.set_option("bp.fast_tile", int(sys.argv[3]))


def run(label, mv, ws):
    import time
    t0 = time.time()
    b = sb.BeamformerGPU(mv, ws)
    t_plan = time.time() - t0
    b.run(feat, wp); torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(3):
        out = b.run(feat, wp)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    ms = float(np.mean(_lib.profile_times_ms(_lib.KERNEL_BP_BEAM)))
    sa = float((ws != 0).sum(1).mean())
    K = mv.shape[0]
    tbs = 4 * sa * cfg["P"] * K * cfg["N"] / (ms * 1e-3) / 1e12
    info = b.plan_info()
    print(f"{which} {label}: S_a={sa:.2f} {ms:.1f} ms  {tbs:.1f} TB/s = {tbs / 157.3:.3f} of the b64 rate; plan {t_plan:.2f} s "
          f"classes tile={info['class_tile']} sources={info['class_sources']} groups={info['class_groups']}", flush=True)
    b.close()
    return out


for ncl in cases:
    geo = syn.make_bp_geometry(cfg["grid"], cfg["S"], cfg["P"], cfg["sr"], n_closest=ncl, depth_slab=slab)
    run(f"n_closest={ncl}", geo["moveouts"], geo["weights_sources"])
if which == "cfg3" and len(sys.argv) <= 2:
    # exactly 10 stations everywhere, then ONE source with 17: the cliff of rounds 1-2
    geo = syn.make_bp_geometry(cfg["grid"], cfg["S"], cfg["P"], cfg["sr"], n_closest=10)
    first = geo["moveouts"][:, :, 0].astype(np.int64) * 64 + np.arange(cfg["S"])[None, :]
    order = np.argsort(first, axis=1)
    w = np.zeros_like(geo["weights_sources"])
    np.put_along_axis(w, order[:, :10], 0.1, axis=1)
    run("exactly 10", geo["moveouts"], w)
    w2 = w.copy()
    w2[12345, order[12345, :17]] = 0.1
    run("exactly 10 + one 17-station source", geo["moveouts"], w2)
