"""Cycles per group entry and wave of the interior BP kernel, by phase (s_memtime inside the kernel;
the library of tools/phase/build_phase_lib.py).  Regenerates profiles/rNN_bp_fast_phase_cycles.txt:

    python tools/phase/build_phase_lib.py && python tools/phase/bp_phase.py cfg5_per_gpu 40 [cfg3 10 ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from seismic_bpmf_amd import _lib  # noqa: E402

_lib.LIBPATH = os.path.join(ROOT, "tools", "phase", "libbpmf_hip_phase.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import seismic_bpmf_amd as sb  # noqa: E402
from seismic_bpmf_amd import synthetic as syn  # noqa: E402

NAMES = ["barrier wait", "desc read + copy issue", "copy latency + barrier 2", "records + gathers"]


def run(which, ncl):
    cfg = syn.BP_CONFIGS[which]
    slab = (0, 64) if cfg["grid"] == (125, 125, 8) else None
    g = torch.Generator(device="cuda")
    g.manual_seed(2)
    feat = torch.randn((cfg["S"], cfg["C"], cfg["N"]), device="cuda", generator=g).abs_()
    wp = syn.phase_weights(cfg["S"], cfg["C"], cfg["P"])
    geo = syn.make_bp_geometry(cfg["grid"], cfg["S"], cfg["P"], cfg["sr"], n_closest=ncl, depth_slab=slab)
    b = sb.BeamformerGPU(geo["moveouts"], geo["weights_sources"])
    info = b.plan_info()
    b.run(feat, wp)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    beam, arg = b.run(feat, wp)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    ms = _lib.profile_times_ms(_lib.KERNEL_BP_BEAM)
    tile = info["class_tile"][0]
    a = arg.cpu().numpy()
    print(f"{which}, {ncl} weighted stations: tile {tile}, classes {info['n_classes']}, groups {info['class_groups']}, "
          f"{ms[0]:.2f} ms (instrumented build)")
    print("  cycles per entry and wave: " + " | ".join(NAMES) + " | sum")
    for wv in range(16):
        v = a[3 * tile + 8 * wv: 3 * tile + 8 * wv + 5].astype(np.int64)
        n = max(1, int(v[4]))
        c = v[:4] * 64.0 / n
        print(f"  wave {wv:2d}: " + " ".join(f"{x:9.0f}" for x in c) + f"   sum {c.sum():9.0f}   ({n} entries)")
    b.close()


if __name__ == "__main__":
    args = sys.argv[1:] or ["cfg5_per_gpu", "40"]
    for i in range(0, len(args), 2):
        run(args[i], int(args[i + 1]))
