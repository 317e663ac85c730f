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
    import ctypes
    raw = (ctypes.c_ulonglong * 32)()
    b.run(feat, wp)
    torch.cuda.synchronize()
    _lib.lib().bpmf_phase_read_bp(raw, 1)          # clear the warm-up launch
    _lib.profile_enable(True)
    beam, arg = b.run(feat, wp)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    ms = _lib.profile_times_ms(_lib.KERNEL_BP_BEAM)
    assert _lib.lib().bpmf_phase_read_bp(raw, 1) == 0
    v = np.array(list(raw), dtype=np.float64).reshape(4, 8)
    sa = float((geo["weights_sources"] != 0).sum(1).mean())
    tbs = 4 * sa * cfg["P"] * geo["moveouts"].shape[0] * cfg["N"] / (ms[0] * 1e-3) / 1e12
    print(f"{which}, {ncl} weighted stations: tile {info['class_tile']}, groups {info['class_groups']}, "
          f"{ms[0]:.2f} ms = {tbs / 157.3:.3f} of the ds_read_b64 rate (instrumented build)")
    print("  cycles per group entry and wave, mean over all waves of an age group (wave / 4 within the workgroup:")
    print("  0 = the oldest wave of every SIMD): " + " | ".join(NAMES) + " | sum")
    for a in range(4):
        n = max(1.0, v[a, 4])
        c = v[a, :4] / n
        print(f"  waves {4 * a:2d}-{4 * a + 3:2d}: " + " ".join(f"{x:9.0f}" for x in c) + f"   sum {c.sum():9.0f}   "
              f"({v[a, 4] / max(1.0, v[a, 5]):.0f} entries per wave)")
    ghz = v[:, :4].sum() / (256 * 16) / (ms[0] * 1e-3) / 1e9
    print(f"  sustained clock ~{ghz:.2f} GHz (all counted cycles / 4096 resident waves / kernel time; the edge launch and the tail excluded)", flush=True)
    b.close()


if __name__ == "__main__":
    args = sys.argv[1:] or ["cfg5_per_gpu", "40"]
    for i in range(0, len(args), 2):
        run(args[i], int(args[i + 1]))
