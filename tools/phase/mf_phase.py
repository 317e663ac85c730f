"""Cycles per channel and wave of mf_mfma_wave_kernel, by phase (s_memtime inside the kernel; the
library of tools/phase/build_phase_lib.py).  Regenerates profiles/rNN_mf_phase_cycles.txt:

    python tools/phase/build_phase_lib.py && python tools/phase/mf_phase.py [L ...]
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

NAMES = ["staging writes", "norm loads + staging issue", "K loop", "epilogue"]


def run(L, T=32, S=20, C=3, N=8_640_000, small=False):
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    data = torch.randn((S, C, N), device="cuda", generator=g)
    tmpl = torch.randn((T, S, C, L), device="cuda", generator=g)
    mv = torch.randint(0, 1500, (T, S, C), device="cuda", dtype=torch.int32, generator=g)
    w = torch.full((T, S, C), 1.0 / (S * C), device="cuda")
    mf = sb.MatchedFilterGPU()
    mf.set_data(data)
    cc = mf.run(tmpl, mv, w, 1)
    torch.cuda.synchronize()
    import ctypes
    _lib.lib().bpmf_phase_read_mf((ctypes.c_ulonglong * 8)(), 1)      # clear the warm-up launch
    _lib.profile_enable(True)
    cc = mf.run(tmpl, mv, w, 1, out=cc)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    ms = _lib.profile_times_ms(_lib.KERNEL_MF_MAIN)[0]
    import ctypes
    raw = (ctypes.c_ulonglong * 8)()
    assert _lib.lib().bpmf_phase_read_mf(raw, 1) == 0
    v = np.array(list(raw), dtype=np.float64)
    c = v[:4] / max(1.0, v[4])                    # cycles per channel and wave, mean over every wave of the launch
    # sustained shader clock: the cycles all waves counted / (waves resident at a time x kernel time);
    # 16 waves per CU are resident (128 VGPRs), the tail of the launch makes this a slight underestimate
    if small:           # fewer than 4 tiles per wave, not every SIMD full: the cycle counts only
        print(f"T = {T}, {S} x {C} channels, L = {L}, N = {N}: " + "  ".join(f"{n} {x:8.0f}" for n, x in zip(NAMES, c)) +
              f"   total {c.sum():8.0f} cycles per channel and wave; kernel {ms * 1e3:.1f} us (instrumented build); "
              f"{v[5]:.0f} waves counted, {v[4] / max(1.0, v[5]):.1f} channels each", flush=True)
        return
    ghz = v[:4].sum() / (256 * 16) / (ms * 1e-3) / 1e9
    flop = 2.0 * L * S * C * T * (N - L + 1)
    kpad = (L + 15 + 15) // 16 * 16
    pipe = 4 * (kpad // 16) * 16 * 32             # cycles the matrix pipe of a SIMD needs per channel for its 4 waves
    print(f"L = {L:4d}: " + "  ".join(f"{n} {x:8.0f}" for n, x in zip(NAMES, c)) +
          f"   total {c.sum():8.0f} cycles per channel and wave (matrix pipe of the SIMD: {pipe} for its 4 waves = "
          f"{pipe / c.sum() * 100:.0f} % busy); kernel {ms:.1f} ms "
          f"({flop / (ms * 1e-3) / 1e12 / 157.3 * 100:.1f} % of the fp32-MFMA peak, instrumented build); "
          f"sustained clock ~{ghz:.2f} GHz (both derived figures assume 4 tiles per wave and 16 resident waves per CU for "
          f"the whole launch); {v[5]:.0f} waves counted, {v[4] / max(1.0, v[5]):.1f} channels each", flush=True)


if __name__ == "__main__":
    for arg in sys.argv[1:] or ["64", "128", "192", "256"]:
        if arg == "configs0":                     # BASELINE configs[0]: 1 h @ 50 Hz, 4 templates, 8 x 3 channels
            run(128, T=4, S=8, C=3, N=180_000, small=True)
        else:
            run(int(arg))
