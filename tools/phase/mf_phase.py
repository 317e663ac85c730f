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


def run(L, T=32, S=20, C=3, N=8_640_000):
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
    _lib.profile_enable(True)
    cc = mf.run(tmpl, mv, w, 1, out=cc)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    ms = _lib.profile_times_ms(_lib.KERNEL_MF_MAIN)[0]
    row = cc[0, 4096 * 3: 4096 * 3 + 64].cpu().numpy()
    per_wave = []
    for wv in range(4):
        v = row[8 * wv: 8 * wv + 5].astype(np.float64)
        per_wave.append(v[:4] / max(1.0, v[4]))
    c = np.mean(per_wave, axis=0)
    # sustained shader clock: a wave's cycles per channel x channels x rounds of workgroups = kernel time
    n_wg = T * ((N - L + 1 + 4095) // 4096)
    rounds = n_wg / (256 * 4)                     # 4 workgroups (16 waves) per CU
    ghz = rounds * S * C * c.sum() / (ms * 1e-3) / 1e9
    flop = 2.0 * L * S * C * T * (N - L + 1)
    print(f"L = {L:4d}: " + "  ".join(f"{n} {x:8.0f}" for n, x in zip(NAMES, c)) +
          f"   total {c.sum():8.0f} cycles per channel and wave; kernel {ms:.1f} ms "
          f"({flop / (ms * 1e-3) / 1e12 / 157.3 * 100:.1f} % of the fp32-MFMA peak, instrumented build); "
          f"sustained clock {ghz:.2f} GHz")


if __name__ == "__main__":
    for L in [int(x) for x in sys.argv[1:]] or [64, 128, 192, 256]:
        run(L)
