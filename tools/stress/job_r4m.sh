python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np, seismic_bpmf_amd as sb
from seismic_bpmf_amd import _lib
for name, (T2, S2, C2, L2, N2) in (("configs0", (4, 8, 3, 128, 180_000)), ("T16_1h", (16, 8, 3, 128, 180_000))):
    g2 = torch.Generator(device="cuda"); g2.manual_seed(77)
    d2 = torch.randn((S2, C2, N2), device="cuda", generator=g2)
    t2 = torch.randn((T2, S2, C2, L2), device="cuda", generator=g2)
    m2 = torch.randint(0, 1500, (T2, S2, C2), device="cuda", dtype=torch.int32, generator=g2)
    w2 = torch.full((T2, S2, C2), 1.0 / (S2 * C2), device="cuda")
    mf2 = sb.MatchedFilterGPU(); mf2.set_data(d2)
    ref = mf2.run(t2, m2, w2, 1).clone(); torch.cuda.synchronize()
    for ntile in (0, 1, 2):
      for stag in (0, 4, 8, 16, 32, 64):
        _lib.set_option("mf.stagger", stag); _lib.set_option("mf.tiles_per_wave", ntile)
        o2 = mf2.run(t2, m2, w2, 1); torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(20): mf2.run(t2, m2, w2, 1, out=o2)
        torch.cuda.synchronize(); _lib.profile_enable(False)
        kms = float(np.mean(_lib.profile_times_ms(_lib.KERNEL_MF_MAIN)))
        flop = 2.0 * L2 * S2 * C2 * T2 * (N2 - L2 + 1)
        print(f"{name} ntile={ntile} stagger={stag}: kernel {kms*1e3:.1f} us  frac {flop/(kms*1e-3)/157.3e12:.3f} same={bool(torch.equal(o2, ref))}", flush=True)
_lib.set_option("mf.stagger", 0); _lib.set_option("mf.tiles_per_wave", 0)
PY
