(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_compat.py -q -m gpu -n 4 -k "mf or MF" > gpurun_out/gpu_mf_norms.log 2>&1; tail -2 gpurun_out/gpu_mf_norms.log)
python tools/probe_mf_short.py 2>&1 | grep -v amdgpu
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np, seismic_bpmf_amd as sb
from seismic_bpmf_amd import _lib
for name, (T2, S2, C2, L2, N2) in (("configs0", (4, 8, 3, 128, 180_000)), ("T16_1h", (16, 8, 3, 128, 180_000)), ("T8_L64", (8, 8, 3, 64, 180_000))):
    g2 = torch.Generator(device="cuda"); g2.manual_seed(77)
    d2 = torch.randn((S2, C2, N2), device="cuda", generator=g2)
    t2 = torch.randn((T2, S2, C2, L2), device="cuda", generator=g2)
    m2 = torch.randint(0, 1500, (T2, S2, C2), device="cuda", dtype=torch.int32, generator=g2)
    w2 = torch.full((T2, S2, C2), 1.0 / (S2 * C2), device="cuda")
    mf2 = sb.MatchedFilterGPU(); mf2.set_data(d2)
    o2 = mf2.run(t2, m2, w2, 1); torch.cuda.synchronize()
    import time
    _lib.profile_enable(True); t0 = time.perf_counter()
    for _ in range(20): mf2.run(t2, m2, w2, 1, out=o2)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20; _lib.profile_enable(False)
    kms = float(np.mean(_lib.profile_times_ms(_lib.KERNEL_MF_MAIN)))
    flop = 2.0 * L2 * S2 * C2 * T2 * (N2 - L2 + 1)
    print(f"{name}: call {wall*1e3:.4f} ms kernel {kms:.4f} ms  kernel frac {flop/(kms*1e-3)/157.3e12:.3f}  call frac {flop/wall/157.3e12:.3f}")
PY
