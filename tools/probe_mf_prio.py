"""MF L <= 257 kernel: option mf.boundary_prio (s_setprio of a wave outside its K loop) x template length."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seismic_bpmf_amd as sb
from seismic_bpmf_amd import _lib
T, S, C, N = 32, 20, 3, 8_640_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
data = torch.randn((S, C, N), device="cuda", generator=g)
m = sb.MatchedFilterGPU(); m.set_data(data)
for L in [int(x) for x in sys.argv[1:]] or [64, 128, 192, 256]:
    tp = torch.randn((T, S, C, L), device="cuda", generator=g)
    mv = torch.randint(0, 3000, (T, S, C), device="cuda", dtype=torch.int32, generator=g)
    w = torch.full((T, S, C), 1.0 / (S * C), device="cuda")
    out = torch.empty((T, N - L + 1), device="cuda")
    ref = None
    for rep in range(2):
        for prio in (0, 1, 2, 3):
            _lib.set_option("mf.boundary_prio", prio)
            m.run(tp, mv, w, 1, out=out); torch.cuda.synchronize()
            _lib.profile_enable(True)
            m.run(tp, mv, w, 1, out=out); torch.cuda.synchronize()
            _lib.profile_enable(False)
            t = _lib.profile_times_ms(_lib.KERNEL_MF_MAIN)[0] / 1e3
            fl = 2.0 * L * S * C * T * (N - L + 1)
            same = True if ref is None else bool(torch.equal(ref, out))
            if ref is None:
                ref = out.clone()
            print(f"L={L} prio={prio}: {t*1e3:.2f} ms  {fl/t/157.3e12*100:.1f}% of peak  identical={same}", flush=True)
_lib.set_option("mf.boundary_prio", 0)
