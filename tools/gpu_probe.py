"""Quick on-box timing probe (not the bench contract): python tools/gpu_probe.py [mf|bp] ..."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import seismic_bpmf_amd as sb
from seismic_bpmf_amd import synthetic as syn

def ev_time(fn, n=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 1e3)
    return min(ts), ts

def mf(T=50, S=20, C=3, L=256, N=8_640_000):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    data = torch.randn((S, C, N), device="cuda", generator=g)
    tp = torch.randn((T, S, C, L), device="cuda", generator=g)
    mv = torch.randint(0, 3000, (T, S, C), device="cuda", dtype=torch.int32, generator=g)
    w = torch.full((T, S, C), 1.0 / (S * C), device="cuda")
    m = sb.MatchedFilterGPU(); m.set_data(data)
    out = torch.empty((T, N - L + 1), device="cuda")
    t, ts = ev_time(lambda: m.run(tp, mv, w, 1, out=out))
    ncc = T * (N - L + 1)
    print(f"MF T={T} S={S} C={C} L={L} N={N}: {t:.4f}s all={['%.4f'%x for x in ts]} -> {ncc/t/1e6:.1f} M net-CC/s, "
          f"{ncc*S*C*2*L/t/1e12:.1f} TFLOP/s direct-form ({ncc*S*C*2*L/t/157.3e12*100:.1f}% of fp32 peak)")

def bp(grid=(50, 50, 20), S=20, C=3, P=2, N=4_320_000, sr=50.0):
    geo = syn.make_bp_geometry(grid, S, P, sr)
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    feat = torch.randn((S, C, N), device="cuda", generator=g).abs_()
    wp = syn.phase_weights(S, C, P)
    t0 = time.time(); b = sb.BeamformerGPU(geo["moveouts"], geo["weights_sources"]); tp_ = time.time() - t0
    K = geo["moveouts"].shape[0]
    t, ts = ev_time(lambda: b.run(feat, wp, "max", "strict"))
    sa = int((geo["weights_sources"] != 0).sum(1).mean())
    print(f"BP K={K} S={S} P={P} N={N} S_a={sa}: plan {tp_:.2f}s, {t:.4f}s all={['%.4f'%x for x in ts]} -> {K*N/t:.3e} gp*samples/s, "
          f"gather {4*sa*P*K*N/t/1e12:.1f} TB/s")

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "both"
    args = [int(x) for x in sys.argv[2:]]
    print(sb.device_info(0))
    if what in ("mf", "both"): mf(*args[:1]) if args else mf()
    if what in ("bp", "both"): bp()
