import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seismic_bpmf_amd as sb
T, S, C, N = 32, 20, 3, 8_640_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
data = torch.randn((S, C, N), device="cuda", generator=g)
m = sb.MatchedFilterGPU(); m.set_data(data)
for L in [int(x) for x in sys.argv[1:]] or [128, 256, 400, 512, 800, 1024, 2000]:
    tp = torch.randn((T, S, C, L), device="cuda", generator=g)
    mv = torch.randint(0, 3000, (T, S, C), device="cuda", dtype=torch.int32, generator=g)
    w = torch.full((T, S, C), 1.0 / (S * C), device="cuda")
    out = torch.empty((T, N - L + 1), device="cuda")
    m.run(tp, mv, w, 1, out=out); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); m.run(tp, mv, w, 1, out=out); b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 1e3
    fl = 2.0 * L * S * C * T * (N - L + 1)
    print(f"L={L}: {t*1e3:.1f} ms  {fl/t/1e12:.1f} TFLOP/s direct-form ({fl/t/157.3e12*100:.1f}%)")
