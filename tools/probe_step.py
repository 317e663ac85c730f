import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seismic_bpmf_amd as sb
T, S, C, L, N = 64, 20, 3, 256, 8_640_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
data = torch.randn((S, C, N), device="cuda", generator=g)
tp = torch.randn((T, S, C, L), device="cuda", generator=g)
mv = torch.randint(0, 3000, (T, S, C), device="cuda", dtype=torch.int32, generator=g)
w = torch.full((T, S, C), 1.0 / (S * C), device="cuda")
m = sb.MatchedFilterGPU(); m.set_data(data)
for step, direct in [(1, False), (4, False), (4, True), (16, False), (16, True)]:
    m.run(tp, mv, w, step, force_direct=direct); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); m.run(tp, mv, w, step, force_direct=direct); b.record(); torch.cuda.synchronize()
    print(f"step={step} direct={direct}: {a.elapsed_time(b):.1f} ms")
