"""rocprofv3 target: a few MF launches at a reduced template count (python tools/prof_mf.py T [split16])."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import seismic_bpmf_amd as sb
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S, C, L, N = 20, 3, 256, 8_640_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
data = torch.randn((S, C, N), device="cuda", generator=g)
tp = torch.randn((T, S, C, L), device="cuda", generator=g)
mv = torch.randint(0, 3000, (T, S, C), device="cuda", dtype=torch.int32, generator=g)
w = torch.full((T, S, C), 1.0 / (S * C), device="cuda")
if len(sys.argv) > 2 and sys.argv[2] == "split16":
    sb.set_option("mf.split16", 1)
m = sb.MatchedFilterGPU(); m.set_data(data)
out = torch.empty((T, N - L + 1), device="cuda")
for _ in range(2):
    m._prepared_for = None
    m.run(tp, mv, w, 1, out=out)
torch.cuda.synchronize()
print("done")
