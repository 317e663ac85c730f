"""What a plain streaming read of a cfg2-sized CC matrix (500 x 8.64 M float32 = 17.3 GB) gets from HBM
on this box: torch's own reductions, for comparison with the detection-stage kernels (post.hip)."""
import time, torch
x = torch.randn((500, 8_639_745), device="cuda")
for name, f in (("sum", lambda: x.sum()), ("max", lambda: x.max()), ("row sum", lambda: x.sum(1)), ("abs().max(1)", lambda: x.abs().amax(1))):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {dt*1e3:.2f} ms  {x.numel()*4/dt/1e12:.2f} TB/s")
