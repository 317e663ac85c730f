"""Where does the device idle inside one host-pointer matched-filter call at cfg2 size?  Run under
   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/probe_e2e_gaps.py
then  python tools/probe_e2e_gaps.py --analyse <dir>  lists the gaps between consecutive kernels of the last call."""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
    import csv
    files = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    rows = [r for r in rows if "mf_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last call = everything after the last gap of more than 200 ms
    ts = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in rows]
    cut = 0
    for i in range(1, len(ts)):
        if ts[i][0] - ts[i - 1][1] > 200e6:
            cut = i
    ts = ts[cut:]
    busy = sum(e - s for s, e, _ in ts)
    span = ts[-1][1] - ts[0][0]
    print(f"last call: {len(ts)} kernels, span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, idle {(span - busy) / 1e6:.1f} ms")
    for i in range(1, len(ts)):
        gap = ts[i][0] - ts[i - 1][1]
        if gap > 0.3e6:
            print(f"  gap {gap / 1e6:6.2f} ms at +{(ts[i][0] - ts[0][0]) / 1e6:7.1f} ms before {ts[i][2]} (after {ts[i - 1][2]}, {(ts[i - 1][1] - ts[i - 1][0]) / 1e6:.1f} ms)")
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import seismic_bpmf_amd as sb  # noqa: E402

T, S, C, L, N = 500, 20, 3, 256, 8_640_000
g = torch.Generator(device="cuda")
g.manual_seed(1)
d = torch.randn((S, C, N), device="cuda", generator=g).cpu().numpy()
tp = torch.randn((T, S, C, L), device="cuda", generator=g).cpu().numpy()
mv = torch.randint(0, 3000, (T, S, C), device="cuda", generator=g, dtype=torch.int32).cpu().numpy()
w = np.full((T, S, C), 1.0 / (S * C), np.float32)
for _ in range(2):
    d_new = d.copy()
    t0 = time.perf_counter()
    cc = sb.matched_filter(tp, mv, w, d_new, 1, arch="gpu", check_zeros=False, device=[0])
    print(f"call: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    del cc, d_new
    time.sleep(0.5)
