"""Timeline of the host-pointer BP calls of tools/probe_bp_e2e.py from a rocprofv3 trace:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python tools/probe_bp_e2e.py --calls 4
   python tools/analyse_bp_e2e_trace.py DIR
Per call (a burst of bp_beam_fast kernels): when its first / last kernel ran, the busy time of its kernels, and its
H2D copies (count, bytes, summed duration, rate)."""
import csv
import glob
import os
import sys

d = sys.argv[1]
kern, cop = [], []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    kern += list(csv.DictReader(open(f)))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    cop += list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in kern)
cs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Bytes", 0) or 0), r.get("Direction", r.get("Kind", ""))) for r in cop)
beam = [k for k in ks if "bp_beam_fast" in k[2]]
calls, cur = [], []
for k in beam:
    if cur and k[0] - cur[-1][1] > 50e6:
        calls.append(cur)
        cur = []
    cur.append(k)
if cur:
    calls.append(cur)
print(f"{len(beam)} bp_beam_fast launches in {len(calls)} bursts; {len(cs)} copies")
for i, c in enumerate(calls):
    t0, t1 = c[0][0], c[-1][1]
    busy = sum(e - s for s, e, _ in c)
    mine = [x for x in cs if t0 - 150e6 <= x[0] <= t1 and "HOST_TO_DEVICE" in x[3].upper().replace(" ", "_") and x[1] - x[0] >= 100e3]
    nbytes = sum(x[2] for x in mine)
    dur = sum(x[1] - x[0] for x in mine)
    first = (mine[0][0] - t0) / 1e6 if mine else float("nan")
    last = (mine[-1][1] - t0) / 1e6 if mine else float("nan")
    gaps = [(c[j][0] - c[j - 1][1]) / 1e6 for j in range(1, len(c))]
    print(f"burst {i}: {len(c)} launches, span {(t1 - t0) / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, largest gaps "
          f"{sorted(gaps)[-3:]} ms; H2D copies of >= 0.1 ms: {len(mine)} copies, {nbytes / 1e9:.2f} GB, summed {dur / 1e6:.1f} ms "
          f"({nbytes / max(dur, 1):.1f} GB/s), first starts {first:.1f} ms / last ends {last:.1f} ms relative to the first kernel")
    if mine:
        print("   copies (start ms rel. first kernel, ms, MB): " +
              " ".join(f"[{(x[0] - t0) / 1e6:.1f} {(x[1] - x[0]) / 1e6:.1f} {x[2] / 1e6:.0f}]" for x in mine[:24]))
