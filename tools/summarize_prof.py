"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into the small files kept under profiles/."""
import csv, collections, json, os, sys

def kernel_stats(path, out):
    rows = list(csv.DictReader(open(path)))
    with open(out, "w") as f:
        f.write("name,calls,total_ms,avg_ms,pct\n")
        for r in rows:
            if float(r["Percentage"]) < 0.01:
                continue
            name = r["Name"].split("(")[0].replace("void ", "")
            f.write(f'{name},{r["Calls"]},{float(r["TotalDurationNs"])/1e6:.3f},{float(r["AverageNs"])/1e6:.4f},{r["Percentage"]}\n')

def kernel_stats_by_grid(trace_path, out, kernels=("mf_mfma", "mf_split", "bp_beam_fast", "bp_beam_wps2")):
    """rocprofv3's kernel_stats.csv averages ALL launches of a kernel, and bench.py launches the
    hot kernels at several sizes (the timed steps at the full workload, plus the end-to-end call's
    batches, the planted-event checks and the small configs[0] runs of the CPU-baseline leg).  This
    breaks the hot kernels' launches down by grid size from the kernel trace, so that the row of the
    timed workload can be compared with `roofline.avg_launch_ms` of the bench line."""
    rows = list(csv.DictReader(open(trace_path)))
    if not rows:
        return
    cols = rows[0].keys()
    gcols = [c for c in cols if c.lower().startswith("grid_size")]
    wcols = [c for c in cols if c.lower().startswith("workgroup_size")]
    groups = collections.defaultdict(list)
    for r in rows:
        name = r.get("Kernel_Name", "")
        if not any(k in name for k in kernels):
            continue
        grid = "x".join(str(r[c]) for c in gcols)
        wg = "x".join(str(r[c]) for c in wcols)
        groups[(name.split("(")[0].replace("void ", ""), grid, wg)].append(
            (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
    with open(out, "w") as f:
        f.write("name,grid_size,workgroup_size,calls,total_ms,avg_ms,min_ms,max_ms\n")
        for (name, grid, wg), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            f.write(f'"{name}",{grid},{wg},{len(v)},{sum(v):.3f},{sum(v)/len(v):.4f},{min(v):.4f},{max(v):.4f}\n')


def pmc(paths, kernels, out):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            for k in kernels:
                if k in r["Kernel_Name"]:
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {k: {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in cs.items()} for k, cs in agg.items()}
    json.dump(res, open(out, "w"), indent=1)
    return res

if __name__ == "__main__":
    d, tag = sys.argv[1], sys.argv[2]
    os.makedirs("profiles", exist_ok=True)
    ks = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("kernel_stats.csv")]
    if ks:  # the directory may hold earlier runs as well: newest file wins
        kernel_stats(max(ks, key=os.path.getmtime), f"profiles/{tag}_kernel_stats.csv")
    kt = [os.path.join(r, f) for r, _, fs in os.walk(os.path.join(d, "stats")) for f in fs if f.endswith("kernel_trace.csv")]
    if kt:
        kernel_stats_by_grid(max(kt, key=os.path.getmtime), f"profiles/{tag}_kernel_stats_by_grid.csv")
    pm = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
    if pm:
        newest = {}   # one counter file per pass directory: the newest
        for f in pm:
            d_ = os.path.dirname(os.path.dirname(f))
            if d_ not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d_]):
                newest[d_] = f
        pm = list(newest.values())
        print(json.dumps(pmc(pm, ["mf_mfma", "bp_beam_fast", "bp_beam_wps2", "mf_csum_local", "bp_prestack", "tdt_window", "bp_window_stats", "bp_extract_peaks", "intertp_cc"], f"profiles/{tag}_pmc.json"), indent=1))
