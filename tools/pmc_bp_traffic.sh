#!/bin/bash
# HBM traffic of the BP beam kernel at cfg3: separate FETCH_SIZE / WRITE_SIZE passes.
TAG=${1:-bp}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  env "$@" timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/bptraffic_$TAG/$c -- python $R/tools/prof_bp.py > /dev/null 2>&1
done
find $R/gpurun_out/bptraffic_$TAG -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for p in glob.glob("$R/gpurun_out/bptraffic_$TAG/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        for kern in ("bp_beam_fast", "bp_beam_wps2"):     # interior kernel / the few edge tiles of a day
            if kern in r["Kernel_Name"]:
                agg[(kern, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (kern, k), v in sorted(agg.items()):
    print("$TAG", kern, k, "KiB mean per launch", sum(v) / len(v), "launches", len(v))
PY
