#!/bin/bash
# MF main kernel, lag blocks per workgroup (BPMF_MF_NSUB): time (gpu_probe) and L2-miss traffic (FETCH_SIZE).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for n in 1 2 4 8; do
  echo "BPMF_MF_NSUB=$n"
  BPMF_MF_NSUB=$n python $R/tools/gpu_probe.py mf 2>&1 | grep "^MF"
  BPMF_MF_NSUB=$n timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/mfnsub_$n -- python $R/tools/prof_mf.py 500 > /dev/null 2>&1
  python - <<PY
import csv, glob
v=[float(r["Counter_Value"]) for p in glob.glob("$R/gpurun_out/mfnsub_$n/*/*counter_collection.csv") for r in csv.DictReader(open(p)) if "mf_mfma" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("  FETCH_SIZE KiB per launch (T=500):", sum(v)/max(len(v),1), "launches", len(v))
PY
  find $R/gpurun_out/mfnsub_$n -name "*kernel_trace.csv" -delete; find $R/gpurun_out/mfnsub_$n -name "*.db" -delete
done
