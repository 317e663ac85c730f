#!/bin/bash
# PMC passes over the split-precision MF kernel (option mf.split16) at cfg2's shape, counters only, one group per run:
#   bash tools/pmc_mf_split16.sh <tag> [T]        -> profiles/<tag>_mf_split16_pmc.txt
TAG=$1; T=${2:-500}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/split16pmc_$TAG
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" \
         "SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_UNALIGNED_STALL"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-48)
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$n -- python $R/tools/prof_mf.py $T split16 > /dev/null 2>&1
done
python - <<PY > $R/gpurun_out/${TAG}_mf_split16_pmc.txt
import csv, glob, collections
agg = collections.defaultdict(list)
dur = []
for p in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "mf_split_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for p in glob.glob("$OUT/*/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(p)):
        if "mf_split_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("mf_split_kernel at T = $T x 20 x 3, L = 256, N = 8 640 000 (tools/prof_mf.py $T split16), rocprofv3 --pmc, one counter group per run")
if dur:
    print("kernel duration under the counter passes: mean %.2f ms over %d launches (min %.2f, max %.2f)" % (sum(dur) / len(dur), len(dur), min(dur), max(dur)))
for k in sorted(agg):
    v = agg[k]
    print("%-28s mean per launch %.6g   (%d launches)" % (k, sum(v) / len(v), len(v)))
if "FETCH_SIZE" in agg:
    f = sum(agg["FETCH_SIZE"]) / len(agg["FETCH_SIZE"]) * 1024 * 2
    print("HBM fetched per launch: %.1f GB (FETCH_SIZE KiB x 1024 x 2 on gfx950)" % (f / 1e9))
if "WRITE_SIZE" in agg:
    w = sum(agg["WRITE_SIZE"]) / len(agg["WRITE_SIZE"]) * 1024
    print("HBM written per launch: %.2f GB" % (w / 1e9))
if "SQ_VALU_MFMA_BUSY_CYCLES" in agg and "SQ_BUSY_CYCLES" in agg:
    print("SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES = %.3f" % ((sum(agg["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(agg["SQ_VALU_MFMA_BUSY_CYCLES"])) / (sum(agg["SQ_BUSY_CYCLES"]) / len(agg["SQ_BUSY_CYCLES"]))))
if "SQ_LDS_BANK_CONFLICT" in agg and "SQ_LDS_IDX_ACTIVE" in agg:
    print("SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.4f" % ((sum(agg["SQ_LDS_BANK_CONFLICT"]) / len(agg["SQ_LDS_BANK_CONFLICT"])) / (sum(agg["SQ_LDS_IDX_ACTIVE"]) / len(agg["SQ_LDS_IDX_ACTIVE"]))))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
cat $R/gpurun_out/${TAG}_mf_split16_pmc.txt
