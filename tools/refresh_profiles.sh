#!/bin/bash
# Refresh the committed profiling evidence on a GPU box:  bash tools/refresh_profiles.sh <tag>
# (gpurun_out/prof_<tag>/ is scratch; the condensed files land in profiles/).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT $R/profiles
cd /tmp && export TMPDIR=/tmp
# 1. the bench line, un-profiled
(cd $R && python bench.py) > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 > $R/profiles/${TAG}_bench.json
# 2. the same command under rocprofv3 --kernel-trace --stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py > $OUT/bench_rocprof.log 2>&1
grep '^{' $OUT/bench_rocprof.log | tail -1 > $R/profiles/${TAG}_bench_under_rocprof.json
# 3. PMC passes, counters only (no other trace domains), small targets
for c in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" \
         "SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-48)
  for t in mf bp; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc/${t}_$n -- python $R/tools/prof_$t.py > /dev/null 2>&1
  done
done
# 4. the one-rank RCCL record on the current kernels: process group initialised, BP steps end with the
#    packed-key all-reduce, detections all-gathered, the per-GPU shares of configs[3] / [4] as extras
(cd $R && BPMF_BENCH_FORCE_DIST=1 timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 --skip-cpu --skip-traffic --skip-dense) > $OUT/bench_force_dist.log 2>&1
grep '^{' $OUT/bench_force_dist.log | tail -1 > $R/profiles/${TAG}_bench_force_dist_1rank.json
# 5. spread of the BP FETCH_SIZE figure: the same pass five times
for i in 1 2 3 4 5; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_spread/bp_$i -- python $R/tools/prof_bp.py > /dev/null 2>&1
done
python - <<PYEOF > $R/profiles/${TAG}_bp_fetch_spread.txt
import csv, glob
print("BP headline launch (cfg3, interior + edge kernels), rocprofv3 --pmc FETCH_SIZE, five separate runs of tools/prof_bp.py;")
print("bytes = counter (KiB) x 1024 x 2 (gfx950: 128-byte requests tallied at 64), mean over the launches of a run")
vals = []
for i in range(1, 6):
    per = {}
    for f in glob.glob("$OUT/pmc_spread/bp_%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE" and ("bp_beam_fast" in r["Kernel_Name"] or "bp_beam_wps2" in r["Kernel_Name"]):
                per.setdefault(r["Kernel_Name"][:40], []).append(float(r["Counter_Value"]))
    tot = sum(sum(v) / len(v) for v in per.values()) * 1024 * 2
    vals.append(tot)
    print("run %d: %.3f GB fetched per launch" % (i, tot / 1e9))
if vals:
    print("min %.3f  max %.3f  mean %.3f GB; algorithmic minimum 1.08 GB" % (min(vals) / 1e9, max(vals) / 1e9, sum(vals) / len(vals) / 1e9))
PYEOF
# 6. cycle accounting of both hot kernels (second library with -DBPMF_PHASE_CYCLES)
(cd $R && python tools/phase/build_phase_lib.py > /dev/null 2>&1 && \
  timeout 600 python tools/phase/bp_phase.py cfg3 10 cfg3 20 cfg5_per_gpu 40 > $R/profiles/${TAG}_bp_fast_phase_cycles.txt 2>&1 ; \
  timeout 600 python tools/phase/mf_phase.py 64 128 192 256 configs0 > $R/profiles/${TAG}_mf_phase_cycles.txt 2>&1)
cd $R && python tools/summarize_prof.py gpurun_out/prof_$TAG $TAG > /dev/null
# profiles/ on the box is not merged back by gpurun, gpurun_out/ is: leave a copy there
mkdir -p $R/gpurun_out/profiles_$TAG && cp $R/profiles/${TAG}_bench.json $R/profiles/${TAG}_bench_under_rocprof.json \
    $R/profiles/${TAG}_kernel_stats.csv $R/profiles/${TAG}_kernel_stats_by_grid.csv $R/profiles/${TAG}_pmc.json \
    $R/profiles/${TAG}_bench_force_dist_1rank.json $R/profiles/${TAG}_bp_fetch_spread.txt \
    $R/profiles/${TAG}_bp_fast_phase_cycles.txt $R/profiles/${TAG}_mf_phase_cycles.txt $R/gpurun_out/profiles_$TAG/ 2>/dev/null
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
ls -la $R/profiles
