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
cd $R && python tools/summarize_prof.py gpurun_out/prof_$TAG $TAG > /dev/null
# profiles/ on the box is not merged back by gpurun, gpurun_out/ is: leave a copy there
mkdir -p $R/gpurun_out/profiles_$TAG && cp $R/profiles/${TAG}_bench.json $R/profiles/${TAG}_bench_under_rocprof.json \
    $R/profiles/${TAG}_kernel_stats.csv $R/profiles/${TAG}_kernel_stats_by_grid.csv $R/profiles/${TAG}_pmc.json $R/gpurun_out/profiles_$TAG/ 2>/dev/null
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
ls -la $R/profiles
