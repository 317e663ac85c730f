#!/bin/bash
# Timing ablations of bp_beam_fast_kernel (run on a GPU box, from the repo root):
#   tools/ablate_bp_fast.sh [cfg3|cfg5_per_gpu]
# Builds libbpmf_hip.so variants with -DBPF_DBG=<bits> (1 no staging, 2 no barriers,
# 8 no record refills, 16 no max update; results are wrong except for 0) and times each.
set -e
CFG=${1:-cfg3}
SRC=seismic_bpmf_amd/csrc
OBJ=seismic_bpmf_amd/lib/obj
python tools/probe_bp_fast.py $CFG
for D in 1 8 9 25; do
  OUT=/tmp/bpf_dbg_$D
  mkdir -p $OUT
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DBPF_DBG=$D -c $SRC/bp_fast.hip -o $OUT/bp_fast.o
  hipcc --offload-arch=gfx950 -shared -fPIC $OUT/bp_fast.o $(ls $OBJ/*.o | grep -v bp_fast) -o $OUT/libbpmf_hip.so
  BPMF_HIP_LIB=$OUT/libbpmf_hip.so BPF_DBG_LABEL=$D python tools/probe_bp_fast.py $CFG
done
