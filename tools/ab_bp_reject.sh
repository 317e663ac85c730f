#!/bin/bash
# Same-box A/B of the max / arg-max update of bp_beam_fast_kernel: BPF_REJECT=1 (early reject against the
# workgroup-wide threshold) vs 0 (per-run running max, 8 compares + 16 selects per source).
set -e
CFG=${1:-cfg3}
SRC=seismic_bpmf_amd/csrc
OBJ=seismic_bpmf_amd/lib/obj
for R in 0 1 0 1; do
  OUT=/tmp/bpf_rej_$R
  if [ ! -f $OUT/libbpmf_hip.so ]; then
    mkdir -p $OUT
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DBPF_REJECT=$R -c $SRC/bp_fast.hip -o $OUT/bp_fast.o
    hipcc --offload-arch=gfx950 -shared -fPIC $OUT/bp_fast.o $(ls $OBJ/*.o | grep -v bp_fast) -o $OUT/libbpmf_hip.so
  fi
  echo "BPF_REJECT=$R"
  BPMF_HIP_LIB=$OUT/libbpmf_hip.so BPF_DBG_LABEL=rej$R python tools/probe_bp_fast.py $CFG 2>&1 | grep -v amdgpu.ids
done
