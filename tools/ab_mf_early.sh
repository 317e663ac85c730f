#!/bin/bash
# Same-box A/B of mf_mfma_wave_kernel with -DMF_EARLY_STAGE=0/1 (next channel's LDS stores / LDS-DMA
# copies issued before the epilogue of the current channel), register-staged and BPMF_MF_DMA=1.
set -e
SRC=seismic_bpmf_amd/csrc
OBJ=seismic_bpmf_amd/lib/obj
for E in 0 1; do
  OUT=/tmp/mf_early_$E
  mkdir -p $OUT
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DMF_EARLY_STAGE=$E -c $SRC/mf.hip -o $OUT/mf.o
  hipcc --offload-arch=gfx950 -shared -fPIC $OUT/mf.o $(ls $OBJ/*.o | grep -v "/mf.hip.o") -o $OUT/libbpmf_hip.so
done
for rep in 1 2; do
  for E in 0 1; do
    for D in 0 1; do
      echo "MF_EARLY_STAGE=$E BPMF_MF_DMA=$D"
      BPMF_HIP_LIB=/tmp/mf_early_$E/libbpmf_hip.so BPMF_MF_DMA=$D python tools/gpu_probe.py mf 2>&1 | grep -v amdgpu.ids | tail -2
    done
  done
done
if [ -n "${1:-}" ]; then
  BPMF_HIP_LIB=/tmp/mf_early_1/libbpmf_hip.so timeout 900 python -m pytest tests -q -m gpu -x -k "mf or matched" 2>&1 | tail -2
fi
