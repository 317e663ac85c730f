#!/bin/bash
# A long randomized parity session on the GPU box: seeds A..B of the signed-moveout sweeps of
# tests/test_gpu_fuzz.py (the default run takes seeds 0..39), 8 xdist workers (a worker that dies
# with the process -- a GPU memory fault aborts it -- is reported and replaced).
# Usage: tools/fuzz_long.sh 40 1500 [seconds] [-k expr] [test file]
set -u
A=${1:-40}; B=${2:-1000}; K=${4:-random_shapes}
mkdir -p gpurun_out
# (2 OpenMP threads per oracle call: 8 workers on the 16 CPUs a box grants; 16 idle-spinning threads per worker
# made round 3's sessions ~10x slower than they had to be)
OMP_NUM_THREADS=${OMP_NUM_THREADS:-2} BPMF_FUZZ_SEEDS=$A:$B timeout ${3:-1500} python -m pytest ${5:-tests/test_gpu_fuzz.py} -q -m gpu -k "$K" -n 8 \
    > gpurun_out/fuzz_long_${A}_${B}.log 2>&1
grep -E "^FAILED|^ERROR|crashed|passed|failed" gpurun_out/fuzz_long_${A}_${B}.log | cut -c1-400 | head -80
