mkdir -p gpurun_out
bash tools/fuzz_long.sh 32000 100000 900 random_shapes
bash tools/fuzz_long.sh 12000 60000 500 random_dense
BPMF_FUZZ_SEEDS=3000:12000 OMP_NUM_THREADS=2 timeout 700 python -m pytest tests/test_gpu_fuzz_workflow.py tests/test_gpu_fuzz_adjacent.py tests/test_gpu_fuzz_regimes.py -q -m gpu -n 8 > gpurun_out/fuzz_other_r4j.log 2>&1; grep -E "passed|failed" gpurun_out/fuzz_other_r4j.log | tail -3 | cut -c1-300
