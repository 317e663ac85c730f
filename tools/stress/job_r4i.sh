mkdir -p gpurun_out
python tools/probe_fuzz_seed.py 15831 17025 2>&1 | grep -v amdgpu.ids | grep "{}"
(timeout 1500 python -m pytest tests -q -m gpu -n 4 > gpurun_out/gpu_suite_r4i.log 2>&1; tail -3 gpurun_out/gpu_suite_r4i.log)
bash tools/fuzz_long.sh 0 32000 700 random_shapes
bash tools/fuzz_long.sh 0 12000 300 random_dense
BPMF_FUZZ_SEEDS=0:3000 OMP_NUM_THREADS=2 timeout 600 python -m pytest tests/test_gpu_fuzz_workflow.py tests/test_gpu_fuzz_adjacent.py tests/test_gpu_fuzz_regimes.py -q -m gpu -n 8 > gpurun_out/fuzz_other_r4i.log 2>&1; tail -2 gpurun_out/fuzz_other_r4i.log | cut -c1-200
