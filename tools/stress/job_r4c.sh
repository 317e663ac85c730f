# round 4, third GPU session
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_variants.py tests/test_gpu_fuzz.py tests/test_gpu_compat.py tests/test_gpu_threads.py tests/test_gpu_threshold.py tests/test_gpu_workflow.py tests/test_gpu_parity.py -q -m gpu -n 4 > gpurun_out/gpu_bp_r4c.log 2>&1; tail -3 gpurun_out/gpu_bp_r4c.log)
(timeout 600 python -m pytest tests/test_gpu_shares.py -q -m gpu -x -k "dense" > gpurun_out/gpu_shares_r4c.log 2>&1; tail -3 gpurun_out/gpu_shares_r4c.log)
timeout 300 python tools/probe_bp_dense.py cfg5_per_gpu 33,40 > gpurun_out/dense_cfg5_r4c.txt 2>&1; cat gpurun_out/dense_cfg5_r4c.txt | tail -2
timeout 300 python tools/phase/bp_phase.py cfg5_per_gpu 40 cfg3 10 > gpurun_out/bp_phase_r4c.txt 2>&1; cat gpurun_out/bp_phase_r4c.txt | tail -20
timeout 300 python tools/phase/mf_phase.py 128 256 > gpurun_out/mf_phase_r4c.txt 2>&1; cat gpurun_out/mf_phase_r4c.txt | tail -3
# the round-3 library under the stress that kills it, native backtraces; 2 OpenMP threads per oracle call
OMP_NUM_THREADS=2 BPMF_STRESS_OLD_LIB=tools/stress/libbpmf_hip_r3.so BPMF_CRASH_BT=1 BPMF_FUZZ_SEEDS=0:30000 timeout 420 python -m pytest tests/test_gpu_fuzz_adjacent.py -q -m gpu -n 8 -k device_lists -p no:faulthandler > gpurun_out/fuzz_r3_crashbt4.log 2>&1
grep -c "crash_bt" gpurun_out/fuzz_r3_crashbt4.log; grep -E "passed|failed|node down" gpurun_out/fuzz_r3_crashbt4.log | tail -3 | cut -c1-200
# the new library, the same test, twice
for i in 1 2; do
OMP_NUM_THREADS=2 BPMF_CRASH_BT=1 BPMF_FUZZ_SEEDS=0:20000 timeout 420 python -m pytest tests/test_gpu_fuzz_adjacent.py -q -m gpu -n 8 -k device_lists -p no:faulthandler > gpurun_out/fuzz_r4_run$i.log 2>&1
grep -E "passed|failed|node down" gpurun_out/fuzz_r4_run$i.log | tail -2 | cut -c1-200
done
