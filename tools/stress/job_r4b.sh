# round 4, second GPU session: the double-buffered multi-residency BP kernel, then the stress legs
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_variants.py tests/test_gpu_fuzz.py tests/test_gpu_compat.py tests/test_gpu_threads.py tests/test_gpu_threshold.py tests/test_gpu_workflow.py -q -m gpu -x -n 4 > gpurun_out/gpu_bp_r4b.log 2>&1; tail -3 gpurun_out/gpu_bp_r4b.log)
(timeout 600 python -m pytest tests/test_gpu_shares.py -q -m gpu -x -k "dense or cfg5" > gpurun_out/gpu_shares_r4b.log 2>&1; tail -3 gpurun_out/gpu_shares_r4b.log)
timeout 300 python tools/probe_bp_dense.py cfg5_per_gpu 33,40 > gpurun_out/dense_cfg5_r4b.txt 2>&1; cat gpurun_out/dense_cfg5_r4b.txt | tail -4
timeout 300 python tools/probe_bp_dense.py cfg3 10,20 > gpurun_out/dense_cfg3_r4b.txt 2>&1; cat gpurun_out/dense_cfg3_r4b.txt | tail -4
timeout 300 python tools/phase/bp_phase.py cfg5_per_gpu 40 cfg3 10 > gpurun_out/bp_phase_r4b.txt 2>&1; cat gpurun_out/bp_phase_r4b.txt | tail -40
timeout 300 python tools/phase/mf_phase.py 128 256 > gpurun_out/mf_phase_r4b.txt 2>&1; cat gpurun_out/mf_phase_r4b.txt | tail -4
# the round-3 library under the stress that kills it, with native backtraces (pytest's faulthandler off)
BPMF_STRESS_OLD_LIB=tools/stress/libbpmf_hip_r3.so BPMF_CRASH_BT=1 BPMF_FUZZ_SEEDS=0:500 timeout 700 python -m pytest tests/test_gpu_fuzz_adjacent.py -q -m gpu -n 8 -k device_lists -p no:faulthandler > gpurun_out/fuzz_r3_crashbt3.log 2>&1
grep -c "crash_bt" gpurun_out/fuzz_r3_crashbt3.log; tail -2 gpurun_out/fuzz_r3_crashbt3.log | cut -c1-200
# the new library, same test, more seeds (2 OpenMP threads per oracle call: 8 workers share 16 CPUs)
OMP_NUM_THREADS=2 BPMF_CRASH_BT=1 BPMF_FUZZ_SEEDS=0:2000 timeout 900 python -m pytest tests/test_gpu_fuzz_adjacent.py -q -m gpu -n 8 -k device_lists -p no:faulthandler > gpurun_out/fuzz_r4_crashbt2.log 2>&1
tail -2 gpurun_out/fuzz_r4_crashbt2.log | cut -c1-200
