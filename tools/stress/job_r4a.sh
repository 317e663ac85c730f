mkdir -p gpurun_out
cp seismic_bpmf_amd/lib/libbpmf_hip.so /tmp/new_lib.so
# (1) round-3 library, native backtraces (pytest's own faulthandler off so that crash_bt keeps the signal)
cp tools/stress/libbpmf_hip_r3.so seismic_bpmf_amd/lib/libbpmf_hip.so
BPMF_CRASH_BT=1 BPMF_FUZZ_SEEDS=1000:1700 timeout 800 python -m pytest tests/test_gpu_fuzz_adjacent.py -q -m gpu -n 8 -k device_lists -p no:faulthandler > gpurun_out/fuzz_r3_crashbt2.log 2>&1
grep -c "crash_bt" gpurun_out/fuzz_r3_crashbt2.log
# (2) the new library: whole GPU suite
cp /tmp/new_lib.so seismic_bpmf_amd/lib/libbpmf_hip.so
timeout 1500 python -m pytest tests -q -m gpu -x -n 4 > gpurun_out/gpu_suite_r4a.log 2>&1
tail -5 gpurun_out/gpu_suite_r4a.log
# (3) the new library under the same stress
BPMF_CRASH_BT=1 BPMF_FUZZ_SEEDS=0:700 timeout 800 python -m pytest tests/test_gpu_fuzz_adjacent.py -q -m gpu -n 8 -k device_lists -p no:faulthandler > gpurun_out/fuzz_r4_crashbt.log 2>&1
tail -3 gpurun_out/fuzz_r4_crashbt.log
