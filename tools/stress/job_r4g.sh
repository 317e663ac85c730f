mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_threads.py tests/test_gpu_fuzz_adjacent.py tests/test_gpu_fullsize.py tests/test_gpu_decimate.py -q -m gpu -n 4 > gpurun_out/gpu_host_r4g.log 2>&1; tail -2 gpurun_out/gpu_host_r4g.log)
timeout 600 python bench.py --skip-cpu --skip-traffic --skip-dense > gpurun_out/bench_r4g.log 2>&1; grep '^{' gpurun_out/bench_r4g.log | tail -1 > gpurun_out/bench_r4g.json
python - <<'PY'
import json
b=json.load(open('gpurun_out/bench_r4g.json'))
print('MF', b['value'], b['ms_per_step'], b['roofline']['frac'], 'e2e', b['end_to_end']['mf_calls_ms'])
print('BP', b['bp']['ms_per_step'], b['bp']['roofline']['frac'], 'e2e', b['bp']['end_to_end']['ms'], b['bp']['end_to_end']['first_call_ms'], b['bp']['end_to_end']['equals_resident_result'])
PY
bash tools/fuzz_long.sh 40 2000 500 random_shapes
bash tools/fuzz_long.sh 30 1500 300 random_dense
BPMF_FUZZ_SEEDS=0:400 OMP_NUM_THREADS=2 timeout 400 python -m pytest tests/test_gpu_fuzz_workflow.py tests/test_gpu_fuzz_adjacent.py tests/test_gpu_fuzz_regimes.py -q -m gpu -n 8 > gpurun_out/fuzz_other_r4g.log 2>&1; tail -2 gpurun_out/fuzz_other_r4g.log | cut -c1-200
