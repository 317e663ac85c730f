mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu -n 4 > gpurun_out/gpu_suite_r4f.log 2>&1; tail -3 gpurun_out/gpu_suite_r4f.log)
timeout 2400 bash tools/refresh_profiles.sh r04 > gpurun_out/refresh_r04.log 2>&1; tail -5 gpurun_out/refresh_r04.log
cat profiles/r04_bench.json | cut -c1-1500
