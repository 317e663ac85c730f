mkdir -p gpurun_out
timeout 330 bash tools/refresh_profiles.sh r04 > gpurun_out/refresh_r04.log 2>&1; tail -2 gpurun_out/refresh_r04.log
(timeout 240 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_suite_r4p.log 2>&1; tail -2 gpurun_out/gpu_suite_r4p.log)
cut -c1-400 profiles/r04_bench.json
