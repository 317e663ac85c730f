mkdir -p gpurun_out
# the driver's own command: serial, first failure stops
(timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_suite_r4k.log 2>&1; tail -3 gpurun_out/gpu_suite_r4k.log)
timeout 1500 bash tools/refresh_profiles.sh r04 > gpurun_out/refresh_r04.log 2>&1; tail -5 gpurun_out/refresh_r04.log
cat profiles/r04_bench.json | cut -c1-1500
