mkdir -p gpurun_out
# the driver's own command: serial, first failure stops
(timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_suite_r4o.log 2>&1; tail -2 gpurun_out/gpu_suite_r4o.log)
timeout 1200 bash tools/refresh_profiles.sh r04 > gpurun_out/refresh_r04.log 2>&1; tail -2 gpurun_out/refresh_r04.log
(timeout 420 python tools/stress/stress_multi.py --procs 8 --threads 1 --calls 400 --hogs 160 --timeout 400 > gpurun_out/stress_r4o.log 2>&1; tail -3 gpurun_out/stress_r4o.log)
bash tools/fuzz_long.sh 250000 275000 260 random_shapes
cut -c1-600 profiles/r04_bench.json
