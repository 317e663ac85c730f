# round 4, fourth GPU session: staging placement of the multi-residency kernel; round-3 crash with CPU hogs
mkdir -p gpurun_out
for v in 40 36 8 4 48 16; do
  BP_OPTS=bp.halves_stage=$v timeout 200 python tools/probe_bp_dense.py cfg5_per_gpu 40 2>&1 | grep "n_closest" | sed "s/^/stage=$v /"
done > gpurun_out/halves_stage_r4d.txt
cat gpurun_out/halves_stage_r4d.txt | cut -c1-120
# the round-3 library: 8 processes of multi-device calls beside 160 busy-loop processes (16 CPUs granted)
timeout 700 python tools/stress/stress_multi.py --procs 8 --threads 1 --calls 400 --hogs 160 --lib tools/stress/libbpmf_hip_r3.so --out gpurun_out/stress_r3_hogs --timeout 600 > gpurun_out/stress_r3_hogs.txt 2>&1
tail -60 gpurun_out/stress_r3_hogs.txt | cut -c1-200
# the new library under the same load
timeout 500 python tools/stress/stress_multi.py --procs 8 --threads 1 --calls 400 --hogs 160 --out gpurun_out/stress_r4_hogs --timeout 400 > gpurun_out/stress_r4_hogs.txt 2>&1
tail -3 gpurun_out/stress_r4_hogs.txt | cut -c1-200
