mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_variants.py tests/test_gpu_fuzz.py tests/test_gpu_compat.py -q -m gpu -n 4 -k "bp or BP" > gpurun_out/gpu_bp_r4e.log 2>&1; tail -2 gpurun_out/gpu_bp_r4e.log)
for v in 40 36 4 8; do
  BP_OPTS=bp.halves_stage=$v timeout 200 python tools/probe_bp_dense.py cfg5_per_gpu 40 2>&1 | grep "n_closest" | sed "s/^/stage=$v /"
done > gpurun_out/halves_stage_r4e.txt
cat gpurun_out/halves_stage_r4e.txt | cut -c1-120
timeout 300 python tools/phase/bp_phase.py cfg5_per_gpu 40 > gpurun_out/bp_phase_r4e.txt 2>&1; cat gpurun_out/bp_phase_r4e.txt | tail -9
timeout 200 python tools/probe_bp_dense.py cfg5_per_gpu 33,48 > gpurun_out/dense_cfg5_r4e.txt 2>&1; tail -2 gpurun_out/dense_cfg5_r4e.txt | cut -c1-200
