mkdir -p gpurun_out
python tools/probe_mf_ntile_T.py 2>&1 | grep -v amdgpu | tee gpurun_out/mf_ntile_T_balanced.txt
python tools/probe_mf_L_ntile.py 2>&1 | grep -v amdgpu | grep "ntile=4"
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_compat.py tests/test_gpu_fullsize.py tests/test_gpu_shares.py -q -m gpu -n 4 -k "mf or MF" > gpurun_out/gpu_mf_xcd.log 2>&1; tail -2 gpurun_out/gpu_mf_xcd.log)
bash tools/fuzz_long.sh 0 20000 400 mf_random
