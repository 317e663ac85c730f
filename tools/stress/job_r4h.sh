mkdir -p gpurun_out
tools/ubench/lds_gather3.bin 3000000 1 > gpurun_out/ubench_gather3.txt 2>&1; cat gpurun_out/ubench_gather3.txt
bash tools/fuzz_long.sh 2000 32000 600 random_shapes
bash tools/fuzz_long.sh 1500 11500 300 random_dense
