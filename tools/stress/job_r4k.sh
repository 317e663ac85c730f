mkdir -p gpurun_out
bash tools/fuzz_long.sh 100000 250000 1500 random_shapes
bash tools/fuzz_long.sh 60000 110000 500 random_dense
