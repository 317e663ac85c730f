// Does the VGPR bank of the A / B operands matter for v_mfma_f32_16x16x4_f32?  Pure MFMA stream,
// 4 accumulators, explicit registers: A in v100, B in v10{0,1,2,3}+4k (same / different bank mod 4).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BOFF>
__global__ __launch_bounds__(256, 4) void k(float* out, int iters)
{
    float r;
    asm volatile(
        "v_mov_b32 v100, 1.0\n v_mov_b32 v101, 2.0\n v_mov_b32 v102, 0.5\n v_mov_b32 v103, 3.0\n"
        "v_mov_b32 v104, 1.0\n v_mov_b32 v105, 2.0\n v_mov_b32 v106, 0.5\n v_mov_b32 v107, 3.0\n"
        "v_mov_b32 v108, 1.0\n v_mov_b32 v109, 2.0\n v_mov_b32 v110, 0.5\n v_mov_b32 v111, 3.0\n"
        "v_mov_b32 v112, 1.0\n v_mov_b32 v113, 2.0\n v_mov_b32 v114, 0.5\n v_mov_b32 v115, 3.0\n"
        "v_mov_b32 v116, 1.0\n v_mov_b32 v117, 2.0\n v_mov_b32 v118, 0.5\n v_mov_b32 v119, 3.0\n"
        "v_mov_b32 v0, 0\n v_mov_b32 v1, 0\n v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n"
        "v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n"
        "s_mov_b32 s20, %1\n"
        "1:\n"
        "v_mfma_f32_16x16x4_f32 v[0:3], v100, v[104+%2], v[0:3]\n"
        "v_mfma_f32_16x16x4_f32 v[4:7], v100, v[108+%2], v[4:7]\n"
        "v_mfma_f32_16x16x4_f32 v[8:11], v100, v[112+%2], v[8:11]\n"
        "v_mfma_f32_16x16x4_f32 v[12:15], v100, v[116+%2], v[12:15]\n"
        "v_mfma_f32_16x16x4_f32 v[0:3], v100, v[104+%2], v[0:3]\n"
        "v_mfma_f32_16x16x4_f32 v[4:7], v100, v[108+%2], v[4:7]\n"
        "v_mfma_f32_16x16x4_f32 v[8:11], v100, v[112+%2], v[8:11]\n"
        "v_mfma_f32_16x16x4_f32 v[12:15], v100, v[116+%2], v[12:15]\n"
        "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
        "s_nop 7\n s_nop 7\n v_add_f32 %0, v0, v5\n"
        : "=v"(r) : "s"(iters), "n"(BOFF)
        : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15",
          "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","s20","scc","memory");
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int BOFF> void run()
{
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    const int iters = 200000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<BOFF><<<1024, 256>>>(d, 100);
    hipEventRecord(e0);
    k<BOFF><<<1024, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 1024.0 * 4 * iters * 8 * 2048.0;
    printf("B operand in bank (A bank + %d) mod 4: %.1f TFLOP/s (%.1f%% of 157.3)\n", BOFF, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
    hipFree(d);
}
int main() { run<0>(); run<0>(); run<1>(); run<2>(); run<3>(); return 0; }
