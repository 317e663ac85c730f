// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 with NACC independent accumulators,
// WPS waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_f32.hip -o mfma_f32 && ./mfma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, bool SAME_A>
__global__ void k(float* out, int iters, float a0, float b0)
{
    f32x4 acc[NACC];
    for (int u = 0; u < NACC; ++u) acc[u] = (f32x4){0, 0, 0, 0};
    float a = a0 + threadIdx.x, b[NACC];
    for (int u = 0; u < NACC; ++u) b[u] = b0 + u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int u = 0; u < NACC; ++u)
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(SAME_A ? a : b[(u + 1) % NACC], b[u], acc[u], 0, 0, 0);
        }
    }
    float s = 0;
    for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, bool SAME_A>
void run(int wps)
{
    float* d; hipMalloc(&d, 256 * 8 * 64 * 4 * sizeof(float) * 4);
    int iters = 4000;
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, SAME_A><<<grid, block>>>(d, 10, 1.f, 2.f);
    hipEventRecord(e0);
    k<NACC, SAME_A><<<grid, block>>>(d, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid.x * 4 * iters * 16 * NACC * 2048.0;
    printf("NACC=%d sameA=%d waves/SIMD=%d: %.1f TFLOP/s (%.1f%% of 157.3)\n", NACC, (int)SAME_A, wps, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
    hipFree(d);
}
int main()
{
    for (int wps : {1, 2, 4}) { run<4, true>(wps); run<4, false>(wps); run<2, true>(wps); run<8, true>(wps); run<1, true>(wps); }
    return 0;
}
