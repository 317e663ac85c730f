// Micro-benchmark (round 4, the judge's exploration item): would v_mfma_f32_4x4x1_16B_f32 pay for the matched
// filter of SHORT templates?  In that form a tile is 16 blocks of 4 x 4 lags, the Toeplitz band is only L + 3 deep
// (L + 15 in the 16x16x4 form: 11 % of the issued MFMA flops are zeros at L = 128), every MFMA is still one exact
// fma per output (k = 1) -- but every lane consumes ONE A and ONE B value per k-step, and a lane's values of
// consecutive k-steps are consecutive in memory, so the natural delivery is one ds_read_b128 per 4 k-steps and
// operand: with 4 accumulators (4 tiles of 256 lags, the A vector shared) that is 5 ds_read_b128 per 16 MFMAs of
// 8 cycles = 5 KB of LDS per 128 matrix-pipe cycles and wave, against 5 ds_read_b32 (1.25 KB) per 4 MFMAs of 32
// cycles in the 16x16x4 form -- four times the LDS bytes per flop.
//   MODE 0: 16x16x4, 5 ds_read_b32 per k-step of 4 MFMAs, reads between the MFMAs           (the kernel's K loop)
//   MODE 1: 4x4x1, 5 ds_read_b128 per 4 k-steps of 16 MFMAs, reads between the MFMAs
//   MODE 2: 4x4x1, no LDS reads                                                             (the form's ceiling)
//   MODE 3: 16x16x4, no LDS reads
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_4x4x1_band.hip -o tools/ubench/mfma_4x4x1_band.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SB __builtin_amdgcn_sched_barrier(0)
#define RD(dst, addr, off) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define MF16(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#define MF4(a, b, c) c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(float* out, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[4 * 2048];
    for (int i = threadIdx.x; i < 4 * 2048; i += 256) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        lds[i] = (float)(int)(h & 0xffff) * 3.0517578e-5f - 1.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    f32x4 acc[4];
    for (int u = 0; u < 4; ++u) acc[u] = (f32x4){0, 0, 0, 0};
    if (MODE == 0) {
        unsigned ap = (unsigned)(size_t)(lds + wv * 2048 + (lane & 15));
        unsigned bp = (unsigned)(size_t)(lds + wv * 2048 + 64 + 18 * (lane & 15) + (lane >> 4));
        float sa[4], sb[4][4];
#define REQ(s, ao, bo) RD(sa[s], ap, ao); RD(sb[s][0], bp, bo); RD(sb[s][1], bp, (bo) + 1152); RD(sb[s][2], bp, (bo) + 2304); RD(sb[s][3], bp, (bo) + 3456)
#define STEP(cur, req, ao, bo) \
    asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); SB; MF16(sa[cur], sb[cur][0], acc[0]); SB; RD(sa[req], ap, ao); SB; \
    MF16(sa[cur], sb[cur][1], acc[1]); SB; RD(sb[req][0], bp, bo); SB; MF16(sa[cur], sb[cur][2], acc[2]); SB; RD(sb[req][1], bp, (bo) + 1152); SB; \
    MF16(sa[cur], sb[cur][3], acc[3]); SB; RD(sb[req][2], bp, (bo) + 2304); RD(sb[req][3], bp, (bo) + 3456); SB
        REQ(0, 0, 0); REQ(1, 16, 16);
        for (int i = 0; i < iters; ++i) { STEP(0, 2, 32, 32); STEP(1, 3, 48, 48); STEP(2, 0, 0, 0); STEP(3, 1, 16, 16); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (MODE == 1) {
        // operand vectors of 4 consecutive k-steps: A (shared by the 4 tiles) and one B per tile, double-buffered
        unsigned p = (unsigned)(size_t)(lds + wv * 2048) + lane * 16;
        f32x4 va[2], vb[2][4];
#define REQ4(s, o) RD128(va[s], p, o); RD128(vb[s][0], p, (o) + 1024); RD128(vb[s][1], p, (o) + 2048); RD128(vb[s][2], p, (o) + 3072); RD128(vb[s][3], p, (o) + 4096)
        // 16 MFMAs of chunk `cur` with the 5 reads of chunk `nxt` spread between them
#define CHUNK(cur, nxt, o) \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB; \
    MF4(va[cur][0], vb[cur][0][0], acc[0]); SB; RD128(va[nxt], p, o); SB; MF4(va[cur][0], vb[cur][1][0], acc[1]); MF4(va[cur][0], vb[cur][2][0], acc[2]); SB; \
    RD128(vb[nxt][0], p, (o) + 1024); SB; MF4(va[cur][0], vb[cur][3][0], acc[3]); MF4(va[cur][1], vb[cur][0][1], acc[0]); MF4(va[cur][1], vb[cur][1][1], acc[1]); SB; \
    RD128(vb[nxt][1], p, (o) + 2048); SB; MF4(va[cur][1], vb[cur][2][1], acc[2]); MF4(va[cur][1], vb[cur][3][1], acc[3]); MF4(va[cur][2], vb[cur][0][2], acc[0]); SB; \
    RD128(vb[nxt][2], p, (o) + 3072); SB; MF4(va[cur][2], vb[cur][1][2], acc[1]); MF4(va[cur][2], vb[cur][2][2], acc[2]); MF4(va[cur][2], vb[cur][3][2], acc[3]); SB; \
    RD128(vb[nxt][3], p, (o) + 4096); SB; MF4(va[cur][3], vb[cur][0][3], acc[0]); MF4(va[cur][3], vb[cur][1][3], acc[1]); MF4(va[cur][3], vb[cur][2][3], acc[2]); \
    MF4(va[cur][3], vb[cur][3][3], acc[3]); SB
        REQ4(0, 0);
        for (int i = 0; i < iters; ++i) { CHUNK(0, 1, 0); CHUNK(1, 0, 0); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (MODE == 2) {
        float a = (float)lane, b[4] = {1.f, 2.f, 3.f, 4.f};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int r = 0; r < 8; ++r) { MF4(a, b[0], acc[0]); MF4(a, b[1], acc[1]); MF4(a, b[2], acc[2]); MF4(a, b[3], acc[3]); }
    } else {
        float a = (float)lane, b[4] = {1.f, 2.f, 3.f, 4.f};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { MF16(a, b[0], acc[0]); MF16(a, b[1], acc[1]); MF16(a, b[2], acc[2]); MF16(a, b[3], acc[3]); }
    }
    float s = 0;
    for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* what)
{
    float* d;
    (void)hipMalloc(&d, (1 << 20) * sizeof(float));
    const int iters = 40000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<256 * 4, 256>>>(d, 10);
    (void)hipEventRecord(e0);
    k<MODE><<<256 * 4, 256>>>(d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // flop per wave and iteration: MODE 0 / 3: 16 MFMAs of 2048; MODE 1 / 2: 32 MFMAs of 512
    const double per_iter = (MODE == 0 || MODE == 3) ? 16 * 2048.0 : 32 * 512.0;
    const double flop = 256.0 * 4 * 4 * (double)iters * per_iter;
    printf("mode %d (%s): %.1f TFLOP/s issued (%.1f%% of 157.3), %.1f ms\n", MODE, what, flop / ms / 1e9,
           flop / ms / 1e9 / 157.3 * 100, ms);
    fflush(stdout);
    (void)hipFree(d);
}
int main()
{
    for (int rep = 0; rep < 2; ++rep) {
        run<3>("16x16x4, no LDS reads");
        run<0>("16x16x4, 5 ds_read_b32 per 4 MFMAs");
        run<2>("4x4x1, no LDS reads");
        run<1>("4x4x1, 5 ds_read_b128 per 16 MFMAs");
    }
    return 0;
}
