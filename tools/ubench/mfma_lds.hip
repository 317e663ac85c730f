// Micro-benchmark: fp32 MFMA stream (4 independent 16x16x4 tiles per wave, 4 waves/SIMD) with
// the K-loop's LDS operand reads placed between the MFMAs, as in mf_mfma_wave_kernel.
//   MODE 0: no LDS reads            MODE 1: 5 ds_read_b32 per k-step (4 MFMAs)
//   MODE 2: 5 ds_read2_b32 per 2 k-steps (same bytes, half the LDS instructions)
// hipcc --offload-arch=gfx950 -O3 mfma_lds.hip -o mfma_lds.bin && ./mfma_lds.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define SB __builtin_amdgcn_sched_barrier(0)
#define RD(dst, addr, off) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define RD2(dst, addr, o0, o1) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "n"(o0), "n"(o1))
#define MF(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
template <int MODE, bool RANDOM_DATA>
__global__ __launch_bounds__(256, 4) void k(float* out, int iters, const float* __restrict__ gsrc)
{
    __shared__ float lds[4 * 1664];
    for (int i = threadIdx.x; i < 4 * 1664; i += 256) { unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; lds[i] = RANDOM_DATA ? (float)(int)(h & 0xffff) * 3.0517578e-5f - 1.0f : (float)(i & 7); }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned ap = (unsigned)(size_t)(lds + wv * 1664 + (lane & 15));
    unsigned bp = (unsigned)(size_t)(lds + wv * 1664 + 64 + 18 * (lane & 15) + (lane >> 4));
    f32x4 acc[4];
    for (int u = 0; u < 4; ++u) acc[u] = (f32x4){0, 0, 0, 0};
    if (MODE == 0) {
        float a = lane, b[4] = {1.f, 2.f, 3.f, 4.f};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { MF(a, b[0], acc[0]); MF(a, b[1], acc[1]); MF(a, b[2], acc[2]); MF(a, b[3], acc[3]); }
    } else if (MODE == 1) {
        float sa[4], sb[4][4];
#define REQ(s, ao, bo) RD(sa[s], ap, ao); RD(sb[s][0], bp, bo); RD(sb[s][1], bp, (bo) + 1152); RD(sb[s][2], bp, (bo) + 2304); RD(sb[s][3], bp, (bo) + 3456)
#define STEP(cur, req, ao, bo) \
    asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); SB; MF(sa[cur], sb[cur][0], acc[0]); SB; RD(sa[req], ap, ao); SB; \
    MF(sa[cur], sb[cur][1], acc[1]); SB; RD(sb[req][0], bp, bo); SB; MF(sa[cur], sb[cur][2], acc[2]); SB; RD(sb[req][1], bp, (bo) + 1152); SB; \
    MF(sa[cur], sb[cur][3], acc[3]); SB; RD(sb[req][2], bp, (bo) + 2304); RD(sb[req][3], bp, (bo) + 3456); SB
        REQ(0, 0, 0); REQ(1, 16, 16);
        for (int i = 0; i < iters; ++i) { STEP(0, 2, 32, 32); STEP(1, 3, 48, 48); STEP(2, 0, 0, 0); STEP(3, 1, 16, 16); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (MODE >= 3) {
        if (MODE == 4) {   // stagger by hardware wave slot
            const unsigned slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 3u;
            for (unsigned i = 0; i < slot; ++i) __builtin_amdgcn_s_sleep(127);
            if (lane == 0 && blockIdx.x < 4) out[1 << 20 | (blockIdx.x * 4 + wv)] = (float)slot;
        }
        if (MODE == 5) {   // stagger by a hash of the workgroup id
            unsigned h = blockIdx.x * 2654435761u; h ^= h >> 13;
            for (unsigned i = 0; i < (h & 7u); ++i) __builtin_amdgcn_s_sleep(64);
        }
        // channel structure of the real kernel: 17 trips, drain, ~60 VALU on the accumulators,
        // 25 ds_write, refill -- per "channel"
        float sa[4], sb[4][4];
        float extra[16];
        for (int u = 0; u < 16; ++u) extra[u] = (float)u;
        for (int c = 0; c < iters / 17; ++c) {
            REQ(0, 0, 0); REQ(1, 16, 16);
            for (int i = 0; i < 17; ++i) { STEP(0, 2, 32, 32); STEP(1, 3, 48, 48); STEP(2, 0, 0, 0); STEP(3, 1, 16, 16); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SB;
            if (MODE == 13) {  // DMA issued first, epilogue overlaps its latency
#pragma unroll
            for (int r = 0; r < 7; ++r)
                __builtin_amdgcn_global_load_lds(gsrc + ((c * 7 + r) & 255) * 256 + lane * 4,
                                                 (__attribute__((address_space(3))) void*)(lds + wv * 1664 + 80 + 256 * (r % 5)), 16, 0, 0);
            }
            if (MODE == 3 || MODE == 7 || MODE == 13) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float nrm = extra[4 * u + r] * 0.999f;
                    float cc = nrm < 1000.f ? acc[u][r] * nrm : 0.f;
                    extra[4 * u + r] = __fmaf_rn(0.5f, cc, extra[4 * u + r]);
                    acc[u][r] = 0.0f;
                }
            }
            if (MODE == 3 || MODE == 8) {
#pragma unroll
            for (int r = 0; r < 25; ++r) ((volatile float*)lds)[wv * 1664 + 80 + lane + 64 * (r % 20)] = extra[r & 15];
            }
            if (MODE == 13) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == 11) {  // LDS-DMA: global -> LDS without VGPRs (25 x 256 B per wave)
#pragma unroll
            for (int r = 0; r < 25; ++r)
                __builtin_amdgcn_global_load_lds(gsrc + ((c * 25 + r) & 1023) * 64 + lane,
                                                 (__attribute__((address_space(3))) void*)(lds + wv * 1664 + 80 + 64 * (r % 20)), 4, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (MODE == 12) {  // LDS-DMA, 16 bytes per lane (7 x 1 KB per wave)
#pragma unroll
            for (int r = 0; r < 7; ++r)
                __builtin_amdgcn_global_load_lds(gsrc + ((c * 7 + r) & 255) * 256 + lane * 4,
                                                 (__attribute__((address_space(3))) void*)(lds + wv * 1664 + 80 + 256 * (r % 5)), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (MODE == 9) {   // same bytes as 8-byte stores
#pragma unroll
            for (int r = 0; r < 13; ++r) { f32x2 v = {extra[r & 15], extra[(r + 1) & 15]}; ((volatile f32x2*)(lds + wv * 1664 + 80))[lane + 64 * (r % 10)] = v; }
            }
            if (MODE == 10) {  // same bytes as 16-byte stores
#pragma unroll
            for (int r = 0; r < 7; ++r) { f32x4 v = {extra[r & 15], extra[(r + 1) & 15], extra[(r + 2) & 15], extra[(r + 3) & 15]}; ((volatile f32x4*)(lds + wv * 1664 + 80))[lane + 64 * (r % 5)] = v; }
            }
            SB;
        }
        for (int u = 0; u < 4; ++u) acc[u][0] += extra[4 * u] + extra[4 * u + 1] + extra[4 * u + 2] + extra[4 * u + 3];
    } else {
        f32x2 sa[2], sb[2][4];
        unsigned b1 = bp + 1152, b2 = bp + 2304, b3 = bp + 3456;
#define REQ2(s, o) RD2(sa[s], ap, o, (o) + 4); RD2(sb[s][0], bp, o, (o) + 4); RD2(sb[s][1], b1, o, (o) + 4); RD2(sb[s][2], b2, o, (o) + 4); RD2(sb[s][3], b3, o, (o) + 4)
#define PAIR(cur, nxt, o) \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB; MF(sa[cur][0], sb[cur][0][0], acc[0]); SB; RD2(sa[nxt], ap, o, (o) + 4); SB; \
    MF(sa[cur][0], sb[cur][1][0], acc[1]); SB; RD2(sb[nxt][0], bp, o, (o) + 4); SB; MF(sa[cur][0], sb[cur][2][0], acc[2]); SB; RD2(sb[nxt][1], b1, o, (o) + 4); SB; \
    MF(sa[cur][0], sb[cur][3][0], acc[3]); SB; RD2(sb[nxt][2], b2, o, (o) + 4); SB; MF(sa[cur][1], sb[cur][0][1], acc[0]); SB; RD2(sb[nxt][3], b3, o, (o) + 4); SB; \
    MF(sa[cur][1], sb[cur][1][1], acc[1]); MF(sa[cur][1], sb[cur][2][1], acc[2]); MF(sa[cur][1], sb[cur][3][1], acc[3]); SB
        REQ2(0, 0);
        for (int i = 0; i < iters; ++i) { PAIR(0, 1, 8); PAIR(1, 0, 0); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0;
    for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, bool RANDOM_DATA>
void run(int wps = 4)
{
    float* d; hipMalloc(&d, (2 << 20) * sizeof(float));
    const int iters = 17 * 3000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t dyn = wps >= 4 ? 0 : (wps == 3 ? 24 : (wps == 2 ? 50 : 110)) * 1024;  // limit workgroups per CU
    hipFuncSetAttribute((const void*)k<MODE, RANDOM_DATA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4 * 1664 * 4);
    k<MODE, RANDOM_DATA><<<256 * wps, 256, dyn>>>(d, 10, d);
    hipEventRecord(e0);
    k<MODE, RANDOM_DATA><<<256 * wps, 256, dyn>>>(d, iters, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * wps * 4 * iters * 16 * 2048.0;
    printf("waves/SIMD=%d random=%d mode %d: %.1f TFLOP/s (%.1f%% of 157.3)\n", wps, (int)RANDOM_DATA, MODE, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
    hipFree(d);
}
int main() { run<1, true>(4); run<1, true>(4); run<1, true>(2); run<1, true>(1); run<6, true>(4); run<7, true>(4); run<8, true>(4); run<9, true>(4); run<10, true>(4); run<3, true>(4); run<12, true>(4); run<13, true>(4); return 0; }
