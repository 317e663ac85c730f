// Micro-benchmark #4 of the BP gather loop (round 5): the loop a ds_read_b128 variant of bp_beam_fast_kernel would
// run at tile 256 -- one (station, phase) term of a source = ONE 16-byte gather per lane (a lane owns 4 consecutive
// samples; the windows would be staged four times, shifted by 0..3 samples, so that every gather is 16-byte
// aligned) -- against the production loop (KIND 0 of lds_gather3.hip: tile 512, four ds_read_b64 per term).
// Same skeleton: address = wave-uniform SGPR offset + one v_add per term, counted lgkmcnt waits, v_pk_fma_f32 with
// the weight as an SGPR-pair operand, the max / arg-max update behind every source of TERMS terms.
//   KIND 0: production b64 loop, 4 gathers + 4 pk_fma per term (8 samples per lane), update of 8 samples
//   KIND 1: b128 loop, 1 gather + 2 pk_fma per term (4 samples per lane), update of 4 samples, ring of 8 gathers
//   KIND 2: b128 loop, a lane owns 8 samples = 2 gathers + 4 pk_fma per term (tile 512 at 4 copies: does not fit
//           the LDS for real plans; the loop shape's ceiling only)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_gather4.hip -o tools/ubench/lds_gather4.bin
//   tools/ubench/lds_gather4.bin [terms per wave] [terms per source]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
#define RD64(dst, addr, o) asm volatile("ds_read_b64 %0, %1 offset:" #o : "=v"(dst) : "v"(addr))
#define RD128(dst, addr, o) asm volatile("ds_read_b128 %0, %1 offset:" #o : "=v"(dst) : "v"(addr))
#define PKFMA_S(acc, sp, x) \
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(sp), "v"(x))
#define WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

template <int KIND, int WPB, int TERMS>
__global__ __launch_bounds__(64 * WPB) void k(float* out, int n_terms, int stride, int prio_mode)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 35840; i += 64 * WPB) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        lds[i] = (float)(h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned base = (unsigned)(size_t)lds + lane * (KIND == 0 ? 8 : 16);
    unsigned so = (unsigned)(wv * 1237 + blockIdx.x * 77) & 0x3ffcu;
    i32x2 sp;
    sp[0] = 0; sp[1] = __float_as_int(0.5f);
    asm volatile("" : "+s"(sp));
    constexpr int NS = KIND == 1 ? 4 : 8;            // samples per lane
    f32x2 ac[NS / 2];
#pragma unroll
    for (int j = 0; j < NS / 2; ++j) ac[j] = (f32x2){0, 0};
    float best[NS]; int arg[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) { best[j] = -1.0f; arg[j] = 0; }
    constexpr int RING = KIND == 1 ? 8 : 4;          // terms in flight
    f32x2 X[4][4];
    f32x4 Y[8][2];
#define NEXT_ADDR() (so = (so + stride) & 0x3ffcu, base + so * 4)
#define ISSUE(u) { const unsigned a_ = NEXT_ADDR(); \
        if (KIND == 0) { RD64(X[(u) & 3][0], a_, 0); RD64(X[(u) & 3][1], a_, 512); RD64(X[(u) & 3][2], a_, 1024); RD64(X[(u) & 3][3], a_, 1536); } \
        else if (KIND == 1) { RD128(Y[(u) & 7][0], a_, 0); } \
        else { RD128(Y[(u) & 3][0], a_, 0); RD128(Y[(u) & 3][1], a_, 1024); } }
#define FMA(u) { if (KIND == 0) { _Pragma("unroll") for (int j = 0; j < 4; ++j) PKFMA_S(ac[j], sp, X[(u) & 3][j]); } \
        else if (KIND == 1) { f32x2 a0 = {Y[(u) & 7][0][0], Y[(u) & 7][0][1]}, a1 = {Y[(u) & 7][0][2], Y[(u) & 7][0][3]}; \
            PKFMA_S(ac[0], sp, a0); PKFMA_S(ac[1], sp, a1); } \
        else { f32x2 a0 = {Y[(u) & 3][0][0], Y[(u) & 3][0][1]}, a1 = {Y[(u) & 3][0][2], Y[(u) & 3][0][3]}, \
                     a2 = {Y[(u) & 3][1][0], Y[(u) & 3][1][1]}, a3 = {Y[(u) & 3][1][2], Y[(u) & 3][1][3]}; \
            PKFMA_S(ac[0], sp, a0); PKFMA_S(ac[1], sp, a1); PKFMA_S(ac[2], sp, a2); PKFMA_S(ac[3], sp, a3); } }
#define WW { if (KIND == 0) WAIT(12); else if (KIND == 1) WAIT(7); else WAIT(6); }
    // round 5b: issue priorities.  1: static, by the wave's place on its SIMD (waves go to SIMDs round-robin, so
    // wv >> 2 = 0..3 on each); 2: rotating -- every source the wave takes the next level; 3: static for the first
    // half of the waves' work, none afterwards is not modelled here (the kernel's progress schedule needs barriers)
    int pl = (wv >> 2) & 3;
    auto setp = [&](int l) { if (l == 3) __builtin_amdgcn_s_setprio(3); else if (l == 2) __builtin_amdgcn_s_setprio(2); else if (l == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); };
    if (prio_mode == 1) setp(pl);
#pragma unroll
    for (int u = 0; u < RING - 1; ++u) ISSUE(u)
    for (int i = 0; i < n_terms; i += TERMS) {
        if (prio_mode == 2) { pl = (pl + 1) & 3; setp(pl); }
#pragma unroll
        for (int u = 0; u < TERMS; ++u) { ISSUE(u + RING - 1) WW FMA(u) }
        const int sid = i;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const float a = ac[j >> 1][j & 1];
            const bool take = a > best[j];
            best[j] = take ? a : best[j];
            arg[j] = take ? sid : arg[j];
        }
#pragma unroll
        for (int j = 0; j < NS / 2; ++j) ac[j] = (f32x2){0, 0};
    }
    WAIT(0);
    float r = 0;
#pragma unroll
    for (int j = 0; j < NS / 2; ++j) r += ac[j][0] + ac[j][1];
#pragma unroll
    for (int j = 0; j < NS; ++j) r += best[j] + (float)arg[j];
    out[blockIdx.x * 64 * WPB + threadIdx.x] = r;
}

static long g_terms = 2000000;
template <int KIND, int WPB, int TERMS>
void run(int stride, int prio_mode = 0)
{
    float* d; hipMalloc(&d, 256 * 2048 * sizeof(float));
    hipFuncSetAttribute((const void*)k<KIND, WPB, TERMS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = (int)(g_terms * 16 / WPB / (KIND == 1 ? 1 : 2));      // the same bytes per CU for every shape
    k<KIND, WPB, TERMS><<<256, 64 * WPB, 140 * 1024>>>(d, 1000 * TERMS, stride, prio_mode);
    hipEventRecord(e0);
    k<KIND, WPB, TERMS><<<256, 64 * WPB, 140 * 1024>>>(d, n / TERMS * TERMS, stride, prio_mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * WPB * (double)(n / TERMS * TERMS) * 64.0 * (KIND == 1 ? 16 : 32);
    printf("prio %d kind %d, %2d waves/CU, %2d terms/source: %.1f TB/s gathered (%.1f%% of 157.3), %.1f ms\n", prio_mode, KIND, WPB, TERMS,
           bytes / ms / 1e9, bytes / ms / 1e9 / 157.3 * 100, ms);
    fflush(stdout);
    hipFree(d);
}
int main(int argc, char** argv)
{
    if (argc > 1) g_terms = atol(argv[1]);
    if (argc > 2) {          // round 5b: static / rotating issue priorities on the production loop and the b128 loop
        for (int rep = 0; rep < 2; ++rep)
            for (int pm = 0; pm < 3; ++pm) { run<0, 16, 20>(338, pm); run<1, 16, 20>(340, pm); run<2, 16, 20>(340, pm); }
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 16, 20>(338);
        run<1, 16, 20>(340); run<1, 8, 20>(340); run<1, 4, 20>(340);
        run<2, 16, 20>(340); run<2, 8, 20>(340); run<2, 4, 20>(340);
        run<1, 16, 40>(340); run<1, 8, 40>(340); run<1, 8, 80>(340);
    }
    return 0;
}
