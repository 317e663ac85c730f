// Micro-benchmark #3 of the BP gather loop (round 4): does ds_read2_b64 -- ONE instruction for two of a
// unit's four 8-byte gathers (same address register, offsets 0 / 512 B resp. 1024 / 1536 B) -- raise the rate of
// the production loop?  Same skeleton as lds_gather2.hip KIND 6 (address = SGPR offset + one v_add per unit, ring
// of 4 units, counted waits, 4 v_pk_fma_f32 with an SGPR-pair weight, the max / arg-max update every 20 units).
//   KIND 0: 4 ds_read_b64 per unit, s_waitcnt lgkmcnt(12)                (the kernel as it is)
//   KIND 1: 2 ds_read2_b64 per unit, s_waitcnt lgkmcnt(6)
//   KIND 2: 1 ds_read2_b64 + 2 ds_read_b64 per unit (mixed), lgkmcnt(9)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_gather3.hip -o tools/ubench/lds_gather3.bin
//   tools/ubench/lds_gather3.bin [units per wave] [random LDS contents 0/1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
#define RD64(dst, addr, o) asm volatile("ds_read_b64 %0, %1 offset:" #o : "=v"(dst) : "v"(addr))
#define RD2_64(dst, addr, o0, o1) asm volatile("ds_read2_b64 %0, %1 offset0:" #o0 " offset1:" #o1 : "=v"(dst) : "v"(addr))
#define PKFMA_S(acc, sp, x) \
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(sp), "v"(x))
#define WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

template <int KIND, int WPB>
__global__ __launch_bounds__(64 * WPB) void k(float* out, int n_units, int stride, int random_data)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 35840; i += 64 * WPB) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        lds[i] = random_data ? (float)(h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f : (float)(i & 15);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned base = (unsigned)(size_t)lds + lane * 8;
    unsigned so = (unsigned)(wv * 1237 + blockIdx.x * 77) & 0x3ffeu;
    i32x2 sp;
    sp[0] = 0; sp[1] = __float_as_int(0.5f);
    asm volatile("" : "+s"(sp));
    f32x2 ac[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ac[j] = (f32x2){0, 0};
    float best[8]; int arg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -1.0f; arg[j] = 0; }
    f32x2 X[4][4];
    f32x4 Y[4][2];
#define ISSUE(u) { so = (so + stride) & 0x3ffeu; const unsigned a_ = base + so * 4; \
        if (KIND == 0) { RD64(X[u][0], a_, 0); RD64(X[u][1], a_, 512); RD64(X[u][2], a_, 1024); RD64(X[u][3], a_, 1536); } \
        else if (KIND == 1) { RD2_64(Y[u][0], a_, 0, 64); RD2_64(Y[u][1], a_, 128, 192); } \
        else { RD2_64(Y[u][0], a_, 0, 64); RD64(X[u][2], a_, 1024); RD64(X[u][3], a_, 1536); } }
#define FMA(u) { if (KIND == 0) { _Pragma("unroll") for (int j = 0; j < 4; ++j) PKFMA_S(ac[j], sp, X[u][j]); } \
        else if (KIND == 1) { f32x2 a0 = {Y[u][0][0], Y[u][0][1]}, a1 = {Y[u][0][2], Y[u][0][3]}, a2 = {Y[u][1][0], Y[u][1][1]}, a3 = {Y[u][1][2], Y[u][1][3]}; \
            PKFMA_S(ac[0], sp, a0); PKFMA_S(ac[1], sp, a1); PKFMA_S(ac[2], sp, a2); PKFMA_S(ac[3], sp, a3); } \
        else { f32x2 a0 = {Y[u][0][0], Y[u][0][1]}, a1 = {Y[u][0][2], Y[u][0][3]}; \
            PKFMA_S(ac[0], sp, a0); PKFMA_S(ac[1], sp, a1); PKFMA_S(ac[2], sp, X[u][2]); PKFMA_S(ac[3], sp, X[u][3]); } }
#define WW { if (KIND == 0) WAIT(12); else if (KIND == 1) WAIT(6); else WAIT(9); }
    ISSUE(0) ISSUE(1) ISSUE(2)
    int since = 0;
    for (int i = 0; i < n_units; i += 4) {
        ISSUE(3) WW FMA(0)
        ISSUE(0) WW FMA(1)
        ISSUE(1) WW FMA(2)
        ISSUE(2) WW FMA(3)
        since += 4;
        if (since == 20) {  // wave-uniform: one "source" done
            since = 0;
            const int sid = i;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = ac[j >> 1][j & 1];
                const bool take = a > best[j];
                best[j] = take ? a : best[j];
                arg[j] = take ? sid : arg[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) ac[j] = (f32x2){0, 0};
        }
    }
    WAIT(0);
    FMA(0) FMA(1) FMA(2)
    float r = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) r += ac[j][0] + ac[j][1];
#pragma unroll
    for (int j = 0; j < 8; ++j) r += best[j] + (float)arg[j];
    out[blockIdx.x * 64 * WPB + threadIdx.x] = r;
}

static int g_random = 1;
static int g_units = 3000000;
template <int KIND, int WPB>
void run(int stride)
{
    float* d; hipMalloc(&d, 256 * 2048 * sizeof(float));
    hipFuncSetAttribute((const void*)k<KIND, WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, WPB><<<256, 64 * WPB, 140 * 1024>>>(d, 1000, stride, g_random);
    hipEventRecord(e0);
    k<KIND, WPB><<<256, 64 * WPB, 140 * 1024>>>(d, g_units, stride, g_random);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * WPB * (double)g_units * 4 * 512.0;
    printf("kind %d, %2d waves/CU, random %d: %.1f TB/s gathered (%.1f%% of 157.3), %.1f ms\n", KIND, WPB, g_random,
           bytes / ms / 1e9, bytes / ms / 1e9 / 157.3 * 100, ms);
    fflush(stdout);
    hipFree(d);
}
int main(int argc, char** argv)
{
    if (argc > 1) g_units = atoi(argv[1]);
    if (argc > 2) g_random = atoi(argv[2]);
    for (int rep = 0; rep < 2; ++rep) { run<0, 16>(338); run<1, 16>(338); run<2, 16>(338); }
    return 0;
}
