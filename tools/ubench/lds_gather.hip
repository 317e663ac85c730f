// Micro-benchmark of the BP gather loop: per unit 4 ds_read_b64 (8 samples per lane) + 4
// v_pk_fma_f32 + 1 address add, ring of 4 units, 3 in flight ahead (s_waitcnt lgkmcnt(12)).
//   MODE 0: reads only   MODE 1: reads + fma (the beam kernel's inner loop)   MODE 2: fma only
// WPB waves per workgroup, one workgroup per CU (LDS 128 KB).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define RD64(dst, addr, o) asm volatile("ds_read_b64 %0, %1 offset:" #o : "=v"(dst) : "v"(addr))
#define PKFMA(acc, b, x) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(b), "v"(x))
template <int MODE, int WPB>
__global__ __launch_bounds__(64 * WPB) void k(float* out, int n_units)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 32768; i += 64 * WPB) lds[i] = (float)(i & 15);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned base = (unsigned)(size_t)lds + lane * 8;
    unsigned h = (threadIdx.x >> 6) * 7919u + blockIdx.x * 104729u;
    f32x2 X[4][4], ac[4], bb = {1.0f, 0.5f};
    for (int j = 0; j < 4; ++j) ac[j] = (f32x2){0, 0};
#define ISSUE(u) { h = h * 1664525u + 1013904223u; const unsigned a_ = base + (__builtin_amdgcn_readfirstlane(h >> 8) & 0x1fff0u) * 4; \
      if (MODE != 2) { RD64(X[u][0], a_, 0); RD64(X[u][1], a_, 512); RD64(X[u][2], a_, 1024); RD64(X[u][3], a_, 1536); } }
#define FMA(u) if (MODE != 0) { PKFMA(ac[0], bb, X[u][0]); PKFMA(ac[1], bb, X[u][1]); PKFMA(ac[2], bb, X[u][2]); PKFMA(ac[3], bb, X[u][3]); } \
               else { asm volatile("" :: "v"(X[u][0]), "v"(X[u][1]), "v"(X[u][2]), "v"(X[u][3])); }
    if (MODE == 2) for (int u = 0; u < 4; ++u) for (int j = 0; j < 4; ++j) X[u][j] = (f32x2){1.f, 2.f};
    ISSUE(0) ISSUE(1) ISSUE(2)
    for (int i = 0; i < n_units; i += 4) {
        ISSUE(3) if (MODE != 2) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); FMA(0)
        ISSUE(0) if (MODE != 2) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); FMA(1)
        ISSUE(1) if (MODE != 2) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); FMA(2)
        ISSUE(2) if (MODE != 2) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); FMA(3)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    FMA(0) FMA(1) FMA(2)
    out[blockIdx.x * 64 * WPB + threadIdx.x] = ac[0][0] + ac[1][1] + ac[2][0] + ac[3][1];
}
template <int MODE, int WPB>
void run()
{
    float* d; hipMalloc(&d, 256 * 1024 * sizeof(float));
    const int n = 400000;
    hipFuncSetAttribute((const void*)k<MODE, WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, WPB><<<256, 64 * WPB, 140 * 1024>>>(d, 1000);
    hipEventRecord(e0);
    k<MODE, WPB><<<256, 64 * WPB, 140 * 1024>>>(d, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * WPB * (double)n * 4 * 512;   // 4 reads x 64 lanes x 8 B per unit
    printf("mode %d, %2d waves/CU: %.1f TB/s gathered (%.1f%% of the 157.3 TB/s ds_read_b64 rate), %.1f ms\n", MODE, WPB,
           bytes / ms / 1e9, bytes / ms / 1e9 / 157.3 * 100, ms);
    hipFree(d);
}
int main() { run<0, 16>(); run<1, 16>(); run<2, 16>(); run<0, 8>(); run<1, 8>(); return 0; }
