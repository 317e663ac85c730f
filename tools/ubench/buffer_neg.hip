// What does a raw buffer load return when voffset is negative (wraps to ~4 GiB) and voffset + the
// instruction's immediate offset lands back inside the buffer?  (mf.hip stages its data windows with
// such loads at the first lags of a template with negative moveouts.)
// Build: hipcc --offload-arch=gfx950 -O3 -o buffer_neg.bin buffer_neg.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int IMM>
__global__ void probe(const float* buf, int n, int k, float* out)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, n * 4, 0x00020000);
    const int lane = threadIdx.x;
    int vo = (lane - k) * 4 - IMM;
    asm volatile("" : "+v"(vo));                 // keep the immediate out of the register
    out[lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo + IMM, 0, 0));
}

// the other end: voffset inside the buffer, voffset + immediate crossing num_records
template <int IMM>
__global__ void probe_end(const float* buf, int n, int k, float* out)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, n * 4, 0x00020000);
    const int lane = threadIdx.x;
    int vo = (n - 64 + lane + k) * 4 - IMM;
    asm volatile("" : "+v"(vo));
    out[lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo + IMM, 0, 0));
}

template <int IMM>
static void run_end(const float* d, int n, float* dout)
{
    for (int k = 0; k <= 5; ++k) {
        probe_end<IMM><<<1, 64>>>(d, n, k, dout);
        float h[64];
        hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) bad += h[l] != (n - 64 + l + k < n ? (float)(n - 64 + l + k + 1) : 0.0f);
        printf("end imm %4d k %d: wrong lanes: %d\n", IMM, k, bad);
    }
}

template <int IMM>
static void run(const float* d, int n, float* dout)
{
    for (int k = 0; k <= 9; ++k) {
        probe<IMM><<<1, 64>>>(d, n, k, dout);
        float h[64];
        hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
        printf("imm %4d k %d:", IMM, k);
        for (int l = 0; l < 16; ++l) printf(" %g", h[l]);
        int bad = 0;
        for (int l = 0; l < 64; ++l) bad += h[l] != (l - k >= 0 ? (float)(l - k + 1) : 0.0f);
        printf("   wrong lanes: %d\n", bad);
    }
}

int main()
{
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)(i + 1);
    float *d, *dout;
    hipMalloc(&d, n * 4 + 4096);
    d += 512;                                     // memory in front of the buffer is mapped
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMalloc(&dout, 256);
    run<0>(d, n, dout);
    run<256>(d, n, dout);
    run<1024>(d, n, dout);
    run_end<0>(d, n, dout);
    run_end<256>(d, n, dout);
    run_end<1024>(d, n, dout);
    return 0;
}
