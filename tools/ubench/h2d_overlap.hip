// Does a host-to-device copy from PAGEABLE memory on one stream run beside a kernel on another?  (round 5: the
// host-pointer calls upload the day in pieces while the first kernels run.)  A spin kernel of ~T ms on stream A,
// 1 GiB to the device on stream B by: 0 = hipMemcpyAsync from pageable, 1 = hipMemcpy2DAsync (60 rows) from
// pageable, 2 = hipMemcpyAsync from pinned, 3 = pageable -> pinned pieces (host threads) -> hipMemcpyAsync,
// 4 = hipMemcpy2DAsync of a STRIDED piece (60 rows of 4 MiB, 16 MiB apart) from pageable, 5 = the same piece as 60
// hipMemcpyAsync calls, 6 = the strided piece from pinned memory, 7 = mode 4 from memory nobody has touched before.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/h2d_overlap.hip -o tools/ubench/h2d_overlap.bin -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
__global__ void spin(long long cycles, int* out)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out) out[0] = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void pcopy(char* dst, const char* src, size_t n, int nth)
{
    std::vector<std::thread> th;
    const size_t per = (n / nth + 4095) & ~(size_t)4095;
    for (int i = 0; i < nth; ++i) {
        const size_t o = (size_t)i * per;
        if (o >= n) break;
        th.emplace_back([=] { memcpy(dst + o, src + o, std::min(per, n - o)); });
    }
    for (auto& t : th) t.join();
}
int main()
{
    const size_t bytes = (size_t)1 << 30, rows = 60, row = bytes / rows / 4 * 4;
    char* h = (char*)malloc(bytes);
    memset(h, 1, bytes);
    char* hp; hipHostMalloc((void**)&hp, bytes, hipHostMallocDefault);
    memset(hp, 2, bytes);
    char* pin[2]; hipHostMalloc((void**)&pin[0], 64 << 20, 0); hipHostMalloc((void**)&pin[1], 64 << 20, 0);
    char* d; hipMalloc((void**)&d, bytes);
    int* flag; hipMalloc((void**)&flag, 4);
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipEvent_t ev[2]; hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
    const long long cyc = 100000 * 100;            // wall_clock64 runs at 100 MHz: 100 ms
    const size_t sw = (size_t)4 << 20, sp = (size_t)16 << 20;      // strided piece: 60 rows x 4 MiB, pitch 16 MiB (240 MiB moved)
    char* hs = (char*)malloc(60 * sp); memset(hs, 3, 60 * sp);
    char* hps; hipHostMalloc((void**)&hps, 60 * sp, hipHostMallocDefault); memset(hps, 4, 60 * sp);
    for (int mode = 0; mode < 8; ++mode) {
        for (int with_kernel = 0; with_kernel < 2; ++with_kernel) {
            hipDeviceSynchronize();
            const double t0 = now();
            if (with_kernel) spin<<<256, 64, 0, a>>>(cyc, flag);
            const double t1 = now();
            if (mode == 0) hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, b);
            else if (mode == 1) hipMemcpy2DAsync(d, row, h, row, row, rows, hipMemcpyHostToDevice, b);
            else if (mode == 2) hipMemcpyAsync(d, hp, bytes, hipMemcpyHostToDevice, b);
            else if (mode == 4) hipMemcpy2DAsync(d, sw, hs, sp, sw, 60, hipMemcpyHostToDevice, b);
            else if (mode == 5) { for (int r = 0; r < 60; ++r) hipMemcpyAsync(d + r * sw, hs + r * sp, sw, hipMemcpyHostToDevice, b); }
            else if (mode == 6) hipMemcpy2DAsync(d, sw, hps, sp, sw, 60, hipMemcpyHostToDevice, b);
            else if (mode == 7) { char* fresh = (char*)malloc(60 * sp); for (size_t i = 0; i < 60 * sp; i += 4096) fresh[i] = 1; hipMemcpy2DAsync(d, sw, fresh, sp, sw, 60, hipMemcpyHostToDevice, b); }
            else {
                const size_t P = 64 << 20;
                for (size_t o = 0, q = 0; o < bytes; o += P, ++q) {
                    if (q >= 2) hipEventSynchronize(ev[q & 1]);
                    pcopy(pin[q & 1], h + o, std::min(P, bytes - o), 8);
                    hipMemcpyAsync(d + o, pin[q & 1], std::min(P, bytes - o), hipMemcpyHostToDevice, b);
                    hipEventRecord(ev[q & 1], b);
                }
            }
            const double t2 = now();
            hipStreamSynchronize(b);
            const double t3 = now();
            hipStreamSynchronize(a);
            const double t4 = now();
            printf("mode %d kernel %d: copy call returned after %6.1f ms, copy done %6.1f ms, all done %6.1f ms (%.1f GB/s)\n", mode,
                   with_kernel, (t2 - t1) * 1e3, (t3 - t1) * 1e3, (t4 - t0) * 1e3, (mode >= 4 ? 60 * sw : bytes) / (t3 - t1) / 1e9);
        }
    }
    return 0;
}
