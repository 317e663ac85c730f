// Post-CC detection step on the device ("next" row 1): the RMS
// time-dependent threshold of BPMF/libc.c:516-673 for a whole (templates x n_corr) CC matrix
// that already lives in HBM, and the extraction of the samples above it, so that only a few
// thousand candidate records -- not 17-173 GB of CCs -- leave the GPU or cross xGMI.
//
// Conventions = oracle/adjacent_oracle.c:tdt_rms_cpu, which is pinned bit for bit to the
// reference's compiled libc.c (single-threaded).  Bit-exactness fixes the summation order:
// every window statistic is a SEQUENTIAL float accumulation over the window (with the squares
// formed in double, as the reference's pow(x - m, 2)), so the parallelism is one thread per
// (series, window) -- 24-32 k threads for a day of 500 templates -- each streaming its window.
// This is the one HBM-bound stage of the workflow (reads the CC matrix four times).
#include "common.h"
#include <algorithm>
#include <type_traits>
#include <utility>
#include "../../include/bpmf_hip.h"

namespace bpmf {

constexpr int GAUSSIAN_LEN = 500;

// dword-aligned 16-byte load: rows and windows start at arbitrary sample offsets
typedef float f32x4a __attribute__((ext_vector_type(4), aligned(4)));
// the same in the GLOBAL address space: a pointer that went through __shfl is generic, and a FLAT load
// counts in lgkmcnt too -- every wait for the LDS tile would drain the whole load pipeline
typedef const __attribute__((address_space(1))) f32x4a* tdt_gptr;
typedef float f32x4v __attribute__((ext_vector_type(4)));   // 16-byte aligned (LDS tile)

// Stream one window per LANE through f(value, index) in strictly ascending order.  The adds of a
// window form one dependent chain, so a thread owns a whole window and a wave walks 64 windows
// side by side -- 64 different cache lines per step.  Rounds 1-2 let every lane load its own line
// 16 bytes at a time: 8 load instructions per line, each of them touching 64 lines, and the stage
// ran at what the texture addresser makes of that (2.1-2.7 TB/s of the matrix, whatever the depth
// of the load pipeline).  Now the wave reads every line ONCE: load k of a step has lanes 8j .. 8j+7
// fetch the eight 16-byte pieces of the line of window 8k + j -- 8 lines per instruction instead
// of 64 -- and a transposition through a 9 KB LDS tile hands each lane its own 32 samples.  The
// loads of the next TDT_DEPTH steps are in flight while the current 32 samples are accumulated.
// Round 3, measured on cfg2's CC matrix (500 x 8.64 M): with the chain removed a pass streams at
// 5.2 TB/s (3.35 ms per 17.3 GB) -- what HBM gives 24 000 concurrent 128-byte streams; the passes
// now run at 3.8-4.4 ms (glob) and 8.3 ms for the window kernel's 2 x 1.31 passes (5.4 TB/s).
// Call with all 64 lanes of a one-wave workgroup (`p` of an idle lane: any readable window).
#ifndef TDT_STEP_V
#define TDT_STEP_V 32
#endif
constexpr int TDT_STEP = TDT_STEP_V;          // samples per lane and step: 32 = one 128-byte line
constexpr int TDT_NL = TDT_STEP / 4;         // 16-byte pieces per window and step = load instructions per step
constexpr int TDT_WPL = 64 / TDT_NL;         // windows served by one load instruction
constexpr int TDT_ROW = TDT_STEP + 4;        // tile row stride in floats (16-byte aligned, spreads the banks)
#ifndef TDT_DEPTH_V
#define TDT_DEPTH_V 6
#endif
constexpr int TDT_DEPTH = TDT_DEPTH_V;       // steps of loads in flight
template <typename F, int... I>
__device__ __forceinline__ void tdt_unroll(std::integer_sequence<int, I...>, F&& f)
{
    (f(std::integral_constant<int, I>{}), ...);
}
// f(value, index): every sample.  g(value, index): used instead of f for the 32 samples of a step in
// which this lane holds an exact zero (the window kernel's replacement path: kept out of the hot loop).
template <typename F, typename G>
__device__ __forceinline__ void tdt_stream2(const float* __restrict__ p, size_t window, float* tile, F f, G g)
{
    const int lane = threadIdx.x & 63;
    const int piece = lane % TDT_NL, sub = lane / TDT_NL;
    const size_t nstep = window / TDT_STEP;
    // the windows whose pieces this lane fetches: TDT_WPL k + sub, k = 0 .. TDT_NL - 1
    tdt_gptr src[TDT_NL];
#pragma unroll
    for (int k = 0; k < TDT_NL; ++k) {
        const unsigned long long q = __shfl((unsigned long long)(size_t)p, TDT_WPL * k + sub, 64);
        src[k] = (tdt_gptr)((const float*)(size_t)q + 4 * piece);
    }
    // TDT_DEPTH steps of loads in flight per wave (the stage has only ~2 waves per CU)
    f32x4a reg[TDT_DEPTH][TDT_NL];
    // Every load below is UNCONDITIONAL (past the last step the index is clamped: the last step is
    // read again and dropped).  With the loads under `if (b + DEPTH < nstep)` the compiler's wait
    // insertion lost count at the merge and waited for vmcnt(7..0) in front of the first tile write of
    // every round -- that is, for the loads of the steps AHEAD as well: the depth bought nothing.
    const size_t last = nstep ? nstep - 1 : 0;
#pragma unroll
    for (int d = 0; d < TDT_DEPTH && nstep; ++d) {
        const size_t at = (size_t)d < last ? (size_t)d : last;
#pragma unroll
        for (int k = 0; k < TDT_NL; ++k) reg[d][k] = src[k][at * TDT_NL];
    }
    auto step = [&](size_t b, auto slot_c) __attribute__((always_inline)) {
        constexpr int SL = decltype(slot_c)::value;
        // (LDS operations of one wave execute in order: the reads of the previous step are done)
#pragma unroll
        for (int k = 0; k < TDT_NL; ++k) *(f32x4v*)(tile + (TDT_WPL * k + sub) * TDT_ROW + 4 * piece) = (f32x4v)reg[SL][k];
        {
            const size_t at = b + TDT_DEPTH < last ? b + TDT_DEPTH : last;
#pragma unroll
            for (int k = 0; k < TDT_NL; ++k) reg[SL][k] = src[k][at * TDT_NL];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4v mine[TDT_NL];
#pragma unroll
        for (int i = 0; i < TDT_NL; ++i) mine[i] = *(const f32x4v*)(tile + lane * TDT_ROW + 4 * i);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float lo = INFINITY;                      // independent of the chain: fills its latency
#pragma unroll
        for (int i = 0; i < TDT_NL; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) lo = fminf(lo, fabsf(mine[i][e]));
        if (__builtin_expect(lo == 0.0f, 0)) {
#pragma unroll 1
            for (int i = 0; i < TDT_NL; ++i)
#pragma unroll 1
                for (int e = 0; e < 4; ++e) g(mine[i][e], b * TDT_STEP + 4 * i + e);
        } else {
#pragma unroll
            for (int i = 0; i < TDT_NL; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) f(mine[i][e], b * TDT_STEP + 4 * i + e);
        }
    };
    size_t b = 0;
    for (; b + TDT_DEPTH <= nstep; b += TDT_DEPTH)
        tdt_unroll(std::make_integer_sequence<int, TDT_DEPTH>{}, [&](auto sc) __attribute__((always_inline)) { step(b + decltype(sc)::value, sc); });
    tdt_unroll(std::make_integer_sequence<int, TDT_DEPTH - 1>{}, [&](auto sc) __attribute__((always_inline)) {
        if (b < nstep) { step(b, sc); ++b; }
    });
    for (size_t j = nstep * TDT_STEP; j < window; ++j) g(p[j], j);
}
template <typename F>
__device__ __forceinline__ void tdt_stream(const float* __restrict__ p, size_t window, float* tile, F f)
{
    tdt_stream2(p, window, tile, f, f);
}

// (1) per (row, global window): sum and count of the non-zero samples.       libc.c:553-571
__global__ __launch_bounds__(64) void tdt_glob_sum_kernel(const float* __restrict__ x, size_t n_rows, size_t n,
                                    size_t window, size_t n_glob, float* __restrict__ part,
                                    unsigned long long* __restrict__ cnt)
{
    __shared__ __attribute__((aligned(16))) float tile[64 * TDT_ROW];
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = idx < n_rows * n_glob;      // idle lanes of the last wave walk window 0 and store nothing
    size_t row = live ? idx / n_glob : 0, q = live ? idx % n_glob : 0;
    const float* p = x + row * n + q * window;
    float acc = 0.0f;
    unsigned c = 0;                               // (window < 2^31: tdt_sizes)
    // strictly sequential adds.  Adding an exact zero leaves a float sum that started at +0 unchanged,
    // sign included (x + -0 = x, +0 + -0 = +0), so only the count needs the test.
    tdt_stream(p, window, tile, [&](float v, size_t) {
        acc += v;
        c += v != 0.0f ? 1u : 0u;
    });
    if (!live) return;
    part[idx] = acc;
    cnt[idx] = c;
}

// (2) per row: centre = (sequential sum of the partials) / count.            libc.c:567-573
__global__ void tdt_glob_centre_kernel(const float* __restrict__ part,
                                       const unsigned long long* __restrict__ cnt, size_t n_rows,
                                       size_t n_glob, float* __restrict__ centre,
                                       unsigned long long* __restrict__ total)
{
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    float acc = 0.0f;
    unsigned long long c = 0;
    for (size_t q = 0; q < n_glob; ++q) { acc += part[row * n_glob + q]; c += cnt[row * n_glob + q]; }
    centre[row] = acc / (float)c;
    total[row] = c;
}

// (3) per (row, global window): sum of squared deviations of the non-zero samples (float
//     accumulator, squares in double).                                       libc.c:574-586
__global__ __launch_bounds__(64) void tdt_glob_dev_kernel(const float* __restrict__ x, const float* __restrict__ centre,
                                    size_t n_rows, size_t n, size_t window, size_t n_glob,
                                    float* __restrict__ part)
{
    __shared__ __attribute__((aligned(16))) float tile[64 * TDT_ROW];
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = idx < n_rows * n_glob;
    size_t row = live ? idx / n_glob : 0, q = live ? idx % n_glob : 0;
    const float* p = x + row * n + q * window;
    const float c = centre[row];
    float acc = 0.0f;
    tdt_stream(p, window, tile, [&](float v, size_t) {
        if (v != 0.0f) {
            double d = (double)(v - c);
            acc = (float)((double)acc + d * d);
        }
    });
    if (live) part[idx] = acc;
}

// (4) per row: dev = sqrtf(sum / count).                                      libc.c:584-587
__global__ void tdt_glob_std_kernel(const float* __restrict__ part,
                                    const unsigned long long* __restrict__ total, size_t n_rows,
                                    size_t n_glob, float* __restrict__ dev)
{
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    float acc = 0.0f;
    for (size_t q = 0; q < n_glob; ++q) acc += part[row * n_glob + q];
    dev[row] = sqrtf(acc / (float)total[row]);
}

// (5) per (row, sliding window): mean + num_dev * std of the window, with exact zeros replaced
//     on the fly by centre + g[i mod 500] * dev -- a separate multiply and add, like mean + num_dev * std
//     below: the reference is compiled with -std=c99, an ISO mode in which gcc contracts nothing (its
//     binary holds no FMA; round 1 had assumed fused operations here, oracle/adjacent_oracle.c).
//                                                                            libc.c:606-627
__global__ __launch_bounds__(64) void tdt_window_kernel(const float* __restrict__ x, const float* __restrict__ gauss,
                                  const float* __restrict__ centre, const float* __restrict__ dev,
                                  float num_dev, size_t n_rows, size_t n, size_t window,
                                  size_t shift, size_t n_win, float* __restrict__ thr_win)
{
    __shared__ __attribute__((aligned(16))) float tile[64 * TDT_ROW];
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = idx < n_rows * n_win;
    size_t row = live ? idx / n_win : 0, q = live ? idx % n_win : 0;
    const size_t i0 = q * shift;
    const float* p = x + row * n + i0;
    const float c = centre[row], dv = dev[row];
    float acc = 0.0f;
    unsigned g0 = (unsigned)(i0 % GAUSSIAN_LEN);  // gauss index of sample j: (g0 + j) mod 500
    tdt_stream2(p, window, tile, [&](float v, size_t) { acc += v; },
                [&](float v, size_t j) {
                    if (v == 0.0f) v = __fadd_rn(c, __fmul_rn(gauss[(g0 + j) % GAUSSIAN_LEN], dv));
                    acc += v;
                });
    const float mean = acc / (float)window;
    float ss = 0.0f;
    tdt_stream2(p, window, tile,
                [&](float v, size_t) {
                    double d = (double)(v - mean);
                    ss = (float)((double)ss + d * d);
                },
                [&](float v, size_t j) {
                    if (v == 0.0f) v = __fadd_rn(c, __fmul_rn(gauss[(g0 + j) % GAUSSIAN_LEN], dv));
                    double d = (double)(v - mean);
                    ss = (float)((double)ss + d * d);
                });
    if (live) thr_win[idx] = __fadd_rn(mean, __fmul_rn(num_dev, sqrtf(ss / (float)window)));
}

// (6) per row: "delay the jump" -- a drop is postponed by one window, a rise anticipated by
//     one window; `diff` is scratch of n_win floats per row.                  libc.c:631-651
__global__ void tdt_smooth_kernel(float* __restrict__ thr_win, float* __restrict__ diff,
                                  size_t n_rows, size_t n_win)
{
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows || n_win < 2) return;
    float* w = thr_win + row * n_win;
    float* d = diff + row * n_win;
    for (size_t q = 0; q + 1 < n_win; ++q) d[q] = w[q + 1] - w[q];
    for (size_t q = 1; q < n_win; ++q) {
        if (d[q - 1] < 0.0f) w[q] -= d[q - 1];
        d[q - 1] = w[q] - w[q - 1];
    }
    for (size_t q = 0; q + 1 < n_win; ++q)
        if (d[q] > 0.0f) w[q] += d[q];
}

__device__ __forceinline__ size_t tdt_window_of(size_t i, size_t n, size_t shift, size_t n_win)
{
    if (i < shift) return 0;
    if (i >= n - shift) return n_win - 1;
    size_t q = i / shift;
    return q > n_win - 1 ? n_win - 1 : q;  // clamp: documented deviation (reference reads past the end)
}

// (7, optional) step-wise expansion to one threshold per sample.             libc.c:654-669
__global__ void tdt_expand_kernel(const float* __restrict__ thr_win, size_t n_rows, size_t n,
                                  size_t shift, size_t n_win, float* __restrict__ thr)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t row = blockIdx.y;
    if (i >= n) return;
    thr[row * n + i] = thr_win[row * n_win + tdt_window_of(i, n, shift, n_win)];
}

// Candidate extraction: every sample with cc > min(threshold, cap[row]) becomes one record
// (row, index, cc, threshold).  BPMF/similarity_search.py:629 (cap) and :231-232 (test).
// Records are appended in arbitrary order; the host sorts the (few thousand) survivors.
constexpr unsigned CAND_CHUNKS = 4;       // 16-byte loads per thread
__global__ __launch_bounds__(256) void cand_extract_kernel(
    const float* __restrict__ x, const float* __restrict__ thr_win, const float* __restrict__ row_cap,
    size_t n_rows, size_t n, size_t shift, size_t n_win, unsigned head_len, unsigned head_win,
    unsigned tail_start, unsigned tail_win, unsigned capacity,
    unsigned* __restrict__ count, int4* __restrict__ records)
{
    // 4 x 4 consecutive samples per thread (four 16-byte loads, all issued before the first test: a
    // workgroup streams 16 KB, not 4 -- 4.2 million 4 KB workgroups ran at 3.5 TB/s); the window of a
    // sample costs a division, so it is looked up for the first and the last of the four and only
    // recomputed in between when they differ (n < 2^31: 32-bit arithmetic)
    const size_t row = blockIdx.y;
    const unsigned nn = (unsigned)n, sh = (unsigned)shift, nw = (unsigned)n_win;
    // samples before head_len take window head_win, samples from tail_start on window tail_win (RMS
    // threshold: the first / last `shift` samples take the first / last window; MAD threshold: the
    // first half window and the last window - half samples repeat the ends of the indexed array)
    auto window_of = [&](unsigned i) -> unsigned {
        if (i < head_len) return head_win;
        if (i >= tail_start) return tail_win;
        const unsigned q = i / sh;
        return q > nw - 1 ? nw - 1 : q;   // clamp: documented deviation (tdt_window_of)
    };
    const float* xr = x + row * n;
    const float* tw = thr_win + row * n_win;
    const float cap = row_cap ? row_cap[row] : INFINITY;
    float v[CAND_CHUNKS][4];
    unsigned i0[CAND_CHUNKS], cnt[CAND_CHUNKS];
#pragma unroll
    for (int c = 0; c < CAND_CHUNKS; ++c) {
        // (< 2^32: n < 2^31 and the grid covers n rounded up to 4096)
        i0[c] = ((blockIdx.x * CAND_CHUNKS + c) * 256u + threadIdx.x) * 4u;
        cnt[c] = i0[c] >= nn ? 0u : (nn - i0[c] < 4u ? nn - i0[c] : 4u);
        v[c][0] = v[c][1] = v[c][2] = v[c][3] = -INFINITY;
        if (cnt[c] == 4) {
            const f32x4a v4 = *(const f32x4a*)(xr + i0[c]);
            v[c][0] = v4[0]; v[c][1] = v4[1]; v[c][2] = v4[2]; v[c][3] = v4[3];
        } else {
            for (unsigned e = 0; e < cnt[c]; ++e) v[c][e] = xr[i0[c] + e];
        }
    }
#pragma unroll
    for (int c = 0; c < CAND_CHUNKS; ++c) {
        if (cnt[c] == 0) continue;
        const unsigned w_first = window_of(i0[c]), w_last = window_of(i0[c] + cnt[c] - 1);
#pragma unroll
        for (unsigned e = 0; e < 4; ++e) {
            if (e >= cnt[c]) break;
            const unsigned w = w_first == w_last ? w_first : window_of(i0[c] + e);
            const float t = fminf(cap, tw[w]);
            if (v[c][e] > t) {
                unsigned slot = atomicAdd(count, 1u);
                if (slot < capacity)
                    records[slot] = make_int4((int)row, (int)(i0[c] + e), __float_as_int(v[c][e]), __float_as_int(t));
            }
        }
    }
}

struct TdtWorkspace {
    float* part;               // [rows, n_glob]
    unsigned long long* cnt;   // [rows, n_glob]
    unsigned long long* total; // [rows]
    float* centre;             // [rows]
    float* dev;                // [rows]
    float* diff;               // [rows, n_win]
    size_t bytes;
};

static TdtWorkspace tdt_carve(void* base, size_t rows, size_t n_glob, size_t n_win)
{
    TdtWorkspace ws;
    char* p = (char*)base;
    size_t o = 0;
    ws.part = (float*)(p + o);               o += align_up(rows * n_glob * sizeof(float), 256);
    ws.cnt = (unsigned long long*)(p + o);   o += align_up(rows * n_glob * 8, 256);
    ws.total = (unsigned long long*)(p + o); o += align_up(rows * 8, 256);
    ws.centre = (float*)(p + o);             o += align_up(rows * sizeof(float), 256);
    ws.dev = (float*)(p + o);                o += align_up(rows * sizeof(float), 256);
    ws.diff = (float*)(p + o);               o += align_up(rows * n_win * sizeof(float), 256);
    ws.bytes = o;
    return ws;
}

static int tdt_sizes(size_t n, size_t half_window, size_t shift, size_t* window, size_t* n_glob,
                     size_t* n_win)
{
    *window = 2 * half_window;
    // (shift == window + 1: an odd sliding window with overlap 0 -- the reference's size_t
    // arithmetic wraps to (n + 1) / shift windows, all inside the series: libc.c:528)
    if (*window == 0 || *window > 0x7fffffffull || shift == 0 || shift > *window + 1 || n < *window) return -1;
    *n_win = (n - (*window - shift)) / shift;
    *n_glob = n / *window;
    return *n_win >= 1 ? 0 : -1;
}

}  // namespace bpmf

using namespace bpmf;

extern "C" size_t bpmf_tdt_num_windows(size_t n, size_t half_window, size_t shift)
{
    size_t w, g, nw;
    return tdt_sizes(n, half_window, shift, &w, &g, &nw) ? 0 : nw;
}

extern "C" size_t bpmf_tdt_workspace_bytes(size_t n_rows, size_t n, size_t half_window, size_t shift)
{
    size_t w, g, nw;
    if (tdt_sizes(n, half_window, shift, &w, &g, &nw)) return 0;
    return tdt_carve(nullptr, n_rows, g, nw).bytes;
}

extern "C" int bpmf_tdt_rms_dev(const float* d_series, const float* d_gaussian, float num_dev,
                                size_t n_rows, size_t n, size_t half_window, size_t shift,
                                void* d_workspace, size_t workspace_bytes, bpmf_stream_t stream_,
                                float* d_thr_windows, float* d_threshold)
{
    hipStream_t stream = (hipStream_t)stream_;
    size_t window, n_glob, n_win;
    if (!d_series || !d_gaussian || !d_workspace || !d_thr_windows || n_rows == 0 ||
        tdt_sizes(n, half_window, shift, &window, &n_glob, &n_win)) {
        set_error("bpmf_tdt_rms_dev: bad argument (n=%zu half_window=%zu shift=%zu)", n,
                  half_window, shift);
        return -1;
    }
    TdtWorkspace ws = tdt_carve(d_workspace, n_rows, n_glob, n_win);
    if (workspace_bytes < ws.bytes) {
        set_error("bpmf_tdt_rms_dev: workspace too small (%zu < %zu)", workspace_bytes, ws.bytes);
        return -1;
    }
    const unsigned B = 64;
    auto blocks = [&](size_t work) { return dim3((unsigned)((work + B - 1) / B)); };
    tdt_glob_sum_kernel<<<blocks(n_rows * n_glob), dim3(B), 0, stream>>>(d_series, n_rows, n, window,
                                                                        n_glob, ws.part, ws.cnt);
    BPMF_LAUNCH_CHECK();
    tdt_glob_centre_kernel<<<blocks(n_rows), dim3(B), 0, stream>>>(ws.part, ws.cnt, n_rows, n_glob,
                                                                   ws.centre, ws.total);
    BPMF_LAUNCH_CHECK();
    tdt_glob_dev_kernel<<<blocks(n_rows * n_glob), dim3(B), 0, stream>>>(d_series, ws.centre, n_rows, n,
                                                                        window, n_glob, ws.part);
    BPMF_LAUNCH_CHECK();
    tdt_glob_std_kernel<<<blocks(n_rows), dim3(B), 0, stream>>>(ws.part, ws.total, n_rows, n_glob,
                                                                ws.dev);
    BPMF_LAUNCH_CHECK();
    tdt_window_kernel<<<blocks(n_rows * n_win), dim3(B), 0, stream>>>(
        d_series, d_gaussian, ws.centre, ws.dev, num_dev, n_rows, n, window, shift, n_win,
        d_thr_windows);
    BPMF_LAUNCH_CHECK();
    tdt_smooth_kernel<<<blocks(n_rows), dim3(B), 0, stream>>>(d_thr_windows, ws.diff, n_rows, n_win);
    BPMF_LAUNCH_CHECK();
    if (d_threshold) {
        if (n_rows > 65535) { set_error("bpmf_tdt_rms_dev: at most 65535 rows per call"); return -1; }
        tdt_expand_kernel<<<dim3((unsigned)((n + 255) / 256), (unsigned)n_rows), dim3(256), 0, stream>>>(
            d_thr_windows, n_rows, n, shift, n_win, d_threshold);
        BPMF_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int bpmf_extract_candidates_dev(const float* d_series, const float* d_thr_windows,
                                           const float* d_row_cap, size_t n_rows, size_t n,
                                           size_t half_window, size_t shift, uint32_t capacity,
                                           bpmf_stream_t stream_, uint32_t* d_count,
                                           bpmf_candidate* d_records)
{
    hipStream_t stream = (hipStream_t)stream_;
    size_t window, n_glob, n_win;
    if (!d_series || !d_thr_windows || !d_count || !d_records || n_rows == 0 || n_rows > 65535 ||
        n > 0x7fffffffull || tdt_sizes(n, half_window, shift, &window, &n_glob, &n_win)) {
        set_error("bpmf_extract_candidates_dev: bad argument");
        return -1;
    }
    BPMF_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(uint32_t), stream));
    cand_extract_kernel<<<dim3((unsigned)((n + 1024 * CAND_CHUNKS - 1) / (1024 * CAND_CHUNKS)), (unsigned)n_rows), dim3(256), 0, stream>>>(
        d_series, d_thr_windows, d_row_cap, n_rows, n, shift, n_win, (unsigned)shift, 0u,
        (unsigned)(n - shift), (unsigned)(n_win - 1), capacity, d_count, (int4*)d_records);
    BPMF_LAUNCH_CHECK();
    return 0;
}

// Validation windows of select_cc_indexes (BPMF/similarity_search.py:253-272): per detection, the samples strictly
// below a level in two adjacent windows of one CC row.
__global__ __launch_bounds__(256) void count_below_kernel(const float* __restrict__ x, size_t n, const int32_t* __restrict__ rows,
                                                          const long long* __restrict__ start, const int32_t* __restrict__ len_l,
                                                          const int32_t* __restrict__ len_r, const float* __restrict__ level,
                                                          int32_t* __restrict__ below)
{
    const size_t q = blockIdx.x;
    const float* row = x + (size_t)rows[q] * n + start[q];
    const int nl = len_l[q], nr = len_r[q];
    const float v = level[q];
    int cl = 0, cr = 0;
    for (int i = threadIdx.x; i < nl + nr; i += 256) {
        const bool b = row[i] < v;
        if (i < nl) cl += b;
        else cr += b;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cl += __shfl_xor(cl, o);
        cr += __shfl_xor(cr, o);
    }
    __shared__ int part[8];
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = cl; part[4 + (threadIdx.x >> 6)] = cr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        below[2 * q] = part[0] + part[1] + part[2] + part[3];
        below[2 * q + 1] = part[4] + part[5] + part[6] + part[7];
    }
}

extern "C" int bpmf_count_below_dev(const float* d_series, size_t n_rows, size_t n, size_t n_detections,
                                    const int32_t* d_rows, const int64_t* d_start, const int32_t* d_len_left,
                                    const int32_t* d_len_right, const float* d_level, bpmf_stream_t stream_,
                                    int32_t* d_below)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (n_detections == 0) return 0;
    if (!d_series || !d_rows || !d_start || !d_len_left || !d_len_right || !d_level || !d_below || n_rows == 0 ||
        n_detections > 0x7fffffffull) {
        set_error("bpmf_count_below_dev: bad argument");
        return -1;
    }
    // (the caller guarantees row < n_rows, start >= 0 and start + len_left + len_right <= n)
    count_below_kernel<<<dim3((unsigned)n_detections), dim3(256), 0, stream>>>(
        d_series, n, d_rows, (const long long*)d_start, d_len_left, d_len_right, d_level, d_below);
    BPMF_LAUNCH_CHECK();
    return 0;
}

// The same extraction against the window values of the MAD threshold (bpmf_tdt_mad_dev): sample i
// takes window min(clamp(i, half, n - (window - half) - 1) / shift, n_win - 1)
// (BPMF/similarity_search.py:1100-1112).
extern "C" int bpmf_extract_candidates_mad_dev(const float* d_series, const float* d_thr_windows,
                                               const float* d_row_cap, size_t n_rows, size_t n,
                                               size_t window, size_t shift, uint32_t capacity,
                                               bpmf_stream_t stream_, uint32_t* d_count,
                                               bpmf_candidate* d_records)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_series || !d_thr_windows || !d_count || !d_records || n_rows == 0 || n_rows > 65535 ||
        n > 0x7fffffffull || shift == 0 || window == 0 || n < window) {
        set_error("bpmf_extract_candidates_mad_dev: bad argument");
        return -1;
    }
    const size_t n_win = (n - window) / shift + 1;
    const size_t half = window / 2;
    if (n - (window - half) <= half) {   // a series of exactly one window: the indexed array is empty
        set_error("bpmf_extract_candidates_mad_dev: series too short (n=%zu window=%zu)", n, window);
        return -1;
    }
    const size_t tail_start = n - (window - half);
    const size_t head_win = std::min(half / shift, n_win - 1);
    const size_t tail_win = std::min((tail_start - 1) / shift, n_win - 1);
    BPMF_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(uint32_t), stream));
    cand_extract_kernel<<<dim3((unsigned)((n + 1024 * CAND_CHUNKS - 1) / (1024 * CAND_CHUNKS)), (unsigned)n_rows), dim3(256), 0, stream>>>(
        d_series, d_thr_windows, d_row_cap, n_rows, n, shift, n_win, (unsigned)half, (unsigned)head_win,
        (unsigned)tail_start, (unsigned)tail_win, capacity, d_count, (int4*)d_records);
    BPMF_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ running kurtosis ---
// BPMF/libc.c:11-53 (kurtosis; wrapper BPMF/clib.py:86-102), conventions of
// oracle/adjacent_oracle.c:kurtosis_cpu: kurto[n] (n >= W) from the W samples before n, mean as a
// sequential float sum, 2nd and 4th moments as float accumulators of double squares, written only
// where the variance exceeds 1e-6 (other samples keep the caller's zeros; W = 2 and 3 divide by
// (W - 2)(W - 3) = 0 and give +-inf / NaN exactly as the reference does, W = 1 never writes).  One thread per output
// sample; a workgroup stages its 256 + W input samples in LDS once.
namespace bpmf {
__global__ __launch_bounds__(256) void kurtosis_kernel(const float* __restrict__ x, int W,
                                                       long long length, float* __restrict__ k)
{
    extern __shared__ float win[];   // x[n0 - W .. n0 + 255]
    const long long ch = blockIdx.y;
    const long long n0 = (long long)W + (long long)blockIdx.x * 256;
    const float* xc = x + ch * length;
    for (int i = threadIdx.x; i < W + 256; i += 256) {
        const long long j = n0 - W + i;
        win[i] = j < length ? xc[j] : 0.0f;
    }
    __syncthreads();
    const long long n = n0 + threadIdx.x;
    if (n >= length) return;
    const float* w = win + threadIdx.x;   // samples n - W .. n - 1
    const float Wf = (float)W;
    float mean = 0.0f, m2 = 0.0f, m4 = 0.0f;
    for (int i = 0; i < W; ++i) mean += w[i];
    mean /= Wf;
    for (int i = 0; i < W; ++i) {
        const double d = (double)(w[i] - mean);
        m2 = (float)((double)m2 + d * d);
        m4 = (float)((double)m4 + (d * d) * (d * d));
    }
    m2 /= Wf;
    m4 /= Wf;
    if (m2 > 0.000001) {
        const double Wd = (double)Wf, m2d = (double)m2;
        k[ch * length + n] = (float)(1.0 / (double)((Wf - 2) * (Wf - 3)) *
                                     ((Wd * Wd - 1.0) * (double)m4 / (m2d * m2d) -
                                      3.0 * ((Wd - 1.0) * (Wd - 1.0))));
    }
}
}  // namespace bpmf

extern "C" int bpmf_kurtosis_dev(const float* d_signal, int W, size_t n_channels, size_t length,
                                 bpmf_stream_t stream_, float* d_kurto)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_signal || !d_kurto || W < 1 || n_channels == 0 || n_channels > 65535 ||
        (size_t)W > 32768) {
        bpmf::set_error("bpmf_kurtosis_dev: bad argument (W=%d, channels=%zu)", W, n_channels);
        return -1;
    }
    if (length <= (size_t)W) return 0;   // nothing to write
    const size_t n_out = length - (size_t)W;
    dim3 grid((unsigned)((n_out + 255) / 256), (unsigned)n_channels);
    bpmf::kurtosis_kernel<<<grid, dim3(256), ((size_t)W + 256) * sizeof(float), stream>>>(
        d_signal, W, (long long)length, d_kurto);
    BPMF_LAUNCH_CHECK();
    return 0;
}
