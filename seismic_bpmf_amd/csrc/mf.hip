// Matched-filter hot path for MI355X (gfx950): sliding normalised cross-correlation of T
// templates against S*C continuous channels, weighted network sum.
//
// Serves fast_matched_filter.matched_filter as called by the reference at
// BPMF/similarity_search.py:526-533 and BPMF/dataset.py:4818-4827 (the arithmetic itself
// is not in the reference tree).  Conventions = oracle/bpmf_oracle.c:mf_cpu, bit for bit.
//
// Kernel design: DESIGN.md section "MF" and the comment above mf_mfma_kernel.  In short: for
// one (template, channel) the numerators of 256 consecutive lags are one 16x16 MFMA tile of a
// Toeplitz band of the template against a strided view of the contiguous data window,
// evaluated with the exact-fp32 v_mfma_f32_16x16x4_f32 (a k-ordered fmaf chain; the band's
// zeros add exact zeros), so every numerator equals the scalar chain of the oracle.
#include "common.h"
#include "context.h"
#include "mf_split_api.h"
#include <system_error>
#include <thread>
#include <chrono>
#include <vector>
#include <mutex>
#include <utility>
#include <cstring>
#include <algorithm>
#include "../../include/bpmf_hip.h"

#include <cstdlib>

namespace bpmf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------ preparation ---

// Valid CC indices [first, last] of a template whose used channels have moveouts mv_min .. mv_max
// (first > last: none).
__device__ __forceinline__ int2 mf_lag_range(bool any, long long mv_min, long long mv_max, long long step,
                                             long long L, long long N, long long n_corr, int exclusive_last)
{
    int2 r = make_int2(1, 0);
    if (any && N >= L) {
        long long first = mv_min < 0 ? (-mv_min + step - 1) / step : 0;
        long long room = N - L - mv_max;
        // compat (option mf.compat_exclusive_last_lag): data offsets i * step < room only -- the loop
        // bound `i < stop_i`, stop_i = N - L - max_moveout, that upstream is recollected to use
        if (exclusive_last & 1) room -= 1;
        if (room >= 0) {
            long long last = room / step;
            if (last > n_corr - 1) last = n_corr - 1;
            if (first <= last) r = make_int2((int)first, (int)last);
        }
    }
    return r;
}

// Per template, one workgroup: r_t[t,s,c] = 1 / sqrtf(E_t), E_t = the fmaf chain of tmpl^2 over l
// ascending (the threads take the channels); then thread 0 writes the valid lag range [first, last]
// (first > last = empty) and the template's compact list of used channels, one int4 {channel, moveout,
// weight bits, r_t bits} per channel with w != 0, in channel order, closed by two {-1,..} sentinels: the
// main kernel walks it with one (prefetched) scalar load per channel instead of chasing weights /
// moveouts / norms through dependent loads.  (Rounds 1-2: two launches; an hour-long search of a
// handful of templates -- BASELINE configs[0] -- is dependent launches of which the main kernel takes
// 63 of 79 us, and a hipGraph of them is no faster: tools/probe_mf_graph.py.)
__global__ __launch_bounds__(64) void mf_prologue_kernel(const float* __restrict__ tmpl, const int* __restrict__ mv,
                                                         const float* __restrict__ w, int T, int n_ch, long long step,
                                                         long long L, long long N, long long n_corr, int exclusive_last,
                                                         int sqrt_norm, float* __restrict__ e_t, int2* __restrict__ range,
                                                         int4* __restrict__ chan_rec)
{
    const int t = blockIdx.x;
    for (int ch = threadIdx.x; ch < n_ch; ch += 64) {
        const float* x = tmpl + ((size_t)t * n_ch + ch) * (size_t)L;
        float acc = 0.0f;
        for (int l = 0; l < (int)L; ++l) acc = __fmaf_rn(x[l], x[l], acc);
        // reciprocal norm r_t (Inf for an all-zero template); mf.compat_sqrt_norm: the energy itself
        e_t[(size_t)t * n_ch + ch] = sqrt_norm ? acc : 1.0f / sqrtf(acc);
    }
    __syncthreads();       // (workgroup-scope release / acquire: thread 0 reads what the others stored)
    if (threadIdx.x != 0) return;
    long long mv_min = 0, mv_max = 0;
    bool any = false, seen = false;
    const bool all_channels = (exclusive_last & 2) != 0;   // option mf.compat_range_all_channels
    int4* rec = chan_rec + (size_t)t * (n_ch + 2);
    int n_used = 0;
    for (int ch = 0; ch < n_ch; ++ch) {
        const float wc = w[(size_t)t * n_ch + ch];
        if (wc == 0.0f && !all_channels) continue;
        long long m = mv[(size_t)t * n_ch + ch];
        if (!seen || m < mv_min) mv_min = m;
        if (!seen || m > mv_max) mv_max = m;
        seen = true;
        if (wc == 0.0f) continue;
        rec[n_used++] = make_int4(ch, (int)m, __float_as_int(wc),
                                  __float_as_int(((volatile const float*)e_t)[(size_t)t * n_ch + ch]));
        any = true;
    }
    rec[n_used] = make_int4(-1, 0, 0, 0);
    rec[n_used + 1] = make_int4(-1, 0, 0, 0);
    range[t] = mf_lag_range(any, mv_min, mv_max, step, L, N, n_corr, exclusive_last);
}

// Chunk-local prefix sums of data^2 in double (one thread per 1024-sample chunk).
// local[ch, n] = sum of squares of samples [chunk_start(n), n]; tot[ch, q] = chunk totals.
__global__ void mf_csum_local_kernel(const float* __restrict__ data, size_t n_ch, size_t N,
                                     size_t nq, double* __restrict__ local,
                                     double* __restrict__ tot, size_t q_lo, size_t q_cnt)
{
    // (chunks [q_lo, q_lo + q_cnt) of every channel: all of them, or the piece of a day that has just arrived)
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_ch * q_cnt) return;
    size_t ch = idx / q_cnt, q = q_lo + idx % q_cnt;
    size_t n0 = q * CSUM_CHUNK;
    size_t n1 = n0 + CSUM_CHUNK < N ? n0 + CSUM_CHUNK : N;
    const float* d = data + ch * N;
    double* lo = local + ch * N;
    double acc = 0.0;
    // 16-byte loads / 32-byte stores (dword-aligned vector types: ch * N need not be a
    // multiple of 4); the additions stay strictly sequential
    typedef float f32x4a __attribute__((ext_vector_type(4), aligned(4)));
    typedef double f64x2a __attribute__((ext_vector_type(2), aligned(8)));
    size_t n = n0;
    // one 128-byte line (8 x 16 B) per batch, the next batch in flight while this one is summed
    constexpr int NB = 8;
    f32x4a cur[NB], nxt[NB];
    const size_t nbatch = (n1 - n0) / (4 * NB);
    if (nbatch) {
#pragma unroll
        for (int i = 0; i < NB; ++i) cur[i] = *(const f32x4a*)(d + n + 4 * i);
    }
    for (size_t b = 0; b < nbatch; ++b, n += 4 * NB) {
        if (b + 1 < nbatch) {
#pragma unroll
            for (int i = 0; i < NB; ++i) nxt[i] = *(const f32x4a*)(d + n + 4 * NB + 4 * i);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const f32x4a v = cur[i];
            f64x2a o0, o1;
            double x = (double)v[0]; acc = acc + x * x; o0[0] = acc;  // squares are exact in double
            x = (double)v[1]; acc = acc + x * x; o0[1] = acc;
            x = (double)v[2]; acc = acc + x * x; o1[0] = acc;
            x = (double)v[3]; acc = acc + x * x; o1[1] = acc;
            *(f64x2a*)(lo + n + 4 * i) = o0;
            *(f64x2a*)(lo + n + 4 * i + 2) = o1;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) cur[i] = nxt[i];
    }
    for (; n + 4 <= n1; n += 4) {
        const f32x4a v = *(const f32x4a*)(d + n);
        f64x2a o0, o1;
        double x = (double)v[0]; acc = acc + x * x; o0[0] = acc;
        x = (double)v[1]; acc = acc + x * x; o0[1] = acc;
        x = (double)v[2]; acc = acc + x * x; o1[0] = acc;
        x = (double)v[3]; acc = acc + x * x; o1[1] = acc;
        *(f64x2a*)(lo + n) = o0;
        *(f64x2a*)(lo + n + 2) = o1;
    }
    for (; n < n1; ++n) {
        double v = (double)d[n];
        acc = acc + v * v;
        lo[n] = acc;
    }
    tot[ch * nq + q] = acc;
}

// Option mf.compat_sequential_csum: local[ch, n] = ONE sequential double chain of data^2 over the whole
// channel (what a plain CPU loop computes; off[] is zeroed, so csum = 0 + local).  One wave per channel:
// the 64 lanes stage 1024 samples in LDS with coalesced loads, lane 0 runs the dependent chain over them
// (8 640 000 dependent v_add_f64 per channel -- tens of milliseconds per day, whatever the channel count
// up to 256: a compatibility path, not a tuned one), the lanes store the 1024 sums coalesced.
__global__ __launch_bounds__(64) void mf_csum_sequential_kernel(const float* __restrict__ data, size_t N, size_t nq,
                                                                double* __restrict__ local, double* __restrict__ off)
{
    __shared__ float s_in[CSUM_CHUNK];
    __shared__ double s_out[CSUM_CHUNK];
    const size_t ch = blockIdx.x;
    const int lane = threadIdx.x;
    const float* d = data + ch * N;
    double* lo = local + ch * N;
    double acc = 0.0;
    for (size_t q = 0; q < nq; ++q) {
        const size_t n0 = q * CSUM_CHUNK;
        const size_t len = n0 + CSUM_CHUNK < N ? CSUM_CHUNK : N - n0;
        for (size_t i = lane; i < len; i += 64) s_in[i] = d[n0 + i];
        if (lane == 0) off[ch * nq + q] = 0.0;
        __syncthreads();
        if (lane == 0) {
            for (size_t i = 0; i < len; ++i) {
                const double v = (double)s_in[i];
                acc = acc + v * v;
                s_out[i] = acc;
            }
        }
        __syncthreads();
        for (size_t i = lane; i < len; i += 64) lo[n0 + i] = s_out[i];
        __syncthreads();
    }
}

// off[ch, q] = sequential sum of tot[ch, 0..q-1]  (one thread per channel).
__global__ void mf_csum_offsets_kernel(const double* __restrict__ tot, size_t n_ch, size_t nq,
                                       double* __restrict__ off, size_t q_lo, size_t q_hi)
{
    size_t ch = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= n_ch) return;
    // chunks [q_lo, q_hi): the chain continues where the previous piece of the day left it (same additions in
    // the same order as one pass over the whole channel)
    const double* tr = tot + ch * nq;
    double* orow = off + ch * nq;
    double acc = q_lo ? orow[q_lo - 1] + tr[q_lo - 1] : 0.0;
    // the additions are one dependent chain; the loads are not: fetch 16 totals at a time
    size_t q = q_lo;
    for (; q + 16 <= q_hi; q += 16) {
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = tr[q + i];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            orow[q + i] = acc;
            acc = acc + v[i];
        }
    }
    for (; q < q_hi; ++q) {
        orow[q] = acc;
        acc = acc + tr[q];
    }
}

// r_d[ch, j] = 1 / sqrtf(E_d),  E_d = (float)(csum[j+L] - csum[j]),
// csum[n] = off[chunk(n-1)] + local[n-1].
__global__ void mf_window_energy_kernel(const double* __restrict__ local,
                                        const double* __restrict__ off, size_t n_ch, size_t N,
                                        size_t nq, size_t L, size_t nwin, int sqrt_norm,
                                        float* __restrict__ e_d, size_t w_lo, size_t w_hi)
{
    // (windows [w_lo, w_hi): all of them, or those the piece of a day that has just arrived completes)
    size_t j = w_lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t ch = blockIdx.y;
    if (j >= w_hi) return;
    const double* lo = local + ch * N;
    const double* of = off + ch * nq;
    size_t nh = j + L - 1;
    double hi = of[nh / CSUM_CHUNK] + lo[nh];
    double low = 0.0;
    if (j > 0) low = of[(j - 1) / CSUM_CHUNK] + lo[j - 1];
    const float e = (float)(hi - low);
    e_d[ch * nwin + j] = sqrt_norm ? e : 1.0f / sqrtf(e);  // reciprocal norm r_d (mf.compat_sqrt_norm: the energy E_d)
}

// --------------------------------------------------------------- MFMA main kernel ---
//
// Tile algebra (v_mfma_f32_16x16x4_f32, exact fp32, k-ordered fmaf chain):
//   one 16x16 tile = 256 consecutive lags of one (template, channel):
//       Out[b][a] = sum_m A[b][m] * D[m][a],   lag = 16 a + b
//       A[b][m] = tmpl[m - b]   (16 x (L+15) Toeplitz band; zeros outside 0 <= m-b < L)
//       D[m][a] = data[x0 + 16 a + m]
//   A wave owns 4 such tiles (1024 lags) with 4 INDEPENDENT accumulators that share every A
//   operand: the matrix pipe always has an independent MFMA to issue (the 16x16x4 form has a
//   40-cycle dependent latency against a 32-cycle issue interval), and 5 LDS reads feed 4
//   MFMAs.  The band's zero rows cost (L+15)/L extra matrix work (6 % at L = 256).
//   A workgroup = 4 waves = 4096 consecutive lags of ONE template; it walks the S*C channels
//   with the weighted CC sums in registers.  Per channel: the next channel's data window and
//   band travel global -> registers during the MFMA loop, registers -> LDS (other buffer)
//   after it, one barrier per channel.
// LDS data layout: dw[x + (x >> 4)] = data[g0 + x]  (one pad float per 16) so that the 16
// tile columns of a B read (stride 16 floats) fall on 16 different banks.

// One channel's CC from its numerator and the two stored norms.  Default: the norms are RECIPROCALS
// (r_t = 1 / sqrtf(E_t), r_d = 1 / sqrtf(E_d)): cc = num * (r_t * r_d), 0 where the product is not below
// 1000.  SQRT_NORM (option mf.compat_sqrt_norm, the form upstream is recollected to use): the stored
// norms are the ENERGIES E_t and E_d, cc = num / sqrtf(E_t * E_d) where the product exceeds 1e-6, else 0
// -- an IEEE square root and an IEEE divide per channel and lag.
template <bool SQRT_NORM>
__device__ __forceinline__ float mf_cc_of(float num, float nt, float nd)
{
    if constexpr (SQRT_NORM) {
        const float den2 = nt * nd;
        return den2 > 1.0e-6f ? num / sqrtf(den2) : 0.0f;
    } else {
        const float nrm = nt * nd;
        return nrm < MAX_NORM ? num * nrm : 0.0f;
    }
}

constexpr int MF_THREADS = 256;
constexpr int MF_LAGS_PER_WAVE = 1024;
constexpr int MF_LAGS_PER_WG = 4096;
constexpr int MF_REC_TERMINATORS = 12;     // {-1, ..} records behind a template's used-channel records in LDS (mf_fused_prologue)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned vector

__device__ __forceinline__ int mf_pad(int x) { return x + (x >> 4); }

__host__ __device__ inline int mf_kpad(int L) { return (L + 15 + 15) / 16 * 16; }
__host__ __device__ inline int mf_band_len(int L) { return mf_kpad(L) + 16; }
__host__ __device__ inline int mf_window_len(int L) { return MF_LAGS_PER_WG - 16 + mf_kpad(L); }
// one LDS buffer = Toeplitz band + padded data window
__host__ __device__ inline int mf_buf_floats(int L)
{
    const int W = mf_window_len(L);
    return mf_band_len(L) + (W + (W >> 4) + 1);
}
// two buffers + slack for the operand prefetch that runs one trip past the end
__host__ inline size_t mf_lds_bytes(int L) { return ((size_t)2 * mf_buf_floats(L) + 64) * sizeof(float); }

// MAXR / MAXT: per-thread staging registers for the data window / the Toeplitz band
// (window <= 256 * MAXR floats, band <= 256 * MAXT floats).
// Workgroup -> (template, lag block).  The hardware deals consecutive workgroup ids round-robin
// to the 8 XCDs, each with its own L2.  With the plain order (template fastest) every XCD walks
// ALL lag blocks and streams the whole day of data and norms from HBM itself (8 x 4 GB at cfg2,
// plus what the ~60 resident workgroups per XCD re-fetch as they drift apart).  XCD-aware order:
// the (lag block, template) pairs -- template fastest -- are cut into 8 contiguous runs of equal
// length, one per XCD, so that the workgroups in flight on one L2 share one or two lag blocks'
// windows AND every XCD gets the same number of workgroups to within one.  (Rounds 1-3 gave every XCD
// ceil(blocks / 8) whole lag blocks: on an hour-long series -- 44 blocks -- XCD 7 then owned 2 blocks
// where the others owned 6, and the launch lasted 6 / 5.5 of what it had to: 0.78 -> 0.83 of the peak
// for 256 templates of 256 samples on configs[0]'s hour, tools/probe_mf_ntile_T.py, round 4.)
// The grid is 8 * ceil(blocks * T / 8) workgroups; the ones past the last pair exit at once.
__device__ __forceinline__ bool mf_tile_of_block(unsigned bid, int T, int n_lag_blocks, int& t,
                                                 long long& lag_block)
{
    const unsigned xcd = bid & 7u, i = bid >> 3;
    const unsigned total = (unsigned)n_lag_blocks * (unsigned)T;       // (< 2^31: checked on the host)
    const unsigned per_xcd = (total + 7u) >> 3;
    const unsigned flat = xcd * per_xcd + i;
    t = (int)(flat % (unsigned)T);
    lag_block = (long long)(flat / (unsigned)T);
    return i < per_xcd && flat < total;
}

// Zero-padded staging loads.  A raw buffer load returns 0 for a lane whose byte offset lies outside
// [0, num_records) -- which is how the windows get their zeros before the start and past the end of a
// trace -- but on gfx950 that check is per lane only while the instruction carries NO immediate
// offset: with one, an aligned group of 4 lanes whose (voffset + immediate) values straddle the
// 32-bit wrap (some lanes before sample 0, some at samples 0..2) comes back as zeros for all 4 lanes
// (tools/ubench/buffer_neg.hip, profiles/r02_buffer_wrap.txt).  Rounds 1 and early 2 lost the first
// 1-3 samples of a trace that way, at the first valid lags of a template whose most negative
// moveout is not a multiple of 4.  So: immediates only when every voffset is non-negative (`first`,
// the window's first sample, is uniform over the wave / workgroup), otherwise one register offset
// per load.
template <int NR, int STRIDE>
__device__ __forceinline__ void mf_stage_rows(float (&rd)[NR], __amdgpu_buffer_rsrc_t rs,
                                              long long first, int idx)
{
    const unsigned o = (unsigned)((first + idx) * 4);   // 32-bit byte offset, wraps like the hardware's
    if (first >= 0) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
            rd[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(o + STRIDE * 4u * r), 0, 0));
    } else {
        unsigned oo = o;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            asm volatile("" : "+v"(oo));           // keep the constant out of the immediate field
            rd[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)oo, 0, 0));
            oo += STRIDE * 4u;
        }
    }
}
// band row r holds template samples idx + STRIDE r - 15: negative only in row 0
template <int NR, int STRIDE>
__device__ __forceinline__ void mf_stage_band(float (&rt)[NR], __amdgpu_buffer_rsrc_t rs, int idx)
{
    int o0 = (idx - 15) * 4;
    asm volatile("" : "+v"(o0));
    rt[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o0, 0, 0));
#pragma unroll
    for (int r = 1; r < NR; ++r)
        rt[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                              rs, idx * 4 + (STRIDE * r - 15) * 4, 0, 0));
}

template <bool NETWORK_SUM, int MAXR, int MAXT, bool STEP1, bool SQRT_NORM = false>
__global__ __launch_bounds__(MF_THREADS, 2) void mf_mfma_kernel(
    const float* __restrict__ tmpl, const int4* __restrict__ chan_rec,
    const float* __restrict__ data, const float* __restrict__ e_d,
    const int2* __restrict__ range, int L, long long N, int T, int n_ch, long long n_corr, int step,
    float* __restrict__ out, int n_lag_blocks, int prio, int lag_block0)
{
    extern __shared__ float smem[];
    const int Kpad = mf_kpad(L);
    const int tp_len = mf_band_len(L);
    const int W = mf_window_len(L);
    const int buf_floats = mf_buf_floats(L);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int a = lane & 15;   // tile column (and band row for the A operand)
    const int kq = lane >> 4;  // k index of the operands / row group of the results

    // XCD-aware workgroup order (mf_tile_of_block)
    int t;
    long long lag_block;
    if (!mf_tile_of_block(blockIdx.x, T, n_lag_blocks, t, lag_block)) return;
    lag_block += lag_block0;          // (a launch over a RANGE of lag blocks: a day still arriving from the host)
    const long long lag0 = lag_block * MF_LAGS_PER_WG;
    // `range` holds CC indices; the kernel works on data-sample offsets (lag = index * step) and
    // simply skips the offsets that are not multiples of step
    const int2 rgi = range[t];
    const int2 rg = make_int2(rgi.x * step, rgi.y * step);
    const long long nwin = N - L + 1;

    f32x4 sum[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) sum[u] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    // (an empty range is stored as first > last: without the first test a workgroup that straddles
    // both ends of it would count as valid and its norm loads -- "lag + 3 >= first && lag <= last"
    // holds for lags last-3..last -- would run at lag + moveout far outside the table)
    const bool wg_valid = rg.x <= rg.y && !(lag0 > rg.y || lag0 + MF_LAGS_PER_WG - 1 < rg.x);
    const bool wg_inside = lag0 >= rg.x && lag0 + MF_LAGS_PER_WG - 1 <= rg.y;  // no range tests needed
    // result (tile u, register r) of this lane is lag  lag_w + 256 u + r
    const long long lag_w = lag0 + (long long)wv * MF_LAGS_PER_WAVE + 16 * a + 4 * kq;

    if (wg_valid) {
        const int a_base = 15 - a + kq;
        const int b_base = 1088 * wv + 17 * a + kq;
        const int4* __restrict__ recs = chan_rec + (size_t)t * (n_ch + 2);

        float rd[MAXR], rt[MAXT];
        // Staging loads go through buffer descriptors: an offset outside [0, bytes) -- a
        // window sample before the start / past the end of the trace, or a band row outside
        // the template -- returns 0 from the hardware bounds check, so the zero padding costs
        // no address clamping or select (see mf_stage_rows for the one trap in that).
        auto issue_stage = [&](int ch, int mvc) {
            const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(data + (size_t)ch * (size_t)N), 0, (int)(N * 4), 0x00020000);
            mf_stage_rows<MAXR, MF_THREADS>(rd, rs_d, lag0 + mvc, tid);
            const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(tmpl + ((size_t)t * n_ch + ch) * (size_t)L), 0, L * 4, 0x00020000);
            mf_stage_band<MAXT, MF_THREADS>(rt, rs_t, tid);
        };
        auto write_stage = [&](float* tp, float* dw) {
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                const int x = tid + MF_THREADS * r;
                if (x < W) dw[mf_pad(x)] = rd[r];
            }
#pragma unroll
            for (int r = 0; r < MAXT; ++r) {
                const int x = tid + MF_THREADS * r;
                if (x < tp_len) tp[x] = rt[r];
            }
        };
        // records of the current / next channel are in SGPRs; the one after is fetched during
        // the K loop, so no scalar-memory latency sits on the per-channel critical path
        int4 rec = recs[0];
        int4 rec1 = recs[1];
        int ri = 0;
        if (rec.x >= 0) issue_stage(rec.x, rec.y);
        int buf = 0;
        while (rec.x >= 0) {
            const int ch = rec.x;
            float* tp = smem + buf * buf_floats;  // tp[15 + l] = tmpl[l], zeros around
            float* dw = tp + tp_len;              // dw[pad(x)] = data[g0 + x]
            write_stage(tp, dw);
            // One barrier per channel: the other buffer is only rewritten after every wave
            // has passed this point, i.e. after it finished reading it.
            __syncthreads();
            const float w = __int_as_float(rec.z);
            const int mvc = rec.y;
            const float et = __int_as_float(rec.w);
            const int4 rec2 = recs[ri + 2];
            // window energies of this lane's 4 x 4 lags: in flight during the MFMA loop
            const float* edc = e_d + (size_t)ch * (size_t)nwin;
            f32x4 ed[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long lag = lag_w + 256 * u;
                // all four lags of the group inside the valid range -> one 16-byte load
                // One 16-byte load per group of 4 lags.  A group that straddles an end of the
                // valid range reads up to 3 floats outside this channel's row of norms -- the
                // neighbouring row or the slack around the array -- and those lanes are masked in
                // the epilogue (`ok`).
                if (wg_inside || (lag + 3 >= rg.x && lag <= rg.y)) {
                    ed[u] = *(const f32x4u*)(edc + lag + mvc);
                } else {
                    ed[u] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
            if (rec1.x >= 0) issue_stage(rec1.x, rec1.y);

            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            // K loop: 16 band rows = 4 k-steps per trip; the operands of k-step j+2 are
            // requested before the MFMAs of k-step j issue (4 rotating register slots).
            const int nq = Kpad >> 4;
            // K loop: 16 band rows = 4 k-steps per trip.  The 5 operands of k-step j+2 are
            // requested from LDS before the 4 MFMAs of k-step j issue, in 4 rotating register
            // slots.  The reads are inline asm with COUNTED waits: hipcc's own waitcnt insertion
            // falls back to lgkmcnt(0) at the loop back-edge, which drains the operand pipeline
            // once per trip (a ~100-cycle bubble every 512 cycles of matrix work).  In steady
            // state 15 reads are in flight when k-step j needs its operands, the 10 newest belong
            // to k-steps j+1 and j+2, and LDS returns in order: s_waitcnt lgkmcnt(10).
            // sched_barrier keeps the compiler from moving MFMAs across the asm waits.
            const unsigned a_addr = (unsigned)(size_t)(tp + a_base) ;      // LDS byte addresses
            const unsigned b_addr = (unsigned)(size_t)(dw + b_base);
            unsigned ap = a_addr, bp = b_addr;
            float sa[4], sb[4][4];
#define MF_LDS_READ(dst, addr, off) \
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define MF_MFMA(slot, u) \
    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[slot], sb[slot][u], acc[u], 0, 0, 0)
#define MF_REQ(slot, aoff, boff)                                    \
    MF_LDS_READ(sa[slot], ap, (aoff));                               \
    MF_LDS_READ(sb[slot][0], bp, (boff));                            \
    MF_LDS_READ(sb[slot][1], bp, (boff) + 1088);                     \
    MF_LDS_READ(sb[slot][2], bp, (boff) + 2176);                     \
    MF_LDS_READ(sb[slot][3], bp, (boff) + 3264)
// One k-step: the 4 MFMAs of slot `cur` with the 5 operand reads of slot `req` (two k-steps
// ahead) issued between them (a wave waits ~32 cycles at every MFMA for the matrix pipe; a read
// placed there issues for free).  On entry the reads of `cur` and of the following k-step are
// outstanding (10) and LDS returns in order: lgkmcnt(5) = "cur has landed".
#define MF_STEP(cur, req, aoff, boff)                                \
    asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");               \
    __builtin_amdgcn_sched_barrier(0);                               \
    MF_MFMA(cur, 0);                                                 \
    __builtin_amdgcn_sched_barrier(0);                               \
    MF_LDS_READ(sa[req], ap, (aoff));                                \
    __builtin_amdgcn_sched_barrier(0);                               \
    MF_MFMA(cur, 1);                                                 \
    __builtin_amdgcn_sched_barrier(0);                               \
    MF_LDS_READ(sb[req][0], bp, (boff));                             \
    __builtin_amdgcn_sched_barrier(0);                               \
    MF_MFMA(cur, 2);                                                 \
    __builtin_amdgcn_sched_barrier(0);                               \
    MF_LDS_READ(sb[req][1], bp, (boff) + 1088);                      \
    __builtin_amdgcn_sched_barrier(0);                               \
    MF_MFMA(cur, 3);                                                 \
    __builtin_amdgcn_sched_barrier(0);                               \
    MF_LDS_READ(sb[req][2], bp, (boff) + 2176);                      \
    MF_LDS_READ(sb[req][3], bp, (boff) + 3264);                      \
    __builtin_amdgcn_sched_barrier(0)
            // (a scalar load may still be in flight here -- the next-but-one channel record; it
            // only makes the counted waits stricter, never laxer: LDS returns in order)
            __builtin_amdgcn_sched_barrier(0);
            if (prio) __builtin_amdgcn_s_setprio(0);
            MF_REQ(0, 0, 0);
            MF_REQ(1, 16, 16);
            for (int q = 0; q < nq; ++q) {
                MF_STEP(0, 2, 32, 32);
                MF_STEP(1, 3, 48, 48);
                MF_STEP(2, 0, 64, 68);   // requests k-step 0 of the next trip (past the end: slack)
                MF_STEP(3, 1, 80, 84);
                ap += 64;
                bp += 68;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // drain the two k-steps read ahead
            __builtin_amdgcn_sched_barrier(0);
            // (option mf.boundary_prio, see mf_mfma_wave_kernel.  Measured without effect here -- L = 400 / 800 /
            // 1024: 85.7 / 90.6 / 91.6 % of the peak at every level -- the waves of a workgroup reach their channel
            // boundary together, behind one barrier; kept so that the option means the same in both kernels)
            if (prio == 1) __builtin_amdgcn_s_setprio(1);
            else if (prio == 2) __builtin_amdgcn_s_setprio(2);
            else if (prio == 3) __builtin_amdgcn_s_setprio(3);
#undef MF_LDS_READ
#undef MF_MFMA
#undef MF_REQ
#undef MF_STEP

            if (NETWORK_SUM && STEP1 && wg_inside) {
                // every lag of this workgroup is inside the template's valid range: no range tests
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float cc = mf_cc_of<SQRT_NORM>(acc[u][r], et, ed[u][r]);
                        sum[u][r] = __fmaf_rn(w, cc, sum[u][r]);
                    }
                }
            } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long lag = lag_w + 256 * u + r;
                    // STEP1 builds carry no division at all; otherwise 32-bit (lags < 2^31)
                    const bool ok = lag >= rg.x && lag <= rg.y && (STEP1 || (unsigned)lag % (unsigned)step == 0);
                    float cc = 0.0f;
                    if (ok) {
                        cc = mf_cc_of<SQRT_NORM>(acc[u][r], et, ed[u][r]);
                        if (!NETWORK_SUM)
                            out[((size_t)t * n_corr + (STEP1 ? lag : (long long)((unsigned)lag / (unsigned)step))) * n_ch + ch] = cc;
                    }
                    if (NETWORK_SUM) sum[u][r] = __fmaf_rn(w, cc, sum[u][r]);
                }
            }
            }
            rec = rec1;
            rec1 = rec2;
            ++ri;
            buf ^= 1;
        }
    }
    if (NETWORK_SUM) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long lag = lag_w + 256 * u;
            float* dst = out + (size_t)t * n_corr + lag;
            if (STEP1 && lag + 3 < n_corr) {
                *(f32x4u*)dst = sum[u];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned smp = (unsigned)(lag + r);
                    const long long i = STEP1 ? (long long)smp : (long long)(smp / (unsigned)step);
                    if ((STEP1 || smp % (unsigned)step == 0) && i < n_corr) out[(size_t)t * n_corr + i] = sum[u][r];
                }
            }
        }
    }
}

// What mf_prologue_kernel computes for template t, by the 256 threads of a workgroup of the wave kernel
// itself and into LDS (FUSED variants: small problems, where a second launch costs as much as the
// correlation; option mf.fused_prologue).  Thread c owns channel c (n_ch <= 256): its weight, its moveout
// and -- the same sequential fmaf chain -- its template's energy; the used channels are compacted in
// channel order (ballot + per-wave counts) into l_rec[0 .. n_used), two terminators behind them; the lag
// range of the template comes back in registers.  l_part: 12 ints of LDS scratch.  Two barriers: every
// thread of the workgroup calls this.
template <bool SQRT_NORM>
__device__ __forceinline__ int2 mf_fused_prologue(const float* __restrict__ tmpl, const int* __restrict__ mv,
                                                  const float* __restrict__ w, int t, int n_ch, long long step, int L,
                                                  long long N, long long n_corr, int exclusive_last,
                                                  int4* l_rec, int* l_part)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float wc = 0.0f, nt = 0.0f;
    int m = 0;
    bool used = false;
    if (tid < n_ch) {
        wc = w[(size_t)t * n_ch + tid];
        used = !(wc == 0.0f);
    }
    const bool all_channels = (exclusive_last & 2) != 0;   // option mf.compat_range_all_channels
    if (tid < n_ch && (used || all_channels)) m = mv[(size_t)t * n_ch + tid];
    if (used) {
        const float* x = tmpl + ((size_t)t * n_ch + tid) * (size_t)L;
        float acc = 0.0f;
        int l = 0;
        for (; l + 16 <= L; l += 16) {             // 4 loads in flight per trip; the chain itself stays in sample order
            const f32x4u v0 = *(const f32x4u*)(x + l), v1 = *(const f32x4u*)(x + l + 4);
            const f32x4u v2 = *(const f32x4u*)(x + l + 8), v3 = *(const f32x4u*)(x + l + 12);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __fmaf_rn(v0[i], v0[i], acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __fmaf_rn(v1[i], v1[i], acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __fmaf_rn(v2[i], v2[i], acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __fmaf_rn(v3[i], v3[i], acc);
        }
        for (; l < L; ++l) acc = __fmaf_rn(x[l], x[l], acc);
        nt = SQRT_NORM ? acc : 1.0f / sqrtf(acc);
    }
    const unsigned long long mask = __ballot(used);
    const int idx = __popcll(mask & ((1ull << lane) - 1ull));
    const bool ranged = used || (all_channels && tid < n_ch);
    int mn = ranged ? m : 0x7fffffff, mx = ranged ? m : (int)0x80000000;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o));
        mx = max(mx, __shfl_xor(mx, o));
    }
    if (lane == 0) {
        l_part[wv] = __popcll(mask);
        l_part[4 + wv] = mn;
        l_part[8 + wv] = mx;
    }
    __syncthreads();
    int base = 0, n_used = 0, mv_min_i = 0x7fffffff, mv_max_i = (int)0x80000000;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = l_part[i];
        if (i < wv) base += c;
        n_used += c;
        mv_min_i = min(mv_min_i, l_part[4 + i]);
        mv_max_i = max(mv_max_i, l_part[8 + i]);
    }
    if (used) l_rec[base + idx] = make_int4(tid, m, __float_as_int(wc), __float_as_int(nt));
    if (tid < MF_REC_TERMINATORS) l_rec[n_used + tid] = make_int4(-1, 0, 0, 0);   // (12: the channel-split variant looks 8 records ahead)
    __syncthreads();
    return mf_lag_range(n_used > 0, mv_min_i, mv_max_i, step, L, N, n_corr, exclusive_last);
}

// ------------------------------------------------- MFMA kernel, independent waves ---
// Same tile algebra and arithmetic as mf_mfma_kernel, different ownership of LDS: every wave
// stages ITS OWN data window (1008 + Kpad floats) and its own copy of the band, so no wave ever
// waits for another one -- there is no barrier in the channel loop -- and, because only the
// owning wave reads a buffer, it can be refilled in place after the K loop (LDS ops of one wave
// execute in order): single-buffered, 6.6 KB per wave instead of 9.9 KB.  The price is the
// 256-float overlap between neighbouring waves' windows (25 % more staging traffic from L2).
// Used for L <= 257 (window 1280 floats = 20 staging registers per lane).
// NTILE: 16x16 tiles (of 256 lags) per wave -- 4 for a day-long search (the A operand of a k-step
// feeds 4 MFMAs); 2 or 1 when the whole problem has too few waves to fill the chip otherwise (an
// hour-long series with a handful of templates, BASELINE configs[0]: 704 waves of 1024 lags on 1024
// SIMDs, each wave alone with its serial work; at 256 lags per wave 2816 waves share the matrix pipes
// three to a SIMD).  The hand-placed operand reads and their counted waits follow NTILE.
// Cycle accounting (tools/phase/build_phase_lib.py builds a second library with -DBPMF_PHASE_CYCLES; the
// shipping library carries none of it): s_memtime at the phase boundaries of a channel, summed per wave
// over its channels and, at the end of the kernel, over all waves of the launch into g_mf_phase[{staging
// writes, norm loads + staging issue, K loop, epilogue, channels, waves}] -- read and reset through
// bpmf_phase_read_mf (exported by such a build only; tools/phase/mf_phase.py).
#ifdef BPMF_PHASE_CYCLES
__device__ unsigned long long g_mf_phase[8];
#define MF_PHASE_DECL unsigned long long ph_last_ = 0, ph_acc_[4] = {0, 0, 0, 0}; unsigned ph_n_ = 0;
#define MF_PHASE_START() asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_last_) :: "memory")
#define MF_PHASE(i)                                                                                \
    do {                                                                                           \
        unsigned long long t_;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");              \
        ph_acc_[i] += t_ - ph_last_;                                                               \
        ph_last_ = t_;                                                                             \
    } while (0)
#else
#define MF_PHASE_DECL
#define MF_PHASE_START() do {} while (0)
#define MF_PHASE(i) do {} while (0)
#endif

// FUSED (option mf.fused_prologue, NTILE < 4 and n_ch <= 256): no mf_prologue_kernel ran -- chan_rec and range
// are not read, every workgroup works its template's channel records and lag range out itself
// (mf_fused_prologue, from mv / wts) and walks the records in LDS.
// CSPLIT (round 5; tiny problems only: one tile per wave, fused prologue, network sum, step 1): the four waves of a
// workgroup take the SAME 256 lags and every fourth used channel each, leave their channels' CC values in LDS, and
// after one barrier the 256 threads run the weighted sum over the channels in channel order -- the same fmaf
// chain, so the same bits -- one lag each.  Four times the waves for the same arithmetic: BASELINE configs[0]
// (4 templates x 24 channels x one hour) has 2 816 tiles of 256 lags for 4 096 wave slots, and a wave that walks
// all 24 channels alone spends 4/5 of its time on the serial work around 36 MFMAs.  Measured (tools/probe_mf_csplit.py,
// profiles/r05_mf_channel_split.txt): one template on an hour 25.7 -> 16.5 us (L = 64) ... 47.2 -> 30.1 us (L = 256), two
// templates -15 ... -20 %, FOUR templates (configs[0]: 2 816 waves already) +3 ... +8 % -- more waves only contend there --
// so the variant is taken up to 2 048 waves of 256 lags (option mf.channel_split).
template <bool NETWORK_SUM, int MAXR, int MAXT, bool STEP1, int NTILE = 4, bool SQRT_NORM = false, bool FUSED = false, bool CSPLIT = false>
__global__ __launch_bounds__(MF_THREADS, 4) void mf_mfma_wave_kernel(
    const float* __restrict__ tmpl, const int4* __restrict__ chan_rec,
    const float* __restrict__ data, const float* __restrict__ e_d,
    const int2* __restrict__ range, int L, long long N, int T, int n_ch, long long n_corr, int step,
    float* __restrict__ out, int n_lag_blocks, int prio, const int* __restrict__ mv,
    const float* __restrict__ wts, int exclusive_last, int lag_block0)
{
    extern __shared__ float smem[];
    const int Kpad = mf_kpad(L);
    const int tp_len = mf_band_len(L);
    static_assert(NTILE == 4 || NTILE == 2 || NTILE == 1, "tiles per wave");
    static_assert(!CSPLIT || (NTILE == 1 && FUSED && NETWORK_SUM && STEP1), "channel split: the tiny-problem variant only");
    constexpr int LAGS_W = 256 * NTILE, LAGS_WG = CSPLIT ? LAGS_W : 4 * LAGS_W;
    constexpr int REC_STRIDE = CSPLIT ? 4 : 1;       // records between a wave's consecutive channels
    const int Ww = LAGS_W - 16 + Kpad;          // this wave's window
    // FULLW (the 1- and 2-tile variants): the wave's buffer has room for ALL its staging registers (64 MAXR window
    // floats), whatever the template length -- see write_stage
    constexpr bool FULLW = NTILE < 4;
    const int Wbuf = FULLW ? 64 * MAXR : Ww;
    const int wave_floats = tp_len + (Wbuf + 2 * (Wbuf >> 4) + 2 + 63) / 64 * 64 + 64;   // (+ 64: whole 64-lane chunks are written)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: lag0 etc. live in SGPRs
    const int a = lane & 15;
    const int kq = lane >> 4;

    int t;
    long long lag_block;
    // (Letting a workgroup walk several consecutive lag blocks of its template, so that the template
    // rows come from L2 after the first block, was measured in round 2 -- BPMF_MF_NSUB in commit
    // "MF: lag blocks per workgroup experiment": FETCH_SIZE 88.4 / 82.0 / 81.3 / 92.2 M KiB and
    // 85.7 / 85.5 / 85.1 / 84.9 % of the fp32 peak for 1 / 2 / 4 / 8 blocks, profiles/r02_mf_nsub.txt;
    // the loop also cost 7 spilled registers.  One block per workgroup.)
    if (!mf_tile_of_block(blockIdx.x, T, n_lag_blocks, t, lag_block)) return;
    lag_block += lag_block0;          // (a launch over a RANGE of lag blocks: a day still arriving from the host)
    int2 rgi;
    int4* l_rec = nullptr;
    if constexpr (FUSED) {
        int* l_part = (int*)(smem + 4 * wave_floats + 64);        // behind the waves' buffers and their read slack
        l_rec = (int4*)(l_part + 16);
        const int2 r = mf_fused_prologue<SQRT_NORM>(tmpl, mv, wts, t, n_ch, (long long)step, L, N, n_corr, exclusive_last,
                                                    l_rec, l_part);
        rgi = make_int2(__builtin_amdgcn_readfirstlane(r.x), __builtin_amdgcn_readfirstlane(r.y));
    } else {
        rgi = range[t];
    }
    // CSPLIT: CC values of the workgroup's 256 lags, one row per used channel, behind the records
    float* exch = nullptr;
    int n_used = 0;
    if constexpr (CSPLIT) {
        const int* l_part = (const int*)(smem + 4 * wave_floats + 64);
        n_used = l_part[0] + l_part[1] + l_part[2] + l_part[3];
        exch = (float*)(l_rec + n_ch + MF_REC_TERMINATORS);
    }
    const long long lag0 = lag_block * LAGS_WG + (CSPLIT ? 0 : (long long)wv * LAGS_W);
    const int2 rg = make_int2(rgi.x * step, rgi.y * step);  // CC indices -> data-sample offsets
    const long long nwin = N - L + 1;

    f32x4 sum[NTILE];
#pragma unroll
    for (int u = 0; u < NTILE; ++u) sum[u] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    const bool wave_valid = rg.x <= rg.y && !(lag0 > rg.y || lag0 + LAGS_W - 1 < rg.x);   // empty range: first > last
    const bool wave_inside = lag0 >= rg.x && lag0 + LAGS_W - 1 <= rg.y;
    const long long lag_w = lag0 + 16 * a + 4 * kq;
    MF_PHASE_DECL

    if (wave_valid) {
        float* tp = smem + wv * wave_floats;  // tp[15 + l] = tmpl[l], zeros around
        float* dw = tp + tp_len;              // dw[pad(x)] = data[g0 + x]
        const int a_base = 15 - a + kq;
        const int b_base = 18 * a + kq;  // window padded 2 floats per 16: conflict-free B reads
        const int4* __restrict__ recs = FUSED ? nullptr : chan_rec + (size_t)t * (n_ch + 2);
        // FUSED: the records are in LDS; one broadcast read, and the four fields back into SGPRs (the channel
        // number goes into buffer descriptors, the tests on it are scalar branches).  The read is issued in
        // front of a K loop's operand reads and lands before them (one wave's LDS operations return in order).
        auto ld_rec = [&](int i) -> int4 {
            if constexpr (FUSED) {
                const int4 v = l_rec[i];
                return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y),
                                 __builtin_amdgcn_readfirstlane(v.z), __builtin_amdgcn_readfirstlane(v.w));
            } else {
                return recs[i];
            }
        };

        float rd[MAXR], rt[MAXT];
        auto issue_stage = [&](int ch, int mvc) {
            const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(data + (size_t)ch * (size_t)N), 0, (int)(N * 4), 0x00020000);
            mf_stage_rows<MAXR, 64>(rd, rs_d, lag0 + mvc, lane);
            const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(tmpl + ((size_t)t * n_ch + ch) * (size_t)L), 0, L * 4, 0x00020000);
            mf_stage_band<MAXT, 64>(rt, rs_t, lane);
        };
        // Staging registers -> LDS in WHOLE 64-lane chunks, as many as the band / the window need (wave-
        // uniform counts, one scalar jump): the lanes of the last chunk past the end of the band land in
        // the first floats of the window, which is written afterwards; those past the end of the window in
        // the 64 floats of slack behind it.  What they carry is real data or the zero fill of the buffer
        // load, and nobody reads it.  A per-lane `if (x < Ww)` cost 160 instructions per channel (exec masks
        // per register) -- and an instruction of a wave that is NOT in its K loop waits 40-60 cycles for
        // an issue slot while the other three waves of the SIMD stream MFMAs: 4 500-8 800 cycles per
        // channel and wave for L < 241 against 860 on the full-window path (s_memtime around the stage,
        // profiles/r03_mf_phase_cycles.txt).
        const int rd_chunks = (Ww + 63) >> 6, rt_chunks = (tp_len + 63) >> 6;
        auto write_stage = [&]() {
            if constexpr (FULLW) {
                // The small problems' variants write EVERY staging register, needed or not: the band chunks past
                // the end of the band land in the window, which is written after them, the window chunks past
                // its end in the room the buffer has for them.  13 / 17 stores without a branch instead of the
                // two jump tables below, which the compiler turns into ~50 scalar branches and ~100 flag moves
                // per channel -- a fifth of what a wave of those shapes executes between two K loops.
#pragma unroll
                for (int r = 0; r < MAXT; ++r) tp[lane + 64 * r] = rt[r];
#pragma unroll
                for (int r = 0; r < MAXR; ++r) {
                    const int x = lane + 64 * r;
                    dw[x + 2 * (x >> 4)] = rd[r];
                }
                return;
            }
#define MF_WT(r) case (r) + 1: if constexpr ((r) < MAXT) tp[lane + 64 * ((r) < MAXT ? (r) : 0)] = rt[(r) < MAXT ? (r) : 0]; [[fallthrough]];
            switch (rt_chunks) {
                MF_WT(4) MF_WT(3) MF_WT(2) MF_WT(1) MF_WT(0)
                default: break;
            }
#undef MF_WT
#define MF_WD(r) case (r) + 1: if constexpr ((r) < MAXR) { const int x = lane + 64 * (r); dw[x + 2 * (x >> 4)] = rd[(r) < MAXR ? (r) : 0]; } [[fallthrough]];
            switch (rd_chunks) {
                MF_WD(19) MF_WD(18) MF_WD(17) MF_WD(16) MF_WD(15) MF_WD(14) MF_WD(13) MF_WD(12) MF_WD(11) MF_WD(10)
                MF_WD(9) MF_WD(8) MF_WD(7) MF_WD(6) MF_WD(5) MF_WD(4) MF_WD(3) MF_WD(2) MF_WD(1) MF_WD(0)
                default: break;
            }
#undef MF_WD
        };

        int4 rec = ld_rec(CSPLIT ? wv : 0);
        int4 rec1 = ld_rec(CSPLIT ? wv + REC_STRIDE : 1);
        int ri = CSPLIT ? wv : 0;
        if (rec.x >= 0) issue_stage(rec.x, rec.y);
        MF_PHASE_START();
        while (rec.x >= 0) {
            const int ch = rec.x;
            write_stage();  // in place: this wave finished reading the previous channel
            MF_PHASE(0);
            const float w = __int_as_float(rec.z);
            const int mvc = rec.y;
            const float et = __int_as_float(rec.w);
            const int4 rec2 = ld_rec(ri + 2 * REC_STRIDE);
            const float* edc = e_d + (size_t)ch * (size_t)nwin;
            f32x4 ed[NTILE];
            // One 16-byte load per group of 4 lags.  A group that straddles an end of the valid range
            // reads up to 3 floats outside this channel's row of norms -- the neighbouring row or the
            // slack around the array -- and those lanes are masked in the epilogue (`ok`).  The wave-
            // uniform test first: all but the first and last waves of a template take the loads without
            // a lane mask (every instruction outside the K loop waits for an issue slot behind the MFMAs
            // of the other waves).  (Requesting the norms of channel c + 1 in front of the K loop of channel
            // c -- for the small problems, whose K loops are short -- changed nothing: round 4.  Nor did the
            // full version of that idea for the 1-tile variant: two register sets, windows / bands / norms of
            // TWO channels in flight, every load unconditional so that the compiler's waits came out as
            // vmcnt(15 .. 27) instead of the vmcnt(0) the epilogue below gets from the `if` around issue_stage --
            // bit-exact, 87 instead of 52 VGPRs, and SLOWER: configs[0] 56 -> 61 us, 8 templates 107 -> 128 us
            // (profiles/r04_mf_fused_prologue.txt).  The small shapes do not wait for memory -- an hour of data
            // sits in the L2 -- but for their own ~400 serial boundary instructions per channel, 2.75 waves to
            // a SIMD; the in-kernel counters, which put 6 600 of 11 300 cycles into the epilogue at configs[0],
            // mostly measure their own s_memtime round trips at that scale.)
            if (wave_inside) {
#pragma unroll
                for (int u = 0; u < NTILE; ++u) ed[u] = *(const f32x4u*)(edc + lag_w + 256 * u + mvc);
            } else {
#pragma unroll
                for (int u = 0; u < NTILE; ++u) {
                    const long long lag = lag_w + 256 * u;
                    if (lag + 3 >= rg.x && lag <= rg.y) {
                        ed[u] = *(const f32x4u*)(edc + lag + mvc);
                    } else {
                        ed[u] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                    }
                }
            }
            if (rec1.x >= 0) issue_stage(rec1.x, rec1.y);
            MF_PHASE(1);

            f32x4 acc[NTILE];
#pragma unroll
            for (int u = 0; u < NTILE; ++u) acc[u] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            const int nq = Kpad >> 4;
            unsigned ap = (unsigned)(size_t)(tp + a_base), bp = (unsigned)(size_t)(dw + b_base);
            float sa[4], sb[4][NTILE];
            // counted waits as in mf_mfma_kernel; the ds_writes above are older than every read
            // and complete first (one wave's LDS ops are ordered), so they can only make the
            // count stricter
#define MF_LDS_READ(dst, addr, off) \
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define MF_MFMA(slot, u) \
    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[slot], sb[slot][u], acc[u], 0, 0, 0)
#define MF_REQ(slot, aoff, boff)                                                    \
    MF_LDS_READ(sa[slot], ap, (aoff));                                               \
    MF_LDS_READ(sb[slot][0], bp, (boff));                                            \
    if constexpr (NTILE > 1) MF_LDS_READ(sb[slot][1 % NTILE], bp, (boff) + 1152);    \
    if constexpr (NTILE > 2) MF_LDS_READ(sb[slot][2 % NTILE], bp, (boff) + 2304);    \
    if constexpr (NTILE > 2) MF_LDS_READ(sb[slot][3 % NTILE], bp, (boff) + 3456)
// One k-step: the NTILE MFMAs of slot `cur` with the NTILE + 1 operand reads of slot `req` (two
// k-steps ahead) issued BETWEEN them.  A wave issues in order and waits ~32 cycles at every MFMA for
// the matrix pipe; a read placed there issues for free, a block of reads after the MFMAs would add
// its issue time to every k-step.  On entry the reads of `cur` and of the k-step after it are
// outstanding (2 (NTILE + 1)), LDS returns in order, so lgkmcnt(NTILE + 1) = "cur has landed".
// (Reading 3 / 4 k-steps ahead in the 2- / 1-tile variants, whose k-steps are only 64 / 32 cycles long,
// changed nothing on configs[0] and its neighbours -- 56.8 -> 56.3 us at L = 128, 91.1 -> 91.1 at L = 256:
// round 4, profiles/r04_mf_fused_prologue.txt.  Those shapes are not waiting for operands.)
#define MF_STEP(cur, req, aoff, boff)                                                \
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(NTILE + 1) : "memory");              \
    __builtin_amdgcn_sched_barrier(0);                                               \
    MF_MFMA(cur, 0);                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                               \
    MF_LDS_READ(sa[req], ap, (aoff));                                                \
    __builtin_amdgcn_sched_barrier(0);                                               \
    if constexpr (NTILE > 1) { MF_MFMA(cur, 1 % NTILE); __builtin_amdgcn_sched_barrier(0); } \
    MF_LDS_READ(sb[req][0], bp, (boff));                                             \
    __builtin_amdgcn_sched_barrier(0);                                               \
    if constexpr (NTILE > 2) { MF_MFMA(cur, 2 % NTILE); __builtin_amdgcn_sched_barrier(0); } \
    if constexpr (NTILE > 1) { MF_LDS_READ(sb[req][1 % NTILE], bp, (boff) + 1152); __builtin_amdgcn_sched_barrier(0); } \
    if constexpr (NTILE > 2) { MF_MFMA(cur, 3 % NTILE); __builtin_amdgcn_sched_barrier(0); } \
    if constexpr (NTILE > 2) { MF_LDS_READ(sb[req][2 % NTILE], bp, (boff) + 2304); MF_LDS_READ(sb[req][3 % NTILE], bp, (boff) + 3456); } \
    __builtin_amdgcn_sched_barrier(0)
            __builtin_amdgcn_sched_barrier(0);
            if (prio) __builtin_amdgcn_s_setprio(0);
            MF_REQ(0, 0, 0);
            MF_REQ(1, 16, 16);
            for (int q = 0; q < nq; ++q) {
                MF_STEP(0, 2, 32, 32);
                MF_STEP(1, 3, 48, 48);
                MF_STEP(2, 0, 64, 72);   // requests k-step 0 of the next trip (past the end: slack)
                MF_STEP(3, 1, 80, 88);
                ap += 64;
                bp += 72;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // option mf.boundary_prio (default 1): a wave outside its K loop (epilogue, staging writes, norm loads,
            // the next channel's loads) asks for a higher issue priority than the waves that stream MFMAs -- its
            // ~250 instructions per channel otherwise wait 40-65 cycles each for a slot behind the other waves'
            // MFMAs (tools/phase/mf_phase.py), and the sooner it is back in its K loop the fewer cycles the matrix
            // pipe idles: 85.0 -> 86.0 % at L = 256, 76.6 -> 77.8 % at L = 128, 62.6 -> 64.6 % at L = 64, same
            // bits (tools/probe_mf_prio.py, round 4).  Round 3 had tried the opposite -- a higher priority INSIDE
            // the K loop -- without effect.
            if (prio == 1) __builtin_amdgcn_s_setprio(1);
            else if (prio == 2) __builtin_amdgcn_s_setprio(2);
            else if (prio == 3) __builtin_amdgcn_s_setprio(3);
#undef MF_LDS_READ
#undef MF_MFMA
#undef MF_REQ
#undef MF_STEP
            MF_PHASE(2);
            if constexpr (CSPLIT) {
                // this channel's CC of the lane's four lags -> row `ri` of the exchange buffer (0 outside the range)
                f32x4 c4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long lag = lag_w + r;
                    const bool ok = wave_inside || (lag >= rg.x && lag <= rg.y);
                    c4[r] = ok ? mf_cc_of<SQRT_NORM>(acc[0][r], et, ed[0][r]) : 0.0f;
                }
                *(f32x4*)(exch + ri * 256 + 16 * a + 4 * kq) = c4;
            } else if (NETWORK_SUM && STEP1 && wave_inside) {
                // every lag of this wave is inside the template's valid range (wave-uniform, all
                // but the first and last tiles of a day): no per-lag range tests
#pragma unroll
                for (int u = 0; u < NTILE; ++u) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float cc = mf_cc_of<SQRT_NORM>(acc[u][r], et, ed[u][r]);
                        sum[u][r] = __fmaf_rn(w, cc, sum[u][r]);
                    }
                }
            } else {
#pragma unroll
            for (int u = 0; u < NTILE; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long lag = lag_w + 256 * u + r;
                    // STEP1 builds carry no division at all; otherwise 32-bit (lags < 2^31)
                    const bool ok = lag >= rg.x && lag <= rg.y && (STEP1 || (unsigned)lag % (unsigned)step == 0);
                    float cc = 0.0f;
                    if (ok) {
                        cc = mf_cc_of<SQRT_NORM>(acc[u][r], et, ed[u][r]);
                        if (!NETWORK_SUM)
                            out[((size_t)t * n_corr + (STEP1 ? lag : (long long)((unsigned)lag / (unsigned)step))) * n_ch + ch] = cc;
                    }
                    if (NETWORK_SUM) sum[u][r] = __fmaf_rn(w, cc, sum[u][r]);
                }
            }
            }
            rec = rec1;
            rec1 = rec2;
            ri += REC_STRIDE;
            MF_PHASE(3);
#ifdef BPMF_PHASE_CYCLES
            ++ph_n_;
#endif
        }
    }
    if constexpr (CSPLIT) {
        __syncthreads();                  // every wave's channels are in the exchange buffer
        const long long lag = lag0 + tid;
        float s = 0.0f;
        if (wave_valid) {
            // the network sum: fmaf(w, cc, sum) over the used channels in channel order (the order of the records)
            for (int i = 0; i < n_used; ++i) s = __fmaf_rn(__int_as_float(l_rec[i].z), exch[i * 256 + tid], s);
        }
        if (lag < n_corr) out[(size_t)t * n_corr + lag] = s;
    } else if (NETWORK_SUM) {
#pragma unroll
        for (int u = 0; u < NTILE; ++u) {
            const long long lag = lag_w + 256 * u;
            float* dst = out + (size_t)t * n_corr + lag;
            if (STEP1 && lag + 3 < n_corr) {
                *(f32x4u*)dst = sum[u];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned smp = (unsigned)(lag + r);
                    const long long i = STEP1 ? (long long)smp : (long long)(smp / (unsigned)step);
                    if ((STEP1 || smp % (unsigned)step == 0) && i < n_corr) out[(size_t)t * n_corr + i] = sum[u][r];
                }
            }
        }
    }
#ifdef BPMF_PHASE_CYCLES
    if (lane == 0 && ph_n_) {
        for (int i = 0; i < 4; ++i) atomicAdd(&g_mf_phase[i], ph_acc_[i]);
        atomicAdd(&g_mf_phase[4], (unsigned long long)ph_n_);
        atomicAdd(&g_mf_phase[5], 1ull);
    }
#endif
}

// ------------------------------------------------------ generic (any step) kernel ---
// One thread per (template, lag); plain fmaf chain.  Used when step != 1 or when the
// template is too long for the LDS tile, and as an independent on-device cross-check.
// SQRT_NORM (option mf.compat_sqrt_norm, a diffing aid, not a production path): the textbook
// normalisation upstream is recollected to use -- cc = num / sqrtf(E_t * E_d) where E_t * E_d exceeds
// 1e-6, else 0 -- instead of num * (r_t * r_d); E_t is the fmaf chain of the template, E_d the
// float difference of the double prefix sums, both as the oracle defines them.
template <bool NETWORK_SUM, bool SQRT_NORM = false>
__global__ __launch_bounds__(256) void mf_direct_kernel(
    const float* __restrict__ tmpl, const int* __restrict__ mv, const float* __restrict__ wgt,
    const float* __restrict__ data, const float* __restrict__ e_t,
    const float* __restrict__ e_d, const int2* __restrict__ range, long long step, int L,
    long long N, int n_ch, long long n_corr, float* __restrict__ out,
    const double* __restrict__ cs_local = nullptr, const double* __restrict__ cs_off = nullptr, long long nq = 0)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (i >= n_corr) return;
    const int2 rg = range[t];
    const bool ok = i >= rg.x && i <= rg.y;
    const long long nwin = N - L + 1;
    float sum = 0.0f;
    if (ok) {
        for (int ch = 0; ch < n_ch; ++ch) {
            const float w = wgt[(size_t)t * n_ch + ch];
            if (w == 0.0f) continue;
            const long long j = i * step + mv[(size_t)t * n_ch + ch];
            const float* tp = tmpl + ((size_t)t * n_ch + ch) * (size_t)L;
            const float* d = data + (size_t)ch * (size_t)N + j;
            float num = 0.0f;
            for (int l = 0; l < L; ++l) num = __fmaf_rn(tp[l], d[l], num);
            float cc = 0.0f;
            if constexpr (SQRT_NORM) {
                float et = 0.0f;
                for (int l = 0; l < L; ++l) et = __fmaf_rn(tp[l], tp[l], et);
                const double* lo = cs_local + (size_t)ch * (size_t)N;
                const double* of = cs_off + (size_t)ch * (size_t)nq;
                const long long nh = j + L - 1;
                const double hi = of[nh / CSUM_CHUNK] + lo[nh];
                const double low = j > 0 ? of[(j - 1) / CSUM_CHUNK] + lo[j - 1] : 0.0;
                const float den2 = et * (float)(hi - low);
                if (den2 > 1.0e-6f) cc = num / sqrtf(den2);
            } else {
                const float nrm = e_t[(size_t)t * n_ch + ch] * e_d[(size_t)ch * nwin + j];  // r_t * r_d
                if (nrm < MAX_NORM) cc = num * nrm;
            }
            if (NETWORK_SUM)
                sum = __fmaf_rn(w, cc, sum);
            else
                out[((size_t)t * n_corr + i) * n_ch + ch] = cc;
        }
    }
    if (NETWORK_SUM) out[(size_t)t * n_corr + i] = sum;
}

// ------------------------------------------------------------------- workspace ---
struct MfWorkspace {
    double* local;  // [n_ch, N]
    double* tot;    // [n_ch, nq]
    double* off;    // [n_ch, nq]
    float* e_d;     // [n_ch, nwin]
    float* e_t;     // [T, n_ch]
    int2* range;    // [T]
    int4* chan_rec; // [T, n_ch + 2] compact used-channel records
    void* sp_day;   // option mf.split16: the split day (mf_split_api.h), nullptr otherwise
    void* sp_batch; // option mf.split16: band images of the template batch
    size_t bytes;
};

static MfWorkspace mf_carve(void* base, size_t L, size_t N, size_t T, size_t n_ch)
{
    MfWorkspace ws;
    const size_t nq = (N + CSUM_CHUNK - 1) / CSUM_CHUNK;
    const size_t nwin = N >= L ? N - L + 1 : 0;
    char* p = (char*)base;
    size_t o = 0;
    ws.local = (double*)(p + o); o += align_up(n_ch * N * sizeof(double), 256);
    ws.tot = (double*)(p + o);   o += align_up(n_ch * nq * sizeof(double), 256);
    ws.off = (double*)(p + o);   o += align_up(n_ch * nq * sizeof(double), 256);
    o += 256;  // slack: the main kernel's 16-byte norm loads may start 3 floats early ...
    ws.e_d = (float*)(p + o);    o += align_up(n_ch * nwin * sizeof(float) + 64, 256);  // ... or end 3 late
    // option mf.split16: the split day lies with the per-day arrays, in front of everything that depends on T (a
    // prepared day stays valid when the template count changes); the option is part of the workspace's size
    const bool split16 = option(OPT_MF_SPLIT16) != 0;
    ws.sp_day = nullptr;
    ws.sp_batch = nullptr;
    if (split16) { ws.sp_day = p + o; o += align_up(sp::day_region_bytes(N, n_ch), 256); }
    ws.e_t = (float*)(p + o);    o += align_up(T * n_ch * sizeof(float), 256);
    ws.range = (int2*)(p + o);   o += align_up(T * sizeof(int2), 256);
    ws.chan_rec = (int4*)(p + o); o += align_up(T * (n_ch + 2) * sizeof(int4), 256);
    if (split16) { ws.sp_batch = p + o; o += align_up(sp::batch_region_bytes(T, n_ch, L), 256); }
    ws.bytes = o;
    return ws;
}

static int mf_check_sizes(size_t step, size_t L, size_t N, size_t T, size_t S, size_t C,
                          size_t n_corr)
{
    if (step == 0 || L == 0 || T == 0 || S == 0 || C == 0) {
        set_error("matched filter: zero-sized dimension (step=%zu L=%zu T=%zu S=%zu C=%zu)", step,
                  L, T, S, C);
        return -1;
    }
    if (N < L) {
        set_error("matched filter: data (N=%zu) shorter than the templates (L=%zu)", N, L);
        return -1;
    }
    if (n_corr != (N - L) / step + 1) {
        set_error("matched filter: n_corr=%zu but (N-L)/step+1=%zu", n_corr, (N - L) / step + 1);
        return -1;
    }
    if (N > 0x7fffffffull || T * S * C > 0x7fffffffull || L > 0x7fffffffull) {
        set_error("matched filter: dimension exceeds the int32 index range");
        return -1;
    }
    return 0;
}

// bpmf_mf_run (host pointers) computes its first template batches piece by piece while the day is still
// arriving from the host: a launch of bpmf_mf_run_dev restricted to the data offsets [t_mf_off_lo,
// t_mf_off_hi) -- multiples of MF_LAGS_PER_WG, or the end -- (hi < 0: all of them), and, for the later
// pieces of a batch, without the per-template preparation and output fill the first piece did.
thread_local long long t_mf_off_lo = 0, t_mf_off_hi = -1;
thread_local bool t_mf_continue = false;

// Which workspaces hold a day that was prepared WITH its fp16 split (option mf.split16): a caller of the *_dev entry
// points that prepares a day with the option off and runs with it on (BPMF_MF_DATA_PREPARED) would otherwise correlate
// whatever the split region holds.  Keyed by the workspace's base address; bpmf_mf_prepare_data_dev sets / clears it.
static std::mutex g_split_days_mutex;
struct SplitDay { const void* data; size_t N, n_ch; };
static std::vector<std::pair<const void*, SplitDay>> g_split_days;
static void split_day_note(const void* ws_base, const void* data, size_t N, size_t n_ch, bool with_split)
{
    std::lock_guard<std::mutex> g(g_split_days_mutex);
    for (size_t i = 0; i < g_split_days.size(); ++i)
        if (g_split_days[i].first == ws_base) { g_split_days.erase(g_split_days.begin() + i); break; }
    if (with_split) {
        if (g_split_days.size() >= 64) g_split_days.erase(g_split_days.begin());
        g_split_days.push_back({ws_base, {data, N, n_ch}});
    }
}
static bool split_day_known(const void* ws_base, const void* data, size_t N, size_t n_ch)
{
    std::lock_guard<std::mutex> g(g_split_days_mutex);
    for (auto& e : g_split_days)
        if (e.first == ws_base) return e.second.data == data && e.second.N == N && e.second.n_ch == n_ch;
    return false;
}

// the launch of bpmf_mf_run_dev takes the MFMA kernels (which can be restricted to a range of lag blocks)
static bool mf_uses_mfma(size_t step, size_t L, size_t N, size_t T, size_t n_corr, int network_sum, int flags)
{
    const size_t n_offsets = (n_corr - 1) * step + 1;
    const size_t n_lag_blocks = (n_offsets + MF_LAGS_PER_WG - 1) / MF_LAGS_PER_WG;
    const int need_r = (mf_window_len((int)L) + MF_THREADS - 1) / MF_THREADS;
    const int need_t = (mf_band_len((int)L) + MF_THREADS - 1) / MF_THREADS;
    const size_t max_mfma_step = (size_t)option(OPT_MF_MAX_MFMA_STEP);
    const bool sqrt_norm = option(OPT_MF_COMPAT_SQRT_NORM) != 0;
    (void)sqrt_norm; (void)network_sum;       // (round 6: the MFMA epilogues store per-channel CCs under the switch too)
    return step <= max_mfma_step && !(flags & BPMF_MF_FORCE_DIRECT) && need_r <= 24 &&
           need_t <= 9 && T * (n_lag_blocks + 8) < 0x7fffffffull && N < ((size_t)1 << 30) - 8192;
}

}  // namespace bpmf

using namespace bpmf;

extern "C" size_t bpmf_mf_workspace_bytes(size_t L, size_t N, size_t T, size_t S, size_t C)
{
    return mf_carve(nullptr, L, N, T, S * C).bytes;
}

// The per-day preparation for the samples [samp_lo, samp_hi) (multiples of CSUM_CHUNK, or the end of the
// trace): chunk-local prefix sums of the chunks in the range, the chain of chunk offsets continued, the norms
// of the windows the range completes.  The whole day in one piece = bpmf_mf_prepare_data_dev; piece by piece
// (bpmf_mf_run, a day still arriving from the host) the same kernels run the same additions in the same order.
static int mf_prepare_range(const float* d_data, size_t L, size_t N, size_t n_ch, const MfWorkspace& ws,
                            hipStream_t stream, size_t samp_lo, size_t samp_hi)
{
    const size_t nq = (N + CSUM_CHUNK - 1) / CSUM_CHUNK;
    const size_t nwin = N - L + 1;
    const size_t q_lo = samp_lo / CSUM_CHUNK, q_hi = (samp_hi + CSUM_CHUNK - 1) / CSUM_CHUNK;
    if (q_hi <= q_lo) return 0;
    {
        size_t n = n_ch * (q_hi - q_lo);
        mf_csum_local_kernel<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream>>>(
            d_data, n_ch, N, nq, ws.local, ws.tot, q_lo, q_hi - q_lo);
        BPMF_LAUNCH_CHECK();
    }
    mf_csum_offsets_kernel<<<dim3((unsigned)((n_ch + 63) / 64)), dim3(64), 0, stream>>>(
        ws.tot, n_ch, nq, ws.off, q_lo, q_hi);
    BPMF_LAUNCH_CHECK();
    // windows [w_lo, w_hi): window j needs the prefix sums up to sample j + L - 1
    const size_t w_lo = samp_lo >= L - 1 ? samp_lo - (L - 1) : 0;
    const size_t w_hi = samp_hi >= N ? nwin : (samp_hi >= L - 1 ? std::min(nwin, samp_hi - (L - 1)) : 0);
    if (w_hi > w_lo) {
        // (option mf.compat_sqrt_norm decides what the norm arrays hold: a caller of the *_dev entry points
        // that switches it prepares the data again -- MatchedFilterGPU keys its prepared state by it)
        mf_window_energy_kernel<<<dim3((unsigned)((w_hi - w_lo + 255) / 256), (unsigned)n_ch), dim3(256), 0,
                                  stream>>>(ws.local, ws.off, n_ch, N, nq, L, nwin,
                                            option(OPT_MF_COMPAT_SQRT_NORM) != 0 ? 1 : 0, ws.e_d, w_lo, w_hi);
        BPMF_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int bpmf_mf_prepare_data_dev(const float* d_data, size_t L, size_t N, size_t S, size_t C,
                                        void* d_workspace, size_t workspace_bytes,
                                        bpmf_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    const size_t n_ch = S * C;
    if (!d_data || !d_workspace || N < L || L == 0 || n_ch == 0) {
        set_error("bpmf_mf_prepare_data_dev: bad argument");
        return -1;
    }
    MfWorkspace ws = mf_carve(d_workspace, L, N, 0, n_ch);
    if (workspace_bytes < ws.bytes) {
        set_error("bpmf_mf_prepare_data_dev: workspace too small (%zu < %zu)", workspace_bytes,
                  ws.bytes);
        return -1;
    }
    if (option(OPT_MF_COMPAT_SEQUENTIAL_CSUM) != 0) {
        // (like mf.compat_sqrt_norm: a caller of the *_dev entry points that switches it prepares the data again)
        const size_t nq = (N + CSUM_CHUNK - 1) / CSUM_CHUNK;
        const size_t nwin = N - L + 1;
        mf_csum_sequential_kernel<<<dim3((unsigned)n_ch), dim3(64), 0, stream>>>(d_data, N, nq, ws.local, ws.off);
        BPMF_LAUNCH_CHECK();
        mf_window_energy_kernel<<<dim3((unsigned)((nwin + 255) / 256), (unsigned)n_ch), dim3(256), 0,
                                  stream>>>(ws.local, ws.off, n_ch, N, nq, L, nwin,
                                            option(OPT_MF_COMPAT_SQRT_NORM) != 0 ? 1 : 0, ws.e_d, 0, nwin);
        BPMF_LAUNCH_CHECK();
    } else if (int rc = mf_prepare_range(d_data, L, N, n_ch, ws, stream, 0, N)) {
        return rc;
    }
    // option mf.split16: the day as fp16 (hi, lo) pairs, each channel scaled by a power of two (mf_split.h)
    const bool with_split = ws.sp_day && sp::usable(L, N);
    split_day_note(d_workspace, d_data, N, n_ch, with_split);
    if (with_split) return sp::prepare_day(d_data, N, n_ch, ws.sp_day, stream);
    return 0;
}

extern "C" int bpmf_mf_run_dev(const float* d_templates, const int32_t* d_moveouts,
                               const float* d_weights, const float* d_data, size_t step, size_t L,
                               size_t N, size_t T, size_t S, size_t C, size_t n_corr,
                               int network_sum, int flags, void* d_workspace,
                               size_t workspace_bytes, bpmf_stream_t stream_, float* d_cc_out)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_templates || !d_moveouts || !d_weights || !d_data || !d_workspace || !d_cc_out) {
        set_error("bpmf_mf_run_dev: null pointer");
        return -1;
    }
    if (int rc = mf_check_sizes(step, L, N, T, S, C, n_corr)) return rc;
    const size_t n_ch = S * C;
    MfWorkspace ws = mf_carve(d_workspace, L, N, T, n_ch);
    if (workspace_bytes < ws.bytes) {
        set_error("bpmf_mf_run_dev: workspace too small (%zu < %zu)", workspace_bytes, ws.bytes);
        return -1;
    }
    if (!(flags & BPMF_MF_DATA_PREPARED)) {
        if (int rc = bpmf_mf_prepare_data_dev(d_data, L, N, S, C, d_workspace, workspace_bytes,
                                              stream_))
            return rc;
    }
    const size_t lds = mf_lds_bytes((int)L);
    // the MFMA kernels evaluate every data-sample offset and keep the multiples of `step`
    const size_t n_offsets = (n_corr - 1) * step + 1;
    const size_t n_lag_blocks = (n_offsets + MF_LAGS_PER_WG - 1) / MF_LAGS_PER_WG;
    // staging registers needed per thread (window / band), rounded to a compiled variant
    const int need_r = (mf_window_len((int)L) + MF_THREADS - 1) / MF_THREADS;
    const int need_t = (mf_band_len((int)L) + MF_THREADS - 1) / MF_THREADS;
    const size_t max_mfma_step = (size_t)option(OPT_MF_MAX_MFMA_STEP);  // beyond this (64) the direct kernel wins
    // (the MFMA kernels address the data through buffer descriptors with 32-bit byte offsets:
    // traces of 2^30 samples or more take the generic kernel)
    // option mf.compat_sqrt_norm: num / sqrtf(E_t * E_d) in the epilogue of the MFMA kernels too (network sums and,
    // since round 6, per-channel output: the inter-template CC keeps MFMA speed under the upstream-recollected profile)
    const bool sqrt_norm = option(OPT_MF_COMPAT_SQRT_NORM) != 0;
    const bool use_mfma = mf_uses_mfma(step, L, N, T, n_corr, network_sum, flags);
    (void)max_mfma_step;
    // a launch over a range of data offsets (bpmf_mf_run, see t_mf_off_lo): MFMA kernels only
    const bool ranged = t_mf_off_hi >= 0;
    if (ranged && (!use_mfma || t_mf_off_lo % MF_LAGS_PER_WG != 0)) {
        set_error("bpmf_mf_run_dev: internal error: a range of lag blocks on a launch that cannot take one");
        return -1;
    }
    const size_t off_lo = ranged ? (size_t)t_mf_off_lo : 0;
    const size_t off_hi = ranged ? std::min<size_t>((size_t)t_mf_off_hi, n_offsets) : n_offsets;
    if (off_hi <= off_lo) return 0;
    const bool first_piece = !(ranged && t_mf_continue);
    // option mf.split16: the split-precision kernel takes every launch the MFMA kernels would take for templates of
    // whatever length (in segments of at most 376 samples; under mf.compat_sqrt_norm its epilogue divides by sqrtf(E_t * E_d))
    // mf.split16 = 1 leaves small launches to the exact kernel (below SP_MIN_BLOCKS (template, 8192-lag block) pairs a
    // wave's chain of per-channel stagings is latency, not rate: configs[0], 88 pairs, ran 0.71x the exact kernel's speed, 176 pairs x 1.1-1.6:
    // profiles/r06_mf_split16.txt); = 2 takes the split kernel for every launch (the small shapes of the tests)
    constexpr size_t SP_MIN_BLOCKS = 128;
    const bool sp_small = option(OPT_MF_SPLIT16) == 1 && T * ((n_offsets + sp::LAGS_PER_WG - 1) / sp::LAGS_PER_WG) < SP_MIN_BLOCKS;
    const bool split16 = ws.sp_day != nullptr && use_mfma && sp::usable(L, N) && !sp_small;
    if (split16 && ranged) {
        set_error("bpmf_mf_run_dev: internal error: a range of lag blocks under mf.split16");
        return -1;
    }
    if (split16 && !split_day_known(d_workspace, d_data, N, n_ch)) {
        set_error("bpmf_mf_run_dev: option mf.split16 is on but the day in this workspace was prepared without it (or is "
                  "another day): call bpmf_mf_prepare_data_dev again, or drop BPMF_MF_DATA_PREPARED");
        return -1;
    }
    const bool wave_kernel = !split16 && use_mfma && option(OPT_MF_WAVE_KERNEL) != 0 && mf_kpad((int)L) <= 272;
    // tiles (of 256 lags) per wave of that kernel: 4 unless the problem is too small to give every SIMD ~4 waves
    // (option mf.tiles_per_wave: 0 = this rule, 1 / 2 / 4 = forced)
    // (calibrated on an hour-long series with 4 .. 256 templates, tools/probe_mf_ntile_T.py,
    // profiles/r04_mf_ntile_T.txt: 4 tiles from two full rounds of 4 waves per SIMD on, 2 tiles -- 5 waves
    // per SIMD at 83-90 VGPRs -- from one wave per SIMD on, 1 tile below)
    // (by the size of the WHOLE problem, also for a launch over a range: every piece takes the same variant)
    const size_t waves4 = T * ((n_offsets + 4095) / 4096) * 4;
    int ntile = waves4 >= 8192 ? 4 : (waves4 >= 1024 ? 2 : 1);
    {
        const long forced = option(OPT_MF_TILES_PER_WAVE);
        if (forced == 1 || forced == 2 || forced == 4) ntile = (int)forced;
    }
    // small problems: the wave kernel does the per-template preparation itself (mf_fused_prologue)
    const bool fused = wave_kernel && ntile < 4 && n_ch <= 256 && option(OPT_MF_FUSED_PROLOGUE) != 0;
    // (bit 0: mf.compat_exclusive_last_lag, bit 1: mf.compat_range_all_channels -- both only shape the valid lag range)
    const int exclusive_last = (option(OPT_MF_COMPAT_EXCLUSIVE_LAST_LAG) != 0 ? 1 : 0) |
                               (option(OPT_MF_COMPAT_RANGE_ALL_CHANNELS) != 0 ? 2 : 0);
    if (!fused) {
        // template norms, lag ranges, channel records: one launch, one workgroup per template (also in front
        // of every later piece of a batch: the pieces of two batches alternate, and the records live in the
        // one workspace both use)
        mf_prologue_kernel<<<dim3((unsigned)T), dim3(64), 0, stream>>>(
            d_templates, d_moveouts, d_weights, (int)T, (int)n_ch, (long long)step, (long long)L, (long long)N,
            (long long)n_corr, exclusive_last, sqrt_norm ? 1 : 0, ws.e_t, ws.range, ws.chan_rec);
        BPMF_LAUNCH_CHECK();
    }
    if (!first_piece) {
    } else if (!network_sum)
        BPMF_HIP_CHECK(hipMemsetAsync(d_cc_out, 0, T * n_corr * n_ch * sizeof(float), stream));
    else if (option(OPT_DEBUG_POISON_OUTPUT) != 0)      // tests: a CC sum that no kernel writes comes back as NaN
        BPMF_HIP_CHECK(hipMemsetAsync(d_cc_out, 0xFF, T * n_corr * sizeof(float), stream));

    profile_mark(BPMF_KERNEL_MF_MAIN, 0, stream);
    if (split16) {
        const size_t nb_cnt = (n_offsets + sp::LAGS_PER_WG - 1) / sp::LAGS_PER_WG;
        if (int rc = sp::run(d_templates, d_moveouts, ws.sp_day, ws.sp_batch, ws.chan_rec, ws.e_d, ws.range, step, L, N, T,
                             n_ch, n_corr, network_sum, sqrt_norm ? 1 : 0, 0, nb_cnt, d_cc_out, stream))
            return rc;
    } else if (use_mfma) {
        // 8 XCDs x ceil(n_lag_blocks x T / 8) (lag block, template) pairs (mf_tile_of_block)
        const size_t nb_lo = off_lo / MF_LAGS_PER_WG, nb_cnt = (off_hi - off_lo + MF_LAGS_PER_WG - 1) / MF_LAGS_PER_WG;
        dim3 grid((unsigned)(8 * ((T * nb_cnt + 7) / 8)));
        const bool big_lds = lds > 64 * 1024;  // long templates: opt in to > 64 KB dynamic LDS
#define BPMF_MF_LAUNCH2(NS, R, TT, S1) \
    do { if (sqrt_norm) BPMF_MF_LAUNCH3(NS, R, TT, S1, true); else BPMF_MF_LAUNCH3(NS, R, TT, S1, false); } while (0)
#define BPMF_MF_LAUNCH3(NS, R, TT, S1, SQ)                                                        \
    do {                                                                                              \
        auto kfn = mf_mfma_kernel<NS, R, TT, S1, SQ>;                                                 \
        if (big_lds)                                                                                  \
            BPMF_HIP_CHECK(hipFuncSetAttribute((const void*)kfn,                                      \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        kfn<<<grid, dim3(MF_THREADS), lds, stream>>>(                                                 \
            d_templates, ws.chan_rec, d_data, ws.e_d, ws.range, (int)L, (long long)N, (int)T,         \
            (int)n_ch, (long long)n_corr, (int)step, d_cc_out, (int)nb_cnt,                            \
            (int)option(OPT_MF_BOUNDARY_PRIO), (int)nb_lo);                                            \
    } while (0)
#define BPMF_MF_LAUNCH(NS, R, TT) \
    do { if (step == 1) BPMF_MF_LAUNCH2(NS, R, TT, true); else BPMF_MF_LAUNCH2(NS, R, TT, false); } while (0)
        if (wave_kernel) {                          // L <= 257: independent waves, no barrier
            const int Kp = mf_kpad((int)L), Ww = 256 * ntile - 16 + Kp;
            // option mf.channel_split = n (0: off): tiny problems -- one tile per wave, fused prologue, network sum,
            // step 1, at most 32 channels, at most n waves of 256 lags in the whole launch, not a piece of a larger
            // launch -- run the CSPLIT variant: four waves per 256 lags, every fourth used channel each
            const long csplit_max = option(OPT_MF_CHANNEL_SPLIT);
            const bool csplit = ntile == 1 && fused && network_sum && step == 1 && n_ch <= 32 && !ranged &&
                                csplit_max > 0 && waves4 * 4 <= (size_t)csplit_max;
            const size_t lags_wg = csplit ? (size_t)256 : (size_t)4 * 256 * ntile;
            const size_t nbw_lo = off_lo / lags_wg;       // (off_lo is a multiple of 4096 = of every variant's span)
            const size_t n_blocks_w = (off_hi - off_lo + lags_wg - 1) / lags_wg;
            if (T * (n_blocks_w + 8) >= 0x7fffffffull) {
                set_error("bpmf_mf_run_dev: grid too large");
                return -1;
            }
            dim3 grid_w((unsigned)(8 * ((T * n_blocks_w + 7) / 8)));
            // (+ 256: slack for the operand prefetch one k-step past the end; FUSED: 16 ints + the channel records)
            const int Wbuf = ntile < 4 ? 64 * (ntile == 2 ? 12 : 8) : Ww;      // (FULLW: room for every staging register)
            const size_t wl = (size_t)4 * (mf_band_len((int)L) + (Wbuf + 2 * (Wbuf >> 4) + 2 + 63) / 64 * 64 + 64) * sizeof(float) + 256 +
                              (fused ? 64 + (n_ch + MF_REC_TERMINATORS) * sizeof(int4) : 0) +
                              (csplit ? n_ch * 256 * sizeof(float) : 0);
#define BPMF_MF_WAVE_LAUNCH4(NS, S1, R, NT, SQ, FU)  BPMF_MF_WAVE_LAUNCH5(NS, S1, R, NT, SQ, FU, false)
#define BPMF_MF_WAVE_LAUNCH5(NS, S1, R, NT, SQ, FU, CS)                                         \
    mf_mfma_wave_kernel<NS, R, 5, S1, NT, SQ, FU, CS><<<grid_w, dim3(MF_THREADS), wl, stream>>>( \
        d_templates, ws.chan_rec, d_data, ws.e_d, ws.range, (int)L, (long long)N, (int)T,        \
        (int)n_ch, (long long)n_corr, (int)step, d_cc_out, (int)n_blocks_w, (int)option(OPT_MF_BOUNDARY_PRIO), \
        d_moveouts, d_weights, exclusive_last, (int)nbw_lo)
#define BPMF_MF_WAVE_LAUNCH3(NS, S1, R, NT, FU) \
    do { if (sqrt_norm) BPMF_MF_WAVE_LAUNCH4(NS, S1, R, NT, true, FU); else BPMF_MF_WAVE_LAUNCH4(NS, S1, R, NT, false, FU); } while (0)
#define BPMF_MF_WAVE_LAUNCH(NS, S1)                                                              \
    do {                                                                                         \
        if (ntile == 4) BPMF_MF_WAVE_LAUNCH3(NS, S1, 20, 4, false);                              \
        else if (ntile == 2 && fused) BPMF_MF_WAVE_LAUNCH3(NS, S1, 12, 2, true);                 \
        else if (ntile == 2) BPMF_MF_WAVE_LAUNCH3(NS, S1, 12, 2, false);                         \
        else if (fused) BPMF_MF_WAVE_LAUNCH3(NS, S1, 8, 1, true);                                \
        else BPMF_MF_WAVE_LAUNCH3(NS, S1, 8, 1, false);                                          \
    } while (0)
            if (csplit) { if (sqrt_norm) BPMF_MF_WAVE_LAUNCH5(true, true, 8, 1, true, true, true); else BPMF_MF_WAVE_LAUNCH5(true, true, 8, 1, false, true, true); }
            else if (network_sum && step == 1) BPMF_MF_WAVE_LAUNCH(true, true);
            else if (network_sum) BPMF_MF_WAVE_LAUNCH(true, false);
            else if (step == 1) BPMF_MF_WAVE_LAUNCH(false, true);
            else BPMF_MF_WAVE_LAUNCH(false, false);
#undef BPMF_MF_WAVE_LAUNCH
#undef BPMF_MF_WAVE_LAUNCH3
#undef BPMF_MF_WAVE_LAUNCH4
#undef BPMF_MF_WAVE_LAUNCH5
        } else if (need_r <= 17 && need_t <= 2) {   // L <= 273
            if (network_sum) BPMF_MF_LAUNCH(true, 17, 2); else BPMF_MF_LAUNCH(false, 17, 2);
        } else if (need_r <= 20 && need_t <= 5) {   // L <= 1041
            if (network_sum) BPMF_MF_LAUNCH(true, 20, 5); else BPMF_MF_LAUNCH(false, 20, 5);
        } else {                                    // L <= 2065
            if (network_sum) BPMF_MF_LAUNCH(true, 24, 9); else BPMF_MF_LAUNCH(false, 24, 9);
        }
#undef BPMF_MF_LAUNCH
#undef BPMF_MF_LAUNCH2
#undef BPMF_MF_LAUNCH3
    } else {
        dim3 grid((unsigned)((n_corr + 255) / 256), (unsigned)T);
        if (T > 65535) {
            set_error("bpmf_mf_run_dev: generic kernel supports at most 65535 templates per call");
            return -1;
        }
        const long long nq = (long long)((N + CSUM_CHUNK - 1) / CSUM_CHUNK);
#define BPMF_MF_DIRECT(NS, SQ)                                                                     \
    mf_direct_kernel<NS, SQ><<<grid, dim3(256), 0, stream>>>(                                      \
        d_templates, d_moveouts, d_weights, d_data, ws.e_t, ws.e_d, ws.range, (long long)step,     \
        (int)L, (long long)N, (int)n_ch, (long long)n_corr, d_cc_out, ws.local, ws.off, nq)
        if (network_sum) { if (sqrt_norm) BPMF_MF_DIRECT(true, true); else BPMF_MF_DIRECT(true, false); }
        else { if (sqrt_norm) BPMF_MF_DIRECT(false, true); else BPMF_MF_DIRECT(false, false); }
#undef BPMF_MF_DIRECT
    }
    BPMF_LAUNCH_CHECK();
    profile_mark(BPMF_KERNEL_MF_MAIN, 1, stream);
    return 0;
}

// Host-pointer call: the CC matrix (17 GB at cfg2) does not have to exist on the device, and its
// way back to pageable host memory (16 GB/s through the runtime's own staging, measured) would
// cost as much as computing it.  Templates run in batches of ~1 GB of output on one stream; a
// second stream drains the previous batch in 64 MB pieces into two pinned buffers, from which a
// few host threads copy into the caller's array while the next piece is in flight: the call takes
// about max(compute, transfer) instead of their sum.

static int bpmf_mf_run_impl(const float* templates, const int32_t* moveouts, const float* weights,
                           const float* data, size_t step, size_t L, size_t N, size_t T, size_t S,
                           size_t C, size_t n_corr, int network_sum, int flags, int device,
                           float* cc_out)
{
    if (!templates || !moveouts || !weights || !data || !cc_out) {
        set_error("bpmf_mf_run: null pointer");
        return -1;
    }
    if (int rc = mf_check_sizes(step, L, N, T, S, C, n_corr)) return rc;
    BPMF_BIND_DEVICE(device);
    t_call_stats = HostCallStats();
    const double t_call0 = host_now_ms();
    const size_t n_ch = S * C;
    const size_t row_bytes = n_corr * (network_sum ? 1 : n_ch) * sizeof(float);   // per template
    // options mf.host_batch_kb / mf.host_piece_kb: sizes of a batch's output and of a pinned piece
    // (defaults 1 GB / 64 MB; the tests shrink them to cross every batch and piece boundary)
    const long e_b = option(OPT_MF_HOST_BATCH_KB), e_p = option(OPT_MF_HOST_PIECE_KB);
    const size_t batch_bytes = e_b > 0 ? (size_t)e_b << 10 : (size_t)1 << 30;
    const size_t PIECE = e_p > 0 ? (size_t)e_p << 10 : (size_t)64 << 20;
    size_t TB = std::max<size_t>(1, batch_bytes / std::max<size_t>(row_bytes, 1));
    if (!e_b) TB = std::max<size_t>(TB, 8);   // a launch of fewer templates wastes the device
    TB = std::min(T, TB);
    // Batches [bt[b], bt[b + 1]): at most TB templates each, TAPERING at the end -- ..., 21, 13, 8, 5, 3, 2.
    // The result of a batch travels to the host while the NEXT batch is computed; behind the last kernel there
    // is nothing left to hide a transfer behind, and a transfer takes about half as long as computing the same
    // templates: with equal batches the call ended with ~30 ms of draining (round 5: 17 batches of 31 and one
    // of 4 -- the drain of the last full batch outlived the 8 ms of the small one).  Each batch of the taper is
    // ~0.6 of the one before, so every drain fits behind the next kernel and the call ends with the drain of two
    // templates.  (Every launch still has thousands of workgroups: the lag blocks.)
    std::vector<size_t> bt;
    {
        std::vector<size_t> taper;                    // from the end: 2, 3, 5, 8, 13, 21, ...  while below TB
        size_t a = 2, b2 = 3, used = 0;
        while (a < TB && used + a <= T) {
            taper.push_back(a);
            used += a;
            const size_t next = b2;                   // 2, 3, 5, 8, 13, 21: each ~1.6 x the one behind it
            b2 = a + b2;
            a = next;
        }
        const size_t rest = T - used;
        const size_t n_full = (rest + TB - 1) / TB;
        bt.push_back(0);
        for (size_t i = 0; i < n_full; ++i)           // near-equal batches of at most TB
            bt.push_back(bt.back() + rest / n_full + (i < rest % n_full ? 1 : 0));
        for (size_t i = taper.size(); i-- > 0;) bt.push_back(bt.back() + taper[i]);
    }
    const size_t n_batch = bt.size() - 1;
    const size_t b_tp = T * n_ch * L * sizeof(float), b_mv = T * n_ch * sizeof(int32_t),
                 b_w = T * n_ch * sizeof(float), b_d = n_ch * N * sizeof(float),
                 b_out = TB * row_bytes, b_ws = bpmf_mf_workspace_bytes(L, N, TB, S, C);
    // streams, events, pinned pieces and the device working set are the device's (context.h): created
    // once, reused by every call, one call per device at a time -- nothing is created or destroyed here
    DeviceContext* ctx = device_context(device);
    if (!ctx) return -2;
    std::lock_guard<std::mutex> call_lock(ctx->call_mutex);
    FanoutScope fan;      // (behind the lock: a source waits for its peers before another call may touch its data)
    size_t o_tp = 0, o_mv = o_tp + align_up(b_tp, 256), o_w = o_mv + align_up(b_mv, 256),
           o_d = o_w + align_up(b_w, 256), o_out0 = o_d + align_up(b_d, 256),
           o_out1 = o_out0 + align_up(b_out, 256),
           o_ws = o_out1 + (n_batch > 1 ? align_up(b_out, 256) : 0), total = o_ws + b_ws;
    char* base = ctx->reserve_device(total);
    if (!base) return -2;
    // (a pinned piece never needs to be larger than one batch of output)
    // (... nor, for the upload of the day through the same pieces, than the day of data)
    if (int prc = ctx->reserve_pinned(std::max(std::min(PIECE, std::max<size_t>(b_out, 4096)),
                                               std::min<size_t>((size_t)64 << 20, std::max<size_t>(b_d, 4096)))))
        return prc;
    char* const* pinned = ctx->pinned;
    hipStream_t s_run = ctx->s_run, s_copy = ctx->s_copy;
    hipEvent_t* ev_batch = ctx->ev_batch;
    hipEvent_t* ev_piece = ctx->ev_piece;
    int rc = 0;
    auto fail = [&](hipError_t e, const char* what) {
        if (!rc) set_error("bpmf_mf_run: %s failed: %s", what, hipGetErrorString(e));
        rc = -2;
    };
#define MF_TRY(expr, what) do { hipError_t e_ = (expr); if (e_ != hipSuccess) fail(e_, what); } while (0)
    // (through the pinned pieces, like the day: a pageable source makes the runtime page-lock the caller's pages for the
    // copy, and arrays that are temporaries of the Python wrapper -- the int32 moveouts, the broadcast weights -- are
    // freed when the call returns: the NEXT call's first copy then stalls for tens of milliseconds, bp.hip: upload())
    if (!rc) MF_TRY(staged_upload_rows(ctx, (float*)(base + o_tp), templates, 1, b_tp / 4, 0, b_tp / 4, s_run), "H2D templates");
    if (!rc) MF_TRY(staged_upload_rows(ctx, (float*)(base + o_mv), (const float*)moveouts, 1, b_mv / 4, 0, b_mv / 4, s_run), "H2D moveouts");
    if (!rc) MF_TRY(staged_upload_rows(ctx, (float*)(base + o_w), weights, 1, b_w / 4, 0, b_w / 4, s_run), "H2D weights");
    // ---- The day of data.  A peer of a multi-device call copies it from the first device; a small problem
    // uploads it in one go.  A day-long series arrives IN PIECES on the copy stream while the first
    // template batch is computed on the lags whose windows have arrived (launches of bpmf_mf_run_dev over
    // ranges of lag blocks, t_mf_off_lo): the 2 GB of configs[1] took 90 ms in front of the first kernel
    // (1072.7 ms end to end against 982.9 resident, round-4 bench; BPMF makes exactly this call,
    // similarity_search.py:526-533).
    const bool use_mfma = mf_uses_mfma(step, L, N, std::min(TB, T), n_corr, network_sum, flags);
    const size_t n_offsets = (n_corr - 1) * step + 1;
    bool from_peer = false;
    if (!rc) {
        const char* what = "";
        hipError_t e_ = hipSuccess;
        from_peer = fanout_peer_copy(fan, ctx, base + o_d, b_d, s_run, &e_, &what);
        if (from_peer && e_ != hipSuccess) fail(e_, what);
    }
    // option mf.host_piece_lags: samples of the first piece (default 131 072; the tests shrink it), 0 = off
    const size_t PIECE0 = align_up((size_t)option(OPT_MF_HOST_PIECE_LAGS), (size_t)MF_LAGS_PER_WG);
    // (not under mf.split16: a channel's scale is its maximum over the WHOLE day)
    const bool pieces = !rc && !from_peer && use_mfma && option(OPT_MF_COMPAT_SEQUENTIAL_CSUM) == 0 &&
                        option(OPT_MF_SPLIT16) == 0 && PIECE0 != 0 && N >= 8 * PIECE0;
    auto launch_range = [&](size_t b, long long off_lo, long long off_hi, bool cont) {
        const size_t t0 = bt[b], nt = bt[b + 1] - bt[b];
        char* d_out = base + ((b & 1) ? o_out1 : o_out0);
        struct Scope {
            Scope(long long lo, long long hi, bool c) { t_mf_off_lo = lo; t_mf_off_hi = hi; t_mf_continue = c; }
            ~Scope() { t_mf_off_lo = 0; t_mf_off_hi = -1; t_mf_continue = false; }
        } scope(off_lo, off_hi, cont);
        return bpmf_mf_run_dev((const float*)(base + o_tp) + t0 * n_ch * L,
                               (const int32_t*)(base + o_mv) + t0 * n_ch,
                               (const float*)(base + o_w) + t0 * n_ch, (const float*)(base + o_d),
                               step, L, N, nt, S, C, n_corr, network_sum,
                               (flags & ~BPMF_MF_DATA_PREPARED) | BPMF_MF_DATA_PREPARED,
                               base + o_ws, b_ws, s_run, (float*)d_out);
    };
    size_t n_streamed = 0;            // batches computed while the data arrived (their events are recorded)
    if (pieces) {
        // ONE batch is computed behind the arriving pieces (measured, round 5: with two, both output buffers are
        // busy when the last piece has landed and batch 2 cannot start before batch 0 has been drained -- 23 ms of
        // idle device in the kernel trace of a cfg2 call; one batch of ~1 GB of output computes for longer than the
        // day takes to arrive, so it hides the whole upload, and batch 1 follows it without a gap)
        n_streamed = 1;
        // largest moveout of a weighted channel per streamed batch: lag block [.., B) reads data up to
        // B + mv_max + L (+ the staging slack of its last wave)
        long long mv_max[2] = {0, 0};
        for (size_t b = 0; b < n_streamed; ++b) {
            const size_t t0 = bt[b], nt = bt[b + 1] - bt[b];
            bool any = false;
            for (size_t i = t0 * n_ch; i < (t0 + nt) * n_ch; ++i)
                if (weights[i] != 0.0f && (!any || moveouts[i] > mv_max[b])) { mv_max[b] = moveouts[i]; any = true; }
            mv_max[b] = std::max<long long>(mv_max[b], 0);
        }
        // (what no piece has brought yet reads as zeros -- finite -- for the loads that run past a lag block's own windows)
        MF_TRY(hipMemsetAsync(base + o_d, 0, b_d, s_run), "memset");
        MF_TRY(hipEventRecord(ctx->ev_chunk[0], s_run), "event record");
        MF_TRY(hipStreamWaitEvent(s_copy, ctx->ev_chunk[0], 0), "wait event");
        const MfWorkspace wsd = mf_carve(base + o_ws, L, N, std::min(TB, T), n_ch);
        size_t have = 0, piece = PIECE0;
        long long done[2] = {0, 0};
        int n_piece = 1;
        while (have < N && !rc) {
            size_t upto = std::min(N, have + piece);
            if (N - upto < PIECE0) upto = N;                      // no sliver at the end
            piece = std::min(piece * 2, (size_t)8 * PIECE0);      // 0.13 M, 0.26 M, 0.5 M, then 1 M samples
            MF_TRY(staged_upload_rows(ctx, (float*)(base + o_d), data, n_ch, N, have, upto, s_copy), "H2D data");
            hipEvent_t ev = ctx->ev_chunk[n_piece++ % DeviceContext::CHUNK_EVENTS];
            MF_TRY(hipEventRecord(ev, s_copy), "event record");
            MF_TRY(hipStreamWaitEvent(s_run, ev, 0), "wait event");
            if (upto == N) MF_TRY(fanout_publish(fan, ctx, base + o_d, s_copy), "event record");
            if (!rc) rc = mf_prepare_range((const float*)(base + o_d), L, N, n_ch, wsd, s_run, have, upto);
            have = upto;
            for (size_t b = 0; b < n_streamed && !rc; ++b) {
                long long hi = have == N ? (long long)n_offsets
                                         : ((long long)have - mv_max[b] - (long long)L - 2 * MF_LAGS_PER_WG) / MF_LAGS_PER_WG * MF_LAGS_PER_WG;
                hi = std::min<long long>(hi, (long long)n_offsets);
                if (hi <= done[b]) continue;
                rc = launch_range(b, done[b], hi, done[b] > 0);
                if (t_call_stats.first_kernel_ms == 0.0) t_call_stats.first_kernel_ms = host_now_ms() - t_call0;
                done[b] = hi;
            }
        }
        for (size_t b = 0; b < n_streamed && !rc; ++b) MF_TRY(hipEventRecord(ev_batch[b & 1], s_run), "event record");
    } else {
        if (!rc && !from_peer) {
            // (through the pinned pieces on the copy stream: the runtime's own path page-locks a host region it
            // has not seen before -- a new day is a new array -- at a third of the rate, context.h)
            MF_TRY(hipEventRecord(ctx->ev_chunk[0], s_run), "event record");        // behind the small uploads above
            MF_TRY(hipStreamWaitEvent(s_copy, ctx->ev_chunk[0], 0), "wait event");
            MF_TRY(staged_upload_rows(ctx, (float*)(base + o_d), data, n_ch, N, 0, N, s_copy), "H2D data");
            if (!rc) MF_TRY(fanout_publish(fan, ctx, base + o_d, s_copy), "event record");
            MF_TRY(hipEventRecord(ctx->ev_chunk[1], s_copy), "event record");
            MF_TRY(hipStreamWaitEvent(s_run, ctx->ev_chunk[1], 0), "wait event");
        }
        if (!rc)
            rc = bpmf_mf_prepare_data_dev((const float*)(base + o_d), L, N, S, C, base + o_ws, b_ws, s_run);
    }
    auto launch = [&](size_t b) {
        if (b < n_streamed) return 0;                            // computed while the data arrived
        int r = launch_range(b, 0, -1, false);
        if (!r) MF_TRY(hipEventRecord(ev_batch[b & 1], s_run), "event record");
        return r;
    };
    const bool verbose = option(OPT_MF_VERBOSE) != 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    double t_wait = 0.0, t_copy = 0.0;
    if (!rc) rc = launch(0);
    if (t_call_stats.first_kernel_ms == 0.0) t_call_stats.first_kernel_ms = host_now_ms() - t_call0;
    for (size_t b = 0; b < n_batch && !rc; ++b) {
        if (b + 1 < n_batch) rc = launch(b + 1);   // its buffer was drained one iteration ago
        if (rc) break;
        const size_t t0 = bt[b], nt = bt[b + 1] - bt[b], bytes = nt * row_bytes;
        const char* d_out = base + ((b & 1) ? o_out1 : o_out0);
        char* h_out = (char*)cc_out + t0 * row_bytes;
        MF_TRY(hipStreamWaitEvent(s_copy, ev_batch[b & 1], 0), "wait event");
        const size_t n_piece = (bytes + PIECE - 1) / PIECE;
        auto enqueue = [&](size_t q) {
            const size_t o = q * PIECE, len = std::min(PIECE, bytes - o);
            MF_TRY(hipMemcpyAsync(pinned[q & 1], d_out + o, len, hipMemcpyDeviceToHost, s_copy), "D2H cc");
            MF_TRY(hipEventRecord(ev_piece[q & 1], s_copy), "event record");
        };
        if (!rc && n_piece) enqueue(0);
        for (size_t q = 0; q < n_piece && !rc; ++q) {
            const double t0w = now();
            MF_TRY(hipEventSynchronize(ev_piece[q & 1]), "event sync");
            const double t1w = now();
            if (q + 1 < n_piece && !rc) enqueue(q + 1);   // into the other pinned buffer
            if (!rc) {
                const size_t o = q * PIECE, len = std::min(PIECE, bytes - o);
                parallel_copy(h_out + o, pinned[q & 1], len);
            }
            t_wait += t1w - t0w;
            t_copy += now() - t1w;
        }
    }
    // (always drained, also after a failure: the working set goes back to its cache)
    (void)hipStreamSynchronize(s_run);
    (void)hipStreamSynchronize(s_copy);
    t_call_stats.device_wait_ms = t_wait * 1e3;      // (waiting for drained pieces of the CC matrix)
    t_call_stats.total_ms = host_now_ms() - t_call0;
    if (verbose)
        fprintf(stderr, "[bpmf] mf_run: %zu batches of at most %zu templates, %.3f s after setup: waiting for the "
                        "device %.3f s, host copies %.3f s\n", n_batch, TB, now() - t_start, t_wait, t_copy);
#undef MF_TRY
    copy_pool_quiesce();  // no host thread of the copy pool still reads the caller's arrays (a straggler of an idempotent fill) when the call returns
    fan.finish();         // a source's peers are through with its copy of the day before the working set may go
    ctx->trim_after_call();
    return rc;
}

extern "C" int bpmf_mf_run(const float* templates, const int32_t* moveouts, const float* weights,
                           const float* data, size_t step, size_t L, size_t N, size_t T, size_t S,
                           size_t C, size_t n_corr, int network_sum, int flags, int device,
                           float* cc_out)
{
    // nothing may cross the C boundary as an exception (std::bad_alloc from the host-side planning, a
    // std::system_error): it becomes status -3 with its text
    try {
        return bpmf_mf_run_impl(templates, moveouts, weights, data, step, L, N, T, S, C, n_corr, network_sum, flags, device, cc_out);
    } catch (const std::exception& e) {
        copy_pool_quiesce();      // (no pool thread may still read the caller's arrays)
        set_error("bpmf_mf_run: exception: %s", e.what());
        return -3;
    } catch (...) {
        copy_pool_quiesce();
        set_error("bpmf_mf_run: unknown exception");
        return -3;
    }
}

#ifdef BPMF_PHASE_CYCLES
// {4 phase sums, channels, waves, -, -}; reset != 0 clears the counters afterwards
extern "C" int bpmf_phase_read_mf(unsigned long long* out8, int reset)
{
    BPMF_HIP_CHECK(hipDeviceSynchronize());
    BPMF_HIP_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(bpmf::g_mf_phase), sizeof(bpmf::g_mf_phase)));
    if (reset) {
        unsigned long long z[8] = {};
        BPMF_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(bpmf::g_mf_phase), z, sizeof(z)));
    }
    return 0;
}
#endif
