// Matched filter, split-precision numerators (option mf.split16; OFF by default -- the default path is the
// exact-fp32 MFMA kernel of mf.hip, bit-identical to the oracle).
//
// Serves the same call as mf.hip (fast_matched_filter.matched_filter at BPMF/similarity_search.py:526-533): the
// caller scrubs NaNs and thresholds the CC sums (:540, :615-618), it never relies on their last bits; the north
// star asks for a float32 TOLERANCE on CC values and exact detection indices.  This kernel keeps everything of
// the fp32 path except the numerator's arithmetic: norms (double prefix sums, r_t, r_d), valid lag ranges, the
// "0 where r_t * r_d >= 1000" rule, the weighted channel sum in channel order.
//
// Numerator.  Every sample x (of the data and of the templates, each channel scaled by a power of two so that
// its largest magnitude lies in [2^14, 2^15)) is split as  x = hi + lo + e,  hi = fp16(x), lo = x - hi rounded to
// fp16, |e| <= 2^-22 |x|.  Three fp16 MFMA products accumulate in ONE fp32 accumulator,
//     num = sum (hi_t hi_d + hi_t lo_d + lo_t hi_d),
// leaving out only the lo * lo term (<= 2^-22 |t||d| per term): |d num| <~ 3 * 2^-22 * sum |t d|  <=  7.2e-7 *
// sqrt(E_t E_d), i.e. a CC error of the size of the fp32 chain's own rounding and far inside 2e-5 (SURVEY
// App. C MF-5).  v_mfma_f32_32x32x16_f16 runs at 16x the rate of the exact-fp32 MFMA; with three products 16/3.
// The DATA's lo is stored scaled by 2^11 (lo' = fp16((x - hi) * 2^11), of the magnitude of x itself): stored plain it
// would be an fp16 subnormal for every sample below 2^-17 of its channel's maximum and lose its bits -- a quiet
// window of a channel that also holds a glitch 10^7 times larger came out 3e-4 wrong (found by the fuzz sweep, seed
// 560).  The product that consumes it, hi_t * lo_d, takes hi_t * 2^-11 instead (one v_pk_mul_f16 per A register
// and k-step: the VALU is idle beside the MFMAs; a template's coefficients span a few binades, the scaled ones stay
// normal).  With that the split resolves a sample down to 2^-50 of its channel's maximum.
//
// Tile algebra (v_mfma_f32_32x32x16_f16): one 32 x 32 tile = 1024 consecutive lags of one (template, channel):
//     Out[b][a] = sum_m A[b][m] * D[m][a],   lag = 32 a + b
//     A[b][m] = tmpl[m - b - r]   (32 x (L + 38) Toeplitz band, zeros outside 0 <= m - b - r < L)
//     D[m][a] = data[xa + 32 a + m],   xa = the window's start rounded DOWN to a multiple of 8 samples,
//                                      r = start - xa  (so that every 16-byte fragment of 8 samples is aligned)
// A wave owns NT = 2 tiles (2048 lags) with one accumulator each that share the A fragments, stages ITS OWN window and band (no barrier in the channel loop, single-buffered: one
// wave's LDS operations execute in order), and walks the used channels of its template with the weighted CC
// sums in registers -- the structure of mf_mfma_wave_kernel.
//
// Layout in HBM (prepared once per day / per template batch):
//   split data  [channel][q][2][8] fp16: q-chunk q = samples 8q .. 8q+7 as 8 hi values then the 8 lo values times 2^11 (32 B);
//               a window is ONE contiguous stream of 16-byte chunks.
//   band image  [template][channel][4096 B]: the LDS image of the band, per plane (hi: bytes 0.., lo: 2048..)
//               an EVEN copy E (dword n = elements 2n, 2n+1 of Br) and an ODD copy O (dword n = elements 2n+1,
//               2n+2) with Br[i] = tmpl[i - 40 - r] * 2^s: a lane whose first element is even reads 4 dwords of
//               E, one whose first element is odd 4 dwords of O (ds_read2_b32 x 2; 16-byte alignment would need
//               8 copies).  O starts 17 dwords (mod 32) behind E: the 16 even and 16 odd lanes of a half wave
//               hit 32 different banks.
// LDS window layout: stream chunk c (16 B) at slot c + (c >> 3) (one pad slot per 8: a row of 32 samples = 8
// chunks = 144 B; the 16 lanes of a ds_read_b128 group, 144 B apart, tile the 64 banks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace bpmf {
namespace sp {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 2;                       // 32 x 32 tiles per wave
constexpr int LAGS_W = 1024 * NT;           // lags per wave
constexpr int WAVES = 4;
constexpr int THREADS = 64 * WAVES;
constexpr int LAGS_WG = WAVES * LAGS_W;     // 8192
constexpr int W_LOADS = 10;                 // 16-byte stream chunks of the window per lane
constexpr int W_CHUNKS = 64 * W_LOADS;      // 640 chunks = 320 q-chunks = 2560 samples
constexpr int W_BYTES = (W_CHUNKS + W_CHUNKS / 8) * 16;     // 11 520
constexpr int B_LOADS = 4;                  // 16-byte chunks of the band image per lane
constexpr int BAND_BYTES = 64 * B_LOADS * 16;               // 4096
constexpr int BAND_PLANE = BAND_BYTES / 2;                  // 2048: the lo plane
constexpr int WAVE_LDS = W_BYTES + BAND_BYTES;              // 15 616
constexpr int WG_LDS = WAVES * WAVE_LDS;                    // 62 464: two workgroups per CU
constexpr int MAX_KS = 26;                  // k-steps of 16: L + 38 <= 416
constexpr int BAND_LEAD = 40;               // Br[i] = tmpl[i - BAND_LEAD - r]
constexpr int S_TARGET = 14;                // largest magnitude of a channel scaled into [2^14, 2^15)
constexpr int S_CLAMP = 60;                 // |s| <= 60: the product of a template's and a channel's 2^-s stays a normal float
constexpr float SP_MAX_NORM = 1000.0f;      // = MAX_NORM of common.h: r_t * r_d >= this -> CC = 0

__host__ __device__ inline int nks_of(int L) { return (L + 38 + 15) / 16; }
__host__ __device__ inline int max_segment_len() { return (16 * MAX_KS - 38) / 8 * 8; }   // 376: what one band image holds
// Templates longer than a band image are correlated in SEGMENTS of equal length (a multiple of 8 samples, so that every
// segment's window keeps the remainder r of its channel): n_seg K loops per channel into the same accumulators, the
// window and the band image re-staged per segment, the epilogue behind the last one.
__host__ __device__ inline int n_segments_of(int L) { return (L + max_segment_len() - 1) / max_segment_len(); }
__host__ __device__ inline int segment_len_of(int L)
{
    const int n = n_segments_of(L);
    return ((L + n - 1) / n + 7) / 8 * 8;
}
__host__ __device__ inline int max_template_len() { return 4096; }
__host__ __device__ inline int band_e_dwords(int nks) { return 8 * nks + 32; }
// first dword index >= the length of E that is 17 (mod 32)
__host__ __device__ inline int band_o_off(int nks)
{
    const int e = band_e_dwords(nks);
    return e + ((17 - e % 32) + 32) % 32;
}
__host__ __device__ inline size_t split_row_bytes(size_t N) { return ((N + 7) / 8) * 32; }

// power-of-two scale exponent of a channel from the bits of its largest magnitude (0: all zero or non-finite)
__host__ __device__ inline int scale_exp_of(unsigned maxbits)
{
    if (maxbits == 0u || maxbits >= 0x7f800000u) return 0;
    const int e = (int)(maxbits >> 23) - 127;           // (subnormal maximum: -127)
    int s = S_TARGET - e;
    if (s > S_CLAMP) s = S_CLAMP;
    if (s < -S_CLAMP) s = -S_CLAMP;
    return s;
}
__host__ __device__ inline float pow2f(int s)           // 2^s, |s| <= 126
{
    union { unsigned u; float f; } v;
    v.u = (unsigned)(s + 127) << 23;
    return v.f;
}

constexpr float LO_SCALE = 2048.0f;          // the data's lo halves are stored times 2^11 (see above)

// x * 2^s -> (hi, lo * lo_scale) as two fp16 bit patterns (lo_scale: LO_SCALE for data, 1 for templates)
__device__ __forceinline__ void split_one(float v, float lo_scale, unsigned short& hi, unsigned short& lo)
{
    const _Float16 h = (_Float16)v;
    const float res = v - (float)h;                     // exact
    const _Float16 l = (_Float16)(res * lo_scale);      // (a power of two: exact)
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
}

// ---------------------------------------------------------------------- per-day preparation
// largest |x| of every channel (bits; a NaN ends up above Inf): grid (blocks, n_ch)
__global__ __launch_bounds__(256) void sp_absmax_kernel(const float* __restrict__ data, size_t N,
                                                        unsigned* __restrict__ maxbits)
{
    const float* d = data + (size_t)blockIdx.y * N;
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (size_t)gridDim.x * 256)
        m = max(m, __float_as_uint(d[i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(maxbits + blockIdx.y, m);
}

// scd[ch] = 2^-s, s_exp[ch] = s
__global__ void sp_scale_kernel(const unsigned* __restrict__ maxbits, int n_ch, int* __restrict__ s_exp,
                                float* __restrict__ scd)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= n_ch) return;
    const int s = scale_exp_of(maxbits[ch]);
    s_exp[ch] = s;
    scd[ch] = pow2f(-s);
}

// one thread per q-chunk (8 samples): grid (ceil(NQ / 256), n_ch)
__global__ __launch_bounds__(256) void sp_split_data_kernel(const float* __restrict__ data, size_t N, size_t NQ,
                                                            const int* __restrict__ s_exp, u32x4* __restrict__ out)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= NQ) return;
    const size_t ch = blockIdx.y;
    const float* d = data + ch * N + 8 * q;
    const float sc = pow2f(s_exp[ch]);
    float v[8];
    if (8 * q + 8 <= N) {
        const f32x4u a = *(const f32x4u*)d, b = *(const f32x4u*)(d + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 8 * q + i < N ? d[i] : 0.0f;
    }
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) split_one(v[i] * sc, LO_SCALE, hi[i], lo[i]);
    u32x4 oh, ol;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        oh[i] = (unsigned)hi[2 * i] | ((unsigned)hi[2 * i + 1] << 16);
        ol[i] = (unsigned)lo[2 * i] | ((unsigned)lo[2 * i + 1] << 16);
    }
    u32x4* o = out + (ch * NQ + q) * 2;
    o[0] = oh;
    o[1] = ol;
}

// ------------------------------------------------------------------ per-template preparation
// One wave per (template, channel, segment): the 4 KB band image and sct[t, ch] = 2^-s (one scale per template channel).
__global__ __launch_bounds__(64) void sp_band_kernel(const float* __restrict__ tmpl, const int* __restrict__ mv,
                                                     int L, int n_seg, int seg_len, unsigned* __restrict__ bands,
                                                     float* __restrict__ sct)
{
    const size_t tc = blockIdx.x / (unsigned)n_seg;
    const int seg = (int)(blockIdx.x % (unsigned)n_seg);
    const int lane = threadIdx.x;
    const float* x = tmpl + tc * (size_t)L;
    unsigned m = 0;
    for (int l = lane; l < L; l += 64) m = max(m, __float_as_uint(x[l]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    const int s = scale_exp_of(m);
    const float sc = pow2f(s);
    if (lane == 0 && seg == 0) sct[tc] = pow2f(-s);
    const int mvc = mv[tc];
    const int r = ((mvc % 8) + 8) % 8;
    const int nks = nks_of(seg_len);
    const int e_d = band_e_dwords(nks), o_off = band_o_off(nks);
    unsigned* img = bands + (size_t)blockIdx.x * (BAND_BYTES / 4);
    // element i of this segment's Br (zero outside the segment / the template), plane p
    auto elem = [&](int i, int p) -> unsigned {
        const int ls = i - BAND_LEAD - r;
        const int l = seg * seg_len + ls;
        if (ls < 0 || ls >= seg_len || l >= L) return 0u;
        unsigned short hi, lo;
        split_one(x[l] * sc, 1.0f, hi, lo);
        return p ? lo : hi;
    };
    for (int w = lane; w < BAND_BYTES / 4; w += 64) {
        const int p = w >= BAND_PLANE / 4 ? 1 : 0;
        const int n = w - p * (BAND_PLANE / 4);
        unsigned val = 0u;
        if (n < e_d) val = elem(2 * n, p) | (elem(2 * n + 1, p) << 16);
        else if (n >= o_off && n < o_off + e_d) {
            const int k = n - o_off;
            val = elem(2 * k + 1, p) | (elem(2 * k + 2, p) << 16);
        }
        img[w] = val;
    }
}

// ---------------------------------------------------------------------------- main kernel
// XCD-aware workgroup order, as mf_tile_of_block (mf.hip): the (lag block, template) pairs, template fastest, cut
// into 8 contiguous runs, one per XCD.
__device__ __forceinline__ bool sp_tile_of_block(unsigned bid, int T, int n_lag_blocks, int& t, long long& lag_block)
{
    const unsigned xcd = bid & 7u, i = bid >> 3;
    const unsigned total = (unsigned)n_lag_blocks * (unsigned)T;
    const unsigned per_xcd = (total + 7u) >> 3;
    const unsigned flat = xcd * per_xcd + i;
    t = (int)(flat % (unsigned)T);
    lag_block = (long long)(flat / (unsigned)T);
    return i < per_xcd && flat < total;
}

#define SP_SB __builtin_amdgcn_sched_barrier(0)
#define SP_RD2(dst, addr, o0) \
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "n"(o0), "n"((o0) + 1))
#define SP_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

struct Frags {
    i32x2 a[2][2];          // [plane][half]
    i32x4 b[NT][2];         // [tile][plane]
};

__device__ __forceinline__ f16x8 sp_h8(i32x2 lo, i32x2 hi)
{
    const i32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(f16x8, v);
}
__device__ __forceinline__ f16x8 sp_h8(i32x4 v) { return __builtin_bit_cast(f16x8, v); }

// chan_rec: the records of mf_prologue_kernel {channel, moveout, weight bits, r_t bits} closed by {-1, ..} x 2;
// e_d: the reciprocal window norms r_d of the fp32 path; range: valid CC indices per template.
// cc = (num * 2^-(s_t + s_d)) * (r_t * r_d) where r_t * r_d < 1000, else 0; sum = fmaf(w, cc, sum).
// `prio`: bits 0-1 the wave priority experiments of the ubench (the library passes 0 there); bit 2: the norms are
// energies and cc = num / sqrtf(E_t * E_d) where the product exceeds 1e-6, else 0 (option mf.compat_sqrt_norm).
// ABLATE (tools/ubench/mfma_split16.hip only; the library instantiates 0): 1 = no norms / scaling in the epilogue,
// 2 = also no staging (the K loop alone, on whatever the LDS holds)
// SEGMENTED: templates of more than one segment (n_seg > 1); false folds the segment logic away
template <bool NETWORK_SUM, bool STEP1, int ABLATE = 0, bool SEGMENTED = false>
__global__ __launch_bounds__(THREADS, 2) void mf_split_kernel(
    const u32x4* __restrict__ sdata, const unsigned* __restrict__ bands, const float* __restrict__ sct,
    const float* __restrict__ scd, const int4* __restrict__ chan_rec, const float* __restrict__ e_d,
    const int2* __restrict__ range, int L, long long N, int T, int n_ch, long long n_corr, int step,
    float* __restrict__ out, int n_lag_blocks, int lag_block0, int prio, int n_seg, int seg_len)
{
    extern __shared__ __attribute__((aligned(16))) char sp_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a = lane & 31;          // tile column (B operand) / band row (A operand)
    const int g = lane >> 5;          // k group of the operands / row group of the results
    const int nks = nks_of(seg_len);
    // prio bit 1 (experiment): every other workgroup runs its K loops at a raised issue priority
    const bool kprio = (((blockIdx.x >> 3) ^ (blockIdx.x >> 8)) & 1u) != 0;

    int t;
    long long lag_block;
    if (!sp_tile_of_block(blockIdx.x, T, n_lag_blocks, t, lag_block)) return;
    lag_block += lag_block0;
    const int2 rgi = range[t];
    const long long lag0 = lag_block * LAGS_WG + (long long)wv * LAGS_W;
    const int2 rg = make_int2(rgi.x * step, rgi.y * step);
    const long long nwin = N - L + 1;
    const size_t row_bytes = split_row_bytes((size_t)N);

    f32x16 sum[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[u][r] = 0.0f;

    const bool wave_valid = rg.x <= rg.y && !(lag0 > rg.y || lag0 + LAGS_W - 1 < rg.x);
    const bool wave_inside = lag0 >= rg.x && lag0 + LAGS_W - 1 <= rg.y;
    // result (tile u, register r) of this lane: lag_w + 1024 u + 8 (r >> 2) + (r & 3)
    const long long lag_w = lag0 + 32 * a + 4 * g;

    if (wave_valid) {
        char* wbase = sp_smem + wv * WAVE_LDS;          // window
        char* bbase = wbase + W_BYTES;                  // band image
        const unsigned w_st = (unsigned)(size_t)wbase + 16u * (unsigned)(lane + (lane >> 3));   // staging store address
        const unsigned b_st = (unsigned)(size_t)bbase + 16u * (unsigned)lane;
        // operand read addresses
        const unsigned b_rd = (unsigned)(size_t)wbase + 144u * (unsigned)a + 32u * (unsigned)g;
        const int odd = a & 1;
        const int a_dw = 4 * g - ((a + 1) >> 1) + BAND_LEAD / 2 + (odd ? band_o_off(nks) : 0);
        const unsigned a_rd = (unsigned)(size_t)bbase + 4u * (unsigned)a_dw;
        const int4* __restrict__ recs = chan_rec + (size_t)t * (n_ch + 2);

        u32x4 rd[W_LOADS], rt[B_LOADS];
        auto issue_stage = [&](int ch, int mvc, int sg) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((const char*)sdata + (size_t)ch * row_bytes), 0, (int)row_bytes, 0x00020000);
            // floor((lag0 + mvc + sg * seg_len) / 8): lag0 and seg_len are multiples of 8
            const long long q0 = (lag0 >> 3) + (long long)(mvc >> 3) + (long long)sg * (seg_len >> 3);
            const unsigned o = (unsigned)(q0 * 32 + 16 * lane);           // wraps like the hardware's offset
            if (q0 >= 0) {
#pragma unroll
                for (int i = 0; i < W_LOADS; ++i)
                    rd[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(o + 1024u * i), 0, 0));
            } else {
                unsigned oo = o;
#pragma unroll
                for (int i = 0; i < W_LOADS; ++i) {
                    asm volatile("" : "+v"(oo));       // keep the constant out of the immediate field (mf_stage_rows)
                    rd[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)oo, 0, 0));
                    oo += 1024u;
                }
            }
            const u32x4* bi = (const u32x4*)(bands + (((size_t)t * n_ch + ch) * n_seg + sg) * (BAND_BYTES / 4)) + lane;
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) rt[i] = bi[64 * i];
        };
        auto write_stage = [&]() {
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i)
                asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(b_st), "v"(rt[i]), "n"(1024 * i) : "memory");
#pragma unroll
            for (int i = 0; i < W_LOADS; ++i)
                asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(w_st), "v"(rd[i]), "n"(1152 * i) : "memory");
        };

        int4 rec = recs[0];
        int4 rec1 = recs[1];
        int ri = 0, seg = 0;
        f32x16 acc[NT];
        f32x4 ed[NT][4];
        if (ABLATE < 2 && rec.x >= 0) issue_stage(rec.x, rec.y, 0);
        while (rec.x >= 0) {
            const int ch = rec.x;
            const bool last_seg = !SEGMENTED || seg == n_seg - 1;
            if constexpr (ABLATE < 2) write_stage();
            const float w = __int_as_float(rec.z);
            const int mvc = rec.y;
            const float rt_n = __int_as_float(rec.w);
            const float s_t = sct[(size_t)t * n_ch + ch];
            const float s_d = scd[ch];
            const float s_td = s_t * s_d;              // (exact: powers of two, |exponent| <= 120)
            const int4 rec2 = recs[ri + 2];
            const float* edc = e_d + (size_t)ch * (size_t)nwin;
            // (the window norms are needed behind the channel's LAST segment only)
            if (!last_seg) {
            } else if constexpr (ABLATE >= 1) {
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) ed[u][i] = (f32x4){1.0f, 1.0f, 1.0f, 1.0f};
            } else if (wave_inside) {
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) ed[u][i] = *(const f32x4u*)(edc + lag_w + 1024 * u + 8 * i + mvc);
            } else {
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const long long lag = lag_w + 1024 * u + 8 * i;
                        if (lag + 3 >= rg.x && lag <= rg.y) ed[u][i] = *(const f32x4u*)(edc + lag + mvc);
                        else ed[u][i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                    }
            }
            // the next thing to correlate: this channel's next segment, or the next channel's first
            if constexpr (ABLATE < 2) {
                if (!last_seg) issue_stage(rec.x, rec.y, seg + 1);
                else if (rec1.x >= 0) issue_stage(rec1.x, rec1.y, 0);
            }

            if (!SEGMENTED || seg == 0) {
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
            }

            unsigned ap = a_rd, ap2 = a_rd + BAND_PLANE, bp = b_rd;
            Frags f[2];
// the reads of the first k-step, in the order SP_STEP expects them
#define SP_REQ(s, dw, off)                                                    \
    SP_RD2(f[s].a[0][0], ap, (dw));     SP_RD2(f[s].a[0][1], ap, (dw) + 2);   \
    SP_RD128(f[s].b[0][0], bp, (off));         SP_RD128(f[s].b[1][0], bp, (off) + 4608);      \
    SP_RD128(f[s].b[0][1], bp, (off) + 16);    SP_RD128(f[s].b[1][1], bp, (off) + 4608 + 16); \
    SP_RD2(f[s].a[1][0], ap2, (dw));    SP_RD2(f[s].a[1][1], ap2, (dw) + 2)
#define SP_MFMA(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0)
// One k-step: the 6 MFMAs of slot `cur` with the 8 operand reads of slot `nxt` (the next k-step) between them.
// The reads of a k-step are issued in the order its MFMAs consume them -- A hi (2), B hi of tile 0, of tile 1, B lo
// of tile 0, of tile 1, A lo (2) -- one k-step (6 MFMAs = 192 matrix-pipe cycles) ahead of their use, and LDS
// returns in order, so every MFMA waits with a COUNT: on entry the 8 reads R1..R8 of `cur` are outstanding;
// lgkmcnt(5) = R1..R3 have landed; behind the first MFMA two reads of `nxt` join the queue, so R4 has landed at
// lgkmcnt(6), and so on (at most 10 in flight; a scalar load in flight only makes a count stricter).
#define SP_WAIT(n) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(n) : "memory")
#define SP_STEP(cur, nxt, dw, off)                                                          \
    {                                                                                        \
        SP_WAIT(5); SP_SB;                                                                   \
        const f16x8 ah = sp_h8(f[cur].a[0][0], f[cur].a[0][1]);                              \
        const f16x8 hs = ah * (_Float16)(1.0f / LO_SCALE);      /* for the product with the data's lo * 2^11 */ \
        SP_MFMA(acc[0], ah, sp_h8(f[cur].b[0][0]));                                          \
        SP_SB;                                                                               \
        SP_RD2(f[nxt].a[0][0], ap, (dw)); SP_RD2(f[nxt].a[0][1], ap, (dw) + 2);              \
        SP_WAIT(6); SP_SB;                                                                   \
        SP_MFMA(acc[1], ah, sp_h8(f[cur].b[1][0]));                                          \
        SP_SB;                                                                               \
        SP_RD128(f[nxt].b[0][0], bp, (off)); SP_RD128(f[nxt].b[1][0], bp, (off) + 4608);     \
        SP_WAIT(7); SP_SB;                                                                   \
        SP_MFMA(acc[0], hs, sp_h8(f[cur].b[0][1]));                                          \
        SP_SB;                                                                               \
        SP_RD128(f[nxt].b[0][1], bp, (off) + 16); SP_RD128(f[nxt].b[1][1], bp, (off) + 4608 + 16); \
        SP_WAIT(8); SP_SB;                                                                   \
        SP_MFMA(acc[1], hs, sp_h8(f[cur].b[1][1]));                                          \
        SP_SB;                                                                               \
        SP_RD2(f[nxt].a[1][0], ap2, (dw)); SP_RD2(f[nxt].a[1][1], ap2, (dw) + 2);            \
        SP_WAIT(8); SP_SB;                                                                   \
        const f16x8 al = sp_h8(f[cur].a[1][0], f[cur].a[1][1]);                              \
        SP_MFMA(acc[0], al, sp_h8(f[cur].b[0][0]));                                          \
        SP_SB;                                                                               \
        SP_MFMA(acc[1], al, sp_h8(f[cur].b[1][0]));                                          \
        SP_SB;                                                                               \
    }
// the odd last k-step: nothing behind it to read ahead for (R1..R8 outstanding on entry, none joins)
#define SP_STEP_LAST(cur)                                                                    \
    {                                                                                        \
        SP_WAIT(5); SP_SB;                                                                   \
        const f16x8 ah = sp_h8(f[cur].a[0][0], f[cur].a[0][1]);                              \
        const f16x8 hs = ah * (_Float16)(1.0f / LO_SCALE);                                   \
        SP_MFMA(acc[0], ah, sp_h8(f[cur].b[0][0]));                                          \
        SP_WAIT(4); SP_SB;                                                                   \
        SP_MFMA(acc[1], ah, sp_h8(f[cur].b[1][0]));                                          \
        SP_WAIT(3); SP_SB;                                                                   \
        SP_MFMA(acc[0], hs, sp_h8(f[cur].b[0][1]));                                          \
        SP_WAIT(2); SP_SB;                                                                   \
        SP_MFMA(acc[1], hs, sp_h8(f[cur].b[1][1]));                                          \
        SP_WAIT(0); SP_SB;                                                                   \
        const f16x8 al = sp_h8(f[cur].a[1][0], f[cur].a[1][1]);                              \
        SP_MFMA(acc[0], al, sp_h8(f[cur].b[0][0]));                                          \
        SP_SB;                                                                               \
        SP_MFMA(acc[1], al, sp_h8(f[cur].b[1][0]));                                          \
        SP_SB;                                                                               \
    }
            SP_SB;
            if (prio & 1) __builtin_amdgcn_s_setprio(0);
            if ((prio & 2) && kprio) __builtin_amdgcn_s_setprio(2);
            SP_REQ(0, 0, 0);
            const int npair = nks >> 1;
            for (int p = 0; p < npair; ++p) {
                SP_STEP(0, 1, 8, 64);         // k-step 2p, requests 2p + 1
                SP_STEP(1, 0, 16, 144);       // k-step 2p + 1, requests 2p + 2
                ap += 64;
                ap2 += 64;
                bp += 144;
            }
            if (nks & 1) { SP_STEP_LAST(0); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SP_SB;
            if (prio & 1) __builtin_amdgcn_s_setprio(1);
            else if (prio & 2) __builtin_amdgcn_s_setprio(0);
#undef SP_REQ
#undef SP_WAIT
#undef SP_MFMA
#undef SP_STEP
#undef SP_STEP_LAST

            if (!last_seg) {            // more of this template channel to come: the accumulators go on
                ++seg;
                continue;
            }
            seg = 0;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float num = acc[u][r] * s_td;
                    const float nrm = rt_n * ed[u][r >> 2][r & 3];
                    float cc;
                    if (prio & 4)       // mf.compat_sqrt_norm: the stored norms are the energies E_t and E_d
                        cc = nrm > 1.0e-6f ? num / sqrtf(nrm) : 0.0f;
                    else
                        cc = nrm < SP_MAX_NORM ? num * nrm : 0.0f;
                    if (NETWORK_SUM && STEP1 && wave_inside) {
                        sum[u][r] = __fmaf_rn(w, cc, sum[u][r]);
                    } else {
                        const long long lag = lag_w + 1024 * u + 8 * (r >> 2) + (r & 3);
                        const bool ok = lag >= rg.x && lag <= rg.y && (STEP1 || (unsigned)lag % (unsigned)step == 0);
                        if (!ok) cc = 0.0f;
                        if (!NETWORK_SUM) {
                            if (ok)
                                out[((size_t)t * n_corr + (STEP1 ? lag : (long long)((unsigned)lag / (unsigned)step))) * n_ch + ch] = cc;
                        } else {
                            sum[u][r] = __fmaf_rn(w, cc, sum[u][r]);
                        }
                    }
                }
            }
            rec = rec1;
            rec1 = rec2;
            ++ri;
        }
    }
    if (NETWORK_SUM) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long long lag = lag_w + 1024 * u + 8 * i;
                float* dst = out + (size_t)t * n_corr + lag;
                const f32x4 v = {sum[u][4 * i], sum[u][4 * i + 1], sum[u][4 * i + 2], sum[u][4 * i + 3]};
                if (STEP1 && lag + 3 < n_corr) {
                    *(f32x4u*)dst = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned smp = (unsigned)(lag + r);
                        const long long idx = STEP1 ? (long long)smp : (long long)(smp / (unsigned)step);
                        if ((STEP1 || smp % (unsigned)step == 0) && idx < n_corr) out[(size_t)t * n_corr + idx] = v[r];
                    }
                }
            }
        }
    }
}

}  // namespace sp
}  // namespace bpmf
