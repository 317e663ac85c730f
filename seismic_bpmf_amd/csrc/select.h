// Exact order statistics of float32 series on the device: a 3-level radix select (11 + 11 + 10 bits
// of an order-preserving key) through an LDS histogram, one 1024-thread workgroup per series.
// Shared by bp_detect.hip (window medians / MADs of the max-beam) and stats.hip (row and window
// medians / MADs of CC series and envelopes).  np.median's conventions: the middle order
// statistic, or the float32 mean of the two middle ones for an even count.
#pragma once
#include "common.h"

namespace bpmf {

__device__ __forceinline__ unsigned f32_key(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

constexpr int SEL_THREADS = 1024;
constexpr int SEL_BINS = 2048;

// What a series element contributes: DEV = |x - centre| instead of x (float32 subtraction, as
// NumPy evaluates np.abs(a - centre)); SKIPZ = elements equal to 0 (either sign) do not belong to
// the series (`a[a != 0]`).
template <bool DEV, bool SKIPZ>
__device__ __forceinline__ bool sel_element(float v, float centre, unsigned& key)
{
    if (SKIPZ && v == 0.0f) return false;
    if (DEV) v = fabsf(__fsub_rn(v, centre));
    key = f32_key(v);
    return true;
}

// What a loaded element is before it enters the series: nothing by default; the MAD threshold replaces exact
// zeros by white noise on the fly (stats.hip: MadFill) instead of selecting on a filled copy of the matrix.
struct SelNoFix {
    __device__ __forceinline__ float operator()(float v, long long) const { return v; }
};

// k-th smallest (0-based) key of the series x[0 .. len), counting only the elements that belong to it,
// and -- when `next` is not null -- the (k+1)-th as well, found in the same three passes whenever it
// shares the k-th one's first 22 key bits (it almost always does; *next_ok = 0 otherwise and the
// caller runs a second select).  hist: SEL_BINS counters in LDS; sel: four words of LDS for the
// hand-over between levels.  The element loop is unrolled by 8 with the loads in front: one
// workgroup streams a series at memory speed instead of one cache line per wave and round trip.
template <bool DEV, bool SKIPZ = false, class Fix = SelNoFix>
__device__ unsigned window_select(const float* __restrict__ x, long long len, float centre, unsigned rank,
                                  unsigned* hist, unsigned* sel, unsigned* next = nullptr, int* next_ok = nullptr,
                                  Fix fix = Fix(), int* nan_seen = nullptr)
{
    const int tid = threadIdx.x;
    unsigned prefix = 0;          // the key bits fixed so far (right-aligned)
    int done = 0;                 // how many
    constexpr int UNR = 8;
#pragma unroll 1
    for (int level = 0; level < 3; ++level) {
        const int nbits = level == 2 ? 10 : 11;
        const int shift = 32 - done - nbits;
        for (int b = tid; b < SEL_BINS; b += SEL_THREADS) hist[b] = 0;
        __syncthreads();
        // `nan_seen` (LDS, zeroed by the caller): raised when an element of the series is a NaN -- noticed on the
        // first level's pass over the elements, which reads them all anyway (np.median of such a series is NaN;
        // a separate pass just to look for one cost the MAD threshold a seventh of its time)
        const bool look = nan_seen != nullptr && level == 0;
        int bad = 0;
        auto count_one = [&](float v) {
            if (look) bad |= v != v;
            unsigned k;
            if (!sel_element<DEV, SKIPZ>(v, centre, k)) return;
            if (done == 0 || (k >> (32 - done)) == prefix)
                atomicAdd(&hist[(k >> shift) & ((1u << nbits) - 1)], 1u);
        };
        long long i = tid;
        for (; i + (long long)(UNR - 1) * SEL_THREADS < len; i += (long long)UNR * SEL_THREADS) {
            float v[UNR];
#pragma unroll
            for (int e = 0; e < UNR; ++e) v[e] = x[i + (long long)e * SEL_THREADS];
#pragma unroll
            for (int e = 0; e < UNR; ++e) count_one(fix(v[e], i + (long long)e * SEL_THREADS));
        }
        for (; i < len; i += SEL_THREADS) count_one(fix(x[i], i));
        if (bad) *nan_seen = 1;
        __syncthreads();
        if (tid < 64) {
            // lane l owns bins [32 l, 32 l + 32): its total, an inclusive scan over the lanes, then
            // the lane whose range holds `rank` walks its bins
            unsigned tot = 0;
            for (int b = 0; b < 32; ++b) tot += hist[tid * 32 + b];
            unsigned inc = tot;
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned o = __shfl_up(inc, d, 64);
                if (tid >= d) inc += o;
            }
            const unsigned exc = inc - tot;
            if (rank >= exc && rank < inc) {
                unsigned r = rank - exc;
                int b = 0;
                for (; b < 32; ++b) {
                    const unsigned h = hist[tid * 32 + b];
                    if (r < h) break;
                    r -= h;
                }
                sel[0] = (unsigned)(tid * 32 + b);
                sel[1] = r;
                // the (rank + 1)-th element: in the same bin, or -- at the last level, where a bin is
                // one key -- in the next non-empty bin of this lane's range; anywhere else: not here
                const unsigned h = hist[tid * 32 + b];
                int nb = -1;
                if (r + 1 < h) nb = tid * 32 + b;
                else if (level == 2) {
                    for (int b2 = b + 1; b2 < 32 && nb < 0; ++b2)
                        if (hist[tid * 32 + b2]) nb = tid * 32 + b2;
                }
                sel[2] = nb >= 0 ? (unsigned)nb : 0u;
                sel[3] = (level == 2 ? nb >= 0 : r + 1 < h) ? 1u : 0u;
            }
        }
        __syncthreads();
        const bool more = sel[3] != 0;
        if (level == 2 && next) {
            *next = (prefix << nbits) | sel[2];
            *next_ok = more ? 1 : 0;
        } else if (level < 2 && next && !more) {
            // the two order statistics part ways above the last level: let the caller select again
            // (keep going for the k-th one)
            *next_ok = 0;
            next = nullptr;
        }
        prefix = (prefix << nbits) | sel[0];
        rank = sel[1];
        done += nbits;
        __syncthreads();
    }
    return prefix;
}

// np.median of the `count` elements that belong to the series (count = len unless SKIPZ)
template <bool DEV, bool SKIPZ = false, class Fix = SelNoFix>
__device__ float window_median(const float* __restrict__ x, long long len, long long count, float centre,
                               unsigned* hist, unsigned* sel, Fix fix = Fix(), int* nan_seen = nullptr)
{
    if (count & 1)
        return key_f32(window_select<DEV, SKIPZ, Fix>(x, len, centre, (unsigned)(count / 2), hist, sel, nullptr, nullptr, fix, nan_seen));
    unsigned hi_key = 0;
    int ok = 0;
    const unsigned lo_key = window_select<DEV, SKIPZ, Fix>(x, len, centre, (unsigned)(count / 2 - 1), hist, sel, &hi_key, &ok, fix, nan_seen);
    if (!ok) hi_key = window_select<DEV, SKIPZ, Fix>(x, len, centre, (unsigned)(count / 2), hist, sel, nullptr, nullptr, fix);
    return (key_f32(lo_key) + key_f32(hi_key)) / 2.0f;   // float32 mean of the two middle values (exact halving)
}

// ---- np.median in TWO passes where the caller knows roughly where the values lie (round 5, the MAD threshold) ----
// The radix select reads the series three times per median, and its first level -- the top 11 bits of a float:
// sign, exponent, two mantissa bits -- puts a window of CC values into a handful of bins: a thousand threads
// hammering a few LDS counters.  Here pass A counts the elements in SEL_BINS buckets of equal WIDTH over
// [lo, lo + SEL_BINS width) (the caller's guess: centre +- 6 sigma of the row -- the values spread over hundreds
// of buckets); the bucket that holds the middle rank(s) has a few hundred elements, which pass B copies into LDS
// (`buf`, SEL_BUF keys) where their exact ranks are counted.  The bucket function is the same deterministic,
// monotone float expression in both passes, so the result is the exact order statistic -- bit for bit what
// window_median returns -- whatever the guess is worth; when it is worth nothing (the two middle values in
// different buckets, more than SEL_BUF elements in the bucket, a non-finite guess) *ok = 0 and the caller falls
// back to window_median.  `nan_seen` as in window_select.  No SKIPZ: every element belongs to the series.
constexpr int SEL_BUF = 4096;
template <bool DEV, class Fix = SelNoFix>
__device__ float window_median_bucketed(const float* __restrict__ x, long long len, float centre, float lo, float inv_width,
                                        unsigned* hist, unsigned* sel, unsigned* buf, Fix fix, int* nan_seen, int* ok,
                                        long long rank_lo = -1, long long rank_hi = -1)
{
    const int tid = threadIdx.x;
    constexpr int UNR = 8;
    auto value_of = [&](float v, long long i) -> float {
        v = fix(v, i);
        return DEV ? fabsf(__fsub_rn(v, centre)) : v;
    };
    auto bucket_of = [&](float v) -> int {
        if (v != v) return SEL_BINS - 1;                 // (a NaN sorts behind everything, as its key does)
        const float q = __fmul_rn(__fsub_rn(v, lo), inv_width);
        if (!(q > 0.0f)) return 0;
        return q >= (float)(SEL_BINS - 1) ? SEL_BINS - 1 : (int)q;
    };
    for (int b = tid; b < SEL_BINS; b += SEL_THREADS) hist[b] = 0;
    if (tid == 0) { sel[0] = 0; sel[1] = 0; sel[2] = 0; sel[3] = 0; }
    __syncthreads();
    int bad = 0;
    {
        long long i = tid;
        for (; i + (long long)(UNR - 1) * SEL_THREADS < len; i += (long long)UNR * SEL_THREADS) {
            float v[UNR];
#pragma unroll
            for (int e = 0; e < UNR; ++e) v[e] = x[i + (long long)e * SEL_THREADS];
#pragma unroll
            for (int e = 0; e < UNR; ++e) {
                const float u = value_of(v[e], i + (long long)e * SEL_THREADS);
                bad |= u != u;
                atomicAdd(&hist[bucket_of(u)], 1u);
            }
        }
        for (; i < len; i += SEL_THREADS) {
            const float u = value_of(x[i], i);
            bad |= u != u;
            atomicAdd(&hist[bucket_of(u)], 1u);
        }
    }
    if (bad && nan_seen) *nan_seen = 1;
    __syncthreads();
    // the bucket of rank r_lo = (len - 1) / 2 and of r_hi = len / 2 (or the two ranks the caller names: the mean
    // of order statistics rank_lo and rank_hi = rank_lo or rank_lo + 1), and the number of elements below it
    const unsigned r_lo = (unsigned)(rank_lo >= 0 ? rank_lo : (len - 1) / 2), r_hi = (unsigned)(rank_lo >= 0 ? rank_hi : len / 2);
    if (tid < 64) {
        unsigned tot = 0;
        for (int b = 0; b < 32; ++b) tot += hist[tid * 32 + b];
        unsigned inc = tot;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(inc, d, 64);
            if (tid >= d) inc += o;
        }
        const unsigned exc = inc - tot;
        if (r_lo >= exc && r_lo < inc) {
            unsigned below = exc;
            int b = 0;
            for (; b < 32; ++b) {
                const unsigned h = hist[tid * 32 + b];
                if (r_lo < below + h) break;
                below += h;
            }
            const unsigned h = hist[tid * 32 + b];
            sel[0] = (unsigned)(tid * 32 + b);           // the bucket
            sel[1] = below;                              // elements in front of it
            sel[2] = h;                                  // elements in it
            sel[3] = (r_hi < below + h && h <= (unsigned)SEL_BUF) ? 1u : 0u;     // both middle ranks inside, and it fits
        }
    }
    __syncthreads();
    const int bucket = (int)sel[0];
    const unsigned below = sel[1], in_bucket = sel[2];
    const bool usable = sel[3] != 0;
    __syncthreads();
    if (!usable || (nan_seen && *nan_seen)) { *ok = usable ? 1 : 0; return __uint_as_float(0x7fc00000u); }
    // pass B: the bucket's elements (as order-preserving keys) into LDS
    if (tid == 0) sel[0] = 0;
    __syncthreads();
    {
        long long i = tid;
        for (; i + (long long)(UNR - 1) * SEL_THREADS < len; i += (long long)UNR * SEL_THREADS) {
            float v[UNR];
#pragma unroll
            for (int e = 0; e < UNR; ++e) v[e] = x[i + (long long)e * SEL_THREADS];
#pragma unroll
            for (int e = 0; e < UNR; ++e) {
                const float u = value_of(v[e], i + (long long)e * SEL_THREADS);
                if (bucket_of(u) == bucket) buf[atomicAdd(&sel[0], 1u)] = f32_key(u);
            }
        }
        for (; i < len; i += SEL_THREADS) {
            const float u = value_of(x[i], i);
            if (bucket_of(u) == bucket) buf[atomicAdd(&sel[0], 1u)] = f32_key(u);
        }
    }
    __syncthreads();
    // exact ranks inside the bucket: element e is the k-th smallest of the bucket for every k in [less, less_or_equal)
    const unsigned k_lo = r_lo - below, k_hi = r_hi - below;
    for (unsigned e = (unsigned)tid; e < in_bucket; e += SEL_THREADS) {
        const unsigned key = buf[e];
        unsigned less = 0, leq = 0;
        for (unsigned j = 0; j < in_bucket; ++j) {
            const unsigned kj = buf[j];
            less += kj < key;
            leq += kj <= key;
        }
        if (less <= k_lo && k_lo < leq) sel[1] = key;       // (equal keys write the same value)
        if (less <= k_hi && k_hi < leq) sel[2] = key;
    }
    __syncthreads();
    const float v_lo = key_f32(sel[1]), v_hi = key_f32(sel[2]);
    __syncthreads();
    *ok = 1;
    return r_lo == r_hi ? v_lo : (v_lo + v_hi) / 2.0f;    // np.median: the middle value, or the float32 mean of the two
}

// ---- np.median in ONE pass where the caller can say [lo, hi] holds the middle value(s) (round 5) ----
// The pass counts the elements below lo and copies those of [lo, hi] into LDS (`band`, SEL_BAND values); when the
// middle rank(s) fall among the copied elements, an LDS radix select among them is the exact order statistic of
// the window -- whatever [lo, hi] was derived from; otherwise *ok = 0 (nothing else is touched) and the caller
// takes a slower route.  SIDE: the same pass also counts |x - side_centre| in SEL_BINS buckets of width
// 1 / side_inv_w (`side_hist`): from them the caller reads roughly where the middle DEVIATION of the window lies
// before the pass that selects it.  `nan_seen` as in window_select.  LEAN: no radix select among the copies when
// the bucket ranking does not apply (*ok = 0 instead): the streaming kernel of the MAD threshold stays small enough
// in registers for two workgroups per CU.
constexpr int SEL_BAND = 12288;
template <bool DEV, bool SIDE, class Fix = SelNoFix, bool LEAN = false>
__device__ float window_median_banded(const float* __restrict__ x, long long len, float centre, float lo, float hi,
                                      unsigned* hist, unsigned* sel, float* band, Fix fix, int* nan_seen, int* ok,
                                      float side_centre = 0.0f, float side_inv_w = 0.0f, unsigned* side_hist = nullptr)
{
    const int tid = threadIdx.x;
    constexpr int UNR = 8;
    if (tid == 0) { sel[0] = 0; sel[1] = 0; }
    if (SIDE)
        for (int b = tid; b < SEL_BINS; b += SEL_THREADS) side_hist[b] = 0;
    __syncthreads();
    unsigned below = 0;
    int bad = 0;
    auto one = [&](float v, long long i) {
        v = fix(v, i);
        if (SIDE) {
            const float q = __fmul_rn(fabsf(__fsub_rn(v, side_centre)), side_inv_w);
            atomicAdd(&side_hist[!(q > 0.0f) ? 0 : (q >= (float)(SEL_BINS - 1) ? SEL_BINS - 1 : (int)q)], 1u);
        }
        const float u = DEV ? fabsf(__fsub_rn(v, centre)) : v;
        bad |= u != u;
        below += u < lo;
        if (u >= lo && u <= hi) {
            const unsigned at = atomicAdd(&sel[0], 1u);
            if (at < (unsigned)SEL_BAND) band[at] = u;
        }
    };
    {
        long long i = tid;
        for (; i + (long long)(UNR - 1) * SEL_THREADS < len; i += (long long)UNR * SEL_THREADS) {
            float v[UNR];
#pragma unroll
            for (int e = 0; e < UNR; ++e) v[e] = x[i + (long long)e * SEL_THREADS];
#pragma unroll
            for (int e = 0; e < UNR; ++e) one(v[e], i + (long long)e * SEL_THREADS);
        }
        for (; i < len; i += SEL_THREADS) one(x[i], i);
    }
    for (int d = 32; d > 0; d >>= 1) below += __shfl_down(below, d, 64);
    if ((tid & 63) == 0 && below) atomicAdd(&sel[1], below);
    if (bad && nan_seen) *nan_seen = 1;
    __syncthreads();
    const unsigned n_band = sel[0], n_below = sel[1];
    const unsigned r_lo = (unsigned)((len - 1) / 2), r_hi = (unsigned)(len / 2);
    const bool usable = n_band <= (unsigned)SEL_BAND && r_lo >= n_below && r_hi < n_below + n_band;
    __syncthreads();
    if (!usable || (nan_seen && *nan_seen)) { *ok = usable ? 1 : 0; return __uint_as_float(0x7fc00000u); }
    // the order statistics among the copied elements: equal-width buckets over [lo, hi] (a handful of elements
    // per bucket) and a ranking of the bucket of the rank, all in LDS (the keys at the end of `band`); the radix
    // select, also in LDS, when that does not apply or work
    if (n_band <= (unsigned)(SEL_BAND - SEL_BUF) && len > SEL_BAND && hi - lo > 0.0f && hi - lo < 1.0e37f) {
        int fine = 0;
        const float v = window_median_bucketed<false, SelNoFix>(band, (long long)n_band, 0.0f, lo, (float)SEL_BINS / (hi - lo), hist, sel,
                                                                (unsigned*)(band + (SEL_BAND - SEL_BUF)), SelNoFix(), nullptr, &fine,
                                                                (long long)(r_lo - n_below), (long long)(r_hi - n_below));
        if (fine) { *ok = 1; return v; }
    }
    if (LEAN && r_hi != r_lo) {
        // (a caller with a second kernel for what this one leaves, and no registers to spare: two middle ranks
        // that do not share a bucket take two single-rank selects of the slim kind instead of the paired one)
        const unsigned k_lo = window_select<false, false>(band, (long long)n_band, 0.0f, r_lo - n_below, hist, sel);
        const unsigned k_hi = window_select<false, false>(band, (long long)n_band, 0.0f, r_hi - n_below, hist, sel);
        *ok = 1;
        return (key_f32(k_lo) + key_f32(k_hi)) / 2.0f;
    }
    unsigned key_hi = 0;
    int both = 0;
    const unsigned key_lo = window_select<false, false>(band, (long long)n_band, 0.0f, r_lo - n_below, hist, sel,
                                                        r_hi != r_lo ? &key_hi : nullptr, r_hi != r_lo ? &both : nullptr);
    if (r_hi != r_lo && !both) key_hi = window_select<false, false>(band, (long long)n_band, 0.0f, r_hi - n_below, hist, sel);
    *ok = 1;
    return r_hi == r_lo ? key_f32(key_lo) : (key_f32(key_lo) + key_f32(key_hi)) / 2.0f;
}

// the buckets of a side histogram (window_median_banded, SIDE) that hold ranks r_lo and r_hi: sel2[0], sel2[1]
// (SEL_BINS when the histogram does not reach the rank); call with all threads, one barrier inside
__device__ __forceinline__ void side_rank_buckets(const unsigned* side_hist, unsigned r_lo, unsigned r_hi, unsigned* sel2)
{
    const int tid = threadIdx.x;
    if (tid < 64) {
        constexpr int PER = SEL_BINS / 64;
        if (tid == 0) { sel2[0] = SEL_BINS; sel2[1] = SEL_BINS; }
        unsigned tot = 0;
        for (int b = 0; b < PER; ++b) tot += side_hist[tid * PER + b];
        unsigned inc = tot;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(inc, d, 64);
            if (tid >= d) inc += o;
        }
        const unsigned exc = inc - tot;
        for (int which = 0; which < 2; ++which) {
            const unsigned r = which ? r_hi : r_lo;
            if (r >= exc && r < inc) {
                unsigned acc = exc;
                int b = 0;
                for (; b < PER - 1; ++b) {
                    const unsigned h = side_hist[tid * PER + b];
                    if (r < acc + h) break;
                    acc += h;
                }
                sel2[which] = (unsigned)(tid * PER + b);
            }
        }
    }
    __syncthreads();
}

}  // namespace bpmf
