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

}  // namespace bpmf
