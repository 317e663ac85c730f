// Detection stage right behind the beamformer, on the device.
//
// Reference: Beamformer.find_detections (BPMF/template_search.py:574-627), its threshold
// template_search.time_dependent_threshold (:1418-1487: sliding median + n_dev * MAD, linear
// interpolation between window centres) and utils._detect_peaks (BPMF/utils.py:2203-2354).
// In the reference all of it is NumPy on the full-length max-beam; here the (N,) max-beam stays in
// HBM and only small records leave the GPU:
//
//   bpmf_bp_window_stats_dev   median and MAD of every sliding window, float32, exactly np.median:
//                              one workgroup per window, 3-pass radix select (11 + 11 + 10 bits) on
//                              order-preserving keys through an LDS histogram; windows are read from
//                              L2 (a window of 90 000 samples is 360 KB), so the kernel is bound by
//                              LDS atomics, not HBM: 4 B per sample per pass.
//   bpmf_bp_extract_peaks_dev  rising-edge local maxima (x[t] > x[t-1] and x[t+1] <= x[t], the
//                              `edge="rising"` rule of _detect_peaks :2292-2301) above a scalar floor,
//                              compacted as (sample, beam, source) records.
//
// The floor is the smallest node of the interpolated threshold: a lower bound of threshold(t) for
// every t, so the compacted list is a superset of the peaks above the threshold, and every peak it
// leaves out is lower than all of them -- the reference's tallest-first min-distance suppression
// (utils.py:2334-2345) run on the list gives the same survivors above the threshold as the run on
// all local maxima.  Suppression, the exact float64 threshold test at the few survivors, the
// +-mpd/2 snap and np.unique run on that list on the host (seismic_bpmf_amd/workflow.py).
#include "select.h"
#include "../../include/bpmf_hip.h"

namespace bpmf {

// window q (1-based, as the reference's loop): samples [q shift, min(n, q shift + window))
__global__ __launch_bounds__(SEL_THREADS) void bp_window_stats_kernel(
    const float* __restrict__ beam, long long n, long long window, long long shift,
    float* __restrict__ med, float* __restrict__ mad)
{
    __shared__ unsigned hist[SEL_BINS];
    __shared__ unsigned sel[4];
    __shared__ int has_nan;
    const long long q = (long long)blockIdx.x + 1;
    const long long i1 = q * shift;
    const long long i2 = i1 + window < n ? i1 + window : n;
    const int len = (int)(i2 - i1);
    if (len <= 0) {                       // empty window: np.median([]) = NaN
        if (threadIdx.x == 0) { med[q] = __uint_as_float(0x7fc00000u); mad[q] = __uint_as_float(0x7fc00000u); }
        return;
    }
    const float* x = beam + i1;
    if (threadIdx.x == 0) has_nan = 0;
    __syncthreads();
    int bad = 0;
    for (int i = threadIdx.x; i < len; i += SEL_THREADS) bad |= x[i] != x[i];
    if (bad) has_nan = 1;
    __syncthreads();
    if (has_nan) {
        if (threadIdx.x == 0) { med[q] = __uint_as_float(0x7fc00000u); mad[q] = __uint_as_float(0x7fc00000u); }
        return;
    }
    const float m = window_median<false>(x, len, len, 0.0f, hist, sel);
    const float d = window_median<true>(x, len, len, m, hist, sel);
    if (threadIdx.x == 0) { med[q] = m; mad[q] = d; }
}

struct BpPeakRecord { int index; float beam; int source; int pad; };

__global__ __launch_bounds__(256) void bp_extract_peaks_kernel(
    const float* __restrict__ beam, const int* __restrict__ arg, long long n, double floor_,
    unsigned capacity, unsigned* __restrict__ count, BpPeakRecord* __restrict__ rec)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    float x1 = 0.0f;
    if (t >= 1 && t < n - 1) {
        const float x0 = beam[t - 1], x2 = beam[t + 1];
        x1 = beam[t];
        // NaN neighbours: the reference turns NaNs into +inf and then drops every peak next to one;
        // comparisons with NaN are false, which excludes the same samples here
        hit = x1 > x0 && x2 <= x1 && (double)x1 > floor_;
    }
    const unsigned long long mask = __ballot(hit);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(count, (unsigned)__popcll(mask));
    base = __shfl(base, 0, 64);
    if (hit) {
        const unsigned slot = base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
        if (slot < capacity) rec[slot] = BpPeakRecord{(int)t, x1, arg ? arg[t] : 0, 0};
    }
}

}  // namespace bpmf

using namespace bpmf;

extern "C" size_t bpmf_bp_num_windows(size_t n, size_t window, size_t shift)
{
    if (shift == 0 || window == 0 || n < window) return 0;
    return (n - window) / shift + 1;      // int((n - window) // shift) + 1, template_search.py:1452
}

extern "C" int bpmf_bp_window_stats_dev(const float* d_beam, size_t n, size_t window, size_t shift,
                                        bpmf_stream_t stream_, float* d_median, float* d_mad)
{
    hipStream_t stream = (hipStream_t)stream_;
    const size_t nw = bpmf_bp_num_windows(n, window, shift);
    if (!d_beam || !d_median || !d_mad || nw == 0) {
        set_error("bpmf_bp_window_stats_dev: bad argument (n=%zu window=%zu shift=%zu)", n, window, shift);
        return -1;
    }
    if (window > 0x7fffffffull || nw > 0x7fffffffull) {
        set_error("bpmf_bp_window_stats_dev: window too long");
        return -1;
    }
    // windows q = 1 .. nw; the last ones may be cut short by the end of the series, and the last one
    // is EMPTY when nw * shift == n (overlap 0 and a series of a whole number of windows): the
    // reference takes np.median of an empty slice there, NaN (template_search.py:1459-1466) -- so
    // does the kernel (len <= 0)
    bp_window_stats_kernel<<<dim3((unsigned)nw), dim3(SEL_THREADS), 0, stream>>>(
        d_beam, (long long)n, (long long)window, (long long)shift, d_median, d_mad);
    BPMF_LAUNCH_CHECK();
    return 0;
}

extern "C" int bpmf_bp_extract_peaks_dev(const float* d_beam, const int32_t* d_sources, size_t n,
                                         double floor_value, uint32_t capacity,
                                         bpmf_stream_t stream_, uint32_t* d_count,
                                         bpmf_bp_peak* d_records)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_beam || !d_count || (!d_records && capacity)) {
        set_error("bpmf_bp_extract_peaks_dev: null pointer");
        return -1;
    }
    static_assert(sizeof(BpPeakRecord) == sizeof(bpmf_bp_peak), "record layout");
    BPMF_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(uint32_t), stream));
    if (n < 3) return 0;
    bp_extract_peaks_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(
        d_beam, d_sources, (long long)n, floor_value, capacity, d_count, (BpPeakRecord*)d_records);
    BPMF_LAUNCH_CHECK();
    return 0;
}
