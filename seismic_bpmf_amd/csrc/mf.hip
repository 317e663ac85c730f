// Matched-filter hot path for MI355X (gfx950): sliding normalised cross-correlation of T
// templates against S*C continuous channels, weighted network sum.
//
// Serves fast_matched_filter.matched_filter as called by the reference at
// BPMF/similarity_search.py:526-533 and BPMF/dataset.py:4818-4827 (the arithmetic itself
// is not in the reference tree).  Conventions = oracle/bpmf_oracle.c:mf_cpu, bit for bit.
//
// Kernel design (see DESIGN.md section "MF"):
//   For one (template, channel) the numerators of 1024 consecutive lags are ONE 32x32
//   MFMA tile:  Out[b][a] = sum_m A[b][m] * D[m][a],  lag = 32a + b,
//     A[b][m] = tmpl[m - b]          (32 x (L+31) Toeplitz band of the template)
//     D[m][a] = data[x0 + 32a + m]   (strided view of the contiguous data window)
//   evaluated with v_mfma_f32_32x32x2_f32, which is an exact, k-ordered fp32 fmaf chain;
//   the band's zeros add exact zeros, so every numerator equals the scalar chain
//   fmaf(tmpl[l], data[i+mv+l], acc) for l = 0..L-1.  The per-channel moveout only moves
//   x0, so no alignment between templates or channels is needed.  A workgroup of 4 waves
//   owns (one template) x (4096 consecutive lags) and walks the S*C channels with the
//   weighted CC sum in registers; data window and Toeplitz band live in LDS.
#include "common.h"
#include "../../include/bpmf_hip.h"

namespace bpmf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------ preparation ---

// E_t[t,s,c]: fmaf chain over l ascending.
__global__ void mf_template_energy_kernel(const float* __restrict__ tmpl, size_t n_rows, int L,
                                          float* __restrict__ e_t)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const float* x = tmpl + i * (size_t)L;
    float acc = 0.0f;
    for (int l = 0; l < L; ++l) acc = __fmaf_rn(x[l], x[l], acc);
    e_t[i] = acc;
}

// Valid lag range [first, last] of each template (first > last = empty).
__global__ void mf_range_kernel(const int* __restrict__ mv, const float* __restrict__ w, int T,
                                int n_ch, long long step, long long L, long long N,
                                long long n_corr, int2* __restrict__ range)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    long long mv_min = 0, mv_max = 0;
    bool any = false;
    for (int ch = 0; ch < n_ch; ++ch) {
        if (w[(size_t)t * n_ch + ch] == 0.0f) continue;
        long long m = mv[(size_t)t * n_ch + ch];
        if (!any || m < mv_min) mv_min = m;
        if (!any || m > mv_max) mv_max = m;
        any = true;
    }
    int2 r = make_int2(1, 0);
    if (any && N >= L) {
        long long first = mv_min < 0 ? (-mv_min + step - 1) / step : 0;
        long long room = N - L - mv_max;
        if (room >= 0) {
            long long last = room / step;
            if (last > n_corr - 1) last = n_corr - 1;
            if (first <= last) r = make_int2((int)first, (int)last);
        }
    }
    range[t] = r;
}

// Chunk-local prefix sums of data^2 in double (one thread per 1024-sample chunk).
// local[ch, n] = sum of squares of samples [chunk_start(n), n]; tot[ch, q] = chunk totals.
__global__ void mf_csum_local_kernel(const float* __restrict__ data, size_t n_ch, size_t N,
                                     size_t nq, double* __restrict__ local,
                                     double* __restrict__ tot)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_ch * nq) return;
    size_t ch = idx / nq, q = idx % nq;
    size_t n0 = q * CSUM_CHUNK;
    size_t n1 = n0 + CSUM_CHUNK < N ? n0 + CSUM_CHUNK : N;
    const float* d = data + ch * N;
    double* lo = local + ch * N;
    double acc = 0.0;
    for (size_t n = n0; n < n1; ++n) {
        double v = (double)d[n];
        acc = acc + v * v;  // v*v is exact in double
        lo[n] = acc;
    }
    tot[idx] = acc;
}

// off[ch, q] = sequential sum of tot[ch, 0..q-1]  (one thread per channel).
__global__ void mf_csum_offsets_kernel(const double* __restrict__ tot, size_t n_ch, size_t nq,
                                       double* __restrict__ off)
{
    size_t ch = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= n_ch) return;
    double acc = 0.0;
    for (size_t q = 0; q < nq; ++q) {
        off[ch * nq + q] = acc;
        acc = acc + tot[ch * nq + q];
    }
}

// E_d[ch, j] = (float)(csum[j+L] - csum[j]),  csum[n] = off[chunk(n-1)] + local[n-1].
__global__ void mf_window_energy_kernel(const double* __restrict__ local,
                                        const double* __restrict__ off, size_t n_ch, size_t N,
                                        size_t nq, size_t L, size_t nwin,
                                        float* __restrict__ e_d)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t ch = blockIdx.y;
    if (j >= nwin) return;
    const double* lo = local + ch * N;
    const double* of = off + ch * nq;
    size_t nh = j + L - 1;
    double hi = of[nh / CSUM_CHUNK] + lo[nh];
    double low = 0.0;
    if (j > 0) low = of[(j - 1) / CSUM_CHUNK] + lo[j - 1];
    e_d[ch * nwin + j] = (float)(hi - low);
}

// --------------------------------------------------------------- MFMA main kernel ---

constexpr int MF_THREADS = 256;
constexpr int MF_LAGS_PER_WAVE = 1024;
constexpr int MF_LAGS_PER_WG = 4096;

__device__ __forceinline__ int mf_pad(int x) { return x + (x >> 5); }

__host__ __device__ inline int mf_kpad(int L) { return (L + 31 + 31) / 32 * 32; }
// LDS floats: Toeplitz band (Kpad + 32) + padded data window.
__host__ __device__ inline int mf_window_len(int L) { return MF_LAGS_PER_WG - 32 + mf_kpad(L); }
__host__ inline size_t mf_lds_bytes(int L)
{
    int W = mf_window_len(L);
    return (size_t)(mf_kpad(L) + 32 + (W + (W >> 5) + 1)) * sizeof(float);
}

template <bool NETWORK_SUM>
__global__ __launch_bounds__(MF_THREADS, 2) void mf_mfma_kernel(
    const float* __restrict__ tmpl, const int* __restrict__ mv, const float* __restrict__ wgt,
    const float* __restrict__ data, const float* __restrict__ e_t,
    const float* __restrict__ e_d, const int2* __restrict__ range, int L, long long N, int T,
    int n_ch, long long n_corr, float* __restrict__ out)
{
    extern __shared__ float smem[];
    const int Kpad = mf_kpad(L);
    const int tp_len = Kpad + 32;
    const int W = mf_window_len(L);
    float* tp = smem;          // tp[31 + l] = tmpl[l], zeros around
    float* dw = smem + tp_len; // dw[pad(x)] = data[g0 + x]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int a = lane & 31;
    const int hi = lane >> 5;

    const int t = blockIdx.x % T;
    const long long lag0 = (long long)(blockIdx.x / T) * MF_LAGS_PER_WG;
    const int2 rg = range[t];
    const long long nwin = N - L + 1;

    float sum[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sum[r] = 0.0f;

    const bool wg_valid = !(lag0 > rg.y || lag0 + MF_LAGS_PER_WG - 1 < rg.x);
    const long long lag_w = lag0 + (long long)wv * MF_LAGS_PER_WAVE + 32 * a + 4 * hi;

    if (wg_valid) {
        const int a_base = 31 - a + hi;
        const int b_base = 1056 * wv + 33 * a + hi;
        for (int ch = 0; ch < n_ch; ++ch) {
            const float w = wgt[(size_t)t * n_ch + ch];
            if (w == 0.0f) continue;
            const int mvc = mv[(size_t)t * n_ch + ch];
            const float et = e_t[(size_t)t * n_ch + ch];
            __syncthreads();  // everyone is done reading the previous channel's LDS
            const float* tsrc = tmpl + ((size_t)t * n_ch + ch) * (size_t)L;
            for (int x = tid; x < tp_len; x += MF_THREADS) {
                int l = x - 31;
                tp[x] = (l >= 0 && l < L) ? tsrc[l] : 0.0f;
            }
            const float* dsrc = data + (size_t)ch * (size_t)N;
            const long long g0 = lag0 + mvc;
            for (int x = tid; x < W; x += MF_THREADS) {
                long long g = g0 + x;
                dw[mf_pad(x)] = (g >= 0 && g < N) ? dsrc[g] : 0.0f;
            }
            __syncthreads();

            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const int nq = Kpad >> 5;
            for (int q = 0; q < nq; ++q) {
                const float* ap = tp + a_base + 32 * q;
                const float* bp = dw + b_base + 33 * q;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * j], bp[2 * j], acc, 0, 0, 0);
            }

            const float* edc = e_d + (size_t)ch * (size_t)nwin;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long lag = lag_w + (r & 3) + 8 * (r >> 2);
                const bool ok = lag >= rg.x && lag <= rg.y;
                float cc = 0.0f;
                if (ok) {
                    const float den = et * edc[lag + mvc];
                    if (den > STABILITY_THRESHOLD) cc = acc[r] / sqrtf(den);
                    if (!NETWORK_SUM) out[((size_t)t * n_corr + lag) * n_ch + ch] = cc;
                }
                if (NETWORK_SUM) sum[r] = __fmaf_rn(w, cc, sum[r]);
            }
        }
    }
    if (NETWORK_SUM) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long lag = lag_w + (r & 3) + 8 * (r >> 2);
            if (lag < n_corr) out[(size_t)t * n_corr + lag] = sum[r];
        }
    }
}

// ------------------------------------------------------ generic (any step) kernel ---
// One thread per (template, lag); plain fmaf chain.  Used when step != 1 or when the
// template is too long for the LDS tile, and as an independent on-device cross-check.
template <bool NETWORK_SUM>
__global__ __launch_bounds__(256) void mf_direct_kernel(
    const float* __restrict__ tmpl, const int* __restrict__ mv, const float* __restrict__ wgt,
    const float* __restrict__ data, const float* __restrict__ e_t,
    const float* __restrict__ e_d, const int2* __restrict__ range, long long step, int L,
    long long N, int n_ch, long long n_corr, float* __restrict__ out)
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
            const float den = e_t[(size_t)t * n_ch + ch] * e_d[(size_t)ch * nwin + j];
            float cc = 0.0f;
            if (den > STABILITY_THRESHOLD) cc = num / sqrtf(den);
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
    ws.e_d = (float*)(p + o);    o += align_up(n_ch * nwin * sizeof(float), 256);
    ws.e_t = (float*)(p + o);    o += align_up(T * n_ch * sizeof(float), 256);
    ws.range = (int2*)(p + o);   o += align_up(T * sizeof(int2), 256);
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

}  // namespace bpmf

using namespace bpmf;

extern "C" size_t bpmf_mf_workspace_bytes(size_t L, size_t N, size_t T, size_t S, size_t C)
{
    return mf_carve(nullptr, L, N, T, S * C).bytes;
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
    const size_t nq = (N + CSUM_CHUNK - 1) / CSUM_CHUNK;
    const size_t nwin = N - L + 1;
    {
        size_t n = n_ch * nq;
        mf_csum_local_kernel<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream>>>(
            d_data, n_ch, N, nq, ws.local, ws.tot);
        BPMF_LAUNCH_CHECK();
    }
    mf_csum_offsets_kernel<<<dim3((unsigned)((n_ch + 63) / 64)), dim3(64), 0, stream>>>(
        ws.tot, n_ch, nq, ws.off);
    BPMF_LAUNCH_CHECK();
    mf_window_energy_kernel<<<dim3((unsigned)((nwin + 255) / 256), (unsigned)n_ch), dim3(256), 0,
                              stream>>>(ws.local, ws.off, n_ch, N, nq, L, nwin, ws.e_d);
    BPMF_LAUNCH_CHECK();
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
    {
        size_t n = T * n_ch;
        mf_template_energy_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(
            d_templates, n, (int)L, ws.e_t);
        BPMF_LAUNCH_CHECK();
        mf_range_kernel<<<dim3((unsigned)((T + 63) / 64)), dim3(64), 0, stream>>>(
            d_moveouts, d_weights, (int)T, (int)n_ch, (long long)step, (long long)L, (long long)N,
            (long long)n_corr, ws.range);
        BPMF_LAUNCH_CHECK();
    }
    if (!network_sum)
        BPMF_HIP_CHECK(hipMemsetAsync(d_cc_out, 0, T * n_corr * n_ch * sizeof(float), stream));

    profile_mark(BPMF_KERNEL_MF_MAIN, 0, stream);
    const size_t lds = mf_lds_bytes((int)L);
    const size_t n_lag_blocks = (n_corr + MF_LAGS_PER_WG - 1) / MF_LAGS_PER_WG;
    const bool use_mfma = step == 1 && !(flags & BPMF_MF_FORCE_DIRECT) && lds <= 64 * 1024 &&
                          T * n_lag_blocks < 0x7fffffffull;
    if (use_mfma) {
        dim3 grid((unsigned)(T * n_lag_blocks));
        if (network_sum)
            mf_mfma_kernel<true><<<grid, dim3(MF_THREADS), lds, stream>>>(
                d_templates, d_moveouts, d_weights, d_data, ws.e_t, ws.e_d, ws.range, (int)L,
                (long long)N, (int)T, (int)n_ch, (long long)n_corr, d_cc_out);
        else
            mf_mfma_kernel<false><<<grid, dim3(MF_THREADS), lds, stream>>>(
                d_templates, d_moveouts, d_weights, d_data, ws.e_t, ws.e_d, ws.range, (int)L,
                (long long)N, (int)T, (int)n_ch, (long long)n_corr, d_cc_out);
    } else {
        dim3 grid((unsigned)((n_corr + 255) / 256), (unsigned)T);
        if (T > 65535) {
            set_error("bpmf_mf_run_dev: generic kernel supports at most 65535 templates per call");
            return -1;
        }
        if (network_sum)
            mf_direct_kernel<true><<<grid, dim3(256), 0, stream>>>(
                d_templates, d_moveouts, d_weights, d_data, ws.e_t, ws.e_d, ws.range,
                (long long)step, (int)L, (long long)N, (int)n_ch, (long long)n_corr, d_cc_out);
        else
            mf_direct_kernel<false><<<grid, dim3(256), 0, stream>>>(
                d_templates, d_moveouts, d_weights, d_data, ws.e_t, ws.e_d, ws.range,
                (long long)step, (int)L, (long long)N, (int)n_ch, (long long)n_corr, d_cc_out);
    }
    BPMF_LAUNCH_CHECK();
    profile_mark(BPMF_KERNEL_MF_MAIN, 1, stream);
    return 0;
}

extern "C" int bpmf_mf_run(const float* templates, const int32_t* moveouts, const float* weights,
                           const float* data, size_t step, size_t L, size_t N, size_t T, size_t S,
                           size_t C, size_t n_corr, int network_sum, int flags, int device,
                           float* cc_out)
{
    if (!templates || !moveouts || !weights || !data || !cc_out) {
        set_error("bpmf_mf_run: null pointer");
        return -1;
    }
    if (int rc = mf_check_sizes(step, L, N, T, S, C, n_corr)) return rc;
    BPMF_HIP_CHECK(hipSetDevice(device));
    const size_t n_ch = S * C;
    const size_t b_tp = T * n_ch * L * sizeof(float), b_mv = T * n_ch * sizeof(int32_t),
                 b_w = T * n_ch * sizeof(float), b_d = n_ch * N * sizeof(float),
                 b_out = T * n_corr * (network_sum ? 1 : n_ch) * sizeof(float),
                 b_ws = bpmf_mf_workspace_bytes(L, N, T, S, C);
    char* base = nullptr;
    size_t o_tp = 0, o_mv = o_tp + align_up(b_tp, 256), o_w = o_mv + align_up(b_mv, 256),
           o_d = o_w + align_up(b_w, 256), o_out = o_d + align_up(b_d, 256),
           o_ws = o_out + align_up(b_out, 256), total = o_ws + b_ws;
    BPMF_HIP_CHECK(hipMalloc((void**)&base, total));
    int rc = 0;
    hipStream_t stream = nullptr;
    auto fail = [&](hipError_t e, const char* what) {
        set_error("bpmf_mf_run: %s failed: %s", what, hipGetErrorString(e));
        rc = -2;
    };
    hipError_t e;
    if ((e = hipMemcpyAsync(base + o_tp, templates, b_tp, hipMemcpyHostToDevice, stream)) != hipSuccess) fail(e, "H2D templates");
    if (!rc && (e = hipMemcpyAsync(base + o_mv, moveouts, b_mv, hipMemcpyHostToDevice, stream)) != hipSuccess) fail(e, "H2D moveouts");
    if (!rc && (e = hipMemcpyAsync(base + o_w, weights, b_w, hipMemcpyHostToDevice, stream)) != hipSuccess) fail(e, "H2D weights");
    if (!rc && (e = hipMemcpyAsync(base + o_d, data, b_d, hipMemcpyHostToDevice, stream)) != hipSuccess) fail(e, "H2D data");
    if (!rc)
        rc = bpmf_mf_run_dev((const float*)(base + o_tp), (const int32_t*)(base + o_mv),
                             (const float*)(base + o_w), (const float*)(base + o_d), step, L, N, T,
                             S, C, n_corr, network_sum, flags & ~BPMF_MF_DATA_PREPARED,
                             base + o_ws, b_ws, stream, (float*)(base + o_out));
    if (!rc && (e = hipMemcpyAsync(cc_out, base + o_out, b_out, hipMemcpyDeviceToHost, stream)) != hipSuccess) fail(e, "D2H cc");
    if (!rc && (e = hipStreamSynchronize(stream)) != hipSuccess) fail(e, "synchronize");
    (void)hipFree(base);
    return rc;
}
