// Batched inter-template CC: the whole (T, T) similarity matrix of
// TemplateGroup.compute_intertemplate_cc (BPMF/dataset.py:4775-4841) in one pass.
//
// The reference loops over the templates: for template t, its own waveform (S, C, Lw) is the "data",
// every template u trimmed by max_lag on both sides (L = Lw - 2 max_lag samples, zero moveouts) is
// correlated against it at the 2 max_lag + 1 lags with fmf.matched_filter(..., network_sum=False)
// (:4818-4827), and intertp[t, u] = sum_{s,c} w_t[u, s, c] * max_lag cc[u, lag, s, c] (:4828-4830),
// where w_t[u] is template t's normalised station/channel weights, or all zeros for the templates
// farther than the distance threshold (:4789-4816).  That is T launches of tiny problems -- latency
// bound, 1.2 ms each on the MI355X.  Here:
//
//   intertp_norms_kernel   per (template, channel): 1/sqrt of the trimmed template's energy and of
//                          the 2 max_lag + 1 window energies of the full waveform -- the matched
//                          filter's own definitions (double prefix sums of the squares in chunks of
//                          1024 samples, oracle/bpmf_oracle.c:mf_data_csum).
//   intertp_cc_kernel      one workgroup per (t, tile of 8 templates u): every (u, channel, lag)
//                          CC as an fmaf chain over the samples (8 lags per thread, register-tiled), the data rows of t staged in LDS,
//                          max over the lags, weight, and the channel sum in NumPy's pairwise
//                          order, so that the result equals the per-template path bit for bit.
//
// Weights come factorised, as the reference builds them: base (T, S, C) = row t's channel weights,
// pair_mask (T, T) = 1 where the pair is within the distance threshold.
#include "common.h"
#include "../../include/bpmf_hip.h"

namespace bpmf {

constexpr int ITP_UB = 8;          // templates u per workgroup
constexpr int ITP_THREADS = 256;
constexpr int ITP_NLG = 8;         // lags per thread

// r_t[u, ch] = 1/sqrtf(sum_l trimmed^2) (float fmaf chain), r_d[t, ch, lag] = 1/sqrtf((float)(cs[lag+L]-cs[lag]))
__global__ void intertp_norms_kernel(const float* __restrict__ wf, int n_rows, int Lw, int max_lag,
                                     float* __restrict__ r_t, float* __restrict__ r_d)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;   // (template, channel)
    if (row >= n_rows) return;
    const float* x = wf + (size_t)row * Lw;
    const int L = Lw - 2 * max_lag, n_lag = 2 * max_lag + 1;
    float e = 0.0f;
    for (int l = 0; l < L; ++l) e = __fmaf_rn(x[max_lag + l], x[max_lag + l], e);
    r_t[row] = 1.0f / sqrtf(e);   // correctly rounded sqrt and divide (hipcc default), as mf.hip
    // prefix sums of the squares, hierarchical exactly like the matched filter's: restart inside
    // every chunk of CSUM_CHUNK samples, chunk totals accumulated sequentially
    double off = 0.0, local = 0.0;   // csum(n) = off + local
    double lo[64];                   // csum(lag), lag = 0 .. n_lag - 1  (n_lag <= 64)
    // one pass over the samples; csum(lag) is stored when n reaches lag, window `lag` is closed when n
    // reaches lag + L (L >= 1, so its csum(lag) is already there -- also when L < 2 max_lag, where the
    // starts of the late windows come AFTER the ends of the early ones)
    int lag_lo = 0, lag_hi = 0;
    for (int n = 0; ; ++n) {
        const double cs = off + local;
        if (lag_lo < n_lag && lag_lo == n) lo[lag_lo++] = cs;
        if (lag_hi < n_lag && lag_hi + L == n) {
            r_d[(size_t)row * n_lag + lag_hi] = 1.0f / sqrtf((float)(cs - lo[lag_hi]));
            ++lag_hi;
        }
        if (n == Lw) break;
        if (n % CSUM_CHUNK == 0 && n) { off += local; local = 0.0; }
        local += (double)x[n] * (double)x[n];
    }
}

// np.sum of n float32 in NumPy's pairwise order (numpy/core/src/umath/loops_utils.h.src)
__device__ float numpy_pairwise_sum(const float* x, int n)
{
    if (n < 8) {
        float r = 0.0f;
        for (int i = 0; i < n; ++i) r = __fadd_rn(r, x[i]);
        return r;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = x[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], x[i + j]);
        float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                              __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
        for (; i < n; ++i) res = __fadd_rn(res, x[i]);
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return __fadd_rn(numpy_pairwise_sum(x, n2), numpy_pairwise_sum(x + n2, n - n2));
}

// LDS: rows[ch_chunk][Lw] data of t | cc[ITP_UB][ch_chunk][n_lag] | prod[ITP_UB][n_ch]
__global__ __launch_bounds__(ITP_THREADS) void intertp_cc_kernel(
    const float* __restrict__ wf, const float* __restrict__ base_w, const unsigned char* __restrict__ mask,
    const float* __restrict__ r_t, const float* __restrict__ r_d, int T, int n_ch, int Lw, int max_lag,
    int ch_chunk, float* __restrict__ out)
{
    extern __shared__ float lds[];
    const int t = blockIdx.y, u0 = blockIdx.x * ITP_UB;
    const int L = Lw - 2 * max_lag, n_lag = 2 * max_lag + 1;
    float* rows = lds;
    float* ccb = rows + (size_t)ch_chunk * Lw;
    float* prod = ccb + (size_t)ITP_UB * ch_chunk * n_lag;
    const int tid = threadIdx.x;
    const float* wt = base_w + (size_t)t * n_ch;
    // which templates of the tile take part: inside the mask and row t has a non-zero weight at all
    bool any_w = false;
    for (int ch = 0; ch < n_ch; ++ch) any_w |= wt[ch] != 0.0f;
    bool tile_live = false;
    for (int j = 0; j < ITP_UB; ++j) {
        const int u = u0 + j;
        tile_live |= u < T && any_w && mask[(size_t)t * T + u];
    }
    if (!tile_live) {
        if (tid < ITP_UB && u0 + tid < T) out[(size_t)t * T + u0 + tid] = 0.0f;
        return;
    }
    for (int c0 = 0; c0 < n_ch; c0 += ch_chunk) {
        const int nc = min(ch_chunk, n_ch - c0);
        __syncthreads();
        for (int i = tid; i < nc * Lw; i += ITP_THREADS) rows[i] = wf[((size_t)t * n_ch + c0) * Lw + i];
        __syncthreads();
        // one thread per (u, channel, group of ITP_NLG consecutive lags): ITP_NLG numerators, each an
        // fmaf chain over the samples in ascending order from 0.  A block of 8 samples needs the 15
        // data values d[l0 + lag0 .. l0 + lag0 + 14]: 7 carried over from the previous block, 8 new
        // ones -- one template load and one LDS read per 8 fmas.
        const int n_grp = (n_lag + ITP_NLG - 1) / ITP_NLG;
        const int n_item = ITP_UB * nc * n_grp;
        for (int it = tid; it < n_item; it += ITP_THREADS) {
            const int grp = it % n_grp, ch = (it / n_grp) % nc, j = it / (n_grp * nc);
            const int u = u0 + j, lag0 = grp * ITP_NLG;
            float num[ITP_NLG];
#pragma unroll
            for (int k = 0; k < ITP_NLG; ++k) num[k] = 0.0f;
            const bool live = u < T && mask[(size_t)t * T + u] && wt[c0 + ch] != 0.0f;
            if (live) {
                const float* tp = wf + ((size_t)u * n_ch + c0 + ch) * Lw + max_lag;
                const float* d = rows + (size_t)ch * Lw + lag0;      // d[l + k] = data sample of lag lag0 + k
                const int dmax = Lw - lag0;                          // d[x] exists for x < dmax
                float win[ITP_NLG + 7];
#pragma unroll
                for (int k = 0; k < 7; ++k) win[k] = k < dmax ? d[k] : 0.0f;
                int l0 = 0;
                for (; l0 + 8 <= L; l0 += 8) {
                    float tv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) tv[i] = tp[l0 + i];
#pragma unroll
                    for (int k = 0; k < 8; ++k) win[7 + k] = l0 + 7 + k < dmax ? d[l0 + 7 + k] : 0.0f;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int k = 0; k < ITP_NLG; ++k) num[k] = __fmaf_rn(tv[i], win[i + k], num[k]);
#pragma unroll
                    for (int k = 0; k < 7; ++k) win[k] = win[8 + k];
                }
                for (; l0 < L; ++l0) {                               // the last L % 8 samples
                    const float tv = tp[l0];
#pragma unroll
                    for (int k = 0; k < ITP_NLG; ++k)
                        if (lag0 + k < n_lag) num[k] = __fmaf_rn(tv, d[l0 + k], num[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < ITP_NLG; ++k) {
                const int lag = lag0 + k;
                if (lag >= n_lag) continue;
                float cc = 0.0f;
                if (live) {
                    const float nrm = __fmul_rn(r_t[(size_t)u * n_ch + c0 + ch],
                                                r_d[((size_t)t * n_ch + c0 + ch) * n_lag + lag]);
                    if (nrm < MAX_NORM) cc = __fmul_rn(num[k], nrm);
                }
                ccb[((size_t)j * nc + ch) * n_lag + lag] = cc;
            }
        }
        __syncthreads();
        // np.max over the lags, times the weight
        for (int it = tid; it < ITP_UB * nc; it += ITP_THREADS) {
            const int ch = it % nc, j = it / nc;
            const float* c = ccb + ((size_t)j * nc + ch) * n_lag;
            float b = c[0];
            for (int lag = 1; lag < n_lag; ++lag) b = c[lag] > b ? c[lag] : b;
            prod[j * n_ch + c0 + ch] = __fmul_rn(wt[c0 + ch], b);
        }
    }
    __syncthreads();
    if (tid < ITP_UB && u0 + tid < T) {
        const int u = u0 + tid;
        out[(size_t)t * T + u] = mask[(size_t)t * T + u] ? numpy_pairwise_sum(prod + tid * n_ch, n_ch) : 0.0f;
    }
}

}  // namespace bpmf

using namespace bpmf;

extern "C" size_t bpmf_intertemplate_workspace_bytes(size_t T, size_t S, size_t C, size_t max_lag)
{
    const size_t n_rows = T * S * C, n_lag = 2 * max_lag + 1;
    return align_up(n_rows * sizeof(float), 256) + align_up(n_rows * n_lag * sizeof(float), 256);
}

extern "C" int bpmf_intertemplate_cc_dev(const float* d_waveforms, const float* d_base_weights,
                                         const uint8_t* d_pair_mask, size_t T, size_t S, size_t C,
                                         size_t Lw, size_t max_lag, void* d_workspace,
                                         size_t workspace_bytes, bpmf_stream_t stream_, float* d_out)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_waveforms || !d_base_weights || !d_pair_mask || !d_workspace || !d_out || T == 0 || S == 0 ||
        C == 0) {
        set_error("bpmf_intertemplate_cc_dev: bad argument");
        return -1;
    }
    if (Lw <= 2 * max_lag || max_lag > 31) {
        set_error("bpmf_intertemplate_cc_dev: need Lw > 2 max_lag and max_lag <= 31 (Lw=%zu max_lag=%zu)",
                  Lw, max_lag);
        return -1;
    }
    const size_t n_ch = S * C, n_rows = T * n_ch, n_lag = 2 * max_lag + 1;
    if (n_rows > 0x7fffffffull || T > 65535 || Lw > 0x7fffffffull) {
        set_error("bpmf_intertemplate_cc_dev: too many templates / channels");
        return -1;
    }
    if (workspace_bytes < bpmf_intertemplate_workspace_bytes(T, S, C, max_lag)) {
        set_error("bpmf_intertemplate_cc_dev: workspace too small");
        return -1;
    }
    float* r_t = (float*)d_workspace;
    float* r_d = (float*)((char*)d_workspace + align_up(n_rows * sizeof(float), 256));
    intertp_norms_kernel<<<dim3((unsigned)((n_rows + 127) / 128)), dim3(128), 0, stream>>>(
        d_waveforms, (int)n_rows, (int)Lw, (int)max_lag, r_t, r_d);
    BPMF_LAUNCH_CHECK();
    // channels of t staged per pass: as many as fit beside the CC and product buffers in 64 KB
    const size_t fixed = (size_t)ITP_UB * n_ch * sizeof(float);
    size_t ch_chunk = n_ch;
    auto need = [&](size_t cc) { return (cc * Lw + (size_t)ITP_UB * cc * n_lag) * sizeof(float) + fixed; };
    while (ch_chunk > 1 && need(ch_chunk) > 64 * 1024) --ch_chunk;
    if (need(ch_chunk) > 64 * 1024) {
        set_error("bpmf_intertemplate_cc_dev: one channel of %zu samples does not fit the LDS budget", Lw);
        return -1;
    }
    dim3 grid((unsigned)((T + ITP_UB - 1) / ITP_UB), (unsigned)T);
    intertp_cc_kernel<<<grid, dim3(ITP_THREADS), need(ch_chunk), stream>>>(
        d_waveforms, d_base_weights, d_pair_mask, r_t, r_d, (int)T, (int)n_ch, (int)Lw, (int)max_lag,
        (int)ch_chunk, d_out);
    BPMF_LAUNCH_CHECK();
    return 0;
}
