// Split-precision matched filter (option mf.split16, off by default): host side.  The kernels and the whole
// design are in mf_split.h; the entry points that reach this file are bpmf_mf_prepare_data_dev and
// bpmf_mf_run_dev (mf.hip), i.e. fast_matched_filter.matched_filter as called at
// BPMF/similarity_search.py:526-533.
#include "mf_split.h"
#include "mf_split_api.h"
#include <algorithm>

namespace bpmf {
namespace sp {

namespace {
struct DayRegion { u32x4* planes; unsigned* maxbits; int* s_exp; float* scd; };
DayRegion carve_day(void* base, size_t N, size_t n_ch)
{
    char* p = (char*)base;
    DayRegion r;
    size_t o = 0;
    r.planes = (u32x4*)(p + o);  o += align_up(n_ch * split_row_bytes(N), 256);
    r.maxbits = (unsigned*)(p + o); o += align_up(n_ch * sizeof(unsigned), 256);
    r.s_exp = (int*)(p + o);     o += align_up(n_ch * sizeof(int), 256);
    r.scd = (float*)(p + o);
    return r;
}
}  // namespace

size_t day_region_bytes(size_t N, size_t n_ch)
{
    return align_up(n_ch * split_row_bytes(N), 256) + 3 * align_up(n_ch * 4, 256);
}

size_t batch_region_bytes(size_t T, size_t n_ch, size_t L)
{
    const size_t n_seg = L ? (size_t)n_segments_of((int)std::min<size_t>(L, (size_t)max_template_len())) : 1;
    return align_up(T * n_ch * n_seg * (size_t)BAND_BYTES, 256) + align_up(T * n_ch * sizeof(float), 256);
}

bool usable(size_t L, size_t N)
{
    return L <= (size_t)max_template_len() && N < ((size_t)1 << 30) - 8192;
}

int prepare_day(const float* d_data, size_t N, size_t n_ch, void* day_region, hipStream_t stream)
{
    const DayRegion r = carve_day(day_region, N, n_ch);
    const size_t NQ = (N + 7) / 8;
    if (n_ch > 65535) {
        set_error("mf.split16: more than 65535 channels");
        return -1;
    }
    BPMF_HIP_CHECK(hipMemsetAsync(r.maxbits, 0, n_ch * sizeof(unsigned), stream));
    const unsigned nb = (unsigned)std::min<size_t>(256, (N + 4095) / 4096);
    sp_absmax_kernel<<<dim3(nb, (unsigned)n_ch), dim3(256), 0, stream>>>(d_data, N, r.maxbits);
    BPMF_LAUNCH_CHECK();
    sp_scale_kernel<<<dim3((unsigned)((n_ch + 63) / 64)), dim3(64), 0, stream>>>(r.maxbits, (int)n_ch, r.s_exp, r.scd);
    BPMF_LAUNCH_CHECK();
    sp_split_data_kernel<<<dim3((unsigned)((NQ + 255) / 256), (unsigned)n_ch), dim3(256), 0, stream>>>(d_data, N, NQ, r.s_exp,
                                                                                                     r.planes);
    BPMF_LAUNCH_CHECK();
    return 0;
}

int run(const float* d_templates, const int32_t* d_moveouts, const void* day_region, void* batch_region,
        const int4* chan_rec, const float* e_d, const int2* range, size_t step, size_t L, size_t N, size_t T,
        size_t n_ch, size_t n_corr, int network_sum, int sqrt_norm, size_t nb_lo, size_t nb_cnt, float* d_cc_out,
        hipStream_t stream)
{
    const DayRegion r = carve_day(const_cast<void*>(day_region), N, n_ch);
    const int n_seg = n_segments_of((int)L), seg_len = segment_len_of((int)L);
    unsigned* bands = (unsigned*)batch_region;
    float* sct = (float*)((char*)batch_region + align_up(T * n_ch * (size_t)n_seg * (size_t)BAND_BYTES, 256));
    if (T * n_ch * (size_t)n_seg >= 0x7fffffffull || T * (nb_cnt + 8) >= 0x7fffffffull) {
        set_error("mf.split16: grid too large");
        return -1;
    }
    sp_band_kernel<<<dim3((unsigned)(T * n_ch * (size_t)n_seg)), dim3(64), 0, stream>>>(d_templates, d_moveouts, (int)L, n_seg,
                                                                                      seg_len, bands, sct);
    BPMF_LAUNCH_CHECK();
    const dim3 grid((unsigned)(8 * ((T * nb_cnt + 7) / 8)));
#define BPMF_SP_LAUNCH(NS, S1)                                                                                       \
    do { if (n_seg > 1) BPMF_SP_LAUNCH2(NS, S1, true); else BPMF_SP_LAUNCH2(NS, S1, false); } while (0)
#define BPMF_SP_LAUNCH2(NS, S1, SG)                                                                                  \
    do {                                                                                                             \
        auto kfn = mf_split_kernel<NS, S1, 0, SG>;                                                                   \
        /* (62 464 bytes of dynamic LDS: below the 64 KB that need no opt-in) */                                     \
        kfn<<<grid, dim3(THREADS), WG_LDS, stream>>>(r.planes, bands, sct, r.scd, chan_rec, e_d, range, (int)L,      \
                                                     (long long)N, (int)T, (int)n_ch, (long long)n_corr, (int)step,   \
                                                     d_cc_out, (int)nb_cnt, (int)nb_lo, sqrt_norm ? 4 : 0, n_seg,     \
                                                     seg_len);                                                        \
    } while (0)
    if (network_sum && step == 1) BPMF_SP_LAUNCH(true, true);
    else if (network_sum) BPMF_SP_LAUNCH(true, false);
    else if (step == 1) BPMF_SP_LAUNCH(false, true);
    else BPMF_SP_LAUNCH(false, false);
#undef BPMF_SP_LAUNCH
#undef BPMF_SP_LAUNCH2
    BPMF_LAUNCH_CHECK();
    return 0;
}

}  // namespace sp
}  // namespace bpmf
