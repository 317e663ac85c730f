// Host-side interface of the split-precision matched filter (mf_split.hip / mf_split.h; option mf.split16) for
// the entry points of mf.hip: where its arrays live in the matched-filter workspace and the two launches.
#pragma once
#include "common.h"

namespace bpmf {
namespace sp {

// The per-day region (behind the window norms, in front of everything that depends on the template count):
// the split data [n_ch][ceil(N / 8)][2][8] fp16, the channel maxima, scale exponents and 2^-s.
// The per-batch region (at the end): the band images [T][n_ch][segments][4096 B] and 2^-s of every template channel.
size_t day_region_bytes(size_t N, size_t n_ch);
size_t batch_region_bytes(size_t T, size_t n_ch, size_t L);

// can this launch take the split kernel?  (templates of up to 4096 samples, in segments of at most 376; N < 2^30 - 8192)
bool usable(size_t L, size_t N);

// once per day, behind bpmf_mf_prepare_data_dev's own kernels: channel maxima -> scales -> split planes
int prepare_day(const float* d_data, size_t N, size_t n_ch, void* day_region, hipStream_t stream);

// one launch over the lag blocks [nb_lo, nb_lo + nb_cnt) of 8192 data offsets: band images, then the kernel.
// chan_rec / range / e_d: what mf_prologue_kernel and the per-day preparation left in the workspace (under
// sqrt_norm, option mf.compat_sqrt_norm, they hold energies and the epilogue divides by sqrtf(E_t * E_d)).
int run(const float* d_templates, const int32_t* d_moveouts, const void* day_region, void* batch_region,
        const int4* chan_rec, const float* e_d, const int2* range, size_t step, size_t L, size_t N, size_t T,
        size_t n_ch, size_t n_corr, int network_sum, int sqrt_norm, size_t nb_lo, size_t nb_cnt, float* d_cc_out,
        hipStream_t stream);

constexpr size_t LAGS_PER_WG = 8192;

}  // namespace sp
}  // namespace bpmf
