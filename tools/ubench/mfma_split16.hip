// Round 6, step 1 of the split-precision matched filter: the kernel of
// seismic_bpmf_amd/csrc/mf_split.h -- fp16 hi/lo split of data and templates, three v_mfma_f32_32x32x16_f16
// products per k-step, fp32 accumulation, real operand delivery (windows and bands staged per wave through
// LDS, norms and the weighted channel sum in the epilogue) -- on a synthetic day, timed against the rate of the
// production exact-fp32 kernel and checked against a float64 brute force.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench/mfma_split16.hip -o tools/ubench/mfma_split16.bin
//   tools/ubench/mfma_split16.bin [T] [n_ch] [L] [N] [iters]
//
// Prints channel-lags/s (one channel-lag = one CC of one template channel at one lag = 2 L direct-form flop),
// the direct-form TFLOP/s, and max |cc_sum - float64| over sampled lags, beside the same error of an fp32 fmaf
// chain (what the production kernel computes).  Also: does the matrix pipe honour fp16 subnormal inputs?
#include "../../seismic_bpmf_amd/csrc/mf_split.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace bpmf::sp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// one MFMA with A = a subnormal fp16 in row 0 / k 0, B = 2^10 in k 0 / column 0: out[0][0] = sub * 1024 or 0
__global__ void subnormal_probe(float* out)
{
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.0f; b[i] = (_Float16)0.0f; }
    if (lane == 0) {
        a[0] = __builtin_bit_cast(_Float16, (unsigned short)0x0011);   // 17 * 2^-24
        b[0] = (_Float16)1024.0f;
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];
}

int main(int argc, char** argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 64;
    const int n_ch = argc > 2 ? atoi(argv[2]) : 60;
    const int L = argc > 3 ? atoi(argv[3]) : 256;
    const long long N = argc > 4 ? atoll(argv[4]) : 2000000;
    const int iters = argc > 5 ? atoi(argv[5]) : 5;
    const int mode = argc > 6 ? atoi(argv[6]) : 0;     // 1: all-zero data and templates (does the clock rise?)
    const int max_mv = 3000;
    const long long n_corr = N - L + 1, nwin = n_corr;
    if (L > max_template_len()) { fprintf(stderr, "L too long\n"); return 1; }
    const int n_seg = n_segments_of(L), seg_len = segment_len_of(L);
    printf("split16 ubench: T %d, channels %d, L %d, N %lld, k-steps %d\n", T, n_ch, L, N, nks_of(seg_len) * n_seg);

    {
        float* d_o; float h = -1.0f;
        CK(hipMalloc(&d_o, 4));
        subnormal_probe<<<1, 64>>>(d_o);
        CK(hipMemcpy(&h, d_o, 4, hipMemcpyDeviceToHost));
        printf("fp16 subnormal input through v_mfma_f32_32x32x16_f16: 17 * 2^-24 * 1024 = %.9g (expected %.9g): %s\n",
               h, 17.0 / 16384.0, h == (float)(17.0 / 16384.0) ? "honoured" : "FLUSHED");
        CK(hipFree(d_o));
    }

    std::mt19937_64 rng(20260930);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    std::vector<float> data((size_t)n_ch * N), tmpl((size_t)T * n_ch * L), w((size_t)T * n_ch);
    std::vector<int> mv((size_t)T * n_ch);
    for (auto& x : data) x = nd(rng);
    // a few channels with awkward scales / spikes: physical units, a large glitch
    if (n_ch > 3) {
        for (long long i = 0; i < N; ++i) data[(size_t)1 * N + i] *= 3.0e-9f;
        data[(size_t)2 * N + N / 3] = 4.0e4f;
        for (long long i = 0; i < N; ++i) data[(size_t)3 * N + i] *= 7.0e5f;
    }
    for (int tc = 0; tc < T * n_ch; ++tc) {
        // 5-tap moving average of white noise, demeaned, unit rms
        std::vector<float> raw(L + 4);
        for (auto& x : raw) x = nd(rng);
        double mean = 0, ss = 0;
        float* tp = &tmpl[(size_t)tc * L];
        for (int l = 0; l < L; ++l) { tp[l] = 0.2f * (raw[l] + raw[l + 1] + raw[l + 2] + raw[l + 3] + raw[l + 4]); mean += tp[l]; }
        mean /= L;
        for (int l = 0; l < L; ++l) { tp[l] -= (float)mean; ss += (double)tp[l] * tp[l]; }
        const float inv = (float)(1.0 / std::sqrt(ss / L));
        for (int l = 0; l < L; ++l) tp[l] *= inv;
        mv[tc] = (int)(rng() % max_mv);
        w[tc] = 1.0f / n_ch;
    }
    // plant template 0 in every channel at lag 100000 (a CC sum near 0.9)
    const long long plant = std::min<long long>(100000, n_corr / 2);
    for (int ch = 0; ch < n_ch; ++ch) {
        double scale = 1.0;
        if (n_ch > 3 && ch == 1) scale = 3.0e-9;
        if (n_ch > 3 && ch == 3) scale = 7.0e5;
        for (int l = 0; l < L; ++l) data[(size_t)ch * N + plant + mv[ch] + l] += (float)(2.0 * scale * tmpl[(size_t)ch * L + l]);
    }
    if (mode & 1) { std::fill(data.begin(), data.end(), 0.0f); std::fill(tmpl.begin(), tmpl.end(), 0.0f); printf("ZERO data and templates\n"); }
    // norms on the host (double prefix sums; the product path has its own bit-exact kernels for these)
    std::vector<float> r_d((size_t)n_ch * nwin), r_t((size_t)T * n_ch);
    {
        std::vector<double> cs(N + 1);
        for (int ch = 0; ch < n_ch; ++ch) {
            cs[0] = 0;
            for (long long i = 0; i < N; ++i) { const double v = data[(size_t)ch * N + i]; cs[i + 1] = cs[i] + v * v; }
            for (long long j = 0; j < nwin; ++j) r_d[(size_t)ch * nwin + j] = 1.0f / sqrtf((float)(cs[j + L] - cs[j]));
        }
        for (int tc = 0; tc < T * n_ch; ++tc) {
            float acc = 0;
            for (int l = 0; l < L; ++l) acc = fmaf(tmpl[(size_t)tc * L + l], tmpl[(size_t)tc * L + l], acc);
            r_t[tc] = 1.0f / sqrtf(acc);
        }
    }
    std::vector<int4> rec((size_t)T * (n_ch + 2));
    std::vector<int2> range(T);
    for (int t = 0; t < T; ++t) {
        int mx = 0;
        for (int ch = 0; ch < n_ch; ++ch) {
            int4 r;
            r.x = ch; r.y = mv[t * n_ch + ch];
            memcpy(&r.z, &w[t * n_ch + ch], 4);
            memcpy(&r.w, &r_t[t * n_ch + ch], 4);
            rec[(size_t)t * (n_ch + 2) + ch] = r;
            mx = std::max(mx, r.y);
        }
        rec[(size_t)t * (n_ch + 2) + n_ch] = make_int4(-1, 0, 0, 0);
        rec[(size_t)t * (n_ch + 2) + n_ch + 1] = make_int4(-1, 0, 0, 0);
        range[t] = make_int2(0, (int)(N - L - mx));
    }

    float *d_data, *d_tmpl, *d_rd, *d_sct, *d_scd, *d_out;
    int *d_mv, *d_sexp;
    unsigned *d_max, *d_bands;
    u32x4* d_split;
    int4* d_rec;
    int2* d_range;
    const size_t NQ = (N + 7) / 8;
    CK(hipMalloc(&d_data, data.size() * 4));
    CK(hipMalloc(&d_tmpl, tmpl.size() * 4));
    CK(hipMalloc(&d_rd, r_d.size() * 4 + 256));
    CK(hipMalloc(&d_sct, (size_t)T * n_ch * 4));
    CK(hipMalloc(&d_scd, n_ch * 4));
    CK(hipMalloc(&d_sexp, n_ch * 4));
    CK(hipMalloc(&d_max, n_ch * 4));
    CK(hipMalloc(&d_mv, mv.size() * 4));
    CK(hipMalloc(&d_bands, (size_t)T * n_ch * n_seg * BAND_BYTES));
    CK(hipMalloc(&d_split, (size_t)n_ch * NQ * 32));
    CK(hipMalloc(&d_rec, rec.size() * sizeof(int4)));
    CK(hipMalloc(&d_range, range.size() * sizeof(int2)));
    CK(hipMalloc(&d_out, (size_t)T * n_corr * 4));
    CK(hipMemcpy(d_data, data.data(), data.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tmpl, tmpl.data(), tmpl.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rd, r_d.data(), r_d.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_mv, mv.data(), mv.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rec, rec.data(), rec.size() * sizeof(int4), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_range, range.data(), range.size() * sizeof(int2), hipMemcpyHostToDevice));

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms;
    // per-day preparation
    CK(hipEventRecord(e0));
    CK(hipMemsetAsync(d_max, 0, n_ch * 4));
    sp_absmax_kernel<<<dim3(64, n_ch), 256>>>(d_data, (size_t)N, d_max);
    sp_scale_kernel<<<(n_ch + 63) / 64, 64>>>(d_max, n_ch, d_sexp, d_scd);
    sp_split_data_kernel<<<dim3((unsigned)((NQ + 255) / 256), n_ch), 256>>>(d_data, (size_t)N, NQ, d_sexp, d_split);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("split of the day (%d x %lld samples): %.3f ms\n", n_ch, N, ms);
    CK(hipEventRecord(e0));
    sp_band_kernel<<<T * n_ch * n_seg, 64>>>(d_tmpl, d_mv, L, n_seg, seg_len, d_bands, d_sct);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("band images (%d): %.3f ms\n", T * n_ch, ms);

    const int n_lag_blocks = (int)((n_corr + LAGS_WG - 1) / LAGS_WG);
    const unsigned grid = 8u * (unsigned)(((size_t)T * n_lag_blocks + 7) / 8);
    auto run = [&](auto kfn, const char* what, int prio) {
        CK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS));
        for (int it = 0; it < iters + 1; ++it) {
            if (it == 1) CK(hipEventRecord(e0));
            kfn<<<grid, THREADS, WG_LDS>>>(d_split, d_bands, d_sct, d_scd, d_rec, d_rd, d_range, L, N, T, n_ch, n_corr, 1,
                                           d_out, n_lag_blocks, 0, prio, n_seg, seg_len);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        const double chlags = (double)T * n_corr * n_ch;
        printf("%s prio %d: %.3f ms per launch: %.3e channel-lags/s, %.1f TFLOP/s direct-form (fp32 peak 157.3: x %.2f; production "
               "kernel 2.62e11 channel-lags/s at cfg2: x %.2f)\n", what, prio, ms, chlags / (ms * 1e-3),
               2.0 * L * chlags / (ms * 1e-3) / 1e12, 2.0 * L * chlags / (ms * 1e-3) / 1e12 / 157.3, chlags / (ms * 1e-3) / 2.62e11);
    };
    if (n_seg > 1) {
        for (int prio = 1; prio >= 0; --prio) run(mf_split_kernel<true, true, 0, true>, "full kernel (segments)", prio);
    } else {
        run(mf_split_kernel<true, true, 2>, "K loop alone      ", 0);
        run(mf_split_kernel<true, true, 1>, "no norms          ", 0);
        for (int prio = 3; prio >= 0; --prio) run(mf_split_kernel<true, true, 0>, "full kernel       ", prio);
    }
    // check sampled lags against float64 (and the fp32 chain against the same)
    std::vector<float> out((size_t)T * n_corr);
    CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    double err_split = 0, err_f32 = 0, peak = 0, d_split_f32 = 0;
    std::mt19937_64 pick(7);
    const int n_pick = 3000;
    for (int s = 0; s < n_pick; ++s) {
        int t = (int)(pick() % T);
        long long lag = (long long)(pick() % (unsigned long long)(range[t].y + 1));
        if (s < 64) { t = 0; lag = plant - 32 + s; }                        // around the planted event
        if (s >= 64 && s < 128) { t = s % T; lag = range[t].y - (s - 64); }  // the last valid lags
        if (s >= 128 && s < 192) { t = s % T; lag = s - 128; }               // the first lags
        double ref = 0, ref32 = 0;
        float sum32 = 0;
        for (int ch = 0; ch < n_ch; ++ch) {
            const float* tp = &tmpl[((size_t)t * n_ch + ch) * L];
            const float* d = &data[(size_t)ch * N + lag + mv[t * n_ch + ch]];
            double num = 0, et = 0, ed = 0;
            float n32 = 0;
            for (int l = 0; l < L; ++l) {
                num += (double)tp[l] * d[l]; et += (double)tp[l] * tp[l]; ed += (double)d[l] * d[l];
                n32 = fmaf(tp[l], d[l], n32);
            }
            // (the build's rule: a channel whose norms' product reaches 1000 contributes 0)
            if (1.0 / std::sqrt(et * ed) < 1000.0) ref += (double)w[t * n_ch + ch] * num / std::sqrt(et * ed);
            const float nrm = r_t[t * n_ch + ch] * r_d[(size_t)ch * nwin + lag + mv[t * n_ch + ch]];
            sum32 = fmaf(w[t * n_ch + ch], nrm < 1000.0f ? n32 * nrm : 0.0f, sum32);
        }
        ref32 = sum32;
        const double got = out[(size_t)t * n_corr + lag];
        err_split = std::max(err_split, std::fabs(got - ref));
        err_f32 = std::max(err_f32, std::fabs(ref32 - ref));
        d_split_f32 = std::max(d_split_f32, std::fabs(got - ref32));
        peak = std::max(peak, std::fabs(ref));
    }
    // lags behind the valid range must be exactly 0
    long long nonzero_outside = 0;
    for (int t = 0; t < T; ++t)
        for (long long i = range[t].y + 1; i < n_corr; ++i) nonzero_outside += out[(size_t)t * n_corr + i] != 0.0f;
    printf("max |cc_sum - float64| over %d sampled lags: split16 %.3e, fp32 chain %.3e (largest |cc_sum| sampled %.3f; tolerance "
           "2e-5 * sum|w| = 2e-5); max |split16 - fp32 chain| %.3e; non-zero values outside the valid ranges: %lld\n", n_pick, err_split,
           err_f32, peak, d_split_f32, nonzero_outside);
    printf("planted event: cc_sum[0, %lld] = %.6f\n", plant, out[plant]);
    return err_split < 2e-5 && nonzero_outside == 0 ? 0 : 2;
}
