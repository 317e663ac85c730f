/* A plain-C consumer of the C ABI (include/bpmf_hip.h): no Python, no torch, no C++ -- what a maintainer of the
 * reference's back-ends would link.  Built and run by tests/test_gpu_c_abi.py:
 *
 *   gcc -std=c99 -O1 -I include tests/c_abi/abi_smoke.c -o abi_smoke -L seismic_bpmf_amd/lib -lbpmf_hip -lm
 *
 * Runs a small matched filter and a small backprojection through the host-pointer entry points, compares them
 * with a scalar restatement written here in the document order of the conventions (DESIGN.md section 3:
 * fmaf chains; the compile uses -ffp-contract=off so that nothing else is fused), and prints "ok". */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bpmf_hip.h"

static unsigned long long g_state = 88172645463325252ull;
static float rnd(void)
{ /* xorshift: uniform in [-1, 1) */
    g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17;
    return (float)((double)(g_state >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}

static int check_bp(void)
{
    enum { K = 37, S = 5, C = 2, P = 2, N = 4000 };
    float *f = malloc(sizeof(float) * S * C * N), *wp = malloc(sizeof(float) * S * C * P);
    float *ws = malloc(sizeof(float) * K * S), *beam = malloc(sizeof(float) * N), *want = malloc(sizeof(float) * N);
    int32_t *tau = malloc(sizeof(int32_t) * K * S * P), *arg = malloc(sizeof(int32_t) * N), *warg = malloc(sizeof(int32_t) * N);
    float *U = malloc(sizeof(float) * S * P * N);
    for (int i = 0; i < S * C * N; ++i) f[i] = fabsf(rnd());
    for (int i = 0; i < S * C * P; ++i) wp[i] = 0.5f + 0.5f * fabsf(rnd());
    for (int i = 0; i < K * S; ++i) ws[i] = fabsf(rnd()) < 0.25f ? 0.0f : 0.25f + fabsf(rnd());
    for (int i = 0; i < K * S * P; ++i) tau[i] = (int32_t)(fabsf(rnd()) * 180.0f);
    /* prestack: U[s,p,t] = fmaf chain over c */
    for (int s = 0; s < S; ++s)
        for (int p = 0; p < P; ++p)
            for (int t = 0; t < N; ++t) {
                float u = 0.0f;
                for (int c = 0; c < C; ++c) u = fmaf(wp[(s * C + c) * P + p], f[(s * C + c) * N + t], u);
                U[(s * P + p) * N + t] = u;
            }
    /* strict beams, scan from (0, source 0), strictly greater replaces */
    for (int t = 0; t < N; ++t) { want[t] = 0.0f; warg[t] = 0; }
    for (int k = 0; k < K; ++k)
        for (int t = 0; t < N; ++t) {
            int ok = 1, any = 0;
            float b = 0.0f;
            for (int s = 0; s < S && ok; ++s) {
                if (ws[k * S + s] == 0.0f) continue;
                for (int p = 0; p < P; ++p) {
                    const int x = t + tau[(k * S + s) * P + p];
                    if (x < 0 || x >= N) { ok = 0; break; }
                    b = fmaf(ws[k * S + s], U[(s * P + p) * N + x], b);
                    any = 1;
                }
            }
            if (ok && any && b > want[t]) { want[t] = b; warg[t] = k; }
        }
    const int dev[1] = {0};
    int rc = bpmf_bp_run_multi(f, tau, wp, ws, N, K, S, C, P, BPMF_BP_STRICT, BPMF_BP_REDUCE_MAX, 1, dev, beam, arg);
    if (rc) { fprintf(stderr, "bpmf_bp_run_multi: %d %s\n", rc, bpmf_last_error()); return 1; }
    if (memcmp(beam, want, sizeof(float) * N) || memcmp(arg, warg, sizeof(int32_t) * N)) {
        fprintf(stderr, "backprojection differs from the scalar restatement\n");
        return 1;
    }
    free(f); free(wp); free(ws); free(beam); free(want); free(tau); free(arg); free(warg); free(U);
    return 0;
}

static int check_mf(void)
{
    enum { T = 3, S = 2, C = 2, L = 40, N = 3000, NC = N - L + 1 };
    float *tp = malloc(sizeof(float) * T * S * C * L), *d = malloc(sizeof(float) * S * C * N);
    float *w = malloc(sizeof(float) * T * S * C), *cc = malloc(sizeof(float) * T * NC), *want = calloc(T * NC, sizeof(float));
    int32_t *mv = malloc(sizeof(int32_t) * T * S * C);
    double *cs = malloc(sizeof(double) * (N + 1));
    for (int i = 0; i < T * S * C * L; ++i) tp[i] = rnd();
    for (int i = 0; i < S * C * N; ++i) d[i] = rnd();
    for (int i = 0; i < T * S * C; ++i) { w[i] = 0.1f + fabsf(rnd()); mv[i] = (int32_t)(fabsf(rnd()) * 90.0f); }
    for (int t = 0; t < T; ++t) {
        int mv_max = 0;
        for (int ch = 0; ch < S * C; ++ch) if (mv[t * S * C + ch] > mv_max) mv_max = mv[t * S * C + ch];
        const int last = N - L - mv_max;              /* last valid lag, inclusive */
        for (int ch = 0; ch < S * C; ++ch) {          /* (s outer, c inner) = channel order */
            const float *x = tp + (t * S * C + ch) * L, *y = d + ch * N;
            float et = 0.0f;
            for (int l = 0; l < L; ++l) et = fmaf(x[l], x[l], et);
            const float rt = 1.0f / sqrtf(et);
            cs[0] = 0.0;                              /* one 1024-sample hierarchy level suffices to restate: */
            {                                         /* chunk-local sums, then chunk offsets, as the library does */
                double off = 0.0, acc = 0.0;
                for (int n = 0; n < N; ++n) {
                    if (n % 1024 == 0) { off += acc; acc = 0.0; }
                    acc = acc + (double)y[n] * (double)y[n];
                    cs[n + 1] = off + acc;
                }
            }
            for (int i = 0; i <= last; ++i) {
                const int j = i + mv[t * S * C + ch];
                float num = 0.0f;
                for (int l = 0; l < L; ++l) num = fmaf(x[l], y[j + l], num);
                const float rd = 1.0f / sqrtf((float)(cs[j + L] - cs[j]));
                const float nrm = rt * rd;
                const float c1 = nrm < 1000.0f ? num * nrm : 0.0f;
                want[t * NC + i] = fmaf(w[t * S * C + ch], c1, want[t * NC + i]);
            }
        }
    }
    const int dev[1] = {0};
    int rc = bpmf_mf_run_multi(tp, mv, w, d, 1, L, N, T, S, C, NC, 1, 0, 1, dev, cc);
    if (rc) { fprintf(stderr, "bpmf_mf_run_multi: %d %s\n", rc, bpmf_last_error()); return 1; }
    if (memcmp(cc, want, sizeof(float) * T * NC)) {
        int bad = 0;
        for (int i = 0; i < T * NC; ++i) bad += cc[i] != want[i];
        fprintf(stderr, "matched filter differs from the scalar restatement in %d of %d values\n", bad, T * NC);
        return 1;
    }
    free(tp); free(d); free(w); free(cc); free(want); free(mv); free(cs);
    return 0;
}

int main(void)
{
    const int n = bpmf_device_count();
    if (n < 1) { fprintf(stderr, "no HIP device: %s\n", bpmf_last_error()); return 2; }
    char name[128];
    size_t mem = 0;
    int cus = 0;
    if (bpmf_device_info(0, name, sizeof name, &mem, &cus)) { fprintf(stderr, "%s\n", bpmf_last_error()); return 2; }
    if (check_bp() || check_mf()) return 1;
    size_t held_dev = 0, held_pin = 0;
    bpmf_device_memory_held(0, &held_dev, &held_pin);
    if (held_dev == 0) { fprintf(stderr, "the host calls keep no working set?\n"); return 1; }
    if (bpmf_release_device_memory(-1)) return 1;
    bpmf_device_memory_held(0, &held_dev, &held_pin);
    if (held_dev != 0 || held_pin != 0) { fprintf(stderr, "release left %zu / %zu bytes\n", held_dev, held_pin); return 1; }
    printf("ok: %s, %d CUs, %d device(s); matched filter and backprojection equal the scalar restatement bit for bit\n",
           name, cus, n);
    return 0;
}
