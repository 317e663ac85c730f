/*
 * adjacent_oracle.c -- CPU restatement of the native helpers that sit right before
 * and after the two hot paths in the reference.
 *
 * TEST INFRASTRUCTURE ONLY (see bpmf_oracle.c header).  Unlike the hot paths these
 * steps DO exist in the reference (BPMF/libc.c), so this restatement is PINNED: the
 * tests compare it with golden vectors produced by the reference's own libc.c
 * compiled in this container (tests/golden/make_goldens.py, oracle/_ref/libc.so),
 * always single-threaded because the reference's OpenMP loops race.
 *
 * Each function names the reference lines it follows.  The reference's Makefile compiles
 * libc.c with -std=c99: in an ISO C mode gcc does NOT contract a*b+c into an FMA (the built
 * oracle/_ref/libc.so holds no vfmadd instruction at all), so every multiply-add here is a
 * separate multiply and add, and this file is itself built with -ffp-contract=off.  (Round 1
 * had assumed fused operations at two places of time_dependent_threshold; the golden vectors
 * happened not to tell the difference, the live comparison of tests/test_reference_live.py
 * did: 1-ulp steps on a few windows.)
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#define GAUSSIAN_SAMPLE_LEN 500

/* mean and population std of a window, float accumulators, squares formed in
 * double (the reference calls pow(x - mean, 2)).   BPMF/libc.c:488-514 */
static float window_mean(const float *x, size_t n)
{
    float acc = 0.0f;
    for (size_t i = 0; i < n; i++) acc += x[i];
    return acc / (float)n;
}

static float window_std(const float *x, float mean, size_t n)
{
    float acc = 0.0f;
    for (size_t i = 0; i < n; i++) {
        double d = (double)(x[i] - mean);
        acc = (float)((double)acc + d * d);
    }
    return sqrtf(acc / (float)n);
}

/*
 * RMS time-dependent detection threshold.            BPMF/libc.c:516-673
 *   series        (n)   CC time series; NOT modified (the reference works on a copy
 *                       made by its Python wrapper, clib.py:284)
 *   gaussian      (500) standard-normal sample used to fill exact zeros
 *   scratch       (n)   work copy of the series with zeros filled
 *   threshold     (n)   output
 * Deviation from the reference: the interior index i / shift is clamped to
 * n_windows-1 (the reference reads one past the end for some (n, window, overlap>0)
 * combinations -- SURVEY.md section 8c).
 * Returns the number of sliding windows, or -1 if the sizes admit no window.
 */
long tdt_rms_cpu(const float *series, const float *gaussian, float num_dev, size_t n,
                 size_t half_window, size_t shift, float *scratch, float *threshold)
{
    const size_t window = 2 * half_window;
    /* the reference computes (n - (window - shift)) / shift in size_t: an odd window with overlap 0
     * gives shift = window + 1 here (window = 2 * (odd // 2)), the inner difference wraps to -1 and
     * the count is (n + 1) / shift -- every window still lies inside the series */
    if (window == 0 || shift == 0 || shift > window + 1 || n < window) return -1;
    const size_t n_win = (n - (window - shift)) / shift;
    if (n_win < 1) return -1;

    /* global centre / deviation over the non-zero samples of the first
     * floor(n/window) whole windows, per-window float partials   libc.c:553-587 */
    const size_t n_glob = n / window;
    float centre = 0.0f;
    size_t n_nonzero = 0;
    for (size_t q = 0; q < n_glob; q++) {
        const float *x = series + q * window;
        float part = 0.0f;
        size_t cnt = 0;
        for (size_t j = 0; j < window; j++)
            if (x[j] != 0.0f) { part += x[j]; cnt++; }
        centre += part;
        n_nonzero += cnt;
    }
    centre /= (float)n_nonzero;
    float dev = 0.0f;
    for (size_t q = 0; q < n_glob; q++) {
        const float *x = series + q * window;
        float part = 0.0f;
        for (size_t j = 0; j < window; j++)
            if (x[j] != 0.0f) {
                double d = (double)(x[j] - centre);
                part = (float)((double)part + d * d);
            }
        dev += part;
    }
    dev = sqrtf(dev / (float)n_nonzero);

    /* zeros -> centre + g[i mod 500] * dev                     libc.c:606-612 */
    for (size_t i = 0; i < n; i++)
        scratch[i] = (series[i] == 0.0f)
                         ? centre + gaussian[i % GAUSSIAN_SAMPLE_LEN] * dev
                         : series[i];

    float *win = (float *)malloc(n_win * sizeof(float));
    if (!win) return -1;
    /* mean + num_dev * std per sliding window                   libc.c:615-627 */
    for (size_t q = 0; q < n_win; q++) {
        const float *x = scratch + q * shift;
        float m = window_mean(x, window);
        float s = window_std(x, m, window);
        win[q] = m + num_dev * s;
    }
    /* "delay the jump": a drop is postponed by one window, a rise is
     * anticipated by one window (always keep the larger value)  libc.c:631-651 */
    if (n_win > 1) {
        float *diff = (float *)malloc((n_win - 1) * sizeof(float));
        if (!diff) { free(win); return -1; }
        for (size_t q = 0; q + 1 < n_win; q++) diff[q] = win[q + 1] - win[q];
        for (size_t q = 1; q < n_win; q++) {
            if (diff[q - 1] < 0.0f) win[q] -= diff[q - 1];
            diff[q - 1] = win[q] - win[q - 1];
        }
        for (size_t q = 0; q + 1 < n_win; q++)
            if (diff[q] > 0.0f) win[q] += diff[q];
        free(diff);
    }
    /* step-wise expansion                                        libc.c:654-669 */
    for (size_t i = 0; i < n; i++) {
        size_t q;
        if (i < shift) q = 0;
        else if (i >= n - shift) q = n_win - 1;
        else { q = i / shift; if (q > n_win - 1) q = n_win - 1; }
        threshold[i] = win[q];
    }
    free(win);
    return (long)n_win;
}

/*
 * Greedy peak de-clustering of a CC series.                BPMF/libc.c:441-485
 * selection[i] = 1 iff cc[i] > threshold[i] and no sample in the preceding
 * search_win samples is larger; an accepted/rejected sample also clears every
 * earlier sample of its look-back window that it dominates.
 */
void select_cc_indexes_cpu(const float *cc, const float *threshold, size_t search_win,
                           size_t n, int32_t *selection)
{
    for (size_t i = 0; i < n; i++) {
        selection[i] = cc[i] > threshold[i];
        size_t j0 = i <= search_win ? 0 : i - search_win;
        for (size_t j = j0; j < i; j++) {
            if (cc[j] > cc[i]) { selection[i] = 0; break; }
            selection[j] = 0;
        }
    }
}

/*
 * Naive running kurtosis over windows of W samples.          BPMF/libc.c:11-53
 * signal/kurto (S, C, n).  kurto[n] is written for n >= W from the window
 * [n-W, n) when its variance exceeds 1e-6; other samples are left untouched
 * (the caller zero-initialises).
 */
void kurtosis_cpu(const float *signal, int W, int n_stations, int n_components,
                  int length, float *kurto)
{
    const float Wf = (float)W;
    for (long ch = 0; ch < (long)n_stations * n_components; ch++) {
        const float *x = signal + ch * length;
        float *k = kurto + ch * length;
        for (int n = W; n < length; n++) {
            const float *w = x + (n - W);
            float mean = 0.0f, m2 = 0.0f, m4 = 0.0f;
            for (int i = 0; i < W; i++) mean += w[i];
            mean /= Wf;
            for (int i = 0; i < W; i++) {
                double d = (double)(w[i] - mean);
                m2 = (float)((double)m2 + d * d);
                m4 = (float)((double)m4 + (d * d) * (d * d));
            }
            m2 /= Wf;
            m4 /= Wf;
            if (m2 > 0.000001)
                k[n] = (float)(1. / (double)((Wf - 2) * (Wf - 3)) *
                               ((pow(Wf, 2) - 1) * m4 / pow(m2, 2) - 3 * pow(Wf - 1., 2)));
        }
    }
}

/* ---- grid decimation by moveout similarity        BPMF/libc.c:55-223, 225-387 ---- */

static void argsort_small(const float *v, size_t n, size_t *idx)
{
    /* selection sort on indexes: same tie behaviour as libc.c:389-410 */
    for (size_t i = 0; i < n; i++) idx[i] = i;
    for (size_t i = 0; i + 1 < n; i++) {
        size_t m = i;
        for (size_t j = i + 1; j < n; j++)
            if (v[idx[j]] < v[idx[m]]) m = j;
        size_t tmp = idx[m]; idx[m] = idx[i]; idx[i] = tmp;
    }
}

static void sort_small(float *v, size_t n)
{
    for (size_t i = 0; i + 1 < n; i++) {
        size_t m = i;
        for (size_t j = i + 1; j < n; j++)
            if (v[j] < v[m]) m = j;
        float tmp = v[m]; v[m] = v[i]; v[i] = tmp;
    }
}

/* summed squared moveout difference of sources a, b over n_diff stations.
 * mode 0 ("smallest", libc.c:55-223): the n_diff smallest squared differences.
 * mode 1 ("closest",  libc.c:225-387): the n_diff stations closest to source a
 *                                       (order[] = argsort of a's moveouts). */
static float pair_dt2(const float *ma, const float *mb, size_t n_stations, size_t n_diff,
                      int mode, const size_t *order, float *work)
{
    float dt2 = 0.0f;
    if (mode == 1) {
        for (size_t s = 0; s < n_diff; s++) {
            double d = (double)(ma[order[s]] - mb[order[s]]);
            dt2 = (float)((double)dt2 + d * d);
        }
    } else {
        for (size_t s = 0; s < n_stations; s++) {
            double d = (double)(ma[s] - mb[s]);
            work[s] = (float)(d * d);
        }
        sort_small(work, n_stations);
        for (size_t s = 0; s < n_diff; s++) dt2 += work[s];
    }
    return dt2;
}

static int in_cell(float lon, float lat, const float *clon, const float *clat, size_t i,
                   size_t j)
{
    return !(lon < clon[i] || lon >= clon[i + 1] || lat < clat[j] || lat >= clat[j + 1]);
}

/*
 * redundant[] must come in zeroed; on return redundant[n] = 1 for every source
 * that a lower-indexed kept source makes redundant.  First pass restricted to
 * pairs inside the same (lon, lat) cell, second pass over all remaining pairs;
 * both greedy in ascending n1, exactly as the reference's sequential loops.
 */
void similar_moveouts_cpu(const float *moveouts, const float *lon, const float *lat,
                          const float *cell_lon, const float *cell_lat, float threshold,
                          size_t n_sources, size_t n_stations, size_t n_cells_lon,
                          size_t n_cells_lat, size_t n_diff, int mode, int32_t *redundant)
{
    const float thr2 = (float)((double)(float)n_diff * pow((double)threshold, 2));
    size_t *order = (size_t *)malloc(n_stations * sizeof(size_t));
    float *work = (float *)malloc(n_stations * sizeof(float));
    if (!order || !work || n_sources < 2) { free(order); free(work); return; }
    for (size_t i = 0; i < n_cells_lon; i++)
        for (size_t j = 0; j < n_cells_lat; j++)
            for (size_t a = 0; a + 1 < n_sources; a++) {
                if (!in_cell(lon[a], lat[a], cell_lon, cell_lat, i, j) || redundant[a]) continue;
                const float *ma = moveouts + a * n_stations;
                if (mode == 1) argsort_small(ma, n_stations, order);
                for (size_t b = a + 1; b < n_sources; b++) {
                    if (!in_cell(lon[b], lat[b], cell_lon, cell_lat, i, j) || redundant[b]) continue;
                    if (pair_dt2(ma, moveouts + b * n_stations, n_stations, n_diff, mode, order, work) < thr2)
                        redundant[b] = 1;
                }
            }
    for (size_t a = 0; a + 1 < n_sources; a++) {
        if (redundant[a]) continue;
        const float *ma = moveouts + a * n_stations;
        if (mode == 1) argsort_small(ma, n_stations, order);
        for (size_t b = a + 1; b < n_sources; b++) {
            if (redundant[b]) continue;
            if (pair_dt2(ma, moveouts + b * n_stations, n_stations, n_diff, mode, order, work) < thr2)
                redundant[b] = 1;
        }
    }
    free(order);
    free(work);
}
