"""GPU: the device post-CC step against golden vectors of the reference's own libc.c
(tests/golden/tdt_rms.npz) and against the pinned CPU oracle.  Bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_tdt_rms_matches_reference_goldens():
    import torch
    from seismic_bpmf_amd.threshold import ThresholdGPU
    g = np.load(os.path.join(GOLD, "tdt_rms.npz"))
    th = ThresholdGPU()
    for j in range(int(g["n_cases"])):
        x = torch.as_tensor(g[f"x_{j}"], device="cuda")
        _, full = th.time_dependent_threshold(x, int(g[f"window_{j}"]), float(g[f"num_dev_{j}"]),
                                              overlap=float(g[f"overlap_{j}"]),
                                              white_noise=g[f"white_noise_{j}"], expand=True)
        got = full.cpu().numpy()[0]
        want = g[f"thr_{j}"]
        assert np.array_equal(got, want), f"case {j}: max |diff| {np.abs(got - want).max()}"


def test_tdt_rms_batched_rows_and_candidates(oracle_lib):
    """Many rows at once (as for a (T, n_corr) CC matrix) + candidate extraction."""
    import torch
    from seismic_bpmf_amd.threshold import ThresholdGPU
    rng = np.random.default_rng(3)
    rows, n, window, overlap = 7, 60_000, 6000, 0.25
    x = (0.02 * rng.standard_normal((rows, n))).astype(np.float32)
    x[2, 1000:4000] = 0.0
    for r in range(rows):
        for p in rng.integers(200, n - 200, 9):
            x[r, p] += rng.uniform(0.3, 0.9)
    wn = rng.standard_normal(500).astype(np.float32)
    th = ThresholdGPU()
    xd = torch.as_tensor(x, device="cuda")
    thr_win, full = th.time_dependent_threshold(xd, window, 8.0, overlap=overlap, white_noise=wn,
                                                expand=True)
    full = full.cpu().numpy()
    for r in range(rows):
        want = oracle_lib.time_dependent_threshold(x[r], window, 8.0, overlap, wn)
        assert np.array_equal(full[r], want), r
    cap = np.full(rows, 0.5, dtype=np.float32)
    cand = th.extract_candidates(xd, thr_win, window, overlap=overlap, row_cap=cap)
    thr_capped = np.minimum(full, cap[:, None])
    rr, ii = np.nonzero(x > thr_capped)
    assert np.array_equal(cand["row"], rr) and np.array_equal(cand["index"], ii)
    assert np.array_equal(cand["cc"], x[rr, ii]) and np.array_equal(cand["threshold"], thr_capped[rr, ii])
    assert 20 < cand.size < 500


def test_end_to_end_planted_events_are_detected_at_exact_samples(oracle_lib):
    """MF on device -> threshold on device -> candidates -> host merge: the planted events come
    out at exactly the planted CC indices, identical to the CPU oracle pipeline."""
    import torch
    from seismic_bpmf_amd import MatchedFilterGPU, postprocess as pp, synthetic as syn
    from seismic_bpmf_amd.threshold import ThresholdGPU
    mf_in = syn.make_mf_inputs(T=3, S=5, C=3, L=64, N=120_000, seed=11, max_moveout=300, n_events=4)
    mf = MatchedFilterGPU()
    mf.set_data(mf_in["data"])
    cc = mf.run(mf_in["templates"], mf_in["moveouts"], mf_in["weights"], 1)
    cc_ref = oracle_lib.matched_filter(mf_in["templates"], mf_in["moveouts"], mf_in["weights"], mf_in["data"], 1)
    assert np.array_equal(cc.cpu().numpy(), cc_ref)
    window, overlap, wn = 20_000, 0.25, np.random.default_rng(1).standard_normal(500).astype(np.float32)
    th = ThresholdGPU()
    thr_win, _ = th.time_dependent_threshold(cc, window, 8.0, overlap=overlap, white_noise=wn)
    cand = th.extract_candidates(cc, thr_win, window, overlap=overlap)
    for t in range(3):
        thr_ref = oracle_lib.time_dependent_threshold(cc_ref[t], window, 8.0, overlap, wn)
        # host merge of the sparse candidates == reference-style selection on the full series
        want = pp.select_cc_indexes(cc_ref[t], thr_ref, 500, step=1, sr=100.0, data_duration_sec=1e9,
                                    n_dev_threshold=8.0, min_freq_hz=2.0, data_buffer_sec=0.0,
                                    remove_edges=False, anomalous_cdf_at_mean_plus_1sig=0.0)
        mine = cand[cand["row"] == t]
        sparse = np.zeros_like(cc_ref[t])
        sparse[mine["index"]] = mine["cc"]
        got = pp.select_cc_indexes(sparse, np.where(sparse > 0, 0.0, np.inf).astype(np.float32), 500,
                                   step=1, sr=100.0, data_duration_sec=1e9, n_dev_threshold=8.0,
                                   min_freq_hz=2.0, data_buffer_sec=0.0, remove_edges=False,
                                   anomalous_cdf_at_mean_plus_1sig=0.0)
        assert np.array_equal(got, want)
        planted = sorted(i0 for tt, i0 in mf_in["planted"] if tt == t)
        assert list(got) == planted, (t, got, planted)


def test_mad_threshold_on_device_equals_the_host_mirror_and_the_reference_golden():
    """threshold_type="mad" (BPMF/similarity_search.py:1079-1113): the device pipeline of
    csrc/stats.hip vs the host mirror (pinned to tests/golden/tdt_mad.npz, the reference's own output),
    bit for bit."""
    import os
    import torch
    from seismic_bpmf_amd import postprocess as pp
    from seismic_bpmf_amd.threshold import ThresholdGPU
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tdt_mad.npz"))
    th = ThresholdGPU()
    got = th.time_dependent_threshold_mad(torch.as_tensor(g["x"], device="cuda"), int(g["window"]), float(g["n_dev"]),
                                          overlap=float(g["overlap"]), white_noise=g["white_noise"]).cpu().numpy()
    assert got.dtype == np.float32 and np.array_equal(got, g["thr"])
    rng = np.random.default_rng(3)
    for n, W, ov in [(30_001, 2001, 0.66), (20_000, 1000, 0.25), (7_777, 500, 0.9)]:
        x = (rng.standard_normal(n) * 0.05).astype(np.float32)
        x[rng.random(n) < 0.02] = 0.0
        x[1000:1400] = 0.0
        wn = rng.standard_normal(int((x == 0).sum())).astype(np.float32)
        want = pp.time_dependent_threshold_mad(x, W, 8.0, overlap=ov, white_noise=wn)
        got = th.time_dependent_threshold_mad(torch.as_tensor(x, device="cuda"), W, 8.0, overlap=ov,
                                              white_noise=wn).cpu().numpy()
        assert np.array_equal(got, want), (n, W, ov)


def test_mad_threshold_batched_rows_and_candidates():
    """All rows of a CC matrix in one call: window values, expanded threshold and the candidates above
    it (bpmf_extract_candidates_mad_dev) equal the host mirror row by row -- rows with different
    numbers of zeros sharing one white-noise vector, a row without zeros, a row of zeros only at the
    ends (the shape of a real CC series), even and odd windows, a row cap."""
    import torch
    from seismic_bpmf_amd import postprocess as pp
    from seismic_bpmf_amd.threshold import ThresholdGPU
    th = ThresholdGPU()
    rng = np.random.default_rng(17)
    for n, W, ov in [(40_000, 3000, 0.66), (25_001, 1001, 0.25), (9_000, 4500, 0.5)]:
        rows = 6
        x = (rng.standard_normal((rows, n)) * 0.05).astype(np.float32)
        x[0, rng.random(n) < 0.05] = 0.0
        x[1, :700] = 0.0
        x[1, -900:] = 0.0
        x[3, 5000:5600] = 0.0
        x[4] = np.round(x[4] * 50) / 50                              # ties
        x[:, ::977] += 0.6                                           # peaks above the threshold
        wn = rng.standard_normal(n).astype(np.float32)
        thr_win, full = th.time_dependent_threshold_mad(torch.as_tensor(x, device="cuda"), W, 8.0, overlap=ov,
                                                        white_noise=wn)
        full = full.cpu().numpy()
        for r in range(rows):
            want = pp.time_dependent_threshold_mad(x[r], W, 8.0, overlap=ov, white_noise=wn)
            assert np.array_equal(full[r], want), (n, W, ov, r)
        cap = np.array([10.0, 10.0, 0.3, 10.0, 10.0, 10.0], np.float32)
        cand = th.extract_candidates(torch.as_tensor(x, device="cuda"), thr_win, W, overlap=ov, row_cap=cap, kind="mad",
                                     capacity=64)
        for r in range(rows):
            t_r = np.minimum(full[r], cap[r])
            idx = np.flatnonzero(x[r] > t_r)
            mine = cand[cand["row"] == r]
            assert np.array_equal(mine["index"], idx), (n, W, ov, r)
            assert np.array_equal(mine["threshold"], t_r[idx]) and np.array_equal(mine["cc"], x[r, idx])
    short = torch.zeros(10, device="cuda")
    with pytest.raises(ValueError):
        th.time_dependent_threshold_mad(short, 100, 8.0)
    few = torch.zeros((1, 5000), device="cuda")
    with pytest.raises(Exception, match="white-noise"):
        th.time_dependent_threshold_mad(few, 1000, 8.0, white_noise=np.zeros(10, np.float32))


def test_row_median_mad_over_non_zero_samples():
    """bpmf_row_median_mad_dev: np.median / MAD of the rows, over all samples and over the non-zero
    ones (the statistics of saturated_envelopes, BPMF/template_search.py:1547-1561)."""
    import torch
    from seismic_bpmf_amd import features
    rng = np.random.default_rng(23)
    for n in (1, 2, 3, 1000, 1001, 70_000):
        x = np.abs(rng.standard_normal((7, n))).astype(np.float32)
        x[1, rng.random(n) < 0.4] = 0.0
        x[2] = 0.0
        x[3] = np.round(x[3] * 4) / 4
        if n > 2:
            x[4, 1] = -0.0
            x[5, 0] = np.nan
        for skip in (False, True):
            med, mad, nz = (t.cpu().numpy() for t in features.row_median_mad(torch.as_tensor(x, device="cuda"), skip))
            for r in range(7):
                v = x[r][x[r] != 0] if skip else x[r]
                with np.errstate(all="ignore"), __import__("warnings").catch_warnings():
                    __import__("warnings").simplefilter("ignore")
                    m = np.median(v) if v.size else np.float32(np.nan)
                    d = np.median(np.abs(v - m)) if v.size else np.float32(np.nan)
                assert np.array_equal(med[r], np.float32(m), equal_nan=True), (n, skip, r)
                assert np.array_equal(mad[r], np.float32(d), equal_nan=True), (n, skip, r)
                assert nz[r] == (x[r] == 0).sum()


@pytest.mark.parametrize("window,overlap", [(4000, 0.25), (4001, 0.5), (257, 0.0), (30_000, 0.66), (13_001, 0.3), (60_000, 0.5)])
def test_mad_bucketed_medians_equal_the_radix_select_on_awkward_rows(hip_opts, window, overlap):
    """Round 5: the window medians of the MAD threshold take two passes (equal-width buckets around the row's own
    centre, the middle bucket ranked in LDS: select.h window_median_bucketed) with the three-pass radix select as
    the fallback.  Both are exact order statistics, so option stats.bucketed_median must not change a bit -- on
    Gaussian rows (the buckets work), and on rows where the guess is worth nothing: constant, two-valued, heavy
    tails, a drift far from the row's centre, blocks of exact zeros (replaced by white noise where they are read),
    an Inf, a NaN, windows of even and odd length; against the host mirror of the reference as well."""
    import torch
    from seismic_bpmf_amd import postprocess as pp
    from seismic_bpmf_amd.threshold import ThresholdGPU
    rng = np.random.default_rng(window)
    n = 90_000
    rows = [rng.standard_normal(n) * 0.05,
            np.full(n, 0.25),
            rng.choice([-0.5, 0.75], n),
            rng.standard_cauchy(n) * 0.01,
            rng.standard_normal(n) * 0.05 + np.linspace(-3.0, 3.0, n),
            np.round(rng.standard_normal(n) * 4) / 64,                       # many duplicates
            rng.standard_normal(n) * 1e-30,
            rng.standard_normal(n) * 0.05, rng.standard_normal(n) * 0.05, rng.standard_normal(n) * 0.05]
    x = np.stack(rows).astype(np.float32)
    x[0, :1500] = 0.0
    x[0, -2500:] = 0.0
    x[5, 30_000:31_000] = 0.0
    x[7, 40_000] = np.inf
    x[8, 50_000] = np.nan
    x[9, 10_000:10_000 + window // 2 + 5] = np.inf                          # more than half a window of +Inf
    wn = rng.standard_normal(n).astype(np.float32)
    th = ThresholdGPU(device=0)
    xd = torch.as_tensor(x, device="cuda")
    outs = {}
    for flag in (2, 1, 0):
        hip_opts("stats.bucketed_median", flag)
        tw, full = th.time_dependent_threshold_mad(xd, window, 8.0, overlap=overlap, white_noise=wn, expand=True)
        outs[flag] = (tw.cpu().numpy(), full.cpu().numpy())
    for flag in (2, 1):
        assert np.array_equal(outs[flag][0], outs[0][0], equal_nan=True), flag
        assert np.array_equal(outs[flag][1], outs[0][1], equal_nan=True), flag
    for r in (0, 2, 3, 4, 5):                                                # the host mirror of similarity_search.py:1079-1113
        want = pp.time_dependent_threshold_mad(x[r], window, 8.0, overlap=overlap, white_noise=wn)
        assert np.array_equal(outs[1][1][r], want.astype(np.float32), equal_nan=True), r


@pytest.mark.parametrize("n", [5, 4096, 70_001, 300_000, 1_000_003])
def test_row_median_mad_two_read_path_equals_the_one_workgroup_path(hip_opts, n):
    """Round 5: rows of at least stats.row_grid_min_n samples are read twice by workgroups from all over the chip
    (a histogram around a sampled guess, then the middle bucket and the two bands that can hold a middle deviation;
    every step verified by ranks, csrc/stats.hip rm_*), any row that fails a check goes to the one-workgroup radix
    select.  Same bits as that kernel alone (option -1) and as NumPy -- on rows where the guess works and on rows
    where it cannot: constant, two-valued, heavy tails, a drift, duplicates, denormal scale, zeros to skip, half
    of the row zero, +-Inf, a NaN, an offset a thousand deviations from zero, a sample that misleads (a burst
    exactly where the sample looks)."""
    import warnings
    import torch
    from seismic_bpmf_amd import features
    rng = np.random.default_rng(n)
    rows = [rng.standard_normal(n) * 0.05,
            np.abs(rng.standard_normal(n)),
            np.full(n, 0.25),
            rng.choice([-0.5, 0.75], n),
            rng.standard_cauchy(n) * 0.01,
            rng.standard_normal(n) * 0.05 + np.linspace(-3.0, 3.0, n),
            np.round(rng.standard_normal(n) * 4) / 64,
            rng.standard_normal(n) * 1e-30,
            rng.standard_normal(n) * 0.05,
            rng.standard_normal(n) * 0.05,
            rng.standard_normal(n) * 0.05,
            rng.standard_normal(n) * 0.001 + 1000.0,
            rng.lognormal(0.0, 2.0, n),
            rng.standard_normal(n) * 0.05,
            np.zeros(n),
            rng.standard_normal(n) * 0.05]
    x = np.stack(rows).astype(np.float32)
    x[1, rng.random(n) < 0.55] = 0.0
    x[8, n // 3] = np.inf
    x[8, n // 2] = -np.inf
    x[9, n // 2] = np.nan
    x[10, : n // 2] = 0.0
    stride = max(1, n // 4096)
    x[13, ::stride] = 50.0 + rng.standard_normal(len(x[13, ::stride])).astype(np.float32)   # the sample sees only the burst
    x[15, -1] = -0.0
    xd = torch.as_tensor(x, device="cuda")
    for skip in (False, True):
        got = {}
        for min_n in (0, -1):
            hip_opts("stats.row_grid_min_n", min_n)
            got[min_n] = [t.cpu().numpy() for t in features.row_median_mad(xd, skip)]
        for a, b in zip(got[0], got[-1]):
            assert np.array_equal(a, b, equal_nan=True), (n, skip)
        med, mad, nz = got[0]
        for r in range(x.shape[0]):
            v = x[r][x[r] != 0] if skip else x[r]
            with np.errstate(all="ignore"), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m = np.median(v) if v.size else np.float32(np.nan)
                d = np.median(np.abs(v - m)) if v.size else np.float32(np.nan)
            assert np.array_equal(med[r], np.float32(m), equal_nan=True), (n, skip, r)
            assert np.array_equal(mad[r], np.float32(d), equal_nan=True), (n, skip, r)
            assert nz[r] == (x[r] == 0).sum()


def test_row_median_mad_two_read_path_when_the_middle_ranks_straddle_a_bucket_edge(hip_opts):
    """The two-read path histograms a row over 4096 buckets laid around a sampled guess; the two middle ranks of an
    even count land in DIFFERENT buckets once in a few hundred rows of this length (once in 10 000 day-long rows: row
    65 of the bench's CC matrix did, and the one-workgroup kernel it was sent to cost 12 ms).  Four seeds where a
    host restatement of the guess (strided sample, float32 median / MAD, the kernel's bucket expression) says they
    straddle: the plan takes the adjacent buckets together, the results are NumPy's."""
    import torch
    from seismic_bpmf_amd import features
    f = np.float32
    rows = []
    for seed in (64, 278, 358, 738):
        x = (np.random.default_rng(900_000 + seed).standard_normal(200_000) * 0.05).astype(f)
        n = x.size
        smp = x[(np.arange(4096, dtype=np.uint64) * np.uint64(n) // np.uint64(4096)).astype(np.int64)]
        m = np.median(smp).astype(f)
        sigma = f(1.4826) * np.median(np.abs(smp - m)).astype(f)
        lo, inv_w = f(m - f(6.0) * sigma), f(f(4096) / (f(12.0) * sigma))
        s = np.sort(x)
        b_lo, b_hi = (int(f(f(v - lo) * inv_w)) for v in (s[(n - 1) // 2], s[n // 2]))
        assert b_hi == b_lo + 1, (seed, b_lo, b_hi)          # (the premise of the test)
        rows.append(x)
    x = np.stack(rows)
    hip_opts("stats.row_grid_min_n", 0)
    med, mad, nz = (t.cpu().numpy() for t in features.row_median_mad(torch.as_tensor(x, device="cuda"), True))
    for r in range(x.shape[0]):
        m = np.median(x[r])
        assert med[r] == m and mad[r] == np.median(np.abs(x[r] - m)) and nz[r] == 0, r
