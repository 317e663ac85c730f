"""GPU: seeded random sweeps of the device stages around the two hot paths -- BP detection
(window statistics, peak extraction, suppression), the RMS threshold + candidate extraction behind
the matched filter, the batched inter-template CC -- against their host mirrors / the oracle, which
tests/test_postprocess.py and tests/test_oracle_pinned.py pin to the reference's goldens.  Bit-exact.
BPMF_FUZZ_SEEDS=a:b widens the sweep (tools/fuzz_long.sh A B secs fuzz_workflow)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fuzz_seeds(default):
    spec = os.environ.get("BPMF_FUZZ_SEEDS")
    if not spec:
        return range(default)
    a, b = spec.split(":")
    return range(int(a), int(b))


@pytest.mark.filterwarnings("ignore:Mean of empty slice", "ignore:invalid value encountered")
@pytest.mark.parametrize("seed", _fuzz_seeds(16))
def test_fuzz_workflow_bp_detections(seed):
    import torch
    from seismic_bpmf_amd import postprocess as pp
    from seismic_bpmf_amd.workflow import beam_detections_device
    rng = np.random.default_rng(31_000 + seed)
    n = int(rng.choice([3, 50, 999, 5_000, 20_001, 65_536, 150_000]))
    window = int(rng.choice([w for w in (2, 7, 64, 500, 2_049, 4_096, 30_000) if w <= n] or [n]))
    overlap = float(rng.choice([0.0, 0.5, 0.75, 0.9]))
    if int((1.0 - overlap) * window) < 1:
        overlap = 0.0
    mpd = int(rng.choice([1, 2, 5, 40, 333, 5_000]))
    n_dev = float(rng.choice([0.0, 3.0, 8.0, 15.0]))
    x = np.abs(rng.standard_normal(n)).astype(np.float32)
    kind = seed % 4
    if kind == 0:
        x = np.round(x, 1)                                   # plateaus and exact ties
    elif kind == 1:
        x[n // 3:] *= np.float32(5.0)                        # the threshold steps up
    elif kind == 2:
        x = (x - np.float32(1.0)).astype(np.float32)         # negative values
    for p in rng.integers(0, n, max(1, n // 700)):
        x[p] += np.float32(rng.uniform(3, 40))
    src = rng.integers(0, 1_000_000, n).astype(np.int32)
    xd, sd = torch.as_tensor(x, device="cuda"), torch.as_tensor(src, device="cuda")
    what = f"seed {seed} n={n} window={window} overlap={overlap} mpd={mpd} n_dev={n_dev}"
    want_thr = pp.bp_time_dependent_threshold(x, window, n_dev, overlap=overlap)
    want_p, want_s = pp.find_beam_detections(x, src, want_thr, mpd)
    peaks, psrc, _ = beam_detections_device(xd, sd, mpd=mpd, window=window, n_dev=n_dev, overlap=overlap)
    assert np.array_equal(peaks, want_p) and np.array_equal(psrc, want_s), what
    c = float(np.quantile(x, 0.98))
    want_p, want_s = pp.find_beam_detections(x, src, c, mpd)
    peaks, psrc, _ = beam_detections_device(xd, sd, mpd=mpd, threshold=c)
    assert np.array_equal(peaks, want_p) and np.array_equal(psrc, want_s), what + " constant threshold"


@pytest.mark.parametrize("seed", _fuzz_seeds(12))
def test_fuzz_workflow_rms_threshold_and_candidates(oracle_lib, seed):
    import torch
    from seismic_bpmf_amd.threshold import ThresholdGPU
    rng = np.random.default_rng(32_000 + seed)
    rows = int(rng.integers(1, 6))
    n = int(rng.choice([600, 5_000, 20_011, 70_000]))
    window = int(rng.choice([w for w in (100, 512, 3_000, 10_000) if 2 * w <= n]))
    overlap = float(rng.choice([0.0, 0.25]))
    num_dev = float(rng.choice([3.0, 8.0]))
    x = (0.05 * rng.standard_normal((rows, n))).astype(np.float32)
    if seed % 3 == 0:
        a = int(rng.integers(0, n // 2))
        x[0, a:a + int(rng.integers(1, n // 3))] = 0.0       # exact zeros: replaced by scaled white noise
    for r in range(rows):
        for p in rng.integers(0, n, 6):
            x[r, p] += np.float32(rng.uniform(0.3, 0.9))
    wn = rng.standard_normal(500).astype(np.float32)
    th = ThresholdGPU()
    xd = torch.as_tensor(x, device="cuda")
    thr_win, full = th.time_dependent_threshold(xd, window, num_dev, overlap=overlap, white_noise=wn, expand=True)
    full = full.cpu().numpy()
    for r in range(rows):
        want = oracle_lib.time_dependent_threshold(x[r], window, num_dev, overlap, wn)
        assert np.array_equal(full[r], want), f"seed {seed} row {r} n={n} window={window} overlap={overlap}"
    cand = th.extract_candidates(xd, thr_win, window, overlap=overlap)
    rr, ii = np.nonzero(x > full)
    assert np.array_equal(cand["row"], rr) and np.array_equal(cand["index"], ii), f"seed {seed}"
    assert np.array_equal(cand["cc"], x[rr, ii])


@pytest.mark.parametrize("seed", _fuzz_seeds(12))
def test_fuzz_workflow_intertemplate_cc(oracle_lib, seed):
    from seismic_bpmf_amd import workflow
    rng = np.random.default_rng(33_000 + seed)
    T = int(rng.integers(1, 40))
    S = int(rng.integers(1, 9))
    C = int(rng.integers(1, 4))
    L = int(rng.choice([1, 2, 9, 40, 100, 257, 1025, 1300]))
    max_lag = int(rng.choice([0, 1, 5, 10, 33]))         # 33: more lags than the batched kernel holds -> the loop
    if L <= 2 * max_lag:
        with pytest.raises(ValueError):
            workflow.intertemplate_cc(np.zeros((2, 1, 1, L), np.float32), np.ones((2, 1, 1), np.float32), max_lag=max_lag)
        max_lag = (L - 1) // 2
    wf = rng.standard_normal((T, S, C, L)).astype(np.float32)
    if T > 2:
        wf[2, 0, 0] = 0.0
    base = (rng.random((T, S, C)) > 0.3).astype(np.float32) * rng.random((T, S, C)).astype(np.float32)
    if seed % 3 == 0:
        base = (base > 0).astype(np.float32)
        base /= np.maximum(base.sum(axis=(1, 2), keepdims=True), 1.0)
    mask = rng.random((T, T)) > rng.random()
    got = workflow.intertemplate_cc(wf, base, max_lag=max_lag, pair_mask=mask)
    full = base[:, None, :, :] * mask[:, :, None, None]
    want = workflow.intertemplate_cc_loop(wf, full, max_lag=max_lag)
    assert np.array_equal(got, want), f"seed {seed} T={T} S={S} C={C} L={L} max_lag={max_lag}"


@pytest.mark.parametrize("seed", _fuzz_seeds(8))
def test_fuzz_workflow_matched_filter_detections(oracle_lib, seed):
    """workflow.matched_filter_detections (CC, RMS threshold, candidate compaction on the device, merge
    on the host) == the same chain from CPU pieces: oracle CC, oracle threshold (pinned to the
    reference's libc.c), postprocess.select_cc_indexes (pinned to the reference's Python)."""
    from scipy.stats import kurtosis
    from seismic_bpmf_amd import postprocess as pp, synthetic as syn, workflow
    rng = np.random.default_rng(34_000 + seed)
    T, S, C = int(rng.integers(1, 5)), int(rng.integers(2, 6)), int(rng.integers(1, 4))
    L = int(rng.choice([32, 64, 200]))
    N = int(rng.choice([40_000, 90_001]))
    step = int(rng.choice([1, 1, 2]))
    sr = 100.0
    m = syn.make_mf_inputs(T=T, S=S, C=C, L=L, N=N, seed=int(rng.integers(1 << 30)),
                           max_moveout=int(rng.choice([50, 400])), n_events=int(rng.integers(1, 6)), step=step)
    window_dur = float(rng.choice([20.0, 75.0]))
    min_iet = float(rng.choice([0.5, 3.0]))
    overlap = float(rng.choice([0.0, 0.25]))
    n_dev = float(rng.choice([6.0, 8.0]))
    wn = rng.standard_normal(500).astype(np.float32)
    got, cc = workflow.matched_filter_detections(m["templates"], m["moveouts"], m["weights"], m["data"], step=step,
                                                 sr=sr, threshold_window_dur=window_dur,
                                                 minimum_interevent_time=min_iet, n_dev=n_dev, overlap=overlap,
                                                 white_noise=wn, remove_edges=False)
    cc_ref = oracle_lib.matched_filter(m["templates"], m["moveouts"], m["weights"], m["data"], step)
    assert np.array_equal(cc.cpu().numpy(), cc_ref)
    window = int(pp.sec_to_samp(window_dur, sr))
    w = np.asarray(m["weights"], np.float32)
    mv = np.asarray(m["moveouts"])
    for t in range(T):
        thr = oracle_lib.time_dependent_threshold(cc_ref[t], window, n_dev, overlap, wn)
        thr = np.minimum(thr, np.float32((0.80 * w.reshape(T, -1).sum(axis=1))[t]))
        win = workflow.search_window(mv[t].reshape(mv.shape[1], -1), int(pp.sec_to_samp(min_iet, sr)), step)
        want = pp.select_cc_indexes(cc_ref[t], thr, win, step=step, sr=sr, data_duration_sec=1e9,
                                    n_dev_threshold=n_dev, min_freq_hz=2.0, data_buffer_sec=0.0,
                                    remove_edges=False, anomalous_cdf_at_mean_plus_1sig=0.0)
        if kurtosis(cc_ref[t].astype(np.float64)) > 100.0:        # the reference's sanity check (:633-642)
            want = np.zeros(0, dtype=np.int64)
        assert np.array_equal(got[t], want), f"seed {seed} template {t}: T={T} S={S} C={C} L={L} N={N} step={step}"


@pytest.mark.parametrize("seed", _fuzz_seeds(10))
def test_fuzz_cc_detections_with_the_anomalous_cdf_validation(oracle_lib, seed):
    """Round 6: the optional validation of MatchedFilter.select_cc_indexes
    (BPMF/similarity_search.py:253-272) inside the device pipeline -- workflow.cc_detections(...,
    anomalous_cdf_at_mean_plus_1sig=, window_for_validation_Tmax=, min_freq_hz=), the counts by
    bpmf_count_below_dev -- against postprocess.select_cc_indexes (pinned to the reference's Python, golden
    select_cc_indexes_py.npz holds validation cases) on the same CC rows and thresholds.  CC rows with stretches
    where the threshold fails (a block of large values, as a gap in the data makes them), detections near both
    ends of the series (the clipped windows), both threshold types, edge removal behind the validation."""
    import torch
    from seismic_bpmf_amd import postprocess as pp, workflow
    rng = np.random.default_rng(36_000 + seed)
    T, n = int(rng.integers(1, 6)), int(rng.choice([30_000, 61_234]))
    S, C = int(rng.integers(2, 5)), int(rng.integers(1, 4))
    sr, step = 100.0, int(rng.choice([1, 1, 2]))
    cc_h = (0.08 * rng.standard_normal((T, n))).astype(np.float32)
    for t in range(T):
        for _ in range(int(rng.integers(2, 9))):                    # isolated detections
            cc_h[t, int(rng.integers(0, n))] = rng.uniform(0.5, 0.95)
        for _ in range(int(rng.integers(0, 4))):                    # stretches where the noise level is 5 x higher
            a = int(rng.integers(0, n - 400))
            b = a + int(rng.integers(50, 400))
            cc_h[t, a:b] *= 5.0
            cc_h[t, int(rng.integers(a, b))] = rng.uniform(0.6, 0.9)
        cc_h[t, int(rng.integers(0, 20))] = 0.9                     # the clipped windows at both ends
        cc_h[t, n - 1 - int(rng.integers(0, 20))] = 0.9
    mv = rng.integers(0, 300, (T, S, C)).astype(np.int32)
    w = np.full((T, S, C), 1.0 / (S * C), np.float32)
    window_dur, min_iet = float(rng.choice([20.0, 60.0])), float(rng.choice([0.5, 2.0]))
    overlap, n_dev = float(rng.choice([0.0, 0.25])), float(rng.choice([6.0, 8.0, 7.3]))
    min_freq, tmax = float(rng.choice([0.5, 2.0, 4.0])), float(rng.choice([100.0, 250.0, 37.0]))
    cut = float(rng.choice([0.5, 0.7, 0.3]))
    kind = str(rng.choice(["rms", "mad"]))
    wn = rng.standard_normal(500 if kind == "rms" else n).astype(np.float32)
    edges = bool(rng.integers(0, 2))
    kw = dict(remove_edges=edges, data_buffer_sec=5.0 if edges else None, data_duration_sec=n * step / sr - 10.0 if edges else None)
    cc = torch.as_tensor(cc_h, device="cuda")
    got = workflow.cc_detections(cc, mv, w, step=step, sr=sr, threshold_window_dur=window_dur, minimum_interevent_time=min_iet,
                                 n_dev=n_dev, overlap=overlap, white_noise=wn, sanity_check=False, threshold_type=kind,
                                 anomalous_cdf_at_mean_plus_1sig=cut, window_for_validation_Tmax=tmax, min_freq_hz=min_freq, **kw)
    plain = workflow.cc_detections(cc, mv, w, step=step, sr=sr, threshold_window_dur=window_dur, minimum_interevent_time=min_iet,
                                   n_dev=n_dev, overlap=overlap, white_noise=wn, sanity_check=False, threshold_type=kind, **kw)
    window = int(pp.sec_to_samp(window_dur, sr))
    dropped = 0
    for t in range(T):
        if kind == "rms":
            thr = oracle_lib.time_dependent_threshold(cc_h[t], window, n_dev, overlap, wn)
        else:
            thr = pp.time_dependent_threshold_mad(cc_h[t], window, n_dev, overlap=overlap, white_noise=wn).astype(np.float32)
        thr = np.minimum(thr, np.float32(0.80 * w[t].sum()))
        win = workflow.search_window(mv[t].reshape(S, -1), int(pp.sec_to_samp(min_iet, sr)), step)
        want = pp.select_cc_indexes(cc_h[t], thr, win, step=step, sr=sr, data_duration_sec=(n * step / sr - 10.0) if edges else 1e9,
                                    n_dev_threshold=n_dev, min_freq_hz=min_freq, data_buffer_sec=5.0 if edges else 0.0,
                                    remove_edges=edges, threshold_type=kind, anomalous_cdf_at_mean_plus_1sig=cut,
                                    window_for_validation_Tmax=tmax)
        assert np.array_equal(got[t], want), f"seed {seed} row {t}: n={n} step={step} {kind} cut={cut} win={int(tmax / min_freq)}"
        dropped += len(plain[t]) - len(got[t])
    print(f"seed {seed}: the validation dropped {dropped} detections")
    with pytest.raises(ValueError, match="min_freq_hz"):
        workflow.cc_detections(cc, mv, w, step=step, sr=sr, threshold_window_dur=window_dur, minimum_interevent_time=min_iet,
                               anomalous_cdf_at_mean_plus_1sig=0.5, remove_edges=False)


def _fuzz_row(rng, n):
    """One row of a distribution the guesses of the order-statistics kernels may or may not cope with."""
    kind = int(rng.integers(0, 12))
    scale = np.float32(10.0 ** rng.uniform(-6, 3))
    x = rng.standard_normal(n)
    if kind == 1:
        x = rng.standard_cauchy(n)
    elif kind == 2:
        x = np.round(x * rng.choice([2, 16, 1000]))                         # ties, from heavy to light
    elif kind == 3:
        x = np.abs(x) ** rng.uniform(0.3, 4.0)                              # skewed
    elif kind == 4:
        x = x + np.where(rng.random(n) < rng.uniform(0.05, 0.6), rng.uniform(3, 300), 0.0)   # two modes
    elif kind == 5:
        x = x * np.linspace(0.01, rng.uniform(1, 50), n)                    # the noise level drifts
    elif kind == 6:
        x = x + np.linspace(-1, 1, n) * rng.uniform(0, 30)                  # the centre drifts
    elif kind == 7:
        x = np.full(n, rng.standard_normal())                               # constant
        x[rng.integers(0, n, max(1, n // 50))] += rng.standard_normal()
    elif kind == 8:
        x = x * 1e-3 + rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(0, 4)  # far from zero
    elif kind == 9:
        x = rng.lognormal(0.0, rng.uniform(0.5, 3.0), n)
    elif kind == 10:
        x = rng.choice(rng.standard_normal(int(rng.integers(1, 9))), n)     # a handful of values
    x = (x * scale).astype(np.float32)
    if rng.random() < 0.5:
        x[rng.random(n) < rng.choice([0.001, 0.1, 0.5, 0.9])] = 0.0
    if rng.random() < 0.3 and n > 10:
        k = int(rng.integers(1, max(2, n // 20)))
        x[:k] = 0.0
        x[-k:] = 0.0
    if rng.random() < 0.1:
        x[rng.integers(0, n)] = np.float32(rng.choice([np.inf, -np.inf]))
    if rng.random() < 0.05:
        x[rng.integers(0, n)] = np.nan
    return x


@pytest.mark.filterwarnings("ignore::RuntimeWarning")
@pytest.mark.parametrize("seed", _fuzz_seeds(12))
def test_fuzz_workflow_row_statistics(hip_opts, seed):
    """Row median / MAD (the two-read path forced for every length, and the one-workgroup path) and the row
    kurtosis against NumPy / SciPy's float32 arithmetic on rows of a dozen distributions."""
    import torch
    from seismic_bpmf_amd import features, postprocess as pp, workflow
    rng = np.random.default_rng(77_000 + seed)
    n = int(rng.choice([1, 2, 17, 1_000, 4_097, 8_192, 65_537, 200_000, 700_001]))
    rows = int(rng.integers(1, 9))
    x = np.stack([_fuzz_row(rng, n) for _ in range(rows)])
    xd = torch.as_tensor(x, device="cuda")
    for skip in (False, True):
        want_m, want_d = [], []
        for r in range(rows):
            v = x[r][x[r] != 0] if skip else x[r]
            with np.errstate(all="ignore"):
                m = np.median(v) if v.size else np.float32(np.nan)
                d = np.median(np.abs(v - m)) if v.size else np.float32(np.nan)
            want_m.append(np.float32(m))
            want_d.append(np.float32(d))
        for min_n in (0, -1):
            hip_opts("stats.row_grid_min_n", min_n)
            med, mad, nz = (t.cpu().numpy() for t in features.row_median_mad(xd, skip))
            what = f"seed {seed} n={n} rows={rows} skip={skip} min_n={min_n}"
            assert np.array_equal(med, np.array(want_m), equal_nan=True), what
            assert np.array_equal(mad, np.array(want_d), equal_nan=True), what
            assert np.array_equal(nz, (x == 0).sum(axis=1)), what
    with np.errstate(all="ignore"):
        want_k = np.array([pp.excess_kurtosis_f32(r) for r in x], dtype=np.float32)
    for full in (1, 0):
        hip_opts("stats.kurt_full_chunks", full)
        assert np.array_equal(workflow.row_excess_kurtosis(xd), want_k, equal_nan=True), f"seed {seed} n={n} full={full}"


@pytest.mark.filterwarnings("ignore::RuntimeWarning")
@pytest.mark.parametrize("seed", _fuzz_seeds(12))
def test_fuzz_workflow_mad_threshold(hip_opts, seed):
    """The MAD threshold of similarity_search.py:1079-1113 on the device, every route of its medians (one pass,
    two passes, radix select) and of its row statistics, against the host mirror pinned to the reference."""
    import torch
    from seismic_bpmf_amd import postprocess as pp
    from seismic_bpmf_amd.threshold import ThresholdGPU
    rng = np.random.default_rng(78_000 + seed)
    n = int(rng.choice([600, 5_000, 40_000, 150_000]))
    window = int(rng.choice([w for w in (50, 257, 4_000, 13_001, 30_000, 60_000) if w <= n]))
    overlap = float(rng.choice([0.0, 0.25, 0.5, 0.66]))
    rows = int(rng.integers(1, 6))
    x = np.stack([_fuzz_row(rng, n) for _ in range(rows)])
    x[~np.isfinite(x)] = 0.0                      # (a CC series holds no Inf; NaN rows are covered by the variants test)
    wn = rng.standard_normal(n).astype(np.float32)
    th = ThresholdGPU(device=0)
    xd = torch.as_tensor(x, device="cuda")
    with np.errstate(all="ignore"):
        want = np.stack([pp.time_dependent_threshold_mad(r, window, 8.0, overlap=overlap, white_noise=wn) for r in x]).astype(np.float32)
    for mode, min_n in ((2, 0), (1, -1), (0, 0)):
        hip_opts("stats.bucketed_median", mode)
        hip_opts("stats.row_grid_min_n", min_n)
        _, full = th.time_dependent_threshold_mad(xd, window, 8.0, overlap=overlap, white_noise=wn, expand=True)
        assert np.array_equal(full.cpu().numpy(), want, equal_nan=True), \
            f"seed {seed} n={n} window={window} overlap={overlap} rows={rows} mode={mode} min_n={min_n}"
