"""GPU: the two detection pipelines and the inter-template CC, end to end."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_matched_filter_detections_recover_planted_events():
    from seismic_bpmf_amd import synthetic as syn
    from seismic_bpmf_amd.workflow import matched_filter_detections
    inp = syn.make_mf_inputs(T=4, S=6, C=3, L=64, N=150_000, seed=3, max_moveout=200, n_events=4)
    det, cc = matched_filter_detections(inp["templates"], inp["moveouts"], inp["weights"], inp["data"],
                                        sr=100.0, threshold_window_dur=300.0, minimum_interevent_time=5.0,
                                        remove_edges=False,
                                        white_noise=np.random.default_rng(0).standard_normal(500).astype(np.float32))
    with pytest.raises(ValueError, match="data_buffer_sec"):      # the reference always trims its buffer: no silent skip
        matched_filter_detections(inp["templates"], inp["moveouts"], inp["weights"], inp["data"], sr=100.0,
                                  threshold_window_dur=300.0, minimum_interevent_time=5.0)
    # the MAD threshold finds the same planted events
    det_mad, _ = matched_filter_detections(inp["templates"], inp["moveouts"], inp["weights"], inp["data"],
                                           sr=100.0, threshold_window_dur=300.0, minimum_interevent_time=5.0,
                                           remove_edges=False, threshold_type="mad",
                                           white_noise=np.random.default_rng(0).standard_normal(5000).astype(np.float32))
    for t in range(4):
        assert set(i0 for tt, i0 in inp["planted"] if tt == t) <= set(det_mad[t].tolist()), t
    for t in range(4):
        planted = sorted(i0 for tt, i0 in inp["planted"] if tt == t)
        assert list(det[t]) == planted, (t, det[t], planted)
    assert cc.shape == (4, 150_000 - 64 + 1)


def test_backprojection_detections_recover_planted_sources(oracle_lib):
    from seismic_bpmf_amd import synthetic as syn
    from seismic_bpmf_amd.workflow import backprojection_detections
    geo = syn.make_bp_geometry((8, 8, 4), S=10, P=2, sr=25.0, extent_km=(40.0, 40.0, 12.0), n_closest=6)
    feat, planted = syn.make_bp_features(geo["moveouts"], 10, 3, 60_000, sr=25.0, n_events=6, amp=10.0)
    wp = syn.phase_weights(10, 3, 2)
    peaks, src, maxbeam, arg = backprojection_detections(
        feat, geo["moveouts"], wp, geo["weights_sources"], sr=25.0, minimum_interevent_time=20.0,
        threshold_window_dur=400.0, n_dev=15.0)
    ob, oa = oracle_lib.beamform(feat, geo["moveouts"], wp, geo["weights_sources"], "strict", "max")
    assert np.array_equal(maxbeam, ob) and np.array_equal(arg, oa)
    for k0, t0 in planted:
        hit = np.flatnonzero(np.abs(peaks - t0) <= 5)
        assert hit.size == 1, (k0, t0, peaks)
        # the located source is the planted one or a grid neighbour with an equal/greater beam
        assert maxbeam[peaks[hit[0]]] >= ob[t0] - 1e-6


def test_intertemplate_cc_against_oracle(oracle_lib):
    from seismic_bpmf_amd.workflow import intertemplate_cc
    rng = np.random.default_rng(5)
    T, S, C, L, max_lag = 6, 4, 3, 80, 7
    wf = rng.standard_normal((T, S, C, L)).astype(np.float32)
    wf[3] = np.roll(wf[1], 3, axis=-1)              # template 3 = template 1 shifted by 3 samples
    w = np.zeros((T, T, S, C), dtype=np.float32)
    w[:, :, :3, :] = 1.0 / 9.0
    w[2, 4] = 0.0                                    # pair beyond the distance threshold
    got = intertemplate_cc(wf, w, max_lag=max_lag)
    want = np.zeros((T, T), dtype=np.float32)
    trimmed = np.ascontiguousarray(wf[..., max_lag:-max_lag])
    for t in range(T):
        keep = np.flatnonzero((w[t] != 0).reshape(T, -1).sum(axis=1) > 0)
        cc = oracle_lib.matched_filter(trimmed[keep], np.zeros((keep.size, S, C), np.int32), w[t][keep], wf[t], 1,
                                       network_sum=False)
        want[t, keep] = np.sum(w[t][keep] * cc.max(axis=1), axis=(-1, -2))
    want = (want + want.T) / 2.0
    assert np.array_equal(got, want)
    assert got[1, 3] > 0.85 and abs(got[0, 0] - 1.0) < 1e-5


def test_bp_threshold_on_device_equals_the_host_mirror():
    """Sliding median/MAD threshold of the max beam: window statistics by radix select on the device
    (BeamDetectorGPU.window_stats) + the node / interpolation logic of the host mirror, against the
    reference's golden (tests/golden/bp_threshold.npz), bit for bit, including windows cut short at the
    end of the trace and even/odd window lengths."""
    import torch
    from seismic_bpmf_amd import postprocess as pp
    from seismic_bpmf_amd.threshold import BeamDetectorGPU
    import os
    det = BeamDetectorGPU()

    def device_threshold(x, window, n_dev, overlap):
        med, mad = det.window_stats(torch.as_tensor(x, device="cuda"), window, overlap)
        centre, thr = pp.bp_threshold_nodes(len(x), window, overlap, med, mad, n_dev)
        return pp.interp_threshold(np.arange(len(x), dtype=np.float64), centre, thr)

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bp_threshold.npz"))
    got = device_threshold(g["maxbeam"], int(g["window"]), float(g["n_dev"]), float(g["overlap"]))
    assert np.array_equal(got, g["thr"])          # interpolation in SciPy's own operation order: the golden itself
    rng = np.random.default_rng(8)
    for n, window, overlap in [(50_001, 3001, 0.75), (40_000, 3000, 0.5), (12_345, 1000, 0.9), (9_000, 9_000, 0.75)]:
        x = (np.abs(rng.standard_normal(n)) * (1 + 5 * (rng.random(n) > 0.999))).astype(np.float32)
        want = pp.bp_time_dependent_threshold(x, window, 15.0, overlap=overlap)
        assert np.array_equal(device_threshold(x, window, 15.0, overlap), want), (n, window, overlap)


# ------------------------------------------------------- BP detection stage on the device ---
def _golden(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))


def test_bp_window_stats_radix_select_equals_np_median():
    """Per-window median / MAD by radix select (csrc/bp_detect.hip) == np.median in float32, bit
    for bit: the reference's golden series, even / odd window lengths, windows cut short by the end
    of the trace, heavy ties (rounded values), negative values, zeros of both signs."""
    import torch
    from seismic_bpmf_amd import postprocess as pp
    from seismic_bpmf_amd.threshold import BeamDetectorGPU
    det = BeamDetectorGPU()
    g = _golden("bp_threshold.npz")
    cases = [(g["maxbeam"], int(g["window"]), float(g["overlap"]))]
    rng = np.random.default_rng(8)
    for n, window, overlap in [(50_001, 3001, 0.75), (40_000, 3000, 0.5), (12_345, 1000, 0.9),
                               (9_000, 9_000, 0.75), (400_000, 90_000, 0.75), (2_100, 2_049, 0.98)]:
        x = (np.abs(rng.standard_normal(n)) * (1 + 5 * (rng.random(n) > 0.999))).astype(np.float32)
        cases.append((x, window, overlap))
    x = np.round(rng.standard_normal(30_000), 1).astype(np.float32)           # ties, negatives, +-0
    x[::7] = -0.0
    cases.append((x, 4_000, 0.75))
    for x, window, overlap in cases:
        n = x.size
        shift = int((1.0 - overlap) * window)
        nw = int((n - window) // shift) + 1
        med, mad = det.window_stats(torch.as_tensor(x, device="cuda"), window, overlap)
        assert med.shape == (nw + 2,)
        for q in range(1, nw + 1):
            seg = x[q * shift:min(n, q * shift + window)]
            m = np.median(seg)
            assert med[q] == m and mad[q] == np.median(np.abs(seg - m)), (n, window, q)
        # the node values and the interpolated threshold are then the host mirror's (pinned to the golden)
        centre, thr = pp.bp_threshold_nodes(n, window, overlap, med, mad, 15.0)
        assert np.array_equal(pp.interp_threshold(np.arange(n), centre, thr),
                              pp.bp_time_dependent_threshold(x, window, 15.0, overlap))
    xn = cases[1][0].copy()
    xn[7_000] = np.nan                                                       # np.median: NaN in -> NaN out
    med, mad = det.window_stats(torch.as_tensor(xn, device="cuda"), 3001, 0.75)
    shift = int(0.25 * 3001)
    for q in range(1, med.size - 1):
        assert np.isnan(med[q]) == (q * shift <= 7_000 < q * shift + 3001)


def test_bp_detections_on_device_equal_the_host_mirror():
    """beam_detections_device (nothing of length N leaves the GPU) == postprocess.find_beam_detections
    on the downloaded series: the reference's goldens (constant threshold), and series with a
    time-dependent threshold, plateaus, peaks below the local threshold next to peaks above it."""
    import torch
    from seismic_bpmf_amd import postprocess as pp
    from seismic_bpmf_amd.workflow import beam_detections_device
    g = _golden("bp_find_detections.npz")
    for j in range(int(g["n_cases"])):
        mb, src, thr, mpd = g[f"maxbeam_{j}"], g[f"sources_{j}"], g[f"thr_{j}"], int(g[f"mpd_{j}"])
        peaks, psrc, _ = beam_detections_device(torch.as_tensor(mb, device="cuda"),
                                                torch.as_tensor(src, device="cuda"), mpd=mpd,
                                                threshold=float(thr[0]))
        assert np.array_equal(peaks, g[f"peaks_{j}"]) and np.array_equal(psrc, g[f"peak_sources_{j}"])
    rng = np.random.default_rng(12)
    for n, window, mpd, gain in [(200_000, 20_000, 250, 1.0), (120_000, 9_000, 40, 4.0), (60_000, 6_000, 1, 1.0),
                                 (80_000, 8_000, 501, 2.5)]:
        x = np.abs(rng.standard_normal(n)).astype(np.float32)
        x[n // 2:] *= np.float32(gain)                                       # noisier second half: threshold steps up
        for p in rng.integers(1000, n - 1000, 40):
            x[p - 3:p + 4] += np.float32(rng.uniform(3, 30)) * np.array([.2, .6, .9, 1, .9, .6, .2], np.float32)
        x = np.round(x, 2)                                                   # exact ties and plateaus
        src = rng.integers(0, 50_000, n).astype(np.int32)
        want_thr = pp.bp_time_dependent_threshold(x, window, 8.0, overlap=0.75)
        want_peaks, want_src = pp.find_beam_detections(x, src, want_thr, mpd)
        peaks, psrc, nodes = beam_detections_device(torch.as_tensor(x, device="cuda"),
                                                    torch.as_tensor(src, device="cuda"), mpd=mpd,
                                                    window=window, n_dev=8.0, overlap=0.75)
        assert np.array_equal(peaks, want_peaks) and np.array_equal(psrc, want_src), (n, window, mpd)
        assert want_peaks.size >= 5


def test_intertemplate_cc_batched_equals_the_per_template_loop(oracle_lib):
    """One batched launch over all template pairs (csrc/intertp.hip) == the reference's
    per-template loop (workflow.intertemplate_cc_loop, itself checked against the oracle above),
    bit for bit: T ~ 200 templates, ragged pair masks, dead channels, dead templates, templates
    longer than one prefix-sum chunk."""
    from seismic_bpmf_amd import workflow
    rng = np.random.default_rng(21)
    for T, S, C, L, max_lag in [(200, 7, 3, 120, 10), (37, 45, 3, 96, 5), (12, 3, 2, 1300, 12), (9, 2, 1, 40, 0)]:
        wf = rng.standard_normal((T, S, C, L)).astype(np.float32)
        wf[1] = np.roll(wf[0], 2, axis=-1)
        wf[2, 1, 0] = 0.0                                   # dead template channel -> CC 0 there
        base = (rng.random((T, S, C)) > 0.3).astype(np.float32)
        base[:, 0, :] = 1.0
        base[3] = 0.0                                       # a template without any weighted channel
        base /= np.maximum(base.sum(axis=(1, 2), keepdims=True), 1.0)
        mask = rng.random((T, T)) > 0.4
        mask[np.arange(T), np.arange(T)] = True
        mask[4] = False                                     # a template with no partner at all
        got = workflow.intertemplate_cc(wf, base, max_lag=max_lag, pair_mask=mask)
        full = base[:, None, :, :] * mask[:, :, None, None]
        want = workflow.intertemplate_cc_loop(wf, full, max_lag=max_lag)
        assert np.array_equal(got, want), (T, S, C, L)
        assert np.array_equal(workflow.intertemplate_cc(wf, full, max_lag=max_lag), want)   # factorisation recognised
        mask01 = (float(mask[0, 1]) + float(mask[1, 0])) / 2.0      # template 1 = template 0 shifted by 2 samples
        if max_lag >= 2:
            assert got[0, 1] > 0.9 * mask01 - 1e-6
        assert abs(got[0, 0] - base[0].sum()) < 1e-5
    # a weight array that does not factorise falls back to the loop
    T, S, C, L = 6, 4, 3, 80
    wf = rng.standard_normal((T, S, C, L)).astype(np.float32)
    w = rng.random((T, T, S, C)).astype(np.float32)
    assert workflow.factorise_pair_weights(w) is None
    assert np.array_equal(workflow.intertemplate_cc(wf, w, max_lag=7), workflow.intertemplate_cc_loop(wf, w, max_lag=7))


def test_bp_detections_on_device_edge_cases():
    """No peak above the threshold, a threshold given as an array, a record buffer that overflows
    and grows, NaNs in the series, a window longer than the series."""
    import torch
    from seismic_bpmf_amd import postprocess as pp
    from seismic_bpmf_amd.threshold import BeamDetectorGPU
    from seismic_bpmf_amd.workflow import beam_detections_device
    rng = np.random.default_rng(2)
    n = 30_000
    x = np.abs(rng.standard_normal(n)).astype(np.float32)
    src = rng.integers(0, 100, n).astype(np.int32)
    xd, sd = torch.as_tensor(x, device="cuda"), torch.as_tensor(src, device="cuda")
    peaks, psrc, _ = beam_detections_device(xd, sd, mpd=50, threshold=100.0)
    assert peaks.size == 0 and psrc.size == 0 and peaks.dtype.kind == "i"
    thr = (2.5 + np.sin(np.arange(n) / 3000.0)).astype(np.float32)
    want_p, want_s = pp.find_beam_detections(x, src, thr, 50)
    peaks, psrc, _ = beam_detections_device(xd, sd, mpd=50, threshold=thr)
    assert np.array_equal(peaks, want_p) and np.array_equal(psrc, want_s) and want_p.size > 10
    det = BeamDetectorGPU()
    few = det.extract_peaks(xd, sd, 1.0, capacity=4)            # overflows, grows, repeats
    many = det.extract_peaks(xd, sd, 1.0)
    assert few.size > 4 and np.array_equal(few, many) and np.all(np.diff(many["index"]) > 0)
    loc = np.flatnonzero((x[1:-1] > x[:-2]) & (x[2:] <= x[1:-1]) & (x[1:-1] > 1.0)) + 1
    assert np.array_equal(many["index"], loc) and np.array_equal(many["source"], src[loc])
    xn = x.copy()
    xn[[500, 501, 9000]] = np.nan                                # peaks next to a NaN are dropped
    want_p, _ = pp.find_beam_detections(xn, src, 2.0, 25)
    peaks, _, _ = beam_detections_device(torch.as_tensor(xn, device="cuda"), sd, mpd=25, threshold=2.0)
    assert np.array_equal(peaks, want_p)
    with pytest.raises(ValueError):
        det.window_stats(xd, n + 1, 0.75)
    # overlap 0 and a series of a whole number of windows: the reference's last window is empty, its
    # node NaN (np.median of an empty slice), and no peak is kept where the interpolated threshold is NaN
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want_thr = pp.bp_time_dependent_threshold(x, 3_000, 3.0, overlap=0.0)
        want_p, want_s = pp.find_beam_detections(x, src, want_thr, 50)
    assert np.isnan(want_thr[-1]) and want_p.size > 3
    peaks, psrc, nodes = beam_detections_device(xd, sd, mpd=50, window=3_000, n_dev=3.0, overlap=0.0)
    assert np.array_equal(peaks, want_p) and np.array_equal(psrc, want_s) and np.isnan(nodes[1][-2])


def test_relocation_focus_equals_numpy_on_the_oracle_volume(oracle_lib):
    """workflow.relocation_focus (beam volume and its arg-max on the device) == the reference's lines
    on the oracle's volume: np.unravel_index(beam.argmax(), beam.shape) and beam[:, time_idx]
    (BPMF/dataset.py:2193-2216), with exact ties in the volume; and the "temporal" flavour."""
    from seismic_bpmf_amd import BeamformerGPU, synthetic as syn
    from seismic_bpmf_amd.workflow import relocation_focus
    geo = syn.make_bp_geometry((12, 12, 6), 9, 2, 50.0, n_closest=5)
    tau, ws = geo["moveouts"], geo["weights_sources"]
    wp = syn.phase_weights(9, 3, 2)
    rng = np.random.default_rng(4)
    bf = BeamformerGPU(tau, ws)
    try:
        for n, ties in [(1500, False), (3000, True), (700, True)]:
            f = np.abs(rng.standard_normal((9, 3, n))).astype(np.float32)
            if ties:
                f = np.round(f)                                              # many equal maxima
            vol = oracle_lib.beamform(f, tau, wp, ws, "flexible", "none")
            k, t = np.unravel_index(vol.argmax(), vol.shape)
            src_idx, time_idx, column = relocation_focus(bf, f, wp, "spatial")
            assert (src_idx, time_idx) == (int(k), int(t)) and np.array_equal(column, vol[:, t]), (n, ties)
            mb, ma = oracle_lib.beamform(f, tau, wp, ws, "flexible", "max")
            src_idx, time_idx, maxbeam = relocation_focus(bf, f, wp, "temporal")
            assert time_idx == int(mb.argmax()) and src_idx == int(ma[time_idx]) and np.array_equal(maxbeam, mb)
    finally:
        bf.close()


def test_matched_filter_detections_sanity_check_rejects_gappy_days(oracle_lib):
    """The reference's sanity check (BPMF/similarity_search.py:633-642, on by default): a template whose
    CC series has an excess kurtosis above max_kurto yields no detection.  Decisions equal
    scipy.stats.kurtosis on the oracle's CC rows; a day that is mostly a gap is rejected, a full day
    is not."""
    from scipy.stats import kurtosis
    from seismic_bpmf_amd import synthetic as syn, workflow
    m = syn.make_mf_inputs(T=3, S=4, C=3, L=64, N=60_000, seed=5, max_moveout=200, n_events=4)
    kw = dict(step=1, sr=100.0, threshold_window_dur=50.0, minimum_interevent_time=2.0, remove_edges=False,
              white_noise=np.random.default_rng(0).standard_normal(500).astype(np.float32))
    full, cc = workflow.matched_filter_detections(m["templates"], m["moveouts"], m["weights"], m["data"], **kw)
    cc_ref = oracle_lib.matched_filter(m["templates"], m["moveouts"], m["weights"], m["data"], 1)
    k = workflow.row_excess_kurtosis(cc)
    # the reference hands SciPy the float32 series: bit-identical to that
    assert k.dtype == np.float32 and np.array_equal(k, kurtosis(cc_ref, axis=1)) and (k < 100).all()
    assert all(len(full[t]) >= 1 for t in range(3))
    gappy = m["data"].copy()
    gappy[:, :, 1_500:] = 0.0                                   # 97 % of the day missing: CC exactly 0 there
    cc_g = oracle_lib.matched_filter(m["templates"], m["moveouts"], m["weights"], gappy, 1)
    want_reject = kurtosis(cc_g.astype(np.float64), axis=1) > 100.0
    assert want_reject.any()
    on, _ = workflow.matched_filter_detections(m["templates"], m["moveouts"], m["weights"], gappy, **kw)
    off, _ = workflow.matched_filter_detections(m["templates"], m["moveouts"], m["weights"], gappy, sanity_check=False, **kw)
    for t in range(3):
        assert (len(on[t]) == 0) if want_reject[t] else np.array_equal(on[t], off[t])


def test_row_kurtosis_is_scipy_on_float32_bit_for_bit(hip_opts):
    """bpmf_row_kurtosis_dev == scipy.stats.kurtosis of the float32 rows (NumPy's pairwise sums over
    8192-element chunks, float32 powers, the float64 division by the count): lengths around every
    boundary of the summation tree, a constant row (NaN), a mostly-zero row, a day-sized row; the full
    chunks summed by one workgroup each through LDS (round 5, the default) and by the thread-per-leaf
    kernels (option stats.kurt_full_chunks 0) -- with an Inf and a NaN in a full chunk as well."""
    import torch
    from scipy.stats import kurtosis
    from seismic_bpmf_amd import postprocess as pp, workflow
    rng = np.random.default_rng(11)
    for n in (1, 5, 7, 8, 9, 127, 128, 129, 143, 255, 257, 1000, 8191, 8192, 8193, 16384, 16385, 24_576 + 57, 100_003, 2_000_001):
        rows = 3 if n > 100_000 else 7
        x = (rng.standard_normal((rows, n)) * rng.uniform(0.01, 2.0, (rows, 1)) + rng.uniform(-1, 1, (rows, 1))).astype(np.float32)
        if n > 4:
            x[1, : n // 2] = 0.0
            x[2] = x[2, 0]
        if rows > 5 and n > 9000:
            x[5, 4097] = np.inf
            x[6, 8000] = np.nan
        with np.errstate(all="ignore"):
            want = np.array([pp.excess_kurtosis_f32(r) for r in x], dtype=np.float32)
            ref = np.asarray([kurtosis(r) for r in x], dtype=np.float32)      # one series at a time, as BPMF calls it
        assert np.array_equal(want, ref, equal_nan=True), n
        for full in (1, 0):
            hip_opts("stats.kurt_full_chunks", full)
            got = workflow.row_excess_kurtosis(torch.as_tensor(x, device="cuda"))
            assert np.array_equal(got, want, equal_nan=True), (n, full, got, want)


def test_row_kurtosis_follows_scipy_on_one_series_where_the_scalar_power_is_not_the_square():
    """tests/golden/kurtosis_scalar_pow_row.npz: a row (found by the fuzz sweep) whose m2 the C library's powf squares
    one ulp away from the exact square.  SciPy on ONE series -- BPMF/similarity_search.py:640 -- works on NumPy
    scalars, whose `**` is that powf; on a 2-D array it squares exactly.  The device hands out (mean, m2, m4) and the
    host finishes on scalars: the result is SciPy's on the series, whichever way this platform's powf rounds."""
    import os
    import torch
    from scipy.stats import kurtosis
    from seismic_bpmf_amd import workflow
    row = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kurtosis_scalar_pow_row.npz"))["row"]
    want = np.float32(kurtosis(row))
    got = workflow.row_excess_kurtosis(torch.as_tensor(np.stack([row, row[::-1].copy()]), device="cuda"))
    assert got[0] == want and got[1] == np.float32(kurtosis(row[::-1].copy()))


def test_relocation_likelihood_on_device_equals_the_host_mirror(oracle_lib):
    """workflow.relocation_likelihood: focus point and likelihood of the beam column computed where the
    (K, N) volume lies, against the oracle's volume + the host mirror pinned to the reference."""
    from seismic_bpmf_amd import BeamformerGPU, postprocess as pp, synthetic as syn, workflow
    geo = syn.make_bp_geometry((12, 12, 6), 9, 2, 50.0, n_closest=9)
    feat, planted = syn.make_bp_features(geo["moveouts"], 9, 3, 6000, sr=50.0, n_events=1, amp=20.0)
    wp = syn.phase_weights(9, 3, 2)
    bf = BeamformerGPU(geo["moveouts"], geo["weights_sources"])
    try:
        dom = np.arange(0, geo["moveouts"].shape[0], 3)
        src, t_idx, like = workflow.relocation_likelihood(bf, feat, wp, "flexible", domain=dom)
        _, _, like_all = workflow.relocation_likelihood(bf, feat, wp, "flexible")
    finally:
        bf.close()
    vol = oracle_lib.beamform(feat, geo["moveouts"], wp, geo["weights_sources"], "flexible", "none")
    k_ref, t_ref = np.unravel_index(vol.argmax(), vol.shape)
    assert (src, t_idx) == (int(k_ref), int(t_ref)) and planted and abs(t_idx - planted[0][1]) <= 3
    want = pp.likelihood(vol[:, t_ref])
    assert like_all.dtype == np.float32 and np.array_equal(like_all, want) and np.array_equal(like, want[dom])
