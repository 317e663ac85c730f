"""The two detection pipelines around the hot paths, index logic only (no obspy Event objects).

* :func:`matched_filter_detections` = ``MatchedFilter.compute_cc_time_series`` +
  ``find_detections`` / ``_find_detections_t`` (BPMF/similarity_search.py:476-546, 548-666):
  CC on the device, RMS threshold on the device, candidates to the host, the reference's
  pair-wise merge on the (tiny) candidate list.  Returns per-template CC indices.
* :func:`backprojection_detections` = ``Beamformer.backproject`` + ``find_detections``
  (BPMF/template_search.py:508-572, 574-627).
* :func:`intertemplate_cc` = the core of ``TemplateGroup.compute_intertemplate_cc``
  (BPMF/dataset.py:4775-4830).
"""
import numpy as np

from . import postprocess as pp
from .beampower import BeamformerGPU
from .matched_filter import MatchedFilterGPU
from .threshold import ThresholdGPU


def search_window(moveouts_t, minimum_interevent_samp, step):
    """Merge distance between detections of one template, BPMF/similarity_search.py:651-659:
    the median over stations of the moveout spread between components, bounded by 1-10x the
    minimum inter-event time, in CC-step units."""
    d_mv = np.max(moveouts_t, axis=-1) - np.min(moveouts_t, axis=-1)
    d_mv = int(np.median(d_mv)) + 1
    win = min(10 * minimum_interevent_samp, max(d_mv, minimum_interevent_samp))
    return win / step


def search_windows(moveouts, minimum_interevent_samp, step):
    """search_window for every template of a (T, S, C) moveout array at once (same integers)."""
    d_mv = np.max(moveouts, axis=-1) - np.min(moveouts, axis=-1)
    d_mv = np.median(d_mv, axis=-1).astype(np.int64) + 1
    win = np.minimum(10 * minimum_interevent_samp, np.maximum(d_mv, minimum_interevent_samp))
    return win / step


def merge_candidates(index, cc, search_win):
    """The reference's sequential pair-wise merge (BPMF/similarity_search.py:240-251) applied to
    the candidate list: neighbours closer than `search_win` keep only the larger CC."""
    idx = np.asarray(index).tolist()
    val = np.asarray(cc).tolist()
    q = 1
    while q < len(idx):
        if idx[q] - idx[q - 1] < search_win:
            drop = q - 1 if val[q] > val[q - 1] else q
            del idx[drop], val[drop]
        else:
            q += 1
    return np.asarray(idx, dtype=np.int64)


def row_excess_kurtosis(cc):
    """scipy.stats.kurtosis (Fisher, biased: m4 / m2**2 - 3) of every row of a (T, n) float32 device
    tensor, evaluated as SciPy evaluates it on the reference's float32 CC series
    (BPMF/similarity_search.py:633-642): float32 moments in NumPy's summation order
    (csrc/stats.hip, bpmf_row_kurtosis_parts_dev; host mirror postprocess.excess_kurtosis_f32).  Returns a
    float32 NumPy array; NaN for a constant row, like SciPy."""
    import ctypes as C
    import torch
    from . import _lib
    x = cc if cc.dim() == 2 else cc.reshape(1, -1)
    x = x.to(dtype=torch.float32).contiguous()
    rows, n = x.shape
    if rows == 0:
        return np.zeros(0, dtype=np.float32)
    lib = _lib.lib()
    parts = torch.empty((rows, 3), dtype=torch.float32, device=x.device)
    chunk = min(rows, 65535)                   # rows per call of the library (its gridDim.y)
    ws = torch.empty(lib.bpmf_row_kurtosis_workspace_bytes(chunk, n), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        for r0 in range(0, rows, chunk):
            rc = lib.bpmf_row_kurtosis_parts_dev(C.c_void_p(x[r0:].data_ptr()), min(chunk, rows - r0), n,
                                                 C.c_void_p(ws.data_ptr()), ws.numel(),
                                                 C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream),
                                                 None, C.c_void_p(parts[r0:].data_ptr()))
            _lib.check(rc, "bpmf_row_kurtosis_parts_dev")
    # (mean, m2, m4) per row from the device; the last expression on NumPy scalars, as SciPy evaluates it on one
    # series -- postprocess.kurtosis_from_moments_f32 says why that is not the same as finishing on the device
    from .postprocess import kurtosis_from_moments_rows
    return kurtosis_from_moments_rows(parts.cpu().numpy())


def validation_windows(index, n, win):
    """The two windows MatchedFilter.select_cc_indexes inspects around CC index `index` of a series of n samples
    (BPMF/similarity_search.py:256-263): (start, len_left, len_right) with left = [start, start + len_left) and
    right the len_right samples behind it."""
    half = win // 2
    i0 = np.maximum(0, np.asarray(index, dtype=np.int64) - half)
    i1 = i0 + win
    late = i1 >= n
    i0 = np.where(late, n - 1 - win, i0)
    return i0, half, win - half


def validate_detections(cc, merged, bounds, c_index, c_thr, *, n_dev, threshold_type, cut, win):
    """The anomalous-CDF validation of select_cc_indexes (BPMF/similarity_search.py:253-272) for the merged
    detections {row: cc indices} of a CC matrix in HBM: returns the same dict with the failed detections removed.
    `c_index` / `c_thr` / `bounds`: the candidate arrays of cc_detections (the threshold AT a detection is the
    candidate record's).  Level and fractions as the reference computes them: float32 threshold / n_dev (x 1.48),
    float32 compares, count / float(len) in float64."""
    import ctypes as C
    import torch
    from . import _lib
    n = int(cc.shape[-1])
    if win < 2 or win > n - 1:
        raise ValueError(f"validation window of {win} CC samples on a series of {n}: the reference's slices need "
                         "2 <= int(window_for_validation_Tmax / min_freq_hz) <= n_corr - 1")
    rows, idxs, lvl = [], [], []
    for t, idx in merged.items():
        if len(idx) == 0:
            continue
        b0, b1 = bounds[t], bounds[t + 1]
        pos = b0 + np.searchsorted(c_index[b0:b1], idx)
        one_sigma = c_thr[pos].astype(np.float32) / np.float32(n_dev)
        if threshold_type == "mad":
            one_sigma = one_sigma * np.float32(1.48)
        rows.append(np.full(len(idx), t, dtype=np.int32))
        idxs.append(np.asarray(idx, dtype=np.int64))
        lvl.append(one_sigma.astype(np.float32))
    if not rows:
        return merged
    rows, idxs, lvl = np.concatenate(rows), np.concatenate(idxs), np.concatenate(lvl)
    start, n_left, n_right = validation_windows(idxs, n, win)
    dev = cc.device
    d_rows = torch.as_tensor(rows, device=dev)
    d_start = torch.as_tensor(start.astype(np.int64), device=dev)
    d_nl = torch.full((len(rows),), n_left, dtype=torch.int32, device=dev)
    d_nr = torch.full((len(rows),), n_right, dtype=torch.int32, device=dev)
    d_lvl = torch.as_tensor(lvl, device=dev)
    below = torch.empty((len(rows), 2), dtype=torch.int32, device=dev)
    x = cc if cc.is_contiguous() else cc.contiguous()
    with torch.cuda.device(dev):
        rc = _lib.lib().bpmf_count_below_dev(C.c_void_p(x.data_ptr()), int(x.shape[0]), n, len(rows),
                                             C.c_void_p(d_rows.data_ptr()), C.c_void_p(d_start.data_ptr()),
                                             C.c_void_p(d_nl.data_ptr()), C.c_void_p(d_nr.data_ptr()),
                                             C.c_void_p(d_lvl.data_ptr()),
                                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream),
                                             C.c_void_p(below.data_ptr()))
    _lib.check(rc, "bpmf_count_below_dev")
    below = below.cpu().numpy().astype(np.float64)
    frac = np.minimum(below[:, 0] / float(n_left), below[:, 1] / float(n_right))
    keep = ~(frac < cut)
    out = dict(merged)
    for t in np.unique(rows):
        m = rows == t
        out[int(t)] = idxs[m][keep[m]]
    return out


def matched_filter_detections(templates, moveouts, weights, data, *, step=1, sr,
                              threshold_window_dur, minimum_interevent_time, n_dev=8.0,
                              overlap=0.25, max_cc_threshold=0.80, white_noise=None, device=None,
                              remove_edges=True, data_buffer_sec=None, data_duration_sec=None,
                              sanity_check=True, max_kurto=100.0, threshold_type="rms",
                              anomalous_cdf_at_mean_plus_1sig=0.0, window_for_validation_Tmax=100.0, min_freq_hz=None):
    """Matched-filter search of one day: returns ({template: cc indices}, cc device tensor).

    `remove_edges` (the reference's default, BPMF/similarity_search.py:274-285) drops detections
    inside the `data_buffer_sec` margins the day was loaded with (`cfg.DATA_BUFFER_SEC`) and past
    `data_duration_sec + data_buffer_sec`; it needs `data_buffer_sec` (pass remove_edges=False for a
    day loaded without margins).  The optional anomalous-CDF validation of :253-272
    (`anomalous_cdf_at_mean_plus_1sig` > 0 with `min_freq_hz`; off by default like the reference's class default)
    runs on the device too, see cc_detections.  `sanity_check` (the reference's default, :633-642): a
    template whose CC series has an excess kurtosis above `max_kurto` -- most of the day missing --
    yields no detection (the reference zeroes its CCs before the peak selection); `threshold_type`:
    the RMS threshold of libc.c (default) or the MAD threshold of similarity_search.py:1079-1113; the kurtosis is
    scipy.stats.kurtosis of the float32 row, evaluated on the device in SciPy's own float32 order.
    `remove_edges=True` without `data_buffer_sec` raises: the reference always trims
    cfg.DATA_BUFFER_SEC, there is no silent default here."""
    if remove_edges and data_buffer_sec is None:
        raise ValueError("remove_edges=True needs data_buffer_sec (the reference trims cfg.DATA_BUFFER_SEC); "
                         "pass remove_edges=False for a day without margins")
    weights = np.asarray(weights, dtype=np.float32)
    mf = MatchedFilterGPU(device=device)
    mf.set_data(data)
    cc = mf.run(templates, moveouts, weights, step)
    out = cc_detections(cc, moveouts, weights, step=step, sr=sr, threshold_window_dur=threshold_window_dur,
                        minimum_interevent_time=minimum_interevent_time, n_dev=n_dev, overlap=overlap,
                        max_cc_threshold=max_cc_threshold, white_noise=white_noise, device=device,
                        remove_edges=remove_edges, data_buffer_sec=data_buffer_sec,
                        data_duration_sec=data_duration_sec, sanity_check=sanity_check, max_kurto=max_kurto,
                        threshold_type=threshold_type, anomalous_cdf_at_mean_plus_1sig=anomalous_cdf_at_mean_plus_1sig,
                        window_for_validation_Tmax=window_for_validation_Tmax, min_freq_hz=min_freq_hz)
    return out, cc


def cc_detections(cc, moveouts, weights, *, step=1, sr, threshold_window_dur, minimum_interevent_time,
                  n_dev=8.0, overlap=0.25, max_cc_threshold=0.80, white_noise=None, device=None,
                  remove_edges=True, data_buffer_sec=None, data_duration_sec=None, sanity_check=True,
                  max_kurto=100.0, threshold_type="rms", with_values=False, timings=None,
                  anomalous_cdf_at_mean_plus_1sig=0.0, window_for_validation_Tmax=100.0, min_freq_hz=None):
    """The detection stage of matched_filter_detections on a (T, n_corr) CC matrix that already lies in
    HBM (``MatchedFilter.find_detections``, BPMF/similarity_search.py:548-666): NaN scrub, threshold and
    candidates on the device, the reference's pair-wise merge, kurtosis sanity check, edge removal.
    `moveouts` / `weights` are those of the rows of `cc`.  Returns {row: cc indices}; with
    `with_values` {row: (cc indices, cc values float32, threshold values float32)}.  `timings`: a dict
    that receives "threshold_ms" / "candidates_ms" / "merge_ms" / "candidates" (the device is synchronised
    around the stages only when it is given).

    `anomalous_cdf_at_mean_plus_1sig` > 0 (the class default of the reference is 0.0 = off, similarity_search.py:40;
    its select_cc_indexes defaults to 0.50): the validation of :253-272 between the merge and the edge removal --
    around every merged detection, two adjacent windows of the CC row (together int(window_for_validation_Tmax /
    min_freq_hz) samples, the reference's own unit) must each hold at least that fraction of samples below
    threshold / n_dev (x 1.48 under the MAD threshold), or the threshold is deemed to have failed there and the
    detection is dropped.  The counts are taken on the device (bpmf_count_below_dev: one workgroup per detection),
    the CC rows stay in HBM; needs `min_freq_hz` (cfg.MIN_FREQ_HZ)."""
    import time
    if remove_edges and data_buffer_sec is None:
        raise ValueError("remove_edges=True needs data_buffer_sec (the reference trims cfg.DATA_BUFFER_SEC); "
                         "pass remove_edges=False for a day without margins")
    weights = np.asarray(weights, dtype=np.float32)
    if device is None and hasattr(cc, "device"):
        device = cc.device.index
    cc.nan_to_num_(nan=0.0)                                    # similarity_search.py:540
    th = ThresholdGPU(device=device)

    def clock():
        if timings is None:
            return 0.0
        import torch
        torch.cuda.synchronize(cc.device)
        return time.perf_counter()

    t_0 = clock()
    window = int(pp.sec_to_samp(threshold_window_dur, sr))
    threshold_type = threshold_type.lower()
    if threshold_type == "rms":
        thr_win, _ = th.time_dependent_threshold(cc, window, n_dev, overlap=overlap,
                                                 white_noise=white_noise)
    elif threshold_type == "mad":                                               # :1079-1113
        # The reference fills the zeros of a series with white_noise[:n_zeros] and draws one value per
        # sample; a shorter array (the 500 values the RMS variant cycles through) is repeated up to
        # the series length here -- the reference itself would fail to broadcast it.
        if white_noise is not None and len(white_noise) < cc.shape[-1]:
            white_noise = np.resize(np.asarray(white_noise, dtype=np.float32), cc.shape[-1])
        thr_win, _ = th.time_dependent_threshold_mad(cc, window, n_dev, overlap=overlap,
                                                     white_noise=white_noise, expand=False)
    else:
        raise ValueError("threshold_type must be 'rms' or 'mad'")
    t_1 = clock()
    cap = max_cc_threshold * weights.reshape(weights.shape[0], -1).sum(axis=1)   # :629
    cand = th.extract_candidates(cc, thr_win, window, overlap=overlap, row_cap=cap, kind=threshold_type)
    t_2 = clock()
    min_iet = int(pp.sec_to_samp(minimum_interevent_time, sr))
    mv = np.asarray(moveouts)
    rejected = row_excess_kurtosis(cc) > max_kurto if sanity_check else np.zeros(weights.shape[0], bool)
    out = {}
    # (the candidates come sorted by row, then index: the rows are slices of four plain arrays, not 500 masks over
    # all records and field views of each; the search windows of all templates in one go)
    n_t = weights.shape[0]
    bounds = np.searchsorted(cand["row"], np.arange(n_t + 1))
    c_index = np.ascontiguousarray(cand["index"])
    c_cc = np.ascontiguousarray(cand["cc"])
    c_thr = np.ascontiguousarray(cand["threshold"])
    wins = search_windows(mv.reshape(n_t, mv.shape[1], -1), min_iet, step)
    lo_edge = pp.sec_to_samp(data_buffer_sec, sr) if remove_edges else None
    hi_edge = pp.sec_to_samp(data_duration_sec + data_buffer_sec, sr) if remove_edges and data_duration_sec is not None else None
    empty = (np.zeros(0, np.int64), np.zeros(0, np.float32), np.zeros(0, np.float32)) if with_values \
        else np.zeros(0, dtype=np.int64)
    merged = {}
    for t in range(n_t):
        if not rejected[t]:
            b0, b1 = bounds[t], bounds[t + 1]
            merged[t] = merge_candidates(c_index[b0:b1], c_cc[b0:b1], wins[t])
    if anomalous_cdf_at_mean_plus_1sig > 0.0:
        if min_freq_hz is None:
            raise ValueError("anomalous_cdf_at_mean_plus_1sig > 0 needs min_freq_hz (the reference's cfg.MIN_FREQ_HZ)")
        merged = validate_detections(cc, merged, bounds, c_index, c_thr, n_dev=n_dev, threshold_type=threshold_type,
                                     cut=anomalous_cdf_at_mean_plus_1sig,
                                     win=int(1.0 / min_freq_hz * window_for_validation_Tmax))
    for t in range(n_t):
        if rejected[t]:
            out[t] = empty
            continue
        b0, b1 = bounds[t], bounds[t + 1]
        idx = merged[t]
        if remove_edges:                        # (data_buffer_sec is given: checked on entry)
            idx = idx[idx * step >= lo_edge]
            if hi_edge is not None:
                idx = idx[idx * step < hi_edge]
        if with_values:
            pos = b0 + np.searchsorted(c_index[b0:b1], idx)     # candidates come sorted by index within a row
            out[t] = (idx, c_cc[pos], c_thr[pos])
        else:
            out[t] = idx
    if timings is not None:
        timings.update(threshold_ms=(t_1 - t_0) * 1e3, candidates_ms=(t_2 - t_1) * 1e3,
                       merge_ms=(time.perf_counter() - t_2) * 1e3, candidates=int(cand.size))
    return out


# ------------------------------------------------------------------ one process per GPU ---
RECORD_WIDTH = 4      # (global template id, cc index, float32 bits of cc, float32 bits of the threshold) as int64


def detections_to_records(detections, t_offset=0):
    """{row: (indices, cc, threshold)} -> (n, 4) int64 NumPy records with GLOBAL template ids."""
    rows = []
    for t in sorted(detections):
        idx, val, thr = detections[t]
        if len(idx) == 0:
            continue
        r = np.empty((len(idx), RECORD_WIDTH), dtype=np.int64)
        r[:, 0] = t + t_offset
        r[:, 1] = idx
        r[:, 2] = np.asarray(val, dtype=np.float32).view(np.int32)
        r[:, 3] = np.asarray(thr, dtype=np.float32).view(np.int32)
        rows.append(r)
    return np.concatenate(rows) if rows else np.zeros((0, RECORD_WIDTH), dtype=np.int64)


def records_to_detections(records, n_templates):
    """(n, 4) int64 records -> {template: (indices int64, cc float32, threshold float32)} for every template."""
    records = np.asarray(records, dtype=np.int64).reshape(-1, RECORD_WIDTH)
    order = np.lexsort((records[:, 1], records[:, 0]))
    records = records[order]
    out = {}
    for t in range(n_templates):
        mine = records[records[:, 0] == t]
        out[t] = (mine[:, 1].copy(), mine[:, 2].astype(np.int32).view(np.float32),
                  mine[:, 3].astype(np.int32).view(np.float32))
    return out


def sharded_matched_filter_detections(templates, moveouts, weights, data, *, group=None, device=None,
                                      engine=None, detector=None, data_src=None, balance=True, step=1,
                                      **detection_kwargs):
    """Matched-filter search of one day on ALL ranks of a torch.distributed group (one process per GPU):
    what ``MatchedFilter.run_matched_filter_search`` (BPMF/similarity_search.py:726-807) does with its
    sequential template chunks, the chunks being the ranks' shards here.

    Every rank passes the SAME templates / moveouts / weights (all T of them).  `data`: the (S, C, N) day on
    every rank, or -- with `data_src=r` -- on rank r only (None elsewhere): rank r uploads it and
    broadcasts it over RCCL / xGMI, the host of the other ranks never touches it (SURVEY.md section 8e:
    "broadcast once per day").  Rank r computes the CC of its block of templates (balanced by weighted
    channels), thresholds it and selects its detections on its own GPU (cc_detections); the (template,
    index, cc, threshold) records -- a few KB -- are all-gathered; the CC matrix never leaves the GPU
    that made it.  `detection_kwargs` are cc_detections' (sr, threshold_window_dur, ...).

    Returns (detections, info): detections = {global template id: (cc indices, cc, threshold)} for ALL
    templates, identical on every rank and equal to what matched_filter_detections(..., with values)
    finds in one process; info = {"templates": (t0, t1), "cc": this rank's CC tensor or None,
    "records_gathered": n, "broadcast_ms": float or None}.

    `engine` / `detector`: stand-ins for the per-rank MatchedFilterGPU and for cc_detections (the CPU tests
    of the choreography over gloo pass oracle-backed ones); None = the HIP path.

    threshold_type="mad" needs an explicit `white_noise` here: without one every rank would draw its own
    (np.random, as the reference does per call) and the ranks' detections would not be those of one process."""
    import time
    import torch
    from . import parallel
    if detector is None and str(detection_kwargs.get("threshold_type", "rms")).lower() == "mad" and \
            detection_kwargs.get("white_noise") is None:
        raise ValueError("sharded_matched_filter_detections: threshold_type='mad' needs white_noise (the same array on "
                         "every rank); each rank would otherwise draw its own")
    smf = parallel.ShardedMatchedFilter(group=group, device=device, local=engine)
    t_b = None
    if data_src is None:
        smf.set_data(data)
    else:
        t0c = time.perf_counter()
        smf.set_data_broadcast(data, src=data_src)
        t_b = (time.perf_counter() - t0c) * 1e3
    weights = np.asarray(weights, dtype=np.float32)
    T = weights.shape[0]
    t0, t1, cc = smf.run(templates, moveouts, weights, step, True, balance=balance)
    if cc is None:
        mine = {}
    elif detector is not None:
        mine = detector(cc, np.asarray(moveouts)[t0:t1], weights[t0:t1], step=step, **detection_kwargs)
    else:
        mine = cc_detections(cc, np.asarray(moveouts)[t0:t1], weights[t0:t1], step=step, device=device,
                             with_values=True, **detection_kwargs)
    rec = detections_to_records(mine, t_offset=t0)
    rec_dev = smf.record_device()
    parts = parallel.allgather_varlen(torch.as_tensor(rec, device=rec_dev), group=group)
    everything = np.concatenate([p.cpu().numpy() for p in parts]) if parts else rec
    info = {"templates": (t0, t1), "cc": cc, "records_gathered": int(everything.shape[0]), "broadcast_ms": t_b}
    return records_to_detections(everything, T), info


def sharded_backprojection_detections(features, moveouts, weights_phases, weights_sources, *, sr,
                                      minimum_interevent_time, threshold_window_dur=None, n_dev=15.0,
                                      overlap=0.75, threshold=None, out_of_bounds="strict", group=None,
                                      device=None, engine_factory=None, detector=None, features_src=None):
    """Backprojection of one day with the source grid tiled over the ranks of a group: every rank scans
    its block of sources (global ids) against the whole day of features, ONE all-reduce(MAX) of packed
    (beam, id) keys leaves the global max-beam and arg-max on every rank (ties -> the lowest source id,
    as one sequential scan), and every rank runs the (3 ms) detection stage on them.  Mirrors
    ``Beamformer.backproject`` + ``find_detections`` (BPMF/template_search.py:508-627).

    `features`: the (S, C, N) day on every rank, or with `features_src=r` on rank r only (broadcast over
    the group).  Returns (peak samples, source indices, maxbeam, argmax) -- the tensors of
    backprojection_detections(return_device=True) -- identical on every rank.
    `engine_factory(moveouts_block, weights_block, source_id_offset)` / `detector(beam, arg)`: CPU
    stand-ins for BeamformerGPU / beam_detections_device in the gloo tests."""
    from . import parallel
    sb = parallel.ShardedBeamformer(moveouts, weights_sources, group=group, device=device,
                                    local_factory=engine_factory)
    try:                                 # (the per-rank engine -- plan and working set -- goes whatever happens)
        if features_src is not None:
            features = sb.broadcast_features(features, src=features_src)
        beam, arg = sb.run(features, weights_phases, "max", out_of_bounds)
        mpd = int(pp.sec_to_samp(minimum_interevent_time, sr))
        if detector is not None:
            peaks, peak_sources = detector(beam, arg)
        else:
            window = None if threshold is not None else int(pp.sec_to_samp(threshold_window_dur, sr))
            peaks, peak_sources, _ = beam_detections_device(beam, arg, mpd=mpd, threshold=threshold, window=window,
                                                           n_dev=n_dev, overlap=overlap, device=device)
    finally:
        sb.close()
    return peaks, peak_sources, beam, arg


def beam_detections_device(beam, arg, *, mpd, threshold=None, window=None, n_dev=15.0, overlap=0.75,
                           device=None):
    """``Beamformer.find_detections`` (BPMF/template_search.py:574-627) on a max-beam that lives
    on the device: `beam` (N,) float32 and `arg` (N,) int32 device tensors (the output of
    BeamformerGPU.run).  Nothing of length N is downloaded:

      1. threshold: per-window median / MAD by radix select on the device (csrc/bp_detect.hip),
         n_windows + 2 node values to the host (`threshold` None), or a scalar given by the caller;
      2. the rising-edge local maxima above the smallest threshold node, compacted on the device;
      3. on that list: tallest-first min-distance suppression, the float64 threshold test at the
         survivors, the +-mpd/2 snap (on windows of the beam gathered in one small transfer) and
         np.unique -- the reference's own index logic (postprocess.find_beam_detections_from_candidates).

    Returns (peak samples int64, source indices int32, threshold nodes (centre, values) or None)."""
    import torch
    from .threshold import BeamDetectorGPU
    det = BeamDetectorGPU(device=device if device is not None else beam.device.index)
    n = beam.numel()
    nodes = None
    if threshold is None:
        med, mad = det.window_stats(beam, window, overlap)
        centre, thr = pp.bp_threshold_nodes(n, int(window), overlap, med, mad, n_dev)
        nodes = (centre, thr)
        # a NaN node (a window that holds a NaN, or the reference's empty last window when the
        # series is a whole number of non-overlapping windows) makes the interpolated threshold
        # NaN around it, where the reference then keeps no peak (`>` is false); the floor is the
        # smallest finite node
        # smallest finite node -- less one float32 ulp: the interpolation (SciPy's order of operations,
        # pp.interp_threshold) can land an ulp of float64 below two equal nodes, and a peak exactly as high
        # as the node would then pass the reference's `>` but not the floor
        finite = thr[np.isfinite(thr)]
        floor = float(np.nextafter(np.float32(finite.min()), np.float32(-np.inf))) if finite.size else float("nan")

        def threshold_at(samples):
            return pp.interp_threshold(samples, centre, thr)
    elif np.ndim(threshold) == 0:
        floor = float(threshold)

        def threshold_at(samples):
            return np.full(len(samples), float(threshold))
    else:
        thr_arr = np.asarray(threshold)
        floor = float(np.nanmin(thr_arr)) if np.isfinite(thr_arr).any() else float("nan")

        def threshold_at(samples):
            return thr_arr[samples]
    if not np.isfinite(floor):               # no finite threshold anywhere: nothing can exceed it
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int32), nodes
    rec = det.extract_peaks(beam, arg, floor)
    if mpd > 1 and pp.has_close_ties(rec["index"], rec["beam"], mpd):
        # two exactly equal peaks closer than mpd: which one survives depends on where NumPy's
        # unstable sort puts them in the list of ALL local maxima (pp._tallest_first) -- fetch that
        # list (12 bytes per local maximum; rare on real beams, common on rounded test series)
        rec = det.extract_peaks(beam, arg, -np.inf)
    # beam windows around every candidate that can survive, in ONE gather + transfer: the snap
    # looks +-mpd/2 around a peak, and a snapped peak can be looked at once more (+-mpd covers it)
    half = int(mpd) + 1
    idx_all = rec["index"].astype(np.int64)
    cache = {}
    if idx_all.size:
        keep_rank = pp._tallest_first(rec["beam"].astype(np.float64))
        keep = pp._suppress(idx_all, keep_rank, mpd) if mpd > 1 else np.ones(idx_all.size, bool)
        cand = idx_all[keep]
        cand = cand[rec["beam"][keep].astype(np.float64) > threshold_at(cand)]
        if cand.size:
            offs = torch.arange(-half, half + 1, device=beam.device)
            pos = (torch.as_tensor(cand, device=beam.device)[:, None] + offs[None, :]).clamp_(0, n - 1)
            wins = beam.reshape(-1)[pos].cpu().numpy()
            for c, wrow in zip(cand, wins):
                cache[int(c)] = wrow
    flat = beam.reshape(-1)

    centres = np.array(sorted(cache), dtype=np.int64)     # a busy day has thousands of candidates: bisect

    def beam_slice(i0, i1):
        # a gathered window [c - half, c + half] that covers [i0, i1): c in [i1 - 1 - half, i0 + half]
        j = int(np.searchsorted(centres, i1 - 1 - half, side="left"))
        while j < centres.size and centres[j] <= i0 + half:
            c = int(centres[j])
            lo = c - half
            if lo >= 0 and c + half < n:         # windows clamped at the ends of the trace are not contiguous
                return cache[c][i0 - lo:i1 - lo]
            j += 1
        return flat[i0:i1].cpu().numpy()         # rare: a window no candidate's gather covers

    peaks = pp.find_beam_detections_from_candidates(idx_all, rec["beam"], threshold_at, mpd, n, beam_slice)
    if peaks.size:
        src = arg.reshape(-1)[torch.as_tensor(peaks, device=arg.device)].cpu().numpy()
    else:
        src = np.zeros(0, dtype=np.int32)
    return peaks, src, nodes


def relocation_focus(beamformer, features, weights_phases, uncertainty_method="spatial",
                     out_of_bounds="flexible"):
    """The beamforming step of the reference's event relocation, ``Event.relocate_beam``
    (BPMF/dataset.py:2186-2216), on a resident BeamformerGPU: the short feature array of one event
    (N ~ 1 500-3 000 samples) is backprojected over the whole grid and the point of maximum focusing
    is found on the device.

    "spatial" (``reduce="none"``): the (K, N) beam volume stays in HBM; returns
    ``(src_idx, time_idx, beam[:, time_idx])`` -- np.unravel_index(beam.argmax(), beam.shape), i.e.
    the first maximum in source-major order, and the column the reference turns into its location
    likelihood (K floats downloaded instead of K*N).  "temporal" (``reduce="max"``): returns
    ``(src_idx, time_idx, maxbeam)`` with time_idx = maxbeam.argmax() (first maximum) and
    src_idx = maxbeam_sources[time_idx].  `out_of_bounds` defaults to the reference's "flexible"
    (:2186)."""
    import torch
    if uncertainty_method == "spatial":
        vol = beamformer.run(features, weights_phases, "none", out_of_bounds)        # (K, N) on the device
        flat = vol.reshape(-1)
        first = int(torch.argmax(flat))      # the first maximum (the first NaN if there is one), as np.argmax;
                                             # a reduction: no K x N temporaries
        src_idx, time_idx = divmod(first, vol.shape[1])
        return src_idx, time_idx, vol[:, time_idx].cpu().numpy()
    if uncertainty_method == "temporal":
        beam, arg = beamformer.run(features, weights_phases, "max", out_of_bounds)
        time_idx = int(torch.argmax(beam))
        return int(arg[time_idx]), time_idx, beam.cpu().numpy()
    raise ValueError("uncertainty_method should be 'spatial' or 'temporal'")


def relocation_likelihood(beamformer, features, weights_phases, out_of_bounds="flexible", domain=None):
    """The spatial branch of ``Event.relocate_beam`` (BPMF/dataset.py:2186-2245) up to the likelihood,
    on the device: backprojection of the event's short feature array over the whole grid
    (``reduce="none"``, the (K, N) volume stays in HBM), the point of maximum focusing, and
    ``Beamformer._likelihood`` of the beam column at that time (BPMF/template_search.py:498-506) --
    float32 subtract / divide / clip, the operations NumPy performs, evaluated on the column where it
    lies.  Only `likelihood[domain]` (or the whole (K,) vector when `domain` is None) comes back.

    Returns (src_idx, time_idx, likelihood): feed `likelihood` and the domain's coordinates to
    postprocess.compute_location_uncertainty for (hunc, vunc)."""
    import torch
    vol = beamformer.run(features, weights_phases, "none", out_of_bounds)
    first = int(torch.argmax(vol.reshape(-1)))
    src_idx, time_idx = divmod(first, vol.shape[1])
    col = vol[:, time_idx]
    lo, hi = col.min(), col.max()
    like = ((col - lo) / (hi - lo)).clamp_(0.0, 1.0)
    if domain is not None:
        like = like[torch.as_tensor(np.asarray(domain, dtype=np.int64), device=like.device)]
    return src_idx, time_idx, like.cpu().numpy()


def backprojection_detections(features, moveouts, weights_phases, weights_sources, *, sr,
                              minimum_interevent_time, threshold_window_dur=None, n_dev=15.0,
                              overlap=0.75, threshold=None, out_of_bounds="strict", device=None,
                              return_device=False):
    """Backprojection of one day: returns (peak samples, source indices, maxbeam, argmax).

    The max-beam and its arg-max stay on the device through the whole detection stage
    (beam_detections_device); they are downloaded at the end only because this function returns
    them as arrays -- pass return_device=True to get the device tensors instead."""
    bf = BeamformerGPU(moveouts, weights_sources, device=device)
    beam, arg = bf.run(features, weights_phases, "max", out_of_bounds)
    mpd = int(pp.sec_to_samp(minimum_interevent_time, sr))
    window = None if threshold is not None else int(pp.sec_to_samp(threshold_window_dur, sr))
    peaks, peak_sources, _ = beam_detections_device(beam, arg, mpd=mpd, threshold=threshold,
                                                   window=window, n_dev=n_dev, overlap=overlap,
                                                   device=device)
    bf.close()
    if return_device:
        return peaks, peak_sources, beam, arg
    return peaks, peak_sources, beam.cpu().numpy(), arg.cpu().numpy()


def numpy_order_sum(x):
    """Sum over the last axis of a torch tensor in the order of numpy's float32 pairwise
    `np.sum` (n < 8: running sum; n <= 128: eight strided partial sums combined as a tree, then
    the tail; larger: split in halves of multiples of 8) -- element-wise adds only, so the
    result is bit-identical to `np.sum(a, axis=(-1, -2))` of the reference
    (dataset.py:4828-4830) whatever the device."""
    n = x.shape[-1]
    if n < 8:
        res = x.new_zeros(x.shape[:-1])
        for i in range(n):
            res = res + x[..., i]
        return res
    if n <= 128:
        r = [x[..., j] for j in range(8)]
        m = n - (n % 8)
        for i in range(8, m, 8):
            r = [r[j] + x[..., i + j] for j in range(8)]
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        for i in range(m, n):
            res = res + x[..., i]
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return numpy_order_sum(x[..., :n2]) + numpy_order_sum(x[..., n2:])


def factorise_pair_weights(weights):
    """(T, T, S, C) weights -> (base (T, S, C), mask (T, T) bool) if, for every t, all non-zero rows
    weights[t, u] are one and the same (S, C) pattern -- how the reference builds them
    (dataset.py:4789-4816: template t's normalised station weights, zeroed for the templates beyond
    the distance threshold); None otherwise."""
    w = np.asarray(weights, dtype=np.float32)
    T = w.shape[0]
    mask = (w != 0).reshape(T, T, -1).any(axis=2)
    base = np.zeros((T,) + w.shape[2:], dtype=np.float32)
    for t in range(T):
        rows = np.flatnonzero(mask[t])
        if rows.size == 0:
            continue
        base[t] = w[t, rows[0]]
        if not np.array_equal(w[t, rows], np.broadcast_to(base[t], (rows.size,) + base[t].shape)):
            return None
    return base, mask


def intertemplate_cc(waveforms_arr, weights, max_lag=10, device=None, pair_mask=None):
    """Pair-wise template similarity: intertp[t, u] = sum_{s,c} w[t][u,s,c] * max_lag CC, symmetrised.

    waveforms_arr (T,S,C,L).  weights: (T, S, C) with pair_mask (T, T) -- row t's channel weights and
    the pairs within the distance threshold, the factors the reference multiplies together
    (dataset.py:4789-4816) -- or the full (T, T, S, C) array / a callable t -> (T, S, C).  Factorised
    weights (given, or recognised in a full array) run as ONE batched launch
    (bpmf_intertemplate_cc_dev); anything else takes the per-template loop of the reference
    (intertemplate_cc_loop).  Both give the same bits."""
    import torch
    if np.shape(waveforms_arr)[-1] <= 2 * int(max_lag) or max_lag < 0:
        raise ValueError("intertemplate_cc: the waveforms must be longer than 2 * max_lag samples")
    if not callable(weights) and max_lag <= 31:      # the batched kernel keeps 2 * max_lag + 1 <= 63 lags per pair
        w = np.asarray(weights, dtype=np.float32)
        fact = None
        if w.ndim == 3:
            T = w.shape[0]
            fact = (w, np.ones((T, T), bool) if pair_mask is None else np.asarray(pair_mask, dtype=bool))
        elif w.ndim == 4 and pair_mask is None:
            fact = factorise_pair_weights(w)
        if fact is not None:
            base, mask = fact
            mf = MatchedFilterGPU(device=device)
            wf = mf._dev(np.ascontiguousarray(waveforms_arr, dtype=np.float32), torch.float32)
            T, S, Cc, Lw = wf.shape
            base_d = mf._dev(np.ascontiguousarray(base, dtype=np.float32), torch.float32)
            mask_d = torch.as_tensor(np.ascontiguousarray(mask, dtype=np.uint8), device=wf.device)
            lib = mf.lib
            ws = torch.empty(lib.bpmf_intertemplate_workspace_bytes(T, S, Cc, max_lag), dtype=torch.uint8,
                             device=wf.device)
            out = torch.empty((T, T), dtype=torch.float32, device=wf.device)
            import ctypes as C
            from . import _lib
            with torch.cuda.device(wf.device):
                rc = lib.bpmf_intertemplate_cc_dev(wf.data_ptr(), base_d.data_ptr(), mask_d.data_ptr(), T, S, Cc,
                                                   Lw, int(max_lag), ws.data_ptr(), ws.numel(),
                                                   C.c_void_p(torch.cuda.current_stream(wf.device).cuda_stream),
                                                   out.data_ptr())
            _lib.check(rc, "bpmf_intertemplate_cc_dev")
            o = out.cpu().numpy()
            return (o + o.T) / 2.0
    if not callable(weights) and np.ndim(weights) == 3:     # factorised weights, too many lags for the batched kernel
        w = np.asarray(weights, dtype=np.float32)
        T = w.shape[0]
        mask = np.ones((T, T), bool) if pair_mask is None else np.asarray(pair_mask, dtype=bool)
        weights = lambda t: w[t][None, :, :] * mask[t][:, None, None]
    return intertemplate_cc_loop(waveforms_arr, weights, max_lag=max_lag, device=device)


def intertemplate_cc_loop(waveforms_arr, weights, max_lag=10, device=None):
    """Pair-wise template similarity: intertp[t, u] = sum_{s,c} w[t][u,s,c] * max_lag CC.

    waveforms_arr (T,S,C,L); weights (T, T, S, C), or a callable t -> (T, S, C) -- row t holds the
    channel weights the reference builds for template t against every other template
    (dataset.py:4789-4816).  For each t the template's own waveform is the "data" and every
    template trimmed by max_lag on both sides is correlated against it at 2*max_lag+1 lags
    (network_sum=False, dataset.py:4818-4827).  Everything stays on the device -- the CCs, their
    max over the lags and the weighted channel sum (in numpy's summation order) -- and only the
    (T, T) matrix comes back."""
    import torch
    wf = np.ascontiguousarray(waveforms_arr, dtype=np.float32)
    if wf.shape[-1] <= 2 * int(max_lag) or max_lag < 0:
        raise ValueError("intertemplate_cc: the waveforms must be longer than 2 * max_lag samples")
    T, S, Cc = wf.shape[:3]
    mf = MatchedFilterGPU(device=device)
    wf_dev = mf._dev(wf, torch.float32)
    tp_dev = wf_dev[..., max_lag:wf.shape[-1] - max_lag].contiguous()
    mv0 = torch.zeros((T, S, Cc), dtype=torch.int32, device=wf_dev.device)
    out_dev = torch.zeros((T, T), dtype=torch.float32, device=wf_dev.device)
    for t in range(T):
        w = np.asarray(weights(t) if callable(weights) else weights[t], dtype=np.float32)
        keep = np.flatnonzero((w != 0).reshape(T, -1).sum(axis=1) > 0)
        if keep.size == 0:
            continue
        keep_dev = torch.as_tensor(keep, device=wf_dev.device)
        w_dev = mf._dev(w[keep], torch.float32)
        mf.set_data(wf_dev[t])
        cc = mf.run(tp_dev[keep_dev], mv0[keep_dev], w_dev, 1, network_sum=False)  # (n_keep, 2*max_lag+1, S, C)
        best = cc.max(dim=1).values
        out_dev[t, keep_dev] = numpy_order_sum((w_dev * best).reshape(keep.size, S * Cc))
    out = out_dev.cpu().numpy()
    return (out + out.T) / 2.0
