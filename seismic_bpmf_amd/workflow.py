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


def merge_candidates(index, cc, search_win):
    """The reference's sequential pair-wise merge (BPMF/similarity_search.py:240-251) applied to
    the candidate list: neighbours closer than `search_win` keep only the larger CC."""
    idx = list(index)
    val = list(cc)
    q = 1
    while q < len(idx):
        if idx[q] - idx[q - 1] < search_win:
            drop = q - 1 if val[q] > val[q - 1] else q
            del idx[drop], val[drop]
        else:
            q += 1
    return np.asarray(idx, dtype=np.int64)


def matched_filter_detections(templates, moveouts, weights, data, *, step=1, sr,
                              threshold_window_dur, minimum_interevent_time, n_dev=8.0,
                              overlap=0.25, max_cc_threshold=0.80, white_noise=None, device=None):
    """Matched-filter search of one day: returns ({template: cc indices}, cc device tensor)."""
    weights = np.asarray(weights, dtype=np.float32)
    mf = MatchedFilterGPU(device=device)
    mf.set_data(data)
    cc = mf.run(templates, moveouts, weights, step)
    cc.nan_to_num_(nan=0.0)                                    # similarity_search.py:540
    th = ThresholdGPU(device=device)
    window = int(pp.sec_to_samp(threshold_window_dur, sr))
    thr_win, _ = th.time_dependent_threshold(cc, window, n_dev, overlap=overlap,
                                             white_noise=white_noise)
    cap = max_cc_threshold * weights.reshape(weights.shape[0], -1).sum(axis=1)   # :629
    cand = th.extract_candidates(cc, thr_win, window, overlap=overlap, row_cap=cap)
    min_iet = int(pp.sec_to_samp(minimum_interevent_time, sr))
    mv = np.asarray(moveouts)
    out = {}
    for t in range(weights.shape[0]):
        mine = cand[cand["row"] == t]
        win = search_window(mv[t].reshape(mv.shape[1], -1), min_iet, step)
        out[t] = merge_candidates(mine["index"], mine["cc"], win)
    return out, cc


def backprojection_detections(features, moveouts, weights_phases, weights_sources, *, sr,
                              minimum_interevent_time, threshold_window_dur=None, n_dev=15.0,
                              overlap=0.75, threshold=None, out_of_bounds="strict", device=None):
    """Backprojection of one day: returns (peak samples, source indices, maxbeam, argmax)."""
    bf = BeamformerGPU(moveouts, weights_sources, device=device)
    beam, arg = bf.run(features, weights_phases, "max", out_of_bounds)
    maxbeam, sources = beam.cpu().numpy(), arg.cpu().numpy()
    bf.close()
    if threshold is None:
        window = int(pp.sec_to_samp(threshold_window_dur, sr))
        threshold = pp.bp_time_dependent_threshold(maxbeam, window, n_dev, overlap=overlap)
    mpd = int(pp.sec_to_samp(minimum_interevent_time, sr))
    peaks, peak_sources = pp.find_beam_detections(maxbeam, sources, threshold, mpd)
    return peaks, peak_sources, maxbeam, sources


def intertemplate_cc(waveforms_arr, weights, max_lag=10, device=None):
    """Pair-wise template similarity: intertp[t, u] = sum_{s,c} w[t][u,s,c] * max_lag CC.

    waveforms_arr (T,S,C,L); weights (T, T, S, C) -- row t holds the channel weights the
    reference builds for template t against every other template (dataset.py:4789-4816).
    For each t the template's own waveform is the "data" and every template trimmed by max_lag
    on both sides is correlated against it at 2*max_lag+1 lags (network_sum=False)."""
    import torch
    wf = np.ascontiguousarray(waveforms_arr, dtype=np.float32)
    T = wf.shape[0]
    trimmed = np.ascontiguousarray(wf[..., max_lag:-max_lag])
    mv0 = np.zeros(wf.shape[:-1], dtype=np.int32)
    mf = MatchedFilterGPU(device=device)
    tp_dev = mf._dev(trimmed, torch.float32)
    out = np.zeros((T, T), dtype=np.float32)
    for t in range(T):
        w = np.asarray(weights[t], dtype=np.float32)
        keep = np.flatnonzero((w != 0).reshape(T, -1).sum(axis=1) > 0)
        if keep.size == 0:
            continue
        mf.set_data(wf[t])
        cc = mf.run(tp_dev[torch.as_tensor(keep, device=tp_dev.device)], mv0[keep], w[keep], 1,
                    network_sum=False)                       # (n_keep, 2*max_lag+1, S, C)
        best = cc.max(dim=1).values.cpu().numpy()
        out[t, keep] = np.sum(w[keep] * best, axis=(-1, -2))
    return (out + out.T) / 2.0
