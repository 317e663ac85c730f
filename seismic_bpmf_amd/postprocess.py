"""Host-side steps right before / after the two hot paths (SURVEY.md section 8a rows
MF-4, MF-5, MF-6, BP-3, BP-4, BP-6), restated from the reference so that detection *indices*
come out identical.  Every function is pinned against golden vectors produced by the reference
itself (tests/golden/, tests/test_postprocess.py).

These are small integer / NumPy routines in the reference too (Python loops in
BPMF/similarity_search.py and BPMF/template_search.py); the heavy part of the post-CC step, the
RMS sliding threshold of BPMF/libc.c:516-673, has a device version in threshold.py.
"""
import numpy as np


# ----------------------------------------------------------------- time conversion ---
def sec_to_samp(t, sr, epsilon=0.2):
    """Seconds -> samples, BPMF/utils.py:1258-1271: truncate |t*sr| + epsilon, restore the sign."""
    t = np.asarray(t, dtype=np.float64)
    return np.int64(np.sign(t) * np.int64(np.abs(t * sr) + epsilon))


def moveouts_to_samples(travel_times_sec, sr, relative_to_first=True):
    """(K,S,P) travel times -> integer moveout table (BP-3).

    BPMF/template_search.py:145-168 (sec_to_samp) and :212-214 (minus the per-source minimum over
    stations and phases).  Returns (moveouts int32, moveout_to_tt int64 per source) -- the second is
    what find_detections adds back to the origin time (:635-639)."""
    tau = sec_to_samp(travel_times_sec, sr)
    first = np.zeros(tau.shape[0], dtype=np.int64)
    if relative_to_first:
        first = tau.reshape(tau.shape[0], -1).min(axis=1)
        tau = tau - first[:, None, None]
    return tau.astype(np.int32), first


# ------------------------------------------------------------------- input conditioning ---
def normalize_data(data_arr):
    """MatchedFilter.set_data conditioning, BPMF/similarity_search.py:181-185: each channel is
    divided by its standard deviation (channels with zero std are left untouched)."""
    d = np.array(data_arr, dtype=np.float32, copy=True)
    std = d.std(axis=-1, keepdims=True)
    std[std == 0.0] = 1.0
    return d / std


def normalize_weights(weights):
    """Sum of channel weights = 1 per template, BPMF/similarity_search.py:469-472."""
    w = np.array(weights, dtype=np.float32, copy=True)
    norm = w.sum(axis=(1, 2), keepdims=True)
    norm[norm == 0.0] = 1.0
    return w / norm


def normalize_templates(waveforms, method="rms"):
    """TemplateGroup.normalize, BPMF/dataset.py:4152-4166: every (template, station, channel)
    window divided by its standard deviation ("rms") or its peak amplitude ("max"); all-zero
    windows stay zero."""
    w = np.array(waveforms, dtype=np.float32, copy=True)
    if method == "rms":
        norm = np.std(w, axis=-1, keepdims=True)
    elif method == "max":
        norm = np.max(np.abs(w), axis=-1, keepdims=True)
    else:
        raise ValueError("method must be 'rms' or 'max'")
    norm[norm == 0.0] = 1.0
    w /= norm
    return w


def network_to_template_map(waveforms, selected_stations=None):
    """TemplateGroup.set_network_to_template_map, BPMF/dataset.py:4977-5008: channel (t, s, c) is
    used when (1) its samples do not sum to 0 (:5001) AND (2) station s is one of the stations
    selected on template t (`tp.stations`, :5003-5007 -- what n_closest_stations /
    n_best_SNR_stations leave on a template).

    `selected_stations`: a (T, S) boolean array, or one sequence of network station indexes per
    template (the reference's `network.station_indexes.loc[tp.stations]`); None = every station of
    every template is selected (templates that carry the whole network)."""
    present = ~(np.sum(np.asarray(waveforms), axis=-1) == 0.0)
    if selected_stations is None:
        return present
    T, S = present.shape[:2]
    if isinstance(selected_stations, np.ndarray) and selected_stations.dtype == bool:
        sel = selected_stations
        if sel.shape != (T, S):
            raise ValueError(f"selected_stations must have shape {(T, S)}, got {sel.shape}")
    else:
        if len(selected_stations) != T:
            raise ValueError("selected_stations needs one entry per template")
        sel = np.zeros((T, S), dtype=bool)
        for t, idx in enumerate(selected_stations):
            sel[t, np.asarray(idx, dtype=np.int64)] = True
    return present & sel[:, :, None]


def station_density_weights(interstation_distances, cutoff_dist=None, lower_percentile=0.0,
                            upper_percentile=100.0):
    """Per-station weights that balance station density (BPMF/template_search.py:898-949, same
    text at similarity_search.py:371-421): w_s = 1 / sum_j exp(-D_sj^2 / cutoff^2), cutoff = median
    non-zero inter-station distance unless given, optionally clipped to percentiles.  Float64 sums
    stored as float32, like the reference."""
    D = np.asarray(interstation_distances, dtype=np.float64)
    if cutoff_dist is None:
        cutoff_dist = np.median(D[D != 0.0])
    w = np.zeros(D.shape[0], dtype=np.float32)
    for s in range(D.shape[0]):
        w[s] = 1.0 / np.sum(np.exp(-(D[s] ** 2) / cutoff_dist ** 2))
    if lower_percentile > 0.0:
        w = np.clip(w, np.percentile(w, lower_percentile), w.max())
    if upper_percentile < 100.0:
        w = np.clip(w, w.min(), np.percentile(w, upper_percentile))
    return w


def _kth_smallest(mv, k):
    """Largest of the k smallest values of every row (the reference's np.partition cut-off)."""
    return np.partition(mv, k - 1, axis=1)[:, :k].max(axis=1, keepdims=True)


def weights_sources_closest(moveouts, n_closest, online=None):
    """Source weights of BP-4, BPMF/template_search.py:779-798: 1 for the stations whose
    first-phase moveout is not larger than the `n_closest`-th smallest one *among the operational
    stations* of each source (ties at the cut-off included), 0 for offline stations.  With
    `n_closest` 0 or >= the number of stations every operational station keeps weight 1."""
    first = np.asarray(moveouts)[:, :, 0]
    K, S = first.shape
    online = np.ones(S, dtype=bool) if online is None else np.asarray(online, dtype=bool)
    w = np.ones((K, S), dtype=np.float32)
    n = min(int(online.sum()), int(n_closest))
    if 0 < n < S:
        w[first > _kth_smallest(first[:, online], n)] = 0.0
    w[:, ~online] = 0.0
    return w


def weights_sources_max_moveout(moveouts, max_moveout, online=None):
    """BPMF/template_search.py:800-814: 1 where the earliest phase of a station arrives less than
    `max_moveout` samples after the source's first arrival, 0 for offline stations."""
    earliest = np.min(np.asarray(moveouts), axis=-1)
    w = np.zeros(earliest.shape, dtype=np.float32)
    w[earliest < max_moveout] = 1.0
    if online is not None:
        w[:, ~np.asarray(online, dtype=bool)] = 0.0
    return w


def set_weights_sources(moveouts, method="closest_stations", n_min_stations=0, normalize=False,
                        online=None, density_weights=None, **kwargs):
    """Beamformer.set_weights_sources (BP-4), BPMF/template_search.py:816-895.

    `online` replaces `data.availability_per_sta`; `density_weights` (S,) replaces the
    `weight_station_density=True` branch (build it with station_density_weights)."""
    if method == "closest_stations":
        if kwargs.get("num_closest_stations") is None:
            raise TypeError("method 'closest_stations' needs num_closest_stations")
        w = weights_sources_closest(moveouts, kwargs["num_closest_stations"], online)
    elif method == "max_moveout":
        if kwargs.get("max_moveout") is None:
            raise TypeError("method 'max_moveout' needs max_moveout")
        w = weights_sources_max_moveout(moveouts, kwargs["max_moveout"], online)
    else:
        raise ValueError("method must be 'closest_stations' or 'max_moveout'")
    if n_min_stations > 0:
        w[np.sum(w > 0.0, axis=-1) < n_min_stations, :] = 0.0
    if density_weights is not None:
        w *= np.asarray(density_weights)[None, :]
    if normalize:
        norm = np.sum(w, axis=1, keepdims=True)
        norm[norm == 0.0] = 1.0
        w /= norm
    return w


def weights_channels_simple(present, min_channels=6, min_stations=3):
    """MatchedFilter._weights_channels_simple, BPMF/similarity_search.py:288-296: weight 1 on
    every present template channel; templates with fewer than `min_channels` channels or
    `min_stations` stations get all-zero weights."""
    w = np.float32(present)
    too_few = (np.sum(w != 0.0, axis=(1, 2)) < min_channels) | \
              (np.sum(np.sum(w, axis=2) > 0.0, axis=1) < min_stations)
    w[too_few] = 0.0
    return w


def _operational(availability, data_channels_ok):
    ok = np.asarray(availability, dtype=bool)
    if data_channels_ok is not None:
        ok = np.logical_and(ok, np.asarray(data_channels_ok, dtype=bool)[None, :, :])
    return ok


def weights_channels_closest(moveouts, availability, num_closest_stations, data_channels_ok=None):
    """MatchedFilter._weights_channels_closest, BPMF/similarity_search.py:298-332.  Stations
    without any operational channel are pushed to the end of the ranking; the cut-off is the
    `num_closest_stations`-th smallest first-phase moveout; offline channels get 0."""
    mv_all = np.asarray(moveouts)
    T, S = mv_all.shape[:2]
    ok = _operational(availability, data_channels_ok)
    w = np.ones(ok.shape, dtype=np.float32)
    first = np.array(mv_all[..., 0], copy=True)
    first[~np.any(ok, axis=-1)] = np.iinfo(np.int32).max
    n = min(S, int(num_closest_stations))
    if 0 < n < S:
        w[mv_all[:, :, 0] > _kth_smallest(first, n), :] = 0.0
    w[~ok] = 0.0
    return w


def weights_channels_max_moveout(moveouts, availability, max_moveout_sec, sr, n_min_stations=0,
                                 max_moveout2_sec=None, data_channels_ok=None):
    """MatchedFilter._weights_channels_max_moveout, BPMF/similarity_search.py:334-367 (including
    its fall-back: when fewer than `n_min_stations` (template, station) pairs qualify in total,
    the second radius is applied without the availability mask)."""
    ok = _operational(availability, data_channels_ok)
    w = np.zeros(ok.shape, dtype=np.float32)
    earliest = np.min(np.asarray(moveouts), axis=-1)
    valid = (earliest < int(max_moveout_sec * sr)) & np.any(ok, axis=-1)
    if np.sum(valid) < n_min_stations and max_moveout2_sec is not None:
        valid = earliest < int(max_moveout2_sec * sr)
    w[valid, :] = 1.0
    w[~ok] = 0.0
    return w


def set_weights_channels(method="simple", n_min_stations=0, normalize=True, density_weights=None,
                         *, present=None, moveouts=None, availability=None, sr=None,
                         min_channels=6, min_stations=3, data_channels_ok=None, **kwargs):
    """MatchedFilter.set_weights_channels (MF-6), BPMF/similarity_search.py:423-474.  The
    template-group attributes the reference reads are passed explicitly: `present` =
    network_to_template_map, `moveouts` = moveouts_arr (T,S,P), `availability` =
    availability_arr (T,S,C), `sr` = template sampling rate."""
    if method == "simple":
        w = weights_channels_simple(present, min_channels, min_stations)
    elif method == "closest_stations":
        if kwargs.get("num_closest_stations") is None:
            raise TypeError("method 'closest_stations' needs num_closest_stations")
        w = weights_channels_closest(moveouts, availability, kwargs["num_closest_stations"],
                                     data_channels_ok)
    elif method == "max_moveout":
        if kwargs.get("max_moveout_sec") is None:
            raise TypeError("method 'max_moveout' needs max_moveout_sec")
        w = weights_channels_max_moveout(moveouts, availability, kwargs["max_moveout_sec"], sr,
                                         n_min_stations, kwargs.get("max_moveout2_sec"),
                                         data_channels_ok)
    else:
        raise ValueError("method must be 'simple', 'closest_stations' or 'max_moveout'")
    if n_min_stations > 0:
        w[np.sum(np.any(w > 0.0, axis=-1), axis=1) < n_min_stations, :] = 0.0
    if density_weights is not None:
        w *= np.asarray(density_weights)[None, :, None]
    if normalize:
        norm = np.sum(w, axis=(1, 2), keepdims=True)
        norm[norm == 0.0] = 1.0
        w /= norm
    return w


# ------------------------------------------------------------------------ peak picking ---
def detect_peaks(x, mpd=1):
    """Indices of local maxima of `x` at least `mpd` samples apart (tallest first wins).

    Restates the path of BPMF/utils.py:2203-2354 that BPMF uses (edge="rising", no height or
    prominence threshold, kpsh=False): a peak is a sample strictly above its left neighbour and
    not below its right one; the first and last sample never qualify; peaks are then visited in
    decreasing height and every other peak within +-mpd of a kept one is dropped.
    """
    x = np.atleast_1d(np.asarray(x)).astype(np.float64)
    if x.size < 3:
        return np.array([], dtype=int)
    dx = np.diff(x)
    nan = np.flatnonzero(np.isnan(x))
    if nan.size:
        x = x.copy()
        x[nan] = np.inf
        dx[np.isnan(dx)] = np.inf
    rising = np.flatnonzero((np.append(dx, 0.0) <= 0) & (np.insert(dx, 0, 0.0) > 0))
    ind = np.unique(rising)
    if ind.size and nan.size:
        bad = np.unique(np.concatenate((nan, nan - 1, nan + 1)))
        ind = ind[~np.isin(ind, bad)]
    if ind.size and ind[0] == 0:
        ind = ind[1:]
    if ind.size and ind[-1] == x.size - 1:
        ind = ind[:-1]
    if ind.size and mpd > 1:
        rank = _tallest_first(x[ind])
        ind = ind[_suppress(ind, rank, mpd)]
    return ind


def _tallest_first(height):
    """Visiting order of the suppression, the reference's own expression
    `np.argsort(x[ind])[::-1]` (BPMF/utils.py:2335).  NumPy's default sort is not stable (for
    float64 a SIMD sort): the order of EXACTLY equal heights depends on the whole array, so it
    can only be reproduced by sorting the same full list of local maxima -- which is what
    detect_peaks does, and what the device path falls back to when two equal peaks closer than mpd
    exist among its candidates (has_close_ties)."""
    return np.argsort(np.asarray(height))[::-1]


def has_close_ties(index, height, mpd):
    """True if two peaks of exactly equal height lie within mpd samples of each other -- the only
    case in which the visiting order of equal heights changes which peaks survive."""
    index = np.asarray(index, dtype=np.int64)
    height = np.asarray(height)
    if index.size < 2:
        return False
    order = np.lexsort((index, height))
    h, i = height[order], index[order]
    return bool(np.any((h[1:] == h[:-1]) & (i[1:] - i[:-1] <= mpd)))


def _suppress(ind, rank, mpd):
    """Mask of the peaks `ind` (ascending) that survive the tallest-first +-mpd suppression of
    BPMF/utils.py:2334-2345, visited in the order `rank` (indices into `ind`).  The reference's
    loop is O(peaks^2) in NumPy -- hours for a day of beam; the library's host routine walks the
    neighbours of each kept peak instead (same order, same result)."""
    ind = np.ascontiguousarray(ind, dtype=np.int64)
    rank = np.ascontiguousarray(rank, dtype=np.int64)
    from . import _lib
    lib = _lib.lib()                              # raises if the library has not been built
    keep = np.empty(ind.size, dtype=np.uint8)
    import ctypes as C
    rc = lib.bpmf_suppress_peaks(ind.ctypes.data_as(C.POINTER(C.c_int64)),
                                 rank.ctypes.data_as(C.POINTER(C.c_int64)), ind.size, float(mpd),
                                 keep.ctypes.data_as(C.POINTER(C.c_uint8)))
    _lib.check(rc, "bpmf_suppress_peaks")
    return keep.astype(bool)


def snap_and_merge_peaks(peaks, n, mpd, beam_slice):
    """The regrouping lines of Beamformer.find_detections, BPMF/template_search.py:612-624: every
    peak, in order, is moved to the largest sample within +-mpd/2 of its CURRENT position (first
    occurrence on ties, np.argmax), together with every other peak that currently sits on the same
    sample; duplicates are merged.  `beam_slice(i0, i1)` returns maxbeam[i0:i1] -- the only access
    to the series, so the same code runs on a host array and on windows fetched from the device."""
    peaks = np.array(peaks, copy=True)
    for q in range(peaks.size):
        lo = max(0, peaks[q] - mpd / 2)
        hi = min(peaks[q] + mpd / 2, n)
        win = np.arange(lo, hi).astype(np.int32)   # same float->int truncation as the reference
        seg = np.asarray(beam_slice(int(win[0]), int(win[-1]) + 1))
        snapped = np.argmax(seg[win - win[0]]) + win[0]
        peaks[peaks == peaks[q]] = snapped
    return np.unique(peaks)


def find_beam_detections(maxbeam, maxbeam_sources, threshold, mpd):
    """Peak logic of Beamformer.find_detections (BP-6), BPMF/template_search.py:604-627.

    Peaks of `maxbeam` at least `mpd` samples apart and above `threshold`, each snapped to the
    largest sample within +-mpd/2, duplicates merged.  Returns (peak sample indices, their source
    indices) -- the "detection sample indices" that must match the reference exactly."""
    maxbeam = np.asarray(maxbeam)
    n = maxbeam.size
    threshold = np.broadcast_to(np.asarray(threshold), (n,))
    peaks = detect_peaks(maxbeam, mpd=mpd)
    peaks = peaks[maxbeam[peaks] > threshold[peaks]]
    peaks = snap_and_merge_peaks(peaks, n, mpd, lambda i0, i1: maxbeam[i0:i1])
    return peaks, np.asarray(maxbeam_sources)[peaks]


def find_beam_detections_from_candidates(index, height, threshold_at, mpd, n, beam_slice):
    """find_beam_detections for a series that is not on the host: `index` (ascending) / `height`
    list every rising-edge local maximum above a floor that no threshold value undercuts (the
    device extraction, csrc/bp_detect.hip); `threshold_at(samples)` evaluates the threshold at a few
    samples.  The tallest-first suppression of BPMF/utils.py:2334-2345 runs on this list: the peaks
    it lacks are lower than every peak above the threshold, so they can neither remove one of those
    nor tie with one, and the survivors above the threshold are the reference's.  Exactly equal
    heights closer than mpd are the one exception (has_close_ties): the caller then passes the list
    of ALL local maxima, on which the order is the reference's (_tallest_first)."""
    index = np.asarray(index, dtype=np.int64)
    height = np.asarray(height).astype(np.float64)
    if index.size and mpd > 1:
        rank = _tallest_first(height)
        keep = _suppress(index, rank, mpd)
        index, height = index[keep], height[keep]
    above = height > np.asarray(threshold_at(index), dtype=np.float64)
    return snap_and_merge_peaks(index[above], n, mpd, beam_slice)


def bp_threshold_nodes(n, window, overlap, median, mad, n_dev):
    """Window centres and node values of template_search.time_dependent_threshold
    (BPMF/template_search.py:1452-1487) from the per-window medians / MADs: `median`, `mad` are
    float32 arrays of n_windows + 2 entries whose entries 1 .. n_windows are filled; the end fills
    are applied here.  Returns (centre float32, threshold float32), both n_windows + 2 long; the
    threshold at sample t is interp_threshold(t, centre, threshold), the end values outside."""
    shift = int((1.0 - overlap) * window)
    n_windows = int((n - window) // shift) + 1
    med = np.array(median, dtype=np.float32, copy=True)
    dev = np.array(mad, dtype=np.float32, copy=True)
    centre = np.zeros(n_windows + 2, dtype=np.float32)
    for q in range(1, n_windows + 1):
        i1 = q * shift
        centre[q] = (i1 + min(n, i1 + window)) / 2.0
    med[0], dev[0], centre[0] = med[1], dev[1], 0.0
    med[-1], dev[-1], centre[-1] = med[-2], dev[-2], n
    return centre, med + n_dev * dev


def interp_threshold(samples, centre, nodes):
    """The reference's interp1d(kind="slinear", bounds_error=False, fill_value=(first, last)) at
    `samples`, in float64 and in SciPy's own operation order, so that the threshold is bit-identical
    to the reference's (checked live against it in tests/test_reference_live.py): an order-1
    B-spline through the nodes, evaluated by de Boor's recursion -- on [xa, xb) with w = 1 / (xb - xa):
    ya * (w * (xb - x)) + yb * (w * (x - xa)).  (np.interp's ya + slope * (x - xa) differs from it in
    the last bit at a third of the samples.)"""
    x = np.asarray(samples, dtype=np.float64)
    t = np.asarray(centre, dtype=np.float64)
    y = np.asarray(nodes, dtype=np.float64)
    if t.size < 2 or np.any(np.diff(t) <= 0):
        # duplicate knots (the empty last window of bp_threshold_nodes): SciPy refuses them; keep the
        # piecewise-linear definition there
        return np.interp(x, t, y, left=nodes[0], right=nodes[-1])
    i = np.clip(np.searchsorted(t, x, side="right") - 1, 0, t.size - 2)
    xa, xb = t[i], t[i + 1]
    w = 1.0 / (xb - xa)
    out = 0.0 + y[i] * (w * (xb - x))
    out = out + y[i + 1] * (w * (x - xa))
    out = np.where(x < t[0], np.float64(nodes[0]), out)
    return np.where(x > t[-1], np.float64(nodes[-1]), out)


def select_cc_indexes(cc_t, threshold, search_win, *, step, sr, data_duration_sec,
                      n_dev_threshold, min_freq_hz, data_buffer_sec, threshold_type="rms",
                      remove_edges=True, anomalous_cdf_at_mean_plus_1sig=0.50,
                      window_for_validation_Tmax=100.0):
    """Peak list of one CC series (MF-4), BPMF/similarity_search.py:187-286.

    Explicit keyword arguments replace the reference's module-level config (`cfg.N_DEV_MF_THRESHOLD`,
    `cfg.MIN_FREQ_HZ`, `cfg.DATA_BUFFER_SEC`) and `self.data.sr / duration / step`.
    Returns CC indices (multiply by `step` for data samples)."""
    cc_t = np.asarray(cc_t)
    threshold = np.broadcast_to(np.asarray(threshold), cc_t.shape)
    idx = list(np.flatnonzero(cc_t > threshold))
    one_sigma = threshold / n_dev_threshold
    if threshold_type == "mad":
        one_sigma = one_sigma * 1.48
    # sequential pair-wise merge with the reference's own bookkeeping: after each removal the
    # comparison pointer stays on the survivor (:240-251)
    removed = 0
    for q in range(1, len(idx)):
        cur, prev = idx[q - removed], idx[q - removed - 1]
        if cur - prev < search_win:
            idx.remove(prev if cc_t[cur] > cc_t[prev] else cur)
            removed += 1
    idx = np.asarray(idx, dtype=np.int64)

    if anomalous_cdf_at_mean_plus_1sig > 0.0 and idx.size:
        win = int(1.0 / min_freq_hz * window_for_validation_Tmax)
        keep = np.ones(idx.size, dtype=bool)
        for q, i in enumerate(idx):
            i0 = max(0, i - win // 2)
            i1 = i0 + win
            if i1 >= cc_t.size:
                i1 = cc_t.size - 1
                i0 = i1 - win
            seg = cc_t[i0:i1]
            left, right = seg[: win // 2], seg[win // 2:]
            frac = min(np.sum(left < one_sigma[i]) / float(len(left)),
                       np.sum(right < one_sigma[i]) / float(len(right)))
            keep[q] = frac >= anomalous_cdf_at_mean_plus_1sig
        idx = idx[keep]

    if remove_edges:
        samples = idx * step
        idx = idx[samples >= sec_to_samp(data_buffer_sec, sr)]
        samples = idx * step
        idx = idx[samples < sec_to_samp(data_duration_sec + data_buffer_sec, sr)]
    return idx


def time_dependent_threshold_mad(time_series, sliding_window, n_dev, overlap=0.66,
                                 white_noise=None):
    """MAD variant of the MF detection threshold (MF-5), BPMF/similarity_search.py:1079-1113."""
    x = np.array(time_series, copy=True)
    n = x.size
    half = sliding_window // 2
    shift = int((1.0 - overlap) * sliding_window)
    zeros = x == 0.0
    n_zeros = int(zeros.sum())
    if white_noise is None:
        white_noise = np.random.normal(size=n_zeros).astype("float32")
    centre0 = np.median(x[~zeros])
    dev0 = np.median(np.abs(x[~zeros] - centre0))
    x[zeros] = white_noise[:n_zeros] * dev0 + centre0
    wins = np.lib.stride_tricks.sliding_window_view(x, sliding_window)[::shift, :]
    centre = np.median(wins, axis=-1)
    dev = np.median(np.abs(wins - centre[:, None]), axis=-1)
    thr = centre + n_dev * dev
    thr[1:] = np.maximum(thr[:-1], thr[1:])
    thr[:-1] = np.maximum(thr[:-1], thr[1:])
    where = np.arange(half, n - (sliding_window - half)) // shift
    where[where >= thr.size] = thr.size - 1
    thr = thr[where]
    return np.hstack((thr[0] * np.ones(half, dtype=np.float32), thr,
                      thr[-1] * np.ones(sliding_window - half, dtype=np.float32)))


def bp_time_dependent_threshold(network_response, window, n_dev, overlap=0.75):
    """Detection threshold on the maximum beam (BP side of row BP-6),
    BPMF/template_search.py:1418-1487: median + n_dev * MAD over sliding windows (float32
    per-window values), the first/last window repeated at the ends, linear interpolation
    between window centres."""
    x = np.asarray(network_response)
    n = x.size
    shift = int((1.0 - overlap) * window)
    n_windows = int((n - window) // shift) + 1
    med = np.zeros(n_windows + 2, dtype=np.float32)
    mad = np.zeros(n_windows + 2, dtype=np.float32)
    for q in range(1, n_windows + 1):
        i1 = q * shift
        seg = x[i1:min(n, i1 + window)]
        m = np.median(seg)
        med[q] = m
        mad[q] = np.median(np.abs(seg - m))
    centre, nodes = bp_threshold_nodes(n, window, overlap, med, mad, n_dev)
    return interp_threshold(np.arange(n, dtype=np.float64), centre, nodes)


def excess_kurtosis_f32(a):
    """scipy.stats.kurtosis(a) (Fisher, biased) of a 1-D float32 series exactly as SciPy evaluates it
    on a float32 array -- the `sanity_check` of MatchedFilter._find_detections_t
    (BPMF/similarity_search.py:633-642): float32 mean (NumPy's pairwise sum / count), float32
    powers of the zero-mean series, float32 means of those, m4 / m2**2 - 3; NaN when
    m2 <= (eps * mean)**2.  Restated without SciPy so that the device kernel
    (csrc/stats.hip, bpmf_row_kurtosis_dev) has a host mirror; tests compare both with SciPy."""
    a = np.asarray(a, dtype=np.float32).reshape(-1)
    mean = np.mean(a, keepdims=True)
    d = a - mean
    s2 = d ** 2
    m2 = np.mean(s2)
    m4 = np.mean(s2 ** 2)
    return kurtosis_from_moments_f32(mean[0], m2, m4)


def kurtosis_from_moments_f32(mean, m2, m4):
    """The last expression of scipy.stats.kurtosis on ONE float32 series, `m4 / m2**2.0 - 3` (NaN when
    m2 <= (eps * mean)**2), on NumPy float32 SCALARS as SciPy evaluates it there.  Not a detail: a NumPy scalar
    `**` is the C library's powf, which glibc does not always round like the exact square an array `**` gives
    (one random row in 2500 ends 2 ulp apart), and BPMF passes one series (similarity_search.py:640).  The device
    hands out (mean, m2, m4) per row (bpmf_row_kurtosis_parts_dev) and the host finishes here, with whatever
    NumPy / libm the reference itself would run on."""
    with np.errstate(all="ignore"):
        return _kurtosis_from_moments(np.float32(mean), np.float32(m2), np.float32(m4))


_F32_EPS = np.finfo(np.float32).eps
_F32_NAN = np.float32(np.nan)
_F32_THREE = np.float32(3)


def _kurtosis_from_moments(mean, m2, m4):
    """(np.float32 scalars in, warnings handled by the caller: a row of a loop over a CC matrix)"""
    if m2 <= (_F32_EPS * mean) ** 2:
        return _F32_NAN
    return np.float32(m4 / m2 ** 2.0) - _F32_THREE


def kurtosis_from_moments_rows(parts):
    """kurtosis_from_moments_f32 for every row of a (rows, 3) float32 array of (mean, m2, m4)."""
    parts = np.asarray(parts, dtype=np.float32)
    with np.errstate(all="ignore"):
        return np.array([_kurtosis_from_moments(p[0], p[1], p[2]) for p in parts], dtype=np.float32)


# ---------------------------------------------------------------- event relocation ---
def likelihood(beam_column):
    """Beamformer._likelihood (BPMF/template_search.py:498-506): the beam over the grid at the time of
    maximum focusing, rescaled to [0, 1] -- (x - min) / (max - min), clipped."""
    x = np.asarray(beam_column)
    like = (x - x.min()) / (x.max() - x.min())
    return np.clip(like, a_min=0.0, a_max=1.0)


def gibbs_weights(maxbeam, effective_kT=0.33):
    """Likelihood of the "temporal" uncertainty method of Event.relocate_beam
    (BPMF/dataset.py:2224-2231): exp(-(max - maxbeam) / effective_kT)."""
    mb = np.asarray(maxbeam)
    return np.exp(-(mb.max() - mb) / effective_kT)


WGS84_A = 6378137.0                 # semi-major axis, m
WGS84_F = 1.0 / 298.257223563       # flattening


def geodesic_distance_m(lon0, lat0, lon, lat, max_iter=200, tol=1e-12, nonconverged="raise"):
    """Length in metres of the WGS84 geodesics from (lon0, lat0) to every (lon[i], lat[i]), degrees in --
    what ``cartopy.geodesic.Geodesic().inverse(...)[:, 0]`` returns in
    Beamformer._compute_location_uncertainty (BPMF/template_search.py:1310-1320; cartopy wraps
    GeographicLib on the same ellipsoid).  Vincenty's inverse iteration, vectorised: within a fraction of a
    millimetre of GeographicLib for every pair that is not nearly antipodal.  Vincenty's iteration does not
    converge for pairs within about half a degree of antipodal, where GeographicLib still returns a length:
    `nonconverged="raise"` (default) raises ValueError rather than return a wrong length;
    `nonconverged="antipodal"` gives those pairs pi (a + b) / 2 = 20 020.7 km -- every geodesic between nearly
    antipodal points is between pi b = 20 003.9 and pi a = 20 037.5 km long, so the value is within 17 km
    (0.09 %) of the true length -- and leaves every other pair exact.
    Host code (float64 NumPy): a domain has a few thousand sources."""
    if nonconverged not in ("raise", "antipodal"):
        raise ValueError("nonconverged must be 'raise' or 'antipodal'")
    lon = np.atleast_1d(np.asarray(lon, dtype=np.float64))
    lat = np.atleast_1d(np.asarray(lat, dtype=np.float64))
    lon, lat = np.broadcast_arrays(lon, lat)
    f, a = WGS84_F, WGS84_A
    b = (1.0 - f) * a
    u1 = np.arctan((1.0 - f) * np.tan(np.deg2rad(float(lat0))))
    u2 = np.arctan((1.0 - f) * np.tan(np.deg2rad(lat)))
    su1, cu1, su2, cu2 = np.sin(u1), np.cos(u1), np.sin(u2), np.cos(u2)
    big_l = np.deg2rad((lon - float(lon0) + 180.0) % 360.0 - 180.0)     # longitude difference in [-180, 180)
    def at(lam):
        sl, cl = np.sin(lam), np.cos(lam)
        sin_sig = np.hypot(cu2 * sl, cu1 * su2 - su1 * cu2 * cl)
        cos_sig = su1 * su2 + cu1 * cu2 * cl
        sig = np.arctan2(sin_sig, cos_sig)
        sin_al = np.where(sin_sig > 0.0, cu1 * cu2 * sl / np.where(sin_sig > 0.0, sin_sig, 1.0), 0.0)
        cos2_al = 1.0 - sin_al * sin_al
        cos_2sm = np.where(cos2_al > 0.0, cos_sig - 2.0 * su1 * su2 / np.where(cos2_al > 0.0, cos2_al, 1.0), 0.0)
        return sin_sig, cos_sig, sig, sin_al, cos2_al, cos_2sm

    lam = big_l.copy()
    done = np.zeros(lam.shape, dtype=bool)
    # (a point that has converged is left alone, and the lengths come from one evaluation at the final
    # longitudes: the result of a point does not depend on what else is in the call)
    for _ in range(max_iter):
        sin_sig, cos_sig, sig, sin_al, cos2_al, cos_2sm = at(lam)
        c = f / 16.0 * cos2_al * (4.0 + f * (4.0 - 3.0 * cos2_al))
        new = big_l + (1.0 - c) * f * sin_al * (
            sig + c * sin_sig * (cos_2sm + c * cos_sig * (-1.0 + 2.0 * cos_2sm * cos_2sm)))
        conv = np.abs(new - lam) < tol
        lam = np.where(done, lam, new)
        done = done | conv
        if done.all():
            break
    if not done.all() and nonconverged == "raise":
        raise ValueError("geodesic_distance_m: no convergence (nearly antipodal points)")
    sin_sig, cos_sig, sig, sin_al, cos2_al, cos_2sm = at(lam)
    usq = cos2_al * (a * a - b * b) / (b * b)
    big_a = 1.0 + usq / 16384.0 * (4096.0 + usq * (-768.0 + usq * (320.0 - 175.0 * usq)))
    big_b = usq / 1024.0 * (256.0 + usq * (-128.0 + usq * (74.0 - 47.0 * usq)))
    d_sig = big_b * sin_sig * (cos_2sm + big_b / 4.0 * (
        cos_sig * (-1.0 + 2.0 * cos_2sm * cos_2sm)
        - big_b / 6.0 * cos_2sm * (-3.0 + 4.0 * sin_sig * sin_sig) * (-3.0 + 4.0 * cos_2sm * cos_2sm)))
    return np.where(done, b * big_a * (sig - d_sig), np.pi * (a + b) / 2.0)


def compute_location_uncertainty(event_longitude, event_latitude, event_depth, likelihood,
                                 source_longitude, source_latitude, source_depth):
    """Beamformer._compute_location_uncertainty (BPMF/template_search.py:1269-1333) without cartopy:
    (hunc, vunc) in km -- the likelihood-weighted mean geodesic distance of the domain's sources to the
    event and their mean absolute depth difference.  `source_*`: the coordinates of the sources OF THE
    DOMAIN (the reference indexes its grid with `domain`), `likelihood`: theirs.
    One limitation against cartopy / GeographicLib: a source within about half a degree of the event's
    ANTIPODE (never the case for a regional source grid) gets the length pi (a + b) / 2, within 17 km (0.09 %)
    of its true geodesic distance, instead of failing the whole relocation (geodesic_distance_m,
    nonconverged="antipodal"); every other distance is exact to a fraction of a millimetre."""
    d_km = geodesic_distance_m(event_longitude, event_latitude, source_longitude, source_latitude,
                               nonconverged="antipodal") / 1000.0
    depth_diff = np.abs(float(event_depth) - np.asarray(source_depth, dtype=np.float64))
    return location_uncertainty(likelihood, d_km, depth_diff)


def location_uncertainty(likelihood_domain, distances_km, depth_diff_km):
    """Beamformer._compute_location_uncertainty (BPMF/template_search.py:1269-1333) behind its geodesic
    call: likelihood-weighted mean epicentral distance and mean absolute depth difference of the
    sources of the domain.  `distances_km`: distance of every source of the domain to the event (the
    reference takes cartopy's Geodesic().inverse(...)[:, 0] / 1000); returns (hunc, vunc) in km."""
    like = np.asarray(likelihood_domain)
    hunc = np.sum(like * np.asarray(distances_km)) / np.sum(like)
    vunc = np.sum(like * np.asarray(depth_diff_km)) / np.sum(like)
    return hunc, vunc
