#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (runs only in the build container).

What is pinned here: the native helpers of BPMF/libc.c and the
NumPy-only Python functions adjacent to the hot paths.  The two hot paths themselves cannot
be pinned -- their arithmetic lives in third-party packages that are not in /root/reference.

How: (1) the reference's own BPMF/libc.c is compiled where it lies by oracle/Makefile
(`make ref` -> oracle/_ref/libc.so) and driven through the reference's own ctypes wrappers
(BPMF/clib.py) -- always with num_threads=1, because its OpenMP loops race; (2) the reference
Python package is imported with stub modules for the dependencies this image lacks
(h5py, obspy, fast_matched_filter, beampower), from a scratch directory that holds a copy of
the tutorial's parameter file, because BPMF reads BPMF_parameters.cfg from the CWD at import.

Only inputs and outputs are stored (data, not source).  Usage: python tests/golden/make_goldens.py
"""
import ctypes as C
import os
import shutil
import subprocess
import sys
import tempfile
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def import_reference():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True,
                   capture_output=True)
    scratch = tempfile.mkdtemp(prefix="bpmf_gold_")
    shutil.copy(os.path.join(REF, "tutorial/notebooks/BPMF_parameters.cfg"), scratch)
    os.chdir(scratch)
    for m in ["h5py", "obspy", "obspy.core", "obspy.signal", "obspy.signal.filter",
              "fast_matched_filter", "beampower", "matplotlib", "matplotlib.pyplot",
              "matplotlib.colors", "matplotlib.cm", "matplotlib.dates", "matplotlib.ticker",
              "mpl_toolkits", "mpl_toolkits.axes_grid1", "mpl_toolkits.axes_grid1.inset_locator",
              "matplotlib.colorbar", "matplotlib.gridspec", "matplotlib.patches"]:
        try:
            __import__(m)
        except Exception:
            sys.modules[m] = MagicMock()
    sys.path.insert(0, REF)
    import BPMF
    from BPMF import clib
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libc.so"))
    f, i = C.POINTER(C.c_float), C.POINTER(C.c_int)
    lib.kurtosis.argtypes = [f, C.c_int, C.c_int, C.c_int, C.c_int, f]
    lib.select_cc_indexes.argtypes = [f, f, C.c_size_t, C.c_size_t, i]
    lib.time_dependent_threshold.argtypes = [f, f, C.c_float, C.c_size_t, C.c_size_t, C.c_size_t,
                                             C.c_int, f]
    for fn in ("find_similar_moveouts", "find_similar_moveouts2"):
        getattr(lib, fn).argtypes = [f] * 5 + [C.c_float] + [C.c_size_t] * 5 + [C.c_int, i]
    clib._libc = lib
    clib.cpu_loaded = True
    return BPMF, clib


def cc_like_series(rng, n, gaps=True):
    """Something shaped like a network CC series: small Gaussian values, a few spikes, zeros."""
    x = (0.02 * rng.standard_normal(n)).astype(np.float32)
    for p in rng.integers(100, n - 100, 12):
        x[p - 2:p + 3] += np.float32(rng.uniform(0.2, 0.8)) * np.array([.3, .7, 1, .7, .3], np.float32)
    if gaps:
        a = int(n * 0.4)
        x[a:a + n // 20] = 0.0
        x[:50] = 0.0
    return x


def relocation_goldens(BPMF):
    """Beamformer._likelihood and Beamformer._compute_location_uncertainty (BPMF/template_search.py:
    498-506, 1269-1333) and the Gibbs weights of Event.relocate_beam (BPMF/dataset.py:2224-2231) on
    seeded beam columns.  cartopy (the geodesic distances of :1311-1321) is absent here: its
    Geodesic.inverse is replaced by a deterministic plane approximation, and the distances it returned
    are stored with the vectors -- what the golden pins is the reference's arithmetic around them."""
    import types
    import pandas as pd
    from BPMF import template_search
    rng = np.random.default_rng(20260929)

    class _Geodesic:
        def inverse(self, point, endpoints):
            point, endpoints = np.asarray(point, float), np.asarray(endpoints, float)
            dx = (endpoints[:, 0] - point[0]) * 111_194.9 * np.cos(np.deg2rad(point[1]))
            dy = (endpoints[:, 1] - point[1]) * 111_194.9
            return np.stack([np.hypot(dx, dy), np.zeros(len(dx)), np.zeros(len(dx))], axis=1)

    sys.modules["cartopy"] = types.ModuleType("cartopy")
    sys.modules["cartopy.geodesic"] = types.ModuleType("cartopy.geodesic")
    sys.modules["cartopy.geodesic"].Geodesic = _Geodesic
    out = {}
    n_cases = 4
    for j in range(n_cases):
        K = (500, 5000, 37, 1200)[j]
        col = np.abs(rng.standard_normal(K)).astype(np.float32) * np.float32((1.0, 30.0, 1e-3, 5.0)[j])
        if j == 3:
            col[::7] = col.max()                     # ties at the maximum
        like = template_search.Beamformer._likelihood(col)
        coords = pd.DataFrame({"longitude": rng.uniform(30.0, 31.0, K), "latitude": rng.uniform(40.0, 41.0, K),
                               "depth": rng.uniform(0.0, 30.0, K)})
        fake = types.SimpleNamespace(source_coordinates=coords)
        k0 = int(col.argmax())
        domain = np.sort(rng.choice(K, size=max(3, K // 3), replace=False))
        ev = (coords["longitude"].iloc[k0], coords["latitude"].iloc[k0], coords["depth"].iloc[k0])
        hunc, vunc = template_search.Beamformer._compute_location_uncertainty(fake, *ev, like[domain], domain)
        dist = _Geodesic().inverse(np.array(ev[:2]), np.hstack((coords["longitude"].values[domain, None],
                                                                coords["latitude"].values[domain, None])))[:, 0] / 1000.0
        # temporal method: Gibbs weights of the max-beam (dataset.py:2224-2231)
        maxbeam = np.abs(rng.standard_normal(3000)).astype(np.float32) * np.float32(2.0)
        gibbs = np.exp(-(maxbeam.max() - maxbeam) / 0.33)
        out.update({f"column_{j}": col, f"likelihood_{j}": like, f"domain_{j}": domain,
                    f"distances_km_{j}": dist, f"depth_diff_{j}": np.abs(ev[2] - coords["depth"].values[domain]),
                    f"hunc_{j}": np.float64(hunc), f"vunc_{j}": np.float64(vunc),
                    f"maxbeam_{j}": maxbeam, f"gibbs_{j}": gibbs})
    np.savez_compressed(os.path.join(HERE, "relocation.npz"), n_cases=n_cases, effective_kT=0.33, **out)


def main():
    BPMF, clib = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "relocation":      # only this file (the others stay untouched)
        relocation_goldens(BPMF)
        print("relocation.npz written")
        return
    from BPMF import similarity_search, template_search, utils
    rng = np.random.default_rng(20260928)
    out = {}

    # ---- libc.c: time_dependent_threshold (rms) ------------------------------------
    cases = []
    for n, window, overlap, gaps in [(200_000, 18_000, 0.0, True), (200_000, 18_000, 0.25, True),
                                     (150_000, 9_000, 0.0, False), (120_000, 12_000, 0.5, True),
                                     (86_400, 4_500, 0.2, False)]:
        shift = int((1.0 - overlap) * window)
        nsw = (n - (window - shift)) // shift
        assert (n - shift - 1) // shift <= nsw - 1, "case would read out of bounds in the reference"
        x = cc_like_series(rng, n, gaps)
        wn = rng.standard_normal(500).astype(np.float32)
        thr = clib.time_dependent_threshold(x, window, 8.0, overlap=overlap, white_noise=wn,
                                            num_threads=1)
        cases.append(dict(x=x, white_noise=wn, window=window, overlap=overlap, num_dev=8.0, thr=thr))
    np.savez_compressed(os.path.join(HERE, "tdt_rms.npz"),
                        **{f"{k}_{j}": np.asarray(v) for j, c in enumerate(cases) for k, v in c.items()},
                        n_cases=len(cases))

    # ---- libc.c: select_cc_indexes ---------------------------------------------------
    cases = []
    for n, win in [(50_000, 500), (50_000, 37), (20_000, 5_000), (3_000, 1)]:
        x = cc_like_series(rng, n, gaps=False)
        thr = (0.1 + 0.02 * rng.random(n)).astype(np.float32)
        sel = clib.select_cc_indexes(x, thr, win)
        cases.append(dict(x=x, thr=thr, win=win, sel=sel))
    x = cc_like_series(rng, 10_000, gaps=False)
    cases.append(dict(x=x, thr=np.float32(0.15) * np.ones(10_000, np.float32), win=200,
                      sel=clib.select_cc_indexes(x, 0.15, 200)))
    np.savez_compressed(os.path.join(HERE, "select_cc_indexes_c.npz"),
                        **{f"{k}_{j}": np.asarray(v) for j, c in enumerate(cases) for k, v in c.items()},
                        n_cases=len(cases))

    # ---- libc.c: kurtosis --------------------------------------------------------------
    sig = rng.standard_normal((2, 3, 3000)).astype(np.float32)
    sig[1, 0, 1000:1400] = 0.0
    sig[0, 2, 500:520] *= 15.0
    np.savez_compressed(os.path.join(HERE, "kurtosis.npz"), signal=sig, W=200,
                        kurto=clib.kurtosis(sig, 200))

    # ---- libc.c: find_similar_moveouts / find_similar_moveouts2 ----------------------------
    nx, ny, nz, S = 12, 10, 5, 9
    lon, lat, dep = np.meshgrid(np.linspace(30.0, 30.6, nx), np.linspace(40.0, 40.5, ny),
                                np.linspace(0, 20, nz), indexing="ij")
    lon, lat, dep = lon.ravel(), lat.ravel(), dep.ravel()
    sta = np.stack([rng.uniform(30.0, 30.6, S), rng.uniform(40.0, 40.5, S)], 1)
    dist = np.sqrt(((lon[:, None] - sta[None, :, 0]) * 85.0) ** 2 +
                   ((lat[:, None] - sta[None, :, 1]) * 111.0) ** 2 + dep[:, None] ** 2)
    mv = (dist / 6.0).astype(np.float32)
    cell_lon = np.linspace(29.99, 30.61, 4).astype(np.float32)
    cell_lat = np.linspace(39.99, 40.51, 3).astype(np.float32)
    res = {}
    for method in ("closest", "smallest"):
        for thr, ndiff in [(0.25, 5), (0.6, 9)]:
            red = clib.find_similar_sources(mv, lon.astype(np.float32), lat.astype(np.float32),
                                            cell_lon, cell_lat, thr, num_threads=1,
                                            num_stations_for_diff=ndiff, method=method)
            res[f"red_{method}_{ndiff}"] = np.asarray(red)
            res[f"thr_{method}_{ndiff}"] = thr
    np.savez_compressed(os.path.join(HERE, "similar_sources.npz"), moveouts=mv,
                        lon=lon.astype(np.float32), lat=lat.astype(np.float32), cell_lon=cell_lon,
                        cell_lat=cell_lat, **res)

    # ---- utils.sec_to_samp, utils._detect_peaks --------------------------------------------
    t = np.concatenate([np.linspace(-30, 30, 601), rng.uniform(-100, 100, 400),
                        np.array([0.0, 0.04, -0.04, 7.96, 7.999999, 12.0 / 25.0])])
    s2s = {f"sr_{sr}": utils.sec_to_samp(t, sr=float(sr)) for sr in (25, 50, 100)}
    peaks = {}
    series = [np.array([0, 1, 0, 2, 0, 3, 0, 2, 0, 1, 0], dtype=np.float64),
              np.abs(rng.standard_normal(5000)) + 5 * (rng.random(5000) > 0.995),
              np.round(np.abs(rng.standard_normal(4000)), 1),  # plateaus and exact ties
              cc_like_series(rng, 20_000, gaps=True).astype(np.float64)]
    for j, x in enumerate(series):
        peaks[f"x_{j}"] = x
        for mpd in (1, 2, 25, 300):
            peaks[f"ind_{j}_{mpd}"] = utils._detect_peaks(x, mpd=mpd)
    np.savez_compressed(os.path.join(HERE, "host_helpers.npz"), t=t, **s2s, **peaks,
                        n_series=len(series))

    # ---- Beamformer.find_detections peak logic (template_search.py:604-627) -----------------
    # dataset.Event needs obspy, so the index logic is replayed here from the reference's
    # own _detect_peaks plus a NumPy transliteration of the regrouping lines.
    det = {}
    for j, (n, mpd) in enumerate([(20_000, 250), (20_000, 40), (5_000, 1000)]):
        mb = (np.abs(rng.standard_normal(n)) + 6 * (rng.random(n) > 0.998) *
              rng.random(n)).astype(np.float32)
        src = rng.integers(0, 5000, n).astype(np.int32)
        thr = np.full(n, 3.0, dtype=np.float32)
        pk = utils._detect_peaks(mb, mpd=mpd)
        pk = pk[mb[pk] > thr[pk]]
        for i in range(len(pk)):
            idx = np.int32(np.arange(max(0, pk[i] - mpd / 2), min(pk[i] + mpd / 2, len(mb))))
            upd = np.where(pk == pk[i])[0]
            pk[upd] = np.argmax(mb[idx]) + idx[0]
        pk = np.unique(pk)
        det.update({f"maxbeam_{j}": mb, f"sources_{j}": src, f"thr_{j}": thr, f"mpd_{j}": mpd,
                    f"peaks_{j}": np.asarray(pk), f"peak_sources_{j}": src[pk]})
    np.savez_compressed(os.path.join(HERE, "bp_find_detections.npz"), n_cases=3, **det)

    # ---- MatchedFilter.select_cc_indexes (Python variant, similarity_search.py:187-286) -----
    class _Data:
        sr = 25.0
        duration = 2000.0
    sel = {}
    for j, (n, win, step, remove_edges, acdf) in enumerate([(80_000, 250, 1, True, 0.0),
                                                             (80_000, 250, 1, True, 0.5),
                                                             (40_000, 60, 2, False, 0.5),
                                                             (30_000, 1000, 1, False, 0.0)]):
        fake = MagicMock()
        fake.data = _Data()
        fake.step = step
        fake.threshold_type = "rms"
        fake.remove_edges = remove_edges
        x = cc_like_series(rng, n, gaps=False)
        thr = (0.16 + 0.0 * x).astype(np.float32)
        idx = similarity_search.MatchedFilter.select_cc_indexes(
            fake, x, thr, win, anomalous_cdf_at_mean_plus_1sig=acdf)
        sel.update({f"x_{j}": x, f"thr_{j}": thr, f"win_{j}": win, f"step_{j}": step,
                    f"remove_edges_{j}": remove_edges, f"acdf_{j}": acdf,
                    f"idx_{j}": np.asarray(idx, dtype=np.int64)})
    np.savez_compressed(os.path.join(HERE, "select_cc_indexes_py.npz"), n_cases=4, sr=25.0,
                        duration=2000.0, min_freq_hz=2.0, n_dev=8.0, data_buffer_sec=500.0, **sel)

    # ---- similarity_search.time_dependent_threshold, 'mad' (NumPy) --------------------------
    x = cc_like_series(rng, 60_000, gaps=True)
    wn = rng.standard_normal(int((x == 0).sum())).astype(np.float32)
    thr = similarity_search.time_dependent_threshold(x, 6000, overlap=0.5, threshold_type="mad",
                                                     white_noise=wn)
    np.savez_compressed(os.path.join(HERE, "tdt_mad.npz"), x=x, white_noise=wn, window=6000,
                        overlap=0.5, n_dev=8.0, thr=np.asarray(thr))
    # ---- template_search.time_dependent_threshold (median/MAD on the max beam) ----------------
    mb = (np.abs(rng.standard_normal(40_000)) + 6 * (rng.random(40_000) > 0.998)).astype(np.float32)
    thr = template_search.time_dependent_threshold(mb, 3000, overlap=0.75, CNR_threshold=15.0)
    np.savez_compressed(os.path.join(HERE, "bp_threshold.npz"), maxbeam=mb, window=3000, overlap=0.75,
                        n_dev=15.0, thr=np.asarray(thr))
    # ---- template_search.saturated_envelopes (feature extraction in front of the beamformer) ----
    # envelope_parallel is the same map over channels through a ProcessPoolExecutor; run serially
    template_search.envelope_parallel = lambda tr: np.float32(
        [template_search.envelope(x_) for x_ in tr.reshape(-1, tr.shape[-1])]).reshape(tr.shape)
    env = {}
    for j, n in enumerate((20_000, 14_999)):
        tr = rng.standard_normal((3, 2, n)).astype(np.float32)
        tr[0, 0] *= 1.0e-3
        tr[0, 1, 5000:5200] = 0.0                 # a gap: the envelope is not zero there
        tr[1, 0] = 0.0                            # dead channel -> dropped
        tr[1, 1] *= 1.0e-13                       # MAD below the anomaly threshold -> dropped
        tr[2, 0, 7000] = 3.0e6                    # spike -> saturates at max_dynamic_range
        tr[2, 1] = np.cumsum(tr[2, 1]) * 0.01     # red noise
        feat, avail = template_search.saturated_envelopes(tr.copy())
        env.update({f"traces_{j}": tr, f"features_{j}": np.asarray(feat, dtype=np.float32),
                    f"availability_{j}": np.asarray(avail, dtype=np.int32),
                    f"envelope_{j}": np.float32([template_search.envelope(x_) for x_ in tr.reshape(-1, n)]).reshape(tr.shape)})
    np.savez_compressed(os.path.join(HERE, "saturated_envelopes.npz"), n_cases=2, **env)
    # ---- BP-4: Beamformer source-weight builders (template_search.py:779-895) ---------------
    # and MF-6: MatchedFilter channel-weight builders / set_data / TemplateGroup.normalize
    # (similarity_search.py:163-185, 288-474; dataset.py:4152-4166, 4996-5001), through fake
    # `self` objects that carry only the attributes those methods read.
    import pandas as pd
    import types

    def density_frame(names, rng_):
        xy = rng_.uniform(0.0, 80.0, (len(names), 2))
        d = np.sqrt(((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1))
        return pd.DataFrame(d, index=names, columns=names)

    wts = {}
    K, S, P = 400, 9, 2
    names = [f"S{j:02d}" for j in range(S)]
    mv_bp = rng.integers(0, 900, (K, S, P)).astype(np.int64)
    mv_bp[:, :, 1] += mv_bp[:, :, 0] // 2
    mv_bp[::7, 3, 0] = mv_bp[::7, 5, 0]                        # exact ties at the cut-off
    mv_bp -= mv_bp.reshape(K, -1).min(axis=1)[:, None, None]
    online = np.ones(S, dtype=bool)
    online[[2, 6]] = False
    dist_bp = density_frame(names, rng)

    class _BF:
        pass
    for name in ("_weights_sources_closest", "_weights_sources_max_moveout", "set_weights_sources",
                 "_station_density_weights"):
        setattr(_BF, name, getattr(template_search.Beamformer, name))

    def fake_bf(with_availability):
        bf = _BF()
        bf.n_sources, bf.n_stations, bf.stations, bf.moveouts = K, S, names, mv_bp
        bf.network = types.SimpleNamespace(n_stations=S, stations=names,
                                           interstation_distances=dist_bp)
        bf.data = types.SimpleNamespace(set_availability=lambda stations: None)
        if with_availability:
            bf.data.availability = True
            bf.data.availability_per_sta = pd.Series(online, index=names)
        return bf

    wts.update(bp_moveouts=mv_bp, bp_online=online, bp_dist=dist_bp.values)
    bp_cases = [
        dict(avail=False, kw=dict(method="closest_stations", num_closest_stations=4)),
        dict(avail=True, kw=dict(method="closest_stations", num_closest_stations=4)),
        dict(avail=True, kw=dict(method="closest_stations", num_closest_stations=8)),   # > online
        dict(avail=True, kw=dict(method="closest_stations", num_closest_stations=0)),
        dict(avail=True, kw=dict(method="closest_stations", num_closest_stations=5, normalize=True,
                                 n_min_stations=5)),
        dict(avail=True, kw=dict(method="max_moveout", max_moveout=350.0, n_min_stations=3,
                                 normalize=True)),
        dict(avail=False, kw=dict(method="max_moveout", max_moveout=200.0)),
        dict(avail=True, kw=dict(method="closest_stations", num_closest_stations=5,
                                 weight_station_density=True, normalize=True)),
        dict(avail=False, kw=dict(method="closest_stations", num_closest_stations=6,
                                  weight_station_density=True, cutoff_dist=25.0,
                                  lower_percentile=10.0, upper_percentile=80.0)),
    ]
    for j, c in enumerate(bp_cases):
        bf = fake_bf(c["avail"])
        bf.set_weights_sources(**c["kw"])
        wts[f"bp_w_{j}"] = np.asarray(bf.weights_sources)
        wts[f"bp_avail_{j}"] = c["avail"]
        for k_, v_ in c["kw"].items():
            wts[f"bp_kw_{j}_{k_}"] = v_
    wts["bp_n_cases"] = len(bp_cases)
    wts["bp_density_default"] = fake_bf(False)._station_density_weights()

    T, S, C = 30, 8, 3
    names = [f"N{j:02d}" for j in range(S)]
    comps = ["N", "E", "Z"]
    wav = rng.standard_normal((T, S, C, 64)).astype(np.float32)
    wav[rng.random((T, S, C)) < 0.25] = 0.0                   # missing channels are zero-filled
    wav[3] = 0.0
    wav[4, 1:] = 0.0
    n2t = ~(np.sum(wav, axis=-1) == 0.0)                      # dataset.py:5001
    avail_arr = n2t.copy()
    mv_mf = rng.integers(0, 700, (T, S, 2)).astype(np.int32)
    mv_mf[::5, 2, 0] = mv_mf[::5, 4, 0]
    cha_ok = pd.DataFrame(rng.random((S, C)) > 0.2, index=names, columns=comps)
    dist_mf = density_frame(names, rng)

    class _MF:
        pass
    for name in ("_weights_channels_simple", "_weights_channels_closest",
                 "_weights_channels_max_moveout", "set_weights_channels",
                 "_station_density_weights"):
        setattr(_MF, name, getattr(similarity_search.MatchedFilter, name))

    def fake_mf():
        mf = _MF()
        mf.min_channels, mf.min_stations = 6, 3
        mf.template_group = types.SimpleNamespace(
            network_to_template_map=n2t, n_templates=T, availability_arr=avail_arr,
            stations=names, moveouts_arr=mv_mf.copy(), templates=[types.SimpleNamespace(sr=25.0)])
        mf.network = types.SimpleNamespace(n_stations=S, n_components=C, stations=names,
                                           interstation_distances=dist_mf)
        mf.data = types.SimpleNamespace()
        return mf

    wts.update(mf_waveforms=wav, mf_moveouts=mv_mf, mf_availability=avail_arr, mf_dist=dist_mf.values,
               mf_sr=25.0, mf_min_channels=6, mf_min_stations=3)
    mf_cases = [
        dict(method="simple"),
        dict(method="simple", normalize=False, n_min_stations=5),
        dict(method="closest_stations", num_closest_stations=4),
        dict(method="closest_stations", num_closest_stations=4, n_min_stations=4, normalize=False),
        dict(method="closest_stations", num_closest_stations=20),
        dict(method="max_moveout", max_moveout_sec=12.0),
        dict(method="max_moveout", max_moveout_sec=2.0, n_min_stations=40, max_moveout2_sec=20.0),
        dict(method="closest_stations", num_closest_stations=5, weight_station_density=True),
        dict(method="simple", weight_station_density=True, cutoff_dist=30.0, lower_percentile=20.0,
             upper_percentile=90.0),
    ]
    for j, kw in enumerate(mf_cases):
        mf = fake_mf()
        mf.set_weights_channels(**kw)
        wts[f"mf_w_{j}"] = np.asarray(mf.weights_channels)
        for k_, v_ in kw.items():
            wts[f"mf_kw_{j}_{k_}"] = v_
    wts["mf_n_cases"] = len(mf_cases)
    try:   # data-side channel availability: np.logical_and((T,S,C) ndarray, (S,C) DataFrame)
        mf = fake_mf()
        mf.data.availability_per_cha = cha_ok
        mf.set_weights_channels(method="closest_stations", num_closest_stations=4)
        wts["mf_w_cha"] = np.asarray(mf.weights_channels)
        wts["mf_cha_ok"] = cha_ok.values
    except Exception as exc:  # the reference itself cannot run this combination
        print("  (reference fails with data.availability_per_cha:", type(exc).__name__, exc, ")")

    # set_data conditioning and TemplateGroup.normalize
    class _SD:
        pass
    _SD.set_data = similarity_search.MatchedFilter.set_data
    d_arr = rng.standard_normal((S, C, 5000)).astype(np.float32) * rng.uniform(0.1, 30, (S, C, 1)).astype(np.float32)
    d_arr[2, 1] = 0.0
    sd = _SD()
    sd.stations, sd.components, sd.normalize = names, comps, True
    sd.offset_win_peak_amp_sec, sd.duration_win_peak_amp_sec = 1.0, 3.0
    sd.set_data(types.SimpleNamespace(sr=25.0, get_np_array=lambda st, components=None: d_arr.copy()))
    wts.update(sd_data=d_arr, sd_data_arr=np.asarray(sd.data_arr), sd_data_norm=np.asarray(sd.data_norm))

    class _TG:
        pass
    _TG.normalize = BPMF.dataset.TemplateGroup.normalize
    for method in ("rms", "max"):
        tg = _TG()
        tg.waveforms_arr = tg._waveforms_arr = wav.copy()
        tg._remember = lambda name: None
        tg.normalize(method=method)
        wts[f"tg_norm_{method}"] = np.asarray(tg._waveforms_arr)
    np.savez_compressed(os.path.join(HERE, "weights.npz"), **wts)
    relocation_goldens(BPMF)

    print("goldens written to", HERE)
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            print(f"  {fn}: {os.path.getsize(os.path.join(HERE, fn)) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
