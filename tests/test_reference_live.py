"""CPU, build container only: the host mirrors and the pinned C oracle against the REFERENCE ITSELF,
imported from /root/reference with stub modules (tests/golden/make_goldens.py:import_reference), on
seeded random inputs -- wider than the committed golden vectors.  Skipped where /root/reference does
not exist (the GPU box); nothing here is needed by the product."""
import importlib.util
import os
import types
from unittest.mock import MagicMock

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "BPMF")), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_goldens", os.path.join(here, "golden", "make_goldens.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    import sys
    cwd, path0, mods0 = os.getcwd(), list(sys.path), set(sys.modules)
    try:
        BPMF, clib = mg.import_reference()           # chdir()s into a scratch directory: BPMF reads its cfg from the CWD
    finally:
        os.chdir(cwd)
    from BPMF import similarity_search, template_search, utils, dataset
    yield types.SimpleNamespace(clib=clib, ss=similarity_search, ts=template_search, utils=utils, ds=dataset, mg=mg)
    # leave no trace for the other test modules: the stub modules (fast_matched_filter, beampower, obspy ...)
    # and the reference's path entry go away again
    for name in set(sys.modules) - mods0:
        if isinstance(sys.modules[name], MagicMock) or name == "BPMF" or name.startswith("BPMF."):
            del sys.modules[name]
    sys.path[:] = path0


def test_live_sec_to_samp_and_detect_peaks(ref):
    from seismic_bpmf_amd import postprocess as pp
    rng = np.random.default_rng(1)
    for sr in (20.0, 25.0, 50.0, 100.0):
        t = np.concatenate([rng.uniform(-200, 200, 500), np.round(rng.uniform(-5, 5, 200), 2)])
        assert np.array_equal(pp.sec_to_samp(t, sr), ref.utils.sec_to_samp(t, sr=sr))
    for j in range(30):
        n = int(rng.choice([3, 10, 500, 4000]))
        x = np.abs(rng.standard_normal(n))
        if j % 3 == 0:
            x = np.round(x, 1)                        # plateaus and exact ties (same NumPy sort on the same array)
        if j % 5 == 0 and n > 20:
            x[rng.integers(0, n, 3)] = np.nan
        for mpd in (1, 2, 7, 60, 1000):
            assert np.array_equal(pp.detect_peaks(x, mpd=mpd), ref.utils._detect_peaks(x.copy(), mpd=mpd)), (j, n, mpd)


def test_live_thresholds(ref):
    from oracle import oracle
    from seismic_bpmf_amd import postprocess as pp
    rng = np.random.default_rng(2)
    for j in range(8):
        n = int(rng.choice([20_000, 43_200, 60_001]))
        window = int(rng.choice([1_000, 3_000, 7_001]))
        overlap = float(rng.choice([0.5, 0.75, 0.9]))
        mb = (np.abs(rng.standard_normal(n)) + 6 * (rng.random(n) > 0.998)).astype(np.float32)
        want = ref.ts.time_dependent_threshold(mb, window, overlap=overlap, CNR_threshold=12.0)
        assert np.array_equal(pp.bp_time_dependent_threshold(mb, window, 12.0, overlap=overlap), want), (j, n, window, overlap)
    for j in range(6):
        n, window = int(rng.choice([30_000, 50_000])), int(rng.choice([2_000, 6_000]))
        overlap = float(rng.choice([0.25, 0.5, 0.66]))
        x = ref.mg.cc_like_series(rng, n, gaps=bool(j % 2))
        wn = rng.standard_normal(int((x == 0).sum())).astype(np.float32)
        want = ref.ss.time_dependent_threshold(x, window, overlap=overlap, threshold_type="mad", white_noise=wn)
        assert np.array_equal(pp.time_dependent_threshold_mad(x, window, 8.0, overlap=overlap, white_noise=wn), want), j
    # the reference's compiled libc.c (num_threads=1: its OpenMP loops race) against the C oracle, at
    # sizes that keep the reference's own expansion index inside its table
    done = 0
    while done < 25:
        n = int(rng.integers(20_000, 120_000))
        window = int(rng.integers(1_000, 15_000))
        overlap = float(rng.choice([0.0, 0.2, 0.25, 0.5]))
        shift = int((1.0 - overlap) * window)
        nsw = (n - (window - shift)) // shift
        if nsw < 2 or (n - shift - 1) // shift > nsw - 1:      # (odd windows with overlap 0 -- shift = 2 (window // 2) + 1 -- included)
            continue
        x = ref.mg.cc_like_series(rng, n, gaps=bool(done % 2))
        wn = rng.standard_normal(500).astype(np.float32)
        want = ref.clib.time_dependent_threshold(x, window, 8.0, overlap=overlap, white_noise=wn, num_threads=1)
        assert np.array_equal(oracle.time_dependent_threshold(x, window, 8.0, overlap, wn), want), (n, window, overlap)
        done += 1


def test_live_select_cc_indexes(ref):
    from oracle import oracle
    from seismic_bpmf_amd import postprocess as pp
    rng = np.random.default_rng(3)

    class _Data:
        sr = 25.0
        duration = 1500.0
    for j in range(10):
        n = int(rng.choice([20_000, 60_000]))
        win = int(rng.choice([1, 30, 250, 2_000]))
        step = int(rng.choice([1, 1, 2]))
        remove_edges = bool(j % 2)
        acdf = float(rng.choice([0.0, 0.5]))
        fake = MagicMock()
        fake.data, fake.step, fake.threshold_type, fake.remove_edges = _Data(), step, "rms", remove_edges
        x = ref.mg.cc_like_series(rng, n, gaps=False)
        thr = (0.12 + 0.05 * rng.random() + 0.0 * x).astype(np.float32)
        want = ref.ss.MatchedFilter.select_cc_indexes(fake, x, thr, win, anomalous_cdf_at_mean_plus_1sig=acdf)
        got = pp.select_cc_indexes(x, thr, win, step=step, sr=25.0, data_duration_sec=1500.0, n_dev_threshold=8.0,
                                   min_freq_hz=2.0, data_buffer_sec=500.0, remove_edges=remove_edges,
                                   anomalous_cdf_at_mean_plus_1sig=acdf)
        assert np.array_equal(np.asarray(got, dtype=np.int64), np.asarray(want, dtype=np.int64)), (j, n, win, step)
        sel = ref.clib.select_cc_indexes(x, thr, win)
        assert np.array_equal(oracle.select_cc_indexes(x, thr, win), np.asarray(sel)), (j, "C variant")


def test_live_kurtosis_and_grid_decimation(ref):
    from oracle import oracle
    rng = np.random.default_rng(4)
    for W in (1, 4, 50, 333):
        sig = (rng.standard_normal((2, 2, 1500)) * np.array([1.0, 1e-3])[None, :, None]).astype(np.float32)
        sig[0, 0, 300:700] = 0.0
        assert np.array_equal(oracle.kurtosis(sig, W), ref.clib.kurtosis(sig, W), equal_nan=True), W
    for j in range(4):
        K, S = int(rng.choice([60, 300, 700])), int(rng.integers(3, 10))
        lon, lat = rng.uniform(0, 1, K).astype(np.float32), rng.uniform(0, 1, K).astype(np.float32)
        sta = rng.uniform(0, 1, (S, 2))
        mv = np.round(np.hypot(lon[:, None] - sta[None, :, 0], lat[:, None] - sta[None, :, 1]) * 20, 1).astype(np.float32)
        cl = np.linspace(-0.01, 1.01, int(rng.integers(2, 5))).astype(np.float32)
        nd, thr = int(rng.integers(1, S + 1)), float(rng.choice([0.1, 0.4, 2.0]))
        for method in ("closest", "smallest"):
            want = ref.clib.find_similar_sources(mv, lon, lat, cl, cl, thr, num_threads=1,
                                                 num_stations_for_diff=nd, method=method)
            got = oracle.find_similar_sources(mv, lon, lat, cl, cl, thr, nd, method)
            assert np.array_equal(np.asarray(got, bool), np.asarray(want, bool)), (j, method, K, S, nd, thr)


def test_live_source_weight_builders(ref):
    """Beamformer.set_weights_sources and its builders (BPMF/template_search.py:779-895) through a fake
    `self` carrying only the attributes they read, on random moveout tables with ties at the cut-off
    and random offline stations, against postprocess.set_weights_sources."""
    import pandas as pd
    from seismic_bpmf_amd import postprocess as pp
    rng = np.random.default_rng(5)

    class _BF:
        pass
    for name in ("_weights_sources_closest", "_weights_sources_max_moveout", "set_weights_sources",
                 "_station_density_weights"):
        setattr(_BF, name, getattr(ref.ts.Beamformer, name))
    for j in range(30):
        K, S, P = int(rng.integers(1, 200)), int(rng.integers(2, 14)), 2
        names = [f"S{i:02d}" for i in range(S)]
        mv = rng.integers(0, int(rng.choice([5, 60, 900])), (K, S, P)).astype(np.int64)     # small ranges: many ties
        mv[:, :, 1] += mv[:, :, 0] // 2
        mv -= mv.reshape(K, -1).min(axis=1)[:, None, None]
        online = rng.random(S) > 0.25
        if not online.any():
            online[0] = True
        with_avail = bool(j % 2)
        xy = rng.uniform(0.0, 80.0, (S, 2))
        dist = pd.DataFrame(np.sqrt(((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1)), index=names, columns=names)
        bf = _BF()
        bf.n_sources, bf.n_stations, bf.stations, bf.moveouts = K, S, names, mv
        bf.network = types.SimpleNamespace(n_stations=S, stations=names, interstation_distances=dist)
        bf.data = types.SimpleNamespace(set_availability=lambda stations: None)
        if with_avail:
            bf.data.availability = True
            bf.data.availability_per_sta = pd.Series(online, index=names)
        kw = dict(normalize=bool(rng.random() < 0.5), n_min_stations=int(rng.choice([0, 0, 2, 5])))
        if rng.random() < 0.6:
            kw.update(method="closest_stations", num_closest_stations=int(rng.integers(0, S + 3)))
        else:
            kw.update(method="max_moveout", max_moveout=float(rng.choice([3.0, 40.0, 500.0])))
        density = bool(rng.random() < 0.3)
        dens = None
        if density:
            kw["weight_station_density"] = True
            dens = pp.station_density_weights(dist.values)
        bf.set_weights_sources(**kw)
        kw.pop("weight_station_density", None)
        got = pp.set_weights_sources(mv, online=online if with_avail else None, density_weights=dens, **kw)
        want = np.asarray(bf.weights_sources)
        assert got.shape == want.shape and np.array_equal(got, want.astype(got.dtype)), (j, K, S, kw, with_avail, density)


def test_live_channel_weight_builders(ref):
    """MatchedFilter.set_weights_channels and its builders (BPMF/similarity_search.py:288-474) through a
    fake `self`, on random template sets with missing channels, dead templates and moveout ties,
    against postprocess.set_weights_channels."""
    import pandas as pd
    from seismic_bpmf_amd import postprocess as pp
    rng = np.random.default_rng(6)

    class _MF:
        pass
    for name in ("_weights_channels_simple", "_weights_channels_closest", "_weights_channels_max_moveout",
                 "set_weights_channels", "_station_density_weights"):
        setattr(_MF, name, getattr(ref.ss.MatchedFilter, name))
    for j in range(30):
        T, S, C = int(rng.integers(1, 25)), int(rng.integers(2, 10)), 3
        names = [f"N{i:02d}" for i in range(S)]
        wav = rng.standard_normal((T, S, C, 16)).astype(np.float32)
        wav[rng.random((T, S, C)) < 0.3] = 0.0
        if T > 2:
            wav[1] = 0.0
        # the map as the reference builds it (dataset.py:4977-5008): non-zero channels AND the
        # stations selected on each template (a random subset, as n_closest_stations leaves them)
        chosen = [sorted(rng.choice(S, size=int(rng.integers(1, S + 1)), replace=False).tolist()) for _ in range(T)]
        tg = types.SimpleNamespace(
            n_templates=T, waveforms_arr=wav,
            network=types.SimpleNamespace(n_stations=S, n_components=C,
                                          station_indexes=pd.Series(np.arange(S), index=names)),
            templates=[types.SimpleNamespace(stations=[names[i] for i in ch]) for ch in chosen])
        ref.ds.TemplateGroup.set_network_to_template_map(tg)
        n2t = np.asarray(tg._network_to_template_map)
        mine = pp.network_to_template_map(wav, chosen)
        assert np.array_equal(mine, n2t), j
        sel = np.zeros((T, S), dtype=bool)
        for t, ch in enumerate(chosen):
            sel[t, ch] = True
        assert np.array_equal(pp.network_to_template_map(wav, sel), n2t), j
        avail = ~(np.sum(wav, axis=-1) == 0.0)
        mv = rng.integers(0, int(rng.choice([4, 50, 700])), (T, S, 2)).astype(np.int32)
        min_ch, min_st = int(rng.integers(1, 8)), int(rng.integers(1, 4))
        xy = rng.uniform(0.0, 80.0, (S, 2))
        dist = pd.DataFrame(np.sqrt(((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1)), index=names, columns=names)
        mf = _MF()
        mf.min_channels, mf.min_stations = min_ch, min_st
        mf.template_group = types.SimpleNamespace(network_to_template_map=n2t, n_templates=T, availability_arr=avail,
                                                  stations=names, moveouts_arr=mv.copy(),
                                                  templates=[types.SimpleNamespace(sr=25.0)])
        mf.network = types.SimpleNamespace(n_stations=S, n_components=C, stations=names, interstation_distances=dist)
        mf.data = types.SimpleNamespace()
        r = rng.random()
        if r < 0.35:
            kw = dict(method="simple")
        elif r < 0.7:
            kw = dict(method="closest_stations", num_closest_stations=int(rng.integers(1, S + 3)))
        else:
            kw = dict(method="max_moveout", max_moveout_sec=float(rng.choice([0.1, 2.0, 30.0])))
        kw.update(normalize=bool(rng.random() < 0.6), n_min_stations=int(rng.choice([0, 0, 3])))
        dens = None
        if rng.random() < 0.3:
            kw["weight_station_density"] = True
            dens = pp.station_density_weights(dist.values)
        mf.set_weights_channels(**kw)
        kw.pop("weight_station_density", None)
        got = pp.set_weights_channels(density_weights=dens, present=mine, moveouts=mv,
                                      availability=avail, sr=25.0, min_channels=min_ch, min_stations=min_st, **kw)
        want = np.asarray(mf.weights_channels)
        assert got.shape == want.shape and np.array_equal(got, want.astype(got.dtype)), (j, T, S, kw)
