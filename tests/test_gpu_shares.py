"""GPU: the per-GPU shares of BASELINE.json configs[3] and configs[4] at their full sizes.

configs[3] = 5000 templates x 40 stations x 3 components, 1 day @ 100 Hz, templates sharded over
8 GPUs -> one rank holds 625 templates x 120 channels (`cfg4_per_gpu`); configs[4] = 1M sources x
40 stations, 1 day @ 100 Hz, grid tiles sharded -> one rank scans 125 000 sources against 80
station-phase rows (`cfg5_per_gpu`: a different plan regime from cfg3 -- ~300 LDS groups instead
of ~50).  The CPU oracle cannot run a day, so the checks are the size-independent ones of
test_gpu_fullsize.py: planted events at their exact index, block concatenation / 2-"rank" merge ==
single pass bit for bit, determinism, and oracle spot-checks on random windows.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _plant_mf(data, tmpl, mv, templates, lags_of, amp=3.0):
    import torch
    S, C = tmpl.shape[1:3]
    L = tmpl.shape[-1]
    for t in templates:
        wf = torch.as_tensor(tmpl[t], device=data.device)
        for i0 in lags_of[t]:
            for s in range(S):
                for c in range(C):
                    j = int(i0 + mv[t, s, c])
                    data[s, c, j:j + L] += amp * wf[s, c]


def test_mf_cfg4_share_full_day(oracle_lib):
    import torch
    from seismic_bpmf_amd import MatchedFilterGPU, synthetic as syn
    cfg = syn.MF_CONFIGS["cfg4_per_gpu"]
    T, S, C, L, N = cfg["T"], cfg["S"], cfg["C"], cfg["L"], cfg["N"]
    g = torch.Generator(device="cuda")
    g.manual_seed(41)
    data = torch.randn((S, C, N), device="cuda", generator=g)
    inp = syn.make_mf_inputs(T, S, C, L, 30_000, seed=44, n_events=0)    # templates / moveouts / weights
    tmpl, mv, w = inp["templates"], inp["moveouts"] * 10, inp["weights"]   # moveouts up to 3000 samples
    rng = np.random.default_rng(45)
    probe = sorted(rng.choice(T, 16, replace=False).tolist() + [0, T - 1])
    lags_of = {t: np.sort(rng.choice(np.arange(10_000, N - 20_000, 4096), 2, replace=False))
               for t in probe}
    _plant_mf(data, tmpl, mv, probe, lags_of)
    data /= data.std(dim=-1, keepdim=True)
    mf = MatchedFilterGPU()
    mf.set_data(data)
    cc = mf.run(tmpl, mv, w, 1)                                           # (625, 8 639 745): 21.6 GB
    assert cc.shape == (T, N - L + 1)
    assert float(cc.abs().max()) <= 1.0 + 1e-5
    # template sharding: blocks computed separately (what 2 ranks of a finer split would do)
    # concatenate to the single pass bit for bit
    k = 313
    lo_half = mf.run(tmpl[:k], mv[:k], w[:k], 1)
    assert torch.equal(cc[:k], lo_half)
    del lo_half
    hi_half = mf.run(tmpl[k:], mv[k:], w[k:], 1)
    assert torch.equal(cc[k:], hi_half)
    del hi_half
    for t in probe:
        row = cc[t].cpu().numpy()
        for i0 in lags_of[t]:
            lo = int(i0) - 2000
            assert lo + int(np.argmax(row[lo:int(i0) + 2000])) == i0, (t, i0)
            assert row[i0] > 0.6
    # oracle spot checks: random 2000-lag stretches of three templates (all 120 channels)
    d_host = None
    for t in (probe[0], probe[7], probe[-1]):
        row = cc[t].cpu().numpy()
        for i0 in rng.integers(0, N - L - 8000, 2):
            i0 = int(i0)
            n_seg = 2000 + L - 1 + int(mv[t].max())
            seg = data[:, :, i0:i0 + n_seg].cpu().numpy()
            want = oracle_lib.matched_filter(tmpl[t:t + 1], mv[t:t + 1], w[t:t + 1], seg, 1)[0, :2000]
            assert np.array_equal(row[i0:i0 + 2000], want), (t, i0)
    del d_host
    # second run of SURVEY 8d: only the 10 "closest" stations of each template keep weight
    sub = np.asarray(probe[:8])
    w10 = w[sub].copy()
    order = np.argsort(mv[sub][:, :, 0], axis=1)
    for q in range(sub.size):
        w10[q, order[q, 10:], :] = 0.0
    w10 /= w10.sum(axis=(1, 2), keepdims=True)
    cc10 = mf.run(tmpl[sub], mv[sub], w10, 1).cpu().numpy()
    for q in (0, 5):
        t = int(sub[q])
        i0 = int(lags_of[t][0]) - 700
        n_seg = 1500 + L - 1 + int(mv[t].max())
        seg = data[:, :, i0:i0 + n_seg].cpu().numpy()
        want = oracle_lib.matched_filter(tmpl[t:t + 1], mv[t:t + 1], w10[q:q + 1], seg, 1)[0, :1500]
        assert np.array_equal(cc10[q, i0:i0 + 1500], want)
        assert i0 + int(np.argmax(cc10[q, i0:i0 + 1500])) == lags_of[t][0]


def test_bp_cfg5_share_full_day(oracle_lib):
    import torch
    from seismic_bpmf_amd import BeamformerGPU, parallel, postprocess as pp, synthetic as syn
    cfg = syn.BP_CONFIGS["cfg5_per_gpu"]
    S, C, P, N = cfg["S"], cfg["C"], cfg["P"], cfg["N"]
    geo = syn.make_bp_geometry(cfg["grid"], S, P, cfg["sr"])
    tau, ws = geo["moveouts"], geo["weights_sources"]
    K = tau.shape[0]
    assert K == 125_000
    g = torch.Generator(device="cuda")
    g.manual_seed(51)
    feat = torch.randn((S, C, N), device="cuda", generator=g).abs_()
    rng = np.random.default_rng(52)
    sig, half = 20.0, 80                                                   # 0.2 s at 100 Hz
    bump = torch.as_tensor(8.0 * np.exp(-0.5 * (np.arange(-half, half + 1) / sig) ** 2),
                           dtype=torch.float32, device="cuda")
    planted = []
    for _ in range(10):
        k0, t0 = int(rng.integers(0, K)), int(rng.integers(20_000, N - 20_000))
        for s in range(S):
            for c in range(C):
                x = t0 + int(tau[k0, s, 0 if c == 0 else 1])
                feat[s, c, x - half:x + half + 1] += bump
        planted.append((k0, t0))
    wp = syn.phase_weights(S, C, P)
    full = BeamformerGPU(tau, ws)
    info = full.plan_info()
    assert info["gather_bytes"] == 8 and info["n_groups"] > 100            # the 80-row regime
    beam, arg = full.run(feat, wp, "max", "strict")
    b2, a2 = full.run(feat, wp, "max", "strict")
    assert torch.equal(beam, b2) and torch.equal(arg, a2)
    # two "ranks" with global ids, merged with the packed-key max (the all-reduce at world size 2)
    k_half = K // 2
    r0 = BeamformerGPU(tau[:k_half], ws[:k_half], source_id_offset=0)
    r1 = BeamformerGPU(tau[k_half:], ws[k_half:], source_id_offset=k_half)
    p0 = parallel.pack_max_keys(*r0.run(feat, wp, "max", "strict"))
    p1 = parallel.pack_max_keys(*r1.run(feat, wp, "max", "strict"))
    mb, ma = parallel.unpack_max_keys(torch.maximum(p0, p1))
    assert torch.equal(mb, beam) and torch.equal(ma, arg)
    assert torch.equal(r0.pack_max(beam, arg), parallel.pack_max_keys(beam, arg))
    maxbeam, sources = beam.cpu().numpy(), arg.cpu().numpy()
    tmax_used = np.where(ws[:, :, None] != 0, tau, -1).max(axis=(1, 2))
    tail = N - int(tmax_used.min())
    assert not maxbeam[tail:].any() and not sources[tail:].any() and maxbeam[tail - 1] > 0
    # oracle spot checks: all 125 000 sources over three 150-sample windows (one around an event)
    tmax = int(tau.max())
    for i0 in (planted[0][1] - 60, int(rng.integers(0, N - tmax - 1000)), 0):
        W = 150
        seg = feat[:, :, i0:i0 + W + tmax + 1].cpu().numpy()
        ob, oa = oracle_lib.beamform(seg, tau, wp, ws, "strict", "max")
        assert np.array_equal(maxbeam[i0:i0 + W], ob[:W]), i0
        assert np.array_equal(sources[i0:i0 + W], oa[:W]), i0
    peaks, psrc = pp.find_beam_detections(maxbeam, sources, np.full(N, 5.0, np.float32), 1000)
    for k0, t0 in planted:
        hit = np.flatnonzero(np.abs(peaks - t0) <= 30)
        assert hit.size == 1, (k0, t0, peaks[np.abs(peaks - t0) < 4000])
        assert maxbeam[peaks[hit[0]]] >= maxbeam[t0] >= 8.0
    for b in (full, r0, r1):
        b.close()


def test_mf_cfg2_all_500_templates_full_day():
    """BASELINE configs[1] at its full size, all 500 templates: every planted event (5 per template)
    is found by the device detection stage at EXACTLY its planted CC index (a handful may share a
    merge window with a stronger one), template blocks concatenate to the single pass bit for bit,
    re-running is bit-identical, |CC| <= 1."""
    import torch
    from seismic_bpmf_amd import MatchedFilterGPU, synthetic as syn
    from seismic_bpmf_amd.threshold import ThresholdGPU
    cfg = syn.MF_CONFIGS["cfg2"]
    T, S, C, L, N = cfg["T"], cfg["S"], cfg["C"], cfg["L"], cfg["N"]
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    data = torch.randn((S, C, N), device=dev, generator=g)
    raw = torch.randn((T, S, C, L + 4), device=dev, generator=g)
    tmpl = sum(raw[..., k:k + L] for k in range(5))
    tmpl = tmpl - tmpl.mean(dim=-1, keepdim=True)
    tmpl = (tmpl / tmpl.std(dim=-1, keepdim=True)).contiguous()
    mv_p = torch.randint(0, 1501, (T, S), device=dev, generator=g)
    mv_s = mv_p + torch.randint(0, 1501, (T, S), device=dev, generator=g)
    mv = torch.empty((T, S, C), dtype=torch.int32, device=dev)
    mv[:, :, 0] = mv_p
    mv[:, :, 1:] = mv_s[:, :, None]
    w = torch.full((T, S, C), 1.0 / (S * C), device=dev)
    n_ev = 5
    slots = torch.randint(0, (N - L - 3002) // (4 * L), (T, n_ev), device=dev, generator=g)
    amps = 1.5 + 2.5 * torch.rand((T, n_ev), device=dev, generator=g)
    ar = torch.arange(L, device=dev)
    for t in range(T):
        for e in range(n_ev):
            j = (slots[t, e] * 4 * L + mv[t].long())[..., None] + ar
            data.scatter_add_(2, j, amps[t, e] * tmpl[t])
    data /= data.std(dim=-1, keepdim=True)
    planted = (slots * 4 * L).cpu().numpy()
    mf = MatchedFilterGPU()
    mf.set_data(data)
    cc = mf.run(tmpl, mv, w, 1)
    assert cc.shape == (T, N - L + 1) and float(cc.abs().max()) <= 1.0 + 1e-5
    again = mf.run(tmpl, mv, w, 1)
    assert torch.equal(cc, again)
    del again
    k = 187
    part = mf.run(tmpl[:k], mv[:k], w[:k], 1)
    assert torch.equal(cc[:k], part)
    del part
    part = mf.run(tmpl[k:], mv[k:], w[k:], 1)
    assert torch.equal(cc[k:], part)
    del part
    th = ThresholdGPU()
    wn = np.random.default_rng(5).standard_normal(500).astype(np.float32)
    thr_win, _ = th.time_dependent_threshold(cc, 180_000, 8.0, overlap=0.25, white_noise=wn)
    cand = th.extract_candidates(cc, thr_win, 180_000, overlap=0.25)
    exact = 0
    for t in range(T):
        mine = cand[cand["row"] == t]
        idx, val = list(mine["index"]), list(mine["cc"])
        q = 1
        while q < len(idx):                       # pair-wise merge, similarity_search.py:240-251
            if idx[q] - idx[q - 1] < 512:
                drop = q - 1 if val[q] > val[q - 1] else q
                del idx[drop], val[drop]
            else:
                q += 1
        exact += len(set(idx) & set(planted[t].tolist()))
    assert exact >= T * n_ev - 5, exact


@pytest.mark.parametrize("name", ["cfg3", "cfg5_per_gpu"])
def test_bp_dense_station_weights_full_day(oracle_lib, name):
    """BASELINE's literal "x 20 / x 40 stations": EVERY station of every source weighted -- what
    `_weights_sources_closest` returns for num_closest_stations >= n_stations
    (BPMF/template_search.py:779-798).  These grids run the station-count classes of bp_fast.hip (17-32
    stations: tile 256, two records per source; 33-64: two LDS residencies per group) that the
    10-closest-station workloads never touch.  Full size: configs[2] (50 000 sources x 20 stations, 1 day
    @ 50 Hz) and one GPU's share of configs[4] (125 000 sources x 40 stations, 1 day @ 100 Hz) --
    all sources against the oracle over three 150-sample windows (one around a planted event, one at
    the start of the day where sources begin to enter, one random), the 2-"rank" packed-key merge
    equal to the single pass bit for bit, determinism, the zero tail of the day."""
    import torch
    from seismic_bpmf_amd import BeamformerGPU, parallel, synthetic as syn
    cfg = syn.BP_CONFIGS[name]
    S, C, P, N = cfg["S"], cfg["C"], cfg["P"], cfg["N"]
    geo = syn.make_bp_geometry(cfg["grid"], S, P, cfg["sr"])
    tau = geo["moveouts"]
    K = tau.shape[0]
    ws = np.full((K, S), 1.0 / S, dtype=np.float32)
    g = torch.Generator(device="cuda")
    g.manual_seed(61)
    feat = torch.randn((S, C, N), device="cuda", generator=g).abs_()
    rng = np.random.default_rng(62)
    sig = 0.2 * cfg["sr"]
    half = int(4 * sig)
    bump = torch.as_tensor(8.0 * np.exp(-0.5 * (np.arange(-half, half + 1) / sig) ** 2),
                           dtype=torch.float32, device="cuda")
    planted = []
    for _ in range(6):
        k0, t0 = int(rng.integers(0, K)), int(rng.integers(20_000, N - 20_000))
        for s in range(S):
            for c in range(C):
                x = t0 + int(tau[k0, s, 0 if c == 0 else 1])
                feat[s, c, x - half:x + half + 1] += bump
        planted.append((k0, t0))
    wp = syn.phase_weights(S, C, P)
    full = BeamformerGPU(tau, ws)
    info = full.plan_info()
    assert info["gather_bytes"] == 8 and info["n_classes"] >= 1 and max(info["class_stations_max"]) == S, info
    beam, arg = full.run(feat, wp, "max", "strict")
    b2, a2 = full.run(feat, wp, "max", "strict")
    assert torch.equal(beam, b2) and torch.equal(arg, a2)
    del b2, a2
    k_half = K // 2 + 17
    r0 = BeamformerGPU(tau[:k_half], ws[:k_half], source_id_offset=0)
    r1 = BeamformerGPU(tau[k_half:], ws[k_half:], source_id_offset=k_half)
    p0 = parallel.pack_max_keys(*r0.run(feat, wp, "max", "strict"))
    p1 = parallel.pack_max_keys(*r1.run(feat, wp, "max", "strict"))
    mb, ma = parallel.unpack_max_keys(torch.maximum(p0, p1))
    assert torch.equal(mb, beam) and torch.equal(ma, arg)
    del p0, p1, mb, ma
    maxbeam, sources = beam.cpu().numpy(), arg.cpu().numpy()
    tail = N - int(tau.max(axis=(1, 2)).min())
    assert not maxbeam[tail:].any() and not sources[tail:].any() and maxbeam[tail - 1] > 0
    tmax = int(tau.max())
    W = 150
    for i0 in (planted[0][1] - 60, 0, int(rng.integers(0, N - tmax - 1000)), N - tmax - W - 1):
        seg = feat[:, :, i0:i0 + W + tmax + 1].cpu().numpy()
        ob, oa = oracle_lib.beamform(seg, tau, wp, ws, "strict", "max")
        assert np.array_equal(maxbeam[i0:i0 + W], ob[:W]), (name, i0)
        assert np.array_equal(sources[i0:i0 + W], oa[:W]), (name, i0)
    # every planted event is the day's maximum in its neighbourhood, located at a source whose beam
    # equals the planted source's
    for k0, t0 in planted:
        lo = t0 - 200
        assert abs(lo + int(np.argmax(maxbeam[lo:t0 + 200])) - t0) <= 3, (k0, t0)
        assert maxbeam[t0] >= 8.0
    for b in (full, r0, r1):
        b.close()
