"""GPU, BASELINE.json's full time-axis sizes: properties that need no oracle run.

The CPU oracle cannot finish a day of data in test time, so at full N the checks are
size-independent properties of the domain:
  * MF (cfg2 shape, 1 day @ 100 Hz, a subset of templates): every planted event is the row maximum
    neighbourhood's peak at EXACTLY its planted lag; |CC| <= 1; template blocks computed separately
    (the multi-GPU sharding) concatenate to the single-pass result bit for bit; re-running is
    bit-identical; a random sample of lags equals the oracle evaluated on just those windows.
  * MF, whole day against the oracle: 4 templates with moveouts of both signs, all 34.6 M CCs bit-exact
    (the oracle needs ~10 s on the GPU box's 128 host threads); BP: a 2 500-source slab of the cfg3 grid
    over the whole day, max-beam and arg-max bit-exact (~25 s).
  * BP (cfg3: 50 000 sources, 1 day @ 50 Hz): the grid split in two "ranks" with global ids and
    merged with the packed-key max equals the single-pass result bit for bit (what the RCCL
    all-reduce does at N = 2); planted events come out within the bump width of their sample with a beam >= the
    planted source's; strict tail is (0, 0).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_mf_full_day_properties(oracle_lib):
    import torch
    from seismic_bpmf_amd import MatchedFilterGPU, synthetic as syn
    S, C, L, N, T = 20, 3, 256, 8_640_000, 6
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    data = torch.randn((S, C, N), device="cuda", generator=g)
    inp = syn.make_mf_inputs(T, S, C, L, 30_000, seed=4, n_events=0)       # templates / moveouts / weights only
    tmpl, mv, w = inp["templates"], inp["moveouts"] * 10, inp["weights"]     # moveouts up to 3000 samples
    planted = {}
    rng = np.random.default_rng(9)
    for t in range(T):
        lags = np.sort(rng.choice(np.arange(10_000, N - 20_000, 4096), 3, replace=False))
        planted[t] = lags
        for i0 in lags:
            for s in range(S):
                for c in range(C):
                    j = int(i0 + mv[t, s, c])
                    data[s, c, j:j + L] += 3.0 * torch.as_tensor(tmpl[t, s, c], device="cuda")
    data /= data.std(dim=-1, keepdim=True)
    mf = MatchedFilterGPU()
    mf.set_data(data)
    cc = mf.run(tmpl, mv, w, 1)
    cc2 = mf.run(tmpl, mv, w, 1)
    assert torch.equal(cc, cc2)                                             # deterministic
    halves = torch.cat([mf.run(tmpl[:2], mv[:2], w[:2], 1), mf.run(tmpl[2:], mv[2:], w[2:], 1)])
    assert torch.equal(cc, halves)                                          # template sharding
    assert float(cc.abs().max()) <= 1.0 + 1e-5
    cch = cc.cpu().numpy()
    for t in range(T):
        for i0 in planted[t]:
            lo = int(i0) - 2000
            assert lo + int(np.argmax(cch[t, lo:int(i0) + 2000])) == i0
            assert cch[t, i0] > 0.6
    # spot check against the oracle: 5 random 3000-lag stretches of template 1
    d_host = data.cpu().numpy()
    for i0 in rng.integers(0, N - L - 8000, 5):
        i0 = int(i0)
        seg = d_host[:, :, i0:i0 + 3000 + L - 1 + int(mv[1].max())]
        want = oracle_lib.matched_filter(tmpl[1:2], mv[1:2], w[1:2], seg, 1)[0, :3000]
        assert np.array_equal(cch[1, i0:i0 + 3000], want)


def test_mf_full_day_signed_moveouts_whole_day_against_the_oracle(oracle_lib):
    """cfg2's day (20 x 3 channels, L = 256, 8.64 M samples) with moveouts of both signs, 4 templates:
    ALL 34.6 M network CCs equal the oracle bit for bit -- the first and last lags, where the valid
    range starts and ends inside a workgroup and the staging loads are zero-filled by the buffer
    bounds check (the two MFMA-kernel defects of round 2 lived there), the zeros outside the valid
    range, and every lag block in between.  (~10 s of oracle time on the GPU box's host cores; a
    segment-wise oracle would not do: the window energies are differences of double prefix sums from
    the start of the trace, and a segment's own origin rounds them differently.)"""
    import torch
    from seismic_bpmf_amd import MatchedFilterGPU
    S, C, L, N, T = 20, 3, 256, 8_640_000, 4
    g = torch.Generator(device="cuda")
    g.manual_seed(21)
    data = torch.randn((S, C, N), device="cuda", generator=g)
    rng = np.random.default_rng(33)
    tmpl = rng.standard_normal((T, S, C, L)).astype(np.float32)
    mv = rng.integers(-1500, 3001, (T, S, C)).astype(np.int32)
    mv[0, 3, 1], mv[1, 0, 0], mv[2, 5, 2] = -1499, -1022, -3          # first valid lags of every residue mod 4
    w = rng.random((T, S, C)).astype(np.float32)
    w[3, :, 1] = 0.0
    mf = MatchedFilterGPU()
    mf.set_data(data)
    cc = mf.run(tmpl, mv, w, 1).cpu().numpy()
    assert cc.shape == (T, N - L + 1)
    want = oracle_lib.matched_filter(tmpl, mv, w, data.cpu().numpy(), 1)
    assert np.array_equal(cc, want), f"{(cc != want).sum()} of {cc.size} CCs differ"
    for t in range(T):
        used = w[t] != 0
        lo, hi = int(-mv[t][used].min()), int(N - L - mv[t][used].max())
        assert not cc[t, :lo].any() and not cc[t, hi + 1:].any() and cc[t, lo] != 0 and cc[t, hi] != 0


def test_bp_full_day_shard_merge_and_planted_events():
    import torch
    from seismic_bpmf_amd import BeamformerGPU, parallel, postprocess as pp, synthetic as syn
    cfg = syn.BP_CONFIGS["cfg3"]
    geo = syn.make_bp_geometry(cfg["grid"], cfg["S"], cfg["P"], cfg["sr"])
    tau, ws = geo["moveouts"], geo["weights_sources"]
    K, N = tau.shape[0], cfg["N"]
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    feat = torch.randn((cfg["S"], cfg["C"], N), device="cuda", generator=g).abs_()
    rng = np.random.default_rng(2)
    sig, half = 10.0, 40
    bump = torch.as_tensor(8.0 * np.exp(-0.5 * (np.arange(-half, half + 1) / sig) ** 2), dtype=torch.float32,
                           device="cuda")
    planted = []
    for _ in range(12):
        k0, t0 = int(rng.integers(0, K)), int(rng.integers(10_000, N - 10_000))
        for s in range(cfg["S"]):
            for c in range(cfg["C"]):
                x = t0 + int(tau[k0, s, 0 if c == 0 else 1])
                feat[s, c, x - half:x + half + 1] += bump
        planted.append((k0, t0))
    wp = syn.phase_weights(cfg["S"], cfg["C"], cfg["P"])
    full = BeamformerGPU(tau, ws)
    beam, arg = full.run(feat, wp, "max", "strict")
    b2, a2 = full.run(feat, wp, "max", "strict")
    assert torch.equal(beam, b2) and torch.equal(arg, a2)
    # two "ranks": contiguous halves of the grid with global ids, merged like the all-reduce does
    k_half = K // 2
    r0 = BeamformerGPU(tau[:k_half], ws[:k_half], source_id_offset=0)
    r1 = BeamformerGPU(tau[k_half:], ws[k_half:], source_id_offset=k_half)
    p0 = parallel.pack_max_keys(*r0.run(feat, wp, "max", "strict"))
    p1 = parallel.pack_max_keys(*r1.run(feat, wp, "max", "strict"))
    mb, ma = parallel.unpack_max_keys(torch.maximum(p0, p1))
    assert torch.equal(mb, beam) and torch.equal(ma, arg)
    # HIP pack kernel == torch packing
    assert torch.equal(r0.pack_max(beam, arg), parallel.pack_max_keys(beam, arg))
    maxbeam, sources = beam.cpu().numpy(), arg.cpu().numpy()
    tmax_used = np.where(ws[:, :, None] != 0, tau, -1).max(axis=(1, 2))   # per source, used stations only
    tail = N - int(tmax_used.min())               # first sample no source can compute under 'strict'
    assert not maxbeam[tail:].any() and not sources[tail:].any() and maxbeam[tail - 1] > 0
    peaks, psrc = pp.find_beam_detections(maxbeam, sources, np.full(N, 5.0, np.float32), 500)
    for k0, t0 in planted:
        # a neighbouring grid node can stack the (20-sample wide) bumps slightly earlier or later
        hit = np.flatnonzero(np.abs(peaks - t0) <= 15)
        assert hit.size == 1, (k0, t0, peaks[np.abs(peaks - t0) < 2000])
        assert maxbeam[peaks[hit[0]]] >= maxbeam[t0] >= 8.0      # 20 terms x 0.1 x (8 + noise)
    for b in (full, r0, r1):
        b.close()


def test_bp_full_day_slab_of_the_grid_against_the_oracle(oracle_lib):
    """cfg3's day (20 stations x 3 components, 4.32 M samples) and a slab of 2 500 sources of its grid
    (50 x 50 x 1: one depth level, 10 closest stations): every one of the 8 438 tiles of the day --
    bp_beam_fast_kernel on the interior ones, round 1's kernel on the ends -- max-beam and arg-max
    equal the oracle bit for bit, strict and flexible.  (~20 s of oracle time per mode on the GPU
    box's host cores.)"""
    import torch
    from seismic_bpmf_amd import BeamformerGPU, synthetic as syn
    cfg = syn.BP_CONFIGS["cfg3"]
    geo = syn.make_bp_geometry((50, 50, 1), cfg["S"], cfg["P"], cfg["sr"], n_closest=10)
    tau, ws = geo["moveouts"], geo["weights_sources"]
    N = cfg["N"]
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    feat = torch.randn((cfg["S"], cfg["C"], N), device="cuda", generator=g).abs_()
    wp = syn.phase_weights(cfg["S"], cfg["C"], cfg["P"])
    f_host = feat.cpu().numpy()
    bf = BeamformerGPU(tau, ws)
    try:
        info = bf.plan_info()
        for oob in ("strict", "flexible"):
            beam, arg = bf.run(feat, wp, "max", oob)
            ob, oa = oracle_lib.beamform(f_host, tau, wp, ws, oob, "max")
            b, a = beam.cpu().numpy(), arg.cpu().numpy()
            assert np.array_equal(b, ob), f"{oob}: {(b != ob).sum()} of {N} beams differ ({info})"
            assert np.array_equal(a, oa), f"{oob}: {(a != oa).sum()} of {N} arg-max differ"
    finally:
        bf.close()


def test_mf_baseline_config0_whole_against_oracle_and_float64(oracle_lib):
    """BASELINE.json configs[0], the reference's own CPU-runnable case, in full: 4 templates x 8
    stations x 3 components, 128-sample templates, 1 h @ 50 Hz.  Bit-exact against the oracle in
    both output layouts, within the float32 tolerance of a float64 brute force on a sample of lags,
    and every planted event detected at its planted index."""
    from oracle import ref_numpy
    from seismic_bpmf_amd import matched_filter
    from seismic_bpmf_amd import synthetic as syn
    cfg = syn.MF_CONFIGS["cfg1"]
    mf = syn.make_mf_inputs(cfg["T"], cfg["S"], cfg["C"], cfg["L"], cfg["N"], seed=20260929)
    args = (mf["templates"], mf["moveouts"], mf["weights"], mf["data"], 1)
    cc = matched_filter(*args, arch="gpu", check_zeros=False)
    assert np.array_equal(cc, oracle_lib.matched_filter(*args))
    per = matched_filter(*args, arch="gpu", network_sum=False)
    assert np.array_equal(per, oracle_lib.matched_filter(*args, network_sum=False))
    for t, i0 in mf["planted"]:
        lo = max(0, i0 - 300)
        assert lo + int(cc[t, lo:i0 + 300].argmax()) == i0
    exact = ref_numpy.matched_filter_f64(*args)       # float64 brute force, no summation-order care
    assert np.abs(cc - exact).max() < 2e-5            # the float32 tolerance on CC values
