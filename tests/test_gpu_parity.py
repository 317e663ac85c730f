"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle.

Bar: BIT-EXACT.  Both hot paths are float32, but every rounding step of the oracle's
association order is reproduced on the device (fp32 MFMA is an exact k-ordered fmaf chain),
so the tests demand array equality; ``TOL`` documents the float32 tolerance that
BASELINE.json's north star would otherwise allow (2e-5 * sum|w| on CC values, 1e-5 relative on
beams) and is only used to print how far apart a failing pair is.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_CC = 2e-5


def _mf_case(rng, T, S, C, L, N, mv_lo, mv_hi, zero_w=True, zero_data=True):
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    mv = rng.integers(mv_lo, mv_hi + 1, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    if zero_w:
        w[0, 0, :] = 0.0          # a dead station
        w[T - 1, :, 1] = 0.0      # a dead component
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    if zero_data:
        d[S - 1, 0, N // 3: N // 3 + 3 * L] = 0.0   # a gap: CC must be exactly 0 there
    return tp, mv, w, d


def _assert_same(got, want, what):
    if not np.array_equal(got, want):
        diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
        bad = int((got != want).sum())
        raise AssertionError(f"{what}: {bad} of {got.size} values differ, max |diff| = {diff.max():.3e}"
                             f" (float32 tolerance would be {TOL_CC:.0e})")


@pytest.mark.parametrize("network_sum", [True, False])
@pytest.mark.parametrize("L,N", [(32, 9000), (100, 12345), (256, 20000)])
def test_mf_mfma_matches_oracle(oracle_lib, network_sum, L, N):
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(L * 7 + N)
    tp, mv, w, d = _mf_case(rng, 3, 4, 3, L, N, -40, 300)
    got = matched_filter(tp, mv, w, d, 1, arch="gpu", network_sum=network_sum, check_zeros=False)
    want = oracle_lib.matched_filter(tp, mv, w, d, 1, network_sum)
    assert got.shape == want.shape
    _assert_same(got, want, f"MF mfma L={L} N={N} network_sum={network_sum}")
    assert np.isfinite(got).all()


@pytest.mark.parametrize("step", [1, 2, 5])
def test_mf_direct_kernel_matches_oracle(oracle_lib, step):
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(100 + step)
    tp, mv, w, d = _mf_case(rng, 2, 3, 3, 64, 7001, -10, 120)
    got = matched_filter(tp, mv, w, d, step, arch="gpu", check_zeros=False, force_direct=True)
    want = oracle_lib.matched_filter(tp, mv, w, d, step, True)
    _assert_same(got, want, f"MF direct step={step}")


def test_mf_mfma_equals_direct_kernel_on_device():
    """Two independent device implementations of the same convention must agree exactly."""
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(5)
    tp, mv, w, d = _mf_case(rng, 5, 6, 3, 128, 50_000, 0, 800)
    a = matched_filter(tp, mv, w, d, 1, check_zeros=False)
    b = matched_filter(tp, mv, w, d, 1, check_zeros=False, force_direct=True)
    _assert_same(a, b, "MF mfma vs direct")


def test_mf_edge_cases(oracle_lib):
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(9)
    # (a) all-zero weights for one template -> row of zeros; (b) moveout that leaves no valid lag
    tp, mv, w, d = _mf_case(rng, 3, 2, 3, 48, 3000, 0, 50, zero_w=False)
    w[1] = 0.0
    mv[2, 0, 0] = 2990
    got = matched_filter(tp, mv, w, d, 1, check_zeros=False)
    want = oracle_lib.matched_filter(tp, mv, w, d, 1, True)
    _assert_same(got, want, "MF edge: dead template / impossible moveout")
    assert not got[1].any() and not got[2].any()
    # (c) N == L: exactly one lag
    tp1 = tp[:, :, :, :48]
    d1 = d[:, :, :48]
    mv0 = np.zeros_like(mv)
    w1 = np.ones_like(w)
    got = matched_filter(tp1, mv0, w1, d1, 1, check_zeros=False)
    want = oracle_lib.matched_filter(tp1, mv0, w1, d1, 1, True)
    assert got.shape == (3, 1)
    _assert_same(got, want, "MF edge: N == L")
    # (d) zero-energy template and zero-energy data window give exactly 0, never NaN/Inf
    tp2 = tp.copy()
    tp2[0, 0, 0] = 0.0
    d2 = d.copy()
    d2[1] = 0.0
    got = matched_filter(tp2, mv0, w1, d2, 1, check_zeros=False)
    want = oracle_lib.matched_filter(tp2, mv0, w1, d2, 1, True)
    _assert_same(got, want, "MF edge: zero energy")
    assert np.isfinite(got).all()
    # (e) (T, S) moveouts / weights broadcast over components like fast_matched_filter
    got = matched_filter(tp, mv[:, :, 0] % 40, w[:, :, 0] + 0.1, d, 1, check_zeros=False)
    want = oracle_lib.matched_filter(tp, np.repeat((mv[:, :, :1] % 40), 3, 2),
                                     np.repeat(w[:, :, :1] + 0.1, 3, 2), d, 1, True)
    _assert_same(got, want, "MF edge: 2-D moveouts")


def test_mf_autocorrelation_property():
    """A template cut out of the data correlates to exactly the weight sum at its own lag."""
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(11)
    S, C, L, N = 3, 3, 64, 5000
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    i0 = 1234
    mv = rng.integers(0, 100, (1, S, C)).astype(np.int32)
    tp = np.stack([[d[s, c, i0 + mv[0, s, c]: i0 + mv[0, s, c] + L] for c in range(C)]
                   for s in range(S)])[None].copy()
    w = np.full((1, S, C), 1.0 / (S * C), dtype=np.float32)
    cc = matched_filter(tp, mv, w, d, 1, check_zeros=False)
    assert cc[0].argmax() == i0
    assert abs(cc[0, i0] - 1.0) < 1e-5
    assert cc.max() <= 1.0 + 1e-5 and cc.min() >= -1.0 - 1e-5


def _bp_case(rng, K, S, C, P, N, tau_max):
    f = np.abs(rng.standard_normal((S, C, N))).astype(np.float32)
    tau = rng.integers(0, tau_max + 1, (K, S, P)).astype(np.int32)
    wp = rng.random((S, C, P)).astype(np.float32)
    ws = rng.random((K, S)).astype(np.float32)
    ws[rng.random((K, S)) < 0.4] = 0.0
    ws[K // 2] = 0.0  # a source with no station at all
    return f, tau, wp, ws


@pytest.mark.parametrize("oob", ["strict", "flexible"])
@pytest.mark.parametrize("K,S,P,N,tau_max", [(37, 5, 2, 3000, 90), (300, 8, 2, 5000, 400),
                                             (64, 4, 1, 1111, 30), (50, 3, 3, 2500, 2600)])
def test_bp_matches_oracle(oracle_lib, oob, K, S, P, N, tau_max):
    from seismic_bpmf_amd import beamform
    rng = np.random.default_rng(K + N)
    f, tau, wp, ws = _bp_case(rng, K, S, 3, P, N, tau_max)
    mb, ma = beamform(f, tau, wp, ws, device="gpu", reduce="max", out_of_bounds=oob)
    ob, oa = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
    assert mb.dtype == np.float32 and ma.dtype == np.int32 and mb.shape == (N,)
    _assert_same(mb, ob, f"BP maxbeam {oob}")
    assert np.array_equal(ma, oa), f"BP argmax {oob}: {(ma != oa).sum()} differ"
    beam = beamform(f, tau, wp, ws, device="gpu", reduce="none", out_of_bounds=oob)
    obeam = oracle_lib.beamform(f, tau, wp, ws, oob, "none")
    assert beam.shape == (K, N)
    _assert_same(beam, obeam, f"BP full beam {oob}")


def test_bp_tie_rule_lowest_source_wins(oracle_lib):
    from seismic_bpmf_amd import beamform
    rng = np.random.default_rng(3)
    f, tau, wp, ws = _bp_case(rng, 40, 4, 3, 2, 2000, 50)
    # duplicate sources -> exact ties; the lowest index must be reported
    tau[30:] = tau[:10]
    ws[30:] = ws[:10]
    mb, ma = beamform(f, tau, wp, ws, reduce="max")
    ob, oa = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
    assert np.array_equal(ma, oa) and np.array_equal(mb, ob)
    assert ma.max() < 30


def test_bp_int64_moveouts_and_strict_tail(oracle_lib):
    """BPMF hands int64 moveouts (utils.py:1270); the strict tail returns (0, 0)."""
    from seismic_bpmf_amd import beamform
    rng = np.random.default_rng(4)
    f, tau, wp, ws = _bp_case(rng, 20, 4, 3, 2, 1500, 200)
    ws[:] = 1.0 / 4
    mb, ma = beamform(f, tau.astype(np.int64), wp, ws, reduce="max", out_of_bounds="strict")
    tail = 1500 - tau.max(axis=(1, 2)).min()
    assert not mb[tail:].any() and not ma[tail:].any()
    ob, oa = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
    assert np.array_equal(mb, ob) and np.array_equal(ma, oa)


def test_device_resident_api_matches_host_api():
    import torch
    from seismic_bpmf_amd import BeamformerGPU, MatchedFilterGPU, beamform, matched_filter
    rng = np.random.default_rng(21)
    tp, mv, w, d = _mf_case(rng, 4, 3, 3, 64, 30_000, 0, 500)
    mf = MatchedFilterGPU()
    mf.set_data(d)
    a = mf.run(tp, mv, w, 1)
    b = mf.run(tp[:2], mv[:2], w[:2], 1)  # second batch re-uses the prepared data energies
    torch.cuda.synchronize()
    ref = matched_filter(tp, mv, w, d, 1, check_zeros=False)
    assert np.array_equal(a.cpu().numpy(), ref)
    assert np.array_equal(b.cpu().numpy(), ref[:2])
    f, tau, wp, ws = _bp_case(rng, 100, 5, 3, 2, 4000, 300)
    bf = BeamformerGPU(tau, ws)
    beam, arg = bf.run(f, wp, "max", "strict")
    torch.cuda.synchronize()
    rb, ra = beamform(f, tau, wp, ws)
    assert np.array_equal(beam.cpu().numpy(), rb) and np.array_equal(arg.cpu().numpy(), ra)
    # pack / unpack round trip of the multi-GPU exchange keys, and their ordering
    packed = bf.pack_max(beam, arg)
    b2, a2 = bf.unpack_max(packed)
    torch.cuda.synchronize()
    assert torch.equal(b2, beam) and torch.equal(a2, arg)
    x = torch.tensor([1.0, 1.0, 2.0, 0.0], device=beam.device)
    k = torch.tensor([5, 3, 9, 0], dtype=torch.int32, device=beam.device)
    p = bf.pack_max(x, k)
    assert p[1] > p[0] and p[2] > p[1] and p[0] > p[3]
    bf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("batch_kb,piece_kb", [(64, 16), (200, 1000), (1, 1), (10_000, 7)])
@pytest.mark.parametrize("network_sum", [True, False])
def test_mf_host_api_batches_and_pieces(oracle_lib, batch_kb, piece_kb, network_sum, hip_opts):
    """bpmf_mf_run pipelines template batches and pinned pieces; shrink both so that a small case
    crosses every boundary (last batch / last piece shorter, one template per batch, ...)."""
    from seismic_bpmf_amd import matched_filter
    hip_opts("mf.host_batch_kb", batch_kb)
    hip_opts("mf.host_piece_kb", piece_kb)
    rng = np.random.default_rng(batch_kb + piece_kb)
    T, S, C, L, N = 7, 3, 2, 40, 9000 if network_sum else 2500
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    data = rng.standard_normal((S, C, N)).astype(np.float32)
    mv = rng.integers(0, 60, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    got = matched_filter(tp, mv, w, data, 1, arch="gpu", network_sum=network_sum, check_zeros=False)
    want = oracle_lib.matched_filter(tp, mv, w, data, 1, network_sum=network_sum)
    _assert_same(got, want, f"host API batch {batch_kb} KB piece {piece_kb} KB")


@pytest.mark.gpu
def test_host_apis_split_work_over_a_device_list(oracle_lib):
    """device=[...] block-partitions templates / sources over host threads, one per listed device
    (the default None lists every visible GPU); listing the same GPU several times exercises the
    split and the merge on a one-GPU box: results equal the single pass and the oracle, ties and the
    (0, source 0) start included."""
    from seismic_bpmf_amd import beamform, matched_filter
    rng = np.random.default_rng(12)
    T, S, C, L, N = 7, 3, 2, 50, 6000
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    data = rng.standard_normal((S, C, N)).astype(np.float32)
    mv = rng.integers(0, 80, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    want = oracle_lib.matched_filter(tp, mv, w, data, 1)
    for dev in (0, [0, 0], [0, 0, 0], [0] * 9, None):
        got = matched_filter(tp, mv, w, data, 1, arch="gpu", device=dev, check_zeros=False)
        _assert_same(got, want, f"MF device={dev}")
    f, tau, wp, ws = _bp_case(rng, 301, 6, 3, 2, 4000, 120)
    f = np.round(f * 2).astype(np.float32)           # exact ties across blocks
    tau[200] = tau[20]; ws[200] = ws[20]             # identical sources in different blocks
    f[:, :, :40] = 0.0                               # samples where no beam is positive
    for oob in ("strict", "flexible"):
        ob, oa = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
        for dev in (0, [0, 0], [0, 0, 0, 0, 0], None):
            mb, ma = beamform(f, tau, wp, ws, device="gpu", reduce="max", out_of_bounds=oob, device_id=dev)
            _assert_same(mb, ob, f"BP beam device_id={dev} {oob}")
            assert ma.dtype == np.int32 and np.array_equal(ma, oa), (dev, oob)
    full = oracle_lib.beamform(f, tau, wp, ws, "strict", "none")
    _assert_same(beamform(f, tau, wp, ws, device="gpu", reduce="none", device_id=[0, 0, 0]), full, "BP none split")


@pytest.mark.gpu
def test_host_apis_on_several_real_devices_and_device_restore(oracle_lib):
    """(a) Every host entry point restores the caller's current device and binds its own thread
    to the device it is given -- also on a plan-cache hit, where round 1 skipped the hipSetDevice
    (ADVICE r1): beamform() twice with the same tables must give the same, correct result the
    second time (the plan comes from the library's cache).  (b) With more than one GPU visible
    the default device list really spans them; run twice to hit every per-device cache slot."""
    import torch
    from seismic_bpmf_amd import beamform, matched_filter
    from seismic_bpmf_amd import _lib
    rng = np.random.default_rng(31)
    f, tau, wp, ws = _bp_case(rng, 257, 6, 3, 2, 5000, 150)
    ob, oa = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
    n_dev = _lib.device_count()
    before = torch.cuda.current_device()
    for dev in ([0], None, list(range(n_dev)), [n_dev - 1]):
        for _ in range(2):                            # second call: plan-cache hit on every device
            mb, ma = beamform(f, tau, wp, ws, device="gpu", reduce="max", device_id=dev)
            assert np.array_equal(mb, ob) and np.array_equal(ma, oa), dev
            assert torch.cuda.current_device() == before
    tp = rng.standard_normal((5, 3, 2, 40)).astype(np.float32)
    data = rng.standard_normal((3, 2, 5000)).astype(np.float32)
    mv = rng.integers(0, 60, (5, 3, 2)).astype(np.int32)
    w = rng.random((5, 3, 2)).astype(np.float32)
    want = oracle_lib.matched_filter(tp, mv, w, data, 1)
    for dev in (None, [n_dev - 1]):
        assert np.array_equal(matched_filter(tp, mv, w, data, 1, arch="gpu", device=dev, check_zeros=False), want)
        assert torch.cuda.current_device() == before
    with pytest.raises(_lib.BpmfHipError, match="out of range"):
        beamform(f, tau, wp, ws, device="gpu", device_id=[n_dev])
