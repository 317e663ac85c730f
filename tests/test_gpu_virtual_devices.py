"""The multi-device branches of bpmf_*_run_multi on a box with ONE GPU (option debug.virtual_devices).

`debug.virtual_devices = k` makes the library see k logical devices, each with its own DeviceContext
(streams, working set, call mutex) and -- in the *_run_multi entry points -- its own host thread, all of
them living on the physical GPU(s) that exist (csrc/common.h).  What this reaches that no earlier round
executed on hardware: `run_blocks` with more than one distinct device (csrc/multi.hip: one std::thread per
device), the hand-over of the day of data from the first device to the others (DataFanout, csrc/context.h:
hipMemcpyPeerAsync behind the source's event), the threaded host merge of the per-device beam maxima with
its global source ids and cross-block ties, `bp.compat_first_computed` across device blocks, and the
profile log with several threads inside the library at once.  Every result is compared bit for bit with
the single-device call of the same inputs and with the CPU oracle.

Reference behaviour mirrored: the third-party GPU back-ends split templates / sources over the visible
devices inside one call (BPMF/similarity_search.py:526-533, BPMF/template_search.py:549-558).
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mf_case(rng, T=11, S=3, C=2, L=64, N=40_000):
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    mv = rng.integers(-40, 300, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    w[rng.random((T, S, C)) < 0.25] = 0.0
    w[2] = 0.0                                   # a template without a weighted channel
    return tp, mv, w, d


def _bp_case(rng, K=240, S=6, N=30_000, ties=True):
    # small-integer features and weights: sums are exact, so equal beams (ties) across blocks are common
    f = np.round(np.abs(rng.standard_normal((S, 2, N))) * 2).astype(np.float32)
    tau = rng.integers(0, 200, (K, S, 2)).astype(np.int32)
    wp = np.zeros((S, 2, 2), np.float32)
    wp[:, 0, 0] = 1.0
    wp[:, 1, 1] = 1.0
    ws = (rng.random((K, S)) < 0.7).astype(np.float32)
    if ties:                                     # identical sources in different device blocks
        tau[K // 2 + 3] = tau[5]
        ws[K // 2 + 3] = ws[5]
        tau[K - 2] = tau[1]
        ws[K - 2] = ws[1]
    return f, tau, wp, ws


@pytest.mark.parametrize("k", [2, 4, 8])
@pytest.mark.parametrize("fanout", [1, 0])
def test_mf_on_k_virtual_devices_equals_one_device(oracle_lib, hip_opts, k, fanout):
    from seismic_bpmf_amd import _lib, matched_filter
    rng = np.random.default_rng(100 + k)
    tp, mv, w, d = _mf_case(rng)
    want = oracle_lib.matched_filter(tp, mv, w, d, 1)
    one = matched_filter(tp, mv, w, d, 1, arch="gpu", device=0, check_zeros=False)
    assert np.array_equal(one, want)
    hip_opts("debug.virtual_devices", k)
    hip_opts("multi.peer_fanout", fanout)
    assert _lib.device_count() == k
    multi = matched_filter(tp, mv, w, d, 1, arch="gpu", device=None, check_zeros=False)   # all k devices
    assert np.array_equal(multi, one)
    # per-channel output, a device list in another order with a repeated device, a step
    per_ch = matched_filter(tp, mv, w, d, 3, arch="gpu", device=[k - 1, 0, k - 1], network_sum=False)
    assert np.array_equal(per_ch, oracle_lib.matched_filter(tp, mv, w, d, 3, network_sum=False))
    # every logical device kept its own working set
    held = [_lib.device_memory_held(dv)[0] for dv in range(k)]
    assert all(h > 0 for h in held), held
    _lib.release_device_memory(-1)
    assert _lib.device_memory_held(-1) == (0, 0)


@pytest.mark.parametrize("k", [2, 4, 8])
@pytest.mark.parametrize("oob", ["strict", "flexible"])
def test_bp_max_on_k_virtual_devices_equals_one_device(oracle_lib, hip_opts, k, oob):
    from seismic_bpmf_amd import beamform
    rng = np.random.default_rng(200 + k)
    f, tau, wp, ws = _bp_case(rng)
    wb, wa = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
    hip_opts("debug.virtual_devices", k)
    mb, ma = beamform(f, tau, wp, ws, device="gpu", out_of_bounds=oob, device_id=None)
    assert np.array_equal(mb, wb) and np.array_equal(ma, wa)
    # ties across blocks really occurred: the duplicated sources never win over their lower-id twins
    assert not np.isin(ma, [len(ws) // 2 + 3, len(ws) - 2]).any()
    # reduce="none": every device fills its own rows
    beam = beamform(f[:, :, :3000], tau, wp, ws, device="gpu", out_of_bounds=oob, reduce="none",
                    device_id=list(range(k)))
    assert np.array_equal(beam, oracle_lib.beamform(f[:, :, :3000], tau, wp, ws, oob, "none"))


@pytest.mark.parametrize("k", [2, 8])
def test_bp_first_computed_across_virtual_device_blocks(oracle_lib, hip_opts, k):
    """bp.compat_first_computed with every beam negative: the running max must start from the first
    COMPUTED beam of the whole grid, whichever device block holds it."""
    from seismic_bpmf_amd import beamform
    rng = np.random.default_rng(300 + k)
    f, tau, wp, ws = _bp_case(rng, K=96, N=12_000, ties=False)
    f = -f - 1.0
    tau[:40] += 11_900                    # the first blocks compute (strict) almost nothing
    hip_opts("debug.virtual_devices", k)
    hip_opts("bp.compat_first_computed", 1)
    with oracle_lib.compat(oracle_lib.COMPAT_FIRST_COMPUTED):
        wb, wa = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
    mb, ma = beamform(f, tau, wp, ws, device="gpu", out_of_bounds="strict", device_id=None)
    assert np.array_equal(mb, wb) and np.array_equal(ma, wa)
    assert (wb < 0).any()


def test_profile_log_with_one_thread_per_virtual_device(hip_opts):
    from seismic_bpmf_amd import _lib, beamform, matched_filter
    rng = np.random.default_rng(8)
    f, tau, wp, ws = _bp_case(rng, K=160, N=40_000)
    tp, mv, w, d = _mf_case(rng, T=8)
    w[2] = 1.0
    hip_opts("debug.virtual_devices", 4)
    _lib.profile_enable(True)
    try:
        beamform(f, tau, wp, ws, device="gpu", device_id=None)
        matched_filter(tp, mv, w, d, 1, arch="gpu", device=None, check_zeros=False)
        for which in (_lib.KERNEL_BP_BEAM, _lib.KERNEL_MF_MAIN):
            ms = _lib.profile_times_ms(which)
            assert len(ms) == 4 and all(0 < m < 1000 for m in ms), ms
            assert sorted(_lib.profile_devices(which)) == [0, 1, 2, 3]
    finally:
        _lib.profile_enable(False)


def test_concurrent_multi_device_calls_on_virtual_devices(oracle_lib, hip_opts):
    """Six Python threads, each in a loop of multi-device calls over 4 logical devices in different
    orders: only one call at a time hands its data from device to device (the others upload from the
    host), nobody waits for anybody in a cycle, every result is bit-exact."""
    from seismic_bpmf_amd import beamform, matched_filter
    hip_opts("debug.virtual_devices", 4)
    rng = np.random.default_rng(55)
    jobs = []
    for j in range(6):
        tp, mv, w, d = _mf_case(rng, T=6 + j, N=8_000 + 512 * j)
        f, tau, wp, ws = _bp_case(rng, K=64 + 16 * j, N=6_000 + 100 * j, ties=False)
        order = list(rng.permutation(4))
        jobs.append((tp, mv, w, d, f, tau, wp, ws, [int(x) for x in order],
                     oracle_lib.matched_filter(tp, mv, w, d, 1), oracle_lib.beamform(f, tau, wp, ws, "strict", "max")))

    def one(job):
        tp, mv, w, d, f, tau, wp, ws, order, wcc, (wmb, wma) = job
        for _ in range(5):
            cc = matched_filter(tp, mv, w, d, 1, arch="gpu", device=order, check_zeros=False)
            mb, ma = beamform(f, tau, wp, ws, device="gpu", device_id=order[::-1])
            if not (np.array_equal(cc, wcc) and np.array_equal(mb, wmb) and np.array_equal(ma, wma)):
                return False
        return True

    with ThreadPoolExecutor(max_workers=6) as pool:
        assert all(pool.map(one, jobs))


def test_virtual_device_out_of_range_is_an_error(hip_opts):
    from seismic_bpmf_amd import _lib, matched_filter
    rng = np.random.default_rng(1)
    tp, mv, w, d = _mf_case(rng, T=3, N=4_000)
    hip_opts("debug.virtual_devices", 2)
    with pytest.raises(_lib.BpmfHipError, match="out of range"):
        matched_filter(tp, mv, w, d, 1, arch="gpu", device=[0, 2], check_zeros=False)


@pytest.mark.parametrize("k", [1, 3])
def test_bp_host_call_streams_a_long_day_in_pieces(oracle_lib, hip_opts, k):
    """bpmf_bp_run uploads the day of features in pieces while the interior tiles of the pieces that have
    arrived run (BpFeed, csrc/bp.hip): a series of 11 rounds of the chip is cut into 1 + 2 + 4 + 4 rounds,
    the edge tiles follow the last piece.  Same bits as the resident engine and as the oracle; with three
    (virtual) devices the first one streams from the host and publishes behind its last piece, the others
    copy device to device."""
    from seismic_bpmf_amd import BeamformerGPU, beamform
    rng = np.random.default_rng(400 + k)
    K, S, N = 48, 4, 1_500_000
    f = np.abs(rng.standard_normal((S, 3, N))).astype(np.float32)
    tau = rng.integers(-70, 900, (K, S, 2)).astype(np.int32)          # signed: edge tiles at both ends
    wp = rng.random((S, 3, 2)).astype(np.float32)
    ws = rng.random((K, S)).astype(np.float32)
    ws[rng.random((K, S)) < 0.3] = 0.0
    wb, wa = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
    if k > 1:
        hip_opts("debug.virtual_devices", k)
    mb, ma = beamform(f, tau, wp, ws, device="gpu", out_of_bounds="strict", device_id=None if k > 1 else 0)
    assert np.array_equal(mb, wb) and np.array_equal(ma, wa)
    if k == 1:
        bf = BeamformerGPU(tau, ws, device=0)
        rb, ra = bf.run(f, wp, "max", "strict")
        assert np.array_equal(rb.cpu().numpy(), mb) and np.array_equal(ra.cpu().numpy(), ma)
        bf.close()
        fb, fa = beamform(f, tau, wp, ws, device="gpu", out_of_bounds="flexible", device_id=0)
        ob, oa = oracle_lib.beamform(f, tau, wp, ws, "flexible", "max")
        assert np.array_equal(fb, ob) and np.array_equal(fa, oa)


def test_configs3_sized_day_on_8_virtual_devices(oracle_lib, hip_opts):
    """BASELINE configs[3]'s day -- 40 stations x 3 components x 8 640 000 samples, 4.15 GB -- through
    bpmf_mf_run_multi on EIGHT logical devices (96 of the 5000 templates, ragged 10-closest-station weights: 12
    +- per device): the first device streams the day in pieces, seven copy it device to device, eight host threads
    drain their rows into one (96, n_corr) array.  Equal, bit for bit, to the same call on one device; oracle
    windows on three templates."""
    import torch
    from seismic_bpmf_amd import _lib, matched_filter, synthetic as syn
    T, S, C, L, N = 96, 40, 3, 256, 8_640_000
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    data = torch.randn((S, C, N), device="cuda", generator=g).cpu().numpy()
    inp = syn.make_mf_inputs(T, S, C, L, 30_000, seed=8, n_events=0)
    tmpl, mv, w = inp["templates"], (inp["moveouts"] * 10).astype(np.int32), inp["weights"].copy()
    rng = np.random.default_rng(9)
    for t in range(T):                                   # 10 closest stations weighted, every seventh template all 40
        if t % 7:
            w[t, np.argsort(mv[t, :, 0])[10:], :] = 0.0
    w /= w.reshape(T, -1).sum(axis=1)[:, None, None]
    one = matched_filter(tmpl, mv, w, data, 1, arch="gpu", device=0, check_zeros=False)
    hip_opts("debug.virtual_devices", 8)
    multi = matched_filter(tmpl, mv, w, data, 1, arch="gpu", device=None, check_zeros=False)
    assert np.array_equal(multi, one)
    del multi
    for t in (0, 41, T - 1):
        for i0 in rng.integers(0, N - L - 8000, 2):
            i0 = int(i0)
            seg = data[:, :, i0:i0 + 1500 + L - 1 + int(mv[t].max())]
            want = oracle_lib.matched_filter(tmpl[t:t + 1], mv[t:t + 1], w[t:t + 1], np.ascontiguousarray(seg), 1)[0, :1500]
            assert np.array_equal(one[t, i0:i0 + 1500], want), (t, i0)
    _lib.release_device_memory(-1)


def test_configs4_sized_day_on_8_virtual_devices(oracle_lib, hip_opts):
    """BASELINE configs[4]'s day of features -- 40 stations x 3 components x 8 640 000 samples -- through
    bpmf_bp_run_multi on eight logical devices, 8 x 2 000 sources (10 closest of 40 stations) standing in for the
    8 x 125 000 of the full grid (tests/test_gpu_shares.py runs one such share at full size): streamed upload on the first device, peer copies on seven, eight plans,
    the threaded host merge over 8.64 M samples with global source ids.  Equal to one device; an oracle window."""
    import torch
    from seismic_bpmf_amd import _lib, beamform, synthetic as syn
    S, C, P, N = 40, 3, 2, 8_640_000
    geo = syn.make_bp_geometry((40, 40, 10), S, P, 100.0, seed=5)            # 16 000 sources
    tau, ws = geo["moveouts"], geo["weights_sources"]
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    feat = torch.randn((S, C, N), device="cuda", generator=g).abs_().cpu().numpy()
    wp = syn.phase_weights(S, C, P)
    ob, oa = beamform(feat, tau, wp, ws, device="gpu", device_id=0)
    hip_opts("debug.virtual_devices", 8)
    mb, ma = beamform(feat, tau, wp, ws, device="gpu", device_id=None)
    assert np.array_equal(mb, ob) and np.array_equal(ma, oa)
    assert len(np.unique(ma // 2000)) == 8                                    # every device's block wins somewhere
    i0 = 5_000_000
    W = 600
    seg = np.ascontiguousarray(feat[:, :, i0:i0 + W + int(tau.max()) + 1])
    wb, wa = oracle_lib.beamform(seg, tau, wp, ws, "flexible", "max")
    assert np.array_equal(ob[i0:i0 + W], wb[:W]) and np.array_equal(oa[i0:i0 + W], wa[:W])
    _lib.release_device_memory(-1)


@pytest.mark.parametrize("k", [2, 4])
def test_a_refused_peer_copy_falls_back_to_host_uploads_with_a_note(oracle_lib, hip_opts, k):
    """Round 6: a platform that refuses the device-to-device copy of the day -- the branch no
    box has run between two physical GPUs yet -- must not fail the call: the peers upload from the host, the
    result is the same, and bpmf_last_error() carries a note.  Option debug.fail_peer_copy makes the probe of
    every pair fail."""
    from seismic_bpmf_amd import _lib, beamform, matched_filter
    rng = np.random.default_rng(900 + k)
    tp, mv, w, d = _mf_case(rng)
    want = oracle_lib.matched_filter(tp, mv, w, d, 1)
    hip_opts("debug.virtual_devices", k)
    hip_opts("debug.fail_peer_copy", 1)
    got = matched_filter(tp, mv, w, d, 1, arch="gpu", device=None, check_zeros=False)
    note = _lib.last_error()
    assert np.array_equal(got, want)
    assert note.startswith("note: device-to-device copy of the day") and "uploads from the host" in note, note
    f, tau, wp, ws = _bp_case(rng)
    wb, wa = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
    mb, ma = beamform(f, tau, wp, ws, device="gpu", out_of_bounds="strict", device_id=None)
    assert np.array_equal(mb, wb) and np.array_equal(ma, wa)
    assert _lib.last_error().startswith("note: device-to-device copy")
    hip_opts.reset("debug.fail_peer_copy")
    got = matched_filter(tp, mv, w, d, 1, arch="gpu", device=None, check_zeros=False)
    assert np.array_equal(got, want) and not _lib.last_error().startswith("note:")


@pytest.mark.parametrize("k", [2, 4])
def test_cache_limit_with_peer_fanout_keeps_the_source_copy_until_the_peers_are_through(oracle_lib, hip_opts, k):
    """Round-5 advisor finding: under host.cache_limit_mb the source device gave its working set -- the copy of
    the day its peers were still reading -- back BEFORE it had waited for them.  Many small multi-device calls
    with the limit at 1 MB: every result exact, nothing held afterwards."""
    from seismic_bpmf_amd import _lib, beamform, matched_filter
    rng = np.random.default_rng(950 + k)
    tp, mv, w, d = _mf_case(rng)
    want = oracle_lib.matched_filter(tp, mv, w, d, 1)
    f, tau, wp, ws = _bp_case(rng)
    wb, wa = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
    _lib.release_device_memory(-1)               # (contexts of earlier tests, on more logical devices, keep theirs)
    hip_opts("debug.virtual_devices", k)
    hip_opts("host.cache_limit_mb", 1)
    for _ in range(8):
        assert np.array_equal(matched_filter(tp, mv, w, d, 1, arch="gpu", device=None, check_zeros=False), want)
        mb, ma = beamform(f, tau, wp, ws, device="gpu", out_of_bounds="strict", device_id=None)
        assert np.array_equal(mb, wb) and np.array_equal(ma, wa)
    assert _lib.device_memory_held(-1)[0] == 0
