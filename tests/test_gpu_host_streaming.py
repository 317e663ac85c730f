"""GPU: the host-pointer calls (what BPMF reaches unedited: fast_matched_filter.matched_filter at
BPMF/similarity_search.py:526-533, beampower.beamform at BPMF/template_search.py:549-558) stream the day in
while the first kernels run -- bpmf_mf_run computes its first two template batches over growing ranges of lag
blocks behind the pieces of data that have arrived (csrc/mf.hip), bpmf_bp_run its interior tiles behind the
pieces of features (csrc/bp.hip, BpFeed).  The options mf.host_piece_lags / bp.host_piece_samples shrink the
first piece so that small cases cross many piece boundaries; every result is compared bit for bit with the
oracle and with the same call made with streaming off."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("step,network_sum", [(1, True), (3, True), (1, False)])
@pytest.mark.parametrize("L", [40, 300])
def test_mf_pieces_equal_one_upload(oracle_lib, hip_opts, step, network_sum, L):
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(10 * L + step)
    T, S, C, N = 9, 3, 2, 150_000
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    mv = rng.integers(-300, 2500, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    w[rng.random((T, S, C)) < 0.2] = 0.0
    mv[3, 1, 0] = 10_000_000                       # a zero-weight channel far outside the trace
    w[3, 1, 0] = 0.0
    want = oracle_lib.matched_filter(tp, mv, w, d, step, network_sum)
    row_kb = want[0].nbytes // 1024 + 1
    hip_opts("mf.host_batch_kb", 2 * row_kb)       # two templates per batch: two streamed batches, three that follow
    hip_opts("mf.host_piece_kb", 64)
    hip_opts("mf.host_piece_lags", 4096)           # pieces of 4096, 8192, ... 32768 samples
    got = matched_filter(tp, mv, w, d, step, arch="gpu", device=0, network_sum=network_sum, check_zeros=False)
    assert np.array_equal(got, want)
    hip_opts("mf.host_piece_lags", 0)              # one upload in front of the first kernel
    assert np.array_equal(matched_filter(tp, mv, w, d, step, arch="gpu", device=0, network_sum=network_sum,
                                         check_zeros=False), want)


def test_mf_pieces_one_batch_and_short_templates(oracle_lib, hip_opts):
    """One batch only (no second output buffer), every tiles-per-wave variant of the L <= 257 kernel."""
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(77)
    T, S, C, L, N = 3, 2, 2, 64, 70_000
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    mv = rng.integers(0, 900, (T, S, C)).astype(np.int32)
    w = np.full((T, S, C), 0.25, np.float32)
    want = oracle_lib.matched_filter(tp, mv, w, d, 1)
    hip_opts("mf.host_piece_lags", 4096)
    for ntile in (0, 1, 2, 4):
        hip_opts("mf.tiles_per_wave", ntile)
        for fused in (1, 0):
            hip_opts("mf.fused_prologue", fused)
            assert np.array_equal(matched_filter(tp, mv, w, d, 1, arch="gpu", device=0, check_zeros=False), want), (ntile, fused)


@pytest.mark.parametrize("seed", range(6))
def test_mf_pieces_random_shapes(oracle_lib, hip_opts, seed):
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(900 + seed)
    T, S, C = int(rng.integers(1, 7)), int(rng.integers(1, 4)), int(rng.integers(1, 4))
    L = int(rng.choice([8, 33, 128, 257, 400]))
    N = int(rng.integers(33_000, 90_000))
    step = int(rng.choice([1, 1, 2, 5]))
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    d[0, 0, 5000:5000 + 2 * L] = 0.0
    mv = rng.integers(-2000, 6000, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    w[rng.random((T, S, C)) < 0.3] = 0.0
    want = oracle_lib.matched_filter(tp, mv, w, d, step)
    hip_opts("mf.host_piece_lags", 4096)
    hip_opts("mf.host_batch_kb", int(rng.choice([1, want[0].nbytes // 1024 + 1, 1 << 20])))
    assert np.array_equal(matched_filter(tp, mv, w, d, step, arch="gpu", device=0, check_zeros=False), want)


@pytest.mark.parametrize("oob", ["strict", "flexible"])
@pytest.mark.parametrize("n_used", [3, 20])
def test_bp_pieces_equal_one_upload(oracle_lib, hip_opts, oob, n_used):
    from seismic_bpmf_amd import beamform
    rng = np.random.default_rng(50 + n_used)
    K, S, N = 90, 20, 60_000
    f = np.abs(rng.standard_normal((S, 3, N))).astype(np.float32)
    tau = rng.integers(-40, 700, (K, S, 2)).astype(np.int32)
    wp = rng.random((S, 3, 2)).astype(np.float32)
    ws = np.zeros((K, S), np.float32)
    for k in range(K):
        ws[k, rng.permutation(S)[:n_used]] = rng.random(n_used).astype(np.float32) + 0.1
    wb, wa = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
    hip_opts("bp.host_piece_samples", 1024)        # pieces of 1024, 2048, 4096, 4096, ... samples
    mb, ma = beamform(f, tau, wp, ws, device="gpu", out_of_bounds=oob, device_id=0)
    assert np.array_equal(mb, wb) and np.array_equal(ma, wa)
    hip_opts("bp.host_piece_samples", 0)
    mb, ma = beamform(f, tau, wp, ws, device="gpu", out_of_bounds=oob, device_id=0)
    assert np.array_equal(mb, wb) and np.array_equal(ma, wa)
    hip_opts("bp.host_piece_samples", 1024)        # the other paths take the whole day first: reduce="none", no fast plan
    assert np.array_equal(beamform(f[:, :, :5000], tau, wp, ws, device="gpu", out_of_bounds=oob, reduce="none", device_id=0),
                          oracle_lib.beamform(f[:, :, :5000], tau, wp, ws, oob, "none"))
    hip_opts("bp.fast", 0)
    mb, ma = beamform(f, tau, wp, ws, device="gpu", out_of_bounds=oob, device_id=0)
    assert np.array_equal(mb, wb) and np.array_equal(ma, wa)


def test_pieces_on_several_virtual_devices(oracle_lib, hip_opts):
    """The first device streams in pieces and publishes behind its last one; the others copy device to device."""
    from seismic_bpmf_amd import beamform, matched_filter
    rng = np.random.default_rng(5)
    T, S, C, L, N = 8, 2, 3, 50, 80_000
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    mv = rng.integers(-100, 1500, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    K = 120
    f = np.abs(rng.standard_normal((S, C, N))).astype(np.float32)
    tau = rng.integers(0, 400, (K, S, 2)).astype(np.int32)
    wp = rng.random((S, C, 2)).astype(np.float32)
    ws = rng.random((K, S)).astype(np.float32)
    hip_opts("debug.virtual_devices", 3)
    hip_opts("mf.host_piece_lags", 4096)
    hip_opts("bp.host_piece_samples", 2048)
    assert np.array_equal(matched_filter(tp, mv, w, d, 1, arch="gpu", check_zeros=False), oracle_lib.matched_filter(tp, mv, w, d, 1))
    mb, ma = beamform(f, tau, wp, ws, device="gpu")
    wb, wa = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
    assert np.array_equal(mb, wb) and np.array_equal(ma, wa)


def test_a_day_that_is_already_page_locked_is_not_staged(oracle_lib, hip_opts):
    """Round 6: staged_upload_rows asks hipPointerGetAttributes first -- a page-locked source (a torch pinned tensor seen
    through NumPy) goes to the device without the copy into the context's pinned pieces; same results, next to nothing in host_copy_ms."""
    import torch
    from seismic_bpmf_amd import _lib, beamform, matched_filter
    rng = np.random.default_rng(4242)
    S, C, N = 6, 3, 400_000
    f_pin = torch.empty((S, C, N), dtype=torch.float32).pin_memory()
    f_pin.copy_(torch.from_numpy(np.abs(rng.standard_normal((S, C, N))).astype(np.float32)))
    f = f_pin.numpy()
    tau = rng.integers(0, 300, (500, S, 2)).astype(np.int32)
    wp = np.zeros((S, C, 2), np.float32)
    wp[:, 0, 0] = 1.0
    wp[:, 1:, 1] = 1.0
    ws = (rng.random((500, S)) < 0.7).astype(np.float32)
    hip_opts("bp.host_piece_samples", 131072)
    b_pin, a_pin = beamform(f, tau, wp, ws, device="gpu", device_id=[0])
    st = _lib.host_call_stats()
    assert st["host_copy_ms"] < 0.05, st             # (only the few hundred bytes of weights_phases were staged)
    b_page, a_page = beamform(f.copy(), tau, wp, ws, device="gpu", device_id=[0])
    assert _lib.host_call_stats()["host_copy_ms"] > 0.2
    wb, wa = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
    assert np.array_equal(b_pin, wb) and np.array_equal(a_pin, wa)
    assert np.array_equal(b_page, wb) and np.array_equal(a_page, wa)
    d_pin = torch.empty((4, 3, 300_000), dtype=torch.float32).pin_memory()
    d_pin.copy_(torch.from_numpy(rng.standard_normal((4, 3, 300_000)).astype(np.float32)))
    tp = rng.standard_normal((3, 4, 3, 64)).astype(np.float32)
    mv = rng.integers(0, 200, (3, 4, 3)).astype(np.int32)
    w = rng.random((3, 4, 3)).astype(np.float32)
    got = matched_filter(tp, mv, w, d_pin.numpy(), 1, check_zeros=False, device=[0])
    assert np.array_equal(got, oracle_lib.matched_filter(tp, mv, w, d_pin.numpy(), 1))
