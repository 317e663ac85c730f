"""GPU: the entry points called concurrently from several Python threads (ctypes releases the GIL;
the reference runs its post-CC routine from a ThreadPoolExecutor, BPMF/similarity_search.py:587-593):
every thread gets the result of its own inputs, identical to the serial run and to the oracle."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_threshold_from_a_thread_pool(oracle_lib):
    import torch
    from seismic_bpmf_amd.threshold import ThresholdGPU
    rng = np.random.default_rng(5)
    n, window, overlap = 50_000, 5_000, 0.25
    wn = rng.standard_normal(500).astype(np.float32)
    rows = []
    for r in range(12):
        x = (0.03 * (1 + r) * rng.standard_normal(n)).astype(np.float32)
        if r % 3 == 0:
            x[1000 * r:1000 * r + 700] = 0.0
        rows.append(x)
    want = [oracle_lib.time_dependent_threshold(x, window, 8.0, overlap, wn) for x in rows]

    def one(x):
        th = ThresholdGPU()                      # one instance (workspace) per call, as a pool worker would
        _, full = th.time_dependent_threshold(torch.as_tensor(x, device="cuda"), window, 8.0, overlap=overlap,
                                              white_noise=wn, expand=True)
        return full.cpu().numpy()[0]

    for _ in range(3):
        with ThreadPoolExecutor(max_workers=6) as pool:
            got = list(pool.map(one, rows))
        for r in range(len(rows)):
            assert np.array_equal(got[r], want[r]), r


def test_hot_paths_from_concurrent_threads(oracle_lib):
    """matched_filter / beamform (host-pointer entry points: own streams, own allocations, the BP plan
    cache behind its mutex) and the device detection stage, 4 threads at once, different inputs."""
    import torch
    from seismic_bpmf_amd import beamform, matched_filter, postprocess as pp
    from seismic_bpmf_amd.workflow import beam_detections_device
    rng = np.random.default_rng(6)
    jobs = []
    for j in range(8):
        T, S, C, L, N = 3, 3, 2, int(rng.choice([40, 130, 300])), 9_000 + 512 * j
        tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
        d = rng.standard_normal((S, C, N)).astype(np.float32)
        mv = rng.integers(-40, 300, (T, S, C)).astype(np.int32)
        w = rng.random((T, S, C)).astype(np.float32)
        K, Sb, Nb = 150 + 10 * j, 5, 6_000 + 100 * j
        f = np.abs(rng.standard_normal((Sb, 2, Nb))).astype(np.float32)
        tau = rng.integers(0, 200, (K, Sb, 2)).astype(np.int32)
        wp = rng.random((Sb, 2, 2)).astype(np.float32)
        ws = rng.random((K, Sb)).astype(np.float32)
        jobs.append((tp, mv, w, d, f, tau, wp, ws))
    want = []
    for tp, mv, w, d, f, tau, wp, ws in jobs:
        cc = oracle_lib.matched_filter(tp, mv, w, d, 1)
        mb, ma = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
        pk = pp.find_beam_detections(mb, ma, float(np.quantile(mb, 0.97)), 30)
        want.append((cc, mb, ma, pk))

    def one(job):
        tp, mv, w, d, f, tau, wp, ws = job
        cc = matched_filter(tp, mv, w, d, 1, arch="gpu", device=0, check_zeros=False)
        mb, ma = beamform(f, tau, wp, ws, device="gpu", device_id=0)
        pk = beam_detections_device(torch.as_tensor(mb, device="cuda"), torch.as_tensor(ma, device="cuda"),
                                    mpd=30, threshold=float(np.quantile(mb, 0.97)))
        return cc, mb, ma, pk[:2]

    for _ in range(3):
        with ThreadPoolExecutor(max_workers=4) as pool:
            got = list(pool.map(one, jobs))
        for j, ((cc, mb, ma, pk), (wcc, wmb, wma, wpk)) in enumerate(zip(got, want)):
            assert np.array_equal(cc, wcc) and np.array_equal(mb, wmb) and np.array_equal(ma, wma), j
            assert np.array_equal(pk[0], wpk[0]) and np.array_equal(pk[1], wpk[1]), j


def test_multi_device_calls_from_competing_processes():
    """4 processes x 40 iterations of ~16 host-pointer calls each with the same GPU listed 2-6 times
    (tools/stress/stress_multi.py: every result compared bit for bit with the single-device call of the same
    inputs).  Rounds 2-3 ran every listed device on its own host thread, each creating streams, events
    and allocations per call, and lost about one process per 500 calls to SIGSEGV / SIGABRT when 8
    processes competed for the GPU; the per-device context (csrc/context.h) takes all of that out."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out", "stress_test")
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "stress", "stress_multi.py"), "--procs", "4",
                          "--threads", "1", "--calls", "40", "--out", out, "--timeout", "600"],
                         capture_output=True, text=True, timeout=700)
    assert res.returncode == 0, res.stdout[-4000:] + res.stderr[-2000:]
    assert "exit codes [0, 0, 0, 0]" in res.stdout, res.stdout[-2000:]


def test_eight_threads_of_one_process_drive_multi_device_calls(oracle_lib):
    """One process, 8 Python threads, each calling beamform(device_id=[0] * 6) and
    matched_filter(device=[0] * 4) in a loop: the calls of one device take turns inside the library."""
    from seismic_bpmf_amd import beamform, matched_filter
    rng = np.random.default_rng(77)
    jobs = []
    for j in range(8):
        T, S, C, L, N = 5, 2, 2, int(rng.choice([16, 64, 300])), 3_000 + 256 * j
        tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
        d = rng.standard_normal((S, C, N)).astype(np.float32)
        mv = rng.integers(-30, 200, (T, S, C)).astype(np.int32)
        w = rng.random((T, S, C)).astype(np.float32)
        K, Sb, Nb = 60 + 30 * j, 4, 2_500 + 300 * j
        f = np.round(np.abs(rng.standard_normal((Sb, 2, Nb))) * 2).astype(np.float32)
        tau = rng.integers(0, 150, (K, Sb, 2)).astype(np.int32)
        wp = rng.random((Sb, 2, 2)).astype(np.float32)
        ws = rng.random((K, Sb)).astype(np.float32)
        jobs.append((tp, mv, w, d, f, tau, wp, ws, oracle_lib.matched_filter(tp, mv, w, d, 1),
                     oracle_lib.beamform(f, tau, wp, ws, "strict", "max")))

    def one(job):
        tp, mv, w, d, f, tau, wp, ws, wcc, (wmb, wma) = job
        for _ in range(6):
            cc = matched_filter(tp, mv, w, d, 1, arch="gpu", device=[0] * 4, check_zeros=False)
            mb, ma = beamform(f, tau, wp, ws, device="gpu", device_id=[0] * 6)
            if not (np.array_equal(cc, wcc) and np.array_equal(mb, wmb) and np.array_equal(ma, wma)):
                return False
        return True

    with ThreadPoolExecutor(max_workers=8) as pool:
        assert all(pool.map(one, jobs))


def test_profile_log_pairs_edges_per_thread_and_device():
    """With kernel timing on, beamform(device_id=[0, 0, 0]) logs three launches of the beam kernel, each
    with a positive duration and its device; edges recorded by concurrent threads never pair up."""
    from seismic_bpmf_amd import _lib, beamform
    rng = np.random.default_rng(8)
    K, S, N = 90, 5, 40_000
    f = np.abs(rng.standard_normal((S, 2, N))).astype(np.float32)
    tau = rng.integers(0, 150, (K, S, 2)).astype(np.int32)
    wp = rng.random((S, 2, 2)).astype(np.float32)
    ws = rng.random((K, S)).astype(np.float32)
    _lib.profile_enable(True)
    try:
        beamform(f, tau, wp, ws, device="gpu", device_id=[0, 0, 0])
        ms = _lib.profile_times_ms(_lib.KERNEL_BP_BEAM)
        assert len(ms) == 3 and all(m > 0 for m in ms), ms
        assert _lib.profile_devices(_lib.KERNEL_BP_BEAM) == [0, 0, 0]
        # concurrent threads, each its own call: as many well-formed pairs as calls
        _lib.profile_enable(True)
        with ThreadPoolExecutor(max_workers=4) as pool:
            list(pool.map(lambda _: beamform(f, tau, wp, ws, device="gpu", device_id=0), range(8)))
        ms = _lib.profile_times_ms(_lib.KERNEL_BP_BEAM)
        assert len(ms) == 8 and all(0 < m < 1000 for m in ms), ms
    finally:
        _lib.profile_enable(False)


def test_host_calls_keep_and_release_their_working_set():
    """The host-pointer calls keep a device working set and two pinned pieces per device between calls
    (no allocation per call); bpmf_release_device_memory gives them back; the next call re-creates them."""
    from seismic_bpmf_amd import _lib, matched_filter
    rng = np.random.default_rng(9)
    tp = rng.standard_normal((3, 2, 2, 64)).astype(np.float32)
    d = rng.standard_normal((2, 2, 30_000)).astype(np.float32)
    mv = np.zeros((3, 2, 2), np.int32)
    w = np.ones((3, 2, 2), np.float32)
    first = matched_filter(tp, mv, w, d, 1, arch="gpu", device=0, check_zeros=False)
    dev_b, pin_b = _lib.device_memory_held(0)
    assert dev_b > 0 and pin_b > 0
    again = matched_filter(tp, mv, w, d, 1, arch="gpu", device=0, check_zeros=False)
    assert _lib.device_memory_held(0) == (dev_b, pin_b) and np.array_equal(first, again)
    _lib.release_device_memory(0)
    assert _lib.device_memory_held(0) == (0, 0)
    assert np.array_equal(matched_filter(tp, mv, w, d, 1, arch="gpu", device=0, check_zeros=False), first)
    assert _lib.device_memory_held(0)[0] > 0


def test_cache_limit_gives_a_large_working_set_back_at_the_end_of_the_call(hip_opts):
    """Option host.cache_limit_mb: a host-pointer call whose device working set exceeds the limit frees it when
    it ends (a workflow that makes ONE large host-pointer call and continues on torch-allocated tensors must not
    strand gigabytes outside torch's allocator); package-level release_device_memory / device_memory_held."""
    import seismic_bpmf_amd as sb
    rng = np.random.default_rng(10)
    tp = rng.standard_normal((3, 2, 2, 64)).astype(np.float32)
    d = rng.standard_normal((2, 2, 400_000)).astype(np.float32)      # ~25 MB of working set
    mv = np.zeros((3, 2, 2), np.int32)
    w = np.ones((3, 2, 2), np.float32)
    sb.release_device_memory(0)
    first = sb.matched_filter(tp, mv, w, d, 1, arch="gpu", device=0, check_zeros=False)
    assert sb.device_memory_held(0)[0] > 8 << 20
    hip_opts("host.cache_limit_mb", 8)
    again = sb.matched_filter(tp, mv, w, d, 1, arch="gpu", device=0, check_zeros=False)
    assert sb.device_memory_held(0) == (0, 0) and np.array_equal(first, again)
    hip_opts("host.cache_limit_mb", 4096)                              # a limit the call stays under: kept
    sb.matched_filter(tp, mv, w, d, 1, arch="gpu", device=0, check_zeros=False)
    assert sb.device_memory_held(0)[0] > 8 << 20
    sb.release_device_memory(0)
