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
