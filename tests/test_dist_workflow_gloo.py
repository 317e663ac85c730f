"""CPU, gloo, world sizes 2 and 8: the one-process-per-GPU WORKFLOW functions of the product
(workflow.sharded_matched_filter_detections / sharded_backprojection_detections) end to end -- shards,
broadcast of the day from one rank, per-rank detection, all-gather of the variable-length records,
packed-key all-reduce -- against the same detection logic run in ONE process over everything.

The per-rank engines are oracle-backed stand-ins for MatchedFilterGPU / BeamformerGPU (the checker in the
kernels' place: there is no GPU here) and the per-rank detectors are the package's own host mirrors of
the reference's logic (postprocess.time_dependent_threshold_mad + select_cc_indexes,
BPMF/similarity_search.py:187-286, 1079-1113; postprocess.bp_time_dependent_threshold +
find_beam_detections, BPMF/template_search.py:574-627, 1418-1487).  World size 8 uses the SHAPES of one
GPU's share of BASELINE configs[3] / configs[4] scaled down in T, K and N: 40 stations x 3 components,
10-closest-station weights, ragged template costs, duplicated sources across shard boundaries.

The reference's counterpart of the sharded driver is MatchedFilter.run_matched_filter_search
(BPMF/similarity_search.py:726-807: sequential template chunks -- the chunks are the ranks here)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SR = 50.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleMF:
    """MatchedFilterGPU's interface on the CPU oracle."""
    device = torch.device("cpu")

    def set_data(self, data):
        self.data = np.ascontiguousarray(data.numpy() if isinstance(data, torch.Tensor) else data, dtype=np.float32)

    def run(self, templates, moveouts, weights, step=1, network_sum=True):
        from oracle import oracle
        return torch.from_numpy(oracle.matched_filter(np.asarray(templates), np.asarray(moveouts), np.asarray(weights),
                                                      self.data, step, network_sum, num_threads=1))


class OracleBP:
    """BeamformerGPU's interface on the CPU oracle (global source ids, kernel default (0, first id))."""
    device = torch.device("cpu")

    def __init__(self, moveouts, weights_sources, source_id_offset):
        self.mv, self.ws, self.k0 = np.asarray(moveouts), np.asarray(weights_sources), int(source_id_offset)

    def run(self, features, weights_phases, reduce="max", out_of_bounds="strict"):
        from oracle import oracle
        f = features.numpy() if isinstance(features, torch.Tensor) else np.asarray(features)
        b, a = oracle.beamform(f, self.mv, np.asarray(weights_phases), self.ws, out_of_bounds, "max", num_threads=1)
        a = a + self.k0
        a[b == 0] = self.k0 if self.k0 == 0 else a[b == 0]      # (0, 0) only on the shard that holds source 0
        return torch.from_numpy(b), torch.from_numpy(a.astype(np.int32))

    def close(self):
        pass


def mf_detector(cc, moveouts, weights, *, step, window, n_dev, search_win, anomalous_cdf=0.0):
    """Host mirror of the reference's MAD threshold + select_cc_indexes (its anomalous-CDF validation included:
    the keyword travels through sharded_matched_filter_detections' detection_kwargs like cc_detections' own
    anomalous_cdf_at_mean_plus_1sig does), rows -> (idx, cc, thr)."""
    from seismic_bpmf_amd import postprocess as pp
    cc = cc.numpy() if isinstance(cc, torch.Tensor) else cc
    out = {}
    wn = np.random.default_rng(3).standard_normal(cc.shape[1]).astype(np.float32)
    for t in range(cc.shape[0]):
        row = cc[t]
        if not (row != 0).any():
            out[t] = (np.zeros(0, np.int64), np.zeros(0, np.float32), np.zeros(0, np.float32))
            continue
        thr = pp.time_dependent_threshold_mad(row, window, n_dev, overlap=0.5, white_noise=wn).astype(np.float32)
        idx = pp.select_cc_indexes(row, thr, search_win, step=step, sr=SR, data_duration_sec=0.0, n_dev_threshold=n_dev,
                                   min_freq_hz=2.0, data_buffer_sec=0.0, threshold_type="mad", remove_edges=False,
                                   anomalous_cdf_at_mean_plus_1sig=anomalous_cdf)
        out[t] = (idx, row[idx].astype(np.float32), thr[idx])
    return out


def bp_detector(mpd, window, n_dev):
    def run(beam, arg):
        from seismic_bpmf_amd import postprocess as pp
        b, a = beam.numpy(), arg.numpy()
        thr = pp.bp_time_dependent_threshold(b, window, n_dev, overlap=0.75)
        return pp.find_beam_detections(b, a, thr, mpd)
    return run


def mf_case(world):
    sys.path.insert(0, ROOT)
    from seismic_bpmf_amd import postprocess as pp, synthetic as syn
    if world == 8:      # one GPU's share of configs[3], scaled: 40 stations x 3 comp, 10 closest stations weighted
        T, S, C, L, N = 40, 40, 3, 32, 3000
    else:
        T, S, C, L, N = 7, 4, 3, 32, 6000
    m = syn.make_mf_inputs(T=T, S=S, C=C, L=L, N=N, seed=12 + world, max_moveout=60, n_events=3)
    w = m["weights"].copy()
    if world == 8:
        rng = np.random.default_rng(4)
        for t in range(T):                      # 10 "closest" stations per template (similarity_search.py:298-332)
            keep = rng.permutation(S)[: (10 if t % 5 else 40)]       # every fifth template uses all 40
            mask = np.zeros(S, bool)
            mask[keep] = True
            w[t, ~mask] = 0.0
        w[7] = 0.0                              # a template without any weighted channel
        w = pp.normalize_weights(w)
    else:
        w[:3, 1:] = 0.0
        w[5] = 0.0
    m["weights"] = w
    return m


def bp_case(world):
    rng = np.random.default_rng(70 + world)
    if world == 8:      # one GPU's share of configs[4], scaled: 40 stations, 10 closest weighted
        S, C, P, K, N = 40, 3, 2, 1000, 2400
    else:
        S, C, P, K, N = 5, 3, 2, 91, 1500
    f = np.abs(rng.standard_normal((S, C, N))).astype(np.float32)
    tau = rng.integers(0, 120, (K, S, P)).astype(np.int32)
    wp = np.zeros((S, C, P), np.float32)
    wp[:, 0, 0] = 1.0
    wp[:, 1:, 1] = 0.5
    ws = np.zeros((K, S), np.float32)
    for k in range(K):
        ws[k, rng.permutation(S)[: min(10, S)]] = 1.0
    # planted events + sources duplicated across shard boundaries (cross-rank ties)
    for e in range(6):
        k, t0 = int(rng.integers(0, K)), int(rng.integers(200, N - 300))
        for s in np.flatnonzero(ws[k]):
            f[s, 0, t0 + tau[k, s, 0]] += 9.0
            f[s, 1:, t0 + tau[k, s, 1]] += 9.0
    step = K // world
    for r in range(1, world):
        tau[r * step + 1] = tau[3]
        ws[r * step + 1] = ws[3]
    return f, tau, wp, ws


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from seismic_bpmf_amd import workflow
    m = mf_case(world)
    src = world - 1                              # the day lives on the LAST rank only: everyone else gets it by broadcast
    det, info = workflow.sharded_matched_filter_detections(
        m["templates"], m["moveouts"], m["weights"], m["data"] if rank == src else None, engine=OracleMF(),
        detector=mf_detector, data_src=src, step=1, window=600, n_dev=6.0, search_win=40, anomalous_cdf=0.45)
    f, tau, wp, ws = bp_case(world)
    peaks, srcs, beam, arg = workflow.sharded_backprojection_detections(
        f if rank == 0 else None, tau, wp, ws, sr=SR, minimum_interevent_time=1.0, engine_factory=OracleBP,
        detector=bp_detector(50, 400, 8.0), features_src=0)
    q.put((rank, info["templates"], info["records_gathered"],
           {t: tuple(np.asarray(x).tolist() for x in v) for t, v in det.items()},
           np.asarray(peaks).tolist(), np.asarray(srcs).tolist(), beam.numpy(), arg.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_workflows_equal_the_single_process_detections(oracle_lib, world):
    sys.path.insert(0, ROOT)
    from seismic_bpmf_amd import parallel
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # ---- matched filter: one process over all templates
    m = mf_case(world)
    T = m["weights"].shape[0]
    costs = (m["weights"].reshape(T, -1) != 0).sum(axis=1)
    bounds = parallel.shard_bounds_weighted(costs, world)
    assert [tuple(r[1]) for r in results] == [tuple(b) for b in bounds]
    assert bounds != parallel.shard_bounds(T, world)             # the weighting is exercised
    cc = oracle_lib.matched_filter(m["templates"], m["moveouts"], m["weights"], m["data"], 1)
    want = mf_detector(cc, m["moveouts"], m["weights"], step=1, window=600, n_dev=6.0, search_win=40, anomalous_cdf=0.45)
    n_det = sum(len(v[0]) for v in want.values())
    assert n_det >= 3 * T // 4                                    # the planted events are found
    for rank, _, n_rec, det, *_ in results:
        assert n_rec == n_det, rank
        assert sorted(det) == list(range(T))
        for t in range(T):
            idx, val, thr = det[t]
            assert idx == want[t][0].tolist(), (rank, t)
            assert np.array_equal(np.asarray(val, np.float32), want[t][1]) and \
                np.array_equal(np.asarray(thr, np.float32), want[t][2]), (rank, t)
    # ---- backprojection: one pass over the whole grid
    f, tau, wp, ws = bp_case(world)
    ob, oa = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
    wpk, wsrc = bp_detector(50, 400, 8.0)(torch.from_numpy(ob), torch.from_numpy(oa))
    assert len(wpk) >= 4
    for rank, *_, peaks, srcs, beam, arg in results:
        assert np.array_equal(beam, ob) and np.array_equal(arg, oa), rank
        assert peaks == wpk.tolist() and srcs == wsrc.tolist(), rank
    # the duplicated sources of the later shards never win a tie against source 3
    K = tau.shape[0]
    assert not np.isin(oa, [r * (K // world) + 1 for r in range(1, world)]).any()


def test_records_round_trip():
    sys.path.insert(0, ROOT)
    from seismic_bpmf_amd import workflow
    det = {0: (np.array([5, 900]), np.array([0.5, -0.25], np.float32), np.array([0.1, 0.2], np.float32)),
           1: (np.zeros(0, np.int64), np.zeros(0, np.float32), np.zeros(0, np.float32)),
           2: (np.array([2**33]), np.array([np.float32(1e-30)]), np.array([np.float32(3.0)]))}
    rec = workflow.detections_to_records(det, t_offset=10)
    assert rec.shape == (3, 4) and rec[:, 0].tolist() == [10, 10, 12]
    back = workflow.records_to_detections(rec[::-1], 13)
    assert back[10][0].tolist() == [5, 900] and back[10][1].tolist() == [0.5, -0.25]
    assert back[12][0].tolist() == [2**33] and back[12][1][0] == np.float32(1e-30) and back[12][2][0] == 3.0
    assert len(back[11][0]) == 0 and len(back[0][0]) == 0
