"""Replay one seed of tests/test_gpu_fuzz.py::test_bp_random_shapes_signed_moveouts outside pytest and say where
the result differs from the oracle, under a few option settings (python tools/probe_fuzz_seed.py SEED ...)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from oracle import oracle  # noqa: E402
from seismic_bpmf_amd import BeamformerGPU, _lib, beamform  # noqa: E402


def inputs(seed):
    rng = np.random.default_rng(8000 + seed)
    K = int(rng.integers(1, 600))
    S = int(rng.integers(1, 22))
    P = int(rng.choice([1, 2, 2, 2, 2, 3]))
    C = int(rng.integers(1, 4))
    N = int(rng.choice([1, 2, 511, 512, 513, 1024, 3000, 6000, 12000, 20011]))
    lo = -int(rng.choice([0, 0, 1, 3, 40, 300, 700]))
    hi = int(rng.choice([0, 1, 9, 200, 600, 1500, N + 5]))
    f = np.abs(rng.standard_normal((S, C, N))).astype(np.float32)
    if seed % 3 == 0:
        f = np.round(f * 2)
    tau = rng.integers(lo, hi + 1, (K, S, P)).astype(np.int32)
    wp = rng.random((S, C, P)).astype(np.float32)
    if seed % 2 == 0:
        n_close = int(rng.integers(1, S + 1))
        order = np.argsort(tau[:, :, 0], axis=1)
        ws = np.zeros((K, S), np.float32)
        np.put_along_axis(ws, order[:, :n_close], 1.0, axis=1)
        if seed % 4 == 0:
            ws /= ws.sum(axis=1, keepdims=True)
    else:
        ws = rng.random((K, S)).astype(np.float32)
        ws[rng.random((K, S)) < rng.random()] = 0.0
    return f, tau, wp, ws, (K, S, P, C, N, lo, hi)


for seed in [int(x) for x in sys.argv[1:]]:
    f, tau, wp, ws, dims = inputs(seed)
    print(f"seed {seed}: K, S, P, C, N, lo, hi = {dims}; weighted stations per source "
          f"{(ws != 0).sum(1).min()}..{(ws != 0).sum(1).max()}, used tau {tau[ws != 0].min() if (ws != 0).any() else None}.."
          f"{tau[ws != 0].max() if (ws != 0).any() else None}")
    for oob in ("strict", "flexible"):
        ob, oa = oracle.beamform(f, tau, wp, ws, oob, "max")
        for opts in ({}, {"bp.fast": 0}, {"bp.split": 1}, {"bp.direct": 1}):
            with _lib.options(**opts):
                bf = BeamformerGPU(tau, ws)
                info = bf.plan_info()
                b, a = bf.run(f, wp, "max", oob)
                b, a = b.cpu().numpy(), a.cpu().numpy()
                bf.close()
                hb, ha = beamform(f, tau, wp, ws, device="gpu", reduce="max", out_of_bounds=oob, device_id=0)
            bad = np.flatnonzero((b != ob) | (a != oa))
            badh = np.flatnonzero((hb != ob) | (ha != oa))
            print(f"  {oob:8s} {str(opts):18s} resident: {bad.size} differ" + (f" [{bad.min()}..{bad.max()}]" if bad.size else "") +
                  f"; host call: {badh.size} differ" + (f" [{badh.min()}..{badh.max()}]" if badh.size else "") +
                  f"; tile {info['class_tile']} groups {info['class_groups']} n_groups {info['n_groups']} gather {info['gather_bytes']}")
