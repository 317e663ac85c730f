#!/usr/bin/env python3
"""The two hot paths inside their workflows, on synthetic data, end to end on one MI355X.

    python examples/synthetic_day.py            # a few seconds of data per path
    python examples/synthetic_day.py --full     # BASELINE configs[1] and configs[2] (one day each)

Backprojection side (BPMF tutorial notebooks 5-6): raw traces -> saturated envelopes ->
beamform (max over the source grid) -> sliding median/MAD threshold -> peaks -> detections.
Matched-filter side (notebook 8): templates x day of data -> CC sums -> RMS time-dependent
threshold -> candidates -> pair-wise merge -> detections.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import seismic_bpmf_amd as sb  # noqa: E402
from seismic_bpmf_amd import features, synthetic as syn, workflow  # noqa: E402


def tic():
    torch.cuda.synchronize()
    return time.perf_counter()


def backprojection(full):
    sr = 50.0
    grid, S, N = ((50, 50, 20), 20, 4_320_000) if full else ((12, 12, 6), 10, 60_000)
    geo = syn.make_bp_geometry(grid, S, 2, sr)
    K = geo["moveouts"].shape[0]
    rng = np.random.default_rng(1)
    traces = rng.standard_normal((S, 3, N)).astype(np.float32)       # "raw" seismograms
    planted = []
    for _ in range(8):                                                # a few impulsive arrivals
        k0 = int(rng.integers(0, K))
        t0 = int(rng.integers(1000, N - int(geo["moveouts"].max()) - 1000))
        planted.append((k0, t0))
        for s in range(S):
            for c in range(3):
                x = t0 + int(geo["moveouts"][k0, s, 0 if c == 0 else 1])
                traces[s, c, x:x + 40] += 25.0 * rng.standard_normal(40).astype(np.float32)
    t0_ = tic()
    feat, avail = features.saturated_envelopes(traces)
    t1 = tic()
    peaks, sources, maxbeam, _ = workflow.backprojection_detections(
        feat, geo["moveouts"], syn.phase_weights(S, 3, 2), geo["weights_sources"], sr=sr,
        minimum_interevent_time=5.0, threshold_window_dur=600.0 if full else 120.0, n_dev=15.0)
    t2 = tic()
    found = sum(int(np.any(np.abs(peaks - t) <= 30)) for _, t in planted)
    print(f"backprojection: {K} sources x {S} stations x {N} samples | envelopes {t1 - t0_:.3f} s, "
          f"beamform + detections {t2 - t1:.3f} s | {peaks.size} detections, {found}/{len(planted)} "
          f"planted arrivals recovered | stations available: {int(avail.min())}..{int(avail.max())} channels")


def matched_filter(full):
    sr = 100.0
    T, S, L, N = (500, 20, 256, 8_640_000) if full else (16, 8, 128, 400_000)
    mf = syn.make_mf_inputs(T, S, 3, L, N, n_events=3)
    t0 = tic()
    det, cc = workflow.matched_filter_detections(
        mf["templates"], mf["moveouts"], mf["weights"], mf["data"], step=1, sr=sr,
        threshold_window_dur=1800.0 if full else 600.0, minimum_interevent_time=5.0, n_dev=8.0,
        remove_edges=False,
        white_noise=np.random.default_rng(5).standard_normal(500).astype(np.float32))
    t1 = tic()
    n_det = sum(len(v) for v in det.values())
    hit = sum(int(i0 in set(np.asarray(det[t]).tolist())) for t, i0 in mf["planted"])
    print(f"matched filter: {T} templates x {S * 3} channels x {N} samples | {t1 - t0:.3f} s | "
          f"{n_det} detections, {hit}/{len(mf['planted'])} planted events at their exact CC index | "
          f"CC matrix {tuple(cc.shape)} stayed in HBM")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="BASELINE.json configs[1] / configs[2] sizes")
    args = ap.parse_args()
    print("device:", sb.device_info(0))
    backprojection(args.full)
    matched_filter(args.full)
