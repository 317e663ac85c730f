"""Backprojection of short series (the reference's event relocation, BPMF/dataset.py:2174-2216:
reduce="none" over N ~ 1500-3000 samples, all sources of the grid): time per call, device resident."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from seismic_bpmf_amd import BeamformerGPU, synthetic as syn
cfg = syn.BP_CONFIGS["cfg3"]
geo = syn.make_bp_geometry(cfg["grid"], cfg["S"], cfg["P"], cfg["sr"])
tau, ws = geo["moveouts"], geo["weights_sources"]
K = tau.shape[0]
wp = syn.phase_weights(cfg["S"], cfg["C"], cfg["P"])
bf = BeamformerGPU(tau, ws)
for N in (1500, 3000, 6000, 20000, 100_000, 150_000, 400_000):
    feat = torch.randn((cfg["S"], cfg["C"], N), device="cuda").abs_()
    for reduce in (("none", "max") if N <= 20000 else ("max",)):
        out = bf.run(feat, wp, reduce)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); out = bf.run(feat, wp, reduce, out=out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        extra = f", {K * N * 4 / t / 1e9:.0f} GB/s of output" if reduce == "none" else ""
        print(f"K={K} N={N} reduce={reduce}: {t * 1e3:.2f} ms, {K * N / t:.3e} gp x samples/s{extra}")
