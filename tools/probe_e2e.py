"""Probe: the host-pointer calls at full size with the day arriving in pieces vs one upload in front.
python tools/probe_e2e.py [T]   (cfg2 shape otherwise; prints wall times and the library's own breakdown)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import seismic_bpmf_amd as sb  # noqa: E402
from seismic_bpmf_amd import _lib, synthetic as syn  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 500
S, C, L, N = 20, 3, 256, 8_640_000
g = torch.Generator(device="cuda")
g.manual_seed(1)
d = torch.randn((S, C, N), device="cuda", generator=g).cpu().numpy()
tp = torch.randn((T, S, C, L), device="cuda", generator=g).cpu().numpy()
mv = torch.randint(0, 3000, (T, S, C), device="cuda", generator=g, dtype=torch.int32).cpu().numpy()
w = np.full((T, S, C), 1.0 / (S * C), np.float32)
_lib.set_option("mf.verbose", 1)
ref = None
for label, lags in (("warm-up", 131072), ("one upload", 0), ("pieces", 131072), ("one upload", 0), ("pieces", 131072)):
    _lib.set_option("mf.host_piece_lags", lags)
    d_new = d.copy()          # a new day is a new array: memory the runtime has not seen (and page-locked) before
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    cc = sb.matched_filter(tp, mv, w, d_new, 1, arch="gpu", check_zeros=False, device=[0])
    dt = time.perf_counter() - t0
    del d_new
    _lib.profile_enable(False)
    kms = _lib.profile_times_ms(_lib.KERNEL_MF_MAIN)
    print(f"MF {label:10s}: {dt * 1e3:8.1f} ms   ({len(kms)} launches of the main kernel, {sum(kms):.1f} ms in total, first {kms[0]:.2f} last {kms[-1]:.2f})", flush=True)
    if ref is None:
        ref = cc[:, ::4097].copy()
    else:
        assert np.array_equal(ref, cc[:, ::4097])
    del cc
del d, tp
bcfg = syn.BP_CONFIGS["cfg3"]
geo = syn.make_bp_geometry(bcfg["grid"], bcfg["S"], bcfg["P"], bcfg["sr"], seed=3)
feat = torch.randn((bcfg["S"], bcfg["C"], bcfg["N"]), device="cuda", generator=g).abs_().cpu().numpy()
wp = syn.phase_weights(bcfg["S"], bcfg["C"], bcfg["P"])
ref = None
for label, smp in (("warm-up", 131072), ("one upload", 0), ("pieces", 131072), ("one upload", 0), ("pieces", 131072), ("pieces x2", 262144)):
    _lib.set_option("bp.host_piece_samples", smp)
    f_new = feat.copy()
    t0 = time.perf_counter()
    b, a = sb.beamform(f_new, geo["moveouts"], wp, geo["weights_sources"], device="gpu", device_id=[0])
    dt = time.perf_counter() - t0
    print(f"BP {label:10s}: {dt * 1e3:8.1f} ms", flush=True)
    if ref is None:
        ref = (b, a)
    else:
        assert np.array_equal(ref[0], b) and np.array_equal(ref[1], a)
