import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from seismic_bpmf_amd import BeamformerGPU, synthetic as syn, parallel
geo = syn.make_bp_geometry((125, 125, 64), 40, 2, 100.0, n_closest=10)
tau, ws = geo["moveouts"], geo["weights_sources"]
K = tau.shape[0]; N = 60_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
feat = torch.randn((40, 3, N), device="cuda", generator=g).abs_()
feat = torch.round(feat * 4) / 4          # ties between sources
wp = syn.phase_weights(40, 3, 2)
t0 = time.time(); full = BeamformerGPU(tau, ws); t1 = time.time()
print("plan 1M sources", round(t1 - t0, 2), "s", full.plan_info())
beam, arg = full.run(feat, wp); torch.cuda.synchronize()
t2 = time.time(); beam, arg = full.run(feat, wp); torch.cuda.synchronize(); print("run", round(time.time() - t2, 3), "s")
packed = None
for r in range(8):
    lo, hi = r * K // 8, (r + 1) * K // 8
    b = BeamformerGPU(tau[lo:hi], ws[lo:hi], source_id_offset=lo)
    p = parallel.pack_max_keys(*b.run(feat, wp))
    packed = p if packed is None else torch.maximum(packed, p)
    b.close()
mb, ma = parallel.unpack_max_keys(packed)
print("equal to 8 merged blocks:", bool(torch.equal(mb, beam)), bool(torch.equal(ma, arg)), "max id", int(arg.max()))
# a sample of sources against the oracle
from oracle import oracle
sel = np.sort(np.random.default_rng(0).choice(K, 3000, replace=False))
ob, oa = oracle.beamform(feat.cpu().numpy(), tau[sel], wp, ws[sel], "strict", "max")
sb = BeamformerGPU(tau[sel], ws[sel]); b2, a2 = sb.run(feat, wp)
print("3000 random sources vs oracle:", np.array_equal(b2.cpu().numpy(), ob), np.array_equal(a2.cpu().numpy(), oa))
print("global max >= subset max everywhere:", bool((beam.cpu().numpy() >= ob).all()))
