"""Synthetic workloads of the shapes BASELINE.json names.

No seismic data ships with the reference, so benchmarks and parity tests use seeded synthetic
inputs conditioned the way the reference conditions real ones:
  * MF data: white noise, each channel divided by its std (BPMF/similarity_search.py:181-185);
    templates: short band-limited wavelets, demeaned, divided by their std
    (BPMF/dataset.py:4152-4166); moveouts: P on the vertical, S on the horizontals
    (BPMF/dataset.py:3451-3462); weights "simple", normalised to sum 1 (:288-296, 469-472);
    a few copies of each template are planted in the data so that the CC series has real peaks.
  * BP: homogeneous half-space travel times -> sec_to_samp -> minus per-source minimum
    (BPMF/template_search.py:145-168, 212-214); half-normal "envelopes" with planted events;
    one-hot phase weights Z->P, N/E->S; 10-closest-station source weights, row-normalised
    (template_search.py:779-798, 890-893).
"""
import numpy as np

# name -> sizes of BASELINE.json configs
MF_CONFIGS = {
    "cfg1": dict(T=4, S=8, C=3, L=128, N=180_000),
    "cfg2": dict(T=500, S=20, C=3, L=256, N=8_640_000),
    "cfg4_per_gpu": dict(T=625, S=40, C=3, L=256, N=8_640_000),
}
BP_CONFIGS = {
    "cfg3": dict(grid=(50, 50, 20), S=20, C=3, P=2, N=4_320_000, sr=50.0),
    "cfg5_per_gpu": dict(grid=(125, 125, 8), S=40, C=3, P=2, N=8_640_000, sr=100.0),
}


def sec_to_samp(t, sr, epsilon=0.2):
    """BPMF/utils.py:1258-1271: round towards zero after adding epsilon to |t*sr|."""
    t = np.asarray(t, dtype=np.float64)
    return (np.sign(t) * np.int64(np.abs(t * sr) + epsilon)).astype(np.int64)


def make_mf_inputs(T, S, C, L, N, seed=20260928, max_moveout=1500, n_events=5, step=1):
    """Returns dict(templates, moveouts, weights, data, planted) as float32/int32 NumPy arrays."""
    rng = np.random.default_rng(seed)
    data = rng.standard_normal((S, C, N), dtype=np.float32)
    tmpl = rng.standard_normal((T, S, C, L + 4), dtype=np.float32)
    # 5-tap moving average = crude band limitation
    tmpl = (tmpl[..., 0:L] + tmpl[..., 1:L + 1] + tmpl[..., 2:L + 2] + tmpl[..., 3:L + 3]
            + tmpl[..., 4:L + 4])
    tmpl -= tmpl.mean(axis=-1, keepdims=True)
    tmpl /= tmpl.std(axis=-1, keepdims=True)
    tmpl = tmpl.astype(np.float32)
    mmax = int(min(max_moveout, max(0, (N - L) // 4)))
    mv_p = rng.integers(0, mmax + 1, size=(T, S))
    mv_s = mv_p + rng.integers(0, mmax + 1, size=(T, S))
    moveouts = np.empty((T, S, C), dtype=np.int32)
    moveouts[:, :, 0] = mv_p  # vertical component <-> P
    moveouts[:, :, 1:] = mv_s[:, :, None]  # horizontals <-> S
    weights = np.full((T, S, C), 1.0 / (S * C), dtype=np.float32)
    planted = []
    span = N - L - 2 * mmax - 1
    if span > 10 * L and n_events > 0:
        for t in range(T):
            slots = rng.choice(max(1, span // (4 * L)), size=min(n_events, max(1, span // (4 * L))),
                               replace=False)
            for slot in slots:
                i0 = int(slot) * 4 * L
                i0 -= i0 % step
                amp = rng.uniform(1.5, 4.0)
                for s in range(S):
                    for c in range(C):
                        j = i0 + moveouts[t, s, c]
                        data[s, c, j:j + L] += np.float32(amp) * tmpl[t, s, c]
                planted.append((t, i0 // step))
    data /= data.std(axis=-1, keepdims=True)
    return dict(templates=tmpl, moveouts=moveouts, weights=weights, data=data.astype(np.float32),
                planted=planted)


def make_bp_geometry(grid, S, P=2, sr=50.0, seed=20260928, extent_km=(100.0, 100.0, 30.0),
                     vp=6.0, vs=3.46, n_closest=10, depth_slab=None):
    """Moveout table (K,S,P) int32 and source weights (K,S) float32 for a regular lattice.

    depth_slab = (index, n_levels_total): the lattice is depth levels
    [index * nz, (index + 1) * nz) of a grid with n_levels_total levels (one GPU's tile of the
    1M-source grid of BASELINE configs[4]: slab r of 125 x 125 x 64)."""
    rng = np.random.default_rng(seed)
    nx, ny, nz = grid
    xs = np.linspace(0.0, extent_km[0], nx)
    ys = np.linspace(0.0, extent_km[1], ny)
    zs = np.linspace(0.5, extent_km[2], nz)
    if depth_slab is not None:
        idx, total = depth_slab
        zs = np.linspace(0.5, extent_km[2], total)[(idx * nz) % total:(idx * nz) % total + nz]
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")  # depth fastest
    src = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    sta = np.stack([rng.uniform(0, extent_km[0], S), rng.uniform(0, extent_km[1], S),
                    np.zeros(S)], axis=1)
    dist = np.linalg.norm(src[:, None, :] - sta[None, :, :], axis=2)  # (K, S)
    vel = np.array([vp, vs] + [vs] * max(0, P - 2), dtype=np.float64)[:P]
    tt = dist[:, :, None] / vel[None, None, :]
    tau = sec_to_samp(tt, sr)
    tau -= tau.min(axis=(1, 2), keepdims=True)  # relative to the first arrival
    K = src.shape[0]
    w_src = np.zeros((K, S), dtype=np.float32)
    n_closest = min(n_closest, S)
    # template_search.py:779-798: cut-off = n_closest-th smallest first-phase moveout
    first = tau[:, :, 0]
    cut = np.partition(first, n_closest - 1, axis=1)[:, n_closest - 1]
    w_src[first <= cut[:, None]] = 1.0
    w_src /= w_src.sum(axis=1, keepdims=True)
    return dict(moveouts=tau.astype(np.int32), weights_sources=w_src.astype(np.float32),
                sources=src, stations=sta)


def phase_weights(S, C=3, P=2):
    """One-hot channel->phase map: component 0 (Z) -> P, the others -> S (nb6 cell 52)."""
    w = np.zeros((S, C, P), dtype=np.float32)
    w[:, 0, 0] = 1.0
    if P > 1:
        w[:, 1:, 1] = 1.0
    else:
        w[:, 1:, 0] = 1.0
    return w


def make_bp_features(moveouts, S, C, N, sr=50.0, seed=20260929, n_events=20, amp=8.0):
    """Half-normal envelopes with Gaussian bumps planted along the moveouts of random sources."""
    rng = np.random.default_rng(seed)
    feat = np.abs(rng.standard_normal((S, C, N), dtype=np.float32))
    K, _, P = moveouts.shape
    planted = []
    sig = max(1.0, 0.2 * sr)
    half = int(4 * sig)
    bump = (amp * np.exp(-0.5 * (np.arange(-half, half + 1) / sig) ** 2)).astype(np.float32)
    tmax = int(moveouts.max())
    if N > 2 * (tmax + 2 * half) + 10:
        for _ in range(n_events):
            k0 = int(rng.integers(0, K))
            t0 = int(rng.integers(half, N - tmax - 2 * half))
            for s in range(S):
                for c in range(C):
                    p = 0 if c == 0 else min(1, P - 1)
                    x = t0 + int(moveouts[k0, s, p])
                    feat[s, c, x - half:x + half + 1] += bump
            planted.append((k0, t0))
    return feat, planted
