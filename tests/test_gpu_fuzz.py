"""GPU: seeded random sweeps of small shapes against the oracle (bit-exact), to reach the corners
the hand-picked cases miss: N smaller than a tile, one source / one template, moveouts larger than
the data, weights all zero, steps that do not divide anything, P = 1..3, strict and flexible."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fuzz_seeds(default):
    """BPMF_FUZZ_SEEDS=a:b widens the sweep for a long session on the GPU box
    (tools/fuzz_long.sh); the default stays small enough for the driver's run."""
    import os
    spec = os.environ.get("BPMF_FUZZ_SEEDS")
    if not spec:
        return range(default)
    a, b = spec.split(":")
    return range(int(a), int(b))


def _same(a, b, what):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(a, b), f"{what}: {(a != b).sum()} of {a.size} values differ"


@pytest.mark.parametrize("seed", _fuzz_seeds(24))
def test_bp_random_shapes(oracle_lib, seed):
    from seismic_bpmf_amd import beamform
    rng = np.random.default_rng(1000 + seed)
    K = int(rng.integers(1, 400))
    S = int(rng.integers(1, 24))
    P = int(rng.choice([1, 2, 2, 2, 3]))
    C = int(rng.integers(1, 4))
    N = int(rng.choice([1, 7, 63, 500, 513, 1500, 4000]))
    tau_max = int(rng.choice([0, 5, 200, max(1, N // 2), N + 10]))
    f = np.abs(rng.standard_normal((S, C, N))).astype(np.float32)
    if seed % 3 == 0:
        f = np.round(f * 2)                      # integer-valued: exact ties between sources
    tau = rng.integers(0, tau_max + 1, (K, S, P)).astype(np.int32)
    wp = rng.random((S, C, P)).astype(np.float32)
    ws = rng.random((K, S)).astype(np.float32)
    ws[rng.random((K, S)) < rng.random()] = 0.0
    if seed % 5 == 0:
        ws[:] = (ws > 0).astype(np.float32)      # binary weights, as BPMF builds them
    for oob in ("strict", "flexible"):
        mb, ma = beamform(f, tau, wp, ws, device="gpu", reduce="max", out_of_bounds=oob)
        ob, oa = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
        _same(mb, ob, f"seed {seed} {oob} maxbeam (K={K} S={S} P={P} N={N} tau<={tau_max})")
        _same(ma, oa, f"seed {seed} {oob} argmax")
    if K * N <= 400_000:
        for oob in ("strict", "flexible"):
            _same(beamform(f, tau, wp, ws, device="gpu", reduce="none", out_of_bounds=oob),
                  oracle_lib.beamform(f, tau, wp, ws, oob, "none"), f"seed {seed} full beam {oob}")


@pytest.mark.parametrize("seed", _fuzz_seeds(24))
def test_mf_random_shapes(oracle_lib, seed):
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(2000 + seed)
    T = int(rng.integers(1, 9))
    S = int(rng.integers(1, 6))
    C = int(rng.integers(1, 4))
    L = int(rng.choice([1, 2, 15, 16, 17, 100, 256, 257, 258, 300]))
    N = int(L + rng.choice([0, 1, 30, 1000, 5000, 9000]))
    step = int(rng.choice([1, 1, 1, 2, 5, 64, 65, 200]))
    mv_max = int(rng.choice([0, 3, 100, max(1, N // 2), N]))
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    data = rng.standard_normal((S, C, N)).astype(np.float32)
    if seed % 4 == 0:
        data[:, :, N // 3:N // 3 + 2 * L + 5] = 0.0           # a gap: zero-energy windows
    if seed % 6 == 0:
        tp[0, 0] = 0.0                                         # a zero-energy template channel
    mv = rng.integers(0, mv_max + 1, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    w[rng.random((T, S, C)) < 0.3] = 0.0
    if seed % 7 == 0:
        w[T - 1] = 0.0                                         # a template without any channel
    for network_sum in (True, False):
        got = matched_filter(tp, mv, w, data, step, arch="gpu", network_sum=network_sum, check_zeros=False)
        want = oracle_lib.matched_filter(tp, mv, w, data, step, network_sum=network_sum)
        _same(got, want, f"seed {seed} network_sum={network_sum} (T={T} S={S} C={C} L={L} N={N} step={step} mv<={mv_max})")
        assert np.isfinite(got).all()


@pytest.mark.parametrize("seed", _fuzz_seeds(40))
def test_mf_random_shapes_signed_moveouts(oracle_lib, seed):
    """Moveouts of both signs (the reference's are relative to the origin time and can be negative
    for a template cut before it), every kernel family (L up to 2100 -> per-wave, per-workgroup
    17/2, 20/5, 24/9, generic), steps on both sides of the MFMA limit, first valid lags that are not
    multiples of 4 (the buffer-load trap of DESIGN.md)."""
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(7000 + seed)
    T = int(rng.integers(1, 7))
    S = int(rng.integers(1, 5))
    C = int(rng.integers(1, 4))
    L = int(rng.choice([1, 3, 16, 31, 48, 64, 100, 128, 200, 256, 257, 258, 273, 274, 400, 1041, 1042, 1500, 2065, 2066, 2100]))
    N = int(L + rng.choice([0, 1, 2, 3, 5, 255, 1000, 4095, 4096, 4097, 9000, 20000]))
    step = int(rng.choice([1, 1, 1, 1, 2, 3, 4, 7, 16, 17, 64, 65]))
    lo = -int(rng.choice([0, 1, 2, 3, 5, 7, 50, 255, 257, 1023, 1026, max(1, N // 2), N + 3]))
    hi = int(rng.choice([0, 1, 2, 3, 100, 1025, max(1, N // 2), N + 3]))
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    data = rng.standard_normal((S, C, N)).astype(np.float32)
    if seed % 5 == 0:
        a = int(rng.integers(0, max(1, N - 1)))
        data[:, :, a:a + L + int(rng.integers(0, 2 * L + 2))] = 0.0   # a gap: zero-energy windows
    if seed % 9 == 0:
        tp[0, 0] = 0.0
    mv = rng.integers(lo, hi + 1, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    w[rng.random((T, S, C)) < 0.2] = 0.0
    for network_sum in (True, False):
        got = matched_filter(tp, mv, w, data, step, arch="gpu", network_sum=network_sum, check_zeros=False)
        want = oracle_lib.matched_filter(tp, mv, w, data, step, network_sum=network_sum)
        _same(got, want, f"seed {seed} network_sum={network_sum} (T={T} S={S} C={C} L={L} N={N} step={step} mv in [{lo},{hi}])")
        assert np.isfinite(got).all()


@pytest.mark.parametrize("direct", [0, 1])
@pytest.mark.parametrize("seed", _fuzz_seeds(40))
def test_bp_random_shapes_signed_moveouts(oracle_lib, seed, direct, hip_opts):
    """Moveouts of both signs, series spanning several 512-sample tiles (interior tiles run
    bp_beam_fast_kernel, the ends round 1's kernel on the side stream), 1..20 weighted stations,
    uniform and per-station weights.  direct = 1: the same grids through bp_direct.hip (option
    bp.direct), the path of grids that have no LDS plan."""
    from seismic_bpmf_amd import beamform
    if direct:
        if seed >= 16 and seed < 40:
            pytest.skip("the direct path takes the first 16 seeds of the default sweep")
        hip_opts("bp.direct", 1)
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
        # n closest stations with equal weights, as set_weights_sources builds them
        n_close = int(rng.integers(1, S + 1))
        order = np.argsort(tau[:, :, 0], axis=1)
        ws = np.zeros((K, S), np.float32)
        np.put_along_axis(ws, order[:, :n_close], 1.0, axis=1)
        if seed % 4 == 0:
            ws /= ws.sum(axis=1, keepdims=True)
    else:
        ws = rng.random((K, S)).astype(np.float32)
        ws[rng.random((K, S)) < rng.random()] = 0.0
    for oob in ("strict", "flexible"):
        mb, ma = beamform(f, tau, wp, ws, device="gpu", reduce="max", out_of_bounds=oob)
        ob, oa = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
        _same(mb, ob, f"seed {seed} {oob} maxbeam (K={K} S={S} P={P} N={N} tau in [{lo},{hi}])")
        _same(ma, oa, f"seed {seed} {oob} argmax")
    if K * N <= 400_000:
        for oob in ("strict", "flexible"):
            _same(beamform(f, tau, wp, ws, device="gpu", reduce="none", out_of_bounds=oob),
                  oracle_lib.beamform(f, tau, wp, ws, oob, "none"), f"seed {seed} full beam {oob}")


@pytest.mark.parametrize("seed", _fuzz_seeds(30))
def test_bp_random_dense_and_mixed_station_counts(oracle_lib, seed, hip_opts):
    """Round 3: sources with 1..70 weighted stations in one grid (station-count classes of the
    interior kernel on tiles of 512 / 256 / 128 samples, multi-part records, the class merge; above 64
    stations the general kernels alone), random forced tiles and group ranges, moveouts of both signs,
    uniform and per-station weights, ties."""
    from seismic_bpmf_amd import BeamformerGPU
    rng = np.random.default_rng(9100 + seed)
    S = int(rng.choice([18, 21, 24, 33, 40, 48, 64, 70]))
    K = int(rng.integers(20, 400))
    C = int(rng.integers(1, 4))
    N = int(rng.choice([700, 1024, 3000, 5000, 9001]))
    lo = -int(rng.choice([0, 0, 5, 60, 300]))
    hi = int(rng.choice([0, 9, 100, 400]))
    f = np.abs(rng.standard_normal((S, C, N))).astype(np.float32)
    if seed % 3 == 0:
        f = np.round(f * 2)
    tau = rng.integers(lo, hi + 1, (K, S, 2)).astype(np.int32)
    if seed % 2:                                     # smooth moveouts: groups of many sources
        base = rng.integers(lo, hi + 1, (1, S, 2))
        tau = (base + rng.integers(-4, 5, (K, S, 2))).astype(np.int32)
    wp = rng.random((S, C, 2)).astype(np.float32)
    ws = np.zeros((K, S), np.float32)
    mode = seed % 4
    for k in range(K):
        if mode == 0:
            n = S                                     # dense
        elif mode == 1:
            n = int(rng.integers(1, S + 1))           # anything
        elif mode == 2:
            n = int(rng.choice([3, 10, 16, 17, min(S, 40)]))
        else:
            n = int(rng.integers(max(1, S - 6), S + 1))
        sel = rng.choice(S, n, replace=False)
        ws[k, sel] = 0.25 if seed % 5 else rng.uniform(0.1, 1.0, n).astype(np.float32)
    if K > 30:
        ws[7] = 0.0
        tau[11], ws[11] = tau[29], ws[29]             # identical sources: the lower id must win
    if seed % 6 == 0:
        hip_opts("bp.fast_tile", int(rng.choice([256, 128])))
    if seed % 7 == 0:
        hip_opts("bp.split", int(rng.integers(2, 6)))
    bf = BeamformerGPU(tau, ws)
    try:
        for oob in ("strict", "flexible"):
            mb, ma = bf.run(f, wp, "max", oob)
            ob, oa = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
            _same(mb.cpu().numpy(), ob, f"seed {seed} {oob} maxbeam (K={K} S={S} N={N} mode={mode} tau in [{lo},{hi}]) {bf.plan_info()}")
            _same(ma.cpu().numpy(), oa, f"seed {seed} {oob} argmax {bf.plan_info()}")
    finally:
        bf.close()
