"""GPU: seeded random sweeps of small shapes against the oracle (bit-exact), to reach the corners
the hand-picked cases miss: N smaller than a tile, one source / one template, moveouts larger than
the data, weights all zero, steps that do not divide anything, P = 1..3, strict and flexible."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(a, b), f"{what}: {(a != b).sum()} of {a.size} values differ"


@pytest.mark.parametrize("seed", range(24))
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
        _same(beamform(f, tau, wp, ws, device="gpu", reduce="none"),
              oracle_lib.beamform(f, tau, wp, ws, "strict", "none"), f"seed {seed} full beam")


@pytest.mark.parametrize("seed", range(24))
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
