"""GPU: seeded random sweeps of the adjacent native helpers (grid decimation, kurtosis) and of the
host entry points' work splitting (device lists, plan cache) against the oracle.  Bit-exact.
BPMF_FUZZ_SEEDS=a:b widens the sweep (tools/fuzz_long.sh A B secs fuzz_adjacent, with the file name
given as 5th argument)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fuzz_seeds(default):
    spec = os.environ.get("BPMF_FUZZ_SEEDS")
    if not spec:
        return range(default)
    a, b = spec.split(":")
    return range(int(a), int(b))


@pytest.mark.parametrize("seed", _fuzz_seeds(12))
def test_fuzz_adjacent_grid_decimation(oracle_lib, seed):
    from seismic_bpmf_amd.decimate import find_similar_sources
    rng = np.random.default_rng(41_000 + seed)
    K = int(rng.choice([1, 2, 17, 255, 256, 257, 700, 2_500]))
    S = int(rng.integers(1, 14))
    lon, lat = rng.uniform(0, 1, K).astype(np.float32), rng.uniform(0, 1, K).astype(np.float32)
    sta = rng.uniform(0, 1, (S, 2))
    mv = (np.hypot(lon[:, None] - sta[None, :, 0], lat[:, None] - sta[None, :, 1]) * 20).astype(np.float32)
    if seed % 2 == 0:
        mv = np.round(mv, 1)                  # ties in the argsort and in the sums
    n_cells = int(rng.integers(2, 7))
    cl = np.linspace(-0.01, 1.01, n_cells).astype(np.float32)
    nd = int(rng.integers(1, S + 1))
    thr = float(rng.choice([0.0, 0.05, 0.3, 1.0, 50.0]))
    for method in ("closest", "smallest"):
        red = find_similar_sources(mv, lon, lat, cl, cl, thr, num_stations_for_diff=nd, method=method)
        want = oracle_lib.find_similar_sources(mv, lon, lat, cl, cl, thr, nd, method)
        assert np.array_equal(red, want), f"seed {seed} {method} K={K} S={S} nd={nd} thr={thr} cells={n_cells}: {(red != want).sum()} differ"


@pytest.mark.parametrize("seed", _fuzz_seeds(12))
def test_fuzz_adjacent_kurtosis(oracle_lib, seed):
    from seismic_bpmf_amd.features import kurtosis
    rng = np.random.default_rng(42_000 + seed)
    S, C = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    n = int(rng.choice([1, 5, 64, 1_000, 4_097, 20_000]))
    W = int(rng.choice([1, 2, 3, 37, 256, 1_000, 5_000]))
    x = (rng.standard_normal((S, C, n)) * 10.0 ** rng.integers(-4, 3, (S, C, 1))).astype(np.float32)
    if seed % 3 == 0 and n > 10:
        x[0, 0, n // 3:n // 3 + W + 3] = 0.0          # a flat stretch: zero variance
    got = kurtosis(x, W).cpu().numpy()
    want = oracle_lib.kurtosis(x, W)
    assert got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True), f"seed {seed} S={S} C={C} n={n} W={W}: {(got != want).sum()} differ"


@pytest.mark.parametrize("seed", _fuzz_seeds(12))
def test_fuzz_adjacent_device_lists_and_plan_cache(oracle_lib, seed):
    """The same GPU listed several times splits templates / sources over host threads; beamform
    calls with alternating moveout tables walk the plan cache (hits, evictions)."""
    from seismic_bpmf_amd import beamform, matched_filter
    rng = np.random.default_rng(43_000 + seed)
    n_dev = int(rng.integers(2, 7))
    T, S, C = int(rng.integers(1, 9)), int(rng.integers(1, 4)), int(rng.integers(1, 3))
    L = int(rng.choice([8, 64, 300]))
    N = int(L + rng.choice([0, 900, 5_000]))
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    data = rng.standard_normal((S, C, N)).astype(np.float32)
    mv = rng.integers(-30, 200, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    for ns in (True, False):
        want = oracle_lib.matched_filter(tp, mv, w, data, 1, network_sum=ns)
        got = matched_filter(tp, mv, w, data, 1, arch="gpu", device=[0] * n_dev, network_sum=ns, check_zeros=False)
        assert np.array_equal(got, want), f"seed {seed} MF split over {n_dev} (T={T})"
    K, Sb, P, Nb = int(rng.integers(1, 500)), int(rng.integers(1, 9)), int(rng.choice([1, 2, 2, 3])), int(rng.choice([300, 2_000, 7_000]))
    f = np.round(np.abs(rng.standard_normal((Sb, 2, Nb))) * 2).astype(np.float32)     # exact ties across blocks
    wp = rng.random((Sb, 2, P)).astype(np.float32)
    tables = []
    for _ in range(int(rng.integers(2, 6))):
        tau = rng.integers(0, 150, (K, Sb, P)).astype(np.int32)
        ws = rng.random((K, Sb)).astype(np.float32)
        ws[rng.random((K, Sb)) < 0.3] = 0.0
        tables.append((tau, ws))
    order = rng.integers(0, len(tables), 8)
    want = {}
    for j in order:
        tau, ws = tables[j]
        oob = "strict" if j % 2 == 0 else "flexible"
        if j not in want:
            want[j] = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
        dev = [0] * n_dev if rng.random() < 0.5 else 0
        mb, ma = beamform(f, tau, wp, ws, device="gpu", reduce="max", out_of_bounds=oob, device_id=dev)
        assert np.array_equal(mb, want[j][0]) and np.array_equal(ma, want[j][1]), f"seed {seed} BP table {j} dev={dev}"
