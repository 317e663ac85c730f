"""GPU: seeded random sweeps in the regimes the production kernels are built for, against the oracle
(bit-exact): geometric moveout tables (lattice sources, surface stations, n closest stations) long
enough for interior tiles and several LDS groups -- bp_beam_fast_kernel with its runs, rolling
records and side-stream edge tiles; series of several lag blocks for the matched filter; the
device-resident handles re-used over several days / template batches.
BPMF_FUZZ_SEEDS=a:b widens the sweep."""
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


def _geometry(rng):
    from seismic_bpmf_amd import synthetic as syn
    grid = (int(rng.integers(2, 14)), int(rng.integers(2, 14)), int(rng.integers(1, 9)))
    S = int(rng.choice([3, 8, 12, 20, 33, 40]))
    n_closest = int(rng.choice([1, 2, 3, 5, 8, 10, 13, 16, 17, 20]))
    sr = float(rng.choice([5.0, 20.0, 50.0, 100.0]))
    extent = (float(rng.choice([20.0, 100.0, 300.0])),) * 2 + (float(rng.choice([5.0, 30.0])),)
    g = syn.make_bp_geometry(grid, S, P=2, sr=sr, seed=int(rng.integers(1 << 30)), extent_km=extent,
                             n_closest=n_closest)
    return g["moveouts"], g["weights_sources"], dict(grid=grid, S=S, n_closest=n_closest, sr=sr, extent=extent)


@pytest.mark.parametrize("seed", _fuzz_seeds(16))
def test_fuzz_regimes_bp_geometric_tables(oracle_lib, seed):
    from seismic_bpmf_amd import BeamformerGPU, beamform, synthetic as syn
    rng = np.random.default_rng(51_000 + seed)
    tau, ws, what = _geometry(rng)
    K, S, P = tau.shape
    if seed % 3 == 1:
        ws = (ws * rng.uniform(0.5, 2.0, ws.shape)).astype(np.float32)   # per-station weights: the packed-record flavour
    if seed % 5 == 2:
        ws[int(rng.integers(0, K))] = 0.0                                 # a source without any station
    C = int(rng.choice([2, 3]))
    N = int(rng.choice([700, 2_048, 5_000, 12_345, 40_000]))
    f = np.abs(rng.standard_normal((S, C, N))).astype(np.float32)
    if seed % 4 == 0:
        f = np.round(f * 2)                                               # exact ties between sources
    wp = syn.phase_weights(S, C, 2) if seed % 2 else rng.random((S, C, 2)).astype(np.float32)
    what = f"seed {seed} K={K} N={N} C={C} tau_max={int(tau.max())} {what}"
    bf = BeamformerGPU(tau, ws)
    try:
        for oob in ("strict", "flexible"):
            ob, oa = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
            mb, ma = bf.run(f, wp, reduce="max", out_of_bounds=oob)
            assert np.array_equal(mb.cpu().numpy(), ob), f"{what} {oob}: {(mb.cpu().numpy() != ob).sum()} beams differ"
            assert np.array_equal(ma.cpu().numpy(), oa), f"{what} {oob}: arg-max differs"
        # a second "day" through the same plan, then the host entry point split into source blocks
        f2 = np.abs(rng.standard_normal((S, C, N))).astype(np.float32)
        ob, oa = oracle_lib.beamform(f2, tau, wp, ws, "strict", "max")
        mb, ma = bf.run(f2, wp)
        assert np.array_equal(mb.cpu().numpy(), ob) and np.array_equal(ma.cpu().numpy(), oa), what + " second day"
        mb, ma = beamform(f2, tau, wp, ws, device="gpu", device_id=[0, 0, 0])
        assert np.array_equal(mb, ob) and np.array_equal(ma, oa), what + " three source blocks"
    finally:
        bf.close()


@pytest.mark.parametrize("seed", _fuzz_seeds(10))
def test_fuzz_regimes_mf_several_lag_blocks(oracle_lib, seed):
    from seismic_bpmf_amd import MatchedFilterGPU
    rng = np.random.default_rng(52_000 + seed)
    T = int(rng.integers(1, 6))
    S = int(rng.integers(1, 5))
    C = int(rng.integers(1, 4))
    L = int(rng.choice([33, 128, 256, 300, 700]))
    N = int(rng.choice([30_000, 65_536 + L - 1, 100_003]))
    step = int(rng.choice([1, 1, 1, 2, 3]))
    data = rng.standard_normal((S, C, N)).astype(np.float32)
    if seed % 3 == 0:
        a = int(rng.integers(0, N - 5 * L))
        data[:, :, a:a + 3 * L] = 0.0
    mf = MatchedFilterGPU()
    mf.set_data(data)
    for batch in range(2):                                     # two template batches against the same prepared day
        tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
        mv = rng.integers(-int(rng.choice([0, 3, 500])), int(rng.choice([1, 700, 3_000])), (T, S, C)).astype(np.int32)
        w = rng.random((T, S, C)).astype(np.float32)
        w[rng.random((T, S, C)) < 0.25] = 0.0
        for ns in (True, False):
            got = mf.run(tp, mv, w, step, network_sum=ns).cpu().numpy()
            want = oracle_lib.matched_filter(tp, mv, w, data, step, network_sum=ns)
            assert np.array_equal(got, want), f"seed {seed} batch {batch} ns={ns} T={T} S={S} C={C} L={L} N={N} step={step}: {(got != want).sum()} differ"
