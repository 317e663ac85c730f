"""GPU: grid decimation against golden vectors from the reference's own libc.c and the pinned oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("method", ["closest", "smallest"])
@pytest.mark.parametrize("ndiff", [5, 9])
def test_matches_reference_goldens(method, ndiff):
    from seismic_bpmf_amd.decimate import find_similar_sources
    g = np.load(os.path.join(GOLD, "similar_sources.npz"))
    red = find_similar_sources(g["moveouts"], g["lon"], g["lat"], g["cell_lon"], g["cell_lat"],
                               float(g[f"thr_{method}_{ndiff}"]), num_stations_for_diff=ndiff, method=method)
    assert red.dtype == bool and np.array_equal(red, g[f"red_{method}_{ndiff}"])


@pytest.mark.parametrize("method", ["closest", "smallest"])
def test_larger_grid_matches_oracle(oracle_lib, method):
    """Several batches of 256 kept sources, several cells, ties in the moveouts."""
    from seismic_bpmf_amd.decimate import find_similar_sources
    rng = np.random.default_rng(12)
    K, S = 3000, 11
    lon, lat = rng.uniform(0, 1, K).astype(np.float32), rng.uniform(0, 1, K).astype(np.float32)
    sta = rng.uniform(0, 1, (S, 2))
    mv = (np.hypot(lon[:, None] - sta[None, :, 0], lat[:, None] - sta[None, :, 1]) * 20).astype(np.float32)
    mv = np.round(mv, 1)                      # quantised -> ties in the argsort and in the sums
    cl = np.linspace(-0.01, 1.01, 4).astype(np.float32)
    for thr, nd in [(0.25, 6), (0.6, 11)]:
        red = find_similar_sources(mv, lon, lat, cl, cl, thr, num_stations_for_diff=nd, method=method)
        want = oracle_lib.find_similar_sources(mv, lon, lat, cl, cl, thr, nd, method)
        assert np.array_equal(red, want), (method, thr, (red != want).sum())
        assert 0.05 * K < red.sum() < 0.98 * K
