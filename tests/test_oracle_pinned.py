"""CPU: the restated adjacent steps (oracle/adjacent_oracle.c) against golden vectors produced
by the reference's own libc.c / Python (tests/golden/make_goldens.py).  Bit-exact."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


def test_time_dependent_threshold_rms(oracle_lib):
    g = load("tdt_rms.npz")
    for j in range(int(g["n_cases"])):
        thr = oracle_lib.time_dependent_threshold(g[f"x_{j}"], int(g[f"window_{j}"]), float(g[f"num_dev_{j}"]),
                                                  float(g[f"overlap_{j}"]), g[f"white_noise_{j}"])
        want = g[f"thr_{j}"]
        assert thr.dtype == np.float32 and thr.shape == want.shape
        assert np.array_equal(thr, want), f"case {j}: max |diff| {np.abs(thr - want).max()}"


def test_select_cc_indexes_c(oracle_lib):
    g = load("select_cc_indexes_c.npz")
    for j in range(int(g["n_cases"])):
        sel = oracle_lib.select_cc_indexes(g[f"x_{j}"], g[f"thr_{j}"], int(g[f"win_{j}"]))
        assert np.array_equal(sel, g[f"sel_{j}"]), f"case {j}"
        assert sel.sum() > 0


def test_kurtosis(oracle_lib):
    g = load("kurtosis.npz")
    k = oracle_lib.kurtosis(g["signal"], int(g["W"]))
    assert np.array_equal(k, g["kurto"]), np.abs(k - g["kurto"]).max()


@pytest.mark.parametrize("method", ["closest", "smallest"])
@pytest.mark.parametrize("ndiff", [5, 9])
def test_find_similar_sources(oracle_lib, method, ndiff):
    g = load("similar_sources.npz")
    red = oracle_lib.find_similar_sources(g["moveouts"], g["lon"], g["lat"], g["cell_lon"], g["cell_lat"],
                                          float(g[f"thr_{method}_{ndiff}"]), ndiff, method)
    want = g[f"red_{method}_{ndiff}"]
    assert np.array_equal(red, want)
    assert 0 < red.sum() < red.size
