"""CPU: the C oracle of the two hot paths against the independent float64 brute force.

The hot-path oracle is "parity unpinned" (no reference source or golden vector exists for
fast_matched_filter / beampower); what CAN be checked here is that the C restatement computes
the mathematical definition the reference's call sites and notebooks pin.
Tolerances are float32 rounding of an L-term / S*P-term sum, written per test.
"""
import numpy as np
import pytest

from oracle import ref_numpy


def _mf_case(rng, T, S, C, L, N, lo, hi):
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    mv = rng.integers(lo, hi + 1, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    w[0, 0] = 0.0
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    d[S - 1, 0, N // 2: N // 2 + 2 * L] = 0.0
    return tp, mv, w, d


@pytest.mark.parametrize("step", [1, 3])
@pytest.mark.parametrize("network_sum", [True, False])
def test_mf_oracle_matches_float64_definition(oracle_lib, step, network_sum):
    rng = np.random.default_rng(step)
    tp, mv, w, d = _mf_case(rng, 3, 4, 3, 40, 900, -7, 60)
    got = oracle_lib.matched_filter(tp, mv, w, d, step, network_sum)
    want = ref_numpy.matched_filter_f64(tp, mv, w, d, step, network_sum)
    assert got.shape == want.shape
    # |num| error <= ~L * eps32 * |t||d|  ->  CC error of a few 1e-7; weights sum < S*C
    assert np.abs(got - want).max() < 5e-6
    # exact zeros: outside the valid lag range and inside the data gap
    assert np.array_equal(got == 0, np.abs(want) < 1e-300) or np.abs(got[np.abs(want) < 1e-300]).max() == 0


def test_mf_oracle_valid_range_and_zero_rows(oracle_lib):
    rng = np.random.default_rng(5)
    tp, mv, w, d = _mf_case(rng, 3, 2, 3, 32, 500, 0, 20)
    mv[1] = -25            # lags 0..24 have a window that starts before the data
    mv[2, 0, 0] = 480      # no lag fits
    cc = oracle_lib.matched_filter(tp, mv, w, d, 1, True)
    assert not cc[1, :25].any() and cc[1, 25] != 0
    assert not cc[2].any()
    w[0] = 0
    cc = oracle_lib.matched_filter(tp, mv, w, d, 1, True)
    assert not cc[0].any()


def test_mf_oracle_window_energy_is_hierarchical_double(oracle_lib):
    """E_d follows the documented chunked prefix sum (chunk = 1024) exactly."""
    import ctypes as C
    rng = np.random.default_rng(6)
    N, L = 5000, 77
    d = (rng.standard_normal((1, N)) * 3).astype(np.float32)
    lib = oracle_lib.load()
    cs = np.zeros((1, N + 1), dtype=np.float64)
    lib.mf_data_csum(d.ctypes.data_as(C.POINTER(C.c_float)), 1, N, cs.ctypes.data_as(C.POINTER(C.c_double)))
    want = np.zeros(N + 1)
    off = 0.0
    for q0 in range(0, N, 1024):
        local = 0.0
        for n in range(q0, min(N, q0 + 1024)):
            local += float(d[0, n]) * float(d[0, n])
            want[n + 1] = off + local
        off += local
    assert np.array_equal(cs[0], want)
    e = np.zeros((1, N - L + 1), dtype=np.float32)
    lib.mf_window_energy(cs.ctypes.data_as(C.POINTER(C.c_double)), 1, N, L, e.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(e[0], (want[L:] - want[:-L]).astype(np.float32))


@pytest.mark.parametrize("oob", ["strict", "flexible"])
def test_bp_oracle_matches_float64_definition(oracle_lib, oob):
    rng = np.random.default_rng(8)
    S, C, P, K, N = 5, 3, 2, 60, 800
    f = np.abs(rng.standard_normal((S, C, N))).astype(np.float32)
    tau = rng.integers(0, 90, (K, S, P)).astype(np.int32)
    wp = rng.random((S, C, P)).astype(np.float32)
    ws = rng.random((K, S)).astype(np.float32)
    ws[:, 1] = 0
    ws[11] = 0
    mb, ma = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
    rb, ra, rbeam = ref_numpy.beamform_f64(f, tau, wp, ws, oob, "max")
    beam = oracle_lib.beamform(f, tau, wp, ws, oob, "none")
    assert np.abs(beam - rbeam).max() < 1e-5 * np.abs(rbeam).max()
    assert np.abs(mb - rb).max() < 1e-5 * rb.max()
    # arg-max may differ only where two beams tie within float32 rounding
    bad = np.flatnonzero(ma != ra)
    for t in bad:
        assert abs(rbeam[ma[t], t] - rbeam[ra[t], t]) < 1e-5 * rb.max()
    if oob == "strict":
        tail = N - tau[ws.any(axis=1)].max()
        assert not mb[N - 1:].any() and mb[: N - 200].all()
        assert tail <= N


def test_bp_oracle_ties_and_floor(oracle_lib):
    S, C, P, K, N = 2, 1, 1, 4, 50
    f = np.ones((S, C, N), dtype=np.float32)
    tau = np.zeros((K, S, P), dtype=np.int32)
    wp = np.ones((S, C, P), dtype=np.float32)
    ws = np.full((K, S), 0.5, dtype=np.float32)
    mb, ma = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
    assert (mb == 1).all() and (ma == 0).all()          # all equal -> lowest index
    mb, ma = oracle_lib.beamform(-f, tau, wp, ws, "strict", "max")
    assert (mb == 0).all() and (ma == 0).all()          # nothing beats the (0, 0) floor
