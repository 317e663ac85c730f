"""GPU: every kernel variant the dispatchers can pick, each bit-exact against the oracle.

MF: staging-register variants (L <= 273 / 1041 / 2065), the generic kernel beyond, step > 1.
BP: wave-per-source (<= 32 terms/source), uniform-VGPR time-split, readlane kernels with 1 / 2 / 4
metadata blocks (up to 256 terms/source), tile fall-backs, P = 1 / 3 / 5.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    assert a.shape == b.shape, what
    if not np.array_equal(a, b):
        raise AssertionError(f"{what}: {(a != b).sum()} of {a.size} differ, max |diff| "
                             f"{np.abs(a.astype(np.float64) - b).max():.3e}")


@pytest.mark.parametrize("L", [1, 7, 273, 274, 600, 1041, 1500, 2065, 2100])
def test_mf_template_length_variants(oracle_lib, L):
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(L)
    T, S, C, N = 2, 2, 3, max(3 * L + 500, 6000)
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    mv = rng.integers(-20, 200, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    for ns in (True, False):
        got = matched_filter(tp, mv, w, d, 1, check_zeros=False, network_sum=ns)
        _same(got, oracle_lib.matched_filter(tp, mv, w, d, 1, ns), f"MF L={L} network_sum={ns}")


@pytest.mark.parametrize("wave_kernel", [0, 1])
def test_mf_both_mfma_kernels(oracle_lib, wave_kernel, hip_opts):
    """L <= 257 defaults to the independent-wave kernel; the workgroup kernel must agree too."""
    from seismic_bpmf_amd import matched_filter
    hip_opts("mf.wave_kernel", wave_kernel)
    rng = np.random.default_rng(77)
    T, S, C, L, N = 3, 5, 3, 200, 30_000
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    mv = rng.integers(-50, 900, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    w[1, 2] = 0.0
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    for ns in (True, False):
        _same(matched_filter(tp, mv, w, d, 1, check_zeros=False, network_sum=ns),
              oracle_lib.matched_filter(tp, mv, w, d, 1, ns), f"wave_kernel={wave_kernel} ns={ns}")


@pytest.mark.parametrize("ntile", [1, 2, 4])
@pytest.mark.parametrize("L", [1, 17, 128, 241, 257])
def test_mf_tiles_per_wave(oracle_lib, ntile, L, hip_opts):
    """mf.tiles_per_wave: the per-wave kernel with 1, 2 or 4 tiles of 256 lags per wave (small problems
    take fewer lags per wave so that every SIMD gets several waves: BASELINE configs[0]); the same bits
    whatever the split -- negative moveouts, zero-weight channels, steps 1 and 3, both layouts, series
    shorter than one workgroup's lags and series that end inside a tile."""
    from seismic_bpmf_amd import matched_filter
    hip_opts("mf.tiles_per_wave", ntile)
    rng = np.random.default_rng(50 * ntile + L)
    for N in (L + 5, 700, 1024 + L, 9000):
        if N < L:
            continue
        T, S, C = 3, 3, 2
        tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
        mv = rng.integers(-70, 400, (T, S, C)).astype(np.int32)
        w = rng.random((T, S, C)).astype(np.float32)
        w[1, 2] = 0.0
        d = rng.standard_normal((S, C, N)).astype(np.float32)
        for step in (1, 3):
            for ns in (True, False):
                _same(matched_filter(tp, mv, w, d, step, check_zeros=False, network_sum=ns),
                      oracle_lib.matched_filter(tp, mv, w, d, step, ns), f"ntile={ntile} L={L} N={N} step={step} ns={ns}")


@pytest.mark.parametrize("fused", [0, 1])
@pytest.mark.parametrize("ntile", [1, 2])
def test_mf_fused_prologue(oracle_lib, fused, ntile, hip_opts):
    """mf.fused_prologue: with fewer than 4 tiles per wave and at most 256 channels the per-wave kernel works the
    template norms, the compacted channel records and the lag range out itself (one launch); with the option
    off, or beyond 256 channels, mf_prologue_kernel runs first.  Same bits either way: channel counts around
    a wave (63 / 64 / 65) and at the limit (256 / 257), channels without weight in every position (first, last,
    whole waves, all of a template), -0.0 weights, negative moveouts, template lengths around the 16-sample
    trips of the energy loop, steps 1 and 3, both layouts, the two MF compat switches."""
    from seismic_bpmf_amd import matched_filter
    hip_opts("mf.tiles_per_wave", ntile)
    hip_opts("mf.fused_prologue", fused)
    rng = np.random.default_rng(900 + 10 * ntile + fused)
    shapes = [(3, 1, 1, 5), (3, 7, 9, 16), (3, 8, 8, 33), (2, 13, 5, 47), (2, 64, 4, 15), (2, 257, 1, 31),
              (4, 8, 3, 128), (3, 2, 1, 257), (3, 21, 3, 100)]
    for T, S, C, L in shapes:
        N = int(rng.integers(L + 300, 2600))
        tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
        mv = rng.integers(-90, 250, (T, S, C)).astype(np.int32)
        w = rng.random((T, S, C)).astype(np.float32)
        w[rng.random((T, S, C)) < 0.3] = 0.0
        w.reshape(T, -1)[0, 0] = 0.0
        w.reshape(T, -1)[0, -1] = 0.0
        w.reshape(T, -1)[1, : min(S * C, 70)] = 0.0            # a whole wave of threads without a used channel
        if S * C > 2:
            w.reshape(T, -1)[1, 1] = -0.0
        w[-1] = 0.0                                             # a template without any used channel
        d = rng.standard_normal((S, C, N)).astype(np.float32)
        for step in (1, 3):
            for ns in (True, False):
                if not ns and T * S * C * N > 3_000_000:
                    continue
                _same(matched_filter(tp, mv, w, d, step, check_zeros=False, network_sum=ns),
                      oracle_lib.matched_filter(tp, mv, w, d, step, ns),
                      f"fused={fused} ntile={ntile} T={T} S={S} C={C} L={L} N={N} step={step} ns={ns}")
    T, S, C, L, N = 3, 8, 3, 70, 1900
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    mv = rng.integers(-40, 200, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    w[1, 2] = 0.0
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    for opt, flag in (("mf.compat_exclusive_last_lag", oracle_lib.COMPAT_EXCLUSIVE_LAST_LAG),
                      ("mf.compat_sqrt_norm", oracle_lib.COMPAT_SQRT_NORM)):
        with oracle_lib.compat(flag):
            want = oracle_lib.matched_filter(tp, mv, w, d, 1)
        hip_opts(opt, 1)
        _same(matched_filter(tp, mv, w, d, 1, check_zeros=False), want, f"fused={fused} ntile={ntile} {opt}")
        hip_opts(opt, 0)


def test_mf_small_problems_pick_fewer_lags_per_wave(oracle_lib):
    """BASELINE configs[0]'s shape (4 templates, one hour at 50 Hz) takes the one-tile kernel by itself
    and still equals the oracle; a day does not change kernels."""
    from seismic_bpmf_amd import matched_filter, synthetic as syn
    m = syn.make_mf_inputs(T=4, S=8, C=3, L=128, N=60_000, seed=2, max_moveout=300, n_events=2)
    got = matched_filter(m["templates"], m["moveouts"], m["weights"], m["data"], 1, check_zeros=False)
    _same(got, oracle_lib.matched_filter(m["templates"], m["moveouts"], m["weights"], m["data"], 1), "configs[0]-shaped")


@pytest.mark.parametrize("L", [48, 130, 256, 257])
@pytest.mark.parametrize("step", [1, 3])
def test_mf_negative_moveouts_zero_weights_steps(oracle_lib, L, step):
    """Negative moveouts, zero-weight channels, both output layouts, steps 1 and 3 at four template
    lengths (the inputs that exposed the buffer-load zero-fill defect in round 2, when they drove
    an LDS-DMA staging experiment since removed)."""
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(1000 + L + step)
    T, S, C, N = 3, 4, 3, 41_000
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    mv = rng.integers(-300, 2500, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    w[1, 2] = 0.0
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    for ns in (True, False):
        _same(matched_filter(tp, mv, w, d, step, check_zeros=False, network_sum=ns),
              oracle_lib.matched_filter(tp, mv, w, d, step, ns), f"L={L} step={step} ns={ns}")


@pytest.mark.parametrize("L", [48, 64, 300, 1100])
@pytest.mark.parametrize("first", [-1, -2, -3, -74, -253, -271, -1023, -1025])
def test_mf_first_samples_of_the_trace_with_negative_moveouts(oracle_lib, L, first, hip_opts):
    """A template whose most negative moveout is not a multiple of 4 has its first valid lags read
    samples 0..2 of the trace through staging loads whose neighbours lie before the trace: on gfx950 a
    raw buffer load WITH an immediate offset zeroes the whole 4-lane group there
    (tools/ubench/buffer_neg.hip), which rounds 1 and early 2 got wrong by up to 3 CCs per template.
    Every MFMA kernel family (per-wave, per-workgroup), both output layouts."""
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(abs(first) * 7 + L)
    T, S, C, N = 2, 3, 3, 12_000
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    mv = rng.integers(0, 900, (T, S, C)).astype(np.int32)
    mv[0, 1, 1] = first
    mv[1, 2, 0] = first + 1 if first < -1 else first
    w = (rng.random((T, S, C)) + 0.1).astype(np.float32)
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    for wave in (1, 0):
        hip_opts("mf.wave_kernel", wave)
        for step in (1, 2):
            for ns in (True, False):
                _same(matched_filter(tp, mv, w, d, step, check_zeros=False, network_sum=ns),
                      oracle_lib.matched_filter(tp, mv, w, d, step, ns),
                      f"first={first} L={L} wave={wave} step={step} ns={ns}")


@pytest.mark.parametrize("L,N,step,lo", [(1, 2, 2, -1026), (3, 5, 1, -1023), (1, 1, 1, -1026), (1, 3, 1, -1023),
                                         (64, 70, 1, -5000), (300, 305, 1, -5000), (300, 9000, 3, -20000),
                                         (1100, 1101, 1, -70000)])
def test_mf_no_valid_lag_at_all(oracle_lib, L, N, step, lo):
    """Every template window lies before the trace: the valid range is empty (stored first > last)
    and the result all zeros.  The MFMA kernels used to take a workgroup that straddles both ends of
    the inverted range for a valid one and load norms at lag + moveout, far outside the table -- a
    GPU memory fault found by the long fuzz session (tools/fuzz_long.sh, seeds 59, 266, 1106, 1277)."""
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(L * 31 + N)
    T, S, C = 3, 4, 2
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    mv = rng.integers(lo, lo // 2, (T, S, C)).astype(np.int32)
    w = (rng.random((T, S, C)) + 0.1).astype(np.float32)
    w[0, 0, 0] = 0.0
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    for ns in (True, False):
        got = matched_filter(tp, mv, w, d, step, check_zeros=False, network_sum=ns)
        _same(got, oracle_lib.matched_filter(tp, mv, w, d, step, ns), f"L={L} N={N} step={step} ns={ns}")
        assert not got.any()


@pytest.mark.parametrize("step", [2, 3, 7, 16, 17, 50])
@pytest.mark.parametrize("L", [64, 400])
def test_mf_step_greater_than_one(oracle_lib, step, L):
    """step <= 16 runs on the MFMA kernels (every offset evaluated, multiples of step kept);
    larger steps on the generic kernel.  Negative moveouts exercise the ceil in the first valid lag."""
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(step * 100 + L)
    T, S, C, N = 3, 3, 3, 9000 + step
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    mv = rng.integers(-37, 300, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    for ns in (True, False):
        got = matched_filter(tp, mv, w, d, step, check_zeros=False, network_sum=ns)
        _same(got, oracle_lib.matched_filter(tp, mv, w, d, step, ns), f"step={step} L={L} ns={ns}")


def test_mf_lag_block_boundaries(oracle_lib):
    """n_corr around multiples of the 4096-lag workgroup / 1024-lag wave / 256-lag tile."""
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(0)
    L = 48
    for n_corr in (1, 255, 256, 257, 1023, 1025, 4095, 4096, 4097, 8193):
        N = n_corr + L - 1
        tp = rng.standard_normal((2, 2, 2, L)).astype(np.float32)
        mv = np.zeros((2, 2, 2), np.int32)
        w = np.full((2, 2, 2), 0.25, np.float32)
        d = rng.standard_normal((2, 2, N)).astype(np.float32)
        _same(matched_filter(tp, mv, w, d, 1, check_zeros=False), oracle_lib.matched_filter(tp, mv, w, d, 1),
              f"n_corr={n_corr}")


def test_mf_many_templates_and_channels(oracle_lib):
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(2)
    T, S, C, L, N = 37, 7, 3, 40, 5000
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    mv = rng.integers(0, 400, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    w[rng.random((T, S, C)) < 0.5] = 0.0   # "closest stations"-style sparse weights
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    _same(matched_filter(tp, mv, w, d, 1, check_zeros=False), oracle_lib.matched_filter(tp, mv, w, d, 1), "MF sparse")


def _bp_inputs(rng, K, S, P, N, tau_max, density):
    f = np.abs(rng.standard_normal((S, 3, N))).astype(np.float32)
    tau = rng.integers(0, tau_max + 1, (K, S, P)).astype(np.int32)
    wp = rng.random((S, 3, P)).astype(np.float32)
    ws = rng.random((K, S)).astype(np.float32)
    ws[rng.random((K, S)) > density] = 0.0
    return f, tau, wp, ws


def _bp_check(oracle_lib, f, tau, wp, ws, what):
    from seismic_bpmf_amd import beamform
    for oob in ("strict", "flexible"):
        mb, ma = beamform(f, tau, wp, ws, reduce="max", out_of_bounds=oob)
        ob, oa = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
        _same(mb, ob, f"{what} {oob} maxbeam")
        assert np.array_equal(ma, oa), f"{what} {oob}: {(ma != oa).sum()} arg-max differ"
    _same(beamform(f, tau, wp, ws, reduce="none"), oracle_lib.beamform(f, tau, wp, ws, "strict", "none"),
          f"{what} full beam")


@pytest.mark.parametrize("S,P,density,label", [(2, 2, 1.0, "2 stations"), (4, 2, 1.0, "4 stations"), (3, 2, 0.7, "<= 3 stations"),
                                               (6, 2, 0.6, "wps <=8..16 terms"), (14, 2, 1.0, "wps 28 terms"),
                                               (20, 2, 1.0, "readlane 40 terms"), (40, 2, 1.0, "readlane 80 terms"),
                                               (45, 3, 1.0, "readlane 135 terms"), (9, 1, 0.8, "P=1"),
                                               (6, 5, 1.0, "P=5 prestack_any")])
def test_bp_term_count_variants(oracle_lib, S, P, density, label):
    rng = np.random.default_rng(S * 10 + P)
    f, tau, wp, ws = _bp_inputs(rng, 120, S, P, 2500, 150, density)
    _bp_check(oracle_lib, f, tau, wp, ws, label)


@pytest.mark.parametrize("env", [{"bp.wps": 0}, {"bp.uvgpr": 0}, {"bp.packed": 0},
                                 {"bp.tpt": 1, "bp.wps": 0}, {"bp.tpt": 4, "bp.wps": 0},
                                 {"bp.reorder": 0}, {"bp.lds_kb": 24}, {"bp.max_group": 5},
                                 {"bp.dual": 0}, {"bp.dual": 0, "bp.wpb": 8},
                                 {"bp.dual": 0, "bp.lds_kb": 24}, {"bp.max_group": 3}])
def test_bp_kernel_and_plan_knobs(oracle_lib, env, hip_opts):
    for k, v in env.items():
        hip_opts(k, v)
    rng = np.random.default_rng(99)
    f, tau, wp, ws = _bp_inputs(rng, 200, 8, 2, 3000, 250, 0.7)
    _bp_check(oracle_lib, f, tau, wp, ws, str(env))


@pytest.mark.parametrize("dual,S,S_used,gather,waves,tile", [(1, 20, 10, 8, 16, 512), (0, 20, 10, 4, 24, 512), (1, 20, 16, 8, 16, 512),
                                                             (1, 20, 17, 8, 16, 256), (1, 20, 20, 8, 16, 256),
                                                             (1, 34, 31, 8, 16, 256), (1, 34, 32, 8, 16, 256),
                                                             (1, 34, 33, 8, 16, 128), (0, 34, 33, 4, 8, 512),
                                                             (0, 34, 31, 4, 16, 512)])
def test_bp_plan_info_and_gather_width(oracle_lib, dual, S, S_used, gather, waves, tile, hip_opts):
    """Dual (8-byte gather) plans whatever the number of weighted stations per source -- tile 512 up to
    16 stations, 256 up to 32, 128 beyond (round 3; rounds 1-2 fell back to 4-byte gathers from 17
    stations on) --, 4-byte gathers with bp.dual = 0; every variant must give the oracle's result,
    ties included (many equal beams: integer-valued features)."""
    import torch
    from seismic_bpmf_amd import BeamformerGPU
    hip_opts("bp.dual", dual)
    rng = np.random.default_rng(5 + S_used)
    K, P, N = 400, 2, 4000
    f = rng.integers(0, 3, (S, 3, N)).astype(np.float32)      # exact ties between sources
    tau = rng.integers(0, 120, (K, S, P)).astype(np.int32)
    tau[K // 3] = tau[K // 3 + 7]                               # two identical sources
    wp = np.ones((S, 3, P), np.float32)
    ws = np.zeros((K, S), np.float32)
    for k in range(K):
        ws[k, rng.choice(S, S_used, replace=False)] = 1.0
    ws[K // 3] = ws[K // 3 + 7]
    bf = BeamformerGPU(tau, ws)
    info = bf.plan_info()
    # (`tile`: the largest the class may take; the cost model goes smaller when the windows of this
    # random geometry -- every group touches every station -- only fit tiny groups)
    assert info["gather_bytes"] == gather and info["n_groups"] >= 1, info
    # (33-64 stations: tile 128, or several LDS residencies per group at tile 256, whichever the model prices lower)
    assert (info["tile"] == tile or (dual and info["tile"] in (256, 128) and info["tile"] < tile) or
            (dual and S_used > 32 and info["tile"] == 256)), info
    assert info["waves_per_cu"] == waves, info
    assert info["n_classes"] == (1 if dual else 0), info
    for oob in ("strict", "flexible"):
        mb, ma = bf.run(torch.as_tensor(f), wp, "max", oob)
        ob, oa = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
        _same(mb.cpu().numpy(), ob, f"dual={dual} S_used={S_used} {oob} maxbeam")
        assert np.array_equal(ma.cpu().numpy(), oa)
    bf.close()


def test_bp_huge_moveout_spread_falls_back_to_smaller_tiles(oracle_lib):
    """Moveouts spread over ~30 000 samples: windows only fit with the smallest tile."""
    rng = np.random.default_rng(7)
    f, tau, wp, ws = _bp_inputs(rng, 30, 6, 2, 40_000, 30_000, 1.0)
    _bp_check(oracle_lib, f, tau, wp, ws, "huge spread")


def test_bad_arguments_raise_with_a_message():
    import seismic_bpmf_amd as sb
    from seismic_bpmf_amd import _lib
    tp = np.zeros((1, 1, 1, 100), np.float32)
    with pytest.raises(ValueError):
        sb.matched_filter(tp, np.zeros((1, 1, 1)), np.ones((1, 1, 1)), np.zeros((1, 1, 50), np.float32), 1)
    with pytest.raises(ValueError, match="step"):
        sb.matched_filter(tp, np.zeros((1, 1, 1)), np.ones((1, 1, 1)), np.zeros((1, 1, 500), np.float32), 0)
    with pytest.raises(_lib.BpmfHipError, match="unknown option"):
        _lib.set_option("bp.no_such_option", 1)


# ------------------------------------------ grids without an LDS plan (bp_direct.hip) ---
@pytest.mark.parametrize("S,P,K,N,tau_max,density,label", [
    (150, 2, 6, 3000, 200, 1.0, "150 stations x 2 phases: 300 terms per source"),
    (90, 3, 5, 2500, 120, 1.0, "270 terms, three phases"),
    (200, 2, 4, 2000, 300, 0.6, "200 stations, sparse weights"),
    (400, 1, 3, 1500, 100, 1.0, "400 stations, one phase")])
def test_bp_grids_without_an_lds_plan_run_the_direct_kernel(oracle_lib, S, P, K, N, tau_max, density, label):
    """More station-phase rows than the 160 KB of LDS hold windows for (or than the 256-term tables of
    the planned kernels): rounds 1-2 failed with "windows do not fit in LDS"; the reference has no such
    limit (beampower.beamform at template_search.py:549-569).  Same results as the oracle, bit for bit."""
    from seismic_bpmf_amd import BeamformerGPU
    rng = np.random.default_rng(S + P)
    f, tau, wp, ws = _bp_inputs(rng, K, S, P, N, tau_max, density)
    ws[K - 1] = 0.0                                   # a source without any station
    tau[0] -= tau_max // 2                            # negative moveouts
    bf = BeamformerGPU(tau, ws)
    info = bf.plan_info()
    assert info["n_groups"] == 0 and info["gather_bytes"] == 4 and info["tile"] == 1024, info
    bf.close()
    _bp_check(oracle_lib, f, tau, wp, ws, label)


@pytest.mark.parametrize("N", [700, 5000, 40_000])
@pytest.mark.parametrize("first_computed", [0, 1])
def test_bp_direct_kernel_on_ordinary_grids(oracle_lib, N, first_computed, hip_opts):
    """Option bp.direct sends a grid that HAS a plan through the direct kernel: source ranges per tile
    and the merge, ties, the arg-max start convention and its compat switch, ids with an offset."""
    from seismic_bpmf_amd import BeamformerGPU
    hip_opts("bp.direct", 1)
    hip_opts("bp.compat_first_computed", first_computed)
    rng = np.random.default_rng(N)
    f, tau, wp, ws = _bp_inputs(rng, 300, 12, 2, N, 180, 0.5)
    f[:, :, N // 3: N // 3 + 50] = 0.0               # equal (zero) beams: ties
    f -= 0.3                                          # beams of both signs
    tau[:, ::3] -= 90
    with oracle_lib.compat(oracle_lib.COMPAT_FIRST_COMPUTED if first_computed else 0):
        for oob in ("strict", "flexible"):
            ob, oa = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
            bf = BeamformerGPU(tau, ws, source_id_offset=1000)
            mb, ma = (x.cpu().numpy() for x in bf.run(f, wp, reduce="max", out_of_bounds=oob))
            bf.close()
            _same(mb, ob, f"direct {oob} maxbeam")
            assert np.array_equal(ma, oa + 1000), f"direct {oob}: {(ma != oa + 1000).sum()} arg-max differ"


# ------------------------------------------------------------ interior-tile fast path ---
@pytest.mark.gpu
@pytest.mark.parametrize("n_used,uniform", [(1, True), (3, True), (4, False), (7, True), (10, False),
                                            (13, True), (16, True), (16, False)])
@pytest.mark.parametrize("oob", ["strict", "flexible"])
def test_bp_fast_path_station_counts_and_weight_kinds(oracle_lib, n_used, uniform, oob, hip_opts):
    """bp_beam_fast_kernel (csrc/bp_fast.hip): every station-count class 4..16 (1-3 stations are
    padded to 4 with zero-slab terms), uniform weights (ready-made address records) and per-station
    weights (packed records), mixed counts inside a group (several runs), a source without any
    station, negative moveouts (edge tiles at the START of the trace as well), exact ties; against
    the oracle and against the general kernel (option bp.fast = 0), bit for bit."""
    from seismic_bpmf_amd import BeamformerGPU
    rng = np.random.default_rng(100 * n_used + int(uniform))
    K, S, C, P, N = 700, 18, 3, 2, 9000
    f = np.round(np.abs(rng.standard_normal((S, C, N))) * 4).astype(np.float32) / 4   # ties
    tau = rng.integers(-150, 600, (K, S, P)).astype(np.int32)
    tau[100:140] = tau[300:340]                                   # identical sources -> lowest id must win
    wp = rng.random((S, C, P)).astype(np.float32)
    ws = np.zeros((K, S), np.float32)
    for k in range(K):
        n = n_used if k % 5 else max(1, n_used - (k // 5) % 3)     # mixed counts -> several runs per group
        sel = rng.choice(S, n, replace=False)
        ws[k, sel] = 0.25 if uniform else rng.uniform(0.1, 1.0, n).astype(np.float32)
    ws[100:140] = ws[300:340]
    ws[77] = 0.0
    ob, oa = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
    got = {}
    for fast in ("1", "0"):
        hip_opts("bp.fast", int(fast))
        bf = BeamformerGPU(tau, ws)
        b, a = bf.run(f, wp, "max", oob)
        got[fast] = (b.cpu().numpy(), a.cpu().numpy())
        bf.close()
    for fast in ("1", "0"):
        assert np.array_equal(got[fast][0], ob) and np.array_equal(got[fast][1], oa), (fast, n_used, uniform, oob)


@pytest.mark.gpu
@pytest.mark.parametrize("n_used,uniform", [(17, True), (18, False), (20, True), (20, False), (23, True), (24, True),
                                            (27, False), (32, True), (33, True), (40, True), (40, False), (47, True),
                                            (64, True)])
def test_bp_fast_path_dense_station_weights(oracle_lib, n_used, uniform, hip_opts):
    """More than 16 weighted stations per source (round 3): the sources run the 8-byte-gather kernel
    on tiles of 256 / 128 samples as 2-5 records ("parts") whose accumulators are carried.  Every
    padded total the kernels serve, uniform and per-station weights, mixed counts inside a group,
    negative moveouts (edge tiles through the general kernel), exact ties, strict and flexible --
    against the oracle and against the general kernels alone (bp.fast = 0), bit for bit."""
    from seismic_bpmf_amd import BeamformerGPU
    rng = np.random.default_rng(1000 * n_used + int(uniform))
    K, S, C, P, N = 300, max(24, n_used + 4), 3, 2, 7000
    f = np.round(np.abs(rng.standard_normal((S, C, N))) * 4).astype(np.float32) / 4   # ties
    tau = rng.integers(-60, 140, (K, S, P)).astype(np.int32)
    tau[100:120] = tau[200:220]                                   # identical sources -> lowest id must win
    wp = rng.random((S, C, P)).astype(np.float32)
    ws = np.zeros((K, S), np.float32)
    for k in range(K):
        n = n_used if k % 4 else max(17, n_used - (k // 4) % 5)    # mixed counts -> several runs per group
        sel = rng.choice(S, n, replace=False)
        ws[k, sel] = 0.125 if uniform else rng.uniform(0.1, 1.0, n).astype(np.float32)
    ws[100:120] = ws[200:220]
    ws[33] = 0.0
    want = {oob: oracle_lib.beamform(f, tau, wp, ws, oob, "max") for oob in ("strict", "flexible")}
    for fast in (1, 0):
        hip_opts("bp.fast", fast)
        bf = BeamformerGPU(tau, ws)
        info = bf.plan_info()
        if fast:
            assert info["n_classes"] >= 1 and info["gather_bytes"] == 8 and info["tile"] in (256, 128), info
        else:
            assert info["n_classes"] == 0, info
        for oob in ("strict", "flexible"):
            b, a = bf.run(f, wp, "max", oob)
            assert np.array_equal(b.cpu().numpy(), want[oob][0]), (fast, n_used, uniform, oob)
            assert np.array_equal(a.cpu().numpy(), want[oob][1]), (fast, n_used, uniform, oob)
        bf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_used,uniform", [(33, True), (36, False), (40, True), (40, False), (41, True), (47, False),
                                            (52, True), (60, False), (61, True), (64, False)])
def test_bp_two_residency_groups_for_33_to_40_stations(oracle_lib, n_used, uniform, hip_opts):
    """Sources with 33-64 weighted stations at tile 256: every group of <= 96 sources is computed in two
    (<= 40 stations), three (<= 60) or four LDS residencies (<= 20 stations of every source each), the
    partial beams of a wave's 6 sources carried in registers between them.  Forced (bp.fast_tile = 256)
    and switched off (bp.halves = 0: tile 128), smooth moveouts (several full groups of 96) and ragged
    ones, mixed station counts, negative moveouts, ties -- the same bits as the oracle."""
    from seismic_bpmf_amd import BeamformerGPU
    rng = np.random.default_rng(77 * n_used + int(uniform))
    K, S, C, P, N = 700, max(44, min(64, n_used + 4)), 3, 2, 5000     # (a group stages at most 256 windows: 64 stations)
    f = np.round(np.abs(rng.standard_normal((S, C, N))) * 4).astype(np.float32) / 4
    base = rng.integers(-30, 90, (1, S, P))
    tau = (base + rng.integers(-6, 7, (K, S, P))).astype(np.int32)          # smooth: groups fill up to 128 sources
    tau[500:] = rng.integers(-50, 150, (200, S, P)).astype(np.int32)          # ragged: small groups
    tau[100:110] = tau[300:310]
    wp = rng.random((S, C, P)).astype(np.float32)
    ws = np.zeros((K, S), np.float32)
    for k in range(K):
        n = n_used if k % 3 else max(33, n_used - (k // 3) % 5)
        sel = rng.choice(S, n, replace=False)
        ws[k, sel] = 0.125 if uniform else rng.uniform(0.1, 1.0, n).astype(np.float32)
    ws[100:110] = ws[300:310]
    want = {oob: oracle_lib.beamform(f, tau, wp, ws, oob, "max") for oob in ("strict", "flexible")}
    for halves, tile in ((1, 256), (0, 0), (1, 0)):
        hip_opts("bp.halves", halves)
        hip_opts("bp.fast_tile", tile)
        bf = BeamformerGPU(tau, ws)
        info = bf.plan_info()
        assert info["n_classes"] == 1 and info["class_tile"][0] == (256 if tile else info["class_tile"][0]), info
        if not halves:
            assert info["class_tile"][0] == 128, info
        for oob in ("strict", "flexible"):
            b, a = bf.run(f, wp, "max", oob)
            assert np.array_equal(b.cpu().numpy(), want[oob][0]), (halves, tile, n_used, uniform, oob, info)
            assert np.array_equal(a.cpu().numpy(), want[oob][1]), (halves, tile, n_used, uniform, oob, info)
        bf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("split", [-1, 2, 5])
@pytest.mark.parametrize("n_used", [40, 52])
def test_bp_multi_residency_class_on_a_short_series_is_cut_between_groups(oracle_lib, split, n_used, hip_opts):
    """A short series (few tiles) deals the groups of a class to several workgroups per tile; a
    multi-residency class must be cut between GROUPS of sources (2-3 consecutive plan entries each),
    never inside one."""
    from seismic_bpmf_amd import BeamformerGPU
    hip_opts("bp.split", split)
    rng = np.random.default_rng(900 + n_used + split)
    K, S, C, P, N = 1100, 56, 2, 2, 3300
    f = np.round(np.abs(rng.standard_normal((S, C, N))) * 4).astype(np.float32) / 4
    tau = (rng.integers(0, 60, (1, S, P)) + rng.integers(-4, 5, (K, S, P))).astype(np.int32)
    wp = rng.random((S, C, P)).astype(np.float32)
    ws = np.zeros((K, S), np.float32)
    for k in range(K):
        ws[k, rng.choice(S, n_used - (k % 3), replace=False)] = 0.25
    bf = BeamformerGPU(tau, ws)
    info = bf.plan_info()
    assert info["class_tile"][0] == 256 and info["class_groups"][0] >= 4, info
    for oob in ("strict", "flexible"):
        b, a = bf.run(f, wp, "max", oob)
        ob, oa = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
        assert np.array_equal(b.cpu().numpy(), ob), (split, n_used, oob, info)
        assert np.array_equal(a.cpu().numpy(), oa), (split, n_used, oob, info)
    bf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [512, 256, 128])
@pytest.mark.parametrize("uniform", [True, False])
def test_bp_fast_path_every_part_size_on_every_tile(oracle_lib, tile, uniform, hip_opts):
    """bp.fast_tile forces the tile of every class: station counts 1..16 then run the 256- and
    128-sample kernels too (part sizes 6..16 resp. 8 / 12 / 16, one part), so that every
    instantiated (tile, part size) pair is compared with the oracle."""
    from seismic_bpmf_amd import BeamformerGPU
    hip_opts("bp.fast_tile", tile)
    rng = np.random.default_rng(tile + int(uniform))
    K, S, C, P, N = 480, 18, 3, 2, 6000
    f = np.round(np.abs(rng.standard_normal((S, C, N))) * 4).astype(np.float32) / 4
    tau = rng.integers(-40, 160, (K, S, P)).astype(np.int32)
    wp = rng.random((S, C, P)).astype(np.float32)
    ws = np.zeros((K, S), np.float32)
    for k in range(K):
        n = 1 + k % 16
        ws[k, rng.choice(S, n, replace=False)] = 0.5 if uniform else rng.uniform(0.1, 1.0, n).astype(np.float32)
    ob, oa = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
    bf = BeamformerGPU(tau, ws)
    info = bf.plan_info()
    assert info["n_classes"] == 1 and info["class_tile"][0] == tile, info
    b, a = bf.run(f, wp, "max", "strict")
    bf.close()
    assert np.array_equal(b.cpu().numpy(), ob) and np.array_equal(a.cpu().numpy(), oa), (tile, uniform)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [2600, 9000])
def test_bp_fast_path_mixed_station_count_classes(oracle_lib, N, hip_opts):
    """One grid, three classes: most sources with <= 16 weighted stations (tile 512), a handful with
    17-32 (tile 256) and with 33-48 (tile 128), one source without any station.  Each class kernel
    writes its own partial rows; the merge keeps the larger beam and the lowest id -- checked with
    sources of different classes that produce exactly equal beams.  Also with forced group ranges."""
    from seismic_bpmf_amd import BeamformerGPU
    rng = np.random.default_rng(N)
    K, S, C, P = 500, 50, 3, 2
    f = np.round(np.abs(rng.standard_normal((S, C, N))) * 2).astype(np.float32) / 2
    tau = rng.integers(-30, 120, (K, S, P)).astype(np.int32)
    wp = rng.random((S, C, P)).astype(np.float32)
    ws = np.zeros((K, S), np.float32)
    for k in range(K):
        n = 10 if k % 50 else (17 if k % 100 else 40)
        ws[k, rng.choice(S, n, replace=False)] = 1.0
    ws[7] = 0.0
    # three sources of three different classes with exactly equal beams: stations 12.. carry no signal
    # (their terms add an exact 0), stations 0..4 a strong one; the lowest id must win the merge
    f[12:] = 0.0
    f[:5] *= 3.0
    for k, n_silent in ((100, 5), (250, 35), (300, 20)):
        ws[k] = 0.0
        ws[k, :5] = 1.0
        ws[k, 12:12 + n_silent] = 1.0
        tau[k] = tau[100]
    want = {oob: oracle_lib.beamform(f, tau, wp, ws, oob, "max") for oob in ("strict", "flexible")}
    bf = BeamformerGPU(tau, ws)
    info = bf.plan_info()
    assert info["n_classes"] == 3 and info["class_tile"][2] == 128 and info["class_sources"] == [490, 5, 4], info
    try:
        for split in (-1, 1, 3):
            hip_opts("bp.split", split)
            for oob in ("strict", "flexible"):
                b, a = bf.run(f, wp, "max", oob)
                assert np.array_equal(b.cpu().numpy(), want[oob][0]), (N, split, oob)
                assert np.array_equal(a.cpu().numpy(), want[oob][1]), (N, split, oob)
        full = oracle_lib.beamform(f, tau, wp, ws, "strict", "none")
        assert np.array_equal(bf.run(f, wp, "none", "strict").cpu().numpy(), full)
    finally:
        bf.close()


@pytest.mark.gpu
def test_bp_fast_path_short_traces_have_no_interior_tiles(oracle_lib):
    """N smaller than tile + largest moveout: every tile is an edge tile, the interior kernel gets
    an empty range; N of a few tiles: both kernels write disjoint parts of the same arrays."""
    from seismic_bpmf_amd import beamform
    rng = np.random.default_rng(5)
    for N in (300, 700, 1100, 2049, 4096):
        K, S = 120, 9
        f = np.abs(rng.standard_normal((S, 3, N))).astype(np.float32)
        tau = rng.integers(0, 260, (K, S, 2)).astype(np.int32)
        wp = rng.random((S, 3, 2)).astype(np.float32)
        ws = (rng.random((K, S)) < 0.5).astype(np.float32)
        ob, oa = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
        mb, ma = beamform(f, tau, wp, ws, device="gpu", reduce="max", out_of_bounds="strict")
        assert np.array_equal(mb, ob) and np.array_equal(ma, oa), N


@pytest.mark.parametrize("N", [300, 1500, 3000, 6000, 20_000])
def test_bp_short_series_split_into_group_ranges(oracle_lib, N, hip_opts):
    """A series of a few tiles (the reference's event relocation beamforms 1 500-3 000 samples over the
    whole grid, BPMF/dataset.py:2174-2216) is computed by several workgroups per tile, each walking a
    range of the plan's source groups; reduce="max" folds partial maxima (value, then lowest id),
    reduce="none" needs no merge.  Automatic and forced split counts, both kernels (interior / edge
    tiles), strict and flexible, ties between sources of different ranges: all bit-exact."""
    from seismic_bpmf_amd import BeamformerGPU, synthetic as syn
    geo = syn.make_bp_geometry((30, 30, 10), 20, 2, 50.0, n_closest=10)
    tau, ws = geo["moveouts"], geo["weights_sources"]
    rng = np.random.default_rng(N)
    f = np.round(np.abs(rng.standard_normal((20, 3, N))) * 2).astype(np.float32)     # exact ties
    wp = syn.phase_weights(20, 3, 2)
    want = {oob: oracle_lib.beamform(f, tau, wp, ws, oob, "max") for oob in ("strict", "flexible")}
    bf = BeamformerGPU(tau, ws)
    try:
        assert bf.plan_info()["n_groups"] >= 4
        for split in (None, "1", "2", "7", "1000"):
            hip_opts("bp.split", -1 if split is None else int(split))
            for oob in ("strict", "flexible"):
                mb, ma = bf.run(f, wp, "max", oob)
                assert np.array_equal(mb.cpu().numpy(), want[oob][0]), (N, split, oob)
                assert np.array_equal(ma.cpu().numpy(), want[oob][1]), (N, split, oob)
        if N <= 3000:
            full = oracle_lib.beamform(f, tau, wp, ws, "strict", "none")
            for split in (None, "5"):
                hip_opts("bp.split", -1 if split is None else int(split))
                assert np.array_equal(bf.run(f, wp, "none", "strict").cpu().numpy(), full), (N, split)
    finally:
        bf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1500, 3000, 3072, 5000, 12_345])
@pytest.mark.parametrize("tau_hi", [-9, -86, -1100])
def test_bp_all_used_moveouts_negative(oracle_lib, N, tau_hi):
    """Every USED moveout negative (tmax_all < -8): the sources stay inside the trace up to its very last
    sample, so the interior range of the fast kernel would end past N -- it must end at the last whole
    tile, and the last partial tile belongs to the edge launch.  Rounds 2-3 left the samples of that tile
    unwritten (found by the 150 000-case fuzz session of round 4)."""
    from seismic_bpmf_amd import BeamformerGPU, beamform
    rng = np.random.default_rng(N - tau_hi)
    K, S, C, P = 40, 6, 2, 2
    f = np.abs(rng.standard_normal((S, C, N))).astype(np.float32)
    tau = rng.integers(tau_hi - 200, tau_hi + 1, (K, S, P)).astype(np.int32)
    tau[:, 0] = 50                                   # an UNUSED station with positive moveouts
    wp = rng.random((S, C, P)).astype(np.float32)
    ws = rng.random((K, S)).astype(np.float32)
    ws[:, 0] = 0.0
    for oob in ("strict", "flexible"):
        ob, oa = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
        assert ob[-1] > 0                             # the last sample does hold a beam
        bf = BeamformerGPU(tau, ws)
        junk = bf.torch.full((N,), -7.0, device="cuda")          # the output buffer starts out as junk
        junk_a = bf.torch.full((N,), -7, device="cuda", dtype=bf.torch.int32)
        b, a = bf.run(f, wp, "max", oob, out=(junk, junk_a))
        assert np.array_equal(b.cpu().numpy(), ob) and np.array_equal(a.cpu().numpy(), oa), (N, tau_hi, oob)
        bf.close()
        hb, ha = beamform(f, tau, wp, ws, device="gpu", reduce="max", out_of_bounds=oob)
        assert np.array_equal(hb, ob) and np.array_equal(ha, oa), (N, tau_hi, oob)


@pytest.mark.parametrize("L", [8, 64, 128, 200, 257])
@pytest.mark.parametrize("S,C", [(8, 3), (1, 1), (5, 2), (16, 2)])
def test_mf_channel_split_variant(oracle_lib, L, S, C, hip_opts):
    """Option mf.channel_split (round 5, tiny problems): the four waves of a workgroup take the same 256 lags and
    every fourth used channel each, the weighted channel sum runs behind one barrier in channel order -- the same
    fmaf chain as the plain kernel and the oracle, bit for bit.  Signed moveouts (partly valid tiles), zero-weight
    channels, a template without any, fewer used channels than waves, zero-energy windows, mf.compat_sqrt_norm."""
    from seismic_bpmf_amd import MatchedFilterGPU, _lib, matched_filter
    rng = np.random.default_rng(1000 * L + 10 * S + C)
    T, N = int(rng.integers(1, 6)), int(rng.integers(L + 300, 9000))
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    d[0, 0, 1000:1000 + 2 * L] = 0.0
    mv = rng.integers(-150, 400, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    w[rng.random((T, S, C)) < 0.3] = 0.0
    if T > 2:
        w[1] = 0.0                       # no used channel at all
        w[2] = 0.0
        w[2, 0, 0] = 0.7                 # one used channel: three of the four waves have nothing to do
    want = oracle_lib.matched_filter(tp, mv, w, d, 1)
    hip_opts("mf.channel_split", 1 << 20)
    hip_opts("mf.tiles_per_wave", 1)
    got = matched_filter(tp, mv, w, d, 1, arch="gpu", device=0, check_zeros=False)
    assert np.array_equal(got, want)
    eng = MatchedFilterGPU(device=0)
    eng.set_data(d)
    assert np.array_equal(eng.run(tp, mv, w, 1).cpu().numpy(), want)
    hip_opts("mf.channel_split", 0)
    assert np.array_equal(eng.run(tp, mv, w, 1).cpu().numpy(), want)
    hip_opts("mf.channel_split", 1 << 20)
    hip_opts("mf.compat_sqrt_norm", 1)
    with oracle_lib.compat(oracle_lib.COMPAT_SQRT_NORM):
        want2 = oracle_lib.matched_filter(tp, mv, w, d, 1)
    assert np.array_equal(matched_filter(tp, mv, w, d, 1, arch="gpu", device=0, check_zeros=False), want2)
