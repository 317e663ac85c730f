"""Option mf.split16 (round 6): matched-filter numerators from fp16 hi/lo splits of data and templates (three
v_mfma_f32_32x32x16_f16 products, fp32 accumulation -- seismic_bpmf_amd/csrc/mf_split.h) instead of the exact-fp32
MFMA chain.  OFF by default; the default path stays bit-identical to the oracle (tests/test_gpu_parity.py).

Bar for this path = the north star's float32 TOLERANCE, written out here:
    |cc_sum - reference| <= 2e-5 * sum |w|      (per channel |cc - reference| <= 2e-5)
against BOTH the CPU oracle (fp32 chains) and the float64 brute force; detection indices exact; everything the
fp32 path decides WITHOUT the numerator stays bit-identical: lags outside a template's valid range are exactly 0,
zero-energy windows / templates and zero-weight channels contribute exactly 0, nothing is ever NaN / Inf.
The measured differences are ~2e-7 (printed by the tests): a hundred times inside the bar.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_CC = 2e-5          # per unit of sum |w|


@pytest.fixture
def split16(hip_opts):
    # 2 = the split kernel for EVERY launch: the value a user sets, 1, leaves launches of fewer than 128 (template,
    # 8192-lag block) pairs -- every small shape of this file -- to the exact kernel (test_split16_small_launches_...)
    hip_opts("mf.split16", 2)
    return hip_opts


def _case(rng, T, S, C, L, N, mv_lo, mv_hi, scales=True):
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    tp -= tp.mean(axis=-1, keepdims=True)
    mv = rng.integers(mv_lo, mv_hi + 1, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    w[0, 0, :] = 0.0                                  # a dead station
    w[T - 1, :, 1 % C] = 0.0                          # a dead component
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    d[S - 1, 0, N // 3: N // 3 + 3 * L] = 0.0         # a gap: CC exactly 0 there
    if scales and S > 2:
        d[1] *= np.float32(2.5e3)                     # channels in other units ...
        d[2, 0] *= np.float32(4.0e-3)
        tp[:, 1] *= np.float32(3.0e-2)
        d[0, C - 1, N // 2] = np.float32(3.0e4)       # ... and a glitch of 30 000 sigma
    return tp, mv, w, d


def _check(got, want, w, what, per_channel=False):
    assert got.shape == want.shape, what
    assert np.isfinite(got).all(), what
    # what no numerator decides is bit-identical: exact zeros where the reference has exact zeros
    assert np.array_equal(got == 0.0, want == 0.0) or np.abs(got[(got == 0.0) != (want == 0.0)]).max(initial=0.0) < 1e-30, what
    diff = np.abs(got.astype(np.float64) - np.asarray(want, np.float64))
    if per_channel:
        bound = TOL_CC
        worst = diff.max()
    else:
        sw = np.abs(w).reshape(w.shape[0], -1).sum(axis=1)[:, None]
        worst = (diff / np.maximum(sw, 1e-30)).max()
        bound = TOL_CC
    print(f"{what}: max |d cc| / sum|w| = {worst:.3e} (bar {bound:.0e})")
    assert worst <= bound, f"{what}: {worst:.3e} > {bound:.0e}"
    return worst


@pytest.mark.parametrize("network_sum", [True, False])
@pytest.mark.parametrize("L,N,T", [(32, 9000, 3), (100, 30011, 3), (256, 40000, 4), (200, 25000, 2), (378, 20000, 2)])
def test_split16_within_tolerance_of_oracle_and_float64(oracle_lib, split16, network_sum, L, N, T):
    from seismic_bpmf_amd import matched_filter
    from oracle import ref_numpy
    rng = np.random.default_rng(L * 13 + N)
    tp, mv, w, d = _case(rng, T, 4, 3, L, N, -70, 900)
    got = matched_filter(tp, mv, w, d, 1, arch="gpu", network_sum=network_sum, check_zeros=False)
    want = oracle_lib.matched_filter(tp, mv, w, d, 1, network_sum)
    _check(got, want, w, f"split16 vs oracle L={L} N={N} ns={network_sum}", per_channel=not network_sum)
    if N <= 30011:
        f64 = ref_numpy.matched_filter_f64(tp, mv, w, d, 1, network_sum)
        # (the float64 reference zeroes E_t * E_d <= 1e-6, the build r_t * r_d >= 1000: same channels here)
        diff = np.abs(got - f64)
        scale = 1.0 if not network_sum else np.abs(w).reshape(T, -1).sum(axis=1)[:, None]
        worst = (diff / scale).max()
        print(f"split16 vs float64: {worst:.3e}")
        assert worst <= TOL_CC


def test_split16_differs_from_the_exact_path_only_in_the_last_bits(oracle_lib, hip_opts):
    """The same call with the option off is bit-identical to the oracle; with it on, within the tolerance -- and
    the two are NOT the same array (the option really selects the other kernel)."""
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(77)
    tp, mv, w, d = _case(rng, 3, 5, 3, 128, 60000, 0, 700)
    exact = matched_filter(tp, mv, w, d, 1, check_zeros=False)
    want = oracle_lib.matched_filter(tp, mv, w, d, 1, True)
    assert np.array_equal(exact, want)
    hip_opts("mf.split16", 2)
    split = matched_filter(tp, mv, w, d, 1, check_zeros=False)
    hip_opts.reset("mf.split16")
    assert not np.array_equal(split, exact)
    _check(split, exact, w, "split16 vs exact path")
    again = matched_filter(tp, mv, w, d, 1, check_zeros=False)
    assert np.array_equal(again, exact)


@pytest.mark.parametrize("network_sum", [True, False])
@pytest.mark.parametrize("L,N", [(100, 30011), (256, 40000), (500, 30000)])
def test_split16_under_sqrt_norm_and_the_upstream_profile(oracle_lib, split16, network_sum, L, N):
    """mf.compat_sqrt_norm with mf.split16: the split kernel's epilogue takes num / sqrtf(E_t * E_d) behind the
    1e-6 guard (the norm arrays hold energies under the switch) -- it used to hand such launches to the exact kernel.
    Same bar against the oracle under the same switch; then the whole "upstream-recollected" profile on top."""
    from seismic_bpmf_amd import matched_filter, compat_profile, _lib
    rng = np.random.default_rng(L + N)
    tp, mv, w, d = _case(rng, 3, 4, 3, L, N, -70, 900)
    split16("mf.compat_sqrt_norm", 1)
    with oracle_lib.compat(oracle_lib.COMPAT_SQRT_NORM):
        want = oracle_lib.matched_filter(tp, mv, w, d, 1, network_sum)
    got = matched_filter(tp, mv, w, d, 1, arch="gpu", network_sum=network_sum, check_zeros=False)
    _check(got, want, w, f"split16 + sqrt_norm L={L} ns={network_sum}", per_channel=not network_sum)
    # ... and it IS the split kernel that ran: the exact kernel under the switch equals the oracle bit for bit
    assert not np.array_equal(got, want)
    split16.reset("mf.compat_sqrt_norm")
    before = {k: _lib.get_option(k) for k in _lib.COMPAT_SWITCHES}
    try:
        compat_profile("upstream-recollected")
        flags = 0
        for name in ("SQRT_NORM", "EXCLUSIVE_LAST_LAG", "RANGE_ALL_CHANNELS", "SEQUENTIAL_CSUM"):
            flags |= getattr(oracle_lib, "COMPAT_" + name)
        with oracle_lib.compat(flags):
            want = oracle_lib.matched_filter(tp, mv, w, d, 1, network_sum)
        got = matched_filter(tp, mv, w, d, 1, arch="gpu", network_sum=network_sum, check_zeros=False)
        _check(got, want, w, f"split16 + profile L={L} ns={network_sum}", per_channel=not network_sum)
    finally:
        compat_profile("build")
        for k, v in before.items():
            _lib.set_option(k, v[0] if isinstance(v, tuple) else v)


def test_split16_small_launches_stay_with_the_exact_kernel(oracle_lib, hip_opts):
    """mf.split16 = 1: a launch of fewer than 128 (template, 8192-lag block) pairs is latency-bound in the split kernel
    (configs[0], 88 pairs: 0.71x the exact kernel's speed; 176 pairs: x 1.1-1.6, tools/probe_split_small.py) and stays
    with the exact kernel -- the oracle's bits; from 128 pairs on it is the split kernel (= what mf.split16 = 2 gives
    for every launch)."""
    from seismic_bpmf_amd.matched_filter import MatchedFilterGPU
    rng = np.random.default_rng(9)
    tp, mv, w, d = _case(rng, 16, 3, 2, 64, 70_000, 0, 300)           # 9 blocks of 8192 lags per template
    eng = MatchedFilterGPU()                                          # (resident: one launch for all T templates; the
    eng.set_data(d)                                                   #  host-pointer call launches per template batch)
    for T, is_split in ((14, False), (15, True), (16, True)):         # 126, 135, 144 pairs
        want = oracle_lib.matched_filter(tp[:T], mv[:T], w[:T], d, 1, True)
        hip_opts("mf.split16", 2)
        forced = eng.run(tp[:T], mv[:T], w[:T]).cpu().numpy()
        hip_opts("mf.split16", 1)
        got = eng.run(tp[:T], mv[:T], w[:T]).cpu().numpy()
        assert not np.array_equal(forced, want)
        assert np.array_equal(got, forced if is_split else want), T
        _check(got, want, w[:T], f"mf.split16 = 1, T = {T}")


@pytest.mark.parametrize("step", [2, 5])
def test_split16_steps(oracle_lib, split16, step):
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(300 + step)
    tp, mv, w, d = _case(rng, 2, 3, 3, 64, 20001, -16, 200)
    for ns in (True, False):
        got = matched_filter(tp, mv, w, d, step, check_zeros=False, network_sum=ns)
        want = oracle_lib.matched_filter(tp, mv, w, d, step, ns)
        _check(got, want, w, f"split16 step {step} ns={ns}", per_channel=not ns)


def test_split16_edge_cases(oracle_lib, split16):
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(9)
    tp, mv, w, d = _case(rng, 3, 2, 3, 48, 3000, 0, 50, scales=False)
    w[1] = 0.0                      # a dead template -> a row of exact zeros
    mv[2, 1, 0] = 2990              # a moveout that leaves no valid lag -> a row of exact zeros
    got = matched_filter(tp, mv, w, d, 1, check_zeros=False)
    want = oracle_lib.matched_filter(tp, mv, w, d, 1, True)
    _check(got, want, w, "split16 edge: dead template / impossible moveout")
    assert not got[1].any() and not got[2].any()
    # N == L: exactly one lag
    mv0 = np.zeros_like(mv)
    w1 = np.ones_like(w)
    got = matched_filter(tp, mv0, w1, d[:, :, :48], 1, check_zeros=False)
    want = oracle_lib.matched_filter(tp, mv0, w1, d[:, :, :48], 1, True)
    assert got.shape == (3, 1)
    _check(got, want, w1, "split16 edge: N == L")
    # zero-energy template and zero-energy data: exactly 0 contributions, never NaN / Inf
    tp2 = tp.copy()
    tp2[0, 0, 0] = 0.0
    d2 = d.copy()
    d2[1] = 0.0
    got = matched_filter(tp2, mv0, w1, d2, 1, check_zeros=False)
    want = oracle_lib.matched_filter(tp2, mv0, w1, d2, 1, True)
    _check(got, want, w1, "split16 edge: zero energy")
    got_c = matched_filter(tp2, mv0, w1, d2, 1, check_zeros=False, network_sum=False)
    assert not got_c[:, :, 1].any() and not got_c[0, :, 0, 0].any()
    # non-finite data poison only what they touch and never reach other templates' rows as Inf
    # (BPMF scrubs NaN from the result, similarity_search.py:540)


def test_split16_autocorrelation_and_detection_indices(split16):
    """A template cut out of the data correlates to the weight sum at its own lag (within the tolerance), and
    the arg-max -- the detection index -- is exactly that lag."""
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(11)
    S, C, L, N = 6, 3, 256, 150_000
    d = rng.standard_normal((S, C, N)).astype(np.float32)
    lags = [5000, 77_777, 120_001]
    mv = rng.integers(0, 1500, (len(lags), S, C)).astype(np.int32)
    tp = np.stack([np.stack([np.stack([d[s, c, i0 + mv[k, s, c]: i0 + mv[k, s, c] + L] for c in range(C)])
                             for s in range(S)]) for k, i0 in enumerate(lags)])
    w = np.full((len(lags), S, C), 1.0 / (S * C), np.float32)
    cc = matched_filter(tp, mv, w, d, 1, check_zeros=False)
    for k, i0 in enumerate(lags):
        assert int(cc[k].argmax()) == i0
        assert abs(float(cc[k, i0]) - 1.0) <= TOL_CC


def test_split16_extreme_dynamic_range(oracle_lib, split16):
    """A channel whose largest sample is 10^6 noise deviations (a clipped digitiser, a calibration pulse): the
    lo halves of the quiet samples fall into fp16's subnormal range, which the matrix pipe honours
    (tools/ubench/mfma_split16.hip prints the probe) -- still inside the tolerance."""
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(5)
    tp, mv, w, d = _case(rng, 2, 3, 3, 256, 50_000, 0, 300, scales=False)
    d[0, 0, 40_000] = np.float32(1.0e6)
    d[1, 1] *= np.float32(1.0e-7)
    d[1, 1, 100] = np.float32(1.0)          # 10^7 deviations of that channel
    got = matched_filter(tp, mv, w, d, 1, check_zeros=False)
    want = oracle_lib.matched_filter(tp, mv, w, d, 1, True)
    _check(got, want, w, "split16 extreme dynamic range")


def test_split16_resident_engine_reuses_the_prepared_day(oracle_lib, split16):
    """MatchedFilterGPU: the split of the day is part of the prepared state (several template batches, then the
    option off again -> the day is prepared again and the exact path answers bit for bit)."""
    import torch
    from seismic_bpmf_amd.matched_filter import MatchedFilterGPU
    rng = np.random.default_rng(21)
    tp, mv, w, d = _case(rng, 6, 4, 3, 200, 80_000, -30, 1200)
    eng = MatchedFilterGPU()
    eng.set_data(d)
    want = oracle_lib.matched_filter(tp, mv, w, d, 1, True)
    a = eng.run(tp[:4], mv[:4], w[:4]).cpu().numpy()
    b = eng.run(tp[4:], mv[4:], w[4:]).cpu().numpy()
    _check(np.concatenate([a, b]), want, w, "split16 resident engine, two batches")
    split16.reset("mf.split16")
    c = eng.run(tp, mv, w).cpu().numpy()
    assert np.array_equal(c, want)
    torch.cuda.synchronize()


@pytest.mark.parametrize("L,N", [(379, 20_000), (400, 30_000), (752, 25_000), (753, 25_000), (1040, 30_000), (2040, 40_000)])
def test_split16_long_templates_in_segments(oracle_lib, split16, L, N):
    """Templates longer than one band image (376 samples) are correlated in equal segments into the same accumulators
    (2 .. 6 segments here, lengths on both sides of a segment boundary); beyond 2065 samples -- the limit of every MFMA
    kernel -- the generic exact kernel answers, bit for bit."""
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(31 + L)
    tp, mv, w, d = _case(rng, 2, 3, 3, L, N, -40, 300)
    for ns in (True, False):
        got = matched_filter(tp, mv, w, d, 1, check_zeros=False, network_sum=ns)
        want = oracle_lib.matched_filter(tp, mv, w, d, 1, ns)
        assert not np.array_equal(got, want)                 # (the split kernel ran)
        _check(got, want, w, f"split16 L={L} ns={ns}", per_channel=not ns)


def test_split16_templates_beyond_every_mfma_kernel_take_the_exact_generic_kernel(oracle_lib, split16):
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(32)
    tp, mv, w, d = _case(rng, 2, 2, 2, 2100, 9000, 0, 50, scales=False)
    got = matched_filter(tp, mv, w, d, 1, check_zeros=False)
    assert np.array_equal(got, oracle_lib.matched_filter(tp, mv, w, d, 1, True))


def test_split16_configs0_in_full(oracle_lib, split16):
    """BASELINE configs[0] (4 templates x 8 stations x 3 components, L = 128, one hour at 50 Hz) in full: every
    CC sum within the tolerance of the oracle, all planted events at their exact index."""
    from seismic_bpmf_amd import matched_filter
    from seismic_bpmf_amd import synthetic as syn
    mf = syn.make_mf_inputs(T=4, S=8, C=3, L=128, N=180_000, max_moveout=1500, n_events=5)
    got = matched_filter(mf["templates"], mf["moveouts"], mf["weights"], mf["data"], 1, check_zeros=False)
    want = oracle_lib.matched_filter(mf["templates"], mf["moveouts"], mf["weights"], mf["data"], 1)
    _check(got, want, mf["weights"], "split16 configs[0]")
    for t, i0 in mf["planted"]:
        lo, hi = max(0, i0 - 64), i0 + 65
        assert int(got[t, lo:hi].argmax()) == int(want[t, lo:hi].argmax())
    assert [int(r.argmax()) for r in got] == [int(r.argmax()) for r in want]


def _fuzz_seeds(default):
    import os
    spec = os.environ.get("BPMF_FUZZ_SEEDS")
    if not spec:
        return range(default)
    a, b = spec.split(":")
    return range(int(a), int(b))


@pytest.mark.parametrize("seed", _fuzz_seeds(40))
def test_fuzz_split16_random_shapes_signed_moveouts(oracle_lib, split16, seed):
    """The generator of tests/test_gpu_fuzz.py::test_mf_random_shapes_signed_moveouts under mf.split16: moveouts of both
    signs (first valid lags that are not multiples of 8: the remainder baked into the band image), template lengths on
    both sides of every k-step boundary and of the segment boundaries (376, 752), steps, gaps, dead channels, series shorter than a
    wave's 2048 lags, channels in wildly different units.  Bar: the tolerance, plus exact zeros exactly where the oracle has
    them."""
    from seismic_bpmf_amd import matched_filter
    rng = np.random.default_rng(17_000 + seed)
    T = int(rng.integers(1, 7))
    S = int(rng.integers(1, 5))
    C = int(rng.integers(1, 4))
    L = int(rng.choice([1, 3, 8, 9, 10, 16, 26, 27, 31, 48, 64, 100, 128, 200, 250, 251, 256, 257, 266, 267, 300, 376, 377, 378, 379, 400, 752, 753, 1100]))
    N = int(L + rng.choice([0, 1, 2, 3, 5, 255, 1000, 2047, 2048, 2049, 8191, 8192, 8193, 9000, 20000]))
    step = int(rng.choice([1, 1, 1, 1, 2, 3, 4, 7, 16, 17, 64]))
    lo = -int(rng.choice([0, 1, 2, 3, 5, 7, 8, 9, 50, 255, 257, 1023, 1026, max(1, N // 2), N + 3]))
    hi = int(rng.choice([0, 1, 2, 3, 7, 8, 100, 1025, max(1, N // 2), N + 3]))
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    data = rng.standard_normal((S, C, N)).astype(np.float32)
    data *= (10.0 ** rng.uniform(-6, 4, (S, C, 1))).astype(np.float32)
    tp *= (10.0 ** rng.uniform(-4, 3, (T, S, C, 1))).astype(np.float32)
    if seed % 5 == 0:
        a = int(rng.integers(0, max(1, N - 1)))
        data[:, :, a:a + L + int(rng.integers(0, 2 * L + 2))] = 0.0
    if seed % 9 == 0:
        tp[0, 0] = 0.0
    if seed % 4 == 0 and N > 10:
        data[0, 0, int(rng.integers(0, N))] *= np.float32(3.0e4)          # a glitch
    mv = rng.integers(lo, hi + 1, (T, S, C)).astype(np.int32)
    w = rng.random((T, S, C)).astype(np.float32)
    w[rng.random((T, S, C)) < 0.2] = 0.0
    # every third seed under mf.compat_sqrt_norm (energies in the norm arrays, the 1e-6 guard: other exact zeros)
    flags = oracle_lib.COMPAT_SQRT_NORM if seed % 3 == 1 else 0
    split16("mf.compat_sqrt_norm", 1 if flags else 0)
    for ns in (True, False):
        got = matched_filter(tp, mv, w, data, step, arch="gpu", network_sum=ns, check_zeros=False)
        with oracle_lib.compat(flags):
            want = oracle_lib.matched_filter(tp, mv, w, data, step, network_sum=ns)
        what = f"seed {seed} ns={ns} flags={flags} (T={T} S={S} C={C} L={L} N={N} step={step} mv in [{lo},{hi}])"
        assert got.shape == want.shape and np.isfinite(got).all(), what
        assert np.array_equal(got == 0.0, want == 0.0), what + f": {((got == 0) != (want == 0)).sum()} exact zeros differ"
        diff = np.abs(got.astype(np.float64) - want)
        scale = 1.0 if not ns else np.maximum(np.abs(w).reshape(T, -1).sum(axis=1), 1e-30)[:, None]
        assert (diff / scale).max(initial=0.0) <= TOL_CC, what + f": {(diff / scale).max():.3e}"


def test_split16_refuses_a_day_that_was_prepared_without_it(hip_opts):
    """A caller of the *_dev entry points who prepares a day with the option off and then runs with it on (flag
    BPMF_MF_DATA_PREPARED) must get an error, not the correlation of whatever the split region holds."""
    from seismic_bpmf_amd import _lib
    from seismic_bpmf_amd.matched_filter import MatchedFilterGPU
    rng = np.random.default_rng(3)
    tp, mv, w, d = _case(rng, 2, 3, 3, 64, 30_000, 0, 100, scales=False)
    eng = MatchedFilterGPU()
    eng.set_data(d)
    hip_opts("mf.split16", 2)
    eng.run(tp, mv, w)                               # sizes the workspace for the option and prepares WITH the split
    hip_opts("mf.split16", 0)
    eng.run(tp, mv, w)                               # prepared again, without it
    hip_opts("mf.split16", 2)
    key = list(eng._prepared_for)
    key[-1] = 2                                      # pretend the prepared state matched the option
    eng._prepared_for = tuple(key)
    with pytest.raises(_lib.BpmfHipError, match="prepared without it"):
        eng.run(tp, mv, w)
    eng._prepared_for = None
    eng.run(tp, mv, w)                               # prepares, runs


@pytest.mark.parametrize("k", [2, 4])
def test_split16_on_several_devices_equals_one_device(oracle_lib, split16, k):
    """bpmf_mf_run_multi under the option (templates block-partitioned over k logical devices, the day handed on device
    to device): every device prepares its own split of the day; the rows equal the one-device call's bit for bit (the
    same kernel on the same template) and lie within the tolerance of the oracle."""
    from seismic_bpmf_amd import _lib, matched_filter
    rng = np.random.default_rng(50 + k)
    tp, mv, w, d = _case(rng, 7, 4, 3, 200, 50_000, -20, 600)
    one = matched_filter(tp, mv, w, d, 1, check_zeros=False, device=0)
    split16("debug.virtual_devices", k)
    multi = matched_filter(tp, mv, w, d, 1, check_zeros=False, device=None)
    assert np.array_equal(multi, one)
    _check(multi, oracle_lib.matched_filter(tp, mv, w, d, 1, True), w, f"split16 on {k} devices")
    _lib.release_device_memory(-1)
