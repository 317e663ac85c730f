"""GPU: the three upstream-compatibility switches (off by default).  Each replaces one
convention of this build that rests on recollection only by the alternative the upstream packages are
recollected to implement, is mirrored by a variant of the CPU oracle (oracle.compat), and is compared
with it bit for bit -- so that a user who holds the real fast_matched_filter / beampower can diff
against either convention."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mf_case(seed, step):
    from seismic_bpmf_amd import synthetic as syn
    m = syn.make_mf_inputs(T=3, S=4, C=3, L=48, N=9000, seed=seed, max_moveout=120, n_events=2, step=step)
    m["moveouts"] = m["moveouts"] - 37          # both signs
    m["weights"][1, 2] = 0.0
    m["data"][0, 0, 2000:2300] = 0.0            # zero-energy windows: the stability rule matters
    return m


@pytest.mark.parametrize("step", [1, 2, 3])
@pytest.mark.parametrize("network_sum", [True, False])
def test_mf_exclusive_last_lag(oracle_lib, hip_opts, step, network_sum):
    """mf.compat_exclusive_last_lag: the last valid data offset obeys i * step < N - L - mv_max (upstream's
    recollected loop bound) instead of <=: at most one CC per template changes, from a regular value to
    0; every kernel family."""
    from seismic_bpmf_amd import matched_filter
    m = _mf_case(3, step)
    args = (m["templates"], m["moveouts"], m["weights"], m["data"], step)
    base = oracle_lib.matched_filter(*args, network_sum)
    with oracle_lib.compat(oracle_lib.COMPAT_EXCLUSIVE_LAST_LAG):
        want = oracle_lib.matched_filter(*args, network_sum)
    changed = (base != want).reshape(3, -1).any(axis=1)
    assert changed.any() and not np.array_equal(base, want)
    hip_opts("mf.compat_exclusive_last_lag", 1)
    for wave in (1, 0):
        hip_opts("mf.wave_kernel", wave)
        got = matched_filter(*args, arch="gpu", network_sum=network_sum, check_zeros=False)
        assert np.array_equal(got, want), (step, network_sum, wave)
    hip_opts("mf.max_mfma_step", 0)               # the generic kernel
    assert np.array_equal(matched_filter(*args, arch="gpu", network_sum=network_sum, check_zeros=False), want)
    hip_opts.reset("mf.compat_exclusive_last_lag")
    assert np.array_equal(matched_filter(*args, arch="gpu", network_sum=network_sum, check_zeros=False), base)


@pytest.mark.parametrize("step", [1, 2])
@pytest.mark.parametrize("network_sum", [True, False])
def test_mf_sqrt_norm(oracle_lib, hip_opts, step, network_sum):
    """mf.compat_sqrt_norm: cc = num / sqrtf(E_t * E_d) where E_t * E_d > 1e-6, else 0 (the textbook form
    with upstream's recollected stability threshold) instead of num * (r_t * r_d): a few ulp apart from
    the default, identical zeros."""
    from seismic_bpmf_amd import matched_filter
    m = _mf_case(5, step)
    args = (m["templates"], m["moveouts"], m["weights"], m["data"], step)
    base = oracle_lib.matched_filter(*args, network_sum)
    with oracle_lib.compat(oracle_lib.COMPAT_SQRT_NORM):
        want = oracle_lib.matched_filter(*args, network_sum)
    assert not np.array_equal(base, want) and np.abs(base - want).max() < 2e-6
    assert np.array_equal(base == 0, want == 0)
    hip_opts("mf.compat_sqrt_norm", 1)
    # round 4: the MFMA kernels carry the form in their epilogue (network sums; per-channel output takes
    # the generic kernel): the independent-wave kernel at 1 / 2 / 4 tiles per wave, the workgroup kernel,
    # the generic kernel
    for wave, ntile in ((1, 0), (1, 1), (1, 2), (1, 4), (0, 0)):
        hip_opts("mf.wave_kernel", wave)
        hip_opts("mf.tiles_per_wave", ntile)
        got = matched_filter(*args, arch="gpu", network_sum=network_sum, check_zeros=False)
        assert np.array_equal(got, want), (step, network_sum, wave, ntile)
    hip_opts("mf.max_mfma_step", 0)
    assert np.array_equal(matched_filter(*args, arch="gpu", network_sum=network_sum, check_zeros=False), want)
    # both MF switches together
    hip_opts("mf.compat_exclusive_last_lag", 1)
    with oracle_lib.compat(oracle_lib.COMPAT_SQRT_NORM | oracle_lib.COMPAT_EXCLUSIVE_LAST_LAG):
        want2 = oracle_lib.matched_filter(*args, network_sum)
    assert np.array_equal(matched_filter(*args, arch="gpu", network_sum=network_sum, check_zeros=False), want2)


@pytest.mark.parametrize("n_used,S", [(6, 9), (20, 24), (40, 44)])
@pytest.mark.parametrize("oob", ["strict", "flexible"])
def test_bp_first_computed(oracle_lib, hip_opts, n_used, S, oob):
    """bp.compat_first_computed: the running maximum starts from the first computed beam, so samples
    whose beams are all <= 0 return their true (negative or zero) maximum and its lowest source instead
    of (0, 0); samples without any computed beam still return (0, 0).  Signed features make the
    difference visible; every station-count class of the interior kernel, the general kernels on the
    edges and alone, forced group ranges."""
    from seismic_bpmf_amd import BeamformerGPU
    rng = np.random.default_rng(n_used + len(oob))
    K, C, P, N = 150, 3, 2, 5000
    f = (np.round(rng.standard_normal((S, C, N)) * 4) / 4).astype(np.float32)
    f[:, :, 1500:2500] = -np.abs(f[:, :, 1500:2500])            # a stretch where every beam is <= 0
    f[:, :, 3000:3200] = 0.0                                     # and one where every beam is exactly 0
    tau = rng.integers(-40, 130, (K, S, P)).astype(np.int32)
    wp = rng.random((S, C, P)).astype(np.float32)
    ws = np.zeros((K, S), np.float32)
    for k in range(K):
        ws[k, rng.choice(S, n_used, replace=False)] = 0.5
    ws[0] = 0.0                                                  # source 0 computes nothing
    base = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
    with oracle_lib.compat(oracle_lib.COMPAT_FIRST_COMPUTED):
        want = oracle_lib.beamform(f, tau, wp, ws, oob, "max")
    assert (want[0] < 0).any() and not np.array_equal(base[1], want[1])
    hip_opts("bp.compat_first_computed", 1)
    for fast, split in ((1, -1), (0, -1), (1, 3)):
        hip_opts("bp.fast", fast)
        hip_opts("bp.split", split)
        bf = BeamformerGPU(tau, ws)
        b, a = bf.run(f, wp, "max", oob)
        bf.close()
        assert np.array_equal(b.cpu().numpy(), want[0]), (n_used, oob, fast, split)
        assert np.array_equal(a.cpu().numpy(), want[1]), (n_used, oob, fast, split)
    # the host entry point with the grid split into several device blocks: every block keeps -inf where it
    # computed nothing and the host finishes (0, first id) after the merge
    from seismic_bpmf_amd import beamform
    hip_opts.reset("bp.fast")
    hip_opts.reset("bp.split")
    for dev in (0, [0, 0, 0]):
        b, a = beamform(f, tau, wp, ws, device="gpu", out_of_bounds=oob, device_id=dev)
        assert np.array_equal(b, want[0]) and np.array_equal(a, want[1]), (n_used, oob, dev)
    hip_opts.reset("bp.compat_first_computed")
    bf = BeamformerGPU(tau, ws)
    b, a = bf.run(f, wp, "max", oob)
    bf.close()
    assert np.array_equal(b.cpu().numpy(), base[0]) and np.array_equal(a.cpu().numpy(), base[1])


def test_unknown_option_and_range_are_errors():
    from seismic_bpmf_amd import _lib
    with pytest.raises(_lib.BpmfHipError, match="unknown option"):
        _lib.set_option("mf.compat_nothing", 1)
    with pytest.raises(_lib.BpmfHipError, match="outside"):
        _lib.set_option("bp.compat_first_computed", 2)
    assert _lib.get_option("mf.compat_sqrt_norm") == (0, 0)


@pytest.mark.parametrize("L", [100, 256, 300, 1100])
def test_mf_sqrt_norm_mfma_kernels_resident(oracle_lib, hip_opts, L):
    """The resident engine under mf.compat_sqrt_norm: every MFMA kernel size (L <= 257 wave kernel, the
    workgroup kernels of L <= 273 / 1041 / 2065), the prepared energies re-made when the switch flips
    between two runs on the same data (the norm arrays hold energies with the switch on, reciprocal norms
    with it off)."""
    import torch
    from seismic_bpmf_amd import MatchedFilterGPU, synthetic as syn
    m = syn.make_mf_inputs(T=3, S=3, C=2, L=L, N=14_000, seed=L, max_moveout=150, n_events=1)
    m["data"][1, 0, 5000:5000 + L + 40] = 0.0
    args = (m["templates"], m["moveouts"], m["weights"], m["data"], 1)
    base = oracle_lib.matched_filter(*args)
    with oracle_lib.compat(oracle_lib.COMPAT_SQRT_NORM):
        want = oracle_lib.matched_filter(*args)
    mf = MatchedFilterGPU()
    mf.set_data(m["data"])
    assert np.array_equal(mf.run(m["templates"], m["moveouts"], m["weights"], 1).cpu().numpy(), base)
    hip_opts("mf.compat_sqrt_norm", 1)
    assert np.array_equal(mf.run(m["templates"], m["moveouts"], m["weights"], 1).cpu().numpy(), want)
    assert np.array_equal(mf.run(m["templates"], m["moveouts"], m["weights"], 1).cpu().numpy(), want)
    hip_opts.reset("mf.compat_sqrt_norm")
    assert np.array_equal(mf.run(m["templates"], m["moveouts"], m["weights"], 1).cpu().numpy(), base)


# ---------------------------------------------------------------------------------------------------
# Round 5: the conventions the round-4 review listed as "rest on SURVEY App. C, no switch".
@pytest.mark.parametrize("step", [1, 3])
@pytest.mark.parametrize("network_sum", [True, False])
def test_mf_range_all_channels(oracle_lib, hip_opts, step, network_sum):
    """mf.compat_range_all_channels: a template's valid lags are those where EVERY channel's window lies
    inside the data, weighted or not (the default looks at the weighted channels only): a zero-weight
    channel with an extreme moveout shortens the range, one with a moveout far outside the trace
    empties it.  Every kernel family, including the in-kernel prologue of the small-problem variants."""
    from seismic_bpmf_amd import matched_filter
    m = _mf_case(7, step)
    m["weights"][0, 3] = 0.0
    m["moveouts"][0, 3] = [-400, 900, 30]       # zero-weight channels beyond every weighted moveout
    m["weights"][2, 0, 1] = 0.0
    m["moveouts"][2, 0, 1] = 50_000             # ... and one far outside the trace: no valid lag at all
    args = (m["templates"], m["moveouts"], m["weights"], m["data"], step)
    base = oracle_lib.matched_filter(*args, network_sum)
    with oracle_lib.compat(oracle_lib.COMPAT_RANGE_ALL_CHANNELS):
        want = oracle_lib.matched_filter(*args, network_sum)
    nz = lambda a: (a.reshape(3, -1) != 0).sum(axis=1)
    assert nz(want)[0] < nz(base)[0] and nz(want)[2] == 0 and nz(base)[2] > 0 and nz(want)[1] <= nz(base)[1]
    hip_opts("mf.compat_range_all_channels", 1)
    for wave, fused in ((1, 1), (1, 0), (0, 0)):
        hip_opts("mf.wave_kernel", wave)
        hip_opts("mf.fused_prologue", fused)
        got = matched_filter(*args, arch="gpu", network_sum=network_sum, check_zeros=False)
        assert np.array_equal(got, want), (step, network_sum, wave, fused)
    hip_opts("mf.max_mfma_step", 0)               # the generic kernel
    assert np.array_equal(matched_filter(*args, arch="gpu", network_sum=network_sum, check_zeros=False), want)
    hip_opts.reset("mf.compat_range_all_channels")
    assert np.array_equal(matched_filter(*args, arch="gpu", network_sum=network_sum, check_zeros=False), base)


def test_mf_sequential_csum(oracle_lib, hip_opts):
    """mf.compat_sequential_csum: the double prefix sum of data^2 is ONE sequential chain per channel
    instead of the 1024-sample hierarchy.  The two differ in the last bits of a double, i.e. in one ulp of
    the float32 window energy at a few windows in a million (the longer the trace and the wider its
    amplitude range, the more: a quiet window behind a loud stretch is the difference of two large prefix
    sums): the case -- 400 000 samples whose amplitude swings by e^2 -- makes the switch change some outputs; with it on, the
    library equals the oracle's sequential variant bit for bit, alone and together with
    mf.compat_sqrt_norm, through the resident engine as well."""
    import torch
    from seismic_bpmf_amd import MatchedFilterGPU, matched_filter
    rng = np.random.default_rng(11)
    T, S, C, L, N = 2, 2, 2, 40, 400_000
    d = (rng.standard_normal((S, C, N)) * np.exp(rng.standard_normal((S, C, 1)) +
                                                 2 * np.sin(np.arange(N) / 7000.0))).astype(np.float32)
    tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
    mv = rng.integers(0, 300, (T, S, C)).astype(np.int32)
    w = np.full((T, S, C), 0.25, np.float32)
    base = oracle_lib.matched_filter(tp, mv, w, d, 1)
    with oracle_lib.compat(oracle_lib.COMPAT_SEQUENTIAL_CSUM):
        want = oracle_lib.matched_filter(tp, mv, w, d, 1)
    n_diff = int((base != want).sum())
    assert 0 < n_diff < base.size // 100, n_diff          # a handful of ulp-level differences
    assert np.abs(base - want).max() < 1e-6
    hip_opts("mf.compat_sequential_csum", 1)
    assert np.array_equal(matched_filter(tp, mv, w, d, 1, arch="gpu", check_zeros=False), want)
    eng = MatchedFilterGPU(device=0)
    eng.set_data(d)
    assert np.array_equal(eng.run(tp, mv, w, 1).cpu().numpy(), want)
    hip_opts.reset("mf.compat_sequential_csum")            # the engine notices: its prepared state is keyed by it
    assert np.array_equal(eng.run(tp, mv, w, 1).cpu().numpy(), base)
    hip_opts("mf.compat_sequential_csum", 1)
    hip_opts("mf.compat_sqrt_norm", 1)
    with oracle_lib.compat(oracle_lib.COMPAT_SEQUENTIAL_CSUM | oracle_lib.COMPAT_SQRT_NORM):
        want2 = oracle_lib.matched_filter(tp, mv, w, d, 1)
    assert np.array_equal(eng.run(tp, mv, w, 1).cpu().numpy(), want2)
    torch.cuda.synchronize()


def _bp_case(rng, K=150, S=7, N=9_000, lo=-120, hi=260):
    f = np.abs(rng.standard_normal((S, 3, N))).astype(np.float32)
    tau = rng.integers(lo, hi, (K, S, 2)).astype(np.int32)
    wp = rng.random((S, 3, 2)).astype(np.float32)
    ws = rng.random((K, S)).astype(np.float32)
    ws[rng.random((K, S)) < 0.5] = 0.0
    ws[5] = 0.0                                           # a source without a weighted station
    return f, tau, wp, ws


@pytest.mark.parametrize("reduce", ["max", "none"])
def test_bp_strict_upper_only(oracle_lib, hip_opts, reduce):
    """bp.compat_strict_upper_only: "strict" only asks t + tau_max < N; a used term in front of sample 0
    is dropped.  Differs from the default only where a used moveout is negative (never in BPMF, whose
    moveouts are relative to the first arrival): there the first samples of the trace get beams the
    default leaves uncomputed.  With non-negative moveouts the switch changes nothing and the planned
    kernels keep running."""
    from seismic_bpmf_amd import BeamformerGPU, beamform
    rng = np.random.default_rng(21)
    f, tau, wp, ws = _bp_case(rng)
    base = oracle_lib.beamform(f, tau, wp, ws, "strict", reduce)
    with oracle_lib.compat(oracle_lib.COMPAT_STRICT_UPPER_ONLY):
        want = oracle_lib.beamform(f, tau, wp, ws, "strict", reduce)
        flex = oracle_lib.beamform(f, tau, wp, ws, "flexible", reduce)
    b0, w0 = (base[0], want[0]) if reduce == "max" else (base, want)
    assert not np.array_equal(b0, w0) and np.array_equal(b0[..., 200:], w0[..., 200:])
    hip_opts("bp.compat_strict_upper_only", 1)
    got = beamform(f, tau, wp, ws, device="gpu", out_of_bounds="strict", reduce=reduce)
    gflex = beamform(f, tau, wp, ws, device="gpu", out_of_bounds="flexible", reduce=reduce)
    if reduce == "max":
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert np.array_equal(gflex[0], flex[0]) and np.array_equal(gflex[1], flex[1])
    else:
        assert np.array_equal(got, want) and np.array_equal(gflex, flex)
    # non-negative moveouts: same results as without the switch, still on an LDS plan
    tau2 = np.abs(tau)
    bf = BeamformerGPU(tau2, ws, device=0)
    assert bf.plan_info()["n_groups"] > 0
    bf.close()
    w2 = oracle_lib.beamform(f, tau2, wp, ws, "strict", reduce)
    g2 = beamform(f, tau2, wp, ws, device="gpu", out_of_bounds="strict", reduce=reduce)
    assert all(np.array_equal(a, b) for a, b in zip(np.atleast_1d(g2) if reduce == "none" else g2,
                                                    np.atleast_1d(w2) if reduce == "none" else w2))


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("direct", [0, 1])
def test_bp_range_all_stations(oracle_lib, hip_opts, fast, direct):
    """bp.compat_range_all_stations: a source's tau_min / tau_max for the strict test cover ALL its
    stations (the default: the weighted ones); zero-weight stations with extreme moveouts shorten the span
    of samples on which the source's beam is computed.  Planned kernels (fast interior + general edge
    tiles), the generic ones, and bp_direct.hip."""
    from seismic_bpmf_amd import beamform
    rng = np.random.default_rng(31)
    f, tau, wp, ws = _bp_case(rng, N=20_000, lo=0, hi=300)
    for k in range(0, len(ws), 3):                         # extreme moveouts on zero-weight stations
        s = int(np.flatnonzero(ws[k] == 0)[0]) if (ws[k] == 0).any() else None
        if s is not None and (ws[k] != 0).any():
            tau[k, s] = [-500 - k, 2_000 + k]
    base = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
    with oracle_lib.compat(oracle_lib.COMPAT_RANGE_ALL_STATIONS):
        want = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
        want_n = oracle_lib.beamform(f[:, :, :4000], tau, wp, ws, "strict", "none")
    assert not np.array_equal(base[0], want[0])
    hip_opts("bp.compat_range_all_stations", 1)
    hip_opts("bp.fast", fast)
    hip_opts("bp.direct", direct)
    got = beamform(f, tau, wp, ws, device="gpu", out_of_bounds="strict")
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(beamform(f[:, :, :4000], tau, wp, ws, device="gpu", out_of_bounds="strict", reduce="none"), want_n)
    hip_opts.reset("bp.compat_range_all_stations")
    got = beamform(f, tau, wp, ws, device="gpu", out_of_bounds="strict")
    assert np.array_equal(got[0], base[0]) and np.array_equal(got[1], base[1])


def test_compat_profile_sets_the_recollected_conventions_of_both_paths(oracle_lib):
    """seismic_bpmf_amd.compat_profile("upstream-recollected") (round 6): one call switches both hot paths to the
    five conventions SURVEY.md Appendix A recollects; the results equal the oracle's under the same flags, bit for
    bit, through the drop-in calls; "build" switches back."""
    import seismic_bpmf_amd as sb
    from seismic_bpmf_amd import _lib
    rng = np.random.default_rng(606)
    tp = rng.standard_normal((3, 4, 3, 96)).astype(np.float32)
    mv = rng.integers(-20, 400, (3, 4, 3)).astype(np.int32)
    w = rng.random((3, 4, 3)).astype(np.float32)
    w[0, 1] = 0.0
    mv[0, 1] = 3000                      # an unweighted channel that shapes the lag range under range_all_channels
    d = rng.standard_normal((4, 3, 20_000)).astype(np.float32)
    f = np.abs(rng.standard_normal((5, 2, 6000))).astype(np.float32)
    tau = rng.integers(0, 150, (60, 5, 2)).astype(np.int32)
    wp = np.zeros((5, 2, 2), np.float32)
    wp[:, 0, 0] = wp[:, 1, 1] = 1.0
    ws = (rng.random((60, 5)) < 0.6).astype(np.float32)
    assert sb.compat_profile() == "build"
    try:
        assert sb.compat_profile("upstream-recollected") == "upstream-recollected"
        assert sb.compat_profile() == "upstream-recollected"
        mf_flags = (oracle_lib.COMPAT_EXCLUSIVE_LAST_LAG | oracle_lib.COMPAT_SQRT_NORM |
                    oracle_lib.COMPAT_RANGE_ALL_CHANNELS | oracle_lib.COMPAT_SEQUENTIAL_CSUM)
        for ns in (True, False):
            got = sb.matched_filter(tp, mv, w, d, 1, check_zeros=False, network_sum=ns)
            with oracle_lib.compat(mf_flags):
                want = oracle_lib.matched_filter(tp, mv, w, d, 1, ns)
            assert np.array_equal(got, want), f"network_sum={ns}"
        gb, ga = sb.beamform(f, tau, wp, ws, device="gpu")
        with oracle_lib.compat(oracle_lib.COMPAT_FIRST_COMPUTED):
            wb, wa = oracle_lib.beamform(f, tau, wp, ws, "strict", "max")
        assert np.array_equal(gb, wb) and np.array_equal(ga, wa)
        _lib.set_option("mf.compat_sqrt_norm", 0)
        assert sb.compat_profile() is None               # a hand-made mixture has no name
    finally:
        sb.compat_profile("build")
    assert sb.compat_profile() == "build"
    assert np.array_equal(sb.matched_filter(tp, mv, w, d, 1, check_zeros=False), oracle_lib.matched_filter(tp, mv, w, d, 1, True))
    with pytest.raises(ValueError):
        sb.compat_profile("upstream")
