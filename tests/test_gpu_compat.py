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
