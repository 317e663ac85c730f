"""Saturated envelopes (BPMF/template_search.py:1525-1617): the host restatement is pinned bit for
bit to the reference's own output; the device version is compared with it on the GPU box."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "saturated_envelopes.npz")


def test_host_restatement_equals_the_reference():
    from oracle.features_host import envelope_host, saturated_envelopes_host
    g = np.load(GOLD)
    for j in range(int(g["n_cases"])):
        tr = g[f"traces_{j}"]
        env = np.float32([envelope_host(x) for x in tr.reshape(-1, tr.shape[-1])]).reshape(tr.shape)
        assert np.array_equal(env, g[f"envelope_{j}"])
        feat, avail = saturated_envelopes_host(tr.copy())
        assert feat.dtype == np.float32 and np.array_equal(feat, g[f"features_{j}"])
        assert np.array_equal(avail, g[f"availability_{j}"])
        assert feat.max() == g[f"features_{j}"].max()
        if j == 0:
            assert feat.max() == np.float32(1.0e5)         # the spike saturates
        assert not feat[1].any() and avail[1] == 0         # dead / anomalous channels dropped


def test_device_path_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from seismic_bpmf_amd import _lib
    from seismic_bpmf_amd.features import saturated_envelopes
    with pytest.raises(_lib.BpmfHipError):
        saturated_envelopes(np.zeros((1, 1, 16), np.float32))


def _envelope_f64(tr):
    from scipy.signal import hilbert
    return np.abs(hilbert(tr.astype(np.float64), axis=-1))


def test_reference_fft_is_single_precision():
    """scipy.fft keeps the input precision, so the reference's hilbert(float32 trace) is a float32
    FFT: its envelopes carry round-off of a few float32 ulp OF THE CHANNEL'S LARGEST VALUE
    everywhere (a 3e6 spike puts +-0.5 on every sample).  This bounds how closely any other
    implementation can agree with the golden file, and is the tolerance of the device test."""
    g = np.load(GOLD)
    for j in range(int(g["n_cases"])):
        tr, gold = g[f"traces_{j}"], g[f"envelope_{j}"]
        exact = _envelope_f64(tr)
        ulp_max = np.spacing(np.abs(gold).max(axis=-1, keepdims=True).astype(np.float32))
        err = np.abs(gold - exact) / ulp_max
        assert err.max() <= 16.0
        assert err.max() > 1.0          # not a correctly rounded float64 result


@pytest.mark.gpu
def test_device_envelopes_against_float64_and_the_reference_output():
    """Tight: the device envelope (float64 hipFFT, cast to float32) equals the float64 SciPy
    envelope rounded to float32 to within 1 ulp.  Loose: against the reference's float32-FFT
    output the bound is the reference's own round-off, 16 ulp of the channel's largest envelope
    value (see test_reference_fft_is_single_precision); the standardised features inherit it
    divided by the channel's MAD.  Availability and the dropped channels match exactly."""
    from scipy.stats import median_abs_deviation as scimad
    from seismic_bpmf_amd.features import envelope, saturated_envelopes
    g = np.load(GOLD)
    for j in range(int(g["n_cases"])):
        tr = g[f"traces_{j}"]
        env = envelope(tr).cpu().numpy()
        exact = _envelope_f64(tr)
        assert np.all(np.abs(env - exact) <= np.spacing(np.float32(exact)) * 1.0001 + 1e-300), j
        gold_env = g[f"envelope_{j}"]
        ulp_max = np.spacing(np.abs(gold_env).max(axis=-1, keepdims=True).astype(np.float32))
        assert np.all(np.abs(env.astype(np.float64) - gold_env) <= 16 * ulp_max), j
        feat, avail = saturated_envelopes(tr)
        feat = feat.cpu().numpy()
        gold = g[f"features_{j}"]
        assert np.array_equal(avail, g[f"availability_{j}"])
        assert np.array_equal((feat != 0).any(axis=-1), (gold != 0).any(axis=-1))   # dropped channels
        for s in range(tr.shape[0]):
            for c in range(tr.shape[1]):
                if not gold[s, c].any():
                    continue
                mad = scimad(gold_env[s, c][gold_env[s, c] != 0])
                noise = 40 * ulp_max[s, c, 0] / mad      # the reference's FFT round-off in MAD units
                # where that noise is large (the channel with the 3e6 spike: +-3 MAD on every
                # sample) it also moves the reference's own median and MAD by a few per cent
                rel = 1e-5 if noise < 1e-3 else 5e-2
                bound = noise + rel * np.abs(gold[s, c])
                assert np.all(np.abs(feat[s, c].astype(np.float64) - gold[s, c]) <= bound), (j, s, c)
        if j == 0:
            assert feat.max() == np.float32(1.0e5)


@pytest.mark.gpu
def test_device_kurtosis_equals_the_reference_output(oracle_lib):
    """bpmf_kurtosis_dev against the golden produced by the reference's compiled libc.c, and
    against the oracle on a second shape (W not a multiple of anything, short trailing block)."""
    from seismic_bpmf_amd.features import kurtosis
    g = np.load(os.path.join(os.path.dirname(GOLD), "kurtosis.npz"))
    got = kurtosis(g["signal"], int(g["W"])).cpu().numpy()
    want = [g[k] for k in g.files if k not in ("signal", "W")][0]
    assert got.shape == want.shape and np.array_equal(got, want)
    rng = np.random.default_rng(4)
    x = (rng.standard_normal((2, 3, 3001)) * np.array([1.0, 1e-4, 30.0])[None, :, None]).astype(np.float32)
    assert np.array_equal(kurtosis(x, 37).cpu().numpy(), oracle_lib.kurtosis(x, 37))


@pytest.mark.gpu
def test_saturate_rows_kernel_equals_the_element_wise_formulation():
    """saturated_envelopes' last step is one kernel (csrc/stats.hip saturate_rows_kernel); it must give the bits of
    the element-wise float32 formulation it replaced -- (x - median) / MAD, 0 for missing samples and dead channels,
    np.minimum against the cap -- on channels with gaps, a dead channel, a channel that is mostly missing, a spike
    beyond the cap, a NaN."""
    import torch
    from seismic_bpmf_amd.features import envelope, row_median_mad, saturated_envelopes
    rng = np.random.default_rng(12)
    tr = rng.standard_normal((3, 3, 20_011)).astype(np.float32)
    tr[0, 0, 5_000:9_000] = 0.0
    tr[0, 1] = 0.0
    tr[1, 0, : 15_000] = 0.0
    tr[1, 1, 7_777] = 3.0e7
    tr[2, 2] *= 1.0e-14                         # MAD below the anomaly threshold
    tr[2, 0, 100] = np.nan
    feat, avail = saturated_envelopes(tr, max_dynamic_range=1.0e3)
    wf = envelope(tr)
    rows = wf.reshape(9, -1)
    median, mad, n_missing = row_median_mad(rows, skip_zeros=True)
    dead = (n_missing.to(torch.float64) > rows.shape[1] / 2) | ~(mad.to(torch.float64) >= 1.0e-11)
    std = (rows - median[:, None]) / mad[:, None]
    std = torch.where(rows == 0.0, torch.zeros((), device=rows.device), std)
    want = torch.where(dead[:, None], torch.zeros((), device=rows.device),
                       torch.minimum(std, torch.tensor(1.0e3, dtype=torch.float32, device=rows.device)))
    assert np.array_equal(feat.cpu().numpy().reshape(9, -1), want.cpu().numpy(), equal_nan=True)
    assert np.array_equal(avail, (~dead).reshape(3, 3).sum(dim=1).cpu().numpy())
    assert float(feat.max()) == 1.0e3                      # (the spike of channel (1, 1) is beyond the cap)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [16, 1001, 20_000, 131_071])
def test_envelope_fused_steps_equal_the_tensor_expressions(n):
    """envelope() on float32 traces runs its two element-wise steps as one kernel each (-i X in place with the DC /
    Nyquist bins cleared; sqrt(x^2 + h^2) from the float32 trace and the float64 transform: csrc/stats.hip); on a
    float64 copy of the same traces it takes the tensor expressions.  Same FFT calls, same IEEE operations: same bits,
    for even and odd lengths, with a zero channel, a gap and a large offset."""
    import torch
    from seismic_bpmf_amd.features import envelope
    rng = np.random.default_rng(n)
    tr = rng.standard_normal((2, 3, n)).astype(np.float32)
    tr[0, 1] = 0.0
    tr[1, 0, n // 3: n // 2] = 0.0
    tr[1, 2] += 1000.0
    fused = envelope(tr).cpu().numpy()
    plain = envelope(torch.as_tensor(tr).double()).cpu().numpy()
    assert fused.dtype == np.float32 and np.array_equal(fused, plain)
