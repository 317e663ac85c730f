"""Detection traces in front of the beamformer: the saturated envelopes of
BPMF/template_search.py:1525-1617 (`envelope`, `envelope_parallel`, `saturated_envelopes`), the
`waveform_features (S, C, N)` that `Beamformer.backproject` hands to `beampower.beamform`
("next" row 4).

The semantics are pinned by a NumPy/SciPy restatement under oracle/ (oracle/features_host.py, test
infrastructure; bit for bit equal to the reference's own output, tests/golden/saturated_envelopes.npz).

* :func:`saturated_envelopes` -- on the MI355X: the analytic signal through a float64 FFT
  (hipFFT behind ``torch.fft``), the per-channel median / MAD by a radix select on the device
  (csrc/stats.hip, one launch for all channels), everything else element-wise.  Median, MAD, standardisation and clipping reproduce NumPy's float32 arithmetic
  exactly.  The FFT cannot be reproduced bit for bit: the reference hands float32 traces to
  ``scipy.signal.hilbert``, and scipy.fft keeps that precision, so ITS envelopes carry a float32
  FFT's round-off -- a few ulp of the channel's largest value on every sample
  (tests/test_features.py measures it against a float64 FFT).  The device path is within 1 ulp of
  the float64 result, i.e. it differs from the reference by the reference's own round-off
  (tolerance in the tests: 16 ulp of the channel maximum).

A day of 60 channels at 50 Hz takes the reference's process pool minutes; here ~0.2 s.
"""
import numpy as np

from . import _lib


def _analytic_weights(n, xp, **kw):
    """The one-sided spectrum weights of scipy.signal.hilbert (h[0] = 1, h[1:n/2] = 2, h[n/2] = 1
    for even n; h[1:(n+1)/2] = 2 for odd n)."""
    h = xp.zeros(n, **kw)
    if n % 2 == 0:
        h[0] = 1.0
        h[n // 2] = 1.0
        h[1:n // 2] = 2.0
    else:
        h[0] = 1.0
        h[1:(n + 1) // 2] = 2.0
    return h


# ------------------------------------------------------------------------ device path ---
def _torch_device(device):
    import torch
    if not torch.cuda.is_available():
        raise _lib.BpmfHipError("seismic_bpmf_amd.features needs a HIP device (no CPU implementation)")
    return torch, torch.device("cuda", torch.cuda.current_device() if device is None else int(device))


def envelope_c2c(traces, device=None, channels_per_batch=16):
    """|analytic signal| of every channel of `traces (..., N)`; float32 device tensor.  The complex-to-complex
    float64 route of round 3 (forward FFT, one-sided weights, inverse FFT): kept as the cross-check of
    envelope()."""
    torch, dev = _torch_device(device)
    x = traces if isinstance(traces, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(traces))
    x = x.to(device=dev)
    shape, n = x.shape, x.shape[-1]
    x = x.reshape(-1, n)
    h = _analytic_weights(n, torch, dtype=torch.float64, device=dev)
    out = torch.empty(x.shape, dtype=torch.float32, device=dev)
    for i in range(0, x.shape[0], channels_per_batch):     # bounds the complex128 work space
        X = torch.fft.fft(x[i:i + channels_per_batch].to(torch.float64), dim=-1)
        out[i:i + channels_per_batch] = torch.fft.ifft(X * h, dim=-1).abs().to(torch.float32)
    return out.reshape(shape)


def envelope(traces, device=None, channels_per_batch=None, precision="float64"):
    """|analytic signal| of every channel of `traces (..., N)`; float32 device tensor.

    envelope = sqrt(x^2 + H[x]^2) with the Hilbert transform through a REAL-input FFT pair (hipFFT behind
    torch.fft): X = rfft(x), H[x] = irfft(-i X) with the DC and Nyquist bins cleared -- the same analytic signal
    as scipy.signal.hilbert's one-sided spectrum (BPMF/template_search.py:1598-1617), at half the transform
    work and a third of the work space of the complex pair (envelope_c2c).  float64 by default: the result is
    the correctly rounded float32 of the exact envelope on all but a few samples in a million (the reference's
    own float32 FFT is several ulp of the channel maximum away from it, module docstring); "float32" halves the
    time again and lands within the same few ulp as the reference itself.
    `channels_per_batch`: None = as many channels as ~6 GB of work space hold."""
    torch, dev = _torch_device(device)
    x = traces if isinstance(traces, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(traces))
    x = x.to(device=dev)
    shape, n = x.shape, x.shape[-1]
    x = x.reshape(-1, n)
    dtype = {"float64": torch.float64, "float32": torch.float32}[precision]
    if channels_per_batch is None:
        per_channel = n * (8 if dtype == torch.float64 else 4) * 5        # x, X, -iX, H[x], temporaries
        channels_per_batch = max(1, min(x.shape[0], int(6e9 // max(per_channel, 1))))
    out = torch.empty(x.shape, dtype=torch.float32, device=dev)
    fused = dtype == torch.float64 and x.dtype == torch.float32 and x.is_contiguous()
    if fused:
        import ctypes as C
        lib = _lib.lib()
        channels_per_batch = min(channels_per_batch, 65535)
    for i in range(0, x.shape[0], channels_per_batch):
        xb = x[i:i + channels_per_batch].to(dtype)
        X = torch.fft.rfft(xb, dim=-1)
        if fused:
            # -i X with the DC (and Nyquist) bin cleared, in place; then sqrt(x^2 + h^2) from the float32 trace and
            # the float64 transform: one pass each (csrc/stats.hip; the tensor expressions below made ten)
            del xb
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            with torch.cuda.device(dev):
                _lib.check(lib.bpmf_hilbert_spectrum_dev(C.c_void_p(X.data_ptr()), X.shape[0], X.shape[1], int(n % 2 == 0),
                                                         stream), "bpmf_hilbert_spectrum_dev")
                h = torch.fft.irfft(X, n=n, dim=-1)
                del X
                xs, os_ = x[i:i + channels_per_batch], out[i:i + channels_per_batch]
                _lib.check(lib.bpmf_envelope_combine_dev(C.c_void_p(xs.data_ptr()), C.c_void_p(h.data_ptr()), xs.numel(),
                                                         stream, C.c_void_p(os_.data_ptr())), "bpmf_envelope_combine_dev")
            del h
            continue
        X[:, 0] = 0                      # the DC bin (and, for even n, the Nyquist bin) has no quadrature part
        if n % 2 == 0:
            X[:, -1] = 0
        h = torch.fft.irfft(torch.complex(X.imag, -X.real), n=n, dim=-1)     # -i X
        del X
        out[i:i + channels_per_batch] = torch.sqrt_(xb * xb + h * h).to(torch.float32)
    return out.reshape(shape)


def row_median_mad(x, skip_zeros=False, device=None):
    """np.median and MAD of every row of a (rows, n) float32 device tensor (over the non-zero
    samples only when `skip_zeros`), exact order statistics on the device (csrc/stats.hip: long rows
    in two reads by the whole chip, short ones by a radix select).  Returns (median, mad, n_zero) device
    tensors; nothing is synchronised."""
    import ctypes as C
    torch, dev = _torch_device(device)
    x = x.to(device=dev, dtype=torch.float32).contiguous()
    rows, n = x.shape
    med = torch.empty(rows, dtype=torch.float32, device=dev)
    mad = torch.empty(rows, dtype=torch.float32, device=dev)
    nz = torch.empty(rows, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    lib = _lib.lib()
    # (long rows are read twice by the whole chip instead of seven times by one workgroup each: that path
    # collects the middle of a row in a workspace, about a tenth of the size of the rows)
    ws = torch.empty(lib.bpmf_row_median_mad_workspace_bytes(rows, n), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.bpmf_row_median_mad_ws_dev(C.c_void_p(x.data_ptr()), rows, n, int(bool(skip_zeros)),
                                            C.c_void_p(ws.data_ptr()), ws.numel(),
                                            C.c_void_p(stream), C.c_void_p(med.data_ptr()),
                                            C.c_void_p(mad.data_ptr()), C.c_void_p(nz.data_ptr()))
    _lib.check(rc, "bpmf_row_median_mad_ws_dev")
    ws.record_stream(torch.cuda.current_stream(dev))
    return med, mad, nz


def saturated_envelopes(traces, anomaly_threshold=1.0e-11, max_dynamic_range=1.0e5, device=None):
    """Device version of BPMF/template_search.py:1525-1572.  Returns (features (S, C, N) float32
    device tensor, data_availability (S,) int32 NumPy array).  The median and MAD of the valid
    samples of all channels come from ONE call (row_median_mad, csrc/stats.hip); the decisions per
    channel (more than half missing, MAD below the anomaly threshold) are taken on the device, and
    the only transfer to the host is the (S,) availability vector at the end."""
    torch, dev = _torch_device(device)
    wf = envelope(traces, device=device)
    n_stations, n_components, n_samples = wf.shape
    rows = wf.reshape(n_stations * n_components, n_samples)
    median, mad, n_missing = row_median_mad(rows, skip_zeros=True, device=dev.index)
    # channels the reference zeroes: more than half of the samples missing, or MAD < threshold (a NaN
    # MAD -- no valid sample -- only occurs together with the first condition)
    dead = (n_missing.to(torch.float64) > n_samples / 2) | ~(mad.to(torch.float64) >= anomaly_threshold)
    # (x - median) / MAD in float32 (NumPy's operations), 0 for missing samples and dead channels, capped: one pass,
    # in place over the envelopes (csrc/stats.hip saturate_rows_kernel; it was five element-wise passes of torch)
    import ctypes as C
    out = rows
    dead_i = dead.to(torch.int32).contiguous()
    with torch.cuda.device(dev):
        for r0 in range(0, rows.shape[0], 65535):
            nr = min(65535, rows.shape[0] - r0)
            rc = _lib.lib().bpmf_saturate_rows_dev(C.c_void_p(rows[r0:].data_ptr()), C.c_void_p(median[r0:].data_ptr()),
                                                   C.c_void_p(mad[r0:].data_ptr()), C.c_void_p(dead_i[r0:].data_ptr()),
                                                   nr, n_samples, float(np.float32(max_dynamic_range)),
                                                   C.c_void_p(torch.cuda.current_stream(dev).cuda_stream),
                                                   C.c_void_p(out[r0:].data_ptr()))
            _lib.check(rc, "bpmf_saturate_rows_dev")
    availability = (~dead).reshape(n_stations, n_components).sum(dim=1).to(torch.int32).cpu().numpy()
    return out.reshape(n_stations, n_components, n_samples), availability


def kurtosis(signal, W, device=None):
    """Device version of BPMF.clib.kurtosis(signal, W) (BPMF/clib.py:86-102): running kurtosis of
    `signal (n_stations, n_components, length)` over the W samples before each sample.  Returns a
    float32 device tensor of the same shape, zero where the reference leaves its zeros."""
    import ctypes as C
    torch, dev = _torch_device(device)
    x = signal if isinstance(signal, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(signal, dtype=np.float32))
    x = x.to(device=dev, dtype=torch.float32).contiguous()
    if x.dim() != 3:
        raise ValueError("signal must be (n_stations, n_components, length)")
    out = torch.zeros_like(x)
    stream = torch.cuda.current_stream(dev).cuda_stream
    rc = _lib.lib().bpmf_kurtosis_dev(C.c_void_p(x.data_ptr()), int(W), x.shape[0] * x.shape[1], x.shape[2],
                                      C.c_void_p(stream), C.c_void_p(out.data_ptr()))
    _lib.check(rc, "bpmf_kurtosis_dev")
    return out
