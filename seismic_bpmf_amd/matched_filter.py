"""Matched-filter call surface: a drop-in for ``fast_matched_filter.matched_filter``.

The reference calls that function at BPMF/similarity_search.py:526-533::

    fmf.matched_filter(templates, moveouts, weights, data, step, arch=device)

and at BPMF/dataset.py:4818-4827 with ``network_sum=False, check_zeros=False``.  This module
keeps the positional order, the keyword names and the return layouts, and runs the work on
the MI355X through libbpmf_hip.so.  There is no CPU path: ``arch="cpu"`` raises.

Two levels:
  * :func:`matched_filter` -- NumPy in / NumPy out (H2D, kernels, D2H inside the C ABI);
  * :class:`MatchedFilterGPU` -- device-resident: the day of data is uploaded and prepared
    once, template batches run against it, results stay in HBM (torch tensors are used only
    as device buffers).
"""
import ctypes as C

import numpy as np

from . import _lib

GPU_ARCHS = ("gpu", "hip", "mi355x")
_accept_cpu_arch = [False, False]        # [opted in, notice printed]


def accept_cpu_arch(accept=True):
    """Opt in to serving calls that ask for arch / device = "cpu" ON THE MI355X (one notice on stderr):
    the reference's call sites default to "cpu" (BPMF/similarity_search.py:476, 729;
    template_search.py:512), and a BPMF run that cannot be edited to pass "gpu" calls this once from
    its driver script.  There is still no CPU path.  (Rounds 2-3 read an environment variable for
    this; the package, like the library, takes no input from the process's variables now.)"""
    _accept_cpu_arch[0] = bool(accept)


def require_gpu_arch(value, keyword):
    """This package has no CPU implementation and never falls back to one: anything but a GPU name
    raises -- unless the user opted in with :func:`accept_cpu_arch`, which serves such calls on the
    MI355X, so that an unedited BPMF run with its default arguments works through the shims."""
    if str(value).lower() in GPU_ARCHS:
        return
    if _accept_cpu_arch[0]:
        if not _accept_cpu_arch[1]:
            _accept_cpu_arch[1] = True
            import sys
            print(f"seismic_bpmf_amd: {keyword}={value!r} requested, accept_cpu_arch() is on: "
                  "running on the MI355X (this package has no CPU path)", file=sys.stderr)
        return
    raise ValueError(
        f"{keyword}={value!r}: seismic_bpmf_amd only implements the MI355X path ({keyword}='gpu'); "
        "it has no CPU implementation (seismic_bpmf_amd.accept_cpu_arch() serves such calls on the GPU)")

FLAG_DATA_PREPARED = 1
FLAG_FORCE_DIRECT = 2


def n_corr_of(n_samples_data, n_samples_template, step):
    return (int(n_samples_data) - int(n_samples_template)) // int(step) + 1


def _prepare_host(templates, moveouts, weights, data):
    tp = np.ascontiguousarray(templates, dtype=np.float32)
    if tp.ndim != 4:
        raise ValueError("templates must be (n_templates, n_stations, n_components, n_samples)")
    T, S, Cc, L = tp.shape
    d = np.ascontiguousarray(data, dtype=np.float32)
    if d.ndim != 3 or d.shape[:2] != (S, Cc):
        raise ValueError(f"data must be ({S}, {Cc}, n_samples); got {d.shape}")
    # fast_matched_filter accepts (T, S) moveouts/weights and broadcasts over components
    mv = np.asarray(moveouts)
    w = np.asarray(weights)
    if mv.shape == (T, S):
        mv = np.repeat(mv[:, :, None], Cc, axis=2)
    if w.shape == (T, S):
        w = np.repeat(w[:, :, None], Cc, axis=2)
    if mv.shape != (T, S, Cc) or w.shape != (T, S, Cc):
        raise ValueError("moveouts/weights must be (T, S) or (T, S, C)")
    mv = np.ascontiguousarray(mv, dtype=np.int32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    return tp, mv, w, d


def _report_zeros(cc_sums, check_zeros):
    # fast_matched_filter prints how many CCs are exactly zero away from the edges
    # (the message stored in tutorial notebook 8, cell 18 output).
    rows = range(cc_sums.shape[0]) if check_zeros == "all" else range(min(1, cc_sums.shape[0]))
    for t in rows:
        row = cc_sums[t]
        nz = np.flatnonzero(row)
        if nz.size == 0:
            continue
        n_zeros = int(np.sum(row[nz[0]:nz[-1] + 1] == 0.0))
        if n_zeros > 10:
            print(f"{n_zeros} correlation computations were skipped on the {t}-th template. "
                  "Can be caused by zeros in data, or too low amplitudes "
                  "(try to increase the gain).")


def _device_list(devices):
    """None -> every visible GPU (what the upstream GPU back-end does), int -> that one, or a list."""
    if devices is None:
        n = _lib.lib().bpmf_device_count()
        if n <= 0:
            raise _lib.BpmfHipError("no HIP device visible")
        return list(range(n))
    if isinstance(devices, (int, np.integer)):
        return [int(devices)]
    return [int(d) for d in devices]


def _device_array(devices):
    """ctypes (n_devices, int*) pair for the *_run_multi entry points."""
    arr = (C.c_int * len(devices))(*devices)
    return len(devices), arr


def matched_filter(templates, moveouts, weights, data, step, arch="gpu", check_zeros="first",
                   normalize="short", network_sum=True, device=None, force_direct=False):
    """Sliding normalised cross-correlation of every template against the data.

    Parameters follow fast_matched_filter: ``templates (T,S,C,L)``, ``moveouts (T,S[,C])`` in
    samples, ``weights (T,S[,C])``, ``data (S,C,N)``, ``step`` in samples.
    Returns ``cc_sums (T, n_corr)`` float32, or ``cc (T, n_corr, S, C)`` when
    ``network_sum=False``; ``n_corr = (N - L)//step + 1`` and lag ``i`` is data sample
    ``i*step`` (BPMF/similarity_search.py:275).

    ``device``: None (default) = all visible GPUs, the templates block-partitioned among them
    inside the library (``bpmf_mf_run_multi``: one host thread per GPU; templates are independent,
    so no data crosses GPUs); an int or a list selects devices.
    """
    require_gpu_arch(arch, "arch")
    if normalize != "short":
        raise NotImplementedError("only normalize='short' (no window-mean removal) is implemented; "
                                  "it is the only mode the BPMF workflow uses")
    step = int(step)
    if step < 1:
        raise ValueError("step must be a positive number of samples")
    tp, mv, w, d = _prepare_host(templates, moveouts, weights, data)
    T, S, Cc, L = tp.shape
    N = d.shape[-1]
    if N < L:
        raise ValueError("data shorter than the templates")
    n_corr = n_corr_of(N, L, step)
    shape = (T, n_corr) if network_sum else (T, n_corr, S, Cc)
    out = np.empty(shape, dtype=np.float32)
    f, i = _lib._f, _lib._i
    flags = FLAG_FORCE_DIRECT if force_direct else 0
    lib = _lib.lib()

    n_dev, dev_arr = _device_array(_device_list(device))
    rc = lib.bpmf_mf_run_multi(tp.ctypes.data_as(f), mv.ctypes.data_as(i), w.ctypes.data_as(f),
                               d.ctypes.data_as(f), step, L, N, T, S, Cc, n_corr,
                               int(bool(network_sum)), flags, n_dev, dev_arr, out.ctypes.data_as(f))
    _lib.check(rc, "bpmf_mf_run_multi")
    if network_sum and check_zeros in ("first", "all"):
        _report_zeros(out, check_zeros)
    return out


class MatchedFilterGPU:
    """Device-resident matched filter: one day of data, many template batches.

    Mirrors how BPMF.similarity_search.MatchedFilter drives the back-end
    (set_data once per day :163-185, then template chunks :773-803), but keeps the data, its
    window energies and the CC matrix in HBM.
    """

    def __init__(self, device=None):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise _lib.BpmfHipError("MatchedFilterGPU needs a HIP device")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.lib = _lib.lib()
        self.data = None
        self._ws = None
        self._prepared_for = None

    def _dev(self, arr, dtype):
        t = self.torch
        if isinstance(arr, t.Tensor):
            return arr.to(device=self.device, dtype=dtype).contiguous()
        return t.as_tensor(np.ascontiguousarray(arr), dtype=dtype, device=self.device).contiguous()

    def set_data(self, data):
        """Upload (or adopt) the (S, C, N) float32 data; invalidates the prepared energies."""
        self.data = self._dev(data, self.torch.float32)
        if self.data.dim() != 3:
            raise ValueError("data must be (S, C, N)")
        self._prepared_for = None

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device)
            self._prepared_for = None
        return self._ws

    def workspace_bytes(self, L, T):
        S, Cc, N = self.data.shape
        return self.lib.bpmf_mf_workspace_bytes(L, N, T, S, Cc)

    def run(self, templates, moveouts, weights, step=1, network_sum=True, out=None,
            force_direct=False):
        """CC of a batch of templates against the resident data.  Returns a device tensor."""
        t = self.torch
        if self.data is None:
            raise RuntimeError("call set_data() first")
        tp = self._dev(templates, t.float32)
        T, S, Cc, L = tp.shape
        if tuple(self.data.shape[:2]) != (S, Cc):
            raise ValueError("templates and data disagree on (S, C)")
        N = self.data.shape[-1]
        mv = self._dev(moveouts, t.int32).reshape(T, S, -1).expand(T, S, Cc).contiguous()
        w = self._dev(weights, t.float32).reshape(T, S, -1).expand(T, S, Cc).contiguous()
        n_corr = n_corr_of(N, L, step)
        shape = (T, n_corr) if network_sum else (T, n_corr, S, Cc)
        if out is None:
            out = t.empty(shape, dtype=t.float32, device=self.device)
        elif tuple(out.shape) != shape or out.dtype != t.float32 or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous float32 tensor of shape {shape}")
        nbytes = self.lib.bpmf_mf_workspace_bytes(L, N, T, S, Cc)
        ws = self._workspace(nbytes)
        flags = FLAG_FORCE_DIRECT if force_direct else 0
        # (the prepared norm arrays hold energies under mf.compat_sqrt_norm, reciprocal norms otherwise; their
        # prefix sums are one chain under mf.compat_sequential_csum; under mf.split16 a prepared day also holds the
        # fp16 split of the data)
        key = (self.data.data_ptr(), int(N), int(L), ws.data_ptr(), _lib.get_option("mf.compat_sqrt_norm")[0],
               _lib.get_option("mf.compat_sequential_csum")[0], _lib.get_option("mf.split16")[0])
        if self._prepared_for == key:
            flags |= FLAG_DATA_PREPARED
        stream = t.cuda.current_stream(self.device).cuda_stream
        with t.cuda.device(self.device):
            rc = self.lib.bpmf_mf_run_dev(tp.data_ptr(), mv.data_ptr(), w.data_ptr(),
                                          self.data.data_ptr(), int(step), L, N, T, S, Cc, n_corr,
                                          int(bool(network_sum)), flags, ws.data_ptr(), ws.numel(),
                                          C.c_void_p(stream), out.data_ptr())
        _lib.check(rc, "bpmf_mf_run_dev")
        self._prepared_for = key
        # keep the inputs alive until the stream has consumed them
        self._keepalive = (tp, mv, w)
        return out
