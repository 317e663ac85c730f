"""Backprojection call surface: a drop-in for ``beampower.beampower.beamform``.

The reference calls it at BPMF/template_search.py:549-558 and :560-569::

    bp.beampower.beamform(waveform_features, moveouts, weights_phases, weights_sources,
                          device=device, out_of_bounds=out_of_bounds,
                          num_threads=num_threads, reduce=reduce)

``reduce="max"`` returns ``(maxbeam float32[N], argmax int32[N])``; ``reduce="none"`` returns
``beam float32[K, N]``.  Work runs on the MI355X through libbpmf_hip.so; ``device="cpu"``
raises (no CPU path in this package).

:class:`BeamformerGPU` is the device-resident form: the moveout table / source weights are
planned and uploaded once (the reference rebuilds its table on every access,
template_search.py:444-454) and then reused for every day of features.
"""
import ctypes as C

import numpy as np

from . import _lib

GPU_DEVICES = ("gpu", "hip", "mi355x")
_OOB = {"strict": 0, "flexible": 1}
_REDUCE = {"max": 0, "none": 1}


def _check_shapes(features, moveouts, w_phases, w_sources):
    f = np.ascontiguousarray(features, dtype=np.float32)
    if f.ndim != 3:
        raise ValueError("waveform_features must be (n_stations, n_components, n_samples)")
    S, Cc, N = f.shape
    mv = np.ascontiguousarray(moveouts, dtype=np.int32)  # delivered as int64 by BPMF (utils.py:1270)
    if mv.ndim != 3 or mv.shape[1] != S:
        raise ValueError("moveouts must be (n_sources, n_stations, n_phases)")
    K, _, P = mv.shape
    wp = np.ascontiguousarray(w_phases, dtype=np.float32)
    if wp.shape != (S, Cc, P):
        raise ValueError(f"weights_phases must be ({S}, {Cc}, {P}); got {wp.shape}")
    ws = np.ascontiguousarray(w_sources, dtype=np.float32)
    if ws.shape != (K, S):
        raise ValueError(f"weights_sources must be ({K}, {S}); got {ws.shape}")
    return f, mv, wp, ws


def beamform(waveform_features, time_delays, weights_phases, weights_sources, device="gpu",
             reduce="max", mode="direct", out_of_bounds="strict", num_threads=None,
             device_id=None):
    """Shift-and-stack beam power over a grid of sources (see module docstring).

    ``device_id``: None (default) = all visible GPUs, the source grid block-partitioned among them
    inside the library (``bpmf_bp_run_multi``: one host thread per GPU) and, for ``reduce="max"``,
    the per-GPU maxima merged in block order with a strict ``>`` -- blocks hold ascending source
    indices, so ties keep the lowest index exactly like a single pass; an int or a list selects
    devices."""
    del num_threads  # CPU-only knob of the reference; accepted for call compatibility
    from .matched_filter import require_gpu_arch
    require_gpu_arch(device, "device")
    if mode != "direct":
        raise NotImplementedError("only mode='direct' is implemented")
    if reduce not in _REDUCE:
        raise ValueError("reduce should be 'max' or 'none'")
    if out_of_bounds not in _OOB:
        raise ValueError("out_of_bounds should be 'strict' or 'flexible'")
    from .matched_filter import _device_array, _device_list
    f, mv, wp, ws = _check_shapes(waveform_features, time_delays, weights_phases, weights_sources)
    S, Cc, N = f.shape
    K, _, P = mv.shape
    pf, pi = _lib._f, _lib._i
    lib = _lib.lib()
    n_dev, dev_arr = _device_array(_device_list(device_id))
    if reduce == "none":
        beam = np.empty((K, N), dtype=np.float32)
        arg = None
    else:
        beam = np.empty(N, dtype=np.float32)
        arg = np.empty(N, dtype=np.int32)
    rc = lib.bpmf_bp_run_multi(f.ctypes.data_as(pf), mv.ctypes.data_as(pi), wp.ctypes.data_as(pf),
                               ws.ctypes.data_as(pf), N, K, S, Cc, P, _OOB[out_of_bounds],
                               _REDUCE[reduce], n_dev, dev_arr, beam.ctypes.data_as(pf),
                               arg.ctypes.data_as(pi) if arg is not None else None)
    _lib.check(rc, "bpmf_bp_run_multi")
    return beam if reduce == "none" else (beam, arg)


class BeamformerGPU:
    """Device-resident beamformer bound to one moveout table and one set of source weights."""

    def __init__(self, moveouts, weights_sources, device=None, source_id_offset=0):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise _lib.BpmfHipError("BeamformerGPU needs a HIP device")
        idx = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", idx)
        self.lib = _lib.lib()
        mv = np.ascontiguousarray(moveouts, dtype=np.int32)
        ws = np.ascontiguousarray(weights_sources, dtype=np.float32)
        if mv.ndim != 3 or ws.shape != mv.shape[:2]:
            raise ValueError("moveouts must be (K, S, P) and weights_sources (K, S)")
        self.K, self.S, self.P = mv.shape
        self._plan = C.c_void_p()
        rc = self.lib.bpmf_bp_plan_create(mv.ctypes.data_as(_lib._i), ws.ctypes.data_as(_lib._f),
                                          self.K, self.S, self.P, idx, int(source_id_offset),
                                          C.byref(self._plan))
        _lib.check(rc, "bpmf_bp_plan_create")
        self._ws = None

    def plan_info(self):
        """Shape of the device plan: dict of bpmf_bp_plan_stats (include/bpmf_hip.h)."""
        st = _lib.BpPlanStats()
        _lib.check(self.lib.bpmf_bp_plan_info(self._plan, C.byref(st)), "bpmf_bp_plan_info")
        out = {}
        for n, _ in st._fields_:
            v = getattr(st, n)
            out[n] = int(v) if isinstance(v, int) else [int(x) for x in v]
        return out

    def close(self):
        if getattr(self, "_plan", None) is not None and self._plan.value:
            self.lib.bpmf_bp_plan_destroy(self._plan)
            self._plan = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _dev(self, arr, dtype):
        t = self.torch
        if isinstance(arr, t.Tensor):
            return arr.to(device=self.device, dtype=dtype).contiguous()
        return t.as_tensor(np.ascontiguousarray(arr), dtype=dtype, device=self.device).contiguous()

    def run(self, features, weights_phases, reduce="max", out_of_bounds="strict", out=None):
        """Beamform one (S, C, N) feature array.  Returns device tensors."""
        t = self.torch
        f = self._dev(features, t.float32)
        S, Cc, N = f.shape
        if S != self.S:
            raise ValueError("features and moveouts disagree on the number of stations")
        wp = self._dev(weights_phases, t.float32)
        if tuple(wp.shape) != (S, Cc, self.P):
            raise ValueError(f"weights_phases must be ({S}, {Cc}, {self.P})")
        nbytes = self.lib.bpmf_bp_workspace_bytes(self._plan, N, Cc)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = t.empty(nbytes, dtype=t.uint8, device=self.device)
        if reduce == "max":
            beam, arg = out if out is not None else (
                t.empty(N, dtype=t.float32, device=self.device),
                t.empty(N, dtype=t.int32, device=self.device))
            arg_ptr = arg.data_ptr()
        else:
            beam = out if out is not None else t.empty((self.K, N), dtype=t.float32,
                                                       device=self.device)
            arg, arg_ptr = None, None
        stream = t.cuda.current_stream(self.device).cuda_stream
        with t.cuda.device(self.device):
            rc = self.lib.bpmf_bp_run_dev(self._plan, f.data_ptr(), wp.data_ptr(), N, Cc,
                                          _OOB[out_of_bounds], _REDUCE[reduce],
                                          self._ws.data_ptr(), self._ws.numel(),
                                          C.c_void_p(stream), beam.data_ptr(), arg_ptr)
        _lib.check(rc, "bpmf_bp_run_dev")
        self._keepalive = (f, wp)
        return (beam, arg) if reduce == "max" else beam

    # -- multi-GPU exchange step of reduce="max"  ----------------
    def pack_max(self, beam, arg):
        """(beam, arg) -> int64 keys whose max is (largest beam, lowest source id)."""
        t = self.torch
        packed = t.empty(beam.numel(), dtype=t.int64, device=self.device)
        stream = t.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.bpmf_bp_pack_max_dev(beam.data_ptr(), arg.data_ptr(), beam.numel(), 1,
                                                 C.c_void_p(stream), packed.data_ptr()),
                   "bpmf_bp_pack_max_dev")
        return packed

    def unpack_max(self, packed):
        t = self.torch
        n = packed.numel()
        beam = t.empty(n, dtype=t.float32, device=self.device)
        arg = t.empty(n, dtype=t.int32, device=self.device)
        stream = t.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.bpmf_bp_unpack_max_dev(packed.data_ptr(), n, 1, C.c_void_p(stream),
                                                   beam.data_ptr(), arg.data_ptr()),
                   "bpmf_bp_unpack_max_dev")
        return beam, arg
