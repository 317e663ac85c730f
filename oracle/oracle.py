"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; the product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f = C.POINTER(C.c_float)
_d = C.POINTER(C.c_double)
_i = C.POINTER(C.c_int32)
_lib_cache = {}


def build(march="x86-64-v3", out_dir=None, quiet=True):
    """Compile liboracle.so (and, in the container, _ref/libc.so).  Returns the .so path."""
    out_dir = out_dir or _HERE
    os.makedirs(out_dir, exist_ok=True)
    cmd = ["make", "-C", _HERE, f"MARCH={march}", f"OUT={os.path.abspath(out_dir)}"]
    subprocess.run(cmd, check=True, capture_output=quiet)
    return os.path.join(out_dir, "liboracle.so")


def build_ref(quiet=True):
    """Compile the reference's own BPMF/libc.c into oracle/_ref/libc.so (container only)."""
    subprocess.run(["make", "-C", _HERE, "ref"], check=True, capture_output=quiet)
    p = os.path.join(_HERE, "_ref", "libc.so")
    return p if os.path.exists(p) else None


def load(path=None):
    path = path or os.path.join(_HERE, "liboracle.so")
    if path in _lib_cache:
        return _lib_cache[path]
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    sz = C.c_size_t
    lib.mf_cpu.argtypes = [_f, _i, _f, _f, sz, sz, sz, sz, sz, sz, sz, C.c_int, C.c_int, _f]
    lib.mf_cpu.restype = C.c_int
    lib.mf_template_energy.argtypes = [_f, sz, sz, _f]
    lib.mf_data_csum.argtypes = [_f, sz, sz, _d]
    lib.mf_window_energy.argtypes = [_d, sz, sz, sz, _f]
    lib.bp_cpu.argtypes = [_f, _i, _f, _f, sz, sz, sz, sz, sz, C.c_int, C.c_int, C.c_int, _f, _i]
    lib.bp_cpu.restype = C.c_int
    lib.bp_prestack.argtypes = [_f, _f, sz, sz, sz, sz, _f]
    lib.bpmf_oracle_max_threads.restype = C.c_int
    lib.bpmf_oracle_set_compat.argtypes = [C.c_int]
    lib.bpmf_oracle_last_phase_seconds.argtypes = [C.POINTER(C.c_double)]
    lib.tdt_rms_cpu.argtypes = [_f, _f, C.c_float, sz, sz, sz, _f, _f]
    lib.tdt_rms_cpu.restype = C.c_long
    lib.select_cc_indexes_cpu.argtypes = [_f, _f, sz, sz, _i]
    lib.kurtosis_cpu.argtypes = [_f, C.c_int, C.c_int, C.c_int, C.c_int, _f]
    lib.similar_moveouts_cpu.argtypes = [_f, _f, _f, _f, _f, C.c_float, sz, sz, sz, sz, sz, C.c_int, _i]
    _lib_cache[path] = lib
    return lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def n_corr_of(N, L, step):
    return (N - L) // step + 1


def matched_filter(templates, moveouts, weights, data, step=1, network_sum=True,
                   num_threads=0, lib=None):
    """Oracle of fast_matched_filter.matched_filter (conventions: bpmf_oracle.c:mf_cpu)."""
    lib = lib or load()
    tp = _c(templates, np.float32)
    T, S, Cc, L = tp.shape
    mv = np.broadcast_to(np.asarray(moveouts).reshape(T, S, -1), (T, S, Cc))
    w = np.broadcast_to(np.asarray(weights).reshape(T, S, -1), (T, S, Cc))
    mv, w = _c(mv, np.int32), _c(w, np.float32)
    d = _c(data, np.float32)
    N = d.shape[-1]
    nc = n_corr_of(N, L, step)
    shape = (T, nc) if network_sum else (T, nc, S, Cc)
    out = np.empty(shape, dtype=np.float32)
    rc = lib.mf_cpu(_p(tp, _f), _p(mv, _i), _p(w, _f), _p(d, _f), step, L, N, T, S, Cc, nc,
                    int(bool(network_sum)), int(num_threads), _p(out, _f))
    if rc != 0:
        raise RuntimeError(f"mf_cpu failed rc={rc}")
    return out


def beamform(features, moveouts, w_phases, w_sources, out_of_bounds="strict", reduce="max",
             num_threads=0, lib=None):
    """Oracle of beampower.beamform (conventions: bpmf_oracle.c:bp_cpu)."""
    lib = lib or load()
    f = _c(features, np.float32)
    S, Cc, N = f.shape
    mv = _c(moveouts, np.int32)
    K, _, P = mv.shape
    wp = _c(w_phases, np.float32)
    ws = _c(w_sources, np.float32)
    oob = {"strict": 0, "flexible": 1}[out_of_bounds]
    red = {"max": 0, "none": 1}[reduce]
    if red == 0:
        beam = np.empty(N, dtype=np.float32)
        arg = np.empty(N, dtype=np.int32)
    else:
        beam = np.empty((K, N), dtype=np.float32)
        arg = np.empty(1, dtype=np.int32)
    rc = lib.bp_cpu(_p(f, _f), _p(mv, _i), _p(wp, _f), _p(ws, _f), N, K, S, Cc, P, oob, red,
                    int(num_threads), _p(beam, _f), _p(arg, _i))
    if rc != 0:
        raise RuntimeError(f"bp_cpu failed rc={rc}")
    return (beam, arg) if red == 0 else beam


def time_dependent_threshold(series, window, num_dev, overlap, white_noise, lib=None):
    """Oracle of BPMF.clib.time_dependent_threshold (rms), arguments as clib.py:257-309."""
    lib = lib or load()
    x = _c(series, np.float32)
    g = _c(white_noise, np.float32)
    assert g.size >= 500
    n = x.size
    half = window // 2
    shift = int((1.0 - overlap) * window)
    scratch = np.empty(n, dtype=np.float32)
    thr = np.zeros(n, dtype=np.float32)
    rc = lib.tdt_rms_cpu(_p(x, _f), _p(g, _f), float(num_dev), n, half, shift, _p(scratch, _f),
                         _p(thr, _f))
    if rc < 0:
        raise RuntimeError("tdt_rms_cpu: sizes admit no window")
    return thr


def select_cc_indexes(cc, threshold, search_win, lib=None):
    lib = lib or load()
    x = _c(cc, np.float32)
    thr = np.float32(threshold) * np.ones(x.size, np.float32) if np.isscalar(threshold) else _c(threshold, np.float32)
    sel = np.zeros(x.size, dtype=np.int32)
    lib.select_cc_indexes_cpu(_p(x, _f), _p(thr, _f), int(search_win), x.size, _p(sel, _i))
    return sel.astype(bool)


def kurtosis(signal, W, lib=None):
    lib = lib or load()
    x = _c(signal, np.float32)
    S, Cc, n = x.shape
    out = np.zeros_like(x)
    lib.kurtosis_cpu(_p(x, _f), int(W), S, Cc, n, _p(out, _f))
    return out


def find_similar_sources(moveouts, lon, lat, cell_lon, cell_lat, threshold, n_diff, method,
                         lib=None):
    lib = lib or load()
    mv = _c(moveouts, np.float32)
    K, S = mv.shape
    lon, lat = _c(lon, np.float32), _c(lat, np.float32)
    cl, ca = _c(cell_lon, np.float32), _c(cell_lat, np.float32)
    red = np.zeros(K, dtype=np.int32)
    mode = {"smallest": 0, "closest": 1}[method]
    lib.similar_moveouts_cpu(_p(mv, _f), _p(lon, _f), _p(lat, _f), _p(cl, _f), _p(ca, _f),
                             float(threshold), K, S, cl.size - 1, ca.size - 1, int(n_diff), mode,
                             _p(red, _i))
    return red.astype(bool)


COMPAT_EXCLUSIVE_LAST_LAG, COMPAT_SQRT_NORM, COMPAT_FIRST_COMPUTED = 1, 2, 4
COMPAT_RANGE_ALL_CHANNELS, COMPAT_SEQUENTIAL_CSUM, COMPAT_STRICT_UPPER_ONLY, COMPAT_RANGE_ALL_STATIONS = 8, 16, 32, 64
# oracle flag -> the library option that mirrors it (bpmf_set_option), and which path it belongs to
COMPAT_OPTIONS = {
    COMPAT_EXCLUSIVE_LAST_LAG: ("mf.compat_exclusive_last_lag", "mf"),
    COMPAT_SQRT_NORM: ("mf.compat_sqrt_norm", "mf"),
    COMPAT_RANGE_ALL_CHANNELS: ("mf.compat_range_all_channels", "mf"),
    COMPAT_SEQUENTIAL_CSUM: ("mf.compat_sequential_csum", "mf"),
    COMPAT_FIRST_COMPUTED: ("bp.compat_first_computed", "bp"),
    COMPAT_STRICT_UPPER_ONLY: ("bp.compat_strict_upper_only", "bp"),
    COMPAT_RANGE_ALL_STATIONS: ("bp.compat_range_all_stations", "bp"),
}


class compat:
    """with oracle.compat(flags): ... -- the upstream-compatibility variants of bpmf_oracle.c (mirrors
    of the library's *.compat_* options); back to the build's conventions on exit."""

    def __init__(self, flags, lib=None):
        self.flags, self.lib = int(flags), lib or load()

    def __enter__(self):
        self.old = self.lib.bpmf_oracle_get_compat()
        self.lib.bpmf_oracle_set_compat(self.flags)
        return self

    def __exit__(self, *exc):
        self.lib.bpmf_oracle_set_compat(self.old)
        return False
