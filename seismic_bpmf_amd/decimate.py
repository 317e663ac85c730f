"""Grid decimation on the MI355X: a drop-in for ``BPMF.clib.find_similar_sources``.

Reference: BPMF/clib.py:104-221 -> BPMF/libc.c:55-387, called from tutorial notebook 4 (cell 35)
to thin the backprojection grid; O(K^2 S) with a sequential greedy dependency.  Same signature and
semantics (``num_threads`` is accepted and ignored); the result equals the reference's
single-threaded run bit for bit.
"""
import numpy as np

from . import _lib

_METHOD = {"smallest": 0, "closest": 1}


def find_similar_sources(moveouts, source_longitude, source_latitude, cell_longitude, cell_latitude,
                         threshold, num_threads=None, num_stations_for_diff=None, method="closest",
                         device=0):
    """Boolean (K,) array: True where a source is redundant with a lower-indexed kept source."""
    del num_threads
    if method not in _METHOD:
        raise ValueError("method should be either of 'closest' or 'smallest'")
    mv = np.ascontiguousarray(moveouts, dtype=np.float32)
    if mv.ndim != 2:
        raise ValueError("moveouts must be (n_sources, n_stations), in seconds")
    K, S = mv.shape
    if num_stations_for_diff is None:
        num_stations_for_diff = S
    lon = np.ascontiguousarray(source_longitude, dtype=np.float32)
    lat = np.ascontiguousarray(source_latitude, dtype=np.float32)
    clon = np.ascontiguousarray(cell_longitude, dtype=np.float32)
    clat = np.ascontiguousarray(cell_latitude, dtype=np.float32)
    if lon.shape != (K,) or lat.shape != (K,) or clon.size < 2 or clat.size < 2:
        raise ValueError("source coordinates must be (n_sources,), cell vertices need >= 2 entries")
    red = np.zeros(K, dtype=np.int32)
    f, i = _lib._f, _lib._i
    rc = _lib.lib().bpmf_find_similar_sources(
        mv.ctypes.data_as(f), lon.ctypes.data_as(f), lat.ctypes.data_as(f), clon.ctypes.data_as(f),
        clat.ctypes.data_as(f), float(np.float32(threshold)), K, S, clon.size - 1, clat.size - 1,
        int(num_stations_for_diff), _METHOD[method], int(device), red.ctypes.data_as(i))
    _lib.check(rc, "bpmf_find_similar_sources")
    return red.astype(bool)
