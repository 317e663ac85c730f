"""ctypes binding of libbpmf_hip.so (the C ABI declared in include/bpmf_hip.h).

Follows the reference's own binding style (BPMF/clib.py:14-84: CDLL + explicit argtypes,
caller-allocated outputs) with one deliberate difference: a missing library is an ERROR
here, not a printed warning -- this package has no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(_HERE, "lib", "libbpmf_hip.so")

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int32)
_u64 = C.POINTER(C.c_uint64)
_sz = C.c_size_t
_vp = C.c_void_p

# name -> (restype, argtypes); must list every symbol of include/bpmf_hip.h
class BpPlanStats(C.Structure):
    """bpmf_bp_plan_stats of include/bpmf_hip.h."""
    _fields_ = ([(n, C.c_int32) for n in ("n_groups", "tile", "lds_bytes", "gather_bytes",
                                          "stations_max", "waves_per_cu", "n_classes")] +
                [(n, C.c_int32 * 3) for n in ("class_tile", "class_sources", "class_groups",
                                              "class_stations_max")])


SIGNATURES = {
    "bpmf_last_error": (C.c_char_p, []),
    "bpmf_device_count": (C.c_int, []),
    "bpmf_device_info": (C.c_int, [C.c_int, C.c_char_p, _sz, C.POINTER(_sz), C.POINTER(C.c_int)]),
    "bpmf_set_option": (C.c_int, [C.c_char_p, C.c_long]),
    "bpmf_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "bpmf_profile_enable": (None, [C.c_int]),
    "bpmf_profile_count": (C.c_int, [C.c_int]),
    "bpmf_profile_get_ms": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "bpmf_profile_get_device": (C.c_int, [C.c_int, C.c_int]),
    "bpmf_release_device_memory": (C.c_int, [C.c_int]),
    "bpmf_device_memory_held": (C.c_int, [C.c_int, C.POINTER(_sz), C.POINTER(_sz)]),
    "bpmf_host_call_stats": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
    "bpmf_mf_workspace_bytes": (_sz, [_sz, _sz, _sz, _sz, _sz]),
    "bpmf_mf_prepare_data_dev": (C.c_int, [_vp, _sz, _sz, _sz, _sz, _vp, _sz, _vp]),
    "bpmf_mf_run_dev": (C.c_int, [_vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _sz, C.c_int,
                                  C.c_int, _vp, _sz, _vp, _vp]),
    "bpmf_mf_run": (C.c_int, [_f, _i, _f, _f, _sz, _sz, _sz, _sz, _sz, _sz, _sz, C.c_int, C.c_int,
                              C.c_int, _f]),
    "bpmf_mf_run_multi": (C.c_int, [_f, _i, _f, _f, _sz, _sz, _sz, _sz, _sz, _sz, _sz, C.c_int, C.c_int,
                                    C.c_int, C.POINTER(C.c_int), _f]),
    "bpmf_mf_shard_bounds": (C.c_int, [_f, _sz, _sz, _sz, _sz, C.POINTER(_sz)]),
    "bpmf_bp_run_multi": (C.c_int, [_f, _i, _f, _f, _sz, _sz, _sz, _sz, _sz, C.c_int, C.c_int, C.c_int,
                                    C.POINTER(C.c_int), _f, _i]),
    "bpmf_bp_plan_create": (C.c_int, [_i, _f, _sz, _sz, _sz, C.c_int, C.c_int32, C.POINTER(_vp)]),
    "bpmf_kurtosis_dev": (C.c_int, [_vp, C.c_int, _sz, _sz, _vp, _vp]),
    "bpmf_suppress_peaks": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int64), _sz, C.c_double,
                                      C.POINTER(C.c_uint8)]),
    "bpmf_bp_plan_destroy": (None, [_vp]),
    "bpmf_bp_plan_info": (C.c_int, [_vp, C.POINTER(BpPlanStats)]),
    "bpmf_bp_workspace_bytes": (_sz, [_vp, _sz, _sz]),
    "bpmf_bp_run_dev": (C.c_int, [_vp, _vp, _vp, _sz, _sz, C.c_int, C.c_int, _vp, _sz, _vp, _vp, _vp]),
    "bpmf_bp_run": (C.c_int, [_f, _i, _f, _f, _sz, _sz, _sz, _sz, _sz, C.c_int, C.c_int, C.c_int,
                              _f, _i]),
    "bpmf_bp_pack_max_dev": (C.c_int, [_vp, _vp, _sz, C.c_int, _vp, _vp]),
    "bpmf_bp_unpack_max_dev": (C.c_int, [_vp, _sz, C.c_int, _vp, _vp, _vp]),
    "bpmf_intertemplate_workspace_bytes": (_sz, [_sz, _sz, _sz, _sz]),
    "bpmf_intertemplate_cc_dev": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _vp, _sz, _vp, _vp]),
    "bpmf_bp_num_windows": (_sz, [_sz, _sz, _sz]),
    "bpmf_bp_window_stats_dev": (C.c_int, [_vp, _sz, _sz, _sz, _vp, _vp, _vp]),
    "bpmf_bp_extract_peaks_dev": (C.c_int, [_vp, _vp, _sz, C.c_double, C.c_uint32, _vp, _vp, _vp]),
    "bpmf_tdt_num_windows": (_sz, [_sz, _sz, _sz]),
    "bpmf_tdt_workspace_bytes": (_sz, [_sz, _sz, _sz, _sz]),
    "bpmf_tdt_rms_dev": (C.c_int, [_vp, _vp, C.c_float, _sz, _sz, _sz, _sz, _vp, _sz, _vp, _vp, _vp]),
    "bpmf_find_similar_sources": (C.c_int, [_f, _f, _f, _f, _f, C.c_float, _sz, _sz, _sz, _sz, _sz,
                                            C.c_int, C.c_int, _i]),
    "bpmf_extract_candidates_dev": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, C.c_uint32, _vp,
                                              _vp, _vp]),
    "bpmf_count_below_dev": (C.c_int, [_vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bpmf_extract_candidates_mad_dev": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, C.c_uint32, _vp,
                                                  _vp, _vp]),
    "bpmf_row_median_mad_dev": (C.c_int, [_vp, _sz, _sz, C.c_int, _vp, _vp, _vp, _vp]),
    "bpmf_row_median_mad_workspace_bytes": (_sz, [_sz, _sz]),
    "bpmf_row_median_mad_ws_dev": (C.c_int, [_vp, _sz, _sz, C.c_int, _vp, _sz, _vp, _vp, _vp, _vp]),
    "bpmf_saturate_rows_dev": (C.c_int, [_vp, _vp, _vp, _vp, _sz, _sz, C.c_float, _vp, _vp]),
    "bpmf_hilbert_spectrum_dev": (C.c_int, [_vp, _sz, _sz, C.c_int, _vp]),
    "bpmf_envelope_combine_dev": (C.c_int, [_vp, _vp, _sz, _vp, _vp]),
    "bpmf_tdt_mad_num_windows": (_sz, [_sz, _sz, _sz]),
    "bpmf_tdt_mad_workspace_bytes": (_sz, [_sz, _sz, _sz, _sz]),
    "bpmf_tdt_mad_dev": (C.c_int, [_vp, _vp, _sz, C.c_float, _sz, _sz, _sz, _sz, _vp, _sz, _vp, _vp, _vp, _vp]),
    "bpmf_row_kurtosis_workspace_bytes": (_sz, [_sz, _sz]),
    "bpmf_row_kurtosis_dev": (C.c_int, [_vp, _sz, _sz, _vp, _sz, _vp, _vp]),
    "bpmf_row_kurtosis_parts_dev": (C.c_int, [_vp, _sz, _sz, _vp, _sz, _vp, _vp, _vp]),
}

_lib = None


class BpmfHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the HIP library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise BpmfHipError(
            f"{LIBPATH} is missing: build it with `python -m seismic_bpmf_amd.build` "
            "(needs hipcc).  There is no CPU fallback in this package.")
    # torch ships its own libamdhip64 (same SONAME); importing it first makes the
    # dynamic loader bind this library to the runtime torch already uses, so that
    # torch streams / device pointers are valid in our launches.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the host-pointer API
        pass
    handle = C.CDLL(LIBPATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError here = header/library mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().bpmf_last_error().decode("utf-8", "replace")
        raise BpmfHipError(f"{what} failed (status {rc}): {msg}")


def last_error():
    """bpmf_last_error() of the calling thread: the text of the last failure -- or, behind a call that returned 0, a
    'note: ...' (e.g. the multi-device hand-over of the day fell back to host uploads)."""
    return lib().bpmf_last_error().decode("utf-8", "replace")


def host_call_stats():
    """Where the time of this thread's last host-pointer call went (bpmf_host_call_stats), milliseconds."""
    buf = (C.c_double * 10)()
    k = lib().bpmf_host_call_stats(buf, 10)
    names = ("total_ms", "first_kernel_start_ms", "host_copy_ms", "device_wait_ms", "pieces", "fill_threads",
             "pinned_wait_ms", "copy_enqueue_ms", "plan_ms", "reserve_ms")
    out = {n: float(buf[i]) for i, n in enumerate(names[:k])}
    for n in ("pieces", "fill_threads"):
        if n in out:
            out[n] = int(out[n])
    return out


def device_count():
    n = lib().bpmf_device_count()
    if n < 0:
        check(n, "bpmf_device_count")
    return n


def device_info(device=0):
    name = C.create_string_buffer(256)
    mem = _sz(0)
    cus = C.c_int(0)
    check(lib().bpmf_device_info(device, name, 256, C.byref(mem), C.byref(cus)), "bpmf_device_info")
    return {"name": name.value.decode(), "total_mem_bytes": mem.value, "compute_units": cus.value}


KERNEL_MF_MAIN, KERNEL_BP_BEAM = 0, 1


def set_option(name, value):
    """bpmf_set_option: choose among code paths with identical results (see include/bpmf_hip.h)."""
    check(lib().bpmf_set_option(name.encode(), int(value)), f"bpmf_set_option({name})")


def get_option(name):
    val, dflt = C.c_long(0), C.c_long(0)
    check(lib().bpmf_get_option(name.encode(), C.byref(val), C.byref(dflt)), f"bpmf_get_option({name})")
    return val.value, dflt.value


class options:
    """Context manager: set execution options, restore the previous values on exit.

        with _lib.options(**{"bp.fast": 0, "bp.split": 3}): ...
    """

    def __init__(self, **kv):
        self.kv = {k.replace("__", "."): v for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: get_option(k)[0] for k in self.kv}
        for k, v in self.kv.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def release_device_memory(device=-1):
    """Give back the working set and pinned pieces the host-pointer calls keep on `device` (-1: all)."""
    check(lib().bpmf_release_device_memory(int(device)), "bpmf_release_device_memory")


def device_memory_held(device=-1):
    """(device bytes, pinned host bytes) the host-pointer calls hold between calls."""
    dv, pn = _sz(0), _sz(0)
    check(lib().bpmf_device_memory_held(int(device), C.byref(dv), C.byref(pn)), "bpmf_device_memory_held")
    return dv.value, pn.value


def profile_enable(on=True):
    lib().bpmf_profile_enable(1 if on else 0)


def _closed_launches(which):
    """Indices of the logged launches of `which` whose stop edge was recorded, with their durations (ms).  A launch
    whose pair was opened but never closed (the call failed between the two edges) is skipped, not an error."""
    out = []
    for i in range(lib().bpmf_profile_count(which)):
        ms = C.c_float(0.0)
        if lib().bpmf_profile_get_ms(which, i, C.byref(ms)) == 0:
            out.append((i, ms.value))
    return out


def profile_devices(which):
    """The device of every closed launch of dominant kernel `which` (same order as profile_times_ms)."""
    return [lib().bpmf_profile_get_device(which, i) for i, _ in _closed_launches(which)]


def profile_times_ms(which):
    """Durations (ms) of every logged (and closed) launch of dominant kernel `which` since profile_enable."""
    return [ms for _, ms in _closed_launches(which)]


# ---- convention profiles (INTEGRATION.md sections A and F) ----
# The arithmetic of both hot paths lives in two third-party packages that are not in the reference tree; seven
# conventions rest on recollection and each has a switch.  A profile sets them all at once.
COMPAT_SWITCHES = ("mf.compat_exclusive_last_lag", "mf.compat_sqrt_norm", "mf.compat_range_all_channels",
                   "mf.compat_sequential_csum", "bp.compat_first_computed", "bp.compat_strict_upper_only",
                   "bp.compat_range_all_stations")
COMPAT_PROFILES = {
    # this build's own conventions (every switch off): what the oracle, the goldens and the headline numbers use
    "build": {},
    # what SURVEY.md Appendix A recollects of upstream's C (UNVERIFIED -- nothing in this image can confirm it;
    # tools/diff_upstream.py is the check on a machine that has the packages): the lag loop stops BEFORE
    # N - L - mv_max, moveout range over ALL channels, cc = num / sqrt(E_t * E_d) above 1e-6, one sequential
    # double prefix sum; the beamformer's max scan starts from the first computed beam
    "upstream-recollected": {"mf.compat_exclusive_last_lag": 1, "mf.compat_sqrt_norm": 1,
                             "mf.compat_range_all_channels": 1, "mf.compat_sequential_csum": 1,
                             "bp.compat_first_computed": 1},
}


def compat_profile(name=None):
    """Select a set of conventions for BOTH hot paths, process-wide (bpmf_set_option on all seven `*.compat_*`
    switches): "build" (the default: this build's own conventions) or "upstream-recollected" (the five
    alternatives SURVEY.md Appendix A recollects of fast_matched_filter / beampower; costs x 1.14 at cfg2 -- the
    IEEE square root and divide per channel and lag -- and ~50 ms per day for the sequential prefix sum).
    Without an argument: the name of the profile the current switches amount to, or None.  The import shims
    (shims/fast_matched_filter, shims/beampower) run whatever is selected here."""
    if name is None:
        now = {n: get_option(n)[0] for n in COMPAT_SWITCHES}
        for prof, on in COMPAT_PROFILES.items():
            if all(now[n] == on.get(n, 0) for n in COMPAT_SWITCHES):
                return prof
        return None
    if name not in COMPAT_PROFILES:
        raise ValueError(f"unknown profile {name!r}: one of {sorted(COMPAT_PROFILES)}")
    for n in COMPAT_SWITCHES:
        set_option(n, COMPAT_PROFILES[name].get(n, 0))
    return name


def profile_of_switches(names_on):
    """The profile whose switched-on options are exactly `names_on` (an iterable of option names), or None."""
    on = set(names_on)
    for prof, sw in COMPAT_PROFILES.items():
        if on == {n for n, v in sw.items() if v}:
            return prof
    return None
