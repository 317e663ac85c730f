// Shared host-side helpers of libbpmf_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

namespace bpmf {

// Thread-local last error text returned by bpmf_last_error().
char* last_error_buf();
void set_error(const char* fmt, ...);

constexpr int CSUM_CHUNK = 1024;             // spec constant, see oracle/bpmf_oracle.c
constexpr float MAX_NORM = 1000.0f;          // r_t * r_d >= this (E_t*E_d <~ 1e-6) -> CC = 0

// bench hook: events around the dominant kernels (util.hip)
void profile_mark(int which, int edge, hipStream_t stream);

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Execution options (util.hip, C ABI: bpmf_set_option / bpmf_get_option).  Every option selects
// among code paths and sizes that produce IDENTICAL results -- kernel family, LDS budget, staging
// batch sizes, group ranges per tile; none of them can change an output bit, and the library reads
// nothing from the environment.  The exceptions: option mf.split16 (off by default; matched-filter numerators on the
// fp16 matrix pipe from hi/lo splits: results within 3e-7 of the exact path instead of bit-identical, mf_split.h), and
// the `*.compat_*` switches at the end (off
// by default): each replaces one convention of this build that rests on recollection only by the
// alternative the upstream packages may implement.  The defaults are the tuned production values; the GPU tests use
// the options to force every kernel family through the same parity cases.
enum Option {
    OPT_BP_LDS_KB, OPT_BP_MAX_GROUP, OPT_BP_TPT, OPT_BP_REORDER, OPT_BP_DUAL, OPT_BP_PACKED,
    OPT_BP_WPS, OPT_BP_UVGPR, OPT_BP_FAST, OPT_BP_FAST_UNIFORM, OPT_BP_SPLIT, OPT_BP_WPB,
    OPT_BP_SMETA, OPT_BP_VERBOSE, OPT_BP_FAST_TILE, OPT_BP_HALVES, OPT_BP_DIRECT, OPT_MF_WAVE_KERNEL, OPT_MF_MAX_MFMA_STEP, OPT_MF_HOST_BATCH_KB,
    OPT_MF_HOST_PIECE_KB, OPT_MF_VERBOSE, OPT_MF_TILES_PER_WAVE, OPT_MF_BOUNDARY_PRIO, OPT_MF_FUSED_PROLOGUE, OPT_DEBUG_POISON_OUTPUT,
    OPT_DEBUG_VIRTUAL_DEVICES, OPT_MULTI_PEER_FANOUT, OPT_MF_HOST_PIECE_LAGS, OPT_BP_HOST_PIECE_SAMPLES,
    OPT_HOST_CACHE_LIMIT_MB, OPT_BP_SLOT_PRIO, OPT_MF_CHANNEL_SPLIT, OPT_STATS_BUCKETED_MEDIAN, OPT_STATS_ROW_GRID_MIN_N, OPT_STATS_KURT_FULL_CHUNKS, OPT_MF_SPLIT16, OPT_DEBUG_FAIL_PEER_COPY,
    // upstream-compatibility switches: the ONLY options that change results (off by default)
    OPT_MF_COMPAT_EXCLUSIVE_LAST_LAG, OPT_MF_COMPAT_SQRT_NORM, OPT_BP_COMPAT_FIRST_COMPUTED,
    OPT_MF_COMPAT_RANGE_ALL_CHANNELS, OPT_MF_COMPAT_SEQUENTIAL_CSUM, OPT_BP_COMPAT_STRICT_UPPER_ONLY,
    OPT_BP_COMPAT_RANGE_ALL_STATIONS, OPT_COUNT
};
long option(Option which);
// counts the option changes since the library was loaded: cached plans are keyed by it (a plan is
// built under the options of its creation)
unsigned long long option_generation();

// Logical and physical devices (util.hip).  Every `device` argument of the C ABI is a LOGICAL device.
// Normally logical == physical.  Option debug.virtual_devices = k > 0 (tests) makes the library see k
// logical devices 0 .. k-1, logical d living on physical GPU d % (GPUs visible): each logical device has
// its own DeviceContext (streams, working set, call mutex) and, in the *_run_multi entry points, its own
// host thread -- which is how the multi-device branches (threads, peer copies of the day of data, the
// host merge of the beam maxima) run under `pytest -m gpu` on a box with ONE GPU.
hipError_t device_counts(int* n_logical, int* n_physical);
int physical_device(int logical);
// the logical device the calling thread is bound to by its innermost DeviceGuard (-1: none)
extern thread_local int t_logical_device;

// Binds the calling thread to (the physical GPU of) logical `device` for the lifetime of the guard and
// restores the device that was current before: a host entry point must not change the caller's (or
// torch's) current device as a side effect.  error() is not hipSuccess if either runtime call failed.
class DeviceGuard {
public:
    explicit DeviceGuard(int device) : DeviceGuard(device, physical_device(device)) {}
    // (a context remembers the physical GPU it was created on: valid whatever the option says now)
    DeviceGuard(int logical, int physical)
    {
        prev_logical_ = t_logical_device;
        t_logical_device = logical;
        err_ = hipGetDevice(&prev_);
        if (err_ == hipSuccess && prev_ != physical) {
            err_ = hipSetDevice(physical);
            switched_ = err_ == hipSuccess;
        }
    }
    ~DeviceGuard()
    {
        if (switched_) (void)hipSetDevice(prev_);
        t_logical_device = prev_logical_;
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
    hipError_t error() const { return err_; }
private:
    int prev_ = 0, prev_logical_ = -1;
    bool switched_ = false;
    hipError_t err_ = hipSuccess;
};

}  // namespace bpmf

#define BPMF_HIP_CHECK(expr)                                                          \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            bpmf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                            __FILE__, __LINE__);                                      \
            return -2;                                                                \
        }                                                                             \
    } while (0)

#define BPMF_LAUNCH_CHECK() BPMF_HIP_CHECK(hipGetLastError())

#define BPMF_BIND_DEVICE(device)                 \
    bpmf::DeviceGuard _device_guard(device);     \
    BPMF_HIP_CHECK(_device_guard.error())
