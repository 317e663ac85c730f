// Error reporting and device diagnostics of libbpmf_hip.so.
#include "common.h"
#include "../../include/bpmf_hip.h"

#include <atomic>
#include <cstring>
#include <mutex>

namespace bpmf {

char* last_error_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
}

}  // namespace bpmf

// ---- execution options (see common.h: identical results whatever their values) ----
namespace bpmf {
namespace {
struct OptionDef { const char* name; long dflt; long lo, hi; };
const OptionDef OPTION_DEFS[OPT_COUNT] = {
    {"bp.lds_kb", 80, 8, 160},          // soft LDS budget of a group (single-window plans)
    {"bp.max_group", 4096, 1, 1 << 20}, // sources per group at most
    {"bp.tpt", 2, 1, 4},                // samples per thread of the generic kernels (tile = 256 x tpt)
    {"bp.reorder", 1, 0, 1},            // kd-tree processing order of the sources
    {"bp.dual", 1, 0, 1},               // dual (shifted) windows + 8-byte gathers where they fit
    {"bp.packed", 1, 0, 1},             // packed per-station records (P = 2)
    {"bp.wps", 1, 0, 1},                // wave-per-source kernels
    {"bp.uvgpr", 1, 0, 1},              // uniform-VGPR metadata kernels
    {"bp.fast", 1, 0, 1},               // interior-tile kernel of bp_fast.hip
    {"bp.fast_uniform", 1, 0, 1},       // ready-made addresses when a source's weights are uniform
    {"bp.split", -1, -1, 1 << 16},      // group ranges per tile: -1 = automatic (short series)
    {"bp.wpb", 12, 8, 12},              // waves per workgroup of the 4-byte-gather kernel
    {"bp.smeta", 1, 0, 1},              // SGPR metadata for 32-station records
    {"bp.verbose", 0, 0, 1},
    {"bp.fast_tile", 0, 0, 512},        // 0: the cost model picks each class's tile; 512 / 256 / 128: only that one
    {"bp.halves", 1, 0, 1},             // 33-64 stations: two LDS residencies per group at tile 256 where cheaper
    {"bp.direct", 0, 0, 1},             // 1: every plan takes the global-memory path of bp_direct.hip (tests)
    {"mf.wave_kernel", 1, 0, 1},        // independent-wave kernel for L <= 257
    {"mf.max_mfma_step", 64, 0, 1 << 20},  // larger steps take the generic kernel
    {"mf.host_batch_kb", 0, 0, 1L << 30},  // host-pointer call: output per batch (0 = 1 GB, >= 8 templates)
    {"mf.host_piece_kb", 0, 0, 1L << 30},  // host-pointer call: pinned piece (0 = 64 MB)
    {"mf.verbose", 0, 0, 1},
    {"mf.tiles_per_wave", 0, 0, 4},     // 16x16 tiles per wave of the L <= 257 kernel: 0 = by problem size, 1 / 2 / 4
    {"mf.compat_exclusive_last_lag", 0, 0, 1},  // last valid data offset i * step < N - L - mv_max (default: <=)
    {"mf.compat_sqrt_norm", 0, 0, 1},           // cc = num / sqrtf(E_t * E_d) above 1e-6 (generic kernel; default: num * r_t * r_d)
    {"bp.compat_first_computed", 0, 0, 1},      // running max starts from the first computed beam (default: from (0, source 0))
};
std::atomic<long> g_options[OPT_COUNT];
std::once_flag g_options_once;
void init_options()
{
    for (int i = 0; i < OPT_COUNT; ++i) g_options[i].store(OPTION_DEFS[i].dflt, std::memory_order_relaxed);
}
int find_option(const char* name)
{
    if (!name) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(name, OPTION_DEFS[i].name) == 0) return i;
    return -1;
}
}  // namespace

std::atomic<unsigned long long> g_option_generation{0};
unsigned long long option_generation() { return g_option_generation.load(std::memory_order_relaxed); }

long option(Option which)
{
    std::call_once(g_options_once, init_options);
    return g_options[which].load(std::memory_order_relaxed);
}
}  // namespace bpmf

extern "C" int bpmf_set_option(const char* name, long value)
{
    std::call_once(bpmf::g_options_once, bpmf::init_options);
    const int i = bpmf::find_option(name);
    if (i < 0) {
        bpmf::set_error("bpmf_set_option: unknown option '%s'", name ? name : "(null)");
        return -1;
    }
    const bpmf::OptionDef& d = bpmf::OPTION_DEFS[i];
    if (value < d.lo || value > d.hi) {
        bpmf::set_error("bpmf_set_option: %s = %ld outside [%ld, %ld]", d.name, value, d.lo, d.hi);
        return -1;
    }
    if (bpmf::g_options[i].exchange(value, std::memory_order_relaxed) != value)
        bpmf::g_option_generation.fetch_add(1, std::memory_order_relaxed);
    return 0;
}

extern "C" int bpmf_get_option(const char* name, long* value, long* default_value)
{
    std::call_once(bpmf::g_options_once, bpmf::init_options);
    const int i = bpmf::find_option(name);
    if (i < 0) {
        bpmf::set_error("bpmf_get_option: unknown option '%s'", name ? name : "(null)");
        return -1;
    }
    if (value) *value = bpmf::g_options[i].load(std::memory_order_relaxed);
    if (default_value) *default_value = bpmf::OPTION_DEFS[i].dflt;
    return 0;
}

// ---- optional per-kernel timing (bench.py's roofline leg) -------------------------
// Every launch of a dominant kernel gets its own start/stop event pair, recorded on the
// launch stream without synchronising; the pairs are read back after the timed region.
namespace bpmf {
constexpr int PROFILE_MAX_LAUNCHES = 512;
struct ProfileLog {
    hipEvent_t ev[PROFILE_MAX_LAUNCHES][2];
    int created = 0;  // event pairs that exist
    int count = 0;    // launches recorded since the last reset
    bool open = false;
};
// The log is shared by every host thread that launches (the multi-device entry points run one
// thread per GPU): all accesses go through g_profile_mutex.  The enable flag is read without the
// lock on the fast path (a launch with profiling off takes no lock at all).
static volatile bool g_profile = false;
static ProfileLog g_log[BPMF_KERNEL_COUNT];
static std::mutex g_profile_mutex;

void profile_mark(int which, int edge, hipStream_t stream)
{
    if (!g_profile || which < 0 || which >= BPMF_KERNEL_COUNT) return;
    std::lock_guard<std::mutex> lock(g_profile_mutex);
    ProfileLog& lg = g_log[which];
    if (edge == 0) {
        lg.open = false;
        if (lg.count >= PROFILE_MAX_LAUNCHES) return;
        if (lg.count >= lg.created) {
            if (hipEventCreate(&lg.ev[lg.created][0]) != hipSuccess) return;
            if (hipEventCreate(&lg.ev[lg.created][1]) != hipSuccess) return;
            lg.created++;
        }
        lg.open = hipEventRecord(lg.ev[lg.count][0], stream) == hipSuccess;
    } else if (lg.open) {
        if (hipEventRecord(lg.ev[lg.count][1], stream) == hipSuccess) lg.count++;
        lg.open = false;
    }
}
}  // namespace bpmf

extern "C" void bpmf_profile_enable(int enable)
{
    std::lock_guard<std::mutex> lock(bpmf::g_profile_mutex);
    bpmf::g_profile = enable != 0;
    if (!enable) return;  // the log stays readable after the timed region is closed
    for (int k = 0; k < BPMF_KERNEL_COUNT; ++k) { bpmf::g_log[k].count = 0; bpmf::g_log[k].open = false; }
}

extern "C" int bpmf_profile_count(int which)
{
    if (which < 0 || which >= BPMF_KERNEL_COUNT) return -1;
    std::lock_guard<std::mutex> lock(bpmf::g_profile_mutex);
    return bpmf::g_log[which].count;
}

extern "C" int bpmf_profile_get_ms(int which, int index, float* ms)
{
    std::lock_guard<std::mutex> lock(bpmf::g_profile_mutex);
    if (which < 0 || which >= BPMF_KERNEL_COUNT || !ms || index < 0 ||
        index >= bpmf::g_log[which].count) {
        bpmf::set_error("bpmf_profile_get_ms: no launch %d recorded for kernel %d", index, which);
        return -1;
    }
    BPMF_HIP_CHECK(hipEventSynchronize(bpmf::g_log[which].ev[index][1]));
    BPMF_HIP_CHECK(hipEventElapsedTime(ms, bpmf::g_log[which].ev[index][0],
                                       bpmf::g_log[which].ev[index][1]));
    return 0;
}

extern "C" const char* bpmf_last_error(void) { return bpmf::last_error_buf(); }

// HIP analogue of the reference's device probe (BPMF/GPU.cu:8-24).
extern "C" int bpmf_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        bpmf::set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return -2;
    }
    return n;
}

extern "C" int bpmf_device_info(int device, char* name, size_t name_len, size_t* total_mem_bytes,
                                int* compute_units)
{
    hipDeviceProp_t prop;
    BPMF_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (name && name_len) {
        strncpy(name, prop.name, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (total_mem_bytes) *total_mem_bytes = prop.totalGlobalMem;
    if (compute_units) *compute_units = prop.multiProcessorCount;
    return 0;
}

// ------------------------------------------------------------- peak suppression (host) ---
// BPMF/utils.py:2334-2345: for every peak, tallest first, unless already removed:
//   idel = idel | (ind >= ind[i] - mpd) & (ind <= ind[i] + mpd); idel[i] = 0
// `positions` is ascending, so the peaks within +-mpd of one are its neighbours in the array.
extern "C" int bpmf_suppress_peaks(const int64_t* positions, const int64_t* order, size_t n,
                                   double mpd, uint8_t* keep)
{
    if ((!positions || !order || !keep) && n) {
        bpmf::set_error("bpmf_suppress_peaks: bad argument");
        return -1;
    }
    for (size_t i = 0; i < n; ++i) keep[i] = 1;
    for (size_t q = 0; q < n; ++q) {
        const int64_t i = order[q];
        if (i < 0 || (size_t)i >= n) {
            bpmf::set_error("bpmf_suppress_peaks: order[%zu] = %lld out of range", q, (long long)i);
            return -1;
        }
        if (!keep[i]) continue;
        const double lo = (double)positions[i] - mpd, hi = (double)positions[i] + mpd;
        for (int64_t j = i - 1; j >= 0 && (double)positions[j] >= lo; --j) keep[j] = 0;
        for (size_t j = (size_t)i + 1; j < n && (double)positions[j] <= hi; ++j) keep[j] = 0;
    }
    return 0;
}
