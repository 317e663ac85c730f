// Error reporting and device diagnostics of libbpmf_hip.so.
#include "common.h"
#include "../../include/bpmf_hip.h"

#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

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
// plan: bpmf_bp_plan_create reads it (a change starts a new generation of cached plans)
struct OptionDef { const char* name; long dflt; long lo, hi; bool plan; };
const OptionDef OPTION_DEFS[OPT_COUNT] = {
    {"bp.lds_kb", 80, 8, 160, true},          // soft LDS budget of a group (single-window plans)
    {"bp.max_group", 4096, 1, 1 << 20, true}, // sources per group at most
    {"bp.tpt", 2, 1, 4, true},                // samples per thread of the generic kernels (tile = 256 x tpt)
    {"bp.reorder", 1, 0, 1, true},            // kd-tree processing order of the sources
    {"bp.dual", 1, 0, 1, true},               // dual (shifted) windows + 8-byte gathers where they fit
    {"bp.packed", 1, 0, 1, true},             // packed per-station records (P = 2)
    {"bp.wps", 1, 0, 1, true},                // wave-per-source kernels
    {"bp.uvgpr", 1, 0, 1, true},              // uniform-VGPR metadata kernels
    {"bp.fast", 1, 0, 1, true},               // interior-tile kernel of bp_fast.hip
    {"bp.fast_uniform", 1, 0, 1, true},       // ready-made addresses when a source's weights are uniform
    {"bp.split", -1, -1, 1 << 16, false},      // group ranges per tile: -1 = automatic (short series)
    {"bp.wpb", 12, 8, 12, false},              // waves per workgroup of the 4-byte-gather kernel
    {"bp.smeta", 1, 0, 1, false},              // SGPR metadata for 32-station records
    {"bp.verbose", 0, 0, 1, false},
    {"bp.fast_tile", 0, 0, 512, true},        // 0: the cost model picks each class's tile; 512 / 256 / 128: only that one
    {"bp.halves", 1, 0, 1, true},             // 33-64 stations: two LDS residencies per group at tile 256 where cheaper
    {"bp.direct", 0, 0, 1, true},             // 1: every plan takes the global-memory path of bp_direct.hip (tests)
    {"mf.wave_kernel", 1, 0, 1, false},        // independent-wave kernel for L <= 257
    {"mf.max_mfma_step", 64, 0, 1 << 20, false},  // larger steps take the generic kernel
    {"mf.host_batch_kb", 0, 0, 1L << 30, false},  // host-pointer call: output per batch (0 = 1 GB, >= 8 templates)
    {"mf.host_piece_kb", 0, 0, 1L << 30, false},  // host-pointer call: pinned piece (0 = 64 MB)
    {"mf.verbose", 0, 0, 1, false},
    {"mf.tiles_per_wave", 0, 0, 4, false},     // 16x16 tiles per wave of the L <= 257 kernel: 0 = by problem size, 1 / 2 / 4
    {"mf.boundary_prio", 1, 0, 3, false},      // issue priority (s_setprio) of a wave of the L <= 257 kernel while it is outside its K loop (0: none; 1 measured +1.0-1.3 % at every L, 2 / 3 the same)
    {"mf.fused_prologue", 1, 0, 1, false},     // small problems (fewer than 4 tiles per wave, <= 256 channels): template norms, channel records and lag range computed by every workgroup of the L <= 257 kernel itself -- one launch per call instead of two
    {"debug.poison_output", 0, 0, 1, false},   // tests: fill the (max-beam, arg-max) / CC-sum output with 0xFF bytes (NaN / -1) before the kernels run -- a sample no kernel writes then shows
    {"debug.virtual_devices", 0, 0, 64, false}, // tests: k > 0 = the library sees k logical devices, logical d on physical GPU d % visible, each with its own context and host thread (common.h)
    {"multi.peer_fanout", 1, 0, 1, false},     // *_run_multi on several devices: the day of data goes to the first device once (host -> device) and from there to the others device -> device (hipMemcpyPeerAsync); 0 = every device uploads from the host itself
    {"mf.host_piece_lags", 131072, 0, 1 << 24, false},     // bpmf_mf_run: the day of data arrives in pieces (first piece this many samples, rounded to 4096; doubling to 8x) while the first two template batches run on the lags that have arrived; 0 = one upload in front of the first kernel
    {"bp.host_piece_samples", 131072, 0, 1 << 24, false},  // bpmf_bp_run: the same for the day of features (first piece this many samples, rounded to 131072 = one round of the chip; then 2x, 4x); 0 = one upload in front
    {"host.cache_limit_mb", 0, 0, 1L << 30, false},        // host-pointer calls: a device working set larger than this many MB is given back when the call ends (0 = kept for the next call, bpmf_release_device_memory frees it)
    {"bp.slot_prio", 1, 0, 2, false},                      // interior-tile kernels: a wave lowers its issue priority (s_setprio 3, 2, 1, 0 by quarters) as it gets through its sources of a group, so that the 16 waves reach the group's barrier together; 0: none (rounds 2-4)
    {"mf.channel_split", 2048, 0, 1 << 24, false},         // tiny matched-filter problems (at most this many waves of 256 lags; one tile per wave, fused prologue, network sum, step 1, <= 32 channels): four waves per 256 lags, every fourth used channel each, the channel sum behind one barrier (0 = off)
    {"stats.bucketed_median", 2, 0, 2, false},             // MAD threshold, window medians: 2 = one pass each (the elements of a narrow band around the row's centre / around the middle of a side histogram of deviations, ranked in LDS), 1 = two passes (equal-width buckets, the middle bucket ranked in LDS; also the route when a band misses), 0 = three-pass radix select only (the last resort of the others)
    {"stats.row_grid_min_n", 131072, -1, 1 << 30, false},   // row median / MAD: rows at least this long are read twice by workgroups from all over the chip (histogram, then the middle bucket and two bands; exact, verified by ranks) instead of seven times by one workgroup; -1 = never
    {"stats.kurt_full_chunks", 1, 0, 1, false},            // row kurtosis: a full 8192-sample chunk of NumPy's summation is summed by one workgroup through LDS (coalesced reads); 0 = the thread-per-leaf kernels for every chunk
    {"mf.split16", 0, 0, 2, false},                        // matched filter (every template length the MFMA kernels take): numerators from fp16 hi/lo splits of data and templates (three v_mfma_f32_32x32x16_f16 products, fp32 accumulation; csrc/mf_split.h) instead of the exact-fp32 MFMA chain: |d cc| ~ 2e-7 instead of bit-identity with the oracle, ~2.4x the rate.  Norms, lag ranges, zero rules unchanged.  OFF by default; 1 = launches of at least 128 (template, 8192-lag block) pairs, 2 = every launch; changes the workspace size and what a prepared day holds
    {"debug.fail_peer_copy", 0, 0, 1, false},              // tests: every device-to-device probe of the multi-device hand-over fails (context.hip: peer_copy_works), so the fall-back -- each device uploads the day from the host, with a note in bpmf_last_error -- runs on a one-GPU box
    {"mf.compat_exclusive_last_lag", 0, 0, 1, false},  // last valid data offset i * step < N - L - mv_max (default: <=)
    {"mf.compat_sqrt_norm", 0, 0, 1, false},           // cc = num / sqrtf(E_t * E_d) above 1e-6 (default: num * r_t * r_d)
    {"bp.compat_first_computed", 0, 0, 1, false},      // running max starts from the first computed beam (default: from (0, source 0))
    {"mf.compat_range_all_channels", 0, 0, 1, false},  // a template's valid lag range over ALL its channels (default: the weighted ones)
    {"mf.compat_sequential_csum", 0, 0, 1, false},     // double prefix sum of data^2 as ONE sequential chain per channel (default: 1024-sample hierarchy)
    {"bp.compat_strict_upper_only", 0, 0, 1, true},    // strict = t + tau_max < N only; used terms in front of sample 0 dropped (default: also t + tau_min >= 0)
    {"bp.compat_range_all_stations", 0, 0, 1, true},   // a source's tau_min / tau_max over ALL its stations (default: the weighted ones)
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

thread_local int t_logical_device = -1;

hipError_t device_counts(int* n_logical, int* n_physical)
{
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    const long v = option(OPT_DEBUG_VIRTUAL_DEVICES);
    if (n_physical) *n_physical = n;
    if (n_logical) *n_logical = (v > 0 && n > 0) ? (int)v : n;
    return e;
}

int physical_device(int logical)
{
    if (logical < 0 || option(OPT_DEBUG_VIRTUAL_DEVICES) <= 0) return logical;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return logical;
    return logical % n;
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
    // cached plans are keyed by the generation: only the options a plan is BUILT under count (a
    // verbosity or matched-filter setting must not orphan every resident backprojection plan)
    if (bpmf::g_options[i].exchange(value, std::memory_order_relaxed) != value && bpmf::OPTION_DEFS[i].plan)
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
//
// The multi-device entry points launch the same kernel from several host threads, one per GPU:
// a pair belongs to the THREAD that opened it (thread-local slot index) and to the DEVICE that was
// current when it was opened (an event can only be recorded on a stream of the device it was
// created on), so edges of two threads never pair up and events never cross devices.
namespace bpmf {
constexpr int PROFILE_MAX_LAUNCHES = 512;
struct ProfilePair {
    hipEvent_t ev[2] = {nullptr, nullptr};
    int device = -1;          // physical GPU the events live on
    int logical = -1;         // logical device of the launch (common.h), what bpmf_profile_get_device reports
    bool closed = false;      // stop edge recorded
};
struct ProfileLog {
    std::vector<ProfilePair> rec;      // launches since the last reset, in the order they were opened
    std::vector<ProfilePair> spare;    // pairs of earlier sessions, reused by device
};
static std::atomic<bool> g_profile{false};
static std::atomic<unsigned> g_profile_session{0};
static ProfileLog g_log[BPMF_KERNEL_COUNT];
static std::mutex g_profile_mutex;
// the pair this thread has open, per kernel kind: (session, index into rec), -1 = none
static thread_local int t_open_slot[BPMF_KERNEL_COUNT] = {-1, -1};
static thread_local unsigned t_open_session[BPMF_KERNEL_COUNT] = {0, 0};

void profile_mark(int which, int edge, hipStream_t stream)
{
    if (!g_profile.load(std::memory_order_relaxed) || which < 0 || which >= BPMF_KERNEL_COUNT) return;
    std::lock_guard<std::mutex> lock(g_profile_mutex);
    ProfileLog& lg = g_log[which];
    const unsigned session = g_profile_session.load(std::memory_order_relaxed);
    if (edge == 0) {
        t_open_slot[which] = -1;
        if (lg.rec.size() >= (size_t)PROFILE_MAX_LAUNCHES) return;
        int device = -1;
        if (hipGetDevice(&device) != hipSuccess) return;
        ProfilePair pr;
        for (size_t i = 0; i < lg.spare.size(); ++i)
            if (lg.spare[i].device == device) {
                pr = lg.spare[i];
                lg.spare.erase(lg.spare.begin() + (long)i);
                break;
            }
        if (!pr.ev[0]) {
            if (hipEventCreate(&pr.ev[0]) != hipSuccess) return;
            if (hipEventCreate(&pr.ev[1]) != hipSuccess) { (void)hipEventDestroy(pr.ev[0]); return; }
            pr.device = device;
        }
        pr.closed = false;
        pr.logical = t_logical_device >= 0 ? t_logical_device : device;
        if (hipEventRecord(pr.ev[0], stream) != hipSuccess) { lg.spare.push_back(pr); return; }
        lg.rec.push_back(pr);
        t_open_slot[which] = (int)lg.rec.size() - 1;
        t_open_session[which] = session;
    } else {
        const int slot = t_open_slot[which];
        t_open_slot[which] = -1;
        if (slot < 0 || t_open_session[which] != session || (size_t)slot >= lg.rec.size()) return;
        lg.rec[slot].closed = hipEventRecord(lg.rec[slot].ev[1], stream) == hipSuccess;
    }
}
}  // namespace bpmf

extern "C" void bpmf_profile_enable(int enable)
{
    std::lock_guard<std::mutex> lock(bpmf::g_profile_mutex);
    bpmf::g_profile.store(enable != 0);
    if (!enable) return;  // the log stays readable after the timed region is closed
    bpmf::g_profile_session.fetch_add(1);
    for (int k = 0; k < BPMF_KERNEL_COUNT; ++k) {
        bpmf::ProfileLog& lg = bpmf::g_log[k];
        for (auto& pr : lg.rec) lg.spare.push_back(pr);
        lg.rec.clear();
    }
}

extern "C" int bpmf_profile_count(int which)
{
    if (which < 0 || which >= BPMF_KERNEL_COUNT) return -1;
    std::lock_guard<std::mutex> lock(bpmf::g_profile_mutex);
    return (int)bpmf::g_log[which].rec.size();
}

extern "C" int bpmf_profile_get_ms(int which, int index, float* ms)
{
    std::lock_guard<std::mutex> lock(bpmf::g_profile_mutex);
    if (which < 0 || which >= BPMF_KERNEL_COUNT || !ms || index < 0 ||
        (size_t)index >= bpmf::g_log[which].rec.size()) {
        bpmf::set_error("bpmf_profile_get_ms: no launch %d recorded for kernel %d", index, which);
        return -1;
    }
    const bpmf::ProfilePair& pr = bpmf::g_log[which].rec[index];
    if (!pr.closed) {
        bpmf::set_error("bpmf_profile_get_ms: launch %d of kernel %d has no stop edge", index, which);
        return -1;
    }
    BPMF_HIP_CHECK(hipEventSynchronize(pr.ev[1]));
    BPMF_HIP_CHECK(hipEventElapsedTime(ms, pr.ev[0], pr.ev[1]));
    return 0;
}

// the device launch `index` of kernel `which` ran on (-1: no such launch)
extern "C" int bpmf_profile_get_device(int which, int index)
{
    std::lock_guard<std::mutex> lock(bpmf::g_profile_mutex);
    if (which < 0 || which >= BPMF_KERNEL_COUNT || index < 0 ||
        (size_t)index >= bpmf::g_log[which].rec.size())
        return -1;
    return bpmf::g_log[which].rec[index].logical;
}

extern "C" const char* bpmf_last_error(void) { return bpmf::last_error_buf(); }

// HIP analogue of the reference's device probe (BPMF/GPU.cu:8-24).
extern "C" int bpmf_device_count(void)
{
    int n = 0;
    hipError_t e = bpmf::device_counts(&n, nullptr);      // logical devices (option debug.virtual_devices)
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
    BPMF_HIP_CHECK(hipGetDeviceProperties(&prop, bpmf::physical_device(device)));
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
