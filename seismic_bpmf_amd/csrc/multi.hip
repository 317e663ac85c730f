// Multi-device host entry points: one call, host pointers in / host pointers out, the work
// block-partitioned over several GPUs of the node from inside the library (one host thread per
// DISTINCT device; a device listed n times computes n blocks one after the other) -- what a ctypes binding of the reference's third-party back-ends would call with
// `n_devices`.  Upstream's GPU back-ends split the same way inside one
// process: templates (matched filter) or sources (backprojection) dealt to the devices, the whole
// day of data copied to each.
//
//   matched filter  templates are independent: device d computes rows [t_d, t_{d+1}) of the CC
//                   matrix straight into the caller's array; nothing crosses devices.
//   backprojection  sources are independent up to the per-sample max / arg-max: device d scans
//                   the sources [k_d, k_{d+1}); the per-device maxima are merged on the host in
//                   ascending block order with a strict >, so that ties keep the lowest source
//                   index exactly like one sequential scan (oracle/bpmf_oracle.c:bp_cpu).
//                   reduce="none": device d writes rows [k_d, k_{d+1}) of the (K, N) beam.
#include "common.h"
#include "../../include/bpmf_hip.h"
#include "bp_plan.h"
#include "context.h"

#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

namespace {

using bpmf::set_error;

// Device list of a call: `devices` (n_devices entries) or, when it is NULL, 0 .. n_devices-1;
// n_devices <= 0 = every visible device.  Never more devices than work items.
int resolve_devices(int n_devices, const int* devices, size_t n_items, std::vector<int>& out,
                    const char* who)
{
    int visible = 0;                     // logical devices (option debug.virtual_devices, common.h)
    hipError_t e = bpmf::device_counts(&visible, nullptr);
    if (e != hipSuccess || visible < 1) {
        set_error("%s: no HIP device visible (%s)", who, hipGetErrorString(e));
        return -2;
    }
    if (n_devices <= 0) { n_devices = visible; devices = nullptr; }
    out.clear();
    for (int i = 0; i < n_devices; ++i) {
        const int d = devices ? devices[i] : i;
        if (d < 0 || d >= visible) {
            set_error("%s: device %d out of range (%d visible)", who, d, visible);
            return -1;
        }
        out.push_back(d);
    }
    if (out.size() > n_items) out.resize(std::max<size_t>(1, n_items));
    return 0;
}

// Contiguous partition of the templates into `parts` blocks of balanced COST: a template costs its
// number of channels with a non-zero weight (the kernels skip the others), and block r ends at the
// first template where the running cost reaches (r + 1) / parts of the total, and no block is empty
// while T >= parts -- the rule of seismic_bpmf_amd.parallel.shard_bounds_weighted
// (np.searchsorted(cumsum, total * r / parts, side="left") on float64 sums), so that the library's own multi-device split and the
// one-process-per-GPU split of torch.distributed agree on who computes which template.
std::vector<size_t> weighted_bounds(const float* weights, size_t T, size_t n_ch, size_t parts)
{
    std::vector<double> cum(T + 1, 0.0);
    for (size_t t = 0; t < T; ++t) {
        size_t used = 0;
        for (size_t c = 0; c < n_ch; ++c) used += weights[t * n_ch + c] != 0.0f;
        cum[t + 1] = cum[t] + (double)used;
    }
    std::vector<size_t> b(parts + 1, 0);
    const double total = cum[T];
    if (total <= 0.0) {          // no weighted channel anywhere: plain balanced blocks
        const size_t base = T / parts, rem = T % parts;
        for (size_t i = 0; i < parts; ++i) b[i + 1] = b[i] + base + (i < rem ? 1 : 0);
        return b;
    }
    for (size_t r = 1; r < parts; ++r) {
        const double target = total * (double)r / (double)parts;
        size_t k = (size_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
        b[r] = std::min(std::max(k, b[r - 1]), T);
    }
    b[parts] = T;
    // no empty block while there are at least as many templates as blocks (one heavy template must
    // not leave a device idle): cuts pushed apart forwards, then pulled back from the end
    if (T >= parts) {
        for (size_t r = 1; r < parts; ++r) b[r] = std::max(b[r], b[r - 1] + 1);
        for (size_t r = parts - 1; r >= 1; --r) b[r] = std::min(b[r], T - (parts - r));
    }
    return b;
}

// balanced contiguous partition of range(n) into `parts` blocks
std::vector<size_t> block_bounds(size_t n, size_t parts)
{
    std::vector<size_t> b(parts + 1, 0);
    const size_t base = n / parts, rem = n % parts;
    for (size_t i = 0; i < parts; ++i) b[i + 1] = b[i] + base + (i < rem ? 1 : 0);
    return b;
}

// Run fn(block) for every block: ONE host thread per DISTINCT device, the blocks of a device one
// after the other on its thread (a device listed several times gets several blocks, not several
// threads -- the per-device context of context.h would serialise them anyway); a single device
// runs on the caller's thread.  The error text of a failing block is thread-local to its thread,
// so it is carried back and re-raised on the caller's thread.
//
// `share_data`: the blocks all read the same day of data from the host; with more than one distinct
// device (and option multi.peer_fanout) the first device uploads it and the others copy it device ->
// device (DataFanout, context.h) -- the first block of every distinct device takes part, fn(i) of those
// blocks runs with the role armed for the host-pointer call it makes.
template <typename Fn>
int run_blocks(const std::vector<int>& dev, Fn fn, bool share_data = false)
{
    const size_t n_blocks = dev.size();
    std::vector<int> rc(n_blocks, 0);
    std::vector<std::string> msg(n_blocks);
    // blocks of every distinct device, in list order
    std::vector<int> distinct;
    std::vector<std::vector<size_t>> blocks_of;
    for (size_t i = 0; i < n_blocks; ++i) {
        size_t q = 0;
        while (q < distinct.size() && distinct[q] != dev[i]) ++q;
        if (q == distinct.size()) { distinct.push_back(dev[i]); blocks_of.emplace_back(); }
        blocks_of[q].push_back(i);
    }
    // Nothing may leave a worker as an exception (an exception that escapes a std::thread, or one thrown
    // while joinable threads are alive, ends the process with std::terminate); a failure stays visible
    // as status -3 with its text.
    std::unique_lock<std::mutex> fan_lock(bpmf::g_fanout_mutex, std::defer_lock);
    std::unique_ptr<bpmf::DataFanout> fan;
    if (share_data && distinct.size() > 1 && bpmf::option(bpmf::OPT_MULTI_PEER_FANOUT) != 0 && fan_lock.try_lock())
        fan.reset(new bpmf::DataFanout((int)distinct.size() - 1));
    auto work = [&](size_t q) {
        for (size_t i : blocks_of[q]) {
            const bool first = i == blocks_of[q][0];
            bpmf::FanoutArm arm(fan.get(), !first ? bpmf::DataFanout::NONE
                                                  : (q == 0 ? bpmf::DataFanout::SOURCE : bpmf::DataFanout::PEER));
            try {
                bpmf::last_error_buf()[0] = 0;
                rc[i] = fn(i);
                // (a status-0 block may leave a NOTE -- the peer copy of the day fell back to host uploads,
                // context.hip -- which the caller's thread passes on)
                if (rc[i] || strncmp(bpmf_last_error(), "note:", 5) == 0) msg[i] = bpmf_last_error();
            } catch (const std::exception& e) {
                rc[i] = -3;
                msg[i] = std::string("exception in a device block: ") + e.what();
            } catch (...) {
                rc[i] = -3;
                msg[i] = "unknown exception in a device block";
            }
            if (rc[i]) break;          // the call fails anyway: the device's other blocks are not run
        }
    };
    if (distinct.size() == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        th.reserve(distinct.size());
        for (size_t q = 1; q < distinct.size(); ++q) {
            try {
                th.emplace_back(work, q);
            } catch (const std::system_error&) {
                // no thread to be had (process / cgroup limit): this device runs here, BEFORE the first device --
                // nobody may wait for a hand-over from it
                if (fan) fan->cancel();
                work(q);
            }
        }
        work(0);                   // the first device on the caller's thread
        for (auto& t : th) t.join();
    }
    for (size_t i = 0; i < n_blocks; ++i)
        if (rc[i]) {
            set_error("%s", msg[i].c_str());
            return rc[i];
        }
    for (size_t i = 0; i < n_blocks; ++i)
        if (!msg[i].empty()) {
            set_error("%s", msg[i].c_str());        // status 0 with a note
            break;
        }
    return 0;
}

}  // namespace

extern "C" int bpmf_mf_run_multi(const float* templates, const int32_t* moveouts,
                                 const float* weights, const float* data, size_t step, size_t L,
                                 size_t N, size_t T, size_t S, size_t C, size_t n_corr,
                                 int network_sum, int flags, int n_devices, const int* devices,
                                 float* cc_out)
{
    if (!templates || !moveouts || !weights || !data || !cc_out || T == 0) {
        set_error("bpmf_mf_run_multi: bad argument");
        return -1;
    }
    std::vector<int> dev;
    if (int rc = resolve_devices(n_devices, devices, T, dev, "bpmf_mf_run_multi")) return rc;
    const size_t n_ch = S * C;
    const std::vector<size_t> b = weighted_bounds(weights, T, n_ch, dev.size());
    const size_t row = n_corr * (network_sum ? 1 : n_ch);   // floats of output per template
    return run_blocks(dev, [&](size_t i) -> int {
        const size_t t0 = b[i], nt = b[i + 1] - b[i];
        if (nt == 0) return 0;
        return bpmf_mf_run(templates + t0 * n_ch * L, moveouts + t0 * n_ch, weights + t0 * n_ch, data,
                           step, L, N, nt, S, C, n_corr, network_sum, flags, dev[i],
                           cc_out + t0 * row);
    }, true);
}

// The split bpmf_mf_run_multi applies, for callers and tests: bounds_out[0 .. n_blocks] (no device needed).
extern "C" int bpmf_mf_shard_bounds(const float* weights, size_t T, size_t S, size_t C, size_t n_blocks,
                                    size_t* bounds_out)
{
    if (!weights || !bounds_out || T == 0 || n_blocks == 0) {
        set_error("bpmf_mf_shard_bounds: bad argument");
        return -1;
    }
    const std::vector<size_t> b = weighted_bounds(weights, T, S * C, n_blocks);
    for (size_t i = 0; i <= n_blocks; ++i) bounds_out[i] = b[i];
    return 0;
}

extern "C" int bpmf_bp_run_multi(const float* features, const int32_t* moveouts,
                                 const float* w_phases, const float* w_sources, size_t N, size_t K,
                                 size_t S, size_t C, size_t P, int out_of_bounds, int reduce,
                                 int n_devices, const int* devices, float* beam_out,
                                 int32_t* arg_out)
{
    if (!features || !moveouts || !w_phases || !w_sources || !beam_out || K == 0) {
        set_error("bpmf_bp_run_multi: bad argument");
        return -1;
    }
    if (reduce == BPMF_BP_REDUCE_MAX && !arg_out) {
        set_error("bpmf_bp_run_multi: reduce=max needs an arg-max output");
        return -1;
    }
    std::vector<int> dev;
    if (int rc = resolve_devices(n_devices, devices, K, dev, "bpmf_bp_run_multi")) return rc;
    const std::vector<size_t> b = block_bounds(K, dev.size());
    if (reduce != BPMF_BP_REDUCE_MAX) {
        return run_blocks(dev, [&](size_t i) -> int {
            const size_t k0 = b[i], nk = b[i + 1] - b[i];
            if (nk == 0) return 0;
            return bpmf_bp_run(features, moveouts + k0 * S * P, w_phases, w_sources + k0 * S, N, nk, S,
                               C, P, out_of_bounds, reduce, dev[i], beam_out + k0 * N, nullptr);
        }, true);
    }
    if (dev.size() == 1)   // (through run_blocks: its exception barrier)
        return run_blocks(dev, [&](size_t) -> int {
            return bpmf_bp_run(features, moveouts, w_phases, w_sources, N, K, S, C, P, out_of_bounds,
                               reduce, dev[0], beam_out, arg_out);
        });
    // option bp.compat_first_computed: every share keeps -inf where it computed no beam (a finished
    // share's (0, first id) could not be told from a real 0); the host finishes after the merge
    const bool first_computed = bpmf::option(bpmf::OPT_BP_COMPAT_FIRST_COMPUTED) != 0;
    // block 0 writes into the caller's arrays, the others into scratch vectors
    std::vector<std::vector<float>> pb(dev.size());
    std::vector<std::vector<int32_t>> pa(dev.size());
    for (size_t i = 1; i < dev.size(); ++i) { pb[i].resize(N); pa[i].resize(N); }
    int rc = run_blocks(dev, [&](size_t i) -> int {
        const size_t k0 = b[i], nk = b[i + 1] - b[i];
        if (nk == 0) return 0;
        struct Defer {
            explicit Defer(bool on) { bpmf::t_bp_defer_finish = on; }
            ~Defer() { bpmf::t_bp_defer_finish = false; }
        } defer(first_computed);
        return bpmf_bp_run(features, moveouts + k0 * S * P, w_phases, w_sources + k0 * S, N, nk, S, C,
                           P, out_of_bounds, reduce, dev[i], i ? pb[i].data() : beam_out,
                           i ? pa[i].data() : arg_out);
    }, true);
    if (rc) return rc;
    // Ascending source blocks and a strict >: block 0 carries the (0, source 0) starting point of
    // the sequential scan, and a later block only replaces a value it beats.
    // (threads by the CPUs the process may use -- affinity and cgroup quota --, not hardware_concurrency())
    const size_t nth = std::max<size_t>(1, std::min<size_t>(16, bpmf::usable_cpus()));
    const std::vector<size_t> tb = block_bounds(N, N < (1u << 20) ? 1 : nth);
    std::vector<std::thread> th;
    auto merge = [&](size_t lo, size_t hi) {
        for (size_t i = 1; i < dev.size(); ++i) {
            if (b[i + 1] == b[i]) continue;
            const float* bb = pb[i].data();
            const int32_t* aa = pa[i].data();
            const int32_t k0 = (int32_t)b[i];
            for (size_t t = lo; t < hi; ++t)
                if (bb[t] > beam_out[t]) { beam_out[t] = bb[t]; arg_out[t] = aa[t] + k0; }
        }
        if (first_computed)
            for (size_t t = lo; t < hi; ++t)
                if (beam_out[t] == -INFINITY) { beam_out[t] = 0.0f; arg_out[t] = 0; }
    };
    if (tb.size() == 2) {
        merge(0, N);
    } else {
        for (size_t q = 0; q + 1 < tb.size(); ++q) {
            try {
                th.emplace_back(merge, tb[q], tb[q + 1]);
            } catch (const std::system_error&) {
                merge(tb[q], tb[q + 1]);
            }
        }
        for (auto& t : th) t.join();
    }
    return 0;
}
