// Per-device context of the host-pointer entry points: see context.h.
#include "context.h"
#include "../../include/bpmf_hip.h"

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <memory>
#include <sched.h>
#include <cstring>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

namespace bpmf {
namespace {
std::mutex g_registry_mutex;
std::vector<DeviceContext*> g_contexts;     // by device index; entries live until the process ends

constexpr size_t DEV_GRANULE = (size_t)2 << 20;
constexpr size_t SHRINK_ABOVE = (size_t)1 << 30;   // a cached block this large is given back ...
constexpr size_t SHRINK_RATIO = 4;                 // ... when calls need less than 1/4 of it ...
constexpr int SHRINK_STREAK = 4;                   // ... four times in a row

void drain(DeviceContext* c)
{
    if (c->s_run) (void)hipStreamSynchronize(c->s_run);
    if (c->s_copy) (void)hipStreamSynchronize(c->s_copy);
    for (hipStream_t s : c->s_side)
        if (s) (void)hipStreamSynchronize(s);
}
}  // namespace

char* DeviceContext::reserve_device(size_t bytes)
{
    bytes = std::max<size_t>(bytes, 256);
    const bool too_small = bytes > dev_cap;
    const bool oversized = dev_cap >= SHRINK_ABOVE && bytes < dev_cap / SHRINK_RATIO;
    small_streak = oversized ? small_streak + 1 : 0;
    const bool too_large = oversized && small_streak >= SHRINK_STREAK;
    if (dev_buf && !too_small && !too_large) return dev_buf;
    small_streak = 0;
    if (dev_buf) {
        drain(this);
        (void)hipFree(dev_buf);
        dev_buf = nullptr;
        dev_cap = 0;
    }
    // growth with slack (a sequence of slightly larger calls does not reallocate every time)
    size_t want = too_small && bytes < SHRINK_ABOVE ? bytes + bytes / 8 : bytes;
    want = align_up(want, DEV_GRANULE);
    hipError_t e = hipMalloc((void**)&dev_buf, want);
    if (e != hipSuccess && want > bytes) {          // the slack is a convenience, not a need
        (void)hipGetLastError();
        want = align_up(bytes, 256);
        e = hipMalloc((void**)&dev_buf, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        dev_buf = nullptr;
        set_error("device %d: hipMalloc(%zu) for the working set failed: %s", device, want, hipGetErrorString(e));
        return nullptr;
    }
    dev_cap = want;
    return dev_buf;
}

int DeviceContext::reserve_pinned(size_t bytes)
{
    bytes = std::max<size_t>(bytes, 4096);
    if (pinned[0] && pinned[1] && bytes <= pinned_cap) return 0;
    drain(this);
    copy_pool_quiesce();          // no host thread is still writing into a piece
    upload_inflight[0] = upload_inflight[1] = false;
    for (int i = 0; i < 2; ++i) {
        if (pinned[i]) (void)hipHostFree(pinned[i]);
        pinned[i] = nullptr;
    }
    pinned_cap = 0;
    for (int i = 0; i < 2; ++i) {
        hipError_t e = hipHostMalloc((void**)&pinned[i], bytes, hipHostMallocDefault);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_error("device %d: hipHostMalloc(%zu) failed: %s", device, bytes, hipGetErrorString(e));
            if (i == 1) { (void)hipHostFree(pinned[0]); pinned[0] = nullptr; }
            pinned[i] = nullptr;
            return -2;
        }
    }
    pinned_cap = bytes;
    return 0;
}

thread_local HostCallStats t_call_stats;

// CPUs this process may run on: the affinity mask cut by the cgroup's CPU quota (v2 cpu.max, v1 cfs quota).
unsigned usable_cpus()
{
    static const unsigned cached = [] {
        unsigned n = std::thread::hardware_concurrency();
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0) {
            const int c = CPU_COUNT(&set);
            if (c > 0) n = (unsigned)c;
        }
        double quota = 0.0;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char a[64] = {0};
            double period = 0.0;
            if (fscanf(f, "%63s %lf", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0) quota = atof(a) / period;
            fclose(f);
        } else {
            double q = -1.0, per = 0.0;
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lf", &q) != 1) q = -1.0; fclose(g); }
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lf", &per) != 1) per = 0.0; fclose(g); }
            if (q > 0 && per > 0) quota = q / per;
        }
        if (quota >= 1.0 && quota < (double)n) n = (unsigned)quota;
        return std::max(1u, n);
    }();
    return cached;
}

namespace {
// The host threads that copy between pageable and pinned memory (the day of data on its way up, the CC matrix
// on its way down): created ONCE per process, on first use, and parked on a condition variable between jobs.
// Rounds 4-5 spawned up to 8 std::threads per 64 MB piece and sized the team by hardware_concurrency() / 2
// (128 on a box that grants 16 CPUs): on a loaded host -- the driver's round-5 bench ran at load average 39 --
// thread creation queued behind the other processes and the first pieces of a day were exposed again (cfg3
// through bpmf_bp_run: 183 ms against 150 resident).  The pool is leaked on purpose: a worker parked inside a
// destroyed condition variable at process exit is worse than a few threads the OS reaps.
class CopyPool {
public:
    static CopyPool& get()
    {
        static CopyPool* p = new CopyPool();
        return *p;
    }
    unsigned team() const { return n_workers + 1; }      // the workers and the calling thread
    // fn(i) for i in [0, n): drawn by the workers and the caller; returns when every task is done.  One job at a
    // time (callers on different devices take turns: a copy saturates the memory system anyway).
    //
    // `token` != nullptr marks the tasks as IDEMPOTENT copies INTO the buffer `token` (a pinned piece being filled
    // from the caller's array): once nothing is left to draw the caller does NOT sleep until the slowest worker
    // reports back -- on a host with more runnable threads than CPUs a worker that lost its CPU in the middle of a
    // block is off for a scheduler period, and the caller's own wake-up costs another -- it copies the blocks that
    // are still marked unfinished ITSELF (the same bytes to the same place) and returns.  A worker that comes back
    // late finishes its block with the same bytes; the only thing that must never happen is that it is still
    // writing when the buffer is filled with the NEXT piece that goes through it: a job into `token` first waits for
    // the stragglers of the earlier jobs into the same token (two pieces ago: it has normally long finished), and
    // quiesce() waits for all of them (before a pinned piece is freed or used for the way down).
    void run(size_t n, const std::function<void(size_t)>& fn, const void* token = nullptr)
    {
        if (n == 0) return;
        if (n == 1 || n_workers == 0) {
            for (size_t i = 0; i < n; ++i) fn(i);
            return;
        }
        std::lock_guard<std::mutex> job_lock(job_mutex);
        auto job = std::make_shared<Job>();
        job->fn = fn;
        job->n = n;
        job->token = token;
        job->state.reset(new std::atomic<unsigned char>[n]);
        for (size_t i = 0; i < n; ++i) job->state[i].store(0, std::memory_order_relaxed);
        {
            std::unique_lock<std::mutex> g(m);
            if (token) cv_done.wait(g, [&] { return stragglers_of(token) == 0; });
            else cv_done.wait(g, [&] { return straggling.empty(); });
            current = job;
            ++generation;
        }
        cv.notify_all();
        draw(*job);
        if (token) {
            // nothing left to draw: finish what others hold, without waiting for them
            for (size_t i = 0; i < n; ++i)
                if (job->state[i].load(std::memory_order_acquire) != 2) {
                    job->fn(i);
                    job->state[i].store(2, std::memory_order_release);
                }
            std::lock_guard<std::mutex> g(m);
            current.reset();
            if (job->inside > 0) straggling.push_back(job);     // (its workers take themselves off the list)
        } else {
            std::unique_lock<std::mutex> g(m);
            cv_done.wait(g, [&] { return job->inside == 0 && job->next.load() >= n; });
            current.reset();
        }
    }
    // every straggler of every earlier job has finished
    void quiesce()
    {
        std::unique_lock<std::mutex> g(m);
        cv_done.wait(g, [&] { return straggling.empty(); });
    }
private:
    struct Job {
        std::function<void(size_t)> fn;
        size_t n = 0;
        const void* token = nullptr;
        std::atomic<size_t> next{0};
        std::unique_ptr<std::atomic<unsigned char>[]> state;      // 0 not drawn, 1 drawn, 2 done
        int inside = 0;                                           // workers inside draw() (under m)
    };
    CopyPool()
    {
        const unsigned cpus = usable_cpus();
        const unsigned want = std::min(11u, cpus > 2 ? cpus - 2 : (cpus > 1 ? 1u : 0u));
        for (unsigned i = 0; i < want; ++i) {
            try {
                std::thread([this] { worker(); }).detach();
                ++n_workers;
            } catch (const std::system_error&) {
                break;            // no thread to be had: the caller copies alone
            }
        }
    }
    size_t stragglers_of(const void* token) const
    {
        size_t k = 0;
        for (auto& j : straggling) k += j->token == token;
        return k;
    }
    static void draw(Job& job)
    {
        for (;;) {
            const size_t i = job.next.fetch_add(1, std::memory_order_relaxed);
            if (i >= job.n) return;
            job.state[i].store(1, std::memory_order_relaxed);
            job.fn(i);
            job.state[i].store(2, std::memory_order_release);
        }
    }
    void worker()
    {
        unsigned long long seen = 0;
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> g(m);
                cv.wait(g, [&] { return generation != seen && current && current->next.load() < current->n; });
                seen = generation;
                job = current;
                ++job->inside;
            }
            draw(*job);
            {
                std::lock_guard<std::mutex> g(m);
                if (--job->inside == 0)
                    for (size_t k = 0; k < straggling.size(); ++k)
                        if (straggling[k] == job) { straggling.erase(straggling.begin() + k); break; }
            }
            cv_done.notify_all();
        }
    }
    std::mutex job_mutex, m;
    std::condition_variable cv, cv_done;
    std::shared_ptr<Job> current;
    std::vector<std::shared_ptr<Job>> straggling;
    unsigned long long generation = 0;
    unsigned n_workers = 0;
};
}  // namespace

void copy_pool_quiesce() { CopyPool::get().quiesce(); }

void parallel_copy(char* dst, const char* src, size_t bytes)
{
    CopyPool& pool = CopyPool::get();
    if (bytes < (8u << 20) || pool.team() == 1) {
        pool.quiesce();
        memcpy(dst, src, bytes);
        return;
    }
    // 2 MB blocks, drawn by whoever is free (a thread that loses its CPU for a while holds up one block, not an
    // eighth of the piece)
    const size_t blk = (size_t)2 << 20, n = (bytes + blk - 1) / blk;
    pool.run(n, [=](size_t i) { memcpy(dst + i * blk, src + i * blk, std::min(blk, bytes - i * blk)); });
}

// Device memory -> the caller's PAGEABLE array through the context's pinned pieces (the way down of
// staged_upload_rows): the runtime's own pageable path page-locks a destination it has not seen before -- a result
// array is a new allocation on every call -- and took 40 ms for the 35 MB of a cfg3 day's (max-beam, arg-max)
// on the second call of a process (tools/probe_bp_e2e.py, profiles/r06_bp_e2e.txt) where the copy itself is
// 0.7 ms.  Blocks until `host` holds the bytes.  Needs reserve_pinned() first.
hipError_t staged_download(DeviceContext* ctx, void* host, const void* d_src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return hipSuccess;
    const size_t cap = ctx->pinned_cap.load();
    if (cap == 0) return hipErrorInvalidValue;
    hipError_t e = hipSuccess;
    // (the pieces may still be read by this call's last uploads)
    for (int i = 0; i < 2; ++i)
        if (ctx->upload_inflight[i]) {
            if ((e = hipEventSynchronize(ctx->ev_piece[i])) != hipSuccess) return e;
            ctx->upload_inflight[i] = false;
        }
    copy_pool_quiesce();
    const size_t n_piece = (bytes + cap - 1) / cap;
    auto enqueue = [&](size_t q) {
        const size_t o = q * cap, len = std::min(cap, bytes - o);
        e = hipMemcpyAsync(ctx->pinned[q & 1], (const char*)d_src + o, len, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_piece[q & 1], stream);
    };
    enqueue(0);
    for (size_t q = 0; q < n_piece && e == hipSuccess; ++q) {
        if ((e = hipEventSynchronize(ctx->ev_piece[q & 1])) != hipSuccess) break;
        if (q + 1 < n_piece) enqueue(q + 1);
        if (e != hipSuccess) break;
        const size_t o = q * cap, len = std::min(cap, bytes - o);
        parallel_copy((char*)host + o, ctx->pinned[q & 1], len);
    }
    return e;
}

hipError_t staged_upload_rows(DeviceContext* ctx, float* d_dst, const float* host, size_t rows, size_t N, size_t c0,
                              size_t c1, hipStream_t stream)
{
    if (c1 <= c0 || rows == 0) return hipSuccess;
    // An array that is ALREADY page-locked (hipHostMalloc / hipHostRegister: a torch pinned tensor seen through
    // NumPy) needs no staging: the DMA engine reads it where it lies.
    {
        hipPointerAttribute_t a0, a1;
        const bool pinned_src = hipPointerGetAttributes(&a0, host) == hipSuccess && a0.type == hipMemoryTypeHost &&
                                hipPointerGetAttributes(&a1, host + rows * N - 1) == hipSuccess && a1.type == hipMemoryTypeHost;
        (void)hipGetLastError();          // (a pageable pointer makes the query fail: not an error of this call)
        if (pinned_src)
            return hipMemcpy2DAsync(d_dst + c0, N * sizeof(float), host + c0, N * sizeof(float), (c1 - c0) * sizeof(float), rows,
                                    hipMemcpyHostToDevice, stream);
    }
    const size_t cap = ctx->pinned_cap.load();
    // samples per pinned piece (whole rows' segments; a multiple of 1024 samples where the piece allows)
    size_t w = cap / (rows * sizeof(float));
    if (w == 0) return hipErrorInvalidValue;
    if (w > 1024) w &= ~(size_t)1023;
    CopyPool& pool = CopyPool::get();
    t_call_stats.fill_threads = (int)pool.team();
    hipError_t e = hipSuccess;
    size_t q = ctx->upload_seq;
    for (size_t a = c0; a < c1 && e == hipSuccess; a += w, ++q) {
        const size_t len = std::min(w, c1 - a);
        char* pin = ctx->pinned[q & 1];
        if (ctx->upload_inflight[q & 1]) {                       // the copy that last read this piece
            const double tw = host_now_ms();
            if ((e = hipEventSynchronize(ctx->ev_piece[q & 1])) != hipSuccess) break;
            ctx->upload_inflight[q & 1] = false;
            const double dw = host_now_ms() - tw;
            t_call_stats.pinned_wait_ms += dw;
            if (option(OPT_BP_VERBOSE) != 0 && dw > 1.0)
                fprintf(stderr, "[bpmf] staged upload: piece %d waited %.1f ms for its pinned buffer\n", t_call_stats.pieces, dw);
        }
        const double t0 = host_now_ms();
        // row segments -> the piece, packed: blocks of at most 1 MB of a row, drawn by whoever is free
        const size_t row_bytes = len * sizeof(float);
        if (rows * row_bytes < (4u << 20) || pool.team() == 1) {
            for (size_t r = 0; r < rows; ++r) memcpy(pin + r * row_bytes, host + r * N + a, row_bytes);
        } else {
            const size_t blk = (size_t)1 << 20, per_row = (row_bytes + blk - 1) / blk;
            pool.run(rows * per_row, [=](size_t i) {
                const size_t r = i / per_row, o = (i % per_row) * blk;
                memcpy(pin + r * row_bytes + o, (const char*)(host + r * N + a) + o, std::min(blk, row_bytes - o));
            }, pin);
        }
        const double t1 = host_now_ms();
        t_call_stats.host_copy_ms += t1 - t0;
        ++t_call_stats.pieces;
        e = hipMemcpy2DAsync(d_dst + a, N * sizeof(float), pin, len * sizeof(float), len * sizeof(float), rows,
                             hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_piece[q & 1], stream);
        if (e == hipSuccess) ctx->upload_inflight[q & 1] = true;
        t_call_stats.enqueue_ms += host_now_ms() - t1;       // (the runtime's own time inside the copy call)
    }
    ctx->upload_seq = q;
    return e;
}

void DeviceContext::trim_after_call()
{
    const long limit_mb = option(OPT_HOST_CACHE_LIMIT_MB);
    if (limit_mb > 0 && dev_cap.load() > (size_t)limit_mb << 20) release_memory();
}

void DeviceContext::release_memory()
{
    drain(this);
    copy_pool_quiesce();
    if (dev_buf) (void)hipFree(dev_buf);
    dev_buf = nullptr;
    dev_cap = 0;
    for (int i = 0; i < 2; ++i) {
        if (pinned[i]) (void)hipHostFree(pinned[i]);
        pinned[i] = nullptr;
    }
    pinned_cap = 0;
}

DeviceContext* device_context(int device)
{
    std::lock_guard<std::mutex> g(g_registry_mutex);
    if (device < 0) {
        set_error("device %d out of range", device);
        return nullptr;
    }
    if ((size_t)device < g_contexts.size() && g_contexts[device]) return g_contexts[device];
    int visible = 0;                     // logical devices (option debug.virtual_devices, common.h)
    hipError_t e = device_counts(&visible, nullptr);
    if (e != hipSuccess || device >= visible) {
        set_error("device %d out of range (%d visible%s%s)", device, visible, e != hipSuccess ? ": " : "",
                  e != hipSuccess ? hipGetErrorString(e) : "");
        return nullptr;
    }
    const int physical = physical_device(device);
    DeviceGuard bind(device, physical);
    if (bind.error() != hipSuccess) {
        set_error("hipSetDevice(%d) failed: %s", physical, hipGetErrorString(bind.error()));
        return nullptr;
    }
    DeviceContext* c = new DeviceContext();
    c->device = device;
    c->physical = physical;
    hipError_t err = hipSuccess;
    auto step = [&](hipError_t r) { if (err == hipSuccess) err = r; };
    step(hipStreamCreateWithFlags(&c->s_run, hipStreamNonBlocking));
    // The copy stream gets the HIGHEST priority, i.e. a hardware queue of its own class: the runtime maps streams
    // onto a handful of hardware queues per priority, and with torch's streams, s_run and the side streams alive
    // the copy stream otherwise lands on a queue it SHARES with a compute stream -- a piece of the day in flight
    // then holds back the kernels queued behind it and the upload no longer runs beside them (round 5: the
    // piecewise uploads gained nothing until this was found).
    {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
        step(hipStreamCreateWithPriority(&c->s_copy, hipStreamNonBlocking, hi));
    }
    for (hipStream_t& s : c->s_side) step(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    step(hipEventCreateWithFlags(&c->ev_data, hipEventDisableTiming));
    for (hipEvent_t& e : c->ev_chunk) step(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) {
        step(hipEventCreateWithFlags(&c->ev_batch[i], hipEventDisableTiming));
        step(hipEventCreateWithFlags(&c->ev_piece[i], hipEventDisableTiming));
    }
    if (err != hipSuccess) {
        // (a half-built context is dropped without destroying what exists: this path means the
        // runtime is already in trouble, and the objects are a few hundred bytes)
        set_error("device %d: creating the streams / events of the host entry points failed: %s", device,
                  hipGetErrorString(err));
        delete c;
        return nullptr;
    }
    if (g_contexts.size() <= (size_t)device) g_contexts.resize((size_t)device + 1, nullptr);
    g_contexts[device] = c;
    return c;
}

std::mutex g_fanout_mutex;
thread_local DataFanout* t_fanout = nullptr;
thread_local int t_fanout_role = DataFanout::NONE;

namespace {
// peer access src -> dst enabled once per ordered pair (a direct xGMI copy instead of one staged through
// the host); a refusal is not an error, hipMemcpyPeerAsync works without it
std::mutex g_peer_mutex;
std::vector<std::pair<int, int>> g_peer_tried;
void ensure_peer_access(int dst_physical, int src_physical)
{
    if (dst_physical == src_physical) return;
    std::lock_guard<std::mutex> g(g_peer_mutex);
    for (auto& p : g_peer_tried)
        if (p.first == dst_physical && p.second == src_physical) return;
    g_peer_tried.emplace_back(dst_physical, src_physical);
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dst_physical, src_physical) == hipSuccess && can)
        (void)hipDeviceEnablePeerAccess(src_physical, 0);     // (the current device is dst_physical)
    (void)hipGetLastError();                                  // "already enabled" (torch, RCCL) is fine
}
}  // namespace

namespace {
// Does a device-to-device copy src -> dst work at all?  Probed once per ordered pair of PHYSICAL devices with a
// 4 KB copy that is waited for: the hand-over between two distinct GPUs (hipDeviceEnablePeerAccess +
// hipMemcpyPeerAsync over xGMI) had never executed when this was written -- no multi-GPU box was available in
// rounds 1-6 -- and a platform that refuses it (IOMMU settings, an isolated partition, a container without the
// other device's render node) must cost one warning, not the first 8-GPU measurement: a pair that fails the probe
// uploads from the host on every device, like option multi.peer_fanout = 0.  Option debug.fail_peer_copy (tests)
// makes every probe fail, same-device pairs included.
std::vector<std::pair<std::pair<int, int>, bool>> g_peer_probed;     // under g_peer_mutex
bool peer_copy_works(int dst_physical, int src_physical, void* d_dst, const void* d_src, size_t bytes, hipStream_t stream)
{
    const bool forced_failure = option(OPT_DEBUG_FAIL_PEER_COPY) != 0;
    if (dst_physical == src_physical && !forced_failure) return true;
    {
        std::lock_guard<std::mutex> g(g_peer_mutex);
        for (auto& p : g_peer_probed)
            if (p.first.first == dst_physical && p.first.second == src_physical && !forced_failure) return p.second;
    }
    hipError_t e = forced_failure ? hipErrorPeerAccessUnsupported
                                  : hipMemcpyPeerAsync(d_dst, dst_physical, d_src, src_physical, std::min<size_t>(bytes, 4096), stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    const bool ok = e == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        // (status stays 0: the text is a note for whoever asks bpmf_last_error() why the call was slower)
        set_error("note: device-to-device copy of the day from GPU %d to GPU %d failed (%s); every device uploads from the "
                  "host instead (as under multi.peer_fanout = 0)", src_physical, dst_physical, hipGetErrorString(e));
    }
    if (!forced_failure) {
        std::lock_guard<std::mutex> g(g_peer_mutex);
        g_peer_probed.push_back({{dst_physical, src_physical}, ok});
    }
    return ok;
}
}  // namespace

bool fanout_peer_copy(FanoutScope& scope, DeviceContext* ctx, void* d_dst, size_t bytes, hipStream_t stream,
                      hipError_t* err, const char** what)
{
    *err = hipSuccess;
    if (!scope.f || scope.role != DataFanout::PEER) return false;
    const void* d_src = nullptr;
    int src_physical = -1;
    hipEvent_t ready = nullptr;
    if (!scope.f->wait_published(&d_src, &src_physical, &ready)) return false;
    ensure_peer_access(ctx->physical, src_physical);
    if (!peer_copy_works(ctx->physical, src_physical, d_dst, d_src, bytes, stream)) return false;   // upload from the host
    *what = "waiting for the first device's upload";
    if ((*err = hipStreamWaitEvent(stream, ready, 0)) != hipSuccess) return true;
    *what = "device-to-device copy of the data";
    *err = hipMemcpyPeerAsync(d_dst, ctx->physical, d_src, src_physical, bytes, stream);
    if (*err != hipSuccess && ctx->physical != src_physical) {
        // the probe passed and the real copy is refused at enqueue time: same fallback, same note
        (void)hipGetLastError();
        set_error("note: device-to-device copy of the day from GPU %d to GPU %d was refused (%s); this device uploads from "
                  "the host instead", src_physical, ctx->physical, hipGetErrorString(*err));
        *err = hipSuccess;
        return false;
    }
    return true;
}

hipError_t fanout_publish(FanoutScope& scope, DeviceContext* ctx, const void* d_src, hipStream_t stream)
{
    if (!scope.f || scope.role != DataFanout::SOURCE) return hipSuccess;
    const hipError_t e = hipEventRecord(ctx->ev_data, stream);
    if (e != hipSuccess) return e;
    scope.f->publish(d_src, ctx->physical, ctx->ev_data);
    return hipSuccess;
}

hipError_t fanout_upload(FanoutScope& scope, DeviceContext* ctx, void* d_dst, const void* host, size_t bytes,
                         hipStream_t stream, const char** what)
{
    hipError_t e;
    if (fanout_peer_copy(scope, ctx, d_dst, bytes, stream, &e, what)) return e;
    *what = "H2D data";
    if ((e = hipMemcpyAsync(d_dst, host, bytes, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    *what = "event record";
    return fanout_publish(scope, ctx, d_dst, stream);
}

hipStream_t device_side_stream(int device)
{
    DeviceContext* c = device_context(device);
    return c ? c->s_side[c->next_side.fetch_add(1) % DeviceContext::SIDE_STREAMS] : nullptr;
}

}  // namespace bpmf

// Gives back what the host-pointer entry points keep between calls on `device` (all devices when
// device < 0): the device working set and the pinned staging pieces.  Waits for a running call.
extern "C" int bpmf_release_device_memory(int device)
{
    std::vector<bpmf::DeviceContext*> todo;
    {
        std::lock_guard<std::mutex> g(bpmf::g_registry_mutex);
        if (device >= 0 && ((size_t)device >= bpmf::g_contexts.size() || !bpmf::g_contexts[device])) return 0;
        for (size_t d = 0; d < bpmf::g_contexts.size(); ++d)
            if (bpmf::g_contexts[d] && (device < 0 || (size_t)device == d)) todo.push_back(bpmf::g_contexts[d]);
    }
    for (bpmf::DeviceContext* c : todo) {
        std::lock_guard<std::mutex> call(c->call_mutex);
        bpmf::DeviceGuard bind(c->device, c->physical);
        if (bind.error() != hipSuccess) {
            bpmf::set_error("bpmf_release_device_memory: hipSetDevice(%d) failed: %s", c->physical,
                            hipGetErrorString(bind.error()));
            return -2;
        }
        c->release_memory();
    }
    return 0;
}

extern "C" int bpmf_host_call_stats(double* out, int n)
{
    const bpmf::HostCallStats& st = bpmf::t_call_stats;
    const double v[10] = {st.total_ms, st.first_kernel_ms, st.host_copy_ms, st.device_wait_ms, (double)st.pieces,
                          (double)st.fill_threads, st.pinned_wait_ms, st.enqueue_ms, st.plan_ms, st.reserve_ms};
    int k = 0;
    for (; out && k < n && k < 10; ++k) out[k] = v[k];
    return k;
}

extern "C" int bpmf_device_memory_held(int device, size_t* device_bytes, size_t* pinned_bytes)
{
    std::lock_guard<std::mutex> g(bpmf::g_registry_mutex);
    size_t dv = 0, pn = 0;
    for (size_t d = 0; d < bpmf::g_contexts.size(); ++d) {
        const bpmf::DeviceContext* c = bpmf::g_contexts[d];
        if (!c || (device >= 0 && (size_t)device != d)) continue;
        dv += c->dev_cap.load();               // (atomics: read without the call mutex)
        pn += 2 * c->pinned_cap.load();
    }
    if (device_bytes) *device_bytes = dv;
    if (pinned_bytes) *pinned_bytes = pn;
    return 0;
}
