// Per-device context of the host-pointer entry points (bpmf_mf_run, bpmf_bp_run and their *_multi
// forms, bpmf_find_similar_sources): everything those calls used to create and destroy on every
// call -- two private streams, a side stream, events, two pinned staging pieces, the device working
// set -- exists ONCE per device and process here, and the calls on one device take turns
// (DeviceContext::call_mutex).  "Re-entrant or internally serialised per device" (SURVEY.md 8b).
//
// Why ("host threads and the HIP runtime"): rounds 2-3 ran every listed device on its
// own host thread even when a device was listed several times; each thread allocated, created
// streams and events, launched on the null stream and freed -- and under 8 competing processes the
// process died about once per 500 calls inside the HIP runtime (stream creation / destruction on one
// thread racing the device-wide synchronisation of hipFree / null-stream work on another).  The
// runtime objects are now created under one mutex, never destroyed while the library is loaded,
// and never touched by two host-pointer calls at once.
#pragma once
#include "common.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>

namespace bpmf {

struct DeviceContext {
    int device = 0;                   // logical device (common.h)
    int physical = 0;                 // the GPU it lives on, fixed at creation
    std::mutex call_mutex;            // one host-pointer call on this device at a time
    hipStream_t s_run = nullptr;      // kernels and H2D of the call (non-blocking: never the null stream)
    hipStream_t s_copy = nullptr;     // D2H of finished batches
    // edge tiles of the backprojection beside the interior kernels: a small pool dealt round-robin to
    // the device's plans at their creation (never created or destroyed afterwards).  Host-pointer calls
    // are serialised by call_mutex whatever stream their plan holds; resident plans driven from several
    // threads through bpmf_bp_run_dev mostly get different side streams and do not queue behind each
    // other's edge tiles (with more than SIDE_STREAMS live plans two of them share one: still correct --
    // every plan forks and joins with its own events -- only serialised).
    static constexpr int SIDE_STREAMS = 4;
    hipStream_t s_side[SIDE_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
    std::atomic<unsigned> next_side{0};
    hipEvent_t ev_data = nullptr;     // *_run_multi: the day of data has arrived on this device (peer fan-out)
    static constexpr int CHUNK_EVENTS = 4;
    hipEvent_t ev_chunk[CHUNK_EVENTS] = {nullptr, nullptr, nullptr, nullptr};   // pieces of a day arriving while the kernels run
    hipEvent_t ev_batch[2] = {nullptr, nullptr};
    hipEvent_t ev_piece[2] = {nullptr, nullptr};
    // grow-only buffers, valid while call_mutex is held
    char* dev_buf = nullptr;
    std::atomic<size_t> dev_cap{0};
    char* pinned[2] = {nullptr, nullptr};
    std::atomic<size_t> pinned_cap{0};
    // staged_upload_rows: which pinned piece comes next, and whether a copy recorded in ev_piece[i] still reads piece i
    size_t upload_seq = 0;
    bool upload_inflight[2] = {false, false};
    int small_streak = 0;             // consecutive calls that needed less than a quarter of a >= 1 GiB block

    // At least `bytes` of device memory (256-byte aligned base).  A larger request frees the old
    // block first (after the streams have drained); a block of 1 GiB or more is given back after four
    // consecutive calls that needed less than a quarter of it (a workflow that alternates a large and a
    // small call keeps the large block).  nullptr + error text on failure.
    char* reserve_device(size_t bytes);
    // Two pinned host pieces of at least `bytes` each.  0 on success.
    int reserve_pinned(size_t bytes);
    // Frees the device working set and the pinned pieces (the streams and events stay).
    void release_memory();
    // End of a host-pointer call (its streams drained, call_mutex held): option host.cache_limit_mb -- a working
    // set above the limit goes back to the allocator right away (a workflow that makes ONE large host-pointer call
    // and then runs on torch-allocated tensors would otherwise strand it outside torch's allocator).
    void trim_after_call();
};

// bpmf_host_call_stats: where the time of the calling thread's last host-pointer call went (include/bpmf_hip.h)
struct HostCallStats {
    double total_ms = 0, first_kernel_ms = 0, host_copy_ms = 0, device_wait_ms = 0, pinned_wait_ms = 0, enqueue_ms = 0, plan_ms = 0, reserve_ms = 0;
    int pieces = 0, fill_threads = 0;
};
extern thread_local HostCallStats t_call_stats;
inline double host_now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// host threads that copy a day into the pinned pieces: the CPUs this process may actually use (affinity mask and
// cgroup quota), not hardware_concurrency() -- a GPU box shows 256 logical CPUs and grants 16
unsigned usable_cpus();

// The day of data of one *_run_multi call on several devices (option multi.peer_fanout): the FIRST device
// of the list uploads it from the host once and publishes where it lies; every other device copies it
// device -> device (hipMemcpyPeerAsync on its own stream, behind the source's event) instead of pulling
// the same gigabytes through the host's pageable-memory staging a second, third ... eighth time (SURVEY.md
// section 8e: "broadcast once per day").  The source keeps its copy alive until every peer is through with
// its call (wait_peers at the end of the source's call, under its context mutex); a peer that finds the
// hand-over cancelled or failed uploads from the host like a single-device call.  Upstream's GPU back-ends
// copy the whole day to every device from the host.
struct DataFanout {
    enum Role { NONE = 0, SOURCE = 1, PEER = 2 };
    explicit DataFanout(int n_peers) : pending(n_peers) {}
    // source: the copy at `d_src` (device memory of physical GPU `physical`) is complete once `ready` fires
    void publish(const void* d_src_, int physical, hipEvent_t ready_)
    {
        std::lock_guard<std::mutex> g(m);
        if (published) return;
        d_src = d_src_; src_physical = physical; ready = ready_;
        published = true;
        cv.notify_all();
    }
    // nothing will be published (the source failed before its upload, or never ran): peers upload themselves
    void cancel()
    {
        std::lock_guard<std::mutex> g(m);
        if (!published) { published = true; failed = true; }
        cv.notify_all();
    }
    // peer: blocks until the source has published or cancelled; false = upload from the host yourself
    bool wait_published(const void** d_src_, int* physical, hipEvent_t* ready_)
    {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return published; });
        if (failed) return false;
        *d_src_ = d_src; *physical = src_physical; *ready_ = ready;
        return true;
    }
    void peer_done()
    {
        std::lock_guard<std::mutex> g(m);
        if (pending > 0) --pending;
        cv.notify_all();
    }
    // source: until every peer has called peer_done (or the hand-over was cancelled)
    void wait_peers()
    {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return pending <= 0 || failed; });
    }
private:
    std::mutex m;
    std::condition_variable cv;
    const void* d_src = nullptr;
    int src_physical = -1;
    hipEvent_t ready = nullptr;
    bool published = false, failed = false;
    int pending;
};
// the hand-over the calling thread's NEXT host-pointer call takes part in (set by *_run_multi around the
// first block of every distinct device; nullptr otherwise) and its role in it
extern thread_local DataFanout* t_fanout;
extern thread_local int t_fanout_role;
// RAII of a participant: whatever happens inside the call, the source cancels an unpublished hand-over
// and waits for its peers, a peer reports that it is through
struct FanoutScope {
    DataFanout* f;
    int role;
    FanoutScope() : f(t_fanout), role(t_fanout_role) { t_fanout = nullptr; t_fanout_role = DataFanout::NONE; }
    ~FanoutScope() { finish(); }
    // Settles the participant's part once (idempotent).  A SOURCE must call this BEFORE anything that may free
    // the buffer it published -- DeviceContext::trim_after_call() under option host.cache_limit_mb -- because its
    // peers copy from that buffer until they report peer_done (the destructor alone runs at function return,
    // behind the trim: round-5 advisor finding).
    void finish()
    {
        if (!f) return;
        if (role == DataFanout::SOURCE) { f->cancel(); f->wait_peers(); }
        else if (role == DataFanout::PEER) f->peer_done();
        f = nullptr;
    }
    FanoutScope(const FanoutScope&) = delete;
    FanoutScope& operator=(const FanoutScope&) = delete;
};
// *_run_multi, around one block: hands the role to the host-pointer call the thread makes next and
// settles it if that call never got as far as taking it (an empty block, an argument error)
struct FanoutArm {
    FanoutArm(DataFanout* f, int role)
    {
        t_fanout = (f && role != DataFanout::NONE) ? f : nullptr;
        t_fanout_role = t_fanout ? role : (int)DataFanout::NONE;
    }
    ~FanoutArm()
    {
        if (t_fanout) {
            if (t_fanout_role == DataFanout::SOURCE) t_fanout->cancel();
            else if (t_fanout_role == DataFanout::PEER) t_fanout->peer_done();
        }
        t_fanout = nullptr;
        t_fanout_role = DataFanout::NONE;
    }
    FanoutArm(const FanoutArm&) = delete;
    FanoutArm& operator=(const FanoutArm&) = delete;
};
// one hand-over at a time per process (*_run_multi try-locks it; a second concurrent multi-device call
// simply uploads from the host on every device): a source waits for its peers under its device's call
// mutex, and two hand-overs with crossed device orders would wait for each other
extern std::mutex g_fanout_mutex;

// Enqueues "the day of data -> d_dst on this thread's current device" on `stream`: from the hand-over when
// the thread is a peer of one that got published (device -> device), from `host` otherwise; a source
// publishes behind its upload.  hipSuccess or the failing call's error (`what` names it).
hipError_t fanout_upload(FanoutScope& scope, DeviceContext* ctx, void* d_dst, const void* host, size_t bytes,
                         hipStream_t stream, const char** what);
// The two halves of it, for a caller that uploads in pieces: a PEER's device-to-device copy (true = enqueued
// or failed with *err set; false = not a peer, or the hand-over was cancelled: upload from the host), and a
// SOURCE's publication behind the last piece it enqueued on `stream` (a no-op for anybody else).
bool fanout_peer_copy(FanoutScope& scope, DeviceContext* ctx, void* d_dst, size_t bytes, hipStream_t stream,
                      hipError_t* err, const char** what);
hipError_t fanout_publish(FanoutScope& scope, DeviceContext* ctx, const void* d_src, hipStream_t stream);

// memcpy on a few host threads (large blocks) -- dst / src pageable or pinned host memory
void parallel_copy(char* dst, const char* src, size_t bytes);
// every host thread of the copy pool has finished what it was doing (context.hip: CopyPool)
void copy_pool_quiesce();
// device memory -> the caller's pageable array through the pinned pieces; blocks until `host` holds the bytes
hipError_t staged_download(DeviceContext* ctx, void* host, const void* d_src, size_t bytes, hipStream_t stream);

// rows x [c0, c1) of a (rows, N) float32 array in PAGEABLE host memory -> the same rows and samples of the
// device array `d_dst`, through the context's two pinned pieces: host threads fill one piece (row segments
// packed back to back) while the other is in flight on `stream` (hipMemcpy2DAsync, truly asynchronous from
// pinned memory).  Why not hand the pageable pointer to the runtime: on FIRST contact with a host region it
// page-locks the whole enclosing range before it copies -- measured 18 GB/s for a contiguous gigabyte, 1.5-5
// GB/s for a strided piece of a larger array (tools/ubench/h2d_overlap.hip, profiles/r05_h2d_overlap.txt) --
// and a new day of data is a new allocation; through pinned pieces the same bytes move at 53 GB/s whatever
// the runtime has seen before, and the call returns when the last piece is IN FLIGHT (the pieces of earlier
// calls' D2H use the same buffers: callers serialise through call_mutex).  Needs reserve_pinned() first.
hipError_t staged_upload_rows(DeviceContext* ctx, float* d_dst, const float* host, size_t rows, size_t N, size_t c0,
                              size_t c1, hipStream_t stream);

// The context of `device`, created on first use (streams and events under the registry mutex, with
// the device bound).  nullptr + error text when the runtime refuses.
DeviceContext* device_context(int device);

// A side stream of this device for a new plan (round-robin over the context's pool; nullptr + error
// text on failure).
hipStream_t device_side_stream(int device);

}  // namespace bpmf
