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
#include <mutex>

namespace bpmf {

struct DeviceContext {
    int device = 0;
    std::mutex call_mutex;            // one host-pointer call on this device at a time
    hipStream_t s_run = nullptr;      // kernels and H2D of the call (non-blocking: never the null stream)
    hipStream_t s_copy = nullptr;     // D2H of finished batches
    hipStream_t s_side = nullptr;     // edge tiles of the backprojection beside the interior kernels
    hipEvent_t ev_batch[2] = {nullptr, nullptr};
    hipEvent_t ev_piece[2] = {nullptr, nullptr};
    // grow-only buffers, valid while call_mutex is held
    char* dev_buf = nullptr;
    std::atomic<size_t> dev_cap{0};
    char* pinned[2] = {nullptr, nullptr};
    std::atomic<size_t> pinned_cap{0};
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
};

// memcpy on a few host threads (large blocks) -- dst / src pageable or pinned host memory
void parallel_copy(char* dst, const char* src, size_t bytes);

// The context of `device`, created on first use (streams and events under the registry mutex, with
// the device bound).  nullptr + error text when the runtime refuses.
DeviceContext* device_context(int device);

// The side stream plans of this device share (nullptr + error text on failure).
hipStream_t device_side_stream(int device);

}  // namespace bpmf
