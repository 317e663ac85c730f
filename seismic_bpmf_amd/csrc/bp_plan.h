// Shared between bp.hip (plan builder, generic beam kernels, C ABI) and bp_fast.hip (the
// interior-tile production kernel): device-side plan records and the host-side plan object.
#pragma once
#include "common.h"
#include "../../include/bpmf_hip.h"

#include <mutex>

namespace bpmf {

constexpr int BP_THREADS = 256;
constexpr size_t BP_LDS_MAX = 160 * 1024;

struct BpGroup {  // one LDS residency: a run of sources and the staging work they need
    int first_src, n_src, first_chunk, n_chunk;
};
struct BpChunk {  // <= BP_THREADS consecutive floats of one prestacked (station, phase) row
    int row;   // row of U (s * P + p)
    int gofs;  // first sample, relative to the tile start t0 (window moveout origin + x0)
    int dst;   // LDS float offset
    int n;     // floats in this chunk
};
struct BpSource {
    int id, tmin, tmax, nterm;  // global id, extreme used moveouts, terms padded to the chunk (0 = unused)
};

// ---- interior-tile fast path (bp_fast.hip) ----
// A group's sources are listed once more, partitioned into RUNS of equal (padded) station count,
// ascending id inside a run.  A source is `nparts` PARTS of `tp` stations (tp even, <= 16, <= 24 at
// tile 128; one part up to 16 stations); one record of `rec_dw` dwords per part:
//   uniform weights : [id, weight, addrP_0, addrS_0, addrP_1, addrS_1, ...]   LDS byte addresses
//   per-station     : [id, 0, offs_0, w_0, offs_1, w_1, ...]   offs = float offsets P | S << 16
// Padding stations address the zero slab (offset 0) with the source's weight (or weight 0).
// Records of a run are laid out so that a wave's consecutive parts are 16 records apart:
// record (first_rec + ((m / 16) * nparts + part) * 16 + m % 16) for the m-th source of the run.
struct BpRun { int first_rec, n_src, tp, nparts; };
struct BpFastGroup { int first_run, n_run, first_win, n_win; };
// Two-residency groups (sources with 33-64 stations at tile 256, see bp_fast.hip): flags in n_run
constexpr int BPF_GROUP_LOAD = 1 << 16, BPF_GROUP_STORE = 1 << 17;
constexpr int BPF_HALVES_SLOTS = 9;      // sources per wave of a multi-residency group (16 waves: 144 per group)
// one staged window of the fast path: `len` floats of row `row` starting at t0 + gofs -> LDS float
// offset dst (len a multiple of 4, dst a multiple of 4: the LDS-DMA copies move 16 bytes per lane)
struct BpWindow { int row, gofs, dst, len; };
// LDS floats [0, BPF_ZERO_SLAB) are the zero slab (padding terms read it), floats
// [BPF_DESC_OFS, BPF_DESC_OFS + 4 * BPF_DESC_MAX) hold the NEXT group's window descriptors
// (copied there while the current group is computed)
constexpr int BPF_ZERO_SLAB = 512, BPF_DESC_OFS = 512, BPF_DESC_MAX = 256;

// One station-count class of sources with its own tile (device tables of bp_fast.hip)
struct BpFastClass {
    int tile = 512;              // 512 / 256 / 128 time samples per workgroup
    bool uniform = false;        // every source's non-zero weights are equal: ready-made addresses
    bool halves = false;         // groups of <= 144 sources computed in 2-4 LDS residencies (<= 20 stations each)
    int n_pass = 1;              // halves: consecutive entries of d_groups per group of sources
    int rec_dw = 0;              // dwords per record
    int n_groups = 0;
    int desc_waves = 1;          // waves that copy the next group's window descriptors
    size_t lds_bytes = 0;
    size_t n_sources = 0;
    int max_stations = 0;        // diagnostics
    BpFastGroup* d_groups = nullptr;
    BpRun* d_runs = nullptr;
    BpWindow* d_wins = nullptr;
    int* d_recs = nullptr;
};
constexpr int BPF_MAX_CLASSES = 3;

}  // namespace bpmf

struct bpmf_bp_plan {
    int device = 0;
    size_t K = 0, S = 0, P = 0;
    int tpt = 2;           // time samples per thread -> tile = BP_THREADS * tpt
    int chunk = 4;         // terms gathered side by side
    int NT = 4;            // padded number of (station, phase) terms per source
    int n_groups = 0;
    size_t lds_bytes = 0;  // largest group
    bool dual = false;     // dual (shifted) windows: every term offset is even
    int id_offset = 0;
    double mean_group = 0; // diagnostics
    bpmf::BpGroup* d_groups = nullptr;
    bpmf::BpChunk* d_chunks = nullptr;
    bpmf::BpSource* d_srcs = nullptr;
    int* d_off = nullptr;
    float* d_beta = nullptr;
    int ntv = 0;                 // > 0: uniform-VGPR fast path with NTV padded terms
    int wps = 1;                 // wave-per-source kernel (needs ntv > 0 and tile 512)
    int nsv = 0;                 // > 0: packed per-station records (P == 2), NSV stations padded
    int4* d_recs = nullptr;      // [K, nsv/2]
    int4* d_hdr2 = nullptr;      // [K] headers with the station count in .w
    void* d_termsv = nullptr;    // [K, ntv] BpTermV (bp.hip)
    // interior-tile fast path: 1-3 station-count classes of sources, each with its own tile
    bool fast = false;
    int n_classes = 0;
    bpmf::BpFastClass cls[bpmf::BPF_MAX_CLASSES];
    bool fast_shares_generic = false;  // the single class was built from the generic (dual) plan: the
                                       // edge tiles run the 8-byte-gather flavour of the generic kernel
    int tmin_all = 0, tmax_all = 0;   // extreme used moveouts over all sources
    // no LDS plan exists for this grid (a source's windows exceed the LDS, or > 256 terms per source):
    // bp_direct.hip gathers from global memory along compact per-source term lists
    bool direct = false;
    int4* d_dhdr = nullptr;            // [K] {any station used, tmin, tmax, -}
    long long* d_dfirst = nullptr;     // [K + 1] first term of every source
    int4* d_dterms = nullptr;          // {row, moveout, weight bits, -}: stations ascending, phases inside
    // the few edge tiles of a day run the general kernel on a side stream, beside the interior
    // kernel (fork / join through the two events): a serial launch of 3-6 workgroups would add the
    // full duration of one tile (5 ms at cfg3) to every call.  A plan serves one call at a time.
    // The stream belongs to the device (context.h: one per device and process, shared by its plans, never
    // destroyed); the events are the plan's own.
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // Two host threads may run the same resident plan on different streams: the fork / join
    // sequence on the shared side stream is enqueued under this mutex, so that the sequences of
    // two calls never interleave (each call's edge tiles stay ordered behind ITS prestack kernel).
    mutable std::mutex enqueue_mutex;
};

namespace bpmf {
// bp.hip, option bp.compat_first_computed: set by bpmf_bp_run_multi on the thread that runs a device's
// share -- the share keeps -inf where it computed no beam, and the host finishes (0, first id) after
// the merge of all shares (a finished share could not be told from a real 0)
extern thread_local bool t_bp_defer_finish;
// bp_fast.hip: running (max, arg-max) over the sources of one class for its tiles [tile_lo, tile_hi)
// (units of fc.tile samples), every one of which lies inside [-tmin_all, N - tmax_all) (no bounds
// test per source).
int launch_beam_fast(const BpFastClass& fc, int id_offset, const float* U, size_t N, long long tile_lo,
                     long long tile_hi, hipStream_t stream, float* beam, int32_t* arg,
                     int n_split = 1, long long split_stride = 0, float best0 = 0.0f);
// bp_direct.hip: the whole series for a plan without LDS windows (pl->direct)
int direct_split_count(const bpmf_bp_plan* pl, size_t N);
int launch_beam_direct(const bpmf_bp_plan* pl, const float* U, size_t N, int oob, int reduce,
                       hipStream_t stream, float* beam, int32_t* arg, int n_split, long long split_stride,
                       float best0);
}
