// Backprojection, production kernel for INTERIOR tiles (round 2).
//
// Same computation and data flow as bp_beam_wps2_kernel<16, NSV, *, MAX, true, true> of bp.hip
// (one 16-wave workgroup per CU owns a tile of 512 time samples and walks all sources; dual LDS
// windows, ds_read_b64 gathers, v_pk_fma_f32 with the weight as an SGPR-pair operand, group-local
// running max), restricted to what the profile of round 1 showed the time goes to:
//
//   * only tiles that lie inside [-tmin_all, N - tmax_all): no source can leave the trace, so the
//     per-source strict-bounds arithmetic (20 SALU + two 64-bit VALU compares per source) is gone.
//     The first/last few tiles of a day run the general kernel (bp.hip launches both).
//   * a group's sources come as RUNS of equal station count: the straight-line gather body is
//     selected once per run, not per source (no compare/branch chain in the source loop).
//   * records hold ready-made LDS byte addresses when the source weights are uniform per source
//     (0/1 or row-normalised weights -- what Beamformer.set_weights_sources produces without
//     density weighting): a unit's address is ONE v_add, no unpacking.
//   * THE RECORD OF THE NEXT SOURCE IS NEVER WAITED FOR WITH ITS LATENCY EXPOSED.  Round 1 kept a
//     source's record in SGPRs and fetched the next one behind an lgkmcnt(0) at the source boundary
//     (scalar loads share lgkmcnt with the LDS and return out of order): every source paid a drain,
//     a scalar-cache round trip and a pipeline refill (67 % of the ds_read_b64 rate).  Round 2 moved
//     the record into VGPRs, refilled quad by quad with wave-uniform VECTOR loads (their own in-order
//     counter, vmcnt), the first three units of the next source issued while the last three of the
//     current one are in flight: no drain at all, 70-71 % -- but every such load costs 16 cycles of
//     the vector memory path (64 lanes x 16 bytes, however uniform the address), per group more than
//     the gathers cost the LDS (cycle counters of round 3, profiles/r03_bp_fast_phase_cycles.txt).
//     Round 3 (walk_g / walk_s): SGPR records again, TWO buffers -- the next record is requested with
//     s_load_dwordx8 at the START of the current part and waited for once, where the look-ahead first
//     needs it; that lgkmcnt(0) drains the wave's own gathers, a bubble the other 15 waves cover, and
//     the scalar-cache round trip is long over.  73-75 % at tile 512, +6-9 % at tile 256, and the
//     20-28 VGPRs of the ring are free.  Records of more than 16 stations (tile 128) keep the VGPR
//     ring (walk): the max / arg-max update of a source runs under the next source's gathers in
//     every variant.
//   * the record pointer advances by a constant (tables are padded by one round of records, so the
//     prefetch of the source after the last needs no clamp): 2 SALU per source.
//   * the accumulators start from the fma's own constant-0 addend (no zero-init moves), and the
//     max / arg-max update is hand-scheduled: 8 v_cmp into 8 SGPR pairs, then 16 v_cndmask -- no
//     VALU-writes-SGPR wait states between a compare and its selects.
//
//
// Round 3: DENSE STATION WEIGHTS.  The dual windows of n weighted stations need 2 n rows x 2 copies
// x (tile + moveout spread) floats of LDS: at tile 512 that ends at n = 16, and rounds 1-2 sent a
// whole grid to the 4-byte-gather kernels as soon as ONE source had 17 stations.  Now (a) the
// kernel is a template on the samples per lane -- tile 512 / 256 / 128, one term of a source
// costing 4 / 2 / 1 ds_read_b64 per lane, a "unit" of the gather ring always being 4 gathers --
// and (b) a source may span several records ("parts" of <= 16 stations) whose accumulators are
// carried, the max update following the last part.  bp.hip sorts the sources of a grid into
// classes by station count and gives every class the largest tile its windows fit.
//
// Arithmetic (and therefore every output bit) is that of oracle/bpmf_oracle.c:bp_cpu: per source
// an fmaf chain over (station outer, phase inner) starting from +0, strict > keeps the lowest id.
#include "bp_plan.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace bpmf {

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
#ifndef BPF_SREC_MAX_TP_V
#define BPF_SREC_MAX_TP_V 16
#endif
constexpr int BPF_SREC_MAX_TP = BPF_SREC_MAX_TP_V;   // longer records keep the VGPR ring (SGPR budget)
// A record {id, weight, NTERM LDS byte addresses} in SGPRs: N8 octets and a tail of 2 (NTERM = 2 mod 8
// leaves 2 dwords) or one more octet (the record table is padded: reading past the record is harmless)
template <int RD>
struct BpSRec {
    static constexpr int N8 = RD / 8, TAIL2 = (RD % 8) == 2, NV = TAIL2 ? N8 : (RD + 7) / 8;
    i32x8 v[NV ? NV : 1];
    i32x2 t;
    template <int I>
    __device__ __forceinline__ int dw() const
    {
        if constexpr (I < 8 * NV) return v[I >> 3][I & 7];
        else return t[I & 1];
    }
};
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define BPF_RD64(dst, addr, o) asm volatile("ds_read_b64 %0, %1 offset:" #o : "=v"(dst) : "v"(addr))
// acc = w * x + acc, w = the high dword of a register pair {id or offsets, weight}
#define BPF_PKFMA(acc2, sp2, x2)                                                       \
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]"        \
                 : "+v"(acc2) : "s"(sp2), "v"(x2))
// acc = w * x + 0  (first unit of a source)
#define BPF_PKFMA0(acc2, sp2, x2)                                                      \
    asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]"         \
                 : "=v"(acc2) : "s"(sp2), "v"(x2))
// 16 / 8 bytes from a wave-uniform address (SGPR base + zero VGPR offset): every lane receives the
// same dwords.  Vector loads return in order and count in vmcnt, independently of the LDS gathers.
#define BPF_LOADX4(dst, vz, sptr, o) \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #o : "=v"(dst) : "v"(vz), "s"(sptr))
#define BPF_LOADX2(dst, vz, sptr, o) \
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:" #o : "=v"(dst) : "v"(vz), "s"(sptr))

// ---- cycle accounting (tools/phase/build_phase_lib.py builds a second library with -DBPMF_PHASE_CYCLES;
// the shipping library carries none of it).  s_memtime at the phase boundaries of a group entry, summed
// per wave over all entries and, at the end of the kernel, over all waves of the launch into
// g_bpf_phase[age group of the wave (wave / 4)][{barrier wait, descriptor read + copy issue, copy latency
// + second barrier (single-residency kernels), records + gathers, entries}] -- read and reset through
// bpmf_phase_read_bp (exported by such a build only; tools/phase/bp_phase.py).
#ifdef BPMF_PHASE_CYCLES
__device__ unsigned long long g_bpf_phase[4][8];
#define BPF_PHASE_DECL unsigned long long ph_last_ = 0, ph_acc_[4] = {0, 0, 0, 0}; unsigned ph_n_ = 0;
#define BPF_PHASE_START() asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_last_) :: "memory")
#define BPF_PHASE(i)                                                                               \
    do {                                                                                           \
        unsigned long long t_;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");              \
        ph_acc_[i] += t_ - ph_last_;                                                               \
        ph_last_ = t_;                                                                             \
    } while (0)
#else
#define BPF_PHASE_DECL
#define BPF_PHASE_START() do {} while (0)
#define BPF_PHASE(i) do {} while (0)
#endif

constexpr int BPF_WPB = 16;             // waves per workgroup = per CU (the dual windows take the LDS)
constexpr int BPF_THREADS = 64 * BPF_WPB;

// ---- the rolling-record pipeline in numbers (all compile-time) ----
// A PART is one record: `TP` stations = 2 TP (station, phase) terms = Q = TP / 2 quads of 4 body
// dwords.  A source is `nparts` consecutive parts with the accumulators carried from part to part
// (1 part for <= 16 stations; 2-4 parts for dense weights).  A lane owns TPW samples of the tile
// (TPW / 2 pairs), so one term costs TPW / 2 ds_read_b64; a UNIT is the bundle of 4 gathers that
// travels together through the ring of 4: TPU = 8 / TPW terms (tile 512: 1 term, 256: 2, 128: 4).
// Quad q of the record is consumed -- its addresses issued, its weights multiplied -- when unit
// bpf_ur(q) has been accumulated, and is then re-loaded with the NEXT part's quad q.
constexpr int bpf_ur(int q, int tpu) { return (4 * q + 3) / tpu; }
// refills of the current part already issued when step u starts (those behind units < u), quads > qq only
constexpr int bpf_refills_before(int u, int qq, int Q, int tpu)
{
    int n = 0;
    for (int j = qq + 1; j < Q; ++j)
        if (bpf_ur(j, tpu) <= u - 1) ++n;
    return n;
}

template <class F, int... I>
__device__ __forceinline__ void bpf_for_each(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}

// Option bp.slot_prio (round 5): the 16 waves of a workgroup start a group together and do the same work, but the
// hardware serves the oldest waves first -- they reach the group's barrier thousands of cycles before the youngest,
// which then gather alone at a fraction of the LDS rate (profiles/r04_bp_fast_phase_cycles.txt: the oldest waves
// wait 13 758 cycles of a 35 185-cycle entry).  A wave therefore LOWERS its issue priority as it gets through its
// parts of the run -- 3, 2, 1, 0 by quarters -- and the waves arrive together.  Wave-uniform, a compare per pair
// of parts.  (The control -- waves ahead go FIRST -- measured no gain: profiles/r05_bp_slot_prio.txt.)
struct BpfYield {
    // (two SGPRs of state: these kernels sit at the 100-SGPR limit with two records in flight -- a first version
    // with three thresholds and the mode kept live made the compiler spill)
    int left, level;
    // mode 1: by progress (above).  mode 2: ROTATING -- the wave takes the next level in front of every pair of
    // parts, starting from its place on its SIMD (a workgroup's waves go to the SIMDs round-robin: wave >> 2), so
    // that the four waves of a SIMD hold four different priorities at any time and each is favoured a quarter of
    // the time.  Measured (profiles/r05_bp_slot_prio.txt): rotating lifts the ISOLATED gather loop from 0.75 to 0.78 (static
    // per-wave priorities: nothing), but in the kernel the progress schedule wins -- 0.771 against 0.758 at cfg3,
    // 0.666 against 0.648 with 40 stations -- because it also brings the waves to the barrier together; a
    // staggered progress schedule (older waves yield earlier) measured equal to the plain one.
    __device__ __forceinline__ BpfYield(int mode, int wave) : left(mode == 1 ? 0 : (mode == 2 ? -1 : 0x7fffffff)), level(mode == 2 ? (wave >> 2) & 3 : 3) {}
    __device__ __forceinline__ void set() const
    {
        if (level == 3) __builtin_amdgcn_s_setprio(3);
        else if (level == 2) __builtin_amdgcn_s_setprio(2);
        else if (level == 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
    // in front of the parts it, it + by of a run of n_it
    __device__ __forceinline__ void at(int n_it, int by)
    {
        if (left < 0) { set(); level = (level + 1) & 3; return; }          // rotating
        if (left > 0) { left -= by; return; }
        set();
        left = level > 0 ? ((n_it + 3) >> 2) - by : 0x7fffffff;
        level = level > 0 ? level - 1 : 0;
    }
};

// TPW: samples per lane (8 / 4 / 2 -> tile 512 / 256 / 128).  The smaller tiles exist for dense
// station weights: the dual windows of 2 n rows must fit 160 KB -- n <= 16 at tile 512, ~28 at 256,
// ~48 at 128 (bp.hip picks the tile per station-count class of sources).
//
// HALVES (tile 256 only): sources with 33-64 weighted stations.  Their dual windows do not fit the
// LDS at tile 256 (80 rows x 2 copies x 256 floats = 164 KB), and tile 128 stops at 0.35 of the gather
// rate.  So a group of at most 144 sources (BPF_HALVES_SLOTS = 9 per wave) is computed in two to four
// LDS RESIDENCIES at tile 256: the first stages the windows of every source's first <= 20 stations and
// leaves the partial beams of the wave's 9 sources in registers (`carry`, 36 VGPRs, statically indexed:
// the source loop is unrolled over the slots), the next ones stage the following stations and continue
// the same fmaf chains, the last one updates the running maximum.  The plan lists the residencies as
// consecutive groups with the same sources in the same order, every source as exactly two records per
// residency (short groups padded with records of weight 0 and id -1: no branch on the slot count);
// BPF_GROUP_STORE / BPF_GROUP_LOAD in the group's run count tell the kernel where it is.
// Registers decide the shape: 128 VGPRs at the 4 waves per SIMD the LDS rate needs.  With the records
// in VGPRs (rounds of this kernel before walk_s) six carried sources fit: 0.52; with the records in
// SGPRs nine: 0.61.
template <bool UNI, int TPW, bool HALVES = false>
__global__ __launch_bounds__(BPF_THREADS) void bp_beam_fast_kernel(
    const float* __restrict__ U, long long N, const BpFastGroup* __restrict__ groups, int n_groups,
    const BpRun* __restrict__ runs, const BpWindow* __restrict__ wins, const int* __restrict__ recs,
    int rec_dw, int id_offset, long long tile_lo, long long n_tiles, float* __restrict__ out_beam,
    int* __restrict__ out_arg, int desc_waves, long long split_stride, float best0, int n_pass, int slot_prio)
{
    // short series: workgroup (tile, y) walks the groups [n_groups y / Y, n_groups (y + 1) / Y) and
    // writes its partial maxima to out + y * split_stride (bp.hip: bp_split_count, bp_merge_splits_kernel);
    // a multi-residency class is cut between groups of sources (n_pass consecutive entries each)
    const int g_lo = (int)((long long)(n_groups / n_pass) * blockIdx.y / gridDim.y) * n_pass;
    const int g_hi = (int)((long long)(n_groups / n_pass) * (blockIdx.y + 1) / gridDim.y) * n_pass;
    out_beam += (size_t)blockIdx.y * (size_t)split_stride;
    out_arg += (size_t)blockIdx.y * (size_t)split_stride;
    extern __shared__ float lds[];
    static_assert(TPW == 8 || TPW == 4 || TPW == 2, "tile 512, 256 or 128");
    static_assert(!HALVES || TPW == 4, "two-residency groups run at tile 256");
    constexpr int NSLOT = BPF_HALVES_SLOTS;      // HALVES: sources per wave and group
    f32x2 carry[HALVES ? NSLOT : 1][HALVES ? TPW / 2 : 1];       // partial beams between the residencies
    constexpr int TILE = 64 * TPW, WPB = BPF_WPB, NTHREADS = BPF_THREADS;
    constexpr int TPU = 8 / TPW;      // terms per unit
    constexpr int RPT = TPW / 2;      // ds_read_b64 per term
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile order (see bp_beam_wps2_kernel): every XCD stages its own run of tiles
    const long long tiles_per_xcd = (gridDim.x + 7) >> 3;
    const long long tile = (long long)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const long long t0 = (tile_lo + tile) * TILE;
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    const unsigned v_base = (unsigned)(size_t)lds + (unsigned)lane * 8u;
    auto slot_x = [&](int j) { return 128 * (j >> 1) + 2 * lane + (j & 1); };

    float best[TPW];
    int arg[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) { best[j] = best0; arg[j] = id_offset; }   // 0, or -inf (bp.compat_first_computed)
    for (int x = tid; x < BPF_ZERO_SLAB; x += NTHREADS) lds[x] = 0.0f;  // the zero slab
    const long long rec_stride = (long long)rec_dw * 4 * WPB;   // bytes between a wave's consecutive parts

    // Window descriptors of a group travel through a 4 KB slab of LDS: waves 0-3 copy the NEXT
    // group's descriptors there (LDS-DMA, 16 bytes per lane) right after the staging barrier of the
    // current group, so that at the next group boundary they are read from LDS instead of paying a
    // memory round trip in front of the window copies.
    auto prefetch_descriptors = [&](int first_win) {
        if (wv < desc_waves) {
            const BpWindow* src = wins + first_win + 64 * wv + lane;     // the table is padded by BPF_DESC_MAX
            float* dst = lds + BPF_DESC_OFS + 256 * wv;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    if (g_hi > g_lo) prefetch_descriptors(groups[g_lo].first_win);

    BPF_PHASE_DECL
    BPF_PHASE_START();
    for (int g = g_lo; g < g_hi; ++g) {
        const BpFastGroup grp = groups[g];
        __syncthreads();  // previous group's gathers are done, this group's descriptors are in LDS
        BPF_PHASE(0);
        // ---- staging by LDS-DMA: a window (tile + moveout spread floats of one prestacked row) goes
        // global -> LDS in pieces of 256 floats, ONE instruction per piece and wave (16 bytes per
        // lane, destination = wave-uniform base + 16 * lane; an unaligned global source is fine),
        // without staging registers and without ds_write_b128 (13 cycles each).  Wave w takes the
        // windows w, w + 16, ...; all its copies are in flight together.  The register-staged
        // version of round 1 paid two dependent round trips per 8 chunks per wave: 5.5 % of the
        // kernel at cfg3, this one 2 %.  The copies count in vmcnt; __syncthreads() waits for them
        // (vmcnt(0)) before the barrier.
        {
            const i32x4* dsc = (const i32x4*)(lds + BPF_DESC_OFS);
            for (int wi = wv; wi < grp.n_win; wi += WPB) {
                const i32x4 d = dsc[wi];                       // {row, first sample relative to t0, LDS float offset, floats}
                const int row = __builtin_amdgcn_readfirstlane(d[0]), gofs = __builtin_amdgcn_readfirstlane(d[1]);
                const int dst0 = __builtin_amdgcn_readfirstlane(d[2]), len = __builtin_amdgcn_readfirstlane(d[3]);
                // interior tile: every sample of every window lies inside [0, N)
                const float* src = U + (size_t)row * (size_t)N + (t0 + gofs) + 4 * lane;
                for (int x0 = 0; x0 < len; x0 += 256) {
                    if (x0 + 4 * lane < len)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + x0),
                                                         (__attribute__((address_space(3))) void*)(lds + dst0 + x0), 16, 0, 0);
                }
            }
        }
        BPF_PHASE(1);
        __syncthreads();
        if (g + 1 < g_hi) prefetch_descriptors(groups[g + 1].first_win);
        BPF_PHASE(2);

        const bool g_load = HALVES && (grp.n_run & BPF_GROUP_LOAD) != 0;     // second residency: continue the chains
        const bool g_store = HALVES && (grp.n_run & BPF_GROUP_STORE) != 0;   // first residency: park them, no max update
        for (int rr = 0; rr < (grp.n_run & 0xffff); ++rr) {
            const BpRun run = runs[grp.first_run + rr];
            const int n_mine = run.n_src > wv ? (run.n_src - wv + WPB - 1) / WPB : 0;   // sources of this wave
            if (n_mine == 0) continue;
            const int nparts = run.nparts;
            const int* p_first = recs + ((long long)run.first_rec + wv) * rec_dw;
            // group-local running max of this run: sources arrive by ascending id, so a plain
            // strict > keeps the lowest id on ties; the full tie rule merges the run into best/arg
            // (HALVES has no registers to spare for it: the tile's running maximum is updated directly,
            // with the full tie rule)
            float bestg[HALVES ? 1 : TPW];
            int argg[HALVES ? 1 : TPW];
            if constexpr (!HALVES) {
#pragma unroll
                for (int j = 0; j < TPW; ++j) { bestg[j] = -INFINITY; argg[j] = 0x7fffffff; }
            }

            // TP stations per part; MULTI = false: every source is ONE part (compile-time: the
            // accumulators start from the fma's constant-0 addend and the max update follows every
            // part); MULTI = true: `nparts` parts per source, accumulators carried, the update
            // behind the last part under a wave-uniform branch.
            auto walk = [&](auto tp_c, auto multi_c) {
                constexpr int TP = decltype(tp_c)::value;
                constexpr bool MULTI = decltype(multi_c)::value;
                constexpr int NTERM = 2 * TP, NU = NTERM / TPU, Q = NTERM / 4;
                // (the VGPR-record walk: records of more than BPF_SREC_MAX_TP stations only -- tile 128 with
                // 20 or 24 stations per part; everything else keeps its records in SGPRs, walk_g / walk_s)
                static_assert(!HALVES, "the multi-residency kernel has its own walk");
                // units in flight ahead of the one being accumulated, and the ring that holds them
                constexpr int AH = 3, RING = AH + 1;
                static_assert(TP >= 4 && TP % 2 == 0 && TP <= 24 && NTERM % TPU == 0 && Q <= 12, "4..24 stations per part, even");
                static_assert(NU >= AH + 1 && bpf_ur(((AH - 1) * TPU) >> 2, TPU) <= NU - 2,
                              "the next part's first units must find their quads requested");
                // The record of the CURRENT part, rolling: quad q = body dwords 4q .. 4q+3 (the
                // addresses -- or {offsets, weight} pairs -- of terms 4q .. 4q+3).  Once the unit that
                // holds term 4q+3 has been accumulated, quad q is re-loaded with the next part's quad q.
                i32x4 R[Q];
                i32x2 h_cur, h_next;          // {id, weight}
                const int* p = p_first;
                BPF_LOADX2(h_cur, vzero, p, 0);
#define BPF_LOADQ(q) \
    if constexpr ((q) < Q) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(R[(q) < Q ? (q) : 0]) : "v"(vzero), "s"(p), "n"(8 + 16 * (q)))
#define BPF_LOADQC(q) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(R[q]) : "v"(vzero), "s"(p), "n"(8 + 16 * (q)))
                BPF_LOADQ(0); BPF_LOADQ(1); BPF_LOADQ(2); BPF_LOADQ(3);
                BPF_LOADQ(4); BPF_LOADQ(5); BPF_LOADQ(6); BPF_LOADQ(7);
                BPF_LOADQ(8); BPF_LOADQ(9); BPF_LOADQ(10); BPF_LOADQ(11);
                // vmcnt(0) with every register of the record as an in/out operand: nothing that uses
                // them can be scheduled above the wait (tools/check_inflight.py checks the listing)
#define BPF_VMWAIT_ALL()                                                                  \
    do {                                                                                  \
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(h_cur));                                 \
        _Pragma("unroll") for (int qi = 0; qi < Q; ++qi) asm volatile("s_waitcnt vmcnt(0)" : "+v"(R[qi])); \
    } while (0)
                BPF_VMWAIT_ALL();
                // ring of 4 units in flight, 4 gathers each.  Unit w of a part sits in slot (w + PH) & 3:
                // the ring runs on across parts, so when a part has NU = 2 mod 4 units the phase PH
                // alternates 0, 2, 0, ... from part to part (the loop below is then unrolled by two).
                f32x2 X[RING][4];
                // LDS byte address of term `tm` of the part whose record R holds
#define BPF_TERM_ADDR(tm)                                                                      \
    (UNI ? v_base + (unsigned)R[(tm) >> 2][(tm) & 3]                                           \
         : v_base + (((((tm) & 1) ? ((unsigned)R[(tm) >> 2][2 * (((tm) >> 1) & 1)] >> 16)      \
                                  : ((unsigned)R[(tm) >> 2][2 * (((tm) >> 1) & 1)] & 0xffffu))) << 2))
                // unit w: TPU terms x RPT gathers of 8 bytes per lane (plain C++ adds for the
                // addresses: an inline-asm add would cost a hazard s_nop before the reads)
#define BPF_ISSUE_U(w, sl)                                                                     \
    {                                                                                          \
        if constexpr (TPU == 1) {                                                              \
            const unsigned a_ = BPF_TERM_ADDR(w);                                              \
            BPF_RD64(X[sl][0], a_, 0); BPF_RD64(X[sl][1], a_, 512);                  \
            BPF_RD64(X[sl][2], a_, 1024); BPF_RD64(X[sl][3], a_, 1536);              \
        } else if constexpr (TPU == 2) {                                                       \
            const unsigned a_ = BPF_TERM_ADDR(2 * (w)), b_ = BPF_TERM_ADDR(2 * (w) + 1);       \
            BPF_RD64(X[sl][0], a_, 0); BPF_RD64(X[sl][1], a_, 512);                  \
            BPF_RD64(X[sl][2], b_, 0); BPF_RD64(X[sl][3], b_, 512);                  \
        } else {                                                                               \
            const unsigned a_ = BPF_TERM_ADDR(4 * (w)), b_ = BPF_TERM_ADDR(4 * (w) + 1);       \
            const unsigned c_ = BPF_TERM_ADDR(4 * (w) + 2), d_ = BPF_TERM_ADDR(4 * (w) + 3);   \
            BPF_RD64(X[sl][0], a_, 0); BPF_RD64(X[sl][1], b_, 0);                    \
            BPF_RD64(X[sl][2], c_, 0); BPF_RD64(X[sl][3], d_, 0);                    \
        }                                                                                      \
    }
                BPF_ISSUE_U(0, 0) BPF_ISSUE_U(1, 1) BPF_ISSUE_U(2, 2)
                static_assert(AH == 3, "prologue");
                f32x2 ac[RPT];
                if constexpr (MULTI) {
#pragma unroll
                    for (int r = 0; r < RPT; ++r) ac[r] = (f32x2){0.0f, 0.0f};
                }
                int part = 0;                 // MULTI: parts of the current source already accumulated
                const int n_it = MULTI ? n_mine * nparts : n_mine;
                auto part_body = [&](auto ph_c) __attribute__((always_inline)) {
                    constexpr int PH = decltype(ph_c)::value;
                    i32x2 sp_u;
                    // the header of the next part (the table is padded by one round of records: no clamp)
                    p = (const int*)((const char*)p + rec_stride);
                    BPF_LOADX2(h_next, vzero, p, 0);
                    // one step per unit, u a compile-time constant (bpf_for_each: a fold over the
                    // unit indices -- every wait count below is an immediate)
                    auto unit_step = [&](auto uc) __attribute__((always_inline)) {
                        constexpr int u = decltype(uc)::value;
                        // (operands of an asm statement alone do not make a generic lambda capture)
                        (void)&vzero; (void)&p; (void)&R; (void)&h_next; (void)&h_cur; (void)&X; (void)&ac; (void)&sp_u;
                        constexpr int SL_ISSUE = (u + AH + PH) % RING, SL_USE = (u + PH) % RING;
                        // ---- keep three units in flight ahead of unit u.  The unit issued now belongs
                        // to this part (u + 3 < NU) or is one of the first three of the NEXT part.  The
                        // first term of a quad is the first use of that quad's re-load: vector loads
                        // return in order, so "at most n younger loads outstanding" = it has landed.
                        if constexpr (u + AH < NU) {
                            if constexpr (((u + AH) * TPU) % 4 == 0) {
                                // quad qq was requested one part ago, followed by the rest of that part's
                                // refills (Q - 1 - qq), this part's header and the refills behind units < u
                                constexpr int qq = ((u + AH) * TPU) >> 2;
                                constexpr int younger = (Q - 1 - qq) + 1 + bpf_refills_before(u, -1, Q, TPU);
                                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(R[qq]) : "n"(younger));
                            }
                            BPF_ISSUE_U(u + AH, SL_ISSUE)
                        } else {
                            if constexpr (((u + AH - NU) * TPU) % 4 == 0) {
                                // the next part's quad qq, requested in this part behind unit bpf_ur(qq);
                                // younger: this part's refills of the quads > qq issued so far.  The header
                                // of the next part is older than all of them.
                                constexpr int qq = ((u + AH - NU) * TPU) >> 2;
                                static_assert(bpf_ur(qq, TPU) <= u - 1, "quad not requested yet");
                                constexpr int younger = bpf_refills_before(u, qq, Q, TPU);
                                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(R[qq]), "+v"(h_next) : "n"(younger));
                            }
                            BPF_ISSUE_U(u + AH - NU, SL_ISSUE)
                        }
                        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(4 * AH) : "memory");
                        // the weight reaches v_pk_fma_f32 as the high half of an SGPR pair: with three
                        // 64-bit VGPR operands the instruction is register-read bound (measured: the
                        // kernel lost 5 points of the LDS rate with the weight in a VGPR pair)
#pragma unroll
                        for (int k = 0; k < TPU; ++k) {
                            const int tm = u * TPU + k;      // term of the part: station tm / 2, phase tm % 2
                            if constexpr (UNI) {
                                if (tm == 0) { sp_u[0] = 0; sp_u[1] = __builtin_amdgcn_readfirstlane(h_cur[1]); }
                            } else if ((tm & 1) == 0) {
                                sp_u[0] = 0;
                                sp_u[1] = __builtin_amdgcn_readfirstlane(R[tm >> 2][2 * ((tm >> 1) & 1) + 1]);
                            }
                            const i32x2 sp = sp_u;
#pragma unroll
                            for (int r = 0; r < RPT; ++r) {
                                if (!MULTI && tm == 0) BPF_PKFMA0(ac[r], sp, X[SL_USE][k * RPT + r]);
                                else BPF_PKFMA(ac[r], sp, X[SL_USE][k * RPT + r]);
                            }
                        }
                        // ---- quads whose last term this unit held are consumed: refill them
                        if constexpr ((u * TPU + TPU) % 4 == 0) {
                            constexpr int q0 = (u * TPU + TPU) / 4 - 1;
                            static_assert(bpf_ur(q0, TPU) == u, "refill slot");
                            BPF_LOADQC(q0);
                        }
                    };
                    bpf_for_each(unit_step, std::make_integer_sequence<int, NU>{});
                    // ---- max / arg-max update behind a source's last part, strict >: TPW compares, then
                    // 2 TPW selects; the next part's first three units are in flight meanwhile
                    bool last = true;
                    if constexpr (MULTI) { ++part; last = part == nparts; }
                    if (last) {
                        unsigned long long mk[TPW];
#pragma unroll
                        for (int j = 0; j < TPW; ++j)
                            asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(mk[j]) : "v"(ac[j >> 1][j & 1]), "v"(bestg[j]));
#pragma unroll
                        for (int j = 0; j < TPW; ++j) {
                            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(bestg[j]) : "v"(ac[j >> 1][j & 1]), "s"(mk[j]));
                            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(argg[j]) : "v"(h_cur[0]), "s"(mk[j]));
                        }
                        if constexpr (MULTI) {
                            part = 0;
#pragma unroll
                            for (int r = 0; r < RPT; ++r) ac[r] = (f32x2){0.0f, 0.0f};
                        }
                    }
                    h_cur = h_next;
                };
                using ic0 = std::integral_constant<int, 0>;
                using ic2 = std::integral_constant<int, 2>;
                BpfYield yield(slot_prio, wv);
                if constexpr (NU % 4 == 0) {
                    for (int it = 0; it < n_it; ++it) { yield.at(n_it, 1); part_body(ic0{}); }
                } else {
                    static_assert(NU % 4 == 2, "even number of units per part");
                    int it = 0;
                    for (; it + 1 < n_it; it += 2) {
                        yield.at(n_it, 2);
                        part_body(ic0{});
                        part_body(ic2{});
                    }
                    if (it < n_it) part_body(ic0{});   // odd number of parts
                }
                // the three units issued past the wave's last part (they read whatever record follows:
                // valid LDS addresses of some group, or the zero slab) and the last refills.  Every
                // register they load is an operand of the wait: the compiler must keep them apart and
                // untouched until here, although nothing reads them any more.
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                             : "+v"(X[0][0]), "+v"(X[0][1]), "+v"(X[0][2]), "+v"(X[0][3]), "+v"(X[1][0]), "+v"(X[1][1]),
                               "+v"(X[1][2]), "+v"(X[1][3]), "+v"(X[2][0]), "+v"(X[2][1]), "+v"(X[2][2]), "+v"(X[2][3]),
                               "+v"(X[RING - 1][0]), "+v"(X[RING - 1][1]), "+v"(X[RING - 1][2]), "+v"(X[RING - 1][3]), "+v"(h_cur)
                             :: "memory");
                BPF_VMWAIT_ALL();
#undef BPF_ISSUE_U
#undef BPF_TERM_ADDR
#undef BPF_LOADQ
#undef BPF_LOADQC
#undef BPF_VMWAIT_ALL
            };
            // ---- Uniform weights, one LDS residency per group: the same SGPR records as walk_s below (read
            // its comment first) under the general part logic of `walk`: any number of parts per source,
            // runs of a run-time number of sources (the two buffers alternate: the part loop is unrolled
            // by two, which also carries the ring phase of NU = 2 mod 4), the group-local maximum.
            auto walk_g = [&](auto tp_c, auto multi_c) {
                constexpr int TP = decltype(tp_c)::value;
                constexpr bool MULTI = decltype(multi_c)::value;
                constexpr int NTERM = 2 * TP, NU = NTERM / TPU, AH = 3, RING = AH + 1, RD = 2 + NTERM;
                static_assert(NU >= AH + 1 && (NU % 4 == 0 || NU % 4 == 2), "units per part");
                using Rec = BpSRec<RD>;
                Rec A, B;
                const int* p = p_first;
#define BPG_LOAD(dst)                                                                          \
    {                                                                                          \
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(dst.v[0]) : "s"(p));                  \
        if constexpr (Rec::NV > 1) asm volatile("s_load_dwordx8 %0, %1, 0x20" : "=s"(dst.v[Rec::NV > 1 ? 1 : 0]) : "s"(p)); \
        if constexpr (Rec::NV > 2) asm volatile("s_load_dwordx8 %0, %1, 0x40" : "=s"(dst.v[Rec::NV > 2 ? 2 : 0]) : "s"(p)); \
        if constexpr (Rec::NV > 3) asm volatile("s_load_dwordx8 %0, %1, 0x60" : "=s"(dst.v[Rec::NV > 3 ? 3 : 0]) : "s"(p)); \
        static_assert(Rec::NV <= 4, "at most 16 stations per record");                         \
        if constexpr (Rec::TAIL2) asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(dst.t) : "s"(p), "n"(32 * Rec::NV)); \
    }
#define BPG_WAIT(dst)                                                                          \
    {                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dst.v[0]) :: "memory");                     \
        if constexpr (Rec::NV > 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dst.v[Rec::NV > 1 ? 1 : 0]) :: "memory"); \
        if constexpr (Rec::NV > 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dst.v[Rec::NV > 2 ? 2 : 0]) :: "memory"); \
        if constexpr (Rec::NV > 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dst.v[Rec::NV > 3 ? 3 : 0]) :: "memory"); \
        if constexpr (Rec::TAIL2) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dst.t) :: "memory"); \
    }
// LDS byte offset of term `tm`: the dword itself (uniform weights), or a 16-bit float offset out of
// the station's {offs_P | offs_S << 16, weight} pair -- scalar shifts, not VALU
#define BPG_OFF(rec, tm)                                                                       \
    (UNI ? (unsigned)rec.template dw<UNI ? 2 + (tm) : 2>()                                     \
         : ((((tm) & 1) ? ((unsigned)rec.template dw<2 + 2 * ((tm) >> 1)>() >> 16)             \
                        : ((unsigned)rec.template dw<2 + 2 * ((tm) >> 1)>() & 0xffffu)) << 2))
#define BPG_ADDR(rec, tm) (v_base + BPG_OFF(rec, tm))
#define BPG_ISSUE(rec, w, sl)                                                                  \
    {                                                                                          \
        if constexpr (TPU == 1) {                                                              \
            const unsigned a_ = BPG_ADDR(rec, (w));                                            \
            BPF_RD64(X[sl][0], a_, 0); BPF_RD64(X[sl][1], a_, 512);                            \
            BPF_RD64(X[sl][2], a_, 1024); BPF_RD64(X[sl][3], a_, 1536);                        \
        } else if constexpr (TPU == 2) {                                                       \
            const unsigned a_ = BPG_ADDR(rec, 2 * (w)), b_ = BPG_ADDR(rec, 2 * (w) + 1);       \
            BPF_RD64(X[sl][0], a_, 0); BPF_RD64(X[sl][1], a_, 512);                            \
            BPF_RD64(X[sl][2], b_, 0); BPF_RD64(X[sl][3], b_, 512);                            \
        } else {                                                                               \
            const unsigned a_ = BPG_ADDR(rec, 4 * (w)), b_ = BPG_ADDR(rec, 4 * (w) + 1);       \
            const unsigned c_ = BPG_ADDR(rec, 4 * (w) + 2), d_ = BPG_ADDR(rec, 4 * (w) + 3);   \
            BPF_RD64(X[sl][0], a_, 0); BPF_RD64(X[sl][1], b_, 0);                              \
            BPF_RD64(X[sl][2], c_, 0); BPF_RD64(X[sl][3], d_, 0);                              \
        }                                                                                      \
    }
                BPG_LOAD(A)
                BPG_WAIT(A)
                f32x2 X[RING][4];
                BPG_ISSUE(A, 0, 0) BPG_ISSUE(A, 1, 1) BPG_ISSUE(A, 2, 2)
                f32x2 ac[RPT];
                if constexpr (MULTI) {
#pragma unroll
                    for (int r = 0; r < RPT; ++r) ac[r] = (f32x2){0.0f, 0.0f};
                }
                int part = 0;                 // MULTI: parts of the current source already accumulated
                const int n_it = MULTI ? n_mine * nparts : n_mine;
                auto part_g = [&](auto& cur, auto& nxt, auto ph_c) __attribute__((always_inline)) {
                    constexpr int PH = decltype(ph_c)::value;
                    p = (const int*)((const char*)p + rec_stride);
                    BPG_LOAD(nxt)                    // (the table is padded by one round of records: no clamp)
                    i32x2 sp;
                    sp[0] = 0;
                    sp[1] = cur.template dw<1>();
                    auto unit_step = [&](auto uc) __attribute__((always_inline)) {
                        constexpr int u = decltype(uc)::value;
                        (void)&X; (void)&ac; (void)&cur; (void)&nxt; (void)&sp;
                        constexpr int SL_ISSUE = (u + AH + PH) % RING, SL_USE = (u + PH) % RING;
                        if constexpr (u + AH < NU) {
                            BPG_ISSUE(cur, u + AH, SL_ISSUE)
                        } else {
                            if constexpr (u + AH == NU) BPG_WAIT(nxt)     // first look into the next record
                            BPG_ISSUE(nxt, u + AH - NU, SL_ISSUE)
                        }
                        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(4 * AH) : "memory");
                        auto term_step = [&](auto kc) __attribute__((always_inline)) {
                            constexpr int k = decltype(kc)::value, tm = u * TPU + k;
                            (void)&X; (void)&ac; (void)&cur; (void)&sp;
                            // per-station weights: the station's weight enters the SGPR pair at its first term
                            if constexpr (!UNI && (tm & 1) == 0) sp[1] = cur.template dw<UNI ? 1 : 3 + 2 * (tm >> 1)>();
#pragma unroll
                            for (int r = 0; r < RPT; ++r) {
                                if (!MULTI && tm == 0) BPF_PKFMA0(ac[r], sp, X[SL_USE][k * RPT + r]);
                                else BPF_PKFMA(ac[r], sp, X[SL_USE][k * RPT + r]);
                            }
                        };
                        bpf_for_each(term_step, std::make_integer_sequence<int, TPU>{});
                    };
                    bpf_for_each(unit_step, std::make_integer_sequence<int, NU>{});
                    bool last = true;
                    if constexpr (MULTI) { ++part; last = part == nparts; }
                    if (last) {
                        int vsid;                    // (v_cndmask takes one scalar operand: the mask)
                        asm volatile("v_mov_b32 %0, %1" : "=v"(vsid) : "s"(cur.template dw<0>()));
                        unsigned long long mk[TPW];
#pragma unroll
                        for (int j = 0; j < TPW; ++j)
                            asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(mk[j]) : "v"(ac[j >> 1][j & 1]), "v"(bestg[j]));
#pragma unroll
                        for (int j = 0; j < TPW; ++j) {
                            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(bestg[j]) : "v"(ac[j >> 1][j & 1]), "s"(mk[j]));
                            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(argg[j]) : "v"(vsid), "s"(mk[j]));
                        }
                        if constexpr (MULTI) {
                            part = 0;
#pragma unroll
                            for (int r = 0; r < RPT; ++r) ac[r] = (f32x2){0.0f, 0.0f};
                        }
                    }
                };
                using ic0 = std::integral_constant<int, 0>;
                using icn = std::integral_constant<int, NU % 4>;      // ring phase of the odd parts (0 or 2)
                int it = 0;
                // option bp.slot_prio (BpfYield above).  Measured, round 5, profiles/r05_bp_slot_prio.txt: cfg3 151.6 ->
                // 146.7 ms = 0.726 -> 0.750 of the gather rate -- the isolated loop's own ceiling --, all 20 stations
                // 0.702 -> 0.741; schedules that change level at 1/2, 3/4, 7/8 or at 1/8, 1/4, 1/2 of the run gain
                // about half of that, the reverse order nothing.
                BpfYield yield(slot_prio, wv);
                for (; it + 1 < n_it; it += 2) {
                    yield.at(n_it, 2);
                    part_g(A, B, ic0{});
                    part_g(B, A, icn{});
                }
                if (it < n_it) part_g(A, B, ic0{});   // odd number of parts
                // the three units issued past the wave's last part (they read whatever record follows: valid LDS
                // addresses of some group, or the zero slab)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(X[0][0]), "+v"(X[0][1]), "+v"(X[0][2]), "+v"(X[0][3]), "+v"(X[1][0]), "+v"(X[1][1]),
                               "+v"(X[1][2]), "+v"(X[1][3]), "+v"(X[2][0]), "+v"(X[2][1]), "+v"(X[2][2]), "+v"(X[2][3]),
                               "+v"(X[3][0]), "+v"(X[3][1]), "+v"(X[3][2]), "+v"(X[3][3])
                             :: "memory");
#undef BPG_LOAD
#undef BPG_WAIT
#undef BPG_ADDR
#undef BPG_ISSUE
            };
            // ---- HALVES with uniform weights: the records live in SGPRs.  A record {id, weight, 2 TP LDS
            // byte addresses} is wave-uniform; as vector loads (walk) every quad of it costs 16 cycles of
            // the vector memory path -- 18 000 cycles per group entry against 15 000 of LDS gathers, 7-10 %
            // of the kernel at every tile (profiles/r03_bp_fast_phase_cycles.txt) -- and a 20-register
            // ring in VGPRs.  Here the wave's NEXT record is fetched with s_load_dwordx8 at the start of a
            // part into the spare one of two SGPR buffers and waited for once, where the look-ahead first
            // needs it (lgkmcnt(0): scalar loads return out of order, so that wait also drains the wave's
            // gathers -- one bubble per record, covered by the other 15 waves).  The addresses reach
            // v_add_u32 as SGPR operands, id and weight need no v_readfirstlane, and the 20 VGPRs go to
            // four more carried sources per wave (10 instead of 6: 160 sources per group entry).
            auto walk_s = [&](auto tp_c) {
                constexpr int TP = decltype(tp_c)::value;
                constexpr int NTERM = 2 * TP, NU = NTERM / TPU, AH = 2, RING = AH + 1;
                constexpr int NV = (2 + NTERM + 7) / 8;          // s_load_dwordx8 per record (the table is padded)
                static_assert(TPU == 2 && NV <= 3 && NU >= AH + 1, "tile 256, at most 11 stations per record");
                i32x8 A[NV], B[NV];
                const int* p = p_first;
#define BPS_LOAD(dst)                                                                          \
    {                                                                                          \
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(dst[0]) : "s"(p));                    \
        if constexpr (NV > 1) asm volatile("s_load_dwordx8 %0, %1, 0x20" : "=s"(dst[NV > 1 ? 1 : 0]) : "s"(p)); \
        if constexpr (NV > 2) asm volatile("s_load_dwordx8 %0, %1, 0x40" : "=s"(dst[NV > 2 ? 2 : 0]) : "s"(p)); \
    }
// lgkmcnt(0) with every register of the record as an in/out operand: nothing that reads them can be
// scheduled above the wait (tools/check_inflight.py checks the listing)
#define BPS_WAIT(dst)                                                                          \
    {                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dst[0]) :: "memory");                       \
        if constexpr (NV > 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dst[NV > 1 ? 1 : 0]) :: "memory"); \
        if constexpr (NV > 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dst[NV > 2 ? 2 : 0]) :: "memory"); \
    }
#define BPS_DW(rec, i) (rec[(i) >> 3][(i) & 7])
// unit w = station w: its two LDS byte addresses (uniform weights), or 16-bit float offsets out of the
// station's {offs_P | offs_S << 16, weight} pair -- scalar shifts, not VALU
#define BPS_ISSUE(rec, w, sl)                                                                  \
    {                                                                                          \
        const unsigned a_ = v_base + (UNI ? (unsigned)BPS_DW(rec, 2 + 2 * (w)) : (((unsigned)BPS_DW(rec, 2 + 2 * (w)) & 0xffffu) << 2)); \
        const unsigned b_ = v_base + (UNI ? (unsigned)BPS_DW(rec, 3 + 2 * (w)) : (((unsigned)BPS_DW(rec, 2 + 2 * (w)) >> 16) << 2)); \
        BPF_RD64(X[sl][0], a_, 0); BPF_RD64(X[sl][1], a_, 512);                                \
        BPF_RD64(X[sl][2], b_, 0); BPF_RD64(X[sl][3], b_, 512);                                \
    }
                BPS_LOAD(A)
                BPS_WAIT(A)
                f32x2 X[RING][4];
                BPS_ISSUE(A, 0, 0) BPS_ISSUE(A, 1, 1)
                f32x2 ac[RPT];
                auto part = [&](auto& cur, auto& nxt, auto ph_c, auto upd_c) __attribute__((always_inline)) {
                    constexpr int PH = decltype(ph_c)::value;
                    constexpr int UPD = decltype(upd_c)::value;
                    p = (const int*)((const char*)p + rec_stride);
                    BPS_LOAD(nxt)                    // (the table is padded by one round of records: no clamp)
                    i32x2 sp;
                    sp[0] = 0;
                    sp[1] = BPS_DW(cur, 1);
                    auto unit_step = [&](auto uc) __attribute__((always_inline)) {
                        constexpr int u = decltype(uc)::value;
                        (void)&X; (void)&ac; (void)&cur; (void)&nxt; (void)&sp;
                        constexpr int SL_ISSUE = (u + AH + PH) % RING, SL_USE = (u + PH) % RING;
                        if constexpr (u + AH < NU) {
                            BPS_ISSUE(cur, u + AH, SL_ISSUE)
                        } else {
                            if constexpr (u + AH == NU) BPS_WAIT(nxt)     // first look into the next record
                            BPS_ISSUE(nxt, u + AH - NU, SL_ISSUE)
                        }
                        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(4 * AH) : "memory");
                        if constexpr (!UNI) sp[1] = BPS_DW(cur, 3 + 2 * u);      // the weight of station u
#pragma unroll
                        for (int k = 0; k < TPU; ++k)
#pragma unroll
                            for (int r = 0; r < RPT; ++r) BPF_PKFMA(ac[r], sp, X[SL_USE][k * RPT + r]);
                    };
                    bpf_for_each(unit_step, std::make_integer_sequence<int, NU>{});
                    if constexpr (UPD == 2) {
                        const int sid = BPS_DW(cur, 0);
                        if (!g_store && sid >= 0) {              // (id -1: the padding of a short group)
#pragma unroll
                            for (int j = 0; j < TPW; ++j) {
                                const float a = ac[j >> 1][j & 1];
                                const bool take = (a > best[j]) | ((a == best[j]) & (sid < arg[j]));
                                best[j] = take ? a : best[j];
                                arg[j] = take ? sid : arg[j];
                            }
                        }
                    }
                };
                using ic1 = std::integral_constant<int, 1>;
                using ic2 = std::integral_constant<int, 2>;
                auto slot_step = [&](auto sc) __attribute__((always_inline)) {
                    constexpr int SLOT = decltype(sc)::value;
                    (void)&carry; (void)&ac; (void)&A; (void)&B;
                    // option bp.slot_prio: the 16 waves of an entry start together and do the same work, but the
                    // hardware serves the oldest first -- they reach the barrier thousands of cycles before the
                    // youngest, which then gather alone at a fraction of the LDS rate.  1: a wave lowers its issue
                    // priority as it gets ahead (slots 0-2: 3, 3-4: 2, 5-6: 1, 7-8: 0).  cfg5's share with
                    // all 40 stations: 3620 -> 3267 ms, 0.607 -> 0.673 of the gather rate (the reverse order: 0.609).
                    if (slot_prio == 1) __builtin_amdgcn_s_setprio(3 - (SLOT * 4) / NSLOT);
                    else if (slot_prio == 2) {
                        const int l = ((wv >> 2) + SLOT) & 3;
                        if (l == 3) __builtin_amdgcn_s_setprio(3);
                        else if (l == 2) __builtin_amdgcn_s_setprio(2);
                        else if (l == 1) __builtin_amdgcn_s_setprio(1);
                        else __builtin_amdgcn_s_setprio(0);
                    }

#pragma unroll
                    for (int r = 0; r < RPT; ++r) {
                        ac[r][0] = g_load ? carry[SLOT][r][0] : 0.0f;
                        ac[r][1] = g_load ? carry[SLOT][r][1] : 0.0f;
                    }
                    part(A, B, std::integral_constant<int, (2 * SLOT * NU) % RING>{}, ic1{});
                    part(B, A, std::integral_constant<int, ((2 * SLOT + 1) * NU) % RING>{}, ic2{});
#pragma unroll
                    for (int r = 0; r < RPT; ++r) carry[SLOT][r] = ac[r];      // (dead in the last residency)
                };
                bpf_for_each(slot_step, std::make_integer_sequence<int, NSLOT>{});
                // the units issued past the wave's last part (they read whatever record follows: valid LDS
                // addresses, or the zero slab) and the record behind it
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(X[0][0]), "+v"(X[0][1]), "+v"(X[0][2]), "+v"(X[0][3]), "+v"(X[1][0]), "+v"(X[1][1]),
                               "+v"(X[1][2]), "+v"(X[1][3]), "+v"(X[2][0]), "+v"(X[2][1]), "+v"(X[2][2]), "+v"(X[2][3])
                             :: "memory");
                BPS_WAIT(A)
#undef BPS_LOAD
#undef BPS_WAIT
#undef BPS_DW
#undef BPS_ISSUE
            };
            using std::integral_constant;
            using one_part = std::false_type;
            using parts = std::true_type;
            // wave-uniform, once per run.  Tile 512 serves <= 16 stations per source (one part);
            // the smaller tiles take any number of parts.
            // uniform weights: records of at most BPF_SREC_MAX_TP stations travel in SGPRs (walk_g)
            auto go = [&](auto tp_c, auto multi_c) __attribute__((always_inline)) {
                if constexpr (!HALVES && decltype(tp_c)::value <= BPF_SREC_MAX_TP) walk_g(tp_c, multi_c);
                else walk(tp_c, multi_c);
            };
            if constexpr (TPW == 8) {
                switch (run.tp) {
                    case 4: go(integral_constant<int, 4>{}, one_part{}); break;
                    case 6: go(integral_constant<int, 6>{}, one_part{}); break;
                    case 8: go(integral_constant<int, 8>{}, one_part{}); break;
                    case 10: go(integral_constant<int, 10>{}, one_part{}); break;
                    case 12: go(integral_constant<int, 12>{}, one_part{}); break;
                    case 14: go(integral_constant<int, 14>{}, one_part{}); break;
                    case 16: go(integral_constant<int, 16>{}, one_part{}); break;
                    default: break;
                }
            } else if constexpr (HALVES) {
                switch (run.tp) {
                    case 6: walk_s(integral_constant<int, 6>{}); break;
                    case 8: walk_s(integral_constant<int, 8>{}); break;
                    case 10: walk_s(integral_constant<int, 10>{}); break;
                    default: break;
                }
            } else if constexpr (TPW == 4) {
                switch (run.tp) {
                    case 6: go(integral_constant<int, 6>{}, parts{}); break;
                    case 8: go(integral_constant<int, 8>{}, parts{}); break;
                    case 10: go(integral_constant<int, 10>{}, parts{}); break;
                    case 12: go(integral_constant<int, 12>{}, parts{}); break;
                    case 14: go(integral_constant<int, 14>{}, parts{}); break;
                    case 16: go(integral_constant<int, 16>{}, parts{}); break;
                    default: break;
                }
            } else {
                switch (run.tp) {
                    case 8: go(integral_constant<int, 8>{}, parts{}); break;
                    case 12: go(integral_constant<int, 12>{}, parts{}); break;
                    case 16: go(integral_constant<int, 16>{}, parts{}); break;
                    case 20: go(integral_constant<int, 20>{}, parts{}); break;
                    case 24: go(integral_constant<int, 24>{}, parts{}); break;
                    default: break;
                }
            }
            if constexpr (!HALVES) {
#pragma unroll
                for (int j = 0; j < TPW; ++j) {
                    const bool take = (bestg[j] > best[j]) | ((bestg[j] == best[j]) & (argg[j] < arg[j]));
                    best[j] = take ? bestg[j] : best[j];
                    arg[j] = take ? argg[j] : arg[j];
                }
            }
        }
        BPF_PHASE(3);
#ifdef BPMF_PHASE_CYCLES
        ++ph_n_;
#endif
    }
    // ---- merge the 16 waves' maxima through LDS (value, then lowest id)
    __syncthreads();
    float* mb = lds;                        // [WPB][TILE]
    int* ma = (int*)(lds + WPB * TILE);     // [WPB][TILE]
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        mb[wv * TILE + slot_x(j)] = best[j];
        ma[wv * TILE + slot_x(j)] = arg[j];
    }
    __syncthreads();
    for (int x = tid; x < TILE; x += NTHREADS) {
        float b = mb[x];
        int a = ma[x];
#pragma unroll
        for (int w = 1; w < WPB; ++w) {
            const float bw = mb[w * TILE + x];
            const int aw = ma[w * TILE + x];
            if (bw > b || (bw == b && aw < a)) { b = bw; a = aw; }
        }
        const long long t = t0 + x;
        if (t < N) { out_beam[t] = b; out_arg[t] = a; }
    }
#ifdef BPMF_PHASE_CYCLES
    if (lane == 0) {
        for (int i = 0; i < 4; ++i) atomicAdd(&g_bpf_phase[wv >> 2][i], ph_acc_[i]);
        atomicAdd(&g_bpf_phase[wv >> 2][4], (unsigned long long)ph_n_);
        atomicAdd(&g_bpf_phase[wv >> 2][5], 1ull);
    }
#endif
}

int launch_beam_fast(const BpFastClass& fc, int id_offset, const float* U, size_t N, long long tile_lo,
                     long long tile_hi, hipStream_t stream, float* beam, int32_t* arg, int n_split,
                     long long split_stride, float best0)
{
    if (tile_hi <= tile_lo) return 0;
    const long long n_tiles = tile_hi - tile_lo;
    const size_t lds = std::max(fc.lds_bytes, (size_t)2 * BPF_WPB * fc.tile * sizeof(float));
    dim3 grid((unsigned)((n_tiles + 7) / 8 * 8), (unsigned)std::max(1, n_split));  // x: multiple of 8 (XCD-aware tile order)
    // waves that copy descriptors = KB of the LDS slab the plan left free (16 bytes per window)
    const int desc_waves = fc.desc_waves;
#define BPF_LAUNCH(...)                                                                            \
    do {                                                                                           \
        auto kern = bp_beam_fast_kernel<__VA_ARGS__>;                                              \
        BPMF_HIP_CHECK(hipFuncSetAttribute((const void*)kern,                                      \
                                           hipFuncAttributeMaxDynamicSharedMemorySize,             \
                                           (int)BP_LDS_MAX));                                      \
        kern<<<grid, dim3(BPF_THREADS), lds, stream>>>(                                            \
            U, (long long)N, fc.d_groups, fc.n_groups, fc.d_runs, fc.d_wins, fc.d_recs,            \
            fc.rec_dw, id_offset, tile_lo, n_tiles, beam, arg, desc_waves, split_stride, best0,   \
            fc.halves ? fc.n_pass : 1, (int)option(OPT_BP_SLOT_PRIO));                                \
    } while (0)
    if (fc.tile == 512) { if (fc.uniform) BPF_LAUNCH(true, 8); else BPF_LAUNCH(false, 8); }
    else if (fc.tile == 256 && fc.halves) { if (fc.uniform) BPF_LAUNCH(true, 4, true); else BPF_LAUNCH(false, 4, true); }
    else if (fc.tile == 256) { if (fc.uniform) BPF_LAUNCH(true, 4); else BPF_LAUNCH(false, 4); }
    else { if (fc.uniform) BPF_LAUNCH(true, 2); else BPF_LAUNCH(false, 2); }
#undef BPF_LAUNCH
    BPMF_LAUNCH_CHECK();
    return 0;
}

}  // namespace bpmf

#ifdef BPMF_PHASE_CYCLES
// 4 age groups x {4 phase sums, entries, waves, -, -}; reset != 0 clears the counters afterwards
extern "C" int bpmf_phase_read_bp(unsigned long long* out32, int reset)
{
    BPMF_HIP_CHECK(hipDeviceSynchronize());
    BPMF_HIP_CHECK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(bpmf::g_bpf_phase), sizeof(bpmf::g_bpf_phase)));
    if (reset) {
        unsigned long long z[4][8] = {};
        BPMF_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(bpmf::g_bpf_phase), z, sizeof(z)));
    }
    return 0;
}
#endif
