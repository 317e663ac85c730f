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
//   * THE GATHER PIPELINE NEVER DRAINS AT A SOURCE BOUNDARY.  Round 1 kept a source's record in
//     SGPRs; scalar loads share lgkmcnt with the LDS and return out of order, so the record of
//     the next source could only be fetched behind an lgkmcnt(0), i.e. with no gather in flight:
//     every source paid a drain, a scalar-cache round trip and a pipeline refill (the inner loop
//     alone reaches 78 % of the ds_read_b64 rate with the max update, the kernel reached 67 %).
//     Here the record lives in VGPRs (wave-uniform values) and is refilled by VECTOR loads, which
//     have their own in-order counter (vmcnt): quad q of the record (4 dwords = 4 units) is
//     re-loaded with the next source's quad as soon as its units have been accumulated, and the
//     first three units of the next source are issued while the last three of the current one are
//     still in flight -- 12 gathers stay outstanding across the whole run of sources, and the
//     max / arg-max update of a source runs under the next source's gathers.
//   * the record pointer advances by a constant (tables are padded by one round of records, so the
//     prefetch of the source after the last needs no clamp): 2 SALU per source.
//   * the accumulators start from the fma's own constant-0 addend (no zero-init moves), and the
//     max / arg-max update is hand-scheduled: 8 v_cmp into 8 SGPR pairs, then 16 v_cndmask -- no
//     VALU-writes-SGPR wait states between a compare and its selects.
//
// Arithmetic (and therefore every output bit) is that of oracle/bpmf_oracle.c:bp_cpu: per source
// an fmaf chain over (station outer, phase inner) starting from +0, strict > keeps the lowest id.
#include "bp_plan.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace bpmf {

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
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

constexpr int BPF_WPB = 16;             // waves per workgroup = per CU (the dual windows take the LDS)
constexpr int BPF_TILE = 512;           // 64 lanes x 4 sample pairs
constexpr int BPF_THREADS = 64 * BPF_WPB;

template <bool UNI>
__global__ __launch_bounds__(BPF_THREADS) void bp_beam_fast_kernel(
    const float* __restrict__ U, long long N, const BpFastGroup* __restrict__ groups, int n_groups,
    const BpRun* __restrict__ runs, const BpWindow* __restrict__ wins, const int* __restrict__ recs,
    int rec_dw, int id_offset, long long tile_lo, long long n_tiles, float* __restrict__ out_beam,
    int* __restrict__ out_arg, int desc_waves, long long split_stride)
{
    // short series: workgroup (tile, y) walks the groups [n_groups y / Y, n_groups (y + 1) / Y) and
    // writes its partial maxima to out + y * split_stride (bp.hip: bp_split_count, bp_merge_splits_kernel)
    const int g_lo = (int)((long long)n_groups * blockIdx.y / gridDim.y);
    const int g_hi = (int)((long long)n_groups * (blockIdx.y + 1) / gridDim.y);
    out_beam += (size_t)blockIdx.y * (size_t)split_stride;
    out_arg += (size_t)blockIdx.y * (size_t)split_stride;
    extern __shared__ float lds[];
    constexpr int TPW = 8, TILE = BPF_TILE, WPB = BPF_WPB, NTHREADS = BPF_THREADS;
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
    for (int j = 0; j < TPW; ++j) { best[j] = 0.0f; arg[j] = id_offset; }
    for (int x = tid; x < TILE; x += NTHREADS) lds[x] = 0.0f;  // the zero slab
    const long long rec_stride = (long long)rec_dw * 4 * WPB;   // bytes between a wave's sources

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

    for (int g = g_lo; g < g_hi; ++g) {
        const BpFastGroup grp = groups[g];
        __syncthreads();  // previous group's gathers are done, this group's descriptors are in LDS
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
        __syncthreads();
        if (g + 1 < g_hi) prefetch_descriptors(groups[g + 1].first_win);

        for (int rr = 0; rr < grp.n_run; ++rr) {
            const BpRun run = runs[grp.first_run + rr];
            const int n_mine = run.n_src > wv ? (run.n_src - wv + WPB - 1) / WPB : 0;   // sources of this wave
            if (n_mine == 0) continue;
            const int* p_first = recs + ((long long)run.first_rec + wv) * rec_dw;
            // group-local running max of this run: sources arrive by ascending id, so a plain
            // strict > keeps the lowest id on ties; the full tie rule merges the run into best/arg
            float bestg[TPW];
            int argg[TPW];
#pragma unroll
            for (int j = 0; j < TPW; ++j) { bestg[j] = -INFINITY; argg[j] = 0x7fffffff; }

            auto walk = [&](auto nst_c) {
                constexpr int NST = decltype(nst_c)::value;
                constexpr int NU = 2 * NST, AH = 3, Q = NU / 4;   // units, units in flight ahead, quads
                static_assert(NST >= 4 && NST % 2 == 0 && NST <= 16, "fast path: 4..16 stations, even");
                // The record of the CURRENT source, rolling: quad q = body dwords 4q .. 4q+3 (the
                // addresses -- or {offsets, weight} pairs -- of units 4q .. 4q+3).  Once step 4q+3 has
                // accumulated its unit, quad q is re-loaded with the next source's quad q.
                i32x4 R[Q];
                i32x2 h_cur, h_next;          // {id, weight}
                const int* p = p_first;
                BPF_LOADX2(h_cur, vzero, p, 0);
#define BPF_LOADQ(q) \
    if constexpr ((q) < Q) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(R[(q) < Q ? (q) : 0]) : "v"(vzero), "s"(p), "n"(8 + 16 * (q)))
                BPF_LOADQ(0); BPF_LOADQ(1); BPF_LOADQ(2); BPF_LOADQ(3);
                BPF_LOADQ(4); BPF_LOADQ(5); BPF_LOADQ(6); BPF_LOADQ(7);
                // vmcnt(0) with every register of the record as an in/out operand: nothing that uses
                // them can be scheduled above the wait (tools/check_inflight.py checks the listing)
#define BPF_VMWAIT_ALL()                                                                              \
    do {                                                                                              \
        if constexpr (Q == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(h_cur), "+v"(R[0]), "+v"(R[1]));            \
        else if constexpr (Q == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(h_cur), "+v"(R[0]), "+v"(R[1]), "+v"(R[2])); \
        else if constexpr (Q == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(h_cur), "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3 % Q])); \
        else if constexpr (Q == 5) asm volatile("s_waitcnt vmcnt(0)" : "+v"(h_cur), "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3 % Q]), "+v"(R[4 % Q])); \
        else if constexpr (Q == 6) asm volatile("s_waitcnt vmcnt(0)" : "+v"(h_cur), "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3 % Q]), "+v"(R[4 % Q]), "+v"(R[5 % Q])); \
        else if constexpr (Q == 7) asm volatile("s_waitcnt vmcnt(0)" : "+v"(h_cur), "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3 % Q]), "+v"(R[4 % Q]), "+v"(R[5 % Q]), "+v"(R[6 % Q])); \
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(h_cur), "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3 % Q]), "+v"(R[4 % Q]), "+v"(R[5 % Q]), "+v"(R[6 % Q]), "+v"(R[7 % Q])); \
    } while (0)
                BPF_VMWAIT_ALL();
                f32x2 X[4][4];  // ring of 4 units in flight
                // unit u of the source whose record R holds: 4 gathers of 8 bytes per lane
#define BPF_ISSUE_U(u)                                                                         \
    {                                                                                          \
        unsigned a_;  /* plain C++ adds: an inline-asm add would cost a hazard s_nop before the reads */ \
        if constexpr (UNI) {                                                                   \
            a_ = v_base + (unsigned)R[(u) >> 2][(u) & 3];                                      \
        } else {                                                                               \
            const unsigned o_ = (unsigned)R[(u) >> 2][2 * (((u) >> 1) & 1)];                   \
            a_ = v_base + ((((u) & 1) ? (o_ >> 16) : (o_ & 0xffffu)) << 2);                    \
        }                                                                                      \
        BPF_RD64(X[(u) & 3][0], a_, 0); BPF_RD64(X[(u) & 3][1], a_, 512);                      \
        BPF_RD64(X[(u) & 3][2], a_, 1024); BPF_RD64(X[(u) & 3][3], a_, 1536);                  \
    }
#pragma unroll
                for (int u = 0; u < AH; ++u) BPF_ISSUE_U(u)
                for (int it = 0; it < n_mine; ++it) {
                    f32x2 ac[4];
                    i32x2 sp_u;
                    // the header of the next source (the table is padded by one round of records: no clamp)
                    p = (const int*)((const char*)p + rec_stride);
                    BPF_LOADX2(h_next, vzero, p, 0);
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        // ---- keep three units in flight ahead of unit u: the unit issued now belongs to
                        // this source (u + 3 < NU) or is one of the first three of the NEXT source, whose
                        // quad 0 was requested at step 3 and is followed by Q - 2 younger loads
                        if (u + AH < NU) {
                            if ((u + AH) % 4 == 0)   // first use of quad (u + 3) / 4, loaded one source ago
                                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(R[((u + AH) >> 2) % Q]) : "n"(Q - 1));
                            BPF_ISSUE_U(u + AH)
                        } else {
                            if (u + AH == NU)
                                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(R[0]), "+v"(h_next) : "n"(Q - 2));
                            BPF_ISSUE_U(u + AH - NU)
                        }
                        asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
                        // the weight reaches v_pk_fma_f32 as the high half of an SGPR pair: with three
                        // 64-bit VGPR operands the instruction is register-read bound (measured: the
                        // kernel lost 5 points of the LDS rate with the weight in a VGPR pair)
                        i32x2 sp;
                        if constexpr (UNI) { if (u == 0) { sp_u[0] = 0; sp_u[1] = __builtin_amdgcn_readfirstlane(h_cur[1]); } sp = sp_u; }
                        else if ((u & 1) == 0) { sp_u[0] = 0; sp_u[1] = __builtin_amdgcn_readfirstlane(R[u >> 2][2 * ((u >> 1) & 1) + 1]); sp = sp_u; }
                        else sp = sp_u;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            if (u == 0) BPF_PKFMA0(ac[jj], sp, X[u & 3][jj]);
                            else BPF_PKFMA(ac[jj], sp, X[u & 3][jj]);
                        }
                        // ---- quad u / 4 is consumed (addresses issued, weights multiplied): refill it
                        if (u % 4 == 3) {
                            switch (u >> 2) {
                                case 0: BPF_LOADQ(0); break; case 1: BPF_LOADQ(1); break;
                                case 2: BPF_LOADQ(2); break; case 3: BPF_LOADQ(3); break;
                                case 4: BPF_LOADQ(4); break; case 5: BPF_LOADQ(5); break;
                                case 6: BPF_LOADQ(6); break; default: BPF_LOADQ(7); break;
                            }
                        }
                    }
                    // ---- max / arg-max update, strict >: 8 compares, then 16 selects; the next source's
                    // first three units are in flight meanwhile
                    {
                        unsigned long long mk[TPW];
#pragma unroll
                        for (int j = 0; j < TPW; ++j)
                            asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(mk[j]) : "v"(ac[j >> 1][j & 1]), "v"(bestg[j]));
#pragma unroll
                        for (int j = 0; j < TPW; ++j) {
                            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(bestg[j]) : "v"(ac[j >> 1][j & 1]), "s"(mk[j]));
                            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(argg[j]) : "v"(h_cur[0]), "s"(mk[j]));
                        }
                    }
                    h_cur = h_next;
                }
                // the three units issued past the wave's last source (they read whatever record follows:
                // valid LDS addresses of some group, or the zero slab) and the last refills
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef BPF_ISSUE_U
#undef BPF_LOADQ
#undef BPF_VMWAIT_ALL
            };
            switch (run.nst) {   // wave-uniform, once per run
                case 4: walk(std::integral_constant<int, 4>{}); break;
                case 6: walk(std::integral_constant<int, 6>{}); break;
                case 8: walk(std::integral_constant<int, 8>{}); break;
                case 10: walk(std::integral_constant<int, 10>{}); break;
                case 12: walk(std::integral_constant<int, 12>{}); break;
                case 14: walk(std::integral_constant<int, 14>{}); break;
                case 16: walk(std::integral_constant<int, 16>{}); break;
                default: break;
            }
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const bool take = (bestg[j] > best[j]) | ((bestg[j] == best[j]) & (argg[j] < arg[j]));
                best[j] = take ? bestg[j] : best[j];
                arg[j] = take ? argg[j] : arg[j];
            }
        }
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
}

int launch_beam_fast(const bpmf_bp_plan* pl, const float* U, size_t N, long long tile_lo,
                     long long tile_hi, hipStream_t stream, float* beam, int32_t* arg, int n_split,
                     long long split_stride)
{
    if (tile_hi <= tile_lo) return 0;
    const long long n_tiles = tile_hi - tile_lo;
    const size_t lds = std::max(pl->lds_bytes, (size_t)2 * BPF_WPB * BPF_TILE * sizeof(float));
    dim3 grid((unsigned)((n_tiles + 7) / 8 * 8), (unsigned)std::max(1, n_split));  // x: multiple of 8 (XCD-aware tile order)
    // waves that copy descriptors = KB of the LDS slab the plan left free (16 bytes per window)
    const int desc_waves = (int)std::min<size_t>(BPF_DESC_MAX, (2 * pl->S * pl->P + 63) / 64 * 64) / 64;
#define BPF_LAUNCH(UNI)                                                                            \
    do {                                                                                           \
        auto kern = bp_beam_fast_kernel<UNI>;                                                      \
        BPMF_HIP_CHECK(hipFuncSetAttribute((const void*)kern,                                      \
                                           hipFuncAttributeMaxDynamicSharedMemorySize,             \
                                           (int)BP_LDS_MAX));                                      \
        kern<<<grid, dim3(BPF_THREADS), lds, stream>>>(                                            \
            U, (long long)N, pl->d_fgroups, pl->n_groups, pl->d_fruns, pl->d_fwins,                \
            pl->d_frecs, pl->fast_rec_dw, pl->id_offset, tile_lo, n_tiles, beam, arg, desc_waves, \
            split_stride);                                                                          \
    } while (0)
    if (pl->fast_uniform) BPF_LAUNCH(true); else BPF_LAUNCH(false);
#undef BPF_LAUNCH
    BPMF_LAUNCH_CHECK();
    return 0;
}

}  // namespace bpmf
