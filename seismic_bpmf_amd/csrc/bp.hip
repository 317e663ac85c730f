// Backprojection hot path for MI355X (gfx950): beam-power shift-and-stack of S*C feature
// traces over K candidate sources and P phases, with per-sample max / arg-max.
//
// Serves beampower.beampower.beamform as called by the reference at
// BPMF/template_search.py:549-558 (reduce="max") and :560-569 (reduce="none"); the beam
// definition is the reference's own (tutorial notebook 5, cells 27/30/32).  The arithmetic
// is not in the reference tree.  Conventions = oracle/bpmf_oracle.c:bp_cpu, bit for bit.
//
// Design (DESIGN.md section "BP"): the path is an on-chip gather, not a GEMM.
//   1. prestack  U[s,p,t] = sum_c alpha[s,c,p] * feat[s,c,t]            (HBM streaming)
//   2. beam      a workgroup owns a tile of consecutive time samples and walks ALL
//      sources, grouped by the host-side plan so that for one group the needed window of
//      every used (station, phase) trace fits in LDS together.  Per source the threads
//      gather lds[window(s,p) + tau[k,s,p] - tau_min + t] (consecutive lanes -> consecutive
//      banks) and accumulate with the source weight in registers, keeping a running
//      (max, arg-max) per time sample: no atomics, no cross-workgroup merge, and the
//      sequential source order gives the "lowest index wins ties" rule for free.
#include "bp_plan.h"
#include "context.h"
#include <mutex>
#include <cstring>
#include <type_traits>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace bpmf {

// ------------------------------------------------------------------- prestack ---
// One thread per (station, time sample): reads the C components once, writes P phases.
template <int MAXP>
__global__ __launch_bounds__(256) void bp_prestack_kernel(const float* __restrict__ feat,
                                                          const float* __restrict__ w_ph,
                                                          long long N, int C, int P,
                                                          float* __restrict__ U, long long t_lo, long long t_hi)
{
    // (samples [t_lo, t_hi): the whole series, or one piece of a day that is still arriving from the host)
    const long long t = t_lo + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    if (t >= t_hi) return;
    float acc[MAXP];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) acc[p] = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float f = feat[((size_t)s * C + c) * (size_t)N + t];
#pragma unroll
        for (int p = 0; p < MAXP; ++p)
            if (p < P) acc[p] = __fmaf_rn(w_ph[((size_t)s * C + c) * P + p], f, acc[p]);
    }
#pragma unroll
    for (int p = 0; p < MAXP; ++p)
        if (p < P) U[((size_t)s * P + p) * (size_t)N + t] = acc[p];
}

// Generic P (slow path, P > 4): one thread per (s, p, t).
__global__ void bp_prestack_any_kernel(const float* __restrict__ feat,
                                       const float* __restrict__ w_ph, long long N, int C, int P,
                                       float* __restrict__ U, long long t_lo, long long t_hi)
{
    const long long t = t_lo + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y / P, p = blockIdx.y % P;
    if (t >= t_hi) return;
    float acc = 0.0f;
    for (int c = 0; c < C; ++c)
        acc = __fmaf_rn(w_ph[((size_t)s * C + c) * P + p], feat[((size_t)s * C + c) * (size_t)N + t],
                        acc);
    U[((size_t)s * P + p) * (size_t)N + t] = acc;
}

// ----------------------------------------------------------------------- beam ---
template <int NBLK>
struct BpMeta {  // one source's wave-uniform metadata, spread over the lanes of a wave
    int hd;
    int mo[NBLK];
    float mb[NBLK];
    __device__ __forceinline__ void load(const int* __restrict__ srcs,
                                         const int* __restrict__ term_off,
                                         const float* __restrict__ term_beta, int NT, int k,
                                         int lane)
    {
        hd = srcs[(size_t)k * 4 + (lane & 3)];
#pragma unroll
        for (int q = 0; q < NBLK; ++q) {
            const int l = lane + 64 * q;
            mo[q] = l < NT ? term_off[(size_t)k * NT + l] : 0;
            mb[q] = l < NT ? term_beta[(size_t)k * NT + l] : 0.0f;
        }
    }
};

__device__ __forceinline__ int lane_bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float lane_bcast(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// LDS layout of a group: [0, tile) is a slab of zeros that padded terms point at, then one
// window per used (station, phase) row: lds[base + x] = U[row][t0 + tau_min(row) + x].
//
// All per-source / per-chunk metadata is wave-uniform.  It is fetched with ONE coalesced
// vector load per wave (lane l <- entry l) one source ahead of its use and broadcast with
// v_readlane: no scalar-memory round trip sits between the LDS gathers (SMEM and LDS share
// the lgkm counter; round 3 found that ONE lgkmcnt(0) per source is cheap with 16 waves per CU --
// bp_fast.hip keeps its records in SGPRs -- but these general kernels predate that and only run the
// edge tiles of a day, reduce="none" and the P != 2 grids).
template <int TPT, int CHUNK, int NBLK, int OOB, int REDUCE>
__global__ __launch_bounds__(BP_THREADS) void bp_beam_kernel(
    const float* __restrict__ U, long long N, const BpGroup* __restrict__ groups, int n_groups,
    const int4* __restrict__ chunks, const int* __restrict__ srcs,
    const int* __restrict__ term_off, const float* __restrict__ term_beta, int NT,
    int id_offset, float* __restrict__ out_beam, int* __restrict__ out_arg, long long tile_base, float best0)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    constexpr int TILE = BP_THREADS * TPT;
    const long long t0 = (tile_base + (long long)blockIdx.x) * TILE;

    float best[TPT];
    int arg[TPT];
#pragma unroll
    for (int j = 0; j < TPT; ++j) {
        best[j] = best0;
        arg[j] = id_offset;
        lds[tid + j * BP_THREADS] = 0.0f;  // the zero slab (never overwritten)
    }

    for (int g = 0; g < n_groups; ++g) {
        const BpGroup grp = groups[g];
        // Per-source metadata (header word l&3, term offsets, term weights: lane l <- entry l).
        // Three register sets rotate through the source loop (unrolled x3, no copies), so a
        // set is loaded two sources before it is consumed.  The first two are issued before
        // the staging.
        const int k_last = grp.first_src + grp.n_src - 1;
        BpMeta<NBLK> m0, m1, m2;
        m0.load(srcs, term_off, term_beta, NT, grp.first_src, lane);
        m1.load(srcs, term_off, term_beta, NT, min(grp.first_src + 1, k_last), lane);
        __syncthreads();  // previous group's gathers are done
        // ---- stage the group's windows: descriptors by readlane, 4 loads in flight per thread
        for (int cb = 0; cb < grp.n_chunk; cb += 64) {
            const int nb = min(64, grp.n_chunk - cb);
            int4 d = make_int4(0, 0, 0, 0);
            if (lane < nb) d = chunks[grp.first_chunk + cb + lane];
            for (int c = 0; c < nb; c += 4) {
                // descriptors first (readlane), then 4 unconditional loads from clamped
                // addresses (no branch between them, so all four stay in flight), then the
                // LDS writes with the out-of-range samples zeroed
                int row[4], dd[4], n[4];
                long long gi[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cc = min(c + u, nb - 1);
                    row[u] = lane_bcast(d.x, cc);
                    gi[u] = t0 + lane_bcast(d.y, cc) + tid;
                    dd[u] = lane_bcast(d.z, cc);
                    n[u] = c + u < nb ? lane_bcast(d.w, cc) : 0;
                }
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long gc = gi[u] < 0 ? 0 : (gi[u] >= N ? N - 1 : gi[u]);
                    v[u] = U[(size_t)row[u] * (size_t)N + gc];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (tid < n[u]) lds[dd[u] + tid] = (gi[u] >= 0 && gi[u] < N) ? v[u] : 0.0f;
            }
        }
        __syncthreads();
        // ---- walk the sources of the group, three per trip.  Past the end of the group the
        // clamped index re-processes the last source, which changes nothing (same id, same
        // value), so the trip needs no branch between its loads and their waits.
        auto process = [&](const BpMeta<NBLK>& m) {
            const int sid = lane_bcast(m.hd, 0), tmin = lane_bcast(m.hd, 1), tmax = lane_bcast(m.hd, 2);
            const int nterm = lane_bcast(m.hd, 3);  // terms padded to CHUNK; 0 = unused source
            float acc[TPT];
#pragma unroll
            for (int j = 0; j < TPT; ++j) acc[j] = 0.0f;

#define BP_LOAD(X, B, Q, C0)                                                          \
    _Pragma("unroll") for (int i = 0; i < CHUNK; ++i) {                               \
        const float* lp = lds + lane_bcast(m.mo[Q], (C0) + i) + tid;                  \
        B[i] = lane_bcast(m.mb[Q], (C0) + i);                                         \
        _Pragma("unroll") for (int j = 0; j < TPT; ++j) X[i][j] = lp[j * BP_THREADS]; \
    }
#define BP_FMA(X, B)                                                                  \
    _Pragma("unroll") for (int i = 0; i < CHUNK; ++i)                                 \
        _Pragma("unroll") for (int j = 0; j < TPT; ++j) acc[j] = __fmaf_rn(B[i], X[i][j], acc[j]);

#pragma unroll
            for (int q = 0; q < NBLK; ++q) {
                // chunks of block q, software-pipelined: the gathers of chunk i+1 are in
                // flight while chunk i is accumulated (same ascending fmaf order)
                const int nc = (min(64, nterm - 64 * q)) / CHUNK;
                if (nc <= 0) continue;
                float xa[CHUNK][TPT], xb[CHUNK][TPT], ba[CHUNK], bb[CHUNK];
                BP_LOAD(xa, ba, q, 0)
                for (int i2 = 1; i2 < nc; i2 += 2) {
                    BP_LOAD(xb, bb, q, i2 * CHUNK)
                    BP_FMA(xa, ba)
                    if (i2 + 1 < nc) { BP_LOAD(xa, ba, q, (i2 + 1) * CHUNK) }
                    BP_FMA(xb, bb)
                }
                if (nc & 1) { BP_FMA(xa, ba) }
            }
#undef BP_LOAD
#undef BP_FMA
#pragma unroll
            for (int j = 0; j < TPT; ++j) {
                const long long t = t0 + tid + j * BP_THREADS;
                bool computed = nterm > 0;
                if (OOB == BPMF_BP_STRICT) computed = computed && (t + tmin >= 0) && (t + tmax < N);
                if (REDUCE == BPMF_BP_REDUCE_MAX) {
                    // sources arrive in plan order, not index order: ties go to the lower id
                    const bool better = acc[j] > best[j] || (acc[j] == best[j] && sid < arg[j]);
                    if (computed && better) { best[j] = acc[j]; arg[j] = sid; }
                } else {
                    if (t < N)
                        out_beam[(size_t)(sid - id_offset) * (size_t)N + t] = computed ? acc[j] : 0.0f;
                }
            }
        };
        for (int k = grp.first_src; k <= k_last; k += 3) {
            m2.load(srcs, term_off, term_beta, NT, min(k + 2, k_last), lane);
            process(m0);
            m0.load(srcs, term_off, term_beta, NT, min(k + 3, k_last), lane);
            process(m1);
            m1.load(srcs, term_off, term_beta, NT, min(k + 4, k_last), lane);
            process(m2);
        }
    }
    if (REDUCE == BPMF_BP_REDUCE_MAX) {
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
            const long long t = t0 + tid + j * BP_THREADS;
            if (t < N) { out_beam[t] = best[j]; out_arg[t] = arg[j]; }
        }
    }
}

// ------------------------------------------------ beam, uniform-VGPR metadata ---
// Fast path for sources with at most NTV (station, phase) terms (NTV <= 32).  The per-term
// metadata {LDS byte offset, weight} is loaded with VECTOR loads from a wave-uniform address
// (every lane receives the same value), two sources ahead, so that the gather loop is three
// instructions per term: v_add (address) / ds_read2st64_b32 (TPT = 2 gathers) / v_pk_fma.
// (Round 1 ruled scalar loads out here -- SMEM shares the lgkm counter with the LDS gathers -- see the
// note above bp_beam_kernel; v_readlane broadcasting costs two more VALU issues per term.)
struct BpTermV {
    int off_bytes;  // LDS byte offset of the term's window origin (+ moveout)
    float beta;     // source weight of the term's station
};

template <int NTV>
struct BpMetaV {
    int4 hd;  // id, tmin, tmax, nterm (padded to 4)
    int4 t[NTV / 2];  // two BpTermV per int4
    __device__ __forceinline__ void load(const int4* __restrict__ srcs4,
                                         const int4* __restrict__ terms, int k, int vzero)
    {
        // `vzero` is 0 held in a VGPR the compiler cannot see through: it keeps these loads on
        // the vector-memory path although their address is wave-uniform
        const size_t kk = (size_t)(k + vzero);
        hd = srcs4[kk];
#pragma unroll
        for (int i = 0; i < NTV / 2; ++i) t[i] = terms[kk * (NTV / 2) + i];
    }
};

template <int TPT, int NTV, int OOB, int REDUCE>
__global__ __launch_bounds__(BP_THREADS) void bp_beam_uvgpr_kernel(
    const float* __restrict__ U, long long N, const BpGroup* __restrict__ groups, int n_groups,
    const int4* __restrict__ chunks, const int4* __restrict__ srcs4,
    const int4* __restrict__ terms, int id_offset, float* __restrict__ out_beam,
    int* __restrict__ out_arg, long long tile_base, float best0)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    constexpr int TILE = BP_THREADS * TPT;
    const long long t0 = (tile_base + (long long)blockIdx.x) * TILE;
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    const char* lds_t = (const char*)lds + tid * 4;

    float best[TPT];
    int arg[TPT];
#pragma unroll
    for (int j = 0; j < TPT; ++j) {
        best[j] = best0;
        arg[j] = id_offset;
        lds[tid + j * BP_THREADS] = 0.0f;  // the zero slab (never overwritten)
    }

    for (int g = 0; g < n_groups; ++g) {
        const BpGroup grp = groups[g];
        const int k_last = grp.first_src + grp.n_src - 1;
        BpMetaV<NTV> m0, m1, m2;
        m0.load(srcs4, terms, grp.first_src, vzero);
        m1.load(srcs4, terms, min(grp.first_src + 1, k_last), vzero);
        __syncthreads();  // previous group's gathers are done
        for (int cb = 0; cb < grp.n_chunk; cb += 64) {
            const int nb = min(64, grp.n_chunk - cb);
            int4 d = make_int4(0, 0, 0, 0);
            if (lane < nb) d = chunks[grp.first_chunk + cb + lane];
            for (int c = 0; c < nb; c += 4) {
                int row[4], dd[4], n[4];
                long long gi[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cc = min(c + u, nb - 1);
                    row[u] = lane_bcast(d.x, cc);
                    gi[u] = t0 + lane_bcast(d.y, cc) + tid;
                    dd[u] = lane_bcast(d.z, cc);
                    n[u] = c + u < nb ? lane_bcast(d.w, cc) : 0;
                }
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long gc = gi[u] < 0 ? 0 : (gi[u] >= N ? N - 1 : gi[u]);
                    v[u] = U[(size_t)row[u] * (size_t)N + gc];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (tid < n[u]) lds[dd[u] + tid] = (gi[u] >= 0 && gi[u] < N) ? v[u] : 0.0f;
            }
        }
        __syncthreads();

        auto process = [&](const BpMetaV<NTV>& m) {
            const int nterm = __builtin_amdgcn_readfirstlane(m.hd.w);
            float acc[TPT];
#pragma unroll
            for (int j = 0; j < TPT; ++j) acc[j] = 0.0f;
#pragma unroll
            for (int c = 0; c < NTV / 4; ++c) {
                if (c * 4 < nterm) {  // wave-uniform
                    float x[4][TPT];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int4 tt = m.t[(c * 4 + i) / 2];
                        const int ob = (i & 1) ? tt.z : tt.x;
                        const float* lp = (const float*)(lds_t + ob);
#pragma unroll
                        for (int j = 0; j < TPT; ++j) x[i][j] = lp[j * BP_THREADS];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int4 tt = m.t[(c * 4 + i) / 2];
                        const float beta = __int_as_float((i & 1) ? tt.w : tt.y);
#pragma unroll
                        for (int j = 0; j < TPT; ++j) acc[j] = __fmaf_rn(beta, x[i][j], acc[j]);
                    }
                }
            }
            const int sid = m.hd.x;
#pragma unroll
            for (int j = 0; j < TPT; ++j) {
                const long long t = t0 + tid + j * BP_THREADS;
                bool computed = nterm > 0;
                if (OOB == BPMF_BP_STRICT) computed = computed && (t + m.hd.y >= 0) && (t + m.hd.z < N);
                if (REDUCE == BPMF_BP_REDUCE_MAX) {
                    const bool better = acc[j] > best[j] || (acc[j] == best[j] && sid < arg[j]);
                    if (computed && better) { best[j] = acc[j]; arg[j] = sid; }
                } else {
                    if (t < N)
                        out_beam[(size_t)(sid - id_offset) * (size_t)N + t] = computed ? acc[j] : 0.0f;
                }
            }
        };
        for (int k = grp.first_src; k <= k_last; k += 3) {
            m2.load(srcs4, terms, min(k + 2, k_last), vzero);
            process(m0);
            m0.load(srcs4, terms, min(k + 3, k_last), vzero);
            process(m1);
            m1.load(srcs4, terms, min(k + 4, k_last), vzero);
            process(m2);
        }
    }
    if (REDUCE == BPMF_BP_REDUCE_MAX) {
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
            const long long t = t0 + tid + j * BP_THREADS;
            if (t < N) { out_beam[t] = best[j]; out_arg[t] = arg[j]; }
        }
    }
}

// ------------------------------------------------ beam, one wave per source ---
// Same data flow as bp_beam_uvgpr_kernel, but inside a workgroup the four waves take
// DIFFERENT sources (k, k+1, k+2, k+3, then +4 ...) and each wave covers the whole time tile
// (TPW samples per lane, tile = 64 * TPW).  The wave-uniform metadata of a source is then
// fetched by one wave only, which divides the vector-memory return traffic of the metadata
// broadcast (the limiter of the time-split kernel: 13 x 1 KiB per source per wave) by four,
// and one address add serves TPW gathers.  Every wave keeps its own running (max, arg-max)
// for the tile; they are merged through LDS at the end with the same (value, lowest id) order.
template <int TPW, int NTV, int OOB, int REDUCE>
__global__ __launch_bounds__(BP_THREADS) void bp_beam_wps_kernel(
    const float* __restrict__ U, long long N, const BpGroup* __restrict__ groups, int n_groups,
    const int4* __restrict__ chunks, const int4* __restrict__ srcs4,
    const int4* __restrict__ terms, int id_offset, float* __restrict__ out_beam,
    int* __restrict__ out_arg, long long tile_base, float best0)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    constexpr int NW = BP_THREADS / 64;
    constexpr int TILE = 64 * TPW;
    const long long t0 = (tile_base + (long long)blockIdx.x) * TILE;
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    // sample j of this lane inside the tile
    auto tmap = [&](int j) { return lane + 64 * j; };
    const char* lds_l = (const char*)lds + lane * 4;

    float best[TPW];
    int arg[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) { best[j] = best0; arg[j] = id_offset; }
    for (int x = tid; x < TILE; x += BP_THREADS) lds[x] = 0.0f;  // the zero slab

    for (int g = 0; g < n_groups; ++g) {
        const BpGroup grp = groups[g];
        const int k_last = grp.first_src + grp.n_src - 1;
        const int k_first = grp.first_src + wv;  // this wave's first source (may be > k_last)
        BpMetaV<NTV> m0, m1, m2;
        m0.load(srcs4, terms, min(k_first, k_last), vzero);
        m1.load(srcs4, terms, min(k_first + NW, k_last), vzero);
        __syncthreads();  // previous group's gathers are done
        for (int cb = 0; cb < grp.n_chunk; cb += 64) {
            const int nb = min(64, grp.n_chunk - cb);
            int4 d = make_int4(0, 0, 0, 0);
            if (lane < nb) d = chunks[grp.first_chunk + cb + lane];
            for (int c = 0; c < nb; c += 4) {
                int row[4], dd[4], n[4];
                long long gi[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cc = min(c + u, nb - 1);
                    row[u] = lane_bcast(d.x, cc);
                    gi[u] = t0 + lane_bcast(d.y, cc) + tid;
                    dd[u] = lane_bcast(d.z, cc);
                    n[u] = c + u < nb ? lane_bcast(d.w, cc) : 0;
                }
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long gc = gi[u] < 0 ? 0 : (gi[u] >= N ? N - 1 : gi[u]);
                    v[u] = U[(size_t)row[u] * (size_t)N + gc];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (tid < n[u]) lds[dd[u] + tid] = (gi[u] >= 0 && gi[u] < N) ? v[u] : 0.0f;
            }
        }
        __syncthreads();

        auto process = [&](const BpMetaV<NTV>& m, bool live) {
            const int nterm = live ? __builtin_amdgcn_readfirstlane(m.hd.w) : 0;
            float acc[TPW];
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[j] = 0.0f;
#pragma unroll
            for (int c = 0; c < NTV / 2; ++c) {
                if (c * 2 < nterm) {  // wave-uniform
                    const int4 tt = m.t[c];
                    const float* lp0 = (const float*)(lds_l + tt.x);
                    const float* lp1 = (const float*)(lds_l + tt.z);
                    float x0[TPW], x1[TPW];
#pragma unroll
                    for (int j = 0; j < TPW; ++j) { x0[j] = lp0[j * 64]; x1[j] = lp1[j * 64]; }
                    const float b0 = __int_as_float(tt.y), b1 = __int_as_float(tt.w);
#pragma unroll
                    for (int j = 0; j < TPW; ++j) acc[j] = __fmaf_rn(b0, x0[j], acc[j]);
#pragma unroll
                    for (int j = 0; j < TPW; ++j) acc[j] = __fmaf_rn(b1, x1[j], acc[j]);
                }
            }
            const int sid = m.hd.x;
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const long long t = t0 + tmap(j);
                bool computed = nterm > 0;
                if (OOB == BPMF_BP_STRICT) computed = computed && (t + m.hd.y >= 0) && (t + m.hd.z < N);
                if (REDUCE == BPMF_BP_REDUCE_MAX) {
                    const bool better = acc[j] > best[j] || (acc[j] == best[j] && sid < arg[j]);
                    if (computed && better) { best[j] = acc[j]; arg[j] = sid; }
                } else {
                    if (live && t < N)
                        out_beam[(size_t)(sid - id_offset) * (size_t)N + t] = computed ? acc[j] : 0.0f;
                }
            }
        };
        for (int k = k_first; k <= k_last; k += 3 * NW) {
            m2.load(srcs4, terms, min(k + 2 * NW, k_last), vzero);
            process(m0, true);
            m0.load(srcs4, terms, min(k + 3 * NW, k_last), vzero);
            process(m1, k + NW <= k_last);
            m1.load(srcs4, terms, min(k + 4 * NW, k_last), vzero);
            process(m2, k + 2 * NW <= k_last);
        }
    }
    if (REDUCE == BPMF_BP_REDUCE_MAX) {
        // merge the four waves' running maxima through LDS (window area is free now)
        __syncthreads();
        float* mb = lds;                       // [NW][TILE]
        int* ma = (int*)(lds + NW * TILE);     // [NW][TILE]
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            mb[wv * TILE + tmap(j)] = best[j];
            ma[wv * TILE + tmap(j)] = arg[j];
        }
        __syncthreads();
        for (int x = tid; x < TILE; x += BP_THREADS) {
            float b = mb[x];
            int a = ma[x];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const float bw = mb[w * TILE + x];
                const int aw = ma[w * TILE + x];
                if (bw > b || (bw == b && aw < a)) { b = bw; a = aw; }
            }
            const long long t = t0 + x;
            if (t < N) { out_beam[t] = b; out_arg[t] = a; }
        }
    }
}

// ------------------------------------- beam, one wave per source, packed metadata (P = 2) ---
// The production kernel for two-phase (P, S) grids.  Same flow as bp_beam_wps_kernel, but a
// source's metadata is packed per STATION as {off_P | off_S << 16, weight} (LDS float offsets
// fit 16 bits; both phases share the station weight): half the registers and half the
// vector-memory traffic of the per-term table, two prefetched sets instead of three.
// WPB = waves per workgroup (all waves share the group's LDS windows and take different
// sources).  Measured on cfg3 with the metadata in VGPRs: WPB 4 (8 waves/CU) 0.342 s, more waves
// spill; with the metadata in SGPRs (BpMetaS, <= 80 VGPRs): WPB 4 0.280 s, WPB 8 (16 waves/CU)
// 0.241 s, WPB 12 (24 waves/CU) 0.236 s.
template <int NSV>
struct BpMetaP {
    int4 hd;            // id, tmin, tmax, stations (padded to 2; 0 = unused source)
    int4 st[NSV / 2];   // two stations per int4: {offs, weight, offs, weight}
    __device__ __forceinline__ void load(const int4* __restrict__ srcs4,
                                         const int4* __restrict__ recs, int k, int vzero)
    {
        const size_t kk = (size_t)(k + vzero);  // vzero: see BpMetaV
        hd = srcs4[kk];
#pragma unroll
        for (int i = 0; i < NSV / 2; ++i) st[i] = recs[kk * (NSV / 2) + i];
    }
    __device__ __forceinline__ unsigned offs(int s) const { return (unsigned)((s & 1) ? st[s >> 1].z : st[s >> 1].x); }
    __device__ __forceinline__ float beta(int s) const { return __int_as_float((s & 1) ? st[s >> 1].w : st[s >> 1].y); }
    __device__ __forceinline__ int id() const { return hd.x; }
    __device__ __forceinline__ int tmin() const { return hd.y; }
    __device__ __forceinline__ int tmax() const { return hd.z; }
    __device__ __forceinline__ int nsta() const { return hd.w; }
};

// Scalar-register variant of the metadata (NSV <= 16): ONE set of 4 + 2 NSV SGPRs, filled by
// inline-asm s_load_dwordx4/x8 for the NEXT source right after the last gather of the current
// one, so the scalar-cache latency hides behind the max/arg-max epilogue.  SMEM shares lgkmcnt
// with the LDS gathers and returns out of order, hence the placement: nothing else is in flight
// between the issue and the lgkmcnt(0) that closes the source.  Frees 2 x (4 + 2 NSV) VGPRs,
// which is what lets 16 waves/CU (128 VGPRs) run without spills.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
template <int NSV>
struct BpMetaS {
    static constexpr int NS = NSV < 16 ? NSV : 16;  // stations held at a time; records have NSV slots
    i32x4 hd;
    i32x8 r[NS / 4];    // station s of the part: dword 2 s = offs, 2 s + 1 = weight
    __device__ __forceinline__ void issue(const int4* __restrict__ srcs4,
                                          const int4* __restrict__ recs, int k)
    {
        const int4* ph = srcs4 + k;
        const int4* pr = recs + (size_t)k * (NSV / 2);
        asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(hd) : "s"(ph));
#pragma unroll
        for (int i = 0; i < NS / 4; ++i) {
            const int4* pi = pr + 2 * i;
            asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(r[i]) : "s"(pi));
        }
    }
    // stations 16 part .. 16 part + 15 of source k (NSV > 16: a source is gathered in parts)
    __device__ __forceinline__ void issue_part(const int4* __restrict__ recs, int k, int part)
    {
        const int4* pr = recs + (size_t)k * (NSV / 2) + 8 * part;
#pragma unroll
        for (int i = 0; i < NS / 4; ++i) {
            const int4* pi = pr + 2 * i;
            asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(r[i]) : "s"(pi));
        }
    }
    // s_waitcnt lgkmcnt(0) with every register of the set as an INPUT: no part of the set is
    // dead (= free for the register allocator to hand out as a temporary) while the loads are
    // in flight.  Inputs only -- a "+s" tie makes the compiler copy the in-flight registers in
    // front of the wait.  The sched_barrier keeps later uses behind the wait.
    __device__ __forceinline__ void wait() const
    {
        static_assert(NSV % 4 == 0 && (NSV <= 16 || NSV % 16 == 0), "BpMetaS: NSV in {4, 8, 12, 16, 32, ...}");
        if constexpr (NS / 4 == 1)
            asm volatile("s_waitcnt lgkmcnt(0)" : : "s"(hd), "s"(r[0]) : "memory");
        else if constexpr (NS / 4 == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : : "s"(hd), "s"(r[0]), "s"(r[1]) : "memory");
        else if constexpr (NS / 4 == 3)
            asm volatile("s_waitcnt lgkmcnt(0)" : : "s"(hd), "s"(r[0]), "s"(r[1]), "s"(r[2]) : "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : : "s"(hd), "s"(r[0]), "s"(r[1]), "s"(r[2]), "s"(r[3]) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ unsigned offs(int s) const { return (unsigned)r[(2 * s) >> 3][(2 * s) & 7]; }
    __device__ __forceinline__ float beta(int s) const { return __int_as_float(r[(2 * s + 1) >> 3][(2 * s + 1) & 7]); }
    // {offs, weight} as one aligned SGPR pair: v_pk_fma_f32 takes the weight from its high half
    __device__ __forceinline__ i32x2 pair(int s) const
    {
        i32x2 p;
        p[0] = r[(2 * s) >> 3][(2 * s) & 7];
        p[1] = r[(2 * s + 1) >> 3][(2 * s + 1) & 7];
        return p;
    }
    __device__ __forceinline__ int id() const { return hd[0]; }
    __device__ __forceinline__ int tmin() const { return hd[1]; }
    __device__ __forceinline__ int tmax() const { return hd[2]; }
    __device__ __forceinline__ int nsta() const { return hd[3]; }
};

// B64 (plans with dual windows, see build_plan): every LDS offset is even, a lane owns the
// sample PAIRS 128 j + 2 lane + {0, 1} and gathers them with ds_read_b64 -- 256 B/clk/CU instead of
// the 128 B/clk/CU of the 4-byte gathers.
template <int WPB, int NSV, int OOB, int REDUCE, bool SMETA = false, bool B64 = false>
__global__ __launch_bounds__(64 * WPB, WPB >= 16 ? WPB / 4 : (WPB * 2 + 3) / 4) void bp_beam_wps2_kernel(
    const float* __restrict__ U, long long N, const BpGroup* __restrict__ groups, int n_groups,
    const int4* __restrict__ chunks, const int4* __restrict__ srcs4,
    const int4* __restrict__ recs, int id_offset, float* __restrict__ out_beam,
    int* __restrict__ out_arg, long long tile_base, long long n_tiles, long long split_stride, float best0)
{
    extern __shared__ float lds[];
    constexpr int TPW = 8;
    constexpr int TILE = 64 * TPW;
    constexpr int NTHREADS = 64 * WPB;
    // Short series (an event relocation: 3-6 tiles) leave most of the 256 CUs without a tile: the
    // launch then carries gridDim.y > 1 and workgroup (tile, y) walks only the groups
    // [n_groups y / Y, n_groups (y + 1) / Y).  reduce="none": every source still belongs to exactly
    // one workgroup of a tile; reduce="max": partial maxima go to out + y * split_stride and
    // bp_merge_splits_kernel folds them (value, then lowest id -- the rule of the 16-wave merge below).
    const int g_lo = (int)((long long)n_groups * blockIdx.y / gridDim.y);
    const int g_hi = (int)((long long)n_groups * (blockIdx.y + 1) / gridDim.y);
    if (REDUCE == BPMF_BP_REDUCE_MAX) {
        out_beam += (size_t)blockIdx.y * (size_t)split_stride;
        out_arg += (size_t)blockIdx.y * (size_t)split_stride;
    }
    constexpr bool GLOCAL = B64 && REDUCE == BPMF_BP_REDUCE_MAX;
    constexpr bool SPAIR = SMETA;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Consecutive workgroup ids go round-robin to the 8 XCDs (one L2 each): give every XCD a
    // contiguous run of tiles, so that each L2 stages its own eighth of the prestack instead of
    // all of it.  (The grid is rounded up to a multiple of 8; tiles past N see only zero fill.)
    const long long tiles_per_xcd = (gridDim.x + 7) >> 3;
    const long long tile_i = (long long)(blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if (tile_i >= n_tiles) return;          // the launch covers the tiles [tile_base, tile_base + n_tiles)
    const long long t0 = (tile_base + tile_i) * TILE;
    if (t0 >= N) return;
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    const char* lds_l = (const char*)lds + lane * (B64 ? 8 : 4);
    // tile sample held in accumulator slot j of this lane
    auto slot_x = [&](int j) { return B64 ? 128 * (j >> 1) + 2 * lane + (j & 1) : lane + 64 * j; };

    float best[TPW];
    int arg[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) { best[j] = best0; arg[j] = id_offset; }
    for (int x = tid; x < TILE; x += NTHREADS) lds[x] = 0.0f;  // the zero slab

    for (int g = g_lo; g < g_hi; ++g) {
        const BpGroup grp = groups[g];
        const int k_last = grp.first_src + grp.n_src - 1;
        const int k_first = grp.first_src + wv;
        using Meta = typename std::conditional<SMETA, BpMetaS<NSV>, BpMetaP<NSV>>::type;
        Meta m0, m1;
        // GLOCAL: the plan lists a group's sources by ascending id, so inside a group a plain
        // strict > keeps the lowest id on ties; the full tie rule runs once per group.
        float bestg[GLOCAL ? TPW : 1];
        int argg[GLOCAL ? TPW : 1];
        if constexpr (GLOCAL) {
#pragma unroll
            for (int j = 0; j < TPW; ++j) { bestg[j] = -INFINITY; argg[j] = 0x7fffffff; }
        }
        if constexpr (!SMETA) m0.load(srcs4, recs, min(k_first, k_last), vzero);
        __syncthreads();  // previous group's gathers are done
        // the first source's metadata travels while the windows are staged (the staging uses no
        // scalar registers of its own, so nothing tempts the compiler to spill this set in flight:
        // tools/check_inflight.py)
        if constexpr (SMETA) { if (k_first <= k_last) m0.issue(srcs4, recs, k_first); }
        // Staging, one chunk per wave and 16 bytes per lane: a chunk is <= 256 consecutive floats
        // of one prestacked row (the plan cuts windows to multiples of 4 floats at 16-byte
        // aligned LDS offsets), so it is one unaligned global_load_dwordx4 + one ds_write_b128
        // per lane.  Each wave issues the loads of STG_R chunks before it writes any of them:
        // a group of ~240 chunks is two memory round trips for the 16 waves, where the former
        // 256-thread / 4-byte version paid fifteen (staging was 7 % of the kernel at cfg3 and a
        // quarter at 80 station-phase rows).  Chunks that reach outside [0, N) -- the first and
        // last tiles of a trace -- take a per-element path with zero fill.
        {
            constexpr int STG_R = WPB >= 16 ? 8 : 4;   // 12-wave workgroups run under an 80-VGPR cap
            typedef float f32x4u4 __attribute__((ext_vector_type(4), aligned(4)));
            typedef float f32x4v __attribute__((ext_vector_type(4)));
            const int nck = grp.n_chunk;
            for (int c0 = wv; c0 < nck; c0 += WPB * STG_R) {
                int4 dsc[STG_R];
                f32x4v v[STG_R];
#pragma unroll
                for (int r = 0; r < STG_R; ++r) {
                    const int c = c0 + r * WPB;
                    // wave-uniform, but fetched with a vector load (index through the opaque zero):
                    // eight descriptors in SGPRs on top of the metadata set make the compiler
                    // spill in-flight scalar loads (tools/check_inflight.py)
                    dsc[r] = chunks[grp.first_chunk + min(c, nck - 1) + vzero];
                    if (c >= nck) dsc[r].w = 0;
                }
#pragma unroll
                for (int r = 0; r < STG_R; ++r) {
                    const long long g0 = t0 + dsc[r].y;
                    const float* src = U + (size_t)dsc[r].x * (size_t)N;
                    v[r] = (f32x4v){0.0f, 0.0f, 0.0f, 0.0f};
                    if (4 * lane < dsc[r].w) {
                        if (g0 >= 0 && g0 + 256 <= N) {     // wave-uniform: whole chunk span inside
                            v[r] = *(const f32x4u4*)(src + g0 + 4 * lane);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const long long gi = g0 + 4 * lane + e;
                                if (gi >= 0 && gi < N) v[r][e] = src[gi];
                            }
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < STG_R; ++r)
                    if (4 * lane < dsc[r].w) *(f32x4v*)(lds + dsc[r].z + 4 * lane) = v[r];
            }
        }
        if constexpr (SMETA) { if (k_first <= k_last) m0.wait(); }
        __syncthreads();

        // Gathers of one station (2 phases x 8 samples = 8 ds_read2st64_b32) are inline asm with
        // counted waits, software-pipelined one station ahead: while station s is accumulated the
        // 8 reads of station s+1 are already queued, so the LDS never drains between stations
        // (hipcc's own schedule is [16 reads][wait][16 fma] per pair of stations, which leaves the
        // LDS idle during the fma / issue phases of all 8 waves: 62 % busy).  LDS returns in
        // order: with 16 reads outstanding, lgkmcnt(8) = "station s has landed".
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const unsigned lds_lu = (unsigned)(size_t)lds_l;
#define BP_RD2(dst, addr, o0, o1) \
    asm volatile("ds_read2st64_b32 %0, %1 offset0:" #o0 " offset1:" #o1 : "=v"(dst) : "v"(addr))
#define BP_PKFMA(acc2, b2, x2) \
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2) : "v"(b2), "v"(x2))
#define BP_PKFMA_S(acc2, sp2, x2)                                                      \
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]"        \
                 : "+v"(acc2) : "s"(sp2), "v"(x2))
#define BP_RD64(dst, addr, o) \
    asm volatile("ds_read_b64 %0, %1 offset:" #o : "=v"(dst) : "v"(addr))
        // One straight-line body per station count (no control flow between the asm reads and
        // their uses: with phis in between, the compiler copies the destination registers of
        // reads that are still in flight).
        // The fma chain is asm volatile as well: plain fmaf()s are pure and get sunk below the
        // later reads at IR level (every station then needs its own 16 destination VGPRs).
        // A unit = one phase of one station = 4 reads (8 samples per lane).  Three units are kept in
        // flight ahead of the one being accumulated (lgkmcnt counts to 15: 12 younger reads may
        // stay outstanding while the wait retires the oldest 4).
        auto gather = [&](auto nst_c, auto first_c, const Meta& m, float (&acc)[TPW]) {
            constexpr int NST = decltype(nst_c)::value;
            constexpr bool FIRST = decltype(first_c)::value;  // false: continue the sums of acc
            constexpr int NU = 2 * NST, AH = 3;
            f32x2 X[4][4];  // ring of 4 units
            f32x2 ac[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                ac[jj][0] = FIRST ? 0.0f : acc[2 * jj];
                ac[jj][1] = FIRST ? 0.0f : acc[2 * jj + 1];
            }
#define BP_ISSUE_U(u)                                                                          \
    {                                                                                          \
        const unsigned o_ = m.offs((u) >> 1);                                                  \
        const unsigned a_ = lds_lu + ((((u) & 1) ? (o_ >> 16) : (o_ & 0xffffu)) << 2);         \
        if constexpr (B64) {                                                                   \
            BP_RD64(X[(u) & 3][0], a_, 0); BP_RD64(X[(u) & 3][1], a_, 512);                    \
            BP_RD64(X[(u) & 3][2], a_, 1024); BP_RD64(X[(u) & 3][3], a_, 1536);                \
        } else {                                                                               \
            BP_RD2(X[(u) & 3][0], a_, 0, 1); BP_RD2(X[(u) & 3][1], a_, 2, 3);                  \
            BP_RD2(X[(u) & 3][2], a_, 4, 5); BP_RD2(X[(u) & 3][3], a_, 6, 7);                  \
        }                                                                                      \
    }
#pragma unroll
            for (int u = 0; u < AH; ++u) BP_ISSUE_U(u)
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                f32x2 bb;
                i32x2 sp;
                if constexpr (SPAIR) { sp = m.pair(u >> 1); }
                else { const float beta = m.beta(u >> 1); bb[0] = beta; bb[1] = beta; }
                if (u + AH < NU) BP_ISSUE_U(u + AH)
                const int left = NU - 1 - u < AH ? NU - 1 - u : AH;  // units that may stay in flight
                if (left == 3) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
                else if (left == 2) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                else if (left == 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {  // phase P then phase S of a station, as the oracle
                    if constexpr (SPAIR) BP_PKFMA_S(ac[jj], sp, X[u & 3][jj]);
                    else BP_PKFMA(ac[jj], bb, X[u & 3][jj]);
                }
            }
#undef BP_ISSUE_U
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { acc[2 * jj] = ac[jj][0]; acc[2 * jj + 1] = ac[jj][1]; }
        };
        // k_next >= 0 (SMETA): refill m with that source once its own gathers are done
        constexpr int NS = SMETA ? (NSV < 16 ? NSV : 16) : NSV;  // stations per gather() call
        auto process = [&](Meta& m, bool live, int k_cur, int k_next) {
            const int nsta = live ? __builtin_amdgcn_readfirstlane(m.nsta()) : 0;
            const int sid = m.id(), tmin = m.tmin(), tmax = m.tmax();
            float acc[TPW];
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[j] = 0.0f;
#define BP_CASE(n) \
    case n: if constexpr (2 * n <= NS) gather(std::integral_constant<int, 2 * n>{}, first_c, m, acc); break;
#define BP_SWITCH(npairs)                                                                              \
    switch (npairs) { /* wave-uniform; station records come in pairs */                               \
        BP_CASE(1) BP_CASE(2) BP_CASE(3) BP_CASE(4) BP_CASE(5) BP_CASE(6) BP_CASE(7) BP_CASE(8)        \
        BP_CASE(9) BP_CASE(10) BP_CASE(11) BP_CASE(12) BP_CASE(13) BP_CASE(14) BP_CASE(15) BP_CASE(16) \
        default: break;                                                                                \
    }
            {
                constexpr std::true_type first_c{};
                BP_SWITCH((nsta < NS ? nsta : NS) >> 1)
            }
            if constexpr (SMETA && NSV > NS) {  // more than 16 stations: refill the set, keep summing
                constexpr std::false_type first_c{};
                for (int part = 1; part * NS < nsta; ++part) {
                    m.issue_part(recs, k_cur, part);
                    m.wait();
                    const int rest = nsta - part * NS;
                    BP_SWITCH((rest < NS ? rest : NS) >> 1)
                }
            }
#undef BP_SWITCH
#undef BP_CASE
            if constexpr (SMETA) m.issue(srcs4, recs, k_next);
            // strict bounds as a wave-uniform window [lo, hi) of the tile: 0 <= t + tmin and
            // t + tmax < N with t = t0 + x
            int lo = 0, hi = TILE;
            if (OOB == BPMF_BP_STRICT) {
                const long long lo64 = -(t0 + tmin), hi64 = N - t0 - tmax;
                lo = (int)(lo64 < 0 ? 0 : (lo64 > TILE ? TILE : lo64));
                hi = (int)(hi64 < 0 ? 0 : (hi64 > TILE ? TILE : hi64));
            }
            if (nsta <= 0) hi = 0;
            if (GLOCAL && lo == 0 && hi == TILE) {  // whole tile inside the bounds (wave-uniform)
#pragma unroll
                for (int j = 0; j < TPW; ++j) {
                    const bool take = acc[j] > bestg[j];
                    bestg[j] = take ? acc[j] : bestg[j];
                    argg[j] = take ? sid : argg[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < TPW; ++j) {
                    const int x = slot_x(j);
                    // bitwise, not short-circuit: keeps the epilogue free of branches
                    const bool computed = (x >= lo) & (x < hi);
                    if (GLOCAL) {
                        const bool take = computed & (acc[j] > bestg[j]);
                        bestg[j] = take ? acc[j] : bestg[j];
                        argg[j] = take ? sid : argg[j];
                    } else if (REDUCE == BPMF_BP_REDUCE_MAX) {
                        const bool take =
                            computed & ((acc[j] > best[j]) | ((acc[j] == best[j]) & (sid < arg[j])));
                        best[j] = take ? acc[j] : best[j];
                        arg[j] = take ? sid : arg[j];
                    } else {
                        const long long t = t0 + x;
                        if (live && t < N)
                            out_beam[(size_t)(sid - id_offset) * (size_t)N + t] = computed ? acc[j] : 0.0f;
                    }
                }
            }
            if constexpr (SMETA) m.wait();
        };
        if constexpr (SMETA) {
            for (int k = k_first; k <= k_last; k += WPB) process(m0, true, k, min(k + WPB, k_last));
        } else {
            for (int k = k_first; k <= k_last; k += 2 * WPB) {
                m1.load(srcs4, recs, min(k + WPB, k_last), vzero);
                process(m0, true, k, -1);
                m0.load(srcs4, recs, min(k + 2 * WPB, k_last), vzero);
                process(m1, k + WPB <= k_last, k + WPB, -1);
            }
        }
        if constexpr (GLOCAL) {
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const bool take = (bestg[j] > best[j]) | ((bestg[j] == best[j]) & (argg[j] < arg[j]));
                best[j] = take ? bestg[j] : best[j];
                arg[j] = take ? argg[j] : arg[j];
            }
        }
#undef BP_RD2
#undef BP_PKFMA_S
#undef BP_RD64
#undef BP_PKFMA
    }
    if (REDUCE == BPMF_BP_REDUCE_MAX) {
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
}

// ------------------------------------------------- multi-GPU max exchange keys ---
__device__ __forceinline__ unsigned f32_to_ordered(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_f32(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// reduce="max" of a short series computed in `n_split` group ranges per tile: fold the partial
// (beam, arg) rows -- larger beam wins, lowest source id on equal beams, the rule of every other merge
// Interior samples [lo_s, hi_s) hold `rows` partial rows (station-count classes x group ranges), the
// edge samples around them -- computed by the general kernel over all sources -- `rows_edge`.
__global__ void bp_merge_splits_kernel(const float* __restrict__ pbeam, const int* __restrict__ parg,
                                       int rows, int rows_edge, long long lo_s, long long hi_s,
                                       size_t N, float* __restrict__ beam, int* __restrict__ arg)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int n_split = ((long long)i >= lo_s && (long long)i < hi_s) ? rows : rows_edge;
    float b = pbeam[i];
    int a = parg[i];
    for (int y = 1; y < n_split; ++y) {
        const float bw = pbeam[(size_t)y * N + i];
        const int aw = parg[(size_t)y * N + i];
        if (bw > b || (bw == b && aw < a)) { b = bw; a = aw; }
    }
    beam[i] = b;
    arg[i] = a;
}

// option bp.compat_first_computed: samples on which no beam was computed still hold the start value
__global__ void bp_finish_first_computed_kernel(float* __restrict__ beam, int* __restrict__ arg, size_t N,
                                                int id_offset)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (beam[i] == -INFINITY) { beam[i] = 0.0f; arg[i] = id_offset; }
}

__global__ void bp_pack_kernel(const float* __restrict__ beam, const int* __restrict__ arg, size_t N,
                               unsigned long long flip, unsigned long long* __restrict__ packed)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    unsigned long long key = ((unsigned long long)f32_to_ordered(beam[i]) << 32) |
                             (unsigned long long)(0xffffffffu - (unsigned)arg[i]);
    packed[i] = key ^ flip;
}

__global__ void bp_unpack_kernel(const unsigned long long* __restrict__ packed, size_t N,
                                 unsigned long long flip, float* __restrict__ beam,
                                 int* __restrict__ arg)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    unsigned long long key = packed[i] ^ flip;
    beam[i] = ordered_to_f32((unsigned)(key >> 32));
    arg[i] = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
}

}  // namespace bpmf

using namespace bpmf;

// ----------------------------------------------------------------------- plan ---
namespace {

struct PlanHost {
    std::vector<BpGroup> groups;
    std::vector<BpChunk> chunks;
    std::vector<BpSource> srcs;   // in processing order
    std::vector<int> off;
    std::vector<float> beta;
    size_t lds_floats = 0;
    int NT = 4;
    int n_pass = 1;             // > 1: every group is n_pass consecutive entries of `groups` (LDS residencies)
    int per = 0;                // n_pass > 1: weighted stations of a source per residency
    int slots = 0;              // n_pass > 1: sources per wave of a group (6, or 9 when every weight is uniform)
};

// Processing order: recursive median bisection of the sources on the moveout column with
// the largest spread (a kd-tree walk), so that consecutive sources have similar moveouts
// on every station and a group's LDS windows stay short.
void bisect_order(const int32_t* mv, size_t SP, std::vector<int>& idx, size_t lo, size_t hi,
                  size_t leaf)
{
    if (hi - lo <= leaf) return;
    size_t best_col = 0;
    long long best_range = -1;
    for (size_t c = 0; c < SP; ++c) {
        int mn = mv[(size_t)idx[lo] * SP + c], mx = mn;
        for (size_t i = lo + 1; i < hi; ++i) {
            const int v = mv[(size_t)idx[i] * SP + c];
            mn = std::min(mn, v);
            mx = std::max(mx, v);
        }
        if ((long long)mx - mn > best_range) { best_range = (long long)mx - mn; best_col = c; }
    }
    if (best_range <= 0) return;
    const size_t mid = lo + (hi - lo) / 2;
    std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, [&](int a, int b) {
        const int va = mv[(size_t)a * SP + best_col], vb = mv[(size_t)b * SP + best_col];
        return va < vb || (va == vb && a < b);
    });
    bisect_order(mv, SP, idx, lo, mid, leaf);
    bisect_order(mv, SP, idx, mid, hi, leaf);
}

// Extreme moveouts of one source for the strict bound test and the number of its used (station, phase)
// terms: over the WEIGHTED stations (the build's convention, oracle/bpmf_oracle.c:bp_cpu), or -- option
// bp.compat_range_all_stations -- over all stations of a source that has at least one weighted station.
int source_tau_range(const int32_t* mv, const float* ws, size_t k, size_t S, size_t P, long long& lo, long long& hi)
{
    const bool all_stations = option(OPT_BP_COMPAT_RANGE_ALL_STATIONS) != 0;
    int n = 0;
    bool seen = false;
    lo = hi = 0;
    for (size_t s = 0; s < S; ++s) {
        const bool used = ws[k * S + s] != 0.0f;
        if (used) n += (int)P;
        else if (!all_stations) continue;
        for (size_t p = 0; p < P; ++p) {
            const long long tau = mv[(k * S + s) * P + p];
            if (!seen || tau < lo) lo = tau;
            if (!seen || tau > hi) hi = tau;
            seen = true;
        }
    }
    if (n == 0) lo = hi = 0;
    return n;
}

// Greedy grouping of consecutive sources (in processing order): a group is closed when the
// next source would push the LDS need (zero slab + sum over used rows of tile + moveout
// spread) past the soft budget.  Returns false if one source alone exceeds `hard_floats`.
// dual: every window is staged twice, the second copy shifted by one sample, both at even
// offsets; a term whose offset into the window is odd reads the shifted copy, so that ALL emitted
// offsets are even (8-byte aligned pairs for the ds_read_b64 kernel).
// `order_in`: the sources of this plan in processing order (all of them, or one station-count class).
bool build_plan(const int32_t* mv, const float* ws, const std::vector<int>& order_in, size_t S, size_t P,
                int tile, int chunk, size_t soft_floats, const size_t hard_floats, int max_group,
                int32_t id_offset, bool dual, PlanHost& ph)
{
    // a window is staged in 16-byte lanes: its length is rounded up to a multiple of 4 floats
    // (the extra samples are real data or zero fill, never addressed by a term)
    auto row_len = [&](int spread) -> size_t { return ((size_t)tile + (size_t)spread + 3) & ~(size_t)3; };
    auto row_cost = [&](int spread) -> size_t { return dual ? 2 * row_len(spread) : row_len(spread); };
    const size_t SP = S * P;
    std::vector<int> order = order_in;
    const size_t K = order.size();
    // the zero slab: one tile of the generic kernels, a fixed 512 floats in front of the
    // descriptor slab of the dual plans (bp_fast.hip, any tile)
    const size_t zero_slab = dual ? (size_t)BPF_ZERO_SLAB : (size_t)tile;

    size_t max_terms = 1;
    ph.srcs.resize(K);
    auto src_of = [&](size_t k) {
        long long lo = 0, hi = 0;
        const int n = source_tau_range(mv, ws, k, S, P, lo, hi);
        max_terms = std::max(max_terms, (size_t)n);
        return BpSource{(int)((long long)k + id_offset), (int)lo, (int)hi,
                        (n + chunk - 1) / chunk * chunk};
    };
    for (size_t q = 0; q < K; ++q) ph.srcs[q] = src_of((size_t)order[q]);
    const int NT = (int)((max_terms + chunk - 1) / chunk * chunk);
    ph.NT = NT;
    // A source's own windows (terms x tile + the zero slab) must leave room for the moveout
    // spread of a useful group.  When they do not even fit the soft budget with 16 samples of
    // spread per row (dense station weights), use the whole LDS (one workgroup per CU) instead of
    // degenerating to one source per group.  (Measured on cfg3 geometry, 10 / 15 / 20 used
    // stations: 0.35 / 0.64 / 0.91 s.)
    // dual plans (bp_fast.hip) keep 4 KB behind the zero slab for the next group's window descriptors
    // (16 bytes per window; a group has at most 2 S P windows)
    const size_t slab_extra = dual ? (size_t)4 * std::min<size_t>(BPF_DESC_MAX, (2 * S * P + 63) / 64 * 64) : 0;
    const size_t base_need = zero_slab + slab_extra + max_terms * row_cost(0);
    // More than 16 stations (P = 2: 32 terms): the packed kernel runs one 16-wave workgroup per CU.
    if (base_need + max_terms * (row_cost(16) - row_cost(0)) > soft_floats || (P == 2 && max_terms > 32))
        soft_floats = hard_floats;
    ph.off.assign(K * (size_t)NT, 0);       // padded terms read the zero slab at offset 0
    ph.beta.assign(K * (size_t)NT, 0.0f);

    std::vector<int> gmin(SP), gmax(SP), base(SP);
    std::vector<char> used(SP);
    struct RowUpdate { size_t row; int lo, hi; };
    std::vector<RowUpdate> upd;
    upd.reserve(SP);
    size_t first = 0;
    while (first < K) {
        std::fill(used.begin(), used.end(), 0);
        size_t need = zero_slab + slab_extra, q = first;  // the zero slab (+ the descriptor slab)
        for (; q < K && (int)(q - first) < max_group; ++q) {
            const size_t k = (size_t)order[q];
            size_t need2 = need;
            upd.clear();
            for (size_t s = 0; s < S; ++s) {
                if (ws[k * S + s] == 0.0f) continue;
                for (size_t p = 0; p < P; ++p) {
                    const size_t r = s * P + p;
                    const int tau = mv[(k * S + s) * P + p];
                    int lo = tau, hi = tau;
                    if (used[r]) {
                        lo = std::min(lo, gmin[r]);
                        hi = std::max(hi, gmax[r]);
                        need2 += row_cost(hi - lo) - row_cost(gmax[r] - gmin[r]);
                    } else {
                        need2 += row_cost(0);
                    }
                    upd.push_back(RowUpdate{r, lo, hi});
                }
            }
            const size_t limit = (q == first) ? hard_floats : soft_floats;
            if (need2 > limit) {
                if (q == first) return false;
                break;
            }
            for (const RowUpdate& u : upd) {
                used[u.row] = 1;
                gmin[u.row] = u.lo;
                gmax[u.row] = u.hi;
            }
            need = need2;
        }
        // close group [first, q): lay the windows out after the zero slab, cut them in chunks
        BpGroup g{(int)first, (int)(q - first), (int)ph.chunks.size(), 0};
        size_t o = zero_slab + slab_extra;
        for (size_t r = 0; r < SP; ++r) {
            base[r] = -1;
            if (!used[r]) continue;
            const int len = (int)row_len(gmax[r] - gmin[r]);
            base[r] = (int)o;
            for (int x0 = 0; x0 < len; x0 += BP_THREADS)
                ph.chunks.push_back(BpChunk{(int)r, gmin[r] + x0, (int)o + x0,
                                            std::min(BP_THREADS, len - x0)});
            if (dual) {  // the copy shifted by one sample, right behind (both bases multiples of 4)
                for (int x0 = 0; x0 < len; x0 += BP_THREADS)
                    ph.chunks.push_back(BpChunk{(int)r, gmin[r] + 1 + x0, (int)o + len + x0,
                                                std::min(BP_THREADS, len - x0)});
            }
            o += row_cost(gmax[r] - gmin[r]);
        }
        g.n_chunk = (int)ph.chunks.size() - g.first_chunk;
        ph.lds_floats = std::max(ph.lds_floats, o);
        if (dual) {  // ascending ids inside the group (see GLOCAL in bp_beam_wps2_kernel)
            std::sort(order.begin() + first, order.begin() + q);
            for (size_t qq = first; qq < q; ++qq) ph.srcs[qq] = src_of((size_t)order[qq]);
        }
        for (size_t qq = first; qq < q; ++qq) {
            const size_t k = (size_t)order[qq];
            size_t j = 0;
            for (size_t s = 0; s < S; ++s) {
                if (ws[k * S + s] == 0.0f) continue;
                for (size_t p = 0; p < P; ++p, ++j) {
                    const size_t r = s * P + p;
                    const int rel = mv[(k * S + s) * P + p] - gmin[r];
                    if (dual && (rel & 1))
                        ph.off[qq * NT + j] = base[r] + (int)row_len(gmax[r] - gmin[r]) + rel - 1;
                    else
                        ph.off[qq * NT + j] = base[r] + rel;
                    ph.beta[qq * NT + j] = ws[k * S + s];
                }
            }
        }
        ph.groups.push_back(g);
        first = q;
    }
    return true;
}

// Multi-residency plan (bp_fast.hip, HALVES): dual windows at `tile`, groups of at most `max_group`
// sources, the weighted stations of every source dealt to residencies of `per` stations each (in
// station order: the first `per`, the next `per`, ...); a group is closed when any residency's windows
// would exceed `hard_floats` of LDS.  Every group becomes ph.n_pass consecutive entries of ph.groups
// (same sources, the chunks of one residency each); ph.off holds the offsets of a source's terms inside
// the residency they belong to.
bool build_plan_halves(const int32_t* mv, const float* ws, const std::vector<int>& order_in, size_t S, size_t P,
                       int tile, int chunk, const size_t hard_floats, int max_group, int32_t id_offset,
                       int per, int n_pass, int slots, PlanHost& ph)
{
    auto row_len = [&](int spread) -> size_t { return ((size_t)tile + (size_t)spread + 3) & ~(size_t)3; };
    auto row_cost = [&](int spread) -> size_t { return 2 * row_len(spread); };
    const size_t SP = S * P;
    std::vector<int> order = order_in;
    const size_t K = order.size();
    const size_t zero_slab = (size_t)BPF_ZERO_SLAB;
    const size_t slab_extra = (size_t)4 * std::min<size_t>(BPF_DESC_MAX, (2 * S * P + 63) / 64 * 64);
    size_t max_terms = 1;
    ph = PlanHost();
    ph.n_pass = n_pass;
    ph.per = per;
    ph.slots = slots;
    ph.srcs.resize(K);
    auto src_of = [&](size_t k) {
        long long lo = 0, hi = 0;
        const int n = source_tau_range(mv, ws, k, S, P, lo, hi);
        max_terms = std::max(max_terms, (size_t)n);
        return BpSource{(int)((long long)k + id_offset), (int)lo, (int)hi, (n + chunk - 1) / chunk * chunk};
    };
    for (size_t q = 0; q < K; ++q) ph.srcs[q] = src_of((size_t)order[q]);
    if (max_terms > (size_t)per * n_pass * P) return false;
    const int NT = (int)((max_terms + chunk - 1) / chunk * chunk);
    ph.NT = NT;
    ph.off.assign(K * (size_t)NT, 0);
    ph.beta.assign(K * (size_t)NT, 0.0f);
    std::vector<std::vector<int>> gmin(n_pass, std::vector<int>(SP)), gmax(n_pass, std::vector<int>(SP)),
        base(n_pass, std::vector<int>(SP));
    std::vector<std::vector<char>> used(n_pass, std::vector<char>(SP));
    struct RowUpdate { int h; size_t row; int lo, hi; };
    std::vector<RowUpdate> upd;
    std::vector<size_t> need(n_pass), need2(n_pass);
    size_t first = 0;
    while (first < K) {
        for (int h = 0; h < n_pass; ++h) {
            std::fill(used[h].begin(), used[h].end(), 0);
            need[h] = zero_slab + slab_extra;
        }
        size_t q = first;
        for (; q < K && (int)(q - first) < max_group; ++q) {
            const size_t k = (size_t)order[q];
            need2 = need;
            upd.clear();
            int ord = 0;
            for (size_t s = 0; s < S; ++s) {
                if (ws[k * S + s] == 0.0f) continue;
                const int h = ord / per;
                ++ord;
                for (size_t p = 0; p < P; ++p) {
                    const size_t r = s * P + p;
                    const int tau = mv[(k * S + s) * P + p];
                    int lo = tau, hi = tau;
                    if (used[h][r]) {
                        lo = std::min(lo, gmin[h][r]);
                        hi = std::max(hi, gmax[h][r]);
                        need2[h] += row_cost(hi - lo) - row_cost(gmax[h][r] - gmin[h][r]);
                    } else {
                        need2[h] += row_cost(0);
                    }
                    upd.push_back(RowUpdate{h, r, lo, hi});
                    // (a row may appear in several residencies of a GROUP -- different sources count a
                    // station differently -- but only once per residency)
                }
            }
            bool fits = true;
            for (int h = 0; h < n_pass; ++h) fits = fits && need2[h] <= hard_floats;
            if (!fits) {
                if (q == first) return false;
                break;
            }
            for (const RowUpdate& u : upd) {
                used[u.h][u.row] = 1;
                gmin[u.h][u.row] = u.lo;
                gmax[u.h][u.row] = u.hi;
            }
            need = need2;
        }
        std::sort(order.begin() + first, order.begin() + q);          // ascending ids inside the group
        for (size_t qq = first; qq < q; ++qq) ph.srcs[qq] = src_of((size_t)order[qq]);
        for (int h = 0; h < n_pass; ++h) {
            BpGroup g{(int)first, (int)(q - first), (int)ph.chunks.size(), 0};
            size_t o = zero_slab + slab_extra;
            for (size_t r = 0; r < SP; ++r) {
                base[h][r] = -1;
                if (!used[h][r]) continue;
                const int len = (int)row_len(gmax[h][r] - gmin[h][r]);
                base[h][r] = (int)o;
                for (int x0 = 0; x0 < len; x0 += BP_THREADS)
                    ph.chunks.push_back(BpChunk{(int)r, gmin[h][r] + x0, (int)o + x0, std::min(BP_THREADS, len - x0)});
                for (int x0 = 0; x0 < len; x0 += BP_THREADS)
                    ph.chunks.push_back(BpChunk{(int)r, gmin[h][r] + 1 + x0, (int)o + len + x0, std::min(BP_THREADS, len - x0)});
                o += row_cost(gmax[h][r] - gmin[h][r]);
            }
            g.n_chunk = (int)ph.chunks.size() - g.first_chunk;
            ph.lds_floats = std::max(ph.lds_floats, o);
            ph.groups.push_back(g);
        }
        for (size_t qq = first; qq < q; ++qq) {
            const size_t k = (size_t)order[qq];
            size_t j = 0;
            int ord = 0;
            for (size_t s = 0; s < S; ++s) {
                if (ws[k * S + s] == 0.0f) continue;
                const int h = ord / per;
                ++ord;
                for (size_t p = 0; p < P; ++p, ++j) {
                    const size_t r = s * P + p;
                    const int rel = mv[(k * S + s) * P + p] - gmin[h][r];
                    ph.off[qq * NT + j] = (rel & 1) ? base[h][r] + (int)row_len(gmax[h][r] - gmin[h][r]) + rel - 1
                                                    : base[h][r] + rel;
                    ph.beta[qq * NT + j] = ws[k * S + s];
                }
            }
        }
        first = q;
    }
    return true;
}

// The tables of a plan travel through ONE pinned scratch buffer of the process (grow-only, under a mutex), not
// straight from the std::vectors that hold them: a pageable source makes the runtime page-lock the vector's pages for
// the copy, the vector is freed when the plan is built, and the NEXT host-pointer call of the process found its
// first piece of the day stalled for 20-40 ms (tools/probe_bp_e2e.py, profiles/r06_bp_e2e.txt: always the call
// behind the one that built a plan, never later ones).
struct PlanScratch {
    std::mutex m;
    char* p = nullptr;
    size_t cap = 0;
};
PlanScratch& plan_scratch()
{
    static PlanScratch* s = new PlanScratch();      // (leaked: pinned memory must not be freed from a static destructor)
    return *s;
}

template <typename Tv>
int upload(const std::vector<Tv>& v, Tv** d)
{
    size_t b = std::max<size_t>(v.size(), 1) * sizeof(Tv);
    BPMF_HIP_CHECK(hipMalloc((void**)d, b));
    if (v.empty()) return 0;
    const size_t bytes = v.size() * sizeof(Tv);
    PlanScratch& sc = plan_scratch();
    std::lock_guard<std::mutex> g(sc.m);
    if (sc.cap < bytes) {
        if (sc.p) (void)hipHostFree(sc.p);
        sc.p = nullptr;
        sc.cap = 0;
        const size_t want = align_up(std::max<size_t>(bytes + bytes / 4, (size_t)4 << 20), (size_t)1 << 20);
        if (hipHostMalloc((void**)&sc.p, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            sc.p = nullptr;
            // (no pinned memory to be had: the runtime's own pageable path)
            BPMF_HIP_CHECK(hipMemcpy(*d, v.data(), bytes, hipMemcpyHostToDevice));
            return 0;
        }
        sc.cap = want;
    }
    memcpy(sc.p, v.data(), bytes);
    BPMF_HIP_CHECK(hipMemcpy(*d, sc.p, bytes, hipMemcpyHostToDevice));
    return 0;
}

}  // namespace

namespace {

// ---- interior-tile fast path: host-side tables of one station-count class (bp_fast.hip) ----
struct FastHost {
    std::vector<BpFastGroup> fg;
    std::vector<BpRun> fr;
    std::vector<BpWindow> fw;
    std::vector<int> rec;
    bool uniform = true;
    int rec_dw = 0, max_sta = 0;
    size_t n_sources = 0;
};

// A source of `n` (even-padded) stations as `nparts` records of `tp` stations for the kernel of
// this tile: the smallest padded total + 2 per part, then the fewest parts.  Part sizes the kernels instantiate:
// tile 512: 4..16 even, one part; tile 256: 6..16 even; tile 128: 8, 12, 16, 20, 24.
bool fast_parts(int n, int tile, int& tp, int& nparts)
{
    static const int t512[] = {4, 6, 8, 10, 12, 14, 16}, t256[] = {6, 8, 10, 12, 14, 16}, t128[] = {8, 12, 16, 20, 24};
    const int* opts = tile == 512 ? t512 : (tile == 256 ? t256 : t128);
    const int n_opts = tile == 512 ? 7 : (tile == 256 ? 6 : 5);
    // A part boundary costs about as much as two stations at tiles 512 / 256 (header, refill
    // pipeline restart).  At tile 128 a unit is a whole quad of the record and the distance between
    // the request of a quad and its first use is (quads per part - 3) units: short parts stall on
    // the record loads (5 parts of 8 stations: 0.26 of the gather rate at cfg5's share) -- prefer
    // the longest parts.
    const int part_cost = tile == 128 ? 8 : 2;
    int best_total = 1 << 30;
    tp = 0;
    nparts = 0;
    for (int i = 0; i < n_opts; ++i) {
        const int k = std::max(1, (n + opts[i] - 1) / opts[i]);
        if (tile == 512 && k > 1) continue;
        if (k > 16) continue;
        const int total = k * opts[i] + part_cost * k;
        if (total < best_total || (total == best_total && k < nparts)) {
            best_total = total;
            tp = opts[i];
            nparts = k;
        }
    }
    return tp != 0;
}

// Relative time per time sample of the interior kernel on this plan: every group pays one staging
// round (two barriers, the window copies: ~6000 cycles measured at cfg3), every source its gathers
// (64 lanes x 8 bytes per ds_read_b64 at ~0.7 x 256 B/clk/CU, a little less on the small tiles,
// whose units carry more address arithmetic per byte), all of it amortised over `tile` samples.
double plan_cost(const PlanHost& ph, int tile)
{
    const double eff = tile == 512 ? 0.70 : (tile == 256 ? 0.66 : 0.40);   // (tile 128: measured 0.35 at 40 stations, VALU-bound)
    double cycles = 0.0;
    for (const BpGroup& g : ph.groups) {
        double terms = 0.0;
        for (int q = g.first_src; q < g.first_src + g.n_src; ++q) terms += ph.srcs[q].nterm;
        // an entry of a multi-residency plan is one residency: `per` stations of every source (padded
        // records), and a short group is padded to 16 x BPF_HALVES_SLOTS sources; ~8800 cycles between the
        // gathers of two entries were measured there (cfg5's share, 40 stations)
        if (ph.n_pass > 1) terms = 2.0 * ph.per * 16 * ph.slots;
        cycles += (ph.n_pass > 1 ? 8800.0 : 6000.0) + terms * (double)tile * 4.0 / (256.0 * eff);   // 4 gathered bytes per term and sample
    }
    return cycles / tile;
}

bool build_fast_host(const PlanHost& ph, int tile, bool allow_uniform, FastHost& fh)
{
    const int NT = ph.NT;
    fh = FastHost();
    fh.uniform = allow_uniform;
    int tp_max = 4;
    const size_t K = ph.srcs.size();
    std::vector<int> tp_of(K, 0), np_of(K, 0);
    for (size_t q = 0; q < K; ++q) {
        const BpSource& sr = ph.srcs[q];
        if (sr.nterm <= 0) continue;
        ++fh.n_sources;
        int tp, np;
        if (!fast_parts(sr.nterm / 2, tile, tp, np)) return false;
        tp_of[q] = tp;
        np_of[q] = np;
        tp_max = std::max(tp_max, tp);
        fh.max_sta = std::max(fh.max_sta, sr.nterm / 2);
        float w0 = 0.0f;
        for (int j = 0; j < NT; j += 2) {
            const float b = ph.beta[q * NT + j];
            if (b == 0.0f) continue;
            if (w0 == 0.0f) w0 = b;
            else if (b != w0) fh.uniform = false;
        }
    }
    if (fh.n_sources == 0) return false;
    const int rec_dw = (2 + 2 * tp_max + 3) / 4 * 4;
    fh.rec_dw = rec_dw;
    std::vector<int> members;
    for (const BpGroup& g : ph.groups) {
        // the group's staging chunks (pieces of <= 256 floats), merged back into whole windows
        BpFastGroup f{(int)fh.fr.size(), 0, (int)fh.fw.size(), 0};
        for (int c = g.first_chunk; c < g.first_chunk + g.n_chunk; ++c) {
            const BpChunk& ck = ph.chunks[c];
            if ((int)fh.fw.size() > f.first_win && fh.fw.back().row == ck.row &&
                fh.fw.back().gofs + fh.fw.back().len == ck.gofs && fh.fw.back().dst + fh.fw.back().len == ck.dst)
                fh.fw.back().len += ck.n;
            else
                fh.fw.push_back(BpWindow{ck.row, ck.gofs, ck.dst, ck.n});
        }
        f.n_win = (int)fh.fw.size() - f.first_win;
        if (f.n_win > BPF_DESC_MAX) return false;       // > 256 windows in a group: general kernel
        // runs of equal (tp, nparts), ascending id inside a run (the plan lists a group's sources by
        // ascending id); at most 16 x 16 combinations, most groups have one or two
        for (int np = 1; np <= 16; ++np) {
            for (int tp = 4; tp <= 24; tp += 2) {
                members.clear();
                for (int q = g.first_src; q < g.first_src + g.n_src; ++q)
                    if (tp_of[q] == tp && np_of[q] == np) members.push_back(q);
                if (members.empty()) continue;
                const size_t first_rec = fh.rec.size() / rec_dw;
                const size_t rounds = (members.size() + 15) / 16;
                fh.rec.resize(fh.rec.size() + rounds * np * 16 * rec_dw, 0);
                fh.fr.push_back(BpRun{(int)first_rec, (int)members.size(), tp, np});
                ++f.n_run;
                for (size_t m = 0; m < members.size(); ++m) {
                    const int q = members[m];
                    float w0 = 0.0f;
                    for (int j = 0; j < NT && w0 == 0.0f; j += 2) w0 = ph.beta[(size_t)q * NT + j];
                    for (int part = 0; part < np; ++part) {
                        const size_t r0 = (first_rec + ((m / 16) * np + part) * 16 + m % 16) * rec_dw;
                        fh.rec[r0] = ph.srcs[q].id;
                        fh.rec[r0 + 1] = fh.uniform ? __builtin_bit_cast(int, w0) : 0;
                        for (int i = 0; i < tp; ++i) {
                            const int st = part * tp + i;
                            const bool real = 2 * st + 1 < NT;  // beyond the term table: the zero slab, weight 0
                            const int oP = real ? ph.off[(size_t)q * NT + 2 * st] : 0;
                            const int oS = real ? ph.off[(size_t)q * NT + 2 * st + 1] : 0;
                            if (fh.uniform) {                   // LDS byte addresses of the two windows
                                fh.rec[r0 + 2 + 2 * i] = oP * 4;
                                fh.rec[r0 + 3 + 2 * i] = oS * 4;
                            } else {                            // {offs_P | offs_S << 16, weight}
                                fh.rec[r0 + 2 + 2 * i] = (int)((unsigned)oP | ((unsigned)oS << 16));
                                fh.rec[r0 + 3 + 2 * i] = real ? __builtin_bit_cast(int, ph.beta[(size_t)q * NT + 2 * st]) : 0;
                            }
                        }
                    }
                }
            }
        }
        fh.fg.push_back(f);
    }
    fh.rec.resize(fh.rec.size() + (size_t)17 * rec_dw, 0);        // one round of records (+ 1: whole s_load_dwordx8): the prefetch past the last part
    fh.fw.resize(fh.fw.size() + BPF_DESC_MAX, BpWindow{0, 0, 0, 0});   // the descriptor prefetch past the last group
    return true;
}

// Tables of a multi-residency class (build_plan_halves): per group ph.n_pass BpFastGroup entries, each
// with ONE run that lists all the group's sources (ascending id: wave w owns sources w, w + 16, ... in
// every residency -- the slots of the kernel's `carry` registers) as exactly two records of ph.per / 2
// stations.
bool build_fast_host_halves(const PlanHost& ph, FastHost& fh, bool allow_uniform)
{
    const int NT = ph.NT;
    fh = FastHost();
    fh.uniform = allow_uniform;
    const size_t K = ph.srcs.size();
    for (size_t q = 0; q < K; ++q) {
        const BpSource& sr = ph.srcs[q];
        if (sr.nterm <= 0) return false;                  // (sources without stations are not in this class)
        ++fh.n_sources;
        fh.max_sta = std::max(fh.max_sta, sr.nterm / 2);
        float w0 = 0.0f;
        for (int j = 0; j < NT; j += 2) {
            const float b = ph.beta[q * NT + j];
            if (b == 0.0f) continue;
            if (w0 == 0.0f) w0 = b;
            else if (b != w0) fh.uniform = false;
        }
    }
    const int tp = ph.per / 2, np = 2, WPB = 16, full = WPB * ph.slots;
    if ((tp != 6 && tp != 8 && tp != 10) || fh.n_sources == 0) return false;
    const int rec_dw = (2 + 2 * tp + 3) / 4 * 4;
    fh.rec_dw = rec_dw;
    for (size_t gi = 0; gi < ph.groups.size(); ++gi) {
        const BpGroup& g = ph.groups[gi];
        const int h = (int)(gi % (size_t)ph.n_pass);
        if (g.n_src > full) return false;
        const int flags = (h > 0 ? BPF_GROUP_LOAD : 0) | (h + 1 < ph.n_pass ? BPF_GROUP_STORE : 0);
        BpFastGroup f{(int)fh.fr.size(), 1 | flags, (int)fh.fw.size(), 0};
        for (int c = g.first_chunk; c < g.first_chunk + g.n_chunk; ++c) {
            const BpChunk& ck = ph.chunks[c];
            if ((int)fh.fw.size() > f.first_win && fh.fw.back().row == ck.row &&
                fh.fw.back().gofs + fh.fw.back().len == ck.gofs && fh.fw.back().dst + fh.fw.back().len == ck.dst)
                fh.fw.back().len += ck.n;
            else
                fh.fw.push_back(BpWindow{ck.row, ck.gofs, ck.dst, ck.n});
        }
        f.n_win = (int)fh.fw.size() - f.first_win;
        if (f.n_win > BPF_DESC_MAX) return false;
        const size_t first_rec = fh.rec.size() / rec_dw;
        // every wave walks exactly ph.slots sources (the kernel's slots are straight-line code):
        // a short group is padded with records of weight 0 at LDS offset 0 and id -1 (never a maximum)
        const size_t n = (size_t)g.n_src, rounds = (size_t)ph.slots;
        fh.rec.resize(fh.rec.size() + rounds * np * WPB * rec_dw, 0);
        fh.fr.push_back(BpRun{(int)first_rec, full, tp, np});
        for (size_t m = n; m < (size_t)full; ++m)
            for (int part = 0; part < np; ++part) fh.rec[(first_rec + ((m / WPB) * np + part) * WPB + m % WPB) * rec_dw] = -1;
        for (size_t m = 0; m < n; ++m) {
            const int q = g.first_src + (int)m;
            const int st_lo = h * ph.per, st_hi = std::min((h + 1) * ph.per, ph.srcs[q].nterm / 2);
            float w0 = 0.0f;
            for (int j = 0; j < NT && w0 == 0.0f; j += 2) w0 = ph.beta[(size_t)q * NT + j];
            for (int part = 0; part < np; ++part) {
                const size_t r0 = (first_rec + ((m / WPB) * np + part) * WPB + m % WPB) * rec_dw;
                fh.rec[r0] = ph.srcs[q].id;
                fh.rec[r0 + 1] = fh.uniform ? __builtin_bit_cast(int, w0) : 0;
                for (int i = 0; i < tp; ++i) {
                    const int st = st_lo + part * tp + i;
                    const bool real = st < st_hi && 2 * st + 1 < NT;    // beyond this residency: the zero slab, weight 0
                    const int oP = real ? ph.off[(size_t)q * NT + 2 * st] : 0;
                    const int oS = real ? ph.off[(size_t)q * NT + 2 * st + 1] : 0;
                    if (fh.uniform) {
                        fh.rec[r0 + 2 + 2 * i] = oP * 4;
                        fh.rec[r0 + 3 + 2 * i] = oS * 4;
                    } else {
                        fh.rec[r0 + 2 + 2 * i] = (int)((unsigned)oP | ((unsigned)oS << 16));
                        fh.rec[r0 + 3 + 2 * i] = real ? __builtin_bit_cast(int, ph.beta[(size_t)q * NT + 2 * st]) : 0;
                    }
                }
            }
        }
        fh.fg.push_back(f);
    }
    // one round of records behind the last one (the look-ahead past a wave's last part), and one more
    // record: the kernel fetches a record in whole s_load_dwordx8 (24 dwords where rec_dw is 20)
    fh.rec.resize(fh.rec.size() + (size_t)17 * rec_dw, 0);
    fh.fw.resize(fh.fw.size() + BPF_DESC_MAX, BpWindow{0, 0, 0, 0});
    return true;
}

void free_fast_class(BpFastClass& fc)
{
    (void)hipFree(fc.d_groups);
    (void)hipFree(fc.d_runs);
    (void)hipFree(fc.d_wins);
    (void)hipFree(fc.d_recs);
    fc = BpFastClass();
}

struct ClassHost {
    PlanHost ph;
    FastHost fh;
    int tile = 0;
    bool halves = false;
};

}  // namespace

extern "C" int bpmf_bp_plan_create(const int32_t* moveouts, const float* w_sources, size_t K,
                                   size_t S, size_t P, int device, int32_t source_id_offset,
                                   bpmf_bp_plan** plan_out)
{
    if (!moveouts || !w_sources || !plan_out || K == 0 || S == 0 || P == 0) {
        set_error("bpmf_bp_plan_create: bad argument");
        return -1;
    }
    if (K > 0x7fffffffull || K * S * P > 0x7fffffffffull) {
        set_error("bpmf_bp_plan_create: grid too large");
        return -1;
    }
    // options (defaults chosen on MI355X)
    const size_t soft_kb = (size_t)std::max(8, (int)option(OPT_BP_LDS_KB));
    const int max_group = std::max(1, (int)option(OPT_BP_MAX_GROUP));
    const int tpt_first = (int)option(OPT_BP_TPT);
    const int chunk = 4;  // terms gathered side by side by the generic kernel (8 measured equal)
    const bool reorder = option(OPT_BP_REORDER) != 0;
    const bool verbose = option(OPT_BP_VERBOSE) != 0;
    const size_t hard = BP_LDS_MAX / sizeof(float);
    const size_t soft = std::min(hard, soft_kb * 1024 / sizeof(float));
    const size_t SP = S * P;

    // weighted stations per source, extreme used moveouts of the grid
    std::vector<int> nsta(K, 0);
    int max_sta = 0, tmin_all = 0, tmax_all = 0;
    bool any_src = false;
    for (size_t k = 0; k < K; ++k) {
        long long lo = 0, hi = 0;
        const int n = source_tau_range(moveouts, w_sources, k, S, P, lo, hi) / (int)P;
        if (n > 0) {
            if (!any_src || lo < tmin_all) tmin_all = (int)lo;
            if (!any_src || hi > tmax_all) tmax_all = (int)hi;
            any_src = true;
        }
        nsta[k] = n;
        max_sta = std::max(max_sta, n);
    }
    // option bp.compat_strict_upper_only: "strict" tests t + tau_max < N only and a used term in front of
    // sample 0 contributes nothing.  With every used moveout >= 0 (BPMF's tables: moveouts relative to the
    // first arrival, template_search.py:212-214) that IS the default; a table with a negative used moveout
    // takes the global-memory kernel of bp_direct.hip, the one that tests every term.
    const bool upper_only = option(OPT_BP_COMPAT_STRICT_UPPER_ONLY) != 0;
    const bool upper_only_direct = upper_only && any_src && tmin_all < 0;
    // processing order of the whole grid (kd-tree walk); a class keeps its members in this order
    std::vector<int> order(K);
    for (size_t k = 0; k < K; ++k) order[k] = (int)k;
    if (reorder) bisect_order(moveouts, SP, order, 0, K, 16);

    // ---- Two-phase grids: the ds_read_b64 kernel on dual windows (bp_fast.hip).  The sources are
    // sorted into classes by their number of weighted stations -- <= 16, 17..32, 33..64 -- and
    // every class gets the largest tile (512 / 256 / 128 samples) at which the dual windows of its
    // groups fit the LDS with the lowest modelled cost; one 17-station source no longer moves a
    // whole grid off the fast path, and dense 20- or 40-station weights run it on the small tiles.
    std::vector<ClassHost> classes;
    const bool want_dual = P == 2 && option(OPT_BP_DUAL) && option(OPT_BP_PACKED) &&
                           option(OPT_BP_WPS) && tpt_first == 2;
    if (want_dual && any_src && max_sta <= 64) {
        static const int bound[4] = {0, 16, 32, 64};
        static const int cand[3][3] = {{512, 256, 128}, {256, 128, 0}, {128, 0, 0}};
        bool ok = true;
        std::vector<int> members;
        for (int c = 0; c < 3 && ok; ++c) {
            members.clear();
            for (size_t q = 0; q < K; ++q) {
                const int n = nsta[order[q]];
                // sources without any station ride along with the first class when the whole grid is
                // one class (they are skipped by the kernels; the shared plan must list every source)
                if ((n > bound[c] && n <= bound[c + 1]) || (c == 0 && n == 0 && max_sta <= 16))
                    members.push_back(order[q]);
            }
            if (members.empty()) continue;
            ClassHost best;
            double best_cost = 0.0;
            const int forced_tile = (int)option(OPT_BP_FAST_TILE);
            for (int i = 0; i < 3 && cand[c][i]; ++i) {
                if (forced_tile && cand[c][i] != forced_tile) continue;
                ClassHost ch;
                ch.tile = cand[c][i];
                if (!build_plan(moveouts, w_sources, members, S, P, ch.tile, chunk, hard, hard, max_group,
                                source_id_offset, true, ch.ph))
                    continue;
                const double cost = plan_cost(ch.ph, ch.tile);
                if (verbose)
                    fprintf(stderr, "[bpmf] bp class %d (%zu sources, %d..%d stations) tile %d: %zu groups, cost %.1f\n",
                            c, members.size(), bound[c] + 1, bound[c + 1], ch.tile, ch.ph.groups.size(), cost);
                if (!best.tile || cost < best_cost) {
                    best = std::move(ch);
                    best_cost = cost;
                }
                // groups of hundreds of sources: a smaller tile cannot win
                if ((double)members.size() / (double)best.ph.groups.size() >= 256.0 && best.tile == cand[c][i]) break;
            }
            // 33-64 stations: two LDS residencies per group at tile 256 (the station halves of every
            // source, partial beams carried in registers) against one at tile 128
            if (c == 2 && (!forced_tile || forced_tile == 256) && option(OPT_BP_HALVES) != 0) {
                ClassHost ch;
                ch.tile = 256;
                ch.halves = true;
                // 2-4 residencies of at most 20 stations, every source as two records of 6 / 8 / 10 stations in
                // each of them
                int cmax = 0;
                for (int m : members) cmax = std::max(cmax, nsta[m]);
                const int n_pass = std::max(2, (cmax + 19) / 20);
                const int tp_h = std::max(6, (((cmax + n_pass - 1) / n_pass + 1) / 2 + 1) / 2 * 2), per = 2 * tp_h;
                const int slots = BPF_HALVES_SLOTS;
                if (tp_h <= 10 &&
                    build_plan_halves(moveouts, w_sources, members, S, P, 256, chunk, hard,
                                      std::min(max_group, 16 * slots), source_id_offset, per, n_pass, slots, ch.ph) &&
                    build_fast_host_halves(ch.ph, ch.fh, option(OPT_BP_FAST_UNIFORM) != 0)) {
                    const double cost = plan_cost(ch.ph, 256);
                    if (verbose)
                        fprintf(stderr, "[bpmf] bp class %d, %d residencies of %d stations at tile 256: %zu entries, cost %.1f\n",
                                c, n_pass, per, ch.ph.groups.size(), cost);
                    if (!best.tile || cost < best_cost || forced_tile == 256) {
                        best = std::move(ch);
                        best_cost = cost;
                    }
                }
            }
            if (!best.tile || (!best.halves && !build_fast_host(best.ph, best.tile, option(OPT_BP_FAST_UNIFORM) != 0, best.fh))) {
                ok = false;
                break;
            }
            classes.push_back(std::move(best));
        }
        if (!ok) classes.clear();
    }
    // A single class at tile 512 that lists every source doubles as the plan of the general kernels
    // (their 8-byte-gather flavour): edge tiles and reduce="none" then gather 8 bytes too, and the
    // grid is planned once.  Otherwise the general kernels get their own single-window plan.
    const bool share = classes.size() == 1 && classes[0].tile == 512 && classes[0].ph.srcs.size() == K;
    const bool use_fast = !classes.empty() && option(OPT_BP_FAST) != 0;
    if (!share && !use_fast) classes.clear();

    PlanHost ph_own;
    int tpt = 0;
    bool dual = false;
    if (share) {
        tpt = 2;
        dual = true;
    } else {
        const int candidates[3] = {tpt_first, 2, 1};
        for (int c = 0; c < 3 && !tpt; ++c) {
            const int cnd = candidates[c];
            if (cnd != 1 && cnd != 2 && cnd != 4) continue;
            ph_own = PlanHost();
            if (build_plan(moveouts, w_sources, order, S, P, BP_THREADS * cnd, chunk, soft, hard,
                           max_group, source_id_offset, false, ph_own))
                tpt = cnd;
        }
    }
    PlanHost& ph = share ? classes[0].ph : ph_own;
    BPMF_BIND_DEVICE(device);
    bpmf_bp_plan* pl = new bpmf_bp_plan();
    pl->device = device;
    pl->K = K; pl->S = S; pl->P = P;
    if (!tpt || ph.NT > 256 || option(OPT_BP_DIRECT) != 0 || upper_only_direct) {
        // No LDS plan: one source's station-phase windows do not fit at the smallest tile, or a source has
        // more than 256 (station, phase) terms (or option bp.direct asks for it: the tests).  The grid runs
        // bp_direct.hip on compact term lists in the oracle's order.
        std::vector<int4> hdr(K);
        std::vector<long long> first(K + 1, 0);
        std::vector<int4> terms;
        for (size_t k = 0; k < K; ++k) {
            long long lo = 0, hi = 0;
            const int any = source_tau_range(moveouts, w_sources, k, S, P, lo, hi) > 0 ? 1 : 0;
            first[k] = (long long)terms.size();
            for (size_t s = 0; s < S; ++s) {
                const float b = w_sources[k * S + s];
                if (b == 0.0f) continue;
                for (size_t p = 0; p < P; ++p)
                    terms.push_back(make_int4((int)(s * P + p), moveouts[(k * S + s) * P + p], __builtin_bit_cast(int, b), 0));
            }
            // (strict-upper-only: the lower test always passes; the kernel drops a term in front of sample 0)
            hdr[k] = make_int4(any, upper_only ? 0 : (int)lo, (int)hi, 0);
        }
        first[K] = (long long)terms.size();
        if (terms.empty()) terms.push_back(make_int4(0, 0, 0, 0));
        pl->direct = true;
        pl->tpt = 4;
        pl->NT = (int)std::min<size_t>(SP, 0x7fffffff);
        pl->n_groups = 0;
        pl->id_offset = source_id_offset;
        pl->tmin_all = tmin_all;
        pl->tmax_all = tmax_all;
        pl->wps = 0;
        if (verbose)
            fprintf(stderr, "[bpmf] bp plan: K=%zu, no LDS plan (%s): global-memory gathers over %zu terms\n", K,
                    !tpt ? "windows exceed the LDS" : (ph.NT > 256 ? "> 256 terms per source" : "bp.direct"),
                    terms.size());
        int rcd = 0;
        if ((rcd = upload(hdr, &pl->d_dhdr)) || (rcd = upload(first, &pl->d_dfirst)) ||
            (rcd = upload(terms, &pl->d_dterms))) {
            bpmf_bp_plan_destroy(pl);
            return rcd;
        }
        *plan_out = pl;
        return 0;
    }
    pl->tpt = tpt;
    pl->chunk = chunk;
    pl->NT = ph.NT;
    pl->n_groups = (int)ph.groups.size();
    pl->lds_bytes = ph.lds_floats * sizeof(float);
    pl->id_offset = source_id_offset;
    pl->mean_group = (double)K / (double)ph.groups.size();
    pl->dual = dual;
    pl->tmin_all = tmin_all;
    pl->tmax_all = tmax_all;
    if (verbose)
        fprintf(stderr, "[bpmf] bp plan: K=%zu groups=%d (mean %.1f src) tile=%d NT=%d chunk=%d lds=%zu B dual=%d classes=%zu\n",
                K, pl->n_groups, pl->mean_group, BP_THREADS * tpt, pl->NT, chunk, pl->lds_bytes, (int)dual,
                classes.size());
    int rc = 0;
    // fast-path copy of the term table: {byte offset, weight} pairs padded to ntv per source
    const int ntv_opts[4] = {8, 16, 24, 32};
    const bool want_uv = option(OPT_BP_UVGPR) != 0;
    pl->wps = (int)option(OPT_BP_WPS);
    std::vector<BpTermV> tv;
    for (int o = 0; o < 4 && want_uv && !pl->ntv; ++o)
        if (ph.NT <= ntv_opts[o]) pl->ntv = ntv_opts[o];
    if (pl->ntv) {
        tv.assign(K * (size_t)pl->ntv, BpTermV{0, 0.0f});
        for (size_t q = 0; q < K; ++q)
            for (int j = 0; j < ph.NT; ++j)
                tv[q * pl->ntv + j] = BpTermV{ph.off[q * ph.NT + j] * 4, ph.beta[q * ph.NT + j]};
        if ((rc = upload(tv, (BpTermV**)&pl->d_termsv))) { bpmf_bp_plan_destroy(pl); return rc; }
    }
    // packed per-station records for the two-phase kernel
    if (P == 2 && ph.NT <= 64 && option(OPT_BP_PACKED)) {
        const int nsta_max = ph.NT / 2;   // NT is a multiple of 4
        const int opts[5] = {4, 8, 12, 16, 32};
        for (int o = 0; o < 5 && !pl->nsv; ++o)
            if (nsta_max <= opts[o]) pl->nsv = opts[o];
    }
    if (pl->nsv) {
        std::vector<int4> recs(K * (size_t)(pl->nsv / 2), make_int4(0, 0, 0, 0));
        std::vector<int4> hdr(K);
        for (size_t q = 0; q < K; ++q) {
            int* r = (int*)&recs[q * (pl->nsv / 2)];
            const int nterm = ph.srcs[q].nterm;  // padded to the chunk (4): pairs of terms = stations
            int nst = 0;
            for (int j = 0; j + 1 < nterm; j += 2) {
                const unsigned o0 = (unsigned)ph.off[q * ph.NT + j], o1 = (unsigned)ph.off[q * ph.NT + j + 1];
                r[2 * nst] = (int)(o0 | (o1 << 16));
                r[2 * nst + 1] = __builtin_bit_cast(int, ph.beta[q * ph.NT + j]);
                ++nst;
            }
            hdr[q] = make_int4(ph.srcs[q].id, ph.srcs[q].tmin, ph.srcs[q].tmax, (nst + 1) / 2 * 2);
        }
        if ((rc = upload(recs, &pl->d_recs)) || (rc = upload(hdr, &pl->d_hdr2))) {
            bpmf_bp_plan_destroy(pl);
            return rc;
        }
    }
    // interior-tile classes
    if (use_fast) {
        for (size_t c = 0; c < classes.size() && !rc; ++c) {
            const ClassHost& ch = classes[c];
            BpFastClass& fc = pl->cls[pl->n_classes];
            fc.tile = ch.tile;
            fc.halves = ch.halves;
            fc.n_pass = ch.halves ? ch.ph.n_pass : 1;
            fc.uniform = ch.fh.uniform;
            fc.rec_dw = ch.fh.rec_dw;
            fc.n_groups = (int)ch.fh.fg.size();
            fc.lds_bytes = ch.ph.lds_floats * sizeof(float);
            fc.n_sources = ch.fh.n_sources;
            fc.max_stations = ch.fh.max_sta;
            fc.desc_waves = (int)std::min<size_t>(BPF_DESC_MAX, (2 * S * P + 63) / 64 * 64) / 64;
            ++pl->n_classes;
            if ((rc = upload(ch.fh.fg, &fc.d_groups)) || (rc = upload(ch.fh.fr, &fc.d_runs)) ||
                (rc = upload(ch.fh.fw, &fc.d_wins)) || (rc = upload(ch.fh.rec, &fc.d_recs)))
                break;
            if (verbose)
                fprintf(stderr, "[bpmf] bp fast class %zu: tile %d, %zu sources (<= %d stations), %d groups, %zu runs, uniform=%d, rec=%d dwords\n",
                        c, fc.tile, fc.n_sources, fc.max_stations, fc.n_groups, ch.fh.fr.size(), (int)fc.uniform, fc.rec_dw);
        }
        if (rc) { bpmf_bp_plan_destroy(pl); return rc; }
        // the side stream is the device's (context.h: created once per device, never destroyed);
        // the fork / join events are the plan's own
        pl->side_stream = device_side_stream(device);
        if (!pl->side_stream) { bpmf_bp_plan_destroy(pl); return -2; }
        hipError_t e2 = hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming);
        hipError_t e3 = hipEventCreateWithFlags(&pl->ev_join, hipEventDisableTiming);
        if (e2 != hipSuccess || e3 != hipSuccess) {
            set_error("bpmf_bp_plan_create: fork / join events: %s",
                      hipGetErrorString(e2 != hipSuccess ? e2 : e3));
            bpmf_bp_plan_destroy(pl);
            return -2;
        }
        pl->fast = pl->n_classes > 0;
        pl->fast_shares_generic = share;
    }
    if ((rc = upload(ph.groups, &pl->d_groups)) || (rc = upload(ph.chunks, &pl->d_chunks)) ||
        (rc = upload(ph.srcs, &pl->d_srcs)) || (rc = upload(ph.off, &pl->d_off)) ||
        (rc = upload(ph.beta, &pl->d_beta))) {
        bpmf_bp_plan_destroy(pl);
        return rc;
    }
    *plan_out = pl;
    return 0;
}

extern "C" void bpmf_bp_plan_destroy(bpmf_bp_plan* pl)
{
    if (!pl) return;
    (void)hipFree(pl->d_dhdr);
    (void)hipFree(pl->d_dfirst);
    (void)hipFree(pl->d_dterms);
    (void)hipFree(pl->d_groups);
    (void)hipFree(pl->d_chunks);
    (void)hipFree(pl->d_srcs);
    (void)hipFree(pl->d_off);
    (void)hipFree(pl->d_beta);
    (void)hipFree(pl->d_termsv);
    (void)hipFree(pl->d_recs);
    (void)hipFree(pl->d_hdr2);
    if (pl->ev_fork) (void)hipEventDestroy(pl->ev_fork);
    if (pl->ev_join) (void)hipEventDestroy(pl->ev_join);
    for (int c = 0; c < BPF_MAX_CLASSES; ++c) free_fast_class(pl->cls[c]);
    delete pl;
}

extern "C" int bpmf_bp_plan_info(const bpmf_bp_plan* pl, bpmf_bp_plan_stats* out)
{
    if (!pl || !out) {
        set_error("bpmf_bp_plan_info: bad argument");
        return -1;
    }
    if (pl->direct) {               // no LDS plan: global-memory gathers (bp_direct.hip)
        memset(out, 0, sizeof(*out));
        out->tile = 1024;
        out->gather_bytes = 4;
        out->waves_per_cu = 8;
        return 0;
    }
    out->n_groups = pl->n_groups;
    out->tile = BP_THREADS * pl->tpt;
    out->lds_bytes = (int32_t)pl->lds_bytes;
    out->gather_bytes = pl->dual ? 8 : 4;
    out->stations_max = pl->wps ? pl->nsv : 0;
    const bool packed = pl->wps && pl->nsv && pl->tpt == 2;
    out->waves_per_cu = !packed ? 8 : (pl->nsv > 16 ? 16 : (pl->dual ? 16 : 24));
    // reduce="max": the interior tiles run the classes of bp_fast.hip (8-byte gathers, 16 waves per CU);
    // tile / n_groups then describe the class that holds most sources
    out->n_classes = pl->fast ? pl->n_classes : 0;
    for (int c = 0; c < 3; ++c) {
        const bool on = pl->fast && c < pl->n_classes;
        out->class_tile[c] = on ? pl->cls[c].tile : 0;
        out->class_sources[c] = on ? (int32_t)pl->cls[c].n_sources : 0;
        out->class_groups[c] = on ? pl->cls[c].n_groups : 0;
        out->class_stations_max[c] = on ? pl->cls[c].max_stations : 0;
    }
    if (pl->fast) {
        int big = 0;
        for (int c = 1; c < pl->n_classes; ++c)
            if (pl->cls[c].n_sources > pl->cls[big].n_sources) big = c;
        out->tile = pl->cls[big].tile;
        out->n_groups = pl->cls[big].n_groups;
        out->lds_bytes = (int32_t)pl->cls[big].lds_bytes;
        out->gather_bytes = 8;
        out->waves_per_cu = 16;
    }
    return 0;
}

namespace {
// Group ranges per tile (gridDim.y of the beam kernels).  A workgroup owns a 512-sample tile and one
// workgroup fills a CU, so a series of fewer than ~128 tiles -- the reference's event relocation
// beamforms 1 500-3 000 samples over the whole grid (BPMF/dataset.py:2174-2216) -- leaves most of the
// 256 CUs idle (and up to ~1000 tiles the last round of workgroups runs half empty): the groups of the
// plan are then dealt to 1024 / tiles workgroups per tile.
// option bp.split: 0/1 = off, n = force n ranges (tests), -1 = automatic.  Only the P = 2 packed kernels take it.
bool generic_can_split(const bpmf_bp_plan* pl)
{
    if (pl->tpt != 2 || !pl->wps || pl->n_groups < 2) return false;
    return pl->nsv == 4 || pl->nsv == 8 || pl->nsv == 12 || pl->nsv == 16 || pl->nsv == 32;   // dispatch_beam<2>'s packed kernels
}

// `forced` = option bp.split as the CALLER read it (once per call: the size check of the workspace and the
// launches must see the same value even if another thread sets the option in between)
long long split_wanted(size_t N, int forced)
{
    const long long n_tiles = (long long)((N + 511) / 512);
    // enough workgroups for ~4 rounds over the 256 CUs (a split costs one merge pass and nothing else:
    // the ranges stage disjoint windows), none from 1024 tiles (N >= 524 288) on
    long long want = n_tiles >= 1024 ? 1 : (1024 + n_tiles - 1) / n_tiles;
    if (forced >= 0) want = forced < 1 ? 1 : forced;
    return want;
}

// the general kernels alone (reduce="none", plans without interior classes)
int bp_split_count(const bpmf_bp_plan* pl, size_t N, int forced)
{
    if (!pl || !generic_can_split(pl)) return 1;
    return (int)std::max<long long>(1, std::min<long long>(split_wanted(N, forced), pl->n_groups));
}

// reduce="max" on a plan with interior classes: group ranges per tile of every class kernel, and of
// the general kernel on the edge tiles (1 when that kernel cannot split)
void bp_fast_split_counts(const bpmf_bp_plan* pl, size_t N, int forced, int& n_split, int& n_split_edge)
{
    long long want = split_wanted(N, forced);
    for (int c = 0; c < pl->n_classes; ++c) {
        want = std::min<long long>(want, pl->cls[c].n_groups / pl->cls[c].n_pass);   // (groups of sources, not entries)
    }
    const bool gsplit = generic_can_split(pl);
    if (gsplit) want = std::min<long long>(want, pl->n_groups);
    n_split = (int)std::max<long long>(1, want);
    n_split_edge = gsplit ? n_split : 1;
}
}  // namespace

namespace {
// workspace of one call under bp.split = `forced`: the prestacked traces + the partial maxima, one row per
// group range of a short series (bp_split_count) and per station-count class of the interior kernel
size_t bp_workspace_bytes(const bpmf_bp_plan* pl, size_t N, int forced)
{
    size_t rows = pl->direct ? (size_t)direct_split_count(pl, N) : (size_t)bp_split_count(pl, N, forced);
    if (pl->fast) {
        int n_split, n_split_edge;
        bp_fast_split_counts(pl, N, forced, n_split, n_split_edge);
        rows = std::max(rows, (size_t)n_split * (size_t)pl->n_classes);
    }
    return align_up(pl->S * pl->P * N * sizeof(float), 256) +
           (rows > 1 ? align_up(rows * N * (sizeof(float) + sizeof(int32_t)), 256) : 0);
}
}  // namespace

extern "C" size_t bpmf_bp_workspace_bytes(const bpmf_bp_plan* pl, size_t N, size_t C)
{
    (void)C;
    if (!pl) return 0;
    return bp_workspace_bytes(pl, N, (int)option(OPT_BP_SPLIT));
}

namespace {

// The caller (interior / edge split of bpmf_bp_run_dev) may restrict a launch of the general kernels
// to the samples [t_samp_lo, t_samp_hi) -- multiples of 1024, i.e. whole tiles of every kernel -- and
// then places the profile marks itself.  t_samp_hi < 0: the whole series.
thread_local long long t_samp_lo = 0, t_samp_hi = -1;
thread_local int t_n_split = 1;                  // group ranges per tile (short series, see bp_split_count)
thread_local long long t_split_stride = 0;       // elements between the partial outputs of reduce="max"
// start value of the running maximum: 0 (the build's convention: a beam that is not > 0 never
// becomes the maximum) or -inf (option bp.compat_first_computed: the maximum over the computed
// beams whatever their sign; samples without any computed beam are set to (0, first id) at the end)
thread_local float t_best0 = 0.0f;
}  // namespace
namespace bpmf { thread_local bool t_bp_defer_finish = false; }
namespace {

// tiles [base, base + count) of a kernel with `tile` samples per workgroup
inline void tile_range(size_t N, size_t tile, long long& base, long long& count)
{
    base = 0;
    count = (long long)((N + tile - 1) / tile);
    if (t_samp_hi >= 0) {
        const long long hi = std::min<long long>(t_samp_hi, (long long)N);
        base = t_samp_lo / (long long)tile;
        count = hi > t_samp_lo ? (hi + (long long)tile - 1) / (long long)tile - base : 0;
    }
}

template <int TPT, int CHUNK, int NBLK, int OOB, int REDUCE>
int launch_beam(const bpmf_bp_plan* pl, const float* U, size_t N, hipStream_t stream, float* beam,
                int32_t* arg)
{
    auto kern = bp_beam_kernel<TPT, CHUNK, NBLK, OOB, REDUCE>;
    if (pl->lds_bytes > 64 * 1024)
        BPMF_HIP_CHECK(hipFuncSetAttribute((const void*)kern,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)BP_LDS_MAX));
    long long tile_base, n_tiles;
    tile_range(N, (size_t)BP_THREADS * TPT, tile_base, n_tiles);
    if (n_tiles <= 0) return 0;
    dim3 grid((unsigned)n_tiles);
    if (t_samp_hi < 0) profile_mark(BPMF_KERNEL_BP_BEAM, 0, stream);
    kern<<<grid, dim3(BP_THREADS), pl->lds_bytes, stream>>>(
        U, (long long)N, pl->d_groups, pl->n_groups, (const int4*)pl->d_chunks,
        (const int*)pl->d_srcs, pl->d_off, pl->d_beta, pl->NT, pl->id_offset, beam, arg, tile_base, t_best0);
    BPMF_LAUNCH_CHECK();
    if (t_samp_hi < 0) profile_mark(BPMF_KERNEL_BP_BEAM, 1, stream);
    return 0;
}

template <int TPT, int CHUNK, int NBLK>
int dispatch_beam3(const bpmf_bp_plan* pl, const float* U, size_t N, int oob, int reduce,
                   hipStream_t stream, float* beam, int32_t* arg)
{
    if (oob == BPMF_BP_STRICT && reduce == BPMF_BP_REDUCE_MAX)
        return launch_beam<TPT, CHUNK, NBLK, BPMF_BP_STRICT, BPMF_BP_REDUCE_MAX>(pl, U, N, stream, beam, arg);
    if (oob == BPMF_BP_FLEXIBLE && reduce == BPMF_BP_REDUCE_MAX)
        return launch_beam<TPT, CHUNK, NBLK, BPMF_BP_FLEXIBLE, BPMF_BP_REDUCE_MAX>(pl, U, N, stream, beam, arg);
    if (oob == BPMF_BP_STRICT)
        return launch_beam<TPT, CHUNK, NBLK, BPMF_BP_STRICT, BPMF_BP_REDUCE_NONE>(pl, U, N, stream, beam, arg);
    return launch_beam<TPT, CHUNK, NBLK, BPMF_BP_FLEXIBLE, BPMF_BP_REDUCE_NONE>(pl, U, N, stream, beam, arg);
}

template <int TPT, int CHUNK>
int dispatch_beam2(const bpmf_bp_plan* pl, const float* U, size_t N, int oob, int reduce,
                   hipStream_t stream, float* beam, int32_t* arg)
{
    if (pl->NT <= 64) return dispatch_beam3<TPT, CHUNK, 1>(pl, U, N, oob, reduce, stream, beam, arg);
    if (pl->NT <= 128) return dispatch_beam3<TPT, CHUNK, 2>(pl, U, N, oob, reduce, stream, beam, arg);
    return dispatch_beam3<TPT, CHUNK, 4>(pl, U, N, oob, reduce, stream, beam, arg);
}

template <int TPT, int NTV, int OOB, int REDUCE>
int launch_beam_uv(const bpmf_bp_plan* pl, const float* U, size_t N, hipStream_t stream,
                   float* beam, int32_t* arg)
{
    auto kern = bp_beam_uvgpr_kernel<TPT, NTV, OOB, REDUCE>;
    if (pl->lds_bytes > 64 * 1024)
        BPMF_HIP_CHECK(hipFuncSetAttribute((const void*)kern,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)BP_LDS_MAX));
    long long tile_base, n_tiles;
    tile_range(N, (size_t)BP_THREADS * TPT, tile_base, n_tiles);
    if (n_tiles <= 0) return 0;
    dim3 grid((unsigned)n_tiles);
    if (t_samp_hi < 0) profile_mark(BPMF_KERNEL_BP_BEAM, 0, stream);
    kern<<<grid, dim3(BP_THREADS), pl->lds_bytes, stream>>>(
        U, (long long)N, pl->d_groups, pl->n_groups, (const int4*)pl->d_chunks,
        (const int4*)pl->d_srcs, (const int4*)pl->d_termsv, pl->id_offset, beam, arg, tile_base, t_best0);
    BPMF_LAUNCH_CHECK();
    if (t_samp_hi < 0) profile_mark(BPMF_KERNEL_BP_BEAM, 1, stream);
    return 0;
}

template <int TPT, int NTV>
int dispatch_beam_uv(const bpmf_bp_plan* pl, const float* U, size_t N, int oob, int reduce,
                     hipStream_t stream, float* beam, int32_t* arg)
{
    if (oob == BPMF_BP_STRICT && reduce == BPMF_BP_REDUCE_MAX)
        return launch_beam_uv<TPT, NTV, BPMF_BP_STRICT, BPMF_BP_REDUCE_MAX>(pl, U, N, stream, beam, arg);
    if (oob == BPMF_BP_FLEXIBLE && reduce == BPMF_BP_REDUCE_MAX)
        return launch_beam_uv<TPT, NTV, BPMF_BP_FLEXIBLE, BPMF_BP_REDUCE_MAX>(pl, U, N, stream, beam, arg);
    if (oob == BPMF_BP_STRICT)
        return launch_beam_uv<TPT, NTV, BPMF_BP_STRICT, BPMF_BP_REDUCE_NONE>(pl, U, N, stream, beam, arg);
    return launch_beam_uv<TPT, NTV, BPMF_BP_FLEXIBLE, BPMF_BP_REDUCE_NONE>(pl, U, N, stream, beam, arg);
}

template <int TPW, int NTV, int OOB, int REDUCE>
int launch_beam_wps(const bpmf_bp_plan* pl, const float* U, size_t N, hipStream_t stream,
                    float* beam, int32_t* arg)
{
    auto kern = bp_beam_wps_kernel<TPW, NTV, OOB, REDUCE>;
    // the end-of-kernel merge needs 2 * 4 * tile floats of LDS
    const size_t lds = std::max(pl->lds_bytes, (size_t)8 * 64 * TPW * sizeof(float));
    if (lds > 64 * 1024)
        BPMF_HIP_CHECK(hipFuncSetAttribute((const void*)kern,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)BP_LDS_MAX));
    long long tile_base, n_tiles;
    tile_range(N, (size_t)64 * TPW, tile_base, n_tiles);
    if (n_tiles <= 0) return 0;
    dim3 grid((unsigned)n_tiles);
    if (t_samp_hi < 0) profile_mark(BPMF_KERNEL_BP_BEAM, 0, stream);
    kern<<<grid, dim3(BP_THREADS), lds, stream>>>(
        U, (long long)N, pl->d_groups, pl->n_groups, (const int4*)pl->d_chunks,
        (const int4*)pl->d_srcs, (const int4*)pl->d_termsv, pl->id_offset, beam, arg, tile_base, t_best0);
    BPMF_LAUNCH_CHECK();
    if (t_samp_hi < 0) profile_mark(BPMF_KERNEL_BP_BEAM, 1, stream);
    return 0;
}

template <int TPW, int NTV>
int dispatch_beam_wps(const bpmf_bp_plan* pl, const float* U, size_t N, int oob, int reduce,
                      hipStream_t stream, float* beam, int32_t* arg)
{
    if (oob == BPMF_BP_STRICT && reduce == BPMF_BP_REDUCE_MAX)
        return launch_beam_wps<TPW, NTV, BPMF_BP_STRICT, BPMF_BP_REDUCE_MAX>(pl, U, N, stream, beam, arg);
    if (oob == BPMF_BP_FLEXIBLE && reduce == BPMF_BP_REDUCE_MAX)
        return launch_beam_wps<TPW, NTV, BPMF_BP_FLEXIBLE, BPMF_BP_REDUCE_MAX>(pl, U, N, stream, beam, arg);
    if (oob == BPMF_BP_STRICT)
        return launch_beam_wps<TPW, NTV, BPMF_BP_STRICT, BPMF_BP_REDUCE_NONE>(pl, U, N, stream, beam, arg);
    return launch_beam_wps<TPW, NTV, BPMF_BP_FLEXIBLE, BPMF_BP_REDUCE_NONE>(pl, U, N, stream, beam, arg);
}

template <int WPB, int NSV, int OOB, int REDUCE, bool SMETA, bool B64>
int launch_beam_wps2(const bpmf_bp_plan* pl, const float* U, size_t N, hipStream_t stream,
                     float* beam, int32_t* arg)
{
    auto kern = bp_beam_wps2_kernel<WPB, NSV, OOB, REDUCE, SMETA, B64>;
    const size_t lds = std::max(pl->lds_bytes, (size_t)2 * WPB * 512 * sizeof(float));
    if (lds > 64 * 1024)
        BPMF_HIP_CHECK(hipFuncSetAttribute((const void*)kern,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)BP_LDS_MAX));
    long long tile_base, n_tiles;
    tile_range(N, 512, tile_base, n_tiles);
    if (n_tiles <= 0) return 0;
    dim3 grid((unsigned)((n_tiles + 7) / 8 * 8), (unsigned)t_n_split);  // x: multiple of 8 (XCD-aware tile order)
    if (t_samp_hi < 0) profile_mark(BPMF_KERNEL_BP_BEAM, 0, stream);
    kern<<<grid, dim3(64 * WPB), lds, stream>>>(U, (long long)N, pl->d_groups, pl->n_groups,
                                                (const int4*)pl->d_chunks, pl->d_hdr2, pl->d_recs,
                                                pl->id_offset, beam, arg, tile_base, n_tiles,
                                                t_split_stride, t_best0);
    BPMF_LAUNCH_CHECK();
    if (t_samp_hi < 0) profile_mark(BPMF_KERNEL_BP_BEAM, 1, stream);
    return 0;
}

template <int WPB, int NSV, bool SMETA, bool B64 = false>
int dispatch_beam_wps2b(const bpmf_bp_plan* pl, const float* U, size_t N, int oob, int reduce,
                        hipStream_t stream, float* beam, int32_t* arg)
{
    if (oob == BPMF_BP_STRICT && reduce == BPMF_BP_REDUCE_MAX)
        return launch_beam_wps2<WPB, NSV, BPMF_BP_STRICT, BPMF_BP_REDUCE_MAX, SMETA, B64>(pl, U, N, stream, beam, arg);
    if (oob == BPMF_BP_FLEXIBLE && reduce == BPMF_BP_REDUCE_MAX)
        return launch_beam_wps2<WPB, NSV, BPMF_BP_FLEXIBLE, BPMF_BP_REDUCE_MAX, SMETA, B64>(pl, U, N, stream, beam, arg);
    if (oob == BPMF_BP_STRICT)
        return launch_beam_wps2<WPB, NSV, BPMF_BP_STRICT, BPMF_BP_REDUCE_NONE, SMETA, B64>(pl, U, N, stream, beam, arg);
    return launch_beam_wps2<WPB, NSV, BPMF_BP_FLEXIBLE, BPMF_BP_REDUCE_NONE, SMETA, B64>(pl, U, N, stream, beam, arg);
}

template <int NSV>
int dispatch_beam_wps2(const bpmf_bp_plan* pl, const float* U, size_t N, int oob, int reduce,
                       hipStream_t stream, float* beam, int32_t* arg)
{
    // NSV <= 16: metadata in SGPRs, 12 waves per workgroup, 2 workgroups (24 waves) per CU --
    // ds_read_b32-class gathers need >= 4 waves/SIMD to reach the LDS rate.  Above 16 stations
    // the SGPR set no longer fits and the VGPR-metadata variant (8 waves/CU) runs.
    const int wpb = (int)option(OPT_BP_WPB);
    if constexpr (NSV <= 16) {
        if (pl->dual) return dispatch_beam_wps2b<16, NSV, true, true>(pl, U, N, oob, reduce, stream, beam, arg);
        if (wpb == 8) return dispatch_beam_wps2b<8, NSV, true>(pl, U, N, oob, reduce, stream, beam, arg);
        return dispatch_beam_wps2b<12, NSV, true>(pl, U, N, oob, reduce, stream, beam, arg);
    }
    const int smeta = (int)option(OPT_BP_SMETA);
    if constexpr (NSV % 16 == 0) {
        if (smeta) return dispatch_beam_wps2b<16, NSV, true>(pl, U, N, oob, reduce, stream, beam, arg);
    }
    return dispatch_beam_wps2b<4, NSV, false>(pl, U, N, oob, reduce, stream, beam, arg);
}

template <int TPT>
int dispatch_beam(const bpmf_bp_plan* pl, const float* U, size_t N, int oob, int reduce,
                  hipStream_t stream, float* beam, int32_t* arg)
{
    if (pl->wps && pl->nsv && TPT == 2) {  // packed two-phase kernel, 16 waves/CU
        switch (pl->nsv) {
            case 4: return dispatch_beam_wps2<4>(pl, U, N, oob, reduce, stream, beam, arg);
            case 8: return dispatch_beam_wps2<8>(pl, U, N, oob, reduce, stream, beam, arg);
            case 12: return dispatch_beam_wps2<12>(pl, U, N, oob, reduce, stream, beam, arg);
            case 16: return dispatch_beam_wps2<16>(pl, U, N, oob, reduce, stream, beam, arg);
            case 32: return dispatch_beam_wps2<32>(pl, U, N, oob, reduce, stream, beam, arg);
            default: break;
        }
    }
    if (pl->wps && pl->ntv && TPT == 2) {  // wave-per-source layout: tile 512 = 64 lanes x 8
        switch (pl->ntv) {
            case 8: return dispatch_beam_wps<8, 8>(pl, U, N, oob, reduce, stream, beam, arg);
            case 16: return dispatch_beam_wps<8, 16>(pl, U, N, oob, reduce, stream, beam, arg);
            case 24: return dispatch_beam_wps<8, 24>(pl, U, N, oob, reduce, stream, beam, arg);
            case 32: return dispatch_beam_wps<8, 32>(pl, U, N, oob, reduce, stream, beam, arg);
            default: break;
        }
    }
    switch (pl->ntv) {  // uniform-VGPR fast path when every source has <= 32 terms
        case 8: return dispatch_beam_uv<TPT, 8>(pl, U, N, oob, reduce, stream, beam, arg);
        case 16: return dispatch_beam_uv<TPT, 16>(pl, U, N, oob, reduce, stream, beam, arg);
        case 24: return dispatch_beam_uv<TPT, 24>(pl, U, N, oob, reduce, stream, beam, arg);
        case 32: return dispatch_beam_uv<TPT, 32>(pl, U, N, oob, reduce, stream, beam, arg);
        default: break;
    }
    return dispatch_beam2<TPT, 4>(pl, U, N, oob, reduce, stream, beam, arg);
}


// prestack of the samples [t_lo, t_hi)
int launch_prestack(const float* d_features, const float* d_w_phases, size_t N, size_t C, int S, int P, float* U,
                    long long t_lo, long long t_hi, hipStream_t stream)
{
    if (t_hi <= t_lo) return 0;
    const unsigned nb = (unsigned)((t_hi - t_lo + 255) / 256);
    if (P <= 4) {
        bp_prestack_kernel<4><<<dim3(nb, (unsigned)S), dim3(256), 0, stream>>>(d_features, d_w_phases, (long long)N,
                                                                               (int)C, P, U, t_lo, t_hi);
    } else {
        bp_prestack_any_kernel<<<dim3(nb, (unsigned)(S * P)), dim3(256), 0, stream>>>(d_features, d_w_phases, (long long)N,
                                                                                     (int)C, P, U, t_lo, t_hi);
    }
    BPMF_LAUNCH_CHECK();
    return 0;
}

// The day of features of a host-pointer call, arriving in pieces (bpmf_bp_run below): need(samp_end, stream)
// returns once the work enqueued on `stream` behind it may read features and prestack of the samples
// [0, samp_end) -- it uploads the missing pieces on the copy stream (the host thread is inside the runtime's
// pageable-memory staging meanwhile; the device computes what was enqueued before) and enqueues their
// prestack on `stream` behind the piece's event.
struct BpFeed {
    virtual int need(long long samp_end, hipStream_t stream) = 0;
    virtual ~BpFeed() {}
};
thread_local BpFeed* t_bp_feed = nullptr;

}  // namespace

extern "C" int bpmf_bp_run_dev(const bpmf_bp_plan* pl, const float* d_features,
                               const float* d_w_phases, size_t N, size_t C, int out_of_bounds,
                               int reduce, void* d_workspace, size_t workspace_bytes,
                               bpmf_stream_t stream_, float* d_beam_out, int32_t* d_arg_out)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!pl || !d_features || !d_w_phases || !d_workspace || !d_beam_out || N == 0 || C == 0) {
        set_error("bpmf_bp_run_dev: bad argument");
        return -1;
    }
    if ((out_of_bounds != BPMF_BP_STRICT && out_of_bounds != BPMF_BP_FLEXIBLE) ||
        (reduce != BPMF_BP_REDUCE_MAX && reduce != BPMF_BP_REDUCE_NONE)) {
        set_error("bpmf_bp_run_dev: unknown out_of_bounds/reduce code");
        return -1;
    }
    if (reduce == BPMF_BP_REDUCE_MAX && !d_arg_out) {
        set_error("bpmf_bp_run_dev: reduce=max needs an arg-max output");
        return -1;
    }
    if (N > 0x7fffffffull) {
        set_error("bpmf_bp_run_dev: N exceeds the int32 index range");
        return -1;
    }
    // option bp.split is read ONCE: the size check and every launch below use this value
    const int forced_split = (int)option(OPT_BP_SPLIT);
    if (workspace_bytes < bp_workspace_bytes(pl, N, forced_split)) {
        set_error("bpmf_bp_run_dev: workspace too small (%zu < %zu)", workspace_bytes,
                  bp_workspace_bytes(pl, N, forced_split));
        return -1;
    }
    // option debug.poison_output (tests): a sample that no kernel writes comes back as NaN / -1 instead of
    // whatever the caller's buffer held
    if (reduce == BPMF_BP_REDUCE_MAX && option(OPT_DEBUG_POISON_OUTPUT) != 0) {
        BPMF_HIP_CHECK(hipMemsetAsync(d_beam_out, 0xFF, N * sizeof(float), stream));
        BPMF_HIP_CHECK(hipMemsetAsync(d_arg_out, 0xFF, N * sizeof(int32_t), stream));
    }
    float* U = (float*)d_workspace;
    const int P = (int)pl->P, S = (int)pl->S;
    // A host-pointer call may stream its day of features in while the kernels run (BpFeed, bpmf_bp_run):
    // the feed uploads AND prestacks piece by piece; only the interior-tile path below consumes it in
    // pieces, every other path asks for the whole series first.
    BpFeed* feed = t_bp_feed;
    // option bp.host_piece_samples: samples of the first piece (default 131 072 = one round of the chip at tile
    // 512; the tests shrink it), 0 = the whole day in front of the first kernel
    const long long piece0 = (long long)align_up((size_t)option(OPT_BP_HOST_PIECE_SAMPLES), 1024);
    const bool feed_pieces = feed && pl->fast && !pl->direct && reduce == BPMF_BP_REDUCE_MAX && piece0 > 0;
    if (feed && !feed_pieces)
        if (int rc = feed->need((long long)N, stream)) return rc;
    if (!feed)
        if (int rc = launch_prestack(d_features, d_w_phases, N, C, S, P, U, 0, (long long)N, stream)) return rc;
    struct SplitScope {          // the launchers read the thread-local pair; always reset on the way out
        SplitScope(int n, long long stride) { t_n_split = n; t_split_stride = stride; }
        ~SplitScope() { t_n_split = 1; t_split_stride = 0; t_samp_hi = -1; t_samp_lo = 0; }
    };
    const bool first_computed = reduce == BPMF_BP_REDUCE_MAX && option(OPT_BP_COMPAT_FIRST_COMPUTED) != 0;
    struct Best0Scope {
        explicit Best0Scope(float v) { t_best0 = v; }
        ~Best0Scope() { t_best0 = 0.0f; }
    } best0_scope(first_computed ? -INFINITY : 0.0f);
    auto finish = [&](float* beam, int32_t* arg) -> int {
        if (first_computed && !t_bp_defer_finish) {
            bp_finish_first_computed_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream>>>(
                beam, arg, N, pl->id_offset);
            BPMF_LAUNCH_CHECK();
        }
        return 0;
    };
    float* const beam_final = d_beam_out;
    int32_t* const arg_final = d_arg_out;
    char* const part = (char*)d_workspace + align_up((size_t)S * P * N * sizeof(float), 256);
    if (pl->direct) {
        // no LDS plan: global-memory gathers, ranges of sources per tile folded by the merge kernel
        const int rows = reduce == BPMF_BP_REDUCE_MAX ? direct_split_count(pl, N) : 1;
        float* pbeam = rows > 1 ? (float*)part : beam_final;
        int32_t* parg = rows > 1 ? (int32_t*)(part + (size_t)rows * N * sizeof(float)) : arg_final;
        profile_mark(BPMF_KERNEL_BP_BEAM, 0, stream);
        int rc = launch_beam_direct(pl, U, N, out_of_bounds, reduce, stream, pbeam, parg, rows, (long long)N, t_best0);
        if (!rc && rows > 1) {
            bp_merge_splits_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream>>>(
                pbeam, parg, rows, rows, 0, (long long)N, N, beam_final, arg_final);
            BPMF_LAUNCH_CHECK();
        }
        if (!rc && reduce == BPMF_BP_REDUCE_MAX) rc = finish(beam_final, arg_final);
        profile_mark(BPMF_KERNEL_BP_BEAM, 1, stream);
        return rc;
    }
    if (pl->fast && reduce == BPMF_BP_REDUCE_MAX) {
        // Samples on which no source can leave the trace -- t + tmin_all >= 0 and t + tmax_all (+ the
        // staging slack of 8 samples) < N, rounded to multiples of 1024 (whole tiles of every kernel) --
        // run the interior kernel of bp_fast.hip, once per station-count class, which is the same for
        // strict and flexible; the few tiles at the ends of the day run the general kernel over all
        // sources.  Several classes, or several group ranges per tile on a short series, write
        // partial rows behind the prestack, folded by one merge launch (value, then lowest id).
        int n_split, n_split_edge;
        bp_fast_split_counts(pl, N, forced_split, n_split, n_split_edge);
        const int rows = n_split * pl->n_classes;
        long long lo_s = pl->tmin_all < 0 ? ((long long)(-pl->tmin_all) + 1023) / 1024 * 1024 : 0;
        long long hi_s = ((long long)N - pl->tmax_all - 8) / 1024 * 1024;
        if ((long long)N - pl->tmax_all - 8 < 0) hi_s = 0;
        // The interior range ends at a WHOLE tile of every kernel inside the series.  (Rounds 2-3 clamped it
        // to N: when every used moveout is negative -- tmax_all < -8 -- N - tmax_all - 8 exceeds N, the clamp
        // left a bound that is no multiple of the tile, the interior launch stopped at the last whole tile
        // below it and the edge launch, starting AT the bound, was empty: the samples of the last partial
        // tile were never written.  Found by the 150 000-case session of round 4, seeds 15831 and 17025 of
        // test_bp_random_shapes_signed_moveouts; pinned by test_bp_all_used_moveouts_negative.)
        hi_s = std::min(hi_s, (long long)N / 1024 * 1024);
        lo_s = std::min(lo_s, (long long)N);
        hi_s = std::max(lo_s, hi_s);
        float* pbeam = rows > 1 ? (float*)part : beam_final;
        int32_t* parg = rows > 1 ? (int32_t*)(part + (size_t)rows * N * sizeof(float)) : arg_final;
        std::lock_guard<std::mutex> enqueue_lock(pl->enqueue_mutex);
        profile_mark(BPMF_KERNEL_BP_BEAM, 0, stream);
        int rc = 0;
        const bool have_edge = lo_s > 0 || hi_s < (long long)N;
        hipStream_t es = stream;          // edge tiles: on the side stream, beside the interior kernels
        if (have_edge && pl->side_stream && !feed_pieces) {
            BPMF_HIP_CHECK(hipEventRecord(pl->ev_fork, stream));
            BPMF_HIP_CHECK(hipStreamWaitEvent(pl->side_stream, pl->ev_fork, 0));
            es = pl->side_stream;
        }
        auto run_edges = [&]() {
            SplitScope scope(n_split_edge, rows > 1 ? (long long)N : 0);
            auto edge = [&](long long from, long long to) {
                if (to <= from || rc) return;
                t_samp_lo = from; t_samp_hi = to;
                switch (pl->tpt) {
                    case 1: rc = dispatch_beam<1>(pl, U, N, out_of_bounds, reduce, es, pbeam, parg); break;
                    case 2: rc = dispatch_beam<2>(pl, U, N, out_of_bounds, reduce, es, pbeam, parg); break;
                    default: rc = dispatch_beam<4>(pl, U, N, out_of_bounds, reduce, es, pbeam, parg); break;
                }
                t_samp_hi = -1;
            };
            edge(0, lo_s);
            edge(hi_s, (long long)N);
        };
        // (a day arriving in pieces: the edge tiles need both ends of the series -- they are forked to the side
        // stream in front of the LAST interior piece, once everything has been asked for, and run beside it)
        bool edges_done = false;
        auto edges_of_a_fed_day = [&]() -> int {
            edges_done = true;
            if (int r = feed->need((long long)N, stream)) return r;
            if (have_edge && pl->side_stream) {
                BPMF_HIP_CHECK(hipEventRecord(pl->ev_fork, stream));
                BPMF_HIP_CHECK(hipStreamWaitEvent(pl->side_stream, pl->ev_fork, 0));
                es = pl->side_stream;
            }
            run_edges();
            if (!rc && es != stream) BPMF_HIP_CHECK(hipEventRecord(pl->ev_join, es));
            return rc;
        };
        if (!feed_pieces) {
            run_edges();
            if (!rc && es != stream) BPMF_HIP_CHECK(hipEventRecord(pl->ev_join, es));
        }
        // The interior tiles: one launch per class -- or, while the day is still arriving from the host
        // (feed_pieces), one launch per class and PIECE of the range, each behind the piece of features it
        // reads.  A piece is a whole number of rounds of the chip (256 workgroups of 512 samples, one per CU):
        // 1, 2, then 4 rounds -- the first kernels start after 1 % of the upload, and a launch boundary costs
        // no partly filled round.
        long long a = lo_s;
        long long piece = piece0;
        while (a < hi_s && !rc) {
            long long b = hi_s;
            if (feed_pieces) {
                b = std::min(hi_s, a + piece);
                if (hi_s - b < piece0) b = hi_s;          // no sliver at the end
                piece = std::min<long long>(piece * 2, 4 * piece0);
                if (b == hi_s) rc = edges_of_a_fed_day();
                else rc = feed->need(std::min<long long>((long long)N, b + std::max(pl->tmax_all, 0) + 8 + 1024), stream);
            }
            for (int c = 0; c < pl->n_classes && !rc; ++c) {
                const BpFastClass& fc = pl->cls[c];
                rc = launch_beam_fast(fc, pl->id_offset, U, N, a / fc.tile, b / fc.tile, stream,
                                      pbeam + (size_t)c * n_split * N, parg + (size_t)c * n_split * N, n_split,
                                      rows > 1 ? (long long)N : 0, t_best0);
            }
            a = b;
        }
        if (feed_pieces && !rc && !edges_done) rc = edges_of_a_fed_day();     // (no interior tile at all)
        if (!rc && es != stream) BPMF_HIP_CHECK(hipStreamWaitEvent(stream, pl->ev_join, 0));
        if (!rc && rows > 1) {
            bp_merge_splits_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream>>>(
                pbeam, parg, rows, n_split_edge, lo_s, hi_s, N, beam_final, arg_final);
            BPMF_LAUNCH_CHECK();
        }
        if (!rc) rc = finish(beam_final, arg_final);
        profile_mark(BPMF_KERNEL_BP_BEAM, 1, stream);
        return rc;
    }
    // the general kernels over the whole series.  Short series: several group ranges per tile
    // (bp_split_count); reduce="max" goes through partial rows and one merge launch
    const int n_split = bp_split_count(pl, N, forced_split);
    SplitScope split_scope(n_split, n_split > 1 && reduce == BPMF_BP_REDUCE_MAX ? (long long)N : 0);
    if (n_split > 1 && reduce == BPMF_BP_REDUCE_MAX) {
        d_beam_out = (float*)part;
        d_arg_out = (int32_t*)(part + (size_t)n_split * N * sizeof(float));
    }
    auto merge_splits = [&]() -> int {
        if (n_split > 1 && reduce == BPMF_BP_REDUCE_MAX) {
            bp_merge_splits_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream>>>(
                d_beam_out, d_arg_out, n_split, n_split, 0, (long long)N, N, beam_final, arg_final);
            BPMF_LAUNCH_CHECK();
        }
        return 0;
    };
    int rc;
    switch (pl->tpt) {
        case 1: rc = dispatch_beam<1>(pl, U, N, out_of_bounds, reduce, stream, d_beam_out, d_arg_out); break;
        case 2: rc = dispatch_beam<2>(pl, U, N, out_of_bounds, reduce, stream, d_beam_out, d_arg_out); break;
        default: rc = dispatch_beam<4>(pl, U, N, out_of_bounds, reduce, stream, d_beam_out, d_arg_out); break;
    }
    if (!rc) rc = merge_splits();
    if (!rc && reduce == BPMF_BP_REDUCE_MAX) rc = finish(beam_final, arg_final);
    return rc;
}

namespace {
// Plans kept by the host-pointer entry point, keyed by (device, shapes, both tables).  A hit
// compares the tables themselves (host copies are kept while they are small; above 1 GiB a
// second, independent 64-bit hash stands in).  One slot per visible device and one spare each, so
// that a process driving every GPU of a node keeps all of its plans from one day to the next.
struct PlanCacheEntry {
    uint64_t key = 0, key2 = 0;
    size_t K = 0, S = 0, P = 0;
    int device = 0;
    std::vector<int32_t> mv;    // empty: table too large to keep, (key, key2) decide
    std::vector<float> ws;
    bpmf_bp_plan* pl = nullptr;
    uint64_t stamp = 0;         // last use (eviction = least recently used)
};
std::vector<PlanCacheEntry> g_plan_cache;
uint64_t g_plan_cache_clock = 0;
std::mutex g_plan_cache_mutex;
constexpr size_t PLAN_KEEP_BYTES = (size_t)1 << 30;
// 64-bit multiply-xorshift over the bytes of a table, 8 at a time (not cryptographic: a cache key)
uint64_t hash_words(const void* p, size_t bytes, uint64_t seed)
{
    const unsigned char* b = (const unsigned char*)p;
    uint64_t h = seed ^ (bytes * 0x9e3779b97f4a7c15ull);
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t w;
        memcpy(&w, b + i, 8);
        h = (h ^ w) * 0xff51afd7ed558ccdull;
        h ^= h >> 32;
    }
    uint64_t w = 0;
    if (i < bytes) memcpy(&w, b + i, bytes - i);
    h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
    return h ^ (h >> 29);
}
size_t plan_cache_capacity()
{
    int n = 0;
    if (device_counts(&n, nullptr) != hipSuccess || n < 1) n = 1;
    return (size_t)std::max(4, 2 * n);
}
}  // namespace

static int bpmf_bp_run_impl(const float* features, const int32_t* moveouts, const float* w_phases,
                           const float* w_sources, size_t N, size_t K, size_t S, size_t C, size_t P,
                           int out_of_bounds, int reduce, int device, float* beam_out,
                           int32_t* arg_out)
{
    if (!features || !moveouts || !w_phases || !w_sources || !beam_out) {
        set_error("bpmf_bp_run: null pointer");
        return -1;
    }
    // Every call binds its thread to `device` first: the plan cache below may skip
    // bpmf_bp_plan_create, and a fresh host thread starts on device 0.
    BPMF_BIND_DEVICE(device);
    t_call_stats = HostCallStats();
    const double t_call0 = host_now_ms();
    // One host-pointer call per device at a time, on the device's own streams and working set
    // (context.h): nothing is created or destroyed per call, nothing runs on the null stream.
    DeviceContext* ctx = device_context(device);
    if (!ctx) return -2;
    std::lock_guard<std::mutex> call_lock(ctx->call_mutex);
    FanoutScope fan;      // (behind the lock: a source waits for its peers before another call may touch its data)
    // BPMF calls beamform once per day with the same moveout table and source weights
    // (template_search.py:549-558); building the plan costs 0.04 s for 50 000 sources but 3 s for a
    // million, so the last plans are kept.
    bpmf_bp_plan* pl = nullptr;
    const size_t b_mv = K * S * P * sizeof(int32_t), b_wsrc = K * S * sizeof(float);
    // (the plan generation is part of the key: a plan is built under the options of its creation,
    // and bpmf_set_option must not leave a plan of the previous settings in use)
    // (tables small enough to be kept are COMPARED on a hit: their keys only pre-select, and hash one 8-byte word
    // in 64 -- 24 MB of hashing per cfg3 call were 4 ms of a 160 ms call; larger tables are hashed in full, twice)
    const bool keep_tables = b_mv + b_wsrc <= PLAN_KEEP_BYTES;
    auto table_hash = [&](const void* p, size_t bytes, uint64_t seed) -> uint64_t {
        if (!keep_tables || bytes < 4096) return hash_words(p, bytes, seed);
        uint64_t h = seed ^ bytes;
        const unsigned char* b = (const unsigned char*)p;
        for (size_t i = 0; i + 8 <= bytes; i += 512) {
            uint64_t w8;
            memcpy(&w8, b + i, 8);
            h = (h ^ w8) * 0xff51afd7ed558ccdull;
            h ^= h >> 32;
        }
        return h ^ hash_words(b + bytes - 64, 64, seed);
    };
    const uint64_t key = table_hash(moveouts, b_mv, 0x9e3779b97f4a7c15ull ^ (K * 31 + S * 7 + P)) ^
                         table_hash(w_sources, b_wsrc, 0xc2b2ae3d27d4eb4full) ^
                         (option_generation() * 0xd6e8feb86659fd93ull);
    const uint64_t key2 = keep_tables ? key * 0x9e3779b97f4a7c15ull
                                      : hash_words(moveouts, b_mv, 0x165667b19e3779f9ull) +
                                        hash_words(w_sources, b_wsrc, 0x27d4eb2f165667c5ull);
    {
        std::lock_guard<std::mutex> g(g_plan_cache_mutex);
        for (auto& e : g_plan_cache) {
            if (!e.pl || e.key != key || e.key2 != key2 || e.K != K || e.S != S || e.P != P ||
                e.device != device)
                continue;
            if (!e.mv.empty() && (memcmp(e.mv.data(), moveouts, b_mv) != 0 ||
                                  memcmp(e.ws.data(), w_sources, b_wsrc) != 0))
                continue;
            pl = e.pl;
            e.pl = nullptr;            // taken out while in use; put back below
            e.stamp = ~0ull;           // ... into this very slot (marks it as reserved)
            break;
        }
    }
    if (!pl)
        if (int rc = bpmf_bp_plan_create(moveouts, w_sources, K, S, P, device, 0, &pl)) return rc;
    t_call_stats.plan_ms = host_now_ms() - t_call0;
    auto release_plan = [&]() {
        std::lock_guard<std::mutex> g(g_plan_cache_mutex);
        const size_t cap = plan_cache_capacity();
        PlanCacheEntry* slot = nullptr;
        for (auto& e : g_plan_cache)   // the slot this plan came from still holds its tables
            if (!e.pl && e.stamp == ~0ull && e.key == key && e.key2 == key2 && e.device == device &&
                e.K == K && e.S == S && e.P == P) {
                e.pl = pl;
                e.stamp = ++g_plan_cache_clock;
                return;
            }
        for (auto& e : g_plan_cache)
            if (!e.pl && e.stamp != ~0ull) { slot = &e; break; }
        if (!slot && g_plan_cache.size() < cap) {
            g_plan_cache.emplace_back();
            slot = &g_plan_cache.back();
        }
        if (!slot) {
            // the least recently used plan OF THIS DEVICE goes (its calls are serialised by the
            // context mutex this thread holds, so nobody can be about to take it); a cache filled
            // by other devices' plans is left alone
            for (auto& e : g_plan_cache)
                if (e.pl && e.device == device && (!slot || e.stamp < slot->stamp)) slot = &e;
            if (!slot) {
                bpmf_bp_plan_destroy(pl);
                return;
            }
            bpmf_bp_plan_destroy(slot->pl);
        }
        slot->key = key; slot->key2 = key2;
        slot->K = K; slot->S = S; slot->P = P;
        slot->device = device;
        slot->pl = pl;
        slot->stamp = ++g_plan_cache_clock;
        if (keep_tables) {
            slot->mv.assign(moveouts, moveouts + K * S * P);
            slot->ws.assign(w_sources, w_sources + K * S);
        } else {
            slot->mv.clear(); slot->mv.shrink_to_fit();
            slot->ws.clear(); slot->ws.shrink_to_fit();
        }
    };
    const size_t b_f = S * C * N * sizeof(float), b_wp = S * C * P * sizeof(float),
                 b_ws = bpmf_bp_workspace_bytes(pl, N, C),
                 b_beam = (reduce == BPMF_BP_REDUCE_MAX ? N : K * N) * sizeof(float),
                 b_arg = N * sizeof(int32_t);
    const size_t o_f = 0, o_wp = o_f + align_up(b_f, 256), o_ws = o_wp + align_up(b_wp, 256),
                 o_beam = o_ws + align_up(b_ws, 256), o_arg = o_beam + align_up(b_beam, 256),
                 total = o_arg + b_arg;
    const double t_res0 = host_now_ms();
    char* base = ctx->reserve_device(total);
    if (!base || ctx->reserve_pinned(std::min<size_t>((size_t)64 << 20, std::max<size_t>(b_f, 4096)))) {
        release_plan();
        return -2;
    }
    t_call_stats.reserve_ms = host_now_ms() - t_res0;
    int rc = 0;
    hipStream_t stream = ctx->s_run;
    hipError_t e = hipSuccess;
    auto fail = [&](hipError_t err, const char* what) {
        set_error("bpmf_bp_run: %s failed: %s", what, hipGetErrorString(err));
        rc = -2;
    };
    // (pageable memory through the runtime's own staging, on the private stream: a hand-made pipeline through
    // the context's pinned pieces filled by 8 host threads measured SLOWER -- cfg3 end to end 194 ms against
    // 176 ms -- the host-side memcpy into the pinned pieces is the bottleneck on the 16 CPUs a box grants)
    // The day of features.  A peer of a multi-device call copies it from the first device.  Everybody else
    // uploads it from the host IN PIECES on the copy stream while the kernels of the pieces that have arrived
    // run (HostFeed: bpmf_bp_run_dev asks for the samples it is about to read) -- the upload of a whole day
    // in front of the first kernel cost cfg3 a third of its time (201.7 ms end to end against 150.6 resident,
    // round-4 bench; BPMF makes exactly this call, template_search.py:549-558).  The pieces travel through the
    // context's pinned buffers (staged_upload_rows, context.h).
    struct HostFeed : BpFeed {
        DeviceContext* ctx; FanoutScope* fan; const float* host; char* d_feat; const float* d_wp; float* U;
        size_t N, C; int S, P; long long have = 0; int n_piece = 0; hipError_t err = hipSuccess; const char* what = "";
        bool published = false;
        double t_call0 = 0.0;
        int need(long long samp_end, hipStream_t stream) override
        {
            samp_end = std::min<long long>((samp_end + 1023) / 1024 * 1024, (long long)N);
            if (samp_end <= have) return 0;
            const size_t rows = (size_t)S * C;
            what = "H2D features";
            // rows x [have, samp_end), through the context's pinned pieces (context.h: staged_upload_rows)
            err = staged_upload_rows(ctx, (float*)d_feat, host, rows, N, (size_t)have, (size_t)samp_end, ctx->s_copy);
            hipEvent_t ev = ctx->ev_chunk[n_piece++ % DeviceContext::CHUNK_EVENTS];
            if (err == hipSuccess) { what = "event record"; err = hipEventRecord(ev, ctx->s_copy); }
            if (err == hipSuccess) { what = "wait event"; err = hipStreamWaitEvent(stream, ev, 0); }
            if (err == hipSuccess && samp_end == (long long)N && !published) {
                published = true;               // the whole day is on its way: the other devices may copy it
                what = "event record";
                err = fanout_publish(*fan, ctx, d_feat, ctx->s_copy);
            }
            if (err != hipSuccess) {
                set_error("bpmf_bp_run: %s failed: %s", what, hipGetErrorString(err));
                return -2;
            }
            const int rc = launch_prestack((const float*)d_feat, d_wp, N, C, S, P, U, have, samp_end, stream);
            if (have == 0) t_call_stats.first_kernel_ms = host_now_ms() - t_call0;      // the first kernels follow
            have = samp_end;
            return rc;
        }
    } host_feed;
    bool from_peer = false;
    {
        const char* what = "";
        from_peer = fanout_peer_copy(fan, ctx, base + o_f, b_f, stream, &e, &what);
        if (from_peer && e != hipSuccess) fail(e, what);
    }
    if (!rc && (e = staged_upload_rows(ctx, (float*)(base + o_wp), w_phases, 1, b_wp / 4, 0, b_wp / 4, stream)) != hipSuccess) fail(e, "H2D weights_phases");
    if (!rc && reduce == BPMF_BP_REDUCE_NONE &&
        (e = hipMemsetAsync(base + o_beam, 0, b_beam, stream)) != hipSuccess) fail(e, "memset");
    if (!rc) {
        host_feed.ctx = ctx; host_feed.fan = &fan; host_feed.host = features; host_feed.d_feat = base + o_f;
        host_feed.d_wp = (const float*)(base + o_wp); host_feed.U = (float*)(base + o_ws);
        host_feed.N = N; host_feed.C = C; host_feed.S = (int)S; host_feed.P = (int)P; host_feed.t_call0 = t_call0;
        if (from_peer) t_call_stats.first_kernel_ms = host_now_ms() - t_call0;
        struct FeedScope {
            explicit FeedScope(BpFeed* f) { t_bp_feed = f; }
            ~FeedScope() { t_bp_feed = nullptr; }
        } feed_scope(from_peer ? nullptr : &host_feed);
        rc = bpmf_bp_run_dev(pl, (const float*)(base + o_f), (const float*)(base + o_wp), N, C,
                             out_of_bounds, reduce, base + o_ws, b_ws, stream,
                             (float*)(base + o_beam), (int32_t*)(base + o_arg));
    }
    // The results' way back, through the pinned pieces (staged_download: the runtime's pageable path page-locks a
    // destination it has not seen before, and a result array is a new allocation on every call)
    const double t_enqueued = host_now_ms();
    if (!rc && (e = staged_download(ctx, beam_out, base + o_beam, b_beam, stream)) != hipSuccess) fail(e, "D2H beam");
    if (!rc && reduce == BPMF_BP_REDUCE_MAX && arg_out &&
        (e = staged_download(ctx, arg_out, base + o_arg, b_arg, stream)) != hipSuccess) fail(e, "D2H argmax");
    // (always drained, also after a failure: the working set and the plan go back to their caches)
    hipError_t es = hipStreamSynchronize(stream);
    (void)hipStreamSynchronize(ctx->s_copy);
    if (pl->side_stream) (void)hipStreamSynchronize(pl->side_stream);
    t_call_stats.device_wait_ms = host_now_ms() - t_enqueued;
    t_call_stats.total_ms = host_now_ms() - t_call0;
    if (!rc && es != hipSuccess) fail(es, "synchronize");
    release_plan();
    copy_pool_quiesce();  // no host thread of the copy pool still reads the caller's arrays (a straggler of an idempotent fill) when the call returns
    fan.finish();         // a source's peers are through with its copy of the day before the working set may go
    ctx->trim_after_call();
    return rc;
}

extern "C" int bpmf_bp_run(const float* features, const int32_t* moveouts, const float* w_phases,
                           const float* w_sources, size_t N, size_t K, size_t S, size_t C, size_t P,
                           int out_of_bounds, int reduce, int device, float* beam_out,
                           int32_t* arg_out)
{
    // nothing may cross the C boundary as an exception (std::bad_alloc from the host-side planning, a
    // std::system_error): it becomes status -3 with its text
    try {
        return bpmf_bp_run_impl(features, moveouts, w_phases, w_sources, N, K, S, C, P, out_of_bounds, reduce, device, beam_out, arg_out);
    } catch (const std::exception& e) {
        copy_pool_quiesce();      // (no pool thread may still read the caller's arrays)
        set_error("bpmf_bp_run: exception: %s", e.what());
        return -3;
    } catch (...) {
        copy_pool_quiesce();
        set_error("bpmf_bp_run: unknown exception");
        return -3;
    }
}

extern "C" int bpmf_bp_pack_max_dev(const float* d_beam, const int32_t* d_arg, size_t N,
                                    int as_signed, bpmf_stream_t stream, uint64_t* d_packed)
{
    if (!d_beam || !d_arg || !d_packed) { set_error("bpmf_bp_pack_max_dev: null pointer"); return -1; }
    if (N == 0) return 0;
    bp_pack_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        d_beam, d_arg, N, as_signed ? 0x8000000000000000ull : 0ull, (unsigned long long*)d_packed);
    BPMF_LAUNCH_CHECK();
    return 0;
}

extern "C" int bpmf_bp_unpack_max_dev(const uint64_t* d_packed, size_t N, int as_signed,
                                      bpmf_stream_t stream, float* d_beam, int32_t* d_arg)
{
    if (!d_beam || !d_arg || !d_packed) { set_error("bpmf_bp_unpack_max_dev: null pointer"); return -1; }
    if (N == 0) return 0;
    bp_unpack_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        (const unsigned long long*)d_packed, N, as_signed ? 0x8000000000000000ull : 0ull, d_beam,
        d_arg);
    BPMF_LAUNCH_CHECK();
    return 0;
}
