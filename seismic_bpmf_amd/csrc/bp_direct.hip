// Beamforming WITHOUT the LDS windows: the safety net behind the planned kernels of bp.hip / bp_fast.hip.
//
// The planned kernels keep, per group of sources, one window of every used (station, phase) row in the
// 160 KB of LDS; a source whose own windows do not fit at the smallest tile (more than ~160 weighted
// rows, i.e. > 80 stations x 2 phases), or a grid with more than 256 (station, phase) terms per
// source, has no plan.  The reference (beampower.beamform, called at BPMF/template_search.py:549-569)
// has no such limit, so those grids run here instead of failing: every thread owns 4 time samples and
// walks the sources' COMPACT term lists {row, moveout, weight} (wave-uniform: scalar loads) in the
// oracle's order -- stations ascending, phases inside, zero-weight stations skipped
// (oracle/bpmf_oracle.c:bp_cpu) -- gathering U[row][t + tau] from global memory (the rows of a tile
// stay in L2).  Same fmaf chain, same strict / flexible rule, same running maximum as everywhere
// else: bit-identical results, at L2-gather speed (4 bytes per term and sample through the vector
// memory path instead of the LDS) -- a correctness path, not a tuned one.
#include "common.h"
#include "bp_plan.h"
#include "../../include/bpmf_hip.h"

namespace bpmf {

constexpr int BPD_THREADS = 256, BPD_TPT = 4, BPD_TILE = BPD_THREADS * BPD_TPT;

// gridDim.y: reduce="max": ranges of sources (partial rows `split_stride` apart, folded by
// bp_merge_splits_kernel); reduce="none": one source per blockIdx.y (+ 65535 z)
template <int OOB, int REDUCE>
__global__ __launch_bounds__(BPD_THREADS) void bp_beam_direct_kernel(
    const float* __restrict__ U, long long N, const int4* __restrict__ hdr,
    const long long* __restrict__ first_term, const int4* __restrict__ terms, int K, int id_offset,
    float* __restrict__ out_beam, int* __restrict__ out_arg, long long split_stride, float best0)
{
    const int tid = threadIdx.x;
    const long long t0 = (long long)blockIdx.x * BPD_TILE + tid;
    int k_lo, k_hi;
    if (REDUCE == BPMF_BP_REDUCE_MAX) {
        const int per = (K + (int)gridDim.y - 1) / (int)gridDim.y;
        k_lo = (int)blockIdx.y * per;
        k_hi = min(K, k_lo + per);
    } else {
        k_lo = (int)(blockIdx.y + (size_t)blockIdx.z * gridDim.y);
        k_hi = min(K, k_lo + 1);
    }
    float best[BPD_TPT];
    int arg[BPD_TPT];
#pragma unroll
    for (int j = 0; j < BPD_TPT; ++j) { best[j] = best0; arg[j] = id_offset; }

    for (int k = k_lo; k < k_hi; ++k) {
        const int4 h = hdr[k];                       // {active, tmin, tmax, -}
        const long long i0 = first_term[k], i1 = first_term[k + 1];
        float acc[BPD_TPT];
        bool computed[BPD_TPT];
#pragma unroll
        for (int j = 0; j < BPD_TPT; ++j) {
            acc[j] = 0.0f;
            const long long t = t0 + (long long)j * BPD_THREADS;
            computed[j] = h.x != 0;
            if (OOB == BPMF_BP_STRICT) computed[j] = computed[j] && t + h.y >= 0 && t + h.z < N;
        }
#pragma unroll 2
        for (long long i = i0; i < i1; ++i) {
            const int4 tm = terms[i];                // {row, tau, weight bits, -}
            const float beta = __int_as_float(tm.z);
            const float* __restrict__ row = U + (size_t)tm.x * (size_t)N;
            float v[BPD_TPT];
            bool in[BPD_TPT];
#pragma unroll
            for (int j = 0; j < BPD_TPT; ++j) {
                const long long x = t0 + (long long)j * BPD_THREADS + tm.y;
                in[j] = x >= 0 && x < N;
                v[j] = row[x < 0 ? 0 : (x >= N ? N - 1 : x)];
            }
            // (a term outside the trace contributes nothing: the chain skips it, it does not add a zero)
#pragma unroll
            for (int j = 0; j < BPD_TPT; ++j) acc[j] = in[j] ? __fmaf_rn(beta, v[j], acc[j]) : acc[j];
        }
#pragma unroll
        for (int j = 0; j < BPD_TPT; ++j) {
            const long long t = t0 + (long long)j * BPD_THREADS;
            if (REDUCE == BPMF_BP_REDUCE_MAX) {
                // sources ascending: a strict > keeps the lowest id on equal beams
                if (computed[j] && acc[j] > best[j]) { best[j] = acc[j]; arg[j] = id_offset + k; }
            } else if (t < N) {
                out_beam[(size_t)k * (size_t)N + t] = computed[j] ? acc[j] : 0.0f;
            }
        }
    }
    if (REDUCE == BPMF_BP_REDUCE_MAX) {
#pragma unroll
        for (int j = 0; j < BPD_TPT; ++j) {
            const long long t = t0 + (long long)j * BPD_THREADS;
            if (t < N) {
                out_beam[(size_t)blockIdx.y * split_stride + t] = best[j];
                out_arg[(size_t)blockIdx.y * split_stride + t] = arg[j];
            }
        }
    }
}

// source ranges per tile of reduce="max": enough workgroups for ~4 rounds over the chip
int direct_split_count(const bpmf_bp_plan* pl, size_t N)
{
    const long long tiles = (long long)((N + BPD_TILE - 1) / BPD_TILE);
    long long want = tiles >= 1024 ? 1 : (1024 + tiles - 1) / tiles;
    return (int)std::max<long long>(1, std::min<long long>({want, (long long)pl->K, 256}));
}

int launch_beam_direct(const bpmf_bp_plan* pl, const float* U, size_t N, int oob, int reduce,
                       hipStream_t stream, float* beam, int32_t* arg, int n_split, long long split_stride,
                       float best0)
{
    const unsigned tiles = (unsigned)((N + BPD_TILE - 1) / BPD_TILE);
    dim3 grid(tiles, 1, 1);
    if (reduce == BPMF_BP_REDUCE_MAX) {
        grid.y = (unsigned)n_split;
    } else {
        grid.y = (unsigned)std::min<size_t>(pl->K, 65535);
        grid.z = (unsigned)((pl->K + grid.y - 1) / grid.y);
    }
#define BPD_LAUNCH(OOB, RED)                                                                       \
    bp_beam_direct_kernel<OOB, RED><<<grid, dim3(BPD_THREADS), 0, stream>>>(                       \
        U, (long long)N, pl->d_dhdr, pl->d_dfirst, pl->d_dterms, (int)pl->K, pl->id_offset, beam, arg, \
        split_stride, best0)
    if (oob == BPMF_BP_STRICT) {
        if (reduce == BPMF_BP_REDUCE_MAX) BPD_LAUNCH(BPMF_BP_STRICT, BPMF_BP_REDUCE_MAX);
        else BPD_LAUNCH(BPMF_BP_STRICT, BPMF_BP_REDUCE_NONE);
    } else {
        if (reduce == BPMF_BP_REDUCE_MAX) BPD_LAUNCH(BPMF_BP_FLEXIBLE, BPMF_BP_REDUCE_MAX);
        else BPD_LAUNCH(BPMF_BP_FLEXIBLE, BPMF_BP_REDUCE_NONE);
    }
#undef BPD_LAUNCH
    BPMF_LAUNCH_CHECK();
    return 0;
}

}  // namespace bpmf
