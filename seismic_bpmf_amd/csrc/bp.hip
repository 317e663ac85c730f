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
#include "common.h"
#include "../../include/bpmf_hip.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace bpmf {

constexpr int BP_THREADS = 256;
constexpr size_t BP_LDS_MAX = 160 * 1024;

// ------------------------------------------------------------------- prestack ---
// One thread per (station, time sample): reads the C components once, writes P phases.
template <int MAXP>
__global__ __launch_bounds__(256) void bp_prestack_kernel(const float* __restrict__ feat,
                                                          const float* __restrict__ w_ph,
                                                          long long N, int C, int P,
                                                          float* __restrict__ U)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    if (t >= N) return;
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
                                       float* __restrict__ U)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y / P, p = blockIdx.y % P;
    if (t >= N) return;
    float acc = 0.0f;
    for (int c = 0; c < C; ++c)
        acc = __fmaf_rn(w_ph[((size_t)s * C + c) * P + p], feat[((size_t)s * C + c) * (size_t)N + t],
                        acc);
    U[((size_t)s * P + p) * (size_t)N + t] = acc;
}

// ----------------------------------------------------------------------- beam ---
struct BpGroup {  // one LDS residency
    int first_src, n_src, first_win, n_win;
};
struct BpWindow {  // one staged (station, phase) trace window
    int sp, tau0, len, base;  // U row, first moveout, floats staged, LDS float offset
};
struct BpSource {
    int n_act, id, tmin, tmax;  // used stations, global id, extreme used moveouts
};

template <int TPT, int OOB, int REDUCE>
__global__ __launch_bounds__(BP_THREADS) void bp_beam_kernel(
    const float* __restrict__ U, long long N, const BpGroup* __restrict__ groups, int n_groups,
    const BpWindow* __restrict__ wins, const BpSource* __restrict__ srcs,
    const float* __restrict__ term_beta, const int* __restrict__ term_off, int A, int P,
    int default_arg, float* __restrict__ out_beam, int* __restrict__ out_arg)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const long long t0 = (long long)blockIdx.x * (BP_THREADS * TPT);

    float best[TPT];
    int arg[TPT];
#pragma unroll
    for (int j = 0; j < TPT; ++j) { best[j] = 0.0f; arg[j] = default_arg; }

    for (int g = 0; g < n_groups; ++g) {
        const BpGroup grp = groups[g];
        __syncthreads();
        for (int wi = 0; wi < grp.n_win; ++wi) {
            const BpWindow w = wins[grp.first_win + wi];
            const float* src = U + (size_t)w.sp * (size_t)N;
            const long long g0 = t0 + w.tau0;
            for (int x = tid; x < w.len; x += BP_THREADS) {
                const long long gi = g0 + x;
                lds[w.base + x] = (gi >= 0 && gi < N) ? src[gi] : 0.0f;
            }
        }
        __syncthreads();
        for (int ks = 0; ks < grp.n_src; ++ks) {
            const int k = grp.first_src + ks;
            const BpSource sc = srcs[k];
            float acc[TPT];
#pragma unroll
            for (int j = 0; j < TPT; ++j) acc[j] = 0.0f;
            const float* tb = term_beta + (size_t)k * A;
            const int* to = term_off + (size_t)k * A * P;
            for (int ai = 0; ai < sc.n_act; ++ai) {
                const float beta = tb[ai];
                for (int p = 0; p < P; ++p) {
                    const float* lp = lds + to[ai * P + p] + tid;
#pragma unroll
                    for (int j = 0; j < TPT; ++j)
                        acc[j] = __fmaf_rn(beta, lp[j * BP_THREADS], acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < TPT; ++j) {
                const long long t = t0 + tid + j * BP_THREADS;
                bool computed = sc.n_act > 0;
                if (OOB == BPMF_BP_STRICT) computed = computed && (t + sc.tmin >= 0) && (t + sc.tmax < N);
                if (REDUCE == BPMF_BP_REDUCE_MAX) {
                    if (computed && acc[j] > best[j]) { best[j] = acc[j]; arg[j] = sc.id; }
                } else {
                    if (t < N) out_beam[(size_t)k * (size_t)N + t] = computed ? acc[j] : 0.0f;
                }
            }
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
struct bpmf_bp_plan {
    int device = 0;
    size_t K = 0, S = 0, P = 0;
    int tpt = 2;          // time samples per thread -> tile = BP_THREADS * tpt
    int A = 1;            // padded number of used stations per source
    int n_groups = 0;
    size_t lds_bytes = 0; // largest group
    int default_arg = 0;
    BpGroup* d_groups = nullptr;
    BpWindow* d_wins = nullptr;
    BpSource* d_srcs = nullptr;
    float* d_beta = nullptr;
    int* d_off = nullptr;
};

namespace {

struct PlanHost {
    std::vector<BpGroup> groups;
    std::vector<BpWindow> wins;
    std::vector<BpSource> srcs;
    std::vector<float> beta;
    std::vector<int> off;
    size_t lds_floats = 0;
    int A = 1;
};

// Greedy grouping of consecutive sources: a group is closed when adding the next source
// would push the LDS need (sum over used (s,p) rows of tile + moveout spread) past budget.
// Returns false if a single source cannot fit in `hard_floats`.
bool build_plan(const int32_t* mv, const float* ws, size_t K, size_t S, size_t P, int tile,
                size_t soft_floats, size_t hard_floats, int max_group, int32_t id_offset,
                PlanHost& ph)
{
    const size_t SP = S * P;
    int A = 1;
    ph.srcs.resize(K);
    for (size_t k = 0; k < K; ++k) {
        int n = 0;
        long long lo = 0, hi = 0;
        for (size_t s = 0; s < S; ++s) {
            if (ws[k * S + s] == 0.0f) continue;
            for (size_t p = 0; p < P; ++p) {
                long long tau = mv[(k * S + s) * P + p];
                if ((n == 0 && p == 0) || tau < lo) lo = tau;
                if ((n == 0 && p == 0) || tau > hi) hi = tau;
            }
            ++n;
        }
        ph.srcs[k] = BpSource{n, (int)((long long)k + id_offset), (int)lo, (int)hi};
        A = std::max(A, n);
    }
    ph.A = A;
    ph.beta.assign(K * (size_t)A, 0.0f);
    ph.off.assign(K * (size_t)A * P, 0);

    std::vector<int> gmin(SP), gmax(SP);
    std::vector<char> used(SP);
    struct RowUpdate { size_t row; int lo, hi; };
    std::vector<RowUpdate> upd;
    upd.reserve(SP);
    size_t first = 0;
    while (first < K) {
        std::fill(used.begin(), used.end(), 0);
        size_t need = 0, k = first;
        for (; k < K && (int)(k - first) < max_group; ++k) {
            size_t need2 = need;
            // tentative extension of every row the source uses
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
                        need2 += (size_t)((hi - lo) - (gmax[r] - gmin[r]));
                    } else {
                        need2 += (size_t)tile;
                    }
                    upd.push_back(RowUpdate{r, lo, hi});
                }
            }
            const size_t limit = (k == first) ? hard_floats : soft_floats;
            if (need2 > limit) {
                if (k == first) return false;
                break;
            }
            for (const RowUpdate& u : upd) {
                used[u.row] = 1;
                gmin[u.row] = u.lo;
                gmax[u.row] = u.hi;
            }
            need = need2;
        }
        // close group [first, k)
        BpGroup g{(int)first, (int)(k - first), (int)ph.wins.size(), 0};
        std::vector<int> base(SP, -1);
        size_t o = 0;
        for (size_t r = 0; r < SP; ++r) {
            if (!used[r]) continue;
            const int len = tile + (gmax[r] - gmin[r]);
            ph.wins.push_back(BpWindow{(int)r, gmin[r], len, (int)o});
            base[r] = (int)o;
            o += (size_t)len;
            ++g.n_win;
        }
        ph.lds_floats = std::max(ph.lds_floats, o);
        for (size_t kk = first; kk < k; ++kk) {
            int ai = 0;
            for (size_t s = 0; s < S; ++s) {
                if (ws[kk * S + s] == 0.0f) continue;
                ph.beta[kk * A + ai] = ws[kk * S + s];
                for (size_t p = 0; p < P; ++p) {
                    const size_t r = s * P + p;
                    ph.off[(kk * A + ai) * P + p] = base[r] + (mv[(kk * S + s) * P + p] - gmin[r]);
                }
                ++ai;
            }
        }
        ph.groups.push_back(g);
        first = k;
    }
    return true;
}

template <typename Tv>
int upload(const std::vector<Tv>& v, Tv** d)
{
    size_t b = std::max<size_t>(v.size(), 1) * sizeof(Tv);
    BPMF_HIP_CHECK(hipMalloc((void**)d, b));
    if (!v.empty()) BPMF_HIP_CHECK(hipMemcpy(*d, v.data(), v.size() * sizeof(Tv), hipMemcpyHostToDevice));
    return 0;
}

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
    size_t soft_kb = 64;
    if (const char* e = getenv("BPMF_BP_LDS_KB")) soft_kb = (size_t)std::max(8, atoi(e));
    int max_group = 1024;
    if (const char* e = getenv("BPMF_BP_MAX_GROUP")) max_group = std::max(1, atoi(e));
    int tpt_first = 2;
    if (const char* e = getenv("BPMF_BP_TPT")) tpt_first = atoi(e);
    const size_t hard = BP_LDS_MAX / sizeof(float);
    const size_t soft = std::min(hard, soft_kb * 1024 / sizeof(float));

    PlanHost ph;
    int tpt = 0;
    const int candidates[3] = {tpt_first, 2, 1};
    for (int c = 0; c < 3 && !tpt; ++c) {
        const int cand = candidates[c];
        if (cand != 1 && cand != 2 && cand != 4) continue;
        ph = PlanHost();
        if (build_plan(moveouts, w_sources, K, S, P, BP_THREADS * cand, soft, hard, max_group,
                       source_id_offset, ph))
            tpt = cand;
    }
    if (!tpt) {
        set_error("bpmf_bp_plan_create: one source's %zu station-phase windows do not fit in LDS",
                  S * P);
        return -1;
    }
    BPMF_HIP_CHECK(hipSetDevice(device));
    bpmf_bp_plan* pl = new bpmf_bp_plan();
    pl->device = device;
    pl->K = K; pl->S = S; pl->P = P;
    pl->tpt = tpt;
    pl->A = ph.A;
    pl->n_groups = (int)ph.groups.size();
    pl->lds_bytes = ph.lds_floats * sizeof(float);
    pl->default_arg = source_id_offset;
    int rc = 0;
    if ((rc = upload(ph.groups, &pl->d_groups)) || (rc = upload(ph.wins, &pl->d_wins)) ||
        (rc = upload(ph.srcs, &pl->d_srcs)) || (rc = upload(ph.beta, &pl->d_beta)) ||
        (rc = upload(ph.off, &pl->d_off))) {
        bpmf_bp_plan_destroy(pl);
        return rc;
    }
    *plan_out = pl;
    return 0;
}

extern "C" void bpmf_bp_plan_destroy(bpmf_bp_plan* pl)
{
    if (!pl) return;
    (void)hipFree(pl->d_groups);
    (void)hipFree(pl->d_wins);
    (void)hipFree(pl->d_srcs);
    (void)hipFree(pl->d_beta);
    (void)hipFree(pl->d_off);
    delete pl;
}

extern "C" size_t bpmf_bp_workspace_bytes(const bpmf_bp_plan* pl, size_t N, size_t C)
{
    (void)C;
    if (!pl) return 0;
    return align_up(pl->S * pl->P * N * sizeof(float), 256);
}

namespace {

template <int TPT, int OOB, int REDUCE>
int launch_beam(const bpmf_bp_plan* pl, const float* U, size_t N, hipStream_t stream, float* beam,
                int32_t* arg)
{
    auto kern = bp_beam_kernel<TPT, OOB, REDUCE>;
    if (pl->lds_bytes > 64 * 1024)
        BPMF_HIP_CHECK(hipFuncSetAttribute((const void*)kern,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)BP_LDS_MAX));
    const size_t tile = (size_t)BP_THREADS * TPT;
    dim3 grid((unsigned)((N + tile - 1) / tile));
    profile_mark(BPMF_KERNEL_BP_BEAM, 0, stream);
    kern<<<grid, dim3(BP_THREADS), pl->lds_bytes, stream>>>(
        U, (long long)N, pl->d_groups, pl->n_groups, pl->d_wins, pl->d_srcs, pl->d_beta, pl->d_off,
        pl->A, (int)pl->P, pl->default_arg, beam, arg);
    BPMF_LAUNCH_CHECK();
    profile_mark(BPMF_KERNEL_BP_BEAM, 1, stream);
    return 0;
}

template <int TPT>
int dispatch_beam(const bpmf_bp_plan* pl, const float* U, size_t N, int oob, int reduce,
                  hipStream_t stream, float* beam, int32_t* arg)
{
    if (oob == BPMF_BP_STRICT && reduce == BPMF_BP_REDUCE_MAX)
        return launch_beam<TPT, BPMF_BP_STRICT, BPMF_BP_REDUCE_MAX>(pl, U, N, stream, beam, arg);
    if (oob == BPMF_BP_FLEXIBLE && reduce == BPMF_BP_REDUCE_MAX)
        return launch_beam<TPT, BPMF_BP_FLEXIBLE, BPMF_BP_REDUCE_MAX>(pl, U, N, stream, beam, arg);
    if (oob == BPMF_BP_STRICT)
        return launch_beam<TPT, BPMF_BP_STRICT, BPMF_BP_REDUCE_NONE>(pl, U, N, stream, beam, arg);
    return launch_beam<TPT, BPMF_BP_FLEXIBLE, BPMF_BP_REDUCE_NONE>(pl, U, N, stream, beam, arg);
}

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
    if (workspace_bytes < bpmf_bp_workspace_bytes(pl, N, C)) {
        set_error("bpmf_bp_run_dev: workspace too small");
        return -1;
    }
    float* U = (float*)d_workspace;
    const int P = (int)pl->P, S = (int)pl->S;
    if (P <= 4) {
        dim3 grid((unsigned)((N + 255) / 256), (unsigned)S);
        bp_prestack_kernel<4><<<grid, dim3(256), 0, stream>>>(d_features, d_w_phases, (long long)N,
                                                              (int)C, P, U);
    } else {
        dim3 grid((unsigned)((N + 255) / 256), (unsigned)(S * P));
        bp_prestack_any_kernel<<<grid, dim3(256), 0, stream>>>(d_features, d_w_phases, (long long)N,
                                                               (int)C, P, U);
    }
    BPMF_LAUNCH_CHECK();
    switch (pl->tpt) {
        case 1: return dispatch_beam<1>(pl, U, N, out_of_bounds, reduce, stream, d_beam_out, d_arg_out);
        case 2: return dispatch_beam<2>(pl, U, N, out_of_bounds, reduce, stream, d_beam_out, d_arg_out);
        default: return dispatch_beam<4>(pl, U, N, out_of_bounds, reduce, stream, d_beam_out, d_arg_out);
    }
}

extern "C" int bpmf_bp_run(const float* features, const int32_t* moveouts, const float* w_phases,
                           const float* w_sources, size_t N, size_t K, size_t S, size_t C, size_t P,
                           int out_of_bounds, int reduce, int device, float* beam_out,
                           int32_t* arg_out)
{
    if (!features || !moveouts || !w_phases || !w_sources || !beam_out) {
        set_error("bpmf_bp_run: null pointer");
        return -1;
    }
    bpmf_bp_plan* pl = nullptr;
    if (int rc = bpmf_bp_plan_create(moveouts, w_sources, K, S, P, device, 0, &pl)) return rc;
    const size_t b_f = S * C * N * sizeof(float), b_wp = S * C * P * sizeof(float),
                 b_ws = bpmf_bp_workspace_bytes(pl, N, C),
                 b_beam = (reduce == BPMF_BP_REDUCE_MAX ? N : K * N) * sizeof(float),
                 b_arg = N * sizeof(int32_t);
    const size_t o_f = 0, o_wp = o_f + align_up(b_f, 256), o_ws = o_wp + align_up(b_wp, 256),
                 o_beam = o_ws + align_up(b_ws, 256), o_arg = o_beam + align_up(b_beam, 256),
                 total = o_arg + b_arg;
    char* base = nullptr;
    hipError_t e = hipMalloc((void**)&base, total);
    if (e != hipSuccess) {
        set_error("bpmf_bp_run: hipMalloc(%zu) failed: %s", total, hipGetErrorString(e));
        bpmf_bp_plan_destroy(pl);
        return -2;
    }
    int rc = 0;
    hipStream_t stream = nullptr;
    auto fail = [&](hipError_t err, const char* what) {
        set_error("bpmf_bp_run: %s failed: %s", what, hipGetErrorString(err));
        rc = -2;
    };
    if ((e = hipMemcpyAsync(base + o_f, features, b_f, hipMemcpyHostToDevice, stream)) != hipSuccess) fail(e, "H2D features");
    if (!rc && (e = hipMemcpyAsync(base + o_wp, w_phases, b_wp, hipMemcpyHostToDevice, stream)) != hipSuccess) fail(e, "H2D weights_phases");
    if (!rc && reduce == BPMF_BP_REDUCE_NONE &&
        (e = hipMemsetAsync(base + o_beam, 0, b_beam, stream)) != hipSuccess) fail(e, "memset");
    if (!rc)
        rc = bpmf_bp_run_dev(pl, (const float*)(base + o_f), (const float*)(base + o_wp), N, C,
                             out_of_bounds, reduce, base + o_ws, b_ws, stream,
                             (float*)(base + o_beam), (int32_t*)(base + o_arg));
    if (!rc && (e = hipMemcpyAsync(beam_out, base + o_beam, b_beam, hipMemcpyDeviceToHost, stream)) != hipSuccess) fail(e, "D2H beam");
    if (!rc && reduce == BPMF_BP_REDUCE_MAX && arg_out &&
        (e = hipMemcpyAsync(arg_out, base + o_arg, b_arg, hipMemcpyDeviceToHost, stream)) != hipSuccess) fail(e, "D2H argmax");
    if (!rc && (e = hipStreamSynchronize(stream)) != hipSuccess) fail(e, "synchronize");
    (void)hipFree(base);
    bpmf_bp_plan_destroy(pl);
    return rc;
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
