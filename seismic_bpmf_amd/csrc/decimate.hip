// Grid decimation by moveout similarity on the device: the step
// that feeds the backprojection grid, BPMF.clib.find_similar_sources (BPMF/clib.py:104-221) ->
// BPMF/libc.c:find_similar_moveouts (:55-223, "smallest") / find_similar_moveouts2 (:225-387,
// "closest").  The reference is O(K^2 S) with a sequential greedy dependency ("can take ~24 hours
// or more for large grids", tutorial notebook 4 cell 34).
//
// Exactness: source n2 is redundant iff some LOWER-indexed source that is itself kept is closer
// than the threshold.  The greedy order is preserved by processing kept candidates in batches of
// 256 ascending indices: (1) pair distances inside the batch, (2) one thread resolves the batch
// sequentially (who survives), (3) every later source tests itself against the batch's survivors
// in parallel.  Distances follow the reference's arithmetic operation for operation (float
// subtraction, square in double, float accumulator), so the result equals the single-threaded
// reference bit for bit (tests/golden/similar_sources.npz).
#include "common.h"
#include "context.h"
#include "../../include/bpmf_hip.h"

#include <algorithm>
#include <vector>

namespace bpmf {

constexpr int FS_BATCH = 256;
constexpr int FS_MAX_STATIONS = 256;

// argsort of every source's moveouts (selection sort on indexes: first minimum wins ties,
// BPMF/libc.c:389-410); order[k*S + r] = station of rank r.
__global__ void fs_argsort_kernel(const float* __restrict__ mv, size_t K, int S, int* __restrict__ order)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float* m = mv + k * S;
    int* o = order + k * S;
    for (int i = 0; i < S; ++i) o[i] = i;
    for (int i = 0; i + 1 < S; ++i) {
        int mn = i;
        for (int j = i + 1; j < S; ++j)
            if (m[o[j]] < m[o[mn]]) mn = j;
        int tmp = o[mn]; o[mn] = o[i]; o[i] = tmp;
    }
}

// summed squared moveout difference between sources a and b over n_diff stations.
//   MODE 1 "closest":  the n_diff stations closest to a (order_a), float accumulator fed with
//                      double squares                                      libc.c:319-325
//   MODE 0 "smallest": the n_diff smallest squared differences (each rounded to float), added in
//                      ascending order                                     libc.c:150-165
template <int MODE>
__device__ float fs_dt2(const float* __restrict__ ma, const float* __restrict__ mb,
                        const int* __restrict__ order_a, int S, int n_diff, float* work)
{
    float dt2 = 0.0f;
    if (MODE == 1) {
        for (int s = 0; s < n_diff; ++s) {
            const int st = order_a[s];
            const double d = (double)(ma[st] - mb[st]);
            dt2 = (float)((double)dt2 + d * d);
        }
    } else {
        for (int s = 0; s < S; ++s) {
            const double d = (double)(ma[s] - mb[s]);
            work[s] = (float)(d * d);
        }
        // partial selection sort: only the n_diff smallest are needed, in ascending order
        for (int i = 0; i < n_diff; ++i) {
            int mn = i;
            for (int j = i + 1; j < S; ++j)
                if (work[j] < work[mn]) mn = j;
            const float tmp = work[mn]; work[mn] = work[i]; work[i] = tmp;
            dt2 += work[i];
        }
    }
    return dt2;
}

struct FsState {
    int cursor;    // next position in the subset to consider
    int n_batch;   // members of the current batch
    int last_pos;  // subset position of the batch's last member
};

// (1) collect the next <= 256 not-yet-redundant members of the subset, ascending.
__global__ void fs_next_batch_kernel(const int* __restrict__ sub, int m, const int* __restrict__ red,
                                     FsState* st, int* __restrict__ batch)
{
    int p = st->cursor, n = 0, last = p - 1;
    while (p < m && n < FS_BATCH) {
        const int k = sub ? sub[p] : p;
        if (!red[k]) { batch[n++] = k; last = p; }
        ++p;
    }
    st->cursor = p;
    st->n_batch = n;
    st->last_pos = last;
}

// (2) "closer than threshold" for every ordered pair (a < b) of the batch.
template <int MODE>
__global__ void fs_intra_kernel(const float* __restrict__ mv, const int* __restrict__ order, int S,
                                int n_diff, float thr2, const FsState* st,
                                const int* __restrict__ batch, unsigned char* __restrict__ close,
                                float* __restrict__ scratch)
{
    const int a = blockIdx.x, b = threadIdx.x;
    const int n = st->n_batch;
    if (a >= n || b >= n || b <= a) return;
    const int ka = batch[a], kb = batch[b];
    float* work = scratch + ((size_t)blockIdx.x * FS_BATCH + threadIdx.x) * (MODE == 0 ? S : 0);
    const float d = fs_dt2<MODE>(mv + (size_t)ka * S, mv + (size_t)kb * S, order + (size_t)ka * S, S,
                                 n_diff, work);
    close[a * FS_BATCH + b] = d < thr2;
}

// (3) sequential resolution inside the batch (one thread): a kept member knocks out every later
//     member it is close to.  Survivors keep red == 0.
__global__ void fs_resolve_kernel(const FsState* st, const int* __restrict__ batch,
                                  const unsigned char* __restrict__ close, int* __restrict__ red,
                                  unsigned char* __restrict__ alive)
{
    const int n = st->n_batch;
    for (int a = 0; a < n; ++a) alive[a] = 1;
    for (int a = 0; a < n; ++a) {
        if (!alive[a]) continue;
        for (int b = a + 1; b < n; ++b)
            if (alive[b] && close[a * FS_BATCH + b]) { alive[b] = 0; red[batch[b]] = 1; }
    }
}

// (4) every later, still-kept member of the subset tests itself against the survivors.
template <int MODE>
__global__ void fs_apply_kernel(const float* __restrict__ mv, const int* __restrict__ order, int S,
                                int n_diff, float thr2, const int* __restrict__ sub, int m,
                                const FsState* st, const int* __restrict__ batch,
                                const unsigned char* __restrict__ alive, int* __restrict__ red,
                                float* __restrict__ scratch)
{
    const int n = st->n_batch;
    const long long p = (long long)st->last_pos + 1 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0 || p >= m) return;
    const int k2 = sub ? sub[p] : (int)p;
    if (red[k2]) return;
    float* work = scratch + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * (MODE == 0 ? S : 0);
    const float* m2 = mv + (size_t)k2 * S;
    for (int a = 0; a < n; ++a) {
        if (!alive[a]) continue;
        const int ka = batch[a];
        if (fs_dt2<MODE>(mv + (size_t)ka * S, m2, order + (size_t)ka * S, S, n_diff, work) < thr2) {
            red[k2] = 1;
            return;
        }
    }
}

template <int MODE>
static int fs_greedy(const float* d_mv, const int* d_order, int S, int n_diff, float thr2,
                     const int* d_sub, int m, FsState* d_st, int* d_batch, unsigned char* d_close,
                     unsigned char* d_alive, float* d_scratch, int* d_red, hipStream_t stream)
{
    if (m < 2) return 0;
    BPMF_HIP_CHECK(hipMemsetAsync(d_st, 0, sizeof(FsState), stream));
    const int n_batches = (m + FS_BATCH - 1) / FS_BATCH;  // upper bound; spent batches are no-ops
    const unsigned apply_blocks = (unsigned)((m + 255) / 256);
    for (int it = 0; it < n_batches; ++it) {
        fs_next_batch_kernel<<<1, 1, 0, stream>>>(d_sub, m, d_red, d_st, d_batch);
        fs_intra_kernel<MODE><<<FS_BATCH, FS_BATCH, 0, stream>>>(d_mv, d_order, S, n_diff, thr2, d_st,
                                                                d_batch, d_close, d_scratch);
        fs_resolve_kernel<<<1, 1, 0, stream>>>(d_st, d_batch, d_close, d_red, d_alive);
        fs_apply_kernel<MODE><<<apply_blocks, 256, 0, stream>>>(d_mv, d_order, S, n_diff, thr2, d_sub, m,
                                                                d_st, d_batch, d_alive, d_red, d_scratch);
    }
    BPMF_LAUNCH_CHECK();
    return 0;
}

}  // namespace bpmf

using namespace bpmf;

// Host-driven: cell membership is computed on the host (two comparisons per source and cell,
// BPMF/libc.c:123-131), the O(K^2 S) part runs on the device.
extern "C" int bpmf_find_similar_sources(const float* moveouts, const float* source_longitude,
                                         const float* source_latitude, const float* cell_longitude,
                                         const float* cell_latitude, float threshold,
                                         size_t n_sources, size_t n_stations, size_t n_cells_longitude,
                                         size_t n_cells_latitude, size_t n_stations_for_diff,
                                         int method, int device, int32_t* redundant_sources)
{
    const size_t K = n_sources, S = n_stations;
    if (!moveouts || !source_longitude || !source_latitude || !cell_longitude || !cell_latitude ||
        !redundant_sources || K == 0 || S == 0 || S > FS_MAX_STATIONS || n_stations_for_diff == 0 ||
        n_stations_for_diff > S || K > 0x7fffffffull || (method != 0 && method != 1)) {
        set_error("bpmf_find_similar_sources: bad argument (K=%zu S=%zu n_diff=%zu method=%d)", K, S,
                  n_stations_for_diff, method);
        return -1;
    }
    BPMF_BIND_DEVICE(device);
    // threshold^2 * n_diff exactly as the reference: (float)n * pow(threshold, 2) -> float
    const float thr2 = (float)((double)(float)n_stations_for_diff * ((double)threshold * (double)threshold));
    // the device's private stream and working set (context.h): nothing allocated or freed per call,
    // nothing on the null stream; calls on one device take turns
    DeviceContext* ctx = device_context(device);
    if (!ctx) return -2;
    std::lock_guard<std::mutex> call_lock(ctx->call_mutex);
    hipStream_t stream = ctx->s_run;
    const size_t scratch_floats = method == 0 ? std::max<size_t>((size_t)FS_BATCH * FS_BATCH, (K + 255) / 256 * 256) * S : 1;
    size_t total = 0;
    auto carve = [&](size_t bytes) { const size_t o = total; total += align_up(std::max<size_t>(bytes, 1), 256); return o; };
    const size_t o_mv = carve(K * S * sizeof(float)), o_order = carve(K * S * sizeof(int)),
                 o_red = carve(K * sizeof(int)), o_sub = carve(K * sizeof(int)), o_st = carve(sizeof(FsState)),
                 o_batch = carve(FS_BATCH * sizeof(int)), o_close = carve((size_t)FS_BATCH * FS_BATCH),
                 o_alive = carve(FS_BATCH), o_scratch = carve(scratch_floats * sizeof(float));
    char* base = ctx->reserve_device(total);
    if (!base) return -2;
    float* d_mv = (float*)(base + o_mv); int* d_order = (int*)(base + o_order); int* d_red = (int*)(base + o_red);
    int* d_sub = (int*)(base + o_sub); FsState* d_st = (FsState*)(base + o_st); int* d_batch = (int*)(base + o_batch);
    unsigned char* d_close = (unsigned char*)(base + o_close); unsigned char* d_alive = (unsigned char*)(base + o_alive);
    float* d_scratch = (float*)(base + o_scratch);
    int rc = 0;
    auto run = [&]() -> int {
        BPMF_HIP_CHECK(hipMemcpyAsync(d_mv, moveouts, K * S * sizeof(float), hipMemcpyHostToDevice, stream));
        BPMF_HIP_CHECK(hipMemsetAsync(d_red, 0, K * sizeof(int), stream));
        BPMF_HIP_CHECK(hipMemsetAsync(d_close, 0, FS_BATCH * FS_BATCH, stream));
        if (method == 1) {
            fs_argsort_kernel<<<dim3((unsigned)((K + 63) / 64)), dim3(64), 0, stream>>>(d_mv, K, (int)S, d_order);
            BPMF_LAUNCH_CHECK();
        }
        // first pass: pairs inside the same (longitude, latitude) cell, cell by cell
        std::vector<int> sub;
        for (size_t i = 0; i < n_cells_longitude; ++i)
            for (size_t j = 0; j < n_cells_latitude; ++j) {
                sub.clear();
                for (size_t k = 0; k < K; ++k)
                    if (!(source_longitude[k] < cell_longitude[i] || source_longitude[k] >= cell_longitude[i + 1] ||
                          source_latitude[k] < cell_latitude[j] || source_latitude[k] >= cell_latitude[j + 1]))
                        sub.push_back((int)k);
                if (sub.size() < 2) continue;
                BPMF_HIP_CHECK(hipMemcpyAsync(d_sub, sub.data(), sub.size() * sizeof(int), hipMemcpyHostToDevice, stream));
                BPMF_HIP_CHECK(hipStreamSynchronize(stream));  // `sub` is reused by the next cell
                int r = method == 1
                    ? fs_greedy<1>(d_mv, d_order, (int)S, (int)n_stations_for_diff, thr2, d_sub, (int)sub.size(), d_st, d_batch, d_close, d_alive, d_scratch, d_red, stream)
                    : fs_greedy<0>(d_mv, d_order, (int)S, (int)n_stations_for_diff, thr2, d_sub, (int)sub.size(), d_st, d_batch, d_close, d_alive, d_scratch, d_red, stream);
                if (r) return r;
            }
        // second pass: all remaining pairs
        int r = method == 1
            ? fs_greedy<1>(d_mv, d_order, (int)S, (int)n_stations_for_diff, thr2, nullptr, (int)K, d_st, d_batch, d_close, d_alive, d_scratch, d_red, stream)
            : fs_greedy<0>(d_mv, d_order, (int)S, (int)n_stations_for_diff, thr2, nullptr, (int)K, d_st, d_batch, d_close, d_alive, d_scratch, d_red, stream);
        if (r) return r;
        BPMF_HIP_CHECK(hipMemcpyAsync(redundant_sources, d_red, K * sizeof(int), hipMemcpyDeviceToHost, stream));
        BPMF_HIP_CHECK(hipStreamSynchronize(stream));
        return 0;
    };
    rc = run();
    (void)hipStreamSynchronize(stream);   // (also after a failure: the working set goes back to its cache)
    return rc;
}
