/*
 * bpmf_hip.h -- C ABI of libbpmf_hip.so, the MI355X (gfx950) implementation of the two
 * data-parallel hot paths of the Seismic_BPMF workflow.
 *
 * Conventions (mirroring the reference's own ctypes layer, BPMF/clib.py:14-84 and
 * BPMF/libc.h:1-11): plain C symbols, C-contiguous float32 / int32 arrays, sizes as
 * size_t, the caller allocates every output.  Unlike the reference (all functions
 * return void and report problems with printf) every entry point returns an int status:
 * 0 = ok, -1 = bad argument, -2 = HIP runtime error; bpmf_last_error() gives the text.
 *
 * Two flavours per path:
 *   *_run      host pointers in / host pointers out (what a ctypes wrapper of the
 *              reference's third-party back-ends binds; does H2D, kernels, D2H);
 *   *_run_dev  device pointers, asynchronous on the given HIP stream, caller-provided
 *              workspace (what the resident / multi-GPU / benchmark path uses).
 *
 * The hot-path arithmetic itself is NOT in the reference tree: BPMF calls the external
 * packages fast_matched_filter and beampower (pyproject.toml:28-29).  Each entry point
 * cites the reference call site it serves.
 */
#ifndef BPMF_HIP_H
#define BPMF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *bpmf_stream_t; /* a hipStream_t; NULL = the default stream */

/* ---------------------------------------------------------------- diagnostics --- */

/* Text of the last error raised on the calling thread ("" if none). */
const char *bpmf_last_error(void);

/* Number of visible HIP devices (<0 on runtime error).
 * Replaces the device query of BPMF/GPU.cu:8-24 (caract_GPU). */
int bpmf_device_count(void);

/* Name, memory and compute-unit count of one device.  BPMF/GPU.cu:14-22. */
int bpmf_device_info(int device, char *name, size_t name_len, size_t *total_mem_bytes,
                     int *compute_units);

/* Optional timing of the dominant kernel of each path with HIP events recorded on the
 * caller's stream (used by bench.py for the roofline figure; off by default).
 * Each launch of the kernel logs its own event pair; read them after synchronising.
 * (The reference only has wall-clock prints: BPMF/similarity_search.py:789-806.) */
#define BPMF_KERNEL_MF_MAIN 0
#define BPMF_KERNEL_BP_BEAM 1
#define BPMF_KERNEL_COUNT 2
void bpmf_profile_enable(int enable); /* enabling clears the log; disabling keeps it */
int bpmf_profile_count(int which_kernel); /* launches logged since enable */
int bpmf_profile_get_ms(int which_kernel, int launch_index, float *milliseconds);
/* The device that launch ran on (-1: no such launch).  Launches made by several host threads (the
 * *_multi entry points: one thread per GPU) are logged per thread and per device: a start edge and
 * a stop edge pair up only on the thread that recorded both. */
int bpmf_profile_get_device(int which_kernel, int launch_index);

/* The host-pointer entry points (bpmf_mf_run, bpmf_bp_run, their *_multi forms,
 * bpmf_find_similar_sources) keep, per device and process: two private streams, a side stream, four
 * events, two pinned staging pieces and the device working set of the largest recent call -- created
 * once, reused by every call, never destroyed while the library is loaded; calls on one device take
 * turns (the reference's counterpart: the third-party back-ends allocate and free per call,
 * BPMF/similarity_search.py:526-533, BPMF/template_search.py:549-558).  bpmf_release_device_memory
 * gives the working set and the pinned pieces back (device < 0: every device; waits for a running
 * call); bpmf_device_memory_held reports what is held right now; option host.cache_limit_mb (0 = no limit) gives a
 * working set above the limit back at the end of the call that needed it.  The *_dev entry points hold nothing. */
int bpmf_release_device_memory(int device);
int bpmf_device_memory_held(int device, size_t *device_bytes, size_t *pinned_bytes);
/* Where the time of the calling thread's LAST host-pointer call went (bpmf_bp_run / bpmf_mf_run, or the first
 * device's block of a *_run_multi call made from this thread), milliseconds on the host clock:
 *   out[0] the whole call; out[1] from entry to the first kernel being enqueued (plan lookup, working set, the
 *   first piece of the day through the pinned buffers); out[2] host threads copying the day into the pinned
 *   pieces (summed over the pieces; overlaps the kernels of earlier pieces); out[3] waiting for the device after
 *   the last launch was enqueued (kernels still running + the results' way back); out[4] pieces the day arrived
 *   in; out[5] host threads that filled them; out[6] waiting for a pinned piece to be free again (its previous
 *   copy still in flight); out[7] inside the runtime's asynchronous-copy calls of the pieces;
 *   out[8] finding (or building) the backprojection plan; out[9] reserving the device working set and the pinned pieces.  Returns the number of values written (<= n).  The reference has no
 *   counterpart (its back-ends' calls are opaque, BPMF/template_search.py:549-558); bench.py reports these beside
 *   the end-to-end times. */
int bpmf_host_call_stats(double *out, int n);

/* Execution options.  The library reads NOTHING from the environment.  An option selects among
 * code paths and sizes that produce identical results (kernel family, LDS budget, batch sizes of
 * the host-pointer calls, group ranges per tile): the GPU tests use them to drive every kernel
 * family through the same parity cases, tools/ to measure alternatives.  Process-wide, thread
 * safe; plans already built keep the variant they were built with.  Names:
 *   bp.lds_kb bp.max_group bp.tpt bp.reorder bp.dual bp.packed bp.wps bp.uvgpr bp.fast
 *   bp.fast_uniform bp.fast_tile bp.halves bp.direct bp.split bp.wpb bp.smeta bp.slot_prio bp.verbose
 *   mf.wave_kernel mf.tiles_per_wave mf.boundary_prio mf.fused_prologue mf.channel_split mf.max_mfma_step mf.host_batch_kb mf.host_piece_kb mf.verbose
 *   stats.bucketed_median (MAD threshold, window medians: 2 one pass each, 1 two passes, 0 radix select only)
 *   stats.row_grid_min_n (row median / MAD: rows at least this long are read twice by the whole chip; -1 never)
 *   stats.kurt_full_chunks (row kurtosis: full 8192-sample chunks summed by one workgroup each through LDS)
 *   mf.host_piece_lags bp.host_piece_samples (host-pointer calls: the day arrives in pieces while the first kernels run; 0 = off)
 *   debug.poison_output (tests: outputs pre-filled with 0xFF bytes, so that a sample no kernel writes shows)
 *   debug.virtual_devices (tests: k > 0 makes every `device` argument a LOGICAL device 0 .. k-1, logical d on
 *     physical GPU d % visible, each with its own streams / working set / call mutex and, in the *_run_multi
 *     entry points, its own host thread: the multi-device branches run on a box with one GPU;
 *     bpmf_device_count then reports k)
 *   multi.peer_fanout (*_run_multi on several devices: 1 = the first device uploads the day of data from the
 *     host ONCE and the others copy it device -> device, hipMemcpyPeerAsync over xGMI; 0 = every device
 *     uploads from the host itself, which is what the upstream back-ends do)
 *     a device-to-device copy the platform refuses falls back to host uploads for that device, status 0, with a
 *     "note: ..." in bpmf_last_error; debug.fail_peer_copy (tests) makes every copy be refused
 *   mf.split16 (off by default; CHANGES RESULTS within a stated tolerance: matched-filter numerators on the fp16
 *     matrix pipe from hi/lo splits of data and templates, three products, fp32 accumulation -- csrc/mf_split.h;
 *     |d cc_sum| <= 3e-7 * sum|w| measured, 2e-5 allowed by the north star; x 2.3-2.4 at configs[1]; templates of up
 *     to 2049 samples (in segments of at most 376); composes with the mf.compat_* switches; 1 leaves launches of fewer
 *     than 128 (template, 8192-lag block) pairs to the exact kernel (latency-bound there), 2 takes every launch; enlarges
 *     bpmf_mf_workspace_bytes and what a prepared day holds: set it before the day's first call)
 *   and the result-changing upstream-compatibility switches (off by default; every one has
 *   a variant of the CPU oracle and tools/diff_upstream.py tells which combination equals the real packages):
 *   mf.compat_exclusive_last_lag mf.compat_sqrt_norm mf.compat_range_all_channels mf.compat_sequential_csum
 *   bp.compat_first_computed bp.compat_strict_upper_only bp.compat_range_all_stations
 * (The reference's counterpart is the `device=` / `arch=` string it forwards to the third-party
 * back-ends, BPMF/similarity_search.py:532, BPMF/template_search.py:554.)
 * -1 for an unknown name or a value outside the option's range. */
int bpmf_set_option(const char *name, long value);
int bpmf_get_option(const char *name, long *value, long *default_value);

/* ------------------------------------------------------------ matched filter --- */
/*
 * Serves fast_matched_filter.matched_filter(templates, moveouts, weights, data, step,
 * arch="gpu", network_sum=...) as called at BPMF/similarity_search.py:526-533 and
 * BPMF/dataset.py:4818-4827.
 *   templates (T,S,C,L) f32   moveouts (T,S,C) i32   weights (T,S,C) f32
 *   data (S,C,N) f32          n_corr = (N-L)/step + 1
 *   network_sum != 0 -> cc_out (T, n_corr)          weighted network sum
 *   network_sum == 0 -> cc_out (T, n_corr, S, C)    per-channel CC (unweighted)
 */

/* flags of bpmf_mf_run_dev */
#define BPMF_MF_DATA_PREPARED 1 /* workspace already holds this data's window energies */
#define BPMF_MF_FORCE_DIRECT 2  /* use the generic (non-MFMA) kernel */

size_t bpmf_mf_workspace_bytes(size_t L, size_t N, size_t T, size_t S, size_t C);

/* Per-day, template-independent preparation (window energies of `data` for length L).
 * Implied by bpmf_mf_run_dev unless BPMF_MF_DATA_PREPARED is set. */
int bpmf_mf_prepare_data_dev(const float *d_data, size_t L, size_t N, size_t S, size_t C,
                             void *d_workspace, size_t workspace_bytes, bpmf_stream_t stream);

int bpmf_mf_run_dev(const float *d_templates, const int32_t *d_moveouts,
                    const float *d_weights, const float *d_data, size_t step, size_t L,
                    size_t N, size_t T, size_t S, size_t C, size_t n_corr, int network_sum,
                    int flags, void *d_workspace, size_t workspace_bytes,
                    bpmf_stream_t stream, float *d_cc_out);

int bpmf_mf_run(const float *templates, const int32_t *moveouts, const float *weights,
                const float *data, size_t step, size_t L, size_t N, size_t T, size_t S,
                size_t C, size_t n_corr, int network_sum, int flags, int device,
                float *cc_out);

/* The same call with the templates block-partitioned over several GPUs inside the library
 * (one host thread per distinct device, every device holds the whole `data`): what upstream's
 * `arch="gpu"` back-end does with every visible device.  The data reach the first device from the
 * host once and the others device -> device (option multi.peer_fanout; upstream copies the day to
 * every device from the host); no CC value crosses devices.
 *   n_devices <= 0: all visible devices; devices == NULL: devices 0 .. n_devices-1. */
int bpmf_mf_run_multi(const float *templates, const int32_t *moveouts, const float *weights,
                      const float *data, size_t step, size_t L, size_t N, size_t T, size_t S,
                      size_t C, size_t n_corr, int network_sum, int flags, int n_devices,
                      const int *devices, float *cc_out);

/* The template ranges of that split: block r computes templates [bounds_out[r], bounds_out[r+1]),
 * balanced by the number of channels with a non-zero weight (the kernels skip the others) -- the
 * same rule as seismic_bpmf_amd.parallel.shard_bounds_weighted, which the torch.distributed path
 * uses.  Pure host function (no device needed); bounds_out has n_blocks + 1 entries. */
int bpmf_mf_shard_bounds(const float *weights, size_t T, size_t S, size_t C, size_t n_blocks,
                         size_t *bounds_out);

/* ------------------------------------------------------------ backprojection --- */
/*
 * Serves beampower.beampower.beamform(waveform_features, moveouts, weights_phases,
 * weights_sources, device="gpu", out_of_bounds=, reduce=) as called at
 * BPMF/template_search.py:549-558 (reduce="max") and :560-569 (reduce="none").
 *   features (S,C,N) f32     moveouts (K,S,P) i32 (samples)
 *   w_phases (S,C,P) f32     w_sources (K,S) f32
 *   reduce max  -> beam_out (N) f32 + arg_out (N) i32   (lowest source index on ties)
 *   reduce none -> beam_out (K,N) f32, arg_out unused (may be NULL)
 */
#define BPMF_BP_STRICT 0
#define BPMF_BP_FLEXIBLE 1
#define BPMF_BP_REDUCE_MAX 0
#define BPMF_BP_REDUCE_NONE 1

typedef struct bpmf_bp_plan bpmf_bp_plan; /* opaque: device-resident moveout table */

/* Build the device-resident plan of one moveout table + source weights (host arrays).
 * The reference rebuilds this table on every access (template_search.py:444-454); the
 * plan is what lets it stay resident across days.  `source_id_offset` is added to the
 * arg-max indices (global ids when the grid is sharded across GPUs). */
int bpmf_bp_plan_create(const int32_t *moveouts, const float *w_sources, size_t K, size_t S,
                        size_t P, int device, int32_t source_id_offset, bpmf_bp_plan **plan);
void bpmf_bp_plan_destroy(bpmf_bp_plan *plan);

/* Shape of a plan (diagnostics; bench.py prices the LDS roofline of the gather width in use). */
typedef struct {
    int32_t n_groups;      /* groups of sources that share one set of LDS windows */
    int32_t tile;          /* time samples per workgroup */
    int32_t lds_bytes;     /* LDS of the largest group */
    int32_t gather_bytes;  /* 8: dual (shifted) windows + ds_read_b64 gathers; 4: 4-byte gathers */
    int32_t stations_max;  /* padded station slots per source of the packed kernel (0: generic kernel) */
    int32_t waves_per_cu;  /* resident waves per CU of the beam kernel this plan dispatches to */
    /* reduce="max": the interior tiles of a two-phase grid run the 8-byte-gather kernel once per
     * station-count class of sources (<= 16, 17..32, 33..64 weighted stations), each class on the
     * largest tile (512 / 256 / 128 samples) at which its dual windows fit the LDS.  0 classes: the
     * general kernels compute everything.  With classes, tile / n_groups / lds_bytes above describe
     * the class that holds most sources. */
    int32_t n_classes;
    int32_t class_tile[3], class_sources[3], class_groups[3], class_stations_max[3];
} bpmf_bp_plan_stats;
int bpmf_bp_plan_info(const bpmf_bp_plan *plan, bpmf_bp_plan_stats *out);

size_t bpmf_bp_workspace_bytes(const bpmf_bp_plan *plan, size_t N, size_t C);

int bpmf_bp_run_dev(const bpmf_bp_plan *plan, const float *d_features,
                    const float *d_w_phases, size_t N, size_t C, int out_of_bounds,
                    int reduce, void *d_workspace, size_t workspace_bytes,
                    bpmf_stream_t stream, float *d_beam_out, int32_t *d_arg_out);

int bpmf_bp_run(const float *features, const int32_t *moveouts, const float *w_phases,
                const float *w_sources, size_t N, size_t K, size_t S, size_t C, size_t P,
                int out_of_bounds, int reduce, int device, float *beam_out, int32_t *arg_out);

/* The same call with the source grid block-partitioned over several GPUs inside the library.
 * reduce max: the per-device maxima are merged on the host in ascending block order with a
 * strict >, so ties keep the lowest source index exactly like one sequential scan;
 * reduce none: every device fills its own rows of beam_out.  n_devices / devices as above; the
 * features reach the devices like the data of bpmf_mf_run_multi (option multi.peer_fanout). */
int bpmf_bp_run_multi(const float *features, const int32_t *moveouts, const float *w_phases,
                      const float *w_sources, size_t N, size_t K, size_t S, size_t C, size_t P,
                      int out_of_bounds, int reduce, int n_devices, const int *devices,
                      float *beam_out, int32_t *arg_out);

/* Multi-GPU exchange step of reduce="max": pack (beam, source id) into one uint64 whose
 * unsigned order is (beam ascending, then source id DEscending), so that an RCCL
 * all-reduce with ncclMax over uint64 (or int64 after the bias below) yields the global
 * maximum with the lowest source id on ties; unpack restores the two vectors.
 * `as_signed` != 0 flips the top bit so that the order also holds for int64 (torch has
 * no uint64 reductions). */
int bpmf_bp_pack_max_dev(const float *d_beam, const int32_t *d_arg, size_t N, int as_signed,
                         bpmf_stream_t stream, uint64_t *d_packed);
int bpmf_bp_unpack_max_dev(const uint64_t *d_packed, size_t N, int as_signed,
                           bpmf_stream_t stream, float *d_beam, int32_t *d_arg);

/* ------------------------------------------------------- inter-template CC, batched --- */
/*
 * The core of TemplateGroup.compute_intertemplate_cc (BPMF/dataset.py:4775-4841) for ALL template
 * pairs in one pass (the reference makes one fast_matched_filter call per template, :4818-4827):
 *   out[t, u] = sum_{s,c} base_weights[t, s, c] * max_{lag} CC(trimmed template u, waveform t at lag)
 * for the pairs with pair_mask[t, u] != 0, else 0; trimmed = samples [max_lag, Lw - max_lag),
 * lag = 0 .. 2 max_lag; CC = the matched filter's normalised cross-correlation (zero moveouts),
 * 0 on channels of zero weight; the channel sum runs in NumPy's pairwise order, so the matrix
 * equals the reference's per-template loop bit for bit.  The caller symmetrises ((out + out^T)/2,
 * dataset.py:4833-4834).
 *   d_waveforms (T,S,C,Lw) f32   d_base_weights (T,S,C) f32   d_pair_mask (T,T) u8   d_out (T,T) f32
 */
size_t bpmf_intertemplate_workspace_bytes(size_t T, size_t S, size_t C, size_t max_lag);
int bpmf_intertemplate_cc_dev(const float *d_waveforms, const float *d_base_weights,
                              const uint8_t *d_pair_mask, size_t T, size_t S, size_t C, size_t Lw,
                              size_t max_lag, void *d_workspace, size_t workspace_bytes,
                              bpmf_stream_t stream, float *d_out);

/* ------------------------------------------ detection stage behind the beamformer --- */
/*
 * Device version of the two full-length steps of Beamformer.find_detections
 * (BPMF/template_search.py:574-627), so that the (N,) max-beam never leaves HBM:
 *
 * bpmf_bp_window_stats_dev: the per-window statistics of template_search.time_dependent_threshold
 * (BPMF/template_search.py:1418-1487): for q = 1 .. n_windows, window q covers the samples
 * [q*shift, min(n, q*shift + window)); d_median[q] / d_mad[q] receive np.median(window) and
 * np.median(|window - median|) in float32, bit for bit (radix select, no sort).  Both arrays need
 * n_windows + 2 entries; entries 0 and n_windows + 1 (the reference's end fills) are left to the
 * caller.  n_windows = bpmf_bp_num_windows(n, window, shift) = (n - window) / shift + 1.
 *
 * bpmf_bp_extract_peaks_dev: every rising-edge local maximum of the max-beam (the peak rule of
 * utils._detect_peaks, BPMF/utils.py:2292-2301: x[t] > x[t-1] and x[t+1] <= x[t], first and last
 * sample excluded, samples next to a NaN excluded) whose value exceeds `floor_value`, as
 * (sample, beam, source) records in arbitrary order.  *d_count receives the number found (may
 * exceed `capacity`; only the first `capacity` are stored).  d_sources may be NULL (source = 0).
 */
typedef struct bpmf_bp_peak {
    int32_t index;  /* time sample */
    float beam;     /* maxbeam[index] */
    int32_t source; /* maxbeam_sources[index] */
    int32_t pad;
} bpmf_bp_peak;

size_t bpmf_bp_num_windows(size_t n, size_t window, size_t shift);
int bpmf_bp_window_stats_dev(const float *d_beam, size_t n, size_t window, size_t shift,
                             bpmf_stream_t stream, float *d_median, float *d_mad);
int bpmf_bp_extract_peaks_dev(const float *d_beam, const int32_t *d_sources, size_t n,
                              double floor_value, uint32_t capacity, bpmf_stream_t stream,
                              uint32_t *d_count, bpmf_bp_peak *d_records);

/* ------------------------------------------------ post-CC detection threshold --- */
/*
 * Device version of the step right after the matched filter:
 * BPMF.clib.time_dependent_threshold (BPMF/clib.py:257-309 -> BPMF/libc.c:516-673, RMS variant)
 * for ALL rows of a (n_rows, n) CC matrix resident in HBM, and the extraction of the samples above
 * the threshold (BPMF/similarity_search.py:231-232, with the cap of :629), so that only candidate
 * records leave the device.  Results equal the reference's single-threaded libc.c bit for bit.
 *   half_window = window // 2, shift = int((1 - overlap) * window)   (as clib.py:292-293)
 *   gaussian: 500 standard-normal floats used to fill exact zeros (libc.c:606-612)
 *   thr_windows (n_rows, n_win): one threshold per sliding window, after "delay the jump"
 *   threshold (n_rows, n) or NULL: the step-wise expansion the reference returns
 */
size_t bpmf_tdt_num_windows(size_t n, size_t half_window, size_t shift);
size_t bpmf_tdt_workspace_bytes(size_t n_rows, size_t n, size_t half_window, size_t shift);
int bpmf_tdt_rms_dev(const float *d_series, const float *d_gaussian, float num_dev, size_t n_rows,
                     size_t n, size_t half_window, size_t shift, void *d_workspace,
                     size_t workspace_bytes, bpmf_stream_t stream, float *d_thr_windows,
                     float *d_threshold);

typedef struct bpmf_candidate {
    int32_t row;     /* template (row of the CC matrix) */
    int32_t index;   /* CC index (x step = data sample) */
    float cc;        /* CC value */
    float threshold; /* threshold it exceeded */
} bpmf_candidate;

/* Append one record per sample with cc > min(threshold, row_cap[row]) (row_cap may be NULL).
 * *d_count receives the number of candidates found (may exceed `capacity`; only the first
 * `capacity` are stored, in arbitrary order). */
int bpmf_extract_candidates_dev(const float *d_series, const float *d_thr_windows,
                                const float *d_row_cap, size_t n_rows, size_t n,
                                size_t half_window, size_t shift, uint32_t capacity,
                                bpmf_stream_t stream, uint32_t *d_count,
                                bpmf_candidate *d_records);

int bpmf_extract_candidates_mad_dev(const float *d_series, const float *d_thr_windows,
                                    const float *d_row_cap, size_t n_rows, size_t n, size_t window,
                                    size_t shift, uint32_t capacity, bpmf_stream_t stream,
                                    uint32_t *d_count, bpmf_candidate *d_records);

/* The optional validation of MatchedFilter.select_cc_indexes (BPMF/similarity_search.py:253-272) on the device:
 * for detection q, how many samples of row d_rows[q] of the (n_rows, n) matrix lie strictly below d_level[q] in
 * [d_start[q], d_start[q] + d_len_left[q]) -> d_below[2 q] and in the d_len_right[q] samples behind -> d_below[2 q + 1]
 * (float32 compares, like the reference's `cc1 < cc_at_mean_plus_1sig[cc_idx[i]]`; one workgroup per detection; a
 * few thousand detections of a few hundred samples each: the CC rows stay in HBM).  The host keeps a detection when
 * min(below_left / len_left, below_right / len_right) >= anomalous_cdf_at_mean_plus_1sig. */
int bpmf_count_below_dev(const float *d_series, size_t n_rows, size_t n, size_t n_detections,
                         const int32_t *d_rows, const int64_t *d_start, const int32_t *d_len_left,
                         const int32_t *d_len_right, const float *d_level, bpmf_stream_t stream,
                         int32_t *d_below);

/* ------------------------------------------------- robust statistics (stats.hip) --- */
/* np.median and MAD (median of |x - median|, float32 like NumPy) of every row of a (rows, n)
 * device array; skip_zeros != 0: over the samples != 0 only (`a[a != 0]`).  NaN for an empty
 * selection or a row that holds a NaN.  d_n_zero (may be NULL) receives the zeros per row.
 * Serves saturated_envelopes (BPMF/template_search.py:1547-1561). */
int bpmf_row_median_mad_dev(const float *d_x, size_t rows, size_t n, int skip_zeros,
                            bpmf_stream_t stream, float *d_median, float *d_mad, int64_t *d_n_zero);

/* The same with a device workspace of bpmf_row_median_mad_workspace_bytes(rows, n) bytes: rows of at least
 * option stats.row_grid_min_n samples are then read twice by workgroups from all over the chip instead of
 * seven times by one workgroup each (a day-long row: 60 envelopes 9.2 -> under 1 ms); same bits.
 * d_workspace NULL = bpmf_row_median_mad_dev. */
size_t bpmf_row_median_mad_workspace_bytes(size_t rows, size_t n);
int bpmf_row_median_mad_ws_dev(const float *d_x, size_t rows, size_t n, int skip_zeros,
                               void *d_workspace, size_t workspace_bytes, bpmf_stream_t stream,
                               float *d_median, float *d_mad, int64_t *d_n_zero);

/* The element-wise steps of the envelope (BPMF/template_search.py:1598-1617) around the library FFTs, one pass each:
 * d_spectrum (rows, bins) complex128, the rfft of the traces, becomes -i X in place with the DC bin -- and, when the
 * trace length is even, the last (Nyquist) bin -- cleared: its irfft is the Hilbert transform; then
 * d_out[i] = (float) sqrt((double) d_x[i]^2 + d_h[i]^2), float64 arithmetic, over `total` samples.  rows <= 65535. */
int bpmf_hilbert_spectrum_dev(void *d_spectrum, size_t rows, size_t bins, int n_is_even, bpmf_stream_t stream);
int bpmf_envelope_combine_dev(const float *d_x, const double *d_h, size_t total, bpmf_stream_t stream,
                              float *d_out);

/* The last step of saturated_envelopes (BPMF/template_search.py:1562-1572) in one pass over the (rows, n)
 * envelopes: (x - median[row]) / mad[row] in float32, 0 for missing samples (exact zeros) and for rows with
 * dead[row] != 0, capped at `cap` (np.minimum: a NaN stays a NaN).  d_out may be d_x.  rows <= 65535 per call. */
int bpmf_saturate_rows_dev(const float *d_x, const float *d_median, const float *d_mad, const int32_t *d_dead,
                           size_t rows, size_t n, float cap, bpmf_stream_t stream, float *d_out);

/* time_dependent_threshold(time_series, sliding_window, overlap, threshold_type="mad",
 * white_noise) of BPMF/similarity_search.py:1079-1113 for every row of a (rows, n) CC matrix:
 * `window` = sliding_window, `shift` = int((1 - overlap) * sliding_window).  d_thr_windows
 * (rows, n_windows) receives centre + num_dev * deviation after the two neighbour-maximum
 * passes; d_thr_full (rows, n), if not NULL, the threshold of every sample.  Row r replaces its
 * zeros, in order, by white_noise[0 .. n_zeros(r)) * deviation0 + centre0; -1 if a row holds
 * more zeros than n_noise (checked with one small D2H and a stream synchronise when n_noise < n:
 * hand over n values, as the reference draws them, to stay asynchronous).
 * Limits: rows <= 65535 per call (gridDim.y; callers with more rows chunk them -- ThresholdGPU
 * does), n < 2^31.  The workspace holds NO copy of the series (since round 5 the zeros are replaced where the
 * medians read them): per row the zero-rank tables (~n / 32 bytes), the window values, and -- for rows of at
 * least option stats.row_grid_min_n samples -- the buffers the two-read row statistics collect the middle of
 * a row in (~n / 6 bytes + 100 KB); bpmf_tdt_mad_workspace_bytes: ~1 GB for 500 rows of a day at 100 Hz, a few
 * KB per row for short rows.  Bound it by calling with fewer rows at a time. */
size_t bpmf_tdt_mad_num_windows(size_t n, size_t window, size_t shift);
size_t bpmf_tdt_mad_workspace_bytes(size_t rows, size_t n, size_t window, size_t shift);
int bpmf_tdt_mad_dev(const float *d_series, const float *d_white_noise, size_t n_noise,
                     float num_dev, size_t rows, size_t n, size_t window, size_t shift,
                     void *d_workspace, size_t workspace_bytes, bpmf_stream_t stream,
                     float *d_thr_windows, float *d_thr_full, int64_t *d_n_zero);

/* scipy.stats.kurtosis(row) (Fisher, biased: m4 / m2^2 - 3, NaN for a constant row) of every row,
 * float32 with NumPy's pairwise summation order: the `sanity_check` of
 * MatchedFilter._find_detections_t (BPMF/similarity_search.py:633-642).  rows <= 65535 per call. */
size_t bpmf_row_kurtosis_workspace_bytes(size_t rows, size_t n);
int bpmf_row_kurtosis_dev(const float *d_x, size_t rows, size_t n, void *d_workspace,
                          size_t workspace_bytes, bpmf_stream_t stream, float *d_kurtosis);

/* The same sums, handed out before the last expression: d_parts (rows, 3) receives (mean, m2, m4) of every row
 * in float32 (d_kurtosis may be NULL).  Why: SciPy evaluates `m4 / m2**2.0 - 3` on NumPy SCALARS when the input is
 * one series -- which is how BPMF calls it (similarity_search.py:640) -- and a NumPy scalar `**` is the C
 * library's powf, which glibc rounds faithfully but not always correctly (one case in 2500 random rows: 2 ulp in
 * the kurtosis); on an ARRAY the same expression is the correctly rounded square, which is what
 * bpmf_row_kurtosis_dev computes.  A host that wants the reference's very bits finishes the expression with its own
 * NumPy scalars (seismic_bpmf_amd.workflow.row_excess_kurtosis does). */
int bpmf_row_kurtosis_parts_dev(const float *d_x, size_t rows, size_t n, void *d_workspace,
                                size_t workspace_bytes, bpmf_stream_t stream, float *d_kurtosis,
                                float *d_parts);

/* ------------------------------------------------------------ running kurtosis --- */
/*
 * Device version of BPMF.clib.kurtosis (BPMF/clib.py:86-102 -> BPMF/libc.c:11-53): kurto[ch][n],
 * n >= W, is the kurtosis of the W samples before n; written only where the window variance
 * exceeds 1e-6 -- zero-initialise d_kurto like the reference wrapper does.
 *   d_signal, d_kurto (n_channels, length) f32, n_channels = stations x components
 */
int bpmf_kurtosis_dev(const float *d_signal, int W, size_t n_channels, size_t length,
                      bpmf_stream_t stream, float *d_kurto);

/* ------------------------------------------------------ peak suppression (host) --- */
/*
 * The height-ordered suppression loop of BPMF/utils.py:2334-2345 (`_detect_peaks`, mpd > 1), which
 * the reference runs as an O(peaks^2) NumPy loop: peaks are visited from the tallest down and a
 * kept peak removes every other peak within +-mpd samples.  Same visiting order (the caller passes
 * the reference's own argsort), same result, O(peaks) time.  Host arrays.
 *   positions (n) i64 ascending sample indices of the candidate peaks
 *   order     (n) i64: order[q] = index into `positions` of the q-th peak to visit
 *   keep      (n) u8 out, indexed like `positions` (1 = survives)
 */
int bpmf_suppress_peaks(const int64_t *positions, const int64_t *order, size_t n, double mpd,
                        uint8_t *keep);

/* ------------------------------------------------------------ grid decimation --- */
/*
 * Device version of BPMF.clib.find_similar_sources (BPMF/clib.py:104-221), i.e. of
 * BPMF/libc.c:find_similar_moveouts (:55-223, method 0 = "smallest") and
 * find_similar_moveouts2 (:225-387, method 1 = "closest"): the one-time O(K^2 S) decimation
 * of the source grid that precedes the backprojection.  Host arrays, same argument order as the
 * reference's C functions (num_threads replaced by the method / device); the result equals the
 * reference run single-threaded.
 *   moveouts (K, S) f32 seconds; redundant_sources (K) i32 out (1 = redundant)
 */
int bpmf_find_similar_sources(const float *moveouts, const float *source_longitude,
                              const float *source_latitude, const float *cell_longitude,
                              const float *cell_latitude, float threshold, size_t n_sources,
                              size_t n_stations, size_t n_cells_longitude,
                              size_t n_cells_latitude, size_t n_stations_for_diff, int method,
                              int device, int32_t *redundant_sources);

#ifdef __cplusplus
}
#endif
#endif /* BPMF_HIP_H */
