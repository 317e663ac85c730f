#!/usr/bin/env python3
"""Benchmark of the two BPMF hot paths on MI355X (contract: see the task brief / DESIGN.md s6).

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM:
  * headline (``value``): matched filter, BASELINE.json configs[1]
    (500 templates x 20 stations x 3 components, 256-sample templates, 1 day @ 100 Hz,
    step 1): one step = data preparation (window energies) + CC of all templates; the
    (T, n_corr) CC matrix stays in HBM.  metric = million network-CC-samples/s.
  * secondary (``bp`` object): backprojection, configs[2] (50 000 sources x 20 stations x 3
    components x 2 phases, 1 day @ 50 Hz, 10-closest-station weights, reduce="max", strict).
N > 1 is weak scaling: every rank owns its own shard of templates (MF) / of the source grid
(BP) against a replicated day of data; MF has no data-path collective, BP ends each step with
the packed (max, arg-max) all-reduce over RCCL.

The JSON line also carries ``roofline`` (dominant kernel, HIP events recorded on the launch
stream inside the timed region) and ``cpu_baseline`` (the CPU oracle timed on this box's host
cores on a bounded sample of the same workload; rank 0, N = 1 only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
MF_LABEL = {"cfg1": "configs[0]", "cfg2": "configs[1]", "cfg4_per_gpu": "configs[3], one GPU's share (5000 templates / 8)"}
BP_LABEL = {"cfg3": "configs[2]", "cfg5_per_gpu": "configs[4], one GPU's share (1M sources / 8)"}
LDS_B32_PEAK_TBS = 256 * 128 * 2.4e9 / 1e12   # ds_read_b32: 128 B/clk/CU x 256 CU x 2.4 GHz
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mf-config", default=None,
                    help="default: cfg2 (configs[1]) per GPU at every N; cfg4_per_gpu = one GPU's share of configs[3]")
    ap.add_argument("--bp-config", default=None,
                    help="default: cfg3 (configs[2]) per GPU at every N; cfg5_per_gpu = one GPU's share of configs[4]")
    ap.add_argument("--skip-shares", action="store_true",
                    help="N > 1: do not time the per-GPU shares of configs[3] / configs[4] as extras")
    ap.add_argument("--skip-dense", action="store_true", help="skip the dense-station-weight BP extras")
    ap.add_argument("--skip-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (rocprofv3 --pmc child passes, ~2 min); "
                         "the committed figures of profiles/ are quoted instead")
    ap.add_argument("--skip-e2e", action="store_true", help="skip the host-pointer end-to-end extra")
    ap.add_argument("--skip-bp", action="store_true")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="target duration of the cpu_baseline sample")
    ap.add_argument("--templates", type=int, default=0, help="override T (debug)")
    return ap.parse_args()


# ------------------------------------------------------------------ synthetic inputs ---
def mf_inputs_device(cfg, device, seed, rank=0):
    """Device-side equivalent of synthetic.make_mf_inputs (same conditioning, torch RNG).  The day of
    noise is the same on every rank (the data are replicated in a template-sharded job); templates,
    moveouts and the events planted for them are the rank's own."""
    T, S, C, L, N = cfg["T"], cfg["S"], cfg["C"], cfg["L"], cfg["N"]
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    data = torch.randn((S, C, N), device=device, generator=g)
    g.manual_seed(seed + 1 + 1000 * rank)
    raw = torch.randn((T, S, C, L + 4), device=device, generator=g)
    tmpl = sum(raw[..., k:k + L] for k in range(5))
    tmpl = tmpl - tmpl.mean(dim=-1, keepdim=True)
    tmpl = (tmpl / tmpl.std(dim=-1, keepdim=True)).contiguous()
    mv_p = torch.randint(0, 1501, (T, S), device=device, generator=g)
    mv_s = mv_p + torch.randint(0, 1501, (T, S), device=device, generator=g)
    mv = torch.empty((T, S, C), dtype=torch.int32, device=device)
    mv[:, :, 0] = mv_p
    mv[:, :, 1:] = mv_s[:, :, None]
    w = torch.full((T, S, C), 1.0 / (S * C), device=device)
    # plant 5 scaled copies of every template (events) before the per-channel normalisation
    n_ev = 5
    slots = torch.randint(0, (N - L - 3002) // (4 * L), (T, n_ev), device=device, generator=g)
    amps = 1.5 + 2.5 * torch.rand((T, n_ev), device=device, generator=g)
    ar = torch.arange(L, device=device)
    for t in range(T):
        for e in range(n_ev):
            j = (slots[t, e] * 4 * L + mv[t].long())[..., None] + ar        # (S, C, L)
            data.scatter_add_(2, j, amps[t, e] * tmpl[t])
    data /= data.std(dim=-1, keepdim=True)
    planted = (slots * 4 * L).cpu().numpy()            # (T, n_ev) CC indices of the planted events
    return tmpl, mv, w, data, planted


def bp_inputs(cfg, device, seed, rank, world):
    from seismic_bpmf_amd import synthetic as syn
    # one GPU's share of configs[4] is a depth slab of the 125 x 125 x 64 grid: rank r scans slab r
    slab = (rank, 64) if cfg["grid"] == (125, 125, 8) else None
    geo = syn.make_bp_geometry(cfg["grid"], cfg["S"], cfg["P"], cfg["sr"], seed=seed, depth_slab=slab)
    g = torch.Generator(device=device)
    g.manual_seed(seed + 1)
    S, C, N = cfg["S"], cfg["C"], cfg["N"]
    feat = torch.randn((S, C, N), device=device, generator=g).abs_()
    wp = torch.as_tensor(syn.phase_weights(S, C, cfg["P"]), device=device)
    # 20 planted events: Gaussian bumps along the moveouts of random sources
    rng = np.random.default_rng(seed + 2)
    sig = 0.2 * cfg["sr"]
    half = int(4 * sig)
    bump = torch.as_tensor(8.0 * np.exp(-0.5 * (np.arange(-half, half + 1) / sig) ** 2),
                           dtype=torch.float32, device=device)
    tau = geo["moveouts"]
    planted = []
    for _ in range(20):
        k0 = int(rng.integers(0, tau.shape[0]))
        t0 = int(rng.integers(half, N - int(tau.max()) - 2 * half))
        planted.append((k0, t0))
        for s in range(S):
            for c in range(C):
                x = t0 + int(tau[k0, s, 0 if c == 0 else 1])
                feat[s, c, x - half:x + half + 1] += bump
    geo["planted"] = planted
    return geo, feat, wp


def bp_detection_stage(beam, arg, geo, bcfg):
    """What follows the beamformer in BPMF (template_search.py:574-627), untimed extra: sliding
    median/MAD threshold, peaks at least 5 s apart, snap + unique, source of each peak -- on the
    full day with the max-beam left in HBM (workflow.beam_detections_device: window medians by
    radix select and local-maximum extraction on the device, the reference's index logic on the
    compacted candidates).  Every planted event must come out within the half-width of its bump,
    located at the planted source or one with an equal beam."""
    from seismic_bpmf_amd import postprocess as pp
    from seismic_bpmf_amd.threshold import BeamDetectorGPU
    from seismic_bpmf_amd.workflow import beam_detections_device
    window = int(pp.sec_to_samp(1800.0, bcfg["sr"]))
    mpd = int(pp.sec_to_samp(5.0, bcfg["sr"]))
    beam_detections_device(beam, arg, mpd=mpd, window=window, n_dev=15.0, overlap=0.75)   # warm-up
    torch.cuda.synchronize()
    det = BeamDetectorGPU(device=beam.device.index)
    t0 = time.perf_counter()
    det.window_stats(beam, window, 0.75)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    peaks, peak_sources, nodes = beam_detections_device(beam, arg, mpd=mpd, window=window, n_dev=15.0,
                                                        overlap=0.75)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    found = same_source = 0
    for k0, ts in geo["planted"]:
        hit = np.flatnonzero(np.abs(peaks - ts) <= 5)
        if hit.size:
            found += 1
            same_source += int(peak_sources[hit[0]] == k0)
    return {"total_ms": round((t2 - t1) * 1e3, 2), "of_which_window_stats_ms": round((t1 - t0) * 1e3, 2),
            "full_length_d2h": False, "detections": int(peaks.size),
            "planted": len(geo["planted"]), "planted_found_within_5_samples": found,
            "planted_located_at_planted_source": same_source}


# ------------------------------------------------------------------ HBM traffic (PMC) ---
def measure_traffic(target, kernels, extra_args=(), timeout=240):
    """HBM-side bytes per launch of the dominant kernel, measured NOW: two separate
    `rocprofv3 --pmc` passes (FETCH_SIZE, then WRITE_SIZE -- the two do not fit one pass, counters
    only, no other trace domain: MI355X_MICROARCH.md, HBM / rocprofv3 sections) over a small script that
    launches the same kernel on the same workload (tools/prof_mf.py 500 / tools/prof_bp.py), as child
    processes after the timed region.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at
    64 bytes), WRITE_SIZE is taken as is (it equals the output size: the calibration).  Returns a
    dict or None when rocprofv3 is not usable here."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bpmf_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.join(ROOT, "tools", target)] + list(extra_args)
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=timeout,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
            vals = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    for k in kernels:
                        if k in r["Kernel_Name"] and r["Counter_Name"] == counter:
                            vals.setdefault(k, []).append(float(r["Counter_Value"]))
            if not vals:
                return None
            # per launch of the workload = the sum over the kernels that make it up (interior + edge)
            out[counter] = sum(sum(v) / len(v) for v in vals.values()) * 1024.0
            out[counter + "_launches"] = max(len(v) for v in vals.values())
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out["hbm_bytes_per_launch"] = 2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]
    return out


# ----------------------------------------------------------------------- CPU baseline ---
def host_cpu_facts():
    """What this process may really use: logical CPUs, the affinity mask, the cgroup CPU quota."""
    facts = {"logical_cpus": os.cpu_count() or 1}
    try:
        facts["affinity_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        facts["affinity_cpus"] = facts["logical_cpus"]
    quota = None
    try:                                                 # cgroup v2: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
        facts["cgroup_cpu_max"] = f"{q} {per}"
    except Exception:
        try:                                             # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q <= 0 else q / per
            facts["cgroup_cpu_max"] = f"{q:g} {per:g}"
        except Exception:
            facts["cgroup_cpu_max"] = "unreadable"
    facts["cgroup_quota_cpus"] = quota
    usable = facts["affinity_cpus"]
    if quota is not None:
        usable = max(1, min(usable, int(quota + 0.5)))
    facts["usable_cpus"] = usable
    try:
        facts["smt_active"] = open("/sys/devices/system/cpu/smt/active").read().strip() == "1"
    except Exception:
        facts["smt_active"] = None
    try:
        facts["loadavg_1min"] = float(open("/proc/loadavg").read().split()[0])
    except Exception:
        pass
    try:
        with open("/proc/cpuinfo") as f:
            facts["cpu_model"] = next(ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name"))
    except Exception:
        facts["cpu_model"] = "unknown"
    return facts


def cpu_baseline(cfg, target_seconds):
    """Time the CPU oracle (rebuilt -march=native for this host) on (1) a bounded sample of the
    headline workload's shape -- `value` --, (2) a smaller sample of the same shape on ONE thread and
    on all threads (`scaling`: what the stated core count is worth), and (3) BASELINE configs[0], the
    reference's own CPU-runnable case, in full, both ways, each repeated for >= 1 s of work
    (SURVEY.md s8d).  `cores` is the number of threads used = the CPUs this process may run on
    (affinity mask and cgroup quota applied), not os.cpu_count()."""
    from oracle import oracle
    from seismic_bpmf_amd import synthetic as syn
    try:
        lib = oracle.load(oracle.build(march="native", out_dir="/tmp/bpmf_oracle_native"))
        march = "native"
    except Exception:
        lib = oracle.load()
        march = "x86-64-v3"
    facts = host_cpu_facts()
    # (the CPUs the process may use; NOT omp_get_max_threads(): a launcher exports OMP_NUM_THREADS=1 to its ranks, and an earlier
    # one-thread timing leaves the runtime at 1 -- the oracle is told its thread count on every call)
    cores = max(1, facts["usable_cpus"])
    S, C, L = cfg["S"], cfg["C"], cfg["L"]

    def run(inp, nth):
        t0 = time.perf_counter()
        oracle.matched_filter(inp["templates"], inp["moveouts"], inp["weights"], inp["data"], 1,
                              num_threads=nth, lib=lib)
        return time.perf_counter() - t0

    def repeat(inp, nth, seconds, max_runs=200):          # mean wall time over >= `seconds` of work
        run(inp, nth)                                      # warm: thread pool, page cache
        total, n = 0.0, 0
        while total < seconds and n < max_runs:
            total += run(inp, nth)
            n += 1
        return total / n, n

    T = min(cfg["T"], 8)
    # (2) scaling sample: sized so that one thread needs ~3 s
    probe = syn.make_mf_inputs(T, S, C, L, 60_000, seed=9, n_events=0)
    run(probe, 1)
    rate1 = T * (60_000 - L + 1) / run(probe, 1)
    n_s = int(min(cfg["N"], max(60_000, rate1 * 3.0 / T)))
    n_s -= n_s % 1000
    smp_s = syn.make_mf_inputs(T, S, C, L, n_s, seed=7, n_events=0)
    d1, r1 = repeat(smp_s, 1, 2.5, max_runs=4)
    dall, rall = repeat(smp_s, cores, 1.0)
    speedup = d1 / dall
    scaling = {"sample": f"{T} templates x {S} stations x {C} comp, L={L}, N={n_s}",
               "seconds_1_thread": round(d1, 4), "runs_1_thread": r1,
               f"seconds_{cores}_threads": round(dall, 5), f"runs_{cores}_threads": rall,
               "value_1_thread": round(T * (n_s - L + 1) / d1 / 1e6, 4),
               f"value_{cores}_threads": round(T * (n_s - L + 1) / dall / 1e6, 4),
               "speedup": round(speedup, 2), "speedup_per_thread": round(speedup / cores, 3),
               "gflops_1_thread": round(2.0 * L * S * C * T * (n_s - L + 1) / d1 / 1e9, 1)}
    # (1) headline sample: whole seconds of the same day for ~target seconds on all threads
    rate = T * (n_s - L + 1) / dall
    n = int(min(cfg["N"], max(120_000, rate * target_seconds / T)))
    n -= n % 1000
    smp = syn.make_mf_inputs(T, S, C, L, n, seed=8, n_events=0)
    run(smp, cores)
    dt = run(smp, cores)
    import ctypes
    ph = (ctypes.c_double * 4)()
    lib.bpmf_oracle_last_phase_seconds(ph)
    value = T * (n - L + 1) / dt / 1e6
    # (3) BASELINE configs[0] in full: 4 templates x 8 stations x 3 comp, L = 128, 1 h @ 50 Hz
    c0 = syn.MF_CONFIGS["cfg1"]
    inp0 = syn.make_mf_inputs(c0["T"], c0["S"], c0["C"], c0["L"], c0["N"], seed=20260929)
    n0 = c0["T"] * (c0["N"] - c0["L"] + 1)
    d_all, n_all = repeat(inp0, cores, 1.0)
    d_one, n_one = repeat(inp0, 1, 2.0, max_runs=64)
    return {"value": round(value, 4), "unit": "M CC-samples/s", "cores": cores,
            "kind": "port", "cpu_model": facts["cpu_model"], "host": facts,
            "sample": f"{T} templates x {S} stations x {C} comp, L={L}, N={n} samples of the "
                      f"{cfg['N']}-sample day, step 1; oracle/bpmf_oracle.c mf_cpu (C99+OpenMP, "
                      f"gcc -O3 -march={march}), {dt:.1f} s wall on {cores} threads "
                      f"(preparation {ph[0]:.2f} s, correlation loop {ph[1]:.2f} s)",
            "gflops": round(2.0 * L * S * C * value * 1e6 / 1e9, 1),
            "scaling": scaling,
            "configs0": {"workload": "BASELINE configs[0] in full: 4 templates x 8 stations x 3 comp, L=128, "
                                     "N=180000 (1 h @ 50 Hz), step 1",
                         "value": round(n0 / d_all / 1e6, 4), "cores": cores, "seconds": round(d_all, 5),
                         "repeats": n_all,
                         "value_1_thread": round(n0 / d_one / 1e6, 5), "seconds_1_thread": round(d_one, 4),
                         "repeats_1_thread": n_one, "speedup": round(d_one / d_all, 2),
                         "unit": "M CC-samples/s"}}


def cpu_baseline_bp(bcfg, geo, target_seconds):
    """The CPU oracle's beamformer (rebuilt -march=native for this host) on a bounded sample of the BP
    workload: `value` on all usable threads, plus the same shape on ONE thread and the speed-up
    (SURVEY.md s8d asks for both timings on both paths)."""
    from oracle import oracle
    from seismic_bpmf_amd import synthetic as syn
    try:
        lib = oracle.load(oracle.build(march="native", out_dir="/tmp/bpmf_oracle_native"))
        march = "native"
    except Exception:
        lib = oracle.load()
        march = "x86-64-v3"
    facts = host_cpu_facts()
    # (the CPUs the process may use; NOT omp_get_max_threads(): a launcher exports OMP_NUM_THREADS=1 to its ranks, and an earlier
    # one-thread timing leaves the runtime at 1 -- the oracle is told its thread count on every call)
    cores = max(1, facts["usable_cpus"])
    S, C, P = bcfg["S"], bcfg["C"], bcfg["P"]
    K = min(geo["moveouts"].shape[0], 2000)
    mv, ws = geo["moveouts"][:K], geo["weights_sources"][:K]
    wp = syn.phase_weights(S, C, P)
    rng = np.random.default_rng(3)

    def run(feat, nth):
        t0 = time.perf_counter()
        oracle.beamform(feat, mv, wp, ws, "strict", "max", num_threads=nth, lib=lib)
        return time.perf_counter() - t0

    n = 100_000
    feat = np.abs(rng.standard_normal((S, C, n))).astype(np.float32)
    run(feat, cores)                                     # warm: thread pool
    dt = run(feat, cores)
    # one thread against all threads on the same sample (sized for ~2 s on one thread)
    d1p = run(feat[:, :, :20_000], 1)
    n1 = int(min(n, max(20_000, 20_000 * 2.0 / d1p)))
    n1 -= n1 % 1000
    f1 = np.ascontiguousarray(feat[:, :, :n1])
    d_one = run(f1, 1)
    d_all = min(run(f1, cores), run(f1, cores))
    # headline sample
    n = int(min(bcfg["N"], max(n, n / dt * target_seconds)))
    n -= n % 1000
    feat = np.abs(rng.standard_normal((S, C, n))).astype(np.float32)
    dt = run(feat, cores)
    s_act = float((ws != 0).sum(axis=1).mean())
    return {"value": K * n / dt, "unit": "grid-points x samples / s", "cores": cores, "kind": "port",
            "cpu_model": facts["cpu_model"], "host": facts,
            "sample": f"first {K} sources of the grid x N={n} samples, {s_act:.0f} closest stations, strict, "
                      f"reduce=max; oracle/bpmf_oracle.c bp_cpu (C99+OpenMP, gcc -O3 -march={march}), "
                      f"{dt:.1f} s wall on {cores} threads",
            "gather_gb_per_s": round(4.0 * s_act * P * K * n / dt / 1e9, 1),
            "scaling": {"sample": f"{K} sources x N={n1} samples", "seconds_1_thread": round(d_one, 4),
                        f"seconds_{cores}_threads": round(d_all, 5),
                        "value_1_thread": K * n1 / d_one, f"value_{cores}_threads": K * n1 / d_all,
                        "speedup": round(d_one / d_all, 2), "speedup_per_thread": round(d_one / d_all / cores, 3)}}


# ------------------------------------------------------- detection stage (untimed extra) ---
def detection_stage(cc, planted, mv, w, local_rank, dist, device, t_offset=0):
    """What follows the hot path in BPMF (similarity_search.py:548-666), on the CC matrix that is still in
    HBM, through the PRODUCT's own functions: workflow.cc_detections (RMS threshold on the device,
    candidates above it, the reference's pair-wise merge on the host) and -- for N > 1 --
    workflow.detections_to_records + parallel.allgather_varlen, the all-gather of the per-rank peak
    records that is the path's only inter-GPU traffic (the same calls
    workflow.sharded_matched_filter_detections makes).  Reported beside the headline numbers, never
    inside the timed region.  Also a full-size correctness check: every planted event must be detected
    at exactly its planted CC index."""
    from seismic_bpmf_amd import parallel, workflow
    n_corr = cc.shape[1]
    sr = 100.0
    window_s = min(1800.0, max(10.0, n_corr / 8 / sr))     # 30 min; short series (configs[0]): 8 windows
    wn = np.random.default_rng(5).standard_normal(500).astype(np.float32)
    tm = {}
    det = workflow.cc_detections(cc, mv.cpu().numpy(), w.cpu().numpy(), step=1, sr=sr, threshold_window_dur=window_s,
                                 minimum_interevent_time=5.12, n_dev=8.0, overlap=0.25, white_noise=wn,
                                 device=local_rank, remove_edges=False, sanity_check=False, with_values=True,
                                 timings=tm)
    found = exact = n_det = 0
    for t in range(planted.shape[0]):
        idx = det[t][0]
        n_det += len(idx)
        exact += len(set(idx.tolist()) & set(planted[t].tolist()))
        found += planted.shape[1]
    out = {"threshold_ms": round(tm["threshold_ms"], 1), "candidates_ms": round(tm["candidates_ms"], 1),
           "host_merge_ms": round(tm["merge_ms"], 1), "candidates": tm["candidates"], "detections": n_det,
           "planted": found, "planted_found_at_exact_index": exact,
           "through": "workflow.cc_detections (sanity_check off: the kurtosis pass is timed beside it)"}
    if dist is None:
        # the two optional passes of the same stage, untimed extras at N = 1: the reference's sanity check
        # (scipy.stats.kurtosis of every CC series, similarity_search.py:633-642; on by default there) and the
        # MAD variant of the threshold (similarity_search.py:1079-1113), each timed on its second call
        from seismic_bpmf_amd.threshold import ThresholdGPU
        th = ThresholdGPU(device=local_rank)
        wn_mad = np.random.default_rng(6).standard_normal(20_000).astype(np.float32)
        for name, fn in (("kurtosis_ms", lambda: workflow.row_excess_kurtosis(cc)),
                         ("mad_threshold_ms", lambda: th.time_dependent_threshold_mad(
                             cc, int(window_s * sr), 8.0, overlap=0.25, white_noise=wn_mad, expand=False))):
            try:
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                out[name] = round((time.perf_counter() - t0) * 1e3, 1)
            except Exception as e:                      # (an extra must not cost the line)
                out[name] = f"failed: {e}"
        del th
        torch.cuda.empty_cache()
    if dist is not None:
        # "RCCL all-gather of CC peaks" (BASELINE configs[3]): every rank contributes its MERGED
        # detections (global template id, CC index, cc, threshold) -- a few thousand 32-byte records,
        # never the CC matrix.  The buffer is sized by the largest rank's count, nothing is dropped.
        mine = workflow.detections_to_records(det, t_offset=t_offset)
        rec = torch.as_tensor(mine, device=device)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        parts = parallel.allgather_varlen(rec)
        torch.cuda.synchronize()
        out["allgather_records_ms"] = round((time.perf_counter() - t3) * 1e3, 2)
        out["records_all_ranks"] = int(sum(len(p) for p in parts))
        out["records_this_rank"] = int(mine.shape[0])
        out["own_records_round_trip_exact"] = bool(np.array_equal(parts[dist.get_rank()].cpu().numpy(), mine))
    return out


# --------------------------------------------------------------------- the JSON line ---
# keys the driver's contract names (+ the two objects of the hot-path tier); tests/test_bench_cli.py holds every
# N to them
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def cgroup_throttle():
    """(periods in which this container's CPU quota ran out, milliseconds its threads stood frozen for it, summed over
    CPUs) so far -- cgroup v2 cpu.stat, v1 cpu.stat as a fallback; (None, None) where neither is readable.  A host-pointer
    call that overlaps a throttled period stands still on the host side whatever the library does
    (profiles/r06_bp_e2e.txt)."""
    for path, key_n, key_t, div in (("/sys/fs/cgroup/cpu.stat", "nr_throttled", "throttled_usec", 1e3),
                                    ("/sys/fs/cgroup/cpu/cpu.stat", "nr_throttled", "throttled_time", 1e6)):
        try:
            kv = dict(ln.split() for ln in open(path))
            return int(kv[key_n]), float(kv[key_t]) / div
        except Exception:
            continue
    return None, None


def throttle_delta(a, b):
    if a[0] is None or b[0] is None:
        return None
    return {"periods": b[0] - a[0], "ms": round(b[1] - a[1], 1)}


def merge_rank_stats(ranks_info, per_rank):
    """Per-rank {step time, kernel time, roofline fraction} (all_gather_object of every rank's own numbers) into
    ranks_info["ranks"][i], matched by rank; also the spread of the kernel times (slowest / fastest)."""
    by_rank = {int(r["rank"]): r for r in per_rank if r}
    for entry in ranks_info.get("ranks", []):
        entry.update({k: v for k, v in by_rank.get(int(entry["rank"]), {}).items() if k != "rank"})
    for key in ("mf_kernel_ms", "bp_kernel_ms"):
        vals = [r[key] for r in by_rank.values() if isinstance(r.get(key), (int, float)) and r[key] == r[key]]
        if vals:
            ranks_info[key + "_spread"] = {"min": min(vals), "max": max(vals), "slowest_over_fastest": round(max(vals) / min(vals), 4)}
    return ranks_info


def assemble_line(*, args, world, n_gpus, mf_value, mf_dt, dims, peak, roofline, cpu, e2e, mf_shapes, bp_obj, detect,
                  shares, compat, split16, ranks_info):
    """The one JSON line of a run, from what the ranks measured (pure Python: the CPU suite builds the N = 8
    line from stubbed numbers and holds it to the keys of the N = 1 line)."""
    T, S, C, L, N = dims
    return {
        "metric": "million network-CC-samples/s (matched filter)",
        "value": round(mf_value, 2), "unit": "M CC-samples/s",
        "n_gpus": n_gpus,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(mf_dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE {MF_LABEL.get(args.mf_config, args.mf_config)}: {T} templates x {S} stations x {C} comp, "
                               f"L={L}, N={N} (1 day @ 100 Hz), step 1, per GPU",
                   "same_per_gpu_workload_at_every_n": args.mf_config == "cfg2",
                   "channel_cc_samples_per_s": round(mf_value * 1e6 * S * C, 0),
                   "parallelism": (f"templates sharded x{world}: every rank holds {T} templates of a {world * T}-template "
                                   "job, data replicated, no data-path collective; all-gather of merged peak records"
                                   if world > 1 else "single GPU"),
                   "row0_peak_cc": round(peak, 4)},
        "roofline": roofline, "cpu_baseline": cpu, "end_to_end": e2e, "mf_shapes": mf_shapes, "bp": bp_obj,
        "detection": detect, "shares": shares, "compat": compat, "mf_split16": split16,
        "ranks": ranks_info,
    }


# ------------------------------------------------------------------------------- main ---
def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves.

    The driver's documented N > 1 command wraps bench.py in `python -m torch.distributed.run`; a bare
    `python bench.py --gpus 8` (the shape of the N = 1 command) used to run ONE rank and report n_gpus 1
    without a word (round-4 review).  Now it re-executes itself under torch.distributed.run with
    --nproc-per-node N on 127.0.0.1 and a free port, or fails loudly when fewer than N devices are visible.
    BPMF_BENCH_FORCE_DIST=1 takes the same route for N = 1 (RCCL initialised with one rank)."""
    import socket
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} HIP device(s) visible "
                         f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')!r}, "
                         f"ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES')!r}); refusing to run fewer ranks")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this host driver
    env["BPMF_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.write(f"[bench] launching {args.gpus} rank(s): {' '.join(cmd)}\n")
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def main():
    args = parse()
    force_dist = os.environ.get("BPMF_BENCH_FORCE_DIST") == "1"
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not launched and (args.gpus > 1 or force_dist):
        self_launch(args)                       # does not return
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    # Every N times the SAME per-GPU workload: the single-GPU configurations BASELINE.json quotes the metric
    # on (configs[1] / configs[2]) -- N ranks = N x 500 templates (N x 50 000 sources) against a replicated
    # day, so that value(N) / (N x value(1)) reads as weak-scaling efficiency.  (Rounds 1-3 switched to the
    # per-GPU shares of configs[3] / [4] for N > 1: a 120-channel CC-sample costs twice a 60-channel one, and
    # a driver dividing the two values read 0.5 at perfect scaling.)  The shares of the 8-GPU configurations
    # are timed as untimed-extra `shares` of the line for N > 1 (--skip-shares turns them off).
    if args.mf_config is None:
        args.mf_config = "cfg2"
    if args.bp_config is None:
        args.bp_config = "cfg3"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    # BPMF_BENCH_FORCE_DIST=1 initialises RCCL even for one rank (exercises the N > 1 code path on
    # a single-GPU box)
    ranks_info = None
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
        # who runs where: (rank, local device index, device name, PCI bus id) of every rank, and one
        # all-reduce over RCCL that only comes out right when every rank took part
        props = torch.cuda.get_device_properties(device)
        mine = {"rank": rank, "device": local_rank, "name": props.name,
                "pci_bus_id": getattr(props, "pci_bus_id", None), "pid": os.getpid()}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ones = torch.ones(1, dtype=torch.int64, device=device)
        dist.all_reduce(ones)
        ranks_info = {"rccl_ranks": int(ones.item()), "backend": dist.get_backend(), "ranks": gathered,
                      "self_launched": os.environ.get("BPMF_BENCH_SELF_LAUNCHED") == "1"}
        if ranks_info["rccl_ranks"] != world:
            raise SystemExit(f"RCCL all-reduce saw {ranks_info['rccl_ranks']} ranks, expected {world}")
        if len({(g["device"]) for g in gathered}) != world:
            raise SystemExit(f"two ranks share a device: {gathered}")

    import seismic_bpmf_amd as sb
    from seismic_bpmf_amd import _lib, synthetic as syn

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    own_dt = []

    def timed(step_fn, steps, warmup):
        for _ in range(warmup):
            step_fn()
        barrier()
        _lib.profile_enable(True)          # clears the launch log
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        barrier()
        dt = time.perf_counter() - t0
        _lib.profile_enable(False)
        own_dt.append(dt)                  # this rank's own clock (the line carries every rank's, `ranks`)
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    # ---------------------------------------------------------------- matched filter
    cfg = dict(syn.MF_CONFIGS[args.mf_config])
    if args.templates:
        cfg["T"] = args.templates
    T, S, C, L, N = cfg["T"], cfg["S"], cfg["C"], cfg["L"], cfg["N"]
    n_corr = N - L + 1
    tmpl, mv, w, data, planted = mf_inputs_device(cfg, device, 20260928, rank)
    mf = sb.MatchedFilterGPU(device=local_rank)
    mf.set_data(data)
    cc = torch.empty((T, n_corr), dtype=torch.float32, device=device)

    def mf_step():
        mf._prepared_for = None            # a step includes the per-day data preparation
        mf.run(tmpl, mv, w, 1, out=cc)

    mf_dt = timed(mf_step, args.steps, args.warmup)
    mf_own_dt = own_dt[-1]
    mf_kernel_ms = _lib.profile_times_ms(_lib.KERNEL_MF_MAIN)
    mf_value = world * T * n_corr * args.steps / mf_dt / 1e6
    flop_per_launch = 2.0 * L * S * C * T * n_corr          # direct-form, all channels weighted
    k_ms = float(np.mean(mf_kernel_ms)) if mf_kernel_ms else float("nan")
    achieved = flop_per_launch / (k_ms * 1e-3) / 1e12
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "mf_main_pmc.json")
    if os.path.exists(pmc_file) and args.mf_config == "cfg2" and not args.templates:
        try:  # PMC passes are separate rocprofv3 runs of this same launch (profiles/)
            traffic = json.load(open(pmc_file)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"kernel": "mf_mfma_wave_kernel", "bound": "mfma", "achieved": round(achieved, 2),
                "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP32_PEAK_TFLOPS, 4),
                "traffic": traffic,
                "traffic_source": ("profiles/mf_main_pmc.json: separate rocprofv3 --pmc passes over this launch, "
                                   "committed, NOT measured in this run") if traffic is not None else None,
                "avg_launch_ms": round(k_ms, 3), "launches": len(mf_kernel_ms),
                "algorithmic": "2*L*S*C flop per network-CC-sample x T*n_corr samples per launch",
                "frac_note": ("peak = 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz (nominal).  Cycle counters in the kernel "
                              "(profiles/r05_mf_phase_cycles.txt, tools/phase/, NOT measured in this run): at L = 256 the matrix "
                              "pipe is busy 95 % of the counted cycles and the chip sustains ~2.3 GHz under this instruction mix; "
                              "L / (L + 16) = 0.941 of the issued MFMA flops are direct-form flops: 0.941 x 0.96 x 0.95 = 0.86"),
                "hbm_frac_informational": round(
                    4.0 * (S * C * N + T * S * C * (L + 2) + T * n_corr) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    rank_stats = {"rank": rank, "mf_ms_per_step": round(mf_own_dt / args.steps * 1e3, 3), "mf_kernel_ms": round(k_ms, 3),
                  "mf_frac": round(achieved / FP32_PEAK_TFLOPS, 4)}
    peak = float(cc[0].max().item())
    detect = detection_stage(cc, planted, mv, w, local_rank, dist, device, t_offset=rank * T)
    # untimed extra: the same step under option mf.compat_sqrt_norm (cc = num / sqrtf(E_t * E_d) in the
    # epilogue of the MFMA kernel: an IEEE square root and divide per channel and lag)
    compat = None
    if rank == 0 and world == 1 and dist is None and not args.skip_e2e:
        with _lib.options(**{"mf.compat_sqrt_norm": 1}):
            mf_step()
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            t0 = time.perf_counter()
            mf_step()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            _lib.profile_enable(False)
            kms = float(np.mean(_lib.profile_times_ms(_lib.KERNEL_MF_MAIN)))
        mf._prepared_for = None
        compat = {"mf.compat_sqrt_norm": {"ms_per_step": round(wall * 1e3, 2), "kernel_ms": round(kms, 2),
                                          "kernel_ms_default": round(k_ms, 2),
                                          "slowdown": round(kms / k_ms, 4),
                                          "frac_of_fp32_peak": round(flop_per_launch / (kms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)}}
    # untimed extra (round 6): the same step under option mf.split16 -- numerators from fp16 hi/lo splits of data and
    # templates, three v_mfma_f32_32x32x16_f16 products, fp32 accumulation (csrc/mf_split.h).  Never `value`: the
    # headline stays the exact-fp32 kernel, bit-identical to the oracle.  Reported: the step and kernel time, the
    # direct-form rate, and what the option costs in accuracy MEASURED on this very matrix -- max |cc - exact| / sum|w|
    # over all T x n_corr CC sums (bar 2e-5: SURVEY App. C MF-5) and whether every planted event is still detected
    # at exactly its index by the product's detection stage.
    split16 = None
    if rank == 0 and world == 1 and dist is None and not args.skip_e2e:
        try:
            cc2 = torch.empty_like(cc)
            with _lib.options(**{"mf.split16": 1}):
                def split_step():
                    mf._prepared_for = None
                    mf.run(tmpl, mv, w, 1, out=cc2)
                split_step()
                torch.cuda.synchronize()
                _lib.profile_enable(True)
                t0 = time.perf_counter()
                for _ in range(3):
                    split_step()
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / 3
                _lib.profile_enable(False)
                kms = float(np.mean(_lib.profile_times_ms(_lib.KERNEL_MF_MAIN)))
            mf._prepared_for = None
            sw = float(w.abs().reshape(T, -1).sum(dim=1).max().item())
            worst = 0.0
            for t0_ in range(0, T, 25):       # (row blocks: no third 17 GB array)
                worst = max(worst, float((cc2[t0_:t0_ + 25] - cc[t0_:t0_ + 25]).abs().max().item()))
            same_zero = bool(((cc2 == 0) == (cc == 0)).all().item())
            det2 = detection_stage(cc2, planted, mv, w, local_rank, None, device)
            split16 = {"option": "mf.split16 = 1 (off by default)", "dtype": "f16x3->f32",
                       "ms_per_step": round(wall * 1e3, 2), "kernel_ms": round(kms, 2), "kernel_ms_exact_fp32": round(k_ms, 2),
                       "speedup_kernel": round(k_ms / kms, 3),
                       "value": round(T * n_corr / wall / 1e6, 1), "unit": "M network-CC-samples/s",
                       "direct_form_tflops": round(flop_per_launch / (kms * 1e-3) / 1e12, 1),
                       "x_fp32_mfma_peak": round(flop_per_launch / (kms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 3),
                       "f16_mfma_issue_frac_of_2500_tflops": round(
                           3.0 * 2.0 * (16 * ((L + 38 + 15) // 16)) * S * C * T * n_corr / (kms * 1e-3) / 1e12 / 2500.0, 4),
                       # the same contract as the headline's: algorithmic (direct-form) flops per launch / the kernel's own launch time,
                       # against the dense fp16 MFMA peak divided by the three products a direct-form multiply-add costs on this path
                       "roofline": {"kernel": "mf_split_kernel", "bound": "mfma", "achieved": round(flop_per_launch / (kms * 1e-3) / 1e12, 2),
                                    "peak": round(2500.0 / 3.0, 1), "unit": "TFLOP/s", "frac": round(flop_per_launch / (kms * 1e-3) / 1e12 / (2500.0 / 3.0), 4),
                                    "traffic": None, "avg_launch_ms": round(kms, 3),
                                    "peak_is": "2.5 PFLOP/s dense fp16 MFMA (MI355X_MICROARCH.md) / 3 products per direct-form flop; the Toeplitz band "
                                               "(304 deep for 256 samples) is inside `achieved`, as in the headline; traffic: profiles/r06_mf_split16_pmc.txt "
                                               "(41.5 GB fetched per 100 templates, not measured in this run)"},
                       "max_abs_diff_vs_exact_over_sum_w": worst / sw, "tolerance": 2e-5,
                       "exact_zeros_identical": same_zero,
                       "planted": det2["planted"], "planted_found_at_exact_index": det2["planted_found_at_exact_index"],
                       "detections": det2["detections"], "detections_exact_fp32": detect["detections"],
                       "note": ("kernel_ms includes the band images of the template batch; the per-day split of the data is in "
                                "ms_per_step (data preparation is part of a step).  The path is power-bound like the fp32 kernel: "
                                "tools/ubench/mfma_split16.hip, profiles/r06_mf_split16.txt")}
            del cc2
        except Exception as e:                      # (an extra must not cost the line)
            split16 = {"failed": str(e)}
    del cc, mf
    torch.cuda.empty_cache()
    # End to end through the host-pointer entry point (what the reference's call site sees:
    # NumPy in, NumPy out): H2D of the day, kernels, D2H of the API-mandated (T, n_corr) matrix.
    # Untimed extra, N = 1 only; two calls, the faster one is reported (each call allocates and
    # first-touches its own 17 GB result array).  Never `value`.
    e2e = None
    if world == 1 and dist is None and not args.skip_e2e and T * n_corr * 4 < 40e9:
        h_t, h_mv, h_w, h_d = (x.cpu().numpy() for x in (tmpl, mv, w, data))
        e2e_ms, mf_thr = [], []
        h_cc = None
        for _ in range(2):
            del h_cc                       # one 17 GB result array at a time
            h_new = h_d.copy()             # a new day is a NEW array: host memory the runtime has not page-locked before
            th0 = cgroup_throttle()
            t0 = time.perf_counter()
            h_cc = sb.matched_filter(h_t, h_mv, h_w, h_new, 1, arch="gpu", check_zeros=False,
                                     device=[local_rank])
            e2e_ms.append((time.perf_counter() - t0) * 1e3)
            mf_stats = _lib.host_call_stats()
            mf_thr.append(throttle_delta(th0, cgroup_throttle()))
            del h_new
        e2e = {"mf_ms": round(min(e2e_ms), 1), "mf_calls_ms": [round(x, 1) for x in e2e_ms],
               "mf_value": round(T * n_corr / (min(e2e_ms) * 1e-3) / 1e6, 1), "unit": "M CC-samples/s",
               "moves": f"H2D {h_d.nbytes / 1e9:.2f} GB data + templates, D2H {h_cc.nbytes / 1e9:.2f} GB cc_sums "
                        "(pageable host memory, a fresh copy of the day per call; pinned staging both ways inside bpmf_mf_run, "
                        "the day arriving in pieces while the first two template batches run)",
               "breakdown_of_last_call": {k: (round(v, 2) if isinstance(v, float) else v) for k, v in mf_stats.items()},
               "cgroup_throttled_by_call": mf_thr,
               "row0_peak_cc": round(float(h_cc[0].max()), 4)}
        # the same drop-in call under option mf.split16 (untimed extra of the untimed extra: one call, the day uploaded in one
        # piece -- a channel's scale is its maximum over the whole day -- then the split kernel; never `value`)
        try:
            h_cc = None                      # (one 17 GB result array at a time)
            with _lib.options(**{"mf.split16": 1}):
                h_new = h_d.copy()
                t0 = time.perf_counter()
                h_cc = sb.matched_filter(h_t, h_mv, h_w, h_new, 1, arch="gpu", check_zeros=False, device=[local_rank])
                e2e["mf_split16_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
                e2e["mf_split16_breakdown"] = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in _lib.host_call_stats().items()}
                e2e["mf_split16_row0_peak_cc"] = round(float(h_cc[0].max()), 4)
                del h_new
        except Exception as e:                      # (an extra must not cost the line)
            e2e["mf_split16_ms"] = f"failed: {e}"
        del h_cc, h_t, h_mv, h_w, h_d

    # untimed extras: the matched filter off its headline shape -- BASELINE
    # configs[0] (the reference's own CPU-runnable case: one hour, 4 templates, L = 128) and a day with
    # 128-sample templates; kernel time from the library's events, the whole call from the host clock
    mf_shapes = None
    if rank == 0 and world == 1 and dist is None and not args.skip_e2e:
        mf_shapes = {}
        # ("tutorial": the shape of the reference's own tutorial, /root/reference/tutorial/notebooks/BPMF_parameters.cfg:7-16 --
        # 10 templates of 8 s at 25 Hz on 8 stations x 3 components, one day at 25 Hz)
        for name, (T2, S2, C2, L2, N2) in (("configs0", (4, 8, 3, 128, 180_000)), ("day_L128", (50, 20, 3, 128, 8_640_000)),
                                           ("tutorial", (10, 8, 3, 200, 2_160_000))):
            g2 = torch.Generator(device=device)
            g2.manual_seed(77)
            d2 = torch.randn((S2, C2, N2), device=device, generator=g2)
            t2 = torch.randn((T2, S2, C2, L2), device=device, generator=g2)
            m2 = torch.randint(0, 1500, (T2, S2, C2), device=device, dtype=torch.int32, generator=g2)
            w2 = torch.full((T2, S2, C2), 1.0 / (S2 * C2), device=device)
            mf2 = sb.MatchedFilterGPU(device=local_rank)
            mf2.set_data(d2)
            o2 = mf2.run(t2, m2, w2, 1)
            torch.cuda.synchronize()
            reps = 50 if N2 < 1_000_000 else 3
            # the whole call from the host clock over back-to-back resident calls (best of 3 rounds), then the
            # kernel from the library's events in a round of its own (two event records per call are not part
            # of a call, and at 60 us per call they show)
            wall = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(reps):
                    mf2.run(t2, m2, w2, 1, out=o2)
                torch.cuda.synchronize()
                wall = min(wall, (time.perf_counter() - t0) / reps)
            _lib.profile_enable(True)
            for _ in range(reps):
                mf2.run(t2, m2, w2, 1, out=o2)
            torch.cuda.synchronize()
            _lib.profile_enable(False)
            kms = float(np.mean(_lib.profile_times_ms(_lib.KERNEL_MF_MAIN)))
            flop = 2.0 * L2 * S2 * C2 * T2 * (N2 - L2 + 1)
            mf_shapes[name] = {"workload": f"{T2} templates x {S2} stations x {C2} comp, L={L2}, N={N2}, step 1 (data resident and prepared)",
                               "ms_per_call": round(wall * 1e3, 4), "kernel_ms": round(kms, 4),
                               "value": round(T2 * (N2 - L2 + 1) / wall / 1e6, 1), "unit": "M CC-samples/s",
                               "roofline": {"kernel": "mf_mfma_wave_kernel", "bound": "mfma",
                                            "achieved": round(flop / (kms * 1e-3) / 1e12, 2), "peak": FP32_PEAK_TFLOPS,
                                            "unit": "TFLOP/s", "frac": round(flop / (kms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                                            "frac_whole_call": round(flop / wall / 1e12 / FP32_PEAK_TFLOPS, 4)}}
            # the same shape under the opt-in split-precision path (option mf.split16; weights sum to 1 per template)
            _lib.set_option("mf.split16", 1)
            try:
                mf3 = sb.MatchedFilterGPU(device=local_rank)
                mf3.set_data(d2)
                o3 = mf3.run(t2, m2, w2, 1)
                torch.cuda.synchronize()
                _lib.profile_enable(True)
                for _ in range(reps):
                    mf3.run(t2, m2, w2, 1, out=o3)
                torch.cuda.synchronize()
                _lib.profile_enable(False)
                kms3 = float(np.mean(_lib.profile_times_ms(_lib.KERNEL_MF_MAIN)))
                pairs = T2 * (-(-(N2 - L2 + 1) // 8192))
                mf_shapes[name]["split16"] = {"kernel_ms": round(kms3, 4), "speedup_kernel": round(kms / kms3, 3),
                                              "template_x_8192_lag_block_pairs": pairs,
                                              "kernel": "mf_split_kernel" if pairs >= 128 else
                                              "the exact kernel: mf.split16 = 1 leaves launches below 128 pairs to it (latency-bound there, tools/probe_split_small.py)",
                                              "max_abs_diff_vs_exact_over_sum_w": float((o3 - o2).abs().max().item()),
                                              "tolerance": 2e-5, "dtype": "f16x3->f32"}
                del mf3, o3
            finally:
                _lib.set_option("mf.split16", 0)
            if name == "tutorial":
                # the SAME call as a BPMF user makes it (nb8 cell 18: MatchedFilter.compute_cc_time_series ->
                # fmf.matched_filter, BPMF/similarity_search.py:526-533): through the import shim, NumPy in / out.
                # BASELINE.md's one published figure -- 4.30 s on 24 CPU threads -- is for exactly this call
                # (context, not a target: other hardware)
                sys.path.insert(0, os.path.join(ROOT, "shims"))
                try:
                    import fast_matched_filter as fmf_shim
                    h = [x.cpu().numpy() for x in (t2, m2, w2, d2)]
                    want = o2.cpu().numpy()
                    calls, st = [], None
                    for _ in range(4):
                        dn = h[3].copy()
                        t0 = time.perf_counter()
                        got = fmf_shim.matched_filter(h[0], h[1], h[2], dn, 1, arch="gpu", check_zeros=False)
                        calls.append((time.perf_counter() - t0) * 1e3)
                        st = _lib.host_call_stats()
                        del dn
                    later = sorted(calls[1:])
                    mf_shapes["tutorial_shim"] = {
                        "workload": f"fast_matched_filter.matched_filter (shims/) NumPy in / out: {T2} x {S2} x {C2}, L={L2}, N={N2}; "
                                    f"H2D {h[3].nbytes / 1e6:.0f} MB, D2H {got.nbytes / 1e6:.0f} MB",
                        "ms": round(later[len(later) // 2], 2), "calls_ms": [round(x, 2) for x in calls],
                        "kernel_ms_resident": round(kms, 3), "resident_call_ms": round(wall * 1e3, 3),
                        "value": round(T2 * (N2 - L2 + 1) / (later[len(later) // 2] * 1e-3) / 1e6, 1), "unit": "M CC-samples/s",
                        "breakdown_of_last_call": {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()},
                        "difference_is": "the day's way up (PCIe, through the pinned pieces), the per-day preparation (prefix sums, norms: "
                                         "a resident engine does it once per day, this call every time), the CC matrix's way down, Python",
                        "equals_resident_result": bool(np.array_equal(got, want)),
                        "published_reference": "4.30 s for this call on 24 CPU threads (tutorial nb8 cell 18; BASELINE.md)"}
                    del h, got, want
                finally:
                    sys.path.remove(os.path.join(ROOT, "shims"))
            del d2, t2, m2, w2, o2, mf2
        torch.cuda.empty_cache()

    # ---------------------------------------------------------------- backprojection
    bp_obj = None
    if not args.skip_bp:
        bcfg = dict(syn.BP_CONFIGS[args.bp_config])
        geo, feat, wp = bp_inputs(bcfg, device, 20260928, rank, world)
        K_all = geo["moveouts"].shape[0]
        # weak scaling: every rank owns its own tile of the (world x larger) grid, global source ids
        bf = sb.BeamformerGPU(geo["moveouts"], geo["weights_sources"], device=local_rank,
                              source_id_offset=rank * K_all)
        Nb = bcfg["N"]
        beam = torch.empty(Nb, dtype=torch.float32, device=device)
        arg = torch.empty(Nb, dtype=torch.int32, device=device)

        def bp_step():
            bf.run(feat, wp, "max", "strict", out=(beam, arg))
            if dist is not None:            # the path's one real exchange step
                packed = bf.pack_max(beam, arg)
                dist.all_reduce(packed, op=dist.ReduceOp.MAX)
                bf.unpack_max(packed)

        bp_dt = timed(bp_step, args.steps, args.warmup)
        bp_own_dt = own_dt[-1]
        bp_ms = _lib.profile_times_ms(_lib.KERNEL_BP_BEAM)
        s_act = float((geo["weights_sources"] != 0).sum(axis=1).mean())
        bk = float(np.mean(bp_ms)) if bp_ms else float("nan")
        gather_tbs = 4.0 * s_act * bcfg["P"] * K_all * Nb / (bk * 1e-3) / 1e12
        pinfo = bf.plan_info()
        # LDS rate of the gather instruction the plan dispatches to (MI355X_MICROARCH.md, LDS table):
        # ds_read_b64 256 B/clk/CU on dual windows, 4-byte gathers 128 B/clk/CU
        lds_peak = LDS_B32_PEAK_TBS * (2.0 if pinfo["gather_bytes"] == 8 else 1.0)
        bp_traffic = None
        bp_pmc = os.path.join(ROOT, "profiles", "bp_beam_pmc.json")
        if os.path.exists(bp_pmc) and args.bp_config == "cfg3":
            try:  # separate rocprofv3 --pmc passes over this same launch (tools/pmc_bp_traffic.sh)
                bp_traffic = json.load(open(bp_pmc)).get("hbm_bytes_per_launch")
            except Exception:
                bp_traffic = None
        bp_obj = {"metric": "grid-points x samples / s", "value": world * K_all * Nb * args.steps / bp_dt,
                  "ms_per_step": round(bp_dt / args.steps * 1e3, 3),
                  "config": {"workload": f"BASELINE {BP_LABEL.get(args.bp_config, args.bp_config)}: {K_all} sources x {bcfg['S']} stations x "
                                         f"{bcfg['C']} comp x {bcfg['P']} phases, N={Nb} (1 day @ {bcfg['sr']:g} Hz), "
                                         f"{s_act:.1f} active stations/source, reduce=max, strict",
                             "parallelism": (f"source grid tiled x{world}: every rank scans {K_all} sources of a {world * K_all}-source "
                                             "grid with global ids, features replicated; one packed-key all-reduce(MAX) of 8 B per "
                                             "time sample over RCCL per step" if world > 1 else "single GPU")},
                  "roofline": {"kernel": "bp_beam_fast_kernel (+ bp_beam_wps2_kernel on the edge tiles of the day)"
                                         if pinfo["gather_bytes"] == 8 else "bp_beam_wps2_kernel",
                               "bound": "lds-gather", "achieved": round(gather_tbs, 2),
                               "peak": round(lds_peak, 1), "unit": "TB/s",
                               "frac": round(gather_tbs / lds_peak, 4),
                               "frac_of_4byte_gather_rate": round(gather_tbs / LDS_B32_PEAK_TBS, 4),
                               "traffic": bp_traffic,
                               "traffic_source": ("profiles/bp_beam_pmc.json: separate rocprofv3 --pmc passes over this "
                                                  "launch, committed, NOT measured in this run") if bp_traffic is not None else None,
                               "frac_note": ("peak = 256 CU x 256 B/clk x 2.4 GHz (nominal).  Under this kernel the chip sustains "
                                             "1.9 - 2.1 GHz depending on the box (cycle counters of the instrumented build, "
                                             "profiles/r05_bp_fast_phase_cycles.txt and ..._slow_box.txt, NOT measured in this run: the "
                                             "same 181 5xx cycles per group and wave on a 143 ms box and on a 157 ms box): 0.89 of the "
                                             "ds_read_b64 rate at the sustained clock on either"),
                               "plan": pinfo, "avg_launch_ms": round(bk, 3),
                               "algorithmic": "4*S_active*P gathered bytes per grid-point x sample",
                               "fp32_frac": round(2.0 * s_act * bcfg["P"] * K_all * Nb / (bk * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)}}
        rank_stats.update({"bp_ms_per_step": round(bp_own_dt / args.steps * 1e3, 3), "bp_kernel_ms": round(bk, 3),
                           "bp_frac": round(gather_tbs / lds_peak, 4)})
        if rank == 0:
            bp_obj["detection"] = bp_detection_stage(beam, arg, geo, bcfg)
        if rank == 0 and world == 1:
            # untimed extra: the path's second caller, the event relocation (BPMF/dataset.py:2174-2216):
            # the whole grid over a series of 3 000 samples, full beam volume (reduce="none")
            n_short = 3000
            short = feat[:, :, :n_short].contiguous()
            vol = bf.run(short, wp, "none", "strict")
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                vol = bf.run(short, wp, "none", "strict", out=vol)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            bp_obj["relocation"] = {"workload": f"{K_all} sources x {n_short} samples, reduce=none (the beam volume stays in HBM)",
                                    "ms": round(min(ts), 3), "output_gb_per_s": round(K_all * n_short * 4 / (min(ts) * 1e-3) / 1e9, 1)}
            del vol, short
        bf.close()
        if world == 1 and dist is None and not args.skip_e2e:
            h_f, h_wp = feat.cpu().numpy(), wp.cpu().numpy()
            ms, stats, bp_thr = [], [], []
            for _ in range(4):           # the first call builds the plan; the later ones find it in the library's cache
                h_new = h_f.copy()       # a new day is a NEW array (host memory the runtime has not page-locked before)
                th0 = cgroup_throttle()
                t0 = time.perf_counter()
                hb, ha = sb.beamform(h_new, geo["moveouts"], h_wp, geo["weights_sources"], device="gpu",
                                     reduce="max", out_of_bounds="strict", device_id=[local_rank])
                ms.append((time.perf_counter() - t0) * 1e3)
                stats.append(_lib.host_call_stats())
                bp_thr.append(throttle_delta(th0, cgroup_throttle()))
                del h_new
            later = sorted(ms[1:])
            med = later[len(later) // 2]
            rep = 1 + ms[1:].index(med)
            bp_obj["end_to_end"] = {"ms": round(med, 1), "first_call_ms": round(ms[0], 1), "calls_ms": [round(x, 1) for x in ms],
                                    "ms_is": "the median of the calls after the first (BPMF calls once per day in a long-running process)",
                                    "resident_ms_per_step": round(bp_dt / args.steps * 1e3, 1),
                                    "value": K_all * Nb / (med * 1e-3),
                                    # where the reported call's time went, by the library's own account (bpmf_host_call_stats)
                                    "breakdown_of_reported_call": {k: (round(v, 2) if isinstance(v, float) else v) for k, v in stats[rep].items()},
                                    "pinned_wait_ms_by_call": [round(st.get("pinned_wait_ms", 0.0), 1) for st in stats],
                                    # periods in which the container's CPU quota ran out during each call (every thread of the
                                    # group frozen until the next period: the host side of a call stands still, profiles/r06_bp_e2e.txt)
                                    "cgroup_throttled_by_call": bp_thr,
                                    "breakdown_note": ("first_kernel_start_ms: entry -> first kernel enqueued; host_copy_ms: the copy pool filling the "
                                                       "pinned pieces (overlaps earlier pieces' kernels); pinned_wait_ms: blocked until a piece's previous "
                                                       "H2D had completed (~10 ms over the 17 pieces of a cfg3 day; rounds 4-5 lost 20-40 ms here in the call behind a plan build: "
                                                       "the plan's tables went up from pageable vectors, profiles/r06_bp_e2e.txt); device_wait_ms: last launch enqueued -> results "
                                                       "in the caller's arrays"),
                                    "host": {"loadavg_1min": round(os.getloadavg()[0], 1), "fill_threads": stats[rep].get("fill_threads")},
                                    "moves": f"H2D {h_f.nbytes / 1e9:.2f} GB features, D2H {(hb.nbytes + ha.nbytes) / 1e6:.0f} MB "
                                             "maxbeam + argmax, both through the pinned pieces; first call also builds the plan",
                                    "equals_resident_result": bool(np.array_equal(hb, beam.cpu().numpy()) and
                                                                   np.array_equal(ha, arg.cpu().numpy()))}
            del h_f, hb, ha
        if rank == 0 and world == 1 and dist is None and not args.skip_e2e:
            # untimed extra: the tutorial's own backprojection (nb5 cell 33: bf.backproject -> beampower.beamform,
            # BPMF/template_search.py:549-558): 35 490 sources (nb4 cell 32) x one day at 25 Hz x 8 stations x 2 phases
            # (PhaseNet features: 2 channels, one per phase; all 8 stations weighted), reduce="max" -- resident, and
            # through the import shim with NumPy in / out and int64 moveouts as BPMF delivers them
            try:
                tg = syn.make_bp_geometry((39, 35, 26), 8, 2, 25.0, n_closest=8)
                Nt = 2_160_000
                ft = torch.randn((8, 2, Nt), device=device, generator=torch.Generator(device=device).manual_seed(11)).abs_()
                wpt = syn.phase_weights(8, 2, 2)
                bft = sb.BeamformerGPU(tg["moveouts"], tg["weights_sources"], device=local_rank)
                wpt_d = torch.as_tensor(wpt, device=device)
                tb, ta = bft.run(ft, wpt_d, "max", "strict")
                torch.cuda.synchronize()
                _lib.profile_enable(True)
                t0 = time.perf_counter()
                for _ in range(3):
                    bft.run(ft, wpt_d, "max", "strict", out=(tb, ta))
                torch.cuda.synchronize()
                t_res = (time.perf_counter() - t0) / 3 * 1e3
                _lib.profile_enable(False)
                tk = float(np.mean(_lib.profile_times_ms(_lib.KERNEL_BP_BEAM)))
                tinfo = bft.plan_info()
                Kt = tg["moveouts"].shape[0]
                gb = 4.0 * 8 * 2 * Kt * Nt / (tk * 1e-3) / 1e12
                tpeak = LDS_B32_PEAK_TBS * (2.0 if tinfo["gather_bytes"] == 8 else 1.0)
                sys.path.insert(0, os.path.join(ROOT, "shims"))
                try:
                    import beampower as bp_shim
                    hft, mv64 = ft.cpu().numpy(), tg["moveouts"].astype(np.int64)
                    calls, st = [], None
                    for _ in range(4):
                        hn = hft.copy()
                        t0 = time.perf_counter()
                        sb_, sa_ = bp_shim.beampower.beamform(hn, mv64, wpt, tg["weights_sources"], device="gpu", reduce="max")
                        calls.append((time.perf_counter() - t0) * 1e3)
                        st = _lib.host_call_stats()
                        del hn
                    later = sorted(calls[1:])
                    same = bool(np.array_equal(sb_, tb.cpu().numpy()) and np.array_equal(sa_, ta.cpu().numpy()))
                finally:
                    sys.path.remove(os.path.join(ROOT, "shims"))
                bp_obj["tutorial"] = {
                    "workload": f"{Kt} sources x {Nt} samples (1 day @ 25 Hz) x 8 stations x 2 phases, 2 channels, all stations weighted, reduce=max",
                    "resident_ms": round(t_res, 2), "kernel_ms": round(tk, 2),
                    "value": Kt * Nt / (t_res * 1e-3), "unit": "grid-points x samples / s",
                    "roofline": {"bound": "lds-gather", "achieved": round(gb, 2), "peak": round(tpeak, 1), "unit": "TB/s",
                                 "frac": round(gb / tpeak, 4)},
                    "shim": {"through": "beampower.beampower.beamform (shims/), NumPy in / out, int64 moveouts",
                             "ms": round(later[len(later) // 2], 2), "calls_ms": [round(x, 2) for x in calls],
                             "breakdown_of_last_call": {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()},
                             "difference_is": f"H2D {hft.nbytes / 1e6:.0f} MB of features in pieces beside the kernels, D2H {(sb_.nbytes + sa_.nbytes) / 1e6:.0f} MB, the "
                                              "int64 -> int32 cast and comparison of the 4.5 MB moveout table (the plan is found in the library's cache "
                                              "after the first call, which builds it), Python",
                             "equals_resident_result": same},
                    "published_reference": "none: nb5 cell 33 prints no timing (BASELINE.md)"}
                bft.close()
                del ft, tb, ta, hft
                torch.cuda.empty_cache()
            except Exception as e:                      # (an extra must not cost the line)
                bp_obj["tutorial"] = {"failed": str(e)}
            # untimed extra: the stage in FRONT of the beamformer in the vanilla workflow -- saturated envelopes of
            # the day (BPMF/template_search.py:1525-1617: analytic signal, per-channel median / MAD, standardise,
            # clip) on configs[2]'s 60 channels x 4.32 M samples, resident in HBM
            from seismic_bpmf_amd import features as ft
            raw = torch.randn((bcfg["S"], bcfg["C"], Nb), device=device, generator=torch.Generator(device=device).manual_seed(9))
            ft.saturated_envelopes(raw, device=local_rank)
            torch.cuda.synchronize()
            t_env = []
            for stage in (lambda: ft.envelope(raw, device=local_rank), lambda: ft.saturated_envelopes(raw, device=local_rank)):
                best = float("inf")
                for _ in range(3):
                    t0 = time.perf_counter()
                    stage()
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) * 1e3)
                t_env.append(best)
            bp_obj["features"] = {"workload": f"saturated_envelopes of {bcfg['S']} x {bcfg['C']} channels x {Nb} samples (float32 in, float32 out)",
                                  "envelope_ms": round(t_env[0], 2), "saturated_envelopes_ms": round(t_env[1], 2),
                                  "fraction_of_a_bp_step": round(t_env[1] / (bp_dt / args.steps * 1e3), 3),
                                  "how": "Hilbert transform by a float64 real-input FFT pair (hipFFT behind torch.fft), median / MAD of the valid samples in two reads of the channels by the whole chip (csrc/stats.hip rm_*)"}
            del raw
            torch.cuda.empty_cache()
        if rank == 0 and world == 1 and not args.skip_dense:
            # untimed extras: DENSE station weights -- BASELINE's literal "x 20 / x 40 stations", what
            # _weights_sources_closest returns for num_closest_stations >= n_stations
            # (BPMF/template_search.py:779-798): every station of every source weighted.  Round 3: these
            # run the 8-byte-gather kernel on tiles of 256 / 128 samples (csrc/bp_fast.hip).
            del beam, arg
            bp_obj["dense"] = {}
            for name, reps in (("cfg3", 3), ("cfg5_per_gpu", 1)):
                dcfg = dict(syn.BP_CONFIGS[name])
                if name == args.bp_config:
                    dgeo, dfeat, dwp = geo, feat, wp
                else:
                    del feat
                    torch.cuda.empty_cache()
                    dgeo, dfeat, dwp = bp_inputs(dcfg, device, 20260928, 0, 1)
                    feat = dfeat
                ws_dense = np.full(dgeo["weights_sources"].shape, 1.0 / dcfg["S"], dtype=np.float32)
                t0 = time.perf_counter()
                dbf = sb.BeamformerGPU(dgeo["moveouts"], ws_dense, device=local_rank)
                plan_s = time.perf_counter() - t0
                dbeam, darg = dbf.run(dfeat, dwp, "max", "strict")
                torch.cuda.synchronize()
                _lib.profile_enable(True)
                for _ in range(reps):
                    dbf.run(dfeat, dwp, "max", "strict", out=(dbeam, darg))
                torch.cuda.synchronize()
                _lib.profile_enable(False)
                dms = float(np.mean(_lib.profile_times_ms(_lib.KERNEL_BP_BEAM)))
                dK = dgeo["moveouts"].shape[0]
                dtbs = 4.0 * dcfg["S"] * dcfg["P"] * dK * dcfg["N"] / (dms * 1e-3) / 1e12
                dinfo = dbf.plan_info()
                dpeak = LDS_B32_PEAK_TBS * (2.0 if dinfo["gather_bytes"] == 8 else 1.0)
                bp_obj["dense"][name] = {
                    "workload": f"{dK} sources x {dcfg['S']} stations (ALL weighted) x {dcfg['P']} phases, N={dcfg['N']}, "
                                "reduce=max, strict",
                    "ms": round(dms, 2), "value": dK * dcfg["N"] / (dms * 1e-3), "unit": "grid-points x samples / s",
                    "launches": reps, "plan_seconds": round(plan_s, 3),
                    "roofline": {"kernel": f"bp_beam_fast_kernel<tile {dinfo['tile']}>" if dinfo["n_classes"] else "general kernels",
                                 "bound": "lds-gather", "achieved": round(dtbs, 2), "peak": round(dpeak, 1), "unit": "TB/s",
                                 "frac": round(dtbs / dpeak, 4), "plan": dinfo,
                                 "algorithmic": "4*S*P gathered bytes per grid-point x sample"}}
                # Result check (the dense classes run only here and in tests/test_gpu_shares.py): one window of
                # the day recomputed by an independent device path -- the global-memory kernels of bp_direct.hip
                # on compact term lists (option bp.direct), bit-exact against the oracle in the GPU suite -- must
                # equal the same samples of the full-day result bit for bit.
                W, tmax_d = 2048, int(dgeo["moveouts"].max())
                i0 = dcfg["N"] // 3
                with _lib.options(**{"bp.direct": 1}):
                    ref_bf = sb.BeamformerGPU(dgeo["moveouts"], ws_dense, device=local_rank)
                    rb, ra = ref_bf.run(dfeat[:, :, i0:i0 + W + tmax_d + 1].contiguous(), dwp, "max", "strict")
                    torch.cuda.synchronize()
                    ref_bf.close()
                same = bool(torch.equal(rb[:W], dbeam[i0:i0 + W]) and torch.equal(ra[:W], darg[i0:i0 + W]))
                bp_obj["dense"][name]["checked"] = {
                    "against": f"bp_direct.hip (global-memory gathers, option bp.direct) on samples [{i0}, {i0 + W}) x all {dK} sources",
                    "bit_identical": same}
                if not same:
                    raise SystemExit(f"bench: dense BP result of {name} differs from the bp_direct reference window")
                dbf.close()
                del dbeam, darg, rb, ra
        if rank == 0 and dist is None and not args.skip_cpu:
            bp_obj["cpu_baseline"] = cpu_baseline_bp(bcfg, geo, max(2.0, args.cpu_seconds / 3))

    # ---------------------------------------------------------------- N > 1: shares of configs[3] / [4]
    # Untimed extras (never `value`): what ONE GPU of the 8-GPU configurations computes -- 625 of the 5000
    # templates against 40 stations x 3 components; 125 000 of the 1M sources against 40 stations (10 closest
    # weighted) followed by the packed-key all-reduce.  One warm-up and one timed step each, same barrier and
    # max-over-ranks timing as the headline.
    shares = None
    if (world > 1 or dist is not None) and not args.skip_shares:
        shares = {}
        try:
            del data, tmpl, mv, w
        except Exception:
            pass
        torch.cuda.empty_cache()
        c4 = dict(syn.MF_CONFIGS["cfg4_per_gpu"])
        T4, S4, C4, L4, N4 = c4["T"], c4["S"], c4["C"], c4["L"], c4["N"]
        tmpl4, mv4, w4, data4, _ = mf_inputs_device(c4, device, 20260931, rank)
        # replication of the day (SURVEY.md 8e "broadcast once per day"): configs[3]'s 4.15 GB of data in
        # pageable host memory on rank 0 only -> upload there -> broadcast over RCCL / xGMI
        # (parallel.broadcast_day, what workflow.sharded_matched_filter_detections(data_src=0) does); the
        # other ranks' hosts never touch it.  Every rank's bit-pattern checksum of what it received is compared with rank 0's own.
        from seismic_bpmf_amd import parallel
        host4 = data4.cpu().numpy() if rank == 0 else None
        barrier()
        t0 = time.perf_counter()
        got4 = parallel.broadcast_day(host4, 0, device)
        barrier()
        bc_ms = (time.perf_counter() - t0) * 1e3
        chk = got4.view(torch.int32).to(torch.int64).sum().reshape(1)      # bit-pattern checksum of what arrived
        chks = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(chks, chk)
        ref_chk = data4.view(torch.int32).to(torch.int64).sum().reshape(1) if rank == 0 else chk
        shares["data_broadcast"] = {
            "what": f"configs[3]'s day of data, {data4.numel() * 4 / 1e9:.2f} GB: pageable host memory of rank 0 -> HBM of rank 0 -> "
                    f"dist.broadcast to {world} rank(s) (parallel.broadcast_day)",
            "ms": round(bc_ms, 1), "GB_per_s_per_receiver": round(data4.numel() * 4 / 1e9 / (bc_ms * 1e-3), 1),
            "every_rank_received_rank0s_bits": bool(all(int(c.item()) == int(ref_chk.item()) for c in chks))}
        del got4, host4
        torch.cuda.empty_cache()
        mf4 = sb.MatchedFilterGPU(device=local_rank)
        mf4.set_data(data4)
        cc4 = torch.empty((T4, N4 - L4 + 1), dtype=torch.float32, device=device)

        def mf4_step():
            mf4._prepared_for = None
            mf4.run(tmpl4, mv4, w4, 1, out=cc4)

        dt4 = timed(mf4_step, 1, 1)
        k4 = _lib.profile_times_ms(_lib.KERNEL_MF_MAIN)
        k4_ms = float(np.mean(k4)) if k4 else float("nan")
        tf4 = 2.0 * L4 * S4 * C4 * T4 * (N4 - L4 + 1) / (k4_ms * 1e-3) / 1e12
        shares["mf_configs3_share"] = {
            "workload": f"BASELINE configs[3], one GPU's share per rank: {T4} templates x {S4} stations x {C4} comp, "
                        f"L={L4}, N={N4}, step 1 ({world} ranks = {world * T4} templates)",
            "value": round(world * T4 * (N4 - L4 + 1) / dt4 / 1e6, 2), "unit": "M CC-samples/s (120-channel samples)",
            "ms_per_step": round(dt4 * 1e3, 2), "steps": 1, "warmup": 1,
            "roofline": {"kernel": "mf_mfma_wave_kernel", "bound": "mfma", "achieved": round(tf4, 2),
                         "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf4 / FP32_PEAK_TFLOPS, 4),
                         "avg_launch_ms": round(k4_ms, 3)}}
        del cc4, mf4, tmpl4, mv4, w4, data4
        torch.cuda.empty_cache()
        if not args.skip_bp:
            c5 = dict(syn.BP_CONFIGS["cfg5_per_gpu"])
            geo5, feat5, wp5 = bp_inputs(c5, device, 20260928, rank, world)
            K5 = geo5["moveouts"].shape[0]
            bf5 = sb.BeamformerGPU(geo5["moveouts"], geo5["weights_sources"], device=local_rank,
                                   source_id_offset=rank * K5)
            beam5 = torch.empty(c5["N"], dtype=torch.float32, device=device)
            arg5 = torch.empty(c5["N"], dtype=torch.int32, device=device)

            def bp5_step():
                bf5.run(feat5, wp5, "max", "strict", out=(beam5, arg5))
                if dist is not None:
                    packed = bf5.pack_max(beam5, arg5)
                    dist.all_reduce(packed, op=dist.ReduceOp.MAX)
                    bf5.unpack_max(packed)

            dt5 = timed(bp5_step, 1, 1)
            k5 = _lib.profile_times_ms(_lib.KERNEL_BP_BEAM)
            k5_ms = float(np.mean(k5)) if k5 else float("nan")
            s_act5 = float((geo5["weights_sources"] != 0).sum(axis=1).mean())
            tb5 = 4.0 * s_act5 * c5["P"] * K5 * c5["N"] / (k5_ms * 1e-3) / 1e12
            info5 = bf5.plan_info()
            peak5 = LDS_B32_PEAK_TBS * (2.0 if info5["gather_bytes"] == 8 else 1.0)
            shares["bp_configs4_share"] = {
                "workload": f"BASELINE configs[4], one GPU's share per rank: {K5} sources x {c5['S']} stations x {c5['C']} comp "
                            f"x {c5['P']} phases, N={c5['N']}, {s_act5:.1f} active stations/source ({world} ranks = "
                            f"{world * K5} sources), packed-key all-reduce(MAX) per step",
                "value": world * K5 * c5["N"] / dt5, "unit": "grid-points x samples / s",
                "ms_per_step": round(dt5 * 1e3, 2), "steps": 1, "warmup": 1,
                "roofline": {"kernel": "bp_beam_fast_kernel", "bound": "lds-gather", "achieved": round(tb5, 2),
                             "peak": round(peak5, 1), "unit": "TB/s", "frac": round(tb5 / peak5, 4),
                             "avg_launch_ms": round(k5_ms, 3)}}
            bf5.close()
            del beam5, arg5, feat5
            torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and dist is None and not args.skip_cpu:
        cpu = cpu_baseline(cfg, args.cpu_seconds)
    # every rank's own step time, kernel time and roofline fraction into `ranks` (a straggler GPU -- the boxes of
    # one pool differ by +-5 % -- is then visible in a weak-scaling ratio)
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, rank_stats)
        merge_rank_stats(ranks_info, per_rank)

    # roofline.traffic measured in this run (separate PMC passes in child processes, after everything
    # else, with this process's device memory released); headline workloads only
    if rank == 0 and world == 1 and dist is None and not args.skip_traffic:
        try:
            del data, tmpl, mv, w
        except Exception:
            pass
        torch.cuda.empty_cache()
        if args.mf_config == "cfg2" and roofline is not None:
            tr = measure_traffic("prof_mf.py", ["mf_mfma"], ["500"])
            if tr:
                roofline["traffic"] = tr["hbm_bytes_per_launch"]
                roofline["traffic_source"] = ("measured in this run: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over "
                                              "tools/prof_mf.py 500 (the same kernel on cfg2's shape), FETCH_SIZE x 2 (gfx950), "
                                              f"{tr['FETCH_SIZE_launches']} launches; fetched {tr['FETCH_SIZE'] * 2 / 1e9:.1f} GB, "
                                              f"written {tr['WRITE_SIZE'] / 1e9:.2f} GB")
        if bp_obj is not None and args.bp_config == "cfg3" and not args.skip_bp:
            tr = measure_traffic("prof_bp.py", ["bp_beam_fast", "bp_beam_wps2"])
            if tr:
                bp_obj["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                bp_obj["roofline"]["traffic_source"] = (
                    "measured in this run: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/prof_bp.py "
                    "(interior + edge kernels of cfg3), FETCH_SIZE x 2 (gfx950); "
                    f"fetched {tr['FETCH_SIZE'] * 2 / 1e9:.2f} GB, written {tr['WRITE_SIZE'] / 1e6:.1f} MB")

    line = None
    if rank == 0:
        line = assemble_line(args=args, world=world, n_gpus=(ranks_info["rccl_ranks"] if ranks_info else 1), mf_value=mf_value,
                             mf_dt=mf_dt, dims=(T, S, C, L, N), peak=peak, roofline=roofline, cpu=cpu, e2e=e2e,
                             mf_shapes=mf_shapes, bp_obj=bp_obj, detect=detect, shares=shares, compat=compat,
                             split16=split16, ranks_info=ranks_info)
    if dist is not None:
        dist.destroy_process_group()
    # N > 1: the CPU baseline AFTER the process group is gone -- the other ranks have exited (a rank waiting in a
    # collective spins on a host core, and the cgroup grants 16), rank 0 times the oracle alone, exactly as at N = 1
    # (taken whenever a process group was initialised -- BPMF_BENCH_FORCE_DIST=1 runs it with one rank on a one-GPU box)
    if line is not None and dist is not None and not args.skip_cpu:
        try:
            del data, tmpl, mv, w
        except Exception:
            pass
        torch.cuda.empty_cache()
        line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_seconds)
        if isinstance(line["cpu_baseline"], dict):
            line["cpu_baseline"]["measured_at_n"] = world
            line["cpu_baseline"]["when"] = "after the timed regions, the process group destroyed: the other ranks' processes have exited"
        if line.get("bp") is not None and not args.skip_bp:
            line["bp"]["cpu_baseline"] = cpu_baseline_bp(bcfg, geo, max(2.0, args.cpu_seconds / 3))
    if line is not None:
        def _clean(o):  # NaN is not JSON
            if isinstance(o, float) and not math.isfinite(o):
                return None
            if isinstance(o, dict):
                return {k: _clean(v) for k, v in o.items()}
            return o
        # the JSON line is the LAST thing on stdout: flush C stdio first (RCCL prints a version
        # banner through it)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(_clean(line)), flush=True)


if __name__ == "__main__":
    main()
