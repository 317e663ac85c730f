"""What the box's host CPUs really are, and how the CPU oracle scales on them.

Prints the affinity mask, the cgroup quota, the load average, and the time of the oracle's matched
filter on one sample at 1, 2, 4, ... threads (preparation and main loop separately when the oracle
exports bpmf_oracle_last_phase_seconds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from seismic_bpmf_amd import synthetic as syn


def read(path):
    try:
        return open(path).read().strip()
    except Exception as e:
        return f"<{type(e).__name__}>"


print("os.cpu_count()", os.cpu_count())
print("sched_getaffinity", len(os.sched_getaffinity(0)))
print("cgroup cpu.max", read("/sys/fs/cgroup/cpu.max"))
print("cgroup v1 quota", read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), read("/sys/fs/cgroup/cpu/cpu.cfs_period_us"))
print("cpuset", read("/sys/fs/cgroup/cpuset.cpus.effective"))
print("smt", read("/sys/devices/system/cpu/smt/active"))
print("loadavg", read("/proc/loadavg"))
print("OMP env", {k: v for k, v in os.environ.items() if k.startswith(("OMP", "GOMP", "KMP"))})
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA|MHz' ")
lib = oracle.load(oracle.build(march="native", out_dir="/tmp/bpmf_oracle_native"))
print("omp max threads", lib.bpmf_oracle_max_threads())
T, S, C, L = 8, 20, 3, 256
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
inp = syn.make_mf_inputs(T, S, C, L, N, seed=8, n_events=0)
flop = 2.0 * L * S * C * T * (N - L + 1)
for nth in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if nth > 2 * (os.cpu_count() or 1):
        break
    best = 1e9
    for _ in range(2 if nth < 8 else 3):
        t0 = time.perf_counter()
        oracle.matched_filter(inp["templates"], inp["moveouts"], inp["weights"], inp["data"], 1,
                              num_threads=nth, lib=lib)
        best = min(best, time.perf_counter() - t0)
    extra = ""
    if hasattr(lib, "bpmf_oracle_last_phase_seconds"):
        import ctypes
        ph = (ctypes.c_double * 4)()
        lib.bpmf_oracle_last_phase_seconds(ph)
        extra = f" prep {ph[0]:.3f}s main {ph[1]:.3f}s"
    print(f"threads {nth:4d}: {best:.3f} s  {T*(N-L+1)/best/1e6:8.2f} M CC/s  {flop/best/1e9:8.1f} GFLOP/s{extra}", flush=True)
