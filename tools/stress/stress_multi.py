"""Stress reproducer for the host-pointer multi-device entry points (bpmf_mf_run_multi /
bpmf_bp_run_multi): P processes x Q Python threads, each making small calls with the SAME GPU listed
2-6 times (the shape of tests/test_gpu_fuzz_adjacent.py::test_fuzz_adjacent_device_lists_and_plan_cache,
which lost workers to SIGABRT / SIGSEGV under 8 xdist processes in round 3).  Every result is
compared bit for bit with the single-device call of the same library.  A crashing process prints the
C stack of the faulting thread (tools/stress/crash_bt.c); the parent reports exit codes.

    python tools/stress/stress_multi.py --procs 8 --threads 1 --calls 300 [--lib PATH] [--out DIR]

Test infrastructure only; talks to the library through ctypes directly (no package import), so that
an older build of libbpmf_hip.so can be put beside the current one (--lib).
"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DEFAULT_LIB = os.path.join(ROOT, "seismic_bpmf_amd", "lib", "libbpmf_hip.so")
CRASH_SRC = os.path.join(HERE, "crash_bt.c")
CRASH_LIB = os.path.join(HERE, "libcrashbt.so")


def build_crash_lib():
    if not os.path.exists(CRASH_LIB) or os.path.getmtime(CRASH_LIB) < os.path.getmtime(CRASH_SRC):
        subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-shared", CRASH_SRC, "-o", CRASH_LIB])
    return CRASH_LIB


def bind(path):
    lib = C.CDLL(path)
    f, i, sz = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_size_t
    lib.bpmf_last_error.restype = C.c_char_p
    if hasattr(lib, "bpmf_set_option"):
        lib.bpmf_set_option.restype = C.c_int
        lib.bpmf_set_option.argtypes = [C.c_char_p, C.c_long]
    lib.bpmf_mf_run_multi.restype = C.c_int
    lib.bpmf_mf_run_multi.argtypes = [f, i, f, f, sz, sz, sz, sz, sz, sz, sz, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_int), f]
    lib.bpmf_bp_run_multi.restype = C.c_int
    lib.bpmf_bp_run_multi.argtypes = [f, i, f, f, sz, sz, sz, sz, sz, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_int), f, i]
    return lib


def child(args):
    import numpy as np
    if not args.no_torch:
        import torch  # noqa: F401  (the product binds to torch's HIP runtime; same here)
    crash = C.CDLL(build_crash_lib())
    crash.crash_bt_install(f"proc{args.child}".encode())
    lib = bind(args.lib)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    # --virtual K (round 5): K logical devices, each with its own context and HOST THREAD inside the *_run_multi
    # entry points (option debug.virtual_devices) -- the device lists below then name DISTINCT devices in random
    # orders, so that every call starts threads, hands the day from the first device to the others and merges
    nv = args.virtual
    if nv:
        assert lib.bpmf_set_option(b"debug.virtual_devices", nv) == 0
        assert lib.bpmf_set_option(b"mf.host_piece_lags", 4096) == 0       # the small cases cross piece boundaries too
        assert lib.bpmf_set_option(b"bp.host_piece_samples", 1024) == 0

    def device_list(rng, n_dev):
        if not nv:
            return [0] * n_dev
        devs = [int(x) for x in rng.permutation(nv)[: min(n_dev, nv)]]
        if rng.random() < 0.3:
            devs.append(devs[0])           # a device listed twice: two blocks on one of the threads
        return devs

    def ptr(a, t):
        return a.ctypes.data_as(t)

    def mf(tp, mv, w, data, ns, devs):
        T, S, Cc, L = tp.shape
        N = data.shape[-1]
        n_corr = N - L + 1
        out = np.zeros((T, n_corr) if ns else (T, n_corr, S, Cc), np.float32)
        d = (C.c_int * len(devs))(*devs)
        rc = lib.bpmf_mf_run_multi(ptr(tp, fp), ptr(mv, ip), ptr(w, fp), ptr(data, fp), 1, L, N, T, S, Cc,
                                   n_corr, 1 if ns else 0, 0, len(devs), d, ptr(out, fp))
        if rc:
            raise RuntimeError(f"mf rc={rc}: {lib.bpmf_last_error().decode()}")
        return out

    def bp(f, tau, wp, ws, oob, devs):
        S, Cc, N = f.shape
        K, _, P = tau.shape
        beam, arg = np.zeros(N, np.float32), np.zeros(N, np.int32)
        d = (C.c_int * len(devs))(*devs)
        rc = lib.bpmf_bp_run_multi(ptr(f, fp), ptr(tau, ip), ptr(wp, fp), ptr(ws, fp), N, K, S, Cc, P, oob, 0,
                                   len(devs), d, ptr(beam, fp), ptr(arg, ip))
        if rc:
            raise RuntimeError(f"bp rc={rc}: {lib.bpmf_last_error().decode()}")
        return beam, arg

    errors = []
    counts = [0] * args.threads

    def worker(q):
        rng = np.random.default_rng(1_000_003 * args.child + 7919 * q + args.seed)
        try:
            for it in range(args.calls):
                n_dev = int(rng.integers(2, 7))
                scale = args.scale
                T, S, Cc = int(rng.integers(1, 9)), int(rng.integers(1, 4)), int(rng.integers(1, 3))
                L = int(rng.choice([8, 64, 300]))
                N = int(L + rng.choice([0, 900, 5_000]) * scale)
                tp = rng.standard_normal((T, S, Cc, L)).astype(np.float32)
                data = rng.standard_normal((S, Cc, N)).astype(np.float32)
                mv = rng.integers(-30, 200, (T, S, Cc)).astype(np.int32)
                w = rng.random((T, S, Cc)).astype(np.float32)
                for ns in (True, False):
                    want = mf(tp, mv, w, data, ns, [0])
                    got = mf(tp, mv, w, data, ns, device_list(rng, n_dev))
                    if not np.array_equal(got, want):
                        raise AssertionError(f"MF mismatch proc {args.child} thread {q} it {it}")
                K, Sb = int(rng.integers(1, 500)), int(rng.integers(1, 9))
                P, Nb = int(rng.choice([1, 2, 2, 3])), int(rng.choice([300, 2_000, 7_000]) * scale)
                f = np.round(np.abs(rng.standard_normal((Sb, 2, Nb))) * 2).astype(np.float32)
                wp = rng.random((Sb, 2, P)).astype(np.float32)
                tables = []
                for _ in range(int(rng.integers(2, 6))):
                    tau = rng.integers(0, 150, (K, Sb, P)).astype(np.int32)
                    ws = rng.random((K, Sb)).astype(np.float32)
                    ws[rng.random((K, Sb)) < 0.3] = 0.0
                    tables.append((tau, ws))
                want = {}
                for j in rng.integers(0, len(tables), 8):
                    tau, ws = tables[j]
                    oob = int(j % 2)
                    if j not in want:
                        want[j] = bp(f, tau, wp, ws, oob, [0])
                    devs = device_list(rng, n_dev) if rng.random() < 0.7 else [0]
                    mb, ma = bp(f, tau, wp, ws, oob, devs)
                    if not (np.array_equal(mb, want[j][0]) and np.array_equal(ma, want[j][1])):
                        raise AssertionError(f"BP mismatch proc {args.child} thread {q} it {it}")
                counts[q] = it + 1
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(q,)) for q in range(args.threads)]
    t0 = time.time()
    for t in th:
        t.start()
    for t in th:
        t.join()
    print(f"proc {args.child}: {sum(counts)} iterations in {time.time() - t0:.1f} s, errors: {errors}", flush=True)
    sys.exit(1 if errors else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--calls", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--scale", type=int, default=1, help="multiplies the series lengths")
    ap.add_argument("--lib", default=DEFAULT_LIB)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stress"))
    ap.add_argument("--no-torch", action="store_true")
    ap.add_argument("--timeout", type=int, default=1500)
    ap.add_argument("--child", type=int, default=-1)
    ap.add_argument("--virtual", type=int, default=0,
                    help="debug.virtual_devices = K in every worker: the multi-device calls start one host thread per "
                         "logical device (0 = the same GPU listed 2-6 times: one thread)")
    ap.add_argument("--hogs", type=int, default=0,
                    help="busy-loop processes beside the workers (CPU oversubscription: what 8 xdist workers x 16 idle-"
                         "spinning OpenMP threads of the oracle did to the round-3 sessions that crashed)")
    args = ap.parse_args()
    if args.child >= 0:
        return child(args)
    build_crash_lib()
    os.makedirs(args.out, exist_ok=True)
    procs = []
    hogs = [subprocess.Popen([sys.executable, "-c", "while True: pass"]) for _ in range(args.hogs)]
    t0 = time.time()
    for p in range(args.procs):
        log = open(os.path.join(args.out, f"proc{p}.log"), "w")
        cmd = [sys.executable, os.path.abspath(__file__), "--child", str(p), "--threads", str(args.threads),
               "--calls", str(args.calls), "--seed", str(args.seed), "--scale", str(args.scale), "--lib", args.lib,
               "--virtual", str(args.virtual)]
        if args.no_torch:
            cmd.append("--no-torch")
        procs.append((subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT), log))
    rcs = []
    for pr, log in procs:
        try:
            rcs.append(pr.wait(timeout=max(1, args.timeout - (time.time() - t0))))
        except subprocess.TimeoutExpired:
            pr.kill()
            rcs.append("timeout")
        log.close()
    for h in hogs:                 # (exact PIDs of processes this script started)
        h.kill()
        h.wait()
    dead = [(p, rc) for p, rc in enumerate(rcs) if rc != 0]
    print(f"stress {os.path.basename(args.lib)} procs={args.procs} threads={args.threads} calls={args.calls}: "
          f"exit codes {rcs} in {time.time() - t0:.1f} s; {len(dead)} abnormal")
    for p, rc in dead:
        print(f"--- proc {p} (exit {rc}) log tail ---")
        with open(os.path.join(args.out, f"proc{p}.log")) as fh:
            print("".join(fh.readlines()[-70:]))
    sys.exit(1 if dead else 0)


if __name__ == "__main__":
    main()
