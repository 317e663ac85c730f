"""cfg3 through the drop-in call (beampower.beamform, NumPy in / out; BPMF/template_search.py:549-558) beside
the resident engine, with the library's own account of the call (bpmf_host_call_stats): time to the first
kernel, host threads copying the day into the pinned pieces, the wait for the device after the last launch.
   python tools/probe_bp_e2e.py [--hogs N] [--calls K]
--hogs N: N busy-loop processes beside the calls (the driver's round-5 bench ran at load average 39 on 16 CPUs)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hogs", type=int, default=0)
    ap.add_argument("--calls", type=int, default=5)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--release-after", type=int, default=-1, help="bpmf_release_device_memory(-1) behind this call")
    ap.add_argument("--opt", action="append", default=[], help="name=value library option")
    ap.add_argument("--warm-dma", action="store_true", help="1 GB pinned H2D / D2H through torch in front of the calls")
    ap.add_argument("--sleep", type=float, default=0.0, help="seconds of idle in front of every call")
    args = ap.parse_args()
    import numpy as np
    import torch
    import seismic_bpmf_amd as sb
    from seismic_bpmf_amd import _lib, synthetic as syn
    bcfg = dict(syn.BP_CONFIGS["cfg3"])
    geo = syn.make_bp_geometry(bcfg["grid"], bcfg["S"], bcfg["P"], bcfg["sr"], n_closest=bcfg.get("n_closest", 10))
    N, S, C = bcfg["N"], bcfg["S"], bcfg["C"]
    g = torch.Generator(device="cuda").manual_seed(3)
    feat = torch.randn((S, C, N), device="cuda", generator=g).abs_()
    wp = syn.phase_weights(S, C, bcfg["P"])
    bf = sb.BeamformerGPU(geo["moveouts"], geo["weights_sources"], device=0)
    wpd = torch.as_tensor(wp, device="cuda")
    beam, arg = bf.run(feat, wpd, "max", "strict")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        bf.run(feat, wpd, "max", "strict", out=(beam, arg))
    torch.cuda.synchronize()
    resident = (time.perf_counter() - t0) / 3 * 1e3
    print(f"resident: {resident:.1f} ms per day (usable CPUs by the library's count: see fill_threads below; "
          f"os.cpu_count {os.cpu_count()}, loadavg {os.getloadavg()[0]:.1f})")
    h_f = feat.cpu().numpy()
    want_b, want_a = beam.cpu().numpy(), arg.cpu().numpy()
    for kv in args.opt:
        k, v = kv.split('=')
        _lib.set_option(k, int(v))
    if args.verbose:
        _lib.set_option("bp.verbose", 1)
    if args.warm_dma:
        pin = torch.empty(256 << 20, dtype=torch.float32).pin_memory()
        for _ in range(3):
            dv = pin.cuda(non_blocking=True)
            pin.copy_(dv, non_blocking=True)
            torch.cuda.synchronize()
        del pin, dv
    # (fresh interpreters, not forks of this process: a fork after the HIP runtime is up shares its KFD event pages, and
    # the first blocking wait of the parent then sat out a 10 s timeout -- an artefact of the probe, seen as "plan_ms
    # 10 000" on the first hogged call)
    import subprocess
    import atexit
    hogs = [subprocess.Popen([sys.executable, "-c", "while True: pass"]) for _ in range(args.hogs)]
    atexit.register(lambda: [h.kill() for h in hogs if h.poll() is None])
    if hogs:
        time.sleep(2.0)
        print(f"{args.hogs} hogs running, loadavg {os.getloadavg()[0]:.1f}")
    def throttle():
        # cgroup v2 cpu.stat of this container: periods in which the quota ran out and every thread of the group --
        # the calling thread and the copy pool included -- was frozen until the next period
        try:
            kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
            return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
        except OSError:
            return 0, 0
    for i in range(args.calls):
        h_new = h_f.copy()
        th0 = throttle()
        if args.sleep:
            time.sleep(args.sleep)
        t0 = time.perf_counter()
        hb, ha = sb.beamform(h_new, geo["moveouts"], wp, geo["weights_sources"], device="gpu", reduce="max",
                             out_of_bounds="strict", device_id=[0])
        ms = (time.perf_counter() - t0) * 1e3
        st = _lib.host_call_stats()
        ok = bool(np.array_equal(hb, want_b) and np.array_equal(ha, want_a))
        th1 = throttle()
        print(f"call {i}: {ms:.1f} ms (+{ms - resident:.1f} over resident), equal {ok}, cgroup throttled {th1[0] - th0[0]} periods / {(th1[1] - th0[1]) / 1e3:.1f} ms, library: " +
              ", ".join(f"{k} {v:.1f}" if isinstance(v, float) else f"{k} {v}" for k, v in st.items()), flush=True)
        del h_new
        if i == args.release_after:
            _lib.release_device_memory(-1)
            print('   (device working set and pinned pieces released)')
    for h in hogs:
        h.kill()
    for h in hogs:
        h.wait()


if __name__ == "__main__":
    main()
