#!/usr/bin/env python3
"""First-contact kit: diff the REAL upstream packages against this build's conventions.

The arithmetic of both hot paths lives in two third-party packages BPMF imports un-vendored and
un-pinned (`fast_matched_filter`, `beampower`; /root/reference/pyproject.toml:28-29; call sites
BPMF/similarity_search.py:526-533, BPMF/dataset.py:4818-4830, BPMF/template_search.py:529-569).  Neither
exists in the image this build was made in, so the CPU oracle of the hot paths (oracle/bpmf_oracle.c) is
"parity unpinned": every convention the call sites do not fix was chosen here, and each has a switch
(`bpmf_set_option("*.compat_*")`, mirrored by `oracle.compat(flags)`).  THIS script is what turns
"unpinned" into "pinned" on a machine that has the real packages:

    python tools/diff_upstream.py                # needs fast_matched_filter and / or beampower installed
    python tools/diff_upstream.py --only mf      # one package
    python tools/diff_upstream.py --write        # also (re)write tests/golden/upstream_{mf,bp}.npz

For every probe case (small committed-by-construction inputs: a seeded generator below) it runs upstream
on the CPU (`arch="cpu"` / `device="cpu"`), each case in its own subprocess (a probe such as "zero-weight
channel whose moveout points outside the trace" may crash an implementation that does not skip it), and
compares the result with the oracle under EVERY combination of the compat flags of that path.  It prints,
per case, which combinations are bit-equal -- and the number of differing values and max |diff| of the
closest one otherwise -- and at the end the combinations that are bit-equal on every case that
discriminates, as the `bpmf_set_option` lines to put in front of a workflow.  With --write the inputs,
upstream's outputs, the package versions and the matching combinations go to
tests/golden/upstream_{mf,bp}.npz; tests/test_upstream_live.py then holds the oracle (CPU suite) and the
HIP library (GPU suite) to those vectors on any machine, with or without the packages.

The import shims of this repo (shims/fast_matched_filter, shims/beampower) are refused: diffing the build
against itself proves nothing.
"""
import argparse
import itertools
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = {"mf": os.path.join(ROOT, "tests", "golden", "upstream_mf.npz"),
          "bp": os.path.join(ROOT, "tests", "golden", "upstream_bp.npz")}


# ------------------------------------------------------------------------------ detection ---
def find_upstream(which):
    """The real package module ('mf' -> fast_matched_filter, 'bp' -> beampower) or (None, reason)."""
    name = {"mf": "fast_matched_filter", "bp": "beampower"}[which]
    try:
        mod = __import__(name)
    except Exception as exc:                         # ImportError, or a broken native library
        return None, f"{name} does not import here ({type(exc).__name__}: {exc})"
    path = os.path.realpath(getattr(mod, "__file__", "") or "")
    shim_dir = os.path.realpath(os.path.join(ROOT, "shims"))
    fn = mod.matched_filter if which == "mf" else getattr(getattr(mod, "beampower", mod), "beamform", None)
    if getattr(mod, "__bpmf_shim__", False) or path.startswith(shim_dir + os.sep) or \
            (getattr(fn, "__module__", "") or "").startswith("seismic_bpmf_amd"):
        return None, f"{name} resolves to this repository's import shim ({path}): refused"
    return mod, path


def package_version(mod):
    for attr in ("__version__", "version"):
        v = getattr(mod, attr, None)
        if isinstance(v, str):
            return v
    try:
        from importlib import metadata
        return metadata.version(mod.__name__)
    except Exception:
        return "unknown"


# ------------------------------------------------------------------------------ probe cases ---
def mf_cases():
    """name -> dict(templates, moveouts, weights, data, step, network_sum, probes=<what the case tells apart>)."""
    cases = {}

    def base(seed, T=3, S=3, C=3, L=24, N=1500, lo=0, hi=60):
        rng = np.random.default_rng(seed)
        tp = rng.standard_normal((T, S, C, L)).astype(np.float32)
        tp -= tp.mean(axis=-1, keepdims=True)
        d = rng.standard_normal((S, C, N)).astype(np.float32)
        mv = rng.integers(lo, hi, (T, S, C)).astype(np.int32)
        w = (rng.random((T, S, C)) + 0.1).astype(np.float32)
        w /= w.reshape(T, -1).sum(axis=1)[:, None, None]
        return dict(templates=tp, moveouts=mv, weights=w, data=d, step=1, network_sum=True)

    c = base(1)
    c["probes"] = "arithmetic of the CC and of the network sum; the last valid lag (mf.compat_exclusive_last_lag)"
    cases["positive_moveouts"] = c
    c = base(2, lo=-40, hi=70)
    c["probes"] = "first valid lag with negative moveouts; last valid lag"
    cases["signed_moveouts"] = c
    c = base(3, lo=-90, hi=-5)
    c["probes"] = "every moveout negative: where the valid range starts and ends"
    cases["all_negative_moveouts"] = c
    c = base(4, lo=-20, hi=50)
    c["step"] = 3
    c["probes"] = "step > 1: which lags are evaluated, rounding of the first valid lag up to a multiple of step"
    cases["step_3"] = c
    c = base(5, lo=0, hi=50)
    c["weights"][0, 1] = 0.0
    c["moveouts"][0, 1] = [-200, 400, 10]            # inside the trace, beyond every weighted moveout
    c["weights"][2, 0, 2] = 0.0
    c["weights"] /= c["weights"].reshape(3, -1).sum(axis=1)[:, None, None]
    c["probes"] = "valid lag range over the weighted channels only, or over all (mf.compat_range_all_channels)"
    cases["zero_weight_channels_extreme_moveouts"] = c
    c = base(6, lo=0, hi=50)
    c["weights"][1, 2, 0] = 0.0
    c["moveouts"][1, 2, 0] = 100_000                 # far outside the trace: an implementation that does not skip may crash
    c["probes"] = "a zero-weight channel whose moveout points outside the trace (crash-isolated)"
    c["may_crash"] = True
    cases["zero_weight_channel_outside_the_trace"] = c
    c = base(7, lo=0, hi=40)
    c["data"][0, 0, 300:420] = 0.0                   # zero-energy windows
    c["data"][1, :, 900:] = 0.0
    c["templates"][2, 1, 1] = 0.0                    # a zero-energy template channel with a non-zero weight
    c["probes"] = "zero-energy windows / templates: the stability rule and its threshold (mf.compat_sqrt_norm)"
    cases["zero_energy"] = c
    c = base(8, lo=-10, hi=40)
    c["network_sum"] = False
    c["probes"] = "network_sum=False: layout (T, n_corr, S, C), unweighted CCs (BPMF/dataset.py:4818-4830)"
    cases["per_channel_output"] = c
    c = base(9, T=2, S=2, C=2, L=32, N=120_000, lo=0, hi=200)
    c["data"] *= np.exp(2.0 * np.sin(np.arange(120_000) / 5000.0)).astype(np.float32)
    c["probes"] = "rounding of the window energies: double prefix sum as one chain or hierarchical (mf.compat_sequential_csum), sqrt form"
    cases["long_trace_amplitude_swing"] = c
    c = base(10, T=2, S=2, C=1, L=16, N=400, lo=0, hi=1)
    c["moveouts"][:] = 0
    c["moveouts"][1, 1, 0] = 37
    c["probes"] = "the last lag exactly: N - L - mv_max inclusive or exclusive"
    cases["last_lag"] = c
    return cases


def bp_cases():
    cases = {}

    def base(seed, K=40, S=5, C=3, P=2, N=1200, lo=0, hi=90, integer=False):
        rng = np.random.default_rng(seed)
        f = np.abs(rng.standard_normal((S, C, N))).astype(np.float32)
        if integer:
            f = np.round(f * 2).astype(np.float32)
        tau = rng.integers(lo, hi, (K, S, P)).astype(np.int32)
        wp = np.zeros((S, C, P), np.float32)
        wp[:, 0, 0] = 1.0
        wp[:, 1:, 1] = 0.5
        ws = (rng.random((K, S)) < 0.6).astype(np.float32)
        ws[ws.sum(axis=1) == 0, 0] = 1.0
        return dict(features=f, moveouts=tau, weights_phases=wp, weights_sources=ws, out_of_bounds="strict", reduce="max")

    c = base(1)
    c["weights_sources"] *= np.random.default_rng(5).random(c["weights_sources"].shape).astype(np.float32) + 0.5
    c["probes"] = "arithmetic: prestack, summation order over stations and phases"
    cases["strict_max"] = c
    c = dict(base(2), out_of_bounds="flexible")
    c["probes"] = "flexible: out-of-range terms dropped"
    cases["flexible_max"] = c
    c = base(3, integer=True)
    c["moveouts"][25] = c["moveouts"][4]
    c["weights_sources"][25] = c["weights_sources"][4]
    c["moveouts"][31] = c["moveouts"][9]
    c["weights_sources"][31] = c["weights_sources"][9]
    c["probes"] = "exact ties between sources: lowest index wins?"
    cases["ties"] = c
    c = base(4)
    c["features"] = -c["features"] - 1.0
    c["probes"] = "every beam negative: does the running max start at 0 or at the first computed beam (bp.compat_first_computed)"
    cases["negative_beams"] = c
    c = base(5)
    for k in range(0, 40, 4):
        z = np.flatnonzero(c["weights_sources"][k] == 0)
        if z.size:
            c["moveouts"][k, z[0]] = [0, 700 + k]
    c["probes"] = "strict range over the weighted stations, or over all (bp.compat_range_all_stations)"
    cases["zero_weight_stations_extreme_moveouts"] = c
    c = base(6, lo=-60, hi=80)
    c["probes"] = "negative moveouts under strict: lower bound tested or not (bp.compat_strict_upper_only); crash-isolated"
    c["may_crash"] = True
    cases["negative_moveouts_strict"] = c
    c = dict(base(7, K=12, N=500), reduce="none")
    c["probes"] = 'reduce="none": (K, N) layout, value of not-computed beams'
    cases["strict_none"] = c
    c = dict(base(8, K=12, N=500), reduce="none", out_of_bounds="flexible")
    c["probes"] = 'reduce="none", flexible'
    cases["flexible_none"] = c
    return cases


# ------------------------------------------------------------------------------ runners ---
def run_upstream(which, case):
    """Upstream's output for one case (called inside the per-case subprocess)."""
    if which == "mf":
        import fast_matched_filter as fmf
        return np.asarray(fmf.matched_filter(case["templates"], case["moveouts"], case["weights"], case["data"],
                                             int(case["step"]), arch="cpu", check_zeros=False,
                                             network_sum=bool(case["network_sum"])), dtype=np.float32)
    import beampower as bp
    out = bp.beampower.beamform(case["features"], case["moveouts"], case["weights_phases"], case["weights_sources"],
                                device="cpu", out_of_bounds=str(case["out_of_bounds"]), reduce=str(case["reduce"]))
    if str(case["reduce"]) == "max":
        return np.asarray(out[0], dtype=np.float32), np.asarray(out[1], dtype=np.int64)
    return np.asarray(out, dtype=np.float32)


def run_oracle(which, case, flags):
    from oracle import oracle
    with oracle.compat(flags):
        if which == "mf":
            return oracle.matched_filter(case["templates"], case["moveouts"], case["weights"], case["data"],
                                         int(case["step"]), bool(case["network_sum"]))
        return oracle.beamform(case["features"], case["moveouts"], case["weights_phases"], case["weights_sources"],
                               str(case["out_of_bounds"]), str(case["reduce"]))


def flag_sets(which):
    from oracle import oracle
    bits = [b for b, (_, path) in oracle.COMPAT_OPTIONS.items() if path == which]
    return [sum(c) for r in range(len(bits) + 1) for c in itertools.combinations(bits, r)]


def flag_names(flags):
    from oracle import oracle
    return [name for b, (name, _) in sorted(oracle.COMPAT_OPTIONS.items()) if flags & b]


def _child(which, case_file, out_file):
    """Subprocess body: one case through upstream; the result (or the exception text) into out_file."""
    case = dict(np.load(case_file, allow_pickle=False))
    case = {k: (v.item() if v.shape == () else v) for k, v in case.items()}
    try:
        res = run_upstream(which, case)
        if isinstance(res, tuple):
            np.savez(out_file, beam=res[0], arg=res[1])
        else:
            np.savez(out_file, out=res)
    except Exception as exc:                                   # noqa: BLE001 - reported to the parent
        np.savez(out_file, error=np.array(f"{type(exc).__name__}: {exc}"))


def upstream_in_subprocess(which, case, extra_path=()):
    arrays = {k: v for k, v in case.items() if k not in ("probes", "may_crash")}
    with tempfile.TemporaryDirectory() as tmp:
        cf, of = os.path.join(tmp, "case.npz"), os.path.join(tmp, "out.npz")
        np.savez(cf, **arrays)
        env = dict(os.environ)
        if extra_path:
            env["PYTHONPATH"] = os.pathsep.join(list(extra_path) + [env.get("PYTHONPATH", "")]).strip(os.pathsep)
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", which, cf, of], env=env,
                             capture_output=True, text=True, timeout=1800)
        if res.returncode != 0 or not os.path.exists(of):
            return None, f"upstream process ended with status {res.returncode}: {res.stderr.strip()[-300:]}"
        out = dict(np.load(of, allow_pickle=False))
        if "error" in out:
            return None, str(out["error"])
        return ((out["beam"], out["arg"]) if "beam" in out else out["out"]), None


def compare(up, mine):
    """(bit-equal?, differing values, max |diff|) of two results (arrays or (beam, arg) pairs)."""
    if isinstance(up, tuple):
        b_eq = np.array_equal(up[0], mine[0])
        a_eq = np.array_equal(np.asarray(up[1], np.int64), np.asarray(mine[1], np.int64))
        nd = int((up[0] != mine[0]).sum() + (np.asarray(up[1], np.int64) != np.asarray(mine[1], np.int64)).sum())
        return b_eq and a_eq, nd, float(np.nanmax(np.abs(up[0].astype(np.float64) - mine[0])))
    if up.shape != mine.shape:
        return False, -1, float("nan")
    both_nan = np.isnan(up) & np.isnan(mine)
    diff = (up != mine) & ~both_nan
    return not diff.any(), int(diff.sum()), float(np.nanmax(np.abs(up.astype(np.float64) - mine))) if up.size else 0.0


def diff_path(which, write=False, extra_path=(), quiet=False):
    """Run every case of one path; returns the report dict (also printed unless quiet)."""
    cases = mf_cases() if which == "mf" else bp_cases()
    sets = flag_sets(which)
    report = {"path": which, "cases": {}, "combinations": {str(f): flag_names(f) for f in sets}}
    golden = {}
    survivors = set(sets)
    say = (lambda *a: None) if quiet else print
    for name, case in cases.items():
        up, err = upstream_in_subprocess(which, case, extra_path)
        entry = {"probes": case["probes"]}
        if up is None:
            entry["upstream_error"] = err
            say(f"[{which}] {name}: upstream failed ({err})" + ("  [expected to be possible: crash probe]" if case.get("may_crash") else ""))
            report["cases"][name] = entry
            continue
        if which == "mf" and np.isnan(up).any():
            entry["upstream_nans"] = int(np.isnan(up).sum())
            up = np.nan_to_num(up, nan=0.0)           # BPMF scrubs them (similarity_search.py:540); the oracle never emits NaN
        rows = {}
        for f in sets:
            eq, nd, mx = compare(up, run_oracle(which, case, f))
            rows[f] = (eq, nd, mx)
        equal = [f for f in sets if rows[f][0]]
        entry["bit_equal"] = [flag_names(f) for f in equal]
        best = min(sets, key=lambda f: (rows[f][1] if rows[f][1] >= 0 else 1 << 60, rows[f][2]))
        entry["closest"] = {"combination": flag_names(best), "differing_values": rows[best][1], "max_abs_diff": rows[best][2]}
        discriminates = len({rows[f][1] for f in sets}) > 1 or len(equal) not in (0, len(sets))
        entry["discriminates"] = bool(discriminates)
        if equal:
            survivors &= set(equal)
        elif not case.get("may_crash"):
            survivors = set()
        say(f"[{which}] {name}: " + (f"bit-equal under {len(equal)} of {len(sets)} combinations, e.g. {flag_names(equal[0]) or ['(defaults)']}"
                                     if equal else
                                     f"NO combination is bit-equal; closest {flag_names(best) or ['(defaults)']}: "
                                     f"{rows[best][1]} values differ, max |diff| {rows[best][2]:.3g}"))
        report["cases"][name] = entry
        for k, v in case.items():
            if k not in ("probes", "may_crash"):
                golden[f"{name}/{k}"] = np.asarray(v)
        if isinstance(up, tuple):
            golden[f"{name}/upstream_beam"], golden[f"{name}/upstream_arg"] = up
        else:
            golden[f"{name}/upstream_out"] = up
        golden[f"{name}/bit_equal_flags"] = np.asarray(equal, dtype=np.int64)
    report["equal_on_every_case"] = [flag_names(f) for f in sorted(survivors)]
    report["equal_on_every_case_flags"] = sorted(int(f) for f in survivors)
    if survivors:
        f = min(survivors, key=lambda x: bin(x).count("1"))
        say(f"[{which}] ==> bit-equal on every case with: {flag_names(f) or '(the defaults: no switch needed)'}")
        for n in flag_names(f):
            say(f'        _lib.set_option("{n}", 1)')
        # a named profile (seismic_bpmf_amd.compat_profile) whose switches of THIS path are exactly these?
        try:
            from seismic_bpmf_amd import _lib as _plib
            mine = {n for n in flag_names(f)}
            for prof, sw in _plib.COMPAT_PROFILES.items():
                if {n for n, v in sw.items() if v and n.startswith(which + ".")} == mine:
                    say(f'[{which}] ==> that is profile "{prof}" for this path: seismic_bpmf_amd.compat_profile("{prof}")')
                    report["profile"] = prof
                    break
        except Exception:
            pass
    else:
        say(f"[{which}] ==> no single combination is bit-equal on every case: see the per-case lines; the closest "
            "combinations bound what a float32 tolerance must cover")
    if write and golden:
        mod, path = find_upstream(which)
        golden["meta"] = np.array(json.dumps({"package": getattr(mod, "__name__", "?"), "version": package_version(mod) if mod else "?",
                                              "module_file": str(path), "numpy": np.__version__,
                                              "equal_on_every_case_flags": report["equal_on_every_case_flags"]}))
        np.savez_compressed(GOLDEN[which], **golden)
        say(f"[{which}] wrote {GOLDEN[which]}")
    return report


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--only", choices=["mf", "bp"])
    ap.add_argument("--write", action="store_true", help="write tests/golden/upstream_{mf,bp}.npz")
    ap.add_argument("--json", help="also dump the full report to this file")
    ap.add_argument("--child", nargs=3, metavar=("PATH", "CASE", "OUT"), help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.child:
        _child(*args.child)
        return 0
    from oracle import oracle
    oracle.load()
    reports, missing = {}, []
    for which in ([args.only] if args.only else ["mf", "bp"]):
        mod, why = find_upstream(which)
        if mod is None:
            print(f"[{which}] skipped: {why}")
            missing.append(which)
            continue
        print(f"[{which}] upstream: {mod.__name__} {package_version(mod)} at {why}")
        reports[which] = diff_path(which, write=args.write)
    if args.json:
        json.dump(reports, open(args.json, "w"), indent=1)
    if not reports:
        print("nothing to diff: install FastMatchedFilter / beampower (pip install FastMatchedFilter beampower, "
              "or their GitHub sources) into this interpreter and run again")
        return 2
    return 0


if __name__ == "__main__":
    sys.exit(main())
