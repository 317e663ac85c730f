"""bench.py's launch contract (CPU): `--gpus N` never silently runs fewer than N ranks.

Round 4's bench read WORLD_SIZE from the environment and ran ONE rank when a driver called
`python bench.py --gpus 8` without a launcher; now it launches the N ranks itself (torch.distributed.run
on 127.0.0.1) or refuses with a message.  The multi-rank exchanges themselves are covered over gloo in
tests/test_dist_gloo.py; the reference has no counterpart (its back-ends split work across the GPUs of
one process, BPMF/similarity_search.py:526-533)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE") + tuple(drop)}
    env["HIP_VISIBLE_DEVICES"] = ""            # no device, whatever the box has
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_gpus_n_without_enough_devices_refuses_loudly():
    res = _run(["--gpus", "8"])
    assert res.returncode != 0
    assert "only 0 HIP device(s) visible" in res.stderr and "refusing to run fewer ranks" in res.stderr
    assert '"metric"' not in res.stdout        # no JSON line from a downgraded run


def test_world_size_and_gpus_must_agree():
    res = _run(["--gpus", "4"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"})
    assert res.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in res.stderr
    res = _run(["--gpus", "1"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"})
    assert res.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in res.stderr


def test_forced_dist_takes_the_self_launch_route_too():
    res = _run(["--gpus", "1"], {"BPMF_BENCH_FORCE_DIST": "1"})
    assert res.returncode != 0 and "only 0 HIP device(s) visible" in res.stderr


def _stub_line(bench, world):
    """assemble_line() on stubbed measurements of `world` ranks (what main() hands it)."""
    import argparse
    args = argparse.Namespace(steps=3, warmup=1, mf_config="cfg2", bp_config="cfg3")
    roofline = {"kernel": "mf_mfma_wave_kernel", "bound": "mfma", "achieved": 135.0, "peak": 157.3, "unit": "TFLOP/s",
                "frac": 0.858, "traffic": None}
    cpu = {"value": 20.0, "unit": "M CC-samples/s", "cores": 16, "kind": "port", "sample": "stub"}
    bp = {"metric": "grid-points x samples / s", "value": world * 1.5e12, "ms_per_step": 145.0,
          "roofline": {"bound": "lds-gather", "achieved": 120.0, "peak": 157.3, "unit": "TB/s", "frac": 0.76},
          "cpu_baseline": {"value": 1.1e9, "cores": 16, "kind": "port"}}
    ranks_info = None
    if world > 1:
        ranks_info = {"rccl_ranks": world, "backend": "nccl", "self_launched": True,
                      "ranks": [{"rank": r, "device": r, "name": "stub", "pci_bus_id": f"0000:{r:02x}:00.0", "pid": 100 + r}
                                for r in range(world)]}
        per_rank = [{"rank": r, "mf_ms_per_step": 990.0 + r, "mf_kernel_ms": 985.0 + r, "mf_frac": 0.86 - 0.001 * r,
                     "bp_ms_per_step": 146.0 + r, "bp_kernel_ms": 144.0 + r, "bp_frac": 0.76} for r in reversed(range(world))]
        bench.merge_rank_stats(ranks_info, per_rank)
    return bench.assemble_line(args=args, world=world, n_gpus=world, mf_value=4360.0 * world, mf_dt=2.97,
                               dims=(500, 20, 3, 256, 8_640_000), peak=0.93, roofline=roofline, cpu=cpu, e2e=None,
                               mf_shapes=None, bp_obj=bp, detect={"planted": 2500}, shares=None, compat=None,
                               split16=None, ranks_info=ranks_info)


def test_the_n8_line_has_every_key_of_the_n1_line():
    """the first 8-GPU run must not be the first time the N > 1 line is assembled.  Both lines
    from the same function on stubbed per-rank numbers: same keys, JSON-serialisable, `roofline` and `cpu_baseline`
    present at N = 8, every rank's kernel time and fraction in `ranks`, matched by rank whatever order the
    gather returned them in."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    l1, l8 = _stub_line(bench, 1), _stub_line(bench, 8)
    assert set(l1) == set(l8)
    for key in bench.LINE_KEYS:
        assert key in l8, key
    assert l8["roofline"]["frac"] and l8["cpu_baseline"]["cores"] == 16 and l8["bp"]["cpu_baseline"]
    assert l8["n_gpus"] == 8 and l8["scaling"] == "weak" and l8["higher_is_better"] is True and l8["vs_baseline"] is None
    assert "templates sharded x8" in l8["config"]["parallelism"] and l1["config"]["parallelism"] == "single GPU"
    ranks = l8["ranks"]["ranks"]
    assert [r["rank"] for r in ranks] == list(range(8))
    for r in ranks:
        assert r["mf_kernel_ms"] == 985.0 + r["rank"] and r["bp_kernel_ms"] == 144.0 + r["rank"]
        assert 0 < r["mf_frac"] < 1 and 0 < r["bp_frac"] < 1 and r["pci_bus_id"]
    assert l8["ranks"]["mf_kernel_ms_spread"] == {"min": 985.0, "max": 992.0, "slowest_over_fastest": round(992.0 / 985.0, 4)}
    assert l1["ranks"] is None
    json.loads(json.dumps(l8))
