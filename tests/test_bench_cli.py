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
