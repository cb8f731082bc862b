"""bench.py starts its own ranks (`python bench.py --gpus N` with no launcher around it): CPU rehearsal of the launcher
branch and of the cross-rank reporting protocol through the gloo dry mode (no device work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=300)


def test_launcher_starts_two_ranks_and_reports_world_size():
    r = _run({"FGS_BENCH_DRY": "gloo"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["max_elapsed"] == 2.0 and out["sum_units"] == 200.0     # MAX / SUM over both ranks arrived


def test_launcher_propagates_child_failure():
    r = _run({"FGS_BENCH_DRY": "no_such_backend"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0


def test_single_rank_does_not_launch():
    r = _run({"FGS_BENCH_DRY": "gloo"}, "--gpus", "1", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 1
