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
    assert out["config"]["launcher"]["attempt_used"] == 1 and "fallback_reason" not in out["config"]
    assert out["config"]["step_mode"] == "dry (graph)" and out["config"]["one_comm"] is None


def _lines(r):
    return [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_every_attempt_failing_still_ends_with_one_line_and_a_nonzero_status():
    r = _run({"FGS_BENCH_DRY": "no_such_backend", "FGS_BENCH_ATTEMPT_DEADLINE_S": "60"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    out = _lines(r)
    assert len(out) == 1 and out[0]["value"] is None and "broken" in out[0]
    att = out[0]["config"]["launcher"]["attempts"]
    assert [a["ok"] for a in att] == [False, False] and att[0]["ranks"] and att[0]["ranks"][0]["exit"] not in (0, None)


def test_a_rank_that_dies_in_the_first_attempt_sends_all_ranks_to_the_conservative_form():
    """the first set of children is made to fail by an env switch; exactly one line arrives, from a FRESH set of
    children started with the conservative switches, and it says so."""
    import time
    t0 = time.time()
    r = _run({"FGS_BENCH_DRY": "gloo", "FGS_BENCH_DRY_FAIL": "exit", "FGS_BENCH_ATTEMPT_DEADLINE_S": "60"},
             "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    out = _lines(r)
    assert len(out) == 1, r.stdout
    cfg = out[0]["config"]
    assert out[0]["n_gpus"] == 2 and out[0]["sum_units"] == 200.0
    assert cfg["step_mode"] == "dry (eager)" and cfg["one_comm"] == "1" and cfg["sdf_tune"] == "0"      # DESIGN 8.4's switches
    assert cfg["launcher"]["attempt_used"] == 2 and [a["ok"] for a in cfg["launcher"]["attempts"]] == [False, True]
    assert "attempt 1" in cfg["fallback_reason"] and "exit 17" in cfg["fallback_reason"] and "failing on purpose" in cfg["fallback_reason"]
    assert time.time() - t0 < 120


def test_a_rank_that_hangs_is_cut_at_the_deadline_and_the_fallback_reports():
    import time
    t0 = time.time()
    r = _run({"FGS_BENCH_DRY": "gloo", "FGS_BENCH_DRY_FAIL": "hang", "FGS_BENCH_DRY_FAIL_RANK": "0", "FGS_BENCH_ATTEMPT_DEADLINE_S": "12"},
             "--gpus", "2", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    out = _lines(r)
    assert len(out) == 1, r.stdout
    cfg = out[0]["config"]
    assert cfg["launcher"]["attempt_used"] == 2 and cfg["launcher"]["attempts"][0]["hit_deadline_s"] == 12.0
    assert "deadline" in cfg["fallback_reason"] and cfg["step_mode"] == "dry (eager)"
    assert time.time() - t0 < 150


def test_under_an_external_launcher_each_rank_supervises_a_child():
    """The driver's own form: torch.distributed.run starts bench.py per rank (WORLD_SIZE set): the launched process supervises, the
    rank is its child with a rendezvous of its own."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"FGS_BENCH_DRY": "gloo", "FGS_BENCH_DRY_FAIL": "exit", "FGS_BENCH_ATTEMPT_DEADLINE_S": "60", "OMP_NUM_THREADS": "2"})
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
                        "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _lines(r)
    assert len(out) == 1 and out[0]["steps"] == 4 and out[0]["config"]["launcher"]["attempt_used"] == 2


def test_single_rank_does_not_launch():
    r = _run({"FGS_BENCH_DRY": "gloo"}, "--gpus", "1", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 1
