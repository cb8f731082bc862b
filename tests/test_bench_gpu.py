"""bench.py end to end on the GPU, as the driver runs it (a child process, a few steps): ONE JSON line on stdout with the contract's
keys, the roofline object timed by the kernels of the timed replays, every parameter group moved; and the N > 1 form rehearsed with
a single-rank RCCL group (collectives inside the hipGraph, the replica check at the end)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline", "cpu_baseline")


def _bench(extra_env, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]            # ONE line on stdout, whatever the libraries print
    return json.loads(lines[0])


def test_default_line_has_the_contract(dev):
    d = _bench({}, "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-pmc")
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["dtype"] == "f32" and d["scaling"] == "weak"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and "broken" not in d
    assert "configs[1]" in d["config"]["workload"] and d["config"]["grid"] == 160 and d["config"]["rays_per_gpu"] == 4096
    assert "hipGraph" in d["config"]["step_mode"]
    assert 100 < d["value"] < 2000 and abs(d["ms_per_step"] * d["value"] / 1e3 - d["config"]["inbbox_samples_per_step_per_gpu"] / 1e6) \
        < 0.02 * d["config"]["inbbox_samples_per_step_per_gpu"] / 1e6
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.3 < r["frac"] < 1.0 and r["traffic"] is None            # (--no-pmc: never a number from another run)
    assert "s_memrealtime" in r["timing"] and "timed region itself" in r["timing"]
    # every MLP launch of every timed replay was stamped: 6 forward chains, 6 backward chains (the two narrow products ride in them
    # as side layers), 6 weight-gradient launches
    assert sorted(c["launches"] for c in r["chains"].values()) == [6, 6, 6] and any("k_mlp_rc2" in k for k in r["chains"])
    assert r["launches"] == 18


def test_forced_single_rank_group_runs_the_captured_exchange(dev):
    d = _bench({"FGS_FORCE_DIST": "1"}, "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-pmc")
    assert "gradient exchange included" in d["config"]["step_mode"] and "broken" not in d
    assert d["config"]["replicas_in_sync"] is True and d["config"]["replica_digest_spread"] == 0.0
    assert d["config"]["k0_exchange"]["bytes_per_step"] > 0


def test_supervised_run_falls_back_to_fresh_conservative_children(dev):
    """The N > 1 safety net on the GPU box (single-rank RCCL group): the supervisor -- which never touches the GPU -- sees its
    worker die after HIP and RCCL initialisation, starts a fresh one with the conservative switches and forwards ITS line."""
    d = _bench({"FGS_FORCE_DIST": "1", "FGS_BENCH_SUPERVISE": "force", "FGS_BENCH_FAIL_ATTEMPT1": "1", "FGS_BENCH_ATTEMPT_DEADLINE_S": "200"},
               "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-pmc")
    cfg = d["config"]
    assert cfg["launcher"]["attempt_used"] == 2 and "exit 17" in cfg["fallback_reason"]
    assert "eager launches" in cfg["step_mode"] and cfg["replicas_in_sync"] is True and "broken" not in d
    assert d["value"] > 50 and d["roofline"]["frac"] > 0.3
