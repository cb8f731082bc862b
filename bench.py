#!/usr/bin/env python
"""Headline benchmark: M ray-samples/s of one full training step of the voxel-NeRF hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic 160^3 SDF (1 ch) + feature grid (12 ch), fine-stage
model (rgbnet 106->256x3->256, refnet 307->256x3->3), 4096 rays per GPU per step drawn from 8 blender-style cameras,
global_step=1000.  One timed step = forward_fine + the fine-stage losses + backward + RCCL gradient averaging (N > 1)
+ dense SDF TV-add-grad + MaskedAdam step (sdf dense, k0 masked, MLPs dense) -- a complete training iteration
(model/nerf_training.py:300-373), all in fp32.  Unit of work = one in-bbox sample point emitted by
sample_pts_on_rays.  Inputs are resident in HBM before the timed region.  Weak scaling: per-GPU rays fixed.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

GRID = 160
RAYS_PER_GPU = 4096
N_BATCHES = 8          # distinct ray batches cycled through, so consecutive steps touch different voxels
GLOBAL_STEP = 1000
STEP_STATS = {"survivors": 0}   # survivors (samples that reach the MLPs) summed over the timed steps of this rank


def make_optimizer(model):
    """Stage param groups (config/shiny_blender.py:184-187,214-216; model/nerf_training.py:9-37)."""
    from fgs_nerf_amd.adam import MaskedAdam
    groups = [{'params': [model.sdf.grid], 'lr': 0.005, 'name': 'sdf', 'skip_zero_grad': False}]
    if model.rgbnet is not None:
        groups.append({'params': list(model.rgbnet.parameters()), 'lr': 1e-3, 'name': 'rgbnet', 'skip_zero_grad': False})
    groups.append({'params': list(model.refnet.parameters()), 'lr': 1e-3, 'name': 'refnet', 'skip_zero_grad': False})
    # k0 last: on N > 1 GPUs its gradient exchange is the longest, and the other updates run while it finishes
    groups.append({'params': [model.k0.grid], 'lr': 0.1, 'name': 'k0', 'skip_zero_grad': True})
    return MaskedAdam(groups, betas=(0.9, 0.99))


def make_batch(b: int, rank: int, dev, n_rays=None):
    """Resident ray batch `b` of `rank`: (rays_o, rays_d, viewdirs, target), each [n_rays, 3] on `dev`."""
    from fgs_nerf_amd import synth
    n_rays = RAYS_PER_GPU if n_rays is None else n_rays
    ro, rd, vd = synth.random_rays(n_rays, seed=synth.SEED + 97 * b + 10007 * rank)
    target = torch.rand(n_rays, 3, generator=torch.Generator().manual_seed(b + 1000 * rank))
    return tuple(t.to(dev).contiguous() for t in (ro, rd, vd, target))


def survivor_capacity(seen: int) -> int:
    """Rows of the survivor buffers of a sync-free / captured step: 1.5 x the largest count seen, rounded up to 4096."""
    return (int(1.5 * max(seen, 16384)) + 4095) // 4096 * 4096


def tv_args(n_rays_global: int):
    """Fine stage: CUDA-side TV on the sdf grid only (weight_tv_k0 = 0), dense (model/nerf_training.py:353-371)."""
    return (0.01 * 0.1 / n_rays_global, True)


def build_captured(model, opt, capacity: int, n_iters: int, n_rays_global: int, averager=None, n_rays=None):
    """The captured form of train_step() below: the step bench.py times (tests/test_bench_step_parity_gpu.py builds it through
    this same function and compares one replay with the oracle)."""
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.graph_step import CapturedFineStep
    return CapturedFineStep(model, opt, synth.FINE_LOSS if model.stage == 'fine' else synth.COARSE_LOSS, synth.RENDER_KWARGS,
                            RAYS_PER_GPU if n_rays is None else n_rays, n_iters=n_iters,
                            global_step_of=lambda it: GLOBAL_STEP, lr_of=lambda it, g: g['lr'],
                            tv=tv_args(n_rays_global), capacity=capacity, averager=averager)


def train_step(model, opt, averager, batch, n_rays_global):
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.losses import fused_render_losses
    ro, rd, vd, target = batch
    loss_cfg = synth.FINE_LOSS if model.stage == 'fine' else synth.COARSE_LOSS
    from fgs_nerf_amd.fused import set_loss_spec
    set_loss_spec(model, target, loss_cfg)        # (fine stage: compositing + losses + their gradients become one launch)
    res = model(ro, rd, vd, global_step=GLOBAL_STEP, **synth.RENDER_KWARGS)
    # nerf_training.py:308-327
    loss = fused_render_losses(res, target, loss_cfg, model)
    pts = res.get('survivor_pts') if hasattr(res, 'get') else None
    if pts is not None:   # every k0 gradient of this step comes from trilinear lookups at the survivors
        # (sync-free: `pts` has CAPACITY rows, the rows that count are behind a device-side counter)
        averager.hint_touched(model.k0.grid, pts, model.xyz_min, model.xyz_max, count_ptr=res.get('survivor_count_ptr'))
    opt.zero_grad(set_to_none=True)
    seed = STEP_STATS.get("seed")
    if seed is None or seed.device != loss.device:
        from fgs_nerf_amd.losses import register_unit_seed
        seed = STEP_STATS["seed"] = register_unit_seed(torch.ones((), dtype=torch.float32, device=loss.device))
    loss.backward(seed)             # (the gradient seed given: one fill launch less than autograd's implicit ones_like)
    averager.average()
    # fine stage: CUDA-side TV on the sdf grid only (weight_tv_k0 = 0), dense (nerf_training.py:353-371)
    model.sdf_total_variation_add_grad(*tv_args(n_rays_global))
    opt.step()
    if res.get('survivor_count_ptr') is None if hasattr(res, 'get') else True:
        STEP_STATS["survivors"] += int(res['weights'].shape[0])     # a host number already (the forward read it)
        STEP_STATS["max_survivors"] = max(STEP_STATS.get("max_survivors", 0), int(res['weights'].shape[0]))
    return loss


def touched_voxels(model, batches) -> float:
    """k0 voxels holding a trilinear corner of a survivor point, mean over the resident batches (what MaskedAdam's voxel update
    visits): a render without gradients per batch, fgs_brick_masks_pts into a scratch byte-per-voxel buffer, one sum."""
    import ctypes
    from fgs_nerf_amd import synth
    from fgs_nerf_amd._lib import call, ptr, stream
    _, _, X, Y, Z = model.k0.grid.shape
    flags = torch.zeros(((X + 3) // 4) * ((Y + 3) // 4) * ((Z + 3) // 4) * 64, dtype=torch.uint8, device=model.k0.grid.device)
    lo = (ctypes.c_float * 3)(*[float(v) for v in model.xyz_min.flatten().tolist()])
    hi = (ctypes.c_float * 3)(*[float(v) for v in model.xyz_max.flatten().tolist()])
    total = 0
    with torch.no_grad():
        for b in batches:
            res = model(b[0], b[1], b[2], global_step=GLOBAL_STEP, **synth.RENDER_KWARGS)
            pts = res['survivor_pts'][:res['weights'].shape[0]].contiguous()
            flags.zero_()
            call("fgs_brick_masks_pts", ptr(pts), pts.shape[0], lo, hi, X, Y, Z, ptr(flags), None, stream())
            total += int(flags.sum())
    return total / max(len(batches), 1)


def cpu_baseline(n_rays=4096, iters=10, warm=3):
    """The oracle (CPU port of the reference path: packed sampling -> F.grid_sample -> NeuS alpha -> early-stop scan ->
    nn.Linear MLPs -> index_add_ -> backward) on a bounded sample of the same workload, host cores of this box."""
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.losses import render_losses
    from oracle import oracle as O
    # the GPU box grants a 16-core share per GPU; more threads than that only oversubscribe it
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, avail)))
    model = synth.build_model(GRID, synth.FINE_MODEL, fused=False)
    P = synth.oracle_params(model)
    leaves = [P['sdf'], P['k0']] + [t for net in (P['rgbnet'], P['refnet']) for wb in net for t in wb]
    for t in leaves:
        t.requires_grad_(True)
    ro, rd, vd = synth.random_rays(n_rays, seed=synth.SEED + 1000)
    target = torch.rand(n_rays, 3, generator=torch.Generator().manual_seed(3))
    times, n_in = [], 0
    for it in range(warm + iters):       # 3 warm-up + 10 timed iterations of one full 4096-ray batch
        for t in leaves:
            t.grad = None
        t0 = time.perf_counter()
        res = O.forward_fine(P, ro, rd, vd, global_step=GLOBAL_STEP, near=2.0, stepsize=0.5, bg=1)
        render_losses(res, target, synth.FINE_LOSS).backward()
        if it >= warm:
            times.append(time.perf_counter() - t0)
        n_in = res['n_inbbox']
    med = sorted(times)[len(times) // 2]
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    return {"value": round(n_in / med / 1e6, 4), "unit": "M ray-samples/s", "cores": torch.get_num_threads(),
            "cpu_model": cpu_model,
            "kind": "port", "sample": f"{n_rays} rays x ({warm} warm-up + {iters} timed) fwd+bwd iterations of the same {GRID}^3 workload "
                                      f"({n_in} in-bbox samples/iter, median {med * 1e3:.0f} ms), torch CPU + C oracle"}


# HBM traffic of the MLP matrix-core kernels, measured live: two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE need more TCC
# counter slots than one pass has) over a short eager run of this same workload, started as child processes before this
# process touches the GPU.  Corrections as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE is in KB and
# tallies a 128-byte request at 64 bytes (x 2); WRITE_SIZE (KB) is exact for 16-byte-per-lane stores and float atomics.
PMC_KERNELS = (("k_mlp_rc2<false", "k_mlp_rc2 forward chain"), ("k_mlp_rc2<true", "k_mlp_rc2 backward chain"),
               ("k_mlp_rc<false", "k_mlp_rc forward chain"), ("k_mlp_rc<true", "k_mlp_rc backward chain"),
               ("k_mlp_wgrad", "k_mlp_wgrad"), ("k_mlp_fwd", "k_mlp_fwd"), ("k_linear_bwd", "k_linear_bwd"), ("k_gemm", "k_gemm"))


def pmc_traffic_live(args, timeout_s=180):
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"error": "rocprofv3 not found"}
    here = os.path.abspath(__file__)
    out = {}
    root = tempfile.mkdtemp(prefix="fgs_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(root, counter)
            cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, here,
                   "--pmc-child", "--no-cpu-baseline", "--mode", "eager", "--steps", "4", "--warmup", "2",
                   "--stage", args.stage, "--grid", str(args.grid), "--rays", str(args.rays)]
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return {"error": f"rocprofv3 --pmc {counter} pass timed out"}
            files = sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True))
            if r.returncode != 0 or not files:
                return {"error": f"rocprofv3 --pmc {counter} pass failed (exit {r.returncode})"}
            acc = {}
            with open(files[-1]) as f:
                for row in csv.DictReader(f):
                    if row.get("Counter_Name") != counter:
                        continue
                    name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
                    for key, label in PMC_KERNELS:
                        if key in name:
                            a = acc.setdefault(label, {})
                            did = row.get("Dispatch_Id", len(a))
                            a[did] = a.get(did, 0.0) + float(row["Counter_Value"])     # (one row per XCD on some versions)
                            break
            out[counter] = {k: (len(v), sum(v.values()) / len(v)) for k, v in acc.items()}
    finally:
        shutil.rmtree(root, ignore_errors=True)
    kernels = {}
    for k in sorted(set(out["FETCH_SIZE"]) & set(out["WRITE_SIZE"])):
        rd, wr = out["FETCH_SIZE"][k][1] * 1024 * 2, out["WRITE_SIZE"][k][1] * 1024
        kernels[k] = {"launches_sampled": out["FETCH_SIZE"][k][0], "hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr),
                      "bytes_per_launch": round(rd + wr)}
    return {"kernels": kernels,
            "method": "live: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two child passes of 6 eager steps of this "
                      "workload); bytes = FETCH_SIZE KB x 1024 x 2 (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE KB x 1024"}


def rccl_graph_selftest(dev, timeout_s: float = 20.0) -> bool:
    """Before the N > 1 step is captured with its collectives inside: can THIS stack replay a hipGraph that holds an RCCL
    all-reduce between these ranks?  A tiny all-reduce on a communicator of its own (a wedged replay must not block the
    communicators the run uses) is captured, replayed and waited for with a timeout; the ranks then agree (MIN) on the outcome.
    False sends the run to the eager form -- a number from the slower path beats a hang."""
    import torch.distributed as dist
    ok = 1
    try:
        grp = dist.new_group(backend="nccl")
        t = torch.ones(4096, device=dev)
        dist.all_reduce(t, group=grp)                      # communicator set-up happens eagerly
        torch.cuda.synchronize()
        t.fill_(1.0)
        side = torch.cuda.Stream(device=dev)
        graph = torch.cuda.CUDAGraph()
        time.sleep(0.35)                                   # (the watchdog's list drains: see CapturedFineStep.capture)
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                dist.all_reduce(t, group=grp)
            graph.replay()
            done = torch.cuda.Event()
            done.record(side)
        t0 = time.perf_counter()
        while not done.query():
            if time.perf_counter() - t0 > timeout_s:
                print("[bench] RCCL-in-hipGraph self-test: the replay did not complete", file=sys.stderr, flush=True)
                ok = 0
                break
            time.sleep(0.005)
        if ok and not bool((t == float(dist.get_world_size())).all()):
            print("[bench] RCCL-in-hipGraph self-test: wrong values after the replay", file=sys.stderr, flush=True)
            ok = 0
    except Exception as e:          # noqa: BLE001
        print(f"[bench] RCCL-in-hipGraph self-test failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


# ---- N > 1: a run that cannot end without a line -------------------------------------------------------------------------
# The process a launcher (the driver's torch.distributed.run, or launch_ranks below) starts per GPU is a SUPERVISOR: it never
# initialises HIP, talks to its peers over gloo (CPU) and runs the real rank -- the WORKER -- as a child process with a
# rendezvous port of its own.  Attempt 1 is the default form (captured step, collectives inside the hipGraph, two
# communicators, the sdf exchange tuned on the node).  If any worker exits non-zero, prints no line or passes the deadline, every
# supervisor kills its worker's process group and all of them start a FRESH set of workers in the conservative form
# (DESIGN section 8.4's switch list), and the line says so (`config.launcher`, `config.fallback_reason`).  Nothing that has
# touched the GPU is ever re-exec'ed or reused: children only.
ATTEMPTS = (
    {"label": "default", "argv": [], "env": {}},
    {"label": "conservative: host-driven step (--mode eager: no collective inside a hipGraph), one communicator "
              "(FGS_DIST_ONE_COMM=1), untuned sdf exchange (FGS_SDF_TUNE=0)",
     "argv": ["--mode", "eager"], "env": {"FGS_DIST_ONE_COMM": "1", "FGS_SDF_TUNE": "0"}},
)


def _free_ports(n):
    import socket
    socks = [socket.socket() for _ in range(n)]
    try:
        for s_ in socks:
            s_.bind(("127.0.0.1", 0))
        return [s_.getsockname()[1] for s_ in socks]
    finally:
        for s_ in socks:
            s_.close()


def _kill_group(proc, grace_s=5.0):
    """SIGTERM, then SIGKILL, to the worker's whole process group (it was started as a session leader)."""
    import signal
    if proc.poll() is not None:
        return
    for sig, wait in ((signal.SIGTERM, grace_s), (signal.SIGKILL, 10.0)):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass
        try:
            proc.wait(timeout=wait)
            return
        except Exception:      # noqa: BLE001  (subprocess.TimeoutExpired)
            continue


def _tail(path, n_lines=6, n_chars=600):
    try:
        with open(path, "rb") as f:
            f.seek(0, 2)
            f.seek(max(0, f.tell() - 8192))
            txt = f.read().decode("utf-8", "replace")
    except OSError:
        return ""
    return "\n".join(txt.strip().splitlines()[-n_lines:])[-n_chars:]


def _json_line(path):
    try:
        with open(path) as f:
            lines = [ln for ln in f.read().splitlines() if ln.strip()]
    except OSError:
        return None
    for ln in reversed(lines):
        if ln.lstrip().startswith("{"):
            try:
                return json.loads(ln)
            except ValueError:
                continue
    return None


def supervise_rank(args, argv) -> int:
    """See the block comment above.  Returns the exit status of this (supervisor) process: 0 when a line with a number was
    written, 3 when both attempts failed (rank 0 still writes a line, `value` null, saying why)."""
    import datetime
    import subprocess
    import tempfile
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    deadline_s = float(os.environ.get("FGS_BENCH_ATTEMPT_DEADLINE_S", "240"))
    if "WORLD_SIZE" not in os.environ:       # (forced single-rank rehearsal: no launcher set the rendezvous up)
        os.environ.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_PORT": str(_free_ports(1)[0])})
    if world != args.gpus:
        print(f"--gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # stdout carries ONE JSON line: gloo prints its connection banner there, so file descriptor 1 points at stderr from here on and
    # the line goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=max(180.0, 2 * deadline_s)))
    box = [_free_ports(len(ATTEMPTS)) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ports = box[0]
    tmp = tempfile.mkdtemp(prefix=f"fgs_bench_r{rank}_", dir=os.environ.get("TMPDIR", "/tmp"))
    history, line, used = [], None, None
    for k, att in enumerate(ATTEMPTS, start=1):
        env = {k_: v for k_, v in os.environ.items() if not k_.startswith("TORCHELASTIC_")}
        env.update(att["env"])
        env.update({"FGS_BENCH_WORKER": "1", "FGS_BENCH_ATTEMPT": str(k), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(ports[k - 1]),
                    "RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": os.environ.get("LOCAL_RANK", str(rank))})
        env.setdefault("FGS_BENCH_PG_TIMEOUT_S", str(int(min(120.0, max(10.0, deadline_s / 2)))))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out_p, err_p = os.path.join(tmp, f"a{k}.out"), os.path.join(tmp, f"a{k}.err")
        with open(out_p, "wb") as fo, open(err_p, "wb") as fe:
            proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv, *att["argv"]], env=env, stdout=fo, stderr=fe,
                                    start_new_session=True)
        t0 = time.perf_counter()
        while True:
            rc = proc.poll()
            mine = torch.tensor([int(rc is not None and rc != 0), int(rc is None), int(time.perf_counter() - t0 > deadline_s)],
                                dtype=torch.int32)
            dist.all_reduce(mine, op=dist.ReduceOp.MAX)          # one decision for all supervisors, every half second
            any_failed, any_running, any_late = (int(v) for v in mine.tolist())
            if any_failed or not any_running or any_late:
                break
            time.sleep(0.5)
        rc = proc.poll()
        _kill_group(proc)
        got = _json_line(out_p) if rank == 0 else None
        # a line with a number means the timed region and the cross-rank reductions behind it completed on every rank: it
        # stands even if some rank then died or hung in teardown (its exit status is recorded next to it)
        ok = [bool(got is not None and got.get("value") is not None) if rank == 0 else None]
        dist.broadcast_object_list(ok, src=0)
        with open(err_p, "rb") as fe:            # the worker's stderr, after the fact (kept in a file so that its tail can be quoted)
            sys.stderr.write(fe.read().decode("utf-8", "replace"))
            sys.stderr.flush()
        report = [None] * world
        dist.all_gather_object(report, {"rank": rank, "exit": rc, "stderr_tail": "" if rc == 0 else _tail(err_p)})
        history.append({"attempt": k, "form": att["label"], "ok": ok[0], "hit_deadline_s": deadline_s if any_late and any_running else None,
                        "ranks": [r for r in report if r["exit"] != 0][:3]})
        if ok[0]:
            line, used = got, k
            break
        if rank == 0:
            print(f"[bench] attempt {k} ({att['label']}) did not produce a line: {history[-1]}", file=sys.stderr, flush=True)
    if rank == 0:
        if line is None:
            line = {"metric": f"M ray-samples/sec (fwd+bwd), {args.grid}^3 grid, {args.rays}-ray batch", "value": None,
                    "unit": "M ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": "configs[1]"}, "broken": "every attempt of the multi-GPU run failed (config.launcher)"}
        cfg = line.setdefault("config", {})
        cfg["launcher"] = {"supervised": True, "attempt_used": used, "attempt_deadline_s": deadline_s, "attempts": history}
        if used is not None and used > 1:
            first = history[0]
            why = ("deadline of %.0f s passed" % deadline_s) if first["hit_deadline_s"] else "a rank exited non-zero or printed no line"
            cfg["fallback_reason"] = (f"attempt 1 ({first['form']}) failed: {why}; "
                                      + " | ".join(f"rank {r['rank']} exit {r['exit']}: {r['stderr_tail'][-300:]}" for r in first["ranks"]))
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return 0 if used is not None else 3


def launch_ranks(n_gpus: int, argv) -> int:
    """`python bench.py --gpus N` with no launcher around it: one supervisor per GPU through torch.distributed.run (rendezvous
    on 127.0.0.1, a free port).  This parent never initialises HIP; it gives the whole run a deadline of its own (both
    attempts + start-up) and kills the launcher's process group if that passes.  Returns the launcher's exit status."""
    import subprocess
    port = _free_ports(1)[0]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    overall = len(ATTEMPTS) * float(os.environ.get("FGS_BENCH_ATTEMPT_DEADLINE_S", "240")) + 180.0
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait(timeout=overall)
    except subprocess.TimeoutExpired:
        print(f"[bench] the launcher did not finish within {overall:.0f} s: killing it", file=sys.stderr, flush=True)
        _kill_group(proc)
        return 124


def replica_digest_spread(digests) -> float:
    """Largest relative difference, over the parameter groups, between the ranks' digests (one scalar per group): MIN- and
    MAX-reduced over the process group.  0.0 = the replicas are bit-identical as far as the digests can tell."""
    import torch.distributed as dist
    dg = torch.stack([d.reshape(()) for d in digests]).to(torch.float64)
    lo, hi = dg.clone(), dg.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return float(((hi - lo) / hi.abs().clamp_min(1e-300)).max().item())


def dry_rank(args) -> int:
    """FGS_BENCH_DRY=gloo: rehearse the launcher, the supervisors and the cross-rank reporting protocol on CPU (no device work, no
    timing claim): process group, barrier, MAX / SUM reductions, rank 0 prints the JSON line with the world size it saw.
    FGS_BENCH_DRY_FAIL=exit|hang makes rank FGS_BENCH_DRY_FAIL_RANK (default: the last) fail that way in attempt 1 only."""
    import datetime
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    fail = os.environ.get("FGS_BENCH_DRY_FAIL") if os.environ.get("FGS_BENCH_ATTEMPT", "1") == "1" else None
    fail_rank = int(os.environ.get("FGS_BENCH_DRY_FAIL_RANK", str(world - 1)))
    if world > 1:
        dist.init_process_group(backend=os.environ["FGS_BENCH_DRY"],
                                timeout=datetime.timedelta(seconds=float(os.environ.get("FGS_BENCH_PG_TIMEOUT_S", "120"))))
        dist.barrier()
    if fail and rank == fail_rank:
        if fail == "hang":
            print(f"[dry] rank {rank}: hanging on purpose", file=sys.stderr, flush=True)
            time.sleep(3600)
        print(f"[dry] rank {rank}: failing on purpose", file=sys.stderr, flush=True)
        os._exit(17)
    stats = torch.tensor([1.0 + rank, 100.0], dtype=torch.float64)
    tmax, ssum = stats[:1].clone(), stats[1:].clone()
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(ssum, op=dist.ReduceOp.SUM)
    seen = dist.get_world_size() if world > 1 else 1
    if world != args.gpus:
        print(f"--gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    if rank == 0:
        print(json.dumps({"metric": "dry run (launcher rehearsal, no device work)", "value": 0.0, "n_gpus": seen,
                          "steps": args.steps, "warmup": args.warmup, "max_elapsed": float(tmax), "sum_units": float(ssum),
                          "config": {"step_mode": f"dry ({args.mode})", "one_comm": os.environ.get("FGS_DIST_ONE_COMM"),
                                     "sdf_tune": os.environ.get("FGS_SDF_TUNE")}}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--composed", action="store_true", help="operator-at-a-time HIP path instead of the fused kernels")
    ap.add_argument("--stage", choices=["fine", "coarse"], default="fine",
                    help="fine = the headline workload (configs[1]); coarse = configs[2]'s forward_coarse step at the same size")
    ap.add_argument("--mode", choices=["graph", "eager", "eager-sync"], default="graph",
                    help="graph (default, 1 GPU, fused path): the whole step is one hipGraph replay, nothing of it reaches "
                         "the host; eager: Python enqueues every launch and reads the survivor count back once per step")
    ap.add_argument("--grid", type=int, default=GRID,
                    help="grid side (default 160 = the headline config; 320 = the per-GPU shape of configs[4], 128 = configs[0])")
    ap.add_argument("--rays", type=int, default=RAYS_PER_GPU,
                    help="rays per GPU per step (default 4096 = the headline config; 8192 = the reference's own fine-stage batch, "
                         "config/shiny_blender.py:182 -- a secondary line, never the headline)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 counter passes behind roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--clock-warmup-ms", type=float, default=0.0,
                    help="captured step: milliseconds of unrelated matrix work in front of the W warm-up steps (an experiment: the "
                         "early steps are not slower because of the clock; recorded in the line when used)")
    args = ap.parse_args()
    globals()["GRID"] = args.grid
    globals()["RAYS_PER_GPU"] = args.rays

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as fresh child processes, BEFORE anything
    # in this process touches the GPU (the parent never initialises HIP and never replaces itself with another program).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    # N > 1 under a launcher (the driver's torch.distributed.run, or ours): this process supervises, a child is the rank
    # (FGS_BENCH_SUPERVISE=force: also at --gpus 1, the one-GPU rehearsal of this machinery with FGS_FORCE_DIST=1; =0: off)
    sup = os.environ.get("FGS_BENCH_SUPERVISE", "1")
    if (args.gpus > 1 and sup != "0" or sup == "force") and os.environ.get("FGS_BENCH_WORKER") != "1":
        sys.exit(supervise_rank(args, sys.argv[1:]))
    if os.environ.get("FGS_BENCH_DRY"):
        sys.exit(dry_rank(args))

    # stdout carries ONE JSON line.  Libraries write there too (RCCL prints its version banner and warnings with printf at
    # communicator creation, whatever NCCL_DEBUG_FILE says): from here on file descriptor 1 points at stderr, and the line
    # is written to the saved original descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    pmc = None
    if (args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not (args.pmc_child or args.no_pmc or args.composed)
            and os.environ.get("FGS_FORCE_DIST") != "1"):
        pmc = pmc_traffic_live(args)          # child processes; this process has not touched the GPU yet

    # host side of this path is one Python thread + the autograd thread; the box grants a 16-CPU quota per GPU and
    # torch would otherwise spawn one OpenMP worker per visible core (256) for the synthetic-scene setup
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // max(1, int(os.environ.get("WORLD_SIZE", "1"))))))

    import torch.distributed as dist
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.dist import GradAverager
    from fgs_nerf_amd.ops import render_utils_cuda

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    force_dist = world == 1 and os.environ.get("FGS_FORCE_DIST") == "1"   # rehearse the RCCL exchange on one GPU
    if world > 1 or force_dist:
        # RCCL's internal streams on their own (high-priority) hardware queues: HIP maps streams onto a few queues, and a
        # collective that lands on the queue of the compute stream would hold up the kernels behind it
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL over xGMI.  The timeout bounds rendezvous AND every collective (the watchdog aborts the process past it): a rank
        # that lost its peers ends within two minutes and the supervisor above starts the conservative attempt
        import datetime
        dist.init_process_group(backend="nccl", device_id=dev,
                                timeout=datetime.timedelta(seconds=float(os.environ.get("FGS_BENCH_PG_TIMEOUT_S", "120"))))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("FGS_BENCH_FAIL_ATTEMPT1") and os.environ.get("FGS_BENCH_ATTEMPT") == "1":
        # test switch: a rank that dies AFTER it has initialised the GPU and joined the group (tests/test_bench_gpu.py)
        torch.zeros(1, device=dev)
        print(f"[bench] rank {rank}: failing on purpose (FGS_BENCH_FAIL_ATTEMPT1)", file=sys.stderr, flush=True)
        os._exit(17)

    model = synth.build_model(GRID, synth.FINE_MODEL if args.stage == "fine" else synth.COARSE_MODEL, device=dev,
                              fused=False if args.composed else None)
    opt = make_optimizer(model)
    averager = GradAverager(model.parameters(), force=force_dist)
    averager.attach(model)          # gradients are handed to the exchange from inside the backward pass (N > 1)
    averager.attach_optimizer(opt)  # ... and the optimizer waits for k0's exchange only when it reaches k0
    early_adam = os.environ.get("FGS_EARLY_ADAM", "1")
    if not args.composed and early_adam == "1":
        # k0's Adam pass is issued from inside the backward pass, right behind the feature-grid scatter (no TV on k0 in this
        # step).  N > 1: on the exchange stream, behind k0's gradient exchange.  One GPU: in place on the backward pass's own
        # stream, i.e. beside the weight-gradient launch of the side branch instead of at the end of the step behind it.
        from fgs_nerf_amd import fused
        fused.enable_early_update(model, opt, averager, inline=not (world > 1 or force_dist))
    n_global = RAYS_PER_GPU * world

    # ray batches, resident in HBM; per-batch in-bbox sample counts (the unit of work)
    batches, n_inbbox, n_total = [], [], []
    for b in range(N_BATCHES):
        batch = make_batch(b, rank, dev)
        out = render_utils_cuda.sample_pts_on_rays(batch[0], batch[1], model.xyz_min, model.xyz_max, 2.0, 1e9,
                                                   float(0.5 * model.voxel_size))
        n_inbbox.append(int((~out[1]).sum().item()))
        n_total.append(int(out[1].numel()))
        batches.append(batch)
    del out

    # Setup, not a step: one forward + backward (no optimizer update) over every resident batch, so that PyTorch's caching
    # allocator already owns blocks for the largest survivor count.  The batches differ in that count (50-64 K rows of
    # every activation) and --warmup may be shorter than the batch cycle; a first-time hipMalloc inside the timed region
    # would cost milliseconds.  (The remaining run-to-run spread, 2.32-2.56 ms/step on a noisy box, is host jitter: the
    # Python launch path needs ~2 ms of CPU per 2.3 ms step.)
    from fgs_nerf_amd.losses import fused_render_losses
    for b in (batches if args.warmup < N_BATCHES else []):     # a warm-up that covers the batch cycle needs no priming
        res = model(b[0], b[1], b[2], global_step=GLOBAL_STEP, **synth.RENDER_KWARGS)
        fused_render_losses(res, b[3], synth.FINE_LOSS if model.stage == 'fine' else synth.COARSE_LOSS, model).backward()
        opt.zero_grad(set_to_none=True)
        del res
    torch.cuda.synchronize()
    # (N > 1: the gradient exchange is part of the captured step -- RCCL collectives as graph nodes, the k0 brick exchange in its
    # device-counted form; the warm-up steps below run the host-counted exchange and thereby measure the union's brick count,
    # identical on every rank, from which the exchange capacity is derived)
    use_graph = (args.mode == "graph" and not args.composed and os.environ.get("FGS_MLP", "rc") == "rc")
    if use_graph and (world > 1 or force_dist) and not rccl_graph_selftest(dev):
        use_graph = False
        print("[bench] collectives inside a hipGraph are not usable here: running the eager form", file=sys.stderr, flush=True)
    # N > 1: the form of the sdf.grad exchange (dense all-reduce, or brick-sparse like k0's) is decided by timing both on THESE
    # ranks -- 16 MB dense against ~7 MB + the occupancy all-reduce + five extra launches at 160^3 is a question of the node's
    # all-reduce latency and bandwidth, which a one-GPU build cannot answer (FGS_SDF_TUNE=0: the shape threshold, 256^3, stays)
    sdf_tune = None
    if (world > 1 or force_dist) and use_graph and args.stage == "fine" and os.environ.get("FGS_SDF_TUNE", "1") == "1":
        sdf_tune = averager.tune_sparse_1ch(model.sdf.grid)
        if rank == 0:
            print(f"[bench] sdf.grad exchange tuned on {world} ranks: {sdf_tune}", file=sys.stderr, flush=True)
    STEP_STATS["max_survivors"] = 0
    for i in range(args.warmup):
        train_step(model, opt, averager, batches[i % N_BATCHES], n_global)
    torch.cuda.synchronize()
    captured = None
    if use_graph:
        # capacity of the survivor buffers: 1.5 x the largest count seen while priming / warming up, rounded to 4096 rows
        seen = max(STEP_STATS["max_survivors"], 16384)
        if world > 1:     # (buffers are rank-local, but one number for all keeps the ranks' graphs alike)
            seen_t = torch.tensor([seen], dtype=torch.int64, device=dev)
            dist.all_reduce(seen_t, op=dist.ReduceOp.MAX)
            seen = int(seen_t.item())
        captured = build_captured(model, opt, survivor_capacity(seen), args.steps + args.warmup + 16, n_global,
                                  averager=averager if (world > 1 or force_dist) else None)
        # Several ranks: a capture that fails on ANY rank (the capture pass itself issues no collective, so a failure is local
        # and leaves the others unharmed) sends ALL ranks to the eager form below -- decided by one MIN all-reduce, so that
        # no rank replays a graph whose collectives the others never launch.
        # In-kernel timing of the matrix-core launches (fgs_dyn_t.stamps): the captured launches write their workgroups' wall-clock
        # start / end into the slot the step counter selects -- the roofline figure then comes from the timed replays themselves.
        from fgs_nerf_amd import fused_ops as _fo
        stamp_buf, stamp_launches = None, None
        if rank == 0 and not args.composed and os.environ.get("FGS_BENCH_STAMPS", "1") == "1":
            stamp_buf = torch.zeros(max(args.steps, 1), _fo.STAMP_LAUNCHES, _fo.STAMP_WORDS, dtype=torch.int64, device=dev)
            _fo.STAMPS["buf"], _fo.STAMPS["counter"] = stamp_buf, captured.counter
        capture_ok = 1
        try:
            captured.capture(batches[0])
            stamp_launches = list(_fo.STAMPS["launches"])
        except Exception as e:        # noqa: BLE001
            if world == 1 and not force_dist:
                raise
            capture_ok = 0
            print(f"[bench] rank {rank}: capturing the step failed ({type(e).__name__}: {e}); falling back to eager launches",
                  file=sys.stderr, flush=True)
        if world > 1:
            ok_t = torch.tensor([capture_ok], dtype=torch.int32, device=dev)
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
            capture_ok = int(ok_t.item())
        if capture_ok:
            packed = [torch.stack(b).contiguous() for b in batches]     # rays_o / rays_d / viewdirs / target as one [4, N, 3] block
            # the W warm-up steps in the form the timed steps have: untimed replays (the first launch of a graph uploads it; the
            # eager warm-up steps above served the allocator and the capacity estimates)
            # (one untimed replay here uploads the graph; the W warm-up replays proper run right in front of the timed region, below)
            captured.replay(packed[0])
            torch.cuda.synchronize()
            if stamp_buf is not None:
                stamp_buf.zero_()
        else:
            _fo.STAMPS["buf"] = None
            captured.release()
            captured, use_graph = None, False
            from fgs_nerf_amd import fused as _f
            _f.reset_grid_grad(model, force=True)
            opt.zero_grad(set_to_none=True)
    # Eager runs (several GPUs, --mode eager on request): the same launches, but with the survivor count left on the device
    # (fused.set_sync_free: fixed-capacity buffers, kernels clamp to the device-side count), so that the host never waits
    # for the GPU inside a step and its ~2 ms of Python per step overlap the previous step's kernels.  --mode eager-sync
    # keeps the reference-shaped form (result tensors sized by a survivor count read back once per step).
    sync_free_eager = (not use_graph and not args.composed and args.mode != "eager-sync"
                       and os.environ.get("FGS_MLP", "rc") == "rc")
    if sync_free_eager:
        from fgs_nerf_amd import fused as _fused
        _fused.set_sync_free(model, survivor_capacity(STEP_STATS["max_survivors"]))
        opt.use_skip_flag(model._fused_cache['sync_free']['flags'][1:2].data_ptr())   # an overflowed step changes nothing
        for i in range(2):                       # allocator warm-up of the capacity-sized buffers
            train_step(model, opt, averager, batches[i % N_BATCHES], n_global)
        torch.cuda.synchronize()
        st_ = model._fused_cache['sync_free_buffers']
        st_['flags'].zero_(); st_['total'].zero_()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events around every launch of the dominant kernel (the MLP GEMM template) inside the timed region, recorded
    # on the stream the kernels are launched on (PyTorch-ROCm's current stream) -> the "roofline" object below
    from fgs_nerf_amd import fused
    fused.set_profiling((rank == 0) and not args.composed and captured is None, clear=True)
    STEP_STATS["survivors"] = 0
    # the cyclic collector stays out of the timed region: a generation-2 pass over torch's object graph takes milliseconds,
    # several steps' worth, and lands in some 30-step runs and not in others (2.29 vs 2.63 ms/step on the same box)
    # guard against a step that is fast because it is broken: every parameter group must have moved by the end of the timed
    # region (a captured step once ran without k0's Adam pass and read 2.6 % "faster")
    def _digest():
        return [torch.stack([p.detach().double().abs().sum() for p in g['params']]).sum() for g in opt.param_groups]
    digest_before = _digest()
    import gc
    gc.collect()
    gc_was_enabled = gc.isenabled()
    if os.environ.get("FGS_BENCH_GC") != "1":
        gc.disable()
    if captured is not None:
        # Everything host-side is done (digests, the collector): the W warm-up steps in the form the timed steps have, their
        # counters cleared by device-side fills queued behind them, one synchronisation, the clock -- the device does not idle
        # between the warm-up steps and the timed ones (with the host-side work in between, 20 timed steps read 1.79 ms instead of
        # 1.73).  What remains is the WORKLOAD's own drift, not a warm-up effect: per-step times of one run (FGS_BENCH_SERIES=1)
        # fall from 1.82 ms to 1.63 over the first dozen steps and on to 1.50-1.59 by the 80th, with or without 300 ms of unrelated
        # matrix work in front (--clock-warmup-ms, default 0: 1.724 against 1.734) -- training smooths the noisy initial sdf and
        # fewer samples survive to the MLPs.  The line's FLOP and survivor counts are those of the timed steps themselves.
        if args.clock_warmup_ms > 0:
            spin_a = torch.randn(4096, 4096, device=dev)
            spin_b = torch.empty_like(spin_a)
            t_spin = time.perf_counter()
            while (time.perf_counter() - t_spin) * 1e3 < args.clock_warmup_ms:
                for _ in range(8):
                    torch.mm(spin_a, spin_a, out=spin_b)
                torch.cuda.synchronize()
            del spin_a, spin_b
        for i in range(args.warmup):
            captured.replay(packed[i % N_BATCHES])
        captured.flush()
        captured.clear_counters()
        if stamp_buf is not None:
            stamp_buf.zero_()         # (the warm-up replays stamped the slots the timed ones reuse; k_gemm's are atomic min / max)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    t0 = time.perf_counter()
    samples = 0
    series = [] if os.environ.get("FGS_BENCH_SERIES") == "1" else None      # (diagnostic: an event per step, printed to stderr)
    for i in range(args.steps):
        if series is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            series.append(ev)
        if captured is not None:
            captured.replay(packed[i % N_BATCHES])
        else:
            train_step(model, opt, averager, batches[i % N_BATCHES], n_global)
        samples += n_inbbox[i % N_BATCHES]
    if captured is not None:
        captured.flush()              # (the last replay's k0 update, which the next replay's head would have applied: inside the clock)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if series is not None and rank == 0:
        print("[bench] per-step ms: " + " ".join(f"{series[k].elapsed_time(series[k + 1]):.3f}" for k in range(len(series) - 1)),
              file=sys.stderr, flush=True)
    if gc_was_enabled:
        gc.enable()
    fused.set_profiling(False)
    digest_after = _digest()          # (device scalars, compared at the very end: a host read here would let the GPU idle -- and
                                      #  its clock drop -- in front of the profiling steps that follow)
    roofline_note = "HIP events on the launch stream immediately around every MLP matrix-core launch in the timed region"
    stamped = None
    if sync_free_eager:
        from fgs_nerf_amd import fused as _fused
        overflow, total = _fused.sync_free_state(model)
        if overflow and world == 1:
            raise SystemExit("survivor capacity overflowed during the timed region: rerun with --mode eager-sync")
        if overflow:          # (several ranks: leaving here would strand the others in the collectives below; flag the line)
            print(f"[bench] rank {rank}: survivor capacity overflowed during the timed region", file=sys.stderr, flush=True)
        STEP_STATS["survivors"] = total
        STEP_STATS["overflow"] = bool(overflow)
    if captured is not None:
        overflow, total = captured.check()
        if overflow and world == 1:
            raise SystemExit(f"survivor capacity {captured.capacity} (or the k0 exchange capacity {captured.exchange_capacity}) "
                             "overflowed during the timed region: rerun with --mode eager")
        if overflow:          # (several ranks: leaving here would strand the others in the collectives below; flag the line)
            print(f"[bench] rank {rank}: a capacity overflowed during the timed region (exchange: {captured.exchange_overflowed()})",
                  file=sys.stderr, flush=True)
            STEP_STATS["overflow"] = True
        survivors_timed = total
        if stamp_buf is not None and stamp_launches:
            # the timed replays timed their own matrix-core launches (a graph replay cannot carry HIP events around a kernel):
            # min(start) .. max(end) over each launch's workgroups, 100 MHz wall clock, one reading per launch and replay
            stamped = _fo.stamps_read(stamp_buf, stamp_launches)
            _fo.STAMPS["buf"] = None
            roofline_note = (f"in-kernel wall-clock readings (s_memrealtime, 100 MHz; fgs_dyn_t.stamps) of every MLP matrix-core launch of "
                             f"the {args.steps} hipGraph replays of the timed region itself: first workgroup's start to last workgroup's end")
            STEP_STATS["survivors"] = survivors_timed
        else:
            # without the stamps: the MLP kernels are timed right behind the timed region, in the same process on the same model
            # and batches, by PROFILE_STEPS eager steps of the same loop
            PROFILE_STEPS = 10
            fused.set_profiling(rank == 0, clear=True)
            for i in range(PROFILE_STEPS):
                train_step(model, opt, averager, batches[i % N_BATCHES], n_global)
            torch.cuda.synchronize()
            fused.set_profiling(False)
            STEP_STATS["survivors"] = survivors_timed        # (the profiling steps above counted theirs)
            roofline_note = (f"HIP events on the launch stream immediately around every MLP matrix-core launch, in {PROFILE_STEPS} eager steps "
                             "of the same loop run right behind the timed region (graph replays cannot carry timing events)")

    # Several ranks: the replicas must have stayed bit-identical (same averaged gradients, same updates) -- each group's digest
    # MIN- and MAX-reduced over the ranks; the line carries the verdict (`config.replicas_in_sync`).  This is the only place where
    # the exchange is checked on REAL peers: the test suite has gloo ranks and a single-rank RCCL group.
    replica_spread = None
    if world > 1 or force_dist:            # (a forced single-rank group runs the same code: trivially in sync)
        replica_spread = replica_digest_spread(digest_after)
    stats = torch.tensor([elapsed, float(samples)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ssum = stats[1:].clone()
        dist.all_reduce(ssum, op=dist.ReduceOp.SUM)
        elapsed, samples = float(tmax.item()), float(ssum.item())
    if rank == 0:
        line = {
            "metric": f"M ray-samples/sec (fwd+bwd), {GRID}^3 grid, {RAYS_PER_GPU}-ray batch",
            "value": round(samples / elapsed / 1e6, 3), "unit": "M ray-samples/s",
            "n_gpus": (dist.get_world_size() if dist.is_initialized() else 1), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"configs[1]: {GRID}^3 sdf(1ch)+k0(12ch) fine-stage training step "
                                    f"(forward_fine + losses + backward + sdf TV + MaskedAdam), {RAYS_PER_GPU} rays/GPU/step")
                       if args.stage == "fine" else
                       ("configs[2] path at the bench size: 160^3 coarse-stage training step (5^3 smoothing + gradient "
                        f"volume + forward_coarse + losses + backward + sdf TV + MaskedAdam), {RAYS_PER_GPU} rays/GPU/step"),
                       "grid": GRID, "rays_per_gpu": RAYS_PER_GPU, "inbbox_samples_per_step_per_gpu": int(sum(n_inbbox) / len(n_inbbox)),
                       "emitted_samples_per_step_per_gpu": int(sum(n_total) / len(n_total)),
                       "mlp_survivors_per_step_per_gpu": int(STEP_STATS["survivors"] / max(args.steps, 1)),
                       "path": "composed" if args.composed else "fused", "parallelism": f"dp{world} rays",
                       "clock_warmup_ms": (args.clock_warmup_ms if captured is not None else 0.0)},
        }
        if STEP_STATS.get("overflow"):
            line["config"]["capacity_overflow_on_rank0"] = True
        if sdf_tune is not None and sdf_tune.get("tuned"):
            line["config"]["sdf_exchange_tuning"] = sdf_tune
        if replica_spread is not None:
            line["config"]["replicas_in_sync"] = bool(replica_spread == 0.0)
            line["config"]["replica_digest_spread"] = replica_spread
            if replica_spread != 0.0:
                print(f"[bench] the ranks' parameters have drifted apart (relative digest spread {replica_spread:.3e}): the gradient "
                      "exchange is not doing its job", file=sys.stderr, flush=True)
        from fgs_nerf_amd import fused as _fz
        if _fz._MLP_COLLAPSE and args.stage == "fine" and not args.composed:
            # a labelled mode, not the headline: less arithmetic than the reference's operation order (roofline books the FLOP
            # it executes, so `frac` cannot rise from doing less)
            line["config"]["mlp_mode"] = ("FGS_MLP_COLLAPSE=1: rgbnet's last Linear and refnet's first as ONE 256x256 layer with a "
                                          "per-step pre-multiplied weight (368 640 instead of 434 176 MAC per survivor and pass); "
                                          "NOT the reference's operation order -- secondary line")
        else:
            line["config"]["mlp_mode"] = "reference operation order"
        if captured is not None and captured.exchange_capacity is not None:
            line["config"]["k0_exchange"] = {"form": "brick-sparse, device-counted, inside the captured step",
                                             "capacity_bricks": captured.exchange_capacity,
                                             "bytes_per_step": captured.exchange_capacity * 64 * 12 * 4}
            if captured.sdf_exchange_capacity is not None:
                line["config"]["sdf_exchange"] = {"form": "brick-sparse, device-counted (occupancy read from the gradient)",
                                                  "capacity_bricks": captured.sdf_exchange_capacity,
                                                  "bytes_per_step": captured.sdf_exchange_capacity * 64 * 4,
                                                  "dense_bytes": 4 * GRID ** 3}
        line["config"]["step_mode"] = ("one hipGraph replay per step (gradient exchange included), no device->host read"
                                       if captured is not None and captured.averager is not None else
                                       "one hipGraph replay per step, no device->host read" if captured is not None
                                       else "eager launches, no device->host read (device-side survivor count)" if sync_free_eager
                                       else "eager launches, one survivor-count read per step")
        # (sync-free eager launches are issued for the buffers' CAPACITY; their algorithmic work is the rows behind the
        # device-side count: scale the FLOP the launch sites booked by rows processed / rows launched)
        flop_scale = 1.0
        if sync_free_eager:
            cap_rows = model._fused_cache['sync_free']['capacity'] * max(args.steps, 1)
            flop_scale = STEP_STATS["survivors"] / max(cap_rows, 1)
        if stamped is not None:       # (captured launches were issued for the CAPACITY too)
            flop_scale = STEP_STATS["survivors"] / max(captured.capacity * max(args.steps, 1), 1)
        line["roofline"] = fused.roofline_report(pmc, flop_scale, stamped=stamped)
        if line["roofline"] is not None:
            line["roofline"]["timing"] = roofline_note
        if line["roofline"] is not None and args.stage == "fine":
            # The path is MFMA-bound, not HBM-bound: both HBM fractions for the record, from the
            # bytes a step MUST move.  sampled path: 688 + 3456 rho algorithmic bytes per in-bbox sample (SURVEY 8d's table);
            # dense per-step streams that remain: sdf TV 3*4*G^3, dense Adam on sdf 7*4*G^3, one zero fill of sdf.grad 4*G^3;
            # k0: 7*4*12 B per TOUCHED voxel (the voxel-granular masked update; counted below from the survivors' trilinear
            # corners) -- no dense k0 Adam pass and no k0.grad fill exist any more.  The survey's dense figure (k0 Adam over
            # all of 12*G^3 and both gradient fills) is kept under its own key.
            n_in = sum(n_inbbox) / len(n_inbbox)
            rho = STEP_STATS["survivors"] / max(args.steps, 1) / n_in
            step_s = elapsed / args.steps
            sampled = (688.0 + 3456.0 * rho) * n_in
            g3 = float(GRID) ** 3
            touched = touched_voxels(model, batches)
            must = sampled + (3 * 4 + 7 * 4 + 4) * g3 + 7 * 4 * 12 * touched
            dense_survey = sampled + (3 * 4 + 7 * 4 + 7 * 4 * 12 + 4 * 13) * g3
            mlp_flop = 3 * 2 * 434176.0 * STEP_STATS["survivors"] / max(args.steps, 1)      # fwd + data + weight gradients
            line["roofline"]["hbm_fraction_for_reference"] = {
                "rho_survivors_per_inbbox_sample": round(rho, 4),
                "k0_voxels_touched_per_step": int(touched),
                "sampled_path": round(sampled / step_s / 8e12, 4),
                "whole_step": round(must / step_s / 8e12, 4),
                "whole_step_bytes": int(must),
                "whole_step_if_k0_were_updated_densely_(survey_8d)": round(dense_survey / step_s / 8e12, 4),
                "note": "bytes the step must move / step time / 8 TB/s.  The step is bound by fp32 matrix throughput: "
                        f"the north_star's >= 40 % HBM-roofline target does not apply to this design ({must / 1e9:.2f} GB per step)"}
            line["roofline"]["mfma_floor_ms_per_step"] = round(mlp_flop / 157.3e12 * 1e3, 4)
            line["roofline"]["mfma_floor_note"] = ("MLP FLOP of a step (3 x 2 x 434 176 per survivor) at the fp32 MFMA peak: no "
                                                   "schedule of exact-fp32 products can run the step faster")
        # (the CPU baseline is timed at N = 1 only: the other ranks of a multi-GPU run would sit in the final barrier meanwhile)
        line["cpu_baseline"] = None if (args.no_cpu_baseline or args.stage != "fine" or world > 1) else cpu_baseline()
        stale = [g.get('name', str(i)) for i, (g, a, b) in enumerate(zip(opt.param_groups, digest_before, digest_after))
                 if bool(a == b)]
        if stale and args.steps > 0:
            line["broken"] = f"parameter group(s) {stale} did not change during the timed region"
            print("[bench] " + line["broken"] + " -- the step is broken", file=sys.stderr, flush=True)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if world > 1 or force_dist:
        dist.barrier()                # (the line is out before any rank starts tearing its communicators down)
        if captured is not None:
            captured.release()        # graphs holding RCCL nodes go before the communicator does
        dist.destroy_process_group()
    if rank == 0 and line.get("broken"):
        raise SystemExit(2)


if __name__ == "__main__":
    main()
