"""Multi-GPU layer: one process per GPU, frames sharded over ranks, ONE collective per step.

The reference has no distributed code at all.  Frames are independent (eval-mode BN, no
cross-sample op in ``forward``), so the path shards by batch with no data-path exchange; the only
collective is the all-gather of the predicted joints — ``[B/world, 15, 3]`` float32, 180 B per frame —
over RCCL/xGMI (``backend="nccl"`` is RCCL on ROCm).  It is latency-bound (a few KB), so it is issued
once per step as a single ``all_gather_into_tensor``; bucket sizes / per-link bandwidth are irrelevant.
On CPU (tests) the same code runs over ``gloo``.
"""
from __future__ import annotations

import os
import sys
from datetime import timedelta

import torch
import torch.distributed as dist

from . import _lib


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def _pin_host_threads(local_rank: int, local_world: int):
    """N dispatch loops on one host (about 170 launches per 10 ms step each) must not fight for cores: every rank takes a
    contiguous share of the cores this process may use and keeps its intra-op pools inside it.  Returns (n_threads, cores)."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return torch.get_num_threads(), None
    per = max(1, len(cores) // max(1, local_world))
    mine = cores[(local_rank % max(1, local_world)) * per:][:per] or cores
    if os.environ.get("SCENEEGO_PIN_CORES", "1") == "1" and local_world > 1:
        try:
            os.sched_setaffinity(0, mine)
        except OSError:
            mine = cores
    n = max(1, min(len(mine), int(os.environ.get("SCENEEGO_RANK_THREADS", "4"))))
    torch.set_num_threads(n)
    return n, mine


def init_from_env(backend: str | None = None, device_type: str = "cuda", timeout_s: float | None = None):
    """Initialise ``torch.distributed`` from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*; returns (rank, world, device).

    With more than one rank the launcher's MASTER_ADDR / MASTER_PORT are REQUIRED (no fixed default port: two jobs on one
    host would meet on it), the process group is created with an explicit timeout (``SCENEEGO_DIST_TIMEOUT_S``, 120 s) and a
    failure to form it ends the rank with exit code 4 and the backend's message - it is never retried by starting a new
    process from a rank that has touched the GPU."""
    rank, local_rank, world = env_world()
    if device_type == "cuda":
        index = local_rank
        if os.environ.get("SCENEEGO_SHARE_GPU") == "1":
            # readiness runs on a box with fewer GPUs than ranks (tests): ranks share devices.  RCCL refuses two ranks on one
            # device, so such a run also sets SCENEEGO_DIST_BACKEND=gloo; everything else (sharding, stream ordering) is the N-GPU code
            index = local_rank % max(1, torch.cuda.device_count())
        elif world > 1 and local_rank >= torch.cuda.device_count():
            print(f"[sceneego dist] rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} HIP device(s) visible "
                  f"(one rank per GPU; SCENEEGO_SHARE_GPU=1 + SCENEEGO_DIST_BACKEND=gloo for shared-device readiness runs)",
                  file=sys.stderr, flush=True)
            sys.exit(4)
        torch.cuda.set_device(index)
        device = torch.device("cuda", index)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            print(f"[sceneego dist] rank {rank}: WORLD_SIZE={world} but MASTER_PORT is not set - start the ranks with "
                  f"torch.distributed.run (or `python bench.py --gpus N`, which does)", file=sys.stderr, flush=True)
            sys.exit(4)
        backend = backend or os.environ.get("SCENEEGO_DIST_BACKEND") or None
        if backend is None:
            backend = "nccl" if device_type == "cuda" else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = device
        timeout_s = float(os.environ.get("SCENEEGO_DIST_TIMEOUT_S", "120")) if timeout_s is None else timeout_s
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        threads, cores = _pin_host_threads(local_rank, local_world)
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timedelta(seconds=timeout_s), **kw)
            if backend == "nccl":
                # form the ring NOW, under the timeout, instead of inside the first timed collective
                probe = torch.zeros(1, device=device)
                dist.all_reduce(probe)
                torch.cuda.synchronize(device)
        except Exception as e:       # RCCL / store errors: say which rank, which device, what the backend said - and stop
            print(f"[sceneego dist] rank {rank}/{world} on {device}: {backend} process group failed within {timeout_s:.0f} s: "
                  f"{type(e).__name__}: {e}", file=sys.stderr, flush=True)
            sys.exit(4)
        if os.environ.get("SCENEEGO_DIST_QUIET") != "1":
            print(f"[sceneego dist] rank {rank}/{world} device {device} backend {dist.get_backend()} "
                  f"master {os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']} host threads {threads}"
                  + (f" cores {cores[0]}-{cores[-1]}" if cores else ""), file=sys.stderr, flush=True)
    return rank, world, device


def shard_range(total: int, rank: int, world: int):
    """Contiguous frame range [lo, hi) of ``rank``; the first ``total % world`` ranks get one extra frame."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_joints(joints: torch.Tensor) -> torch.Tensor:
    """[b,J,3] on every rank (equal b) -> [world*b,J,3] in rank order.  No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return joints
    world = dist.get_world_size()
    joints = joints.contiguous()
    # bench.py's separate timing pass: HIP events on the issuing stream around the collective (RCCL runs it on its own stream, which
    # waits for this one and which this one waits for: the bracket covers it) -> `allgather_us` on the bench line
    with _lib._timed(("allgather", world, int(joints.numel()))):
        if joints.is_cuda and dist.get_backend() == "gloo":
            # shared-GPU readiness mode only (init_from_env): gloo moves host memory, so the shard goes through the host behind the
            # producing stream; the production backend (RCCL) gathers device tensors in place, stream-ordered
            host = joints.cpu()           # synchronises with the current stream, which the caller made wait for the forward
            out = torch.empty((world * host.shape[0],) + tuple(host.shape[1:]), dtype=host.dtype)
            dist.all_gather_into_tensor(out, host)
            return out.to(joints.device)
        out = torch.empty((world * joints.shape[0],) + tuple(joints.shape[1:]), dtype=joints.dtype, device=joints.device)
        dist.all_gather_into_tensor(out, joints)
        return out


def all_gather_joints_ragged(joints: torch.Tensor, total: int) -> torch.Tensor:
    """As above when ``total`` does not divide evenly: pads every shard to the largest, gathers, trims."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return joints
    world = dist.get_world_size()
    per = (total + world - 1) // world
    pad = torch.zeros((per,) + tuple(joints.shape[1:]), dtype=joints.dtype, device=joints.device)
    pad[: joints.shape[0]] = joints
    out = all_gather_joints(pad)
    pieces = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        pieces.append(out[r * per: r * per + (hi - lo)])
    return torch.cat(pieces, dim=0)


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_values(value: float, device) -> list:
    """``value`` of every rank, in rank order, on every rank ([value] without a process group)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    on = "cpu" if dist.get_backend() == "gloo" else device
    t = torch.tensor([value], dtype=torch.float64, device=on)
    out = torch.empty(dist.get_world_size(), dtype=torch.float64, device=on)
    dist.all_gather_into_tensor(out, t)
    return [float(v) for v in out.tolist()]


def describe() -> dict:
    """What the collective really ran on: {"world_size", "backend"} of the live process group (backend "nccl" = RCCL on ROCm)."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"world_size": 1, "backend": None}
    return {"world_size": dist.get_world_size(), "backend": dist.get_backend()}
