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

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_from_env(backend: str | None = None, device_type: str = "cuda"):
    """Initialise ``torch.distributed`` from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*; returns (rank, world, device)."""
    rank, local_rank, world = env_world()
    if device_type == "cuda":
        index = local_rank
        if os.environ.get("SCENEEGO_SHARE_GPU") == "1":
            # readiness runs on a box with fewer GPUs than ranks (tests): ranks share devices.  RCCL refuses two ranks on one
            # device, so such a run also sets SCENEEGO_DIST_BACKEND=gloo; everything else (sharding, stream ordering) is the N-GPU code
            index = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(index)
        device = torch.device("cuda", index)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = backend or os.environ.get("SCENEEGO_DIST_BACKEND") or None
        if backend is None:
            backend = "nccl" if device_type == "cuda" else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
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
