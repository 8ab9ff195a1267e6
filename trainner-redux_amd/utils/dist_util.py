"""One process per GPU over RCCL (backend "nccl" on ROCm) -- replaces the reference's DDP / DataParallel
wrapping (/root/reference/traiNNer/utils/dist_util.py:13-88, models/base_model.py:228-236).

The gradient exchange itself lives in models/sr_model.py: parameters and gradients are flat arenas, so the
data-parallel step is ONE in-place all-reduce(sum) of the gradient arena followed by the fused optimizer sweep
(which applies 1/world_size) -- no module wrapper, no per-parameter hooks, no scatter/gather.
"""
from __future__ import annotations

import functools
import os

import torch
import torch.distributed as dist


def init_dist(launcher: str = "pytorch", backend: str | None = None, **kwargs) -> None:
    if launcher not in ("pytorch", "slurm"):
        raise ValueError(f"Invalid launcher type: {launcher}")
    if launcher == "slurm":
        # same environment translation as the reference's slurm path (dist_util.py:31-60)
        proc_id = int(os.environ["SLURM_PROCID"])
        ntasks = int(os.environ["SLURM_NTASKS"])
        os.environ.setdefault("MASTER_PORT", str(kwargs.pop("port", 29500)))
        os.environ.setdefault("MASTER_ADDR", os.environ.get("SLURM_LAUNCH_NODE_IPADDR", "127.0.0.1"))
        os.environ["WORLD_SIZE"] = str(ntasks)
        os.environ["RANK"] = str(proc_id)
        os.environ.setdefault("LOCAL_RANK", str(proc_id % max(torch.cuda.device_count(), 1)))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        backend = backend or "nccl"
    else:
        backend = backend or "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, **kwargs)


def get_dist_info() -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def master_only(func):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        rank, _ = get_dist_info()
        if rank == 0:
            return func(*args, **kwargs)
        return None

    return wrapper


def allreduce_flat_(flat: torch.Tensor, bucket_bytes: int = 32 << 20, async_op: bool = False):
    """In-place sum of a flat gradient arena across ranks.

    xGMI is point-to-point: RCCL rings are per-link bound, so few large messages beat many small ones;
    the arena is cut into `bucket_bytes` chunks only so a caller can overlap them with compute."""
    _, world = get_dist_info()
    if world == 1:
        return []
    per = max(bucket_bytes // flat.element_size(), 1)
    works = []
    for off in range(0, flat.numel(), per):
        works.append(dist.all_reduce(flat[off:off + per], op=dist.ReduceOp.SUM, async_op=async_op))
    return works
