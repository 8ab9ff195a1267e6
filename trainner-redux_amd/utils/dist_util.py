"""One process per GPU over RCCL (backend "nccl" on ROCm) -- replaces the reference's DDP / DataParallel
wrapping (/root/reference/traiNNer/utils/dist_util.py:13-88, models/base_model.py:228-236).

Parameters and gradients of a network are flat arenas (archs/_arena.py), so the data-parallel step is an in-place all-reduce(sum) of the
gradient arena -- no module wrapper, no per-parameter hooks, no scatter/gather.  `GradExchange` cuts the arena into a few large buckets
(xGMI is point-to-point: rings are per-link bound, few large messages beat many small ones) and launches each bucket as soon as the
hand-scheduled backward has finished the parameters it covers, so the exchange overlaps the rest of backward (what DDP's reducer does with
its 25 MB buckets, base_model.py:228-234).
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


class GradExchange:
    """Bucketed, backward-overlapped all-reduce of ONE flat gradient arena.

    The arena holds the parameters in registration order and the hand-scheduled backward finishes them from the last layer towards the
    first, so "everything at arena offsets >= `off` is final" is the only notification needed (`ready_from`).  A bucket is launched
    (async, on the process group's own stream, which orders itself after the launches already enqueued on the current stream) once it lies
    entirely above that offset; `finish()` launches what is left, makes the current stream wait for all of them and applies DDP's
    1/world_size average.  With world_size 1 every method is a no-op.

        ex.begin(); net.backward_hip(g)  # -> net calls ex.ready_from(off) as it goes
        ex.finish()
    """

    def __init__(self, flat: torch.Tensor, bucket_bytes: int = 24 << 20) -> None:
        self.flat = flat
        per = max(bucket_bytes // flat.element_size(), 4) // 4 * 4
        n = flat.numel()
        # cut from the END (the first gradients to complete): [(lo, hi)] in launch order; the last bucket absorbs the remainder
        cuts = []
        hi = n
        while hi > 0:
            lo = max(hi - per, 0)
            if lo < per // 2:
                lo = 0
            cuts.append((lo, hi))
            hi = lo
        self.buckets = cuts
        self.next = 0
        self.works: list = []
        self.active = False
        self.launch_log: list = []          # (bucket index, ready offset at launch): tests / profiling

    def begin(self) -> None:
        _, world = get_dist_info()
        self.active = world > 1
        self.next, self.works, self.launch_log = 0, [], []

    def _launch(self, i: int, off: int) -> None:
        lo, hi = self.buckets[i]
        self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        self.launch_log.append((i, off))

    def ready_from(self, off: int) -> None:
        """Every gradient at arena index >= off has been fully accumulated by launches already enqueued on the current stream."""
        if not self.active:
            return
        while self.next < len(self.buckets) and self.buckets[self.next][0] >= off:
            self._launch(self.next, off)
            self.next += 1

    def finish(self) -> None:
        if not self.active:
            return
        self.ready_from(0)
        for w in self.works:
            w.wait()                        # nccl: the current stream waits for the collective; gloo: the host does
        self.works = []
        self.active = False
        _, world = get_dist_info()
        self.flat.mul_(1.0 / world) if not self.flat.is_cuda else _scale_(self.flat, 1.0 / world)


def _scale_(flat: torch.Tensor, s: float) -> None:
    from .. import ops

    gv = flat.view(-1, 4)
    ops.axpby(ops.view(gv), s, ops.NULL_VIEW, 0.0, ops.view(gv), gv.shape[0])
