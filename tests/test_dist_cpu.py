"""world_size-2 `gloo` rehearsal of the data-parallel path (one process per GPU in production, RCCL there):
the gradient exchange is ONE bucketed in-place all-reduce of the flat gradient arena followed by the 1/world scale that
the fused optimizer sweep applies, and the sample partition comes from EnlargedSampler -- no other collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import trainner_redux_amd  # noqa: F401
    from trainner_redux_amd.archs import build_network
    from trainner_redux_amd.data import EnlargedSampler
    from trainner_redux_amd.utils.dist_util import allreduce_flat_, get_dist_info, init_dist, master_only

    init_dist("pytorch", backend="gloo")
    assert get_dist_info() == (rank, world)
    torch.manual_seed(1024)                       # same seed on every rank => identical replicas (DDP broadcast equivalent)
    net = build_network({"type": "esrgan", "scale": 4, "num_blocks": 1})
    params, grads = net.flat_parameters()
    ref = params.clone()
    dist.broadcast(params, 0)
    assert torch.equal(params, ref)               # replicas were already identical
    g = torch.Generator().manual_seed(100 + rank)
    grads.copy_(torch.randn(grads.numel(), generator=g))
    mine = grads.clone()
    allreduce_flat_(grads, bucket_bytes=1 << 20)  # several buckets
    grads.mul_(1.0 / world)
    others = [torch.randn(grads.numel(), generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    assert torch.allclose(grads, sum(others) / world, atol=1e-6)
    assert torch.equal(others[rank], mine)
    # every parameter's .grad is a view of the arena, so p.grad already holds the averaged gradient
    p0 = next(net.parameters())
    assert torch.equal(p0.grad.reshape(-1), grads[: p0.numel()])
    # GradExchange: buckets cut from the END of the arena, each launched as soon as backward reports its range final, 1/world applied at the end
    from trainner_redux_amd.utils.dist_util import GradExchange

    grads.copy_(mine)
    ex = GradExchange(grads, bucket_bytes=1 << 20)
    assert ex.buckets[0][1] == grads.numel() and ex.buckets[-1][0] == 0 and all(a[0] == b[1] for a, b in zip(ex.buckets, ex.buckets[1:]))
    net.grad_exchange = ex
    ex.begin()
    plist = [p for p in net.parameters()]
    order = [plist[-1], plist[len(plist) // 2], plist[len(plist) // 2], plist[1]]      # backward finishes parameters from the last towards the first
    launched = []
    for p in order:
        net._grads_final_from(p)
        launched.append(ex.next)
    assert launched[0] <= launched[1] == launched[2] <= launched[3] < len(ex.buckets)   # monotone, idempotent, the first bucket waits for finish()
    assert all(ex.buckets[i][0] >= off for i, off in ex.launch_log)                    # no bucket went out before its whole range was final
    ex.finish()
    assert ex.next == len(ex.buckets) and not ex.active
    assert torch.allclose(grads, sum(others) / world, atol=1e-6)
    net.grad_exchange = None
    s = EnlargedSampler(list(range(11)), world, rank, ratio=2)
    s.set_epoch(3)
    idx = torch.tensor(list(iter(s)))
    gathered = [torch.zeros_like(idx) for _ in range(world)]
    dist.all_gather(gathered, idx)                # test-only collective
    if rank == 0:
        full = torch.randperm(s.total_size, generator=torch.Generator().manual_seed(3)) % 11
        inter = torch.stack(gathered, 1).reshape(-1)
        assert torch.equal(inter, full)           # ranks interleave one epoch-seeded permutation

    @master_only
    def write():
        open(os.path.join(out_dir, "ok"), "w").write("1")
        return True

    assert write() is (True if rank == 0 else None)
    # checkpoints: the training loop calls model.save() on every rank (train.py), only rank 0 may write (base_model.py `@master_only`)
    from trainner_redux_amd.models.sr_model import SRModel

    class _Self:                                   # save_network / save_training_state touch nothing of the trainer but its options
        pass

    tiny = torch.nn.Conv2d(3, 4, 3)
    SRModel.save_network(_Self(), tiny, os.path.join(out_dir, "ckpt", f"net_g_1.safetensors"))
    dist.barrier()
    files = sorted(os.listdir(os.path.join(out_dir, "ckpt")))
    assert files == ["net_g_1.safetensors"], files          # one writer, one file, and every rank sees it complete
    from safetensors.torch import load_file

    sd = load_file(os.path.join(out_dir, "ckpt", "net_g_1.safetensors"))
    assert torch.equal(sd["weight"], tiny.state_dict()["weight"]) if rank == 0 else sd["weight"].shape == tiny.weight.shape
    assert hasattr(SRModel.save_training_state, "__wrapped__") and hasattr(SRModel.save_network, "__wrapped__")
    # loader workers: worker k of rank r seeds with num_workers * r + k + seed (data/__init__.py:168-178) -> no two (rank, worker) pairs share a stream
    from trainner_redux_amd.data.prefetch import worker_seed

    seeds = torch.tensor([worker_seed(1024, 4, rank, k) for k in range(4)])
    allseeds = [torch.zeros_like(seeds) for _ in range(world)]
    dist.all_gather(allseeds, seeds)               # test-only collective
    flat = torch.cat(allseeds).tolist()
    assert len(set(flat)) == len(flat) and flat[0] == 1024
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gradient_exchange_and_sharding(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
