"""Rank process of tests/test_dp_gpu.py (started by torch.distributed.run, 2 ranks, BOTH on cuda:0, gloo backend -- a one-GPU rehearsal of the
data-parallel path that production runs one-rank-per-GPU over RCCL): one `SRModel.optimize_parameters` step on this rank's half of a global
batch; rank 0 writes what the parent compares with its own single-process step on the whole batch."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_opt(accum: int = 1):
    from trainner_redux_amd.utils.options import parse_options_dict

    return parse_options_dict(dict(
        name="dp", scale=4, num_gpu=1, path={}, manual_seed=7, use_amp=False,
        network_g=dict(type="esrgan", num_blocks=1), network_d=dict(type="unetdiscriminatorsn", num_feat=16),
        datasets=dict(train=dict(name="x", type="pairedimagedataset", lq_size=16, batch_size_per_gpu=2, accum_iter=accum)),
        train=dict(total_iter=100, ema_decay=0.999, ema_power=0.75,
                   optim_g=dict(type="AdamW", lr=2e-4, weight_decay=0, betas=[0.9, 0.99]),
                   optim_d=dict(type="AdamW", lr=1e-4, weight_decay=0, betas=[0.9, 0.99]),
                   losses=[dict(type="l1loss", loss_weight=1.0), dict(type="ganloss", gan_type="vanilla", loss_weight=0.1)])))


def global_batch():
    g = torch.Generator().manual_seed(99)
    gt = torch.rand(4, 3, 64, 64, generator=g)
    lq = torch.nn.functional.interpolate(gt, scale_factor=0.25, mode="bicubic", antialias=True).clamp(0, 1)
    return lq, gt


def build(accum: int = 1):
    import trainner_redux_amd  # noqa: F401
    from trainner_redux_amd.models import build_model
    from trainner_redux_amd.utils.misc import set_random_seed

    set_random_seed(7)                                 # identical replicas on every rank (what DDP's constructor broadcast gives, base_model.py:228-234)
    return build_model(make_opt(accum))


def snapshot(model):
    og, od = model.optimizers[0], model.optimizers[1]
    m_g = torch.cat([og.state[p]["exp_avg"].flatten() for p in model.net_g.parameters() if p.requires_grad]).cpu()
    m_d = torch.cat([od.state[p]["exp_avg"].flatten() for p in model.net_d.parameters() if p.requires_grad and p in od.state]).cpu()
    w_g = model.net_g.flat_parameters()[0].detach().cpu().clone()
    w_d = model.net_d.flat_parameters()[0].detach().cpu().clone()
    return dict(m_g=m_g, m_d=m_d, w_g=w_g, w_d=w_d)


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)                           # one GPU, two ranks
    from trainner_redux_amd.utils.dist_util import get_dist_info, init_dist

    os.environ["LOCAL_RANK"] = "0"
    init_dist("pytorch", backend="gloo")
    assert get_dist_info() == (rank, world)
    model = build()
    lq, gt = global_batch()
    n = lq.shape[0] // world
    model.feed_data({"lq": lq[rank * n:(rank + 1) * n], "gt": gt[rank * n:(rank + 1) * n]})
    model.optimize_parameters(1, 0, True)
    ex = model.net_g.grad_exchange
    assert ex is not None and len(ex.buckets) >= 1 and ex.next == len(ex.buckets)
    log = model.get_current_log()                      # rank mean (base_model.py:847-870), evaluated at log time here
    snap = snapshot(model)
    # replicas stay identical: every rank applied the same averaged gradient
    for key in ("w_g", "w_d"):
        t = snap[key].cuda()
        both = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(both, t)                       # test-only collective
        assert torch.equal(both[0], both[1]), key
    # gradient accumulation: micro-step 1 of 2 must not exchange or step, micro-step 2 exchanges the local sum
    model2 = build(accum=2)
    w0 = model2.net_g.flat_parameters()[0].clone()
    model2.feed_data({"lq": lq[rank * n:(rank + 1) * n], "gt": gt[rank * n:(rank + 1) * n]})
    model2.optimize_parameters(1, 0, False)
    assert torch.equal(model2.net_g.flat_parameters()[0], w0) and getattr(model2.net_g, "grad_exchange", None) is None
    model2.optimize_parameters(1, 1, True)
    assert not torch.equal(model2.net_g.flat_parameters()[0], w0) and model2.net_g.grad_exchange.next == len(model2.net_g.grad_exchange.buckets)
    snap2 = snapshot(model2)
    if rank == 0:
        torch.save(dict(log=log, snap=snap, snap_accum=snap2, launch_log=ex.launch_log, buckets=ex.buckets), os.path.join(out_dir, "rank0.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
