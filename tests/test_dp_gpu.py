"""Data-parallel semantics of `SRModel.optimize_parameters` (the reference wraps net_g / net_d in DistributedDataParallel,
/root/reference/traiNNer/models/base_model.py:228-234, and mean-reduces the log, :847-870): 2 ranks x batch 2 must equal 1 rank x batch 4.

Two fresh rank processes (tests/_dp_step_worker.py, started with torch.distributed.run) share the one GPU of the box over the `gloo` backend;
they run the generator step (L1 + vanilla GAN through UNetDiscriminatorSN) and the discriminator step through the bucketed `GradExchange`,
check that the replicas stay bit-identical, and hand rank 0's first moments / weights / log back.  The parent runs the same step on the whole
batch in one process.  fp32 compute, so the only difference is the summation order over the batch: first moments (0.1 x gradient) agree to
1e-4 rel-L2; weights after the first AdamW step move by lr * sign-like(g), so they are compared with an absolute bound of 2 * lr on the few
elements whose gradient is at rounding level and 1e-6 elsewhere."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def test_two_ranks_equal_one_rank_with_twice_the_batch(cuda, tmp_path):
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TRAINNER_BUCKET_MB="1")        # 1 MB buckets: several per network, so the overlap path runs
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_dp_step_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    two = torch.load(tmp_path / "rank0.pt", weights_only=False)           # written by our own worker a moment ago
    # the overlap really happened: more than one bucket, none launched before its range was final, all but the last before backward ended
    assert len(two["buckets"]) > 2 and all(two["buckets"][i][0] >= off for i, off in two["launch_log"])
    assert sum(1 for _, off in two["launch_log"] if off > 0) >= len(two["buckets"]) - 1

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _dp_step_worker as W

    model = W.build()
    lq, gt = W.global_batch()
    model.feed_data({"lq": lq, "gt": gt})
    model.optimize_parameters(1, 0, True)
    log1 = model.get_current_log()
    one = W.snapshot(model)
    for k in ("m_g", "m_d"):
        e = rel_l2(two["snap"][k], one[k])
        assert e < 1e-4, (k, e)
    for k, lr in (("w_g", 2e-4), ("w_d", 1e-4)):
        d = (two["snap"][k] - one[k]).abs()
        assert d.max().item() <= 2 * lr * 1.001 and (d > 1e-6).float().mean().item() < 0.02, (k, d.max().item(), (d > 1e-6).float().mean().item())
    assert set(two["log"]) == set(log1) and {"l_g_l1", "l_g_gan", "l_g_total", "l_d_real", "l_d_fake", "grad_norm_g", "grad_norm_d"} <= set(log1)
    for k, v in log1.items():
        assert abs(two["log"][k] - v) <= 2e-5 * max(abs(v), 1e-3), (k, two["log"][k], v)
    # accumulation over two micro-batches of the same half batch == twice its gradient / 2 -> the same averaged gradient as step 1 of `snap`
    e = rel_l2(two["snap_accum"]["m_g"], two["snap"]["m_g"])
    print(f"2 ranks vs 1 rank: first-moment rel-L2 G {rel_l2(two['snap']['m_g'], one['m_g']):.2e} D {rel_l2(two['snap']['m_d'], one['m_d']):.2e}; accum-2 vs accum-1 {e:.2e}")


def test_bench_starts_its_own_ranks_and_reports_the_whole_job(cuda):
    """`python bench.py --gpus 2` without a launcher (what the driver's scaling run does with N = 2, 4, 8): the parent starts the ranks through
    torch.distributed.run, every rank runs the timed steps between barriers, the time is the max over ranks and rank 0 prints ONE JSON line whose value is
    the whole job's images/s.  Rehearsed on the one GPU of this box (TRAINNER_BENCH_REHEARSE=1: shared device, gloo) -- the numbers mean nothing here."""
    import json

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TRAINNER_BENCH_REHEARSE="1", TRAINNER_BUCKET_MB="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--blocks", "2", "--batch", "2", "--lq", "32",
           "--no-cpu-baseline", "--no-variants", "--no-loader"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and abs(d["value"] - 4 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]
    assert "roofline" in d and "cpu_baseline" not in d
