#!/usr/bin/env python3
"""Benchmark of the hot path: training images/sec of `SRModel.optimize_parameters` on synthetic data.

    python bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU; under torch.distributed.run the ranks are given, with
                                                            WORLD_SIZE unset bench.py starts the N rank processes itself and relays rank 0's line)

Workload at N = 1 (BASELINE.json configs[1]): ESRGAN 4x (RRDBNet 23 blocks), L1 + perceptual (VGG19, `perceptualloss`
= fp32), bf16 compute, 256x256 -> 1024x1024 tiles, batch 16 per GPU, AdamW + EMA, every step applies the gradient.
A "step" = one full `optimize_parameters` call on one resident batch; value = global images / max-over-ranks time.
Weak scaling: per-GPU batch is fixed; gradients are exchanged with ONE RCCL all-reduce of the flat gradient arena.

The JSON line also carries
  roofline     -- dominant kernel group by time (configs[1]: the fp32 VGG convs, run as fused Winograd F(4x4,3x3) MFMA kernels): FLOPs the kernel
                  EXECUTES on the MFMA pipe per launch / its average launch duration, measured live with HIP events on the launch stream during
                  the timed region (`frac` <= 1 against the dense MFMA peak of the dtype); the algorithmic 3x3-convolution rate is reported next
                  to it (`algorithmic_tflops`, `algorithmic_vs_direct_peak`), as are algorithmic bytes, counter-measured HBM bytes and `hbm_frac`;
  cpu_baseline -- the oracle's step (pure PyTorch fp32, the reference's CPU path) timed on this host on a bounded
                  sample and scaled to the same unit.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--arch", default="esrgan", choices=["esrgan", "swinir_m", "swinir_s", "span", "span_s", "mosr", "mosr_t"],
                    help="esrgan = BASELINE.json configs[1] (the default bench line); swinir_* = configs[3] (64->256 tiles); "
                         "span* / mosr* = configs[4] (2x, batch 64 per GPU; add --compute f16 for its fp16 + loss-scaling mode)")
    ap.add_argument("--scale", type=int, default=None, help="default 4 (2 for span)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 16 for esrgan = configs[1], 8 for swinir = configs[3] / SURVEY.md cfg4)")
    ap.add_argument("--lq", type=int, default=None, help="LR tile size (default 256 for esrgan, 64 for swinir)")
    ap.add_argument("--blocks", type=int, default=23)
    ap.add_argument("--losses", default="l1+perceptual", choices=["l1", "l1+perceptual", "l1+perceptualfp16"])
    ap.add_argument("--compute", default="bf16", choices=["bf16", "f16", "f32"], help="f16 = the reference's default AMP mode: fp16 operands + loss scaling")
    ap.add_argument("--gan", action="store_true", help="BASELINE.json configs[2] heads: add the UNetDiscriminatorSN + vanilla GAN term (weight 0.1) and its D update")
    ap.add_argument("--otf", action="store_true", help="BASELINE.json configs[2] feed: RealESRGANModel -- every step's feed_data runs the on-the-fly degradation "
                                                       "stages (template values of options/_templates/train/ESRGAN/ESRGAN_OTF_gan.yml) on GT tiles + blur kernels")
    ap.add_argument("--fast-matmul", action="store_true", help="the reference's fast_matmul option: split-bf16 (3 bf16 MFMAs per product) for the fp32-forced VGG; NOT the default line")
    ap.add_argument("--graphs", action="store_true", help="use_compile: true -> the generator's forward / backward replay as hipGraphs (helps the host-launch-bound "
                                                          "small-tile configs; the per-launch roofline timers then see only the launches outside the graphs)")
    ap.add_argument("--loader", action="store_true", help="also time the same steps fed by the real loader (PNG files -> PairedImageDataset(as_uint8) -> pinned "
                                                          "DevicePrefetcher, side-stream H2D + on-device uint8 -> fp32 NCHW): reported as variants.real_loader, never as `value`")
    ap.add_argument("--no-loader", action="store_true", help="skip the real-loader leg (it runs by default for the esrgan workloads on one GPU)")
    ap.add_argument("--loader-workers", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the informational fast_matmul measurement")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()
    swin = args.arch != "esrgan"
    span = args.arch.startswith("span") or args.arch.startswith("mosr")     # the configs[4] shape
    args.scale = args.scale or (2 if span else 4)
    args.batch = args.batch or (64 if span else 8 if swin else 16)
    args.lq = args.lq or (64 if swin else 256)
    args.loader = (args.loader or args.arch == "esrgan") and not args.no_loader and not args.otf
    if swin and args.losses != "l1":
        args.losses = "l1"            # configs[3] names no perceptual head; the reference's SwinIR template trains on a pixel loss
    return args


def make_opt(args):
    from trainner_redux_amd.utils.options import parse_options_dict

    losses = [dict(type="l1loss", loss_weight=1.0)]
    if args.losses == "l1+perceptual":
        losses.append(dict(type="perceptualloss", loss_weight=0.01, criterion="charbonnier"))
    elif args.losses == "l1+perceptualfp16":
        losses.append(dict(type="perceptualfp16loss", loss_weight=0.01, criterion="charbonnier"))
    extra = {}
    train_extra = {}
    if args.gan:
        losses.append(dict(type="ganloss", gan_type="vanilla", loss_weight=0.1))
        extra["network_d"] = dict(type="unetdiscriminatorsn")
        train_extra["optim_d"] = dict(type="AdamW", lr=1e-4, weight_decay=0, betas=[0.9, 0.99])
    if args.otf:
        extra.update(high_order_degradation=True, queue_size=120, blur_prob=0.0, resize_mode_list3=["bilinear", "bicubic", "nearest-exact", "lanczos"],
                     resize_mode_prob3=[0.25, 0.25, 0.25, 0.25])
    return parse_options_dict(dict(
        **extra,
        name="bench", scale=args.scale, num_gpu=1, path={}, manual_seed=1024, use_compile=args.graphs, fast_matmul=args.fast_matmul, use_amp=args.compute != "f32", amp_bf16=args.compute == "bf16",
        network_g=dict(type="esrgan", num_blocks=args.blocks) if args.arch == "esrgan" else dict(type=args.arch, scale=args.scale),
        datasets=dict(train=dict(name="synthetic", type="pairedimagedataset", lq_size=args.lq, batch_size_per_gpu=args.batch, accum_iter=1)),
        train=dict(total_iter=1000000, ema_decay=0.999, ema_power=0.75,
                   optim_g=dict(type="AdamW", lr=2e-4, weight_decay=0, betas=[0.9, 0.99]),
                   scheduler=dict(type="MultiStepLR", milestones=[500000], gamma=0.5), losses=losses, **train_extra),
    ))


class ConvTimer:
    """HIP events around every tr_conv3x3 launch (recorded on the launch stream = torch's current stream)."""

    def __init__(self):
        self.records = []   # (key, flops, ev0, ev1)
        self.enabled = False
        self.only = None    # timed region: the one kernel group whose launches carry events (the dominant group found in the last warm-up step)

    def install(self):
        from trainner_redux_amd import _lib as L, ops

        orig = ops.conv3x3
        timer = self

        def timed(**kw):
            if not timer.enabled or torch.cuda.is_current_stream_capturing():      # a launch recorded into a hipGraph cannot carry timing events
                return orig(**kw)
            cd = "bf16" if kw["compute"] == L.BF16 else "f16" if kw["compute"] == L.F16 else "f32w" if kw["compute"] == L.F32_WINOGRAD else "f32w4" if kw["compute"] == L.F32_WINOGRAD43 else "f32"
            ks = kw.get("ksize", 3)
            if timer.only is not None and (cd, 64 if kw["cout"] > 32 else 32, ks) != timer.only:
                return orig(**kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(**kw)
            e1.record()
            npix = float(kw["n"]) * kw["h"] * kw["w"]
            flops = 2.0 * npix * ks * ks * kw["cin"] * kw["cout"]
            # algorithmic bytes of the launch: input read once (a quarter of the pixels under nearest-2x), every output / side view once
            es = lambda v: 4 if v.dtype == L.F32 else 2  # noqa: E731
            nbytes = npix * kw["cin"] * es(kw["inp"]) * (0.25 if kw.get("in_mode") == L.IN_NEAREST2X else 1.0)
            for name in ("out", "out2", "res1", "res2", "mask"):
                v = kw.get(name)
                if v is not None and v.ptr:
                    nbytes += npix * kw["cout"] * es(v) * (0.25 if (name == "out2" and cd == "f32w4") else 1.0)      # Winograd's second output is the 2x2-pooled map
            fam = "winograd" if cd in ("f32w", "f32w4") else ops.last_conv_family()
            timer.records.append(((cd, 64 if kw["cout"] > 32 else 32, ks), flops, e0, e1, nbytes, fam))

        ops.conv3x3 = timed
        # archs call `ops.conv3x3(...)` through the module attribute, so the wrapper is seen everywhere

    def summary(self, records=None):
        agg = {}
        for key, flops, e0, e1, nbytes, fam in (self.records if records is None else records):
            ms = e0.elapsed_time(e1)
            a = agg.setdefault(key, [0, 0.0, 0.0, 0.0, {}])
            a[0] += 1
            a[1] += flops
            a[2] += ms
            a[3] += nbytes
            a[4][fam] = a[4].get(fam, 0) + 1
        return agg


def host_threads() -> int:
    """CPU share of this process: affinity mask, cgroup quota, capped at 16 (the GPU box's share for one GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def log(msg: str) -> None:
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(args, perceptual: bool):
    """Oracle step on the host cores: 1 image, 64 -> 256 tile = 1/16 of the pixels of one 256 -> 1024 image."""
    from collections import OrderedDict

    from oracle import esrgan as OE, losses as OL, step as OS, swinir as OW, vgg as OV

    threads = host_threads()
    torch.set_num_threads(threads)
    if args.arch == "esrgan":
        hp = OE.esrgan_hparams(scale=4, num_blocks=args.blocks)
        sd = OE.init_state_dict(hp, 0)
        fwd = lambda p, x: OE.forward(p, x, hp)  # noqa: E731
    elif args.arch.startswith("span"):
        from oracle import span as OSP

        hp = OSP.hparams(feature_channels=52 if args.arch == "span" else 48, scale=args.scale)
        g0 = torch.Generator().manual_seed(0)
        sd = OrderedDict((k, torch.randn(shp, generator=g0) * 0.05) for k, shp in OSP.param_table(hp).items() if "eval_conv" not in k)
        fwd = lambda p, x: OSP.forward(p, x, hp)  # noqa: E731
    elif args.arch.startswith("mosr"):
        from oracle import mosr as OM

        hp = OM.hparams(scale=args.scale, **OM.PRESETS[args.arch])
        sd = OM.init_params(hp, 0)
        fwd = lambda p, x: OM.forward(p, x, hp)  # noqa: E731
    else:
        hp = OW.hparams(args.arch, scale=4)
        sd = OW.init_state_dict(hp, 0)
        fwd = lambda p, x: OW.forward(p, x, hp)  # noqa: E731
    params = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    st = OS.AdamWState(OrderedDict((k, v.data) for k, v in params.items()), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
    ema = OS.EMAState(OrderedDict((k, v.data) for k, v in params.items()), beta=0.999, power=0.75)
    g = torch.Generator().manual_seed(1024)
    s_lq = 64
    gt = torch.rand(1, 3, args.scale * s_lq, args.scale * s_lq, generator=g)
    lq = torch.nn.functional.interpolate(gt, scale_factor=1.0 / args.scale, mode="bicubic", antialias=True).clamp(0, 1)
    fns = OrderedDict(l_g_l1=(OL.l1, 1.0))
    if perceptual:
        vsd = OV.synthetic_state_dict(0)
        fns["l_g_perceptual"] = (lambda o, t: OV.perceptual_loss(vsd, o, t), 0.01)
    OS.generator_step(params, fwd, fns, lq, gt, st, ema)  # warm-up
    t0 = time.perf_counter()
    reps = 0
    while reps < 2 or time.perf_counter() - t0 < 10.0:
        OS.generator_step(params, fwd, fns, lq, gt, st, ema)
        reps += 1
        if time.perf_counter() - t0 > 40.0:
            break
    dt = (time.perf_counter() - t0) / reps
    scale = (args.lq / s_lq) ** 2
    res = dict(value=1.0 / (dt * scale), unit="images/sec", cores=threads, kind="port",
               sample=f"{reps} oracle steps of batch 1 at {s_lq}->{args.scale * s_lq} ({dt:.2f} s each, fp32, {threads} threads); "
                      f"one {args.lq}->{args.scale * args.lq} image = {scale:.0f}x those pixels, value scaled by 1/{scale:.0f}")
    if args.arch == "esrgan":
        res["cfg1"] = cpu_baseline_cfg1(args.blocks, threads)
    return res


def cpu_baseline_cfg1(blocks: int, threads: int):
    """BASELINE.json configs[0] exactly (SURVEY.md section 8d "CPU baseline beside it"): ESRGAN 4x, L1 only, batch 2, 64 -> 256, fp32, AdamW + EMA -- the
    oracle's step on all host threads, and batch 1 on ONE thread for a per-core figure.  Bounded: one warm-up + timed steps for ~6 s each."""
    from collections import OrderedDict

    from oracle import esrgan as OE, losses as OL, step as OS

    hp = OE.esrgan_hparams(scale=4, num_blocks=blocks)
    sd = OE.init_state_dict(hp, 0)
    out = {}
    for tag, nthr, nb in (("all_threads", threads, 2), ("one_thread", 1, 1)):
        torch.set_num_threads(nthr)
        params = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
        st = OS.AdamWState(OrderedDict((k, v.data) for k, v in params.items()), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
        ema = OS.EMAState(OrderedDict((k, v.data) for k, v in params.items()), beta=0.999, power=0.75)
        g = torch.Generator().manual_seed(1024)
        gt = torch.rand(nb, 3, 256, 256, generator=g)
        lq = torch.nn.functional.interpolate(gt, scale_factor=0.25, mode="bicubic", antialias=True).clamp(0, 1)
        fns = OrderedDict(l_g_l1=(OL.l1, 1.0))
        fwd = lambda p, x: OE.forward(p, x, hp)  # noqa: E731
        if nthr > 1:
            OS.generator_step(params, fwd, fns, lq, gt, st, ema)      # warm-up (the one-thread leg is too slow to repeat: its single step is the sample)
        t0 = time.perf_counter()
        reps = 0
        while reps < 1 or (nthr > 1 and time.perf_counter() - t0 < 6.0):
            OS.generator_step(params, fwd, fns, lq, gt, st, ema)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        out[tag] = dict(value=nb / dt, unit="images/sec (64->256 tiles)", cores=nthr, batch=nb, s_per_step=dt, steps=reps,
                        tflops=nb * 0.4406 / dt)                     # 0.4406 TFLOP per trained 64 -> 256 image
    torch.set_num_threads(threads)
    return out


def real_loader_leg(args, model, n, s, sc, it):
    """The same training steps fed by the real loader ("data already resident (synthetic) and, separately, with the real loader"):
    synthetic PNG pairs on local disk -> PairedImageDataset(as_uint8) (PIL decode, paired flip / rot / crop on uint8) -> DataLoader workers -> pinned
    DevicePrefetcher (side-stream H2D of uint8 HWC, tr_u8hwc_to_nchw on the device) -> feed_data -> optimize_parameters."""
    import tempfile

    import numpy as np
    from PIL import Image

    from trainner_redux_amd.data.paired_image_dataset import PairedImageDataset
    from trainner_redux_amd.data.prefetch import DevicePrefetcher, build_dataloader

    workers = args.loader_workers if args.loader_workers is not None else max(2, min(12, host_threads() - 2))
    steps = args.steps
    with tempfile.TemporaryDirectory(prefix="tr_bench_") as root:
        rng = np.random.default_rng(7)
        npairs = max(2 * n, 32)
        gsz = sc * s + 2 * sc * 8                       # a little larger than the crop, so that the random crop really moves
        os.makedirs(os.path.join(root, "gt")); os.makedirs(os.path.join(root, "lq"))
        for i in range(npairs):
            gt = rng.integers(0, 256, (gsz, gsz, 3), dtype=np.uint8)
            Image.fromarray(gt).save(os.path.join(root, "gt", f"{i:04d}.png"), compress_level=1)
            Image.fromarray(gt[::sc, ::sc]).save(os.path.join(root, "lq", f"{i:04d}.png"), compress_level=1)
        ds = PairedImageDataset(dict(name="bench", phase="train", scale=sc, dataroot_gt=[os.path.join(root, "gt")], dataroot_lq=[os.path.join(root, "lq")],
                                     gt_size=sc * s, use_hflip=True, use_rot=True), as_uint8=True)

        class Repeat(torch.utils.data.Dataset):        # enough items for warm-up + timed steps (dataset_enlarge_ratio of the reference's sampler)
            def __len__(self):
                return n * (steps + 3)

            def __getitem__(self, i):
                return ds[i % len(ds)]

        loader = build_dataloader(Repeat(), n, num_workers=workers, seed=1024)
        pre = DevicePrefetcher(loader, model.device)
        for _ in range(2):                              # warm-up: worker start-up, first decode
            it += 1
            model.feed_data(pre.next())
            model.optimize_parameters(it, 0, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            it += 1
            model.feed_data(pre.next())
            model.optimize_parameters(it, 0, True)
            model.update_learning_rate(it, -1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        del pre, loader
    return {"value": n * steps / dt, "unit": "images/sec", "ms_per_step": dt / steps * 1e3, "workers": workers,
            "what": f"same workload, every batch decoded from {sc * s + 16 * sc}^2 / {s + 16}^2 PNG pairs by PairedImageDataset(as_uint8) in {workers} loader workers, "
                    "shipped as pinned uint8 HWC by DevicePrefetcher and converted on the device; not the headline value"}


PEAK_TFLOPS = {"f32": 157.3, "f32w": 157.3, "f32w4": 157.3, "bf16": 2500.0, "f16": 2500.0}     # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0
WINO_FACTOR = {"f32w": 2.25, "f32w4": 4.0}          # algorithmic multiplies per multiply the Winograd forms execute on the MFMA pipe


def workload_key(args, n, s):
    return f"{args.arch}/{args.losses}/{args.compute}/batch{n}/lq{s}" + (f"/blocks{args.blocks}" if args.arch == "esrgan" else "") + ("/gan" if args.gan else "")


def roofline(args, agg, dt, n, s, warm=None):
    """The dominant kernel group (by summed launch time) against the roofline that bounds it.  `agg`: launches timed inside the timed region (only the
    dominant group when the last warm-up step has named it); `warm`: every group, timed in the last warm-up step -- the `all` table's other rows.

    executed FLOPs = algorithmic 3x3-convolution FLOPs / WINO_FACTOR (1 for the direct kernels); arithmetic intensity = executed FLOPs /
    algorithmic bytes; bound = mfma when that intensity lies above the ridge (peak FLOP/s / 8 TB/s), else hbm.  `achieved` / `frac` are the
    executed MFMA rate (mfma) or the algorithmic byte rate (hbm) against that roof -- never above 1.  `traffic` = HBM bytes per launch from the
    committed PMC passes (profiles/traffic.json, only for the workload they were taken on); `hbm_frac` = traffic / launch time / 8 TB/s."""
    traffic_tab = {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        traffic_tab = tj.get("workloads", {}).get(workload_key(args, n, s), {})
    except Exception:
        pass

    def entry(k, v):
        cnt, flops, msum, nbytes, fams = v
        wf = WINO_FACTOR.get(k[0], 1.0)
        peak = PEAK_TFLOPS[k[0]]
        alg = flops / (msum * 1e-3) / 1e12
        exe = alg / wf
        gbps = nbytes / (msum * 1e-3) / 1e9
        ridge = peak * 1e12 / (PEAK_HBM_GBPS * 1e9)
        ai = (flops / wf) / max(nbytes, 1.0)
        tkey = {"f32w": "f32/winograd", "f32w4": "f32/winograd43"}.get(k[0], f"{k[0]}/cg{k[1]}") + ("" if k[2] == 3 else f"/k{k[2]}")
        tr = traffic_tab.get(tkey, {}).get("traffic_bytes_per_launch")
        e = {"launches": cnt, "ms": msum, "avg_launch_ms": msum / cnt, "kernel_family": fams,
             "algorithmic_tflops": alg, "executed_mfma_tflops": exe, "mfma_frac": exe / peak,
             "algorithmic_bytes_per_launch": nbytes / cnt, "algorithmic_gbps": gbps, "algorithmic_hbm_frac": gbps / PEAK_HBM_GBPS,
             "intensity_flop_per_byte": ai, "ridge_flop_per_byte": ridge, "bound": "mfma" if ai >= ridge else "hbm"}
        if tr is not None:
            e["traffic"] = tr
            e["hbm_frac"] = tr / (msum / cnt * 1e-3) / (PEAK_HBM_GBPS * 1e9)
            e["traffic_over_algorithmic"] = tr / (nbytes / cnt)
        return tkey, e

    table = {}
    replayed = not agg            # the timed region replayed hipGraphs: no launch went through the timer, the eager warm-up step is all there is
    for k, v in (warm or {}).items():
        if k not in agg:
            tk, e0 = entry(k, v)
            e0["measured_in"] = "eager warm-up step (1 step); the timed region replays the step as hipGraphs" if replayed else "last warm-up step (1 step)"
            table[tk] = e0
    for k, v in agg.items():
        tk, e0 = entry(k, v)
        e0["measured_in"] = "timed region"
        table[tk] = e0
    nsteps_agg = 1 if replayed else None
    if replayed:
        agg = warm
    key = max(agg, key=lambda k: agg[k][2])
    tkey, e = entry(key, agg[key])
    names = {"f32w": "winograd_f32_kernel (fused Winograd F(2x2,3x3), fp32 MFMA)", "f32w4": "winograd43_f32_kernel (fused Winograd F(4x4,3x3), fp32 MFMA)"}
    fam = max(e["kernel_family"], key=e["kernel_family"].get)
    fam_name = {"ring": "conv_ring_kernel (persistent LDS-DMA ring)", "reg": "conv3x3_kernel_reg (register-staged)", "linear": "linear_kernel (token GEMM)", "linear_dma": "linear_dma_kernel (LDS-DMA staged token GEMM)",
                "dma": "conv3x3_kernel_dma"}.get(fam, fam)
    kname = names.get(key[0], f"{fam_name} <{key[0]}, cout-group {key[1]}, ksize {key[2]}>")
    mfma = e["bound"] == "mfma"
    r = {"bound": e["bound"], "kernel": kname,
         "achieved": e["executed_mfma_tflops"] if mfma else e["algorithmic_gbps"], "peak": PEAK_TFLOPS[key[0]] if mfma else PEAK_HBM_GBPS,
         "unit": "TFLOP/s" if mfma else "GB/s", "frac": e["mfma_frac"] if mfma else e["algorithmic_hbm_frac"],
         "traffic": e.get("traffic"), "hbm_frac": e.get("hbm_frac"), "algorithmic_bytes_per_launch": e["algorithmic_bytes_per_launch"],
         "traffic_over_algorithmic": e.get("traffic_over_algorithmic"),
         "flop_per_launch": agg[key][1] / agg[key][0], "launches": agg[key][0], "avg_launch_ms": e["avg_launch_ms"], "share_of_step": agg[key][2] / (nsteps_agg or args.steps) / (dt * 1e3),
         "algorithmic_tflops": e["algorithmic_tflops"], "algorithmic_vs_direct_peak": e["algorithmic_tflops"] / PEAK_TFLOPS[key[0]],
         "note": ("achieved / frac = FLOPs the kernel EXECUTES on the MFMA pipe (algorithmic 3x3-conv FLOPs / "
                  f"{WINO_FACTOR.get(key[0], 1.0):g}) over the launch time, against the dense {key[0][:3]} MFMA peak; algorithmic_* = the 2*9*Cin*Cout-per-pixel "
                  "rate of the convolution itself (DESIGN.md section 4)") if mfma else
                 "achieved / frac = algorithmic bytes (input once + every output / side view once) over the launch time, against 8 TB/s",
         "all": table}
    if replayed:
        r["measured_in"] = "eager warm-up step: the timed region replays the step as hipGraphs, where a launch cannot carry HIP events"
    if r["traffic"] is not None:
        # the PMC passes (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md) count every L2 miss, Infinity-Cache hits included
        r["traffic_note"] = ("traffic = L2-miss bytes per launch from the committed FETCH_SIZE / WRITE_SIZE passes; Infinity-Cache hits are counted.  "
                             + ("Per layer (profiles/r03_w43_traffic_per_layer.txt) it is input x 1.2 + output + the transformed-weight array re-streamed by each XCD "
                                "once per round of tiles (<= 37.7 MB, cache resident): the input is NOT re-fetched per output group" if key[0] == "f32w4" else ""))
    return r


# TRAINNER_BENCH_REHEARSE=1: run the multi-rank code path (own ranks, barrier + max-over-ranks timing, bucketed gradient exchange, rank-0 JSON line) on a
# box with fewer GPUs than ranks -- the ranks share the devices and talk over gloo.  The numbers of such a run mean nothing; it exists so that the
# path the driver launches on an 8-GPU node is executed by a test on the one-GPU box.
REHEARSE = os.environ.get("TRAINNER_BENCH_REHEARSE") == "1"


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per GPU, RCCL) as children through torch.distributed.run --
    decided before this process touches the GPU -- and relay their output; rank 0 prints the JSON line.  The ranks see the same argv."""
    import socket
    import subprocess

    ndev = torch.cuda.device_count()            # counting devices does not initialise the GPU
    if ndev < n and not REHEARSE:
        print(json.dumps({"error": f"--gpus {n} asked for, {ndev} visible"}))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *sys.argv[1:]]
    log(f"starting {n} ranks: {' '.join(cmd[1:8])} ...")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"note: --gpus {args.gpus} but the launcher started {world} rank(s); n_gpus reports the ranks RCCL sees")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print(json.dumps({"error": "bench.py needs an MI355X; this package has no CPU execution path"}))
        sys.exit(2)
    if REHEARSE:        # one-GPU rehearsal of the N-rank path (tests/test_dp_gpu.py): the ranks share the visible devices, gloo instead of RCCL
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if REHEARSE else "nccl")
        world = dist.get_world_size()
    dev = torch.device("cuda", local_rank)

    import trainner_redux_amd  # noqa: F401
    from trainner_redux_amd.models import build_model
    from trainner_redux_amd.utils.misc import set_random_seed

    opt = make_opt(args)
    set_random_seed(1024)   # same weights on every replica (DDP broadcasts rank 0's, base_model.py:228-234)
    model = build_model(opt)
    if world > 1:
        p, _ = model.net_g.flat_parameters()
        dist.broadcast(p, 0)
    torch.manual_seed(1024 + rank)
    n, s = args.batch, args.lq
    sc = args.scale
    gt = torch.rand(n, 3, sc * s, sc * s, device=dev)
    lq = torch.nn.functional.interpolate(gt, scale_factor=1.0 / sc, mode="bicubic", antialias=True).clamp(0, 1)
    batch = {"lq": lq, "gt": gt}
    if args.otf:
        # what RealESRGANDataset hands over (data/realesrgan_dataset.py): GT tiles of gt_size + 32 and three 21 x 21 kernels per sample, on the host
        import numpy as np

        from trainner_redux_amd.data.kernels import random_mixed_kernels

        rng = np.random.default_rng(1024 + rank)
        kinds, kp = ["iso", "aniso", "generalized_iso", "generalized_aniso", "plateau_iso", "plateau_aniso"], [0.45, 0.25, 0.12, 0.03, 0.12, 0.03]

        def kern(sig):
            ks = 2 * int(rng.integers(2, 9)) + 1
            k = random_mixed_kernels(kinds, kp, ks, sig, sig, (-3.14159, 3.14159), (0.5, 4), (1, 2), rng=rng)
            p = (21 - ks) // 2
            return torch.from_numpy(np.pad(k, ((p, p), (p, p)))).float()

        pulse = torch.zeros(21, 21); pulse[10, 10] = 1
        batch = {"gt": torch.rand(n, 3, sc * s + 32, sc * s + 32), "kernel1": torch.stack([kern((0.2, 2)) for _ in range(n)]),
                 "kernel2": torch.stack([kern((0.2, 1)) for _ in range(n)]), "sinc_kernel": pulse.repeat(n, 1, 1)}
    timer = ConvTimer()
    if not args.no_roofline:
        timer.install()

    def step(i):
        model.feed_data(batch)
        model.optimize_parameters(i, 0, True)
        model.update_learning_rate(i, -1)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"model built; warm-up {args.warmup} steps")
    # Two events around EVERY conv launch cost a few % of the 256 -> 1024 step and a third of the launch-bound 64 -> 256 steps (the timestamps serialise
    # back-to-back launches).  So: every group is timed in the LAST WARM-UP step (untimed region) -- that fills `roofline.all` and tells which group
    # dominates -- and inside the timed region only the dominant group's launches carry events (45 per step on the headline workload).
    it = 0
    warm_records = None
    for k in range(args.warmup):
        it += 1
        if (k >= 1 or args.warmup == 1) and not args.no_roofline:
            # every warm-up step after the first is timed; the latest one that recorded launches counts.  A generator that replays its step as
            # hipGraphs (SwinIR) launches nothing through the Python wrapper once captured: its last EAGER warm-up step supplies the table.
            fence()
            timer.enabled = True
            step(it)
            fence()
            timer.enabled = False
            if timer.records:
                warm_records = timer.records
            timer.records = []
        else:
            step(it)
    fence()
    if warm_records:
        wagg = timer.summary(warm_records)
        timer.only = max(wagg, key=lambda k: wagg[k][2])
    log(f"warm-up done; timing {args.steps} steps" + (f" (events on the {timer.only} launches only)" if timer.only else ""))
    timer.enabled = not args.no_roofline
    t0 = time.perf_counter()
    for _ in range(args.steps):
        it += 1
        step(it)
    fence()
    dt = time.perf_counter() - t0
    log(f"timed region {dt:.3f} s")
    timer.enabled = False
    stats = model.get_current_log()
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        out = {
            "metric": f"training images/sec ({s}->{sc * s} {sc}x SR) " + ("ESRGAN" if args.arch == "esrgan" else "SPAN" if args.arch.startswith("span") else "MoSR" if args.arch.startswith("mosr") else "SwinIR"),
            "value": n * world * args.steps / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.compute, "data": "synthetic",
            "config": {"workload": (f"ESRGAN 4x (RRDBNet {args.blocks} blocks)" if args.arch == "esrgan" else f"{args.arch} {sc}x")
                                   + f" {args.losses} {args.compute}" + (" +GAN(UNetDiscriminatorSN)" if args.gan else "") + (" +OTF feed (RealESRGANModel.feed_data per step, host codec round trip included)" if args.otf else "") + (" fast_matmul(split-bf16 VGG)" if args.fast_matmul else "") + (" hipGraph replay of G fwd/bwd" if args.graphs else "") + f", {s}->{sc * s} tiles, batch {n}/GPU, AdamW+EMA, optimize_parameters per step",
                       "global_batch": n * world, "lq_size": s, "parallelism": f"dp{world}", "losses": args.losses,
                       "l_g_total": stats.get("l_g_total")},
        }
        if not args.no_roofline and (timer.records or warm_records):
            out["roofline"] = roofline(args, timer.summary(), dt / args.steps, n, s, timer.summary(warm_records) if warm_records else None)
        if world == 1 and not args.fast_matmul and "perceptual" in args.losses and not args.no_variants:
            # informational second measurement (NOT `value`): the reference's `fast_matmul: true` option, which on gfx950 (no TF32) is the
            # split-bf16 path of the fp32-forced VGG head (3 bf16 MFMAs per product; measured 3e-6 rel-L2 per conv vs exact fp32)
            head = model.losses.get("l_g_perceptual")
            if head is not None and hasattr(head, "set_fast_matmul"):
                head.set_fast_matmul(True)
                it += 1
                step(it)
                fence()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    it += 1
                    step(it)
                fence()
                dt2 = time.perf_counter() - t1
                head.set_fast_matmul(False)
                out["variants"] = {"fast_matmul": {"value": n * args.steps / dt2, "unit": "images/sec", "ms_per_step": dt2 / args.steps * 1e3,
                                                   "what": "same workload with the reference option fast_matmul: true (split-bf16 MFMAs in the fp32-forced "
                                                           "VGG head; everything else unchanged); not the headline value"}}
        if args.loader and world == 1:
            out.setdefault("variants", {})["real_loader"] = real_loader_leg(args, model, n, s, sc, it)
        if not args.no_cpu_baseline and world == 1:
            log(f"cpu baseline on {host_threads()} threads")
            out["cpu_baseline"] = cpu_baseline(args, "perceptual" in args.losses)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
