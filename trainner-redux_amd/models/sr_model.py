"""`SRModel`: the trainer object whose `optimize_parameters` is the hot path.

Surface and step semantics follow /root/reference/traiNNer/models/sr_model.py (ctor :42-176, training setup :178-432,
`feed_data` :434-450, `optimize_parameters` :452-828) and base_model.py (`get_current_log` :186-200, `reduce_loss_dict`
:847-870, LR schedule :258-313,369-390).  What is different is *how* a step runs on MI355X:

  * no autograd graph for the generator: `net_g.forward_hip` -> loss heads write d(loss)/d(output) into one buffer
    (HIP heads fused with their reduction; foreign `nn.Module` losses through a one-node autograd on the output) ->
    `net_g.backward_hip` accumulates into the flat gradient arena;
  * data parallelism = one process per GPU, ONE RCCL all-reduce of the gradient arena (no DDP wrapper, no DataParallel);
  * grad-norm, clipping, AdamW, EMA and zero_grad are one fused sweep; loss scalars stay on the device and are read
    back only by `get_current_log()` (the reference syncs twice per step, :665 and :678);
  * the reference's host-side quirks that change numbers are kept: dict-valued losses are multiplied by `accum_iters`
    (:654-659), scalar ones divided (:661-663); the D loss is scaled by the G-side `loss_weight` (:763-768).
The NaN/Inf check of :665 is evaluated on the host at log time (same RuntimeError text) instead of every step.
"""
from __future__ import annotations

import math
import os
import os.path as osp
from collections import OrderedDict
from typing import Any

import torch
from torch import Tensor, nn

from .. import ops
from ..archs import build_network
from ..losses import build_loss
from ..optimizers import AdamW, build_optimizer
from ..utils.dist_util import GradExchange, get_dist_info, master_only
from ..utils.misc import loss_type_to_label
from .ema import EMA

ARCHS_WITHOUT_FP16_PREFIXES = ("swinir",)  # arch_info.py:48-50 (all swinir_* are bf16-only)


class LossScaler:
    """torch.amp.GradScaler's schedule (init 65536, x2 after 2000 clean steps, x0.5 on inf/nan) with the scale kept on the host: the
    scaled loss gradient is produced by the loss heads (grad_weight * scale), `unscale_` is folded into the AdamW sweep, and the
    inf check is the finiteness of the (already computed) gradient norm -- the reference's scaler.step() syncs on found_inf too."""

    def __init__(self, enabled: bool, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000):
        self.enabled, self.scale = enabled, float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor, self.growth_interval, self._growth_tracker = growth_factor, backoff_factor, growth_interval, 0

    def update(self, found_inf: bool) -> None:
        if not self.enabled:
            return
        if found_inf:
            self.scale *= self.backoff_factor
            self._growth_tracker = 0
        else:
            self._growth_tracker += 1
            if self._growth_tracker == self.growth_interval:
                self.scale *= self.growth_factor
                self._growth_tracker = 0

    def state_dict(self) -> dict:
        return {"scale": self.scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": self._growth_tracker} if self.enabled else {}

    def load_state_dict(self, sd: dict) -> None:
        if self.enabled and sd:
            self.scale, self._growth_tracker = float(sd["scale"]), int(sd["_growth_tracker"])


class SRModel:
    def __init__(self, opt) -> None:
        self.opt = opt
        self.is_train = bool(opt.is_train)
        if not torch.cuda.is_available() or opt.num_gpu == 0:
            raise RuntimeError("SRModel (HIP) needs an MI355X: this package has no CPU execution path")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.use_amp = bool(opt.use_amp)
        self.amp_bf16 = bool(opt.amp_bf16)
        # use_amp without amp_bf16 = fp16 autocast + GradScaler in the reference (sr_model.py:188-191): fp16 MFMA operands + loss scaling here
        self.compute_dtype = (torch.bfloat16 if self.amp_bf16 else torch.float16) if self.use_amp else torch.float32
        self.amp_dtype = self.compute_dtype
        fp16 = self.use_amp and not self.amp_bf16
        self.scaler_g = LossScaler(enabled=fp16)
        self.scaler_d = LossScaler(enabled=fp16)
        self.schedulers: list = []
        self.optimizers: list = []
        self.optimizers_skipped: list[bool] = []
        self.log_dict: dict[str, Any] = {}
        self.loss_samples = 0
        self.lq: Tensor | None = None
        self.gt: Tensor | None = None
        self.output: Tensor | None = None
        self.with_metrics = False
        self.first_val_completed = False
        self.metric_results: dict = {}
        self.best_metric_results: dict = {}
        self.training_automation_manager = None
        self.batch_augment = None
        self.dynamic_loss_scheduler = None

        # input / output pixel formats (sr_model.py:471-481, color_util.py:237-272): "rgb" is the identity; "yuv444" converts LQ to BT.601
        # YCbCr in front of the generator and its output back to RGB (losses see RGB); the 1- and 2-channel formats ("y", "uv") would need
        # generators with other channel counts than the HIP ones build
        fin, fout = getattr(opt, "input_pixel_format", "rgb") or "rgb", getattr(opt, "output_pixel_format", "rgb") or "rgb"
        for f in (fin, fout):
            if f not in ("rgb", "yuv444"):
                raise NotImplementedError(f"pixel format {f!r} is not built on the HIP path (rgb, yuv444)")
        self._yuv_in, self._yuv_out = fin == "yuv444", fout == "yuv444"
        assert opt.network_g is not None, "network_g must be defined"
        self.net_g = build_network({**opt.network_g, "scale": opt.scale}).to(self.device)
        if hasattr(self.net_g, "compute_dtype"):
            self.net_g.compute_dtype = self.compute_dtype
        if hasattr(self.net_g, "hip_graphs") and os.environ.get("TRAINNER_HIP_GRAPHS") != "0" and \
                (getattr(opt, "use_compile", False) or getattr(self.net_g, "prefer_hip_graphs", False)):
            # the reference's `use_compile` (torch.compile around the networks, base_model.py:214-226: spend warm-up time, run steps faster)
            # maps onto hipGraph replay of the generator's forward / backward launches -- there is no tracing compiler on this path.  Generators whose
            # training step is launch- / latency-bound at training tile sizes (SwinIR: ~1,900 launches of ~15 us) ask for it themselves.
            self.net_g.hip_graphs = True
        self.net_d: nn.Module | None = None
        self.net_g_ema: EMA | None = None
        self.losses: "OrderedDict[str, nn.Module]" = OrderedDict()
        self.optimizer_g = None
        self.optimizer_d = None
        self.has_gan = False
        self.accum_iters = 1
        self.grad_clip = False
        self.ema_decay = 0
        if opt.path.pretrain_network_g is not None:
            self.load_network(self.net_g, opt.path.pretrain_network_g, opt.path.strict_load_g, opt.path.param_key_g)
        if self.is_train:
            self._detect_gan()
            self.init_training_settings()

    # ------------------------------------------------------------------ setup ------------------------------
    def _detect_gan(self) -> None:
        train_opt = self.opt.train
        for loss in (train_opt.losses or []):
            if loss_type_to_label(loss["type"]) in ("l_g_gan", "l_g_featurematching") and float(loss.get("loss_weight", 0)) != 0:
                self.has_gan = True
        if train_opt.gan_opt is not None and float(train_opt.gan_opt.get("loss_weight", 0)) != 0:
            self.has_gan = True
        if self.has_gan:
            if train_opt.optim_d is None:
                raise ValueError("GAN loss requires discriminator optimizer (optim_d)")   # sr_model.py:147-154
            if self.opt.network_d is None:
                raise ValueError("GAN loss requires discriminator network (network_d)")
            self.net_d = build_network(self.opt.network_d).to(self.device)
            if hasattr(self.net_d, "compute_dtype"):
                self.net_d.compute_dtype = self.compute_dtype
            if self.opt.path.pretrain_network_d is not None:
                self.load_network(self.net_d, self.opt.path.pretrain_network_d, self.opt.path.strict_load_d, self.opt.path.param_key_d)

    def init_training_settings(self) -> None:
        train_opt = self.opt.train
        assert train_opt is not None
        self.net_g.train()
        if self.net_d is not None:
            self.net_d.train()
        ds = self.opt.datasets.get("train")
        self.accum_iters = ds.accum_iter if ds is not None else 1
        self.ema_decay = train_opt.ema_decay
        if self.ema_decay > 0:
            init_ema = None
            if self.opt.path.pretrain_network_g_ema is not None:
                init_ema = build_network({**self.opt.network_g, "scale": self.opt.scale}).to(self.device)
                self.load_network(init_ema, self.opt.path.pretrain_network_g_ema, self.opt.path.strict_load_g, "params_ema")
            switch_iter = train_opt.ema_switch_iter or None
            self.net_g_ema = EMA(self.net_g, ema_model=init_ema, beta=self.ema_decay, allow_different_devices=True,
                                 update_after_step=train_opt.ema_update_after_step, update_every=1, power=train_opt.ema_power,
                                 update_model_with_ema_every=switch_iter).to(self.device)
            self.net_g_ema.step = self.net_g_ema.step.cpu()
            self.net_g_ema.initted = self.net_g_ema.initted.cpu()
            if hasattr(self.net_g_ema.ema_model, "compute_dtype"):
                self.net_g_ema.ema_model.compute_dtype = self.compute_dtype
        # adaptive_d (sr_model.py:195-197, 558-569): the D update is skipped on steps where the EMA of the generator-side GAN loss rises by more
        # than `adaptive_d_threshold`; the EMA lives on the device (base_model.py:55), the comparison is the reference's one host read per step
        self.adaptive_d = bool(train_opt.adaptive_d)
        self.adaptive_d_ema_decay = float(train_opt.adaptive_d_ema_decay)
        self.adaptive_d_threshold = float(train_opt.adaptive_d_threshold)
        self.l_g_gan_ema = torch.zeros(1, device=self.device)
        # options that change the step and are not built must not be ignored silently
        if train_opt.use_moa:
            raise NotImplementedError("use_moa (batch mixing augmentations, sr_model.py:434-450) is not on the HIP path yet")
        dls = train_opt.dynamic_loss_scheduling
        if isinstance(dls, dict) and dls.get("enabled"):
            raise NotImplementedError("dynamic_loss_scheduling is not on the HIP path yet")
        ta = train_opt.training_automations
        if isinstance(ta, dict) and ta.get("enabled"):
            raise NotImplementedError("training_automations belong to the reference's training loop (train.py), which this package does not replace")
        self.grad_clip = train_opt.grad_clip
        self.grad_clip_threshold = 1.0   # get_automation_clipping_threshold() default without automations
        if train_opt.losses is None:
            train_opt.losses = [getattr(train_opt, k) for k in train_opt.legacy_loss_opts if getattr(train_opt, k) is not None]
        for loss in train_opt.losses:
            assert "type" in loss, "all losses must define type"
            assert "loss_weight" in loss, f"{loss['type']} must define loss_weight"
            if float(loss["loss_weight"]) != 0:
                label = loss_type_to_label(loss["type"])
                self.losses[label] = build_loss(loss).to(self.device)
                if hasattr(self.losses[label], "set_compute_dtype"):
                    self.losses[label].set_compute_dtype(self.compute_dtype)
                if hasattr(self.losses[label], "set_fast_matmul"):
                    self.losses[label].set_fast_matmul(bool(getattr(self.opt, "fast_matmul", False)))
        assert self.losses, "At least one loss must be defined."
        self.setup_optimizers()
        self.setup_schedulers()
        dev = self.device
        self._scal = torch.zeros(64, dtype=torch.float32, device=dev)   # device-resident loss scalars
        self._gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._zero1 = torch.zeros(1, dtype=torch.float32, device=dev)
        self._pending: list = []

    def setup_optimizers(self) -> None:
        train_opt = self.opt.train
        assert train_opt.optim_g is not None
        params = [p for p in self.net_g.parameters() if p.requires_grad]
        if hasattr(self.net_g, "flat_parameters"):
            self.net_g.flat_parameters()
            params = [p for p in self.net_g.parameters() if p.requires_grad]
        self.optimizer_g = build_optimizer(params, train_opt.optim_g)
        self.optimizers.append(self.optimizer_g)
        self.optimizers_skipped.append(False)
        if self.net_d is not None and train_opt.optim_d is not None:
            if hasattr(self.net_d, "flat_parameters"):
                self.net_d.flat_parameters()
            self.optimizer_d = build_optimizer(list(self.net_d.parameters()), train_opt.optim_d)
            self.optimizers.append(self.optimizer_d)
            self.optimizers_skipped.append(False)

    def setup_schedulers(self) -> None:
        sch = self.opt.train.scheduler
        if sch is None:
            return
        sch = dict(sch)
        kind = sch.pop("type")
        for o in self.optimizers:
            if kind in ("MultiStepLR", "MultiStepRestartLR"):
                self.schedulers.append(torch.optim.lr_scheduler.MultiStepLR(o, milestones=sch["milestones"], gamma=sch.get("gamma", 0.1)))
            elif kind == "CosineAnnealingLR":
                self.schedulers.append(torch.optim.lr_scheduler.CosineAnnealingLR(o, **sch))
            elif kind == "ConstantLR":
                self.schedulers.append(torch.optim.lr_scheduler.ConstantLR(o, **sch))
            else:
                raise NotImplementedError(f"Scheduler {kind} is not implemented yet.")
        self._init_lr = [[g["lr"] for g in o.param_groups] for o in self.optimizers]

    # ------------------------------------------------------------------ loop surface -------------------------
    def feed_data(self, data: dict) -> None:
        """sr_model.py:434-450.  Besides the reference's fp32 CHW DataFeed this accepts uint8 (N, H, W, 3) patches (what
        `data.DevicePrefetcher` ships: a quarter of the PCIe bytes) and converts them on the device with `tr_u8hwc_to_nchw`."""
        assert "lq" in data
        self.lq = self._to_device_f32(data["lq"])
        if "gt" in data:
            self.gt = self._to_device_f32(data["gt"])

    def _to_input_format(self, img: torch.Tensor) -> torch.Tensor:
        from .. import ops

        return ops.color601(img, 0) if self._yuv_in else img

    def _to_rgb(self, img: torch.Tensor) -> torch.Tensor:
        from .. import ops

        return ops.color601(img, 1) if self._yuv_out else img

    def _to_device_f32(self, t: torch.Tensor) -> torch.Tensor:
        t = t.to(self.device, non_blocking=True)
        if t.dtype == torch.uint8:
            from .. import ops

            out = torch.empty(t.shape[0], t.shape[3], t.shape[1], t.shape[2], dtype=torch.float32, device=self.device)
            ops.u8hwc_to_nchw(t.contiguous(), out)
            return out
        return t

    def get_bare_model(self, net):
        return net

    def _exchange(self, net) -> GradExchange:
        """The (cached) bucketed gradient exchange of a flat-arena network; rebuilt when the arena moved."""
        _, garr = net.flat_parameters()
        ex = getattr(net, "grad_exchange", None)
        if ex is None or ex.flat.data_ptr() != garr.data_ptr() or ex.flat.numel() != garr.numel():
            ex = GradExchange(garr, bucket_bytes=int(os.environ.get("TRAINNER_BUCKET_MB", "24")) << 20)
            net.grad_exchange = ex
        return ex

    def get_current_learning_rate(self) -> list[float]:
        return [g["lr"] for g in self.optimizers[0].param_groups]

    def update_learning_rate(self, current_iter: int, warmup_iter: int = -1) -> None:
        # base_model.py:369-390: every call steps every scheduler unless that optimizer's AMP step was skipped (optimizers_skipped is only ever set by a
        # loss-scaler overflow), then the linear warm-up overrides the groups' lr from their INITIAL lr
        for i, s in enumerate(self.schedulers):
            if not self.optimizers_skipped[i]:
                s.step()
        if current_iter < warmup_iter and self.schedulers:
            for o, init in zip(self.optimizers, self._init_lr):
                for g, lr0 in zip(o.param_groups, init):
                    g["lr"] = lr0 / warmup_iter * current_iter

    def reset_current_log(self) -> None:
        self.log_dict = {}
        self.loss_samples = 0

    def _flush_pending(self) -> None:
        """Fold device-resident scalars of finished steps into log_dict (one D2H copy per call)."""
        if not self._pending:
            return
        torch.cuda.current_stream().synchronize()
        for keys, idx, vals, n in self._pending:
            block = vals.tolist()
            for k, v in zip(keys, (block[i] for i in idx)):
                if k == "l_g_total" and not (v == v and abs(v) != float("inf")):
                    self._pending = []
                    raise RuntimeError("Training failed: NaN/Inf found in loss.")
                self.log_dict[k] = self.log_dict.get(k, 0.0) + v * n
        self._pending = []

    def get_current_log(self) -> dict[str, float]:
        self._flush_pending()
        out = {}
        for k, v in self.log_dict.items():
            out[k] = (float(v) / self.loss_samples) if self.loss_samples > 0 else float(v)
        rank, world = get_dist_info()
        if world > 1:  # base_model.py:847-870 (mean over ranks), done at log time instead of every step
            keys = sorted(out)
            t = torch.tensor([out[k] for k in keys], dtype=torch.float32, device=self.device)
            torch.distributed.all_reduce(t)
            out = {k: float(v) / world for k, v in zip(keys, t.tolist())}
        return out

    # ------------------------------------------------------------------ THE HOT PATH ---------------------------
    def _adaptive_d_skip(self, l_g_gan: Tensor) -> bool:
        """sr_model.py:558-569: ema' = decay * ema + (1 - decay) * l_g_gan; skip the discriminator update of this step (and its scheduler step,
        base_model.py:378) when ema' > ema * adaptive_d_threshold.  `l_g_gan` is a 1-element device tensor; the comparison is a host read,
        as in the reference."""
        ema = self.adaptive_d_ema_decay * self.l_g_gan_ema + (1.0 - self.adaptive_d_ema_decay) * l_g_gan
        skip = bool(ema > self.l_g_gan_ema * self.adaptive_d_threshold)
        self.l_g_gan_ema = ema
        if skip and len(self.optimizers_skipped) > 1:
            self.optimizers_skipped[1] = True
        return skip

    def optimize_parameters(self, current_iter: int, current_accum_iter: int, apply_gradient: bool) -> None:
        assert self.lq is not None and self.gt is not None
        if self.net_d is not None and not hasattr(self.net_d, "forward_hip"):
            for p in self.net_d.parameters():
                p.requires_grad = False
        n_samples = self.gt.shape[0]
        self.loss_samples += n_samples
        net = self.net_g
        hip_net = hasattr(net, "forward_hip")
        lq, gt = self.lq, self.gt
        if lq.dtype != torch.float32:
            lq, gt = lq.float(), gt.float()
        gt = gt.contiguous()

        lq = self._to_input_format(lq)       # sr_model.py:471: rgb2pixelformat_pt (identity for "rgb")
        if hip_net:
            out = net.train_forward(lq) if hasattr(net, "train_forward") else net.forward_hip(lq, train=True)
            out = self._to_rgb(out)          # :479-481: pixelformat2rgb_pt
            out_leaf = None
        else:  # foreign generator (stock nn.Module from the open plugin surface): ordinary autograd
            if self._yuv_in or self._yuv_out:
                raise NotImplementedError("yuv444 pixel formats are wired for the HIP generators only")
            with torch.autocast(device_type="cuda", dtype=self.amp_dtype, enabled=self.use_amp):
                out = net(lq)
        self.output = out.detach()
        gout = torch.empty_like(self.output)
        first = True                      # gout not written yet
        scal = self._scal                 # device-resident scalars of this step
        scal.zero_()
        TOTAL = 63                        # scal[TOTAL] accumulates l_g_total
        total = scal[TOTAL:TOTAL + 1]
        keys: list[str] = []              # log key -> scal[i]
        acc = self.accum_iters
        skip_d_update = False
        foreign_total = None
        S = self.scaler_g.scale            # loss scale (1.0 unless fp16): only the GRADIENT weights carry it, logged values do not
        if self.scaler_g.enabled and not hip_net:
            raise NotImplementedError("fp16 + loss scaling is built for the HIP generators only; use amp_bf16: true with foreign modules")

        lq_target = None
        for label, loss in self.losses.items():
            # IterativeLossWrapper (losses/__init__.py:79-110): the wrapped value is loss * schedule(iter), SRModel multiplies by abs(loss_weight)
            # on top (sr_model.py:623-663) -- both are host scalars here and reach the head as one weight
            wrapped = hasattr(loss, "loss_module")
            head = loss.loss_module if wrapped else loss
            wb = abs(loss.loss_weight)
            sched = loss.step_weight(current_iter) if wrapped else 1.0
            w = wb * sched
            target = gt
            if loss.loss_weight < 0:
                # sr_model.py:503-516: a negative weight selects the bicubic-upsampled LQ as the target of that loss (its magnitude is the weight):
                # clamp(F.interpolate(lq, scale_factor=scale, mode="bicubic", antialias=True), 0, 1) -- the separable gather of the OTF resize
                # (tr_resample_axis over ATen's antialias tables, data/otf.py) computes exactly that
                if lq_target is None:
                    from ..data.otf import resize_pt

                    lq_target = resize_pt(self.lq.float(), "bicubic", size=(self.lq.shape[2] * self.opt.scale, self.lq.shape[3] * self.opt.scale)).contiguous()
                target = lq_target
            slot = len(keys)
            if label == "l_g_gan" and hasattr(head, "hip_value_and_grad") and hasattr(self.net_d, "forward_hip"):
                # generator-side GAN term, all on HIP: D(output) -> loss(pred, real) -> dL/dlogits -> D backward to the image only
                logits = self.net_d.forward_hip(self.output, train=True)
                g_logits = torch.empty_like(logits)
                head.hip_value_and_grad(logits, True, False, value_weight=w, value_out=scal[slot:], total_weight=w / acc, total_out=total,
                                        grad_weight=w / acc * S, grad=g_logits)
                self.net_d.backward_hip(g_logits, param_grads=False, input_grad=gout, input_grad_accumulate=not first)
                keys.append(label)
                first = False
                if self.adaptive_d:
                    # the reference's l_g_loss is the value before abs(loss_weight) (wrapped: times the schedule factor); scal[slot] holds it times wb
                    skip_d_update = self._adaptive_d_skip(scal[slot:slot + 1] / wb)
            elif hasattr(head, "hip_value_and_grad") and label != "l_g_gan":
                # fused head.  scalar head: total += loss*w/acc ; dict head: total += v*w*acc  (sr_model.py:654-663)
                is_dict = getattr(head, "returns_dict", False)
                k = float(acc) if is_dict else 1.0 / acc
                subs = head.hip_value_and_grad(self.output, target, value_weight=w, value_out=scal[slot:], total_weight=w * k, total_out=total,
                                               grad_weight=w * k * S, grad=gout, first=first)
                keys += [f"{label}_{sub}" for sub in subs] if subs else [label]
                first = False
            else:
                # foreign head (any registered nn.Module, incl. GAN on a stock discriminator): one-node autograd from the output
                if out_leaf is None:
                    out_leaf = self.output.clone().requires_grad_(True) if hip_net else out
                with torch.autocast(device_type="cuda", dtype=self.amp_dtype, enabled=self.use_amp):
                    it = {"current_iter": current_iter} if wrapped else {}       # the wrapper's forward applies the schedule factor itself
                    if label == "l_g_gan":
                        assert self.net_d is not None
                        val = loss(self.net_d(out_leaf), True, is_disc=False, **it)
                        if self.adaptive_d:
                            skip_d_update = self._adaptive_d_skip(val.detach().float().reshape(1))
                    else:
                        val = loss(out_leaf, target, **it)
                items = [(f"{label}_{sub}", v, float(acc)) for sub, v in val.items()] if isinstance(val, dict) else [(label, val, 1.0 / acc)]
                for key, v, k in items:
                    wv = v * wb
                    foreign_total = wv * k if foreign_total is None else foreign_total + wv * k
                    scal[len(keys)] = wv.detach().float()
                    keys.append(key)

        if foreign_total is not None:
            total += foreign_total.detach().float()
            if hip_net:
                (g_foreign,) = torch.autograd.grad(foreign_total * S, out_leaf)
                if first:
                    gout.copy_(g_foreign)
                else:
                    flat_g, flat_f = gout.view(-1), g_foreign.contiguous().view(-1)
                    n4 = flat_g.numel() // 4
                    ops.axpby(ops.view(flat_g[: n4 * 4].view(n4, 4)), 1.0, ops.view(flat_f[: n4 * 4].view(n4, 4)), 1.0,
                              ops.view(flat_g[: n4 * 4].view(n4, 4)), n4)
                    if flat_g.numel() % 4:
                        flat_g[n4 * 4:] += flat_f[n4 * 4:]
                first = False
            else:
                foreign_total.backward()
        nloss = len(keys)
        assert nloss < 56, "too many loss terms for the scalar block"
        GN_G, LD_REAL, LD_FAKE, GN_D = 56, 57, 58, 59
        idx = list(range(nloss)) + [TOTAL]
        keys.append("l_g_total")

        rank, world = get_dist_info()
        if hip_net:
            if self._yuv_out:
                gout = ops.color601(gout, 2)     # d loss / d(RGB) -> d loss / d(YCbCr network output)
            # data parallel (DDP's reducer, base_model.py:228-234): the gradient arena is exchanged in a few large buckets, each launched by
            # backward_hip as soon as the parameters it covers are final.  Only on steps that apply the gradient: with accum_iter > 1 the
            # micro-steps accumulate locally (the reference reduces on every micro-batch because it never uses no_sync -- same result).
            ex_g = self._exchange(net) if (world > 1 and apply_gradient) else None
            if ex_g is not None:
                ex_g.begin()
            if hasattr(net, "train_backward"):
                net.train_backward(gout)
            else:
                net.backward_hip(gout)

        if apply_gradient:
            if hip_net:
                parr, garr = net.flat_parameters()
                if ex_g is not None:
                    ex_g.finish()                # remaining buckets, wait, 1/world (DDP averages)
                ops.sumsq_norm(garr, scal[GN_G:GN_G + 1])
                self._gnorm = scal[GN_G:GN_G + 1]
                # sr_model.py:665-666 raises on a non-finite l_g_total before backward; the total lives on the device here, so it poisons the
                # norm instead (x * 0 is NaN for NaN / Inf, 0 otherwise) and tr_adamw_ema skips the update on a non-finite norm
                self._gnorm.addcmul_(total, self._zero1)
                keys.append("grad_norm_g")
                idx.append(GN_G)
                found_inf = False
                if self.scaler_g.enabled:           # GradScaler.step(): skip the update when the scaled gradients overflowed
                    found_inf = not math.isfinite(float(self._gnorm))
                if found_inf:
                    garr.zero_()
                    self.optimizers_skipped[0] = True
                else:
                    ema_arena, ema_decay = None, 1.0
                    if self.net_g_ema is not None:
                        d = self.net_g_ema.advance()
                        ema_arena = self.net_g_ema.ema_arena()
                        deferred_ema = None
                        if d is None:
                            ema_arena = None                 # no EMA update this step (update_every)
                        elif ema_arena is None:
                            deferred_ema = d                 # no arena to fold the lerp into: stand-alone lerp after the optimizer step
                        else:
                            ema_decay = d
                    if isinstance(self.optimizer_g, AdamW):
                        self.optimizer_g.set_step_extras(ema_arena=ema_arena, ema_decay=ema_decay, inv_scale=1.0 / S,
                                                         clip=self.grad_clip_threshold if self.grad_clip else 0.0, grad_norm=self._gnorm)
                        self.optimizer_g.step()
                        if self.net_g_ema is not None and deferred_ema is not None:
                            self.net_g_ema._lerp(deferred_ema)
                    else:
                        if S != 1.0:
                            garr.mul_(1.0 / S)
                        if self.grad_clip:
                            torch.nn.utils.clip_grad_norm_(net.parameters(), self.grad_clip_threshold)
                        self.optimizer_g.step()
                        garr.zero_()
                        if self.net_g_ema is not None and (ema_arena is not None or deferred_ema is not None):
                            self.net_g_ema._lerp(ema_decay if deferred_ema is None else deferred_ema)
                    if self.net_g_ema is not None:
                        self.net_g_ema.maybe_update_model_with_ema()   # ema_switch_iter (sr_model.py:223-234): EMA -> online every N steps
                    if S != 1.0:
                        self._gnorm.mul_(1.0 / S)       # logged norm = after unscale_ (sr_model.py:688-705); stream-ordered after the sweep
                    self.optimizers_skipped[0] = False
                self.scaler_g.update(found_inf)
            else:
                if world > 1:
                    for p in net.parameters():
                        if p.grad is not None:
                            torch.distributed.all_reduce(p.grad)
                            p.grad.div_(world)
                if self.grad_clip:
                    torch.nn.utils.clip_grad_norm_(net.parameters(), self.grad_clip_threshold)
                self.optimizer_g.step()
                self.optimizer_g.zero_grad()
                if self.net_g_ema is not None:
                    self.net_g_ema.update()
                self.optimizers_skipped[0] = False

        # ---- discriminator update (sr_model.py:725-808) ------------------------------------------------------------------
        cri_gan = self.losses.get("l_g_gan")
        if self.net_d is not None and cri_gan is not None and self.optimizer_d is not None and not skip_d_update:
            # the reference scales the D loss by the G-side weight, or by the schedule's weight of this iteration when wrapped (sr_model.py:763-768)
            wd = cri_gan.get_current_weight(current_iter) if hasattr(cri_gan, "get_current_weight") else cri_gan.loss_weight
            cri_gan = getattr(cri_gan, "loss_module", cri_gan)       # D-side values come from the unwrapped module (:739-741)
            if hasattr(self.net_d, "forward_hip") and hasattr(cri_gan, "hip_value_and_grad"):
                dnet = self.net_d
                for img, is_real, slot_ in ((gt, True, LD_REAL), (self.output, False, LD_FAKE)):
                    logits = dnet.forward_hip(img, train=True)
                    g_logits = torch.empty_like(logits)
                    cri_gan.hip_value_and_grad(logits, is_real, True, value_weight=1.0, value_out=scal[slot_:], grad_weight=wd / acc * self.scaler_d.scale,
                                               grad=g_logits)
                    dnet.backward_hip(g_logits, param_grads=True)
                keys += ["l_d_real", "l_d_fake"]
                idx += [LD_REAL, LD_FAKE]
                if apply_gradient:
                    parr_d, garr_d = dnet.flat_parameters()
                    if world > 1:                   # the discriminator's arena (17.5 MB for UNetDiscriminatorSN): one bucket after its two backward passes
                        ex_d = self._exchange(dnet)
                        ex_d.begin()
                        ex_d.finish()
                    ops.sumsq_norm(garr_d, scal[GN_D:GN_D + 1])
                    keys.append("grad_norm_d")
                    idx.append(GN_D)
                    Sd = self.scaler_d.scale
                    found_inf_d = self.scaler_d.enabled and not math.isfinite(float(scal[GN_D]))
                    if found_inf_d:
                        garr_d.zero_()
                    elif isinstance(self.optimizer_d, AdamW):
                        self.optimizer_d.set_step_extras(clip=self.grad_clip_threshold if self.grad_clip else 0.0, grad_norm=scal[GN_D:GN_D + 1],
                                                         inv_scale=1.0 / Sd)
                        self.optimizer_d.step()
                    else:
                        if Sd != 1.0:
                            garr_d.mul_(1.0 / Sd)
                        if self.grad_clip:
                            torch.nn.utils.clip_grad_norm_(dnet.parameters(), self.grad_clip_threshold)
                        self.optimizer_d.step()
                        garr_d.zero_()
                    if Sd != 1.0 and not found_inf_d:
                        scal[GN_D:GN_D + 1].mul_(1.0 / Sd)
                    self.scaler_d.update(found_inf_d)
                    if len(self.optimizers_skipped) > 1:
                        self.optimizers_skipped[-1] = found_inf_d
            else:   # stock-PyTorch discriminator from the open plugin surface: ordinary autograd
                for p in self.net_d.parameters():
                    p.requires_grad = True
                with torch.autocast(device_type="cuda", dtype=self.amp_dtype, enabled=self.use_amp):
                    real_d_pred = self.net_d(gt)
                    fake_d_pred = self.net_d(self.output.detach())
                    l_d_real = cri_gan(real_d_pred, True, is_disc=True)
                    l_d_fake = cri_gan(fake_d_pred, False, is_disc=True)
                    l_d_total = (l_d_real + l_d_fake) * wd
                (l_d_total / acc).backward()
                scal[LD_REAL] = l_d_real.detach().float()
                scal[LD_FAKE] = l_d_fake.detach().float()
                keys += ["l_d_real", "l_d_fake"]
                idx += [LD_REAL, LD_FAKE]
                if apply_gradient:
                    if world > 1:
                        for p in self.net_d.parameters():
                            if p.grad is not None:
                                torch.distributed.all_reduce(p.grad)
                                p.grad.div_(world)
                    gn = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.grad, 2) for p in self.net_d.parameters()
                                                               if p.grad is not None]))
                    scal[GN_D] = gn
                    keys.append("grad_norm_d")
                    idx.append(GN_D)
                    if self.grad_clip:
                        torch.nn.utils.clip_grad_norm_(self.net_d.parameters(), self.grad_clip_threshold)
                    self.optimizer_d.step()
                    self.optimizer_d.zero_grad()
            if apply_gradient and len(self.optimizers_skipped) > 1 and not (hasattr(self.net_d, "forward_hip") and hasattr(cri_gan, "hip_value_and_grad")):
                self.optimizers_skipped[-1] = False

        # device-resident log values of this step; read back lazily by get_current_log().  The whole 64-float block is cloned on the stream
        # and indexed on the host at read-back: `scal[idx]` with a Python list uploads an index tensor, i.e. a synchronous host -> device copy
        # that drained the launch queue at the end of EVERY step (found with torch.cuda.set_sync_debug_mode, tools/find_sync.py)
        self._pending.append((keys, idx, scal.clone(), n_samples))
        if len(self._pending) >= 64:
            self._flush_pending()

    # ------------------------------------------------------------------ inference / io -----------------------------
    @torch.no_grad()
    def test(self) -> None:
        """sr_model.py:902-945: EMA net if present, whole image or `val.tile_size` tiles."""
        net = self.net_g_ema.ema_model if self.net_g_ema is not None else self.net_g
        assert self.lq is not None
        lq = self._to_input_format(self.lq.float())              # sr_model.py:912-914
        val = self.opt.val
        if val is not None and (val.tile_size or 0) > 0:
            out = self.infer_tiled(net, lq)
        else:
            out = net.forward_hip(lq, train=False) if hasattr(net, "forward_hip") else net(lq)
        self.output = self._to_rgb(out)                          # :936-938

    @torch.no_grad()
    def infer_tiled(self, net: nn.Module, lq: Tensor) -> Tensor:
        """sr_model.py:830-900: reflect-pad to a tile multiple, run `tile_size` tiles with `tile_overlap`, blend with triangular weights
        (`tr_tile_blend`), divide by the weight map (`tr_tile_normalize`).  Tile geometry and edge handling as in the reference."""
        from .. import _lib as L

        tile, overlap, scale = self.opt.val.tile_size, self.opt.val.tile_overlap or 0, self.opt.scale
        fwd = (lambda t: net.forward_hip(t.contiguous(), train=False)) if hasattr(net, "forward_hip") else net
        b, c, h, w = lq.shape
        assert b == 1, "Only batch size 1 is supported for tiled inference"
        if h <= tile and w <= tile:
            return fwd(lq)
        pad_h = (tile - (h % tile)) % tile if h > tile else 0
        pad_w = (tile - (w % tile)) % tile if w > tile else 0
        lq = torch.nn.functional.pad(lq, (0, pad_w, 0, pad_h), mode="reflect")
        _, _, hp, wp = lq.shape
        acc = torch.zeros(c, hp * scale, wp * scale, device=lq.device)
        wmap = torch.zeros_like(acc)
        stride = tile - overlap
        tiles_y = max(1, (hp - overlap + stride - 1) // stride)
        tiles_x = max(1, (wp - overlap + stride - 1) // stride)
        lib, st = L.lib(), ops.stream_ptr
        for y in range(tiles_y):
            for x in range(tiles_x):
                y0, x0 = y * stride, x * stride
                patch = lq[:, :, y0:min(y0 + tile, hp), x0:min(x0 + tile, wp)]
                ph, pw = patch.shape[-2:]
                pb = max(tile - ph, 0) if ph < tile else 0
                pr = max(tile - pw, 0) if pw < tile else 0
                if pb > 0 or pr > 0:
                    patch = torch.nn.functional.pad(patch, (0, min(pr, pw - 1), 0, min(pb, ph - 1)), mode="reflect")
                out = fwd(patch).contiguous()
                L.check(lib.tr_tile_blend(st(), out.data_ptr(), c, out.shape[2], out.shape[3], acc.data_ptr(), wmap.data_ptr(), hp * scale, wp * scale,
                                          y0 * scale, x0 * scale, ph * scale, pw * scale, tile * scale), "tr_tile_blend")
        res = torch.empty(1, c, h * scale, w * scale, device=lq.device)
        L.check(lib.tr_tile_normalize(st(), acc.data_ptr(), wmap.data_ptr(), c, hp * scale, wp * scale, res.data_ptr(), h * scale, w * scale), "tr_tile_normalize")
        return res

    def validation(self, dataloader, current_iter: int, tb_logger=None, save_img: bool = False, multi_val_datasets: bool = False) -> None:
        """base_model.py `validation`: rank 0 validates (dist_validation, sr_model.py:947-958)."""
        rank, _ = get_dist_info()
        if rank == 0:
            self.nondist_validation(dataloader, current_iter, tb_logger, save_img, multi_val_datasets)

    def nondist_validation(self, dataloader, current_iter: int, tb_logger=None, save_img: bool = False, multi_val_datasets: bool = False) -> None:
        """sr_model.py:960-1139: per validation pair feed_data -> test -> the `val.metrics` on the device; means over the loader in
        `metric_results`, running best per dataset in `best_metric_results`; with `save_img` the 8-bit SR image goes to the reference's
        path under `path.visualization` (sr_model.py:1053-1113; PNG through PIL instead of cv2: same pixels)."""
        from ..metrics import calculate_metric

        val = self.opt.val
        assert val is not None
        ds = dataloader.dataset
        name = (ds.opt.get("name") if isinstance(getattr(ds, "opt", None), dict) else getattr(getattr(ds, "opt", None), "name", None)) or "val"
        with_metrics = bool(val.metrics_enabled) and val.metrics is not None
        if save_img and len(dataloader) > 0:
            if getattr(self.opt.path, "visualization", None) is None:
                raise ValueError("save_img needs path.visualization (sr_model.py:972)")
            os.makedirs(self.opt.path.visualization, exist_ok=True)
        if with_metrics:
            if not self.metric_results:
                self.metric_results = dict.fromkeys(val.metrics.keys(), 0)
            self._initialize_best_metric_results(name)
            self.metric_results = dict.fromkeys(self.metric_results, 0)
        n = 0
        for data in dataloader:
            self.feed_data(data)
            self.test()
            n += 1
            if save_img:
                self._dump_validation_image(data, dataloader, name, current_iter, multi_val_datasets)
            if with_metrics:
                for key, mopt in val.metrics.items():
                    self.metric_results[key] += calculate_metric({"img": self.output, "img2": self.gt}, mopt)
        if with_metrics and n:
            for key in self.metric_results:
                self.metric_results[key] /= n
                self._update_best_metric_result(name, key, self.metric_results[key], current_iter)
            if tb_logger is not None:
                for key, v in self.metric_results.items():
                    tb_logger.add_scalar(f"metrics/{name}/{key}", v, current_iter)
        self.first_val_completed = True

    def _dump_validation_image(self, data, dataloader, dataset_name: str, current_iter: int, multi_val_datasets: bool) -> None:
        """sr_model.py:1053-1113: where the SR image of one validation item is written; tensor2img's clamp / x255 / round (img_util.py:251-277)."""
        import shutil

        from PIL import Image

        lq_path = data["lq_path"][0] if isinstance(data.get("lq_path"), (list, tuple)) else data.get("lq_path")
        if lq_path is None:
            raise ValueError("save_img needs `lq_path` in the validation batch (sr_model.py:1007)")
        img_name = osp.splitext(osp.basename(lq_path))[0]
        vis = self.opt.path.visualization
        save_dir = None
        if self.opt.is_train:
            if multi_val_datasets:
                save_dir = osp.join(vis, f"{dataset_name} - {img_name}")
            else:
                dsopt = getattr(dataloader.dataset, "opt", None)
                roots = (dsopt.get("dataroot_lq") if isinstance(dsopt, dict) else getattr(dsopt, "dataroot_lq", None))
                assert roots is not None, "dataroot_lq is required for val set"
                norm = osp.normpath(lq_path)
                root = next((r for r in roots if norm.startswith(osp.normpath(r) + osp.sep)), None)
                if root is None:
                    raise ValueError(f"The lq_path {lq_path} does not match any of the provided dataroot_lq paths.")
                save_dir = osp.join(vis, osp.relpath(osp.splitext(lq_path)[0], root))
            path = osp.join(save_dir, f"{img_name}_{current_iter:06d}.png")
        elif self.opt.val.suffix:
            path = osp.join(vis, dataset_name, f"{img_name}_{self.opt.val.suffix}.png")
        else:
            path = osp.join(vis, dataset_name, f"{img_name}.png")
        os.makedirs(osp.dirname(path), exist_ok=True)
        u8 = (self.output[0].float().clamp(0, 1) * 255.0).round().to(torch.uint8).permute(1, 2, 0).cpu().numpy()
        Image.fromarray(u8[..., 0] if u8.shape[2] == 1 else u8).save(path)
        if self.opt.is_train and not self.first_val_completed and osp.exists(lq_path):
            lr_target = osp.join(save_dir, f"{img_name}_lr.png")
            if not osp.exists(lr_target):
                shutil.copy(lq_path, lr_target)

    def _initialize_best_metric_results(self, dataset_name: str) -> None:
        """base_model.py:160-173"""
        if dataset_name in self.best_metric_results:
            return
        rec = {}
        for metric, content in self.opt.val.metrics.items():
            better = content.get("better", "higher")
            rec[metric] = {"better": better, "val": float("-inf") if better == "higher" else float("inf"), "iter": -1}
        self.best_metric_results[dataset_name] = rec

    def _update_best_metric_result(self, dataset_name: str, metric: str, val: float, current_iter: int) -> None:
        """base_model.py:175-184"""
        r = self.best_metric_results[dataset_name][metric]
        if (r["better"] == "higher" and val >= r["val"]) or (r["better"] != "higher" and val <= r["val"]):
            r["val"], r["iter"] = val, current_iter

    def load_network(self, net: nn.Module, load_path: str, strict: bool = True, param_key: str | None = None) -> None:
        """safetensors / .pth state dicts (base_model.py:570-648); `param_key` falls back params_ema -> params -> flat."""
        if load_path.endswith(".safetensors"):
            from safetensors.torch import load_file

            sd = load_file(load_path)
        else:
            sd = torch.load(load_path, map_location="cpu", weights_only=True)
            for k in ([param_key] if param_key else []) + ["params_ema", "params"]:
                if isinstance(sd, dict) and k in sd and isinstance(sd[k], dict):
                    sd = sd[k]
                    break
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        net.load_state_dict(sd, strict=strict)

    @master_only
    def save_network(self, net: nn.Module, path: str, fmt: str = "safetensors") -> None:
        """base_model.py:395-482 (`@master_only` there too: the loop calls save() on every rank, only rank 0 writes)."""
        sd = {k: v.detach().cpu().contiguous() for k, v in net.state_dict().items()}
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        if fmt == "safetensors":
            from safetensors.torch import save_file

            save_file(sd, path)
        else:
            torch.save(sd, path)

    def save(self, epoch: int, current_iter: int) -> None:
        """EMA net -> models/, raw net_g/net_d -> models/resume_models/ (sr_model.py:1174-1205)."""
        self._flush_pending()       # a NaN / Inf loss queued since the last read-back must raise before anything is written
        p = self.opt.path
        fmt = self.opt.logger.save_checkpoint_format if self.opt.logger is not None else "safetensors"
        if p.models is None:
            raise ValueError("opt.path.models is not set")
        ext = "safetensors" if fmt == "safetensors" else "pth"
        if self.net_g_ema is not None:
            self.save_network(self.net_g_ema.ema_model, os.path.join(p.models, f"net_g_ema_{current_iter}.{ext}"), fmt)
            self.save_network(self.net_g, os.path.join(p.resume_models or p.models, f"net_g_{current_iter}.{ext}"), fmt)
        else:
            self.save_network(self.net_g, os.path.join(p.models, f"net_g_{current_iter}.{ext}"), fmt)
        if self.net_d is not None:
            self.save_network(self.net_d, os.path.join(p.resume_models or p.models, f"net_d_{current_iter}.{ext}"), fmt)

    @master_only
    def save_training_state(self, epoch: int, current_iter: int) -> None:
        """`{iter}.state` in the reference's layout (base_model.py:713-791): epoch, iter, optimizer and scheduler state dicts (the
        optimizer's are torch.optim.AdamW-compatible per-parameter entries), EMA step, scaler_g / scaler_d under fp16."""
        self._flush_pending()
        p = self.opt.path
        if p.training_states is None:
            raise ValueError("opt.path.training_states is not set")
        if current_iter == -1:
            return
        state = {"epoch": epoch, "iter": current_iter, "optimizers": [o.state_dict() for o in self.optimizers],
                 "schedulers": [s.state_dict() for s in self.schedulers]}
        if self.net_g_ema is not None:
            state["ema_step"] = self.net_g_ema.step.detach().cpu().clone()
        if self.scaler_g.enabled:
            state["scaler_g"], state["scaler_d"] = self.scaler_g.state_dict(), self.scaler_d.state_dict()
        os.makedirs(p.training_states, exist_ok=True)
        torch.save(state, os.path.join(p.training_states, f"{current_iter}.state"))

    def resume_training(self, resume_state: dict) -> None:
        """base_model.py:794-845."""
        ro, rs = resume_state["optimizers"], resume_state["schedulers"]
        assert len(ro) == len(self.optimizers), "Wrong lengths of optimizers"
        assert len(rs) == len(self.schedulers), "Wrong lengths of schedulers"
        for o, sd in zip(self.optimizers, ro):
            o.load_state_dict(sd)
        for s, sd in zip(self.schedulers, rs):
            s.load_state_dict(sd)
        if "scaler_g" in resume_state:
            self.scaler_g.load_state_dict(resume_state["scaler_g"])
        if "scaler_d" in resume_state:
            self.scaler_d.load_state_dict(resume_state["scaler_d"])
        if "ema_step" in resume_state and self.net_g_ema is not None:
            self.net_g_ema.step.copy_(resume_state["ema_step"])
            self.net_g_ema.initted.fill_(True)
