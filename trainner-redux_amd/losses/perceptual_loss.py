"""VGG19 perceptual loss heads `perceptualloss` (fp32, as the reference forces) and `perceptualfp16loss`
(follows the AMP compute dtype) on the HIP conv / pool / loss kernels.

Same constructor signature, layer naming, default tapped layers and weights as
/root/reference/traiNNer/losses/perceptual_fp16_loss.py (:12-71 tables, :74-151 ctor, :216-243 forward, :246-329 VGG) and
/root/reference/traiNNer/losses/perceptual_loss.py:8-12 (fp32 via custom_fwd(cast_inputs=float32)).
`forward(x, gt)` returns `dict[layer -> weighted scalar]` like the reference, so SRModel logs `l_g_perceptual_<layer>` and
applies the `* accum_iters` rule for dict losses (sr_model.py:654-659).

Scope: the conv-layer criteria `l1` / `charbonnier` (the BASELINE configs); `pd` / `fd` / `*+l1` raise NotImplementedError.
Weights: the reference downloads torchvision's VGG19 (`:250`), impossible offline.  A torchvision-format state dict
(`features.N.weight`) is taken from `$TRAINNER_VGG19_WEIGHTS` or torch hub's cache if present, otherwise the head
is initialised with seeded He-normal weights and says so (benchmarks and parity tests use those).

Execution: gt pass (features only), x pass (activations kept), then ONE reverse sweep in which every tapped layer's
loss kernel produces its value and adds its gradient into the running feature gradient; input gradients reuse the MFMA
conv kernel on transposed slabs with the ReLU derivative as an epilogue mask; 2x2 max-pool (+ReLU) has its own kernels;
conv1_1's replicate-padded input gradient and the `1/std` of the normalisation are folded into one stem kernel that
accumulates straight into d(loss)/d(output) of the generator.
"""
from __future__ import annotations

import math
import os
import warnings
from collections import OrderedDict
from typing import Literal

import torch
from torch import Tensor, nn

from .. import _lib as L
from .. import ops
from ..utils.registry import LOSS_REGISTRY

VGG19_LAYERS: list[str] = []
for _b, _n in enumerate((2, 2, 4, 4, 4), start=1):
    for _i in range(1, _n + 1):
        VGG19_LAYERS += [f"conv{_b}_{_i}", f"relu{_b}_{_i}"]
    VGG19_LAYERS.append(f"pool{_b}")
VGG19_CHANNELS = [64, 128, 256, 512, 512]
VGG19_CONV_LAYER_WEIGHTS = {"conv1_2": 0.1, "conv2_2": 0.1, "conv3_4": 1, "conv4_4": 1, "conv5_4": 1}
VGG19_RELU_LAYER_WEIGHTS = {"relu1_2": 0.1, "relu2_2": 0.1, "relu3_4": 1, "relu4_4": 1, "relu5_4": 1}
VGG19_CONV_CRITERION = {"l1", "charbonnier", "pd+l1", "fd+l1"}
VGG19_RELU_CRITERION = {"pd", "fd", "pd+l1", "fd+l1"}
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class _ConvP(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.weight = nn.Parameter(torch.zeros(cout, cin, 3, 3), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)


class _Stage(nn.Module):
    pass


def _find_vgg_weights() -> str | None:
    cands = [os.environ.get("TRAINNER_VGG19_WEIGHTS"), os.path.join(torch.hub.get_dir(), "checkpoints", "vgg19-dcbb9e9d.pth")]
    for c in cands:
        if c and os.path.isfile(c):
            return c
    return None


class VGG(nn.Module):
    """Parameter container with the reference's naming (`stages.<layer>.<features idx>.weight`) + the HIP executor."""

    def __init__(self, layer_name_list: list[str]) -> None:
        super().__init__()
        self.layer_names = sorted(layer_name_list, key=VGG19_LAYERS.index)
        last = VGG19_LAYERS.index(self.layer_names[-1])
        self.stages = nn.ModuleDict()
        self.convs: list[tuple[str, int, _ConvP]] = []
        stage = _Stage()
        cin = 3
        for idx, name in enumerate(VGG19_LAYERS[: last + 1]):
            if name.startswith("conv"):
                cout = VGG19_CHANNELS[int(name[4]) - 1]
                m = _ConvP(cin, cout)
                stage.add_module(str(idx), m)
                self.convs.append((name, idx, m))
                cin = cout
            if name in self.layer_names:
                self.stages[name] = stage
                stage = _Stage()
        self.register_buffer("mean", torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor(IMAGENET_STD).view(1, 3, 1, 1))
        self.synthetic = True
        path = _find_vgg_weights()
        if path is not None:
            self.load_torchvision_state_dict(torch.load(path, map_location="cpu", weights_only=True))
        else:
            self.init_synthetic(0)
        self._build_program()
        self._packs = None
        self._ws: dict = {}
        # exact-fp32 convs run as fused Winograd F(2x2, 3x3) MFMA kernels (same arithmetic type, 2.25x fewer multiplies, ~1e-6 from the
        # direct sum); TRAINNER_VGG_WINOGRAD=0 keeps the direct kernel
        self.winograd = os.environ.get("TRAINNER_VGG_WINOGRAD", "1") != "0"
        self.winograd43 = os.environ.get("TRAINNER_VGG_WINOGRAD", "1") not in ("0", "2")       # "2": F(2x2, 3x3) only

    # -- weights ------------------------------------------------------------------------------------------
    def init_synthetic(self, seed: int = 0) -> None:
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for _, _, m in self.convs:
                m.weight.copy_(torch.randn(m.cout, m.cin, 3, 3, generator=g) * math.sqrt(2.0 / (m.cin * 9)))
                m.bias.copy_(torch.randn(m.cout, generator=g) * 0.05)
        self.synthetic = True
        self._packs = None

    def load_torchvision_state_dict(self, sd: dict) -> None:
        with torch.no_grad():
            for _, idx, m in self.convs:
                m.weight.copy_(sd[f"features.{idx}.weight"])
                m.bias.copy_(sd[f"features.{idx}.bias"])
        self.synthetic = False
        self._packs = None

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._packs = None
        self._ws = {}
        return out

    # -- layer program ------------------------------------------------------------------------------------
    def _build_program(self) -> None:
        """[(kind, ...)]: conv(name, module, fused_relu, tapped) / pool(relu_first)."""
        last = VGG19_LAYERS.index(self.layer_names[-1])
        names = VGG19_LAYERS[: last + 1]
        prog = []
        i = 0
        conv_iter = iter(self.convs)
        while i < len(names):
            nm = names[i]
            if nm.startswith("conv"):
                _, _, m = next(conv_iter)
                tapped = nm in self.layer_names
                nxt = names[i + 1] if i + 1 < len(names) else None
                if tapped:
                    # tapped pre-ReLU: the ReLU that follows must be fusable into a pool (or the net ends here)
                    if nxt is not None:
                        after = names[i + 2] if i + 2 < len(names) else None
                        if after is None:
                            i += 1  # trailing relu is never consumed
                        elif not after.startswith("pool"):
                            raise NotImplementedError(f"tapping {nm} (followed by {after}) is not on the HIP path")
                    prog.append(("conv", nm, m, False, True))
                    i += 1
                else:
                    if nxt is None or not nxt.startswith("relu"):
                        raise NotImplementedError("VGG19 conv without ReLU")
                    relu_tapped = nxt in self.layer_names
                    if relu_tapped:
                        raise NotImplementedError("relu-layer taps (pd / fd criteria) are not on the HIP path")
                    prog.append(("conv", nm, m, True, False))
                    i += 2
            elif nm.startswith("relu"):
                # only reached after a tapped conv: fused into the next pool
                nxt = names[i + 1] if i + 1 < len(names) else None
                if nxt is None:
                    i += 1
                    continue
                prog.append(("pool", True))
                i += 2
            else:
                prog.append(("pool", False))
                i += 1
        self.program = prog

    # -- slabs / workspaces -----------------------------------------------------------------------------
    def _pack(self, cd: int, dev) -> dict:
        if self._packs is not None and self._packs[0] == (cd, str(dev)):
            return self._packs[1]
        plan = ops.PackPlan(dev)
        pk = {}
        for k, (name, _, m) in enumerate(self.convs):
            cin_f = 8 if k == 0 else m.cin
            f = torch.zeros(ops.packed_bytes(cin_f, m.cout, cd), dtype=torch.uint8, device=dev)
            plan.add(m.weight.data, f, cin_f, m.cout, cd)
            # transposed / flipped slab of the input-gradient conv (for conv1_1: 64 -> 3 channels on the padded domain)
            b = torch.zeros(ops.packed_bytes(m.cout, m.cin, cd), dtype=torch.uint8, device=dev)
            plan.add(m.weight.data, b, m.cout, m.cin, cd, transpose=True)
            pk[name] = (f, b)
            if cd == L.F32 and self.winograd and k > 0 and m.cin % 8 == 0 and m.cout % 64 == 0 and m.cin % 64 == 0:
                # exact-fp32 mode: the fused Winograd F(2x2, 3x3) kernel (csrc/winograd.hip) for every conv but the replicate-padded stem;
                # forward slab and the slab of the input-gradient conv (VGG weights are frozen: packed once)
                pk[name] = (f, b, ops.pack_winograd(m.weight.data), ops.pack_winograd(m.weight.data, transpose=True))
                if self.winograd43:
                    # F(4x4, 3x3) slabs (csrc/winograd43.hip): 0.5625 of the MFMA cycles, ~2e-6 per layer instead of ~4e-7; used where h and w are multiples of 4
                    pk[name] = pk[name] + (ops.pack_winograd43(m.weight.data), ops.pack_winograd43(m.weight.data, transpose=True))
        plan.finalize().run()
        self._packs = ((cd, str(dev)), pk, plan)
        return pk

    def _workspace(self, n, h, w, cdt, dev):
        key = (n, h, w, cdt)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        self._ws.clear()
        e = lambda *s: torch.empty(*s, dtype=cdt, device=dev)  # noqa: E731
        ws = dict(a0=e(n, h, w, 8), acts=[], feats_gt={}, scratch=[e(n, h, w, 64), e(n, h, w, 64)],
                  scratch_pool=e(n, h // 2, w // 2, 64))      # pooled output of a conv + pool pair on the pass that keeps nothing
        hh, ww = h, w
        levels = {}
        for op in self.program:
            if op[0] == "conv":
                ws["acts"].append(e(n, hh, ww, op[2].cout))
                levels[(hh, ww)] = max(levels.get((hh, ww), 0), op[2].cout)
                if op[4]:
                    ws["feats_gt"][op[1]] = e(n, hh, ww, op[2].cout)
            else:
                hh, ww = hh // 2, ww // 2
                ws["acts"].append(e(n, hh, ww, ws["acts"][-1].shape[-1]))
        # gradient ping-pong per resolution level
        ws["g"] = {k: [e(n, k[0], k[1], c), e(n, k[0], k[1], c)] for k, c in levels.items()}
        ws["gp"] = torch.empty(n, h + 2, w + 2, 4, dtype=torch.float32, device=dev)   # gradient w.r.t. the replicate-padded image
        self._ws[key] = ws
        return ws

    # -- execution -----------------------------------------------------------------------------------------
    def _forward_pass(self, img: Tensor, ws, pk, cd, keep: bool):
        """Runs the program; keep=True stores every activation in ws['acts'], else ping-pongs through scratch and keeps
        only the tapped features in ws['feats_gt']."""
        n, _, h, w = img.shape
        V = ops.view
        # fp32 head: conv1_1 runs as the dedicated stem kernel straight from the NCHW image (normalisation and replicate padding fused); the 16-bit
        # head (perceptualfp16loss) keeps the generic path: layout pass into an 8-channel NHWC buffer + MFMA conv
        stem = ws["a0"].dtype == torch.float32 and img.dtype == torch.float32 and img.shape[1] <= 3 and not os.environ.get("TRAINNER_NO_STEM")
        if not stem:
            ops.nchw_to_nhwc(img, ws["a0"], IMAGENET_MEAN, IMAGENET_STD)
        cur, cur_c = ws["a0"], 8
        hh, ww = h, w
        first = True
        flip = 0
        fuse_pool = not os.environ.get("TRAINNER_NO_POOL_FUSION")
        pooled = None           # output of the pool op that the previous conv has already written (Winograd F(4,3) epilogue)
        ws.pop("_pool_view", None)      # per pass: the marker of "this conv's input is the shared pooled scratch"
        for k, op in enumerate(self.program):
            if op[0] == "conv":
                _, name, m, fused_relu, tapped = op
                if keep:
                    out = ws["acts"][k]
                elif tapped:
                    out = ws["feats_gt"][name]
                else:
                    out = ws["scratch"][flip].view(-1)[: n * hh * ww * m.cout].view(n, hh, ww, m.cout)
                    flip ^= 1
                wino = len(pk[name]) > 2 and not first and hh % 2 == 0 and ww % 2 == 0
                w43 = wino and len(pk[name]) > 4 and ops.winograd43_eligible(cur_c, m.cout, hh, ww, cur.shape[-1], out.shape[-1])
                wino = wino and (w43 or ops.winograd_eligible(cur_c, m.cout, hh, ww))      # an ineligible layer / size falls back: F(4,3) -> F(2,2) -> direct
                if first and stem:
                    ops.conv3x3_stem(img.contiguous(), m.weight.data, m.bias.data, V(out), mean=IMAGENET_MEAN, std=IMAGENET_STD, pad_mode=L.PAD_REPLICATE,
                                     act=L.ACT_RELU if fused_relu else L.ACT_NONE)
                else:
                    nxt = self.program[k + 1] if k + 1 < len(self.program) else None
                    kw = {}
                    # conv -> (ReLU) -> MaxPool: the F(4,3) kernel writes max_pool2d(relu(out)) next to out, the pool kernel is skipped
                    if fuse_pool and w43 and nxt is not None and nxt[0] == "pool" and (nxt[1] or fused_relu) and cur is not ws.get("_pool_view"):
                        if keep:
                            pooled = ws["acts"][k + 1]
                        else:       # its own buffer: both ping-pong buffers are taken (this conv's input and output)
                            pooled = ws["scratch_pool"].view(-1)[: n * (hh // 2) * (ww // 2) * m.cout].view(n, hh // 2, ww // 2, m.cout)
                            ws["_pool_view"] = pooled
                        kw["out2"] = V(pooled)
                    ops.conv3x3(n=n, h=hh, w=ww, cin=cur_c, cout=m.cout, compute=L.F32_WINOGRAD43 if w43 else L.F32_WINOGRAD if wino else cd,
                                inp=V(cur, 0, cur_c), out=V(out), wpacked=pk[name][4 if w43 else 2 if wino else 0], bias=m.bias.data,
                                act=L.ACT_RELU if fused_relu else L.ACT_NONE, pad_mode=L.PAD_REPLICATE if first else L.PAD_ZEROS, **kw)
                first = False
                cur, cur_c = out, m.cout
            else:
                relu_first = op[1]
                if pooled is not None:
                    out, pooled = pooled, None
                else:
                    if keep:
                        out = ws["acts"][k]
                    else:
                        out = ws["scratch"][flip].view(-1)[: n * (hh // 2) * (ww // 2) * cur_c].view(n, hh // 2, ww // 2, cur_c)
                        flip ^= 1
                    ops.maxpool2x2(V(cur), V(out), n, hh, ww, relu_first)
                hh, ww = hh // 2, ww // 2
                cur = out

    def loss_and_grad(self, x: Tensor, gt: Tensor, *, layer_weights: dict, kind: int, eps: float, compute_dtype, value_weight: float,
                      value_out: Tensor, total_weight: float, total_out: Tensor | None, grad_weight: float, grad: Tensor | None,
                      first: bool, layer_grad_scale: dict[str, float] | None = None, compute_code: int | None = None) -> list[str]:
        """value_out[j] = value_weight * lw_j * crit(F_j(x), F_j(gt)); total_out[0] += total_weight * (same);
        grad (+)= grad_weight * d(sum_j lw_j crit_j)/dx.  Returns the tapped layer names in value order."""
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape == gt.shape and x.shape[1] == 3
        n, _, h, w = x.shape
        npool = sum(1 for op in self.program if op[0] == "pool")
        if h % (1 << npool) or w % (1 << npool):
            raise ValueError(f"perceptual head needs H, W divisible by {1 << npool} (got {h}x{w})")
        dev = x.device
        cd = ops.dt_code(compute_dtype) if compute_code is None else compute_code     # L.BF16X3: fp32 buffers, split-bf16 MFMAs
        pk = self._pack(cd, dev)
        ws = self._workspace(n, h, w, compute_dtype, dev)
        V = ops.view
        self._forward_pass(gt.contiguous(), ws, pk, cd, keep=False)
        self._forward_pass(x.contiguous(), ws, pk, cd, keep=True)
        names = [op[1] for op in self.program if op[0] == "conv" and op[4]]
        slot = {nm: j for j, nm in enumerate(names)}
        # ---- reverse sweep -------------------------------------------------------------------------------
        prog = self.program
        dims = []
        hh, ww = h, w
        for op in prog:
            dims.append((hh, ww))
            if op[0] == "pool":
                hh, ww = hh // 2, ww // 2
        g = None            # gradient w.r.t. the output of prog[k] (None: nothing flows from above)
        pool_gout = None    # gradient of a pool's output whose backward is folded into the loss-gradient pass of the tapped conv below it
        fuse_pool_bwd = not os.environ.get("TRAINNER_NO_POOL_FUSION")
        for k in range(len(prog) - 1, -1, -1):
            op = prog[k]
            hh, ww = dims[k]
            if op[0] == "pool":
                if grad is None:
                    continue
                src = ws["acts"][k - 1]                      # pool input (pre-ReLU tapped feature or post-ReLU activation)
                pool_g = ws["g"][(hh, ww)]
                gin = pool_g[0].view(-1)[: src.numel()].view(src.shape)
                below = prog[k - 1]
                if (fuse_pool_bwd and below[0] == "conv" and below[4] and src.dtype == torch.float32 and g.dtype == torch.float32 and g.is_contiguous()
                        and src.shape[-1] % 4 == 0):
                    pool_gout = g                            # the tapped conv below takes the pool's backward into its loss-gradient pass
                    g = gin
                    continue
                ops.maxpool2x2_bwd(V(src), V(g), V(gin), n, hh, ww, True)
                g = gin
                continue
            _, name, m, fused_relu, tapped = op
            act = ws["acts"][k]
            if tapped:
                if g is None:
                    g = ws["g"][(hh, ww)][0].view(-1)[: act.numel()].view(act.shape)
                    acc_g = False
                else:
                    acc_g = True
                j = slot[name]
                lw = float(layer_weights[name])
                gs = 1.0 if layer_grad_scale is None else float(layer_grad_scale[name])
                ops.pixel_loss(kind, act, ws["feats_gt"][name], value_out[j:], value_weight=value_weight * lw, total_out=total_out,
                               total_weight=total_weight * lw, grad=g if grad is not None else None, grad_weight=grad_weight * lw * gs,
                               accumulate_grad=acc_g, eps=eps, pool_gout=pool_gout)
                pool_gout = None
            if grad is None:
                continue
            if k == 0:
                # conv1_1 ran on a replicate-padded image: its input gradient is the 'full' correlation on the padded domain
                # (MFMA conv, 64 -> 3 channels), folded back onto the image border, scaled by 1/std, added into NCHW grad
                # fp32 head, 64 -> 3: the tiled VALU kernel reads g once and writes the image gradient directly
                if g.dtype == torch.float32 and m.cout == 64 and m.cin == 3 and not os.environ.get("TRAINNER_NO_STEM"):
                    ops.conv3x3_dgrad_image(V(g), m.weight.data, n, hh, ww, grad, pad_mode=L.PAD_REPLICATE, ch_scale=[1.0 / s for s in IMAGENET_STD],
                                            accumulate=not first)
                    break
                ops.conv3x3(n=n, h=hh + 2, w=ww + 2, cin=m.cout, cout=m.cin, compute=cd, inp=V(g), out=V(ws["gp"]), wpacked=pk[name][1],
                            in_mode=L.IN_FULL)
                ops.fold_replicate_nchw(V(ws["gp"]), grad, ch_scale=[1.0 / s for s in IMAGENET_STD], accumulate=not first)
                break
            # gradient w.r.t. this conv's input = transposed conv of g; the input came through a fused ReLU unless it is a pool output
            prev = prog[k - 1]
            inp_act = ws["acts"][k - 1]
            pool_pair = ws["g"][(hh, ww)]
            dst_buf = pool_pair[1] if g.data_ptr() == pool_pair[0].data_ptr() else pool_pair[0]
            gin = dst_buf.view(-1)[: inp_act.numel()].view(inp_act.shape)
            mask = V(inp_act) if (prev[0] == "conv" and prev[3]) else ops.NULL_VIEW
            wino = len(pk[name]) > 2 and hh % 2 == 0 and ww % 2 == 0
            w43 = wino and len(pk[name]) > 4 and ops.winograd43_eligible(m.cout, m.cin, hh, ww, g.shape[-1], gin.shape[-1])
            wino = wino and (w43 or ops.winograd_eligible(m.cout, m.cin, hh, ww))
            ops.conv3x3(n=n, h=hh, w=ww, cin=m.cout, cout=m.cin, compute=L.F32_WINOGRAD43 if w43 else L.F32_WINOGRAD if wino else cd, inp=V(g), out=V(gin),
                        wpacked=pk[name][5 if w43 else 3 if wino else 1], mask=mask, mask_slope=0.0)
            g = gin
        return names


@LOSS_REGISTRY.register()
class PerceptualFP16Loss(nn.Module):
    returns_dict = True
    force_fp32 = False

    def __init__(self, loss_weight: float, layer_weights: dict[str, float] | None = None, w_lambda: float = 0.01,
                 alpha: list[float] | None = None,
                 criterion: Literal["pd+l1", "fd+l1", "pd", "fd", "charbonnier", "l1"] = "charbonnier", num_proj_fd: int = 256,
                 phase_weight_fd: float = 1.0, stride_fd: int = 1, clip_min: int = 0) -> None:
        super().__init__()
        if criterion not in VGG19_CONV_CRITERION | VGG19_RELU_CRITERION:
            raise NotImplementedError(f"{criterion} criterion has not been supported.")
        if criterion not in ("l1", "charbonnier"):
            raise NotImplementedError(f"criterion {criterion!r} (relu-layer pd / fd terms) is not on the HIP hot path")
        if layer_weights is None:
            layer_weights = dict(VGG19_CONV_LAYER_WEIGHTS)
        self.vgg = VGG(list(layer_weights.keys()))
        if alpha is None:
            alpha = [0.0 if k.startswith("conv") else 1.0 for k in self.vgg.layer_names]
        if any(a != 0 for a in alpha):
            raise NotImplementedError("alpha > 0 (relu-layer criterion) is not on the HIP hot path")
        self.loss_weight = loss_weight
        self.w_lambda, self.layer_weights, self.alpha = w_lambda, layer_weights, alpha
        self.phase_weight_fd, self.stride_fd, self.clip_min = phase_weight_fd, stride_fd, clip_min
        self.criterion = criterion
        self.kind = L.LOSS_L1 if criterion == "l1" else L.LOSS_CHARBONNIER
        self.compute_dtype = torch.float32
        self.fast_matmul = False
        if self.vgg.synthetic:
            warnings.warn("VGG19 weights not found (set TRAINNER_VGG19_WEIGHTS to a torchvision vgg19 state dict); "
                          "the perceptual head uses seeded synthetic weights", stacklevel=2)

    def set_compute_dtype(self, dtype: torch.dtype) -> None:
        """`perceptualloss` ignores AMP (fp32 forced, perceptual_loss.py:10); `perceptualfp16loss` follows it."""
        self.compute_dtype = torch.float32 if self.force_fp32 else dtype

    def set_fast_matmul(self, on: bool) -> None:
        """The reference's `fast_matmul` option ("Trade precision for performance", redux_options.py:479-481: TF32 matmuls on its CUDA
        targets).  gfx950 has no TF32; the equivalent here is split-bf16: the fp32-forced VGG keeps fp32 activations and weights but
        every product runs as three bf16 MFMAs (hi*hi + hi*lo + lo*hi), ~2^-17 relative per product (TF32: 2^-11)."""
        self.fast_matmul = bool(on)

    def _compute_code(self):
        return L.BF16X3 if (self.fast_matmul and self.compute_dtype == torch.float32) else None

    def hip_value_and_grad(self, pred: Tensor, target: Tensor, *, value_weight: float, value_out: Tensor, total_weight: float,
                           total_out: Tensor | None, grad_weight: float, grad: Tensor | None, first: bool) -> list[str]:
        return self.vgg.loss_and_grad(pred, target, layer_weights=self.layer_weights, kind=self.kind, eps=1e-12,
                                      compute_dtype=self.compute_dtype, value_weight=value_weight, value_out=value_out,
                                      total_weight=total_weight, total_out=total_out, grad_weight=grad_weight, grad=grad, first=first,
                                      compute_code=self._compute_code())

    def forward(self, x: Tensor, gt: Tensor) -> dict[str, Tensor]:
        vals = _PerceptualFn.apply(x, gt, self)
        return OrderedDict(zip(self.vgg.layer_names, vals.unbind(0)))


class _PerceptualFn(torch.autograd.Function):
    """Autograd bridge for foreign trainers: forward = features + per-layer values; backward re-runs the fused sweep with
    the upstream per-layer gradients as layer scales (SRModel never takes this path)."""

    @staticmethod
    def forward(ctx, x, gt, mod: "PerceptualFP16Loss"):
        x32, gt32 = x.detach().float().contiguous(), gt.detach().float().contiguous()
        vals = torch.zeros(len(mod.vgg.layer_names), dtype=torch.float32, device=x.device)
        mod.vgg.loss_and_grad(x32, gt32, layer_weights=mod.layer_weights, kind=mod.kind, eps=1e-12, compute_dtype=mod.compute_dtype,
                              value_weight=1.0, value_out=vals, total_weight=0.0, total_out=None, grad_weight=0.0, grad=None, first=True)
        ctx.save_for_backward(x32, gt32)
        ctx.mod = mod
        return vals

    @staticmethod
    def backward(ctx, gvals):
        x32, gt32 = ctx.saved_tensors
        mod = ctx.mod
        scales = dict(zip(mod.vgg.layer_names, gvals.tolist()))
        grad = torch.empty_like(x32)
        dummy = torch.zeros(len(scales), dtype=torch.float32, device=x32.device)
        mod.vgg.loss_and_grad(x32, gt32, layer_weights=mod.layer_weights, kind=mod.kind, eps=1e-12, compute_dtype=mod.compute_dtype,
                              value_weight=1.0, value_out=dummy, total_weight=0.0, total_out=None, grad_weight=1.0, grad=grad, first=True,
                              layer_grad_scale=scales)
        return grad, None, None


@LOSS_REGISTRY.register()
class PerceptualLoss(PerceptualFP16Loss):
    force_fp32 = True
