"""`unetdiscriminatorsn` and `vggstylediscriminator` on the HIP kernels (gfx950).

Same registry name, constructor arguments and state-dict keys (`convN.weight_orig / weight_u / weight_v`, `conv0/conv9.weight|bias`)
as /root/reference/traiNNer/archs/discriminator_arch.py:98-160.  Pinned by the reference's own tensors
(tests/golden/unet_discriminator_sn.pt).

How the reference ops map here:
  * 3x3 convs (+bias, LeakyReLU)            -> tr_conv3x3 / tr_wgrad3x3 (MFMA)
  * 4x4 stride-2 pad-1 convs (:119-121)     -> tr_space_to_depth2 + the SAME 3x3 MFMA conv on weights expanded by tr_expand_w4x4s2
                                               (16 of the 36 expanded taps are non-zero: 2.25x MFMA work, no new conv kernel)
  * bilinear x2 (:139,144,149)              -> tr_bilinear_up2x / tr_bilinear_up2x_bwd
  * skip adds `x4 + x2` ...                 -> tr_axpby (the pre-skip activation is kept: its sign is the LeakyReLU derivative)
  * spectral_norm hook (:115-128)           -> tr_spectral_norm per SN conv per training-mode forward (power iteration updates
                                               weight_u / weight_v in place), tr_spectral_norm_bwd maps dL/d(W/sigma) to dL/dW_orig
Backward is hand-scheduled like the generator's; one forward may be in flight at a time (the trainer runs
D(fake) -> backward to the image, then D(real) and D(fake.detach()) -> backward to the weights, sequentially).
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import _lib as L
from .. import ops
from ..utils.registry import ARCH_REGISTRY
from ._arena import FlatParamModule

SLOPE = 0.2


class _PlainConv(nn.Module):
    def __init__(self, cin, cout, k=3):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        ref = nn.Conv2d(cin, cout, k, 1, 1)
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone())


class _SNConv(nn.Module):
    """Holder named like a spectral-normed nn.Conv2d (bias=False): weight_orig (Parameter), weight_u / weight_v (buffers)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        ref = nn.Conv2d(cin, cout, k, 1, 1, bias=False)
        w = ref.weight.detach().clone()
        self.weight_orig = nn.Parameter(w)
        self.register_buffer("weight_u", torch.nn.functional.normalize(torch.randn(cout), dim=0, eps=1e-12))
        self.register_buffer("weight_v", torch.nn.functional.normalize(torch.randn(cin * k * k), dim=0, eps=1e-12))


@ARCH_REGISTRY.register()
class UNetDiscriminatorSN(FlatParamModule):
    def __init__(self, num_in_ch: int = 3, num_feat: int = 64, skip_connection: bool = True) -> None:
        super().__init__()
        if num_feat % 8:
            raise ValueError("the HIP layout needs num_feat % 8 == 0")
        nf = num_feat
        self.num_in_ch, self.nf, self.skip_connection = num_in_ch, nf, skip_connection
        self.conv0 = _PlainConv(num_in_ch, nf)
        self.conv1 = _SNConv(nf, nf * 2, 4)
        self.conv2 = _SNConv(nf * 2, nf * 4, 4)
        self.conv3 = _SNConv(nf * 4, nf * 8, 4)
        self.conv4 = _SNConv(nf * 8, nf * 4, 3)
        self.conv5 = _SNConv(nf * 4, nf * 2, 3)
        self.conv6 = _SNConv(nf * 2, nf, 3)
        self.conv7 = _SNConv(nf, nf, 3)
        self.conv8 = _SNConv(nf, nf, 3)
        self.conv9 = _PlainConv(nf, 1)
        self.compute_dtype = torch.bfloat16
        self._init_arena_state()
        self._sn: dict = {}

    # ------------------------------------------------------------------ per-forward weight preparation -----------------
    def _cd(self) -> int:
        return ops.dt_code(self.compute_dtype)

    def _sn_layers(self):
        return [(f"conv{i}", getattr(self, f"conv{i}")) for i in range(1, 9)]

    def _prepare_weights(self, train: bool) -> None:
        """spectral norm (+ power iteration in training mode), 4x4 -> 3x3 expansion, MFMA slabs.  Runs every forward."""
        self._ensure_arena()
        dev = self._arena.device
        cd = self._cd()
        st = self._sn
        key = (cd, str(dev), self._arena.data_ptr())
        if st.get("key") != key:
            st.clear()
            st["key"] = key
            plan = ops.PackPlan(dev)

            def slabs(src, cin_f, cout, cin_b=None):
                f = torch.zeros(ops.packed_bytes(cin_f, cout, cd), dtype=torch.uint8, device=dev)
                plan.add(src, f, cin_f, cout, cd)
                cb = cin_b or cout
                b = torch.zeros(ops.packed_bytes(cb, src.shape[1], cd), dtype=torch.uint8, device=dev)
                plan.add(src, b, cb, src.shape[1], cd, transpose=True)
                return f, b

            for name, m in self._sn_layers():
                w_sn = torch.empty_like(m.weight_orig.data)
                ent = dict(w_sn=w_sn, sigma=torch.zeros(1, device=dev), g_sn=torch.empty_like(w_sn))
                if m.k == 4:
                    ent["w3"] = torch.empty(m.cout, 4 * m.cin, 3, 3, device=dev)
                    ent["g3"] = torch.empty(m.cout, 4 * m.cin, 3, 3, device=dev)
                    ent["f"], ent["b"] = slabs(ent["w3"], 4 * m.cin, m.cout)
                else:
                    ent["f"], ent["b"] = slabs(w_sn, m.cin, m.cout)
                st[name] = ent
            c0, c9 = self.conv0, self.conv9
            f0 = torch.zeros(ops.packed_bytes(8, c0.cout, cd), dtype=torch.uint8, device=dev)
            plan.add(c0.weight.data, f0, 8, c0.cout, cd)
            b0 = torch.zeros(ops.packed_bytes(c0.cout, c0.cin, cd), dtype=torch.uint8, device=dev)
            plan.add(c0.weight.data, b0, c0.cout, c0.cin, cd, transpose=True)
            st["conv0"] = dict(f=f0, b=b0)
            f9, b9 = slabs(c9.weight.data, c9.cin, 1, cin_b=8)
            st["conv9"] = dict(f=f9, b=b9)
            st["plan"] = plan.finalize()
        for name, m in self._sn_layers():
            ent = st[name]
            ops.spectral_norm(m.weight_orig.data, m.weight_u, m.weight_v, ent["w_sn"], ent["sigma"], power_iteration=train)
            if m.k == 4:
                ops.expand_w4x4s2(ent["w_sn"], ent["w3"])
        st["plan"].run()

    def _workspace(self, n, h, w):
        key = (n, h, w, self._cd())
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        self._ws.clear()
        dev, cdt, nf = self._arena.device, self.compute_dtype, self.nf
        e = lambda *s, dt=cdt: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
        ws = dict(
            x8=e(n, h, w, 8), x0=e(n, h, w, nf), s0=e(n, h // 2, w // 2, 4 * nf), x1=e(n, h // 2, w // 2, 2 * nf),
            s1=e(n, h // 4, w // 4, 8 * nf), x2=e(n, h // 4, w // 4, 4 * nf), s2=e(n, h // 8, w // 8, 16 * nf), x3=e(n, h // 8, w // 8, 8 * nf),
            u3=e(n, h // 4, w // 4, 8 * nf), a4=e(n, h // 4, w // 4, 4 * nf), x4=e(n, h // 4, w // 4, 4 * nf),
            u4=e(n, h // 2, w // 2, 4 * nf), a5=e(n, h // 2, w // 2, 2 * nf), x5=e(n, h // 2, w // 2, 2 * nf),
            u5=e(n, h, w, 2 * nf), a6=e(n, h, w, nf), x6=e(n, h, w, nf), x7=e(n, h, w, nf), x8b=e(n, h, w, nf),
            o4=e(n, h, w, 4, dt=torch.float32),
        )
        self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ forward -----------------------------------------
    def forward_hip(self, x: torch.Tensor, train: bool | None = None) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("UNetDiscriminatorSN (HIP) runs on the GPU only; there is no CPU path in this package")
        L.lib()
        if train is None:
            train = self.training
        x = x.contiguous().float()
        n, c, h, w = x.shape
        if c != self.num_in_ch or h % 8 or w % 8:
            raise ValueError(f"expected {self.num_in_ch} channels and H, W divisible by 8, got {tuple(x.shape)}")
        self._prepare_weights(train)
        ws = self._workspace(n, h, w)
        st, cd, nf = self._sn, self._cd(), self.nf
        V = ops.view
        K = dict(n=n, compute=cd)
        lre = dict(act=L.ACT_LRELU, slope=SLOPE)
        ops.nchw_to_nhwc(x, ws["x8"])
        ops.conv3x3(h=h, w=w, cin=8, cout=nf, inp=V(ws["x8"]), out=V(ws["x0"]), wpacked=st["conv0"]["f"], bias=self.conv0.bias.data, **lre, **K)
        # down path: 4x4 s2 conv == space-to-depth + 3x3 conv
        for src, s, dst, name, hh, ww, cin, cout in (("x0", "s0", "x1", "conv1", h, w, nf, 2 * nf), ("x1", "s1", "x2", "conv2", h // 2, w // 2, 2 * nf, 4 * nf),
                                                    ("x2", "s2", "x3", "conv3", h // 4, w // 4, 4 * nf, 8 * nf)):
            ops.space_to_depth2(V(ws[src]), V(ws[s]), n, hh, ww)
            ops.conv3x3(h=hh // 2, w=ww // 2, cin=4 * cin, cout=cout, inp=V(ws[s]), out=V(ws[dst]), wpacked=st[name]["f"], **lre, **K)
        # up path: bilinear x2 -> conv -> LeakyReLU -> (+ skip)
        for src, u, a, xo, skip, name, hh, ww, cin, cout in (("x3", "u3", "a4", "x4", "x2", "conv4", h // 8, w // 8, 8 * nf, 4 * nf),
                                                            ("x4", "u4", "a5", "x5", "x1", "conv5", h // 4, w // 4, 4 * nf, 2 * nf),
                                                            ("x5", "u5", "a6", "x6", "x0", "conv6", h // 2, w // 2, 2 * nf, nf)):
            ops.bilinear_up2x(V(ws[src]), V(ws[u]), n, hh, ww)
            ops.conv3x3(h=2 * hh, w=2 * ww, cin=cin, cout=cout, inp=V(ws[u]), out=V(ws[a]), wpacked=st[name]["f"], **lre, **K)
            if self.skip_connection:
                ops.axpby(V(ws[a]), 1.0, V(ws[skip]), 1.0, V(ws[xo]), n * 4 * hh * ww)
        x6 = ws["x6"] if self.skip_connection else ws["a6"]
        ops.conv3x3(h=h, w=w, cin=nf, cout=nf, inp=V(x6), out=V(ws["x7"]), wpacked=st["conv7"]["f"], **lre, **K)
        ops.conv3x3(h=h, w=w, cin=nf, cout=nf, inp=V(ws["x7"]), out=V(ws["x8b"]), wpacked=st["conv8"]["f"], **lre, **K)
        ops.conv3x3(h=h, w=w, cin=nf, cout=1, inp=V(ws["x8b"]), out=V(ws["o4"]), wpacked=st["conv9"]["f"], bias=self.conv9.bias.data, **K)
        out = torch.empty(n, 1, h, w, dtype=torch.float32, device=x.device)
        ops.nhwc_to_nchw(V(ws["o4"]), out)
        self._saved = (ws, n, h, w, train)
        return out

    # ------------------------------------------------------------------ backward ----------------------------------------
    def _bwd_ws(self, ws, n, h, w):
        if "g" in ws:
            return
        dev, cdt, nf = self._arena.device, self.compute_dtype, self.nf
        e = lambda *s, dt=cdt: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
        g = {}
        g["g9"] = e(n, h, w, 8)
        for lvl, (hh, ww, c) in {0: (h, w, 2 * nf), 1: (h // 2, w // 2, 4 * nf), 2: (h // 4, w // 4, 8 * nf), 3: (h // 8, w // 8, 16 * nf)}.items():
            g[f"A{lvl}"], g[f"B{lvl}"], g[f"C{lvl}"] = e(n, hh, ww, c), e(n, hh, ww, c), e(n, hh, ww, c)
        g["img"] = e(n, h, w, 4, dt=torch.float32)
        need = 0
        for (hh, ww, cin, cout) in [(h, w, 3, nf), (h // 2, w // 2, 4 * nf, 2 * nf), (h // 4, w // 4, 8 * nf, 4 * nf), (h // 8, w // 8, 16 * nf, 8 * nf),
                                    (h // 4, w // 4, 8 * nf, 4 * nf), (h // 2, w // 2, 4 * nf, 2 * nf), (h, w, 2 * nf, nf), (h, w, nf, nf), (h, w, nf, 1)]:
            need = max(need, ops.wgrad_workspace_bytes(n, hh, ww, cin, cout))
        g["wg"] = torch.empty(need // 4 + 16, dtype=torch.float32, device=dev)
        ws["g"] = g

    def backward_hip(self, grad_out: torch.Tensor, *, param_grads: bool = True, input_grad: torch.Tensor | None = None,
                     input_grad_accumulate: bool = True) -> None:
        """grad_out: dL/dlogits (N x 1 x H x W).  param_grads: accumulate into the gradient arena.  input_grad: NCHW fp32 tensor that
        receives (+)= dL/dx (the generator-side use)."""
        if self._saved is None:
            raise RuntimeError("backward_hip needs a preceding forward_hip")
        ws, n, h, w, _ = self._saved
        self._saved = None
        self._ensure_arena()
        self._bwd_ws(ws, n, h, w)
        g, st, cd, nf = ws["g"], self._sn, self._cd(), self.nf
        V = ops.view
        K = dict(n=n, compute=cd)
        skip = self.skip_connection

        def view_as(buf, ref):
            return buf.view(-1)[: ref.numel()].view(ref.shape)

        def wgrad_plain(conv, inp, gy, hh, ww, cin=None):
            if param_grads:
                ops.wgrad3x3(n=n, h=hh, w=ww, cin=cin or conv.cin, cout=conv.cout, compute=cd, inp=inp, gy=gy, dw=conv.weight.grad,
                             db=conv.bias.grad, workspace=g["wg"], accumulate=True)

        def wgrad_sn(name, m, inp, gy, hh, ww):
            if not param_grads:
                return
            ent = st[name]
            if m.k == 4:
                ops.wgrad3x3(n=n, h=hh, w=ww, cin=4 * m.cin, cout=m.cout, compute=cd, inp=inp, gy=gy, dw=ent["g3"], db=None, workspace=g["wg"])
                ops.collapse_g3x3s2d(ent["g3"], ent["g_sn"])
            else:
                ops.wgrad3x3(n=n, h=hh, w=ww, cin=m.cin, cout=m.cout, compute=cd, inp=inp, gy=gy, dw=ent["g_sn"], db=None, workspace=g["wg"])
            ops.spectral_norm_bwd(ent["g_sn"], ent["w_sn"], m.weight_u, m.weight_v, ent["sigma"], m.weight_orig.grad)

        ops.nchw_to_nhwc(grad_out.contiguous().float(), g["g9"])
        x6 = ws["x6"] if skip else ws["a6"]
        # conv9 / conv8 / conv7
        wgrad_plain(self.conv9, V(ws["x8b"]), V(g["g9"]), h, w)
        gc8 = view_as(g["A0"], ws["x8b"])
        ops.conv3x3(h=h, w=w, cin=8, cout=nf, inp=V(g["g9"]), out=V(gc8), wpacked=st["conv9"]["b"], mask=V(ws["x8b"]), mask_slope=SLOPE, **K)
        wgrad_sn("conv8", self.conv8, V(ws["x7"]), V(gc8), h, w)
        gc7 = view_as(g["B0"], ws["x7"])
        ops.conv3x3(h=h, w=w, cin=nf, cout=nf, inp=V(gc8), out=V(gc7), wpacked=st["conv8"]["b"], mask=V(ws["x7"]), mask_slope=SLOPE, **K)
        wgrad_sn("conv7", self.conv7, V(x6), V(gc7), h, w)
        gx6 = view_as(g["A0"], x6)                       # unmasked: x6 = lrelu(c6) + x0 feeds both the skip and conv6's branch
        ops.conv3x3(h=h, w=w, cin=nf, cout=nf, inp=V(gc7), out=V(gx6), wpacked=st["conv7"]["b"], **K)
        # up path, reversed.  g_skip[level] collects the gradient that flows through the skip connection.
        g_skip = {}
        gx = gx6
        for name, a, u, src, lvl, hh, ww in (("conv6", "a6", "u5", "x5", 0, h, w), ("conv5", "a5", "u4", "x4", 1, h // 2, w // 2),
                                            ("conv4", "a4", "u3", "x3", 2, h // 4, w // 4)):
            m = getattr(self, name)
            g_skip[lvl] = gx if skip else None
            gc = view_as(g[f"B{lvl}"], ws[a])
            ops.axpby(V(gx), 1.0, ops.NULL_VIEW, 0.0, V(gc), n * hh * ww, mask=V(ws[a]), mask_slope=SLOPE)   # through LeakyReLU of this conv
            wgrad_sn(name, m, V(ws[u]), V(gc), hh, ww)
            gu = view_as(g[f"C{lvl}"], ws[u])
            ops.conv3x3(h=hh, w=ww, cin=m.cout, cout=m.cin, inp=V(gc), out=V(gu), wpacked=st[name]["b"], **K)
            gx = view_as(g[f"A{lvl + 1}"], ws[src])
            ops.bilinear_up2x_bwd(V(gu), V(gx), n, hh // 2, ww // 2)
        # gx = dL/dx3 (x3 = lrelu(conv3(...))): down path, reversed
        for name, xk, s, lvl, hh, ww in (("conv3", "x3", "s2", 3, h // 8, w // 8), ("conv2", "x2", "s1", 2, h // 4, w // 4),
                                        ("conv1", "x1", "s0", 1, h // 2, w // 2)):
            m = getattr(self, name)
            gc = view_as(g[f"B{lvl}"], ws[xk])
            ops.axpby(V(gx), 1.0, ops.NULL_VIEW, 0.0, V(gc), n * hh * ww, mask=V(ws[xk]), mask_slope=SLOPE)
            wgrad_sn(name, m, V(ws[s]), V(gc), hh, ww)
            gs = view_as(g[f"C{lvl}"], ws[s])
            ops.conv3x3(h=hh, w=ww, cin=m.cout, cout=4 * m.cin, inp=V(gc), out=V(gs), wpacked=st[name]["b"], **K)
            prev = {"conv3": "x2", "conv2": "x1", "conv1": "x0"}[name]
            gdown = view_as(g[f"B{lvl - 1}"], ws[prev])
            ops.depth_to_space2(V(gs), V(gdown), n, 2 * hh, 2 * ww)
            gx = view_as(g[f"C{lvl - 1}"], ws[prev])
            sk = g_skip.get(lvl - 1)
            if sk is not None:
                ops.axpby(V(gdown), 1.0, V(sk), 1.0, V(gx), n * 4 * hh * ww)
            else:
                ops.axpby(V(gdown), 1.0, ops.NULL_VIEW, 0.0, V(gx), n * 4 * hh * ww)
        # gx = dL/dx0 ; x0 = lrelu(conv0(img))
        gc0 = view_as(g["B0"], ws["x0"])
        ops.axpby(V(gx), 1.0, ops.NULL_VIEW, 0.0, V(gc0), n * h * w, mask=V(ws["x0"]), mask_slope=SLOPE)
        wgrad_plain(self.conv0, V(ws["x8"]), V(gc0), h, w, cin=self.conv0.cin)
        if input_grad is not None:
            ops.conv3x3(h=h, w=w, cin=nf, cout=self.num_in_ch, inp=V(gc0), out=V(g["img"]), wpacked=st["conv0"]["b"], **K)
            ops.nhwc_to_nchw_scaled(V(g["img"]), input_grad, [1.0] * 3, accumulate=input_grad_accumulate)

    # ------------------------------------------------------------------ nn.Module surface -------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if self._trigger is None or self._trigger.device != x.device:
                self._trigger = torch.zeros(1, device=x.device, requires_grad=True)
            return _UNetDFn.apply(x, self._trigger, self)
        return self.forward_hip(x)


class _UNetDFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, trigger, net):
        ctx.net = net
        ctx.need_x = x.requires_grad
        ctx.params = any(p.requires_grad for p in net.parameters())
        return net.forward_hip(x)

    @staticmethod
    def backward(ctx, gy):
        gx = torch.zeros(gy.shape[0], ctx.net.num_in_ch, gy.shape[2], gy.shape[3], device=gy.device) if ctx.need_x else None
        ctx.net.backward_hip(gy, param_grads=ctx.params, input_grad=gx, input_grad_accumulate=False)
        return gx, torch.zeros(1, device=gy.device), None


# =====================================================================================================================================
# VGGStyleDiscriminator (/root/reference/traiNNer/archs/discriminator_arch.py:9-95): ten (input 128) or twelve (input 256) conv -> BatchNorm2d ->
# LeakyReLU(0.2) stages halving the resolution five / six times down to 4 x 4, then Linear(num_feat*8*16, 100) -> LeakyReLU -> Linear(100, 1).
# Same registry name, constructor arguments and state-dict keys (convK_J.weight, bnK_J.{weight,bias,running_mean,running_var,num_batches_tracked},
# linear1 / linear2) as the reference; pinned by the reference module's own tensors (tests/golden/vgg_style_discriminator.pt).
#   3x3 convs                      -> tr_conv3x3 / tr_wgrad3x3
#   4x4 stride-2 convs             -> tr_space_to_depth2 + the 3x3 MFMA conv on weights expanded by tr_expand_w4x4s2 (as in the U-Net above)
#   BatchNorm2d + LeakyReLU        -> tr_batchnorm_fwd / tr_batchnorm_bwd (batch statistics in training mode, running statistics updated like torch;
#                                     BatchNorm makes the samples of a batch interact: under data parallelism every replica normalises with its OWN
#                                     batch statistics, exactly as DDP does without SyncBatchNorm -- SURVEY.md section 8e)
#   feat.reshape(n, -1) (NCHW)     -> tr_nhwc_to_nchw of the 4 x 4 map (the flattening order the Linear weights are stored for)
#   Linear layers                  -> the ksize-1 MFMA path (one "pixel" per sample)
# =====================================================================================================================================
class _NoBiasConv(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        ref = nn.Conv2d(cin, cout, k, 1, 1, bias=False)
        self.weight = nn.Parameter(ref.weight.detach().clone())


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c = c
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._nbt = None                 # host-side count of training forwards since the buffer was last written (no per-forward device op)


class _Lin(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout = cin, cout
        ref = nn.Linear(cin, cout)
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone())


def _p32(c: int) -> int:
    return (c + 31) // 32 * 32


@ARCH_REGISTRY.register()
class VGGStyleDiscriminator(FlatParamModule):
    def __init__(self, num_in_ch: int = 3, num_feat: int = 64, input_size: int = 128) -> None:
        super().__init__()
        assert input_size in (128, 256), f"input size must be 128 or 256, but received {input_size}"
        if num_feat % 8:
            raise ValueError("the HIP layout needs num_feat % 8 == 0")
        nf = num_feat
        self.num_in_ch, self.nf, self.input_size = num_in_ch, nf, input_size
        self.conv0_0 = _PlainConv(num_in_ch, nf)
        widths = [(nf, nf), (nf, 2 * nf), (2 * nf, 4 * nf), (4 * nf, 8 * nf), (8 * nf, 8 * nf)] + ([(8 * nf, 8 * nf)] if input_size == 256 else [])
        self.stages = []                 # (conv name, bn name, kernel) in forward order
        for k, (ci, co) in enumerate(widths):
            if k > 0:
                setattr(self, f"conv{k}_0", _NoBiasConv(ci, co, 3))
                setattr(self, f"bn{k}_0", _BN(co))
                self.stages.append((f"conv{k}_0", f"bn{k}_0", 3))
            setattr(self, f"conv{k}_1", _NoBiasConv(co, co, 4))
            setattr(self, f"bn{k}_1", _BN(co))
            self.stages.append((f"conv{k}_1", f"bn{k}_1", 4))
        self.linear1 = _Lin(8 * nf * 16, 100)
        self.linear2 = _Lin(100, 1)
        self.compute_dtype = torch.bfloat16
        self._init_arena_state()
        self._st: dict = {}

    def _cd(self) -> int:
        return ops.dt_code(self.compute_dtype)

    def state_dict(self, *args, **kwargs):
        for _, bn, _ in self.stages:      # the counters live on the host between saves
            m = getattr(self, bn)
            if m._nbt:
                m.num_batches_tracked += m._nbt
                m._nbt = 0
        return super().state_dict(*args, **kwargs)

    # ------------------------------------------------------------------ slabs ------------------------------------------------------------
    def _prepare(self) -> None:
        self._ensure_arena()
        dev, cd = self._arena.device, self._cd()
        st = self._st
        key = (cd, str(dev), self._arena.data_ptr())
        if st.get("key") != key:
            st.clear()
            st["key"] = key
            plan = ops.PackPlan(dev)

            def z(nbytes):
                return torch.zeros(nbytes, dtype=torch.uint8, device=dev)

            def slabs(src, cin_f, cout, cin_b, ks=3):
                f = z(ops.packed_bytes(cin_f, cout, cd, ks))
                plan.add(src, f, cin_f, cout, cd)
                b = z(ops.packed_bytes(cin_b, src.shape[1], cd, ks))
                plan.add(src, b, cin_b, src.shape[1], cd, transpose=True)
                return f, b

            c0 = self.conv0_0
            f0 = z(ops.packed_bytes(8, c0.cout, cd))
            plan.add(c0.weight.data, f0, 8, c0.cout, cd)
            b0 = z(ops.packed_bytes(c0.cout, c0.cin, cd))
            plan.add(c0.weight.data, b0, c0.cout, c0.cin, cd, transpose=True)
            st["conv0_0"] = dict(f=f0, b=b0)
            for cname, _, k in self.stages:
                m = getattr(self, cname)
                ent = {}
                if k == 4:
                    ent["w3"] = torch.empty(m.cout, 4 * m.cin, 3, 3, device=dev)
                    ent["g3"] = torch.empty(m.cout, 4 * m.cin, 3, 3, device=dev)
                    ent["f"], ent["b"] = slabs(ent["w3"], 4 * m.cin, m.cout, m.cout)
                else:
                    ent["f"], ent["b"] = slabs(m.weight.data, m.cin, m.cout, m.cout)
                st[cname] = ent
            l1, l2 = self.linear1, self.linear2
            f1, b1 = slabs(l1.weight.data, l1.cin, l1.cout, _p32(l1.cout), ks=1)
            f2, b2 = slabs(l2.weight.data, _p32(l2.cin), l2.cout, 32, ks=1)
            st["linear1"], st["linear2"] = dict(f=f1, b=b1), dict(f=f2, b=b2)
            st["bnws"] = ops.batchnorm_workspace(8 * self.nf, dev)
            st["plan"] = plan.finalize()
        for cname, _, k in self.stages:
            if k == 4:
                ops.expand_w4x4s2(getattr(self, cname).weight.data, st[cname]["w3"])
        st["plan"].run()

    def _workspace(self, n, h, w):
        key = (n, h, w, self._cd())
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        self._ws.clear()
        dev, cdt = self._arena.device, self.compute_dtype
        e = lambda *s, dt=cdt: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
        ws = dict(x8=e(n, h, w, 8), a0=e(n, h, w, self.nf), layers=[])
        hh, ww = h, w
        for cname, bn, k in self.stages:
            m = getattr(self, cname)
            ent = {}
            if k == 4:
                ent["s"] = e(n, hh // 2, ww // 2, 4 * m.cin)
                hh, ww = hh // 2, ww // 2
            ent.update(c=e(n, hh, ww, m.cout), y=e(n, hh, ww, m.cout), mean=e(m.cout, dt=torch.float32), rstd=e(m.cout, dt=torch.float32), h=hh, w=ww)
            ws["layers"].append(ent)
        cf = 8 * self.nf
        ws.update(flat32=e(n, cf, 4, 4, dt=torch.float32), flat=e(n, 1, 1, cf * 16), h1=e(n, 1, 1, _p32(100)), o4=e(n, 1, 1, 4, dt=torch.float32))
        ws["h1"].zero_()                 # channels 100..127 are never written: they must read as zero for linear2's padded K
        self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ forward ----------------------------------------------------------
    def forward_hip(self, x: torch.Tensor, train: bool | None = None) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("VGGStyleDiscriminator (HIP) runs on the GPU only; there is no CPU path in this package")
        L.lib()
        if train is None:
            train = self.training
        x = x.contiguous().float()
        n, c, h, w = x.shape
        assert h == self.input_size and w == self.input_size, f"Input size must be identical to input_size, but received {x.size()}."
        if c != self.num_in_ch:
            raise ValueError(f"expected {self.num_in_ch} input channels, got {c}")
        self._prepare()
        ws = self._workspace(n, h, w)
        st, cd, V = self._st, self._cd(), ops.view
        K = dict(n=n, compute=cd)
        ops.nchw_to_nhwc(x, ws["x8"])
        ops.conv3x3(h=h, w=w, cin=8, cout=self.nf, inp=V(ws["x8"]), out=V(ws["a0"]), wpacked=st["conv0_0"]["f"], bias=self.conv0_0.bias.data, act=L.ACT_LRELU,
                    slope=SLOPE, **K)
        cur, hh, ww = ws["a0"], h, w
        for (cname, bname, k), ent in zip(self.stages, ws["layers"]):
            m, bn = getattr(self, cname), getattr(self, bname)
            if k == 4:
                ops.space_to_depth2(V(cur), V(ent["s"]), n, hh, ww)
                hh, ww = hh // 2, ww // 2
                ops.conv3x3(h=hh, w=ww, cin=4 * m.cin, cout=m.cout, inp=V(ent["s"]), out=V(ent["c"]), wpacked=st[cname]["f"], **K)
            else:
                ops.conv3x3(h=hh, w=ww, cin=m.cin, cout=m.cout, inp=V(cur), out=V(ent["c"]), wpacked=st[cname]["f"], **K)
            ops.batchnorm_fwd(V(ent["c"]), V(ent["y"]), n * hh * ww, m.cout, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, ent["mean"], ent["rstd"],
                              st["bnws"], training=train, act=L.ACT_LRELU, slope=SLOPE)
            if train:
                bn._nbt = (bn._nbt or 0) + 1
            cur = ent["y"]
        assert hh == 4 and ww == 4
        cf = 8 * self.nf
        ops.nhwc_to_nchw(V(cur), ws["flat32"])                                   # feat.reshape(n, -1) flattens (c, y, x)
        f32v = ws["flat32"].view(n, 1, 1, cf * 16)
        ops.axpby(V(f32v), 1.0, ops.NULL_VIEW, 0.0, V(ws["flat"]), n)           # -> compute dtype
        l1, l2 = self.linear1, self.linear2
        ops.conv3x3(h=1, w=1, cin=cf * 16, cout=l1.cout, inp=V(ws["flat"]), out=V(ws["h1"]), wpacked=st["linear1"]["f"], bias=l1.bias.data, act=L.ACT_LRELU,
                    slope=SLOPE, ksize=1, **K)
        ops.conv3x3(h=1, w=1, cin=_p32(l2.cin), cout=1, inp=V(ws["h1"]), out=V(ws["o4"]), wpacked=st["linear2"]["f"], bias=l2.bias.data, ksize=1, **K)
        out = ws["o4"].view(n, 4)[:, :1].clone()
        self._saved = (ws, n, h, w, train)
        return out

    # ------------------------------------------------------------------ backward ---------------------------------------------------------
    def backward_hip(self, grad_out: torch.Tensor, *, param_grads: bool = True, input_grad: torch.Tensor | None = None,
                     input_grad_accumulate: bool = True) -> None:
        """grad_out: dL/dlogits (N x 1).  Same contract as UNetDiscriminatorSN.backward_hip."""
        if self._saved is None:
            raise RuntimeError("backward_hip needs a preceding forward_hip")
        ws, n, h, w, train = self._saved
        self._saved = None
        if not train:
            raise RuntimeError("VGGStyleDiscriminator.backward_hip needs a training-mode forward (batch statistics)")
        self._ensure_arena()
        st, cd, V = self._st, self._cd(), ops.view
        dev, cdt = grad_out.device, self.compute_dtype
        K = dict(n=n, compute=cd)
        if "g" not in ws:
            e = lambda *s, dt=cdt: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
            big = n * h * w * self.nf                                      # the largest activation (a0); every gradient buffer is a prefix of these
            need = 0
            hh, ww = h, w
            shapes = [(h, w, 3, self.nf, 3)]
            for (cname, _, k) in self.stages:
                m = getattr(self, cname)
                if k == 4:
                    hh, ww = hh // 2, ww // 2
                    shapes.append((hh, ww, 4 * m.cin, m.cout, 3))
                else:
                    shapes.append((hh, ww, m.cin, m.cout, 3))
            shapes += [(1, 1, 8 * self.nf * 16, 100, 1), (1, 1, 100, 1, 1)]
            for (a, b, ci, co, ks) in shapes:
                need = max(need, ops.wgrad_workspace_bytes(n, a, b, ci, co, ks))
            ws["g"] = dict(A=e(big), B=e(big), C=e(big), wg=torch.empty(need // 4 + 16, dtype=torch.float32, device=dev), g8=e(n, 1, 1, 32),
                           gflat32=e(n, 1, 1, 8 * self.nf * 16, dt=torch.float32), img=e(n, h, w, 4, dt=torch.float32))
        g = ws["g"]

        def buf(name, shape):
            return g[name][: math.prod(shape)].view(shape)

        def wgrad(conv, inp, gy, hh, ww, cin, cout, ks=3, dw=None, db=None):
            if param_grads:
                ops.wgrad3x3(n=n, h=hh, w=ww, cin=cin, cout=cout, compute=cd, inp=inp, gy=gy, dw=dw if dw is not None else conv.weight.grad, db=db, workspace=g["wg"],
                             accumulate=dw is None, ksize=ks)

        l1, l2 = self.linear1, self.linear2
        cf = 8 * self.nf
        g["g8"].zero_()
        g["g8"].view(n, 32)[:, :1].copy_(grad_out.reshape(n, 1).to(cdt))       # (n, 1) logits gradient into the 32-wide token record
        wgrad(l2, V(ws["h1"], 0, 104), V(g["g8"], 0, 8), 1, 1, 100, 1, ks=1, db=l2.bias.grad)      # views in whole 8-channel pieces (16-bit operands)
        gh = buf("A", (n, 1, 1, _p32(100)))
        gh.zero_()
        ops.conv3x3(h=1, w=1, cin=32, cout=100, inp=V(g["g8"]), out=V(gh), wpacked=st["linear2"]["b"], mask=V(ws["h1"]), mask_slope=SLOPE, ksize=1, **K)
        wgrad(l1, V(ws["flat"]), V(gh, 0, 104), 1, 1, cf * 16, 100, ks=1, db=l1.bias.grad)
        ops.conv3x3(h=1, w=1, cin=_p32(100), cout=cf * 16, inp=V(gh), out=V(g["gflat32"]), wpacked=st["linear1"]["b"], ksize=1, **K)
        gy = buf("B", (n, 4, 4, cf))
        ops.nchw_to_nhwc(g["gflat32"].view(n, cf, 4, 4), gy)                   # back to NHWC: the gradient at the last activation
        names = ["A", "C", "B"]
        for idx in range(len(self.stages) - 1, -1, -1):
            cname, bname, k = self.stages[idx]
            m, bn, ent = getattr(self, cname), getattr(self, bname), ws["layers"][idx]
            hh, ww = ent["h"], ent["w"]
            free = [nm for nm in names if g[nm].data_ptr() != gy.data_ptr()]
            gc = buf(free[0], (n, hh, ww, m.cout))
            ops.batchnorm_bwd(V(gy), V(ent["c"]), V(ent["y"]), V(gc), n * hh * ww, m.cout, bn.weight.data, ent["mean"], ent["rstd"],
                              bn.weight.grad if param_grads else None, bn.bias.grad if param_grads else None, st["bnws"], slope=SLOPE, accumulate=True)
            prev = ws["layers"][idx - 1]["y"] if idx > 0 else ws["a0"]
            if k == 4:
                if param_grads:
                    ops.wgrad3x3(n=n, h=hh, w=ww, cin=4 * m.cin, cout=m.cout, compute=cd, inp=V(ent["s"]), gy=V(gc), dw=st[cname]["g3"], db=None, workspace=g["wg"])
                    ops.collapse_g3x3s2d(st[cname]["g3"], m.weight.grad, accumulate=True)
                gs = buf(free[1], (n, hh, ww, 4 * m.cin))
                ops.conv3x3(h=hh, w=ww, cin=m.cout, cout=4 * m.cin, inp=V(gc), out=V(gs), wpacked=st[cname]["b"], **K)
                gy = buf(free[0], (n, 2 * hh, 2 * ww, m.cin))                  # gc is dead: its buffer takes the full-resolution gradient
                ops.depth_to_space2(V(gs), V(gy), n, 2 * hh, 2 * ww)
            else:
                wgrad(m, V(prev), V(gc), hh, ww, m.cin, m.cout)
                gy = buf(free[1], (n, hh, ww, m.cin))
                ops.conv3x3(h=hh, w=ww, cin=m.cout, cout=m.cin, inp=V(gc), out=V(gy), wpacked=st[cname]["b"], **K)
        # gy = dL/da0 ; a0 = lrelu(conv0_0(img) + bias)
        free = [nm for nm in names if g[nm].data_ptr() != gy.data_ptr()]
        gc0 = buf(free[0], (n, h, w, self.nf))
        ops.axpby(V(gy), 1.0, ops.NULL_VIEW, 0.0, V(gc0), n * h * w, mask=V(ws["a0"]), mask_slope=SLOPE)
        c0 = self.conv0_0
        wgrad(c0, V(ws["x8"]), V(gc0), h, w, c0.cin, c0.cout, db=c0.bias.grad)
        if input_grad is not None:
            ops.conv3x3(h=h, w=w, cin=self.nf, cout=self.num_in_ch, inp=V(gc0), out=V(g["img"]), wpacked=st["conv0_0"]["b"], **K)
            ops.nhwc_to_nchw_scaled(V(g["img"]), input_grad, [1.0] * 3, accumulate=input_grad_accumulate)

    # ------------------------------------------------------------------ nn.Module surface ------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if self._trigger is None or self._trigger.device != x.device:
                self._trigger = torch.zeros(1, device=x.device, requires_grad=True)
            return _VGGDFn.apply(x, self._trigger, self)
        return self.forward_hip(x)


class _VGGDFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, trigger, net):
        ctx.net = net
        ctx.need_x = x.requires_grad
        ctx.params = any(p.requires_grad for p in net.parameters())
        ctx.shape = x.shape
        return net.forward_hip(x)

    @staticmethod
    def backward(ctx, gy):
        gx = torch.zeros(ctx.shape, device=gy.device) if ctx.need_x else None
        ctx.net.backward_hip(gy, param_grads=ctx.params, input_grad=gx, input_grad_accumulate=False)
        return gx, torch.zeros(1, device=gy.device), None
