"""SPAN (`span`, `span_s`, `span_f32/f64/f96`) on the HIP kernels (gfx950).

Same registry names, factory signatures and state-dict keys as /root/reference/traiNNer/archs/span_arch.py:326-420 (incl. the
non-trainable `eval_conv.*` merged kernels and the `no_norm` buffer, so checkpoints interchange with spandrel).  Pinned to the
reference's own module: tests/golden/span.pt.  Checksums: span 2,621,964 / span_s 2,236,728 parameters (docs/source/benchmarks.md:25,27).

Execution.  Training AND inference run Conv3XC in its three-conv form (span_arch.py:197-204) -- the merged 3x3 of `update_params`
is the same function (the 1x1 runs over the zero-PADDED input, so its bias reaches the 3x3's border taps), kept current in
`eval_conv.*` only for export:
  t1  = 1x1 (ci -> 2ci) over the (h+2) x (w+2) grid, reading x at (y-1, x-1) with zeros outside      tr_conv3x3 ksize 1, TR_IN_FULL
  t2  = 3x3 valid (2ci -> 2co)                                                                        tr_conv3x3 ksize 3, TR_IN_VALID
  y   = 1x1 over the CONCATENATED channels [t2 | x] with [W3 | W_sk] (+ b3 + b_sk): conv[2] and the 1x1 skip in ONE launch, with
        SiLU (pre-activation kept for the backward) fused in its epilogue
t2 is written into the first channels of a buffer whose tail already holds x (the producer of x wrote it there), so there is no
concat and no separate skip branch.  SPAB's `(out3 + x) * (sigmoid(out3) - 0.5)` is one element-wise kernel writing the fp32 value
(the next block's `x`) and the compute-dtype copy (the next block's conv operand).  Reference quirk kept: `act1` is
`nn.SiLU(inplace=True)`, so the `out1` a block returns -- block 6's goes into the final concat -- is the ACTIVATED tensor.
Backward is hand-scheduled over the same kernels (dgrad of the valid conv = TR_IN_FULL with the mirrored slab, etc.).
"""
from __future__ import annotations

import torch
from torch import nn
from torch.nn import functional as F

from .. import _lib as L
from .. import ops
from ..utils.registry import ARCH_REGISTRY, SPANDREL_REGISTRY
from ._arena import FlatParamModule

RGB_MEAN = (0.4488, 0.4371, 0.4040)


def _p8(c: int) -> int:
    return (c + 7) // 8 * 8


class _Conv(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        ref = nn.Conv2d(cin, cout, k)
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone())


class _Seq(nn.Module):
    def __init__(self, mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)


class Conv3XC(nn.Module):
    """Parameter holder with the reference's key layout: sk, conv.{0,1,2}, eval_conv (requires_grad False)."""

    def __init__(self, c_in, c_out, gain1=2):
        super().__init__()
        self.ci, self.co, self.gain = c_in, c_out, gain1
        self.sk = _Conv(c_in, c_out, 1)
        self.conv = _Seq([_Conv(c_in, c_in * gain1, 1), _Conv(c_in * gain1, c_out * gain1, 3), _Conv(c_out * gain1, c_out, 1)])
        self.eval_conv = _Conv(c_in, c_out, 3)
        self.eval_conv.weight.requires_grad = False
        self.eval_conv.bias.requires_grad = False
        self.update_params()

    @torch.no_grad()
    def update_params(self) -> None:
        """span_arch.py:160-195 -- export-time merge of the three convs and the skip into `eval_conv` (not on the hot path)."""
        c0, c1, c2 = getattr(self.conv, "0"), getattr(self.conv, "1"), getattr(self.conv, "2")
        w1, b1, w2, b2, w3, b3 = c0.weight, c0.bias, c1.weight, c1.bias, c2.weight, c2.bias
        w = F.conv2d(w1.flip(2, 3).permute(1, 0, 2, 3), w2, padding=2).flip(2, 3).permute(1, 0, 2, 3)
        b = (w2 * b1.reshape(1, -1, 1, 1)).sum((1, 2, 3)) + b2
        wc = F.conv2d(w.flip(2, 3).permute(1, 0, 2, 3), w3).flip(2, 3).permute(1, 0, 2, 3)
        bc = (w3 * b.reshape(1, -1, 1, 1)).sum((1, 2, 3)) + b3
        self.eval_conv.weight.copy_(wc + F.pad(self.sk.weight, [1, 1, 1, 1]))
        self.eval_conv.bias.copy_(bc + self.sk.bias)


class SPAB(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c1_r, self.c2_r, self.c3_r = Conv3XC(c, c), Conv3XC(c, c), Conv3XC(c, c)


class SPAN(FlatParamModule):
    def __init__(self, *, num_in_ch: int = 3, num_out_ch: int = 3, feature_channels: int = 48, upscale: int = 4, bias: bool = True,
                 norm: bool = True, img_range: float = 255.0, rgb_mean=RGB_MEAN) -> None:
        super().__init__()
        if num_in_ch != 3 or num_out_ch != 3:
            raise NotImplementedError("SPAN (HIP): 3-channel images only")
        if feature_channels % 4:
            raise NotImplementedError("SPAN (HIP): feature_channels must be a multiple of 4")
        self.C, self.scale, self.img_range, self.rgb_mean = feature_channels, upscale, float(img_range), tuple(float(v) for v in rgb_mean)
        self.hyperparameters = dict(num_in_ch=num_in_ch, num_out_ch=num_out_ch, feature_channels=feature_channels, upscale=upscale, bias=bias,
                                    norm=norm, img_range=img_range, rgb_mean=tuple(rgb_mean))
        if not norm:
            self.register_buffer("no_norm", torch.zeros(1))
        else:
            self.no_norm = None
        C = feature_channels
        self.conv_1 = Conv3XC(num_in_ch, C)
        for b in range(1, 7):
            setattr(self, f"block_{b}", SPAB(C))
        self.conv_cat = _Conv(4 * C, C, 1)
        self.conv_2 = Conv3XC(C, C)
        self.upsampler = _Seq([_Conv(C, num_out_ch * upscale * upscale, 3)])
        self.compute_dtype = torch.bfloat16
        self._init_arena_state()

    @property
    def is_norm(self) -> bool:
        return self.no_norm is None

    def _frozen_param_names(self) -> set:
        return {n for n, _ in self.named_parameters() if ".eval_conv." in n}

    def _units(self):
        yield "conv_1", self.conv_1
        for b in range(1, 7):
            blk = getattr(self, f"block_{b}")
            for k in (1, 2, 3):
                yield f"block_{b}.c{k}_r", getattr(blk, f"c{k}_r")
        yield "conv_2", self.conv_2

    def update_params(self) -> None:
        for _, m in self._units():
            m.update_params()

    def state_dict(self, *args, **kwargs):
        self.update_params()                 # the exported merged kernels always match the trained branches
        return super().state_dict(*args, **kwargs)

    def _cd(self):
        return ops.dt_code(self.compute_dtype)

    # ------------------------------------------------------------------ slabs ------------------------------------------------
    def repack(self):
        self._ensure_arena()
        dev, cd = self._arena.device, self._cd()
        key = (cd, str(dev), self._arena.data_ptr())
        st = self.__dict__.setdefault("_slabs", {})
        if st.get("key") != key:
            st.clear()
            st["key"] = key
            plan = ops.PackPlan(dev)

            def z(cin, cout, ks):
                return torch.zeros(ops.packed_bytes(cin, cout, cd, ks), dtype=torch.uint8, device=dev)

            for name, m in self._units():
                ci, co = m.ci, m.co
                cip, gi, gip, go, gop, cop = _p8(ci), 2 * ci, _p8(2 * ci), 2 * co, _p8(2 * co), _p8(co)
                c0, c1, c2 = getattr(m.conv, "0"), getattr(m.conv, "1"), getattr(m.conv, "2")
                e = {}
                e["w1"] = z(cip, gi, 1); plan.add(c0.weight.data, e["w1"], cip, gi, cd)
                e["w2"] = z(gip, go, 3); plan.add(c1.weight.data, e["w2"], gip, go, cd)
                e["w3"] = z(gop + cip, co, 1)                       # [W3 | W_sk] over the concatenated channels [t2 | x]
                plan.add(c2.weight.data, e["w3"], gop + cip, co, cd)
                plan.add(m.sk.weight.data, e["w3"], gop + cip, co, cd, dst_i_off=gop)
                e["w3t"] = z(cop, gop + cip, 1)                     # gradient wrt [t2 | x]
                plan.add(c2.weight.data, e["w3t"], cop, gop + cip, cd, transpose=True)
                plan.add(m.sk.weight.data, e["w3t"], cop, gop + cip, cd, transpose=True, dst_o_off=gop)
                e["w2t"] = z(gop, gi, 3); plan.add(c1.weight.data, e["w2t"], gop, gi, cd, transpose=True)
                e["w1t"] = z(gip, ci, 1); plan.add(c0.weight.data, e["w1t"], gip, ci, cd, transpose=True)
                e["bsum"] = torch.zeros(co, dtype=torch.float32, device=dev)
                st[name] = e
            C, Cp = self.C, _p8(self.C)
            st["cat"] = z(4 * Cp, C, 1)
            st["cat_t"] = z(Cp, 4 * Cp, 1)
            for k in range(4):
                plan.add(self.conv_cat.weight.data, st["cat"], 4 * Cp, C, cd, i_off=k * C, i_cnt=C, dst_i_off=k * Cp)
                plan.add(self.conv_cat.weight.data, st["cat_t"], Cp, 4 * Cp, cd, i_off=k * C, i_cnt=C, dst_o_off=k * Cp, transpose=True)
            up = getattr(self.upsampler, "0")
            co_up = up.weight.shape[0]
            st["up"] = z(Cp, co_up, 3); plan.add(up.weight.data, st["up"], Cp, co_up, cd)
            st["up_t"] = z(_p8(co_up), C, 3); plan.add(up.weight.data, st["up_t"], _p8(co_up), C, cd, transpose=True)
            st["plan"] = plan.finalize()
        st["plan"].run()
        for name, m in self._units():                                # b3 + b_sk: one bias for the fused [conv[2] | skip] launch
            c2 = getattr(m.conv, "2")
            ops.axpby(ops.view(c2.bias.data.view(1, -1)), 1.0, ops.view(m.sk.bias.data.view(1, -1)), 1.0, ops.view(st[name]["bsum"].view(1, -1)), 1)

    # ------------------------------------------------------------------ forward ------------------------------------------------
    def forward_hip(self, x: torch.Tensor, train: bool | None = None) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("SPAN (HIP) runs on the GPU only; there is no CPU path in this package")
        L.lib()
        if train is None:
            train = torch.is_grad_enabled() and self.training
        x = x.contiguous().float()
        B, cin, h, w = x.shape
        if cin != 3 or h < 3 or w < 3:
            raise NotImplementedError("SPAN (HIP) needs 3-channel inputs of at least 3x3 pixels")
        self.repack()
        dev, cdt, cd = x.device, self.compute_dtype, self._cd()
        st, V, f32 = self._slabs, ops.view, torch.float32
        Z = self._pool("fwd", dev, cdt)
        C, Cp, r = self.C, _p8(self.C), self.scale
        T = B * h * w
        bufs = {}
        for name, m in self._units():
            bufs[name] = dict(t1=Z(B, h + 2, w + 2, _p8(2 * m.ci)), cat=Z(B, h, w, _p8(2 * m.co) + _p8(m.ci)))

        units = dict(self._units())

        def tail(name):
            m = units[name]
            return V(bufs[name]["cat"], _p8(2 * m.co), _p8(m.ci))

        def run(name, out, out2=ops.NULL_VIEW, act=L.ACT_NONE):
            m, b, e = units[name], bufs[name], st[name]
            c0, c1 = getattr(m.conv, "0"), getattr(m.conv, "1")
            ops.conv3x3(n=B, h=h + 2, w=w + 2, cin=_p8(m.ci), cout=2 * m.ci, compute=cd, inp=tail(name), out=V(b["t1"]), wpacked=e["w1"],
                        bias=c0.bias.data, ksize=1, in_mode=L.IN_FULL)
            ops.conv3x3(n=B, h=h, w=w, cin=_p8(2 * m.ci), cout=2 * m.co, compute=cd, inp=V(b["t1"]), out=V(b["cat"], 0, _p8(2 * m.co)),
                        wpacked=e["w2"], bias=c1.bias.data, in_mode=L.IN_VALID)
            ops.conv3x3(n=B, h=h, w=w, cin=_p8(2 * m.co) + _p8(m.ci), cout=m.co, compute=cd, inp=V(b["cat"]), out=out, out2=out2, act=act,
                        wpacked=e["w3"], bias=e["bsum"], ksize=1)

        if self.is_norm:
            ops.nchw_to_nhwc(x, bufs["conv_1"]["cat"], self.rgb_mean, [1.0 / self.img_range] * 3, dst_view=tail("conv_1"))
        else:
            ops.nchw_to_nhwc(x, bufs["conv_1"]["cat"], dst_view=tail("conv_1"))
        ccat = Z(B, h, w, 4 * Cp)
        feat32 = Z(B, h, w, Cp, dt=f32)
        run("conv_1", V(feat32), out2=tail("block_1.c1_r"))
        ops.axpby(V(feat32), 1.0, ops.NULL_VIEW, 0.0, V(ccat, 0, Cp), T)
        sv = dict(B=B, h=h, w=w, bufs=bufs, ccat=ccat, feat32=feat32, blocks=[])
        x32 = feat32
        for b in range(1, 7):
            p = f"block_{b}"
            pre1, pre2, o3, out32 = Z(B, h, w, Cp), Z(B, h, w, Cp), Z(B, h, w, Cp, dt=f32), Z(B, h, w, Cp, dt=f32)
            run(p + ".c1_r", tail(p + ".c2_r"), out2=V(pre1), act=L.ACT_SILU)
            run(p + ".c2_r", tail(p + ".c3_r"), out2=V(pre2), act=L.ACT_SILU)
            run(p + ".c3_r", V(o3))
            ops.span_gate(V(o3), V(x32), V(out32), tail(f"block_{b + 1}.c1_r") if b < 6 else tail("conv_2"), T)
            if b == 1:
                ops.axpby(V(out32), 1.0, ops.NULL_VIEW, 0.0, V(ccat, 2 * Cp, Cp), T)
            if b == 6:      # the in-place SiLU quirk: the block's `out1` is silu(c1_r(x)), i.e. exactly c2_r's input
                ops.axpby(tail(p + ".c2_r"), 1.0, ops.NULL_VIEW, 0.0, V(ccat, 3 * Cp, Cp), T)
            sv["blocks"].append(dict(pre1=pre1, pre2=pre2, o3=o3, x32=x32, out32=out32))
            x32 = out32
        run("conv_2", V(ccat, Cp, Cp))
        cc = Z(B, h, w, Cp)
        ops.conv3x3(n=B, h=h, w=w, cin=4 * Cp, cout=C, compute=cd, inp=V(ccat), out=V(cc), wpacked=st["cat"], bias=self.conv_cat.bias.data, ksize=1)
        up = getattr(self.upsampler, "0")
        cup = _p8(3 * r * r)
        u = Z(B, h, w, cup, dt=f32)
        ops.conv3x3(n=B, h=h, w=w, cin=Cp, cout=3 * r * r, compute=cd, inp=V(cc), out=V(u), wpacked=st["up"], bias=up.bias.data)
        o4 = Z(B, h * r, w * r, 4, dt=f32)
        ops.pixel_shuffle(V(u), V(o4), B, h, w, 3, r)
        out = torch.empty(B, 3, h * r, w * r, dtype=f32, device=dev)
        ops.nhwc_to_nchw(V(o4), out)
        sv["cc"] = cc
        self._saved = sv if train else None
        return out

    # ------------------------------------------------------------------ backward ------------------------------------------------
    def backward_hip(self, grad_out: torch.Tensor) -> None:
        if self._saved is None:
            raise RuntimeError("backward_hip needs a preceding training-mode forward_hip")
        sv = self._saved
        self._saved = None
        self._ensure_arena()
        B, h, w, bufs, ccat = sv["B"], sv["h"], sv["w"], sv["bufs"], sv["ccat"]
        dev, cdt, cd = grad_out.device, self.compute_dtype, self._cd()
        st, V, f32 = self._slabs, ops.view, torch.float32
        Z = self._pool("bwd", dev, cdt)
        C, Cp, r = self.C, _p8(self.C), self.scale
        T, H, W = B * h * w, h * r, w * r
        units = dict(self._units())
        cup = _p8(3 * r * r)
        need = 0
        for (hh, ww, ci, co, ks) in [(h, w, 2 * C, C, 1), (h, w, 2 * C, 2 * C, 3), (h + 2, w + 2, C, 2 * C, 1), (h, w, C, 3 * r * r, 3), (h, w, C, C, 1),
                                     (h, w, 6, 2 * C, 3), (h + 2, w + 2, 3, 6, 1), (h, w, 3, C, 1)]:
            need = max(need, ops.wgrad_workspace_bytes(B, hh, ww, ci, co, ks))
        wg = Z(need // 4 + 16, dt=f32)

        def tail(name):
            m = units[name]
            return V(bufs[name]["cat"], _p8(2 * m.co), _p8(m.ci))

        def wgrad(conv, inp, gy, hh, ww, ks, cin, in_mode=L.IN_DIRECT, db=True, **kw):
            ops.wgrad3x3(n=B, h=hh, w=ww, cin=cin, cout=conv.weight.shape[0], compute=cd, inp=inp, gy=gy, dw=conv.weight.grad,
                         db=conv.bias.grad if db else None, workspace=wg, accumulate=True, ksize=ks, in_mode=in_mode, **kw)

        g_cat = {}

        def bwd(name, gy, g_in=None, res2=ops.NULL_VIEW, mask=ops.NULL_VIEW, mask_mode=0):
            """Conv3XC backward: gy = d/d(output before any activation); g_in (+ epilogue terms) = d/d(input)."""
            m, b, e = units[name], bufs[name], st[name]
            c0, c1, c2 = getattr(m.conv, "0"), getattr(m.conv, "1"), getattr(m.conv, "2")
            gop, cip, gip = _p8(2 * m.co), _p8(m.ci), _p8(2 * m.ci)
            wgrad(c2, V(b["cat"], 0, gop), gy, h, w, 1, 2 * m.co)
            wgrad(m.sk, tail(name), gy, h, w, 1, m.ci)
            gc = g_cat.get((gop, cip))
            if gc is None:
                gc = g_cat[(gop, cip)] = (Z(B, h, w, gop + cip), Z(B, h + 2, w + 2, gip))
            gcat, gt1 = gc
            ops.conv3x3(n=B, h=h, w=w, cin=_p8(m.co), cout=gop + cip, compute=cd, inp=gy, out=V(gcat), wpacked=e["w3t"], ksize=1)
            wgrad(c1, V(b["t1"]), V(gcat, 0, gop), h, w, 3, 2 * m.ci, in_mode=L.IN_VALID)
            ops.conv3x3(n=B, h=h + 2, w=w + 2, cin=gop, cout=2 * m.ci, compute=cd, inp=V(gcat, 0, gop), out=V(gt1), wpacked=e["w2t"],
                        in_mode=L.IN_FULL)
            wgrad(c0, tail(name), V(gt1), h + 2, w + 2, 1, m.ci, in_mode=L.IN_FULL)
            if g_in is not None:
                ops.conv3x3(n=B, h=h, w=w, cin=gip, cout=m.ci, compute=cd, inp=V(gt1), out=g_in, wpacked=e["w1t"], ksize=1, in_mode=L.IN_VALID,
                            res1=V(gcat, gop, cip), res2=res2, mask=mask, mask_mode=mask_mode)

        g8 = Z(B, H, W, 8)
        ops.nchw_to_nhwc(grad_out.contiguous().float(), g8)
        gu = Z(B, h, w, cup)
        ops.pixel_shuffle(V(g8), V(gu), B, h, w, 3, r, inverse=True)
        up = getattr(self.upsampler, "0")
        wgrad(up, V(sv["cc"]), V(gu), h, w, 3, C)
        g_cc = Z(B, h, w, Cp)
        ops.conv3x3(n=B, h=h, w=w, cin=cup, cout=C, compute=cd, inp=V(gu), out=V(g_cc), wpacked=st["up_t"])
        for k in range(4):
            wgrad(self.conv_cat, V(ccat, k * Cp, Cp), V(g_cc), h, w, 1, C, db=k == 0, dw_ci0=k * C, dw_ld=4 * C)
        g_ccat = Z(B, h, w, 4 * Cp)
        ops.conv3x3(n=B, h=h, w=w, cin=Cp, cout=4 * Cp, compute=cd, inp=V(g_cc), out=V(g_ccat), wpacked=st["cat_t"], ksize=1)
        G = [Z(B, h, w, Cp, dt=f32) for _ in range(2)]       # d/d(block output), ping-pong
        bwd("conv_2", V(g_ccat, Cp, Cp), g_in=V(G[0]))
        g3, gy2, gy1, gxg = Z(B, h, w, Cp), Z(B, h, w, Cp), Z(B, h, w, Cp), Z(B, h, w, Cp, dt=f32)
        cur = 0
        for b in range(6, 0, -1):
            p, s = f"block_{b}", sv["blocks"][b - 1]
            if b == 1:      # out_b1 also feeds the final concat
                ops.axpby(V(G[cur]), 1.0, V(g_ccat, 2 * Cp, Cp), 1.0, V(G[cur]), T)
            ops.span_gate_bwd(V(G[cur]), V(s["o3"]), V(s["x32"]), V(g3), V(gxg), T)
            bwd(p + ".c3_r", V(g3), g_in=V(gy2), mask=V(s["pre2"]), mask_mode=2)
            bwd(p + ".c2_r", V(gy2), g_in=V(gy1), mask=V(s["pre1"]), mask_mode=2, res2=V(g_ccat, 3 * Cp, Cp) if b == 6 else ops.NULL_VIEW)
            bwd(p + ".c1_r", V(gy1), g_in=V(G[cur ^ 1]), res2=V(gxg))
            cur ^= 1
        ops.axpby(V(G[cur]), 1.0, V(g_ccat, 0, Cp), 1.0, V(g3), T)     # d/d(feat): block 1's input + its slice of the concat
        bwd("conv_1", V(g3))

    # ------------------------------------------------------------------ nn.Module surface ------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            if self._trigger is None or self._trigger.device != x.device:
                self._trigger = torch.zeros(1, device=x.device, requires_grad=True)
            return _SpanFn.apply(x, self._trigger, self)
        return self.forward_hip(x, train=False)


class _SpanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, trigger, net):
        ctx.net = net
        return net.forward_hip(x, train=True)

    @staticmethod
    def backward(ctx, gy):
        ctx.net.backward_hip(gy)
        return None, torch.zeros(1, device=gy.device), None


def _factory(default_fc):
    def make(num_in_ch: int = 3, num_out_ch: int = 3, feature_channels: int = default_fc, scale: int = 4, bias: bool = True, norm: bool = False,
             img_range: float = 255.0, rgb_mean: tuple[float, float, float] = RGB_MEAN) -> SPAN:
        return SPAN(upscale=scale, num_in_ch=num_in_ch, num_out_ch=num_out_ch, feature_channels=feature_channels, bias=bias, norm=norm,
                    img_range=img_range, rgb_mean=rgb_mean)
    return make


span = _factory(52); span.__name__ = "span"; SPANDREL_REGISTRY.register()(span)
span_s = _factory(48); span_s.__name__ = "span_s"; SPANDREL_REGISTRY.register()(span_s)
span_f32 = _factory(32); span_f32.__name__ = "span_f32"; ARCH_REGISTRY.register()(span_f32)
span_f64 = _factory(64); span_f64.__name__ = "span_f64"; ARCH_REGISTRY.register()(span_f64)
span_f96 = _factory(96); span_f96.__name__ = "span_f96"; ARCH_REGISTRY.register()(span_f96)
