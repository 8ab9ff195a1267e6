"""MoSR (`mosr`, `mosr_t`) on the HIP kernels (gfx950).

Same registry names and factory signatures as /root/reference/traiNNer/archs/mosr_arch.py:23-74 (the reference forwards them to
`spandrel.architectures.MoSR.MoSR`, a wheel that is absent here: the module body follows oracle/mosr.py, which documents its sources and
reproduces all six published parameter counts, docs/source/benchmarks.md:43,132,211,226).  State-dict keys: `gblocks.N.*`,
`shortcut.block.N.*`, `shortcut.conv11.*`, `upsampler.0.*`.

Execution (per GatedCNNBlock, tokens = pixels of NHWC buffers, residual stream fp32):
  xn   = LayerNorm over channels (eps 1e-6)                                     tr_layernorm
  F    = fc1 3x3 (D -> 2 hid) = [g | i | c]            ONE buffer, the split is three channel-slice views      tr_conv3x3
  Cq   = depthwise k x k conv of the c slice                                    tr_dwconv
  P    = mish(g) * cat(i, Cq)                                                   tr_gate_mish (the cat is two views)
  x'   = mish(fc2 3x3 (P)) * keep + x    Mish, DropPath scale and residual in the conv epilogue, pre-activation kept     tr_conv3x3
Backward is hand-scheduled over the same kernels (+ tr_act_grad, tr_gate_mish_bwd, tr_dwconv flip / tr_dwconv_wgrad, tr_layernorm_bwd).
Scope: upsampler "pixelshuffle"; "dysample" / "geoensemblepixelshuffle" raise NotImplementedError.
"""
from __future__ import annotations

from typing import Literal

import torch
from torch import nn

from .. import _lib as L
from .. import ops
from ..utils.registry import ARCH_REGISTRY
from ._arena import FlatParamModule

upsampler_type = Literal["pixelshuffle", "dysample", "geoensemblepixelshuffle"]
upsampler_map = {"pixelshuffle": "ps", "dysample": "dys", "geoensemblepixelshuffle": "gps"}


def get_upsampler(s: str) -> str:
    return upsampler_map.get(s, s)


def _p8(c: int) -> int:
    return (c + 7) // 8 * 8


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, groups=1):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, k, k))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.trunc_normal_(self.weight, std=0.02)       # `_init_weights` of the block's twins (mosrv2_arch.py:110-115)


class _Norm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _Mish(nn.Module):
    """Parameter-free slot so that the Sequential indices (and therefore the state-dict keys) match the published module."""


class GatedCNNBlock(nn.Module):
    def __init__(self, dim, hidden, cc, k):
        super().__init__()
        self.norm = _Norm(dim)
        self.fc1 = _Conv(dim, 2 * hidden, 3)
        self.conv = _Conv(cc, cc, k, groups=cc)
        self.fc2 = _Conv(hidden, dim, 3)


class _Seq(nn.Module):
    def __init__(self, mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def __getitem__(self, i):
        return getattr(self, str(i))


class _ConvBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.block = _Seq([_Conv(cin, cout, 3), _Mish(), _Conv(cout, cout, 3), _Mish()])
        self.conv11 = _Conv(cin, cout, 1)


class MoSR(FlatParamModule):
    def __init__(self, *, in_ch: int = 3, out_ch: int = 3, upscale: int = 4, n_block: int = 24, dim: int = 64, upsampler: str = "ps",
                 drop_path: float = 0.0, kernel_size: int = 7, expansion_ratio: float = 1.5, conv_ratio: float = 1.0) -> None:
        super().__init__()
        if in_ch != 3 or out_ch != 3:
            raise NotImplementedError("MoSR (HIP): 3-channel images only")
        if upsampler != "ps":
            raise NotImplementedError(f"MoSR (HIP): upsampler {upsampler!r} is not built (pixelshuffle only)")
        hidden, cc = int(expansion_ratio * dim), int(conv_ratio * dim)
        if dim % 8 or hidden % 8 or cc % 4 or (hidden - cc) % 4 or not 0 < cc < hidden or kernel_size not in (3, 5, 7):
            raise NotImplementedError(f"MoSR (HIP): dim {dim} / hidden {hidden} / conv channels {cc} / kernel {kernel_size} outside the built shapes "
                                      "(dim, hidden multiples of 8; 0 < conv channels < hidden, multiples of 4; kernel 3, 5 or 7)")
        self.D, self.hid, self.cc, self.k, self.nb, self.scale = dim, hidden, cc, kernel_size, n_block, upscale
        self.drop_path = float(drop_path)
        self.hyperparameters = dict(in_ch=in_ch, out_ch=out_ch, upscale=upscale, n_block=n_block, dim=dim, upsampler=upsampler, drop_path=drop_path,
                                    kernel_size=kernel_size, expansion_ratio=expansion_ratio, conv_ratio=conv_ratio)
        self.gblocks = _Seq([_Conv(in_ch, dim, 3)] + [GatedCNNBlock(dim, hidden, cc, kernel_size) for _ in range(n_block)] +
                            [_Conv(dim, 2 * dim, 3), _Mish(), _Conv(2 * dim, dim, 3), _Mish(), _Conv(dim, dim, 1)])
        self.shortcut = _ConvBlock(in_ch, dim)
        self.upsampler = _Seq([_Conv(dim, out_ch * upscale * upscale, 3)])
        self.compute_dtype = torch.bfloat16
        self._init_arena_state()

    def _cd(self):
        return ops.dt_code(self.compute_dtype)

    def _blocks(self):
        return [self.gblocks[b] for b in range(1, self.nb + 1)]

    def _keep_masks(self, b, dev):
        """DropPath: per block and sample, keep / (1 - p) with p rising linearly over the blocks (`torch.linspace(0, drop_path, n_block)`)."""
        rates = torch.linspace(0, self.drop_path, self.nb).tolist()
        out = {}
        for i, p in enumerate(rates):
            if p > 0:
                out[i + 1] = (torch.rand(b, device=dev) >= p).float() / (1.0 - p)
        return out

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
            D, hid, nb = self.D, self.hid, self.nb

            def slab(conv, cin, cout, ks, transpose=False):
                ci, co = (cout, cin) if transpose else (cin, cout)
                t = torch.zeros(ops.packed_bytes(_p8(ci), co, cd, ks), dtype=torch.uint8, device=dev)
                plan.add(conv.weight.data, t, _p8(ci), co, cd, transpose=transpose)
                return t

            st["head"] = slab(self.gblocks[0], 3, D, 3)
            for b, blk in enumerate(self._blocks()):
                st[b] = dict(w1=slab(blk.fc1, D, 2 * hid, 3), w1t=slab(blk.fc1, D, 2 * hid, 3, True), w2=slab(blk.fc2, hid, D, 3),
                             w2t=slab(blk.fc2, hid, D, 3, True))
            t1, t2, t3 = self.gblocks[nb + 1], self.gblocks[nb + 3], self.gblocks[nb + 5]
            st["t1"], st["t1t"] = slab(t1, D, 2 * D, 3), slab(t1, D, 2 * D, 3, True)
            st["t2"], st["t2t"] = slab(t2, 2 * D, D, 3), slab(t2, 2 * D, D, 3, True)
            st["t3"], st["t3t"] = slab(t3, D, D, 1), slab(t3, D, D, 1, True)
            s1, s2 = self.shortcut.block[0], self.shortcut.block[2]
            st["s1"] = slab(s1, 3, D, 3)
            st["s2"], st["s2t"] = slab(s2, D, D, 3), slab(s2, D, D, 3, True)
            st["s11"] = slab(self.shortcut.conv11, 3, D, 1)
            up = self.upsampler[0]
            st["up"], st["upt"] = slab(up, D, up.weight.shape[0], 3), slab(up, D, up.weight.shape[0], 3, True)
            st["plan"] = plan.finalize()
        st["plan"].run()

    # ------------------------------------------------------------------ forward ------------------------------------------------
    def forward_hip(self, x: torch.Tensor, train: bool | None = None, keep_masks: dict | None = None) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("MoSR (HIP) runs on the GPU only; there is no CPU path in this package")
        L.lib()
        if train is None:
            train = torch.is_grad_enabled() and self.training
        x = x.contiguous().float()
        B, cin, h, w = x.shape
        if cin != 3:
            raise NotImplementedError("MoSR (HIP) needs 3-channel inputs")
        self.repack()
        dev, cdt, cd = x.device, self.compute_dtype, self._cd()
        st, V, f32 = self._slabs, ops.view, torch.float32
        Z = self._pool("fwd", dev, cdt)
        D, hid, cc, nb, r = self.D, self.hid, self.cc, self.nb, self.scale
        T = B * h * w
        if train and keep_masks is None and self.drop_path > 0:
            keep_masks = self._keep_masks(B, dev)
        keep_masks = keep_masks or {}
        conv = ops.conv3x3
        geo = dict(n=B, h=h, w=w, compute=cd)

        x8 = Z(B, h, w, 8)
        ops.nchw_to_nhwc(x, x8)
        xs = Z(B, h, w, D, dt=f32)
        head = self.gblocks[0]
        conv(cin=8, cout=D, inp=V(x8), out=V(xs), wpacked=st["head"], bias=head.bias.data, **geo)
        sv = dict(B=B, h=h, w=w, x8=x8, blocks=[], keep=keep_masks)
        xl = None
        for b, blk in enumerate(self._blocks()):
            last = b == nb - 1
            xn, Fb, Cq, P, pre = Z(B, h, w, D), Z(B, h, w, 2 * hid), Z(B, h, w, cc), Z(B, h, w, hid), Z(B, h, w, D)
            mean, rstd = Z(T, dt=f32), Z(T, dt=f32)
            ops.layernorm(V(xs), V(xn), T, D, blk.norm.weight.data, blk.norm.bias.data, 1e-6, mean, rstd)
            conv(cin=D, cout=2 * hid, inp=V(xn), out=V(Fb), wpacked=st[b]["w1"], bias=blk.fc1.bias.data, **geo)
            ops.dwconv(V(Fb, 2 * hid - cc, cc), V(Cq), blk.conv.weight.data, blk.conv.bias.data, B, h, w, cc)
            ops.gate_mish(V(Fb, 0, hid), V(Fb, hid, hid - cc), V(Cq), V(P), T)
            # the last block's output feeds a conv only: it is written in the compute dtype; the others continue the fp32 stream
            nxt = Z(B, h, w, D) if last else Z(B, h, w, D, dt=f32)
            conv(cin=hid, cout=D, inp=V(P), out=V(nxt), out2=V(pre), act=L.ACT_MISH, res1=V(xs), wpacked=st[b]["w2"], bias=blk.fc2.bias.data,
                 sample_scale=keep_masks.get(b + 1), **geo)
            sv["blocks"].append(dict(x=xs, xn=xn, F=Fb, Cq=Cq, P=P, pre=pre, mean=mean, rstd=rstd))
            xs = nxt
            xl = nxt
        t1c, t2c, t3c = self.gblocks[nb + 1], self.gblocks[nb + 3], self.gblocks[nb + 5]
        t1a, p1, t2a, p2 = Z(B, h, w, 2 * D), Z(B, h, w, 2 * D), Z(B, h, w, D), Z(B, h, w, D)
        conv(cin=D, cout=2 * D, inp=V(xl), out=V(t1a), out2=V(p1), act=L.ACT_MISH, wpacked=st["t1"], bias=t1c.bias.data, **geo)
        conv(cin=2 * D, cout=D, inp=V(t1a), out=V(t2a), out2=V(p2), act=L.ACT_MISH, wpacked=st["t2"], bias=t2c.bias.data, **geo)
        s1c, s2c, s11 = self.shortcut.block[0], self.shortcut.block[2], self.shortcut.conv11
        s1a, q1, s2a, q2, sc = Z(B, h, w, D), Z(B, h, w, D), Z(B, h, w, D, dt=f32), Z(B, h, w, D), Z(B, h, w, D, dt=f32)
        conv(cin=8, cout=D, inp=V(x8), out=V(s1a), out2=V(q1), act=L.ACT_MISH, wpacked=st["s1"], bias=s1c.bias.data, **geo)
        conv(cin=D, cout=D, inp=V(s1a), out=V(s2a), out2=V(q2), act=L.ACT_MISH, wpacked=st["s2"], bias=s2c.bias.data, **geo)
        conv(cin=8, cout=D, inp=V(x8), out=V(sc), res1=V(s2a), wpacked=st["s11"], bias=s11.bias.data, ksize=1, **geo)
        feat = Z(B, h, w, D)
        conv(cin=D, cout=D, inp=V(t2a), out=V(feat), res1=V(sc), wpacked=st["t3"], bias=t3c.bias.data, ksize=1, **geo)
        up = self.upsampler[0]
        cup = _p8(3 * r * r)
        u = Z(B, h, w, cup, dt=f32)
        conv(cin=D, cout=3 * r * r, inp=V(feat), out=V(u), wpacked=st["up"], bias=up.bias.data, **geo)
        o4 = Z(B, h * r, w * r, 4, dt=f32)
        ops.pixel_shuffle(V(u), V(o4), B, h, w, 3, r)
        out = torch.empty(B, 3, h * r, w * r, dtype=f32, device=dev)
        ops.nhwc_to_nchw(V(o4), out)
        sv.update(xl=xl, t1a=t1a, p1=p1, t2a=t2a, p2=p2, s1a=s1a, q1=q1, q2=q2, feat=feat)
        self._saved = sv if train else None
        return out

    # ------------------------------------------------------------------ backward ------------------------------------------------
    def backward_hip(self, grad_out: torch.Tensor) -> None:
        if self._saved is None:
            raise RuntimeError("backward_hip needs a preceding training-mode forward_hip")
        sv = self._saved
        self._saved = None
        self._ensure_arena()
        B, h, w = sv["B"], sv["h"], sv["w"]
        dev, cdt, cd = grad_out.device, self.compute_dtype, self._cd()
        st, V, f32 = self._slabs, ops.view, torch.float32
        Z = self._pool("bwd", dev, cdt)
        D, hid, cc, nb, r, k = self.D, self.hid, self.cc, self.nb, self.scale, self.k
        T, H, W = B * h * w, h * r, w * r
        cup = _p8(3 * r * r)
        conv = ops.conv3x3
        geo = dict(n=B, h=h, w=w, compute=cd)
        need = 0
        for (ci, co, ks) in [(D, 2 * hid, 3), (hid, D, 3), (D, 2 * D, 3), (2 * D, D, 3), (D, D, 1), (D, D, 3), (3, D, 3), (3, D, 1), (D, 3 * r * r, 3)]:
            need = max(need, ops.wgrad_workspace_bytes(B, h, w, ci, co, ks))
        need = max(need, ops.dwconv_wgrad_workspace_bytes(B, h, w, cc, k), ops.layernorm_bwd_workspace_bytes(D))
        wg = Z(need // 4 + 16, dt=f32)

        def wgrad(c, inp, gy, ks, cin):
            ops.wgrad3x3(n=B, h=h, w=w, cin=cin, cout=c.weight.shape[0], compute=cd, inp=inp, gy=gy, dw=c.weight.grad, db=c.bias.grad, workspace=wg,
                         accumulate=True, ksize=ks)

        g8 = Z(B, H, W, 8)
        ops.nchw_to_nhwc(grad_out.contiguous().float(), g8)
        gu = Z(B, h, w, cup)
        ops.pixel_shuffle(V(g8), V(gu), B, h, w, 3, r, inverse=True)
        up = self.upsampler[0]
        wgrad(up, V(sv["feat"]), V(gu), 3, D)
        g_feat = Z(B, h, w, D)
        conv(cin=cup, cout=D, inp=V(gu), out=V(g_feat), wpacked=st["upt"], **geo)
        # ---- shortcut: sc = mish(s2(mish(s1(x)))) + conv11(x)
        s1c, s2c, s11 = self.shortcut.block[0], self.shortcut.block[2], self.shortcut.conv11
        wgrad(s11, V(sv["x8"]), V(g_feat), 1, 3)
        g_s2, g_s1 = Z(B, h, w, D), Z(B, h, w, D)
        ops.act_grad(L.ACT_MISH, V(g_feat), V(sv["q2"]), V(g_s2), T)
        wgrad(s2c, V(sv["s1a"]), V(g_s2), 3, D)
        conv(cin=D, cout=D, inp=V(g_s2), out=V(g_s1), wpacked=st["s2t"], mask=V(sv["q1"]), mask_mode=3, **geo)
        wgrad(s1c, V(sv["x8"]), V(g_s1), 3, 3)
        # ---- tail of the body: t3(mish(t2(mish(t1(x_last)))))
        t1c, t2c, t3c = self.gblocks[nb + 1], self.gblocks[nb + 3], self.gblocks[nb + 5]
        wgrad(t3c, V(sv["t2a"]), V(g_feat), 1, D)
        g_t2, g_t1 = Z(B, h, w, D), Z(B, h, w, 2 * D)
        conv(cin=D, cout=D, inp=V(g_feat), out=V(g_t2), wpacked=st["t3t"], ksize=1, mask=V(sv["p2"]), mask_mode=3, **geo)
        wgrad(t2c, V(sv["t1a"]), V(g_t2), 3, 2 * D)
        conv(cin=D, cout=2 * D, inp=V(g_t2), out=V(g_t1), wpacked=st["t2t"], mask=V(sv["p1"]), mask_mode=3, **geo)
        wgrad(t1c, V(sv["xl"]), V(g_t1), 3, D)
        G = [Z(B, h, w, D, dt=f32) for _ in range(2)]        # d/d(block output), ping-pong
        conv(cin=2 * D, cout=D, inp=V(g_t1), out=V(G[0]), wpacked=st["t1t"], **geo)
        cur = 0
        gpre, dP, dF, dCq, dxn, g0 = Z(B, h, w, D), Z(B, h, w, hid), Z(B, h, w, 2 * hid), Z(B, h, w, cc), Z(B, h, w, D, dt=f32), Z(B, h, w, D)
        keepm = sv["keep"]
        blocks = self._blocks()
        for b in range(nb - 1, -1, -1):
            blk, s = blocks[b], sv["blocks"][b]
            ops.act_grad(L.ACT_MISH, V(G[cur]), V(s["pre"]), V(gpre), T, sample_scale=keepm.get(b + 1), pix_per_sample=h * w)
            wgrad(blk.fc2, V(s["P"]), V(gpre), 3, hid)
            conv(cin=D, cout=hid, inp=V(gpre), out=V(dP), wpacked=st[b]["w2t"], **geo)
            Fb = s["F"]
            ops.gate_mish_bwd(V(dP), V(Fb, 0, hid), V(Fb, hid, hid - cc), V(s["Cq"]), V(dF, 0, hid), V(dF, hid, hid - cc), V(dCq), T)
            ops.dwconv(V(dCq), V(dF, 2 * hid - cc, cc), blk.conv.weight.data, None, B, h, w, cc, flip=True)
            ops.dwconv_wgrad(V(Fb, 2 * hid - cc, cc), V(dCq), blk.conv.weight.grad, blk.conv.bias.grad, B, h, w, cc, wg)
            wgrad(blk.fc1, V(s["xn"]), V(dF), 3, D)
            conv(cin=2 * hid, cout=D, inp=V(dF), out=V(dxn), wpacked=st[b]["w1t"], **geo)
            dx = V(g0) if b == 0 else V(G[cur ^ 1])          # d/d(head conv output) feeds a weight gradient only: compute dtype
            ops.layernorm_bwd(V(dxn), V(s["x"]), T, D, blk.norm.weight.data, s["mean"], s["rstd"], V(G[cur]), dx, blk.norm.weight.grad, blk.norm.bias.grad,
                              wg, accumulate=True)
            cur ^= 1
        wgrad(self.gblocks[0], V(sv["x8"]), V(g0), 3, 3)

    # ------------------------------------------------------------------ nn.Module surface ------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            if self._trigger is None or self._trigger.device != x.device:
                self._trigger = torch.zeros(1, device=x.device, requires_grad=True)
            return _MosrFn.apply(x, self._trigger, self)
        return self.forward_hip(x, train=False)


class _MosrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, trigger, net):
        ctx.net = net
        return net.forward_hip(x, train=True)

    @staticmethod
    def backward(ctx, gy):
        ctx.net.backward_hip(gy)
        return None, torch.zeros(1, device=gy.device), None


@ARCH_REGISTRY.register()
def mosr(scale: int = 4, in_ch: int = 3, out_ch: int = 3, n_block: int = 24, dim: int = 64, upsampler: upsampler_type = "pixelshuffle",
         drop_path: float = 0.0, kernel_size: int = 7, expansion_ratio: float = 1.5, conv_ratio: float = 1.0) -> MoSR:
    return MoSR(upscale=scale, in_ch=in_ch, out_ch=out_ch, n_block=n_block, dim=dim, upsampler=get_upsampler(upsampler), drop_path=drop_path,
                kernel_size=kernel_size, expansion_ratio=expansion_ratio, conv_ratio=conv_ratio)


@ARCH_REGISTRY.register()
def mosr_t(scale: int = 4, in_ch: int = 3, out_ch: int = 3, n_block: int = 5, dim: int = 48, upsampler: upsampler_type = "pixelshuffle",
           drop_path: float = 0.0, kernel_size: int = 7, expansion_ratio: float = 1.5, conv_ratio: float = 1.0) -> MoSR:
    return MoSR(upscale=scale, in_ch=in_ch, out_ch=out_ch, n_block=n_block, dim=dim, upsampler=get_upsampler(upsampler), drop_path=drop_path,
                kernel_size=kernel_size, expansion_ratio=expansion_ratio, conv_ratio=conv_ratio)
