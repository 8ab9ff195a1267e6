"""SwinIR (`swinir_s`, `swinir_m`) on the HIP kernels (gfx950).

Registered under the reference's names with the reference's factory signatures
(/root/reference/traiNNer/archs/swinir_arch.py:64-163); arithmetic follows the in-tree twin of spandrel's SwinIR
(/root/reference/traiNNer/archs/drct_arch.py:52-223,415-575,624-725); state-dict keys are the official SwinIR ones (incl. the
`relative_position_index` / `attn_mask` buffers) so checkpoints interchange.  Checksums: 929,628 / 11,900,199 parameters
(docs/source/benchmarks.md:71,75).

Execution: tokens ARE the pixels of NHWC buffers, so PatchEmbed / PatchUnEmbed, torch.roll, window_partition and
window_reverse are index arithmetic inside the kernels (the reference makes 4 full copies per shifted block);
  * nn.Linear (qkv / proj / fc1 / fc2: 71 % of the FLOPs)  -> the MFMA conv kernel with ksize 1 on an (n, h*w/32, 32) token grid,
    with bias, exact GELU, residual add into the fp32 stream and the DropPath keep mask fused into its epilogue;
  * 3x3 convs (22 %)                                         -> the MFMA conv kernel (channel strides padded 180 -> 192);
  * LayerNorm, window attention, PixelShuffle                -> token.hip.
The residual stream stays fp32; MFMA operands are the compute dtype.  Backward is hand-scheduled; activations are saved, not
recomputed (6.5 KB per token and block).  Scope: upsampler `pixelshuffle` / `pixelshuffledirect`, `1conv`, window 8, H and W
multiples of 8 (training tiles); other variants raise NotImplementedError.
"""
from __future__ import annotations

import math
from collections.abc import Sequence

import torch
from torch import nn

from .. import _lib as L
from .. import ops
from ..utils.registry import SPANDREL_REGISTRY
from ._arena import FlatParamModule

RGB_MEAN = (0.4488, 0.4371, 0.4040)


def _p32(c: int) -> int:
    return (c + 31) // 32 * 32


class _Lin(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.weight = nn.Parameter(torch.randn(cout, cin) * 0.02)       # official init: trunc_normal_(std=.02), zero bias
        self.bias = nn.Parameter(torch.zeros(cout))


class _LN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class _Conv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout = cin, cout
        ref = nn.Conv2d(cin, cout, 3, 1, 1)
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone())


class _Seq(nn.Module):
    def __init__(self, items: dict):
        super().__init__()
        for k, v in items.items():
            self.add_module(str(k), v)


class _Attn(nn.Module):
    def __init__(self, c, nh, ws):
        super().__init__()
        self.relative_position_bias_table = nn.Parameter(torch.randn((2 * ws - 1) ** 2, nh) * 0.02)
        coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws - 1
        rel[:, :, 1] += ws - 1
        rel[:, :, 0] *= 2 * ws - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.qkv = _Lin(c, 3 * c)
        self.proj = _Lin(c, c)


class _Mlp(nn.Module):
    def __init__(self, c, hid):
        super().__init__()
        self.fc1, self.fc2 = _Lin(c, hid), _Lin(hid, c)


def _shift_mask(res: int, ws: int, shift: int) -> torch.Tensor:
    img = torch.zeros(1, res, res, 1)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = img.view(1, res // ws, ws, res // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


class _Block(nn.Module):
    def __init__(self, c, nh, ws, shift, hid, img_size):
        super().__init__()
        self.shift = shift
        self.norm1 = _LN(c)
        self.attn = _Attn(c, nh, ws)
        self.norm2 = _LN(c)
        self.mlp = _Mlp(c, hid)
        if shift > 0 and img_size > ws:     # the official module keeps the mask of its construction-time resolution as a buffer
            self.register_buffer("attn_mask", _shift_mask(img_size, ws, shift))
        else:
            self.attn_mask = None


class _Group(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.blocks = _Seq(dict(enumerate(blocks)))


class _RSTB(nn.Module):
    def __init__(self, c, nh, ws, depth, hid, img_size):
        super().__init__()
        self.residual_group = _Group([_Block(c, nh, ws, 0 if j % 2 == 0 else ws // 2, hid, img_size) for j in range(depth)])
        self.conv = _Conv(c, c)


class _PE(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = _LN(c)


class SwinIR(FlatParamModule):
    graph_safe_drop_path = True       # `_keep_masks` draws on the device into persistent buffers: a hipGraph of the step re-draws on every replay
    prefer_hip_graphs = True          # ~1,900 launches of ~15 us per training step at 64 x 64 tiles: the trainer replays the step as two hipGraphs by default

    def __init__(self, *, upscale=4, img_size=64, patch_size=1, in_chans=3, embed_dim=180, depths: Sequence[int] = (6,) * 6,
                 num_heads: Sequence[int] = (6,) * 6, window_size=8, mlp_ratio=2.0, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.1, ape=False, patch_norm=True, use_checkpoint=False, img_range=1.0, upsampler="pixelshuffle",
                 resi_connection="1conv", start_unshuffle=1) -> None:
        super().__init__()
        bad = []
        if window_size != 8: bad.append("window_size != 8")
        if patch_size != 1 or ape or not patch_norm or not qkv_bias or qk_scale is not None: bad.append("patch_size/ape/patch_norm/qkv_bias/qk_scale")
        if drop_rate or attn_drop_rate: bad.append("dropout")
        if upsampler not in ("pixelshuffle", "pixelshuffledirect"): bad.append(f"upsampler {upsampler!r}")
        if resi_connection != "1conv": bad.append(f"resi_connection {resi_connection!r}")
        if start_unshuffle != 1 or in_chans != 3: bad.append("start_unshuffle / in_chans")
        if upscale & (upscale - 1): bad.append("upscale not a power of two")
        if any(embed_dim % nh or embed_dim // nh > 32 for nh in num_heads): bad.append("head_dim > 32")
        if bad:
            raise NotImplementedError("SwinIR variant not on the HIP path yet: " + ", ".join(bad))
        C = embed_dim
        self.C, self.ws, self.scale, self.img_range, self.upsampler = C, window_size, upscale, float(img_range), upsampler
        self.depths, self.heads, self.hid, self.nf = list(depths), list(num_heads), int(C * mlp_ratio), 64
        self.drop_path_rate = float(drop_path_rate)
        self.hyperparameters = dict(upscale=upscale, img_size=img_size, embed_dim=embed_dim, depths=list(depths), num_heads=list(num_heads),
                                    window_size=window_size, mlp_ratio=mlp_ratio, upsampler=upsampler, resi_connection=resi_connection,
                                    img_range=img_range)
        self.conv_first = _Conv(in_chans, C)
        self.patch_embed = _PE(C)
        self.layers = _Seq({i: _RSTB(C, nh, window_size, d, self.hid, img_size) for i, (d, nh) in enumerate(zip(depths, num_heads))})
        self.norm = _LN(C)
        self.conv_after_body = _Conv(C, C)
        if upsampler == "pixelshuffle":
            self.conv_before_upsample = _Seq({0: _Conv(C, self.nf)})
            self.upsample = _Seq({2 * k: _Conv(self.nf, 4 * self.nf) for k in range(int(math.log2(upscale)))})
            self.conv_last = _Conv(self.nf, in_chans)
        else:
            self.upsample = _Seq({0: _Conv(C, upscale * upscale * in_chans)})
        self.compute_dtype = torch.bfloat16
        self._init_arena_state()

    # ------------------------------------------------------------------ slabs ------------------------------------------------
    def _cd(self):
        return ops.dt_code(self.compute_dtype)

    def _blocks(self):
        for i, (d, nh) in enumerate(zip(self.depths, self.heads)):
            layer = getattr(self.layers, str(i))
            for j in range(d):
                yield i, j, nh, getattr(layer.residual_group.blocks, str(j))

    def repack(self):
        self._ensure_arena()
        dev, cd = self._arena.device, self._cd()
        key = (cd, str(dev), self._arena.data_ptr())
        st = self.__dict__.setdefault("_slabs", {})      # transient: an EMA deep copy rebuilds its own
        if st.get("key") != key:
            st.clear()
            st["key"] = key
            plan = ops.PackPlan(dev)

            def z(nbytes):
                return torch.zeros(nbytes, dtype=torch.uint8, device=dev)

            def lin(m: _Lin):
                kp, npad = _p32(m.cin), _p32(m.cout)
                f = z(ops.packed_bytes(kp, m.cout, cd, 1))
                plan.add(m.weight.data, f, kp, m.cout, cd)
                b = z(ops.packed_bytes(npad, m.cin, cd, 1))
                plan.add(m.weight.data, b, npad, m.cin, cd, transpose=True)
                return f, b

            def conv(m: _Conv, cin_pad=None, cout_pad_b=None):
                ci = cin_pad or _p32(m.cin)
                f = z(ops.packed_bytes(ci, m.cout, cd))
                plan.add(m.weight.data, f, ci, m.cout, cd)
                cb = cout_pad_b or _p32(m.cout)
                b = z(ops.packed_bytes(cb, m.cin, cd))
                plan.add(m.weight.data, b, cb, m.cin, cd, transpose=True)
                return f, b

            st["conv_first"] = conv(self.conv_first, cin_pad=8)
            for i, j, _, blk in self._blocks():
                st[("qkv", i, j)], st[("proj", i, j)] = lin(blk.attn.qkv), lin(blk.attn.proj)
                st[("fc1", i, j)], st[("fc2", i, j)] = lin(blk.mlp.fc1), lin(blk.mlp.fc2)
            for i in range(len(self.depths)):
                st[("lconv", i)] = conv(getattr(self.layers, str(i)).conv)
            st["cab"] = conv(self.conv_after_body)
            if self.upsampler == "pixelshuffle":
                st["cbu"] = conv(getattr(self.conv_before_upsample, "0"))
                for k in range(int(math.log2(self.scale))):
                    st[("up", k)] = conv(getattr(self.upsample, str(2 * k)))
                st["last"] = conv(self.conv_last, cout_pad_b=8)
            else:
                st["updirect"] = conv(getattr(self.upsample, "0"))
            st["plan"] = plan.finalize()
        st["plan"].run()

    # ------------------------------------------------------------------ forward ------------------------------------------------
    def _keep_masks(self, b, dev):
        """DropPath (drct_arch.py:572-573): per block and sample, keep/(1-p) with p rising linearly over the blocks.

        ONE draw for all blocks into a persistent (blocks x batch) buffer (three launches per step instead of four per block), rows handed out as
        views: the addresses stay fixed from step to step, so a captured hipGraph of the step reads this step's masks (archs/_arena.py)."""
        nblk = sum(self.depths)
        st = self.__dict__.get("_keep_state")
        if st is None or st["key"] != (b, str(dev), nblk, self.drop_path_rate):
            rates = torch.linspace(0, self.drop_path_rate, nblk)
            st = self._keep_state = dict(key=(b, str(dev), nblk, self.drop_path_rate), rates=rates.tolist(), p=rates.to(dev).view(nblk, 1),
                                         inv=(1.0 / (1.0 - rates)).to(dev).view(nblk, 1), u=torch.empty(nblk, b, device=dev), keep=torch.empty(nblk, b, device=dev))
        st["u"].uniform_()                                   # same distribution as the reference's per-block rand(); one stream instead of 36
        torch.ge(st["u"], st["p"], out=st.setdefault("ge", torch.empty(nblk, b, dtype=torch.bool, device=dev)))
        torch.mul(st["ge"], st["inv"], out=st["keep"])
        out, k = {}, 0
        for i, j, _, _ in self._blocks():
            if st["rates"][k] > 0:
                out[(i, j)] = st["keep"][k]
            k += 1
        return out

    def forward_hip(self, x: torch.Tensor, train: bool | None = None, keep_masks: dict | None = None) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("SwinIR (HIP) runs on the GPU only; there is no CPU path in this package")
        L.lib()
        if train is None:
            train = torch.is_grad_enabled() and self.training
        x = x.contiguous().float()
        B, cin, h, w = x.shape
        if cin != 3 or h % self.ws or w % self.ws:
            raise NotImplementedError(f"SwinIR (HIP) needs 3 channels and H, W multiples of {self.ws} (reflect padding of odd sizes is not built)")
        self.repack()
        dev, cdt, cd = x.device, self.compute_dtype, self._cd()
        C, CP, HP, C3P = self.C, _p32(self.C), _p32(self.hid), _p32(3 * self.C)
        T = B * h * w
        st = self._slabs
        V = ops.view
        f32 = torch.float32
        Z = self._pool("fwd", dev, cdt)       # zeroed once: kernels write real channels only, so the channel padding keeps reading as 0
        if train and keep_masks is None and self.drop_path_rate > 0:
            keep_masks = self._keep_masks(B, dev)
        keep_masks = keep_masks or {}
        tg = dict(n=B, h=h * w // 32, w=32, compute=cd, ksize=1)           # token-grid geometry of the Linear layers
        ig = dict(n=B, h=h, w=w, compute=cd)
        sv = dict(B=B, h=h, w=w, blocks={}, keep=keep_masks)
        x8 = Z(B, h, w, 8)
        ops.nchw_to_nhwc(x, x8, RGB_MEAN, [1.0 / self.img_range] * 3)
        fea = Z(B, h, w, CP, dt=f32)
        ops.conv3x3(cin=8, cout=C, inp=V(x8), out=V(fea), wpacked=st["conv_first"][0], bias=self.conv_first.bias.data, **ig)
        xs = Z(B, h, w, CP, dt=f32)
        pe_mean, pe_rstd = Z(T, dt=f32), Z(T, dt=f32)
        pe = self.patch_embed.norm
        ops.layernorm(V(fea, 0, C), V(xs), T, C, pe.weight.data, pe.bias.data, 1e-5, pe_mean, pe_rstd)
        sv.update(x8=x8, fea=fea, pe=(pe_mean, pe_rstd))
        cur_layer_in = xs
        x_in = xs
        xcd = None
        for i, j, nh, blk in self._blocks():
            if j == 0:
                cur_layer_in = x_in
            keep = keep_masks.get((i, j))
            s = {}
            ln1 = Z(B, h, w, CP)
            s["m1"], s["r1"] = Z(T, dt=f32), Z(T, dt=f32)
            ops.layernorm(V(x_in, 0, C), V(ln1), T, C, blk.norm1.weight.data, blk.norm1.bias.data, 1e-5, s["m1"], s["r1"])
            qkv = Z(B, h, w, C3P)                    # compute dtype: bf16 q, k, v are the MFMA operands of attn_mfma.hip
            ops.conv3x3(cin=CP, cout=3 * C, inp=V(ln1), out=V(qkv), wpacked=st[("qkv", i, j)][0], bias=blk.attn.qkv.bias.data, **tg)
            ao = Z(B, h, w, CP)
            ops.window_attention(V(qkv), V(ao), blk.attn.relative_position_bias_table.data, b=B, h=h, w=w, c=C, nh=nh, shift=blk.shift)
            x_mid = Z(B, h, w, CP, dt=f32)
            ops.conv3x3(cin=CP, cout=C, inp=V(ao), out=V(x_mid), wpacked=st[("proj", i, j)][0], bias=blk.attn.proj.bias.data, res1=V(x_in),
                        sample_scale=keep, **tg)
            ln2 = Z(B, h, w, CP)
            s["m2"], s["r2"] = Z(T, dt=f32), Z(T, dt=f32)
            ops.layernorm(V(x_mid, 0, C), V(ln2), T, C, blk.norm2.weight.data, blk.norm2.bias.data, 1e-5, s["m2"], s["r2"])
            post, pre = Z(B, h, w, HP), Z(B, h, w, HP)
            ops.conv3x3(cin=CP, cout=self.hid, inp=V(ln2), out=V(post), out2=V(pre), wpacked=st[("fc1", i, j)][0], bias=blk.mlp.fc1.bias.data,
                        act=L.ACT_GELU, **tg)
            x_out = Z(B, h, w, CP, dt=f32)
            last_in_layer = j == self.depths[i] - 1
            kw = {}
            if last_in_layer:
                xcd = Z(B, h, w, CP)
                kw["out2"] = V(xcd)
            ops.conv3x3(cin=HP, cout=C, inp=V(post), out=V(x_out), wpacked=st[("fc2", i, j)][0], bias=blk.mlp.fc2.bias.data, res1=V(x_mid),
                        sample_scale=keep, **kw, **tg)
            s.update(x_in=x_in, ln1=ln1, qkv=qkv, ao=ao, x_mid=x_mid, ln2=ln2, post=post, pre=pre)
            sv["blocks"][(i, j)] = s
            x_in = x_out
            if last_in_layer:
                layer = getattr(self.layers, str(i))
                nxt = Z(B, h, w, CP, dt=f32)
                ops.conv3x3(cin=CP, cout=C, inp=V(xcd), out=V(nxt), wpacked=st[("lconv", i)][0], bias=layer.conv.bias.data, res1=V(cur_layer_in), **ig)
                sv[("lconv_in", i)] = xcd
                x_in = nxt
        lnf = Z(B, h, w, CP)
        fm, fr = Z(T, dt=f32), Z(T, dt=f32)
        ops.layernorm(V(x_in, 0, C), V(lnf), T, C, self.norm.weight.data, self.norm.bias.data, 1e-5, fm, fr)
        body = Z(B, h, w, CP)
        ops.conv3x3(cin=CP, cout=C, inp=V(lnf), out=V(body), wpacked=st["cab"][0], bias=self.conv_after_body.bias.data, res1=V(fea), **ig)
        sv.update(x_final=x_in, fnorm=(fm, fr), lnf=lnf, body=body)
        r = self.scale
        H, W = h * r, w * r
        out = torch.empty(B, 3, H, W, dtype=f32, device=dev)
        if self.upsampler == "pixelshuffle":
            cbu = getattr(self.conv_before_upsample, "0")
            u = Z(B, h, w, self.nf)
            ops.conv3x3(cin=CP, cout=self.nf, inp=V(body), out=V(u), wpacked=st["cbu"][0], bias=cbu.bias.data, act=L.ACT_LRELU, slope=0.01, **ig)
            us = [u]
            hh, ww = h, w
            for k in range(int(math.log2(r))):
                m = getattr(self.upsample, str(2 * k))
                c4 = Z(B, hh, ww, 4 * self.nf)
                ops.conv3x3(n=B, h=hh, w=ww, compute=cd, cin=self.nf, cout=4 * self.nf, inp=V(us[-1]), out=V(c4), wpacked=st[("up", k)][0],
                            bias=m.bias.data)
                nu = Z(B, 2 * hh, 2 * ww, self.nf)
                ops.pixel_shuffle(V(c4), V(nu), B, hh, ww, self.nf, 2)
                us.append(nu)
                hh, ww = 2 * hh, 2 * ww
            o4 = Z(B, H, W, 4, dt=f32)
            ops.conv3x3(n=B, h=H, w=W, compute=cd, cin=self.nf, cout=3, inp=V(us[-1]), out=V(o4), wpacked=st["last"][0], bias=self.conv_last.bias.data)
            sv.update(us=us)
        else:
            m = getattr(self.upsample, "0")
            cpo = _p32(r * r * 3)
            c48 = Z(B, h, w, cpo, dt=f32)
            ops.conv3x3(cin=CP, cout=r * r * 3, inp=V(body), out=V(c48), wpacked=st["updirect"][0], bias=m.bias.data, **ig)
            o4 = Z(B, H, W, 4, dt=f32)
            ops.pixel_shuffle(V(c48), V(o4), B, h, w, 3, r)
        ops.nhwc_to_nchw_affine(V(o4), out, [1.0 / self.img_range] * 3, RGB_MEAN)
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
        C, CP, HP, C3P = self.C, _p32(self.C), _p32(self.hid), _p32(3 * self.C)
        T = B * h * w
        st = self._slabs
        V = ops.view
        f32 = torch.float32
        Z = self._pool("bwd", dev, cdt)
        tg = dict(n=B, h=h * w // 32, w=32, compute=cd, ksize=1)
        ig = dict(n=B, h=h, w=w, compute=cd)
        r = self.scale
        H, W = h * r, w * r
        need = ops.layernorm_bwd_workspace_bytes(C)
        for (hh, ww, ci, co, ks) in [(h * w // 32, 32, CP, 3 * C, 1), (h * w // 32, 32, HP, C, 1), (h * w // 32, 32, CP, self.hid, 1), (h, w, CP, C, 3),
                                     (h, w, 3, C, 3), (H, W, self.nf, 3, 3), (H // 2, W // 2, self.nf, 4 * self.nf, 3), (h, w, CP, r * r * 3, 3)]:
            need = max(need, ops.wgrad_workspace_bytes(B, hh, ww, ci, co, ks))
        wg = Z(need // 4 + 16, dt=f32)
        aws = Z(ops.window_attention_bwd_workspace_bytes(B, h, w, max(self.heads)) // 4, dt=f32)

        def wgrad_conv(m, inp, gy, hh, ww, ks=3):
            ops.wgrad3x3(n=B, h=hh, w=ww, cin=m.cin, cout=m.cout, compute=cd, inp=inp, gy=gy, dw=m.weight.grad, db=m.bias.grad, workspace=wg,
                         accumulate=True, ksize=ks)

        def wgrad_lin(m, inp, gy):
            wgrad_conv(m, inp, gy, h * w // 32, 32, ks=1)

        g8 = Z(B, H, W, 8)
        ops.nchw_to_nhwc(grad_out.contiguous().float(), g8, (0.0, 0.0, 0.0), [self.img_range] * 3)
        g_body32, g_body = Z(B, h, w, CP, dt=f32), Z(B, h, w, CP)
        if self.upsampler == "pixelshuffle":
            us = sv["us"]
            wgrad_conv(self.conv_last, V(us[-1]), V(g8), H, W)
            gu = Z(B, H, W, self.nf)
            ops.conv3x3(n=B, h=H, w=W, compute=cd, cin=8, cout=self.nf, inp=V(g8), out=V(gu), wpacked=st["last"][1])
            hh, ww = H, W
            for k in range(int(math.log2(r)) - 1, -1, -1):
                m = getattr(self.upsample, str(2 * k))
                hh, ww = hh // 2, ww // 2
                gc4 = Z(B, hh, ww, 4 * self.nf)
                ops.pixel_shuffle(V(gu), V(gc4), B, hh, ww, self.nf, 2, inverse=True)
                wgrad_conv(m, V(us[k]), V(gc4), hh, ww)
                gu = Z(B, hh, ww, self.nf)
                kw = dict(mask=V(us[0]), mask_slope=0.01) if k == 0 else {}
                ops.conv3x3(n=B, h=hh, w=ww, compute=cd, cin=4 * self.nf, cout=self.nf, inp=V(gc4), out=V(gu), wpacked=st[("up", k)][1], **kw)
            cbu = getattr(self.conv_before_upsample, "0")
            wgrad_conv(cbu, V(sv["body"]), V(gu), h, w)
            ops.conv3x3(cin=self.nf, cout=C, inp=V(gu), out=V(g_body32), out2=V(g_body), wpacked=st["cbu"][1], **ig)
        else:
            m = getattr(self.upsample, "0")
            cpo = _p32(r * r * 3)
            gc = Z(B, h, w, cpo)
            ops.pixel_shuffle(V(g8), V(gc), B, h, w, 3, r, inverse=True)
            wgrad_conv(m, V(sv["body"]), V(gc), h, w)
            ops.conv3x3(cin=cpo, cout=C, inp=V(gc), out=V(g_body32), out2=V(g_body), wpacked=st["updirect"][1], **ig)
        wgrad_conv(self.conv_after_body, V(sv["lnf"]), V(g_body), h, w)
        g_lnf = Z(B, h, w, CP, dt=f32)
        ops.conv3x3(cin=CP, cout=C, inp=V(g_body), out=V(g_lnf), wpacked=st["cab"][1], **ig)
        free = [Z(B, h, w, CP, dt=f32) for _ in range(5)]      # fp32 gradient-stream buffers, recycled block after block (stream order)
        gx = free.pop()
        fm, fr = sv["fnorm"]
        keepm = sv["keep"]
        blocks = list(self._blocks())
        gcd, g_pre, dqkv = Z(B, h, w, CP), Z(B, h, w, HP), Z(B, h, w, C3P)
        # Every LayerNorm backward also writes the compute-dtype copy of its output that the NEXT GEMMs read (`dx2`, scaled by that branch's DropPath
        # keep mask): the separate scale-and-convert pass over the gradient stream (73 launches per step) is gone.  `copied` says that `gcd` already
        # holds the copy the next consumer needs.
        ops.layernorm_bwd(V(g_lnf, 0, C), V(sv["x_final"], 0, C), T, C, self.norm.weight.data, fm, fr, ops.NULL_VIEW, V(gx, 0, C), self.norm.weight.grad,
                          self.norm.bias.grad, wg, dx2=V(gcd, 0, C))          # the last block closes an RSTB: its tail conv reads the unscaled copy
        copied = True
        g_ln, g_ao = Z(B, h, w, CP, dt=f32), Z(B, h, w, CP)
        g_layer_out = None
        for i, j, nh, blk in reversed(blocks):
            s = sv["blocks"][(i, j)]
            keep = keepm.get((i, j))
            if j == self.depths[i] - 1:
                # RSTB tail: X_next = conv(x_out) + layer_in
                layer = getattr(self.layers, str(i))
                g_layer_out = gx
                if not copied:
                    ops.axpby(V(gx), 1.0, ops.NULL_VIEW, 0.0, V(gcd), T)
                wgrad_conv(layer.conv, V(sv[("lconv_in", i)]), V(gcd), h, w)
                gx = free.pop()
                ops.conv3x3(cin=CP, cout=C, inp=V(gcd), out=V(gx), wpacked=st[("lconv", i)][1], **ig)
                copied = False
            # ---- MLP branch: x_out = x_mid + keep * fc2(gelu(fc1(LN2(x_mid))))
            if not copied:
                ops.axpby(V(gx), 1.0, ops.NULL_VIEW, 0.0, V(gcd), T, sample_scale=keep, pix_per_sample=h * w)
            wgrad_lin(blk.mlp.fc2, V(s["post"]), V(gcd))
            ops.conv3x3(cin=CP, cout=self.hid, inp=V(gcd), out=V(g_pre), wpacked=st[("fc2", i, j)][1], mask=V(s["pre"]), mask_mode=1, **tg)
            wgrad_lin(blk.mlp.fc1, V(s["ln2"]), V(g_pre))
            ops.conv3x3(cin=HP, cout=C, inp=V(g_pre), out=V(g_ln), wpacked=st[("fc1", i, j)][1], **tg)
            g_mid = free.pop()
            ops.layernorm_bwd(V(g_ln, 0, C), V(s["x_mid"], 0, C), T, C, blk.norm2.weight.data, s["m2"], s["r2"], V(gx, 0, C), V(g_mid, 0, C),
                              blk.norm2.weight.grad, blk.norm2.bias.grad, wg, dx2=V(gcd, 0, C), sample_scale=keep, pix_per_sample=h * w)
            free.append(gx)
            # ---- attention branch: x_mid = x_in + keep * proj(attn(qkv(LN1(x_in)))) ; gcd = keep * g_mid came out of the LayerNorm backward above
            wgrad_lin(blk.attn.proj, V(s["ao"]), V(gcd))
            ops.conv3x3(cin=CP, cout=C, inp=V(gcd), out=V(g_ao), wpacked=st[("proj", i, j)][1], **tg)
            ops.window_attention_bwd(V(s["qkv"]), V(g_ao), V(dqkv), blk.attn.relative_position_bias_table.data,
                                     blk.attn.relative_position_bias_table.grad, aws, b=B, h=h, w=w, c=C, nh=nh, shift=blk.shift)
            wgrad_lin(blk.attn.qkv, V(s["ln1"]), V(dqkv))
            ops.conv3x3(cin=C3P, cout=C, inp=V(dqkv), out=V(g_ln), wpacked=st[("qkv", i, j)][1], **tg)
            res = g_mid
            if j == 0:
                # the layer input also feeds the RSTB skip
                tmp = free.pop()
                ops.axpby(V(g_mid), 1.0, V(g_layer_out), 1.0, V(tmp), T)
                free.append(g_layer_out)
                res = tmp
            g_in = free.pop()
            # what reads g_in next: the MLP branch of block (i, j - 1) -- its keep mask -- or, below an RSTB boundary, the tail conv of layer i - 1 (unscaled)
            nxt = dict(dx2=V(gcd, 0, C), sample_scale=keepm.get((i, j - 1)), pix_per_sample=h * w) if j > 0 else (dict(dx2=V(gcd, 0, C)) if i > 0 else {})
            ops.layernorm_bwd(V(g_ln, 0, C), V(s["x_in"], 0, C), T, C, blk.norm1.weight.data, s["m1"], s["r1"], V(res, 0, C), V(g_in, 0, C),
                              blk.norm1.weight.grad, blk.norm1.bias.grad, wg, **nxt)
            copied = bool(nxt)
            free.append(g_mid)
            if res is not g_mid:
                free.append(res)
            gx = g_in
        # patch_embed.norm, long skip, conv_first
        pe = self.patch_embed.norm
        g_fea = Z(B, h, w, CP, dt=f32)
        ops.layernorm_bwd(V(gx, 0, C), V(sv["fea"], 0, C), T, C, pe.weight.data, sv["pe"][0], sv["pe"][1], V(g_body32, 0, C), V(g_fea, 0, C),
                          pe.weight.grad, pe.bias.grad, wg)
        gF = Z(B, h, w, CP)
        ops.axpby(V(g_fea), 1.0, ops.NULL_VIEW, 0.0, V(gF), T)
        wgrad_conv(self.conv_first, V(sv["x8"]), V(gF), h, w)

    # ------------------------------------------------------------------ nn.Module surface ------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            if self._trigger is None or self._trigger.device != x.device:
                self._trigger = torch.zeros(1, device=x.device, requires_grad=True)
            return _SwinFn.apply(x, self._trigger, self)
        return self.forward_hip(x, train=False)


class _SwinFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, trigger, net):
        ctx.net = net
        return net.forward_hip(x, train=True)

    @staticmethod
    def backward(ctx, gy):
        ctx.net.backward_hip(gy)
        return None, torch.zeros(1, device=gy.device), None


def _mk(scale, kw, defaults):
    cfg = dict(defaults)
    cfg.update(kw)
    return SwinIR(upscale=scale, **cfg)


@SPANDREL_REGISTRY.register()
def swinir_m(scale: int = 4, **kw) -> SwinIR:
    return _mk(scale, kw, dict(img_size=48, embed_dim=180, depths=[6] * 6, num_heads=[6] * 6, window_size=8, mlp_ratio=2.0, drop_path_rate=0.1,
                               upsampler="pixelshuffle", resi_connection="1conv"))


@SPANDREL_REGISTRY.register()
def swinir_s(scale: int = 4, **kw) -> SwinIR:
    return _mk(scale, kw, dict(img_size=64, embed_dim=60, depths=[6] * 4, num_heads=[6] * 4, window_size=8, mlp_ratio=2.0, drop_path_rate=0.1,
                               upsampler="pixelshuffledirect", resi_connection="1conv"))


@SPANDREL_REGISTRY.register()
def swinir_l(scale: int = 4, **kw) -> SwinIR:
    return _mk(scale, kw, dict(img_size=64, embed_dim=240, depths=[6] * 9, num_heads=[8] * 9, window_size=8, mlp_ratio=2.0, drop_path_rate=0.1,
                               upsampler="nearest+conv", resi_connection="3conv"))
