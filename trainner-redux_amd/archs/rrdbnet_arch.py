"""ESRGAN / RRDBNet generator on hand-written HIP kernels (gfx950).

Registered under the reference's names with the reference's factory signature
(/root/reference/traiNNer/archs/rrdbnet_arch.py:10-56): `esrgan`, `esrgan_lite` in SPANDREL_REGISTRY.
Arithmetic follows the in-tree twin of spandrel's ESRGAN
(/root/reference/traiNNer/archs/scunet_aaf6aa_arch.py:310-384); state-dict keys use spandrel's
old-arch layout so `.pth` / `.safetensors` files interchange (base_model.py:590-605).

Execution model:
  * fp32 OIHW master weights live in ONE flat arena (+ a matching gradient arena) so the optimizer
    sweep is a single kernel; MFMA operand slabs (bf16 or fp32) are re-packed from it by one launch;
  * activations are NHWC; each residual dense block owns one 192-channel buffer and every conv
    writes its output into a channel slice of it (torch.cat never happens);
  * the 64-channel residual stream is carried in fp32 (conv epilogue reads/writes it), a
    compute-dtype copy feeds the next MFMA operand;
  * backward is hand-scheduled, not autograd: input gradients run through the SAME conv kernel on
    transposed/flipped weight slabs in a mirrored dense layout (one 192-channel gradient buffer per
    block), LeakyReLU derivatives are an epilogue mask taken from the saved activations, weight
    gradients accumulate straight into the gradient arena.
There is no PyTorch fallback: without libtrainner_hip.so or off-GPU, forward raises.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import _lib as L
from .. import ops
from ..utils.registry import SPANDREL_REGISTRY
from ._arena import FlatParamModule

SLOPE = 0.2
RES = 0.2
_pixel_unshuffle_scales = (1, 2)


class _Conv(nn.Module):
    """Parameter holder named like nn.Conv2d (weight OIHW, bias)."""

    def __init__(self, cin: int, cout: int) -> None:
        super().__init__()
        self.cin, self.cout = cin, cout
        bound = 1.0 / math.sqrt(cin * 9)  # == kaiming_uniform_(a=sqrt(5)) bound of nn.Conv2d
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))


class _Seq(nn.Module):
    """`Sequential`-style numbering with gaps (parameter-free layers keep their index)."""

    def __init__(self, items: dict[int, nn.Module]) -> None:
        super().__init__()
        for k, v in items.items():
            self.add_module(str(k), v)


class _RDB(nn.Module):
    def __init__(self, nf: int, gc: int) -> None:
        super().__init__()
        for c in range(1, 6):
            self.add_module(f"conv{c}", _Seq({0: _Conv(nf + (c - 1) * gc, gc if c < 5 else nf)}))


class _RRDB(nn.Module):
    def __init__(self, nf: int, gc: int) -> None:
        super().__init__()
        self.RDB1, self.RDB2, self.RDB3 = _RDB(nf, gc), _RDB(nf, gc), _RDB(nf, gc)


class _Shortcut(nn.Module):
    def __init__(self, nf: int, gc: int, nb: int) -> None:
        super().__init__()
        self.sub = _Seq({**{i: _RRDB(nf, gc) for i in range(nb)}, nb: _Conv(nf, nf)})


class _Workspace:
    pass


class RRDBNet(FlatParamModule):
    hyperparameters: dict

    def __init__(self, *, in_nc=3, out_nc=3, num_filters=64, num_blocks=23, scale=4, shuffle_factor=None, gc=32) -> None:
        super().__init__()
        if num_filters % 32 or gc != 32:
            raise ValueError("the HIP dense-block layout needs num_filters % 32 == 0 and growth 32")
        if scale & (scale - 1):
            raise ValueError("scale must be a power of two")
        self.in_nc, self.out_nc, self.nf, self.nb, self.gc = in_nc, out_nc, num_filters, num_blocks, gc
        self.net_scale, self.shuffle_factor = scale, shuffle_factor
        self.n_up = int(math.log2(scale))
        self.hyperparameters = dict(in_nc=in_nc, out_nc=out_nc, num_filters=num_filters, num_blocks=num_blocks, scale=scale,
                                    shuffle_factor=shuffle_factor)
        items: dict[int, nn.Module] = {0: _Conv(in_nc, num_filters), 1: _Shortcut(num_filters, gc, num_blocks)}
        idx = 2
        for _ in range(self.n_up):
            items[idx + 1] = _Conv(num_filters, num_filters)
            idx += 3
        items[idx] = _Conv(num_filters, num_filters)
        items[idx + 2] = _Conv(num_filters, out_nc)
        self._idx_hr, self._idx_last = idx, idx + 2
        self.model = _Seq(items)
        self.compute_dtype = torch.bfloat16
        self._init_arena_state()

    # ------------------------------------------------------------------ weight packing --------------
    def _cd(self) -> int:
        return ops.dt_code(self.compute_dtype)

    def _build_plan(self) -> None:
        dev = self._arena.device
        cd = self._cd()
        plan = ops.PackPlan(dev)
        pk: dict = {}

        def slab(cin, cout):
            return torch.zeros(ops.packed_bytes(cin, cout, cd), dtype=torch.uint8, device=dev)

        def both(conv: _Conv, key, cin_pad=None, cout_pad_bwd=None):
            cin_f = cin_pad or conv.cin
            f = slab(cin_f, conv.cout)
            plan.add(conv.weight.data, f, cin_f, conv.cout, cd)
            bcin = cout_pad_bwd or conv.cout
            b = slab(bcin, conv.cin)
            plan.add(conv.weight.data, b, bcin, conv.cin, cd, transpose=True)
            pk[key] = (f, b)

        m = self.model
        c0 = getattr(m, "0")
        f0 = slab(8 * ((c0.cin + 7) // 8), c0.cout)
        plan.add(c0.weight.data, f0, 8 * ((c0.cin + 7) // 8), c0.cout, cd)
        pk["first"] = (f0, None)
        sub = getattr(m, "1").sub
        nf, gc = self.nf, self.gc
        for i in range(self.nb):
            rrdb = getattr(sub, str(i))
            for r in (1, 2, 3):
                rdb = getattr(rrdb, f"RDB{r}")
                cv = [getattr(getattr(rdb, f"conv{c}"), "0") for c in range(1, 6)]
                fw = []
                for c in cv:
                    s = slab(c.cin, c.cout)
                    plan.add(c.weight.data, s, c.cin, c.cout, cd)
                    fw.append(s)
                # mirrored dense backward: gradient buffer G = [g_c5 (nf) | g_c4 | g_c3 | g_c2 | g_c1]
                bw = {}
                for k in (4, 3, 2, 1, 0):
                    bcin = nf + gc * (4 - k) if k > 0 else nf + 4 * gc
                    bcout = gc if k > 0 else nf
                    i_off = nf + gc * (k - 1) if k > 0 else 0       # forward input-channel slice owned by x_k
                    s = slab(bcin, bcout)
                    for j in range(5, max(k, 0), -1):                # source convs conv_j, j = 5 .. k+1
                        g_off = 0 if j == 5 else nf + gc * (4 - j)   # where g_cj sits in G
                        plan.add(cv[j - 1].weight.data, s, bcin, bcout, cd, i_off=i_off, i_cnt=bcout, dst_i_off=g_off, transpose=True)
                    bw[k] = s
                pk[("rdb", i, r)] = (fw, bw)
        both(getattr(sub, str(self.nb)), "trunk")
        idx = 2
        for j in range(self.n_up):
            both(getattr(m, str(idx + 1)), ("up", j))
            idx += 3
        both(getattr(m, str(self._idx_hr)), "hr")
        both(getattr(m, str(self._idx_last)), "last", cout_pad_bwd=8 * ((self.out_nc + 7) // 8))
        plan.finalize()
        self._plan, self._pk, self._plan_cd = plan, pk, cd

    def repack(self) -> None:
        self._ensure_arena()
        if self._plan is None or self._plan_cd != self._cd():
            self._build_plan()
        self._plan.run()

    # ------------------------------------------------------------------ workspaces ------------------
    def _workspace(self, n: int, h: int, w: int, train: bool) -> _Workspace:
        key = (n, h, w, self._cd(), train)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        if len(self._ws) > 4:
            self._ws.clear()
        dev, cdt = self._arena.device, self.compute_dtype
        nf, gc, nb = self.nf, self.gc, self.nb
        ws = _Workspace()
        self._pool_epoch += 1       # new buffers: captured hipGraphs (archs/_arena.py) must not outlive the geometry they were recorded on
        e = lambda *s, dt=cdt: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
        ws.x8 = e(n, h, w, 8 * ((self.in_nc + 7) // 8))
        nblk = 3 * nb
        ws.X = [e(n, h, w, nf + 4 * gc) for _ in range(nblk if train else min(nblk, 2))]
        ws.S = [e(n, h, w, nf, dt=torch.float32) for _ in range(4)]     # S[0] = conv_first output
        ws.T = e(n, h, w, nf)
        ws.U = [e(n, h << j, w << j, nf) for j in range(self.n_up + 1)]
        H, W = h << self.n_up, w << self.n_up
        ws.UH = e(n, H, W, nf)
        ws.O4 = e(n, H, W, 4, dt=torch.float32)
        ws.H, ws.W = H, W
        self._ws[key] = ws
        return ws

    def _bwd_workspace(self, ws: _Workspace, n: int, h: int, w: int) -> None:
        if hasattr(ws, "G"):
            return
        self._pool_epoch += 1
        dev, cdt = self._arena.device, self.compute_dtype
        nf, gc = self.nf, self.gc
        e = lambda *s, dt=cdt: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
        ws.G = [e(n, h, w, nf + 4 * gc) for _ in range(2)]
        ws.GS = [e(n, h, w, nf, dt=torch.float32) for _ in range(4)]
        ws.g8 = e(n, ws.H, ws.W, 8 * ((self.out_nc + 7) // 8))
        ws.gUH = e(n, ws.H, ws.W, nf)
        ws.gU = [None] + [e(n, h << j, w << j, nf) for j in range(1, self.n_up + 1)]
        ws.gTmp = [None] + [e(n, h << j, w << j, nf) for j in range(1, self.n_up + 1)]
        ws.gT32 = e(n, h, w, nf, dt=torch.float32)
        ws.gT = e(n, h, w, nf)
        ws.gF = e(n, h, w, nf)
        need = 0
        shapes = [(h, w, nf + k * gc, gc) for k in range(4)] + [(h, w, nf + 4 * gc, nf), (h, w, self.in_nc, nf), (ws.H, ws.W, nf, nf),
                  (ws.H, ws.W, nf, self.out_nc)] + [(h << j, w << j, nf, nf) for j in range(self.n_up + 1)]
        for (hh, ww, cin, cout) in shapes:
            need = max(need, ops.wgrad_workspace_bytes(n, hh, ww, cin, cout))
        ws.wg = torch.empty(need // 4 + 16, dtype=torch.float32, device=dev)
        # one more workspace per conv of a dense block: its five weight gradients go out as one launch (ops.wgrad3x3_multi)
        need5 = max(ops.wgrad_workspace_bytes(n, h, w, nf + k * gc, gc if k < 4 else nf) for k in range(5))
        ws.wg5 = [ws.wg] + [torch.empty(need5 // 4 + 16, dtype=torch.float32, device=dev) for _ in range(4)]

    # ------------------------------------------------------------------ forward ---------------------
    def forward_hip(self, x: torch.Tensor, train: bool | None = None) -> torch.Tensor:
        """x: N x in_nc x h x w float32 on the GPU -> N x out_nc x (h*scale) x (w*scale) float32."""
        if not x.is_cuda:
            raise RuntimeError("RRDBNet (HIP) runs on the GPU only; there is no CPU path in this package")
        L.lib()
        if train is None:
            train = torch.is_grad_enabled() and self.training
        if self.shuffle_factor:
            # pure index permutation of the 3-channel input (rrdbnet_arch.py:19-29); no arithmetic
            x = torch.nn.functional.pixel_unshuffle(x, self.shuffle_factor)
        x = x.contiguous().float()
        n, cin, h, w = x.shape
        if cin != self.in_nc:
            raise ValueError(f"expected {self.in_nc} input channels, got {cin}")
        self.repack()
        cd = self._cd()
        ws = self._workspace(n, h, w, train)
        nf, gc, nb = self.nf, self.gc, self.nb
        V = ops.view
        pk, m = self._pk, self.model
        K = dict(n=n, compute=cd)
        ops.nchw_to_nhwc(x, ws.x8)
        c0 = getattr(m, "0")
        X = ws.X
        ops.conv3x3(h=h, w=w, cin=ws.x8.shape[-1], cout=nf, inp=V(ws.x8), out=V(ws.S[0]), out2=V(X[0], 0, nf), wpacked=pk["first"][0],
                    bias=c0.bias.data, **K)
        sub = getattr(m, "1").sub
        s_in = ws.S[0]
        nX = len(X)
        for i in range(nb):
            rrdb_in = s_in
            rrdb = getattr(sub, str(i))
            for r in (1, 2, 3):
                b = 3 * i + r - 1
                xb = X[b % nX]
                rdb = getattr(rrdb, f"RDB{r}")
                fw, _ = pk[("rdb", i, r)]
                for c in range(1, 5):
                    cv = getattr(getattr(rdb, f"conv{c}"), "0")
                    ops.conv3x3(h=h, w=w, cin=cv.cin, cout=gc, inp=V(xb, 0, cv.cin), out=V(xb, cv.cin, gc), wpacked=fw[c - 1],
                                bias=cv.bias.data, act=L.ACT_LRELU, slope=SLOPE, **K)
                cv = getattr(rdb.conv5, "0")
                s_out = next(s for s in ws.S[1:] if s is not s_in and s is not rrdb_in)
                last = b == 3 * nb - 1
                nxt = V(ws.T) if last else V(X[(b + 1) % nX], 0, nf)
                kw = dict(res1=V(s_in), scale1=RES)
                if r == 3:
                    kw.update(res2=V(rrdb_in), scale2=RES)
                ops.conv3x3(h=h, w=w, cin=cv.cin, cout=nf, inp=V(xb), out=V(s_out), out2=nxt, wpacked=fw[4], bias=cv.bias.data, **kw, **K)
                s_in = s_out
        ct = getattr(sub, str(nb))
        ops.conv3x3(h=h, w=w, cin=nf, cout=nf, inp=V(ws.T), out=V(ws.U[0]), wpacked=pk["trunk"][0], bias=ct.bias.data, res1=V(ws.S[0]),
                    scale1=1.0, **K)
        idx = 2
        for j in range(self.n_up):
            cu = getattr(m, str(idx + 1))
            ops.conv3x3(h=h << (j + 1), w=w << (j + 1), cin=nf, cout=nf, inp=V(ws.U[j]), out=V(ws.U[j + 1]), wpacked=pk[("up", j)][0],
                        bias=cu.bias.data, act=L.ACT_LRELU, slope=SLOPE, in_mode=L.IN_NEAREST2X, **K)
            idx += 3
        H, W = ws.H, ws.W
        ch = getattr(m, str(self._idx_hr))
        ops.conv3x3(h=H, w=W, cin=nf, cout=nf, inp=V(ws.U[self.n_up]), out=V(ws.UH), wpacked=pk["hr"][0], bias=ch.bias.data,
                    act=L.ACT_LRELU, slope=SLOPE, **K)
        cl = getattr(m, str(self._idx_last))
        ops.conv3x3(h=H, w=W, cin=nf, cout=self.out_nc, inp=V(ws.UH), out=V(ws.O4), wpacked=pk["last"][0], bias=cl.bias.data, **K)
        out = torch.empty(n, self.out_nc, H, W, dtype=torch.float32, device=x.device)
        ops.nhwc_to_nchw(V(ws.O4), out)
        self._saved = (ws, n, h, w) if train else None
        return out

    # ------------------------------------------------------------------ backward --------------------
    def backward_hip(self, grad_out: torch.Tensor) -> None:
        """Accumulates d(loss)/d(param) into the gradient arena (== every `p.grad`), given d(loss)/d(output)."""
        if self._saved is None:
            raise RuntimeError("backward_hip needs a preceding training-mode forward_hip")
        ws, n, h, w = self._saved
        self._saved = None
        self._ensure_arena()
        self._bwd_workspace(ws, n, h, w)
        grad_out = grad_out.contiguous().float()
        cd = self._cd()
        nf, gc, nb = self.nf, self.gc, self.nb
        V = ops.view
        pk, m = self._pk, self.model
        K = dict(n=n, compute=cd)
        H, W = ws.H, ws.W

        def wgrad(conv: _Conv, inp, gy, hh, ww, cin=None, cout=None, in_mode=L.IN_DIRECT):
            ops.wgrad3x3(n=n, h=hh, w=ww, cin=cin or conv.cin, cout=cout or conv.cout, compute=cd, inp=inp, gy=gy, dw=conv.weight.grad,
                         db=conv.bias.grad, workspace=ws.wg, in_mode=in_mode, accumulate=True)

        ops.nchw_to_nhwc(grad_out, ws.g8)
        cl = getattr(m, str(self._idx_last))
        wgrad(cl, V(ws.UH), V(ws.g8), H, W)
        self._grads_final_from(cl.weight)       # data parallel: the gradient exchange starts on finished arena ranges while backward runs on
        ops.conv3x3(h=H, w=W, cin=ws.g8.shape[-1], cout=nf, inp=V(ws.g8), out=V(ws.gUH), wpacked=pk["last"][1], mask=V(ws.UH),
                    mask_slope=SLOPE, **K)
        ch = getattr(m, str(self._idx_hr))
        top = ws.U[self.n_up]
        wgrad(ch, V(top), V(ws.gUH), H, W)
        self._grads_final_from(ch.weight)
        if self.n_up == 0:
            ops.conv3x3(h=H, w=W, cin=nf, cout=nf, inp=V(ws.gUH), out=V(ws.gT32), out2=V(ws.gT), wpacked=pk["hr"][1], **K)
        else:
            ops.conv3x3(h=H, w=W, cin=nf, cout=nf, inp=V(ws.gUH), out=V(ws.gU[self.n_up]), wpacked=pk["hr"][1], mask=V(top),
                        mask_slope=SLOPE, **K)
        idx = 2 + 3 * (self.n_up - 1)
        for j in range(self.n_up, 0, -1):
            cu = getattr(m, str(idx + 1))
            hj, wj = h << j, w << j
            wgrad(cu, V(ws.U[j - 1]), V(ws.gU[j]), hj, wj, in_mode=L.IN_NEAREST2X)
            self._grads_final_from(cu.weight)
            ops.conv3x3(h=hj, w=wj, cin=nf, cout=nf, inp=V(ws.gU[j]), out=V(ws.gTmp[j]), wpacked=pk[("up", j - 1)][1], **K)
            if j > 1:
                ops.sumpool2x2(V(ws.gTmp[j]), V(ws.gU[j - 1]), V(ws.U[j - 1]), SLOPE, n, hj, wj)
            else:
                ops.sumpool2x2(V(ws.gTmp[j]), V(ws.gT32), ops.NULL_VIEW, 1.0, n, hj, wj)
                ops.axpby(V(ws.gT32), 1.0, ops.NULL_VIEW, 0.0, V(ws.gT), n * h * w)
            idx -= 3
        sub = getattr(m, "1").sub
        ct = getattr(sub, str(nb))
        wgrad(ct, V(ws.T), V(ws.gT), h, w)
        self._grads_final_from(ct.weight)
        G = ws.G
        cur = 0
        g_z = ws.GS[0]
        ops.conv3x3(h=h, w=w, cin=nf, cout=nf, inp=V(ws.gT), out=V(g_z), out2=V(G[cur], 0, nf), out2_scale=RES * RES,
                    wpacked=pk["trunk"][1], **K)
        for i in range(nb - 1, -1, -1):
            rrdb = getattr(sub, str(i))
            g_y = g_z
            for r in (3, 2, 1):
                b = 3 * i + r - 1
                xb = ws.X[b]
                rdb = getattr(rrdb, f"RDB{r}")
                _, bw = pk[("rdb", i, r)]
                g = G[cur]
                for k in (4, 3, 2, 1):
                    bcin = nf + gc * (4 - k)
                    ops.conv3x3(h=h, w=w, cin=bcin, cout=gc, inp=V(g, 0, bcin), out=V(g, bcin, gc), wpacked=bw[k],
                                mask=V(xb, nf + gc * (k - 1), gc), mask_slope=SLOPE, **K)
                busy = {id(g_y), id(g_z)}
                g_next = next(s for s in ws.GS if id(s) not in busy)
                kw = {}
                if r == 3:
                    kw.update(res1=V(g_z), res1_scale=RES, out2=V(G[cur ^ 1], 0, nf), out2_scale=RES)
                elif r == 2:
                    kw.update(res1=V(g_y), out2=V(G[cur ^ 1], 0, nf), out2_scale=RES)
                else:
                    kw.update(res1=V(g_y), res2=V(g_z))
                    if i > 0:
                        kw.update(out2=V(G[cur ^ 1], 0, nf), out2_scale=RES * RES)
                ops.conv3x3(h=h, w=w, cin=nf + 4 * gc, cout=nf, inp=V(g), out=V(g_next), wpacked=bw[0], **kw, **K)
                # the five weight gradients of the block read channel prefixes of xb and slices of g: one launch (tr_wgrad3x3_multi)
                convs = []
                for c in range(1, 6):
                    cv = getattr(getattr(rdb, f"conv{c}"), "0")
                    goff = 0 if c == 5 else nf + gc * (4 - c)
                    convs.append(dict(cin=cv.cin, cout=cv.cout, inp=V(xb, 0, cv.cin), gy=V(g, goff, cv.cout), dw=cv.weight.grad, db=cv.bias.grad,
                                      workspace=ws.wg5[c - 1]))
                ops.wgrad3x3_multi(convs, n=n, h=h, w=w, compute=cd, accumulate=True)
                self._grads_final_from(getattr(rdb.conv1, "0").weight)      # conv1 is the block's first parameter in the arena
                g_y = g_next
                cur ^= 1
            g_z = g_y
        # conv_first sees the dense-trunk gradient plus the long skip (fea + trunk_conv(...))
        ops.axpby(V(g_z), 1.0, V(ws.gT32), 1.0, V(ws.gF), n * h * w)
        c0 = getattr(m, "0")
        wgrad(c0, V(ws.x8), V(ws.gF), h, w, cin=c0.cin)

    # ------------------------------------------------------------------ nn.Module surface -----------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            if self._trigger is None or self._trigger.device != x.device:
                self._trigger = torch.zeros(1, device=x.device, requires_grad=True)
            return _RRDBNetFn.apply(x, self._trigger, self)
        return self.forward_hip(x, train=False)


class _RRDBNetFn(torch.autograd.Function):
    """Autograd bridge for callers that use `loss.backward()` (e.g. the reference's own SRModel,
    sr_model.py:478,681): parameter gradients are accumulated into `p.grad` by backward_hip."""

    @staticmethod
    def forward(ctx, x, trigger, net):
        ctx.net = net
        return net.forward_hip(x, train=True)

    @staticmethod
    def backward(ctx, gy):
        ctx.net.backward_hip(gy)
        return None, torch.zeros(1, device=gy.device), None


@SPANDREL_REGISTRY.register()
def esrgan(scale: int = 4, use_pixel_unshuffle: bool = True, in_nc: int = 3, out_nc: int = 3, num_filters: int = 64,
           num_blocks: int = 23) -> RRDBNet:
    if use_pixel_unshuffle and scale in _pixel_unshuffle_scales:
        in_nc *= 4 ** (3 - scale)
        shuffle_factor = int(math.sqrt(in_nc / out_nc))
        return RRDBNet(scale=4, in_nc=in_nc, out_nc=out_nc, shuffle_factor=shuffle_factor, num_blocks=num_blocks, num_filters=num_filters)
    return RRDBNet(scale=scale, in_nc=in_nc, out_nc=out_nc, num_blocks=num_blocks, num_filters=num_filters)


@SPANDREL_REGISTRY.register()
def esrgan_lite(scale: int = 4, use_pixel_unshuffle: bool = True, in_nc: int = 3, out_nc: int = 3, num_filters: int = 32,
                num_blocks: int = 12) -> RRDBNet:
    return esrgan(scale=scale, use_pixel_unshuffle=use_pixel_unshuffle, in_nc=in_nc, out_nc=out_nc, num_filters=num_filters,
                  num_blocks=num_blocks)
