"""Thin tensor-level wrappers over the C ABI (device pointers + current HIP stream).

PyTorch is used for device memory and streams only; every function here launches hand-written
HIP kernels from libtrainner_hip.so on `torch.cuda.current_stream()` and never computes with ATen.
NHWC activations are torch tensors of shape (N, H, W, C), contiguous, dtype float32 or bfloat16.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}


def dt_code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {dtype}; the HIP path handles float32, bfloat16 and float16") from None


def torch_dtype(code: int) -> torch.dtype:
    return torch.bfloat16 if code == L.BF16 else torch.float16 if code == L.F16 else torch.float32


def stream_ptr() -> int:
    # raw handle of torch's current stream on the current device; `torch.cuda.current_stream().cuda_stream` builds a Stream object per
    # call (8 us x ~1,700 launches per ESRGAN step)
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:      # private API moved: the public path
        return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("trainner_redux_amd ops run on the GPU only (tensor is on %s)" % t.device)


def view(t: torch.Tensor | None, coff: int = 0, c: int | None = None) -> L.View:
    """Channel-slice view of an NHWC tensor (last dim = channels, contiguous)."""
    if t is None:
        return L.View(None, 0, 0, 0, 0)
    if not t.is_contiguous():
        raise ValueError("NHWC buffer must be contiguous")
    cs = t.shape[-1]
    c = cs - coff if c is None else c
    if coff < 0 or c <= 0 or coff + c > cs:
        raise ValueError(f"bad channel slice [{coff}, {coff + c}) of {cs}")
    return L.View(t.data_ptr(), dt_code(t.dtype), cs, coff, c)


NULL_VIEW = L.View(None, 0, 0, 0, 0)


def tune_set(key: str, value: int) -> None:
    """tr_tune_set: process-wide kernel-selection switch (include/trainner_hip.h lists the keys)."""
    L.check(L.lib().tr_tune_set(key.encode(), int(value)), f"tr_tune_set({key})")


def tune_get(key: str) -> int:
    v = int(L.lib().tr_tune_get(key.encode()))
    if v == -1 and key not in ("ring_cfg",):
        L.check(-1, f"tr_tune_get({key})")
    return v


class tuned:
    """`with ops.tuned(conv_impl=2, ring_cfg=32): ...` -- set switches for a block, restore the previous values after it."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = int(L.lib().tr_tune_get(k.encode()))
            tune_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            tune_set(k, v)
        return False


CONV_FAMILIES = {0: "none", 1: "reg", 2: "dma", 3: "linear", 4: "ring", 5: "linear_dma"}


def last_conv_family() -> str:
    """Which kernel family served the last tr_conv3x3 launch of this process."""
    return CONV_FAMILIES[tune_get("last_conv_family")]


def packed_bytes(cin: int, cout: int, compute: int, ksize: int = 3) -> int:
    return int(L.lib().tr_conv_packed_bytes(cin, cout, compute, ksize))


def conv3x3(*, n, h, w, cin, cout, compute, inp: L.View, out: L.View, wpacked: torch.Tensor, bias: torch.Tensor | None = None,
            in_mode=L.IN_DIRECT, pad_mode=L.PAD_ZEROS, act=L.ACT_NONE, slope=0.2, res1: L.View = NULL_VIEW, scale1=1.0,
            res2: L.View = NULL_VIEW, scale2=1.0, res1_scale=1.0, res2_scale=1.0, ksize=3, mask_mode=0, sample_scale: torch.Tensor | None = None, mask: L.View = NULL_VIEW, mask_slope=0.2, out2: L.View = NULL_VIEW, out2_scale=1.0):
    d = L.Conv3x3Desc()
    d.n, d.h, d.w, d.cin, d.cout = n, h, w, cin, cout
    d.compute, d.in_mode, d.pad_mode, d.act = compute, in_mode, pad_mode, act
    d.slope, d.scale1, d.scale2, d.mask_slope, d.out2_scale = slope, scale1, scale2, mask_slope, out2_scale
    d.res1_scale, d.res2_scale = res1_scale, res2_scale
    d.inp, d.out, d.out2, d.res1, d.res2, d.mask = inp, out, out2, res1, res2, mask
    d.wpacked = wpacked.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.ksize, d.mask_mode = ksize, mask_mode
    d.sample_scale = sample_scale.data_ptr() if sample_scale is not None else None
    if compute == L.F32_WINOGRAD:
        d.compute = L.F32
        L.check(L.lib().tr_conv3x3_winograd_f32(stream_ptr(), C.byref(d)), "tr_conv3x3_winograd_f32")
        return
    if compute == L.F32_WINOGRAD43:
        d.compute = L.F32
        L.check(L.lib().tr_conv3x3_winograd43_f32(stream_ptr(), C.byref(d)), "tr_conv3x3_winograd43_f32")
        return
    L.check(L.lib().tr_conv3x3(stream_ptr(), C.byref(d)), "tr_conv3x3")


def winograd_eligible(cin: int, cout: int, h: int, w: int) -> bool:
    return cin % 8 == 0 and cout % 64 == 0 and h % 2 == 0 and w % 2 == 0


def pack_winograd(weight: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    """fp32 OIHW 3x3 weight -> Winograd-domain slab of tr_conv3x3_winograd_f32 (transpose: the slab of the input-gradient conv)."""
    assert weight.dtype == torch.float32 and weight.is_contiguous() and weight.dim() == 4 and weight.shape[2:] == (3, 3)
    o, i = weight.shape[:2]
    cout, cin = (i, o) if transpose else (o, i)
    dst = torch.empty(cin * cout * 16, dtype=torch.float32, device=weight.device)
    L.check(L.lib().tr_pack_winograd(stream_ptr(), weight.data_ptr(), dst.data_ptr(), cout, cin, 1 if transpose else 0), "tr_pack_winograd")
    return dst


def winograd43_eligible(cin: int, cout: int, h: int, w: int, cstride_in: int | None = None, cstride_out: int | None = None) -> bool:
    """The limits tr_conv3x3_winograd43_f32 TR_REQUIREs: more than four 8-channel chunks (the halo stream's lookahead), 64-wide output groups,
    H and W multiples of 4, one fp32 image of either view below the 2 GiB of buffer addressing."""
    img = h * w * 4
    return cin % 8 == 0 and cin > 32 and cout % 64 == 0 and h % 4 == 0 and w % 4 == 0 and \
        img * (cstride_in or cin) < 0x7fff0000 and img * (cstride_out or cout) < 0x7fff0000


def pack_winograd43(weight: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    """fp32 OIHW 3x3 weight -> F(4x4, 3x3) Winograd-domain slab of tr_conv3x3_winograd43_f32 (transpose: the slab of the input-gradient conv)."""
    assert weight.dtype == torch.float32 and weight.is_contiguous() and weight.dim() == 4 and weight.shape[2:] == (3, 3)
    o, i = weight.shape[:2]
    cout, cin = (i, o) if transpose else (o, i)
    dst = torch.empty(cin * cout * 36, dtype=torch.float32, device=weight.device)
    L.check(L.lib().tr_pack_winograd43(stream_ptr(), weight.data_ptr(), dst.data_ptr(), cout, cin, 1 if transpose else 0), "tr_pack_winograd43")
    return dst


class PackPlan:
    """A device-resident segment table: one launch re-packs every conv operand of a network."""

    def __init__(self, device):
        self.device = device
        self.segs: list[L.PackSeg] = []
        self.total = 0
        self._dev = None
        self._keep = []

    def add(self, src: torch.Tensor, dst: torch.Tensor, dst_cin: int, dst_cout: int, compute: int, *, o_off=0, o_cnt=None, i_off=0,
            i_cnt=None, dst_o_off=0, dst_i_off=0, transpose=False):
        assert src.dtype == torch.float32 and src.is_contiguous()
        if src.dim() == 2:
            ks = 1                                   # nn.Linear weight (out_features, in_features)
        else:
            assert src.dim() == 4 and src.shape[2] == src.shape[3] and src.shape[2] in (1, 3)
            ks = src.shape[2]
        so, si = src.shape[0], src.shape[1]
        o_cnt = so - o_off if o_cnt is None else o_cnt
        i_cnt = si - i_off if i_cnt is None else i_cnt
        s = L.PackSeg(src.data_ptr(), dst.data_ptr(), so, si, o_off, o_cnt, i_off, i_cnt, dst_cin, dst_cout, dst_o_off, dst_i_off,
                      1 if transpose else 0, compute, ks, 0, self.total)
        self.segs.append(s)
        self.total += o_cnt * i_cnt * ks * ks
        self._keep.append((src, dst))
        self._dev = None

    def finalize(self):
        arr = (L.PackSeg * len(self.segs))(*self.segs)
        raw = bytes(arr)
        host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        self._dev = host.to(self.device)
        return self

    def run(self):
        if not self.segs:
            return
        if self._dev is None:
            self.finalize()
        L.check(L.lib().tr_pack_conv3x3(stream_ptr(), self._dev.data_ptr(), len(self.segs), self.total), "tr_pack_conv3x3")


def wgrad_workspace_bytes(n, h, w, cin, cout, ksize: int = 3) -> int:
    return int(L.lib().tr_wgrad_workspace_bytes(n, h, w, cin, cout, ksize))


def wgrad3x3(*, n, h, w, cin, cout, compute, inp: L.View, gy: L.View, dw: torch.Tensor, db: torch.Tensor | None, workspace: torch.Tensor,
             in_mode=L.IN_DIRECT, pad_mode=L.PAD_ZEROS, accumulate=False, ksize=3, dw_ci0=0, dw_ld=0):
    """dw_ld / dw_ci0: fill the input-channel block [dw_ci0, dw_ci0 + cin) of a weight whose full input width is dw_ld."""
    d = L.Wgrad3x3Desc()
    d.n, d.h, d.w, d.cin, d.cout = n, h, w, cin, cout
    d.compute, d.in_mode, d.pad_mode, d.accumulate = compute, in_mode, pad_mode, 1 if accumulate else 0
    d.inp, d.gy = inp, gy
    d.ksize = ksize
    d.dw_ld = dw_ld
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == cout * (dw_ld or cin) * ksize * ksize and dw_ci0 + cin <= (dw_ld or cin)
    d.dw = dw.data_ptr() + 4 * dw_ci0 * ksize * ksize
    d.db = db.data_ptr() if db is not None else None
    d.partials = workspace.data_ptr()
    d.partials_bytes = workspace.numel() * workspace.element_size()
    L.check(L.lib().tr_wgrad3x3(stream_ptr(), C.byref(d)), "tr_wgrad3x3")


def wgrad3x3_multi(convs, *, n, h, w, compute, pad_mode=L.PAD_ZEROS, accumulate=False):
    """Weight / bias gradients of several 3x3 convs over channel slices of one input buffer and one output-gradient buffer (a dense block) in one
    launch (tr_wgrad3x3_multi).  `convs`: dicts with cin, cout, inp, gy (views), dw, db (tensors or None), workspace (its own tensor per conv)."""
    descs = (L.Wgrad3x3Desc * len(convs))()
    for d, c in zip(descs, convs):
        d.n, d.h, d.w, d.cin, d.cout = n, h, w, c["cin"], c["cout"]
        d.compute, d.in_mode, d.pad_mode, d.accumulate = compute, L.IN_DIRECT, pad_mode, 1 if accumulate else 0
        d.inp, d.gy = c["inp"], c["gy"]
        d.ksize, d.dw_ld = 3, 0
        dw, db, ws = c["dw"], c.get("db"), c["workspace"]
        assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == c["cout"] * c["cin"] * 9
        d.dw = dw.data_ptr()
        d.db = db.data_ptr() if db is not None else None
        d.partials = ws.data_ptr()
        d.partials_bytes = ws.numel() * ws.element_size()
    L.check(L.lib().tr_wgrad3x3_multi(stream_ptr(), descs, len(convs)), "tr_wgrad3x3_multi")


def _f3(vals):
    return (C.c_float * 3)(*[float(v) for v in vals])


def nchw_to_nhwc(src: torch.Tensor, dst: torch.Tensor, mean=None, std=None, dst_view: L.View | None = None):
    _need_cuda(src, dst)
    assert src.dtype == torch.float32 and src.is_contiguous() and src.dim() == 4
    n, c, h, w = src.shape
    v = dst_view if dst_view is not None else view(dst)
    m = _f3(mean) if mean is not None else None
    s = _f3(std) if std is not None else None
    L.check(L.lib().tr_nchw_to_nhwc(stream_ptr(), src.data_ptr(), n, c, h, w, v, m, s), "tr_nchw_to_nhwc")


def nhwc_to_nchw(src_view: L.View, dst: torch.Tensor):
    assert dst.dtype == torch.float32 and dst.is_contiguous() and dst.dim() == 4
    n, c, h, w = dst.shape
    L.check(L.lib().tr_nhwc_to_nchw(stream_ptr(), src_view, dst.data_ptr(), n, c, h, w), "tr_nhwc_to_nchw")


def nhwc_to_nchw_scaled(src_view: L.View, dst: torch.Tensor, inv_std, accumulate: bool):
    assert dst.dtype == torch.float32 and dst.is_contiguous() and dst.dim() == 4
    n, c, h, w = dst.shape
    L.check(L.lib().tr_nhwc_to_nchw_scaled(stream_ptr(), src_view, dst.data_ptr(), n, c, h, w, _f3(inv_std), 1 if accumulate else 0),
            "tr_nhwc_to_nchw_scaled")


def conv3x3_dgrad_image(gy: L.View, w_oihw: torch.Tensor, n, h, w, dst: torch.Tensor, *, pad_mode=L.PAD_ZEROS, ch_scale=None,
                        accumulate=False):
    assert w_oihw.dtype == torch.float32 and w_oihw.is_contiguous() and dst.dtype == torch.float32 and dst.is_contiguous()
    cout, cin = w_oihw.shape[0], w_oihw.shape[1]
    sc = (C.c_float * cin)(*[float(v) for v in ch_scale]) if ch_scale is not None else None
    L.check(L.lib().tr_conv3x3_dgrad_image(stream_ptr(), gy, w_oihw.data_ptr(), cout, cin, n, h, w, pad_mode, sc, dst.data_ptr(),
                                           1 if accumulate else 0), "tr_conv3x3_dgrad_image")


def conv3x3_stem(img: torch.Tensor, w_oihw: torch.Tensor, bias: torch.Tensor | None, out: L.View, *, mean=None, std=None, pad_mode=L.PAD_ZEROS,
                 act=L.ACT_NONE):
    """3x3 stem conv straight from an NCHW fp32 image batch: normalise, pad, conv (cin <= 3), bias, ReLU -> NHWC fp32 view."""
    _need_cuda(img, w_oihw, bias)
    assert img.dtype == torch.float32 and img.is_contiguous() and img.dim() == 4 and w_oihw.dtype == torch.float32 and w_oihw.is_contiguous()
    n, cin, h, w = img.shape
    cout = w_oihw.shape[0]
    assert w_oihw.shape[1] == cin and tuple(w_oihw.shape[2:]) == (3, 3)
    L.check(L.lib().tr_conv3x3_stem(stream_ptr(), img.data_ptr(), w_oihw.data_ptr(), bias.data_ptr() if bias is not None else None,
                                    _f3(mean) if mean is not None else None, _f3(std) if std is not None else None, cin, cout, n, h, w, pad_mode, act, out),
            "tr_conv3x3_stem")


def fold_replicate_nchw(gp: L.View, dst: torch.Tensor, ch_scale=None, accumulate=False):
    assert dst.dtype == torch.float32 and dst.is_contiguous() and dst.dim() == 4
    n, c, h, w = dst.shape
    sc = (C.c_float * c)(*[float(v) for v in ch_scale]) if ch_scale is not None else None
    L.check(L.lib().tr_fold_replicate_nchw(stream_ptr(), gp, dst.data_ptr(), n, c, h, w, sc, 1 if accumulate else 0), "tr_fold_replicate_nchw")


def nhwc_to_nchw_affine(src_view: L.View, dst: torch.Tensor, scale, offset):
    assert dst.dtype == torch.float32 and dst.is_contiguous() and dst.dim() == 4
    n, c, h, w = dst.shape
    L.check(L.lib().tr_nhwc_to_nchw_affine(stream_ptr(), src_view, dst.data_ptr(), n, c, h, w, _f3(scale), _f3(offset)), "tr_nhwc_to_nchw_affine")


def layernorm(x: L.View, y: L.View, ntok: int, c: int, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, mean: torch.Tensor | None = None,
              rstd: torch.Tensor | None = None):
    L.check(L.lib().tr_layernorm(stream_ptr(), x, y, ntok, c, gamma.data_ptr(), beta.data_ptr(), eps, mean.data_ptr() if mean is not None else None,
                                 rstd.data_ptr() if rstd is not None else None), "tr_layernorm")


def layernorm_bwd(g: L.View, x: L.View, ntok: int, c: int, gamma, mean, rstd, res: L.View, dx: L.View, dgamma, dbeta, workspace: torch.Tensor,
                  accumulate=True, dx2: L.View = NULL_VIEW, sample_scale: torch.Tensor | None = None, pix_per_sample: int = 0):
    """dx2: optional second output = dx * sample_scale[token // pix_per_sample] in the view's dtype (the GEMM-operand copy of the gradient stream)."""
    L.check(L.lib().tr_layernorm_bwd(stream_ptr(), g, x, ntok, c, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), res, dx, dgamma.data_ptr(),
                                     dbeta.data_ptr(), 1 if accumulate else 0, workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                     dx2, sample_scale.data_ptr() if sample_scale is not None else None, pix_per_sample),
            "tr_layernorm_bwd")


def layernorm_bwd_workspace_bytes(c: int) -> int:
    return int(L.lib().tr_layernorm_bwd_workspace_bytes(c))


def _attn_desc(qkv, table, b, h, w, c, nh, shift):
    d = L.WindowAttnDesc()
    d.qkv, d.table = qkv, table.data_ptr()
    d.b, d.h, d.w, d.c, d.nh, d.window, d.shift = b, h, w, c, nh, 8, shift
    d.scale = float((c // nh) ** -0.5)
    return d


def window_attention(qkv: L.View, out: L.View, table: torch.Tensor, *, b, h, w, c, nh, shift):
    d = _attn_desc(qkv, table, b, h, w, c, nh, shift)
    d.out = out
    L.check(L.lib().tr_window_attention(stream_ptr(), C.byref(d)), "tr_window_attention")


def window_attention_bwd_workspace_bytes(b, h, w, nh) -> int:
    return int(L.lib().tr_window_attention_bwd_workspace_bytes(b, h, w, nh))


def window_attention_bwd(qkv: L.View, dout: L.View, dqkv: L.View, table: torch.Tensor, dtable: torch.Tensor | None, workspace: torch.Tensor | None, *,
                         b, h, w, c, nh, shift, accumulate=True):
    d = _attn_desc(qkv, table, b, h, w, c, nh, shift)
    d.dout, d.dqkv = dout, dqkv
    d.dtable = dtable.data_ptr() if dtable is not None else None
    d.dbias_ws = workspace.data_ptr() if (workspace is not None and dtable is not None) else None
    d.accumulate_dtable = 1 if accumulate else 0
    L.check(L.lib().tr_window_attention_bwd(stream_ptr(), C.byref(d)), "tr_window_attention_bwd")


def pixel_shuffle(inp: L.View, out: L.View, n, h, w, c, r, inverse=False):
    L.check(L.lib().tr_pixel_shuffle(stream_ptr(), inp, out, n, h, w, c, r, 1 if inverse else 0), "tr_pixel_shuffle")


def axpby(a: L.View, sa: float, b: L.View, sb: float, out: L.View, npix: int, mask: L.View | None = None, mask_slope: float = 0.2,
          sample_scale: torch.Tensor | None = None, pix_per_sample: int = 0):
    L.check(L.lib().tr_axpby(stream_ptr(), a, sa, b, sb, out, npix, mask if mask is not None else NULL_VIEW, mask_slope,
                             sample_scale.data_ptr() if sample_scale is not None else None, pix_per_sample), "tr_axpby")


def u8hwc_to_nchw(src: torch.Tensor, dst: torch.Tensor, mean=None, std=None):
    """uint8 (N, H, W, C) device batch -> fp32 (N, C, H, W) in [0, 1] (optionally normalised)."""
    _need_cuda(src, dst)
    assert src.dtype == torch.uint8 and src.is_contiguous() and src.dim() == 4 and dst.dtype == torch.float32 and dst.is_contiguous()
    n, h, w, c = src.shape
    assert dst.shape == (n, c, h, w)
    L.check(L.lib().tr_u8hwc_to_nchw(stream_ptr(), src.data_ptr(), n, h, w, c, dst.data_ptr(), _f3(mean) if mean is not None else None,
                                     _f3(std) if std is not None else None), "tr_u8hwc_to_nchw")


def span_gate(o3: L.View, x: L.View, out: L.View, out2: L.View | None, npix: int):
    L.check(L.lib().tr_span_gate(stream_ptr(), o3, x, out, out2 if out2 is not None else NULL_VIEW, npix), "tr_span_gate")


def span_gate_bwd(g: L.View, o3: L.View, x: L.View, g_o3: L.View, g_x: L.View | None, npix: int, accumulate_gx: bool = False):
    L.check(L.lib().tr_span_gate_bwd(stream_ptr(), g, o3, x, g_o3, g_x if g_x is not None else NULL_VIEW, 1 if accumulate_gx else 0, npix),
            "tr_span_gate_bwd")


def dwconv(inp: L.View, out: L.View, weight: torch.Tensor, bias: torch.Tensor | None, n, h, w, c, flip: bool = False):
    """Depthwise k x k conv (groups = c) on NHWC channel slices; flip = the input gradient of the same conv."""
    assert weight.dtype == torch.float32 and weight.is_contiguous() and weight.shape[0] == c and weight.shape[1] == 1 and weight.shape[2] == weight.shape[3]
    L.check(L.lib().tr_dwconv(stream_ptr(), inp, out, weight.data_ptr(), bias.data_ptr() if bias is not None else None, n, h, w, c, weight.shape[2],
                              1 if flip else 0), "tr_dwconv")


def dwconv_wgrad_workspace_bytes(n, h, w, c, ksize) -> int:
    return int(L.lib().tr_dwconv_wgrad_workspace_bytes(n, h, w, c, ksize))


def dwconv_wgrad(inp: L.View, gy: L.View, dw: torch.Tensor, db: torch.Tensor | None, n, h, w, c, workspace: torch.Tensor, accumulate=True):
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.shape[0] == c and dw.shape[1] == 1 and dw.shape[2] == dw.shape[3]
    L.check(L.lib().tr_dwconv_wgrad(stream_ptr(), inp, gy, dw.data_ptr(), db.data_ptr() if db is not None else None, n, h, w, c, dw.shape[2],
                                    1 if accumulate else 0, workspace.data_ptr(), workspace.numel() * workspace.element_size()), "tr_dwconv_wgrad")


def gate_mish(g: L.View, a: L.View, b: L.View, out: L.View, npix: int):
    L.check(L.lib().tr_gate_mish(stream_ptr(), g, a, b, out, npix), "tr_gate_mish")


def gate_mish_bwd(d: L.View, g: L.View, a: L.View, b: L.View, dg: L.View, da: L.View, db: L.View, npix: int):
    L.check(L.lib().tr_gate_mish_bwd(stream_ptr(), d, g, a, b, dg, da, db, npix), "tr_gate_mish_bwd")


def act_grad(act: int, g: L.View, pre: L.View, out: L.View, npix: int, sample_scale: torch.Tensor | None = None, pix_per_sample: int = 0):
    L.check(L.lib().tr_act_grad(stream_ptr(), act, g, pre, out, npix, sample_scale.data_ptr() if sample_scale is not None else None, pix_per_sample),
            "tr_act_grad")


def space_to_depth2(inp: L.View, out: L.View, n, h, w):
    L.check(L.lib().tr_space_to_depth2(stream_ptr(), inp, out, n, h, w), "tr_space_to_depth2")


def depth_to_space2(inp: L.View, out: L.View, n, h, w):
    L.check(L.lib().tr_depth_to_space2(stream_ptr(), inp, out, n, h, w), "tr_depth_to_space2")


def expand_w4x4s2(w4: torch.Tensor, w3: torch.Tensor):
    cout, cin = w4.shape[0], w4.shape[1]
    assert w4.shape[2:] == (4, 4) and w3.shape == (cout, 4 * cin, 3, 3) and w4.is_contiguous() and w3.is_contiguous()
    L.check(L.lib().tr_expand_w4x4s2(stream_ptr(), w4.data_ptr(), w3.data_ptr(), cout, cin), "tr_expand_w4x4s2")


def collapse_g3x3s2d(g3: torch.Tensor, g4: torch.Tensor, accumulate=False):
    cout, cin = g4.shape[0], g4.shape[1]
    assert g4.shape[2:] == (4, 4) and g3.shape == (cout, 4 * cin, 3, 3) and g3.is_contiguous() and g4.is_contiguous()
    L.check(L.lib().tr_collapse_g3x3s2d(stream_ptr(), g3.data_ptr(), g4.data_ptr(), cout, cin, 1 if accumulate else 0), "tr_collapse_g3x3s2d")


def bilinear_up2x(inp: L.View, out: L.View, n, h_in, w_in):
    L.check(L.lib().tr_bilinear_up2x(stream_ptr(), inp, out, n, h_in, w_in), "tr_bilinear_up2x")


def bilinear_up2x_bwd(gout: L.View, gin: L.View, n, h_in, w_in):
    L.check(L.lib().tr_bilinear_up2x_bwd(stream_ptr(), gout, gin, n, h_in, w_in), "tr_bilinear_up2x_bwd")


def spectral_norm(w_orig: torch.Tensor, u: torch.Tensor, v: torch.Tensor, w_out: torch.Tensor, sigma: torch.Tensor, power_iteration: bool,
                  eps: float = 1e-12):
    rows = w_orig.shape[0]
    cols = w_orig.numel() // rows
    assert u.numel() == rows and v.numel() == cols and w_out.numel() == w_orig.numel() and w_orig.is_contiguous()
    L.check(L.lib().tr_spectral_norm(stream_ptr(), w_orig.data_ptr(), u.data_ptr(), v.data_ptr(), w_out.data_ptr(), sigma.data_ptr(), rows, cols,
                                     1 if power_iteration else 0, eps, scratch(w_orig.device, rows + cols + 64).data_ptr()), "tr_spectral_norm")


def spectral_norm_bwd(g_wsn: torch.Tensor, w_sn: torch.Tensor, u, v, sigma, g_worig: torch.Tensor):
    rows = w_sn.shape[0]
    cols = w_sn.numel() // rows
    assert g_wsn.numel() == w_sn.numel() == g_worig.numel() and g_worig.is_contiguous()
    L.check(L.lib().tr_spectral_norm_bwd(stream_ptr(), g_wsn.data_ptr(), w_sn.data_ptr(), u.data_ptr(), v.data_ptr(), sigma.data_ptr(),
                                         g_worig.data_ptr(), rows, cols, scratch(w_sn.device).data_ptr()), "tr_spectral_norm_bwd")


def batchnorm_workspace(c: int, device) -> torch.Tensor:
    return torch.empty(int(L.lib().tr_batchnorm_workspace_bytes(c)) // 8 + 2, dtype=torch.float64, device=device)


def batchnorm_fwd(x: L.View, y: L.View, npix: int, c: int, gamma: torch.Tensor, beta: torch.Tensor, running_mean: torch.Tensor | None, running_var: torch.Tensor | None,
                  save_mean: torch.Tensor | None, save_rstd: torch.Tensor | None, workspace: torch.Tensor, *, training: bool, momentum: float = 0.1, eps: float = 1e-5,
                  act: int = L.ACT_NONE, slope: float = 0.2):
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    L.check(L.lib().tr_batchnorm_fwd(stream_ptr(), x, y, npix, c, gamma.data_ptr(), beta.data_ptr(), p(running_mean), p(running_var), momentum, eps, 1 if training else 0,
                                     act, slope, p(save_mean), p(save_rstd), workspace.data_ptr(), workspace.numel() * workspace.element_size()), "tr_batchnorm_fwd")


def batchnorm_bwd(gy: L.View, x: L.View, y: L.View, dx: L.View, npix: int, c: int, gamma: torch.Tensor, save_mean: torch.Tensor, save_rstd: torch.Tensor,
                  dgamma: torch.Tensor | None, dbeta: torch.Tensor | None, workspace: torch.Tensor, *, slope: float = 0.2, accumulate: bool = True):
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    L.check(L.lib().tr_batchnorm_bwd(stream_ptr(), gy, x, y, dx, npix, c, gamma.data_ptr(), save_mean.data_ptr(), save_rstd.data_ptr(), slope, p(dgamma), p(dbeta),
                                     1 if accumulate else 0, workspace.data_ptr(), workspace.numel() * workspace.element_size()), "tr_batchnorm_bwd")


def gan_loss(gan_type: str, logits: torch.Tensor, target_is_real: bool, is_disc: bool, value_out: torch.Tensor, *, value_weight=1.0,
             total_out=None, total_weight=0.0, grad: torch.Tensor | None = None, grad_weight=1.0, real_label=1.0, fake_label=0.0):
    _need_cuda(logits, value_out, grad, total_out)
    assert logits.dtype == torch.float32 and logits.is_contiguous()
    d = L.GanLossDesc()
    d.gan_type, d.target_is_real, d.is_disc = L.GAN_TYPES[gan_type], int(bool(target_is_real)), int(bool(is_disc))
    d.logits, d.numel = logits.data_ptr(), logits.numel()
    d.real_label, d.fake_label = real_label, fake_label
    d.value_weight, d.total_weight, d.grad_weight = value_weight, total_weight, grad_weight
    d.value_out = value_out.data_ptr()
    d.total_out = total_out.data_ptr() if total_out is not None else None
    d.grad = grad.data_ptr() if grad is not None else None
    d.partials = scratch(logits.device).data_ptr()
    L.check(L.lib().tr_gan_loss(stream_ptr(), C.byref(d)), "tr_gan_loss")


def sumpool2x2(inp: L.View, out: L.View, mask: L.View, mask_slope: float, n, h_in, w_in):
    L.check(L.lib().tr_sumpool2x2(stream_ptr(), inp, out, mask, mask_slope, n, h_in, w_in), "tr_sumpool2x2")


def maxpool2x2(inp: L.View, out: L.View, n, h_in, w_in, relu_first: bool):
    L.check(L.lib().tr_maxpool2x2(stream_ptr(), inp, out, n, h_in, w_in, 1 if relu_first else 0), "tr_maxpool2x2")


def maxpool2x2_bwd(inp: L.View, gout: L.View, gin: L.View, n, h_in, w_in, relu_first: bool):
    L.check(L.lib().tr_maxpool2x2_bwd(stream_ptr(), inp, gout, gin, n, h_in, w_in, 1 if relu_first else 0), "tr_maxpool2x2_bwd")


_partials: dict = {}


def scratch(device, nfloats: int = 4096) -> torch.Tensor:
    key = (str(device), stream_ptr())
    t = _partials.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.empty(max(nfloats, 4096), dtype=torch.float32, device=device)
        _partials[key] = t
    return t


def pixel_loss(kind: int, pred: torch.Tensor, target: torch.Tensor, value_out: torch.Tensor, *, value_weight: float = 1.0,
               total_out: torch.Tensor | None = None, total_weight: float = 0.0, grad: torch.Tensor | None = None,
               grad_weight: float | None = None, accumulate_grad: bool = False, eps: float = 1e-12, pool_gout: torch.Tensor | None = None):
    """value_out[0] = value_weight*mean ; total_out[0] += total_weight*mean ; grad (+)= grad_weight * dmean/dpred.
    pool_gout (n x h/2 x w/2 x c, fp32): pred (NHWC fp32) feeds max_pool2d(relu(.), 2, 2) and pool_gout is the gradient of the pool's output:
    grad = grad_weight * dmean/dpred + max_pool2d_backward(pool_gout) through the ReLU, written once (tr_pixel_loss_pool_bwd)."""
    _need_cuda(pred, target, value_out, grad, total_out)
    assert pred.shape == target.shape and pred.dtype == target.dtype and pred.is_contiguous() and target.is_contiguous()
    assert value_out.dtype == torch.float32 and (total_out is None or total_out.dtype == torch.float32)
    d = L.PixelLossDesc()
    d.kind, d.dtype = kind, dt_code(pred.dtype)
    d.pred, d.target, d.numel = pred.data_ptr(), target.data_ptr(), pred.numel()
    d.eps, d.value_weight, d.total_weight = eps, value_weight, total_weight
    d.grad_weight = value_weight if grad_weight is None else grad_weight
    d.value_out = value_out.data_ptr()
    d.total_out = total_out.data_ptr() if total_out is not None else None
    if grad is not None:
        assert grad.is_contiguous() and grad.numel() == pred.numel()
        d.grad, d.grad_dtype, d.accumulate_grad = grad.data_ptr(), dt_code(grad.dtype), 1 if accumulate_grad else 0
    d.partials = scratch(pred.device).data_ptr()
    if pool_gout is not None:
        n, h, w, c = pred.shape
        assert grad is not None and pool_gout.is_contiguous() and pool_gout.dtype == torch.float32 and tuple(pool_gout.shape) == (n, h // 2, w // 2, c)
        L.check(L.lib().tr_pixel_loss_pool_bwd(stream_ptr(), C.byref(d), pool_gout.data_ptr(), n, h, w, c), "tr_pixel_loss_pool_bwd")
        return
    L.check(L.lib().tr_pixel_loss(stream_ptr(), C.byref(d)), "tr_pixel_loss")


def sumsq_norm(x: torch.Tensor, out: torch.Tensor):
    _need_cuda(x, out)
    assert x.dtype == torch.float32 and x.is_contiguous()
    L.check(L.lib().tr_sumsq(stream_ptr(), x.data_ptr(), x.numel(), out.data_ptr(), scratch(x.device).data_ptr()), "tr_sumsq")


def adamw_ema(p, g, m, v, ema, *, lr, beta1, beta2, eps, weight_decay, step: int, ema_decay: float = 1.0, clip: float = 0.0,
              grad_norm: torch.Tensor | None = None, zero_grad=True, inv_scale: float = 1.0):
    _need_cuda(p, g, m, v, ema)
    d = L.AdamWDesc()
    d.p, d.g, d.m, d.v = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
    d.ema = ema.data_ptr() if ema is not None else None
    d.numel = p.numel()
    d.lr, d.beta1, d.beta2, d.eps, d.weight_decay = lr, beta1, beta2, eps, weight_decay
    d.bias_corr1 = 1.0 - beta1 ** step
    d.bias_corr2 = 1.0 - beta2 ** step
    d.ema_decay = ema_decay
    d.clip = clip
    d.grad_norm = grad_norm.data_ptr() if grad_norm is not None else None
    d.zero_grad = 1 if zero_grad else 0
    d.inv_scale = inv_scale
    L.check(L.lib().tr_adamw_ema(stream_ptr(), C.byref(d)), "tr_adamw_ema")


def color601(src: torch.Tensor, mode: int) -> torch.Tensor:
    """BT.601 conversion of an fp32 (N, 3, H, W) batch: 0 rgb -> ycbcr, 1 ycbcr -> rgb, 2 adjoint of 1 (utils/color_util.py:275-324)."""
    _need_cuda(src)
    src = src.contiguous().float()
    assert src.dim() == 4 and src.shape[1] == 3, "BT.601 conversions need 3-channel batches"
    dst = torch.empty_like(src)
    L.check(L.lib().tr_color601(stream_ptr(), src.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[2] * src.shape[3], mode), "tr_color601")
    return dst
