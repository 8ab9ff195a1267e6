/*
 * trainner_hip.h -- C ABI of libtrainner_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the SRModel.optimize_parameters hot path of traiNNer-redux
 * (reference: traiNNer/models/sr_model.py:452-828).  The reference is 100 % Python and has
 * no FFI of its own: its plugin surface is ARCH_REGISTRY / LOSS_REGISTRY
 * factories returning torch.nn.Module.  The entry points below are therefore what a ctypes
 * binding inside those factories binds (INTEGRATION.md shows the stub); each one cites the
 * reference arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); nothing allocates,
 *     synchronises or touches the host: every entry point is hipGraph-capturable;
 *   - activations are NHWC ("channels last", the layout the reference selects with
 *     use_channels_last, sr_model.py:50-54); a tr_view addresses a channel slice
 *     [coff, coff+c) of an NHWC buffer whose pixel record holds `cstride` channels, which is
 *     how dense-block concatenation (torch.cat in ResidualDenseBlock, reference twin
 *     traiNNer/archs/scunet_aaf6aa_arch.py:328-334) becomes free;
 *   - return value: 0 on success, a hipError_t (>0) from the launch, or a negative TR_E* code
 *     for a rejected argument.  tr_last_error() returns a message for the calling thread.
 *     Messages never contain the words "allocate" or "CUDA": the reference's training loop
 *     treats such RuntimeErrors as out-of-memory and retries (train.py:757-849).
 */
#ifndef TRAINNER_HIP_H
#define TRAINNER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TR_ABI_VERSION 4   /* 4: tr_layernorm_bwd takes dx2 / sample_scale / pix_per_sample; 3: tr_batchnorm_fwd / _bwd; 2: tr_tune_set / tr_tune_get; tr_adamw_ema skips the update on the device when *grad_norm is not finite */

enum { TR_F32 = 0, TR_BF16 = 1, TR_F16 = 2, TR_BF16X3 = 3 };   /* TR_F16: the reference's default AMP dtype (use_amp without amp_bf16) */
enum { TR_EINVAL = -1, TR_EUNSUPPORTED = -2, TR_EALIGN = -3 };
enum { TR_ACT_NONE = 0, TR_ACT_LRELU = 1, TR_ACT_RELU = 2, TR_ACT_GELU = 3, TR_ACT_SILU = 4, TR_ACT_MISH = 5 };
enum { TR_PAD_ZEROS = 0, TR_PAD_REPLICATE = 1 };
enum { TR_IN_DIRECT = 0, TR_IN_NEAREST2X = 1, TR_IN_FULL = 2, TR_IN_VALID = 3 };
enum { TR_LOSS_L1 = 0, TR_LOSS_MSE = 1, TR_LOSS_CHARBONNIER = 2 };
enum { TR_GAN_VANILLA = 0, TR_GAN_LSGAN = 1, TR_GAN_WGAN = 2, TR_GAN_WGAN_SOFTPLUS = 3, TR_GAN_HINGE = 4 };

/* channel-slice view of an NHWC buffer (ptr == NULL means "absent") */
typedef struct tr_view {
    void*   ptr;      /* base of the NHWC buffer                                  */
    int32_t dtype;    /* TR_F32 | TR_BF16 | TR_F16                                 */
    int32_t cstride;  /* channels per pixel record of the underlying buffer        */
    int32_t coff;     /* first channel of the view                                 */
    int32_t c;        /* channels in the view                                      */
} tr_view;

int         tr_abi_version(void);
const char* tr_last_error(void);
/* Kernel-selection switches (no reference counterpart: cuDNN picks its algorithms behind torch.backends.cudnn.benchmark, train.py:276-283).
 * Keys: "conv_impl" (0 automatic | 1 register-staged kernels | 2 LDS-DMA ring kernel whenever eligible), "ring_cfg" (-1 automatic | 33 / 32 / 31 |
 * 0), "ring_generic" (0 | 1), "wgrad_impl" (0 | 1 | 2), "wgrad_multi" (1 | 0), "wgrad_reduce" (0 16-byte reads | 1 4-byte form), "dgrad_image_ref" (0 | 1), "linear_impl" (0 LDS-DMA ring | 1 register-staged token GEMM), "linear_epilogue" (0 straight-line forms | 1 generic); read-only records of the last
 * tr_conv3x3 launch: "last_conv_family" (1 register-staged, 2 LDS-DMA two-stage, 3 token GEMM, 4 ring, 5 token GEMM with LDS-DMA staging), "last_conv_instance", and the counter "ring_launches".  Initial values
 * come from the TRAINNER_* environment variables once, at load time; launches never read the environment.  Process-wide, not thread-safe
 * against concurrent launches (the step is driven by one host thread, SURVEY.md section 8b "Threading").
 * tr_tune_set returns TR_EINVAL for an unknown key; tr_tune_get returns -1 and sets tr_last_error(). */
int         tr_tune_set(const char* key, long long value);
long long   tr_tune_get(const char* key);
/* number of bytes of packed weights for a (cin, cout) 3x3 conv in `compute` dtype */
size_t      tr_conv3x3_packed_bytes(int cin, int cout, int compute);
/* same for a ksize x ksize kernel (ksize = 1: token GEMMs, i.e. nn.Linear over an NHWC token grid) */
size_t      tr_conv_packed_bytes(int cin, int cout, int compute, int ksize);

/* ---- 3x3 stride-1 pad-1 convolution as implicit GEMM on MFMA ------------------------------
 * Replaces torch.nn.Conv2d(k=3,s=1,p=1) + the pointwise ops the reference applies around it:
 *   RDB conv1-4 + LeakyReLU(0.2)                 scunet_aaf6aa_arch.py:321-333
 *   RDB conv5 `x5*0.2 + x`, RRDB `out*0.2 + x`   scunet_aaf6aa_arch.py:334-336,355-360
 *   nn.Upsample(nearest x2) -> conv -> LeakyReLU scunet_aaf6aa_arch.py:371-377
 *   VGG19 conv + ReLU, replicate pad on conv1_1  losses/perceptual_fp16_loss.py:262-274,321-329
 * and, with flipped/transposed packed weights, the input-gradient of the same convs
 * (what autograd's conv backward computes at sr_model.py:681).
 *
 *   v  = conv(in) + bias ; v = act(v)
 *   v  = v*scale1 + res1_scale*res1 ; v = v*scale2 + res2_scale*res2   (each only if resN.ptr != NULL)
 *   v *= (mask > 0 ? 1 : mask_slope)                      (only if mask.ptr  != NULL)
 *   out = (out.dtype) v ; out2 = (out2.dtype)(v*out2_scale)   (out2 optional)
 *
 * compute = TR_BF16 / TR_F16: in.dtype must equal it, 16-bit operands, fp32 accumulate
 *           (v_mfma_f32_32x32x16_bf16 / _f16); compute = TR_BF16X3 (3x3 only, fp32 views): the reference's `fast_matmul` trade
 *           ("Trade precision for performance", redux_options.py:479-481; TF32 on its CUDA targets) done the gfx950 way, which has no
 *           TF32: every fp32 operand is split on the fly into hi + lo bf16 and each product is 3 bf16 MFMAs (hi*hi + hi*lo + lo*hi),
 *           ~2^-17 relative per product -- between TF32 and fp32; compute = TR_F32: in.dtype TR_F32, exact fp32
 *           (v_mfma_f32_32x32x2_f32).  Epilogue arithmetic is always fp32.
 */
typedef struct tr_conv3x3_desc {
    int32_t n, h, w;          /* OUTPUT batch / height / width                                */
    int32_t cin, cout;
    int32_t compute;          /* TR_F32 | TR_BF16 | TR_F16                                    */
    int32_t in_mode;          /* TR_IN_DIRECT (input n x h x w) | TR_IN_NEAREST2X (n x h/2 x w/2) |
                                 TR_IN_FULL (input n x (h-2) x (w-2): 'full' correlation, out(y,x) = sum in(y+ky-2, x+kx-2) W,
                                 i.e. the input gradient of a conv over an explicitly padded image) |
                                 TR_IN_VALID (input n x (h+2) x (w+2), no padding: out(y,x) = sum in(y+ky, x+kx) W for ksize 3,
                                 in(y+1, x+1) for ksize 1 -- the padding-free 3x3 of SPAN's Conv3XC, span_arch.py:134-141,199-204) */
    int32_t pad_mode;         /* TR_PAD_ZEROS | TR_PAD_REPLICATE                              */
    int32_t act;              /* TR_ACT_*                                                     */
    float   slope;            /* LeakyReLU negative slope                                     */
    float   scale1, scale2;
    float   res1_scale, res2_scale;
    float   mask_slope;
    float   out2_scale;
    tr_view in, out, out2, res1, res2, mask;
    const void*  wpacked;     /* from tr_pack_conv3x3 (layout documented in DESIGN.md)        */
    const float* bias;        /* cout floats or NULL                                          */
    int32_t ksize;            /* 3 (3x3, pad 1) or 1 (1x1: nn.Linear over tokens laid out as an n x h x w grid,
                                 traiNNer/archs/drct_arch.py:64-66,166-168)                     */
    int32_t mask_mode;        /* 0: LeakyReLU/ReLU derivative by sign ; 1: exact-GELU derivative of `mask` ; 2: SiLU derivative ; 3: Mish derivative */
    const float* sample_scale;/* optional n floats multiplied into v per image before res1 (DropPath keep mask,
                                 drct_arch.py:572-573); with act = TR_ACT_GELU / TR_ACT_SILU / TR_ACT_MISH, out2 receives the pre-activation */
} tr_conv3x3_desc;

int tr_conv3x3(void* stream, const tr_conv3x3_desc* d);

/* The same convolution (3x3, pad 1, zero padding, fp32 views, cin % 8 == 0, cout % 64 == 0, even h and w; bias, ReLU and the sign mask of the
 * epilogue only) as a fused Winograd F(2x2, 3x3) kernel in exact fp32 arithmetic: 2.25x fewer MFMA cycles than the direct form for the
 * fp32-forced VGG19 perceptual head (losses/perceptual_loss.py:8-12, perceptual_fp16_loss.py:262-329).  `wpacked` comes from
 * tr_pack_winograd (U = G g G^T per (cout, cin) pair; transpose != 0 packs the input-gradient conv of an OIHW weight, i.e. cout / cin name
 * the channels of THIS launch's output / input).  d->compute is ignored.  Differs from tr_conv3x3 by fp32 rounding only (~1e-6). */
size_t tr_winograd_packed_bytes(int cin, int cout);
int    tr_pack_winograd(void* stream, const float* w_oihw, float* dst, int cout, int cin, int transpose);
int    tr_conv3x3_winograd_f32(void* stream, const tr_conv3x3_desc* d);
/* The same contract as a fused Winograd F(4x4, 3x3) kernel (h and w multiples of 4): 0.5625 of the MFMA cycles of the F(2x2, 3x3) form, every
 * operation still an fp32 add / multiply / MFMA; differs from tr_conv3x3 by fp32 rounding (~2e-6 relative per layer; F(2x2): ~4e-7).
 * `wpacked` comes from tr_pack_winograd43 (U = G g G^T, 6 x 6 per (cout, cin) pair, in the kernel's fragment order).
 * `out2` of THIS entry point, if set, receives max_pool2d(relu(out), 2, 2) at half resolution (fp32 view of cout channels, out2_scale 1,
 * no mask): VGG19's conv -> ReLU -> MaxPool pairs (perceptual_fp16_loss.py:12-50) without reading the activation back. */
size_t tr_winograd43_packed_bytes(int cin, int cout);
int    tr_pack_winograd43(void* stream, const float* w_oihw, float* dst, int cout, int cin, int transpose);
int    tr_conv3x3_winograd43_f32(void* stream, const tr_conv3x3_desc* d);

/* Gradient of a <=4-channel stem conv (VGG19 conv1_1, perceptual_fp16_loss.py:272-274) with respect to the NCHW fp32
 * image: conv_backward_input + adjoint of zero/replicate padding + per-channel scale (1/std of the input
 * normalisation, host pointer, may be NULL); dst (+)= result.  Weights are the fp32 OIHW master tensor. */
int tr_conv3x3_dgrad_image(void* stream, tr_view gy, const float* w_oihw, int cout, int cin, int n, int h, int w, int pad_mode,
                           const float* ch_scale_host, float* dst_nchw, int accumulate);

/* Stem convolution straight from the NCHW fp32 image (the DataFeed contract, utils/types.py:10-17): (x - mean[c]) / std[c]
 * (perceptual_fp16_loss.py:322; host pointers, may be NULL), zero or replicate padding (:272-274), 3x3 conv with cin <= 3 input channels,
 * bias, optional ReLU -> NHWC fp32 view of cout channels.  VGG19 conv1_1 of the perceptual head.  Weights / bias: fp32 OIHW master tensors. */
int tr_conv3x3_stem(void* stream, const float* src_nchw, const float* w_oihw, const float* bias, const float* mean_host, const float* std_host,
                    int cin, int cout, int n, int h, int w, int pad_mode, int act, tr_view out);

/* Adjoint of F.pad(mode="replicate", pad=1) + NHWC->NCHW + per-channel scale, for the stem of the perceptual head
 * (perceptual_fp16_loss.py:272-274,322): gp is the gradient w.r.t. the PADDED image, NHWC n x (h+2) x (w+2);
 * dst[n,c,y,x] (+)= ch_scale[c] * sum of gp over every padded position that replicate-padding filled from (y,x). */
int tr_fold_replicate_nchw(void* stream, tr_view gp, float* dst_nchw, int n, int c, int h, int w, const float* ch_scale_host,
                           int accumulate);

/* ---- weight packing -------------------------------------------------------------------------
 * Master weights stay fp32 OIHW (the state-dict layout the reference saves and spandrel loads,
 * base_model.py:395-482).  A segment copies the sub-block src[o_off:o_off+o_cnt, i_off:i_off+i_cnt]
 * of an OIHW tensor into a packed conv operand as rows [dst_o_off, +o_cnt) / columns
 * [dst_i_off, +i_cnt); `transpose` != 0 swaps the roles of O and I and flips the taps
 * (the weights of the input-gradient conv).  Segments live in device memory so one launch
 * re-packs a whole network after an optimizer step.
 */
typedef struct tr_pack_seg {
    const float* src;         /* OIHW fp32, dims src_o x src_i x 3 x 3                        */
    void*        dst;         /* packed operand of a conv with dims (dst_cin, dst_cout)       */
    int32_t src_o, src_i;
    int32_t o_off, o_cnt, i_off, i_cnt;
    int32_t dst_cin, dst_cout;
    int32_t dst_o_off, dst_i_off;   /* in the packed conv's (cout, cin) index space           */
    int32_t transpose;
    int32_t compute;          /* TR_F32 | TR_BF16 | TR_F16                                    */
    int32_t ksize;            /* 3 or 1: src is OI(ksize)(ksize)                              */
    int32_t reserved;
    int64_t first;            /* exclusive prefix sum of o_cnt*i_cnt*9 over segments (host-filled) */
} tr_pack_seg;

int tr_pack_conv3x3(void* stream, const tr_pack_seg* segs_dev, int nsegs, int64_t total_elems);

/* ---- weight gradient of the 3x3 conv --------------------------------------------------------
 * dW[co][ci][ky][kx] (+)= sum_{n,y,x} gy[n,y,x,co] * in[n,y+ky-1,x+kx-1,ci]   (OIHW fp32)
 * db[co] (+)= sum gy[...,co]            -- what autograd accumulates into .grad at sr_model.py:681.
 * `partials` is a caller-provided fp32 workspace of tr_wgrad3x3_workspace_bytes(); the result is
 * bitwise reproducible (no float atomics).  accumulate != 0 adds into dw/db.
 */
typedef struct tr_wgrad3x3_desc {
    int32_t n, h, w;          /* spatial size of gy (= conv output size)                      */
    int32_t cin, cout;
    int32_t compute;
    int32_t in_mode, pad_mode;  /* in_mode as in tr_conv3x3_desc: TR_IN_FULL reads in(y+ky-2, x+kx-2) of an (h-2) x (w-2) input (zero outside),
                                   TR_IN_VALID reads in(y+ky, x+kx) of an (h+2) x (w+2) input; for ksize 1 the centre tap only */
    int32_t accumulate;
    tr_view in, gy;
    float*  dw;               /* cout x cin x 3 x 3                                           */
    float*  db;               /* cout or NULL                                                 */
    float*  partials;
    size_t  partials_bytes;
    int32_t ksize;            /* 3 or 1 (dw is then cout x cin, the nn.Linear weight gradient) */
    int32_t dw_ld;            /* 0, or the input-channel count of the tensor `dw` points INTO when this launch fills the column block
                                 [ci0, ci0 + cin) of a wider weight (dw = base + ci0 * ksize^2): torch.cat'ed inputs whose slices are
                                 stored apart (span_arch.py:319-320) */
} tr_wgrad3x3_desc;

size_t tr_wgrad3x3_workspace_bytes(int n, int h, int w, int cin, int cout);
size_t tr_wgrad_workspace_bytes(int n, int h, int w, int cin, int cout, int ksize);
int    tr_wgrad3x3(void* stream, const tr_wgrad3x3_desc* d);
/* The weight / bias gradients of several 3x3 convs that read channel slices of ONE input buffer and ONE output-gradient buffer (RRDBNet's
 * ResidualDenseBlock_5C, spandrel ESRGAN: conv k sees cat(x, x1..x_{k-1}); autograd's five conv_backward_weight calls of sr_model.py:681) in one
 * launch, so that the shared bytes are fetched once; every descriptor carries its own dw / db / partials.  Same results, bit for bit per conv,
 * as ndesc calls of tr_wgrad3x3 with the same partial slot count; falls back to exactly those calls for sets it cannot fuse. */
int    tr_wgrad3x3_multi(void* stream, const tr_wgrad3x3_desc* descs, int ndesc);

/* ---- layout / pointwise ----------------------------------------------------------------------
 * NCHW fp32 image batch (the DataFeed contract, utils/types.py:10-17) -> NHWC view, optionally
 * (x - mean[c]) / std[c] (perceptual_fp16_loss.py:322) ; channels of the view beyond c_src are zeroed.
 */
int tr_nchw_to_nhwc(void* stream, const float* src, int n, int c_src, int h, int w,
                    tr_view dst, const float* mean3, const float* std3);
/* NHWC view -> NCHW fp32 (network output handed back to the trainer, sr_model.py:478-479) */
int tr_nhwc_to_nchw(void* stream, tr_view src, float* dst, int n, int c_dst, int h, int w);
/* gradient of tr_nchw_to_nhwc wrt src: dst_nchw[c] = g[c] / std[c] */
int tr_nhwc_to_nchw_scaled(void* stream, tr_view src, float* dst, int n, int c_dst, int h, int w,
                           const float* inv_std3, int accumulate);
/* dst[c] = src[c] * scale[c] + offset[c]  (SwinIR's `x / img_range + mean` on the way out; host pointers, 3 floats each) */
int tr_nhwc_to_nchw_affine(void* stream, tr_view src, float* dst, int n, int c_dst, int h, int w, const float* scale3_host,
                           const float* offset3_host);
/* out = (out.dtype)((a*sa*s[p / pix_per_sample] + b*sb) * (mask > 0 ? 1 : mask_slope)) for NHWC views of equal shape
 * (b, mask and the per-sample scale s optional) -- residual / skip adds of activations and gradient streams, dtype conversion,
 * LeakyReLU derivative of a skip branch, DropPath keep mask on a gradient */
int tr_axpby(void* stream, tr_view a, float sa, tr_view b, float sb, tr_view out, int64_t npix, tr_view mask, float mask_slope,
             const float* sample_scale, int64_t pix_per_sample);
/* adjoint of nn.Upsample(scale_factor=2, mode="nearest") (scunet_aaf6aa_arch.py:371): 2x2 sum pooling,
 * optionally times the LeakyReLU derivative taken from `mask` (the saved activation at low resolution) */
int tr_sumpool2x2(void* stream, tr_view in, tr_view out, tr_view mask, float mask_slope, int n, int h_in, int w_in);
/* 2x2/2 max pooling (torchvision VGG19 features 'M', perceptual_fp16_loss.py:12-50) with fused
 * preceding ReLU: out = maxpool(relu?(in)); backward routes to the first maximal element like ATen. */
int tr_maxpool2x2(void* stream, tr_view in, tr_view out, int n, int h_in, int w_in, int relu_first);
int tr_maxpool2x2_bwd(void* stream, tr_view in, tr_view gout, tr_view gin, int n, int h_in, int w_in,
                      int relu_first);

/* ---- pixel-loss heads -------------------------------------------------------------------------
 * l1 / mse / charbonnier with reduction 'mean' (basic_loss.py:16-28, loss_util.py:8-59) over flat fp32 or
 * bf16 arrays (NCHW image batches and NHWC feature maps alike), fused with the bookkeeping of
 * sr_model.py:639-663 so that one pass serves the logged value, the running total and the gradient:
 *   m = mean(elem(pred - target))
 *   value_out[0]  = value_weight * m                      (the weighted loss that is logged)
 *   total_out[0] += total_weight * m                      (l_g_total; optional)
 *   grad        (+)= grad_weight * d m / d pred           (optional; accumulate_grad selects += vs =)
 * pred and target are read once, grad written once; the scalar is reduced without float atomics
 * (fixed-order two-level sum) so it is bitwise reproducible.
 */
typedef struct tr_pixel_loss_desc {
    int32_t kind;             /* TR_LOSS_*                                   */
    int32_t dtype;            /* dtype of pred and target                    */
    int32_t grad_dtype;
    int32_t accumulate_grad;
    const void* pred;
    const void* target;
    int64_t numel;
    float   eps;              /* charbonnier                                 */
    float   value_weight;
    float   total_weight;
    float   grad_weight;
    float*  value_out;        /* 1 float                                     */
    float*  total_out;        /* 1 float or NULL                             */
    void*   grad;             /* numel elements of grad_dtype or NULL        */
    float*  partials;         /* >= 1024 floats of scratch                   */
} tr_pixel_loss_desc;
int tr_pixel_loss(void* stream, const tr_pixel_loss_desc* d);
/* The same criterion on the fp32 NHWC activation (n x h x w x c) that feeds F.max_pool2d(F.relu(.), 2, 2) (a tapped VGG19 conv in front of a
 * pool, perceptual_fp16_loss.py:12-50,262-329), fused with that pool's backward: grad = grad_weight/numel * dcrit(pred - target)
 * + max_pool2d_backward(gout) through the ReLU, written once (d->accumulate_grad is ignored; fp32 only).  gout: n x h/2 x w/2 x c. */
int tr_pixel_loss_pool_bwd(void* stream, const tr_pixel_loss_desc* d, const float* gout, int n, int h, int w, int c);

/* ---- U-Net discriminator pieces (archs/discriminator_arch.py:98-160) ---------------------------------------------------
 * 4x4 stride-2 pad-1 convs run as space-to-depth(2) + the 3x3 MFMA conv on expanded weights:
 *   s2d: out[n, y/2, x/2, ((y&1)*2 + (x&1))*C + c] = in[n, y, x, c]          (h, w = full-resolution dims, even)
 *   w3[co][(dy,dx,ci)][ky][kx] = w4[co][ci][2ky+dy-1][2kx+dx-1] or 0        (tr_expand_w4x4s2 / adjoint tr_collapse_g3x3s2d) */
int tr_space_to_depth2(void* stream, tr_view in, tr_view out, int n, int h, int w);
int tr_depth_to_space2(void* stream, tr_view in, tr_view out, int n, int h, int w);
int tr_expand_w4x4s2(void* stream, const float* w4_oihw, float* w3_oihw, int cout, int cin);
int tr_collapse_g3x3s2d(void* stream, const float* g3_oihw, float* g4_oihw, int cout, int cin, int accumulate);
/* F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) (:139,144,149) and its adjoint */
int tr_bilinear_up2x(void* stream, tr_view in, tr_view out, int n, int h_in, int w_in);
int tr_bilinear_up2x_bwd(void* stream, tr_view gout, tr_view gin, int n, int h_in, int w_in);
/* torch.nn.utils.spectral_norm, old-style hook (:115-128): W is rows x cols (= cout x cin*kh*kw); with power_iteration != 0
 * v <- normalize(W^T u), u <- normalize(W v) in place (eps 1e-12), then sigma = u^T W v, w_out = W / sigma.
 * scratch: rows + cols floats.  Backward: g_worig += (G - <G, W/sigma> u v^T) / sigma; scratch: 256 floats. */
int tr_spectral_norm(void* stream, const float* w_orig, float* u, float* v, float* w_out, float* sigma_out, int rows, int cols,
                     int power_iteration, float eps, float* scratch);
int tr_spectral_norm_bwd(void* stream, const float* g_wsn, const float* w_sn, const float* u, const float* v, const float* sigma,
                         float* g_worig, int rows, int cols, float* scratch);

/* ---- GAN loss head (losses/gan_loss.py:13-126), mean over fp32 logits; same value / total / grad contract as tr_pixel_loss */
typedef struct tr_gan_loss_desc {
    int32_t gan_type;         /* TR_GAN_*                                    */
    int32_t target_is_real;
    int32_t is_disc;
    int32_t reserved;
    const float* logits;
    int64_t numel;
    float   real_label, fake_label;
    float   value_weight, total_weight, grad_weight;
    float*  value_out;
    float*  total_out;        /* or NULL                                     */
    float*  grad;             /* numel floats or NULL: d(grad_weight*loss)/dlogits (overwritten) */
    float*  partials;         /* >= 1024 floats                              */
} tr_gan_loss_desc;
int tr_gan_loss(void* stream, const tr_gan_loss_desc* d);

/* ---- SwinIR token path (in-tree twin of spandrel's SwinIR: traiNNer/archs/drct_arch.py) ---------------------------------
 * Tokens are the pixels of an NHWC tensor ([ntok][channels]); nn.Linear layers are tr_conv3x3 with ksize 1.
 * LayerNorm over the c logical channels of x (:464,476,658-662), eps as given; channels [c, y.c) of y are zeroed (K padding of
 * the following GEMM); mean / rstd (ntok floats each, optional) are kept for the backward. */
int tr_layernorm(void* stream, tr_view x, tr_view y, int64_t ntok, int c, const float* gamma, const float* beta, float eps,
                 float* mean_out, float* rstd_out);
/* dx = (res ? res : 0) + dLN/dx ; dgamma / dbeta (+)= ...  (fixed-order reduction; workspace from the _bytes call).
 * dx2 (optional view, any dtype): dx2 = dx * (sample_scale ? sample_scale[token / pix_per_sample] : 1) -- the compute-dtype copy of the residual-stream
 * gradient, scaled by the next branch's DropPath keep mask, that the following weight-gradient / input-gradient GEMMs read (drct_arch.py:572-573 backward);
 * it saves the separate scale-and-convert pass over the stream. */
size_t tr_layernorm_bwd_workspace_bytes(int c);
int tr_layernorm_bwd(void* stream, tr_view g, tr_view x, int64_t ntok, int c, const float* gamma, const float* mean, const float* rstd,
                     tr_view res, tr_view dx, float* dgamma, float* dbeta, int accumulate, float* partials, size_t partials_bytes,
                     tr_view dx2, const float* sample_scale, int64_t pix_per_sample);

/* (Shifted-)window multi-head self-attention core (:175-223, roll / partition / reverse :78-109,:532-568, mask :492-521):
 * per 8x8 window and head  softmax(q*scale @ k^T + table[rel_index] + shift_mask) @ v, windows taken from the image rolled by
 * -shift; qkv holds [q | k | v], each c = nh*hd channels; out receives the attention output at the ORIGINAL token positions.
 * Backward recomputes the probabilities; dqkv receives d/dq, d/dk, d/dv; dtable (+)= d/d(relative_position_bias_table). */
typedef struct tr_window_attn_desc {
    tr_view qkv, out;         /* forward operands (out unused in backward)                           */
    tr_view dout, dqkv;       /* backward operands                                                   */
    const float* table;       /* (2*window-1)^2 x nh fp32                                            */
    float*  dtable;           /* backward: gradient of the table or NULL                             */
    float*  dbias_ws;         /* backward: tr_window_attention_bwd_workspace_bytes() bytes, needed with dtable */
    int32_t b, h, w;          /* batch, token-grid height / width (multiples of window)              */
    int32_t c, nh;            /* embedding width and head count (c / nh <= 32)                        */
    int32_t window, shift;    /* 8 ; 0 or window/2                                                   */
    float   scale;            /* head_dim ** -0.5                                                    */
    int32_t accumulate_dtable;
} tr_window_attn_desc;
int    tr_window_attention(void* stream, const tr_window_attn_desc* d);
size_t tr_window_attention_bwd_workspace_bytes(int b, int h, int w, int nh);
int    tr_window_attention_bwd(void* stream, const tr_window_attn_desc* d);
/* nn.PixelShuffle(r) on NHWC views (:714-717): out[n, y*r+dy, x*r+dx, ch] = in[n, y, x, ch*r*r + dy*r + dx]; inverse != 0 is
 * pixel_unshuffle (its adjoint).  n, h, w are the LOW-resolution dims, c the high-resolution channel count. */
int tr_pixel_shuffle(void* stream, tr_view in, tr_view out, int n, int h, int w, int c, int r, int inverse);

/* ---- nn.BatchNorm2d(affine=True) + activation on NHWC views (VGGStyleDiscriminator: conv -> BatchNorm2d -> LeakyReLU(0.2), discriminator_arch.py:31-57,70-92)
 * training != 0: per-channel statistics of THIS batch over the npix = N*H*W pixel records (biased variance normalises; running_mean / running_var, if given,
 *   move by `momentum` towards the batch mean / UNBIASED variance, torch's rule); save_mean / save_rstd (c floats each) keep what backward needs;
 * training == 0: the running statistics normalise.  y = act(gamma * (x - mean) * rstd + beta), act in {TR_ACT_NONE, TR_ACT_LRELU, TR_ACT_RELU}.
 * Backward: gy is the gradient at y; with y given, the activation's derivative is taken from its sign (slope for y <= 0).  dx, and dgamma / dbeta
 * (+)= when accumulate != 0.  Reductions run in a fixed order (double-precision fold of per-block fp32 partials): reproducible bit for bit.
 * workspace: tr_batchnorm_workspace_bytes(c) bytes, 16-byte aligned; c a multiple of 4. */
size_t tr_batchnorm_workspace_bytes(int c);
int tr_batchnorm_fwd(void* stream, tr_view x, tr_view y, int64_t npix, int c, const float* gamma, const float* beta, float* running_mean, float* running_var,
                     float momentum, float eps, int training, int act, float slope, float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes);
int tr_batchnorm_bwd(void* stream, tr_view gy, tr_view x, tr_view y, tr_view dx, int64_t npix, int c, const float* gamma, const float* save_mean,
                     const float* save_rstd, float slope, float* dgamma, float* dbeta, int accumulate, void* workspace, size_t workspace_bytes);

/* ---- optimizer sweep --------------------------------------------------------------------------
 * torch.optim.AdamW (optimizers/__init__.py:30-47, templates: lr 2e-4, betas (0.9, 0.99), wd 0)
 * + ema-pytorch lerp (sr_model.py:226-235, :826-828) + global-norm clipping (sr_model.py:696-711)
 * over ONE flat fp32 parameter arena (params / grads / exp_avg / exp_avg_sq / ema all numel long).
 *   tr_sumsq: partial sums of squares of grads -> norm_out[0] = ||g||_2   (grad_norm_g)
 *   tr_adamw_ema: p,m,v,ema updated in one pass; grad scaled by min(1, clip/(norm+1e-6)) when
 *   clip > 0 (clip_grad_norm_ semantics), using the device-resident norm (no host sync);
 *   grads are zeroed in the same pass when zero_grad != 0 (optimizer.zero_grad(), sr_model.py:719).
 *   When grad_norm is given and holds NaN / Inf the step is skipped on the device: p, m, v, ema stay untouched (the reference raises
 *   "Training failed: NaN/Inf found in loss." before backward, sr_model.py:665-666; the host raises the same error at its next read-back).
 */
int tr_sumsq(void* stream, const float* x, int64_t numel, float* norm_out, float* partials);
typedef struct tr_adamw_desc {
    float*  p; float* g; float* m; float* v; float* ema;   /* ema may be NULL */
    int64_t numel;
    float   lr, beta1, beta2, eps, weight_decay;
    float   bias_corr1, bias_corr2;     /* 1 - beta^t, host computed */
    float   ema_decay;                  /* decay for this step (1.0 = keep, 0.0 = copy) */
    float   clip;                       /* <= 0: no clipping */
    const float* grad_norm;             /* device scalar from tr_sumsq (needed if clip > 0) */
    int32_t zero_grad;
    float   inv_scale;                  /* 0 or 1: gradients as they are; otherwise g *= inv_scale first (GradScaler.unscale_, sr_model.py:688-690)
                                           and the clip coefficient uses grad_norm * inv_scale */
} tr_adamw_desc;
int tr_adamw_ema(void* stream, const tr_adamw_desc* d);

/* ---- SPAN parameter-free attention (span_arch.py:243-247): sim_att = sigmoid(o3) - 0.5 ; out = (o3 + x) * sim_att --------------
 * forward writes out (any dtype) and optionally a second copy out2; backward, given g = d/d(out):
 *   g_o3 = g * (sim_att + (o3 + x) * s * (1 - s)),  g_x = g * sim_att   (s = sigmoid(o3)); gx (+)= when accumulate_gx != 0. */
int tr_span_gate(void* stream, tr_view o3, tr_view x, tr_view out, tr_view out2, int64_t npix);
int tr_span_gate_bwd(void* stream, tr_view g, tr_view o3, tr_view x, tr_view g_o3, tr_view g_x, int accumulate_gx, int64_t npix);

/* ---- MoSR GatedCNNBlock pieces (spandrel MoSR -- absent wheel; in-tree twins of the block: archs/mosrv2_arch.py:88-126,
 * archs/moesr_arch.py:66-106; factory arguments archs/mosr_arch.py:23-74) ------------------------------------------------------------
 * tr_dwconv: nn.Conv2d(c, c, k, 1, k // 2, groups=c) on an NHWC channel slice; weight is the fp32 OIHW tensor (c x 1 x k x k), bias c floats
 *            or NULL; flip != 0 mirrors the taps (= the input gradient of the same conv; pass bias NULL).  k in {3, 5, 7}, c % 4 == 0.
 * tr_dwconv_wgrad: dw[c][ky][kx] (+)= sum gy[p][c] * in[p + (ky - k/2, kx - k/2)][c], db[c] (+)= sum gy[p][c]; fixed-order reduction
 *            through `partials` (tr_dwconv_wgrad_workspace_bytes), bitwise reproducible.
 * tr_gate_mish: out = mish(g) * cat(a, b)   (`self.act(g) * torch.cat((i, c), dim=1)`, mosrv2_arch.py:124), nn.Mish = x * tanh(softplus(x)).
 * tr_gate_mish_bwd: given d = d/d(out): dg = d * cat(a, b) * mish'(g) ; da | db = d * mish(g).
 * tr_act_grad: out = g * s[p / pix_per_sample] * act'(pre) for act = TR_ACT_MISH / TR_ACT_SILU (the gradient entering a conv whose output
 *            went through the activation, when no producing conv can carry the mask in its epilogue); s = optional n floats (DropPath keep mask). */
int    tr_dwconv(void* stream, tr_view in, tr_view out, const float* weight, const float* bias, int n, int h, int w, int c, int ksize, int flip);
size_t tr_dwconv_wgrad_workspace_bytes(int n, int h, int w, int c, int ksize);
int    tr_dwconv_wgrad(void* stream, tr_view in, tr_view gy, float* dw, float* db, int n, int h, int w, int c, int ksize, int accumulate,
                       float* partials, size_t partials_bytes);
int    tr_gate_mish(void* stream, tr_view g, tr_view a, tr_view b, tr_view out, int64_t npix);
int    tr_gate_mish_bwd(void* stream, tr_view d, tr_view g, tr_view a, tr_view b, tr_view dg, tr_view da, tr_view db, int64_t npix);
int    tr_act_grad(void* stream, int act, tr_view g, tr_view pre, tr_view out, int64_t npix, const float* sample_scale, int64_t pix_per_sample);

/* ---- feed: uint8 HWC image batch (what a decoder produces; 4x fewer PCIe bytes than the reference's fp32 CHW DataFeed,
 * paired_image_dataset.py:133-146) -> fp32 NCHW in [0,1], optionally (x/255 - mean[c]) / std[c] (torchvision normalize, :141-143) */
int tr_u8hwc_to_nchw(void* stream, const uint8_t* src, int n, int h, int w, int c, float* dst, const float* mean3, const float* std3);

/* ---- validation: metrics and tiled-inference blending (sr_model.py:830-899,960-1139; metrics/psnr_ssim.py) --------------------------
 * tr_metric_prepare: fp32 CHW image in [0,1] -> 8-bit levels (clamp, x255, round: tensor2img), border crop, optional BT.601 luma
 *                    (metric_util.py:34-47); planes = c' x (h-2crop) x (w-2crop) floats, c' = 1 with y_channel on 3-channel input.
 * tr_metric_sqerr / tr_metric_ssim: double-precision per-block partial sums (host adds them in order): sum (a-b)^2, and the sum of the
 *                    SSIM map (11x11 Gaussian window `gauss11`, valid region) over all channels.
 * tr_tile_blend / tr_tile_normalize: infer_tiled's triangular-weight accumulation and the final division. */
int tr_metric_prepare(void* stream, const float* img_chw, int c, int h, int w, int crop, int y_channel, float* planes);
int tr_metric_sqerr(void* stream, const float* a, const float* b, int64_t n, double* partials, int* n_partials);
int tr_metric_ssim(void* stream, const float* a, const float* b, int nc, int h, int w, const double* gauss11_host, double* partials, int* n_partials);
int tr_tile_blend(void* stream, const float* patch, int c, int patch_h, int patch_w, float* out, float* wmap, int out_h, int out_w, int y0, int x0,
                  int ph, int pw, int hr_tile);
int tr_tile_normalize(void* stream, const float* acc, const float* wmap, int c, int acc_h, int acc_w, float* out, int out_h, int out_w);

/* ---- OTF degradation primitives (first slice of SURVEY.md 8f-1) ------------------------------------------------------------------
 * tr_filter2d: utils/img_process_util.py:8-32 -- reflect pad k/2, cross-correlation of every channel of sample b with kernels[b]
 *              (per_sample != 0: kernels is n x k x k) or with the single kernel (per_sample == 0); fp32 NCHW in / out; even k is an error.
 * tr_quantize_u8: models/realesrgan_model.py:621 -- clamp(round(x * 255), 0, 255) / 255. */
int tr_filter2d(void* stream, const float* img_nchw, const float* kernels, int per_sample, int n, int c, int h, int w, int k, float* out_nchw);
int tr_quantize_u8(void* stream, const float* x, float* y, int64_t n);

/* ---- OTF degradation stages of RealESRGANModel.feed_data (models/realesrgan_model.py:456-664), fp32 NCHW planes ------------------------
 * tr_resample_axis: one 1-D pass of a separable resize: out[o] = sum_t weights[o][t] * in[idx[o][t]] along x (axis 1) or y (axis 0); idx / weights are
 *              device tables of out_len x taps entries built by the host for data/degradations.py:961-1021 `resize_pt` (F.interpolate bilinear /
 *              bicubic with antialias, nearest-exact, nearest; Lanczos-3 prefilter with reflect padding + bicubic) ; clamp01 != 0 applies the
 *              trailing .clamp(0, 1).
 * tr_crop_nchw: img[:, :, top:top+out_h, left:left+out_w] (paired_random_crop, data/transforms.py:118-137).
 * tr_gain_noise_clamp: clamp(x * gain[channel] + noise * noise_std, 0, 1): exposure / colour temperature / sensor noise
 *              (models/paragon_otf_degradations.py:345-414); gain_host = c floats on the host, noise optional (device, same shape).
 * tr_filter2d_zeropad: F.conv2d(img, one k x k kernel repeated per channel, padding = pad, groups = c): motion blur (:251-272) and the box blur of
 *              the oversharpening stage (realesrgan_model.py:215-232); output is (h + 2 pad - k + 1) x (w + 2 pad - k + 1).
 * tr_axpby_clamp01: clamp(a * wa + b * wb, 0, 1)  (img + (img - blurred) * strength, realesrgan_model.py:234-241). */
int tr_resample_axis(void* stream, const float* src, float* dst, int64_t planes, int in_h, int in_w, int axis, int out_len, const int* idx,
                     const float* weights, int taps, int clamp01);
int tr_crop_nchw(void* stream, const float* src, float* dst, int64_t planes, int in_h, int in_w, int top, int left, int out_h, int out_w);
int tr_gain_noise_clamp(void* stream, const float* src, const float* noise, float* dst, int n, int c, int64_t hw, const float* gain_host, float noise_std);
int tr_filter2d_zeropad(void* stream, const float* img_nchw, const float* kernel, int64_t planes, int h, int w, int k, int pad, float* out);
int tr_axpby_clamp01(void* stream, const float* a, float wa, const float* b, float wb, float* out, int64_t n);
/* tr_warp_bilinear: F.grid_sample(mode="bilinear", align_corners=False) over a grid computed in the kernel; params_host = 4 floats.
 *   mode 0 lens distortion (paragon_otf_degradations.py:297-343; params[0] = strength; reflection padding; the reference's swapped grid axes kept),
 *   mode 1 rolling shutter (:417-456; params[0] = slant = strength * h / w; reflection padding),
 *   mode 2 chromatic aberration (realesrgan_model.py:244-311; params[ch] = affine scale of channel ch, 1 = untouched; zero padding, clamp to [0, 1]). */
int tr_warp_bilinear(void* stream, const float* src, float* dst, int n, int c, int h, int w, int mode, const float* params_host);

/* tr_bayer_round_trip: the demosaicing-artifact stage (realesrgan_model.py:312-363 / paragon_otf_degradations.py:526-556) of an fp32 (N, 3, H, W)
 * batch: 8-bit quantisation, the reference's one-colour-per-site mosaic, bilinear reconstruction as cv2.demosaicing(COLOR_BAYER_BG2BGR) does it
 * (OpenCV is a third-party dependency absent here: restated from its published algorithm, parity unpinned), / 255. */
int tr_bayer_round_trip(void* stream, const float* src, float* dst, int n, int h, int w);

/* ---- input / output pixel formats (sr_model.py:471-481, utils/color_util.py:237-324): BT.601 conversions of fp32 NCHW 3-channel batches.
 * mode 0 = rgb2ycbcr_pt, 1 = ycbcr2rgb_pt, 2 = adjoint of 1 (gradient of the loss w.r.t. a YCbCr network output given d/d(RGB)). */
int tr_color601(void* stream, const float* src, float* dst, int n, int64_t hw, int mode);

#ifdef __cplusplus
}
#endif
#endif /* TRAINNER_HIP_H */
