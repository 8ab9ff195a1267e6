// Input-image gradient of a 3-channel stem convolution (VALU kernel; K = 27 is far too narrow for MFMA).
//
// Contract: include/trainner_hip.h (tr_conv3x3_dgrad_image).  Replaces autograd's conv_backward_input +
// the adjoint of F.pad(mode="replicate") + the adjoint of `(x - mean) / std` for VGG19 conv1_1
// (traiNNer/losses/perceptual_fp16_loss.py:272-274,322), writing straight into the NCHW fp32 gradient of the
// generator output so the perceptual head adds to the pixel heads without an extra pass.
//
//   gp(py,px)[ci] = sum_{ky,kx,co} gy[py-ky+1, px-kx+1, co] * W[co,ci,ky,kx]        (gy = 0 outside the image)
//   zeros     : gin[iy,ix] = gp(iy,ix)
//   replicate : gin[iy,ix] = sum of gp over every padded position that was filled from (iy,ix)
//   dst[n,ci,iy,ix] (+)= gin * ch_scale[ci]
#include "common.h"

namespace {

constexpr int MAXC = 4;

__global__ __launch_bounds__(256) void dgrad_image_kernel(tr_view gy, const float* __restrict__ w, int cout, int cin, int n, int h, int wd,
                                                           int replicate, float4 chs, float* __restrict__ dst, int accumulate, int vec) {
    extern __shared__ float wl[];   // [tap][co][MAXC]
    for (int i = threadIdx.x; i < 9 * cout * MAXC; i += blockDim.x) {
        const int c = i % MAXC, co = (i / MAXC) % cout, tap = i / (MAXC * cout);
        wl[i] = c < cin ? w[((int64_t)co * cin + c) * 9 + tap] : 0.f;
    }
    __syncthreads();
    const int64_t hw = (int64_t)h * wd, total = (int64_t)n * hw;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int nb = (int)(p / hw);
        const int iy = (int)((p - nb * hw) / wd), ix = (int)(p - nb * hw - (int64_t)iy * wd);
        float acc[MAXC] = {0.f, 0.f, 0.f, 0.f};
        const int ny0 = (replicate && iy == 0) ? -1 : iy, ny1 = (replicate && iy == h - 1) ? h : iy;
        const int nx0 = (replicate && ix == 0) ? -1 : ix, nx1 = (replicate && ix == wd - 1) ? wd : ix;
        for (int py = ny0; py <= ny1; ++py)
            for (int px = nx0; px <= nx1; ++px)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int qy = py - ky + 1;
                    if (qy < 0 || qy >= h) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int qx = px - kx + 1;
                        if (qx < 0 || qx >= wd) continue;
                        const int64_t q = ((int64_t)nb * h + qy) * wd + qx;
                        const float* wt = wl + (ky * 3 + kx) * cout * MAXC;
                        for (int co = 0; co < cout; co += 4) {
                            const float4 g = tr::view_load4(gy, q, co, vec);
                            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float4 ww = *(const float4*)(wt + (co + j) * MAXC);
                                acc[0] += gv[j] * ww.x; acc[1] += gv[j] * ww.y; acc[2] += gv[j] * ww.z; acc[3] += gv[j] * ww.w;
                            }
                        }
                    }
                }
        const float sc[4] = {chs.x, chs.y, chs.z, chs.w};
        for (int c = 0; c < cin; ++c) {
            float* o = dst + ((int64_t)nb * cin + c) * hw + (int64_t)iy * wd + ix;
            const float v = acc[c] * sc[c];
            *o = accumulate ? *o + v : v;
        }
    }
}

__global__ __launch_bounds__(256) void fold_replicate_kernel(tr_view gp, float* __restrict__ dst, int n, int c, int h, int w, float4 chs,
                                                             int accumulate) {
    const int64_t hw = (int64_t)h * w, total = (int64_t)n * hw;
    const int wp = w + 2, hp = h + 2;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int nb = (int)(p / hw);
        const int iy = (int)((p - nb * hw) / w), ix = (int)(p - nb * hw - (int64_t)iy * w);
        // padded rows/cols that were filled from (iy, ix): itself (+1) plus the border copies
        const int y0 = iy == 0 ? 0 : iy + 1, y1 = iy == h - 1 ? hp - 1 : iy + 1;
        const int x0 = ix == 0 ? 0 : ix + 1, x1 = ix == w - 1 ? wp - 1 : ix + 1;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) {
                const int64_t q = ((int64_t)nb * hp + y) * wp + x;
                for (int k = 0; k < c; ++k) acc[k] += tr::view_load(gp, q, k);
            }
        const float sc[4] = {chs.x, chs.y, chs.z, chs.w};
        for (int k = 0; k < c; ++k) {
            float* o = dst + ((int64_t)nb * c + k) * hw + (int64_t)iy * w + ix;
            const float v = acc[k] * sc[k];
            *o = accumulate ? *o + v : v;
        }
    }
}

}  // namespace

extern "C" int tr_fold_replicate_nchw(void* stream, tr_view gp, float* dst_nchw, int n, int c, int h, int w, const float* ch_scale,
                                      int accumulate) {
    TR_REQUIRE(gp.ptr && dst_nchw && n > 0 && h > 0 && w > 0 && c >= 1 && c <= 4 && gp.c >= c, TR_EINVAL, "tr_fold_replicate_nchw: bad arguments");
    float t[4] = {1.f, 1.f, 1.f, 1.f};
    if (ch_scale)
        for (int i = 0; i < c; ++i) t[i] = ch_scale[i];
    const int64_t total = (int64_t)n * h * w;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(fold_replicate_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gp, dst_nchw, n, c, h, w,
                       make_float4(t[0], t[1], t[2], t[3]), accumulate);
    return tr::launch_status("tr_fold_replicate_nchw");
}

extern "C" int tr_conv3x3_dgrad_image(void* stream, tr_view gy, const float* w_oihw, int cout, int cin, int n, int h, int w, int pad_mode,
                                      const float* ch_scale, float* dst_nchw, int accumulate) {
    TR_REQUIRE(gy.ptr && w_oihw && dst_nchw && n > 0 && h > 0 && w > 0, TR_EINVAL, "tr_conv3x3_dgrad_image: bad arguments");
    TR_REQUIRE(cin >= 1 && cin <= MAXC && cout % 4 == 0 && gy.c >= cout && cout <= 256, TR_EUNSUPPORTED,
               "tr_conv3x3_dgrad_image: needs cin <= 4 and cout a multiple of 4 (<= 256)");
    float4 chs = make_float4(1.f, 1.f, 1.f, 1.f);
    if (ch_scale) {
        float t[4] = {1.f, 1.f, 1.f, 1.f};
        for (int i = 0; i < cin; ++i) t[i] = ch_scale[i];
        chs = make_float4(t[0], t[1], t[2], t[3]);
    }
    const int64_t total = (int64_t)n * h * w;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(dgrad_image_kernel, dim3((unsigned)blocks), dim3(256), 9 * cout * MAXC * sizeof(float), (hipStream_t)stream, gy, w_oihw,
                       cout, cin, n, h, w, pad_mode == TR_PAD_REPLICATE, chs, dst_nchw, accumulate, (int)tr::view_vec4_ok(gy));
    return tr::launch_status("tr_conv3x3_dgrad_image");
}
