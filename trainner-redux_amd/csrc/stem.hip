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

// Tiled form of the above for the shape that matters (VGG19 conv1_1: 64 -> 3 channels, fp32 gradient at the generator's output size):
// workgroup = 8 x 32 output pixels, one thread per pixel.  The 10 x 34 halo of gy goes through LDS 32 channels at a time (48 KB, three
// workgroups per CU, so one workgroup's staging overlaps its neighbours' FMAs); the weights are read in OIHW order with wave-uniform
// addresses (108 contiguous floats per channel quad -> scalar loads, FMAs take them as SGPR operands).  Algorithmic traffic = gy once
// (256 B per pixel) + 12 B of output; the MFMA 'full'-mode conv it replaces spent 29 of 32 N-lanes on zeros and a second pass to fold.
__global__ __launch_bounds__(256) void dgrad_image_tiled_kernel(tr_view gy, const float* __restrict__ w, int n, int h, int wd, int replicate, float4 chs,
                                                                 float* __restrict__ dst, int accumulate, int ntx, int nty) {
    constexpr int PS = 36;   // floats per staged pixel: 32 channels + 4 of padding (16-byte reads of a row of lanes hit distinct banks)
    __shared__ float gl[10 * 34 * PS];
    const int col = threadIdx.x & 31, row = threadIdx.x >> 5;
    const float* g = (const float*)gy.ptr;
    const int64_t hw = (int64_t)h * wd;
    const int ntiles = n * ntx * nty;
    // software pipeline over (tile, channel half) steps: the loads of step s+1 are issued before the FMAs of step s and land in LDS after them
    const int q8 = threadIdx.x & 7, c8 = threadIdx.x >> 3;
    float4 sv[11];
    auto issue = [&](int tile, int half) {
        int t_ = tile;
        const int x0 = (t_ % ntx) * 32; t_ /= ntx;
        const int y0 = (t_ % nty) * 8;
        const int nb = t_ / nty;
        const float* gb = g + (int64_t)nb * hw * gy.cstride + gy.coff + half * 32 + q8 * 4;
        int r = 0, c = c8;
#pragma unroll
        for (int pass = 0; pass < 11; ++pass) {   // 340 halo pixels x 8 float4, thread = (pixel % 32, q); branch-free so the 11 loads fly together
            const int y = y0 + r - 1, x = x0 + c - 1;
            const bool ok = r < 10 && y >= 0 && y < h && x >= 0 && x < wd;
            const int yc = min(max(y, 0), h - 1), xc = min(max(x, 0), wd - 1);
            sv[pass] = *(const float4*)(gb + ((int64_t)yc * wd + xc) * gy.cstride);
            if (!ok) sv[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
            c += 32;
            if (c >= 34) { c -= 34; ++r; }
        }
    };
    auto land = [&]() {
        int r = 0, c = c8;
#pragma unroll
        for (int pass = 0; pass < 11; ++pass) {
            if (r < 10) *(float4*)(gl + (r * 34 + c) * PS + q8 * 4) = sv[pass];
            c += 32;
            if (c >= 34) { c -= 34; ++r; }
        }
    };
    if ((int)blockIdx.x < ntiles) issue(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t_ = tile;
        const int x0 = (t_ % ntx) * 32; t_ /= ntx;
        const int y0 = (t_ % nty) * 8;
        const int nb = t_ / nty;
        float acc[3] = {0.f, 0.f, 0.f};
        for (int half = 0; half < 2; ++half) {
            __syncthreads();                     // the previous step's FMAs have read the stage
            land();
            __syncthreads();
            if (half == 0) issue(tile, 1);
            else if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x, 0);
            // gp(py,px) reads gy[py-ky+1, px-kx+1] = staged pixel (row + 2 - ky, col + 2 - kx)
            const float* lp = gl + ((row + 2) * 34 + col + 2) * PS;
            for (int q = 0; q < 8; ++q) {
                const float* wq = w + (half * 32 + q * 4) * 27;       // wave-uniform: [co 4][ci 3][tap 9]
                float4 v[9];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) v[ky * 3 + kx] = *(const float4*)(lp - (ky * 34 + kx) * PS + q * 4);
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {          // two channels (54 scalar weights) at a time: all 108 at once spill SGPRs
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const float gv[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
#pragma unroll
                        for (int j = jp * 2; j < jp * 2 + 2; ++j)
#pragma unroll
                            for (int ci = 0; ci < 3; ++ci) acc[ci] += gv[j] * wq[(j * 3 + ci) * 9 + t];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        const int oy = y0 + row, ox = x0 + col;
        if (oy >= h || ox >= wd) continue;
        if (replicate && (oy == 0 || oy == h - 1 || ox == 0 || ox == wd - 1)) {
            // border pixels also collect the padded positions that were filled from them (4 in 1024 pixels: straight from global memory)
            const int ny0 = oy == 0 ? -1 : oy, ny1 = oy == h - 1 ? h : oy;
            const int nx0 = ox == 0 ? -1 : ox, nx1 = ox == wd - 1 ? wd : ox;
            for (int py = ny0; py <= ny1; ++py)
                for (int px = nx0; px <= nx1; ++px) {
                    if (py == oy && px == ox) continue;
                    for (int ky = 0; ky < 3; ++ky) {
                        const int qy = py - ky + 1;
                        if (qy < 0 || qy >= h) continue;
                        for (int kx = 0; kx < 3; ++kx) {
                            const int qx = px - kx + 1;
                            if (qx < 0 || qx >= wd) continue;
                            const float* gp = g + (((int64_t)nb * h + qy) * wd + qx) * gy.cstride + gy.coff;
                            for (int co = 0; co < 64; co += 4) {
                                const float4 v = *(const float4*)(gp + co);
                                const float gv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                                for (int j = 0; j < 4; ++j)
#pragma unroll
                                    for (int ci = 0; ci < 3; ++ci) acc[ci] += gv[j] * w[((co + j) * 3 + ci) * 9 + ky * 3 + kx];
                            }
                        }
                    }
                }
        }
        const float sc[3] = {chs.x, chs.y, chs.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float* o = dst + ((int64_t)nb * 3 + c) * hw + (int64_t)oy * wd + ox;
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


// ---- stem convolution of an NCHW fp32 image: normalise, pad, 3x3 conv (<= 4 -> cout channels), bias, ReLU -> NHWC fp32 view ------------------
// VGG19 conv1_1 (perceptual_fp16_loss.py:272-274,322): K = 27 is far too narrow for MFMA and the layer is bound by its output (256 B per pixel
// against 12 B of input), so this is a VALU kernel whose only job is to write full pixel records: 16 lanes per pixel, 4 output channels each
// (their 27 x 4 weights live in registers), one 16-byte store per lane = 256 contiguous bytes per pixel.  Replaces tr_nchw_to_nhwc + the
// generic fp32 MFMA conv (3.3 ms per launch at batch 16, 1024^2: 29 of its 32 K-lanes multiply zeros).
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ src, const float* __restrict__ w, const float* __restrict__ bias, float4 mean,
                                                         float4 istd, int cin, int cout, int n, int h, int wd, int replicate, int relu, tr_view out, int ntx,
                                                         int nty) {
    // workgroup = one 8 x 32 pixel tile: its normalised, padded 10 x 34 x 3 input patch goes through LDS once (coalesced plane reads), then
    // lane (pixel p, channel quad q) walks the tile 16 pixels at a time; a first version fetched the 27 inputs of a pixel per lane from global
    // memory and was latency bound at 3 waves per SIMD (8.5 ms, worse than the MFMA kernel it replaces)
    __shared__ float xin[3][10][36];
    const int cq = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int ncq = cout >> 2;
    const float mn[4] = {mean.x, mean.y, mean.z, mean.w}, is[4] = {istd.x, istd.y, istd.z, istd.w};
    const int64_t hw = (int64_t)h * wd;
    const int ntiles = ntx * nty * n;
    for (int q0 = 0; q0 < ncq; q0 += 16) {            // cout > 64: a lane takes one channel quad per sweep over the tiles
        const int q = q0 + cq;
        const bool qok = q < ncq;
        float wr[3][9][4];                             // [ci][tap][co of the quad]
        float b4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b4[j] = (bias && qok) ? bias[q * 4 + j] : 0.f;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int t = 0; t < 9; ++t) wr[ci][t][j] = (ci < cin && qok) ? w[((int64_t)(q * 4 + j) * cin + ci) * 9 + t] : 0.f;
        }
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            int t_ = tile;
            const int x0 = (t_ % ntx) * 32; t_ /= ntx;
            const int y0 = (t_ % nty) * 8;
            const int nb = t_ / nty;
            __syncthreads();                          // the previous tile's patch has been consumed
            for (int i = threadIdx.x; i < 3 * 10 * 34; i += 256) {
                const int ci = i / 340, r = (i - ci * 340) / 34, c = i - ci * 340 - r * 34;
                const int y = y0 + r - 1, x = x0 + c - 1;
                const bool ok = ci < cin && (replicate || (y >= 0 && y < h && x >= 0 && x < wd));
                const int yc = min(max(y, 0), h - 1), xc = min(max(x, 0), wd - 1);
                // zero padding pads the NORMALISED image; columns / rows past the image on partial tiles read clamped pixels (never stored)
                xin[ci][r][c] = ok ? (src[((int64_t)nb * cin + min(ci, cin - 1)) * hw + (int64_t)yc * wd + xc] - mn[ci]) * is[ci] : 0.f;
            }
            __syncthreads();
#pragma unroll 2
            for (int pass = 0; pass < 16; ++pass) {
                const int p = pass * 16 + pl, ry = p >> 5, rx = p & 31;
                float acc[4] = {b4[0], b4[1], b4[2], b4[3]};
#pragma unroll
                for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float v = xin[ci][ry + ky][rx + kx];
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[j] += v * wr[ci][ky * 3 + kx][j];
                        }
                if (relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaxf(acc[j], 0.f);
                }
                const int oy = y0 + ry, ox = x0 + rx;
                if (qok && oy < h && ox < wd)
                    *(float4*)((float*)out.ptr + (((int64_t)nb * h + oy) * wd + ox) * out.cstride + out.coff + q * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            }
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
    if (cout == 64 && cin == 3 && gy.dtype == TR_F32 && tr::view_vec4_ok(gy) && !tr::tune().dgrad_image_ref) {
        const int ntx = (w + 31) / 32, nty = (h + 7) / 8;
        const int64_t tiles = (int64_t)n * ntx * nty;
        hipLaunchKernelGGL(dgrad_image_tiled_kernel, dim3((unsigned)(tiles < 256 * 24 ? tiles : 256 * 24)), dim3(256), 0, (hipStream_t)stream, gy, w_oihw, n,
                           h, w, pad_mode == TR_PAD_REPLICATE, chs, dst_nchw, accumulate, ntx, nty);
        return tr::launch_status("tr_conv3x3_dgrad_image");
    }
    const int64_t total = (int64_t)n * h * w;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(dgrad_image_kernel, dim3((unsigned)blocks), dim3(256), 9 * cout * MAXC * sizeof(float), (hipStream_t)stream, gy, w_oihw,
                       cout, cin, n, h, w, pad_mode == TR_PAD_REPLICATE, chs, dst_nchw, accumulate, (int)tr::view_vec4_ok(gy));
    return tr::launch_status("tr_conv3x3_dgrad_image");
}

extern "C" int tr_conv3x3_stem(void* stream, const float* src_nchw, const float* w_oihw, const float* bias, const float* mean_host, const float* std_host,
                               int cin, int cout, int n, int h, int w, int pad_mode, int act, tr_view out) {
    TR_REQUIRE(src_nchw && w_oihw && out.ptr && n > 0 && h > 0 && w > 0, TR_EINVAL, "tr_conv3x3_stem: bad arguments");
    TR_REQUIRE(cin >= 1 && cin <= 3 && cout % 4 == 0 && cout > 0, TR_EUNSUPPORTED, "tr_conv3x3_stem: needs cin <= 3 and cout a multiple of 4");
    TR_REQUIRE(out.dtype == TR_F32 && out.c >= cout && out.coff + cout <= out.cstride && tr::view_vec4_ok(out), TR_EINVAL,
               "tr_conv3x3_stem: out must be a 16-byte sliceable fp32 view of cout channels");
    TR_REQUIRE(act == TR_ACT_NONE || act == TR_ACT_RELU, TR_EUNSUPPORTED, "tr_conv3x3_stem: act must be none or ReLU");
    float m[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {1.f, 1.f, 1.f, 1.f};
    for (int i = 0; i < cin; ++i) {
        if (mean_host) m[i] = mean_host[i];
        if (std_host) is[i] = 1.f / std_host[i];
    }
    const int ntx = (w + 31) / 32, nty = (h + 7) / 8;
    int64_t blocks = (int64_t)ntx * nty * n;
    if (blocks > 256 * 12) blocks = 256 * 12;
    hipLaunchKernelGGL(stem_conv_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src_nchw, w_oihw, bias, make_float4(m[0], m[1], m[2], m[3]),
                       make_float4(is[0], is[1], is[2], is[3]), cin, cout, n, h, w, pad_mode == TR_PAD_REPLICATE, act == TR_ACT_RELU, out, ntx, nty);
    return tr::launch_status("tr_conv3x3_stem");
}
