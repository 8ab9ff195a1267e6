// Layout conversion and pooling kernels (HBM-bound; one pass, vectorised NHWC accesses).
// Contract: include/trainner_hip.h.  Reference arithmetic replaced:
//   img2tensor CHW float32 batches -> channels_last   traiNNer/models/sr_model.py:434-446
//   (x - mean) / std                                  traiNNer/losses/perceptual_fp16_loss.py:322
//   nn.MaxPool2d(2, 2) of torchvision VGG19 features  traiNNer/losses/perceptual_fp16_loss.py:12-50
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, int c_src, int64_t hw, int64_t npix,
                                                            tr_view dst, float3 mean, float3 istd, int norm) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t nb = p / hw, s = p - nb * hw;
        const float* sp = src + nb * c_src * hw + s;
        for (int c = 0; c < dst.c; ++c) {
            float v = 0.f;
            if (c < c_src) {
                v = sp[(int64_t)c * hw];
                if (norm) {
                    const float m = c == 0 ? mean.x : (c == 1 ? mean.y : mean.z);
                    const float is = c == 0 ? istd.x : (c == 1 ? istd.y : istd.z);
                    v = (v - m) / is;   // `is` holds std here (division like the reference)
                }
            }
            tr::view_store(dst, p, c, v);
        }
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(tr_view src, float* __restrict__ dst, int c_dst, int64_t hw, int64_t npix,
                                                            float3 scale, int use_scale, int accumulate, float3 offset = make_float3(0.f, 0.f, 0.f)) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t nb = p / hw, s = p - nb * hw;
        float* dp = dst + nb * c_dst * hw + s;
        for (int c = 0; c < c_dst; ++c) {
            float v = tr::view_load(src, p, c);
            if (use_scale) v = v * (c == 0 ? scale.x : (c == 1 ? scale.y : scale.z)) + (c == 0 ? offset.x : (c == 1 ? offset.y : offset.z));
            if (accumulate) v += dp[(int64_t)c * hw];
            dp[(int64_t)c * hw] = v;
        }
    }
}

__global__ __launch_bounds__(256) void axpby_kernel(tr_view a, float sa, tr_view b, float sb, tr_view out, int64_t npix, int c4, int va,
                                                     int vb, int vout, tr_view mask, float mslope, int vm, const float* __restrict__ sscale,
                                                     int64_t pps) {
    const int64_t total = npix * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / c4;
        const int c = (int)(i - p * c4) * 4;
        float4 v = tr::view_load4(a, p, c, va);
        const float sa_ = sscale ? sa * sscale[p / pps] : sa;
        v.x *= sa_; v.y *= sa_; v.z *= sa_; v.w *= sa_;
        if (b.ptr) {
            const float4 w = tr::view_load4(b, p, c, vb);
            v.x += sb * w.x; v.y += sb * w.y; v.z += sb * w.z; v.w += sb * w.w;
        }
        if (mask.ptr) {
            const float4 m = tr::view_load4(mask, p, c, vm);
            v.x *= m.x > 0.f ? 1.f : mslope; v.y *= m.y > 0.f ? 1.f : mslope;
            v.z *= m.z > 0.f ? 1.f : mslope; v.w *= m.w > 0.f ? 1.f : mslope;
        }
        tr::view_store4(out, p, c, v, vout);
    }
}


// ---- feed: uint8 NHWC -> fp32 NCHW ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void u8hwc_to_nchw_kernel(const uint8_t* __restrict__ src, int n, int64_t hw, int c, float* __restrict__ dst, float3 mean,
                                                             float3 istd) {
    // one thread = 4 consecutive pixels of one image: 4*c contiguous source bytes, c float4 stores (one per plane).
    // x / 255 is a correctly rounded division, bit-identical to the reference's `img.float() / 255.` (utils/img_util.py img2tensor)
    const int64_t q = hw >> 2, total = (int64_t)n * q;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t nb = i / q, p0 = (i - nb * q) * 4;
        const uint8_t* s = src + (nb * hw + p0) * c;
        const float m[3] = {mean.x, mean.y, mean.z}, is[3] = {istd.x, istd.y, istd.z};
        for (int ch = 0; ch < c; ++ch) {
            float4 v;
            v.x = ((float)s[ch] / 255.f - m[ch % 3]) * is[ch % 3];
            v.y = ((float)s[c + ch] / 255.f - m[ch % 3]) * is[ch % 3];
            v.z = ((float)s[2 * c + ch] / 255.f - m[ch % 3]) * is[ch % 3];
            v.w = ((float)s[3 * c + ch] / 255.f - m[ch % 3]) * is[ch % 3];
            *(float4*)(dst + (nb * c + ch) * hw + p0) = v;
        }
    }
}

// ---- SPAN parameter-free attention (span_arch.py:243-247) ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void span_gate_kernel(tr_view o3, tr_view x, tr_view out, tr_view out2, int64_t npix, int c4, int v3, int vx, int vo,
                                                         int vo2) {
    const int64_t total = npix * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / c4;
        const int c = (int)(i - p * c4) * 4;
        const float4 a = tr::view_load4(o3, p, c, v3), b = tr::view_load4(x, p, c, vx);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = (av[k] + bv[k]) * (1.f / (1.f + __expf(-av[k])) - 0.5f);
        const float4 o = make_float4(r[0], r[1], r[2], r[3]);
        tr::view_store4(out, p, c, o, vo);
        if (out2.ptr) tr::view_store4(out2, p, c, o, vo2);
    }
}
__global__ __launch_bounds__(256) void span_gate_bwd_kernel(tr_view g, tr_view o3, tr_view x, tr_view g3, tr_view gx, int acc, int64_t npix, int c4, int vg,
                                                             int v3, int vx, int vg3, int vgx) {
    const int64_t total = npix * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / c4;
        const int c = (int)(i - p * c4) * 4;
        const float4 gg = tr::view_load4(g, p, c, vg), a = tr::view_load4(o3, p, c, v3), b = tr::view_load4(x, p, c, vx);
        const float gv[4] = {gg.x, gg.y, gg.z, gg.w}, av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
        float r3[4], rx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sg = 1.f / (1.f + __expf(-av[k]));
            rx[k] = gv[k] * (sg - 0.5f);
            r3[k] = rx[k] + gv[k] * (av[k] + bv[k]) * sg * (1.f - sg);
        }
        tr::view_store4(g3, p, c, make_float4(r3[0], r3[1], r3[2], r3[3]), vg3);
        if (gx.ptr) {
            if (acc) {
                const float4 o = tr::view_load4(gx, p, c, vgx);
                rx[0] += o.x; rx[1] += o.y; rx[2] += o.z; rx[3] += o.w;
            }
            tr::view_store4(gx, p, c, make_float4(rx[0], rx[1], rx[2], rx[3]), vgx);
        }
    }
}

// one thread = one LOW-resolution pixel x 4 channels
__global__ __launch_bounds__(256) void sumpool_kernel(tr_view in, tr_view out, tr_view mask, float mslope, int n, int hin, int win, int vin,
                                                       int vout, int vmask) {
    const int ho = hin >> 1, wo = win >> 1, c4 = out.c >> 2;
    const int64_t total = (int64_t)n * ho * wo * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int c = (int)(t % c4) * 4; t /= c4;
        const int x = (int)(t % wo); t /= wo;
        const int y = (int)(t % ho);
        const int nb = (int)(t / ho);
        const int64_t p00 = ((int64_t)nb * hin + 2 * y) * win + 2 * x;
        const float4 a = tr::view_load4(in, p00, c, vin), b = tr::view_load4(in, p00 + 1, c, vin);
        const float4 d = tr::view_load4(in, p00 + win, c, vin), e = tr::view_load4(in, p00 + win + 1, c, vin);
        float4 s = make_float4((a.x + b.x) + (d.x + e.x), (a.y + b.y) + (d.y + e.y), (a.z + b.z) + (d.z + e.z), (a.w + b.w) + (d.w + e.w));
        const int64_t po = ((int64_t)nb * ho + y) * wo + x;
        if (mask.ptr) {
            const float4 m = tr::view_load4(mask, po, c, vmask);
            s.x *= m.x > 0.f ? 1.f : mslope; s.y *= m.y > 0.f ? 1.f : mslope;
            s.z *= m.z > 0.f ? 1.f : mslope; s.w *= m.w > 0.f ? 1.f : mslope;
        }
        tr::view_store4(out, po, c, s, vout);
    }
}

// one thread = one output pixel x 4 channels
__global__ __launch_bounds__(256) void maxpool_kernel(tr_view in, tr_view out, int n, int hin, int win, int relu, int vin, int vout) {
    const int ho = hin >> 1, wo = win >> 1, c4 = out.c >> 2;
    const int64_t total = (int64_t)n * ho * wo * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int c = (int)(t % c4) * 4; t /= c4;
        const int x = (int)(t % wo); t /= wo;
        const int y = (int)(t % ho);
        const int nb = (int)(t / ho);
        const int64_t p00 = ((int64_t)nb * hin + 2 * y) * win + 2 * x;
        float4 a = tr::view_load4(in, p00, c, vin), b = tr::view_load4(in, p00 + 1, c, vin);
        float4 d = tr::view_load4(in, p00 + win, c, vin), e = tr::view_load4(in, p00 + win + 1, c, vin);
        float4 m;
        m.x = fmaxf(fmaxf(a.x, b.x), fmaxf(d.x, e.x)); m.y = fmaxf(fmaxf(a.y, b.y), fmaxf(d.y, e.y));
        m.z = fmaxf(fmaxf(a.z, b.z), fmaxf(d.z, e.z)); m.w = fmaxf(fmaxf(a.w, b.w), fmaxf(d.w, e.w));
        if (relu) { m.x = fmaxf(m.x, 0.f); m.y = fmaxf(m.y, 0.f); m.z = fmaxf(m.z, 0.f); m.w = fmaxf(m.w, 0.f); }
        tr::view_store4(out, ((int64_t)nb * ho + y) * wo + x, c, m, vout);
    }
}

__device__ __forceinline__ void route(float a, float b, float d, float e, float g, int relu, float& ga, float& gb, float& gd, float& ge) {
    if (relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); d = fmaxf(d, 0.f); e = fmaxf(e, 0.f); }
    // first maximal element in window scan order (ATen max_pool2d keeps the earliest index on ties)
    int k = 0; float m = a;
    if (b > m) { m = b; k = 1; }
    if (d > m) { m = d; k = 2; }
    if (e > m) { m = e; k = 3; }
    if (relu && !(m > 0.f)) g = 0.f;   // ReLU gradient is 0 at and below 0 (threshold backward)
    ga = k == 0 ? g : 0.f; gb = k == 1 ? g : 0.f; gd = k == 2 ? g : 0.f; ge = k == 3 ? g : 0.f;
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(tr_view in, tr_view gout, tr_view gin, int n, int hin, int win, int relu,
                                                           int vin, int vgo, int vgi) {
    const int ho = hin >> 1, wo = win >> 1, c4 = gout.c >> 2;
    const int64_t total = (int64_t)n * ho * wo * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int c = (int)(t % c4) * 4; t /= c4;
        const int x = (int)(t % wo); t /= wo;
        const int y = (int)(t % ho);
        const int nb = (int)(t / ho);
        const int64_t p00 = ((int64_t)nb * hin + 2 * y) * win + 2 * x;
        const float4 a = tr::view_load4(in, p00, c, vin), b = tr::view_load4(in, p00 + 1, c, vin);
        const float4 d = tr::view_load4(in, p00 + win, c, vin), e = tr::view_load4(in, p00 + win + 1, c, vin);
        const float4 g = tr::view_load4(gout, ((int64_t)nb * ho + y) * wo + x, c, vgo);
        float4 ga, gb, gd, ge;
        route(a.x, b.x, d.x, e.x, g.x, relu, ga.x, gb.x, gd.x, ge.x);
        route(a.y, b.y, d.y, e.y, g.y, relu, ga.y, gb.y, gd.y, ge.y);
        route(a.z, b.z, d.z, e.z, g.z, relu, ga.z, gb.z, gd.z, ge.z);
        route(a.w, b.w, d.w, e.w, g.w, relu, ga.w, gb.w, gd.w, ge.w);
        tr::view_store4(gin, p00, c, ga, vgi); tr::view_store4(gin, p00 + 1, c, gb, vgi);
        tr::view_store4(gin, p00 + win, c, gd, vgi); tr::view_store4(gin, p00 + win + 1, c, ge, vgi);
    }
}

inline unsigned grid_for(int64_t work) {
    int64_t b = (work + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int tr_nchw_to_nhwc(void* stream, const float* src, int n, int c_src, int h, int w, tr_view dst,
                               const float* mean3, const float* std3) {
    TR_REQUIRE(src && dst.ptr && n > 0 && c_src > 0 && h > 0 && w > 0, TR_EINVAL, "tr_nchw_to_nhwc: bad arguments");
    TR_REQUIRE(dst.coff + dst.c <= dst.cstride && dst.c >= c_src, TR_EINVAL, "tr_nchw_to_nhwc: view narrower than c_src=%d", c_src);
    TR_REQUIRE((mean3 == nullptr) == (std3 == nullptr) && (!mean3 || c_src == 3), TR_EINVAL,
               "tr_nchw_to_nhwc: mean/std need 3 source channels (host pointers)");
    float3 m = make_float3(0, 0, 0), s = make_float3(1, 1, 1);
    if (mean3) { m = make_float3(mean3[0], mean3[1], mean3[2]); s = make_float3(std3[0], std3[1], std3[2]); }
    const int64_t npix = (int64_t)n * h * w;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, src, c_src, (int64_t)h * w, npix,
                       dst, m, s, mean3 ? 1 : 0);
    return tr::launch_status("tr_nchw_to_nhwc");
}

extern "C" int tr_nhwc_to_nchw(void* stream, tr_view src, float* dst, int n, int c_dst, int h, int w) {
    TR_REQUIRE(src.ptr && dst && n > 0 && c_dst > 0 && src.c >= c_dst, TR_EINVAL, "tr_nhwc_to_nchw: bad arguments");
    const int64_t npix = (int64_t)n * h * w;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, src, dst, c_dst, (int64_t)h * w,
                       npix, make_float3(1, 1, 1), 0, 0);
    return tr::launch_status("tr_nhwc_to_nchw");
}

extern "C" int tr_nhwc_to_nchw_scaled(void* stream, tr_view src, float* dst, int n, int c_dst, int h, int w,
                                      const float* inv_std3, int accumulate) {
    TR_REQUIRE(src.ptr && dst && n > 0 && c_dst == 3 && src.c >= 3 && inv_std3, TR_EINVAL, "tr_nhwc_to_nchw_scaled: bad arguments");
    const int64_t npix = (int64_t)n * h * w;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, src, dst, c_dst, (int64_t)h * w,
                       npix, make_float3(inv_std3[0], inv_std3[1], inv_std3[2]), 1, accumulate);
    return tr::launch_status("tr_nhwc_to_nchw_scaled");
}

extern "C" int tr_nhwc_to_nchw_affine(void* stream, tr_view src, float* dst, int n, int c_dst, int h, int w, const float* scale3,
                                      const float* offset3) {
    TR_REQUIRE(src.ptr && dst && n > 0 && c_dst == 3 && src.c >= 3 && scale3 && offset3, TR_EINVAL, "tr_nhwc_to_nchw_affine: bad arguments");
    const int64_t npix = (int64_t)n * h * w;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, src, dst, c_dst, (int64_t)h * w, npix,
                       make_float3(scale3[0], scale3[1], scale3[2]), 1, 0, make_float3(offset3[0], offset3[1], offset3[2]));
    return tr::launch_status("tr_nhwc_to_nchw_affine");
}

extern "C" int tr_axpby(void* stream, tr_view a, float sa, tr_view b, float sb, tr_view out, int64_t npix, tr_view mask, float mask_slope,
                        const float* sample_scale, int64_t pix_per_sample) {
    TR_REQUIRE(a.ptr && out.ptr && npix > 0 && a.c == out.c && a.c % 4 == 0 && (!b.ptr || b.c == a.c) && (!mask.ptr || mask.c == a.c), TR_EINVAL,
               "tr_axpby: views must have equal channel counts that are multiples of 4");
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(npix * (a.c / 4))), dim3(256), 0, (hipStream_t)stream, a, sa, b, sb, out, npix, a.c / 4,
                       (int)tr::view_vec4_ok(a), (int)tr::view_vec4_ok(b), (int)tr::view_vec4_ok(out), mask, mask_slope, (int)tr::view_vec4_ok(mask),
                       sample_scale, pix_per_sample > 0 ? pix_per_sample : npix);
    return tr::launch_status("tr_axpby");
}

extern "C" int tr_u8hwc_to_nchw(void* stream, const uint8_t* src, int n, int h, int w, int c, float* dst, const float* mean3, const float* std3) {
    TR_REQUIRE(src && dst && n > 0 && h > 0 && w > 0 && c > 0 && c <= 4, TR_EINVAL, "tr_u8hwc_to_nchw: bad arguments");
    TR_REQUIRE(((int64_t)h * w) % 4 == 0 && ((uintptr_t)dst) % 16 == 0, TR_EALIGN, "tr_u8hwc_to_nchw: h*w must be a multiple of 4 and dst 16-byte aligned");
    const float3 mean = mean3 ? make_float3(mean3[0], mean3[1], mean3[2]) : make_float3(0.f, 0.f, 0.f);
    const float3 istd = std3 ? make_float3(1.f / std3[0], 1.f / std3[1], 1.f / std3[2]) : make_float3(1.f, 1.f, 1.f);
    const int64_t hw = (int64_t)h * w;
    hipLaunchKernelGGL(u8hwc_to_nchw_kernel, dim3(grid_for(n * (hw / 4))), dim3(256), 0, (hipStream_t)stream, src, n, hw, c, dst, mean, istd);
    return tr::launch_status("tr_u8hwc_to_nchw");
}

extern "C" int tr_span_gate(void* stream, tr_view o3, tr_view x, tr_view out, tr_view out2, int64_t npix) {
    TR_REQUIRE(o3.ptr && x.ptr && out.ptr && npix > 0 && o3.c % 4 == 0 && x.c == o3.c && out.c == o3.c && (!out2.ptr || out2.c == o3.c), TR_EINVAL,
               "tr_span_gate: views must have equal channel counts that are multiples of 4");
    hipLaunchKernelGGL(span_gate_kernel, dim3(grid_for(npix * (o3.c / 4))), dim3(256), 0, (hipStream_t)stream, o3, x, out, out2, npix, o3.c / 4,
                       (int)tr::view_vec4_ok(o3), (int)tr::view_vec4_ok(x), (int)tr::view_vec4_ok(out), (int)tr::view_vec4_ok(out2));
    return tr::launch_status("tr_span_gate");
}
extern "C" int tr_span_gate_bwd(void* stream, tr_view g, tr_view o3, tr_view x, tr_view g_o3, tr_view g_x, int accumulate_gx, int64_t npix) {
    TR_REQUIRE(g.ptr && o3.ptr && x.ptr && g_o3.ptr && npix > 0 && o3.c % 4 == 0 && g.c == o3.c && x.c == o3.c && g_o3.c == o3.c &&
                   (!g_x.ptr || g_x.c == o3.c), TR_EINVAL, "tr_span_gate_bwd: views must have equal channel counts that are multiples of 4");
    hipLaunchKernelGGL(span_gate_bwd_kernel, dim3(grid_for(npix * (o3.c / 4))), dim3(256), 0, (hipStream_t)stream, g, o3, x, g_o3, g_x, accumulate_gx, npix,
                       o3.c / 4, (int)tr::view_vec4_ok(g), (int)tr::view_vec4_ok(o3), (int)tr::view_vec4_ok(x), (int)tr::view_vec4_ok(g_o3),
                       (int)tr::view_vec4_ok(g_x));
    return tr::launch_status("tr_span_gate_bwd");
}

extern "C" int tr_sumpool2x2(void* stream, tr_view in, tr_view out, tr_view mask, float mask_slope, int n, int h_in, int w_in) {
    TR_REQUIRE(in.ptr && out.ptr && n > 0 && in.c == out.c && in.c % 4 == 0 && (!mask.ptr || mask.c == in.c), TR_EINVAL,
               "tr_sumpool2x2: channel mismatch");
    TR_REQUIRE(h_in % 2 == 0 && w_in % 2 == 0 && h_in > 0 && w_in > 0, TR_EUNSUPPORTED, "tr_sumpool2x2: odd spatial dims %dx%d", h_in, w_in);
    const int64_t work = (int64_t)n * (h_in / 2) * (w_in / 2) * (in.c / 4);
    hipLaunchKernelGGL(sumpool_kernel, dim3(grid_for(work)), dim3(256), 0, (hipStream_t)stream, in, out, mask, mask_slope, n, h_in, w_in,
                       (int)tr::view_vec4_ok(in), (int)tr::view_vec4_ok(out), (int)tr::view_vec4_ok(mask));
    return tr::launch_status("tr_sumpool2x2");
}

extern "C" int tr_maxpool2x2(void* stream, tr_view in, tr_view out, int n, int h_in, int w_in, int relu_first) {
    TR_REQUIRE(in.ptr && out.ptr && n > 0 && in.c == out.c && in.c % 4 == 0, TR_EINVAL, "tr_maxpool2x2: channel mismatch");
    TR_REQUIRE(h_in % 2 == 0 && w_in % 2 == 0 && h_in > 0 && w_in > 0, TR_EUNSUPPORTED, "tr_maxpool2x2: odd spatial dims %dx%d", h_in, w_in);
    const int64_t work = (int64_t)n * (h_in / 2) * (w_in / 2) * (in.c / 4);
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(work)), dim3(256), 0, (hipStream_t)stream, in, out, n, h_in, w_in, relu_first,
                       (int)tr::view_vec4_ok(in), (int)tr::view_vec4_ok(out));
    return tr::launch_status("tr_maxpool2x2");
}

extern "C" int tr_maxpool2x2_bwd(void* stream, tr_view in, tr_view gout, tr_view gin, int n, int h_in, int w_in, int relu_first) {
    TR_REQUIRE(in.ptr && gout.ptr && gin.ptr && in.c == gout.c && in.c == gin.c && in.c % 4 == 0, TR_EINVAL,
               "tr_maxpool2x2_bwd: channel mismatch");
    TR_REQUIRE(h_in % 2 == 0 && w_in % 2 == 0 && h_in > 0 && w_in > 0, TR_EUNSUPPORTED, "tr_maxpool2x2_bwd: odd spatial dims");
    const int64_t work = (int64_t)n * (h_in / 2) * (w_in / 2) * (in.c / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(work)), dim3(256), 0, (hipStream_t)stream, in, gout, gin, n, h_in, w_in,
                       relu_first, (int)tr::view_vec4_ok(in), (int)tr::view_vec4_ok(gout), (int)tr::view_vec4_ok(gin));
    return tr::launch_status("tr_maxpool2x2_bwd");
}
