// Validation-side kernels: image-quality metrics and tile blending.
//
// Reference arithmetic replaced:
//   tensor2img                      traiNNer/utils/img_util.py (clamp to [0,1], x255, round -> uint8 levels)
//   calculate_psnr / calculate_ssim traiNNer/metrics/psnr_ssim.py:13-61,101-156,201-232 (float64 MSE; 11x11 Gaussian(1.5) window, valid region,
//                                   C1 = (0.01*255)^2, C2 = (0.03*255)^2, mean of the SSIM map, mean over channels)
//   to_y_channel                    traiNNer/metrics/metric_util.py:34-47 (BT.601 luma of the 8-bit image, not rounded)
//   infer_tiled blend               traiNNer/models/sr_model.py:852-899 (triangular weights, weighted sum / weight map)
// All sums are double precision, reduced per block and finished on the host in a fixed order (bitwise reproducible).
#include "common.h"

namespace {

__device__ __forceinline__ float q255(float x) { return rintf(fminf(fmaxf(x, 0.f), 1.f) * 255.f); }   // round half to even, as torch.round

// quantise to 8-bit levels, crop the border, optionally reduce to luma: planes[c'][h'][w'] (float, values 0..255)
__global__ __launch_bounds__(256) void metric_prepare_kernel(const float* __restrict__ img, int c, int h, int w, int crop, int ychan, float* __restrict__ out) {
    const int hc = h - 2 * crop, wc = w - 2 * crop;
    const int64_t total = (int64_t)hc * wc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / wc) + crop, x = (int)(i % wc) + crop;
        if (ychan && c == 3) {
            const float r = q255(img[((int64_t)0 * h + y) * w + x]) / 255.f, g = q255(img[((int64_t)1 * h + y) * w + x]) / 255.f,
                        b = q255(img[((int64_t)2 * h + y) * w + x]) / 255.f;
            out[i] = ((65.481f * r + 128.553f * g + 24.966f * b) + 16.0f) / 255.f * 255.f;
        } else {
            for (int ch = 0; ch < c; ++ch) out[(int64_t)ch * total + i] = q255(img[((int64_t)ch * h + y) * w + x]);
        }
    }
}

__device__ __forceinline__ void block_sum_store(double v, double* out) {
    __shared__ double sh[256];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void sqerr_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, double* __restrict__ partial) {
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double d = (double)a[i] - (double)b[i];
        s += d * d;
    }
    block_sum_store(s, partial);
}

struct Gauss11 { double g[11]; };

__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, int nc, int h, int w, Gauss11 k,
                                                    double* __restrict__ partial) {
    const int ho = h - 10, wo = w - 10;
    const int64_t per = (int64_t)ho * wo, total = per * nc;
    const double c1 = (0.01 * 255) * (0.01 * 255), c2 = (0.03 * 255) * (0.03 * 255);
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i / per);
        const int64_t r = i - (int64_t)ch * per;
        const int y = (int)(r / wo), x = (int)(r % wo);
        const float* pa = a + (int64_t)ch * h * w + (int64_t)y * w + x;
        const float* pb = b + (int64_t)ch * h * w + (int64_t)y * w + x;
        double m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
        for (int dy = 0; dy < 11; ++dy) {
            double r1 = 0, r2 = 0, r11 = 0, r22 = 0, r12 = 0;
            for (int dx = 0; dx < 11; ++dx) {
                const double va = pa[dy * w + dx], vb = pb[dy * w + dx], g = k.g[dx];
                r1 += g * va; r2 += g * vb; r11 += g * va * va; r22 += g * vb * vb; r12 += g * va * vb;
            }
            const double g = k.g[dy];
            m1 += g * r1; m2 += g * r2; s11 += g * r11; s22 += g * r22; s12 += g * r12;
        }
        const double v1 = s11 - m1 * m1, v2 = s22 - m2 * m2, cv = s12 - m1 * m2;
        s += ((2 * m1 * m2 + c1) * (2 * cv + c2)) / ((m1 * m1 + m2 * m2 + c1) * (v1 + v2 + c2));
    }
    block_sum_store(s, partial);
}

// output[:, y0:y0+ph, x0:x0+pw] += patch[:, :ph, :pw] * wgt ; wmap likewise += wgt ; wgt(y, x) = tri(y) * tri(x),
// tri(i) = 1 - |i / (hr_tile - 1) - 0.5| * 2
__global__ __launch_bounds__(256) void tile_blend_kernel(const float* __restrict__ patch, int c, int pph, int ppw, float* __restrict__ out,
                                                          float* __restrict__ wmap, int oh, int ow, int y0, int x0, int ph, int pw, int hr_tile) {
    const int64_t total = (int64_t)c * ph * pw;
    const float inv = hr_tile > 1 ? 1.f / (float)(hr_tile - 1) : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % pw), y = (int)((i / pw) % ph), ch = (int)(i / ((int64_t)pw * ph));
        const float wy = 1.f - fabsf((float)y * inv - 0.5f) * 2.f, wx = 1.f - fabsf((float)x * inv - 0.5f) * 2.f;
        const float wg = wy * wx;
        const int64_t o = ((int64_t)ch * oh + (y0 + y)) * ow + (x0 + x);
        out[o] += patch[((int64_t)ch * pph + y) * ppw + x] * wg;
        wmap[o] += wg;
    }
}
__global__ __launch_bounds__(256) void tile_normalize_kernel(const float* __restrict__ acc, const float* __restrict__ wmap, int c, int ah, int aw,
                                                              float* __restrict__ out, int oh, int ow) {
    const int64_t total = (int64_t)c * oh * ow;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % ow), y = (int)((i / ow) % oh), ch = (int)(i / ((int64_t)ow * oh));
        const int64_t s = ((int64_t)ch * ah + y) * aw + x;
        out[i] = acc[s] / fmaxf(wmap[s], 1e-6f);
    }
}


// ---- OTF degradation primitives (first slice): per-sample blur and the 8-bit clamp/round -----------------------------
//   filter2d              traiNNer/utils/img_process_util.py:8-32  (reflect pad k/2, cross-correlation, one k x k kernel per SAMPLE
//                         shared by its channels -- or one kernel for the whole batch)
//   clamp / round         traiNNer/models/realesrgan_model.py:621  torch.clamp((out * 255.0).round(), 0, 255) / 255.0
__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__global__ __launch_bounds__(256) void filter2d_kernel(const float* __restrict__ img, const float* __restrict__ kern, int per_sample, int c, int h, int w,
                                                        int k, float* __restrict__ out) {
    extern __shared__ float kw[];                        // this sample's k x k weights
    const int nb = blockIdx.z, ch = blockIdx.y;
    const float* kp = kern + (per_sample ? (int64_t)nb * k * k : 0);
    for (int i = threadIdx.x; i < k * k; i += blockDim.x) kw[i] = kp[i];
    __syncthreads();
    const float* src = img + ((int64_t)nb * c + ch) * h * w;
    float* dst = out + ((int64_t)nb * c + ch) * h * w;
    const int r = k / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)h * w; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / w), x = (int)(i % w);
        float acc = 0.f;
        for (int dy = 0; dy < k; ++dy) {
            const float* row = src + (int64_t)reflect_idx(y + dy - r, h) * w;
            const float* kr = kw + dy * k;
            if (x - r >= 0 && x + r < w) {
                for (int dx = 0; dx < k; ++dx) acc += row[x + dx - r] * kr[dx];
            } else {
                for (int dx = 0; dx < k; ++dx) acc += row[reflect_idx(x + dx - r, w)] * kr[dx];
            }
        }
        dst[i] = acc;
    }
}
__global__ __launch_bounds__(256) void quantize_u8_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = fminf(fmaxf(rintf(x[i] * 255.0f), 0.f), 255.f) / 255.0f;
}

unsigned nblocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (unsigned)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int tr_metric_prepare(void* stream, const float* img_chw, int c, int h, int w, int crop, int y_channel, float* planes) {
    TR_REQUIRE(img_chw && planes && c > 0 && crop >= 0 && h > 2 * crop && w > 2 * crop, TR_EINVAL, "tr_metric_prepare: bad arguments");
    hipLaunchKernelGGL(metric_prepare_kernel, dim3(nblocks((int64_t)(h - 2 * crop) * (w - 2 * crop))), dim3(256), 0, (hipStream_t)stream, img_chw, c, h, w, crop,
                       y_channel, planes);
    return tr::launch_status("tr_metric_prepare");
}
extern "C" int tr_metric_sqerr(void* stream, const float* a, const float* b, int64_t n, double* partials, int* n_partials) {
    TR_REQUIRE(a && b && partials && n_partials && n > 0, TR_EINVAL, "tr_metric_sqerr: bad arguments");
    const unsigned nb = nblocks(n);
    *n_partials = (int)nb;
    hipLaunchKernelGGL(sqerr_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, n, partials);
    return tr::launch_status("tr_metric_sqerr");
}
extern "C" int tr_metric_ssim(void* stream, const float* a, const float* b, int nc, int h, int w, const double* gauss11, double* partials, int* n_partials) {
    TR_REQUIRE(a && b && partials && n_partials && gauss11 && nc > 0 && h > 10 && w > 10, TR_EINVAL, "tr_metric_ssim: images must exceed the 11x11 window");
    Gauss11 k;
    for (int i = 0; i < 11; ++i) k.g[i] = gauss11[i];
    const unsigned nb = nblocks((int64_t)nc * (h - 10) * (w - 10));
    *n_partials = (int)nb;
    hipLaunchKernelGGL(ssim_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, nc, h, w, k, partials);
    return tr::launch_status("tr_metric_ssim");
}
extern "C" int tr_tile_blend(void* stream, const float* patch, int c, int patch_h, int patch_w, float* out, float* wmap, int out_h, int out_w, int y0, int x0,
                             int ph, int pw, int hr_tile) {
    TR_REQUIRE(patch && out && wmap && c > 0 && ph > 0 && pw > 0 && ph <= patch_h && pw <= patch_w && y0 >= 0 && x0 >= 0 && y0 + ph <= out_h && x0 + pw <= out_w,
               TR_EINVAL, "tr_tile_blend: patch does not fit the output");
    hipLaunchKernelGGL(tile_blend_kernel, dim3(nblocks((int64_t)c * ph * pw)), dim3(256), 0, (hipStream_t)stream, patch, c, patch_h, patch_w, out, wmap, out_h,
                       out_w, y0, x0, ph, pw, hr_tile);
    return tr::launch_status("tr_tile_blend");
}
extern "C" int tr_tile_normalize(void* stream, const float* acc, const float* wmap, int c, int acc_h, int acc_w, float* out, int out_h, int out_w) {
    TR_REQUIRE(acc && wmap && out && c > 0 && out_h > 0 && out_w > 0 && out_h <= acc_h && out_w <= acc_w, TR_EINVAL, "tr_tile_normalize: bad arguments");
    hipLaunchKernelGGL(tile_normalize_kernel, dim3(nblocks((int64_t)c * out_h * out_w)), dim3(256), 0, (hipStream_t)stream, acc, wmap, c, acc_h, acc_w, out, out_h,
                       out_w);
    return tr::launch_status("tr_tile_normalize");
}

extern "C" int tr_filter2d(void* stream, const float* img_nchw, const float* kernels, int per_sample, int n, int c, int h, int w, int k, float* out_nchw) {
    TR_REQUIRE(img_nchw && kernels && out_nchw && n > 0 && c > 0 && h > 0 && w > 0, TR_EINVAL, "tr_filter2d: bad arguments");
    TR_REQUIRE(k % 2 == 1, TR_EINVAL, "tr_filter2d: Wrong kernel size");           // the reference raises ValueError("Wrong kernel size")
    TR_REQUIRE(k / 2 < h && k / 2 < w && k <= 63, TR_EUNSUPPORTED, "tr_filter2d: reflect padding needs k/2 < H, W (k <= 63)");
    unsigned gx = nblocks((int64_t)h * w);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(filter2d_kernel, dim3(gx, c, n), dim3(256), (size_t)k * k * sizeof(float), (hipStream_t)stream, img_nchw, kernels, per_sample, c, h, w,
                       k, out_nchw);
    return tr::launch_status("tr_filter2d");
}
extern "C" int tr_quantize_u8(void* stream, const float* x, float* y, int64_t n) {
    TR_REQUIRE(x && y && n > 0, TR_EINVAL, "tr_quantize_u8: bad arguments");
    hipLaunchKernelGGL(quantize_u8_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return tr::launch_status("tr_quantize_u8");
}
