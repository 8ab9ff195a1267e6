// On-the-fly degradation kernels beyond filter2d / 8-bit rounding (metrics.hip): separable resampling, crop, per-channel gain + noise.
// fp32 NCHW images, like the reference's OTF stage (models/realesrgan_model.py:456-664 works on the DataFeed tensors).  HBM-bound sweeps.
// Contract: include/trainner_hip.h.  Reference arithmetic replaced:
//   resize_pt / _resample                  traiNNer/data/degradations.py:961-1021  (F.interpolate bilinear / bicubic with antialias,
//                                          nearest-exact, Lanczos-3 prefilter + bicubic) -- every mode is a pair of 1-D gathers with
//                                          per-output (index, weight) tables, which the host builds (data/otf.py)
//   paired_random_crop (Tensor branch)     traiNNer/data/transforms.py:118-137
//   exposure / colour temperature / sensor noise     traiNNer/models/paragon_otf_degradations.py:345-414
#include "common.h"

namespace {

// dst[p][y][x] = sum_t w[o][t] * src[p][..idx[o][t]..], o = x (axis 1) or y (axis 0); taps accumulate in table order (like ATen's loops)
__global__ __launch_bounds__(256) void resample_axis_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t planes, int in_h, int in_w, int axis,
                                                             int out_len, const int* __restrict__ idx, const float* __restrict__ wts, int taps, int clamp01) {
    const int oh = axis == 0 ? out_len : in_h, ow = axis == 1 ? out_len : in_w;
    const int64_t total = planes * oh * ow;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % ow);
        const int64_t r = i / ow;
        const int y = (int)(r % oh);
        const int64_t p = r / oh;
        const float* sp = src + p * in_h * in_w;
        const int o = axis == 1 ? x : y;
        const int* ip = idx + (int64_t)o * taps;
        const float* wp = wts + (int64_t)o * taps;
        float acc = 0.f;
        if (axis == 1) {
            const float* row = sp + (int64_t)y * in_w;
            for (int t = 0; t < taps; ++t) acc += wp[t] * row[ip[t]];
        } else {
            for (int t = 0; t < taps; ++t) acc += wp[t] * sp[(int64_t)ip[t] * in_w + x];
        }
        if (clamp01) acc = fminf(fmaxf(acc, 0.f), 1.f);
        dst[i] = acc;
    }
}

__global__ __launch_bounds__(256) void crop_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t planes, int in_h, int in_w, int top, int left,
                                                    int oh, int ow) {
    const int64_t total = planes * oh * ow;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % ow);
        const int64_t r = i / ow;
        const int y = (int)(r % oh);
        const int64_t p = r / oh;
        dst[i] = src[(p * in_h + top + y) * in_w + left + x];
    }
}

// out[p][y][x] = sum_{dy,dx} kern[dy][dx] * in[p][y + dy - pad][x + dx - pad], zeros outside (F.conv2d(groups = c, padding = pad) with one shared kernel)
__global__ __launch_bounds__(256) void filter2d_zeropad_kernel(const float* __restrict__ img, const float* __restrict__ kern, int64_t planes, int h, int w, int k,
                                                                int pad, int oh, int ow, float* __restrict__ out) {
    extern __shared__ float kw[];
    for (int i = threadIdx.x; i < k * k; i += blockDim.x) kw[i] = kern[i];
    __syncthreads();
    const int64_t total = planes * oh * ow;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % ow);
        const int64_t r = i / ow;
        const int y = (int)(r % oh);
        const float* src = img + (r / oh) * h * w;
        float acc = 0.f;
        for (int dy = 0; dy < k; ++dy) {
            const int iy = y + dy - pad;
            if (iy < 0 || iy >= h) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int ix = x + dx - pad;
                if (ix >= 0 && ix < w) acc += src[(int64_t)iy * w + ix] * kw[dy * k + dx];
            }
        }
        out[i] = acc;
    }
}

__global__ __launch_bounds__(256) void axpby_clamp_kernel(const float* __restrict__ a, float wa, const float* __restrict__ b, float wb, float* __restrict__ out,
                                                           int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = fminf(fmaxf(a[i] * wa + b[i] * wb, 0.f), 1.f);
}

struct Gain4 { float g[4]; };
__global__ __launch_bounds__(256) void gain_noise_kernel(const float* __restrict__ src, const float* __restrict__ noise, float* __restrict__ dst, int c, int64_t hw,
                                                          int64_t total, Gain4 gain, float noise_std) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)((i / hw) % c);
        float v = src[i] * gain.g[ch < 4 ? ch : 3];
        if (noise) v += noise[i] * noise_std;
        dst[i] = fminf(fmaxf(v, 0.f), 1.f);
    }
}

inline unsigned sweep_grid(int64_t items) {
    int64_t b = (items + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

extern "C" int tr_resample_axis(void* stream, const float* src, float* dst, int64_t planes, int in_h, int in_w, int axis, int out_len, const int* idx,
                                const float* weights, int taps, int clamp01) {
    TR_REQUIRE(src && dst && idx && weights, TR_EINVAL, "tr_resample_axis: null pointer");
    TR_REQUIRE(planes > 0 && in_h > 0 && in_w > 0 && out_len > 0 && taps > 0 && (axis == 0 || axis == 1), TR_EINVAL,
               "tr_resample_axis: bad dims planes=%lld in=%dx%d axis=%d out_len=%d taps=%d", (long long)planes, in_h, in_w, axis, out_len, taps);
    const int64_t total = planes * (axis == 0 ? (int64_t)out_len * in_w : (int64_t)in_h * out_len);
    hipLaunchKernelGGL(resample_axis_kernel, dim3(sweep_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, planes, in_h, in_w, axis, out_len, idx, weights,
                       taps, clamp01 ? 1 : 0);
    return tr::launch_status("tr_resample_axis");
}

extern "C" int tr_crop_nchw(void* stream, const float* src, float* dst, int64_t planes, int in_h, int in_w, int top, int left, int out_h, int out_w) {
    TR_REQUIRE(src && dst && planes > 0, TR_EINVAL, "tr_crop_nchw: null pointer or no planes");
    TR_REQUIRE(top >= 0 && left >= 0 && out_h > 0 && out_w > 0 && top + out_h <= in_h && left + out_w <= in_w, TR_EINVAL,
               "tr_crop_nchw: window %dx%d at (%d, %d) leaves the %dx%d image", out_h, out_w, top, left, in_h, in_w);
    hipLaunchKernelGGL(crop_kernel, dim3(sweep_grid(planes * out_h * out_w)), dim3(256), 0, (hipStream_t)stream, src, dst, planes, in_h, in_w, top, left, out_h,
                       out_w);
    return tr::launch_status("tr_crop_nchw");
}

extern "C" int tr_gain_noise_clamp(void* stream, const float* src, const float* noise, float* dst, int n, int c, int64_t hw, const float* gain_host,
                                   float noise_std) {
    TR_REQUIRE(src && dst && gain_host && n > 0 && c > 0 && c <= 4 && hw > 0, TR_EINVAL, "tr_gain_noise_clamp: bad arguments (1 to 4 channels)");
    Gain4 g;
    for (int i = 0; i < 4; ++i) g.g[i] = i < c ? gain_host[i] : 1.f;
    const int64_t total = (int64_t)n * c * hw;
    hipLaunchKernelGGL(gain_noise_kernel, dim3(sweep_grid(total)), dim3(256), 0, (hipStream_t)stream, src, noise, dst, c, hw, total, g, noise_std);
    return tr::launch_status("tr_gain_noise_clamp");
}

extern "C" int tr_filter2d_zeropad(void* stream, const float* img_nchw, const float* kernel, int64_t planes, int h, int w, int k, int pad, float* out) {
    TR_REQUIRE(img_nchw && kernel && out && planes > 0 && h > 0 && w > 0 && k > 0 && k <= 63 && pad >= 0, TR_EINVAL, "tr_filter2d_zeropad: bad arguments (k <= 63)");
    const int oh = h + 2 * pad - k + 1, ow = w + 2 * pad - k + 1;
    TR_REQUIRE(oh > 0 && ow > 0, TR_EINVAL, "tr_filter2d_zeropad: empty output");
    hipLaunchKernelGGL(filter2d_zeropad_kernel, dim3(sweep_grid(planes * oh * ow)), dim3(256), (size_t)k * k * sizeof(float), (hipStream_t)stream, img_nchw, kernel,
                       planes, h, w, k, pad, oh, ow, out);
    return tr::launch_status("tr_filter2d_zeropad");
}

extern "C" int tr_axpby_clamp01(void* stream, const float* a, float wa, const float* b, float wb, float* out, int64_t n) {
    TR_REQUIRE(a && b && out && n > 0, TR_EINVAL, "tr_axpby_clamp01: bad arguments");
    hipLaunchKernelGGL(axpby_clamp_kernel, dim3(sweep_grid(n)), dim3(256), 0, (hipStream_t)stream, a, wa, b, wb, out, n);
    return tr::launch_status("tr_axpby_clamp01");
}
