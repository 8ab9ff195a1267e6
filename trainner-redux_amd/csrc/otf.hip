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

// ---- F.grid_sample(bilinear, align_corners=False) over analytic grids ---------------------------------------------------------------
// mode 0  lens distortion   (paragon_otf_degradations.py:312-343): with u = lin(H)[y], v = lin(W)[x] (lin = torch.linspace(-1, 1, n)),
//         f = 1 + k (u^2 + v^2): the grid's x coordinate is u f and its y coordinate v f -- the reference stacks its `ij` meshgrid in that
//         order, i.e. it samples the TRANSPOSED position; kept as it is.  Reflection padding.
// mode 1  rolling shutter   (:430-456): x' = lin(W)[x] + slant * lin(H)[y], y' = lin(H)[y].  Reflection padding.
// mode 2  per-channel scale (realesrgan_model.py:262-311, affine_grid of diag(s, s)): x' = s (2x + 1 - W) / W, same for y; s = p[channel].
//         Zero padding, clamp to [0, 1].
__device__ __forceinline__ float lin11(int i, int n) {          // torch.linspace(-1, 1, n)[i]
    if (n == 1) return -1.f;
    const float step = 2.f / (float)(n - 1);
    return i < n / 2 ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}
__device__ __forceinline__ float reflect_coord(float in, int size) {   // ATen reflect_coordinates(in, -1, 2 size - 1) then clip
    const float mn = -0.5f, span = (float)size;
    in = fabsf(in - mn);
    const float extra = fmodf(in, span);
    const int flips = (int)floorf(in / span);
    const float r = (flips & 1) == 0 ? extra + mn : span - extra + mn;
    return fminf((float)(size - 1), fmaxf(r, 0.f));
}
__global__ __launch_bounds__(256) void warp_bilinear_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int c, int h, int w, int mode, float p0,
                                                             float p1, float p2, float p3) {
    const int64_t total = (int64_t)n * c * h * w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w);
        const int64_t r = i / w;
        const int y = (int)(r % h);
        const int64_t pl = r / h;
        const int ch = (int)(pl % c);
        float gx, gy;
        if (mode == 0) {
            const float u = lin11(y, h), v = lin11(x, w);
            float rr = sqrtf(u * u + v * v);
            float rd = rr * (1.f + p0 * rr * rr);
            if (rr == 0.f) { rd = 0.f; rr = 1e-6f; }
            const float f = rd / rr;
            gx = u * f; gy = v * f;
        } else if (mode == 1) {
            gy = lin11(y, h);
            gx = lin11(x, w) + p0 * gy;
        } else {
            const float s = ch == 0 ? p0 : ch == 1 ? p1 : ch == 2 ? p2 : p3;
            gx = s * ((2.f * x + 1.f) / (float)w - 1.f);
            gy = s * ((2.f * y + 1.f) / (float)h - 1.f);
        }
        float ix = ((gx + 1.f) * (float)w - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)h - 1.f) * 0.5f;
        if (mode != 2) { ix = reflect_coord(ix, w); iy = reflect_coord(iy, h); }
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float tx = ix - fx, ty = iy - fy;
        const float* sp = src + pl * h * w;
        auto at = [&](int yy, int xx) { return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? sp[(int64_t)yy * w + xx] : 0.f; };
        float v = at(y0, x0) * (1.f - tx) * (1.f - ty) + at(y0, x0 + 1) * tx * (1.f - ty) + at(y0 + 1, x0) * (1.f - tx) * ty + at(y0 + 1, x0 + 1) * tx * ty;
        if (mode == 2) v = fminf(fmaxf(v, 0.f), 1.f);
        dst[i] = v;
    }
}

// BT.601 RGB <-> YCbCr on fp32 NCHW 3-channel batches, in the reference's operation order (utils/color_util.py:275-324):
//   mode 0  rgb2ycbcr_pt:  (rgb . W[:, c] + bias[c]) / 255
//   mode 1  ycbcr2rgb_pt:  r = 1.164 (y255 - 16) + 1.596 (cr255 - 128) ..., / 255, with y255 = y * 255
//   mode 2  adjoint of mode 1 (gradient w.r.t. the YCbCr image given the gradient w.r.t. the RGB image; the map is affine with a 255 / 255 = 1 gain)
__global__ __launch_bounds__(256) void color601_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t hw, int64_t total, int mode) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t nb = i / hw, p = i - nb * hw;
        const float* s = src + nb * 3 * hw + p;
        float* d = dst + nb * 3 * hw + p;
        const float a = s[0], b = s[hw], c = s[2 * hw];
        float o0, o1, o2;
        if (mode == 0) {
            o0 = ((a * 65.481f + b * 128.553f + c * 24.966f) + 16.f) / 255.f;
            o1 = ((a * -37.797f + b * -74.203f + c * 112.0f) + 128.f) / 255.f;
            o2 = ((a * 112.0f + b * -93.786f + c * -18.214f) + 128.f) / 255.f;
        } else if (mode == 1) {
            const float y = a * 255.f, cb = b * 255.f, cr = c * 255.f;
            o0 = (1.164f * (y - 16.f) + 1.596f * (cr - 128.f)) / 255.f;
            o1 = (1.164f * (y - 16.f) - 0.392f * (cb - 128.f) - 0.813f * (cr - 128.f)) / 255.f;
            o2 = (1.164f * (y - 16.f) + 2.017f * (cb - 128.f)) / 255.f;
        } else {
            o0 = 1.164f * (a + b + c);
            o1 = -0.392f * b + 2.017f * c;
            o2 = 1.596f * a - 0.813f * b;
        }
        d[0] = o0; d[hw] = o1; d[2 * hw] = o2;
    }
}

// Demosaicing-artifact stage (realesrgan_model.py:312-363): quantise to 8 bits, keep one colour per site -- channel 2 at (even, even), channel 0 at (odd, odd),
// channel 1 elsewhere, the reference's mosaic -- and rebuild the other two by OpenCV's bilinear cv2.demosaicing(COLOR_BAYER_BG2BGR) (opencv 4.x
// imgproc/demosaicing.cpp Bayer2RGB_, restated: 2-neighbour (a + b + 1) >> 1, 4-neighbour (a + b + c + d + 2) >> 2 in integers; the first / last column
// repeat their inner neighbour, then the first / last row repeat theirs).  One thread per pixel, three outputs; u8 / 255 back to fp32.
__device__ __forceinline__ int bayer_site(const float* __restrict__ img, int64_t hw, int w, int y, int x) {
    const int ch = (y & 1) ? ((x & 1) ? 0 : 1) : ((x & 1) ? 1 : 2);
    const float v = fminf(fmaxf(img[ch * hw + (int64_t)y * w + x], 0.f), 1.f);
    return (int)(unsigned char)(v * 255.f);
}

__global__ __launch_bounds__(256) void bayer_round_trip_kernel(const float* __restrict__ src, float* __restrict__ dst, int h, int w, int64_t total) {
    const int64_t hw = (int64_t)h * w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t nb = i / hw, p = i - nb * hw;
        const int y0 = (int)(p / w), x0 = (int)(p - (int64_t)y0 * w);
        const int y = min(max(y0, 1), h - 2), x = min(max(x0, 1), w - 2);
        const float* img = src + nb * 3 * hw;
        const int ctr = bayer_site(img, hw, w, y, x);
        const int up = bayer_site(img, hw, w, y - 1, x), dn = bayer_site(img, hw, w, y + 1, x);
        const int lf = bayer_site(img, hw, w, y, x - 1), rt = bayer_site(img, hw, w, y, x + 1);
        int o[3];
        if ((y ^ x) & 1) {                                   // green site: the other two colours sit beside it, one pair per axis
            const int horiz = (lf + rt + 1) >> 1, vert = (up + dn + 1) >> 1;
            o[1] = ctr;
            o[(y & 1) ? 0 : 2] = horiz;                      // odd rows hold channel 0 beside the greens, even rows channel 2
            o[(y & 1) ? 2 : 0] = vert;
        } else {
            const int diag = (bayer_site(img, hw, w, y - 1, x - 1) + bayer_site(img, hw, w, y - 1, x + 1) + bayer_site(img, hw, w, y + 1, x - 1) +
                              bayer_site(img, hw, w, y + 1, x + 1) + 2) >> 2;
            o[1] = (up + lf + rt + dn + 2) >> 2;
            o[(y & 1) ? 0 : 2] = ctr;
            o[(y & 1) ? 2 : 0] = diag;
        }
        float* d = dst + nb * 3 * hw + p;
        d[0] = (float)o[0] / 255.f; d[hw] = (float)o[1] / 255.f; d[2 * hw] = (float)o[2] / 255.f;
    }
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

extern "C" int tr_warp_bilinear(void* stream, const float* src, float* dst, int n, int c, int h, int w, int mode, const float* params_host) {
    TR_REQUIRE(src && dst && params_host && n > 0 && c > 0 && h > 0 && w > 0, TR_EINVAL, "tr_warp_bilinear: bad arguments");
    TR_REQUIRE(mode >= 0 && mode <= 2 && (mode != 2 || c <= 4), TR_EINVAL, "tr_warp_bilinear: mode %d (0 lens, 1 shear, 2 per-channel scale with <= 4 channels)", mode);
    hipLaunchKernelGGL(warp_bilinear_kernel, dim3(sweep_grid((int64_t)n * c * h * w)), dim3(256), 0, (hipStream_t)stream, src, dst, n, c, h, w, mode, params_host[0],
                       params_host[1], params_host[2], params_host[3]);
    return tr::launch_status("tr_warp_bilinear");
}

extern "C" int tr_color601(void* stream, const float* src, float* dst, int n, int64_t hw, int mode) {
    TR_REQUIRE(src && dst && n > 0 && hw > 0 && mode >= 0 && mode <= 2, TR_EINVAL, "tr_color601: bad arguments (mode 0 rgb->ycbcr, 1 ycbcr->rgb, 2 adjoint of 1)");
    const int64_t total = (int64_t)n * hw;
    hipLaunchKernelGGL(color601_kernel, dim3(sweep_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, hw, total, mode);
    return tr::launch_status("tr_color601");
}

extern "C" int tr_bayer_round_trip(void* stream, const float* src, float* dst, int n, int h, int w) {
    TR_REQUIRE(src && dst && n > 0 && h >= 3 && w >= 3, TR_EINVAL, "tr_bayer_round_trip: bad arguments (3-channel fp32 NCHW, h and w >= 3)");
    const int64_t total = (int64_t)n * h * w;
    hipLaunchKernelGGL(bayer_round_trip_kernel, dim3(sweep_grid(total)), dim3(256), 0, (hipStream_t)stream, src, dst, h, w, total);
    return tr::launch_status("tr_bayer_round_trip");
}
