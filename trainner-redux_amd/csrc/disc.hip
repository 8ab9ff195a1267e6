// Kernels specific to the U-Net discriminator and the GAN loss head.
//
// Contract: include/trainner_hip.h.  Reference arithmetic replaced:
//   UNetDiscriminatorSN.forward            traiNNer/archs/discriminator_arch.py:131-160
//     4x4 stride-2 pad-1 convs (:119-121)  -> space-to-depth(2) + the 3x3 MFMA conv on expanded weights (tr_expand_w4x4s2)
//     F.interpolate(x2, bilinear, align_corners=False) (:139,144,149) -> tr_bilinear_up2x (+ adjoint)
//     torch.nn.utils.spectral_norm (:115-128, old-style hook: one power iteration per training forward) -> tr_spectral_norm(_bwd)
//   GANLoss.forward                        traiNNer/losses/gan_loss.py:99-126
// All HBM-bound element-wise / small reduction kernels; the heavy lifting stays in conv3x3.hip / wgrad3x3.hip.
#include "common.h"

namespace {

inline unsigned grid_for(int64_t work, int cap = 256 * 16) {
    int64_t b = (work + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ---- space-to-depth / depth-to-space (block 2), channel order (dy, dx, c) -----------------------------------------------
__global__ __launch_bounds__(256) void s2d_kernel(tr_view in, tr_view out, int n, int h, int w, int c4, int to_depth, int vi, int vo) {
    const int ho = h >> 1, wo = w >> 1;
    const int64_t total = (int64_t)n * h * w * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int c = (int)(t % c4) * 4; t /= c4;
        const int x = (int)(t % w); t /= w;
        const int y = (int)(t % h);
        const int nb = (int)(t / h);
        const int64_t pf = ((int64_t)nb * h + y) * w + x;                       // full-resolution pixel
        const int64_t pc = ((int64_t)nb * ho + (y >> 1)) * wo + (x >> 1);       // cell
        const int cc = ((y & 1) * 2 + (x & 1)) * (c4 * 4) + c;                  // channel inside the cell
        if (to_depth) tr::view_store4(out, pc, cc, tr::view_load4(in, pf, c, vi), vo);
        else tr::view_store4(out, pf, c, tr::view_load4(in, pc, cc, vi), vo);
    }
}

// ---- bilinear x2, align_corners = False --------------------------------------------------------------------------------
// out[2y+a] = a == 0 ? .25*in[max(y-1,0)] + .75*in[y] : .75*in[y] + .25*in[min(y+1,H-1)]   (separable)
__global__ __launch_bounds__(256) void bilinear_up_kernel(tr_view in, tr_view out, int n, int h, int w, int c4, int vi, int vo) {
    const int H = h * 2, W = w * 2;
    const int64_t total = (int64_t)n * H * W * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int c = (int)(t % c4) * 4; t /= c4;
        const int X = (int)(t % W); t /= W;
        const int Y = (int)(t % H);
        const int nb = (int)(t / H);
        const int y = Y >> 1, x = X >> 1;
        const int y2 = (Y & 1) ? min(y + 1, h - 1) : max(y - 1, 0);
        const int x2 = (X & 1) ? min(x + 1, w - 1) : max(x - 1, 0);
        const int64_t base = (int64_t)nb * h;
        const float4 a = tr::view_load4(in, (base + y) * w + x, c, vi), b = tr::view_load4(in, (base + y) * w + x2, c, vi);
        const float4 d = tr::view_load4(in, (base + y2) * w + x, c, vi), e = tr::view_load4(in, (base + y2) * w + x2, c, vi);
        float4 r;
        r.x = 0.75f * (0.75f * a.x + 0.25f * b.x) + 0.25f * (0.75f * d.x + 0.25f * e.x);
        r.y = 0.75f * (0.75f * a.y + 0.25f * b.y) + 0.25f * (0.75f * d.y + 0.25f * e.y);
        r.z = 0.75f * (0.75f * a.z + 0.25f * b.z) + 0.25f * (0.75f * d.z + 0.25f * e.z);
        r.w = 0.75f * (0.75f * a.w + 0.25f * b.w) + 0.25f * (0.75f * d.w + 0.25f * e.w);
        tr::view_store4(out, ((int64_t)nb * H + Y) * W + X, c, r, vo);
    }
}

// adjoint (gather form): gin[y] = sum over the (at most 4) output rows that read in[y], border taps folded by the clamp
__global__ __launch_bounds__(256) void bilinear_up_bwd_kernel(tr_view gout, tr_view gin, int n, int h, int w, int c4, int vg, int vi) {
    const int H = h * 2, W = w * 2;
    const int64_t total = (int64_t)n * h * w * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int c = (int)(t % c4) * 4; t /= c4;
        const int x = (int)(t % w); t /= w;
        const int y = (int)(t % h);
        const int nb = (int)(t / h);
        float wy[4], wx[4];
        int ry[4], rx[4];
        ry[0] = 2 * y - 1; wy[0] = y >= 1 ? 0.25f : 0.f;
        ry[1] = 2 * y;     wy[1] = 0.75f + (y == 0 ? 0.25f : 0.f);
        ry[2] = 2 * y + 1; wy[2] = 0.75f + (y == h - 1 ? 0.25f : 0.f);
        ry[3] = 2 * y + 2; wy[3] = y + 1 <= h - 1 ? 0.25f : 0.f;
        rx[0] = 2 * x - 1; wx[0] = x >= 1 ? 0.25f : 0.f;
        rx[1] = 2 * x;     wx[1] = 0.75f + (x == 0 ? 0.25f : 0.f);
        rx[2] = 2 * x + 1; wx[2] = 0.75f + (x == w - 1 ? 0.25f : 0.f);
        rx[3] = 2 * x + 2; wx[3] = x + 1 <= w - 1 ? 0.25f : 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (wy[a] == 0.f) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (wx[b] == 0.f) continue;
                const float4 g = tr::view_load4(gout, ((int64_t)nb * H + ry[a]) * W + rx[b], c, vg);
                const float k = wy[a] * wx[b];
                acc.x += k * g.x; acc.y += k * g.y; acc.z += k * g.z; acc.w += k * g.w;
            }
        }
        tr::view_store4(gin, ((int64_t)nb * h + y) * w + x, c, acc, vi);
    }
}

// ---- 4x4 stride-2 pad-1 weights  <->  3x3 weights over the space-to-depth(2) input ----------------------------------------
// w3[co][(dy,dx,ci)][ky][kx] = w4[co][ci][2ky+dy-1][2kx+dx-1]  (zero when the 4x4 index falls outside 0..3)
__global__ __launch_bounds__(256) void expand_w4_kernel(const float* __restrict__ w4, float* __restrict__ w3, int cout, int cin, float scale_dummy) {
    const int64_t total = (int64_t)cout * 4 * cin * 9;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int kx = (int)(t % 3); t /= 3;
        const int ky = (int)(t % 3); t /= 3;
        const int cc = (int)(t % (4 * cin));
        const int co = (int)(t / (4 * cin));
        const int sub = cc / cin, ci = cc - sub * cin;
        const int ii = 2 * ky + (sub >> 1) - 1, jj = 2 * kx + (sub & 1) - 1;
        w3[i] = (ii >= 0 && ii < 4 && jj >= 0 && jj < 4) ? w4[(((int64_t)co * cin + ci) * 4 + ii) * 4 + jj] : 0.f;
    }
}
__global__ __launch_bounds__(256) void collapse_g3_kernel(const float* __restrict__ g3, float* __restrict__ g4, int cout, int cin, int accumulate) {
    const int64_t total = (int64_t)cout * cin * 16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int jj = (int)(t % 4); t /= 4;
        const int ii = (int)(t % 4); t /= 4;
        const int ci = (int)(t % cin);
        const int co = (int)(t / cin);
        const int dy = (ii + 1) & 1, ky = (ii + 1) >> 1, dx = (jj + 1) & 1, kx = (jj + 1) >> 1;
        const float v = g3[(((int64_t)co * 4 * cin + (dy * 2 + dx) * cin + ci) * 3 + ky) * 3 + kx];
        g4[i] = accumulate ? g4[i] + v : v;
    }
}

// ---- spectral norm ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float block_sum_bcast(float v, float* sh) {
    v = wave_sum(v);
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (l == 0) sh[wv] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    return r;
}
// t[col] = sum_r W[r][col] * u[r].  A block owns 16 columns x 16 row lanes (lane k sums rows k, k + 16, ...; 64 contiguous bytes per row segment), the
// lanes are folded through LDS in a fixed order: cols / 16 blocks instead of cols / 256 (one thread per column walked all rows: 16 blocks and a 512-deep
// dependent chain for the 512 x 4096 layer, 72 us per power iteration; round 3)
__global__ __launch_bounds__(256) void matvec_t_kernel(const float* __restrict__ w, const float* __restrict__ u, float* __restrict__ t, int rows, int cols) {
    __shared__ float part[16][17];
    const int c = threadIdx.x & 15, k = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + c;
    float s = 0.f;
    if (col < cols)
        for (int r = k; r < rows; r += 16) s += w[(int64_t)r * cols + col] * u[r];
    part[k][c] = s;
    __syncthreads();
    if (k == 0 && col < cols) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) a += part[j][c];
        t[col] = a;
    }
}
// s[row] = sum_c W[row][c] * v[c]        (one block per row)
__global__ __launch_bounds__(256) void matvec_kernel(const float* __restrict__ w, const float* __restrict__ v, float* __restrict__ s, int cols) {
    __shared__ float sh[4];
    const float* wr = w + (int64_t)blockIdx.x * cols;
    float a = 0.f;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) a += wr[c] * v[c];
    const float r = block_sum_bcast(a, sh);
    if (threadIdx.x == 0) s[blockIdx.x] = r;
}
// dst = src / max(||src||, eps)          (single block)
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, float eps) {
    __shared__ float sh[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += src[i] * src[i];
    const float nrm = fmaxf(sqrtf(block_sum_bcast(a, sh)), eps);
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i] / nrm;
}
// sigma = dot(a, b)                      (single block)
__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += a[i] * b[i];
    const float r = block_sum_bcast(s, sh);
    if (threadIdx.x == 0) out[0] = r;
}
__global__ __launch_bounds__(256) void scale_by_inv_kernel(const float* __restrict__ w, const float* __restrict__ sigma, float* __restrict__ out, int64_t n) {
    const float inv = 1.f / sigma[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = w[i] * inv;
}
// partial dot products <G, Wsn> per block
__global__ __launch_bounds__(256) void dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ partials, int64_t n) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += a[i] * b[i];
    const float r = block_sum_bcast(s, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}
// g_orig += (G - <G,Wsn> u v^T) / sigma
__global__ __launch_bounds__(256) void sn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ u, const float* __restrict__ v,
                                                      const float* __restrict__ sigma, const float* __restrict__ partials, int nparts,
                                                      float* __restrict__ g_orig, int rows, int cols) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += partials[i];
    const float d = block_sum_bcast(s, sh);
    const float inv = 1.f / sigma[0];
    const int64_t n = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        g_orig[i] += (g[i] - d * u[r] * v[c]) * inv;
    }
}

// ---- GAN loss ----------------------------------------------------------------------------------------------------------
// mode: 0 vanilla (BCE with logits vs label), 1 lsgan (MSE vs label), 2 linear (sign * x), 3 hinge-D (relu(1 + sign*x)),
//       4 softplus(sign * x)
__device__ __forceinline__ void gan_elem(int mode, float x, float label, float sign, float& val, float& dv) {
    if (mode == 0) {
        val = fmaxf(x, 0.f) - x * label + log1pf(expf(-fabsf(x)));
        dv = 1.f / (1.f + expf(-x)) - label;
    } else if (mode == 1) {
        const float d = x - label;
        val = d * d; dv = 2.f * d;
    } else if (mode == 2) {
        val = sign * x; dv = sign;
    } else if (mode == 3) {
        const float z = 1.f + sign * x;
        val = fmaxf(z, 0.f); dv = z > 0.f ? sign : 0.f;
    } else {
        const float z = sign * x;
        val = fmaxf(z, 0.f) + log1pf(expf(-fabsf(z)));
        dv = sign / (1.f + expf(-z));
    }
}
__global__ __launch_bounds__(256) void gan_loss_kernel(const float* __restrict__ x, int64_t n, int mode, float label, float sign, float gscale,
                                                        float* __restrict__ grad, float* __restrict__ partials) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v, d;
        gan_elem(mode, x[i], label, sign, v, d);
        s += v;
        if (grad) grad[i] = d * gscale;
    }
    const float r = block_sum_bcast(s, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}
__global__ __launch_bounds__(256) void gan_finish_kernel(const float* __restrict__ partials, int nparts, float vscale, float* value_out, float tscale,
                                                          float* total_out) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += partials[i];
    const float t = block_sum_bcast(s, sh);
    if (threadIdx.x == 0) {
        value_out[0] = t * vscale;
        if (total_out) total_out[0] += t * tscale;
    }
}

}  // namespace

extern "C" int tr_space_to_depth2(void* stream, tr_view in, tr_view out, int n, int h, int w) {
    TR_REQUIRE(in.ptr && out.ptr && n > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0 && in.c % 4 == 0 && out.c == 4 * in.c, TR_EINVAL,
               "tr_space_to_depth2: need even dims, in.c %% 4 == 0 and out.c == 4 * in.c");
    hipLaunchKernelGGL(s2d_kernel, dim3(grid_for((int64_t)n * h * w * (in.c / 4))), dim3(256), 0, (hipStream_t)stream, in, out, n, h, w, in.c / 4, 1,
                       (int)tr::view_vec4_ok(in), (int)tr::view_vec4_ok(out));
    return tr::launch_status("tr_space_to_depth2");
}
extern "C" int tr_depth_to_space2(void* stream, tr_view in, tr_view out, int n, int h, int w) {
    TR_REQUIRE(in.ptr && out.ptr && n > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0 && out.c % 4 == 0 && in.c == 4 * out.c, TR_EINVAL,
               "tr_depth_to_space2: need even (full-resolution) dims, out.c %% 4 == 0 and in.c == 4 * out.c");
    hipLaunchKernelGGL(s2d_kernel, dim3(grid_for((int64_t)n * h * w * (out.c / 4))), dim3(256), 0, (hipStream_t)stream, in, out, n, h, w, out.c / 4,
                       0, (int)tr::view_vec4_ok(in), (int)tr::view_vec4_ok(out));
    return tr::launch_status("tr_depth_to_space2");
}
extern "C" int tr_bilinear_up2x(void* stream, tr_view in, tr_view out, int n, int h_in, int w_in) {
    TR_REQUIRE(in.ptr && out.ptr && n > 0 && h_in > 0 && w_in > 0 && in.c == out.c && in.c % 4 == 0, TR_EINVAL, "tr_bilinear_up2x: channel mismatch");
    hipLaunchKernelGGL(bilinear_up_kernel, dim3(grid_for((int64_t)n * h_in * w_in * in.c)), dim3(256), 0, (hipStream_t)stream, in, out, n, h_in, w_in,
                       in.c / 4, (int)tr::view_vec4_ok(in), (int)tr::view_vec4_ok(out));
    return tr::launch_status("tr_bilinear_up2x");
}
extern "C" int tr_bilinear_up2x_bwd(void* stream, tr_view gout, tr_view gin, int n, int h_in, int w_in) {
    TR_REQUIRE(gout.ptr && gin.ptr && n > 0 && h_in > 0 && w_in > 0 && gin.c == gout.c && gin.c % 4 == 0, TR_EINVAL,
               "tr_bilinear_up2x_bwd: channel mismatch");
    hipLaunchKernelGGL(bilinear_up_bwd_kernel, dim3(grid_for((int64_t)n * h_in * w_in * (gin.c / 4))), dim3(256), 0, (hipStream_t)stream, gout, gin, n,
                       h_in, w_in, gin.c / 4, (int)tr::view_vec4_ok(gout), (int)tr::view_vec4_ok(gin));
    return tr::launch_status("tr_bilinear_up2x_bwd");
}
extern "C" int tr_expand_w4x4s2(void* stream, const float* w4, float* w3, int cout, int cin) {
    TR_REQUIRE(w4 && w3 && cout > 0 && cin > 0, TR_EINVAL, "tr_expand_w4x4s2: bad arguments");
    hipLaunchKernelGGL(expand_w4_kernel, dim3(grid_for((int64_t)cout * 4 * cin * 9)), dim3(256), 0, (hipStream_t)stream, w4, w3, cout, cin, 1.f);
    return tr::launch_status("tr_expand_w4x4s2");
}
extern "C" int tr_collapse_g3x3s2d(void* stream, const float* g3, float* g4, int cout, int cin, int accumulate) {
    TR_REQUIRE(g3 && g4 && cout > 0 && cin > 0, TR_EINVAL, "tr_collapse_g3x3s2d: bad arguments");
    hipLaunchKernelGGL(collapse_g3_kernel, dim3(grid_for((int64_t)cout * cin * 16)), dim3(256), 0, (hipStream_t)stream, g3, g4, cout, cin, accumulate);
    return tr::launch_status("tr_collapse_g3x3s2d");
}

extern "C" int tr_spectral_norm(void* stream, const float* w_orig, float* u, float* v, float* w_out, float* sigma_out, int rows, int cols,
                                int power_iteration, float eps, float* scratch) {
    TR_REQUIRE(w_orig && u && v && w_out && sigma_out && scratch && rows > 0 && cols > 0, TR_EINVAL, "tr_spectral_norm: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    float* t = scratch;            // cols floats
    float* s = scratch + cols;     // rows floats
    if (power_iteration) {
        hipLaunchKernelGGL(matvec_t_kernel, dim3((cols + 15) / 16), dim3(256), 0, st, w_orig, u, t, rows, cols);
        hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(256), 0, st, t, v, cols, eps);
        hipLaunchKernelGGL(matvec_kernel, dim3(rows), dim3(256), 0, st, w_orig, v, s, cols);
        hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(256), 0, st, s, u, rows, eps);
    } else {
        hipLaunchKernelGGL(matvec_kernel, dim3(rows), dim3(256), 0, st, w_orig, v, s, cols);
    }
    hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(256), 0, st, u, s, sigma_out, rows);
    hipLaunchKernelGGL(scale_by_inv_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, st, w_orig, sigma_out, w_out, (int64_t)rows * cols);
    return tr::launch_status("tr_spectral_norm");
}

extern "C" int tr_spectral_norm_bwd(void* stream, const float* g_wsn, const float* w_sn, const float* u, const float* v, const float* sigma,
                                    float* g_worig, int rows, int cols, float* scratch) {
    TR_REQUIRE(g_wsn && w_sn && u && v && sigma && g_worig && scratch && rows > 0 && cols > 0, TR_EINVAL, "tr_spectral_norm_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)rows * cols;
    const int nb = (int)grid_for(n, 256);
    hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(256), 0, st, g_wsn, w_sn, scratch, n);
    hipLaunchKernelGGL(sn_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, g_wsn, u, v, sigma, scratch, nb, g_worig, rows, cols);
    return tr::launch_status("tr_spectral_norm_bwd");
}

extern "C" int tr_gan_loss(void* stream, const tr_gan_loss_desc* d) {
    TR_REQUIRE(d && d->logits && d->value_out && d->partials && d->numel > 0, TR_EINVAL, "tr_gan_loss: bad arguments");
    int mode;
    float label = d->target_is_real ? d->real_label : d->fake_label;
    float sign = 1.f;
    switch (d->gan_type) {
        case TR_GAN_VANILLA: mode = 0; break;
        case TR_GAN_LSGAN: mode = 1; break;
        case TR_GAN_WGAN: mode = 2; sign = d->target_is_real ? -1.f : 1.f; break;
        case TR_GAN_WGAN_SOFTPLUS: mode = 4; sign = d->target_is_real ? -1.f : 1.f; break;
        case TR_GAN_HINGE:
            if (d->is_disc) { mode = 3; sign = d->target_is_real ? -1.f : 1.f; }
            else { mode = 2; sign = -1.f; }     // generator: -mean(pred), whatever target_is_real says (gan_loss.py:118-119)
            break;
        default:
            TR_REQUIRE(false, TR_EINVAL, "tr_gan_loss: unknown gan_type %d", d->gan_type);
    }
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)grid_for(d->numel, 1024);
    const float inv = 1.f / (float)d->numel;
    hipLaunchKernelGGL(gan_loss_kernel, dim3(nb), dim3(256), 0, st, d->logits, d->numel, mode, label, sign, d->grad_weight * inv, d->grad, d->partials);
    hipLaunchKernelGGL(gan_finish_kernel, dim3(1), dim3(256), 0, st, d->partials, nb, d->value_weight * inv, d->value_out, d->total_weight * inv,
                       d->total_out);
    return tr::launch_status("tr_gan_loss");
}
