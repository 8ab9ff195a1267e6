// nn.BatchNorm2d(affine=True) on NHWC channel-slice views: the normalisation layers of the reference's VGGStyleDiscriminator
// (traiNNer/archs/discriminator_arch.py:31-57: conv -> BatchNorm2d -> LeakyReLU(0.2), ten or twelve times), forward + backward.
//
// Contract: include/trainner_hip.h (tr_batchnorm_fwd / tr_batchnorm_bwd).  HBM-bound elementwise / reduction work (no MFMA):
//   forward, training:  pass 1  per-channel sum and sum of squares over N*H*W pixels -- per-block fp32 partials folded in DOUBLE (fixed order:
//                               reproducible; torch's CPU kernel is a two-pass / Welford fp32 reduction, the double fold keeps E[x^2] - E[x]^2 at
//                               fp32 rounding), finalize: mean, biased variance, rstd, running statistics (momentum, unbiased variance), a = gamma * rstd,
//                               b = beta - mean * a
//                       pass 2  y = act(a[c] * x + b[c])          (LeakyReLU fused: the activation follows every BatchNorm of the discriminator)
//   forward, eval:      a / b from the running statistics, pass 2 only
//   backward:           pass 1  g = gy * act'(y);  sum g, sum g * xhat per channel (same fold); finalize: dbeta, dgamma (+)=, dx coefficients
//                       pass 2  dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat))
// One thread owns a 4-channel quad of a pixel (16 B fp32 / 8 B 16-bit accesses, consecutive lanes = consecutive channels of a pixel record).
#include "common.h"

namespace {

constexpr int BN_THREADS = 256;
constexpr int BN_MAXBLK = 256;       // partial rows of pass 1

__device__ __forceinline__ float act_grad(float y, float slope) { return y > 0.f ? 1.f : slope; }

// pass 1 (both directions): per-block partial sums of (s0, s1) per channel.
//   MODE 0: s0 = x, s1 = x * x                               (forward statistics)
//   MODE 1: s0 = g, s1 = g * (x - mean) * rstd,  g = gy * act'(y)   (backward reductions)
template <int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_reduce_kernel(tr_view x, tr_view gy, tr_view y, int64_t npix, int c4, int vx, int vg, int vy, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float slope, double* __restrict__ partials) {
    __shared__ float red[BN_THREADS][8];
    const int tid = threadIdx.x;
    const int cq = tid % c4, lane = tid / c4, nlane = BN_THREADS / c4;      // c4 divides 256 or is larger (then nlane == 0 and quads are looped)
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (nlane > 0) {
        if (lane < nlane) {
            float4 m = make_float4(0.f, 0.f, 0.f, 0.f), r = make_float4(1.f, 1.f, 1.f, 1.f);
            if (MODE == 1) { m = *(const float4*)(mean + cq * 4); r = *(const float4*)(rstd + cq * 4); }
            for (int64_t p = (int64_t)blockIdx.x * nlane + lane; p < npix; p += (int64_t)gridDim.x * nlane) {
                const float4 v = tr::view_load4(x, p, cq * 4, vx);
                if (MODE == 0) {
                    s0[0] += v.x; s0[1] += v.y; s0[2] += v.z; s0[3] += v.w;
                    s1[0] += v.x * v.x; s1[1] += v.y * v.y; s1[2] += v.z * v.z; s1[3] += v.w * v.w;
                } else {
                    float4 g = tr::view_load4(gy, p, cq * 4, vg);
                    if (y.ptr) {
                        const float4 o = tr::view_load4(y, p, cq * 4, vy);
                        g.x *= act_grad(o.x, slope); g.y *= act_grad(o.y, slope); g.z *= act_grad(o.z, slope); g.w *= act_grad(o.w, slope);
                    }
                    s0[0] += g.x; s0[1] += g.y; s0[2] += g.z; s0[3] += g.w;
                    s1[0] += g.x * (v.x - m.x) * r.x; s1[1] += g.y * (v.y - m.y) * r.y; s1[2] += g.z * (v.z - m.z) * r.z; s1[3] += g.w * (v.w - m.w) * r.w;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[tid][k] = s0[k]; red[tid][4 + k] = s1[k]; }
        __syncthreads();
        if (tid < c4) {     // fold the pixel lanes of this block (fixed order), in double
            double d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int ln = 0; ln < nlane; ++ln)
#pragma unroll
                for (int k = 0; k < 8; ++k) d[k] += (double)red[ln * c4 + tid][k];
            double* dst = partials + ((int64_t)blockIdx.x * c4 + tid) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[k] = d[k];
        }
    } else {
        // more than 256 channel quads (c > 1024): not a shape of the discriminator; one quad loop per thread
        for (int q = tid; q < c4; q += BN_THREADS) {
            double d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            float4 m = make_float4(0.f, 0.f, 0.f, 0.f), r = make_float4(1.f, 1.f, 1.f, 1.f);
            if (MODE == 1) { m = *(const float4*)(mean + q * 4); r = *(const float4*)(rstd + q * 4); }
            for (int64_t p = blockIdx.x; p < npix; p += gridDim.x) {
                const float4 v = tr::view_load4(x, p, q * 4, vx);
                if (MODE == 0) {
                    d[0] += v.x; d[1] += v.y; d[2] += v.z; d[3] += v.w;
                    d[4] += (double)v.x * v.x; d[5] += (double)v.y * v.y; d[6] += (double)v.z * v.z; d[7] += (double)v.w * v.w;
                } else {
                    float4 g = tr::view_load4(gy, p, q * 4, vg);
                    if (y.ptr) {
                        const float4 o = tr::view_load4(y, p, q * 4, vy);
                        g.x *= act_grad(o.x, slope); g.y *= act_grad(o.y, slope); g.z *= act_grad(o.z, slope); g.w *= act_grad(o.w, slope);
                    }
                    d[0] += g.x; d[1] += g.y; d[2] += g.z; d[3] += g.w;
                    d[4] += g.x * (v.x - m.x) * r.x; d[5] += g.y * (v.y - m.y) * r.y; d[6] += g.z * (v.z - m.z) * r.z; d[7] += g.w * (v.w - m.w) * r.w;
                }
            }
            double* dst = partials + ((int64_t)blockIdx.x * c4 + q) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[k] = d[k];
        }
    }
}

// forward finalize: one thread per channel
__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(const double* __restrict__ partials, int nblk, int c, double npix, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              float momentum, float eps, float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                              float* __restrict__ ab) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const int q = ch >> 2, k = ch & 3, c4 = c >> 2;
    double s = 0, ss = 0;
    for (int b = 0; b < nblk; ++b) {
        const double* p = partials + ((int64_t)b * c4 + q) * 8;
        s += p[k];
        ss += p[4 + k];
    }
    const double mean = s / npix;
    double var = ss / npix - mean * mean;              // biased (what normalises the batch)
    var = var > 0 ? var : 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[ch] = (float)mean;
    save_rstd[ch] = rstd;
    if (running_mean) running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)mean;
    if (running_var) running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)(var * (npix > 1 ? npix / (npix - 1) : 1.0));   // unbiased
    const float a = gamma[ch] * rstd;
    ab[ch] = a;
    ab[c + ch] = beta[ch] - (float)mean * a;
}

__global__ __launch_bounds__(256) void bn_eval_coef_kernel(int c, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ running_mean,
                                                           const float* __restrict__ running_var, float eps, float* __restrict__ ab) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const float a = gamma[ch] / sqrtf(running_var[ch] + eps);
    ab[ch] = a;
    ab[c + ch] = beta[ch] - running_mean[ch] * a;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(tr_view x, tr_view y, int64_t npix, int c4, int vx, int vy, const float* __restrict__ ab, int c, int act,
                                                       float slope) {
    const int64_t total = npix * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / c4;
        const int ch = (int)(i - p * c4) * 4;
        const float4 v = tr::view_load4(x, p, ch, vx);
        const float4 a = *(const float4*)(ab + ch), b = *(const float4*)(ab + c + ch);
        float4 o = make_float4(fmaf(a.x, v.x, b.x), fmaf(a.y, v.y, b.y), fmaf(a.z, v.z, b.z), fmaf(a.w, v.w, b.w));
        if (act == TR_ACT_LRELU) {
            o.x *= o.x > 0.f ? 1.f : slope; o.y *= o.y > 0.f ? 1.f : slope; o.z *= o.z > 0.f ? 1.f : slope; o.w *= o.w > 0.f ? 1.f : slope;
        } else if (act == TR_ACT_RELU) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        tr::view_store4(y, p, ch, o, vy);
    }
}

// backward finalize: dbeta / dgamma and the three dx coefficients per channel:  dx = k0 * g + k1 * xhat_raw + k2  with xhat = (x - mean) * rstd
//   dx = gamma * rstd * (g - sg / n - xhat * sgx / n)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ partials, int nblk, int c, double npix, const float* __restrict__ gamma,
                                                              const float* __restrict__ rstd, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                              float* __restrict__ coef) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const int q = ch >> 2, k = ch & 3, c4 = c >> 2;
    double sg = 0, sgx = 0;
    for (int b = 0; b < nblk; ++b) {
        const double* p = partials + ((int64_t)b * c4 + q) * 8;
        sg += p[k];
        sgx += p[4 + k];
    }
    if (dbeta) dbeta[ch] = (accumulate ? dbeta[ch] : 0.f) + (float)sg;
    if (dgamma) dgamma[ch] = (accumulate ? dgamma[ch] : 0.f) + (float)sgx;
    const float gr = gamma[ch] * rstd[ch];
    coef[ch] = gr;
    coef[c + ch] = (float)(sg / npix);
    coef[2 * c + ch] = (float)(sgx / npix);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(tr_view gy, tr_view x, tr_view y, tr_view dx, int64_t npix, int c4, int vg, int vx, int vy, int vd,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ coef, int c,
                                                           float slope) {
    const int64_t total = npix * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / c4;
        const int ch = (int)(i - p * c4) * 4;
        float4 g = tr::view_load4(gy, p, ch, vg);
        if (y.ptr) {
            const float4 o = tr::view_load4(y, p, ch, vy);
            g.x *= act_grad(o.x, slope); g.y *= act_grad(o.y, slope); g.z *= act_grad(o.z, slope); g.w *= act_grad(o.w, slope);
        }
        const float4 v = tr::view_load4(x, p, ch, vx);
        const float4 m = *(const float4*)(mean + ch), r = *(const float4*)(rstd + ch);
        const float4 gr = *(const float4*)(coef + ch), mg = *(const float4*)(coef + c + ch), mgx = *(const float4*)(coef + 2 * c + ch);
        float4 d;
        d.x = gr.x * (g.x - mg.x - (v.x - m.x) * r.x * mgx.x);
        d.y = gr.y * (g.y - mg.y - (v.y - m.y) * r.y * mgx.y);
        d.z = gr.z * (g.z - mg.z - (v.z - m.z) * r.z * mgx.z);
        d.w = gr.w * (g.w - mg.w - (v.w - m.w) * r.w * mgx.w);
        tr::view_store4(dx, p, ch, d, vd);
    }
}

inline unsigned grid_for(int64_t work) {
    const int64_t b = (work + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
inline int reduce_blocks(int64_t npix, int c4) {
    const int nlane = c4 <= BN_THREADS ? BN_THREADS / c4 : 1;
    int64_t b = (npix + (int64_t)nlane * 8 - 1) / ((int64_t)nlane * 8);      // at least ~8 pixels per lane
    if (b < 1) b = 1;
    return (int)(b > BN_MAXBLK ? BN_MAXBLK : b);
}
inline bool bn_c_ok(int c) { return c > 0 && c % 4 == 0; }

}  // namespace

extern "C" size_t tr_batchnorm_workspace_bytes(int c) {
    if (c <= 0) return 0;
    return (size_t)BN_MAXBLK * (size_t)((c + 3) / 4) * 8 * sizeof(double) + (size_t)3 * c * sizeof(float) + 64;
}

extern "C" int tr_batchnorm_fwd(void* stream, tr_view x, tr_view y, int64_t npix, int c, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, float momentum, float eps, int training, int act, float slope, float* save_mean, float* save_rstd,
                                void* workspace, size_t workspace_bytes) {
    TR_REQUIRE(x.ptr && y.ptr && gamma && beta && npix > 0 && bn_c_ok(c) && x.c == c && y.c == c, TR_EINVAL,
               "tr_batchnorm_fwd: views of c = %d channels (a multiple of 4), gamma and beta are required", c);
    TR_REQUIRE(act == TR_ACT_NONE || act == TR_ACT_LRELU || act == TR_ACT_RELU, TR_EUNSUPPORTED, "tr_batchnorm_fwd: activation %d is not built (none, LeakyReLU, ReLU)", act);
    TR_REQUIRE(workspace && workspace_bytes >= tr_batchnorm_workspace_bytes(c) && ((uintptr_t)workspace) % 16 == 0, TR_EINVAL,
               "tr_batchnorm_fwd: workspace of tr_batchnorm_workspace_bytes(c) bytes, 16-byte aligned");
    TR_REQUIRE(training ? (save_mean && save_rstd) : (running_mean && running_var), TR_EINVAL,
               "tr_batchnorm_fwd: training needs save_mean / save_rstd, evaluation needs the running statistics");
    hipStream_t st = (hipStream_t)stream;
    const int c4 = c / 4;
    double* partials = (double*)workspace;
    float* ab = (float*)((char*)workspace + (size_t)BN_MAXBLK * c4 * 8 * sizeof(double));
    const int vx = (int)tr::view_vec4_ok(x), vy = (int)tr::view_vec4_ok(y);
    if (training) {
        const int nblk = reduce_blocks(npix, c4);
        hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(nblk), dim3(BN_THREADS), 0, st, x, tr_view{}, tr_view{}, npix, c4, vx, 0, 0, nullptr, nullptr, 0.f, partials);
        hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((c + 255) / 256), dim3(256), 0, st, partials, nblk, c, (double)npix, gamma, beta, running_mean, running_var,
                           momentum, eps, save_mean, save_rstd, ab);
    } else {
        hipLaunchKernelGGL(bn_eval_coef_kernel, dim3((c + 255) / 256), dim3(256), 0, st, c, gamma, beta, running_mean, running_var, eps, ab);
    }
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(npix * c4)), dim3(256), 0, st, x, y, npix, c4, vx, vy, ab, c, act, slope);
    return tr::launch_status("tr_batchnorm_fwd");
}

extern "C" int tr_batchnorm_bwd(void* stream, tr_view gy, tr_view x, tr_view y, tr_view dx, int64_t npix, int c, const float* gamma, const float* save_mean,
                                const float* save_rstd, float slope, float* dgamma, float* dbeta, int accumulate, void* workspace, size_t workspace_bytes) {
    TR_REQUIRE(gy.ptr && x.ptr && dx.ptr && gamma && save_mean && save_rstd && npix > 0 && bn_c_ok(c) && gy.c == c && x.c == c && dx.c == c && (!y.ptr || y.c == c),
               TR_EINVAL, "tr_batchnorm_bwd: views of c = %d channels, gamma and the saved statistics are required", c);
    TR_REQUIRE(workspace && workspace_bytes >= tr_batchnorm_workspace_bytes(c) && ((uintptr_t)workspace) % 16 == 0, TR_EINVAL,
               "tr_batchnorm_bwd: workspace of tr_batchnorm_workspace_bytes(c) bytes, 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int c4 = c / 4;
    double* partials = (double*)workspace;
    float* coef = (float*)((char*)workspace + (size_t)BN_MAXBLK * c4 * 8 * sizeof(double));
    const int vg = (int)tr::view_vec4_ok(gy), vx = (int)tr::view_vec4_ok(x), vy = (int)tr::view_vec4_ok(y), vd = (int)tr::view_vec4_ok(dx);
    const int nblk = reduce_blocks(npix, c4);
    hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3(nblk), dim3(BN_THREADS), 0, st, x, gy, y, npix, c4, vx, vg, vy, save_mean, save_rstd, slope, partials);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((c + 255) / 256), dim3(256), 0, st, partials, nblk, c, (double)npix, gamma, save_rstd, dgamma, dbeta, accumulate,
                       coef);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(npix * c4)), dim3(256), 0, st, gy, x, y, dx, npix, c4, vg, vx, vy, vd, save_mean, save_rstd, coef, c, slope);
    return tr::launch_status("tr_batchnorm_bwd");
}
