// Pixel-loss heads: l1 / mse / charbonnier, reduction 'mean', fused forward + d/dpred.
// Contract: include/trainner_hip.h (tr_pixel_loss).  Reference arithmetic replaced:
//   l1_loss / mse_loss / charbonnier_loss   traiNNer/losses/basic_loss.py:16-28
//   mean reduction                          traiNNer/losses/loss_util.py:8-26
// HBM-bound: 16-byte loads of pred and target, one 16-byte (or 8-byte bf16) grad store, a wave
// shuffle + LDS block reduction, and a fixed-order second-stage sum (no float atomics).
#include "common.h"

namespace {

constexpr int MAX_BLOCKS = 1024;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (l == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    }
    return r;
}

template <int KIND>
__device__ __forceinline__ void elem(float d, float eps, float& val, float& dv) {
    if (KIND == TR_LOSS_L1) {
        val = fabsf(d);
        dv = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    } else if (KIND == TR_LOSS_MSE) {
        val = d * d;
        dv = 2.f * d;
    } else {
        val = sqrtf(d * d + eps);
        dv = d / val;
    }
}

__device__ __forceinline__ float4 ld4(const void* p, int dtype, int64_t i) {
    if (dtype == TR_BF16) {
        const uint2 u = *(const uint2*)((const unsigned short*)p + i);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xffff0000u));
    }
    return *(const float4*)((const float*)p + i);
}
__device__ __forceinline__ float ld1(const void* p, int dtype, int64_t i) {
    return dtype == TR_BF16 ? tr::bf16_to_f32(((const unsigned short*)p)[i]) : ((const float*)p)[i];
}
__device__ __forceinline__ void st1(void* p, int dtype, int64_t i, float v, int acc) {
    if (dtype == TR_BF16) {
        unsigned short* q = (unsigned short*)p + i;
        if (acc) v += tr::bf16_to_f32(*q);
        *q = tr::f32_to_bf16(v);
    } else {
        float* q = (float*)p + i;
        if (acc) v += *q;
        *q = v;
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void pixel_loss_kernel(const void* __restrict__ pred, const void* __restrict__ target, int dtype,
                                                          int64_t numel, float gscale, float eps, void* grad, int gdtype, int gacc,
                                                          float* __restrict__ partials) {
    __shared__ float sh[4];
    float sum = 0.f;
    const int64_t n4 = numel >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 p = ld4(pred, dtype, i * 4), t = ld4(target, dtype, i * 4);
        float v0, v1, v2, v3, g0, g1, g2, g3;
        elem<KIND>(p.x - t.x, eps, v0, g0); elem<KIND>(p.y - t.y, eps, v1, g1);
        elem<KIND>(p.z - t.z, eps, v2, g2); elem<KIND>(p.w - t.w, eps, v3, g3);
        sum += (v0 + v1) + (v2 + v3);
        if (grad) {
            st1(grad, gdtype, i * 4 + 0, g0 * gscale, gacc); st1(grad, gdtype, i * 4 + 1, g1 * gscale, gacc);
            st1(grad, gdtype, i * 4 + 2, g2 * gscale, gacc); st1(grad, gdtype, i * 4 + 3, g3 * gscale, gacc);
        }
    }
    // tail (numel % 4) handled by block 0
    if (blockIdx.x == 0) {
        for (int64_t i = (n4 << 2) + threadIdx.x; i < numel; i += blockDim.x) {
            float v, g;
            elem<KIND>(ld1(pred, dtype, i) - ld1(target, dtype, i), eps, v, g);
            sum += v;
            if (grad) st1(grad, gdtype, i, g * gscale, gacc);
        }
    }
    const float bs = block_sum(sum, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = bs;
}

// tr_pixel_loss on the fp32 NHWC activation that feeds a ReLU + 2x2 max pool, fused with the pool's backward: one thread per (window, channel
// quad) reads its four pixels of pred and target and the pooled gradient, writes grad = gscale * dcrit + route(gout) once.  The separate
// sequence (tr_maxpool2x2_bwd, then an accumulating tr_pixel_loss) moved 6.25 values per element, this one 3.25.
template <int KIND>
__global__ __launch_bounds__(256) void pixel_loss_pool_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, const float* __restrict__ gout,
                                                                   float* __restrict__ grad, int n, int h, int w, int c, float gscale, float eps,
                                                                   float* __restrict__ partials) {
    __shared__ float sh[4];
    float sum = 0.f;
    const int ho = h >> 1, wo = w >> 1, c4 = c >> 2;
    const int64_t total = (int64_t)n * ho * wo * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int cq = (int)(t % c4) * 4; t /= c4;
        const int x = (int)(t % wo); t /= wo;
        const int y = (int)(t % ho);
        const int nb = (int)(t / ho);
        const int64_t p00 = (((int64_t)nb * h + 2 * y) * w + 2 * x) * c + cq;
        const int64_t offs[4] = {p00, p00 + c, p00 + (int64_t)w * c, p00 + (int64_t)w * c + c};
        float4 a[4], tg[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = *(const float4*)(pred + offs[k]);
            tg[k] = *(const float4*)(target + offs[k]);
        }
        const float4 g = *(const float4*)(gout + (((int64_t)nb * ho + y) * wo + x) * c + cq);
        const float gv[4] = {g.x, g.y, g.z, g.w};
        float o[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float av[4] = {((const float*)&a[0])[e], ((const float*)&a[1])[e], ((const float*)&a[2])[e], ((const float*)&a[3])[e]};
            // first maximal element of relu(.) in window scan order takes the pooled gradient (ATen keeps the earliest index on ties); none at max <= 0
            int km = 0;
            float m = fmaxf(av[0], 0.f);
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const float r = fmaxf(av[k], 0.f);
                if (r > m) { m = r; km = k; }
            }
            const float gp = m > 0.f ? gv[e] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v, dv;
                elem<KIND>(av[k] - ((const float*)&tg[k])[e], eps, v, dv);
                sum += v;
                o[k][e] = dv * gscale + (k == km ? gp : 0.f);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) *(float4*)(grad + offs[k]) = make_float4(o[k][0], o[k][1], o[k][2], o[k][3]);
    }
    const float bs = block_sum(sum, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = bs;
}

__global__ __launch_bounds__(256) void finish_kernel(const float* __restrict__ partials, int nparts, float scale, float* out,
                                                     float scale2, float* out2, int take_sqrt) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += partials[i];
    const float t = block_sum(s, sh);
    if (threadIdx.x == 0) {
        out[0] = take_sqrt ? sqrtf(t) : t * scale;
        if (out2) out2[0] += t * scale2;
    }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t numel, float* __restrict__ partials) {
    __shared__ float sh[4];
    float sum = 0.f;
    const int64_t n4 = numel >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = *(const float4*)(x + i * 4);
        sum += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < numel; i += blockDim.x) sum += x[i] * x[i];
    const float bs = block_sum(sum, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = bs;
}

inline int blocks_for(int64_t numel) {
    int64_t b = (numel / 4 + 255) / 256;
    if (b > MAX_BLOCKS) b = MAX_BLOCKS;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int tr_pixel_loss(void* stream, const tr_pixel_loss_desc* d) {
    TR_REQUIRE(d && d->pred && d->target && d->value_out && d->partials && d->numel > 0, TR_EINVAL, "tr_pixel_loss: bad arguments");
    TR_REQUIRE(d->kind >= TR_LOSS_L1 && d->kind <= TR_LOSS_CHARBONNIER, TR_EINVAL, "tr_pixel_loss: unknown kind %d", d->kind);
    TR_REQUIRE(((uintptr_t)d->pred) % 16 == 0 && ((uintptr_t)d->target) % 16 == 0, TR_EALIGN,
               "tr_pixel_loss: pred/target must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int nb = blocks_for(d->numel);
    const float inv = 1.f / (float)d->numel;
    const float gscale = d->grad_weight * inv;
#define TR_LAUNCH_LOSS(K)                                                                                                          \
    hipLaunchKernelGGL(pixel_loss_kernel<K>, dim3(nb), dim3(256), 0, st, d->pred, d->target, d->dtype, d->numel, gscale, d->eps, d->grad, \
                       d->grad_dtype, d->accumulate_grad, d->partials)
    switch (d->kind) {
        case TR_LOSS_L1: TR_LAUNCH_LOSS(TR_LOSS_L1); break;
        case TR_LOSS_MSE: TR_LAUNCH_LOSS(TR_LOSS_MSE); break;
        default: TR_LAUNCH_LOSS(TR_LOSS_CHARBONNIER);
    }
#undef TR_LAUNCH_LOSS
    int rc = tr::launch_status("tr_pixel_loss");
    if (rc) return rc;
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, st, d->partials, nb, d->value_weight * inv, d->value_out,
                       d->total_weight * inv, d->total_out, 0);
    return tr::launch_status("tr_pixel_loss(finish)");
}

extern "C" int tr_pixel_loss_pool_bwd(void* stream, const tr_pixel_loss_desc* d, const float* gout, int n, int h, int w, int c) {
    TR_REQUIRE(d && d->pred && d->target && d->value_out && d->partials && d->grad && gout, TR_EINVAL, "tr_pixel_loss_pool_bwd: bad arguments");
    TR_REQUIRE(d->kind >= TR_LOSS_L1 && d->kind <= TR_LOSS_CHARBONNIER, TR_EINVAL, "tr_pixel_loss_pool_bwd: unknown kind %d", d->kind);
    TR_REQUIRE(d->dtype == TR_F32 && d->grad_dtype == TR_F32, TR_EUNSUPPORTED, "tr_pixel_loss_pool_bwd: fp32 activations and gradients only");
    TR_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && h % 2 == 0 && w % 2 == 0 && c % 4 == 0 && d->numel == (int64_t)n * h * w * c, TR_EINVAL,
               "tr_pixel_loss_pool_bwd: needs even h, w, channels in fours and numel = n*h*w*c");
    TR_REQUIRE(((uintptr_t)d->pred) % 16 == 0 && ((uintptr_t)d->target) % 16 == 0 && ((uintptr_t)d->grad) % 16 == 0 && ((uintptr_t)gout) % 16 == 0, TR_EALIGN,
               "tr_pixel_loss_pool_bwd: buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int nb = blocks_for(d->numel / 4);
    const float inv = 1.f / (float)d->numel;
    const float gscale = d->grad_weight * inv;
#define TR_LAUNCH_LOSSP(K)                                                                                                         \
    hipLaunchKernelGGL(pixel_loss_pool_bwd_kernel<K>, dim3(nb), dim3(256), 0, st, (const float*)d->pred, (const float*)d->target, gout, \
                       (float*)d->grad, n, h, w, c, gscale, d->eps, d->partials)
    switch (d->kind) {
        case TR_LOSS_L1: TR_LAUNCH_LOSSP(TR_LOSS_L1); break;
        case TR_LOSS_MSE: TR_LAUNCH_LOSSP(TR_LOSS_MSE); break;
        default: TR_LAUNCH_LOSSP(TR_LOSS_CHARBONNIER);
    }
#undef TR_LAUNCH_LOSSP
    int rc = tr::launch_status("tr_pixel_loss_pool_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, st, d->partials, nb, d->value_weight * inv, d->value_out, d->total_weight * inv, d->total_out, 0);
    return tr::launch_status("tr_pixel_loss_pool_bwd(finish)");
}

extern "C" int tr_sumsq(void* stream, const float* x, int64_t numel, float* norm_out, float* partials) {
    TR_REQUIRE(x && norm_out && partials && numel > 0, TR_EINVAL, "tr_sumsq: bad arguments");
    TR_REQUIRE(((uintptr_t)x) % 16 == 0, TR_EALIGN, "tr_sumsq: x must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int nb = blocks_for(numel);
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, st, x, numel, partials);
    int rc = tr::launch_status("tr_sumsq");
    if (rc) return rc;
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, st, partials, nb, 1.f, norm_out, 0.f, (float*)nullptr, 1);
    return tr::launch_status("tr_sumsq(finish)");
}
