// Fused optimizer sweep over one flat fp32 parameter arena: clip scale + AdamW + EMA + zero_grad.
// Contract: include/trainner_hip.h (tr_adamw_ema).  Reference arithmetic replaced:
//   clip_grad_norm_                         traiNNer/models/sr_model.py:707-711
//   torch.optim.AdamW.step                  traiNNer/optimizers/__init__.py:30-47
//   ema_pytorch EMA.update (lerp)           traiNNer/models/sr_model.py:226-235, :826-828
//   optimizer.zero_grad()                   traiNNer/models/sr_model.py:719
// HBM-bound: 16 B read of p,g,m,v,ema and 16 B write of p,m,v,ema,(g) per 4 parameters
// = 36 B/param (vs ~6 separate sweeps in the reference).
#include "common.h"

namespace {

struct Hyper {
    float lr, b1, b2, eps, wd, bc1, rsqrt_bc2, ema_decay, clip, inv_scale;
};

__device__ __forceinline__ void upd(float& p, float g, float& m, float& v, const Hyper& h) {
    // torch/optim/adamw.py (single-tensor path): decoupled decay, lerp for exp_avg,
    // denom = sqrt(v)/sqrt(bc2) + eps, p -= lr/bc1 * m/denom
    p *= 1.f - h.lr * h.wd;
    m = m + (g - m) * (1.f - h.b1);
    v = v * h.b2 + (1.f - h.b2) * g * g;
    const float denom = sqrtf(v) * h.rsqrt_bc2 + h.eps;
    p -= (h.lr / h.bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, float* __restrict__ ema, int64_t numel, Hyper h,
                                                     const float* __restrict__ gnorm, int zero_grad) {
    float gs = h.inv_scale;                            // GradScaler.unscale_ (1 when no loss scaling is in use)
    // a non-finite gradient norm (NaN / Inf loss: the reference raises at sr_model.py:665-666 BEFORE backward; here the loss scalars are read
    // back lazily) must not reach the weights: p, m, v and the EMA stay as they are, only the gradients are cleared.  Wave-uniform branch.
    const bool poisoned = gnorm != nullptr && !isfinite(gnorm[0]);
    if (poisoned) {
        if (zero_grad) {
            for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (numel >> 2); i += (int64_t)gridDim.x * blockDim.x)
                ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (blockIdx.x == 0)
                for (int64_t i = ((numel >> 2) << 2) + threadIdx.x; i < numel; i += blockDim.x) g[i] = 0.f;
        }
        return;
    }
    if (h.clip > 0.f) {
        const float c = h.clip / (gnorm[0] * h.inv_scale + 1e-6f);   // clip_grad_norm_: clamp(max_norm / (total + 1e-6), max = 1)
        gs *= c < 1.f ? c : 1.f;
    }
    const float one_minus_decay = 1.f - h.ema_decay;
    const int64_t n4 = numel >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 P = ((float4*)p)[i], G = ((float4*)g)[i], M = ((float4*)m)[i], V = ((float4*)v)[i];
        G.x *= gs; G.y *= gs; G.z *= gs; G.w *= gs;
        upd(P.x, G.x, M.x, V.x, h); upd(P.y, G.y, M.y, V.y, h); upd(P.z, G.z, M.z, V.z, h); upd(P.w, G.w, M.w, V.w, h);
        ((float4*)p)[i] = P; ((float4*)m)[i] = M; ((float4*)v)[i] = V;
        if (ema) {
            float4 E = ((float4*)ema)[i];
            E.x += (P.x - E.x) * one_minus_decay; E.y += (P.y - E.y) * one_minus_decay;
            E.z += (P.z - E.z) * one_minus_decay; E.w += (P.w - E.w) * one_minus_decay;
            ((float4*)ema)[i] = E;
        }
        if (zero_grad) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (blockIdx.x == 0) {
        for (int64_t i = (n4 << 2) + threadIdx.x; i < numel; i += blockDim.x) {
            float P = p[i], G = g[i] * gs, M = m[i], V = v[i];
            upd(P, G, M, V, h);
            p[i] = P; m[i] = M; v[i] = V;
            if (ema) ema[i] += (P - ema[i]) * one_minus_decay;
            if (zero_grad) g[i] = 0.f;
        }
    }
}

}  // namespace

extern "C" int tr_adamw_ema(void* stream, const tr_adamw_desc* d) {
    TR_REQUIRE(d && d->p && d->g && d->m && d->v && d->numel > 0, TR_EINVAL, "tr_adamw_ema: bad arguments");
    TR_REQUIRE(d->clip <= 0.f || d->grad_norm, TR_EINVAL, "tr_adamw_ema: clipping needs grad_norm");
    TR_REQUIRE(((uintptr_t)d->p | (uintptr_t)d->g | (uintptr_t)d->m | (uintptr_t)d->v | (uintptr_t)d->ema) % 16 == 0, TR_EALIGN,
               "tr_adamw_ema: arenas must be 16-byte aligned");
    TR_REQUIRE(d->bias_corr1 > 0.f && d->bias_corr2 > 0.f, TR_EINVAL, "tr_adamw_ema: bias corrections must be positive");
    Hyper h;
    h.lr = d->lr; h.b1 = d->beta1; h.b2 = d->beta2; h.eps = d->eps; h.wd = d->weight_decay;
    h.bc1 = d->bias_corr1; h.rsqrt_bc2 = 1.f / sqrtf(d->bias_corr2); h.ema_decay = d->ema_decay; h.clip = d->clip;
    h.inv_scale = (d->inv_scale == 0.f) ? 1.f : d->inv_scale;
    int64_t b = (d->numel / 4 + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, d->p, d->g, d->m, d->v, d->ema, d->numel, h,
                       d->grad_norm, d->zero_grad);
    return tr::launch_status("tr_adamw_ema");
}
