// Token-path kernels of the SwinIR block: LayerNorm, window attention (W-MSA / SW-MSA), PixelShuffle.
//
// Contract: include/trainner_hip.h.  Reference arithmetic replaced (in-tree twin of spandrel's SwinIR):
//   nn.LayerNorm(dim)                          traiNNer/archs/drct_arch.py:464,476,658-662
//   WindowAttention.forward                    traiNNer/archs/drct_arch.py:175-223  (q*scale, q@k^T, + relative-position bias
//                                              gathered from the (2w-1)^2 x heads table, + shift mask (0/-100, :492-521), softmax, @v)
//   torch.roll +-shift, window_partition/reverse  :78-109, :532-568  -> pure index arithmetic here (no copies)
//   nn.PixelShuffle                            :714-717
// Mapping to gfx950: one wave per (window, head); lane = query token.  A 64-token window row of scores lives in the lane's
// registers, so the softmax max/sum are lane-local (no cross-lane traffic at all); K and V sit in LDS and are read as broadcasts.
// The contractions here are 6 % of SwinIR's FLOPs and run on the VALU in fp32; the Linear layers
// (71 %) and convs (22 %) are on MFMA (conv3x3.hip, ksize 1 / 3).
#include "attn.h"

namespace {

using tr::AttnArgs;
constexpr int WS = 8, WT = 64, HDMAX = 32;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- LayerNorm: one wave per token, lane owns channels lane, lane+64, ... ------------------------------------------------
constexpr int LN_MAXK = 12;   // C <= 768
constexpr int LN_BWD_BLOCKS = 1024;

template <int NK>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(tr_view x, tr_view y, int64_t ntok, int c, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, float* __restrict__ mean_out,
                                                             float* __restrict__ rstd_out) {
    const int l = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    float gm[NK], bt[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int ch = l + 64 * k;
        gm[k] = ch < c ? gamma[ch] : 0.f;
        bt[k] = ch < c ? beta[ch] : 0.f;
    }
    for (int64_t t = wave; t < ntok; t += nwaves) {
        float v[NK];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int ch = l + 64 * k;
            v[k] = ch < c ? tr::view_load(x, t, ch) : 0.f;
            s += v[k];
        }
        const float mean = wave_sum(s) / (float)c;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int ch = l + 64 * k;
            const float d = ch < c ? v[k] - mean : 0.f;
            q += d * d;
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)c + eps);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int ch = l + 64 * k;
            if (ch < y.c) tr::view_store(y, t, ch, ch < c ? (v[k] - mean) * rstd * gm[k] + bt[k] : 0.f);
        }
        if (l == 0 && mean_out) { mean_out[t] = mean; rstd_out[t] = rstd; }
    }
}

// The forward on 16-byte accesses (c % 4 == 0, 4-channel aligned views): 16 lanes own a token, a wave takes 2 x 4 tokens per iteration (the scalar
// form above moves 4 bytes per lane and instruction: 52 us for MoSR's 262k x 64 tokens against ~20 us of bytes).  Same two-pass statistics.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(tr_view x, tr_view y, int64_t ntok, int c, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps, float* __restrict__ mean_out,
                                                                 float* __restrict__ rstd_out) {
    const int l = threadIdx.x & 63, sub = l & 15, row = l >> 4;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    float4 gm[NV], bt[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int ch = (sub + 16 * k) * 4;
        gm[k] = ch < c ? make_float4(gamma[ch], gamma[ch + 1], gamma[ch + 2], gamma[ch + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        bt[k] = ch < c ? make_float4(beta[ch], beta[ch + 1], beta[ch + 2], beta[ch + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t t0 = wave * 8; t0 < ntok; t0 += nwaves * 8) {
        float4 v[2][NV];
        float s[2] = {0.f, 0.f}, q[2] = {0.f, 0.f};
        bool live[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t t = t0 + u * 4 + row;
            live[u] = t < ntok;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int ch = (sub + 16 * k) * 4;
                v[u][k] = (live[u] && ch < c) ? tr::view_load4(x, t, ch, true) : make_float4(0.f, 0.f, 0.f, 0.f);
                s[u] += (v[u][k].x + v[u][k].y) + (v[u][k].z + v[u][k].w);
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s[0] += __shfl_xor(s[0], o, 64); s[1] += __shfl_xor(s[1], o, 64); }
        float mean[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            mean[u] = s[u] / (float)c;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int ch = (sub + 16 * k) * 4;
                if (ch < c) {
                    const float dx = v[u][k].x - mean[u], dy = v[u][k].y - mean[u], dz = v[u][k].z - mean[u], dw = v[u][k].w - mean[u];
                    v[u][k] = make_float4(dx, dy, dz, dw);
                    q[u] += (dx * dx + dy * dy) + (dz * dz + dw * dw);
                }
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { q[0] += __shfl_xor(q[0], o, 64); q[1] += __shfl_xor(q[1], o, 64); }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!live[u]) continue;
            const int64_t t = t0 + u * 4 + row;
            const float rstd = rsqrtf(q[u] / (float)c + eps);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int ch = (sub + 16 * k) * 4;
                if (ch < y.c) {
                    const float4 d = v[u][k];
                    const float4 o = ch < c ? make_float4(d.x * rstd * gm[k].x + bt[k].x, d.y * rstd * gm[k].y + bt[k].y, d.z * rstd * gm[k].z + bt[k].z,
                                                          d.w * rstd * gm[k].w + bt[k].w)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
                    tr::view_store4(y, t, ch, o, true);
                }
            }
            if (sub == 0 && mean_out) { mean_out[t] = mean[u]; rstd_out[t] = rstd; }
        }
    }
}

// dx = rstd * (a - mean_c(a) - xhat * mean_c(a * xhat)), a = g * gamma ;  out = (res ? res : 0) + dx
// partial dgamma / dbeta per block -> partials[block][2][c].  NK = ceil(c / 64) register slots per lane (templated: a runtime bound
// would index the arrays dynamically); two tokens per wave iteration keep twice the loads in flight (the kernel is latency bound).
template <int NK>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(tr_view g, tr_view x, int64_t ntok, int c, const float* __restrict__ gamma,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd, tr_view res,
                                                             tr_view dx, float* __restrict__ partials, tr_view dx2, const float* __restrict__ sscale,
                                                             int64_t pps) {
    __shared__ float sh[4][2][NK * 64];
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    float dg[NK], db[NK], gm[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        dg[k] = db[k] = 0.f;
        gm[k] = (l + 64 * k) < c ? gamma[l + 64 * k] : 0.f;
    }
    for (int64_t t0 = wave * 2; t0 < ntok; t0 += nwaves * 2) {
        float a[2][NK], xh[2][NK], rs[2];
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
        bool live[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t t = t0 + u;
            live[u] = t < ntok;
            const float mu = live[u] ? mean[t] : 0.f;
            rs[u] = live[u] ? rstd[t] : 0.f;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int ch = l + 64 * k;
                float gv = 0.f, xv = 0.f;
                if (live[u] && ch < c) { gv = tr::view_load(g, t, ch); xv = tr::view_load(x, t, ch); }
                xh[u][k] = (live[u] && ch < c) ? (xv - mu) * rs[u] : 0.f;
                a[u][k] = gv * gm[k];
                dg[k] += gv * xh[u][k];
                db[k] += gv;
                s1[u] += a[u][k];
                s2[u] += a[u][k] * xh[u][k];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s1[0] += __shfl_xor(s1[0], o, 64); s2[0] += __shfl_xor(s2[0], o, 64);
            s1[1] += __shfl_xor(s1[1], o, 64); s2[1] += __shfl_xor(s2[1], o, 64);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!live[u]) continue;
            const int64_t t = t0 + u;
            const float m1 = s1[u] / (float)c, m2 = s2[u] / (float)c;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int ch = l + 64 * k;
                if (ch < c) {
                    float o = rs[u] * (a[u][k] - m1 - xh[u][k] * m2);
                    if (res.ptr) o += tr::view_load(res, t, ch);
                    tr::view_store(dx, t, ch, o);
                    if (dx2.ptr) tr::view_store(dx2, t, ch, sscale ? o * sscale[t / pps] : o);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) { sh[wv][0][l + 64 * k] = dg[k]; sh[wv][1][l + 64 * k] = db[k]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * c; i += blockDim.x) {
        const int which = i / c, ch = i - which * c;
        partials[(int64_t)blockIdx.x * 2 * c + i] = (sh[0][which][ch] + sh[1][which][ch]) + (sh[2][which][ch] + sh[3][which][ch]);
    }
}
// The same arithmetic on 16-byte accesses (c % 4 == 0, every view 4-channel aligned): 16 lanes own one token (lane sub: channels (sub + 16 k) * 4 .. + 3),
// a wave takes 2 x 4 tokens per iteration with the gradient, the activation and the residual of all eight requested up front (the scalar form above
// waited on 4-byte loads three times per token pair: 38 us for 32,768 x 180 tokens against ~19 us of bytes).  Row sums fold over xor 8 / 4 / 2 / 1,
// the dgamma / dbeta terms of the four token rows over xor 16 / 32 before the block's partial row is written.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(tr_view g, tr_view x, int64_t ntok, int c, const float* __restrict__ gamma,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd, tr_view res,
                                                                 tr_view dx, float* __restrict__ partials, tr_view dx2, const float* __restrict__ sscale,
                                                                 int64_t pps) {
    __shared__ float sh[4][2][NV * 64];
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, sub = l & 15, row = l >> 4;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    float4 dg[NV], db[NV], gm[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int ch = (sub + 16 * k) * 4;
        dg[k] = db[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        gm[k] = ch < c ? make_float4(gamma[ch], gamma[ch + 1], gamma[ch + 2], gamma[ch + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bool has_res = res.ptr != nullptr;
    for (int64_t t0 = wave * 8; t0 < ntok; t0 += nwaves * 8) {
        float4 a[2][NV], xh[2][NV], rv[2][NV];
        float rs[2], mu[2], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
        bool live[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t t = t0 + u * 4 + row;
            live[u] = t < ntok;
            mu[u] = live[u] ? mean[t] : 0.f;
            rs[u] = live[u] ? rstd[t] : 0.f;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int ch = (sub + 16 * k) * 4;
                const bool on = live[u] && ch < c;
                a[u][k] = on ? tr::view_load4(g, t, ch, true) : make_float4(0.f, 0.f, 0.f, 0.f);
                xh[u][k] = on ? tr::view_load4(x, t, ch, true) : make_float4(mu[u], mu[u], mu[u], mu[u]);
                rv[u][k] = (on && has_res) ? tr::view_load4(res, t, ch, true) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const float4 gv = a[u][k];
                float4 h = xh[u][k];
                h.x = (h.x - mu[u]) * rs[u]; h.y = (h.y - mu[u]) * rs[u]; h.z = (h.z - mu[u]) * rs[u]; h.w = (h.w - mu[u]) * rs[u];
                xh[u][k] = h;
                dg[k].x += gv.x * h.x; dg[k].y += gv.y * h.y; dg[k].z += gv.z * h.z; dg[k].w += gv.w * h.w;
                db[k].x += gv.x; db[k].y += gv.y; db[k].z += gv.z; db[k].w += gv.w;
                float4 av;
                av.x = gv.x * gm[k].x; av.y = gv.y * gm[k].y; av.z = gv.z * gm[k].z; av.w = gv.w * gm[k].w;
                a[u][k] = av;
                s1[u] += (av.x + av.y) + (av.z + av.w);
                s2[u] += (av.x * h.x + av.y * h.y) + (av.z * h.z + av.w * h.w);
            }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            s1[0] += __shfl_xor(s1[0], o, 64); s2[0] += __shfl_xor(s2[0], o, 64);
            s1[1] += __shfl_xor(s1[1], o, 64); s2[1] += __shfl_xor(s2[1], o, 64);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!live[u]) continue;
            const int64_t t = t0 + u * 4 + row;
            const float m1 = s1[u] / (float)c, m2 = s2[u] / (float)c;
            const float ss = (dx2.ptr && sscale) ? sscale[t / pps] : 1.f;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int ch = (sub + 16 * k) * 4;
                if (ch < c) {
                    const float4 av = a[u][k], h = xh[u][k], r = rv[u][k];
                    float4 o;
                    o.x = rs[u] * (av.x - m1 - h.x * m2) + r.x; o.y = rs[u] * (av.y - m1 - h.y * m2) + r.y;
                    o.z = rs[u] * (av.z - m1 - h.z * m2) + r.z; o.w = rs[u] * (av.w - m1 - h.w * m2) + r.w;
                    tr::view_store4(dx, t, ch, o, true);
                    if (dx2.ptr) tr::view_store4(dx2, t, ch, make_float4(o.x * ss, o.y * ss, o.z * ss, o.w * ss), true);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float v[8] = {dg[k].x, dg[k].y, dg[k].z, dg[k].w, db[k].x, db[k].y, db[k].z, db[k].w};
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] += __shfl_xor(v[i], 16, 64); v[i] += __shfl_xor(v[i], 32, 64); }
        if (row == 0) {
            const int ch = (sub + 16 * k) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) { sh[wv][0][ch + i] = v[i]; sh[wv][1][ch + i] = v[4 + i]; }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * c; i += blockDim.x) {
        const int which = i / c, ch = i - which * c;
        partials[(int64_t)blockIdx.x * 2 * c + i] = (sh[0][which][ch] + sh[1][which][ch]) + (sh[2][which][ch] + sh[3][which][ch]);
    }
}
// column sums of partials[nblocks][2c]: a block owns 32 columns, 32 row groups stride through the blocks with four independent load chains each
// (8 row groups x 2 chains walked 1,024 partial rows in 64 dependent steps: 21 us, as long as the sweep itself), LDS tree at the end in a fixed order
__global__ __launch_bounds__(1024) void layernorm_bwd_finish_kernel(const float* __restrict__ partials, int nblocks, int c, float* __restrict__ dgamma,
                                                                     float* __restrict__ dbeta, int accumulate) {
    __shared__ float sh[32][33];
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < 2 * c) {
        const float* p = partials + i;
        const int64_t ld = 2 * (int64_t)c;
        int b = rg;
        for (; b + 96 < nblocks; b += 128) { s0 += p[b * ld]; s1 += p[(b + 32) * ld]; s2 += p[(b + 64) * ld]; s3 += p[(b + 96) * ld]; }
        for (; b < nblocks; b += 32) s0 += p[b * ld];
    }
    sh[rg][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && i < 2 * c) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 32; r += 4) s += (sh[r][col] + sh[r + 1][col]) + (sh[r + 2][col] + sh[r + 3][col]);
        float* o = i < c ? dgamma + i : dbeta + (i - c);
        *o = accumulate ? *o + s : s;
    }
}

// ---- window attention ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ int region(int s, int n, int shift) { return s < n - WS ? 0 : (s < n - shift ? 1 : 2); }

// score row of query i against the 64 keys in LDS, + relative-position bias + shift mask; returns row max
template <bool DUMMY>
__device__ __forceinline__ void score_row(const AttnArgs& a, const float* q, const float* kl, int i, int head, int wy, int wx, float (&s)[WT]) {
    const int iy = i >> 3, ix = i & 7;
    const int ri = a.shift ? region(wy * WS + iy, a.h, a.shift) * 3 + region(wx * WS + ix, a.w, a.shift) : 0;
#pragma unroll
    for (int j = 0; j < WT; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < HDMAX; ++d) acc += q[d] * kl[j * HDMAX + d];   // head_dim is zero-padded to 32 in registers and LDS
        const int jy = j >> 3, jx = j & 7;
        acc += a.table[((iy - jy + WS - 1) * (2 * WS - 1) + (ix - jx + WS - 1)) * a.nh + head];
        if (a.shift) {
            const int rj = region(wy * WS + jy, a.h, a.shift) * 3 + region(wx * WS + jx, a.w, a.shift);
            if (rj != ri) acc += -100.f;
        }
        s[j] = acc;
    }
}

__device__ __forceinline__ void softmax_row(float (&s)[WT]) {
    float m = s[0];
#pragma unroll
    for (int j = 1; j < WT; ++j) m = fmaxf(m, s[j]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < WT; ++j) { s[j] = expf(s[j] - m); sum += s[j]; }
    const float inv = 1.f / sum;
#pragma unroll
    for (int j = 0; j < WT; ++j) s[j] *= inv;
}

__device__ __forceinline__ int64_t token_of(const AttnArgs& a, int bb, int wy, int wx, int i) {
    // shifted-window coordinates -> original image coordinates (torch.roll by -shift before partition, +shift after reverse)
    const int y = (wy * WS + (i >> 3) + a.shift) % a.h, x = (wx * WS + (i & 7) + a.shift) % a.w;
    return ((int64_t)bb * a.h + y) * a.w + x;
}

__global__ __launch_bounds__(64) void window_attn_fwd_kernel(AttnArgs a) {
    __shared__ float kl[WT * HDMAX], vl[WT * HDMAX];
    const int i = threadIdx.x;
    int blk = blockIdx.x;
    const int head = blk % a.nh; blk /= a.nh;
    const int nwx = a.w / WS, nwy = a.h / WS;
    const int wx = blk % nwx; blk /= nwx;
    const int wy = blk % nwy;
    const int bb = blk / nwy;
    const int64_t t = token_of(a, bb, wy, wx, i);
    float q[HDMAX];
#pragma unroll
    for (int d = 0; d < HDMAX; ++d) {
        const bool ok = d < a.hd;
        q[d] = ok ? tr::view_load(a.qkv, t, head * a.hd + d) * a.scale : 0.f;
        kl[i * HDMAX + d] = ok ? tr::view_load(a.qkv, t, a.c + head * a.hd + d) : 0.f;
        vl[i * HDMAX + d] = ok ? tr::view_load(a.qkv, t, 2 * a.c + head * a.hd + d) : 0.f;
    }
    __syncthreads();
    float s[WT];
    score_row<true>(a, q, kl, i, head, wy, wx, s);
    softmax_row(s);
#pragma unroll
    for (int d = 0; d < HDMAX; ++d) {
        float o = 0.f;
#pragma unroll
        for (int j = 0; j < WT; ++j) o += s[j] * vl[j * HDMAX + d];
        if (d < a.hd) tr::view_store(a.out, t, head * a.hd + d, o);
    }
}

__global__ __launch_bounds__(64) void window_attn_bwd_kernel(AttnArgs a) {
    // q (scaled), k, v and dO of the window live in LDS with rows padded to 33 floats: a lane reading ITS row is conflict free,
    // all lanes reading the SAME row is a broadcast.  Only the 64-wide score / probability rows are register arrays.
    constexpr int HP = HDMAX + 1;
    extern __shared__ float sm[];
    float* kl = sm;
    float* vl = kl + WT * HP;
    float* ql = vl + WT * HP;
    float* gl = ql + WT * HP;
    float* dsl = gl + WT * HP;            // dS  [64][65]
    float* pl = dsl + WT * (WT + 1);      // P   [64][65]
    const int i = threadIdx.x;
    int blk = blockIdx.x;
    const int head = blk % a.nh; blk /= a.nh;
    const int nwx = a.w / WS, nwy = a.h / WS;
    const int wx = blk % nwx; blk /= nwx;
    const int wy = blk % nwy;
    const int bb = blk / nwy;
    const int64_t t = token_of(a, bb, wy, wx, i);
    const int hd = a.hd;
    for (int d = 0; d < hd; ++d) {
        ql[i * HP + d] = tr::view_load(a.qkv, t, head * hd + d) * a.scale;
        gl[i * HP + d] = tr::view_load(a.dout, t, head * hd + d);
        kl[i * HP + d] = tr::view_load(a.qkv, t, a.c + head * hd + d);
        vl[i * HP + d] = tr::view_load(a.qkv, t, 2 * a.c + head * hd + d);
    }
    __syncthreads();
    const int iy = i >> 3, ix = i & 7;
    const int ri = a.shift ? region(wy * WS + iy, a.h, a.shift) * 3 + region(wx * WS + ix, a.w, a.shift) : 0;
    float s[WT];
#pragma unroll
    for (int j = 0; j < WT; ++j) {
        float acc = 0.f;
        for (int d = 0; d < hd; ++d) acc += ql[i * HP + d] * kl[j * HP + d];
        const int jy = j >> 3, jx = j & 7;
        acc += a.table[((iy - jy + WS - 1) * (2 * WS - 1) + (ix - jx + WS - 1)) * a.nh + head];
        if (a.shift) {
            const int rj = region(wy * WS + jy, a.h, a.shift) * 3 + region(wx * WS + jx, a.w, a.shift);
            if (rj != ri) acc += -100.f;
        }
        s[j] = acc;
    }
    softmax_row(s);                                      // s = P row
    float delta = 0.f;
    float dp[WT];
#pragma unroll
    for (int j = 0; j < WT; ++j) {
        float acc = 0.f;
        for (int d = 0; d < hd; ++d) acc += gl[i * HP + d] * vl[j * HP + d];
        dp[j] = acc;
        delta += acc * s[j];
    }
    float* dbw = a.dbias_ws ? a.dbias_ws + ((((int64_t)bb * nwy + wy) * nwx + wx) * a.nh + head) * (WT * WT) + (int64_t)i * WT : nullptr;
#pragma unroll
    for (int j = 0; j < WT; ++j) {
        const float ds = s[j] * (dp[j] - delta);
        dsl[i * (WT + 1) + j] = ds;
        pl[i * (WT + 1) + j] = s[j];
        if (dbw) dbw[j] = ds;
        dp[j] = ds;
    }
    // dQ_i = scale * sum_j dS[i][j] K_j
    for (int d = 0; d < hd; ++d) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < WT; ++j) acc += dp[j] * kl[j * HP + d];
        tr::view_store(a.dqkv, t, head * hd + d, acc * a.scale);
    }
    __syncthreads();
    // lane now acts as key/value token j = i:  dK_j = sum_i dS[i][j] * qs_i ;  dV_j = sum_i P[i][j] * dO_i
    for (int d = 0; d < hd; ++d) {
        float dk = 0.f, dv = 0.f;
        for (int r = 0; r < WT; ++r) {
            dk += dsl[r * (WT + 1) + i] * ql[r * HP + d];
            dv += pl[r * (WT + 1) + i] * gl[r * HP + d];
        }
        tr::view_store(a.dqkv, t, a.c + head * hd + d, dk);
        tr::view_store(a.dqkv, t, 2 * a.c + head * hd + d, dv);
    }
}

// dbias_full[h][i][j] = sum over (batch, window) ; then table_grad[idx][h] += sum over (i,j) with rel-index idx
__global__ __launch_bounds__(256) void dbias_sum_kernel(const float* __restrict__ ws, int nbw, int nh, float* __restrict__ full) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nh * WT * WT) return;
    float s0 = 0.f, s1 = 0.f;
    int k = 0;
    for (; k + 2 <= nbw; k += 2) { s0 += ws[(int64_t)k * nh * WT * WT + i]; s1 += ws[(int64_t)(k + 1) * nh * WT * WT + i]; }
    if (k < nbw) s0 += ws[(int64_t)k * nh * WT * WT + i];
    full[i] = s0 + s1;
}
__global__ __launch_bounds__(256) void dbias_table_kernel(const float* __restrict__ full, int nh, float* __restrict__ dtable, int accumulate) {
    const int tdx = blockIdx.x * blockDim.x + threadIdx.x;
    const int ntab = (2 * WS - 1) * (2 * WS - 1);
    if (tdx >= ntab * nh) return;
    const int idx = tdx / nh, head = tdx - idx * nh;
    const int dy = idx / (2 * WS - 1) - (WS - 1), dx = idx % (2 * WS - 1) - (WS - 1);   // iy - jy, ix - jx
    float s = 0.f;
    for (int iy = 0; iy < WS; ++iy) {
        const int jy = iy - dy;
        if (jy < 0 || jy >= WS) continue;
        for (int ix = 0; ix < WS; ++ix) {
            const int jx = ix - dx;
            if (jx < 0 || jx >= WS) continue;
            s += full[(head * WT + (iy * WS + ix)) * WT + (jy * WS + jx)];
        }
    }
    dtable[tdx] = accumulate ? dtable[tdx] + s : s;
}

// ---- PixelShuffle(r): out[n, y*r+dy, x*r+dx, c] = in[n, y, x, c*r*r + dy*r + dx] ; inverse = pixel_unshuffle ------------------
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(tr_view in, tr_view out, int n, int h, int w, int c, int r, int inverse) {
    const int64_t total = (int64_t)n * h * w * c * r * r;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int cc = (int)(t % (c * r * r)); t /= (c * r * r);
        const int x = (int)(t % w); t /= w;
        const int y = (int)(t % h);
        const int nb = (int)(t / h);
        const int ch = cc / (r * r), dy = (cc / r) % r, dx = cc % r;
        const int64_t plow = ((int64_t)nb * h + y) * w + x;
        const int64_t phigh = ((int64_t)nb * h * r + (y * r + dy)) * (w * r) + (x * r + dx);
        if (inverse) tr::view_store(out, plow, cc, tr::view_load(in, phigh, ch));
        else tr::view_store(out, phigh, ch, tr::view_load(in, plow, cc));
    }
}

inline unsigned grid_for(int64_t work, int cap = 256 * 16) {
    int64_t b = (work + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int tr_layernorm(void* stream, tr_view x, tr_view y, int64_t ntok, int c, const float* gamma, const float* beta, float eps,
                            float* mean_out, float* rstd_out) {
    TR_REQUIRE(x.ptr && y.ptr && gamma && beta && ntok > 0 && c > 0 && c <= LN_MAXK * 64 && x.c >= c && y.c >= c && y.c <= LN_MAXK * 64, TR_EINVAL,
               "tr_layernorm: bad arguments (c <= %d)", LN_MAXK * 64);
    TR_REQUIRE((mean_out == nullptr) == (rstd_out == nullptr), TR_EINVAL, "tr_layernorm: mean/rstd outputs come together");
    const int nk = (y.c + 63) / 64;         // the store loop covers the padded width of y
    hipStream_t st = (hipStream_t)stream;
    if (c % 4 == 0 && y.c % 4 == 0 && nk <= 4 && tr::view_vec4_ok(x) && tr::view_vec4_ok(y)) {
        const dim3 gridv(grid_for(ntok * 8, 2048));          // eight tokens per wave iteration
#define TR_LN_FWDV(NV) hipLaunchKernelGGL(layernorm_fwd_vec_kernel<NV>, gridv, dim3(256), 0, st, x, y, ntok, c, gamma, beta, eps, mean_out, rstd_out)
        if (nk <= 1) TR_LN_FWDV(1); else if (nk <= 2) TR_LN_FWDV(2); else if (nk <= 3) TR_LN_FWDV(3); else TR_LN_FWDV(4);
#undef TR_LN_FWDV
        return tr::launch_status("tr_layernorm");
    }
    const dim3 grid(grid_for(ntok * 64, 2048));
#define TR_LN_FWD(NK) hipLaunchKernelGGL(layernorm_fwd_kernel<NK>, grid, dim3(256), 0, st, x, y, ntok, c, gamma, beta, eps, mean_out, rstd_out)
    if (nk <= 1) TR_LN_FWD(1); else if (nk <= 2) TR_LN_FWD(2); else if (nk <= 3) TR_LN_FWD(3); else if (nk <= 4) TR_LN_FWD(4);
    else if (nk <= 6) TR_LN_FWD(6); else if (nk <= 8) TR_LN_FWD(8); else TR_LN_FWD(12);
#undef TR_LN_FWD
    return tr::launch_status("tr_layernorm");
}

extern "C" int tr_layernorm_bwd(void* stream, tr_view g, tr_view x, int64_t ntok, int c, const float* gamma, const float* mean, const float* rstd,
                                tr_view res, tr_view dx, float* dgamma, float* dbeta, int accumulate, float* partials, size_t partials_bytes,
                                tr_view dx2, const float* sample_scale, int64_t pix_per_sample) {
    TR_REQUIRE(!dx2.ptr || dx2.c >= c, TR_EINVAL, "tr_layernorm_bwd: dx2 view too narrow");
    TR_REQUIRE(!sample_scale || (dx2.ptr && pix_per_sample > 0), TR_EINVAL, "tr_layernorm_bwd: sample_scale needs dx2 and pix_per_sample");
    if (!sample_scale) pix_per_sample = 1;
    TR_REQUIRE(g.ptr && x.ptr && dx.ptr && gamma && mean && rstd && dgamma && dbeta && partials && ntok > 0 && c > 0 && c <= LN_MAXK * 64, TR_EINVAL,
               "tr_layernorm_bwd: bad arguments");
    // a wave takes two tokens per iteration; at least 8 iterations per wave so that the partial rows (one per block, summed by the
    // finish kernel) stay few when the token count is small
    int nb = (int)grid_for(ntok * 4, LN_BWD_BLOCKS);
    TR_REQUIRE(partials_bytes >= (size_t)nb * 2 * c * sizeof(float), TR_EINVAL, "tr_layernorm_bwd: workspace too small (need %zu bytes)",
               (size_t)nb * 2 * c * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    const int nk = (c + 63) / 64;
    if (c % 4 == 0 && nk <= 4 && tr::view_vec4_ok(g) && tr::view_vec4_ok(x) && tr::view_vec4_ok(res) && tr::view_vec4_ok(dx) && tr::view_vec4_ok(dx2)) {
        nb = (int)grid_for(ntok * 8, LN_BWD_BLOCKS);       // eight tokens per wave iteration
#define TR_LN_BWDV(NV) hipLaunchKernelGGL(layernorm_bwd_vec_kernel<NV>, dim3(nb), dim3(256), 0, st, g, x, ntok, c, gamma, mean, rstd, res, dx, partials, dx2, sample_scale, pix_per_sample)
        if (nk <= 1) TR_LN_BWDV(1); else if (nk <= 2) TR_LN_BWDV(2); else if (nk <= 3) TR_LN_BWDV(3); else TR_LN_BWDV(4);
#undef TR_LN_BWDV
        hipLaunchKernelGGL(layernorm_bwd_finish_kernel, dim3((2 * c + 31) / 32), dim3(1024), 0, st, partials, nb, c, dgamma, dbeta, accumulate);
        return tr::launch_status("tr_layernorm_bwd");
    }
#define TR_LN_BWD(NK) hipLaunchKernelGGL(layernorm_bwd_kernel<NK>, dim3(nb), dim3(256), 0, st, g, x, ntok, c, gamma, mean, rstd, res, dx, partials, dx2, sample_scale, pix_per_sample)
    if (nk <= 1) TR_LN_BWD(1); else if (nk <= 2) TR_LN_BWD(2); else if (nk <= 3) TR_LN_BWD(3); else if (nk <= 4) TR_LN_BWD(4);
    else if (nk <= 6) TR_LN_BWD(6); else if (nk <= 8) TR_LN_BWD(8); else TR_LN_BWD(12);
#undef TR_LN_BWD
    hipLaunchKernelGGL(layernorm_bwd_finish_kernel, dim3((2 * c + 31) / 32), dim3(1024), 0, st, partials, nb, c, dgamma, dbeta, accumulate);
    return tr::launch_status("tr_layernorm_bwd");
}
extern "C" size_t tr_layernorm_bwd_workspace_bytes(int c) { return (size_t)LN_BWD_BLOCKS * 2 * c * sizeof(float); }

static int attn_check(const tr_window_attn_desc* d) {
    TR_REQUIRE(d && d->qkv.ptr && d->table && d->b > 0 && d->h > 0 && d->w > 0 && d->nh > 0, TR_EINVAL, "tr_window_attention: bad arguments");
    TR_REQUIRE(d->window == WS, TR_EUNSUPPORTED, "tr_window_attention: window size %d (only 8 is built)", d->window);
    TR_REQUIRE(d->h % WS == 0 && d->w % WS == 0, TR_EINVAL, "tr_window_attention: H, W must be multiples of the window");
    TR_REQUIRE(d->c % d->nh == 0 && d->c / d->nh <= HDMAX, TR_EUNSUPPORTED, "tr_window_attention: head_dim must be <= %d", HDMAX);
    TR_REQUIRE(d->shift >= 0 && d->shift < WS && d->qkv.c >= 3 * d->c, TR_EINVAL, "tr_window_attention: bad shift / qkv width");
    return 0;
}
static AttnArgs attn_args(const tr_window_attn_desc* d) {
    AttnArgs a;
    a.qkv = d->qkv; a.out = d->out; a.dout = d->dout; a.dqkv = d->dqkv; a.table = d->table; a.dbias_ws = d->dbias_ws;
    a.b = d->b; a.h = d->h; a.w = d->w; a.nh = d->nh; a.c = d->c; a.hd = d->c / d->nh; a.shift = d->shift; a.scale = d->scale;
    return a;
}
extern "C" int tr_window_attention(void* stream, const tr_window_attn_desc* d) {
    int rc = attn_check(d);
    if (rc) return rc;
    TR_REQUIRE(d->out.ptr && d->out.c >= d->c, TR_EINVAL, "tr_window_attention: output view too narrow");
    if (tr::attn_mfma_ok(attn_args(d), false)) return tr::launch_window_attn_fwd_mfma((hipStream_t)stream, attn_args(d));
    const int nblk = d->b * (d->h / WS) * (d->w / WS) * d->nh;
    hipLaunchKernelGGL(window_attn_fwd_kernel, dim3(nblk), dim3(64), 0, (hipStream_t)stream, attn_args(d));
    return tr::launch_status("tr_window_attention");
}
extern "C" size_t tr_window_attention_bwd_workspace_bytes(int b, int h, int w, int nh) {
    const size_t valu = ((size_t)b * (h / WS) * (w / WS) * nh + (size_t)nh) * WT * WT;            // per-window dS (fp32 path)
    const size_t mfma = tr::window_attn_bwd_mfma_workspace_floats(b * (h / WS) * (w / WS) * nh, nh);  // per-wave sums (bf16 path)
    return (valu > mfma ? valu : mfma) * sizeof(float);
}
extern "C" int tr_window_attention_bwd(void* stream, const tr_window_attn_desc* d) {
    int rc = attn_check(d);
    if (rc) return rc;
    TR_REQUIRE(d->dout.ptr && d->dqkv.ptr && d->dqkv.c >= 3 * d->c && d->dout.c >= d->c, TR_EINVAL, "tr_window_attention_bwd: gradient views too narrow");
    TR_REQUIRE((d->dtable == nullptr) == (d->dbias_ws == nullptr), TR_EINVAL, "tr_window_attention_bwd: dtable and dbias_ws come together");
    hipStream_t st = (hipStream_t)stream;
    if (tr::attn_mfma_ok(attn_args(d), true)) return tr::launch_window_attn_bwd_mfma(st, attn_args(d), d->dtable, d->accumulate_dtable);
    const int nbw = d->b * (d->h / WS) * (d->w / WS);
    const size_t lds = (size_t)(4 * WT * (HDMAX + 1) + 2 * WT * (WT + 1)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)window_attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(window_attn_bwd_kernel, dim3(nbw * d->nh), dim3(64), lds, st, attn_args(d));
    if (d->dtable) {
        float* full = d->dbias_ws + (size_t)nbw * d->nh * WT * WT;
        hipLaunchKernelGGL(dbias_sum_kernel, dim3((d->nh * WT * WT + 255) / 256), dim3(256), 0, st, d->dbias_ws, nbw, d->nh, full);
        hipLaunchKernelGGL(dbias_table_kernel, dim3(((2 * WS - 1) * (2 * WS - 1) * d->nh + 255) / 256), dim3(256), 0, st, full, d->nh, d->dtable,
                           d->accumulate_dtable);
    }
    return tr::launch_status("tr_window_attention_bwd");
}

extern "C" int tr_pixel_shuffle(void* stream, tr_view in, tr_view out, int n, int h, int w, int c, int r, int inverse) {
    TR_REQUIRE(in.ptr && out.ptr && n > 0 && h > 0 && w > 0 && c > 0 && r >= 1, TR_EINVAL, "tr_pixel_shuffle: bad arguments");
    TR_REQUIRE(inverse ? (in.c >= c && out.c >= c * r * r) : (in.c >= c * r * r && out.c >= c), TR_EINVAL, "tr_pixel_shuffle: views too narrow");
    hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(grid_for((int64_t)n * h * w * c * r * r)), dim3(256), 0, (hipStream_t)stream, in, out, n, h, w, c, r,
                       inverse);
    return tr::launch_status("tr_pixel_shuffle");
}
