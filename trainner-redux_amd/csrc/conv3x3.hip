// 3x3 / stride 1 / pad 1 convolution on NHWC channel-slice views as an implicit GEMM on MFMA.
//
// Contract: include/trainner_hip.h (tr_conv3x3).  Reference arithmetic replaced:
//   ResidualDenseBlock / RRDB      traiNNer/archs/scunet_aaf6aa_arch.py:310-360
//   nearest x2 upconv tail         traiNNer/archs/scunet_aaf6aa_arch.py:363-384
//   VGG19 conv + ReLU stack        traiNNer/losses/perceptual_fp16_loss.py:246-329
//
// Mapping to gfx950 (DESIGN.md "conv3x3 kernel"):
//   * one workgroup = 4 waves = a (4*R) x 32 pixel tile x one group of 32/64 output channels;
//   * the K loop walks the input channels in 64-byte chunks (32 bf16 / 16 f32 channels); per
//     chunk the (4R+2) x 34 pixel halo and the 9 x CG x 64 B weight slab are staged in LDS by LDS-DMA
//     (buffer_load ... lds, two stages: chunk c+1 lands while chunk c feeds the MFMAs, one barrier per chunk)
//     (16-byte slots XOR-swizzled by (row>>2)&3 so every ds_read_b128 of an MFMA operand is
//     bank-conflict free) and all nine taps are read out of the same halo image by address
//     offset -- im2col is never materialised;
//   * MFMA orientation D[cout][pixel] = W[cout][k] * X[k][pixel]: the accumulator holds, per
//     lane, one pixel and 16 output channels in groups of 4 consecutive channels, so the fused
//     epilogue (bias, LeakyReLU/ReLU, two scaled residuals, activation-derivative mask, dual
//     dtype stores) works on 8/16-byte NHWC vectors.
#include "common.h"
#include "conv_args.h"
#include <stdlib.h>
#include <type_traits>
#include <string.h>

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TW = 32;        // pixels along x per MFMA tile
constexpr int HALO_W = TW + 2;   // 3x3 kernels (the LDS-DMA variant is 3x3 only)
constexpr int NTHREADS = 256;



// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic, not for its global memory traffic.  __syncthreads() adds a
// release fence, i.e. s_waitcnt vmcnt(0), and on gfx9 vmcnt counts stores too: at a tile boundary of the persistent loop every wave then
// sat out the full HBM acknowledgement of the epilogue stores it had just issued (measured: 40 % of the 64 -> 64 kernel at 1024 x 1024).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// s_waitcnt vmcnt(0) as an instruction the compiler's wait-count bookkeeping sees (expcnt / lgkmcnt fields left at "no wait")
#define TR_WAIT_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)

// ---- 9 taps x one K-chunk of MFMAs out of the staged LDS image (hl: halo, wl: weight slab) --------------------------
// Software pipelined over the 2 * KS * KS (tap, 16-byte slot pair) steps: the operand fragments of step s + 1 are read from LDS
// before the MFMAs of step s issue, so a wave overlaps its own LDS latency instead of relying on the other resident wave
// (the straight version waited on a fresh ds_read in front of every pair of MFMAs).
template <int CT, int NT, int R, int KS = 3>
__device__ __forceinline__ void conv_chunk_mfma(const char* hl, const char* wl, f32x16 (&acc)[R][NT], int wv, int lr, int lh) {
    constexpr int CGW = NT * 32;
    constexpr int HALO_W = TW + KS - 1;
    if constexpr (CT == TR_BF16X3) {
        // split-bf16 "fp32": a row holds 16 channels as [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15] (16-byte slots, same swizzle), so one tap
        // is ONE 16-deep MFMA step per operand pairing: acc += w_hi*x_hi + w_hi*x_lo + w_lo*x_hi   (lo*lo ~ 2^-18 is dropped)
#pragma unroll
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
            bf16x8 wh[NT], wlo[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int row = tap * CGW + t * 32 + lr;
                const int sw = (row >> 2) & 3;
                wh[t] = *(const bf16x8*)(wl + row * 64 + ((lh ^ sw) << 4));
                wlo[t] = *(const bf16x8*)(wl + row * 64 + (((2 + lh) ^ sw) << 4));
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int pp = (wv * R + r + ky) * HALO_W + lr + kx;
                const int sw = (pp >> 2) & 3;
                const bf16x8 xh = *(const bf16x8*)(hl + pp * 64 + ((lh ^ sw) << 4));
                const bf16x8 xl = *(const bf16x8*)(hl + pp * 64 + (((2 + lh) ^ sw) << 4));
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[t], xh, acc[r][t], 0, 0, 0);
                    acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[t], xl, acc[r][t], 0, 0, 0);
                    acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[t], xh, acc[r][t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    constexpr int NSTEP = KS * KS * 2;
    typedef typename std::conditional<CT == TR_F32, f32x4, bf16x8>::type frag_t;
    frag_t wa[2][NT], xb[2][R];
#define TR_CONV_FRAGS(BUF, STEP)                                                                                        \
    {                                                                                                                   \
        const int tap_ = (STEP) >> 1, ky_ = tap_ / KS, kx_ = tap_ % KS, slot_ = ((STEP) & 1) * 2 + lh;                   \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                                 \
            const int row = tap_ * CGW + t * 32 + lr;                                                                    \
            wa[BUF][t] = *(const frag_t*)(wl + row * 64 + ((slot_ ^ ((row >> 2) & 3)) << 4));                            \
        }                                                                                                               \
        _Pragma("unroll") for (int r = 0; r < R; ++r) {                                                                  \
            const int pp = (wv * R + r + ky_) * HALO_W + lr + kx_;                                                       \
            xb[BUF][r] = *(const frag_t*)(hl + pp * 64 + ((slot_ ^ ((pp >> 2) & 3)) << 4));                              \
        }                                                                                                               \
    }
    // the 8-row / 32-channel variant lives on a 168-VGPR budget (three workgroups per CU): no second fragment set there
    constexpr bool PIPE = !(NT == 1 && R == 2 && KS == 3);
    if constexpr (PIPE) TR_CONV_FRAGS(0, 0)
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int cur = PIPE ? (s & 1) : 0;
        if constexpr (PIPE) {
            if (s + 1 < NSTEP) TR_CONV_FRAGS(cur ^ 1, s + 1)
        } else {
            TR_CONV_FRAGS(0, s)
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (CT == TR_BF16) {
                    acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[cur][t], xb[cur][r], acc[r][t], 0, 0, 0);
                } else if constexpr (CT == TR_F16) {
                    acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa[cur][t]), __builtin_bit_cast(f16x8, xb[cur][r]), acc[r][t], 0, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][t][e], xb[cur][r][e], acc[r][t], 0, 0, 0);
                }
            }
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch exactly one step ahead (VGPR budget)
    }
#undef TR_CONV_FRAGS
}

typedef __attribute__((ext_vector_type(2))) float f32x2c;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2c;
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {   // element 0 in the low half
    const f32x2c v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2c));
}

__device__ __forceinline__ float silu_grad(float x) {   // d/dx [x * sigmoid(x)]
    const float sg = 1.f / (1.f + __expf(-x));
    return sg * (1.f + x * (1.f - sg));
}
__device__ __forceinline__ float gelu_grad(float x) {   // d/dx [x * Phi(x)], exact (erf) GELU like nn.GELU()
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
    return cdf + x * 0.3989422804014327f * expf(-0.5f * x * x);
}
// Phi(x) and e^{-x^2/2} for the 16-bit straight-line token-GEMM forms: erfc(z) = poly(t) e^{-z^2}, t = 1 / (1 + 0.3275911 z), z = |x| / sqrt 2
// (Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7 on erf; measured in fp32 against float64: GELU 4.2e-7, GELU' 3.2e-7 absolute over [-12, 12]) --
// one v_rcp and one v_exp instead of ocml's two-branch erff, whose ~60 divergent instructions per element made the fc1 Linear (GELU + saved
// pre-activation) 14 us longer than the same GEMM without it at 32k tokens.  The results are rounded to bf16 / f16 right after.
struct GeluParts { float cdf, e; };
__device__ __forceinline__ GeluParts gelu_parts_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * z);
    const float e = __expf(-z * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float h = 0.5f * poly * e;
    return {x >= 0.f ? 1.f - h : h, e};
}
__device__ __forceinline__ float gelu_fast(float x) { return x * gelu_parts_fast(x).cdf; }
__device__ __forceinline__ float gelu_grad_fast(float x) {
    const GeluParts g = gelu_parts_fast(x);
    return g.cdf + x * 0.3989422804014327f * g.e;
}
__device__ __forceinline__ float act_pre_grad(int mode, float m) {   // mask_mode 1 / 2 / 3: derivative of GELU / SiLU / Mish at the pre-activation m
    return mode == 1 ? gelu_grad(m) : mode == 2 ? silu_grad(m) : tr::mish_grad(m);
}

// ---- fused epilogue: bias, activation, two scaled residuals, derivative mask, dual-dtype NHWC stores ------------------
// The MFMA accumulator holds, per lane, ONE pixel and 16 scattered channel quads, so storing from it directly issues
// 8-byte stores to 64 different pixel records per instruction (measured: the stores of a 64->64 conv at 1024^2 cost
// 0.75 of its 2.0 ms).  Each wave therefore transposes one 32-pixel row at a time through a private LDS strip
// ([pixel][CGW] fp32, rows padded by 16 B against bank conflicts) and then walks it with lane = (pixel, 4-channel
// segment): residual / mask loads and the stores become 8-16 B per lane over contiguous pixel records.
template <int NT, int R, bool SIDE>
__device__ __forceinline__ void conv_epilogue_impl(const ConvArgs& a, f32x16 (&acc)[R][NT], char* smem, int nb, int y0, int x0, int cg, int wv,
                                              int l, bool wave_active, int strip_id) {
    constexpr int CGW = NT * 32;
    constexpr int ROWB = CGW * 4 + 16;
    constexpr int SEGS = CGW / 4;            // 16-byte fp32 segments per pixel
    constexpr int ITER = 32 * SEGS / 64;
    const int lr = l & 31, lh = l >> 5;
    char* strip = smem + (strip_id < 0 ? wv : strip_id) * (32 * ROWB);   // wv: row group of the tile; strip_id: wave (when they differ)
    const bool generic = !(a.vec_out && a.vec_out2 && a.vec_res1 && a.vec_res2 && a.vec_mask);
    // a lane keeps the same 4-channel segment across the iterations of a row when 64 % SEGS == 0: its bias is loaded once per tile
    // (it used to be four dependent global loads inside every iteration of a rolled loop)
    constexpr bool FIXED_SEG = (64 % SEGS) == 0;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (FIXED_SEG && a.bias) {
        const int co_ = cg * CGW + (l % SEGS) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (co_ + k < a.cout) bias4[k] = a.bias[co_ + k];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int oy = y0 + wv * R + r;
        if (oy >= a.h) break;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(float4*)(strip + lr * ROWB + (t * 32 + q * 8 + lh * 4) * 4) =
                    make_float4(acc[r][t][q * 4 + 0], acc[r][t][q * 4 + 1], acc[r][t][q * 4 + 2], acc[r][t][q * 4 + 3]);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // a rolled loop on purpose: the body (erf GELU, derivative mask, generic tails) is large, and unrolling it ITER x R times
        // pushes the kernel past the unroll budget -- the r loop then stays rolled and `acc` is demoted to scratch memory
#pragma unroll 2
        for (int i = 0; i < ITER; ++i) {
            const int idx = i * 64 + l;
            const int p = idx / SEGS, seg = idx - p * SEGS;
            const int ox = x0 + p;
            const int co = cg * CGW + seg * 4;
            const float4 raw = *(const float4*)(strip + p * ROWB + seg * 16);
            if (ox >= a.w || co >= a.cout) continue;
            const int64_t pix = ((int64_t)nb * a.h + oy) * a.w + ox;
            float v[4] = {raw.x, raw.y, raw.z, raw.w};
            const bool full = (co + 3) < a.cout;
            if (FIXED_SEG) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] += bias4[k];
            } else if (a.bias) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (co + k < a.cout) v[k] += a.bias[co + k];
            }
            if (a.act == TR_ACT_LRELU) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * a.slope;
            } else if (a.act == TR_ACT_RELU) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
            }
            float pre[4] = {v[0], v[1], v[2], v[3]};   // value before GELU / SiLU: what out2 stores in those modes (needed by the backward)
            if (a.act == TR_ACT_GELU) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = 0.5f * v[k] * (1.f + erff(v[k] * 0.70710678118654752f));
            } else if (a.act == TR_ACT_SILU) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = v[k] / (1.f + __expf(-v[k]));
            } else if (a.act == TR_ACT_MISH) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = tr::mish(v[k]);
            }
            if (a.sample_scale) {
                const float ss = a.sample_scale[nb];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] *= ss;
            }
            if (full && !generic) {
                if (SIDE && a.res1.ptr) {
                    const float4 rr = tr::view_load4(a.res1, pix, co, true);
                    v[0] = v[0] * a.scale1 + a.rs1 * rr.x; v[1] = v[1] * a.scale1 + a.rs1 * rr.y;
                    v[2] = v[2] * a.scale1 + a.rs1 * rr.z; v[3] = v[3] * a.scale1 + a.rs1 * rr.w;
                }
                if (SIDE && a.res2.ptr) {
                    const float4 rr = tr::view_load4(a.res2, pix, co, true);
                    v[0] = v[0] * a.scale2 + a.rs2 * rr.x; v[1] = v[1] * a.scale2 + a.rs2 * rr.y;
                    v[2] = v[2] * a.scale2 + a.rs2 * rr.z; v[3] = v[3] * a.scale2 + a.rs2 * rr.w;
                }
                if (SIDE && a.mask.ptr) {
                    const float4 mm = tr::view_load4(a.mask, pix, co, true);
                    if (a.mask_mode == 1) {
                        v[0] *= gelu_grad(mm.x); v[1] *= gelu_grad(mm.y); v[2] *= gelu_grad(mm.z); v[3] *= gelu_grad(mm.w);
                    } else if (a.mask_mode == 2) {
                        v[0] *= silu_grad(mm.x); v[1] *= silu_grad(mm.y); v[2] *= silu_grad(mm.z); v[3] *= silu_grad(mm.w);
                    } else if (a.mask_mode == 3) {
                        v[0] *= tr::mish_grad(mm.x); v[1] *= tr::mish_grad(mm.y); v[2] *= tr::mish_grad(mm.z); v[3] *= tr::mish_grad(mm.w);
                    } else {
                        v[0] *= mm.x > 0.f ? 1.f : a.mask_slope; v[1] *= mm.y > 0.f ? 1.f : a.mask_slope;
                        v[2] *= mm.z > 0.f ? 1.f : a.mask_slope; v[3] *= mm.w > 0.f ? 1.f : a.mask_slope;
                    }
                }
                tr::view_store4(a.out, pix, co, make_float4(v[0], v[1], v[2], v[3]), true);
                if (a.out2.ptr) {
                    const float sc = a.out2_scale;
                    if (a.act >= TR_ACT_GELU) tr::view_store4(a.out2, pix, co, make_float4(pre[0], pre[1], pre[2], pre[3]), true);
                    else tr::view_store4(a.out2, pix, co, make_float4(v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc), true);
                }
            } else {
                for (int k = 0; k < 4 && co + k < a.cout; ++k) {
                    float x = v[k];
                    if (SIDE && a.res1.ptr) x = x * a.scale1 + a.rs1 * tr::view_load(a.res1, pix, co + k);
                    if (SIDE && a.res2.ptr) x = x * a.scale2 + a.rs2 * tr::view_load(a.res2, pix, co + k);
                    if (SIDE && a.mask.ptr) {
                        const float m = tr::view_load(a.mask, pix, co + k);
                        x *= a.mask_mode ? act_pre_grad(a.mask_mode, m) : (m > 0.f ? 1.f : a.mask_slope);
                    }
                    tr::view_store(a.out, pix, co + k, x);
                    if (a.out2.ptr) tr::view_store(a.out2, pix, co + k, a.act >= TR_ACT_GELU ? pre[k] : x * a.out2_scale);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// Two compiled bodies, picked per launch: without side inputs (residuals, derivative mask) the epilogue issues no global loads, so the compiler
// places no s_waitcnt vmcnt in its loop.  In the single body every iteration opened with vmcnt(0) -- the conservative wait for a possibly
// pending residual load -- and vmcnt counts stores on gfx9: each of the 16 store instructions of a tile waited for the HBM acknowledgement
// of the one before it (40 % of the 64 -> 64 kernel at 1024 x 1024).
template <int NT, int R>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[R][NT], char* smem, int nb, int y0, int x0, int cg, int wv,
                                              int l, bool wave_active, int strip_id = -1) {
    __syncthreads();                          // every wave is done with the staged operands: LDS is free
    if (!wave_active || (a.dbg & 4)) return;
    if (a.res1.ptr || a.res2.ptr || a.mask.ptr) conv_epilogue_impl<NT, R, true>(a, acc, smem, nb, y0, x0, cg, wv, l, wave_active, strip_id);
    else conv_epilogue_impl<NT, R, false>(a, acc, smem, nb, y0, x0, cg, wv, l, wave_active, strip_id);
}

// ---- epilogue of the persistent 3x3 kernel: all rows of the wave through LDS first, side inputs fetched a chunk ahead --------------------
// conv_epilogue above walks a row at a time and, with residual / mask inputs, loads each 4-channel piece right where it is used: one global
// round trip per iteration, 8-16 in a row per tile (measured: the masked dgrad form ran 40-70 % behind the plain conv, the two-residual
// form at half its rate).  Here the wave first parks all its R accumulator rows in LDS -- the accumulators are dead from then on, which
// frees the registers -- and then walks the R * ITER pieces in chunks of PF: the raw side pieces of a whole chunk are requested
// back to back (no conversion, branch-free clamped addresses: nothing in that phase reads a loaded register, so the compiler places
// no wait inside it), then the chunk is converted, combined and stored.  A tile costs R * ITER / PF round trips instead of R * ITER.
template <typename RAW>
__device__ __forceinline__ RAW side_raw4(const tr_view& v, int64_t pix, int c) {   // RAW = uint2: 16-bit views, uint4: fp32 views
    const int64_t idx = pix * v.cstride + v.coff + c;
    return *(const RAW*)((const char*)v.ptr + idx * (int64_t)(sizeof(RAW) / 4));
}
__device__ __forceinline__ float4 side_cvt4(const tr_view& v, uint2 u) {   // same conversions as tr::view_load4(..., vec = true)
    float4 r;
    if (v.dtype == TR_BF16) {
        r.x = __uint_as_float(u.x << 16); r.y = __uint_as_float(u.x & 0xffff0000u);
        r.z = __uint_as_float(u.y << 16); r.w = __uint_as_float(u.y & 0xffff0000u);
    } else {
        r.x = tr::f16_to_f32((unsigned short)(u.x & 0xffffu)); r.y = tr::f16_to_f32((unsigned short)(u.x >> 16));
        r.z = tr::f16_to_f32((unsigned short)(u.y & 0xffffu)); r.w = tr::f16_to_f32((unsigned short)(u.y >> 16));
    }
    return r;
}
__device__ __forceinline__ float4 side_cvt4(const tr_view&, uint4 u) {
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}

// SIDE: 0 = no side inputs, 1 = all of them 16-bit views (8-byte pieces), 2 = all of them fp32 views (16-byte pieces)
template <int NT, int R, int SIDE>
__device__ __forceinline__ void conv_epilogue_rows_impl(const ConvArgs& a, f32x16 (&acc)[R][NT], char* smem, int nb, int y0, int x0, int cg, int wv,
                                                        int l) {
    constexpr int CGW = NT * 32;
    constexpr int ROWB = CGW * 4 + 16;
    constexpr int SEGS = CGW / 4;            // 16-byte fp32 segments per pixel
    constexpr int ITER = 32 * SEGS / 64;     // pieces per lane and row
    constexpr int PPI = 64 / SEGS;           // pixels one wave-wide piece covers
    constexpr int PF = (NT == 2 && SIDE != 2) ? 4 : 2;   // pieces per chunk, by the register room next to the staged pieces of the next tile
    typedef typename std::conditional<SIDE == 2, uint4, uint2>::type raw_t;
    static_assert(64 % SEGS == 0 && ITER % PF == 0, "a lane keeps its channel segment; chunks do not straddle rows");
    const int lr = l & 31, lh = l >> 5;
    char* strip = smem + wv * (R * 32 * ROWB);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (y0 + wv * R + r >= a.h) break;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(float4*)(strip + (r * 32 + lr) * ROWB + (t * 32 + q * 8 + lh * 4) * 4) =
                    make_float4(acc[r][t][q * 4 + 0], acc[r][t][q * 4 + 1], acc[r][t][q * 4 + 2], acc[r][t][q * 4 + 3]);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int seg = l % SEGS, p_lane = l / SEGS;
    const int co = cg * CGW + seg * 4;
    if (co >= a.cout) return;                 // cout % 4 == 0 and 4-channel-aligned views on this path (conv_epilogue_rows): whole pieces only
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (co + k < a.cout) bias4[k] = a.bias[co + k];
    }
    // read before the loop: a global load inside it would put an s_waitcnt vmcnt(0) -- which also waits for the stores -- into every piece
    const float ss = a.sample_scale ? a.sample_scale[nb] : 1.f;
#pragma unroll 1
    for (int c = 0; c < R * ITER / PF; ++c) {
        const int r = c / (ITER / PF);
        const int oy = y0 + wv * R + r;
        if (oy >= a.h) break;
        const int i0 = (c - r * (ITER / PF)) * PF;
        const int64_t rowpix = ((int64_t)nb * a.h + oy) * a.w;
        raw_t q1[PF], q2[PF], qm[PF];
        if constexpr (SIDE != 0) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int oxc = min(x0 + (i0 + u) * PPI + p_lane, a.w - 1);        // columns past the image edge re-read the last one
                if (a.res1.ptr) q1[u] = side_raw4<raw_t>(a.res1, rowpix + oxc, co);
                if (a.res2.ptr) q2[u] = side_raw4<raw_t>(a.res2, rowpix + oxc, co);
                if (a.mask.ptr) qm[u] = side_raw4<raw_t>(a.mask, rowpix + oxc, co);
            }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int p = (i0 + u) * PPI + p_lane;
            const int ox = x0 + p;
            const float4 raw = *(const float4*)(strip + (r * 32 + p) * ROWB + seg * 16);
            if (ox >= a.w) continue;
            const int64_t pix = rowpix + ox;
            float v[4] = {raw.x + bias4[0], raw.y + bias4[1], raw.z + bias4[2], raw.w + bias4[3]};
            if (a.act == TR_ACT_LRELU) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * a.slope;
            } else if (a.act == TR_ACT_RELU) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
            }
            float pre[4] = {v[0], v[1], v[2], v[3]};   // value before GELU / SiLU / Mish: what out2 stores in those modes
            if (a.act == TR_ACT_GELU) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = 0.5f * v[k] * (1.f + erff(v[k] * 0.70710678118654752f));
            } else if (a.act == TR_ACT_SILU) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = v[k] / (1.f + __expf(-v[k]));
            } else if (a.act == TR_ACT_MISH) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = tr::mish(v[k]);
            }
            if (a.sample_scale) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] *= ss;
            }
            if constexpr (SIDE != 0) {
                if (a.res1.ptr) {
                    const float4 rr = side_cvt4(a.res1, q1[u]);
                    v[0] = v[0] * a.scale1 + a.rs1 * rr.x; v[1] = v[1] * a.scale1 + a.rs1 * rr.y;
                    v[2] = v[2] * a.scale1 + a.rs1 * rr.z; v[3] = v[3] * a.scale1 + a.rs1 * rr.w;
                }
                if (a.res2.ptr) {
                    const float4 rr = side_cvt4(a.res2, q2[u]);
                    v[0] = v[0] * a.scale2 + a.rs2 * rr.x; v[1] = v[1] * a.scale2 + a.rs2 * rr.y;
                    v[2] = v[2] * a.scale2 + a.rs2 * rr.z; v[3] = v[3] * a.scale2 + a.rs2 * rr.w;
                }
                if (a.mask.ptr) {
                    const float4 mm = side_cvt4(a.mask, qm[u]);
                    if (a.mask_mode == 1) {
                        v[0] *= gelu_grad(mm.x); v[1] *= gelu_grad(mm.y); v[2] *= gelu_grad(mm.z); v[3] *= gelu_grad(mm.w);
                    } else if (a.mask_mode == 2) {
                        v[0] *= silu_grad(mm.x); v[1] *= silu_grad(mm.y); v[2] *= silu_grad(mm.z); v[3] *= silu_grad(mm.w);
                    } else if (a.mask_mode == 3) {
                        v[0] *= tr::mish_grad(mm.x); v[1] *= tr::mish_grad(mm.y); v[2] *= tr::mish_grad(mm.z); v[3] *= tr::mish_grad(mm.w);
                    } else {
                        v[0] *= mm.x > 0.f ? 1.f : a.mask_slope; v[1] *= mm.y > 0.f ? 1.f : a.mask_slope;
                        v[2] *= mm.z > 0.f ? 1.f : a.mask_slope; v[3] *= mm.w > 0.f ? 1.f : a.mask_slope;
                    }
                }
            }
            tr::view_store4(a.out, pix, co, make_float4(v[0], v[1], v[2], v[3]), true);
            if (a.out2.ptr) {
                const float sc = a.out2_scale;
                if (a.act >= TR_ACT_GELU) tr::view_store4(a.out2, pix, co, make_float4(pre[0], pre[1], pre[2], pre[3]), true);
                else tr::view_store4(a.out2, pix, co, make_float4(v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc), true);
            }
        }
    }
}

template <int NT, int R>
__device__ __forceinline__ void conv_epilogue_rows(const ConvArgs& a, f32x16 (&acc)[R][NT], char* smem, int nb, int y0, int x0, int cg, int wv,
                                                   int l, bool wave_active) {
    __syncthreads();                          // every wave is done with the staged operands: LDS is free
    if (!wave_active || (a.dbg & 4)) return;
    const bool side = a.res1.ptr || a.res2.ptr || a.mask.ptr;
    // side inputs present, and how many of them are fp32 views: the chunked body takes them all 16-bit or all fp32
    const int nside = (a.res1.ptr != nullptr) + (a.res2.ptr != nullptr) + (a.mask.ptr != nullptr);
    const int wide = (a.res1.ptr && a.res1.dtype == TR_F32) + (a.res2.ptr && a.res2.dtype == TR_F32) + (a.mask.ptr && a.mask.dtype == TR_F32);
    const bool whole = a.vec_out && a.vec_out2 && a.vec_res1 && a.vec_res2 && a.vec_mask && (a.cout & 3) == 0 && (wide == 0 || wide == nside);
    if (!whole || (a.dbg & 8)) {       // (dbg 8: A/B switch) unaligned views / ragged channel counts: the row-at-a-time body with its scalar tails (wave wv's strip sits inside its rows area)
        if (side) conv_epilogue_impl<NT, R, true>(a, acc, smem + wv * ((R - 1) * 32 * (NT * 32 * 4 + 16)), nb, y0, x0, cg, wv, l, wave_active, -1);
        else conv_epilogue_impl<NT, R, false>(a, acc, smem + wv * ((R - 1) * 32 * (NT * 32 * 4 + 16)), nb, y0, x0, cg, wv, l, wave_active, -1);
    } else if (!side) conv_epilogue_rows_impl<NT, R, 0>(a, acc, smem, nb, y0, x0, cg, wv, l);
    else if (wide == 0) conv_epilogue_rows_impl<NT, R, 1>(a, acc, smem, nb, y0, x0, cg, wv, l);
    else conv_epilogue_rows_impl<NT, R, 2>(a, acc, smem, nb, y0, x0, cg, wv, l);
}

// ---- variant A: register-staged (global -> VGPR -> LDS), persistent over tiles ------------------------------------------
// A workgroup walks tiles (cout group fastest, so neighbours in the walk share the halo in L2).  While the MFMAs of
// (tile t, chunk c) run, the 16-byte pieces of the NEXT (tile, chunk) -- possibly the first chunk of the next tile -- are
// already in flight into registers, so neither the first-load latency of a tile nor its epilogue stores are exposed.
// One LDS stage; two workgroups per CU cover each other's LDS write phases.
template <int CT, int NT, int R, int KS>
__global__ __launch_bounds__(NTHREADS, (NT == 1 && R == 2 && (CT == TR_BF16 || CT == TR_F16)) ? 3 : 2) void conv3x3_kernel_reg(ConvArgs a) {
    constexpr int TH = 4 * R;
    constexpr int PAD = KS / 2;                        // KS = 3: 3x3 pad 1 ; KS = 1: 1x1 (token GEMMs: Linear layers)
    constexpr int HALO_W = TW + KS - 1;
    constexpr int HALO_PIX = (TH + KS - 1) * HALO_W;
    constexpr int HALO_ITEMS = HALO_PIX * 4;           // 16-byte slots
    constexpr int HALO_BYTES = HALO_PIX * 64;
    constexpr int CGW = NT * 32;
    constexpr int W_ITEMS = KS * KS * CGW * 4;
    constexpr int ES = (CT == TR_F32 || CT == TR_BF16X3) ? 4 : 2;
    constexpr int EPS = 16 / ES;                       // elements per slot
    constexpr int CK = 64 / ES;                        // channels per chunk
    constexpr int HALO_IT = (HALO_ITEMS + NTHREADS - 1) / NTHREADS;
    constexpr int W_IT = (W_ITEMS + NTHREADS - 1) / NTHREADS;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* hl = smem;
    char* wl = smem + HALO_BYTES;

    const int tid = threadIdx.x;
    const int l = tid & 63, wv = tid >> 6;
    const int lr = l & 31, lh = l >> 5;
    // staging plan: item it = tid + 256*j -> halo pixel pp = (tid >> 2) + 64*j, slot s = tid & 3 (constant per thread, and so
    // is the swizzle term (pp >> 2) & 3), hence LDS offset = lds0 + 4096*j.
    const int s_slot = tid & 3;
    const int pp0 = tid >> 2;
    const int lds0 = pp0 * 64 + ((s_slot ^ ((pp0 >> 2) & 3)) << 4);
    const int ldsx_hi = pp0 * 64 + ((((s_slot >> 1)) ^ ((pp0 >> 2) & 3)) << 4) + (s_slot & 1) * 8;        // TR_BF16X3 split rows
    const int ldsx_lo = pp0 * 64 + (((2 + (s_slot >> 1)) ^ ((pp0 >> 2) & 3)) << 4) + (s_slot & 1) * 8;
    const char* in_base = (const char*)a.in.ptr + (int64_t)(a.in.coff + s_slot * EPS) * ES;
    const int64_t in_pix_bytes = (int64_t)a.in.cstride * ES;
    const int ntiles = a.ntx * a.nty * a.n * a.ncg;

    uint4 hreg[HALO_IT], wreg[W_IT];
    // decode flat tile -> (cg, tx, ty, nb); cg fastest
#define TR_DECODE(T, CG, NB, Y0, X0)                                                                                    \
    int CG, NB, Y0, X0;                                                                                                 \
    {                                                                                                                   \
        int t_ = (T);                                                                                                   \
        CG = t_ % a.ncg; t_ /= a.ncg;                                                                                    \
        X0 = (t_ % a.ntx) * TW; t_ /= a.ntx;                                                                             \
        Y0 = (t_ % a.nty) * TH;                                                                                          \
        NB = t_ / a.nty;                                                                                                 \
    }
#define TR_ISSUE_LOADS(T, CH)                                                                                           \
    {                                                                                                                   \
        TR_DECODE(T, cg_, nb_, y0_, x0_)                                                                                 \
        const bool cvalid = (((CH) * CK + s_slot * EPS) < a.cin) && !(a.dbg & 2);                                        \
        const char* src = in_base + (int64_t)(CH) * 64 + (int64_t)nb_ * a.hin * a.win * in_pix_bytes;                    \
        int ppb = pp0;                                                                                                  \
        asm volatile("" : "+v"(ppb)); /* recompute the per-piece coordinates here instead of keeping them live */       \
        _Pragma("unroll") for (int j = 0; j < HALO_IT; ++j) {                                                            \
            const int pp = ppb + 64 * j;                                                                                 \
            const int py = pp / HALO_W, px = pp - py * HALO_W;                                                           \
            int oy = y0_ + py - PAD - a.shift, ox = x0_ + px - PAD - a.shift;                                                \
            bool ok = (pp < HALO_PIX) & (oy >= 0) & (oy < a.ih) & (ox >= 0) & (ox < a.iw);                               \
            if (a.replicate) {                                                                                          \
                oy = min(max(oy, 0), a.ih - 1);                                                                          \
                ox = min(max(ox, 0), a.iw - 1);                                                                          \
                ok = pp < HALO_PIX;                                                                                      \
            }                                                                                                           \
            hreg[j] = make_uint4(0, 0, 0, 0);                                                                            \
            if (ok && cvalid) hreg[j] = *(const uint4*)(src + ((int64_t)(oy >> a.up) * a.win + (ox >> a.up)) * in_pix_bytes); \
        }                                                                                                               \
        const char* wsrc = a.wpacked + ((int64_t)cg_ * a.nchunks + (CH)) * (W_ITEMS * 16) + tid * 16;                    \
        _Pragma("unroll") for (int j = 0; j < W_IT; ++j) {                                                               \
            wreg[j] = make_uint4(0, 0, 0, 0);                                                                            \
            if (((W_ITEMS % NTHREADS == 0) || (tid + j * NTHREADS < W_ITEMS)) && !(a.dbg & 2))                           \
                wreg[j] = *(const uint4*)(wsrc + j * (NTHREADS * 16));                                                   \
        }                                                                                                               \
    }

    // Wide output groups (NT = 2) have only 2304 MFMA cycles per chunk to hide the address arithmetic of 15 pieces behind, so
    // they keep the per-tile piece offsets in registers (12 VGPRs) and recompute only for the cross-tile prefetch.
    constexpr bool PRE = NT == 2;
    int64_t goff[PRE ? HALO_IT : 1];
    unsigned hmask = 0;
#define TR_ISSUE_LOADS_PRE(CG, CH)                                                                                      \
    {                                                                                                                   \
        const bool cvalid = (((CH) * CK + s_slot * EPS) < a.cin) && !(a.dbg & 2);                                        \
        const char* src = in_base + (int64_t)(CH) * 64;                                                                  \
        _Pragma("unroll") for (int j = 0; j < HALO_IT; ++j) {                                                            \
            hreg[j] = make_uint4(0, 0, 0, 0);                                                                            \
            if (cvalid && ((hmask >> j) & 1u)) hreg[j] = *(const uint4*)(src + goff[PRE ? j : 0]);                       \
        }                                                                                                               \
        const char* wsrc = a.wpacked + ((int64_t)(CG) * a.nchunks + (CH)) * (W_ITEMS * 16) + tid * 16;                   \
        _Pragma("unroll") for (int j = 0; j < W_IT; ++j) {                                                               \
            wreg[j] = make_uint4(0, 0, 0, 0);                                                                            \
            if (((W_ITEMS % NTHREADS == 0) || (tid + j * NTHREADS < W_ITEMS)) && !(a.dbg & 2))                           \
                wreg[j] = *(const uint4*)(wsrc + j * (NTHREADS * 16));                                                   \
        }                                                                                                               \
    }

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    TR_ISSUE_LOADS(tile, 0)
    TR_WAIT_VMEM();    // staging registers are always complete at the top of a chunk (see the wait behind the MFMAs), on every path into it
    while (tile < ntiles) {
        TR_DECODE(tile, cg, nb, y0, x0)
        const bool wave_active = (y0 + wv * R) < a.h;
        const int next_tile = tile + gridDim.x;
        if constexpr (PRE) {
            hmask = 0;
#pragma unroll
            for (int j = 0; j < HALO_IT; ++j) {
                const int pp = pp0 + 64 * j;
                const int py = pp / HALO_W, px = pp - py * HALO_W;
                int oy = y0 + py - PAD - a.shift, ox = x0 + px - PAD - a.shift;
                bool ok = (pp < HALO_PIX) & (oy >= 0) & (oy < a.ih) & (ox >= 0) & (ox < a.iw);
                if (a.replicate) {
                    oy = min(max(oy, 0), a.ih - 1);
                    ox = min(max(ox, 0), a.iw - 1);
                    ok = pp < HALO_PIX;
                }
                goff[j] = ok ? (((int64_t)nb * a.hin + (oy >> a.up)) * a.win + (ox >> a.up)) * in_pix_bytes : 0;
                hmask |= ok ? (1u << j) : 0u;
            }
        }
        f32x16 acc[R][NT];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[r][t][i] = 0.f;
        for (int ch = 0; ch < a.nchunks; ++ch) {
            lds_barrier();     // every wave has finished reading the previous LDS image (or the previous tile's epilogue strips)
#pragma unroll
            for (int j = 0; j < HALO_IT; ++j)
                if ((HALO_ITEMS % NTHREADS == 0) || (pp0 + 64 * j < HALO_PIX)) {
                    if constexpr (CT == TR_BF16X3) {
                        // 4 fp32 channels -> 4 bf16 hi + 4 bf16 lo; piece s covers channels 4s..4s+3: hi bytes [8s, 8s+8), lo bytes 32 + [8s, 8s+8)
                        const float f0 = __uint_as_float(hreg[j].x), f1 = __uint_as_float(hreg[j].y), f2 = __uint_as_float(hreg[j].z), f3 = __uint_as_float(hreg[j].w);
                        const unsigned h01 = pack_bf16x2(f0, f1), h23 = pack_bf16x2(f2, f3);
                        const unsigned l01 = pack_bf16x2(f0 - __uint_as_float(h01 << 16), f1 - __uint_as_float(h01 & 0xffff0000u));
                        const unsigned l23 = pack_bf16x2(f2 - __uint_as_float(h23 << 16), f3 - __uint_as_float(h23 & 0xffff0000u));
                        *(uint2*)(hl + ldsx_hi + 4096 * j) = make_uint2(h01, h23);
                        *(uint2*)(hl + ldsx_lo + 4096 * j) = make_uint2(l01, l23);
                    } else {
                        *(uint4*)(hl + lds0 + 4096 * j) = hreg[j];
                    }
                }
#pragma unroll
            for (int j = 0; j < W_IT; ++j)
                if ((W_ITEMS % NTHREADS == 0) || (tid + j * NTHREADS < W_ITEMS)) *(uint4*)(wl + tid * 16 + j * (NTHREADS * 16)) = wreg[j];
            lds_barrier();
            // next pieces go in flight while this chunk's MFMAs run
            if (ch + 1 < a.nchunks) {
                if constexpr (PRE) TR_ISSUE_LOADS_PRE(cg, ch + 1) else TR_ISSUE_LOADS(tile, ch + 1)
            } else if (next_tile < ntiles) TR_ISSUE_LOADS(next_tile, 0)
            if (wave_active && !(a.dbg & 1)) conv_chunk_mfma<CT, NT, R, KS>(hl, wl, acc, wv, lr, lh);
            // the pieces have had the MFMAs to land: waiting here -- in front of the epilogue at the last chunk, whose pieces belong to the
            // next tile -- keeps the epilogue's stores out of the wait at the top of the next chunk (vmcnt is one in-order counter)
            TR_WAIT_VMEM();
        }
        // the 8-row / 32-channel tiling lives on 168 VGPRs (three workgroups per CU): the chunked body's piece arrays spill there
        if constexpr (NT == 1 && R == 2) {
            __syncthreads();
            if (wave_active && !(a.dbg & 4)) conv_epilogue_impl<NT, R, true>(a, acc, smem, nb, y0, x0, cg, wv, l, wave_active, -1);   // one body: a second one spills here
        }
        else conv_epilogue_rows<NT, R>(a, acc, smem, nb, y0, x0, cg, wv, l, wave_active);
        tile = next_tile;
    }
#undef TR_ISSUE_LOADS
#undef TR_ISSUE_LOADS_PRE
#undef TR_DECODE
}

// ---- variant B: LDS-DMA staged (buffer_load ... lds), two LDS stages, one workgroup per CU -----------------------
template <int CT, int NT, int R>
__global__ __launch_bounds__(NTHREADS) void conv3x3_kernel_dma(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource / LDS-DMA builtins exist only in the device pass; the host pass needs just the stub
    constexpr int TH = 4 * R;
    constexpr int HALO_PIX = (TH + 2) * HALO_W;
    constexpr int HALO_ITEMS = HALO_PIX * 4;                                 // 16-byte slots
    constexpr int HALO_IT = (HALO_ITEMS + NTHREADS - 1) / NTHREADS;           // LDS-DMA instructions per wave (halo)
    constexpr int HALO_BYTES = HALO_IT * NTHREADS * 16;                       // padded to whole 4 KiB rounds
    constexpr int CGW = NT * 32;
    constexpr int W_ITEMS = 9 * CGW * 4;
    constexpr int W_IT = (W_ITEMS + NTHREADS - 1) / NTHREADS;
    constexpr int W_BYTES = W_IT * NTHREADS * 16;
    constexpr int STAGE_BYTES = HALO_BYTES + W_BYTES;
    constexpr int ES = (CT == TR_F32 || CT == TR_BF16X3) ? 4 : 2;
    constexpr int EPS = 16 / ES;                       // elements per slot
    constexpr int CK = 64 / ES;                        // channels per chunk
    constexpr unsigned OOB = 0x7fffff00u;              // beyond any num_records: the buffer load returns 0

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int l = tid & 63, wv = tid >> 6;
    const int lr = l & 31, lh = l >> 5;

    int bid = blockIdx.x;
    const int tx = bid % a.ntx; bid /= a.ntx;
    const int ty = bid % a.nty;
    const int nb = bid / a.nty;
    const int cg = blockIdx.y;

    const int y0 = ty * TH, x0 = tx * TW;
    const bool wave_active = (y0 + wv * R) < a.h;

    f32x16 acc[R][NT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][t][i] = 0.f;

    // ---- staging plan: LDS-DMA (buffer_load ... lds), two LDS stages ---------------------------------------
    // One wave instruction lands 64 x 16 B contiguously in LDS (wave-uniform base + lane*16), so the LDS image is
    // position-linear: position P = 256*j + tid  <->  halo pixel pp = P >> 2, stored slot s' = P & 3.  The bank swizzle
    // therefore goes on the SOURCE: the lane fetches channel slot s = s' ^ ((pp >> 2) & 3) of that pixel (constant per
    // thread).  Pixels outside the image (zero padding) and the tail of the last round use an out-of-range buffer
    // offset, for which the hardware returns zeros.  The K-chunk advances through the scalar offset.
    const int pp0 = tid >> 2;
    const int s_src = (tid & 3) ^ ((pp0 >> 2) & 3);
    const int64_t in_pix_bytes = (int64_t)a.in.cstride * ES;
    const char* img_base = (const char*)a.in.ptr + (int64_t)nb * a.hin * a.win * in_pix_bytes;
    const unsigned img_bytes = (unsigned)((int64_t)a.hin * a.win * in_pix_bytes);
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)img_base, 0, img_bytes, 0x00020000);
    const char* w_cg = a.wpacked + (int64_t)cg * a.nchunks * (W_ITEMS * 16);
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w_cg, 0, (unsigned)(a.nchunks * W_ITEMS * 16), 0x00020000);
    unsigned hoff[HALO_IT];
#pragma unroll
    for (int j = 0; j < HALO_IT; ++j) {
        const int pp = pp0 + 64 * j;
        const int py = pp / HALO_W, px = pp - py * HALO_W;
        int oy = y0 + py - 1 - a.shift, ox = x0 + px - 1 - a.shift;
        bool ok = (pp < HALO_PIX) & (oy >= 0) & (oy < a.ih) & (ox >= 0) & (ox < a.iw);
        if (a.replicate) {
            oy = min(max(oy, 0), a.ih - 1);
            ox = min(max(ox, 0), a.iw - 1);
            ok = pp < HALO_PIX;
        }
        const int sy = oy >> a.up, sx = ox >> a.up;
        const unsigned off = (unsigned)(((int64_t)sy * a.win + sx) * in_pix_bytes + (int64_t)(a.in.coff + s_src * EPS) * ES);
        hoff[j] = ok ? off : OOB;
    }
    const unsigned woff0 = (unsigned)tid * 16u;
    const bool partial_k = (a.cin % CK) != 0;
    if (partial_k) {
        // channel slots past cin are never fetched (masked below): give them a defined value once
        for (int i = tid; i < (2 * STAGE_BYTES) / 16; i += NTHREADS) *(uint4*)(smem + i * 16) = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }

#define TR_STAGE(CH, BUF)                                                                                               \
    {                                                                                                                   \
        char* hb = smem + (BUF) * STAGE_BYTES + wv * 1024;                                                               \
        const bool cvalid = !partial_k || (((CH) * CK + s_src * EPS) < a.cin);                                           \
        if (cvalid) {                                                                                                   \
            _Pragma("unroll") for (int j = 0; j < HALO_IT; ++j)                                                          \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (__attribute__((address_space(3))) void*)(hb + j * 4096), 16, \
                                                         hoff[j], (CH) * 64, 0, 0);                                      \
        }                                                                                                               \
        char* wb = smem + (BUF) * STAGE_BYTES + HALO_BYTES + wv * 1024;                                                  \
        _Pragma("unroll") for (int j = 0; j < W_IT; ++j) {                                                               \
            if ((W_ITEMS % NTHREADS == 0) || (j * NTHREADS + wv * 64 < W_ITEMS))                                         \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(wb + j * 4096), 16, \
                                                         woff0 + j * 4096, (CH) * (W_ITEMS * 16), 0, 0);                 \
        }                                                                                                               \
    }

    TR_STAGE(0, 0)
    for (int ch = 0; ch < a.nchunks; ++ch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of chunk `ch` have landed
        __syncthreads();                                   // ... and everybody's; all waves are done reading the other stage
        if (ch + 1 < a.nchunks) TR_STAGE(ch + 1, (ch + 1) & 1)   // in flight while this chunk's MFMAs run
        const char* hl = smem + (ch & 1) * STAGE_BYTES;
        if (wave_active) conv_chunk_mfma<CT, NT, R>(hl, hl + HALO_BYTES, acc, wv, lr, lh);
    }
#undef TR_STAGE

    conv_epilogue<NT, R>(a, acc, smem, nb, y0, x0, cg, wv, l, wave_active);
#endif
}

template <int CT, int NT, int R, int KS>
int launch_reg(hipStream_t st, const ConvArgs& a, int ncg) {
    constexpr size_t stage = (size_t)(4 * R + KS - 1) * (TW + KS - 1) * 64 + (size_t)KS * KS * NT * 32 * 64;
    constexpr size_t strips = (size_t)4 * R * 32 * (NT * 32 * 4 + 16);      // epilogue transposition strips: all R rows of every wave
    constexpr size_t lds = stage > strips ? stage : strips;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv3x3_kernel_reg<CT, NT, R, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    // resident workgroups: 2 per CU, or 3 for the 8-row / 32-channel tile (164 VGPRs, 40 KB LDS) -- measured +15 % on the thin
    // RDB convs (cin <= 96), whose K loop is too short to hide the load latency with two workgroups
    static const int persist_env = getenv("TRAINNER_CONV_WGS") ? atoi(getenv("TRAINNER_CONV_WGS")) : 0;
    const int persist = persist_env ? persist_env : ((NT == 1 && R == 2 && (CT == TR_BF16 || CT == TR_F16)) ? 768 : 512);
    const int64_t ntiles = (int64_t)a.ntx * a.nty * a.n * ncg;
    const unsigned grid = (unsigned)(ntiles < persist ? ntiles : persist);
    tr::tune().last_conv_family = 1;
    hipLaunchKernelGGL((conv3x3_kernel_reg<CT, NT, R, KS>), dim3(grid), dim3(NTHREADS), lds, st, a);
    return tr::launch_status("tr_conv3x3");
}


// ---- straight-line epilogue of the token GEMM kernels (round 3) ----------------------------------------------------------------------------
// The generic epilogue above walks a row in a ROLLED loop whose body branches at run time on activation, dtypes, side views and mask mode and, with a
// residual, waits for one global round trip per 4-channel piece: measured on SwinIR-M's Linears it was 13 - 25 us of a 24 - 34 us launch
// (tools/bench_linear.py with TRAINNER_CONV_DBG=4: 192 -> 540 24.3 -> 11.1 us, 192 -> 180 + residual 26.1 -> 7.2 us without it).  The five forms SwinIR's
// forward and backward use get compile-time bodies: all side pieces (residual, mask) of a row are requested back to back, then the row is
// finished and stored; no run-time dispatch inside.  The bias is not a side piece: the kernels seed their accumulators with it when a form is set
// (12 float4 of bias per row next to 12 of residual and the 96 accumulator registers spilled 93 dwords at NT = 3).  Semantics as conv_epilogue_impl: bias, [GELU with the pre-activation to out2], per-sample
// scale, residual, GELU-derivative mask.  Host-checked: cout % 4 == 0, vector-accessible views, w == 32.
enum { LF_PLAIN_CD = 0, LF_PLAIN_F32 = 1, LF_RES_F32 = 2, LF_GELU_CD = 3, LF_MASKG_CD = 4 };

template <int CT>
__device__ __forceinline__ uint2 lin_pack4(float a, float b, float c, float d) {
    if constexpr (CT == TR_BF16) return make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
    else {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        const f32x2c u = {a, b}, v = {c, d};
        return make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(u, h2)), __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2)));
    }
}
template <int CT>
__device__ __forceinline__ float4 lin_unpack4(uint2 u) {
    if constexpr (CT == TR_BF16) return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    else return make_float4(tr::f16_to_f32(u.x & 0xffffu), tr::f16_to_f32(u.x >> 16), tr::f16_to_f32(u.y & 0xffffu), tr::f16_to_f32(u.y >> 16));
}

template <int CT, int NT, int FORM>
__device__ __forceinline__ void lin_epi_form(const ConvArgs& a, f32x16 (&acc)[2][NT], char* strip, int nb, int row0, int cbase, int l) {
    constexpr int CGW = NT * 32, ROWB = CGW * 4 + 16, SEGS = CGW / 4, ITER = 32 * SEGS / 64;
    const int lr = l & 31, lh = l >> 5;
    const float ss = a.sample_scale ? a.sample_scale[nb] : 1.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int oy = row0 + r;
        if (oy >= a.h) break;
        const int64_t tok0 = ((int64_t)nb * a.h + oy) * 32;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(float4*)(strip + lr * ROWB + (t * 32 + q * 8 + lh * 4) * 4) =
                    make_float4(acc[r][t][q * 4 + 0], acc[r][t][q * 4 + 1], acc[r][t][q * 4 + 2], acc[r][t][q * 4 + 3]);
        __builtin_amdgcn_wave_barrier();
        // side pieces of the whole row first: one memory latency per row
        float4 side[(FORM == LF_RES_F32) ? ITER : 1];
        uint2 mk[(FORM == LF_MASKG_CD) ? ITER : 1];
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int idx = i * 64 + l;
            const int p = idx / SEGS, seg = idx - p * SEGS;
            const int co = min(cbase + seg * 4, a.cout - 4);            // clamped: loads stay in range, stores are masked below
            (void)co;
            if constexpr (FORM == LF_RES_F32) side[i] = *(const float4*)((const float*)a.res1.ptr + (tok0 + p) * a.res1.cstride + a.res1.coff + co);
            if constexpr (FORM == LF_MASKG_CD) mk[i] = *(const uint2*)((const unsigned short*)a.mask.ptr + (tok0 + p) * a.mask.cstride + a.mask.coff + co);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int idx = i * 64 + l;
            const int p = idx / SEGS, seg = idx - p * SEGS;
            const int co = cbase + seg * 4;
            const float4 raw = *(const float4*)(strip + p * ROWB + seg * 16);
            float v[4] = {raw.x, raw.y, raw.z, raw.w};            // the bias seeded the accumulators (lin_seed_bias)
            const bool ok = co < a.cout;
            const int64_t tok = tok0 + p;
            if constexpr (FORM == LF_GELU_CD) {
                if (ok) *(uint2*)((unsigned short*)a.out2.ptr + tok * a.out2.cstride + a.out2.coff + co) = lin_pack4<CT>(v[0], v[1], v[2], v[3]);      // pre-activation
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = gelu_fast(v[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] *= ss;
            if constexpr (FORM == LF_RES_F32) {
                v[0] = v[0] * a.scale1 + a.rs1 * side[i].x; v[1] = v[1] * a.scale1 + a.rs1 * side[i].y;
                v[2] = v[2] * a.scale1 + a.rs1 * side[i].z; v[3] = v[3] * a.scale1 + a.rs1 * side[i].w;
            }
            if constexpr (FORM == LF_MASKG_CD) {
                const float4 m = lin_unpack4<CT>(mk[i]);
                v[0] *= gelu_grad_fast(m.x); v[1] *= gelu_grad_fast(m.y); v[2] *= gelu_grad_fast(m.z); v[3] *= gelu_grad_fast(m.w);
            }
            if (ok) {
                if constexpr (FORM == LF_PLAIN_F32 || FORM == LF_RES_F32) {
                    *(float4*)((float*)a.out.ptr + tok * a.out.cstride + a.out.coff + co) = make_float4(v[0], v[1], v[2], v[3]);
                    if (FORM == LF_RES_F32 && a.out2.ptr) {
                        const float sc = a.out2_scale;
                        *(uint2*)((unsigned short*)a.out2.ptr + tok * a.out2.cstride + a.out2.coff + co) = lin_pack4<CT>(v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc);
                    }
                } else {
                    *(uint2*)((unsigned short*)a.out.ptr + tok * a.out.cstride + a.out.coff + co) = lin_pack4<CT>(v[0], v[1], v[2], v[3]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();       // the strip is rewritten by the next row
    }
}

// accumulator seed of the straight-line forms: lane (lr, lh) holds channels t * 32 + q * 8 + lh * 4 + k of its pixel in acc[.][t][q * 4 + k]
template <int NT>
__device__ __forceinline__ void lin_seed_bias(const ConvArgs& a, f32x16 (&acc)[2][NT], int cbase, int lh) {
    if (!a.bias) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co = cbase + t * 32 + q * 8 + lh * 4;
            const float4 b = co < a.cout ? *(const float4*)(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 2; ++r) { acc[r][t][q * 4 + 0] = b.x; acc[r][t][q * 4 + 1] = b.y; acc[r][t][q * 4 + 2] = b.z; acc[r][t][q * 4 + 3] = b.w; }
        }
}

template <int CT, int NT>
__device__ __forceinline__ void lin_epilogue_fast(const ConvArgs& a, f32x16 (&acc)[2][NT], char* smem, int nb, int row0, int cbase, int wv, int l, bool wave_active) {
    __syncthreads();                          // every wave is done with the staged operands: LDS is free
    if (!wave_active || (a.dbg & 4)) return;
    char* strip = smem + wv * (32 * (NT * 32 * 4 + 16));
    if constexpr (CT == TR_BF16 || CT == TR_F16) {
        switch (a.lin_form) {                 // wave-uniform
            case LF_PLAIN_CD: lin_epi_form<CT, NT, LF_PLAIN_CD>(a, acc, strip, nb, row0, cbase, l); break;
            case LF_PLAIN_F32: lin_epi_form<CT, NT, LF_PLAIN_F32>(a, acc, strip, nb, row0, cbase, l); break;
            case LF_RES_F32: lin_epi_form<CT, NT, LF_RES_F32>(a, acc, strip, nb, row0, cbase, l); break;
            case LF_GELU_CD: lin_epi_form<CT, NT, LF_GELU_CD>(a, acc, strip, nb, row0, cbase, l); break;
            default: lin_epi_form<CT, NT, LF_MASKG_CD>(a, acc, strip, nb, row0, cbase, l); break;
        }
    }
}

// ---- Linear layers: ksize 1 on a 32-wide token grid, as a GEMM with up to 192 output channels per workgroup ---------------------
// A 1x1 "conv" has no tap reuse: with the 3x3 tiling (one 64-channel group per workgroup) the token rows are re-read by every
// output group and the kernel runs at L2 speed (measured 37 TFLOP/s on SwinIR's qkv/proj/fc1/fc2).  Here a workgroup owns
// 128 tokens x NCG 64-channel groups of the SAME packed slab: 2 x 2 waves, each 64 tokens x NCG*32 channels (2 x NCG MFMA tiles),
// so the token tile is fetched cout/192 times instead of cout/64 and each LDS operand read feeds 1.2-1.5 MFMAs.  Register-staged
// prefetch of the next K-chunk, 2-3 workgroups per CU; same MFMA chunk routine and fused epilogue as the 3x3 kernel.  Workgroup
// ids are dealt so that the workgroups sharing a token tile sit on one XCD (ids round-robin over the 8 XCDs) and share its L2.
template <int CT, int NCG>
__global__ __launch_bounds__(NTHREADS, 2) void linear_kernel(ConvArgs a, int nblk) {
    constexpr int R = 2, NT = NCG;
    constexpr int ES = (CT == TR_F32 || CT == TR_BF16X3) ? 4 : 2;
    constexpr int EPS = 16 / ES;
    constexpr int CK = 64 / ES;
    constexpr int X_BYTES = 128 * 64;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* hl = smem;
    char* wl = smem + X_BYTES;

    const int tid = threadIdx.x;
    const int l = tid & 63, wv = tid >> 6;
    const int lr = l & 31, lh = l >> 5;
    const int wr = wv >> 1, wc = wv & 1;

    const int total = nblk * a.nty * a.n;
    const int per = (total + 7) >> 3;
    int t_ = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t_ >= total) return;
    const int cb = t_ % nblk; t_ /= nblk;
    const int y0 = (t_ % a.nty) * 4;
    const int nb = t_ / a.nty;

    const int s_slot = tid & 3, pp0 = tid >> 2;
    const int lds0 = pp0 * 64 + ((s_slot ^ ((pp0 >> 2) & 3)) << 4);
    const int64_t pixb = (int64_t)a.in.cstride * ES;
    const char* src0 = (const char*)a.in.ptr + (int64_t)(a.in.coff + s_slot * EPS) * ES + (((int64_t)nb * a.h + y0) * 32 + pp0) * pixb;
    const bool rowok0 = (y0 + (pp0 >> 5)) < a.h, rowok1 = (y0 + (pp0 >> 5) + 2) < a.h;
    const char* wsrc0 = a.wpacked + (int64_t)(cb * NCG) * a.nchunks * 4096 + tid * 16;

    uint4 xreg[2], wreg[NCG];
#define TR_LIN_ISSUE(CH)                                                                                                \
    {                                                                                                                   \
        const bool cvalid = (((CH) * CK + s_slot * EPS) < a.cin) && !(a.dbg & 2);                                       \
        xreg[0] = make_uint4(0, 0, 0, 0);                                                                                \
        xreg[1] = make_uint4(0, 0, 0, 0);                                                                                \
        if (cvalid && rowok0) xreg[0] = *(const uint4*)(src0 + (int64_t)(CH) * 64);                                      \
        if (cvalid && rowok1) xreg[1] = *(const uint4*)(src0 + (int64_t)(CH) * 64 + 64 * pixb);                          \
        _Pragma("unroll") for (int g = 0; g < NCG; ++g) {                                                                \
            wreg[g] = make_uint4(0, 0, 0, 0);                                                                            \
            if (cb * NCG + g < a.ncg && !(a.dbg & 2)) wreg[g] = *(const uint4*)(wsrc0 + ((int64_t)g * a.nchunks + (CH)) * 4096); \
        }                                                                                                               \
    }

    f32x16 acc[R][NT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][t][i] = 0.f;
    const bool wave_active = (y0 + wr * R) < a.h;
    if constexpr (CT == TR_BF16 || CT == TR_F16) { if (a.lin_form >= 0) lin_seed_bias<NT>(a, acc, (cb * 2 + wc) * NT * 32, lh); }
    TR_LIN_ISSUE(0)
    for (int ch = 0; ch < a.nchunks; ++ch) {
        __syncthreads();
        *(uint4*)(hl + lds0) = xreg[0];
        *(uint4*)(hl + lds0 + 4096) = xreg[1];
#pragma unroll
        for (int g = 0; g < NCG; ++g) *(uint4*)(wl + g * 4096 + tid * 16) = wreg[g];
        __syncthreads();
        if (ch + 1 < a.nchunks) TR_LIN_ISSUE(ch + 1)
        if (wave_active && !(a.dbg & 1)) conv_chunk_mfma<CT, NT, R, 1>(hl, wl + wc * (NCG * 32 * 64), acc, wr, lr, lh);
    }
#undef TR_LIN_ISSUE
    if ((CT == TR_BF16 || CT == TR_F16) && a.lin_form >= 0) lin_epilogue_fast<CT, NT>(a, acc, smem, nb, y0 + wr * R, (cb * 2 + wc) * NT * 32, wv, l, wave_active);
    else conv_epilogue<NT, R>(a, acc, smem, nb, y0, 0, cb * 2 + wc, wr, l, wave_active, wv);
}

// ---- the same GEMM tiling with LDS-DMA staging (round 3) ------------------------------------------------------------------------------
// linear_kernel above prefetches ONE K-chunk ahead through registers: with K = 192 .. 384 (6 - 12 chunks) a workgroup is a chain of dependent
// global round trips and every SwinIR Linear took >= 20 us whatever its FLOPs (tools/bench_linear.py: 192 -> 180, 2.3 GFLOP, 25.9 us; 192 -> 540,
// 6.8 GFLOP, 24.4 us).  Here the operands of chunk ch + 3 go global -> LDS by buffer_load ... lds (no VGPR round trip) into a four-stage ring while
// chunk ch is multiplied: three chunks (60 KB per workgroup, two workgroups per CU) are in flight, one barrier per chunk, a counted vmcnt wait.
// The LDS image is position-linear under DMA (lane l of wave-instruction q lands at (q * 64 + l) * 16), so the bank swizzle of the token rows is
// applied on the SOURCE address; the weight rows are already stored as their LDS image (pack.hip).  Same MFMA chunk routine, same epilogue.
template <int NP>
__device__ __forceinline__ void lin_wait_chunks_left(int left) {      // wait until at most `left` newer chunks (NP pieces each) of this wave are outstanding
    if (left >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
    else if (left == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int CT, int NCG, int S>
__global__ __launch_bounds__(NTHREADS, S <= 2 ? 3 : 2) void linear_dma_kernel(ConvArgs a, int nblk) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int R = 2, NT = NCG;
    constexpr int ES = 2, EPS = 8, CK = 32;                  // 16-bit operands only
    constexpr int X_BYTES = 128 * 64;
    constexpr int STAGE = X_BYTES + NCG * 4096;
    constexpr int P = S - 1;                                 // S ring stages, P chunks requested ahead
    constexpr int NP = 2 + NCG;                              // DMA pieces per wave and chunk
    constexpr unsigned OOB = 0x7fffff00u;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int l = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = l & 31, lh = l >> 5;
    const int wr = wv >> 1, wc = wv & 1;

    const int total = nblk * a.nty * a.n;
    const int per = (total + 7) >> 3;
    int t_ = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t_ >= total) return;
    const int cb = t_ % nblk; t_ /= nblk;
    const int y0 = (t_ % a.nty) * 4;
    const int nb = t_ / a.nty;

    // token rows: piece (q, l) of a chunk = pixel pp = q * 64 + wv * 16 + (l >> 2) (q = 0, 1), stored slot l & 3, source slot swizzled by the pixel
    const int64_t pixb = (int64_t)a.in.cstride * ES;
    const unsigned img_bytes = (unsigned)((int64_t)a.h * 32 * pixb);
    const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in.ptr + (int64_t)nb * a.h * 32 * pixb), 0, img_bytes, 0x00020000);
    unsigned xoff[2];
    int xslot[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int pp = q * 64 + wv * 16 + (l >> 2);
        const int s = (l & 3) ^ ((pp >> 2) & 3);
        const bool rowok = (y0 + (pp >> 5)) < a.h;
        xslot[q] = s;
        xoff[q] = rowok ? (unsigned)(((int64_t)y0 * 32 + pp) * pixb + (int64_t)(a.in.coff + s * EPS) * ES) : OOB;
    }
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpacked, 0, (unsigned)((int64_t)a.ncg * a.nchunks * 4096), 0x00020000);
    unsigned woff[NCG];
#pragma unroll
    for (int g = 0; g < NCG; ++g) woff[g] = (cb * NCG + g < a.ncg) ? (unsigned)(((int64_t)(cb * NCG + g) * a.nchunks) * 4096 + wv * 1024 + l * 16) : OOB;

    auto issue = [&](int ch) {
        char* st = smem + (ch % S) * STAGE;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool cvalid = (ch * CK + xslot[q] * EPS) < a.cin;      // channel slots past cin read as zero
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)(st + q * 4096 + wv * 1024), 16, cvalid ? xoff[q] : OOB,
                                                     ch * 64, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < NCG; ++g)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(st + X_BYTES + g * 4096 + wv * 1024), 16, woff[g], ch * 4096,
                                                     0, 0);
    };

    f32x16 acc[R][NT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][t][i] = 0.f;
    const bool wave_active = (y0 + wr * R) < a.h;
    if (a.lin_form >= 0) lin_seed_bias<NT>(a, acc, (cb * 2 + wc) * NT * 32, lh);
    const int nch = a.nchunks;
    for (int c = 0; c < P && c < nch; ++c) issue(c);
    for (int ch = 0; ch < nch; ++ch) {
        // chunk ch has landed once at most min(P - 1, chunks after ch) newer chunks of this wave are outstanding
        const int newer = nch - 1 - ch;
        lin_wait_chunks_left<NP>(newer < P - 1 ? newer : P - 1);
        __syncthreads();                                     // everybody's pieces of chunk ch; every wave has finished reading stage (ch - 1) % S
        if (ch + P < nch) issue(ch + P);                     // into stage (ch + P) % S == (ch - 1) % S
        const char* st = smem + (ch % S) * STAGE;
        if (wave_active) conv_chunk_mfma<CT, NT, R, 1>(st, st + X_BYTES + wc * (NCG * 32 * 64), acc, wr, lr, lh);
    }
    if (a.lin_form >= 0) lin_epilogue_fast<CT, NT>(a, acc, smem, nb, y0 + wr * R, (cb * 2 + wc) * NT * 32, wv, l, wave_active);
    else conv_epilogue<NT, R>(a, acc, smem, nb, y0, 0, cb * 2 + wc, wr, l, wave_active, wv);
#endif
}

template <int CT, int NCG, int S>
int launch_linear_dma(hipStream_t st, const ConvArgs& a, int nblk) {
    constexpr size_t stage = 128 * 64 + (size_t)NCG * 4096;
    constexpr size_t strips = (size_t)4 * 32 * (NCG * 32 * 4 + 16);
    constexpr size_t lds = S * stage > strips ? S * stage : strips;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)linear_dma_kernel<CT, NCG, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int total = nblk * a.nty * a.n;
    const unsigned grid = (unsigned)(((total + 7) / 8) * 8);
    tr::tune().last_conv_family = 5;
    hipLaunchKernelGGL((linear_dma_kernel<CT, NCG, S>), dim3(grid), dim3(NTHREADS), lds, st, a, nblk);
    return tr::launch_status("tr_conv3x3(linear, LDS-DMA)");
}

template <int CT, int NCG>
int launch_linear(hipStream_t st, const ConvArgs& a, int nblk) {
    if constexpr (CT == TR_BF16 || CT == TR_F16) {
        // LDS-DMA variant: 16-bit operands, buffer addressing (one sample's token rows and the slab below 2 GiB), 16-byte sliceable input view
        const bool fits = (int64_t)a.h * 32 * a.in.cstride * 2 < (int64_t)0x7fff0000 && (int64_t)a.ncg * a.nchunks * 4096 < (int64_t)0x7fff0000;
        // four ring stages, two workgroups per CU.  (Measured and dropped: two stages at three workgroups per CU -- one round of workgroups instead
        // of 1.5 -- spills the side-input registers of the residual form under the 170-VGPR cap: 192 -> 180 + residual 21.6 -> 51.0 us.)
        if (tr::tune().linear_impl != 1 && fits && a.nchunks >= 2) return launch_linear_dma<CT, NCG, 4>(st, a, nblk);
    }
    constexpr size_t stage = 128 * 64 + (size_t)NCG * 4096;
    constexpr size_t strips = (size_t)4 * 32 * (NCG * 32 * 4 + 16);
    constexpr size_t lds = stage > strips ? stage : strips;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)linear_kernel<CT, NCG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int total = nblk * a.nty * a.n;
    const unsigned grid = (unsigned)(((total + 7) / 8) * 8);
    tr::tune().last_conv_family = 3;
    hipLaunchKernelGGL((linear_kernel<CT, NCG>), dim3(grid), dim3(NTHREADS), lds, st, a, nblk);
    return tr::launch_status("tr_conv3x3(linear)");
}

template <int CT, int NT, int R>
int launch_dma(hipStream_t st, const ConvArgs& a, int ncg) {
    constexpr int halo_it = ((4 * R + 2) * HALO_W * 4 + NTHREADS - 1) / NTHREADS;
    constexpr int w_it = (9 * NT * 32 * 4 + NTHREADS - 1) / NTHREADS;
    constexpr size_t lds = 2 * (size_t)(halo_it + w_it) * NTHREADS * 16;
    static_assert(lds <= 160 * 1024, "two LDS stages must fit the 160 KiB of a CU");
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv3x3_kernel_dma<CT, NT, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid((unsigned)(a.ntx * a.nty * a.n), (unsigned)ncg, 1);
    tr::tune().last_conv_family = 2;
    hipLaunchKernelGGL((conv3x3_kernel_dma<CT, NT, R>), grid, dim3(NTHREADS), lds, st, a);
    return tr::launch_status("tr_conv3x3");
}

bool view_ok(const tr_view& v, int c_needed) {
    if (!v.ptr) return true;
    return (v.dtype == TR_F32 || v.dtype == TR_BF16 || v.dtype == TR_F16) && v.cstride > 0 && v.coff >= 0 && v.coff + c_needed <= v.cstride &&
           v.c >= c_needed;
}

}  // namespace

extern "C" int tr_conv3x3(void* stream, const tr_conv3x3_desc* d) {
    TR_REQUIRE(d != nullptr, TR_EINVAL, "tr_conv3x3: null descriptor");
    TR_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, TR_EINVAL,
               "tr_conv3x3: bad dims n=%d h=%d w=%d cin=%d cout=%d", d->n, d->h, d->w, d->cin, d->cout);
    TR_REQUIRE(d->compute >= TR_F32 && d->compute <= TR_BF16X3, TR_EINVAL, "tr_conv3x3: bad compute dtype %d", d->compute);
    TR_REQUIRE(d->compute != TR_BF16X3 || d->ksize == 3, TR_EUNSUPPORTED, "tr_conv3x3: split-bf16 (fast_matmul) mode is built for 3x3 only");
    TR_REQUIRE(d->in.ptr && d->out.ptr && d->wpacked, TR_EINVAL, "tr_conv3x3: in/out/wpacked must be set");
    TR_REQUIRE(d->in.dtype == (d->compute == TR_BF16X3 ? TR_F32 : d->compute), TR_EUNSUPPORTED, "tr_conv3x3: input dtype %d does not fit compute dtype %d",
               d->in.dtype, d->compute);
    const int eps = (d->compute == TR_F32 || d->compute == TR_BF16X3) ? 4 : 8;
    TR_REQUIRE(d->cin % eps == 0, TR_EALIGN, "tr_conv3x3: cin=%d must be a multiple of %d (pad the view)", d->cin, eps);
    TR_REQUIRE(view_ok(d->in, d->cin), TR_EINVAL, "tr_conv3x3: input view does not hold cin=%d channels", d->cin);
    TR_REQUIRE(tr::view_vec16B_ok(d->in), TR_EALIGN, "tr_conv3x3: input view is not 16-byte sliceable");
    TR_REQUIRE(view_ok(d->out, d->cout) && view_ok(d->out2, d->cout) && view_ok(d->res1, d->cout) &&
                   view_ok(d->res2, d->cout) && view_ok(d->mask, d->cout),
               TR_EINVAL, "tr_conv3x3: an epilogue view does not hold cout=%d channels", d->cout);
    TR_REQUIRE(d->in_mode != TR_IN_NEAREST2X || (d->h % 2 == 0 && d->w % 2 == 0), TR_EINVAL,
               "tr_conv3x3: nearest-2x input needs even output dims");
    TR_REQUIRE(d->in_mode != TR_IN_FULL || (d->h > 2 && d->w > 2 && d->pad_mode == TR_PAD_ZEROS), TR_EINVAL,
               "tr_conv3x3: full mode needs output dims > 2 and zero padding");
    TR_REQUIRE(d->in_mode != TR_IN_VALID || d->pad_mode == TR_PAD_ZEROS, TR_EINVAL, "tr_conv3x3: valid mode has no padding to replicate");
    TR_REQUIRE(d->in_mode >= TR_IN_DIRECT && d->in_mode <= TR_IN_VALID, TR_EINVAL, "tr_conv3x3: bad in_mode %d", d->in_mode);
    TR_REQUIRE(((uintptr_t)d->wpacked) % 16 == 0, TR_EALIGN, "tr_conv3x3: wpacked must be 16-byte aligned");

    ConvArgs a;
    a.n = d->n; a.h = d->h; a.w = d->w;
    if (d->ksize == 1 && d->in_mode == TR_IN_DIRECT && d->w != 32 && ((int64_t)d->h * d->w) % 32 == 0) {
        // a direct 1x1 conv does not care how the pixels of an image are folded into rows: relabel them as a 32-wide token grid so
        // that the GEMM-tiled kernel applies (rows of an image stay inside the image: per-sample scales still index by n)
        a.h = (int)((int64_t)d->h * d->w / 32);
        a.w = 32;
    }
    a.up = d->in_mode == TR_IN_NEAREST2X ? 1 : 0;
    a.shift = d->in_mode == TR_IN_FULL ? 1 : (d->in_mode == TR_IN_VALID ? -1 : 0);   // input origin relative to the output origin
    a.ih = a.h - 2 * a.shift; a.iw = a.w - 2 * a.shift;
    a.hin = a.ih >> a.up; a.win = a.iw >> a.up;
    a.cin = d->cin; a.cout = d->cout;
    a.replicate = d->pad_mode == TR_PAD_REPLICATE;
    a.act = d->act; a.slope = d->slope;
    a.scale1 = d->scale1; a.scale2 = d->scale2; a.rs1 = d->res1_scale; a.rs2 = d->res2_scale; a.mask_slope = d->mask_slope; a.out2_scale = d->out2_scale;
    a.in = d->in; a.out = d->out; a.out2 = d->out2; a.res1 = d->res1; a.res2 = d->res2; a.mask = d->mask;
    a.wpacked = (const char*)d->wpacked; a.bias = d->bias;
    a.mask_mode = d->mask_mode; a.sample_scale = d->sample_scale;
    const tr::PackGeom g = tr::pack_geom(d->cin, d->cout, d->compute);   // geometry is per tap; ksize only scales the slab
    a.nchunks = g.nchunks;
    a.ncg = g.ncg;
    a.vec_out = tr::view_vec4_ok(d->out); a.vec_out2 = tr::view_vec4_ok(d->out2);
    a.vec_res1 = tr::view_vec4_ok(d->res1); a.vec_res2 = tr::view_vec4_ok(d->res2); a.vec_mask = tr::view_vec4_ok(d->mask);
    hipStream_t st = (hipStream_t)stream;
    static const int dbg = getenv("TRAINNER_CONV_DBG") ? atoi(getenv("TRAINNER_CONV_DBG")) : 0;
    a.dbg = dbg;
#ifdef TR_RING_DEV
    a.stamps = (long long*)tr::tune().stamps;   // development builds only: tr_tune_set("stamps", device address)
#else
    a.stamps = nullptr;
#endif
    tr::tune().last_conv_instance = 0;
    a.ntx = (a.w + TW - 1) / TW;
    // Staging variant and tile height.  Measured on MI355X (profiles/): the register-staged kernel at two workgroups per
    // CU beats the LDS-DMA kernel at one (each LDS-DMA piece costs the issuing wave 60-100 cycles and the per-CU DMA rate
    // does not cover a 76 KB stage inside one chunk of MFMAs), so it is the default; TRAINNER_CONV_STAGING=dma selects B.
    {   // the persistent LDS-DMA ring kernel (conv_ring.hip) takes the large 16-bit 3x3 launches of the RRDB trunk / upsampling tail
        const int rc = tr::conv3x3_ring(st, a, d);
        if (rc != -1) return rc;
    }
    static const bool use_dma = getenv("TRAINNER_CONV_STAGING") && !strcmp(getenv("TRAINNER_CONV_STAGING"), "dma");
    static const int force_r = getenv("TRAINNER_CONV_R") ? atoi(getenv("TRAINNER_CONV_R")) : 0;
    const bool roomy = a.h >= 16 && (int64_t)d->n * ((a.h + 15) / 16) * a.ntx * g.ncg >= 512;
    TR_REQUIRE(d->ksize == 3 || d->ksize == 1, TR_EUNSUPPORTED, "tr_conv3x3: ksize must be 3 or 1 (got %d)", d->ksize);
    TR_REQUIRE(d->ksize == 3 || (d->in_mode != TR_IN_NEAREST2X && d->pad_mode == TR_PAD_ZEROS), TR_EINVAL, "tr_conv3x3: 1x1 mode takes direct / full / valid input");
    if (use_dma && d->ksize == 3 && d->in_mode != TR_IN_VALID && d->compute != TR_F16 && d->compute != TR_BF16X3) {
        TR_REQUIRE((int64_t)a.hin * a.win * d->in.cstride * tr::esize(d->in.dtype) < (int64_t)0x7fffff00, TR_EUNSUPPORTED,
                   "tr_conv3x3: one image of the input view must stay below 2 GiB (buffer addressing)");
        const bool big = force_r ? force_r == 4 : roomy;
        a.nty = big ? (a.h + 15) / 16 : (a.h + 7) / 8;
        if (big) {
            if (d->compute == TR_BF16) return g.cgw == 64 ? launch_dma<TR_BF16, 2, 4>(st, a, g.ncg) : launch_dma<TR_BF16, 1, 4>(st, a, g.ncg);
            return g.cgw == 64 ? launch_dma<TR_F32, 2, 4>(st, a, g.ncg) : launch_dma<TR_F32, 1, 4>(st, a, g.ncg);
        }
        if (d->compute == TR_BF16) return g.cgw == 64 ? launch_dma<TR_BF16, 2, 2>(st, a, g.ncg) : launch_dma<TR_BF16, 1, 2>(st, a, g.ncg);
        return g.cgw == 64 ? launch_dma<TR_F32, 2, 2>(st, a, g.ncg) : launch_dma<TR_F32, 1, 2>(st, a, g.ncg);
    }
    // register-staged: 32-wide output groups take 16-row tiles (twice the MFMAs per staged weight slab); 64-wide groups
    // keep 8-row tiles, whose accumulators + prefetch registers already fill the 256-VGPR budget of two workgroups per CU.
    const bool big = g.cgw == 32 && (force_r ? force_r == 4 : (roomy && !((d->compute == TR_BF16 || d->compute == TR_F16) && d->ksize == 3 && d->cin <= 96)));
    const int ct = d->compute;
#define TR_BY_CT(FN, ...) (ct == TR_BF16 ? FN<TR_BF16, __VA_ARGS__> : ct == TR_F16 ? FN<TR_F16, __VA_ARGS__> : FN<TR_F32, __VA_ARGS__>)
#define TR_BY_CT3(FN, ...) (ct == TR_BF16X3 ? FN<TR_BF16X3, __VA_ARGS__> : TR_BY_CT(FN, __VA_ARGS__))
    static const bool no_linear = getenv("TRAINNER_LINEAR") && !strcmp(getenv("TRAINNER_LINEAR"), "conv");
    a.lin_form = -1;
    if (d->ksize == 1 && a.w == 32 && g.cgw == 64 && d->in_mode == TR_IN_DIRECT && !no_linear) {
        // token grid (Linear layers): GEMM tiling, 128 tokens x up to 192 output channels per workgroup
        // straight-line epilogue forms (lin_epilogue_fast): 16-bit compute, whole 4-channel pieces, vector-accessible views
        if ((ct == TR_BF16 || ct == TR_F16) && d->cout % 4 == 0 && a.vec_out && a.vec_out2 && a.vec_res1 && a.vec_mask && !d->res2.ptr &&
            (!d->bias || ((uintptr_t)d->bias) % 16 == 0) && tr::tune().linear_epilogue != 1) {
            const bool side = d->res1.ptr || d->mask.ptr;
            if (!side && d->act == TR_ACT_NONE && !d->out2.ptr) a.lin_form = d->out.dtype == ct ? LF_PLAIN_CD : (d->out.dtype == TR_F32 ? LF_PLAIN_F32 : -1);
            else if (d->res1.ptr && d->res1.dtype == TR_F32 && !d->mask.ptr && d->act == TR_ACT_NONE && d->out.dtype == TR_F32 && (!d->out2.ptr || d->out2.dtype == ct))
                a.lin_form = LF_RES_F32;
            else if (!side && d->act == TR_ACT_GELU && d->out.dtype == ct && d->out2.ptr && d->out2.dtype == ct) a.lin_form = LF_GELU_CD;
            else if (d->mask.ptr && d->mask_mode == 1 && d->mask.dtype == ct && !d->res1.ptr && d->act == TR_ACT_NONE && d->out.dtype == ct && !d->out2.ptr)
                a.lin_form = LF_MASKG_CD;
        }
        const int ncgb = g.ncg == 4 ? 2 : (g.ncg < 3 ? g.ncg : 3);
        const int nblk = (g.ncg + ncgb - 1) / ncgb;
        a.nty = (a.h + 3) / 4;
        if (ncgb == 3) return TR_BY_CT(launch_linear, 3)(st, a, nblk);
        if (ncgb == 2) return TR_BY_CT(launch_linear, 2)(st, a, nblk);
        return TR_BY_CT(launch_linear, 1)(st, a, nblk);
    }
    if (d->ksize == 1) {
        if (big) {
            a.nty = (a.h + 15) / 16;
            return TR_BY_CT(launch_reg, 1, 4, 1)(st, a, g.ncg);
        }
        a.nty = (a.h + 7) / 8;
        return g.cgw == 64 ? TR_BY_CT(launch_reg, 2, 2, 1)(st, a, g.ncg) : TR_BY_CT(launch_reg, 1, 2, 1)(st, a, g.ncg);
    }
    if (big) {
        a.nty = (a.h + 15) / 16;
        return TR_BY_CT3(launch_reg, 1, 4, 3)(st, a, g.ncg);
    }
    a.nty = (a.h + 7) / 8;
    return g.cgw == 64 ? TR_BY_CT3(launch_reg, 2, 2, 3)(st, a, g.ncg) : TR_BY_CT3(launch_reg, 1, 2, 3)(st, a, g.ncg);
#undef TR_BY_CT3
#undef TR_BY_CT
}
