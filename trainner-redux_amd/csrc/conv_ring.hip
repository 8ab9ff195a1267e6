// 3x3 / stride 1 / pad 1 convolution on NHWC channel-slice views, 16-bit operands: persistent LDS-DMA ring kernel.
//
// Contract: include/trainner_hip.h (tr_conv3x3); this is the kernel the RRDB trunk and the upsampling tail of ESRGAN run on
// (reference arithmetic: traiNNer/archs/scunet_aaf6aa_arch.py:310-384, see conv3x3.hip).  Same packed weight slabs and the same MFMA
// orientation D[cout][pixel] as conv3x3_kernel_reg; the bias seeds the accumulators here and is added in the epilogue there, so the two
// kernels differ in the last fp32 place (measured 2.4e-7 rel-L2 on fp32 outputs, 99.9 % of 16-bit outputs identical): each is held to
// F.conv2d separately (tests/test_conv_ring_gpu.py, tests/test_kernels_gpu.py).
//
// Why a second kernel ("ring kernel"): the register-staged kernel moves every operand global -> VGPR -> LDS
// with two workgroup barriers per 32-channel chunk and transposes its accumulators through LDS before storing; on the RRDB shapes
// (K = 9 x 64..192, 32 or 64 output channels) its phases add up instead of overlapping and it sustains 20-30 % of the MFMA peak at
// 2.2-2.5 TB/s.  Here:
//   * one workgroup per CU, persistent over tiles (the XCD's tiles are contiguous, so vertical halo overlap hits its L2);
//   * waves are specialised: NCW consumer waves issue nothing but ds_read + MFMA in the K loop; 4 loader waves move the halo
//     (and the weight chunk) of chunk j + D straight into LDS with buffer_load ... lds (no VGPR round trip, no ds_write), D chunks
//     ahead, across tile boundaries: the first chunks of the next tile land while the consumers are in the epilogue;
//   * ONE raw s_barrier per chunk: the loaders arrive after a counted s_waitcnt vmcnt (chunk j - D + 1 has landed), the consumers
//     after their last ds_read of the chunk before (its stage may be refilled);
//   * with WRES the weights of all K chunks of the output group stay resident in LDS for the lifetime of the workgroup
//     (36-110 KB for the RRDB convs): the ring then carries halo bytes only;
//   * the epilogue stores straight from the accumulators: v_permlane32_swap pairs turn the MFMA layout (one pixel, four scattered
//     4-channel quads per lane) into 16 consecutive channels per lane, i.e. 2 (bf16) or 4 (fp32) dwordx4 stores per row and lane --
//     no LDS transposition, no extra barrier.
#include "common.h"
#include "conv_args.h"
#include <stdlib.h>
#include <string.h>
#include <utility>

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2c;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2c;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2c;

#ifndef TR_RING_STORE_AUX
#define TR_RING_STORE_AUX 0    /* cache policy bits of the epilogue stores (A/B: 2 = nt) */
#endif
#ifndef TR_RING_SPREAD
#define TR_RING_SPREAD 1     /* A/B: 0 keeps the epilogue stages as blocks between MFMA steps */
#endif

constexpr int TW = 32;
constexpr int HALO_W = TW + 2;
constexpr unsigned OOB = 0x7fffff00u;        // beyond any num_records: the buffer load returns 0
constexpr int LDS_MAX = 160 * 1024;

template <int NT, int R, int NCW, int NLW, int D, bool WRES, bool PIPE = false>
struct Cfg {
    static constexpr int TH = R * NCW;
    static constexpr int HALO_PIX = (TH + 2) * HALO_W;
    static constexpr int HALO_ITEMS = HALO_PIX * 4;                              // 16-byte slots
    static constexpr int HALO_BYTES = HALO_ITEMS * 16;
    static constexpr int HALO_IT = (HALO_ITEMS + 64 * NLW - 1) / (64 * NLW);     // DMA pieces per loader wave and chunk
    static constexpr int CGW = NT * 32;
    static constexpr int W_ITEMS = 9 * CGW * 4;
    static constexpr int W_BYTES = W_ITEMS * 16;
    static constexpr int W_IT = (W_ITEMS + 64 * NLW - 1) / (64 * NLW);
    static constexpr int S = D + 1;                                              // ring stages
    static constexpr int STAGE_BYTES = HALO_BYTES + (WRES ? 0 : W_BYTES);
    static constexpr int NP = HALO_IT + (WRES ? 0 : W_IT);                       // vmcnt events per loader wave and chunk
    static constexpr int NTHREADS = (NCW + NLW) * 64;
    static constexpr int NSTRIP = (NCW * 2 * (32 * (32 * 4 + 16)) <= STAGE_BYTES) ? 2 : 1;   // epilogue strips per consumer wave inside one (consumed) stage
    static_assert(PIPE || NCW * NSTRIP * (32 * (32 * 4 + 16)) <= STAGE_BYTES, "the epilogue strips alias one ring stage");
    static constexpr int STRIPS_BYTES = PIPE ? NCW * (32 * (32 * 4 + 16)) : 0;               // pipelined kernels: one strip per consumer wave, LDS of its own
    static constexpr int MASK_BYTES = PIPE ? R * NCW * 32 * 64 : 0;                          // pipelined SIDE 1: one tile of 16-bit mask values [row][32 pixels][32 channels]
    static constexpr int U = R * NT;                                                         // epilogue units per wave and tile
    static constexpr int EC = (U + 1 + 4) / 5;                                               // chunks of the NEXT tile the U + 1 epilogue stages are spread over (<= 5 hook slots per chunk)
    static_assert(HALO_ITEMS % 4 == 0 && HALO_ITEMS >= 64 && W_ITEMS % 64 == 0, "piece geometry");
};

__device__ __forceinline__ unsigned pack2(float a, float b, int dtype) {   // element 0 in the low half; RNE like tr::f32_to_h
    const f32x2c v = {a, b};
    if (dtype == TR_BF16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2c));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2c));
}

// ---- 9 taps x one 32-channel chunk of MFMAs out of a staged halo image and weight chunk; software pipelined one step ahead ---------------
// LDS addressing: the halo image holds pixel (py, px) at (py * HALO_W + px) * 64 with its four 16-byte slots XOR-swizzled by (px >> 2) & 3 -- by the
// column only, so the address of an operand fragment is  [stage + first row of the wave] + lane term (kx, k-half) + (r + ky) * HALO_W * 64:
// six lane terms for the nine taps, everything else an immediate offset (a swizzle by the linear pixel index cost an address computation per
// fragment and ~40 VGPRs).  A wave's ds_read_b128 covers 32 consecutive pixels of one row: conflict free for any kx (the four lanes of a
// 16-lane group that share px mod 4 have px >> 2 = m, m + 3, m + 5, m + 6: distinct mod 4).  The weight rows keep pack.hip's swizzle, which
// depends on the lane only (tap * CGW + t * 32 is a multiple of 32).
struct FragAddr {
    int hx[3][2];     // [kx][k-half]: (lr + kx) * 64 + (((half * 2 + lh) ^ (((lr + kx) >> 2) & 3)) << 4)
    int wx[2];        // [k-half]:     lr * 64 + (((half * 2 + lh) ^ ((lr >> 2) & 3)) << 4)
};
__device__ __forceinline__ FragAddr make_frag_addr(int lr, int lh) {
    FragAddr f;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) f.hx[kx][hf] = (lr + kx) * 64 + (((hf * 2 + lh) ^ (((lr + kx) >> 2) & 3)) << 4);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) f.wx[hf] = lr * 64 + (((hf * 2 + lh) ^ ((lr >> 2) & 3)) << 4);
    return f;
}

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// The same chunk with NH hook slots between its MFMA steps (slot i behind step 1 + i * HS): `hook(integral_constant<HB + i>)` carries one
// stage of the PREVIOUS tile's epilogue (pipelined kernels), whose VALU / LDS / store instructions then issue in the shadow of the MFMAs.
template <int CT, int NT, int R, int HB, int NH, typename HookF>
__device__ __forceinline__ void ring_chunk_mfma_hooked(const char* hl, const char* wl, f32x16 (&acc)[R][NT], const FragAddr& fa, HookF&& hook) {
    constexpr int CGW = NT * 32;
    constexpr int NSTEP = 18;
    constexpr int HS = NH > 1 ? 16 / (NH - 1) : 16;
    bf16x8 wa[2][NT], xb[2][R];
    // hl: stage + (first halo row of the wave) * HALO_W * 64
    const char* hb[3][2];
    const char* wb[2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) hb[kx][hf] = hl + fa.hx[kx][hf];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) wb[hf] = wl + fa.wx[hf];
    auto frags = [&](auto bufc, auto stepc) {
        constexpr int BUF = decltype(bufc)::value, STEP = decltype(stepc)::value;
        constexpr int tap_ = STEP >> 1, ky_ = tap_ / 3, kx_ = tap_ % 3, hf_ = STEP & 1;
#pragma unroll
        for (int t = 0; t < NT; ++t) wa[BUF][t] = *(const bf16x8*)(wb[hf_] + (tap_ * CGW + t * 32) * 64);
#pragma unroll
        for (int r = 0; r < R; ++r) xb[BUF][r] = *(const bf16x8*)(hb[kx_][hf_] + (r + ky_) * (HALO_W * 64));
    };
    frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    auto step = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int cur = s & 1;
        if constexpr (s + 1 < NSTEP) frags(std::integral_constant<int, cur ^ 1>{}, std::integral_constant<int, s + 1>{});
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (CT == TR_BF16) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[cur][t], xb[cur][r], acc[r][t], 0, 0, 0);
                else acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa[cur][t]), __builtin_bit_cast(f16x8, xb[cur][r]), acc[r][t], 0, 0, 0);
            }
        if constexpr (NH > 0 && s >= 1 && (s - 1) % HS == 0 && (s - 1) / HS < NH) hook(std::integral_constant<int, HB + (s - 1) / HS>{});
        if constexpr (NH == 0 || !TR_RING_SPREAD) __builtin_amdgcn_sched_barrier(0);      // plain chunks: keep the fragment prefetch exactly one step ahead
    };
    static_for<NSTEP>(step);
    if constexpr (NH > 0 && TR_RING_SPREAD) {
        // Hooked chunks are ONE scheduling region: the epilogue stages (straight-line VALU / LDS / store code) are dealt out behind the MFMAs,
        // a few instructions per MFMA, instead of sitting between two steps as blocks of 70-140 instructions during which the matrix pipe idles
        // (a consumer wave has its SIMD to itself: whatever issues in the shadow of an MFMA -- ~5 instructions per 32-cycle MFMA -- is free).
        auto grp = [&](auto) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // DS read (operand fragments, strip read-back)
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);     // VALU
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // DS write (strip)
            __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);     // VMEM write
        };
        static_for<NSTEP * R * NT>(grp);
    }
}

template <int CT, int NT, int R>
__device__ __forceinline__ void ring_chunk_mfma(const char* hl, const char* wl, f32x16 (&acc)[R][NT], const FragAddr& fa) {
    auto nohook = [](auto) {};
    ring_chunk_mfma_hooked<CT, NT, R, 0, 0>(hl, wl, acc, fa, nohook);
}

// ---- epilogue: one (row, 32-channel tile) unit of a consumer wave -----------------------------------------------------------------------
// acc of lane (lr, lh): pixel lr, channels q*8 + lh*4 + j (register q*4 + j).  Storing from that layout puts every lane of a store instruction
// on its own pixel record; measured (in-kernel stamps, 64 -> 32 at 256^2): 8 such dwordx4 stores per wave cost 6.7k cycles per tile -- more than
// the tile's MFMAs -- and the same shape of side-input loads as much again: the memory pipeline takes pixel-per-lane accesses at a fraction of
// the rate of lane-contiguous ones.  So a unit crosses LDS once (fp32, [pixel][32 channels], rows padded by 16 B: conflict-free 16-byte
// writes and reads) and comes back as lane = (pixel l >> 2 (+16), 8-channel piece l & 3): four neighbouring lanes cover 64 (16-bit) or 128
// (fp32) contiguous bytes of one pixel record, in the stores and in the residual / mask loads alike.  The strips live in the ring stage the
// tile's last chunk has just been consumed from (free until the loaders pass the next chunk barrier): no LDS of their own.
// The bias already sits in the accumulators (they are seeded with it).  Side inputs come in the two shapes of the hot path:
//   SIDE 1: a 16-bit activation-derivative mask (input-gradient convs of the dense block);   SIDE 2: one or two fp32 residuals (conv5 / trunk);
// their raw pieces for unit u + 1 are requested before unit u is processed (the first unit's before the MFMAs of the tile's last chunk).
// FAST (host-checked): 16-bit views have the compute dtype CT, act none / LeakyReLU with slope in [0, 1], SIDE 0 / 1 store 16-bit `out` only,
// SIDE 2 stores fp32 `out` and optionally 16-bit `out2`; otherwise the run-time-typed body.
constexpr int STRIP_ROW = 32 * 4 + 16;
constexpr int STRIP_BYTES = 32 * STRIP_ROW;

template <int SIDE>
struct SideRaw {
    uint4 q[SIDE == 0 ? 1 : (SIDE == 1 ? 2 : 8)];   // [pixel half][pieces]: SIDE 1: 1 x 16 B ; SIDE 2: res1 2 x 16 B, res2 2 x 16 B
};

template <int SIDE>
__device__ __forceinline__ void side_issue(const ConvArgs& a, SideRaw<SIDE>& raw, int64_t rowpix, int x0, int cobase, int l) {
    const int co = min(cobase + (l & 3) * 8, a.cout - 8);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int64_t pix = rowpix + min(x0 + (l >> 2) + hf * 16, a.w - 1);      // clamped: loads stay inside the image, stores are masked
        if constexpr (SIDE == 1) {
            if (a.mask.dtype == TR_F32) {   // generic body only (FAST masks are 16-bit): keep the sign pattern of 8 fp32 values in 16-bit form
                const float4* p = (const float4*)((const float*)a.mask.ptr + pix * a.mask.cstride + a.mask.coff + co);
                const float4 m0 = p[0], m1 = p[1];
                raw.q[hf] = make_uint4((m0.x > 0.f ? 0x3f80u : 0u) | (m0.y > 0.f ? 0x3f800000u : 0u), (m0.z > 0.f ? 0x3f80u : 0u) | (m0.w > 0.f ? 0x3f800000u : 0u),
                                       (m1.x > 0.f ? 0x3f80u : 0u) | (m1.y > 0.f ? 0x3f800000u : 0u), (m1.z > 0.f ? 0x3f80u : 0u) | (m1.w > 0.f ? 0x3f800000u : 0u));
            } else {
                raw.q[hf] = *(const uint4*)((const unsigned short*)a.mask.ptr + pix * a.mask.cstride + a.mask.coff + co);
            }
        } else if constexpr (SIDE == 2) {
            const uint4* p1 = (const uint4*)((const float*)a.res1.ptr + pix * a.res1.cstride + a.res1.coff + co);
            raw.q[hf * 4] = p1[0];
            raw.q[hf * 4 + 1] = p1[1];
            if (a.res2.ptr) {
                const uint4* p2 = (const uint4*)((const float*)a.res2.ptr + pix * a.res2.cstride + a.res2.coff + co);
                raw.q[hf * 4 + 2] = p2[0];
                raw.q[hf * 4 + 3] = p2[1];
            }
        }
    }
}

template <int CT>
__device__ __forceinline__ unsigned pack2c(float a, float b) {
    const f32x2c v = {a, b};
    if constexpr (CT == TR_BF16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2c));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2c));
}

// 8 consecutive channels -> a view of any dtype (fp32: 2 x 16 B, 16-bit: 16 B): the generic epilogue body
__device__ __forceinline__ void store8(const tr_view& v, int64_t pix, int co, const float (&x)[8], float sc) {
    const int64_t idx = pix * v.cstride + v.coff + co;
    if (v.dtype == TR_F32) {
        float4* p = (float4*)((float*)v.ptr + idx);
        p[0] = make_float4(x[0] * sc, x[1] * sc, x[2] * sc, x[3] * sc);
        p[1] = make_float4(x[4] * sc, x[5] * sc, x[6] * sc, x[7] * sc);
    } else {
        uint4* p = (uint4*)((unsigned short*)v.ptr + idx);
        *p = make_uint4(pack2(x[0] * sc, x[1] * sc, v.dtype), pack2(x[2] * sc, x[3] * sc, v.dtype), pack2(x[4] * sc, x[5] * sc, v.dtype),
                        pack2(x[6] * sc, x[7] * sc, v.dtype));
    }
}

__device__ __forceinline__ void epi_write(const f32x16& acc, char* strip, int l) {
    const int lr = l & 31, lh = l >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *(float4*)(strip + lr * STRIP_ROW + (q * 8 + lh * 4) * 4) = make_float4(acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]);
}
// no wait between the writes above and these reads: one wave's DS instructions execute in issue order, and the compiler keeps the (possibly
// aliasing) write -> read order; only the read DATA is waited for
__device__ __forceinline__ void epi_read(const char* strip, int l, float4 (&rd)[2][2]) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int p = (l >> 2) + hf * 16;
        rd[hf][0] = *(const float4*)(strip + p * STRIP_ROW + (l & 3) * 32);
        rd[hf][1] = *(const float4*)(strip + p * STRIP_ROW + (l & 3) * 32 + 16);
    }
}

template <int CT, int SIDE, bool FAST>
__device__ __forceinline__ void epi_finish(const ConvArgs& a, const float4 (&rd)[2][2], const SideRaw<SIDE>& raw, int64_t rowpix, bool rowok, int x0, int cobase, int l) {
    const int co = cobase + (l & 3) * 8;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int ox = x0 + (l >> 2) + hf * 16;
        float v[8] = {rd[hf][0].x, rd[hf][0].y, rd[hf][0].z, rd[hf][0].w, rd[hf][1].x, rd[hf][1].y, rd[hf][1].z, rd[hf][1].w};
        const bool ok = rowok && ox < a.w && co < a.cout && !(a.dbg & 4);      // dbg 4 (TRAINNER_CONV_DBG, timing experiments): no stores
        const int64_t pix = rowpix + min(ox, a.w - 1);
        if (a.act == TR_ACT_LRELU) {
            if constexpr (FAST) {   // slope in [0, 1] (host-checked): LeakyReLU(v) = max(v, v * slope) = med3(v, v * slope, +inf): one packed multiply per pair + one med3
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_fmed3f(v[i], v[i] * a.slope, __builtin_inff());
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] *= v[i] > 0.f ? 1.f : a.slope;
            }
        } else if (!FAST && a.act == TR_ACT_RELU) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        if constexpr (SIDE == 2) {
            const uint4 qa = raw.q[hf * 4], qb = raw.q[hf * 4 + 1];
            const float rr[8] = {__uint_as_float(qa.x), __uint_as_float(qa.y), __uint_as_float(qa.z), __uint_as_float(qa.w),
                                 __uint_as_float(qb.x), __uint_as_float(qb.y), __uint_as_float(qb.z), __uint_as_float(qb.w)};
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = v[i] * a.scale1 + a.rs1 * rr[i];
            if (a.res2.ptr) {
                const uint4 qc = raw.q[hf * 4 + 2], qd = raw.q[hf * 4 + 3];
                const float r2[8] = {__uint_as_float(qc.x), __uint_as_float(qc.y), __uint_as_float(qc.z), __uint_as_float(qc.w),
                                     __uint_as_float(qd.x), __uint_as_float(qd.y), __uint_as_float(qd.z), __uint_as_float(qd.w)};
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] * a.scale2 + a.rs2 * r2[i];
            }
        }
        if constexpr (SIDE == 1) {
            const uint4 q = raw.q[hf];
            const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // m > 0 on the raw 16-bit patterns (bf16 and fp16 alike; activations are finite): sign clear and magnitude non-zero = a positive integer
                v[2 * k] *= (int)(u[k] << 16) > 0 ? 1.f : a.mask_slope;
                v[2 * k + 1] *= (int)u[k] > 0xffff ? 1.f : a.mask_slope;
            }
        }
        if constexpr (FAST) {
            if constexpr (SIDE == 2) {
                if (ok) {
                    float4* po = (float4*)((float*)a.out.ptr + pix * a.out.cstride + a.out.coff + co);
                    po[0] = make_float4(v[0], v[1], v[2], v[3]);
                    po[1] = make_float4(v[4], v[5], v[6], v[7]);
                }
                if (a.out2.ptr) {
                    const float sc = a.out2_scale;
                    const uint4 pk = make_uint4(pack2c<CT>(v[0] * sc, v[1] * sc), pack2c<CT>(v[2] * sc, v[3] * sc), pack2c<CT>(v[4] * sc, v[5] * sc), pack2c<CT>(v[6] * sc, v[7] * sc));
                    if (ok) *(uint4*)((unsigned short*)a.out2.ptr + pix * a.out2.cstride + a.out2.coff + co) = pk;
                }
            } else {
                const uint4 pk = make_uint4(pack2c<CT>(v[0], v[1]), pack2c<CT>(v[2], v[3]), pack2c<CT>(v[4], v[5]), pack2c<CT>(v[6], v[7]));
                if (ok) *(uint4*)((unsigned short*)a.out.ptr + pix * a.out.cstride + a.out.coff + co) = pk;
            }
        } else if (ok) {
            store8(a.out, pix, co, v, 1.f);
            if (a.out2.ptr) store8(a.out2, pix, co, v, a.out2_scale);
        }
    }
}

// ---- fast epilogue forms: straight-line code on buffer resources ---------------------------------------------------------------------------
// The per-tile context holds image-based buffer resources and, per lane, the byte offset of its (column, 8-channel piece) inside a pixel row
// -- an out-of-range offset where the column or the channels lie outside, so stores to them are dropped and loads return zeros by the
// buffer bounds check instead of by an exec-masked branch.  A unit then adds ONE scalar offset (row, 32-channel tile) per view.  No
// branches: the stages of the previous tile's epilogue sit in one basic block with the MFMAs of the chunk they are hooked into, and the
// scheduler may spread them over the MFMA gaps (ring_chunk_mfma_hooked).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SIDE>
struct EpiTile {
    __amdgpu_buffer_rsrc_t r_out, r_out2, r_s1, r_s2;
    int c_out[2], c_out2[2], c_s1[2], c_s2[2];
    int y0, cs_out, cs_out2, cs_s1, cs_s2;      // first row of the tile; row pitches are w * cstride * esize (bytes)
    int cg_left;                                // output channels from the group's first one to cout: tile t of a 64-wide group exists iff t * 32 < cg_left
    float slope, scale2, rs2;                   // act none: slope 1 ; no res2: v * 1 + 0 * res1
};

template <int CT, int SIDE>
__device__ __forceinline__ void epi_tile_setup(const ConvArgs& a, EpiTile<SIDE>& T, int nb, int y0, int x0, int cgbase, int l) {
    constexpr int ES_OUT = SIDE == 2 ? 4 : 2;
    auto mk = [&](const tr_view& v, int es) {
        const int64_t img = (int64_t)a.h * a.w * v.cstride * es;
        return __builtin_amdgcn_make_buffer_rsrc((void*)((char*)v.ptr + (int64_t)nb * img), 0, (unsigned)img, 0x00020000);
    };
    T.r_out = mk(a.out, ES_OUT);
    T.r_out2 = (SIDE == 2 && a.out2.ptr) ? mk(a.out2, 2) : T.r_out;
    if constexpr (SIDE == 1) T.r_s1 = mk(a.mask, 2);
    if constexpr (SIDE == 2) {
        T.r_s1 = mk(a.res1, 4);
        T.r_s2 = a.res2.ptr ? mk(a.res2, 4) : T.r_s1;
    }
    const tr_view& s2v = (SIDE == 2 && a.res2.ptr) ? a.res2 : a.res1;
    T.y0 = y0;
    T.cg_left = a.cout - cgbase;
    T.cs_out = a.w * a.out.cstride * ES_OUT;
    T.cs_out2 = (SIDE == 2 && a.out2.ptr) ? a.w * a.out2.cstride * 2 : 0;
    T.cs_s1 = SIDE == 1 ? a.w * a.mask.cstride * 2 : SIDE == 2 ? a.w * a.res1.cstride * 4 : 0;
    T.cs_s2 = SIDE == 2 ? a.w * s2v.cstride * 4 : 0;
    T.slope = a.act == TR_ACT_LRELU ? a.slope : 1.f;
    T.scale2 = (SIDE == 2 && a.res2.ptr) ? a.scale2 : 1.f;
    T.rs2 = (SIDE == 2 && a.res2.ptr) ? a.rs2 : 0.f;
    const int co = cgbase + (l & 3) * 8;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int ox = x0 + (l >> 2) + hf * 16;
        const bool ok = ox < a.w && co < a.cout;
        const int oxc = min(ox, a.w - 1), coc = min(co, a.cout - 8);
        T.c_out[hf] = ok ? (ox * a.out.cstride + a.out.coff + co) * ES_OUT : (int)OOB;
        T.c_out2[hf] = (SIDE == 2 && a.out2.ptr && ok) ? (ox * a.out2.cstride + a.out2.coff + co) * 2 : (int)OOB;
        if constexpr (SIDE == 1) T.c_s1[hf] = (oxc * a.mask.cstride + a.mask.coff + coc) * 2;
        if constexpr (SIDE == 2) {
            T.c_s1[hf] = (oxc * a.res1.cstride + a.res1.coff + coc) * 4;
            T.c_s2[hf] = (oxc * s2v.cstride + s2v.coff + coc) * 4;
        }
    }
}

// side pieces of unit (row r of the wave, tile t) -> raw; rows past the image read the last row (their stores are dropped)
template <int SIDE>
__device__ __forceinline__ void side_issue_fast(const EpiTile<SIDE>& T, SideRaw<SIDE>& raw, int oyc, int t) {
    if constexpr (SIDE == 1) {
        const int so = oyc * T.cs_s1 + t * 64;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(T.r_s1, T.c_s1[hf], so, 0);
            raw.q[hf] = make_uint4(q.x, q.y, q.z, q.w);
        }
    } else if constexpr (SIDE == 2) {
        const int so1 = oyc * T.cs_s1 + t * 128, so2 = oyc * T.cs_s2 + t * 128;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(T.r_s1, T.c_s1[hf], so1, 0);
            const u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(T.r_s1, T.c_s1[hf] + 16, so1, 0);
            const u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(T.r_s2, T.c_s2[hf], so2, 0);
            const u32x4 q3 = __builtin_amdgcn_raw_buffer_load_b128(T.r_s2, T.c_s2[hf] + 16, so2, 0);
            raw.q[hf * 4] = make_uint4(q0.x, q0.y, q0.z, q0.w);
            raw.q[hf * 4 + 1] = make_uint4(q1.x, q1.y, q1.z, q1.w);
            raw.q[hf * 4 + 2] = make_uint4(q2.x, q2.y, q2.z, q2.w);
            raw.q[hf * 4 + 3] = make_uint4(q3.x, q3.y, q3.z, q3.w);
        }
    }
}

template <int CT, int SIDE>
__device__ __forceinline__ void epi_finish_fast(const ConvArgs& a, const EpiTile<SIDE>& T, const float4 (&rd)[2][2], const SideRaw<SIDE>& raw, int oy, int t) {
    const bool rowok = oy < a.h;
    const int oyc = min(oy, a.h - 1);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        float v[8] = {rd[hf][0].x, rd[hf][0].y, rd[hf][0].z, rd[hf][0].w, rd[hf][1].x, rd[hf][1].y, rd[hf][1].z, rd[hf][1].w};
        if constexpr (SIDE != 2) {   // LeakyReLU with slope in [0, 1] = max(v, v * slope); act none: slope 1.  The raw instruction: fmaxf() puts a
            // canonicalising v_max v, v, v in front of every maximum (3 instructions per element instead of 1.5 with the packed multiply)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float vs = v[i] * T.slope;
                asm("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(vs));
            }
        }
        if constexpr (SIDE == 2) {
            const uint4 qa = raw.q[hf * 4], qb = raw.q[hf * 4 + 1], qc = raw.q[hf * 4 + 2], qd = raw.q[hf * 4 + 3];
            const float r1[8] = {__uint_as_float(qa.x), __uint_as_float(qa.y), __uint_as_float(qa.z), __uint_as_float(qa.w),
                                 __uint_as_float(qb.x), __uint_as_float(qb.y), __uint_as_float(qb.z), __uint_as_float(qb.w)};
            const float r2[8] = {__uint_as_float(qc.x), __uint_as_float(qc.y), __uint_as_float(qc.z), __uint_as_float(qc.w),
                                 __uint_as_float(qd.x), __uint_as_float(qd.y), __uint_as_float(qd.z), __uint_as_float(qd.w)};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = v[i] * a.scale1 + a.rs1 * r1[i];
                v[i] = v[i] * T.scale2 + T.rs2 * r2[i];
            }
        }
        if constexpr (SIDE == 1) {
            const uint4 q = raw.q[hf];
            const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // m > 0 on the raw 16-bit patterns (bf16 and fp16 alike; activations are finite): sign clear and magnitude non-zero = a positive integer
                v[2 * k] *= (int)(u[k] << 16) > 0 ? 1.f : a.mask_slope;
                v[2 * k + 1] *= (int)u[k] > 0xffff ? 1.f : a.mask_slope;
            }
        }
        // The (row, tile) scalar goes into the VECTOR offset of the stores, not into their SGPR offset: a store of more than 8 bytes that uses an SGPR
        // offset may have its data registers overwritten by the VALU instruction behind it (observed on gfx950: the packed multiply of the out2
        // scaling changed one element of the fp32 `out` piece of the last lanes of each row; loads have no such exposure).  Out-of-range stays
        // out of range: OOB + any image offset is below 2^32 and above every num_records.
        // The lane offsets were checked against cout for tile 0 of the group; a 64-wide group whose second tile lies past cout (cout = 64 k + 32: MoSR's
        // 64 -> 96 input gradient) must drop that tile's stores too -- unmasked they landed on channels 0-31 of the NEXT pixel (found by the
        // batch-halves property test, round 3; every earlier parity case had cout % 64 == 0 or took the generic body).
        const bool tok = t * 32 < T.cg_left;
        const unsigned vo = ((rowok && tok && !(a.dbg & 4)) ? (unsigned)T.c_out[hf] : OOB);      // dbg 4 (TRAINNER_CONV_DBG, timing experiments): stores dropped by the bounds check
        if constexpr (SIDE == 2) {
            const unsigned so = (unsigned)(oyc * T.cs_out + t * 128);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, T.r_out, (int)(vo + so), 0, TR_RING_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, T.r_out, (int)(vo + so + 16u), 0, TR_RING_STORE_AUX);
            const float sc = a.out2_scale;
            const unsigned vo2 = ((rowok && tok) ? (unsigned)T.c_out2[hf] : OOB) + (unsigned)(oyc * T.cs_out2 + t * 64);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{pack2c<CT>(v[0] * sc, v[1] * sc), pack2c<CT>(v[2] * sc, v[3] * sc), pack2c<CT>(v[4] * sc, v[5] * sc), pack2c<CT>(v[6] * sc, v[7] * sc)},
                                                   T.r_out2, (int)vo2, 0, TR_RING_STORE_AUX);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{pack2c<CT>(v[0], v[1]), pack2c<CT>(v[2], v[3]), pack2c<CT>(v[4], v[5]), pack2c<CT>(v[6], v[7])}, T.r_out,
                                                   (int)(vo + (unsigned)(oyc * T.cs_out + t * 64)), 0, TR_RING_STORE_AUX);
        }
    }
}

template <int CT, int SIDE, bool FAST>
__device__ __forceinline__ void ring_epilogue_unit(const ConvArgs& a, f32x16& acc, const SideRaw<SIDE>& raw, char* strip, int64_t rowpix, bool rowok, int x0,
                                                   int cobase, int l) {
    float4 rd[2][2];
    epi_write(acc, strip, l);
    epi_read(strip, l, rd);
    epi_finish<CT, SIDE, FAST>(a, rd, raw, rowpix, rowok, x0, cobase, l);
}

// diagnostics: wave-uniform time stamp into a[stamps]; slot layout [workgroup][role 0 consumer / 1 loader][256 stamps]
#ifdef TR_RING_DEV
#define TR_STAMP(ROLE)                                                                                                  \
    if (a.stamps && l == 0 && (ROLE == 0 ? wv == 0 : wv == NCW) && nst < 256 && blockIdx.x < 1024)                                           \
        a.stamps[((int64_t)blockIdx.x * 2 + ROLE) * 256 + nst++] = (long long)__builtin_amdgcn_s_memtime();
#else
#define TR_STAMP(ROLE)
#endif

template <int CT, int NT, int R, int NCW, int NLW, int WGS, int D, bool WRES, int EPI, bool PIPE>
__global__ __launch_bounds__((NCW + NLW) * 64, WGS * (NCW + NLW) / 4) void conv_ring_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int SIDE = EPI % 3;          // side-input shape; EPI < 3: the fast epilogue forms (ring_epilogue_row_fast)
    constexpr bool FAST = EPI < 3;
    typedef Cfg<NT, R, NCW, NLW, D, WRES, PIPE> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int l = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = l & 31, lh = l >> 5;
    int nst = 0;

    // tile walk: workgroup b takes tiles first + k * grid.  Ids are dealt so that the workgroups of one XCD (ids equal mod 8 under
    // round-robin placement: speed only) own runs of consecutive tiles, i.e. vertically adjacent tiles whose halos overlap.
    const int G = gridDim.x;
    const int ntiles = a.ntx * a.nty * a.n * a.ncg;
    int b = blockIdx.x;
    if ((G & 7) == 0) b = (b & 7) * (G >> 3) + (b >> 3);
    const int nmine = (ntiles - b + G - 1) / G;
    const int total = nmine * a.nchunks;
    char* ring = smem + (WRES ? a.nchunks * C::W_BYTES : 0);
    const int64_t in_pix_bytes = (int64_t)a.in.cstride * 2;

#define TR_RING_DECODE(K, CG, NB, Y0, X0)                                                                               \
    int CG, NB, Y0, X0;                                                                                                 \
    {                                                                                                                   \
        int t_ = b + (K) * G;                                                                                           \
        CG = t_ % a.ncg; t_ /= a.ncg;                                                                                    \
        X0 = (t_ % a.ntx) * TW; t_ /= a.ntx;                                                                             \
        Y0 = (t_ % a.nty) * C::TH;                                                                                       \
        NB = t_ / a.nty;                                                                                                 \
    }

    if (wv >= NCW) {
        // =================================================== loader waves ===================================================
        const int lw = wv - NCW;
        const unsigned img_bytes = (unsigned)((int64_t)a.hin * a.win * in_pix_bytes);
        const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpacked, 0, (unsigned)((int64_t)a.ncg * a.nchunks * C::W_BYTES), 0x00020000);
        if constexpr (WRES) {
            // all K chunks of output group 0 (WRES launches have ncg == 1), once: older than every halo piece, so the first counted wait covers them
            const int npieces = a.nchunks * (C::W_ITEMS / 64);
            for (int p = lw; p < npieces; p += NLW)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(smem + p * 1024), 16, (unsigned)(p * 1024 + l * 16), 0, 0, 0);
        }
        unsigned hoff[C::HALO_IT];
        int cgk = 0;
        __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.in.ptr, 0, img_bytes, 0x00020000);
        int k = 0, ch = 0;
        int cch = 0;          // the consumers' chunk-in-tile counter, D - 1 iterations behind: the loaders join the per-tile epilogue barrier
        bool tile_end = false;
        // Pipelined mask instances: the activation-derivative mask of tile km goes to LDS as well (the consumers' epilogue hooks then issue no
        // global loads at all: a load would sit on the same in-order vmcnt as their stores and as the next tile's requests).  Tile km's mask is
        // read during the first EC chunks of tile km + 1 and its buffer is free again after that, so with MB buffers it may be issued from
        // iteration c0(km) + EC + D - (MB - 1) * n on (n chunks per tile) and must be issued by iteration c0(km) + n; it is issued AHEAD of the
        // iteration's chunk pieces, so the counted wait of the iteration D - 1 later covers it.
        constexpr bool MASKDMA = PIPE && SIDE == 1;
        constexpr int MASK_IT = MASKDMA ? (C::TH * 2) / NLW : 1;      // 1 KiB pieces per loader wave and tile (a tile row is 32 x 64 B = 2 pieces)
        char* maskbuf = ring + C::S * C::STAGE_BYTES + a.ncg * C::CGW * 4 + C::STRIPS_BYTES;
        int km = 0, jmn = MASKDMA ? min(max(C::EC + D - (a.ring_mb - 1) * a.nchunks, 0), a.nchunks) : 0;
        auto issue_mask = [&]() {
            if constexpr (MASKDMA) {
                TR_RING_DECODE(km, cg, nb, y0, x0)
                const int64_t img = (int64_t)a.h * a.w * a.mask.cstride * 2;
                const auto m_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)a.mask.ptr + (int64_t)nb * img), 0, (unsigned)img, 0x00020000);
                char* mb = maskbuf + (km % a.ring_mb) * C::MASK_BYTES;
#pragma unroll
                for (int i = 0; i < MASK_IT; ++i) {
                    const int q = i * NLW + lw;
                    const int oy = y0 + (q >> 1), ox = x0 + (q & 1) * 16 + (l >> 2), co = cg * C::CGW + (l & 3) * 8;
                    const bool ok = (oy < a.h) & (ox < a.w) & (co < a.cout);
                    const unsigned off = ok ? (unsigned)((((int64_t)oy * a.w + ox) * a.mask.cstride + a.mask.coff + co) * 2) : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(m_rsrc, (__attribute__((address_space(3))) void*)(mb + q * 1024), 16, off, 0, 0, 0);
                }
                ++km;
                jmn += a.nchunks;
            }
        };
        for (int j = 0; j < total + D - 1; ++j) {
            if (MASKDMA && j == jmn && km < nmine) issue_mask();
            if (j < total) {
                if (ch == 0) {
                    TR_RING_DECODE(k, cg, nb, y0, x0)
                    cgk = cg;
                    in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in.ptr + (int64_t)nb * a.hin * a.win * in_pix_bytes), 0, img_bytes, 0x00020000);
#pragma unroll
                    for (int i = 0; i < C::HALO_IT; ++i) {
                        int p0 = (i * NLW + lw) * 64;
                        p0 = p0 < C::HALO_ITEMS - 64 ? p0 : C::HALO_ITEMS - 64;     // the tail piece re-fetches the end of the image (duplicates are benign)
                        const int P = p0 + l;
                        const int pp = P >> 2;
                        const int py = pp / HALO_W, px = pp - py * HALO_W;
                        const int s = (P & 3) ^ ((px >> 2) & 3);                    // bank swizzle (by the column) on the SOURCE: the LDS image is position-linear
                        int oy = y0 + py - 1 - a.shift, ox = x0 + px - 1 - a.shift;
                        bool ok = (oy >= 0) & (oy < a.ih) & (ox >= 0) & (ox < a.iw);
                        if (a.replicate) {
                            oy = min(max(oy, 0), a.ih - 1);
                            ox = min(max(ox, 0), a.iw - 1);
                            ok = true;
                        }
                        const unsigned off = (unsigned)(((int64_t)(oy >> a.up) * a.win + (ox >> a.up)) * in_pix_bytes + (int64_t)(a.in.coff + s * 8) * 2);
                        hoff[i] = ok ? off : OOB;
                    }
                }
                char* st = ring + (j % C::S) * C::STAGE_BYTES;
#pragma unroll
                for (int i = 0; i < C::HALO_IT; ++i) {
                    int p0 = (i * NLW + lw) * 64;
                    p0 = p0 < C::HALO_ITEMS - 64 ? p0 : C::HALO_ITEMS - 64;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (__attribute__((address_space(3))) void*)(st + p0 * 16), 16, hoff[i], ch * 64, 0, 0);
                }
                if constexpr (!WRES) {
                    const unsigned wbase = (unsigned)((cgk * a.nchunks + ch) * C::W_BYTES);
#pragma unroll
                    for (int i = 0; i < C::W_IT; ++i) {
                        int p0 = (i * NLW + lw) * 64;
                        p0 = p0 < C::W_ITEMS - 64 ? p0 : C::W_ITEMS - 64;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(st + C::HALO_BYTES + p0 * 16), 16,
                                                                 (unsigned)(p0 * 16 + l * 16), wbase, 0, 0);
                    }
                }
                if (++ch == a.nchunks) { ch = 0; ++k; }
                if (!PIPE && tile_end) __builtin_amdgcn_s_barrier();      // the consumers' "stage is free for the strips" barrier of the tile that ended in iteration j - 1
                TR_STAMP(1)
                // chunk j - (D - 1) has landed once at most (D - 1) chunks of this wave's pieces are outstanding
                if constexpr (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if constexpr (D == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NP) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C::NP) : "memory");
            } else {
                if (!PIPE && tile_end) __builtin_amdgcn_s_barrier();
                TR_STAMP(1)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            TR_STAMP(1)
            __builtin_amdgcn_s_barrier();
            TR_STAMP(1)
            tile_end = false;
            if (j >= D - 1 && ++cch == a.nchunks) { cch = 0; tile_end = true; }
        }
        if (!PIPE && tile_end) __builtin_amdgcn_s_barrier();              // the last tile's epilogue barrier
        if constexpr (PIPE) {      // tail barrier of the pipelined kernels: the last tile's mask has landed before its (unhidden) epilogue
            if (MASKDMA && km < nmine) issue_mask();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ===================================================== consumer waves =====================================================
    // the bias seeds the accumulators (no add in the epilogue): all output groups, zero padded, in LDS behind the ring
    // EVERY consumer wave writes the whole array (identical values: the duplicates are benign) and waits for its own writes: a pipelined instance
    // with ring depth 1 seeds its first accumulators before any barrier, so a wave may not rely on another wave's writes here (round 3: with one
    // wave writing, waves 1..3 of the D = 1 instances read stale LDS -- found by tests/test_conv_ring_gpu.py with ring_cfg = 31)
    float* bias_lds = (float*)(ring + C::S * C::STAGE_BYTES);
    for (int i = l; i < a.ncg * C::CGW; i += 64) bias_lds[i] = (a.bias && i < a.cout) ? a.bias[i] : 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const FragAddr fa = make_frag_addr(lr, lh);
    if constexpr (PIPE) {
        // ---- pipelined consumers (fast epilogue forms with SIDE 0 / 1 only): two accumulator sets; tile k's epilogue runs as U + 1 stages hooked
        // between the MFMA steps of tile k + 1's first EC chunks (stage H: finish unit H - 1 from the registers its strip read landed in, then park
        // unit H in the wave's strip and read it back transposed).  The mask pieces of a whole tile are requested when the tile STARTS (a second
        // register set, like the accumulators): a full tile of MFMAs later they are there.  A consumer wave is alone on its SIMD: VALU, LDS and
        // store instructions issued in the shadow of its MFMAs are free, issued after them they are MFMA-pipe idle time.
        static_assert(FAST && SIDE != 2, "pipelined instances exist for the fast forms without residual inputs");
        constexpr int U = C::U, EC = C::EC;
        char* strip = (char*)bias_lds + a.ncg * C::CGW * 4 + wv * STRIP_BYTES;
        f32x16 acc[2][R][NT];
        SideRaw<SIDE> mraw;                              // SIDE 1: the mask pieces of the unit being finished, read from the LDS mask tile
        const char* maskbuf = strip - wv * STRIP_BYTES + C::STRIPS_BYTES;
        int pk = 0;                                      // tile number of the pending epilogue (mask buffer pk % MB)
        float4 rd[2][2];
        int py0 = 0;                                     // first row of the tile whose epilogue is pending
        EpiTile<SIDE> T;                                 // its buffer-resource context: built at the start of a tile's last chunk (nchunks > EC)
        bool have_prev = false;
        for (int i = 0; i < D - 1; ++i) __builtin_amdgcn_s_barrier();
        int c = 0;                                       // chunk counter of this workgroup (ring stage = c % S)
        auto run_tile = [&](auto pc, int k) {
            constexpr int P = decltype(pc)::value;
            TR_RING_DECODE(k, cg, nb, y0, x0)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b4 = *(const float4*)(bias_lds + cg * C::CGW + t * 32 + q * 8 + lh * 4);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        acc[P][r][t][q * 4 + 0] = b4.x; acc[P][r][t][q * 4 + 1] = b4.y; acc[P][r][t][q * 4 + 2] = b4.z; acc[P][r][t][q * 4 + 3] = b4.w;
                    }
                }
            auto hook = [&](auto hc) {
                constexpr int H = decltype(hc)::value;
                if constexpr (H >= 1) {
                    constexpr int u = H - 1, r = u / NT, t = u % NT;
                    epi_finish_fast<CT, SIDE>(a, T, rd, mraw, py0 + wv * R + r, t);
                }
                if constexpr (H < U) {
                    constexpr int r = H / NT, t = H % NT;
                    epi_write(acc[1 - P][r][t], strip, l);
                    epi_read(strip, l, rd);
                    if constexpr (SIDE == 1) {
                        const char* mb = maskbuf + (pk % a.ring_mb) * C::MASK_BYTES + ((wv * R + r) * 32 + (l >> 2)) * 64 + (l & 3) * 16;
                        mraw.q[0] = *(const uint4*)mb;
                        mraw.q[1] = *(const uint4*)(mb + 16 * 64);
                    }
                }
            };
            for (int ch = 0; ch < a.nchunks; ++ch, ++c) {
                TR_STAMP(0)
                __builtin_amdgcn_s_barrier();
                TR_STAMP(0)
                const char* st = ring + (c % C::S) * C::STAGE_BYTES;
                const char* wl = WRES ? smem + ch * C::W_BYTES : st + C::HALO_BYTES;
                if (have_prev && ch == 0) ring_chunk_mfma_hooked<CT, NT, R, 0, (U + 1 < 5 ? U + 1 : 5)>(st + wv * R * (HALO_W * 64), wl, acc[P], fa, hook);
                else if (EC > 1 && have_prev && ch == 1) ring_chunk_mfma_hooked<CT, NT, R, 5, (U + 1 > 5 ? U + 1 - 5 : 0)>(st + wv * R * (HALO_W * 64), wl, acc[P], fa, hook);
                else ring_chunk_mfma<CT, NT, R>(st + wv * R * (HALO_W * 64), wl, acc[P], fa);
                TR_STAMP(0)
                // nchunks > EC: the previous tile's epilogue is over, the store context passes to this tile
                if (ch == EC - 1 || (ch == 0 && !have_prev)) {
                    epi_tile_setup<CT, SIDE>(a, T, nb, y0, x0, cg * C::CGW, l);
                    py0 = y0;
                    pk = k;
                }
            }
            have_prev = true;
            // the last tile of the workgroup: its epilogue has no MFMAs to hide behind
            if (k + 1 == nmine) {
                __builtin_amdgcn_s_barrier();                    // tail barrier: the loaders have waited for the last mask tile
                auto tail = [&](auto hc) {
                    constexpr int H = decltype(hc)::value;
                    if constexpr (H >= 1) {
                        constexpr int u = H - 1, r = u / NT, t = u % NT;
                        epi_finish_fast<CT, SIDE>(a, T, rd, mraw, py0 + wv * R + r, t);
                    }
                    if constexpr (H < U) {
                        constexpr int r = H / NT, t = H % NT;
                        epi_write(acc[P][r][t], strip, l);
                        epi_read(strip, l, rd);
                        if constexpr (SIDE == 1) {
                            const char* mb = maskbuf + (pk % a.ring_mb) * C::MASK_BYTES + ((wv * R + r) * 32 + (l >> 2)) * 64 + (l & 3) * 16;
                            mraw.q[0] = *(const uint4*)mb;
                            mraw.q[1] = *(const uint4*)(mb + 16 * 64);
                        }
                    }
                };
                static_for<U + 1>(tail);
            }
        };
        for (int k = 0; k < nmine; k += 2) {
            run_tile(std::integral_constant<int, 0>{}, k);
            if (k + 1 < nmine) run_tile(std::integral_constant<int, 1>{}, k + 1);
        }
        return;
    }
    f32x16 acc[R][NT];
    SideRaw<SIDE> raw[2];
    EpiTile<SIDE> T;
    int k = 0, ch = 0;
    int cg = 0, nb = 0, y0 = 0, x0 = 0;
    for (int j = 0; j < total + D - 1; ++j) {
        TR_STAMP(0)
        __builtin_amdgcn_s_barrier();
        TR_STAMP(0)
        if (j < D - 1) continue;
        if (ch == 0) {
            TR_RING_DECODE(k, cg_, nb_, y0_, x0_)
            cg = cg_; nb = nb_; y0 = y0_; x0 = x0_;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b4 = *(const float4*)(bias_lds + cg * C::CGW + t * 32 + q * 8 + lh * 4);   // accumulator register q*4 + j <-> channel q*8 + lh*4 + j
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        acc[r][t][q * 4 + 0] = b4.x; acc[r][t][q * 4 + 1] = b4.y; acc[r][t][q * 4 + 2] = b4.z; acc[r][t][q * 4 + 3] = b4.w;
                    }
                }
        }
        const int c = j - (D - 1);
        const char* st = ring + (c % C::S) * C::STAGE_BYTES;
        const char* wl = WRES ? smem + ch * C::W_BYTES : st + C::HALO_BYTES;
        const bool last = ch + 1 == a.nchunks;
        // pixel of this lane in row r of the wave (clamped to the image: loads stay in range, stores are masked by `ok`)
        const int64_t rowbase = (int64_t)nb * a.h;
        if constexpr (FAST) {
            if (last) {
                epi_tile_setup<CT, SIDE>(a, T, nb, y0, x0, cg * C::CGW, l);
                if constexpr (SIDE != 0) side_issue_fast<SIDE>(T, raw[0], min(y0 + wv * R, a.h - 1), 0);
            }
        } else if (SIDE != 0 && last) side_issue<SIDE>(a, raw[0], (rowbase + min(y0 + wv * R, a.h - 1)) * a.w, x0, cg * C::CGW, l);
        ring_chunk_mfma<CT, NT, R>(st + wv * R * (HALO_W * 64), wl, acc, fa);
        TR_STAMP(0)
        if (last) {
            ch = 0;
            ++k;
            // every consumer wave is done with the stage of this chunk: its bytes become the epilogue strips (two per wave).  The loaders join this
            // barrier right after issuing their next chunk, which goes to another stage (S >= 2).
            __builtin_amdgcn_s_barrier();
            TR_STAMP(0)
            char* strips = const_cast<char*>(st) + wv * (C::NSTRIP * STRIP_BYTES);
#pragma unroll
            for (int u = 0; u < R * NT; ++u) {
                const int r = u / NT, t = u % NT;
                const int oy = y0 + wv * R + r;
                char* strip = strips + (u % C::NSTRIP) * STRIP_BYTES;
                if constexpr (FAST) {
                    if constexpr (SIDE != 0) {
                        if (u + 1 < R * NT) side_issue_fast<SIDE>(T, raw[(u + 1) & 1], min(y0 + wv * R + (u + 1) / NT, a.h - 1), (u + 1) % NT);
                    }
                    float4 rd[2][2];
                    epi_write(acc[r][t], strip, l);
                    epi_read(strip, l, rd);
                    epi_finish_fast<CT, SIDE>(a, T, rd, raw[u & 1], oy, t);
                } else {
                    if (SIDE != 0 && u + 1 < R * NT) {
                        const int r1 = (u + 1) / NT, t1 = (u + 1) % NT;
                        side_issue<SIDE>(a, raw[(u + 1) & 1], (rowbase + min(y0 + wv * R + r1, a.h - 1)) * a.w, x0, cg * C::CGW + t1 * 32, l);
                    }
                    ring_epilogue_unit<CT, SIDE, FAST>(a, acc[r][t], raw[u & 1], strip, (rowbase + min(oy, a.h - 1)) * a.w, oy < a.h, x0, cg * C::CGW + t * 32, l);
                }
            }
        } else {
            ++ch;
        }
    }
#undef TR_RING_DECODE
#endif
}

template <int CT, int NT, int R, int NCW, int NLW, int WGS, int D, bool WRES, bool PIPE, int EPI>
int launch_ring_side(hipStream_t st, const ConvArgs& a_in) {
    typedef Cfg<NT, R, NCW, NLW, D, WRES, PIPE> C;
    ConvArgs a = a_in;
    if (PIPE && a.nchunks <= C::EC) return -1;             // the previous tile's epilogue must be over before a tile's last chunk
    a.ring_mb = 1;
    if (PIPE && EPI == 1) {                                // LDS mask tiles: one buffer if a tile has room for "free, then refill" (n >= EC + D), else two
        static_assert(!(PIPE && EPI == 1) || NT == 1, "the LDS mask tile is one 32-channel group wide");
        a.ring_mb = a.nchunks >= C::EC + D ? 1 : 2;
        if (C::EC + D > a.ring_mb * a.nchunks) return -1;
    }
    const size_t lds = (size_t)C::S * C::STAGE_BYTES + (WRES ? (size_t)a.nchunks * C::W_BYTES : 0) + (size_t)a.ncg * C::CGW * 4 + C::STRIPS_BYTES +
                       ((PIPE && EPI == 1) ? (size_t)a.ring_mb * C::MASK_BYTES : 0);   // ring + resident weights + bias (+ strips, + mask tiles)
    if (lds > (size_t)(LDS_MAX / WGS)) return -1;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_ring_kernel<CT, NT, R, NCW, NLW, WGS, D, WRES, EPI, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
        attr_set = true;
    }
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu <= 0) ncu = 256;
    }
    const int64_t ntiles = (int64_t)a.ntx * a.nty * a.n * a.ncg;
    const unsigned grid = (unsigned)(ntiles < (int64_t)ncu * WGS ? ntiles : (int64_t)ncu * WGS);
    hipLaunchKernelGGL((conv_ring_kernel<CT, NT, R, NCW, NLW, WGS, D, WRES, EPI, PIPE>), dim3(grid), dim3(C::NTHREADS), lds, st, a);
    return tr::launch_status("tr_conv3x3(ring)");
}

template <int CT, int NT, int R, int NCW, int NLW, int WGS, int D, bool WRES, bool PIPE>
int launch_ring(hipStream_t st, const ConvArgs& a, int epi) {
    if constexpr (PIPE) {      // pipelined instances: fast forms without residual inputs
        if (epi == 0) return launch_ring_side<CT, NT, R, NCW, NLW, WGS, D, WRES, PIPE, 0>(st, a);
        if (epi == 1) return launch_ring_side<CT, NT, R, NCW, NLW, WGS, D, WRES, PIPE, 1>(st, a);
        return -1;
    } else {
        switch (epi) {
            case 0: return launch_ring_side<CT, NT, R, NCW, NLW, WGS, D, WRES, PIPE, 0>(st, a);
            case 1: return launch_ring_side<CT, NT, R, NCW, NLW, WGS, D, WRES, PIPE, 1>(st, a);
            case 2: return launch_ring_side<CT, NT, R, NCW, NLW, WGS, D, WRES, PIPE, 2>(st, a);
            case 3: return launch_ring_side<CT, NT, R, NCW, NLW, WGS, D, WRES, PIPE, 3>(st, a);
            case 4: return launch_ring_side<CT, NT, R, NCW, NLW, WGS, D, WRES, PIPE, 4>(st, a);
            default: return launch_ring_side<CT, NT, R, NCW, NLW, WGS, D, WRES, PIPE, 5>(st, a);
        }
    }
}

bool vec16_ok(const tr_view& v) {   // 16 consecutive channels of a record as 16-byte pieces
    if (!v.ptr) return true;
    const int per = 16 / tr::esize(v.dtype);
    return (v.cstride % per == 0) && (v.coff % per == 0) && (((uintptr_t)v.ptr) % 16 == 0);
}

}  // namespace

namespace tr {

int conv3x3_ring(hipStream_t st, ConvArgs& a, const tr_conv3x3_desc* d) {
    const tr::Tune& tn = tr::tune();                  // tr_tune_set("conv_impl", ...): 1 = never ; 2 = whenever eligible ; 0: eligible and large enough
    if (tn.conv_impl == 1) return -1;
    const bool force = tn.conv_impl == 2;
    if (d->ksize != 3 || (d->compute != TR_BF16 && d->compute != TR_F16)) return -1;
    if (d->in_mode == TR_IN_VALID || d->in_mode == TR_IN_FULL) return -1;
    if (d->cin % 32 != 0 || d->cout % 8 != 0) return -1;
    if (d->act != TR_ACT_NONE && d->act != TR_ACT_LRELU && d->act != TR_ACT_RELU) return -1;
    if (d->mask_mode != 0 || d->sample_scale || d->cout > 1024) return -1;
    int side = 0;                                                    // the two side-input shapes of the hot path (ring_epilogue_row)
    if (d->mask.ptr && !d->res1.ptr && !d->res2.ptr) side = 1;
    else if (d->res1.ptr && d->res1.dtype == TR_F32 && !d->mask.ptr && (!d->res2.ptr || d->res2.dtype == TR_F32)) side = 2;
    else if (d->mask.ptr || d->res1.ptr || d->res2.ptr) return -1;
    // fast epilogue forms (ring_epilogue_row_fast); everything else eligible takes the generic body (epi = side + 3)
    const int ct = d->compute;
    const bool lrelu_ok = d->act == TR_ACT_NONE || (d->act == TR_ACT_LRELU && d->slope >= 0.f && d->slope <= 1.f);
    bool fast = false;
    if (side == 0) fast = lrelu_ok && d->out.dtype == ct && !d->out2.ptr;
    else if (side == 1) fast = d->act == TR_ACT_NONE && d->out.dtype == ct && d->mask.dtype == ct && !d->out2.ptr;
    else fast = d->act == TR_ACT_NONE && d->out.dtype == TR_F32 && (!d->out2.ptr || d->out2.dtype == ct);
    auto img_ok = [&](const tr_view& v) { return !v.ptr || (int64_t)d->h * d->w * v.cstride * tr::esize(v.dtype) < (int64_t)0x7fff0000; };
    if (d->cout % 32 != 0 || !img_ok(d->out) || !img_ok(d->out2) || !img_ok(d->res1) || !img_ok(d->res2) || !img_ok(d->mask)) fast = false;   // buffer addressing of the fast forms
    if (tn.ring_generic) fast = false;                               // A/B switch (and the parity tests of the generic body)
    const int epi = fast ? side : side + 3;
    if (!vec16_ok(d->out) || !vec16_ok(d->out2) || !vec16_ok(d->res1) || !vec16_ok(d->res2) || !vec16_ok(d->mask)) return -1;
    if ((int64_t)a.hin * a.win * d->in.cstride * 2 >= (int64_t)0x7fff0000) return -1;            // buffer addressing: one image below 2 GiB
    const tr::PackGeom g = tr::pack_geom(d->cin, d->cout, d->compute);
    if ((int64_t)g.ncg * g.nchunks * 9 * g.cgw * 64 >= (int64_t)0x7fff0000) return -1;
    const int cfg_env = tn.ring_cfg;
    const bool bf = d->compute == TR_BF16;
    const int64_t npix = (int64_t)a.n * a.h * a.w;
    // No size threshold any more (round 3): measured with tools/ab_ring.py, the ring kernel is ahead of the register-staged one at EVERY size down to a
    // single 16 x 16 tile (8 x 64 x 64 pixels, the per-GPU batch of BASELINE configs[2]: 64 -> 32 7 vs 14 us, 160 -> 32 9 vs 21 us, 192 -> 64 21 vs 25 us;
    // one 16 x 16 tile: 7 vs 11 us) -- its straight-line epilogue and DMA staging have the shorter single-tile latency as well.  `conv_impl` 1 selects the
    // register-staged kernels (A/B, and the cross-check of tests/test_hotpath_gpu.py).
    (void)npix;
    (void)force;
    a.ntx = (a.w + TW - 1) / TW;
#define TR_RING_LAUNCH(NT, R, NCW, NLW, WGS, D, WRES, PIPE)                                                             \
    {                                                                                                                   \
        a.nty = (a.h + (R) * (NCW) - 1) / ((R) * (NCW));                                                                 \
        const int rc = TR_RING_DISPATCH_CT(NT, R, NCW, NLW, WGS, D, WRES, PIPE);                                         \
        if (rc != -1) {                                                                                                 \
            tr::tune().last_conv_family = 4;                                                                            \
            ++tr::tune().ring_launches;                                                                                 \
            tr::tune().last_conv_instance = (NT) * 10000 + (R) * 1000 + (D) * 100 + ((WRES) ? 10 : 0) + epi;             \
            return rc;                                                                                                  \
        }                                                                                                               \
    }
#ifdef TR_RING_DEV   /* development builds (TRAINNER_DEV_BUILD=1): bf16 instances only, a third of the compile time */
#define TR_RING_DISPATCH_CT(NT, R, NCW, NLW, WGS, D, WRES, PIPE) (bf ? launch_ring<TR_BF16, NT, R, NCW, NLW, WGS, D, WRES, PIPE>(st, a, epi) : -1)
#else
#define TR_RING_DISPATCH_CT(NT, R, NCW, NLW, WGS, D, WRES, PIPE)                                                        \
    (bf ? launch_ring<TR_BF16, NT, R, NCW, NLW, WGS, D, WRES, PIPE>(st, a, epi) : launch_ring<TR_F16, NT, R, NCW, NLW, WGS, D, WRES, PIPE>(st, a, epi))
#endif
    // Instances (measured per shape on MI355X, tools/ab_ring.py; TRAINNER_RING_CFG overrides for A/B runs):
    //   32-wide output groups, cin <= 128: resident weights, 8-row tiles, pipelined epilogue, as many ring stages as LDS holds (cfg 33 -> D 3, 2, 1);
    //   32-wide, cin >= 160 (or cfg 0): streamed weights, 16-row tiles;
    //   64-wide groups: streamed weights, 16-row tiles.
    if (g.cgw == 32) {
        const int cfg = cfg_env >= 0 ? cfg_env : (d->cin <= 128 ? 33 : 0);
        if (g.ncg == 1 && cfg >= 30) {
            if (cfg == 33) TR_RING_LAUNCH(1, 2, 4, 4, 1, 3, true, true)
            if (cfg >= 32) TR_RING_LAUNCH(1, 2, 4, 4, 1, 2, true, true)
            TR_RING_LAUNCH(1, 2, 4, 4, 1, 1, true, true)
        }
        TR_RING_LAUNCH(1, 4, 4, 4, 1, 1, false, false)
    } else {
        // 64-wide groups: streamed weights, 16-row tiles.  (Round 3, measured and not kept: with K = 64 -- the 64 -> 64 convs of the upsampling tail -- both
        // weight chunks (74 KB) resident and the ring carrying halo bytes only: same time, 2.24 vs 2.24 ms at 16 x 1024^2.  In-kernel stamps of that launch
        // (tools/ring_stamps.py, 37k cycles per tile): MFMAs 11.2k, the eight epilogue units of a wave 10.2k (store-issue bound: 64-byte pieces of 128-byte
        // pixel records), waits for the one-chunk-ahead halo 7-10k.)
        TR_RING_LAUNCH(2, 4, 4, 4, 1, 1, false, false)
    }
#undef TR_RING_LAUNCH
#undef TR_RING_DISPATCH_CT
    return -1;
}

}  // namespace tr
