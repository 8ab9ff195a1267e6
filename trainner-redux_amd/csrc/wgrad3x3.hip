// Weight / bias gradient of the 3x3 convolution on NHWC views, reduction over pixels on MFMA.
//
// Contract: include/trainner_hip.h (tr_wgrad3x3).  Replaces what autograd's conv backward
// accumulates into Conv2d.weight.grad / .bias.grad at traiNNer/models/sr_model.py:681.
//
//   dW[co][tap][ci] = sum_p gy[p][co] * x[p + tap][ci]       GEMM: M = co, N = ci, K = pixels
//
// Mapping to gfx950 (DESIGN.md "wgrad kernel"):
//   * grid = (pixel-tile slots, 32-channel ci chunks, cout groups); a workgroup of 3 waves walks
//     its share of 8x32-pixel tiles and keeps dW[CG][ky][0..2][32 ci] in accumulators (wave = ky),
//     so partial sums leave the chip once per workgroup, not once per tile;
//   * K (pixels) is the strided axis of both NHWC operands, so bf16 fragments are fetched with
//     ds_read_b64_tr_b16 (4 pixels x 16 channels transposed per 16-lane group); f32 fragments are
//     one float per lane and need no transpose (v_mfma_f32_32x32x2_f32);
//   * partials go to a caller workspace and a second kernel sums them in fixed slot order:
//     bitwise reproducible, no float atomics.
#include "common.h"
#include "wgrad_args.h"
#include <stdlib.h>
#include <string.h>

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2v;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TW = 32, TH = 8;
constexpr int NTHREADS = 192;
constexpr int CKW = 32;   // input channels per workgroup


template <int CT, int NT, int KS>
__global__ __launch_bounds__(NTHREADS, (CT == TR_F32 && NT == 2) ? 1 : 2) void wgrad_kernel(WgArgs a) {
    // KS = 3: wave = ky, three kx accumulators.  KS = 1 (nn.Linear over tokens): one tap; the three waves take interleaved
    // tile rows and emit their own partial blocks (the reduction kernel sums 3x as many).
    constexpr int PAD = KS / 2;
    constexpr int HALO_W = TW + KS - 1, HALO_H = TH + KS - 1;
    constexpr int ES = CT == TR_F32 ? 4 : 2;
    constexpr int EPS = 16 / ES;
    constexpr int CG = NT * 32;
    constexpr int XPIX_BYTES = CKW * ES;                 // 64 (bf16) / 128 (f32)
    constexpr int XSLOTS = XPIX_BYTES / 16;
    constexpr int GPIX_BYTES = CG * ES;
    constexpr int GSLOTS = GPIX_BYTES / 16;
    constexpr int X_BYTES = HALO_H * HALO_W * XPIX_BYTES;
    constexpr bool GSWZ = (CT != TR_F32 && NT == 2);     // 128-byte 16-bit rows: swap 64-byte halves on odd pixel pairs

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xl = smem;
    char* gl = smem + X_BYTES;

    const int tid = threadIdx.x;
    const int l = tid & 63, ky = tid >> 6;
    const int lr = l & 31, lh = l >> 5;
    const int slot = blockIdx.x, chunk = blockIdx.y, cg = blockIdx.z;

    f32x16 acc[KS][NT];
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[k][t][i] = 0.f;

    // bias gradient rides along in the (chunk 0, ky 0) wave: column sums of the gy fragments it already holds
    const bool do_bias = (a.db_partials != nullptr) & (chunk == 0) & ((KS == 1) | (ky == 0));
    const int kyo = KS == 3 ? ky : 0;          // row offset of this wave's tap
    const int r_first = KS == 3 ? 0 : ky, r_step = KS == 3 ? 1 : 3;
    float bsum[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bsum[t] = 0.f;

    const char* in_base = (const char*)a.in.ptr;
    const char* gy_base = (const char*)a.gy.ptr;
    const int64_t in_pix_bytes = (int64_t)a.in.cstride * ES;
    const int64_t gy_pix_bytes = (int64_t)a.gy.cstride * ES;

    // staging: all 16-byte pieces of the NEXT tile are issued into registers before the current tile's MFMAs
    // (same T14-style prefetch as conv3x3.hip), written to LDS after the barrier that ends the MFMAs.
    constexpr int X_ITEMS = HALO_H * HALO_W * XSLOTS;
    constexpr int G_ITEMS = TH * TW * GSLOTS;
    constexpr int X_IT = (X_ITEMS + NTHREADS - 1) / NTHREADS;
    constexpr int G_IT = (G_ITEMS + NTHREADS - 1) / NTHREADS;
    uint4 xreg[X_IT], greg[G_IT];
#define TR_WG_ISSUE(TILE)                                                                                               \
    {                                                                                                                   \
        int t_ = (TILE);                                                                                                \
        const int x0_ = (t_ % a.ntx) * TW; t_ /= a.ntx;                                                                  \
        const int y0_ = (t_ % a.nty) * TH;                                                                               \
        const int nb_ = t_ / a.nty;                                                                                      \
        int tb = tid;                                                                                                   \
        asm volatile("" : "+v"(tb)); /* recompute piece coordinates here instead of keeping them live */                \
        _Pragma("unroll") for (int j = 0; j < X_IT; ++j) {                                                               \
            const int it = tb + j * NTHREADS;                                                                            \
            const int pp = it / XSLOTS, s_ = it - pp * XSLOTS;                                                           \
            const int py = pp / HALO_W, px = pp - py * HALO_W;                                                           \
            int oy = y0_ + py - PAD - a.shift, ox = x0_ + px - PAD - a.shift;                                                \
            bool ok = (it < X_ITEMS) & (oy >= 0) & (oy < a.ih) & (ox >= 0) & (ox < a.iw);                                \
            if (a.replicate) {                                                                                          \
                oy = min(max(oy, 0), a.ih - 1);                                                                          \
                ox = min(max(ox, 0), a.iw - 1);                                                                          \
                ok = it < X_ITEMS;                                                                                       \
            }                                                                                                           \
            const int c0 = chunk * CKW + s_ * EPS;                                                                       \
            ok = ok & (c0 + EPS <= a.in.c);                                                                              \
            xreg[j] = make_uint4(0, 0, 0, 0);                                                                            \
            if (ok) xreg[j] = *(const uint4*)(in_base + (((int64_t)nb_ * a.hin + (oy >> a.up)) * a.win + (ox >> a.up)) * in_pix_bytes + \
                                              (int64_t)(a.in.coff + c0) * ES);                                           \
        }                                                                                                               \
        _Pragma("unroll") for (int j = 0; j < G_IT; ++j) {                                                               \
            const int it = tb + j * NTHREADS;                                                                            \
            const int gp = it / GSLOTS, s_ = it - gp * GSLOTS;                                                           \
            const int oy = y0_ + (gp >> 5), ox = x0_ + (gp & 31);                                                        \
            const int c0 = cg * CG + s_ * EPS;                                                                           \
            const bool ok = (it < G_ITEMS) & (oy < a.h) & (ox < a.w) & (c0 + EPS <= a.gy.c);                             \
            greg[j] = make_uint4(0, 0, 0, 0);                                                                            \
            if (ok) greg[j] = *(const uint4*)(gy_base + (((int64_t)nb_ * a.h + oy) * a.w + ox) * gy_pix_bytes +          \
                                              (int64_t)(a.gy.coff + c0) * ES);                                           \
        }                                                                                                               \
    }

    if (slot < a.ntiles) TR_WG_ISSUE(slot)
    for (int tile = slot; tile < a.ntiles; tile += a.nslots) {
        __syncthreads();   // previous tile's fragments have all been read
#pragma unroll
        for (int j = 0; j < X_IT; ++j) {
            const int it = tid + j * NTHREADS;
            if (it < X_ITEMS) {
                const int pp = it / XSLOTS, s_ = it - pp * XSLOTS;
                *(uint4*)(xl + pp * XPIX_BYTES + s_ * 16) = xreg[j];
            }
        }
#pragma unroll
        for (int j = 0; j < G_IT; ++j) {
            const int it = tid + j * NTHREADS;
            if (it < G_ITEMS) {
                const int gp = it / GSLOTS, s_ = it - gp * GSLOTS;
                int so = s_ * 16;
                if (GSWZ) so ^= ((gp >> 1) & 1) << 6;
                *(uint4*)(gl + gp * GPIX_BYTES + so) = greg[j];
            }
        }
        __syncthreads();
        if (tile + a.nslots < a.ntiles) TR_WG_ISSUE(tile + a.nslots)   // in flight during this tile's MFMAs

        if constexpr (CT != TR_F32) {
            // Software-pipelined over the 16 (row, 16-pixel half) steps of the tile: the transposing LDS reads of step s+1 are issued
            // before the MFMAs of step s, and every address is ONE lane-constant base plus a compile-time offset (the loop is fully
            // unrolled), so a step costs its MFMAs + 2*(NT+KS) ds_read_b64_tr_b16 and nothing else.  (The rolled version spent its
            // issue slots on 25 pointer increments per iteration and waited for each read right before its MFMA.)
            const int li = l & 15, q = li >> 2, p = li & 3;
            const int i0 = ((l >> 4) & 1) * 16;
            const int gsw = GSWZ ? ((q >> 1) & 1) : 0;       // bit 1 of the pixel index comes from q alone (all other terms are multiples of 4)
            const char* gb[2] = {gl + (8 * lh + q) * GPIX_BYTES + (i0 + 4 * p) * 2 + (gsw ? 64 : 0),
                                 gl + (8 * lh + q) * GPIX_BYTES + (i0 + 4 * p) * 2 + (gsw ? 0 : 64)};
            const char* xb = xl + (kyo * HALO_W + 8 * lh + q) * XPIX_BYTES + (i0 + 4 * p) * 2;
            constexpr int NR = KS == 3 ? TH : (TH + 2) / 3;       // rows this wave walks
            constexpr int NSTEP = NR * 2;
#define TR_TRRD(PTR) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(PTR))
#define TR_WG_FRAGS(FA, FB, STEP)                                                                                       \
    {                                                                                                                   \
        const int r_ = KS == 3 ? ((STEP) >> 1) : ky + 3 * ((STEP) >> 1);                                                 \
        const int kb_ = (STEP) & 1;                                                                                      \
        const int go_ = (r_ * 32 + kb_ * 16) * GPIX_BYTES, xo_ = (r_ * HALO_W + kb_ * 16) * XPIX_BYTES;                  \
        _Pragma("unroll") for (int t2 = 0; t2 < NT; ++t2) {                                                              \
            const s16x4 lo = TR_TRRD(gb[t2] + go_), hi = TR_TRRD(gb[t2] + go_ + 4 * GPIX_BYTES);                         \
            FA[t2] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                                   \
        }                                                                                                               \
        _Pragma("unroll") for (int kx = 0; kx < KS; ++kx) {                                                              \
            const s16x4 lo = TR_TRRD(xb + xo_ + kx * XPIX_BYTES), hi = TR_TRRD(xb + xo_ + (kx + 4) * XPIX_BYTES);         \
            FB[kx] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                                   \
        }                                                                                                               \
    }
            bf16x8 fa[2][NT], fb[2][KS];
            TR_WG_FRAGS(fa[0], fb[0], 0)
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                if (KS == 1 && ky + 3 * (s >> 1) >= TH) break;       // 1x1: the three waves take interleaved rows
                const int cur = s & 1;
                if (s + 1 < NSTEP && (KS == 3 || ky + 3 * ((s + 1) >> 1) < TH)) TR_WG_FRAGS(fa[cur ^ 1], fb[cur ^ 1], s + 1)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                    for (int t2 = 0; t2 < NT; ++t2)
                    {
                        if constexpr (CT == TR_BF16) acc[kx][t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][t2], fb[cur][kx], acc[kx][t2], 0, 0, 0);
                        else acc[kx][t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[cur][t2]), __builtin_bit_cast(f16x8, fb[cur][kx]), acc[kx][t2], 0, 0, 0);
                    }
                if (do_bias) {
#pragma unroll
                    for (int t2 = 0; t2 < NT; ++t2) {
                        const uint4 u = __builtin_bit_cast(uint4, fa[cur][t2]);
                        float sacc = bsum[t2];
                        const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if constexpr (CT == TR_BF16) sacc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, w4[e]), __builtin_bit_cast(bf16x2v, 0x3F803F80u), sacc, false);
                            else sacc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2v, w4[e]), __builtin_bit_cast(f16x2v, 0x3C003C00u), sacc, false);
                        }
                        bsum[t2] = sacc;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);                   // keep the prefetch one step ahead, not further (VGPR budget)
            }
#undef TR_WG_FRAGS
#undef TR_TRRD
        } else {
            for (int r = r_first; r < TH; r += r_step) {
#pragma unroll 4
                for (int kp = 0; kp < 16; ++kp) {
                    float af[NT];
#pragma unroll
                    for (int t2 = 0; t2 < NT; ++t2)
                    {
                        af[t2] = *(const float*)(gl + (r * 32 + kp * 2 + lh) * GPIX_BYTES + (t2 * 32 + lr) * 4);
                        if (do_bias) bsum[t2] += af[t2];
                    }
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        const float bf = *(const float*)(xl + ((r + kyo) * HALO_W + kp * 2 + lh + kx) * XPIX_BYTES + lr * 4);
#pragma unroll
                        for (int t2 = 0; t2 < NT; ++t2)
                            acc[kx][t2] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t2], bf, acc[kx][t2], 0, 0, 0);
                    }
                }
            }
        }
    }

    if (do_bias) {
#pragma unroll
        for (int t2 = 0; t2 < NT; ++t2) {
            const float v = bsum[t2] + __shfl_xor(bsum[t2], 32, 64);   // the two lane halves hold disjoint pixels of one channel
            const int64_t sl = KS == 3 ? slot : (int64_t)slot * 3 + ky;
            if (lh == 0) a.db_partials[(sl * a.ncg + cg) * CG + t2 * 32 + lr] = v;
        }
    }
#undef TR_WG_ISSUE
    // partial block: [slot][chunk][cg][co_local CG][tap 9][ci_local 32]
    const int64_t slot_eff = KS == 3 ? slot : (int64_t)slot * 3 + ky;
    float* dst = a.partials + ((slot_eff * a.nchunks + chunk) * a.ncg + cg) * (CG * KS * KS * CKW);
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
        for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = t2 * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                dst[(co * (KS * KS) + kyo * KS + kx) * CKW + lr] = acc[kx][t2][reg];
            }
}

// second stage of the two-stage reduction: blocks [0, nb_dw) sum the weight partials, blocks [nb_dw, gridDim.x) the bias partials
// (per-slot column sums written by the (chunk 0, ky 0) waves) -- one launch for both
template <int NW>      // waves per block: 4, or 16 for small weights with many slots (SPAN: 5,408 outputs x 384 slots ran on 85 blocks of 4 waves, 96 slots per wave)
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ partials, float* __restrict__ dw, int cin, int cout,
                                                  int nslots, int nchunks, int ncg, int cgw, int accumulate, int taps, int nb_dw,
                                                  const float* __restrict__ db_partials, float* __restrict__ db, int ld, float (*red)[64]) {
    if ((int)blockIdx.x >= nb_dw) {
        // bias partials: a block owns 64 columns, its NW waves share the slots (one thread per column walked all 96-384 slots alone: 31 us per launch on SPAN)
        const int bl = threadIdx.x & 63, bw = threadIdx.x >> 6;
        const int co = ((int)blockIdx.x - nb_dw) * 64 + bl;
        float s = 0.f;
        if (co < cout) {
            const int cg = co / cgw, col = co - cg * cgw;
            const float* p = db_partials + (int64_t)cg * cgw + col;
            const int64_t stride = (int64_t)ncg * cgw;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int k = bw;
            for (; k + 3 * NW < nslots; k += 4 * NW) {
                s0 += p[(k + 0) * stride]; s1 += p[(k + NW) * stride]; s2 += p[(k + 2 * NW) * stride]; s3 += p[(k + 3 * NW) * stride];
            }
            for (; k < nslots; k += NW) s0 += p[k * stride];
            s = (s0 + s1) + (s2 + s3);
        }
        red[bw][bl] = s;
        __syncthreads();
        if (bw == 0 && co < cout) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < NW; q += 4) t += (red[q][bl] + red[q + 1][bl]) + (red[q + 2][bl] + red[q + 3][bl]);
            db[co] = accumulate ? db[co] + t : t;
        }
        return;
    }
    // A block owns 64 consecutive outputs (co, tap, ci with ci fastest: coalesced 256-byte partial rows); its four waves take the slots
    // k = w, w + 4, ... with four accumulators each, i.e. 16 independent load chains per output, and are combined through LDS in a fixed
    // order (bitwise reproducible).  One thread per output walked its 96-768 slots alone: ~1.5 blocks per CU, latency bound (34 us for the
    // 37 MB of SPAN's 104 -> 104 partials, 15 % of its step).
    const int64_t total = (int64_t)cout * taps * cin;
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)nb_dw * 64) {
        const int64_t i = base + l;
        float s = 0.f;
        int ci = 0, tap = 0, co = 0;
        if (i < total) {
            ci = (int)(i % cin);
            tap = (int)((i / cin) % taps);
            co = (int)(i / ((int64_t)cin * taps));
            const int chunk = ci / CKW, cil = ci - chunk * CKW;
            const int cg = co / cgw, col = co - cg * cgw;
            const int64_t blk = (int64_t)cgw * taps * CKW;
            const float* p = partials + ((int64_t)chunk * ncg + cg) * blk + ((int64_t)col * taps + tap) * CKW + cil;
            const int64_t stride = (int64_t)nchunks * ncg * blk;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // fixed association: reproducible
            int k = wv;
            for (; k + 3 * NW < nslots; k += 4 * NW) {
                s0 += p[(k + 0) * stride]; s1 += p[(k + NW) * stride]; s2 += p[(k + 2 * NW) * stride]; s3 += p[(k + 3 * NW) * stride];
            }
            for (; k < nslots; k += NW) s0 += p[k * stride];
            s = (s0 + s1) + (s2 + s3);
        }
        red[wv][l] = s;
        __syncthreads();
        if (wv == 0 && i < total) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < NW; q += 4) t += (red[q][l] + red[q + 1][l]) + (red[q + 2][l] + red[q + 3][l]);
            float* o = dw + ((int64_t)co * ld + ci) * taps + tap;
            *o = accumulate ? *o + t : t;
        }
        __syncthreads();
    }
}


template <int NW>
__global__ __launch_bounds__(NW * 64) void wgrad_reduce_kernel(const float* __restrict__ partials, float* __restrict__ dw, int cin, int cout,
                                                               int nslots, int nchunks, int ncg, int cgw, int accumulate, int taps, int nb_dw,
                                                               const float* __restrict__ db_partials, float* __restrict__ db, int ld) {
    __shared__ float red[NW][64];
    wgrad_reduce_body<NW>(partials, dw, cin, cout, nslots, nchunks, ncg, cgw, accumulate, taps, nb_dw, db_partials, db, ld, red);
}
// The reductions of the convs of ONE fused dense-block launch (tr_wgrad3x3_multi) in one launch: blockIdx.y picks the conv.  Five launches of 6 us for
// 18-110 k outputs each were 7 % of the configs[2] step (64 -> 256 tiles) and 0.8 % of the headline step.
struct ReduceOne {
    const float* partials; float* dw; const float* db_partials; float* db;
    int cin, cout, nslots, nchunks, ncg, cgw, accumulate, taps, nb_dw, nb_db, ld;
};
struct ReduceMulti { ReduceOne r[8]; };
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(ReduceMulti m) {
    __shared__ float red[4][64];
    const ReduceOne& q = m.r[blockIdx.y];
    if ((int)blockIdx.x >= q.nb_dw + q.nb_db) return;           // the grid is as wide as the widest conv
    wgrad_reduce_body<4>(q.partials, q.dw, q.cin, q.cout, q.nslots, q.nchunks, q.ncg, q.cgw, q.accumulate, q.taps, q.nb_dw, q.db_partials, q.db, q.ld, red);
}

// The same reduction on 16-byte partial reads (cin % 4 == 0): a lane owns four consecutive input channels of one (co, tap) row -- contiguous in the
// partial blocks ([..][tap][ci 32]) -- so a block covers 256 outputs and every load chain carries four times the bytes (SwinIR-M's 156 reductions per step cost 9.5 us each
// in the 4-byte form).  Same slot order per output: bitwise
// equal to the scalar form.  `vst`: dw rows are 16-byte addressable (taps == 1, ld % 4 == 0, aligned base) -> one float4 store.
template <int NW>
__global__ __launch_bounds__(NW * 64) void wgrad_reduce_vec_kernel(const float* __restrict__ partials, float* __restrict__ dw, int cin, int cout,
                                                                int nslots, int nchunks, int ncg, int cgw, int accumulate, int taps, int nb_dw,
                                                                const float* __restrict__ db_partials, float* __restrict__ db, int ld, int vst) {
    __shared__ float4 red[NW][64];
    if ((int)blockIdx.x >= nb_dw) {
        // bias partials: a block owns 64 columns, its NW waves share the slots (one thread per column walked all 96-384 slots alone: 31 us per launch on SPAN)
        const int bl = threadIdx.x & 63, bw = threadIdx.x >> 6;
        const int co = ((int)blockIdx.x - nb_dw) * 64 + bl;
        float s = 0.f;
        if (co < cout) {
            const int cg = co / cgw, col = co - cg * cgw;
            const float* p = db_partials + (int64_t)cg * cgw + col;
            const int64_t stride = (int64_t)ncg * cgw;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int k = bw;
            for (; k + 3 * NW < nslots; k += 4 * NW) {
                s0 += p[(k + 0) * stride]; s1 += p[(k + NW) * stride]; s2 += p[(k + 2 * NW) * stride]; s3 += p[(k + 3 * NW) * stride];
            }
            for (; k < nslots; k += NW) s0 += p[k * stride];
            s = (s0 + s1) + (s2 + s3);
        }
        red[bw][bl].x = s;
        __syncthreads();
        if (bw == 0 && co < cout) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < NW; q += 4) t += (red[q][bl].x + red[q + 1][bl].x) + (red[q + 2][bl].x + red[q + 3][bl].x);
            db[co] = accumulate ? db[co] + t : t;
        }
        return;
    }
    const int64_t total4 = (int64_t)cout * taps * cin / 4;
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < total4; base += (int64_t)nb_dw * 64) {
        const int64_t i = (base + l) * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int ci = 0, tap = 0, co = 0;
        const bool on = base + l < total4;
        if (on) {
            ci = (int)(i % cin);
            tap = (int)((i / cin) % taps);
            co = (int)(i / ((int64_t)cin * taps));
            const int chunk = ci / CKW, cil = ci - chunk * CKW;
            const int cg = co / cgw, col = co - cg * cgw;
            const int64_t blk = (int64_t)cgw * taps * CKW;
            const float* p = partials + ((int64_t)chunk * ncg + cg) * blk + ((int64_t)col * taps + tap) * CKW + cil;
            const int64_t stride = (int64_t)nchunks * ncg * blk;
            float4 s0 = s, s1 = s, s2 = s, s3 = s;   // fixed association per component: the scalar kernel's
#define TR_ADD4(acc, q) { const float4 v = *(const float4*)(q); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            int k = wv;
            for (; k + 3 * NW < nslots; k += 4 * NW) {
                TR_ADD4(s0, p + (k + 0) * stride) TR_ADD4(s1, p + (k + NW) * stride) TR_ADD4(s2, p + (k + 2 * NW) * stride) TR_ADD4(s3, p + (k + 3 * NW) * stride)
            }
            for (; k < nslots; k += NW) TR_ADD4(s0, p + k * stride)
#undef TR_ADD4
            s = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
        }
        red[wv][l] = s;
        __syncthreads();
        if (wv == 0 && on) {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NW; q += 4) {
                const float4 a = red[q][l], b = red[q + 1][l], c = red[q + 2][l], d = red[q + 3][l];
                t[0] += (a.x + b.x) + (c.x + d.x); t[1] += (a.y + b.y) + (c.y + d.y); t[2] += (a.z + b.z) + (c.z + d.z); t[3] += (a.w + b.w) + (c.w + d.w);
            }
            if (vst) {
                float4* o = (float4*)(dw + (int64_t)co * ld + ci);
                float4 v = make_float4(t[0], t[1], t[2], t[3]);
                if (accumulate) { const float4 u = *o; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
                *o = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float* o = dw + ((int64_t)co * ld + ci + e) * taps + tap;
                    *o = accumulate ? *o + t[e] : t[e];
                }
            }
        }
        __syncthreads();
    }
}

// second-stage launch shared by every weight-gradient kernel
static void launch_wgrad_reduce(hipStream_t st, const float* partials, float* dw, int cin, int cout, int nslots, int nchunks, int ncg, int cgw, int accumulate,
                                int taps, const float* db_partials, float* db, int ld) {
    const int64_t total = (int64_t)cout * taps * cin;
    const int nb_db = db ? (cout + 63) / 64 : 0;            // a bias block owns 64 columns
    // same-box A/B (bench.py, MI355X): token GEMMs (taps 1, float4 stores) +1.1 % on SwinIR-M's step; 3x3 weights (stride-9 stores either way) within
    // noise on MoSR / SPAN and -0.5 % on ESRGAN's L1-only step, so only the ksize-1 reductions take the 16-byte form
    // few outputs, many slots (SPAN's 52 / 104 / 156-channel 1x1 and 3x3 weights: a few thousand outputs x 96-384 slots): 16 waves per block share an
    // output's slots, so the walk is 6 dependent steps instead of 24 and the launch has four times the waves; the bias columns' slots are shared by a block's waves the same way
    // (one thread per column walked all 384 slots alone: that, not the weight part, was SPAN's 31 us per launch)
    const bool wide = nslots >= 64 && total < (int64_t)64 * 2048;
    if (tr::tune().wgrad_reduce == 0 && taps == 1 && cin % 4 == 0 && ((uintptr_t)partials % 16) == 0) {
        int64_t b = (total / 4 + 63) / 64;
        if (b > 8192) b = 8192;
        const int vst = taps == 1 && ld % 4 == 0 && ((uintptr_t)dw % 16) == 0;
        const int nbb = nb_db;
        if (wide) hipLaunchKernelGGL(wgrad_reduce_vec_kernel<16>, dim3((unsigned)(b + nbb)), dim3(1024), 0, st, partials, dw, cin, cout, nslots, nchunks, ncg, cgw, accumulate,
                                     taps, (int)b, db_partials, db, ld, vst);
        else hipLaunchKernelGGL(wgrad_reduce_vec_kernel<4>, dim3((unsigned)(b + nbb)), dim3(256), 0, st, partials, dw, cin, cout, nslots, nchunks, ncg, cgw, accumulate,
                                taps, (int)b, db_partials, db, ld, vst);
        return;
    }
    int64_t b = (total + 63) / 64;
    if (b > 8192) b = 8192;
    if (wide) {
        const int nbb = nb_db;
        hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)(b + nbb)), dim3(1024), 0, st, partials, dw, cin, cout, nslots, nchunks, ncg, cgw, accumulate, taps,
                           (int)b, db_partials, db, ld);
        return;
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)(b + nb_db)), dim3(256), 0, st, partials, dw, cin, cout, nslots, nchunks, ncg, cgw, accumulate, taps,
                       (int)b, db_partials, db, ld);
}


// ---- nn.Linear weight gradient (ksize 1, direct input) as a split-K GEMM ----------------------------------------------------------
// dW[co][ci] = sum_t gy[t][co] x[t][ci]: M = cout, N = cin, K = tokens.  The 3x3 kernel's tiling (one 32-channel input chunk x one
// 64-channel output group per workgroup) has no tap reuse to amortise its loads at ksize 1 (measured 110 TFLOP/s on SwinIR-M's qkv).
// Here a workgroup owns a 128 (cout) x 192 (cin) block -- 2 x 2 waves, 64 x 96 each = 2 x 3 MFMA tiles -- and walks its share of the
// tokens 32 at a time: both operands are K-major in memory, so both MFMA fragments come out of row-major LDS tiles through
// ds_read_b64_tr_b16; next step's 20 KB are prefetched into registers.  Partial blocks go out in the 3x3 kernel's layout, so the same
// deterministic second-stage reduction applies.
struct LwArgs {
    tr_view in, gy;
    float* partials;
    float* db_partials;
    int64_t ntok;
    int cin, cout, nchunks, ncg, nslots, nsteps;
};
constexpr int LW_TOK = 32, LW_CO = 128, LW_CI = 192;
constexpr int LW_GROW = LW_CO * 2 + 16, LW_XROW = LW_CI * 2 + 16;   // padded LDS rows (bytes): 4 consecutive rows hit different banks

struct LwPlan { int ncg, nchunks, ncb, nib, nsteps, nslots; size_t dw_floats, db_floats; };
LwPlan lw_plan(int64_t ntok, int cin, int cout) {
    LwPlan p;
    p.ncg = (cout + 63) / 64;
    p.nchunks = (cin + CKW - 1) / CKW;
    p.ncb = (p.ncg + 1) / 2;
    p.nib = (p.nchunks + 5) / 6;
    p.nsteps = (int)((ntok + LW_TOK - 1) / LW_TOK);
    int target = 512 / (p.ncb * p.nib);
    if (target < 1) target = 1;
    // every slot writes a full cout x cin partial block: keep >= LW_MIN_STEPS token steps per slot so that the partial traffic stays
    // well below the operand traffic (at 32k tokens 102 slots wrote 45 MB of partials against 48 MB of operands)
    // same-box A/B on SwinIR-M (32,768 tokens): 8 steps per slot 409 images/s, 16: 405, 4: 404 (half the CUs idle on the 180 -> 180 Linear at 16)
    static const int min_steps = getenv("TRAINNER_LW_MINSTEPS") ? atoi(getenv("TRAINNER_LW_MINSTEPS")) : 8;
    const int cap = p.nsteps / min_steps > 0 ? p.nsteps / min_steps : 1;
    if (target > cap) target = cap;
    p.nslots = p.nsteps < target ? p.nsteps : target;
    p.dw_floats = (size_t)p.nslots * p.nchunks * p.ncg * 64 * CKW;
    p.db_floats = (size_t)p.nslots * p.ncg * 64;
    return p;
}

template <int CT>
__global__ __launch_bounds__(256, 2) void wgrad_linear_kernel(LwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* gl = smem;                              // [32 tokens][LW_GROW]
    char* xl = smem + LW_TOK * LW_GROW;           // [32 tokens][LW_XROW]
    const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6, lh = l >> 5, lr = l & 31;
    const int wm = wv >> 1, wn = wv & 1;          // cout half (64), cin half (96)
    const int slot = blockIdx.x, ib = blockIdx.y, cb = blockIdx.z;
    const int co0 = cb * LW_CO, ci0 = ib * LW_CI;
    const unsigned short* gy = (const unsigned short*)a.gy.ptr + a.gy.coff;
    const unsigned short* xin = (const unsigned short*)a.in.ptr + a.in.coff;
    const int64_t gcs = a.gy.cstride, xcs = a.in.cstride;

    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float bsum[2] = {0.f, 0.f};
    const bool do_bias = a.db_partials != nullptr && ib == 0 && wn == 0;

    // staging: 16-byte pieces; G tile 32 x 16 pieces (2 per thread), X tile 32 x 24 pieces (3 per thread)
    uint4 greg[2], xreg[3];
#define TR_LW_ISSUE(STEP)                                                                                               \
    {                                                                                                                   \
        const int64_t t0 = (int64_t)(STEP) * LW_TOK;                                                                     \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                  \
            const int it = tid + j * 256, tk = it >> 4, pc = it & 15;                                                    \
            const int c = co0 + pc * 8;                                                                                  \
            greg[j] = make_uint4(0, 0, 0, 0);                                                                            \
            if (t0 + tk < a.ntok && c + 8 <= a.gy.c) greg[j] = *(const uint4*)(gy + (t0 + tk) * gcs + c);                \
        }                                                                                                               \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                                  \
            const int it = tid + j * 256, tk = it / 24, pc = it - tk * 24;                                               \
            const int c = ci0 + pc * 8;                                                                                  \
            xreg[j] = make_uint4(0, 0, 0, 0);                                                                            \
            if (t0 + tk < a.ntok && c + 8 <= a.in.c) xreg[j] = *(const uint4*)(xin + (t0 + tk) * xcs + c);               \
        }                                                                                                               \
    }
    // transposing fragment reads: lane -> (row q of 4, 8-byte column group p) inside a 16-lane group; result: 4 tokens for column l & 31
    const int li = l & 15, q = li >> 2, p4 = li & 3, i0 = ((l >> 4) & 1) * 16;
    const char* gb = gl + (8 * lh + q) * LW_GROW + (wm * 64 + i0 + 4 * p4) * 2;
    const char* xb = xl + (8 * lh + q) * LW_XROW + (wn * 96 + i0 + 4 * p4) * 2;
#define TR_TRRD(PTR) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(PTR))

    int step = slot;
    if (step < a.nsteps) TR_LW_ISSUE(step)
    for (; step < a.nsteps; step += a.nslots) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int it = tid + j * 256, tk = it >> 4, pc = it & 15;
            *(uint4*)(gl + tk * LW_GROW + pc * 16) = greg[j];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int it = tid + j * 256, tk = it / 24, pc = it - tk * 24;
            *(uint4*)(xl + tk * LW_XROW + pc * 16) = xreg[j];
        }
        __syncthreads();
        if (step + a.nslots < a.nsteps) TR_LW_ISSUE(step + a.nslots)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {              // two 16-token MFMA steps
            bf16x8 fa[2], fb[3];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const s16x4 lo = TR_TRRD(gb + (kb * 16) * LW_GROW + i * 64), hi = TR_TRRD(gb + (kb * 16 + 4) * LW_GROW + i * 64);
                fa[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const s16x4 lo = TR_TRRD(xb + (kb * 16) * LW_XROW + j * 64), hi = TR_TRRD(xb + (kb * 16 + 4) * LW_XROW + j * 64);
                fb[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if constexpr (CT == TR_BF16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i]), __builtin_bit_cast(f16x8, fb[j]), acc[i][j], 0, 0, 0);
                }
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint4 u = __builtin_bit_cast(uint4, fa[i]);
                    const unsigned w4[4] = {u.x, u.y, u.z, u.w};
                    float sacc = bsum[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (CT == TR_BF16) sacc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, w4[e]), __builtin_bit_cast(bf16x2v, 0x3F803F80u), sacc, false);
                        else sacc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2v, w4[e]), __builtin_bit_cast(f16x2v, 0x3C003C00u), sacc, false);
                    }
                    bsum[i] = sacc;
                }
            }
        }
    }
#undef TR_LW_ISSUE
#undef TR_TRRD
    const int cg = cb * 2 + wm;
    if (cg < a.ncg) {
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);   // the lane halves hold disjoint tokens of one channel
                if (lh == 0) a.db_partials[((int64_t)slot * a.ncg + cg) * 64 + i * 32 + lr] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int chunk = ib * 6 + wn * 3 + j;
            if (chunk >= a.nchunks) continue;
            float* dst = a.partials + (((int64_t)slot * a.nchunks + chunk) * a.ncg + cg) * (64 * CKW);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int co = i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                    dst[co * CKW + lr] = acc[i][j][reg];
                }
        }
    }
}

struct Plan {
    int ntx, nty, ntiles, nslots, nchunks, ncg, cgw;
    size_t dw_floats, db_floats;
};

Plan make_plan(int n, int h, int w, int cin, int cout, int ks = 3, int total_wgs = 768) {
    Plan p;
    p.ntx = (w + TW - 1) / TW;
    p.nty = (h + TH - 1) / TH;
    p.ntiles = p.ntx * p.nty * n;
    p.nchunks = (cin + CKW - 1) / CKW;
    p.cgw = cout > 32 ? 64 : 32;
    p.ncg = (cout + p.cgw - 1) / p.cgw;
    int target = total_wgs / (p.nchunks * p.ncg);      // resident workgroups (3 waves each) / (chunk, cg) pairs that walk the same tiles
    if (target < 1) target = 1;
    p.nslots = p.ntiles < target ? p.ntiles : target;
    // workgroup id = slot + nslots * (chunk + nchunks * cg) and ids are dealt round-robin to the 8 XCDs: with nslots a multiple of 8
    // the (chunk, cg) workgroups that walk the SAME tiles sit on one XCD and share its L2 instead of each pulling the tile from HBM
    static const bool xcd_align = !(getenv("TRAINNER_WGRAD_XCD") && atoi(getenv("TRAINNER_WGRAD_XCD")) == 0);   // A/B switch
    if (xcd_align && p.nslots >= 8) p.nslots &= ~7;
    const int mult = ks == 1 ? 3 : 1;
    p.dw_floats = (size_t)p.nslots * mult * p.nchunks * p.ncg * p.cgw * ks * ks * CKW;
    p.db_floats = (size_t)p.nslots * mult * p.ncg * p.cgw;
    return p;
}

template <int CT, int NT, int KS>
int launch_wgrad(hipStream_t st, const WgArgs& a) {
    constexpr int ES = CT == TR_F32 ? 4 : 2;
    constexpr size_t lds = (size_t)(TH + KS - 1) * (TW + KS - 1) * CKW * ES + (size_t)TH * TW * NT * 32 * ES;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)wgrad_kernel<CT, NT, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_kernel<CT, NT, KS>), dim3(a.nslots, a.nchunks, a.ncg), dim3(NTHREADS), lds, st, a);
    return tr::launch_status("tr_wgrad3x3");
}

}  // namespace

extern "C" size_t tr_wgrad3x3_workspace_bytes(int n, int h, int w, int cin, int cout) {
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return 0;
    const Plan p = make_plan(n, h, w, cin, cout);
    const tr::WgRingPlan r = tr::wgrad_ring_plan(n, h, w, cin, cout);
    const size_t a = p.dw_floats + p.db_floats, b = r.dw_floats + r.db_floats;
    return (a > b ? a : b) * sizeof(float);
}
extern "C" size_t tr_wgrad_workspace_bytes(int n, int h, int w, int cin, int cout, int ksize) {
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || (ksize != 1 && ksize != 3)) return 0;
    const Plan p = make_plan(n, h, w, cin, cout, ksize);
    size_t fl = p.dw_floats + p.db_floats;
    if (ksize == 3) {
        const tr::WgRingPlan r = tr::wgrad_ring_plan(n, h, w, cin, cout);
        if (r.dw_floats + r.db_floats > fl) fl = r.dw_floats + r.db_floats;
    }
    if (ksize == 1 && cout > 32) {     // the split-K GEMM path (16-bit compute, direct input) keeps 128 x 192 blocks per workgroup
        const LwPlan q = lw_plan((int64_t)n * h * w, cin, cout);
        if (q.dw_floats + q.db_floats > fl) fl = q.dw_floats + q.db_floats;
    }
    return fl * sizeof(float);
}

extern "C" int tr_wgrad3x3(void* stream, const tr_wgrad3x3_desc* d) {
    TR_REQUIRE(d != nullptr, TR_EINVAL, "tr_wgrad3x3: null descriptor");
    TR_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, TR_EINVAL, "tr_wgrad3x3: bad dims");
    TR_REQUIRE(d->compute == TR_F32 || d->compute == TR_BF16 || d->compute == TR_F16, TR_EINVAL, "tr_wgrad3x3: bad compute dtype");
    TR_REQUIRE(d->in.ptr && d->gy.ptr && d->dw && d->partials, TR_EINVAL, "tr_wgrad3x3: in/gy/dw/partials must be set");
    TR_REQUIRE(d->in.dtype == d->compute && d->gy.dtype == d->compute, TR_EUNSUPPORTED,
               "tr_wgrad3x3: operand dtypes must equal the compute dtype");
    const int eps = d->compute == TR_F32 ? 4 : 8;
    TR_REQUIRE(d->in.c % eps == 0 && d->gy.c % eps == 0 && d->in.c >= d->cin && d->gy.c >= d->cout, TR_EALIGN,
               "tr_wgrad3x3: view channel counts must be multiples of %d covering cin/cout", eps);
    TR_REQUIRE(tr::view_vec16B_ok(d->in) && tr::view_vec16B_ok(d->gy), TR_EALIGN, "tr_wgrad3x3: views are not 16-byte sliceable");
    TR_REQUIRE(d->in.coff + d->in.c <= d->in.cstride && d->gy.coff + d->gy.c <= d->gy.cstride, TR_EINVAL, "tr_wgrad3x3: view exceeds its buffer");
    TR_REQUIRE(d->in_mode != TR_IN_NEAREST2X || (d->h % 2 == 0 && d->w % 2 == 0), TR_EINVAL, "tr_wgrad3x3: nearest-2x input needs even dims");
    TR_REQUIRE(d->in_mode >= TR_IN_DIRECT && d->in_mode <= TR_IN_VALID && (d->in_mode != TR_IN_FULL || (d->h > 2 && d->w > 2)), TR_EINVAL,
               "tr_wgrad3x3: bad in_mode");
    TR_REQUIRE(d->in_mode < TR_IN_FULL || d->pad_mode == TR_PAD_ZEROS, TR_EINVAL, "tr_wgrad3x3: full / valid modes take zero padding");
    TR_REQUIRE(d->ksize == 3 || d->ksize == 1, TR_EUNSUPPORTED, "tr_wgrad3x3: ksize must be 3 or 1");
    TR_REQUIRE(d->dw_ld == 0 || d->dw_ld >= d->cin, TR_EINVAL, "tr_wgrad3x3: dw_ld must be 0 or >= cin");
    static const bool no_lw = getenv("TRAINNER_LINEAR") && !strcmp(getenv("TRAINNER_LINEAR"), "conv");
    if (d->ksize == 1 && d->in_mode == TR_IN_DIRECT && d->compute != TR_F32 && d->cout > 32 && !no_lw) {
        const LwPlan q = lw_plan((int64_t)d->n * d->h * d->w, d->cin, d->cout);
        TR_REQUIRE(d->partials_bytes >= (q.dw_floats + q.db_floats) * sizeof(float), TR_EINVAL, "tr_wgrad3x3: workspace too small (%zu < %zu bytes)",
                   d->partials_bytes, (q.dw_floats + q.db_floats) * sizeof(float));
        LwArgs la;
        la.in = d->in; la.gy = d->gy; la.partials = d->partials; la.db_partials = d->db ? d->partials + q.dw_floats : nullptr;
        la.ntok = (int64_t)d->n * d->h * d->w; la.cin = d->cin; la.cout = d->cout; la.nchunks = q.nchunks; la.ncg = q.ncg; la.nslots = q.nslots;
        la.nsteps = q.nsteps;
        const size_t lds = (size_t)LW_TOK * (LW_GROW + LW_XROW);
        hipStream_t st = (hipStream_t)stream;
        if (d->compute == TR_BF16) hipLaunchKernelGGL(wgrad_linear_kernel<TR_BF16>, dim3(q.nslots, q.nib, q.ncb), dim3(256), lds, st, la);
        else hipLaunchKernelGGL(wgrad_linear_kernel<TR_F16>, dim3(q.nslots, q.nib, q.ncb), dim3(256), lds, st, la);
        int rc = tr::launch_status("tr_wgrad3x3(linear)");
        if (rc) return rc;
        launch_wgrad_reduce(st, d->partials, d->dw, d->cin, d->cout, q.nslots, q.nchunks, q.ncg, 64, d->accumulate, 1, d->partials + q.dw_floats, d->db,
                            d->dw_ld > 0 ? d->dw_ld : d->cin);
        return tr::launch_status("tr_wgrad3x3(linear reduce)");
    }
    // one round of RESIDENT workgroups (more would run as a half-empty second round): 3 per CU for the 32-wide bf16 variant
    // (146 VGPRs), 2 for 64-wide bf16 and 32-wide f32, 1 for 64-wide f32; the workspace query assumes the largest (768)
    static const int wgs_env = getenv("TRAINNER_WGRAD_WGS") ? atoi(getenv("TRAINNER_WGRAD_WGS")) : 0;
    const bool wide = d->cout > 32;
    const int wgs = wgs_env ? wgs_env : (d->compute != TR_F32 ? (wide ? 512 : 768) : (wide ? 256 : 512));
    const Plan p = make_plan(d->n, d->h, d->w, d->cin, d->cout, d->ksize, wgs);
    TR_REQUIRE(d->partials_bytes >= (p.dw_floats + p.db_floats) * sizeof(float), TR_EINVAL,
               "tr_wgrad3x3: workspace too small (%zu < %zu bytes)", d->partials_bytes, (p.dw_floats + p.db_floats) * sizeof(float));

    WgArgs a;
    a.n = d->n; a.h = d->h; a.w = d->w;
    a.up = d->in_mode == TR_IN_NEAREST2X ? 1 : 0;
    a.shift = d->in_mode == TR_IN_FULL ? 1 : (d->in_mode == TR_IN_VALID ? -1 : 0);
    a.ih = d->h - 2 * a.shift; a.iw = d->w - 2 * a.shift;
    a.hin = a.ih >> a.up; a.win = a.iw >> a.up;
    a.cin = d->cin; a.cout = d->cout;
    a.replicate = d->pad_mode == TR_PAD_REPLICATE;
    a.in = d->in; a.gy = d->gy;
    a.partials = d->partials;
    a.db_partials = d->db ? d->partials + p.dw_floats : nullptr;
    a.ntx = p.ntx; a.nty = p.nty; a.ntiles = p.ntiles; a.nslots = p.nslots; a.nchunks = p.nchunks; a.ncg = p.ncg;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    {   // the persistent LDS-DMA ring kernel (wgrad_ring.hip) takes the large 16-bit 3x3 launches; same partial layout, same reduction
        WgArgs ar = a;
        tr::WgRingPlan rp;
        rc = tr::wgrad3x3_ring(st, d, ar, rp);
        if (rc > 0) return rc;
        if (rc == 0) {
            launch_wgrad_reduce(st, d->partials, d->dw, d->cin, d->cout, rp.slots_eff, rp.nchunks, rp.ncg, rp.cgw, d->accumulate, 9, d->partials + rp.dw_floats,
                                d->db, d->dw_ld > 0 ? d->dw_ld : d->cin);
            return tr::launch_status("tr_wgrad3x3(ring reduce)");
        }
    }
    if (d->ksize == 1) {
        if (d->compute == TR_BF16) rc = p.cgw == 64 ? launch_wgrad<TR_BF16, 2, 1>(st, a) : launch_wgrad<TR_BF16, 1, 1>(st, a);
        else if (d->compute == TR_F16) rc = p.cgw == 64 ? launch_wgrad<TR_F16, 2, 1>(st, a) : launch_wgrad<TR_F16, 1, 1>(st, a);
        else rc = p.cgw == 64 ? launch_wgrad<TR_F32, 2, 1>(st, a) : launch_wgrad<TR_F32, 1, 1>(st, a);
    } else {
        if (d->compute == TR_BF16) rc = p.cgw == 64 ? launch_wgrad<TR_BF16, 2, 3>(st, a) : launch_wgrad<TR_BF16, 1, 3>(st, a);
        else if (d->compute == TR_F16) rc = p.cgw == 64 ? launch_wgrad<TR_F16, 2, 3>(st, a) : launch_wgrad<TR_F16, 1, 3>(st, a);
        else rc = p.cgw == 64 ? launch_wgrad<TR_F32, 2, 3>(st, a) : launch_wgrad<TR_F32, 1, 3>(st, a);
    }
    if (rc) return rc;
    {
        const int taps = d->ksize * d->ksize;
        const int mult = d->ksize == 1 ? 3 : 1;
        launch_wgrad_reduce(st, d->partials, d->dw, d->cin, d->cout, p.nslots * mult, p.nchunks, p.ncg, p.cgw, d->accumulate, taps, d->partials + p.dw_floats,
                            d->db, d->dw_ld > 0 ? d->dw_ld : d->cin);
        rc = tr::launch_status("tr_wgrad3x3(reduce)");
    }
    return rc;
}

// Several 3x3 weight gradients over channel slices of ONE input buffer and ONE output-gradient buffer (the five convs of a dense block:
// conv k reads x[:, 0:cin_k] and its own 32 / 64 channels of the block's gradient buffer) in one launch of the ring kernel, then one reduction
// per conv.  Launched separately, x[:, 0:64] is fetched five times and the gradient slices once per 64-channel input block (1.7 GB per dense
// block at batch 16, 256^2); here the workgroups that walk the same tiles sit on one XCD and share every byte through its L2 (0.8 GB).
// Falls back to one tr_wgrad3x3 per descriptor when the set does not fit that shape.
extern "C" int tr_wgrad3x3_multi(void* stream, const tr_wgrad3x3_desc* descs, int ndesc) {
    TR_REQUIRE(descs != nullptr && ndesc > 0, TR_EINVAL, "tr_wgrad3x3_multi: no descriptors");
    bool ok = ndesc > 1;
    for (int k = 0; k < ndesc && ok; ++k) {
        const tr_wgrad3x3_desc& d = descs[k];
        ok = d.n > 0 && d.h > 0 && d.w > 0 && d.cin > 0 && d.cout > 0 && d.in.ptr && d.gy.ptr && d.dw && d.partials && d.in.dtype == d.compute &&
             d.gy.dtype == d.compute && d.in.c % 8 == 0 && d.gy.c % 8 == 0 && d.in.c >= d.cin && d.gy.c >= d.cout && tr::view_vec16B_ok(d.in) &&
             tr::view_vec16B_ok(d.gy) && d.in.coff + d.in.c <= d.in.cstride && d.gy.coff + d.gy.c <= d.gy.cstride && (d.dw_ld == 0 || d.dw_ld >= d.cin);
    }
    int nslots = 0;
    int rc = ok ? tr::wgrad3x3_ring_multi((hipStream_t)stream, descs, ndesc, &nslots) : -1;
    if (rc > 0) return rc;
    if (rc < 0) {
        for (int k = 0; k < ndesc; ++k) {
            rc = tr_wgrad3x3(stream, &descs[k]);
            if (rc) return rc;
        }
        return 0;
    }
    if (ndesc <= 8) {
        ReduceMulti m;
        unsigned gx = 1;
        for (int k = 0; k < ndesc; ++k) {
            const tr_wgrad3x3_desc& d = descs[k];
            const int nchunks = (d.cin + 31) / 32, ncg = (d.cout + 31) / 32;
            const size_t dw_floats = (size_t)nslots * nchunks * ncg * 32 * 9 * 32;
            int64_t b = ((int64_t)d.cout * 9 * d.cin + 63) / 64;
            if (b > 8192) b = 8192;
            ReduceOne& q = m.r[k];
            q.partials = d.partials; q.dw = d.dw; q.db_partials = d.partials + dw_floats; q.db = d.db;
            q.cin = d.cin; q.cout = d.cout; q.nslots = nslots; q.nchunks = nchunks; q.ncg = ncg; q.cgw = 32; q.accumulate = d.accumulate; q.taps = 9;
            q.nb_dw = (int)b; q.nb_db = d.db ? (d.cout + 63) / 64 : 0; q.ld = d.dw_ld > 0 ? d.dw_ld : d.cin;
            if ((unsigned)(q.nb_dw + q.nb_db) > gx) gx = (unsigned)(q.nb_dw + q.nb_db);
        }
        hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(gx, (unsigned)ndesc), dim3(256), 0, (hipStream_t)stream, m);
        return tr::launch_status("tr_wgrad3x3_multi(reduce)");
    }
    for (int k = 0; k < ndesc; ++k) {
        const tr_wgrad3x3_desc& d = descs[k];
        const int nchunks = (d.cin + 31) / 32, ncg = (d.cout + 31) / 32;
        const size_t dw_floats = (size_t)nslots * nchunks * ncg * 32 * 9 * 32;
        launch_wgrad_reduce((hipStream_t)stream, d.partials, d.dw, d.cin, d.cout, nslots, nchunks, ncg, 32, d.accumulate, 9, d.partials + dw_floats, d.db,
                            d.dw_ld > 0 ? d.dw_ld : d.cin);
    }
    return tr::launch_status("tr_wgrad3x3_multi(reduce)");
}
