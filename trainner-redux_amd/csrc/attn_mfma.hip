// Window attention (W-MSA / SW-MSA) on MFMA for bf16 operands -- forward and backward.
//
// Reference arithmetic replaced: WindowAttention.forward, traiNNer/archs/drct_arch.py:175-223 (+ shift mask :492-521, roll /
// window partition :78-109,532-568 as index arithmetic).  Numerics: q, k, v, P, dO, dS are bf16 MFMA operands; scores, the
// relative-position bias, the shift mask, the softmax and every accumulation are fp32 (oracle/precision.py `bf16_ops` matmul).
//
// Mapping to gfx950: one wave per (window, head), four independent waves per workgroup, persistent over tasks with the head fixed
// per wave.  A window is 64 tokens, head_dim <= 32 (zero padded), so every contraction is a handful of 32x32x16 MFMAs:
//   S^T = K Q^T        A = K rows, B = Q rows: operands come straight from global memory (16 B per lane), no LDS;
//   softmax            the accumulator layout puts one QUERY per lane (column) and 32 of its 64 keys in the lane's registers;
//                      the other 32 sit in lane ^ 32: max / sum are 31 in-lane ops + one cross-lane swap;
//   O^T = V^T P^T      P^T is already in B-operand layout in the accumulator registers (the K index is permuted to the
//                      accumulator's row order and V^T is read with the same permutation), V^T comes out of a row-major LDS tile
//                      through ds_read_b64_tr_b16;
//   backward           runs both orientations (lane = query for dQ and the softmax statistics, lane = key for dK / dV), so no
//                      score matrix is ever transposed through memory; the per-wave sum of dS (relative-position-bias gradient)
//                      stays in registers across the wave's windows and is reduced by a second kernel (no float atomics).
// Outputs go through a wave-private LDS tile so global stores are contiguous per token.
#include "attn.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2n;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2n;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef unsigned u32x4a __attribute__((ext_vector_type(4), aligned(4)));   // head slices start at 4-byte boundaries (even head_dim), not 16

constexpr int WS = 8, WT = 64, NTAB = (2 * WS - 1) * (2 * WS - 1);
constexpr int TILE = 4096;            // 64 rows x 32 bf16, 16-byte slots XOR-swizzled by (row >> 2) & 3

using tr::AttnArgs;

__device__ __forceinline__ int tile_off(int row, int byte) { return row * 64 + ((((byte >> 4) ^ (row >> 2)) & 3) << 4) + (byte & 15); }
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
template <int CT>
__device__ __forceinline__ unsigned pack2(float a, float b) {   // two fp32 -> packed 16-bit operands (bf16 or fp16, round to nearest even)
    const f32x2 v = {a, b};
    if constexpr (CT == TR_BF16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2n));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2n));
}
template <int CT>
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int j, float mul) {   // accumulator registers 8j .. 8j+7 -> 16-bit operand
    const unsigned a = pack2<CT>(v[8 * j + 0] * mul, v[8 * j + 1] * mul), b = pack2<CT>(v[8 * j + 2] * mul, v[8 * j + 3] * mul);
    const unsigned c = pack2<CT>(v[8 * j + 4] * mul, v[8 * j + 5] * mul), d = pack2<CT>(v[8 * j + 6] * mul, v[8 * j + 7] * mul);
    return __builtin_bit_cast(bf16x8, make_uint4(a, b, c, d));
}
template <int CT>
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (CT == TR_BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int region(int s, int n, int shift) { return s < n - WS ? 0 : (s < n - shift ? 1 : 2); }
__device__ __forceinline__ int64_t token_of(const AttnArgs& a, int bb, int wy, int wx, int i) {
    const int y = (wy * WS + (i >> 3) + a.shift) % a.h, x = (wx * WS + (i & 7) + a.shift) % a.w;
    return ((int64_t)bb * a.h + y) * a.w + x;
}
// 16-byte piece (8 head channels from d0) of a token row; channels >= hd (the next head's, or the row padding) read as zero
__device__ __forceinline__ uint4 load_piece(const unsigned short* p, int d0, int hd) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (d0 < hd) {
        const u32x4a t = *(const u32x4a*)(p + d0);
        v = make_uint4(t[0], t[1], t[2], t[3]);
        const int nv = hd - d0;                    // even, >= 2
        if (nv < 8) { v.w = 0; if (nv < 6) v.z = 0; if (nv < 4) v.y = 0; }
    }
    return v;
}
__device__ __forceinline__ void store_piece(unsigned short* p, int d0, int hd, uint4 v) {
    const int nv = hd - d0;
    if (nv >= 8) *(u32x4a*)(p + d0) = (u32x4a){v.x, v.y, v.z, v.w};
    else {
        unsigned* q = (unsigned*)(p + d0);
        if (nv >= 2) q[0] = v.x;
        if (nv >= 4) q[1] = v.y;
        if (nv >= 6) q[2] = v.z;
    }
}
// MFMA operand with lane = token row (l & 31 of tile tt), contraction over head channels: k-step ks covers d = 16 ks .. 16 ks + 15
__device__ __forceinline__ bf16x8 row_frag_lds(const char* tile, int tt, int ks, int lr, int hh) {
    return *(const bf16x8*)(tile + tile_off(tt * 32 + lr, ks * 32 + hh * 16));
}
// MFMA A operand with lane = head channel (l & 31), contraction over tokens in ACCUMULATOR ROW ORDER: element i of the k-step
// is token base + (i & 3) + 8 * (i >> 2) (+ 4 * (l >> 5), folded into `base` by the caller) -- two transposing LDS reads
__device__ __forceinline__ bf16x8 col_frag_lds(const char* tile, int base, int l) {
    const int li = l & 15, q = li >> 2, byte = ((((l >> 4) & 1) * 16) + 4 * (li & 3)) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + tile_off(base + q, byte)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + tile_off(base + 8 + q, byte)));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// accumulator tile (rows = head channels, column = this lane's token) -> row `row` of a swizzled LDS tile, as bf16
template <int CT>
__device__ __forceinline__ void acc_to_tile(char* tile, int row, int hh, const f32x16& v, float mul) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *(uint2*)(tile + tile_off(row, 16 * g + 8 * hh)) = make_uint2(pack2<CT>(v[4 * g] * mul, v[4 * g + 1] * mul), pack2<CT>(v[4 * g + 2] * mul, v[4 * g + 3] * mul));
}

#define MFMA(A, B, C) mfma16<CT>(A, B, C)

// scores of one orientation: x = s * scale + bias(table) (+ shift mask); the bias index is (qy - ky + 7) * 15 + (qx - kx + 7)
struct MaskCtx { bool on; unsigned ym, xm; };
__device__ __forceinline__ MaskCtx mask_ctx(const AttnArgs& a, int wy, int wx, int nwy, int nwx) {
    MaskCtx m;
    m.on = a.shift != 0 && (wy == nwy - 1 || wx == nwx - 1);
    m.ym = m.xm = 0;
    if (m.on) {
#pragma unroll
        for (int i = 0; i < WS; ++i) {
            m.ym |= (unsigned)region(wy * WS + i, a.h, a.shift) << (2 * i);
            m.xm |= (unsigned)region(wx * WS + i, a.w, a.shift) << (2 * i);
        }
    }
    return m;
}
__device__ __forceinline__ int reg_of(const MaskCtx& m, int y, int x) { return (int)((m.ym >> (2 * y)) & 3u) * 3 + (int)((m.xm >> (2 * x)) & 3u); }

// ============================================================ forward ==========================================================
template <int CT>
__global__ __launch_bounds__(256, 2) void window_attn_fwd_mfma_kernel(AttnArgs a, int ntasks) {
    constexpr int WAVE_LDS = TILE + 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = l & 31, hh = l >> 5;
    char* tv = smem + wv * WAVE_LDS;
    float* tab = (float*)(tv + TILE);
    const int gw = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    const int head = gw % a.nh, hd = a.hd;
    for (int i = l; i < NTAB; i += 64) tab[i] = a.table[i * a.nh + head];
    const int nwx = a.w / WS, nwy = a.h / WS;
    const unsigned short* qkv = (const unsigned short*)a.qkv.ptr + a.qkv.coff + head * hd;
    unsigned short* outp = (unsigned short*)a.out.ptr + a.out.coff + head * hd;
    const int64_t qcs = a.qkv.cstride, ocs = a.out.cstride;
    const float sc = a.scale;
    const int prow = l >> 2, pd0 = (l & 3) * 8;      // piece staging: 4 lanes x 16 B per token row, 16 rows per pass

    for (int task = gw; task < ntasks; task += nwaves) {
        int t = task / a.nh;
        const int wx = t % nwx; t /= nwx;
        const int wy = t % nwy;
        const int bb = t / nwy;
        wave_sync();                                  // the previous task's output pieces have left the tile
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = j * 16 + prow;
            const uint4 v = load_piece(qkv + token_of(a, bb, wy, wx, row) * qcs + 2 * a.c, pd0, hd);
            *(uint4*)(tv + tile_off(row, pd0 * 2)) = v;
        }
        bf16x8 fq[2][2], fk[2][2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const unsigned short* p = qkv + token_of(a, bb, wy, wx, tt * 32 + lr) * qcs;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                fq[tt][ks] = __builtin_bit_cast(bf16x8, load_piece(p, ks * 16 + hh * 8, hd));
                fk[tt][ks] = __builtin_bit_cast(bf16x8, load_piece(p + a.c, ks * 16 + hh * 8, hd));
            }
        }
        f32x16 st[2][2];                              // [key tile][query tile]: rows = keys, column (lane) = query
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                for (int i = 0; i < 16; ++i) st[kt][qt][i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) st[kt][qt] = MFMA(fk[kt][ks], fq[qt][ks], st[kt][qt]);
            }
        const MaskCtx mk = mask_ctx(a, wy, wx, nwy, nwx);
        wave_sync();                                  // V tile is in LDS
        f32x16 ot[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int qy = 4 * qt + (lr >> 3), qx = l & 7;
            const float* tb = tab + (qy + WS - 1) * (2 * WS - 1) + (qx + WS - 1) - 4 * hh;   // - (ky * 15 + (r & 3)) per element
            const int rq = mk.on ? reg_of(mk, qy, qx) : 0;
            float m = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ky = 4 * kt + (r >> 2);
                    float x = st[kt][qt][r] * sc + tb[-(ky * (2 * WS - 1) + (r & 3))];
                    if (mk.on && reg_of(mk, ky, (r & 3) + 4 * hh) != rq) x -= 100.f;
                    st[kt][qt][r] = x;
                    m = fmaxf(m, x);
                }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __expf(st[kt][qt][r] - m);
                    st[kt][qt][r] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.f / sum;
#pragma unroll
            for (int i = 0; i < 16; ++i) ot[qt][i] = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    ot[qt] = MFMA(col_frag_lds(tv, kt * 32 + 16 * j + 4 * hh, l), pack8<CT>(st[kt][qt], j, inv), ot[qt]);
        }
        wave_sync();                                  // every lane is done reading V: the tile becomes the output tile
        acc_to_tile<CT>(tv, lr, hh, ot[0], 1.f);
        acc_to_tile<CT>(tv, 32 + lr, hh, ot[1], 1.f);
        wave_sync();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = j * 16 + prow;
            store_piece(outp + token_of(a, bb, wy, wx, row) * ocs, pd0, hd, *(const uint4*)(tv + tile_off(row, pd0 * 2)));
        }
    }
}

// ============================================================ backward =========================================================
// ~300 live VGPRs (four 32x32 accumulator pairs + the bias-gradient sums): one workgroup per CU with the 512-register budget
template <int CT>
__global__ __launch_bounds__(256, 1) void window_attn_bwd_mfma_kernel(AttnArgs a, int ntasks, float* __restrict__ dbpart) {
    constexpr int WAVE_LDS = 4 * TILE + 768 + 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = l & 31, hh = l >> 5;
    char* tq = smem + wv * WAVE_LDS;
    char* tk = tq + TILE;
    char* tg = tk + TILE;                 // dO
    char* to = tg + TILE;                 // output staging
    float* stat = (float*)(to + TILE);    // [0,64) row max, [64,128) 1 / row sum, [128,192) delta = sum_j P dP
    float* tab = stat + 192;
    const int gw = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    const int head = gw % a.nh, hd = a.hd;
    for (int i = l; i < NTAB; i += 64) tab[i] = a.table[i * a.nh + head];
    const int nwx = a.w / WS, nwy = a.h / WS;
    const unsigned short* qkv = (const unsigned short*)a.qkv.ptr + a.qkv.coff + head * hd;
    const unsigned short* dout = (const unsigned short*)a.dout.ptr + a.dout.coff + head * hd;
    unsigned short* dqkv = (unsigned short*)a.dqkv.ptr + a.dqkv.coff + head * hd;
    const int64_t qcs = a.qkv.cstride, gcs = a.dout.cstride, dcs = a.dqkv.cstride;
    const float sc = a.scale;
    const int prow = l >> 2, pd0 = (l & 3) * 8;

    f32x16 dbacc[2][2];                               // [key tile][query tile] sum of dS over this wave's windows
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int i = 0; i < 16; ++i) dbacc[kt][qt][i] = 0.f;

    for (int task = gw; task < ntasks; task += nwaves) {
        int t = task / a.nh;
        const int wx = t % nwx; t /= nwx;
        const int wy = t % nwy;
        const int bb = t / nwy;
        wave_sync();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = j * 16 + prow;
            const int64_t tok = token_of(a, bb, wy, wx, row);
            const int off = tile_off(row, pd0 * 2);
            *(uint4*)(tq + off) = load_piece(qkv + tok * qcs, pd0, hd);
            *(uint4*)(tk + off) = load_piece(qkv + tok * qcs + a.c, pd0, hd);
            *(uint4*)(tg + off) = load_piece(dout + tok * gcs, pd0, hd);
        }
        bf16x8 fv[2][2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const unsigned short* p = qkv + token_of(a, bb, wy, wx, tt * 32 + lr) * qcs + 2 * a.c;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fv[tt][ks] = __builtin_bit_cast(bf16x8, load_piece(p, ks * 16 + hh * 8, hd));
        }
        const MaskCtx mk = mask_ctx(a, wy, wx, nwy, nwx);
        wave_sync();

        // ---- orientation T: rows = keys, column (lane) = query -> softmax statistics, dS^T, dQ ------------------------------------
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            f32x16 s[2], dp[2];
            {
                const bf16x8 bq0 = row_frag_lds(tq, qt, 0, lr, hh), bq1 = row_frag_lds(tq, qt, 1, lr, hh);
                const bf16x8 bg0 = row_frag_lds(tg, qt, 0, lr, hh), bg1 = row_frag_lds(tg, qt, 1, lr, hh);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) { s[kt][i] = 0.f; dp[kt][i] = 0.f; }
                    s[kt] = MFMA(row_frag_lds(tk, kt, 0, lr, hh), bq0, s[kt]);
                    s[kt] = MFMA(row_frag_lds(tk, kt, 1, lr, hh), bq1, s[kt]);
                    dp[kt] = MFMA(fv[kt][0], bg0, dp[kt]);
                    dp[kt] = MFMA(fv[kt][1], bg1, dp[kt]);
                }
            }
            const int qy = 4 * qt + (lr >> 3), qx = l & 7;
            const float* tb = tab + (qy + WS - 1) * (2 * WS - 1) + (qx + WS - 1) - 4 * hh;
            const int rq = mk.on ? reg_of(mk, qy, qx) : 0;
            float m = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ky = 4 * kt + (r >> 2);
                    float x = s[kt][r] * sc + tb[-(ky * (2 * WS - 1) + (r & 3))];
                    if (mk.on && reg_of(mk, ky, (r & 3) + 4 * hh) != rq) x -= 100.f;
                    s[kt][r] = x;
                    m = fmaxf(m, x);
                }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __expf(s[kt][r] - m);
                    s[kt][r] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.f / sum;
            float delta = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[kt][r] *= inv;
                    delta += s[kt][r] * dp[kt][r];
                }
            delta += __shfl_xor(delta, 32, 64);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float ds = s[kt][r] * (dp[kt][r] - delta);
                    dp[kt][r] = ds;
                    dbacc[kt][qt][r] += ds;
                }
            if (hh == 0) {
                stat[qt * 32 + lr] = m;
                stat[64 + qt * 32 + lr] = inv;
                stat[128 + qt * 32 + lr] = delta;
            }
            f32x16 dq;
#pragma unroll
            for (int i = 0; i < 16; ++i) dq[i] = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int j = 0; j < 2; ++j) dq = MFMA(col_frag_lds(tk, kt * 32 + 16 * j + 4 * hh, l), pack8<CT>(dp[kt], j, 1.f), dq);
            acc_to_tile<CT>(to, qt * 32 + lr, hh, dq, sc);
            __builtin_amdgcn_sched_barrier(0);     // keep the two query tiles' live ranges apart (VGPR budget)
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = j * 16 + prow;
            store_piece(dqkv + token_of(a, bb, wy, wx, row) * dcs, pd0, hd, *(const uint4*)(to + tile_off(row, pd0 * 2)));
        }

        __builtin_amdgcn_sched_barrier(0);
        // ---- orientation N: rows = queries, column (lane) = key -> dK, dV -----------------------------------------------------
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 s[2], dp[2];
            {
                const bf16x8 bk0 = row_frag_lds(tk, kt, 0, lr, hh), bk1 = row_frag_lds(tk, kt, 1, lr, hh);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) { s[qt][i] = 0.f; dp[qt][i] = 0.f; }
                    s[qt] = MFMA(row_frag_lds(tq, qt, 0, lr, hh), bk0, s[qt]);
                    s[qt] = MFMA(row_frag_lds(tq, qt, 1, lr, hh), bk1, s[qt]);
                    dp[qt] = MFMA(row_frag_lds(tg, qt, 0, lr, hh), fv[kt][0], dp[qt]);
                    dp[qt] = MFMA(row_frag_lds(tg, qt, 1, lr, hh), fv[kt][1], dp[qt]);
                }
            }
            const int ky = 4 * kt + (lr >> 3), kx = l & 7;
            const float* tb = tab + (WS - 1 - ky) * (2 * WS - 1) + (WS - 1 - kx) + 4 * hh;      // + qy * 15 + (r & 3) per element
            const int rk = mk.on ? reg_of(mk, ky, kx) : 0;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int q0 = qt * 32 + 8 * g + 4 * hh;                       // 4 consecutive queries: registers 4g .. 4g+3
                    const float4 mv = *(const float4*)(stat + q0), iv = *(const float4*)(stat + 64 + q0), dv4 = *(const float4*)(stat + 128 + q0);
                    const float mm[4] = {mv.x, mv.y, mv.z, mv.w}, ii[4] = {iv.x, iv.y, iv.z, iv.w}, dd[4] = {dv4.x, dv4.y, dv4.z, dv4.w};
                    const int qy = 4 * qt + g;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        float x = s[qt][r] * sc + tb[qy * (2 * WS - 1) + e];
                        if (mk.on && reg_of(mk, qy, e + 4 * hh) != rk) x -= 100.f;
                        const float p = __expf(x - mm[e]) * ii[e];
                        s[qt][r] = p;
                        dp[qt][r] = p * (dp[qt][r] - dd[e]);
                    }
                }
            f32x16 dk, dv;
#pragma unroll
            for (int i = 0; i < 16; ++i) { dk[i] = 0.f; dv[i] = 0.f; }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    dv = MFMA(col_frag_lds(tg, qt * 32 + 16 * j + 4 * hh, l), pack8<CT>(s[qt], j, 1.f), dv);
                    dk = MFMA(col_frag_lds(tq, qt * 32 + 16 * j + 4 * hh, l), pack8<CT>(dp[qt], j, 1.f), dk);
                }
            wave_sync();                              // the previous pieces have left the staging tile
            acc_to_tile<CT>(to, lr, hh, dk, sc);
            acc_to_tile<CT>(to, 32 + lr, hh, dv, 1.f);
            wave_sync();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = j * 16 + prow;        // 0 .. 31: key kt * 32 + row
                unsigned short* p = dqkv + token_of(a, bb, wy, wx, kt * 32 + row) * dcs;
                store_piece(p + a.c, pd0, hd, *(const uint4*)(to + tile_off(row, pd0 * 2)));
                store_piece(p + 2 * a.c, pd0, hd, *(const uint4*)(to + tile_off(32 + row, pd0 * 2)));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (dbpart) {                                     // [wave][key][query]: lanes (queries) contiguous
        float* o = dbpart + (int64_t)gw * (WT * WT);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    o[key * WT + qt * 32 + lr] = dbacc[kt][qt][r];
                }
    }
}

// partial[k][head][key][query] (k-th wave of that head) -> chunked sums full[chunk][head][key*64+query]
constexpr int DB_CHUNKS = 8;
__global__ __launch_bounds__(256) void dbias_part_sum_kernel(const float* __restrict__ part, int nper, int nh, float* __restrict__ full) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nh * WT * WT) return;
    const int chunk = blockIdx.y;
    const int k0 = (int)((int64_t)nper * chunk / DB_CHUNKS), k1 = (int)((int64_t)nper * (chunk + 1) / DB_CHUNKS);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = k0;
    const int64_t stride = (int64_t)nh * WT * WT;
    for (; k + 4 <= k1; k += 4) {
        s0 += part[k * stride + i]; s1 += part[(k + 1) * stride + i]; s2 += part[(k + 2) * stride + i]; s3 += part[(k + 3) * stride + i];
    }
    for (; k < k1; ++k) s0 += part[k * stride + i];
    full[(int64_t)chunk * stride + i] = (s0 + s1) + (s2 + s3);
}
// table_grad[idx][head] (+)= sum over (query, key) pairs with relative index idx, over the chunked sums (key-major storage)
__global__ __launch_bounds__(64) void dbias_table_kq_kernel(const float* __restrict__ full, int nh, float* __restrict__ dtable, int accumulate) {
    // one wave per table entry: lane = (chunk, query row); fixed shuffle tree -> deterministic
    const int tdx = blockIdx.x;
    const int idx = tdx / nh, head = tdx - idx * nh;
    const int dy = idx / (2 * WS - 1) - (WS - 1), dx = idx % (2 * WS - 1) - (WS - 1);   // qy - ky, qx - kx
    const int chunk = threadIdx.x >> 3, qy = threadIdx.x & 7;
    const int ky = qy - dy;
    float s = 0.f;
    if (ky >= 0 && ky < WS) {
        const float* f = full + ((int64_t)chunk * nh + head) * (WT * WT);
        for (int qx = 0; qx < WS; ++qx) {
            const int kx = qx - dx;
            if (kx >= 0 && kx < WS) s += f[(ky * WS + kx) * WT + (qy * WS + qx)];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) dtable[tdx] = accumulate ? dtable[tdx] + s : s;
}

// workgroups (4 waves = 4 tasks in flight each): at most `cap` (one round of resident workgroups), a multiple of nh so that
// waves * 4 % nh == 0 and every wave keeps its head across its tasks
int grid_for_tasks(int ntasks, int nh, int cap) {
    int g = (ntasks + 3) / 4;
    if (g > cap) g = cap;
    g = g >= nh ? g / nh * nh : nh;
    return g;
}
constexpr int FWD_CAP = 512, BWD_CAP = 256;      // resident workgroups: 2 per CU forward, 1 per CU backward (register budget)

}  // namespace

namespace tr {

bool attn_mfma_ok(const AttnArgs& a, bool bwd) {
    auto ok = [&](const tr_view& v, int width) {
        // bf16, 4-byte aligned head slices, and the last 16-byte piece of the last head stays inside the row
        const int last = width - a.hd + (a.hd + 7) / 8 * 8;
        return v.ptr && v.dtype == a.qkv.dtype && v.cstride % 2 == 0 && v.coff % 2 == 0 && v.coff + last <= v.cstride && ((uintptr_t)v.ptr % 4) == 0;
    };
    if (a.hd % 2 != 0 || a.hd > 32 || a.nh > 64 || (a.qkv.dtype != TR_BF16 && a.qkv.dtype != TR_F16)) return false;
    if (!ok(a.qkv, 3 * a.c)) return false;
    if (!bwd) return ok(a.out, a.c);
    return ok(a.dout, a.c) && ok(a.dqkv, 3 * a.c);
}

int launch_window_attn_fwd_mfma(hipStream_t st, const AttnArgs& a) {
    const int ntasks = a.b * (a.h / WS) * (a.w / WS) * a.nh;
    const size_t lds = 4 * (size_t)(TILE + 1024);
    if (a.qkv.dtype == TR_BF16) hipLaunchKernelGGL(window_attn_fwd_mfma_kernel<TR_BF16>, dim3(grid_for_tasks(ntasks, a.nh, FWD_CAP)), dim3(256), lds, st, a, ntasks);
    else hipLaunchKernelGGL(window_attn_fwd_mfma_kernel<TR_F16>, dim3(grid_for_tasks(ntasks, a.nh, FWD_CAP)), dim3(256), lds, st, a, ntasks);
    return tr::launch_status("tr_window_attention(mfma)");
}

size_t window_attn_bwd_mfma_workspace_floats(int ntasks, int nh) {
    return ((size_t)grid_for_tasks(ntasks, nh, BWD_CAP) * 4 + (size_t)DB_CHUNKS * nh) * WT * WT;
}

int launch_window_attn_bwd_mfma(hipStream_t st, const AttnArgs& a, float* dtable, int accumulate) {
    const int ntasks = a.b * (a.h / WS) * (a.w / WS) * a.nh;
    const size_t lds = 4 * (size_t)(4 * TILE + 768 + 1024);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)window_attn_bwd_mfma_kernel<TR_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)window_attn_bwd_mfma_kernel<TR_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int grid = grid_for_tasks(ntasks, a.nh, BWD_CAP);
    float* part = dtable ? a.dbias_ws : nullptr;
    if (a.qkv.dtype == TR_BF16) hipLaunchKernelGGL(window_attn_bwd_mfma_kernel<TR_BF16>, dim3(grid), dim3(256), lds, st, a, ntasks, part);
    else hipLaunchKernelGGL(window_attn_bwd_mfma_kernel<TR_F16>, dim3(grid), dim3(256), lds, st, a, ntasks, part);
    int rc = tr::launch_status("tr_window_attention_bwd(mfma)");
    if (rc || !dtable) return rc;
    float* full = part + (size_t)grid * 4 * WT * WT;
    hipLaunchKernelGGL(dbias_part_sum_kernel, dim3((a.nh * WT * WT + 255) / 256, DB_CHUNKS), dim3(256), 0, st, part, grid * 4 / a.nh, a.nh, full);
    static_assert(DB_CHUNKS == 8, "dbias_table_kq_kernel maps 8 chunks x 8 query rows onto one wave");
    hipLaunchKernelGGL(dbias_table_kq_kernel, dim3(NTAB * a.nh), dim3(64), 0, st, full, a.nh, dtable, accumulate);
    return tr::launch_status("tr_window_attention_bwd(mfma, bias)");
}

}  // namespace tr
