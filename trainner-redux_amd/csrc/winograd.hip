// Winograd F(2x2, 3x3) convolution in exact fp32 on MFMA, fully fused (input transform, 16 batched GEMMs, output transform, epilogue in
// one kernel) -- for the fp32-forced VGG19 perceptual head (reference: traiNNer/losses/perceptual_loss.py:8-12 casts the head's inputs to
// float32; perceptual_fp16_loss.py:262-329 is the conv / ReLU / pool stack), whose 3x3 convs are 55 % of the headline step at
// v_mfma_f32_32x32x2_f32's 157 TFLOP/s.  Y = A^T [ (G g G^T) (.) (B^T d B) ] A needs 16 multiplies per 2x2 outputs instead of 36: 2.25x
// fewer MFMA cycles.  A non-fused formulation would move 18x the tensor through HBM (the 16 transformed planes are 4x the activations,
// written and read on both sides of the GEMMs): here V = B^T d B is computed by the loader threads on the way into LDS and the 16
// products M_xi leave the accumulators only through LDS, so HBM sees the input once per output-channel group and the output once.
//
// Workgroup = 256 threads = 4 waves; tile = 64 output channels x 32 Winograd tiles (4 x 8 tiles = 8 x 16 output pixels).
// K loop over 8-channel chunks.  Per chunk:
//   * loader role (thread = (tile, channel)): 16 scalar loads of the 4 x 4 patch (8 lanes = 32 contiguous bytes of a pixel record), issued
//     one chunk ahead; B^T d B in registers (32 adds); the 16 values go to the 16 xi-planes of LDS in B-operand order [xi][kk][tile][s]
//     (channel c = 2 s + kk, the K pairing of v_mfma_f32_32x32x2_f32); the pre-transformed weights U (packed [group][chunk][xi][kk][k][s]
//     by tr_pack_winograd) are copied straight into LDS: 32 KB, 8 x 16 B per thread.
//   * MFMA role: wave j owns the four positions xi = (i, j), i = 0..3: per position 1 B fragment + 2 A fragments (ds_read_b128, conflict
//     free) feed 8 MFMAs; 32 MFMAs per wave and chunk.
// Epilogue: wave j folds its column in registers, T_j = A^T M[:, j] (2 x 16 adds on accumulator tiles of identical layout), the four
// waves exchange T through LDS ([j][r][tile][36-float rows]: b128 writes and b32 reads conflict free), and every thread finishes
// Y[r][:] = T[r] A for its (channel, tile) pairs with bias / ReLU / ReLU-derivative mask, storing 128 contiguous bytes per pixel and wave half.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int W_TB = 32;                      // tiles per workgroup (4 rows x 8 cols)
constexpr int W_KB = 64;                      // output channels per workgroup
constexpr int W_CK = 8;                       // input channels per chunk
constexpr int U_FLOATS = 16 * 2 * W_KB * 4;   // 8192 floats = 32 KB per (group, chunk)
constexpr int V_FLOATS = 16 * 2 * W_TB * 4;   // 4096 floats = 16 KB
constexpr int T_STRIDE = 36;                  // floats per (j, r, tile) row of the exchange buffer (32 + 4: b128-aligned, conflict free)
constexpr int SMEM_BYTES = (U_FLOATS + V_FLOATS) * 4;   // 48 KB (the exchange buffer, 4*2*32*36*4 = 36,864 B, reuses it)

struct WinoArgs {
    int n, h, w, cin, cout;
    int act;
    float mask_slope;
    tr_view in, out, mask;
    const float* u;
    const float* bias;
    int nty, ntx, nchunks, nkg;
};

__global__ __launch_bounds__(256, 2) void winograd_f32_kernel(WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ul = (float*)smem;
    float* vl = ul + U_FLOATS;
    const int tid = threadIdx.x;
    const int l = tid & 63, wv = tid >> 6, lr = l & 31, lh = l >> 5;
    int b = blockIdx.x;
    const int kg = b % a.nkg; b /= a.nkg;      // output-channel group fastest: neighbours in the grid share the input tile in L2
    const int bx = b % a.ntx; b /= a.ntx;
    const int by = b % a.nty;
    const int nb = b / a.nty;

    // ---- loader role: channel lc of the chunk, tile lt of the workgroup ---------------------------------------------------------------
    const int lc = tid & 7, lt = tid >> 3;
    const int py0 = 2 * (by * 4 + (lt >> 3)) - 1, px0 = 2 * (bx * 8 + (lt & 7)) - 1;     // top-left pixel of the 4 x 4 input patch (pad 1)
    unsigned okmask = 0;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int y = py0 + (p >> 2), x = px0 + (p & 3);
        if (y >= 0 && y < a.h && x >= 0 && x < a.w) okmask |= 1u << p;
    }
    const float* inb = (const float*)a.in.ptr + ((int64_t)nb * a.h * a.w) * a.in.cstride + a.in.coff + lc;
    const int rowp = a.w * a.in.cstride, colp = a.in.cstride;
    const int off0 = py0 * rowp + px0 * colp;                                            // may be negative: guarded by okmask
    const f32x4* ub = (const f32x4*)(a.u + (int64_t)kg * a.nchunks * U_FLOATS) + tid;
    float d[16];
    f32x4 ureg[8];
    auto issue = [&](int ch) {
        const float* src = inb + ch * W_CK;
#pragma unroll
        for (int p = 0; p < 16; ++p) d[p] = ((okmask >> p) & 1u) ? src[off0 + (p >> 2) * rowp + (p & 3) * colp] : 0.f;
        const f32x4* us = ub + (int64_t)ch * (U_FLOATS / 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) ureg[j] = us[j * 256];
    };
    float* vdst = vl + ((lc & 1) * W_TB + lt) * 4 + (lc >> 1);                           // + xi * 256 floats

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][kh][e] = 0.f;

    issue(0);
    for (int ch = 0; ch < a.nchunks; ++ch) {
        __syncthreads();                                   // every wave has consumed the previous chunk's operands
        {
            // V = B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
            float t[16];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                t[0 + x] = d[0 + x] - d[8 + x];
                t[4 + x] = d[4 + x] + d[8 + x];
                t[8 + x] = d[8 + x] - d[4 + x];
                t[12 + x] = d[4 + x] - d[12 + x];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                vdst[(i * 4 + 0) * 256] = t[i * 4 + 0] - t[i * 4 + 2];
                vdst[(i * 4 + 1) * 256] = t[i * 4 + 1] + t[i * 4 + 2];
                vdst[(i * 4 + 2) * 256] = t[i * 4 + 2] - t[i * 4 + 1];
                vdst[(i * 4 + 3) * 256] = t[i * 4 + 1] - t[i * 4 + 3];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ((f32x4*)ul)[tid + j * 256] = ureg[j];
        }
        __syncthreads();
        if (ch + 1 < a.nchunks) issue(ch + 1);             // next chunk in flight behind this chunk's MFMAs
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int xi = i * 4 + wv;
            const f32x4 bv = *(const f32x4*)(vl + xi * 256 + (lh * W_TB + lr) * 4);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const f32x4 au = *(const f32x4*)(ul + xi * 512 + (lh * W_KB + kh * 32 + lr) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][kh] = __builtin_amdgcn_mfma_f32_32x32x2f32(au[e], bv[e], acc[i][kh], 0, 0, 0);
            }
        }
    }

    // ---- output transform + epilogue, one 32-channel half at a time ------------------------------------------------------------------------
    float* tl = (float*)smem;
    const int ek = tid & 31, etg = tid >> 5;
    const float* ob = nullptr;
    (void)ob;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        __syncthreads();                                   // LDS free (operands of the last chunk / the previous half's exchange)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 t0, t1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float m0 = acc[0][kh][q * 4 + e], m1 = acc[1][kh][q * 4 + e], m2 = acc[2][kh][q * 4 + e], m3 = acc[3][kh][q * 4 + e];
                t0[e] = m0 + m1 + m2;                      // A^T = [1 1 1 0; 0 1 -1 -1] applied down column j = wv
                t1[e] = m1 - m2 - m3;
            }
            *(f32x4*)(tl + ((wv * 2 + 0) * W_TB + lr) * T_STRIDE + q * 8 + lh * 4) = t0;
            *(f32x4*)(tl + ((wv * 2 + 1) * W_TB + lr) * T_STRIDE + q * 8 + lh * 4) = t1;
        }
        __syncthreads();
        const int kc = kg * W_KB + kh * 32 + ek;
        const float bias = a.bias ? a.bias[kc] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = etg + 8 * i;
            const int ty = by * 4 + (t >> 3), tx = bx * 8 + (t & 7);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float T0 = tl[((0 * 2 + r) * W_TB + t) * T_STRIDE + ek], T1 = tl[((1 * 2 + r) * W_TB + t) * T_STRIDE + ek];
                const float T2 = tl[((2 * 2 + r) * W_TB + t) * T_STRIDE + ek], T3 = tl[((3 * 2 + r) * W_TB + t) * T_STRIDE + ek];
                const float y[2] = {T0 + T1 + T2, T1 - T2 - T3};
                const int oy = 2 * ty + r;
                if (oy >= a.h) continue;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int ox = 2 * tx + s;
                    if (ox >= a.w) continue;
                    const int64_t pix = ((int64_t)nb * a.h + oy) * a.w + ox;
                    float v = y[s] + bias;
                    if (a.act == TR_ACT_RELU) v = fmaxf(v, 0.f);
                    if (a.mask.ptr) v *= ((const float*)a.mask.ptr)[pix * a.mask.cstride + a.mask.coff + kc] > 0.f ? 1.f : a.mask_slope;
                    ((float*)a.out.ptr)[pix * a.out.cstride + a.out.coff + kc] = v;
                }
            }
        }
    }
}

// U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], for g = w[co][ci] (or, transposed, the input-gradient kernel w[ci][co] with its taps
// mirrored); written in the LDS image order of the kernel above: [co / 64][ci / 8][xi][kk = ci & 1][co % 64][s = (ci % 8) >> 1].
__global__ __launch_bounds__(256) void winograd_pack_kernel(const float* __restrict__ w, float* __restrict__ dst, int cout, int cin, int src_o, int src_i,
                                                             int transpose) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)cout * cin) return;
    const int co = (int)(i / cin), ci = (int)(i % cin);
    float g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = transpose ? w[((int64_t)ci * src_i + co) * 9 + (8 - t)] : w[((int64_t)co * src_i + ci) * 9 + t];
    (void)src_o;
    float gg[12];                                           // G g : 4 x 3
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        gg[0 + x] = g[0 + x];
        gg[3 + x] = 0.5f * (g[0 + x] + g[3 + x] + g[6 + x]);
        gg[6 + x] = 0.5f * (g[0 + x] - g[3 + x] + g[6 + x]);
        gg[9 + x] = g[6 + x];
    }
    const int nchunks = cin / W_CK;
    float* base = dst + ((int64_t)(co / W_KB) * nchunks + ci / W_CK) * U_FLOATS + (((ci & 1) * W_KB) + (co % W_KB)) * 4 + ((ci % W_CK) >> 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float a0 = gg[r * 3 + 0], a1 = gg[r * 3 + 1], a2 = gg[r * 3 + 2];
        base[(r * 4 + 0) * 512] = a0;
        base[(r * 4 + 1) * 512] = 0.5f * (a0 + a1 + a2);
        base[(r * 4 + 2) * 512] = 0.5f * (a0 - a1 + a2);
        base[(r * 4 + 3) * 512] = a2;
    }
}

inline bool wino_view_ok(const tr_view& v, int c) { return v.ptr && v.dtype == TR_F32 && v.c >= c && v.coff >= 0 && v.coff + c <= v.cstride; }

}  // namespace

extern "C" size_t tr_winograd_packed_bytes(int cin, int cout) { return (size_t)cin * cout * 16 * sizeof(float); }

extern "C" int tr_pack_winograd(void* stream, const float* w_oihw, float* dst, int cout, int cin, int transpose) {
    TR_REQUIRE(w_oihw && dst && cout > 0 && cin > 0 && cout % W_KB == 0 && cin % W_CK == 0, TR_EINVAL,
               "tr_pack_winograd: cout=%d must be a multiple of %d and cin=%d of %d", cout, W_KB, cin, W_CK);
    TR_REQUIRE(((uintptr_t)dst) % 16 == 0, TR_EALIGN, "tr_pack_winograd: dst must be 16-byte aligned");
    const int64_t n = (int64_t)cout * cin;
    // the OIHW source is cout x cin (direct) or cin x cout (transpose: the conv being differentiated maps cout-many inputs ... see header)
    hipLaunchKernelGGL(winograd_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, dst, cout, cin,
                       transpose ? cin : cout, transpose ? cout : cin, transpose ? 1 : 0);
    return tr::launch_status("tr_pack_winograd");
}

extern "C" int tr_conv3x3_winograd_f32(void* stream, const tr_conv3x3_desc* d) {
    TR_REQUIRE(d != nullptr, TR_EINVAL, "tr_conv3x3_winograd_f32: null descriptor");
    TR_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->h % 2 == 0 && d->w % 2 == 0, TR_EUNSUPPORTED, "tr_conv3x3_winograd_f32: needs even output dims (got %dx%d)",
               d->h, d->w);
    TR_REQUIRE(d->cin > 0 && d->cin % W_CK == 0 && d->cout > 0 && d->cout % W_KB == 0, TR_EUNSUPPORTED,
               "tr_conv3x3_winograd_f32: cin=%d must be a multiple of %d and cout=%d of %d", d->cin, W_CK, d->cout, W_KB);
    TR_REQUIRE(d->ksize == 3 && d->in_mode == TR_IN_DIRECT && d->pad_mode == TR_PAD_ZEROS, TR_EUNSUPPORTED,
               "tr_conv3x3_winograd_f32: 3x3, direct input, zero padding only");
    TR_REQUIRE(d->act == TR_ACT_NONE || d->act == TR_ACT_RELU, TR_EUNSUPPORTED, "tr_conv3x3_winograd_f32: activation %d is not built (none, ReLU)", d->act);
    TR_REQUIRE(!d->res1.ptr && !d->res2.ptr && !d->out2.ptr && !d->sample_scale && d->mask_mode == 0, TR_EUNSUPPORTED,
               "tr_conv3x3_winograd_f32: residual / second output / per-sample scale / activation-derivative masks are not built");
    TR_REQUIRE(wino_view_ok(d->in, d->cin) && wino_view_ok(d->out, d->cout) && (!d->mask.ptr || wino_view_ok(d->mask, d->cout)) && d->wpacked, TR_EINVAL,
               "tr_conv3x3_winograd_f32: fp32 views holding cin / cout channels and packed weights are required");
    TR_REQUIRE(((uintptr_t)d->wpacked) % 16 == 0, TR_EALIGN, "tr_conv3x3_winograd_f32: wpacked must be 16-byte aligned");
    TR_REQUIRE((int64_t)d->h * d->w * d->in.cstride < (int64_t)0x7fffffff, TR_EUNSUPPORTED, "tr_conv3x3_winograd_f32: one image of the input must stay below 2^31 elements");
    WinoArgs a;
    a.n = d->n; a.h = d->h; a.w = d->w; a.cin = d->cin; a.cout = d->cout;
    a.act = d->act; a.mask_slope = d->mask_slope;
    a.in = d->in; a.out = d->out; a.mask = d->mask;
    a.u = (const float*)d->wpacked; a.bias = d->bias;
    a.nty = (d->h / 2 + 3) / 4; a.ntx = (d->w / 2 + 7) / 8;
    a.nchunks = d->cin / W_CK; a.nkg = d->cout / W_KB;
    const int64_t nblk = (int64_t)d->n * a.nty * a.ntx * a.nkg;
    TR_REQUIRE(nblk < (int64_t)0x7fffffff, TR_EUNSUPPORTED, "tr_conv3x3_winograd_f32: grid too large");
    hipLaunchKernelGGL(winograd_f32_kernel, dim3((unsigned)nblk), dim3(256), SMEM_BYTES, (hipStream_t)stream, a);
    return tr::launch_status("tr_conv3x3_winograd_f32");
}
