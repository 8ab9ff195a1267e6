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
//     chunk the (4R+2) x 34 pixel halo and the 9 x CG x 64 B weight slab are staged in LDS
//     (16-byte slots XOR-swizzled by (row>>2)&3 so every ds_read_b128 of an MFMA operand is
//     bank-conflict free) and all nine taps are read out of the same halo image by address
//     offset -- im2col is never materialised;
//   * MFMA orientation D[cout][pixel] = W[cout][k] * X[k][pixel]: the accumulator holds, per
//     lane, one pixel and 16 output channels in groups of 4 consecutive channels, so the fused
//     epilogue (bias, LeakyReLU/ReLU, two scaled residuals, activation-derivative mask, dual
//     dtype stores) works on 8/16-byte NHWC vectors.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TW = 32;        // pixels along x per MFMA tile
constexpr int HALO_W = TW + 2;
constexpr int NTHREADS = 256;

struct ConvArgs {
    int n, h, w, hin, win;
    int cin, cout;
    int up;            // 0/1: nearest x2 upsample fused into the load
    int replicate;
    int act;
    float slope, scale1, scale2, rs1, rs2, mask_slope, out2_scale;
    tr_view in, out, out2, res1, res2, mask;
    const char* wpacked;
    const float* bias;
    int ntx, nty, nchunks;
    int vec_out, vec_out2, vec_res1, vec_res2, vec_mask;
};

template <int CT, int NT, int R>
__global__ __launch_bounds__(NTHREADS) void conv3x3_kernel(ConvArgs a) {
    constexpr int TH = 4 * R;
    constexpr int HALO_PIX = (TH + 2) * HALO_W;
    constexpr int HALO_ITEMS = HALO_PIX * 4;           // 16-byte slots
    constexpr int HALO_BYTES = HALO_PIX * 64;
    constexpr int CGW = NT * 32;
    constexpr int W_ITEMS = 9 * CGW * 4;
    constexpr int ES = CT == TR_BF16 ? 2 : 4;
    constexpr int EPS = 16 / ES;                       // elements per slot
    constexpr int CK = 64 / ES;                        // channels per chunk

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* hl = smem;
    char* wl = smem + HALO_BYTES;

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

    const char* in_base = (const char*)a.in.ptr;
    const int64_t in_pix_bytes = (int64_t)a.in.cstride * ES;

    for (int ch = 0; ch < a.nchunks; ++ch) {
        __syncthreads();
        // ---- stage the halo chunk --------------------------------------------------------
        for (int it = tid; it < HALO_ITEMS; it += NTHREADS) {
            const int pp = it >> 2, s = it & 3;
            const int py = pp / HALO_W, px = pp - py * HALO_W;
            int oy = y0 + py - 1, ox = x0 + px - 1;
            bool ok = (oy >= 0) & (oy < a.h) & (ox >= 0) & (ox < a.w);
            if (a.replicate) {
                oy = min(max(oy, 0), a.h - 1);
                ox = min(max(ox, 0), a.w - 1);
                ok = true;
            }
            const int c0 = ch * CK + s * EPS;
            ok = ok & (c0 < a.cin);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) {
                const int sy = oy >> a.up, sx = ox >> a.up;
                const int64_t pix = ((int64_t)nb * a.hin + sy) * a.win + sx;
                v = *(const uint4*)(in_base + pix * in_pix_bytes + (int64_t)(a.in.coff + c0) * ES);
            }
            *(uint4*)(hl + pp * 64 + ((s ^ ((pp >> 2) & 3)) << 4)) = v;
        }
        // ---- stage the weight slab (already swizzled by the packer) -----------------------
        {
            const char* wsrc = a.wpacked + ((int64_t)cg * a.nchunks + ch) * (W_ITEMS * 16);
            for (int it = tid; it < W_ITEMS; it += NTHREADS)
                *(uint4*)(wl + it * 16) = *(const uint4*)(wsrc + it * 16);
        }
        __syncthreads();
        if (!wave_active) continue;
        // ---- 9 taps x K-chunk of MFMAs ------------------------------------------------------
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int slot = q * 2 + lh;
                if constexpr (CT == TR_BF16) {
                    bf16x8 wa[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int row = tap * CGW + t * 32 + lr;
                        wa[t] = *(const bf16x8*)(wl + row * 64 + ((slot ^ ((row >> 2) & 3)) << 4));
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int pp = (wv * R + r + ky) * HALO_W + lr + kx;
                        const bf16x8 xb = *(const bf16x8*)(hl + pp * 64 + ((slot ^ ((pp >> 2) & 3)) << 4));
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[t], xb, acc[r][t], 0, 0, 0);
                    }
                } else {
                    f32x4 wa[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int row = tap * CGW + t * 32 + lr;
                        wa[t] = *(const f32x4*)(wl + row * 64 + ((slot ^ ((row >> 2) & 3)) << 4));
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int pp = (wv * R + r + ky) * HALO_W + lr + kx;
                        const f32x4 xb = *(const f32x4*)(hl + pp * 64 + ((slot ^ ((pp >> 2) & 3)) << 4));
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[t][e], xb[e], acc[r][t], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- fused epilogue -------------------------------------------------------------------------
    const int ox = x0 + lr;
    if (!wave_active || ox >= a.w) return;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int oy = y0 + wv * R + r;
        if (oy >= a.h) break;
        const int64_t pix = ((int64_t)nb * a.h + oy) * a.w + ox;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = cg * CGW + t * 32 + q * 8 + lh * 4;
                if (co >= a.cout) continue;
                const bool full = (co + 3) < a.cout;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[r][t][q * 4 + i];
                if (a.bias) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (co + i < a.cout) v[i] += a.bias[co + i];
                }
                if (a.act == TR_ACT_LRELU) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * a.slope;
                } else if (a.act == TR_ACT_RELU) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                }
                if (full) {
                    if (a.res1.ptr) {
                        const float4 rr = tr::view_load4(a.res1, pix, co, a.vec_res1);
                        v[0] = v[0] * a.scale1 + a.rs1 * rr.x; v[1] = v[1] * a.scale1 + a.rs1 * rr.y;
                        v[2] = v[2] * a.scale1 + a.rs1 * rr.z; v[3] = v[3] * a.scale1 + a.rs1 * rr.w;
                    }
                    if (a.res2.ptr) {
                        const float4 rr = tr::view_load4(a.res2, pix, co, a.vec_res2);
                        v[0] = v[0] * a.scale2 + a.rs2 * rr.x; v[1] = v[1] * a.scale2 + a.rs2 * rr.y;
                        v[2] = v[2] * a.scale2 + a.rs2 * rr.z; v[3] = v[3] * a.scale2 + a.rs2 * rr.w;
                    }
                    if (a.mask.ptr) {
                        const float4 mm = tr::view_load4(a.mask, pix, co, a.vec_mask);
                        v[0] *= mm.x > 0.f ? 1.f : a.mask_slope; v[1] *= mm.y > 0.f ? 1.f : a.mask_slope;
                        v[2] *= mm.z > 0.f ? 1.f : a.mask_slope; v[3] *= mm.w > 0.f ? 1.f : a.mask_slope;
                    }
                    tr::view_store4(a.out, pix, co, make_float4(v[0], v[1], v[2], v[3]), a.vec_out);
                    if (a.out2.ptr) {
                        const float s = a.out2_scale;
                        tr::view_store4(a.out2, pix, co, make_float4(v[0] * s, v[1] * s, v[2] * s, v[3] * s), a.vec_out2);
                    }
                } else {
                    for (int i = 0; i < 4 && co + i < a.cout; ++i) {
                        float x = v[i];
                        if (a.res1.ptr) x = x * a.scale1 + a.rs1 * tr::view_load(a.res1, pix, co + i);
                        if (a.res2.ptr) x = x * a.scale2 + a.rs2 * tr::view_load(a.res2, pix, co + i);
                        if (a.mask.ptr) x *= tr::view_load(a.mask, pix, co + i) > 0.f ? 1.f : a.mask_slope;
                        tr::view_store(a.out, pix, co + i, x);
                        if (a.out2.ptr) tr::view_store(a.out2, pix, co + i, x * a.out2_scale);
                    }
                }
            }
        }
    }
}

template <int CT, int NT, int R>
int launch(hipStream_t st, const ConvArgs& a, int ncg) {
    constexpr int TH = 4 * R;
    constexpr size_t lds = (size_t)(TH + 2) * HALO_W * 64 + (size_t)9 * NT * 32 * 64;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv3x3_kernel<CT, NT, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid((unsigned)(a.ntx * a.nty * a.n), (unsigned)ncg, 1);
    hipLaunchKernelGGL((conv3x3_kernel<CT, NT, R>), grid, dim3(NTHREADS), lds, st, a);
    return tr::launch_status("tr_conv3x3");
}

bool view_ok(const tr_view& v, int c_needed) {
    if (!v.ptr) return true;
    return (v.dtype == TR_F32 || v.dtype == TR_BF16) && v.cstride > 0 && v.coff >= 0 && v.coff + c_needed <= v.cstride &&
           v.c >= c_needed;
}

}  // namespace

extern "C" int tr_conv3x3(void* stream, const tr_conv3x3_desc* d) {
    TR_REQUIRE(d != nullptr, TR_EINVAL, "tr_conv3x3: null descriptor");
    TR_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, TR_EINVAL,
               "tr_conv3x3: bad dims n=%d h=%d w=%d cin=%d cout=%d", d->n, d->h, d->w, d->cin, d->cout);
    TR_REQUIRE(d->compute == TR_F32 || d->compute == TR_BF16, TR_EINVAL, "tr_conv3x3: bad compute dtype %d", d->compute);
    TR_REQUIRE(d->in.ptr && d->out.ptr && d->wpacked, TR_EINVAL, "tr_conv3x3: in/out/wpacked must be set");
    TR_REQUIRE(d->in.dtype == d->compute, TR_EUNSUPPORTED, "tr_conv3x3: input dtype %d must equal compute dtype %d",
               d->in.dtype, d->compute);
    const int eps = d->compute == TR_BF16 ? 8 : 4;
    TR_REQUIRE(d->cin % eps == 0, TR_EALIGN, "tr_conv3x3: cin=%d must be a multiple of %d (pad the view)", d->cin, eps);
    TR_REQUIRE(view_ok(d->in, d->cin), TR_EINVAL, "tr_conv3x3: input view does not hold cin=%d channels", d->cin);
    TR_REQUIRE(tr::view_vec16B_ok(d->in), TR_EALIGN, "tr_conv3x3: input view is not 16-byte sliceable");
    TR_REQUIRE(view_ok(d->out, d->cout) && view_ok(d->out2, d->cout) && view_ok(d->res1, d->cout) &&
                   view_ok(d->res2, d->cout) && view_ok(d->mask, d->cout),
               TR_EINVAL, "tr_conv3x3: an epilogue view does not hold cout=%d channels", d->cout);
    TR_REQUIRE(d->in_mode == TR_IN_DIRECT || (d->h % 2 == 0 && d->w % 2 == 0), TR_EINVAL,
               "tr_conv3x3: nearest-2x input needs even output dims");
    TR_REQUIRE(((uintptr_t)d->wpacked) % 16 == 0, TR_EALIGN, "tr_conv3x3: wpacked must be 16-byte aligned");

    ConvArgs a;
    a.n = d->n; a.h = d->h; a.w = d->w;
    a.up = d->in_mode == TR_IN_NEAREST2X ? 1 : 0;
    a.hin = d->h >> a.up; a.win = d->w >> a.up;
    a.cin = d->cin; a.cout = d->cout;
    a.replicate = d->pad_mode == TR_PAD_REPLICATE;
    a.act = d->act; a.slope = d->slope;
    a.scale1 = d->scale1; a.scale2 = d->scale2; a.rs1 = d->res1_scale; a.rs2 = d->res2_scale; a.mask_slope = d->mask_slope; a.out2_scale = d->out2_scale;
    a.in = d->in; a.out = d->out; a.out2 = d->out2; a.res1 = d->res1; a.res2 = d->res2; a.mask = d->mask;
    a.wpacked = (const char*)d->wpacked; a.bias = d->bias;
    const tr::PackGeom g = tr::pack_geom(d->cin, d->cout, d->compute);
    a.nchunks = g.nchunks;
    a.vec_out = tr::view_vec4_ok(d->out); a.vec_out2 = tr::view_vec4_ok(d->out2);
    a.vec_res1 = tr::view_vec4_ok(d->res1); a.vec_res2 = tr::view_vec4_ok(d->res2); a.vec_mask = tr::view_vec4_ok(d->mask);
    hipStream_t st = (hipStream_t)stream;
    constexpr int R = 2;
    a.ntx = (d->w + TW - 1) / TW;
    a.nty = (d->h + 4 * R - 1) / (4 * R);
    if (d->compute == TR_BF16) {
        if (g.cgw == 64) return launch<TR_BF16, 2, R>(st, a, g.ncg);
        return launch<TR_BF16, 1, R>(st, a, g.ncg);
    }
    if (g.cgw == 64) return launch<TR_F32, 2, R>(st, a, g.ncg);
    return launch<TR_F32, 1, R>(st, a, g.ncg);
}
