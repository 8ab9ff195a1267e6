// MoSR GatedCNNBlock pieces that are not dense contractions: depthwise k x k convolution (forward / input gradient / weight gradient),
// the Mish gate and the Mish derivative sweep.  All of them are stencil / element-wise work on NHWC channel slices (HBM / VALU bound).
// Contract: include/trainner_hip.h.  Reference arithmetic replaced (spandrel's MoSR is absent from /root/reference; the block is the
// one of its in-tree twins):
//   g, i, c = split(fc1(norm(x)))            traiNNer/archs/mosrv2_arch.py:121-122, moesr_arch.py:101-102
//   c = conv(c)        (depthwise, MoSR v1: nn.Conv2d(cc, cc, k, 1, k // 2, groups=cc))
//   act(g) * cat(i, c)                       mosrv2_arch.py:124, moesr_arch.py:104
#include "common.h"

namespace {

constexpr int DW_TH = 16;      // output rows per workgroup strip
constexpr int DW_TW = 32;      // output columns per workgroup
constexpr int DW_CB = 32;      // channels per workgroup (8 lanes x 4)

struct DwGeom {
    int nstrips, ncolblk, ncb;
    int64_t nsp;               // spatial workgroups: n * nstrips * ncolblk
};
inline DwGeom dw_geom(int n, int h, int w, int c) {
    DwGeom g;
    g.nstrips = (h + DW_TH - 1) / DW_TH;
    g.ncolblk = (w + DW_TW - 1) / DW_TW;
    g.ncb = (c + DW_CB - 1) / DW_CB;
    g.nsp = (int64_t)n * g.nstrips * g.ncolblk;
    return g;
}

__device__ __forceinline__ void fma4(float4& a, const float4& w, const float4& v) {
    a.x = fmaf(w.x, v.x, a.x); a.y = fmaf(w.y, v.y, a.y); a.z = fmaf(w.z, v.z, a.z); a.w = fmaf(w.w, v.w, a.w);
}

// Branch-free operand fetch: the raw 4-channel load is issued unconditionally (from a clamped address) so that a whole column of loads
// is in flight at once; tr::view_load4 switches on the runtime dtype, which serialises every load behind the previous one's use
// (measured: 149 us instead of 40 for 64 channels at 64 x 64 x 64).  H16 = the view holds bf16 / fp16 (8-byte loads), else fp32.
template <bool H16> struct Raw4 { typedef float4 T; };
template <> struct Raw4<true> { typedef uint2 T; };
// Addressing: a uniform 64-bit image base (scalar registers) plus a 32-bit byte offset per lane = uniform row offset + the lane's column
// offset, i.e. one 32-bit add per load.  (Indexing through 64-bit pixel numbers cost four quarter-rate integer multiplies per load:
// 570 of them against 1568 packed FMAs in the 7 x 7 kernel.)  One image of a view must therefore stay below 2 GiB.
template <bool H16>
__device__ __forceinline__ const char* image_base(const tr_view& v, int64_t nb, int h, int w) {
    return (const char*)v.ptr + ((int64_t)nb * h * w * v.cstride + v.coff) * (H16 ? 2 : 4);
}
template <bool H16>
__device__ __forceinline__ typename Raw4<H16>::T raw_load4(const char* base, uint32_t off) {
    return *(const typename Raw4<H16>::T*)(base + off);
}
__device__ __forceinline__ float4 raw_cvt4(const float4& r, int, bool valid) { return valid ? r : make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 raw_cvt4(const uint2& u, int dtype, bool valid) {
    float4 r;
    if (dtype == TR_BF16) {
        r.x = __uint_as_float(u.x << 16); r.y = __uint_as_float(u.x & 0xffff0000u);
        r.z = __uint_as_float(u.y << 16); r.w = __uint_as_float(u.y & 0xffff0000u);
    } else {
        r.x = tr::f16_to_f32(u.x & 0xffffu); r.y = tr::f16_to_f32(u.x >> 16); r.z = tr::f16_to_f32(u.y & 0xffffu); r.w = tr::f16_to_f32(u.y >> 16);
    }
    return valid ? r : make_float4(0.f, 0.f, 0.f, 0.f);
}

// One thread = one output column x one 4-channel quad, walking down a strip of DW_TH rows: for a fixed kx the K + DW_TH - 1 input values
// of that column are loaded once each and feed up to K output rows from registers, so the kernel issues (K + TH - 1) * K loads for
// TH * K * K float4 FMAs.  Lanes of a wave cover 8 columns x 32 contiguous channels (64 B of 16-bit data per pixel record).
// Weights [c][K][K] (the OIHW tensor of a groups = c conv, I = 1) are staged per workgroup in LDS as [tap][channel]; `flip` mirrors
// the taps, which turns the forward into the input gradient.  The column of the NEXT kx is in flight while the current one is
// consumed (16-bit views: two register stages; fp32 views have no room for the second stage).  Padding: out-of-range columns zero the
// kx weights of that lane, out-of-range rows (uniform per workgroup) skip their FMAs -- no per-load selects.
template <int K, bool H16>
__global__ __launch_bounds__(256) void dwconv_kernel(tr_view in, tr_view out, const float* __restrict__ wt, const float* __restrict__ bias, int h, int w, int c,
                                                      int flip, int nstrips, int ncolblk, int ncb, int vin, int vout) {
    constexpr int P = K / 2;
    constexpr int NR = DW_TH + K - 1;
    constexpr int NST = H16 ? 2 : 1;
    typedef typename Raw4<H16>::T RT;
    __shared__ float wl[K * K][DW_CB];
    int b = blockIdx.x;
    const int cb = b % ncb; b /= ncb;
    const int xb = b % ncolblk; b /= ncolblk;
    const int sb = b % nstrips;
    const int nb = b / nstrips;
    for (int i = threadIdx.x; i < K * K * DW_CB; i += 256) {
        const int ch = i % DW_CB, tap = i / DW_CB;
        const int cc = cb * DW_CB + ch;
        wl[tap][ch] = cc < c ? wt[(int64_t)cc * K * K + (flip ? K * K - 1 - tap : tap)] : 0.f;
    }
    __syncthreads();
    const int q = threadIdx.x & 7, col = threadIdx.x >> 3;
    const int c0 = cb * DW_CB + q * 4;
    const int ox = xb * DW_TW + col;
    const int y0 = sb * DW_TH;
    if (c0 >= c || ox >= w) return;
    float4 acc[DW_TH];
    {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) b4 = make_float4(bias[c0], bias[c0 + 1], bias[c0 + 2], bias[c0 + 3]);
#pragma unroll
        for (int o = 0; o < DW_TH; ++o) acc[o] = b4;
    }
    const int64_t img = (int64_t)nb * h;
    constexpr uint32_t ES = H16 ? 2 : 4;
    const char* ibase = image_base<H16>(in, nb, h, w);
    const uint32_t rowpitch = (uint32_t)w * in.cstride * ES;
    RT raw[NST][NR];
    auto fetch = [&](int kx, RT (&dst)[NR]) {
        const uint32_t coloff = ((uint32_t)min(max(ox + kx - P, 0), w - 1) * in.cstride + c0) * ES;
#pragma unroll
        for (int r = 0; r < NR; ++r) dst[r] = raw_load4<H16>(ibase, (uint32_t)min(max(y0 + r - P, 0), h - 1) * rowpitch + coloff);
    };
    auto consume = [&](int kx, const RT (&src)[NR]) {
        const int ix = ox + kx - P;
        const bool xin = ix >= 0 && ix < w;
        float4 wc[K];
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            wc[ky] = *(const float4*)&wl[ky * K + kx][q * 4];
            if (!xin) wc[ky] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int iy = y0 + r - P;
            if (iy < 0 || iy >= h) continue;          // uniform over the workgroup
            const float4 v = raw_cvt4(src[r], in.dtype, true);
#pragma unroll
            for (int o = 0; o < DW_TH; ++o) {
                const int ky = r - o;
                if (ky >= 0 && ky < K) fma4(acc[o], wc[ky], v);
            }
        }
    };
    if constexpr (H16) {
        fetch(0, raw[0]);
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            if (kx + 1 < K) fetch(kx + 1, raw[(kx + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            consume(kx, raw[kx & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // kx stays a rolled loop: unrolled, the scheduler hoists all NR * K loads to the top and spills
#pragma unroll 1
        for (int kx = 0; kx < K; ++kx) {
            fetch(kx, raw[0]);
            consume(kx, raw[0]);
        }
    }
#pragma unroll
    for (int o = 0; o < DW_TH; ++o) {
        const int oy = y0 + o;
        if (oy < h) tr::view_store4(out, (img + oy) * w + ox, c0, acc[o], vout);
    }
}

// 16-bit views, LDS-staged: the (TH + K - 1) x (TW + K - 1) halo of the workgroup's 32 channels is loaded ONCE with 16-byte pieces (zeros where
// the image ends, so the inner loop carries no padding logic) and the K * (TH + K - 1) column reads of every thread come from LDS
// (ds_read_b64, 512 contiguous bytes per wave).  The direct-from-global version above re-fetched every input value K times through L1 / L2
// (63 us for 64 channels at 64 x 64 x 64, neither VALU- nor HBM-bound); fp32 views keep it (their halo does not fit two workgroups per CU).
template <int K>
__global__ __launch_bounds__(256) void dwconv_lds_kernel(tr_view in, tr_view out, const float* __restrict__ wt, const float* __restrict__ bias, int h, int w, int c,
                                                          int flip, int nstrips, int ncolblk, int ncb, int vout) {
    constexpr int P = K / 2;
    constexpr int NR = DW_TH + K - 1, NC = DW_TW + K - 1;
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    uint2* hl = (uint2*)dsm;                                  // [NR][NC][8] x 4 channels
    float* wl = (float*)(dsm + NR * NC * 64);                 // [K*K][32]
    int b = blockIdx.x;
    const int cb = b % ncb; b /= ncb;
    const int xb = b % ncolblk; b /= ncolblk;
    const int sb = b % nstrips;
    const int nb = b / nstrips;
    const int y0 = sb * DW_TH, x0 = xb * DW_TW;
    for (int i = threadIdx.x; i < K * K * DW_CB; i += 256) {
        const int ch = i % DW_CB, tap = i / DW_CB;
        const int cc = cb * DW_CB + ch;
        wl[i] = cc < c ? wt[(int64_t)cc * K * K + (flip ? K * K - 1 - tap : tap)] : 0.f;
    }
    {
        const char* ibase = image_base<true>(in, nb, h, w) + (size_t)cb * DW_CB * 2;
        const uint32_t rowpitch = (uint32_t)w * in.cstride * 2, colpitch = (uint32_t)in.cstride * 2;
        constexpr int NPIECE = NR * NC * 4;
        constexpr int NIT = (NPIECE + 255) / 256;
        uint4 piece[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pi = threadIdx.x + it * 256;
            const int pix = pi >> 2, part = pi & 3;
            const int r = pix / NC, cc = pix - r * NC;
            const int iy = y0 + r - P, ix = x0 + cc - P;
            const int ch0 = cb * DW_CB + part * 8;
            const bool ok = pi < NPIECE && iy >= 0 && iy < h && ix >= 0 && ix < w && ch0 < c;
            const char* src = ibase + (uint32_t)min(max(iy, 0), h - 1) * rowpitch + (uint32_t)min(max(ix, 0), w - 1) * colpitch + part * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) {
                if (ch0 + 8 <= c) v = *(const uint4*)src;
                else { const uint2 lo = *(const uint2*)src; v.x = lo.x; v.y = lo.y; }      // a 4-channel tail (c % 8 == 4)
            }
            piece[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pi = threadIdx.x + it * 256;
            if (pi < NPIECE) ((uint4*)hl)[pi] = piece[it];
        }
    }
    __syncthreads();
    const int q = threadIdx.x & 7, col = threadIdx.x >> 3;
    const int c0 = cb * DW_CB + q * 4;
    const int ox = x0 + col;
    if (c0 >= c || ox >= w) return;
    float4 acc[DW_TH];
    {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) b4 = make_float4(bias[c0], bias[c0 + 1], bias[c0 + 2], bias[c0 + 3]);
#pragma unroll
        for (int o = 0; o < DW_TH; ++o) acc[o] = b4;
    }
#pragma unroll 1
    for (int kx = 0; kx < K; ++kx) {
        float4 wc[K];
#pragma unroll
        for (int ky = 0; ky < K; ++ky) wc[ky] = *(const float4*)&wl[(ky * K + kx) * DW_CB + q * 4];
        const uint2* colp = hl + (col + kx) * 8 + q;
        uint2 raw[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) raw[r] = colp[r * NC * 8];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float4 v = raw_cvt4(raw[r], in.dtype, true);
#pragma unroll
            for (int o = 0; o < DW_TH; ++o) {
                const int ky = r - o;
                if (ky >= 0 && ky < K) fma4(acc[o], wc[ky], v);
            }
        }
    }
    const int64_t img = (int64_t)nb * h;
#pragma unroll
    for (int o = 0; o < DW_TH; ++o) {
        const int oy = y0 + o;
        if (oy < h) tr::view_store4(out, (img + oy) * w + ox, c0, acc[o], vout);
    }
}

// Weight gradient: dW[c][ky][kx] = sum_p gy[p][c] * x[p + (ky - P, kx - P)][c].  A thread keeps the DW_TH gradient values of its
// (column, quad) in registers and, for each kx in turn, walks the K + TH - 1 input rows of the column shifted by kx - P accumulating the K
// vertical taps (the next kx's column is in flight meanwhile).  After each kx the 8 columns of a wave are summed through a private LDS
// strip by 56 of its lanes (fixed order), which keep the (quad, ky) sums of all kx in registers; at the end the 4 waves are added and the
// workgroup writes ONE partial row [spatial workgroup][channel][K*K + 1] (last slot: bias gradient).
template <int K, bool H16>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(tr_view in, tr_view gy, float* __restrict__ partials, int h, int w, int c, int nstrips, int ncolblk,
                                                            int ncb, int vin, int vgy) {
    constexpr int P = K / 2;
    constexpr int NS = K * K + 1;
    constexpr int NR = DW_TH + K - 1;
    constexpr int NST = H16 ? 2 : 1;
    typedef typename Raw4<H16>::T RT;
    __shared__ float4 strip[4][8][K + 1][8];          // [wave][quad][ky (+ bias)][column of the wave]
    __shared__ float4 wsuml[4][8 * (K + 1)][K];       // [wave][(quad, ky)][kx]
    int b = blockIdx.x;
    const int cb = b % ncb; b /= ncb;
    const int sp = b;                            // spatial workgroup index (image, strip, column block)
    const int xb = b % ncolblk; b /= ncolblk;
    const int sb = b % nstrips;
    const int nb = b / nstrips;
    const int q = threadIdx.x & 7, col = threadIdx.x >> 3;
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, wcol = l >> 3;
    const int c0 = cb * DW_CB + q * 4;
    const int ox = xb * DW_TW + col;
    const int y0 = sb * DW_TH;
    const int64_t img = (int64_t)nb * h;
    const bool active = c0 < c && ox < w;
    const int oxc = min(ox, w - 1), c0c = active ? c0 : 0;
    (void)img;
    float4 g[DW_TH];
    float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr uint32_t ES = H16 ? 2 : 4;
    const char* ibase = image_base<H16>(in, nb, h, w);
    const char* gbase = image_base<H16>(gy, nb, h, w);
    const uint32_t rowpitch = (uint32_t)w * in.cstride * ES, growpitch = (uint32_t)w * gy.cstride * ES;
    {
        RT graw[DW_TH];
        const uint32_t gcol = ((uint32_t)oxc * gy.cstride + c0c) * ES;
#pragma unroll
        for (int o = 0; o < DW_TH; ++o) graw[o] = raw_load4<H16>(gbase, (uint32_t)min(y0 + o, h - 1) * growpitch + gcol);
#pragma unroll
        for (int o = 0; o < DW_TH; ++o) {
            g[o] = raw_cvt4(graw[o], gy.dtype, active && y0 + o < h);
            gsum.x += g[o].x; gsum.y += g[o].y; gsum.z += g[o].z; gsum.w += g[o].w;
        }
    }
    RT raw[NST][NR];
    auto fetch = [&](int kx, RT (&dst)[NR]) {
        const uint32_t coloff = ((uint32_t)min(max(ox + kx - P, 0), w - 1) * in.cstride + c0c) * ES;
#pragma unroll
        for (int r = 0; r < NR; ++r) dst[r] = raw_load4<H16>(ibase, (uint32_t)min(max(y0 + r - P, 0), h - 1) * rowpitch + coloff);
    };
    float4 wsum[K];                              // lanes < 8 * (K + 1): sums over the wave's 8 columns for (quad = l & 7, ky = l >> 3), per kx
    auto consume = [&](int kx, const RT (&src)[NR]) {
        const int ix = ox + kx - P;
        const bool xin = active && ix >= 0 && ix < w;
        float4 aw[K];
#pragma unroll
        for (int k = 0; k < K; ++k) aw[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int iy = y0 + r - P;
            if (iy < 0 || iy >= h) continue;          // uniform over the workgroup
            const float4 v = raw_cvt4(src[r], in.dtype, true);
#pragma unroll
            for (int o = 0; o < DW_TH; ++o) {
                const int ky = r - o;
                if (ky >= 0 && ky < K) fma4(aw[ky], g[o], v);
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) strip[wv][q][k][wcol] = xin ? aw[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (kx == P) strip[wv][q][K][wcol] = gsum;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (l < 8 * (K + 1)) {
            const int qq = l & 7, k = l >> 3;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < K || kx == P) {
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    const float4 t = strip[wv][qq][k][cc];
                    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
                }
            }
            wsum[kx] = s;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    if constexpr (H16) {
        fetch(0, raw[0]);
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            if (kx + 1 < K) fetch(kx + 1, raw[(kx + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            consume(kx, raw[kx & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            fetch(kx, raw[0]);
            __builtin_amdgcn_sched_barrier(0);
            consume(kx, raw[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (l < 8 * (K + 1)) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx) wsuml[wv][l][kx] = wsum[kx];
    }
    __syncthreads();
    // 8 quads x (K taps + bias) x K kx items; item (qq, k, kx): k < K -> slot k * K + kx ; k == K -> bias slot (kx == P only)
    for (int it = threadIdx.x; it < 8 * (K + 1) * K; it += 256) {
        const int kx = it % K, lk = it / K;
        const int qq = lk & 7, k = lk >> 3;
        if (k == K && kx != P) continue;
        float4 s = wsuml[0][lk][kx];
#pragma unroll
        for (int v2 = 1; v2 < 4; ++v2) {
            const float4 t = wsuml[v2][lk][kx];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const int ch = cb * DW_CB + qq * 4;
        if (ch < c) {
            const int slot = k < K ? k * K + kx : K * K;
            float* dst = partials + ((int64_t)sp * c + ch) * NS + slot;
            dst[0] = s.x; dst[NS] = s.y; dst[2 * NS] = s.z; dst[3 * NS] = s.w;
        }
    }
}

// dw[c][tap] (+)= sum over the spatial partial rows, in a fixed order: one wave per (channel, slot), lanes stride the rows and a
// butterfly adds them (same pattern every run, so the result is bitwise reproducible).
__global__ __launch_bounds__(256) void dwconv_wgrad_finish_kernel(const float* __restrict__ partials, int64_t nsp, int c, int ns, float* __restrict__ dw,
                                                                   float* __restrict__ db, int accumulate) {
    const int l = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= c * ns) return;
    const int ch = item / ns, slot = item - ch * ns;
    float s = 0.f;
    for (int64_t r = l; r < nsp; r += 64) s += partials[(r * c + ch) * ns + slot];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (l == 0) {
        if (slot < ns - 1) {
            float* d = dw + (int64_t)ch * (ns - 1) + slot;
            *d = accumulate ? *d + s : s;
        } else if (db) {
            db[ch] = accumulate ? db[ch] + s : s;
        }
    }
}

// ---- Mish gate: out[p][j] = mish(g[p][j]) * u[p][j], u = cat(a, b) along channels ------------------------------------------------------
__global__ __launch_bounds__(256) void gate_mish_kernel(tr_view g, tr_view a, tr_view b, tr_view out, int64_t npix, int c4, int ca, int vg, int va, int vb,
                                                         int vo) {
    const int64_t total = npix * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / c4;
        const int ch = (int)(i - p * c4) * 4;
        const float4 gg = tr::view_load4(g, p, ch, vg);
        const float4 u = ch < ca ? tr::view_load4(a, p, ch, va) : tr::view_load4(b, p, ch - ca, vb);
        tr::view_store4(out, p, ch, make_float4(tr::mish(gg.x) * u.x, tr::mish(gg.y) * u.y, tr::mish(gg.z) * u.z, tr::mish(gg.w) * u.w), vo);
    }
}

// given d = d/d(out): dg = d * u * mish'(g) ; du = d * mish(g), written to the a / b parts of the gradient views
__global__ __launch_bounds__(256) void gate_mish_bwd_kernel(tr_view d, tr_view g, tr_view a, tr_view b, tr_view dg, tr_view da, tr_view db, int64_t npix,
                                                             int c4, int ca, int vd, int vg, int va, int vb, int vdg, int vda, int vdb) {
    const int64_t total = npix * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / c4;
        const int ch = (int)(i - p * c4) * 4;
        const float4 dd = tr::view_load4(d, p, ch, vd), gg = tr::view_load4(g, p, ch, vg);
        const float4 u = ch < ca ? tr::view_load4(a, p, ch, va) : tr::view_load4(b, p, ch - ca, vb);
        const float4 r = make_float4(dd.x * u.x * tr::mish_grad(gg.x), dd.y * u.y * tr::mish_grad(gg.y), dd.z * u.z * tr::mish_grad(gg.z),
                                     dd.w * u.w * tr::mish_grad(gg.w));
        const float4 s = make_float4(dd.x * tr::mish(gg.x), dd.y * tr::mish(gg.y), dd.z * tr::mish(gg.z), dd.w * tr::mish(gg.w));
        tr::view_store4(dg, p, ch, r, vdg);
        if (ch < ca) tr::view_store4(da, p, ch, s, vda);
        else tr::view_store4(db, p, ch - ca, s, vdb);
    }
}

// out = g * s[p / pix_per_sample] * act'(pre)
__global__ __launch_bounds__(256) void act_grad_kernel(tr_view g, tr_view pre, tr_view out, int act, int64_t npix, int c4, int vg, int vp, int vo,
                                                        const float* __restrict__ sscale, int64_t pps) {
    const int64_t total = npix * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / c4;
        const int ch = (int)(i - p * c4) * 4;
        const float4 gg = tr::view_load4(g, p, ch, vg), m = tr::view_load4(pre, p, ch, vp);
        float mv[4] = {m.x, m.y, m.z, m.w}, d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (act == TR_ACT_MISH) d[k] = tr::mish_grad(mv[k]);
            else {   // TR_ACT_SILU
                const float sg = 1.f / (1.f + __expf(-mv[k]));
                d[k] = sg * (1.f + mv[k] * (1.f - sg));
            }
        }
        const float ss = sscale ? sscale[p / pps] : 1.f;
        tr::view_store4(out, p, ch, make_float4(gg.x * ss * d[0], gg.y * ss * d[1], gg.z * ss * d[2], gg.w * ss * d[3]), vo);
    }
}

inline bool view_has(const tr_view& v, int c) { return v.ptr && v.c >= c && v.coff >= 0 && v.coff + c <= v.cstride; }

inline unsigned sweep_grid(int64_t items) {
    int64_t b = (items + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int tr_dwconv(void* stream, tr_view in, tr_view out, const float* weight, const float* bias, int n, int h, int w, int c, int ksize, int flip) {
    TR_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, TR_EINVAL, "tr_dwconv: bad dims n=%d h=%d w=%d c=%d (c must be a multiple of 4)", n, h, w, c);
    TR_REQUIRE(ksize == 3 || ksize == 5 || ksize == 7, TR_EUNSUPPORTED, "tr_dwconv: kernel size %d is not built (3, 5, 7)", ksize);
    TR_REQUIRE(view_has(in, c) && view_has(out, c) && weight, TR_EINVAL, "tr_dwconv: in / out views must hold c=%d channels and weight must be set", c);
    TR_REQUIRE(tr::view_vec4_ok(in), TR_EALIGN, "tr_dwconv: the input view must allow aligned 4-channel access (offset / stride multiples of 4)");
    TR_REQUIRE((int64_t)h * w * in.cstride * tr::esize(in.dtype) < (int64_t)0x7fffff00, TR_EUNSUPPORTED, "tr_dwconv: one image of the input view must stay below 2 GiB");
    const DwGeom g = dw_geom(n, h, w, c);
    TR_REQUIRE(g.nsp * g.ncb < (int64_t)0x7fffffff, TR_EUNSUPPORTED, "tr_dwconv: grid too large");
    const dim3 grid((unsigned)(g.nsp * g.ncb));
    hipStream_t st = (hipStream_t)stream;
    const int vin = tr::view_vec4_ok(in), vout = tr::view_vec4_ok(out);
#define TR_DW(K, H) hipLaunchKernelGGL((dwconv_kernel<K, H>), grid, dim3(256), 0, st, in, out, weight, bias, h, w, c, flip ? 1 : 0, g.nstrips, g.ncolblk, g.ncb, vin, vout)
    const bool lds_ok = in.dtype != TR_F32 && in.cstride % 8 == 0 && in.coff % 8 == 0 && ((uintptr_t)in.ptr) % 16 == 0;
#define TR_DWL(K) hipLaunchKernelGGL(dwconv_lds_kernel<K>, grid, dim3(256), (size_t)((DW_TH + K - 1) * (DW_TW + K - 1) * 64 + K * K * DW_CB * 4), st, in, out, weight, \
                                     bias, h, w, c, flip ? 1 : 0, g.nstrips, g.ncolblk, g.ncb, vout)
    if (lds_ok) { if (ksize == 7) TR_DWL(7); else if (ksize == 5) TR_DWL(5); else TR_DWL(3); }
    else if (in.dtype == TR_F32) { if (ksize == 7) TR_DW(7, false); else if (ksize == 5) TR_DW(5, false); else TR_DW(3, false); }
    else { if (ksize == 7) TR_DW(7, true); else if (ksize == 5) TR_DW(5, true); else TR_DW(3, true); }
#undef TR_DWL
#undef TR_DW
    return tr::launch_status("tr_dwconv");
}

extern "C" size_t tr_dwconv_wgrad_workspace_bytes(int n, int h, int w, int c, int ksize) {
    const DwGeom g = dw_geom(n, h, w, c);
    return (size_t)g.nsp * c * (ksize * ksize + 1) * sizeof(float);
}

extern "C" int tr_dwconv_wgrad(void* stream, tr_view in, tr_view gy, float* dw, float* db, int n, int h, int w, int c, int ksize, int accumulate,
                               float* partials, size_t partials_bytes) {
    TR_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, TR_EINVAL, "tr_dwconv_wgrad: bad dims n=%d h=%d w=%d c=%d", n, h, w, c);
    TR_REQUIRE(ksize == 3 || ksize == 5 || ksize == 7, TR_EUNSUPPORTED, "tr_dwconv_wgrad: kernel size %d is not built (3, 5, 7)", ksize);
    TR_REQUIRE(view_has(in, c) && view_has(gy, c) && dw && partials, TR_EINVAL, "tr_dwconv_wgrad: in / gy views must hold c=%d channels; dw and partials must be set", c);
    TR_REQUIRE(tr::view_vec4_ok(in) && tr::view_vec4_ok(gy) && in.dtype == gy.dtype, TR_EALIGN,
               "tr_dwconv_wgrad: in / gy views must share a dtype and allow aligned 4-channel access");
    TR_REQUIRE((int64_t)h * w * (in.cstride > gy.cstride ? in.cstride : gy.cstride) * tr::esize(in.dtype) < (int64_t)0x7fffff00, TR_EUNSUPPORTED,
               "tr_dwconv_wgrad: one image of a view must stay below 2 GiB");
    const size_t need = tr_dwconv_wgrad_workspace_bytes(n, h, w, c, ksize);
    TR_REQUIRE(partials_bytes >= need, TR_EINVAL, "tr_dwconv_wgrad: workspace too small (need %zu bytes)", need);
    const DwGeom g = dw_geom(n, h, w, c);
    TR_REQUIRE(g.nsp * g.ncb < (int64_t)0x7fffffff, TR_EUNSUPPORTED, "tr_dwconv_wgrad: grid too large");
    const dim3 grid((unsigned)(g.nsp * g.ncb));
    hipStream_t st = (hipStream_t)stream;
    const int vin = tr::view_vec4_ok(in), vgy = tr::view_vec4_ok(gy);
#define TR_DWG(K, H) hipLaunchKernelGGL((dwconv_wgrad_kernel<K, H>), grid, dim3(256), 0, st, in, gy, partials, h, w, c, g.nstrips, g.ncolblk, g.ncb, vin, vgy)
    if (in.dtype == TR_F32) { if (ksize == 7) TR_DWG(7, false); else if (ksize == 5) TR_DWG(5, false); else TR_DWG(3, false); }
    else { if (ksize == 7) TR_DWG(7, true); else if (ksize == 5) TR_DWG(5, true); else TR_DWG(3, true); }
#undef TR_DWG
    const int ns = ksize * ksize + 1;
    hipLaunchKernelGGL(dwconv_wgrad_finish_kernel, dim3((unsigned)((c * ns + 3) / 4)), dim3(256), 0, st, partials, g.nsp, c, ns, dw, db, accumulate ? 1 : 0);
    return tr::launch_status("tr_dwconv_wgrad");
}

extern "C" int tr_gate_mish(void* stream, tr_view g, tr_view a, tr_view b, tr_view out, int64_t npix) {
    TR_REQUIRE(g.ptr && a.ptr && b.ptr && out.ptr && npix > 0, TR_EINVAL, "tr_gate_mish: views must be set");
    TR_REQUIRE(g.c == a.c + b.c && out.c == g.c && g.c % 4 == 0 && a.c % 4 == 0, TR_EINVAL,
               "tr_gate_mish: channel counts g=%d a=%d b=%d out=%d (g = a + b = out, multiples of 4)", g.c, a.c, b.c, out.c);
    const int c4 = g.c / 4;
    hipLaunchKernelGGL(gate_mish_kernel, dim3(sweep_grid(npix * c4)), dim3(256), 0, (hipStream_t)stream, g, a, b, out, npix, c4, a.c, tr::view_vec4_ok(g),
                       tr::view_vec4_ok(a), tr::view_vec4_ok(b), tr::view_vec4_ok(out));
    return tr::launch_status("tr_gate_mish");
}

extern "C" int tr_gate_mish_bwd(void* stream, tr_view d, tr_view g, tr_view a, tr_view b, tr_view dg, tr_view da, tr_view db, int64_t npix) {
    TR_REQUIRE(d.ptr && g.ptr && a.ptr && b.ptr && dg.ptr && da.ptr && db.ptr && npix > 0, TR_EINVAL, "tr_gate_mish_bwd: views must be set");
    TR_REQUIRE(g.c == a.c + b.c && d.c == g.c && dg.c == g.c && da.c == a.c && db.c == b.c && g.c % 4 == 0 && a.c % 4 == 0, TR_EINVAL,
               "tr_gate_mish_bwd: channel counts do not fit (g=%d a=%d b=%d)", g.c, a.c, b.c);
    const int c4 = g.c / 4;
    hipLaunchKernelGGL(gate_mish_bwd_kernel, dim3(sweep_grid(npix * c4)), dim3(256), 0, (hipStream_t)stream, d, g, a, b, dg, da, db, npix, c4, a.c,
                       tr::view_vec4_ok(d), tr::view_vec4_ok(g), tr::view_vec4_ok(a), tr::view_vec4_ok(b), tr::view_vec4_ok(dg), tr::view_vec4_ok(da),
                       tr::view_vec4_ok(db));
    return tr::launch_status("tr_gate_mish_bwd");
}

extern "C" int tr_act_grad(void* stream, int act, tr_view g, tr_view pre, tr_view out, int64_t npix, const float* sample_scale, int64_t pix_per_sample) {
    TR_REQUIRE(!sample_scale || pix_per_sample > 0, TR_EINVAL, "tr_act_grad: a per-sample scale needs pix_per_sample > 0");
    TR_REQUIRE(g.ptr && pre.ptr && out.ptr && npix > 0, TR_EINVAL, "tr_act_grad: views must be set");
    TR_REQUIRE(act == TR_ACT_MISH || act == TR_ACT_SILU, TR_EUNSUPPORTED, "tr_act_grad: activation %d is not built (SiLU, Mish)", act);
    TR_REQUIRE(g.c == pre.c && out.c == g.c && g.c % 4 == 0, TR_EINVAL, "tr_act_grad: channel counts g=%d pre=%d out=%d", g.c, pre.c, out.c);
    const int c4 = g.c / 4;
    hipLaunchKernelGGL(act_grad_kernel, dim3(sweep_grid(npix * c4)), dim3(256), 0, (hipStream_t)stream, g, pre, out, act, npix, c4, tr::view_vec4_ok(g),
                       tr::view_vec4_ok(pre), tr::view_vec4_ok(out), sample_scale, pix_per_sample > 0 ? pix_per_sample : 1);
    return tr::launch_status("tr_act_grad");
}
