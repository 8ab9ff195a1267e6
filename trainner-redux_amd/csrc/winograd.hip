// Winograd F(2x2, 3x3) convolution in exact fp32 on MFMA, fully fused (input transform, 16 batched GEMMs, output transform, epilogue in
// one kernel) -- for the fp32-forced VGG19 perceptual head (reference: traiNNer/losses/perceptual_loss.py:8-12 casts the head's inputs to
// float32; perceptual_fp16_loss.py:262-329 is the conv / ReLU / pool stack), whose 3x3 convs are 55 % of the headline step at
// v_mfma_f32_32x32x2_f32's 157 TFLOP/s.  Y = A^T [ (G g G^T) (.) (B^T d B) ] A needs 16 multiplies per 2x2 outputs instead of 36: 2.25x
// fewer MFMA cycles.  A non-fused formulation would move 18x the tensor through HBM (the 16 transformed planes are 4x the activations,
// written and read on both sides of the GEMMs): here V = B^T d B is computed by the loader threads on the way into LDS and the 16
// products M_xi leave the accumulators only through LDS, so HBM sees the input once per output-channel group and the output once.
//
// Workgroup = 512 threads = 8 waves, one per CU; tile = 64 output channels x 64 Winograd tiles (8 x 8 tiles = 16 x 16 output pixels).
// K loop over 8-channel chunks, two LDS stages (2 x 64 KB), ONE barrier per chunk.  Per chunk:
//   * loader role (thread = (tile, channel)): 16 scalar loads of the 4 x 4 patch (8 lanes = 32 contiguous bytes of a pixel record), issued
//     one chunk ahead; B^T d B in registers (32 adds); the 16 values go to the 16 xi-planes of the NEXT stage in B-operand order
//     [xi][kk][tile][s] (channel c = 2 s + kk, the K pairing of v_mfma_f32_32x32x2_f32); the pre-transformed weights U (packed
//     [group][chunk][xi][kk][k][s] by tr_pack_winograd) are copied straight into LDS: 32 KB, 4 x 16 B per thread.
//   * MFMA role: wave (j, half) owns the four positions xi = (i, j), i = 0..3, for 32 of the 64 tiles: per position 1 B fragment + 2 A
//     fragments (ds_read_b128, conflict free) feed 8 MFMAs; 32 MFMAs per wave and chunk.  Waves drift apart between barriers, so one wave's
//     stage writes overlap another's MFMAs.
// The first version (256 threads, 32 tiles, one stage, two workgroups per CU) reached 160-208 TFLOP/s effective and was bound by the
// L2 -> LDS traffic of U (32 KB per 32 tiles and chunk: ~20 TB/s chip-wide at full MFMA rate); 64 tiles per staged U halve it.
// Epilogue: wave (j, half) folds its column in registers, T_j = A^T M[:, j] (2 x 16 adds on accumulator tiles of identical layout), the
// waves exchange T through LDS ([j][r][tile][36-float rows]: b128 writes and b32 reads conflict free), and every thread finishes
// Y[r][:] = T[r] A for its (channel, tile) pairs with bias / ReLU / ReLU-derivative mask, storing 128 contiguous bytes per pixel and wave half.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int W_TB = 64;                      // tiles per workgroup (8 rows x 8 cols)
constexpr int W_KB = 64;                      // output channels per workgroup
constexpr int W_CK = 8;                       // input channels per chunk
constexpr int W_THREADS = 512;
constexpr int U_FLOATS = 16 * 2 * W_KB * 4;   // 8192 floats = 32 KB per (group, chunk)
constexpr int V_FLOATS = 16 * 2 * W_TB * 4;   // 8192 floats = 32 KB
constexpr int STAGE_FLOATS = U_FLOATS + V_FLOATS;
constexpr int T_STRIDE = 36;                  // floats per (j, r, tile) row of the exchange buffer (32 + 4: b128-aligned, conflict free)
constexpr int HALO = 18;                      // 8 x 8 tiles of 2 x 2 outputs + the 3 x 3 window: 18 x 18 input pixels per workgroup
constexpr int RAW_STRIDE = 12;                // floats per halo pixel in LDS (8 channels + 4: 16-byte aligned rows, 2-way instead of 4-way read conflicts)
constexpr int RAW_FLOATS = HALO * HALO * RAW_STRIDE;    // 3888 floats = 15.2 KB per buffer
constexpr int SMEM_BYTES = (2 * STAGE_FLOATS + 2 * RAW_FLOATS) * 4;      // 128 KB + 30.4 KB (the exchange buffer, 73,728 B, reuses the stages)

struct WinoArgs {
    int n, h, w, cin, cout;
    int act;
    float mask_slope;
    tr_view in, out, mask;
    const float* u;
    const float* bias;
    int nty, ntx, nchunks, nkg;
};

__global__ __launch_bounds__(W_THREADS, 1) void winograd_f32_kernel(WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* stage = (float*)smem;
    const int tid = threadIdx.x;
    const int l = tid & 63, wv = tid >> 6, lr = l & 31, lh = l >> 5;
    const int wj = wv & 3, wth = wv >> 2;       // Winograd column of this wave, and its half of the 64 tiles
    int b = blockIdx.x;
    const int kg = b % a.nkg; b /= a.nkg;      // output-channel group fastest: neighbours in the grid share the input tile in L2
    const int bx = b % a.ntx; b /= a.ntx;
    const int by = b % a.nty;
    const int nb = b / a.nty;

    // ---- loader role: channel lc of the chunk, tile lt of the workgroup ---------------------------------------------------------------
    const int lc = tid & 7, lt = tid >> 3;
    const int py0 = 2 * (by * 8 + (lt >> 3)) - 1, px0 = 2 * (bx * 8 + (lt & 7)) - 1;     // top-left pixel of the 4 x 4 input patch (pad 1)
    // The 18 x 18 x 8-channel halo of a chunk is fetched ONCE per workgroup with 16-byte loads (648 pieces: one or two per thread, zero
    // padding applied on the way into LDS) and every thread then picks the 16 values of its (tile, channel) patch out of LDS.  Fetching
    // the patches directly cost 16 scalar loads per thread and chunk -- 3.2x redundant between neighbouring tiles -- and kept the CU's
    // texture-address unit busier than its MFMA pipes (8 waves x 20 vector-memory instructions per chunk of 32 MFMAs each).
    float* rawb = stage + 2 * STAGE_FLOATS;
    const float* inimg = (const float*)a.in.ptr + ((int64_t)nb * a.h * a.w) * a.in.cstride + a.in.coff;
    int roff[2], rdst[2];
    bool rok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = tid + j * W_THREADS;
        const int pix = min(piece, HALO * HALO * 2 - 1) >> 1, half = piece & 1;
        const int hy = pix / HALO, hx = pix - hy * HALO;
        const int y = by * 16 - 1 + hy, x = bx * 16 - 1 + hx;
        rok[j] = y >= 0 && y < a.h && x >= 0 && x < a.w;
        roff[j] = (min(max(y, 0), a.h - 1) * a.w + min(max(x, 0), a.w - 1)) * a.in.cstride + half * 4;
        rdst[j] = pix * RAW_STRIDE + half * 4;
    }
    const bool second = tid < HALO * HALO * 2 - W_THREADS;       // threads that carry a second piece
    const f32x4* ub = (const f32x4*)(a.u + (int64_t)kg * a.nchunks * U_FLOATS) + tid;
    f32x4 rawreg[2];
    f32x4 ureg[4];
    auto issue_in = [&](int ch) {                // ch may run past the last chunk (clamped: a redundant load instead of a branch)
        const float* src = inimg + min(ch, a.nchunks - 1) * W_CK;
        rawreg[0] = *(const f32x4*)(src + roff[0]);
        rawreg[1] = *(const f32x4*)(src + roff[1]);
    };
    auto issue_u = [&](int ch) {
        const f32x4* us = ub + (int64_t)min(ch, a.nchunks - 1) * (U_FLOATS / 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) ureg[j] = us[j * W_THREADS];
    };
    auto put_raw = [&](float* rb) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *(f32x4*)(rb + rdst[0]) = rok[0] ? rawreg[0] : z;
        if (second) *(f32x4*)(rb + rdst[1]) = rok[1] ? rawreg[1] : z;
    };
    const int rsrc = (((lt >> 3) * 2) * HALO + (lt & 7) * 2) * RAW_STRIDE + lc;          // top-left value of this thread's patch in a raw buffer
    const int voff = U_FLOATS + ((lc & 1) * W_TB + lt) * 4 + (lc >> 1);                  // + xi * 512 floats within a stage
    auto fill = [&](float* st, const float* rb) {    // raw halo + weight registers of a chunk -> one LDS stage
        // V = B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        float t[16], d[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) d[p] = rb[rsrc + ((p >> 2) * HALO + (p & 3)) * RAW_STRIDE];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            t[0 + x] = d[0 + x] - d[8 + x];
            t[4 + x] = d[4 + x] + d[8 + x];
            t[8 + x] = d[8 + x] - d[4 + x];
            t[12 + x] = d[4 + x] - d[12 + x];
        }
        float* vdst = st + voff;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vdst[(i * 4 + 0) * 512] = t[i * 4 + 0] - t[i * 4 + 2];
            vdst[(i * 4 + 1) * 512] = t[i * 4 + 1] + t[i * 4 + 2];
            vdst[(i * 4 + 2) * 512] = t[i * 4 + 2] - t[i * 4 + 1];
            vdst[(i * 4 + 3) * 512] = t[i * 4 + 1] - t[i * 4 + 3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) ((f32x4*)st)[tid + j * W_THREADS] = ureg[j];
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][kh][e] = 0.f;
    // 4 steps (position i) of 8 MFMAs; the two channel halves alternate so that consecutive MFMAs never accumulate into the same tile, and
    // the operand fragments of step i + 1 are read from LDS before the MFMAs of step i issue (a fresh ds_read in front of every group cost
    // its full latency, ~100+ cycles under the fill traffic, per 256 cycles of MFMAs)
    auto mfmas = [&](const float* ul, const float* vl) {
        const float* ua = ul + (lh * W_KB + lr) * 4;
        const float* vb = vl + (lh * W_TB + wth * 32 + lr) * 4;
        f32x4 bv[2], au[2][2];
        bv[0] = *(const f32x4*)(vb + wj * 512);
        au[0][0] = *(const f32x4*)(ua + wj * 512);
        au[0][1] = *(const f32x4*)(ua + wj * 512 + 128);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i + 1 < 4) {
                const int xi2 = (i + 1) * 4 + wj;
                bv[(i + 1) & 1] = *(const f32x4*)(vb + xi2 * 512);
                au[(i + 1) & 1][0] = *(const f32x4*)(ua + xi2 * 512);
                au[(i + 1) & 1][1] = *(const f32x4*)(ua + xi2 * 512 + 128);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(au[i & 1][0][e], bv[i & 1][e], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(au[i & 1][1][e], bv[i & 1][e], acc[i][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    issue_in(0);
    issue_u(0);
    put_raw(rawb);
    issue_in(1);
    __syncthreads();
    fill(stage, rawb);
    put_raw(rawb + RAW_FLOATS);
    issue_in(2);
    issue_u(1);
    __syncthreads();
    for (int ch = 0; ch < a.nchunks; ++ch) {
        const float* ul = stage + (ch & 1) * STAGE_FLOATS;
        float* nst = stage + ((ch + 1) & 1) * STAGE_FLOATS;
        const float* rb = rawb + ((ch + 1) & 1) * RAW_FLOATS;
        // Three chunks in flight: MFMAs on chunk ch; stage fill for ch + 1 (from the raw halo written one iteration ago) tucked behind the
        // first two MFMA steps of this wave; raw halo of ch + 2 into LDS and the loads of ch + 3 behind the second step.
        float d[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) d[p] = rb[rsrc + ((p >> 2) * HALO + (p & 3)) * RAW_STRIDE];     // lands during the first MFMA step
        const float* ua = ul + (lh * W_KB + lr) * 4;
        const float* vb = ul + U_FLOATS + (lh * W_TB + wth * 32 + lr) * 4;
        f32x4 bv[2], au[2][2];
        bv[0] = *(const f32x4*)(vb + wj * 512);
        au[0][0] = *(const f32x4*)(ua + wj * 512);
        au[0][1] = *(const f32x4*)(ua + wj * 512 + 128);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i + 1 < 4) {
                const int xi2 = (i + 1) * 4 + wj;
                bv[(i + 1) & 1] = *(const f32x4*)(vb + xi2 * 512);
                au[(i + 1) & 1][0] = *(const f32x4*)(ua + xi2 * 512);
                au[(i + 1) & 1][1] = *(const f32x4*)(ua + xi2 * 512 + 128);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(au[i & 1][0][e], bv[i & 1][e], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(au[i & 1][1][e], bv[i & 1][e], acc[i][1], 0, 0, 0);
            }
            if (i == 0) {
#pragma unroll
                for (int x = 0; x < 4; ++x) {                // B^T d in place
                    const float a0 = d[0 + x], a1 = d[4 + x], a2 = d[8 + x], a3 = d[12 + x];
                    d[0 + x] = a0 - a2; d[4 + x] = a1 + a2; d[8 + x] = a2 - a1; d[12 + x] = a1 - a3;
                }
            }
            {   // one quarter of the next stage's fill behind every MFMA step: V row i and the i-th 16-byte piece of U; that piece's register
                // is re-loaded for chunk ch + 2 at once (rolling prefetch: a whole chunk of lead, no second register set)
                float* vdst = nst + voff;
                vdst[(i * 4 + 0) * 512] = d[i * 4 + 0] - d[i * 4 + 2];
                vdst[(i * 4 + 1) * 512] = d[i * 4 + 1] + d[i * 4 + 2];
                vdst[(i * 4 + 2) * 512] = d[i * 4 + 2] - d[i * 4 + 1];
                vdst[(i * 4 + 3) * 512] = d[i * 4 + 1] - d[i * 4 + 3];
                ((f32x4*)nst)[tid + i * W_THREADS] = ureg[i];
                ureg[i] = (ub + (int64_t)min(ch + 2, a.nchunks - 1) * (U_FLOATS / 4))[i * W_THREADS];
            }
            if (i == 2) {
                put_raw(rawb + (ch & 1) * RAW_FLOATS);     // chunk ch + 2; this buffer's previous content (chunk ch) was consumed an iteration ago
                issue_in(ch + 3);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- output transform + epilogue ------------------------------------------------------------------------------------------------------------
    // One exchange round for both 32-channel halves ([kh][j][r][tile][36 floats] = 144 KB: the stages and raw buffers are free by now), then
    // every thread finishes one tile x 4 consecutive channels per half with 16-byte LDS reads and 16-byte stores (8 lanes = the 128
    // contiguous bytes of a pixel's 32 channels).
    float* tl = (float*)smem;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 t0, t1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float m0 = acc[0][kh][q * 4 + e], m1 = acc[1][kh][q * 4 + e], m2 = acc[2][kh][q * 4 + e], m3 = acc[3][kh][q * 4 + e];
                t0[e] = m0 + m1 + m2;                      // A^T = [1 1 1 0; 0 1 -1 -1] applied down column j
                t1[e] = m1 - m2 - m3;
            }
            *(f32x4*)(tl + (((kh * 4 + wj) * 2 + 0) * W_TB + wth * 32 + lr) * T_STRIDE + q * 8 + lh * 4) = t0;
            *(f32x4*)(tl + (((kh * 4 + wj) * 2 + 1) * W_TB + wth * 32 + lr) * T_STRIDE + q * 8 + lh * 4) = t1;
        }
    __syncthreads();
    const int ek4 = (tid & 7) * 4, et = tid >> 3;
    const int ty = by * 8 + (et >> 3), tx = bx * 8 + (et & 7);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        const int kc = kg * W_KB + kh * 32 + ek4;
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) bias = *(const f32x4*)(a.bias + kc);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float* tp = tl + ((kh * 4 * 2 + r) * W_TB + et) * T_STRIDE + ek4;
            const f32x4 T0 = *(const f32x4*)(tp), T1 = *(const f32x4*)(tp + 2 * W_TB * T_STRIDE);
            const f32x4 T2 = *(const f32x4*)(tp + 4 * W_TB * T_STRIDE), T3 = *(const f32x4*)(tp + 6 * W_TB * T_STRIDE);
            const f32x4 y[2] = {T0 + T1 + T2, T1 - T2 - T3};
            const int oy = 2 * ty + r;
            if (oy >= a.h) continue;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ox = 2 * tx + s;
                if (ox >= a.w) continue;
                const int64_t pix = ((int64_t)nb * a.h + oy) * a.w + ox;
                f32x4 v = y[s] + bias;
                if (a.act == TR_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (a.mask.ptr) {
                    const f32x4 m = *(const f32x4*)((const float*)a.mask.ptr + pix * a.mask.cstride + a.mask.coff + kc);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= m[e] > 0.f ? 1.f : a.mask_slope;
                }
                *(f32x4*)((float*)a.out.ptr + pix * a.out.cstride + a.out.coff + kc) = v;
            }
        }
    }
}

// ---- second formulation: 16x16x4 MFMAs, one wave = all 16 positions of a 16-channel x 32-tile block --------------------------------------
// Workgroup = 256 threads = 4 waves = 64 output channels x 32 tiles (4 x 8 tiles, 8 x 16 output pixels); two to three workgroups per CU.
// Differences from the kernel above, each aimed at what its counters showed (MFMA pipe idle 35 %: every wave fills LDS at the same time,
// and most of that fill is LDS WRITE time -- ds_write_b32 moves 64 B/clk, ds_write_b128 79 B/clk):
//   * the weights never touch LDS: a wave's A fragments (its own 16 channels, 8 input channels, 16 positions: 32 VGPRs) come straight from
//     the packed array with 8-byte loads that are contiguous over the wave, and each position's registers are re-loaded for the NEXT chunk
//     right after their last MFMA of this chunk (a rolling prefetch with almost a whole chunk of lead and no second register set);
//   * only V goes through LDS: 16 KB per chunk and workgroup instead of 64 KB;
//   * a lane holds all 16 positions of its (channel quad, tile) elements, so the output transform A^T M A is pure register arithmetic:
//     no exchange through LDS, no epilogue barriers, 16-byte stores;
//   * the two waves that share a SIMD belong to different workgroups and drift apart by themselves.
typedef __attribute__((ext_vector_type(2))) float f32x2;
constexpr int B_TB = 32, B_KB = 64, B_THREADS = 256;
constexpr int BV_FLOATS = 16 * 256;            // V of one chunk: [xi][tile block 2][kq 4][tile 16][step 2] = 16 KB
constexpr int BU_FLOATS = 4 * 16 * 64 * 2;      // packed U of one (group, chunk): [wave 4][xi 16][lane 64][step 2] = 32 KB

__global__ __launch_bounds__(B_THREADS, 2) void winograd16_f32_kernel(WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float vstage[2][BV_FLOATS];
    const int tid = threadIdx.x;
    const int l = tid & 63, wv = tid >> 6;
    int b = blockIdx.x;
    const int kg = b % a.nkg; b /= a.nkg;
    const int bx = b % a.ntx; b /= a.ntx;
    const int by = b % a.nty;
    const int nb = b / a.nty;
    // ---- loader role -------------------------------------------------------------------------------------------------------------------
    const int lc = tid & 7, lt = tid >> 3;
    const int py0 = 2 * (by * 4 + (lt >> 3)) - 1, px0 = 2 * (bx * 8 + (lt & 7)) - 1;
    unsigned okmask = 0;
    int yo[4], xo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = py0 + r, x = px0 + r;
        if (y >= 0 && y < a.h) okmask |= 0xFu << (4 * r);
        yo[r] = min(max(y, 0), a.h - 1) * (a.w * a.in.cstride);
        xo[r] = min(max(x, 0), a.w - 1) * a.in.cstride;
    }
    {
        unsigned cm = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (px0 + c >= 0 && px0 + c < a.w) cm |= 0x1111u << c;
        okmask &= cm;
    }
    const float* inb = (const float*)a.in.ptr + ((int64_t)nb * a.h * a.w) * a.in.cstride + a.in.coff + lc;
    float d[16];
    auto issue = [&](int ch) {
        const float* src = inb + min(ch, a.nchunks - 1) * W_CK;
#pragma unroll
        for (int p = 0; p < 16; ++p) d[p] = src[yo[p >> 2] + xo[p & 3]];
    };
    // V element of (tile lt, channel lc): [xi][tb = lt >> 4][kq = lc & 3][j = lt & 15][step = lc >> 2]
    const int voff = (((lt >> 4) * 4 + (lc & 3)) * 16 + (lt & 15)) * 2 + (lc >> 2);
    auto fill = [&](float* st) {
        float t[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) d[p] = ((okmask >> p) & 1u) ? d[p] : 0.f;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            t[0 + x] = d[0 + x] - d[8 + x];
            t[4 + x] = d[4 + x] + d[8 + x];
            t[8 + x] = d[8 + x] - d[4 + x];
            t[12 + x] = d[4 + x] - d[12 + x];
        }
        float* vdst = st + voff;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vdst[(i * 4 + 0) * 256] = t[i * 4 + 0] - t[i * 4 + 2];
            vdst[(i * 4 + 1) * 256] = t[i * 4 + 1] + t[i * 4 + 2];
            vdst[(i * 4 + 2) * 256] = t[i * 4 + 2] - t[i * 4 + 1];
            vdst[(i * 4 + 3) * 256] = t[i * 4 + 1] - t[i * 4 + 3];
        }
    };
    // ---- MFMA role -----------------------------------------------------------------------------------------------------------------------
    const f32x2* ubase = (const f32x2*)(a.u + ((int64_t)kg * a.nchunks * 4 + wv) * (16 * 64 * 2)) + l;     // + (ch * 4 * 16 + xi) * 64
    auto loadA = [&](int ch, int xi) { return ubase[((int64_t)min(ch, a.nchunks - 1) * 64 + xi) * 64]; };
    f32x2 af[16];
    f32x4 acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
        af[xi] = loadA(0, xi);
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) acc[xi][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    issue(0);
    fill(vstage[0]);
    issue(1);
    __syncthreads();
    for (int ch = 0; ch < a.nchunks; ++ch) {
        const float* vl = vstage[ch & 1] + l * 2;            // lane (kq = l >> 4, j = l & 15): ((tb * 4 + kq) * 16 + j) * 2 = tb * 128 + l * 2
        fill(vstage[(ch + 1) & 1]);
        issue(ch + 2);
        f32x2 bq[2][2];
        bq[0][0] = *(const f32x2*)(vl);
        bq[0][1] = *(const f32x2*)(vl + 128);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            if (xi + 1 < 16) {
                bq[(xi + 1) & 1][0] = *(const f32x2*)(vl + (xi + 1) * 256);
                bq[(xi + 1) & 1][1] = *(const f32x2*)(vl + (xi + 1) * 256 + 128);
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[xi][st], bq[xi & 1][0][st], acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[xi][st], bq[xi & 1][1][st], acc[xi][1], 0, 0, 0);
            }
            af[xi] = loadA(ch + 1, xi);                       // rolling prefetch: this position's registers are free until the next chunk
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // ---- output transform in registers + epilogue ------------------------------------------------------------------------------------------
    const int kq4 = (l >> 4) * 4;                            // the lane's 4 consecutive channels: D[i = 4 (l / 16) + r][j = l % 16]
    const int kc = kg * B_KB + wv * 16 + kq4;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[r] = a.bias[kc + r];
    }
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int t = tb * 16 + (l & 15);
        const int ty = by * 4 + (t >> 3), tx = bx * 8 + (t & 7);
        float y[2][2][4];                                     // [row][col][channel]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float T[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float m0 = acc[0 + j][tb][r], m1 = acc[4 + j][tb][r], m2 = acc[8 + j][tb][r], m3 = acc[12 + j][tb][r];
                T[j][0] = m0 + m1 + m2;
                T[j][1] = m1 - m2 - m3;
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                y[rr][0][r] = T[0][rr] + T[1][rr] + T[2][rr];
                y[rr][1][r] = T[1][rr] - T[2][rr] - T[3][rr];
            }
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int oy = 2 * ty + rr;
            if (oy >= a.h) continue;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ox = 2 * tx + s;
                if (ox >= a.w) continue;
                const int64_t pix = ((int64_t)nb * a.h + oy) * a.w + ox;
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float e = y[rr][s][r] + bias4[r];
                    if (a.act == TR_ACT_RELU) e = fmaxf(e, 0.f);
                    v[r] = e;
                }
                if (a.mask.ptr) {
                    const f32x4 m = *(const f32x4*)((const float*)a.mask.ptr + pix * a.mask.cstride + a.mask.coff + kc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= m[r] > 0.f ? 1.f : a.mask_slope;
                }
                *(f32x4*)((float*)a.out.ptr + pix * a.out.cstride + a.out.coff + kc) = v;
            }
        }
    }
}

// packed U for the kernel above: [co / 64][ci / 8][wave = (co % 64) / 16][xi][lane = ((ci % 8) & 3) * 16 + co % 16][step = (ci % 8) >> 2]
__global__ __launch_bounds__(256) void winograd16_pack_kernel(const float* __restrict__ w, float* __restrict__ dst, int cout, int cin, int src_i, int transpose) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)cout * cin) return;
    const int co = (int)(i / cin), ci = (int)(i % cin);
    float g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = transpose ? w[((int64_t)ci * src_i + co) * 9 + (8 - t)] : w[((int64_t)co * src_i + ci) * 9 + t];
    float gg[12];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        gg[0 + x] = g[0 + x];
        gg[3 + x] = 0.5f * (g[0 + x] + g[3 + x] + g[6 + x]);
        gg[6 + x] = 0.5f * (g[0 + x] - g[3 + x] + g[6 + x]);
        gg[9 + x] = g[6 + x];
    }
    const int nchunks = cin / W_CK, cc = ci % W_CK;
    float* base = dst + (((int64_t)(co / 64) * nchunks + ci / W_CK) * 4 + (co % 64) / 16) * (16 * 64 * 2) + ((cc & 3) * 16 + co % 16) * 2 + (cc >> 2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float a0 = gg[r * 3 + 0], a1 = gg[r * 3 + 1], a2 = gg[r * 3 + 2];
        base[(r * 4 + 0) * 128] = a0;
        base[(r * 4 + 1) * 128] = 0.5f * (a0 + a1 + a2);
        base[(r * 4 + 2) * 128] = 0.5f * (a0 - a1 + a2);
        base[(r * 4 + 3) * 128] = a2;
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

inline bool wino_impl16() {      // TRAINNER_WINO_IMPL=16: the 16x16x4 formulation (pack and conv must agree: read once per process)
    static const bool v = getenv("TRAINNER_WINO_IMPL") && atoi(getenv("TRAINNER_WINO_IMPL")) == 16;
    return v;
}

inline bool wino_view_ok(const tr_view& v, int c) { return v.ptr && v.dtype == TR_F32 && v.c >= c && v.coff >= 0 && v.coff + c <= v.cstride; }

}  // namespace

extern "C" size_t tr_winograd_packed_bytes(int cin, int cout) { return (size_t)cin * cout * 16 * sizeof(float); }

extern "C" int tr_pack_winograd(void* stream, const float* w_oihw, float* dst, int cout, int cin, int transpose) {
    TR_REQUIRE(w_oihw && dst && cout > 0 && cin > 0 && cout % W_KB == 0 && cin % W_CK == 0, TR_EINVAL,
               "tr_pack_winograd: cout=%d must be a multiple of %d and cin=%d of %d", cout, W_KB, cin, W_CK);
    TR_REQUIRE(((uintptr_t)dst) % 16 == 0, TR_EALIGN, "tr_pack_winograd: dst must be 16-byte aligned");
    const int64_t n = (int64_t)cout * cin;
    if (wino_impl16()) {
        hipLaunchKernelGGL(winograd16_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, dst, cout, cin,
                           transpose ? cout : cin, transpose ? 1 : 0);
        return tr::launch_status("tr_pack_winograd");
    }
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
    TR_REQUIRE(wino_impl16() || (d->in.cstride % 4 == 0 && d->in.coff % 4 == 0 && ((uintptr_t)d->in.ptr) % 16 == 0), TR_EALIGN,
               "tr_conv3x3_winograd_f32: the input view must allow aligned 4-channel access");
    TR_REQUIRE((int64_t)d->h * d->w * d->in.cstride < (int64_t)0x7fffffff, TR_EUNSUPPORTED, "tr_conv3x3_winograd_f32: one image of the input must stay below 2^31 elements");
    WinoArgs a;
    a.n = d->n; a.h = d->h; a.w = d->w; a.cin = d->cin; a.cout = d->cout;
    a.act = d->act; a.mask_slope = d->mask_slope;
    a.in = d->in; a.out = d->out; a.mask = d->mask;
    a.u = (const float*)d->wpacked; a.bias = d->bias;
    a.nty = (d->h / 2 + 7) / 8; a.ntx = (d->w / 2 + 7) / 8;
    a.nchunks = d->cin / W_CK; a.nkg = d->cout / W_KB;
    TR_REQUIRE(d->out.cstride % 4 == 0 && d->out.coff % 4 == 0 && ((uintptr_t)d->out.ptr) % 16 == 0 &&
                   (!d->mask.ptr || (d->mask.cstride % 4 == 0 && d->mask.coff % 4 == 0 && ((uintptr_t)d->mask.ptr) % 16 == 0)) &&
                   (!d->bias || ((uintptr_t)d->bias) % 16 == 0),
               TR_EALIGN, "tr_conv3x3_winograd_f32: output / mask views and the bias must allow aligned 4-channel access");
    if (wino_impl16()) {
        a.nty = (d->h / 2 + 3) / 4;
        const int64_t nb16 = (int64_t)d->n * a.nty * a.ntx * a.nkg;
        TR_REQUIRE(nb16 < (int64_t)0x7fffffff, TR_EUNSUPPORTED, "tr_conv3x3_winograd_f32: grid too large");
        hipLaunchKernelGGL(winograd16_f32_kernel, dim3((unsigned)nb16), dim3(B_THREADS), 0, (hipStream_t)stream, a);
        return tr::launch_status("tr_conv3x3_winograd_f32");
    }
    const int64_t nblk = (int64_t)d->n * a.nty * a.ntx * a.nkg;
    TR_REQUIRE(nblk < (int64_t)0x7fffffff, TR_EUNSUPPORTED, "tr_conv3x3_winograd_f32: grid too large");
    static const hipError_t attr = hipFuncSetAttribute((const void*)winograd_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    TR_REQUIRE(attr == hipSuccess, TR_EUNSUPPORTED, "tr_conv3x3_winograd_f32: the device does not grant %d bytes of LDS per workgroup", SMEM_BYTES);
    hipLaunchKernelGGL(winograd_f32_kernel, dim3((unsigned)nblk), dim3(W_THREADS), SMEM_BYTES, (hipStream_t)stream, a);
    return tr::launch_status("tr_conv3x3_winograd_f32");
}
