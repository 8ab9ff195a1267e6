// Weight / bias gradient of the 16-bit 3x3 convolutions on NHWC views: persistent LDS-DMA ring kernel.
//
// Contract: include/trainner_hip.h (tr_wgrad3x3); replaces what autograd's conv backward accumulates into Conv2d.weight.grad / .bias.grad at
// traiNNer/models/sr_model.py:681 for the RRDB trunk and the upsampling tail (see wgrad3x3.hip for the general kernel and the reduction stage).
//
//   dW[co][tap][ci] = sum_p gy[p][co] * x[p + tap][ci]            GEMM per tap: M = co, N = ci, K = pixels
//
// Why a second kernel ("wgrad ring kernel").  wgrad_kernel gives a workgroup of THREE waves one 32-channel input chunk and
// stages every operand global -> VGPR -> LDS between two workgroup barriers per 8 x 32 tile; it sustains ~20 % of the MFMA peak on the RRDB shapes,
// bound by how many staged bytes a tile of MFMAs needs (1 byte per 124 FLOP at 32 output channels) and by its barrier structure.  Here, as in
// conv_ring.hip:
//   * one workgroup per CU, persistent over its tiles, accumulators resident until the end (one partial block per consumer wave);
//   * a workgroup owns 64 input channels x one output group x 9 taps (1 byte per 158 / 247 FLOP at 32 / 64 output channels);
//     its four consumer waves split that block 2 x 2: (input-channel half) x (pixel-row half of the tile) at 32 output channels,
//     (input-channel half) x (output-channel half) at 64 -- nine 32 x 32 accumulators each;
//   * four loader waves move the x halo tile and the gy tile of the NEXT tile into the other LDS stage with buffer_load ... lds while the
//     consumers issue nothing but transposing LDS reads (ds_read_b64_tr_b16: K = pixels is the strided axis of both NHWC operands) and MFMAs;
//     ONE raw s_barrier per tile;
//   * operand fragments are prefetched four MFMAs ahead: a consumer wave is alone on its SIMD and has no partner to hide an LDS round trip.
// Partial blocks keep wgrad_kernel's layout [slot][32-channel chunk][output group][co][tap][ci], so wgrad_reduce_kernel finishes both kernels.
#include "common.h"
#include "wgrad_args.h"
#include <stdlib.h>
#include <string.h>
#include <utility>
#ifndef TR_WGR_ABL
#define TR_WGR_ABL 0     // timing experiments: 1 = consumers skip their tiles, 2 = loaders issue nothing (results are wrong)
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2v;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// tile height: 8 rows / two stages (one tile in flight behind the one in work) or 4 rows / three stages (two in flight).  Measured on the dense-block
// shapes: 4 rows gain on the narrow single launches (128 -> 32: 0.119 -> 0.106 ms) and lose on the fused launch (0.492 -> 0.512 ms per block: the
// halo rows are 1.5x instead of 1.25x of a tile and there are twice as many barriers), which is where the time is
#ifndef TR_WGR_TH
#define TR_WGR_TH 8
#endif
constexpr int TW = 32, TH = TR_WGR_TH, HALO_W = TW + 2, HALO_H = TH + 2;
constexpr int NSTAGE = TH == 4 ? 3 : 2;                    // LDS stages: 4-row tiles leave room for a third, i.e. two tiles in flight behind the one in work
static_assert(NSTAGE == 2 || NSTAGE == 3, "vmcnt bookkeeping of the loader loop");
constexpr int NCW = 4, NLW = 4;
constexpr int XPIX = 128;                                   // 64 input channels x 2 bytes per halo pixel
constexpr int X_ITEMS = HALO_H * HALO_W * 8;                // 16-byte slots
constexpr int X_BYTES = X_ITEMS * 16;
constexpr int X_IT = (X_ITEMS + 64 * NLW - 1) / (64 * NLW);
constexpr unsigned OOB = 0x7fffff00u;
constexpr int LDS_MAX = 160 * 1024;

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int NT>
struct WCfg {
    static constexpr int CG = NT * 32;
    static constexpr int GPIX = CG * 2;                      // bytes per gy pixel
    static constexpr int G_ITEMS = TH * TW * (GPIX / 16);
    static constexpr int G_BYTES = G_ITEMS * 16;
    static constexpr int G_IT = G_ITEMS / (64 * NLW);
    static constexpr int STAGE = X_BYTES + G_BYTES;
    static constexpr int NP = X_IT + G_IT;
    static constexpr int NSTEP = NT == 1 ? TH : 2 * TH;      // 16-pixel K steps per wave and tile
    static_assert(G_ITEMS % (64 * NLW) == 0, "gy pieces");
};

#define TR_TRRD(PTR) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(PTR))

template <int CT, int NT>
__global__ __launch_bounds__((NCW + NLW) * 64) void wgrad_ring_kernel(WgArgs a, WgRoles roles) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef WCfg<NT> C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int l = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup id = slot + nslots * role (wgrad_args.h): with nslots a multiple of 8 the workgroups that walk the SAME tiles share an XCD's L2
    const int slot = blockIdx.x % a.nslots;
    const WgRole& R = roles.r[blockIdx.x / a.nslots];
    const int pc = R.pc;
    // tile walk: slot s takes tiles s + k * nslots (dealing the slots of one XCD a run of consecutive tiles instead, for halo reuse in its L2,
    // measured +5-10 % FETCH_SIZE and no change in time: the neighbours in a run are not in step)
    const int first = slot;
    const int nmine = (a.ntiles - first + a.nslots - 1) / a.nslots;

    if (wv >= NCW) {
        // =================================================== loader waves ===================================================
        const int lw = wv - NCW;
        const int64_t in_pix_bytes = (int64_t)a.in.cstride * 2, gy_pix_bytes = (int64_t)a.gy.cstride * 2;
        const unsigned in_img = (unsigned)((int64_t)a.hin * a.win * in_pix_bytes), gy_img = (unsigned)((int64_t)a.h * a.w * gy_pix_bytes);
        auto issue = [&](int k, int stage) {
            int t_ = first + k * a.nslots;
            const int x0 = (t_ % a.ntx) * TW; t_ /= a.ntx;
            const int y0 = (t_ % a.nty) * TH;
            const int nb = t_ / a.nty;
            const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in.ptr + (int64_t)nb * a.hin * a.win * in_pix_bytes), 0, in_img, 0x00020000);
            const auto gy_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.gy.ptr + (int64_t)nb * a.h * a.w * gy_pix_bytes), 0, gy_img, 0x00020000);
            char* st = smem + stage * C::STAGE;
            // x halo: position P = pixel * 8 + stored slot; the 64-byte halves of a pixel are swapped where bit 1 of its COLUMN is set (bank swizzle
            // of the transposed reads, on the source address: the LDS image is position-linear)
#pragma unroll
            for (int i = 0; i < X_IT; ++i) {
                int p0 = (i * NLW + lw) * 64;
                p0 = p0 < X_ITEMS - 64 ? p0 : X_ITEMS - 64;                        // the tail piece re-fetches the end of the tile
                const int P = p0 + l;
                const int pp = P >> 3;
                const int py = pp / HALO_W, px = pp - py * HALO_W;
                const int s = (P & 7) ^ (((px >> 1) & 1) << 2);
                int oy = y0 + py - 1 - a.shift, ox = x0 + px - 1 - a.shift;
                bool ok = (oy >= 0) & (oy < a.ih) & (ox >= 0) & (ox < a.iw);
                if (a.replicate) {
                    oy = min(max(oy, 0), a.ih - 1);
                    ox = min(max(ox, 0), a.iw - 1);
                    ok = true;
                }
                const int c0 = pc * 64 + s * 8;
                ok = ok & (c0 + 8 <= R.in_c);
                const unsigned off = (unsigned)(((int64_t)(oy >> a.up) * a.win + (ox >> a.up)) * in_pix_bytes + (int64_t)(a.in.coff + c0) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (__attribute__((address_space(3))) void*)(st + p0 * 16), 16, ok ? off : OOB, 0, 0, 0);
            }
            // gy tile: [row][32 pixels][CG channels]; 128-byte rows (64 output channels) swap their halves where bit 1 of the column is set
#pragma unroll
            for (int i = 0; i < C::G_IT; ++i) {
                const int p0 = (i * NLW + lw) * 64;
                const int P = p0 + l;
                constexpr int SL = C::GPIX / 16;
                const int gp = P / SL;
                int s = P - gp * SL;
                const int col = gp & 31;
                if (NT == 2) s ^= ((col >> 1) & 1) << 2;
                const int oy = y0 + (gp >> 5), ox = x0 + col;
                const int hf = NT == 2 ? s >> 2 : 0;              // 32-channel output half of the piece: its own slice of the gy buffer
                const int c0 = (NT == 2 ? s & 3 : s) * 8;
                const bool ok = (oy < a.h) & (ox < a.w) & (c0 + 8 <= R.gy_c[hf]);
                const unsigned off = (unsigned)(((int64_t)oy * a.w + ox) * gy_pix_bytes + (int64_t)(R.gy_coff[hf] + c0) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(gy_rsrc, (__attribute__((address_space(3))) void*)(st + X_BYTES + p0 * 16), 16, ok ? off : OOB, 0, 0, 0);
            }
        };
        // NSTAGE - 1 tiles in flight while the consumers work on tile k: tile k + NSTAGE - 1 is requested right AFTER barrier k, into the stage of
        // tile k - 1 (its consumption ended before that barrier).  Loads retire in order: tile k has landed when only the NP pieces of each
        // later request are outstanding.
        constexpr int D = NSTAGE - 1;
        int stage_n = 0;                                         // stage of the next tile to request
        for (int k = 0; k < D && k < nmine; ++k) {
            if (!(TR_WGR_ABL & 2)) issue(k, stage_n);
            stage_n = stage_n == NSTAGE - 1 ? 0 : stage_n + 1;
        }
        for (int k = 0; k < nmine; ++k) {
            const int later = min(nmine - 1 - k, D - 1);         // tiles requested behind tile k at this point
            if (later >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (k + D < nmine) {
                if (!(TR_WGR_ABL & 2)) issue(k + D, stage_n);
                stage_n = stage_n == NSTAGE - 1 ? 0 : stage_n + 1;
            }
        }
        if (NT == 1) {                                           // the consumers' exchange of the two pixel-row halves (below)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ===================================================== consumer waves =====================================================
    const int cih = wv & 1;                       // input-channel half of the workgroup's 64
    const int hi = wv >> 1;                       // NT 1: pixel-row half of the tile ; NT 2: output-channel half
    const int chunk = pc * 2 + cih;               // 32-channel chunk of the partial layout
    const int hf = NT == 2 ? hi : 0;              // output half of this wave
    const bool live = chunk < R.nchunks[hf];          // a 32-channel tail (cin = 96, 160): the other half of the last pair has nothing to do
    const int lh = l >> 5, li = l & 15, q = li >> 2, p = li & 3;
    const int i0 = ((l >> 4) & 1) * 16;
    const int rbase = NT == 1 ? hi * (TH / 2) : 0;
    const int coh = NT == 2 ? hi : 0;
    // lane terms of the fragment addresses (everything else is an immediate): x by kx, gy once
    int xo[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xo[kx] = (kx + 8 * lh + q) * XPIX + ((cih * 64 + (i0 + 4 * p) * 2) ^ ((((kx + q) >> 1) & 1) << 6));
    const int go = (8 * lh + q) * C::GPIX + (NT == 2 ? ((coh * 64 + (i0 + 4 * p) * 2) ^ (((q >> 1) & 1) << 6)) : (i0 + 4 * p) * 2);

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // bias gradient: column sums of the gy fragments, taken by the waves of the first pair-chunk's first input half (every output channel once per K share)
    const bool do_bias = (R.db_partials[hf] != nullptr) & (pc == 0) & (cih == 0);
    float bsum = 0.f;

    int stage_c = 0;
    for (int k = 0; k < nmine; ++k) {
        __builtin_amdgcn_s_barrier();
        const char* st = smem + stage_c * C::STAGE;
        stage_c = stage_c == NSTAGE - 1 ? 0 : stage_c + 1;
        if (!live || (TR_WGR_ABL & 1)) continue;
        // the wave's first tile row goes into the base pointers: every fragment address below is base + immediate
        const char* xb[3] = {st + xo[0] + rbase * (HALO_W * XPIX), st + xo[1] + rbase * (HALO_W * XPIX), st + xo[2] + rbase * (HALO_W * XPIX)};
        const char* gb = st + X_BYTES + go + rbase * (32 * C::GPIX);
        constexpr int NI = C::NSTEP * 9;          // (K step, tap) items, one MFMA each
        constexpr int PF = 4;                     // operand fragments are requested PF items ahead
        bf16x8 fb[PF], fa[2];
        auto load_b = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr int s = I / 9, tap = I % 9, ky = tap / 3, kx = tap % 3;
            constexpr int r = s >> 1, kb = s & 1;
            const char* ptr = xb[kx] + ((r + ky) * HALO_W + kb * 16) * XPIX;
            const s16x4 lo = TR_TRRD(ptr), hi4 = TR_TRRD(ptr + 4 * XPIX);
            fb[I % PF] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        };
        auto load_a = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int r = s >> 1, kb = s & 1;
            const char* ptr = gb + (r * 32 + kb * 16) * C::GPIX;
            const s16x4 lo = TR_TRRD(ptr), hi4 = TR_TRRD(ptr + 4 * C::GPIX);
            fa[s & 1] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        };
        load_a(std::integral_constant<int, 0>{});
        load_b(std::integral_constant<int, 0>{});
        load_b(std::integral_constant<int, 1>{});
        load_b(std::integral_constant<int, 2>{});
        auto item = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr int s = I / 9, tap = I % 9;
            if constexpr (I + PF - 1 < NI) load_b(std::integral_constant<int, I + PF - 1>{});
            if constexpr (tap == 4 && s + 1 < C::NSTEP) load_a(std::integral_constant<int, s + 1>{});
            if constexpr (CT == TR_BF16) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s & 1], fb[I % PF], acc[tap], 0, 0, 0);
            else acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[s & 1]), __builtin_bit_cast(f16x8, fb[I % PF]), acc[tap], 0, 0, 0);
            if constexpr (tap == 8) {
                if (do_bias) {
                    const uint4 u = __builtin_bit_cast(uint4, fa[s & 1]);
                    const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (CT == TR_BF16) bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, w4[e]), __builtin_bit_cast(bf16x2v, 0x3F803F80u), bsum, false);
                        else bsum = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2v, w4[e]), __builtin_bit_cast(f16x2v, 0x3C003C00u), bsum, false);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);    // keep the prefetch distance (VGPR budget: nine accumulators)
        };
        static_for<NI>(item);
    }

    // 32-wide form: the two pixel-row halves of the tile (waves hi = 0 / 1 of an input-channel half) hold partial sums of the SAME block; the upper
    // half hands its nine accumulators (and bias sums) over through the idle stages, so that a workgroup writes one partial block per chunk, not two
    // (halves the partial traffic of the launch and of the reduction stage: 33 -> 16 MB written and read back per dense-block conv at batch 16)
    if (NT == 1) {
        float* xch = (float*)smem + cih * (9 * 16 * 64 + 64);
        __builtin_amdgcn_s_barrier();                             // every consumer wave is past its last fragment read
        if (hi == 1) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) xch[(tap * 16 + reg) * 64 + l] = acc[tap][reg];
            xch[9 * 16 * 64 + l] = bsum;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (hi == 1) return;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc[tap][reg] += xch[(tap * 16 + reg) * 64 + l];
            asm volatile("" : "+v"(acc[tap]));                    // one accumulator's worth of reads in flight: all 144 at once spill
        }
        bsum += xch[9 * 16 * 64 + l];
    }
    // partial block of this wave: [slot][chunk][32-wide group][co_local 32][tap 9][ci_local 32] of its half's conv
    const int64_t slot_eff = slot;
    if (do_bias) {
        const float v = bsum + __shfl_xor(bsum, 32, 64);             // the two lane halves hold disjoint pixels of one channel
        if (lh == 0) R.db_partials[hf][(slot_eff * R.ncg[hf] + R.cg[hf]) * 32 + (l & 31)] = v;
    }
    if (!live) return;
    float* dst = R.partials[hf] + ((slot_eff * R.nchunks[hf] + chunk) * R.ncg[hf] + R.cg[hf]) * (32 * 9 * 32);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int co = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            dst[(co * 9 + tap) * 32 + (l & 31)] = acc[tap][reg];
        }
#endif
}

}  // namespace

namespace tr {

WgRingPlan wgrad_ring_plan(int n, int h, int w, int cin, int cout) {
    WgRingPlan p;
    p.ntx = (w + TW - 1) / TW;
    p.nty = (h + TH - 1) / TH;
    p.ntiles = p.ntx * p.nty * n;
    p.nchunks = (cin + 31) / 32;
    p.npc = (p.nchunks + 1) / 2;
    p.cgw = cout > 32 ? 64 : 32;
    p.ncg = (cout + p.cgw - 1) / p.cgw;
    int target = 256 / (p.npc * p.ncg);                      // one workgroup per CU
    if (target < 1) target = 1;
    p.nslots = p.ntiles < target ? p.ntiles : target;
    if (p.nslots >= 8) p.nslots &= ~7;                       // XCD alignment of the workgroups that walk the same tiles
    p.slots_eff = p.nslots;
    p.dw_floats = (size_t)p.slots_eff * p.nchunks * p.ncg * p.cgw * 9 * 32;
    p.db_floats = (size_t)p.slots_eff * p.ncg * p.cgw;
    return p;
}

// -1: not eligible (the caller takes wgrad_kernel); 0: launched (a.nslots etc. describe the partial layout for the reduction); > 0: launch error
int wgrad3x3_ring(hipStream_t st, const tr_wgrad3x3_desc* d, WgArgs& a, WgRingPlan& p) {
    const int impl = tr::tune().wgrad_impl;                  // tr_tune_set("wgrad_impl", ...): 1 = never ; 2 = whenever eligible ; 0: eligible and large enough
    if (impl == 1) return -1;
    const bool force = impl == 2;
    if (d->ksize != 3 || (d->compute != TR_BF16 && d->compute != TR_F16)) return -1;
    if (d->in_mode == TR_IN_VALID || d->in_mode == TR_IN_FULL) return -1;
    if (d->cin % 8 != 0 || d->cout % 8 != 0) return -1;
    if ((int64_t)a.hin * a.win * d->in.cstride * 2 >= (int64_t)0x7fff0000 || (int64_t)d->h * d->w * d->gy.cstride * 2 >= (int64_t)0x7fff0000) return -1;
    if (!force && (int64_t)d->n * d->h * d->w < 256 * 512) return -1;
    p = wgrad_ring_plan(d->n, d->h, d->w, d->cin, d->cout);
    if (d->partials_bytes < (p.dw_floats + p.db_floats) * sizeof(float)) return -1;
    a.ntx = p.ntx; a.nty = p.nty; a.ntiles = p.ntiles; a.nslots = p.nslots; a.nchunks = p.nchunks; a.ncg = p.ncg;
    a.partials = d->partials;
    a.db_partials = d->db ? d->partials + p.dw_floats : nullptr;
    if (p.npc * p.ncg > TR_WG_MAX_ROLES) return -1;
    WgRoles roles;
    memset(&roles, 0, sizeof(roles));
    roles.nroles = p.npc * p.ncg;
    const int halves = p.cgw / 32;
    for (int cg = 0; cg < p.ncg; ++cg)
        for (int pc = 0; pc < p.npc; ++pc) {
            WgRole& r = roles.r[pc + p.npc * cg];
            r.pc = pc;
            r.in_c = d->in.c;
            for (int hf = 0; hf < halves; ++hf) {
                const int c0 = cg * p.cgw + hf * 32;
                r.gy_coff[hf] = d->gy.coff + c0;
                r.gy_c[hf] = d->gy.c - c0 < 0 ? 0 : (d->gy.c - c0 > 32 ? 32 : d->gy.c - c0);
                r.nchunks[hf] = p.nchunks; r.ncg[hf] = p.ncg * halves; r.cg[hf] = cg * halves + hf;
                r.partials[hf] = a.partials; r.db_partials[hf] = a.db_partials;
            }
        }
    const unsigned grid = (unsigned)(p.nslots * roles.nroles);
#define TR_WGR_LAUNCH(CT, NT)                                                                                           \
    {                                                                                                                   \
        constexpr int lds = NSTAGE * WCfg<NT>::STAGE;                                                                    \
        static_assert(lds <= LDS_MAX, "stages");                                                                        \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            hipFuncSetAttribute((const void*)wgrad_ring_kernel<CT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL((wgrad_ring_kernel<CT, NT>), dim3(grid), dim3((NCW + NLW) * 64), lds, st, a, roles);          \
    }
    if (d->compute == TR_BF16) {
        if (p.cgw == 64) TR_WGR_LAUNCH(TR_BF16, 2) else TR_WGR_LAUNCH(TR_BF16, 1)
    } else {
        if (p.cgw == 64) TR_WGR_LAUNCH(TR_F16, 2) else TR_WGR_LAUNCH(TR_F16, 1)
    }
#undef TR_WGR_LAUNCH
    return tr::launch_status("tr_wgrad3x3(ring)");
}

int wgrad3x3_ring_multi(hipStream_t st, const tr_wgrad3x3_desc* descs, int ndesc, int* nslots_out) {
    const int impl = tr::tune().wgrad_impl;
    if (impl == 1 || !tr::tune().wgrad_multi) return -1;
    const tr_wgrad3x3_desc& d0 = descs[0];
    if (d0.ksize != 3 || (d0.compute != TR_BF16 && d0.compute != TR_F16) || d0.in_mode != TR_IN_DIRECT) return -1;
    if ((int64_t)d0.h * d0.w * d0.in.cstride * 2 >= (int64_t)0x7fff0000 || (int64_t)d0.h * d0.w * d0.gy.cstride * 2 >= (int64_t)0x7fff0000) return -1;
    // no size threshold (round 3): the fused dense-block launch beats five separate weight gradients at every size measured (tools/ab_wgrad.py multi:
    // 8 x 64 x 64 pixels 0.048 vs 0.116 ms per block, 2 x 64 x 64 0.045 vs 0.064, 1 x 24 x 40 0.033 vs 0.054)
    // 32-wide output units (conv, 64-channel input block, group), paired up per input block into 64-wide roles: an x tile staged in LDS serves two
    struct Unit { int k, pc, cg; };
    Unit units[3][2 * TR_WG_MAX_ROLES];
    int nu[3] = {0, 0, 0};
    for (int k = 0; k < ndesc; ++k) {
        const tr_wgrad3x3_desc& d = descs[k];
        // one x buffer and one gy buffer, x slices starting at the same channel; same geometry and arithmetic
        if (d.ksize != 3 || d.compute != d0.compute || d.in_mode != TR_IN_DIRECT || d.pad_mode != d0.pad_mode || d.n != d0.n || d.h != d0.h || d.w != d0.w) return -1;
        if (d.in.ptr != d0.in.ptr || d.in.cstride != d0.in.cstride || d.in.coff != d0.in.coff || d.gy.ptr != d0.gy.ptr || d.gy.cstride != d0.gy.cstride) return -1;
        if (d.cin % 8 != 0 || d.cout % 8 != 0) return -1;
        const int nchunks = (d.cin + 31) / 32, npc = (nchunks + 1) / 2, ncg = (d.cout + 31) / 32;
        if (npc > 3) return -1;
        for (int pc = 0; pc < npc; ++pc)
            for (int cg = 0; cg < ncg; ++cg) {
                if (nu[pc] >= 2 * TR_WG_MAX_ROLES) return -1;
                units[pc][nu[pc]++] = Unit{k, pc, cg};
            }
    }
    const int nr = (nu[0] + 1) / 2 + (nu[1] + 1) / 2 + (nu[2] + 1) / 2;
    if (nr > TR_WG_MAX_ROLES) return -1;
    WgRingPlan g = wgrad_ring_plan(d0.n, d0.h, d0.w, 64, 32);          // tile geometry only
    int nslots = 256 / nr;
    if (nslots > g.ntiles) nslots = g.ntiles;
    if (nslots >= 8) nslots &= ~7;
    if (nslots < 1) return -1;
    // partial blocks: every conv keeps the single-launch layout with this slot count, in its own workspace
    size_t dwf[64];
    for (int k = 0; k < ndesc; ++k) {
        const tr_wgrad3x3_desc& d = descs[k];
        const int nchunks = (d.cin + 31) / 32, ncg = (d.cout + 31) / 32;
        dwf[k] = (size_t)nslots * nchunks * ncg * 32 * 9 * 32;
        const size_t db_floats = (size_t)nslots * ncg * 32;
        if (ndesc > 64 || !d.partials || d.partials_bytes < (dwf[k] + db_floats) * sizeof(float)) return -1;
    }
    WgRoles roles;
    memset(&roles, 0, sizeof(roles));
    roles.nroles = nr;
    int ri = 0;
    for (int pc = 0; pc < 3; ++pc)
        for (int u = 0; u < nu[pc]; u += 2) {
            WgRole& r = roles.r[ri++];
            r.pc = pc;
            for (int hf = 0; hf < 2 && u + hf < nu[pc]; ++hf) {
                const Unit& un = units[pc][u + hf];
                const tr_wgrad3x3_desc& d = descs[un.k];
                const int c0 = un.cg * 32;
                if (d.in.c > r.in_c) r.in_c = d.in.c;
                r.gy_coff[hf] = d.gy.coff + c0;
                r.gy_c[hf] = d.gy.c - c0 > 32 ? 32 : d.gy.c - c0;
                r.nchunks[hf] = (d.cin + 31) / 32; r.ncg[hf] = (d.cout + 31) / 32; r.cg[hf] = un.cg;
                r.partials[hf] = d.partials;
                r.db_partials[hf] = d.db ? d.partials + dwf[un.k] : nullptr;
            }
        }
    WgArgs a;
    memset(&a, 0, sizeof(a));
    a.n = d0.n; a.h = d0.h; a.w = d0.w;
    a.up = 0; a.shift = 0;
    a.ih = d0.h; a.iw = d0.w; a.hin = d0.h; a.win = d0.w;
    a.replicate = d0.pad_mode == TR_PAD_REPLICATE;
    a.in = d0.in; a.gy = d0.gy;
    a.ntx = g.ntx; a.nty = g.nty; a.ntiles = g.ntiles; a.nslots = nslots;
    const unsigned grid = (unsigned)(nslots * roles.nroles);
    constexpr int lds = NSTAGE * WCfg<2>::STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)wgrad_ring_kernel<TR_BF16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute((const void*)wgrad_ring_kernel<TR_F16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    if (d0.compute == TR_BF16) hipLaunchKernelGGL((wgrad_ring_kernel<TR_BF16, 2>), dim3(grid), dim3((NCW + NLW) * 64), lds, st, a, roles);
    else hipLaunchKernelGGL((wgrad_ring_kernel<TR_F16, 2>), dim3(grid), dim3((NCW + NLW) * 64), lds, st, a, roles);
    *nslots_out = nslots;
    return tr::launch_status("tr_wgrad3x3_multi(ring)");
}

}  // namespace tr
