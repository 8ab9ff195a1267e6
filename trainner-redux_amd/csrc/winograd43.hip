// Winograd F(4x4, 3x3) convolution in fp32 arithmetic on MFMA, fully fused -- the second form of the fp32-forced VGG19 perceptual head
// (reference: traiNNer/losses/perceptual_loss.py:8-12 casts the head's inputs to float32; perceptual_fp16_loss.py:262-329 is the conv / ReLU /
// pool stack).  Y = A^T [ (G g G^T) (.) (B^T d B) ] A with 6 x 6 transformed tiles: 36 multiplies per 4 x 4 outputs instead of 144, i.e. 0.5625 of
// the MFMA cycles of the F(2x2, 3x3) kernel (winograd.hip) and 0.25 of the direct form, at ~2e-6 relative error per layer in fp32 (F(2x2): 4e-7,
// direct: 2e-7; measured against fp64 on VGG-shaped layers) -- every operation is an fp32 add / multiply / MFMA.
//
// Workgroups of 768 threads = 12 waves, one per CU, PERSISTENT over tiles of 64 output channels x 32 Winograd tiles (4 x 8 tiles = 16 x 32 output
// pixels); ids are dealt so that the workgroups of an XCD walk consecutive tiles (the channel groups of a patch share its input in one L2).
// Wave (j, kh) owns column j of the 6 x 6 positions for output-channel half kh: six 32 x 32 accumulators (channel x tile), so the column fold
// A^T M[:, j] of the output transform is register arithmetic.  K loop over 8-channel chunks, one barrier per chunk (in front of position 5):
//   * the transformed weights U never touch LDS: no two waves share an element, so a wave fetches the A fragments of its six positions straight
//     from the packed array (16 bytes per lane and position, contiguous over the wave) into a three-position register ring, two positions ahead;
//   * the raw 18 x 34 x 8-channel halo of a chunk goes global -> LDS by buffer_load ... lds from four loader waves (8-11), four buffers, one request
//     stream over (tile, chunk) that runs four chunks ahead and through the epilogues; only V = B^T d B is computed into LDS (two 36 KB stages):
//     thread (tile, channel, row pair) builds two of the six rows of its 6 x 6 patch transform (compile-time pairs with shared subexpressions:
//     the K loop exists once per pair), spread behind the MFMAs of positions 0-4;
//   * epilogue: column fold in registers (in place), four exchange rounds (channel half x row pair) through [j][r][tile][36-float rows] inside
//     the V stages, row fold Y[r][:] = T[r] A, bias / ReLU / ReLU-derivative mask / optional 2 x 2 max pool, 16-byte buffer stores (8 lanes =
//     the 128 contiguous bytes of a pixel's 32 channels).
// What bounded each step of this structure, measured with the in-kernel stamps: DESIGN.md section 4.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#ifndef TR_W43_ABL
#define TR_W43_ABL 0     // timing experiments: 1 = no MFMAs, 2 = no fill (results are wrong)
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int Q_TB = 32;                       // tiles per workgroup (4 rows x 8 cols of 4 x 4 output pixels)
constexpr int Q_KB = 64;                       // output channels per workgroup
constexpr int Q_CK = 8;                        // input channels per chunk
constexpr int Q_THREADS = 768;
constexpr int QV_FLOATS = 36 * 2 * Q_TB * 4;   // V of one chunk: [xi 36][kk 2][tile 32][s 4] (channel c = 2 s + kk) = 36 KB
constexpr int Q_HR = 18, Q_HC = 34;            // halo rows / columns of the workgroup's input patch
// raw halo of one chunk in LDS: 16-byte slots [row 18][f 68], f = QF_BASE[x & 3] + (x >> 2) * 2 + half (x = halo column, half = channels 0-3 / 4-7):
// the fill role reads 8 tiles (x step 4) x 2 halves per instruction = 16 consecutive slots, every bank once.  The slots are written by LDS-DMA
// (lane-linear: instruction q, lane l -> slot q * 64 + l), 20 instructions per chunk, the last one padded with out-of-range lanes.
constexpr int QRAW_SLOTS = Q_HR * 68;
constexpr int QRAW_DMA = (QRAW_SLOTS + 63) / 64;
constexpr int QRAW_FLOATS = QRAW_DMA * 64 * 4;
__host__ __device__ constexpr int qf_base(int xr) { return xr == 0 ? 0 : xr == 1 ? 18 : xr == 2 ? 36 : 52; }
constexpr unsigned Q_OOB = 0x7fffff00u;        // beyond any num_records: the buffer load returns 0 = the zero padding of the convolution
constexpr int Q_TS = 36;                       // floats per (j, r, tile) row of the exchange buffer
constexpr int QX_FLOATS = 6 * 2 * Q_TB * Q_TS; // one exchange round (channel half x row pair): 55,296 B, inside the two V stages
constexpr int QU_CHUNK = 12 * 6 * 256;         // packed U floats per (group, chunk): [wave 12][i 6][lane 64][e 4]
constexpr int Q_NRAW = 4;                      // raw halo buffers: the DMA of chunk ch + Q_NRAW is issued while chunk ch is multiplied
constexpr int QDN_WORDS = 4 * 5 * 64;            // piece offsets of the tile after the one the halo stream is in: [loader wave 4][piece 5][lane 64]
constexpr int Q_SMEM = (2 * QV_FLOATS + Q_NRAW * QRAW_FLOATS + QDN_WORDS) * 4;
static_assert(QX_FLOATS <= 2 * QV_FLOATS && Q_SMEM <= 160 * 1024, "LDS budget");

struct Wino43Args {
    int n, h, w, cin, cout;
    int act;
    float mask_slope;
    tr_view in, out, mask, pool;
    const float* u;
    const float* bias;
    int nty, ntx, nchunks, nkg;
    long long* stamps;     // diagnostics (builds with -DTR_W43_STAMPS only)
};

// diagnostics: wave-uniform time stamps (s_memtime) of waves 0 (a loader) and 4 into a.stamps[workgroup][role][64], workgroups below 1024
#ifdef TR_W43_STAMPS
#define Q_STAMP()                                                                                                          \
    if (a.stamps && l == 0 && (wv == 0 || wv == 4) && n_stamp < 64 && blockIdx.x < 1024)                                       \
        a.stamps[((int64_t)blockIdx.x * 2 + (wv >> 2)) * 64 + n_stamp++] = (long long)__builtin_amdgcn_s_memtime();
#else
#define Q_STAMP()
#endif
#if defined(TR_W43_STAMPS) && TR_W43_STAMPS >= 2
#define Q_STAMP2() Q_STAMP()        /* epilogue detail: after the fold / DMA requests, and around each round's barriers */
#else
#define Q_STAMP2()
#endif

// B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1] applied to a 6-vector, two rows at a time in the
// pairing that shares subexpressions: P = 0 -> rows (0, 5), 1 -> rows (1, 2), 2 -> rows (3, 4); 4 / 6 / 4 operations per pair.
template <int P>
__device__ __forceinline__ void bt_pair(const float (&d)[6], float& ra, float& rb) {
    if constexpr (P == 0) {
        ra = fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4]));
        rb = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
    } else if constexpr (P == 1) {
        ra = fmaf(-4.f, d[1] + d[2], d[4] + d[3]);
        rb = fmaf(4.f, d[1] - d[2], d[4] - d[3]);
    } else {
        const float u = d[3] - d[1], v = d[4] - d[2];
        ra = fmaf(2.f, u, v);
        rb = fmaf(-2.f, u, v);
    }
}
__host__ __device__ constexpr int bt_row_a(int p) { return p == 0 ? 0 : p == 1 ? 1 : 3; }
__host__ __device__ constexpr int bt_row_b(int p) { return p == 0 ? 5 : p == 1 ? 2 : 4; }
template <int V> struct QInt { static constexpr int value = V; };

// MASK: the input-gradient form (ReLU-derivative mask on the output).  A template parameter because the two forms spend the epilogue's registers
// differently: the masked form holds the round's four mask vectors across the exchange barrier, the plain form holds both bias vectors.
// POOL: the epilogue also writes the 2x2 max pool of ReLU(out) at half resolution (VGG19's conv -> ReLU -> MaxPool pairs: the pool kernel would read
// the whole activation back).
template <bool MASK, bool POOL>
__global__ __launch_bounds__(Q_THREADS, 1) void winograd43_f32_kernel(Wino43Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* vst = (float*)smem;                       // two V stages; the exchange rounds of the epilogue alias them
    float* rawb = vst + 2 * QV_FLOATS;               // Q_NRAW raw halo buffers: the halo stream runs on across tiles and through the epilogue
    const int tid = threadIdx.x;
    const int l = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = l & 31, lh = l >> 5;
    const int wj = wv % 6, wkh = wv / 6;
    int n_stamp = 0;
    (void)n_stamp;
    Q_STAMP()

    // ---- persistent tile walk: workgroup b takes tiles first + k * grid, output-channel group fastest.  Ids are dealt so that the workgroups of
    // one XCD (ids equal mod 8 under round-robin placement: speed only) own a run of consecutive tiles: the channel groups of a patch, which read
    // the same input, and horizontally adjacent patches, whose halos overlap, meet in that XCD's L2.
    const int G = gridDim.x;
    const int ntiles = a.n * a.nty * a.ntx * a.nkg;
    int first = blockIdx.x;
    if ((G & 7) == 0) first = (first & 7) * (G >> 3) + (first >> 3);
    // tile index -> (channel group, tile column, tile row, image): decoded once (runtime divisions cost ~50 instructions each), then advanced by the
    // grid size digit by digit
    struct TilePos { int kg, bx, by, nb; };
    auto decode = [&](int t) {
        TilePos p;
        p.kg = t % a.nkg; t /= a.nkg;
        p.bx = t % a.ntx; t /= a.ntx;
        p.by = t % a.nty;
        p.nb = t / a.nty;
        return p;
    };
    const TilePos gstep = decode(G);
    auto advance = [&](TilePos p) {                  // p + G in the mixed radix (nkg, ntx, nty, .)
        p.kg += gstep.kg;
        int c = p.kg >= a.nkg;
        p.kg -= c ? a.nkg : 0;
        p.bx += gstep.bx + c;
        c = p.bx >= a.ntx;
        p.bx -= c ? a.ntx : 0;
        p.by += gstep.by + c;
        c = p.by >= a.nty;
        p.by -= c ? a.nty : 0;
        p.nb += gstep.nb + c;
        return p;
    };

    // ---- loader role (waves 8-11, one per SIMD; they hold no global loads or stores of the epilogue): the 18 x 34 x 8-channel halo of a chunk as
    // 1224 16-byte pieces, five LDS-DMA instructions per wave.  Only these four waves have the halo's memory latency in their (in-order) vmcnt
    // queue in front of the weight prefetches; with the loads spread over all twelve waves every wave stalled on it once per chunk at the same
    // point and the MFMA pipe drained (-20 %).  The requests form ONE stream over (tile, chunk) that runs Q_NRAW chunks ahead of the MFMAs and
    // does not stop at tile boundaries: the first chunks of the next tile are requested during this tile's last K iterations and land before /
    // while its epilogue runs (the exchange rounds stay inside the V stages, the raw buffers are not aliased).  Requesting them in the epilogue
    // instead held its first barrier for up to ~9.7k ticks (20 pieces per wave against the memory pipeline's back-pressure).  The piece offsets
    // of a tile cost ~100 VALU with temporaries the K loop has no registers for: they are computed in the epilogue one tile ahead and parked in
    // LDS; the stream's switch to the next tile inside the K loop is five LDS reads into the registers of the offsets it has finished with.
    const bool loader = wv >= 8;
    const int img_bytes = a.h * a.w * a.in.cstride * 4;
    int d_tile = first, d_ch = 0, d_buf = 0;         // halo stream: next chunk to request (tile, chunk) and the raw buffer it goes to
    TilePos d_pos = decode(first);                   // position of d_tile ; d_pos2: of d_tile + G (whose offsets are parked in LDS)
    TilePos d_pos2 = advance(d_pos);
    __amdgpu_buffer_rsrc_t d_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.in.ptr, 0, 0, 0x00020000);
    unsigned dmaoff[5] = {Q_OOB, Q_OOB, Q_OOB, Q_OOB, Q_OOB};
    unsigned* const dnext0 = (unsigned*)(rawb + Q_NRAW * QRAW_FLOATS) + (wv & 3) * 5 * 64;     // + k * 64 + lane
    auto set_rsrc = [&](const TilePos& p) {
        d_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in.ptr + (int64_t)p.nb * img_bytes + a.in.coff * 4), 0, (unsigned)img_bytes, 0x00020000);
    };
    auto piece_offset = [&](const TilePos& p, int k, int le) -> unsigned {    // loader waves: buffer offset of DMA piece k of the tile at p
        const int slot = ((wv & 3) * 5 + k) * 64 + le;
        const int hy = slot / 68, f = slot - hy * 68;
        const int xr = f < 18 ? 0 : f < 36 ? 1 : f < 52 ? 2 : 3;
        const int r = f - qf_base(xr);
        const int hx = (r >> 1) * 4 + xr, half = r & 1;
        const int y = p.by * 16 - 1 + hy, x = p.bx * 32 - 1 + hx;
        const bool ok = slot < QRAW_SLOTS && y >= 0 && y < a.h && x >= 0 && x < a.w;
        return ok ? (unsigned)(((y * a.w + x) * a.in.cstride + half * 4) * 4) : Q_OOB;
    };
    auto park_next = [&](int k0, int k1) {           // loader waves: LDS <- offsets of pieces [k0, k1) of the tile after the stream's current one
        if (d_tile + G < ntiles) {
            int le = l;                              // opaque copy: the slot arithmetic is redone per call instead of being hoisted out of the tile
            asm volatile("" : "+v"(le));             // loop and kept (spilled) across the K loop
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (k >= k0 && k < k1) {
                    unsigned o = piece_offset(d_pos2, k, le);
                    dnext0[k * 64 + le] = o;
                    asm volatile("" ::: "memory");   // one piece at a time: no array of temporaries
                }
        }
    };
    auto stream_issue = [&]() -> bool {              // loader waves (wave-uniform): request the next chunk of the stream, if one is left
        if (d_tile >= ntiles) return false;
        float* rb = rawb + d_buf * QRAW_FLOATS;
#pragma unroll
        for (int k = 0; k < 5; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rsrc, (__attribute__((address_space(3))) void*)(rb + ((wv & 3) * 5 + k) * 256), 16, dmaoff[k],
                                                     d_ch * (Q_CK * 4), 0, 0);
        d_buf = d_buf == Q_NRAW - 1 ? 0 : d_buf + 1;
        if (++d_ch == a.nchunks) {                   // on to the next tile: its offsets were parked one epilogue ago (nchunks > Q_NRAW: host check)
            d_ch = 0;
            d_tile += G;
            d_pos = d_pos2;
            d_pos2 = advance(d_pos2);
            if (d_tile < ntiles) {
                set_rsrc(d_pos);
                int le = l;
                asm volatile("" : "+v"(le));
#pragma unroll
                for (int k = 0; k < 5; ++k) dmaoff[k] = dnext0[k * 64 + le];
            }
        }
        return true;
    };

    // ---- fill role: (tile ft, channel fc) of the chunk, one row pair of its 6 x 6 transform B^T d B (pair fp is uniform per wave and a
    // compile-time constant of the tile loop below: the loop exists three times and a wave enters its own copy once) ----------------------------
    const int fc = tid & 7, ft = (tid >> 3) & 31;
    const int fp = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int rsrc = (((ft >> 3) * 4) * 68 + (ft & 7) * 2 + (fc >> 2)) * 4 + (fc & 3);   // top-left value of this thread's patch in a raw buffer
    const int vdst = (((fc & 1) * Q_TB) + ft) * 4 + (fc >> 1);                       // + xi * 256 floats within a stage

    auto tile_loop = [&](auto fp_c) {
        constexpr int FP = decltype(fp_c)::value;
        constexpr int RA = bt_row_a(FP), RB = bt_row_b(FP);
        float ta[6], tb[6];
        auto fill_col = [&](const float* rb, int col) {      // column `col` of the patch -> entry `col` of the two B^T d rows
            float d[6];
#pragma unroll
            for (int k = (FP == 0 ? 0 : 1); k < (FP == 0 ? 6 : 5); ++k) d[k] = rb[rsrc + (k * 68 + qf_base(col & 3) + (col >> 2) * 2) * 4];
            if (FP != 0) d[0] = d[5] = 0.f;                  // rows 1-4 of B^T do not touch d0 / d5
            bt_pair<FP>(d, ta[col], tb[col]);
        };
        auto fill_out = [&](float* st, int pair) {           // (B^T d) B: the output-column pair of both rows, into the stage
            const int ja = bt_row_a(pair), jb = bt_row_b(pair);
            float oa, ob;
            if (pair == 0) bt_pair<0>(ta, oa, ob); else if (pair == 1) bt_pair<1>(ta, oa, ob); else bt_pair<2>(ta, oa, ob);
            st[vdst + (RA * 6 + ja) * 256] = oa;
            st[vdst + (RA * 6 + jb) * 256] = ob;
            if (pair == 0) bt_pair<0>(tb, oa, ob); else if (pair == 1) bt_pair<1>(tb, oa, ob); else bt_pair<2>(tb, oa, ob);
            st[vdst + (RB * 6 + ja) * 256] = oa;
            st[vdst + (RB * 6 + jb) * 256] = ob;
        };

        f32x4 ureg[3];                                       // ring: position p lives in ureg[p % 3] and is requested two positions ahead of its MFMAs
        int cbuf = 0;                                        // raw buffer of the next chunk (in stream order) the fill role consumes
        const int uoff = l * 4;
        {
            const float* ub = a.u + ((int64_t)d_pos.kg * a.nchunks * 12 + wv) * (6 * 256);
            ureg[0] = *(const f32x4*)(ub + uoff);
            ureg[1] = *(const f32x4*)(ub + 256 + uoff);
            ureg[2] = *(const f32x4*)(ub + 512 + uoff);
            asm volatile("" ::: "memory");                   // the two fragments stay in front of the halo requests in the queue, as on later tiles
            if (loader) {
#pragma unroll
                for (int k = 0; k < 5; ++k) dmaoff[k] = piece_offset(d_pos, k, l);
                set_rsrc(d_pos);
                park_next(0, 5);
                for (int c = 0; c < Q_NRAW; ++c) stream_issue();
            }
        }
        TilePos cpos = decode(first);                        // the tile being multiplied
        for (int tile = first; tile < ntiles; tile += G) {
            const int kg = cpos.kg, bx = cpos.bx, by = cpos.by, nb = cpos.nb;
            cpos = advance(cpos);
            // packed weights of this wave: wave-uniform base (+ chunk * QU_CHUNK + i * 256) + the lane's 4 floats
            const float* ub = a.u + ((int64_t)kg * a.nchunks * 12 + wv) * (6 * 256);
            const float* ubn = ub;                                                                // chunk 0 of the next tile (clamped at the end)
            if (tile + G < ntiles) ubn = a.u + ((int64_t)cpos.kg * a.nchunks * 12 + wv) * (6 * 256);
            f32x16 acc[6];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

            // ---- V of chunk 0.  First tile: the queue of a loader wave holds [3 weight fragments, Q_NRAW halo chunks of 5 pieces] (loads retire in
            // order): chunk 0 has landed when only the later chunks are in flight.  Later tiles: their first chunks were requested during the
            // previous tile's last K iterations and checked off there (the loop's rule: two chunks ahead have landed at every chunk barrier), and
            // every wave has passed the previous epilogue's last barrier.
            if (tile == first) {
                if (loader) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            Q_STAMP()
            {
                const float* rb0 = rawb + cbuf * QRAW_FLOATS;
                cbuf = cbuf == Q_NRAW - 1 ? 0 : cbuf + 1;
#pragma unroll
                for (int c = 0; c < 6; ++c) fill_col(rb0, c);
#pragma unroll
                for (int pr = 0; pr < 3; ++pr) fill_out(vst, pr);
            }
            if (tile == first && loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the other chunks requested up front
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            Q_STAMP()
            f32x4 bv[2];
            bv[0] = *(const f32x4*)(vst + (lh * Q_TB + lr) * 4 + wj * 256);
            for (int ch = 0; ch < a.nchunks; ++ch) {
                const float* vl = vst + (ch & 1) * QV_FLOATS + (lh * Q_TB + lr) * 4;          // + xi * 256
                float* nst = vst + ((ch + 1) & 1) * QV_FLOATS;
                const float* rb = rawb + cbuf * QRAW_FLOATS;                                  // raw halo of the next chunk in stream order
                if (ch + 1 < a.nchunks) cbuf = cbuf == Q_NRAW - 1 ? 0 : cbuf + 1;             // (a tile's last chunk builds an unused stage from it)
                const float* uc = ub + (int64_t)ch * QU_CHUNK;
                const float* un = ch + 1 < a.nchunks ? uc + QU_CHUNK : ubn;
                bool issued = false;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    if (i == 5) {
                        // The chunk's barrier sits in front of position 5, not behind it: every fill of the next stage is done by position 4, and
                        // position 5 needs nothing from LDS any more (its B fragment is in registers), so the MFMA pipe has work the moment the
                        // barrier releases and the first fragment read of the next stage hides behind these four MFMAs.
                        // Loads retire in order: the halo chunk two ahead in the stream (requested Q_NRAW - 2 iterations ago, or up front) has
                        // landed when only the loads behind it in the queue are still in flight: 5 weight fragments of its own iteration +
                        // (1 + 5 DMA pieces + 5) per later iteration, less this iteration's last fragment which is requested after the barrier.
                        if (loader) {
                            static_assert(Q_NRAW == 4, "vmcnt bookkeeping below; the epilogue spreads the next tile's first Q_NRAW chunks over its 4 rounds");
                            if (issued) {
                                asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
                            } else {
                                asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                            }
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        Q_STAMP()
                        __builtin_amdgcn_s_barrier();
                        Q_STAMP()
                        bv[0] = *(const f32x4*)(nst + (lh * Q_TB + lr) * 4 + wj * 256);       // position 0 of the next chunk
                    } else {
                        bv[(i + 1) & 1] = *(const f32x4*)(vl + ((i + 1) * 6 + wj) * 256);
                    }
                    // weights two positions ahead (positions 4 and 5 request positions 0 and 1 of the next chunk, or of the next tile's first).
                    // Position 2 of a tile's FIRST chunk is already there, requested before the previous tile's epilogue: a finishing wave's loads
                    // queue behind its output stores (one vmcnt counter, retired in order), so the first fragment requested after the epilogue
                    // should be needed as late as possible; this way it is position 3, needed three positions into the loop (the acknowledgement
                    // of the stores takes ~10k ticks, so the tile's first chunk still runs 4-5k ticks long; peeling that chunk to start from C = 0
                    // and hold positions 3 and 4 in the idle accumulator registers as well cost more in spills than it hid)
                    if (i == 0) {
                        if (ch != 0) ureg[2] = *(const f32x4*)(uc + 2 * 256 + uoff);
                    } else {
                        ureg[(i + 2) % 3] = i + 2 < 6 ? *(const f32x4*)(uc + (i + 2) * 256 + uoff) : *(const f32x4*)(un + (i + 2 - 6) * 256 + uoff);
                    }
#if !(TR_W43_ABL & 1)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ureg[i % 3][e], bv[i & 1][e], acc[i], 0, 0, 0);
#else
                    acc[i][0] += ureg[i % 3][0] * bv[i & 1][0] + ureg[i % 3][1] * bv[i & 1][1] + ureg[i % 3][2] * bv[i & 1][2] + ureg[i % 3][3] * bv[i & 1][3];
#endif
                    // the next stage's fill rides behind the MFMAs: three patch columns behind each of positions 0 and 1, one output-column pair
                    // behind each of positions 2-4
#if !(TR_W43_ABL & 2)
                    // (on a tile's last chunk the fill builds an unused stage from a stale raw buffer: cheaper than branches around it)
                    if (i < 2) {
                        fill_col(rb, 3 * i);
                        fill_col(rb, 3 * i + 1);
                        fill_col(rb, 3 * i + 2);
                    } else if (i < 5) {
                        fill_out(nst, i - 2);
                    }
#endif
                    // the stream's next chunk (Q_NRAW ahead of the one being multiplied, possibly in the next tile) goes to the raw buffer whose
                    // chunk -- this one -- was consumed an iteration ago: three chunks of lead (a halo chunk is 4-5 us away under load: start ->
                    // first halo landed = 11-15k ticks in the stamps)
                    if (i == 0 && loader) issued = stream_issue();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            ureg[2] = *(const f32x4*)(ubn + 2 * 256 + uoff);                                     // position 2 of the next tile's first chunk
            // the exchange rounds below alias the stages: every wave is past its last fragment read (and every raw buffer has been consumed)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            Q_STAMP()

            // ---- output transform + epilogue ----------------------------------------------------------------------------------------------
            // bias of the two channel halves, requested ahead of the rounds that add it (finishing threads: channel quad ek4 = (tid & 7) * 4)
            f32x4 bias2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (!MASK && a.bias && tid < 512) {
                bias2[0] = *(const f32x4*)(a.bias + kg * Q_KB + (tid & 7) * 4);
                bias2[1] = *(const f32x4*)(a.bias + kg * Q_KB + 32 + (tid & 7) * 4);
            }
            // column fold T[r] = sum_i A^T[r][i] M[i][j], A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1], in place: acc[r] <- T[r]
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float a0 = acc[0][e], a1 = acc[1][e], a2 = acc[2][e], a3 = acc[3][e], a4 = acc[4][e], a5 = acc[5][e];
                const float s12 = a1 + a2, d12 = a1 - a2, s34 = a3 + a4, d34 = a3 - a4;
                float t0 = a0 + s12 + s34, t1 = d12 + 2.f * d34, t2 = s12 + 4.f * s34, t3 = d12 + 8.f * d34 + a5;
                // pinned here: left to itself the compiler sinks this arithmetic to the exchange rounds and keeps all six accumulators alive
                asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
                acc[0][e] = t0;
                acc[1][e] = t1;
                acc[2][e] = t2;
                acc[3][e] = t3;
            }
            // the next tile's first halo chunks land during this tile's epilogue (requested here, behind the fold: a third of the accumulator
            // registers has just been freed for the offset arithmetic)
            // four exchange rounds (channel half x row pair) through [j 6][row 2][tile 32][36-float rows] = 54 KB inside the stages, so that the
            // raw buffers keep the next tile's first halo chunks; then the row fold Y[r][:] = T[r] A, bias / ReLU / ReLU-derivative mask, 16-byte
            // stores (8 lanes = the 128 contiguous bytes of a pixel's 32 channels).  Raw barriers with LDS-only waits: the stores drain behind.
            float* tl = (float*)smem;
            // finishing role (threads < 512): channel quad, row of the round's pair, tile.  Stores and mask loads go through buffer resources of
            // this image: 32-bit offsets, out-of-range rows / columns of partial tiles are dropped by the bounds check (no branches)
            // (thread-id terms are re-derived behind an opaque copy of tid: hoisted out of the tile loop they would have to live across the K loop,
            // where every register is taken, and came back from scratch at the latency of a memory access per round)
            int te = tid;
            asm volatile("" : "+v"(te));
            const int ek4 = (te & 7) * 4, erl = (te >> 3) & 1, et = te >> 4;
            const int elr = te & 31, elh = (te >> 5) & 1;
            const int out_img = a.h * a.w * a.out.cstride * 4, mask_img = a.h * a.w * a.mask.cstride * 4;
            const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((char*)a.out.ptr + (int64_t)nb * out_img + (a.out.coff + kg * Q_KB) * 4), 0, (unsigned)out_img, 0x00020000);
            const __amdgpu_buffer_rsrc_t mask_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((char*)a.mask.ptr + (int64_t)nb * mask_img + (a.mask.coff + kg * Q_KB) * 4), 0, (unsigned)mask_img, 0x00020000);
            const int ox0 = bx * 32 + (et & 7) * 4, oy0 = by * 16 + (et >> 3) * 4 + erl;
            const int pool_img = (a.h >> 1) * (a.w >> 1) * a.pool.cstride * 4;
            const __amdgpu_buffer_rsrc_t pool_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((char*)a.pool.ptr + (int64_t)nb * pool_img + (a.pool.coff + kg * Q_KB) * 4), 0, POOL ? (unsigned)pool_img : 0u, 0x00020000);
            Q_STAMP2()
#pragma unroll
            for (int round = 0; round < 4; ++round) {
                const int rkh = round >> 1, rp = round & 1;
                // finishing threads request the round's ReLU-derivative mask values first: they arrive while the exchange goes through LDS
                const int oy = oy0 + rp * 2;
                u32x4 mu[4];
                auto mask_load = [&](int sx) {
                    const bool ok = oy < a.h && ox0 + sx < a.w;
                    mu[sx] = __builtin_amdgcn_raw_buffer_load_b128(mask_rsrc, ok ? ((oy * a.w + ox0 + sx) * a.mask.cstride + rkh * 32 + ek4) * 4 : (int)Q_OOB, 0, 0);
                };
                if (MASK && te < 512) {                               // (two of the four: all four cost the K loop its last registers)
                    mask_load(0);
                    mask_load(1);
                }
                if (wkh == rkh) {
#pragma unroll
                    for (int rl = 0; rl < 2; ++rl)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x16& t = acc[rp * 2 + rl];
                            const f32x4 v = {t[q * 4], t[q * 4 + 1], t[q * 4 + 2], t[q * 4 + 3]};
                            *(f32x4*)(tl + ((wj * 2 + rl) * Q_TB + elr) * Q_TS + q * 8 + elh * 4) = v;
                        }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                Q_STAMP2()
                __builtin_amdgcn_s_barrier();
                Q_STAMP2()
                // loader waves (which finish nothing: threads >= 512, and write nothing in the first two rounds) use those rounds' finishing phases
                // to park the piece offsets of the tile after the stream's current one
                if (round == 0 && loader) park_next(0, 3);
                if (round == 1 && loader) park_next(3, 5);
                if (te < 512) {
                    if (MASK) {
                        mask_load(2);
                        mask_load(3);
                    }
                    const int kc = rkh * 32 + ek4;                    // channel within the workgroup's 64
                    f32x4 bias = bias2[rkh];
                    if (MASK && a.bias) bias = *(const f32x4*)(a.bias + kg * Q_KB + kc);
                    const float* tp = tl + (erl * Q_TB + et) * Q_TS + ek4;
                    f32x4 T[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) T[j] = *(const f32x4*)(tp + j * 2 * Q_TB * Q_TS);
                    const f32x4 s12 = T[1] + T[2], d12 = T[1] - T[2], s34 = T[3] + T[4], d34 = T[3] - T[4];
                    const f32x4 y[4] = {T[0] + s12 + s34, d12 + 2.f * d34, s12 + 4.f * s34, d12 + 8.f * d34 + T[5]};
                    f32x4 pm[2];                                      // POOL: max of ReLU over the thread's two pixel pairs
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) {
                        const int ox = ox0 + sx;
                        const bool ok = oy < a.h && ox < a.w;
                        const int pixi = oy * a.w + ox;
                        f32x4 v = y[sx] + bias;
                        if (a.act == TR_ACT_RELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        if (MASK) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= __uint_as_float(mu[sx][e]) > 0.f ? 1.f : a.mask_slope;
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])},
                                                               out_rsrc, ok ? (pixi * a.out.cstride + kc) * 4 : (int)Q_OOB, 0, 0);
                        if (POOL) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float r = fmaxf(v[e], 0.f);
                                pm[sx >> 1][e] = (sx & 1) ? fmaxf(pm[sx >> 1][e], r) : r;
                            }
                        }
                    }
                    if (POOL) {
                        // the other row of the 2 x 2 windows is held by the thread whose row bit (tid bit 3) differs: same wave, lane ^ 8
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) pm[j][e] = fmaxf(pm[j][e], __shfl_xor(pm[j][e], 8, 64));
                        if (erl == 0) {
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const bool okp = oy < a.h && ox0 + 2 * j < a.w;
                                const int pp = (oy >> 1) * (a.w >> 1) + ((ox0 >> 1) + j);
                                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(pm[j][0]), __float_as_uint(pm[j][1]), __float_as_uint(pm[j][2]), __float_as_uint(pm[j][3])},
                                                                       pool_rsrc, okp ? (pp * a.pool.cstride + kc) * 4 : (int)Q_OOB, 0, 0);
                            }
                        }
                    }
                }
                Q_STAMP2()
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                Q_STAMP2()
            }
            Q_STAMP()
        }
    };
    if (fp == 0) tile_loop(QInt<0>{});
    else if (fp == 1) tile_loop(QInt<1>{});
    else tile_loop(QInt<2>{});
}

// U = G g G^T, G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1], for g = w[co][ci] (or, transposed, the
// input-gradient kernel w[ci][co] with its taps mirrored), in the kernel's fragment order [co / 64][ci / 8][wave = (co % 64 / 32) * 6 + j][i]
// [lane = (ci & 1) * 32 + co % 32][e = (ci % 8) >> 1].
__global__ __launch_bounds__(256) void winograd43_pack_kernel(const float* __restrict__ w, float* __restrict__ dst, int cout, int cin, int src_i, int transpose) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)cout * cin) return;
    const int co = (int)(idx / cin), ci = (int)(idx % cin);
    float g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = transpose ? w[((int64_t)ci * src_i + co) * 9 + (8 - t)] : w[((int64_t)co * src_i + ci) * 9 + t];
    const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6}, {1.f / 24, 1.f / 12, 1.f / 6},
                           {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    float gg[6][3];                                          // G g
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int x = 0; x < 3; ++x) gg[i][x] = G[i][0] * g[0 + x] + G[i][1] * g[3 + x] + G[i][2] * g[6 + x];
    const int nchunks = cin / Q_CK, cc = ci % Q_CK, col = co % Q_KB;
    const int lane = (cc & 1) * 32 + (col & 31), e = cc >> 1, kh = col >> 5;
    float* base = dst + ((int64_t)(co / Q_KB) * nchunks + ci / Q_CK) * QU_CHUNK + lane * 4 + e;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) base[((kh * 6 + j) * 6 + i) * 256] = gg[i][0] * G[j][0] + gg[i][1] * G[j][1] + gg[i][2] * G[j][2];
}

inline bool q_view_ok(const tr_view& v, int c) { return v.ptr && v.dtype == TR_F32 && v.c >= c && v.coff >= 0 && v.coff + c <= v.cstride; }

}  // namespace

extern "C" size_t tr_winograd43_packed_bytes(int cin, int cout) { return (size_t)cin * cout * 36 * sizeof(float); }

extern "C" int tr_pack_winograd43(void* stream, const float* w_oihw, float* dst, int cout, int cin, int transpose) {
    TR_REQUIRE(w_oihw && dst && cout > 0 && cin > 0 && cout % Q_KB == 0 && cin % Q_CK == 0, TR_EINVAL,
               "tr_pack_winograd43: cout=%d must be a multiple of %d and cin=%d of %d", cout, Q_KB, cin, Q_CK);
    TR_REQUIRE(((uintptr_t)dst) % 16 == 0, TR_EALIGN, "tr_pack_winograd43: dst must be 16-byte aligned");
    const int64_t n = (int64_t)cout * cin;
    // the OIHW source is cout x cin (direct) or cin x cout (transpose: the input-gradient conv of that weight)
    hipLaunchKernelGGL(winograd43_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, dst, cout, cin,
                       transpose ? cout : cin, transpose ? 1 : 0);
    return tr::launch_status("tr_pack_winograd43");
}

extern "C" int tr_conv3x3_winograd43_f32(void* stream, const tr_conv3x3_desc* d) {
    TR_REQUIRE(d != nullptr, TR_EINVAL, "tr_conv3x3_winograd43_f32: null descriptor");
    TR_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->h % 4 == 0 && d->w % 4 == 0, TR_EUNSUPPORTED,
               "tr_conv3x3_winograd43_f32: needs output dims that are multiples of 4 (got %dx%d)", d->h, d->w);
    TR_REQUIRE(d->cin > Q_NRAW * Q_CK && d->cin % Q_CK == 0 && d->cout > 0 && d->cout % Q_KB == 0, TR_EUNSUPPORTED,
               "tr_conv3x3_winograd43_f32: cin=%d must be a multiple of %d above %d and cout=%d a multiple of %d", d->cin, Q_CK, Q_NRAW * Q_CK, d->cout, Q_KB);
    TR_REQUIRE(d->ksize == 3 && d->in_mode == TR_IN_DIRECT && d->pad_mode == TR_PAD_ZEROS, TR_EUNSUPPORTED,
               "tr_conv3x3_winograd43_f32: 3x3, direct input, zero padding only");
    TR_REQUIRE(d->act == TR_ACT_NONE || d->act == TR_ACT_RELU, TR_EUNSUPPORTED, "tr_conv3x3_winograd43_f32: activation %d is not built (none, ReLU)", d->act);
    TR_REQUIRE(!d->res1.ptr && !d->res2.ptr && !d->sample_scale && d->mask_mode == 0, TR_EUNSUPPORTED,
               "tr_conv3x3_winograd43_f32: residual / per-sample scale / activation-derivative masks are not built");
    // out2 of this entry point: the 2x2 max pool of ReLU(out) at half resolution (fp32, cout channels), plain form only
    TR_REQUIRE(!d->out2.ptr || (!d->mask.ptr && d->out2_scale == 1.f && q_view_ok(d->out2, d->cout)), TR_EUNSUPPORTED,
               "tr_conv3x3_winograd43_f32: out2 (pooled output) needs an fp32 view of cout channels, scale 1, and no mask");
    TR_REQUIRE(q_view_ok(d->in, d->cin) && q_view_ok(d->out, d->cout) && (!d->mask.ptr || q_view_ok(d->mask, d->cout)) && d->wpacked, TR_EINVAL,
               "tr_conv3x3_winograd43_f32: fp32 views holding cin / cout channels and packed weights are required");
    TR_REQUIRE(((uintptr_t)d->wpacked) % 16 == 0, TR_EALIGN, "tr_conv3x3_winograd43_f32: wpacked must be 16-byte aligned");
    auto al = [](const tr_view& v) { return !v.ptr || (v.cstride % 4 == 0 && v.coff % 4 == 0 && ((uintptr_t)v.ptr) % 16 == 0); };
    TR_REQUIRE(al(d->in) && al(d->out) && al(d->mask) && al(d->out2), TR_EALIGN, "tr_conv3x3_winograd43_f32: views must allow aligned 4-channel access");
    TR_REQUIRE((int64_t)d->h * d->w * d->in.cstride * 4 < (int64_t)Q_OOB && (int64_t)d->h * d->w * d->out.cstride * 4 < (int64_t)Q_OOB &&
                   (!d->mask.ptr || (int64_t)d->h * d->w * d->mask.cstride * 4 < (int64_t)Q_OOB),
               TR_EUNSUPPORTED, "tr_conv3x3_winograd43_f32: one image of the input / output / mask must stay below 2 GB (32-bit buffer offsets)");
    Wino43Args a;
    a.n = d->n; a.h = d->h; a.w = d->w; a.cin = d->cin; a.cout = d->cout;
    a.act = d->act; a.mask_slope = d->mask_slope;
    a.in = d->in; a.out = d->out; a.mask = d->mask; a.pool = d->out2;
    a.u = (const float*)d->wpacked; a.bias = d->bias;
    a.nty = (d->h + 15) / 16; a.ntx = (d->w + 31) / 32;
    a.nchunks = d->cin / Q_CK; a.nkg = d->cout / Q_KB;
#ifdef TR_W43_STAMPS
    a.stamps = (long long*)tr::tune().stamps;   // development builds only: tr_tune_set("stamps", device address)
#else
    a.stamps = nullptr;
#endif
    const int64_t nblk = (int64_t)d->n * a.nty * a.ntx * a.nkg;
    TR_REQUIRE(nblk < (int64_t)0x7fffffff, TR_EUNSUPPORTED, "tr_conv3x3_winograd43_f32: grid too large");
    static const hipError_t attr = [] {
        const hipError_t e0 = hipFuncSetAttribute((const void*)winograd43_f32_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Q_SMEM);
        const hipError_t e1 = hipFuncSetAttribute((const void*)winograd43_f32_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Q_SMEM);
        const hipError_t e2 = hipFuncSetAttribute((const void*)winograd43_f32_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Q_SMEM);
        return e0 != hipSuccess ? e0 : e1 != hipSuccess ? e1 : e2;
    }();
    TR_REQUIRE(attr == hipSuccess, TR_EUNSUPPORTED, "tr_conv3x3_winograd43_f32: the device does not grant %d bytes of LDS per workgroup", Q_SMEM);
    static const int ncu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    const unsigned grid = (unsigned)(nblk < ncu ? nblk : ncu);          // persistent: one workgroup per CU walks its tiles
    if (a.mask.ptr) hipLaunchKernelGGL((winograd43_f32_kernel<true, false>), dim3(grid), dim3(Q_THREADS), Q_SMEM, (hipStream_t)stream, a);
    else if (a.pool.ptr) hipLaunchKernelGGL((winograd43_f32_kernel<false, true>), dim3(grid), dim3(Q_THREADS), Q_SMEM, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((winograd43_f32_kernel<false, false>), dim3(grid), dim3(Q_THREADS), Q_SMEM, (hipStream_t)stream, a);
    return tr::launch_status("tr_conv3x3_winograd43_f32");
}
