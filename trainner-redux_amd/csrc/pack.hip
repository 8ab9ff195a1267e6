// Weight packing: fp32 OIHW master weights -> MFMA-operand slabs consumed by conv3x3.hip.
//
// Packed layout (bytes), for a conv with (cin, cout) in compute dtype T:
//   [cout group g][K-chunk c][row = tap*CGW + co_local][64 B = CK channels of T]
//   CGW = 32 if cout <= 32 else 64; CK = 32 (bf16) / 16 (f32); inside a row the four 16-byte
//   slots are stored XOR-swizzled: slot s lives at position s ^ ((row >> 2) & 3) so that the LDS
//   image is a straight copy of global memory and MFMA operand reads are bank-conflict free.
// Rows / channels beyond (cout, cin) stay zero: the caller zero-fills the slab once.
//
// transpose != 0 builds the operand of the input-gradient convolution
//   gx[ci] = sum_{co,tap} gy[co] (shifted by -tap) * W[co][ci][tap]
// i.e. a 3x3 conv whose "input channels" are the forward output channels, whose "output channels"
// are the forward input channels, with the taps mirrored (autograd's conv_backward_input).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void pack_kernel(const tr_pack_seg* __restrict__ segs, int nsegs, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // binary search the segment holding flat element i
        int lo = 0, hi = nsegs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (segs[mid].first <= i) lo = mid; else hi = mid - 1;
        }
        const tr_pack_seg s = segs[lo];
        const int kk = s.ksize == 1 ? 1 : 9;
        int64_t r = i - s.first;
        const int tap = (int)(r % kk); r /= kk;
        const int ii = (int)(r % s.i_cnt);
        const int oo = (int)(r / s.i_cnt);
        const float val = s.src[((int64_t)(s.o_off + oo) * s.src_i + (s.i_off + ii)) * kk + tap];
        const int co = s.dst_o_off + (s.transpose ? ii : oo);
        const int ci = s.dst_i_off + (s.transpose ? oo : ii);
        const int tp = s.transpose ? kk - 1 - tap : tap;
        const tr::PackGeom g = tr::pack_geom(s.dst_cin, s.dst_cout, s.compute);
        const int cg = co / g.cgw, col = co - cg * g.cgw;
        const int chunk = ci / g.ck, cil = ci - chunk * g.ck;
        const int row = tp * g.cgw + col;
        char* rowp = (char*)s.dst + (((int64_t)cg * g.nchunks + chunk) * kk * g.cgw + row) * 64;
        const int swz = (row >> 2) & 3;
        if (s.compute == TR_BF16X3) {
            // split row: bytes [0, 32) = bf16 hi of the 16 channels, [32, 64) = bf16 lo (val - hi), same slot swizzle
            const unsigned short hi = tr::f32_to_bf16(val);
            const unsigned short lo = tr::f32_to_bf16(val - tr::bf16_to_f32(hi));
            const int bh = cil * 2, bl = 32 + cil * 2;
            *(unsigned short*)(rowp + ((((bh >> 4) ^ swz) & 3) << 4) + (bh & 15)) = hi;
            *(unsigned short*)(rowp + ((((bl >> 4) ^ swz) & 3) << 4) + (bl & 15)) = lo;
        } else {
            const int slot = cil / g.eps, e = cil - slot * g.eps;
            char* dst = rowp + ((slot ^ swz) << 4);
            if (s.compute != TR_F32) ((unsigned short*)dst)[e] = tr::f32_to_h(val, s.compute);
            else ((float*)dst)[e] = val;
        }
    }
}

}  // namespace

extern "C" int tr_pack_conv3x3(void* stream, const tr_pack_seg* segs_dev, int nsegs, int64_t total_elems) {
    TR_REQUIRE(segs_dev != nullptr && nsegs > 0 && total_elems > 0, TR_EINVAL, "tr_pack_conv3x3: empty segment table");
    int64_t blocks = (total_elems + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, segs_dev, nsegs, total_elems);
    return tr::launch_status("tr_pack_conv3x3");
}
