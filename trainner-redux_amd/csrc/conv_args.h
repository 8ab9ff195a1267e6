// Kernel-side argument block of tr_conv3x3, shared by the register-staged kernels (conv3x3.hip) and the LDS-DMA ring kernel (conv_ring.hip).
#pragma once
#include "common.h"

struct ConvArgs {
    int n, h, w, hin, win;
    int ih, iw, shift;   // input extent in output-resolution coordinates, and the origin shift of the 'full' mode
    int cin, cout;
    int up;            // 0/1: nearest x2 upsample fused into the load
    int replicate;
    int act;
    float slope, scale1, scale2, rs1, rs2, mask_slope, out2_scale;
    int mask_mode;                 // 0: multiply by (m > 0 ? 1 : mask_slope) ; 1 / 2 / 3: multiply by gelu'(m) / silu'(m) / mish'(m)
    const float* sample_scale;     // optional n floats: v *= sample_scale[image] before the first residual (DropPath keep mask)
    tr_view in, out, out2, res1, res2, mask;
    const char* wpacked;
    const float* bias;
    int ntx, nty, nchunks, ncg;
    int vec_out, vec_out2, vec_res1, vec_res2, vec_mask;
    int ring_mb;         // conv_ring.hip, pipelined mask instances: number of LDS mask-tile buffers (1 or 2)
    int lin_form;        // token GEMM kernels: straight-line epilogue form (-1: the generic epilogue), conv3x3.hip lin_epilogue_fast
    long long* stamps;   // diagnostics only (TRAINNER_RING_STAMPS = device address): s_memtime stamps of workgroup phases, conv_ring.hip
    int dbg;   // timing experiments only (TRAINNER_CONV_DBG): 1 = skip MFMAs, 2 = skip global loads, 4 = skip epilogue stores, 8 = row-at-a-time epilogue
};

namespace tr {
// conv_ring.hip: persistent LDS-DMA ring kernel for the 16-bit 3x3 convolutions of the RRDB trunk / upsampling tail.
// Returns -1 when the launch is not eligible (the caller then takes the register-staged kernel), 0 on success, an error code otherwise.
int conv3x3_ring(hipStream_t st, ConvArgs& a, const tr_conv3x3_desc* d);
}
