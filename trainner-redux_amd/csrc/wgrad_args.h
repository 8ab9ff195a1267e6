// Kernel-side argument block of tr_wgrad3x3, shared by wgrad3x3.hip (register-staged kernels, reduction stage) and wgrad_ring.hip.
#pragma once
#include "common.h"

struct WgArgs {
    int n, h, w, hin, win;
    int ih, iw, shift;   // input extent in output coordinates and its origin shift (TR_IN_FULL: +1, TR_IN_VALID: -1)
    int cin, cout;
    int up, replicate;
    tr_view in, gy;
    float* partials;
    float* db_partials;   // [slot][ncg][CG] or null
    int ntx, nty, ntiles, nslots, nchunks, ncg;
};

namespace tr {
struct WgRingPlan {
    int ntx, nty, ntiles, nslots, slots_eff, nchunks, npc, ncg, cgw;
    size_t dw_floats, db_floats;
};
// wgrad_ring.hip: persistent LDS-DMA ring kernel for the 16-bit 3x3 weight gradients of the RRDB trunk / upsampling tail
WgRingPlan wgrad_ring_plan(int n, int h, int w, int cin, int cout);
// -1: not eligible (the caller takes wgrad_kernel); 0: launched, `p` describes the partial layout for wgrad_reduce_kernel; > 0: launch error
int wgrad3x3_ring(hipStream_t st, const tr_wgrad3x3_desc* d, WgArgs& a, WgRingPlan& p);
}
