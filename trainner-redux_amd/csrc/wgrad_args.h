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

// Role table of the ring kernel: workgroup id = slot + nslots * role.  A role is one (conv, 64-input-channel block, output group) of the launch;
// a single conv has npc x ncg roles, tr_wgrad3x3_multi the union over the convs of a dense block, which read channel slices of the SAME two
// buffers: the roles of a slot walk the same tiles on one XCD, so every x / gy byte is fetched from memory once per launch, not once per conv.
constexpr int TR_WG_MAX_ROLES = 16;
struct WgRole {
    int pc, cg;            // 64-channel input block, output group of this role's conv
    int in_c;              // input channels of the conv (x channels at and beyond it read as zero)
    int gy_coff, gy_c;     // channel offset / count of the conv's output gradient inside the gy buffer (gy_coff is absolute: view offset included)
    int nchunks, ncg;      // partial layout of the conv: [slot][nchunks][ncg][CG][9][32]
    int pad_;
    float* partials;
    float* db_partials;    // [slot][ncg][CG] or null
};
struct WgRoles {
    int nroles;
    int pad_;
    WgRole r[TR_WG_MAX_ROLES];
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
// several convs over channel slices of one x and one gy buffer in one launch (32-wide output groups); same return convention, `nslots_out`
// is the slot count of every conv's partial layout
int wgrad3x3_ring_multi(hipStream_t st, const tr_wgrad3x3_desc* descs, int ndesc, int* nslots_out);
}
