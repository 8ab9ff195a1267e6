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

// Role table of the ring kernel: workgroup id = slot + nslots * role.  A role is one 64-input-channel block of x against one (32-wide kernel) or
// two (64-wide kernel) 32-channel output groups, which need not belong to the same conv; a single conv has npc x ncg roles, tr_wgrad3x3_multi
// pairs up the output groups of the convs of a dense block, which read channel slices of the SAME two buffers: the roles of a slot walk the
// same tiles on one XCD, so every x / gy byte is fetched from memory once per launch, not once per conv, and an x tile staged in LDS serves two
// output groups.
constexpr int TR_WG_MAX_ROLES = 16;
struct WgRole {
    int pc;                // 64-channel input block
    int in_c;              // input channels staged (x channels at and beyond it read as zero): the larger of the two halves' convs
    // per 32-channel output half (the 32-wide kernel uses half 0 only; a dead half has gy_c = 0 and nchunks = 0):
    int gy_coff[2], gy_c[2];   // channel offset (absolute: view offset included) / count of the half's output gradient inside the gy buffer
    int nchunks[2], ncg[2];    // partial layout of the half's conv in 32-wide groups: [slot][nchunks][ncg][32][9][32] (a 64-wide group = 2 of them)
    int cg[2];                 // the half's 32-wide group index in that layout
    float* partials[2];
    float* db_partials[2];     // [slot][ncg][32] or null
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
