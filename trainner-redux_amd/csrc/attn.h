// Window attention: argument block shared by the VALU (fp32-exact, token.hip) and MFMA (bf16, attn_mfma.hip) kernels.
#pragma once
#include "common.h"

namespace tr {

struct AttnArgs {
    tr_view qkv, out;          // forward: qkv ([q | k | v] x heads x hd), out = attention output (C channels)
    tr_view dout, dqkv;        // backward
    const float* table;        // (2*WS-1)^2 x nh
    float* dbias_ws;           // backward workspace (layout is the kernel family's own)
    int b, h, w, nh, hd, c, shift;
    float scale;
};

// bf16 views, even head_dim, rows wide enough for 16-byte operand pieces: the MFMA kernels apply
bool attn_mfma_ok(const AttnArgs& a, bool bwd);
int launch_window_attn_fwd_mfma(hipStream_t st, const AttnArgs& a);
int launch_window_attn_bwd_mfma(hipStream_t st, const AttnArgs& a, float* dtable, int accumulate);
size_t window_attn_bwd_mfma_workspace_floats(int ntasks, int nh);

}  // namespace tr
