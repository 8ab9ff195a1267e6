// Shared helpers for libtrainner_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "trainner_hip.h"

namespace tr {

void set_error(const char* fmt, ...);

// Kernel-selection switches and the record of the last selection (tr_tune_set / tr_tune_get in trainner_hip.h).  The initial values come
// from the TRAINNER_* environment variables ONCE, when the library is loaded; launches never call getenv.
struct Tune {
    int conv_impl;        // tr_conv3x3, 16-bit 3x3: 0 = automatic, 1 = register-staged kernels only, 2 = the LDS-DMA ring kernel whenever eligible
    int ring_cfg;         // -1 = automatic; 33 / 32 / 31 = resident-weight pipelined instance with ring depth 3 / 2 / 1; 0 = streamed weights
    int ring_generic;     // 1 = generic epilogue body instead of the fast forms
    int wgrad_impl;       // tr_wgrad3x3: 0 = automatic, 1 = register-staged, 2 = ring whenever eligible
    int wgrad_multi;      // tr_wgrad3x3_multi: 1 = one fused launch per dense block when eligible, 0 = one launch per conv
    int wgrad_reduce;     // second stage of the weight gradients: 0 = 16-byte partial reads where cin % 4 == 0, 1 = the 4-byte form
    int dgrad_image_ref;  // tr_conv3x3_dgrad_image: 1 = per-pixel reference kernel
    int linear_epilogue;  // token GEMM epilogue: 0 = straight-line forms where they apply, 1 = always the generic epilogue
    int linear_impl;      // token GEMM (ksize 1 on a 32-wide token grid), 16-bit: 0 = LDS-DMA staged ring, 1 = register-staged one-chunk prefetch
    int last_conv_family; // which kernel family served the calling thread's last tr_conv3x3: 1 reg, 2 dma, 3 linear, 4 ring, 5 linear (LDS-DMA)
    int last_conv_instance;   // ring: NT*10000 + R*1000 + D*100 + WRES*10 + EPI ; others 0
    int ring_launches;    // number of tr_conv3x3 launches the ring kernel has served since the counter was last set
    long long stamps;     // development builds (-DTR_RING_DEV / -DTR_W43_STAMPS) only: device address for in-kernel time stamps
};
Tune& tune();

#define TR_REQUIRE(cond, code, ...)                 \
    do {                                            \
        if (!(cond)) {                              \
            ::tr::set_error(__VA_ARGS__);           \
            return (code);                          \
        }                                           \
    } while (0)

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

__host__ __device__ inline int esize(int dtype) { return dtype == TR_F32 ? 4 : 2; }

// ---- device-side scalar conversions -------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned)b) << 16);
}
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN preserving) on gfx950
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

__device__ __forceinline__ float f16_to_f32(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ unsigned short f32_to_f16(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }   // RNE, saturates to inf
// 16-bit storage (bf16 / fp16) <-> fp32
__device__ __forceinline__ float h_to_f32(unsigned short h, int dtype) { return dtype == TR_BF16 ? bf16_to_f32(h) : f16_to_f32(h); }
__device__ __forceinline__ unsigned short f32_to_h(float f, int dtype) { return dtype == TR_BF16 ? f32_to_bf16(f) : f32_to_f16(f); }

// nn.Mish: x * tanh(softplus(x)), softplus with torch's threshold 20 (the in-tree twins of MoSR's block: mosrv2_arch.py:101,
// moesr_arch.py:82).  With e = exp(x), n = e (e + 2): tanh(log(1 + e)) = n / (n + 2).
__device__ __forceinline__ float mish(float x) {
    if (x > 20.f) return x;
    const float e = __expf(x), n = e * (e + 2.f);
    return x * (n / (n + 2.f));
}
__device__ __forceinline__ float mish_grad(float x) {   // tanh(sp) + x * (1 - tanh(sp)^2) * sigmoid(x)
    if (x > 20.f) return 1.f;
    const float e = __expf(x), n = e * (e + 2.f), t = n / (n + 2.f);
    return t + x * (1.f - t * t) * (e / (1.f + e));
}

__device__ __forceinline__ float view_load(const tr_view& v, int64_t pix, int c) {
    int64_t idx = pix * v.cstride + v.coff + c;
    if (v.dtype != TR_F32) return h_to_f32(((const unsigned short*)v.ptr)[idx], v.dtype);
    return ((const float*)v.ptr)[idx];
}
__device__ __forceinline__ void view_store(const tr_view& v, int64_t pix, int c, float x) {
    int64_t idx = pix * v.cstride + v.coff + c;
    if (v.dtype != TR_F32) ((unsigned short*)v.ptr)[idx] = f32_to_h(x, v.dtype);
    else ((float*)v.ptr)[idx] = x;
}

// 4 consecutive channels; `vec` = the view allows 4-element vector access at multiples of 4
__device__ __forceinline__ float4 view_load4(const tr_view& v, int64_t pix, int c, bool vec) {
    int64_t idx = pix * v.cstride + v.coff + c;
    float4 r;
    if (v.dtype == TR_BF16) {
        const unsigned short* p = (const unsigned short*)v.ptr + idx;
        if (vec) {
            uint2 u = *(const uint2*)p;
            r.x = __uint_as_float(u.x << 16); r.y = __uint_as_float(u.x & 0xffff0000u);
            r.z = __uint_as_float(u.y << 16); r.w = __uint_as_float(u.y & 0xffff0000u);
        } else {
            r.x = bf16_to_f32(p[0]); r.y = bf16_to_f32(p[1]); r.z = bf16_to_f32(p[2]); r.w = bf16_to_f32(p[3]);
        }
    } else if (v.dtype == TR_F16) {
        const unsigned short* p = (const unsigned short*)v.ptr + idx;
        unsigned short q[4];
        if (vec) { const uint2 u = *(const uint2*)p; q[0] = u.x & 0xffffu; q[1] = u.x >> 16; q[2] = u.y & 0xffffu; q[3] = u.y >> 16; }
        else { q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = p[3]; }
        r.x = f16_to_f32(q[0]); r.y = f16_to_f32(q[1]); r.z = f16_to_f32(q[2]); r.w = f16_to_f32(q[3]);
    } else {
        const float* p = (const float*)v.ptr + idx;
        if (vec) r = *(const float4*)p;
        else { r.x = p[0]; r.y = p[1]; r.z = p[2]; r.w = p[3]; }
    }
    return r;
}
__device__ __forceinline__ void view_store4(const tr_view& v, int64_t pix, int c, float4 x, bool vec) {
    int64_t idx = pix * v.cstride + v.coff + c;
    if (v.dtype != TR_F32) {
        unsigned short* p = (unsigned short*)v.ptr + idx;
        unsigned short a = f32_to_h(x.x, v.dtype), b = f32_to_h(x.y, v.dtype), c2 = f32_to_h(x.z, v.dtype), d = f32_to_h(x.w, v.dtype);
        if (vec) { uint2 u; u.x = (unsigned)a | ((unsigned)b << 16); u.y = (unsigned)c2 | ((unsigned)d << 16); *(uint2*)p = u; }
        else { p[0] = a; p[1] = b; p[2] = c2; p[3] = d; }
    } else {
        float* p = (float*)v.ptr + idx;
        if (vec) *(float4*)p = x;
        else { p[0] = x.x; p[1] = x.y; p[2] = x.z; p[3] = x.w; }
    }
}

// host: can the view be accessed as aligned 4-channel vectors?
inline bool view_vec4_ok(const tr_view& v) {
    if (!v.ptr) return true;
    int es = esize(v.dtype);
    return (v.cstride % 4 == 0) && (v.coff % 4 == 0) && (((uintptr_t)v.ptr) % (4 * es) == 0);
}
// host: 16-byte chunk access (conv operand staging)
inline bool view_vec16B_ok(const tr_view& v) {
    int es = esize(v.dtype);
    int per = 16 / es;
    return (v.cstride % per == 0) && (v.coff % per == 0) && (((uintptr_t)v.ptr) % 16 == 0);
}

// packed conv operand geometry (must match pack.hip and conv3x3.hip)
struct PackGeom {
    int ck;       // channels per K-chunk: 32 (bf16) / 16 (f32) -> 64 bytes per row
    int eps;      // elements per 16-byte slot: 8 / 4
    int cgw;      // output channels per group: 32 or 64
    int ncg;      // number of output-channel groups
    int nchunks;  // number of K-chunks
};
__host__ __device__ inline PackGeom pack_geom(int cin, int cout, int compute) {
    PackGeom g;
    g.ck = (compute == TR_F32 || compute == TR_BF16X3) ? 16 : 32;     // 64-byte rows: 16 fp32, 16 x (hi + lo) bf16, or 32 16-bit channels
    g.eps = (compute == TR_F32 || compute == TR_BF16X3) ? 4 : 8;
    g.cgw = cout > 32 ? 64 : 32;
    g.ncg = (cout + g.cgw - 1) / g.cgw;
    g.nchunks = (cin + g.ck - 1) / g.ck;
    return g;
}

}  // namespace tr
