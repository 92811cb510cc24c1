// Device helpers shared by the blend kernels (render.hip, render_bwd2.hip).
#pragma once
#include "dgm_common.hpp"

namespace dgm {

// Make a wave-uniform 64-bit value provably uniform (SGPR pair) for the scalar bit loops.  NB: the builtin
// returns a signed int -- widen through `unsigned`, or the low half is sign-extended into the high half.
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (unsigned long long)lo | ((unsigned long long)hi << 32);
}

// Which of the tile's four 8x8 quadrants can receive alpha >= 1/255 from this splat?
// alpha >= 1/255  <=>  q(d) = a dx^2 + 2 b dx dy + c dy^2 <= tau = 2 ln(255 o);  the ellipse {q <= tau} has the
// axis-aligned half extents sqrt(tau * Sxx), sqrt(tau * Syy) with S = conic^-1.  Inflated by 0.1 % + 0.01 px
// and written as "not provably outside" so that rounding or NaN can only keep a splat, never drop one.
__device__ __forceinline__ unsigned quadrant_mask(float x, float y, float a, float b, float c, float o, float tx0,
                                                  float ty0) {
    const float o255 = o * 255.0f;
    if (o255 < 1.0f) return 0u;  // alpha = min(.99, o*G) <= o < 1/255 for every pixel (G <= 1 where power <= 0)
    const float tau = 2.0f * __logf(o255) * 1.001f + 0.01f;
    const float det = a * c - b * b;
    const float inv = 1.0f / det;
    const float ex = sqrtf(tau * c * inv) * 1.001f + 0.01f;
    const float ey = sqrtf(tau * a * inv) * 1.001f + 0.01f;
    unsigned m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float qx0 = tx0 + (float)((q & 1) * 8), qy0 = ty0 + (float)((q >> 1) * 8);
        const bool outside = (x + ex < qx0) || (x - ex > qx0 + 7.0f) || (y + ey < qy0) || (y - ey > qy0 + 7.0f);
        if (!outside) m |= 1u << q;
    }
    return m;
}

// Transposed butterfly reduction of EIGHT values over the 64 lanes of a wave: on return every lane l holds the wave
// total of value number (l & 7).  Each xor-exchange stage keeps, per pair of values, the one selected by the lane's
// bit and ships the other to the partner lane, so the number of live values halves per stage:
//   xor 1, xor 2 : DPP quad_perm;  xor 4, xor 8 : DPP row_shl/row_shr with complementary bank masks;
//   xor 16, 32   : v_permlane16_swap / v_permlane32_swap (gfx950).
// 21 DPP/permlane adds + 14 selects instead of 8 x 6 full reductions.  Each asm block opens with the two wait
// states a DPP read needs after a VALU write of the same VGPR (hipcc does not pad inline asm).
__device__ __forceinline__ float wave_reduce8t(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                               float v7, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float t0 = b0 ? v0 : v1, k0 = b0 ? v1 : v0;
    float t1 = b0 ? v2 : v3, k1 = b0 ? v3 : v2;
    float t2 = b0 ? v4 : v5, k2 = b0 ? v5 : v4;
    float t3 = b0 ? v6 : v7, k3 = b0 ? v7 : v6;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %5, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %6, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %7, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3)
        : "v"(t0), "v"(t1), "v"(t2), "v"(t3));
    // k0: values {0,1} by bit0, k1: {2,3}, k2: {4,5}, k3: {6,7}
    float u0 = b1 ? k0 : k1, m0 = b1 ? k1 : k0;
    float u1 = b1 ? k2 : k3, m1 = b1 ? k3 : k2;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %2, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %3, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "+v"(m0), "+v"(m1)
        : "v"(u0), "v"(u1));
    // m0: values {0..3} by (l & 3), m1: {4..7}
    float w = b2 ? m0 : m1, n = b2 ? m1 : m0;
    float r8, r16;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %2, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %2, %3 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %1, %0, %0 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc"
        : "=&v"(r8), "=&v"(r16)
        : "v"(w), "v"(n));
    // r16: every lane holds the sum over its row of 16 lanes of value (l & 7)
    const unsigned x = __float_as_uint(r16);
    auto s16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    const float r32 = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
    const unsigned y = __float_as_uint(r32);
    auto s32 = __builtin_amdgcn_permlane32_swap(y, y, false, false);
    return __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
}

// In-place inclusive-scan style reduction of ONE value: the wave total ends up in lane 63.
__device__ __forceinline__ float wave_reduce1_lane63(float v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}

// Half extents (inflated, conservative) of the screen-space box outside of which alpha < 1/255 for this splat;
// returns false when the splat can never reach 1/255 (opacity too low).  See quadrant_mask for the derivation.
__device__ __forceinline__ bool alpha_extent(float a, float b, float c, float o, float& ex, float& ey) {
    const float o255 = o * 255.0f;
    if (o255 < 1.0f) return false;
    const float tau = 2.0f * __logf(o255) * 1.001f + 0.01f;
    const float inv = 1.0f / (a * c - b * b);
    ex = sqrtf(tau * c * inv) * 1.001f + 0.01f;
    ey = sqrtf(tau * a * inv) * 1.001f + 0.01f;
    return true;
}

}  // namespace dgm
