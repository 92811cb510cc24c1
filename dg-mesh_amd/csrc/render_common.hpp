// Device helpers shared by the blend kernels (render.hip, render_bwd4.hip).
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

// Minimum of q(d) = a dx^2 + 2 b dx dy + c dy^2, d = (x, y) - p, over the rectangle of points p in [X0, X1] x [Y0, Y1]
// (a convex quadric: zero if the centre is inside, otherwise attained on an edge; along an edge the unconstrained
// minimiser is clamped to the edge).  NaN inputs give NaN, which the callers keep.
__device__ __forceinline__ float rect_min_q(float x, float y, float a, float b, float c, float ra, float rc, float X0, float X1,
                                            float Y0, float Y1) {
    const bool inside = x >= X0 && x <= X1 && y >= Y0 && y <= Y1;
    float best;
    {
        const float dx = x - X0;
        const float py = fminf(fmaxf(y + b * dx * rc, Y0), Y1), dy = y - py;
        best = (a * dx + 2.f * b * dy) * dx + c * dy * dy;
    }
    {
        const float dx = x - X1;
        const float py = fminf(fmaxf(y + b * dx * rc, Y0), Y1), dy = y - py;
        best = fminf(best, (a * dx + 2.f * b * dy) * dx + c * dy * dy);
    }
    {
        const float dy = y - Y0;
        const float px = fminf(fmaxf(x + b * dy * ra, X0), X1), dx = x - px;
        best = fminf(best, (a * dx + 2.f * b * dy) * dx + c * dy * dy);
    }
    {
        const float dy = y - Y1;
        const float px = fminf(fmaxf(x + b * dy * ra, X0), X1), dx = x - px;
        best = fminf(best, (a * dx + 2.f * b * dy) * dx + c * dy * dy);
    }
    return inside ? 0.f : best;
}

// Which of the tile's four 8x8 quadrants can receive alpha >= 1/255 from this splat?
// alpha >= 1/255  <=>  q(d) = a dx^2 + 2 b dx dy + c dy^2 <= tau = 2 ln(255 o): the quadrant is kept unless the minimum of q
// over its block of pixel centres provably exceeds tau (inflated by 0.1 % + 0.01; "not provably outside", so that rounding or
// NaN can only keep a splat, never drop one).
__device__ __forceinline__ unsigned quadrant_mask(float x, float y, float a, float b, float c, float o, float tx0,
                                                  float ty0) {
    const float o255 = o * 255.0f;
    if (o255 < 1.0f) return 0u;  // alpha = min(.99, o*G) <= o < 1/255 for every pixel (G <= 1 where power <= 0)
    const float tau = 2.0f * __logf(o255) * 1.001f + 0.01f;
    const float ra = __builtin_amdgcn_rcpf(a), rc = __builtin_amdgcn_rcpf(c);
    unsigned m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float qx0 = tx0 + (float)((q & 1) * 8), qy0 = ty0 + (float)((q >> 1) * 8);
        if (!(rect_min_q(x, y, a, b, c, ra, rc, qx0, qx0 + 7.0f, qy0, qy0 + 7.0f) > tau)) m |= 1u << q;
    }
    return m;
}

// The same test for ONE 8x8 block of pixel centres with its first pixel at (qx0, qy0): the culling unit of a wave that stages for itself
// (render_fwd_async_kernel).
__device__ __forceinline__ bool quadrant_hit(float x, float y, float a, float b, float c, float o, float qx0, float qy0) {
    const float o255 = o * 255.0f;
    if (o255 < 1.0f) return false;
    const float tau = 2.0f * __logf(o255) * 1.001f + 0.01f;
    const float ra = __builtin_amdgcn_rcpf(a), rc = __builtin_amdgcn_rcpf(c);
    return !(rect_min_q(x, y, a, b, c, ra, rc, qx0, qx0 + 7.0f, qy0, qy0 + 7.0f) > tau);
}

// offs[g] + k: the row of Gaussian g's instance on tile (tile_x, tile_y) in the per-Gaussian order -- k counts the tiles of its rectangle
// row by row, as binning.hip's count / scatter walk them.  rect = rec[g][9] (pack_rect), offs = rec[g][10]: the backward gets both with
// the third 16-byte load of the splat's record, so the (R x 4)-byte array rounds 2-4 carried through the tile sort is gone.
__device__ __forceinline__ unsigned instance_row(unsigned rect, unsigned offs, unsigned tile_x, unsigned tile_y) {
    unsigned xmin, ymin, w;
    unpack_rect(rect, xmin, ymin, w);
    return offs + (tile_y - ymin) * w + (tile_x - xmin);
}

// The same test for the two 16 x 8 half tiles (bit 0: rows 0..7, bit 1: rows 8..15): the backward's culling unit.
__device__ __forceinline__ unsigned half_mask(float x, float y, float a, float b, float c, float o, float tx0, float ty0) {
    const float o255 = o * 255.0f;
    if (o255 < 1.0f) return 0u;
    const float tau = 2.0f * __logf(o255) * 1.001f + 0.01f;
    const float ra = __builtin_amdgcn_rcpf(a), rc = __builtin_amdgcn_rcpf(c);
    unsigned m = 0;
    if (!(rect_min_q(x, y, a, b, c, ra, rc, tx0, tx0 + 15.0f, ty0, ty0 + 7.0f) > tau)) m |= 1u;
    if (!(rect_min_q(x, y, a, b, c, ra, rc, tx0, tx0 + 15.0f, ty0 + 8.0f, ty0 + 15.0f) > tau)) m |= 2u;
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

// The same sums with the butterfly's stages in the order that needs NO selects: the stages that still carry several values run
// along lane bits 2 and 3 (row_shl / row_shr by 4 and 8, where DPP's bank mask picks the half of the lanes that keeps each
// value of a pair) and along bits 4 and 5 (v_permlane16_swap / v_permlane32_swap exchange the halves of a PAIR of registers,
// so one swap + one add reduces two values), and only the last two stages -- one register left -- run inside the quads.
// 14 DPP adds + 2 swaps + 2 adds + 1 copy (wave_reduce8t: 14 selects + 10 DPP adds + 2 swaps + 2 adds + 2 copies).
// On return every lane l holds the wave total of value number (l >> 2) & 7.
__device__ __forceinline__ float wave_reduce8m(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                               float v7) {
    float k0, k1, k2, k3, m0, m1;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %6, %6 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %8, %8 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %9, %9 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %10, %10 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %11, %11 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %12, %12 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %3, %13, %13 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        // k0: values {0,1} by lane bit 2, k1: {2,3}, k2: {4,5}, k3: {6,7}
        "v_add_f32_dpp %4, %0, %0 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %5, %2, %2 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %5, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xc"
        : "=&v"(k0), "=&v"(k1), "=&v"(k2), "=&v"(k3), "=&v"(m0), "=&v"(m1)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    // m0: values {0..3} by lane bits (2, 3), m1: {4..7}; every lane holds the sum over its row of 16 lanes' matching quarter
    auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(m0), __float_as_uint(m1), false, false);
    const float c = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);  // rows 0, 2: m0's sums over a row pair; rows 1, 3: m1's
    const unsigned cb = __float_as_uint(c);
    auto s32 = __builtin_amdgcn_permlane32_swap(cb, cb, false, false);
    float x = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "+v"(x));
    return x;
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

}  // namespace dgm
