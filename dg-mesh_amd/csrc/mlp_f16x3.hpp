// fp32 GEMMs on the f16 matrix cores ("f16x3"): every fp32 operand is scaled by a power of two (exact) and split into
// two binary16 numbers
//     s x = h + l + e        h = rne16(s x),  l = rne16(s x - h),  |e| <= 2^-23 |s x|   (or 2^-25 absolute when l is subnormal)
// and a product is evaluated as the three partial products  ah*bh + ah*bl + al*bh  on v_mfma_f32_32x32x16_f16 (each
// product of two binary16 numbers is exact in fp32, accumulation in fp32).  What is dropped (al*bl) is below
// 2^-22 |a b|; measured against fp64 the split error is 4e-8 of sum |a||b| -- the same as the bf16x6 split and an order
// below the rounding error of an fp32 GEMM itself (tests/test_mlp.py::test_split_gemm_is_an_fp32_gemm).
//
// binary16 has 5 exponent bits, so operands must be brought into range first.  The scale only has to be constant along
// the contraction index:
//   * layer GEMM (contraction over features):  one scale per activation ROW (its max over K -> 2^14), one per weight
//     COLUMN (prepared with the planes); the epilogue multiplies the accumulator by both inverse powers of two;
//   * weight gradient (contraction over rows): one scale per COLUMN of X and of G, from column maxima that the
//     producing GEMM's epilogue accumulates (one atomic per column and workgroup) -- no extra pass over the data.
// With max |s x| in [2^14, 2^15) an element 2^-k of the maximum keeps 22 significant bits down to k = 11 and an absolute
// error of 2^-39 of the maximum below that: better than fp32 wherever it matters for the sum.
//
// Why: three MFMAs per 16-deep K step instead of six (bf16x6), and two planes instead of three -- the stationary
// weights of a wave shrink from 192 to 128 registers, which frees the registers the bf16x6 kernel lacked for deeper
// prefetch.  With 48 MFMAs per 32-row tile and wave the 256-wide layers become HBM-bound (A in, C out, fp32).
#pragma once
#include "dgm_common.hpp"

namespace dgm {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef DGM_F32X16_DEFINED
#define DGM_F32X16_DEFINED
typedef float f32x16 __attribute__((ext_vector_type(16)));
#endif

__device__ __forceinline__ f16x8 as_f16x8(const uint4 v) { return __builtin_bit_cast(f16x8, v); }

// two already-scaled floats -> (h, l) dwords, first element in the low half (v_cvt_pk_f16_f32: round to nearest even)
__device__ __forceinline__ void split2h(float a0, float a1, unsigned& h, unsigned& l) {
    const f16x2 hh = __builtin_convertvector((f32x2){a0, a1}, f16x2);
    const float r0 = a0 - (float)hh.x, r1 = a1 - (float)hh.y;
    const f16x2 ll = __builtin_convertvector((f32x2){r0, r1}, f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}

// scale = 2^(14 - floor(log2 max)) from the bits of a non-negative float maximum; the exponent is clamped so that both
// the scale and its inverse are normal numbers (an all-zero row / column gets a harmless finite scale)
__device__ __forceinline__ void scale_from_max_bits(unsigned bits, float& scale, float& inv) {
    int e = (int)((bits >> 23) & 0xffu);
    e = e < 20 ? 20 : (e > 250 ? 250 : e);
    scale = __uint_as_float((unsigned)(268 - e) << 23);  // 2^(127 + 14 - e - 127 + 127 ...): exponent field 268 - e
    inv = __uint_as_float((unsigned)(e - 14) << 23);     // its reciprocal
}

// wave-wide maximum of a non-negative float; the result is valid in lane 63 (DPP row_shr 1,2,4,8 + row_bcast 15,31)
__device__ __forceinline__ float wave_max_nonneg_lane63(float v) {
#define DGM_MAXDPP(ctrl_, rmask_)                                                                                      \
    v = fmaxf(v, __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), ctrl_, rmask_, 0xf, false)))
    DGM_MAXDPP(0x111, 0xf);
    DGM_MAXDPP(0x112, 0xf);
    DGM_MAXDPP(0x114, 0xf);
    DGM_MAXDPP(0x118, 0xf);
    DGM_MAXDPP(0x142, 0xa);
    DGM_MAXDPP(0x143, 0xc);
#undef DGM_MAXDPP
    return v;
}

// ---- weight planes ------------------------------------------------------------------------------------------------
// Bp[((stage*2 + plane)*2 + g)*ncols + col] holds the 8 halves B[k = stage*16 + g*8 + e][col] * scale[col], e = 0..7;
// inv_scale[col] = 1 / scale[col].  Source mapping as in mlp_bf16x6.hpp (mode 0: forward, B = W^T through the trunk's K
// mapping; mode 1: backward data, B = W[:, hoff:hoff+ncols]).
struct Prep3Job {
    int mode, Kp, ncols, in_features, emb_dim, hoff, k_valid, col_valid;
    const float* W;
    uint4* Bp;
    float* inv_scale;
};
static constexpr int PREP3_MAX_JOBS = 16;
struct Prep3Batch {
    Prep3Job job[PREP3_MAX_JOBS];
};

__device__ __forceinline__ float prep3_src(const Prep3Job& j, int k, int col) {
    if (j.mode == 0) {
        int src = k + j.hoff;  // (hoff: first input feature of a K = 256 slice -- the skip layer's trunk half)
        if (j.Kp == 96) src = k < j.emb_dim ? k : -1;
        else if (j.Kp == 352) src = k < 96 ? (k < j.emb_dim ? k : -1) : k - 96 + j.emb_dim;
        return (src >= 0 && col < j.col_valid) ? j.W[(size_t)col * j.in_features + src] : 0.f;
    }
    return k < j.k_valid ? j.W[(size_t)k * j.in_features + j.hoff + col] : 0.f;
}

// Every trunk matrix of a network in ONE launch (f16x3 forward pass), together with the two small jobs that used to precede
// it: grid (8, n3 + 1).
//   blockIdx.y < n3 : job y, columns [32 x, 32 x + 32): thread = (column, one of eight k-group slots); a thread keeps its
//                     <= 6 k-groups of 8 values in registers, the eight slot maxima of a column meet in LDS (no atomics, no
//                     zero-filled maxima needed), then the column's power-of-two scale, the split, the two planes and the inverse scale;
//   blockIdx.y == n3: x < 4: the heads' bf16x6 planes (mlp_bf16x6.hpp: prep6_one, 32 x 32 threads' worth);
//                     x >= 4: clear `zero_words` words at `zero` (the running column maxima of this forward / backward pass).
__global__ void __launch_bounds__(256)
mlp_prep3_all_kernel(const Prep3Batch b, int n3, const Prep6Job heads, unsigned* __restrict__ zero, int zero_words) {
    __shared__ float smax[8][32];
    const int tid = threadIdx.x;
    if ((int)blockIdx.y == n3) {
        if (blockIdx.x < 4) {
            for (int idx = blockIdx.x * 256 + tid; idx < (heads.Kp >> 3) * heads.ncols; idx += 1024) prep6_one(heads, idx);
        } else {
            for (int i = (blockIdx.x - 4) * 256 + tid; i < zero_words; i += 1024) zero[i] = 0u;
        }
        return;
    }
    const Prep3Job& j = b.job[blockIdx.y];
    const int col = blockIdx.x * 32 + (tid & 31), slot = tid >> 5, nkg = j.Kp >> 3;
    float e[6][8];
    float mx = 0.f;
#pragma unroll
    for (int it = 0; it < 6; it++) {
        const int kg = slot + 8 * it;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            e[it][i] = kg < nkg ? prep3_src(j, kg * 8 + i, col) : 0.f;
            mx = fmaxf(mx, fabsf(e[it][i]));
        }
    }
    smax[slot][tid & 31] = mx;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 8; g++) mx = fmaxf(mx, smax[g][tid & 31]);
    float sc, inv;
    scale_from_max_bits(__float_as_uint(mx), sc, inv);
    if (slot == 0) j.inv_scale[col] = inv;
#pragma unroll
    for (int it = 0; it < 6; it++) {
        const int kg = slot + 8 * it;
        if (kg < nkg) {
            uint4 H, L;
            split2h(e[it][0] * sc, e[it][1] * sc, H.x, L.x);
            split2h(e[it][2] * sc, e[it][3] * sc, H.y, L.y);
            split2h(e[it][4] * sc, e[it][5] * sc, H.z, L.z);
            split2h(e[it][6] * sc, e[it][7] * sc, H.w, L.w);
            uint4* dst = j.Bp + ((size_t)(kg >> 1) * 4 + (kg & 1)) * j.ncols + col;
            dst[0] = H;
            dst[2 * j.ncols] = L;
        }
    }
}

// ---- the trunk-layer GEMM, weights stationary in registers -------------------------------------------------------------
// C[M x 256] = [A1 | A2] * B with K = KS * 16 (A1: first K1 columns, A2 the rest).
//   EPI 0: C = relu(acc + bias), ReLU mask bits saved (mask[row][col / 32] bit col % 32)
//   EPI 1: C = acc where the saved mask bit is set, else 0 (backward data)
// Persistent grid of one 8-wave workgroup per CU; wave w owns output columns [32 w, 32 w + 32) and keeps both planes of
// its B slice in registers for the whole kernel (KS * 8 VGPRs).  Activations stream in 32-row tiles:
//   * producer role: wave w fetches rows 4w .. 4w+3 of a tile, ONE ROW PER INSTRUCTION (64 lanes x 16 B = 1 KiB
//     coalesced), two tiles ahead; the row maximum is a wave reduction (DPP) -> per-row power-of-two scale in SGPRs;
//     the scaled row is split into the two binary16 planes and written to LDS as [row][plane][k] (row pitch
//     2 * 2K + 16 bytes): a wave's 8-byte stores of one row are contiguous, and a fragment read (lane = row, 16 B of 8
//     consecutive k) walks the rows at an odd multiple of 16 B -- both conflict-free;
//   * consumer role: per K step two ds_read_b128 (h, l fragments of the lane's row) feed three MFMAs;
//   * epilogue: accumulator * (1/row scale) * (1/column scale), bias / ReLU / mask, stores, running column maxima of the
//     output (for the weight-gradient kernel's scales): one atomicMax per column and workgroup at the very end.
// Waves w and w + 4 share a SIMD and run the two halves of a tile step in opposite order (one does its VALU work --
// split, loads, stores -- while the other multiplies), one barrier per tile.
// wave-wide maxima of FOUR non-negative floats at once (valid in lane 63): the four DPP chains are interleaved, so each
// stage's instructions fill the two wait states a DPP read needs after a VALU write of the same register.  Two halves
// (stages 1-3, 4-6) so that a software-pipelined caller can place them in different MFMA shadows.
#define DGM_ST(ctrl_)                                                       \
    "v_max_f32_dpp %0, %0, %0 " ctrl_ "\n\t"                               \
    "v_max_f32_dpp %1, %1, %1 " ctrl_ "\n\t"                               \
    "v_max_f32_dpp %2, %2, %2 " ctrl_ "\n\t"                               \
    "v_max_f32_dpp %3, %3, %3 " ctrl_ "\n\t"
__device__ __forceinline__ void wave_max4_stage_a(float& a, float& b, float& c, float& d) {
    asm volatile("s_nop 1\n\t" DGM_ST("row_shr:1 row_mask:0xf bank_mask:0xf") DGM_ST("row_shr:2 row_mask:0xf bank_mask:0xf")
                     DGM_ST("row_shr:4 row_mask:0xf bank_mask:0xf") "s_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void wave_max4_stage_b(float& a, float& b, float& c, float& d) {
    asm volatile("s_nop 1\n\t" DGM_ST("row_shr:8 row_mask:0xf bank_mask:0xf") DGM_ST("row_bcast:15 row_mask:0xa bank_mask:0xf")
                     DGM_ST("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
#undef DGM_ST
__device__ __forceinline__ void wave_max4_nonneg_lane63(float& a, float& b, float& c, float& d) {
    wave_max4_stage_a(a, b, c, d);
    wave_max4_stage_b(a, b, c, d);
}

// first half of the epilogue, right after the MFMA phase (the row scales of the tile are still in LDS):
// accumulator * (1 / row scale) * (1 / column scale)
__device__ __forceinline__ void gemm3r_unscale(f32x16& acc, const float* rinv_t, int g, float binv) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 iv = *reinterpret_cast<const float4*>(rinv_t + q * 8 + 4 * g);
        acc[q * 4 + 0] *= iv.x * binv, acc[q * 4 + 1] *= iv.y * binv, acc[q * 4 + 2] *= iv.z * binv, acc[q * 4 + 3] *= iv.w * binv;
    }
}

// second half: bias / ReLU / mask bits (EPI 0) or mask application (EPI 1), stores, running column maximum.
// lane: column li of the wave's 32, rows (r & 3) + 8 (r >> 2) + 4 g
template <int EPI, bool FULL>
__device__ __forceinline__ void gemm3r_store(const f32x16& acc, const unsigned* mlds_w, int row0, int M, float* __restrict__ cb,
                                             unsigned* __restrict__ mb, int g, int li, float bv, float& cmax) {
    unsigned mwsel = 0u;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint4 mq = make_uint4(0u, 0u, 0u, 0u);
        if (EPI == 1) mq = *reinterpret_cast<const uint4*>(mlds_w + q * 8 + 4 * g);
        const unsigned mws[4] = {mq.x, mq.y, mq.z, mq.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int r = q * 4 + e, ro = e + 8 * q;
            float v = acc[r];
            if (EPI == 0) {
                v = fmaxf(v + bv, 0.f);
                const unsigned long long bal = __ballot(v > 0.f);  // low half: rows of g = 0, high half: g = 1
                const unsigned mw = g ? (unsigned)(bal >> 32) : (unsigned)bal;
                mwsel = (li == r) ? mw : mwsel;
            } else {
                v = ((mws[e] >> li) & 1u) ? v : 0.f;
            }
            if (FULL || row0 + ro < M) {
                cb[ro * 256] = v;
                cmax = fmaxf(cmax, fabsf(v));
            }
        }
    }
    if (EPI == 0 && li < 16) {  // lane li keeps the mask word of accumulator register li: one store for all sixteen rows
        const int ro = (li & 3) + 8 * (li >> 2);
        if (FULL || row0 + ro < M) mb[ro * 8] = mwsel;
    }
}

// DUAL: a second weight matrix over the same activations -- C2 = A * B2 + bias2, no ReLU (layer 0 and the embedding half of
// the skip layer both consume the embedding: one read of it, one set of row scales and planes, two outputs; C2 is the C_in
// of mlp_gemm3p_kernel<2>).  One accumulator chain per output in that case.
template <int EPI, int KS, int PF, bool DUAL = false>
__global__ void __launch_bounds__(512)
mlp_gemm3r_kernel(int M, int ntiles, const float* __restrict__ A1, int lda1, int K1, const float* __restrict__ A2, int lda2,
                  const uint4* __restrict__ Bp, const float* __restrict__ b_inv_scale, const float* __restrict__ bias,
                  unsigned* __restrict__ mask, float* __restrict__ C, unsigned* __restrict__ colmax,
                  const uint4* __restrict__ Bp2 = nullptr, const float* __restrict__ b_inv_scale2 = nullptr,
                  const float* __restrict__ bias2 = nullptr, float* __restrict__ C2 = nullptr) {
    constexpr int K = KS * 16;
    constexpr int NI = (K + 255) / 256;        // row-load instructions per row
    constexpr int RS = 4 * K + 16;             // bytes per LDS row: two planes of K halves + pad (odd multiple of 16 mod 256)
    constexpr int PLANE = 2 * K;               // bytes per plane
    static_assert((RS % 256) % 32 == 16, "row pitch must be an odd multiple of 16 bytes modulo 256");
    static_assert(PF == 1 || PF == 2, "tiles in flight in registers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ps = smem;                                          // [2][32][RS]
    float* rinv = reinterpret_cast<float*>(smem + 2 * 32 * RS);        // [2][32] inverse row scales
    unsigned* mlds = reinterpret_cast<unsigned*>(rinv + 64);           // [8 waves][32] mask words of the current tile (EPI 1)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    const int col = wv * 32 + li;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;

    f16x8 bh[KS], bl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        const uint4* b = Bp + ((size_t)ks * 4 + g) * 256 + col;
        bh[ks] = as_f16x8(b[0]), bl[ks] = as_f16x8(b[512]);
    }
    const float binv = b_inv_scale[col];
    const float bv = (EPI == 0) ? bias[col] : 0.f;
    float cmax = 0.f;
    f16x8 bh2[DUAL ? KS : 1], bl2[DUAL ? KS : 1];
    float binv2 = 0.f, bv2 = 0.f;
    if (DUAL) {
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const uint4* b = Bp2 + ((size_t)ks * 4 + g) * 256 + col;
            bh2[DUAL ? ks : 0] = as_f16x8(b[0]), bl2[DUAL ? ks : 0] = as_f16x8(b[512]);
        }
        binv2 = b_inv_scale2[col], bv2 = bias2[col];
    }

    float4 R[PF][4][NI];
#define R3_LOAD(slot_, tile_)                                                                                          \
    {                                                                                                                  \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++) {                                                             \
            int grow_ = (tile_) * 32 + wv * 4 + r_;                                                                    \
            grow_ = grow_ < M ? grow_ : M - 1;                                                                         \
            _Pragma("unroll") for (int i_ = 0; i_ < NI; i_++) {                                                        \
                const int k_ = i_ * 256 + lane * 4;                                                                    \
                R[slot_][r_][i_] = make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
                if (K % 256 == 0 || k_ < K) {                                                                          \
                    const float* s_ = (k_ < K1) ? (A1 + (size_t)grow_ * lda1 + k_) : (A2 + (size_t)grow_ * lda2 + (k_ - K1)); \
                    R[slot_][r_][i_] = *reinterpret_cast<const float4*>(s_);                                           \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
    }
#define R3_SPLIT(slot_, pb_)                                                                                           \
    {                                                                                                                  \
        float m_[4];                                                                                                   \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++) {                                                             \
            m_[r_] = 0.f;                                                                                              \
            _Pragma("unroll") for (int i_ = 0; i_ < NI; i_++) {                                                        \
                const float4 v_ = R[slot_][r_][i_];                                                                    \
                m_[r_] = fmaxf(fmaxf(m_[r_], fmaxf(fabsf(v_.x), fabsf(v_.y))), fmaxf(fabsf(v_.z), fabsf(v_.w)));       \
            }                                                                                                          \
        }                                                                                                              \
        wave_max4_nonneg_lane63(m_[0], m_[1], m_[2], m_[3]);                                                           \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++) {                                                             \
            const unsigned mb_ = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(m_[r_]), 63);                \
            float sc_, inv_;                                                                                           \
            scale_from_max_bits(mb_, sc_, inv_);                                                                       \
            const int row_ = wv * 4 + r_;                                                                              \
            rinv[(pb_) * 32 + row_] = inv_; /* every lane writes the same word: no exec juggling */                    \
            unsigned char* d_ = Ps + ((pb_) * 32 + row_) * RS;                                                         \
            _Pragma("unroll") for (int i_ = 0; i_ < NI; i_++) {                                                        \
                const int k_ = i_ * 256 + lane * 4;                                                                    \
                if (K % 256 == 0 || k_ < K) {                                                                          \
                    const float4 v_ = R[slot_][r_][i_];                                                                \
                    unsigned h0_, l0_, h1_, l1_;                                                                       \
                    split2h(v_.x * sc_, v_.y * sc_, h0_, l0_);                                                         \
                    split2h(v_.z * sc_, v_.w * sc_, h1_, l1_);                                                         \
                    *reinterpret_cast<uint2*>(d_ + 2 * k_) = make_uint2(h0_, h1_);                                     \
                    *reinterpret_cast<uint2*>(d_ + PLANE + 2 * k_) = make_uint2(l0_, l1_);                             \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
    }
    // fragments are fetched two K steps ahead of their MFMAs (ds_read latency ~ 100+ cycles, one step = 96 cycles of MFMA)
#define R3_MFMA(pb_)                                                                                                   \
    {                                                                                                                  \
        const unsigned char* ps_ = Ps + ((pb_) * 32 + li) * RS + g * 16;                                               \
        f16x8 fh_[3], fl_[3];                                                                                          \
        fh_[0] = as_f16x8(*reinterpret_cast<const uint4*>(ps_));                                                       \
        fl_[0] = as_f16x8(*reinterpret_cast<const uint4*>(ps_ + PLANE));                                               \
        constexpr int PD_ = DUAL ? 1 : 2; /* DUAL: a step is six MFMAs (192 cycles): one step ahead covers the LDS latency */ \
        if (PD_ == 2) {                                                                                                \
            fh_[1] = as_f16x8(*reinterpret_cast<const uint4*>(ps_ + (KS > 1 ? 32 : 0)));                               \
            fl_[1] = as_f16x8(*reinterpret_cast<const uint4*>(ps_ + PLANE + (KS > 1 ? 32 : 0)));                       \
        }                                                                                                              \
        _Pragma("unroll") for (int ks = 0; ks < KS; ks++) {                                                            \
            if (ks + PD_ < KS) {                                                                                       \
                fh_[(ks + PD_) % 3] = as_f16x8(*reinterpret_cast<const uint4*>(ps_ + (ks + PD_) * 32));                \
                fl_[(ks + PD_) % 3] = as_f16x8(*reinterpret_cast<const uint4*>(ps_ + PLANE + (ks + PD_) * 32));        \
            }                                                                                                          \
            /* two accumulator chains (cross terms | leading term): a dependent MFMA waits for its predecessor's */    \
            /* result, alternating chains keeps the matrix pipe issuing every 32 cycles                           */    \
            if (DUAL) { /* acc: first output, acc2: second output, alternated */                                      \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh_[ks % 3], bl[ks], ks == 0 ? zero16 : acc, 0, 0, 0);    \
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh_[ks % 3], bl2[DUAL ? ks : 0], ks == 0 ? zero16 : acc2, 0, 0, 0); \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl_[ks % 3], bh[ks], acc, 0, 0, 0);                       \
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl_[ks % 3], bh2[DUAL ? ks : 0], acc2, 0, 0, 0);         \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh_[ks % 3], bh[ks], acc, 0, 0, 0);                       \
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh_[ks % 3], bh2[DUAL ? ks : 0], acc2, 0, 0, 0);         \
            } else if (ks == 0) {                                                                                      \
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh_[0], bl[0], zero16, 0, 0, 0);                         \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh_[0], bh[0], zero16, 0, 0, 0);                          \
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl_[0], bh[0], acc2, 0, 0, 0);                           \
            } else {                                                                                                   \
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh_[ks % 3], bl[ks], acc2, 0, 0, 0);                     \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh_[ks % 3], bh[ks], acc, 0, 0, 0);                       \
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl_[ks % 3], bh[ks], acc2, 0, 0, 0);                     \
            }                                                                                                          \
        }                                                                                                              \
    }
#define R3_UNSCALE(pb_)                                                                                                \
    if (DUAL) gemm3r_unscale(acc2, rinv + (pb_) * 32, g, binv2);                                                       \
    else acc += acc2;                                                                                                  \
    gemm3r_unscale(acc, rinv + (pb_) * 32, g, binv);
#define R3_MWORD() if (EPI == 1) mlds[wv * 32 + li] = mword; /* both halves write the same 32 words; read back by this wave only */
#define R3_STORE(tile_)                                                                                                \
    {                                                                                                                  \
        const int row0_ = (tile_) * 32 + 4 * g;                                                                        \
        float* cb_ = C + (size_t)row0_ * 256 + col;                                                                    \
        unsigned* mb_ = mask + (size_t)row0_ * 8 + wv;                                                                 \
        if ((tile_) * 32 + 32 <= M) gemm3r_store<EPI, true>(acc, mlds + wv * 32, row0_, M, cb_, mb_, g, li, bv, cmax);   \
        else gemm3r_store<EPI, false>(acc, mlds + wv * 32, row0_, M, cb_, mb_, g, li, bv, cmax);                       \
        if (DUAL) { /* second output: linear, bias only */                                                             \
            float* c2_ = C2 + (size_t)row0_ * 256 + col;                                                               \
            if ((tile_) * 32 + 32 <= M) {                                                                              \
                _Pragma("unroll") for (int r_ = 0; r_ < 16; r_++) c2_[((r_ & 3) + 8 * (r_ >> 2)) * 256] = acc2[r_] + bv2; \
            } else {                                                                                                   \
                _Pragma("unroll") for (int r_ = 0; r_ < 16; r_++) {                                                    \
                    const int ro_ = (r_ & 3) + 8 * (r_ >> 2);                                                          \
                    if (row0_ + ro_ < M) c2_[ro_ * 256] = acc2[r_] + bv2;                                              \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
    }
    // EPI 1: the tile's 32 mask words of this wave's column group, one per lane, requested a phase before their use
#define R3_MASK(tile_)                                                                                                 \
    if (EPI == 1) {                                                                                                    \
        int mrow_ = (tile_) * 32 + li;                                                                                 \
        mrow_ = mrow_ < M ? mrow_ : M - 1;                                                                             \
        mword = mask[(size_t)mrow_ * 8 + wv];                                                                          \
    }

    f32x16 acc, acc2;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    unsigned mword = 0u;

    // prologue: tile 0 split into buffer 0, tiles 1 .. PF in flight in the register slots (slot of tile j = j % PF)
    if (my_tiles > 0) {
        R3_LOAD(0, blockIdx.x)
        R3_SPLIT(0, 0)
        if (PF == 1) {
            if (my_tiles > 1) R3_LOAD(0, blockIdx.x + G)
        } else {
            if (my_tiles > 1) R3_LOAD(PF - 1, blockIdx.x + G)
            if (my_tiles > 2) R3_LOAD(0, blockIdx.x + 2 * G)
        }
    }
    if (wv < 4 && my_tiles > 0) R3_MASK(blockIdx.x)
    __syncthreads();
    // One tile step.  Waves w and w + 4 share a SIMD and run the two halves of a step in opposite order, so that one
    // multiplies while the other does its VALU work:
    //   waves 0..3:  [ MFMA(j), unscale ]                     [ split(j+1), loads(j+1+PF), store(j) ]   barrier
    //   waves 4..7:  [ split(j+1), loads(..), store(j-1) ]    [ MFMA(j), unscale ]                      barrier
    // (the high waves keep tile j's scaled accumulators across the barrier and store them in the next step).
    // Memory-counter discipline (gfx9's vmcnt counts loads AND stores; with both kinds pending the compiler can only wait
    // with vmcnt(0)):  the ONE wait of a step sits at the top of split(), where everything outstanding -- the tile's row
    // loads, the mask words, the previous stores -- was issued a whole step (minus the split) earlier: the new loads go
    // out right after the wait, the stores after them, and nothing touches their results before the next step's split.
    // The barrier is an LDS-only barrier (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() would also drain vmcnt, i.e.
    // stall every wave on the loads it has just prefetched.
    // SLOT_ = register slot holding tile j + 1 (compile-time: two steps per loop trip).
#define R3_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define R3_STEP(j_, SLOT_)                                                                                             \
    {                                                                                                                  \
        const int tile = blockIdx.x + (j_) * G;                                                                        \
        const int pb = (j_) & 1;                                                                                       \
        if (wv < 4) {                                                                                                  \
            R3_MFMA(pb)                                                                                                \
            R3_UNSCALE(pb)                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            if ((j_) + 1 < my_tiles) R3_SPLIT(SLOT_, pb ^ 1)                                                           \
            R3_MWORD()                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            if ((j_) + 1 + PF < my_tiles) R3_LOAD(SLOT_, tile + (1 + PF) * G)                                          \
            if ((j_) + 1 < my_tiles) R3_MASK(tile + G)                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            R3_STORE(tile)                                                                                             \
        } else {                                                                                                       \
            if ((j_) + 1 < my_tiles) R3_SPLIT(SLOT_, pb ^ 1)                                                           \
            if ((j_) > 0) R3_MWORD()                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            if ((j_) + 1 + PF < my_tiles) R3_LOAD(SLOT_, tile + (1 + PF) * G)                                          \
            R3_MASK(tile)                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            if ((j_) > 0) R3_STORE(tile - G)                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            R3_MFMA(pb)                                                                                                \
            R3_UNSCALE(pb)                                                                                             \
        }                                                                                                              \
        R3_LDS_BARRIER();                                                                                              \
    }
    for (int j = 0; j < my_tiles; j += 2) {
        R3_STEP(j, PF - 1)          // tile j + 1 (odd) lives in slot 1 (PF == 2) / slot 0 (PF == 1)
        if (j + 1 < my_tiles) R3_STEP(j + 1, 0)
    }
    if (wv >= 4 && my_tiles > 0) {
        R3_MWORD()
        R3_STORE(blockIdx.x + (my_tiles - 1) * G)
    }
#undef R3_STEP
#undef R3_LDS_BARRIER
    if (colmax != nullptr) {
        // lanes li and li + 32 hold the same column: fold, then one atomic per column and workgroup
        const float o = __shfl_xor(cmax, 32, 64);
        cmax = fmaxf(cmax, o);
        if (g == 0) atomicMax(colmax + col, __float_as_uint(cmax));
    }
#undef R3_LOAD
#undef R3_SPLIT
#undef R3_MFMA
#undef R3_UNSCALE
#undef R3_STORE
#undef R3_MWORD
#undef R3_MASK
}

// ---- the K = 256 layer GEMM, software-pipelined ----------------------------------------------------------------------
// Same data flow as mlp_gemm3r_kernel (weights stationary, [row][plane][k] LDS tiles, per-row scales), but every wave runs
// ONE instruction stream in which the VALU work of the neighbouring tiles is sliced between the MFMAs of the current one:
// per 16-deep K step (3 MFMAs, 2 fragment reads) a wave also executes
//   steps 0..7 : the epilogue of the PREVIOUS tile, two accumulator registers per step (bias / ReLU / mask, store, column max);
//   step  8    : the one memory wait of the tile, the loads of tile j+2, the element maxima of tile j+1's rows;
//   steps 9..11: the cross-lane row maxima (two halves of the DPP chain), the scales;
//   steps 12..15: the split of tile j+1, one row per step, into the other LDS buffer.
// Measured on the two-role kernel (tools/g3_micro.hip): MFMA and VALU phases of DIFFERENT waves of a SIMD do not overlap in
// this instruction mix, while VALU instructions issued by the same wave right behind an MFMA run in its shadow (about five
// per MFMA).  The stores sit in the first half of a tile step and the wait in the middle, so the (unavoidable, gfx9 vmcnt
// counts loads and stores together) vmcnt(0) finds stores that are half a step old and loads that are a whole step old.
// Tail handling keeps the loop body branch-free: tile indices beyond the workgroup's last tile are clamped (the extra
// split lands in an LDS buffer nobody reads), the first step has its own copy without an epilogue, and the globally last --
// possibly partial -- tile is always some workgroup's final tile and is stored by the predicated path after the loop.
// CMAX_IN: also accumulate the column maxima of the INPUT rows (for a producer that cannot deliver them) into colmax_in.
// EPI 2 (the skip layer's 256-wide half): C = relu(acc + C_in), where C already holds the other half of the product plus the
// bias (written by the K = 96 kernel's second output); the tile's sixteen C_in values per lane are requested at the tile's
// memory slot (step 8) and consumed half a tile later by the epilogue, in place.
template <int EPI, bool CMAX_IN>
__global__ void __launch_bounds__(512)
mlp_gemm3p_kernel(int M, int ntiles, const float* __restrict__ A, int lda, const uint4* __restrict__ Bp,
                  const float* __restrict__ b_inv_scale, const float* __restrict__ bias, unsigned* __restrict__ mask,
                  float* __restrict__ C, unsigned* __restrict__ colmax, unsigned* __restrict__ colmax_in) {
    constexpr int KS = 16, K = 256, RS = 4 * K + 16, PLANE = 2 * K;
    constexpr int G3P_PD = 1;  // fragment prefetch distance in K steps (2 measured: no gain, 14 more registers)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ps = smem;                                          // [2][32][RS]
    float* rinv = reinterpret_cast<float*>(smem + 2 * 32 * RS);        // [2][32] inverse row scales
    unsigned* mlds = reinterpret_cast<unsigned*>(rinv + 64);           // [2][8 waves][32] mask words of tile t in mlds[t & 1] (EPI 1)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    const int col = wv * 32 + li;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    if (my_tiles <= 0) return;
    const int last_tile = blockIdx.x + (my_tiles - 1) * G;

    f16x8 bh[KS], bl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        const uint4* b = Bp + ((size_t)ks * 4 + g) * 256 + col;
        bh[ks] = as_f16x8(b[0]), bl[ks] = as_f16x8(b[512]);
    }
    const float binv = b_inv_scale[col];
    const float bv = (EPI == 0) ? bias[col] : 0.f;
    float cmax = 0.f;
    float4 cin = make_float4(0.f, 0.f, 0.f, 0.f);  // CMAX_IN: running maxima of input columns 4 lane .. 4 lane + 3
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc, out;
    float ev[EPI == 2 ? 16 : 1];  // EPI 2: C_in of the tile whose accumulators are in flight
    float4 R[2][4];     // rows of tile t live in R[t & 1]
    unsigned mw[2] = {0u, 0u};  // EPI 1: mask word (row li of the tile, this wave's column group) of tile t in mw[t & 1]

#define P3_TILE(j_) min((int)blockIdx.x + (j_) * G, last_tile)  /* int on both sides: a mixed unsigned / int min() resolves to the double overload */
#define P3_LOAD(slot_, tile_)                                                                                          \
    {                                                                                                                  \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++) {                                                             \
            int grow_ = (tile_) * 32 + wv * 4 + r_;                                                                    \
            grow_ = grow_ < M ? grow_ : M - 1;                                                                         \
            R[slot_][r_] = *reinterpret_cast<const float4*>(A + (size_t)grow_ * lda + lane * 4);                       \
        }                                                                                                              \
        if (EPI == 1) {                                                                                                \
            int mrow_ = (tile_) * 32 + li;                                                                             \
            mrow_ = mrow_ < M ? mrow_ : M - 1;                                                                         \
            mw[slot_] = mask[(size_t)mrow_ * 8 + wv];                                                                  \
        }                                                                                                              \
    }
#define P3_ROWMAX(slot_)                                                                                               \
    {                                                                                                                  \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++) {                                                             \
            const float4 v_ = R[slot_][r_];                                                                            \
            m_[r_] = fmaxf(fmaxf(fabsf(v_.x), fabsf(v_.y)), fmaxf(fabsf(v_.z), fabsf(v_.w)));                          \
            if (CMAX_IN) {                                                                                             \
                cin.x = fmaxf(cin.x, fabsf(v_.x)), cin.y = fmaxf(cin.y, fabsf(v_.y));                                  \
                cin.z = fmaxf(cin.z, fabsf(v_.z)), cin.w = fmaxf(cin.w, fabsf(v_.w));                                  \
            }                                                                                                          \
        }                                                                                                              \
    }
#define P3_SCALES(pb_)                                                                                                 \
    {                                                                                                                  \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++) {                                                             \
            const unsigned mb_ = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(m_[r_]), 63);                \
            float inv_;                                                                                                \
            scale_from_max_bits(mb_, sc_[r_], inv_);                                                                   \
            rinv[(pb_) * 32 + wv * 4 + r_] = inv_;                                                                     \
        }                                                                                                              \
    }
#define P3_SPLIT_ROW(slot_, pb_, r_)                                                                                   \
    {                                                                                                                  \
        const float4 v_ = R[slot_][r_];                                                                                \
        unsigned h0_, l0_, h1_, l1_;                                                                                   \
        split2h(v_.x * sc_[r_], v_.y * sc_[r_], h0_, l0_);                                                             \
        split2h(v_.z * sc_[r_], v_.w * sc_[r_], h1_, l1_);                                                             \
        unsigned char* d_ = Ps + ((pb_) * 32 + wv * 4 + (r_)) * RS + 8 * lane;                                         \
        *reinterpret_cast<uint2*>(d_) = make_uint2(h0_, h1_);                                                          \
        *reinterpret_cast<uint2*>(d_ + PLANE) = make_uint2(l0_, l1_);                                                  \
    }
    // epilogue of accumulator register r_ of the previous tile (full tile: no predicates)
#define P3_STORE_REG(r_)                                                                                               \
    {                                                                                                                  \
        const int ro_ = ((r_) & 3) + 8 * ((r_) >> 2);                                                                  \
        float v_ = out[r_];                                                                                            \
        if (EPI == 0 || EPI == 2) {                                                                                    \
            v_ = fmaxf(v_ + (EPI == 2 ? ev[EPI == 2 ? (r_) : 0] : bv), 0.f);                                           \
            const unsigned long long bal_ = __ballot(v_ > 0.f);                                                        \
            const unsigned mwd_ = g ? (unsigned)(bal_ >> 32) : (unsigned)bal_;                                         \
            mwsel = (li == (r_)) ? mwd_ : mwsel;                                                                       \
        } else {                                                                                                       \
            const unsigned mq_ = mlp_[ro_]; /* mask word of the row: the two halves read two broadcast addresses */   \
            v_ = ((mq_ >> li) & 1u) ? v_ : 0.f;                                                                        \
        }                                                                                                              \
        cb[ro_ * 256] = v_;                                                                                            \
        cmax = fmaxf(cmax, fabsf(v_));                                                                                 \
    }

    // one tile step.  FIRST: no previous tile to store.  SN = register slot of tile j+1 (compile-time).
#define P3_STEP(j_, SN_, FIRST_)                                                                                       \
    {                                                                                                                  \
        const int pb = (j_) & 1;                                                                                       \
        const unsigned char* ps_ = Ps + (pb * 32 + li) * RS + g * 16;                                                  \
        const int ptile_ = blockIdx.x + ((j_) - 1) * G;                                                                \
        float* cb = C + (size_t)(ptile_ * 32 + 4 * g) * 256 + col;                                                     \
        unsigned* mb = mask + (size_t)(ptile_ * 32 + 4 * g) * 8 + wv;                                                  \
        unsigned mwsel = 0u;                                                                                           \
        const unsigned* mlp_ = mlds + (1 - pb) * 256 + wv * 32 + 4 * g; /* previous tile's mask words */               \
        float m_[4], sc_[4];                                                                                           \
        /* fragments PD_ K steps ahead (the slices between the MFMAs cover the LDS latency)                               */ \
        constexpr int PD_ = G3P_PD;                                                                                    \
        f16x8 fh_[PD_ + 1], fl_[PD_ + 1];                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < PD_; i_++) {                                                           \
            fh_[i_] = as_f16x8(*reinterpret_cast<const uint4*>(ps_ + i_ * 32));                                        \
            fl_[i_] = as_f16x8(*reinterpret_cast<const uint4*>(ps_ + PLANE + i_ * 32));                                \
        }                                                                                                              \
        _Pragma("unroll") for (int ks = 0; ks < KS; ks++) {                                                            \
            if (ks + PD_ < KS) {                                                                                       \
                fh_[(ks + PD_) % (PD_ + 1)] = as_f16x8(*reinterpret_cast<const uint4*>(ps_ + (ks + PD_) * 32));        \
                fl_[(ks + PD_) % (PD_ + 1)] = as_f16x8(*reinterpret_cast<const uint4*>(ps_ + PLANE + (ks + PD_) * 32)); \
            }                                                                                                          \
            /* one accumulator chain: the VALU slices between the MFMAs cover the dependent-issue latency, and the */   \
            /* second chain's 16 registers are what keeps this kernel out of scratch                                */   \
            if (ks == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh_[0], bl[0], zero16, 0, 0, 0);                 \
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh_[ks % (PD_ + 1)], bl[ks], acc, 0, 0, 0);              \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl_[ks % (PD_ + 1)], bh[ks], acc, 0, 0, 0);                   \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh_[ks % (PD_ + 1)], bh[ks], acc, 0, 0, 0);                   \
            if (!(FIRST_) && ks < 8) {                                                                                 \
                P3_STORE_REG(2 * ks)                                                                                   \
                P3_STORE_REG(2 * ks + 1)                                                                               \
                if ((EPI == 0 || EPI == 2) && ks == 7 && li < 16) mb[((li & 3) + 8 * (li >> 2)) * 8] = mwsel;          \
            }                                                                                                          \
            if (EPI == 1 && ks == 0) mlds[pb * 256 + wv * 32 + li] = mw[1 - (SN_)]; /* this tile's mask words */        \
            if (ks == 8) {                                                                                             \
                P3_ROWMAX(SN_) /* first use of tile j+1's rows: the tile's one memory wait */                          \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
                if (EPI == 2) { /* C_in of THIS tile, ahead of the row loads: the epilogue's wait leaves those in flight */ \
                    const int t_ = blockIdx.x + (j_) * G;                                                              \
                    _Pragma("unroll") for (int r_ = 0; r_ < 16; r_++) {                                                \
                        int er_ = t_ * 32 + 4 * g + (r_ & 3) + 8 * (r_ >> 2);                                          \
                        er_ = er_ < M ? er_ : M - 1;                                                                   \
                        ev[EPI == 2 ? r_ : 0] = C[(size_t)er_ * 256 + col];                                            \
                    }                                                                                                  \
                }                                                                                                      \
                P3_LOAD(1 - (SN_), P3_TILE((j_) + 2))                                                                  \
            }                                                                                                          \
            if (ks == 9) wave_max4_stage_a(m_[0], m_[1], m_[2], m_[3]);                                                \
            if (ks == 10) wave_max4_stage_b(m_[0], m_[1], m_[2], m_[3]);                                               \
            if (ks == 11) P3_SCALES(pb ^ 1)                                                                            \
            if (ks >= 12) P3_SPLIT_ROW(SN_, pb ^ 1, ks - 12)                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
        }                                                                                                              \
        /* unscale into `out` (stored during the next step) */                                                         \
        _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                                \
            const float4 iv = *reinterpret_cast<const float4*>(rinv + pb * 32 + q * 8 + 4 * g);                        \
            out[q * 4 + 0] = acc[q * 4 + 0] * (iv.x * binv);                                       \
            out[q * 4 + 1] = acc[q * 4 + 1] * (iv.y * binv);                                       \
            out[q * 4 + 2] = acc[q * 4 + 2] * (iv.z * binv);                                       \
            out[q * 4 + 3] = acc[q * 4 + 3] * (iv.w * binv);                                       \
        }                                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                                \
    }

    // prologue: tile 0 split into buffer 0 (not overlapped), tile 1 in flight
    {
        float m_[4], sc_[4];
        P3_LOAD(0, P3_TILE(0))
        P3_ROWMAX(0)
        wave_max4_nonneg_lane63(m_[0], m_[1], m_[2], m_[3]);
        P3_SCALES(0)
#pragma unroll
        for (int r = 0; r < 4; r++) P3_SPLIT_ROW(0, 0, r)
        P3_LOAD(1, P3_TILE(1))
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    P3_STEP(0, 1, true)
    for (int j = 1; j < my_tiles; j += 2) {
        P3_STEP(j, 0, false)
        if (j + 1 < my_tiles) P3_STEP(j + 1, 1, false)
    }
    {   // the workgroup's final tile (the only one that can be partial): predicated store of `out`
        const int tile = last_tile;
        const int row0 = tile * 32 + 4 * g;
        float* cb = C + (size_t)row0 * 256 + col;
        unsigned* mb = mask + (size_t)row0 * 8 + wv;
        const unsigned* ml = mlds + ((my_tiles - 1) & 1) * 256 + wv * 32;
        if (EPI == 2) {
#pragma unroll
            for (int r = 0; r < 16; r++) out[r] += ev[EPI == 2 ? r : 0];  // (rows beyond M: clamped loads, never stored)
        }
        constexpr int SE = EPI == 2 ? 0 : EPI;  // bias already inside C_in: plain ReLU / mask epilogue with bias 0
        if (tile * 32 + 32 <= M) gemm3r_store<SE, true>(out, ml, row0, M, cb, mb, g, li, bv, cmax);
        else gemm3r_store<SE, false>(out, ml, row0, M, cb, mb, g, li, bv, cmax);
    }
    if (CMAX_IN && colmax_in != nullptr) {
        // the clamped tail re-reads the last tile (same values: harmless for a maximum); fold the eight waves through LDS
        unsigned* red = reinterpret_cast<unsigned*>(smem);  // the plane buffers are dead after the last barrier ...
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // ... once every wave is past its last fragment read
        if (tid < 256) red[tid] = 0u;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        atomicMax(red + 4 * lane + 0, __float_as_uint(cin.x));
        atomicMax(red + 4 * lane + 1, __float_as_uint(cin.y));
        atomicMax(red + 4 * lane + 2, __float_as_uint(cin.z));
        atomicMax(red + 4 * lane + 3, __float_as_uint(cin.w));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid < 256) atomicMax(colmax_in + tid, red[tid]);
    }
    if (colmax != nullptr) {
        const float o = __shfl_xor(cmax, 32, 64);
        cmax = fmaxf(cmax, o);
        if (g == 0) atomicMax(colmax + col, __float_as_uint(cmax));
    }
#undef P3_TILE
#undef P3_LOAD
#undef P3_ROWMAX
#undef P3_SCALES
#undef P3_SPLIT_ROW
#undef P3_STORE_REG
#undef P3_STEP
}

// ---- weight gradient of the K = 256 layers --------------------------------------------------------------------------
// partial[chunk][k][j] = sum_{rows of chunk} X[row][k] * G[row][j]   (then * 1 / (sx[k] sg[j])).
// Same decomposition as mlp_dw6b_kernel (one 8-wave workgroup per chunk of rows covering all 256 x 256 outputs, waves
// 4 x 2, wave tile 64 x 128, stages of 16 rows, every thread stages an 8-row x 2-column piece), with two binary16 planes:
// the staging thread multiplies its two columns by their power-of-two scales (from the column maxima xmax / gmax the
// producing kernels accumulated), splits, and writes one 16-byte granule (8 rows of one column) per plane.
static constexpr int DW3_U = 4 * 256;  // uint4 per operand stage: [plane][row half][column]

__global__ void __launch_bounds__(512)
mlp_dw3b_kernel(int M, int rows_per_chunk, const float* __restrict__ X, int ldx, const float* __restrict__ G,
                const unsigned* __restrict__ xmax, const unsigned* __restrict__ gmax, float* __restrict__ partial,
                float* __restrict__ partial_db) {
    __shared__ uint4 Xs[2][DW3_U];
    __shared__ uint4 Gs[2][DW3_U];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1, g = lane >> 5, li = lane & 31;
    const int chunk = blockIdx.x;
    const int r0 = chunk * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    const int nst = (r1 - r0 + 15) >> 4;
    const bool isG = tid >= 256;
    const int rg = (tid >> 7) & 1, c2 = tid & 127;
    const float* sp = isG ? (G + c2 * 2) : (X + c2 * 2);
    const int sld = isG ? 256 : ldx;
    uint4* sdst0 = (isG ? &Gs[0][0] : &Xs[0][0]) + rg * 256 + c2 * 2;
    float sc0, sc1, inv_unused;
    {
        const unsigned* mx = isG ? gmax : xmax;
        scale_from_max_bits(mx[c2 * 2], sc0, inv_unused);
        scale_from_max_bits(mx[c2 * 2 + 1], sc1, inv_unused);
    }
    float2 v[8];
    float2 colsum = make_float2(0.f, 0.f);

#define DW3_LOAD(st_)                                                                                 \
    {                                                                                                 \
        const int rb_ = r0 + (st_) * 16 + rg * 8;                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) { /* branch-free: rows past the chunk load its last row, then zero */ \
            const float2 t_ = *reinterpret_cast<const float2*>(sp + (size_t)min(rb_ + i_, r1 - 1) * sld); \
            v[i_] = rb_ + i_ < r1 ? t_ : make_float2(0.f, 0.f);                                       \
        }                                                                                             \
    }
#define DW3_STORE(buf_)                                                                               \
    {                                                                                                 \
        uint4* d_ = sdst0 + (buf_) * DW3_U;                                                           \
        uint4 H_, L_;                                                                                 \
        split2h(v[0].x * sc0, v[1].x * sc0, H_.x, L_.x);                                              \
        split2h(v[2].x * sc0, v[3].x * sc0, H_.y, L_.y);                                              \
        split2h(v[4].x * sc0, v[5].x * sc0, H_.z, L_.z);                                              \
        split2h(v[6].x * sc0, v[7].x * sc0, H_.w, L_.w);                                              \
        d_[0] = H_, d_[512] = L_;                                                                     \
        split2h(v[0].y * sc1, v[1].y * sc1, H_.x, L_.x);                                              \
        split2h(v[2].y * sc1, v[3].y * sc1, H_.y, L_.y);                                              \
        split2h(v[4].y * sc1, v[5].y * sc1, H_.z, L_.z);                                              \
        split2h(v[6].y * sc1, v[7].y * sc1, H_.w, L_.w);                                              \
        d_[1] = H_, d_[513] = L_;                                                                     \
        if (isG) {                                                                                    \
            _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) colsum.x += v[i_].x, colsum.y += v[i_].y; \
        }                                                                                             \
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;

    DW3_LOAD(0)
    DW3_STORE(0)
    if (nst > 1) DW3_LOAD(1)
    __syncthreads();
    for (int st = 0; st < nst; st++) {
        const int buf = st & 1;
        if (st + 1 < nst) {
            DW3_STORE(buf ^ 1)
            if (st + 2 < nst) DW3_LOAD(st + 2)
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint4* xs = Xs[buf];
        const uint4* gs = Gs[buf];
        f16x8 ah[2], al[2];
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const int ai = g * 256 + wm * 64 + mt * 32 + li;
            ah[mt] = as_f16x8(xs[ai]), al[mt] = as_f16x8(xs[512 + ai]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const int bi = g * 256 + wn * 128 + nt * 32 + li;
            const f16x8 gh = as_f16x8(gs[bi]), gl = as_f16x8(gs[512 + bi]);
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], gl, acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], gh, acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], gh, acc[mt][nt], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the rows just prefetched
    }
#undef DW3_LOAD
#undef DW3_STORE

    float* out = partial + (size_t)chunk * 256 * 256;
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
        const int col = wn * 128 + nt * 32 + li;
        float sg, ig;
        scale_from_max_bits(gmax[col], sg, ig);
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int k = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                float sx, ix;
                scale_from_max_bits(xmax[k], sx, ix);
                out[(size_t)k * 256 + col] = acc[mt][nt][r] * (ix * ig);
            }
        }
    }
    if (isG && partial_db != nullptr)
        *reinterpret_cast<float2*>(partial_db + ((size_t)chunk * 2 + rg) * 256 + c2 * 2) = colsum;
}

// ---- weight gradient of the layers that consume the embedding (K = 96: layer 0; K = 352 = 96 | 256: the skip layer) --------
// partial[chunk][k][j] = sum_{rows of chunk} [X1 | X2][row][k] * G[row][j]; same arithmetic and staging granules as
// mlp_dw3b_kernel.  One 8-wave workgroup per chunk covers all MT * 32 x 256 outputs: wave w owns the 32 gradient columns
// [32 w, 32 w + 32) against ALL MT * 32 input columns (one G fragment pair feeds 3 MT MFMAs; MT * 16 accumulator registers).
// Stagers: thread t < MT * 16 carries two input columns, threads 256 .. 383 two gradient columns, sixteen rows each per stage.
template <int MT>
__global__ void __launch_bounds__(512)
mlp_dw3e_kernel(int M, int rows_per_chunk, const float* __restrict__ X1, int ldx1, int K1, const float* __restrict__ X2, int ldx2,
                const float* __restrict__ G, const unsigned* __restrict__ xmax1, const unsigned* __restrict__ xmax2,
                const unsigned* __restrict__ gmax, float* __restrict__ partial, float* __restrict__ partial_db) {
    constexpr int NX = MT * 32;   // input columns
    constexpr int XU = 4 * NX;    // uint4 per X stage: [plane][row half][column]
    extern __shared__ __attribute__((aligned(16))) uint4 dw3e_lds[];
    uint4* Xs = dw3e_lds;                 // [2][XU]
    uint4* Gs = dw3e_lds + 2 * XU;        // [2][DW3_U]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    const int chunk = blockIdx.x;
    const int r0 = chunk * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    const int nst = (r1 - r0 + 15) >> 4;
    const bool isX = tid < NX / 2, isG = tid >= 256 && tid < 384;
    const bool stager = isX || isG;
    const int c2 = isG ? tid - 256 : tid;  // column pair
    const float* sp = G;
    int sld = 256;
    float sc0 = 0.f, sc1 = 0.f;
    uint4* sdst0 = Gs;
    if (stager) {
        float unused;
        const int col = 2 * c2;
        if (isG) {
            sp = G + col;
            scale_from_max_bits(gmax[col], sc0, unused);
            scale_from_max_bits(gmax[col + 1], sc1, unused);
            sdst0 = Gs + col;
        } else {
            const bool first = col < K1;
            sp = first ? X1 + col : X2 + (col - K1);
            sld = first ? ldx1 : ldx2;
            const unsigned* mx = first ? xmax1 + col : xmax2 + (col - K1);
            scale_from_max_bits(mx[0], sc0, unused);
            scale_from_max_bits(mx[1], sc1, unused);
            sdst0 = Xs + col;
        }
    }
    const int snc = isG ? 256 : NX;           // columns of the staged operand
    const int sbuf = isG ? DW3_U : XU;        // uint4 per stage buffer
    float2 v[16];
    float2 colsum = make_float2(0.f, 0.f);

#define DW3E_LOAD(st_)                                                                                \
    if (stager) {                                                                                     \
        const int rb_ = r0 + (st_) * 16;                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 16; i_++) { /* branch-free: rows past the chunk load its last row, then zero */ \
            const float2 t_ = *reinterpret_cast<const float2*>(sp + (size_t)min(rb_ + i_, r1 - 1) * sld); \
            v[i_] = rb_ + i_ < r1 ? t_ : make_float2(0.f, 0.f);                                       \
        }                                                                                             \
    }
#define DW3E_STORE(buf_)                                                                              \
    if (stager) {                                                                                     \
        uint4* d_ = sdst0 + (buf_) * sbuf;                                                            \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; h_++) {                                            \
            uint4 H_, L_;                                                                             \
            split2h(v[8 * h_ + 0].x * sc0, v[8 * h_ + 1].x * sc0, H_.x, L_.x);                        \
            split2h(v[8 * h_ + 2].x * sc0, v[8 * h_ + 3].x * sc0, H_.y, L_.y);                        \
            split2h(v[8 * h_ + 4].x * sc0, v[8 * h_ + 5].x * sc0, H_.z, L_.z);                        \
            split2h(v[8 * h_ + 6].x * sc0, v[8 * h_ + 7].x * sc0, H_.w, L_.w);                        \
            d_[h_ * snc] = H_, d_[2 * snc + h_ * snc] = L_;                                           \
            split2h(v[8 * h_ + 0].y * sc1, v[8 * h_ + 1].y * sc1, H_.x, L_.x);                        \
            split2h(v[8 * h_ + 2].y * sc1, v[8 * h_ + 3].y * sc1, H_.y, L_.y);                        \
            split2h(v[8 * h_ + 4].y * sc1, v[8 * h_ + 5].y * sc1, H_.z, L_.z);                        \
            split2h(v[8 * h_ + 6].y * sc1, v[8 * h_ + 7].y * sc1, H_.w, L_.w);                        \
            d_[h_ * snc + 1] = H_, d_[2 * snc + h_ * snc + 1] = L_;                                   \
        }                                                                                             \
        if (isG) {                                                                                    \
            _Pragma("unroll") for (int i_ = 0; i_ < 16; i_++) colsum.x += v[i_].x, colsum.y += v[i_].y; \
        }                                                                                             \
    }

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[mt][r] = 0.f;

    DW3E_LOAD(0)
    DW3E_STORE(0)
    if (nst > 1) DW3E_LOAD(1)
    __syncthreads();
    for (int st = 0; st < nst; st++) {
        const int buf = st & 1;
        if (st + 1 < nst) {
            DW3E_STORE(buf ^ 1)
            if (st + 2 < nst) DW3E_LOAD(st + 2)
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint4* xs = Xs + buf * XU;
        const uint4* gs = Gs + buf * DW3_U;
        const int bi = g * 256 + wv * 32 + li;
        const f16x8 gh = as_f16x8(gs[bi]), gl = as_f16x8(gs[512 + bi]);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int ai = g * NX + mt * 32 + li;
            const f16x8 ah = as_f16x8(xs[ai]), al = as_f16x8(xs[2 * NX + ai]);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gl, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, gh, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gh, acc[mt], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the rows just prefetched
    }
#undef DW3E_LOAD
#undef DW3E_STORE

    float* out = partial + (size_t)chunk * NX * 256;
    const int col = wv * 32 + li;
    float sg, ig;
    scale_from_max_bits(gmax[col], sg, ig);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int k = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            float sx, ix;
            scale_from_max_bits(k < K1 ? xmax1[k] : xmax2[k - K1], sx, ix);
            out[(size_t)k * 256 + col] = acc[mt][r] * (ix * ig);
        }
    }
    if (isG && partial_db != nullptr) *reinterpret_cast<float2*>(partial_db + (size_t)chunk * 256 + c2 * 2) = colsum;
}

}  // namespace dgm
