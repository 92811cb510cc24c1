// The MLP trunk on "plane-format" activations ("f16x3p", dgm_mlp_set_gemm(3), the default).
//
// Same arithmetic as mlp_f16x3.hpp -- every fp32 operand scaled by a power of two and split into two binary16 numbers, three
// partial products per product on v_mfma_f32_32x32x16_f16, fp32 accumulation -- but the split happens ONCE, in the epilogue of
// the kernel that produces a tensor, and the tensor lives in HBM as its two binary16 planes (4 bytes per element, the traffic
// of fp32):
//     T[row] = [ h : K halves | l : K halves ]          value = (h + l) * 2^-e[row / 32]
// with ONE exponent per 32-row tile (block floating point: the tile's maximum lands in [2^14, 2^15), so an element keeps 22
// significant bits down to 2^-11 of its tile's maximum and 2^-39 of that maximum absolutely below).  A scale that is constant
// over a 32-row tile is constant along the contraction of the layer GEMMs (features) AND, up to a per-tile power of two that the
// weight-gradient kernel folds into one operand, along the contraction of the weight gradients (rows) -- so the three consumers
// of a tensor (next layer, backward data, weight gradient) read the planes as they are: no row maxima, no scales, no split, no
// staging registers.  The layer GEMM streams them into LDS with global_load_lds_dwordx4 and feeds the fragments to the MFMAs
// unchanged; the weight-gradient kernel only transposes 8 x 8 blocks of halves in registers (v_perm).
//
// All tensors are padded to a multiple of 32 rows (the workspace is ours), padded rows hold finite values and the gradient
// tensors' padded rows are exactly zero (they descend from zero rows of dOut), so no kernel predicates rows.
#pragma once
#include <type_traits>

#include "dgm_common.hpp"

namespace dgm {

// ---- the split itself (rounds 2-4 kept these in mlp_f16x3.hpp, the row-format predecessor of this file, retired in round 5):
// every fp32 operand is scaled by a power of two (exact) and split into two binary16 numbers
//     s x = h + l + e        h = rne16(s x),  l = rne16(s x - h),  |e| <= 2^-23 |s x|   (or 2^-25 absolute when l is subnormal)
// and a product is evaluated as the three partial products  ah*bh + ah*bl + al*bh  on v_mfma_f32_32x32x16_f16 (each product of two
// binary16 numbers is exact in fp32, accumulation in fp32).  What is dropped (al*bl) is below 2^-22 |a b|; measured against
// fp64 the split error is 4e-8 of sum |a||b| -- an order below the rounding error of an fp32 GEMM itself
// (tests/test_mlp.py::test_split_arithmetics_are_fp32_gemms).
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
        // round 6, one time value per call: the embedding is [x, PE(x), 0] (64 columns), the time columns sit in the bias
        else if (j.Kp == 64) src = k < 63 ? k : -1;
        else if (j.Kp == 320) src = k < 64 ? (k < 63 ? k : -1) : k - 64 + j.emb_dim;
        return (src >= 0 && col < j.col_valid) ? j.W[(size_t)col * j.in_features + src] : 0.f;
    }
    return k < j.k_valid ? j.W[(size_t)k * j.in_features + j.hoff + col] : 0.f;
}


// exponent e of a tile from the float bits of its non-negative maximum: stored = value * 2^e, maximum -> [2^14, 2^15).
// Clamped so that 2^e and 2^-e are normal floats (an all-zero tile gets a harmless finite scale).
__device__ __forceinline__ int p4_exp_from_max_bits(unsigned bits) {
    int eb = (int)((bits >> 23) & 0xffu);
    eb = eb < 20 ? 20 : (eb > 250 ? 250 : eb);
    return 141 - eb;
}
// max of two floats as ONE instruction: fmaxf() first canonicalises operands whose origin the compiler cannot see (values from
// LDS, asm outputs) with an extra v_max_f32 v, v, v each.  No NaNs here.
__device__ __forceinline__ float p4_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float p4_max_abs(float a, float b) {  // max(a, |b|)
    float r;
    asm("v_max_f32 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float p4_pow2(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }  // e in [-126, 127]

// 2^d (d <= 0) as a pair of binary16 (for v_pk_mul_f16); exact down to the smallest subnormal, 0 below
__device__ __forceinline__ unsigned p4_pow2_h2(int d) {
    unsigned h;
    if (d >= -14) h = (unsigned)(d + 15) << 10;
    else if (d >= -24) h = 1u << (d + 24);
    else h = 0u;
    return h | (h << 16);
}

// (h, l) of two floats scaled by 2^e, without packed-f32 instructions (v_pk_mul_f32 / v_pk_fma_f32 are what the compiler
// makes of split2h(x0 * s, x1 * s, ...) and they cost ~25 cycles each beside MFMAs): v_ldexp_f32, v_cvt_pk_f16_f32, and the
// residual x - h as v_fma_mix_f32 (h read as binary16 straight from the packed register)
__device__ __forceinline__ void p4_split2(float v0, float v1, int e, unsigned& h, unsigned& l) {
    const float x0 = __builtin_amdgcn_ldexpf(v0, e), x1 = __builtin_amdgcn_ldexpf(v1, e);
    const f16x2 hh = __builtin_convertvector((f32x2){x0, x1}, f16x2);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hh), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hh), "v"(x1));
    const f16x2 ll = __builtin_convertvector((f32x2){r0, r1}, f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}

// one wave-wide 1 KiB copy global -> LDS (lane i: 16 bytes from its own global address to lds_dst + 16 i); the destination
// is wave-uniform.  Issued from inline asm: through the builtin the compiler treats the LDS-DMA as a store that may alias
// every later ds_read and drains vmcnt in front of them.  Completion is ordered by hand (s_waitcnt vmcnt(0) + barrier).
__device__ __forceinline__ void p4_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
#ifdef P4_TIMING
// phase timers of mlp_gemm4_kernel<16, 1024, 512, 0, false, 8> (tools/build_variant.sh timing "-DP4_TIMING"): per wave, the
// s_memtime cycles spent in each phase of the steady-state tile steps, summed over the steps
__device__ unsigned long long g_p4_timing[256 * 8 * 16];
#define P4_T(k_)                                                                        \
    {                                                                                   \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                   \
        if (P4_TIME_ON) tacc[k_] += now_ - tlast;                                       \
        tlast = now_;                                                                   \
    }
#else
#define P4_T(k_)
#endif
#define P4_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define P4_STEP_BARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ unsigned p4_lds_addr(const void* p) {
    return (unsigned)(unsigned long)((__attribute__((address_space(3))) const char*)p);
}

// ---- per-matrix maxima + positional encoding ------------------------------------------------------------------------
struct AbsMaxJob {
    const float* W;
    int n;
};
static constexpr int P4_MAX_MATS = 12;
struct AbsMaxBatch {
    int n_jobs;
    AbsMaxJob job[P4_MAX_MATS];
};

// emb planes [Np][2][EW] binary16 (EW = 96, or 64 without the time columns) + one exponent per 32-row tile, rows >= N zero.
//   emb[r] = [x, sin(x 2^0), cos(x 2^0), ..., sin(x 2^9), cos(x 2^9) | t_emb[r] | 0...]   (time_utils.py:24-55)
// Grid: 8 am.n_jobs workgroups that reduce max |W| of an eighth of one weight tensor each into matmax[job][8] (float bits; the
// weight preparation folds the eight) -- the scales of the weight planes; riding along here saves a launch in front of the weight
// preparation, which needs them -- followed by ntiles workgroups (one 32-row tile each).
__device__ __forceinline__ void embed4_tile(const int N, const int tile, const float* __restrict__ x, const float* __restrict__ temb,
                                            const int temb_stride, const int T, unsigned char* __restrict__ Ep, int* __restrict__ Eexp,
                                            const int EW, float (*se)[96 + 1], float* smax);
__global__ void __launch_bounds__(256)
mlp_embed4_kernel(int N, int ntiles, const float* __restrict__ x, const float* __restrict__ temb, int temb_stride, int T,
                  unsigned char* __restrict__ Ep, int* __restrict__ Eexp, const AbsMaxBatch am, unsigned* __restrict__ matmax,
                  const int EW) {
    __shared__ float se[32][96 + 1];
    __shared__ float smax[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int AS = 8;  // workgroups per weight tensor (the maxima jobs come FIRST in the grid: dispatched last they would be its tail)
    const int n_am = am.n_jobs * AS;
    if ((int)blockIdx.x < n_am) {
        const int job = (int)blockIdx.x / AS, part = (int)blockIdx.x % AS;
        const AbsMaxJob& jb = am.job[job];
        float m = 0.f;
        const int per = ((jb.n + AS - 1) / AS + 3) & ~3;  // a multiple of four floats
        const int i0 = part * per, i1 = min(jb.n, i0 + per);
        if ((((uintptr_t)jb.W) & 15) == 0) {
            const float4* p = reinterpret_cast<const float4*>(jb.W + i0);
            const int n4 = i1 > i0 ? (i1 - i0) >> 2 : 0;
            for (int i = tid; i < n4; i += 256) {
                const float4 v = p[i];
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
            for (int i = i0 + n4 * 4 + tid; i < i1; i += 256) m = fmaxf(m, fabsf(jb.W[i]));
        } else {
            for (int i = i0 + tid; i < i1; i += 256) m = fmaxf(m, fabsf(jb.W[i]));
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
        if (lane == 0) smax[wv] = m;
        __syncthreads();
        if (tid == 0) matmax[job * AS + part] = __float_as_uint(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
        return;
    }
    embed4_tile(N, (int)blockIdx.x - n_am, x, temb, temb_stride, T, Ep, Eexp, EW, se, smax);
}
__device__ __forceinline__ void embed4_tile(const int N, const int tile, const float* __restrict__ x, const float* __restrict__ temb,
                                            const int temb_stride, const int T, unsigned char* __restrict__ Ep, int* __restrict__ Eexp,
                                            const int EW, float (*se)[96 + 1], float* smax) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r0 = tile * 32;
    float mx = 0.f;
    // sin / cos: lane < 60 of wave w evaluates one (row, frequency, axis) triple of rows 8 w + 2 it + {0, 1}
#pragma unroll
    for (int it = 0; it < 4; it++) {
        if (lane < 60) {
            const int sub = lane / 30, p = lane - 30 * sub, rl = 8 * wv + 2 * it + sub, r = r0 + rl;
            const int q = p / 3, a = p - 3 * q;
            float sn = 0.f, cs = 0.f;
            if (r < N) sincosf(x[3 * r + a] * (float)(1 << q), &sn, &cs);
            se[rl][3 + 6 * q + a] = sn;
            se[rl][6 + 6 * q + a] = cs;
        }
    }
    // x, the time embedding, the zero padding: 3 + (EW - 63) columns per row (EW = 64: no time columns -- one time value per call
    // is folded into the biases, mlp_prep4c_kernel -- just the zero column 63)
    const int rest = 3 + EW - 63;
    for (int i = tid; i < 32 * rest; i += 256) {
        const int rl = i / rest, j = i - rest * rl, r = r0 + rl;
        float v = 0.f;
        int c;
        if (j < 3) {
            c = j;
            if (r < N) v = x[3 * r + j];
        } else {
            const int t = j - 3;
            c = 63 + t;
            if (EW == 96 && r < N && t < T) v = temb[(size_t)r * temb_stride + t];
        }
        se[rl][c] = v;
        mx = fmaxf(mx, fabsf(v));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    if (lane == 0) smax[wv] = mx;
    __syncthreads();
    // sin / cos columns are bounded by 1: fold that bound in instead of reducing them
    const float tm = fmaxf(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])), 1.0f);
    const int e = p4_exp_from_max_bits(__float_as_uint(tm));
    const float sc = p4_pow2(e);
    if (tid == 0) Eexp[tile] = e;
    // 32 rows x EW / 4 column quads: one 8-byte store per plane and quad
    const int nq = EW >> 2;
    for (int i = tid; i < 32 * nq; i += 256) {
        const int rl = i / nq, c4 = (i - nq * rl) * 4;
        unsigned h0, l0, h1, l1;
        split2h(se[rl][c4] * sc, se[rl][c4 + 1] * sc, h0, l0);
        split2h(se[rl][c4 + 2] * sc, se[rl][c4 + 3] * sc, h1, l1);
        unsigned char* d = Ep + (size_t)(r0 + rl) * (EW * 4) + c4 * 2;
        *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(d + EW * 2) = make_uint2(l0, l1);
    }
}

// ---- weight planes with ONE power-of-two scale per matrix ---------------------------------------------------------------
// Same layout and source mapping as mlp_prep3_all_kernel (Bp[((stage*2 + plane)*2 + g)*ncols + col] = 8 halves of
// B[k = stage*16 + g*8 + e][col] * scale), but the scale comes from the matrix maximum the embed launch left in matmax[mat]:
// in the plane-format GEMM the weights are the M-side operand, a per-column scale would cost a multiplication per output
// element and tile.  inv_scale[0] = 1 / scale.  Grid (ncols / 32 <= 8, jobs).
struct Prep4Job {
    Prep3Job j;
    int mat;
};
static constexpr int P4_MAX_JOBS = 20;
struct Prep4Batch {
    Prep4Job job[P4_MAX_JOBS];
};
__global__ void __launch_bounds__(256)
mlp_prep4_kernel(const Prep4Batch b, const unsigned* __restrict__ matmax) {
    const Prep4Job& pj = b.job[blockIdx.y];
    const Prep3Job& j = pj.j;
    if ((int)blockIdx.x * 32 >= j.ncols) return;
    const int tid = threadIdx.x, col = blockIdx.x * 32 + (tid & 31), slot = tid >> 5, nkg = j.Kp >> 3;
    float sc, inv;
    unsigned mb = 0u;  // (non-negative floats order like their bit patterns)
#pragma unroll
    for (int i = 0; i < 8; i++) mb = max(mb, matmax[pj.mat * 8 + i]);
    scale_from_max_bits(mb, sc, inv);
    if (blockIdx.x == 0 && tid == 0) j.inv_scale[0] = inv;
#pragma unroll
    for (int it = 0; it < 6; it++) {
        const int kg = slot + 8 * it;
        if (kg < nkg) {
            float e[8];
#pragma unroll
            for (int i = 0; i < 8; i++) e[i] = prep3_src(j, kg * 8 + i, col);
            uint4 H, L;
            split2h(e[0] * sc, e[1] * sc, H.x, L.x);
            split2h(e[2] * sc, e[3] * sc, H.y, L.y);
            split2h(e[4] * sc, e[5] * sc, H.z, L.z);
            split2h(e[6] * sc, e[7] * sc, H.w, L.w);
            uint4* dst = j.Bp + ((size_t)(kg >> 1) * 4 + (kg & 1)) * j.ncols + col;
            dst[0] = H;
            dst[2 * j.ncols] = L;
        }
    }
}

// ---- weight planes with ONE power-of-two scale PER OUTPUT COLUMN (round 6; COLSC kernels) ---------------------------------------------
// One scale per matrix keeps 22 bits of an entry only down to 2^-17 of the MATRIX maximum: a network whose hidden units differ in
// scale by 2^+-10 (ReLU is positively homogeneous: such a network computes the same function) has rows and columns 2^+-10 apart in
// every matrix, the entries of one matrix span 2^40, and the smallest -- which multiply the largest activations -- are flushed
// (tests/test_mlp.py::test_plane_format_envelope...).  Here column c of B (forward: output unit c, i.e. row c of W; backward data:
// input feature c, i.e. column hoff + c of W) is scaled by its own power of two: B'[k][c] = B[k][c] 2^ls[c], maximum -> [2^14, 2^15);
// the GEMM's epilogue multiplies output column c by inv_scale[c] = 2^-ls[c] (a power of two: exact), and the forward's bias enters
// pre-scaled, bias_out[c] = bias_in[c] 2^ls[c], so that ReLU still reads the sign of one fused multiply-add.  What is left inside a
// column is the spread over the contraction index only.  The block computes its columns' maxima itself (no maxima pass in front).
struct Prep4cJob {
    Prep3Job j;             // j.inv_scale: [ncols] floats
    const float* bias_in;   // may be NULL
    float* bias_out;        // [ncols]
    int fold;               // 1: the call's time row is folded into this layer's bias (layer 0 and the skip layer, one time value per call):
                            //    bias[c] = bias_in[c] + sum_t W[c][63 + t] t_emb[t]   (R/utils/time_utils.py:104-129: h = cat([x_emb, t_emb]))
};
struct Prep4cBatch {
    Prep4cJob job[P4_MAX_JOBS];
    const float* temb;      // the call's time row (fold jobs), T values
    float* temb_row;        // ... copied here for the backward pass (dW's time columns are db (x) t_emb)
    int T;
};
// A workgroup takes P4C_COLS = 16 output columns (lane = column within the block, 16 k-groups of 8 side by side): 16 workgroups per
// 256-column matrix, 270 in the launch.  (First version: 32 columns per workgroup, 136 workgroups -- the launch lasted as long as one
// CU needs for its 12 k scattered 32-byte row reads, 11.7 us; W is (out, in), a column of the GEMM operand is a row of W.)
static constexpr int P4C_COLS = 16, P4C_SLOTS = 256 / P4C_COLS, P4C_ITERS = 48 / P4C_SLOTS;  // (48 k-groups: Kp <= 384)
__device__ __forceinline__ void prep4c_block(const Prep4cBatch& b, const int bx, const int by) {
    __shared__ float smax[P4C_SLOTS][P4C_COLS], sfold[8][P4C_COLS];
    const Prep4cJob& pj = b.job[by];
    const Prep3Job& j = pj.j;
    if (bx * P4C_COLS >= j.ncols) return;
    const int tid = threadIdx.x, cl = tid % P4C_COLS, col = bx * P4C_COLS + cl, slot = tid / P4C_COLS, nkg = j.Kp >> 3;
    float e[P4C_ITERS][8];
    float m = 0.f;
#pragma unroll
    for (int it = 0; it < P4C_ITERS; it++) {
        const int kg = slot + P4C_SLOTS * it;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            e[it][i] = kg < nkg ? prep3_src(j, kg * 8 + i, col) : 0.f;
            m = fmaxf(m, fabsf(e[it][i]));
        }
    }
    smax[slot][cl] = m;
    if (pj.fold && slot < 8) {  // the fold's dot product, eight partial sums per column: t = slot, slot + 8, ...
        float sacc = 0.f;
        if (col < j.col_valid) {
            const float* wt = j.W + (size_t)col * j.in_features + 63;
            for (int t = slot; t < b.T; t += 8) sacc = fmaf(wt[t], b.temb[t], sacc);
        }
        sfold[slot][cl] = sacc;
    }
    __syncthreads();
    float cm = smax[0][cl];
#pragma unroll
    for (int q = 1; q < P4C_SLOTS; q++) cm = fmaxf(cm, smax[q][cl]);
    float sc, inv;
    scale_from_max_bits(__float_as_uint(cm), sc, inv);
    if (slot == 0) {
        j.inv_scale[col] = inv;
        if (pj.bias_out != nullptr) {
            float bv = (pj.bias_in != nullptr && col < j.col_valid) ? pj.bias_in[col] : 0.f;
            if (pj.fold && col < j.col_valid)  // (round 6, first version: a launch of its own in front of this one, mlp_fold_bias_kernel)
                bv += ((sfold[0][cl] + sfold[1][cl]) + (sfold[2][cl] + sfold[3][cl])) + ((sfold[4][cl] + sfold[5][cl]) + (sfold[6][cl] + sfold[7][cl]));
            pj.bias_out[col] = bv * sc;
        }
    }
    if (bx == 0 && by == 0 && b.temb_row != nullptr && tid < b.T) b.temb_row[tid] = b.temb[tid];
#pragma unroll
    for (int it = 0; it < P4C_ITERS; it++) {
        const int kg = slot + P4C_SLOTS * it;
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

__global__ void __launch_bounds__(256)
mlp_prep4c_kernel(const Prep4cBatch b) {
    prep4c_block(b, (int)blockIdx.x, (int)blockIdx.y);
}
// The embedding planes and the weight preparation in ONE launch (round 6's forms: neither needs anything of the other): the first
// n_prep = 16 x jobs workgroups prepare the weights (dispatched first: their scattered row reads are the longer chain), the rest take
// one 32-row tile of the embedding each.  (Two launches: 13.1 + 10.2 us.)
__global__ void __launch_bounds__(256)
mlp_embed4c_kernel(const int N, const float* __restrict__ x, unsigned char* __restrict__ Ep, int* __restrict__ Eexp, const int EW,
                   const Prep4cBatch b, const int n_prep) {
    if ((int)blockIdx.x < n_prep) {  // (workgroup-uniform)
        prep4c_block(b, (int)blockIdx.x % (256 / P4C_COLS), (int)blockIdx.x / (256 / P4C_COLS));
        return;
    }
    __shared__ float se[32][96 + 1];
    __shared__ float smax[4];
    embed4_tile(N, (int)blockIdx.x - n_prep, x, nullptr, 0, 0, Ep, Eexp, EW, se, smax);
}

// ---- dOut (N, n_out) fp32 -> planes [Np][2][32] + exponents, and the heads' bias-gradient partial sums --------------------
// One wave per 32-row tile (lane = row, column half); columns >= n_out and rows >= N are zero, which is what makes every
// gradient tensor's padded rows exactly zero.  partial_b[tile][16] = column sums of the tile (the heads' bias gradient).
__global__ void __launch_bounds__(256)
mlp_dout4_kernel(int N, int ntiles, int NC, const float* __restrict__ dOut, unsigned char* __restrict__ Dp, int* __restrict__ Dexp,
                 float* __restrict__ partial_b) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wv;
    if (tile >= ntiles) return;
    const int rl = lane & 31, hf = lane >> 5, r = tile * 32 + rl;
    float v[8];
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int c = 8 * hf + i;
        v[i] = (r < N && c < NC) ? dOut[(size_t)r * NC + c] : 0.f;
        mx = fmaxf(mx, fabsf(v[i]));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    const int e = p4_exp_from_max_bits(__float_as_uint(mx));
    const float sc = p4_pow2(e);
    if (lane == 0) Dexp[tile] = e;
    uint4 H, L;
    split2h(v[0] * sc, v[1] * sc, H.x, L.x);
    split2h(v[2] * sc, v[3] * sc, H.y, L.y);
    split2h(v[4] * sc, v[5] * sc, H.z, L.z);
    split2h(v[6] * sc, v[7] * sc, H.w, L.w);
    unsigned char* d = Dp + (size_t)r * 128 + hf * 16;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(d) = H;
    *reinterpret_cast<uint4*>(d + 32) = z;
    *reinterpret_cast<uint4*>(d + 64) = L;
    *reinterpret_cast<uint4*>(d + 96) = z;
    // column sums over the 32 rows of the tile (lanes of one half), fixed order
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float s = v[i];
#pragma unroll
        for (int dd = 16; dd >= 1; dd >>= 1) s += __shfl_xor(s, dd, 64);
        if (rl == 0) partial_b[(size_t)tile * 16 + 8 * hf + i] = s;
    }
}

// ---- the layer GEMM on planes ----------------------------------------------------------------------------------------------
// C^T tile = W-planes (M side: 32 output columns per wave, stationary in registers) x A-planes^T (N side: the 32 rows of a
// tile, from LDS): with the activations on the N side a lane of the accumulator holds ONE row and 16 of the wave's 32 output
// columns -- four runs of four consecutive columns -- so the epilogue emits 8-byte pieces of the output planes (no 2-byte
// stores), needs no ballots for the ReLU mask, and the per-tile input scale is one scalar.
//   EPI 0: Y = relu(acc c + bias)            planes + tile exponent + ReLU mask out          (c = 2^-e_in / weight scale)
//   EPI 1: G' = mask ? acc c : 0             planes + tile exponent out (backward data; mask of the layer below)
//   EPI 2: Y = relu(acc c + Cin)             like EPI 0, Cin = the embedding half of the skip layer incl. its bias
//   EPI 3: out[row][o] = acc c + bias[o]     fp32, o < n_valid (the heads; only wave 0 computes, the others help loading)
//   DUAL : a second weight matrix over the same activations, out2 = acc2 c2 + bias2 (fp32, lane-native layout: the Cin of
//          EPI 2) -- layer 0 and the embedding half of the skip layer share one read of the embedding.
// Persistent grid, one 8-wave workgroup per CU, tiles round-robin.  Per tile step j every wave runs
//   S : stores of tile j-2's output planes (staged in LDS by E2, read back as 1 KiB rows) and of tile j-1's mask block
//   L : the global->LDS copy of tile j+1's rows (this wave's share), straight into the other A buffer
//   M : the MFMAs of tile j, fragments from LDS, with
//   E2: the second half of tile j-1's epilogue sliced in between -- tile maximum (8 wave maxima from LDS) -> exponent ->
//       scale, split, 8-byte stores into the LDS staging tile
//   E1: first half of tile j's epilogue -- unscale, bias / ReLU / mask, element maxima, wave maximum -> LDS
// and one barrier.  The tile exponent needs the maximum over all 8 waves' columns, hence the two halves around a barrier.
// Mask block of a tile: [32 rows][8 waves] words, bit 16 g + 4 q + e of word (row, w) <-> column 32 w + 8 q + 4 g + e
// (private to this kernel: EPI 0 / 2 write it, EPI 1 reads it).
struct Gemm4Args {
    int ntiles, M;
    const unsigned char* A;     // input planes, row pitch ROWB
    const int* Aexp;            // [ntiles]
    const uint4* Bp;            // weight planes
    const float* b_inv;         // [1] 1 / weight scale
    const float* bias;          // [256] (EPI 0) / [n_valid] (EPI 3)
    const unsigned* mask_in;    // EPI 1
    unsigned* mask_out;         // EPI 0 / 2
    unsigned char* C;           // output planes [Np][2][256]
    int* Cexp;                  // [ntiles]
    const float4* cin;          // EPI 2: [ntiles][8][4][64] float4
    const uint4* Bp2;           // DUAL
    const float* b_inv2;
    const float* bias2;
    float4* out2;               // DUAL: [ntiles][8][4][64] float4
    float* out;                 // EPI 3
    int ldo, n_valid;
    int exps_limit;             // tiles of a workgroup whose input exponents come from its LDS table (<= Gemm4Cfg::EXPS); later ones from HBM
};

template <int KS, int ROWB, int PLANEB, int EPI, bool DUAL, int NCW, bool COLSC = false>
struct Gemm4Cfg {
    static constexpr int PITCH = ROWB + 16;
    static constexpr int ABYTES = 32 * PITCH;
    static constexpr int NBUF = 3;                                  // A tiles in LDS: in use, landed, in flight
    static constexpr int A_END = (NBUF * ABYTES + 255) & ~255;
    static constexpr int MI_BYTES = EPI == 1 ? NBUF * 1024 : 0;     // mask blocks of the same three tiles (EPI 1)
    static constexpr bool PLANES_OUT = EPI != 3;
    static constexpr int O_BYTES = PLANES_OUT ? 32768 : 0;          // staging tile of the output planes
    static constexpr int EXPS = 512;                                // input exponents of the workgroup's tiles
    static constexpr int LDS = A_END + MI_BYTES + O_BYTES + 2 * 1024 + EXPS * 4 + 64 + 1024 + (COLSC ? 1024 : 0);  // (+ the bias vector, + the column scales)
};

// Schedule of one tile step j (all waves; MFMAs of tile j in the A buffer j % 3):
//   first half of the K loop : E1 of tile j-1 sliced between the MFMAs (unscale, bias / Cin / mask, ReLU bits, maxima), then
//                              the wave maximum -> LDS, the mask half-words -> LDS, tile j-2's staged rows LDS -> registers
//   MID BARRIER              : s_waitcnt vmcnt(0) + barrier -- the ONE place vector memory is waited for; everything issued
//                              right behind the previous mid barrier has had a whole tile step to complete
//   right behind it          : tile j-2's row stores, the copy of tile j+2 (+ its mask block) into the buffer tile j-1
//                              occupied, tile j-1's exponent
//   second half              : E2 of tile j-1 sliced between the MFMAs (tile exponent from the eight wave maxima, scale,
//                              split, 8-byte stores into the staging tile)
//   END BARRIER              : LDS only
// All vector-memory traffic of a step is one burst with a full step of slack, so neither HBM latency nor a momentary
// shortage of bandwidth stalls the matrix cores; loads and stores may complete in any order (the counter is only ever
// waited down to zero).
// COLSC: a.b_inv points to one inverse scale per output column (mlp_prep4c_kernel) instead of one per matrix, a.bias (EPI 0) is
// pre-scaled; the epilogue multiplies output column c by b_inv[c] behind the ReLU / the mask (powers of two: exact).
template <int KS, int ROWB, int PLANEB, int EPI, bool DUAL, int NCW, bool COLSC = false>
__device__ __forceinline__ void gemm4_body(const Gemm4Args& a, const int bx, const int G, unsigned char* smem, const int tiles_in = -1) {
    using Cfg = Gemm4Cfg<KS, ROWB, PLANEB, EPI, DUAL, NCW, COLSC>;
    static_assert(!COLSC || (!DUAL && EPI != 2), "column scales: the round-6 kernel forms only");
    constexpr int PITCH = Cfg::PITCH, ABYTES = Cfg::ABYTES;
    constexpr int LPR = ROWB / 16;                           // 16-byte pieces (lanes) per row
    constexpr int RPI = LPR == 64 ? 1 : 64 / (LPR + 1);      // rows per copy instruction (one idle lane = the row pad)
    constexpr int NI = (32 + RPI - 1) / RPI;                 // copy instructions per tile
    constexpr int NCOLS = NCW * 32;
    static_assert(!DUAL || EPI == 0, "the second output rides along with a ReLU layer");
    unsigned char* Abuf = smem;                                                  // [3][32][PITCH]
    unsigned char* Mibuf = smem + Cfg::A_END;                                    // [3][1024] mask blocks in (EPI 1)
    unsigned char* Obuf = smem + Cfg::A_END + Cfg::MI_BYTES;                     // [32][1024] staging of the output planes (swizzled)
    unsigned* mbuf = reinterpret_cast<unsigned*>(Obuf + Cfg::O_BYTES);           // [2][32][8] mask words out
    float* tmaxs = reinterpret_cast<float*>(mbuf + 512);                         // [8] wave maxima
    int* exps = reinterpret_cast<int*>(tmaxs + 16);                              // [EXPS]
    float* biasl = reinterpret_cast<float*>(exps + Cfg::EXPS);                   // [256] (EPI 0: read per tile, not held in registers)
    float* scl = biasl + 256;                                                    // [256] (COLSC: inverse column scales)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    // tiles bx, bx + G, ... (tiles_in < 0: all of them up to a.ntiles; else exactly tiles_in of them)
    const int my_tiles = tiles_in >= 0 ? tiles_in : (a.ntiles - bx + G - 1) / G;
    if (my_tiles <= 0) return;
    const bool computing = wv < NCW;
#ifdef P4_TIMING
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif

    // copy geometry of this lane: instruction n covers rows n RPI .. n RPI + RPI - 1
    const int cp_row = LPR == 64 ? 0 : lane / (LPR + 1), cp_piece = LPR == 64 ? lane : lane % (LPR + 1);
    const bool cp_lane_ok = cp_piece < LPR && cp_row < RPI;
    const unsigned abuf_lds = p4_lds_addr(Abuf), mibuf_lds = p4_lds_addr(Mibuf);
#define G4_COPY(tile_, buf_)                                                                                           \
    {                                                                                                                  \
        _Pragma("unroll") for (int n0_ = 0; n0_ < NI; n0_ += 8) {                                                      \
            const int n_ = n0_ + wv;                                                                                   \
            if (n_ < NI) {                                                                                             \
                const int row_ = n_ * RPI + cp_row;                                                                    \
                if (cp_lane_ok && row_ < 32) {                                                                         \
                    const unsigned char* src_ = a.A + ((size_t)(tile_) * 32 + row_) * ROWB + cp_piece * 16;            \
                    p4_glds16(src_, __builtin_amdgcn_readfirstlane(abuf_lds + (buf_) * ABYTES + n_ * RPI * PITCH));    \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
        if (EPI == 1 && wv == 7)                                                                                       \
            p4_glds16(reinterpret_cast<const unsigned char*>(a.mask_in) + (size_t)(tile_) * 1024 + lane * 16,          \
                      __builtin_amdgcn_readfirstlane(mibuf_lds + (buf_) * 1024));                                      \
    }

    // first the tiles (their latency is the longest), then the exponents and the stationary weights
    G4_COPY(bx, 0)
    if (my_tiles > 1) G4_COPY(bx + G, 1)
    for (int t = tid; t < my_tiles && t < Cfg::EXPS; t += 512) exps[t] = a.Aexp[bx + t * G];
    if (EPI == 0 && tid < 256) biasl[tid] = a.bias[tid];
    if (COLSC && EPI != 3 && tid < 256) scl[tid] = tid < NCOLS ? a.b_inv[tid] : 1.0f;
#ifdef P4_TIMING
    const unsigned long long t_p1 = __builtin_amdgcn_s_memtime();
#endif

    // stationary weights: M-side fragments of this wave's 32 output columns
    f16x8 wh[KS], wl[KS];
    f16x8 wh2[DUAL ? KS : 1], wl2[DUAL ? KS : 1];
    float binv = 0.f, binv2 = 0.f;
    float bias[16], bias2[DUAL ? 16 : 1];
    float scol[(COLSC && EPI == 3) ? 16 : 1];
    if (computing) {
        const int col = wv * 32 + li;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const uint4* b = a.Bp + ((size_t)ks * 4 + g) * NCOLS + col;
            wh[ks] = as_f16x8(b[0]), wl[ks] = as_f16x8(b[2 * NCOLS]);
            if (DUAL) {
                const uint4* b2 = a.Bp2 + ((size_t)ks * 4 + g) * NCOLS + col;
                wh2[DUAL ? ks : 0] = as_f16x8(b2[0]), wl2[DUAL ? ks : 0] = as_f16x8(b2[2 * NCOLS]);
            }
        }
        binv = COLSC ? 1.0f : a.b_inv[0];
        if (DUAL) binv2 = a.b_inv2[0];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int o = wv * 32 + (i & 3) + 8 * (i >> 2) + 4 * g;
            bias[i] = 0.f;
            if (EPI == 3) bias[i] = o < a.n_valid ? a.bias[o] : 0.f;
            if (COLSC && EPI == 3) scol[(COLSC && EPI == 3) ? i : 0] = o < a.n_valid ? a.b_inv[o] : 0.f;
            if (DUAL) bias2[DUAL ? i : 0] = a.bias2[o];
        }
    }

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc, acc2;

    // fragment reads PD K steps ahead of their MFMAs; the MFMAs of a tile (two accumulator chains alternated MFMA by MFMA, or
    // one chain per output with DUAL)
#ifndef P4_PD
#define P4_PD 1
#endif
#ifndef P4_CH2
#define P4_CH2 0  // 1: two accumulator chains alternated MFMA by MFMA (16 more registers: spills in the ReLU variants; measured no faster)
#endif
    // ablation switches for tools/build_variant.sh timing experiments (results are wrong with any of them set)
#ifndef P4_ABL
#define P4_ABL 0   // bit 0: no global stores, 1: no tile copies after the first two, 2: no MFMAs, 3: no epilogue slices, 4: no fragment reads
#endif
#define G4_FRAG(ks_) \
    if (!(P4_ABL & 16) || (ks_) < P4_PD) \
    fh[(ks_) % (P4_PD + 1)] = as_f16x8(*reinterpret_cast<const uint4*>(ps + (ks_) * 32)), \
    fl[(ks_) % (P4_PD + 1)] = as_f16x8(*reinterpret_cast<const uint4*>(ps + PLANEB + (ks_) * 32))
#define G4_MFMA(ks_)                                                                                                   \
    {                                                                                                                  \
        const f16x8 ah_ = fh[(ks_) % (P4_PD + 1)], al_ = fl[(ks_) % (P4_PD + 1)];                                      \
        if (P4_ABL & 4) {                                                                                              \
            if ((ks_) == 0) acc = zero16, acc2 = zero16;                                                               \
            acc[0] += (float)ah_[0] + (float)al_[0];                                                                   \
        } else if (DUAL) {                                                                                             \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks_], ah_, (ks_) == 0 ? zero16 : acc, 0, 0, 0);            \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl2[DUAL ? (ks_) : 0], ah_, (ks_) == 0 ? zero16 : acc2, 0, 0, 0); \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks_], al_, acc, 0, 0, 0);                                  \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh2[DUAL ? (ks_) : 0], al_, acc2, 0, 0, 0);                  \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks_], ah_, acc, 0, 0, 0);                                  \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh2[DUAL ? (ks_) : 0], ah_, acc2, 0, 0, 0);                  \
        } else if (!P4_CH2) { /* one accumulator chain */                                                              \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks_], ah_, (ks_) == 0 ? zero16 : acc, 0, 0, 0);            \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks_], al_, acc, 0, 0, 0);                                  \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks_], ah_, acc, 0, 0, 0);                                  \
        } else if ((ks_) == 0) {                                                                                       \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[0], ah_, zero16, 0, 0, 0);                                \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0], ah_, zero16, 0, 0, 0);                                 \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0], al_, acc2, 0, 0, 0);                                  \
        } else if ((ks_) & 1) {                                                                                        \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks_], ah_, acc, 0, 0, 0);                                  \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks_], ah_, acc2, 0, 0, 0);                                \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks_], al_, acc, 0, 0, 0);                                  \
        } else {                                                                                                       \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks_], ah_, acc2, 0, 0, 0);                                \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks_], ah_, acc, 0, 0, 0);                                  \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks_], al_, acc2, 0, 0, 0);                                \
        }                                                                                                              \
    }

#ifdef P4_TIMING
    const unsigned long long t_p2 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_p3 = __builtin_amdgcn_s_memtime();
#endif
    P4_STEP_BARRIER();  // tiles 0 and 1 landed, exponents visible

    if (EPI == 3) {
        // ---- the heads: no output planes, no pipeline -- copy two tiles ahead, multiply (wave 0), store fp32
        int ab = 0;
        for (int j = 0; j < my_tiles; j++) {
            const int tile = bx + j * G;
            if (j + 2 < my_tiles) G4_COPY(tile + 2 * G, ab == 0 ? 2 : ab - 1)
            if (computing) {
                const int e_in = a.Aexp[tile];
                const unsigned char* ps = Abuf + ab * ABYTES + li * PITCH + g * 16;
                f16x8 fh[P4_PD + 1], fl[P4_PD + 1];
#pragma unroll
                for (int i = 0; i < P4_PD && i < KS; i++) G4_FRAG(i);
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    if (ks + P4_PD < KS) G4_FRAG(ks + P4_PD);
                    G4_MFMA(ks)
                }
                const float c = binv * p4_pow2(-__builtin_amdgcn_readfirstlane(e_in));
                const int row = tile * 32 + li;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int o = (i & 3) + 8 * (i >> 2) + 4 * g;
                    if (o < a.n_valid && row < a.M)
                        a.out[(size_t)row * a.ldo + o] = (P4_CH2 ? acc[i] + acc2[i] : acc[i]) * (COLSC ? c * scol[(COLSC && EPI == 3) ? i : 0] : c) + bias[i];
                }
            }
            ab = ab == 2 ? 0 : ab + 1;
            P4_STEP_BARRIER();
        }
        return;
    }

    // ---- the pipelined loop for the plane-producing variants (all eight waves compute)
    float pv[16];             // raw accumulator sums of the previous tile; E1 turns them into its outputs in place, E2 splits them
    float4 cin_prev[EPI == 2 ? 4 : 1];
    // lane constants of the staging-tile swizzle: 8-byte chunk u = 64 p + 8 w + 2 q + g of row r lives at chunk u ^ (r & 15)
    const unsigned st_x8 = (unsigned)(((8 * wv + g) ^ (li & 15)) << 3);
    unsigned char* const ow = Obuf + li * 1024;
    constexpr int H1 = KS >= 2 ? KS / 2 : 1;  // K steps in front of the mid barrier (E1 slices), the rest carry the E2 slices
#ifdef P4_TIMING
    constexpr bool P4_TIME_ON = KS == 16 && EPI == 0;
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long t_loop = tlast;
#endif

    // One tile step: M = MFMAs of tile j, E = both epilogue halves of tile j-1, S = stores of tile j-2.  The flags are
    // compile-time constants in the steady state and run-time values in the drain steps.  ab = j % 3.
    // carried from step to step: the exponent / mask half-word the NEXT step's E1 needs (read a step ahead: no LDS latency at
    // the head of a step)
    int e_next = 0;
    unsigned mh_next = 0u;
    // also fetched at the END of the previous step, so that a step opens with MFMAs instead of LDS latency: the first fragments
    // of the tile and the bias of this lane's sixteen columns (from LDS: held in registers only through the first half)
    f16x8 fh[P4_PD + 1], fl[P4_PD + 1];
    float4 bq[EPI == 0 ? 4 : 1];
    float4 sqn = make_float4(1.f, 1.f, 1.f, 1.f);  // COLSC: the column scales of the next E1 group (fetched a group ahead)
    // the copy of a tile, one instruction at a time (spread over the second half's K steps)
#define G4_COPY1(tile_, buf_, n0_)                                                                                     \
    {                                                                                                                  \
        const int n_ = (n0_) + wv;                                                                                     \
        if ((n0_) < NI && n_ < NI) {                                                                                   \
            const int row_ = n_ * RPI + cp_row;                                                                        \
            if (cp_lane_ok && row_ < 32) {                                                                             \
                const unsigned char* src_ = a.A + ((size_t)(tile_) * 32 + row_) * ROWB + cp_piece * 16;                \
                p4_glds16(src_, __builtin_amdgcn_readfirstlane(abuf_lds + (buf_) * ABYTES + n_ * RPI * PITCH));        \
            }                                                                                                          \
        }                                                                                                              \
    }
    constexpr int NIW = (NI + 7) / 8;  // copy instructions per wave and tile
    auto step = [&](auto HM, auto HE, auto HS, const int j, const int ab) __attribute__((always_inline)) {
        const int pb = j & 1;
        const int tile = bx + j * G;
        const int abp = ab == 0 ? 2 : ab - 1;  // buffer of tile j-1 (= the one tile j+2 goes to)
        const int e_prev = e_next;
        const unsigned mh_prev = mh_next;
        if (HM.value) {  // for the next step's E1 (tile j)
            e_next = j < a.exps_limit ? exps[j] : a.Aexp[tile];  // (beyond the 512-entry table -- N > 4 M rows -- straight from HBM)
            if (EPI == 1) mh_next = reinterpret_cast<const unsigned short*>(Mibuf + ab * 1024)[(li * 8 + wv) * 2 + g];
        }
        const float c = binv * p4_pow2(-e_prev);
        unsigned bits = 0u;
        float m = 0.f;
        int eo = 0;
        float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
        float4 sq = make_float4(1.f, 1.f, 1.f, 1.f);
        uint4 sv[4], smv = make_uint4(0u, 0u, 0u, 0u);
        const unsigned char* ps = Abuf + ab * ABYTES + li * PITCH + g * 16;
        if (HS.value && KS > 1) {  // tile j-2's staged rows (and its mask block) back from LDS, long before E2 overwrites the tile
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int r = wv * 4 + rr;
                sv[rr] = *reinterpret_cast<const uint4*>(Obuf + r * 1024 + ((lane ^ ((r >> 1) & 7)) << 4));
            }
            if ((EPI == 0 || EPI == 2) && wv == 7) smv = reinterpret_cast<const uint4*>(mbuf + pb * 256)[lane];
        }
        P4_T(7)
        // E1 of element i of tile j-1: unscale, bias / Cin / mask, ReLU bit, running maximum -- in place in pv
#define G4_E1(i_)                                                                                                      \
    {                                                                                                                  \
        float t_;                                                                                                      \
        if (COLSC && ((i_) & 3) == 3) {  /* this group's scales were asked for a group ago (the first: at the previous step's end) */ \
            sq = sqn;                                                                                                  \
            if ((i_) >= 4) sqn = *reinterpret_cast<const float4*>(scl + wv * 32 + 8 * (((i_) >> 2) - 1) + 4 * g);       \
        }                                                                                                              \
        const float s_ = ((i_) & 3) == 0 ? sq.x : ((i_) & 3) == 1 ? sq.y : ((i_) & 3) == 2 ? sq.z : sq.w;              \
        if (EPI == 0) {                                                                                                \
            const float4 bb_ = bq[EPI == 0 ? ((i_) >> 2) : 0];                                                         \
            t_ = pv[i_] * c + (((i_) & 3) == 0 ? bb_.x : ((i_) & 3) == 1 ? bb_.y : ((i_) & 3) == 2 ? bb_.z : bb_.w);    \
        } else if (EPI == 2) {                                                                                           \
            const float4 cq_ = cin_prev[EPI == 2 ? ((i_) >> 2) : 0];                                                   \
            t_ = pv[i_] * c + (((i_) & 3) == 0 ? cq_.x : ((i_) & 3) == 1 ? cq_.y : ((i_) & 3) == 2 ? cq_.z : cq_.w);    \
        } else t_ = pv[i_] * c;                                                                                        \
        if (EPI == 1) pv[i_] = ((mh_prev >> (i_)) & 1u) ? (COLSC ? t_ * s_ : t_) : 0.f;                                \
        else {                                                                                                         \
            asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(t_) : "vcc"); \
            pv[i_] = COLSC ? p4_max(t_, 0.f) * s_ : p4_max(t_, 0.f);                                                   \
        }                                                                                                              \
        m = p4_max_abs(m, pv[i_]);                                                                                     \
    }
#define G4_E2(q_)                                                                                                      \
    {                                                                                                                  \
        unsigned h0_, l0_, h1_, l1_;                                                                                   \
        p4_split2(pv[4 * (q_)], pv[4 * (q_) + 1], eo, h0_, l0_);                                                       \
        p4_split2(pv[4 * (q_) + 2], pv[4 * (q_) + 3], eo, h1_, l1_);                                                   \
        unsigned char* d_ = ow + (st_x8 ^ ((q_) << 4));                                                                \
        *reinterpret_cast<uint2*>(d_) = make_uint2(h0_, h1_);                                                          \
        *reinterpret_cast<uint2*>(d_ + 512) = make_uint2(l0_, l1_);                                                    \
    }
    // tile j-2's staged row rr: 1 KiB row store
#define G4_SROW(rr_)                                                                                                   \
    if (HS.value) {                                                                                                    \
        const int r_ = wv * 4 + (rr_);                                                                                 \
        uint4 val_ = sv[rr_];                                                                                          \
        if ((rr_) & 1) val_ = make_uint4(val_.z, val_.w, val_.x, val_.y);                                              \
        if (!(P4_ABL & 1) || val_.x == 0x12345678u)                                                                    \
            *reinterpret_cast<uint4*>(a.C + ((size_t)(tile - 2 * G) * 32 + r_) * 1024 + lane * 16) = val_;             \
    }
        // K steps of the E1 slices: all but the last of the first half (the last one carries the wave reduction and the
        // staged rows' LDS reads, under its MFMAs)
        constexpr int HE1 = H1 >= 2 ? H1 - 1 : 1;
        constexpr int H2 = KS - H1;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            if (HM.value) {
                if (ks + P4_PD < KS) G4_FRAG(ks + P4_PD);
                G4_MFMA(ks)
            }
            if (KS > 1 && ks >= 1 && ks < H1) {  // tile j-2's row stores, early in the first half (one per K step when there are enough)
                constexpr bool WIDE = H1 >= 5;
                if (ks == 1) G4_SROW(0)
                if (ks == (WIDE ? 2 : 1)) G4_SROW(1)
                if (ks == (WIDE ? 3 : (H1 >= 3 ? 2 : 1))) G4_SROW(2)   // (H1 == 2 -- K = 64 -- has the one K step for all four)
                if (ks == (WIDE ? 4 : (H1 >= 3 ? 2 : 1))) {
                    G4_SROW(3)
                    if (HS.value && (EPI == 0 || EPI == 2) && wv == 7) reinterpret_cast<uint4*>(a.mask_out + (size_t)(tile - 2 * G) * 256)[lane] = smv;
                }
            }
            if (HE.value && !(P4_ABL & 8)) {
                if (ks < HE1) {  // E1 slices, elements 15 .. 0 in order (the mask bits shift in)
#pragma unroll
                    for (int i = 15 - (16 * ks) / HE1; i > 15 - (16 * (ks + 1)) / HE1; i--) G4_E1(i)
                }
            }
            if (ks == H1 - 1) {
                P4_T(0)
                if (HE.value) {
                    m = wave_max_nonneg_lane63(m);
                    if (lane == 63) tmaxs[wv] = m;
                    if (EPI != 1) reinterpret_cast<unsigned short*>(mbuf + (1 - pb) * 256)[(li * 8 + wv) * 2 + g] = (unsigned short)bits;
                }
                if (HS.value && KS == 1) {  // (no K loop: staged rows read here, stored right behind the barrier)
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        const int r = wv * 4 + rr;
                        sv[rr] = *reinterpret_cast<const uint4*>(Obuf + r * 1024 + ((lane ^ ((r >> 1) & 7)) << 4));
                    }
                    if ((EPI == 0 || EPI == 2) && wv == 7) smv = reinterpret_cast<const uint4*>(mbuf + pb * 256)[lane];
                }
                P4_T(1)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                P4_T(2)
                P4_STEP_BARRIER();  // MID: the tile maximum needs all eight waves; the vector memory issued so far is done
                P4_T(3)
                if (HE.value) {
                    t0 = *reinterpret_cast<const float4*>(tmaxs);
                    t1 = *reinterpret_cast<const float4*>(tmaxs + 4);
                }
                if (KS == 1) {  // (no K loop to spread anything over)
                    G4_SROW(0) G4_SROW(1) G4_SROW(2) G4_SROW(3)
                    if (HS.value && (EPI == 0 || EPI == 2) && wv == 7) reinterpret_cast<uint4*>(a.mask_out + (size_t)(tile - 2 * G) * 256)[lane] = smv;
                    if (HM.value && j + 2 < my_tiles && !(P4_ABL & 2)) G4_COPY(tile + 2 * G, abp)
                }
                P4_T(4)
            }
            // ---- second half: this step's vector memory, one or two instructions per K step, and the E2 slices
            if (KS > 1 && ks >= H1) {
                const int s2 = ks - H1;
                if (s2 == 0 && HE.value) {  // the tile's exponent from the eight wave maxima
                    const float tm = p4_max(p4_max(p4_max(t0.x, t0.y), p4_max(t0.z, t0.w)), p4_max(p4_max(t1.x, t1.y), p4_max(t1.z, t1.w)));
                    eo = p4_exp_from_max_bits(__float_as_uint(tm));
                    if (tid == 0) a.Cexp[tile - G] = eo;
                }
                if (HM.value && j + 2 < my_tiles && !(P4_ABL & 2)) {
#pragma unroll
                    for (int n = 0; n < NIW; n++)
                        if (s2 == (2 + n < H2 ? 2 + n : H2 - 1)) G4_COPY1(tile + 2 * G, abp, 8 * n)
                    if (EPI == 1 && wv == 7 && s2 == (2 + NIW < H2 ? 2 + NIW : H2 - 1))
                        p4_glds16(reinterpret_cast<const unsigned char*>(a.mask_in) + (size_t)(tile + 2 * G) * 1024 + lane * 16,
                                  __builtin_amdgcn_readfirstlane(mibuf_lds + abp * 1024));
                }
                if (EPI == 2 && HM.value && s2 == 0) {  // Cin of tile j (next step's E1) into the registers E1 has released
#pragma unroll
                    for (int q = 0; q < 4; q++) cin_prev[EPI == 2 ? q : 0] = a.cin[(((size_t)tile * 8 + wv) * 4 + q) * 64 + lane];
                }
                if (HE.value && !(P4_ABL & 8)) {
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (s2 == (H2 >= 8 ? 2 + (3 * q) / 2 : (q * H2) / 4)) G4_E2(q)
                }
            }
            if (KS == 1 && HE.value && !(P4_ABL & 8)) {
                const float tm = p4_max(p4_max(p4_max(t0.x, t0.y), p4_max(t0.z, t0.w)), p4_max(p4_max(t1.x, t1.y), p4_max(t1.z, t1.w)));
                eo = p4_exp_from_max_bits(__float_as_uint(tm));
                if (tid == 0) a.Cexp[tile - G] = eo;
                G4_E2(0) G4_E2(1) G4_E2(2) G4_E2(3)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef G4_E1
#undef G4_E2
#undef G4_SROW
        if (HM.value) {
#pragma unroll
            for (int i = 0; i < 16; i++) pv[i] = (DUAL || !P4_CH2) ? acc[i] : acc[i] + acc2[i];
            if (P4_ABL & 8) {  // (ablation: a never-true sink keeps the accumulators alive)
                float sink_ = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i++) sink_ += pv[i];
                if (sink_ == 1.2345678e33f) a.Cexp[0] = 1;
            }
            if (DUAL) {  // second output: linear, lane-native fp32 (coalesced 1 KiB stores)
                const float c2 = binv2 * p4_pow2(-e_next);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float4 o;
                    o.x = acc2[4 * q + 0] * c2 + bias2[DUAL ? 4 * q + 0 : 0];
                    o.y = acc2[4 * q + 1] * c2 + bias2[DUAL ? 4 * q + 1 : 0];
                    o.z = acc2[4 * q + 2] * c2 + bias2[DUAL ? 4 * q + 2 : 0];
                    o.w = acc2[4 * q + 3] * c2 + bias2[DUAL ? 4 * q + 3 : 0];
                    a.out2[(((size_t)tile * 8 + wv) * 4 + q) * 64 + lane] = o;
                }
            }
        }
        if (HM.value) {  // head start for the next step (its tile landed before this step's mid barrier)
            if (j + 1 < my_tiles) {
                const unsigned char* ps = Abuf + (ab == 2 ? 0 : ab + 1) * ABYTES + li * PITCH + g * 16;
#pragma unroll
                for (int i = 0; i < P4_PD && i < KS; i++) G4_FRAG(i);
            }
            if (EPI == 0) {
#pragma unroll
                for (int q = 0; q < 4; q++) bq[EPI == 0 ? q : 0] = *reinterpret_cast<const float4*>(biasl + wv * 32 + 8 * q + 4 * g);
            }
            if (COLSC) sqn = *reinterpret_cast<const float4*>(scl + wv * 32 + 8 * 3 + 4 * g);
        }
        P4_T(5)
        // END: the staging tile and the A buffer change hands (the reads just issued stay in flight: this wave's LDS writes are
        // older and LDS operations complete in order)
        if (HM.value && j + 1 < my_tiles)
            asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" ::"n"((EPI == 0 ? 4 : 0) + 2 * (P4_PD < KS ? P4_PD : KS) + (COLSC ? 1 : 0)) : "memory");
        else if (HM.value && EPI == 0) asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" ::"n"(4 + (COLSC ? 1 : 0)) : "memory");
        else P4_LDS_BARRIER();
        P4_T(6)
    };
    {   // first fragments of tile 0
        const unsigned char* ps = Abuf + li * PITCH + g * 16;
#pragma unroll
        for (int i = 0; i < P4_PD && i < KS; i++) G4_FRAG(i);
    }
    typedef std::integral_constant<bool, true> T_;
    typedef std::integral_constant<bool, false> F_;
    step(T_{}, F_{}, F_{}, 0, 0);
    if (my_tiles > 1) step(T_{}, T_{}, F_{}, 1, 1);
    int ab = 2;
    for (int j = 2; j < my_tiles; j++) {
        step(T_{}, T_{}, T_{}, j, ab);
        ab = ab == 2 ? 0 : ab + 1;
    }
    if (my_tiles >= 2) step(F_{}, T_{}, T_{}, my_tiles, my_tiles % 3);
    else step(F_{}, T_{}, F_{}, my_tiles, my_tiles % 3);
    step(F_{}, F_{}, T_{}, my_tiles + 1, (my_tiles + 1) % 3);
#ifdef P4_TIMING
    if (P4_TIME_ON && lane == 0 && bx < 256) {
#pragma unroll
        for (int k = 0; k < 8; k++) g_p4_timing[(bx * 8 + wv) * 16 + k] = tacc[k];
        g_p4_timing[(bx * 8 + wv) * 16 + 8] = t_loop - t_entry;
        g_p4_timing[(bx * 8 + wv) * 16 + 9] = __builtin_amdgcn_s_memtime() - t_entry;
        g_p4_timing[(bx * 8 + wv) * 16 + 10] = t_entry;
        g_p4_timing[(bx * 8 + wv) * 16 + 11] = __builtin_amdgcn_s_memtime();
        g_p4_timing[(bx * 8 + wv) * 16 + 12] = my_tiles;
        g_p4_timing[(bx * 8 + wv) * 16 + 13] = t_p1 - t_entry;
        g_p4_timing[(bx * 8 + wv) * 16 + 14] = t_p2 - t_p1;
        g_p4_timing[(bx * 8 + wv) * 16 + 15] = t_p3 - t_p2;
    }
#endif
#undef G4_COPY
#undef G4_COPY1
#undef G4_FRAG
#undef G4_MFMA
}

template <int KS, int ROWB, int PLANEB, int EPI, bool DUAL, int NCW, bool COLSC = false>
__global__ void __launch_bounds__(512)
mlp_gemm4_kernel(const Gemm4Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char p4_smem[];
    gemm4_body<KS, ROWB, PLANEB, EPI, DUAL, NCW, COLSC>(a, (int)blockIdx.x, (int)gridDim.x, p4_smem);
}

// ---- weight gradient on planes -----------------------------------------------------------------------------------------------
// partial[chunk][k][j] = sum over the chunk's rows of X[row][k] * G[row][j], X and G both in plane format.
// A chunk is a run of whole 32-row tiles; tile t carries the scale 2^(ex_t + eg_t).  With E_ref = min over the chunk, the X side
// of tile t is multiplied by 2^(E_ref - E_t) <= 1 in binary16 (exact; skipped when it is 1) and the sum is unscaled by
// 2^-E_ref at the end -- tiles far below the chunk's largest lose their lowest bits, which weigh nothing in the sum.
// Staging: a thread loads an 8-row x 8-column block of one plane (eight 16-byte loads, a wave instruction = two 512-byte
// row segments), transposes it with v_perm into eight granules "8 rows of one column" and writes them where the MFMA
// fragments are read as 16-byte granules: [plane][row group of 8][column], the column position XOR-swizzled inside groups of
// eight so that the eight granule stores of a thread and the fragment reads are both conflict-free.
//   NT == 8: wave w owns gradient columns [32 w, 32 w + 32) against all MT x 32 input columns.
//   NT == 1: 32 gradient columns in all (the heads: G = dOut planes), wave w owns input columns [32 w, 32 w + 32); the result
//            goes to the heads' partial layout partial[(chunk * 16 + o) * 256 + c].
// Bias gradients (db = column sums of G) ride along in the G stagers: v_dot2_f32_f16 against (1, 1) per transposed dword, the
// tile's sums unscaled by 2^-eg_t; partial_db[(chunk * 8 + plane * 4 + row group)][column].
struct Dw4Args {
    int ntiles, tiles_per_chunk;
    const unsigned char* X;
    const int* Xexp;
    const unsigned char* G;
    const int* Gexp;
    float* partial;       // + row offset already applied
    size_t chunk_stride;  // floats between chunks
    float* partial_db;    // may be NULL
#ifdef P4_XCD_REDUCE      // (experiment, variant builds only: see dw4_body)
    unsigned* xsync;      // one arrival word per group of chunks (c % 8), 128 bytes apart, zeroed once; counts up over launches
    float* xpartial;      // [8][256][256]: the groups' pre-reduced tiles
    int xgen, xchunks;    // launches before this one on these words; chunks of this launch
#endif
};

__device__ __forceinline__ int dw4_pos(int col) { return (col & ~7) | ((col ^ (col >> 3)) & 7); }
__device__ __forceinline__ unsigned dw4_dword(const uint4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
// acc + both halves of a dword (v_dot2_f32_f16 against (1, 1))
__device__ __forceinline__ float dw4_dot_ones(unsigned d, float acc) {
    const f16x2 one = {(_Float16)1.0f, (_Float16)1.0f};
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, d), one, acc, false);
}
__device__ __forceinline__ unsigned dw4_pk_mul(unsigned d, unsigned f2) {
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(f16x2, d) * __builtin_bit_cast(f16x2, f2));
}

template <int MT, int NT, int XROWB, int XPLANEB, int GROWB, int GPLANEB>
struct Dw4Cfg {
    static constexpr int XK = MT * 32, GK = NT * 32;
    static constexpr int XU = 8 * XK, GU = 8 * GK;  // uint4 granules per stage buffer: [plane 2][row group 4][column]
    static constexpr int LDS = 2 * (XU + GU) * 16;
};

// (round 5, measured and not kept: a second register set so that the loads of TWO tiles are in flight -- for the stand-alone
// launches, the embedding rows' gradient <3, 8> and the heads' <8, 1>, whose 18 / 6 MFMAs per tile hide little: 36.9 -> 36.4 us and
// 31.0 -> 31.6 us.  Their 3.3-3.8 TB/s is not one outstanding round trip per workgroup.)
template <int MT, int NT, int XROWB, int XPLANEB, int GROWB, int GPLANEB>
__device__ __forceinline__ void dw4_body(const Dw4Args& a, const int chunk, unsigned char* smem) {
    using Cfg = Dw4Cfg<MT, NT, XROWB, XPLANEB, GROWB, GPLANEB>;
    constexpr int XK = Cfg::XK, GK = Cfg::GK, XU = Cfg::XU, GU = Cfg::GU;
    constexpr int XCG = XK / 8, GCG = GK / 8;    // column groups
    constexpr int XB = XCG * 8, GB = GCG * 8;    // stager threads per operand
    constexpr int MTW = NT == 8 ? MT : 1;        // m-tiles per wave
    static_assert(NT == 8 || (NT == 1 && MT == 8), "wave decomposition");
    static_assert(XB <= 256 && GB <= 256, "stager layout");
    uint4* const dw4_lds = reinterpret_cast<uint4*>(smem);
    uint4* Xs = dw4_lds;             // [2][XU]
    uint4* Gs = dw4_lds + 2 * XU;    // [2][GU]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    const int t0 = chunk * a.tiles_per_chunk, t1 = min(a.ntiles, t0 + a.tiles_per_chunk);
    const int nst = t1 - t0;
    if (nst <= 0) return;

    // E_ref = min over the chunk's tiles of ex + eg
    int eref = 0x7fffffff;
    for (int t = t0; t < t1; t++) eref = min(eref, a.Xexp[t] + a.Gexp[t]);
    eref = __builtin_amdgcn_readfirstlane(eref);

    // stager geometry
    const bool isX = tid < XB, isG = tid >= 256 && tid < 256 + GB;
    const int st = isG ? tid - 256 : tid;
    const int ncg = isG ? GCG : XCG;
#ifndef DW4_ROWMAJOR_STAGERS
#define DW4_ROWMAJOR_STAGERS 1
#endif
    // stager -> (column group, plane, row group): plane next to the column group, so that a wave's load instruction covers whole
    // rows ([h | l] = one contiguous 1 KiB at K = 256) instead of two half rows eight rows apart
    const int cg = st % ncg;
    const int pl = DW4_ROWMAJOR_STAGERS ? (st / ncg) & 1 : st / (ncg * 4);
    const int rg = DW4_ROWMAJOR_STAGERS ? st / (ncg * 2) : (st / ncg) & 3;
    const unsigned char* src = isG ? a.G + (size_t)pl * GPLANEB + cg * 16 : a.X + (size_t)pl * XPLANEB + cg * 16;
    const int srow = isG ? GROWB : XROWB;
    uint4* sdst = (isG ? Gs : Xs) + (pl * 4 + rg) * (isG ? GK : XK) + cg * 8;
    const int sbuf = isG ? GU : XU;
    const int swz = cg & 7;
    uint4 R[8];
    float cs[8];
#pragma unroll
    for (int i = 0; i < 8; i++) cs[i] = 0.f;

#define DW4_LOAD(t_)                                                                                                   \
    if (isX || isG) {                                                                                                  \
        const unsigned char* p_ = src + ((size_t)(t_) * 32 + rg * 8) * srow;                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) R[i_] = *reinterpret_cast<const uint4*>(p_ + (size_t)i_ * srow); \
    }
    // transpose the 8 x 8 block: granule c (column 8 cg + c) = rows 0..7 = four dwords of two rows each
#define DW4_STORE(t_, buf_)                                                                                            \
    if (isX || isG) {                                                                                                  \
        const int et_ = a.Xexp[t_] + a.Gexp[t_];                                                                       \
        const unsigned f2_ = p4_pow2_h2(eref - et_);                                                                   \
        const bool scale_ = isX && et_ != eref;                                                                        \
        float tsum_[8];                                                                                                \
        _Pragma("unroll") for (int c_ = 0; c_ < 8; c_++) {                                                             \
            const unsigned sel_ = (c_ & 1) ? 0x07060302u : 0x05040100u;                                                \
            uint4 o_;                                                                                                  \
            o_.x = __builtin_amdgcn_perm(dw4_dword(R[1], c_ >> 1), dw4_dword(R[0], c_ >> 1), sel_);                    \
            o_.y = __builtin_amdgcn_perm(dw4_dword(R[3], c_ >> 1), dw4_dword(R[2], c_ >> 1), sel_);                    \
            o_.z = __builtin_amdgcn_perm(dw4_dword(R[5], c_ >> 1), dw4_dword(R[4], c_ >> 1), sel_);                    \
            o_.w = __builtin_amdgcn_perm(dw4_dword(R[7], c_ >> 1), dw4_dword(R[6], c_ >> 1), sel_);                    \
            if (isG && a.partial_db != nullptr) {                                                                      \
                float s_ = dw4_dot_ones(o_.x, 0.f);                                                                    \
                s_ = dw4_dot_ones(o_.y, s_), s_ = dw4_dot_ones(o_.z, s_), s_ = dw4_dot_ones(o_.w, s_);                 \
                tsum_[c_] = s_;                                                                                        \
            }                                                                                                          \
            if (scale_) o_.x = dw4_pk_mul(o_.x, f2_), o_.y = dw4_pk_mul(o_.y, f2_), o_.z = dw4_pk_mul(o_.z, f2_), o_.w = dw4_pk_mul(o_.w, f2_); \
            sdst[(buf_) * sbuf + (c_ ^ swz)] = o_;                                                                     \
        }                                                                                                              \
        if (isG && a.partial_db != nullptr) {                                                                          \
            const float ig_ = p4_pow2(-a.Gexp[t_]);                                                                    \
            _Pragma("unroll") for (int c_ = 0; c_ < 8; c_++) cs[c_] += tsum_[c_] * ig_;                                \
        }                                                                                                              \
    }

    f32x16 acc[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[mt][r] = 0.f;

    // fragment positions of this lane
    const int gcol = NT == 8 ? wv * 32 + li : li;
    const int gpos = dw4_pos(gcol);

    DW4_LOAD(t0)
    DW4_STORE(t0, 0)
    if (nst > 1) DW4_LOAD(t0 + 1)
    __syncthreads();
    for (int s = 0; s < nst; s++) {
        const int buf = s & 1;
        if (s + 1 < nst) {
            DW4_STORE(t0 + s + 1, buf ^ 1)
            if (s + 2 < nst) DW4_LOAD(t0 + s + 2)
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint4* xs = Xs + buf * XU;
        const uint4* gs = Gs + buf * GU;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const int kg = 2 * kk + g;
            const f16x8 gh = as_f16x8(gs[kg * GK + gpos]), gl = as_f16x8(gs[(4 + kg) * GK + gpos]);
            if (MTW == 1) {
                const int xpos = dw4_pos(wv * 32 + li);
                const f16x8 ah = as_f16x8(xs[kg * XK + xpos]), al = as_f16x8(xs[(4 + kg) * XK + xpos]);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gl, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, gh, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gh, acc[0], 0, 0, 0);
            } else {
#pragma unroll
                for (int mp = 0; mp < (MTW + 1) / 2; mp++) {
                    const int m0 = 2 * mp, m1 = 2 * mp + 1 < MTW ? 2 * mp + 1 : 2 * mp;
                    const int xp0 = dw4_pos(m0 * 32 + li), xp1 = dw4_pos(m1 * 32 + li);
                    const f16x8 ah0 = as_f16x8(xs[kg * XK + xp0]), al0 = as_f16x8(xs[(4 + kg) * XK + xp0]);
                    if (m1 != m0) {  // two accumulators alternate
                        const f16x8 ah1 = as_f16x8(xs[kg * XK + xp1]), al1 = as_f16x8(xs[(4 + kg) * XK + xp1]);
                        acc[m0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, gl, acc[m0], 0, 0, 0);
                        acc[m1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, gl, acc[m1], 0, 0, 0);
                        acc[m0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, gh, acc[m0], 0, 0, 0);
                        acc[m1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, gh, acc[m1], 0, 0, 0);
                        acc[m0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, gh, acc[m0], 0, 0, 0);
                        acc[m1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, gh, acc[m1], 0, 0, 0);
                    } else {
                        acc[m0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, gl, acc[m0], 0, 0, 0);
                        acc[m0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, gh, acc[m0], 0, 0, 0);
                        acc[m0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, gh, acc[m0], 0, 0, 0);
                    }
                }
            }
        }
        P4_LDS_BARRIER();  // LDS-only: the rows just prefetched stay in flight
    }
#undef DW4_LOAD
#undef DW4_STORE

    if (NT == 8) {
        float* out = a.partial + (size_t)chunk * a.chunk_stride;
        const int col = wv * 32 + li;
#pragma unroll
        for (int mt = 0; mt < MTW; mt++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int k = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                out[(size_t)k * 256 + col] = ldexpf(acc[mt][r], -eref);
            }
        }
#ifdef P4_XCD_REDUCE
        // EXPERIMENT (measured with tools/power_probe.py, not in the product): the chunks of one XCD (c % 8 --
        // consecutive workgroup ids go round the XCDs) meet at a device counter once their partial tiles are written, and each
        // then sums its 1 / members share of the group's tiles in fixed chunk order into ONE tile per group, so that the final
        // reduction would read 8 tiles per layer instead of 128.
        if (MT == 8 && a.xsync != nullptr) {
            const int grp = chunk & 7, member = chunk >> 3;
            const int members = (a.xchunks - grp + 7) >> 3;
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                atomicAdd(&a.xsync[grp * 32], 1u);
                const unsigned want = (unsigned)members * (unsigned)(a.xgen + 1);
                for (int spin = 0; spin < (1 << 20); spin++) {  // (bounded: a probe must not hang the box)
                    if (__hip_atomic_load(&a.xsync[grp * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            // this member's rows of the 256-row tile: [r0, r1); 64 float4 per row
            const int rows_per = (256 + members - 1) / members, r0 = member * rows_per, r1 = min(256, r0 + rows_per);
            for (int i = r0 * 64 + tid; i < r1 * 64; i += 512) {
                float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int m = 0; m < members; m++) {
                    const float4 v = *reinterpret_cast<const float4*>(a.partial + (size_t)(grp + 8 * m) * a.chunk_stride + (size_t)i * 4);
                    sacc.x += v.x, sacc.y += v.y, sacc.z += v.z, sacc.w += v.w;
                }
                *reinterpret_cast<float4*>(a.xpartial + ((size_t)grp * 65536 + (size_t)i * 4)) = sacc;
            }
        }
#endif
    } else {
        if (li < 16) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int c = wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                a.partial[((size_t)chunk * 16 + li) * 256 + c] = ldexpf(acc[0][r], -eref);
            }
        }
    }
    if (isG && a.partial_db != nullptr) {
        float* d = a.partial_db + ((size_t)chunk * 8 + pl * 4 + rg) * GK + cg * 8;
        *reinterpret_cast<float4*>(d) = make_float4(cs[0], cs[1], cs[2], cs[3]);
        *reinterpret_cast<float4*>(d + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
    }
}

template <int MT, int NT, int XROWB, int XPLANEB, int GROWB, int GPLANEB>
__global__ void __launch_bounds__(512)
mlp_dw4_kernel(const Dw4Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char p4_smem[];
    dw4_body<MT, NT, XROWB, XPLANEB, GROWB, GPLANEB>(a, (int)blockIdx.x, p4_smem);
}

// ---- backward data of layer l and the weight gradient of layer l in ONE launch ------------------------------------------------
// Both consume G_l and nothing of each other, and they stress different parts of the chip: the layer GEMM is limited by power
// in the matrix cores at ~0.5 of the HBM roof, the weight gradient by HBM.  The first n_dw workgroups run the weight-gradient
// body on n_dw row chunks, the others the layer GEMM on the remaining CUs.  Besides running a memory-bound and a compute-bound
// stream side by side this cuts the per-chunk partial tiles (and the reduction that reads them) from one per CU to n_dw.
template <bool COLSC>
__global__ void __launch_bounds__(512)
mlp_bwd_pair_kernel(const Gemm4Args ga, const Dw4Args da, const int n_dw, const int chunked) {
    extern __shared__ __attribute__((aligned(16))) unsigned char p4_smem[];
    if ((int)blockIdx.x < n_dw) {
        dw4_body<8, 8, 1024, 512, 1024, 512>(da, (int)blockIdx.x, p4_smem);
    } else if (chunked) {
        // The GEMM workgroup w takes the SAME run of tiles as weight-gradient workgroup w (blockIdx w and n_dw + w land on the
        // same XCD when n_dw is a multiple of 8): both stream the same rows of G_l at about the same time, so the second
        // reader can be served by that XCD's L2 instead of the fabric.
        const int w = (int)blockIdx.x - n_dw;
        const int first = w * da.tiles_per_chunk;
        const int cnt = min(da.tiles_per_chunk, ga.ntiles - first);
        gemm4_body<16, 1024, 512, 1, false, 8, COLSC>(ga, first, 1, p4_smem, cnt > 0 ? cnt : 0);
    } else {
        gemm4_body<16, 1024, 512, 1, false, 8, COLSC>(ga, (int)blockIdx.x - n_dw, (int)gridDim.x - n_dw, p4_smem);
    }
}

// ---- gradient w.r.t. the positions -------------------------------------------------------------------------------------------
// dL/dx of the trunk: the embedding PE(x) feeds layer 0 and (skip) layer 5, so dL/dPE = G_0 W_0[:, :63] + G_5 W_5[:, :63] and
// dL/dx_d = dPE[d] + sum_k 2^k (cos(2^k x_d) dPE[3 + 6 k + d] - sin(2^k x_d) dPE[6 + 6 k + d])        (R/utils/time_utils.py:30-55).
// Needed where a network's INPUT carries a gradient: the appearance network on mesh vertices moved by deform_back
// (R/utils/renderer.py:179-181).  Two launches of one kernel, when G_5 / G_0 are the live gradient buffer: the first writes
// G_5 W_5[:, :63] per row (fp32, 64 floats) to a scratch area, the second adds G_0 W_0[:, :63] and applies the derivative of PE.
// G is in plane format; the products run in fp32 on the vector ALU: K = 256, 63 columns -- 2 x 3.3 GFLOP at N = 100 k, a small
// fraction of a network pass, and the vertex count of the mesh phase is smaller still.  One 32-row tile per workgroup, thread
// (ty, tx) owns rows {2 ty, 2 ty + 1} x columns {4 tx .. 4 tx + 3}; G and W go through LDS in K chunks of 32.
// The same kernel serves a time input PER ROW (temb_stride != 0): dL/dt_emb[r] = G_5[r] W_5[:, 63:63+T] + G_0[r] W_0[:, 63:63+T],
// columns [col0, col0 + ncol) of W with ncol <= 64; finish 0: product into scratch, 1: + scratch, derivative of PE -> dX (N, 3),
// 2: + scratch -> out rows of `ldo` floats (dX = out).
__global__ void __launch_bounds__(256)
mlp_dx4_kernel(int N, const unsigned char* __restrict__ G, const int* __restrict__ Gexp, const float* __restrict__ W, int in_features,
               int col0, int ncol, float* __restrict__ scratch, int finish, const float* __restrict__ x, float* __restrict__ dX,
               int ldo) {
    __shared__ float sG[32][32 + 1];
    __shared__ float sW[32][64 + 1];
    __shared__ float sE[32][64 + 1];
    const int tile = blockIdx.x, r0 = tile * 32, tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    const float scale = __builtin_ldexpf(1.0f, -Gexp[tile]);  // value = (h + l) * 2^-e
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int k0 = 0; k0 < 256; k0 += 32) {
        {   // G chunk: 32 rows x 32 k, h + l   (thread: row tid / 8, four k)
            const int r = tid >> 3, kk = (tid & 7) * 4;
            const unsigned char* row = G + (size_t)(r0 + r) * 1024;
            const uint2 h = *reinterpret_cast<const uint2*>(row + (k0 + kk) * 2);
            const uint2 l = *reinterpret_cast<const uint2*>(row + 512 + (k0 + kk) * 2);
            const f16x2 h0 = __builtin_bit_cast(f16x2, h.x), h1 = __builtin_bit_cast(f16x2, h.y);
            const f16x2 l0 = __builtin_bit_cast(f16x2, l.x), l1 = __builtin_bit_cast(f16x2, l.y);
            sG[r][kk + 0] = (float)h0[0] + (float)l0[0];
            sG[r][kk + 1] = (float)h0[1] + (float)l0[1];
            sG[r][kk + 2] = (float)h1[0] + (float)l1[0];
            sG[r][kk + 3] = (float)h1[1] + (float)l1[1];
        }
        for (int i = tid; i < 32 * 64; i += 256) {  // W chunk: rows k0 .. k0 + 31 (output units), columns 0 .. 62 (+ one zero)
            const int kk = i >> 6, c = i & 63;
            sW[kk][c] = c < ncol ? W[(size_t)(k0 + kk) * in_features + col0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < 32; kk++) {
            const float g0 = sG[2 * ty][kk], g1 = sG[2 * ty + 1][kk];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float wv = sW[kk][4 * tx + c];
                acc[0][c] = fmaf(g0, wv, acc[0][c]);
                acc[1][c] = fmaf(g1, wv, acc[1][c]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int r = r0 + 2 * ty + e;
        float4 v = make_float4(acc[e][0] * scale, acc[e][1] * scale, acc[e][2] * scale, acc[e][3] * scale);
        float4* sp = reinterpret_cast<float4*>(scratch + (size_t)r * 64 + 4 * tx);  // (rows are padded to whole tiles)
        if (finish) {
            const float4 p = *sp;
            v.x += p.x, v.y += p.y, v.z += p.z, v.w += p.w;
            sE[2 * ty + e][4 * tx + 0] = v.x, sE[2 * ty + e][4 * tx + 1] = v.y, sE[2 * ty + e][4 * tx + 2] = v.z, sE[2 * ty + e][4 * tx + 3] = v.w;
        } else {
            *sp = v;
        }
    }
    if (!finish) return;
    if (finish == 2) {  // per-row time gradient: the sums as they are
        __syncthreads();
        for (int i = tid; i < 32 * ncol; i += 256) {
            const int rl = i / ncol, c = i - rl * ncol;
            if (r0 + rl < N) dX[(size_t)(r0 + rl) * ldo + c] = sE[rl][c];
        }
        return;
    }
    __syncthreads();
    if (tid < 96) {
        const int rl = tid / 3, d = tid - 3 * rl, r = r0 + rl;
        if (r < N) {
            const float xv = x[3 * r + d];
            float g = sE[rl][d];
#pragma unroll
            for (int q = 0; q < 10; q++) {
                float sn, cs;
                const float f = (float)(1 << q);
                sincosf(xv * f, &sn, &cs);
                g += f * (cs * sE[rl][3 + 6 * q + d] - sn * sE[rl][6 + 6 * q + d]);
            }
            dX[3 * r + d] = g;
        }
    }
}

}  // namespace dgm
