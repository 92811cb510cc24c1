// fp32 GEMMs on the bf16 matrix cores ("bf16x6"): every fp32 operand is split EXACTLY into three bf16 numbers
//     x = h + m + l          h = top 8 significand bits, m = next 8, l = last 8   (truncation, so the sum is exact)
// and a product a*b is evaluated as the six partial products whose weight is >= 2^-16 of the leading one:
//     a*b ~= ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh)
// Each partial product of two bf16 numbers is exact in fp32 and is accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
// What is dropped (am*bl + al*bm + al*bl) is below 2^-23 |a*b|, i.e. under one fp32 ulp of the product: the result is
// an fp32 GEMM to rounding (tests/test_mlp.py checks it against fp64 next to a plain fp32 GEMM).
//
// Why: on gfx950 the bf16 MFMA runs at 16x the fp32 MFMA rate (32x32x16 in 32 cycles vs 32x32x2 in 64), so six bf16
// MFMAs per 16-deep K step cost 192 cycles where the fp32 instruction needs 512 -- the 256-wide layers move from
// matrix-core-bound to HBM-bound.
//
// Operand layout (v_mfma_f32_32x32x16_bf16): lane l supplies row/col (l & 31) and the 8 consecutive K indices
// 8*(l >> 5) .. +7 as one 16-byte register quad.  An A fragment of the activations is therefore two float4 global
// loads of the lane's own row -- activations never pass through LDS -- followed by the split in registers.  The B
// operand (weights) is split once per step by mlp_prep6_kernel into the exact LDS image of a K=16 stage,
//     Bp[stage][plane h/m/l][k-half g][column] = 8 bf16 (16 bytes)
// which the GEMM streams into LDS with global_load_lds_dwordx4 (no staging registers, no ds_write) and reads back
// with conflict-free ds_read_b128 (consecutive lanes = consecutive 16-byte granules).
#pragma once
#include "dgm_common.hpp"

namespace dgm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// upper 16 bits of two floats -> one dword (first element in the low half)
__device__ __forceinline__ unsigned pack_hi16(unsigned first, unsigned second) {
    return __builtin_amdgcn_perm(second, first, 0x07060302u);
}

__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    h = pack_hi16(u0, u1);
    m = pack_hi16(v0, v1);
    l = pack_hi16(__float_as_uint(s0), __float_as_uint(s1));
}

__device__ __forceinline__ void split8(float e0, float e1, float e2, float e3, float e4, float e5, float e6, float e7,
                                       uint4& H, uint4& Mi, uint4& L) {
    split2(e0, e1, H.x, Mi.x, L.x);
    split2(e2, e3, H.y, Mi.y, L.y);
    split2(e4, e5, H.z, Mi.z, L.z);
    split2(e6, e7, H.w, Mi.w, L.w);
}

__device__ __forceinline__ bf16x8 as_bf16x8(const uint4 v) { return __builtin_bit_cast(bf16x8, v); }

#define DGM_MFMA6(acc_, ah_, am_, al_, bh_, bm_, bl_)                                   \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_, bh_, acc_, 0, 0, 0);            \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bl_, acc_, 0, 0, 0);            \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am_, bm_, acc_, 0, 0, 0);            \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am_, bh_, acc_, 0, 0, 0);            \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bm_, acc_, 0, 0, 0);            \
    acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bh_, acc_, 0, 0, 0);

// ---- weight planes ------------------------------------------------------------------------------------------------
// Bp[((stage*3 + plane)*2 + g)*ncols + col] holds B[k = stage*16 + g*8 + e][col], e = 0..7.
//   mode 0 (forward, B = W^T through the trunk's K mapping): B[k][col] = W[col][src(k)]   (col < col_valid)
//       Kp == 96 : src(k) = k for k < emb_dim (else zero row);  Kp == 352: [emb | h] -> k, k - 96 + emb_dim;  else k
//   mode 1 (backward data, B = W[:, hoff:hoff+ncols]):      B[k][col] = W[k][hoff + col]  (k < k_valid)
struct Prep6Job {
    int mode, Kp, ncols, in_features, emb_dim, hoff, k_valid, col_valid;
    const float* W;
    uint4* Bp;
};
static constexpr int PREP6_MAX_JOBS = 20;
struct Prep6Batch {
    Prep6Job job[PREP6_MAX_JOBS];
};

__device__ __forceinline__ void prep6_one(const Prep6Job& j, int idx) {
    if (idx >= (j.Kp >> 3) * j.ncols) return;
    const int kg = idx / j.ncols, col = idx % j.ncols;
    float e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int k = kg * 8 + i;
        float v = 0.f;
        if (j.mode == 0) {
            int src = k;
            if (j.Kp == 96) src = k < j.emb_dim ? k : -1;
            else if (j.Kp == 352) src = k < 96 ? (k < j.emb_dim ? k : -1) : k - 96 + j.emb_dim;
            if (src >= 0 && col < j.col_valid) v = j.W[(size_t)col * j.in_features + src];
        } else {
            if (k < j.k_valid) v = j.W[(size_t)k * j.in_features + j.hoff + col];
        }
        e[i] = v;
    }
    uint4 H, Mi, L;
    split8(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], H, Mi, L);
    const int stage = kg >> 1, g = kg & 1;
    uint4* dst = j.Bp + ((size_t)stage * 6 + g) * j.ncols + col;
    dst[0] = H;
    dst[2 * j.ncols] = Mi;
    dst[4 * j.ncols] = L;
}


__global__ void mlp_prep6_kernel(int mode, int Kp, int ncols, int in_features, int emb_dim, int hoff, int k_valid,
                                 int col_valid, const float* __restrict__ W, uint4* __restrict__ Bp) {
    Prep6Job j;
    j.mode = mode, j.Kp = Kp, j.ncols = ncols, j.in_features = in_features, j.emb_dim = emb_dim, j.hoff = hoff;
    j.k_valid = k_valid, j.col_valid = col_valid, j.W = W, j.Bp = Bp;
    prep6_one(j, blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- C[M x ncols] = [A1 | A2] * B -----------------------------------------------------------------------------------
// Workgroup = 4 waves arranged WM x WN; a wave owns MT x NT MFMA tiles (32 x 32 each): rows per workgroup
// WM*MT*32, columns WN*NT*32 (= all of B's columns).
//   EPI 0: C = relu(acc + bias), ReLU mask bits saved: mask[row][col / 32] bit (col % 32)
//   EPI 1: C = acc where the saved mask bit is set, else 0   (backward data -> pre-masked gradient of the layer below)
//   EPI 2: C[row * ldc + col] = acc + bias[col] for col < n_valid   (linear heads)
//   NARROW: A1 is [M x a_valid] with a_valid <= 16 unaligned floats per row (head gradients); K is padded to 16
template <int EPI, int WM, int WN, int MT, int NT, bool NARROW>
__global__ void __launch_bounds__(256, 2)
mlp_gemm6_kernel(int M, const float* __restrict__ A1, int lda1, int K1, const float* __restrict__ A2, int lda2, int K2,
                 int a_valid, const uint4* __restrict__ Bp, const float* __restrict__ bias, unsigned* __restrict__ mask,
                 float* __restrict__ C, int ldc, int n_valid) {
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int NCOLS = WN * NT * 32;
    constexpr int STAGE = 6 * NCOLS;   // uint4 per K=16 stage
    constexpr int NDMA = STAGE / 64;   // wave-wide 1 KiB copies per stage
    __shared__ uint4 Bs[2][STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / WN, wn = wv % WN;
    const int g = lane >> 5, li = lane & 31;
    const int nk = (K1 + K2) >> 4;
    const int row0 = blockIdx.x * (WM * MT * 32) + wm * (MT * 32) + li;

    // The stage copy is issued from inline asm on purpose: through the builtin the compiler treats the LDS-DMA as a
    // store that may alias every later ds_read and puts `s_waitcnt vmcnt(0)` in front of the first fragment read of
    // the CURRENT stage -- which serialises the copy of the next stage (and the A prefetch) with the MFMA phase.
    // Completion is ordered by hand instead: vmcnt(0) + barrier at the end of the step, one full MFMA phase later.
    const unsigned lds_base = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)(&Bs[0][0]));
#define G6_DMA(kt_, buf_)                                                                                              \
    {                                                                                                                  \
        const uint4* src_ = Bp + (size_t)(kt_) * STAGE;                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < (NDMA + 3) / 4; i_++) {                                                \
            const int piece_ = i_ * 4 + wv;                                                                            \
            if (piece_ < NDMA) {                                                                                       \
                const unsigned dst_ = __builtin_amdgcn_readfirstlane(lds_base + ((buf_) * STAGE + piece_ * 64) * 16);   \
                const uint4* g_ = src_ + piece_ * 64 + lane;                                                           \
                unsigned keep_;                                                                                        \
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                             : "=&s"(keep_)                                                                            \
                             : "v"(g_), "s"(dst_)                                                                      \
                             : "memory");                                                                              \
            }                                                                                                          \
        }                                                                                                              \
    }

    float4 ra[MT][2];
#define G6_LOAD_A(kt_)                                                                                                 \
    {                                                                                                                  \
        const int k_ = (kt_) * 16 + g * 8;                                                                             \
        _Pragma("unroll") for (int mt_ = 0; mt_ < MT; mt_++) {                                                         \
            const int row_ = row0 + mt_ * 32;                                                                          \
            ra[mt_][0] = make_float4(0.f, 0.f, 0.f, 0.f);                                                              \
            ra[mt_][1] = make_float4(0.f, 0.f, 0.f, 0.f);                                                              \
            if (row_ < M) {                                                                                            \
                if (NARROW) {                                                                                          \
                    const float* s_ = A1 + (size_t)row_ * lda1;                                                        \
                    float t_[8];                                                                                       \
                    _Pragma("unroll") for (int e_ = 0; e_ < 8; e_++) t_[e_] = (k_ + e_ < a_valid) ? s_[k_ + e_] : 0.f; \
                    ra[mt_][0] = make_float4(t_[0], t_[1], t_[2], t_[3]);                                              \
                    ra[mt_][1] = make_float4(t_[4], t_[5], t_[6], t_[7]);                                              \
                } else {                                                                                               \
                    const float* s_ = (k_ < K1) ? (A1 + (size_t)row_ * lda1 + k_) : (A2 + (size_t)row_ * lda2 + (k_ - K1)); \
                    ra[mt_][0] = *reinterpret_cast<const float4*>(s_);                                                 \
                    ra[mt_][1] = *reinterpret_cast<const float4*>(s_ + 4);                                             \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;

    G6_DMA(0, 0)
    G6_LOAD_A(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        // first the split of this step's A rows (their loads were drained at the end of the previous step), THEN the
        // prefetches: any compiler-placed vmcnt wait for `ra` must sit in front of the new copies, not behind them
        bf16x8 ah[MT], am[MT], al[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            uint4 H, Mi, L;
            split8(ra[mt][0].x, ra[mt][0].y, ra[mt][0].z, ra[mt][0].w, ra[mt][1].x, ra[mt][1].y, ra[mt][1].z, ra[mt][1].w,
                   H, Mi, L);
            ah[mt] = as_bf16x8(H), am[mt] = as_bf16x8(Mi), al[mt] = as_bf16x8(L);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) G6_DMA(kt + 1, buf ^ 1)
        if (kt + 1 < nk) G6_LOAD_A(kt + 1)
        const uint4* bs = Bs[buf];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int bi = g * NCOLS + wn * (NT * 32) + nt * 32 + li;
            const bf16x8 bh = as_bf16x8(bs[bi]), bm = as_bf16x8(bs[2 * NCOLS + bi]), bl = as_bf16x8(bs[4 * NCOLS + bi]);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                DGM_MFMA6(acc[mt][nt], ah[mt], am[mt], al[mt], bh, bm, bl)
            }
        }
        // the stage copied during this step must have landed, and every wave must be done reading the current one
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef G6_DMA
#undef G6_LOAD_A

    // epilogue: D[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31]
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int col = wn * (NT * 32) + nt * 32 + li;
        bv[nt] = 0.f;
        if (EPI == 0) bv[nt] = bias[col];
        if (EPI == 2) bv[nt] = col < n_valid ? bias[col] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = blockIdx.x * (WM * MT * 32) + wm * (MT * 32) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            const bool ok = row < M;
            unsigned mw[NT];  // this wave's NT mask words of the row (NT == 4: one 16-byte access)
            if (EPI == 1) {
#pragma unroll
                for (int nt = 0; nt < NT; nt++) mw[nt] = 0u;
                if (ok) {
                    if (NT == 4) {
                        const uint4 q = *reinterpret_cast<const uint4*>(mask + (size_t)row * 8 + wn * 4);
                        mw[0] = q.x, mw[1 % NT] = q.y, mw[2 % NT] = q.z, mw[3 % NT] = q.w;
                    } else {
#pragma unroll
                        for (int nt = 0; nt < NT; nt++) mw[nt] = mask[(size_t)row * 8 + wn * NT + nt];
                    }
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const int col = wn * (NT * 32) + nt * 32 + li;
                float v = acc[mt][nt][r];
                if (EPI == 0) {
                    v = fmaxf(v + bv[nt], 0.f);
                    const unsigned long long bal = __ballot(v > 0.f);  // low half: row, high half: row + 4
                    mw[nt] = (unsigned)(bal >> (g * 32));
                    if (ok) C[(size_t)row * ldc + col] = v;
                } else if (EPI == 1) {
                    v = ((mw[nt] >> li) & 1u) ? v : 0.f;
                    if (ok) C[(size_t)row * ldc + col] = v;
                } else {
                    if (ok && col < n_valid) C[(size_t)row * ldc + col] = v + bv[nt];
                }
            }
            if (EPI == 0 && li == 0 && ok) {
                if (NT == 4) {
                    *reinterpret_cast<uint4*>(mask + (size_t)row * 8 + wn * 4) = make_uint4(mw[0], mw[1 % NT], mw[2 % NT], mw[3 % NT]);
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) mask[(size_t)row * 8 + wn * NT + nt] = mw[nt];
                }
            }
        }
    }
}



}  // namespace dgm
