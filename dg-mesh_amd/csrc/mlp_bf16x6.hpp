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

// every weight matrix of a network in one launch: blockIdx.y = job (the jobs travel in the kernel argument block)
__global__ void __launch_bounds__(256) mlp_prep6_batch_kernel(const Prep6Batch b) {
    prep6_one(b.job[blockIdx.y], blockIdx.x * 256 + threadIdx.x);
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

// ---- the trunk-layer GEMM, weights stationary in REGISTERS ------------------------------------------------------------
// C[M x 256] = [A1 | A2] * B  (EPI 0 / EPI 1 as above).  The whole B operand of a wave -- all K of its NT*32 output
// columns, three bf16 planes -- lives in registers for the lifetime of the kernel (K = 256, NT = 2: 384 of the 512
// registers a wave owns at one wave per SIMD); the matrix-core operands are read straight from there.  What streams is
// the activation side only, in tiles of 32 rows:
//   * persistent grid, one 4-wave workgroup per CU; wave w owns columns [w*NT*32, (w+1)*NT*32) (+ blockIdx.y * 128
//     when NT == 1, the K = 352 skip layer, whose planes would not fit otherwise);
//   * tile t+2 is fetched with plain coalesced float4 loads (8 rows x 128 B per wave instruction) into registers while
//     tile t is multiplied; at the top of the next step it is split into the three bf16 planes and written to LDS as
//     [k step][plane][k half][row] 16-byte granules (double buffered), so an A fragment is one ds_read_b128 and a wave
//     reads 1 KiB contiguous -- no bank conflicts on either side;
//   * per tile and wave: KS * 3 fragment reads feed KS * 6 * NT MFMAs; one barrier per tile; nothing but the
//     activations ever crosses LDS, and no weight byte is re-read.
// HBM traffic is the algorithmic minimum (A once, C once); rows are balanced over the CUs at 32-row granularity.
template <int EPI, int KS, int NT, int NW>  // NW waves, each NT*32 columns: NW * NT * 32 = 256 (or 128 with gridDim.y = 2)
__global__ void __launch_bounds__(NW * 64)
mlp_gemm6r_kernel(int M, int ntiles, const float* __restrict__ A1, int lda1, int K1, const float* __restrict__ A2, int lda2,
                  const uint4* __restrict__ Bp, const float* __restrict__ bias, unsigned* __restrict__ mask,
                  float* __restrict__ C) {
    constexpr int K = KS * 16;
    constexpr int PU = KS * 6 * 32;     // granules per plane tile
    constexpr int NR = 4 * (K / 32) / NW;  // producer blocks (8 rows x 32 floats) per wave and tile
    static_assert(NR * NW == 4 * (K / 32), "producer blocks must divide over the waves");
    __shared__ uint4 Ps[2 * PU];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    const int col0 = blockIdx.y * (NW * NT * 32) + wv * (NT * 32);
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;

    bf16x8 bh[KS][NT], bm[KS][NT], bl[KS][NT];
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const uint4* b = Bp + ((size_t)ks * 6 + g) * 256 + col0 + nt * 32 + li;
            bh[ks][nt] = as_bf16x8(b[0]), bm[ks][nt] = as_bf16x8(b[512]), bl[ks][nt] = as_bf16x8(b[1024]);
        }
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) bv[nt] = (EPI == 0) ? bias[col0 + nt * 32 + li] : 0.f;

    // producer role: block i of this wave = rows (b & 3) * 8 + (lane & 7), floats (b >> 2) * 32 + (lane >> 3) * 4 .. +3
    float4 R[NR];
    const int p_r = lane & 7, p_k = (lane >> 3) * 4;
#define R6_LOAD(tile_)                                                                                                 \
    {                                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < NR; i_++) {                                                            \
            const int b_ = wv * NR + i_;                                                                               \
            int grow_ = (tile_) * 32 + (b_ & 3) * 8 + p_r;                                                             \
            grow_ = grow_ < M ? grow_ : M - 1;                                                                         \
            const int k_ = (b_ >> 2) * 32;                                                                             \
            const float* s_ = (k_ < K1) ? (A1 + (size_t)grow_ * lda1 + k_ + p_k) : (A2 + (size_t)grow_ * lda2 + (k_ - K1) + p_k); \
            R[i_] = *reinterpret_cast<const float4*>(s_);                                                              \
        }                                                                                                              \
    }
#define R6_SPLIT(pb_)                                                                                                  \
    {                                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < NR; i_++) {                                                            \
            const int b_ = wv * NR + i_;                                                                               \
            const int row_ = (b_ & 3) * 8 + p_r;                                                                       \
            const int k_ = (b_ >> 2) * 32 + p_k;                                                                       \
            unsigned h0_, m0_, l0_, h1_, m1_, l1_;                                                                     \
            split2(R[i_].x, R[i_].y, h0_, m0_, l0_);                                                                   \
            split2(R[i_].z, R[i_].w, h1_, m1_, l1_);                                                                   \
            uint2* d_ = reinterpret_cast<uint2*>(&Ps[(pb_) * PU + ((k_ >> 4) * 6 + ((k_ >> 3) & 1)) * 32 + row_]) + ((k_ >> 2) & 1); \
            d_[0] = make_uint2(h0_, h1_);                                                                              \
            d_[2 * 64] = make_uint2(m0_, m1_);                                                                         \
            d_[4 * 64] = make_uint2(l0_, l1_);                                                                         \
        }                                                                                                              \
    }
#define R6_MFMA(pb_)                                                                                                   \
    {                                                                                                                  \
        const uint4* ps = &Ps[(pb_) * PU + g * 32 + li];                                                               \
        /* A fragments are single-buffered (no registers to spare): each plane is re-read right after its last use, */ \
        /* and the products are ordered h, h, h, m, m, l so that every read has MFMAs of the same step to hide behind */ \
        bf16x8 ah = as_bf16x8(ps[0]), am = as_bf16x8(ps[64]), al = as_bf16x8(ps[128]);                                 \
        _Pragma("unroll") for (int ks = 0; ks < KS; ks++) {                                                            \
            const int nx = (ks + 1 < KS ? ks + 1 : ks) * 192;                                                          \
            _Pragma("unroll") for (int nt = 0; nt < NT; nt++) {                                                       \
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ks][nt], acc[nt], 0, 0, 0);                   \
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[ks][nt], acc[nt], 0, 0, 0);                   \
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ks][nt], acc[nt], 0, 0, 0);                   \
            }                                                                                                          \
            ah = as_bf16x8(ps[nx]);                                                                                    \
            _Pragma("unroll") for (int nt = 0; nt < NT; nt++) {                                                       \
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[ks][nt], acc[nt], 0, 0, 0);                   \
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[ks][nt], acc[nt], 0, 0, 0);                   \
            }                                                                                                          \
            am = as_bf16x8(ps[nx + 64]);                                                                               \
            _Pragma("unroll") for (int nt = 0; nt < NT; nt++)                                                         \
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ks][nt], acc[nt], 0, 0, 0);                   \
            al = as_bf16x8(ps[nx + 128]);                                                                              \
        }                                                                                                              \
    }
#define R6_EPILOGUE(tile_)                                                                                             \
    {                                                                                                                  \
        /* addresses: one base per tile and lane; everything else is a compile-time offset (rows of a register are */  \
        /* (r & 3) + 8 (r >> 2) + 4 g: 1 KiB and 8 KiB steps in C, 32 B and 256 B steps in the mask)               */  \
        const int row0_ = (tile_) * 32 + 4 * g;                                                                        \
        float* cb_ = C + (size_t)row0_ * 256 + col0 + li;                                                              \
        unsigned* mb_ = mask + (size_t)row0_ * 8 + (col0 >> 5);                                                        \
        const bool full_ = ((tile_) * 32 + 32 <= M);                                                                   \
        _Pragma("unroll") for (int rb = 0; rb < 16; rb += 8) {                                                         \
            unsigned mws[8][NT];                                                                                       \
            if (EPI == 1) { /* the mask words of eight rows first (clamped rows: no branches), so the loads overlap */  \
                _Pragma("unroll") for (int r = rb; r < rb + 8; r++) {                                                  \
                    const int ro = (r & 3) + 8 * (r >> 2);                                                             \
                    const unsigned* mp = mb_ + ro * 8;                                                                 \
                    if (!full_) mp = mask + (size_t)min(row0_ + ro, M - 1) * 8 + (col0 >> 5);                          \
                    if (NT == 2) {                                                                                     \
                        const uint2 q = *reinterpret_cast<const uint2*>(mp);                                           \
                        mws[r - rb][0] = q.x, mws[r - rb][NT - 1] = q.y;                                               \
                    } else {                                                                                           \
                        mws[r - rb][0] = *mp;                                                                          \
                    }                                                                                                  \
                }                                                                                                      \
            }                                                                                                          \
            _Pragma("unroll") for (int r = rb; r < rb + 8; r++) {                                                      \
                const int ro = (r & 3) + 8 * (r >> 2);                                                                 \
                const bool ok = full_ || (row0_ + ro < M);                                                             \
                unsigned mw[NT];                                                                                       \
                _Pragma("unroll") for (int nt = 0; nt < NT; nt++) {                                                    \
                    float v = acc[nt][r];                                                                              \
                    if (EPI == 0) {                                                                                    \
                        v = fmaxf(v + bv[nt], 0.f);                                                                    \
                        const unsigned long long bal_ = __ballot(v > 0.f);                                             \
                        mw[nt] = g ? (unsigned)(bal_ >> 32) : (unsigned)bal_;  /* select, not a 64-bit VALU shift */   \
                    } else {                                                                                           \
                        v = ((mws[r - rb][nt] >> li) & 1u) ? v : 0.f;                                                  \
                    }                                                                                                  \
                    if (ok) cb_[ro * 256 + nt * 32] = v;                                                               \
                    acc[nt][r] = 0.f;                                                                                  \
                }                                                                                                      \
                if (EPI == 0 && li == 0 && ok) {                                                                       \
                    if (NT == 2) *reinterpret_cast<uint2*>(mb_ + ro * 8) = make_uint2(mw[0], mw[NT - 1]);              \
                    else mb_[ro * 8] = mw[0];                                                                          \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
    }

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[nt][r] = 0.f;

    if (my_tiles > 0) {
        R6_LOAD(blockIdx.x)
        R6_SPLIT(0)
        if (my_tiles > 1) R6_LOAD(blockIdx.x + G)
    }
    __syncthreads();
    // Waves w and w + 4 share a SIMD (NW == 8): the low wave runs [split, load | multiply, store], the high wave
    // [multiply, store | split, load].  Both MFMA phases overlap for most of the step ON PURPOSE: the A fragments are
    // single-buffered (no registers left), so one wave alone is bound by the ds_read -> MFMA latency (measured: 65
    // cycles per MFMA with strict alternation, 40 with the two phases overlapping).
    for (int j = 0; j < my_tiles; j++) {
        const int tile = blockIdx.x + j * G;
        if (NW == 4 || wv < 4) {
            if (j + 1 < my_tiles) R6_SPLIT((j + 1) & 1)      // R holds tile j+1 (fetched during the previous step)
            if (j + 2 < my_tiles) R6_LOAD(tile + 2 * G)
            __builtin_amdgcn_sched_barrier(0);
            R6_MFMA(j & 1)
            R6_EPILOGUE(tile)
        } else {
            R6_MFMA(j & 1)
            R6_EPILOGUE(tile)
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < my_tiles) R6_SPLIT((j + 1) & 1)
            if (j + 2 < my_tiles) R6_LOAD(tile + 2 * G)
        }
        __syncthreads();
    }
#undef R6_LOAD
#undef R6_SPLIT
#undef R6_MFMA
#undef R6_EPILOGUE
}

// ---- weight gradient: partial[chunk][k][j] = sum_{rows of chunk} X[row][k] * G[row][j] ------------------------------
// The contraction runs over ROWS, so both MFMA operands need "8 consecutive rows of one column" per lane: a stage of
// 16 rows is split and transposed on its way into LDS.  A staging thread owns an 8-row x 4-column block (eight
// float4 global loads, one per row), and writes, per column and plane, the 8 row values as one 16-byte granule:
//     Xs[plane][g = row half][column]   (same image as the weight planes above, so the fragment reads are identical)
// Workgroup = 128 K-columns (slab) x 256 gradient columns for one chunk of rows; waves 2 x 2, each 64 x 128.
// The bias gradient rides along: the G staging threads also keep column sums (partial_db[chunk][row half][col]).
static constexpr int DW6_SLAB = 128;
static constexpr int DW6_XU = 6 * DW6_SLAB;  // uint4 per X stage
static constexpr int DW6_GU = 6 * 256;       // uint4 per G stage

__global__ void __launch_bounds__(256, 2)
mlp_dw6_kernel(int M, int rows_per_chunk, const float* __restrict__ X1, int ldx1, int K1, const float* __restrict__ X2,
               int ldx2, int K2, const float* __restrict__ G, float* __restrict__ partial,
               float* __restrict__ partial_db) {
    __shared__ uint4 Xs[2][DW6_XU];
    __shared__ uint4 Gs[2][DW6_GU];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1, g = lane >> 5, li = lane & 31;
    const int Kp = K1 + K2;
    const int slab = blockIdx.x, chunk = blockIdx.y;
    const int r0 = chunk * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    const int nst = (r1 - r0 + 15) >> 4;
    // staging role: threads 0..63 -> X blocks (2 row halves x 32 column quads), 64..191 -> G blocks (2 x 64)
    const bool isX = tid < 64, isG = tid >= 64 && tid < 192;
    const int rg = isX ? (tid >> 5) : ((tid - 64) >> 6);
    const int c4 = isX ? (tid & 31) : ((tid - 64) & 63);
    const float* sp = nullptr;
    int sld = 0;
    if (isX) {
        const int xk = slab * DW6_SLAB + c4 * 4;
        if (xk < K1) sp = X1 + xk, sld = ldx1;
        else if (xk < Kp) sp = X2 + (xk - K1), sld = ldx2;
    } else if (isG) {
        sp = G + c4 * 4, sld = 256;
    }
    uint4* sdst0 = isX ? &Xs[0][rg * DW6_SLAB + c4 * 4] : &Gs[0][rg * 256 + c4 * 4];
    const int sbuf = isX ? DW6_XU : DW6_GU;       // uint4 between the two buffers
    const int splane = isX ? 2 * DW6_SLAB : 512;  // uint4 between planes
    float4 v[8];
    float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);

#define DW6_LOAD(st_)                                                                                 \
    {                                                                                                 \
        const int rb_ = r0 + (st_) * 16 + rg * 8;                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) {                                            \
            v[i_] = make_float4(0.f, 0.f, 0.f, 0.f);                                                  \
            if (sp != nullptr && rb_ + i_ < r1) v[i_] = *reinterpret_cast<const float4*>(sp + (size_t)(rb_ + i_) * sld); \
        }                                                                                             \
    }
#define DW6_STORE(buf_)                                                                               \
    if (isX || isG) {                                                                                 \
        uint4* d_ = sdst0 + (buf_) * sbuf;                                                            \
        uint4 H_, M_, L_;                                                                             \
        split8(v[0].x, v[1].x, v[2].x, v[3].x, v[4].x, v[5].x, v[6].x, v[7].x, H_, M_, L_);           \
        d_[0] = H_, d_[splane] = M_, d_[2 * splane] = L_;                                             \
        split8(v[0].y, v[1].y, v[2].y, v[3].y, v[4].y, v[5].y, v[6].y, v[7].y, H_, M_, L_);           \
        d_[1] = H_, d_[splane + 1] = M_, d_[2 * splane + 1] = L_;                                     \
        split8(v[0].z, v[1].z, v[2].z, v[3].z, v[4].z, v[5].z, v[6].z, v[7].z, H_, M_, L_);           \
        d_[2] = H_, d_[splane + 2] = M_, d_[2 * splane + 2] = L_;                                     \
        split8(v[0].w, v[1].w, v[2].w, v[3].w, v[4].w, v[5].w, v[6].w, v[7].w, H_, M_, L_);           \
        d_[3] = H_, d_[splane + 3] = M_, d_[2 * splane + 3] = L_;                                     \
        if (isG) {                                                                                    \
            _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) {                                        \
                colsum.x += v[i_].x, colsum.y += v[i_].y, colsum.z += v[i_].z, colsum.w += v[i_].w;   \
            }                                                                                         \
        }                                                                                             \
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;

    DW6_LOAD(0)
    DW6_STORE(0)
    __syncthreads();
    for (int st = 0; st < nst; st++) {
        const int buf = st & 1;
        if (st + 1 < nst) DW6_LOAD(st + 1)
        const uint4* xs = Xs[buf];
        const uint4* gs = Gs[buf];
        bf16x8 ah[2], am[2], al[2];
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const int ai = g * DW6_SLAB + wm * 64 + mt * 32 + li;
            ah[mt] = as_bf16x8(xs[ai]), am[mt] = as_bf16x8(xs[2 * DW6_SLAB + ai]), al[mt] = as_bf16x8(xs[4 * DW6_SLAB + ai]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const int bi = g * 256 + wn * 128 + nt * 32 + li;
            const bf16x8 bh = as_bf16x8(gs[bi]), bm = as_bf16x8(gs[512 + bi]), bl = as_bf16x8(gs[1024 + bi]);
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                DGM_MFMA6(acc[mt][nt], ah[mt], am[mt], al[mt], bh, bm, bl)
            }
        }
        if (st + 1 < nst) DW6_STORE(buf ^ 1)
        __syncthreads();
    }
#undef DW6_LOAD
#undef DW6_STORE

    float* out = partial + (size_t)chunk * Kp * 256;
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const int col = wn * 128 + nt * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int k = slab * DW6_SLAB + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (k < Kp) out[(size_t)k * 256 + col] = acc[mt][nt][r];
            }
        }
    if (isG && slab == 0 && partial_db != nullptr)
        *reinterpret_cast<float4*>(partial_db + ((size_t)chunk * 2 + rg) * 256 + c4 * 4) = colsum;
}

// Weight gradient of the K = 256 layers, one 8-wave workgroup per CU covering ALL 256 K columns of a chunk of rows
// (waves 4 x 2, wave tile 64 x 128): G is split and transposed once instead of once per 128-column slab.  EVERY
// thread stages: threads 0..255 the X block of a 16-row stage, 256..511 the G block, each an 8-row x 2-column piece
// (eight float2 loads, six 16-byte LDS writes), so the split work is spread evenly over the eight waves.
__global__ void __launch_bounds__(512)
mlp_dw6b_kernel(int M, int rows_per_chunk, const float* __restrict__ X, int ldx, const float* __restrict__ G,
                float* __restrict__ partial, float* __restrict__ partial_db) {
    __shared__ uint4 Xs[2][DW6_GU];
    __shared__ uint4 Gs[2][DW6_GU];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1, g = lane >> 5, li = lane & 31;
    const int chunk = blockIdx.x;
    const int r0 = chunk * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    const int nst = (r1 - r0 + 15) >> 4;
    // staging role: (row half rg, column pair c2) of X (threads 0..255) or G (256..511)
    const bool isG = tid >= 256;
    const int rg = (tid >> 7) & 1, c2 = tid & 127;
    const float* sp = isG ? (G + c2 * 2) : (X + c2 * 2);
    const int sld = isG ? 256 : ldx;
    uint4* sdst0 = (isG ? &Gs[0][0] : &Xs[0][0]) + rg * 256 + c2 * 2;
    float2 v[8];
    float2 colsum = make_float2(0.f, 0.f);

#define DWB_LOAD(st_)                                                                                 \
    {                                                                                                 \
        const int rb_ = r0 + (st_) * 16 + rg * 8;                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) {                                            \
            v[i_] = make_float2(0.f, 0.f);                                                            \
            if (rb_ + i_ < r1) v[i_] = *reinterpret_cast<const float2*>(sp + (size_t)(rb_ + i_) * sld); \
        }                                                                                             \
    }
#define DWB_STORE(buf_)                                                                               \
    {                                                                                                 \
        uint4* d_ = sdst0 + (buf_) * DW6_GU;                                                          \
        uint4 H_, M_, L_;                                                                             \
        split8(v[0].x, v[1].x, v[2].x, v[3].x, v[4].x, v[5].x, v[6].x, v[7].x, H_, M_, L_);           \
        d_[0] = H_, d_[512] = M_, d_[1024] = L_;                                                      \
        split8(v[0].y, v[1].y, v[2].y, v[3].y, v[4].y, v[5].y, v[6].y, v[7].y, H_, M_, L_);           \
        d_[1] = H_, d_[513] = M_, d_[1025] = L_;                                                      \
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

    // stage st+1 is split and stored at the top of step st (its data was requested a whole step earlier), then the
    // registers are refilled for st+2
    DWB_LOAD(0)
    DWB_STORE(0)
    if (nst > 1) DWB_LOAD(1)
    __syncthreads();
    for (int st = 0; st < nst; st++) {
        const int buf = st & 1;
        if (st + 1 < nst) {
            DWB_STORE(buf ^ 1)
            if (st + 2 < nst) DWB_LOAD(st + 2)
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint4* xs = Xs[buf];
        const uint4* gs = Gs[buf];
        bf16x8 ah[2], am[2], al[2];
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const int ai = g * 256 + wm * 64 + mt * 32 + li;
            ah[mt] = as_bf16x8(xs[ai]), am[mt] = as_bf16x8(xs[512 + ai]), al[mt] = as_bf16x8(xs[1024 + ai]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const int bi = g * 256 + wn * 128 + nt * 32 + li;
            const bf16x8 bh = as_bf16x8(gs[bi]), bm = as_bf16x8(gs[512 + bi]), bl = as_bf16x8(gs[1024 + bi]);
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                DGM_MFMA6(acc[mt][nt], ah[mt], am[mt], al[mt], bh, bm, bl)
            }
        }
        __syncthreads();
    }
#undef DWB_LOAD
#undef DWB_STORE

    float* out = partial + (size_t)chunk * 256 * 256;
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const int col = wn * 128 + nt * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int k = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                out[(size_t)k * 256 + col] = acc[mt][nt][r];
            }
        }
    if (isG && partial_db != nullptr)
        *reinterpret_cast<float2*>(partial_db + ((size_t)chunk * 2 + rg) * 256 + c2 * 2) = colsum;
}

}  // namespace dgm
