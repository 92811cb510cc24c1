// Deformation / appearance MLP trunk for gfx950.
//
// Replaces the nn.Linear + F.relu chain of DeformNetwork* / AppearanceNetwork (R/utils/time_utils.py:104-129,
// 178-204, 252-266, 310-323): positional encoding of x, 8 x 256 ReLU layers with the skip re-injection of
// [PE(x), t_emb] before layer 5, linear heads -- forward and backward (dX, dW, db, dt_emb) -- plus the single-row
// time branch timenet(PE(t)).
//
// Arithmetics of the 256-wide GEMMs, selected by dgm_mlp_set_gemm() / DGM_MLP_GEMM (the list and the default are further down,
// at g_gemm_mode):
//  * f16x3p (mlp_planes.hpp, default): power-of-two scaled two-way binary16 split, three partial products per fp32 product on the
//    f16 matrix cores, activations and gradients kept in HBM as their two planes;
//  * f32 (this file): v_mfma_f32_32x32x2_f32; 64 x 128 output tile per 256-thread workgroup, K consumed in 16-deep
//    register-staged global -> LDS stages (A k-major with row pitch 66 = 2 mod 8: conflict-free transposing stores and
//    fragment reads), dW as row-chunk x 128-column-slab partial tiles.
// (Rounds 1-4 also shipped "bf16x6" and "f16x3" -- the same split arithmetic on fp32 rows, re-split by every consumer; round 5
// retired them: the plane path now takes per-row time inputs and batches of any size itself.)
// Common to both:
//  * the skip layer reads its input as TWO K-segments ([emb | h4]) -- the concatenation is never materialised;
//  * bias + ReLU are the forward epilogue, which also saves the ReLU mask as bits; the backward-data epilogue applies
//    the mask of the layer BELOW, so each backward GEMM directly emits the next layer's pre-masked gradient;
//  * weight gradients reduce over the rows in fixed order (partials + ordered reduction: deterministic, no atomics);
//    bias gradients ride along as column sums of the same G tiles;
//  * t is the same for every row in training, so dL/dt_emb = db . W[:, t-columns] (no per-row GEMM); the general
//    per-row case has its own small kernel.
#include <stdlib.h>
#include <string.h>

#include "dgm_common.hpp"
#include "mlp_planes.hpp"
#include "mlp_planes5.hpp"

namespace dgm {

static constexpr int MLP_W = 256;     // trunk width (the reference hard-codes W=256)
static constexpr int MLP_EMB = 96;    // padded width of [PE(x) | t_emb]: 63 + 30 (blender) or 63 + 21, zero padded
static constexpr int MLP_XE = 63;     // PE(x) width: 3 + 3*2*10
static constexpr int GM = 64, GK = 16, GAP = 66;  // GEMM tile rows, K stage, padded LDS row stride of A (= 2 mod 8)
static constexpr int DW_ROWS = 512;   // rows per dW chunk
static constexpr int DW_SLAB = 128;   // K columns per dW workgroup
static constexpr int HD_ROWS = 256;   // rows per head-gradient chunk

// ---- weight preparation -----------------------------------------------------------------------------------------
// Wt (forward B operand): [Kp x 256] with Wt[k][j] = W[j][src(k)], zero rows for padding.
//   layer 0: Kp = 96,  src(k) = k for k < emb_dim
//   skip   : Kp = 352, src(k) = k for k < emb_dim ; src(k) = k - 96 + emb_dim for k >= 96
//   others : Kp = 256, src(k) = k
// Wd (backward-data B operand): [256 x 256] with Wd[k][j] = W[k][hoff + j]   (hoff = emb_dim for the skip layer)
__global__ void mlp_prep_kernel(int in_features, int Kp, int emb_dim, int is_skip, const float* __restrict__ W,
                                float* __restrict__ Wt, float* __restrict__ Wd) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < Kp * MLP_W) {
        const int k = idx / MLP_W, j = idx % MLP_W;
        int src = -1;
        if (Kp == MLP_W) src = k;
        else if (k < MLP_EMB) src = k < emb_dim ? k : -1;
        else src = k - MLP_EMB + emb_dim;
        Wt[idx] = src >= 0 ? W[(size_t)j * in_features + src] : 0.f;
    }
    if (Wd != nullptr && idx < MLP_W * MLP_W) {
        const int k = idx / MLP_W, j = idx % MLP_W;
        Wd[idx] = W[(size_t)k * in_features + (is_skip ? emb_dim : 0) + j];
    }
}

// ---- positional encoding -------------------------------------------------------------------------------------------
// emb[r] = [x, sin(x 2^0), cos(x 2^0), ..., sin(x 2^9), cos(x 2^9) | t_emb[r] | 0...]   (time_utils.py:24-55)
// One wave per TWO rows: lanes 0..59 evaluate sin AND cos of one (row, frequency, axis) triple with a single argument
// reduction (sincosf; the kernel is VALU-bound -- 60 transcendentals per row); then the 64 lanes copy x, the time embedding and
// the zero padding of both rows (36 columns each).
__global__ void __launch_bounds__(256)
mlp_embed_kernel(int N, const float* __restrict__ x, const float* __restrict__ temb, int temb_stride, int T,
                 float* __restrict__ emb) {
    const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2, lane = threadIdx.x & 63;
    if (r0 >= N) return;
    if (lane < 60) {
        const int sub = lane / 30, p = lane - 30 * sub, r = r0 + sub;
        if (r < N) {
            const int q = p / 3, a = p - 3 * q;
            float sn, cs;
            sincosf(x[3 * r + a] * (float)(1 << q), &sn, &cs);
            float* e = emb + (size_t)r * MLP_EMB + 3 + 6 * q + a;
            e[0] = sn;
            e[3] = cs;
        }
    }
    constexpr int REST = 3 + (MLP_EMB - MLP_XE);  // x and everything behind the positional encoding
    for (int i = lane; i < 2 * REST; i += 64) {
        const int sub = i / REST, j = i - REST * sub, r = r0 + sub;
        if (r >= N) continue;
        float* e = emb + (size_t)r * MLP_EMB;
        if (j < 3) e[j] = x[3 * r + j];
        else {
            const int t = j - 3;
            e[MLP_XE + t] = t < T ? temb[(size_t)r * temb_stride + t] : 0.f;
        }
    }
}

// ---- the 256-wide GEMM: C[M x 256] = [A1 | A2][M x (K1+K2)] * Bt[(K1+K2) x 256] ------------------------------------
// EPI 0: C = relu(acc + bias), and the ReLU mask is saved as bits: mask[row][col / 32] bit (col % 32)
// EPI 1: C = acc where the saved mask bit of the layer below is set, else 0  (backward data: the product is
//        directly the gradient w.r.t. the pre-activation of the layer below)
// 64 x 128 output tile per 256-thread workgroup (blockIdx.y = column half; 4 waves side by side, each 64 x 32 =
// two MFMA tiles, 32 accumulator VGPRs).  Small tiles keep the 100k-row problem balanced over 256 CUs (3126 tiles)
// and the low register count lets 5 workgroups share a CU, which is what hides the global->LDS staging latency.
static constexpr int GN = 128;
template <int EPI>
__global__ void __launch_bounds__(256)
mlp_gemm_kernel(int M, const float* __restrict__ A1, int lda1, int K1, const float* __restrict__ A2, int lda2, int K2,
                const float* __restrict__ Bt, const float* __restrict__ bias, unsigned* __restrict__ mask,
                float* __restrict__ C) {
    __shared__ float As[2][GK * GAP];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK * GN];
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int m0 = blockIdx.x * GM, n0 = blockIdx.y * GN;
    const int nk = (K1 + K2) / GK;
    const int a_row = tid >> 2, a_c4 = tid & 3;  // A tile: 64 rows x 4 float4
    const int b_k0 = tid >> 5, b_j4 = tid & 31;  // B tile: rows b_k0 and b_k0 + 8, 32 float4 per row
    const bool a_ok = (m0 + a_row) < M;
    // register staging of the next K stage (plain scalars + macros: an array or a by-reference lambda capture here
    // is demoted to scratch memory by the compiler)
    float4 ra, rb0, rb1;
#define MLP_LOAD_STAGE(kt_)                                                                                          \
    {                                                                                                                \
        const int k_ = (kt_) * GK;                                                                                   \
        const float* src_ = (k_ < K1) ? (A1 + (size_t)(m0 + a_row) * lda1 + k_)                                      \
                                      : (A2 + (size_t)(m0 + a_row) * lda2 + (k_ - K1));                              \
        ra = a_ok ? *reinterpret_cast<const float4*>(src_ + a_c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);             \
        const float* bsrc_ = Bt + (size_t)(k_ + b_k0) * MLP_W + n0 + b_j4 * 4;                                       \
        rb0 = *reinterpret_cast<const float4*>(bsrc_);                                                               \
        rb1 = *reinterpret_cast<const float4*>(bsrc_ + 8 * MLP_W);                                                   \
    }
#define MLP_STORE_STAGE(buf_)                                                                                        \
    {                                                                                                                \
        float* a_ = As[buf_] + (a_c4 * 4) * GAP + a_row;                                                             \
        a_[0] = ra.x;                                                                                                \
        a_[GAP] = ra.y;                                                                                              \
        a_[2 * GAP] = ra.z;                                                                                          \
        a_[3 * GAP] = ra.w;                                                                                          \
        float* b_ = Bs[buf_] + b_k0 * GN + b_j4 * 4;                                                                 \
        *reinterpret_cast<float4*>(b_) = rb0;                                                                        \
        *reinterpret_cast<float4*>(b_ + 8 * GN) = rb1;                                                               \
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) acc0[r] = 0.f, acc1[r] = 0.f;

    MLP_LOAD_STAGE(0)
    MLP_STORE_STAGE(0)
    __syncthreads();
    const int a_off = lane & 31, b_off = wn * 32 + (lane & 31), kh = lane >> 5;
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) MLP_LOAD_STAGE(kt + 1)
        const float* as = As[buf];
        const float* bs = Bs[buf];
#pragma unroll
        for (int kk = 0; kk < GK / 2; kk++) {
            const int k = 2 * kk + kh;
            const float a0 = as[k * GAP + a_off], a1 = as[k * GAP + a_off + 32];
            const float b0 = bs[k * GN + b_off];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc1, 0, 0, 0);
        }
        if (kt + 1 < nk) MLP_STORE_STAGE(buf ^ 1)
        __syncthreads();
    }
    // epilogue: D[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31]
    const int col = n0 + wn * 32 + (lane & 31);
    const int mword = (n0 >> 5) + wn;  // 32-column group of this wave
    const float bv = (EPI == 0) ? bias[col] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            float v = mt == 0 ? acc0[r] : acc1[r];
            if (EPI == 0) {
                v = fmaxf(v + bv, 0.f);
                const unsigned long long bal = __ballot(v > 0.f);  // low half: row, high half: row + 4
                if ((lane & 31) == 0 && row < M) mask[(size_t)row * 8 + mword] = (unsigned)(bal >> (kh * 32));
            } else {
                const unsigned bits = row < M ? mask[(size_t)row * 8 + mword] : 0u;
                v = ((bits >> (lane & 31)) & 1u) ? v : 0.f;
            }
            if (row < M) C[(size_t)row * MLP_W + col] = v;
        }
    }
}

#undef MLP_LOAD_STAGE
#undef MLP_STORE_STAGE

// ---- weight gradient: partial[chunk][k][j] = sum_{rows of chunk} X[row][k] * G[row][j] --------------------------------
__global__ void __launch_bounds__(512)
mlp_dw_kernel(int M, const float* __restrict__ X1, int ldx1, int K1, const float* __restrict__ X2, int ldx2, int K2,
              const float* __restrict__ G, float* __restrict__ partial, float* __restrict__ partial_db) {
    __shared__ __attribute__((aligned(16))) float Xs[2][GK * DW_SLAB];
    __shared__ __attribute__((aligned(16))) float Gs[2][GK * MLP_W];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 2, wn = wv & 3;
    const int Kp = K1 + K2;
    const int slab = blockIdx.x, chunk = blockIdx.y;
    const int r0 = chunk * DW_ROWS;
    const int nst = DW_ROWS / GK;
    const int x_r = tid >> 5, x_c4 = tid & 31;      // X tile: 16 rows x 32 float4
    const int g_r0 = tid >> 6, g_j4 = tid & 63;     // G tile: rows g_r0, g_r0 + 8
    const int xk = slab * DW_SLAB + x_c4 * 4;       // concatenated K index of this thread's float4
    float4 rx, rg0, rg1;
    auto load_stage = [&](int st) {
        const int row = r0 + st * GK;
        const int rr = row + x_r;
        rx = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr < M && xk < Kp)
            rx = (xk < K1) ? *reinterpret_cast<const float4*>(X1 + (size_t)rr * ldx1 + xk)
                           : *reinterpret_cast<const float4*>(X2 + (size_t)rr * ldx2 + (xk - K1));
        const int ga = row + g_r0, gb = row + g_r0 + 8;
        rg0 = ga < M ? *reinterpret_cast<const float4*>(G + (size_t)ga * MLP_W + g_j4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        rg1 = gb < M ? *reinterpret_cast<const float4*>(G + (size_t)gb * MLP_W + g_j4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_stage = [&](int buf) {
        *reinterpret_cast<float4*>(Xs[buf] + x_r * DW_SLAB + x_c4 * 4) = rx;
        *reinterpret_cast<float4*>(Gs[buf] + g_r0 * MLP_W + g_j4 * 4) = rg0;
        *reinterpret_cast<float4*>(Gs[buf] + (g_r0 + 8) * MLP_W + g_j4 * 4) = rg1;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    float colsum = 0.f;  // bias gradient of column `tid` (slab 0, threads 0..255)
    const bool do_db = (slab == 0) && (tid < MLP_W) && partial_db != nullptr;

    load_stage(0);
    store_stage(0);
    __syncthreads();
    const int a_off = wm * 64 + (lane & 31), b_off = wn * 64 + (lane & 31), kh = lane >> 5;
    for (int st = 0; st < nst; st++) {
        const int buf = st & 1;
        if (st + 1 < nst) load_stage(st + 1);
        const float* xs = Xs[buf];
        const float* gs = Gs[buf];
#pragma unroll
        for (int kk = 0; kk < GK / 2; kk++) {
            const int k = 2 * kk + kh;  // row inside the stage = reduction index
            const float a0 = xs[k * DW_SLAB + a_off], a1 = xs[k * DW_SLAB + a_off + 32];
            const float b0 = gs[k * MLP_W + b_off], b1 = gs[k * MLP_W + b_off + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (do_db) {
#pragma unroll
            for (int r = 0; r < GK; r++) colsum += gs[r * MLP_W + tid];
        }
        if (st + 1 < nst) store_stage(buf ^ 1);
        __syncthreads();
    }
    float* out = partial + (size_t)chunk * Kp * MLP_W;
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const int col = wn * 64 + nt * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int k = slab * DW_SLAB + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (k < Kp) out[(size_t)k * MLP_W + col] = acc[mt][nt][r];
            }
        }
    if (do_db) partial_db[(size_t)chunk * MLP_W + tid] = colsum;
}

// dW[j][dst(k)] = sum_chunks partial[c][k][j] (PyTorch (out, in) layout, padding rows dropped); db[j] likewise
__global__ void mlp_reduce_dw_kernel(int chunks, int db_chunks, int Kp, int in_features, int emb_dim,
                                     const float* __restrict__ partial, const float* __restrict__ partial_db,
                                     float* __restrict__ dW, float* __restrict__ db) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < Kp * MLP_W) {
        const int k = idx / MLP_W, j = idx % MLP_W;
        int dst;
        if (Kp == MLP_W) dst = k;
        else if (k < MLP_EMB) dst = k < emb_dim ? k : -1;
        else dst = k - MLP_EMB + emb_dim;
        if (dst >= 0) {
            // sixteen independent partial sums keep sixteen loads in flight (a single dependent chain of
            // `chunks` loads is latency bound); the order is still fixed
            float sp[16];
#pragma unroll
            for (int u = 0; u < 16; u++) sp[u] = 0.f;
            const float* src = partial + (size_t)k * MLP_W + j;
            const size_t cs = (size_t)Kp * MLP_W;
            int c = 0;
            for (; c + 16 <= chunks; c += 16) {
#pragma unroll
                for (int u = 0; u < 16; u++) sp[u] += src[(size_t)(c + u) * cs];
            }
            for (; c < chunks; c++) sp[0] += src[(size_t)c * cs];
#pragma unroll
            for (int u = 8; u >= 1; u >>= 1)
#pragma unroll
                for (int v = 0; v < u; v++) sp[v] += sp[v + u];
            const float s = sp[0];
            dW[(size_t)j * in_features + dst] = s;
        }
    }
    if (idx < MLP_W && db != nullptr) {
        float sp[8];
#pragma unroll
        for (int u = 0; u < 8; u++) sp[u] = 0.f;
        int c = 0;
        for (; c + 8 <= db_chunks; c += 8) {
#pragma unroll
            for (int u = 0; u < 8; u++) sp[u] += partial_db[(size_t)(c + u) * MLP_W + idx];
        }
        for (; c < db_chunks; c++) sp[0] += partial_db[(size_t)c * MLP_W + idx];
        db[idx] = ((sp[0] + sp[1]) + (sp[2] + sp[3])) + ((sp[4] + sp[5]) + (sp[6] + sp[7]));
    }
}

// The same reduction in ONE pass for the matrix-core paths (256 partial tiles of 256 KB per layer: the read is the cost).
struct ReduceDwJob {
    int chunks, db_rows, Kp, in_features, nblocks;
    int emb_rows, k_valid;  // Kp != 256 jobs: rows [0, emb_rows) are the embedding's (the first k_valid of them real columns), the rest the trunk's
    int dst_off;  // Kp == 256 jobs: first input feature the rows go to (the skip layer's trunk rows as a job of their own)
    const float* partial;
    const float* partial_db;
    float* dW;
    float* db;
};
struct ReduceDwBatch {
    int emb_dim, n_jobs;
    ReduceDwJob job[9];
    // blockIdx.y == n_jobs: the heads' partial sums (mlp_heads_bwd_kernel) ride along, see reduce_heads_body
    int h_chunks, h_bchunks, h_nout;  // (h_bchunks: rows of h_partial_b -- the plane path leaves one per 32-row tile)
    const float* h_partial_W;
    const float* h_partial_b;
    float* h_dW;
    float* h_db;
};
__device__ __forceinline__ void reduce_heads_body(int bx, int o, int chunks, int bchunks, const float* __restrict__ partial_W,
                                                  const float* __restrict__ partial_b, float* __restrict__ dWh,
                                                  float* __restrict__ dbh);
// All eight layers of a network in ONE launch at the end of its backward pass (blockIdx.y = layer; every layer keeps its own
// partial tiles until then).  Workgroup = a KT x JT piece of the (k, j) plane as 64 float4 positions (one wave-wide load =
// KT row segments of JT * 4 contiguous bytes of one partial tile), times four groups of chunks (the four waves): every
// thread streams its group's chunks with sixteen 16-byte loads in flight, the four group sums meet in LDS (fixed order), and
// the piece is written transposed into the PyTorch (out, in) layout.  With eight layers in the grid there are enough
// workgroups (>= 2048) to read in segments of 256+ bytes and still fill the chip, which a single layer's launch could not.
// The first 32 workgroups of a layer also reduce eight columns each of its bias-gradient rows (db_rows of them).
// (round 4, measured at N = 100 k: 86-88 us whatever the piece shape -- RDW_KT = 1, 2, 4, 8, i.e. segments of 1 KB .. 128 B --
// and whatever the distance between consecutive chunks' tiles -- 256 KB, or padded by 256 B / 4 KB / 8.25 KB against channel
// aliasing: 323 MB of L2 misses at 3.7 TB/s is what a read of tiles the previous launches just wrote gets here)
#ifndef RDW_KT
#define RDW_KT 4
#endif
static constexpr int RDW_JT = 256 / RDW_KT;  // KT * JT = 256 elements per workgroup
static inline int reduce_dw_blocks(int Kp) { return (Kp / RDW_KT) * (MLP_W / RDW_JT); }
__global__ void __launch_bounds__(256)
mlp_reduce_dw_all_kernel(const ReduceDwBatch rb) {
    constexpr int KT = RDW_KT, JT = RDW_JT, JQ = JT / 4, JB = MLP_W / JT;
    __shared__ float4 red[4][64];
    __shared__ float redb[32][8];
    if ((int)blockIdx.y == rb.n_jobs) {  // (workgroup-uniform branch)
        if ((int)blockIdx.x < 8 * rb.h_nout)
            reduce_heads_body(blockIdx.x & 7, blockIdx.x >> 3, rb.h_chunks, rb.h_bchunks, rb.h_partial_W, rb.h_partial_b, rb.h_dW, rb.h_db);
        return;
    }
    const ReduceDwJob& jb = rb.job[blockIdx.y];
    if ((int)blockIdx.x >= jb.nblocks) return;
    const int chunks = jb.chunks, db_rows = jb.db_rows, Kp = jb.Kp, in_features = jb.in_features, emb_dim = rb.emb_dim;
    const float* __restrict__ partial = jb.partial;
    const float* __restrict__ partial_db = jb.partial_db;
    float* __restrict__ dW = jb.dW;
    float* __restrict__ db = jb.db;
    const int tid = threadIdx.x, pos = tid & 63, grp = tid >> 6;
    const int k0 = ((int)blockIdx.x / JB) * KT, j0 = ((int)blockIdx.x % JB) * JT;
    {
        const int k = k0 + pos / JQ, j = j0 + (pos % JQ) * 4;
        const int per = (chunks + 3) >> 2;
        const int c1 = min(chunks, (grp + 1) * per);
        int c = grp * per;
        const size_t cs = (size_t)Kp * MLP_W;
        const float* src = partial + (size_t)k * MLP_W + j;
        float4 sp[16];
#pragma unroll
        for (int u = 0; u < 16; u++) sp[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (; c + 16 <= c1; c += 16) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = *reinterpret_cast<const float4*>(src + (size_t)(c + u) * cs);
#pragma unroll
            for (int u = 0; u < 16; u++) sp[u].x += v[u].x, sp[u].y += v[u].y, sp[u].z += v[u].z, sp[u].w += v[u].w;
        }
        for (; c < c1; c++) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)c * cs);
            sp[0].x += v.x, sp[0].y += v.y, sp[0].z += v.z, sp[0].w += v.w;
        }
#pragma unroll
        for (int u = 8; u >= 1; u >>= 1)
#pragma unroll
            for (int v = 0; v < u; v++)
                sp[v].x += sp[v + u].x, sp[v].y += sp[v + u].y, sp[v].z += sp[v + u].z, sp[v].w += sp[v + u].w;
        red[grp][pos] = sp[0];
    }
    const bool do_db = db != nullptr && blockIdx.x < 32;
    if (do_db) {
        const int col = blockIdx.x * 8 + (tid & 7), g32 = tid >> 3;
        float s = 0.f;
        for (int r = g32; r < db_rows; r += 32) s += partial_db[(size_t)r * MLP_W + col];
        redb[g32][tid & 7] = s;
    }
    __syncthreads();
    {   // thread -> row k0 + tid % KT, column j0 + tid / KT: consecutive threads write consecutive k of one output row
        const int kk = tid % KT, jj = tid / KT;
        const float* r = reinterpret_cast<const float*>(&red[0][0]) + (kk * JQ + (jj >> 2)) * 4 + (jj & 3);
        const float s = ((r[0] + r[256]) + r[512]) + r[768];
        const int k = k0 + kk;
        int dst;
        if (Kp == MLP_W) dst = k + jb.dst_off;
        else if (k < jb.emb_rows) dst = k < jb.k_valid ? k : -1;
        else dst = k - jb.emb_rows + emb_dim;
        if (dst >= 0) dW[(size_t)(j0 + jj) * in_features + dst] = s;
    }
    if (do_db && tid < 8) {
        float s = redb[0][tid];
        for (int g = 1; g < 32; g++) s += redb[g][tid];
        db[blockIdx.x * 8 + tid] = s;
    }
}

// ---- small dense ops around the trunk ----------------------------------------------------------------------------------
// out[r][c] (+)= sum_j A[r][j] * B(j, c) + bias[c],  j < 256, c < NC <= 16, B(j,c) = Bp[j*sj + c*sc].
// Four lanes share a row (each covers 64 of the 256 inputs in 16-byte pieces), so a wave load instruction touches
// 16 rows x 64 contiguous bytes.
__global__ void __launch_bounds__(256)
mlp_rows_small_kernel(int N, int NC, const float* __restrict__ A, const float* __restrict__ Bp, int sj, int sc,
                      const float* __restrict__ bias, float* __restrict__ out, int ldo, int accumulate) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int r = gid >> 2, q = gid & 3;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; c++) acc[c] = 0.f;
    if (r < N) {
        const float* a = A + (size_t)r * MLP_W;
        for (int i = 0; i < 16; i++) {
            const int j = (i * 4 + q) * 4;
            const float4 v = *reinterpret_cast<const float4*>(a + j);
#pragma unroll
            for (int c = 0; c < 16; c++) {
                if (c < NC) {
                    const float* b = Bp + (size_t)j * sj + (size_t)c * sc;
                    acc[c] += v.x * b[0] + v.y * b[sj] + v.z * b[2 * sj] + v.w * b[3 * sj];
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 16; c++) {
        acc[c] += __shfl_xor(acc[c], 1, 64);
        acc[c] += __shfl_xor(acc[c], 2, 64);
    }
    if (r < N && q == 0) {
#pragma unroll
        for (int c = 0; c < 16; c++) {
            if (c < NC) {
                float v = acc[c] + (bias ? bias[c] : 0.f);
                if (accumulate) v += out[(size_t)r * ldo + c];
                out[(size_t)r * ldo + c] = v;
            }
        }
    }
}

// Everything the heads' backward needs from one pass over Y7 (fp32 FMA on the vector units: K = n_out <= 16 is too thin for
// the matrix cores, and the pass is HBM-bound either way -- read Y7, write G7):
//   G7[r][c]                = (sum_o dOut[r][o] * Wh[o][c]) * (Y7[r][c] > 0)        (gradient entering layer 7)
//   partial_W[chunk][o][c]  = sum_rows dOut[r][o] * Y7[r][c],   partial_b[chunk][o] = sum_rows dOut[r][o]
// A wave owns whole rows (lane l: columns 4 l .. 4 l + 3, one 16-byte load / store per row; lane o also loads dOut[r][o], which
// v_readlane turns into the wave-uniform multiplier), keeps its four columns of Wh and its 16 x 4 weight-gradient sums in
// registers, eight rows in flight; the four waves of the workgroup meet in LDS in a fixed order.
typedef float hf2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256)
mlp_heads_bwd_kernel(int N, int NC, const float* __restrict__ dOut, const float* __restrict__ Wh,
                     const float* __restrict__ Y7, float* __restrict__ G7, float* __restrict__ partial_W,
                     float* __restrict__ partial_b) {
    __shared__ __attribute__((aligned(16))) float sW[16][MLP_W];
    __shared__ float sB[4][16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, chunk = blockIdx.x;
    const int r0 = chunk * HD_ROWS, r1 = min(N, r0 + HD_ROWS);
    hf2 wa[16], wb[16], aa[16], ab[16];
#pragma unroll
    for (int o = 0; o < 16; o++) {
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (o < NC) w = *reinterpret_cast<const float4*>(Wh + (size_t)o * MLP_W + lane * 4);
        wa[o] = (hf2){w.x, w.y}, wb[o] = (hf2){w.z, w.w};
        aa[o] = (hf2){0.f, 0.f}, ab[o] = (hf2){0.f, 0.f};
    }
    float accb = 0.f;
    for (int rb = r0 + wv * 8; rb < r1; rb += 32) {
        float4 y[8];
        float dv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int r = rb + u;
            y[u] = make_float4(0.f, 0.f, 0.f, 0.f), dv[u] = 0.f;
            if (r < r1) {
                y[u] = *reinterpret_cast<const float4*>(Y7 + (size_t)r * MLP_W + lane * 4);
                if (lane < NC) dv[u] = dOut[(size_t)r * NC + lane];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (rb + u < r1) {  // wave-uniform
                const hf2 ya = {y[u].x, y[u].y}, yb = {y[u].z, y[u].w};
                hf2 sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
                for (int o = 0; o < 16; o++) {  // (columns o >= NC: dv and Wh are zero)
                    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dv[u]), o));
                    sa += d * wa[o], sb += d * wb[o];
                    aa[o] += d * ya, ab[o] += d * yb;
                }
                float4 g;
                g.x = y[u].x > 0.f ? sa.x : 0.f;
                g.y = y[u].y > 0.f ? sa.y : 0.f;
                g.z = y[u].z > 0.f ? sb.x : 0.f;
                g.w = y[u].w > 0.f ? sb.y : 0.f;
                *reinterpret_cast<float4*>(G7 + (size_t)(rb + u) * MLP_W + lane * 4) = g;
                accb += dv[u];
            }
        }
    }
    if (lane < 16) sB[wv][lane] = accb;
    for (int w = 0; w < 4; w++) {  // fixed order: wave 0, 1, 2, 3
        if (wv == w) {
#pragma unroll
            for (int o = 0; o < 16; o++) {
                float4* p = reinterpret_cast<float4*>(&sW[o][lane * 4]);
                float4 v = make_float4(aa[o].x, aa[o].y, ab[o].x, ab[o].y);
                if (w > 0) {
                    const float4 q = *p;
                    v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
                }
                *p = v;
            }
        }
        __syncthreads();
    }
    const int c = threadIdx.x;
#pragma unroll
    for (int o = 0; o < 16; o++)
        if (o < NC) partial_W[((size_t)chunk * 16 + o) * MLP_W + c] = sW[o][c];
    if (c < 16) partial_b[(size_t)chunk * 16 + c] = ((sB[0][c] + sB[1][c]) + sB[2][c]) + sB[3][c];
}

// dWh[o][c] = sum_chunks partial_W[chunk][o][c], dbh[o] = sum_chunks partial_b[chunk][o].  Grid (8 column blocks, n_out):
// a workgroup owns 32 columns of one output row as 8 float4 positions x 32 chunk groups (a wave instruction reads 128 contiguous
// bytes of eight chunks), group sums meet in LDS in a fixed order; the first column block of each row also reduces its bias.
__device__ __forceinline__ void reduce_heads_body(int bx, int o, int chunks, int bchunks, const float* __restrict__ partial_W,
                                                  const float* __restrict__ partial_b, float* __restrict__ dWh,
                                                  float* __restrict__ dbh) {
    __shared__ float4 red[32][8];
    __shared__ float redb[256];
    const int tid = threadIdx.x, pos = tid & 7, grp = tid >> 3;
    const int c0 = bx * 32 + pos * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* src = partial_W + (size_t)o * MLP_W + c0;
    int k = grp;
    for (; k + 96 < chunks; k += 128) {  // four loads in flight
        const float4 v0 = *reinterpret_cast<const float4*>(src + (size_t)k * 16 * MLP_W);
        const float4 v1 = *reinterpret_cast<const float4*>(src + (size_t)(k + 32) * 16 * MLP_W);
        const float4 v2 = *reinterpret_cast<const float4*>(src + (size_t)(k + 64) * 16 * MLP_W);
        const float4 v3 = *reinterpret_cast<const float4*>(src + (size_t)(k + 96) * 16 * MLP_W);
        s.x += (v0.x + v1.x) + (v2.x + v3.x), s.y += (v0.y + v1.y) + (v2.y + v3.y);
        s.z += (v0.z + v1.z) + (v2.z + v3.z), s.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; k < chunks; k += 32) {
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)k * 16 * MLP_W);
        s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    red[grp][pos] = s;
    float sb = 0.f;
    if (bx == 0)
        for (int c = tid; c < bchunks; c += 256) sb += partial_b[(size_t)c * 16 + o];
    redb[tid] = sb;
    __syncthreads();
    if (tid < 8) {
        float4 t = red[0][tid];
        for (int g = 1; g < 32; g++) t.x += red[g][tid].x, t.y += red[g][tid].y, t.z += red[g][tid].z, t.w += red[g][tid].w;
        *reinterpret_cast<float4*>(dWh + (size_t)o * MLP_W + bx * 32 + tid * 4) = t;
    }
    if (bx == 0 && tid == 8) {
        float t = 0.f;
        for (int i = 0; i < 256; i++) t += redb[i];
        dbh[o] = t;
    }
}
__global__ void __launch_bounds__(256)
mlp_reduce_heads_kernel(int chunks, int NC, const float* __restrict__ partial_W, const float* __restrict__ partial_b,
                        float* __restrict__ dWh, float* __restrict__ dbh) {
    reduce_heads_body(blockIdx.x, blockIdx.y, chunks, chunks, partial_W, partial_b, dWh, dbh);
    (void)NC;
}

// broadcast t: dtemb[c] = sum_j db0[j] W0[j][63+c] + db5[j] W5[j][63+c]   (one 256-thread block per column c)
__global__ void __launch_bounds__(256)
mlp_dtemb_bcast_kernel(int T, const float* __restrict__ db0, const float* __restrict__ W0, int in0,
                       const float* __restrict__ db5, const float* __restrict__ W5, int in5, float* __restrict__ dtemb) {
    __shared__ float red[4];
    const int c = blockIdx.x, j = threadIdx.x;
    float v = db0[j] * W0[(size_t)j * in0 + MLP_XE + c] + db5[j] * W5[(size_t)j * in5 + MLP_XE + c];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((j & 63) == 0) red[j >> 6] = v;
    __syncthreads();
    if (j == 0) dtemb[c] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- time branch for ONE row: t_emb = timenet(PE(t)) (R/utils/time_utils.py:52-55, 178-204 with is_blender) ------------
// t is identical for every Gaussian of a training iteration (R/train.py:158), so the branch is a 13 -> 256 -> 30
// network on a single row: one workgroup, instead of ~45 one-element PyTorch kernels forward and ~50 backward.
// save = [pe (2 n_freq + 1) | h (hidden)]
__global__ void __launch_bounds__(256)
mlp_timenet_fwd_kernel(const float* __restrict__ t, int n_freq, const float* __restrict__ W1, const float* __restrict__ b1,
                       int hidden, const float* __restrict__ W2, const float* __restrict__ b2, int n_out,
                       float* __restrict__ save, float* __restrict__ out) {
    __shared__ float pe[64];
    __shared__ float h[1024];
    const int n_pe = 2 * n_freq + 1;
    const int tid = threadIdx.x;
    if (tid < n_pe) {
        const float tv = t[0];
        float v = tv;
        if (tid > 0) {
            const int q = (tid - 1) >> 1;
            const float arg = tv * (float)(1 << q);  // freq_bands = 2^linspace(0, n_freq - 1, n_freq)
            v = ((tid - 1) & 1) ? cosf(arg) : sinf(arg);
        }
        pe[tid] = v;
        save[tid] = v;
    }
    __syncthreads();
    for (int j = tid; j < hidden; j += 256) {
        float a = b1[j];
        for (int i = 0; i < n_pe; i++) a += W1[j * n_pe + i] * pe[i];
        a = fmaxf(a, 0.f);
        h[j] = a;
        save[n_pe + j] = a;
    }
    __syncthreads();
    // eight lanes per output, 32 outputs per round (one round for the reference's 30): every thread streams its slice of the
    // row with all loads in flight, three shuffles finish the sum -- the kernel is pure latency, so rounds are what costs
    for (int o0 = 0; o0 < n_out; o0 += 32) {
        const int o = o0 + (tid >> 3), l8 = tid & 7;
        float a = 0.f;
        if (o < n_out) {
#pragma unroll 8
            for (int j = l8; j < hidden; j += 8) a += W2[o * hidden + j] * h[j];
        }
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        a += __shfl_xor(a, 4, 64);
        if (l8 == 0 && o < n_out) out[o] = a + b2[o];
    }
}

__global__ void __launch_bounds__(256)
mlp_timenet_bwd_kernel(const float* __restrict__ d_out, int n_freq, const float* __restrict__ W2, int hidden, int n_out,
                       const float* __restrict__ save, float* __restrict__ dW1, float* __restrict__ db1,
                       float* __restrict__ dW2, float* __restrict__ db2) {
    __shared__ float dO[64];
    const int n_pe = 2 * n_freq + 1;
    const int tid = threadIdx.x;
    if (tid < n_out) {
        dO[tid] = d_out[tid];
        db2[tid] = d_out[tid];
    }
    __syncthreads();
    const float* pe = save;
    const float* h = save + n_pe;
    for (int j = tid; j < hidden; j += 256) {
        const float hj = h[j];
        float dh = 0.f;
        for (int o = 0; o < n_out; o++) {
            dh += dO[o] * W2[o * hidden + j];
            dW2[o * hidden + j] = dO[o] * hj;
        }
        dh = hj > 0.f ? dh : 0.f;
        db1[j] = dh;
        for (int i = 0; i < n_pe; i++) dW1[j * n_pe + i] = dh * pe[i];
    }
}

}  // namespace dgm

// =================================================================================================================
// C ABI
// =================================================================================================================
using namespace dgm;

namespace dgm {
void set_last_error(const char* msg);  // c_api.hip
void prof_begin(int stage, hipStream_t st);
void prof_end(int stage, hipStream_t st);
}
namespace {
int mlp_fail(const char* msg) {
    dgm::set_last_error(msg);
    return 1;
}
struct Ws {
    float *emb, *Y[8], *Wt[8], *Wd[8], *Ga, *Gb, *partial, *partial_db, *partial_h, *partial_hb;
    float *partial_l[8], *partial_db_l[8];  // per layer: the weight-gradient partial tiles wait for the pass's one reduction
    unsigned* mask[8];
    // plane path (mlp_planes.hpp): weight planes and their inverse scales (wsc_e: the embedding half of the skip layer), tile
    // exponents of the embedding / Y_l / dOut / the two gradient buffers, the lane-native fp32 half of the skip layer, the dOut
    // planes, per-matrix maxima, the heads' planes and inverse scales
    uint4 *Wt3[8], *Wd3[8];
    float *wsc_f[8], *wsc_d[8], *wsc_e;
    int *Eexp, *Yexp[8], *Dexp, *Gexp[2];
    float4* Cin;
    unsigned char* Dp;
    unsigned* matmax;
    uint4 *Wh4f, *Wh4b;
    float *wsc_hf, *wsc_hb;
    float *bias_s[8];           // round 6: biases times their layer's column scales (mlp_prep4c_kernel)
    float *beff[2], *temb_row;  // round 6: the call's time row, kept for the backward pass (beff: the folded biases of the first version, unused -- they are formed where the biases are pre-scaled)
    size_t bytes;
};
int num_cus() {
    static int cache[DGM_MAX_DEVICES] = {0};  // per device: one process may drive several GPUs
    const int slot = current_device_slot();
    if (cache[slot] == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            n <= 0)
            n = 256;
        cache[slot] = n;
    }
    return cache[slot];
}
// 3: f16x3p (default; mlp_planes.hpp): power-of-two scaled operands split into 2 binary16, 3 partial products on the f16 matrix
//    cores, on plane-format activations -- split once by the producer, one exponent per 32-row tile.
// 4: the same arithmetic on round 3-5's kernel forms ("f16x3p8": eight waves x 32 columns per layer GEMM, the skip layer's
//    embedding half as an fp32 `Cin` stream written by layer 0) -- what mode 3 still runs for a time input PER ROW; kept
//    selectable as the A/B partner of round 6's forms (one wave per SIMD, K = 320 skip layer, time row folded into the biases).
// 5: mode 3 with EVERY 256-wide layer GEMM in the one-wave-per-SIMD form ("f16x3p5": mlp_gemm5_kernel<0, 0> forward, <0, 1> in
//    unpaired backward passes).  Measured slower than the eight-wave form (50.5 vs 48.9 us under the power probe, 54.5
//    vs 48.4 inside a network pass), so mode 3 uses that form for the K = 320 skip layer only; kept selectable for the record.
// 1: native fp32 MFMA (v_mfma_f32_32x32x2_f32).  Initial value from DGM_MLP_GEMM=f16x3p|f16x3p8|f16x3p5|f32.
// (0 = "bf16x6 for every GEMM" was retired in round 4, 2 = "f16x3" on fp32 rows in round 5: both are ignored by dgm_mlp_set_gemm.)
int g_gemm_mode = [] {
    const char* e = getenv("DGM_MLP_GEMM");
    if (e == nullptr) return 3;
    if (strcmp(e, "f32") == 0) return 1;
    if (strcmp(e, "f16x3p8") == 0) return 4;
    if (strcmp(e, "f16x3p5") == 0) return 5;
    if (strcmp(e, "f16x3p") != 0) fprintf(stderr, "libdgmesh_hip: unknown DGM_MLP_GEMM=\"%s\" (f16x3p | f16x3p8 | f32): using f16x3p\n", e);
    return 3;
}();
// arithmetic the last forward pass on a workspace ran in: the backward pass must match (its scales and masks were produced by
// that forward pass).  A small host-side table keyed by the workspace pointer; an unknown workspace is not checked.
struct WsMode {
    const void* ws;
    int mode;
};
WsMode g_ws_mode[64] = {};
unsigned g_ws_next = 0;
void remember_ws_mode(const void* ws, int mode) {
    for (auto& e : g_ws_mode)
        if (e.ws == ws) {
            e.mode = mode;
            return;
        }
    g_ws_mode[g_ws_next++ % 64] = WsMode{ws, mode};
}
int recall_ws_mode(const void* ws) {
    for (auto& e : g_ws_mode)
        if (e.ws == ws) return e.mode;
    return -1;
}
// plane path: a weight-gradient chunk is a run of whole 32-row tiles, one chunk per CU (or fewer)
struct P4Plan {
    int ntiles, tiles_per_chunk, chunks;
};
P4Plan p4_plan(int N) {
    P4Plan d;
    d.ntiles = (N + 31) / 32;
    d.tiles_per_chunk = (d.ntiles + num_cus() - 1) / num_cus();
    if (d.tiles_per_chunk < 1) d.tiles_per_chunk = 1;
    d.chunks = (d.ntiles + d.tiles_per_chunk - 1) / d.tiles_per_chunk;
    return d;
}
Ws carve(char* base, int N) {
    Ws w;
    char* p = align_ptr(base);
    auto take = [&](size_t b) {
        char* at = p;
        p = align_ptr(p + b);
        return (float*)at;
    };
    const size_t n = ((size_t)N + 31) / 32 * 32;  // every per-row tensor is padded to whole 32-row tiles (plane path)
    const size_t ntiles = n / 32;
    const size_t chunks = ((size_t)N + DW_ROWS - 1) / DW_ROWS;   // fp32-MFMA path: 512-row chunks
    const P4Plan pp = p4_plan(N > 0 ? N : 1);                     // plane path: chunks of whole tiles, at most one per CU
    w.emb = take(n * MLP_EMB * 4);
    for (int l = 0; l < 8; l++) w.Y[l] = take(n * MLP_W * 4);
    for (int l = 0; l < 8; l++) w.mask[l] = (unsigned*)take(n * 8 * 4);
    for (int l = 0; l < 8; l++) w.Wt[l] = take((size_t)(MLP_EMB + MLP_W) * MLP_W * 4);
    for (int l = 0; l < 8; l++) w.Wd[l] = take((size_t)MLP_W * MLP_W * 4);
    w.Ga = take(n * MLP_W * 4);
    w.Gb = take(n * MLP_W * 4);
    for (int l = 0; l < 8; l++) {  // (geometry is fixed: layer 0 consumes the embedding, layer 5 embedding | trunk)
        const int Kp = l == 0 ? MLP_EMB : (l == 5 ? MLP_EMB + MLP_W : MLP_W);
        size_t rows = (size_t)pp.chunks * Kp, dbr = (size_t)pp.chunks * 8;  // (plane path: 8 bias-gradient rows per chunk)
        if (l == 5) {  // the fp32-MFMA path reduces layer by layer through this one: room for its widest layer
            if (chunks * (MLP_EMB + MLP_W) > rows) rows = chunks * (MLP_EMB + MLP_W);
            if (chunks > dbr) dbr = chunks;
        }
        w.partial_l[l] = take(rows * MLP_W * 4);
        w.partial_db_l[l] = take(dbr * MLP_W * 4);
    }
    w.partial = w.partial_l[5];
    w.partial_db = w.partial_db_l[5];
    for (int l = 0; l < 8; l++) w.Wt3[l] = (uint4*)take((size_t)(MLP_EMB + MLP_W) * MLP_W * 4);
    for (int l = 0; l < 8; l++) w.Wd3[l] = (uint4*)take((size_t)MLP_W * MLP_W * 4);
    for (int l = 0; l < 8; l++) w.wsc_f[l] = take(1024);  // (one inverse scale per output column in round 6's kernel forms, [0] = the matrix's otherwise)
    for (int l = 0; l < 8; l++) w.wsc_d[l] = take(1024);
    w.wsc_e = take(256);
    size_t hchunks = ((size_t)N + HD_ROWS - 1) / HD_ROWS;
    if ((size_t)pp.chunks > hchunks) hchunks = pp.chunks;
    w.partial_h = take(hchunks * 16 * MLP_W * 4);
    w.partial_hb = take((hchunks > ntiles ? hchunks : ntiles) * 16 * 4);
    w.Eexp = (int*)take(ntiles * 4);
    for (int l = 0; l < 8; l++) w.Yexp[l] = (int*)take(ntiles * 4);
    w.Dexp = (int*)take(ntiles * 4);
    w.Gexp[0] = (int*)take(ntiles * 4);
    w.Gexp[1] = (int*)take(ntiles * 4);
    w.Cin = (float4*)take(n * MLP_W * 4);
    w.Dp = (unsigned char*)take(n * 128);
    w.matmax = (unsigned*)take(P4_MAX_MATS * 8 * 4);
    w.Wh4f = (uint4*)take((size_t)MLP_W * 32 * 4);
    w.Wh4b = (uint4*)take((size_t)16 * MLP_W * 4);
    w.wsc_hf = take(1024);
    w.wsc_hb = take(1024);
    w.beff[0] = take(MLP_W * 4);
    w.beff[1] = take(MLP_W * 4);
    for (int l = 0; l < 8; l++) w.bias_s[l] = take(MLP_W * 4);
    w.temb_row = take(64 * 4);
    w.bytes = (size_t)(p - base) + 256;
    return w;
}
int layer_in(const dgm_mlp_params* p, int l) {
    if (l == 0) return p->emb_dim;
    if (l == p->skip_layer) return p->emb_dim + MLP_W;
    return MLP_W;
}
int layer_kp(const dgm_mlp_params* p, int l) {
    if (l == 0) return MLP_EMB;
    if (l == p->skip_layer) return MLP_EMB + MLP_W;
    return MLP_W;
}
int check_params(const dgm_mlp_params* p) {
    if (!p) return mlp_fail("mlp: params is NULL");
    if (p->n_layers != 8 || p->width != MLP_W) return mlp_fail("mlp: only D=8, W=256 is supported (the reference's only configuration)");
    if (p->emb_dim != MLP_XE + p->t_dim || p->emb_dim > MLP_EMB) return mlp_fail("mlp: emb_dim must be 63 + t_dim <= 96");
    if (p->skip_layer != 5) return mlp_fail("mlp: skip re-injection must feed layer 5 (skips=[4])");
    if (p->n_out < 1 || p->n_out > 16) return mlp_fail("mlp: 1..16 head outputs supported");
    for (int l = 0; l < 8; l++)
        if (!p->W[l] || !p->b[l]) return mlp_fail("mlp: NULL weight pointer");
    if (!p->Wh || !p->bh) return mlp_fail("mlp: NULL head pointer");
    return 0;
}

// ---- plane path (mlp_planes.hpp) -----------------------------------------------------------------------------------------
template <typename K>
hipError_t p4_lds_attr(K kernel, int bytes, bool* done) {  // dynamic LDS above 48 KB needs the attribute once per device
    if (*done) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) *done = true;
    return e;
}
#define P4_LAUNCH(kernel_, cfg_lds_, grid_, st_, args_)                                                                \
    {                                                                                                                  \
        static bool done_[DGM_MAX_DEVICES] = {false};                                                                  \
        if (p4_lds_attr(kernel_, cfg_lds_, &done_[current_device_slot()]) != hipSuccess)                               \
            return mlp_fail("mlp: cannot raise the LDS limit of a plane-path kernel");                                 \
        hipLaunchKernelGGL(kernel_, dim3(grid_), dim3(512), cfg_lds_, st_, args_);                                     \
    }
#define P5_LAUNCH(kernel_, cfg_lds_, grid_, st_, args_)                                                                \
    {                                                                                                                  \
        static bool done_[DGM_MAX_DEVICES] = {false};                                                                  \
        if (p4_lds_attr(kernel_, cfg_lds_, &done_[current_device_slot()]) != hipSuccess)                               \
            return mlp_fail("mlp: cannot raise the LDS limit of a plane-path kernel");                                 \
        hipLaunchKernelGGL(kernel_, dim3(grid_), dim3(256), cfg_lds_, st_, args_);                                     \
    }
typedef Gemm4Cfg<16, 1024, 512, 0, false, 8> CfgFwd;
typedef Gemm4Cfg<16, 1024, 512, 2, false, 8> CfgSkip;
typedef Gemm4Cfg<16, 1024, 512, 1, false, 8> CfgBwd;
typedef Gemm4Cfg<6, 384, 192, 0, true, 8> CfgL0;
typedef Gemm4Cfg<16, 1024, 512, 3, false, 1> CfgHeads;
typedef Gemm4Cfg<1, 128, 64, 1, false, 8> CfgG7;
typedef Gemm4Cfg<4, 256, 128, 0, false, 8, true> CfgL0F;   // layer 0 on the 64-column embedding
typedef Gemm4Cfg<16, 1024, 512, 0, false, 8, true> CfgFwdC;   // round 6's forms: one weight scale per output column
typedef Gemm4Cfg<16, 1024, 512, 1, false, 8, true> CfgBwdC;
typedef Gemm4Cfg<16, 1024, 512, 3, false, 1, true> CfgHeadsC;
typedef Gemm4Cfg<1, 128, 64, 1, false, 8, true> CfgG7C;
typedef Gemm5Cfg<0, 0> Cfg5Fwd;
typedef Gemm5Cfg<0, 1> Cfg5Bwd;
typedef Gemm5Cfg<4, 0> Cfg5Skip;
typedef Gemm5Cfg<0, 2> Cfg5FwdH;
typedef Dw4Cfg<2, 8, 256, 128, 1024, 512> CfgDwE64;
typedef Dw4Cfg<8, 8, 1024, 512, 1024, 512> CfgDw;
typedef Dw4Cfg<3, 8, 384, 192, 1024, 512> CfgDwE;
typedef Dw4Cfg<8, 1, 1024, 512, 128, 64> CfgDwH;
#ifndef P4_PAIR_SHARE_NUM
#define P4_PAIR_SHARE_NUM 128  // share of the CUs on the weight gradient in a paired backward launch: NUM / DEN (measured at N = 100 k: 96 -> 1.50, 112 -> 1.43, 128 -> 1.41, 144 -> 1.43, off -> 1.56 ms per network pass)
#define P4_PAIR_SHARE_DEN 256
#endif

// Tiles per workgroup whose exponents the layer GEMM keeps in its LDS table (512: N <= 4 M rows on 256 CUs; the rest are read from
// HBM as they come).  DGM_P4_EXPS_LIMIT=<n> lowers it so that a test reaches the second path at an ordinary batch size.
int p4_exps_limit() {
    static int v = [] {
        const char* e = getenv("DGM_P4_EXPS_LIMIT");
        const int n = e ? atoi(e) : 512;
        return n < 1 ? 1 : (n > 512 ? 512 : n);
    }();
    return v;
}

// workgroups of a backward launch that run the weight gradient (mlp_bwd_pair_kernel); 0 = two separate launches.
// DGM_MLP_PAIR=<n> overrides (0 disables); default: a fixed share of the CUs, chosen by measurement at N = 100 k.
int p4_pair_split(int ntiles, int gx) {
    static int env = [] {
        const char* e = getenv("DGM_MLP_PAIR");
        return e ? atoi(e) : -1;
    }();
    if (gx < 64 || ntiles < 4 * gx) return 0;  // small problems: nothing to balance
    int n = env >= 0 ? env : (gx * P4_PAIR_SHARE_NUM) / P4_PAIR_SHARE_DEN;
    if (n >= gx) n = gx - 1;
    return n;
}

// fold: one time value per call (temb_stride == 0) on round 6's kernel forms -- the embedding is 64 columns wide, the time row sits
// in the biases of layer 0 and the skip layer, the skip layer is one K = 320 GEMM (no `Cin`), the 256-wide layers run one wave per SIMD.
int forward_planes(const dgm_mlp_params* p, int N, const float* x, const float* temb, int temb_stride, const Ws& w, float* out,
                   hipStream_t st, const bool fold, const bool all5) {
    const P4Plan pl = p4_plan(N);
    const int nt = pl.ntiles, gx = nt < num_cus() ? nt : num_cus();
    const int sk = p->skip_layer;
    const int EW = fold ? 64 : MLP_EMB;
    // the heads inside layer 7's launch (mlp_gemm5_kernel<0, 2>) instead of a launch of their own that reads Y_7 again: 64 us against
    // 50 + 26, +0.95 % on the bench (308.8 / 309.4 / 309.3 against 306.0 / 306.1 / 306.5 it/s, interleaved on one box);
    // DGM_MLP_HEADS_FUSED=0 restores the separate launch (A/B runs)
    static const bool heads_env = [] { const char* e = getenv("DGM_MLP_HEADS_FUSED"); return !(e && atoi(e) == 0); }();
    const bool heads_fused = fold && heads_env && p->n_out <= 16;
    // embedding planes + the maxima of the nine weight tensors (W_0 .. W_7, Wh)
    AbsMaxBatch am;
    am.n_jobs = fold ? 0 : 9;  // (round 6's forms scale per output column: mlp_prep4c_kernel takes its columns' maxima itself)
    for (int l = 0; l < 8; l++) am.job[l].W = p->W[l], am.job[l].n = MLP_W * layer_in(p, l);
    am.job[8].W = p->Wh, am.job[8].n = p->n_out * MLP_W;
    if (!fold)
        hipLaunchKernelGGL(mlp_embed4_kernel, dim3(nt + 8 * am.n_jobs), dim3(256), 0, st, N, nt, x, temb, temb_stride, p->t_dim,
                           (unsigned char*)w.emb, w.Eexp, am, w.matmax, EW);
    if (fold) {  // every weight matrix as two binary16 planes, one power-of-two scale per OUTPUT COLUMN, biases pre-scaled
        Prep4cBatch pc;
        int nc = 0;
        pc.temb = temb, pc.temb_row = w.temb_row, pc.T = p->t_dim;
        auto addc = [&](int mode, int Kp, int ncols, int in_features, int hoff, int k_valid, int col_valid, const float* Wp, uint4* Bp,
                        float* inv_scale, const float* bias_in, float* bias_out, int fold_time = 0) {
            Prep4cJob& q = pc.job[nc++];
            q.fold = fold_time;
            q.j.mode = mode, q.j.Kp = Kp, q.j.ncols = ncols, q.j.in_features = in_features, q.j.emb_dim = p->emb_dim, q.j.hoff = hoff;
            q.j.k_valid = k_valid, q.j.col_valid = col_valid, q.j.W = Wp, q.j.Bp = Bp, q.j.inv_scale = inv_scale;
            q.bias_in = bias_in, q.bias_out = bias_out;
        };
        for (int l = 0; l < 8; l++) {
            const int Kp = l == 0 ? 64 : (l == sk ? 320 : MLP_W);
            addc(0, Kp, MLP_W, layer_in(p, l), 0, MLP_W, MLP_W, p->W[l], w.Wt3[l], w.wsc_f[l], p->b[l], w.bias_s[l], (l == 0 || l == sk) ? 1 : 0);
            if (l >= 1) addc(1, MLP_W, MLP_W, layer_in(p, l), l == sk ? p->emb_dim : 0, MLP_W, MLP_W, p->W[l], w.Wd3[l], w.wsc_d[l], nullptr, nullptr);
        }
        addc(0, MLP_W, 32, MLP_W, 0, MLP_W, p->n_out, p->Wh, w.Wh4f, w.wsc_hf, nullptr, nullptr);   // heads forward: B[k][o] = Wh[o][k]
        addc(1, 16, MLP_W, MLP_W, 0, p->n_out, MLP_W, p->Wh, w.Wh4b, w.wsc_hb, nullptr, nullptr);    // heads backward: B[o][c] = Wh[o][c]
        // ... in ONE launch with the embedding planes (DGM_MLP_EMBED_MERGED=0: two launches, for A/B runs)
        static const bool merged = [] { const char* e = getenv("DGM_MLP_EMBED_MERGED"); return !(e && atoi(e) == 0); }();
        if (merged) {
            const int n_prep = (MLP_W / P4C_COLS) * nc;
            hipLaunchKernelGGL(mlp_embed4c_kernel, dim3(n_prep + nt), dim3(256), 0, st, N, x, (unsigned char*)w.emb, w.Eexp, EW, pc, n_prep);
        } else {
            hipLaunchKernelGGL(mlp_embed4_kernel, dim3(nt), dim3(256), 0, st, N, nt, x, temb, temb_stride, p->t_dim,
                               (unsigned char*)w.emb, w.Eexp, am, w.matmax, EW);
            hipLaunchKernelGGL(mlp_prep4c_kernel, dim3(MLP_W / P4C_COLS, nc), dim3(256), 0, st, pc);
        }
    }
    // (rounds 3-5's forms: one power-of-two scale per matrix)
    Prep4Batch pb;
    int nj = 0;
    auto add = [&](int mode, int Kp, int ncols, int in_features, int hoff, int k_valid, int col_valid, const float* Wp, uint4* Bp,
                   float* inv_scale, int mat) {
        Prep4Job& q = pb.job[nj++];
        q.j.mode = mode, q.j.Kp = Kp, q.j.ncols = ncols, q.j.in_features = in_features, q.j.emb_dim = p->emb_dim, q.j.hoff = hoff;
        q.j.k_valid = k_valid, q.j.col_valid = col_valid, q.j.W = Wp, q.j.Bp = Bp, q.j.inv_scale = inv_scale, q.mat = mat;
    };
    for (int l = 0; l < 8; l++) {
        if (fold && l == 0) add(0, 64, MLP_W, layer_in(p, l), 0, MLP_W, MLP_W, p->W[l], w.Wt3[l], w.wsc_f[l], l);
        else if (fold && l == sk) add(0, 320, MLP_W, layer_in(p, l), 0, MLP_W, MLP_W, p->W[l], w.Wt3[l], w.wsc_f[l], l);
        else if (l == sk) {  // K = 352 = embedding half (rides along with layer 0) | trunk half
            add(0, MLP_EMB, MLP_W, layer_in(p, l), 0, MLP_W, MLP_W, p->W[l], w.Wt3[l], w.wsc_e, l);
            add(0, MLP_W, MLP_W, layer_in(p, l), p->emb_dim, MLP_W, MLP_W, p->W[l], w.Wt3[l] + (size_t)MLP_EMB * MLP_W / 4, w.wsc_f[l], l);
        } else
            add(0, layer_kp(p, l), MLP_W, layer_in(p, l), 0, MLP_W, MLP_W, p->W[l], w.Wt3[l], w.wsc_f[l], l);
        if (l >= 1) add(1, MLP_W, MLP_W, layer_in(p, l), l == sk ? p->emb_dim : 0, MLP_W, MLP_W, p->W[l], w.Wd3[l], w.wsc_d[l], l);
    }
    add(0, MLP_W, 32, MLP_W, 0, MLP_W, p->n_out, p->Wh, w.Wh4f, w.wsc_hf, 8);   // heads forward: B[k][o] = Wh[o][k]
    add(1, 16, MLP_W, MLP_W, 0, p->n_out, MLP_W, p->Wh, w.Wh4b, w.wsc_hb, 8);    // heads backward: B[o][c] = Wh[o][c]
    if (!fold) hipLaunchKernelGGL(mlp_prep4_kernel, dim3(8, nj), dim3(256), 0, st, pb, w.matmax);

    Gemm4Args a;
    memset(&a, 0, sizeof(a));
    a.ntiles = nt, a.M = N, a.exps_limit = p4_exps_limit();
    if (fold) {
        // layer 0: K = 64, eight waves (HBM-bound: 26 MB in, 102 MB of planes out)
        a.A = (const unsigned char*)w.emb, a.Aexp = w.Eexp, a.Bp = w.Wt3[0], a.b_inv = w.wsc_f[0], a.bias = w.bias_s[0];
        a.mask_out = w.mask[0], a.C = (unsigned char*)w.Y[0], a.Cexp = w.Yexp[0];
        P4_LAUNCH((mlp_gemm4_kernel<4, 256, 128, 0, false, 8, true>), CfgL0F::LDS, gx, st, a)
        Gemm5Args b;
        memset(&b, 0, sizeof(b));
        b.ntiles = nt, b.M = N, b.exps_limit = p4_exps_limit();
        for (int l = 1; l < 8; l++) {
            b.A = (const unsigned char*)w.Y[l - 1], b.Aexp = w.Yexp[l - 1], b.Bp = w.Wt3[l], b.b_inv = w.wsc_f[l];
            b.mask_out = w.mask[l], b.C = (unsigned char*)w.Y[l], b.Cexp = w.Yexp[l];
            if (l == sk) {  // K = 320: the embedding's four K steps, then the trunk's sixteen
                b.A2 = (const unsigned char*)w.emb, b.A2exp = w.Eexp, b.bias = w.bias_s[sk];
                P5_LAUNCH((mlp_gemm5_kernel<4, 0>), Cfg5Skip::LDS, gx, st, b)
                b.A2 = nullptr, b.A2exp = nullptr;
            } else if (l == 7 && heads_fused) {  // the last trunk layer with the heads riding along (one wave per SIMD: the form with the registers for it)
                b.bias = w.bias_s[l];
                b.hBp = w.Wh4f, b.h_inv = w.wsc_hf, b.h_bias = p->bh, b.h_out = out, b.h_nout = p->n_out;
                P5_LAUNCH((mlp_gemm5_kernel<0, 2>), Cfg5FwdH::LDS, gx, st, b)
            } else if (all5) {
                b.bias = w.bias_s[l];
                dgm::prof_begin(DGM_STAGE_MLP_LAYER_FWD, st);
                P5_LAUNCH((mlp_gemm5_kernel<0, 0>), Cfg5Fwd::LDS, gx, st, b)
                dgm::prof_end(DGM_STAGE_MLP_LAYER_FWD, st);
            } else {  // the eight-wave form (measured faster for K = 256)
                a.A = b.A, a.Aexp = b.Aexp, a.Bp = b.Bp, a.b_inv = b.b_inv, a.bias = w.bias_s[l], a.mask_out = b.mask_out, a.C = b.C, a.Cexp = b.Cexp;
                dgm::prof_begin(DGM_STAGE_MLP_LAYER_FWD, st);
                P4_LAUNCH((mlp_gemm4_kernel<16, 1024, 512, 0, false, 8, true>), CfgFwdC::LDS, gx, st, a)
                dgm::prof_end(DGM_STAGE_MLP_LAYER_FWD, st);
            }
        }
    } else {
    // layer 0 (K = 96) with the embedding half of the skip layer as second output
    a.A = (const unsigned char*)w.emb, a.Aexp = w.Eexp, a.Bp = w.Wt3[0], a.b_inv = w.wsc_f[0], a.bias = p->b[0];
    a.mask_out = w.mask[0], a.C = (unsigned char*)w.Y[0], a.Cexp = w.Yexp[0];
    a.Bp2 = w.Wt3[sk], a.b_inv2 = w.wsc_e, a.bias2 = p->b[sk], a.out2 = w.Cin;
    P4_LAUNCH((mlp_gemm4_kernel<6, 384, 192, 0, true, 8>), CfgL0::LDS, gx, st, a)
    a.Bp2 = nullptr, a.b_inv2 = nullptr, a.bias2 = nullptr, a.out2 = nullptr;
    for (int l = 1; l < 8; l++) {
        a.A = (const unsigned char*)w.Y[l - 1], a.Aexp = w.Yexp[l - 1], a.b_inv = w.wsc_f[l], a.bias = p->b[l];
        a.mask_out = w.mask[l], a.C = (unsigned char*)w.Y[l], a.Cexp = w.Yexp[l];
        if (l == sk) {
            a.Bp = w.Wt3[l] + (size_t)MLP_EMB * MLP_W / 4, a.cin = w.Cin;
            P4_LAUNCH((mlp_gemm4_kernel<16, 1024, 512, 2, false, 8>), CfgSkip::LDS, gx, st, a)
            a.cin = nullptr;
        } else {
            a.Bp = w.Wt3[l];
            dgm::prof_begin(DGM_STAGE_MLP_LAYER_FWD, st);
            P4_LAUNCH((mlp_gemm4_kernel<16, 1024, 512, 0, false, 8>), CfgFwd::LDS, gx, st, a)
            dgm::prof_end(DGM_STAGE_MLP_LAYER_FWD, st);
        }
    }
    }
    a.Bp2 = nullptr, a.b_inv2 = nullptr, a.bias2 = nullptr, a.out2 = nullptr, a.cin = nullptr;
    // heads: out = Y7 Wh^T + bh (one computing wave per workgroup; HBM-bound: one pass over Y7)
    a.A = (const unsigned char*)w.Y[7], a.Aexp = w.Yexp[7], a.Bp = w.Wh4f, a.b_inv = w.wsc_hf, a.bias = p->bh;
    a.mask_out = nullptr, a.C = nullptr, a.Cexp = nullptr, a.out = out, a.ldo = p->n_out, a.n_valid = p->n_out;
    if (fold && heads_fused) {}  // (written by layer 7's launch)
    else if (fold) P4_LAUNCH((mlp_gemm4_kernel<16, 1024, 512, 3, false, 1, true>), CfgHeadsC::LDS, gx, st, a)
    else P4_LAUNCH((mlp_gemm4_kernel<16, 1024, 512, 3, false, 1>), CfgHeads::LDS, gx, st, a)
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mlp_fail(hipGetErrorString(e));
    return 0;
}

int backward_planes(const dgm_mlp_params* p, int N, const float* dOut, int temb_stride, const Ws& w, float* const* dW, float* const* db,
                    float* dWh, float* dbh, float* dtemb, hipStream_t st, const bool fold, const bool all5, const float* x_in = nullptr,
                    float* dX = nullptr) {
    const P4Plan pl = p4_plan(N);
    const int nt = pl.ntiles, gx = nt < num_cus() ? nt : num_cus();
    const int sk = p->skip_layer;
    const int EW = fold ? 64 : MLP_EMB;  // rows of the embedding's block of the K = EW / EW + 256 weight gradients (fold: see forward_planes)
    // dOut -> planes (zero columns / rows beyond n_out / N) + the heads' bias-gradient partial sums
    hipLaunchKernelGGL(mlp_dout4_kernel, dim3((nt + 3) / 4), dim3(256), 0, st, N, nt, p->n_out, dOut, w.Dp, w.Dexp, w.partial_hb);
    unsigned char* G = (unsigned char*)w.Ga;
    unsigned char* Gn = (unsigned char*)w.Gb;
    int *Ge = w.Gexp[0], *Gne = w.Gexp[1];
    Gemm4Args a;
    memset(&a, 0, sizeof(a));
    a.ntiles = nt, a.M = N, a.exps_limit = p4_exps_limit();
    // G_7 = (dOut Wh) masked by layer 7's ReLU
    a.A = w.Dp, a.Aexp = w.Dexp, a.Bp = w.Wh4b, a.b_inv = w.wsc_hb, a.mask_in = w.mask[7], a.C = G, a.Cexp = Ge;
    Dw4Args d;
    memset(&d, 0, sizeof(d));
    d.ntiles = nt, d.tiles_per_chunk = pl.tiles_per_chunk;
    // heads' weight gradient: dWh[o][c] = sum_r dOut[r][o] Y7[r][c]
    d.X = (const unsigned char*)w.Y[7], d.Xexp = w.Yexp[7], d.G = w.Dp, d.Gexp = w.Dexp, d.partial = w.partial_h, d.chunk_stride = 0;
    d.partial_db = nullptr;
    // (tried, round 6: this launch on a side stream beside the G_7 launch -- a read stream over Y_7 beside a write stream, joined again in
    // front of the reduction; G_7 with one or two workgroups per CU: 308.8 / 309.0 and 306.3 / 308.2 it/s against 311.7 / 311.8 in order)
    {   // one K step per tile: nothing to hide the epilogue's barriers under, so two workgroups per CU (104 VGPRs, 54 KB of LDS)
        static const int mult = [] { const char* e = getenv("DGM_P4_G7_MULT"); return e ? atoi(e) : 2; }();
        const int g7 = nt < mult * num_cus() ? nt : mult * num_cus();
        if (fold) P4_LAUNCH((mlp_gemm4_kernel<1, 128, 64, 1, false, 8, true>), CfgG7C::LDS, g7, st, a)
        else P4_LAUNCH((mlp_gemm4_kernel<1, 128, 64, 1, false, 8>), CfgG7::LDS, g7, st, a)
    }
    P4_LAUNCH((mlp_dw4_kernel<8, 1, 1024, 512, 128, 64>), CfgDwH::LDS, pl.chunks, st, d)
    ReduceDwBatch rb;
    rb.emb_dim = p->emb_dim, rb.n_jobs = 0;
    int rb_blocks = 0, n_pending = 0;
    auto add_job = [&](int slot, int chunks, int db_rows, int Kp, int in_features, int dst_off, const float* partial,
                       const float* partial_db, float* dWp, float* dbp) {
        ReduceDwJob& jb = rb.job[slot];
        jb.chunks = chunks, jb.db_rows = db_rows, jb.Kp = Kp, jb.in_features = in_features, jb.nblocks = reduce_dw_blocks(Kp);
        jb.emb_rows = EW, jb.k_valid = fold ? MLP_XE : p->emb_dim;
        jb.dst_off = dst_off, jb.partial = partial, jb.partial_db = partial_db, jb.dW = dWp, jb.db = dbp;
        if (jb.nblocks > rb_blocks) rb_blocks = jb.nblocks;
    };
    // Layers 7 .. 1: backward data (G_l -> G_{l-1}) and the weight gradient of layer l both consume G_l and nothing of each
    // other: ONE launch, the first n_dw workgroups on the weight gradient (n_dw row chunks), the rest on the layer GEMM
    // (mlp_bwd_pair_kernel).  n_dw = 0 (DGM_MLP_PAIR=0) runs them as two launches over all CUs.  The embedding's rows of the
    // skip layer's gradient stay a launch of their own over all CUs (own partial tiles, own reduction job).
    int n_dw = p4_pair_split(nt, gx);
    if (n_dw > pl.chunks) n_dw = pl.chunks;  // (the skip layer's partial buffer is carved for pl.chunks tiles of 352 rows)
    const bool per_row_t = temb_stride != 0 && dtemb != nullptr;
    if (per_row_t && dX) return mlp_fail("mlp_backward: position gradients need a broadcast time row");
    for (int l = 7; l >= 0; l--) {
        const int Kp = l == 0 ? EW : (l == sk ? EW + MLP_W : MLP_W);
        const bool paired = n_dw > 0 && l >= 1;
        // (the paired launch gives each of its n_dw weight-gradient workgroups ceil(nt / n_dw) tiles: the last few may get none
        // and write no partial tile -- the reduction must only read the ones that exist)
        const int tpc_pair = paired ? (nt + n_dw - 1) / n_dw : 0;
        const int chunks_l = paired ? (nt + tpc_pair - 1) / tpc_pair : pl.chunks;
        if (l >= 1) {  // backward data: G_{l-1} = (G_l W_l) masked, into the other buffer
            a.A = G, a.Aexp = Ge, a.Bp = w.Wd3[l], a.b_inv = w.wsc_d[l], a.mask_in = w.mask[l - 1], a.C = Gn, a.Cexp = Gne;
        }
        d.G = G, d.Gexp = Ge;
        if (dX && (l == sk || l == 0))  // gradient w.r.t. the positions: G_5 W_5[:, :63] into scratch (the forward's Cin, idle now), then + G_0 W_0[:, :63] and PE'
            hipLaunchKernelGGL(mlp_dx4_kernel, dim3(nt), dim3(256), 0, st, N, (const unsigned char*)G, (const int*)Ge, p->W[l],
                               layer_in(p, l), 0, MLP_XE, reinterpret_cast<float*>(w.Cin), l == 0 ? 1 : 0, x_in, dX, 0);
        if (per_row_t && (l == sk || l == 0))  // a time input per row: dL/dt_emb[r] = G_5[r] W_5[:, 63:63+T] + G_0[r] W_0[:, 63:63+T]
            hipLaunchKernelGGL(mlp_dx4_kernel, dim3(nt), dim3(256), 0, st, N, (const unsigned char*)G, (const int*)Ge, p->W[l],
                               layer_in(p, l), MLP_XE, p->t_dim, reinterpret_cast<float*>(w.Cin), l == 0 ? 2 : 0, (const float*)nullptr,
                               dtemb, p->t_dim);
        // partial tiles of the layer: [chunk][Kp][256]; paired skip layer: trunk rows [n_dw][256][256], then embedding rows [chunks][96][256]
        float* part_emb = w.partial_l[l];
        float* part_trunk = w.partial_l[l] + (l == sk ? (size_t)EW * MLP_W : 0);
        size_t stride_emb = (size_t)Kp * MLP_W, stride_trunk = (size_t)Kp * MLP_W;
        if (paired && l == sk) {
            part_trunk = w.partial_l[l], stride_trunk = (size_t)MLP_W * MLP_W;
            part_emb = w.partial_l[l] + (size_t)n_dw * MLP_W * MLP_W, stride_emb = (size_t)EW * MLP_W;
        }
        if (l == 0 || l == sk) {  // the embedding's rows of the gradient (rows 0 .. 95 of the K = 96 / 352 partial tile)
            d.tiles_per_chunk = pl.tiles_per_chunk;
            d.X = (const unsigned char*)w.emb, d.Xexp = w.Eexp, d.partial = part_emb, d.chunk_stride = stride_emb, d.partial_db = w.partial_db_l[l];
            if (fold) P4_LAUNCH((mlp_dw4_kernel<2, 8, 256, 128, 1024, 512>), CfgDwE64::LDS, pl.chunks, st, d)
            else P4_LAUNCH((mlp_dw4_kernel<3, 8, 384, 192, 1024, 512>), CfgDwE::LDS, pl.chunks, st, d)
        }
        if (l != 0) {
            d.tiles_per_chunk = paired ? tpc_pair : pl.tiles_per_chunk;
            d.X = (const unsigned char*)w.Y[l - 1], d.Xexp = w.Yexp[l - 1];
            d.partial = part_trunk, d.chunk_stride = stride_trunk;
            d.partial_db = l == sk ? nullptr : w.partial_db_l[l];  // (the skip layer's bias gradient came with its embedding half)
        }
        if (paired) {
            dgm::prof_begin(DGM_STAGE_MLP_BWD_PAIR, st);
            {
                static bool done_[DGM_MAX_DEVICES] = {false}, donec_[DGM_MAX_DEVICES] = {false};
                constexpr int LDS_PAIR = CfgBwdC::LDS > CfgDw::LDS ? CfgBwdC::LDS : CfgDw::LDS;
                if ((fold ? p4_lds_attr(mlp_bwd_pair_kernel<true>, LDS_PAIR, &donec_[current_device_slot()])
                          : p4_lds_attr(mlp_bwd_pair_kernel<false>, LDS_PAIR, &done_[current_device_slot()])) != hipSuccess)
                    return mlp_fail("mlp: cannot raise the LDS limit of mlp_bwd_pair_kernel");
                const int grid = n_dw + (gx - n_dw > 0 ? gx - n_dw : 1);
                // chunked: the GEMM role walks the weight-gradient role's row chunks (needs as many GEMM workgroups as chunks)
                static const int chunk_env = [] { const char* e = getenv("DGM_MLP_PAIR_CHUNKED"); return e ? atoi(e) : 1; }();
                const int chunked = (chunk_env && grid - n_dw == n_dw && (n_dw % 8) == 0) ? 1 : 0;
                if (fold) hipLaunchKernelGGL(mlp_bwd_pair_kernel<true>, dim3(grid), dim3(512), LDS_PAIR, st, a, d, n_dw, chunked);
                else hipLaunchKernelGGL(mlp_bwd_pair_kernel<false>, dim3(grid), dim3(512), LDS_PAIR, st, a, d, n_dw, chunked);
            }
            dgm::prof_end(DGM_STAGE_MLP_BWD_PAIR, st);
        } else if (l >= 1) {
            dgm::prof_begin(DGM_STAGE_MLP_LAYER_BWD, st);
            if (all5) {
                Gemm5Args b;
                memset(&b, 0, sizeof(b));
                b.ntiles = nt, b.M = N, b.exps_limit = a.exps_limit;
                b.A = a.A, b.Aexp = a.Aexp, b.Bp = a.Bp, b.b_inv = a.b_inv, b.mask_in = a.mask_in, b.C = a.C, b.Cexp = a.Cexp;
                P5_LAUNCH((mlp_gemm5_kernel<0, 1>), Cfg5Bwd::LDS, gx, st, b)
            } else if (fold)
                P4_LAUNCH((mlp_gemm4_kernel<16, 1024, 512, 1, false, 8, true>), CfgBwdC::LDS, gx, st, a)
            else
                P4_LAUNCH((mlp_gemm4_kernel<16, 1024, 512, 1, false, 8>), CfgBwd::LDS, gx, st, a)
            dgm::prof_end(DGM_STAGE_MLP_LAYER_BWD, st);
            dgm::prof_begin(DGM_STAGE_MLP_LAYER_DW, st);
            P4_LAUNCH((mlp_dw4_kernel<8, 8, 1024, 512, 1024, 512>), CfgDw::LDS, chunks_l, st, d)
            dgm::prof_end(DGM_STAGE_MLP_LAYER_DW, st);
        }
        if (paired && l == sk) {  // two reduction jobs: embedding rows (with the bias gradient), trunk rows
            add_job(n_pending++, pl.chunks, 8 * pl.chunks, EW, layer_in(p, l), 0, part_emb, w.partial_db_l[l], dW[l], db[l]);
            add_job(n_pending++, chunks_l, 0, MLP_W, layer_in(p, l), p->emb_dim, part_trunk, nullptr, dW[l], nullptr);
        } else
            add_job(n_pending++, chunks_l, 8 * chunks_l, Kp, layer_in(p, l), 0, w.partial_l[l], w.partial_db_l[l], dW[l], db[l]);
        if (l >= 1) {
            unsigned char* t = G;
            G = Gn, Gn = t;
            int* te = Ge;
            Ge = Gne, Gne = te;
        }
    }
    // (tried: the previous layer's partial tiles summed by the GEMM workgroups of the next paired launch, before their weight
    // prologue: the launch grows by 9 us (87 -> 96), six launches, while this reduction only shrinks 87 -> 52 us: a net loss)
    // (tried, round 5: each layer's tiles summed by a launch of its own right behind the layer's paired launch, while they are still
    // in the 256 MB MALL -- nine launches of 23 us against this one of 89: a single layer's 256 workgroups do not fill the chip)
    // (tried, round 6: TWO launches, layers 7 .. 4 reduced right behind layer 4's paired launch -- 130-160 MB of partial tiles each, a
    // thousand workgroups each, so that the reads could hit the MALL: 304.3 / 304.1 it/s against 309.1 / 308.3 with the single launch,
    // interleaved on one box, scripts/gpu_r6_rsplit.sh; the split at layer 3 or 5: 305.0 / 303.6)
    // (tried, round 6: this launch and what follows it on a side stream -- a second entry point running the pass in two pieces, the
    // trainer joining the streams before it looks at a gradient -- so that the first network's reduction runs beside the rasterizer's
    // VALU-bound backward: 305.5 / 305.6 / 305.4 it/s against 309.8 / 309.0 / 309.6 on one stream, and every consumer of a gradient
    // becomes a place to race; an unsafe timing run with NO join at all, the second network's reduction under Adam and the next
    // forward pass, showed +2 % -- that overlap is not available to a correct step)
    // (tried: each layer's reduction on a side stream under the next layer's launch -- its small workgroups do fit beside a
    // resident pair workgroup -- 265 vs 274 it/s: slower, the pair kernel's HBM-bound half gets the competition)
    rb.n_jobs = n_pending;
    rb.h_chunks = pl.chunks, rb.h_bchunks = nt, rb.h_nout = p->n_out, rb.h_partial_W = w.partial_h, rb.h_partial_b = w.partial_hb;
    rb.h_dW = dWh, rb.h_db = dbh;
    hipLaunchKernelGGL(mlp_reduce_dw_all_kernel, dim3(rb_blocks, rb.n_jobs + 1), dim3(256), 0, st, rb);
    if (fold) {  // the time columns of dW_0 / dW_skip: db (x) t_emb; and dL/dt_emb (one launch)
        FoldGradArgs f;
        f.dW[0] = dW[0], f.db[0] = db[0], f.W[0] = p->W[0], f.in_features[0] = layer_in(p, 0);
        f.dW[1] = dW[sk], f.db[1] = db[sk], f.W[1] = p->W[sk], f.in_features[1] = layer_in(p, sk);
        f.temb = w.temb_row, f.T = p->t_dim, f.dtemb = (dtemb != nullptr && !per_row_t) ? dtemb : nullptr;
        hipLaunchKernelGGL(mlp_fold_grad_kernel, dim3(f.dtemb != nullptr ? 3 : 2, p->t_dim), dim3(256), 0, st, f);
    } else if (dtemb != nullptr && !per_row_t)
        hipLaunchKernelGGL(mlp_dtemb_bcast_kernel, dim3(p->t_dim), dim3(256), 0, st, p->t_dim, db[0], p->W[0], layer_in(p, 0),
                           db[sk], p->W[sk], layer_in(p, sk), dtemb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mlp_fail(hipGetErrorString(e));
    return 0;
}
}  // namespace

extern "C" {

int dgm_mlp_set_gemm(int mode) {
    const int prev = g_gemm_mode;
    if (mode == 1 || mode == 3 || mode == 4 || mode == 5) g_gemm_mode = mode;  // (0, "bf16x6 for every GEMM", and 2, "f16x3" on fp32 rows, are retired: ignored)
    return prev;
}

size_t dgm_mlp_workspace_bytes(int N) {
    Ws w = carve(nullptr, N > 0 ? N : 0);
    return w.bytes;
}

#ifdef P4_TIMING
int dgm_p4_timing(unsigned long long* out) {  // (variant builds only: phase timers of the layer GEMM, see mlp_planes.hpp)
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dgm::g_p4_timing), sizeof(unsigned long long) * 256 * 8 * 16);
}
#endif

#ifdef P4_PROBE
// (variant builds only, tools/power_probe.py: ONE plane-path kernel in a loop on synthetic operands, for clock / power telemetry)
//   kind 0: forward layer GEMM (mlp_gemm4_kernel<16,1024,512,0>)   1: weight gradient alone (mlp_dw4_kernel<8,8>)
//   kind 2: the paired backward launch (mlp_bwd_pair_kernel, the production split)        3: backward-data GEMM alone
// zero != 0: all operands zero (same instruction stream, no data toggling).  The buffers are this probe's own.
namespace {
__global__ void p4_probe_fill_kernel(unsigned* __restrict__ p, size_t n, unsigned seed, int zero) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15, x *= 0x2c1b3c6du, x ^= x >> 12;
        // two binary16 values of magnitude < 2: sign | exponent <= 15 | mantissa
        p[i] = zero ? 0u : (x & 0xbfffbfffu);
    }
}
}  // namespace
int dgm_p4_probe(int N, int kind, int iters, int zero, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const P4Plan pl = p4_plan(N);
    const int nt = pl.ntiles, gx = nt < num_cus() ? nt : num_cus();
    static unsigned char *A = nullptr, *C = nullptr, *G = nullptr;
    static int* ex = nullptr;
    static uint4* Bp = nullptr;
    static float *binv = nullptr, *bias = nullptr, *partial = nullptr, *pdb = nullptr;
    static unsigned* mask = nullptr;
    static int cap = 0;
    const size_t rows = (size_t)nt * 32;
    if (cap < nt) {
        if (hipMalloc((void**)&A, rows * 1024) != hipSuccess || hipMalloc((void**)&C, rows * 1024) != hipSuccess ||
            hipMalloc((void**)&G, rows * 1024) != hipSuccess || hipMalloc((void**)&ex, (size_t)nt * 4 * 4) != hipSuccess ||
            hipMalloc((void**)&Bp, 16 * 4 * 256 * 16) != hipSuccess || hipMalloc((void**)&binv, 1024) != hipSuccess ||
            hipMalloc((void**)&bias, 1024) != hipSuccess || hipMalloc((void**)&mask, (size_t)nt * 1024) != hipSuccess ||
            hipMalloc((void**)&partial, (size_t)256 * 256 * 256 * 4) != hipSuccess || hipMalloc((void**)&pdb, (size_t)256 * 8 * 256 * 4) != hipSuccess)
            return mlp_fail("p4_probe: hipMalloc failed");
        cap = nt;
    }
    hipLaunchKernelGGL(p4_probe_fill_kernel, dim3(1024), dim3(256), 0, st, (unsigned*)A, rows * 256, 1u, zero);
    hipLaunchKernelGGL(p4_probe_fill_kernel, dim3(1024), dim3(256), 0, st, (unsigned*)G, rows * 256, 2u, zero);
    hipLaunchKernelGGL(p4_probe_fill_kernel, dim3(64), dim3(256), 0, st, (unsigned*)Bp, (size_t)16 * 4 * 256 * 4, 3u, zero);
    hipLaunchKernelGGL(p4_probe_fill_kernel, dim3(256), dim3(256), 0, st, mask, (size_t)nt * 256, 4u, 0);
    (void)hipMemsetAsync(ex, 0, (size_t)nt * 4 * 4, st);
    (void)hipMemsetAsync(bias, 0, 1024, st);
    const float one = 1.0f / 65536.0f;
    {
        float ones[256];
        for (int i = 0; i < 256; i++) ones[i] = one;
        (void)hipMemcpyAsync(binv, ones, sizeof(ones), hipMemcpyHostToDevice, st);
        (void)hipStreamSynchronize(st);
    }
    Gemm4Args a;
    memset(&a, 0, sizeof(a));
    a.exps_limit = 512;
    a.ntiles = nt, a.M = N, a.A = kind == 0 ? A : G, a.Aexp = ex, a.Bp = Bp, a.b_inv = binv, a.bias = bias, a.mask_in = mask,
    a.mask_out = mask + (size_t)0, a.C = C, a.Cexp = ex + nt;
    Dw4Args d;
    memset(&d, 0, sizeof(d));
    d.ntiles = nt, d.tiles_per_chunk = pl.tiles_per_chunk, d.X = A, d.Xexp = ex + 2 * nt, d.G = G, d.Gexp = ex + 3 * nt, d.partial = partial,
    d.chunk_stride = (size_t)256 * 256, d.partial_db = pdb;
    int n_dw = p4_pair_split(nt, gx);
    if (n_dw > pl.chunks) n_dw = pl.chunks;
    Gemm5Args a5;
    memset(&a5, 0, sizeof(a5));
    a5.exps_limit = 512, a5.ntiles = nt, a5.M = N, a5.A = (kind == 4) ? A : G, a5.Aexp = ex, a5.Bp = Bp, a5.b_inv = binv, a5.bias = bias;
    a5.mask_in = mask, a5.mask_out = mask, a5.C = C, a5.Cexp = ex + nt;
    for (int it = 0; it < iters; it++) {
        if (kind == 4) {  // round 6: the one-wave-per-SIMD forms
            P5_LAUNCH((mlp_gemm5_kernel<0, 0>), Cfg5Fwd::LDS, gx, st, a5)
        } else if (kind == 5) {
            P5_LAUNCH((mlp_gemm5_kernel<0, 1>), Cfg5Bwd::LDS, gx, st, a5)
        } else if (kind == 0) {
            P4_LAUNCH((mlp_gemm4_kernel<16, 1024, 512, 0, false, 8>), CfgFwd::LDS, gx, st, a)
        } else if (kind == 3) {
            P4_LAUNCH((mlp_gemm4_kernel<16, 1024, 512, 1, false, 8>), CfgBwd::LDS, gx, st, a)
        } else if (kind == 1) {
            P4_LAUNCH((mlp_dw4_kernel<8, 8, 1024, 512, 1024, 512>), CfgDw::LDS, pl.chunks, st, d)
        } else {
            if (n_dw <= 0) return mlp_fail("p4_probe: no pair split at this size");
            static bool done_[DGM_MAX_DEVICES] = {false};
            constexpr int LDS_PAIR = CfgBwd::LDS > CfgDw::LDS ? CfgBwd::LDS : CfgDw::LDS;
            if (p4_lds_attr(mlp_bwd_pair_kernel<false>, LDS_PAIR, &done_[current_device_slot()]) != hipSuccess) return mlp_fail("p4_probe: LDS attribute");
            Dw4Args dp = d;
            dp.tiles_per_chunk = (nt + n_dw - 1) / n_dw;
#ifdef P4_XCD_REDUCE
            {
                static unsigned* xs = nullptr;
                static float* xp = nullptr;
                static int gen = 0;
                if (!xs) {
                    if (hipMalloc((void**)&xs, 8 * 128) != hipSuccess || hipMalloc((void**)&xp, (size_t)8 * 65536 * 4) != hipSuccess) return mlp_fail("p4_probe: hipMalloc");
                    (void)hipMemsetAsync(xs, 0, 8 * 128, st);
                }
                dp.xsync = getenv("DGM_P4_XCD_OFF") ? nullptr : xs, dp.xpartial = xp, dp.xgen = gen++, dp.xchunks = (nt + dp.tiles_per_chunk - 1) / dp.tiles_per_chunk;
            }
#endif
            const int grid = n_dw + (gx - n_dw > 0 ? gx - n_dw : 1);
            const int chunked = (grid - n_dw == n_dw && (n_dw % 8) == 0) ? 1 : 0;
            hipLaunchKernelGGL(mlp_bwd_pair_kernel<false>, dim3(grid), dim3(512), LDS_PAIR, st, a, dp, n_dw, chunked);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mlp_fail(hipGetErrorString(e));
    return 0;
}
#endif

int dgm_mlp_describe_workspace(int N, size_t* offs, int capacity) {
    Ws w = carve(nullptr, N > 0 ? N : 0);
    const char* base = nullptr;
    const void* f[DGM_MLP_WS_FIELDS];
    int n = 0;
    f[n++] = w.emb;
    for (int l = 0; l < 8; l++) f[n++] = w.Y[l];
    for (int l = 0; l < 8; l++) f[n++] = w.mask[l];
    f[n++] = w.Ga, f[n++] = w.Gb;
    f[n++] = w.Eexp;
    for (int l = 0; l < 8; l++) f[n++] = w.Yexp[l];
    f[n++] = w.Dexp, f[n++] = w.Gexp[0], f[n++] = w.Gexp[1];
    f[n++] = w.Cin, f[n++] = w.Dp;
    for (int l = 0; l < 8; l++) f[n++] = w.partial_l[l];
    for (int l = 0; l < 8; l++) f[n++] = w.partial_db_l[l];
    for (int i = 0; i < n && i < capacity; i++) offs[i] = (size_t)((const char*)f[i] - base);
    return n;
}

int dgm_mlp_forward(const dgm_mlp_params* p, int N, const float* x, const float* temb, int temb_stride, char* workspace,
                    float* out, void* stream) {
    if (check_params(p)) return 1;
    if (N <= 0) return 0;
    if (!x || !temb || !workspace || !out) return mlp_fail("mlp_forward: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    Ws w = carve(workspace, N);
    const int mode = g_gemm_mode;
    const bool fold = (mode == 3 || mode == 5) && temb_stride == 0;
    remember_ws_mode(workspace, mode + (fold ? 16 : 0));
    if (mode >= 3) return forward_planes(p, N, x, temb, temb_stride, w, out, st, fold, fold && mode == 5);
    // ---- native fp32 MFMA
    for (int l = 0; l < 8; l++) {
        const int Kp = layer_kp(p, l);
        const int n = (Kp > MLP_W ? Kp : MLP_W) * MLP_W;
        hipLaunchKernelGGL(mlp_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, st, layer_in(p, l), Kp, p->emb_dim,
                           l == p->skip_layer ? 1 : 0, p->W[l], w.Wt[l], l >= 1 ? w.Wd[l] : nullptr);
    }
    hipLaunchKernelGGL(mlp_embed_kernel, dim3((N + 7) / 8), dim3(256), 0, st, N, x, temb, temb_stride, p->t_dim, w.emb);
    const int grid = (N + GM - 1) / GM;
    for (int l = 0; l < 8; l++) {
        const float *A1, *A2 = nullptr;
        int lda1, K1, lda2 = 0, K2 = 0;
        if (l == 0) {
            A1 = w.emb, lda1 = MLP_EMB, K1 = MLP_EMB;
        } else if (l == p->skip_layer) {
            A1 = w.emb, lda1 = MLP_EMB, K1 = MLP_EMB, A2 = w.Y[l - 1], lda2 = MLP_W, K2 = MLP_W;
        } else {
            A1 = w.Y[l - 1], lda1 = MLP_W, K1 = MLP_W;
        }
        hipLaunchKernelGGL(mlp_gemm_kernel<0>, dim3(grid, MLP_W / GN), dim3(256), 0, st, N, A1, lda1, K1, A2, lda2, K2, w.Wt[l],
                           p->b[l], w.mask[l], w.Y[l]);
    }
    hipLaunchKernelGGL(mlp_rows_small_kernel, dim3((N * 4 + 255) / 256), dim3(256), 0, st, N, p->n_out, w.Y[7], p->Wh, 1,
                       MLP_W, p->bh, out, p->n_out, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mlp_fail(hipGetErrorString(e));
    return 0;
}

int dgm_mlp_backward(const dgm_mlp_params* p, int N, const float* dOut, int temb_stride, char* workspace, float* const* dW,
                     float* const* db, float* dWh, float* dbh, float* dtemb, void* stream) {
    return dgm_mlp_backward_dx(p, N, dOut, temb_stride, workspace, dW, db, dWh, dbh, dtemb, nullptr, nullptr, stream);
}

int dgm_mlp_backward_dx(const dgm_mlp_params* p, int N, const float* dOut, int temb_stride, char* workspace, float* const* dW,
                        float* const* db, float* dWh, float* dbh, float* dtemb, const float* x, float* dX, void* stream) {
    if (check_params(p)) return 1;
    if (N <= 0) return 0;
    if (!dOut || !workspace || !dW || !db || !dWh || !dbh) return mlp_fail("mlp_backward: NULL pointer");
    if ((x == nullptr) != (dX == nullptr)) return mlp_fail("mlp_backward: x and dX come together");
    const int mode = g_gemm_mode;
    if (dX && (mode < 3 || temb_stride != 0))
        return mlp_fail("mlp_backward: the gradient w.r.t. the positions exists in the plane arithmetic only (f16x3p, broadcast time row)");
    {
        const int fm = recall_ws_mode(workspace);
        if (fm >= 0 && (fm & 15) != mode)
            return mlp_fail("mlp_backward: the arithmetic mode changed since the forward pass on this workspace (dgm_mlp_set_gemm)");
        if (fm >= 0 && ((fm & 16) != 0) != ((mode == 3 || mode == 5) && temb_stride == 0))
            return mlp_fail("mlp_backward: temb_stride differs from the forward pass on this workspace");
    }
    hipStream_t st = (hipStream_t)stream;
    Ws w = carve(workspace, N);
    if (mode >= 3) {
        const bool fold = (mode == 3 || mode == 5) && temb_stride == 0;
        return backward_planes(p, N, dOut, temb_stride, w, dW, db, dWh, dbh, dtemb, st, fold, fold && mode == 5, x, dX);
    }
    // ---- native fp32 MFMA
    const int chunks = (N + DW_ROWS - 1) / DW_ROWS;
    const int grid = (N + GM - 1) / GM;
    // heads: G7 (masked by layer 7's ReLU, read off Y7 itself) and the partial sums of dWh / dbh in one pass over Y7
    const int hchunks = (N + HD_ROWS - 1) / HD_ROWS;
    hipLaunchKernelGGL(mlp_heads_bwd_kernel, dim3(hchunks), dim3(256), 0, st, N, p->n_out, dOut, p->Wh, w.Y[7], w.Ga,
                       w.partial_h, w.partial_hb);
    hipLaunchKernelGGL(mlp_reduce_heads_kernel, dim3(8, p->n_out), dim3(256), 0, st, hchunks, p->n_out, w.partial_h,
                       w.partial_hb, dWh, dbh);
    float* G = w.Ga;
    float* Gn = w.Gb;
    const bool per_row_t = temb_stride != 0 && dtemb != nullptr;
    for (int l = 7; l >= 0; l--) {
        const float *X1, *X2 = nullptr;
        int ldx1, K1, ldx2 = 0, K2 = 0;
        if (l == 0) {
            X1 = w.emb, ldx1 = MLP_EMB, K1 = MLP_EMB;
        } else if (l == p->skip_layer) {
            X1 = w.emb, ldx1 = MLP_EMB, K1 = MLP_EMB, X2 = w.Y[l - 1], ldx2 = MLP_W, K2 = MLP_W;
        } else {
            X1 = w.Y[l - 1], ldx1 = MLP_W, K1 = MLP_W;
        }
        // backward data first: G_{l-1} = (G_l W_l) masked, into the other buffer
        if (l >= 1)
            hipLaunchKernelGGL(mlp_gemm_kernel<1>, dim3(grid, MLP_W / GN), dim3(256), 0, st, N, G, MLP_W, MLP_W,
                               (const float*)nullptr, 0, 0, w.Wd[l], (const float*)nullptr, w.mask[l - 1], Gn);
        const int Kp = K1 + K2;
        const int slabs = (Kp + DW_SLAB - 1) / DW_SLAB;
        hipLaunchKernelGGL(mlp_dw_kernel, dim3(slabs, chunks), dim3(512), 0, st, N, X1, ldx1, K1, X2, ldx2, K2, G, w.partial,
                           w.partial_db);
        hipLaunchKernelGGL(mlp_reduce_dw_kernel, dim3((Kp * MLP_W + 255) / 256), dim3(256), 0, st, chunks, chunks, Kp,
                           layer_in(p, l), p->emb_dim, w.partial, w.partial_db, dW[l], db[l]);
        if (per_row_t && (l == p->skip_layer || l == 0))  // dL/dt_emb[r] += G_l[r] . W_l[:, 63:63+T]
            for (int c0 = 0; c0 < p->t_dim; c0 += 16)     // the small kernel handles 16 output columns per pass
                hipLaunchKernelGGL(mlp_rows_small_kernel, dim3((N * 4 + 255) / 256), dim3(256), 0, st, N,
                                   p->t_dim - c0 < 16 ? p->t_dim - c0 : 16, G, p->W[l] + MLP_XE + c0, layer_in(p, l), 1,
                                   (const float*)nullptr, dtemb + c0, p->t_dim, l == 0 ? 1 : 0);
        if (l >= 1) {  // (the backward-data GEMM of this layer was launched above: G_l stays intact in its buffer)
            float* t = G;
            G = Gn;
            Gn = t;
        }
    }
    if (!per_row_t && dtemb != nullptr)
        hipLaunchKernelGGL(mlp_dtemb_bcast_kernel, dim3(p->t_dim), dim3(256), 0, st, p->t_dim, db[0], p->W[0], layer_in(p, 0),
                           db[p->skip_layer], p->W[p->skip_layer], layer_in(p, p->skip_layer), dtemb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mlp_fail(hipGetErrorString(e));
    return 0;
}

int dgm_timenet_forward(const float* t, int n_freq, const float* W1, const float* b1, int hidden, const float* W2,
                        const float* b2, int n_out, float* save, float* out, void* stream) {
    if (!t || !W1 || !b1 || !W2 || !b2 || !save || !out) return mlp_fail("timenet_forward: NULL pointer");
    if (n_freq < 0 || 2 * n_freq + 1 > 64 || hidden < 1 || hidden > 1024 || n_out < 1 || n_out > 64)
        return mlp_fail("timenet_forward: unsupported size (2 n_freq + 1 <= 64, hidden <= 1024, n_out <= 64)");
    hipLaunchKernelGGL(mlp_timenet_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, t, n_freq, W1, b1, hidden, W2, b2,
                       n_out, save, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mlp_fail(hipGetErrorString(e));
    return 0;
}

int dgm_timenet_backward(const float* d_out, int n_freq, const float* W2, int hidden, int n_out, const float* save,
                         float* dW1, float* db1, float* dW2, float* db2, void* stream) {
    if (!d_out || !W2 || !save || !dW1 || !db1 || !dW2 || !db2) return mlp_fail("timenet_backward: NULL pointer");
    if (n_freq < 0 || 2 * n_freq + 1 > 64 || hidden < 1 || hidden > 1024 || n_out < 1 || n_out > 64)
        return mlp_fail("timenet_backward: unsupported size");
    hipLaunchKernelGGL(mlp_timenet_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d_out, n_freq, W2, hidden, n_out,
                       save, dW1, db1, dW2, db2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mlp_fail(hipGetErrorString(e));
    return 0;
}

}  // extern "C"
