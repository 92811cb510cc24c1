// The 256-wide layer GEMM of the plane path with ONE wave per SIMD: 4 waves x 64 output columns (round 6).
//
// Same arithmetic, same operand / result formats and the same per-element summation order as gemm4_body (mlp_planes.hpp; results
// are bit-identical) -- what changes is who reads what.  At the socket's power cap a layer GEMM lasts E / P, and 19 %
// of its energy were the LDS fragment reads: eight waves each read the whole 32 x K tile of both planes to feed 32 output columns.
// Here a wave owns 64 columns (two accumulator blocks sharing every A fragment), so a tile's fragments are read four times instead
// of eight; a single wave per SIMD may use the whole 512-entry register file (weights: 2 blocks x KS x 2 planes x 4 = 256 registers
// at K = 256, 320 at K = 320), which is also what lets the skip layer run as ONE K = 320 GEMM:
//   KS2 = 4: four more K steps over the 64-column embedding planes [x, PE(x), 0] (a second DMA stream into its own LDS rows) run
//            FIRST; the accumulators are then rescaled by 2^(e_trunk - e_emb) (exact) and the 16 trunk steps follow.  The time
//            embedding is one row per call (R/train.py:158) and sits in the bias (mlp_prep4c_kernel's fold jobs) -- the 102 MB fp32 `Cin`
//            round trip of rounds 2-5 is gone.
// Two independent accumulator chains alternate MFMA by MFMA, so the matrix pipe needs no second wave to stay busy; everything
// else (E1 / E2 epilogue slices between the MFMAs, one vmcnt(0) per tile step at the mid barrier, three A buffers filled by
// global_load_lds, XOR-swizzled staging tile, 1 KiB row stores) follows gemm4_body's schedule with twice the per-wave counts.
#pragma once
#include "mlp_planes.hpp"

#ifndef P5_ABL
#define P5_ABL 0  // ablation builds (tools/build_variant.sh; results are wrong with any bit set): 1 no global stores, 2 no tile copies
                  // after the first two, 4 no MFMAs, 8 no epilogue slices, 16 no fragment reads
#endif
#ifndef P5_W_AGPR
#define P5_W_AGPR 12  // K steps whose weight fragments live in AGPRs (16 registers each); the accumulators take 32 more
#endif
#ifndef P5_BQ_AHEAD
#define P5_BQ_AHEAD 1  // the bias of an E1 slice fetched one slice ahead (0: with the slice's column scales, just in time)
#endif
#ifndef P5_W_AGPR_SKIP
#define P5_W_AGPR_SKIP 14  // the same for the K = 320 form (13: the one split that leaves no scratch access inside the steady-state
                           // loop -- a scratch load there waits for vmcnt(0), i.e. for the tile copies just issued)
#endif

namespace dgm {

struct Gemm5Args {
    int ntiles, M;
    const unsigned char* A;     // trunk input planes [Np][2][256] binary16
    const int* Aexp;            // [ntiles]
    const unsigned char* A2;    // KS2 > 0: embedding planes [Np][2][64]
    const int* A2exp;           // [ntiles]
    const uint4* Bp;            // weight planes, K steps in execution order: KS2 embedding steps, then the 16 trunk steps
    const float* b_inv;         // [256] inverse column scales (mlp_prep4c_kernel); bias is pre-scaled
    const float* bias;          // [256] (EPI 0)
    const unsigned* mask_in;    // EPI 1
    unsigned* mask_out;         // EPI 0
    unsigned char* C;           // output planes [Np][2][256]
    int* Cexp;                  // [ntiles]
    int exps_limit;             // tiles of a workgroup whose input exponents come from its LDS table; later ones from HBM
    // EPI 2 (the last trunk layer with the output heads riding along): out[row][o] = Y[row] . Wh[o] + bh[o], o < h_nout <= 16
    const uint4* hBp;           // heads' weight planes [16 K steps][2 halves | 2 planes][32 columns] (mlp_prep4c_kernel, ncols = 32)
    const float* h_inv;         // [h_nout] inverse column scales of hBp
    const float* h_bias;        // [h_nout]
    float* h_out;               // [M][h_nout] fp32
    int h_nout;
};

template <int KS2, int EPI>
struct Gemm5Cfg {
    static constexpr int KS = 16 + KS2;
    static constexpr int PITCH1 = 1024 + 16, A1BYTES = 32 * PITCH1;
    static constexpr int PITCH2 = 256 + 16, A2BYTES = KS2 > 0 ? 32 * PITCH2 : 0;
    static constexpr int ABYTES = A1BYTES + A2BYTES;
    static constexpr int NBUF = 3;                                   // A tiles in LDS: in use, landed, in flight
    static constexpr int A_END = NBUF * ABYTES;
    static constexpr int MI_BYTES = EPI == 1 ? NBUF * 1024 : 0;      // mask blocks of the same three tiles (EPI 1)
    static constexpr int O_BYTES = 32768;                            // staging tile of the output planes
    static constexpr int EXPS = KS2 > 0 ? 64 : 128;                  // input exponents of the workgroup's tiles (>= 0.5 M rows on 256 CUs)
    static constexpr int H_BYTES = EPI == 2 ? 256 + 4 * 32 * 16 * 4 : 0;  // EPI 2: heads' scales and biases, the four waves' partial outputs
    static constexpr int LDS = A_END + MI_BYTES + O_BYTES + 2 * 1024 + 64 + EXPS * 4 * (KS2 > 0 ? 2 : 1) + 1024 + 1024 + H_BYTES;  // (+ bias, + column scales)
    static_assert(LDS <= 160 * 1024, "LDS budget of a gfx950 CU");
};

// EPI 0: Y = relu(acc c + bias), planes + tile exponent + ReLU mask out     EPI 1: G' = mask ? acc c : 0 (backward data)
// EPI 2: EPI 0 plus the output heads on the tile just produced (the last trunk layer): while a tile's staged planes wait in LDS for
//        their row stores (the first half of the step after next), every wave multiplies ITS 64 columns of them with the heads'
//        weights (4 K steps, 12 MFMAs, the staged rows read as B fragments through the staging swizzle), the four partial 32 x 16
//        results meet in LDS at the mid barrier and leave as fp32 rows -- the heads launch and its 102 MB read of Y_7 are gone.
// bx / G / tiles_in as in gemm4_body (tiles bx, bx + G, ...; tiles_in >= 0: exactly that many).
template <int KS2, int EPI_>
__device__ __forceinline__ void gemm5_body(const Gemm5Args& a, const int bx, const int G, unsigned char* smem, const int tiles_in = -1) {
    using Cfg = Gemm5Cfg<KS2, EPI_>;
    constexpr int EPI = EPI_ == 2 ? 0 : EPI_;  // (everything below that says EPI == 0 holds for the heads-carrying variant too)
    constexpr bool HEADS = EPI_ == 2;
    constexpr int KS = Cfg::KS, P1 = Cfg::PITCH1, P2 = Cfg::PITCH2, ABYTES = Cfg::ABYTES, A1BYTES = Cfg::A1BYTES;
    static_assert(EPI_ == 0 || EPI_ == 1 || EPI_ == 2, "plane-producing variants only");
    static_assert(KS2 == 0 || (KS2 == 4 && EPI_ == 0), "the embedding steps belong to the forward skip layer");
    unsigned char* Abuf = smem;                                                   // [3][ A1: 32 x P1 | A2: 32 x P2 ]
    unsigned char* Mibuf = smem + Cfg::A_END;                                     // [3][1024] mask blocks in (EPI 1)
    unsigned char* Obuf = Mibuf + Cfg::MI_BYTES;                                  // [32][1024] staging of the output planes (swizzled)
    unsigned* mbuf = reinterpret_cast<unsigned*>(Obuf + Cfg::O_BYTES);            // [2][32][8] mask words out
    float* tmaxs = reinterpret_cast<float*>(mbuf + 512);                          // [4] wave maxima
    int* exps = reinterpret_cast<int*>(tmaxs + 16);                               // [EXPS]
    int* exps2 = exps + Cfg::EXPS;                                                // [EXPS] (KS2)
    float* biasl = reinterpret_cast<float*>(exps2 + (KS2 > 0 ? Cfg::EXPS : 0));   // [256]
    float* scl = biasl + 256;                                                     // [256] inverse column scales
    float* hinv = scl + 256;                                                      // HEADS: [32] inverse column scales of the heads' planes
    float* hbias = hinv + 32;                                                     //        [32] their biases
    float* hp = hbias + 32;                                                       //        [4 waves][32 rows][16] partial outputs
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, li = lane & 31;
    const int my_tiles = tiles_in >= 0 ? tiles_in : (a.ntiles - bx + G - 1) / G;
    if (my_tiles <= 0) return;
    const int limit = a.exps_limit < Cfg::EXPS ? a.exps_limit : Cfg::EXPS;

    // ---- copies: trunk rows one per instruction (64 lanes x 16 B = 1 KiB), embedding rows three per instruction (16 lanes each
    // + one idle lane = the 16-byte row pad).  A wave issues instructions i = 0..7 (trunk row 4 i + wv) and, with KS2, 8..10
    // (embedding instruction n = 4 (i - 8) + wv < 11).
    const unsigned abuf_lds = p4_lds_addr(Abuf), mibuf_lds = p4_lds_addr(Mibuf);
    // (the embedding copies' lane geometry -- row lane / 17, piece lane % 17 -- is recomputed at each use: three registers matter here)
    constexpr int NCP = KS2 > 0 ? 11 : 8;  // copy instructions per wave and tile
#define G5_COPY1(tile_, buf_, i_)                                                                                      \
    {                                                                                                                  \
        if ((i_) < 8) {                                                                                                \
            const int row_ = 4 * (i_) + wv;                                                                            \
            p4_glds16(a.A + ((size_t)(tile_) * 32 + row_) * 1024 + lane * 16,                                          \
                      __builtin_amdgcn_readfirstlane(abuf_lds + (buf_) * ABYTES + row_ * P1));                         \
        } else if (KS2 > 0) {                                                                                          \
            const int n_ = 4 * ((i_) - 8) + wv;                                                                        \
            if (n_ < 11) {                                                                                             \
                int l2_ = lane;                                                                                        \
                asm volatile("" : "+v"(l2_));  /* (opaque: keeps the compiler from hoisting this lane geometry out of the tile  \
                                                  loop into a register pair it then spills to scratch) */                      \
                const int c2_row = l2_ / 17, c2_piece = l2_ - 17 * c2_row;                                             \
                const bool c2_ok = c2_piece < 16 && c2_row < 3;                                                        \
                const int row_ = 3 * n_ + c2_row;                                                                      \
                if (c2_ok && row_ < 32)                                                                                \
                    p4_glds16(a.A2 + ((size_t)(tile_) * 32 + row_) * 256 + c2_piece * 16,                              \
                              __builtin_amdgcn_readfirstlane(abuf_lds + (buf_) * ABYTES + A1BYTES + n_ * 3 * P2));     \
            }                                                                                                          \
        }                                                                                                              \
    }
#define G5_COPY_MASK(tile_, buf_)                                                                                      \
    if (EPI == 1 && wv == 3)                                                                                           \
        p4_glds16(reinterpret_cast<const unsigned char*>(a.mask_in) + (size_t)(tile_) * 1024 + lane * 16,              \
                  __builtin_amdgcn_readfirstlane(mibuf_lds + (buf_) * 1024));

    // first the tiles (their latency is the longest), then the tables and the stationary weights
#pragma unroll
    for (int i = 0; i < NCP; i++) G5_COPY1(bx, 0, i)
    G5_COPY_MASK(bx, 0)
    if (my_tiles > 1) {
#pragma unroll
        for (int i = 0; i < NCP; i++) G5_COPY1(bx + G, 1, i)
        G5_COPY_MASK(bx + G, 1)
    }
    for (int t = tid; t < my_tiles && t < Cfg::EXPS; t += 256) {
        exps[t] = a.Aexp[bx + t * G];
        if (KS2 > 0) exps2[t] = a.A2exp[bx + t * G];
    }
    if (EPI == 0) biasl[tid] = a.bias[tid];
    scl[tid] = a.b_inv[tid];
    if (HEADS && tid < 32) hinv[tid] = tid < a.h_nout ? a.h_inv[tid] : 0.f, hbias[tid] = tid < a.h_nout ? a.h_bias[tid] : 0.f;

    // stationary weights: the M-side fragments of this wave's two 32-column blocks (blocks 2 wv, 2 wv + 1 of the eight)
    // The first P5_W_AGPR K steps' fragments are loaded straight into accumulation registers (an asm load with an "a" result:
    // left to itself the allocator keeps every weight in a VGPR, "spills" what does not fit to AGPRs and copies it back with
    // v_accvgpr_read in front of every use -- 136 copies per tile step); the MFMAs take them from there as they are.
    f16x8 wh[2][KS], wl[2][KS];
    constexpr int WA = KS2 > 0 ? P5_W_AGPR_SKIP : P5_W_AGPR;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
#pragma unroll
        for (int cb = 0; cb < 2; cb++) {
            const uint4* b = a.Bp + ((size_t)ks * 4 + g) * 256 + (2 * wv + cb) * 32 + li;
            if (ks < WA) {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wh[cb][ks]) : "v"(b) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wl[cb][ks]) : "v"(b + 512) : "memory");
            } else
                wh[cb][ks] = as_f16x8(b[0]), wl[cb][ks] = as_f16x8(b[512]);
        }
    }

    // HEADS: the heads' weight fragments of this wave's four K steps (columns 64 wv .. 64 wv + 63 of Y)
    f16x8 hwh[HEADS ? 4 : 1], hwl[HEADS ? 4 : 1];
    if (HEADS) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const uint4* hb = a.hBp + ((size_t)(4 * wv + kk) * 4 + g) * 32 + li;
            hwh[HEADS ? kk : 0] = as_f16x8(hb[0]), hwl[HEADS ? kk : 0] = as_f16x8(hb[64]);
        }
    }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc0 = zero16, acc1 = zero16;
    f32x16 hacc = zero16;  // HEADS: this wave's partial 32 x 16 (rows on the lanes; accumulator entries 0..7 are outputs (i & 3) + 8 (i >> 2) + 4 g)
    int eo_hs = 0;         // HEADS: exponent of the tile whose staged planes the next step multiplies
    f16x8 fh[2], fl[2];  // fragments one K step ahead of their MFMAs
    // K step ks_ of the tile in buffer ps1_ / ps2_: the embedding steps first
#define G5_FRAG(ks_)                                                                                                   \
    if (!(P5_ABL & 16) || (ks_) < 2) {                                                                                 \
        const unsigned char* pf_ = (ks_) < KS2 ? ps2 + (ks_) * 32 : ps1 + ((ks_) - KS2) * 32;                          \
        fh[(ks_) & 1] = as_f16x8(*reinterpret_cast<const uint4*>(pf_));                                                \
        fl[(ks_) & 1] = as_f16x8(*reinterpret_cast<const uint4*>(pf_ + ((ks_) < KS2 ? 128 : 512)));                    \
    }
#define G5_MFMA(ks_)                                                                                                   \
    {                                                                                                                  \
        const f16x8 ah_ = fh[(ks_) & 1], al_ = fl[(ks_) & 1];                                                          \
        if (P5_ABL & 4) {                                                                                              \
            if ((ks_) == 0) acc0 = zero16, acc1 = zero16;                                                              \
            acc0[0] += (float)ah_[0] + (float)al_[0];                                                                  \
        } else {                                                                                                       \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[0][ks_], ah_, (ks_) == 0 ? zero16 : acc0, 0, 0, 0);           \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[1][ks_], ah_, (ks_) == 0 ? zero16 : acc1, 0, 0, 0);           \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0][ks_], al_, acc0, 0, 0, 0);                                 \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[1][ks_], al_, acc1, 0, 0, 0);                                 \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0][ks_], ah_, acc0, 0, 0, 0);                                 \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[1][ks_], ah_, acc1, 0, 0, 0);                                 \
        }                                                                                                              \
    }

    P4_STEP_BARRIER();  // tiles 0 and 1 landed, tables visible, the asm-loaded weights arrived (vmcnt(0))
#pragma unroll
    for (int ks = 0; ks < KS && ks < WA; ks++) {  // (their uses may not be scheduled above the wait: route the values through it)
        asm volatile("" : "+a"(wh[0][ks]), "+a"(wl[0][ks]), "+a"(wh[1][ks]), "+a"(wl[1][ks]));
    }

    float pv0[16], pv1[16];   // raw accumulator sums of the previous tile; E1 turns them into its outputs in place, E2 splits them
    // lane constants of the staging-tile swizzle: 8-byte chunk u = 64 p + 8 b + 2 q + g (b = 2 wv + cb) of row r lives at chunk u ^ (r & 15)
    const unsigned st_x8_0 = (unsigned)(((8 * (2 * wv) + g) ^ (li & 15)) << 3);
    const unsigned st_x8_1 = (unsigned)(((8 * (2 * wv + 1) + g) ^ (li & 15)) << 3);
    unsigned char* const ow = Obuf + li * 1024;
    constexpr int H1 = KS / 2;       // K steps in front of the mid barrier
    constexpr int H2 = KS - H1;
    static_assert(H1 >= 8 && H2 >= 8, "eight E1 and eight E2 slices");

    int e_next = 0;                  // exponent that unscales the tile whose MFMAs ran last
    unsigned mh_next0 = 0u, mh_next1 = 0u;
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);  // bias of the next E1 slice (EPI 0), fetched one slice ahead

    auto step = [&](auto HM, auto HE, auto HS, const int j, const int ab) __attribute__((always_inline)) {
        const int pb = j & 1;
        const int tile = bx + j * G;
        const int abp = ab == 0 ? 2 : ab - 1;  // buffer of tile j-1 (= the one tile j+2 goes to)
        const int e_prev = e_next;
        const unsigned mh_prev0 = mh_next0, mh_prev1 = mh_next1;
        int dd = 0;
        if (HM.value) {
            const int e4 = j < limit ? exps[j] : a.Aexp[tile];
            if (KS2 > 0) {
                // the embedding steps run at scale 2^ee, the trunk at 2^e4: the accumulators move to min(e4, ee + 64) (beyond 2^64
                // a trunk tile is nothing beside the embedding's product -- and the clamp keeps the rescaled sums finite)
                const int ee = j < limit ? exps2[j] : a.A2exp[tile];
                dd = e4 - ee < 64 ? e4 - ee : 64;
                e_next = ee + dd;
            } else e_next = e4;
            if (EPI == 1) {
                const unsigned short* mp = reinterpret_cast<const unsigned short*>(Mibuf + ab * 1024) + (li * 8 + 2 * wv) * 2 + g;
                mh_next0 = mp[0], mh_next1 = mp[2];
            }
        }
        const float c = p4_pow2(-e_prev);
        unsigned bits0 = 0u, bits1 = 0u;
        float m = 0.f;
        int eo = 0;
        float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f);
        uint4 sv[8], smv = make_uint4(0u, 0u, 0u, 0u);
        const unsigned char* ps1 = Abuf + ab * ABYTES + li * P1 + g * 16;
        const unsigned char* ps2 = Abuf + ab * ABYTES + A1BYTES + li * P2 + g * 16;
        // E1 of elements 4 q + 3 .. 4 q of block cb_ of tile j-1: unscale, bias / mask, ReLU bit, running maximum -- in place
#define G5_E1_ONE(pv_, bits_, mh_, i_, b_, s_)                                                                         \
    {                                                                                                                  \
        float t_ = EPI == 0 ? pv_[i_] * c + (b_) : pv_[i_] * c;                                                        \
        if (EPI == 1) pv_[i_] = (((mh_) >> (i_)) & 1u) ? t_ * (s_) : 0.f;                                              \
        else {                                                                                                         \
            asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits_) : "v"(t_) : "vcc"); \
            pv_[i_] = p4_max(t_, 0.f) * (s_);                                                                          \
        }                                                                                                              \
        m = p4_max_abs(m, pv_[i_]);                                                                                    \
    }
#define G5_E1(s_)                                                                                                      \
    {                                                                                                                  \
        constexpr int cb_ = (s_) >> 2, q_ = 3 - ((s_) & 3);                                                            \
        const float4 bb_ = (EPI == 0 && !P5_BQ_AHEAD) ? *reinterpret_cast<const float4*>(biasl + (2 * wv + cb_) * 32 + 8 * q_ + 4 * g) : bq; \
        const float4 ss_ = *reinterpret_cast<const float4*>(scl + (2 * wv + cb_) * 32 + 8 * q_ + 4 * g);               \
        if (EPI == 0 && P5_BQ_AHEAD && (s_) < 7) {                                                                                    \
            constexpr int cn_ = ((s_) + 1) >> 2, qn_ = 3 - (((s_) + 1) & 3);                                           \
            bq = *reinterpret_cast<const float4*>(biasl + (2 * wv + cn_) * 32 + 8 * qn_ + 4 * g);                       \
        }                                                                                                              \
        if (cb_ == 0) {                                                                                                \
            G5_E1_ONE(pv0, bits0, mh_prev0, 4 * q_ + 3, bb_.w, ss_.w) G5_E1_ONE(pv0, bits0, mh_prev0, 4 * q_ + 2, bb_.z, ss_.z) \
            G5_E1_ONE(pv0, bits0, mh_prev0, 4 * q_ + 1, bb_.y, ss_.y) G5_E1_ONE(pv0, bits0, mh_prev0, 4 * q_ + 0, bb_.x, ss_.x) \
        } else {                                                                                                       \
            G5_E1_ONE(pv1, bits1, mh_prev1, 4 * q_ + 3, bb_.w, ss_.w) G5_E1_ONE(pv1, bits1, mh_prev1, 4 * q_ + 2, bb_.z, ss_.z) \
            G5_E1_ONE(pv1, bits1, mh_prev1, 4 * q_ + 1, bb_.y, ss_.y) G5_E1_ONE(pv1, bits1, mh_prev1, 4 * q_ + 0, bb_.x, ss_.x) \
        }                                                                                                              \
    }
        // E2 of the same four elements: scale to the tile exponent, split, 8 bytes per plane into the staging tile
#define G5_E2(s_)                                                                                                      \
    {                                                                                                                  \
        constexpr int cb_ = (s_) >> 2, q_ = (s_) & 3;                                                                  \
        unsigned h0_, l0_, h1_, l1_;                                                                                   \
        if (cb_ == 0) {                                                                                                \
            p4_split2(pv0[4 * q_], pv0[4 * q_ + 1], eo, h0_, l0_);                                                     \
            p4_split2(pv0[4 * q_ + 2], pv0[4 * q_ + 3], eo, h1_, l1_);                                                 \
        } else {                                                                                                       \
            p4_split2(pv1[4 * q_], pv1[4 * q_ + 1], eo, h0_, l0_);                                                     \
            p4_split2(pv1[4 * q_ + 2], pv1[4 * q_ + 3], eo, h1_, l1_);                                                 \
        }                                                                                                              \
        unsigned char* d_ = ow + ((cb_ == 0 ? st_x8_0 : st_x8_1) ^ (q_ << 4));                                         \
        *reinterpret_cast<uint2*>(d_) = make_uint2(h0_, h1_);                                                          \
        *reinterpret_cast<uint2*>(d_ + 512) = make_uint2(l0_, l1_);                                                    \
    }
        // tile j-2's staged row 8 wv + rr_: back from LDS / out as a 1 KiB row store
#define G5_SREAD(rr_)                                                                                                  \
    {                                                                                                                  \
        const int r_ = wv * 8 + (rr_);                                                                                 \
        sv[rr_] = *reinterpret_cast<const uint4*>(Obuf + r_ * 1024 + ((lane ^ ((r_ >> 1) & 7)) << 4));                 \
    }
#define G5_SROW(rr_)                                                                                                   \
    {                                                                                                                  \
        const int r_ = wv * 8 + (rr_);                                                                                 \
        uint4 val_ = sv[rr_];                                                                                          \
        if ((rr_) & 1) val_ = make_uint4(val_.z, val_.w, val_.x, val_.y);                                              \
        if (!(P5_ABL & 1) || val_.x == 0x12345678u)                                                                    \
            *reinterpret_cast<uint4*>(a.C + ((size_t)(tile - 2 * G) * 32 + r_) * 1024 + lane * 16) = val_;             \
    }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            if (HM.value) {
                if (ks + 1 < KS) G5_FRAG(ks + 1)
                G5_MFMA(ks)
                if (KS2 > 0 && ks == KS2 - 1) {  // embedding product -> the trunk's scale (exact power of two)
#pragma unroll
                    for (int i = 0; i < 16; i++) acc0[i] = __builtin_amdgcn_ldexpf(acc0[i], dd), acc1[i] = __builtin_amdgcn_ldexpf(acc1[i], dd);
                }
            }
            if (ks < H1) {
                if (HS.value) {  // tile j-2's rows: LDS reads up front, two row stores per K step early in the first half
                    // (two rows at a time, read one K step ahead of their stores: eight staged rows in flight at once cost the K = 320
                    // form the registers it does not have -- one weight fragment went to scratch, and a scratch reload waits for vmcnt(0))
                    if (ks == 0) {
                        G5_SREAD(0) G5_SREAD(1)
                        if (EPI == 0 && wv == 3) smv = reinterpret_cast<const uint4*>(mbuf + pb * 256)[lane];
                    }
                    if (ks == 1) { G5_SROW(0) G5_SROW(1) G5_SREAD(2) G5_SREAD(3) }
                    if (ks == 2) { G5_SROW(2) G5_SROW(3) G5_SREAD(4) G5_SREAD(5) }
                    if (ks == 3) { G5_SROW(4) G5_SROW(5) G5_SREAD(6) G5_SREAD(7) }
                    if (ks == 4) {
                        G5_SROW(6) G5_SROW(7)
                        if (EPI == 0 && wv == 3) reinterpret_cast<uint4*>(a.mask_out + (size_t)(tile - 2 * G) * 256)[lane] = smv;
                    }
                    if (HEADS && ks >= 4 && ks < 8) {
                        // K step 4 wv + kk of the heads on tile j-2's staged planes: columns 16 (4 wv + kk) + 8 g .. + 7 of row li are the
                        // 8-byte chunks u0, u0 + 1 (u0 = 4 (4 wv + kk) + 2 g) of the row, stored at chunk u ^ (li & 15): one 16-byte read
                        // at (u0 ^ (s & ~1)), its halves swapped when s is odd; the low plane 512 bytes further
                        const int kk = ks - 4;
                        const unsigned s_ = (unsigned)li & 15u;
                        const unsigned char* pr = Obuf + li * 1024 + (((unsigned)(4 * (4 * wv + kk) + 2 * g) ^ (s_ & ~1u)) << 3);
                        uint4 vh = *reinterpret_cast<const uint4*>(pr), vl = *reinterpret_cast<const uint4*>(pr + 512);
                        if (s_ & 1u) vh = make_uint4(vh.z, vh.w, vh.x, vh.y), vl = make_uint4(vl.z, vl.w, vl.x, vl.y);
                        const f16x8 bh_ = as_f16x8(vh), bl_ = as_f16x8(vl);
                        hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hwl[HEADS ? kk : 0], bh_, kk == 0 ? zero16 : hacc, 0, 0, 0);
                        hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hwh[HEADS ? kk : 0], bl_, hacc, 0, 0, 0);
                        hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hwh[HEADS ? kk : 0], bh_, hacc, 0, 0, 0);
                    }
                }
                if (HE.value && !(P5_ABL & 8)) {
                    if (ks == 0) G5_E1(0)
                    if (ks == 1) G5_E1(1)
                    if (ks == 2) G5_E1(2)
                    if (ks == 3) G5_E1(3)
                    if (ks == 4) G5_E1(4)
                    if (ks == 5) G5_E1(5)
                    if (ks == 6) G5_E1(6)
                    if (ks == 7) G5_E1(7)
                }
                if (ks == H1 - 1) {
                    if (HE.value) {
                        m = wave_max_nonneg_lane63(m);
                        if (lane == 63) tmaxs[wv] = m;
                        if (EPI != 1) {
                            unsigned short* mw = reinterpret_cast<unsigned short*>(mbuf + (1 - pb) * 256) + (li * 8 + 2 * wv) * 2 + g;
                            mw[0] = (unsigned short)bits0, mw[2] = (unsigned short)bits1;
                        }
                    }
                    if (HEADS && HS.value) {  // this wave's partial heads of tile j-2: outputs 4 g .. 4 g + 3 and 8 + 4 g .. of row li
                        float* hw_ = hp + ((size_t)wv * 32 + li) * 16 + 4 * g;
                        *reinterpret_cast<float4*>(hw_) = make_float4(hacc[0], hacc[1], hacc[2], hacc[3]);
                        *reinterpret_cast<float4*>(hw_ + 8) = make_float4(hacc[4], hacc[5], hacc[6], hacc[7]);
                    }
                    P4_STEP_BARRIER();  // MID: the tile maximum needs all four waves; the vector memory issued a step ago is done
                    if (HE.value) t0 = *reinterpret_cast<const float4*>(tmaxs);
                    if (HEADS && HS.value) {  // the four partials in wave order, unscaled, + bias: two outputs of one row per thread
                        const int hr = tid >> 3, ho = (tid & 7) * 2;
                        const float2 p0 = *reinterpret_cast<const float2*>(hp + ((size_t)0 * 32 + hr) * 16 + ho);
                        const float2 p1 = *reinterpret_cast<const float2*>(hp + ((size_t)1 * 32 + hr) * 16 + ho);
                        const float2 p2 = *reinterpret_cast<const float2*>(hp + ((size_t)2 * 32 + hr) * 16 + ho);
                        const float2 p3 = *reinterpret_cast<const float2*>(hp + ((size_t)3 * 32 + hr) * 16 + ho);
                        const float2 iv = *reinterpret_cast<const float2*>(hinv + ho), bv = *reinterpret_cast<const float2*>(hbias + ho);
                        const float ch = p4_pow2(-eo_hs);
                        const int row = (tile - 2 * G) * 32 + hr;
                        if (row < a.M) {
                            if (ho < a.h_nout) a.h_out[(size_t)row * a.h_nout + ho] = (((p0.x + p1.x) + p2.x) + p3.x) * (ch * iv.x) + bv.x;
                            if (ho + 1 < a.h_nout) a.h_out[(size_t)row * a.h_nout + ho + 1] = (((p0.y + p1.y) + p2.y) + p3.y) * (ch * iv.y) + bv.y;
                        }
                    }
                }
            } else {
                const int s2 = ks - H1;
                if (s2 == 0 && HE.value) {  // the tile's exponent from the four wave maxima
                    const float tm = p4_max(p4_max(t0.x, t0.y), p4_max(t0.z, t0.w));
                    eo = p4_exp_from_max_bits(__float_as_uint(tm));
                    if (tid == 0) a.Cexp[tile - G] = eo;
                }
                if (HM.value && j + 2 < my_tiles && !(P5_ABL & 2)) {  // tile j+2 into the buffer tile j-1 left: one copy instruction per K step
#pragma unroll
                    for (int i = 0; i < NCP; i++)
                        if (s2 == (i < H2 ? i : i - H2)) G5_COPY1(tile + 2 * G, abp, i)
                    if (s2 == 1) G5_COPY_MASK(tile + 2 * G, abp)
                }
                if (HE.value && !(P5_ABL & 8)) {
                    if (s2 == 0) G5_E2(0)
                    if (s2 == 1) G5_E2(1)
                    if (s2 == 2) G5_E2(2)
                    if (s2 == 3) G5_E2(3)
                    if (s2 == 4) G5_E2(4)
                    if (s2 == 5) G5_E2(5)
                    if (s2 == 6) G5_E2(6)
                    if (s2 == 7) G5_E2(7)
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef G5_E1_ONE
#undef G5_E1
#undef G5_E2
#undef G5_SREAD
#undef G5_SROW
        if (HEADS && HE.value) eo_hs = eo;  // (tile j-1's exponent: the next step multiplies its staged planes)
        if (HM.value) {
#pragma unroll
            for (int i = 0; i < 16; i++) pv0[i] = acc0[i], pv1[i] = acc1[i];
            if (P5_ABL & 8) {  // (ablation: a never-true sink keeps the accumulators alive)
                float sink_ = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i++) sink_ += pv0[i] + pv1[i];
                if (sink_ == 1.2345678e33f) a.Cexp[0] = 1;
            }
            // head start for the next step (its tile landed before this step's mid barrier): first fragments, first bias slice
            if (j + 1 < my_tiles) {
                const int abn = ab == 2 ? 0 : ab + 1;
                const unsigned char* ps1 = Abuf + abn * ABYTES + li * P1 + g * 16;
                const unsigned char* ps2 = Abuf + abn * ABYTES + A1BYTES + li * P2 + g * 16;
                G5_FRAG(0)
            }
            if (EPI == 0 && P5_BQ_AHEAD) bq = *reinterpret_cast<const float4*>(biasl + (2 * wv) * 32 + 8 * 3 + 4 * g);
        }
        // END: the staging tile and the A buffer change hands (the reads just issued stay in flight: this wave's LDS writes are
        // older and LDS operations complete in order)
        if (HM.value && j + 1 < my_tiles) asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" ::"n"((EPI == 0 && P5_BQ_AHEAD ? 1 : 0) + 2) : "memory");
        else if (HM.value && EPI == 0 && P5_BQ_AHEAD) asm volatile("s_waitcnt lgkmcnt(1)\n\ts_barrier" ::: "memory");
        else P4_LDS_BARRIER();
    };
    {   // first fragments of tile 0
        const unsigned char* ps1 = Abuf + li * P1 + g * 16;
        const unsigned char* ps2 = Abuf + A1BYTES + li * P2 + g * 16;
        G5_FRAG(0)
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
#undef G5_COPY1
#undef G5_COPY_MASK
#undef G5_FRAG
#undef G5_MFMA
}

template <int KS2, int EPI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
mlp_gemm5_kernel(const Gemm5Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char p4_smem[];
    gemm5_body<KS2, EPI>(a, (int)blockIdx.x, (int)gridDim.x, p4_smem);
}

// ---- the time embedding as a bias -----------------------------------------------------------------------------------------------
// With ONE time value per call (R/train.py:158: fid expanded over the Gaussians) the time columns of the two layers that consume
// the embedding contribute the same vector to every row: beff[j] = b[j] + sum_t W[j][63 + t] t_emb[t]   (R/utils/time_utils.py:
// 104-129: h = cat([x_emb, t_emb]) -> linear[0]; the skip re-injection feeds linear[5] the same way).  The fold itself happens where
// the two layers' biases are pre-scaled (mlp_prep4c_kernel, mlp_planes.hpp: `fold` jobs); first version: a launch of its own.

// The adjoint of the fold (one launch, after the reduction has produced db; grid (3 or 2, T)).  (Tried: the same work inside
// mlp_reduce_dw_all_kernel -- its bias-reducing workgroups writing the time columns, the last of them to arrive dL/dt_emb -- saves this
// launch and costs the HBM-bound reduction more than that: 304.0 / 302.5 / 304.0 against 309.8 / 310.7 / 310.6 it/s.):
//   blockIdx.x = 0, 1: the weights' time columns of layer 0 / the skip layer, dW[j][63 + t] = db[j] t_emb[t];
//   blockIdx.x = 2   : dL/dt_emb[t] = sum_j db0[j] W0[j][63 + t] + db5[j] W5[j][63 + t]  (first version: mlp_dtemb_bcast_kernel, a
//                      launch of its own behind this one -- still the per-row-time path's; same summation order here).
struct FoldGradArgs {
    float* dW[2];
    const float* db[2];
    const float* W[2];
    int in_features[2];
    const float* temb;
    float* dtemb;  // may be NULL (then the grid is (2, T))
    int T;
};
__global__ void __launch_bounds__(256)
mlp_fold_grad_kernel(const FoldGradArgs f) {
    __shared__ float red[4];
    const int k = blockIdx.x, t = blockIdx.y, j = threadIdx.x;
    if (k < 2) {  // (workgroup-uniform)
        f.dW[k][(size_t)j * f.in_features[k] + 63 + t] = f.db[k][j] * f.temb[t];
        return;
    }
    float v = f.db[0][j] * f.W[0][(size_t)j * f.in_features[0] + 63 + t] + f.db[1][j] * f.W[1][(size_t)j * f.in_features[1] + 63 + t];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((j & 63) == 0) red[j >> 6] = v;
    __syncthreads();
    if (j == 0) f.dtemb[t] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace dgm
