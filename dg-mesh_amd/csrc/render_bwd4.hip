// render_bwd, fourth generation: ONE WAVE PER (TILE, UNIT), four pixels per lane, moment accumulation.
//
// Replays BACKWARD::render / renderCUDA<3> (DGR/cuda_rasterizer/backward.cu:401-557) and writes one 36-byte row per
// (tile, splat) instance that some pixel blended, plus one liveness byte per instance.  What changed against
// render_bwd3 (round 2/3; DESIGN.md section 4b has the cycle accounting, tools/valu_micro.hip the instruction costs):
//   * the cross-lane reduction.  A fully processed splat cost 40 (header) + 2 x 208 (half tiles) + 205 cycles of VALU
//     issue (tools/valu_micro.hip: a plain VALU instruction occupies the SIMD for 4 cycles, a packed-f32 one 4.6, exp / rcp
//     8, a DPP add 4.2, a permlane swap 9.4), the last item being the transposing butterfly (9 adds, 14 selects, 16 DPP adds,
//     2 swaps and their wait states).  The stages now run in the order that needs no selects (wave_reduce8m: bank-masked
//     row shifts first, register-pair swaps next, the quad stages last on the one register left): 14 DPP adds + 2 swaps.
//     Tried and measured slower (RB4_REDUCE=1, kept for the record): the nine sums on the matrix pipe --
//     v_mfma_f32_16x16x4_f32 with A = the lane's value v and B = [lane % 16 == v] gives D[i][v] = sum_k q_v[16 k + i];
//     three in-lane adds and two swaps finish it.  75 cycles of VALU issue instead of 205, but the fp32 MFMA is issued at
//     the vector pipe's rate and did not overlap with other waves' VALU work (0.456 vs 0.40 ms at cfg2);
//   * the moments in x.  A lane's four pixels share their column, so dx is a per-lane constant of a splat: only
//     sum g, sum g dy, sum g dy^2 are accumulated per pixel and the x factors are applied once per lane
//     (-6 packed instructions per half tile);
//   * culling.  A half tile is entered only if the ellipse {alpha >= 1/255} itself -- not its bounding box -- reaches
//     the half's 16 x 8 block of pixel centres (minimum of the quadric over the rectangle, evaluated by the staging
//     lane: one lane per splat);
//   * units.  Tiles with at most 4096 list entries -- every tile of the BASELINE scenes -- are replayed in units of 64 entries
//     from the forward's 64-entry checkpoints (render.hip: `ckpt64`, 64 bytes per entry), 16 workgroups per tile.  A unit is
//     a serial chain of ~1000 cycles per blended entry for a lone wave; on a trained scene (lists of 200-1400 entries, half
//     the tiles empty) the kernel's duration was that of ONE 256-entry segment -- 0.9 waves resident per SIMD, VALU busy 0.26
//     (`profiles/r04_rasterbench_sq_trained_ksplit8.json`) -- and on the initial scene the shorter units trim the tail as well.  Longer
//     lists keep 256-entry segments (16 bytes of checkpoints per entry);
//   * rows.  36 bytes (9 floats), written only for instances some pixel blended; `live[row]` (one byte per instance,
//     always written) tells preprocess_bwd which rows to read.  No zero rows are written or read.
// No atomics, bit-reproducible (fixed reduction and summation order).
#include "dgm_common.hpp"
#include "render_common.hpp"

namespace dgm {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#ifndef RB4_KSPLIT_N
#define RB4_KSPLIT_N 16
#endif
static constexpr int RB4_KSPLIT = RB4_KSPLIT_N;  // workgroups per tile; units are dealt round-robin (a 1024-entry tile: one 64-entry unit each)
static constexpr int RB4_RS = 9;      // floats per staged output row (= DGM_SLAB_STRIDE)

// replay state of one row pair (two pixels of the lane)
struct Pair4 {
    f2 T, S;           // transmittance behind the current splat; S' = sum_ch S_ch dL/dC_ch + T_final (bg . dL/dC)
    f2 dpr, dpg, dpb;  // dL/dC of the two pixels
    f2 py;             // pixel rows
    unsigned lc0, lc1; // n_contrib; 0 for pixels outside the image
};

// per-lane partial sums of one splat over the lane's pixels
struct Sums4 {
    f2 c0, c1, c2;  // colour: sum w dL/dC_ch
    f2 s0, s1, s2;  // g, g dy, g dy^2   with g = G dL/dalpha
};

// One row pair of one splat.  FIRST: the sums are written, not added to (saves their zero-initialisation).  Returns false
// -- state and sums untouched -- when no pixel of the half blends the splat.
template <bool FIRST>
__device__ __forceinline__ bool blend_pair4(Pair4& p, const float4 A, const float4 B, const float cb, const float adx2,
                                            const float bdx, const unsigned cidx, Sums4& q) {
    // A.z, A.w, B.x hold the conic pre-multiplied by -log2(e)/2, -log2(e), -log2(e)/2 (staging), so `power` is the reference's
    // exponent times log2(e): same sign, and G = 2^power
    const f2 dy = A.y - p.py;
    const f2 power = (B.x * dy) * dy + adx2 + bdx * dy;
    f2 G;
    G.x = __builtin_amdgcn_exp2f(power.x);
    G.y = __builtin_amdgcn_exp2f(power.y);
    f2 alpha = B.y * G;
    alpha.x = fminf(0.99f, alpha.x);
    alpha.y = fminf(0.99f, alpha.y);
    // the decisions as lane masks in scalar registers: one ballot per compare (a ballot of a compound condition is lowered
    // through a select and a second compare), combined with scalar ands
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(cidx < p.lc0) & __builtin_amdgcn_ballot_w64(!(power.x > 0.0f)) &
                                  __builtin_amdgcn_ballot_w64(!(alpha.x < 1.0f / 255.0f));
    const unsigned long long m1 = __builtin_amdgcn_ballot_w64(cidx < p.lc1) & __builtin_amdgcn_ballot_w64(!(power.y > 0.0f)) &
                                  __builtin_amdgcn_ballot_w64(!(alpha.y < 1.0f / 255.0f));
    if ((m0 | m1) == 0ull) return false;
    const f2 vm = {__builtin_amdgcn_inverse_ballot_w64(m0) ? 1.f : 0.f, __builtin_amdgcn_inverse_ballot_w64(m1) ? 1.f : 0.f};
    alpha = alpha * vm;  // a skipped pixel is alpha = 0: T, S' stay, every sum gets zero
    const f2 one_m_a = 1.f - alpha;
    f2 inv;
    inv.x = __builtin_amdgcn_rcpf(one_m_a.x);
    inv.y = __builtin_amdgcn_rcpf(one_m_a.y);
    const f2 Tn = p.T * inv;  // transmittance in front of this splat
    const f2 w = alpha * Tn;  // dC/dcolour
    const f2 cdp = B.z * p.dpr + B.w * p.dpg + cb * p.dpb;
    const f2 dL_dalpha = (p.T * cdp - p.S) * inv * vm;
    p.S += cdp * w;
    p.T = Tn;
    const f2 g = G * dL_dalpha;
    const f2 gdy = g * dy;
    if (FIRST) {
        q.c0 = w * p.dpr, q.c1 = w * p.dpg, q.c2 = w * p.dpb;
        q.s0 = g, q.s1 = gdy, q.s2 = gdy * dy;
    } else {
        q.c0 += w * p.dpr, q.c1 += w * p.dpg, q.c2 += w * p.dpb;
        q.s0 += g, q.s1 += gdy, q.s2 += gdy * dy;
    }
    return true;
}

#ifndef RB4_REDUCE
#define RB4_REDUCE 0  // 0: select-free DPP butterfly (render_common.hpp: wave_reduce8m); 1: v_mfma_f32_16x16x4_f32 with selector columns
#endif
#ifndef RB4_XCD_MAP
#define RB4_XCD_MAP 1
#endif
#ifndef RB4_WAVES_PER_EU
#define RB4_WAVES_PER_EU 5
#endif
#ifndef RB4_TRACE
#define RB4_TRACE 0  // 1: every wave that had work records (start, end, hardware id, blended entries): tools/raster_bench.py --trace
#endif
#if RB4_TRACE
__device__ unsigned long long rb4_trace[4 * 65536 + 1];
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RB4_WAVES_PER_EU, RB4_WAVES_PER_EU)))
render_bwd4_kernel(const uint2* __restrict__ ranges, const unsigned* __restrict__ point_list, int W, int H, int gridx,
                   const float* __restrict__ bg, const float* __restrict__ rec, const float4* __restrict__ cfin,
                   const float4* __restrict__ ckpt, const float4* __restrict__ ckpt64, const unsigned* __restrict__ n_contrib,
                   const float* __restrict__ dL_dpixels, const unsigned* __restrict__ nproc_in,
                   const unsigned* __restrict__ upos, float* __restrict__ slab, uint8_t* __restrict__ live, const int ntiles) {
    // staged splats, 48 bytes each: x, y, conic a * -log2(e)/2, conic b * -log2(e) | conic c * -log2(e)/2, opacity, r, g | b
    __shared__ float4 sR[64 * 3];
    __shared__ float sOut[64 * RB4_RS];
    // Workgroup -> (tile, unit slot).  Consecutive workgroup ids go round the 8 XCDs (id % 8), each with its own L2.  All 16 unit
    // slots of a tile, and 8 horizontally adjacent tiles, are given to ONE XCD and to nearby ids: the tile's pixel state (8 KB: final
    // blend state, dL/dpixel, n_contrib) is then fetched into that L2 once instead of once per unit, and the 36-byte rows of a
    // Gaussian's neighbouring instances -- adjacent in the slab -- meet in the same L2 before they are written back.
    //   id = (((tile / 64) * KSPLIT + slot) * 8 + tile % 8) * 8 + (tile / 8) % 8
    const unsigned wid = blockIdx.x;
#if RB4_XCD_MAP
    const unsigned xcd = wid & 7u, qq = wid >> 3;
    const unsigned rr = qq >> 3;
    const int slot = (int)(rr % RB4_KSPLIT);
    const int tile = (int)(((rr / RB4_KSPLIT) << 6) | (xcd << 3) | (qq & 7u));
    if (tile >= ntiles) return;
#else  // (A/B: slot-major, a tile's units 2500 ids apart and on alternating XCDs -- the round-4 form before this mapping)
    const int slot = (int)(wid / (unsigned)ntiles), tile = (int)(wid % (unsigned)ntiles);
    if (slot >= RB4_KSPLIT) return;
#endif
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const int nproc = (int)nproc_in[tile];
    const bool shortlist = n <= DGM_SHORT_LIST;
    const int ulen = shortlist ? 64 : 256;
    const int nunits = (nproc + ulen - 1) / ulen;
    // list entries behind the deepest contributor of the tile are never replayed: dead, no row; the tile's workgroups share them
    for (int pos = nproc + slot * 64 + (int)threadIdx.x; pos < n; pos += RB4_KSPLIT * 64) live[upos[range.x + pos]] = 0;
    if (slot >= nunits) return;
#if RB4_TRACE
    const unsigned long long tr_t0 = __builtin_readcyclecounter(), tr_w0 = wall_clock64();
    unsigned tr_blended = 0;
#endif
    const int tile_x = tile % gridx, tile_y = tile / gridx;
    const int lane = threadIdx.x;
    const int px = tile_x * DGM_TILE + (lane & 15);
    const int pyb = tile_y * DGM_TILE + (lane >> 4);  // rows pyb + {0, 4, 8, 12}
    const float pxf = (float)px;
    const float tx0 = (float)(tile_x * DGM_TILE), ty0 = (float)(tile_y * DGM_TILE);
    const size_t plane = (size_t)W * H;

#if RB4_REDUCE == 1
    // B operands of the matrix-pipe reduction: column selector [lane % 16 == v]
    float bsel[9];
#pragma unroll
    for (int v = 0; v < 9; v++) bsel[v] = (lane & 15) == v ? 1.f : 0.f;
#endif

    // per-pixel constants; pixel j of this lane = row pyb + 4 j, state index 64 j + lane (the forward's checkpoint order)
    Pair4 P[2];
    f2 full[2];  // C_final . dL/dC + T_final (bg . dL/dC): S' at the very front of the list
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        float dr[2], dg[2], db[2], tf[2], cf[2];
        unsigned lc[2];
        bool in[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int j = 2 * h + e, py = pyb + 4 * j;
            in[e] = px < W && py < H;
            const size_t pid = (size_t)W * py + px;
            const float4 f = cfin[(size_t)tile * 256 + j * 64 + lane];
            dr[e] = in[e] ? dL_dpixels[pid] : 0.f;
            dg[e] = in[e] ? dL_dpixels[plane + pid] : 0.f;
            db[e] = in[e] ? dL_dpixels[2 * plane + pid] : 0.f;
            lc[e] = in[e] ? n_contrib[pid] : 0u;
            tf[e] = f.x * (bg0 * dr[e] + bg1 * dg[e] + bg2 * db[e]);
            cf[e] = f.y * dr[e] + f.z * dg[e] + f.w * db[e] + tf[e];
            if (e == 0) P[h].T.x = f.x, P[h].S.x = tf[e];
            else P[h].T.y = f.x, P[h].S.y = tf[e];
        }
        P[h].dpr = (f2){dr[0], dr[1]};
        P[h].dpg = (f2){dg[0], dg[1]};
        P[h].dpb = (f2){db[0], db[1]};
        P[h].py = (f2){(float)(pyb + 8 * h), (float)(pyb + 8 * h + 4)};
        P[h].lc0 = lc[0], P[h].lc1 = lc[1];
        full[h] = (f2){cf[0], cf[1]};
    }
    const f2 Tfin[2] = {P[0].T, P[1].T}, Sfin[2] = {P[0].S, P[1].S};
    // deepest contributor of each half tile (wave-uniform): list entries at or beyond it cannot touch that half
    const unsigned lc_top = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_max_u32(max(P[0].lc0, P[0].lc1)));
    const unsigned lc_bot = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_max_u32(max(P[1].lc0, P[1].lc1)));

    for (int k = slot; k < nunits; k += RB4_KSPLIT) {
        const int seg_begin = k * ulen;
        const int seg_end = min(nproc, seg_begin + ulen);
        // replay state at the back end of the unit
        if (k == nunits - 1) {
#pragma unroll
            for (int h = 0; h < 2; h++) P[h].T = Tfin[h], P[h].S = Sfin[h];
        } else {
            const float4* cp = shortlist ? ckpt64 + ((size_t)(range.x >> 6) + (size_t)(k + 1)) * 256
                                         : ckpt + (size_t)((range.x + ((unsigned)(k + 1) << 8)) >> 8) * 256;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float4 c0 = cp[(2 * h) * 64 + lane], c1 = cp[(2 * h + 1) * 64 + lane];
                P[h].T = (f2){c0.x, c1.x};
                const f2 front = {c0.y * P[h].dpr.x + c0.z * P[h].dpg.x + c0.w * P[h].dpb.x,
                                  c1.y * P[h].dpr.y + c1.z * P[h].dpg.y + c1.w * P[h].dpb.y};
                P[h].S = full[h] - front;  // (C_final - C_front) . dL/dC + T_final (bg . dL/dC)
            }
        }
        const int nb = (seg_end - seg_begin + 63) >> 6;
        for (int t = 0; t < nb; t++) {
            const int base_pos = seg_end - 1 - t * 64;  // list position staged by lane 0; lane l stages base_pos - l
            const int pos = base_pos - lane;
            unsigned qm = 0u, row = 0u;
            if (pos >= seg_begin) {
                const unsigned g = point_list[range.x + pos];
                row = upos[range.x + pos];  // fetched with the splat: the dependent stores below do not wait for it
                const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g * DGM_REC_STRIDE);
                const float4 r0 = r4[0], r1 = r4[1];
                const float l2e = 1.4426950408889634f;
                sR[3 * lane] = make_float4(r0.x, r0.y, -0.5f * l2e * r0.z, -l2e * r0.w);
                sR[3 * lane + 1] = make_float4(-0.5f * l2e * r1.x, r1.y, r1.z, r1.w);
                sR[3 * lane + 2].x = r4[2].x;
                qm = half_mask(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, tx0, ty0);
            }
            const unsigned long long m_top = uniform_u64(__ballot(qm & 1u));
            const unsigned long long m_bot = uniform_u64(__ballot(qm & 2u));
            unsigned long long m = m_top | m_bot;
            unsigned long long alive = 0ull;  // splats of this batch that some pixel blended (wave-uniform)
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                const float4 A = sR[3 * j];
                const float4 B = sR[3 * j + 1];
                const float cb = sR[3 * j + 2].x;
                const unsigned cidx = (unsigned)(base_pos - j);  // contributor index (backward.cu:486-488)
                const float dx = A.x - pxf;
                const float adx2 = A.z * dx * dx, bdx = A.w * dx;
                const bool top = ((m_top >> j) & 1ull) && cidx < lc_top;
                const bool bot = ((m_bot >> j) & 1ull) && cidx < lc_bot;
                Sums4 q;
                // (the two halves as ONE instruction stream -- two independent chains for a lone wave to interleave -- was measured
                // and dropped: 126 registers or 33 spilled ones, 0.289 -> 0.339 ms on the initial scene, 0.079 -> 0.086 trained-like)
                bool any = top && blend_pair4<true>(P[0], A, B, cb, adx2, bdx, cidx, q);
                if (bot) {
                    if (!any) q.c0 = q.c1 = q.c2 = q.s0 = q.s1 = q.s2 = (f2){0.f, 0.f};  // (only when the top half did not write them)
                    any |= blend_pair4<false>(P[1], A, B, cb, adx2, bdx, cidx, q);
                }
#if RB4_TRACE
                tr_blended += any ? 1u : 0u;
#endif
                if (!any) continue;  // wave-uniform: inside the ellipse's reach, but no pixel blends it -- dead, no row
                alive |= 1ull << j;
                // the lane's nine values: colour r, g, b | moments dx, dy | dx^2, dx dy, dy^2 | 1
                const float C0 = q.c0.x + q.c0.y, C1 = q.c1.x + q.c1.y, C2 = q.c2.x + q.c2.y;
                const float S0 = q.s0.x + q.s0.y, S1 = q.s1.x + q.s1.y, S2 = q.s2.x + q.s2.y;
                const float Mx = dx * S0;
#if RB4_REDUCE == 1
                // (measured: slower than the butterfly -- the fp32 MFMA is issued at the vector pipe's own rate and did not
                // overlap with the other waves' VALU work: render_bwd 0.456 vs 0.40 ms at cfg2; kept for the record)
                f4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(C0, bsel[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(C1, bsel[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(C2, bsel[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Mx, bsel[3], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(S1, bsel[4], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dx * Mx, bsel[5], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dx * S1, bsel[6], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(S2, bsel[7], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(S0, bsel[8], acc, 0, 0, 0);
                // lane (g, j) register r = sum of value j over the lanes {16 k + 4 g + r}: add the registers, then the four groups
                const float part = (acc[0] + acc[1]) + (acc[2] + acc[3]);
                const unsigned xb = __float_as_uint(part);
                auto s16 = __builtin_amdgcn_permlane16_swap(xb, xb, false, false);
                const float r32 = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
                const unsigned yb = __float_as_uint(r32);
                auto s32 = __builtin_amdgcn_permlane32_swap(yb, yb, false, false);
                const float tot = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
                if (lane < RB4_RS) sOut[j * RB4_RS + lane] = tot;
#else
                // row: colour r, g, b | moments dx, dy | dx^2, dx dy, dy^2 | 1.  Lane l ends up with value (l >> 2) & 7.
                const float r = wave_reduce8m(C0, C1, C2, Mx, S1, dx * Mx, dx * S1, S2);
                const float r8 = wave_reduce1_lane63(S0);
                if ((lane & 35) == 0) sOut[j * RB4_RS + (lane >> 2)] = r;  // lanes 0, 4, .., 28
                if (lane == 63) sOut[j * RB4_RS + 8] = r8;
#endif
            }
            if (pos >= seg_begin) {
                const bool is_live = (alive >> lane) & 1ull;
                live[row] = is_live ? 1 : 0;
                if (is_live) {
                    const float* o = sOut + lane * RB4_RS;
                    // row of the instance in the per-Gaussian order: the sum over a Gaussian's instances reads adjacent rows
                    float* dst = slab + (size_t)row * DGM_SLAB_STRIDE;
                    *reinterpret_cast<dgm_f4u*>(dst) = (dgm_f4u){o[0], o[1], o[2], o[3]};
                    *reinterpret_cast<dgm_f4u*>(dst + 4) = (dgm_f4u){o[4], o[5], o[6], o[7]};
                    dst[8] = o[8];
                }
            }
        }
    }
#if RB4_TRACE
    if (threadIdx.x == 0) {
        const unsigned long long tr_t1 = __builtin_readcyclecounter(), tr_w1 = wall_clock64();
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
        const unsigned long long i = atomicAdd(&rb4_trace[4 * 65536], 1ull) & 65535ull;
        // (s_memtime is a per-CU counter: durations only; start / end on the constant 100 MHz clock)
        rb4_trace[4 * i] = tr_w0, rb4_trace[4 * i + 1] = tr_w1, rb4_trace[4 * i + 2] = ((unsigned long long)xcc << 32) | hw;
        rb4_trace[4 * i + 3] = ((tr_t1 - tr_t0) << 32) | ((unsigned long long)tr_blended << 24) | ((unsigned long long)tile << 4) | (unsigned)(slot & 15);
    }
#endif
}
#if RB4_TRACE
extern "C" int dgm_debug_rb4_trace(void* dst, size_t bytes, int reset) {
    int e = (int)hipDeviceSynchronize();
    if (!e) e = (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(rb4_trace), bytes);
    unsigned long long z = 0;
    if (!e && reset) e = (int)hipMemcpyToSymbol(HIP_SYMBOL(rb4_trace), &z, 8, 4 * 65536 * 8);
    return e;
}
#endif

void launch_render_bwd4(hipStream_t st, int tiles, const uint2* ranges, const unsigned* point_list, int W, int H,
                        int gridx, const float* bg, const float* rec, const float4* cfin, const float4* ckpt,
                        const float4* ckpt64, const unsigned* n_contrib, const float* dL_dpix, const unsigned* nproc,
                        const unsigned* upos, float* slab, uint8_t* live) {
    const unsigned groups = ((unsigned)tiles + 63u) / 64u;
    hipLaunchKernelGGL(render_bwd4_kernel, dim3(groups * RB4_KSPLIT * 64u), dim3(64), 0, st, ranges, point_list, W, H, gridx, bg, rec,
                       cfin, ckpt, ckpt64, n_contrib, dL_dpix, nproc, upos, slab, live, tiles);
}

}  // namespace dgm
