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
//     (32 on sparse frames, R < 2^20) from the forward's checkpoints (render.hip: `ckpt64`), handed out by tickets (round 5, below).  A unit is
//     a serial chain of ~1000 cycles per blended entry for a lone wave; on a trained scene (lists of 200-1400 entries, half
//     the tiles empty) the kernel's duration was that of ONE 256-entry segment -- 0.9 waves resident per SIMD, VALU busy 0.26
//     (`profiles/r04_rasterbench_sq_trained_ksplit8.json`) -- and on the initial scene the shorter units trim the tail as well.  Longer
//     lists keep 256-entry segments (16 bytes of checkpoints per entry);
//   * rows.  36 bytes (9 floats), written only for instances some pixel blended; `live[row]` (one byte per instance, cleared by
//     render_fwd, set here for the rows written) tells preprocess_bwd which rows to read.  No zero rows are written or read.
// No atomics, bit-reproducible (fixed reduction and summation order).
#include "dgm_common.hpp"
#include "render_common.hpp"

namespace dgm {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

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
#ifndef RB4_WAVES_PER_EU
#define RB4_WAVES_PER_EU 5
#endif
#ifndef RB4_TRACE
#define RB4_TRACE 0  // 1: every wave records (start, end, hardware id, blended entries, units): tools/raster_bench.py --trace
#endif
#if RB4_TRACE
__device__ unsigned long long rb4_trace[4 * 65536 + 1];
#endif

// Round 5: a COMPACT work list, statically mapped.  The forward lists the replay units -- (tile, 64- / 32- / 256-entry run of the
// tile's list) -- as its tiles finish (render.hip): the full units in one array (a tile's run of them contiguous), the tiles' last,
// shorter units in another.  The launch has one single-wave workgroup per POSSIBLE unit (the host only knows the bound R / u +
// tiles).  Consecutive workgroup ids go round the 8 XCDs (id % 8): XCD x takes the x-th eighth of the full units and then the x-th
// eighth of the last units, in id order -- so every XCD works through the same mix, long units first, a tile's units meet in one L2
// (its pixel state is fetched there once; the neighbouring rows of a Gaussian's instances merge there), and a workgroup beyond the
// listed count leaves after one read of the two counters.  The hardware dispatcher balances: a SIMD gets the next workgroup
// whenever one of its five slots frees up.  Why: round 4 launched 16 unit slots per tile; on a trained scene 36 k of its 41 k
// workgroups had nothing to do and sat between the working ones in dispatch order, working waves landed 4 to 8 per SIMD where 5
// fit, and the kernel lasted as long as the SIMD that drew the most list entries (p50 174, max 286).  A unit's rows do not depend on
// who computes them, nor on the order of the list: results are bit-identical whatever the schedule.
// Also measured on top of this form and not kept: the forward leaving every list entry's record in list order (+ its row index) on
// sparse frames, so that a unit reads one contiguous 1.5 KB instead of list slice -> record gathers (a dependent round trip less):
// 0.0628 against 0.0637 ms, forward +3 us; 16-entry units there: 0.0632, forward +9 us.  On a trained-like frame the kernel is bound by
// the VALU work of the SIMD that draws the most blended entries (p50 170, max 240 of ~455 issue cycles each), not by its loads.
// Built first and measured slower (tools/exp/render_bwd4_tickets_r5.hip.txt): atomic tickets over per-XCD lists
// with stealing -- (i) a persistent grid of 4 / 5 waves per SIMD pulling unit after unit, next ticket and descriptor prefetched:
// 0.354 / 0.415 ms on the initial scene (0.27 before), 0.126 / 0.163 trained-like (0.078): gfx9 has ONE in-order counter for loads
// and stores, so the next unit's first load waits for the previous unit's scattered 36-byte row stores to drain, which a wave that
// ends leaves to the memory system; (ii) one ticket per workgroup: 0.66 / 0.165 ms -- counts -> ticket -> descriptor are three
// more DEPENDENT memory round trips in front of every unit, and a round trip under this kernel's gather traffic costs 7-15 us
// (an atomic more).  Hence: no atomics here, every address from the workgroup id and two hot words.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RB4_WAVES_PER_EU, RB4_WAVES_PER_EU)))
render_bwd4_kernel(const uint2* __restrict__ ranges, const unsigned* __restrict__ point_list, int W, int H, int gridx,
                   const float* __restrict__ bg, const float* __restrict__ rec, const float4* __restrict__ cfin,
                   const float4* __restrict__ ckpt, const float4* __restrict__ ckpt64, const unsigned* __restrict__ n_contrib,
                   const float* __restrict__ dL_dpixels, float* __restrict__ slab,
                   uint8_t* __restrict__ live, const int ntiles, const int ulog, const unsigned* __restrict__ uctl,
                   const uint4* __restrict__ ulist_full, const uint4* __restrict__ ulist_last) {
    // staged splats, 48 bytes each: x, y, conic a * -log2(e)/2, conic b * -log2(e) | conic c * -log2(e)/2, opacity, r, g | b
    __shared__ float4 sR[64 * 3];
    __shared__ float sOut[64 * RB4_RS];
    const int lane = threadIdx.x;
    const size_t plane = (size_t)W * H;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
#if RB4_TRACE
    const unsigned long long tr_t0 = __builtin_readcyclecounter(), tr_w0 = wall_clock64();
    unsigned tr_blended = 0;
#endif
#if RB4_REDUCE == 1
    float bsel[9];
#pragma unroll
    for (int v = 0; v < 9; v++) bsel[v] = (lane & 15) == v ? 1.f : 0.f;
#endif

    // per-pixel constants of the tile in hand; pixel j of this lane = row pyb + 4 j, state index 64 j + lane
    Pair4 P[2];
    f2 full[2];        // C_final . dL/dC + T_final (bg . dL/dC): S' at the very front of the list
    f2 bgd[2];         // bg . dL/dC (the last unit starts from T_final, S' = T_final (bg . dL/dC))
    unsigned lc_top = 0u, lc_bot = 0u;
    float pxf = 0.f, tx0 = 0.f, ty0 = 0.f;
    unsigned tile_xu = 0u, tile_yu = 0u;

    // ---- this workgroup's unit (see above): x = id % 8, t = id / 8
    uint4 d;
    {
        const unsigned nf = uctl[0], nl = uctl[DGM_UCTL_LINE];
        const unsigned per_f = (nf + 7u) >> 3, per_l = (nl + 7u) >> 3;
        const unsigned x = blockIdx.x & 7u, t = blockIdx.x >> 3;
        if (t >= per_f + per_l) return;
        const bool is_last = t >= per_f;
        // (measured: consecutive units on consecutive XCDs -- u = 8 t + x -- instead of an eighth of the list per XCD: 0.0658 against
        // 0.0632 ms on the trained-like scene, 0.2764 against 0.2739 on the initial one; a Gaussian's rows then fill up in eight L2s)
        // (measured, round 6: an XCD's share of either list taken back to front -- the forward now finishes its longest tiles last, so
        // their units sit at the end -- 0.0644-0.0654 against 0.0648-0.0650 ms trained-like, 0.269-0.272 against 0.271-0.275 on the
        // initial scene: the order of the list does not matter to this kernel)
        const unsigned u = is_last ? x * per_l + (t - per_f) : x * per_f + t;
        if (u >= (is_last ? nl : nf)) return;
        d = is_last ? ulist_last[u] : ulist_full[u];
    }
    {
        const int tile = __builtin_amdgcn_readfirstlane((int)(d.x & 0x7fffffffu));
        const bool shortlist = (__builtin_amdgcn_readfirstlane((int)d.x) >> 31) != 0;
        const int k = __builtin_amdgcn_readfirstlane((int)d.y);
        const unsigned first = (unsigned)__builtin_amdgcn_readfirstlane((int)d.z);
        const int nproc = __builtin_amdgcn_readfirstlane((int)d.w);
        const int ulen = shortlist ? 1 << ulog : 256;
        const int nunits = max(1, (nproc + ulen - 1) / ulen);
        // (instances no pixel blends -- behind the tile's deepest contributor, outside the ellipse's reach, or skipped by every pixel --
        // keep the liveness byte 0 that render_fwd left: only live rows cost a store)
        if (nproc != 0) {
        const int seg_begin = k * ulen;
        const int seg_end = min(nproc, seg_begin + ulen);
        // hop 1 of the unit (with the pixel state below): the first batch's slice of the list
        const int pos0 = seg_end - 1 - lane;
        unsigned g0 = 0u;
        if (pos0 >= seg_begin) g0 = point_list[first + pos0];
        // ... and the blend state at the unit's back end (the last unit starts from the final state, which is part of the pixel state)
        const bool from_ckpt = k != nunits - 1;
        float4 ck[4];
        if (from_ckpt) {
            const float4* cp = shortlist ? ckpt64 + ((size_t)(first >> ulog) + (size_t)(k + 1)) * 256
                                         : ckpt + (size_t)((first + ((unsigned)(k + 1) << 8)) >> 8) * 256;
#pragma unroll
            for (int j = 0; j < 4; j++) ck[j] = cp[j * 64 + lane];
        }
        f2 Tf[2];
        {
            const int tile_x = tile % gridx, tile_y = tile / gridx;
            tile_xu = (unsigned)tile_x, tile_yu = (unsigned)tile_y;
            const int px = tile_x * DGM_TILE + (lane & 15);
            const int pyb = tile_y * DGM_TILE + (lane >> 4);  // rows pyb + {0, 4, 8, 12}
            pxf = (float)px;
            tx0 = (float)(tile_x * DGM_TILE), ty0 = (float)(tile_y * DGM_TILE);
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
                    const float bd = bg0 * dr[e] + bg1 * dg[e] + bg2 * db[e];
                    tf[e] = f.x * bd;
                    cf[e] = f.y * dr[e] + f.z * dg[e] + f.w * db[e] + tf[e];
                    if (e == 0) bgd[h].x = bd, Tf[h].x = f.x;
                    else bgd[h].y = bd, Tf[h].y = f.x;
                }
                P[h].dpr = (f2){dr[0], dr[1]};
                P[h].dpg = (f2){dg[0], dg[1]};
                P[h].dpb = (f2){db[0], db[1]};
                P[h].py = (f2){(float)(pyb + 8 * h), (float)(pyb + 8 * h + 4)};
                P[h].lc0 = lc[0], P[h].lc1 = lc[1];
                full[h] = (f2){cf[0], cf[1]};
            }
            // deepest contributor of each half tile (wave-uniform): list entries at or beyond it cannot touch that half
            lc_top = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_max_u32(max(P[0].lc0, P[0].lc1)));
            lc_bot = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_max_u32(max(P[1].lc0, P[1].lc1)));
        }
        // hop 2: the first batch's splat records, together with the checkpoint
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2v = q0;  // (q2v: colour b | rectangle | offs[g] | -)
        if (pos0 >= seg_begin) {
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g0 * DGM_REC_STRIDE);
            q0 = r4[0], q1 = r4[1], q2v = r4[2];
        }
        // replay state at the back end of the unit
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (from_ckpt) {
                const float4 c0 = ck[2 * h], c1 = ck[2 * h + 1];
                P[h].T = (f2){c0.x, c1.x};
                const f2 front = {c0.y * P[h].dpr.x + c0.z * P[h].dpg.x + c0.w * P[h].dpb.x,
                                  c1.y * P[h].dpr.y + c1.z * P[h].dpg.y + c1.w * P[h].dpb.y};
                P[h].S = full[h] - front;  // (C_final - C_front) . dL/dC + T_final (bg . dL/dC)
            } else {
                P[h].T = Tf[h];
                P[h].S = Tf[h] * bgd[h];
            }
        }
        // the first batch's gradient rows (the record is in by now: its loads were issued behind the checkpoint's)
        const float q2 = q2v.x;
        const unsigned row0 = instance_row(__float_as_uint(q2v.y), __float_as_uint(q2v.z), tile_xu, tile_yu);
        const int nb = (seg_end - seg_begin + 63) >> 6;
        for (int t = 0; t < nb; t++) {
            const int base_pos = seg_end - 1 - t * 64;  // list position staged by lane 0; lane l stages base_pos - l
            const int pos = base_pos - lane;
            unsigned qm = 0u, row = 0u;
            if (pos >= seg_begin) {
                float4 r0 = q0, r1 = q1;
                float r2x = q2;
                row = row0;
                if (t != 0) {  // (256-entry units of the long lists: later batches are fetched as they come)
                    const unsigned g = point_list[first + pos];
                    const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g * DGM_REC_STRIDE);
                    const float4 r2 = r4[2];
                    r0 = r4[0], r1 = r4[1], r2x = r2.x;
                    row = instance_row(__float_as_uint(r2.y), __float_as_uint(r2.z), tile_xu, tile_yu);  // the instance's gradient row
                }
                const float l2e = 1.4426950408889634f;
                sR[3 * lane] = make_float4(r0.x, r0.y, -0.5f * l2e * r0.z, -l2e * r0.w);
                sR[3 * lane + 1] = make_float4(-0.5f * l2e * r1.x, r1.y, r1.z, r1.w);
                sR[3 * lane + 2].x = r2x;
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
                if (is_live) {
                    live[row] = 1;
                    const float* o = sOut + lane * RB4_RS;
                    // row of the instance in the per-Gaussian order: the sum over a Gaussian's instances reads adjacent rows
                    float* dst = slab + (size_t)row * DGM_SLAB_STRIDE;
#if DGM_SLAB_STRIDE == 16  // (A/B, measured no faster: one whole, aligned 64-byte line per row -- 0.279 vs 0.274 ms init, 0.066 vs 0.065 trained, preprocess_bwd 0.067 vs 0.061)
                    float4* d4 = reinterpret_cast<float4*>(dst);
                    d4[0] = make_float4(o[0], o[1], o[2], o[3]);
                    d4[1] = make_float4(o[4], o[5], o[6], o[7]);
                    d4[2] = make_float4(o[8], 0.f, 0.f, 0.f);
                    d4[3] = make_float4(0.f, 0.f, 0.f, 0.f);
#else
                    *reinterpret_cast<dgm_f4u*>(dst) = (dgm_f4u){o[0], o[1], o[2], o[3]};
                    *reinterpret_cast<dgm_f4u*>(dst + 4) = (dgm_f4u){o[4], o[5], o[6], o[7]};
                    dst[8] = o[8];
#endif
                }
            }
        }
        }  // nproc != 0
    }
#if RB4_TRACE
    if (threadIdx.x == 0) {
        const unsigned long long tr_t1 = __builtin_readcyclecounter(), tr_w1 = wall_clock64();
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
        const unsigned long long i = atomicAdd(&rb4_trace[4 * 65536], 1ull) & 65535ull;
        // (s_memtime is a per-CU counter: durations only; start / end on the constant 100 MHz clock)
        rb4_trace[4 * i] = tr_w0, rb4_trace[4 * i + 1] = tr_w1, rb4_trace[4 * i + 2] = ((unsigned long long)xcc << 32) | hw;
        rb4_trace[4 * i + 3] = ((tr_t1 - tr_t0) << 32) | ((unsigned long long)(tr_blended & 0xffffu) << 16) | 1ull;
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

void launch_render_bwd4(hipStream_t st, int tiles, size_t R, const uint2* ranges, const unsigned* point_list, int W, int H,
                        int gridx, const float* bg, const float* rec, const float4* cfin, const float4* ckpt,
                        const float4* ckpt64, const unsigned* n_contrib, const float* dL_dpix, float* slab,
                        uint8_t* live, const unsigned* uctl, const uint4* ulist_full, const uint4* ulist_last) {
    const int ulog = replay_unit_log2(R);
    // one workgroup per possible unit: at most R / u full units and one last unit per tile; XCD x's share of either list is at most
    // ceil(count / 8) long, so 8 (ceil(R / u / 8) + ceil(tiles / 8)) ids cover every (x, t)
    const size_t grid = 8 * ((((R >> ulog) + 7) >> 3) + (((size_t)tiles + 7) >> 3));
    hipLaunchKernelGGL(render_bwd4_kernel, dim3((unsigned)grid), dim3(64), 0, st, ranges, point_list, W, H, gridx, bg, rec, cfin, ckpt,
                       ckpt64, n_contrib, dL_dpix, slab, live, tiles, ulog, uctl, ulist_full, ulist_last);
}

}  // namespace dgm
