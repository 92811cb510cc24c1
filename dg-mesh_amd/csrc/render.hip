// Tile alpha-compositing for gfx950: the forward blend (the backward replay lives in render_bwd4.hip).
//
// Replaces FORWARD::render / renderCUDA<3> (DGR/cuda_rasterizer/forward.cu:263-374).  Same tile size (16x16), same
// per-pixel arithmetic (power, alpha = min(.99, o*exp(power)), 1/255 and 1e-4 thresholds, n_contrib / final_T
// semantics), so `out_color`, `final_T`, `n_contrib` agree with the reference to rounding (tests: 1e-4; integer
// n_contrib exact away from the thresholds).
//
// MI355X design:
//  * one 256-thread workgroup per tile = 4 wave64, and each WAVE owns an 8x8 pixel quadrant (not 4 rows of
//    16): the 64 lanes of a wave are spatially compact, which makes wave-uniform decisions (early exit, culling)
//    effective;
//  * splats are staged 256 at a time into LDS from ONE 48-byte record per Gaussian (3 x 16 B gathers);
//  * while staging, each thread tests the ellipse {alpha >= 1/255} of its splat (q(d) <= 2 ln(255 o)) against the
//    four quadrants -- the minimum of the quadric over each quadrant's 8 x 8 block of pixel centres, not a bounding
//    box -- and four wave ballots turn that into one 64-bit mask per (staging wave, quadrant).  The blend loop of a wave is a
//    SCALAR loop over the set bits of its masks (s_ff1 / s_andn2), so pairs that the reference would discard
//    with `alpha < 1/255` after evaluating exp() are never issued.  The test is conservative (inflated box;
//    NaN => keep), hence results are unchanged.
#include <stdlib.h>
#include <string.h>

#include "dgm_common.hpp"
#include "render_common.hpp"

namespace dgm {

#ifndef RF_TRACE
#define RF_TRACE 0  // 1: every wave records (start, first blend, end, hardware id, entries staged / tested / rounds): tools/raster_bench.py --trace-fwd
#endif
#if RF_TRACE
__device__ unsigned long long rf_trace[4 * 65536 + 1];
#endif

template <bool SPARSE>
__global__ void __launch_bounds__(256)
render_fwd_kernel(const uint2* __restrict__ ranges, const unsigned* __restrict__ point_list, int W, int H, int gridx,
                  const float* __restrict__ rec, const float* __restrict__ bg, float* __restrict__ out_color,
                  float* __restrict__ final_T, unsigned* __restrict__ n_contrib, float4* __restrict__ ckpt,
                  float4* __restrict__ cfin, float4* __restrict__ ckpt64, unsigned* __restrict__ nproc_out, const int ulog,
                  unsigned* __restrict__ uctl, uint4* __restrict__ ulist_full, uint4* __restrict__ ulist_last,
                  uint8_t* __restrict__ live, const unsigned* __restrict__ tile_order) {
    // staged splats, 48 bytes each: x, y, conic a * -log2(e)/2, conic b * -log2(e) | conic c * -log2(e)/2, opacity, r, g | b
    // (one record per splat: the blend loop forms ONE address per entry for its three broadcast reads)
    __shared__ float4 sR[256 * 3];
    __shared__ unsigned long long sMask[4][4];  // [staging wave][quadrant]
    __shared__ unsigned sMaxC[4];               // per-wave maximum of last_contributor
    __shared__ unsigned sUnitBase;              // first slot of this tile's run of full units in the backward's work list
    const int tile = tile_order != nullptr ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;
    const int tile_x = tile % gridx, tile_y = tile / gridx;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int px = tile_x * DGM_TILE + (wv & 1) * 8 + (lane & 7);
    const int py = tile_y * DGM_TILE + (wv >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float tx0 = (float)(tile_x * DGM_TILE), ty0 = (float)(tile_y * DGM_TILE);
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const int rounds = (n + 255) >> 8;
#if RF_TRACE
    const unsigned long long tr_t0 = __builtin_readcyclecounter(), tr_w0 = wall_clock64();
    unsigned long long tr_t_first = 0;
    unsigned tr_tested = 0;
#endif
    // liveness bytes of the backward's per-instance gradient rows (render_bwd4.hip sets the ones it writes): the tiles' list
    // ranges partition the row index space, so each tile clears a stretch as long as its list
    for (int i = threadIdx.x; i < n; i += 256) live[range.x + i] = 0;
    // short lists: the blend state is left after every u = 64 entries (u = 32 on sparse frames, `ulog` = log2 u; slot
    // (first + u s) / u: unique per (tile, s)) instead of every 256, so that the backward can replay such a tile in u-entry
    // units on several waves (render_bwd4.hip)
    const bool shortlist = n <= DGM_SHORT_LIST;
    const int nsub = 256 >> ulog;  // checkpoint intervals per staged round
    const int lxy = (((wv >> 1) * 8 + (lane >> 3)) >> 2) * 64 + ((((wv >> 1) * 8 + (lane >> 3)) & 3) << 4) + (wv & 1) * 8 + (lane & 7);
    float4* const c64 = ckpt64 + (size_t)(range.x >> ulog) * 256 + lxy;  // + s * 256: (range.x + u s) >> ulog = (range.x >> ulog) + s

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    unsigned last_contributor = 0;
    unsigned long long done_m = __builtin_amdgcn_ballot_w64(!inside);  // lanes whose pixel is finished (wave-uniform lane mask)

    // SPARSE (frames with R < 2^20, picked by the launch like the unit length): a tile is one serial chain with about one wave per
    // SIMD to hide anything behind.  A round's staging is two dependent gathers (list slice -> records); they are software-pipelined: the next round's list entry is
    // asked for at the top of a round's blend loop and its record half way through (by then the entry has arrived), and the next
    // round starts from registers (trained-like 0.097 -> 0.092 ms).  It costs 19 registers -- 5 waves per SIMD instead of 6 -- which
    // a dense frame, VALU-bound with every slot busy, pays for (0.217 -> 0.221 ms): the launch picks by R, like the unit length.
    // (Also tried for sparse frames and dropped: four list entries per trip blended speculatively -- the running products formed
    // for all four as if nothing stopped, the stop tests and contributions selected afterwards, bit-identical results without the
    // ballot -> scalar mask -> inverse ballot round trip per entry: 0.102 ms against 0.091, and 0.266 against 0.218 on a dense frame.
    // The per-entry decision chain is not what a trained tile waits for.  Round 5, same verdict for the LDS reads: the first 16 bytes
    // of the NEXT pair's records read while the current pair blends -- 0.105 ms against 0.093.)
    unsigned g_next = 0u;
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0;
    float ncb = 0.f;
    bool fetched = false;  // (workgroup-uniform)
    for (int i = 0; i < rounds; i++) {
        if (__syncthreads_and(done_m == ~0ull)) break;  // also orders the previous round's LDS reads before the refill
        if (i > 0 && !shortlist) {
            // state after the first 256 i list entries, for the segment-parallel backward (render_bwd4.hip): slot
            // floor((range.x + 256 i) / 256) is unique per (tile, i); pixel order = the backward's lane mapping
            // (row = 4 j + (l >> 4), column = l & 15  ->  index 64 j + l)
            const int lx = (wv & 1) * 8 + (lane & 7), ly = (wv >> 1) * 8 + (lane >> 3);
            const size_t slot = (size_t)((range.x + ((unsigned)i << 8)) >> 8);
            ckpt[slot * 256 + (size_t)((ly >> 2) * 64 + ((ly & 3) << 4) + lx)] = make_float4(T, C0, C1, C2);
        }
        const int at = (i << 8) + threadIdx.x;
        unsigned qm = 0;
        if (at < n) {
            float4 r0 = n0, r1 = n1;
            float cb = ncb;
            if (!fetched) {
                const unsigned g = point_list[range.x + at];
                const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g * DGM_REC_STRIDE);
                r0 = r4[0], r1 = r4[1], cb = r4[2].x;
            }
            // the conic is staged pre-multiplied so that the exponent below comes out times log2(e), ready for v_exp_f32
            // (same sign as the reference's `power`; render_bwd4 stages the same way)
            const float l2e = 1.4426950408889634f;
            sR[3 * threadIdx.x] = make_float4(r0.x, r0.y, -0.5f * l2e * r0.z, -l2e * r0.w);
            sR[3 * threadIdx.x + 1] = make_float4(-0.5f * l2e * r1.x, r1.y, r1.z, r1.w);
            sR[3 * threadIdx.x + 2].x = cb;
            qm = quadrant_mask(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, tx0, ty0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned long long bal = __ballot((qm >> q) & 1u);
            if (lane == 0) sMask[wv][q] = bal;
        }
        __syncthreads();
        const unsigned base = (unsigned)(i << 8);
        const int at_next = ((i + 1) << 8) + (int)threadIdx.x;
        fetched = SPARSE && i + 1 < rounds;
        if (fetched && at_next < n) g_next = point_list[range.x + at_next];
        if (done_m == ~0ull) {  // whole quadrant finished: keep helping with staging only
            if (shortlist)
                for (int sb = 0; sb < nsub; sb++) {
                    const int s = nsub * i + sb + 1;
                    if ((s << ulog) < n) c64[(size_t)s * 256] = make_float4(T, C0, C1, C2);
                }
            if (fetched && at_next < n) {
                const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g_next * DGM_REC_STRIDE);
                n0 = r4[0], n1 = r4[1], ncb = r4[2].x;
            }
            continue;
        }
#pragma unroll 1
        for (int sb = 0; sb < nsub; sb++) {
            if (sb == (nsub >> 1) && fetched && at_next < n) {  // (the list entry asked for at the top has arrived by now)
                const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g_next * DGM_REC_STRIDE);
                n0 = r4[0], n1 = r4[1], ncb = r4[2].x;
            }
            const int sw = sb >> (6 - ulog);  // staging wave = 64-entry block of the round
            unsigned long long m = sMask[sw][wv];
            if (ulog < 6) m &= ((1ull << (1 << ulog)) - 1ull) << ((sb & ((1 << (6 - ulog)) - 1)) << ulog);  // this sub-block's entries
            m = uniform_u64(m);
            // Two list entries per trip: the second one's geometry (exponent, exp, alpha) does not depend on the first one's
            // blend, so a lone wave -- a trained scene leaves about one per SIMD -- overlaps the two dependency chains instead of
            // waiting out each instruction's latency; the blends themselves stay in list order.  (Odd counts: the last entry is
            // evaluated twice and blended once.)
            while (m) {
#if RF_TRACE
                if (tr_t_first == 0) tr_t_first = __builtin_readcyclecounter();
                tr_tested += (unsigned)__popcll(m) >= 2u ? 2u : 1u;
#endif
                const int ja = (sw << 6) + __builtin_ctzll(m);
                m &= m - 1;
                const bool two = m != 0ull;
                const int jb = two ? (sw << 6) + __builtin_ctzll(m) : ja;
                m &= m - 1;  // (no-op when m == 0)
                const float4 Aa = sR[3 * ja], Ab = sR[3 * jb];
                const float4 Ba = sR[3 * ja + 1], Bb = sR[3 * jb + 1];
                const float ca = sR[3 * ja + 2].x, cbb = sR[3 * jb + 2].x;
                const float dxa = Aa.x - pxf, dya = Aa.y - pyf;
                const float dxb = Ab.x - pxf, dyb = Ab.y - pyf;
                const float power_a = (Aa.z * dxa + Aa.w * dya) * dxa + (Ba.x * dya) * dya;  // the reference's exponent times log2(e)
                const float power_b = (Ab.z * dxb + Ab.w * dyb) * dxb + (Bb.x * dyb) * dyb;
                const float alpha_a = fminf(0.99f, Ba.y * __builtin_amdgcn_exp2f(power_a));
                const float alpha_b = fminf(0.99f, Bb.y * __builtin_amdgcn_exp2f(power_b));
                // the blend decisions as 64-bit lane masks in scalar registers (ballot / inverse ballot): "done", "passes",
                // "stops here" and "blends" combine with scalar and / andn2 instead of a second vector compare each
                // (one ballot per compare, combined in scalar registers: a ballot of a compound condition is lowered through a
                // select and a second compare)
                const unsigned long long geo_a = __builtin_amdgcn_ballot_w64(!(power_a > 0.0f)) & __builtin_amdgcn_ballot_w64(!(alpha_a < 1.0f / 255.0f));
                const unsigned long long geo_b = __builtin_amdgcn_ballot_w64(!(power_b > 0.0f)) & __builtin_amdgcn_ballot_w64(!(alpha_b < 1.0f / 255.0f)) &
                                                 (two ? ~0ull : 0ull);
                {
                    const unsigned long long pass = geo_a & ~done_m;
                    const float test_T = T * (1.0f - alpha_a);
                    const unsigned long long low = __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
                    const unsigned long long valid_m = pass & ~low;
                    done_m |= pass & low;
                    const bool valid = __builtin_amdgcn_inverse_ballot_w64(valid_m);
                    // branch-free: a pixel that skips the splat blends it with weight zero
                    const float w = valid ? alpha_a * T : 0.f;
                    C0 += Ba.z * w;
                    C1 += Ba.w * w;
                    C2 += ca * w;
                    T = valid ? test_T : T;
                    last_contributor = valid ? base + (unsigned)ja + 1u : last_contributor;
                }
                {
                    const unsigned long long pass = geo_b & ~done_m;
                    const float test_T = T * (1.0f - alpha_b);
                    const unsigned long long low = __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
                    const unsigned long long valid_m = pass & ~low;
                    done_m |= pass & low;
                    const bool valid = __builtin_amdgcn_inverse_ballot_w64(valid_m);
                    const float w = valid ? alpha_b * T : 0.f;
                    C0 += Bb.z * w;
                    C1 += Bb.w * w;
                    C2 += cbb * w;
                    T = valid ? test_T : T;
                    last_contributor = valid ? base + (unsigned)jb + 1u : last_contributor;
                }
            }
            if (shortlist) {
                const int s = nsub * i + sb + 1;
                if ((s << ulog) < n) c64[(size_t)s * 256] = make_float4(T, C0, C1, C2);
            }
        }
    }
    {   // per-tile bound of the backward replay: the deepest contributor index of any pixel
        const unsigned m = wave_max_u32(inside ? last_contributor : 0u);
        if (lane == 0) sMaxC[wv] = m;
        __syncthreads();
        const unsigned np = min(max(max(sMaxC[0], sMaxC[1]), max(sMaxC[2], sMaxC[3])), (unsigned)n);
        if (threadIdx.x == 0) nproc_out[tile] = np;
        // ... and the tile's entries of the backward's work list (render_bwd4.hip): it is replayed in units of u list entries
        // (256 beyond DGM_SHORT_LIST), all of them full except the last; a tile with entries but nothing to replay keeps one unit
        // (its rows have to be marked dead).  Tiles are listed in the order they finish: a tile's full units as one contiguous run
        // of ulist_full, its last unit in ulist_last (the backward starts the long units first); two counters, each on its own
        // 128-byte line.  Record: (tile | short flag, unit, first slot, bound).
        if (n > 0) {
            const unsigned nunits = max(1u, shortlist ? (np + (1u << ulog) - 1u) >> ulog : (np + 255u) >> 8);
            const unsigned tag = (unsigned)tile | (shortlist ? 0x80000000u : 0u);
            if (threadIdx.x == 0) {
                sUnitBase = nunits > 1u ? atomicAdd(&uctl[0], nunits - 1u) : 0u;
                const unsigned bl = atomicAdd(&uctl[DGM_UCTL_LINE], 1u);
                ulist_last[bl] = make_uint4(tag, nunits - 1u, range.x, np);
            }
            __syncthreads();
            uint4* dst = ulist_full + sUnitBase;
            for (unsigned k = threadIdx.x; k + 1u < nunits; k += 256u) dst[k] = make_uint4(tag, k, range.x, np);
        }
    }
    {   // final state (T, C without background) in the backward's pixel order, for every lane of the tile
        const int lx = (wv & 1) * 8 + (lane & 7), ly = (wv >> 1) * 8 + (lane >> 3);
        cfin[(size_t)tile * 256 + (size_t)((ly >> 2) * 64 + ((ly & 3) << 4) + lx)] = make_float4(T, C0, C1, C2);
    }
    if (inside) {
        const size_t pid = (size_t)W * py + px;
        const size_t plane = (size_t)W * H;
        final_T[pid] = T;
        n_contrib[pid] = last_contributor;
        out_color[pid] = C0 + T * bg[0];
        out_color[plane + pid] = C1 + T * bg[1];
        out_color[2 * plane + pid] = C2 + T * bg[2];
    }
#if RF_TRACE
    if (lane == 0) {
        const unsigned long long tr_t1 = __builtin_readcyclecounter(), tr_w1 = wall_clock64();
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
        const unsigned long long i = atomicAdd(&rf_trace[4 * 65536], 1ull) & 65535ull;
        // start / end on the constant 100 MHz clock; | xcc, hw id | duration (shader cycles) << 32, cycles before the first blend | list
        // length << 32, entries this wave tested << 8, wave
        rf_trace[4 * i] = (tr_w0 & ((1ull << 48) - 1ull)) | ((tr_w1 - tr_w0) << 48);
        rf_trace[4 * i + 1] = ((unsigned long long)xcc << 32) | hw;
        rf_trace[4 * i + 2] = ((tr_t1 - tr_t0) << 32) | ((tr_t_first ? tr_t_first - tr_t0 : 0ull) & 0xffffffffull);
        rf_trace[4 * i + 3] = ((unsigned long long)(unsigned)n << 32) | ((unsigned long long)tr_tested << 8) | (unsigned)wv;
    }
#endif
}
// ---- sparse frames (R < 2^20), round 6: the four quadrant waves of a tile run ASYNCHRONOUSLY ------------------------------------------
// render_fwd_kernel<true> stages 256 list entries per round for the whole tile and its four quadrant waves meet at two barriers a
// round: the per-wave trace of round 5 (profiles/r05_render_fwd_trace.txt) showed a trained-like frame bound by a few hundred long
// tiles in which every wave spends ~500 cycles per entry it tests -- ~250 for the blend, the rest waiting at the round's barrier for
// the tile's busiest quadrant -- so a tile costs the SUM over rounds of the per-round maxima.  Here every wave stages for itself:
// 64 entries a round into its own 3 KB of LDS (the list slice and the records of the NEXT round are in flight while the current one
// blends), tested against its own 8 x 8 block only, no barrier inside the loop.  A tile then costs the maximum over its quadrants of
// their own total work.  The price is four reads of the tile's list and records instead of one (L2 hits; a sparse frame leaves the
// memory system idle).  Same per-pixel arithmetic, same order of the blends, same outputs as the kernel above -- bit-identical.
// Measured on the trained-like scene (tools/raster_bench.py cfg2 --kind trained, one MI355X): 0.095 ms -> 0.082 ms, and 0.069 ms with
// the tiles handed out longest first (tile_order, written by tile_scan_kernel).  The trace of THIS kernel (-DRF_TRACE,
// profiles/r06_render_fwd_async_trace.txt) shows what is left: all 2 500 workgroups are resident from the first microsecond, a wave's
// end time follows the work of the SIMD it sits on (correlation 0.81 with the sum of entries tested by the waves sharing its SIMD),
// i.e. the kernel is bound by instruction issue per SIMD times the imbalance between SIMDs, not by memory.  (Tried on top and
// dropped: the NEXT pair's geometry -- LDS reads, exponent, exp, alpha, threshold ballots -- formed while the current pair blends, a
// hand-made software pipeline of the bit loop: 0.090 ms, the loop's live state doubles and the scheduler serialises it anyway;
// records and list slices fetched one round further ahead: no change; render_fwd_ring_kernel below: 0.090 ms.)
// Checkpoints: a wave writes its pixels' state at the unit boundaries it passes; a quadrant whose pixels have all terminated leaves the
// loop, and after the tile's one barrier (which yields the replay bound) fills in the boundaries up to that bound with its final state
// -- exactly the set the backward reads.
__global__ void __launch_bounds__(256)
render_fwd_async_kernel(const uint2* __restrict__ ranges, const unsigned* __restrict__ point_list, int W, int H, int gridx,
                        const float* __restrict__ rec, const float* __restrict__ bg, float* __restrict__ out_color,
                        float* __restrict__ final_T, unsigned* __restrict__ n_contrib, float4* __restrict__ ckpt,
                        float4* __restrict__ cfin, float4* __restrict__ ckpt64, unsigned* __restrict__ nproc_out, const int ulog,
                        unsigned* __restrict__ uctl, uint4* __restrict__ ulist_full, uint4* __restrict__ ulist_last,
                        uint8_t* __restrict__ live, const unsigned* __restrict__ tile_order) {
    __shared__ float4 sRw[4][64 * 3];  // per wave: the round's 64 staged splats (layout as in render_fwd_kernel)
    __shared__ unsigned sMaxC[4];
    __shared__ unsigned sUnitBase;
    // (tiles in descending order of their list length, tile_scan_kernel: every workgroup of a sparse frame is resident at once, so
    // what balances the CUs is the order in which the tiles are handed out)
    const int tile = tile_order != nullptr ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;
    const int tile_x = tile % gridx, tile_y = tile / gridx;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int px = tile_x * DGM_TILE + (wv & 1) * 8 + (lane & 7);
    const int py = tile_y * DGM_TILE + (wv >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float qx0 = (float)(tile_x * DGM_TILE + (wv & 1) * 8), qy0 = (float)(tile_y * DGM_TILE + (wv >> 1) * 8);
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const int rounds = (n + 63) >> 6;
#if RF_TRACE
    const unsigned long long tr_t0 = __builtin_readcyclecounter(), tr_w0 = wall_clock64();
    unsigned long long tr_t_first = 0, tr_t_loop_end = 0;
    unsigned tr_tested = 0;
#endif
    for (int i = threadIdx.x; i < n; i += 256) live[range.x + i] = 0;
    const bool shortlist = n <= DGM_SHORT_LIST;
    const int ulen_log = shortlist ? ulog : 8;          // checkpoint interval: u entries, 256 beyond DGM_SHORT_LIST
    const int lxy = (((wv >> 1) * 8 + (lane >> 3)) >> 2) * 64 + ((((wv >> 1) * 8 + (lane >> 3)) & 3) << 4) + (wv & 1) * 8 + (lane & 7);
    // boundary s (the state after s << ulen_log entries) lives at ...
    float4* const cbase = shortlist ? ckpt64 + (size_t)(range.x >> ulog) * 256 + lxy
                                    : ckpt + (size_t)(range.x >> 8) * 256 + lxy;  // (+ s * 256; the long form: slot (range.x + 256 s) >> 8)
    float4* const sR = sRw[wv];

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    unsigned last_contributor = 0;
    unsigned long long done_m = __builtin_amdgcn_ballot_w64(!inside);
    int s_next = 1;  // next boundary this wave has not written

    // Staging is two dependent gathers (list slice -> records).  A quadrant that culls most of a round's entries has little work
    // to hide them behind -- its rounds then cost the memory round trips, not the blend -- so both are asked for well ahead: at the
    // top of round i the records of round i + 2 (their list entries were asked for a round earlier) and the list slice of round i + 3.
    // (First version: the next round's list slice at the top of a round and its records half way through: 0.079-0.086 ms on the
    // trained-like scene; two, or four, entries per trip made no difference -- the rounds were waiting for memory.)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 n0 = z4, n1 = z4, m0 = z4, m1 = z4;   // n: records of the round about to be staged, m: of the round behind it
    float ncb = 0.f, mcb = 0.f;
    unsigned g3 = 0u;                            // list entry of the round two behind it
    {
        unsigned ga = 0u, gb = 0u;
        if (lane < n) ga = point_list[range.x + lane];
        if (64 + lane < n) gb = point_list[range.x + 64 + lane];
        if (128 + lane < n) g3 = point_list[range.x + 128 + lane];
        if (lane < n) {
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)ga * DGM_REC_STRIDE);
            n0 = r4[0], n1 = r4[1], ncb = r4[2].x;
        }
        if (64 + lane < n) {
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)gb * DGM_REC_STRIDE);
            m0 = r4[0], m1 = r4[1], mcb = r4[2].x;
        }
    }
    for (int i = 0; i < rounds; i++) {
        if (done_m == ~0ull) break;  // every pixel of this quadrant has terminated (the boundaries left are filled in below)
        const int at = (i << 6) + lane;
        bool hit = false;
        if (at < n) {
            const float l2e = 1.4426950408889634f;
            sR[3 * lane] = make_float4(n0.x, n0.y, -0.5f * l2e * n0.z, -l2e * n0.w);
            sR[3 * lane + 1] = make_float4(-0.5f * l2e * n1.x, n1.y, n1.z, n1.w);
            sR[3 * lane + 2].x = ncb;
            hit = quadrant_hit(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, qx0, qy0);
        }
        const unsigned long long mask = uniform_u64(__ballot(hit));
        const unsigned base = (unsigned)(i << 6);
        n0 = m0, n1 = m1, ncb = mcb;  // (staged: the next round's records move up)
        if (((i + 2) << 6) + lane < n) {
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g3 * DGM_REC_STRIDE);
            m0 = r4[0], m1 = r4[1], mcb = r4[2].x;
        }
        if (((i + 3) << 6) + lane < n) g3 = point_list[range.x + ((i + 3) << 6) + lane];
        const int nsub = 64 >> (ulen_log < 6 ? ulen_log : 6);  // checkpoint intervals inside a round (2 at u = 32, else 1)
#pragma unroll 1
        for (int sb = 0; sb < nsub; sb++) {
            unsigned long long m = mask;
            if (nsub == 2) m &= sb == 0 ? 0xffffffffull : 0xffffffff00000000ull;
            while (m) {
#if RF_TRACE
                if (tr_t_first == 0) tr_t_first = __builtin_readcyclecounter();
                tr_tested += (unsigned)__popcll(m) >= 2u ? 2u : 1u;
#endif
                const int ja = __builtin_ctzll(m);
                m &= m - 1;
                const bool two = m != 0ull;
                const int jb = two ? __builtin_ctzll(m) : ja;
                m &= m - 1;
                const float4 Aa = sR[3 * ja], Ab = sR[3 * jb];
                const float4 Ba = sR[3 * ja + 1], Bb = sR[3 * jb + 1];
                const float ca = sR[3 * ja + 2].x, cbb = sR[3 * jb + 2].x;
                const float dxa = Aa.x - pxf, dya = Aa.y - pyf;
                const float dxb = Ab.x - pxf, dyb = Ab.y - pyf;
                const float power_a = (Aa.z * dxa + Aa.w * dya) * dxa + (Ba.x * dya) * dya;
                const float power_b = (Ab.z * dxb + Ab.w * dyb) * dxb + (Bb.x * dyb) * dyb;
                const float alpha_a = fminf(0.99f, Ba.y * __builtin_amdgcn_exp2f(power_a));
                const float alpha_b = fminf(0.99f, Bb.y * __builtin_amdgcn_exp2f(power_b));
                const unsigned long long geo_a = __builtin_amdgcn_ballot_w64(!(power_a > 0.0f)) & __builtin_amdgcn_ballot_w64(!(alpha_a < 1.0f / 255.0f));
                const unsigned long long geo_b = __builtin_amdgcn_ballot_w64(!(power_b > 0.0f)) & __builtin_amdgcn_ballot_w64(!(alpha_b < 1.0f / 255.0f)) &
                                                 (two ? ~0ull : 0ull);
                {
                    const unsigned long long pass = geo_a & ~done_m;
                    const float test_T = T * (1.0f - alpha_a);
                    const unsigned long long low = __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
                    const unsigned long long valid_m = pass & ~low;
                    done_m |= pass & low;
                    const bool valid = __builtin_amdgcn_inverse_ballot_w64(valid_m);
                    const float w = valid ? alpha_a * T : 0.f;
                    C0 += Ba.z * w;
                    C1 += Ba.w * w;
                    C2 += ca * w;
                    T = valid ? test_T : T;
                    last_contributor = valid ? base + (unsigned)ja + 1u : last_contributor;
                }
                {
                    const unsigned long long pass = geo_b & ~done_m;
                    const float test_T = T * (1.0f - alpha_b);
                    const unsigned long long low = __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
                    const unsigned long long valid_m = pass & ~low;
                    done_m |= pass & low;
                    const bool valid = __builtin_amdgcn_inverse_ballot_w64(valid_m);
                    const float w = valid ? alpha_b * T : 0.f;
                    C0 += Bb.z * w;
                    C1 += Bb.w * w;
                    C2 += cbb * w;
                    T = valid ? test_T : T;
                    last_contributor = valid ? base + (unsigned)jb + 1u : last_contributor;
                }
            }
            // the boundary behind this sub-block: entries processed so far = 64 i + 32 (sb + 1) (u = 32) or 64 (i + 1)
            const int processed = (i << 6) + ((sb + 1) << (nsub == 2 ? 5 : 6));
            if ((processed & ((1 << ulen_log) - 1)) == 0 && processed < n) {
                cbase[(size_t)(processed >> ulen_log) * 256] = make_float4(T, C0, C1, C2);
                s_next = (processed >> ulen_log) + 1;
            }
        }
    }
#if RF_TRACE
    tr_t_loop_end = __builtin_readcyclecounter();
#endif
    {
        const unsigned mx = wave_max_u32(inside ? last_contributor : 0u);
        if (lane == 0) sMaxC[wv] = mx;
        __syncthreads();
        const unsigned np = min(max(max(sMaxC[0], sMaxC[1]), max(sMaxC[2], sMaxC[3])), (unsigned)n);
        if (threadIdx.x == 0) nproc_out[tile] = np;
        const unsigned ulen = 1u << ulen_log;
        const unsigned nunits = n > 0 ? max(1u, (np + ulen - 1u) >> ulen_log) : 0u;
        // boundaries 1 .. nunits - 1 are read by the backward: the ones this wave did not reach get its final state (it left the loop
        // with every pixel terminated -- or at the end of the list, where nothing is missing)
        for (int s = s_next; s < (int)nunits; s++) cbase[(size_t)s * 256] = make_float4(T, C0, C1, C2);
        if (n > 0) {
            const unsigned tag = (unsigned)tile | (shortlist ? 0x80000000u : 0u);
            if (threadIdx.x == 0) {
                sUnitBase = nunits > 1u ? atomicAdd(&uctl[0], nunits - 1u) : 0u;
                const unsigned bl = atomicAdd(&uctl[DGM_UCTL_LINE], 1u);
                ulist_last[bl] = make_uint4(tag, nunits - 1u, range.x, np);
            }
            __syncthreads();
            uint4* dst = ulist_full + sUnitBase;
            for (unsigned k = threadIdx.x; k + 1u < nunits; k += 256u) dst[k] = make_uint4(tag, k, range.x, np);
        }
    }
    {
        const int lx = (wv & 1) * 8 + (lane & 7), ly = (wv >> 1) * 8 + (lane >> 3);
        cfin[(size_t)tile * 256 + (size_t)((ly >> 2) * 64 + ((ly & 3) << 4) + lx)] = make_float4(T, C0, C1, C2);
    }
    if (inside) {
        const size_t pid = (size_t)W * py + px;
        const size_t plane = (size_t)W * H;
        final_T[pid] = T;
        n_contrib[pid] = last_contributor;
        out_color[pid] = C0 + T * bg[0];
        out_color[plane + pid] = C1 + T * bg[1];
        out_color[2 * plane + pid] = C2 + T * bg[2];
    }
#if RF_TRACE
    if (lane == 0) {
        const unsigned long long tr_t1 = __builtin_readcyclecounter(), tr_w1 = wall_clock64();
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
        const unsigned long long i = atomicAdd(&rf_trace[4 * 65536], 1ull) & 65535ull;
        // as render_fwd_kernel's record, with the loop's share of the lifetime in the otherwise unused bits 32..47 of word 1 (units of 16 cycles)
        rf_trace[4 * i] = (tr_w0 & ((1ull << 48) - 1ull)) | ((tr_w1 - tr_w0) << 48);
        rf_trace[4 * i + 1] = ((unsigned long long)(xcc & 15u) << 32) | hw | ((((tr_t_loop_end - tr_t0) >> 4) & 0xffffull) << 36);
        rf_trace[4 * i + 2] = ((tr_t1 - tr_t0) << 32) | ((tr_t_first ? tr_t_first - tr_t0 : 0ull) & 0xffffffffull);
        rf_trace[4 * i + 3] = ((unsigned long long)(unsigned)n << 32) | ((unsigned long long)tr_tested << 8) | (unsigned)wv;
    }
#endif
}

// ---- sparse frames, second form (round 6): shared staging WITHOUT the lock step -------------------------------------------------------
// render_fwd_async_kernel removed the barriers by letting every quadrant wave stage the whole list for itself; its trace
// (tools/raster_bench.py --trace-fwd on a -DRF_TRACE=1 build) shows what that costs: a busy wave spends ~370 cycles per LIST entry --
// not per entry it blends --, i.e. ~23 k cycles per 64-entry round, waiting for its own gathers of 48-byte records (64 distinct lines
// per load instruction, ten waves per SIMD doing the same: the CU's address unit, not the blend, sets the pace), and every record is
// gathered four times.  Here the tile's four waves share the staging again -- wave w gathers rounds w, w + 4, w + 8, ... into a ring
// of RING slots in LDS and tests them against all four quadrants -- but nobody waits at a barrier: a round is published through a
// per-wave counter in LDS, a consumer takes round c as soon as wave c % 4 has published it, and a stager refills a slot as soon as the
// slowest consumer has left it.  A quadrant may run up to RING - 1 rounds ahead of the slowest one: a tile again costs the maximum
// over its quadrants of their own work, with the gather traffic of the barrier form.  A wave whose pixels have all terminated keeps
// serving its staging duty until every wave of the tile is through.  Every wait is bounded (a wave that spins too long raises a flag
// in the frame's counter words and leaves: the frame is then wrong, but the GPU is not hung).  Same arithmetic, same blend order,
// same outputs as the two kernels above, bit for bit; checkpoints as in the asynchronous form.
#ifndef RF_RING
#define RF_RING 8
#endif
__global__ void __launch_bounds__(256)
render_fwd_ring_kernel(const uint2* __restrict__ ranges, const unsigned* __restrict__ point_list, int W, int H, int gridx,
                       const float* __restrict__ rec, const float* __restrict__ bg, float* __restrict__ out_color,
                       float* __restrict__ final_T, unsigned* __restrict__ n_contrib, float4* __restrict__ ckpt,
                       float4* __restrict__ cfin, float4* __restrict__ ckpt64, unsigned* __restrict__ nproc_out, const int ulog,
                       unsigned* __restrict__ uctl, uint4* __restrict__ ulist_full, uint4* __restrict__ ulist_last,
                       uint8_t* __restrict__ live) {
    constexpr int RING = RF_RING;
    static_assert(RING % 4 == 0 && RING >= 4, "a slot belongs to one staging wave");
    __shared__ float4 sRing[RING][64 * 3];
    __shared__ unsigned long long sMaskQ[RING][4];  // [slot][quadrant]
    __shared__ int sStaged[4], sConsumed[4], sFinished;
    __shared__ unsigned sMaxC[4];
    __shared__ unsigned sUnitBase;
    const int tile = blockIdx.x;
    const int tile_x = tile % gridx, tile_y = tile / gridx;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int px = tile_x * DGM_TILE + (wv & 1) * 8 + (lane & 7);
    const int py = tile_y * DGM_TILE + (wv >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float tx0 = (float)(tile_x * DGM_TILE), ty0 = (float)(tile_y * DGM_TILE);
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const int rounds = (n + 63) >> 6;
    if (threadIdx.x < 4) sStaged[threadIdx.x] = 0, sConsumed[threadIdx.x] = 0;
    if (threadIdx.x == 0) sFinished = 0;
    for (int i = threadIdx.x; i < n; i += 256) live[range.x + i] = 0;
    const bool shortlist = n <= DGM_SHORT_LIST;
    const int ulen_log = shortlist ? ulog : 8;
    const int lxy = (((wv >> 1) * 8 + (lane >> 3)) >> 2) * 64 + ((((wv >> 1) * 8 + (lane >> 3)) & 3) << 4) + (wv & 1) * 8 + (lane & 7);
    float4* const cbase = shortlist ? ckpt64 + (size_t)(range.x >> ulog) * 256 + lxy : ckpt + (size_t)(range.x >> 8) * 256 + lxy;
    volatile int* const vStaged = sStaged;
    volatile int* const vConsumed = sConsumed;
    volatile int* const vFinished = &sFinished;

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    unsigned last_contributor = 0;
    unsigned long long done_m = __builtin_amdgcn_ballot_w64(!inside);
    int s_next = 1;

    // this wave's staging duty: rounds wv, wv + 4, ...; the records of the next two duties and the list slice of the third are in
    // flight (as in the asynchronous form)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 n0 = z4, n1 = z4, m0 = z4, m1 = z4;
    float ncb = 0.f, mcb = 0.f;
    unsigned g3 = 0u;
    {
        unsigned ga = 0u, gb = 0u;
        const int a0 = (wv << 6) + lane, a1 = ((wv + 4) << 6) + lane, a2 = ((wv + 8) << 6) + lane;
        if (a0 < n) ga = point_list[range.x + a0];
        if (a1 < n) gb = point_list[range.x + a1];
        if (a2 < n) g3 = point_list[range.x + a2];
        if (a0 < n) {
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)ga * DGM_REC_STRIDE);
            n0 = r4[0], n1 = r4[1], ncb = r4[2].x;
        }
        if (a1 < n) {
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)gb * DGM_REC_STRIDE);
            m0 = r4[0], m1 = r4[1], mcb = r4[2].x;
        }
    }
    __syncthreads();  // the counters are initialised

    int c = 0;            // next round this wave consumes
    int sp = wv;          // next round this wave stages
    bool active = true;   // still consuming (wave-uniform)
    int spins = 0;
    while (true) {
        if (!active && *vFinished >= 4) break;  // every quadrant is through: nobody left to stage for
        // ---- 1. staging duty: as far ahead as the ring allows
        if (sp < rounds) {
            const int cmin = min(min(vConsumed[0], vConsumed[1]), min(vConsumed[2], vConsumed[3]));
            if (sp < cmin + RING) {
                const int slot = sp & (RING - 1);
                float4* const sR = sRing[slot];
                const int at = (sp << 6) + lane;
                unsigned qm = 0u;
                if (at < n) {
                    const float l2e = 1.4426950408889634f;
                    sR[3 * lane] = make_float4(n0.x, n0.y, -0.5f * l2e * n0.z, -l2e * n0.w);
                    sR[3 * lane + 1] = make_float4(-0.5f * l2e * n1.x, n1.y, n1.z, n1.w);
                    sR[3 * lane + 2].x = ncb;
                    qm = quadrant_mask(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, tx0, ty0);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const unsigned long long bal = __ballot((qm >> q) & 1u);
                    if (lane == 0) sMaskQ[slot][q] = bal;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) vStaged[wv] = (sp >> 2) + 1;
                n0 = m0, n1 = m1, ncb = mcb;
                if (((sp + 8) << 6) + lane < n) {
                    const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g3 * DGM_REC_STRIDE);
                    m0 = r4[0], m1 = r4[1], mcb = r4[2].x;
                }
                if (((sp + 12) << 6) + lane < n) g3 = point_list[range.x + ((sp + 12) << 6) + lane];
                sp += 4;
                spins = 0;
                continue;
            }
        }
        // ---- 2. consume the next round, if its stager has published it
        if (active) {
            if (c >= rounds || done_m == ~0ull) {
                active = false;
                if (lane == 0) {
                    vConsumed[wv] = 0x3fffffff;  // (never holds a stager back again)
                    atomicAdd(&sFinished, 1);
                }
                continue;
            }
            if (vStaged[c & 3] > (c >> 2)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const int slot = c & (RING - 1);
                const float4* const sR = sRing[slot];
                const unsigned long long mask = uniform_u64(sMaskQ[slot][wv]);
                const unsigned base = (unsigned)(c << 6);
                const int nsub = 64 >> (ulen_log < 6 ? ulen_log : 6);
#pragma unroll 1
                for (int sb = 0; sb < nsub; sb++) {
                    unsigned long long m = mask;
                    if (nsub == 2) m &= sb == 0 ? 0xffffffffull : 0xffffffff00000000ull;
                    while (m) {
                        const int ja = __builtin_ctzll(m);
                        m &= m - 1;
                        const bool two = m != 0ull;
                        const int jb = two ? __builtin_ctzll(m) : ja;
                        m &= m - 1;
                        const float4 Aa = sR[3 * ja], Ab = sR[3 * jb];
                        const float4 Ba = sR[3 * ja + 1], Bb = sR[3 * jb + 1];
                        const float ca = sR[3 * ja + 2].x, cbb = sR[3 * jb + 2].x;
                        const float dxa = Aa.x - pxf, dya = Aa.y - pyf;
                        const float dxb = Ab.x - pxf, dyb = Ab.y - pyf;
                        const float power_a = (Aa.z * dxa + Aa.w * dya) * dxa + (Ba.x * dya) * dya;
                        const float power_b = (Ab.z * dxb + Ab.w * dyb) * dxb + (Bb.x * dyb) * dyb;
                        const float alpha_a = fminf(0.99f, Ba.y * __builtin_amdgcn_exp2f(power_a));
                        const float alpha_b = fminf(0.99f, Bb.y * __builtin_amdgcn_exp2f(power_b));
                        const unsigned long long geo_a = __builtin_amdgcn_ballot_w64(!(power_a > 0.0f)) & __builtin_amdgcn_ballot_w64(!(alpha_a < 1.0f / 255.0f));
                        const unsigned long long geo_b = __builtin_amdgcn_ballot_w64(!(power_b > 0.0f)) & __builtin_amdgcn_ballot_w64(!(alpha_b < 1.0f / 255.0f)) &
                                                         (two ? ~0ull : 0ull);
                        {
                            const unsigned long long pass = geo_a & ~done_m;
                            const float test_T = T * (1.0f - alpha_a);
                            const unsigned long long low = __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
                            const unsigned long long valid_m = pass & ~low;
                            done_m |= pass & low;
                            const bool valid = __builtin_amdgcn_inverse_ballot_w64(valid_m);
                            const float w = valid ? alpha_a * T : 0.f;
                            C0 += Ba.z * w;
                            C1 += Ba.w * w;
                            C2 += ca * w;
                            T = valid ? test_T : T;
                            last_contributor = valid ? base + (unsigned)ja + 1u : last_contributor;
                        }
                        {
                            const unsigned long long pass = geo_b & ~done_m;
                            const float test_T = T * (1.0f - alpha_b);
                            const unsigned long long low = __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
                            const unsigned long long valid_m = pass & ~low;
                            done_m |= pass & low;
                            const bool valid = __builtin_amdgcn_inverse_ballot_w64(valid_m);
                            const float w = valid ? alpha_b * T : 0.f;
                            C0 += Bb.z * w;
                            C1 += Bb.w * w;
                            C2 += cbb * w;
                            T = valid ? test_T : T;
                            last_contributor = valid ? base + (unsigned)jb + 1u : last_contributor;
                        }
                    }
                    const int processed = (c << 6) + ((sb + 1) << (nsub == 2 ? 5 : 6));
                    if ((processed & ((1 << ulen_log) - 1)) == 0 && processed < n) {
                        cbase[(size_t)(processed >> ulen_log) * 256] = make_float4(T, C0, C1, C2);
                        s_next = (processed >> ulen_log) + 1;
                    }
                }
                c++;
                if (lane == 0) vConsumed[wv] = c;  // (behind this wave's reads of the slot: LDS operations of a wave complete in order)
                spins = 0;
                continue;
            }
        } else if (sp >= rounds) {
            break;  // through, and nothing left to stage
        }
        // ---- 3. nothing to do right now
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) {  // (~seconds: a protocol error must not hang the device)
            if (lane == 0) atomicOr(&uctl[1], 1u);
            break;
        }
    }
    {
        const unsigned mx = wave_max_u32(inside ? last_contributor : 0u);
        if (lane == 0) sMaxC[wv] = mx;
        __syncthreads();
        const unsigned np = min(max(max(sMaxC[0], sMaxC[1]), max(sMaxC[2], sMaxC[3])), (unsigned)n);
        if (threadIdx.x == 0) nproc_out[tile] = np;
        const unsigned ulen = 1u << ulen_log;
        const unsigned nunits = n > 0 ? max(1u, (np + ulen - 1u) >> ulen_log) : 0u;
        for (int s = s_next; s < (int)nunits; s++) cbase[(size_t)s * 256] = make_float4(T, C0, C1, C2);
        if (n > 0) {
            const unsigned tag = (unsigned)tile | (shortlist ? 0x80000000u : 0u);
            if (threadIdx.x == 0) {
                sUnitBase = nunits > 1u ? atomicAdd(&uctl[0], nunits - 1u) : 0u;
                const unsigned bl = atomicAdd(&uctl[DGM_UCTL_LINE], 1u);
                ulist_last[bl] = make_uint4(tag, nunits - 1u, range.x, np);
            }
            __syncthreads();
            uint4* dst = ulist_full + sUnitBase;
            for (unsigned k = threadIdx.x; k + 1u < nunits; k += 256u) dst[k] = make_uint4(tag, k, range.x, np);
        }
    }
    {
        const int lx = (wv & 1) * 8 + (lane & 7), ly = (wv >> 1) * 8 + (lane >> 3);
        cfin[(size_t)tile * 256 + (size_t)((ly >> 2) * 64 + ((ly & 3) << 4) + lx)] = make_float4(T, C0, C1, C2);
    }
    if (inside) {
        const size_t pid = (size_t)W * py + px;
        const size_t plane = (size_t)W * H;
        final_T[pid] = T;
        n_contrib[pid] = last_contributor;
        out_color[pid] = C0 + T * bg[0];
        out_color[plane + pid] = C1 + T * bg[1];
        out_color[2 * plane + pid] = C2 + T * bg[2];
    }
}

#if RF_TRACE
}  // namespace dgm
extern "C" int dgm_debug_rf_trace(void* dst, size_t bytes, int reset) {
    int e = (int)hipDeviceSynchronize();
    if (!e) e = (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(dgm::rf_trace), bytes);
    unsigned long long z = 0;
    if (!e && reset) e = (int)hipMemcpyToSymbol(HIP_SYMBOL(dgm::rf_trace), &z, 8, 4 * 65536 * 8);
    return e;
}
namespace dgm {
#endif

void launch_render_fwd(hipStream_t st, int tiles, const uint2* ranges, const unsigned* point_list, int W, int H,
                       int gridx, const float* rec, const float* bg, float* out_color, float* final_T,
                       unsigned* n_contrib, float4* ckpt, float4* cfin, float4* ckpt64, unsigned* nproc, size_t R, unsigned* uctl,
                       uint4* ulist_full, uint4* ulist_last, uint8_t* live, const unsigned* tile_order) {
    const int ulog = replay_unit_log2(R);
    static const bool no_order = [] { const char* e = getenv("DGM_RF_ORDER"); return e && strcmp(e, "0") == 0; }();
    // sparse frames: the asynchronous-quadrant kernel (DGM_RF_SPARSE=sync selects round 5's barrier-per-round form, for A/B runs)
    static const bool sparse_sync = [] { const char* e = getenv("DGM_RF_SPARSE"); return e && strcmp(e, "sync") == 0; }();
    static const bool sparse_ring = [] { const char* e = getenv("DGM_RF_SPARSE"); return e && strcmp(e, "ring") == 0; }();
    if (ulog < 6 && sparse_ring)
        hipLaunchKernelGGL(render_fwd_ring_kernel, dim3(tiles), dim3(256), 0, st, ranges, point_list, W, H, gridx, rec, bg,
                           out_color, final_T, n_contrib, ckpt, cfin, ckpt64, nproc, ulog, uctl, ulist_full, ulist_last, live);
    else if (ulog < 6 && !sparse_sync)
        hipLaunchKernelGGL(render_fwd_async_kernel, dim3(tiles), dim3(256), 0, st, ranges, point_list, W, H, gridx, rec, bg,
                           out_color, final_T, n_contrib, ckpt, cfin, ckpt64, nproc, ulog, uctl, ulist_full, ulist_last, live,
                           no_order ? (const unsigned*)nullptr : tile_order);
    else if (ulog < 6)
        hipLaunchKernelGGL(render_fwd_kernel<true>, dim3(tiles), dim3(256), 0, st, ranges, point_list, W, H, gridx, rec, bg,
                           out_color, final_T, n_contrib, ckpt, cfin, ckpt64, nproc, ulog, uctl, ulist_full, ulist_last, live,
                           (const unsigned*)nullptr);
    else
        hipLaunchKernelGGL(render_fwd_kernel<false>, dim3(tiles), dim3(256), 0, st, ranges, point_list, W, H, gridx, rec, bg,
                           out_color, final_T, n_contrib, ckpt, cfin, ckpt64, nproc, ulog, uctl, ulist_full, ulist_last, live,
                           no_order ? (const unsigned*)nullptr : tile_order);
}

}  // namespace dgm
