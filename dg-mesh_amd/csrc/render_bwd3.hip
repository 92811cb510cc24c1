// render_bwd, third generation: ONE WAVE PER (TILE, SEGMENT), four pixels per lane, moment accumulation.
//
// Replays BACKWARD::render / renderCUDA<3> (DGR/cuda_rasterizer/backward.cu:401-557) and writes one 48-byte row per
// (tile, splat) instance into the slab.  What changed against render_bwd2 (two waves per tile, two pixels per lane):
//   * the forward pass now leaves, per tile, the blend state (T, C) of its 256 pixels after every 256 list entries
//     (render.hip: `ckpt`) and the final state (`cfin`).  With them every 256-entry SEGMENT of a tile's list can be
//     replayed independently: at the back end of segment k the transmittance is T_{k+1} and the colour accumulated
//     behind it is C_final - C_{k+1}.  The grid is (tiles, KSPLIT) single-wave workgroups; workgroup (t, s) takes the
//     segments s, s + KSPLIT, ... of tile t -- long lists spread over several CUs, no barriers anywhere;
//   * one wave covers the whole 16 x 16 tile: lane l owns column l & 15 and the rows (l >> 4) + {0, 4, 8, 12}; the per-pixel
//     arithmetic runs on 2-vectors (v_pk_*_f32) for the row pairs {r, r+4} (upper half tile) and {r+8, r+12} (lower half),
//     each skipped when the splat's alpha >= 1/255 box misses that half.  The 9-value cross-lane reduction and the row
//     write are paid once per 256 pixel pairs instead of once per 128;
//   * per pair only the MOMENTS of g = G dL/dalpha about the splat centre are accumulated (1, dx, dy, dx^2, dx dy, dy^2)
//     next to the three colour sums: dL/dmean2D, dL/dconic and dL/dopacity are linear in them with per-Gaussian
//     coefficients, which preprocess_bwd applies once per Gaussian after its gather;
//   * the recurrence carries ONE scalar per pixel, S' = sum_ch S_ch dL/dC_ch + T_final (bg . dL/dC), instead of three
//     colour sums:  dL/dalpha = ((c . dL/dC) T - S') / (1 - alpha),  S' += (c . dL/dC) alpha T / (1 - alpha).
// No atomics, bit-reproducible (fixed reduction and summation order).
#include "dgm_common.hpp"
#include "render_common.hpp"

namespace dgm {

typedef float f2 __attribute__((ext_vector_type(2)));

static constexpr int RB3_SEG = 256;    // list entries per segment (= one staging round of the forward pass)
static constexpr int RB3_KSPLIT = 8;   // workgroups per tile; segments are dealt round-robin
static constexpr int RB3_RS = 9;       // floats per staged output row

// bit 0: the alpha >= 1/255 box reaches rows 0..7 of the tile, bit 1: rows 8..15
__device__ __forceinline__ unsigned half_mask3(float x, float y, float a, float b, float c, float o, float tx0, float ty0) {
    float ex, ey;
    if (!alpha_extent(a, b, c, o, ex, ey)) return 0u;
    if ((x + ex < tx0) || (x - ex > tx0 + 15.0f)) return 0u;
    unsigned m = 0;
    if (!((y + ey < ty0) || (y - ey > ty0 + 7.0f))) m |= 1u;
    if (!((y + ey < ty0 + 8.0f) || (y - ey > ty0 + 15.0f))) m |= 2u;
    return m;
}

// one row pair of one splat: updates the pair's replay state and adds its nine partial sums
struct PairState {
    f2 T, S;          // transmittance behind the current splat; S' (see header)
    f2 dpr, dpg, dpb; // dL/dC of the two pixels
    f2 py;            // pixel rows
    unsigned lc0, lc1;  // 0 for pixels outside the image
};

__device__ __forceinline__ bool blend_pair(PairState& p, const float4 A, const float4 B, const float cb, const float dx,
                                           const float adx2, const float bdx, const unsigned cidx, f2 (&q)[9]) {
    // A.z, A.w, B.x hold the conic pre-multiplied by -log2(e)/2, -log2(e), -log2(e)/2 (staging), so `power` is the
    // reference's exponent times log2(e): same sign, and G = 2^power
    const f2 dy = A.y - p.py;
    const f2 power = (B.x * dy) * dy + adx2 + bdx * dy;
    f2 G;
    G.x = __builtin_amdgcn_exp2f(power.x);
    G.y = __builtin_amdgcn_exp2f(power.y);
    f2 alpha = B.y * G;
    alpha.x = fminf(0.99f, alpha.x);
    alpha.y = fminf(0.99f, alpha.y);
    // (pixels outside the image have lc = 0)
    const bool v0 = cidx < p.lc0 && !(power.x > 0.0f) && !(alpha.x < 1.0f / 255.0f);
    const bool v1 = cidx < p.lc1 && !(power.y > 0.0f) && !(alpha.y < 1.0f / 255.0f);
    if (__ballot(v0 || v1) == 0ull) return false;  // nothing contributes in this half tile: state and sums unchanged
    const f2 vm = {v0 ? 1.f : 0.f, v1 ? 1.f : 0.f};
    alpha = alpha * vm;  // a skipped pair is alpha = 0: T, S' stay, all partial sums get zero
    const f2 one_m_a = 1.f - alpha;
    f2 inv;
    inv.x = __builtin_amdgcn_rcpf(one_m_a.x);
    inv.y = __builtin_amdgcn_rcpf(one_m_a.y);
    const f2 Tn = p.T * inv;   // transmittance in front of this splat
    const f2 w = alpha * Tn;   // dC/dcolor
    const f2 cdp = B.z * p.dpr + B.w * p.dpg + cb * p.dpb;
    const f2 dL_dalpha = (p.T * cdp - p.S) * inv * vm;
    p.S += cdp * w;
    p.T = Tn;
    const f2 g = G * dL_dalpha;
    const f2 gdx = g * dx, gdy = g * dy;
    q[0] += w * p.dpr;
    q[1] += w * p.dpg;
    q[2] += w * p.dpb;
    q[3] += gdx;
    q[4] += gdy;
    q[5] += gdx * dx;
    q[6] += gdx * dy;
    q[7] += gdy * dy;
    q[8] += g;
    return true;
}

__global__ void __launch_bounds__(64)
render_bwd3_kernel(const uint2* __restrict__ ranges, const unsigned* __restrict__ point_list, int W, int H, int gridx,
                   const float* __restrict__ bg, const float* __restrict__ rec, const float4* __restrict__ cfin,
                   const float4* __restrict__ ckpt, const unsigned* __restrict__ n_contrib,
                   const float* __restrict__ dL_dpixels, const unsigned* __restrict__ nproc_in,
                   const unsigned* __restrict__ upos, float* __restrict__ slab) {
    __shared__ float4 sA[64];  // x, y, conic a * -log2(e)/2, conic b * -log2(e)
    __shared__ float4 sB[64];  // conic c * -log2(e)/2, opacity, r, g
    __shared__ float sC[64];   // b
    __shared__ float sOut[64 * RB3_RS];
    const int tile = blockIdx.x;
    const uint2 range = ranges[tile];
    const int nproc = (int)nproc_in[tile];
    const int nseg = (nproc + RB3_SEG - 1) / RB3_SEG;
    // list entries behind the deepest contributor of the tile are never replayed: their rows are zeros (the per-Gaussian
    // sum of preprocess_bwd reads every row); the tile's workgroups share them
    for (int pos = nproc + (int)blockIdx.y * 64 + (int)threadIdx.x; pos < (int)(range.y - range.x); pos += RB3_KSPLIT * 64) {
        float4* dst = reinterpret_cast<float4*>(slab + (size_t)upos[range.x + pos] * DGM_SLAB_STRIDE);
        dst[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        dst[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        dst[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if ((int)blockIdx.y >= nseg) return;
    const int tile_x = tile % gridx, tile_y = tile / gridx;
    const int lane = threadIdx.x;
    const int px = tile_x * DGM_TILE + (lane & 15);
    const int pyb = tile_y * DGM_TILE + (lane >> 4);  // rows pyb + {0, 4, 8, 12}
    const float pxf = (float)px;
    const float tx0 = (float)(tile_x * DGM_TILE), ty0 = (float)(tile_y * DGM_TILE);
    const size_t plane = (size_t)W * H;

    // per-pixel constants; pixel j of this lane = row pyb + 4 j, state index 64 j + lane (the forward's checkpoint order)
    PairState P[2];
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

    for (int k = (int)blockIdx.y; k < nseg; k += RB3_KSPLIT) {
        const int seg_begin = k * RB3_SEG;
        const int seg_end = min(nproc, seg_begin + RB3_SEG);
        // replay state at the back end of the segment
        if (k == nseg - 1) {
#pragma unroll
            for (int h = 0; h < 2; h++) P[h].T = Tfin[h], P[h].S = Sfin[h];
        } else {
            const size_t slot = (size_t)((range.x + ((unsigned)(k + 1) << 8)) >> 8);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float4 c0 = ckpt[slot * 256 + (2 * h) * 64 + lane], c1 = ckpt[slot * 256 + (2 * h + 1) * 64 + lane];
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
                row = upos[range.x + pos];  // fetched with the splat: the dependent row store below does not wait for it
                const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g * DGM_REC_STRIDE);
                const float4 r0 = r4[0], r1 = r4[1];
                const float l2e = 1.4426950408889634f;
                sA[lane] = make_float4(r0.x, r0.y, -0.5f * l2e * r0.z, -l2e * r0.w);
                sB[lane] = make_float4(-0.5f * l2e * r1.x, r1.y, r1.z, r1.w);
                sC[lane] = r4[2].x;
                qm = half_mask3(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, tx0, ty0);
            }
            const unsigned long long m_top = uniform_u64(__ballot(qm & 1u));
            const unsigned long long m_bot = uniform_u64(__ballot(qm & 2u));
            if (qm == 0u) {  // rows of splats that miss the tile (or of idle lanes) are zeros
#pragma unroll
                for (int i = 0; i < RB3_RS; i++) sOut[lane * RB3_RS + i] = 0.f;
            }
            unsigned long long m = m_top | m_bot;
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                const float4 A = sA[j];
                const float4 B = sB[j];
                const float cb = sC[j];
                const unsigned cidx = (unsigned)(base_pos - j);  // contributor index (backward.cu:486-488)
                const float dx = A.x - pxf;
                const float adx2 = A.z * dx * dx, bdx = A.w * dx;
                f2 q[9];
#pragma unroll
                for (int i = 0; i < 9; i++) q[i] = (f2){0.f, 0.f};
                bool any = false;
                if (((m_top >> j) & 1ull) && cidx < lc_top) any |= blend_pair(P[0], A, B, cb, dx, adx2, bdx, cidx, q);
                if (((m_bot >> j) & 1ull) && cidx < lc_bot) any |= blend_pair(P[1], A, B, cb, dx, adx2, bdx, cidx, q);
                float r = 0.f, r8 = 0.f;
                if (any) {  // wave-uniform: splats inside the box that no pixel blends get a zero row without the reduction
                    r = wave_reduce8t(q[0].x + q[0].y, q[1].x + q[1].y, q[2].x + q[2].y, q[3].x + q[3].y, q[4].x + q[4].y,
                                      q[5].x + q[5].y, q[6].x + q[6].y, q[7].x + q[7].y, lane);
                    r8 = wave_reduce1_lane63(q[8].x + q[8].y);
                }
                // row: colour r,g,b | moments dx, dy | dx^2, dx dy, dy^2 | 1
                if (lane < 8) sOut[j * RB3_RS + lane] = r;
                if (lane == 63) sOut[j * RB3_RS + 8] = r8;
            }
            if (pos >= seg_begin) {
                const float* o = sOut + lane * RB3_RS;
                // row of the instance in the per-Gaussian order: the sum over a Gaussian's instances reads adjacent rows
                float4* dst = reinterpret_cast<float4*>(slab + (size_t)row * DGM_SLAB_STRIDE);
                dst[0] = make_float4(o[0], o[1], o[2], o[3]);
                dst[1] = make_float4(o[4], o[5], o[6], o[7]);
                dst[2] = make_float4(o[8], 0.f, 0.f, 0.f);
            }
        }
    }
}

void launch_render_bwd3(hipStream_t st, int tiles, const uint2* ranges, const unsigned* point_list, int W, int H,
                        int gridx, const float* bg, const float* rec, const float4* cfin, const float4* ckpt,
                        const unsigned* n_contrib, const float* dL_dpix, const unsigned* nproc, const unsigned* upos,
                        float* slab) {
    hipLaunchKernelGGL(render_bwd3_kernel, dim3(tiles, RB3_KSPLIT), dim3(64), 0, st, ranges, point_list, W, H, gridx, bg, rec,
                       cfin, ckpt, n_contrib, dL_dpix, nproc, upos, slab);
}

}  // namespace dgm
