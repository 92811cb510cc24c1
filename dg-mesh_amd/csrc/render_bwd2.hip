// render_bwd, second generation: TWO pixels per lane.
//
// Replays BACKWARD::render / renderCUDA<3> (DGR/cuda_rasterizer/backward.cu:401-557) and writes one 48-byte gradient
// row per (tile, splat) instance into the slab; mapping of the 16x16 tile onto the machine:
//   * 128-thread workgroup = 2 waves; wave w owns the 16 x 8 half tile of rows [8w, 8w+8); lane l owns the two pixels
//     (col = l & 15, row = 8w + (l >> 4)) and (col, row + 4): same column, so dx and a*dx*dx are shared;
//   * the per-pixel arithmetic is written on 2-vectors and compiles to v_pk_{mul,add,fma}_f32 -- the packed fp32 ops
//     that make up half of the chip's fp32 VALU peak -- so one instruction advances two (pixel, splat) pairs;
//   * the expensive per-(wave, splat) tail -- 9-value cross-lane reduction (DPP) + LDS accumulation -- is now paid
//     once per 128 pixel pairs instead of once per 64.
// Culling works on the two half tiles (ballot masks + scalar bit loop as before).
#include "dgm_common.hpp"
#include "render_common.hpp"

namespace dgm {

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned half_mask(float x, float y, float a, float b, float c, float o, float tx0, float ty0) {
    float ex, ey;
    if (!alpha_extent(a, b, c, o, ex, ey)) return 0u;
    if ((x + ex < tx0) || (x - ex > tx0 + 15.0f)) return 0u;
    unsigned m = 0;
    if (!((y + ey < ty0) || (y - ey > ty0 + 7.0f))) m |= 1u;
    if (!((y + ey < ty0 + 8.0f) || (y - ey > ty0 + 15.0f))) m |= 2u;
    return m;
}

static constexpr int RB = 128;   // splats staged per round (one per thread)
static constexpr int RS = 9;     // floats per per-wave accumulator row

__global__ void __launch_bounds__(128)
render_bwd2_kernel(const uint2* __restrict__ ranges, const unsigned* __restrict__ point_list, int W, int H, int gridx,
                   const float* __restrict__ bg, const float* __restrict__ rec, const float* __restrict__ final_Ts,
                   const unsigned* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
                   float* __restrict__ slab, unsigned* __restrict__ nproc_out) {
    __shared__ float4 sA[RB];  // x, y, conic a, conic b
    __shared__ float4 sB[RB];  // conic c, opacity, r, g
    __shared__ float sC[RB];   // b
    __shared__ float sAcc[2][RB * RS];          // per-wave partial gradient rows: plain stores, no atomics
    __shared__ unsigned long long sMask[2][2];  // [staging wave][half tile]
    __shared__ unsigned sMax[2];
    const int tile = blockIdx.x;
    const int tile_x = tile % gridx, tile_y = tile / gridx;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int px = tile_x * DGM_TILE + (lane & 15);
    const int py0 = tile_y * DGM_TILE + wv * 8 + (lane >> 4), py1 = py0 + 4;
    const bool in0 = px < W && py0 < H, in1 = px < W && py1 < H;
    const float pxf = (float)px;
    const f2 pyf = {(float)py0, (float)py1};
    const float tx0 = (float)(tile_x * DGM_TILE), ty0 = (float)(tile_y * DGM_TILE);
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const size_t plane = (size_t)W * H;
    const size_t pid0 = (size_t)W * py0 + px, pid1 = (size_t)W * py1 + px;

    const f2 T_final = {in0 ? final_Ts[pid0] : 0.f, in1 ? final_Ts[pid1] : 0.f};
    const unsigned lc0 = in0 ? n_contrib[pid0] : 0u, lc1 = in1 ? n_contrib[pid1] : 0u;
    f2 dpr = {0.f, 0.f}, dpg = {0.f, 0.f}, dpb = {0.f, 0.f};
    if (in0) {
        dpr.x = dL_dpixels[pid0];
        dpg.x = dL_dpixels[plane + pid0];
        dpb.x = dL_dpixels[2 * plane + pid0];
    }
    if (in1) {
        dpr.y = dL_dpixels[pid1];
        dpg.y = dL_dpixels[plane + pid1];
        dpb.y = dL_dpixels[2 * plane + pid1];
    }
    const f2 tf_bg = T_final * (bg[0] * dpr + bg[1] * dpg + bg[2] * dpb);  // T_final * (bg . dL_dpixel)
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;                   // backward.cu:460-461

    {
        const unsigned m = wave_max_u32(lc0 > lc1 ? lc0 : lc1);
        if (lane == 0) sMax[wv] = m;
    }
    __syncthreads();
    int nproc = (int)max(sMax[0], sMax[1]);
    nproc = min(nproc, n);
    if (threadIdx.x == 0) nproc_out[tile] = (unsigned)nproc;
    const int rounds = (nproc + RB - 1) / RB;

    // Replay state.  With S_ch = sum over the splats BEHIND the current one of c_ch * alpha * T (un-normalised), the
    // reference's recurrence (accum_rec / last_alpha / last_color, backward.cu:515-534) reads
    //     dL/dalpha = sum_ch (c_ch * T_j - S_ch / (1 - alpha)) * dL/dC_ch  -  T_final / (1 - alpha) * (bg . dL/dC)
    // and a skipped pair is simply alpha = 0 (T, S unchanged): no per-state selects are needed.
    f2 T = T_final;
    f2 Sr = {0.f, 0.f}, Sg = {0.f, 0.f}, Sb = {0.f, 0.f};

    for (int i = 0; i < rounds; i++) {
        __syncthreads();
        const int pos = nproc - 1 - (i * RB + (int)threadIdx.x);  // list position, back to front
        unsigned qm = 0u;
        if (pos >= 0) {
            const unsigned g = point_list[range.x + pos];
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g * DGM_REC_STRIDE);
            const float4 r0 = r4[0], r1 = r4[1];
            sA[threadIdx.x] = r0;
            sB[threadIdx.x] = r1;
            sC[threadIdx.x] = r4[2].x;
            qm = half_mask(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, tx0, ty0);
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const unsigned long long bal = __ballot((qm >> q) & 1u);
            if (lane == 0) sMask[wv][q] = bal;
        }
        __syncthreads();
        const int base_pos = nproc - 1 - i * RB;
        float* acc_w = sAcc[wv];
#pragma unroll 1
        for (int sw = 0; sw < 2; sw++) {
            unsigned long long m = uniform_u64(sMask[sw][wv]);
            while (m) {
                const int j = (sw << 6) + __builtin_ctzll(m);
                m &= m - 1;
                const float4 A = sA[j];
                const float4 B = sB[j];
                const unsigned cidx = (unsigned)(base_pos - j);  // contributor index (backward.cu:486-488)
                const float dx = A.x - pxf;
                const f2 dy = A.y - pyf;
                const float adx2 = A.z * dx * dx, bdx = A.w * dx;
                const f2 power = -0.5f * (adx2 + B.x * dy * dy) - bdx * dy;
                f2 G;
                G.x = fast_exp(power.x);
                G.y = fast_exp(power.y);
                f2 alpha = B.y * G;
                alpha.x = fminf(0.99f, alpha.x);
                alpha.y = fminf(0.99f, alpha.y);
                const bool v0 = in0 && cidx < lc0 && !(power.x > 0.0f) && !(alpha.x < 1.0f / 255.0f);
                const bool v1 = in1 && cidx < lc1 && !(power.y > 0.0f) && !(alpha.y < 1.0f / 255.0f);
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f, s8 = 0.f;
                if (__ballot(v0 || v1) != 0ull) {
                    const float cb = sC[j];
                    const f2 vm = {v0 ? 1.f : 0.f, v1 ? 1.f : 0.f};
                    alpha = alpha * vm;  // skipped pair == alpha 0
                    const f2 one_m_a = 1.f - alpha;
                    f2 inv;
                    inv.x = __builtin_amdgcn_rcpf(one_m_a.x);
                    inv.y = __builtin_amdgcn_rcpf(one_m_a.y);
                    const f2 Tn = T * inv;      // transmittance in front of this splat
                    const f2 w = alpha * Tn;    // dC/dcolor
                    f2 dL_dalpha = (B.z * Tn - Sr * inv) * dpr + (B.w * Tn - Sg * inv) * dpg + (cb * Tn - Sb * inv) * dpb;
                    dL_dalpha = (dL_dalpha - tf_bg * inv) * vm;
                    Sr += B.z * w;
                    Sg += B.w * w;
                    Sb += cb * w;
                    T = Tn;
                    const f2 dL_dG = B.y * dL_dalpha;
                    const f2 gdx = G * dx, gdy = G * dy;
                    const f2 dG_ddelx = -gdx * A.z - gdy * A.w;
                    const f2 dG_ddely = -gdy * B.x - gdx * A.w;
                    const f2 q0 = w * dpr, q1 = w * dpg, q2 = w * dpb;
                    const f2 q3 = dL_dG * dG_ddelx * ddelx_dx, q4 = dL_dG * dG_ddely * ddely_dy;
                    const f2 hg = -0.5f * dL_dG;
                    const f2 q5 = hg * gdx * dx, q6 = hg * gdx * dy, q7 = hg * gdy * dy;
                    const f2 q8 = G * dL_dalpha;
                    s0 = q0.x + q0.y, s1 = q1.x + q1.y, s2 = q2.x + q2.y, s3 = q3.x + q3.y, s4 = q4.x + q4.y;
                    s5 = q5.x + q5.y, s6 = q6.x + q6.y, s7 = q7.x + q7.y, s8 = q8.x + q8.y;
                    const float r = wave_reduce8t(s0, s1, s2, s3, s4, s5, s6, s7, lane);
                    s8 = wave_reduce1_lane63(s8);
                    s0 = r;
                }
                // row layout: dcolor r,g,b | dmean2D x,y | dconic a,b,c | dopacity  (zeros when nothing contributed)
                if (lane < 8) acc_w[j * RS + lane] = s0;
                if (lane == 63) acc_w[j * RS + 8] = s8;
            }
        }
        __syncthreads();
        if (pos >= 0) {
            // sum the waves that were assigned this entry, in fixed order (deterministic)
            const int e = threadIdx.x;
            const bool h0 = (sMask[wv][0] >> lane) & 1ull, h1 = (sMask[wv][1] >> lane) & 1ull;
            float o[RS];
#pragma unroll
            for (int k = 0; k < RS; k++) {
                float v = 0.f;
                if (h0) v += sAcc[0][e * RS + k];
                if (h1) v += sAcc[1][e * RS + k];
                o[k] = v;
            }
            float4* dst = reinterpret_cast<float4*>(slab + (size_t)(range.x + pos) * DGM_SLAB_STRIDE);
            dst[0] = make_float4(o[0], o[1], o[2], o[3]);
            dst[1] = make_float4(o[4], o[5], o[6], o[7]);
            dst[2] = make_float4(o[8], 0.f, 0.f, 0.f);
        }
    }
}

void launch_render_bwd2(hipStream_t st, int tiles, const uint2* ranges, const unsigned* point_list, int W, int H,
                        int gridx, const float* bg, const float* rec, const float* final_T, const unsigned* n_contrib,
                        const float* dL_dpix, float* slab, unsigned* nproc) {
    hipLaunchKernelGGL(render_bwd2_kernel, dim3(tiles), dim3(128), 0, st, ranges, point_list, W, H, gridx, bg, rec,
                       final_T, n_contrib, dL_dpix, slab, nproc);
}

}  // namespace dgm
