// Tile alpha-compositing for gfx950: forward blend and its backward replay.
//
// Replaces FORWARD::render / renderCUDA<3> (DGR/cuda_rasterizer/forward.cu:263-374) and BACKWARD::render /
// renderCUDA<3> (DGR/cuda_rasterizer/backward.cu:401-557).  Same tile size (16x16), same per-pixel
// arithmetic (power, alpha = min(.99, o*exp(power)), 1/255 and 1e-4 thresholds, n_contrib / final_T
// semantics), so `out_color`, `final_T`, `n_contrib` and all gradients agree with the reference to
// rounding (tests: 1e-4; integer n_contrib exact away from the thresholds).
//
// MI355X design:
//  * one 256-thread workgroup per tile = 4 wave64, and each WAVE owns an 8x8 pixel quadrant (not 4 rows of
//    16): the 64 lanes of a wave are spatially compact, which makes wave-uniform decisions effective;
//  * splats are staged 256 at a time into LDS from ONE 48-byte record per Gaussian (3 x 16 B gathers);
//  * while staging, each thread computes the exact screen-space bounding box of "alpha >= 1/255" for its
//    splat (half-extent sqrt(2 ln(255 o) * Sigma_xx|yy)) and tests it against the four quadrants; four
//    wave ballots turn that into one 64-bit mask per (staging wave, quadrant).  The blend loop of a wave is a
//    SCALAR loop over the set bits of its masks (s_ff1 / s_andn2), so pairs that the reference would discard
//    with `alpha < 1/255` after evaluating exp() are never issued.  The test is conservative (inflated box;
//    NaN => keep), hence results are unchanged;
//  * backward: no global atomics.  The 9 partial gradients of a (pixel, splat) pair are summed across the
//    wave with DPP row_shr / row_bcast adds (no LDS traffic), the 4 waves combine through ds_add_f32 into a
//    per-batch LDS accumulator, and each (tile, splat) instance writes ONE 48-byte row into a slab indexed by
//    its slot in the tile list (coalesced).  A later per-Gaussian kernel gathers its rows through the
//    inverse map, which also makes the gradient summation order deterministic (the reference's is not);
//  * backward replays only the first max(n_contrib) instances of the tile instead of the whole list.
#include <stdlib.h>

#include "dgm_common.hpp"
#include "render_common.hpp"

namespace dgm {

template <bool CULL, int MODE = 0>
__global__ void __launch_bounds__(256)
render_fwd_kernel(const uint2* __restrict__ ranges, const unsigned* __restrict__ point_list, int W, int H, int gridx,
                  const float* __restrict__ rec, const float* __restrict__ bg, float* __restrict__ out_color,
                  float* __restrict__ final_T, unsigned* __restrict__ n_contrib) {
    __shared__ float4 sA[256];  // x, y, conic a, conic b
    __shared__ float4 sB[256];  // conic c, opacity, r, g
    __shared__ float sC[256];   // b
    __shared__ unsigned long long sMask[4][4];  // [staging wave][quadrant]
    const int tile = blockIdx.x;
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

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    unsigned last_contributor = 0;
    bool done = !inside;

    for (int i = 0; i < rounds; i++) {
        if (MODE & 1) {
            __syncthreads();
        } else if (__syncthreads_and(done)) break;  // also orders the previous round's LDS reads before the refill
        const int at = (i << 8) + threadIdx.x;
        unsigned qm = 0;
        if (at < n) {
            const unsigned g = point_list[range.x + at];
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g * DGM_REC_STRIDE);
            const float4 r0 = r4[0], r1 = r4[1];
            const float cb = r4[2].x;
            sA[threadIdx.x] = r0;
            sB[threadIdx.x] = r1;
            sC[threadIdx.x] = cb;
            qm = CULL ? quadrant_mask(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, tx0, ty0) : 15u;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned long long bal = __ballot((qm >> q) & 1u);
            if (lane == 0) sMask[wv][q] = bal;
        }
        __syncthreads();
        if (!(MODE & 2) && __ballot(!done) == 0ull) continue;  // whole quadrant finished: keep helping with staging only
        const unsigned base = (unsigned)(i << 8);
#pragma unroll 1
        for (int sw = 0; sw < 4; sw++) {
            unsigned long long m = sMask[sw][wv];
            m = uniform_u64(m);
            while (m) {
                const int j = (sw << 6) + __builtin_ctzll(m);
                m &= m - 1;
                const float4 A = sA[j];
                const float4 B = sB[j];
                const float dx = A.x - pxf, dy = A.y - pyf;
                const float power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
                const float alpha = fminf(0.99f, B.y * fast_exp(power));
                bool valid = !done && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                const float test_T = T * (1.0f - alpha);
                if (valid && test_T < 0.0001f) {
                    done = true;
                    valid = false;
                }
                if (valid) {
                    const float w = alpha * T;
                    C0 += B.z * w;
                    C1 += B.w * w;
                    C2 += sC[j] * w;
                    T = test_T;
                    last_contributor = base + (unsigned)j + 1u;
                }
            }
        }
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

__global__ void __launch_bounds__(256)
render_bwd_kernel(const uint2* __restrict__ ranges, const unsigned* __restrict__ point_list, int W, int H, int gridx,
                  const float* __restrict__ bg, const float* __restrict__ rec, const float* __restrict__ final_Ts,
                  const unsigned* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
                  float* __restrict__ slab, unsigned* __restrict__ nproc_out) {
    __shared__ float4 sA[256];
    __shared__ float4 sB[256];
    __shared__ float sC[256];
    __shared__ __attribute__((aligned(16))) float sAcc[256 * DGM_SLAB_STRIDE];
    __shared__ unsigned long long sMask[4][4];
    __shared__ unsigned sMax[4];
    const int tile = blockIdx.x;
    const int tile_x = tile % gridx, tile_y = tile / gridx;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int px = tile_x * DGM_TILE + (wv & 1) * 8 + (lane & 7);
    const int py = tile_y * DGM_TILE + (wv >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float tx0 = (float)(tile_x * DGM_TILE), ty0 = (float)(tile_y * DGM_TILE);
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const size_t pid = (size_t)W * py + px;
    const size_t plane = (size_t)W * H;

    const float T_final = inside ? final_Ts[pid] : 0.f;
    const unsigned last_contributor = inside ? n_contrib[pid] : 0u;
    float dpx0 = 0.f, dpx1 = 0.f, dpx2 = 0.f;
    if (inside) {
        dpx0 = dL_dpixels[pid];
        dpx1 = dL_dpixels[plane + pid];
        dpx2 = dL_dpixels[2 * plane + pid];
    }
    const float bg_dot_dpixel = bg[0] * dpx0 + bg[1] * dpx1 + bg[2] * dpx2;
    // gradient of pixel coordinate w.r.t. NDC (backward.cu:460-461)
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    // instances the tile has to replay: max over its pixels of n_contrib
    {
        const unsigned m = wave_max_u32(last_contributor);
        if (lane == 0) sMax[wv] = m;
    }
    __syncthreads();
    int nproc = (int)max(max(sMax[0], sMax[1]), max(sMax[2], sMax[3]));
    nproc = min(nproc, n);
    if (threadIdx.x == 0) nproc_out[tile] = (unsigned)nproc;
    const int rounds = (nproc + 255) >> 8;

    float T = T_final;
    float ar0 = 0.f, ar1 = 0.f, ar2 = 0.f;  // accum_rec
    float lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;  // last_color
    float last_alpha = 0.f;

    for (int i = 0; i < rounds; i++) {
        __syncthreads();  // previous round's flush has read sAcc; previous blend has read sA/sB/sC
        const int at = (i << 8) + threadIdx.x;  // 0 = last replayed instance (back to front)
        const int pos = nproc - 1 - at;
        unsigned qm = 0;
        if (pos >= 0) {
            const unsigned g = point_list[range.x + pos];
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g * DGM_REC_STRIDE);
            const float4 r0 = r4[0], r1 = r4[1];
            sA[threadIdx.x] = r0;
            sB[threadIdx.x] = r1;
            sC[threadIdx.x] = r4[2].x;
            qm = quadrant_mask(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, tx0, ty0);
        }
        {
            float4* z = reinterpret_cast<float4*>(sAcc + threadIdx.x * DGM_SLAB_STRIDE);
            z[0] = make_float4(0.f, 0.f, 0.f, 0.f);
            z[1] = make_float4(0.f, 0.f, 0.f, 0.f);
            z[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned long long bal = __ballot((qm >> q) & 1u);
            if (lane == 0) sMask[wv][q] = bal;
        }
        __syncthreads();
        const int base_pos = nproc - 1 - (i << 8);  // list position of staged entry j is base_pos - j
#pragma unroll 1
        for (int sw = 0; sw < 4; sw++) {
            unsigned long long m = sMask[sw][wv];
            m = uniform_u64(m);
            while (m) {
                const int j = (sw << 6) + __builtin_ctzll(m);
                m &= m - 1;
                const float4 A = sA[j];
                const float4 B = sB[j];
                // contributor index of this entry is (base_pos - j); skip if it is behind this pixel's last
                // contributor (backward.cu:486-488)
                bool valid = inside && (unsigned)(base_pos - j) < last_contributor;
                const float dx = A.x - pxf, dy = A.y - pyf;
                const float power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
                const float G = fast_exp(power);
                const float alpha = fminf(0.99f, B.y * G);
                valid = valid && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                if (__ballot(valid) == 0ull) continue;
                const float cb = sC[j];
                float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f, v8 = 0.f;
                if (valid) {
                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;
                    ar0 = last_alpha * lc0 + (1.f - last_alpha) * ar0;
                    ar1 = last_alpha * lc1 + (1.f - last_alpha) * ar1;
                    ar2 = last_alpha * lc2 + (1.f - last_alpha) * ar2;
                    lc0 = B.z;
                    lc1 = B.w;
                    lc2 = cb;
                    float dL_dalpha = (B.z - ar0) * dpx0 + (B.w - ar1) * dpx1 + (cb - ar2) * dpx2;
                    v0 = dchannel_dcolor * dpx0;
                    v1 = dchannel_dcolor * dpx1;
                    v2 = dchannel_dcolor * dpx2;
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                    const float dL_dG = B.y * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * A.z - gdy * A.w;
                    const float dG_ddely = -gdy * B.x - gdx * A.w;
                    v3 = dL_dG * dG_ddelx * ddelx_dx;
                    v4 = dL_dG * dG_ddely * ddely_dy;
                    v5 = -0.5f * gdx * dx * dL_dG;
                    v6 = -0.5f * gdx * dy * dL_dG;
                    v7 = -0.5f * gdy * dy * dL_dG;
                    v8 = G * dL_dalpha;
                }
                wave_reduce9(v0, v1, v2, v3, v4, v5, v6, v7, v8);
                if (lane == 63) {
                    float* acc = sAcc + j * DGM_SLAB_STRIDE;
                    atomicAdd(acc + 0, v0);
                    atomicAdd(acc + 1, v1);
                    atomicAdd(acc + 2, v2);
                    atomicAdd(acc + 3, v3);
                    atomicAdd(acc + 4, v4);
                    atomicAdd(acc + 5, v5);
                    atomicAdd(acc + 6, v6);
                    atomicAdd(acc + 7, v7);
                    atomicAdd(acc + 8, v8);
                }
            }
        }
        __syncthreads();
        if (pos >= 0) {
            // row layout: dcolor r,g,b | dmean2D x,y | dconic a,b,c | dopacity | 0 0 0
            const float4* a4 = reinterpret_cast<const float4*>(sAcc + threadIdx.x * DGM_SLAB_STRIDE);
            float4* dst = reinterpret_cast<float4*>(slab + (size_t)(range.x + pos) * DGM_SLAB_STRIDE);
            dst[0] = a4[0];
            dst[1] = a4[1];
            dst[2] = a4[2];
        }
    }
}

void launch_render_fwd(hipStream_t st, int tiles, const uint2* ranges, const unsigned* point_list, int W, int H,
                       int gridx, const float* rec, const float* bg, float* out_color, float* final_T,
                       unsigned* n_contrib) {
    static const bool no_cull = getenv("DGM_NO_CULL") != nullptr;  // debugging aid: blend every staged splat
    const char* mode = getenv("DGM_FWD_MODE");
    if (mode && mode[0] == '1')
        hipLaunchKernelGGL((render_fwd_kernel<true, 1>), dim3(tiles), dim3(256), 0, st, ranges, point_list, W, H, gridx, rec,
                           bg, out_color, final_T, n_contrib);
    else if (mode && mode[0] == '2')
        hipLaunchKernelGGL((render_fwd_kernel<true, 2>), dim3(tiles), dim3(256), 0, st, ranges, point_list, W, H, gridx, rec,
                           bg, out_color, final_T, n_contrib);
    else if (mode && mode[0] == '3')
        hipLaunchKernelGGL((render_fwd_kernel<true, 3>), dim3(tiles), dim3(256), 0, st, ranges, point_list, W, H, gridx, rec,
                           bg, out_color, final_T, n_contrib);
    else if (no_cull)
        hipLaunchKernelGGL(render_fwd_kernel<false>, dim3(tiles), dim3(256), 0, st, ranges, point_list, W, H, gridx, rec,
                           bg, out_color, final_T, n_contrib);
    else
        hipLaunchKernelGGL(render_fwd_kernel<true>, dim3(tiles), dim3(256), 0, st, ranges, point_list, W, H, gridx, rec,
                           bg, out_color, final_T, n_contrib);
}

void launch_render_bwd(hipStream_t st, int tiles, const uint2* ranges, const unsigned* point_list, int W, int H,
                       int gridx, const float* bg, const float* rec, const float* final_T, const unsigned* n_contrib,
                       const float* dL_dpix, float* slab, unsigned* nproc) {
    hipLaunchKernelGGL(render_bwd_kernel, dim3(tiles), dim3(256), 0, st, ranges, point_list, W, H, gridx, bg, rec,
                       final_T, n_contrib, dL_dpix, slab, nproc);
}

}  // namespace dgm
