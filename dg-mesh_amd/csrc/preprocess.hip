// Per-Gaussian forward preprocessing for gfx950: view/clip projection, EWA 2D covariance, conic,
// screen-space radius and tile rectangle, SH -> RGB.
//
// Replaces FORWARD::preprocess / preprocessCUDA<3> (DGR/cuda_rasterizer/forward.cu:156-256 with helpers
// :20-152 and auxiliary.h:41-97,139-164) and checkFrustum (rasterizer_impl.cu:54-66).
//
// PARITY: integer outputs (radii, tile rect, tiles_touched) and depth bits feed the sort keys and must
// be bit-identical to the oracle, so this translation unit is compiled with -ffp-contract=off and every
// expression below is written in the canonical order of DESIGN.md section 3 (source order of the
// reference, GLM products as sum_k A(r,k)B(k,c) left to right, one rounding per operation).
//
// MI355X notes: one thread per Gaussian, 256-thread workgroups.  The (P,16,3) SH block is the only wide
// per-Gaussian input (192 B); it is streamed with coalesced 16-byte loads into LDS (row stride 49 dwords:
// odd => conflict-free column reads) instead of 48 strided 4-byte loads per lane.  All per-splat data the
// blend kernels need is packed into ONE 48-byte record so that they gather 3 x 16 B per instance.
#include "dgm_common.hpp"

#pragma clang fp contract(off)

namespace dgm {

__constant__ float kSH_C0 = 0.28209479177387814f;
__constant__ float kSH_C1 = 0.4886025119029199f;
__constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f};
__constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f,  -0.5900435899266435f};

__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* B) {
    B[0] = kSH_C0;
    if (deg > 0) {
        B[1] = -kSH_C1 * y;
        B[2] = kSH_C1 * z;
        B[3] = -kSH_C1 * x;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = kSH_C2[0] * xy;
            B[5] = kSH_C2[1] * yz;
            B[6] = kSH_C2[2] * (2.0f * zz - xx - yy);
            B[7] = kSH_C2[3] * xz;
            B[8] = kSH_C2[4] * (xx - yy);
            if (deg > 2) {
                B[9] = kSH_C3[0] * y * (3.0f * xx - yy);
                B[10] = kSH_C3[1] * xy * z;
                B[11] = kSH_C3[2] * y * (4.0f * zz - xx - yy);
                B[12] = kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                B[13] = kSH_C3[4] * x * (4.0f * zz - xx - yy);
                B[14] = kSH_C3[5] * z * (xx - yy);
                B[15] = kSH_C3[6] * x * (xx - 3.0f * yy);
            }
        }
    }
}

// Cooperative, coalesced copy of the first 3*n floats of each SH row of this workgroup's Gaussians
// into LDS with an odd row stride.  Must be called by all threads of the block.
// shs_rest != nullptr: the rows come in two pieces, as the model stores them -- shs = the DC term (P,1,3), shs_rest = the
// other M-1 coefficients (P,M-1,3) -- which saves the caller the concatenation (get_features).
__device__ __forceinline__ void stage_sh(const float* __restrict__ shs, const float* __restrict__ shs_rest, int base, int cnt,
                                         int M, int n, int stride, float* lds) {
    const int L = 3 * n;
    if (shs_rest != nullptr) {
        for (int i = threadIdx.x; i < cnt * 3; i += blockDim.x) {
            const int g = i / 3, k = i - g * 3;
            lds[g * stride + k] = shs[(size_t)base * 3 + i];
        }
        const int Lr = L - 3, Mr = 3 * (M - 1);
        const float* src = shs_rest + (size_t)base * Mr;
        if (n == M && ((cnt * Mr) & 3) == 0 && (((uintptr_t)src) & 15) == 0) {  // whole rows: 16-byte loads
            const float4* s4 = reinterpret_cast<const float4*>(src);
            const int total4 = (cnt * Mr) >> 2;
            for (int i = threadIdx.x; i < total4; i += blockDim.x) {
                const float4 v = s4[i];
                const float e4[4] = {v.x, v.y, v.z, v.w};
                const int e = i << 2;
#pragma unroll
                for (int c = 0; c < 4; c++) {  // Mr = 45: the four floats may straddle two rows
                    const int g = (e + c) / Mr, k = (e + c) - g * Mr;
                    lds[g * stride + 3 + k] = e4[c];
                }
            }
        } else {
            for (int i = threadIdx.x; i < cnt * Lr; i += blockDim.x) {
                const int g = i / Lr, k = i - g * Lr;
                lds[g * stride + 3 + k] = src[(size_t)g * Mr + k];
            }
        }
        return;
    }
    if (n == M && (L & 3) == 0) {
        const float4* src = reinterpret_cast<const float4*>(shs + (size_t)base * 3 * M);
        const int total4 = cnt * (L >> 2);
        for (int i = threadIdx.x; i < total4; i += blockDim.x) {
            float4 v = src[i];
            int e = i << 2;
            int g = e / L, k = e - g * L;
            float* d = lds + g * stride + k;  // L % 4 == 0 => the four floats stay in one row
            d[0] = v.x;
            d[1] = v.y;
            d[2] = v.z;
            d[3] = v.w;
        }
    } else {
        const int total = cnt * L;
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            int g = i / L, k = i - g * L;
            lds[g * stride + k] = shs[((size_t)(base + g) * M) * 3 + k];
        }
    }
}

__global__ void __launch_bounds__(DGM_PRE_BLOCK)
preprocess_fwd_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
                      float scale_modifier, const float* __restrict__ rotations, const float* __restrict__ opacities,
                      const float* __restrict__ shs, const float* __restrict__ shs_rest,
                      const float* __restrict__ cov3D_precomp,
                      const float* __restrict__ colors_precomp, const float* __restrict__ viewmatrix,
                      const float* __restrict__ projmatrix, const float* __restrict__ cam_pos, int W, int H,
                      float tan_fovx, float tan_fovy, float focal_x, float focal_y, int gridx, int gridy,
                      int prefiltered, int* __restrict__ radii_out, float* __restrict__ rec, float* __restrict__ depth,
                      int* __restrict__ radii_int, unsigned* __restrict__ tiles_touched, float* __restrict__ cov3Ds,
                      uint8_t* __restrict__ clamped, unsigned* __restrict__ block_sums) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ unsigned wave_tot[DGM_PRE_BLOCK / 64];
    const int base = blockIdx.x * DGM_PRE_BLOCK;
    const int cnt = min(DGM_PRE_BLOCK, P - base);
    const int idx = base + threadIdx.x;
    const int n_sh = (D + 1) * (D + 1);
    const int stride = (3 * n_sh) | 1;
    const bool use_sh = (colors_precomp == nullptr) && shs != nullptr && M > 0;
    if (use_sh) stage_sh(shs, shs_rest, base, cnt, M, n_sh, stride, lds);
    __syncthreads();

    unsigned my_tiles = 0;
    bool culled_prefiltered = false;
    if (idx < P) {
        int my_radius_i = 0;
        unsigned rect = 0;
        float px = 0.f, py = 0.f, con_a = 0.f, con_b = 0.f, con_c = 0.f, opac = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
        float zview = 0.f;
        uint8_t clamp_bits = 0;
        const float p0 = means3D[3 * idx], p1 = means3D[3 * idx + 1], p2 = means3D[3 * idx + 2];
        const float* vm = viewmatrix;
        const float* pm = projmatrix;
        // transformPoint4x3 / 4x4 (auxiliary.h:58-77)
        const float vx = vm[0] * p0 + vm[4] * p1 + vm[8] * p2 + vm[12];
        const float vy = vm[1] * p0 + vm[5] * p1 + vm[9] * p2 + vm[13];
        const float vz = vm[2] * p0 + vm[6] * p1 + vm[10] * p2 + vm[14];
        bool alive = !(vz <= 0.2f);  // auxiliary.h:154
        culled_prefiltered = !alive && prefiltered;  // reference: printf + __trap (auxiliary.h:156-160); reported through block_sums' top bit
        if (alive) {
            const float hx = pm[0] * p0 + pm[4] * p1 + pm[8] * p2 + pm[12];
            const float hy = pm[1] * p0 + pm[5] * p1 + pm[9] * p2 + pm[13];
            const float hw = pm[3] * p0 + pm[7] * p1 + pm[11] * p2 + pm[15];
            const float p_w = 1.0f / (hw + 0.0000001f);
            const float projx = hx * p_w, projy = hy * p_w;
            float c3[6];
            if (cov3D_precomp != nullptr) {
#pragma unroll
                for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * idx + i];
            } else {
                // computeCov3D (forward.cu:118-152), quaternion not normalised
                const float s0 = scale_modifier * scales[3 * idx], s1 = scale_modifier * scales[3 * idx + 1],
                            s2 = scale_modifier * scales[3 * idx + 2];
                const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
                const float r = q.x, x = q.y, y = q.z, z = q.w;
                float Rm[3][3];
                Rm[0][0] = 1.f - 2.f * (y * y + z * z);
                Rm[1][0] = 2.f * (x * y - r * z);
                Rm[2][0] = 2.f * (x * z + r * y);
                Rm[0][1] = 2.f * (x * y + r * z);
                Rm[1][1] = 1.f - 2.f * (x * x + z * z);
                Rm[2][1] = 2.f * (y * z - r * x);
                Rm[0][2] = 2.f * (x * z - r * y);
                Rm[1][2] = 2.f * (y * z + r * x);
                Rm[2][2] = 1.f - 2.f * (x * x + y * y);
                float Mm[3][3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    Mm[0][c] = s0 * Rm[0][c];
                    Mm[1][c] = s1 * Rm[1][c];
                    Mm[2][c] = s2 * Rm[2][c];
                }
#define DGM_SIG(r_, c_) (Mm[0][r_] * Mm[0][c_] + Mm[1][r_] * Mm[1][c_] + Mm[2][r_] * Mm[2][c_])
                c3[0] = DGM_SIG(0, 0);
                c3[1] = DGM_SIG(1, 0);
                c3[2] = DGM_SIG(2, 0);
                c3[3] = DGM_SIG(1, 1);
                c3[4] = DGM_SIG(2, 1);
                c3[5] = DGM_SIG(2, 2);
#undef DGM_SIG
#pragma unroll
                for (int i = 0; i < 6; i++) cov3Ds[6 * idx + i] = c3[i];
            }
            // computeCov2D (forward.cu:74-113)
            const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
            const float txtz = vx / vz, tytz = vy / vz;
            const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
            const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
            const float J00 = focal_x / vz;
            const float J20 = -(focal_x * tx) / (vz * vz);
            const float J11 = focal_y / vz;
            const float J21 = -(focal_y * ty) / (vz * vz);
            float T[3][2];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const float w0 = vm[4 * r + 0], w1 = vm[4 * r + 1], w2 = vm[4 * r + 2];
                T[r][0] = (w0 * J00 + w1 * 0.0f) + w2 * J20;
                T[r][1] = (w0 * 0.0f + w1 * J11) + w2 * J21;
            }
            const float V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
            float U[2][3];
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int k = 0; k < 3; k++) U[r][k] = T[0][r] * V[k][0] + T[1][r] * V[k][1] + T[2][r] * V[k][2];
            float ca = U[0][0] * T[0][0] + U[0][1] * T[1][0] + U[0][2] * T[2][0];
            const float cbv = U[1][0] * T[0][0] + U[1][1] * T[1][0] + U[1][2] * T[2][0];
            float cc = U[1][0] * T[0][1] + U[1][1] * T[1][1] + U[1][2] * T[2][1];
            ca += 0.3f;
            cc += 0.3f;
            const float det = (ca * cc - cbv * cbv);
            if (det != 0.0f) {
                const float det_inv = 1.f / det;
                con_a = cc * det_inv;
                con_b = -cbv * det_inv;
                con_c = ca * det_inv;
                const float mid = 0.5f * (ca + cc);
                const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
                const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
                const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
                // ndc2Pix in binary64 (auxiliary.h:41-44)
                px = (float)((((double)projx + 1.0) * (double)W - 1.0) * 0.5);
                py = (float)((((double)projy + 1.0) * (double)H - 1.0) * 0.5);
                const int ri = f2i_sat(my_radius);
                // getRect (auxiliary.h:46-56)
                const int rminx = min(gridx, max(0, f2i_sat((px - ri) / DGM_TILE)));
                const int rminy = min(gridy, max(0, f2i_sat((py - ri) / DGM_TILE)));
                const int rmaxx = min(gridx, max(0, f2i_sat((px + ri + DGM_TILE - 1) / DGM_TILE)));
                const int rmaxy = min(gridy, max(0, f2i_sat((py + ri + DGM_TILE - 1) / DGM_TILE)));
                const unsigned touched = (unsigned)(rmaxx - rminx) * (unsigned)(rmaxy - rminy);
                if (touched != 0) {
                    my_tiles = touched;
                    my_radius_i = ri;
                    rect = pack_rect((unsigned)rminx, (unsigned)rminy, (unsigned)(rmaxx - rminx));
                    zview = vz;
                    opac = opacities[idx];
                    if (colors_precomp != nullptr) {
                        cr = colors_precomp[3 * idx];
                        cg = colors_precomp[3 * idx + 1];
                        cb = colors_precomp[3 * idx + 2];
                    } else if (use_sh) {
                        // computeColorFromSH (forward.cu:20-71), direction from the (deformed) mean
                        const float d0 = p0 - cam_pos[0], d1 = p1 - cam_pos[1], d2 = p2 - cam_pos[2];
                        const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
                        float B[16];
                        sh_basis(D, d0 / len, d1 / len, d2 / len, B);
                        const float* sh = lds + threadIdx.x * stride;
                        float res[3];
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) res[ch] = B[0] * sh[ch];
                        // constant indices only (a runtime-indexed B[] would be demoted to scratch memory)
#pragma unroll
                        for (int k = 1; k < 16; k++) {
                            if (k < n_sh) {
#pragma unroll
                                for (int ch = 0; ch < 3; ch++) res[ch] = res[ch] + B[k] * sh[3 * k + ch];
                            }
                        }
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            res[ch] += 0.5f;
                            if (res[ch] < 0) clamp_bits |= (uint8_t)(1u << ch);
                            res[ch] = fmaxf(res[ch], 0.0f);
                        }
                        cr = res[0];
                        cg = res[1];
                        cb = res[2];
                    }
                }
            }
        }
        // splat record (48 B): what render_fwd / render_bwd gather per instance
        float4* r4 = reinterpret_cast<float4*>(rec + (size_t)idx * DGM_REC_STRIDE);
        r4[0] = make_float4(px, py, con_a, con_b);
        r4[1] = make_float4(con_c, opac, cr, cg);
        r4[2] = make_float4(cb, __uint_as_float(rect), __uint_as_float(0u), 0.0f);
        depth[idx] = zview;
        radii_int[idx] = my_radius_i;
        if (radii_out) radii_out[idx] = my_radius_i;
        tiles_touched[idx] = my_tiles;
        clamped[idx] = clamp_bits;
    }
    // per-block sum of tiles_touched (first level of the exclusive scan over Gaussians)
    // (bit 31: some Gaussian of the block was culled although the caller said prefiltered -- the count kernel gathers these into
    // counters[1], so that no counter word has to be zero before this kernel runs; a block's sum stays below 2^31: 256 x 36 k tiles)
    unsigned s = wave_sum_u32(my_tiles);
    if (__any(culled_prefiltered)) s |= 0x80000000u;
    if (lane_id() == 0) wave_tot[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0, bad = 0;
#pragma unroll
        for (int w = 0; w < DGM_PRE_BLOCK / 64; w++) t += wave_tot[w] & 0x7fffffffu, bad |= wave_tot[w] & 0x80000000u;
        block_sums[blockIdx.x] = t | bad;
    }
}

__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ vm,
                                    uint8_t* __restrict__ present) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float p0 = means3D[3 * idx], p1 = means3D[3 * idx + 1], p2 = means3D[3 * idx + 2];
    const float vz = vm[2] * p0 + vm[6] * p1 + vm[10] * p2 + vm[14];
    present[idx] = !(vz <= 0.2f);
}

void launch_preprocess_fwd(hipStream_t st, int P, int D, int M, const float* means3D, const float* scales,
                           float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                           const float* shs_rest, const float* cov3D_precomp, const float* colors_precomp,
                           const float* viewmatrix,
                           const float* projmatrix, const float* cam_pos, int W, int H, float tan_fovx, float tan_fovy,
                           int gridx, int gridy, int prefiltered, int* radii_out, float* rec, float* depth,
                           int* radii_int, unsigned* tiles_touched, float* cov3Ds, uint8_t* clamped,
                           unsigned* block_sums) {
    const float focal_y = H / (2.0f * tan_fovy);  // rasterizer_impl.cu:222-223
    const float focal_x = W / (2.0f * tan_fovx);
    const int n_sh = (D + 1) * (D + 1);
    const size_t lds_bytes = (colors_precomp == nullptr && shs != nullptr && M > 0)
                                 ? (size_t)DGM_PRE_BLOCK * ((3 * n_sh) | 1) * sizeof(float)
                                 : 16;
    const int nblk = (P + DGM_PRE_BLOCK - 1) / DGM_PRE_BLOCK;
    hipLaunchKernelGGL(preprocess_fwd_kernel, dim3(nblk), dim3(DGM_PRE_BLOCK), lds_bytes, st, P, D, M, means3D, scales,
                       scale_modifier, rotations, opacities, shs, shs_rest, cov3D_precomp, colors_precomp, viewmatrix, projmatrix,
                       cam_pos, W, H, tan_fovx, tan_fovy, focal_x, focal_y, gridx, gridy, prefiltered, radii_out, rec,
                       depth, radii_int, tiles_touched, cov3Ds, clamped, block_sums);
}

void launch_mark_visible(hipStream_t st, int P, const float* means3D, const float* viewmatrix, uint8_t* present) {
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, means3D, viewmatrix, present);
}

}  // namespace dgm
