// Per-Gaussian backward for gfx950: gathers the per-instance gradient rows written by render_bwd and
// back-propagates through conic -> cov2D -> cov3D -> (scale, rotation), the projection of the mean, and the
// SH colour evaluation.  ONE kernel, one thread per Gaussian, every output written exactly once.
//
// Replaces the 9 fp32 atomicAdd per (pixel, splat) of the reference's renderCUDA backward
// (DGR/cuda_rasterizer/backward.cu:523-554) by a deterministic gather, and fuses computeCov2DCUDA
// (backward.cu:144-274) with preprocessCUDA<3> backward (backward.cu:347-396; helpers :20-139 SH,
// :278-341 cov3D, auxiliary.h:107-117 dnormvdv).
//
// Gather: Gaussian g owns instances k = 0..tiles_touched-1 (row-major over its tile rectangle); render_bwd wrote
// live[offs[g] + k] for every one of them and, where that is 1, the 36-byte row of instance k at slab row offs[g] + k, so
// the rows and flags of a Gaussian -- and of neighbouring Gaussians -- are adjacent.  Only live rows are read (instances no
// pixel blended have none); they are summed in ascending k (fixed order => reproducible grads).
//
// The SH block (192 B in, 192 B out per Gaussian) goes through LDS both ways so that global traffic is
// coalesced 16-byte accesses (same scheme as preprocess.hip).
#include "dgm_common.hpp"

namespace dgm {

__constant__ float kbSH_C0 = 0.28209479177387814f;
__constant__ float kbSH_C1 = 0.4886025119029199f;
__constant__ float kbSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                 -1.0925484305920792f, 0.5462742152960396f};
__constant__ float kbSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                 -0.4570457994644658f, 1.445305721320277f,  -0.5900435899266435f};

__global__ void __launch_bounds__(DGM_PRE_BLOCK)
preprocess_bwd_kernel(int P, int D, int M, int gridx, const float* __restrict__ means3D, const int* __restrict__ radii,
                      const float* __restrict__ shs, const float* __restrict__ shs_rest, const uint8_t* __restrict__ clamped,
                      const float* __restrict__ scales, const float* __restrict__ rotations, float scale_modifier,
                      const float* __restrict__ cov3Ds, const float* __restrict__ vm, const float* __restrict__ proj,
                      const float* __restrict__ campos, float h_x, float h_y, float tan_fovx, float tan_fovy, int W, int H,
                      const float* __restrict__ rec, const unsigned* __restrict__ tiles_touched,
                      const unsigned* __restrict__ offs, const float* __restrict__ slab, const uint8_t* __restrict__ live,
                      float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
                      float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor, float* __restrict__ dL_dmean3D,
                      float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh, float* __restrict__ dL_dsh_rest,
                      float* __restrict__ dL_dscale, float* __restrict__ dL_drot) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int base = blockIdx.x * DGM_PRE_BLOCK;
    const int cnt = min(DGM_PRE_BLOCK, P - base);
    const int idx = base + threadIdx.x;
    const int n_sh = (D + 1) * (D + 1);
    const bool use_sh = shs != nullptr && M > 0;
    const int L = 3 * M;            // full row: gradients of unused coefficients are written as zeros
    const int stride = L | 1;

    // ---- everything that does not depend on the gather is asked for FIRST: the Gaussian's own inputs (used after the barrier) and
    // the SH block's staging loads.  Round 1-5a order -- gather, then staging, barrier, then the inputs -- made five dependent
    // memory round trips out of what is three (sizes and offsets -> flags -> rows): the kernel is one round of workgroups, 1.5 waves
    // per SIMD, so its time is the length of that chain.
    float4 in_r0 = make_float4(0.f, 0.f, 0.f, 0.f), in_r1 = in_r0, in_rot = in_r0;
    float in_m[3] = {0.f, 0.f, 0.f}, in_c3[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, in_s[3] = {0.f, 0.f, 0.f};
    int in_radius = 0;
    unsigned in_n = 0u, in_first = 0u;
    uint8_t in_cl = 0;
    if (idx < P) {
        in_radius = radii[idx];
        in_n = tiles_touched[idx];
        in_first = offs[idx];
        in_r0 = reinterpret_cast<const float4*>(rec + (size_t)idx * DGM_REC_STRIDE)[0];
        in_r1 = reinterpret_cast<const float4*>(rec + (size_t)idx * DGM_REC_STRIDE)[1];
#pragma unroll
        for (int i = 0; i < 3; i++) in_m[i] = means3D[3 * idx + i];
#pragma unroll
        for (int i = 0; i < 6; i++) in_c3[i] = cov3Ds[6 * idx + i];
        in_cl = clamped[idx];
        if (scales != nullptr) {
#pragma unroll
            for (int i = 0; i < 3; i++) in_s[i] = scales[3 * idx + i];
            in_rot = make_float4(rotations[4 * idx], rotations[4 * idx + 1], rotations[4 * idx + 2], rotations[4 * idx + 3]);
        }
    }
    if (use_sh) {
        // stage the whole (cnt, M, 3) block, coalesced
        const int total = cnt * L;
        const float* src = shs + (size_t)base * L;
        if (shs_rest != nullptr) {  // rows in two pieces (DC | rest), see preprocess.hip: stage_sh
            for (int i = threadIdx.x; i < cnt * 3; i += DGM_PRE_BLOCK) {
                const int g = i / 3, k = i - g * 3;
                lds[g * stride + k] = shs[(size_t)base * 3 + i];
            }
            const int Mr = L - 3;
            const float* sr = shs_rest + (size_t)base * Mr;
            if (((cnt * Mr) & 3) == 0 && (((uintptr_t)sr) & 15) == 0) {
                const float4* s4 = reinterpret_cast<const float4*>(sr);
                for (int i = threadIdx.x; i < ((cnt * Mr) >> 2); i += DGM_PRE_BLOCK) {
                    const float4 v = s4[i];
                    const float e4[4] = {v.x, v.y, v.z, v.w};
                    const int e = i << 2;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int g = (e + c) / Mr, k = (e + c) - g * Mr;
                        lds[g * stride + 3 + k] = e4[c];
                    }
                }
            } else {
                for (int i = threadIdx.x; i < cnt * Mr; i += DGM_PRE_BLOCK) {
                    const int g = i / Mr, k = i - g * Mr;
                    lds[g * stride + 3 + k] = sr[i];
                }
            }
        } else if ((L & 3) == 0 && (((uintptr_t)src) & 15) == 0) {
            const float4* s4 = reinterpret_cast<const float4*>(src);
            for (int i = threadIdx.x; i < (total >> 2); i += DGM_PRE_BLOCK) {
                const float4 v = s4[i];
                const int e = i << 2;
                const int g = e / L, k = e - g * L;
                float* d = lds + g * stride + k;
                d[0] = v.x;
                d[1] = v.y;
                d[2] = v.z;
                d[3] = v.w;
            }
        } else {
            for (int i = threadIdx.x; i < total; i += DGM_PRE_BLOCK) {
                const int g = i / L, k = i - g * L;
                lds[g * stride + k] = src[i];
            }
        }
    }
    // ---- gather: this Gaussian's gradient rows, summed ---------------------------------------------------------------------
    // Gaussian g owns slab rows offs[g] .. offs[g] + tiles_touched[g] - 1 and their liveness bytes: one thread walks them eight at
    // a time -- ONE 8-byte load for the flags, then 16-byte loads for the rows render_bwd wrote (a row whose flag is 0 holds stale
    // bytes and is never read), summed in ascending order (fixed order => reproducible).
    // Measured at cfg2 (3.0 M rows, about half of them dead), same kernel otherwise: this form 0.084 ms; flags read as eight
    // byte loads 0.106; every row read and dead ones dropped by a select 0.118; 48-byte aligned rows with the dead ones written
    // as zeros (round 3's form, no flags) 0.069 in round 3's library and 0.110 here with the flag loads added; the workgroup's
    // row span streamed through LDS in 1024-row chunks (coalesced, nine loads in flight) 0.126 -- only ~34 of the 256 threads
    // own rows of a given chunk.  The kernel is bound by the number of load instructions whose lanes touch 64 different
    // cache lines, so the byte loads of the flags cost as much as the rows' 16-byte loads.
    float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (idx < P && in_radius > 0) {
        const unsigned n = in_n;
        const size_t first = in_first;
        const uint8_t* fl = live + first;
        const float* row = slab + first * DGM_SLAB_STRIDE;
        // eight flags in ONE load (byte-aligned 8-byte access; up to 7 bytes past the Gaussian's own flags, inside the array's padding);
        // the next eight are asked for before this batch's rows, so that a batch costs one dependent round trip, not two
        unsigned long long f8_next = n ? *reinterpret_cast<const dgm_u64u*>(fl) : 0ull;
        for (unsigned k0 = 0; k0 < n; k0 += 8) {
            const unsigned long long f8 = f8_next;
            if (k0 + 8 < n) f8_next = *reinterpret_cast<const dgm_u64u*>(fl + k0 + 8);
            bool on[8];
#pragma unroll
            for (int j = 0; j < 8; j++) on[j] = k0 + j < n && ((f8 >> (8 * j)) & 0xffull) != 0;
            float r[8][9];
#pragma unroll
            for (int j = 0; j < 8; j++) {
#pragma unroll
                for (int i = 0; i < 9; i++) r[j][i] = 0.f;
                if (on[j]) {
                    const float* rp = row + (size_t)(k0 + j) * DGM_SLAB_STRIDE;
                    const dgm_f4u a0 = *reinterpret_cast<const dgm_f4u*>(rp), a1 = *reinterpret_cast<const dgm_f4u*>(rp + 4);
                    r[j][0] = a0.x, r[j][1] = a0.y, r[j][2] = a0.z, r[j][3] = a0.w;
                    r[j][4] = a1.x, r[j][5] = a1.y, r[j][6] = a1.z, r[j][7] = a1.w;
                    r[j][8] = rp[8];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
#pragma unroll
                for (int i = 0; i < 9; i++) acc[i] += r[j][i];
            }
        }
    }

    __syncthreads();

    if (idx < P) {
        const bool vis = in_radius > 0;
        if (vis) {
            // render_bwd4 rows hold colour sums and the moments of g = G dL/dalpha about the splat centre
            // (d = xy - pixel): acc[3..8] = sum g dx, g dy, g dx^2, g dx dy, g dy^2, g.  The gradients of
            // backward.cu:536-554 are linear in them with per-Gaussian coefficients (dL/dG = opacity dL/dalpha):
            //   dL/dmean2D = -o (a Mx + b My) W/2,  -o (c My + b Mx) H/2        (dG/ddel = -G (a dx + b dy), ...)
            //   dL/dconic  = -o/2 (Mxx, Mxy, Myy),   dL/dopacity = M0
            const float4 r0 = in_r0, r1 = in_r1;
            const float ca = r0.z, cb = r0.w, cc = r1.x, op = r1.y;
            const float mx = acc[3], my = acc[4];
            acc[3] = -op * (ca * mx + cb * my) * (0.5f * W);
            acc[4] = -op * (cc * my + cb * mx) * (0.5f * H);
            acc[5] *= -0.5f * op;
            acc[6] *= -0.5f * op;
            acc[7] *= -0.5f * op;
        }
        // outputs of the blend backward (reference: atomically accumulated arrays)
        dL_dcolor[3 * idx + 0] = acc[0];
        dL_dcolor[3 * idx + 1] = acc[1];
        dL_dcolor[3 * idx + 2] = acc[2];
        dL_dmean2D[3 * idx + 0] = acc[3];
        dL_dmean2D[3 * idx + 1] = acc[4];
        dL_dmean2D[3 * idx + 2] = 0.f;
        reinterpret_cast<float4*>(dL_dconic)[idx] = make_float4(acc[5], acc[6], 0.f, acc[7]);
        dL_dopacity[idx] = acc[8];

        float dmean[3] = {0.f, 0.f, 0.f};
        float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float dscale[3] = {0.f, 0.f, 0.f};
        float drot[4] = {0.f, 0.f, 0.f, 0.f};
        float* my_sh = lds + threadIdx.x * stride;
        if (vis) {
            const float m0 = in_m[0], m1 = in_m[1], m2 = in_m[2];
            // ---- computeCov2DCUDA (backward.cu:144-274) ----
            {
                float c3[6];
#pragma unroll
                for (int i = 0; i < 6; i++) c3[i] = in_c3[i];
                float t0 = vm[0] * m0 + vm[4] * m1 + vm[8] * m2 + vm[12];
                float t1 = vm[1] * m0 + vm[5] * m1 + vm[9] * m2 + vm[13];
                const float t2 = vm[2] * m0 + vm[6] * m1 + vm[10] * m2 + vm[14];
                const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
                const float txtz = t0 / t2, tytz = t1 / t2;
                t0 = fminf(limx, fmaxf(-limx, txtz)) * t2;
                t1 = fminf(limy, fmaxf(-limy, tytz)) * t2;
                const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
                const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
                const float J00 = h_x / t2, J20 = -(h_x * t0) / (t2 * t2);
                const float J11 = h_y / t2, J21 = -(h_y * t1) / (t2 * t2);
                float Tm[3][2];
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    Tm[r][0] = vm[4 * r + 0] * J00 + vm[4 * r + 2] * J20;
                    Tm[r][1] = vm[4 * r + 1] * J11 + vm[4 * r + 2] * J21;
                }
                const float V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
                float U[2][3];
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int k = 0; k < 3; k++) U[r][k] = Tm[0][r] * V[k][0] + Tm[1][r] * V[k][1] + Tm[2][r] * V[k][2];
                const float a = U[0][0] * Tm[0][0] + U[0][1] * Tm[1][0] + U[0][2] * Tm[2][0] + 0.3f;
                const float b = U[1][0] * Tm[0][0] + U[1][1] * Tm[1][0] + U[1][2] * Tm[2][0];
                const float c = U[1][0] * Tm[0][1] + U[1][1] * Tm[1][1] + U[1][2] * Tm[2][1] + 0.3f;
                const float g0 = acc[5], g1 = acc[6], g2 = acc[7];
                const float denom = a * c - b * b;
                float dL_da = 0, dL_db = 0, dL_dc = 0;
                const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
                // GLM T[i][j] = Tm[j][i]
#define TT(i_, j_) (Tm[j_][i_])
                if (denom2inv != 0) {
                    dL_da = denom2inv * (-c * c * g0 + 2 * b * c * g1 + (denom - a * c) * g2);
                    dL_dc = denom2inv * (-a * a * g2 + 2 * a * b * g1 + (denom - a * c) * g0);
                    dL_db = denom2inv * 2 * (b * c * g0 - (denom + 2 * b * b) * g1 + a * b * g2);
                    dcov[0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
                    dcov[3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
                    dcov[5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
                    dcov[1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db +
                              2 * TT(1, 0) * TT(1, 1) * dL_dc;
                    dcov[2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db +
                              2 * TT(1, 0) * TT(1, 2) * dL_dc;
                    dcov[4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db +
                              2 * TT(1, 1) * TT(1, 2) * dL_dc;
                }
                // TV(i,k) = sum_j T[i][j] * Vrk[k][j]
#define TV(i_, k_) (TT(i_, 0) * V[k_][0] + TT(i_, 1) * V[k_][1] + TT(i_, 2) * V[k_][2])
                const float dL_dT00 = 2 * TV(0, 0) * dL_da + TV(1, 0) * dL_db;
                const float dL_dT01 = 2 * TV(0, 1) * dL_da + TV(1, 1) * dL_db;
                const float dL_dT02 = 2 * TV(0, 2) * dL_da + TV(1, 2) * dL_db;
                const float dL_dT10 = 2 * TV(1, 0) * dL_dc + TV(0, 0) * dL_db;
                const float dL_dT11 = 2 * TV(1, 1) * dL_dc + TV(0, 1) * dL_db;
                const float dL_dT12 = 2 * TV(1, 2) * dL_dc + TV(0, 2) * dL_db;
#undef TV
#undef TT
                // GLM W[i][j] = vm[4*j + i]
                const float dL_dJ00 = vm[0] * dL_dT00 + vm[4] * dL_dT01 + vm[8] * dL_dT02;
                const float dL_dJ02 = vm[2] * dL_dT00 + vm[6] * dL_dT01 + vm[10] * dL_dT02;
                const float dL_dJ11 = vm[1] * dL_dT10 + vm[5] * dL_dT11 + vm[9] * dL_dT12;
                const float dL_dJ12 = vm[2] * dL_dT10 + vm[6] * dL_dT11 + vm[10] * dL_dT12;
                const float tz = 1.f / t2, tz2 = tz * tz, tz3 = tz2 * tz;
                const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
                const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
                const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t0) * tz3 * dL_dJ02 +
                                     (2 * h_y * t1) * tz3 * dL_dJ12;
                dmean[0] = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
                dmean[1] = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
                dmean[2] = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;
            }
            // ---- projection of the mean (backward.cu:370-387) ----
            {
                const float hw = proj[3] * m0 + proj[7] * m1 + proj[11] * m2 + proj[15];
                const float m_w = 1.0f / (hw + 0.0000001f);
                const float mul1 = (proj[0] * m0 + proj[4] * m1 + proj[8] * m2 + proj[12]) * m_w * m_w;
                const float mul2 = (proj[1] * m0 + proj[5] * m1 + proj[9] * m2 + proj[13]) * m_w * m_w;
                const float gx = acc[3], gy = acc[4];
                dmean[0] += (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
                dmean[1] += (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
                dmean[2] += (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
            }
            // ---- SH colour (backward.cu:20-139); reads this thread's LDS row, then overwrites it with dL_dsh ----
            if (use_sh) {
                const float d0 = m0 - campos[0], d1 = m1 - campos[1], d2 = m2 - campos[2];
                const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
                const float x = d0 / len, y = d1 / len, z = d2 / len;
                const uint8_t cl = in_cl;
                float dRGB[3] = {(cl & 1) ? 0.f : acc[0], (cl & 2) ? 0.f : acc[1], (cl & 4) ? 0.f : acc[2]};
                float dx[3] = {0, 0, 0}, dy[3] = {0, 0, 0}, dz[3] = {0, 0, 0};
                float Bk[16];
#pragma unroll
                for (int k = 0; k < 16; k++) Bk[k] = 0.f;
                Bk[0] = kbSH_C0;
#define S(k_) my_sh[3 * (k_) + ch]
                if (D > 0) {
                    Bk[1] = -kbSH_C1 * y;
                    Bk[2] = kbSH_C1 * z;
                    Bk[3] = -kbSH_C1 * x;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        dx[ch] = -kbSH_C1 * S(3);
                        dy[ch] = -kbSH_C1 * S(1);
                        dz[ch] = kbSH_C1 * S(2);
                    }
                    if (D > 1) {
                        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        Bk[4] = kbSH_C2[0] * xy;
                        Bk[5] = kbSH_C2[1] * yz;
                        Bk[6] = kbSH_C2[2] * (2.f * zz - xx - yy);
                        Bk[7] = kbSH_C2[3] * xz;
                        Bk[8] = kbSH_C2[4] * (xx - yy);
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            dx[ch] += kbSH_C2[0] * y * S(4) + kbSH_C2[2] * 2.f * -x * S(6) + kbSH_C2[3] * z * S(7) +
                                      kbSH_C2[4] * 2.f * x * S(8);
                            dy[ch] += kbSH_C2[0] * x * S(4) + kbSH_C2[1] * z * S(5) + kbSH_C2[2] * 2.f * -y * S(6) +
                                      kbSH_C2[4] * 2.f * -y * S(8);
                            dz[ch] += kbSH_C2[1] * y * S(5) + kbSH_C2[2] * 2.f * 2.f * z * S(6) + kbSH_C2[3] * x * S(7);
                        }
                        if (D > 2) {
                            Bk[9] = kbSH_C3[0] * y * (3.f * xx - yy);
                            Bk[10] = kbSH_C3[1] * xy * z;
                            Bk[11] = kbSH_C3[2] * y * (4.f * zz - xx - yy);
                            Bk[12] = kbSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                            Bk[13] = kbSH_C3[4] * x * (4.f * zz - xx - yy);
                            Bk[14] = kbSH_C3[5] * z * (xx - yy);
                            Bk[15] = kbSH_C3[6] * x * (xx - 3.f * yy);
#pragma unroll
                            for (int ch = 0; ch < 3; ch++) {
                                dx[ch] += (kbSH_C3[0] * S(9) * 3.f * 2.f * xy + kbSH_C3[1] * S(10) * yz +
                                           kbSH_C3[2] * S(11) * -2.f * xy + kbSH_C3[3] * S(12) * -3.f * 2.f * xz +
                                           kbSH_C3[4] * S(13) * (-3.f * xx + 4.f * zz - yy) +
                                           kbSH_C3[5] * S(14) * 2.f * xz + kbSH_C3[6] * S(15) * 3.f * (xx - yy));
                                dy[ch] += (kbSH_C3[0] * S(9) * 3.f * (xx - yy) + kbSH_C3[1] * S(10) * xz +
                                           kbSH_C3[2] * S(11) * (-3.f * yy + 4.f * zz - xx) +
                                           kbSH_C3[3] * S(12) * -3.f * 2.f * yz + kbSH_C3[4] * S(13) * -2.f * xy +
                                           kbSH_C3[5] * S(14) * -2.f * yz + kbSH_C3[6] * S(15) * -3.f * 2.f * xy);
                                dz[ch] += (kbSH_C3[1] * S(10) * xy + kbSH_C3[2] * S(11) * 4.f * 2.f * yz +
                                           kbSH_C3[3] * S(12) * 3.f * (2.f * zz - xx - yy) +
                                           kbSH_C3[4] * S(13) * 4.f * 2.f * xz + kbSH_C3[5] * S(14) * (xx - yy));
                            }
                        }
                    }
                }
#undef S
                // constant indices only (a runtime-indexed Bk[] would be demoted to scratch memory)
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (k < M) {
                        const float bk = k < n_sh ? Bk[k] : 0.f;
                        my_sh[3 * k + 0] = bk * dRGB[0];
                        my_sh[3 * k + 1] = bk * dRGB[1];
                        my_sh[3 * k + 2] = bk * dRGB[2];
                    }
                }
                for (int k = 48; k < L; k++) my_sh[k] = 0.f;
                const float dd0 = dx[0] * dRGB[0] + dx[1] * dRGB[1] + dx[2] * dRGB[2];
                const float dd1 = dy[0] * dRGB[0] + dy[1] * dRGB[1] + dy[2] * dRGB[2];
                const float dd2 = dz[0] * dRGB[0] + dz[1] * dRGB[1] + dz[2] * dRGB[2];
                // dnormvdv (auxiliary.h:107-117)
                const float sum2 = d0 * d0 + d1 * d1 + d2 * d2;
                const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
                dmean[0] += ((+sum2 - d0 * d0) * dd0 - d1 * d0 * dd1 - d2 * d0 * dd2) * invsum32;
                dmean[1] += (-d0 * d1 * dd0 + (sum2 - d1 * d1) * dd1 - d2 * d1 * dd2) * invsum32;
                dmean[2] += (-d0 * d2 * dd0 - d1 * d2 * dd1 + (sum2 - d2 * d2) * dd2) * invsum32;
            }
            // ---- cov3D -> scale / rotation (backward.cu:278-341) ----
            if (scales != nullptr) {
                const float r = in_rot.x, x = in_rot.y, y = in_rot.z, z = in_rot.w;
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
                const float s[3] = {scale_modifier * in_s[0], scale_modifier * in_s[1], scale_modifier * in_s[2]};
                const float Dm[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                        {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                        {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
                float G[3][3], Hh[3][3];
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = 0; b < 3; b++) {
                        G[a][b] = (2.0f * (s[a] * Rm[a][0])) * Dm[0][b] + (2.0f * (s[a] * Rm[a][1])) * Dm[1][b] +
                                  (2.0f * (s[a] * Rm[a][2])) * Dm[2][b];
                    }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    dscale[k] = Rm[k][0] * G[k][0] + Rm[k][1] * G[k][1] + Rm[k][2] * G[k][2];
#pragma unroll
                    for (int j = 0; j < 3; j++) Hh[k][j] = G[k][j] * s[k];
                }
#define Mt(i_, j_) Hh[i_][j_]
                drot[0] = 2 * z * (Mt(0, 1) - Mt(1, 0)) + 2 * y * (Mt(2, 0) - Mt(0, 2)) + 2 * x * (Mt(1, 2) - Mt(2, 1));
                drot[1] = 2 * y * (Mt(1, 0) + Mt(0, 1)) + 2 * z * (Mt(2, 0) + Mt(0, 2)) + 2 * r * (Mt(1, 2) - Mt(2, 1)) -
                          4 * x * (Mt(2, 2) + Mt(1, 1));
                drot[2] = 2 * x * (Mt(1, 0) + Mt(0, 1)) + 2 * r * (Mt(2, 0) - Mt(0, 2)) + 2 * z * (Mt(1, 2) + Mt(2, 1)) -
                          4 * y * (Mt(2, 2) + Mt(0, 0));
                drot[3] = 2 * r * (Mt(0, 1) - Mt(1, 0)) + 2 * x * (Mt(2, 0) + Mt(0, 2)) + 2 * y * (Mt(1, 2) + Mt(2, 1)) -
                          4 * z * (Mt(1, 1) + Mt(0, 0));
#undef Mt
            }
        } else if (use_sh) {
            for (int k = 0; k < L; k++) my_sh[k] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 3; i++) dL_dmean3D[3 * idx + i] = dmean[i];
#pragma unroll
        for (int i = 0; i < 6; i++) dL_dcov3D[6 * idx + i] = dcov[i];
#pragma unroll
        for (int i = 0; i < 3; i++) dL_dscale[3 * idx + i] = dscale[i];
        reinterpret_cast<float4*>(dL_drot)[idx] = make_float4(drot[0], drot[1], drot[2], drot[3]);
    }
    if (use_sh && dL_dsh != nullptr) {
        __syncthreads();
        const int total = cnt * L;
        float* dst = dL_dsh + (size_t)base * L;
        if (dL_dsh_rest != nullptr) {  // two outputs: dL_dsh = the DC rows (P,1,3), dL_dsh_rest = the rest (P,M-1,3)
            for (int i = threadIdx.x; i < cnt * 3; i += DGM_PRE_BLOCK) {
                const int g = i / 3, k = i - g * 3;
                dL_dsh[(size_t)base * 3 + i] = lds[g * stride + k];
            }
            const int Mr = L - 3;
            float* dr = dL_dsh_rest + (size_t)base * Mr;
            if (((cnt * Mr) & 3) == 0 && (((uintptr_t)dr) & 15) == 0) {
                float4* d4 = reinterpret_cast<float4*>(dr);
                for (int i = threadIdx.x; i < ((cnt * Mr) >> 2); i += DGM_PRE_BLOCK) {
                    float e4[4];
                    const int e = i << 2;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int g = (e + c) / Mr, k = (e + c) - g * Mr;
                        e4[c] = lds[g * stride + 3 + k];
                    }
                    d4[i] = make_float4(e4[0], e4[1], e4[2], e4[3]);
                }
            } else {
                for (int i = threadIdx.x; i < cnt * Mr; i += DGM_PRE_BLOCK) {
                    const int g = i / Mr, k = i - g * Mr;
                    dr[i] = lds[g * stride + 3 + k];
                }
            }
        } else if ((L & 3) == 0 && (((uintptr_t)dst) & 15) == 0) {
            float4* d4 = reinterpret_cast<float4*>(dst);
            for (int i = threadIdx.x; i < (total >> 2); i += DGM_PRE_BLOCK) {
                const int e = i << 2;
                const int g = e / L, k = e - g * L;
                const float* s = lds + g * stride + k;
                d4[i] = make_float4(s[0], s[1], s[2], s[3]);
            }
        } else {
            for (int i = threadIdx.x; i < total; i += DGM_PRE_BLOCK) {
                const int g = i / L, k = i - g * L;
                dst[i] = lds[g * stride + k];
            }
        }
    }
}

void launch_preprocess_bwd(hipStream_t st, int P, int D, int M, int gridx, const float* means3D, const int* radii,
                           const float* shs, const float* shs_rest, const uint8_t* clamped, const float* scales,
                           const float* rotations,
                           float scale_modifier, const float* cov3Ds, const float* viewmatrix, const float* projmatrix,
                           const float* campos, float focal_x, float focal_y, float tan_fovx, float tan_fovy, int W, int H,
                           const float* rec, const unsigned* tiles_touched, const unsigned* offs, const float* slab,
                           const uint8_t* live, float* dL_dmean2D,
                           float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                           float* dL_dsh, float* dL_dsh_rest, float* dL_dscale, float* dL_drot) {
    const size_t lds_bytes = (shs != nullptr && M > 0) ? (size_t)DGM_PRE_BLOCK * ((3 * M) | 1) * sizeof(float) : 16;
    const int nblk = (P + DGM_PRE_BLOCK - 1) / DGM_PRE_BLOCK;
    hipLaunchKernelGGL(preprocess_bwd_kernel, dim3(nblk), dim3(DGM_PRE_BLOCK), lds_bytes, st, P, D, M, gridx, means3D,
                       radii, shs, shs_rest, clamped, scales, rotations, scale_modifier, cov3Ds, viewmatrix, projmatrix, campos,
                       focal_x, focal_y, tan_fovx, tan_fovy, W, H, rec, tiles_touched, offs, slab, live,
                       dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dsh_rest, dL_dscale, dL_drot);
}

}  // namespace dgm
