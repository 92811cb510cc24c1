/*
 * dgr_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Scalar C restatement of the reference's differentiable 3D-Gaussian rasterizer
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the shipped HIP path never links or calls it.
 *
 * Parity status: the reference ships no golden vectors / tests for this path
 * (SURVEY.md section 4), and its CUDA sources cannot run in a CPU container, so this
 * restatement is pinned by (i) analytic known-answer tests, (ii) an autograd
 * cross-check of every gradient (tests/test_oracle_*.py) and (iii) -- on the GPU
 * box -- the reference's own kernels compiled from /root/reference into
 * oracle/_ref (see oracle/build_ref.sh).  Without (iii) read "parity unpinned".
 *
 * Canonical arithmetic (the parity definition): every reference
 * expression is evaluated in IEEE-754 binary32 in SOURCE ORDER, one rounding per
 * operation, NO fused multiply-add (build with -ffp-contract=off), correctly
 * rounded / and sqrt, GLM mat3 products as sum_k A(r,k)*B(k,c) for k = 0,1,2 left
 * to right.  ndc2Pix is evaluated in binary64 as in the reference.
 *
 * Each function cites the reference lines it follows.  Short-hands:
 *   DGR/ = dgmesh/submodules/diff-gaussian-rasterization/
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLOCK_X 16 /* DGR/cuda_rasterizer/config.h:16-17 */
#define BLOCK_Y 16
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)

/* DGR/cuda_rasterizer/auxiliary.h:22-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* float -> int as the GPU does it (cvt.rzi.s32.f32 / v_cvt_i32_f32): truncate toward
 * zero, saturate, NaN -> 0.  A plain C cast is undefined out of range. */
static int f2i_sat(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
/* CUDA fminf/fmaxf semantics: if one operand is NaN return the other. */
static float fmin_c(float a, float b) { return fminf(a, b); }
static float fmax_c(float a, float b) { return fmaxf(a, b); }

/* DGR/cuda_rasterizer/auxiliary.h:41-44 (double arithmetic, rounded to float once) */
static float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

/* DGR/cuda_rasterizer/auxiliary.h:46-56 */
static void getRect(float px, float py, int max_radius, int gx, int gy, uint32_t* rmin, uint32_t* rmax) {
    rmin[0] = (uint32_t)imin(gx, imax(0, f2i_sat((px - max_radius) / BLOCK_X)));
    rmin[1] = (uint32_t)imin(gy, imax(0, f2i_sat((py - max_radius) / BLOCK_Y)));
    rmax[0] = (uint32_t)imin(gx, imax(0, f2i_sat((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    rmax[1] = (uint32_t)imin(gy, imax(0, f2i_sat((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* DGR/cuda_rasterizer/auxiliary.h:58-77 */
static void transformPoint4x3(const float* p, const float* m, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void transformPoint4x4(const float* p, const float* m, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* Math-convention (row,col) 3x3 helpers.  GLM stores M[col][row]; every use below
 * spells out which (row,col) entry it means, see DESIGN.md section 3 and SURVEY A.2. */

/* DGR/cuda_rasterizer/forward.cu:118-152.  Mm(r,c) = s_r * Rm(r,c) with
 * Rm = GLM "R" read as a math matrix (= transpose of the usual quaternion rotation);
 * Sigma = Mm^T Mm; quaternion NOT normalised (forward.cu:127). */
static void quat_to_Rm(const float* q, float Rm[3][3]) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    Rm[0][0] = 1.f - 2.f * (y * y + z * z);
    Rm[1][0] = 2.f * (x * y - r * z);
    Rm[2][0] = 2.f * (x * z + r * y);
    Rm[0][1] = 2.f * (x * y + r * z);
    Rm[1][1] = 1.f - 2.f * (x * x + z * z);
    Rm[2][1] = 2.f * (y * z - r * x);
    Rm[0][2] = 2.f * (x * z - r * y);
    Rm[1][2] = 2.f * (y * z + r * x);
    Rm[2][2] = 1.f - 2.f * (x * x + y * y);
}
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D) {
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    float Rm[3][3], Mm[3][3];
    quat_to_Rm(rot, Rm);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) Mm[r][c] = s[r] * Rm[r][c];
    /* Sigma(r,c) = sum_k Mm(k,r)*Mm(k,c); stored [S00,S10,S20,S11,S21,S22] (GLM Sigma[0][1] = row 1, col 0) */
#define SIG(r, c) (Mm[0][r] * Mm[0][c] + Mm[1][r] * Mm[1][c] + Mm[2][r] * Mm[2][c])
    cov3D[0] = SIG(0, 0);
    cov3D[1] = SIG(1, 0);
    cov3D[2] = SIG(2, 0);
    cov3D[3] = SIG(1, 1);
    cov3D[4] = SIG(2, 1);
    cov3D[5] = SIG(2, 2);
#undef SIG
}

/* Shared by forward (forward.cu:74-113) and backward (backward.cu:160-199):
 * t (after the fov clamp), Tm = Wm*Jm (3x2 non-zero part), cov2D (a,b,c) incl. +0.3. */
typedef struct {
    float t[3];
    float txtz, tytz, limx, limy;
    float T[3][2]; /* Tm(r,c), c < 2; Tm(r,2) = 0 */
    float a, b, c;
} Cov2D;
static void cov2d_common(const float* mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                         const float* cov3D, const float* vm, Cov2D* o) {
    float t[3];
    transformPoint4x3(mean, vm, t);
    o->limx = 1.3f * tan_fovx;
    o->limy = 1.3f * tan_fovy;
    o->txtz = t[0] / t[2];
    o->tytz = t[1] / t[2];
    t[0] = fmin_c(o->limx, fmax_c(-o->limx, o->txtz)) * t[2];
    t[1] = fmin_c(o->limy, fmax_c(-o->limy, o->tytz)) * t[2];
    o->t[0] = t[0];
    o->t[1] = t[1];
    o->t[2] = t[2];
    float J00 = focal_x / t[2];
    float J20 = -(focal_x * t[0]) / (t[2] * t[2]);
    float J11 = focal_y / t[2];
    float J21 = -(focal_y * t[1]) / (t[2] * t[2]);
    /* Wm(r,k): Wm = [[v0,v1,v2],[v4,v5,v6],[v8,v9,v10]] */
    for (int r = 0; r < 3; r++) {
        float w0 = vm[4 * r + 0], w1 = vm[4 * r + 1], w2 = vm[4 * r + 2];
        /* Tm(r,0) = w0*J00 + w1*0 + w2*J20 ; Tm(r,1) = w0*0 + w1*J11 + w2*J21 */
        o->T[r][0] = (w0 * J00 + w1 * 0.0f) + w2 * J20;
        o->T[r][1] = (w0 * 0.0f + w1 * J11) + w2 * J21;
    }
    /* Vm symmetric from cov3D */
    float V[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}};
    /* U = Tm^T * Vm^T : U(r,k) = Tm(0,r)*V(k,0) + Tm(1,r)*V(k,1) + Tm(2,r)*V(k,2) */
    float U[2][3];
    for (int r = 0; r < 2; r++)
        for (int k = 0; k < 3; k++) U[r][k] = o->T[0][r] * V[k][0] + o->T[1][r] * V[k][1] + o->T[2][r] * V[k][2];
    /* cov(r,c) = U(r,0)*Tm(0,c) + U(r,1)*Tm(1,c) + U(r,2)*Tm(2,c); GLM cov[0][1] = cov(1,0) */
    o->a = U[0][0] * o->T[0][0] + U[0][1] * o->T[1][0] + U[0][2] * o->T[2][0];
    o->b = U[1][0] * o->T[0][0] + U[1][1] * o->T[1][0] + U[1][2] * o->T[2][0];
    o->c = U[1][0] * o->T[0][1] + U[1][1] * o->T[1][1] + U[1][2] * o->T[2][1];
    o->a += 0.3f;
    o->c += 0.3f;
}

/* SH basis up to degree 3 for a unit direction; forward.cu:30-59 / backward.cu:47-90 */
static void sh_basis(int deg, float x, float y, float z, float* B) {
    B[0] = SH_C0;
    if (deg > 0) {
        B[1] = -SH_C1 * y;
        B[2] = SH_C1 * z;
        B[3] = -SH_C1 * x;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = SH_C2[0] * xy;
            B[5] = SH_C2[1] * yz;
            B[6] = SH_C2[2] * (2.0f * zz - xx - yy);
            B[7] = SH_C2[3] * xz;
            B[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                B[9] = SH_C3[0] * y * (3.0f * xx - yy);
                B[10] = SH_C3[1] * xy * z;
                B[11] = SH_C3[2] * y * (4.0f * zz - xx - yy);
                B[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                B[13] = SH_C3[4] * x * (4.0f * zz - xx - yy);
                B[14] = SH_C3[5] * z * (xx - yy);
                B[15] = SH_C3[6] * x * (xx - 3.0f * yy);
            }
        }
    }
}

/* DGR/cuda_rasterizer/forward.cu:20-71.  sh layout (P, M, 3): coefficient-major, RGB innermost. */
static void computeColorFromSH(int idx, int deg, int M, const float* means, const float* campos, const float* shs,
                               uint8_t* clamped, float* rgb) {
    float dir[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    float x = dir[0] / len, y = dir[1] / len, z = dir[2] / len;
    const float* sh = shs + (size_t)idx * M * 3;
    float B[16];
    sh_basis(deg, x, y, z, B);
    int n = (deg + 1) * (deg + 1);
    for (int ch = 0; ch < 3; ch++) {
        /* result = C0*sh0; result = result - C1*y*sh1 + C1*z*sh2 - C1*x*sh3 ; ... (left to right) */
        float res = B[0] * sh[ch];
        for (int k = 1; k < n; k++) res = res + B[k] * sh[3 * k + ch];
        res += 0.5f;
        clamped[3 * idx + ch] = (res < 0);
        rgb[3 * idx + ch] = fmax_c(res, 0.0f);
    }
}

/* DGR/cuda_rasterizer/forward.cu:156-256 (preprocessCUDA) + auxiliary.h:139-164 (in_frustum).
 * Outputs are left untouched (caller zero-fills) for culled Gaussians, except radii/tiles_touched = 0. */
void orc_preprocess_fwd(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                        const float* rotations, const float* opacities, const float* shs, const float* cov3D_precomp,
                        const float* colors_precomp, const float* viewmatrix, const float* projmatrix,
                        const float* cam_pos, int W, int H, float tan_fovx, float tan_fovy, int* radii,
                        float* means2D, float* depths, float* cov3Ds, float* rgb, float* conic_opacity,
                        uint32_t* tiles_touched, uint8_t* clamped) {
    const float focal_y = H / (2.0f * tan_fovy); /* DGR/cuda_rasterizer/rasterizer_impl.cu:222-223 */
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        const float* p_orig = means3D + 3 * idx;
        float p_view[3], p_hom[4];
        transformPoint4x3(p_orig, viewmatrix, p_view);
        if (p_view[2] <= 0.2f) continue; /* auxiliary.h:154 */
        transformPoint4x4(p_orig, projmatrix, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};
        const float* cov3D;
        if (cov3D_precomp) {
            cov3D = cov3D_precomp + 6 * idx;
        } else {
            computeCov3D(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3Ds + 6 * idx);
            cov3D = cov3Ds + 6 * idx;
        }
        Cov2D cv;
        cov2d_common(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &cv);
        float det = (cv.a * cv.c - cv.b * cv.b);
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {cv.c * det_inv, -cv.b * det_inv, cv.a * det_inv};
        float mid = 0.5f * (cv.a + cv.c);
        float lambda1 = mid + sqrtf(fmax_c(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmax_c(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmax_c(lambda1, lambda2)));
        float px = ndc2Pix(p_proj[0], W), py = ndc2Pix(p_proj[1], H);
        uint32_t rmin[2], rmax[2];
        getRect(px, py, f2i_sat(my_radius), gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (!colors_precomp) computeColorFromSH(idx, D, M, means3D, cam_pos, shs, clamped, rgb);
        depths[idx] = p_view[2];
        radii[idx] = f2i_sat(my_radius);
        means2D[2 * idx] = px;
        means2D[2 * idx + 1] = py;
        conic_opacity[4 * idx + 0] = conic[0];
        conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2];
        conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
    }
}

/* DGR/cuda_rasterizer/rasterizer_impl.cu:54-66 (checkFrustum / markVisible) */
void orc_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present) {
    (void)projmatrix;
    for (int idx = 0; idx < P; idx++) {
        float p_view[3];
        transformPoint4x3(means3D + 3 * idx, viewmatrix, p_view);
        present[idx] = !(p_view[2] <= 0.2f);
    }
}

/* cub::DeviceScan::InclusiveSum, rasterizer_impl.cu:277; returns num_rendered (rasterizer_impl.cu:281) */
uint32_t orc_inclusive_scan(int P, const uint32_t* tiles_touched, uint32_t* point_offsets) {
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) {
        acc += tiles_touched[i];
        point_offsets[i] = acc;
    }
    return acc;
}

/* DGR/cuda_rasterizer/rasterizer_impl.cu:35-50 */
uint32_t orc_get_higher_msb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb)
            msb += step;
        else
            msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* duplicateWithKeys (rasterizer_impl.cu:70-111) + cub::DeviceRadixSort::SortPairs (stable LSD
 * radix sort, rasterizer_impl.cu:303-308) + memset/identifyTileRanges (rasterizer_impl.cu:310-317, 116-138).
 * keys_sorted / point_list have R entries, ranges has 2*tiles entries. */
void orc_bin(int P, int W, int H, const float* means2D, const float* depths, const int* radii,
             const uint32_t* point_offsets, uint32_t R, uint64_t* keys_sorted, uint32_t* point_list,
             uint32_t* ranges) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (R ? R : 1));
    uint32_t* vals = (uint32_t*)malloc(sizeof(uint32_t) * (R ? R : 1));
    uint64_t* keys2 = (uint64_t*)malloc(sizeof(uint64_t) * (R ? R : 1));
    uint32_t* vals2 = (uint32_t*)malloc(sizeof(uint32_t) * (R ? R : 1));
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : point_offsets[idx - 1];
            uint32_t rmin[2], rmax[2];
            getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
            uint32_t dbits;
            memcpy(&dbits, &depths[idx], 4);
            for (uint32_t y = rmin[1]; y < rmax[1]; y++)
                for (uint32_t x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * (uint32_t)gx + x);
                    key <<= 32;
                    key |= dbits;
                    keys[off] = key;
                    vals[off] = (uint32_t)idx;
                    off++;
                }
        }
    }
    /* stable LSD radix sort, 8 passes of 8 bits (cub sorts bits [0, 32+bit); higher bits are zero) */
    uint64_t *ka = keys, *kb = keys2;
    uint32_t *va = vals, *vb = vals2;
    for (int pass = 0; pass < 8; pass++) {
        size_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        int sh = 8 * pass;
        for (uint32_t i = 0; i < R; i++) cnt[((ka[i] >> sh) & 0xff) + 1]++;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (uint32_t i = 0; i < R; i++) {
            size_t dst = cnt[(ka[i] >> sh) & 0xff]++;
            kb[dst] = ka[i];
            vb[dst] = va[i];
        }
        uint64_t* tk = ka;
        ka = kb;
        kb = tk;
        uint32_t* tv = va;
        va = vb;
        vb = tv;
    }
    memcpy(keys_sorted, ka, sizeof(uint64_t) * R);
    memcpy(point_list, va, sizeof(uint32_t) * R);
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
    for (uint32_t i = 0; i < R; i++) {
        uint32_t cur = (uint32_t)(keys_sorted[i] >> 32);
        if (i == 0)
            ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(keys_sorted[i - 1] >> 32);
            if (cur != prev) {
                ranges[2 * prev + 1] = i;
                ranges[2 * cur] = i;
            }
        }
        if (i == R - 1) ranges[2 * cur + 1] = R;
    }
    free(keys);
    free(vals);
    free(keys2);
    free(vals2);
}

/* DGR/cuda_rasterizer/forward.cu:263-374 (renderCUDA forward).
 * fragile (optional, per pixel): set when a decision of the blend loop sits within rounding
 * distance of its threshold, i.e. when an implementation with a different exp() or FMA placement
 * may legitimately take the other branch.  bit 0: `power > 0` (|power| <= 1e-5 * sum|terms|) or
 * `alpha < 1/255` (|255*alpha - 1| < 2e-5): changes the colour by up to alpha*T.  bit 1:
 * `T' < 1e-4` (|1e4*T' - 1| < 1e-4): changes n_contrib, the colour by < 1e-4. */
void orc_render_fwd(const uint32_t* ranges, const uint32_t* point_list, int W, int H, const float* means2D,
                    const float* colors, const float* conic_opacity, const float* bg, float* out_color,
                    float* final_T, uint32_t* n_contrib, uint8_t* fragile) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; tile++) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
                if (pxi >= W || pyi >= H) continue;
                float pixx = (float)pxi, pixy = (float)pyi;
                float T = 1.0f, C[3] = {0, 0, 0};
                uint32_t contributor = 0, last = 0;
                uint8_t frag = 0;
                for (uint32_t s = r0; s < r1; s++) {
                    /* NB: a pixel keeps counting/"fetching" until the whole tile is done in the reference,
                     * but nothing observable depends on that once done is set. */
                    contributor++;
                    uint32_t id = point_list[s];
                    float dx = means2D[2 * id] - pixx, dy = means2D[2 * id + 1] - pixy;
                    const float* co = conic_opacity + 4 * id;
                    float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    float mag = 0.5f * (fabsf(co[0]) * dx * dx + fabsf(co[2]) * dy * dy) + fabsf(co[1] * dx * dy);
                    if (fabsf(power) <= 1e-5f * mag && mag > 0.0f) frag |= 1;
                    if (power > 0.0f) continue;
                    float alpha = fmin_c(0.99f, co[3] * expf(power));
                    if (fabsf(alpha * 255.0f - 1.0f) < 2e-5f) frag |= 1;
                    if (alpha < 1.0f / 255.0f) continue;
                    float test_T = T * (1 - alpha);
                    if (fabsf(test_T * 10000.0f - 1.0f) < 1e-4f) frag |= 2;
                    if (test_T < 0.0001f) break; /* done = true */
                    for (int ch = 0; ch < 3; ch++) C[ch] += colors[3 * id + ch] * alpha * T;
                    T = test_T;
                    last = contributor;
                }
                size_t pid = (size_t)W * pyi + pxi;
                final_T[pid] = T;
                n_contrib[pid] = last;
                for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pid] = C[ch] + T * bg[ch];
                if (fragile) fragile[pid] = frag;
            }
    }
}

/* DGR/cuda_rasterizer/backward.cu:401-557 (renderCUDA backward).  The reference accumulates with
 * fp32 atomicAdd in an unspecified order; the oracle accumulates the same fp32 terms in binary64
 * (tile-private, then reduced in tile order) and rounds once, i.e. it is the order-free sum.
 * dL_dmean2D: (P,3) (z untouched), dL_dconic: (P,4) slots x,y,w, dL_dopacity: (P), dL_dcolors: (P,3).
 * All four are OVERWRITTEN (zero for Gaussians that receive nothing). */
void orc_render_bwd(int P, const uint32_t* ranges, const uint32_t* point_list, int W, int H, const float* bg,
                    const float* means2D, const float* conic_opacity, const float* colors, const float* final_Ts,
                    const uint32_t* n_contrib, const float* dL_dpixels, float* dL_dmean2D, float* dL_dconic,
                    float* dL_dopacity, float* dL_dcolors) {
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    /* thread-private binary64 accumulators (<= 32 threads), reduced in thread order afterwards */
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
    if (nthreads > 32) nthreads = 32;
#endif
    double* acc_all = (double*)calloc((size_t)nthreads * P * 9, sizeof(double));
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int tile = 0; tile < gx * gy; tile++) {
        int tid_ = 0;
#ifdef _OPENMP
        tid_ = omp_get_thread_num();
#endif
        double* acc = acc_all + (size_t)tid_ * P * 9;
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        if (r0 == r1) continue;
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
                if (pxi >= W || pyi >= H) continue;
                size_t pid = (size_t)W * pyi + pxi;
                float pixx = (float)pxi, pixy = (float)pyi;
                const float T_final = final_Ts[pid];
                float T = T_final;
                const uint32_t last_contributor = n_contrib[pid];
                float accum_rec[3] = {0, 0, 0}, dL_dpixel[3], last_alpha = 0, last_color[3] = {0, 0, 0};
                for (int ch = 0; ch < 3; ch++) dL_dpixel[ch] = dL_dpixels[(size_t)ch * H * W + pid];
                for (uint32_t s = r0 + (last_contributor < r1 - r0 ? last_contributor : r1 - r0); s-- > r0;) {
                    uint32_t id = point_list[s];
                    float dx = means2D[2 * id] - pixx, dy = means2D[2 * id + 1] - pixy;
                    const float* co = conic_opacity + 4 * id;
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float G = expf(power);
                    const float alpha = fmin_c(0.99f, co[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;
                    float dL_dalpha = 0.0f;
                    double* a = acc + (size_t)id * 9;
                    for (int ch = 0; ch < 3; ch++) {
                        const float c = colors[3 * id + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        const float dL_dchannel = dL_dpixel[ch];
                        dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                        a[ch] += (double)(dchannel_dcolor * dL_dchannel);
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    float bg_dot_dpixel = 0;
                    for (int i = 0; i < 3; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                    const float dL_dG = co[3] * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const float dG_ddely = -gdy * co[2] - gdx * co[1];
                    a[3] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                    a[4] += (double)(dL_dG * dG_ddely * ddely_dy);
                    a[5] += (double)(-0.5f * gdx * dx * dL_dG);
                    a[6] += (double)(-0.5f * gdx * dy * dL_dG);
                    a[7] += (double)(-0.5f * gdy * dy * dL_dG);
                    a[8] += (double)(G * dL_dalpha);
                }
            }
    }
    for (int i = 0; i < P; i++) {
        double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = 0; t < nthreads; t++)
            for (int k = 0; k < 9; k++) a[k] += acc_all[((size_t)t * P + i) * 9 + k];
        dL_dcolors[3 * i] = (float)a[0];
        dL_dcolors[3 * i + 1] = (float)a[1];
        dL_dcolors[3 * i + 2] = (float)a[2];
        dL_dmean2D[3 * i] = (float)a[3];
        dL_dmean2D[3 * i + 1] = (float)a[4];
        dL_dmean2D[3 * i + 2] = 0.0f;
        dL_dconic[4 * i] = (float)a[5];
        dL_dconic[4 * i + 1] = (float)a[6];
        dL_dconic[4 * i + 2] = 0.0f;
        dL_dconic[4 * i + 3] = (float)a[7];
        dL_dopacity[i] = (float)a[8];
    }
    free(acc_all);
}

/* DGR/cuda_rasterizer/backward.cu:144-274 (computeCov2DCUDA).  Writes dL_dcov (P,6) and ASSIGNS
 * dL_dmeans (P,3) for radii > 0; other rows untouched (caller zero-fills). */
void orc_cov2d_bwd(int P, const float* means, const int* radii, const float* cov3Ds, float h_x, float h_y,
                   float tan_fovx, float tan_fovy, const float* vm, const float* dL_dconics, float* dL_dmeans,
                   float* dL_dcov) {
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float* cov3D = cov3Ds + 6 * idx;
        float g[3] = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
        Cov2D cv;
        cov2d_common(means + 3 * idx, h_x, h_y, tan_fovx, tan_fovy, cov3D, vm, &cv);
        const float x_grad_mul = (cv.txtz < -cv.limx || cv.txtz > cv.limx) ? 0.f : 1.f;
        const float y_grad_mul = (cv.tytz < -cv.limy || cv.tytz > cv.limy) ? 0.f : 1.f;
        /* GLM T[i][j] = Tm(j,i) */
#define TT(i, j) (cv.T[j][i])
        float V[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}};
        float a = cv.a, b = cv.b, c = cv.c;
        float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float* dc = dL_dcov + 6 * idx;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * g[0] + 2 * b * c * g[1] + (denom - a * c) * g[2]);
            dL_dc = denom2inv * (-a * a * g[2] + 2 * a * b * g[1] + (denom - a * c) * g[0]);
            dL_db = denom2inv * 2 * (b * c * g[0] - (denom + 2 * b * b) * g[1] + a * b * g[2]);
            dc[0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
            dc[3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
            dc[5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
            dc[1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db +
                    2 * TT(1, 0) * TT(1, 1) * dL_dc;
            dc[2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db +
                    2 * TT(1, 0) * TT(1, 2) * dL_dc;
            dc[4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db +
                    2 * TT(1, 1) * TT(1, 2) * dL_dc;
        } else {
            for (int i = 0; i < 6; i++) dc[i] = 0;
        }
        float dL_dT00 = 2 * (TT(0, 0) * V[0][0] + TT(0, 1) * V[0][1] + TT(0, 2) * V[0][2]) * dL_da +
                        (TT(1, 0) * V[0][0] + TT(1, 1) * V[0][1] + TT(1, 2) * V[0][2]) * dL_db;
        float dL_dT01 = 2 * (TT(0, 0) * V[1][0] + TT(0, 1) * V[1][1] + TT(0, 2) * V[1][2]) * dL_da +
                        (TT(1, 0) * V[1][0] + TT(1, 1) * V[1][1] + TT(1, 2) * V[1][2]) * dL_db;
        float dL_dT02 = 2 * (TT(0, 0) * V[2][0] + TT(0, 1) * V[2][1] + TT(0, 2) * V[2][2]) * dL_da +
                        (TT(1, 0) * V[2][0] + TT(1, 1) * V[2][1] + TT(1, 2) * V[2][2]) * dL_db;
        float dL_dT10 = 2 * (TT(1, 0) * V[0][0] + TT(1, 1) * V[0][1] + TT(1, 2) * V[0][2]) * dL_dc +
                        (TT(0, 0) * V[0][0] + TT(0, 1) * V[0][1] + TT(0, 2) * V[0][2]) * dL_db;
        float dL_dT11 = 2 * (TT(1, 0) * V[1][0] + TT(1, 1) * V[1][1] + TT(1, 2) * V[1][2]) * dL_dc +
                        (TT(0, 0) * V[1][0] + TT(0, 1) * V[1][1] + TT(0, 2) * V[1][2]) * dL_db;
        float dL_dT12 = 2 * (TT(1, 0) * V[2][0] + TT(1, 1) * V[2][1] + TT(1, 2) * V[2][2]) * dL_dc +
                        (TT(0, 0) * V[2][0] + TT(0, 1) * V[2][1] + TT(0, 2) * V[2][2]) * dL_db;
#undef TT
        /* GLM W[i][j] = Wm(j,i) = vm[4*j+i] */
#define WW(i, j) (vm[4 * (j) + (i)])
        float dL_dJ00 = WW(0, 0) * dL_dT00 + WW(0, 1) * dL_dT01 + WW(0, 2) * dL_dT02;
        float dL_dJ02 = WW(2, 0) * dL_dT00 + WW(2, 1) * dL_dT01 + WW(2, 2) * dL_dT02;
        float dL_dJ11 = WW(1, 0) * dL_dT10 + WW(1, 1) * dL_dT11 + WW(1, 2) * dL_dT12;
        float dL_dJ12 = WW(2, 0) * dL_dT10 + WW(2, 1) * dL_dT11 + WW(2, 2) * dL_dT12;
#undef WW
        float tz = 1.f / cv.t[2];
        float tz2 = tz * tz;
        float tz3 = tz2 * tz;
        float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * cv.t[0]) * tz3 * dL_dJ02 +
                       (2 * h_y * cv.t[1]) * tz3 * dL_dJ12;
        /* transformVec4x3Transpose, auxiliary.h:89-97 */
        dL_dmeans[3 * idx + 0] = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
        dL_dmeans[3 * idx + 1] = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
        dL_dmeans[3 * idx + 2] = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;
    }
}

/* DGR/cuda_rasterizer/backward.cu:20-139 (computeColorFromSH backward) */
static void sh_bwd(int idx, int deg, int M, const float* means, const float* campos, const float* shs,
                   const uint8_t* clamped, const float* dL_dcolor, float* dL_dmeans, float* dL_dshs) {
    float dir_orig[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
    float x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
    const float* sh = shs + (size_t)idx * M * 3;
    float* dsh = dL_dshs + (size_t)idx * M * 3;
    float dRGB[3];
    for (int ch = 0; ch < 3; ch++) dRGB[ch] = dL_dcolor[3 * idx + ch] * (clamped[3 * idx + ch] ? 0.f : 1.f);
    float B[16];
    sh_basis(deg, x, y, z, B);
    int n = (deg + 1) * (deg + 1);
    for (int k = 0; k < n; k++)
        for (int ch = 0; ch < 3; ch++) dsh[3 * k + ch] = B[k] * dRGB[ch];
    float dx[3] = {0, 0, 0}, dy[3] = {0, 0, 0}, dz[3] = {0, 0, 0};
#define S(k) sh[3 * (k) + ch]
    for (int ch = 0; ch < 3; ch++) {
        if (deg > 0) {
            dx[ch] = -SH_C1 * S(3);
            dy[ch] = -SH_C1 * S(1);
            dz[ch] = SH_C1 * S(2);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                dx[ch] += SH_C2[0] * y * S(4) + SH_C2[2] * 2.f * -x * S(6) + SH_C2[3] * z * S(7) + SH_C2[4] * 2.f * x * S(8);
                dy[ch] += SH_C2[0] * x * S(4) + SH_C2[1] * z * S(5) + SH_C2[2] * 2.f * -y * S(6) + SH_C2[4] * 2.f * -y * S(8);
                dz[ch] += SH_C2[1] * y * S(5) + SH_C2[2] * 2.f * 2.f * z * S(6) + SH_C2[3] * x * S(7);
                if (deg > 2) {
                    dx[ch] += (SH_C3[0] * S(9) * 3.f * 2.f * xy + SH_C3[1] * S(10) * yz + SH_C3[2] * S(11) * -2.f * xy +
                               SH_C3[3] * S(12) * -3.f * 2.f * xz + SH_C3[4] * S(13) * (-3.f * xx + 4.f * zz - yy) +
                               SH_C3[5] * S(14) * 2.f * xz + SH_C3[6] * S(15) * 3.f * (xx - yy));
                    dy[ch] += (SH_C3[0] * S(9) * 3.f * (xx - yy) + SH_C3[1] * S(10) * xz +
                               SH_C3[2] * S(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * S(12) * -3.f * 2.f * yz +
                               SH_C3[4] * S(13) * -2.f * xy + SH_C3[5] * S(14) * -2.f * yz +
                               SH_C3[6] * S(15) * -3.f * 2.f * xy);
                    dz[ch] += (SH_C3[1] * S(10) * xy + SH_C3[2] * S(11) * 4.f * 2.f * yz +
                               SH_C3[3] * S(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * S(13) * 4.f * 2.f * xz +
                               SH_C3[5] * S(14) * (xx - yy));
                }
            }
        }
    }
#undef S
    float ddir[3] = {dx[0] * dRGB[0] + dx[1] * dRGB[1] + dx[2] * dRGB[2], dy[0] * dRGB[0] + dy[1] * dRGB[1] + dy[2] * dRGB[2],
                     dz[0] * dRGB[0] + dz[1] * dRGB[1] + dz[2] * dRGB[2]};
    /* dnormvdv, auxiliary.h:107-117 */
    const float* v = dir_orig;
    float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    float o0 = ((+sum2 - v[0] * v[0]) * ddir[0] - v[1] * v[0] * ddir[1] - v[2] * v[0] * ddir[2]) * invsum32;
    float o1 = (-v[0] * v[1] * ddir[0] + (sum2 - v[1] * v[1]) * ddir[1] - v[2] * v[1] * ddir[2]) * invsum32;
    float o2 = (-v[0] * v[2] * ddir[0] - v[1] * v[2] * ddir[1] + (sum2 - v[2] * v[2]) * ddir[2]) * invsum32;
    dL_dmeans[3 * idx + 0] += o0;
    dL_dmeans[3 * idx + 1] += o1;
    dL_dmeans[3 * idx + 2] += o2;
}

/* DGR/cuda_rasterizer/backward.cu:278-341 (computeCov3D backward) */
static void cov3d_bwd(int idx, const float* scale, float mod, const float* rot, const float* dL_dcov3Ds,
                      float* dL_dscales, float* dL_drots) {
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    float Rm[3][3], Mm[3][3];
    quat_to_Rm(rot, Rm);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) Mm[a][b] = s[a] * Rm[a][b];
    const float* d = dL_dcov3Ds + 6 * idx;
    float D[3][3] = {{d[0], 0.5f * d[1], 0.5f * d[2]}, {0.5f * d[1], d[3], 0.5f * d[4]}, {0.5f * d[2], 0.5f * d[4], d[5]}};
    /* dL_dM = (2*M) * dL_dSigma : G(r,c) = sum_k (2*Mm(r,k)) * D(k,c) */
    float G[3][3];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++)
            G[a][b] = (2.0f * Mm[a][0]) * D[0][b] + (2.0f * Mm[a][1]) * D[1][b] + (2.0f * Mm[a][2]) * D[2][b];
    /* Rt[k] = row k of Rm ; dL_dMt[k] = row k of G */
    for (int k = 0; k < 3; k++) dL_dscales[3 * idx + k] = Rm[k][0] * G[k][0] + Rm[k][1] * G[k][1] + Rm[k][2] * G[k][2];
    float Hh[3][3];
    for (int k = 0; k < 3; k++)
        for (int j = 0; j < 3; j++) Hh[k][j] = G[k][j] * s[k];
#define Mt(i, j) Hh[i][j]
    float q0 = 2 * z * (Mt(0, 1) - Mt(1, 0)) + 2 * y * (Mt(2, 0) - Mt(0, 2)) + 2 * x * (Mt(1, 2) - Mt(2, 1));
    float q1 = 2 * y * (Mt(1, 0) + Mt(0, 1)) + 2 * z * (Mt(2, 0) + Mt(0, 2)) + 2 * r * (Mt(1, 2) - Mt(2, 1)) -
               4 * x * (Mt(2, 2) + Mt(1, 1));
    float q2 = 2 * x * (Mt(1, 0) + Mt(0, 1)) + 2 * r * (Mt(2, 0) - Mt(0, 2)) + 2 * z * (Mt(1, 2) + Mt(2, 1)) -
               4 * y * (Mt(2, 2) + Mt(0, 0));
    float q3 = 2 * r * (Mt(0, 1) - Mt(1, 0)) + 2 * x * (Mt(2, 0) + Mt(0, 2)) + 2 * y * (Mt(1, 2) + Mt(2, 1)) -
               4 * z * (Mt(1, 1) + Mt(0, 0));
#undef Mt
    dL_drots[4 * idx + 0] = q0;
    dL_drots[4 * idx + 1] = q1;
    dL_drots[4 * idx + 2] = q2;
    dL_drots[4 * idx + 3] = q3;
}

/* DGR/cuda_rasterizer/backward.cu:347-396 (preprocessCUDA backward).  dL_dmeans is ACCUMULATED
 * (+=) on top of what orc_cov2d_bwd assigned; shs / scales may be NULL (colors / cov3D precomputed). */
void orc_preprocess_bwd(int P, int D, int M, const float* means, const int* radii, const float* shs,
                        const uint8_t* clamped, const float* scales, const float* rotations, float scale_modifier,
                        const float* proj, const float* campos, const float* dL_dmean2D, float* dL_dmeans,
                        const float* dL_dcolor, const float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot) {
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float* m = means + 3 * idx;
        float m_hom[4];
        transformPoint4x4(m, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
        float gx = dL_dmean2D[3 * idx], gy = dL_dmean2D[3 * idx + 1];
        float d0 = (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
        float d1 = (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
        float d2 = (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
        dL_dmeans[3 * idx + 0] += d0;
        dL_dmeans[3 * idx + 1] += d1;
        dL_dmeans[3 * idx + 2] += d2;
        if (shs) sh_bwd(idx, D, M, means, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
        if (scales) cov3d_bwd(idx, scales + 3 * idx, scale_modifier, rotations + 4 * idx, dL_dcov3D, dL_dscale, dL_drot);
    }
}
