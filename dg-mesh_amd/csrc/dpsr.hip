// Differentiable Poisson surface reconstruction (DPSR) pieces for gfx950: oriented points -> indicator grid.
//
// Replaces the index-tensor formulation of R/nvdiffrast_utils/dpsr_utils.py (point_rasterize :143-198 with
// scatter_to_grid :120-141, grid_interp :69-118) and the spectral solve of R/nvdiffrast_utils/dpsr.py:28-69 between its two
// FFTs (which stay rocFFT, reached through torch.fft on the caller's stream):
//   * splat:    raster[f][cell] += w_c(p) * N[p][f] over the 8 periodic neighbours of point p -- one thread per point, 24
//               fp32 atomics; the reference materialises (n, 8, 3, 5) int64 index tensors (~1 KB per point) first;
//   * spectral: Phi(k) = sum_d (-i c_d(k)) Nhat_d(k),  c_d = omega_d G / (Lap + 1e-6),  omega = 2 pi fftfreq,
//               G = exp(-0.5 (2 sig |f| / res)^2) (evaluated in binary64 like spec_gaussian_filter :56-62),
//               Lap = -|omega|^2, Phi(0) = 0 -- one pass over the half spectrum instead of ~12 elementwise kernels; the
//               adjoint (i c_d) dPhi is the same kernel with a flag;
//   * interp:   trilinear read-back of phi at the points (the shift / scale normalisation, dpsr.py:56-69).
// Each has its backward kernel (d/dN, d/dV through the trilinear weights exactly as autograd differentiates
// prod_d |p_d - pos_d| / cube, and d/dgrid), so the whole DPSR stays differentiable w.r.t. points and normals.
// Index / weight arithmetic follows the reference expression by expression (cube = 1 / res in fp32, floor(p / cube),
// fmod(ceil(p / cube), res)), so cells and weights match it bit for bit; sums differ by atomic order only.
#include "dgm_common.hpp"

namespace dgm {
void set_last_error(const char* msg);  // c_api.hip

struct Corner8 {
    int idx[8];
    float w[8];
    float dw[8][3];  // d w / d p_d
};

// the 8 neighbours of p in a periodic res^3 grid, weights and their derivatives (dpsr_utils.py:160-181)
__device__ __forceinline__ void corners(const float* __restrict__ p, int R, Corner8& c) {
    const float size = (float)R;
    const float cube = 1.0f / size;
    int i0[3], i1[3];
    float a0[3], a1[3], s0[3], s1[3];  // |p - pos| / cube for the low / high corner and the sign of (p - pos)
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float q = p[d] / cube;
        const float f = floorf(q);
        // (the reference asserts on the device for a coordinate outside [0, 1): its caller clamps; here both corner indices are
        // wrapped into the periodic grid and a NaN coordinate lands in cell 0 -- never an out-of-bounds access)
        int lo = f == f ? (int)fmodf(f, size) : 0;
        i0[d] = lo < 0 ? lo + R : (lo >= R ? R - 1 : lo);
        int hi = q == q ? (int)fmodf(ceilf(q), size) : 0;
        i1[d] = hi < 0 ? hi + R : (hi >= R ? R - 1 : hi);
        const float x0 = f * cube, x1 = (f + 1.0f) * cube;
        // weight of the LOW corner uses the distance to the HIGH corner position and vice versa
        const float e0 = p[d] - x1, e1 = p[d] - x0;
        a0[d] = fabsf(e0) / cube, a1[d] = fabsf(e1) / cube;
        s0[d] = (e0 > 0.f ? 1.f : (e0 < 0.f ? -1.f : 0.f)) / cube;
        s1[d] = (e1 > 0.f ? 1.f : (e1 < 0.f ? -1.f : 0.f)) / cube;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {  // com_ order of the reference: bit of dim 0 is the most significant
        const int b0 = (k >> 2) & 1, b1 = (k >> 1) & 1, b2 = k & 1;
        const int ix = b0 ? i1[0] : i0[0], iy = b1 ? i1[1] : i0[1], iz = b2 ? i1[2] : i0[2];
        const float wx = b0 ? a1[0] : a0[0], wy = b1 ? a1[1] : a0[1], wz = b2 ? a1[2] : a0[2];
        c.idx[k] = (ix * R + iy) * R + iz;
        c.w[k] = wx * wy * wz;
        c.dw[k][0] = (b0 ? s1[0] : s0[0]) * wy * wz;
        c.dw[k][1] = wx * (b1 ? s1[1] : s0[1]) * wz;
        c.dw[k][2] = wx * wy * (b2 ? s1[2] : s0[2]);
    }
}

__global__ void __launch_bounds__(256)
dpsr_splat_fwd_kernel(int n, int R, const float* __restrict__ V, const float* __restrict__ N, float* __restrict__ grid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Corner8 c;
    corners(V + 3 * i, R, c);
    const size_t cells = (size_t)R * R * R;
    const float n0 = N[3 * i], n1 = N[3 * i + 1], n2 = N[3 * i + 2];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        atomicAdd(grid + c.idx[k], c.w[k] * n0);
        atomicAdd(grid + cells + c.idx[k], c.w[k] * n1);
        atomicAdd(grid + 2 * cells + c.idx[k], c.w[k] * n2);
    }
}

__global__ void __launch_bounds__(256)
dpsr_splat_bwd_kernel(int n, int R, const float* __restrict__ V, const float* __restrict__ N, const float* __restrict__ dgrid,
                      float* __restrict__ dV, float* __restrict__ dN) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Corner8 c;
    corners(V + 3 * i, R, c);
    const size_t cells = (size_t)R * R * R;
    const float n0 = N[3 * i], n1 = N[3 * i + 1], n2 = N[3 * i + 2];
    float gn[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float g0 = dgrid[c.idx[k]], g1 = dgrid[cells + c.idx[k]], g2 = dgrid[2 * cells + c.idx[k]];
        gn[0] += c.w[k] * g0, gn[1] += c.w[k] * g1, gn[2] += c.w[k] * g2;
        const float s = g0 * n0 + g1 * n1 + g2 * n2;
        gv[0] += c.dw[k][0] * s, gv[1] += c.dw[k][1] * s, gv[2] += c.dw[k][2] * s;
    }
#pragma unroll
    for (int d = 0; d < 3; d++) dN[3 * i + d] = gn[d], dV[3 * i + d] = gv[d];
}

__global__ void __launch_bounds__(256)
dpsr_interp_fwd_kernel(int n, int R, const float* __restrict__ phi, const float* __restrict__ V, float* __restrict__ fv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Corner8 c;
    corners(V + 3 * i, R, c);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) s += phi[c.idx[k]] * c.w[k];
    fv[i] = s;
}

__global__ void __launch_bounds__(256)
dpsr_interp_bwd_kernel(int n, int R, const float* __restrict__ phi, const float* __restrict__ V, const float* __restrict__ dfv,
                       float* __restrict__ dphi, float* __restrict__ dV) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Corner8 c;
    corners(V + 3 * i, R, c);
    const float g = dfv[i];
    float gv[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        atomicAdd(dphi + c.idx[k], c.w[k] * g);
        const float ph = phi[c.idx[k]] * g;
        gv[0] += c.dw[k][0] * ph, gv[1] += c.dw[k][1] * ph, gv[2] += c.dw[k][2] * ph;
    }
#pragma unroll
    for (int d = 0; d < 3; d++) dV[3 * i + d] = gv[d];
}

// forward:  out[k] = sum_d (-i c_d) in[d][k]          (in: 3 spectra, out: 1)
// adjoint:  out[d][k] = (+i c_d) in[k]                (in: 1 spectrum, out: 3)
__global__ void __launch_bounds__(256)
dpsr_spectral_kernel(int R, float sig, const float2* __restrict__ in, float2* __restrict__ out, int adjoint) {
    const int Rh = R / 2 + 1;
    const size_t K = (size_t)R * R * Rh;
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const int l = (int)(k % Rh), j = (int)((k / Rh) % R), i = (int)(k / ((size_t)Rh * R));
    const float f[3] = {(float)(i < (R + 1) / 2 ? i : i - R), (float)(j < (R + 1) / 2 ? j : j - R), (float)l};
    // spec_gaussian_filter: binary64, then .float()
    const double dis = sqrt((double)f[0] * f[0] + (double)f[1] * f[1] + (double)f[2] * f[2]);
    const double t = (double)sig * 2.0 * dis / (double)R;
    const float G = (float)exp(-0.5 * t * t);
    const float two_pi = (float)(2.0 * 3.14159265358979323846);
    const float w[3] = {f[0] * two_pi, f[1] * two_pi, f[2] * two_pi};
    const float lap = -(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const float inv = 1.0f / (lap + 1e-6f);
    const bool dc = (k == 0);
    if (!adjoint) {
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const float2 x = in[(size_t)d * K + k];
            // N_ = Nhat G;  DivN = sum_d -(i N_) omega_d = sum_d omega_d G (im, -re)
            re += (x.y * G) * w[d];
            im += (-(x.x * G)) * w[d];
        }
        out[k] = dc ? make_float2(0.f, 0.f) : make_float2(re * inv, im * inv);
    } else {
        const float2 g = in[k];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const float cd = dc ? 0.f : w[d] * G * inv;  // (+i c_d)(g.x + i g.y) = c_d (-g.y, g.x)
            out[(size_t)d * K + k] = make_float2(-cd * g.y, cd * g.x);
        }
    }
}

// ---- umbrella-operator Laplacian regulariser of a triangle mesh (R/nvdiffrast_utils/regularizer.py:40-60) -------------------
//   term[v] = sum over the faces at v of (a - v) + (b - v)   (a, b the face's other corners; an interior edge counts twice)
//   norm[v] = 2 x (faces at v);   loss = mean over the 3 V components of (term / max(norm, 1))^2
// Three launches forward (per face: 9 + 3 fp32 atomics, like the reference's scatter_add_; per vertex: normalise, square, block
// sums; one workgroup: the sum), one backward (per face: the transposed stencil of d loss / d term, 9 atomics).  acc = [term (V, 3)
// | norm (V)], zeroed by the caller's entry point; the normalised term overwrites term for the backward.
__global__ void __launch_bounds__(256)
laplace_face_fwd_kernel(int F, const float* __restrict__ v, const int* __restrict__ f, float* __restrict__ term, float* __restrict__ norm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= F) return;
    const int i0 = f[3 * i], i1 = f[3 * i + 1], i2 = f[3 * i + 2];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float a = v[3 * i0 + c], b = v[3 * i1 + c], d = v[3 * i2 + c];
        atomicAdd(&term[3 * i0 + c], (b - a) + (d - a));
        atomicAdd(&term[3 * i1 + c], (a - b) + (d - b));
        atomicAdd(&term[3 * i2 + c], (a - d) + (b - d));
    }
    atomicAdd(&norm[i0], 2.0f), atomicAdd(&norm[i1], 2.0f), atomicAdd(&norm[i2], 2.0f);
}
__global__ void __launch_bounds__(256)
laplace_vertex_kernel(int V, float* __restrict__ term, const float* __restrict__ norm, float* __restrict__ partial) {
    __shared__ float red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    if (i < V) {
        const float inv = 1.0f / fmaxf(norm[i], 1.0f);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float t = term[3 * i + c] * inv;
            term[3 * i + c] = t;
            s += t * t;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(256)
laplace_sum_kernel(int nblk, int V, const float* __restrict__ partial, float* __restrict__ loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (3.0f * (float)V);
}
// d loss / d v: g[v] = dloss * 2 t[v] / (3 V max(norm[v], 1)) is the gradient w.r.t. term[v]; term is linear in the corners
__global__ void __launch_bounds__(256)
laplace_face_bwd_kernel(int F, int V, const int* __restrict__ f, const float* __restrict__ tn, const float* __restrict__ norm,
                        const float* __restrict__ dloss, float* __restrict__ dv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= F) return;
    const int i0 = f[3 * i], i1 = f[3 * i + 1], i2 = f[3 * i + 2];
    const float k = dloss[0] * 2.0f / (3.0f * (float)V);
    const float s0 = k / fmaxf(norm[i0], 1.0f), s1 = k / fmaxf(norm[i1], 1.0f), s2 = k / fmaxf(norm[i2], 1.0f);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float g0 = tn[3 * i0 + c] * s0, g1 = tn[3 * i1 + c] * s1, g2 = tn[3 * i2 + c] * s2;
        atomicAdd(&dv[3 * i0 + c], -2.0f * g0 + g1 + g2);
        atomicAdd(&dv[3 * i1 + c], g0 - 2.0f * g1 + g2);
        atomicAdd(&dv[3 * i2 + c], g0 + g1 - 2.0f * g2);
    }
}

}  // namespace dgm

using namespace dgm;

namespace {
int pfail(const char* m) {
    dgm::set_last_error(m);
    return 1;
}
int done() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : pfail(hipGetErrorString(e));
}
}  // namespace

extern "C" {

int dgm_dpsr_splat_forward(int n, int res, const float* V, const float* N, float* grid, void* stream) {
    if (res <= 0 || res > 1024) return pfail("dpsr_splat_forward: bad resolution");
    if (!grid) return pfail("dpsr_splat_forward: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(grid, 0, (size_t)3 * res * res * res * sizeof(float), st) != hipSuccess) return pfail("dpsr_splat_forward: memset failed");
    if (n <= 0) return 0;
    if (!V || !N) return pfail("dpsr_splat_forward: NULL pointer");
    hipLaunchKernelGGL(dpsr_splat_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, res, V, N, grid);
    return done();
}

int dgm_dpsr_splat_backward(int n, int res, const float* V, const float* N, const float* dgrid, float* dV, float* dN, void* stream) {
    if (n <= 0) return 0;
    if (!V || !N || !dgrid || !dV || !dN) return pfail("dpsr_splat_backward: NULL pointer");
    hipLaunchKernelGGL(dpsr_splat_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, res, V, N, dgrid, dV, dN);
    return done();
}

int dgm_dpsr_interp_forward(int n, int res, const float* phi, const float* V, float* fv, void* stream) {
    if (n <= 0) return 0;
    if (!phi || !V || !fv) return pfail("dpsr_interp_forward: NULL pointer");
    hipLaunchKernelGGL(dpsr_interp_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, res, phi, V, fv);
    return done();
}

int dgm_dpsr_interp_backward(int n, int res, const float* phi, const float* V, const float* dfv, float* dphi, float* dV, void* stream) {
    if (!dphi) return pfail("dpsr_interp_backward: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dphi, 0, (size_t)res * res * res * sizeof(float), st) != hipSuccess) return pfail("dpsr_interp_backward: memset failed");
    if (n <= 0) return 0;
    if (!phi || !V || !dfv || !dV) return pfail("dpsr_interp_backward: NULL pointer");
    hipLaunchKernelGGL(dpsr_interp_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, res, phi, V, dfv, dphi, dV);
    return done();
}

int dgm_dpsr_spectral(int res, float sig, const float* in, float* out, int adjoint, void* stream) {
    if (res <= 0 || !in || !out) return pfail("dpsr_spectral: bad argument");
    const size_t K = (size_t)res * res * (res / 2 + 1);
    hipLaunchKernelGGL(dpsr_spectral_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, res, sig,
                       (const float2*)in, (float2*)out, adjoint);
    return done();
}

size_t dgm_laplace_scratch_floats(int V) { return V > 0 ? (size_t)4 * V + (size_t)((V + 255) / 256) : 0; }

int dgm_laplace_forward(int V, int F, const float* v_pos, const int* faces, float* scratch, float* loss, void* stream) {
    if (V <= 0 || F < 0 || !v_pos || !scratch || !loss || (F > 0 && !faces)) return pfail("laplace_forward: bad argument");
    hipStream_t st = (hipStream_t)stream;
    float *term = scratch, *norm = scratch + (size_t)3 * V, *partial = scratch + (size_t)4 * V;
    if (hipMemsetAsync(scratch, 0, (size_t)4 * V * sizeof(float), st) != hipSuccess) return pfail("laplace_forward: memset failed");
    const int nblk = (V + 255) / 256;
    if (F > 0) hipLaunchKernelGGL(laplace_face_fwd_kernel, dim3((F + 255) / 256), dim3(256), 0, st, F, v_pos, faces, term, norm);
    hipLaunchKernelGGL(laplace_vertex_kernel, dim3(nblk), dim3(256), 0, st, V, term, norm, partial);
    hipLaunchKernelGGL(laplace_sum_kernel, dim3(1), dim3(256), 0, st, nblk, V, partial, loss);
    return done();
}

int dgm_laplace_backward(int V, int F, const int* faces, const float* scratch, const float* dloss, float* dv, void* stream) {
    if (V <= 0 || F < 0 || !scratch || !dloss || !dv || (F > 0 && !faces)) return pfail("laplace_backward: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dv, 0, (size_t)3 * V * sizeof(float), st) != hipSuccess) return pfail("laplace_backward: memset failed");
    if (F > 0)
        hipLaunchKernelGGL(laplace_face_bwd_kernel, dim3((F + 255) / 256), dim3(256), 0, st, F, V, faces, scratch, scratch + (size_t)3 * V,
                           dloss, dv);
    return done();
}

}  // extern "C"
