// Per-Gaussian glue of the train step as four kernels instead of ~85 one-op PyTorch kernels.
//
//  * gaussian_apply: the activations and deformation of R/gaussian_renderer/__init__.py:77-95 with the accessors of
//    R/scene/gaussian_model_dpsr_dynamic_anchor.py:92-128:
//        means3D   = xyz + d_xyz                         scales  = exp(scaling) + d_scaling
//        rotations = normalize(rotation) + d_rotation    opacity = sigmoid(opacity_logit)
//    (normalize = torch.nn.functional.normalize: q / max(|q|, 1e-12)); delta is the raw (P, ld) head output of the
//    deformation network, columns [d_xyz 0:3 | d_rotation 3:7 | d_scaling 7:10 | ...].
//  * cycle loss (R/train.py:221-238): (mean|b_xyz + d_xyz| + mean|b_rot + d_rot| + mean|b_scale + d_scale|) / 3 over
//    the raw outputs of deform (a) and deform_back (b); two-level fixed-order reduction (deterministic).
#include "dgm_common.hpp"

namespace dgm {
void set_last_error(const char* msg);  // c_api.hip

__global__ void __launch_bounds__(256)
gaussian_apply_fwd_kernel(int P, const float* __restrict__ xyz, const float* __restrict__ scaling,
                          const float* __restrict__ rotation, const float* __restrict__ opacity,
                          const float* __restrict__ delta, int ld, float* __restrict__ means, float* __restrict__ scales,
                          float* __restrict__ rots, float* __restrict__ opac) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float* d = delta + (size_t)i * ld;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        means[3 * i + c] = xyz[3 * i + c] + d[c];
        scales[3 * i + c] = expf(scaling[3 * i + c]) + d[7 + c];
    }
    const float q0 = rotation[4 * i], q1 = rotation[4 * i + 1], q2 = rotation[4 * i + 2], q3 = rotation[4 * i + 3];
    const float n = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);
    rots[4 * i] = q0 / n + d[3];
    rots[4 * i + 1] = q1 / n + d[4];
    rots[4 * i + 2] = q2 / n + d[5];
    rots[4 * i + 3] = q3 / n + d[6];
    opac[i] = 1.f / (1.f + expf(-opacity[i]));
}

__global__ void __launch_bounds__(256)
gaussian_apply_bwd_kernel(int P, const float* __restrict__ scaling, const float* __restrict__ rotation,
                          const float* __restrict__ opacity, const float* __restrict__ g_means,
                          const float* __restrict__ g_scales, const float* __restrict__ g_rots,
                          const float* __restrict__ g_opac, float* __restrict__ d_xyz, float* __restrict__ d_scaling,
                          float* __restrict__ d_rotation, float* __restrict__ d_opacity, float* __restrict__ d_delta,
                          int ld) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float* dd = d_delta + (size_t)i * ld;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float gm = g_means[3 * i + c], gs = g_scales[3 * i + c];
        d_xyz[3 * i + c] = gm;
        dd[c] = gm;
        d_scaling[3 * i + c] = gs * expf(scaling[3 * i + c]);
        dd[7 + c] = gs;
    }
    const float q[4] = {rotation[4 * i], rotation[4 * i + 1], rotation[4 * i + 2], rotation[4 * i + 3]};
    const float g[4] = {g_rots[4 * i], g_rots[4 * i + 1], g_rots[4 * i + 2], g_rots[4 * i + 3]};
    const float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (nrm > 1e-12f) {  // y = q / |q| :  dq = (g - y (y . g)) / |q|
        const float inv = 1.f / nrm;
        const float dot = (q[0] * g[0] + q[1] * g[1] + q[2] * g[2] + q[3] * g[3]) * inv * inv;
#pragma unroll
        for (int c = 0; c < 4; c++) d_rotation[4 * i + c] = (g[c] - q[c] * dot) * inv;
    } else {             // clamped denominator: y = q / 1e-12
#pragma unroll
        for (int c = 0; c < 4; c++) d_rotation[4 * i + c] = g[c] / 1e-12f;
    }
#pragma unroll
    for (int c = 0; c < 4; c++) dd[3 + c] = g[c];
    for (int c = 10; c < ld; c++) dd[c] = 0.f;
    const float s = 1.f / (1.f + expf(-opacity[i]));
    d_opacity[i] = g_opac[i] * s * (1.f - s);
}

static constexpr int CYC_ROWS = 1024;  // rows per block of the first reduction level

// partial[block][3] = sum over the block's rows of |a + b| for the xyz / rotation / scaling column groups
__global__ void __launch_bounds__(256)
cycle_fwd_kernel(int N, const float* __restrict__ a, const float* __restrict__ b, int ld, float* __restrict__ partial) {
    __shared__ float red[3][4];
    const int r0 = blockIdx.x * CYC_ROWS, r1 = min(N, r0 + CYC_ROWS);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int r = r0 + threadIdx.x; r < r1; r += 256) {
        const float* pa = a + (size_t)r * ld;
        const float* pb = b + (size_t)r * ld;
#pragma unroll
        for (int c = 0; c < 3; c++) s0 += fabsf(-pb[c] - pa[c]);
#pragma unroll
        for (int c = 3; c < 7; c++) s1 += fabsf(-pb[c] - pa[c]);
#pragma unroll
        for (int c = 7; c < 10; c++) s2 += fabsf(-pb[c] - pa[c]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s0 += __shfl_xor(s0, d, 64);
        s1 += __shfl_xor(s1, d, 64);
        s2 += __shfl_xor(s2, d, 64);
    }
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s0, red[1][threadIdx.x >> 6] = s1, red[2][threadIdx.x >> 6] = s2;
    __syncthreads();
    if (threadIdx.x < 3) partial[blockIdx.x * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// out = {loss, mean|xyz|, mean|rot|, mean|scale|}
__global__ void __launch_bounds__(256)
cycle_finish_kernel(int N, int blocks, const float* __restrict__ partial, float* __restrict__ out) {
    __shared__ float red[3][4];
    float s[3] = {0.f, 0.f, 0.f};
    for (int k = threadIdx.x; k < blocks; k += 256) {
        s[0] += partial[k * 3], s[1] += partial[k * 3 + 1], s[2] += partial[k * 3 + 2];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
        for (int c = 0; c < 3; c++) s[c] += __shfl_xor(s[c], d, 64);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int c = 0; c < 3; c++) red[c][threadIdx.x >> 6] = s[c];
    __syncthreads();
    if (threadIdx.x == 0) {
        const float lx = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (3.f * N);
        const float lr = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (4.f * N);
        const float ls = ((red[2][0] + red[2][1]) + (red[2][2] + red[2][3])) / (3.f * N);
        out[0] = (lx + lr + ls) / 3.f;
        out[1] = lx, out[2] = lr, out[3] = ls;
    }
}

// d|x|/dx = sgn(x) with sgn(0) = 0 (torch.abs backward);  x = -b - a  =>  d/da = d/db = -sgn(x) * weight
__global__ void __launch_bounds__(256)
cycle_bwd_kernel(int N, const float* __restrict__ a, const float* __restrict__ b, int ld, const float* __restrict__ grad,
                 float* __restrict__ d_a, float* __restrict__ d_b) {
    // one thread per ELEMENT of the (N, ld) row-major arrays: consecutive lanes touch consecutive floats (a thread per row
    // walked the rows at a 52-byte pitch, thirteen partial-line passes over the same memory)
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)N * ld) return;
    const int c = (int)(idx % (size_t)ld);
    const float g = grad[0] / 3.f;
    float v = 0.f;
    if (c < 10) {
        const float x = -b[idx] - a[idx];
        const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
        v = -sg * ((c >= 3 && c < 7) ? g / (4.f * N) : g / (3.f * N));
    }
    d_a[idx] = v;
    d_b[idx] = v;
}

}  // namespace dgm

using namespace dgm;

namespace {
int glue_fail(const char* m) {
    dgm::set_last_error(m);
    return 1;
}
int glue_done() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return glue_fail(hipGetErrorString(e));
    return 0;
}
}  // namespace

// ---- 6-DoF deformation heads (is_6dof=True): screw motion -> rigid transform -> moved centres ------------------------------------
// R/utils/time_utils.py:116-123 + R/utils/rigid_utils.py:10-83 (Modern Robotics 3.51 / 3.88) + the 6-DoF branch of render()
// (R/gaussian_renderer/__init__.py:68-75): from the raw head outputs (w_raw, v_raw) of a row
//     theta = |w_raw|,  w = w_raw / theta + 1e-5,  v = v_raw / theta + 1e-5          (the reference adds the 1e-5 AFTER the division)
//     R = I + sin(theta) [w] + (1 - cos(theta)) [w]^2
//     p = (theta I + (1 - cos(theta)) [w] + (theta - sin(theta)) [w]^2) v
//     T = [[R, p], [0 0 0 1]]                                                         (N, 4, 4), row-major
// and, for the renderer, means3D = (T [xyz, 1])[:3] / (T [xyz, 1])[3] = R xyz + p (the bottom row is constant).
// One thread per row; the backward kernels are the hand-derived adjoints (checked against autograd of the reference's functions).
struct Se3Row {
    float th, s, c, w[3], v[3], W[9], W2[9];
};
__device__ __forceinline__ void se3_row(const float* __restrict__ o, Se3Row& r) {
    const float w0 = o[0], w1 = o[1], w2 = o[2];
    r.th = sqrtf(w0 * w0 + w1 * w1 + w2 * w2);
    const float inv = 1.0f / r.th;
#pragma unroll
    for (int k = 0; k < 3; k++) r.w[k] = o[k] * inv + 1e-5f, r.v[k] = o[3 + k] * inv + 1e-5f;
    r.s = sinf(r.th), r.c = cosf(r.th);
    const float a = r.w[0], b = r.w[1], d = r.w[2];
    const float W[9] = {0.f, -d, b, d, 0.f, -a, -b, a, 0.f};
#pragma unroll
    for (int i = 0; i < 9; i++) r.W[i] = W[i];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
}

__global__ void __launch_bounds__(256)
se3_exp_fwd_kernel(int N, const float* __restrict__ wv, int ld, float* __restrict__ T) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    Se3Row r;
    se3_row(wv + (size_t)i * ld, r);
    float* t = T + (size_t)i * 16;
    const float omc = 1.0f - r.c, tms = r.th - r.s;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float p = 0.f;
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const float id = a == b ? 1.f : 0.f;
            t[4 * a + b] = id + r.s * r.W[3 * a + b] + omc * r.W2[3 * a + b];
            p += (r.th * id + omc * r.W[3 * a + b] + tms * r.W2[3 * a + b]) * r.v[b];
        }
        t[4 * a + 3] = p;
    }
    t[12] = 0.f, t[13] = 0.f, t[14] = 0.f, t[15] = 1.f;
}

// dT (N, 16) -> d(w_raw, v_raw) written to d_wv[i * ldd + 0..5]
__global__ void __launch_bounds__(256)
se3_exp_bwd_kernel(int N, const float* __restrict__ wv, int ld, const float* __restrict__ dT, float* __restrict__ d_wv, int ldd) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float* o = wv + (size_t)i * ld;
    Se3Row r;
    se3_row(o, r);
    const float* g = dT + (size_t)i * 16;
    const float omc = 1.0f - r.c, tms = r.th - r.s;
    float dR[9], dp[3], dG[9], dv[3], A[9];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        dp[a] = g[4 * a + 3];
#pragma unroll
        for (int b = 0; b < 3; b++) dR[3 * a + b] = g[4 * a + b];
    }
    // p = G v:  dv = G^T dp,  dG = dp v^T
#pragma unroll
    for (int b = 0; b < 3; b++) {
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float id = a == b ? 1.f : 0.f;
            sacc += (r.th * id + omc * r.W[3 * a + b] + tms * r.W2[3 * a + b]) * dp[a];
            dG[3 * a + b] = dp[a] * r.v[b];
        }
        dv[b] = sacc;
    }
    // gradient w.r.t. the skew matrix: sin dR + (1 - cos) dG + A W^T + W^T A with A = (1 - cos) dR + (theta - sin) dG
    float dth = 0.f;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        A[k] = omc * dR[k] + tms * dG[k];
        const float id = (k == 0 || k == 4 || k == 8) ? 1.f : 0.f;
        dth += dR[k] * (r.c * r.W[k] + r.s * r.W2[k]) + dG[k] * (id + r.s * r.W[k] + omc * r.W2[k]);
    }
    float dW[9];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            float x = r.s * dR[3 * a + b] + omc * dG[3 * a + b];
#pragma unroll
            for (int k = 0; k < 3; k++) x += A[3 * a + k] * r.W[3 * b + k] + r.W[3 * k + a] * A[3 * k + b];  // (A W^T)[a][b] + (W^T A)[a][b]
            dW[3 * a + b] = x;
        }
    const float dw[3] = {dW[7] - dW[5], dW[2] - dW[6], dW[3] - dW[1]};
    // w = w_raw / theta + 1e-5, v = v_raw / theta + 1e-5, theta = |w_raw|
    const float inv = 1.0f / r.th;
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) dot += dw[k] * o[k] + dv[k] * o[3 + k];
    const float dth_tot = dth - dot * inv * inv;
    float* d = d_wv + (size_t)i * ldd;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        d[k] = dw[k] * inv + dth_tot * o[k] * inv;
        d[3 + k] = dv[k] * inv;
    }
}

// out = (T [xyz, 1])[:3] / (T [xyz, 1])[3]   (general 4 x 4 rows: the division is kept, as render() has it)
__global__ void __launch_bounds__(256)
se3_transform_fwd_kernel(int N, const float* __restrict__ T, const float* __restrict__ xyz, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float* t = T + (size_t)i * 16;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const float h = t[12] * x + t[13] * y + t[14] * z + t[15];
#pragma unroll
    for (int a = 0; a < 3; a++) out[3 * i + a] = (t[4 * a] * x + t[4 * a + 1] * y + t[4 * a + 2] * z + t[4 * a + 3]) / h;
}

__global__ void __launch_bounds__(256)
se3_transform_bwd_kernel(int N, const float* __restrict__ T, const float* __restrict__ xyz, const float* __restrict__ g_out,
                         float* __restrict__ dT, float* __restrict__ d_xyz) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float* t = T + (size_t)i * 16;
    const float hom[4] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 1.0f};
    float num[4];
#pragma unroll
    for (int a = 0; a < 4; a++) num[a] = t[4 * a] * hom[0] + t[4 * a + 1] * hom[1] + t[4 * a + 2] * hom[2] + t[4 * a + 3];
    const float ih = 1.0f / num[3];
    // out_a = num_a / h:  d num_a = g_a / h (a < 3),  d h = -sum_a g_a num_a / h^2
    float dn[4];
    float dh = 0.f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float ga = g_out[3 * i + a];
        dn[a] = ga * ih;
        dh -= ga * num[a] * ih * ih;
    }
    dn[3] = dh;
    float dx[3] = {0.f, 0.f, 0.f};
    float* dt = dT + (size_t)i * 16;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            dt[4 * a + b] = dn[a] * hom[b];
            if (b < 3) dx[b] += dn[a] * t[4 * a + b];
        }
    d_xyz[3 * i] = dx[0], d_xyz[3 * i + 1] = dx[1], d_xyz[3 * i + 2] = dx[2];
}

extern "C" {

int dgm_se3_exp_forward(int N, const float* wv, int ld, float* T, void* stream) {
    if (N <= 0) return 0;
    if (!wv || !T) return glue_fail("se3_exp_forward: NULL pointer");
    if (ld < 6) return glue_fail("se3_exp_forward: rows need at least 6 columns (w, v)");
    hipLaunchKernelGGL(se3_exp_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, wv, ld, T);
    return glue_done();
}

int dgm_se3_exp_backward(int N, const float* wv, int ld, const float* dT, float* d_wv, int ldd, void* stream) {
    if (N <= 0) return 0;
    if (!wv || !dT || !d_wv) return glue_fail("se3_exp_backward: NULL pointer");
    if (ld < 6 || ldd < 6) return glue_fail("se3_exp_backward: rows need at least 6 columns (w, v)");
    hipLaunchKernelGGL(se3_exp_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, wv, ld, dT, d_wv, ldd);
    return glue_done();
}

int dgm_se3_transform_forward(int N, const float* T, const float* xyz, float* out, void* stream) {
    if (N <= 0) return 0;
    if (!T || !xyz || !out) return glue_fail("se3_transform_forward: NULL pointer");
    hipLaunchKernelGGL(se3_transform_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, T, xyz, out);
    return glue_done();
}

int dgm_se3_transform_backward(int N, const float* T, const float* xyz, const float* g_out, float* dT, float* d_xyz, void* stream) {
    if (N <= 0) return 0;
    if (!T || !xyz || !g_out || !dT || !d_xyz) return glue_fail("se3_transform_backward: NULL pointer");
    hipLaunchKernelGGL(se3_transform_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, T, xyz, g_out, dT, d_xyz);
    return glue_done();
}

int dgm_gaussian_apply_forward(int P, const float* xyz, const float* scaling, const float* rotation, const float* opacity,
                               const float* delta, int ld, float* means3D, float* scales, float* rotations, float* opacities,
                               void* stream) {
    if (P <= 0) return 0;
    if (!xyz || !scaling || !rotation || !opacity || !delta || !means3D || !scales || !rotations || !opacities)
        return glue_fail("gaussian_apply_forward: NULL pointer");
    if (ld < 10) return glue_fail("gaussian_apply_forward: delta needs at least 10 columns (d_xyz, d_rotation, d_scaling)");
    hipLaunchKernelGGL(gaussian_apply_fwd_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, xyz, scaling,
                       rotation, opacity, delta, ld, means3D, scales, rotations, opacities);
    return glue_done();
}

int dgm_gaussian_apply_backward(int P, const float* scaling, const float* rotation, const float* opacity,
                                const float* g_means3D, const float* g_scales, const float* g_rotations,
                                const float* g_opacities, float* d_xyz, float* d_scaling, float* d_rotation, float* d_opacity,
                                float* d_delta, int ld, void* stream) {
    if (P <= 0) return 0;
    if (!scaling || !rotation || !opacity || !g_means3D || !g_scales || !g_rotations || !g_opacities || !d_xyz || !d_scaling ||
        !d_rotation || !d_opacity || !d_delta)
        return glue_fail("gaussian_apply_backward: NULL pointer");
    if (ld < 10) return glue_fail("gaussian_apply_backward: delta needs at least 10 columns");
    hipLaunchKernelGGL(gaussian_apply_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, scaling, rotation,
                       opacity, g_means3D, g_scales, g_rotations, g_opacities, d_xyz, d_scaling, d_rotation, d_opacity, d_delta,
                       ld);
    return glue_done();
}

size_t dgm_cycle_loss_workspace_bytes(int N) { return (size_t)((N + CYC_ROWS - 1) / CYC_ROWS + 1) * 3 * sizeof(float); }

int dgm_cycle_loss_forward(int N, const float* a, const float* b, int ld, char* workspace, float* out, void* stream) {
    if (N <= 0) return glue_fail("cycle_loss_forward: N must be positive");
    if (!a || !b || !workspace || !out) return glue_fail("cycle_loss_forward: NULL pointer");
    if (ld < 10) return glue_fail("cycle_loss_forward: needs at least 10 columns");
    const int blocks = (N + CYC_ROWS - 1) / CYC_ROWS;
    hipLaunchKernelGGL(cycle_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, N, a, b, ld, (float*)workspace);
    hipLaunchKernelGGL(cycle_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, N, blocks, (const float*)workspace, out);
    return glue_done();
}

int dgm_cycle_loss_backward(int N, const float* a, const float* b, int ld, const float* grad_out, float* d_a, float* d_b,
                            void* stream) {
    if (N <= 0) return 0;
    if (!a || !b || !grad_out || !d_a || !d_b) return glue_fail("cycle_loss_backward: NULL pointer");
    hipLaunchKernelGGL(cycle_bwd_kernel, dim3((unsigned)(((size_t)N * ld + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, a, b, ld,
                       grad_out, d_a, d_b);
    return glue_done();
}

}  // extern "C"
