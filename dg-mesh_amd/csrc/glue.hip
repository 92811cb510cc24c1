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

extern "C" {

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
