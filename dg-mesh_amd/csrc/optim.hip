// One-launch Adam over many tensors.
//
// The train step of DG-Mesh ends with three torch.optim.Adam(eps=1e-15) instances (Gaussians: 6-7 parameter groups,
// deform, deform_back; R/scene/gaussian_model_dpsr_dynamic_anchor.py:186-212, R/scene/deform_model.py:33-44): even
// with PyTorch's fused implementation that is one kernel per group plus the step-counter updates, ~0.5 ms per
// iteration of mostly launch latency for 35 MB of state.  Here every tensor of every group is updated by ONE kernel:
// the per-tensor pointers, learning rates and bias corrections travel in the kernel argument block (no device-side
// tables to maintain), a block binary-searches which tensor its 4096-element chunk belongs to.
//
// Update rule (torch.optim.Adam with amsgrad=False, weight_decay=0, maximize=False):
//   m = b1 m + (1 - b1) g ;  v = b2 v + (1 - b2) g^2 ;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "dgm_common.hpp"

namespace dgm {
void set_last_error(const char* msg);  // c_api.hip

static constexpr int ADAM_MAX = 64;      // tensors per launch
static constexpr int ADAM_CHUNK = 4096;  // elements per block

struct AdamArgs {
    float* p[ADAM_MAX];
    const float* g[ADAM_MAX];
    float* m[ADAM_MAX];
    float* v[ADAM_MAX];
    int n[ADAM_MAX];
    float step_size[ADAM_MAX];   // lr / (1 - b1^t)
    float inv_bc2s[ADAM_MAX];    // 1 / sqrt(1 - b2^t)
    int blk0[ADAM_MAX + 1];      // first block of each tensor
    int count;
    float b1, b2, eps;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float b1, float b2, float eps, float ss, float ib) {
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    p -= ss * (m / (sqrtf(v) * ib + eps));
}

__global__ void __launch_bounds__(256) adam_kernel(const AdamArgs a) {
    const int b = blockIdx.x;
    int lo = 0, hi = a.count;  // largest t with blk0[t] <= b
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.blk0[mid] <= b) lo = mid;
        else hi = mid;
    }
    const int t = lo;
    const int n = a.n[t];
    const int i0 = (b - a.blk0[t]) * ADAM_CHUNK, i1 = min(n, i0 + ADAM_CHUNK);
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    const float ss = a.step_size[t], ib = a.inv_bc2s[t], b1 = a.b1, b2 = a.b2, eps = a.eps;
    const bool vec = ((((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0);
    if (vec) {
        const int e1 = i0 + ((i1 - i0) & ~3);
        for (int i = i0 + threadIdx.x * 4; i < e1; i += 256 * 4) {
            float4 P = *reinterpret_cast<float4*>(p + i), M = *reinterpret_cast<float4*>(m + i), V = *reinterpret_cast<float4*>(v + i);
            const float4 Gv = *reinterpret_cast<const float4*>(g + i);
            adam1(P.x, Gv.x, M.x, V.x, b1, b2, eps, ss, ib);
            adam1(P.y, Gv.y, M.y, V.y, b1, b2, eps, ss, ib);
            adam1(P.z, Gv.z, M.z, V.z, b1, b2, eps, ss, ib);
            adam1(P.w, Gv.w, M.w, V.w, b1, b2, eps, ss, ib);
            *reinterpret_cast<float4*>(p + i) = P;
            *reinterpret_cast<float4*>(m + i) = M;
            *reinterpret_cast<float4*>(v + i) = V;
        }
        for (int i = e1 + threadIdx.x; i < i1; i += 256) adam1(p[i], g[i], m[i], v[i], b1, b2, eps, ss, ib);
    } else {
        for (int i = i0 + threadIdx.x; i < i1; i += 256) adam1(p[i], g[i], m[i], v[i], b1, b2, eps, ss, ib);
    }
}

}  // namespace dgm

using namespace dgm;

extern "C" int dgm_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, const long long* numel, const float* lr, const int* step,
                             float beta1, float beta2, float eps, void* stream) {
    if (n_tensors < 0 || (n_tensors > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr || !step))) {
        dgm::set_last_error("adam_step: NULL argument");
        return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    int i = 0;  // next tensor to place: a batch takes the next ADAM_MAX NON-EMPTY tensors, wherever they end
    while (i < n_tensors) {
        AdamArgs a;
        a.count = 0;
        a.b1 = beta1, a.b2 = beta2, a.eps = eps;
        int blocks = 0;
        for (; i < n_tensors && a.count < ADAM_MAX; i++) {
            if (numel[i] <= 0) continue;
            if (numel[i] > 0x7fffffffLL || step[i] < 1 || !params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]) {
                dgm::set_last_error("adam_step: bad tensor (NULL pointer, step < 1 or more than 2^31-1 elements)");
                return 1;
            }
            const int c = a.count++;
            a.p[c] = params[i], a.g[c] = grads[i], a.m[c] = exp_avg[i], a.v[c] = exp_avg_sq[i];
            a.n[c] = (int)numel[i];
            const double bc1 = 1.0 - pow((double)beta1, (double)step[i]), bc2 = 1.0 - pow((double)beta2, (double)step[i]);
            a.step_size[c] = (float)((double)lr[i] / bc1);
            a.inv_bc2s[c] = (float)(1.0 / sqrt(bc2));
            a.blk0[c] = blocks;
            blocks += (a.n[c] + ADAM_CHUNK - 1) / ADAM_CHUNK;
        }
        a.blk0[a.count] = blocks;
        if (blocks > 0) hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dgm::set_last_error(hipGetErrorString(e));
        return 1;
    }
    return 0;
}
