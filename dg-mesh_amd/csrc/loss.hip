// Fused image loss for gfx950:  loss = (1 - lambda) * mean|I - G| + lambda * (1 - mean(SSIM(I, G)))
//
// Replaces, for the Gaussian-branch image loss of the train step (R/train.py:307-311), the PyTorch graph of
// l1_loss + ssim (R/utils/loss_utils.py:18-19, 32-76): five grouped 11x11 Gaussian convolutions (window sigma 1.5,
// zero padding 5) forward and their autograd backward -- 8 MIOpen convolutions of ~0.4 ms each on a 3x800x800
// image -- by two HBM-bound kernels.
//
//   forward : one 16x16 pixel tile per workgroup per channel; the 26x26 haloed tiles of I and G go to LDS once, the
//             five windowed moments (E[I], E[G], E[I^2], E[G^2], E[IG]) come from a separable pass through LDS, the
//             SSIM map value and the three per-pixel partials (ds/dmu1_total, ds/dE11, ds/dE12) are formed in
//             registers; partial sums per workgroup, summed in fixed order by a second tiny kernel (deterministic);
//   backward: d sum(SSIM) / dI = conv(a1) + 2 I conv(a11) + G conv(a12)   (the Gaussian window is symmetric, so the
//             adjoint of the zero-padded convolution is the same convolution) + the L1 sign term.
// G (the ground-truth image) receives no gradient, as in the reference.
#include "dgm_common.hpp"

namespace dgm {

static constexpr int LT = 32, LTY = 16, LH = 5;            // tile (32 wide, 16 tall), halo
static constexpr int LR = LT + 2 * LH, LRY = LTY + 2 * LH;  // haloed tile: 42 x 26
static constexpr int LP = 44;                               // its LDS row pitch: a multiple of 4 (16-byte reads)

// the 11-tap window (R/utils/loss_utils.py:32-34), evaluated once on the host and passed BY VALUE: the taps sit in scalar
// registers for the whole kernel instead of being recomputed (11 expf and the normalisation) by every thread
struct Gauss11 {
    float w[11];
};
static Gauss11 gauss11_host() {
    Gauss11 g;
    float s = 0.f;
    for (int k = 0; k < 11; k++) {
        g.w[k] = expf(-(float)((k - 5) * (k - 5)) / (2.0f * 1.5f * 1.5f));
        s += g.w[k];
    }
    for (int k = 0; k < 11; k++) g.w[k] /= s;
    return g;
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Both kernels: one 32x16 pixel tile of one channel per workgroup; the 42x26 haloed tile goes to LDS once; separable window --
// a horizontal pass (a thread forms FOUR neighbouring outputs of a row from four 16-byte LDS reads per input field; 208 such tasks:
// one round of the 256 threads) into LDS, then a vertical pass (a thread forms TWO outputs of a column from 12 reads per field).
// Rounds 1-4: 16x16 tiles, one output per thread and pass -- 103 LDS instructions per pixel and a halo of 2.6x the tile; now 33 and
// 2.1x.  Every output accumulates its eleven taps in the same ascending order as before.
template <int NO>
__device__ __forceinline__ void window(const float (&v)[NO + 10], const float* w, float (&out)[NO]) {
#pragma unroll
    for (int j = 0; j < NO; j++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) acc += w[k] * v[j + k];
        out[j] = acc;
    }
}

__global__ void __launch_bounds__(256)
loss_fwd_kernel(const Gauss11 gw, const float* __restrict__ I, const float* __restrict__ G, int H, int W, float* __restrict__ a1,
                float* __restrict__ a11, float* __restrict__ a12, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float sI[LRY * LP], sG[LRY * LP];
    __shared__ __attribute__((aligned(16))) float hq[5][LRY * LT];
    __shared__ float red[4];
    const float* w = gw.w;
    const int ch = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const float* Ic = I + ch * plane;
    const float* Gc = G + ch * plane;
    const int x0 = blockIdx.x * LT - LH, y0 = blockIdx.y * LTY - LH;
    for (int i = threadIdx.x; i < LRY * LP; i += 256) {  // (the two pad columns too: the 16-byte reads below touch them)
        const int y = i / LP, x = i - y * LP;
        const int gy = y0 + y, gx = x0 + x;
        const bool in = x < LR && gy >= 0 && gy < H && gx >= 0 && gx < W;
        sI[i] = in ? Ic[(size_t)gy * W + gx] : 0.f;
        sG[i] = in ? Gc[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < LRY * (LT / 4)) {  // horizontal pass: (row, four columns)
        const int y = threadIdx.x >> 3, xg = (threadIdx.x & 7) * 4;
        float a[16], b[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 fa = *reinterpret_cast<const float4*>(&sI[y * LP + xg + 4 * q]);
            const float4 fb = *reinterpret_cast<const float4*>(&sG[y * LP + xg + 4 * q]);
            a[4 * q] = fa.x, a[4 * q + 1] = fa.y, a[4 * q + 2] = fa.z, a[4 * q + 3] = fa.w;
            b[4 * q] = fb.x, b[4 * q + 1] = fb.y, b[4 * q + 2] = fb.z, b[4 * q + 3] = fb.w;
        }
#pragma unroll
        for (int f = 0; f < 5; f++) {
            float v[14], o[4];
#pragma unroll
            for (int e = 0; e < 14; e++) v[e] = f == 0 ? a[e] : f == 1 ? b[e] : f == 2 ? a[e] * a[e] : f == 3 ? b[e] * b[e] : a[e] * b[e];
            window<4>(v, w, o);
            *reinterpret_cast<float4*>(&hq[f][y * LT + xg]) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = (threadIdx.x >> 5) * 2;  // vertical pass: column tx, rows ty, ty + 1
    float m[5][2];
#pragma unroll
    for (int f = 0; f < 5; f++) {
        float v[12];
#pragma unroll
        for (int e = 0; e < 12; e++) v[e] = hq[f][(ty + e) * LT + tx];
        window<2>(v, w, m[f]);
    }
    const int px = blockIdx.x * LT + tx;
    float ssim_sum = 0.f, l1_sum = 0.f;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int py = blockIdx.y * LTY + ty + j;
        if (px < W && py < H) {
            const float mu1 = m[0][j], mu2 = m[1][j], E11 = m[2][j], E22 = m[3][j], E12 = m[4][j];
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float s1 = E11 - mu1 * mu1, s2 = E22 - mu2 * mu2, s12 = E12 - mu1 * mu2;
            const float A = 2.f * mu1 * mu2 + C1, B = 2.f * s12 + C2;
            const float C = mu1 * mu1 + mu2 * mu2 + C1, D = s1 + s2 + C2;
            const float iCD = 1.f / (C * D);
            const float ssim = A * B * iCD;
            // partials of the map value w.r.t. (E11, E12, mu1) with sigma1^2 = E11 - mu1^2, sigma12 = E12 - mu1 mu2
            const float d11 = -ssim / D;
            const float d12 = 2.f * A * iCD;
            const float d1 = 2.f * mu2 * B * iCD - ssim * 2.f * mu1 / C + d11 * (-2.f * mu1) + d12 * (-mu2);
            const size_t o = ch * plane + (size_t)py * W + px;
            a1[o] = d1;
            a11[o] = d11;
            a12[o] = d12;
            ssim_sum += ssim;
            l1_sum += fabsf(sI[(ty + j + LH) * LP + tx + LH] - sG[(ty + j + LH) * LP + tx + LH]);
        }
    }
    const float ssum = block_sum_256(ssim_sum, red);
    const float lsum = block_sum_256(l1_sum, red);
    if (threadIdx.x == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * b] = ssum;
        partial[2 * b + 1] = lsum;
    }
}

__global__ void __launch_bounds__(256)
loss_reduce_kernel(int nblocks, const float* __restrict__ partial, float inv_n, float lambda, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f, l = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        s += partial[2 * i];
        l += partial[2 * i + 1];
    }
    const float S = block_sum_256(s, red);
    const float Lsum = block_sum_256(l, red);
    if (threadIdx.x == 0) {
        out[0] = (1.f - lambda) * (Lsum * inv_n) + lambda * (1.f - S * inv_n);
        out[1] = Lsum * inv_n;  // L1 term
        out[2] = S * inv_n;     // mean SSIM
    }
}

__global__ void __launch_bounds__(256)
loss_bwd_kernel(const Gauss11 gw, const float* __restrict__ I, const float* __restrict__ G, const float* __restrict__ a1,
                const float* __restrict__ a11, const float* __restrict__ a12, int H, int W, float inv_n, float lambda,
                const float* __restrict__ gout, float* __restrict__ dI) {
    __shared__ __attribute__((aligned(16))) float sA[3][LRY * LP];
    __shared__ __attribute__((aligned(16))) float hq[3][LRY * LT];
    const float* w = gw.w;
    const int ch = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const int x0 = blockIdx.x * LT - LH, y0 = blockIdx.y * LTY - LH;
    for (int i = threadIdx.x; i < LRY * LP; i += 256) {
        const int y = i / LP, x = i - y * LP;
        const int gy = y0 + y, gx = x0 + x;
        const bool in = x < LR && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t o = ch * plane + (size_t)gy * W + gx;
        sA[0][i] = in ? a1[o] : 0.f;
        sA[1][i] = in ? a11[o] : 0.f;
        sA[2][i] = in ? a12[o] : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < LRY * (LT / 4)) {  // horizontal pass: (row, four columns)
        const int y = threadIdx.x >> 3, xg = (threadIdx.x & 7) * 4;
#pragma unroll
        for (int f = 0; f < 3; f++) {
            float v[16], o[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 fa = *reinterpret_cast<const float4*>(&sA[f][y * LP + xg + 4 * q]);
                v[4 * q] = fa.x, v[4 * q + 1] = fa.y, v[4 * q + 2] = fa.z, v[4 * q + 3] = fa.w;
            }
            const float (&v14)[14] = *reinterpret_cast<const float (*)[14]>(&v[0]);
            window<4>(v14, w, o);
            *reinterpret_cast<float4*>(&hq[f][y * LT + xg]) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = (threadIdx.x >> 5) * 2;
    float c[3][2];
#pragma unroll
    for (int f = 0; f < 3; f++) {
        float v[12];
#pragma unroll
        for (int e = 0; e < 12; e++) v[e] = hq[f][(ty + e) * LT + tx];
        window<2>(v, w, c[f]);
    }
    const int px = blockIdx.x * LT + tx;
    const float go = gout[0];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int py = blockIdx.y * LTY + ty + j;
        if (px < W && py < H) {
            const size_t o = ch * plane + (size_t)py * W + px;
            const float iv = I[o], gv = G[o];
            const float d = iv - gv;
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            const float dssim = c[0][j] + 2.f * iv * c[1][j] + gv * c[2][j];
            dI[o] = go * ((1.f - lambda) * inv_n * sgn - lambda * inv_n * dssim);
        }
    }
}

}  // namespace dgm

using namespace dgm;
namespace dgm {
void set_last_error(const char* msg);
}

extern "C" {

size_t dgm_image_loss_workspace_bytes(int channels, int H, int W) {
    const size_t n = (size_t)channels * H * W;
    const size_t nb = (size_t)channels * ((H + LTY - 1) / LTY) * ((W + LT - 1) / LT);
    return align_up(n * 4, 256) * 3 + align_up(nb * 2 * 4, 256) + 512;
}

// image, gt: (channels, H, W) fp32.  out: 3 floats (loss, L1 term, mean SSIM).  workspace is kept for backward.
int dgm_image_loss_forward(const float* image, const float* gt, int channels, int H, int W, float lambda_dssim,
                           char* workspace, float* out, void* stream) {
    if (!image || !gt || !workspace || !out || channels <= 0 || H <= 0 || W <= 0) {
        set_last_error("image_loss_forward: bad argument");
        return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)channels * H * W;
    char* p = align_ptr(workspace);
    float* a1 = (float*)p;
    float* a11 = (float*)(p + align_up(n * 4, 256));
    float* a12 = (float*)(p + 2 * align_up(n * 4, 256));
    float* partial = (float*)(p + 3 * align_up(n * 4, 256));
    dim3 grid((W + LT - 1) / LT, (H + LTY - 1) / LTY, channels);
    hipLaunchKernelGGL(loss_fwd_kernel, grid, dim3(256), 0, st, gauss11_host(), image, gt, H, W, a1, a11, a12, partial);
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, (int)(grid.x * grid.y * grid.z), partial,
                       1.0f / (float)n, lambda_dssim, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(hipGetErrorString(e));
        return 1;
    }
    return 0;
}

// grad_out: device scalar (dL/dloss).  d_image: (channels, H, W), fully written.
int dgm_image_loss_backward(const float* image, const float* gt, int channels, int H, int W, float lambda_dssim,
                            const char* workspace, const float* grad_out, float* d_image, void* stream) {
    if (!image || !gt || !workspace || !grad_out || !d_image) {
        set_last_error("image_loss_backward: NULL pointer");
        return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)channels * H * W;
    const char* p = align_ptr((char*)workspace);
    const float* a1 = (const float*)p;
    const float* a11 = (const float*)(p + align_up(n * 4, 256));
    const float* a12 = (const float*)(p + 2 * align_up(n * 4, 256));
    dim3 grid((W + LT - 1) / LT, (H + LTY - 1) / LTY, channels);
    hipLaunchKernelGGL(loss_bwd_kernel, grid, dim3(256), 0, st, gauss11_host(), image, gt, a1, a11, a12, H, W, 1.0f / (float)n, lambda_dssim,
                       grad_out, d_image);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(hipGetErrorString(e));
        return 1;
    }
    return 0;
}

}  // extern "C"
