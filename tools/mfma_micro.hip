// How fast does v_mfma_f32_32x32x16_bf16 issue?  (dependent chains of NACC accumulators, 1 or 2 waves per SIMD)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_micro.hip -o tools/bin/mfma_micro
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NACC, int VALU>
__global__ void __launch_bounds__(512) k(const uint4* in, float* out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; a++)
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    const bf16x8 x = __builtin_bit_cast(bf16x8, in[threadIdx.x & 63]), y = __builtin_bit_cast(bf16x8, in[64 + (threadIdx.x & 63)]);
    float v0 = threadIdx.x, v1 = 1.0001f, v2 = 0.5f, v3 = 0.25f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int rep = 0; rep < 24 / NACC; rep++) {
#pragma unroll
            for (int a = 0; a < NACC; a++) {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < VALU; q++) {  // independent fp32 VALU fillers
                    v0 = v0 * v1 + v2;
                    v2 = v2 * v1 + v3;
                }
            }
        }
    }
    float s = v0 + v2;
    for (int a = 0; a < NACC; a++)
        for (int r = 0; r < 16; r++) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    uint4* in;
    float* out;
    CK(hipMalloc(&in, 4096));
    CK(hipMemset(in, 0, 4096));
    CK(hipMalloc(&out, 256 * 512 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000;
    auto run = [&](const char* name, auto kern, int threads) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, in, out, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double mfma_per_simd = (double)iters * 24 * (threads / 256);
        printf("%-44s %8.1f us  -> %.1f ns per MFMA per SIMD (32 cyc = %.1f ns @2.4GHz)\n", name, ms * 1e3, ms * 1e6 / mfma_per_simd, 32 / 2.4);
        return 0;
    };
    run("1 wave/SIMD, 1 accumulator", k<1, 0>, 256);
    run("1 wave/SIMD, 2 accumulators", k<2, 0>, 256);
    run("1 wave/SIMD, 4 accumulators", k<4, 0>, 256);
    run("1 wave/SIMD, 8 accumulators", k<8, 0>, 256);
    run("2 waves/SIMD, 1 accumulator", k<1, 0>, 512);
    run("2 waves/SIMD, 2 accumulators", k<2, 0>, 512);
    run("2 waves/SIMD, 4 accumulators", k<4, 0>, 512);
    run("1 wave/SIMD, 4 acc + 2 FMA per MFMA", k<4, 1>, 256);
    run("1 wave/SIMD, 4 acc + 4 FMA per MFMA", k<4, 2>, 256);
    run("1 wave/SIMD, 4 acc + 8 FMA per MFMA", k<4, 4>, 256);
    run("2 waves/SIMD, 4 acc + 4 FMA per MFMA", k<4, 2>, 512);
    run("2 waves/SIMD, 4 acc + 8 FMA per MFMA", k<4, 4>, 512);
    return 0;
}
