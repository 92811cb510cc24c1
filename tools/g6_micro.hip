// Micro benchmark of the trunk-layer GEMM kernels on synthetic data (no torch): timing per variant and the per-phase
// cycle breakdown of the pipelined kernel.   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dg-mesh_amd/csrc
//                                                   tools/g6_micro.hip -o tools/bin/g6_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mlp_bf16x6.hpp"
using namespace dgm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 100000, K = 256, iters = 20;
    float *A, *C, *bias, *W;
    unsigned* mask;
    uint4* Bp;
    unsigned long long* dbg;
    CK(hipMalloc(&A, (size_t)M * K * 4));
    CK(hipMalloc(&C, (size_t)M * 256 * 4));
    CK(hipMalloc(&bias, 1024));
    CK(hipMalloc(&W, 256 * 256 * 4));
    CK(hipMalloc(&mask, (size_t)M * 32));
    CK(hipMalloc(&Bp, (size_t)K * 256 * 6));
    CK(hipMalloc(&dbg, 256));
    std::vector<float> h((size_t)M * K);
    srand(1);
    for (auto& v : h) v = (rand() / (float)RAND_MAX) - 0.3f;
    CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 256 * 256; i++) h[i] = ((rand() / (float)RAND_MAX) - 0.5f) * 0.1f;
    CK(hipMemcpy(W, h.data(), 256 * 256 * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 1024));
    hipLaunchKernelGGL(mlp_prep6_kernel, dim3((K / 8 * 256 + 255) / 256), dim3(256), 0, 0, 0, K, 256, 256, 0, 0, 0, 256, W, Bp);
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    const int t128 = (M + 127) / 128, t32 = (M + 31) / 32;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; i++) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %8.1f us\n", name, ms * 1000.f / iters);
    };
    printf("M=%d K=%d CUs=%d\n", M, K, ncu);
    timeit("gemm6 2WG x 4 waves", [&] {
        hipLaunchKernelGGL((mlp_gemm6_kernel<0, 2, 2, 2, 4, false>), dim3(t128), dim3(256), 0, 0, M, A, K, K, (const float*)nullptr, 0, 0, 0, Bp, bias, mask, C, 256, 256);
    });
    timeit("gemm6 2WG x 4 waves as 4x1 (32x256)", [&] {
        hipLaunchKernelGGL((mlp_gemm6_kernel<0, 4, 1, 1, 8, false>), dim3(t128), dim3(256), 0, 0, M, A, K, K, (const float*)nullptr, 0, 0, 0, Bp, bias, mask, C, 256, 256);
    });
    timeit("  same, no epilogue", [&] {
        hipLaunchKernelGGL((mlp_gemm6_kernel<0, 4, 1, 1, 8, false, 4>), dim3(t128), dim3(256), 0, 0, M, A, K, K, (const float*)nullptr, 0, 0, 0, Bp, bias, mask, C, 256, 256);
    });
    timeit("  same, no MFMA", [&] {
        hipLaunchKernelGGL((mlp_gemm6_kernel<0, 4, 1, 1, 8, false, 2>), dim3(t128), dim3(256), 0, 0, M, A, K, K, (const float*)nullptr, 0, 0, 0, Bp, bias, mask, C, 256, 256);
    });
    timeit("gemm6r 4 waves x 64 cols", [&] {
        hipLaunchKernelGGL((mlp_gemm6r_kernel<0, 16, 2, 4>), dim3(t32 < ncu ? t32 : ncu), dim3(256), 0, 0, M, t32, A, K, K, (const float*)nullptr, 0, Bp, bias, mask, C);
    });
    timeit("gemm6r 8 waves x 32 cols", [&] {
        hipLaunchKernelGGL((mlp_gemm6r_kernel<0, 16, 1, 8>), dim3(t32 < ncu ? t32 : ncu), dim3(512), 0, 0, M, t32, A, K, K, (const float*)nullptr, 0, Bp, bias, mask, C);
    });
    timeit("  8w: no MFMA", [&] {
        hipLaunchKernelGGL((mlp_gemm6r_kernel<0, 16, 1, 8, 2>), dim3(t32 < ncu ? t32 : ncu), dim3(512), 0, 0, M, t32, A, K, K, (const float*)nullptr, 0, Bp, bias, mask, C);
    });
    timeit("  8w: no stores", [&] {
        hipLaunchKernelGGL((mlp_gemm6r_kernel<0, 16, 1, 8, 4>), dim3(t32 < ncu ? t32 : ncu), dim3(512), 0, 0, M, t32, A, K, K, (const float*)nullptr, 0, Bp, bias, mask, C);
    });
    timeit("  8w: no C stores", [&] {
        hipLaunchKernelGGL((mlp_gemm6r_kernel<0, 16, 1, 8, 6>), dim3(t32 < ncu ? t32 : ncu), dim3(512), 0, 0, M, t32, A, K, K, (const float*)nullptr, 0, Bp, bias, mask, C, (unsigned long long*)nullptr);
    });
    timeit("  8w: no mask stores", [&] {
        hipLaunchKernelGGL((mlp_gemm6r_kernel<0, 16, 1, 8, 7>), dim3(t32 < ncu ? t32 : ncu), dim3(512), 0, 0, M, t32, A, K, K, (const float*)nullptr, 0, Bp, bias, mask, C, (unsigned long long*)nullptr);
    });
    timeit("  8w: no loads", [&] {
        hipLaunchKernelGGL((mlp_gemm6r_kernel<0, 16, 1, 8, 5>), dim3(t32 < ncu ? t32 : ncu), dim3(512), 0, 0, M, t32, A, K, K, (const float*)nullptr, 0, Bp, bias, mask, C);
    });
    timeit("memcpy A -> C (102 MB)", [&] { CK(hipMemcpyAsync(C, A, (size_t)M * 1024, hipMemcpyDeviceToDevice, 0)); });
    unsigned long long hd[16];
    CK(hipMemset(A, 0, (size_t)M * K * 4));
    timeit("gemm6r 8w, A = zeros (DVFS check)", [&] {
        hipLaunchKernelGGL((mlp_gemm6r_kernel<0, 16, 1, 8>), dim3(t32 < ncu ? t32 : ncu), dim3(512), 0, 0, M, t32, A, K, K, (const float*)nullptr, 0, Bp, bias, mask, C, (unsigned long long*)nullptr);
    });
    CK(hipMemset(Bp, 0, (size_t)K * 256 * 6));
    timeit("gemm6r 8w, A = B = zeros", [&] {
        hipLaunchKernelGGL((mlp_gemm6r_kernel<0, 16, 1, 8>), dim3(t32 < ncu ? t32 : ncu), dim3(512), 0, 0, M, t32, A, K, K, (const float*)nullptr, 0, Bp, bias, mask, C, (unsigned long long*)nullptr);
    });
    CK(hipMemset(dbg, 0, 256));
    hipLaunchKernelGGL((mlp_gemm6r_kernel<0, 16, 1, 8, 0, true>), dim3(t32 < ncu ? t32 : ncu), dim3(512), 0, 0, M, t32, A, K, K, (const float*)nullptr, 0, Bp, bias, mask, C, dbg);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hd, dbg, 128, hipMemcpyDeviceToHost));
    const char* names2[5] = {"split", "load", "mfma", "epilogue", "barrier"};
    for (int w = 0; w < 2; w++) {
        printf("gemm6r-8w wave %d (steps %llu): ", w * 4, hd[w * 8 + 6]);
        for (int i = 0; i < 5; i++) printf("%s %.0f  ", names2[i], (double)hd[w * 8 + i] / (double)hd[w * 8 + 6]);
        printf(" [ticks per step]\n");
    }
    return 0;
}
