// Ablation harness for the f16x3 trunk-layer GEMM (tools only; the kernel body is generated from the product header
// into tools/exp/g3_kernel.inc with ablation hooks: VAR bit0 no MFMA, bit1 no stores, bit3 no loads).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dg-mesh_amd/csrc tools/g3_micro.hip -o tools/bin/g3_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mlp_f16x3.hpp"
namespace dgm {
#include "exp/g3_kernel.inc"
#include "exp/g3p_kernel.inc"
}
using namespace dgm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int EPI, int VAR>
float run(int M, int ncu, const float* A, const uint4* Bp, const float* binv, const float* bias, unsigned* mask, float* C, unsigned* cmax) {
    const int nt = (M + 31) / 32, gx = nt < ncu ? nt : ncu;
    const int lds = 2 * 32 * (4 * 256 + 16) + 256 + 1024;
    hipFuncSetAttribute((const void*)g3_kernel<EPI, 16, 1, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 3; i++)
        hipLaunchKernelGGL((g3_kernel<EPI, 16, 1, VAR>), dim3(gx), dim3(512), lds, 0, M, nt, A, 256, 256, (const float*)nullptr, 0, Bp, binv, bias, mask, C, cmax);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++)
        hipLaunchKernelGGL((g3_kernel<EPI, 16, 1, VAR>), dim3(gx), dim3(512), lds, 0, M, nt, A, 256, 256, (const float*)nullptr, 0, Bp, binv, bias, mask, C, cmax);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 20 * 1000.f;
}

template <int EPI>
float run_p(int M, int ncu, const float* A, const uint4* Bp, const float* binv, const float* bias, unsigned* mask, float* C, unsigned* cmax) {
    const int nt = (M + 31) / 32, gx = nt < ncu ? nt : ncu;
    const int lds = 2 * 32 * (4 * 256 + 16) + 256 + 2048;
    hipFuncSetAttribute((const void*)mlp_gemm3p_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 3; i++)
        hipLaunchKernelGGL((mlp_gemm3p_kernel<EPI, false>), dim3(gx), dim3(512), lds, 0, M, nt, A, 256, Bp, binv, bias, mask, C, cmax, (unsigned*)nullptr);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++)
        hipLaunchKernelGGL((mlp_gemm3p_kernel<EPI, false>), dim3(gx), dim3(512), lds, 0, M, nt, A, 256, Bp, binv, bias, mask, C, cmax, (unsigned*)nullptr);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 20 * 1000.f;
}

template <int EPI, int VAR>
float run_pv(int M, int ncu, const float* A, const uint4* Bp, const float* binv, const float* bias, unsigned* mask, float* C, unsigned* cmax) {
    const int nt = (M + 31) / 32, gx = nt < ncu ? nt : ncu;
    const int lds = 2 * 32 * (4 * 256 + 16) + 256 + 2048;
    hipFuncSetAttribute((const void*)g3p_kernel<EPI, false, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 3; i++)
        hipLaunchKernelGGL((g3p_kernel<EPI, false, VAR>), dim3(gx), dim3(512), lds, 0, M, nt, A, 256, Bp, binv, bias, mask, C, cmax, (unsigned*)nullptr);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++)
        hipLaunchKernelGGL((g3p_kernel<EPI, false, VAR>), dim3(gx), dim3(512), lds, 0, M, nt, A, 256, Bp, binv, bias, mask, C, cmax, (unsigned*)nullptr);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 20 * 1000.f;
}

int main() {
    const int M = 100000;
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    float *A, *C, *binv, *bias;
    uint4* Bp;
    unsigned *mask, *cmax;
    CK(hipMalloc(&A, (size_t)M * 256 * 4));
    CK(hipMalloc(&C, (size_t)M * 256 * 4));
    CK(hipMalloc(&Bp, 256 * 256 * 4));
    CK(hipMalloc(&binv, 1024));
    CK(hipMalloc(&bias, 1024));
    CK(hipMalloc(&mask, (size_t)M * 32));
    CK(hipMalloc(&cmax, 4096));
    std::vector<float> h((size_t)M * 256);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.3f;
    CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(Bp, 0x11, 256 * 256 * 4));
    CK(hipMemset(binv, 0, 1024));
    CK(hipMemset(bias, 0, 1024));
    CK(hipMemset(mask, 0xff, (size_t)M * 32));
    CK(hipMemset(cmax, 0, 1024));
    printf("CUs %d, M %d; times in us per launch\n", ncu, M);
#define RUN(E, V, what) printf("EPI %d VAR %2d %-40s %8.1f\n", E, V, what, run<E, V>(M, ncu, A, Bp, binv, bias, mask, C, cmax));
    printf("product mlp_gemm3p_kernel<0> (software-pipelined)   %8.1f\n", run_p<0>(M, ncu, A, Bp, binv, bias, mask, C, cmax));
    printf("product mlp_gemm3p_kernel<1>                         %8.1f\n", run_p<1>(M, ncu, A, Bp, binv, bias, mask, C, cmax));
#define RUNP(E, V, what) printf("gemm3p EPI %d VAR %2d %-40s %8.1f\n", E, V, what, run_pv<E, V>(M, ncu, A, Bp, binv, bias, mask, C, cmax));
    RUNP(0, 0, "full (instrumented copy)")
    RUNP(0, 1, "no MFMA")
    RUNP(0, 2, "no stores")
    RUNP(0, 8, "no loads")
    RUNP(0, 3, "no MFMA, no stores")
    RUNP(0, 9, "no MFMA, no loads")
    RUNP(0, 10, "no stores, no loads")
    RUNP(0, 11, "no MFMA / stores / loads (VALU + LDS skeleton)")
    RUNP(0, 32, "no epilogue / maxima / split (MFMA + frag reads)")
    RUNP(0, 33, "no epilogue/split, no MFMA (frag reads + barrier)")
    RUNP(0, 96, "no epilogue/split, no frag reads (MFMA only)")
    RUNP(0, 64, "no frag reads")
    RUNP(0, 65, "no frag reads, no MFMA (VALU skeleton w/o LDS reads)")
    RUNP(1, 0, "bwd full")
    RUNP(1, 1, "bwd no MFMA")
    {
        run_pv<0, 16>(M, ncu, A, Bp, binv, bias, mask, C, cmax);
        unsigned long long hd[16];
        CK(hipMemcpy(hd, cmax + 256, 128, hipMemcpyDeviceToHost));
        for (int w = 0; w < 2; w++)
            printf("gemm3p timing wave %d (%llu tiles): steps 0-7 (epilogue) %llu  step 8 (wait+loads+rowmax) %llu  steps 9-15 (scales+split) %llu  unscale+barrier %llu  cycles per tile\n",
                   w * 4, hd[w * 8 + 5], hd[w * 8 + 0] / hd[w * 8 + 5], hd[w * 8 + 1] / hd[w * 8 + 5], hd[w * 8 + 2] / hd[w * 8 + 5], hd[w * 8 + 3] / hd[w * 8 + 5]);
    }
    if (getenv("G3_ONLY_P")) return 0;
    RUN(0, 0, "full")
    RUN(0, 1, "no MFMA")
    RUN(0, 2, "no stores")
    RUN(0, 8, "no loads")
    RUN(0, 3, "no MFMA, no stores")
    RUN(0, 9, "no MFMA, no loads")
    RUN(0, 10, "no stores, no loads")
    RUN(0, 11, "no MFMA/stores/loads (split+epilogue VALU)")
    RUN(0, 32, "no row-max DPP chain")
    RUN(0, 64, "no plane ds_writes")
    RUN(0, 128, "rinv written by lane 0 only")
    RUN(0, 96, "no max chain, no plane writes")
    RUN(0, 256, "all waves same order")
    RUN(1, 0, "bwd full")
    RUN(1, 1, "bwd no MFMA")
    RUN(1, 2, "bwd no stores")
    RUN(1, 10, "bwd no stores, no loads")
    {
        run<0, 16>(M, ncu, A, Bp, binv, bias, mask, C, cmax);
        unsigned long long hd[16];
        CK(hipMemcpy(hd, cmax + 256, 128, hipMemcpyDeviceToHost));
        for (int w = 0; w < 2; w++)
            printf("timing wave %d (%llu tiles): vmcnt-wait %llu  split %llu  mfma %llu  store(+loads) %llu  barrier %llu  cycles per tile\n", w * 4, hd[w * 8 + 5],
                   hd[w * 8 + 4] / hd[w * 8 + 5], hd[w * 8 + 0] / hd[w * 8 + 5], hd[w * 8 + 1] / hd[w * 8 + 5], hd[w * 8 + 2] / hd[w * 8 + 5], hd[w * 8 + 3] / hd[w * 8 + 5]);
    }
    // plain copy of the same bytes for reference
    {
        hipEvent_t a, b;
        hipEventCreate(&a), hipEventCreate(&b);
        hipMemcpyAsync(C, A, (size_t)M * 256 * 4, hipMemcpyDeviceToDevice, 0);
        hipEventRecord(a);
        for (int i = 0; i < 20; i++) hipMemcpyAsync(C, A, (size_t)M * 256 * 4, hipMemcpyDeviceToDevice, 0);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("hipMemcpy D2D of A -> C %8.1f\n", ms / 20 * 1000.f);
    }
    return 0;
}
