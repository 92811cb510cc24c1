// Can v_mfma_f32_32x32x16_bf16 take its B operand straight from AccVGPRs at full rate?  (inline asm, "a" constraint)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_micro3.hip -o tools/bin/mfma_micro3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: builtin, B in VGPR; 1: asm, B in VGPR; 2: asm, B in AGPR; 3: asm, B and acc in AGPR
__global__ void __launch_bounds__(256) k(const u32x4* in, float* out, int iters) {
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; r++) acc0[r] = 0.f, acc1[r] = 0.f;
    u32x4 x = in[threadIdx.x & 63];
    u32x4 b[8];
    for (int q = 0; q < 8; q++) b[q] = in[64 + ((threadIdx.x + q) & 63)];
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (MODE == 0) {
                typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, b[q]), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, b[q]), acc1, 0, 0, 0);
            } else if (MODE == 1) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(x), "v"(b[q]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(x), "v"(b[q]));
            } else if (MODE == 2) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(x), "a"(b[q]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(x), "a"(b[q]));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(x), "a"(b[q]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(x), "a"(b[q]));
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
    for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    u32x4* in;
    float* out;
    hipMalloc(&in, 4096);
    unsigned h[1024];
    for (int i = 0; i < 1024; i++) h[i] = 0x3f803f80u;  // bf16 1.0 pairs
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    auto run = [&](const char* name, auto kern) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, in, out, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms, v;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&v, out, 4, hipMemcpyDeviceToHost);
        printf("%-34s %6.1f ns per MFMA per SIMD   out[0] = %.1f (expect %.1f)\n", name, ms * 1e6 / ((double)iters * 16), v,
               2.0 * 16 * 16.0 * 8 * iters);
    };
    run("builtin, B in VGPR", k<0>);
    run("asm, B in VGPR", k<1>);
    run("asm, B in AGPR", k<2>);
    run("asm, B and acc in AGPR", k<3>);
    return 0;
}
