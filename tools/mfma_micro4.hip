// Do two waves on one SIMD overlap when one issues only MFMAs and the other only VALU / LDS work?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_micro4.hip -o tools/bin/mfma_micro4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE bit0: waves 0..3 run the MFMA stream; bit1: waves 4..7 run the VALU stream; bit2: waves 4..7 run an LDS-read stream
template <int MODE, int NACC>
__global__ void __launch_bounds__(512) k(const uint4* in, float* out, int iters) {
    __shared__ uint4 lds[1024];
    lds[threadIdx.x] = in[threadIdx.x & 63];
    lds[threadIdx.x + 512] = in[threadIdx.x & 63];
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    float s = 0.f;
    if (wv < 4) {
        if (MODE & 1) {
            f32x16 acc[NACC];
            for (int a = 0; a < NACC; a++)
                for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
            const bf16x8 x = __builtin_bit_cast(bf16x8, in[threadIdx.x & 63]), y = __builtin_bit_cast(bf16x8, in[64 + (threadIdx.x & 63)]);
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int rep = 0; rep < 32 / NACC; rep++)
#pragma unroll
                    for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
            }
            for (int a = 0; a < NACC; a++)
                for (int r = 0; r < 16; r++) s += acc[a][r];
        }
    } else {
        if (MODE & 2) {
            float f[8];
            for (int q = 0; q < 8; q++) f[q] = threadIdx.x + q;
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int rep = 0; rep < 32; rep++)
#pragma unroll
                    for (int q = 0; q < 8; q++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q]) : "v"(f[(q + 3) & 7]));
            }
            for (int q = 0; q < 8; q++) s += f[q];
        }
        if (MODE & 4) {
            const unsigned lp = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)(&lds[0])) + (threadIdx.x & 63) * 16;
            unsigned acc = 0;
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int rep = 0; rep < 32; rep++) {
                    uint4 t;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(lp));
                    asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                    acc ^= t.x;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            s += acc;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    uint4* in;
    float* out;
    hipMalloc(&in, 4096);
    hipMemset(in, 0, 4096);
    hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 1000;
    auto run = [&](const char* name, auto kern) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, in, out, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %8.1f us\n", name, ms * 1e3);
    };
    run("MFMA wave alone (32 MFMA/iter, 4 acc)", k<1, 4>);
    run("MFMA wave alone (32 MFMA/iter, 1 acc)", k<1, 1>);
    run("VALU wave alone (256 v_fma/iter)", k<2, 4>);
    run("LDS wave alone (32 ds_read_b128/iter)", k<4, 4>);
    run("MFMA wave + VALU wave on the same SIMD (4 acc)", k<3, 4>);
    run("MFMA wave + VALU wave on the same SIMD (1 acc)", k<3, 1>);
    run("MFMA wave + LDS wave on the same SIMD", k<5, 4>);
    run("MFMA wave + VALU + LDS wave", k<7, 4>);
    return 0;
}
