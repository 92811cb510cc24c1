// What issues in the shadow of v_mfma_f32_32x32x16_bf16?  One wave per SIMD, 4 accumulators; per MFMA add N
// independent fillers of one kind.   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_micro2.hip -o tools/bin/mfma_micro2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KIND 0: none, 1: independent v_fma (8 chains), 2: ds_read_b128, 3: s_add (SALU), 4: v_and+v_sub (split-like, independent)
template <int KIND, int N, int WAVES>
__global__ void __launch_bounds__(WAVES * 256) k(const uint4* in, float* out, int iters) {
    __shared__ uint4 lds[1024];
    f32x16 acc[4];
    for (int a = 0; a < 4; a++)
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    lds[threadIdx.x & 1023] = in[threadIdx.x & 63];
    __syncthreads();
    const bf16x8 x = __builtin_bit_cast(bf16x8, in[threadIdx.x & 63]), y = __builtin_bit_cast(bf16x8, in[64 + (threadIdx.x & 63)]);
    float f[8];
    for (int q = 0; q < 8; q++) f[q] = threadIdx.x + q;
    uint4 d = make_uint4(0, 0, 0, 0);
    int sacc = iters;
    const unsigned lp = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)(&lds[0])) + (threadIdx.x & 63) * 16;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int a = 0; a < 4; a++) {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < N; q++) {
                    if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[(q + a * N) & 7]) : "v"(f[7 - ((q + a * N) & 7)]));
                    if (KIND == 2) {
                        uint4 t;
                        asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(lp));
                        d.x ^= t.x;
                    }
                    if (KIND == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
                    if (KIND == 4) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(f[(q + a * N) & 7]));
                }
            }
        }
    }
    float s = sacc + d.x;
    for (int q = 0; q < 8; q++) s += f[q];
    for (int a = 0; a < 4; a++)
        for (int r = 0; r < 16; r++) s += acc[a][r];
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
    auto run = [&](const char* name, auto kern, int threads) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, in, out, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %6.1f ns per MFMA per SIMD\n", name, ms * 1e6 / ((double)iters * 32 * (threads / 256)));
    };
    run("1w: MFMA only", k<0, 0, 1>, 256);
    run("1w: + 2 v_fma", k<1, 2, 1>, 256);
    run("1w: + 4 v_fma", k<1, 4, 1>, 256);
    run("1w: + 6 v_fma", k<1, 6, 1>, 256);
    run("1w: + 8 v_fma", k<1, 8, 1>, 256);
    run("1w: + 4 v_and", k<4, 4, 1>, 256);
    run("1w: + 8 v_and", k<4, 8, 1>, 256);
    run("1w: + 1 ds_read_b128", k<2, 1, 1>, 256);
    run("1w: + 2 ds_read_b128", k<2, 2, 1>, 256);
    run("1w: + 4 s_add", k<3, 4, 1>, 256);
    run("1w: + 8 s_add", k<3, 8, 1>, 256);
    run("2w: MFMA only", k<0, 0, 2>, 512);
    run("2w: + 4 v_fma", k<1, 4, 2>, 512);
    run("2w: + 8 v_fma", k<1, 8, 2>, 512);
    run("2w: + 8 s_add", k<3, 8, 2>, 512);
    run("2w: + 2 ds_read_b128", k<2, 2, 2>, 512);
    return 0;
}
