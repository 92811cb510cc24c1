// What does one VALU instruction cost on gfx950, by kind and by waves per SIMD?  The blend kernels (render.hip, render_bwd3.hip)
// are VALU-bound; DESIGN.md section 4b prices them in "instructions", this prices the instructions.
// Each wave runs ITERS x 64 independent instructions of one kind (8 independent register chains); the kernel is launched with
// 1, 2, 4, 8 waves per SIMD (256-thread workgroups = one wave per SIMD each, WPS workgroups per CU) and reports
// cycles per instruction per SIMD = s_memtime span of the slowest wave x 1 / (instructions per wave x waves per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_micro.hip -o tools/bin/valu_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

enum { FMA, PKFMA, PKMUL, PKADD, EXP, RCP, DPPQ, DPPROW, SWAP32, CNDMASK, CMP, MINF, FMA_SALU, FMA_SALU2, MFMA_ONLY, MFMA_VALU4, MFMA_VALU8,
       MFMA_VALU12, LDSW, NKIND };
static const char* names[NKIND] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_exp_f32", "v_rcp_f32",
                                   "v_add_f32_dpp quad_perm", "v_add_f32_dpp row_shr", "v_permlane32_swap", "v_cndmask_b32",
                                   "v_cmp_lt_f32", "v_min_f32", "v_fma + 1 salu each", "v_fma + 2 salu each", "mfma_16x16x4_f32 only",
                                   "mfma_16x16x4 + 4 v_fma", "mfma_16x16x4 + 8 v_fma", "mfma_16x16x4 + 12 v_fma", "ds_write_b32"};

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* span, int iters) {
    __shared__ float lds[256 * 9];
    float f[8];
    f2 p[8];
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < 8; q++) f[q] = 1.0f + 1e-3f * (threadIdx.x + q), p[q] = (f2){f[q], f[q] * 0.5f};
    const float c = 0.999f;
    const f2 c2 = {0.999f, 0.998f};
    unsigned sreg = 0;
    const unsigned lp = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)(&lds[0])) + threadIdx.x * 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q]) : "v"(c));
                if (KIND == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[q]) : "v"(c2));
                if (KIND == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[q]) : "v"(c2));
                if (KIND == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[q]) : "v"(c2));
                if (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(f[q]));
                if (KIND == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[q]));
                if (KIND == DPPQ) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(f[q]) : "v"(f[(q + 4) & 7]));
                if (KIND == DPPROW) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(f[q]) : "v"(f[(q + 4) & 7]));
                if (KIND == SWAP32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(f[q]), "+v"(f[(q + 4) & 7]));
                if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[q]) : "v"(c) : "vcc");
                if (KIND == CMP) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[q]), "v"(c) : "vcc");
                if (KIND == MINF) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[q]) : "v"(c));
                if (KIND == FMA_SALU) asm volatile("v_fma_f32 %0, %0, %2, %2\n\ts_add_u32 %1, %1, 1" : "+v"(f[q]), "+s"(sreg) : "v"(c));
                if (KIND == FMA_SALU2) asm volatile("v_fma_f32 %0, %0, %2, %2\n\ts_add_u32 %1, %1, 1\n\ts_lshl_b32 %1, %1, 1" : "+v"(f[q]), "+s"(sreg) : "v"(c));
                if (KIND == LDSW) asm volatile("ds_write_b32 %0, %1 offset:0" : : "v"(lp), "v"(f[q]) : "memory");
            }
            if (KIND == MFMA_ONLY || KIND == MFMA_VALU4 || KIND == MFMA_VALU8 || KIND == MFMA_VALU12) {
                // per rep: two MFMAs on one accumulator chain (the reduction's pattern) + n plain VALU
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(f[0]), "v"(c));
                const int nv = KIND == MFMA_VALU4 ? 4 : KIND == MFMA_VALU8 ? 8 : KIND == MFMA_VALU12 ? 12 : 0;
#pragma unroll
                for (int q = 0; q < nv; q++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[1 + (q % 7)]) : "v"(c));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = (float)sreg + acc[0] + acc[1] + acc[2] + acc[3];
    for (int q = 0; q < 8; q++) s += f[q] + p[q].x + p[q].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) span[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(float* out, unsigned long long* span, int iters) {
    int cus = 256;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, span, iters);  // warm-up
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, span, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), span, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2], mx = (double)h.back();
        // instructions of the kind per wave (the MFMA kinds: per rep 1 MFMA + nv VALU)
        const bool mf = KIND >= MFMA_ONLY && KIND <= MFMA_VALU12;
        const double per_wave = mf ? (double)iters * 8 : (double)iters * 64;
        // s_memtime / readcyclecounter ticks at a fixed 100 MHz on some parts: report both the tick-based number and the wall-based one
        const double wall_cyc_2p4 = ms * 1e-3 * 2.4e9;
        printf("%-28s wps=%d  ticks/inst/SIMD med %.2f max %.2f   wall %.3f ms -> %.2f cyc@2.4GHz per inst per SIMD\n", names[KIND], wps,
               med / (per_wave * wps), mx / (per_wave * wps), ms, wall_cyc_2p4 / (per_wave * wps));
    }
}

int main() {
    float* out;
    unsigned long long* span;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&span, 256 * 8 * 4 * 8);
    const int iters = 2000;
    run<FMA>(out, span, iters);
    run<PKFMA>(out, span, iters);
    run<PKMUL>(out, span, iters);
    run<PKADD>(out, span, iters);
    run<EXP>(out, span, iters);
    run<RCP>(out, span, iters);
    run<DPPQ>(out, span, iters);
    run<DPPROW>(out, span, iters);
    run<SWAP32>(out, span, iters);
    run<CNDMASK>(out, span, iters);
    run<CMP>(out, span, iters);
    run<MINF>(out, span, iters);
    run<FMA_SALU>(out, span, iters);
    run<FMA_SALU2>(out, span, iters);
    run<MFMA_ONLY>(out, span, iters);
    run<MFMA_VALU4>(out, span, iters);
    run<MFMA_VALU8>(out, span, iters);
    run<MFMA_VALU12>(out, span, iters);
    run<LDSW>(out, span, iters);
    return 0;
}
