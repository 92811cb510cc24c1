// Which SIMD does wave w of a 512-thread workgroup land on?  (HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13])
// build: hipcc --offload-arch=gfx950 -O3 tools/hwid_micro.hip -o tools/bin/hwid_micro
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k(unsigned* out) {
    const unsigned id = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d;
    hipMalloc(&d, 4 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(4), dim3(512), 66560, 0, d);
    unsigned h[32];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; b++) {
        printf("block %d:", b);
        for (int w = 0; w < 8; w++) printf("  w%d simd %u wave_slot %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    return 0;
}
