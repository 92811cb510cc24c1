// LDS canary: small workgroups that fill their LDS with a pattern and keep re-reading it while other kernels share the CU.
// Any change of the pattern means ANOTHER workgroup wrote into this one's LDS allocation.  (tools/exp/lds_canary.py)
#include <hip/hip_runtime.h>

__global__ void __launch_bounds__(64) lds_canary_kernel(unsigned* __restrict__ out, int iters, int words) {
    extern __shared__ unsigned buf[];
    const unsigned tid = threadIdx.x, salt = 0x9e3779b9u * (blockIdx.x + 1);
    for (int k = tid; k < words; k += 64) buf[k] = salt ^ (unsigned)k;
    __builtin_amdgcn_s_waitcnt(0);
    unsigned bad = 0, first = 0xffffffffu, seen = 0;
    for (int it = 0; it < iters; it++) {
        for (int k = tid; k < words; k += 64) {
            const unsigned v = buf[k];
            if (v != (salt ^ (unsigned)k)) {
                bad++;
                if (first == 0xffffffffu) first = (unsigned)k, seen = v;
                buf[k] = salt ^ (unsigned)k;
            }
        }
        __builtin_amdgcn_s_sleep(64);
    }
    if (bad) {
        atomicAdd(out, bad);
        atomicAdd(out + 1, 1u);
        out[2] = first, out[3] = seen, out[4] = blockIdx.x;
    }
}

extern "C" int lds_canary_launch(void* stream, unsigned* out, int blocks, int iters, int words) {
    hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(64), (size_t)words * 4, (hipStream_t)stream, out, iters, words);
    return (int)hipGetLastError();
}
