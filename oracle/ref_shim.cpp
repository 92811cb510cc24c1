// ref_shim.cpp -- TEST INFRASTRUCTURE.  A thin extern "C" wrapper (written for this repo) around the REFERENCE's own
// C++ entry points CudaRasterizer::Rasterizer::{forward,backward} (DGR/cuda_rasterizer/rasterizer.h:20-85) and
// SimpleKNN::knn (KNN/simple_knn.h), so that the reference kernels -- translated at build time by hipify-perl from
// the sources where they lie under /root/reference and compiled for gfx950 by oracle/build_ref.sh into
// oracle/_ref/ -- can be run on the GPU box as a second oracle.  No reference source is stored in this repository.
#include <hip/hip_runtime.h>

#include <functional>
#include <vector>

#include "rasterizer.h"  // hipified reference header (from the temporary build directory)
#ifdef REF_WITH_KNN
#include "simple_knn.h"
#endif

namespace {
struct Buf {
    char* p = nullptr;
    size_t n = 0;
    char* get(size_t N) {
        if (N > n) {
            if (p) (void)hipFree(p);
            (void)hipMalloc((void**)&p, N);
            n = N;
        }
        return p;
    }
};
Buf g_geom, g_bin, g_img;
size_t g_sizes[3];
}  // namespace

extern "C" {

// All pointers are device pointers.  Returns num_rendered.  The three state buffers stay owned by the shim; their
// base addresses / sizes are reported through `state` (6 x uint64: geom ptr, size, binning ptr, size, image ptr, size).
int ref_forward(int P, int D, int M, const float* background, int W, int H, const float* means3D, const float* shs,
                const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                const float* cam_pos, float tan_fovx, float tan_fovy, float* out_color, int* radii,
                unsigned long long* state) {
    std::function<char*(size_t)> fg = [&](size_t N) { g_sizes[0] = N; return g_geom.get(N); };
    std::function<char*(size_t)> fb = [&](size_t N) { g_sizes[1] = N; return g_bin.get(N); };
    std::function<char*(size_t)> fi = [&](size_t N) { g_sizes[2] = N; return g_img.get(N); };
    (void)hipMemset(out_color, 0, sizeof(float) * 3 * (size_t)W * H);
    (void)hipMemset(radii, 0, sizeof(int) * (size_t)P);
    int n = CudaRasterizer::Rasterizer::forward(fg, fb, fi, P, D, M, background, W, H, means3D, shs, colors_precomp,
                                                opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                                                projmatrix, cam_pos, tan_fovx, tan_fovy, false, out_color, radii, false);
    (void)hipDeviceSynchronize();
    if (state) {
        state[0] = (unsigned long long)g_geom.p, state[1] = g_sizes[0];
        state[2] = (unsigned long long)g_bin.p, state[3] = g_sizes[1];
        state[4] = (unsigned long long)g_img.p, state[5] = g_sizes[2];
    }
    return n;
}

// Gradient arrays must be zero-filled by the caller (the reference accumulates with atomics).
void ref_backward(int P, int D, int M, int R, const float* background, int W, int H, const float* means3D,
                  const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                  const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                  const float* campos, float tan_fovx, float tan_fovy, const int* radii, const float* dL_dpix,
                  float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                  float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot) {
    CudaRasterizer::Rasterizer::backward(P, D, M, R, background, W, H, means3D, shs, colors_precomp, scales,
                                         scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos,
                                         tan_fovx, tan_fovy, radii, g_geom.p, g_bin.p, g_img.p, dL_dpix, dL_dmean2D,
                                         dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
                                         dL_drot, false);
    (void)hipDeviceSynchronize();
}

#ifdef REF_WITH_KNN
void ref_knn(int P, float* points, float* mean_dists) {
    SimpleKNN::knn(P, (float3*)points, mean_dists);
    (void)hipDeviceSynchronize();
}
#endif
}
