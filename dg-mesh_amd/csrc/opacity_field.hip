// Opacity field of a Gaussian set on a regular grid (normal initialisation of the mesh branch).
//
// Replaces get_opacity_field_from_gaussians (R/utils/mesh_utils.py:7-76), a Python triple loop over num_blocks^3 blocks
// that builds [cells x gaussians x 3] tensors per block, with two kernels:
//   1. per Gaussian: covariance from scale / rotation (build_covariance_from_scaling_rotation,
//      R/utils/general_utils.py:152-170, quaternion normalised, Sigma = (R S)(R S)^T), its inverse by cofactors with the
//      reference's 1e-24 regulariser (gaussian_3d_coeff, :173-192) -> one 48-byte record (centre, 6 inverse entries, opacity);
//   2. one workgroup per block of split^3 cells: Gaussians are streamed 256 at a time, those whose CENTRE lies strictly
//      inside the block's bounding box grown by block_size * relax_ratio (and whose opacity exceeds the threshold) are
//      compacted into LDS by wave ballots, and every thread accumulates opacity * exp(power) for its cells
//      (power > 0 -> weight 0, as the reference's `power[power > 0] = -1e10`).
// Same cell coordinates (the caller passes torch.linspace's values), same selection rule, same per-pair arithmetic; sums
// run in Gaussian index order (the reference sums batches of 1024 first): fp32 rounding differences only.
#include "dgm_common.hpp"

namespace dgm {
void set_last_error(const char* msg);  // c_api.hip

__global__ void __launch_bounds__(256)
opacity_field_prep_kernel(int P, const float* __restrict__ xyz, const float* __restrict__ rot, const float* __restrict__ scale,
                          const float* __restrict__ opacity, float thr, float* __restrict__ recs) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float r = rot[4 * i], x = rot[4 * i + 1], y = rot[4 * i + 2], z = rot[4 * i + 3];
    const float nrm = sqrtf(r * r + x * x + y * y + z * z);
    r /= nrm, x /= nrm, y /= nrm, z /= nrm;
    const float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                           {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                           {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    const float s[3] = {scale[3 * i], scale[3 * i + 1], scale[3 * i + 2]};
    float L[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) L[a][b] = R[a][b] * s[b];
    float S[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) S[a][b] = L[a][0] * L[b][0] + L[a][1] * L[b][1] + L[a][2] * L[b][2];
    const float a = S[0][0], b = S[0][1], c = S[0][2], d = S[1][1], e = S[1][2], f = S[2][2];
    const float inv_det = 1.f / (a * d * f + 2.f * e * c * b - e * e * a - c * c * d - b * b * f + 1e-24f);
    float* o = recs + (size_t)i * 12;
    o[0] = xyz[3 * i], o[1] = xyz[3 * i + 1], o[2] = xyz[3 * i + 2];
    o[3] = (d * f - e * e) * inv_det;  // inv_a
    o[4] = (e * c - b * f) * inv_det;  // inv_b
    o[5] = (e * b - c * d) * inv_det;  // inv_c
    o[6] = (a * f - c * c) * inv_det;  // inv_d
    o[7] = (b * c - e * a) * inv_det;  // inv_e
    o[8] = (a * d - b * b) * inv_det;  // inv_f
    const float op = opacity[i];
    o[9] = op;
    o[10] = op > thr ? 1.f : 0.f;
    o[11] = 0.f;
}

static constexpr int OF_MAX_CELLS = 16;  // cells per thread: split^3 <= 4096

__global__ void __launch_bounds__(256)
opacity_field_kernel(int P, int res, int nb, int split, float margin, const float* __restrict__ coords,
                     const float* __restrict__ recs, float* __restrict__ occ) {
    __shared__ float sRec[256][10];
    __shared__ int sCount;
    __shared__ int sWave[4];
    const int bz = blockIdx.x % nb, by = (blockIdx.x / nb) % nb, bx = blockIdx.x / (nb * nb);
    const int x0 = bx * split, y0 = by * split, z0 = bz * split;
    const int nx = min(split, res - x0), ny = min(split, res - y0), nz = min(split, res - z0);
    if (nx <= 0 || ny <= 0 || nz <= 0) return;
    const float lo[3] = {coords[x0] - margin, coords[y0] - margin, coords[z0] - margin};
    const float hi[3] = {coords[x0 + nx - 1] + margin, coords[y0 + ny - 1] + margin, coords[z0 + nz - 1] + margin};
    const int cells = nx * ny * nz;
    float acc[OF_MAX_CELLS], cx[OF_MAX_CELLS], cy[OF_MAX_CELLS], cz[OF_MAX_CELLS];
#pragma unroll
    for (int k = 0; k < OF_MAX_CELLS; k++) {
        const int cidx = threadIdx.x + k * 256;
        acc[k] = 0.f;
        const int ix = cidx / (ny * nz), iy = (cidx / nz) % ny, iz = cidx % nz;
        const bool ok = cidx < cells;
        cx[k] = ok ? coords[x0 + ix] : 0.f, cy[k] = ok ? coords[y0 + iy] : 0.f, cz[k] = ok ? coords[z0 + iz] : 0.f;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int base = 0; base < P; base += 256) {
        const int g = base + threadIdx.x;
        bool in = false;
        float r[10];
        if (g < P) {
            const float* s = recs + (size_t)g * 12;
#pragma unroll
            for (int q = 0; q < 10; q++) r[q] = s[q];
            in = s[10] != 0.f && r[0] < hi[0] && r[1] < hi[1] && r[2] < hi[2] && r[0] > lo[0] && r[1] > lo[1] && r[2] > lo[2];
        }
        // order-preserving compaction: wave ballots + prefix over the four waves
        const unsigned long long bal = __ballot(in);
        if (lane == 0) sWave[wv] = __builtin_popcountll(bal);
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wv; w++) off += sWave[w];
        if (threadIdx.x == 255) sCount = off + sWave[3];
        if (in) {
            const int slot = off + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
#pragma unroll
            for (int q = 0; q < 10; q++) sRec[slot][q] = r[q];
        }
        __syncthreads();
        const int n = sCount;
        for (int j = 0; j < n; j++) {
            const float gx = sRec[j][0], gy = sRec[j][1], gz = sRec[j][2];
            const float ia = sRec[j][3], ib = sRec[j][4], ic = sRec[j][5], id = sRec[j][6], ie = sRec[j][7], iff = sRec[j][8];
            const float op = sRec[j][9];
#pragma unroll
            for (int k = 0; k < OF_MAX_CELLS; k++) {
                const float x = cx[k] - gx, y = cy[k] - gy, z = cz[k] - gz;
                const float power = -0.5f * (x * x * ia + y * y * id + z * z * iff) - x * y * ib - x * z * ic - y * z * ie;
                const float w = power > 0.f ? 0.f : expf(power);
                acc[k] += op * w;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < OF_MAX_CELLS; k++) {
        const int cidx = threadIdx.x + k * 256;
        if (cidx < cells) {
            const int ix = cidx / (ny * nz), iy = (cidx / nz) % ny, iz = cidx % nz;
            occ[((size_t)(x0 + ix) * res + (y0 + iy)) * res + (z0 + iz)] = acc[k];
        }
    }
}

}  // namespace dgm

using namespace dgm;

extern "C" {

size_t dgm_opacity_field_scratch_bytes(int P) { return (size_t)(P > 0 ? P : 0) * 12 * sizeof(float) + 256; }

// occ[res^3] (every cell written); coords[res] = the grid's coordinates along each axis (torch.linspace(-b, b, res));
// margin = (2 / num_blocks) * relax_ratio, the reference's block growth.
int dgm_opacity_field(int P, const float* xyz, const float* rotations, const float* scalings, const float* opacities,
                      float opacity_threshold, int res, int num_blocks, float margin, const float* coords, char* scratch,
                      float* occ, void* stream) {
    auto fail = [](const char* m) {
        dgm::set_last_error(m);
        return 1;
    };
    if (res <= 0 || num_blocks <= 0 || !coords || !occ) return fail("opacity_field: bad argument");
    const int split = res / num_blocks;
    if (split <= 0 || split * split * split > 256 * OF_MAX_CELLS) return fail("opacity_field: resolution / num_blocks must be in 1..16");
    const int nb = (res + split - 1) / split;  // torch's .split(split_size) yields a shorter last chunk when res % split != 0
    hipStream_t st = (hipStream_t)stream;
    if (P > 0) {
        if (!xyz || !rotations || !scalings || !opacities || !scratch) return fail("opacity_field: NULL pointer");
        hipLaunchKernelGGL(opacity_field_prep_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, xyz, rotations, scalings,
                           opacities, opacity_threshold, (float*)scratch);
    }
    hipLaunchKernelGGL(opacity_field_kernel, dim3(nb * nb * nb), dim3(256), 0, st, P > 0 ? P : 0, res, nb, split, margin, coords,
                       (const float*)scratch, occ);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(hipGetErrorString(e));
    return 0;
}

}  // extern "C"
