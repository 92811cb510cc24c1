// simple-knn for gfx950: mean squared distance of every point to its 3 nearest neighbours.
//
// Replaces SimpleKNN::knn (KNN/simple_knn.cu:185-221: cub reductions, coord2Morton :45-70,
// cub::DeviceRadixSort, boxMinMax :78-117, boxMeanDist :147-183).  The result is the EXACT 3-NN mean of the
// fp32 values d2 = dx*dx + dy*dy + dz*dz (evaluated left to right without FMA: this file is built with
// -ffp-contract=off), i.e. bit-identical to the reference algorithm, whose box/reject heuristic only prunes.
//
// MI355X design: Morton order (same 30-bit code as the reference) only provides spatial coherence; the
// search is WAVE-cooperative: a wave owns 64 consecutive sorted points, tests every box AABB per lane with
// scalar-loaded box data, and if ANY lane needs a box all 64 lanes scan it with wave-uniform (scalar cache)
// candidate loads -- extra candidates can only confirm the exact minimum-3, and control flow stays uniform.
// Boxes hold 256 points (the reference uses 1024; the box size does not affect the result).  No host
// synchronisation (the reference does two D2H copies for the bounding box) and no device allocation
// (the reference does seven); scratch comes from one arena cached per stream.
#include <float.h>

#include "dgm_common.hpp"

#pragma clang fp contract(off)

namespace dgm {

void launch_scan_blocks(hipStream_t st, int n, const unsigned* in, unsigned* out, unsigned* total);

static constexpr int KNN_BOX = 256;
static constexpr int RS_TILE = 2048;  // keys per radix-sort workgroup

__device__ __forceinline__ unsigned prep_morton(unsigned x) {  // KNN/simple_knn.cu:45-52
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

// stage 1: partial bounding boxes (both reductions start from 0 like the reference, simple_knn.cu:191-199)
__global__ void __launch_bounds__(256)
knn_bbox_partial_kernel(int P, const float* __restrict__ pts, float* __restrict__ partial) {
    __shared__ float red[6][256];
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float v = pts[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        red[a][threadIdx.x] = mn[a];
        red[3 + a][threadIdx.x] = mx[a];
    }
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
#pragma unroll
            for (int a = 0; a < 3; a++) {
                red[a][threadIdx.x] = fminf(red[a][threadIdx.x], red[a][threadIdx.x + off]);
                red[3 + a][threadIdx.x] = fmaxf(red[3 + a][threadIdx.x], red[3 + a][threadIdx.x + off]);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 6) partial[blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void knn_bbox_final_kernel(int nparts, const float* __restrict__ partial, float* __restrict__ bbox) {
    if (threadIdx.x < 6) {
        float v = 0.f;
        for (int b = 0; b < nparts; b++)
            v = threadIdx.x < 3 ? fminf(v, partial[b * 6 + threadIdx.x]) : fmaxf(v, partial[b * 6 + threadIdx.x]);
        bbox[threadIdx.x] = v;
    }
}

__global__ void knn_morton_kernel(int P, const float* __restrict__ pts, const float* __restrict__ bbox,
                                  unsigned* __restrict__ codes, unsigned* __restrict__ idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    unsigned c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {  // KNN/simple_knn.cu:54-61
        const float mn = bbox[a], mx = bbox[3 + a];
        c[a] = prep_morton(f2u_sat(((pts[3 * i + a] - mn) / (mx - mn)) * ((1 << 10) - 1)));
    }
    codes[i] = c[0] | (c[1] << 1) | (c[2] << 2);
    idx[i] = (unsigned)i;
}

// ---- stable LSD radix sort of (key, value) pairs, 8 bits per pass -----------------------------------------
__global__ void __launch_bounds__(256)
rs_hist_kernel(int n, int shift, int nblk, const unsigned* __restrict__ keys, unsigned* __restrict__ hist) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE;
    for (int i = base + threadIdx.x; i < min(n, base + RS_TILE); i += 256) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    __syncthreads();
    hist[threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];  // digit-major: one scan gives global offsets
}

__global__ void __launch_bounds__(256)
rs_scatter_kernel(int n, int shift, int nblk, const unsigned* __restrict__ keys_in, const unsigned* __restrict__ vals_in,
                  const unsigned* __restrict__ scanned, unsigned* __restrict__ keys_out, unsigned* __restrict__ vals_out) {
    __shared__ unsigned basep[256];
    __shared__ unsigned cnt[4][256];
    basep[threadIdx.x] = scanned[threadIdx.x * nblk + blockIdx.x];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const int base = blockIdx.x * RS_TILE;
    for (int r = 0; r < RS_TILE / 256; r++) {
#pragma unroll
        for (int w = 0; w < 4; w++) cnt[w][threadIdx.x] = 0;
        __syncthreads();
        const int i = base + r * 256 + threadIdx.x;
        const bool act = i < n;
        unsigned k = 0, v = 0, d = 0;
        if (act) {
            k = keys_in[i];
            v = vals_in[i];
            d = (k >> shift) & 255u;
        }
        // lanes of this wave holding the same digit
        unsigned long long same = __ballot(act);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const unsigned rank = __popcll(same & ((1ull << lane) - 1ull));
        if (act && rank == 0) cnt[wv][d] = (unsigned)__popcll(same);
        __syncthreads();
        if (act) {
            unsigned pre = basep[d];
            for (int w = 0; w < wv; w++) pre += cnt[w][d];
            keys_out[pre + rank] = k;
            vals_out[pre + rank] = v;
        }
        __syncthreads();
        basep[threadIdx.x] += cnt[0][threadIdx.x] + cnt[1][threadIdx.x] + cnt[2][threadIdx.x] + cnt[3][threadIdx.x];
        __syncthreads();
    }
}

// sorted positions (xyz + original index) and per-box AABBs
__global__ void __launch_bounds__(KNN_BOX)
knn_box_kernel(int P, const float* __restrict__ pts, const unsigned* __restrict__ idx_sorted,
               float4* __restrict__ sorted, float* __restrict__ boxes) {
    __shared__ float red[6][KNN_BOX];
    const int i = blockIdx.x * KNN_BOX + threadIdx.x;
    float p[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, q[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (i < P) {
        const unsigned o = idx_sorted[i];
        const float x = pts[3 * o], y = pts[3 * o + 1], z = pts[3 * o + 2];
        sorted[i] = make_float4(x, y, z, __uint_as_float(o));
        p[0] = q[0] = x;
        p[1] = q[1] = y;
        p[2] = q[2] = z;
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        red[a][threadIdx.x] = p[a];
        red[3 + a][threadIdx.x] = q[a];
    }
    __syncthreads();
    for (int off = KNN_BOX / 2; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
#pragma unroll
            for (int a = 0; a < 3; a++) {
                red[a][threadIdx.x] = fminf(red[a][threadIdx.x], red[a][threadIdx.x + off]);
                red[3 + a][threadIdx.x] = fmaxf(red[3 + a][threadIdx.x], red[3 + a][threadIdx.x + off]);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 6) boxes[blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0];
}

__device__ __forceinline__ void update3(float px, float py, float pz, float cx, float cy, float cz, float& b0, float& b1,
                                        float& b2) {  // updateKBest<3>, KNN/simple_knn.cu:131-145
    const float dx = cx - px, dy = cy - py, dz = cz - pz;
    float dist = dx * dx + dy * dy + dz * dz;
    if (b0 > dist) {
        const float t = b0;
        b0 = dist;
        dist = t;
    }
    if (b1 > dist) {
        const float t = b1;
        b1 = dist;
        dist = t;
    }
    if (b2 > dist) b2 = dist;
}

__global__ void __launch_bounds__(256)
knn_search_kernel(int P, int nboxes, const float4* __restrict__ sorted, const float* __restrict__ boxes,
                  float* __restrict__ dists) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool act = i < P;
    const float4 me = act ? sorted[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    if (act) {  // reject radius from the +-3 Morton neighbours (KNN/simple_knn.cu:156-163)
        for (int j = max(0, i - 3); j <= min(P - 1, i + 3); j++) {
            if (j == i) continue;
            const float4 c = sorted[j];
            update3(me.x, me.y, me.z, c.x, c.y, c.z, b0, b1, b2);
        }
    }
    const float reject = b2;
    b0 = b1 = b2 = FLT_MAX;
    for (int b = 0; b < nboxes; b++) {
        const float* bx = boxes + 6 * b;  // wave-uniform => scalar loads
        float d[3] = {0.f, 0.f, 0.f};
        const float pc[3] = {me.x, me.y, me.z};
#pragma unroll
        for (int a = 0; a < 3; a++) {  // distBoxPoint, KNN/simple_knn.cu:119-129
            const float lo = bx[a], hi = bx[3 + a];
            if (pc[a] < lo || pc[a] > hi) d[a] = fminf(fabsf(pc[a] - lo), fabsf(pc[a] - hi));
        }
        const float dist = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        const bool want = act && !(dist > reject || dist > b2);
        if (__ballot(want) == 0ull) continue;
        const int j0 = b * KNN_BOX, j1 = min(P, j0 + KNN_BOX);
        for (int j = j0; j < j1; j++) {
            const float4 c = sorted[j];  // wave-uniform address
            if (j != i) update3(me.x, me.y, me.z, c.x, c.y, c.z, b0, b1, b2);
        }
    }
    if (act) dists[__float_as_uint(me.w)] = (b0 + b1 + b2) / 3.0f;
}

// ---- host side ------------------------------------------------------------------------------------------------
size_t knn_scratch_bytes(int P) {
    const size_t n = (size_t)P;
    const size_t nblk = (n + RS_TILE - 1) / RS_TILE;
    const size_t nboxes = (n + KNN_BOX - 1) / KNN_BOX;
    size_t o = 0;
    auto take = [&](size_t b) { o = align_up(o + b, 256); };
    take(128 * 6 * 4);       // partial bboxes
    take(8 * 4);             // bbox
    take(n * 4);             // codes A
    take(n * 4);             // codes B
    take(n * 4);             // idx A
    take(n * 4);             // idx B
    take(256 * nblk * 4);    // hist
    take(256 * nblk * 4);    // scanned
    take(n * 16);            // sorted float4
    take(nboxes * 6 * 4);    // boxes
    return o + 256;
}

void launch_knn(hipStream_t st, int P, const float* pts, float* dists, char* scratch) {
    const size_t n = (size_t)P;
    const int nblk = (P + RS_TILE - 1) / RS_TILE;
    const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
    char* p = align_ptr(scratch);
    auto take = [&](size_t b) {
        char* at = p;
        p = align_ptr(p + b);
        return at;
    };
    float* partial = (float*)take(128 * 6 * 4);
    float* bbox = (float*)take(8 * 4);
    unsigned* codesA = (unsigned*)take(n * 4);
    unsigned* codesB = (unsigned*)take(n * 4);
    unsigned* idxA = (unsigned*)take(n * 4);
    unsigned* idxB = (unsigned*)take(n * 4);
    unsigned* hist = (unsigned*)take((size_t)256 * nblk * 4);
    unsigned* scanned = (unsigned*)take((size_t)256 * nblk * 4);
    float4* sorted = (float4*)take(n * 16);
    float* boxes = (float*)take((size_t)nboxes * 6 * 4);

    const int nparts = min(128, (P + 255) / 256);
    hipLaunchKernelGGL(knn_bbox_partial_kernel, dim3(nparts), dim3(256), 0, st, P, pts, partial);
    hipLaunchKernelGGL(knn_bbox_final_kernel, dim3(1), dim3(64), 0, st, nparts, partial, bbox);
    hipLaunchKernelGGL(knn_morton_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, pts, bbox, codesA, idxA);
    unsigned *kin = codesA, *kout = codesB, *vin = idxA, *vout = idxB;
    for (int pass = 0; pass < 4; pass++) {
        hipLaunchKernelGGL(rs_hist_kernel, dim3(nblk), dim3(256), 0, st, P, pass * 8, nblk, kin, hist);
        launch_scan_blocks(st, 256 * nblk, hist, scanned, nullptr);
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(nblk), dim3(256), 0, st, P, pass * 8, nblk, kin, vin, scanned, kout,
                           vout);
        unsigned* t = kin;
        kin = kout;
        kout = t;
        t = vin;
        vin = vout;
        vout = t;
    }
    hipLaunchKernelGGL(knn_box_kernel, dim3(nboxes), dim3(KNN_BOX), 0, st, P, pts, vin, sorted, boxes);
    hipLaunchKernelGGL(knn_search_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, nboxes, sorted, boxes, dists);
}

}  // namespace dgm
