// Densification / pruning of the Gaussian set on the device.
//
// Replaces the tensor surgery of GaussianModelDPSRDynamicAnchor.densify_and_prune
// (R/scene/gaussian_model_dpsr_dynamic_anchor.py:462-551: densify_and_clone, densify_and_split, prune, with
// cat_tensors_to_optimizer :421-446, _prune_optimizer :383-401, prune_points :403-419) -- in the reference ~60 boolean
// index / cat / repeat kernels and three re-allocations of every parameter and Adam moment per call -- by ONE decision
// kernel, one scan, one index kernel and one gather launch that writes every surviving row of every tensor exactly once.
//
// The reference's sequence  clone -> split (+ removal of the split originals) -> prune  is order-equivalent to
//   out = [ originals kept | clones kept | first split children kept | second split children kept ]
// where, with grad = accum / denom (NaN -> 0), smax = max(exp(scaling)):
//   clone  = grad >= thr and smax <= percent_dense * extent        split = grad >= thr and smax >  percent_dense * extent
//   pruned(x) = sigmoid(opacity) < min_opacity or (size limit on and max(exp(scaling_x)) > 0.1 extent)
//   original kept = not split and not pruned(original);  clone kept = clone and not pruned(original)   (a clone is a copy)
//   child kept    = split and not pruned(child),  child scaling = log(exp(scaling) / (0.8 * 2))
// (the screen-size criterion of prune() never fires inside densify_and_prune: densification_postfix has just zeroed
// max_radii2D, :458-460).  Adam moments travel with kept originals and are zero for every new row (:429-432).
// Split children: xyz = R(q / |q|) (exp(scaling) * z) + xyz with z ~ N(0, 1) SUPPLIED by the caller (shape [2][P][3]) --
// a shared-seed generator makes data-parallel replicas draw identical samples.
#include "dgm_common.hpp"

namespace dgm {
void set_last_error(const char* msg);  // c_api.hip

// flags: bit 0 original kept, bit 1 clone kept, bit 2 split children kept
__global__ void __launch_bounds__(256)
densify_flags_kernel(int P, const float* __restrict__ grad_accum, const float* __restrict__ denom,
                     const float* __restrict__ scaling, const float* __restrict__ opacity, float grad_thr,
                     float dense_extent, float min_opacity, float big_extent, uint8_t* __restrict__ flags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float g = grad_accum[i] / denom[i];
    if (g != g) g = 0.f;
    const float s0 = expf(scaling[3 * i]), s1 = expf(scaling[3 * i + 1]), s2 = expf(scaling[3 * i + 2]);
    const float smax = fmaxf(s0, fmaxf(s1, s2));
    const bool hot = g >= grad_thr;
    const bool clone = hot && smax <= dense_extent, split = hot && smax > dense_extent;
    const float op = 1.f / (1.f + expf(-opacity[i]));
    const bool low = op < min_opacity;
    const bool pruned_o = low || smax > big_extent;
    // the children's scaling parameter is log(s / 1.6); what prune() compares is exp() of it
    const float c0 = expf(logf(s0 / 1.6f)), c1 = expf(logf(s1 / 1.6f)), c2 = expf(logf(s2 / 1.6f));
    const bool pruned_c = low || fmaxf(c0, fmaxf(c1, c2)) > big_extent;
    flags[i] = (uint8_t)((!split && !pruned_o ? 1 : 0) | (clone && !pruned_o ? 2 : 0) | (split && !pruned_c ? 4 : 0));
}

// plain keep-mask variant (prune_points): flags = keep ? 1 : 0 is supplied by the caller.

// three exclusive scans over the flag bits in one single-workgroup pass; totals[0..2] = number kept / cloned / split
__global__ void __launch_bounds__(1024)
densify_scan_kernel(int P, const uint8_t* __restrict__ flags, unsigned* __restrict__ off_keep,
                    unsigned* __restrict__ off_clone, unsigned* __restrict__ off_split, unsigned* __restrict__ totals) {
    __shared__ unsigned wave_tot[3][16];
    __shared__ unsigned carry[3];
    if (threadIdx.x < 3) carry[threadIdx.x] = 0;
    __syncthreads();
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    for (int base = 0; base < P; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned f = i < P ? flags[i] : 0u;
        unsigned v[3] = {f & 1u, (f >> 1) & 1u, (f >> 2) & 1u}, inc[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            inc[c] = wave_inclusive_scan_u32(v[c]);
            if (lane == 63) wave_tot[c][wv] = inc[c];
        }
        __syncthreads();
        unsigned pre[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            pre[c] = carry[c];
            for (int w = 0; w < wv; w++) pre[c] += wave_tot[c][w];
        }
        if (i < P) {
            off_keep[i] = pre[0] + inc[0] - v[0];
            off_clone[i] = pre[1] + inc[1] - v[1];
            off_split[i] = pre[2] + inc[2] - v[2];
        }
        __syncthreads();
        if (threadIdx.x == 1023) {
#pragma unroll
            for (int c = 0; c < 3; c++) carry[c] = pre[c] + inc[c];
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) totals[threadIdx.x] = carry[threadIdx.x];
}

// src[j] = source row | kind << 30   (kind 0 original, 1 clone, 2 / 3 first / second split child)
__global__ void __launch_bounds__(256)
densify_index_kernel(int P, const uint8_t* __restrict__ flags, const unsigned* __restrict__ off_keep,
                     const unsigned* __restrict__ off_clone, const unsigned* __restrict__ off_split, unsigned K, unsigned C,
                     unsigned S, unsigned* __restrict__ src) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const unsigned f = flags[i];
    if (f & 1u) src[off_keep[i]] = (unsigned)i;
    if (f & 2u) src[K + off_clone[i]] = (unsigned)i | (1u << 30);
    if (f & 4u) {
        src[K + C + off_split[i]] = (unsigned)i | (2u << 30);
        src[K + C + S + off_split[i]] = (unsigned)i | (3u << 30);
    }
}

static constexpr int GATHER_MAX = 24;
struct GatherJobs {
    const float* in[GATHER_MAX];
    float* out[GATHER_MAX];
    int width[GATHER_MAX];     // floats per row
    int is_moment[GATHER_MAX]; // 1: rows of new points are zero (Adam exp_avg / exp_avg_sq)
    int count;
};

// out_t[j][:] = in_t[src[j]][:] for every tensor t of the table (blockIdx.y = tensor); new rows of moments are zero
__global__ void __launch_bounds__(256) densify_gather_kernel(int Pn, const unsigned* __restrict__ src, const GatherJobs jobs) {
    const int t = blockIdx.y;
    const int w = jobs.width[t];
    const float* __restrict__ in = jobs.in[t];
    float* __restrict__ out = jobs.out[t];
    const bool mom = jobs.is_moment[t] != 0;
    const size_t total = (size_t)Pn * w;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const unsigned j = (unsigned)(e / w), c = (unsigned)(e - (size_t)j * w);
        const unsigned s = src[j];
        const unsigned row = s & 0x3fffffffu, kind = s >> 30;
        out[e] = (mom && kind != 0u) ? 0.f : in[(size_t)row * w + c];
    }
}

// rows of split children: position sample and shrunken scale (densify_and_split, :476-484; build_rotation,
// R/utils/general_utils.py:130-149)
__global__ void __launch_bounds__(256)
densify_split_kernel(int P, unsigned first, unsigned count, const unsigned* __restrict__ src, const float* __restrict__ xyz_in,
                     const float* __restrict__ scaling_in, const float* __restrict__ rot_in, const float* __restrict__ z,
                     float* __restrict__ xyz_out, float* __restrict__ scaling_out) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    const unsigned j = first + t;
    const unsigned s = src[j];
    const unsigned i = s & 0x3fffffffu, copy = (s >> 30) - 2u;
    const float s0 = expf(scaling_in[3 * i]), s1 = expf(scaling_in[3 * i + 1]), s2 = expf(scaling_in[3 * i + 2]);
    const float* zz = z + ((size_t)copy * P + i) * 3;
    const float v0 = s0 * zz[0], v1 = s1 * zz[1], v2 = s2 * zz[2];
    float qr = rot_in[4 * i], qx = rot_in[4 * i + 1], qy = rot_in[4 * i + 2], qz = rot_in[4 * i + 3];
    const float n = sqrtf(qr * qr + qx * qx + qy * qy + qz * qz);
    qr /= n, qx /= n, qy /= n, qz /= n;
    const float R00 = 1.f - 2.f * (qy * qy + qz * qz), R01 = 2.f * (qx * qy - qr * qz), R02 = 2.f * (qx * qz + qr * qy);
    const float R10 = 2.f * (qx * qy + qr * qz), R11 = 1.f - 2.f * (qx * qx + qz * qz), R12 = 2.f * (qy * qz - qr * qx);
    const float R20 = 2.f * (qx * qz - qr * qy), R21 = 2.f * (qy * qz + qr * qx), R22 = 1.f - 2.f * (qx * qx + qy * qy);
    xyz_out[3 * j + 0] = R00 * v0 + R01 * v1 + R02 * v2 + xyz_in[3 * i + 0];
    xyz_out[3 * j + 1] = R10 * v0 + R11 * v1 + R12 * v2 + xyz_in[3 * i + 1];
    xyz_out[3 * j + 2] = R20 * v0 + R21 * v1 + R22 * v2 + xyz_in[3 * i + 2];
    scaling_out[3 * j + 0] = logf(s0 / 1.6f);
    scaling_out[3 * j + 1] = logf(s1 / 1.6f);
    scaling_out[3 * j + 2] = logf(s2 / 1.6f);
}

// per-iteration statistics (R/train.py:489-496 + add_densification_stats, gaussian_model_dpsr_dynamic_anchor.py:679-682):
// for the Gaussians with radii > 0: max_radii2D = max(max_radii2D, radii), xyz_gradient_accum += |dL/dmeans2D[:2]|,
// denom += 1.  One thread per Gaussian, in place.
__global__ void __launch_bounds__(256)
densify_stats_kernel(int P, const float* __restrict__ grad2d, const int* __restrict__ radii, float* __restrict__ max_radii2D,
                     float* __restrict__ grad_accum, float* __restrict__ denom) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r <= 0) return;
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
    if (grad2d != nullptr) {
        const float gx = grad2d[3 * (size_t)i], gy = grad2d[3 * (size_t)i + 1];
        grad_accum[i] += sqrtf(gx * gx + gy * gy);
        denom[i] += 1.0f;
    }
}

}  // namespace dgm

using namespace dgm;

namespace {
int dfail(const char* m) {
    dgm::set_last_error(m);
    return 1;
}
}  // namespace

extern "C" {

int dgm_densify_stats(int P, const float* grad2d, const int* radii, float* max_radii2D, float* grad_accum, float* denom,
                      void* stream) {
    if (P <= 0) return 0;
    if (!radii || !max_radii2D || (grad2d && (!grad_accum || !denom))) return dfail("densify_stats: NULL pointer");
    hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, grad2d, radii,
                       max_radii2D, grad_accum, denom);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return dfail(hipGetErrorString(e));
    return 0;
}

// Decision + scans.  scratch: (P + 3 P * 4 + 16) bytes = flags | off_keep | off_clone | off_split | totals[3] (u32);
// after the stream has reached this point, totals (at dgm_densify_totals_offset(P)) hold K, C, S.
size_t dgm_densify_scratch_bytes(int P) { return (size_t)((P + 255) / 256 * 256) * 13 + 256; }
size_t dgm_densify_totals_offset(int P) { return (size_t)((P + 255) / 256 * 256) * 13; }

int dgm_densify_decide(int P, const float* grad_accum, const float* denom, const float* scaling, const float* opacity,
                       float grad_threshold, float dense_extent, float min_opacity, float big_extent,
                       const uint8_t* keep_mask, char* scratch, void* stream) {
    if (P <= 0) return 0;
    if (!scratch || (!keep_mask && (!grad_accum || !denom || !scaling || !opacity))) return dfail("densify_decide: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const size_t Pp = (size_t)((P + 255) / 256 * 256);
    uint8_t* flags = (uint8_t*)scratch;
    unsigned* off = (unsigned*)(scratch + Pp);
    unsigned* totals = (unsigned*)(scratch + Pp * 13);
    if (keep_mask) {  // prune_points(): the caller's keep mask (bytes 0 / 1) is the flag array
        if (hipMemcpyAsync(flags, keep_mask, (size_t)P, hipMemcpyDeviceToDevice, st) != hipSuccess) return dfail("densify_decide: copy failed");
    } else {
        hipLaunchKernelGGL(densify_flags_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, grad_accum, denom, scaling, opacity,
                           grad_threshold, dense_extent, min_opacity, big_extent, flags);
    }
    hipLaunchKernelGGL(densify_scan_kernel, dim3(1), dim3(1024), 0, st, P, flags, off, off + Pp, off + 2 * Pp, totals);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return dfail(hipGetErrorString(e));
    return 0;
}

// Gather every tensor into its new size Pn = K + C + 2 S.  `src_scratch` needs Pn * 4 bytes.  Tensor order is free;
// xyz / scaling / rotation indices say which table entries are the position, log-scale and quaternion parameters (their
// split-children rows are rewritten from the samples z[2][P][3]); pass S == 0 and z == NULL for plain pruning.
int dgm_densify_apply(int P, unsigned K, unsigned C, unsigned S, const char* scratch, unsigned* src_scratch, int n_tensors,
                      const float* const* in, float* const* out, const int* width, const int* is_moment, int xyz_index,
                      int scaling_index, int rotation_index, const float* z, void* stream) {
    if (P <= 0) return 0;
    const unsigned Pn = K + C + 2u * S;
    if (Pn == 0) return 0;
    if (!scratch || !src_scratch || !in || !out || !width || !is_moment) return dfail("densify_apply: NULL pointer");
    if (n_tensors < 0 || n_tensors > GATHER_MAX) return dfail("densify_apply: too many tensors");
    if (S > 0 && (!z || xyz_index < 0 || scaling_index < 0 || rotation_index < 0 || xyz_index >= n_tensors ||
                  scaling_index >= n_tensors || rotation_index >= n_tensors))
        return dfail("densify_apply: split needs the samples and the xyz / scaling / rotation tensors");
    hipStream_t st = (hipStream_t)stream;
    const size_t Pp = (size_t)((P + 255) / 256 * 256);
    const uint8_t* flags = (const uint8_t*)scratch;
    const unsigned* off = (const unsigned*)(scratch + Pp);
    hipLaunchKernelGGL(densify_index_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, flags, off, off + Pp, off + 2 * Pp, K, C,
                       S, src_scratch);
    GatherJobs jobs;
    jobs.count = n_tensors;
    int wmax = 1;
    for (int t = 0; t < n_tensors; t++) {
        if (!in[t] || !out[t] || width[t] <= 0) return dfail("densify_apply: bad tensor entry");
        jobs.in[t] = in[t], jobs.out[t] = out[t], jobs.width[t] = width[t], jobs.is_moment[t] = is_moment[t];
        if (width[t] > wmax) wmax = width[t];
    }
    if (n_tensors > 0) {
        size_t blocks = ((size_t)Pn * wmax + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(densify_gather_kernel, dim3((unsigned)blocks, n_tensors), dim3(256), 0, st, (int)Pn, src_scratch, jobs);
    }
    if (S > 0)
        hipLaunchKernelGGL(densify_split_kernel, dim3((2 * S + 255) / 256), dim3(256), 0, st, P, K + C, 2 * S, src_scratch,
                           in[xyz_index], in[scaling_index], in[rotation_index], z, out[xyz_index], out[scaling_index]);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return dfail(hipGetErrorString(e));
    return 0;
}

}  // extern "C"
