// Tile binning for gfx950: builds, for every 16x16 tile, the depth-ordered list of Gaussians whose
// screen rectangle overlaps it.
//
// Replaces the reference chain cub::DeviceScan::InclusiveSum -> duplicateWithKeys -> 64-bit global
// cub::DeviceRadixSort::SortPairs -> identifyTileRanges (DGR/cuda_rasterizer/rasterizer_impl.cu:70-138,
// 277-317) and produces IDENTICAL `point_list` and `ranges` (tile-major, then depth bits ascending, ties
// by ascending Gaussian index -- exactly what a stable sort of the index-ordered emission yields).
//
// MI355X design (not a radix-sort port; ~20 B/instance of HBM traffic instead of ~150 B/instance):
//   1. count   : <=256 chunk workgroups, each owning a contiguous run of Gaussians, histogram their
//                instances per tile in LDS (ds_add, no global atomics) and store one histogram row;
//   2. colscan : per tile, exclusive prefix over the chunk rows + tile totals; scan_tiles: exclusive
//                scan over tiles -> tile segment starts, `ranges` (empty tiles stay (0,0) like the
//                reference's memset) and R;
//   3. scatter : same chunk workgroups, LDS cursors initialised to (tile start + chunk prefix), each
//                instance takes a slot with one LDS atomic and writes its 8-byte key
//                (depth_bits << 32 | gaussian) straight into its tile segment;
//   4. sort    : one workgroup per tile sorts its segment in LDS (normalised bitonic network on the
//                64-bit keys => deterministic order independent of the atomic arrival order) and emits
//                point_list plus upos[slot] = offs[g]+k, the instance's position in the per-Gaussian order (backward rows).
// Wave-cooperative rectangle expansion: a wave loads 64 Gaussians, then iterates over the lanes that own a
// non-empty rectangle (scalar bit loop on the ballot mask) and lets all 64 lanes cover that rectangle's
// tiles, so a Gaussian spanning thousands of tiles costs the same lane-cycles as many small ones.
#include "dgm_common.hpp"

namespace dgm {

// ---- generic single-workgroup exclusive scan of n u32 (n up to a few 100k) --------------------------------
__global__ void __launch_bounds__(1024)
scan_exclusive_kernel(int n, const unsigned* __restrict__ in, unsigned* __restrict__ out, unsigned* __restrict__ total) {
    __shared__ unsigned wave_tot[16];
    __shared__ unsigned carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned v = i < n ? in[i] : 0u;
        const unsigned inc = wave_inclusive_scan_u32(v);
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        unsigned pre = carry_s;
        for (int w = 0; w < wv; w++) pre += wave_tot[w];
        if (i < n) out[i] = pre + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = pre + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry_s;
}

// ---- shared by count and scatter: iterate all (gaussian, tile) instances of a chunk -------------------------
// F(tile, g, k) is invoked once per instance with all lanes of the wave active on different k.
template <typename F>
__device__ __forceinline__ void for_each_instance(int g, unsigned tt, unsigned rect, int gridx, F&& f) {
    unsigned long long m = __ballot(tt != 0u);
    while (m) {
        const int src = __builtin_ctzll(m);
        m &= m - 1;
        const unsigned tt_i = __builtin_amdgcn_readlane(tt, src);
        const unsigned rect_i = __builtin_amdgcn_readlane(rect, src);
        const int g_i = __builtin_amdgcn_readlane(g, src);
        unsigned xmin, ymin, w;
        unpack_rect(rect_i, xmin, ymin, w);
        // exact k / w for k*w < 2^32 via a 32x32->hi multiply by ceil(2^32 / w)  (w >= 2)
        const unsigned magic = w > 1 ? (0xFFFFFFFFu / w + 1u) : 0u;
        for (unsigned k = lane_id(); k < tt_i; k += 64) {
            const unsigned y = w > 1 ? __umulhi(k, magic) : k;
            const unsigned x = k - y * w;
            const unsigned tile = (ymin + y) * (unsigned)gridx + xmin + x;
            f(tile, g_i, k);
        }
    }
}

// count: also finishes the exclusive scan over Gaussians (offs) from the per-256 block offsets.
__global__ void __launch_bounds__(DGM_BIN_THREADS)
count_tiles_kernel(int P, int chunk, int tiles, int gridx, const unsigned* __restrict__ tiles_touched,
                   float* __restrict__ rec, const unsigned* __restrict__ block_offs, unsigned* __restrict__ offs,
                   unsigned* __restrict__ hist) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_hist[];
    __shared__ unsigned wave_tot[DGM_BIN_THREADS / 64];
    for (int t = threadIdx.x; t < tiles; t += DGM_BIN_THREADS) lds_hist[t] = 0u;
    __syncthreads();
    const int g0 = blockIdx.x * chunk, g1 = min(P, g0 + chunk);
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    for (int base = g0; base < g1; base += DGM_BIN_THREADS) {
        const int g = base + threadIdx.x;
        unsigned tt = 0u, rect = 0u;
        if (g < g1) {
            tt = tiles_touched[g];
            rect = __float_as_uint(rec[(size_t)g * DGM_REC_STRIDE + 9]);
        }
        // exclusive offset = block_offs[g / 256] + prefix inside the 256-group (4 waves)
        const unsigned inc = wave_inclusive_scan_u32(tt);
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        if (g < g1) {
            unsigned pre = block_offs[g / DGM_PRE_BLOCK];
            for (int w = wv & ~3; w < wv; w++) pre += wave_tot[w];
            const unsigned o = pre + inc - tt;
            offs[g] = o;
            rec[(size_t)g * DGM_REC_STRIDE + 10] = __uint_as_float(o);
        }
        for_each_instance(g, tt, rect, gridx, [&](unsigned tile, int, unsigned) { atomicAdd(&lds_hist[tile], 1u); });
        __syncthreads();
    }
    __syncthreads();
    unsigned* row = hist + (size_t)blockIdx.x * tiles;
    for (int t = threadIdx.x; t < tiles; t += DGM_BIN_THREADS) row[t] = lds_hist[t];
}

// per tile: exclusive prefix down the chunk rows (in place) and the tile total
__global__ void __launch_bounds__(256)
colscan_kernel(int tiles, int nchunks, unsigned* __restrict__ hist, unsigned* __restrict__ tile_count) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= tiles) return;
    unsigned run = 0;
    int c = 0;
    for (; c + 8 <= nchunks; c += 8) {
        unsigned v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = hist[(size_t)(c + u) * tiles + t];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            hist[(size_t)(c + u) * tiles + t] = run;
            run += v[u];
        }
    }
    for (; c < nchunks; c++) {
        const unsigned v = hist[(size_t)c * tiles + t];
        hist[(size_t)c * tiles + t] = run;
        run += v;
    }
    tile_count[t] = run;
}

// ranges exactly as the reference leaves them: [start,end) for non-empty tiles, (0,0) otherwise
__global__ void __launch_bounds__(256)
write_ranges_kernel(int tiles, int small_cap, const unsigned* __restrict__ tile_count,
                    const unsigned* __restrict__ tile_offset, uint2* __restrict__ ranges,
                    unsigned* __restrict__ big_list, unsigned* __restrict__ big_count) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= tiles) return;
    const unsigned c = tile_count[t], o = tile_offset[t];
    ranges[t] = c ? make_uint2(o, o + c) : make_uint2(0u, 0u);
    if (c > (unsigned)small_cap) big_list[atomicAdd(big_count, 1u)] = (unsigned)t;  // order irrelevant
}

__global__ void __launch_bounds__(DGM_BIN_THREADS)
scatter_kernel(int P, int chunk, int tiles, int gridx, const unsigned* __restrict__ tiles_touched,
               const float* __restrict__ rec, const float* __restrict__ depth, const unsigned* __restrict__ hist,
               const unsigned* __restrict__ tile_offset, unsigned long long* __restrict__ keys) {
    extern __shared__ __attribute__((aligned(16))) unsigned cursor[];
    const unsigned* row = hist + (size_t)blockIdx.x * tiles;
    for (int t = threadIdx.x; t < tiles; t += DGM_BIN_THREADS) cursor[t] = tile_offset[t] + row[t];
    __syncthreads();
    const int g0 = blockIdx.x * chunk, g1 = min(P, g0 + chunk);
    for (int base = g0; base < g1; base += DGM_BIN_THREADS) {
        const int g = base + threadIdx.x;
        unsigned tt = 0u, rect = 0u, dbits = 0u;
        if (g < g1) {
            tt = tiles_touched[g];
            rect = __float_as_uint(rec[(size_t)g * DGM_REC_STRIDE + 9]);
            dbits = __float_as_uint(depth[g]);
        }
        unsigned long long m = __ballot(tt != 0u);
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const unsigned tt_i = __builtin_amdgcn_readlane(tt, src);
            const unsigned rect_i = __builtin_amdgcn_readlane(rect, src);
            const unsigned d_i = __builtin_amdgcn_readlane(dbits, src);
            const unsigned g_i = (unsigned)__builtin_amdgcn_readlane(g, src);
            unsigned xmin, ymin, w;
            unpack_rect(rect_i, xmin, ymin, w);
            const unsigned magic = w > 1 ? (0xFFFFFFFFu / w + 1u) : 0u;
            const unsigned long long key = ((unsigned long long)d_i << 32) | g_i;
            for (unsigned k = lane_id(); k < tt_i; k += 64) {
                const unsigned y = w > 1 ? __umulhi(k, magic) : k;
                const unsigned x = k - y * w;
                const unsigned tile = (ymin + y) * (unsigned)gridx + xmin + x;
                const unsigned slot = atomicAdd(&cursor[tile], 1u);
                keys[slot] = key;
            }
        }
    }
}

// ---- per-tile sort ------------------------------------------------------------------------------------------
// Normalised bitonic network (every compare-exchange moves the minimum to the lower index), so elements
// beyond n behave as +inf without being stored.  Pair p of a stage with half-block h: j = p mod h,
// lo = 2(p - j) + j ; flip stage partner = lo's block end mirrored, disperse stage partner = lo + h.
#define DGM_CEX(lo_, hi_)                                  \
    if ((hi_) < n) {                                       \
        const unsigned long long a_ = s[lo_], c_ = s[hi_]; \
        if (a_ > c_) {                                     \
            s[lo_] = c_;                                   \
            s[hi_] = a_;                                   \
        }                                                  \
    }

template <int THREADS, bool GLOBAL>
__device__ __forceinline__ void bitonic_sort(unsigned long long* s, int n) {
    int N = 1;
    while (N < n) N <<= 1;
    const int half = N >> 1;
    for (int k = 2; k <= N; k <<= 1) {
        const int hk = k >> 1;
        for (int p = threadIdx.x; p < half; p += THREADS) {
            const int j = p & (hk - 1);
            const int blk = (p - j) << 1;
            const int lo = blk + j, hi = blk + k - 1 - j;
            DGM_CEX(lo, hi)
        }
        if (GLOBAL) __threadfence_block();
        __syncthreads();
        for (int h = k >> 2; h >= 1; h >>= 1) {
            for (int p = threadIdx.x; p < half; p += THREADS) {
                const int j = p & (h - 1);
                const int lo = ((p - j) << 1) + j, hi = lo + h;
                DGM_CEX(lo, hi)
            }
            if (GLOBAL) __threadfence_block();
            __syncthreads();
        }
    }
}
#undef DGM_CEX

__device__ __forceinline__ void emit_sorted(const unsigned long long* s, int n, unsigned r0, int tile, int gridx,
                                            const float* __restrict__ rec, unsigned* __restrict__ point_list,
                                            unsigned* __restrict__ upos, int threads) {
    const unsigned tx = (unsigned)(tile % gridx), ty = (unsigned)(tile / gridx);
    for (int i = threadIdx.x; i < n; i += threads) {
        const unsigned g = (unsigned)s[i];
        point_list[r0 + i] = g;
        const float4 r2 = reinterpret_cast<const float4*>(rec + (size_t)g * DGM_REC_STRIDE)[2];
        unsigned xmin, ymin, w;
        unpack_rect(__float_as_uint(r2.y), xmin, ymin, w);
        const unsigned k = (ty - ymin) * w + (tx - xmin);
        upos[r0 + i] = __float_as_uint(r2.z) + k;
    }
}

// small segments (1..kSmallCap keys): one 256-thread workgroup per tile, 32 KB of LDS
__global__ void __launch_bounds__(256)
tile_sort_small_kernel(int cap, int gridx, const uint2* __restrict__ ranges,
                       const unsigned long long* __restrict__ keys, const float* __restrict__ rec,
                       unsigned* __restrict__ point_list, unsigned* __restrict__ upos) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
    const int tile = blockIdx.x;
    const uint2 r = ranges[tile];
    const int n = (int)(r.y - r.x);
    if (n < 1 || n > cap) return;
    for (int i = threadIdx.x; i < n; i += 256) skeys[i] = keys[r.x + i];
    __syncthreads();
    bitonic_sort<256, false>(skeys, n);
    emit_sorted(skeys, n, r.x, tile, gridx, rec, point_list, upos, 256);
}

// big segments come from a device-built worklist (write_ranges_kernel), walked by a FIXED grid so that no
// host read-back is needed and an empty list costs one trivial launch.  Up to `cap` keys are sorted in
// 128 KB of LDS; anything larger falls back to the same network on the global key array.
__global__ void __launch_bounds__(1024)
tile_sort_big_kernel(int cap, int gridx, const unsigned* __restrict__ big_list, const unsigned* __restrict__ big_count,
                     const uint2* __restrict__ ranges, unsigned long long* __restrict__ keys,
                     const float* __restrict__ rec, unsigned* __restrict__ point_list, unsigned* __restrict__ upos) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
    const unsigned count = *big_count;
    for (unsigned w = blockIdx.x; w < count; w += gridDim.x) {
        const int tile = (int)big_list[w];
        const uint2 r = ranges[tile];
        const int n = (int)(r.y - r.x);
        if (n <= cap) {
            for (int i = threadIdx.x; i < n; i += 1024) skeys[i] = keys[r.x + i];
            __syncthreads();
            bitonic_sort<1024, false>(skeys, n);
            emit_sorted(skeys, n, r.x, tile, gridx, rec, point_list, upos, 1024);
        } else {
            bitonic_sort<1024, true>(keys + r.x, n);
            emit_sorted(keys + r.x, n, r.x, tile, gridx, rec, point_list, upos, 1024);
        }
        __syncthreads();
    }
}

// ---- host launchers -----------------------------------------------------------------------------------------
static constexpr int kSmallCap = 4096;   // 32 KB of LDS keys, 256 threads (cfg2 centre tiles reach 2-3 k entries on some frames)
static constexpr int kLargeCap = 16384;  // 128 KB of LDS keys, 1024 threads

int binning_lds_limit_tiles() { return 36 * 1024; }  // 144 KB of u32 counters

void launch_scan_blocks(hipStream_t st, int n, const unsigned* in, unsigned* out, unsigned* total) {
    hipLaunchKernelGGL(scan_exclusive_kernel, dim3(1), dim3(1024), 0, st, n, in, out, total);
}

hipError_t launch_count(hipStream_t st, int P, int chunk, int nchunks, int tiles, int gridx,
                        const unsigned* tiles_touched, float* rec, const unsigned* block_offs, unsigned* offs,
                        unsigned* hist) {
    const size_t lds = (size_t)tiles * 4;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)count_tiles_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(count_tiles_kernel, dim3(nchunks), dim3(DGM_BIN_THREADS), lds, st, P, chunk, tiles, gridx,
                       tiles_touched, rec, block_offs, offs, hist);
    return hipSuccess;
}

void launch_tile_scan(hipStream_t st, int tiles, int nchunks, unsigned* hist, unsigned* tile_count,
                      unsigned* tile_offset, uint2* ranges, unsigned* big_list, unsigned* big_count) {
    hipLaunchKernelGGL(colscan_kernel, dim3((tiles + 255) / 256), dim3(256), 0, st, tiles, nchunks, hist, tile_count);
    hipLaunchKernelGGL(scan_exclusive_kernel, dim3(1), dim3(1024), 0, st, tiles, tile_count, tile_offset,
                       tile_offset + tiles);
    hipLaunchKernelGGL(write_ranges_kernel, dim3((tiles + 255) / 256), dim3(256), 0, st, tiles, kSmallCap, tile_count,
                       tile_offset, ranges, big_list, big_count);
}

hipError_t launch_scatter(hipStream_t st, int P, int chunk, int nchunks, int tiles, int gridx,
                          const unsigned* tiles_touched, const float* rec, const float* depth, const unsigned* hist,
                          const unsigned* tile_offset, unsigned long long* keys) {
    const size_t lds = (size_t)tiles * 4;
    if (lds > 48 * 1024) {
        hipError_t e =
            hipFuncSetAttribute((const void*)scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(scatter_kernel, dim3(nchunks), dim3(DGM_BIN_THREADS), lds, st, P, chunk, tiles, gridx,
                       tiles_touched, rec, depth, hist, tile_offset, keys);
    return hipSuccess;
}

hipError_t launch_tile_sort(hipStream_t st, int tiles, int gridx, const uint2* ranges, unsigned long long* keys,
                            const float* rec, unsigned* point_list, unsigned* upos, const unsigned* big_list,
                            const unsigned* big_count) {
    static bool attr_set_dev[DGM_MAX_DEVICES] = {false};  // function attributes are per device
    bool& attr_set = attr_set_dev[current_device_slot()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)tile_sort_big_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLargeCap * 8);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(tile_sort_small_kernel, dim3(tiles), dim3(256), kSmallCap * 8, st, kSmallCap, gridx, ranges, keys,
                       rec, point_list, upos);
    hipLaunchKernelGGL(tile_sort_big_kernel, dim3(256), dim3(1024), kLargeCap * 8, st, kLargeCap, gridx, big_list,
                       big_count, ranges, keys, rec, point_list, upos);
    return hipSuccess;
}

}  // namespace dgm
