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
//                instance takes a slot with one LDS atomic and writes its 8-byte record (gaussian, depth bits) straight
//                into its tile segment;
//   4. sort    : one workgroup per tile sorts its segment -- an LSD radix sort on the depth bits with an index tie-break, pairs in
//                registers / LDS for segments up to 4096 entries, in global memory (L2) beyond that; the result does not
//                depend on the atomic arrival order -- and emits point_list.  (Rounds 2-4 also carried offs[g] + k, the instance's
//                row in the per-Gaussian order, through the sort into a `upos` array for the backward: 8 more bytes per record
//                in the scatter, 4 written here and 4 read by render_bwd4 -- which loads rec[g], where the rectangle and offs[g]
//                sit, anyway, and now forms the row itself: render_common.hpp instance_row.)
// Wave-cooperative rectangle expansion: a wave loads 64 Gaussians, then iterates over the lanes that own a
// non-empty rectangle (scalar bit loop on the ballot mask) and lets all 64 lanes cover that rectangle's
// tiles, so a Gaussian spanning thousands of tiles costs the same lane-cycles as many small ones.
#include <stdlib.h>

#include "dgm_common.hpp"

namespace dgm {

static constexpr int kRadixCap = 2048;  // longest tile segment the one-workgroup-per-tile radix sort takes (256 threads x 8)
static constexpr int kBigLds = 8192;    // longest segment whose (key, payload) pairs ping-pong in LDS (2 x 64 KB) in the big path

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
// exact k / w for k*w < 2^32 via a 32x32->hi multiply by ceil(2^32 / w)  (w >= 2; 0 stands for w <= 1: y = k).  Every lane divides
// for its OWN Gaussian, once, before the wave walks the Gaussians one by one: the walk is a serial chain per wave (64 steps,
// one wave per SIMD), and a scalar 32-bit division inside it was ~25 dependent instructions of every step.
__device__ __forceinline__ unsigned rect_magic(unsigned rect) {
    const unsigned w = rect >> 20;
    return w > 1 ? (0xFFFFFFFFu / w + 1u) : 0u;
}

template <typename F>
__device__ __forceinline__ void for_each_instance(int g, unsigned tt, unsigned rect, int gridx, F&& f) {
    unsigned long long m = __ballot(tt != 0u);
    const unsigned magic_l = rect_magic(rect);
    while (m) {
        const int src = __builtin_ctzll(m);
        m &= m - 1;
        const unsigned tt_i = __builtin_amdgcn_readlane(tt, src);
        const unsigned rect_i = __builtin_amdgcn_readlane(rect, src);
        const int g_i = __builtin_amdgcn_readlane(g, src);
        const unsigned magic = __builtin_amdgcn_readlane(magic_l, src);
        unsigned xmin, ymin, w;
        unpack_rect(rect_i, xmin, ymin, w);
        for (unsigned k = lane_id(); k < tt_i; k += 64) {
            const unsigned y = w > 1 ? __umulhi(k, magic) : k;
            const unsigned x = k - y * w;
            const unsigned tile = (ymin + y) * (unsigned)gridx + xmin + x;
            f(tile, g_i, k);
        }
    }
}

// count: also makes the exclusive scan over Gaussians (offs) from preprocess_fwd's per-256 block sums -- every workgroup adds up the
// sums in front of its chunk itself (a few hundred words), which retired the single-workgroup scan launch between the two kernels.
__global__ void __launch_bounds__(DGM_BIN_THREADS)
count_tiles_kernel(int P, int chunk, int tiles, int gridx, const unsigned* __restrict__ tiles_touched,
                   float* __restrict__ rec, const unsigned* __restrict__ block_sums, unsigned* __restrict__ offs,
                   unsigned* __restrict__ hist, unsigned* __restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_hist[];
    __shared__ unsigned wave_tot[DGM_BIN_THREADS / 64];
    for (int t = threadIdx.x; t < tiles; t += DGM_BIN_THREADS) lds_hist[t] = 0u;
    const int g0 = blockIdx.x * chunk, g1 = min(P, g0 + chunk);
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    unsigned carry = 0u;  // offs of the pass's first Gaussian
    if (blockIdx.x == 0) {
        // the forward call's counter words: everything from tile_scan_kernel on expects them zero, and [1] carries the "culled
        // although prefiltered" flag that preprocess_fwd leaves in the top bits of its block sums (c_api.hip reads it back with R)
        unsigned bad = 0u;
        for (int i = threadIdx.x; i < (P + DGM_PRE_BLOCK - 1) / DGM_PRE_BLOCK; i += DGM_BIN_THREADS) bad |= block_sums[i] >> 31;
        bad = __syncthreads_or((int)bad) ? 1u : 0u;
        if (threadIdx.x < 8 + DGM_UCTL_WORDS) counters[threadIdx.x] = threadIdx.x == 1 ? bad : 0u;
    }
    {
        unsigned part = 0u;
        for (int i = threadIdx.x; i < g0 / DGM_PRE_BLOCK; i += DGM_BIN_THREADS) part += block_sums[i] & 0x7fffffffu;  // (chunks: multiples)
        part = wave_inclusive_scan_u32(part);
        if (lane == 63) wave_tot[wv] = part;
        __syncthreads();  // (also: lds_hist is cleared)
#pragma unroll
        for (int w = 0; w < DGM_BIN_THREADS / 64; w++) carry += wave_tot[w];
        __syncthreads();
    }
    for (int base = g0; base < g1; base += DGM_BIN_PASS) {
        const int g = base + wv * 32 + (lane & 31);  // (32 Gaussians per wave, in its lower half)
        const bool mine = lane < 32 && g < g1;
        unsigned tt = 0u, rect = 0u;
        if (mine) {
            tt = tiles_touched[g];
            rect = __float_as_uint(rec[(size_t)g * DGM_REC_STRIDE + 9]);
        }
        const unsigned inc = wave_inclusive_scan_u32(tt);
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        unsigned pre = carry;
#pragma unroll
        for (int w = 0; w < DGM_BIN_THREADS / 64; w++) {
            pre += w < wv ? wave_tot[w] : 0u;
            carry += wave_tot[w];
        }
        if (mine) {
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

// Length class of a tile for the ordered hand-out (class 0 = longest): steps of 8 entries up to 1024, of 32 up to 5120, one class
// beyond -- the sparse frames' lists (tens to hundreds) and the dense frames' (one to a few thousand) both spread over many classes.
__device__ __forceinline__ unsigned ord_bucket(const unsigned len) {
    const unsigned k = len < 1024u ? len >> 3 : 128u + min((len - 1024u) >> 5, 127u);
    return 255u - k;
}

// ---- column prefixes, tile totals, tile starts, `ranges` and the sort's worklists in ONE launch ------------------------------
// (round 3: colscan + a single-workgroup scan + write_ranges, three launches and 22 us for 2 MB of histogram.)
// A workgroup owns 64 tiles; lane = tile, so every load of a chunk row is a 256-byte segment.  Its four waves take a quarter of
// the chunk rows each -- all of a wave's rows are loaded before any is used, <= 64 loads in flight per lane -- scan them in
// registers, exchange their totals through LDS and store prefix + offset.  The workgroup that arrives LAST at the device
// counter (release fence before it, acquire fence behind) scans the tile totals -- 256 threads, a contiguous run of tiles each --
// and writes tile starts, `ranges` exactly as the reference leaves them ([start, end) for non-empty tiles, (0, 0) otherwise),
// and the "big" / "mid" worklists of the tile sort.  `arrive` is one of the words the forward call clears.
__global__ void __launch_bounds__(256)
tile_scan_kernel(int tiles, int nchunks, int small_cap, unsigned* __restrict__ hist, unsigned* __restrict__ tile_count,
                 unsigned* __restrict__ tile_offset, uint2* __restrict__ ranges, unsigned* __restrict__ big_list,
                 unsigned* __restrict__ big_count, unsigned* __restrict__ arrive, unsigned* __restrict__ total,
                 const unsigned capacity, unsigned* __restrict__ tiles_touched, const int P) {
    constexpr int MAXC = DGM_MAX_CHUNKS / 4;  // chunk rows per wave
    __shared__ unsigned wtot[4][64];
    __shared__ unsigned wave_sum[4];
    __shared__ int last;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    {
        const int t = blockIdx.x * 64 + lane;
        const int per = (nchunks + 3) / 4, c0 = wv * per, c1 = min(nchunks, c0 + per);
        unsigned v[MAXC];
#pragma unroll
        for (int u = 0; u < MAXC; u++) v[u] = (t < tiles && c0 + u < c1) ? hist[(size_t)(c0 + u) * tiles + t] : 0u;
        unsigned run = 0;
#pragma unroll
        for (int u = 0; u < MAXC; u++) {
            const unsigned x = v[u];
            v[u] = run;
            run += x;
        }
        wtot[wv][lane] = run;
        __syncthreads();
        unsigned off = 0;
        for (int w = 0; w < wv; w++) off += wtot[w][lane];
        if (t < tiles) {
#pragma unroll
            for (int u = 0; u < MAXC; u++)
                if (c0 + u < c1) hist[(size_t)(c0 + u) * tiles + t] = v[u] + off;
            if (wv == 3) tile_count[t] = off + run;
        }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(arrive, 1u) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!last) return;
    __threadfence();
    // exclusive scan of the tile totals: thread i takes the K = 16 consecutive tiles base + 16 i .. (four 16-byte loads in flight
    // together, a scan in registers, ONE exchange of the thread totals), 4096 tiles a pass.  (First version: a contiguous run of
    // ceil(tiles / 256) tiles per thread -- a serial walk of 127 strided loads per thread at 4K-class images; second: 256 tiles a
    // pass with thread i on tile base + i -- ten dependent load -> scan -> barrier round trips at cfg2's 2 500 tiles, ~10 us of this
    // kernel's 20 with every other workgroup gone.)  The block of 16 that holds the last tile may reach past `tiles`: tile_count's
    // allocation is padded to the layout's alignment (>= 64 bytes), the values read there are masked.
    constexpr int K = 16;
    constexpr int ORD_MAX = 256 * K;  // tiles the length-ordered hand-out below covers (4096: a 1024 x 1024 image)
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    __shared__ unsigned ohist[256];
    __shared__ unsigned short oslot[ORD_MAX];
    ohist[threadIdx.x] = 0u;
    __syncthreads();
    unsigned run_total = 0;
    for (int base = 0; base < tiles; base += 256 * K) {
        const int t0 = base + (int)threadIdx.x * K;
        unsigned c[K];
#pragma unroll
        for (int q = 0; q < K / 4; q++) {
            u4v v = {0u, 0u, 0u, 0u};
            if (t0 + 4 * q < tiles) v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(tile_count + t0) + q);
            c[4 * q] = v.x, c[4 * q + 1] = v.y, c[4 * q + 2] = v.z, c[4 * q + 3] = v.w;
        }
        unsigned tot = 0u;
#pragma unroll
        for (int q = 0; q < K; q++) {
            if (t0 + q >= tiles) c[q] = 0u;
            tot += c[q];
        }
        const unsigned inc = wave_inclusive_scan_u32(tot);
        if (lane == 63) wave_sum[wv] = inc;
        __syncthreads();
        unsigned start = run_total + inc - tot;
        for (int w = 0; w < wv; w++) start += wave_sum[w];
#pragma unroll
        for (int q = 0; q < K; q++) {
            const int t = t0 + q;
            if (t < tiles) {
                tile_offset[t] = start;
                ranges[t] = c[q] ? make_uint2(start, start + c[q]) : make_uint2(0u, 0u);
                atomicAdd(&ohist[ord_bucket(c[q])], 1u);
                // worklists (order irrelevant): "big" grows from the front of big_list, "mid" (kRadixCap + 1 .. small_cap entries) from its end
                if (c[q] > (unsigned)small_cap) big_list[atomicAdd(big_count, 1u)] = (unsigned)t;
                else if (c[q] > (unsigned)kRadixCap) big_list[tiles - 1 - (int)atomicAdd(big_count + 1, 1u)] = (unsigned)t;
            }
            start += c[q];
        }
        run_total += wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        __syncthreads();  // (wave_sum is rewritten by the next pass)
    }
    if (threadIdx.x == 0) tile_offset[tiles] = run_total, *total = run_total;  // (R: the host reads it back to size the binning buffer)
    // Tiles in descending order of their list length (counting sort over 256 length classes, ord_bucket; the histogram was taken in
    // the loop above), left in tile_count -- which nothing reads after this kernel -- for the forward blend, which hands workgroup k
    // the k-th tile of this order.  A trained frame's long lists sit next to each other in raster order and its workgroups are all
    // resident at once, so the hardware cannot balance them: in this order workgroups k, k + #CUs, k + 2 #CUs ... -- what one CU
    // receives -- mix long and short lists (render_fwd_async_kernel 0.082 -> 0.069 ms on the trained-like scene).  A dense frame's
    // workgroups are handed out as others retire; longest first shortens the tail (render_fwd_kernel 0.217 -> 0.204 ms on the
    // initial scene).  Costs ~2 us here.  Images of more than ORD_MAX tiles get the identity.
    {
        const bool ordered = tiles <= ORD_MAX;  // (workgroup-uniform)
        __syncthreads();
        if (ordered) {
            if (threadIdx.x < 64) {  // exclusive scan of the 256 bins: four per lane
                unsigned c4[4], tot4 = 0u;
#pragma unroll
                for (int q = 0; q < 4; q++) c4[q] = ohist[4 * threadIdx.x + q], tot4 += c4[q];
                unsigned start = wave_inclusive_scan_u32(tot4) - tot4;
#pragma unroll
                for (int q = 0; q < 4; q++) ohist[4 * threadIdx.x + q] = start, start += c4[q];
            }
            __syncthreads();
            const int t0 = (int)threadIdx.x * K;  // (tiles <= ORD_MAX = 256 K: one pass, the loads independent)
            unsigned c[K];
#pragma unroll
            for (int q = 0; q < K / 4; q++) {
                u4v v = {0u, 0u, 0u, 0u};
                if (t0 + 4 * q < tiles) v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(tile_count + t0) + q);
                c[4 * q] = v.x, c[4 * q + 1] = v.y, c[4 * q + 2] = v.z, c[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int q = 0; q < K; q++)
                if (t0 + q < tiles) oslot[atomicAdd(&ohist[ord_bucket(c[q])], 1u)] = (unsigned short)(t0 + q);
            __syncthreads();
            for (int i = threadIdx.x; i < tiles; i += 256) tile_count[i] = oslot[i];
        } else {
            for (int i = threadIdx.x; i < tiles; i += 256) tile_count[i] = (unsigned)i;
        }
    }
    // Capacity mode (dgm_rasterize_forward_capacity: the binning buffer was sized BEFORE R was known): a frame that does not fit is
    // neutralised here, by the one workgroup that knows R -- every tile's range and both sort worklists emptied, every Gaussian's
    // tile count zeroed (the scatter and the backward's gather walk those), bit 1 of the flag word raised for the host, which
    // discards the frame and renders it again with a larger buffer.  Nothing downstream writes or reads past `capacity` rows.
    if (run_total > capacity) {  // (workgroup-uniform)
        __syncthreads();
        for (int t = threadIdx.x; t < tiles; t += 256) ranges[t] = make_uint2(0u, 0u);
        for (int g = threadIdx.x; g < P; g += 256) tiles_touched[g] = 0u;
        if (threadIdx.x == 0) big_count[0] = 0u, big_count[1] = 0u, total[1] |= 2u;
    }
}

// Workgroup b = 8 * chunk + x runs on XCD x = b % 8 and emits the instances of its chunk that fall on the x-th BAND of tile rows
// (dense frames; nbands = 8).
// Why bands: a record is 8 bytes and successive records of one 128-byte line arrive a good fraction of the kernel apart; with a
// workgroup per chunk writing to all tiles (rounds 1-5a) an XCD's L2 had ~60 k lines open at a time, 7.7 MB against its 4 MB, and
// every record left for memory as a partial sector of its own -- `profiles/r05_pmc_scatter.json`: 113 MB written for 24 MB of records.
// A band keeps the open lines of an L2 to (chunks) x (tiles of the band) and every line is completed in the one L2 that
// holds it.  Price: each chunk is walked by eight workgroups, each skipping the Gaussians whose rectangle misses its band (the
// skip is decided per lane before the walk and costs the walk nothing).
__global__ void __launch_bounds__(DGM_BIN_THREADS)
scatter_kernel(int P, int chunk, int nbands, int tiles, int gridx, int gridy, const unsigned* __restrict__ tiles_touched,
               const float* __restrict__ rec, const float* __restrict__ depth, const unsigned* __restrict__ hist,
               const unsigned* __restrict__ tile_offset, uint2* __restrict__ inst) {
    extern __shared__ __attribute__((aligned(16))) unsigned cursor[];  // the band's tiles only
    // (nbands = 8, or 1 on sparse frames -- R < 2^20 -- where the lines are few enough to stay in L2 anyway and eight walks per chunk
    // cost more than they save: 18.9 vs 23.0 us on the trained-like scene)
    const int chunk_id = (int)blockIdx.x / nbands, band = (int)blockIdx.x - chunk_id * nbands;
    const int rows_per = (gridy + nbands - 1) / nbands;
    const int ty0 = band * rows_per, ty1 = min(gridy, ty0 + rows_per);
    if (ty0 >= ty1) return;
    const int t0 = ty0 * gridx, nt = (ty1 - ty0) * gridx;
    const unsigned* row = hist + (size_t)chunk_id * tiles + t0;
    for (int t = threadIdx.x; t < nt; t += DGM_BIN_THREADS) cursor[t] = tile_offset[t0 + t] + row[t];
    __syncthreads();
    const int g0 = chunk_id * chunk, g1 = min(P, g0 + chunk);
    const int lane = lane_id();
    for (int base = g0; base < g1; base += DGM_BIN_PASS) {
        const int gw = base + (int)(threadIdx.x >> 6) * 32;  // this wave's 32 Gaussians, in its lower half (as in count_tiles_kernel)
        const int g = gw + (lane & 31);
        unsigned tt = 0u, rect = 0u, dbits = 0u;
        if (lane < 32 && g < g1) {
            tt = tiles_touched[g];
            rect = __float_as_uint(rec[(size_t)g * DGM_REC_STRIDE + 9]);
            dbits = __float_as_uint(depth[g]);
        }
        // the part of the rectangle inside the band: rows [ya, yb), n_l tiles; geo_l = width | xmin << 12 | (ya - ty0) << 22
        unsigned xmin, ymin, w;
        unpack_rect(rect, xmin, ymin, w);
        const unsigned magic_l = rect_magic(rect);
        const unsigned h = w > 1 ? __umulhi(tt, magic_l) : tt;  // tt = w * h
        const int ya = max((int)ymin, ty0), yb = min((int)(ymin + h), ty1);
        const unsigned n_l = (tt != 0u && yb > ya) ? (unsigned)(yb - ya) * w : 0u;
        const unsigned geo_l = w | (xmin << 12) | ((unsigned)(ya - ty0) << 22);
        unsigned long long m = __ballot(n_l != 0u);
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const unsigned n_i = __builtin_amdgcn_readlane(n_l, src);
            const unsigned geo = __builtin_amdgcn_readlane(geo_l, src);
            const unsigned d_i = __builtin_amdgcn_readlane(dbits, src);
            const unsigned magic = __builtin_amdgcn_readlane(magic_l, src);
            const unsigned g_i = (unsigned)(gw + src);
            const unsigned w_i = geo & 4095u, x_i = (geo >> 12) & 1023u, y_i = geo >> 22;
            for (unsigned k = lane; k < n_i; k += 64) {
                const unsigned y = w_i > 1 ? __umulhi(k, magic) : k;
                const unsigned x = k - y * w_i;
                const unsigned slot = atomicAdd(&cursor[(y_i + y) * (unsigned)gridx + x_i + x], 1u);
                inst[slot] = make_uint2(g_i, d_i);  // one 8-byte store: Gaussian and depth bits (the sort key: depth, then Gaussian)
            }
        }
    }
}

// ---- per-tile sort ------------------------------------------------------------------------------------------
// small segments (1..kSmallCap keys): one 256-thread workgroup per tile.  LSD radix sort on the 32 depth bits, 8 bits per
// pass, with the entry's position in the unsorted segment (< 4096) as payload:
//   * every thread keeps up to 16 (depth, position) pairs in registers; wave w owns the w-th quarter of the array, so
//     "wave, then batch, then lane" is the array order and a stable pass only needs ranks in that order;
//   * rank inside a 64-key batch: eight ballots give each lane the mask of lanes with its digit (mbcnt = rank, popcount =
//     group size); the first lane of a group advances the wave's digit counter in LDS and broadcasts the old value;
//   * one wave turns the WAVES x 256 counters into destinations (digit major, wave minor); pairs move through LDS;
//   * a pass whose digit is the same for every key of the tile is skipped (the exponent byte almost always is);
//   * equal depths must end in ascending Gaussian index (what the reference's stable sort of the index-ordered emission
//     gives): after the last pass entries with an equal neighbour are placed inside their run by counting smaller indices;
//     a segment with a long run of equal depths is sorted again with index passes in front of the depth passes.
// The Gaussian index is fetched from the instance records through the payload only once, at the end (one gather inside the
// tile's own segment), and the output is written in order.
// (WAVES, RS_MAXB pairs per thread) = (4, 8): 256 threads, segments up to 2048 entries, one workgroup per tile at 65 VGPRs
// (sixteen pairs per thread cost 248); (8, 8): 512 threads, up to 4096 entries, the tiles of the device-built "mid"
// worklist -- a long segment is spread over more waves instead of more registers.
// OR / AND over the wave (DPP, dgm_common.hpp), then lane 63 merges the pair into the two LDS words
__device__ __forceinline__ unsigned dgm_op_or(unsigned a, unsigned b) { return a | b; }
__device__ __forceinline__ unsigned dgm_op_and(unsigned a, unsigned b) { return a & b; }
__device__ __forceinline__ void wave_or_and_to(unsigned vo, unsigned va, unsigned* s_red) {
    DGM_DPP_SCAN(dgm_op_or, vo, 0u)
    DGM_DPP_SCAN(dgm_op_and, va, 0xFFFFFFFFu)
    if ((threadIdx.x & 63) == 63) {
        atomicOr(&s_red[0], vo);
        atomicAnd(&s_red[1], va);
    }
}

// bits in which the keys of the segment differ (workgroup-wide OR / AND through two LDS words)
template <int RS_MAXB>
__device__ __forceinline__ unsigned radix_varying_bits(const unsigned (&key)[RS_MAXB], int n, int nb, int w0, unsigned* s_red) {
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) s_red[0] = 0u, s_red[1] = 0xFFFFFFFFu;
    __syncthreads();
    unsigned vo = 0u, va = 0xFFFFFFFFu;
#pragma unroll
    for (int b = 0; b < RS_MAXB; b++)
        if (b < nb && w0 + b * 64 + lane < n) vo |= key[b], va &= key[b];
    wave_or_and_to(vo, va, s_red);  // (one LDS atomic pair per wave: 256 threads on two words serialise)
    __syncthreads();
    const unsigned v = s_red[0] ^ s_red[1];
    __syncthreads();  // (s_red is reused)
    return v;
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "binning.hip uses v_bitop3_b32 and wave64 DPP row_bcast: gfx950 only"
#endif
// Lanes of the wave that hold the same 8-bit digit as this one (among the `valid` lanes), as the two halves of a 64-bit mask: per bit
// one ballot and m &= (bit set ? ballot : ~ballot) = m & ~(ballot ^ x) with x = 0 / -1 the sign-extended bit -- ONE v_bitop3_b32 per
// half (truth table 0x90 over (m, ballot, x)), 4 instructions per bit; the select form (`bs ? bb : ~bb`) compiled to 9.
__device__ __forceinline__ void match_digit(unsigned d, bool valid, unsigned& m_lo, unsigned& m_hi) {
    const unsigned long long mv = __ballot(valid);
    m_lo = (unsigned)mv, m_hi = (unsigned)(mv >> 32);
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
        const int x = __builtin_amdgcn_sbfe((int)d, bit, 1);
        unsigned long long bb;  // = __ballot(x != 0), as the compare of x itself (the compiler shifts d again for its own: +1 per bit)
        // (volatile: the result depends on EXEC, which the compiler cannot see -- it must not be hoisted, sunk or merged across
        // divergent control flow)
        asm volatile("v_cmp_gt_i32_e64 %0, 0, %1" : "=s"(bb) : "v"(x));
        m_lo = __builtin_amdgcn_bitop3_b32(m_lo, (unsigned)bb, (unsigned)x, 0x90);
        m_hi = __builtin_amdgcn_bitop3_b32(m_hi, (unsigned)(bb >> 32), (unsigned)x, 0x90);
    }
}

// stable LSD passes over the bits set in `varying`, 8 bits per pass; on return sk[0 .. n) holds the (key, payload) pairs in
// order and key[] / pay[] the pairs at this thread's own array positions
template <int WAVES, int RS_MAXB>
__device__ __forceinline__ void radix_passes(unsigned (&key)[RS_MAXB], unsigned (&pay)[RS_MAXB], unsigned varying, int n, int nb,
                                             int w0, uint2* sk, unsigned (*cnt)[256]) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    bool moved = false;
#pragma unroll 1
    for (int shift = 0; shift < 32; shift += 8) {
        if (((varying >> shift) & 255u) == 0u) continue;  // workgroup-uniform
        moved = true;
        reinterpret_cast<uint4*>(&cnt[0][0])[tid] = make_uint4(0u, 0u, 0u, 0u);  // WAVES * 64 threads x 4 = WAVES x 256 counters
        __syncthreads();
        unsigned rank[RS_MAXB];
#pragma unroll
        for (int b = 0; b < RS_MAXB; b++) {
            rank[b] = 0u;
            if (b < nb) {
                const bool valid = w0 + b * 64 + lane < n;
                const unsigned d = (key[b] >> shift) & 255u;
                unsigned m_lo, m_hi;
                match_digit(d, valid, m_lo, m_hi);
                const unsigned rk = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
                // every lane of a digit group reads the wave's counter (one address: a broadcast), THEN the group's first lane
                // advances it -- a wave's LDS operations execute in order, so no lane sees the update (first version: only the
                // first lane read, and ds_bpermute handed its value round: a find-first-set pair and one more LDS operation a batch)
                // (tried: every lane of the group reads the counter -- one address, a broadcast -- and the first lane then advances it,
                // which needs neither the find-first-set pair nor the ds_bpermute: 47.5 -> 54.4 us at cfg2.  Sixty-four lanes on the
                // counters' banks cost more than the ~40 group leaders plus the permute.)
                unsigned old = 0u;
                if (valid && rk == 0u) {  // first lane of its digit group
                    old = cnt[wv][d];
                    cnt[wv][d] = old + (unsigned)(__popc(m_lo) + __popc(m_hi));
                }
                const int leader = valid ? (m_lo ? __builtin_ctz(m_lo) : 32 + __builtin_ctz(m_hi)) : lane;
                old = (unsigned)__shfl((int)old, leader, 64);
                rank[b] = old + rk;
            }
        }
        __syncthreads();
        if (wv == 0) {  // digit-major, wave-minor exclusive prefix over the WAVES x 256 counters: lane l owns digits 4 l .. 4 l + 3
            uint4 c[WAVES];
            uint4 t = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int w = 0; w < WAVES; w++) {
                c[w] = reinterpret_cast<const uint4*>(&cnt[w][0])[lane];
                t.x += c[w].x, t.y += c[w].y, t.z += c[w].z, t.w += c[w].w;
            }
            const unsigned tot = t.x + t.y + t.z + t.w;
            const unsigned base = wave_inclusive_scan_u32(tot) - tot;
            uint4 run = make_uint4(base, base + t.x, base + t.x + t.y, base + t.x + t.y + t.z);
#pragma unroll
            for (int w = 0; w < WAVES; w++) {
                reinterpret_cast<uint4*>(&cnt[w][0])[lane] = run;
                run.x += c[w].x, run.y += c[w].y, run.z += c[w].z, run.w += c[w].w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < RS_MAXB; b++) {
            if (b < nb && w0 + b * 64 + lane < n) {
                const unsigned d = (key[b] >> shift) & 255u;
                sk[cnt[wv][d] + rank[b]] = make_uint2(key[b], pay[b]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < RS_MAXB; b++) {
            if (b < nb) {
                const int i = w0 + b * 64 + lane;
                if (i < n) {
                    const uint2 e = sk[i];
                    key[b] = e.x, pay[b] = e.y;
                }
            }
        }
    }
    if (!moved) {  // every key equal (or a single one): the array order stays
#pragma unroll
        for (int b = 0; b < RS_MAXB; b++) {
            if (b < nb && w0 + b * 64 + lane < n) sk[w0 + b * 64 + lane] = make_uint2(key[b], pay[b]);
        }
        __syncthreads();
    }
}

static constexpr int kTieRun = 32;  // runs of equal depth up to this length are ordered in place, longer ones by index passes

template <int WAVES, int RS_MAXB>
__device__ __forceinline__ void radix_sort_tile(const uint2 r, const uint2* __restrict__ inst, unsigned* __restrict__ point_list, uint2* sk,
                                                unsigned (*cnt)[256], unsigned* s_red) {
    const int n = (int)(r.y - r.x);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nb = (n + WAVES * 64 - 1) / (WAVES * 64);  // batches of 64 per wave
    const int w0 = wv * nb * 64;    // first array index of this wave
    const uint2* seg = inst + r.x;  // (Gaussian, depth bits) in arrival order
    // Equal depths must come out by ascending Gaussian index.  Short runs (the normal case: none, or a pair) are ordered in
    // place below; if any run is longer than kTieRun -- a scene whose Gaussians share one view depth -- the segment is sorted
    // again from scratch with index passes in front of the depth passes (LSD: the later key is the more significant).
    // phase 0: by depth;  then, only after a long run:  phase 1: by index,  phase 2: by depth again (one copy of the passes).
    unsigned key[RS_MAXB], pay[RS_MAXB];
    bool resort = false;
#pragma unroll 1
    for (int phase = 0; phase < 3; phase++) {
#pragma unroll
        for (int b = 0; b < RS_MAXB; b++) {
            const int i = w0 + b * 64 + lane;
            const bool valid = b < nb && i < n;
            if (phase < 2) pay[b] = valid ? (unsigned)i : 0u;
            key[b] = 0xFFFFFFFFu;
            if (valid) {
                const uint2 e = seg[pay[b]];
                key[b] = phase == 1 ? e.x : e.y;
            }
        }
        const unsigned varying = radix_varying_bits<RS_MAXB>(key, n, nb, w0, s_red);
        radix_passes<WAVES, RS_MAXB>(key, pay, varying, n, nb, w0, sk, cnt);
        if (phase == 0) {
            bool long_run = false;
#pragma unroll
            for (int b = 0; b < RS_MAXB; b++) {
                const int i = w0 + b * 64 + lane;
                if (b < nb && i + kTieRun < n && sk[i + kTieRun].x == key[b]) long_run = true;
            }
            resort = __syncthreads_or(long_run) != 0;  // (also: every wave is past its reads of sk)
            if (!resort) break;
        } else {
            __syncthreads();
        }
    }
    // gather the Gaussian index and the per-Gaussian position through the payload; place the members of short runs
    unsigned gid[RS_MAXB], dst[RS_MAXB];
#pragma unroll
    for (int b = 0; b < RS_MAXB; b++) {
        gid[b] = dst[b] = 0u;
        if (b < nb) {
            const int i = w0 + b * 64 + lane;
            if (i < n) {
                gid[b] = seg[pay[b]].x;
                dst[b] = (unsigned)i;
                const bool tie = !resort && ((i > 0 && sk[i - 1].x == key[b]) || (i + 1 < n && sk[i + 1].x == key[b]));
                if (tie) {
                    int s0 = i, s1 = i + 1;
                    while (s0 > 0 && sk[s0 - 1].x == key[b]) s0--;
                    while (s1 < n && sk[s1].x == key[b]) s1++;
                    unsigned below = 0u;
                    for (int j = s0; j < s1; j++) below += seg[sk[j].y].x < gid[b] ? 1u : 0u;
                    dst[b] = (unsigned)s0 + below;
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < RS_MAXB; b++) {
        if (b < nb && w0 + b * 64 + lane < n) point_list[r.x + dst[b]] = gid[b];
    }
}

__global__ void __launch_bounds__(256)
tile_sort_radix_kernel(const uint2* __restrict__ ranges, const uint2* __restrict__ inst, unsigned* __restrict__ point_list) {
    __shared__ __attribute__((aligned(16))) uint2 sk[kRadixCap];
    __shared__ __attribute__((aligned(16))) unsigned cnt[4][256];
    __shared__ unsigned s_red[2];
    const uint2 r = ranges[blockIdx.x];
    const int n = (int)(r.y - r.x);
    if (n < 1 || n > kRadixCap) return;  // longer segments: the worklists
    radix_sort_tile<4, kRadixCap / 256>(r, inst, point_list, sk, cnt, s_red);
}

// segments of kRadixCap + 1 .. 4096 entries: the "mid" worklist (filled from the END of big_list by write_ranges_kernel), fixed grid
__global__ void __launch_bounds__(512)
tile_sort_radix_mid_kernel(int tiles, const unsigned* __restrict__ big_list, const unsigned* __restrict__ mid_count,
                           const uint2* __restrict__ ranges, const uint2* __restrict__ inst, unsigned* __restrict__ point_list) {
    __shared__ __attribute__((aligned(16))) uint2 sk[4096];
    __shared__ __attribute__((aligned(16))) unsigned cnt[8][256];
    __shared__ unsigned s_red[2];
    const unsigned count = *mid_count;
    for (unsigned w = blockIdx.x; w < count; w += gridDim.x) {
        radix_sort_tile<8, 8>(ranges[big_list[tiles - 1 - (int)w]], inst, point_list, sk, cnt, s_red);
        __syncthreads();
    }
}

// Segments of more than kSmallCap entries (any length; the device-built "big" worklist, walked by a FIXED grid so that no host
// read-back is needed and an empty list costs one trivial launch): the same LSD radix sort with the pairs in GLOBAL memory.
// A pass reads the segment twice -- count (per-wave digit histogram from the ballot groups), one-wave scan, scatter (the
// ballots again give the rank inside a batch, the wave's LDS counter the running destination) -- so nothing is kept per key
// and the length is unbounded; (key, payload) pairs ping-pong between two buffers: in LDS for segments up to kBigLds entries,
// else carved from the backward's row slab, which is idle during the forward pass (16 bytes per entry of its 48; all traffic
// stays inside the tile's own segment, i.e. in L2).
// Equal depths: as in the small kernel -- runs up to kTieRun are placed by Gaussian index while gathering (among thousands of
// entries a pair of equal depth bits is the rule), and only a longer run costs an index sort in front of a second depth sort.
template <int WAVES>
__device__ __forceinline__ void radix_pass_global(const uint2* __restrict__ src, uint2* __restrict__ dst, int n, int shift,
                                                  unsigned (*cnt)[256], unsigned* dstart = nullptr) {
    constexpr int GB = 8;  // batches whose loads are in flight together (the passes are latency-bound: L2 round trips)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nbw = (n + WAVES * 64 - 1) / (WAVES * 64);  // batches of 64 per wave
    const int w0 = wv * nbw * 64;
    reinterpret_cast<uint4*>(&cnt[0][0])[tid] = make_uint4(0u, 0u, 0u, 0u);  // WAVES * 64 threads x 4 = WAVES x 256 counters
    __syncthreads();
    for (int b0 = 0; b0 < nbw; b0 += GB) {  // count
        unsigned key[GB];
#pragma unroll
        for (int u = 0; u < GB; u++) {
            const int i = w0 + (b0 + u) * 64 + lane;
            key[u] = (b0 + u < nbw && i < n) ? src[i].x : 0u;
        }
#pragma unroll
        for (int u = 0; u < GB; u++) {
            const bool valid = b0 + u < nbw && w0 + (b0 + u) * 64 + lane < n;
            const unsigned d = (key[u] >> shift) & 255u;
            if (__ballot(valid) != 0ull) {
                unsigned m_lo, m_hi;
                match_digit(d, valid, m_lo, m_hi);
                const unsigned rk = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
                if (valid && rk == 0u) cnt[wv][d] += (unsigned)(__popc(m_lo) + __popc(m_hi));
            }
        }
    }
    __syncthreads();
    if (wv == 0) {  // digit-major, wave-minor exclusive prefix: lane l owns digits 4 l .. 4 l + 3
        uint4 c[WAVES];
        uint4 t = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
            c[w] = reinterpret_cast<const uint4*>(&cnt[w][0])[lane];
            t.x += c[w].x, t.y += c[w].y, t.z += c[w].z, t.w += c[w].w;
        }
        const unsigned tot = t.x + t.y + t.z + t.w;
        const unsigned base = wave_inclusive_scan_u32(tot) - tot;
        uint4 run = make_uint4(base, base + t.x, base + t.x + t.y, base + t.x + t.y + t.z);
        if (dstart != nullptr) {  // where each digit's bucket begins in dst (dstart[256] = n)
            reinterpret_cast<uint4*>(dstart)[lane] = run;
            if (lane == 63) dstart[256] = base + tot;
        }
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
            reinterpret_cast<uint4*>(&cnt[w][0])[lane] = run;
            run.x += c[w].x, run.y += c[w].y, run.z += c[w].z, run.w += c[w].w;
        }
    }
    __syncthreads();
    for (int b0 = 0; b0 < nbw; b0 += GB) {  // scatter: cnt[wv][d] is now the wave's running destination for digit d
        uint2 e[GB];
#pragma unroll
        for (int u = 0; u < GB; u++) {
            const int i = w0 + (b0 + u) * 64 + lane;
            e[u] = make_uint2(0u, 0u);
            if (b0 + u < nbw && i < n) e[u] = src[i];
        }
#pragma unroll
        for (int u = 0; u < GB; u++) {
            const bool valid = b0 + u < nbw && w0 + (b0 + u) * 64 + lane < n;
            const unsigned d = (e[u].x >> shift) & 255u;
            if (__ballot(valid) != 0ull) {
                unsigned m_lo, m_hi;
                match_digit(d, valid, m_lo, m_hi);
                const unsigned rk = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
                unsigned old = 0u;
                if (valid && rk == 0u) {
                    old = cnt[wv][d];
                    cnt[wv][d] = old + (unsigned)(__popc(m_lo) + __popc(m_hi));
                }
                const int leader = valid ? (m_lo ? __builtin_ctz(m_lo) : 32 + __builtin_ctz(m_hi)) : lane;
                old = (unsigned)__shfl((int)old, leader, 64);
                if (valid) dst[old + rk] = e[u];
            }
        }
    }
    __threadfence_block();  // the pairs cross waves through memory (one CU, one L1)
    __syncthreads();
}

// OR / AND over .x of n pairs -> bits that differ
template <int WAVES>
__device__ __forceinline__ unsigned varying_bits_global(const uint2* __restrict__ src, int n, unsigned* s_red) {
    if (threadIdx.x == 0) s_red[0] = 0u, s_red[1] = 0xFFFFFFFFu;
    __syncthreads();
    unsigned vo = 0u, va = 0xFFFFFFFFu;
#pragma unroll 4
    for (int i = threadIdx.x; i < n; i += WAVES * 64) vo |= src[i].x, va &= src[i].x;
    wave_or_and_to(vo, va, s_red);
    __syncthreads();
    const unsigned v = s_red[0] ^ s_red[1];
    __syncthreads();
    return v;
}

// Segments that do not fit the LDS pair buffers: ONE pass through global memory -- a stable partition by the most significant
// varying byte of the depth -- then consecutive buckets are taken through LDS in groups of up to kBigLds entries (copy in, the
// remaining passes in LDS, gather + tie placement straight into point_list).  Equal keys share a bucket, so runs never
// straddle groups.  Returns false (nothing final written) for what this shape does not cover -- one depth for the whole segment, a
// bucket larger than the LDS buffers, a run of equal depths longer than kTieRun -- and the caller's all-global passes take over.
template <int WAVES>
__device__ __forceinline__ bool msd_first_sort(int n, unsigned rx, const uint2* __restrict__ seg, uint2* __restrict__ gA,
                                               uint2* __restrict__ gB, uint2* lds, unsigned (*cnt)[256], unsigned* s_red,
                                               unsigned* dstart, unsigned* __restrict__ point_list) {
    const int tid = threadIdx.x;
#pragma unroll 4
    for (int i = tid; i < n; i += WAVES * 64) gA[i] = make_uint2(seg[i].y, (unsigned)i);
    __threadfence_block();
    __syncthreads();
    const unsigned varying = varying_bits_global<WAVES>(gA, n, s_red);
    if (varying == 0u) return false;
    const int tb = (31 - __clz((int)varying)) & ~7;
    radix_pass_global<WAVES>(gA, gB, n, tb, cnt, dstart);
    if (__syncthreads_or(tid < 256 && dstart[tid + 1] - dstart[tid] > (unsigned)kBigLds) != 0) return false;
    uint2* L0 = lds;
    uint2* L1 = lds + kBigLds;
    bool ok = true;
    for (int d0 = 0; d0 < 256;) {  // (every thread walks the same buckets: dstart is in LDS)
        const unsigned s0 = dstart[d0];
        int d1 = d0 + 1;
        while (d1 < 256 && dstart[d1 + 1] - s0 <= (unsigned)kBigLds) d1++;
        const int m = (int)(dstart[d1] - s0);
        if (m > 0) {
#pragma unroll 4
            for (int i = tid; i < m; i += WAVES * 64) L0[i] = gB[s0 + i];
            __syncthreads();
            uint2* cur = L0;
            uint2* oth = L1;
            for (int shift = 0; shift <= tb; shift += 8) {
                if (((varying >> shift) & 255u) == 0u) continue;
                if (shift == tb && d1 - d0 == 1) continue;  // a single bucket: its top byte is one value
                radix_pass_global<WAVES>(cur, oth, m, shift, cnt);
                uint2* t = cur;
                cur = oth;
                oth = t;
            }
            bool long_run = false;
#pragma unroll 4
            for (int i = tid; i + kTieRun < m; i += WAVES * 64) long_run = long_run | (cur[i].x == cur[i + kTieRun].x);
            if (__syncthreads_or(long_run) != 0) {
                ok = false;
                break;
            }
#pragma unroll 2
            for (int i = tid; i < m; i += WAVES * 64) {
                const uint2 me = cur[i];
                const uint2 e = seg[me.y];
                unsigned at = (unsigned)i;
                if ((i > 0 && cur[i - 1].x == me.x) || (i + 1 < m && cur[i + 1].x == me.x)) {
                    int a0 = i, a1 = i + 1;
                    while (a0 > 0 && cur[a0 - 1].x == me.x) a0--;
                    while (a1 < m && cur[a1].x == me.x) a1++;
                    unsigned below = 0u;
                    for (int j = a0; j < a1; j++) below += seg[cur[j].y].x < e.x ? 1u : 0u;
                    at = (unsigned)a0 + below;
                }
                point_list[rx + s0 + at] = e.x;
            }
            __syncthreads();
        }
        d0 = d1;
    }
    return ok;
}

__global__ void __launch_bounds__(1024)
tile_sort_radix_big_kernel(const unsigned* __restrict__ big_list, const unsigned* __restrict__ big_count,
                           const uint2* __restrict__ ranges, const uint2* __restrict__ inst, uint2* __restrict__ pairs, size_t R,
                           unsigned* __restrict__ point_list) {
    constexpr int WAVES = 16;
    extern __shared__ __attribute__((aligned(16))) uint2 lds_pairs[];  // 2 x kBigLds pairs: segments up to kBigLds entries
    __shared__ __attribute__((aligned(16))) unsigned cnt[WAVES][256];  // ping-pong in LDS, longer ones in global memory
    __shared__ unsigned s_red[2];
    __shared__ __attribute__((aligned(16))) unsigned dstart[260];
    const unsigned count = *big_count;
    for (unsigned w = blockIdx.x; w < count; w += gridDim.x) {
        const uint2 r = ranges[big_list[w]];
        const int n = (int)(r.y - r.x);
        const uint2* seg = inst + r.x;
        uint2* bufA = n <= kBigLds ? lds_pairs : pairs + r.x;  // two pair buffers of this segment
        uint2* bufB = n <= kBigLds ? lds_pairs + kBigLds : pairs + R + r.x;
        if (n > kBigLds &&
            msd_first_sort<WAVES>(n, r.x, seg, bufA, bufB, lds_pairs, cnt, s_red, dstart, point_list)) {  // (uniform)
            __syncthreads();
            continue;
        }
        bool resort = false;
        for (int phase = 0; phase < 3; phase++) {  // 0: by depth; after a tie only: 1: by index, 2: by depth again
            uint2* cur = bufA;
            uint2* oth = bufB;
            if (phase == 2) {  // keys = depths of the index-sorted order (payloads kept); result of phase 1 is in `last`
                // (handled below: phase 1 leaves its result in bufA)
#pragma unroll 4
                for (int i = threadIdx.x; i < n; i += WAVES * 64) bufA[i].x = seg[bufA[i].y].y;
            } else {
#pragma unroll 4
                for (int i = threadIdx.x; i < n; i += WAVES * 64) {
                    const uint2 e = seg[i];
                    bufA[i] = make_uint2(phase == 1 ? e.x : e.y, (unsigned)i);
                }
            }
            __threadfence_block();
            __syncthreads();
            const unsigned varying = varying_bits_global<WAVES>(bufA, n, s_red);
            for (int shift = 0; shift < 32; shift += 8) {
                if (((varying >> shift) & 255u) == 0u) continue;  // workgroup-uniform
                radix_pass_global<WAVES>(cur, oth, n, shift, cnt);
                uint2* t = cur;
                cur = oth;
                oth = t;
            }
            if (cur != bufA) {  // keep the result in bufA (one copy at most per phase)
#pragma unroll 4
                for (int i = threadIdx.x; i < n; i += WAVES * 64) bufA[i] = bufB[i];
                __threadfence_block();
                __syncthreads();
            }
            if (phase == 0) {  // as in radix_sort_tile: only a run longer than kTieRun pays for the index passes
                bool long_run = false;
#pragma unroll 4
                for (int i = threadIdx.x; i + kTieRun < n; i += WAVES * 64) long_run = long_run | (bufA[i].x == bufA[i + kTieRun].x);
                resort = __syncthreads_or(long_run) != 0;
                if (!resort) break;
            }
        }
        // gather through the payload; the members of short runs of equal depth (with thousands of entries between two
        // depths a pair is the rule, not the exception) are placed by ascending Gaussian index here
#pragma unroll 2
        for (int i = threadIdx.x; i < n; i += WAVES * 64) {
            const uint2 me = bufA[i];
            const uint2 e = seg[me.y];
            unsigned at = (unsigned)i;
            if (!resort && ((i > 0 && bufA[i - 1].x == me.x) || (i + 1 < n && bufA[i + 1].x == me.x))) {
                int s0 = i, s1 = i + 1;
                while (s0 > 0 && bufA[s0 - 1].x == me.x) s0--;
                while (s1 < n && bufA[s1].x == me.x) s1++;
                unsigned below = 0u;
                for (int j = s0; j < s1; j++) below += seg[bufA[j].y].x < e.x ? 1u : 0u;
                at = (unsigned)s0 + below;
            }
            point_list[r.x + at] = e.x;
        }
        __syncthreads();
    }
}

// ---- host launchers -----------------------------------------------------------------------------------------
static constexpr int kSmallCap = 4096;   // 32 KB of LDS pairs, 256 threads x 16 (cfg2 centre tiles reach 2-3 k entries on some frames)

int binning_lds_limit_tiles() { return 36 * 1024; }  // 144 KB of u32 counters

void launch_scan_blocks(hipStream_t st, int n, const unsigned* in, unsigned* out, unsigned* total) {
    hipLaunchKernelGGL(scan_exclusive_kernel, dim3(1), dim3(1024), 0, st, n, in, out, total);
}

hipError_t launch_count(hipStream_t st, int P, int chunk, int nchunks, int tiles, int gridx,
                        const unsigned* tiles_touched, float* rec, const unsigned* block_sums, unsigned* offs,
                        unsigned* hist, unsigned* counters) {
    const size_t lds = (size_t)tiles * 4;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)count_tiles_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(count_tiles_kernel, dim3(nchunks), dim3(DGM_BIN_THREADS), lds, st, P, chunk, tiles, gridx,
                       tiles_touched, rec, block_sums, offs, hist, counters);
    return hipSuccess;
}

void launch_tile_scan(hipStream_t st, int tiles, int nchunks, unsigned* hist, unsigned* tile_count,
                      unsigned* tile_offset, uint2* ranges, unsigned* big_list, unsigned* big_count, unsigned* arrive,
                      unsigned* total, unsigned capacity, unsigned* tiles_touched, int P) {
    hipLaunchKernelGGL(tile_scan_kernel, dim3((tiles + 63) / 64), dim3(256), 0, st, tiles, nchunks, kSmallCap, hist, tile_count,
                       tile_offset, ranges, big_list, big_count, arrive, total, capacity, tiles_touched, P);
}

hipError_t launch_scatter(hipStream_t st, int P, int chunk, int nchunks, int tiles, int gridx, int gridy, size_t R,
                          const unsigned* tiles_touched, const float* rec, const float* depth, const unsigned* hist,
                          const unsigned* tile_offset, uint2* inst) {
    const int nbands = R < DGM_FINE_UNITS_BELOW ? 1 : 8;
    const size_t lds = (size_t)((gridy + nbands - 1) / nbands) * gridx * 4;  // cursors of one band of tile rows
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(scatter_kernel, dim3(nbands * nchunks), dim3(DGM_BIN_THREADS), lds, st, P, chunk, nbands, tiles, gridx, gridy,
                       tiles_touched, rec, depth, hist, tile_offset, inst);
    return hipSuccess;
}

hipError_t launch_tile_sort(hipStream_t st, int tiles, const uint2* ranges, const uint2* inst, uint2* pairs, size_t R,
                            unsigned* point_list, const unsigned* big_list, const unsigned* big_count, unsigned n_big, unsigned n_mid) {
    static_assert(kSmallCap == 8 * 64 * 8, "tile_sort_radix_mid_kernel covers segments up to kSmallCap");
    // n_big / n_mid: the lengths of the two worklists, which the host reads back together with R (tile_scan_kernel has built them
    // by then).  Rounds 2-4 launched all three classes on every frame with fixed grids -- an empty worklist cost a 5 us launch, and
    // both are empty on most frames of a trained scene.
    // (tried: the mid worklist and the one-tile-per-workgroup class in ONE launch of 512-thread workgroups, short segments on eight
    // waves with four pairs a thread, so that the mid class's 32 us would run beside the short class: 84 -> 97-100 us at cfg2 --
    // eight waves pay twice the counter scan and barrier population for a 2048-entry segment.  Round 5: the mid worklist inside the
    // big launch, sixteen waves x four pairs per tile -- 3.6 us less with a few dozen mid tiles, 3.4 us more with a few hundred, which
    // 256 workgroups of 145 KB LDS take in turns.  Also round 5: hipExtAnyOrderLaunch on the second and third launch -- packets without
    // the barrier bit, so that the classes, which sort disjoint tiles, run side by side.  The runtime ignores the flag on gfx9: the gap
    // between the scatter's end and render_fwd's start stayed at the sum of the three kernels, `tools/chain_wall.py`.  And: ONE launch
    // of 512-thread workgroups in which a short segment keeps waves 0-3 with the code below unchanged (the others leave at once) and a
    // long one all eight, its keys and payloads crossing the 16 KB exchange buffer one after the other -- 64.6 us against 31.7 + 24.9
    // late in the bench: 100 registers and eight wave slots per workgroup at dispatch thin out the short class.)
    if (n_mid > 0u)
        hipLaunchKernelGGL(tile_sort_radix_mid_kernel, dim3(n_mid < 768u ? n_mid : 768u), dim3(512), 0, st, tiles, big_list, big_count + 1,
                           ranges, inst, point_list);
    hipLaunchKernelGGL(tile_sort_radix_kernel, dim3(tiles), dim3(256), 0, st, ranges, inst, point_list);
    if (n_big > 0u) {
        static bool attr_set_dev[DGM_MAX_DEVICES] = {false};  // function attributes are per device
        bool& attr_set = attr_set_dev[current_device_slot()];
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)tile_sort_radix_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               2 * kBigLds * 8);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
        hipLaunchKernelGGL(tile_sort_radix_big_kernel, dim3(n_big < 256u ? n_big : 256u), dim3(1024), 2 * kBigLds * 8, st, big_list, big_count,
                           ranges, inst, pairs, R, point_list);
    }
    return hipSuccess;
}

}  // namespace dgm
