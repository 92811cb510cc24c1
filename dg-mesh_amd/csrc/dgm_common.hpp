// Shared host/device declarations of libdgmesh_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/dgmesh_hip.h"

#define DGM_TILE 16          // BLOCK_X = BLOCK_Y of the reference (config.h:16-17): keeps tile ids / ranges identical
#define DGM_REC_STRIDE 12    // floats per splat record (48 B = 3 x float4)
#ifndef DGM_SLAB_STRIDE
#define DGM_SLAB_STRIDE 9    // floats per per-instance gradient row (36 B, dword-aligned); 12 = padded to 48 B, 16-byte aligned
#endif
#define DGM_SHORT_LIST 4096  // tiles with at most this many list entries get 64-entry checkpoints / backward units
// ... and 32-entry ones on sparse frames: with fewer 64-entry units than the ~5 k waves the backward keeps resident, its duration is
// that of the unit with the most blended entries (render_bwd4.hip); finer units give the ticket scheduler something to balance.
// A function of R alone, so the forward, the backward and the binning-buffer layout agree without further state.
#define DGM_FINE_UNITS_BELOW (1u << 20)
// u32 words of the replay-unit control block behind `counters` (render_bwd4.hip): word 0 = full units listed, word 32 = last units
// listed -- each on its own 128-byte line: atomics on ONE line serialise (~88 per us), whatever the word
#define DGM_UCTL_WORDS 64
#define DGM_UCTL_LINE 32
#define DGM_PRE_BLOCK 256    // Gaussians per preprocess workgroup (also the granularity of block_sums)
#define DGM_BIN_PASS 512      // Gaussians a binning chunk workgroup takes per pass (chunks are multiples of it)
#define DGM_BIN_THREADS 1024  // its threads: a wave holds 32 Gaussians of the pass in its lower half (binning.hip: the wave walks its
                              // Gaussians one by one, a serial chain -- 32 steps on sixteen waves instead of 64 on eight)
#define DGM_MAX_CHUNKS 256   // chunk workgroups = rows of the per-chunk tile histogram
#define DGM_MAX_GRID_DIM 1023  // tiles per axis that fit the 10-bit rect packing

namespace dgm {

struct Layout : dgm_state_layout {};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int replay_unit_log2(size_t R) { return R < (size_t)DGM_FINE_UNITS_BELOW ? 5 : 6; }

// Pure function of (P, W, H, R): backward re-derives every pointer from the same three chunks.
static inline void compute_layout(int P, int W, int H, int R, dgm_state_layout* L) {
    const size_t A = 256;
    const int tx = (W + DGM_TILE - 1) / DGM_TILE, ty = (H + DGM_TILE - 1) / DGM_TILE;
    const size_t tiles = (size_t)tx * ty;
    const size_t Pz = (size_t)(P > 0 ? P : 0), Rz = (size_t)(R > 0 ? R : 0);
    const size_t nblk = (Pz + DGM_PRE_BLOCK - 1) / DGM_PRE_BLOCK;
    // chunk = contiguous run of Gaussians owned by one binning workgroup; multiple of DGM_BIN_PASS
    size_t chunk = (Pz + DGM_MAX_CHUNKS - 1) / DGM_MAX_CHUNKS;
    chunk = align_up(chunk ? chunk : 1, DGM_BIN_PASS);
    size_t nchunks = (Pz + chunk - 1) / chunk;
    if (nchunks == 0) nchunks = 1;
    L->tiles_x = tx;
    L->tiles_y = ty;
    L->n_chunks = (int)nchunks;
    L->chunk_size = (int)chunk;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o = align_up(o + bytes, A);
        return at;
    };
    L->rec = take(Pz * DGM_REC_STRIDE * 4);
    L->depth = take(Pz * 4);
    L->radii = take(Pz * 4);
    L->tiles_touched = take(Pz * 4);
    L->offs = take(Pz * 4);
    L->cov3D = take(Pz * 24);
    L->clamped = take(Pz);
    L->block_sums = take(nblk * 4);
    L->hist = take(nchunks * tiles * 4);
    L->tile_count = take(tiles * 4);
    L->tile_offset = take((tiles + 1) * 4);
    L->big_list = take(tiles * 4);
    L->counters = take((8 + DGM_UCTL_WORDS) * 4);  // (the replay-unit control block rides behind them: one memset clears both)
    L->geometry_bytes = o + A;
    o = 0;
    L->inst = take(Rz * 8);
    L->point_list = take(Rz * 4);
    L->slab = take(Rz * (DGM_SLAB_STRIDE * 4 > 16 ? DGM_SLAB_STRIDE * 4 : 16));  // (>= 16 B per entry: the tile sort's scratch)
    L->live = take(Rz + 8);  // (+8: flags are read eight at a time)
    L->ckpt = take((Rz / 256 + 1) * 256 * 16);  // per (tile, 256-entry round boundary): (T, C) of the tile's 256 pixels
    const int ulog = replay_unit_log2(Rz);
    L->ckpt64 = take(((Rz >> ulog) + 1) * 256 * 16);  // short tiles: per (tile, 64- or 32-entry boundary) instead
    L->ulist_full = take(((Rz >> ulog) + 1) * 16);  // the backward's work list: full replay units, a tile's run contiguous
    L->binning_bytes = o + A;
    o = 0;
    L->final_T = take((size_t)W * H * 4);
    L->n_contrib = take((size_t)W * H * 4);
    L->ranges = take(tiles * 8);
    L->nproc = take(tiles * 4);
    L->cfin = take(tiles * 256 * 16);  // final (T, C) per pixel, tile-major in the backward's lane order
    L->ulist_last = take((tiles + 1) * 16);  // ... and the tiles' last (shorter) units
    L->image_bytes = o + A;
}

// Per-device host caches (one process may drive several GPUs): index = current HIP device, clamped.
static constexpr int DGM_MAX_DEVICES = 16;
static inline int current_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev < DGM_MAX_DEVICES ? dev : DGM_MAX_DEVICES - 1;
}

static inline char* align_ptr(char* p, size_t a = 256) {
    return (char*)(((uintptr_t)p + a - 1) / a * a);
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__

// float -> int with the GPU conversion semantics the parity definition fixes (truncate toward
// zero, saturate, NaN -> 0) -- identical to oracle f2i_sat.
__device__ __forceinline__ int f2i_sat(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}
__device__ __forceinline__ unsigned f2u_sat(float f) {
    if (f != f) return 0u;
    if (f >= 4294967296.0f) return 4294967295u;
    if (f <= 0.0f) return 0u;
    return (unsigned)f;
}

// rect packing: xmin | ymin << 10 | width << 20  (all < 1024)
__device__ __forceinline__ unsigned pack_rect(unsigned xmin, unsigned ymin, unsigned w) {
    return xmin | (ymin << 10) | (w << 20);
}
__device__ __forceinline__ void unpack_rect(unsigned r, unsigned& xmin, unsigned& ymin, unsigned& w) {
    xmin = r & 1023u;
    ymin = (r >> 10) & 1023u;
    w = r >> 20;
}

// exp() of the blend loops.  v_exp_f32 on x*log2(e): relative error ~1e-6 on the range the blend
// loop uses (power in [-6, 0]); forward and backward share it so T replays consistently.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// 36-byte gradient rows are dword-aligned only: 16-byte accesses to them go through this type (gfx950 runs in unaligned access
// mode: a global_load/store_dwordx4 needs 4-byte alignment)
typedef float dgm_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned long long dgm_u64u __attribute__((aligned(1)));

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// wave-wide inclusive scan / sum through ds_bpermute-free shuffles (light kernels only)
// Wave-wide scans / reductions on the DPP data path (VALU rate): a Hillis-Steele scan inside each row of 16 lanes (row_shr 1, 2, 4,
// 8; lanes without a source add the identity), then lane 15 of rows 0 and 2 into rows 1 and 3 (row_bcast15), then lane 31 into the
// upper half (row_bcast31) -- six instructions.  (Rounds 1-5a: six dependent __shfl_up / __shfl_xor steps, each a ds_bpermute
// through the LDS crossbar -- ~700 cycles per scan, on the critical path of the tile sort's counter scan in every pass, the binning
// kernels' offsets and the backward's unit prologue.)  All 64 lanes must be active.
#define DGM_DPP_STEP(OP, v, ident, ctrl, rmask) OP(v, (unsigned)__builtin_amdgcn_update_dpp((int)(ident), (int)(v), ctrl, rmask, 0xf, false))
#define DGM_DPP_SCAN(OP, v, ident)                 \
    v = DGM_DPP_STEP(OP, v, ident, 0x111, 0xf); /* row_shr:1 */  \
    v = DGM_DPP_STEP(OP, v, ident, 0x112, 0xf); /* row_shr:2 */  \
    v = DGM_DPP_STEP(OP, v, ident, 0x114, 0xf); /* row_shr:4 */  \
    v = DGM_DPP_STEP(OP, v, ident, 0x118, 0xf); /* row_shr:8 */  \
    v = DGM_DPP_STEP(OP, v, ident, 0x142, 0xa); /* row_bcast:15 -> rows 1, 3 */ \
    v = DGM_DPP_STEP(OP, v, ident, 0x143, 0xc); /* row_bcast:31 -> rows 2, 3 */
// (row_bcast exists on gfx9-family hardware only, and these helpers return garbage from a partially active wave: a -DDGM_DEBUG build
// traps on both instead of computing on)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "dgm_common.hpp: the DPP wave scans use row_bcast15 / row_bcast31 (wave64, gfx9 family): gfx950 only"
#endif
#ifdef DGM_DEBUG
#define DGM_ASSERT_FULL_WAVE() do { if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap(); } while (0)
#else
#define DGM_ASSERT_FULL_WAVE() do { } while (0)
#endif
__device__ __forceinline__ unsigned dgm_op_add(unsigned a, unsigned b) { return a + b; }
__device__ __forceinline__ unsigned dgm_op_max(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned wave_inclusive_scan_u32(unsigned v) {
    DGM_ASSERT_FULL_WAVE();
    DGM_DPP_SCAN(dgm_op_add, v, 0u)
    return v;
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {  // (in every lane)
    DGM_ASSERT_FULL_WAVE();
    DGM_DPP_SCAN(dgm_op_add, v, 0u)
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {  // (in every lane)
    DGM_ASSERT_FULL_WAVE();
    DGM_DPP_SCAN(dgm_op_max, v, 0u)
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

#endif  // __HIPCC__
}  // namespace dgm
