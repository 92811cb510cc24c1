// extern "C" entry points of libdgmesh_hip.so (see include/dgmesh_hip.h for the contract and the reference
// interfaces each one replaces).  Host orchestration only: stream-ordered kernel launches, argument checks
// with the reference's error points (DGR/rasterize_points.cu:57-59, rasterizer_impl.cu:242-245), optional
// hipEvent stage timing.
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>

#include "dgm_common.hpp"

namespace dgm {
// preprocess.hip
void launch_preprocess_fwd(hipStream_t st, int P, int D, int M, const float* means3D, const float* scales,
                           float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                           const float* shs_rest, const float* cov3D_precomp, const float* colors_precomp,
                           const float* viewmatrix,
                           const float* projmatrix, const float* cam_pos, int W, int H, float tan_fovx, float tan_fovy,
                           int gridx, int gridy, int prefiltered, int* radii_out, float* rec, float* depth,
                           int* radii_int, unsigned* tiles_touched, float* cov3Ds, uint8_t* clamped,
                           unsigned* block_sums);
void launch_mark_visible(hipStream_t st, int P, const float* means3D, const float* viewmatrix, uint8_t* present);
// binning.hip
int binning_lds_limit_tiles();
void launch_scan_blocks(hipStream_t st, int n, const unsigned* in, unsigned* out, unsigned* total);
hipError_t launch_count(hipStream_t st, int P, int chunk, int nchunks, int tiles, int gridx,
                        const unsigned* tiles_touched, float* rec, const unsigned* block_sums, unsigned* offs,
                        unsigned* hist, unsigned* counters);
void launch_tile_scan(hipStream_t st, int tiles, int nchunks, unsigned* hist, unsigned* tile_count,
                      unsigned* tile_offset, uint2* ranges, unsigned* big_list, unsigned* big_count, unsigned* arrive,
                      unsigned* total, unsigned capacity, unsigned* tiles_touched, int P);
hipError_t launch_scatter(hipStream_t st, int P, int chunk, int nchunks, int tiles, int gridx, int gridy, size_t R,
                          const unsigned* tiles_touched, const float* rec, const float* depth, const unsigned* hist,
                          const unsigned* tile_offset, uint2* inst);
hipError_t launch_tile_sort(hipStream_t st, int tiles, const uint2* ranges, const uint2* inst, uint2* pairs, size_t R,
                            unsigned* point_list, const unsigned* big_list, const unsigned* big_count, unsigned n_big, unsigned n_mid);
// render.hip
void launch_render_fwd(hipStream_t st, int tiles, const uint2* ranges, const unsigned* point_list, int W, int H,
                       int gridx, const float* rec, const float* bg, float* out_color, float* final_T,
                       unsigned* n_contrib, float4* ckpt, float4* cfin, float4* ckpt64, unsigned* nproc, size_t R, unsigned* uctl,
                       uint4* ulist_full, uint4* ulist_last, uint8_t* live, const unsigned* tile_order);
void launch_render_bwd4(hipStream_t st, int tiles, size_t R, const uint2* ranges, const unsigned* point_list, int W, int H,
                        int gridx, const float* bg, const float* rec, const float4* cfin, const float4* ckpt,
                        const float4* ckpt64, const unsigned* n_contrib, const float* dL_dpix, float* slab,
                        uint8_t* live, const unsigned* uctl, const uint4* ulist_full, const uint4* ulist_last);
// preprocess_bwd.hip
void launch_preprocess_bwd(hipStream_t st, int P, int D, int M, int gridx, const float* means3D, const int* radii,
                           const float* shs, const float* shs_rest, const uint8_t* clamped, const float* scales,
                           const float* rotations,
                           float scale_modifier, const float* cov3Ds, const float* viewmatrix, const float* projmatrix,
                           const float* campos, float focal_x, float focal_y, float tan_fovx, float tan_fovy, int W, int H,
                           const float* rec, const unsigned* tiles_touched, const unsigned* offs, const float* slab,
                           const uint8_t* live, float* dL_dmean2D,
                           float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                           float* dL_dsh, float* dL_dsh_rest, float* dL_dscale, float* dL_drot);
// knn.hip
size_t knn_scratch_bytes(int P);
void launch_knn(hipStream_t st, int P, const float* pts, float* dists, char* scratch);
}  // namespace dgm

using namespace dgm;

namespace {
thread_local std::string g_err;
}
namespace dgm {
void set_last_error(const char* msg) { g_err = msg ? msg : ""; }
}  // namespace dgm

namespace {

// Profiling state is process-wide: PyTorch runs backward on its own autograd thread, and forward + backward of
// one step must land in the same table.
std::atomic<int> g_profile{0};  // 0 off, 1 immediate (syncs the stream every call), 2 deferred (no sync)
float g_stage_ms[DGM_STAGE_COUNT] = {0};
std::mutex g_prof_mu;

// deferred mode: event pairs are parked here and only read by dgm_collect_stage_ms() after the caller has
// synchronised, so the timed region contains no extra synchronisation.
struct EventPair {
    hipEvent_t a, b;
};
constexpr int kMaxDeferred = 4096;
EventPair g_deferred[DGM_STAGE_COUNT][kMaxDeferred];
int g_deferred_n[DGM_STAGE_COUNT] = {0};
int g_deferred_created[DGM_STAGE_COUNT] = {0};
// sampling: only every g_sample_every-th launch of a stage is bracketed (two hipEventRecord per bracket cost host time
// and a marker packet each; ~90 brackets per train step otherwise)
int g_sample_every = 1;
int g_calls[DGM_STAGE_COUNT] = {0};
bool g_open[DGM_STAGE_COUNT] = {false};

int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define DGM_HIP(call)                                                                                \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Records hipEvents around stages on the caller's stream when profiling is on.
struct StageTimer {
    hipStream_t st;
    int mode;
    hipEvent_t ev[2 * DGM_STAGE_COUNT];
    bool used[DGM_STAGE_COUNT];
    explicit StageTimer(hipStream_t s) : st(s), mode(g_profile) {
        memset(used, 0, sizeof(used));
        if (mode == 1)
            for (auto& e : ev) (void)hipEventCreate(&e);
    }
    void begin(int s) {
        if (mode == 1) {
            (void)hipEventRecord(ev[2 * s], st);
            used[s] = true;
        } else if (mode == 2) {
            std::lock_guard<std::mutex> lk(g_prof_mu);
            if ((g_calls[s]++ % g_sample_every) != 0 || g_deferred_n[s] >= kMaxDeferred) return;
            const int i = g_deferred_n[s];
            if (i >= g_deferred_created[s]) {
                (void)hipEventCreate(&g_deferred[s][i].a);
                (void)hipEventCreate(&g_deferred[s][i].b);
                g_deferred_created[s] = i + 1;
            }
            (void)hipEventRecord(g_deferred[s][i].a, st);
            used[s] = true;
        }
    }
    void end(int s) {
        if (mode == 1) {
            (void)hipEventRecord(ev[2 * s + 1], st);
        } else if (mode == 2 && used[s]) {
            std::lock_guard<std::mutex> lk(g_prof_mu);
            (void)hipEventRecord(g_deferred[s][g_deferred_n[s]].b, st);
            g_deferred_n[s]++;
            used[s] = false;
        }
    }
    void finish() {
        if (mode != 1) return;
        (void)hipStreamSynchronize(st);
        for (int s = 0; s < DGM_STAGE_COUNT; s++)
            if (used[s]) (void)hipEventElapsedTime(&g_stage_ms[s], ev[2 * s], ev[2 * s + 1]);
        for (auto& e : ev) (void)hipEventDestroy(e);
        mode = 0;
    }
    ~StageTimer() { finish(); }
};

}  // namespace

namespace dgm {
// deferred-mode stage brackets for translation units without a StageTimer (mlp.hip); no-ops unless mode == 2
void prof_begin(int s, hipStream_t st) {
    if (g_profile != 2) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_open[s] = false;
    if ((g_calls[s]++ % g_sample_every) != 0 || g_deferred_n[s] >= kMaxDeferred) return;
    g_open[s] = true;
    const int i = g_deferred_n[s];
    if (i >= g_deferred_created[s]) {
        (void)hipEventCreate(&g_deferred[s][i].a);
        (void)hipEventCreate(&g_deferred[s][i].b);
        g_deferred_created[s] = i + 1;
    }
    (void)hipEventRecord(g_deferred[s][i].a, st);
}
void prof_end(int s, hipStream_t st) {
    if (g_profile != 2) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_open[s]) return;
    g_open[s] = false;
    if (g_deferred_n[s] >= kMaxDeferred || g_deferred_n[s] >= g_deferred_created[s]) return;
    (void)hipEventRecord(g_deferred[s][g_deferred_n[s]].b, st);
    g_deferred_n[s]++;
}
}  // namespace dgm

namespace {

int check_launch(const char* what, bool debug, hipStream_t st) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: launch failed: %s", what, hipGetErrorString(e));
    if (debug) {  // CHECK_CUDA(.., debug): synchronise and throw (DGR/cuda_rasterizer/auxiliary.h:166-173)
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) return fail("%s: %s", what, hipGetErrorString(e));
    }
    return 0;
}
#define DGM_CHECK(what)                               \
    do {                                              \
        if (check_launch(what, debug != 0, st)) return 1; \
    } while (0)

}  // namespace

extern "C" {

int dgm_abi_version(void) { return DGM_ABI_VERSION; }
const char* dgm_last_error(void) { return g_err.c_str(); }

void dgm_set_profiling(int mode) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_profile = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
    if (g_profile == 2)
        for (int s = 0; s < DGM_STAGE_COUNT; s++) g_deferred_n[s] = 0, g_calls[s] = 0, g_open[s] = false;
}
void dgm_set_profiling_sampling(int every) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_sample_every = every < 1 ? 1 : every;
}
int dgm_collect_stage_ms(float* avg_ms, int* counts, int capacity) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = capacity < DGM_STAGE_COUNT ? capacity : DGM_STAGE_COUNT;
    for (int s = 0; s < n; s++) {
        double tot = 0;
        int ok = 0;
        for (int i = 0; i < g_deferred_n[s]; i++) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, g_deferred[s][i].a, g_deferred[s][i].b) == hipSuccess) {
                tot += ms;
                ok++;
            }
        }
        avg_ms[s] = ok ? (float)(tot / ok) : 0.f;
        if (counts) counts[s] = ok ? g_calls[s] : 0;  // launches since the mode was set (the average is over the sampled ones)
        g_deferred_n[s] = 0;
        g_calls[s] = 0;
    }
    return n;
}
int dgm_get_stage_ms(float* ms, int capacity) {
    int n = capacity < DGM_STAGE_COUNT ? capacity : DGM_STAGE_COUNT;
    for (int i = 0; i < n; i++) ms[i] = g_stage_ms[i];
    return n;
}
const char* dgm_stage_name(int s) {
    static const char* names[DGM_STAGE_COUNT] = {"preprocess_fwd", "bin_count",      "bin_scan",      "bin_scatter",
                                                 "tile_sort",      "render_fwd",     "render_bwd",    "preprocess_bwd",
                                                 "mlp_layer_fwd",  "mlp_layer_bwd",  "mlp_layer_dw",  "mlp_bwd_pair"};
    return (s >= 0 && s < DGM_STAGE_COUNT) ? names[s] : "?";
}

size_t dgm_geometry_bytes(int P, int width, int height) {
    dgm_state_layout L;
    compute_layout(P, width, height, 0, &L);
    return L.geometry_bytes;
}
size_t dgm_binning_bytes(int R) {
    dgm_state_layout L;
    compute_layout(0, 16, 16, R, &L);
    return L.binning_bytes;
}
size_t dgm_image_bytes(int width, int height) {
    dgm_state_layout L;
    compute_layout(0, width, height, 0, &L);
    return L.image_bytes;
}
int dgm_describe_state(int P, int width, int height, int R, dgm_state_layout* out) {
    if (!out) return fail("dgm_describe_state: out is NULL");
    compute_layout(P, width, height, R, out);
    return 0;
}

int dgm_rasterize_forward(dgm_alloc_fn geom_alloc, void* geom_ctx, dgm_alloc_fn binning_alloc, void* binning_ctx,
                          dgm_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background,
                          int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                          const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                          const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                          const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                          int* radii, int debug, void* stream, int* num_rendered) {
    return dgm_rasterize_forward_split_sh(geom_alloc, geom_ctx, binning_alloc, binning_ctx, image_alloc, image_ctx, P, D, M,
                                          background, width, height, means3D, shs, nullptr, colors_precomp, opacities, scales,
                                          scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx,
                                          tan_fovy, prefiltered, out_color, radii, debug, stream, num_rendered);
}

// capacity < 0: the reference's protocol -- R is read back (one blocking 16-byte copy) and the binning buffer sized with it.
// capacity >= 0 (dgm_rasterize_forward_capacity): the binning buffer is sized for `capacity` instances up front, nothing waits for
// the device; {R, flags, worklist lengths} are copied to `host_result` asynchronously and a frame with R > capacity is neutralised
// on the device (tile_scan_kernel) and flagged (flags bit 1).
static int forward_impl(dgm_alloc_fn geom_alloc, void* geom_ctx, dgm_alloc_fn binning_alloc, void* binning_ctx,
                        dgm_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background,
                        int width, int height, const float* means3D, const float* shs, const float* shs_rest,
                        const float* colors_precomp, const float* opacities, const float* scales,
                        float scale_modifier, const float* rotations, const float* cov3D_precomp,
                        const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                        float tan_fovy, int prefiltered, float* out_color, int* radii, int debug, void* stream,
                        int* num_rendered, const long long capacity, unsigned* host_result) {
    hipStream_t st = (hipStream_t)stream;
    if (num_rendered) *num_rendered = 0;
    if (shs_rest && (!shs || M < 2)) return fail("rasterize_forward: shs_rest needs the DC rows in shs and M >= 2");
    if (P < 0 || width <= 0 || height <= 0) return fail("rasterize_forward: bad sizes P=%d W=%d H=%d", P, width, height);
    if (!geom_alloc || !binning_alloc || !image_alloc) return fail("rasterize_forward: allocator callback is NULL");
    if (!background || !viewmatrix || !projmatrix || !cam_pos || !out_color)
        return fail("rasterize_forward: NULL required pointer");
    dgm_state_layout L;
    compute_layout(P, width, height, 0, &L);
    const int gridx = L.tiles_x, gridy = L.tiles_y, tiles = gridx * gridy;
    if (gridx > DGM_MAX_GRID_DIM || gridy > DGM_MAX_GRID_DIM)
        return fail("rasterize_forward: image %dx%d exceeds %d tiles per axis", width, height, DGM_MAX_GRID_DIM);
    if (tiles > binning_lds_limit_tiles())
        return fail("rasterize_forward: %d tiles exceed the LDS histogram capacity (%d)", tiles, binning_lds_limit_tiles());

    char* img = image_alloc(image_ctx, L.image_bytes);
    if (!img) return fail("rasterize_forward: image allocator returned NULL");
    img = align_ptr(img);
    float* final_T = (float*)(img + L.final_T);
    unsigned* n_contrib = (unsigned*)(img + L.n_contrib);
    uint2* ranges = (uint2*)(img + L.ranges);
    unsigned* nproc = (unsigned*)(img + L.nproc);
    float4* cfin = (float4*)(img + L.cfin);

    if (P == 0) {  // reference: kernels skipped, rendered = 0, out_color stays 0 (rasterize_points.cu:68,81)
        DGM_HIP(hipMemsetAsync(out_color, 0, (size_t)3 * width * height * sizeof(float), st));
        if (host_result) host_result[0] = host_result[1] = host_result[2] = host_result[3] = 0u;
        return 0;
    }
    if (!means3D || !opacities) return fail("rasterize_forward: NULL required pointer");
    if (!colors_precomp && !shs)  // reference: needs one of them (python wrapper raises, __init__.py:191-192)
        return fail("rasterize_forward: provide either SHs or precomputed colors");
    if (!cov3D_precomp && (!scales || !rotations))
        return fail("rasterize_forward: provide either scale/rotation pair or precomputed 3D covariance");
    if (shs && !colors_precomp && (D + 1) * (D + 1) > M)
        return fail("rasterize_forward: SH degree %d needs %d coefficients, got M=%d", D, (D + 1) * (D + 1), M);
    if (shs && !colors_precomp && D > 3) return fail("rasterize_forward: SH degree %d > 3 unsupported", D);
    if (rotations && (((uintptr_t)rotations) & 15)) return fail("rasterize_forward: rotations must be 16-byte aligned");

    char* geom = geom_alloc(geom_ctx, L.geometry_bytes);
    if (!geom) return fail("rasterize_forward: geometry allocator returned NULL");
    geom = align_ptr(geom);
    float* rec = (float*)(geom + L.rec);
    float* depth = (float*)(geom + L.depth);
    int* radii_int = (int*)(geom + L.radii);
    unsigned* tiles_touched = (unsigned*)(geom + L.tiles_touched);
    unsigned* offs = (unsigned*)(geom + L.offs);
    float* cov3D = (float*)(geom + L.cov3D);
    uint8_t* clamped = (uint8_t*)(geom + L.clamped);
    unsigned* block_sums = (unsigned*)(geom + L.block_sums);
    unsigned* hist = (unsigned*)(geom + L.hist);
    unsigned* tile_count = (unsigned*)(geom + L.tile_count);
    unsigned* tile_offset = (unsigned*)(geom + L.tile_offset);
    unsigned* big_list = (unsigned*)(geom + L.big_list);
    unsigned* counters = (unsigned*)(geom + L.counters);

    StageTimer tm(st);
    // (no memset of the counter words -- 8 + the replay units' control block: count_tiles_kernel's first workgroup clears them, and
    // nothing before it touches them)

    tm.begin(DGM_STAGE_PREPROCESS);
    launch_preprocess_fwd(st, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, shs_rest, cov3D_precomp,
                          colors_precomp, viewmatrix, projmatrix, cam_pos, width, height, tan_fovx, tan_fovy, gridx,
                          gridy, prefiltered, radii, rec, depth, radii_int, tiles_touched, cov3D, clamped, block_sums);
    DGM_CHECK("preprocess_fwd");
    tm.end(DGM_STAGE_PREPROCESS);

    // per-chunk tile histograms, then tile totals / starts / ranges / worklists and R = counters[0]: neither needs the binning
    // buffer, so both run before the host learns R
    tm.begin(DGM_STAGE_BIN_COUNT);
    DGM_HIP(launch_count(st, P, L.chunk_size, L.n_chunks, tiles, gridx, tiles_touched, rec, block_sums, offs, hist, counters));
    DGM_CHECK("count_tiles");
    tm.end(DGM_STAGE_BIN_COUNT);

    tm.begin(DGM_STAGE_BIN_SCAN);
    launch_tile_scan(st, tiles, L.n_chunks, hist, tile_count, tile_offset, ranges, big_list, counters + 2, counters + 4, counters,
                     capacity >= 0 ? (unsigned)capacity : 0xffffffffu, tiles_touched, P);
    DGM_CHECK("tile_scan");
    tm.end(DGM_STAGE_BIN_SCAN);

    int R;
    unsigned n_big, n_mid;
    if (capacity >= 0) {
        // no read-back on the critical path: the words go to the caller's (pinned) memory behind the scan, the caller looks at them
        // after an event of its own; every size below follows `capacity`, both sort worklists are launched (an empty one leaves at once)
        DGM_HIP(hipMemcpyAsync(host_result, counters, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        R = (int)capacity, n_big = n_mid = 0xffffffffu;
        if (num_rendered) *num_rendered = R;
    } else {

    // R is needed on the host to size the binning buffer (rasterizer_impl.cu:281 does the same read-back); the
    // "prefiltered but culled" flag of auxiliary.h:156-160 rides in the same 16-byte copy, so it is ALWAYS checked
    // (the reference traps the kernel unconditionally), not only with debug on.
    // (into PINNED host memory, one small buffer per calling thread: a copy to pageable memory is staged through the runtime's
    // own bounce buffer and costs the GPU a longer idle gap per frame.  Tried on top and not kept: the scan kernel mailing
    // {R, flags, sequence number} into host-mapped memory with the host polling it -- +0.6 % at cfg2, -3 % at the host-bound cfg1)
    static thread_local unsigned* pinned_words = nullptr;
    if (!pinned_words && hipHostMalloc((void**)&pinned_words, 64, hipHostMallocDefault) != hipSuccess) pinned_words = nullptr;
    unsigned stack_words[4] = {0, 0, 0, 0};
    unsigned* host_words = pinned_words ? pinned_words : stack_words;
    // (R, flags, and the lengths of the tile sort's "big" and "mid" worklists: empty ones are not launched)
    DGM_HIP(hipMemcpyAsync(host_words, counters, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    DGM_HIP(hipStreamSynchronize(st));
    const unsigned R_host = host_words[0];
    if (R_host > 0x7fffffffu) return fail("rasterize_forward: %u tile instances overflow int", R_host);
    R = (int)R_host, n_big = host_words[2], n_mid = host_words[3];
    if (num_rendered) *num_rendered = R;
    if (host_words[1] & 1u) return fail("Point is filtered although prefiltered is set. This shouldn't happen!");
    }

    compute_layout(P, width, height, R, &L);
    char* bin = binning_alloc(binning_ctx, L.binning_bytes);
    if (!bin) return fail("rasterize_forward: binning allocator returned NULL");
    bin = align_ptr(bin);
    uint2* inst = (uint2*)(bin + L.inst);
    unsigned* point_list = (unsigned*)(bin + L.point_list);
    float4* ckpt = (float4*)(bin + L.ckpt);
    float4* ckpt64 = (float4*)(bin + L.ckpt64);

    if (R > 0) {
        tm.begin(DGM_STAGE_BIN_SCATTER);
        DGM_HIP(launch_scatter(st, P, L.chunk_size, L.n_chunks, tiles, gridx, gridy, (size_t)R, tiles_touched, rec, depth, hist,
                               tile_offset, inst));
        DGM_CHECK("scatter");
        tm.end(DGM_STAGE_BIN_SCATTER);

        tm.begin(DGM_STAGE_TILE_SORT);
        // (segments beyond 4096 entries sort in global memory: their pair buffers are carved from the backward's row slab,
        // 36 bytes per entry -- they need 16 -- and idle during the forward pass)
        DGM_HIP(launch_tile_sort(st, tiles, ranges, inst, (uint2*)(bin + L.slab), (size_t)R, point_list, big_list, counters + 2,
                                 n_big, n_mid));
        DGM_CHECK("tile_sort");
        tm.end(DGM_STAGE_TILE_SORT);
    }

    tm.begin(DGM_STAGE_RENDER_FWD);
    launch_render_fwd(st, tiles, ranges, point_list, width, height, gridx, rec, background, out_color, final_T,
                      n_contrib, ckpt, cfin, ckpt64, nproc, (size_t)R, counters + 8, (uint4*)(bin + L.ulist_full),
                      (uint4*)(img + L.ulist_last), (uint8_t*)(bin + L.live), tile_count);
    DGM_CHECK("render_fwd");
    tm.end(DGM_STAGE_RENDER_FWD);
    tm.finish();
    return 0;
}

int dgm_rasterize_forward_split_sh(dgm_alloc_fn geom_alloc, void* geom_ctx, dgm_alloc_fn binning_alloc, void* binning_ctx,
                                   dgm_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background,
                                   int width, int height, const float* means3D, const float* shs, const float* shs_rest,
                                   const float* colors_precomp, const float* opacities, const float* scales,
                                   float scale_modifier, const float* rotations, const float* cov3D_precomp,
                                   const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                                   float tan_fovy, int prefiltered, float* out_color, int* radii, int debug, void* stream,
                                   int* num_rendered) {
    return forward_impl(geom_alloc, geom_ctx, binning_alloc, binning_ctx, image_alloc, image_ctx, P, D, M, background, width, height,
                        means3D, shs, shs_rest, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                        projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, radii, debug, stream, num_rendered, -1, nullptr);
}

int dgm_rasterize_forward_capacity(dgm_alloc_fn geom_alloc, void* geom_ctx, dgm_alloc_fn binning_alloc, void* binning_ctx,
                                   dgm_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background,
                                   int width, int height, const float* means3D, const float* shs, const float* shs_rest,
                                   const float* colors_precomp, const float* opacities, const float* scales,
                                   float scale_modifier, const float* rotations, const float* cov3D_precomp,
                                   const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                                   float tan_fovy, int prefiltered, float* out_color, int* radii, int debug, void* stream,
                                   int capacity, unsigned* host_result) {
    if (capacity < 1) return fail("rasterize_forward_capacity: capacity must be >= 1 tile instance");
    if (!host_result) return fail("rasterize_forward_capacity: host_result is NULL");
    return forward_impl(geom_alloc, geom_ctx, binning_alloc, binning_ctx, image_alloc, image_ctx, P, D, M, background, width, height,
                        means3D, shs, shs_rest, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                        projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, radii, debug, stream, nullptr, capacity,
                        host_result);
}

int dgm_rasterize_backward(int P, int D, int M, int R, const float* background, int width, int height,
                           const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                           float scale_modifier, const float* rotations, const float* cov3D_precomp,
                           const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                           float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer,
                           char* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                           float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                           float* dL_dscale, float* dL_drot, int debug, void* stream) {
    return dgm_rasterize_backward_split_sh(P, D, M, R, background, width, height, means3D, shs, nullptr, colors_precomp, scales,
                                           scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx,
                                           tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_dmean2D,
                                           dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, nullptr, dL_dscale,
                                           dL_drot, debug, stream);
}

int dgm_rasterize_backward_split_sh(int P, int D, int M, int R, const float* background, int width, int height,
                                    const float* means3D, const float* shs, const float* shs_rest,
                                    const float* colors_precomp, const float* scales, float scale_modifier,
                                    const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                    const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                                    const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                                    const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                    float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dsh_rest,
                                    float* dL_dscale, float* dL_drot, int debug, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (P <= 0) return 0;  // rasterize_points.cu:161
    if ((shs_rest != nullptr) != (dL_dsh_rest != nullptr) || (shs_rest && (!shs || !dL_dsh || M < 2)))
        return fail("rasterize_backward: shs_rest and dL_dsh_rest come together (with shs, dL_dsh and M >= 2)");
    if (width <= 0 || height <= 0 || R < 0) return fail("rasterize_backward: bad sizes");
    if (!geom_buffer || !image_buffer || (R > 0 && !binning_buffer)) return fail("rasterize_backward: NULL state buffer");
    if (!dL_dpix || !dL_dmean2D || !dL_dconic || !dL_dopacity || !dL_dcolor || !dL_dmean3D || !dL_dcov3D || !dL_dscale ||
        !dL_drot)
        return fail("rasterize_backward: NULL gradient pointer");
    if ((((uintptr_t)dL_dconic) & 15) || (((uintptr_t)dL_drot) & 15))
        return fail("rasterize_backward: dL_dconic / dL_drot must be 16-byte aligned");
    dgm_state_layout L;
    compute_layout(P, width, height, R, &L);
    const int gridx = L.tiles_x, tiles = L.tiles_x * L.tiles_y;
    char* geom = align_ptr(geom_buffer);
    char* bin = align_ptr(binning_buffer);
    char* img = align_ptr(image_buffer);
    const float* rec = (const float*)(geom + L.rec);
    const int* radii_int = (const int*)(geom + L.radii);
    const unsigned* tiles_touched = (const unsigned*)(geom + L.tiles_touched);
    const unsigned* offs = (const unsigned*)(geom + L.offs);
    const float* cov3D = (const float*)(geom + L.cov3D);
    const uint8_t* clamped = (const uint8_t*)(geom + L.clamped);
    const unsigned* point_list = (const unsigned*)(bin + L.point_list);
    float* slab = (float*)(bin + L.slab);
    uint8_t* live = (uint8_t*)(bin + L.live);
    const unsigned* n_contrib = (const unsigned*)(img + L.n_contrib);
    const uint2* ranges = (const uint2*)(img + L.ranges);
    const float4* cfin = (const float4*)(img + L.cfin);
    const float4* ckpt = (const float4*)(bin + L.ckpt);
    const float4* ckpt64 = (const float4*)(bin + L.ckpt64);
    if (!radii) radii = radii_int;  // rasterizer_impl.cu:375-378

    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);

    StageTimer tm(st);
    tm.begin(DGM_STAGE_RENDER_BWD);
    launch_render_bwd4(st, tiles, (size_t)R, ranges, point_list, width, height, gridx, background, rec, cfin, ckpt, ckpt64,
                       n_contrib, dL_dpix, slab, live, (const unsigned*)(geom + L.counters) + 8, (const uint4*)(bin + L.ulist_full),
                       (const uint4*)(img + L.ulist_last));
    DGM_CHECK("render_bwd");
    tm.end(DGM_STAGE_RENDER_BWD);

    tm.begin(DGM_STAGE_PREPROCESS_BWD);
    const float* cov3D_ptr = cov3D_precomp ? cov3D_precomp : cov3D;  // rasterizer_impl.cu:411
    // with precomputed colours the SH branch is skipped (backward.cu:390: `if (shs)`)
    launch_preprocess_bwd(st, P, D, M, gridx, means3D, radii, colors_precomp ? nullptr : shs,
                          colors_precomp ? nullptr : shs_rest, clamped, scales, rotations,
                          scale_modifier, cov3D_ptr, viewmatrix, projmatrix, campos, focal_x, focal_y, tan_fovx,
                          tan_fovy, width, height, rec, tiles_touched, offs, slab, live, dL_dmean2D, dL_dconic,
                          dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, colors_precomp ? nullptr : dL_dsh_rest, dL_dscale,
                          dL_drot);
    DGM_CHECK("preprocess_bwd");
    tm.end(DGM_STAGE_PREPROCESS_BWD);
    tm.finish();
    return 0;
}

int dgm_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream) {
    (void)projmatrix;
    if (P <= 0) return 0;
    if (!means3D || !viewmatrix || !present) return fail("mark_visible: NULL pointer");
    launch_mark_visible((hipStream_t)stream, P, means3D, viewmatrix, present);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("mark_visible: %s", hipGetErrorString(e));
    return 0;
}

size_t dgm_knn_scratch_bytes(int P) { return P > 0 ? knn_scratch_bytes(P) : 0; }

int dgm_knn_mean_dist2(int P, const float* points, float* mean_dists, char* scratch, void* stream) {
    if (P <= 0) return 0;
    if (!points || !mean_dists || !scratch) return fail("knn_mean_dist2: NULL pointer");
    if ((uintptr_t)scratch & 255) return fail("knn_mean_dist2: scratch must be 256-byte aligned");
    launch_knn((hipStream_t)stream, P, points, mean_dists, scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("knn_mean_dist2: %s", hipGetErrorString(e));
    return 0;
}

}  // extern "C"
