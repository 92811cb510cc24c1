/*
 * dgmesh_hip.h -- C ABI of libdgmesh_hip.so, the MI355X (gfx950) implementation of the
 * DG-Mesh training hot path: differentiable 3D-Gaussian rasterizer, simple-knn, fused
 * deformation MLP.
 *
 * This header is the drop-in boundary.  Every entry point replaces one interface of the
 * reference (paths relative to /root/reference/dgmesh/submodules/):
 *
 *   dgm_rasterize_forward   <- CudaRasterizer::Rasterizer::forward
 *                              diff-gaussian-rasterization/cuda_rasterizer/rasterizer.h:35-59
 *                              (impl rasterizer_impl.cu:198-336), reached from Python through
 *                              RasterizeGaussiansCUDA, rasterize_points.cu:35-114
 *   dgm_rasterize_backward  <- CudaRasterizer::Rasterizer::backward, rasterizer.h:61-85
 *                              (impl rasterizer_impl.cu:340-434); RasterizeGaussiansBackwardCUDA,
 *                              rasterize_points.cu:117-196
 *   dgm_mark_visible        <- CudaRasterizer::Rasterizer::markVisible, rasterizer.h:28-33
 *                              (impl rasterizer_impl.cu:141-153); markVisible, rasterize_points.cu:198-217
 *   dgm_knn_mean_dist2      <- SimpleKNN::knn, simple-knn/simple_knn.h:16-19 (impl simple_knn.cu:185-221);
 *                              distCUDA2, simple-knn/spatial.cu:15-26
 *   dgm_mlp_*               <- DeformNetwork* / AppearanceNetwork forward+backward,
 *                              dgmesh/utils/time_utils.py:58-323 (plain nn.Linear there)
 *
 * Conventions (same as the reference unless stated):
 *   - all `const float*` / `float*` / `int*` arguments are DEVICE pointers; optional inputs are
 *     passed as NULL exactly where the reference receives data_ptr()==nullptr from a 0-element
 *     tensor (colors_precomp, cov3D_precomp, shs, scales, rotations);
 *   - matrices are the reference's transposed (row-vector) 4x4 fp32, i.e. column-major for the
 *     kernels (dgmesh/scene/cameras.py:60-71);
 *   - scratch memory is caller-owned: forward obtains three opaque byte buffers through
 *     allocator callbacks (the reference's std::function<char*(size_t)>), backward receives the
 *     same three pointers back.  Their layout is private to this library (dgm_describe_state
 *     exposes it for the parity tests only);
 *   - every function returns 0 on success, non-zero on error (message via dgm_last_error());
 *     the reference throws std::runtime_error / AT_ERROR at the same points;
 *   - `stream` is a hipStream_t (NULL = default stream).  Unlike the reference (legacy default
 *     stream everywhere) all work is enqueued on the caller's stream;
 *   - outputs are fully OVERWRITTEN (the reference accumulates into pre-zeroed tensors); callers
 *     need not zero-fill out_color, radii or any dL_d* array.
 */
#ifndef DGMESH_HIP_H
#define DGMESH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (a binding must check dgm_abi_version() against the header it was generated from):
 *   3: dgm_knn_mean_dist2 takes a caller-owned scratch buffer (dgm_knn_scratch_bytes); two replay-unit lists in dgm_state_layout.
 *   4: dgm_state_layout lost `upos` and `block_offs`, `inst` became uint2[R] (8-byte instance records; the backward forms the
 *      gradient row of an instance itself); dgm_laplace_* added.  Entry points' signatures are unchanged from 3.
 *   5: dgm_rasterize_forward_capacity added (a forward that never waits for the device); dgm_se3_* added (6-DoF heads);
 *      dgm_mlp_set_gemm knows modes 4 and 5.
 *      Every entry point of 4 is unchanged. */
#define DGM_ABI_VERSION 5

/* Allocator callback: must return a device pointer to at least `bytes` bytes (128-byte aligned),
 * valid until the matching backward has run.  Mirrors resizeFunctional, rasterize_points.cu:27-33. */
typedef char* (*dgm_alloc_fn)(void* ctx, size_t bytes);

int dgm_abi_version(void);
const char* dgm_last_error(void);

/* ---- rasterizer ------------------------------------------------------------------------- */

/* P Gaussians, D = active SH degree, M = SH coefficients per Gaussian (0 if shs == NULL).
 * out_color: (3,H,W) fp32.  radii: (P) int32 or NULL.  *num_rendered receives R (host int),
 * which costs one 4-byte device->host read-back exactly like rasterizer_impl.cu:281. */
int dgm_rasterize_forward(dgm_alloc_fn geom_alloc, void* geom_ctx, dgm_alloc_fn binning_alloc, void* binning_ctx,
                          dgm_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background,
                          int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                          const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                          const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                          const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                          int* radii, int debug, void* stream, int* num_rendered);

/* Gradient outputs (all fully written): dL_dmean2D (P,3), dL_dconic (P,4: slots x,y,w used),
 * dL_dopacity (P), dL_dcolor (P,3), dL_dmean3D (P,3), dL_dcov3D (P,6), dL_dsh (P,M,3) or NULL when
 * M == 0, dL_dscale (P,3), dL_drot (P,4). */
int dgm_rasterize_backward(int P, int D, int M, int R, const float* background, int width, int height,
                           const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                           float scale_modifier, const float* rotations, const float* cov3D_precomp,
                           const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                           float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer,
                           char* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                           float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                           float* dL_dscale, float* dL_drot, int debug, void* stream);

/* The same two calls for SH coefficients stored in two tensors, as the reference's model keeps them (_features_dc (P,1,3)
 * and _features_rest (P,M-1,3), gaussian_model_dpsr_dynamic_anchor.py:135-138): shs = the DC rows, shs_rest = the others;
 * the backward writes dL_dsh (P,1,3) and dL_dsh_rest (P,M-1,3).  Spares the caller torch.cat (get_features) and the
 * strided split of its gradient.  shs_rest == NULL (and dL_dsh_rest == NULL): exactly the calls above. */
int dgm_rasterize_forward_split_sh(dgm_alloc_fn geom_alloc, void* geom_ctx, dgm_alloc_fn binning_alloc, void* binning_ctx,
                                   dgm_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background,
                                   int width, int height, const float* means3D, const float* shs, const float* shs_rest,
                                   const float* colors_precomp, const float* opacities, const float* scales,
                                   float scale_modifier, const float* rotations, const float* cov3D_precomp,
                                   const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                                   float tan_fovy, int prefiltered, float* out_color, int* radii, int debug, void* stream,
                                   int* num_rendered);
/* The forward WITHOUT the host round trip (round 6).  Replaces -- and improves on -- the one blocking step of
 * CudaRasterizer::Rasterizer::forward, DGR/cuda_rasterizer/rasterizer_impl.cu:281 (`cudaMemcpy(&num_rendered, ...)` between the
 * scan and the allocation of the binning state): the caller sizes the binning buffer for `capacity` tile instances UP FRONT (e.g.
 * 1.25 x the previous frame's R), every array offset follows dgm_describe_state(P, width, height, capacity), and the call only
 * enqueues.  {R, flags, two worklist lengths} are copied to host_result[0..3] (caller-owned, page-locked host memory) behind the
 * scan; they are valid once an event the caller records after this call has completed.  flags bit 0: "point is filtered although
 * prefiltered is set" (auxiliary.h:156-160; the synchronous calls fail on it); bit 1: R > capacity -- the frame was neutralised on
 * the device (no tile has a range, no Gaussian a tile: image = background, the backward writes zero gradients, nothing is written
 * or read beyond `capacity` rows) and must be rendered again with capacity >= host_result[0].  The matching backward call takes
 * R = capacity.  Otherwise exactly dgm_rasterize_forward_split_sh (shs_rest may be NULL). */
int dgm_rasterize_forward_capacity(dgm_alloc_fn geom_alloc, void* geom_ctx, dgm_alloc_fn binning_alloc, void* binning_ctx,
                                   dgm_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background,
                                   int width, int height, const float* means3D, const float* shs, const float* shs_rest,
                                   const float* colors_precomp, const float* opacities, const float* scales,
                                   float scale_modifier, const float* rotations, const float* cov3D_precomp,
                                   const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                                   float tan_fovy, int prefiltered, float* out_color, int* radii, int debug, void* stream,
                                   int capacity, unsigned* host_result);
int dgm_rasterize_backward_split_sh(int P, int D, int M, int R, const float* background, int width, int height,
                                    const float* means3D, const float* shs, const float* shs_rest,
                                    const float* colors_precomp, const float* scales, float scale_modifier,
                                    const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                    const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                                    const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                                    const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                    float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dsh_rest,
                                    float* dL_dscale, float* dL_drot, int debug, void* stream);

/* present: (P) bytes, 1 = view-space z > 0.2 */
int dgm_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream);

/* Sizes of the three scratch buffers (what the allocator callbacks will be asked for). */
size_t dgm_geometry_bytes(int P, int width, int height);
size_t dgm_binning_bytes(int R);
size_t dgm_image_bytes(int width, int height);

/* Byte offsets of the arrays inside the scratch buffers -- TEST/INTROSPECTION ONLY. */
typedef struct {
    /* geometry buffer */
    size_t rec;           /* float[P][12]: x, y, conic a, b, c, opacity, r, g, b, rect(u32), offs(u32), 0 */
    size_t depth;         /* float[P]   view-space z */
    size_t radii;         /* int[P]     */
    size_t tiles_touched; /* u32[P]     */
    size_t offs;          /* u32[P]     EXCLUSIVE scan of tiles_touched (reference stores the inclusive scan) */
    size_t cov3D;         /* float[P][6] */
    size_t clamped;       /* u8[P]      bit ch set = colour channel ch clamped at 0 */
    size_t block_sums;    /* u32[ceil(P/256)] tiles touched per 256 Gaussians (preprocess -> the count kernel's offs) */
    size_t hist;          /* u32[n_chunks][tiles] */
    size_t tile_count;    /* u32[tiles]: per-tile list lengths during binning; after the forward call, the tiles in descending order of
                             their list length class (the order in which the forward blend takes them) */
    size_t tile_offset;   /* u32[tiles+1] */
    size_t big_list;      /* u32[tiles] worklists: tiles with more than 4096 instances from the front, tiles with 2049 ..
                             4096 from the end */
    size_t counters;      /* u32[8 + 64]: [0] = num_rendered, [1] = error flags, [2] = big worklist length, [3] = mid worklist length,
                             [4] = arrival counter of the tile scan; [8] = full replay units listed, [8 + 32] = last replay units
                             listed (the backward's work list, written by the forward; one 128-byte line each) */
    size_t geometry_bytes;
    /* binning buffer */
    size_t inst;       /* uint2[R] instance records (gaussian, depth bits), grouped by tile, in arrival order within a tile
                          (the tile sort's input) */
    size_t point_list; /* u32[R]  gaussian ids, by tile then depth then id: identical to the reference's */
    size_t slab;       /* float[R][9] per-instance gradient rows written by backward AT ROW offs[g] + k for Gaussian g's k-th tile
                          instance (k counts the tiles of g's rectangle row by row; rec[g][9] = packed rectangle, rec[g][10] =
                          offs[g]: the backward forms the row from the record it loads anyway -- ABI <= 3 carried it in a
                          u32[R] array `upos`).  Rows are therefore grouped by Gaussian and the per-Gaussian sum reads them
                          contiguously: colour r,g,b | moments of
                          g = G dL/dalpha about the splat centre: dx, dy, dx^2, dx dy, dy^2, 1.  Only rows with live[row] != 0
                          are written */
    size_t live;       /* u8[R] live[row] = 1 iff some pixel blended the instance, i.e. slab[row] was written by this backward */
    size_t ckpt;       /* float4[R/256 + 1][256]: (T, C.rgb) of a tile's pixels after each 256 list entries (forward ->
                          segment-parallel backward) */
    size_t ckpt64;     /* float4[R/u + 1][256], u = 64 (32 when R < 2^20: sparse frames): tiles with <= 4096 list entries leave
                          (T, C.rgb) after each u entries instead (slot (first + u i) / u), so that the backward replays them in
                          u-entry units on more waves */
    size_t ulist_full; /* uint4[R/u + 1]: the backward's work list, full u-entry units (256-entry ones for lists beyond 4096) in the
                          order the forward's tiles finished, a tile's run contiguous: (tile | short-list flag << 31, unit index,
                          first list slot, replay bound) */
    size_t binning_bytes;
    /* image buffer */
    size_t final_T;   /* float[H*W] */
    size_t n_contrib; /* u32[H*W] */
    size_t ranges;    /* uint2[tiles] */
    size_t nproc;     /* u32[tiles] list entries the backward has to replay (deepest contributor of the tile) */
    size_t cfin;      /* float4[tiles][256]: final (T, C.rgb without background) per pixel, backward lane order */
    size_t ulist_last; /* uint4[tiles + 1]: the tiles' last (partial) replay units, same record */
    size_t image_bytes;
    int tiles_x, tiles_y, n_chunks, chunk_size;
} dgm_state_layout;

int dgm_describe_state(int P, int width, int height, int R, dgm_state_layout* out);

/* Per-stage device timings, measured with hipEvents recorded on the caller's stream around each stage.
 * dgm_set_profiling(1): immediate -- every forward/backward call synchronises the stream and
 *     dgm_get_stage_ms() returns the durations of the most recent call;
 * dgm_set_profiling(2): deferred -- calls only record events (no synchronisation inside a timed region);
 *     after the caller has synchronised, dgm_collect_stage_ms() returns the average duration and the
 *     number of launches of every stage since the mode was set (and resets them);
 * dgm_set_profiling(0): off.  State is process-wide (PyTorch runs backward on its own thread).
 * dgm_set_profiling_sampling(n): in deferred mode bracket only every n-th launch of a stage (default 1); the averages are
 *     over the sampled launches, the counts are all launches. */
enum {
    DGM_STAGE_PREPROCESS = 0,
    DGM_STAGE_BIN_COUNT,
    DGM_STAGE_BIN_SCAN,
    DGM_STAGE_BIN_SCATTER,
    DGM_STAGE_TILE_SORT,
    DGM_STAGE_RENDER_FWD,
    DGM_STAGE_RENDER_BWD,
    DGM_STAGE_PREPROCESS_BWD,
    DGM_STAGE_MLP_LAYER_FWD, /* one 256 -> 256 trunk layer, forward GEMM launch (deferred mode only) */
    DGM_STAGE_MLP_LAYER_BWD, /* one 256 -> 256 backward-data GEMM launch */
    DGM_STAGE_MLP_LAYER_DW,  /* one 256 x 256 weight-gradient GEMM launch (without its reductions) */
    DGM_STAGE_MLP_BWD_PAIR,  /* backward data + weight gradient of one 256-wide layer in one launch (plane arithmetic) */
    DGM_STAGE_COUNT
};
void dgm_set_profiling(int mode);
void dgm_set_profiling_sampling(int every);
int dgm_get_stage_ms(float* ms, int capacity);
int dgm_collect_stage_ms(float* avg_ms, int* counts, int capacity);
const char* dgm_stage_name(int stage);

/* ---- simple-knn --------------------------------------------------------------------------- */

/* points: (P,3) fp32 device; mean_dists: (P) fp32 device = mean of the 3 smallest squared distances.
 * scratch: dgm_knn_scratch_bytes(P) bytes of caller-owned device memory, 256-byte aligned (Morton codes, the sort's
 * ping-pong buffers, the sorted points and the box table; contents are meaningless between calls).  Like every other entry
 * point the library neither allocates nor synchronises here (the reference's SimpleKNN::knn allocates five thrust vectors per
 * call, simple_knn.cu:187-210). */
size_t dgm_knn_scratch_bytes(int P);
int dgm_knn_mean_dist2(int P, const float* points, float* mean_dists, char* scratch, void* stream);

/* ---- deformation / appearance MLP trunk ---------------------------------------------------------- */

/* Weights in PyTorch nn.Linear layout (out_features, in_features), fp32, device pointers.
 * Geometry is the reference's only one: D = 8 layers, W = 256, skips = [4] (layer 5 consumes
 * [PE(x) | t_emb | h]), PE(x) = 63 columns, t_emb = t_dim columns (30 = timenet output for
 * is_blender, 21 = PE(t) otherwise).  Heads are passed concatenated: Wh (n_out, 256), bh (n_out). */
typedef struct {
    int n_layers;   /* 8 */
    int width;      /* 256 */
    int emb_dim;    /* 63 + t_dim */
    int t_dim;      /* 30 or 21 */
    int skip_layer; /* 5 */
    int n_out;      /* total head outputs, <= 16 */
    const float* W[8];
    const float* b[8];
    const float* Wh;
    const float* bh;
} dgm_mlp_params;

/* Arithmetic of the trunk GEMMs.  3 (default): "f16x3p" -- every fp32 operand scaled by a power of two and split into two binary16
 * numbers, three partial products accumulated in fp32 on the f16 matrix cores, on plane-format activations: every activation /
 * gradient tensor lives in HBM as its two binary16 planes with one power-of-two exponent per 32-row tile, split once by the kernel
 * that produces it.  1: native fp32 MFMA.  Both are fp32 GEMMs to rounding.  (0, "bf16x6", and 2, "f16x3" on fp32 rows, were
 * retired in rounds 4 / 5 and are ignored.)  Returns the previous mode; any other value only queries.  Process-wide; the initial
 * value comes from the environment variable DGM_MLP_GEMM=f16x3p|f32.  A forward and its backward must run in the same mode. */
int dgm_mlp_set_gemm(int mode);

/* Bytes of the workspace that forward fills (embedding, the 8 post-ReLU activations, re-laid-out
 * weights, scratch) and backward consumes; caller-owned, must stay alive between the two calls. */
size_t dgm_mlp_workspace_bytes(int N);

/* Introspection for tests and tools: byte offsets of the workspace's per-row tensors for N rows, in this order:
 * emb, Y[0..7], mask[0..7], Ga, Gb, Eexp, Yexp[0..7], Dexp, Gexp[0..1], Cin, Dp, partial[0..7], partial_db[0..7]
 * (DGM_MLP_WS_FIELDS entries).  In the plane arithmetic (mode 3) emb / Y / G rows are [h : K halves | l : K halves] with one
 * int exponent per 32-row tile in the matching *exp array: value = (h + l) * 2^-e.  Returns the number of fields. */
#define DGM_MLP_WS_FIELDS 49
int dgm_mlp_describe_workspace(int N, size_t* offs, int capacity);

/* out (N, n_out) = heads(trunk(x (N,3), temb)).  temb: (N, t_dim) with row stride temb_stride floats,
 * or ONE row broadcast to all N when temb_stride == 0 (t is identical for all rows in training). */
int dgm_mlp_forward(const dgm_mlp_params* p, int N, const float* x, const float* temb, int temb_stride,
                    char* workspace, float* out, void* stream);

/* Given dOut (N, n_out): dW[l] / db[l] in the layout of W[l] / b[l], dWh, dbh, and dtemb
 * ((t_dim) summed over rows when temb_stride == 0, else (N, t_dim)); dtemb may be NULL. */
int dgm_mlp_backward(const dgm_mlp_params* p, int N, const float* dOut, int temb_stride, char* workspace,
                     float* const* dW, float* const* db, float* dWh, float* dbh, float* dtemb, void* stream);
/* The same, plus dX (N, 3) = dL/dx through both uses of PE(x) (layer 0 and the skip layer) and the derivative of the positional
 * encoding; x is the forward pass's input.  For networks whose input carries a gradient -- the appearance network on mesh
 * vertices moved by deform_back (dgmesh/utils/renderer.py:179-181, dgmesh/utils/time_utils.py:269-323).  x and dX may both be
 * NULL (= dgm_mlp_backward).  Plane arithmetic (the default) with a broadcast time row only: otherwise an error is returned rather
 * than the gradient dropped. */
int dgm_mlp_backward_dx(const dgm_mlp_params* p, int N, const float* dOut, int temb_stride, char* workspace,
                        float* const* dW, float* const* db, float* dWh, float* dbh, float* dtemb, const float* x, float* dX,
                        void* stream);

/* The time branch of the is_blender networks for ONE row (t is identical for every Gaussian of an iteration,
 * dgmesh/train.py:158): out (n_out) = W2 relu(W1 PE(t) + b1) + b2 with PE(t) = [t, sin(2^k t), cos(2^k t), k < n_freq]
 * (dgmesh/utils/time_utils.py:24-55, 150-153: timenet = Linear(13, 256), ReLU, Linear(256, 30), n_freq = 6).
 * t: one device float; W1 (hidden, 2 n_freq + 1), W2 (n_out, hidden) row-major as nn.Linear stores them;
 * save: 2 n_freq + 1 + hidden floats kept for backward.  backward writes dW1, db1, dW2, db2 (t gets no gradient). */
int dgm_timenet_forward(const float* t, int n_freq, const float* W1, const float* b1, int hidden, const float* W2,
                        const float* b2, int n_out, float* save, float* out, void* stream);
int dgm_timenet_backward(const float* d_out, int n_freq, const float* W2, int hidden, int n_out, const float* save,
                         float* dW1, float* db1, float* dW2, float* db2, void* stream);

/* ---- fused image loss -------------------------------------------------------------------------------- */

/* loss = (1 - lambda) * mean|image - gt| + lambda * (1 - mean SSIM(image, gt)), the Gaussian-branch image loss of
 * dgmesh/train.py:307-311 (l1_loss + ssim of dgmesh/utils/loss_utils.py:18-19, 32-76; 11x11 window, sigma 1.5,
 * zero padding).  image, gt: (channels, H, W) fp32 device; out: 3 floats {loss, L1 term, mean SSIM};
 * the workspace filled by forward is consumed by backward.  gt receives no gradient. */
size_t dgm_image_loss_workspace_bytes(int channels, int H, int W);
int dgm_image_loss_forward(const float* image, const float* gt, int channels, int H, int W, float lambda_dssim,
                           char* workspace, float* out, void* stream);
/* grad_out: device scalar dL/dloss; d_image (channels, H, W) is fully written. */
int dgm_image_loss_backward(const float* image, const float* gt, int channels, int H, int W, float lambda_dssim,
                            const char* workspace, const float* grad_out, float* d_image, void* stream);

/* ---- per-Gaussian glue of the train step ------------------------------------------------------------------ */

/* Activations + deformation in front of the rasterizer (dgmesh/gaussian_renderer/__init__.py:77-95 with the accessors
 * of dgmesh/scene/gaussian_model_dpsr_dynamic_anchor.py:92-128):  means3D = xyz + d_xyz, scales = exp(scaling) +
 * d_scaling, rotations = normalize(rotation) + d_rotation, opacities = sigmoid(opacity).  delta: raw (P, ld) head
 * output of the deformation network, columns [d_xyz 0:3 | d_rotation 3:7 | d_scaling 7:10 | ...], ld >= 10.
 * backward: given the rasterizer's gradients writes the parameter gradients and d_delta (P, ld; columns >= 10 zero). */
int dgm_gaussian_apply_forward(int P, const float* xyz, const float* scaling, const float* rotation, const float* opacity,
                               const float* delta, int ld, float* means3D, float* scales, float* rotations,
                               float* opacities, void* stream);
int dgm_gaussian_apply_backward(int P, const float* scaling, const float* rotation, const float* opacity,
                                const float* g_means3D, const float* g_scales, const float* g_rotations,
                                const float* g_opacities, float* d_xyz, float* d_scaling, float* d_rotation,
                                float* d_opacity, float* d_delta, int ld, void* stream);

/* 6-DoF deformation heads (is_6dof=True; round 6).  dgm_se3_exp_*: rows wv[i * ld + 0..5] = the raw outputs (w, v) of branch_w /
 * branch_v -> T (N, 4, 4) row-major, exactly dgmesh/utils/time_utils.py:116-123 (theta = |w|, w / theta + 1e-5, v / theta + 1e-5)
 * followed by exp_se3 of dgmesh/utils/rigid_utils.py:60-83 (Rodrigues' formula :40-57, rp_to_se3 :23-37); backward: dT (N, 16) ->
 * d_wv[i * ldd + 0..5].  dgm_se3_transform_*: the 6-DoF branch of render() (dgmesh/gaussian_renderer/__init__.py:68-75):
 * out = (T [xyz, 1])[:3] / (T [xyz, 1])[3]; backward writes dT (N, 16) and d_xyz (N, 3) in full. */
int dgm_se3_exp_forward(int N, const float* wv, int ld, float* T, void* stream);
int dgm_se3_exp_backward(int N, const float* wv, int ld, const float* dT, float* d_wv, int ldd, void* stream);
int dgm_se3_transform_forward(int N, const float* T, const float* xyz, float* out, void* stream);
int dgm_se3_transform_backward(int N, const float* T, const float* xyz, const float* g_out, float* dT, float* d_xyz, void* stream);

/* Cycle-consistency loss of dgmesh/train.py:221-238 on the raw (N, ld) outputs a (deform) and b (deform_back):
 * out[0] = (mean|b_xyz + a_xyz| + mean|b_rot + a_rot| + mean|b_scale + a_scale|) / 3, out[1..3] the three terms.
 * Fixed-order two-level reduction.  backward: grad_out = device scalar; d_a, d_b (N, ld) fully written. */
size_t dgm_cycle_loss_workspace_bytes(int N);
int dgm_cycle_loss_forward(int N, const float* a, const float* b, int ld, char* workspace, float* out, void* stream);
int dgm_cycle_loss_backward(int N, const float* a, const float* b, int ld, const float* grad_out, float* d_a, float* d_b,
                            void* stream);

/* ---- optimizer ---------------------------------------------------------------------------------------- */

/* torch.optim.Adam (amsgrad=False, weight_decay=0) over n_tensors tensors in ONE kernel launch: the Adam steps that
 * close a train iteration (dgmesh/train.py:518-524; groups of gaussian_model_dpsr_dynamic_anchor.py:186-212 and
 * deform_model.py:33-44).  All arrays are HOST arrays of length n_tensors; params / grads / exp_avg / exp_avg_sq hold
 * fp32 device pointers, lr[i] the tensor's learning rate, step[i] >= 1 its step count AFTER this update (bias
 * correction 1 - beta^step).  Tensors with numel 0 are skipped. */
int dgm_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const long long* numel, const float* lr, const int* step, float beta1,
                  float beta2, float eps, void* stream);

/* ---- densification / pruning on the device (csrc/densify.hip) ----------------------------------------------------------
 * Replaces GaussianModelDPSRDynamicAnchor.densify_and_prune / prune_points
 * (dgmesh/scene/gaussian_model_dpsr_dynamic_anchor.py:383-551): decide keep / clone / split / prune per Gaussian, scan,
 * then gather every parameter and Adam moment into the final set with one multi-tensor launch.
 *   dgm_densify_decide: grad = grad_accum / denom (NaN -> 0); clone if grad >= thr and max(exp(scaling)) <= dense_extent,
 *     split if grad >= thr and larger; pruned if sigmoid(opacity) < min_opacity or max(exp(scaling)) > big_extent
 *     (pass +inf to switch the size limit off).  With keep_mask != NULL (bytes 0/1, device) the decision is the mask
 *     (prune_points).  After the stream reaches this point the three u32 at scratch + dgm_densify_totals_offset(P) hold
 *     the numbers of kept originals K, kept clones C and kept split parents S; new size = K + C + 2 S, ordered
 *     [originals | clones | first children | second children] like the reference's cat / prune sequence.
 *   dgm_densify_apply: out[t][j] = in[t][source(j)] for all n_tensors (<= 24) tensors of row width width[t] floats; rows of
 *     new points are zero where is_moment[t] != 0; the xyz / scaling rows of split children are rewritten from the
 *     caller's standard-normal samples z[2][P][3] (device).  src_scratch: (K + C + 2 S) * 4 bytes. */
/* Per-iteration statistics (dgmesh/train.py:489-496, gaussian_model_dpsr_dynamic_anchor.py:679-682), in place, for the
 * Gaussians with radii > 0: max_radii2D = max(max_radii2D, radii); with grad2d != NULL (dL/dmeans2D, (P,3)) also
 * grad_accum += |grad2d[:, :2]| and denom += 1.  All (P) fp32 on the device. */
int dgm_densify_stats(int P, const float* grad2d, const int* radii, float* max_radii2D, float* grad_accum, float* denom,
                      void* stream);
size_t dgm_densify_scratch_bytes(int P);
size_t dgm_densify_totals_offset(int P);
int dgm_densify_decide(int P, const float* grad_accum, const float* denom, const float* scaling, const float* opacity,
                       float grad_threshold, float dense_extent, float min_opacity, float big_extent,
                       const uint8_t* keep_mask, char* scratch, void* stream);
int dgm_densify_apply(int P, unsigned K, unsigned C, unsigned S, const char* scratch, unsigned* src_scratch, int n_tensors,
                      const float* const* in, float* const* out, const int* width, const int* is_moment, int xyz_index,
                      int scaling_index, int rotation_index, const float* z, void* stream);

/* ---- DPSR pieces (csrc/dpsr.hip) --------------------------------------------------------------------------------------
 * Replace point_rasterize / grid_interp (dgmesh/nvdiffrast_utils/dpsr_utils.py:69-198) and the spectral solve between the
 * two FFTs of DPSR.forward (dgmesh/nvdiffrast_utils/dpsr.py:28-69).  V: points in (0,1)^3 [n][3]; N: normals [n][3];
 * grids are res^3 (periodic), row-major [x][y][z]; spectra are the rfftn layout [res][res][res/2+1] of float2.
 *   splat_forward : grid[3][res^3] = sum_p w_c(p) N[p]  (the callee zeroes grid);  splat_backward: dV, dN from dgrid
 *   interp_forward: fv[p] = trilinear phi(V[p]);  interp_backward: dphi (zeroed by the callee, then accumulated), dV
 *   spectral      : adjoint == 0: out[k] = sum_d (-i c_d) in[d][k];  adjoint != 0: out[d][k] = (+i c_d) in[k];
 *                   c_d = omega_d G / (Lap + 1e-6), G = exp(-0.5 (2 sig |f| / res)^2), out(0) = 0. */
int dgm_dpsr_splat_forward(int n, int res, const float* V, const float* N, float* grid, void* stream);
int dgm_dpsr_splat_backward(int n, int res, const float* V, const float* N, const float* dgrid, float* dV, float* dN,
                            void* stream);
int dgm_dpsr_interp_forward(int n, int res, const float* phi, const float* V, float* fv, void* stream);
int dgm_dpsr_interp_backward(int n, int res, const float* phi, const float* V, const float* dfv, float* dphi, float* dV,
                             void* stream);
int dgm_dpsr_spectral(int res, float sig, const float* in, float* out, int adjoint, void* stream);

/* Umbrella-operator Laplacian regulariser of a triangle mesh, laplace_regularizer_const (dgmesh/nvdiffrast_utils/regularizer.py:40-60):
 * loss[0] = mean over the 3 V components of (term / max(norm, 1))^2, term[v] = sum over the faces at v of (a - v) + (b - v),
 * norm[v] = 2 x faces at v.  v_pos (V, 3) fp32, faces (F, 3) int32, scratch: dgm_laplace_scratch_floats(V) floats of caller-owned
 * device memory that the backward reads again (normalised term, norm).  dv (V, 3) = dloss[0] x d loss / d v_pos (overwritten).
 * Sums use fp32 atomics like the reference's scatter_add_: results agree to rounding, not bit for bit, run to run. */
size_t dgm_laplace_scratch_floats(int V);
int dgm_laplace_forward(int V, int F, const float* v_pos, const int* faces, float* scratch, float* loss, void* stream);
int dgm_laplace_backward(int V, int F, const int* faces, const float* scratch, const float* dloss, float* dv, void* stream);

/* ---- opacity field on a regular grid (csrc/opacity_field.hip) ------------------------------------------------------------
 * Replaces get_opacity_field_from_gaussians (dgmesh/utils/mesh_utils.py:7-76): occ[res^3] = sum over the Gaussians of the
 * cell's block (centre strictly inside the block's box grown by `margin`, opacity > opacity_threshold) of
 * opacity * exp(-0.5 d^T Sigma^-1 d), Sigma from (scalings, rotations) as build_covariance_from_scaling_rotation,
 * inverse by cofactors + 1e-24 (gaussian_3d_coeff).  coords[res]: the grid coordinates along an axis. */
size_t dgm_opacity_field_scratch_bytes(int P);
int dgm_opacity_field(int P, const float* xyz, const float* rotations, const float* scalings, const float* opacities,
                      float opacity_threshold, int res, int num_blocks, float margin, const float* coords, char* scratch,
                      float* occ, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DGMESH_HIP_H */
