/*
 * hgs.h -- C ABI of libhgs.so, the MI355X (gfx950) implementation of the
 * hierarchical-3D-Gaussian rasterizer hot path.
 *
 * Plain C: raw device pointers, sizes and a hipStream_t passed as void*.  No
 * torch / C++ types cross this boundary.  Every entry point below names the
 * reference interface it stands in for (paths relative to the reference
 * checkout, graphdeco-inria/hierarchical-3d-gaussians):
 *
 *   diff_gaussian_rasterization._C   (submodules/hierarchy-rasterizer, .gitmodules:4-6;
 *                                     imported at gaussian_renderer/__init__.py:14,17)
 *   gaussian_hierarchy._C            (submodules/gaussianhierarchy, .gitmodules:10-12;
 *                                     imported at train_post.py:26, render_hierarchy.py:27,
 *                                     scene/gaussian_model.py:24)
 *   simple_knn._C                    (submodules/simple-knn, .gitmodules:7-9;
 *                                     imported at scene/gaussian_model.py:21)
 *
 * Conventions
 *   - return value 0 = success; anything else is an error code and
 *     hgs_last_error() (thread-local) holds the message;
 *   - every function that touches the GPU takes (stream, device) and calls
 *     hipSetDevice(device) first -- backward runs on an autograd worker thread;
 *   - all device buffers are caller-owned (torch's caching allocator in the
 *     Python host); the library keeps no pointer after a call returns;
 *   - matrices are the reference's stored (row-vector convention) [4,4]
 *     float32 tensors, i.e. column-major standard matrices
 *     (scene/cameras.py:95-97).
 */
#ifndef HGS_H
#define HGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HGS_ABI_VERSION 7
#define HGS_TILE 16
#define HGS_INST_GRAD_STRIDE 10 /* floats per (tile, Gaussian) instance in the backward scratch (40 bytes: the ten sums) */

enum {
  HGS_OK = 0,
  HGS_ERR_INVALID = 1, /* bad argument */
  HGS_ERR_HIP = 2,     /* a HIP runtime call or kernel failed */
  HGS_ERR_IO = 3,      /* file I/O */
  HGS_ERR_NOMEM = 4,
  HGS_ERR_CAPACITY = 5 /* hgs_raster_fwd: L exceeded the caller's capacity; redo with stage1/stage2 */
};

typedef void* hgs_stream_t; /* hipStream_t */

int hgs_abi_version(void);
const char* hgs_last_error(void);
/* number of HIP devices visible; <0 on error.  Used by the host to fail loudly. */
int hgs_device_count(void);

/* ---------------------------------------------------------------------------
 * Rasterizer.  Replaces diff_gaussian_rasterization._C.rasterize_gaussians /
 * rasterize_gaussians_backward as called through GaussianRasterizer.forward
 * (gaussian_renderer/__init__.py:64,105-113; :267-277; :339,381-389) with the
 * settings of GaussianRasterizationSettings (:44-62).
 * ------------------------------------------------------------------------- */
typedef struct hgs_raster_args {
  int32_t P;          /* Gaussians passed to the op */
  int32_t M;          /* SH coefficients stored per Gaussian ((max_sh_degree+1)^2), 0 with colors_precomp */
  int32_t sh_degree;  /* active degree 0..3 */
  int32_t width, height;
  float tanfovx, tanfovy;
  float scale_modifier;
  int32_t do_depth;   /* write the inverse-depth channel */
  int32_t debug;      /* synchronise + check after every launch */
  int32_t accumulate_grads; /* backward: add into the gradient buffers instead of overwriting them (accumulation over
                             * the views of one optimizer step); dL_dmeans2D and dL_dcolors are per-view quantities
                             * and are always overwritten */
  const float* bg;          /* device [3] */
  const float* viewmatrix;  /* device [16] */
  const float* projmatrix;  /* device [16] */
  const float* campos;      /* device [3] */
  const float* means3D;         /* [P,3] */
  const float* shs;             /* [P,M,3] or NULL */
  const float* colors_precomp;  /* [P,3]  or NULL (exactly one of shs / colors_precomp) */
  const float* opacities;       /* [P] */
  const float* scales;          /* [P,3] or NULL */
  const float* rotations;       /* [P,4] or NULL */
  const float* cov3D_precomp;   /* [P,6] or NULL (exactly one of scales+rotations / cov3D_precomp) */
  const float* interpolation_weights; /* [>=P] or NULL : hierarchy mode (gaussian_renderer/__init__.py:262) */
  const int32_t* num_node_kids;       /* [>=P] or NULL : hierarchy mode (gaussian_renderer/__init__.py:263) */
  /* Raw-parameter fast path: the op applies the activations of
   * scene/gaussian_model.py:108-128 itself, so the caller passes the optimiser's raw tensors and gets gradients
   * w.r.t. them -- no exp / normalize / sigmoid / cat kernels (and their backward) around the op. */
  const float* shs_rest;    /* NULL: shs is [P,M,3].  Else shs = features_dc [P,1,3], shs_rest = features_rest
                             * [P,M-1,3] (the two tensors get_features concatenates, scene/gaussian_model.py:121-124) */
  int32_t activations;      /* OR of HGS_ACT_*; 0 = inputs are already activated (the reference's call) */
  int32_t defer_sh_bwd;     /* backward: skip the SH part (dL_dshs and the view-direction term of dL_dmeans3D); the
                             * caller finishes it for several views at once with hgs_raster_sh_bwd_batched */
  /* ABI 7 (the eight bytes that held a pointer until ABI 4 and nothing since): how non-empty interpolation_weights /
   * num_node_kids are used.  0 (default): a per-GAUSSIAN remap of the opacity in the per-Gaussian kernel,
   *     o' = w o + (1 - w) (1 - (1 - min(o, 0.99))^(1/k))   for k >= 2;
   * 1: the same remap applied per PIXEL to alpha = o G inside the compositing kernels -- k coincident children at w = 0 then
   * composite exactly like their parent at every pixel, not only at the centre (which of the two the
   * reference's kernel implements is not recoverable from its checkout: gaussian_renderer/__init__.py:258-265 only
   * passes the tensors on).  Must hold the same value in the forward and the backward call. */
  int32_t lod_per_pixel;
  int32_t reserved1;
  /* The caller will run hgs_raster_bwd on the workspaces of this forward (must hold the SAME value in the forward and
   * in the backward call).  The forward's per-Gaussian kernel then also stores, next to the colour, the 3x3 Jacobian
   * d(rgb)/d(view direction) (36 bytes per Gaussian, in geom_ws): it has the SH coefficients in registers anyway, and
   * the backward's SH kernel no longer has to read them again (192 of its 396 bytes per Gaussian at M = 16).
   * 0: nothing extra is stored (inference), the backward recomputes from the coefficients. */
  int32_t prepare_backward;
  /* In-kernel LOD interpolation.  lod_render_indices != NULL: the attribute arrays (means3D,
   * scales, rotations, opacities, shs) hold ALL lod_rows hierarchy Gaussians and row i of the op is
   *     i <  lod_n :  w_i * attr[lod_render_indices[i]] + (1 - w_i) * attr[lod_parent_indices[i]],  w_i = interpolation_weights[i]
   *                   (every product and the sum rounded separately, as the torch expression of
   *                   gaussian_renderer/__init__.py:204-218 rounds; the parent quaternion flipped into the node's hemisphere)
   *     i >= lod_n :  attr[lod_rows - (P - lod_n) + (i - lod_n)]   -- the skybox tail (:220-234)
   * computed in registers by the per-Gaussian kernels of the forward AND of the backward: no interpolated row is ever
   * written to memory.  Needs shs + scales + rotations (no precomputed colours / covariances, no activations),
   * interpolation_weights / num_node_kids with >= P entries, prepare_backward = 1 for a differentiable call.
   * Backward, lod_scatter = 0: the gradients w.r.t. the INTERPOLATED rows ([P, ...]) are written; hgs_lod_gather_bwd
   * scatters them.  lod_scatter = 1 (needs 3M % 4 == 0): the backward's per-Gaussian kernels scatter themselves --
   * grads.dL_dmeans3D / dL_dscales / dL_drotations / dL_dopacity / dL_dshs then point at FULL arrays ([lod_rows, ...],
   * ZERO-FILLED by the caller) and receive w_i * g_i at the node row and the sum of (1 - w_i) * g_i over each run of
   * siblings at the parent row (the parent quaternion's hemisphere flip applied); no row gradient is ever written to
   * memory.  Runs are found among CONSECUTIVE rows (expand_to_size emits non-decreasing parents: every parent is then
   * written once, without atomics, bit-reproducibly); other orders are detected and fall back to atomic adds.
   * dL_dmeans2D stays per row ([P, 3]).  Precondition (as for hgs_lod_gather_bwd): render indices unique, no drawn row is
   * another entry's parent row -- true for any LOD cut. */
  int32_t lod_n;
  const int32_t* lod_render_indices;
  const int32_t* lod_parent_indices;
  int32_t lod_rows;
  int32_t lod_scatter;
} hgs_raster_args;

enum {
  HGS_ACT_SCALE_EXP = 1,       /* scales = exp(raw)                     scene/gaussian_model.py:110 */
  HGS_ACT_ROT_NORMALIZE = 2,   /* rotations = raw / max(|raw|, 1e-12)   scene/gaussian_model.py:114 */
  HGS_ACT_OPACITY_SIGMOID = 4, /* opacities = sigmoid(raw)              scene/gaussian_model.py:126-127 */
  HGS_ACT_OPACITY_ABS = 8      /* opacities = |raw| (hierarchy mode)    scene/gaussian_model.py:393 */
};
/* Activated values are DEFINED as the double-precision result rounded once to float32 (deterministic on every
 * platform); the discrete outputs (radii, tile rectangles, sort keys) follow from those float32 values. */

/* Workspace sizes in bytes.  geom: per-Gaussian state (P); bin: per-instance
 * keys/lists + tile ranges (L = number of (tile,Gaussian) instances, known after
 * stage 1); img: per-pixel state; bwd: backward scratch. */
int hgs_raster_ws_sizes(int32_t P, int32_t width, int32_t height, uint32_t L,
                        size_t* geom_bytes, size_t* bin_bytes, size_t* img_bytes, size_t* bwd_bytes);

/* Stage 1: per-Gaussian preprocess (cull, project, 3D->2D covariance, SH colour,
 * tile rectangle) + offsets scan.  Writes radii[P]; returns L on the host (one
 * 4-byte D2H copy + stream sync -- the only sync of the forward pass). */
int hgs_raster_fwd_stage1(const hgs_raster_args* a, void* geom_ws, int32_t* radii,
                          uint32_t* L_out_host, hgs_stream_t stream, int device);

/* Stage 2: key generation, radix sort by (tile|depth), tile ranges, tile
 * compositing.  out_color [3,H,W], out_invdepth [1,H,W] (may be NULL when
 * !do_depth). */
int hgs_raster_fwd_stage2(const hgs_raster_args* a, void* geom_ws, void* bin_ws, void* img_ws,
                          uint32_t L, float* out_color, float* out_invdepth,
                          hgs_stream_t stream, int device);

/* Single-call forward for callers that can bound L in advance (the Python host uses 1.25 x the previous
 * frame's count): bin_ws / bwd scratch are sized with L_cap, every kernel is enqueued before the host reads
 * L, so the GPU never idles between the stages (the two-stage path leaves a ~50 us bubble).  The call still
 * returns the exact L: the kernel that completes the scans of K1's workgroup sums (K3's last workgroup; the scan launch
 * on the fallback routes) stores it into a pinned, device-mapped host word and the host polls the event
 * recorded behind that kernel (no copy command in the stream; HGS_COUNT_BY_COPY / HGS_BLOCKING_WAIT in the environment
 * select a copy / a sleeping wait instead -- same results).  If L > L_cap it returns HGS_ERR_CAPACITY,
 * the outputs are invalid and the caller continues with hgs_raster_fwd_stage2 on an exactly sized bin_ws
 * (geom_ws / radii from this call stay valid).  Later calls (backward, views) must pass the same L_cap as L.
 * Library state: this entry point keeps, per (device, stream), 40 KB of zeroed device memory between calls (K1 adds its
 * workgroup sums there, a later kernel of the same call clears them; INTEGRATION.md "What the library keeps").  Not for
 * hipStreamPerThread (one handle, one stream per host thread) and not for a capturing stream: both take the scan launch.
 * hgs_release_device_state frees the blocks. */
int hgs_raster_fwd(const hgs_raster_args* a, void* geom_ws, void* bin_ws, void* img_ws, uint32_t L_cap,
                   int32_t* radii, float* out_color, float* out_invdepth, uint32_t* L_out_host,
                   hgs_stream_t stream, int device);

/* Frees what hgs_raster_fwd keeps per (device, stream) for `device` (< 0: every device); returns the number of blocks
 * freed.  No call of the library may be in flight on that device.  The next forward on a stream creates its block again. */
int hgs_release_device_state(int device);

typedef struct hgs_raster_grads {
  float* dL_dmeans3D;   /* [P,3] */
  float* dL_dmeans2D;   /* [P,3]  gradient w.r.t. NDC-scaled screen position (consumer: scene/gaussian_model.py:687-689) */
  float* dL_dshs;       /* [P,M,3] or NULL */
  float* dL_dcolors;    /* [P,3]  or NULL */
  float* dL_dopacity;   /* [P] */
  float* dL_dscales;    /* [P,3] or NULL */
  float* dL_drotations; /* [P,4] or NULL */
  float* dL_dcov3D;     /* [P,6] or NULL */
  float* dL_dshs_rest;  /* [P,M-1,3] with args.shs_rest (dL_dshs is then [P,1,3]), else NULL */
} hgs_raster_grads;

/* Backward.  Needs the forward outputs (out_color, out_invdepth) and the
 * workspaces exactly as stage 2 left them.  dL_dinvdepth may be NULL. */
int hgs_raster_bwd(const hgs_raster_args* a, const void* geom_ws, const void* bin_ws,
                   const void* img_ws, void* bwd_ws, uint32_t L,
                   const float* out_color, const float* out_invdepth,
                   const float* dL_dcolor, const float* dL_dinvdepth,
                   const hgs_raster_grads* grads, hgs_stream_t stream, int device);

/* SH part of the backward for up to HGS_MAX_DEFERRED_VIEWS views of the SAME Gaussians in one pass (gradient
 * accumulation over the views of one optimiser step / of one data-parallel rank): the [P,M,3] coefficients are read
 * once and dL_dshs is written once, instead of once per view.  Each view ran hgs_raster_bwd with
 * args.defer_sh_bwd = 1; its geom_ws and bwd_ws (the per-Gaussian colour gradients live there) must still be intact.
 * dL_dshs = (accumulate ? dL_dshs : 0) + sum over the views; dL_dmeans3D += the views' view-direction terms. */
#define HGS_MAX_DEFERRED_VIEWS 8
typedef struct hgs_sh_bwd_view {
  const void* geom_ws;
  const void* bwd_ws;
  const float* campos; /* device [3] */
  uint32_t L;          /* the L the view's hgs_raster_bwd was called with */
  uint32_t reserved;
} hgs_sh_bwd_view;
int hgs_raster_sh_bwd_batched(const hgs_sh_bwd_view* views, int32_t n_views, int32_t P, int32_t M, int32_t sh_degree,
                              const float* means3D, const float* shs, float* dL_dshs, float* dL_dmeans3D,
                              int32_t accumulate, hgs_stream_t stream, int device);

/* View-dependent colours of the same Gaussians for up to HGS_MAX_DEFERRED_VIEWS cameras in one pass over the SH
 * coefficients -- the batched HIP form of the reference's convert_SHs_python branch (gaussian_renderer/__init__.py:
 * 84-89: eval_sh + 0.5, clamped at 0, result handed to the rasterizer as colors_precomp).  Forward writes rgb [P,3] and
 * a clamp mask [P] (bit c set: channel c was clamped) per view; backward takes dL/d(rgb) per view (the dL_dcolors of
 * the views' hgs_raster_bwd), masks it, and writes dL_dshs = (accumulate ? dL_dshs : 0) + sum over the views and
 * dL_dmeans3D += the view-direction terms. */
typedef struct hgs_sh_color_view {
  const float* campos;  /* device [3] */
  float* rgb;           /* forward out [P,3] */
  uint8_t* clamp;       /* forward out / backward in [P] */
  const float* d_rgb;   /* backward in [P,3] */
} hgs_sh_color_view;
int hgs_sh_colors_batched(const hgs_sh_color_view* views, int32_t n_views, int32_t P, int32_t M, int32_t sh_degree,
                          const float* means3D, const float* shs, hgs_stream_t stream, int device);
int hgs_sh_colors_batched_bwd(const hgs_sh_color_view* views, int32_t n_views, int32_t P, int32_t M, int32_t sh_degree,
                              const float* means3D, const float* shs, float* dL_dshs, float* dL_dmeans3D,
                              int32_t accumulate, hgs_stream_t stream, int device);

/* Introspection for the parity tests ("bit-exact on tile/sort indices"):
 * device pointers into the workspaces after stage 2. */
typedef struct hgs_raster_views {
  const uint32_t* tile_ids_sorted; /* [L] tile id per sorted instance; with depths[point_list[i]] this is the
                                     (tile<<32 | depth bits) key sequence of the reference's sort.  Written by a forward
                                     with args.debug != 0 (and on the radix path of tile grids above 32 768 tiles); the
                                     binning path does not need the column -- entry i lies in [ranges[t][0], ranges[t][1])
                                     of its tile t, which is how the Python host rebuilds it */
  const uint32_t* point_list;    /* [L] Gaussian id per sorted instance */
  const uint32_t* ranges;        /* [T,2] start,end per tile */
  const uint32_t* tiles_touched; /* [P] */
  const uint32_t* offsets;       /* [P] exclusive scan of tiles_touched */
  const float* depths;           /* [P] view-space z */
  const uint32_t* rects;         /* [P,2] packed (x | y<<16) min, max in tile units */
  const float* records;          /* [P,16] per-Gaussian 2D record, 64 bytes */
  const float* final_T;          /* [H*W] */
  const uint32_t* n_contrib;     /* [H*W] */
} hgs_raster_views;
int hgs_raster_views_get(int32_t P, int32_t width, int32_t height, uint32_t L, const void* geom_ws,
                         const void* bin_ws, const void* img_ws, hgs_raster_views* out);

/* Standalone device sort of (u64 key, u32 value) pairs, stable, LSD radix over
 * bits [0, end_bit).  Exposed for the sort parity tests.  tmp_bytes from
 * hgs_sort_tmp_bytes. Result lands in keys_out/vals_out. */
size_t hgs_sort_tmp_bytes(uint32_t n);
int hgs_sort_pairs(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out,
                   uint32_t* vals_out, void* tmp, uint32_t n, int end_bit,
                   hgs_stream_t stream, int device);

/* Optional per-stage timing with hipEvents recorded on the caller's stream (bench.py's
 * per-kernel roofline figures).  Off by default; no reference counterpart (the reference
 * records CUDA events it never reads: train_single.py:41-42,86,124). */
/* on: 0 = off, 1 = every stage, otherwise a mask with bit (1 + i) selecting stage i only (two events per timed
 * stage are recorded on the stream, so timing one kernel perturbs a measured region less than timing all). */
int hgs_timing_enable(int on);
int hgs_timing_stage_count(void);
const char* hgs_timing_stage_name(int i);
/* Resolves all pending events (blocks until they completed) and returns the accumulated
 * milliseconds and call counts per stage; reset != 0 clears the accumulators. */
int hgs_timing_read(double* ms_out, uint32_t* calls_out, int reset);

/* ---------------------------------------------------------------------------
 * Hierarchy LOD cut.  Replaces gaussian_hierarchy._C.expand_to_size and
 * get_interpolation_weights (train_post.py:91-113, render_hierarchy.py:58-80).
 *   nodes int32 [N,7] = depth,parent,start,count_leafs,count_merged,start_children,count_children
 *   boxes f32  [N,2,4] = min.xyz + extent, max.xyz + pad
 * size(n, v) = extent(n) / dist(v, AABB(n)), FLT_MAX for v inside the box.
 * Cut (top-down from the root): a reached node with size >= tau is too coarse -- its count_leafs own Gaussians
 * [start, start + count_leafs) are drawn and its children are reached; a reached node with size < tau is drawn as
 * a whole (count_leafs + count_merged Gaussians from start).  parent index = nodes[parent].start (own index at the
 * root).  Output in ascending node order.
 * Weight of a cut node: 1 at the root; else with p = min(size(parent), 2 tau), s0 = max(p / 2, size(n)):
 * t = 1 if p <= s0, else max(1 - max(0, tau - s0) / (p - s0), 0); num_siblings = count_children of the parent.
 * ------------------------------------------------------------------------- */
size_t hgs_expand_tmp_bytes(int32_t N);
/* Fills render_indices / parent_indices / nodes_for_render_indices (device, capacity
 * `capacity` entries) and returns the number of entries in *count_out_host (host sync). */
int hgs_expand_to_size(const int32_t* nodes, const float* boxes, int32_t N, float size,
                       const float viewpoint[3], const float viewdir[3],
                       int32_t* render_indices, int32_t* parent_indices,
                       int32_t* nodes_for_render_indices, int32_t capacity, void* tmp,
                       int32_t* count_out_host, hgs_stream_t stream, int device);
/* Same cut in ONE pass over the nodes instead of one launch per tree level -- valid when the boxes NEST (every
 * child's AABB inside its parent's, child extent <= parent extent): then size(parent) >= size(child) from every
 * viewpoint and a node only has to look at its parent.  hgs_hier_boxes_nested checks the precondition (view
 * independent: once per hierarchy; tmp >= 4 bytes of device memory); the Python binding caches the answer per
 * (nodes, boxes) pair and falls back to hgs_expand_to_size when it is 0.  Same arguments, same outputs. */
int hgs_expand_to_size_nested(const int32_t* nodes, const float* boxes, int32_t N, float size,
                              const float viewpoint[3], const float viewdir[3],
                              int32_t* render_indices, int32_t* parent_indices,
                              int32_t* nodes_for_render_indices, int32_t capacity, void* tmp,
                              int32_t* count_out_host, hgs_stream_t stream, int device);
int hgs_hier_boxes_nested(const int32_t* nodes, const float* boxes, int32_t N, void* tmp, int32_t* nested_out_host,
                          hgs_stream_t stream, int device);
int hgs_interp_weights(const int32_t* node_indices, int32_t n, float size, const int32_t* nodes,
                       const float* boxes, int32_t N, const float viewpoint[3], const float viewdir[3],
                       float* interpolation_weights, int32_t* num_siblings,
                       hgs_stream_t stream, int device);

/* In-op LOD attribute interpolation: the gather + lerp that render_post does in Python
 * (gaussian_renderer/__init__.py:199-218), for callers that pass GaussianRasterizationSettings.render_indices /
 * parent_indices non-empty.  out_i = w_i * attr[render_indices[i]] + (1 - w_i) * attr[parent_indices[i]]; rotations
 * with the parent quaternion flipped into the node's hemisphere.  Any attribute pointer may be NULL (skipped). */
int hgs_lod_gather(const int32_t* render_indices, const int32_t* parent_indices, const float* weights, int32_t n,
                   int32_t M, const float* means3D, const float* scales, const float* rotations, const float* shs,
                   const float* opacities, float* o_means3D, float* o_scales, float* o_rotations, float* o_shs,
                   float* o_opacities, hgs_stream_t stream, int device);
/* Backward: g_* = gradients of the n interpolated rows; d_* = gradients of the full arrays, ZERO-INITIALISED by
 * the caller; rotations = the full forward input (sign of the hemisphere flip); flag_tmp = 4 bytes of device
 * scratch.  Precondition: render_indices are unique and no rendered row is another entry's parent row (true for
 * any LOD cut). */
int hgs_lod_gather_bwd(const int32_t* render_indices, const int32_t* parent_indices, const float* weights, int32_t n,
                       int32_t M, const float* rotations, const float* g_means3D, const float* g_scales,
                       const float* g_rotations, const float* g_shs, const float* g_opacities, float* d_means3D,
                       float* d_scales, float* d_rotations, float* d_shs, float* d_opacities, uint32_t* flag_tmp,
                       hgs_stream_t stream, int device);

/* ---------------------------------------------------------------------------
 * Fused row-sparse Adam.  Replaces the gather / update / scatter chain of
 * scene/OurAdam.py:249-337 (_single_tensor_adam; dense variant :339-420) for ALL parameter tensors of the model in
 * one launch.  Every tensor is [P, row_len] contiguous f32; bias corrections and step size are computed by the caller
 * (step_size = lr / (1 - beta1^t), bias_correction2_sqrt = sqrt(1 - beta2^t), as the Python code does in double).
 * rows != NULL: update the n_rows listed rows (int64, as `relevant` in train_single.py:171-174);
 * row_mask_grad != NULL: update row r iff row_mask_grad[r] != 0 (the same selection, evaluated in-kernel);
 * both NULL: dense update of all P rows.
 * ------------------------------------------------------------------------- */
#define HGS_ADAM_MAX_TENSORS 8
typedef struct hgs_adam_tensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int32_t row_len;
  float step_size;
  float beta1, one_minus_beta1; /* both rounded from double by the caller, as torch does with its scalar arguments */
  float beta2, one_minus_beta2;
  float eps, weight_decay;
  float bias_correction2_sqrt;
  int32_t reserved;
} hgs_adam_tensor;
int hgs_adam_step(const hgs_adam_tensor* tensors, int32_t n_tensors, int64_t P, const int64_t* rows, int64_t n_rows,
                  const float* row_mask_grad, hgs_stream_t stream, int device);

/* ---------------------------------------------------------------------------
 * simple_knn._C.distCUDA2 (scene/gaussian_model.py:190): mean squared distance
 * to the 3 nearest neighbours.  tmp_bytes from hgs_knn_tmp_bytes.
 * ------------------------------------------------------------------------- */
size_t hgs_knn_tmp_bytes(int32_t P);
int hgs_dist2_knn3(const float* xyz, int32_t P, float* out_mean_d2, void* tmp,
                   hgs_stream_t stream, int device);

/* ---------------------------------------------------------------------------
 * .hier files.  Replaces gaussian_hierarchy._C.load_hierarchy / write_hierarchy
 * (scene/gaussian_model.py:329,420-427).  Host memory only.
 * ------------------------------------------------------------------------- */
enum {
  HGS_HIER_UPSTREAM = 0,      /* header-less layout of the gaussian-hierarchy tools: int P, pos, rot, log-scale, alpha,
                               * sh[16][3], int N, nodes, boxes (restated from the public repository; see hier_io.cpp) */
  HGS_HIER_PRIVATE = 1,       /* "HGSHIER1" + P, N, M: any SH count */
  HGS_HIER_UPSTREAM_HALF = 2  /* upstream layout with P < 0: rot / scale / alpha / sh stored as IEEE half (read only) */
};
typedef struct hgs_hier_host {
  int32_t P;      /* Gaussians */
  int32_t N;      /* nodes */
  int32_t M;      /* SH coefficients per Gaussian (16) */
  int32_t reserved; /* layout: hgs_hier_write takes HGS_HIER_UPSTREAM or HGS_HIER_PRIVATE; hgs_hier_load reports what it found */
  float* xyz;         /* [P,3] */
  float* shs;         /* [P,M,3] */
  float* alpha;       /* [P] activated opacity */
  float* log_scales;  /* [P,3] */
  float* rots;        /* [P,4] */
  int32_t* nodes;     /* [N,7] */
  float* boxes;       /* [N,2,4] */
} hgs_hier_host;
int hgs_hier_load(const char* path, hgs_hier_host* out); /* allocates; release with hgs_hier_free */
int hgs_hier_write(const char* path, const hgs_hier_host* in);
void hgs_hier_free(hgs_hier_host* h);

/* ---------------------------------------------------------------------------
 * Direct (two-shot) SUM all-reduce over peer pointers: the exchange step of per-view data parallelism (SURVEY.md
 * section 8(e); the reference itself is single-GPU: train_single.py:57-59 renders one camera per step, nothing to
 * replace).  One process per GPU; every rank allocates its gradient bucket and a small flag block with hgs_p2p_alloc,
 * exports both (hipIpc), opens its peers' and then calls hgs_p2p_allreduce_sum: rank r sums shard r of ALL buckets in
 * rank order (reading the peers' memory over xGMI: all links at once, where a ring is bound by one) and writes it back
 * into its own bucket, then copies the other ranks' reduced shards from their buckets -- every rank ends with
 * bit-identical sums.  Three flag barriers per call (release / acquire at system scope, bounded spin: a peer that
 * never arrives sets the error word instead of hanging the device).  world <= HGS_P2P_MAX_WORLD.
 * Opt-in (hgs/dp.py: HGS_DP_ALLREDUCE=direct); the default exchange is RCCL's all-reduce through torch.distributed.
 * ------------------------------------------------------------------------- */
#define HGS_P2P_MAX_WORLD 8
#define HGS_P2P_HANDLE_BYTES 64
#define HGS_P2P_FLAG_BYTES 256   /* size of a rank's flag block: barrier words 0..2, error word 3 */
/* flags: bit 0 = uncached memory (the flag block); bit 1 = fine-grained device memory (a bucket that must be coherent
 * across devices INSIDE a kernel; the protocol below only hands buckets over at kernel boundaries and uses ordinary
 * memory, flags = 0); a barrier that waits longer than HGS_P2P_TIMEOUT_S (default 60) sets the sticky error word and
 * makes the rank's reduce / gather write NaN (csrc/p2p.hip) */
int hgs_p2p_alloc(size_t bytes, int32_t flags, void** ptr, int device);
int hgs_p2p_free(void* ptr, int device);
int hgs_p2p_export(void* ptr, uint8_t handle[HGS_P2P_HANDLE_BYTES], int device);
int hgs_p2p_open(const uint8_t handle[HGS_P2P_HANDLE_BYTES], void** ptr, int device);
int hgs_p2p_close(void* ptr, int device);
/* bufs[world] / flag_blocks[world]: device pointers valid in THIS process (own allocation at [rank], opened peers
 * elsewhere).  Reduces the floats [offset, offset + n) of every bucket; offset and n multiples of 4.  epoch: a counter
 * that every rank increments by one per call (all ranks must make the same sequence of calls).  Stream-ordered: the
 * buckets must have been written by work enqueued earlier on `stream`; the result is complete, and the bucket may be
 * overwritten, after the calls' kernels.  The error word (flag block word 3) is non-zero after a timed-out barrier. */
int hgs_p2p_allreduce_sum(int32_t rank, int32_t world, void* const* bufs, void* const* flag_blocks, size_t offset,
                          size_t n, uint32_t epoch, hgs_stream_t stream, int device);

/* ---------------------------------------------------------------------------
 * Budgeted residency of a hierarchy's attribute rows ("VRAM-budgeted streaming LOD": the `--budget <MB>` of the
 * reference's hierarchy viewer, README.md:233-235 -- "this only defines the budget for the SCENE representation" --
 * whose implementation lives in the un-vendored SIBR viewer; BASELINE configs[4] names it).  Opt-in layer BESIDE the
 * drop-in path (hgs/residency.py): the attributes of ALL rows stay in pinned, device-mapped HOST memory
 * (hgs_host_alloc) as packed rows (HGS_RESID_HOST_ROW_FLOATS); the GPU holds `B` rows in slot arrays.  Per view, after
 * the LOD cut and its weights:
 *   hgs_resid_mark    every row the cut needs (node row of each entry, and its parent row unless `weights` -- nullable,
 *                     the entries' interpolation weights -- says the weight is exactly 1: the in-op LOD gather does
 *                     not read that parent, and po then repeats the node's slot): resident -> stamped with the
 *                     frame number; absent -> appended ONCE to the miss list (slot_of: >= 0 slot, -1 absent, -2 queued
 *                     this frame); ro / po receive the slots of the resident rows.  Waits; *miss_count_host = rows
 *                     queued -- also when the call fails with HGS_ERR_INVALID on an index outside [0, G): the caller
 *                     has to take them out of the queue again (slot_of back to -1).
 *   hgs_resid_evict   when the free list is shorter than the miss list: frees the slots that have gone unused for
 *                     the longest (age histogram on the device, threshold chosen on the host; rows stamped this frame
 *                     are never evicted).  HGS_ERR_CAPACITY if even that is not enough: the working set of the view
 *                     exceeds the budget -- the caller raises tau, as the reference's viewer "auto-regulates".
 *   hgs_resid_fetch   assigns free slots to the missing rows and copies their attributes from the packed host rows
 *                     (HGS_RESID_HOST_ROW_FLOATS) into the slot arrays: ONE kernel reading host memory directly (zero
 *                     copy over PCIe, one coalesced 256-byte read per row), no staging buffer, no host-side gather.
 *   hgs_resid_remap   render_indices / parent_indices (Gaussian rows) -> slot indices, for the in-op LOD path of the
 *                     rasterizer (hgs_raster_args.lod_*) running on the slot arrays.
 * slot_of int32 [G] (initialised to -1), stamp uint32 [B], id_of_slot int32 [B] (initialised to -1), free_list int32
 * [B] (initialised to B-1 .. 0: slot 0 is handed out first), counters uint32 [HGS_RESID_COUNTER_WORDS] (device
 * scratch), miss_ids int32
 * [2 x capacity of the index arrays].  All calls are ordered on `stream`.
 * ------------------------------------------------------------------------- */
#define HGS_RESID_COUNTER_WORDS 68
/* host side of the budgeted residency: one packed row of 64 floats (256 B = four 64-byte PCIe reads) per Gaussian, in
 * memory from hgs_host_alloc:  [0, 3 M) SH coefficients, [48, 52) rotation, [52, 55) mean, [55, 58) scale, [58] opacity */
#define HGS_RESID_HOST_ROW_FLOATS 64
typedef struct hgs_resid_rows {
  float* means3D;    /* [rows, 3]    */
  float* shs;        /* [rows, M, 3] */
  float* opacities;  /* [rows]       */
  float* scales;     /* [rows, 3]    */
  float* rotations;  /* [rows, 4]    */
} hgs_resid_rows;
void* hgs_host_alloc(size_t bytes);      /* pinned host memory mapped into every device's address space; NULL on failure */
void hgs_host_free(void* p);
int hgs_resid_mark(const int32_t* render_indices, const int32_t* parent_indices, const float* weights, int32_t n, int32_t G,
                   int32_t* slot_of, uint32_t* stamp, uint32_t frame, int32_t* miss_ids, uint32_t* counters,
                   int32_t* ro, int32_t* po, uint32_t* miss_count_host, hgs_stream_t stream, int device);
int hgs_resid_evict(uint32_t* stamp, int32_t* id_of_slot, int32_t* slot_of, int32_t B, uint32_t frame, uint32_t need,
                    int32_t* free_list, uint32_t* counters, uint32_t* free_top_inout_host, hgs_stream_t stream,
                    int device);
int hgs_resid_fetch(const int32_t* miss_ids, uint32_t m, const int32_t* free_list, uint32_t free_top, int32_t* slot_of,
                    int32_t* id_of_slot, uint32_t* stamp, uint32_t frame, const float* host_rows_packed,
                    const hgs_resid_rows* slot_rows, int32_t M, hgs_stream_t stream, int device);
int hgs_resid_remap(const int32_t* render_indices, const int32_t* parent_indices, const float* weights, int32_t n,
                    const int32_t* slot_of, int32_t* ro, int32_t* po, hgs_stream_t stream, int device);

#ifdef __cplusplus
}
#endif
#endif /* HGS_H */
