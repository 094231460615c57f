/* gaussreg_hip.h -- C ABI of libgaussreg_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for GaussReg's two data-parallel hot paths (BASELINE.json north_star):
 *   (1) the GeoTransformer point-cloud op stack
 *   (2) the 3D-Gaussian-splatting rasterizer forward
 * Every entry point below names the reference interface it replaces (paths relative to the
 * GaussReg tree).  Conventions:
 *   - plain C types only; all pointers are DEVICE pointers unless the name starts with `h_`;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it.  Entry points that must
 *     return a data-dependent size to the host (documented per function) synchronise that stream;
 *   - the caller owns every buffer, including the workspace `ws` (size it with the matching
 *     *_workspace_bytes function; contents are scratch unless stated otherwise);
 *   - return value: 0 = ok, <0 = error (see gr_last_error()).  Nothing is ever computed on the CPU:
 *     if no gfx950 device is usable the call fails, it does not fall back.
 */
#ifndef GAUSSREG_HIP_H_
#define GAUSSREG_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GR_OK 0
#define GR_ERR_INVALID -1   /* bad argument */
#define GR_ERR_HIP -2       /* a HIP runtime call failed */
#define GR_ERR_WORKSPACE -3 /* workspace too small */
#define GR_ERR_UNSUPPORTED -4

/* Thread-local description of the last error returned on this thread. */
const char* gr_last_error(void);
/* Library/ABI version (major*1000 + minor). */
int gr_version(void);

/* Optional per-kernel timing with HIP events recorded on the launch stream (measurement aid, off
 * by default).  Timer names: "radius_bin", "radius_count", "radius_fill", "radius_fused", "raster_preprocess",
 * "raster_depth_sort", "raster_bin", "raster_blend", "fps", "sinkhorn", "lgr", "ransac", "kpconv", "group_norm",
 * "geo_embedding", "rpe_attention", "rpe_scores", "gs_fuse".  gr_timing_read waits for the recorded events and returns total ms / launches. */
void gr_timing_enable(int on);
void gr_timing_reset(void);
int gr_timing_read(const char* name, double* total_ms, int64_t* calls);

/* ------------------------------------------------------------------------------------------------
 * radius_neighbors  -- replaces
 *   geotransformer/extensions/cpu/radius_neighbors/radius_neighbors.cpp:5-68     (entry, alloc)
 *   geotransformer/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91 (search)
 * bound in Python as geotransformer.ext.radius_neighbors (geotransformer/extensions/pybind.cpp:8-12).
 *
 * Stack mode: q (nq,3) / s (ns,3) fp32 row-major hold `batch` clouds back to back; h_q_lengths /
 * h_s_lengths (host, int64[batch]) give their sizes.  A support j is a neighbour of query i (same
 * batch element) iff fp32 ((dx*dx + dy*dy) + dz*dz) < radius*radius; rows are sorted by ascending
 * distance (ties: ascending index) and padded with `ns`.
 *
 * The row width is data dependent (radius_neighbors.cpp:54), so the op is two calls:
 *   gr_radius_count : bins the supports into a uniform grid, counts, SYNCHRONISES `stream`, writes
 *                     h_info[0] = max_count (the width the reference would return),
 *                     h_info[1..3] = opaque plan values for the fill call.
 *   gr_radius_fill  : writes out (nq, width) int64, width <= h_info[0] keeps the nearest `width`
 *                     (== the truncation radius_search applies, modules/ops/radius_search.py:25-26).
 * `ws` must be the same, untouched buffer in both calls.
 */
size_t gr_radius_workspace_bytes(int64_t nq, int64_t ns, int64_t batch);
int gr_radius_count(const float* q, const float* s, const int64_t* h_q_lengths,
                    const int64_t* h_s_lengths, int64_t nq, int64_t ns, int64_t batch, float radius,
                    void* ws, size_t ws_bytes, int64_t* h_info /*[4]*/, void* stream);
/* Same as gr_radius_count, for callers that search the SAME supports (pointer, lengths, radius) several times with
 * different queries -- the data pyramid does so three times per level (geotransformer/utils/data.py:44-75).  The
 * call with reuse_support = 0 bins the supports and writes a signature to h_support_sig[4]; later calls with
 * reuse_support = 1, the same `ws` (sized for the largest nq used) and that signature skip the support binning
 * (checked: a different cloud pointer / lengths / radius is an error).  The support data must not change between. */
int gr_radius_count_cached(const float* q, const float* s, const int64_t* h_q_lengths,
                           const int64_t* h_s_lengths, int64_t nq, int64_t ns, int64_t batch, float radius,
                           void* ws, size_t ws_bytes, int64_t* h_info /*[4]*/, int64_t* h_support_sig /*[4]*/,
                           int reuse_support, void* stream);
int gr_radius_fill(const float* q, const float* s, int64_t nq, int64_t ns, int64_t batch,
                   float radius, int64_t width, const int64_t* h_info /*[4]*/, int64_t* out,
                   void* ws, size_t ws_bytes, void* stream);
/* radius_search with a positive neighbor_limit -- replaces the wrapper
 *   geotransformer/modules/ops/radius_search.py:7-27  (ext.radius_neighbors, then `[:, :neighbor_limit]`, :25-26)
 * The kept width min(max_count, limit) is at most `limit`, so the caller allocates `out` as (nq, limit) int64 BEFORE
 * anything is counted (row stride = limit; columns past a query's hit count hold the padding value ns) and the host is
 * not needed between the kernels.  SYNCHRONISES `stream` once, at the end, to return
 *   h_info[0] = max_count (the reference's untruncated width): the result is out[:, :min(max_count, limit)];
 *   h_info[4] = 1 if ONE kernel produced `out` (modes 1 - 5 below), 0 if count + fill did.
 * Modes (gr_radius_search_mode; returns the previous mode; a negative argument only queries; GR_RADIUS_SINGLE_PASS=<mode>
 * sets the process default).  Every mode returns the same rows; gr_radius_count / gr_radius_fill follow the mode too:
 *   3  (default) per (radius, limit) call site the library picks one of the kernels below and remembers a give-up, or a
 *      call in which more than an eighth of the queries went beyond the network: the thread-per-query kernel with its
 *      32-hit network -> the same with its 64-hit network -> the 64-hit network behind a pre-selection of the `limit`
 *      nearest hits (limit <= 56: rows truncated far below the hit count, the coarsest pyramid levels) -> count + fill,
 *      stepping back one level every 256 calls;
 *   5  always try the pre-selecting kernel first (gr_radius_search only; gr_radius_count has no width to select for and
 *      takes the plain 64-hit network);
 *   2 / 4  always try the thread-per-query kernel first, 32- / 64-hit network (csrc/radius_tq.hpp: one wave = 64
 *      cell-ordered queries, candidates through the vector L1, hits sorted in registers, rows through LDS); queries beyond
 *      the network are finished exactly by their wave; a query with more than 192 hits hands the call back to count + fill;
 *   0  count, host, fill -- three threads per query, the rows allocated up front;
 *   1  the three-threads-per-query single pass (tests, LDS ranking, whole-row stores); falls back to count + fill when one
 *      query has more hits than a workgroup's key area or the limit is too wide for LDS.
 * With modes 2 - 5 gr_radius_count does the whole search (compact u32 rows in the workspace, h_info[1] = -1) and
 * gr_radius_fill only widens them to int64 rows of the final width.
 * h_support_sig / reuse_support as in gr_radius_count_cached (may be NULL / 0).  `ws`: gr_radius_workspace_bytes. */
int gr_radius_search_mode(int mode);
int gr_radius_search(const float* q, const float* s, const int64_t* h_q_lengths, const int64_t* h_s_lengths,
                     int64_t nq, int64_t ns, int64_t batch, float radius, int64_t limit, int64_t* out /* nq x limit */,
                     void* ws, size_t ws_bytes, int64_t* h_info /*[6]*/, int64_t* h_support_sig /*[4] or NULL*/,
                     int reuse_support, void* stream);

/* ------------------------------------------------------------------------------------------------
 * grid_subsampling -- replaces
 *   geotransformer/extensions/cpu/grid_subsampling/grid_subsampling.cpp:5-62      (entry)
 *   geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:3-75  (voxel hashing)
 * bound as geotransformer.ext.grid_subsampling (pybind.cpp:13-17).
 *
 * out_points has capacity (n,3); h_out_lengths (host, int64[batch]) receives m_b; *h_total_m the
 * total.  Barycentres are bit-identical to the reference (sequential fp32 sums in input order,
 * times float(1.0/count)).  order_mode selects the ROW ORDER inside each cloud:
 *   GR_ORDER_REFERENCE  the reference's std::unordered_map iteration order, evaluated on the device in closed form
 *                       (csrc/hash_order_device.hip; checked against the host replay of libstdc++'s linking rules,
 *                       gr_host_unordered_map_order), bit-for-bit the tensor the reference returns;
 *   GR_ORDER_CELL       ascending voxel key -- fully on device, same multiset of rows.
 * Synchronises `stream` (the output size is data dependent).
 */
/* Host-only helper behind GR_ORDER_REFERENCE (exported so it can be tested without a GPU):
 * h_perm[j] = index of the j-th key std::unordered_map<size_t,...> iterates after inserting the
 * n distinct h_keys in order. */
int gr_host_unordered_map_order(const uint64_t* h_keys, int64_t n, int32_t* h_perm);
#define GR_ORDER_REFERENCE 0
#define GR_ORDER_CELL 1
/* Test hooks for the row order: gr_host_unordered_map_order replays libstdc++'s container on the host;
 * gr_hash_order_device evaluates the same order on the device for `batch` clouds at once (d_keys: distinct keys per cloud
 * in insertion order, clouds contiguous; h_begins: batch + 1 HOST offsets; d_perm[begin_c + j] = global index of the j-th
 * key the container would iterate).  gr_hash_order_device is asynchronous: d_perm is complete in `stream` order. */
size_t gr_hash_order_device_workspace_bytes(int64_t n, int64_t batch);
int gr_hash_order_device(const uint64_t* d_keys, const int64_t* h_begins, int64_t batch, int32_t* d_perm, void* ws,
                         size_t ws_bytes, void* stream);
/* Test switch: 1 = the device evaluation takes the slab-table pre-scan of its > 4 M-clock stages at any size (same order).
 * Returns the old value; another argument only queries. */
int gr_hash_order_debug_force_prescan(int on);
/* Test switch: 0 = gr_grid_subsample always sorts with the general three-pass radix sort, 1 (default) = batches that qualify take
 * the bucket sort (same rows, bit for bit).  Returns the old value; 2 returns instead how many calls of this process started over
 * with the general sort because a bucket overflowed; another argument only queries the switch. */
int gr_grid_subsample_debug_bucket_sort(int on);
size_t gr_grid_subsample_workspace_bytes(int64_t n, int64_t batch);
int gr_grid_subsample(const float* points, const int64_t* h_lengths, int64_t n, int64_t batch,
                      float voxel_size, int order_mode, float* out_points, int64_t* h_out_lengths,
                      int64_t* h_total_m, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 3D-Gaussian-splatting rasterizer, forward only -- the op GaussReg's fine-registration rendering
 * needs (BASELINE.json north_star).  NOT PRESENT in the reference tree (the only
 * trace is the acknowledgement at README.md:150); the interface replaced is the public
 *   diff_gaussian_rasterization.GaussianRasterizer.forward / .markVisible
 *   (graphdeco-inria/diff-gaussian-rasterization: rasterize_points.cu RasterizeGaussiansCUDA,
 *    cuda_rasterizer/rasterizer_impl.cu CudaRasterizer::Rasterizer::forward / markVisible)
 * with the same argument meaning.  Algorithm + numerics contract: oracle/rasterizer_oracle.c
 * (the HIP image is bit-identical to that restatement).
 *
 * Batched over views: one Gaussian set, `num_views` cameras (HOST array of gr_raster_view), so
 * per-Gaussian inputs are read once per batch.  Two calls because the number of (tile, Gaussian)
 * instances R is data dependent (upstream syncs at the same place):
 *   gr_raster_preprocess : cull / project / cov2D / SH->RGB / depth order / per-tile counts and prefix sums into
 *                          `geom` (size gr_raster_geom_bytes), SYNCHRONISES `stream`,
 *                          h_num_rendered has num_views + 1 entries: [num_views] is a binning hint for
 *                          gr_raster_render (largest per-chunk instance count); pass the array on unchanged.
 *                          h_num_rendered[v] = R_v = instances binned (<= upstream's count: pairs that
 *                          cannot reach alpha = 1/255 inside the tile are dropped, the image is unaffected).
 *                          Also writes radii (num_views, P) int32, exactly upstream's values.
 *   gr_raster_render     : ranked scatter of the depth-ordered Gaussians into per-tile lists (== upstream's sort by
 *                          (tile, depth), without a key stream), tile ranges, per-tile front-to-back alpha blend
 *                          -> out_color (num_views, 3, H, W) fp32.
 *                          `bin` is scratch of size gr_raster_bin_bytes(sum R_v, ...).
 * Exactly one of shs / colors_precomp and exactly one of (scales, rotations) / cov3D_precomp is
 * non-null.  All views of one call share image size and SH degree (checked).
 */
typedef struct gr_raster_view {
  int32_t image_height, image_width;
  float tanfovx, tanfovy;
  float bg[3];
  float scale_modifier;
  float viewmatrix[16]; /* as the Python API passes it: world->view transposed, read column-major */
  float projmatrix[16]; /* full projection, same convention */
  float campos[3];
  int32_t sh_degree;
  int32_t prefiltered;
  int32_t debug;
} gr_raster_view;

size_t gr_raster_geom_bytes(int64_t P, int num_views, int width, int height);
/* Test hook: byte offsets inside a geometry buffer of, in this order, the per-(view, Gaussian) depth fields (uint32,
 * 0 = culled), the depth-ordered Gaussian ids (int32, stride P per view), their packed rectangles (uint32) and the
 * per-view visible counts (int32) -- what the depth sort of the last frame left there.  Returns 4. */
int gr_raster_debug_geom_layout(int64_t P, int num_views, int width, int height, int64_t* h_offsets);
/* Test hook: rasterizer calls of this host thread that stay with the three-pass depth sort because a bucket of the
 * four-launch one overflowed (0 = the four-launch sort is tried).  Returns the count; set >= 0 replaces it (a large value
 * pins the three-pass sort, 0 ends the period), set < 0 only queries. */
int gr_raster_debug_bucket_cooldown(int set);
size_t gr_raster_bin_bytes(int64_t total_rendered, int width, int height, int num_views);
int gr_raster_preprocess(int64_t P, int sh_coeffs, const float* means3D, const float* shs,
                         const float* colors_precomp, const float* opacities, const float* scales,
                         const float* rotations, const float* cov3D_precomp,
                         const gr_raster_view* h_views, int num_views, int32_t* radii, void* geom,
                         size_t geom_bytes, int64_t* h_num_rendered, void* stream);
int gr_raster_render(int64_t P, const gr_raster_view* h_views, int num_views,
                     const int64_t* h_num_rendered, const void* geom, size_t geom_bytes, void* bin,
                     size_t bin_bytes, float* out_color, void* stream);
/* gr_raster_render with options.  GR_RASTER_FAST_EXP: the blend evaluates alpha = opacity * exp(power) with the hardware
 * exponential (v_exp_f32) instead of the deterministic polynomial of oracle/rasterizer_oracle.c: the image is then within
 * 1e-5 relative of the bit-exact one instead of identical to it (north_star's bar for rendered RGB); ~20 % faster blend.
 * gr_raster_render itself takes this flag from the environment variable GR_RASTER_FAST_EXP=1 (default: exact). */
#define GR_RASTER_FAST_EXP 1
#define GR_RETRY_BIN 1 /* gr_raster_forward: `bin` too small for this call's instances */
int gr_raster_render_ex(int64_t P, const gr_raster_view* h_views, int num_views, const int64_t* h_num_rendered,
                        const void* geom, size_t geom_bytes, void* bin, size_t bin_bytes, float* out_color, int flags,
                        void* stream);
/* gr_raster_preprocess + gr_raster_render_ex in one call, for callers that keep a binning buffer between calls (one camera
 * per call: the boundary of GaussianRasterizer.forward): the host leaves the library only once per frame.  `bin` is sized
 * from an earlier call's h_num_rendered; returns GR_RETRY_BIN (> 0, h_num_rendered filled, nothing rendered yet) when
 * this call needs more -- the caller then allocates gr_raster_bin_bytes(sum) and calls gr_raster_render_ex itself. */
int gr_raster_forward(int64_t P, int M, const float* means3D, const float* shs, const float* colors_precomp,
                      const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                      const gr_raster_view* h_views, int num_views, int32_t* radii, void* geom, size_t geom_bytes,
                      void* bin, size_t bin_bytes, float* out_color, int flags, int64_t* h_num_rendered, void* stream);
/* Split form for callers with host work of their own per frame (the Python wrapper's stream joins and bookkeeping):
 * with GR_RASTER_SPLIT in `flags`, gr_raster_forward returns GR_PENDING as soon as the frame is enqueued (speculatively,
 * on `bin`) instead of waiting for the instance counts; gr_raster_forward_finish -- same host thread, before its next
 * gr_raster_forward -- waits for them, fills h_num_rendered ([num_views] + the staging figure) and returns GR_OK (the frame
 * is complete on the stream), GR_RETRY_BIN (as above), or GR_RETRY_FULL (a depth >= 8192 needs the full-width sort: the
 * caller repeats the frame with an unsplit gr_raster_forward).  Without speculation (no `bin`, > 4 views, a verification
 * frame) gr_raster_forward ignores the flag and returns its usual codes. */
#define GR_RASTER_SPLIT 2
#define GR_RASTER_SHARE 4 /* with GR_RASTER_SPLIT: another frame of the caller runs next to this one -- the blend is launched at
                             half its usual occupancy so that the other frame's short kernels find wave slots */
#define GR_PENDING 2
#define GR_RETRY_FULL 3
int gr_raster_forward_finish(int64_t* h_num_rendered);
/* Which ranking the tile-binning scatter uses on the current device: 1 = one LDS atomic per instance (the device was
 * probed and serves equal-address lanes of a ds_add_rtn in lane order), 0 = explicit ballot ranking (probe failed, or
 * GR_RASTER_BALLOT_RANKING=1), -1 = no render call has probed the device yet.  Both produce the same lists. */
int gr_raster_lds_atomics_lane_ordered(void);
/* on = 1: the depth sort and the tile scatter take their stable ranks from explicit ballot ranking (architecturally
 * guaranteed) whatever the probe of the lane order of ds_add_rtn said; on = 0: as probed (default; GR_RASTER_BALLOT_RANKING=1
 * starts the process with 1); any other value only queries.  Process-wide; returns the previous setting.  Same images
 * either way (tests/test_gpu_rasterizer.py); bench.py reports both (config.ballot_ranking_views_per_s). */
int gr_raster_ballot_ranking(int on);
/* present[i] = 1 iff Gaussian i passes the near-plane test of `viewmatrix` (markVisible). */
int gr_raster_mark_visible(int64_t P, const float* means3D, const float* h_viewmatrix,
                           uint8_t* present, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Matching operators -- in the reference these are chains of stock ATen calls inside nn.Modules
 * (no native code: SURVEY.md section 0 F2); the entry points replace the bodies of:
 *
 * gr_pairwise_distance       geotransformer/modules/ops/pairwise_distance.py:4-31 (channel-last, 2-D):
 *                            out (n,m) = clamp((|x|^2 - 2 x y^T) + |y|^2, 0), or clamp(2 - 2 x y^T, 0) if
 *                            `normalized`; x y^T on fp32 MFMA.
 * gr_superpoint_matching     geotransformer/modules/geotransformer/superpoint_matching.py:13-50 (forward):
 *                            masks (uint8, may be null = all true), L2-normalised features (nr,c)/(ns,c);
 *                            out_* have room for num_correspondences entries; *h_num_out =
 *                            min(num_correspondences, #valid_ref * #valid_src).  Synchronises `stream`.
 * gr_corr_matrix             .../point_matching.py:32-66 == local_global_registration.py:49-83
 *                            (compute_correspondence_matrix on exp(score_mat), use_dustbin=False):
 *                            corr_mat (batch,k1,k2) uint8; *h_num_corr (optional) = number of true
 *                            entries (synchronises when non-null).
 * gr_corr_gather             .../point_matching.py:107-113: torch.nonzero order gathers; outputs sized by
 *                            the count gr_corr_matrix returned; `ws` must be the buffer used there.
 * gr_point_to_node_partition geotransformer/modules/ops/pointcloud_partition.py:61-111 (return_count=False):
 *                            point_to_node (n) i64, node_masks (m) u8, node_knn_indices (m,point_limit) i64
 *                            padded with n, node_knn_masks (m,point_limit) u8.
 */
/* gr_sinkhorn ("next" row): geotransformer/modules/sinkhorn/learnable_sinkhorn.py:20-66
 * LearnableLogOptimalTransport.forward: scores (batch,m,n), masks uint8 (null = all valid), alpha read
 * from DEVICE memory (the module's learnable parameter), out (batch, m+1, n+1) -- or, with drop_dustbin != 0,
 * (batch, m, n): the matrix without its dustbin row and column, which is all GaussReg keeps (model.py:197-198).  `workspace`: device
 * scratch of gr_sinkhorn_workspace_bytes(batch) (the work list of matrices too large for the one-wave kernel). */
size_t gr_sinkhorn_workspace_bytes(int64_t batch);
int gr_sinkhorn(const float* scores, int64_t batch, int64_t m, int64_t n, const uint8_t* row_masks,
                const uint8_t* col_masks, const float* alpha_dev, int num_iterations, float inf, int drop_dustbin,
                float* out, void* workspace, size_t workspace_bytes, void* stream);
/* gr_kpconv_forward ("next" row): geotransformer/modules/kpconv/kpconv.py:79-122 KPConv.forward
 * (rigid kernel points): s_feats (n,cin), q_points (m,3), s_points (n,3), neighbor_indices (m,h) int64 padded
 * with n, kernel_points (k,3), weights (k,cin,cout), bias (cout) or null -> out (m,cout).
 * gr_neighbor_pool: kpconv/functional.py maxpool (mode 0, :54-67) / nearest_upsample (mode 1, :6-22). */
size_t gr_kpconv_workspace_bytes(int64_t n, int64_t m, int64_t k, int64_t cin);
int gr_kpconv_forward(const float* s_feats, const float* q_points, const float* s_points,
                      const int64_t* neighbor_indices, int64_t n, int64_t m, int64_t h, int64_t cin, int64_t cout,
                      const float* kernel_points, int64_t k, const float* weights, const float* bias, float sigma,
                      float inf, float* out, void* ws, size_t ws_bytes, void* stream);
/* gr_gather_rows: geotransformer/modules/ops/index_select.py:4-31 for the case the backbone uses (dim 0 of a 2-D fp32
 * tensor, any index shape flattened to m entries): out (m, c) = data[index].  *d_error_flag (device int, caller zeroes
 * it) is set if an index is outside [0, n) -- torch raises there; the wrapper checks the flag lazily. */
int gr_gather_rows(const float* data, int64_t n, int64_t c, const int64_t* index, int64_t m, float* out,
                   int* d_error_flag, void* stream);
int gr_neighbor_pool(const float* x, int64_t n, int64_t c, const int64_t* neighbor_indices, int64_t m, int64_t h,
                     int mode, float* out, void* stream);
/* gr_group_norm: geotransformer/modules/kpconv/modules.py:32-50 GroupNorm.forward on the (N, C) matrix itself (the
 * reference transposes to (1, C, N) for nn.GroupNorm): statistics per group of C / groups channels over all n rows, biased
 * variance, y = (x - mean) / sqrt(var + eps) * gamma + beta, then LeakyReLU(negative_slope) -- pass 1.0f for none (the
 * blocks of modules.py:53-145 always follow the norm with LeakyReLU(0.1)).  gamma / beta (c) may be null.  C must be a
 * multiple of 4 with C / 4 dividing 256 or a multiple of 256. */
size_t gr_group_norm_workspace_bytes(int64_t groups);
int gr_group_norm(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta, float eps,
                  float negative_slope, float* out, void* ws, size_t ws_bytes, void* stream);
/* The same normalisation for several stack-mode batches at once: rows [seg_off[s], seg_off[s+1]) (device array, nseg + 1
 * entries) are normalised with their own statistics -- what modules.py:32-50 computes when every batch (scene pair) goes
 * through the network alone.  max_seg_rows = rows of the longest segment (sizes the grid). */
size_t gr_group_norm_seg_workspace_bytes(int64_t groups, int64_t nseg);
int gr_group_norm_seg(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta, float eps,
                      float negative_slope, float* out, const int64_t* seg_off, int64_t nseg, int64_t max_seg_rows, void* ws,
                      size_t ws_bytes, void* stream);
/* GroupNorm -> + residual -> LeakyReLU in one apply pass: the tail of a residual block (kpconv/modules.py:135-138).
 * seg_off null = one segment (gr_group_norm's workspace), else gr_group_norm_seg's arguments; residual (n,c) or null. */
int gr_group_norm_res(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta, float eps,
                      float negative_slope, const float* residual, float* out, const int64_t* seg_off, int64_t nseg,
                      int64_t max_seg_rows, void* ws, size_t ws_bytes, void* stream);
/* gr_gs_fuse ("next" row): gs_fusion.py:231-262 gaussian_fuse on the GS .ply wire format.
 * rec1 / rec2: device arrays of 62-float vertex records (gs_fusion.py:172-184 property order).  The host
 * passes the similarity transform split as the reference does (:237-240): h_rotation (3x3 row-major, scale
 * divided out), h_translation (3), h_scale, and the three SH band transforms (3x3, 5x5, 7x7 row-major,
 * new = old @ T).  out_rec has room for n1 + n2 records; *h_num_out = kept vertices.  Synchronises.
 * rec1 / rec2 must be 16-byte aligned (records are fetched as 16-byte vectors), out_rec 8-byte aligned. */
size_t gr_gs_fuse_workspace_bytes(int64_t n1, int64_t n2);
int gr_gs_fuse(const float* rec1, int64_t n1, const float* rec2, int64_t n2, const double* h_rotation,
               const double* h_translation, double h_scale, const float* h_sh_t1, const float* h_sh_t2,
               const float* h_sh_t3, float* out_rec, int64_t* h_num_out, void* ws, size_t ws_bytes, void* stream);
size_t gr_pairwise_distance_workspace_bytes(int64_t n, int64_t m);
int gr_pairwise_distance(const float* x, const float* y, int64_t n, int64_t m, int64_t c, int normalized,
                         float* out, void* ws, size_t ws_bytes, void* stream);
/* The same over a batch: x (batch, n, c), y (batch, m, c) -> out (batch, n, m) in ONE launch (grid.z = matrix); the
 * leading dimensions of pairwise_distance.py:4-31.  Outputs of at least ~192 tiles of 128 x 128 (over the whole batch) take
 * the 128 x 128-per-workgroup kernel, smaller ones 64 x 64 tiles; both return the same bits. */
size_t gr_pairwise_distance_batch_workspace_bytes(int64_t batch, int64_t n, int64_t m);
int gr_pairwise_distance_batch(const float* x, const float* y, int64_t batch, int64_t n, int64_t m, int64_t c, int normalized,
                               float* out, void* workspace, size_t workspace_bytes, void* stream);
size_t gr_superpoint_matching_workspace_bytes(int64_t nr, int64_t ns);
int gr_superpoint_matching(const float* ref_feats, const float* src_feats, int64_t nr, int64_t ns, int64_t c,
                           const uint8_t* ref_masks, const uint8_t* src_masks, int num_correspondences,
                           int dual_normalization, int64_t* out_ref_idx, int64_t* out_src_idx,
                           float* out_scores, int64_t* h_num_out, void* ws, size_t ws_bytes, void* stream);
/* Stack mode over `npairs` scene pairs (test.py:146-212 runs model.py:156-160 once per pair): superpoint features stacked as
 * [ref_0, src_0, ref_1, src_1, ...] with 2 npairs + 1 host offsets, masks likewise (null = all true).  Pair b's matches go
 * to row b of the (npairs, num_correspondences) outputs; h_num_out[b] = how many of them are valid.  All pairs are launched
 * back to back; ONE read-back (their headers) at the end. */
size_t gr_superpoint_matching_batch_workspace_bytes(const int64_t* h_node_off, int64_t npairs);
int gr_superpoint_matching_batch(const float* feats, const int64_t* h_node_off, int64_t npairs, int64_t c,
                                 const uint8_t* masks, int num_correspondences, int dual_normalization,
                                 int64_t* out_ref_idx, int64_t* out_src_idx, float* out_scores, int64_t* h_num_out,
                                 void* ws, size_t ws_bytes, void* stream);
size_t gr_point_matching_workspace_bytes(int64_t batch);
int gr_corr_matrix(const float* score_mat, int64_t batch, int64_t k1, int64_t k2,
                   const uint8_t* ref_knn_masks, const uint8_t* src_knn_masks, int k, int mutual,
                   float confidence_threshold, uint8_t* corr_mat, int64_t* h_num_corr, void* ws,
                   size_t ws_bytes, void* stream);
/* gr_corr_matrix_exp: the same operator for callers that already hold exp(score_mat) -- what
 * PointMatching.compute_correspondence_matrix / LocalGlobalRegistration.compute_correspondence_matrix receive
 * (point_matching.py:32-66 thresholds the matrix it is given; no log/exp round trip). */
int gr_corr_matrix_exp(const float* exp_score_mat, int64_t batch, int64_t k1, int64_t k2,
                       const uint8_t* ref_knn_masks, const uint8_t* src_knn_masks, int k, int mutual,
                       float confidence_threshold, uint8_t* corr_mat, int64_t* h_num_corr, void* ws,
                       size_t ws_bytes, void* stream);
int gr_corr_gather(const float* score_mat, int64_t batch, int64_t k1, int64_t k2, const uint8_t* corr_mat,
                   const float* ref_knn_points, const float* src_knn_points, const int64_t* ref_knn_indices,
                   const int64_t* src_knn_indices, const float* global_scores, int use_global_score,
                   float* out_ref_points, float* out_src_points, int64_t* out_ref_indices,
                   int64_t* out_src_indices, float* out_scores, void* ws, size_t ws_bytes, void* stream);
/* gr_lgr_register: geotransformer/modules/geotransformer/local_global_registration.py:135-193
 * (local_to_global_registration, correspondence_limit = None) + registration/procrustes.py:6-82, with no
 * host round trip.  Inputs are the outputs of gr_corr_gather (num_corr rows, torch.nonzero order) and `pm_ws`,
 * the workspace gr_corr_matrix / gr_corr_gather used (it holds the per-patch counts / offsets).
 * out_transform: 16 floats, row-major 4x4, on the device. */
size_t gr_lgr_workspace_bytes(int64_t batch);
/* gr_lgr_register_verify: the same with a separate verification set (correspondence_limit is not None,
 * local_global_registration.py:145-152): hypotheses are fitted on all correspondences of each patch, scored and
 * refined on the `num_verify` rows of verify_* (the top-`correspondence_limit` global scores). */
int gr_lgr_register_verify(const float* ref_corr_points, const float* src_corr_points, const float* corr_scores,
                           int64_t num_corr, int64_t batch, const void* pm_ws, const float* verify_ref_points,
                           const float* verify_src_points, const float* verify_scores, int64_t num_verify,
                           float acceptance_radius, int correspondence_threshold, int num_refinement_steps,
                           float* out_transform, void* ws, size_t ws_bytes, void* stream);
int gr_lgr_register(const float* ref_corr_points, const float* src_corr_points, const float* corr_scores,
                    int64_t num_corr, int64_t batch, const void* pm_ws, float acceptance_radius,
                    int correspondence_threshold, int num_refinement_steps, float* out_transform, void* ws,
                    size_t ws_bytes, void* stream);
/* Stack mode over `nseg` scene pairs: the `batch` patches (gr_corr_matrix / gr_corr_gather over ALL patches of the batch)
 * belong to pairs, pair s owning patches [seg_patch_off[s], seg_patch_off[s+1]) (DEVICE int32, nseg + 1 entries).
 * out_transforms: nseg x 16 floats on the device; out_seg_rows (optional, device int32[nseg + 1]): first correspondence row
 * of every pair, last entry = num_corr.  A pair without correspondences gets the identity.  Three launches, no host
 * synchronisation. */
int gr_lgr_register_seg(const float* ref_corr_points, const float* src_corr_points, const float* corr_scores,
                        int64_t num_corr, int64_t batch, const void* pm_ws, const int32_t* seg_patch_off, int64_t nseg,
                        float acceptance_radius, int correspondence_threshold, int num_refinement_steps,
                        float* out_transforms, int32_t* out_seg_rows, void* ws, size_t ws_bytes, void* stream);
/* gr_ransac_similarity ("next" row; PARITY UNPINNED -- Open3D is not in the reference tree):
 * stands in for geotransformer/utils/open3d.py:169-198 (registration_ransac_based_on_correspondence with
 * TransformationEstimationPointToPoint(with_scaling)), called from model.py:209-215.  Row i of src_points
 * corresponds to row i of ref_points.  out_transform: 16 floats (4x4 row-major, scale folded into the 3x3
 * block) on the device; out_stats (optional, device int32[2]) = {inliers of the best hypothesis, its id}.
 * gr_ransac_sample_hash exposes the counter hash the sampler uses (index = hash % num_corr, retried with
 * attempt+1 on duplicates) so a checker can replay the same hypotheses. */
uint32_t gr_ransac_sample_hash(uint32_t seed, uint32_t hypothesis, uint32_t k, uint32_t attempt);
size_t gr_ransac_workspace_bytes(int64_t num_hypotheses);
int gr_ransac_similarity(const float* src_points, const float* ref_points, int64_t num_corr, int ransac_n,
                         int64_t num_hypotheses, uint32_t seed, float distance_threshold, int with_scaling,
                         int refine, float* out_transform, int32_t* out_stats, void* ws, size_t ws_bytes,
                         void* stream);
/* Stack mode over `nseg` scene pairs: pair s owns the correspondence rows [seg_row_off[s], seg_row_off[s+1]) (DEVICE int32,
 * nseg + 1 entries, e.g. gr_lgr_register_seg's out_seg_rows) and draws its hypotheses with seed + s.  A pair with fewer than
 * ransac_n correspondences keeps fallback_transforms[s] (nseg x 16, null = identity) -- model.py:209-220.  Three launches,
 * no host synchronisation. */
size_t gr_ransac_seg_workspace_bytes(int64_t num_hypotheses, int64_t nseg);
int gr_ransac_similarity_seg(const float* src_points, const float* ref_points, const int32_t* seg_row_off, int64_t nseg,
                             int ransac_n, int64_t num_hypotheses, uint32_t seed, float distance_threshold,
                             int with_scaling, int refine, const float* fallback_transforms, float* out_transforms,
                             int32_t* out_stats, void* ws, size_t ws_bytes, void* stream);
/* gr_fps ("next" row; PARITY UNPINNED -- fpsample is not in the reference tree): exact farthest
 * point sampling in stack mode, stands in for fpsample.bucket_fps_kdline_sampling (demo.py:46, test.py:46,
 * dataset.py:127).  points (n,3) hold `batch` clouds (h_lengths); cloud b yields h_num_samples[b] LOCAL indices
 * starting with h_start_indices[b] (null = 0), concatenated in out_indices (int64).  Synchronises. */
/* gr_geo_embedding ("next" row): GeometricStructureEmbedding.forward for one cloud
 * (geotransformer/modules/geotransformer/geotransformer.py:26-73 + transformer/positional_embedding.py:8-34):
 * out (n,n,c) = proj_d(sinemb(|p_a-p_b| / sigma_d)) + reduce_k proj_a(sinemb(angle(p_knn(a,k)-p_a, p_b-p_a) * factor_a)).
 * w_d, w_a: (c,c) nn.Linear weights (row = output channel); b_d, b_a: (c); div_term: (c/2) buffer of the sinusoidal
 * embedding; reduction_mean: bit 0 = 'mean' instead of 'max', bit 1 = run the projections on fp32 MFMAs (c % 32 == 0)
 * instead of the default split-bf16 scheme (three bf16 parts per operand, six bf16 MFMAs per product, fp32 accumulate:
 * fp32 accuracy up to summation order; c % 16 == 0, c <= 512); angle_k <= 8.  Asynchronous on `stream`. */
size_t gr_geo_embedding_workspace_bytes(int64_t n, int64_t angle_k);
int gr_geo_embedding(const float* points, int64_t n, const float* w_d, const float* b_d, const float* w_a,
                     const float* b_a, const float* div_term, int64_t c, float sigma_d, float factor_a,
                     int64_t angle_k, int reduction_mean, float* out, void* ws, size_t ws_bytes, void* stream);
/* The same embedding from two function tables.  Both projections act on the sinusoidal embedding of ONE scalar, so
 * F_d(x) = W_d phi(x) + b_d and F_a(x) = W_a phi(x) + b_a are smooth maps R -> R^c: the caller tabulates them once per set
 * of weights on a uniform grid (tab[j] = F((j - 1) / inv_h), rows x c fp32, 16-byte aligned) and the kernel evaluates them
 * by 4-point Lagrange interpolation (error ~ 2e-8 at inv_h = 32) instead of 2 (1 + k) n^2 c^2 flop of GEMM; an index beyond
 * a table is evaluated directly from the weights.  Same result as gr_geo_embedding within fp32 rounding. */
int gr_geo_embedding_table(const float* points, int64_t n, const float* tab_d, int64_t rows_d, const float* tab_a,
                           int64_t rows_a, float inv_h, const float* w_d, const float* b_d, const float* w_a, const float* b_a,
                           const float* div_term, int64_t c, float sigma_d, float factor_a, int64_t angle_k, int reduction_mean,
                           float* out, void* ws, size_t ws_bytes, void* stream);
/* gr_rpe_scores: positional score term of RPEMultiHeadAttention (geotransformer/modules/transformer/
 * rpe_transformer.py:55-57) re-associated so the (N,M,C) embedding is read once and never projected:
 * out[h][n][m] = sum_j embed[n][m][j] * u[n][h][j] + add[n][h], with u = q W_p (per head) and add = q . b_p computed
 * by the caller.  embed (n,m,c), u (n,heads,c), add (n,heads) or null, out (heads,n,m).  c in {64,128,256}. */
int gr_rpe_scores(const float* embed, const float* u, const float* add, int64_t n, int64_t m, int64_t c,
                  int64_t heads, float* out, void* stream);
/* gr_rpe_attention: the whole RPEMultiHeadAttention row pipeline of rpe_transformer.py:51-72 after the four input
 * projections, fused per query row: scores = (q k^T + embed . u + add) / sqrt(c / heads), then attention_factors (n,m),
 * key_weights (m), key_masks (m, uint8, nonzero = masked to -inf) -- each optional (null), applied in the reference's
 * order -- softmax over the keys, and hidden = scores @ v.  q (n,c), k / v (m,c) are the projected tensors in their
 * natural (rows, heads*channels) layout; embed (n,m,c); u (n,heads,c) and add (n,heads) as for gr_rpe_scores.
 * Outputs: out_scores (heads,n,m) = the attention probabilities, out_hidden (n,c). */
int gr_rpe_attention(const float* embed, const float* u, const float* add, const float* q, const float* k, const float* v,
                     const float* attention_factors, const float* key_weights, const uint8_t* key_masks, int64_t n,
                     int64_t m, int64_t c, int64_t heads, float* out_scores, float* out_hidden, void* stream);
size_t gr_fps_workspace_bytes(int64_t n, int64_t batch);
int gr_fps(const float* points, const int64_t* h_lengths, const int64_t* h_num_samples,
           const int64_t* h_start_indices, int64_t n, int64_t batch, int64_t* out_indices, void* ws,
           size_t ws_bytes, void* stream);
size_t gr_point_to_node_workspace_bytes(int64_t n, int64_t m);
int gr_point_to_node_partition(const float* points, int64_t n, const float* nodes, int64_t m, int point_limit,
                               int64_t* point_to_node, uint8_t* node_masks, int64_t* node_knn_indices,
                               uint8_t* node_knn_masks, void* ws, size_t ws_bytes, void* stream);
/* Stack mode over `nclouds` (fine cloud, coarse nodes) pairs -- model.py:99-104 for every cloud of a batch of scene pairs
 * (test.py:146-212) in one call: h_point_off / h_node_off hold nclouds + 1 ascending host offsets into points / nodes;
 * outputs are the single-cloud outputs concatenated (node_knn_* at row h_node_off[c]); indices stay LOCAL to their cloud
 * (padding value = that cloud's point count).  Asynchronous on `stream`, no host synchronisation. */
size_t gr_point_to_node_batch_workspace_bytes(const int64_t* h_point_off, const int64_t* h_node_off, int64_t nclouds);
int gr_point_to_node_partition_batch(const float* points, const int64_t* h_point_off, const float* nodes,
                                     const int64_t* h_node_off, int64_t nclouds, int point_limit,
                                     int64_t* point_to_node, uint8_t* node_masks, int64_t* node_knn_indices,
                                     uint8_t* node_knn_masks, void* ws, size_t ws_bytes, void* stream);

/* Test switch for gr_fps: 0 = normal; 1 = the co-operative launch counts as refused; 2 = its result is discarded as if the
 * inter-workgroup exchange had timed out.  Both force the one-workgroup-per-cloud retry (same indices).  Returns the old mode;
 * a negative argument only queries. */
int gr_fps_debug_force_fallback(int mode);
/* Test switch: 0 = the curve pre-pass of gr_fps always takes the radix sort (default 1: the bucket sort of depth_sort.hip where
 * the batch allows it; a bucket that does not fit repeats the order on the radix sort).  Returns the previous value; the
 * sample sets do not depend on it. */
int gr_fps_debug_bucket_sort(int on);

/* ------------------------------------------------------------------------------------------------
 * Harness support, NOT a reference interface: stand-in position descriptors of the configs[4] pair pipeline
 * (gaussreg_amd/pair_pipeline.py; the learned features are not available offline).  out (n, c) =
 * mask * scale * cos((T[transform_id] p) W + b), rows optionally L2-normalised; transforms (k, 3, 4) row-major or NULL,
 * transform_id (n) int32 (-1 = identity) or NULL, mask (n) uint8 or NULL, w (3, c), b (c). */
int gr_standin_descriptors(const float* pts, int64_t n, const float* transforms, const int32_t* transform_id,
                           const uint8_t* mask, const float* w, const float* b, int64_t c, float scale, int normalize,
                           float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GAUSSREG_HIP_H_ */
