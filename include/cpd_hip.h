/*
 * cpd_hip.h -- C-ABI of libcpd_hip.so: the MI355X (gfx950) implementation of CPD's detection
 * hot path  voxelize -> sparse 3D conv backbone -> BEV dense head -> rotated NMS.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types. Every DATA pointer is
 * a DEVICE pointer unless its comment says HOST; small geometry arrays (voxel size, ranges,
 * shapes, kernel/stride/pad triples) are HOST arrays read at call time. All functions enqueue on
 * `stream` (a hipStream_t passed as void*) and return without synchronising unless stated. They
 * never allocate device memory and never call exit(): workspaces are caller-owned (size queries
 * below) so a whole frame can be captured into a hipGraph. Return value: CPD_OK (0) or a
 * negative CPD_ERR_* code (the reference's exit(-1) CHECK_INPUT macros, iou3d_nms.cpp:14-26,
 * become error codes; the Python shim turns them into exceptions).
 *
 * Each entry point cites the reference interface it replaces (path:line under /root/reference;
 * [SPCONV] = behaviour of the un-vendored dependency spconv-cu111==2.1.22 as used at that call
 * site). INTEGRATION.md shows the reference-side binding for each.
 */
#ifndef CPD_HIP_H
#define CPD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *cpd_stream_t; /* hipStream_t; NULL = the default stream, as the reference uses */

#define CPD_OK 0
#define CPD_ERR_ARG (-1)         /* bad argument (null pointer, non-positive size, bad shape) */
#define CPD_ERR_WORKSPACE (-2)   /* caller workspace smaller than the *_bytes() query         */
#define CPD_ERR_LAUNCH (-3)      /* HIP runtime reported an error; see cpd_last_hip_error()   */
#define CPD_ERR_UNSUPPORTED (-4) /* size outside the supported envelope (e.g. >2^31 cells)    */

const char *cpd_version(void);
/* Diagnostics: with the log enabled every conv / weight-gradient call notes the kernel instantiation it launched
 * ("rowwave_conv_f16s_kernel<64,2>", "wgrad_f16_kernel<128,128>", ...). dump writes "name count" lines and returns the bytes
 * needed; enable(…) also clears the log. Host only; tests use it to assert which kernels a full-size step runs.            */
void cpd_launch_log_enable(int on);
void cpd_launch_log_note(const char *kernel);
size_t cpd_launch_log_dump(char *buf, size_t cap);
int cpd_last_hip_error(void); /* hipError_t of the last CPD_ERR_LAUNCH on this thread */

/* ===== B1. Voxelizer (+ fused MeanVFE) =====================================================
 * Replaces VoxelGeneratorWrapper.generate -> [SPCONV] Point2VoxelCPU3d.point_to_voxel
 * (cpd/datasets/processor/data_processor.py:14-59, driver l.128-183) and MeanVFE.forward
 * (cpd/models/backbones_3d/vfe/mean_vfe.py:41-43).
 * Semantics: serial first-appearance voxel order, first `max_points` points of a voxel in
 * point order, fp32 floor((p-lo)/vs), upper bound exclusive, voxel cap `max_voxels`.
 */
/* grid = round((hi-lo)/vs) in fp32, returned (z,y,x). HOST in, HOST out. */
int cpd_voxel_grid_size(const float vsize_xyz[3], const float range_xyz[6], int32_t grid_zyx[3]);
size_t cpd_voxelize_workspace_bytes(int n_points, int max_points, int max_voxels,
                                    const float vsize_xyz[3], const float range_xyz[6]);
/* points [n_points, c] f32 (x,y,z,...). Outputs sized for cap = min(max_voxels, n_points) rows:
 *   voxels        [cap, max_points, c] f32, zero padded            (may be NULL)
 *   coords        [cap, coord_cols] i32; coord_cols 3 -> (z,y,x), 4 -> (batch_idx,z,y,x)
 *                 (the pad of collate_batch, cpd/datasets/dataset.py:264)
 *   num_points    [cap] i32
 *   mean_features [cap, c] f32 = sum_p voxels / max(num,1)         (may be NULL)
 *   n_voxels      device i32 scalar                                                         */
int cpd_voxelize(const float *points, int n_points, int c, const float vsize_xyz[3],
                 const float range_xyz[6], int max_points, int max_voxels, int batch_idx,
                 int coord_cols, float *voxels, int32_t *coords, int32_t *num_points,
                 float *mean_features, int32_t *n_voxels, void *workspace, size_t workspace_bytes,
                 cpd_stream_t stream);

/* All frames of a batch in one set of launches. `points` [frame_offsets[n_frames], c] holds the
 * frames back to back; `frame_offsets` is a HOST array [n_frames + 1] (frame f = rows
 * [off[f], off[f+1])), n_frames <= 64. Per frame the result is exactly cpd_voxelize's (same
 * first-appearance order, max_voxels cap, max_points per voxel); rows of frame f follow those of
 * frame f-1, coords are always (b,z,y,x) with b = f. n_voxels: device int32 [n_frames + 1] =
 * voxels per frame, then their sum. Outputs need min(n_frames*max_voxels, n_total) rows. */
size_t cpd_voxelize_batch_workspace_bytes(int n_total, int n_frames, int max_points_per_voxel,
                                          int max_voxels, const float vsize_xyz[3],
                                          const float range_xyz[6]);
int cpd_voxelize_batch(const float *points, const int32_t *frame_offsets, int n_frames, int c,
                       const float vsize_xyz[3], const float range_xyz[6], int max_points_per_voxel,
                       int max_voxels, float *voxels, int32_t *coords, int32_t *num_points,
                       float *mean_features, int32_t *n_voxels, void *workspace, size_t workspace_bytes,
                       cpd_stream_t stream);
/* cpd_voxelize_batch that builds its occupancy bitmap, popcount prefix and rank -> row map INSIDE the caller's site index
 * (cpd_index_bytes(n_frames, {grid_z + z_extra, grid_y, grid_x}, n_total) bytes) instead of its own workspace: the index is
 * then the level-0 site index of the voxel list (what cpd_index_build would produce from `coords`) at no extra cost.
 * z_extra = 1 gives the backbone's sparse_shape = grid_size[::-1] + [1,0,0] (spconv_backbone.py:412). Voxels beyond a
 * frame's max_voxels cap are not sites (their lookups return -1). */
int cpd_voxelize_batch_index(const float *points, const int32_t *frame_offsets, int n_frames, int c,
                             const float vsize_xyz[3], const float range_xyz[6], int max_points_per_voxel,
                             int max_voxels, float *voxels, int32_t *coords, int32_t *num_points,
                             float *mean_features, int32_t *n_voxels, void *workspace, size_t workspace_bytes,
                             void *index, size_t index_bytes, int z_extra, cpd_stream_t stream);
/* cpd_voxelize_batch_index with the voxel rows in CANONICAL order -- ascending (frame, z, y, x) = the rank of the voxel's cell in
 * the occupancy bitmap -- instead of first appearance. Same voxels, same points per voxel (the max_points smallest point indices),
 * same means; only the row order differs, the site index left behind is canonical (row = rank, no map) and the first-appearance
 * scan over the points is not run. The order at the B1 boundary (Point2VoxelCPU3d) is first appearance: this form is for hosts
 * that consume the rows themselves (CenterPointEngine). The max_voxels cap is defined on first-appearance order and is NOT
 * applied here: n_voxels [n_frames + 1] reports the true counts, a caller that sees a frame above its cap falls back to
 * cpd_voxelize_batch_index. Outputs need min(n_frames * max_voxels, n_total) rows as there (a frame above the cap may exceed
 * that: check the workspace-sized capacity n_total instead when caps can bind). */
int cpd_voxelize_batch_canonical(const float *points, const int32_t *frame_offsets, int n_frames, int c,
                                 const float vsize_xyz[3], const float range_xyz[6], int max_points_per_voxel,
                                 int max_voxels, float *voxels, int32_t *coords, int32_t *num_points,
                                 float *mean_features, int32_t *n_voxels, void *workspace, size_t workspace_bytes,
                                 void *index, size_t index_bytes, int z_extra, cpd_stream_t stream);
/* cpd_voxelize_batch / _index / _canonical for frames that sit in SEPARATE device allocations (clouds arrive one by one from the
 * dataloader, data_processor.py:43-59 is called per frame): frame_points = HOST array of n_frames device pointers, frame f holding
 * frame_offsets[f + 1] - frame_offsets[f] rows of c floats. No concatenated copy of the batch's points is needed (153 MB at 48 frames).
 * index = NULL: cpd_voxelize_batch; index != NULL: cpd_voxelize_batch_index (canonical = 0) or cpd_voxelize_batch_canonical (1). */
int cpd_voxelize_batch_frames(const float *const *frame_points, const int32_t *frame_offsets, int n_frames, int c,
                              const float vsize_xyz[3], const float range_xyz[6], int max_points, int max_voxels,
                              float *voxels, int32_t *coords, int32_t *num_points, float *mean_features,
                              int32_t *n_voxels, void *workspace, size_t workspace_bytes, void *index,
                              size_t index_bytes, int z_extra, int canonical, cpd_stream_t stream);

/* ===== B2. Sparse convolution ==============================================================
 * Replaces [SPCONV] SparseConvTensor / SubMConv3d / SparseConv3d / .dense() as called from
 * cpd/models/backbones_3d/spconv_backbone.py:17-21,108-115,414-455,524-529 and
 * cpd/models/backbones_2d/map_to_bev/height_compression.py:136-138.
 *
 * Site index: a dense occupancy bitmap over (batch, z, y, x) plus a popcount prefix, giving
 * coordinate -> row lookup and, for free, the canonical ascending (b,z,y,x) order of a site set.
 * Rulebook: output-stationary neighbour table nbr[kv][n_out] (i32, -1 = no input), tap index
 * t = (tz*kH + ty)*kW + tx as in the spconv-2.x weight layout (Cout,kD,kH,kW,Cin). The same
 * table is the `indice_dict` entry a SparseConvTensor caches per indice_key.
 */
size_t cpd_index_bytes(int batch, const int32_t shape_zyx[3], int n_capacity);
/* Build the index of `indices` [n,4] i32 (b,z,y,x) (any order; row ids = positions). */
int cpd_index_build(const int32_t *indices, int n, int batch, const int32_t shape_zyx[3],
                    void *index, size_t index_bytes, cpd_stream_t stream);
/* Row order of a level. cpd_order_rows_by_taps: for the canonical-order site list `indices` [n][4] of a level and its site
 * index, a permutation that sorts the rows of every chunk of `chunk_rows` consecutive rows (1024, 4096, 8192 or 16384) by their
 * neighbour pattern under the sub-manifold kernel `ksize` (<= 32 taps): new_to_old [n], old_to_new [n], and (optional)
 * indices_out [n][4] = the site list in the new order. Rows ordered this way make cpd_gather_conv's 16-row tap skipping nearly
 * exact (executed / useful MFMAs 1.3-1.55 -> 1.06-1.12 on Waymo-shape levels) at unchanged cache locality. workspace: n * 4
 * bytes. cpd_index_set_order installs a caller-owned rank -> row map (old_to_new; NULL = canonical again) in a site index:
 * every lookup through that index (cpd_rulebook_subm / _conv, cpd_voxel_query_index) then returns rows in the new order, so a
 * level is re-ordered by passing its site list in the new order and installing the map -- no kernel sees a difference. The map
 * must outlive the index's use. */
int cpd_order_rows_by_taps(const int32_t *indices, int n, int batch, const int32_t shape_zyx[3],
                           const int32_t ksize[3], const void *index, int chunk_rows, int32_t *new_to_old,
                           int32_t *old_to_new, int32_t *indices_out, void *workspace, size_t workspace_bytes,
                           cpd_stream_t stream);
int cpd_index_set_order(void *index, const int32_t *rank_to_row, cpd_stream_t stream);
/* Brick order of a level (round 4; any row order is a valid SparseConvTensor, spconv_backbone.py:502-558 never reads one): rows
 * sorted by (b, z, y / brick_y, x / brick_x, y, x) -- brick_y x brick_x bricks of one z-plane, computed WITHOUT a sort from rank
 * queries on the level's canonical site index -- and then, inside every tile of `tile_rows` (128 or 256) consecutive rows of that order, by their 27-bit
 * sub-manifold neighbour pattern (what cpd_order_rows_by_taps does per chunk). A tile of 128 output rows then touches ~2.9 distinct
 * input rows per row instead of 4.5 (canonical) or 8.8 (4096-row pattern chunks): what cpd_gather_conv_planned stages in LDS.
 * `indices` [n][4]: the CANONICAL list of the level, `index`: its canonical index. Outputs as cpd_order_rows_by_taps.
 * workspace: 2 * align256(4 n) + align256(16 n) bytes. */
int cpd_order_rows_bricks(const int32_t *indices, int n, int batch, const int32_t shape_zyx[3], const void *index,
                          int brick_y, int brick_x, int tile_rows, int32_t *new_to_old, int32_t *old_to_new, int32_t *indices_out,
                          void *workspace, size_t workspace_bytes, cpd_stream_t stream);
/* SubMConv3d rulebook: output set == input set, same order; tap t reads coord + t - k/2.
 * tapmask (optional, u32 [ceil(n/16)], kernel volume <= 32): bit t of word s is set iff some row
 * of the 16-row group s has a neighbour at tap t -- lets cpd_gather_conv skip empty
 * (row group, tap) pairs without touching nbr. */
int cpd_rulebook_subm(const int32_t *indices, int n, int batch, const int32_t shape_zyx[3],
                      const int32_t ksize[3], const void *index, int32_t *nbr, uint32_t *tapmask,
                      cpd_stream_t stream);
/* Row plan of a 3 x 3 x 3 sub-manifold rulebook nbr[27][n_out] (round 4): per tile of 128 consecutive output rows and per dz group of
 * nine taps (tap = (dz * 3 + dy) * 3 + dx; the three groups read three different z-planes and share no input row)
 * (tile_rows = 128 or 256: the rows of one workgroup of cpd_gather_conv_planned)
 *   ulist [tiles][3][9 * tile_rows] i32   the distinct input rows the group touches (any order), `count` of them;
 *   slots [tiles][27][tile_rows] u16   per (tap, row of the tile): position of nbr[tap][row] in its group's list, 0xffff = no neighbour;
 *   count [tiles][4]       i32   the three list lengths and their sum.
 * cpd_rulebook_plan_bytes(n_out, which): bytes of slots (0), ulist (1), count (2). Any row order is planned correctly. */
size_t cpd_rulebook_plan_bytes(int n_out, int tile_rows, int which);
int cpd_rulebook_plan(const int32_t *nbr, int kv, int n_out, int tile_rows, uint16_t *slots, int32_t *ulist, int32_t *count,
                      cpd_stream_t stream);
/* cpd_rulebook_subm / cpd_rulebook_conv for a CHUNK-ORDERED output level (cpd_order_rows_by_taps, chunk_rows = 4096): out_canonical
 * [n_out][4] = the level's canonical site list, out_old_to_new [n_out] = its order (NULL = canonical). Gives, bit for bit, the table and
 * tap masks the plain builders give over the re-ordered list, but walks every chunk in canonical order (neighbouring lanes read the
 * same bitmap words) and re-orders it in LDS: a tap-ordered level cost the plain builder 2.2x a canonical one. 3 x 3 x 3 kernels. */
int cpd_rulebook_chunk_ordered(const int32_t *out_canonical, const int32_t *out_old_to_new, int n_out, int batch,
                               const int32_t in_shape[3], const int32_t ksize[3], const int32_t stride[3],
                               const int32_t pad[3], const void *in_index, int chunk_rows, int32_t *nbr, uint32_t *tapmask,
                               cpd_stream_t stream);
/* out_shape = (in + 2*pad - k)/stride + 1. HOST only. */
int cpd_conv_out_shape(const int32_t in_shape[3], const int32_t ksize[3], const int32_t stride[3],
                       const int32_t pad[3], int32_t out_shape[3]);
/* SparseConv3d output set: marks every output coordinate reached by an active input, builds the
 * OUTPUT site index in `out_index` (sized by cpd_index_bytes(batch, out_shape, capacity)) and
 * writes the number of active outputs to the device scalar n_out.                           */
int cpd_conv_outset(const int32_t *in_indices, int n_in, int batch, const int32_t in_shape[3],
                    const int32_t ksize[3], const int32_t stride[3], const int32_t pad[3],
                    void *out_index, size_t out_index_bytes, int32_t *n_out, cpd_stream_t stream);
/* Write the index's sites as [n,4] i32 rows in canonical ascending (b,z,y,x) order. */
int cpd_index_emit(const void *index, int batch, const int32_t shape_zyx[3], int32_t *indices,
                   int n_capacity, cpd_stream_t stream);
/* SparseConv3d rulebook: nbr[t][o] = input row at o*stride - pad + t. */
int cpd_rulebook_conv(const int32_t *out_indices, int n_out, int batch, const int32_t in_shape[3],
                      const int32_t ksize[3], const int32_t stride[3], const int32_t pad[3],
                      const void *in_index, int32_t *nbr, uint32_t *tapmask, cpd_stream_t stream);

/* Weights for cpd_gather_conv: pack a dense [kv][c_in][c_out] f32 tensor (device) into the
 * MFMA-fragment order the kernel streams (zero padded to multiples of 16). One buffer holds every image the conv may run on: the
 * fp32 image; for c_in % 32 == 0 the split-bf16 and split-fp16 images (the latter pre-scaled per output column by a power of two,
 * followed by the descale factors); for c_in == 16 the K = 16 split-fp16 image Ph16[t][g][n][hi 4 | lo 4] + its descale factors
 * (round 3: 16-channel pair rows, CPD_GC_IN_PAIRS). cpd_packed_weight_floats sizes the buffer; every packer (this one, the adjoint
 * one, the batched one) writes every image the buffer has, bit for bit alike.                                                   */
size_t cpd_packed_weight_floats(int kv, int c_in, int c_out);
int cpd_pack_weight(const float *w_kio, int kv, int c_in, int c_out, float *packed,
                    cpd_stream_t stream);
/* The one GEMM-shaped kernel of the path (fp32 MFMA, exact fp32 fma chain):
 *   out[j, :] = act( (sum_t W[t]^T . in[nbr[t][j], :]) * scale + shift + residual[j, :] )
 * Used for SubMConv3d / SparseConv3d (rulebook from above, bias/eval-BatchNorm1d folded into
 * scale/shift, SparseBasicBlock residual + ReLU fused: spconv_backbone.py:100-136) and, with a
 * dense pixel rulebook, for every Conv2d/ConvTranspose2d(+BN+ReLU) of BaseBEVBackbone
 * (cpd/models/backbones_2d/base_bev_backbone.py:31-59) and CenterHead
 * (cpd/models/dense_heads/center_head.py:21-27,73-80) on channels-last maps.
 *   in   [n_in rows, in_ld floats per row], first c_in floats of a row are the features
 *   nbr  [kv][n_out] or NULL (kv must be 1: identity, i.e. a 1x1 conv / linear layer)
 *   tapmask: the rulebook's tap masks (see cpd_rulebook_subm) or NULL (every tap computed)
 *   scale, shift [c_out] or NULL (1 / 0); residual [n_out, res_ld] or NULL; relu 0/1
 *   out  [n_out, out_ld]; only columns [0, c_out) of each row are written
 *   out_row_map: NULL, or i32 destination rows. With out_col_group == 0 it is [n_out]: row j
 *                is written to row out_row_map[j]. With out_col_group = G > 0 the c_out columns
 *                are G-wide groups: column c of row j goes to row out_row_map[(c/G)*n_out + j],
 *                column c % G -- one launch computes the k*k taps of ConvTranspose2d(k, s=k)
 *                (base_bev_backbone.py:52-56) and interleaves them into the upsampled map.
 *   flags: 0 or CPD_GC_DENSE (a performance hint only; results are identical).               */
int cpd_gather_conv(const float *in, int in_ld, int n_in, int c_in, const float *packed_w,
                    const int32_t *nbr, const uint32_t *tapmask, int kv, int n_out, int c_out, const float *scale,
                    const float *shift, const float *residual, int res_ld, int relu, float *out,
                    int out_ld, const int32_t *out_row_map, int out_col_group, int flags,
                    cpd_stream_t stream);
#define CPD_GC_DENSE 1
/* allow the split-bf16 matrix path: fp32 operands split exactly into 3 bf16 terms, 6 partial
 * products accumulated in fp32 (error <= 2^-22 relative per product, i.e. fp32-level); used for
 * dense layers with c_in % 32 == 0 and c_out % 64 == 0 */
#define CPD_GC_BF16X3 2 /* (CPD_GC_DENSE: the rulebook has (almost) no -1 entries -- prefer the LDS-tiled
                          workgroup kernel over the tap-skipping wave kernel) */
/* allow the split-fp16 matrix path instead: fp32 operands written as h + l, two fp16 terms (2 x 11 bits + sign: x to
 * 2^-24 relative, or 2^-25 absolute below 0.5), 3 partial products (hh, hl, lh) accumulated in fp32 -- fp32-level error at
 * half the matrix work of CPD_GC_BF16X3. fp16's range: an input of magnitude >= 65504 gives inf / NaN (0 behind a ReLU
 * epilogue) UNLESS the call comes with its input's absmax block (cpd_gather_conv_ranged / cpd_conv3x3_rows_ranged below: the
 * input is then pre-scaled by a power of two, exact at any magnitude); weights of any magnitude (pre-scaled per output column
 * by a power of two at pack time, undone exactly in the epilogue). Takes precedence over CPD_GC_BF16X3 when both are set. */
#define CPD_GC_F16X2 4
/* fp16-pair rows (engine-internal storage between f16x2 sparse layers; the boundary tensors stay fp32): a row keeps its 4 * C
 * bytes, but every 32-channel block holds the fp16 HIGH terms of its channels (64 B, natural order) followed by the fp16 LOW terms
 * (64 B) -- exactly the split x = h + l the f16x2 kernels make of a gathered fp32 row, made ONCE by the epilogue that produced
 * the row instead of by each of the <= 27 gathers of it. The partial products are those of fp32 rows (the accumulation order inside
 * a 32-channel block differs: fp32 rounding); a residual read back from pairs is h + l (x to 2^-24 relative). Same range as CPD_GC_F16X2 without an absmax block: |x| < 65504
 * (the engine's range guard re-runs such a step on fp32 rows with pre-scaling).
 *   CPD_GC_IN_PAIRS   `in` rows are pairs: needs CPD_GC_F16X2, c_in % 32 == 0, c_out % 32 == 0, kv <= 28, < 4 GB of input, no
 *                     in_absmax, not CPD_GC_DENSE (the row-wave kernel is the one that reads them) -- else CPD_ERR_UNSUPPORTED
 *                     -- or c_in == 16 (c_out 16 or 32; the wave kernel's K = 16 MFMA form): 16-channel pair rows hold, per group of
 *                     four channels, the four high terms (8 B) then the four low terms (8 B)
 *   CPD_GC_OUT_PAIRS  `out` rows are written as pairs (the sparse kernels' epilogues; c_out % 32 == 0 or c_out == 16, no out_col_group)
 *   CPD_GC_RES_PAIRS  `residual` rows are pairs (c_out % 32 == 0 or c_out == 16)
 * Round 5 -- DENSE pair maps (the Conv2d / ConvTranspose2d layers of BaseBEVBackbone and CenterHead, base_bev_backbone.py:31-59,
 * center_head.py:11-45,73-94, between two split-fp16 layers): CPD_GC_DENSE | CPD_GC_F16X2 | CPD_GC_IN_PAIRS | CPD_GC_OUT_PAIRS on
 * cpd_gather_conv* runs the 128 x 128 pair tile kernel (strided conv through its pixel table, 1 x 1 GEMM, ConvTranspose(k = s) through
 * out_row_map / out_col_group with groups of a multiple of 32 columns): c_in % 32 == 0, c_out % 128 == 0, pairs in AND out, no residual,
 * no in_absmax, < 4 GB of input; on cpd_conv3x3_rows* (below) the window kernel's 128 x 128 and 256 x 64 tiles (pairs in and out) and
 * its 256 x 16 tile (pairs in, fp32 rows out: c_out <= 16) -- the tiles cpd_conv3x3_rows_tile reports for the problem. Any other
 * shape: CPD_ERR_UNSUPPORTED (nothing else reads dense pair rows; keep that map fp32). */
#define CPD_GC_IN_PAIRS 16
#define CPD_GC_OUT_PAIRS 32
#define CPD_GC_RES_PAIRS 64

/* 3x3 / stride 1 / pad 1 convolution (+ folded BN / bias, residual, ReLU: same epilogue as
 * cpd_gather_conv) over channels-last pixel rows in[frames*h*w][c_in] WITHOUT a rulebook -- the
 * nn.Conv2d(3, padding=1) layers of BaseBEVBackbone (base_bev_backbone.py:38-62) and of
 * CenterHead / SeparateHead (center_head.py:24-52,78-90). The input row of output row R at tap
 * (dy, dx) is R + dy*w + dx when that pixel exists, so the three dx taps share one gathered and
 * split 130-row window per (32-channel block, dy). packed_w = cpd_pack_weight image of the
 * [9][c_in][c_out] weights (tap = ky*3 + kx). Split-bf16 arithmetic only: returns
 * CPD_ERR_UNSUPPORTED unless flags has CPD_GC_BF16X3, c_in % 32 == 0, c_out % 64 == 0, `in` rows are
 * 16-byte aligned and the problem fills the chip (cpd_conv3x3_rows_supported tells, pointers aside);
 * the caller then uses cpd_rulebook_conv2d + cpd_gather_conv, which computes the same thing.     */
int cpd_conv3x3_rows_supported(int frames, int h, int w, int c_in, int c_out, int flags);
/* Introspection: the (rows x columns) workgroup tile cpd_conv3x3_rows runs for this problem. HOST only. */
int cpd_conv3x3_rows_tile(int frames, int h, int w, int c_in, int c_out, int flags, int *bm, int *bn);
int cpd_conv3x3_rows(const float *in, int in_ld, int frames, int h, int w, int c_in,
                     const float *packed_w, int c_out, const float *scale, const float *shift,
                     const float *residual, int res_ld, int relu, float *out, int out_ld, int flags,
                     cpd_stream_t stream);
/* cpd_gather_conv for a 3 x 3 x 3 SubMConv3d layer (spconv_backbone.py:108-115: the SparseBasicBlock convs) whose rulebook comes with
 * a row plan (cpd_rulebook_plan): the STAGED row-wave kernel. Per tile of 128 output rows and dz group it copies the group's distinct
 * input rows (128-byte channel blocks, whole cache lines) into an LDS window once and forms the MFMA fragments of all nine taps from
 * LDS -- a (row, tap) pair costs a ds_read instead of a gather through the vector L1 (which is what bounded the row-wave kernels:
 * profiles/r03_rowwave_pmc.json). Same arithmetic as cpd_gather_conv with CPD_GC_F16X2 | CPD_GC_IN_PAIRS (identical partial products,
 * accumulated in the same tap order). Takes: kv = 27, c_in % 32 == 0, c_out = 32 / 64 / 128, fp16-pair input rows, >= 512 tiles
 * (cpd_gather_conv_planned_supported tells; otherwise CPD_ERR_UNSUPPORTED and the caller uses cpd_gather_conv_ws: same result).
 * flags: CPD_GC_F16X2 | CPD_GC_IN_PAIRS [| CPD_GC_OUT_PAIRS | CPD_GC_RES_PAIRS]. */
int cpd_gather_conv_planned_supported(int n_in, int n_out, int c_in, int c_out, int in_ld, int kv, int flags);
int cpd_gather_conv_planned(const float *in, int in_ld, int n_in, int c_in, const float *packed_w, const uint32_t *tapmask,
                            const uint16_t *plan_slots, const int32_t *plan_ulist, const int32_t *plan_count, int tile_rows, int kv,
                            int n_out, int c_out, const float *scale, const float *shift, const float *residual, int res_ld,
                            int relu, float *out, int out_ld, int flags, uint32_t *out_absmax, cpd_stream_t stream);
/* Introspection for benchmarks/profilers: which kernel instantiation cpd_gather_conv runs for
 * this problem: wg=1 -> tile_conv_kernel<a,b> (a x b workgroup tile), wg=0 ->
 * gather_conv_kernel<a,b,vec> ((16a) x (16b) wave tile; vec = 16-byte A pieces). HOST only.  */
int cpd_gather_conv_tile(int n_out, int c_in, int c_out, int in_ld, int flags, int *wg, int *a,
                         int *b, int *vec);

/* SparseConvTensor.dense() + view(N, C*D, H, W) (height_compression.py:136-138).
 *   nchw: out (B, C*D, H, W), channel = c*D + z   -- the reference layout
 *   nhwc: out (B, H, W, D*C), channel = z*C + c   -- channels-last, feeds cpd_gather_conv
 *   nhwc_cd: out (B, H, W, C*D), channel = c*D + z -- channels-last memory with the reference's channel order: the (N, C*D, H, W)
 *            tensor of height_compression.py:136-138 in torch's channels_last format (what the module path hands to Conv2d)
 * All zero-fill `out` themselves.                                                            */
int cpd_densify_nchw(const float *feat, const int32_t *indices, int n, int c, int batch,
                     const int32_t shape_zyx[3], float *out, cpd_stream_t stream);
int cpd_densify_nhwc(const float *feat, const int32_t *indices, int n, int c, int batch,
                     const int32_t shape_zyx[3], float *out, cpd_stream_t stream);
/* cpd_densify_nhwc into a PERSISTENT map that is all zero on entry (no clear inside), and the call that restores that state afterwards by
 * zeroing the same rows -- queue it after the map's last reader. SparseConvTensor.dense() semantics as above (height_compression.py:136-138);
 * moves 2 x sites x c floats instead of the whole map. */
int cpd_densify_nhwc_rows(const float *feat, const int32_t *indices, int n, int c, int batch,
                          const int32_t shape_zyx[3], float *out, cpd_stream_t stream);
int cpd_densify_nhwc_clear(const int32_t *indices, int n, int c, int batch, const int32_t shape_zyx[3], float *out,
                           cpd_stream_t stream);
int cpd_densify_nhwc_cd(const float *feat, const int32_t *indices, int n, int c, int batch,
                        const int32_t shape_zyx[3], float *out, cpd_stream_t stream);
/* Dense-pixel rulebooks for the BEV convs: nbr[kh*kw][ho*wo*batch] for a (kh x kw, stride, pad)
 * Conv2d over a (batch, h, w) channels-last map. */
int cpd_rulebook_conv2d(int batch, int h, int w, int kh, int kw, int stride, int pad,
                        int32_t *nbr, cpd_stream_t stream);

/* ===== CenterHead decode ===================================================================
 * Replaces CenterHead.generate_predicted_boxes (center_head.py:252-303) ->
 * centernet_utils.decode_bbox_from_heatmap / _topk (centernet_utils.py:136-216) for one sample:
 * sigmoid(hm), exp(dim), per-class top-K then top-K over classes, gather, atan2(sin,cos),
 * centre scaling, POST_CENTER_LIMIT_RANGE and SCORE_THRESH masks, order-preserving compaction.
 * Head maps are addressed as map[pixel*pix_stride + channel*ch_stride] so both the reference's
 * NCHW planes and this library's channels-last rows can be decoded in place.
 * `batch` samples are decoded by one call: sample b's maps start sample_stride floats after
 * sample b-1's. Outputs (capacity K per sample): boxes [batch,K,7] (x,y,z,dx,dy,dz,heading),
 * scores [batch,K], labels [batch,K] i32 (0-based class id), n_out [batch] device counts.     */
size_t cpd_center_decode_workspace_bytes(int batch, int num_class, int hw, int k);
int cpd_center_decode(const float *hm, const float *center, const float *center_z,
                      const float *dim, const float *rot, int batch, long long sample_stride,
                      int pix_stride, int ch_stride, int num_class, int h, int w, int k,
                      float feature_map_stride,
                      const float voxel_xy[2], const float range_lo_xy[2],
                      const float limit_range[6], float score_thresh, float *boxes, float *scores,
                      int32_t *labels, int32_t *n_out, void *workspace, size_t workspace_bytes,
                      cpd_stream_t stream);

/* ===== B3. iou3d_nms =======================================================================
 * Replaces the iou3d_nms_cuda extension (cpd/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17).
 * Boxes are [n,7] f32 (x,y,z,dx,dy,dz,heading), contiguous.                                  */
/* boxes_overlap_bev_gpu (iou3d_nms.cpp:49-68): out[n,m] rotated BEV intersection area. */
int cpd_boxes_overlap_bev(const float *a, int n, const float *b, int m, float *out,
                          cpd_stream_t stream);
/* boxes_iou_bev_gpu (iou3d_nms.cpp:70-88): out[n,m] rotated BEV IoU. */
int cpd_boxes_iou_bev(const float *a, int n, const float *b, int m, float *out,
                      cpd_stream_t stream);
/* boxes_iou3d_gpu (iou3d_nms_utils.py:67-100) fused: BEV overlap x height overlap / union. */
int cpd_boxes_iou3d(const float *a, int n, const float *b, int m, float *out,
                    cpd_stream_t stream);
/* nms_gpu / nms_normal_gpu (iou3d_nms.cpp:90-137,139-186): boxes sorted by descending score.
 * The 64x64 bitmask (iou3d_nms_kernel.cu:267-311) AND the greedy scan run on the device; keep
 * [n] i64 and num_keep (i32 scalar) are DEVICE buffers -- the reference's blocking D2H copy of
 * the whole mask is gone; the shim copies keep[:num] back to satisfy the CPU-`keep` contract. */
size_t cpd_nms_workspace_bytes(int n);
int cpd_nms_rotated(const float *boxes, int n, float thresh, int64_t *keep, int32_t *num_keep,
                    void *workspace, size_t workspace_bytes, cpd_stream_t stream);
int cpd_nms_normal(const float *boxes, int n, float thresh, int64_t *keep, int32_t *num_keep,
                   void *workspace, size_t workspace_bytes, cpd_stream_t stream);
/* Batched form for the per-sample NMS loop of generate_predicted_boxes (center_head.py:281-296):
 * boxes [batch, capacity, 7] sorted by descending score per sample, valid counts [batch] ON THE
 * DEVICE (e.g. cpd_center_decode's n_out -- no host read-back in between); keep [batch, capacity],
 * num_keep [batch]; workspace = batch * cpd_nms_workspace_bytes(capacity).                      */
int cpd_nms_batch(const float *boxes, const int32_t *counts, int batch, int capacity, float thresh,
                  int normal, int64_t *keep, int32_t *num_keep, void *workspace,
                  size_t workspace_bytes, cpd_stream_t stream);
/* cpd_nms_batch for callers that keep only the FIRST max_keep survivors of a sample -- class_agnostic_nms keeps
 * selected[:NMS_POST_MAXSIZE] (model_nms_utils.py:115-134), RoIHeadTemplate.proposal_layer 4096 candidates -> the first few hundred
 * (roi_head_template.py:53-114): greedy suppression decides box i from boxes before i only, so the first max_keep survivors are known
 * as soon as the scan has found them. The mask is built for the first `row_limit` boxes of each sample only ((row_limit)^2 / 2 IoUs
 * instead of capacity^2 / 2) and the scan stops after the 64-box block that holds the max_keep-th survivor: keep[b][0 .. min(num_keep[b],
 * max_keep)) are EXACTLY cpd_nms_batch's first entries. incomplete[b] = 1 when sample b has more than row_limit boxes and fewer than
 * max_keep of its first row_limit survived -- the answer then needs boxes the mask does not cover, and the caller runs cpd_nms_batch
 * (the flag is a device word: read it with the counts). workspace = batch * cpd_nms_workspace_bytes(min(row_limit, capacity)). */
int cpd_nms_batch_first(const float *boxes, const int32_t *counts, int batch, int capacity, float thresh,
                        int normal, int max_keep, int row_limit, int64_t *keep, int32_t *num_keep,
                        int32_t *incomplete, void *workspace, size_t workspace_bytes, cpd_stream_t stream);
/* ... and the full call for exactly the samples that asked for it, without a host read-back in between: cpd_nms_batch over the
 * samples b with where[b] != 0 (a device array, e.g. cpd_nms_batch_first's `incomplete`); the others keep their keep / num_keep.
 * The launch costs its empty workgroups when no sample asks (measured: DESIGN 5h).                                              */
int cpd_nms_batch_where(const float *boxes, const int32_t *counts, const int32_t *where, int batch,
                        int capacity, float thresh, int normal, int64_t *keep, int32_t *num_keep,
                        void *workspace, size_t workspace_bytes, cpd_stream_t stream);
/* The score half of Detector3DTemplate.post_processing (cpd/models/detectors/detector3d_template.py:222-343, MULTI_CLASSES_NMS False) and of
 * class_agnostic_nms (cpd/models/model_utils/model_nms_utils.py:113-124) for a whole batch in one launch: per frame, score = max over the
 * n_cls columns of sigmoid(cls) (of cls itself when normalized != 0), rows with score >= score_thresh ranked by score descending (ties: lower
 * index first -- a stable sort), the others after them with score -1; out_boxes / out_scores / out_labels (int32) = the r rows in that order,
 * n_ok[b] = min(#rows above the threshold, pre_max): what cpd_nms_batch + cpd_select_boxes take next. cls [batch, r, n_cls], boxes
 * [batch, r, 7], labels_i64 [batch, r]. r <= 8192, else CPD_ERR_UNSUPPORTED. */
int cpd_rank_scores(const float *cls, int n_cls, const float *boxes, const void *labels_i64, int batch, int r, float score_thresh,
                    int pre_max, int normalized, float *out_boxes, float *out_scores, int32_t *out_labels, int32_t *n_ok,
                    cpd_stream_t stream);
/* class_agnostic_nms tail (model_nms_utils.py:126-127,134) + the `+1` of center_head.py:301, batched:
 * out[b][k] = in[b][keep[b][k]] for k < out_n[b] = min(num_keep[b], post_max).
 * out_boxes [batch,post_max,7], out_scores [batch,post_max], out_labels [batch,post_max] i64.  */
int cpd_select_boxes(const float *boxes, const float *scores, const int32_t *labels,
                     const int64_t *keep, const int32_t *num_keep, int batch, int capacity,
                     int post_max, int label_offset, float *out_boxes, float *out_scores,
                     int64_t *out_labels, int32_t *out_n, cpd_stream_t stream);
/* boxes_iou_bev_cpu (iou3d_cpu.cpp:232-252): HOST pointers, runs on the calling thread. This is
 * the one CPU entry point the reference extension itself exports (used by the dataloader's
 * gt-sampling, database_sampler.py:445-446); it is product code, not the test oracle.        */
int cpd_boxes_iou_bev_cpu(const float *a, int n, const float *b, int m, float *out);


/* ===== Training step (BASELINE config 3) ====================================================
 * Forward in training mode is cpd_gather_conv (no folded BN) + cpd_bn_stats + cpd_affine_rows; the
 * input gradient of every conv is cpd_gather_conv again on transposed (tap-flipped for SubM /
 * stride-1) weights with the same or the transposed rulebook; the rest is below. These replace
 * what torch autograd + cuDNN + spconv's backward ops do under tools/train_utils/train_utils.py:41
 * (loss.backward()) for the modules of the path.                                               */
size_t cpd_col_reduce_workspace_bytes(int n, int c);
/* column sums of x[n, c] (bias gradients). */
int cpd_col_sum(const float *x, int ldx, int n, int c, float *sum, void *ws, size_t ws_bytes,
                cpd_stream_t stream);
/* training-mode BatchNorm statistics: sum[c], sumsq[c] over the n rows (two-stage, deterministic). */
int cpd_bn_stats(const float *x, int ldx, int n, int c, float *sum, float *sumsq, void *ws,
                 size_t ws_bytes, cpd_stream_t stream);
/* Per-channel bookkeeping after cpd_bn_stats: mean, invstd = 1/sqrt(var_biased + eps), the affine
 * scale = gamma*invstd / shift = beta - mean*scale, and (optional) running-stat update
 * running = (1-momentum)*running + momentum*batch (unbiased variance), as torch.nn.BatchNorm does. */
int cpd_bn_finalize(const float *sum, const float *sumsq, int n, int c, float eps, float momentum,
                    const float *gamma, const float *beta, float *mean, float *invstd, float *scale,
                    float *shift, float *running_mean, float *running_var, cpd_stream_t stream);
/* cpd_bn_stats + cpd_bn_finalize in one call (two launches): the training-mode BatchNorm forward
 * bookkeeping of one layer. */
int cpd_bn_stats_finalize(const float *x, int ldx, int n, int c, float eps, float momentum,
                          const float *gamma, const float *beta, float *mean, float *invstd,
                          float *scale, float *shift, float *running_mean, float *running_var,
                          void *workspace, size_t workspace_bytes, cpd_stream_t stream);
/* out = act(x * scale + shift + residual): BatchNorm apply (scale = gamma*invstd,
 * shift = beta - mean*scale), SparseBasicBlock tail (spconv_backbone.py:131-134). In place allowed. */
int cpd_affine_rows(const float *x, int ldx, int n, int c, const float *scale, const float *shift,
                    const float *residual, int ldr, int relu, float *out, int ldo,
                    cpd_stream_t stream);
/* BatchNorm(+ReLU) backward. dy_m = dy * (y > 0) (y = NULL: no ReLU);
 * reduce: dbeta = sum dy_m, dgamma = sum dy_m * xhat; apply: dx = gamma*invstd*(dy_m - dbeta/n -
 * xhat*dgamma/n), optional dres = dy_m (gradient into a residual added before the ReLU).       */
int cpd_bn_bwd_reduce(const float *dy, int lddy, const float *y, int ldy, const float *x, int ldx,
                      const float *mean, const float *invstd, int n, int c, float *dbeta,
                      float *dgamma, void *ws, size_t ws_bytes, cpd_stream_t stream);
int cpd_bn_bwd_apply(const float *dy, int lddy, const float *y, int ldy, const float *x, int ldx,
                     int n, int c, const float *mean, const float *invstd, const float *gamma,
                     const float *dbeta, const float *dgamma, float *dx, int lddx, float *dres,
                     int lddres, uint32_t *dx_absmax, cpd_stream_t stream);
/* SyncBatchNorm (tools/train.py:32,117 `--sync_bn` -> torch.nn.SyncBatchNorm.convert_sync_batchnorm; off by default in the reference):
 * the host all-reduces (sum, sumsq, row count) between cpd_bn_stats and the finalize, and (sum dy, sum dy * xhat) between
 * cpd_bn_bwd_reduce and the apply, over the data-parallel ranks (RCCL). These two entry points take the row count the sums now stand
 * for as a DEVICE float (`n_total`: the all-reduced count -- ranks hold different numbers of sparse rows, and no rank knows the total
 * on the host); otherwise they are cpd_bn_finalize / cpd_bn_bwd_apply. dgamma / dbeta of the PARAMETERS stay the local sums
 * (torch.nn.SyncBatchNorm's backward: the gradient all-reduce averages them like every other gradient). */
int cpd_bn_finalize_sync(const float *sum, const float *sumsq, const float *n_total, int c, float eps, float momentum,
                         const float *gamma, const float *beta, float *mean, float *invstd, float *scale, float *shift,
                         float *running_mean, float *running_var, cpd_stream_t stream);
int cpd_bn_bwd_apply_sync(const float *dy, int lddy, const float *y, int ldy, const float *x, int ldx, int n, int c,
                          const float *mean, const float *invstd, const float *gamma, const float *sum_dy,
                          const float *sum_dy_xhat, const float *n_total, float *dx, int lddx, float *dres, int lddres,
                          uint32_t *dx_absmax, cpd_stream_t stream);
/* dx_absmax (optional): an "absmax block" -- CPD_ABSMAX_SLOTS device words CPD_ABSMAX_STRIDE uint32 apart (one per
 * 128-byte line: atomics on one line serialise), zeroed by the caller beforehand -- whose words the kernel raises (atomic
 * max) so that their maximum is the bits of max |dx|: what cpd_gather_conv_scaled / cpd_conv_wgrad_scaled need to run a
 * gradient through the split-fp16 path. A caller that knows the maximum writes it to word 0 of a zeroed block. */
#define CPD_ABSMAX_SLOTS 16
#define CPD_ABSMAX_STRIDE 32
#define CPD_ABSMAX_WORDS (CPD_ABSMAX_SLOTS * CPD_ABSMAX_STRIDE)
int cpd_relu_bwd(const float *dy, int lddy, const float *y, int ldy, int n, int c, float *dx,
                 int lddx, cpd_stream_t stream);
/* Weight gradient of cpd_gather_conv: dw[t][ci][co] (+)= sum_j in[nbr[t][j]][ci] * dy[j][co]
 * (dense [kv][c_in][c_out] layout, the layout cpd_pack_weight consumes). Deterministic (row-chunk
 * partials in `ws`, summed in a fixed order). flags: bit 0 = accumulate into dw_kio, CPD_GC_BF16X3 =
 * split-bf16 arithmetic (fp32-equivalent, as in cpd_gather_conv) when c_in and c_out are multiples
 * of 32, with rows that lack the tap compacted away before staging; otherwise fp32 MFMA.        */
size_t cpd_conv_wgrad_workspace_bytes(int n_out, int c_in, int c_out, int kv);
int cpd_conv_wgrad(const float *in, int in_ld, int c_in, const float *dy, int dy_ld, int c_out,
                   const int32_t *nbr, int kv, int n_out, float *dw_kio, int flags, void *ws,
                   size_t ws_bytes, cpd_stream_t stream);
/* The same with flags CPD_GC_F16X2 allowed: split-fp16 arithmetic (three products instead of six). Gradients do not live in
 * fp16's range, so each operand may come with an absmax block (see cpd_bn_bwd_apply) holding the bits of its max |value|
 * (in_absmax for `in`, dy_absmax for `dy`; NULL = use as is): the kernel multiplies the operand by the power of two that puts that maximum at
 * [2^14, 2^15) before splitting and divides the partial sums by it again -- exact, and an element 2^-18 of the maximum or
 * larger keeps its full 2^-24 relative precision (smaller ones 2^-43 of the maximum absolute). */
int cpd_conv_wgrad_scaled(const float *in, int in_ld, int c_in, const float *dy, int dy_ld, int c_out,
                          const int32_t *nbr, int kv, int n_out, float *dw_kio, int flags,
                          const uint32_t *in_absmax, const uint32_t *dy_absmax, void *ws, size_t ws_bytes,
                          cpd_stream_t stream);
/* Input-gradient convolutions through the split-fp16 kernels: cpd_gather_conv / cpd_conv3x3_rows with `in` pre-scaled the
 * same way (in_absmax as above; ignored by the fp32 and split-bf16 kernels, which need no range help). */
int cpd_gather_conv_scaled(const float *in, int in_ld, int n_in, int c_in, const float *packed_w,
                           const int32_t *nbr, const uint32_t *tapmask, int kv, int n_out, int c_out, const float *scale,
                           const float *shift, const float *residual, int res_ld, int relu, float *out,
                           int out_ld, const int32_t *out_row_map, int out_col_group, int flags,
                           const uint32_t *in_absmax, cpd_stream_t stream);
int cpd_conv3x3_rows_scaled(const float *in, int in_ld, int frames, int h, int w, int c_in,
                            const float *packed_w, int c_out, const float *scale, const float *shift,
                            const float *residual, int res_ld, int relu, float *out, int out_ld, int flags,
                            const uint32_t *in_absmax, cpd_stream_t stream);
/* The range guard of the f16x2 INFERENCE path (fp16's narrow exponent is the one way split-fp16 arithmetic differs from the
 * reference's fp32 convolutions, spconv_backbone.py:108-136 / base_bev_backbone.py:31-59): every launch may leave the bits of
 * max |out| behind in an absmax block (out_absmax; zero the block before the first launch that writes it -- several launches
 * may raise the same block, e.g. the two halves of a concat buffer), and takes its input's block as in_absmax exactly like the
 * _scaled calls above: the split-fp16 kernels then pre-scale the input by a power of two and undo it in the epilogue (exact),
 * so an activation of any fp32 magnitude gives the fp32 answer instead of inf / NaN. in_absmax = NULL: input used as is;
 * out_absmax = NULL: nothing recorded. cpd_absmax_rows raises a block to the range of a tensor that came from elsewhere. */
int cpd_gather_conv_ranged(const float *in, int in_ld, int n_in, int c_in, const float *packed_w,
                           const int32_t *nbr, const uint32_t *tapmask, int kv, int n_out, int c_out, const float *scale,
                           const float *shift, const float *residual, int res_ld, int relu, float *out,
                           int out_ld, const int32_t *out_row_map, int out_col_group, int flags,
                           const uint32_t *in_absmax, uint32_t *out_absmax, cpd_stream_t stream);
/* The same with a workspace: small sparse launches (one frame, the train step: fewer row-wave workgroups than ~2 per CU) deal the
 * taps of a row tile to several workgroups that write partial sums into the workspace; a second launch adds them in a fixed order
 * and runs the epilogue (results equal the unsplit launch up to fp32 summation order). cpd_gather_conv_split_bytes: the workspace
 * this problem wants (0: it runs unsplit, and cpd_gather_conv_ws with any workspace is cpd_gather_conv_ranged). HOST only. */
size_t cpd_gather_conv_split_bytes(int n_out, int c_in, int c_out, int in_ld, int kv, int flags);
int cpd_gather_conv_ws(const float *in, int in_ld, int n_in, int c_in, const float *packed_w,
                       const int32_t *nbr, const uint32_t *tapmask, int kv, int n_out, int c_out, const float *scale,
                       const float *shift, const float *residual, int res_ld, int relu, float *out,
                       int out_ld, const int32_t *out_row_map, int out_col_group, int flags,
                       const uint32_t *in_absmax, uint32_t *out_absmax, void *workspace, size_t workspace_bytes,
                       cpd_stream_t stream);
int cpd_conv3x3_rows_ranged(const float *in, int in_ld, int frames, int h, int w, int c_in,
                            const float *packed_w, int c_out, const float *scale, const float *shift,
                            const float *residual, int res_ld, int relu, float *out, int out_ld, int flags,
                            const uint32_t *in_absmax, uint32_t *out_absmax, cpd_stream_t stream);
int cpd_absmax_rows(const float *x, int ld, long long n, int c, uint32_t *absmax_block, cpd_stream_t stream);
/* All packed images of a model in three launches (a train step rewrites every one of them after the optimiser step; one
 * cpd_pack_weight / cpd_pack_weight_adjoint call is four launches). A job names one packed buffer (sized by
 * cpd_packed_weight_floats for the conv the image is FOR) and the [kv][c_in][c_out] tensor it is built from: adjoint = 0
 * is cpd_pack_weight, adjoint = 1 cpd_pack_weight_adjoint(flip_taps). cpd_pack_batch_prepare (HOST call, synchronous copy)
 * writes the device table once -- the pointers must stay valid -- and returns the three grid sizes; cpd_pack_batch_run
 * rebuilds every image: images bit 0 = split-bf16, bit 1 = split-fp16 (the fp32 image always). Results are identical to
 * the per-tensor calls. */
typedef struct {
    const float *w;
    float *packed;
    int32_t kv, c_in, c_out, adjoint, flip_taps;
} cpd_pack_job;
size_t cpd_pack_batch_table_bytes(int n_jobs);
int cpd_pack_batch_prepare(const cpd_pack_job *jobs, int n_jobs, void *table, size_t table_bytes,
                           int32_t grid_blocks[3]);
int cpd_pack_batch_run(const void *table, int n_jobs, const int32_t grid_blocks[3], int images,
                       cpd_stream_t stream);
/* Packed weights of the adjoint conv used for input gradients: Wd[t'][co][ci] = W[t][ci][co],
 * t = kv-1-t' if flip_taps (SubM / stride-1: the forward rulebook is its own transpose up to the
 * tap flip) else t = t' (use with a transposed rulebook). Size: cpd_packed_weight_floats(kv, c_out, c_in). */
int cpd_pack_weight_adjoint(const float *w_kio, int kv, int c_in, int c_out, int flip_taps,
                            float *packed, cpd_stream_t stream);
/* Transposed rulebooks for the input gradient of strided convs: nbr_t[t][i] = output row that
 * input row i feeds through tap t (or -1). Sparse: out_index is the index cpd_conv_outset built. */
int cpd_rulebook_conv_transpose(const int32_t *in_indices, int n_in, int batch,
                                const int32_t in_shape[3], const int32_t ksize[3],
                                const int32_t stride[3], const int32_t pad[3], const void *out_index,
                                int32_t *nbr_t, cpd_stream_t stream);
int cpd_rulebook_conv2d_transpose(int batch, int h, int w, int kh, int kw, int stride, int pad,
                                  int32_t *nbr_t, cpd_stream_t stream);
/* CenterHead.get_loss (center_head.py:225-250) fused with its gradient, on channels-last head rows
 * [batch*hw][ld] (columns 0..7 = center(2), center_z, dim(3), rot(2); columns hm_col.. = heatmap logits):
 * hm loss = FocalLossCenterNet on clamp(sigmoid, 1e-4, 1-1e-4) (loss_utils.py:265-300) x cls_weight,
 * loc loss = sum_d code_weights[d] * RegLossCenterNet_d (masked L1 at the object pixels / max(#objects, 1),
 * loss_utils.py:315-386) x loc_weight. heat [batch][num_classes][hw], target [batch][k][8], inds / masks
 * [batch][k] int64 as assign_targets builds them. Writes d(total)/d(rows) to d_rows [batch*hw][ld] (all
 * columns) and losses[3] = {total, hm, loc} on the device; deterministic (fixed-order sums, no atomics). */
size_t cpd_center_loss_workspace_bytes(int batch, int hw, int ld);
int cpd_center_loss(const float *rows, int ld, int batch, int hw, int num_classes, int hm_col,
                    const float *heat, const float *target, const int64_t *inds,
                    const int64_t *masks, int k, const float code_weights[8], float loc_weight,
                    float cls_weight, float *d_rows, float *losses, void *ws, size_t ws_bytes,
                    cpd_stream_t stream);
/* CenterHead.assign_targets / assign_target_of_single_head (center_head.py:103-219) with centernet_utils.gaussian_radius /
 * draw_gaussian_to_heatmap (centernet_utils.py:9-69) for every sample of a batch, on the device, nothing read back (the reference
 * walks the boxes on the CPU, l.204). gt_boxes [batch][m][8] = x, y, z, dx, dy, dz, heading, class (1..num_classes; < 1 = padding):
 * the boxes with class >= 1 are taken in their order (l.180-196), the first k of them (NUM_MAX_OBJS, l.113) give
 * heat [batch][num_classes][h][w] (gaussians of radius max(min_radius, int(gaussian_radius(dx, dy / pixel, gaussian_overlap))),
 * max-merged), target [batch][k][8] = (x - int x, y - int y, z, log dx, log dy, log dz, cos, sin), inds [batch][k] = y * w + x of the
 * centre pixel, masks [batch][k] = 1 -- boxes with dx <= 0 or dy <= 0 and unused slots give zeros. fp32 in the reference's operation
 * order; the gaussian in double, rounded to float (numpy). Outputs must be 16-byte aligned; m <= 12000. Deterministic. Three launches. */
size_t cpd_center_targets_workspace_bytes(int batch, int k);
int cpd_center_targets(const float *gt_boxes, int batch, int m, int num_classes, int h, int w, int k,
                       const float pc_range_xy[2], const float voxel_xy[2], int feature_map_stride,
                       double gaussian_overlap, int min_radius, float *heat, float *target, int64_t *inds,
                       int64_t *masks, void *ws, size_t ws_bytes, cpd_stream_t stream);
/* AnchorHeadTemplate.get_loss (anchor_head_template.py:179-334) fused with its gradient: sigmoid focal
 * classification loss (alpha 0.25, gamma 2), smooth-L1 (beta 1/9, code_weights) on the residual-coded boxes
 * with the sin-difference heading encoding, cross entropy on the direction bins (dir_preds NULL: no direction
 * classifier); per-sample normalisation by max(#positive anchors, 1), sums divided by batch, times the three
 * loss weights. cls_preds [batch][n_anchors][num_class], box_preds / reg_targets [batch][n_anchors][7],
 * dir_preds [batch][n_anchors][num_dir_bins], labels [batch][n_anchors] (-1 ignore, 0 background, k class k:
 * what cpd_anchor_assign emits), anchors [n_anchors][7]. Outputs: d_cls / d_box / d_dir (same shapes as the
 * predictions) and losses[4] = {total, cls, loc, dir} on the device. Deterministic.                       */
size_t cpd_anchor_loss_workspace_bytes(int batch, int n_anchors);
int cpd_anchor_loss(const float *cls_preds, const float *box_preds, const float *dir_preds,
                    const int32_t *labels, const float *reg_targets, const float *anchors, int batch,
                    int n_anchors, int num_class, int num_dir_bins, float dir_offset,
                    const float code_weights[7], float cls_weight, float loc_weight, float dir_weight,
                    float *d_cls, float *d_box, float *d_dir, float *losses, void *ws,
                    size_t ws_bytes, cpd_stream_t stream);
/* Adam with decoupled weight decay on a flat buffer (tools/train_utils/optimization/fastai_optim.py:
 * 132-150 true_wd semantics): grad is multiplied first by grad_scale (1/world after all-reduce) and,
 * when grad_scale_dev is not NULL, by the float it points to in device memory (the clip factor of
 * clip_grad_norm_, computed on the device so that the step needs no host read-back). `step` counts
 * from 1.                                                                                        */
int cpd_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  float grad_scale, const float *grad_scale_dev, cpd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * RoI-head feature pooling: the pointnet2_stack CUDA extension's voxel query and
 * grouping (cpd/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp: voxel_query_wrapper,
 * group_points_wrapper) and generate_voxel2pinds (cpd/utils/spconv_utils.py:4-21).
 * ------------------------------------------------------------------------------------------ */
/* Dense (B,Z,Y,X) int32 volume: row of the voxel in `indices` [n,4] (b,z,y,x), -1 = empty. */
int cpd_voxel2pinds(const int32_t *indices, int n, int batch, const int32_t shape_zyx[3],
                    int32_t *out_volume, cpd_stream_t stream);
/* voxel_query_wrapper (voxel_query_gpu.cu:10-87): for each of m query points (new_xyz [m,3],
 * new_coords [m,4] = (b,z,y,x) voxel coordinates) scan the (2zr+1)(2yr+1)(2xr+1) neighbourhood in
 * dz, dy, dx order and keep the first `nsample` voxels whose centre (xyz [N,3]) lies within
 * `radius`; the first hit pre-fills all slots; no hit: idx[0] = -1 and the other slots are left
 * as the caller initialised them (the reference zero-fills). idx [m, nsample] i32. */
int cpd_voxel_query(int m, int r1, int r2, int r3, int nsample, float radius, int z_range, int y_range,
                    int x_range, const float *new_xyz, const float *xyz, const int32_t *new_coords,
                    const int32_t *point_indices, int32_t *idx, cpd_stream_t stream);
/* Same query through a site index (cpd_index_build / cpd_conv_outset) of the sparse tensor instead
 * of the dense volume. Whether the index's ranks are row ids (canonical lists of cpd_conv_outset)
 * or go through its rank -> row permutation (cpd_index_build over any row order) is recorded in
 * the index and read on the device; `use_perm` is ignored (kept for ABI stability). */
int cpd_voxel_query_index(int m, int batch, int r1, int r2, int r3, int nsample, float radius,
                          int z_range, int y_range, int x_range, const float *new_xyz, const float *xyz,
                          const int32_t *new_coords, const void *index, int use_perm, int n_sites,
                          int32_t *idx, cpd_stream_t stream);
/* RoI grid points and their cell coordinates in one launch: VoxelRCNNHead.get_global_grid_points_of_roi + the `cur_coords` arithmetic of
 * roi_grid_pool (cpd/models/roi_heads/voxel_rcnn_head.py:186-273, 365-386; rotate_points_along_z, cpd/utils/common_utils.py:35-57).
 * rois [n_rois, roi_ld >= 7] (x, y, z, dx, dy, dz, heading), frame of RoI i = i / rois_per_frame. grid_xyz [n_rois * G^3, 3] = the global
 * grid points (fp32, the reference's operation order); for each of n_levels (<= 4) strides, level_coords[l] [n_rois * G^3, 4] int32 =
 * (b, x, y, z) with x = ((px - range_lo.x) // voxel_size.x) // stride (torch floor division on floats, then int), or (b, z, y, x) when
 * bzyx != 0 (the order the neighbour queries take). level_coords pointers must be 16-byte aligned. */
int cpd_roi_grid_points(const float *rois, int roi_ld, int n_rois, int rois_per_frame, int grid_size, const float voxel_size[3],
                        const float range_lo[3], int n_levels, const int32_t *strides, int32_t *const *level_coords, int bzyx,
                        float *grid_xyz, cpd_stream_t stream);
/* The same query for a VOXEL level, where the reference passes xyz = get_voxel_centers(indices) (common_utils.py:66-82;
 * voxel_rcnn_head.py:236-241): a site's coordinates are (cell + 0.5) * cell_xyz + origin_xyz per axis in fp32, so the kernel evaluates
 * the distance test from the cell coordinates (same operations, same order: identical decisions) and reads nothing per candidate
 * until it is a hit. cell_xyz = fp32(voxel_size) * stride, origin_xyz = point_cloud_range[0:3] (host arrays, x y z).
 * 2 * x_range + 1 <= 32, else CPD_ERR_UNSUPPORTED (use cpd_voxel_query_index). */
int cpd_voxel_query_index_grid(int m, int batch, int r1, int r2, int r3, int nsample, float radius,
                               int z_range, int y_range, int x_range, const float *new_xyz,
                               const int32_t *new_coords, const void *index, int n_sites,
                               const float cell_xyz[3], const float origin_xyz[3], int32_t *idx, cpd_stream_t stream);
/* group_points_wrapper (group_points_gpu.cu:69-99): out[pt][c][s] = features[start(batch of pt) +
 * idx[pt][s]][c]; *_batch_cnt are device int32[b]. out [m, c, nsample]. */
int cpd_group_points(int b, int m, int c, int nsample, const float *features,
                     const int32_t *features_batch_cnt, const int32_t *idx, const int32_t *idx_batch_cnt,
                     float *out, cpd_stream_t stream);
/* group_points_grad_wrapper (group_points_gpu.cu:9-66): grad_features [n, c] (zeroed here) += grad_out [m, c, nsample]
 * scattered through idx; the backward of cpd_group_points. Float atomics: the summation order is not fixed, as in the
 * reference. */
int cpd_group_points_grad(int b, int m, int c, int nsample, int n, const float *grad_out,
                          const int32_t *features_batch_cnt, const int32_t *idx, const int32_t *idx_batch_cnt,
                          float *grad_features, cpd_stream_t stream);
/* Fused grouping + position encoding + ReLU + max-pool of NeighborVoxelSAModuleMSG.forward
 * (voxel_pool_modules.py:96-117): out[m][ch] = max_s relu(features_in[idx[m][s]][ch] +
 * (xyz[idx[m][s]] - new_xyz[m]) . w_pos[:, ch] + b_pos[ch]), relu(b_pos[ch]) for an empty ball
 * (idx[m][0] < 0). idx holds global rows; w_pos [3, c] / b_pos [c] = Conv2d(3, c) with its
 * eval BatchNorm folded. */
int cpd_voxel_pool_max(int m, int c, int nsample, const float *features_in, int features_ld,
                       const float *xyz, const float *new_xyz, const int32_t *idx, const float *w_pos,
                       const float *b_pos, float *out, int out_ld, cpd_stream_t stream);
/* ... followed in the same kernel by the module's output MLP (mlps_out, voxel_pool_modules.py:118-121: 1 x 1 conv, eval BatchNorm,
 * ReLU): out[m][co] = act(sum_ch pooled[m][ch] * w_out[ch][co] + t_out[co]), w_out [c, c_out] with the BatchNorm scale folded in,
 * t_out [c_out] its shift; the pooled [m, c] tensor is never stored. out may be a column block of a wider row (out_ld): the
 * scales of a level are written side by side, as the reference's torch.cat lays them out. c = 16, 32 or 64 and c_out <= 2 c,
 * else CPD_ERR_UNSUPPORTED. */
int cpd_voxel_pool_max_mlp(int m, int c, int nsample, const float *features_in, int features_ld,
                           const float *xyz, const float *new_xyz, const int32_t *idx, const float *w_pos,
                           const float *b_pos, const float *w_out, const float *t_out, int c_out, int relu,
                           float *out, int out_ld, cpd_stream_t stream);
/* ... and raising `out_absmax` (an absmax block, see cpd_gather_conv_ranged; may be NULL) to the bits of max |out| of the rows this call
 * writes: the RoI head's 27648-wide pooled rows (voxel_rcnn_head.py:186-273 -> shared_fc_layers, l.694-705) reach the split-fp16 FC
 * GEMMs with their range block already filled -- no separate pass over 880 MB of pooled features. */
int cpd_voxel_pool_max_mlp_ranged(int m, int c, int nsample, const float *features_in, int features_ld,
                                  const float *xyz, const float *new_xyz, const int32_t *idx, const float *w_pos,
                                  const float *b_pos, const float *w_out, const float *t_out, int c_out, int relu,
                                  float *out, int out_ld, uint32_t *out_absmax, cpd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dataloader pre-filter on the device.
 * ------------------------------------------------------------------------------------------ */
/* mask_points_by_range + boolean indexing (cpd/utils/common_utils.py:60-63,
 * data_processor.py:84-85): keeps, in order, the rows of points [n, c] with range[0] <= x <= range[3]
 * and range[1] <= y <= range[4]. out [n, c]; n_out device int32. */
size_t cpd_mask_points_workspace_bytes(int n);
int cpd_mask_points_by_range(const float *points, int n, int c, const float range_xyz[6], float *out,
                             int32_t *n_out, void *workspace, size_t workspace_bytes, cpd_stream_t stream);
/* Multi-sweep merge (waymo_unsupervised_dataset.py:333-360 get_frame with 192-202 points_rigid_transform): the sweeps'
 * points [sweep_offsets[n_sweeps], c] (sweeps back to back; sweep_offsets, poses and cur_pose_inv are HOST arrays) are taken
 * sweep -> world by poses[s] (row-major 4x4 float64) and world -> current frame by cur_pose_inv = inverse(current pose),
 * each product formed in float64 from the float32 coordinates and rounded to float32 as the reference's np.mat arithmetic
 * does; column 3 (intensity) and the last column are set to 0 (l.347-351); other columns are copied. n_sweeps <= 16. */
int cpd_merge_sweeps(const float *points, const int32_t *sweep_offsets, int n_sweeps, int c, const double *poses,
                     const double *cur_pose_inv, float *out, cpd_stream_t stream);
/* points_in_boxes_gpu (roiaware_pool3d_kernel.cu:313-336; check_pt_in_box3d l.23-35 uses MARGIN
 * 1e-5, the CPU twin roiaware_pool3d.cpp:128-140 1e-2): boxes [batch, boxes_num, 7], pts
 * [batch, pts_num, pts_ld >= 3] -> box_idx_of_points [batch, pts_num] = first containing box or -1. */
int cpd_points_in_boxes(int batch, int boxes_num, int pts_num, const float *boxes, const float *pts,
                        int pts_ld, float margin, int32_t *box_idx_of_points, cpd_stream_t stream);

/* Prototype box crop, the point work of sample_prototype_cpu (waymo_unsupervised_dataset.py:205-331).
 * cpd_points_in_boxes_mask = roiaware_pool3d_utils.points_in_boxes_cpu (roiaware_pool3d.cpp:128-168, MARGIN 1e-2, double
 * comparisons): out [k][n] = 1 where point n lies in box k. */
int cpd_points_in_boxes_mask(const float *boxes, int k, const float *pts, int n, int pts_ld, int32_t *out,
                             cpd_stream_t stream);
/* The two retained clouds of l.255-259 / l.317-318 in one pass, without the k x n matrix: out_no_object = the rows of
 * points [n][c] that lie in NO box, out_good_object = the rows that lie in no box with discard[k] != 0; both stable
 * (order preserved), counts on the device. Same in-box test as cpd_points_in_boxes_mask. */
size_t cpd_crop_boxes_workspace_bytes(int n);
int cpd_crop_boxes(const float *points, int n, int c, const float *boxes, const int32_t *discard, int k,
                   float *out_no_object, int32_t *n_no_object, float *out_good_object, int32_t *n_good_object,
                   void *workspace, size_t workspace_bytes, cpd_stream_t stream);
/* Prototype placement (l.277-305): rows (x, y, z, 1) of points [n][ld] times A^T, then times B^T (row-major 4x4 doubles:
 * A = inverse of the prototype box's pose, B = the target box's pose), float64 products without contraction; out [n][c_out]
 * gets the first three components rounded to fp32 and zeros in the other columns. */
int cpd_transform_points(const float *points, int n, int ld, const double a[16], const double b[16], int c_out,
                         float *out, cpd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Anchor head: anchor_head_template.py / axis_aligned_target_assigner.py / box_utils.py.
 * ------------------------------------------------------------------------------------------ */
/* box_utils.boxes3d_nearest_bev_iou (box_utils.py:275-287): a [n,7], b [m,7] -> out [n,m]. */
int cpd_nearest_bev_iou(const float *a, int n, const float *b, int m, float *out, cpd_stream_t stream);
/* AxisAlignedTargetAssigner.assign_targets_single (axis_aligned_target_assigner.py:153-243) with
 * match_height = False and POS_FRACTION < 0: per-anchor max/argmax IoU and per-GT max are computed
 * without an n x m matrix in HBM. labels [n] (-1 ignore, 0 background, class id), bbox_targets [n,7]
 * (ResidualCoder.encode_torch), reg_weights [n], gt_ious [n]; m <= 2048. */
size_t cpd_anchor_assign_workspace_bytes(int n, int m);
int cpd_anchor_assign(const float *anchors, int n, const float *gt_boxes, int m, const int32_t *gt_classes,
                      float matched_threshold, float unmatched_threshold, int norm_by_num_examples,
                      int32_t *labels, float *bbox_targets, float *reg_weights, float *gt_ious,
                      void *workspace, size_t workspace_bytes, cpd_stream_t stream);
/* AnchorHeadTemplate.generate_predicted_boxes (anchor_head_template.py:336-383): ResidualCoder
 * decode of box_preds [batch,n,7] against anchors [n,7] plus the direction-classifier correction
 * (dir_cls_preds [batch,n,num_dir_bins] or NULL). out [batch,n,7]. */
int cpd_anchor_decode(const float *box_preds, const float *anchors, const float *dir_cls_preds, int batch,
                      int n, int num_dir_bins, float dir_offset, float dir_limit_offset, float *out,
                      cpd_stream_t stream);

/* ATSSTargetAssigner.assign_targets_single (atss_target_assigner.py:76-141) for ONE frame and ONE anchor set, without the
 * n x m IoU / distance / "ious_inf" matrices the reference builds: per GT the `topk` nearest anchors (ties: lower index; the
 * reference's torch.topk leaves them unspecified), their IoUs (boxes_iou_bev, or boxes_iou3d_gpu when match_height), the
 * mean + std threshold and the centre-in-box test; per anchor the candidate GT of highest IoU; then every GT's argmax anchor
 * (lowest index on ties, anchor 0 when nothing overlaps) is forced to it, in GT order. gt_boxes [m][gt_ld], columns 0-6 the
 * box, column gt_ld-1 the class (float, as in gt_boxes_with_classes), trailing all-zero rows already trimmed (l.40-45).
 * Outputs (zeroed here): labels [n] float class ids (0 = background), reg_targets [n][7] (ResidualCoder.encode_torch),
 * reg_weights [n]. topk <= 64, topk * m <= 4096. */
size_t cpd_atss_workspace_bytes(int m, int topk);
int cpd_atss_assign(const float *anchors, int n_anchors, const float *gt_boxes, int gt_ld, int m, int topk,
                    int match_height, float *labels, float *reg_targets, float *reg_weights, void *workspace,
                    size_t workspace_bytes, cpd_stream_t stream);

/* Diagnostic (csrc/diag.hip; replaces nothing in the reference): the matrix pipe's sustained rate on THIS part under its socket power
 * cap -- `blocks` workgroups of four waves, each wave `iters` x 16 v_mfma_f32_16x16x32_f16 on register operands (a 64 x 64 x 32 tile
 * step), nothing else in the loop. a_operands / b_operands: 512 x 16 bytes of fp16 values each (what the operand DATA is matters: the
 * clock the cap allows depends on it); sink: blocks * 256 floats. cpd_mfma_burn_flops: the flops of one such launch (HOST only).
 * bench.py times it beside the dominant kernel: roofline.power_capped_peak. */
double cpd_mfma_burn_flops(int blocks, int iters);
int cpd_mfma_burn(const void *a_operands, const void *b_operands, float *sink, int blocks, int iters, cpd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CPD_HIP_H */
