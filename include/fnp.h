/*
 * fnp.h — C ABI of libfnp_hip.so, the MI355X (gfx950) drop-in for the point-cloud hot path of
 * djamahl99/findnpropagate (an OpenPCDet fork).
 *
 * Every entry point is what the reference's FFI for this path binds today (pybind11 torch
 * extensions and the third-party spconv package); the reference interface each one replaces is
 * cited as  <file>:<line>  relative to the reference tree.
 *
 * Conventions:
 *   - all pointers are DEVICE pointers borrowed from the caller (torch tensors), unless a
 *     parameter is documented as "host"; the caller allocates every output and every workspace;
 *   - no hidden allocation, no host synchronisation, no exit(): 0 = success, <0 = error code;
 *   - every launch goes to the hipStream_t handed in (the reference launches on the legacy
 *     default stream, roiaware_pool3d_kernel.cu:348, iou3d_nms_kernel.cu:394);
 *   - row counts that are data-dependent (voxels, active output sites) live in device memory
 *     (`const int *n_rows`), kernels grid-stride over them, so a whole forward is sync-free
 *     and hipGraph-capturable.
 */
#ifndef FNP_H
#define FNP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *fnp_stream_t; /* hipStream_t */

enum {
    FNP_OK = 0,
    FNP_ERR_ARG = -1,      /* bad shape / null pointer / unsupported channel count */
    FNP_ERR_LAUNCH = -2,   /* hipGetLastError() after a launch */
    FNP_ERR_HIP = -3,      /* a HIP runtime call failed */
    FNP_ERR_WORKSPACE = -4 /* workspace too small */
};

enum { FNP_F32 = 0, FNP_BF16 = 1, FNP_F16 = 2 };

/* Library / build identification: returns a static string "fnp-hip gfx950 <abi version>". */
const char *fnp_version(void);
int fnp_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * roiaware_pool3d — replaces pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:98-118
 * (points_in_boxes_gpu) and its kernel roiaware_pool3d_kernel.cu:16-36,313-359.
 * ------------------------------------------------------------------------------------------ */

/* boxes (B,T,7) [x,y,z,dx,dy,dz,heading], pts (B,M,3); box_idx_of_points (B,M) receives the
 * first (lowest) box index containing the point, else -1 (the reference needs the caller to
 * pre-fill -1, roiaware_pool3d_utils.py:38; this entry point writes every element). */
int fnp_points_in_boxes(const float *boxes, const float *pts, int *box_idx_of_points,
                        int B, int T, int M, fnp_stream_t stream);

/* Batched form of the Box Seeker's hot loop 4 (frustum_proposals_v1.py:930-932): for each of
 * T candidate boxes count the points (M,3) inside it, with the same test as above.
 * counts (T,) int32.  One launch instead of T launches + T host syncs. */
int fnp_points_in_boxes_count(const float *boxes, const float *pts, int *counts,
                              int T, int M, fnp_stream_t stream);

/* Dense (T,M) 0/1 membership with the reference's CPU margin 1e-2 and z test
 * (roiaware_pool3d.cpp:121-168, points_in_boxes_cpu) — device-side equivalent. */
int fnp_points_in_boxes_dense(const float *boxes, const float *pts, int *pts_indices,
                              int T, int M, fnp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * iou3d_nms — replaces pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:49-209 and
 * iou3d_nms_kernel.cu (box_overlap :104-225, iou_bev :227-234, iou_normal :327-338,
 * nms_kernel :280-324, nms_normal_kernel :341-385).
 * ------------------------------------------------------------------------------------------ */

/* boxes_overlap_bev_gpu (iou3d_nms.cpp:71-89): a (N,7), b (M,7) -> ans (N,M) overlap area. */
int fnp_boxes_overlap_bev(const float *boxes_a, int num_a, const float *boxes_b, int num_b,
                          float *ans_overlap, fnp_stream_t stream);
/* boxes_iou_bev_gpu (iou3d_nms.cpp:91-111): -> ans (N,M) rotated BEV IoU. */
int fnp_boxes_iou_bev(const float *boxes_a, int num_a, const float *boxes_b, int num_b,
                      float *ans_iou, fnp_stream_t stream);
/* boxes_aligned_overlap_bev_gpu (iou3d_nms.cpp:49-69): a (N,7), b (N,7) -> ans (N,). */
int fnp_boxes_aligned_overlap_bev(const float *boxes_a, const float *boxes_b, int num,
                                  float *ans_overlap, fnp_stream_t stream);
/* Fused boxes_iou3d_gpu (iou3d_nms_utils.py:48-81): BEV overlap x height overlap / union,
 * z = box centre.  ans (N,M). */
int fnp_boxes_iou3d(const float *boxes_a, int num_a, const float *boxes_b, int num_b,
                    float *ans_iou, fnp_stream_t stream);

/* boxes_aligned_iou3d_gpu (iou3d_nms_utils.py:83-117) fused: a (N,7), b (N,7) -> ans (N,) 3D IoU of pair i. */
int fnp_boxes_aligned_iou3d(const float *boxes_a, const float *boxes_b, int num, float *ans_iou, fnp_stream_t stream);

/* Recall bookkeeping of one frame, Detector3DTemplate.generate_recall_record
 * (pcdet/models/detectors/detector3d_template.py:314-399, called per frame by tools/extract_pseudo_labels.py:124),
 * accumulated on the device: counters (5 + 6*num_thresh) int64 DEVICE, layout
 *   [gt, num_3known, num_6known, num_4unknown, num_7unknown] then per threshold
 *   [roi, rcnn, rcnn_3known, rcnn_6known, rcnn_4unknown, rcnn_7unknown]   (added to, never zeroed here).
 * preds: rows of pred_stride floats starting with a (7) box; pred_count: DEVICE float holding the number of live rows
 * (the header cell of an extraction record) or NULL = max_preds.  gt (num_gt, gt_stride >= 8): box (7) ... class label
 * (1-based) in the LAST column; trailing all-zero rows are padding (:342-346).  rois: optional (num_rois, rois_stride).
 * thresh: HOST floats (num_thresh <= 8).  known3_bits / known6_bits: bit l set iff label l is a known class. */
int fnp_recall_counters(const float *preds, int pred_stride, int max_preds, const float *pred_count,
                        const float *gt, int num_gt, int gt_stride, const float *rois, int num_rois, int rois_stride,
                        const float *thresh, int num_thresh, unsigned known3_bits, unsigned known6_bits,
                        int64_t *counters, fnp_stream_t stream);

/* Host-side dense point-in-box test of the pseudo-label mixing (PseudoSampler.points_in_boxes,
 * pcdet/datasets/augmentor/pseudo_loader.py:270-316): points (N,C>=3), boxes (T,7) -> in_box (T,N) u8 with
 * INCLUSIVE faces, and optionally the points in each box frame (T,N,C) (points_out may be NULL). */
int fnp_host_points_in_boxes_frame(const float *points, int n, int C, const float *boxes, int t,
                                   unsigned char *in_box, float *points_out);

/* Host-side rotated BEV IoU (no device, no stream): replaces boxes_iou_bev_cpu (N x M) and
 * boxes_aligned_iou_bev_cpu (N pairs) of pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:232-272, called by the
 * pseudo-label mixing in dataloader workers (pseudo_loader.py:29-55).  Host pointers, (n,7) boxes. */
int fnp_host_boxes_iou_bev(const float *boxes_a, int na, const float *boxes_b, int nb, float *iou_out);
int fnp_host_boxes_aligned_iou_bev(const float *boxes_a, const float *boxes_b, int n, float *iou_out);

/* Bytes of the u64 suppression-mask workspace for N boxes. */
int64_t fnp_nms_workspace_bytes(int num_boxes);
/* nms_gpu / nms_normal_gpu (iou3d_nms.cpp:113-209): boxes (N,7) pre-sorted by score desc.
 * keep (N,) int64 DEVICE, num_keep (1,) int32 DEVICE.  The greedy sweep that the reference
 * runs on the host after a synchronous cudaMemcpy (iou3d_nms.cpp:139-155,189-205) runs on
 * the device, so the call is asynchronous. */
int fnp_nms_rotated(const float *boxes, int num_boxes, float thresh, void *workspace,
                    int64_t *keep, int *num_keep, fnp_stream_t stream);
int fnp_nms_normal(const float *boxes, int num_boxes, float thresh, void *workspace,
                   int64_t *keep, int *num_keep, fnp_stream_t stream);
/* (ABI 12) Several score-sorted lists in one launch pair — the per-class NMS of multi_classes_nms (model_nms_utils.py:30-66): list z holds
 * counts[z] (a DEVICE array, each <= cap) boxes at boxes + z * cap * 7; keep (lists, cap) int64 and num_keep (lists) int32 receive,
 * per list, what fnp_nms_rotated (rotated != 0) / fnp_nms_normal gives for its first counts[z] boxes.  workspace:
 * fnp_nms_batched_workspace_bytes(lists, cap).  No host synchronisation. */
int64_t fnp_nms_batched_workspace_bytes(int lists, int cap);
int fnp_nms_batched(const float *boxes, const int *counts, int lists, int cap, float thresh, int rotated, void *workspace,
                    int64_t *keep, int *num_keep, fnp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Rank grid — the voxel index behind voxelisation and rulebook building.  A grid of
 * (B, D, H, W) cells is cut into 4x4x4 blocks; block w owns one u64 occupancy word
 * (bit = (z&3)*16 + (y&3)*4 + (x&3)) and one u32 exclusive popcount prefix, so
 *     row(cell) = base[w] + popc(bits[w] & below(bit))
 * is a collision-free lookup (at most 8 words for a 3x3x3 neighbourhood).  A second level,
 * one summary bit per block, lets the prefix scan, the coordinate emission and the clearing
 * visit only occupied blocks (a 41x1440x1440 lidar grid is ~1.5 % occupied at block level).
 * Replaces spconv's hash / direct table (call sites pcdet/utils/spconv_utils.py:3-10).
 * ------------------------------------------------------------------------------------------ */
typedef struct fnp_rankgrid {
    int B, D, H, W;     /* cells */
    uint64_t *bits;     /* (nblk)  occupancy words; all zero before a build */
    uint32_t *base;     /* (nblk)  exclusive popcount prefix, defined where bits != 0 */
    uint64_t *summary;  /* (nsum)  bit j of word i set iff bits[64*i + j] != 0; zero before a build */
    int *perm;          /* (cap)   rank -> row, or NULL when rows are stored in rank order */
    uint32_t *counters; /* (ABI 13, nullable) fnp_rankgrid_counter_words() words, ALL ZERO before a build: the marking kernels of
                         * fnp_voxelize / fnp_rulebook_strided add the cells they set to per-unit / per-group / per-chunk counters
                         * here (one atomic per workgroup and counter), and the rank prefix is ONE launch instead of three (count
                         * pass, totals, prefix pass).  The fnp_rankgrid_clear* calls zero them again.  NULL: the three-launch prefix. */
} fnp_rankgrid;

int64_t fnp_rankgrid_num_blocks(int B, int D, int H, int W);   /* nblk */
int64_t fnp_rankgrid_num_summary(int B, int D, int H, int W);  /* nsum = ceil(nblk / 64) */
int64_t fnp_rankgrid_counter_words(int B, int D, int H, int W); /* (ABI 13) uint32 words of fnp_rankgrid.counters */
int64_t fnp_rankgrid_workspace_bytes(int B, int D, int H, int W);

/* Index an existing coordinate list (N,4) [b,z,y,x] (e.g. a SparseConvTensor built from user
 * tensors): sets bits/summary/base and, when grid->perm != NULL, perm[rank] = row. */
int fnp_rankgrid_build(const int *coords, const int *n_rows, int cap, const fnp_rankgrid *grid,
                       void *workspace, int64_t workspace_bytes, fnp_stream_t stream);

/* Zero the occupancy + summary words touched by `coords` (O(rows) instead of O(grid)). */
int fnp_rankgrid_clear(const int *coords, const int *n_rows, int cap, const fnp_rankgrid *grid,
                       fnp_stream_t stream);

/* fnp_rankgrid_clear for up to 8 grids in one launch (HOST arrays of `count` device pointers / capacities / grids). */
int fnp_rankgrid_clear_multi(int count, const int *const *coords, const int *const *n_rows, const int *caps,
                             const fnp_rankgrid *grids, fnp_stream_t stream);

/* (ABI 13) Zero up to 8 grids through their SUMMARY level in one launch: every occupancy word a summary bit names, the summary
 * words, the counters.  No coordinate list: whatever was marked goes (cells of dropped voxels and of rows beyond a capacity too). */
int fnp_rankgrid_clear_summary(int count, const fnp_rankgrid *grids, fnp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Voxelisation + MeanVFE — replaces spconv.utils.Point2VoxelCPU3d.point_to_voxel as called
 * at pcdet/datasets/processor/data_processor.py:38-61 and MeanVFE.forward
 * (pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31).
 * ------------------------------------------------------------------------------------------ */
typedef struct fnp_voxel_cfg {
    float range_min[3];  /* x,y,z of POINT_CLOUD_RANGE[0:3] */
    float voxel_size[3]; /* x,y,z */
    int grid[3];         /* x,y,z = round((max-min)/voxel_size), data_processor.py:257-258 */
    int num_features;    /* C of points (N,C); xyz are columns 0..2 */
    int max_points;      /* MAX_POINTS_PER_VOXEL (10) */
    int max_voxels;      /* MAX_NUMBER_OF_VOXELS per scene */
} fnp_voxel_cfg;

/* The rank grid the voxels are indexed in may be larger than the voxel grid
 * {cfg.grid[2], cfg.grid[1], cfg.grid[0]} (the backbone's sparse_shape adds one z layer,
 * spconv_backbone.py:191); grid->B is the number of scenes. */
int64_t fnp_voxelize_workspace_bytes(int64_t n_points, const fnp_voxel_cfg *cfg, const fnp_rankgrid *grid);

/* A frame into the STATIC inputs of a captured forward (a hipGraph reads fixed addresses), one launch: dst_points[0, n) = points,
 * dst_points[n, n_prev) = pad in every column (rows the previous frame filled: pad lies outside every range, the voxeliser drops such
 * points), dst_offsets = batch_offsets (n_offsets words).  All device pointers; (n, num_features) f32 rows, contiguous.
 * The reference's loops hand a frame to the model through load_data_to_gpu (pcdet/models/__init__.py:23-37). */
int fnp_stage_points(const float *points, int64_t n_points, int64_t n_prev_points, int num_features, float pad, float *dst_points,
                     const int *batch_offsets, int n_offsets, int *dst_offsets, fnp_stream_t stream);

/* points (N,C) f32, scenes concatenated; batch_offsets (B+1,) int32 device (scene b owns
 * points [off[b], off[b+1])).  Outputs, all capacity `cap` rows (cap >= N is always enough):
 *   coords (cap,4) int32 [b,z,y,x] in the sequential first-come order of the reference,
 *   num_points (cap,) int32, mean_feats (cap,C) f32 = MeanVFE, voxels (cap,max_points,C) f32
 *   zero padded (nullable), n_voxels (1,) int32 device.
 * The grid (bits/summary zero on entry, perm != NULL) is left describing the voxels for the first
 * rulebook: perm[rank] = voxel row, -1 for a cell whose voxel fell to the max_voxels cut.
 * n_cells (1,) int32 device, nullable: number of occupied cells (>= n_voxels); rows
 * [n_voxels, n_cells) of coords then list the cells of the dropped voxels in no particular order, so
 * that fnp_rankgrid_clear(coords, n_cells, ...) wipes every word the voxeliser set in a persistent grid. */
int fnp_voxelize(const float *points, int n_points, const int *batch_offsets,
                 const fnp_voxel_cfg *cfg, const fnp_rankgrid *grid,
                 void *workspace, int64_t workspace_bytes,
                 int *coords, int *num_points, float *mean_feats, float *voxels,
                 int *n_voxels, int *n_cells, int cap, fnp_stream_t stream);

/* HOST voxeliser (no device, no stream; host pointers): spconv.utils.Point2VoxelCPU3d.point_to_voxel as DataProcessor
 * calls it inside DataLoader worker processes (data_processor.py:38-61,255-302), where a GPU context cannot be created
 * after a fork.  One scene: points (n, C) -> voxels (max_rows, max_points, C) zero padded, coords (max_rows, 3) [z,y,x],
 * num_points (max_rows); rows = first-come order, capped at min(cfg->max_voxels, max_rows).  Returns the number of
 * voxels or a negative error code.  Results equal fnp_voxelize's bit for bit. */
int fnp_host_voxelize(const float *points, int n_points, const fnp_voxel_cfg *cfg, float *voxels, int *coords,
                      int *num_points, int max_rows);

/* ------------------------------------------------------------------------------------------
 * Rulebooks — replace spconv's indice-pair generation for SubMConv3d / SparseConv3d
 * (call sites pcdet/models/backbones_3d/spconv_backbone.py:12-17,39-46,193-234).
 * Output-stationary layout: nbr[k*cap + o] = input row feeding output row o through kernel
 * offset k, or -1.
 * ------------------------------------------------------------------------------------------ */
typedef struct fnp_conv_geom {
    int ksize[3];   /* kD,kH,kW */
    int stride[3];
    int padding[3];
    int in_shape[3];  /* D,H,W of the input grid */
    int out_shape[3]; /* D,H,W of the output grid ( = in_shape for SubM ) */
} fnp_conv_geom;

/* SubM: outputs = inputs, same order. */
int fnp_rulebook_subm(const int *coords, const int *n_rows, int cap, const fnp_conv_geom *geom,
                      const fnp_rankgrid *grid, int *nbr, fnp_stream_t stream);

/* Strided SparseConv3d: builds the output rank grid (bits/summary zero on entry; perm unused),
 * the output coordinate list (rows in rank-grid order: spatially blocked, deterministic) and
 * nbr (nullable: only grid + coordinates, for fnp_spconv_forward_strided).  workspace:
 * fnp_rankgrid_workspace_bytes of the output grid. */
int fnp_rulebook_strided(const int *in_coords, const int *n_in, int cap_in, const fnp_conv_geom *geom,
                         const fnp_rankgrid *in_grid, const fnp_rankgrid *out_grid,
                         int *out_coords, int *n_out, int cap_out, int *nbr,
                         void *workspace, int64_t workspace_bytes, fnp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sparse convolution forward (implicit GEMM, output-stationary, no atomics) with the
 * BatchNorm1d(eval) + residual + ReLU epilogue of spconv_backbone.py:51-67 fused in.
 *   feat_in  (n_in_rows, Cin) in_dtype  weight (K, Cout, Cin) packed, w_dtype = in_dtype
 *   feat_out (n_out, Cout) out_dtype    residual (n_out, Cout) out_dtype or NULL
 *   scale/shift (Cout,) f32 or NULL (identity).  out = act(acc*scale + shift + residual)
 * bf16 x bf16 -> fp32 accumulate runs on MFMA (v_mfma_f32_16x16x32_bf16); f32 x f32 runs on the f32 MFMA
 * (v_mfma_f32_16x16x4_f32: bit for bit the offset-ascending, channel-ascending fmaf chain of the CPU oracle) for the
 * backbone's channel pairs, on VALU fma chains of the same order otherwise.
 * hints: performance only, never changes results.  FNP_HINT_ROWS_RANKED states that neighbour row
 * ids lie close to the output row ids (input and output rows both in rank-grid order, i.e. a
 * SubM convolution on the output of fnp_rulebook_strided): the kernel then keeps a window of
 * input rows in LDS instead of gathering every (site, offset) pair from L2.
 * ------------------------------------------------------------------------------------------ */
#define FNP_HINT_ROWS_RANKED 1
/* f32 only: take the thread-per-element fmaf chain (validation path) instead of the f32 MFMA kernel; the two are
 * bit-identical, this only selects which one computes */
#define FNP_HINT_VALU 2
/* f32 MFMA path only: `weight` is stored with every group of 16 input channels transposed 4 x 4 — position 4q + r of a
 * group holds channel 4r + q (q, r in 0..3) — so that a lane's 16-byte load holds its channels of four consecutive
 * MFMA steps.  Layout statement, not a numerical option: results are bit-identical to the plain layout.  Shapes the
 * f32 MFMA kernel does not cover return FNP_ERR_ARG with this hint (the other kernels read the plain layout). */
#define FNP_HINT_W_PERMUTED 4
int fnp_spconv_forward(const void *feat_in, int in_dtype, int n_in_rows, const void *weight,
                       const int *nbr, int nbr_stride, int K,
                       const int *n_out, int cap_out,
                       void *feat_out, int out_dtype,
                       const float *scale, const float *shift, const void *residual, int relu,
                       int hints, int Cin, int Cout, fnp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The same convolution on a TILE RULEBOOK, for the ranked 16-bit 32 -> 32 and 64 -> 64 layers of 3x3x3 kernels (the SubM
 * convolutions of stages 2 and 3 of VoxelResBackBone8x, spconv_backbone.py:210-219: one rulebook per stage, indice_keys
 * 'subm2' / 'subm3', each used four times per forward).  fnp_tile_rulebook_build restates the (27, cap) int32 table once,
 * per tile of FNP_TILE_ROWS (32 channels) / FNP_TILE64_ROWS (64 channels) output rows, as 16-bit addresses into an LDS
 * image of the tile's neighbourhood (a window of input rows around the tile + up to 256 / 128 far rows, deduplicated) —
 * FNP_TILE_RECORD_BYTES / FNP_TILE64_RECORD_BYTES per tile, 58 bytes per row instead of 108 — and
 * fnp_spconv_forward_tiled sweeps the offsets from LDS alone (64 channels: with the weight slabs streamed).  Bit-identical to fnp_spconv_forward on the int32
 * table for ANY row order; the tiled form is the faster one when rows are in rank-grid order (FNP_HINT_ROWS_RANKED's
 * condition): with the tile rulebook written by fnp_rulebook_subm_tiled, four tiled convolutions + that rulebook were ahead
 * of four gather convolutions + fnp_rulebook_subm from 1 to 64 scenes of the backbone (1-4 % at 1-3 scenes, 10 % at 32-64).
 *   tile_rb  fnp_tile_rulebook_bytes(cap_out, channels) bytes, 16-byte aligned; valid for the (nbr, n_out) it was built from
 *   nbr      the int32 table itself: read only for entries the tile record could not hold (more distinct far rows in a tile than its
 *            image has overflow rows: arbitrary row orders)
 * K must be 27, Cin == Cout == 32 or 64 (the channel count the tile rulebook was built for), dtype FNP_BF16 or FNP_F16 (features, weights, residual and output alike);
 * FNP_ERR_ARG otherwise, and for tensors beyond 32-bit byte offsets.
 * ------------------------------------------------------------------------------------------ */
#define FNP_TILE_ROWS 256            /* 32 channels */
#define FNP_TILE_RECORD_BYTES 14864
#define FNP_TILE64_ROWS 128          /* 64 channels */
#define FNP_TILE64_RECORD_BYTES 7440
/* Diagnostic: the 32-channel kernel hands tile images between its producer and consumer waves through counters in LDS; a
 * wait that times out (2^20 polls, tens of milliseconds; never, unless that protocol is broken) ends the workgroup with
 * wrong output and counts here.  fnp_spconv_tiled_aborts synchronises the device (>= 0, or a negative error code);
 * fnp_spconv_tiled_aborts_copy enqueues a device-to-device copy of the counter (one int32) on `stream`, for a caller that
 * reads it with its other per-forward counts (the host layer raises when it has grown); fnp_debug_tile_hold(1) is the
 * test hook that makes every hand-over time out (producers stop publishing), fnp_debug_tile_hold(0) restores them. */
int fnp_spconv_tiled_aborts(void);
int fnp_spconv_tiled_aborts_copy(int *dst, fnp_stream_t stream);
/* dst[i] = *srcs[i] for n <= 16 one-word device counters (srcs: HOST array of device pointers), the time-out counter above where
 * srcs[i] is NULL: what a forward hands to the host in its one synchronisation, in one launch (capturable).  Bit i of reset_mask:
 * *srcs[i] = 0 after it is read (a counter its owner keeps across forwards, e.g. fnp_rulebook_ell's pool_used). */
int fnp_gather_counts(int *const *srcs, int n, unsigned reset_mask, int *dst, fnp_stream_t stream);
/* The same launch, handing the counts to the HOST itself (ABI 14): besides dst (device) it stores them into host_dst — PINNED host
 * memory of >= FNP_COUNTS_SEQ_SLOT + 1 words, addressed by the device through its host pointer — and then, behind a system-scope fence,
 * ++*seq (a device word the caller zeroed once) into host_dst[FNP_COUNTS_SEQ_SLOT].  A host thread that has counted its launches polls that
 * word and reads the counts when it arrives: no event, no device-to-host copy, and the launch may sit in the MIDDLE of a hipGraph — the
 * captured forward is one graph, its counts leave where they are final (in front of the last SubM stage) and the host sizes the
 * outputs while the rest of the graph runs.  The reference synchronises once per forward where spconv reads its pair counts
 * (pcdet/models/backbones_3d/spconv_backbone.py:193-234 through spconv's indice-pair call). */
#define FNP_COUNTS_SEQ_SLOT 16
int fnp_gather_counts_host(int *const *srcs, int n, unsigned reset_mask, int *dst, int *host_dst, unsigned *seq, fnp_stream_t stream);
int fnp_debug_tile_hold(int on);
long long fnp_tile_rulebook_bytes(int cap_out, int channels);
int fnp_tile_rulebook_build(const int *nbr, int nbr_stride, int K, const int *n_out, int cap_out, int channels,
                            void *tile_rb, fnp_stream_t stream);
/* fnp_rulebook_subm for a 3x3x3 kernel that writes the tile rulebook (for `channels` = 32 or 64) in the same pass: what
 * fnp_rulebook_subm followed by fnp_tile_rulebook_build leaves, without reading the table back. */
int fnp_rulebook_subm_tiled(const int *coords, const int *n_rows, int cap, const fnp_conv_geom *geom,
                            const fnp_rankgrid *grid, int *nbr, int channels, void *tile_rb, fnp_stream_t stream);
/* The same, but the int32 table receives only the rows of tiles whose record holds an escape entry (the only rows
 * fnp_spconv_forward_tiled ever looks up in it; ~1e-5 of the tiles of a rank-ordered tensor): for a caller whose every
 * consumer of this rulebook is fnp_spconv_forward_tiled — the fused inference backbone; the int32 table was 108 of the 166
 * bytes per row this kernel stores.  nbr must still be a (27, cap) buffer; rows outside such tiles stay unwritten. */
int fnp_rulebook_subm_tiled_lean(const int *coords, const int *n_rows, int cap, const fnp_conv_geom *geom,
                                 const fnp_rankgrid *grid, int *nbr, int channels, void *tile_rb,
                                 const fnp_rankgrid *mark_grid, const fnp_conv_geom *mark_geom, int *escape_groups,
                                 fnp_stream_t stream);
/* escape_groups (ABI 10, nullable): a device counter that receives += the number of 32-row groups whose entries hold an escape:
 * the statistic the tiled convolutions' advantage rests on (a group with an escape fetches through the int32 table).  Measured
 * on the synthetic scenes: ~1e-5 of the groups at single-sweep density (tiled kernels 1.15-2.1x the gather kernels), 2 % / 11 %
 * (32 / 64 channels) at the 10-sweep density of transfusion_lidar.yaml (tiled kernels 4-8 % SLOWER): the fused engine reads
 * the counter with its per-forward counts and runs a stage on the gather kernels while it exceeds 0.4 %. */
/* mark_grid / mark_geom (both NULL: none; also on fnp_rulebook_subm_masked): the rows whose rulebook is built are the input
 * sites of the NEXT strided convolution (mark_geom, in_shape = their grid); while the kernel has their coordinates in its
 * registers it also marks that convolution's output sites in mark_grid (all zero on entry), so that
 * fnp_rulebook_strided_premarked — fnp_rulebook_strided without its marking launch — can follow: the separate pass over the
 * coordinates, a chain of dependent loads and atomics per wave, goes.  Geometries with at most two outputs per input cell and
 * axis (ceil(k / s) <= 2); FNP_ERR_ARG otherwise. */
int fnp_rulebook_strided_premarked(const int *in_coords, const int *n_in, int cap_in, const fnp_conv_geom *geom,
                                   const fnp_rankgrid *in_grid, const fnp_rankgrid *out_grid, int *out_coords, int *n_out,
                                   int cap_out, int *nbr, void *workspace, int64_t workspace_bytes, fnp_stream_t stream);
int fnp_spconv_forward_tiled(const void *feat_in, int dtype, int n_in_rows, const void *weight,
                             const void *tile_rb, const int *nbr, int nbr_stride,
                             const int *n_out, int cap_out, void *feat_out,
                             const float *scale, const float *shift, const void *residual, int relu,
                             int Cin, int Cout, fnp_stream_t stream);

/* f32 rows -> two bf16 tensors hi = bf16(x), lo = bf16(x - hi) (ABI 10): hi + lo equals x to 2^-17 of |x|.  The activation
 * format of the fused backbone's "bf16x3" precision (FNP_DTYPE: bf16x3): every convolution is three fnp_spconv_forward launches
 * with f32 outputs chained through `residual` — lo x W_hi, hi x W_lo, hi x W_hi — on the bf16 matrix pipe, the f32 result to
 * ~1e-5 relative.  Rows >= *n_rows are not touched; C a multiple of 4. */
int fnp_split_bf16(const float *x, const int *n_rows, int cap_rows, int C, void *hi, void *lo, fnp_stream_t stream);
/* The same behind y = x + float(t) (t: a bf16 tensor of the same shape; then ReLU if relu != 0), y written as f32 (y == x allowed):
 * where the two cross terms of a bf16x3 convolution were summed in bf16 by the fast bf16-out kernels (they are 2^-8 of the result
 * and need bf16 precision only) and only the main product ran with an f32 output. */
int fnp_split_bf16_add(const float *x, const void *t, int relu, const int *n_rows, int cap_rows, int C, float *y, void *hi,
                       void *lo, fnp_stream_t stream);

/* The bf16x3 engine's MAIN product with the sum and the split in its epilogue (ABI 11): fnp_spconv_forward with 16-bit features and
 * weights and an f32 output,
 *     y = act( conv * scale + shift + residual (f32, nullable) + float(addend) (16-bit rows of the features' dtype, nullable) ),
 * written as f32 rows (feat_out) AND as their split out_hi = 16bit(y), out_lo = 16bit(y - out_hi) — what fnp_spconv_forward (f32 out,
 * no ReLU) followed by fnp_split_bf16_add computes, bit for bit, without the pass over the rows in between (it was 12 % of a
 * bf16x3 forward).  feat_out may be NULL: only the split is written (a layer whose f32 rows nobody reads).  Shapes of the matrix
 * kernel only (FNP_ERR_ARG otherwise).  fnp_spconv_forward_tiled_split: the same on the
 * tile rulebook (fnp_spconv_forward_tiled's arguments and conditions; Cin == Cout == 32 or 64); fnp_spconv_forward_sorted_split:
 * the same as the class-sorted sweep of the 128 -> 128 layers (fnp_spconv_forward_sorted's arguments and conditions). */
int fnp_spconv_forward_split(const void *feat_in, int in_dtype, int n_in_rows, const void *weight, const int *nbr, int nbr_stride,
                             int K, const int *n_out, int cap_out, float *feat_out, const float *scale, const float *shift,
                             const float *residual, const void *addend, int relu, int hints, int Cin, int Cout, void *out_hi,
                             void *out_lo, fnp_stream_t stream);
int fnp_spconv_forward_sorted_split(const void *feat_in, int dtype, int n_in_rows, const void *weight, const int *nbr, int nbr_stride,
                                    const int *perm, const unsigned *blockmask, const int *n_out, int cap_out, float *feat_out,
                                    const float *scale, const float *shift, const float *residual, const void *addend, int relu,
                                    int Cin, int Cout, void *out_hi, void *out_lo, fnp_stream_t stream);
int fnp_spconv_forward_tiled_split(const void *feat_in, int dtype, int n_in_rows, const void *weight, const void *tile_rb,
                                   const int *nbr, int nbr_stride, const int *n_out, int cap_out, float *feat_out,
                                   const float *scale, const float *shift, const float *residual, const void *addend, int relu,
                                   int Cin, int Cout, void *out_hi, void *out_lo, fnp_stream_t stream);

/* COMPACT RULEBOOK for the sparse-neighbourhood layers (conv_input 5 -> 16, the four 16 -> 16 SubM layers, the strided
 * 16 -> 32 layer: spconv_backbone.py:193-210).  A stage-1 voxel has 3.6 of its 27 neighbours, an output site of the first
 * strided layer 2.1 of 27 inputs: the (27, cap) int32 table spends 108 bytes per row on that and the matrix kernel a gather
 * and a matrix step per (16-row block, offset).  fnp_rulebook_ell writes 32 bytes per row instead —
 *     record = 8 x uint32: (k << 27) | input row, ascending k; 0xFFFFFFFF = empty; slot 7 may be (31 << 27) | e = link to
 *     extension record e of a pool behind the cap_rows row records (rows with more than 8 neighbours)
 * — and fnp_spconv_forward_ell sums sum_k W_k^T x over the entries that exist, on the VALU (v_dot2c_f32_bf16 /
 * v_dot2_f32_f16 for 16-bit rows; the oracle's fmaf chain, bit for bit, for the f32 point features of conv_input), with the
 * BatchNorm(eval) scale / shift, residual and ReLU epilogue of fnp_spconv_forward.
 *   records    fnp_ell_bytes(cap_rows, pool_records) bytes, 16-byte aligned
 *   coords     the rows' cells: the tensor's own coordinates (SubM: geom with in_shape == out_shape, stride 1, padding 1) or
 *              the output coordinates of fnp_rulebook_strided(nbr = NULL) (strided 3x3x3: geom as given there)
 *   pool_used  one int32 (device): ends as the number of extension records the rows asked for; above pool_records the
 *              chains were cut and the caller must discard the result and come back with a larger pool.
 *              pool_used_is_zero != 0: the word already holds zero (a counter the caller keeps and has reset when it is read,
 *              fnp_gather_counts): the call does not spend a launch on clearing it
 *   nbr        optional (27, cap) int32 table of the same rows (fnp_rulebook_subm's / fnp_rulebook_strided's), written in the
 *              same pass: measured on MI355X the VALU convolution wins 2.3x on conv_input (5 input channels) and loses 30 %
 *              on the 16-channel layers (256 products per pair are v_dot2c work at a quarter of the FMA rate), so the
 *              backbone reads the records in conv_input and keeps the table for the four 16 -> 16 layers
 * fnp_spconv_forward_ell: K = 27; (in_dtype FNP_F32, Cin 4 or 5, Cout 16, any out_dtype) or (in_dtype = out_dtype = FNP_BF16
 * / FNP_F16, Cin 16, Cout 16 or 32); FNP_ERR_ARG otherwise.  weight: packed (27, Cout, Cin) in in_dtype. */
long long fnp_ell_bytes(int cap_rows, int pool_records);
int fnp_rulebook_ell(const int *coords, const int *n_rows, int cap, const fnp_conv_geom *geom, const fnp_rankgrid *in_grid,
                     void *records, int pool_records, int *pool_used, int pool_used_is_zero, int *nbr, fnp_stream_t stream);
int fnp_spconv_forward_ell(const void *feat_in, int in_dtype, int n_in_rows, const void *weight, const void *records,
                           int cap_rows, int pool_records, const int *n_out, void *feat_out, int out_dtype,
                           const float *scale, const float *shift, const void *residual, int relu, int Cin, int Cout,
                           fnp_stream_t stream);
/* The same call for the 16-input-channel layers (Cin 16, Cout 16 or 32, FNP_BF16 / FNP_F16 in = out) ON THE MATRIX PIPE (ABI 8):
 * the MFMA kernel of fnp_spconv_forward with the entries of a tile's rows expanded from the records into LDS at the top of
 * the tile — no (27, cap) table is written or read (108 of the ~170 bytes a row of those layers moves), no entry load shares
 * the gathers' in-order queue.  Values: fnp_spconv_forward's on the table of the same rows, bit for bit. */
int fnp_spconv_forward_ell_mfma(const void *feat_in, int in_dtype, int n_in_rows, const void *weight, const void *records,
                                int cap_rows, int pool_records, const int *n_out, void *feat_out, int out_dtype,
                                const float *scale, const float *shift, const void *residual, int relu, int Cin, int Cout,
                                fnp_stream_t stream);

/* CLASS-SORTED sweep of the 128 -> 128 SubM layers (the four 3x3x3 convolutions of stage 4, spconv_backbone.py:219-224).
 * After three stride-2 layers a lidar surface is two cells thick: ~36 % of the stage-4 sites have neighbours only in the
 * plane above, ~36 % only in the plane below.  fnp_rulebook_classsort orders the rows each persistent workgroup of the
 * convolution sweeps concurrently by that class (once per forward; the four convolutions share it): perm[position] = row, and
 * blockmask[position / 16] = union of the 27-bit neighbour masks of 16 consecutive positions.  fnp_spconv_forward_sorted
 * then sweeps, per tile, only the kernel offsets some row of the tile has a neighbour at (on lidar scenes 20 % of the
 * (tile, offset) pairs go, with their slab loads, barriers, gathers and matrix work); every row still sums its own
 * neighbours in ascending offset order, so the values equal fnp_spconv_forward's.
 * The convolution's workgroups that share an XCD (blocks b, b + 8, ...) own one contiguous run of rows together and take
 * its tiles round-robin; the sort orders the rows of each ROUND of tiles, so that the tiles in flight on an XCD cover one
 * contiguous region of the feature map (its L2) and all but the two or three tiles at the class boundaries are of one class.
 *   rowmask    (cap_out) uint32, bit k = row has a neighbour at offset k: written by fnp_rulebook_subm_masked (the SubM
 *              rulebook kernel with that one extra store per row); NULL: classsort derives it from nbr into `workspace`
 *              (fnp_classsort_workspace_bytes(cap_out) bytes)
 *   perm       (cap_out) int32, blockmask (cap_out / 16 + 1) uint32: written by classsort, read by the convolution; valid
 *              for the (nbr, n_out, cap_out) they were built from
 * K must be 27, (Cin, Cout) = (128, 128), dtype FNP_BF16 or FNP_F16 for features, weights, residual and output alike;
 * FNP_ERR_ARG otherwise. */
long long fnp_classsort_workspace_bytes(int cap_out);
int fnp_rulebook_subm_masked(const int *coords, const int *n_rows, int cap, const fnp_conv_geom *geom,
                             const fnp_rankgrid *grid, int *nbr, unsigned *rowmask,
                             const fnp_rankgrid *mark_grid, const fnp_conv_geom *mark_geom, fnp_stream_t stream);
int fnp_rulebook_classsort(const int *nbr, int nbr_stride, int K, const unsigned *rowmask, const int *n_out, int cap_out, int Cin, int Cout,
                           int *perm, unsigned *blockmask, void *workspace, long long workspace_bytes, fnp_stream_t stream);
int fnp_spconv_forward_sorted(const void *feat_in, int dtype, int n_in_rows, const void *weight, const int *nbr, int nbr_stride,
                              const int *perm, const unsigned *blockmask, const int *n_out, int cap_out, void *feat_out,
                              const float *scale, const float *shift, const void *residual, int relu, int Cin, int Cout,
                              fnp_stream_t stream);

/* The f32 engine's form of the class sort (the f32 3x3x3 SubM layers, 16 / 32 / 64 / 128 channels, run on v_mfma_f32_16x16x4_f32
 * and are bound by the matrix pipe; the kernel skips the MFMAs of a (16-row block, offset) pair without any neighbour).
 * fnp_rulebook_classsort_f32 orders the rows of every workgroup range of that kernel's grid by class (perm (cap_out) int32;
 * rowmask as written by fnp_rulebook_subm_masked), fnp_spconv_forward_f32_sorted sweeps the ranges in that order.  Cin == Cout;
 * wperm != 0: weight in the 4 x 4-transposed layout (FNP_HINT_W_PERMUTED).  Bit-identical to fnp_spconv_forward on f32. */
int fnp_rulebook_classsort_f32(const unsigned *rowmask, const int *n_out, int cap_out, int Cin, int Cout, int *perm, fnp_stream_t stream);
int fnp_spconv_forward_f32_sorted(const void *feat_in, int n_in_rows, const void *weight, const int *nbr, int nbr_stride, const int *perm,
                                  const int *n_out, int cap_out, void *feat_out, const float *scale, const float *shift,
                                  const void *residual, int relu, int wperm, int Cin, int Cout, fnp_stream_t stream);

/* The same convolution for a STRIDED 3x3x3 layer whose rulebook has no other user (the three down-sampling
 * layers of VoxelResBackBone8x, spconv_backbone.py:207,214,221): the kernel computes the rulebook rows of its
 * tiles itself from the input rank grid and the output coordinates (fnp_rulebook_strided with nbr = NULL builds
 * those), so no (27, cap) table is written and read back.  bf16 in/out, (Cin, Cout) in {(16,32), (32,64),
 * (64,128)}; FNP_ERR_ARG otherwise (take the table path).  Results are identical to the table path. */
int fnp_spconv_forward_strided(const void *feat_in, int in_dtype, int n_in_rows, const void *weight,
                               const fnp_rankgrid *in_grid, const fnp_conv_geom *geom, const int *out_coords,
                               const int *n_out, int cap_out, void *feat_out, int out_dtype,
                               const float *scale, const float *shift, int relu, int Cin, int Cout,
                               fnp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Backward of the sparse convolution (spconv's autograd behind SubMConv3d / SparseConv3d in the
 * self-training step, tools/train_st.py; modules at spconv_backbone.py:12-17,39-46).
 *   dgrad: dx[i] = sum_k W_k dy[nbrT[k][i]] = fnp_spconv_forward on the transposed rulebook with the
 *          transposed slabs (weight (K, Cin, Cout) passed as (K, "Cout"=Cin, "Cin"=Cout)).
 *   fnp_rulebook_transpose: nbr (K, nbr_stride) over output rows -> nbr_t (K, cap_in) over input rows
 *          (nbr_t[k][i] = o iff nbr[k][o] = i, else -1).
 *   fnp_spconv_wgrad: grad_weight (K, Cout, Cin) f32 = sum_o grad_out[o] (x) feat_in[nbr[k][o]];
 *          deterministic (per-chunk partials in `workspace`, added in chunk order).
 * ------------------------------------------------------------------------------------------ */
int fnp_rulebook_transpose(const int *nbr, int nbr_stride, int K, const int *n_out, int cap_out,
                           int *nbr_t, int cap_in, fnp_stream_t stream);
int64_t fnp_spconv_wgrad_workspace_bytes(int K, int Cin, int Cout);
int fnp_spconv_wgrad(const void *feat_in, int in_dtype, const void *grad_out, int grad_dtype,
                     const int *nbr, int nbr_stride, int K, const int *n_out, int cap_out,
                     float *grad_weight, int grad_layout, int Cin, int Cout,
                     void *workspace, int64_t workspace_bytes, fnp_stream_t stream);
/* grad_layout (ABI 9): 0 = grad_weight is the packed (K, Cout, Cin), 1 = the module parameter's (Cout, K, Cin) (the final
 * reduction writes the .grad tensor itself: no permute-copy per layer afterwards).
 * fnp_pack_weight: the module's f32 weight (Cout, K, Cin) -> the packed (K, Cout, Cin) slabs in `dtype`, and in the same launch
 * the slabs the data gradient of a SubM layer reads: mirror_mode 1 = (K-1-k, Cout, Cin) for fnp_spconv_dgrad on the forward's
 * table (which transposes them itself), 2 = (K-1-k, Cin, Cout): the forward kernel run on the gradient of a SubM layer, 3 =
 * (k, Cin, Cout): the same for a strided layer on its transposed table (fnp_rulebook_transpose); 0 = none (mirror NULL). */
int fnp_pack_weight(const float *weight, int Cout, int K, int Cin, int dtype, void *packed, void *mirror, int mirror_mode,
                    fnp_stream_t stream);
/* the same for count <= 32 layers in one launch (host arrays of device pointers and shapes; one dtype for all) */
int fnp_pack_weight_multi(int count, const float *const *weights, const int *couts, const int *ks, const int *cins, int dtype,
                          void *const *packed, void *const *mirrors, const int *mirror_modes, fnp_stream_t stream);
/* PAIR LISTS of a rulebook (ABI 8), for the weight gradient: offset k only sums over the output rows that HAVE a neighbour
 * at k (44-54 % of the rows on lidar scenes); fnp_rulebook_pairs compacts every column of the table once per rulebook —
 * pair_o[k][j] = the j-th such output row (ascending), pair_i[k][j] = its neighbour, pair_count[k] — and
 * fnp_spconv_wgrad_pairs runs fnp_spconv_wgrad's sum over them (16-bit features and gradients, the MFMA channel pairs;
 * FNP_ERR_ARG otherwise: take fnp_spconv_wgrad).  pair_o / pair_i: (K, pair_stride) int32, pair_stride >= cap_out = the rows the
 * caller knows to exist (not the table's stride: a strided layer's table is sized for 27 outputs per input); order-preserving. */
int64_t fnp_rulebook_pairs_workspace_bytes(int K, int cap_out);
int fnp_rulebook_pairs(const int *nbr, int nbr_stride, int K, const int *n_out, int cap_out, int *pair_o, int *pair_i,
                       int pair_stride, int *pair_count, void *workspace, int64_t workspace_bytes, fnp_stream_t stream);
int fnp_spconv_wgrad_pairs(const void *feat_in, int in_dtype, const void *grad_out, int grad_dtype, const int *pair_o,
                           const int *pair_i, const int *pair_count, int pair_stride, int K, const int *n_out, int cap_out,
                           float *grad_weight, int grad_layout, int Cin, int Cout, void *workspace, int64_t workspace_bytes,
                           fnp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm1d in TRAINING mode on sparse rows, fused with the ReLU / residual add that follow it in post_act_block and
 * SparseBasicBlock (pcdet/models/backbones_3d/spconv_backbone.py:8-27,51-67; norm_fn = BatchNorm1d(eps 1e-3, momentum
 * 0.01), :189) — forward and backward of the dense part of the self-training step (tools/train_st.py).
 *   x, residual, y, grad_* : (cap, C) rows of `dtype` (f32 / bf16 / fp16), C in {8, 16, 32, 64, 128, 256}; the n_rows valid
 *   rows (DEVICE scalar) form the batch.  gamma, beta, running_*, save_*, grad_gamma, grad_beta: (C,) f32.
 *   forward : y = act((x - mean) * invstd * gamma + beta [+ residual]); biased variance for the normalisation, running
 *             statistics updated in place like torch (momentum, unbiased variance); running_* may both be NULL.
 *   backward: g = grad_out * [y > 0] (relu) ; grad_beta = sum g ; grad_gamma = sum g * xhat ;
 *             grad_x = gamma * invstd * (g - grad_beta / n - xhat * grad_gamma / n) ; grad_residual (nullable) = g.
 * Two passes per direction, f64 partial sums added in a fixed order: deterministic, no atomics, no host sync.
 * workspace: fnp_bn_workspace_bytes(C).
 * ------------------------------------------------------------------------------------------ */
int64_t fnp_bn_workspace_bytes(int C);
int fnp_bn_train_forward(const void *x, int dtype, const int *n_rows, int cap, int C, const float *gamma, const float *beta,
                         float *running_mean, float *running_var, float momentum, float eps, const void *residual, int relu,
                         void *y, float *save_mean, float *save_invstd, long long *num_batches_tracked, void *workspace,
                         int64_t workspace_bytes, fnp_stream_t stream);
/* num_batches_tracked (ABI 9, nullable): nn.BatchNorm1d's int64 counter, advanced by one on the device with the statistics. */
int fnp_bn_train_backward(const void *grad_out, const void *x, const void *y, int dtype, const int *n_rows, int cap, int C,
                          const float *gamma, const float *save_mean, const float *save_invstd, int relu, void *grad_x,
                          void *grad_residual, float *grad_gamma, float *grad_beta, void *workspace, int64_t workspace_bytes,
                          fnp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * CLIP-crop scoring, geometry half (CLIPBoxClassification.forward,
 * pcdet/models/dense_heads/clip_box_classification.py:230-379): which camera sees which box and the
 * square image crop CLIP looks at.
 *   fnp_clipcrop_plan: boxes (N,7) device; lidar_aug_rot_inv (9) and lidar_aug_trans (3) HOST floats
 *       (torch.inverse of the augmentation rotation is the caller's, like the reference :205-207);
 *       lidar2image, img_aug (6,4,4) device -> rect (N,6,4) f32 [x1, y1, side, has_crop] and
 *       cam_mask (N,6) u8 (box visible in the camera, set even when the crop is < min_crop px).
 *   fnp_clipcrop_sample: images (6, C, H, W) f32/f16 device, pairs (M,2) i32 [box, cam] in the caller's
 *       order, unit_grid (out_size) f32 = the [0,1] sampling positions (F.affine_grid normalised as in
 *       :318-319) -> crops (M, C, out_size, out_size), F.grid_sample arithmetic (bilinear,
 *       align_corners = False, zeros padding).
 * ------------------------------------------------------------------------------------------ */
int fnp_clipcrop_plan(const float *boxes, int n, const float *lidar_aug_rot_inv, const float *lidar_aug_trans,
                      const float *lidar2image, const float *img_aug, int image_h, int image_w, int min_crop,
                      float *rect, unsigned char *cam_mask, fnp_stream_t stream);
int fnp_clipcrop_sample(const void *images, int dtype, int channels, int image_h, int image_w,
                        const float *rect, const int *pairs, int num_pairs, const float *unit_grid,
                        int out_size, void *crops, fnp_stream_t stream);

/* SparseConvTensor.dense() as used by HeightCompression (height_compression.py:20-24):
 * feats (n,C) -> out (B,C,D,H,W) of the same dtype (viewed as (B, C*D, H, W) by the caller).
 * With a workspace of fnp_sparse_to_dense_workspace_bytes() (a cell -> row map) every element of
 * `out` is written exactly once, zeros included, in 128-byte segments: `out` need not be zeroed.
 * With workspace == NULL a row-driven scatter runs and `out` must be zero on entry. */
int64_t fnp_sparse_to_dense_workspace_bytes(int B, int D, int H, int W);
int fnp_sparse_to_dense(const void *feats, int dtype, const int *coords, const int *n_rows, int cap,
                        int C, int B, int D, int H, int W, void *out,
                        void *workspace, int64_t workspace_bytes, fnp_stream_t stream);
/* The same with a per-channel value (fill: C floats, device) for the cells WITHOUT a row instead of 0: the dense map behind
 * a convolution whose epilogue gives unreached cells a constant — relu(BatchNorm shift) — i.e. the output of the first
 * BaseBEVBackbone block (base_bev_backbone.py:31-40) evaluated on the sparse rows.  Needs the workspace. */
int fnp_sparse_to_dense_fill(const void *feats, int dtype, const int *coords, const int *n_rows, int cap, int C,
                             int B, int D, int H, int W, void *out, const float *fill, void *workspace,
                             int64_t workspace_bytes, fnp_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Greedy Box Seeker — replaces hot loops 2-4 of FrustumProposerOG.get_proposals
 * (pcdet/models/dense_heads/frustum_proposals_v1.py:593-1048) for topk = 1: per 2D detection
 * ("frustum") select the points inside the 2D box, take the depth quantiles, build the frustum,
 * generate num_mags x num_rotations x num_sizes candidate boxes, score them by 2D IoU of their
 * projected corners and by point density, return the best one.  One launch for all frustums of
 * a batch of scenes; no host synchronisation.
 * ------------------------------------------------------------------------------------------ */
typedef struct fnp_seeker_params {
    float lq, uq, cq;            /* depth quantiles: frustum near, far, centre (PARAMS lq/uq/cq) */
    float iou_w, dst_w, dns_w;   /* score weights (:997) */
    float min_cam_iou;           /* :904 */
    float max_dist;              /* :871, also caps the frustum depth (:647) */
    int num_mags, num_rotations, num_sizes;
    int topk;                    /* boxes per frustum: the first topk of the 3D NMS in score order (:1030-1045); >= 1 */
    int clamp_bottom;            /* :817 */
    int image_h, image_w;        /* 900, 1600 (:205) */
    int point_stride;            /* floats per point row */
    int xyz_offset;              /* column of x in a point row */
    int has_img_aug;             /* cam_mats carries a non-identity img_aug_matrix (:1456-1458, :1525-1527) */
    int mult;                    /* MODEL_CFG.MULT: product instead of sum of the score terms (:997-1000) */
    float ego_w;                 /* PARAMS ego_w: + ego_w * ||centre|| / max ||centre|| (:1017-1021) */
    /* ABI 8 — the options no shipped configuration sets (:154-196) */
    float nms_normal;            /* threshold of that NMS on the axis-aligned BEV footprints (nms_normal_gpu, :1030) */
    float search_depth;          /* > 0: frustum far plane = near quantile + search_depth, search axis of that length (:617-622,:841) */
    float occl_w;                /* + occl_w * (1 - occl / max occl), occl = calc_occl_scores as it runs (:1007-1014, :408-477) */
    int occl_mult;               /* MODEL_CFG.OCCL_MULT: score = density * IoU * occl (:1022-1027) */
    int multicam;                /* MODEL_CFG.MULTICAM_IOU: IoU averaged over the cameras of the scene's same-label frustums (:885, :1413) */
    int count_only;              /* pre-pass of multicam: only dbg_npts (points per frustum) is written */
    int num_frustums;            /* rows of npts_all (= num_frustums of the call) */
    const int *npts_all;         /* multicam: points per frustum from the count_only pre-pass (device) */
    const float *rand_noise;     /* rand_center (:847): (num_frustums, num_mags, 3) draws added to the weighted centre, or NULL */
} fnp_seeker_params;

int64_t fnp_boxseeker_workspace_bytes(int num_frustums, int max_points_per_scene);

/* scene_mats (S,21) / cam_mats (S,6,45) of fnp_boxseeker made ON THE DEVICE (ABI 8) from the batch's own device tensors —
 * lidar_aug (S,4,4), lidar2image / camera2lidar / intrinsics (S,6,4,4), img_aug (S,6,4,4) or NULL (identity) — the matrix
 * algebra of project_to_camera / get_geometry_at_image_coords (frustum_proposals_v1.py:1431-1475, :1509-1545; the reference
 * calls torch.inverse on 3x3 blocks: here a Gauss-Jordan elimination with partial pivoting in f32).  No host copy, no sync. */
int fnp_seeker_prepare_matrices(const float *lidar_aug, const float *lidar2image, const float *camera2lidar,
                                const float *intrinsics, const float *img_aug, int num_scenes, float *scene_mats,
                                float *cam_mats, fnp_stream_t stream);

/* HOST function (all pointers are host memory, no GPU work): frustum enumeration of :561-594 —
 * per scene, per camera in image_order, torchvision.batched_nms (coordinate trick, f32) on the
 * 2D detections, then the score threshold.  boxes (D,4) xyxy f32, labels/batch_idx/cam_idx (D,)
 * int64, scores (D,) f32 (the CPU tensors PreprocessedGLIP returns, preprocessed_detector.py:47-101).
 * rows (max_rows,8) f32 receives [scene, cam, x1, y1, x2, y2, label, score]; returns the number
 * of rows or a negative error code. */
int fnp_host_enumerate_frustums(const float *boxes, const int64_t *labels, const float *scores,
                                const int64_t *batch_idx, const int64_t *cam_idx, int num_dets,
                                int num_scenes, const int *image_order, int num_cams,
                                float nms_thr, float score_thr, float *rows, int max_rows);

/* points: rows of `point_stride` floats, scenes concatenated; scene_offsets (S+1,) int32.
 * scene_mats (S,21) f32: lidar_aug rotation (9, row major) | its inverse (9) | translation (3).
 * cam_mats (S,6,45) f32: lidar2image[:3,:3] (9) | lidar2image[:3,3] (3) |
 *                        camera2lidar_rot @ inv(intrinsics) (9) | camera2lidar_trans (3) |
 *                        img_aug[:3,:3] (9) | img_aug[:3,3] (3) | inv(img_aug[:3,:3]) (9)   (the last 21 are read only
 *                        when params->has_img_aug)
 *                        (the matrices of :1431-1475 and :1509-1545, prepared on the host).
 * frustums (F,8) f32: scene, camera, x1, y1, x2, y2, label (1-based), score — already NMS'ed,
 *                        score-filtered and in the reference's enumeration order (:582-594).
 * base_boxes (10,R,7), base_corners (10,R,8,3), R = num_rotations*num_sizes (:284-298); mags (num_mags,).
 * Outputs per frustum: out_valid (number of boxes, 0 .. topk), out_box (topk, 7), out_score (topk: second-stage
 * scores), out_best (topk: candidate indices, -1 past out_valid).  dbg_* are optional (NULL) per-candidate dumps:
 * dbg_npts (F), dbg_frust (F,8,3), dbg_cand (F,NC,7), dbg_iou (F,NC), dbg_count (F,NC),
 * dbg_valid (F,NC: 0 dropped by max_dist, 1 dropped by min_cam_iou, 2 scored). */
int fnp_boxseeker(const float *points, const int *scene_offsets, int num_scenes, int max_points_per_scene,
                  const fnp_seeker_params *params, const float *scene_mats, const float *cam_mats,
                  const float *frustums, int num_frustums,
                  const float *base_boxes, const float *base_corners, const float *mags,
                  void *workspace, int64_t workspace_bytes,
                  int *out_valid, float *out_box, float *out_score, int *out_best,
                  int *dbg_npts, float *dbg_frust, float *dbg_cand, float *dbg_iou, int *dbg_count, int *dbg_valid,
                  fnp_stream_t stream);

/* Exchange records of the sharded extraction (BASELINE.json configs[3]; the reference gathers pickled objects or files,
 * pcdet/utils/commu_utils.py:50-111, common_utils.py:229-250): packs the fnp_boxseeker outputs of a batch of scenes
 * into records (num_scenes, rows_per_scene, 9) f32 — row 0 = [count, tags[scene], 0 ...], rows 1.. =
 * [box (7), 2D detection score, label] of the scene's frustums with out_valid != 0, in frustum order; all other rows
 * zero.  frustums / out_valid / out_box as in fnp_boxseeker (device); tags: HOST floats (num_scenes <= 64). */
int fnp_seeker_pack_records(const float *frustums, const int *out_valid, const float *out_box, int num_frustums,
                            const float *tags, int num_scenes, int rows_per_scene, float *records, fnp_stream_t stream);

/* Capacity overflow: data-dependent counts (n_voxels, n_out) always hold the TRUE count; every
 * kernel clamps to the capacity it was given, so a count larger than its capacity means rows
 * were dropped and the caller must re-run with larger buffers. */

#ifdef __cplusplus
}
#endif
#endif /* FNP_H */
