/*
 * spx.h — C ABI of libspx.so: the MI355X (gfx950) sparse-3D-convolution kernel library.
 *
 * This is the drop-in boundary "B3" of SURVEY.md §8(b).  The reference repository
 * (blindopen/TSM-Det-Pointcloud-, an OpenPCDet 0.5.2 fork) reaches all of this arithmetic through
 * the third-party `spconv.pytorch` / `spconv.utils` / `cumm.tensorview` Python packages
 * (pinned spconv_cu118==2.3.8, cumm_cu118==0.7.11, reference `requirements.txt:1,21`), which are
 * not vendored; every entry point below therefore cites the reference CALL SITE it serves.
 *
 * Conventions (all entry points):
 *   - plain C, `extern "C"`, no torch / C++ types in any signature;
 *   - every pointer named d_* or documented "device" is a device (HBM) pointer owned by the caller;
 *     the library never allocates, frees, or synchronises: all scratch comes in through (ws, ws_bytes),
 *     sized by the matching *_ws_bytes() query;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every kernel is enqueued on
 *     it and the call returns immediately (graph-capture safe);
 *   - data-dependent counts (number of voxels, number of active outputs) are RETURNED THROUGH DEVICE
 *     POINTERS and may be CONSUMED through device pointers (`d_n*` arguments, nullable): when a `d_n`
 *     argument is non-NULL the kernels read the live row count from it (it must be <= the host-side
 *     capacity `n` that sizes the launch); when NULL the host value `n` is exact.  This lets a caller
 *     chain voxelise -> rulebooks -> convolutions with no host synchronisation in between;
 *   - return value: 0 (SPX_OK) or a negative SPX_ERR_* code; never throws, never exits;
 *   - re-entrant; no global mutable state; one HIP context per process;
 *   - row indices are int32, linear voxel keys are 64-bit, features are fp32 row-major [rows, channels];
 *   - voxel indices are int32 [rows,4] = (batch, z, y, x), spatial shapes are (D,H,W) = (z,y,x) extents.
 */
#ifndef SPX_H_
#define SPX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPX_ABI_VERSION 2

#define SPX_OK 0
#define SPX_ERR_INVALID_ARG (-1)  /* null pointer, non-positive extent, kernel volume > SPX_MAX_KVOL ... */
#define SPX_ERR_WORKSPACE (-2)    /* ws == NULL or ws_bytes smaller than the *_ws_bytes() answer          */
#define SPX_ERR_UNSUPPORTED (-3)  /* channel count / mode this build has no kernel for                     */
#define SPX_ERR_LAUNCH (-4)       /* hipGetLastError() != hipSuccess after a launch                        */
#define SPX_ERR_TOO_LARGE (-5)    /* rows >= 2^31 or grid cells >= 2^40                                    */
#define SPX_ERR_CAPACITY (-7)     /* DEVICE-side: a strided rule table found more active outputs than the caller's row
                                     capacity (static-capacity mode); rows beyond it were dropped; via d_status          */
#define SPX_ERR_TABLE_FULL (-6)   /* DEVICE-side: a hash probe sequence found no free slot (stale workspace declared
                                     pre-cleared); reported through a d_status word, see spx_read_status()  */
#define SPX_ERR_RING_STALL (-8)   /* DEVICE-side: a wave of spx_conv_gemm_ring gave up a (bounded) wait on the weight ring:
                                     the launch's output is incomplete; via d_status.  Never seen; the exit exists so that a
                                     protocol fault ends as an error code and not as a hung GPU                            */

/* flags of the entry points that keep a hash table in their workspace (spx_voxelize, spx_subm_rulebook) */
#define SPX_WS_PRECLEARED 1 /* the caller has already initialised the workspace (hash keys = 0xFF bytes, values / point
                               slots = 0x7F bytes, e.g. by one bulk fill for several calls): the library skips its own
                               clearing launches.  A workspace that is NOT clean makes the kernels drop the rows they cannot
                               place and raise SPX_ERR_TABLE_FULL in d_status; every probe loop is bounded by the slot count */

#define SPX_ROWS_UNIQUE 2   /* spx_subm_rulebook: the caller guarantees one row per cell (a voxeliser's output; what spconv
                               requires of SparseConvTensor.indices).  The table is then symmetric and only half of it is probed,
                               every hit written twice.  With duplicate rows under this flag the result is undefined (without
                               it duplicates resolve to the smallest row)                                                  */

#define SPX_MAX_KVOL 32 /* largest kernel volume kz*ky*kx supported (27 = 3x3x3 is the reference's max) */

typedef void *spx_stream_t;

/* Human-readable text for an SPX_ERR_* code (static storage). */
const char *spx_strerror(int code);
/* Returns SPX_ABI_VERSION of the loaded library. */
int spx_abi_version(void);

/* Device status word.  Errors that only a kernel can detect (SPX_ERR_TABLE_FULL) are written with atomicMin into a
 * caller-owned device int32 (`d_status`, nullable, initialised to 0 by the caller, sticky across calls).
 * spx_read_status copies it to the host — the ONE entry point that synchronises `stream` — and returns it (0 or a negative
 * SPX_ERR_* code).  A caller that reads a row count back anyway can fetch the word with the same copy instead. */
int spx_read_status(const int32_t *d_status, spx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 1. Hard voxelisation (+ fused MeanVFE)
 *    replaces: spconv.utils.Point2VoxelCPU3d(...).point_to_voxel(tv.from_numpy(points))
 *      reference call site pcdet/datasets/processor/data_processor.py:37-43,55 (VoxelGeneratorWrapper),
 *      driven by DataProcessor.transform_points_to_voxels, data_processor.py:127-155, and the batch
 *      concatenation of DatasetTemplate.collate_batch, pcdet/datasets/dataset.py:161-229;
 *    and (mean != NULL) MeanVFE.forward, pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31.
 *
 *    Semantics: for each point in input order, c_j = floor((p_j-lo_j)/vsize_j)
 *    in fp32 with a true division; the point is dropped if any c_j is outside [0, grid_j).  Voxels are
 *    numbered in first-occurrence order per frame; a new voxel is created only while the frame has
 *    fewer than max_voxels; a point is appended to its voxel only while the voxel holds fewer than
 *    max_points points.  Frames are processed independently and emitted batch-major.
 *
 *    points     device [n_points, point_stride] fp32.  xyz = columns xyz_col..xyz_col+2; the `c`
 *               features copied to the voxel are columns feat_col..feat_col+c-1.
 *    batch_col  column holding the frame index as a float (collate_batch's leading column), or -1
 *               for a single frame.  Points of one frame must be contiguous, frames ascending.
 *    range      host float[6] = (x0,y0,z0,x1,y1,z1); vsize host float[3] = (vx,vy,vz);
 *    grid       host int32[3] = (gx,gy,gz) = round((hi-lo)/vsize), data_processor.py:129-130.
 *    voxels     device [cap, max_points, c] fp32, zero padded (may be NULL when only `mean` is wanted)
 *    coords     device [cap, 4] int32 (b,z,y,x)
 *    num_points device [cap] int32
 *    mean       device [cap, c] fp32 = sum over kept points / max(num,1)  (NULL to skip)
 *    d_num_voxels device int64[1]: total voxels M written (rows [0,M) of every output are valid)
 *    cap        rows available in the outputs; must be >= min(n_points, batch*max_voxels)
 *    flags      0 or SPX_WS_PRECLEARED;  d_status: device status word (nullable), see spx_read_status()
 * ---------------------------------------------------------------------------------------------- */
size_t spx_voxelize_ws_bytes(int64_t n_points, int batch, int max_points);
int spx_voxelize(const float *points, int64_t n_points, int point_stride, int xyz_col, int feat_col, int c,
                 int batch_col, int batch, const float *range, const float *vsize, const int32_t *grid,
                 int max_points, int max_voxels, float *voxels, int32_t *coords, int32_t *num_points,
                 float *mean, int64_t *d_num_voxels, int64_t cap, int flags, int32_t *d_status, void *ws,
                 size_t ws_bytes, spx_stream_t stream);

/* Stand-alone MeanVFE for voxels produced elsewhere (e.g. by CPU dataloader workers):
 * out[v,:] = sum_t voxels[v,t,:] / max(num[v],1); replaces mean_vfe.py:26-29. */
int spx_mean_vfe(const float *voxels, const int32_t *num_points, int64_t n, const int64_t *d_n, int max_points,
                 int c, float *out, spx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 1b. Dynamic voxelisation + mean (SURVEY.md §8a row a5')
 *    replaces: DynamicMeanVFE.forward, pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:38-76
 *      (torch.floor((xyz - range_min) / voxel_size).int(), in-range mask, merge key, torch.unique, scatter_mean).
 *    Every in-range point is kept (no per-voxel / per-frame caps).  Voxels = unique cells in ASCENDING key
 *      ((b*X + cx)*Y + cy)*Z + cz  (the order torch.unique yields in the reference).
 *      points          device [n_points, stride] f32; column batch_col = frame index (or -1: single frame), columns
 *                      [xyz_col, xyz_col + num_features) = x, y, z, extra features
 *      voxel_features  device [cap, num_features] f32 : mean of those columns over the voxel's points, summed in point
 *                      order (the reference sums with float atomics; this is bitwise reproducible)
 *      voxel_coords    device [cap, 4] int32 (b, z, y, x)  -- the column order the reference returns (:72)
 *      point_to_voxel  device [n_points] int32 : voxel row of every point (torch.unique's inverse), -1 if dropped
 *      d_num_voxels    device int64 : number of voxels (may exceed cap: rows beyond cap are dropped)
 *      grid3 = (X, Y, Z) cells; range6 = (xmin, ymin, zmin, xmax, ymax, zmax); voxel_size3 = (vx, vy, vz)
 * ---------------------------------------------------------------------------------------------- */
size_t spx_dynamic_voxelize_ws_bytes(int64_t n_points, int batch, const int32_t *grid3, int64_t cap);
int spx_dynamic_voxelize(const float *points, int64_t n_points, int stride, int batch_col, int xyz_col, int num_features,
                         const float *range6, const float *voxel_size3, const int32_t *grid3, int batch,
                         float *voxel_features, int32_t *voxel_coords, int32_t *point_to_voxel, int64_t *d_num_voxels,
                         int64_t cap, void *ws, size_t ws_bytes, spx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 2. Submanifold rulebook (hash insert + kernel-offset probe)
 *    replaces: the indice-pair build inside spconv.pytorch.SubMConv3d.forward, reference call sites
 *      pcdet/models/backbones_3d/spconv_backbone.py:86,93,99-100,106-107,113-114 (indice_key subm1..4).
 *    pair[k*pair_ld + o] = row of the active voxel at coord(o) + (k - ksize/2)*dil in the same batch
 *    element, or -1;  k = (kz*KH + ky)*KW + kx.   Output rows == input rows (same order).
 *    The backward (dgrad) table of a submanifold conv is the same table read at K-1-k.
 *    cnt    device int32[K]: number of valid pairs per offset (may be NULL).
 *    flags  0 or SPX_WS_PRECLEARED;  d_status: device status word (nullable), see spx_read_status().
 * ---------------------------------------------------------------------------------------------- */
size_t spx_subm_rulebook_ws_bytes(int64_t n);
int spx_subm_rulebook(const int32_t *idx, int64_t n, const int64_t *d_n, int batch, const int32_t *shape,
                      const int32_t *ksize, const int32_t *dil, int32_t *pair, int64_t pair_ld, int32_t *cnt,
                      int flags, int32_t *d_status, void *ws, size_t ws_bytes, spx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 3. Regular (strided) sparse-convolution rulebook
 *    replaces: the indice-pair build inside spconv.pytorch.SparseConv3d.forward, reference call sites
 *      spconv_backbone.py:98,105,112 (k3 s2, keys spconv2..4) and :121-122 (k(3,1,1) s(2,1,1), spconv_down2).
 *    Candidate outputs o = (i + pad - k*dil)/stride where divisible and 0 <= o < out_shape;
 *    out_idx = unique candidates in ASCENDING linear key ((b*D+z)*H+y)*W+x (canonical order).
 *      out_shape  host int32[3] = floor((in + 2*pad - dil*(k-1) - 1)/stride) + 1  (caller computes; checked)
 *      out_idx    device [cap,4] int32
 *      pair_fwd   device [K, cap]  int32 : pair_fwd[k*cap + o] = input row feeding output o at offset k, or -1
 *      pair_bwd   device [K, n_in] int32 : pair_bwd[k*n_in + i] = output row fed by input i at offset k, or -1
 *      cnt        device int32[K] (nullable);  d_n_out device int64[1] = number of active outputs
 *      cap        output-row capacity.  spx_conv_out_cap() = min(prod(ceil(k/s)) * n_in, batch*out cells) can never
 *                 overflow; a smaller static capacity is allowed (graph mode): rows beyond cap are dropped and
 *                 *d_n_out still reports the true count, so *d_n_out > cap signals overflow, and the status word
 *                 d_status (nullable, see spx_read_status) receives SPX_ERR_CAPACITY
 *      subm_pair  (nullable) device [Ks, cap] int32: the SUBMANIFOLD table of the output level (section 2 semantics over
 *                 out_idx, kernel subm_ksize / subm_dil, leading dimension cap) built from the same rank bitmap in the
 *                 same call — the reference's stages are a strided conv followed by submanifold convs on its output
 *                 (spconv_backbone.py:98-100,105-107,112-114), and the output rows are in rank order, so a neighbour
 *                 lookup is one bitmap word + a popcount: no hash is built for that level.  subm_cnt: int32[Ks], nullable
 * ---------------------------------------------------------------------------------------------- */
int64_t spx_conv_out_cap(int64_t n_in, int batch, const int32_t *out_shape, const int32_t *ksize,
                         const int32_t *stride);
size_t spx_conv_rulebook_ws_bytes(int64_t n_in, int batch, const int32_t *out_shape);
int spx_conv_rulebook(const int32_t *idx, int64_t n_in, const int64_t *d_n_in, int batch, const int32_t *in_shape,
                      const int32_t *out_shape, const int32_t *ksize, const int32_t *stride, const int32_t *pad,
                      const int32_t *dil, int32_t *out_idx, int32_t *pair_fwd, int32_t *pair_bwd, int32_t *cnt,
                      int64_t *d_n_out, int64_t cap, const int32_t *subm_ksize, const int32_t *subm_dil,
                      int32_t *subm_pair, int32_t *subm_cnt, int32_t *d_status, void *ws, size_t ws_bytes,
                      spx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 4. Sparse convolution arithmetic
 *    replaces: spconv.pytorch.{SubMConv3d,SparseConv3d}.forward and their autograd backward, the 12
 *      call sites spconv_backbone.py:86,93,98-100,105-107,112-114,121; backward is triggered by
 *      loss.backward() at tools/train_utils/train_utils.py:53.
 *
 *    Weights arrive in the reference parameter layout  w[Cout][K][Cin]  (= spconv 2.x
 *    weight[Cout,kz,ky,kx,Cin], detector3d_template.py:547-562) and are re-laid for the MFMA operand
 *    order by spx_pack_weight (mode 0: forward operand W_k[ci][co]; mode 1: dgrad operand W_k^T; mode 2: both in one
 *    launch, forward operand first).  packed size = K*Cin*Cout floats (twice that for mode 2).
 *
 *    forward : out[o,co] = sum_k sum_ci in[pair[k*ld+o], ci] * w[co][k][ci]          (pair = forward table)
 *    dgrad   : din[i,ci] = sum_k sum_co dout[pairT[k*ld+i], co] * w[co][k][ci]       (pairT = backward table;
 *              for a submanifold conv pass the forward table and flip_k = 1)
 *      -> both are spx_conv_gemm: "rows of `src` gathered through `pair`, contracted with packed weights".
 *    Optional fused epilogue:  y = acc*scale[c] + shift[c] (both nullable), then ReLU if relu != 0.
 *    wgrad   : dw[co][k][ci] = sum_o dout[o,co] * in[pair[k*ld+o], ci]                (reference layout, fp32)
 *              (fixed summation order: bitwise reproducible from run to run)
 * ---------------------------------------------------------------------------------------------- */
int spx_pack_weight(const float *w, int cout, int kvol, int cin, int mode, float *packed, spx_stream_t stream);

/* All weights of a network in one launch: d_desc = device int64[n][6] = {w (device pointer, contiguous [Cout][K][Cin]), packed
 * (device pointer, 2*K*Cin*Cout floats: forward operand then dgrad operand, as mode 2), Cout, K, Cin, first block}; weight i
 * owns blocks [first block_i, first block_{i+1}) with ceil(2*K*Cin*Cout / 256) blocks each; total_blocks = their sum. */
int spx_pack_weight_batched(const int64_t *d_desc, int n, int64_t total_blocks, spx_stream_t stream);

int spx_conv_gemm(const float *src, int c_src, const float *w_packed, int c_dst, int kvol, int flip_k,
                  const int32_t *pair, int64_t pair_ld, int64_t n_dst, const int64_t *d_n_dst,
                  const float *scale, const float *shift, int relu, float *dst, spx_stream_t stream);

/* MFMA-work-balanced schedule of the same product (csrc/conv_balanced.hip): spx_conv_plan counts the non-empty
 * (16-row tile, offset) units of a rule table once and cuts them into equal ranges for a persistent grid; the plan
 * depends only on (pair, n_dst), so it is reused by every convolution that reads the table (forward, dgrad with
 * flip_k, the second layer of a submanifold pair).  spx_conv_gemm_balanced = spx_conv_gemm under that schedule;
 * returns SPX_ERR_UNSUPPORTED for channel pairs it does not cover (use spx_conv_gemm).  Same reference call sites.
 * Super-tiles whose offsets are split between workgroups are combined inside the launch (arrival counters kept in `plan`,
 * which is therefore not const: one launch per plan at a time — stream order is enough); partial sums are added in a fixed
 * order, so results are bitwise reproducible. */
/* Optional row order for that schedule (csrc/conv_group.hip): a 16-row MFMA tile multiplies offset k for all its rows
 * as soon as one of them has it, so tiles whose rows share the same offsets issue fewer wasted MFMAs.  spx_conv_group orders
 * the destination rows of a rule table by (window, group key) — windows = contiguous ranges of at most 4096 live rows, at
 * least eight of them (one XCD's L2 then serves one part of the feature matrix); group key = an 11-bit digest of the row's
 * offset mask (3x3x3: the nine in-plane offsets bit by bit, any offset in the plane below, any in the plane above) — stable
 * (equal keys keep the table order; rows beyond the live count keep their place), with a hand-written counting sort, and
 * writes perm[n_dst] (position -> table row) and pair_grouped[kvol][n_dst] = pair[k][perm[j]] (-1 beyond the live rows).
 * Build the plan over pair_grouped (ld = n_dst) and pass pair_grouped + perm to spx_conv_gemm_balanced: position j is
 * written to dst row perm[j].  Results do not depend on the row order (every row is the same sum over k).  kvol <= 30. */
size_t spx_conv_group_ws_bytes(int64_t n_dst);
int spx_conv_group(const int32_t *pair, int64_t pair_ld, int kvol, int64_t n_dst, const int64_t *d_n_dst, int32_t *perm,
                   int32_t *pair_grouped, void *ws, size_t ws_bytes, spx_stream_t stream);
/* perm (below): NULL = rows in table order */
size_t spx_conv_plan_bytes(int64_t n_dst);
int spx_conv_plan(const int32_t *pair, int64_t pair_ld, int kvol, int64_t n_dst, const int64_t *d_n_dst, int32_t *plan,
                  spx_stream_t stream);
size_t spx_conv_gemm_balanced_ws_bytes(int c_dst, int64_t n_dst);
int spx_conv_gemm_balanced(const float *src, int c_src, const float *w_packed, int c_dst, int kvol, int flip_k,
                           const int32_t *pair, int64_t pair_ld, int64_t n_dst, const int64_t *d_n_dst,
                           const float *scale, const float *shift, int relu, int32_t *plan, const int32_t *perm,
                           float *dst, void *ws, size_t ws_bytes, spx_stream_t stream);

/* Round-3 schedule of the same product (csrc/conv_ring.hip): every 16-row tile belongs to ONE wave for the whole launch
 * (no tile is split between workgroups: no partial-sum slabs, no tickets), the K weight slices stream through an LDS ring
 * filled by a loader wave, consumers run barrier-free.  spx_conv_ring_plan (cached per rule table, like spx_conv_plan) deals
 * the tiles to the chip's 1024 SIMDs by their number of non-empty offsets; the plan is written by the planning kernels only
 * (one debug counter aside), so any number of launches may share it.  pair / perm: exactly as spx_conv_gemm_balanced
 * (a table grouped by spx_conv_group, or the plain table with perm = NULL).  n_src = rows of `src` (entries are bounds-checked
 * against it by the buffer hardware: an entry of -1 reads zeros).  stats (nullable): [spx_conv_ring_stat_rows()][2][c_dst]
 * floats, row b = column sums of the written values and of their squares over the rows workgroup b wrote — the statistics
 * pass of the training-mode BatchNorm1d that follows (reference spconv_backbone.py:26-27,81), consumed by
 * spx_bn_relu_fwd_from_sums.  Every output row is the same sum over k in ascending order whatever the plan: bitwise
 * reproducible.  Channel pairs: (32|64) x (32|64); others SPX_ERR_UNSUPPORTED.  kvol <= 31.  d_status (nullable, see
 * spx_read_status) receives SPX_ERR_RING_STALL if a wave's bounded wait on the ring gave up. */
/* spx_conv_ring_tiles_per_wave(set): tuning knob of spx_conv_ring_plan, process-wide.  0 (default): by size — a wave holds one
 * 16-row tile per turn of the weight ring while one turn covers all live rows, two tiles (sharing the ring protocol of every
 * offset) beyond that; 1 / 2: always that many (environment SPX_RING_TM presets it; tests and A/B runs).  Returns the
 * setting; any other `set` only queries.  The plan records what it was dealt for and spx_conv_gemm_ring follows the plan. */
size_t spx_conv_ring_plan_bytes(int64_t n_dst);
int spx_conv_ring_stat_rows(void);
int spx_conv_ring_tiles_per_wave(int set);
int spx_conv_ring_plan(const int32_t *pair, int64_t pair_ld, int kvol, int64_t n_dst, const int64_t *d_n_dst, int32_t *plan,
                       spx_stream_t stream);
int spx_conv_gemm_ring(const float *src, int64_t n_src, int c_src, const float *w_packed, int c_dst, int kvol, int flip_k,
                       const int32_t *pair, int64_t pair_ld, int64_t n_dst, const int64_t *d_n_dst, const float *scale,
                       const float *shift, int relu, int32_t *plan, const int32_t *perm, float *dst, float *stats,
                       int32_t *d_status, spx_stream_t stream);

size_t spx_conv_wgrad_ws_bytes(int cin, int cout, int kvol, int64_t n_out);
/* counts (nullable): the table's pair counts from spx_conv_wgrad_counts (device, spx_conv_wgrad_counts_bytes); they depend on
 * the rule table only, so a table that serves several layers / steps of a replayed graph is counted once.  NULL: counted
 * inside the call. */
size_t spx_conv_wgrad_counts_bytes(int kvol, int64_t n_out);
int spx_conv_wgrad_counts(const int32_t *pair, int64_t pair_ld, int kvol, int64_t n_out, const int64_t *d_n_out,
                          int32_t *counts, spx_stream_t stream);
int spx_conv_wgrad(const float *in, int cin, const float *dout, int cout, int kvol, const int32_t *pair,
                   int64_t pair_ld, int64_t n_out, const int64_t *d_n_out, const int32_t *counts, float *dw, void *ws,
                   size_t ws_bytes, spx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 5. Densify (BEV collapse feed)
 *    replaces: spconv.pytorch.SparseConvTensor.dense(), reference call site
 *      pcdet/models/backbones_2d/map_to_bev/height_compression.py:21 (followed by the view at :22-23).
 *    layout 0: dense[b][c][z][y][x]  (contiguous NCDHW, what .dense() returns)
 *    layout 1: dense[b][y][x][c][z]  (the same logical [B,C,D,H,W] tensor stored so that
 *              view(B, C*D, H, W) is channels_last: BEV channel c*D+z is the fastest axis)
 *    The caller zero-fills `dense` (hipMemsetAsync) before spx_densify; spx_densify_bwd gathers
 *    dfeat[row,c] = ddense[...] (autograd of .dense()).
 * ---------------------------------------------------------------------------------------------- */
int spx_densify(const float *feat, const int32_t *idx, int64_t n, const int64_t *d_n, int c, int batch,
                const int32_t *shape, int layout, float *dense, spx_stream_t stream);
int spx_densify_bwd(const float *ddense, const int32_t *idx, int64_t n, const int64_t *d_n, int c, int batch,
                    const int32_t *shape, int layout, float *dfeat, spx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 6. Rotated-BEV IoU and NMS  (post-processing)
 *    replaces: iou3d_nms_cuda.boxes_overlap_bev_gpu / boxes_iou_bev_gpu / nms_gpu / nms_normal_gpu, reference
 *      pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:236-325, src/iou3d_nms.cpp:53-190, called from
 *      pcdet/ops/iou3d_nms/iou3d_nms_utils.py:48-118 by model_nms_utils.py:6-87.
 *    boxes are device [n,7] fp32 (x,y,z,dx,dy,dz,heading).
 *    spx_boxes_iou_bev : out[n,m] = BEV IoU (overlap_only != 0: intersection AREA) of every pair.
 *    spx_nms_bev       : boxes must already be sorted by descending score; keep[0..*d_num_keep) receives the kept
 *                        positions in ascending order (greedy: a box is kept iff no earlier kept box has IoU > thresh).
 *                        axis_aligned != 0 uses the heading-less IoU of nms_normal_gpu.  The suppression mask and its
 *                        reduction stay on the device (the reference copies the mask to the host and reduces there).
 * ---------------------------------------------------------------------------------------------- */
int spx_boxes_iou_bev(const float *boxes_a, int64_t n, const float *boxes_b, int64_t m, int overlap_only, float *out,
                      spx_stream_t stream);
size_t spx_nms_ws_bytes(int64_t n);
int spx_nms_bev(const float *boxes, int64_t n, float thresh, int axis_aligned, int64_t *keep, int64_t *d_num_keep,
                void *ws, size_t ws_bytes, spx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 7. Anchor target assignment (training)
 *    replaces: AxisAlignedTargetAssigner.assign_targets, reference
 *      pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:36-210 (per-sample / per-class python
 *      loops over torch ops), incl. boxes3d_nearest_bev_iou (pcdet/utils/box_utils.py:249-298) and
 *      ResidualCoder.encode_torch (pcdet/utils/box_coder_utils.py:13-43); SECOND settings only
 *      (POS_FRACTION < 0, NORM_BY_NUM_EXAMPLES False, MATCH_HEIGHT False, single head).
 *    anchors   device [n_sets][anchors_per_set][7], every set laid out (z=1, y, x, size, rot) as AnchorGenerator emits
 *    gt_boxes  device [batch][max_gt][8] = (x,y,z,dx,dy,dz,heading,class 1..n_classes), zero padded
 *    d_set_class device int32[n_sets]: 0-based class index each anchor set is matched against
 *    outputs in the head's anchor order (y, x, set, within-location), A_total = n_sets * anchors_per_set:
 *      labels int32 [batch][A_total] (class id / 0 background / -1 ignored), targets fp32 [batch][A_total][7],
 *      weights fp32 [batch][A_total] (1 where labels > 0)
 * ---------------------------------------------------------------------------------------------- */
size_t spx_assign_targets_ws_bytes(int batch, int n_sets, int max_gt);
int spx_assign_targets(const float *anchors, int n_sets, int64_t anchors_per_set, int per_location,
                       const float *gt_boxes, int batch, int max_gt, const int32_t *d_set_class, int n_classes,
                       const float *d_matched, const float *d_unmatched, int32_t *labels, float *targets,
                       float *weights, void *ws, size_t ws_bytes, spx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 8. Anchor-head losses, forward + gradient in one pass (training)
 *    replaces: AnchorHeadTemplate.get_cls_layer_loss / get_box_reg_layer_loss and their autograd, reference
 *      pcdet/models/dense_heads/anchor_head_template.py:101-224 with the loss classes of
 *      pcdet/utils/loss_utils.py:9-77 (sigmoid focal, alpha/gamma=2), :140-209 (smooth L1, beta, unit code weights,
 *      after add_sin_difference) and :310-338 (weighted cross entropy on direction bins).
 *    cls_preds [batch][A][num_class], box_preds [batch][A][7], dir_preds [batch][A][num_dir_bins] (NULL: no direction
 *    classifier), labels int32 [batch][A], reg_targets [batch][A][7], anchors [A][7] — all device, anchor order of the head.
 *    losses  device fp32[3] = (cls, loc, dir), each already divided by batch and multiplied by its weight
 *    dcls/dbox/ddir  device, same shapes as the predictions: d(weighted loss)/d(prediction)
 * ---------------------------------------------------------------------------------------------- */
size_t spx_anchor_loss_ws_bytes(int batch, int64_t n_anchors);
int spx_anchor_loss(const float *cls_preds, const float *box_preds, const float *dir_preds, const int32_t *labels,
                    const float *reg_targets, const float *anchors, int batch, int64_t n_anchors, int num_class,
                    int num_dir_bins, float dir_offset, float cls_weight, float loc_weight, float dir_weight,
                    float beta, float alpha, float *losses, float *dcls, float *dbox, float *ddir, void *ws,
                    size_t ws_bytes, spx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 9. BatchNorm1d (+ReLU) over sparse feature rows, training mode
 *    replaces: nn.BatchNorm1d(C, eps=1e-3, momentum=0.01) + nn.ReLU applied to SparseConvTensor.features by
 *      SparseSequential, reference pcdet/models/backbones_3d/spconv_backbone.py:81,24-33 (three torch launches forward,
 *      three backward per layer).  C must divide 1024 and be a multiple of 4.
 *    fwd: batch mean / biased variance over the n rows -> save_mean, save_invstd; running_mean/var (nullable) updated
 *         with `momentum` (unbiased variance), y = relu?((x-mean)*invstd*gamma + beta)
 *    bwd: dx, dgamma, dbeta from dy (ReLU mask taken from y when relu != 0)
 * ---------------------------------------------------------------------------------------------- */
size_t spx_bn_relu_ws_bytes(int c);
int spx_bn_relu_fwd(const float *x, int64_t n, const int64_t *d_n, int c, const float *gamma, const float *beta,
                    float *running_mean, float *running_var, float momentum, float eps, int relu, float *y,
                    float *save_mean, float *save_invstd, void *ws, size_t ws_bytes, spx_stream_t stream);
/* backward: the ReLU mask is recomputed from x (same instruction sequence as the forward), y is not needed; d_n (nullable)
 * = device-side live row count as in the forward */
int spx_bn_relu_bwd(const float *x, const float *dy, int64_t n, const int64_t *d_n, int c, const float *gamma, const float *beta,
                    const float *save_mean, const float *save_invstd, int relu, float *dx, float *dgamma, float *dbeta,
                    void *ws, size_t ws_bytes, spx_stream_t stream);

/* y = relu(bn(x) + res): the tail of SparseBasicBlock (reference spconv_backbone.py:56-72: bn2, `out.features +
 * identity.features`, ReLU; SURVEY.md §8 row f-3 "fused residual add epilogue").  res [n, c] or NULL (= spx_bn_relu_*);
 * backward also writes dres [n, c] (may be NULL) = dy masked by the ReLU, the gradient of the identity branch.
 * num_batches_tracked: nn.BatchNorm's int64 counter, incremented by one inside the kernels (NULL: not touched).
 * y_ld / dy_ld: row stride in floats of y / dy (0 = c): a layer can write its output straight into a channel slice of a
 * wider [n, y_ld] matrix (the channel concatenation of the BEV up-sampling branches, reference
 * base_bev_backbone.py:99-106) and read its gradient from the same slice; multiples of 4, 16-byte aligned base. */
int spx_bn_add_relu_fwd(const float *x, const float *res, int64_t n, const int64_t *d_n, int c, const float *gamma,
                        const float *beta, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                        float momentum, float eps, int relu, float *y, int64_t y_ld, float *save_mean, float *save_invstd,
                        void *ws, size_t ws_bytes, spx_stream_t stream);
int spx_bn_add_relu_bwd(const float *x, const float *res, const float *dy, int64_t dy_ld, int64_t n, const int64_t *d_n,
                        int c, const float *gamma, const float *beta, const float *save_mean, const float *save_invstd, int relu,
                        float *dx, float *dres, float *dgamma, float *dbeta, void *ws, size_t ws_bytes,
                        spx_stream_t stream);

/* Inference-mode BatchNorm (+ residual) (+ ReLU) with GIVEN statistics, one pass: y = relu?((x - mean) * invstd * gamma + beta
 * (+ res)).  replaces: nn.BatchNorm2d (eval) + nn.ReLU of the BEV backbone, reference
 * pcdet/models/backbones_2d/base_bev_backbone.py:35-44,60-73 (two elementwise passes in torch), applied to the channels_last
 * map as [B*H*W, C] rows; y_ld as in spx_bn_add_relu_fwd (a channel slice of the concatenated map, :99-106). */
int spx_bn_apply(const float *x, const float *res, int64_t n, const int64_t *d_n, int c, const float *mean,
                 const float *invstd, const float *gamma, const float *beta, int relu, float *y, int64_t y_ld,
                 spx_stream_t stream);

/* Training-mode BatchNorm (+ReLU) from per-block sums taken by the producer of x (spx_conv2d_wino's stat_partials):
 * partial[nblk][2][c] = sums of x and x*x; finalize + apply, no statistics pass over x.  replaces: the same nn.BatchNorm2d
 * (train) + nn.ReLU as spx_bn_add_relu_fwd for the 3x3 layers of the BEV backbone, base_bev_backbone.py:38-49, and (with the
 * sums of spx_conv_gemm_ring's epilogue) the nn.BatchNorm1d + nn.ReLU of the 64-channel sparse blocks,
 * spconv_backbone.py:26-27,81.  d_n (nullable): device-side live row count of a static-capacity row matrix. */
int spx_bn_relu_fwd_from_sums(const float *x, int64_t n, const int64_t *d_n, int c, const float *partial, int64_t nblk,
                              const float *gamma,
                              const float *beta, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                              float momentum, float eps, int relu, float *y, int64_t y_ld, float *save_mean,
                              float *save_invstd, spx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 9b. Dense 3x3 / stride 1 / pad 1 convolution of the BEV backbone, Winograd F(2x2, 3x3) on the exact-fp32 MFMA
 *    replaces: nn.Conv2d(c, c, kernel_size=3, padding=1, bias=False) forward and its data gradient, reference
 *    pcdet/models/backbones_2d/base_bev_backbone.py:38-49 (cuDNN / MIOpen implicit GEMM: 9 multiplies per (pixel, ci, co)
 *    against 4 here).  Maps are channels-last: pixel p = (n*H + y)*W + x, channel c at x[p * x_ld + c].
 *
 * spx_wino_weight: weight element (co, ci, a, b) at w[co*s_o + ci*s_i + a*s_a + b*s_b] (strides in floats: OIHW or
 *    channels_last) -> the transformed, fragment-ordered image u (spx_wino_weight_floats(cin, cout) floats).
 *    flip = 0: forward filter, cin/cout = the layer's.  flip = 1: filter of the data gradient — pass cin = the layer's
 *    Cout and cout = the layer's Cin; taps are rotated by 180 degrees and the channel roles swapped inside.
 *    cin % 32 == 0 and cout % 128 == 0, else SPX_ERR_INVALID_ARG.
 * spx_conv2d_wino: y = conv3x3(x) with optional epilogue y = relu?(y * scale[co] + shift[co]) (eval BatchNorm folded to
 *    scale/shift; null = identity).  x_ld >= cin, y_ld >= cout, both multiples of 4; x, y 16-byte aligned.
 *    stat_partials (or null): [spx_wino_stat_rows(n, h, w)][2][cout] floats, row b = the sums of y and of y*y over the pixels
 *    of tile block b — the statistics pass of the training-mode BatchNorm that follows, taken where y is produced
 *    (spx_bn_relu_fwd_from_sums consumes them). */
int64_t spx_wino_stat_rows(int32_t n, int32_t h, int32_t w);
int64_t spx_wino_weight_floats(int32_t cin, int32_t cout);
int spx_wino_weight(const float *w, int64_t s_o, int64_t s_i, int64_t s_a, int64_t s_b, int32_t cin, int32_t cout, int flip,
                    float *u, spx_stream_t stream);
int spx_conv2d_wino(const float *x, int64_t x_ld, const float *u, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout,
                    const float *scale, const float *shift, int relu, float *y, int64_t y_ld, float *stat_partials,
                    spx_stream_t stream);

/* spx_conv2d_wino_wgrad: weight gradient of the same convolution in the Winograd domain (csrc/wino_wgrad.hip):
 *    dw[co*s_o + ci*s_i + a*s_a + b*s_b] = sum over pixels of x[.., ci] (shifted by the tap) * dy[.., co]  — the weight half
 *    of convolution_backward for the layers of 9b.  cin % 128 == 0 and cout % 128 == 0 (else SPX_ERR_INVALID_ARG), map at
 *    least 15 pixels wide (else SPX_ERR_UNSUPPORTED: callers keep the vendor kernel).  ws: spx_wino_wgrad_ws_bytes bytes
 *    (per-split partial sums, summed in a fixed order: deterministic, no atomics). */
size_t spx_wino_wgrad_ws_bytes(int32_t cin, int32_t cout);
int spx_conv2d_wino_wgrad(const float *x, int64_t x_ld, const float *dy, int64_t dy_ld, int32_t n, int32_t h, int32_t w,
                          int32_t cin, int32_t cout, float *dw, int64_t s_o, int64_t s_i, int64_t s_a, int64_t s_b, void *ws,
                          size_t ws_bytes, spx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 10. Voxel query (consumers of multi_scale_3d_features)
 *    replaces: pointnet2_stack_cuda.voxel_query_wrapper, reference
 *      pcdet/ops/pointnet2/pointnet2_stack/src/voxel_query_gpu.cu:10-122 (python side voxel_query_utils.py:12-50).
 *      new_xyz [m,3] f32 query centres; xyz [n,3] f32 positions of the rows; new_coords [m,4] int32 (b, z, y, x);
 *      point_indices [batch, Z, Y, X] int32 = row at each cell or -1 (generate_voxel2pinds); shape3 = (Z, Y, X);
 *      range3 = (z_range, y_range, x_range) cells scanned each way; idx [m, nsample] int32 out (slot 0 = -1 for an
 *      empty ball, exactly as the reference kernel leaves it); cnt_unique [m] = occupied cells scanned.
 *    The reservoir step that applies once more than nsample neighbours lie within the radius uses cuRAND's XORWOW
 *    algorithm seeded with the query index like the reference; that stream is parity-unpinned here (csrc/voxel_query.hip).
 * ---------------------------------------------------------------------------------------------- */
int spx_voxel_query(const float *new_xyz, const float *xyz, const int32_t *new_coords, const int32_t *point_indices,
                    int64_t m, int batch, const int32_t *shape3, int nsample, float radius, const int32_t *range3,
                    int32_t *idx, int32_t *cnt_unique, spx_stream_t stream);

/* replaces: pointnet2_stack_cuda.voxel_query_dilated_wrapper, reference voxel_query_gpu.cu:125-236 (python side
 *   voxel_query_utils.py:117-158).  As spx_voxel_query, with the scan stepping by stride3 = (z, y, x) cells, neighbours
 *   closer than former_radius dropped as well, and idx_cnt [m] = number of slots filled before padding (<= nsample). */
int spx_voxel_query_dilated(const float *new_xyz, const float *xyz, const int32_t *new_coords,
                            const int32_t *point_indices, int64_t m, int batch, const int32_t *shape3, int nsample,
                            float former_radius, float radius, const int32_t *range3, const int32_t *stride3,
                            int32_t *idx, int32_t *cnt_unique, int32_t *idx_cnt, spx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPX_H_ */
