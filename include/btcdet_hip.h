/*
 * btcdet_hip.h -- C ABI of libbtcdet_hip.so, the MI355X (gfx950) implementation of BtcDet's
 * data-parallel hot path: point-cloud voxelizer, sparse-3D-conv rulebook construction, sparse conv /
 * max-pool apply (forward + backward), sparse->dense scatter and the occupancy-target kernels.
 *
 * This is the drop-in boundary.  The reference reaches this code through
 * third-party `spconv` v1.2.1 (pybind11; /root/reference/README.md:42) and through chains of torch
 * ops; each entry point below names the reference interface it replaces.  The Python host side
 * (btcdet_amd/spconv/ and the btcdet_amd modules) binds these symbols with ctypes and mirrors the reference's
 * operator / module API (INTEGRATION.md shows the binding a maintainer would add).
 *
 * Rules common to every entry point
 *   - all pointers are DEVICE pointers unless the name starts with h_ (host); no torch types;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised, unless stated;
 *   - caller allocates every output and workspace buffer (sizes given per function; the *_ws_bytes
 *     helpers are pure host functions);
 *   - return value: 0 on success, a negative BTC_E* code on error (never exit(), cf. the reference's
 *     exit(-1) in btcdet/ops/iou3d_nms/src/iou3d_nms.cpp:14-38); btc_last_error() gives the text;
 *   - indices are int32 rows [b,z,y,x]; features are row-major float32 (N,C);
 *   - kernel offset index k = (kz*KH + ky)*KW + kx; weights are float32 [K][Cin][Cout]
 *     (= spconv's parameter layout [kD,kH,kW,Cin,Cout]);
 *   - neighbour maps: nbr_out (n_out,K) = input row gathered by output row i at offset k, or -1;
 *                     nbr_in  (n_in ,K) = output row that input row j feeds at offset k, or -1.
 *     They are the output-stationary / input-stationary forms of spconv's indice_pairs
 *     (SURVEY.md App. B.4); btc_pairs_from_nbr converts to the (2,K,N) / (K,) layout.
 */
#ifndef BTCDET_HIP_H
#define BTCDET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BTC_OK 0
#define BTC_EINVAL (-1)   /* bad argument */
#define BTC_ELAUNCH (-2)  /* HIP launch / runtime error */
#define BTC_ERANGE (-3)   /* problem too large for the cell index range (see the entry point) */

#define BTC_MODE_SUBM 0
#define BTC_MODE_CONV 1       /* also the geometry of SparseMaxPool3d */
#define BTC_MODE_TRANSPOSE 2

const char* btc_last_error(void);
int btc_version(void);

/* Kernel-selection override for tuning / A-B measurements (tools/conv_bench.py); value 0 restores the built-in policy.
 * Results never depend on keys 0-2 (every variant produces the same bits).  Keys:
 *   BTC_TUNE_APPLY_KERNEL  1 = register-staged conv_apply, 2 = LDS-DMA pipelined conv_apply_g (where supported)
 *   BTC_TUNE_APPLY_NT      16-column tiles per workgroup (1, 2, 4, 8); for kernel 2 the wave shape WR*100 + WC*10 + NTW
 *   BTC_TUNE_APPLY_KC      kernel 2: reduction channels per pipeline item (16, 32, 64)
 *   BTC_TUNE_APPLY_XCD     1 = XCD-contiguous row-tile mapping off, 2 = on */
#define BTC_TUNE_APPLY_KERNEL 0
#define BTC_TUNE_APPLY_NT 1
#define BTC_TUNE_APPLY_XCD 2
#define BTC_TUNE_APPLY_KC 4
#define BTC_TUNE_WGRAD_PH 5   /* conv_wgrad_rows: phases (of KB offsets) per offset group: 1, 2, 4, 7 */
#define BTC_TUNE_WGRAD_WGS 6  /* conv_wgrad_rows: target number of workgroups (row splits x offset groups) */
#define BTC_TUNE_POV_SELECT 7 /* PassOccVox top-k: 1 = single-workgroup pov_select (cross-check of the multi-workgroup path) */
#define BTC_TUNE_BF16_OPERANDS 8 /* host bindings: 1 = keep fp32 weights under bf16 activations (conv_apply_g, bit-exact fmaf chain) instead of btc_conv_*_bf16w */
#define BTC_TUNE_BN_FWD_KB 9  /* bn_stats: KB of input per workgroup (0 = built-in 64) */
#define BTC_TUNE_BN_BWD_KB 10 /* bn_bwd_stats: KB of input (x, y, dy) per workgroup (0 = built-in 128) */
#define BTC_TUNE_WGRAD_PIPE 11 /* conv_wgrad_rows: 1 = the two-barrier kernel instead of the software-pipelined one */
#define BTC_TUNE_BN_FUSE 12 /* btc_conv_bn_relu_fwd: 1 = statistics by the separate bn_stats launch instead of the conv epilogue */
#define BTC_TUNE_SPLIT_Z 15 /* split-operand kernel: workgroups sharing a tile's items (0 = built-in policy, 1 = never, 2..4 = always that many) */
#define BTC_TUNE_SPLIT_LOADERS 17 /* split-operand kernel: 0 = built-in policy, 1 = the product waves issue their own LDS-DMA pieces, 2 / 4 = that many loader waves per workgroup issue them all (same bits in every mode) */
#define BTC_TUNE_SPLIT 14 /* host bindings: 1 = never take the split-operand kernel (conv_apply_g's exact fmaf chain everywhere) */
#define BTC_TUNE_APPLY_STAGES 13 /* conv_apply_g: depth of the LDS ring (3..8; 0 = built-in policy) */
#define BTC_TUNE_WGRAD_X 18 /* weight gradient on the bf16 matrix pipe (conv_wgrad_x.hip): 0 = where supported, 1 = never (the fp32-pipe kernels) */
#define BTC_TUNE_WGRAD_X_DEPTH 20 /* (key 16 is retired: it named the deleted in-kernel z-split reduction) conv_wgrad_x: items of gathered rows in flight ahead of the products: 0 = built-in (2 for bf16 activations, 1 for split fp32), 1, 2 (same bits) */
#define BTC_TUNE_RB_MARK_MULTI 19 /* chain rulebooks: 1 = mark every level by its own launch (rb_mark / rb_mark_b) instead of one launch for the leading run of strided conv layers (cross-check: same levels) */
#define BTC_TUNE_SPLIT_PAIR 21 /* split-operand kernel and bf16-operand kernel, 32-channel reductions: 0 = built-in policy, 1 = one offset per item, 2 = two offsets per 64-channel item wherever a tile shape has the instance (same bits) */
#define BTC_TUNE_WGRAD_NARROW 22 /* weight gradient of a layer with <= 8 result channels walked over its input rows (conv_wgrad_n.hip; needs the backward map or nbr_in == nbr_out): 0 = where supported, 1 = never */
#define BTC_TUNE_APPLY_DEBUG 3 /* timing experiments only (WRONG results): 1 = no MFMA phase, 2 = no loads in the main loop */
int btc_tune_set(int key, int value);
int btc_tune_value(int key);   /* current value of a key (0 = built-in policy) */

/* ------------------------------------------------------------------------------------------------
 * Voxelizer.  Replaces spconv.utils.VoxelGeneratorV2.generate (points_to_voxel_3d_np), called at
 * /root/reference/btcdet/datasets/processor/data_processor.py:85,136,177, for a whole batch at once.
 * Semantics: c = floor((p - lo) / vs) in float32, point dropped unless
 * 0 <= c < grid on every axis; voxels numbered by first appearance in input order (per scene);
 * at most max_voxels voxels per scene (later-appearing voxels dropped with all their points);
 * at most max_points points per voxel (earliest kept, input order); padding 0.
 *
 *   points        (n, ld) float32; xyz at columns [xyz_col, xyz_col+3); the C copied features start
 *                 at feat_col (the reference copies the whole row: feat_col == xyz_col, C == ld).
 *   scene_offsets (batch+1) int32, ascending, points of scene b are rows [off[b], off[b+1]).
 *   range[6], vsize[3], grid[3] (x,y,z) host arrays (grid = round((hi-lo)/vs), computed by caller).
 * Outputs (capacity cap_vox = batch*max_voxels rows):
 *   voxels (cap_vox, max_points, C) f32 ; coords (cap_vox, 4) i32 [b,z,y,x] ; num (cap_vox) i32 ;
 *   d_total (1) i32 = number of voxels M (rows [0,M) are valid; rows >= M are left untouched).
 * ws: btc_voxelize_ws_bytes(n, batch, max_points) bytes.
 * ---------------------------------------------------------------------------------------------- */
size_t btc_voxelize_ws_bytes(int n, int batch, int max_points);
int btc_voxelize(const float* points, int n, int ld, int xyz_col, int feat_col, int C,
                 const int32_t* scene_offsets, int batch, const float* h_range, const float* h_vsize,
                 const int32_t* h_grid, int max_points, int max_voxels, float* voxels, int32_t* coords,
                 int32_t* num, int32_t* d_total, void* ws, size_t ws_bytes, void* stream);

/* Cartesian -> cylinder / sphere coordinates of the leading xyz columns, other columns copied.
 * Replaces coords_utils.absxyz_2_cylinxyz_np / absxyz_2_spherexyz_np
 * (/root/reference/btcdet/utils/coords_utils.py:268-292).  mode 1 = cylinder, 2 = sphere.
 * Optional per-row azimuth shift: out[:,1] -= rot_z_deg[scene(row)] when rot_z_deg != NULL is NOT
 * applied here (the reference shifts voxel payloads after voxelization, data_processor.py:148-149;
 * see btc_add_scalar_by_batch). */
int btc_cart_to_occ_coords(const float* in, float* out, int n, int ld, int mode, void* stream);

/* voxels[v, p, col] += sign * rot[b(v)] for every slot p (padded slots too).
 * Replaces `voxels[..., 1] = voxels[..., 1] - data_dict['rot_z']` (data_processor.py:148-149). */
int btc_voxel_shift_col(float* voxels, const int32_t* coords, int m, int max_points, int C, int col,
                        const float* rot_by_batch, float sign, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-scene pre-steps on a resident batch (csrc/prestep.hip).
 *
 * btc_range_mask_compact replaces the point half of DataProcessor.mask_points_and_boxes_outside_range
 * (/root/reference/btcdet/datasets/processor/data_processor.py:23-29 -> common_utils.mask_points_by_range,
 * btcdet/utils/common_utils.py:59-62) for a whole batch: a row is kept iff x in [x_lo, x_hi] and y in [y_lo, y_hi]
 * (inclusive, z not tested, NaN dropped); kept rows keep their order (numpy boolean indexing).  The same mask is applied to
 * an optional second array (`pre_rot_points`, data_processor.py:27-28).
 *   points (n, ld) f32, xy in columns 0,1 ; points_b (n, ld_b) f32 or NULL ; scene_offsets (batch+1) i32
 *   h_range_xyxy  host float[4] = x_lo, y_lo, x_hi, y_hi
 *   out (n, ld), out_b (n, ld_b): rows [0, n') valid ; out_offsets (batch+1) i32: scene offsets after masking,
 *   out_offsets[batch] = n' ; keep_idx (n) i32 or NULL: source row of every kept row.  ws: btc_range_mask_ws_bytes(n).
 *
 * btc_gather_rows: out[i] = src[idx[i]] (DataProcessor.shuffle_points, data_processor.py:41-51: points[shuffle_idx]).
 *   an index outside [0, n_src) writes a zero row and increments *bad_count (if given) instead of reading out of bounds. */
size_t btc_range_mask_ws_bytes(int n);
int btc_range_mask_compact(const float* points, const float* points_b, int n, int ld, int ld_b, const int32_t* scene_offsets, int batch,
                           const float* h_range_xyxy, float* out, float* out_b, int32_t* out_offsets, int32_t* keep_idx, void* ws,
                           size_t ws_bytes, void* stream);
int btc_gather_rows(const float* src, const int32_t* idx, int n_out, int ld, int n_src, float* out, int32_t* bad_count, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Rulebook.  Replaces spconv ops.get_indice_pairs behind
 * SubMConv3d / SparseConv3d / SparseConvTranspose3d / SparseMaxPool3d
 * (/root/reference/btcdet/models/backbones_3d/spconv_backbone.py:12-29).
 * ---------------------------------------------------------------------------------------------- */
/* o = (i + 2p - d(k-1) - 1)/s + 1 (conv), (i-1)s - 2p + k + outpad (transpose), i (subm). host only */
int btc_out_shape(const int32_t* h_in_shape, const int32_t* h_k, const int32_t* h_s, const int32_t* h_p,
                  const int32_t* h_d, const int32_t* h_outpad, int mode, int32_t* h_out_shape);

/* SubM: outputs are the inputs in input order (any order: the cell -> row lookup is a hash with 64-bit cell keys).
 * Kernel sizes must be odd.  nbr_out (n, K) int32, fully written.  nbr_in: (n, K), or NULL -- it is nbr_out's mirror image
 * (nbr_in[i][K-1-k] == nbr_out[i][k]); the apply kernels read nbr_out mirrored instead (BTC_PASS_DGRAD_MIRROR below), so the hot
 * path never materialises it.  ws: btc_rulebook_subm_ws_bytes(n). */
size_t btc_rulebook_subm_ws_bytes(int n);
int btc_rulebook_subm(const int32_t* indices, int n, int batch, const int32_t* h_shape, const int32_t* h_k,
                      const int32_t* h_d, int32_t* nbr_out, int32_t* nbr_in, void* ws, size_t ws_bytes,
                      void* stream);

/* Regular / transposed conv and pooling geometry, two phases around one 4-byte read-back:
 *   phase 1 (count): marks the reachable output cells in a bitmap over the output grid (1 bit per cell) and ranks it
 *                    (csrc/rulebook.hip, "LEVEL"); *d_n_out = n_out.
 *   phase 2 (fill) : out_indices (n_out,4) ascending in (b,z,y,x); nbr_out (n_out,K); nbr_in (n,K).
 * The same ws (btc_rulebook_conv_ws_bytes(batch, out_shape) = out volume / 8 bytes + prefixes) must be passed to both
 * phases, untouched in between.  Limits: one scene's grid < 2^31 cells, batch * grid < 2^43 cells. */
size_t btc_rulebook_conv_ws_bytes(int batch, const int32_t* h_out_shape);
int btc_rulebook_conv_count(const int32_t* indices, int n, int batch, const int32_t* h_in_shape,
                            const int32_t* h_out_shape, const int32_t* h_k, const int32_t* h_s,
                            const int32_t* h_p, const int32_t* h_d, int mode, int32_t* d_n_out, void* ws,
                            size_t ws_bytes, void* stream);
int btc_rulebook_conv_fill(const int32_t* indices, int n, int batch, const int32_t* h_in_shape,
                           const int32_t* h_out_shape, const int32_t* h_k, const int32_t* h_s,
                           const int32_t* h_p, const int32_t* h_d, int mode, int n_out, int32_t* out_indices,
                           int32_t* nbr_out, int32_t* nbr_in, void* ws, size_t ws_bytes, void* stream);

/* Every rulebook of a CHAIN of sparse layers (an encoder / decoder branch: the layers of
 * /root/reference/btcdet/models/backbones_3d/spconv_backbone.py:106-128 or :656-767 in execution order) from the input
 * coordinates alone, with ONE read-back for the whole chain instead of one per strided layer (spconv synchronises in every
 * get_indice_pairs call).  A layer either builds a submanifold rulebook on the current level (kind 0), builds a strided /
 * transposed one and moves to the level it creates (kind 1), is the inverse of layer `ref` (kind 2: continues on ref's
 * input level) or reuses layer ref's rulebook (kind 3: indice_key hit or identical geometry; continues on its output level).
 *   phase A  btc_chain_levels : builds every level on the device.  out_indices[i] (kind-1 layers; capacity h_cap[i] rows from
 *            btc_chain_caps, 16 bytes a row) receives the level's rows in ascending (b,z,y,x) order, d_counts[i] its row
 *            count.  Nothing is read back; level l+1 is marked from level l's rows with their count taken from d_counts.
 *   -- the caller copies d_counts (n_layers int32) to the host and sizes the maps --
 *   phase B  btc_chain_maps   : nbr_out[i] (rows_out_i, K_i) and nbr_in[i] (rows_in_i, K_i) of every kind-0 / kind-1 layer in
 *            one multi-job launch.  A strided layer's nbr_in is written by its input rows probing the output level, its nbr_out
 *            by its OUTPUT rows probing the input level (a gather: no -1 fill, no scattered stores) -- except for a layer that
 *            consumes the chain's arbitrary-order input level, whose nbr_out is filled with -1 and scattered into (hand in
 *            adjacent buffers to make that one fill).  A kind-0 layer's nbr_in[i] may be NULL (mirror image of nbr_out, see
 *            btc_rulebook_subm); with nbr_out[i] == nbr_in[i] == NULL the layer is skipped (the caller built that rulebook).
 * ws (btc_chain_ws_bytes) must be the same, untouched, for both phases; the phases may run on different streams as long
 * as phase B is ordered behind phase A (the detection backbone runs phase A on a side stream beside its first stage). */
#define BTC_CHAIN_MAX_LAYERS 32
typedef struct BtcChainLayer {
  int32_t kind, ref, mode;
  int32_t in_shape[3], out_shape[3], k[3], s[3], p[3], d[3];
} BtcChainLayer;
size_t btc_chain_ws_bytes(const BtcChainLayer* h_layers, int n_layers, int batch, int n0);
int btc_chain_caps(const BtcChainLayer* h_layers, int n_layers, int batch, int n0, int64_t* h_cap);
int btc_chain_levels(const int32_t* indices, int n0, int batch, const BtcChainLayer* h_layers, int n_layers,
                     int32_t* const* h_out_indices, const int64_t* h_cap, int32_t* d_counts, void* ws, size_t ws_bytes, void* stream);
int btc_chain_maps(const int32_t* indices, int n0, int batch, const BtcChainLayer* h_layers, int n_layers, const int32_t* h_counts,
                   int32_t* const* h_out_indices, int32_t* const* h_nbr_out, int32_t* const* h_nbr_in,
                   int32_t* const* h_first_out /* NULL, or per layer NULL / (rows_out) keys of nbr_out for btc_row_orders_keyed (values
                                                  above K mean K) */,
                   int32_t* const* h_first_in /* the same for nbr_in (rows_in) */, void* ws, size_t ws_bytes, void* stream);

/* spconv-layout view of a rulebook: pairs (2,K,n_in) int32 padded with -1, pair_num (K) int32,
 * pairs within an offset ordered by output row (the canonical order of SURVEY.md App. B.4).
 * ws: btc_pairs_from_nbr_ws_bytes(n_out, K); enqueued like everything else (no allocation, no synchronisation). */
size_t btc_pairs_from_nbr_ws_bytes(int n_out, int K);
int btc_pairs_from_nbr(const int32_t* nbr_out, int n_out, int K, int n_in, int32_t* pairs, int32_t* pair_num, void* ws,
                       size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sparse convolution apply.  Replaces spconv's indice_conv / indice_subm_conv /
 * indice_inverse_conv autograd functions: fp32 in, fp32 accumulate on
 * the fp32 MFMA pipe, fused over all K offsets, output-stationary (no atomics, deterministic).
 *   fwd   : out[i]  = bias + sum_k feat[nbr_out[i][k]] @ W[k]
 *   dgrad : din[j]  = sum_k dout[nbr_in[j][k]] @ W[k]^T
 *   wgrad : dW[k]   = sum_i feat[nbr_out[i][k]]^T dout[i]    (ws: btc_conv_wgrad_ws_bytes; nbr_in / n_in are optional
 *           (NULL / -1): when given, the kernel may walk the smaller side of the rulebook, and a layer with <= 8 result channels is
 *           walked over its input rows (conv_wgrad_n.hip).  A submanifold layer stores one map (its backward map is the mirror image,
 *           btc_rulebook_subm): pass nbr_in == nbr_out (the same pointer) and n_in == n_out to say so)
 * An inverse conv (SparseInverseConv3d) is fwd with nbr_in of the cached rulebook as the map.
 * ---------------------------------------------------------------------------------------------- */
int btc_conv_fwd(const float* feat, const float* W, const float* bias /* may be NULL */, const int32_t* nbr_out,
                 int n_out, int K, int Cin, int Cout, float* out, void* stream);
int btc_conv_dgrad(const float* dout, const float* W, const int32_t* nbr_in, int n_in, int K, int Cin,
                   int Cout, float* din, void* stream);
size_t btc_conv_wgrad_ws_bytes(int n_out, int K, int Cin, int Cout, int n_in);
int btc_conv_wgrad(const float* feat, const float* dout, const int32_t* nbr_out, int n_out, const int32_t* nbr_in,
                   int n_in, int K, int Cin, int Cout, float* dW, void* ws, size_t ws_bytes, void* stream);

/* "bf16 features" (BASELINE.json configs[2] and [4]): the activations -- feat / out, dout / din -- are bfloat16 in HBM
 * (half the gather traffic); weights, bias, accumulation and dW stay fp32.  A result is the round-to-nearest-even bf16 of
 * exactly the fp32 fmaf chain btc_conv_fwd / btc_conv_dgrad compute on the same inputs, so parity with the oracle stays
 * bit-exact.  Forward / dgrad need Cin and Cout to be multiples of 16 (BTC_EINVAL otherwise: convert that layer's
 * activations to fp32 and call the fp32 entry point, as btcdet_amd/spconv/ops.py does for the 4/6/34-channel layers). */
int btc_conv_fwd_bf16(const void* feat, const float* W, const float* bias, const int32_t* nbr_out, int n_out, int K, int Cin,
                      int Cout, void* out, void* stream);
int btc_conv_dgrad_bf16(const void* dout, const float* W, const int32_t* nbr_in, int n_in, int K, int Cin, int Cout,
                        void* din, void* stream);
int btc_conv_wgrad_bf16(const void* feat, const void* dout, const int32_t* nbr_out, int n_out, const int32_t* nbr_in,
                        int n_in, int K, int Cin, int Cout, float* dW, void* ws, size_t ws_bytes, void* stream);

/* bfloat16 OPERANDS (BASELINE.json configs[2] / [4]): activations bf16 as above AND a bf16 copy of the weights, multiplied on
 * the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, 16x the fp32 MFMA rate), fp32 accumulate, result rounded to bf16 once.
 * (btc_conv_fwd_bf16 / btc_conv_dgrad_bf16 above keep fp32 weights: same bits as the fp32 fmaf chain, fp32-MFMA speed.)
 *   btc_weights_to_bf16  : W fp32 [K][Cin][Cout] -> w_bf16 [K][Cin][Cout] (dgrad operand) and wt_bf16 [K][Cout][Cin]
 *                          (forward operand); once per optimizer step and layer (K*Cin*Cout*2 bytes each)
 *   btc_conv_fwd_bf16w   : out[i] = bf16(bias + sum_k feat[nbr_out[i][k]] @ bf16(W[k]))     needs Cin % 32 == 0, Cout % 16 == 0
 *   btc_conv_dgrad_bf16w : din[j] = bf16(sum_k dout[nbr_in[j][k]] @ bf16(W[k])^T)           needs Cout % 32 == 0, Cin % 16 == 0
 * Tolerance vs the fp32 path: see csrc/conv_apply_bf16.hip and tests/test_hip_bf16_mfma.py. */
int btc_conv_bf16w_supported(int K, int Cred, int Cres);
int btc_weights_to_bf16(const float* W, int K, int Cin, int Cout, void* w_bf16, void* wt_bf16, void* stream);
/* ... of n weights in one launch (host arrays of device pointers / sizes): a parameter group's layers right behind its optimizer step */
int btc_weights_to_bf16_multi(const float* const* W, void* const* w_bf16, void* const* wt_bf16, const int32_t* K, const int32_t* Cin,
                              const int32_t* Cout, int n, void* stream);
int btc_conv_fwd_bf16w(const void* feat, const void* wt_bf16, const float* bias, const int32_t* nbr_out, int n_out, int K, int Cin,
                       int Cout, void* out, void* stream);
int btc_conv_dgrad_bf16w(const void* dout, const void* w_bf16, const int32_t* nbr_in, int n_in, int K, int Cin, int Cout, void* din,
                         void* stream);

/* fp32 activations, fp32 results, products on the bf16 matrix pipe (csrc/conv_apply_split.hip): every operand is split exactly into
 * three bfloat16 pieces (hi + mid + lo) and a * b is taken as the six largest of the nine piece products, fp32 accumulate --
 * 6 bf16 MFMAs instead of 8 fp32 MFMAs per 32 reduction channels at 16x the rate.  Not the bit pattern of the exact fmaf chain
 * (btc_conv_fwd / btc_conv_dgrad stay the parity reference): within 2e-6 of the result's scale of it, deterministic
 * (tests/test_hip_split.py).  For the layers whose matrix phase is the long pole: Cred % 32 == 0, Cres % 32 == 0.
 *   btc_weights_split3 : W fp32 [K][Cin][Cout] -> w_split [3][K][Cin][Cout] bf16 (dgrad operand), wt_split [3][K][Cout][Cin]
 *                        (forward operand); once per optimizer step and layer (3*K*Cin*Cout*2 bytes each)
 * used through btc_conv_apply_ordered / btc_conv_bn_relu_fwd with operands = BTC_OPERANDS_F32_SPLIT and W = the planes of that pass */
int btc_conv_split_supported(int K, int Cred, int Cres);
/* Scratch memory for a stream: `bytes` of device memory the library may use during any launch on `stream` (the caller owns it and
 * keeps it alive until it registers another buffer, or NULL, for that stream).  Used by the split-operand kernel on levels of a few
 * thousand rows: up to four workgroups share a tile's (offset, chunk) items, write partial sums to slabs of n_rows x Cout floats in
 * the scratch and a second launch adds the slabs in order (deterministic; bias and the BatchNorm statistics move to that launch).
 * Without a buffer (or with one too small for two slabs) such launches run unsplit.  The registry is keyed by (device of `ptr`,
 * stream handle) -- the default stream's handle is 0 on every device --; with ptr == NULL the current device's entry is removed. */
int btc_set_scratch(void* stream, void* ptr, size_t bytes);
/* the host bindings' policy: 1 = an fp32 launch of n_rows rows should take the split-operand kernel (0 always under BTC_TUNE_SPLIT = 1) */
int btc_conv_split_wanted(int K, int Cred, int Cres, int n_rows);
int btc_weights_split3(const float* W, int K, int Cin, int Cout, void* w_split, void* wt_split, void* stream);
/* the same for n weights in one launch (host arrays of device pointers / sizes): a parameter group's layers right after its optimizer step */
int btc_weights_split3_multi(const float* const* W, void* const* w_split, void* const* wt_split, const int32_t* K, const int32_t* Cin,
                             const int32_t* Cout, int n, void* stream);

/* Row-order hints (csrc/row_order.hip).  The apply kernels work on tiles of 16 consecutive map rows and pay for every
 * offset ANY row of the tile has; which rows share a tile changes no result.  btc_row_orders sorts the rows of up to
 * BTC_ROW_ORDER_MAX_MAPS neighbour maps by their FIRST PRESENT OFFSET (lowest k with nbr[row][k] >= 0; K if none) -- a
 * stable counting sort inside blocks of 2048 consecutive rows, one launch for all maps; rows of a strided layer's dgrad
 * map that can share offsets at all end up together: order[off_j + t] = the row of map j that tile slot t works on,
 * off_j = n_rows[0] + .. + n_rows[j-1] (K <= 64).  btc_conv_apply_ordered is btc_conv_fwd / _dgrad
 * (+ _bf16 / _bf16w, by `operands`) with such a hint -- any permutation of 0..n_rows-1 is valid; order == NULL is the map
 * order; the result is bit-identical either way.
 *   pass     : BTC_PASS_FWD   dst[i] = bias + sum_k src[nbr[i][k]] @ W[k]        (nbr = nbr_out, n_rows = n_out)
 *              BTC_PASS_DGRAD dst[j] = sum_k src[nbr[j][k]] @ W[k]^T             (nbr = nbr_in,  n_rows = n_in, bias NULL)
 *              BTC_PASS_DGRAD_MIRROR  the same for a SUBMANIFOLD layer given its nbr_OUT: dst[j] = sum_k src[nbr[j][K-1-k]] @ W[k]^T
 *                             (nbr_in[j][k] == nbr_out[j][K-1-k] there; same summation order, same bits as BTC_PASS_DGRAD on nbr_in)
 *   operands : BTC_OPERANDS_F32; BTC_OPERANDS_BF16_ACT (bf16 src / dst, fp32 W); BTC_OPERANDS_BF16 (bf16 src / dst and W =
 *              the bf16 copy btc_weights_to_bf16 made for that pass: wt_bf16 for FWD, w_bf16 for DGRAD); BTC_OPERANDS_F32_SPLIT
 *              (fp32 src / dst, W = the planes btc_weights_split3 made for that pass: wt_split for FWD, w_split for DGRAD)
 * btc_conv_wgrad_ordered: btc_conv_wgrad (bf16_act = 0) / btc_conv_wgrad_bf16 (1) walking the rows in the given order(s)
 * (either may be NULL); dW is then the fp32 sum in THAT row order -- deterministic for a given order. */
#define BTC_ROW_ORDER_MAX_MAPS 64
#define BTC_PASS_FWD 0
#define BTC_PASS_DGRAD 1
#define BTC_PASS_DGRAD_MIRROR 2
#define BTC_OPERANDS_F32 0
#define BTC_OPERANDS_BF16_ACT 1
#define BTC_OPERANDS_BF16 2
#define BTC_OPERANDS_F32_SPLIT 3
int btc_row_orders(const int32_t* const* nbrs /* host array of device pointers */, const int32_t* n_rows /* host */,
                   const int32_t* Ks /* host */, int n_maps, int32_t* order /* device, sum n_rows */, void* stream);
/* the same with the sort keys handed in where they exist: firsts[j] (n_rows[j]) = first present offset of every row of map j (K for a
 * row without neighbours), as btc_chain_maps writes them while it fills the map -- order_local then reads 4 bytes a row instead of
 * the whole map; firsts == NULL or firsts[j] == NULL: the keys are taken from nbrs[j] */
int btc_row_orders_keyed(const int32_t* const* nbrs, const int32_t* const* firsts, const int32_t* n_rows, const int32_t* Ks, int n_maps,
                         int32_t* order, void* stream);
int btc_conv_apply_ordered(int pass, int operands, const void* src, const void* W, const float* bias, const int32_t* nbr,
                           const int32_t* order, int n_rows, int K, int Cin, int Cout, void* dst, void* stream);
/* the same with the row count of `src` stated.  BTC_OPERANDS_F32_SPLIT gathers through 32-bit byte offsets: its launches NEED src_rows
 * (>= 0) and are refused with BTC_EINVAL -- on the host, nothing traps on the device -- for a source of 4 GB or more (take
 * BTC_OPERANDS_F32 then).  btc_conv_apply_ordered knows the count only for BTC_PASS_DGRAD_MIRROR (a submanifold layer: n_rows). */
int btc_conv_apply_src(int pass, int operands, const void* src, long long src_rows, const void* W, const float* bias, const int32_t* nbr,
                       const int32_t* order, int n_rows, int K, int Cin, int Cout, void* dst, void* stream);
int btc_conv_wgrad_ordered(int bf16_act, const void* feat, const void* dout, const int32_t* nbr_out, int n_out,
                           const int32_t* nbr_in, int n_in, const int32_t* order_out, const int32_t* order_in, int K, int Cin,
                           int Cout, float* dW, void* ws, size_t ws_bytes, void* stream);
/* The weight gradient in two halves, for a caller that has MANY of them in flight and needs none before a common point (the backward
 * pass of a network: spconv's indice_conv_backward, one per layer, /root/reference/btcdet/models/backbones_3d/spconv_backbone.py:7-43
 * through spconv/functional.py -- every dW is first read by the optimizer).  btc_conv_wgrad_slabs = btc_conv_wgrad_ordered without
 * its final reduction: *n_slabs = S >= 1 partial sums of K Cin Cout floats each, in `ws` (dW untouched), or 0: dW is complete (a layer
 * without rows: zeros).  btc_wgrad_reduce_multi adds the slabs of up to any number of such jobs in ONE launch per
 * BTC_WGRAD_MULTI_MAX jobs (host arrays of device pointers / sizes): dWs[j][e] = sum_s parts[j][s * counts[j] + e] in slab order --
 * the same sums, in the same order, as the one-call entry points.  `ws` must stay untouched until that launch has run.
 * Row counts: n_in > 0 is the row count of `feat` also when nbr_in is NULL (NULL with 0 = unknown); with it the walk may run on the bf16 matrix pipe
 * (csrc/conv_wgrad_x.hip: bf16 activations as stored; fp32 activations as three exact bf16 pieces, six products per pair, unless
 * BTC_TUNE_SPLIT = 1 or BTC_TUNE_WGRAD_X = 1), whose 32-bit gather offsets need both operands under 4 GB -- larger operands, or an
 * unknown row count, take the fp32-pipe kernels. */
#define BTC_WGRAD_MULTI_MAX 64
int btc_conv_wgrad_slabs(int bf16_act, const void* feat, const void* dout, const int32_t* nbr_out, int n_out, const int32_t* nbr_in,
                         int n_in, const int32_t* order_out, const int32_t* order_in, int K, int Cin, int Cout, float* dW, void* ws,
                         size_t ws_bytes, int* n_slabs, void* stream);
int btc_wgrad_reduce_multi(const float* const* parts, float* const* dWs, const int* n_slabs, const long long* counts, int n_jobs,
                           void* stream);


/* One parameter group's optimizer step of the reference's loop (tools/train_utils/train_utils.py:121-124: clip_grad_norm_,
 * then OptimWrapper.step = decoupled weight decay + torch.optim.Adam) in three launches (csrc/optim.hip).  The group's
 * parameters and the two Adam moments are ONE flat fp32 buffer each; gradient s (fp32, contiguous, the size of parameter s)
 * is read where it lies through grads[s].  The parameters are cut into chunks of <= 1024 elements that never straddle two
 * parameters: chunk c covers elements [chunk_off[c], chunk_off[c] + chunk_len[c]) of parameter s, which sit at
 * chunk_flat[c] in the flat buffers, chunk_seg[c] = s % BTC_ADAM_MAX_SEGMENTS; seg_chunk0[s] = first chunk of parameter s,
 * seg_chunk0[n_seg] = n_chunks (chunks ordered by parameter).  chunk_* are device arrays, grads / seg_chunk0 host arrays.
 *   coef = 1 / max(1, (||g|| + 1e-6) / clip)   (clip <= 0: no clipping);   g *= coef;   p *= 1 - weight_decay * lr;
 *   m = lerp(m, g, 1 - beta1);  v = beta2 v + (1 - beta2) g^2;  p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * step = the 1-based count of this update.  ws: btc_adam_group_ws_bytes(n_chunks); its first 8 bytes hold ||g||^2 (double)
 * after the call. */
#define BTC_ADAM_MAX_SEGMENTS 448
int btc_adam_max_segments(void);   /* = BTC_ADAM_MAX_SEGMENTS of the loaded library: what chunk_seg[] must be reduced by */
size_t btc_adam_group_ws_bytes(int n_chunks);
/* flat[chunk_flat[c] + e] = grads[s][chunk_off[c] + e]: packs a list of gradients into a flat bucket (the buffer a
 * data-parallel all-reduce runs on) in one launch; same chunk tables as btc_adam_group_step */
int btc_grads_pack(const float* const* grads, int n_seg, const int32_t* chunk_seg, const int32_t* chunk_off,
                   const int32_t* chunk_len, const int64_t* chunk_flat, const int32_t* seg_chunk0, float* flat, void* stream);
int btc_adam_group_step(const float* const* grads, int n_seg, const int32_t* chunk_seg, const int32_t* chunk_off,
                        const int32_t* chunk_len, const int64_t* chunk_flat, const int32_t* seg_chunk0, int n_chunks,
                        float* params, float* exp_avg, float* exp_avg_sq, long long step, float lr, float beta1, float beta2,
                        float eps, float weight_decay, float clip, void* ws, size_t ws_bytes, void* stream);

/* Sparse max-pool (spconv indice_maxpool, App. B.6): out = max(0, max_k feat[nbr_out[i][k]]);
 * backward routes dout to every input equal to its output. */
int btc_maxpool_fwd(const float* feat, const int32_t* nbr_out, int n_out, int K, int C, float* out, void* stream);
int btc_maxpool_bwd(const float* feat, const float* out, const float* dout, const int32_t* nbr_in, int n_in,
                    int K, int C, float* din, void* stream);

/* SparseConvTensor.dense() (height_compression.py:21, occ_head_3D.py:46,51):
 * dense (B,C,D,H,W) must be zero-filled by the caller; backward gathers the active cells. */
int btc_dense_fwd(const float* feat, const int32_t* indices, int n, int C, const int32_t* h_shape, float* dense,
                  void* stream);
int btc_dense_bwd(const float* ddense, const int32_t* indices, int n, int C, const int32_t* h_shape, float* dfeat,
                  void* stream);
/* dense() of a tensor whose channels are two heads side by side -- the occupancy head's merged conv_cls | conv_res output
 * (occ_head_3D.py:25-31,46,51) -- into the two dense maps in one launch: dense_a (B,Ca,D,H,W), dense_b (B,Cb,D,H,W), both zero-filled
 * by the caller (one fill when they are carved out of one buffer); backward gathers both gradients (either may be NULL = zeros)
 * into dfeat (n, Ca + Cb).  Same values as btc_dense_fwd / _bwd on the column slices. */
int btc_dense_split_fwd(const float* feat, const int32_t* indices, int n, int Ca, int Cb, const int32_t* h_shape, float* dense_a,
                        float* dense_b, void* stream);
int btc_dense_split_bwd(const float* grad_a, const float* grad_b, const int32_t* indices, int n, int Ca, int Cb,
                        const int32_t* h_shape, float* dfeat, void* stream);
/* prob (B, cells) = softmax over the two channels of logit (B, 2, cells), last channel, times mask (B, cells) of 0 / 1 bytes: the
 * occupancy head's `logit2prob(logit)[:, -1] * general_cls_loss_mask` (/root/reference/btcdet/models/occ_dense_heads/occ_head_3D.py:34-38)
 * in one launch; no gradient (the caller uses it where the probability is a detached input: PASS_GRAD False). */
int btc_occ_prob(const float* logit, const unsigned char* mask, int B, long long ncell, float* prob, void* stream);
/* out (n, cout) = [a (n, ca) | b (n, cb) | zeros]: the detection backbone's sparse_cat (spconv_backbone.py:869-873) together with the
 * zero channels the apply kernels want (34 -> 64 / 48); backward splits grad (n, cout) into da, db.  elem_bytes: 4 (fp32) | 2 (bf16). */
int btc_cat_pad_fwd(const void* a, int ca, const void* b, int cb, long long n, int cout, int elem_bytes, void* out, void* stream);
int btc_cat_pad_bwd(const void* grad, int cout, long long n, int elem_bytes, void* da, int ca, void* db, int cb, void* stream);
/* A single wave that idles for `microseconds` on `stream` (0 .. 100000).  Not part of the data path: btcdet_amd/streams.py uses it to find
 * out which of its streams the runtime dealt the same hardware queue (work on one then waits behind work on the other), see DESIGN.md section 5. */
int btc_spin(int microseconds, void* stream);
/* out[0] = ka sum a^2 + kb sum b^2 over two tensors of fp32 (x_bf16 = 0) or bfloat16 (1) elements, fp64 accumulation in a fixed order
 * (b may be NULL with nb = 0); backward: da = a * (g[0] * ka2), db = b * (g[0] * kb2), the factor rounded to the tensor's type first.
 * The L2 stand-in loss of the heads behind the hot path (btcdet_amd/trainer.py stand_in_det_loss), not a reference operator.
 * ws: btc_sumsq2_ws_bytes() bytes, first 256 zero before the first call (kept zero). */
size_t btc_sumsq2_ws_bytes(void);
int btc_sumsq2_fwd(const void* a, long long na, int a_bf16, double ka, const void* b, long long nb, int b_bf16, double kb, float* out,
                   void* ws, size_t ws_bytes, void* stream);
int btc_sumsq2_bwd(const void* a, long long na, int a_bf16, float ka2, void* da, const void* b, long long nb, int b_bf16, float kb2,
                   void* db, const float* g, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Sorted-unique re-voxelization.  Replaces torch.unique(coords, dim=0, sorted=True,
 * return_inverse, return_counts) + sort + scatter-pad of PassOccVox
 * (/root/reference/btcdet/models/occ_pnt/add_occ_template.py:248-268): points with integer cell
 * coords [b,z,y,x] are grouped per cell, cells ascending in (b,z,y,x), points of a cell in input order.
 *   phase 1: *d_m = number of cells, *d_pmax = max points per cell       (4+4 byte read-back)
 *   phase 2: voxels (m,pmax,C) zero padded, vcoords (m,4) int64, vnum (m) int64
 * ws: btc_revoxelize_ws_bytes(n, batch, shape).
 * ---------------------------------------------------------------------------------------------- */
size_t btc_revoxelize_ws_bytes(int n, int batch, const int32_t* h_shape);
int btc_revoxelize_count(const int64_t* coords, int n, int batch, const int32_t* h_shape, int32_t* d_m,
                         int32_t* d_pmax, void* ws, size_t ws_bytes, void* stream);
int btc_revoxelize_fill(const float* points, const int64_t* coords, int n, int C, int batch,
                        const int32_t* h_shape, int m, int pmax, float* voxels, int64_t* vcoords,
                        int64_t* vnum, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Occupancy / occlusion target generator.  Replaces OccTargets3D.forward -> create_voxel_res_label
 * (/root/reference/btcdet/models/occ_pnt/occ_training_targets/occ_targets_3d.py:18-171,
 * occ_targets_template.py:82-255,330-447; geometry helpers coords_utils.py:180-239,
 * point_box_utils.py:70-121,241-329) for COORD_TYPE cylinder, REG True: seven launches, no host sync.
 *
 * In/out:
 *   voxels (M,P,C) f32   cylinder payload (rho, azimuth_deg - rot_z, z, ...) -> rewritten IN PLACE to
 *                        absolute xyz (USE_ABSXYZ True, occ_targets_3d.py:45-47), padded slots included
 *   voxel_coords (M,4) i32 [b,z,y,x]; voxel_num (M) i32
 *   gt_boxes (B,G,8) f32 [x,y,z,dx,dy,dz,yaw,cls]; gt_num (B) i32; mirr_flag (B,G) f32
 *   bm_points (n_bm,4) f32 [b,x,y,z] (may be NULL when n_bm == 0); rot_z (B) f32 degrees
 *   centers (nz,ny,nx,3) f32: Cartesian centres of the cylinder cells (detector3d_template.py:52-63)
 * Outputs (BtcOccBuffers, all [B,nz,ny,nx] unless noted; caller allocates, need not be initialised):
 *   the eleven uint8 masks and forebox_label (int8) of occ_targets_template.py:360-400,
 *   general_cls_loss_mask_float / general_reg_loss_mask_float (f32), res_mtrx [B,3,nz,ny,nx] f32,
 *   pos_all_num (1) i32.
 * ---------------------------------------------------------------------------------------------- */
typedef struct BtcOccConfig {
  int32_t batch;
  int32_t grid[3];          /* cylinder grid nx, ny, nz (209,157,9) */
  int32_t sphere_grid[3];   /* support sphere grid nx, ny, nz (214,157,49) */
  int32_t dist_kern[3];     /* DIST_KERN z,y,x (5,9,5) */
  int32_t concede_x;        /* CONCEDE_X or DIST_KERN[-1]//2 when HALF_X */
  int32_t empt_sur_thresh;  /* EMPT_SUR_THRESH, < 0 or >= 9 disables the empty-ray fix */
  int32_t max_boxes;        /* G */
  int32_t use_box_weight;   /* BOX_WEIGHT != 1.0 */
  float occ_range[6], occ_voxel[3];
  float sphere_range[6], sphere_voxel[3];
  float det_zmin, det_zmax; /* DATA_CONFIG.POINT_CLOUD_RANGE[2], [5] */
  float w_fore_cls, w_mirr_cls, w_bm_cls, w_neg_cls, w_fore_res, w_mirr_res, w_bm_res, box_weight;
  /* optional DEVICE table (snz, sny, snx) i32: the cylinder cell (z*ny*nx + y*nx + x) the corner of sphere cell [sz][sy][sx]
   * back-projects into, -1 = out of range (occ_targets_template.py:146-152 -- a function of the two grids alone, so it is
   * static).  NULL: the kernel evaluates the back-projection inline with correctly-rounded transcendentals.  The reference
   * quantises these values exactly ON azimuth cell boundaries, so its result depends on the last ulp of the platform's
   * cos / sin / atan2 / sqrt; a table evaluated with that platform's arithmetic reproduces its cells exactly
   * (btc_occ_backproject_lut fills one with the device's own arithmetic). */
  const int32_t* backproject_lut;
  int32_t reverse_vis;      /* MODEL.OCC.PARAMS.REVERSE_VIS (occ_targets_template.py:110-134): 0 NOTHING (configured), 1 VCC, 2 BACK_TRACK */
  int32_t vis_half;         /* VCC: (DIST_KERN[2] + 1) / 2 cells in front of a hit stay visible */
} BtcOccConfig;

typedef struct BtcOccBuffers {
  uint8_t *vcc_mask, *voxelwise_mask, *bm_voxelwise_mask, *occ_voxelwise_mask, *fore_voxelwise_mask, *pos_mask,
      *general_cls_loss_mask, *occ_fore_cls_mask, *occ_mirr_cls_mask, *occ_bm_cls_mask, *general_reg_loss_mask;
  int8_t* forebox_label;
  float *general_cls_loss_mask_float, *general_reg_loss_mask_float, *res_mtrx;
  int32_t* pos_all_num;
} BtcOccBuffers;

size_t btc_occ_targets_ws_bytes(const BtcOccConfig* cfg);
/* lut (snz, sny, snx) i32 <- the back-projection table above in the device's correctly-rounded arithmetic (cfg->batch unused) */
int btc_occ_backproject_lut(const BtcOccConfig* cfg, int32_t* lut, void* stream);
int btc_occ_targets(const BtcOccConfig* cfg, float* voxels, const int32_t* voxel_coords, const int32_t* voxel_num,
                    int M, int max_points, int C, const float* gt_boxes, const int32_t* gt_num, const float* mirr_flag,
                    const float* bm_points, int n_bm, const float* rot_z, const float* centers,
                    const BtcOccBuffers* out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused BatchNorm1d (+ReLU) over sparse-tensor features (N,C).  Replaces the nn.BatchNorm1d(eps=1e-3,
 * momentum=0.01) + nn.ReLU pair that spconv.SparseSequential applies to `.features` after every sparse conv
 * (/root/reference/btcdet/models/backbones_3d/spconv_backbone.py:33-43,96,634).  Training mode normalises with
 * the batch statistics (biased variance) and updates running_mean / running_var (unbiased) / num_batches_tracked
 * exactly like torch; eval mode uses the running statistics.  ws: btc_bn_ws_bytes(C) bytes whose FIRST 256 bytes
 * must be zero before the first call and are left zero by every call (arrival counter).
 * ---------------------------------------------------------------------------------------------- */
size_t btc_bn_ws_bytes(int C);
int btc_bn_relu_fwd(const float* x, int N, int C, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, long long* num_batches_tracked, float momentum, float eps, int training,
                    int relu, float* y, float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* stream);
int btc_bn_relu_bwd(const float* x, const float* y, const float* dy, int N, int C, const float* gamma,
                    const float* save_mean, const float* save_rstd, int training, int relu, float* dx, float* dgamma,
                    float* dbeta, void* ws, size_t ws_bytes, void* stream);
/* bfloat16 activations (x, y, dy, dx); parameters, statistics (fp64 partial sums) and their gradients stay fp32 */
int btc_bn_relu_fwd_bf16(const void* x, int N, int C, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, long long* num_batches_tracked, float momentum, float eps, int training,
                         int relu, void* y, float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* stream);
int btc_bn_relu_bwd_bf16(const void* x, const void* y, const void* dy, int N, int C, const float* gamma,
                         const float* save_mean, const float* save_rstd, int training, int relu, void* dx, float* dgamma,
                         float* dbeta, void* ws, size_t ws_bytes, void* stream);
/* out[c] = sum_r x[r][c] of an (N,C) matrix: the bias gradient of a sparse conv (grad_out.sum(0) in the reference's
 * autograd graph, spconv v1.2.1 SparseConvFunction.backward).  fp64 accumulation in a fixed order.  ws as above. */
/* conv (forward, operands as btc_conv_apply_ordered) -> training-mode BatchNorm1d -> [ReLU] of the reference's post_act_block
 * (spconv_backbone.py:33-43) in TWO launches: the conv kernel's epilogue gathers the batch statistics of its result (fp64 slot
 * sums, last workgroup finalises mean / rstd / running statistics: csrc/bn_fuse.h), the second launch normalises.  x (n_rows, Cout)
 * receives the conv result (saved for backward), y the BatchNorm output.  fuse_ws: btc_bn_fuse_ws_bytes() bytes, ZERO-initialised
 * once and then owned by this entry point on ONE stream (it leaves them zeroed); NULL, or BTC_OPERANDS_BF16, or Cout > 1024: the
 * statistics come from the separate pass of btc_bn_relu_fwd (ws / ws_bytes as there).  n_rows >= 1. */
size_t btc_bn_fuse_ws_bytes(void);
int btc_conv_bn_relu_fwd(int operands, const void* src, const void* W, const float* bias, const int32_t* nbr, const int32_t* order,
                         int n_rows, int K, int Cin, int Cout, void* x, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu, void* y,
                         float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* fuse_ws, void* stream);
/* ... with the row count of `src` stated (required for BTC_OPERANDS_F32_SPLIT, see btc_conv_apply_src) */
int btc_conv_bn_relu_fwd_src(int operands, const void* src, long long src_rows, const void* W, const float* bias, const int32_t* nbr, const int32_t* order,
                         int n_rows, int K, int Cin, int Cout, void* x, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu, void* y,
                         float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* fuse_ws, void* stream);
int btc_col_sum(const float* x, int N, int C, float* out, void* ws, size_t ws_bytes, void* stream);
int btc_col_sum_bf16(const void* x, int N, int C, float* out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * PassOccVox, fused.  Replaces /root/reference/btcdet/models/occ_pnt/pass_occ_vox.py:10-59 with
 * add_occ_template.py:78-190,248-268: per scene the cells with p > OCC_THRESH (exact top-MAX_NUM_OCC_PNTS by
 * probability when there are more), cell centre (+ predicted residual) -> Cartesian -> detection-grid cell, merged
 * with the valid points of the detection voxels into lexicographically sorted voxels (M'',Pmax,C+code_dim).
 *   count: enqueues everything up to the member lists; d_info (2+B) i32 = [M'', Pmax, added points per scene]
 *   fill : voxels (M'',Pmax,C+code) f32, vcoords (M'',4) i64 [b,z,y,x], vnum (M'') i64,
 *          occ_pnts (K,4) f32 [x,y,z,prob], occ_b (K) i64, K = sum of the per-scene counts
 * Selected cells are processed in ascending cell order (the reference's topk(sorted=False) order is unspecified).
 * ---------------------------------------------------------------------------------------------- */
typedef struct BtcPovConfig {
  int32_t batch, max_k;
  int32_t occ_grid[3];      /* nx, ny, nz of the occupancy (cylinder) grid */
  int32_t det_grid[3];      /* nx, ny, nz of the detection grid */
  float occ_origin[3], occ_voxel[3];
  float det_origin[3], det_voxel[3];
  float occ_thresh, inten;
  int32_t code_dim;         /* CODE_NUM_DIM: appended [prob, flag] channels */
} BtcPovConfig;

size_t btc_pass_occ_vox_ws_bytes(const BtcPovConfig* cfg, int M, int P);
int btc_pass_occ_vox_count(const BtcPovConfig* cfg, const float* probs /* B,nz,ny,nx */, const float* residuals /* B,3,nz,ny,nx or NULL */,
                           const uint8_t* use_occ /* B or NULL */, const float* rot_z /* B or NULL */, const int32_t* det_coords,
                           const int32_t* det_num, int M, int P, int C, int32_t* d_info, void* ws, size_t ws_bytes, void* stream);
int btc_pass_occ_vox_fill(const BtcPovConfig* cfg, const float* det_voxels, int M, int P, int C, int m, int pmax, int k_total,
                          float* voxels, int64_t* vcoords, int64_t* vnum, float* occ_pnts, int64_t* occ_b, void* ws,
                          size_t ws_bytes, void* stream);
/* the same, and int32 twins of the merged voxels' coordinates (m, 4) and point counts (m) beside the int64 tensors the reference's
 * torch.unique hands on (either may be NULL): what the next modules of this package read, without a conversion launch each */
int btc_pass_occ_vox_fill_i32(const BtcPovConfig* cfg, const float* det_voxels, int M, int P, int C, int m, int pmax, int k_total,
                              float* voxels, int64_t* vcoords, int64_t* vnum, float* occ_pnts, int64_t* occ_b, int32_t* vcoords32,
                              int32_t* vnum32, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Occupancy head losses, fused.  Replaces OccHeadTemplate.get_loss (occ_head_template.py:52-111) with
 * SoftmaxFocalClassificationLoss (alpha 1, gamma 2, eps 1e-6; loss_utils.py:140-152) and WeightedSmoothL1Loss
 * (beta; loss_utils.py:199-233) over the dense head outputs:
 *   out2[0] = w_cls * sum_{cls_mask} cls_w * FL / max(sum cls_w, 1);  out2[1] = w_res * sum_{reg_mask} reg_w * SL1 / max(sum reg_w, 1)
 * logit (B,2,ncell) f32, res / res_target (B,3,ncell) f32 (res may be NULL: no regression), masks (B,ncell) u8, weights f32.
 * norms2 (2) f32 is saved for the backward, which writes d_logit / d_res (dense, zero-filled by the caller).
 * ws: btc_occ_loss_ws_bytes() bytes, first 256 zero before the first call (kept zero).
 * ---------------------------------------------------------------------------------------------- */
size_t btc_occ_loss_ws_bytes(void);
int btc_occ_loss_fwd(const float* logit, const float* res, const float* res_target, const uint8_t* pos_mask,
                     const uint8_t* cls_mask, const float* cls_w, const uint8_t* reg_mask, const float* reg_w, int B,
                     long long ncell, float beta, float w_cls, float w_res, float* out2, float* norms2, void* ws, size_t ws_bytes,
                     void* stream);
int btc_occ_loss_bwd(const float* logit, const float* res, const float* res_target, const uint8_t* pos_mask,
                     const uint8_t* cls_mask, const float* cls_w, const uint8_t* reg_mask, const float* reg_w, int B,
                     long long ncell, float beta, const float* norms2, const float* grad2, float* d_logit, float* d_res, void* stream);
/* the same pair for a caller that wants the SUM (OccHeadTemplate.get_loss returns occ_loss_cls + occ_loss_res, occ_head_template.py:
 * 52-111): out3[2] = out3[0] + out3[1] comes out of the forward launch, and the backward takes ONE upstream gradient (of the sum) and
 * writes EVERY cell of d_logit / d_res (zeros outside the masks): the caller allocates them uninitialised. */
int btc_occ_loss_fwd_total(const float* logit, const float* res, const float* res_target, const uint8_t* pos_mask,
                           const uint8_t* cls_mask, const float* cls_w, const uint8_t* reg_mask, const float* reg_w, int B,
                           long long ncell, float beta, float w_cls, float w_res, float* out3, float* norms2, void* ws, size_t ws_bytes,
                           void* stream);
int btc_occ_loss_bwd_total(const float* logit, const float* res, const float* res_target, const uint8_t* pos_mask,
                           const uint8_t* cls_mask, const float* cls_w, const uint8_t* reg_mask, const float* reg_w, int B,
                           long long ncell, float beta, const float* norms2, const float* grad_total, float* d_logit, float* d_res,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * Voxel feature encoders, one launch each.  Replace the torch arithmetic of
 * /root/reference/btcdet/models/backbones_3d/vfe/mean_vfe.py:27-38 (maxprob = False) and occ_vfe.py:24-55.
 * voxels (M,P,C) float32; num_points (M,) float32 (num_is_float = 1, as load_data_to_gpu leaves it) or int32 (0).
 *   btc_mean_vfe : out (M,C) = sum over the P slots / max(count, 1)
 *   btc_occ_vfe  : slots whose last channel is >= 0.05 are occupancy points; feat (M,F) = [mean of the R raw channels
 *                  over the raw slots (over the occupancy slots for voxels holding only occupancy points) | max of the
 *                  F - R code channels over ALL P slots]; occ (M,F-R) = those maxima
 * ------------------------------------------------------------------------------------------------ */
int btc_mean_vfe(const float* voxels, const void* num_points, int num_is_float, int M, int P, int C, float* out, void* stream);
int btc_occ_vfe(const float* voxels, const void* num_points, int num_is_float, int M, int P, int F, int R, float* feat,
                float* occ, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Rotated BEV overlap / IoU and NMS (SURVEY.md §8f row 1, next to the hot path).  Replaces the compiled module
 * btcdet.ops.iou3d_nms.iou3d_nms_cuda (/root/reference/btcdet/ops/iou3d_nms/src/iou3d_nms.cpp:40-188:
 * boxes_overlap_bev_gpu, boxes_iou_bev_gpu, nms_gpu, nms_normal_gpu; kernels iou3d_nms_kernel.cu:107-362).
 * boxes: (N,7) float32 [x, y, z, dx, dy, dz, heading].
 *   btc_boxes_pairwise_bev : out (na, nb) = overlap area (mode 0) or BEV IoU (mode 1)
 *   btc_nms                : boxes already sorted by descending score; box j > i is suppressed by a kept box i when
 *                            iou(i, j) > thresh (rotated = 1: rotated BEV IoU, 0: axis-aligned, the "normal" variant);
 *                            keep (n) int64 gets the kept positions ascending, *d_num_keep (device int32) their number.
 *                            The suppression mask stays in `ws` on the device (the reference copies it to the host and
 *                            walks it there); nothing is synchronised.
 * ------------------------------------------------------------------------------------------------ */
int btc_boxes_pairwise_bev(const float* boxes_a, int na, const float* boxes_b, int nb, int mode, float* out, void* stream);
size_t btc_nms_ws_bytes(int n);
int btc_nms(const float* boxes_sorted, int n, float thresh, int rotated, long long* keep, int32_t* d_num_keep, void* ws,
            size_t ws_bytes, void* stream);
/* The same greedy chain, stopped at max_keep kept boxes, for a BATCH of scenes in shared launches and without any host read-back:
 * boxes_sorted (batch, n, 7) by descending score per scene; keep (batch, max_keep) int64 = the first max_keep entries of btc_nms's
 * keep list, padded with -1; num_keep (batch).  What model_nms_utils.class_agnostic_nms keeps after its
 * `selected[:NMS_POST_MAXSIZE]` (/root/reference/btcdet/models/model_utils/model_nms_utils.py:6-25) -- the decision for a box depends
 * on the kept boxes before it only, so the truncated chain is exact.  ws: btc_nms_topk_ws_bytes(batch, n, max_keep). */
/* ------------------------------------------------------------------------------------------------
 * ROI head, pooling stage: trilinear read-out of a sparse tensor at lattice points WITHOUT densifying it.  Replaces
 * bilinear_interpolate_torch-style read-outs of /root/reference/btcdet/models/roi_heads/conv_head.py:509-610
 * (common_utils.py:247-311, normalize False) on `x.dense()`: 8 advanced-indexing gathers of (points, C) forward and 8
 * index_put(accumulate) backward over EVERY lattice point, four fifths of which read zeros and are dropped right after.
 *   btc_trilinear_corners: xyz (n_points, 3) world coordinates, points_per_batch consecutive points per scene; range_lo (x, y, z),
 *     voxel (x, y, z), stride_zyx, grid_dhw of the sparse tensor; cell_row (batch, D, H, W) i32 = row of the cell or -1 ->
 *     rows (n_points, 8) i32 (corner order dz, dy, dx; -1 = empty / outside), weights (n_points, 8) f32 (|w|, 0 where no row),
 *     flag (n_points) u8 = some corner contributes; row_live (n_rows) u8 or NULL: rows with a non-zero entry -- the reference keeps a
 *     point iff its interpolated vector has one, so an all-zero row is read (and receives its gradient) but keeps no point alive
 *   btc_trilinear_gather: rows / weights (n_keep, 8) of the KEPT points -> out (n_keep, C) = sum_c weights[p][c] * feat[rows[p][c]] in
 *     corner order
 *   btc_trilinear_scatter: the adjoint, deterministic: pair_sorted (n_pairs) i32 = the (point, corner) pairs e = 8 p + c stably sorted by
 *     the row they read, seg (n_rows + 1) i32 = each row's segment -> grad_feat (n_rows, C) = sum in segment order of
 *     weights[e] * grad_out[e >> 3]
 * ---------------------------------------------------------------------------------------------- */
int btc_trilinear_corners(const float* xyz, long long n_points, long long points_per_batch, const float* range_lo, const float* voxel,
                          const float* stride_zyx, const int32_t* grid_dhw, int batch, const int32_t* cell_row, const uint8_t* row_live,
                          int32_t* rows, float* weights, uint8_t* flag, void* stream);
int btc_trilinear_gather(const float* feat, int C, long long n_keep, const int32_t* rows, const float* weights, float* out, void* stream);
int btc_trilinear_scatter(const float* grad_out, int C, const int32_t* pair_sorted, const int32_t* seg, const float* weights, int n_rows,
                          float* grad_feat, void* stream);

size_t btc_nms_topk_ws_bytes(int batch, int n, int max_keep);
int btc_nms_topk(const float* boxes_sorted, int batch, int n, float thresh, int rotated, int max_keep, long long* keep,
                 int32_t* num_keep, void* ws, size_t ws_bytes, void* stream);


/* ------------------------------------------------------------------------------------------------
 * pointnet2_stack.  Entry points = the functions the reference binds from its compiled module
 * `pointnet2_stack_cuda` (/root/reference/btcdet/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp; call sites
 * pointnet2_utils.py:34-36,76,98,208,237,271,289).  Stacked batches: rows of scene b are contiguous, *_batch_cnt (B) i32.
 *   btc_ball_query: idx (M,nsample) i32 = the first nsample points of the query's scene, in index order, with
 *     inner_radius^2 <= d^2 < outer_radius^2 (inner_radius < 0: plain ball), as indices LOCAL to the scene; unused slots
 *     repeat the first hit; a query without hits gets idx[q][0] = -1 and zeros (ball_query_gpu.cu:15-66,
 *     shell_query_gpu.cu:15-68; the reference zero-fills idx before the launch, so does this call)
 *   btc_group_points: out (M,C,nsample) = features[scene start + idx]; _grad accumulates into grad_features (N,C),
 *     zeroed by the call (group_points_gpu.cu:14-46,64-97)
 *   btc_furthest_point_sampling: xyz (B,N,3), temp (B,N) running distances (1e10 on entry), idx (B,npoint); ties are
 *     resolved exactly as the reference's strided scan + tree reduction does -- among equal maxima the smallest
 *     bit-reversed (k mod T), then the smallest k, T = the reference's thread count for N (sampling_gpu.cu:16-142)
 *   btc_three_nn: dist2 / idx (N,3): squared distances and GLOBAL row indices of the three nearest known points of the
 *     same scene, first-found wins ties (interpolate_gpu.cu:14-69); btc_three_interpolate(_grad): :106-121,141-158
 * ---------------------------------------------------------------------------------------------- */
int btc_ball_query(const float* new_xyz, const int32_t* new_xyz_batch_cnt, const float* xyz, const int32_t* xyz_batch_cnt, int B,
                   int M, float inner_radius, float outer_radius, int nsample, int32_t* idx, void* stream);
int btc_group_points(const float* features, const int32_t* features_batch_cnt, const int32_t* idx, const int32_t* idx_batch_cnt,
                     int B, int M, int C, int nsample, float* out, void* stream);
int btc_group_points_grad(const float* grad_out, const int32_t* idx, const int32_t* idx_batch_cnt, const int32_t* features_batch_cnt,
                          int B, int M, int C, int N, int nsample, float* grad_features, void* stream);
int btc_furthest_point_sampling(const float* xyz, int B, int N, int npoint, float* temp, int32_t* idx, void* stream);
int btc_three_nn(const float* unknown, const int32_t* unknown_batch_cnt, const float* known, const int32_t* known_batch_cnt, int B,
                 int N, float* dist2, int32_t* idx, void* stream);
int btc_three_interpolate(const float* features, const int32_t* idx, const float* weight, int N, int C, float* out, void* stream);
int btc_three_interpolate_grad(const float* grad_out, const int32_t* idx, const float* weight, int N, int C, int M,
                               float* grad_features, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BTCDET_HIP_H */
