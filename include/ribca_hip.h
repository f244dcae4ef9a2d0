/* libribca_hip.so -- C ABI of the MI355X-native RIBCA hot path (gfx950 only).
 *
 * The reference (sun-huangqingbo/multiplexed-image-annotator) is pure Python and has no FFI: its compute boundary is a
 * set of Python methods.  Each entry point below replaces the body of one of them; the Python host in
 * multiplexed-image-annotator_amd/ binds these symbols with ctypes and keeps the reference's class/method surface
 * (INTEGRATION.md shows the stub a maintainer would add to the reference).  Paths cited are relative to
 * src/multiplexed_image_annotator/cell_type_annotation/ of the reference.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named *_host; sizes are element counts unless named *_bytes;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work and return;
 *   - the caller owns all memory; only ribca_vit_create allocates (the packed-weight handle);
 *   - return value 0 = ok, non-zero = error, text via ribca_last_error() (thread-local).  A request the library has no kernel for (a shape,
 *     a geometry, an attribute the runtime refuses) is such an error: no entry point ends the calling process, whatever it is handed
 *     (the reference's convention for a bad request is an exception the caller can catch: model.py:636, 770).
 *   - the kernel-level hooks tests/ and tools/ drive are NOT part of this ABI: include/ribca_hip_test.h, libribca_hip_test.so.
 */
#ifndef RIBCA_HIP_H
#define RIBCA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: the declarations between this push and the pop are its whole dynamic symbol table */
#pragma GCC visibility push(default)

typedef struct ribca_vit ribca_vit_t;
typedef struct ribca_mae ribca_mae_t;

int ribca_version(void);
const char* ribca_last_error(void);

/* ---- pre-processing ------------------------------------------------------------------------------------------ */

/* out2[0] = max(mask), out2[1] = min(mask).  Sizes the label table (preprocess.py:159-181 builds a dict instead). */
int ribca_mask_minmax(const int32_t* mask, int64_t n, int32_t* out2, void* stream);

/* Replaces ImageProcessor._cell_pos_dict (preprocess.py:159-211) for everything the hot path reads from it.
 * L = max label + 1.  tab_i32 = [5][L]: row min, row max, col min, col max, pixel count (count 0 = label absent);
 * tab_u64 = [2][L]: sum of rows, sum of cols.  Label 0 (background) is skipped; labels must be in [1, L). */
int ribca_label_table(const int32_t* mask, int32_t H, int32_t W, int32_t L, int32_t* tab_i32, uint64_t* tab_u64, void* stream);

/* Replaces ImageProcessor._move_image_range (preprocess.py:153-157): per-channel minimum of an fp32 (C, H*W) image. */
int ribca_channel_min(const float* image, int32_t C, int64_t hw, float* out_min, void* stream);

/* Replaces utils.crop_cell + utils.smooth (utils.py:226-270) for patch_size 40 (cell_size 30): for each of n cells writes
 * the soft-masked fp32 patch of EVERY image channel, (n, C, 40, 40), bit-identical to the reference arithmetic, and
 * optionally avg (n, C) fp64 = mean over labelled pixels of the window (reference avg_int, before the (x+1)/2 of
 * preprocess.py:145-149).  bbox = (n, 4) int32 rmin, rmax, cmin, cmax from the label table.  taps = 27 fp64 Gaussian
 * weights: sigma 1 -> taps[0..4], sigma 2 -> taps[5..13], sigma 3 -> taps[14..26], entry k = weight at distance k
 * (host computes them exactly as scipy.ndimage._gaussian_kernel1d does). */
int ribca_extract_patches(const float* image, int32_t C, int32_t H, int32_t W, const int32_t* mask, const float* chan_min,
                          const int32_t* cell_id, const int32_t* bbox, const double* taps, int32_t n, float* patches, double* avg,
                          void* stream);

/* Same for cell_size != 30 (preprocess.py:78,106): window patch_size = int(40 * cell_size / 30), then
 * skimage.transform.resize(patch, (C, 40, 40), order=0, anti_aliasing=True, preserve_range=True): fp64 Gaussian pre-filter
 * (aa_taps[0..aa_radius], weight at distance k, sigma = (patch_size/40 - 1)/2; aa_radius = 0 when patch_size <= 40) in
 * 'mirror' mode, nearest-neighbour grid sampling at src_index[0..39] (host: floor(((o + 0.5) * patch_size/40 - 0.5) + 0.5) in
 * fp64, as scipy.ndimage.zoom(order=0, grid_mode=True)), one rounding to fp32.  avg is taken over the patch_size window
 * before the resize, as crop_cell does.  4 <= patch_size <= 90. */
int ribca_extract_patches_scaled(const float* image, int32_t C, int32_t H, int32_t W, const int32_t* mask, const float* chan_min,
                                 const int32_t* cell_id, const int32_t* bbox, const double* taps, int32_t n, int32_t patch_size,
                                 const double* aa_taps, int32_t aa_radius, const int32_t* src_index, float* patches, double* avg,
                                 void* stream);

/* ---- whole-image normalisation primitives (ImageProcessor._normalize, preprocess.py:214-239) ---------------------
 * The host drives them per image (ops.normalize_image in the Python package shows the sequence): the percentile needs
 * a data-dependent host decision, everything touching pixels runs here.  Results are bit-identical to the reference. */
int ribca_u16_to_f32(const uint16_t* in, float* out, int64_t n, void* stream);
/* One axis of scipy.ndimage.gaussian_filter on `planes` fp32 (H, W) images: fp64 accumulation in scipy's order, fp32 result.
 * taps[k] = weight at distance k (k = 0..R); mode 0 = 'reflect', 1 = 'nearest'; axis 0 = rows, 1 = columns; in != out. */
int ribca_gauss1d(const float* in, float* out, int32_t planes, int32_t H, int32_t W, int32_t axis, const double* taps, int32_t R,
                  int32_t mode, void* stream);
/* x = max(x - min(bg, cap), 0) (preprocess.py:219-222) */
int ribca_bg_subtract(float* x, const float* bg, int64_t n, float cap, void* stream);
/* out[p] = max over plane p of non-negative fp32 data */
int ribca_plane_max(const float* x, int32_t planes, int64_t hw, float* out, void* stream);
/* One radix-select pass over non-negative fp32 keys: hist[p][b] = #{k in plane p : (k & mask_hi) == prefix[p],
 * (k >> shift) & (2^bits - 1) == b}; hist is (planes, 2048) uint32.  Three passes (11 + 11 + 10 bits) locate any order
 * statistic exactly; np.percentile's interpolation between two of them is done by the host. */
int ribca_radix_hist(const float* x, int32_t planes, int64_t hw, const uint32_t* prefix, uint32_t mask_hi, int32_t shift, int32_t bits,
                     uint32_t* hist, void* stream);
/* per plane: mode 0 -> fill -1; else x = 2 * (min(x, clip) / denom) - 1 (preprocess.py:229-238) */
int ribca_norm_finalize(float* x, int32_t planes, int64_t hw, const int32_t* mode, const float* clip, const float* denom, void* stream);

/* ---- ViT classifier (timm VisionTransformer subclass, model.py:31-88) -------------------------------------- */

/* Number of fp32 values in the flat parameter blob ribca_vit_create expects, in this order:
 *   cls_token[D], pos_embed[101*D], patch_embed.proj.weight[D*C*16], patch_embed.proj.bias[D],
 *   per block: norm1.weight[D], norm1.bias[D], attn.qkv.weight[3D*D], attn.qkv.bias[3D], attn.proj.weight[D*D],
 *              attn.proj.bias[D], norm2.weight[D], norm2.bias[D], mlp.fc1.weight[4D*D], mlp.fc1.bias[4D],
 *              mlp.fc2.weight[D*4D], mlp.fc2.bias[D],
 *   norm.weight[D], norm.bias[D], head.weight[K*D], head.bias[K]
 * (the state-dict key order of the checkpoints Annotator.load_models reads, model.py:188-239). */
/* 1 if classifiers of width D run mlp.fc1 -> mlp.fc2 as the MX pair (csrc/gemm_mx.hip: fp16 hi * hi + two block-scaled corrections, 1.75
 * matrix units per product and 3 bytes per element of h instead of 3 passes / 4 bytes): 4 D % 128 == 0, D % 48 == 0 and RIBCA_MX != 0 */
int32_t ribca_mx_enabled(int32_t D);
/* 1 if, beyond that, the residual rows of a classifier of width D are ALSO kept in the MX3 format and attn.qkv (where it is a GEMM of its own)
 * and mlp.fc1 run on the MX kernel too: ribca_mx_enabled(D), D % 192 == 0 and RIBCA_MXZ != 0 */
int32_t ribca_mxz_enabled(int32_t D);
int64_t ribca_vit_blob_len(int32_t D, int32_t C, int32_t K, int32_t depth);

/* Replaces Annotator.load_models for one model: repacks the fp32 parameters (device blob) into the MFMA layouts.
 * D % 48 == 0 (12 heads, head dim % 4 == 0), D <= 768, K <= 16. */
int ribca_vit_create(const float* blob, int64_t blob_len, int32_t D, int32_t C, int32_t K, int32_t depth, void* stream,
                     ribca_vit_t** out);
void ribca_vit_destroy(ribca_vit_t* m);

/* Scratch bytes ribca_vit_forward needs to process `chunk_cells` cells at a time. */
int64_t ribca_vit_workspace_bytes(const ribca_vit_t* m, int32_t chunk_cells);

/* Replaces the body of Annotator._predict_cell_types' inner loop (model.py:397-406): probs = softmax(model(x), dim=1).
 * patches: (n_cells, c_img, 40, 40) fp32 full-channel patches from ribca_extract_patches (or the imputer);
 * src_chan: (C) int32, image channel feeding each model channel, -1 = blank plane of -1.0 (preprocess.py:110-120);
 * probs: (n_cells, K) fp32.  Cells are processed in chunks of chunk_cells through `workspace`. */
int ribca_vit_forward(const ribca_vit_t* m, const float* patches, int32_t c_img, const int32_t* src_chan, int32_t n_cells,
                      float* probs, void* workspace, int64_t workspace_bytes, int32_t chunk_cells, void* stream);
/* The same forward with every product as three fp16 passes (the 22-bit operands of round 3), whatever ribca_mx_enabled says: what
 * Annotator.predict re-evaluates the few cells with whose fast result lies within the MX arithmetic's error of a decision boundary
 * (top-2 margin, confidence thresholds), so that labels are those of the full-precision path.  Same arguments, same workspace. */
int ribca_vit_forward_precise(const ribca_vit_t* m, const float* patches, int32_t c_img, const int32_t* src_chan, int32_t n_cells,
                      float* probs, void* workspace, int64_t workspace_bytes, int32_t chunk_cells, void* stream);

/* Algorithmic FLOPs per cell of this model (BASELINE.md section 3 formula). */
double ribca_vit_flops_per_cell(const ribca_vit_t* m);

/* ---- marker imputer (MarkerImputer / MaskedAutoencoderViT, markerImputer.py:69-329) --------------------------------
 * Tokens are the panel's L channels (each 40x40 plane = 1600 pixels); encoder 768 wide / 12 heads over the present
 * channels + CLS, decoder 512 / 8 heads over all L + CLS, linear prediction of the missing planes.
 * Blob order (state-dict keys of the *_impute.pth checkpoints): cls_token[768], pos_embed[(L+1)*768],
 * patch_embed.proj.weight[768*1600], patch_embed.proj.bias[768], blocks.* (same 12 tensors per block as the classifier),
 * norm.weight, norm.bias, decoder_embed.weight[512*768], decoder_embed.bias[512], mask_token[512],
 * decoder_pos_embed[(L+1)*512], decoder_blocks.*, decoder_norm.weight, decoder_norm.bias,
 * decoder_pred.weight[1600*512], decoder_pred.bias[1600]. */
int64_t ribca_mae_blob_len(int32_t L, int32_t enc_depth, int32_t dec_depth);
int ribca_mae_create(const float* blob, int64_t blob_len, int32_t L, int32_t enc_depth, int32_t dec_depth, void* stream,
                     ribca_mae_t** out);
void ribca_mae_destroy(ribca_mae_t* m);
int64_t ribca_mae_workspace_bytes(const ribca_mae_t* m, int32_t chunk_cells, int32_t n_present);
/* Replaces MarkerImputer.impute (markerImputer.py:294-329): patches (n_cells, L, 40, 40) fp32 in place -- every channel
 * position NOT listed in present_host (HOST array, strictly increasing, n_present entries) is overwritten by the
 * prediction; listed channels are left bit-for-bit untouched.  Synchronises the stream once (index tables upload). */
int ribca_mae_impute(const ribca_mae_t* m, float* patches, const int32_t* present_host, int32_t n_present, int32_t n_cells,
                     void* workspace, int64_t workspace_bytes, int32_t chunk_cells, void* stream);

/* ---- label painting (Annotator.colorize, model.py:806-858, without tissue regions) ---------------------------------
 * mask (n_pixels) int32; label_to_cell (L) int32: row of the label in the per-cell arrays or -1; per-cell colours
 * cell_type_rgb / cell_conf_rgb (n_cells, 3) uint8 and cell_type_idx (n_cells) uint8 (= cell-type index + 1).
 * Outputs: (n_pixels, 3), (n_pixels, 3), (n_pixels) uint8; background and unknown labels are 0. */
int ribca_colorize(const int32_t* mask, int64_t n_pixels, const int32_t* label_to_cell, int32_t L, const uint8_t* cell_type_rgb,
                   const uint8_t* cell_conf_rgb, const uint8_t* cell_type_idx, uint8_t* out_type_rgb, uint8_t* out_conf_rgb,
                   uint8_t* out_type_idx, void* stream);

/* ---- neighbourhood analysis (spatial_methods.neighborhood_analysis, spatial_methods.py:13-130) ------------------------
 * x, y (n_cells) fp64 cell centroids (mean column, mean row), cell_type (n_cells) int32 in [0, n_types).  For every cell the
 * n_neighbors nearest cells in fp64 (itself first; ties towards the lower index -- the reference's ball tree leaves ties
 * unspecified) and matrix[type(cell)][type(neighbour)] += 1 for the other n_neighbors - 1.  matrix (n_types, n_types) uint64 is
 * ACCUMULATED into (zero it first; call once per image for the integrated mode).  n_neighbors <= 32, n_types <= 32. */
int ribca_knn_cooccurrence(const double* x, const double* y, const int32_t* cell_type, int32_t n_cells, int32_t n_neighbors, int32_t n_types,
                           uint64_t* matrix, void* stream);

/* Neighbourhood compositions of spatial_methods.tissue_region_partition (spatial_methods.py:133-180): sizes (DEVICE array, n_sizes <= 8,
 * strictly increasing, max <= 255; the reference uses 10,20,30,50,75,100,150,200) -> counts (n_cells, n_sizes, n_types) uint16 =
 * number of cells of each type among the nearest sizes[l] OTHER cells (fp64 distances, ties towards the lower index).  The
 * reference divides each row by its sum and feeds PCA + KMeans on the host.  Synchronises the stream once (reads sizes). */
int ribca_knn_compositions(const double* x, const double* y, const int32_t* cell_type, int32_t n_cells, int32_t n_types, const int32_t* sizes,
                           int32_t n_sizes, uint16_t* counts, void* stream);

/* ---- vote (Annotator.merge_by_voting, model.py:481-633) ------------------------------------------------------ */
/* Global class ids: 0..16 = key order of utils.get_void_vote (utils.py:143-146), 17 = "Others".
 * p_a (n, k_a) and optional p_b (n, k_b) are softmax outputs; map_* (k) int8 give each class's global id;
 * type_conf (18) fp32 per-type thresholds (negative = unset); label (n) int8 and conf (n) fp32 (-1 = thresholded). */
int ribca_vote(const float* p_a, int32_t k_a, const int8_t* map_a, const float* p_b, int32_t k_b, const int8_t* map_b,
               const float* type_conf, float conf, int32_t n, int8_t* label, float* out_conf, void* stream);

/* ---- profiling (kernel-level test hooks: include/ribca_hip_test.h, a library of their own) ------------------------- */
/* When enabled, every kernel launch of the ViT forward is bracketed by HIP events on its stream; ribca_prof_read
 * synchronises and returns, per kernel class, total milliseconds and launch count since the last reset.
 * classes: 0 gemm_qkv 1 gemm_proj 2 gemm_fc1 3 gemm_fc2 4 gemm_embed 5 attention 6 layernorm (row statistics) 7 cell_qkv_attention
 * (the per-cell fused norm1 -> qkv -> attention kernel of D <= 384) 8 head 9 other */
int ribca_prof_enable(int32_t on);
int ribca_prof_read(double* ms_out10, int64_t* count_out10);
const char* ribca_prof_name(int32_t cls);

/* NOT part of the stable ABI.  The versioned table of host launchers that libribca_hip_test.so (the kernel-level hooks of tests/ and tools/,
 * include/ribca_hip_test.h) binds instead of C++ symbols: csrc/ribca_internal.h describes it and belongs to one build.  NULL for any other
 * version than that build's RIBCA_INTERNAL_VERSION.  A caller of the product ABI never needs it. */
const void* ribca_internal_table(int32_t version);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* RIBCA_HIP_H */
