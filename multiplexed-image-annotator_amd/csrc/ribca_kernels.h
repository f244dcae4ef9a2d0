// Host-side launcher declarations shared by the translation units of libribca_hip.so.
// Every launcher enqueues on the given stream and returns immediately (no allocation, no sync).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ribca {

// ----- GEMM (gemm_split16.hip): C = A * W^T with A [M][2*Kp] and W [Np][2*Kp] in packed-split fp16 ---------------
int gemm_pick_bn(int N);            // column-tile width used for an N-wide weight (64 / 96 / 128)
int gemm_padded_n(int N);
int gemm_set_stamp_buffer(void* dev_ptr, long long capacity_blocks);   // diagnostics (variant 12): 20 x uint64 per workgroup; larger grids do not stamp
void gemm_set_variant(int v);      // 0 = production; 3/4/5/7/9 = A/B and timing-ablation forms of the same kernel           // N rounded up to that tile width (rows the packed weight must have)

struct GemmArgs {
  const uint16_t* A; int lda;       // activations, row stride in 16-bit elements (= 2*Kp)
  const uint16_t* W; int ldw;       // packed weight, Np rows
  int M, N, Kp;
  const float* bias;                // [N]
  const uint16_t* WF = nullptr;     // the same weight in fragment order (gemm_duo.hip, launch_pack_wf); nullptr = made on demand
};
// W [Np][2*Kp] packed-split -> fragment order for the two-workgroups-per-CU kernel: block (jt, s) of 2 KB = [hi | lo] x 64 lanes x 16 B
void launch_pack_wf(const uint16_t* W, int ldw, int Np, int Kp, uint16_t* WF, hipStream_t s);
// z[m][n] += acc + bias                                   (attn.proj, mlp.fc2)
void launch_gemm_resid(const GemmArgs& g, float* z, int ldz, hipStream_t s);
// out_ps[m][n] = gelu(acc + bias), packed-split          (mlp.fc1)
void launch_gemm_gelu(const GemmArgs& g, uint16_t* out, int ldo, hipStream_t s);
// ---- classifier blocks: LayerNorm folded into qkv / fc1, residual stream packed-split (gemm_epi.h: EpiResidPS, EpiGeluLn, EpiQKVLn)
// z_ps[m][n] = (z_ps[m][n] - prev[m * prev_stride].y) + acc + bias on the packed-split residual stream (prev = (rstd, mean) of the
// stored rows: re-centring, see EpiResidPS; nullptr = none); part [gemm_resid_tiles(N)][M] receives (mean, centred sum of squares)
// of every (row, column tile) of the NEW z, or nullptr
// Returns the geometry of `part` it used: with a fragment-order weight (g.WF) the GEMM runs on the two-workgroups-per-CU
// kernel with the residual tile riding the A ring (EpiResidZK, gemm_duo.hip) -- for the shapes where that measured faster, or for
// every shape it supports with force_duo -- and the statistics come per WAVE column block (16 / 32 / 48 columns) instead of per column
// tile; ln_finalize takes either.  RIBCA_RESID_DUO = 0 / 1 / 2: never / where it pays (default) / wherever supported.
struct ResidStatGeom { int tiles, bn; };
// an activation matrix in the three-plane MX3 format (described with the MX GEMM below)
struct MxAct { uint16_t* hi; unsigned char* l8; unsigned char* sc; int Kp; int M; };
// zmx != nullptr: the new rows are ALSO written in the MX3 format (the operand of the MX forms of the next qkv / fc1); needs N % 192 == 0
// and a fragment-order weight (the 128 x 192 tile of the two-workgroups form emits it), aborts otherwise
ResidStatGeom launch_gemm_resid_ps(const GemmArgs& g, uint16_t* z, int ldz, float2* part, const float2* prev, int prev_stride, hipStream_t s,
                                   bool force_duo = false, const MxAct* zmx = nullptr);
int gemm_resid_tiles(int N);        // column tiles of an N-wide residual GEMM on the one-workgroup kernel
int gemm_resid_bn(int N);           // their width
int gemm_resid_part_rows(int N);    // rows `part` must have for either kernel (the finer of the two geometries)
// the same as launch_gemm_gelu / launch_gemm_qkv with rowstat[m] = (rstd, mean):  x = rstd * acc + (-mean * rstd * csum[n] + bias[n])
void launch_gemm_gelu_ln(const GemmArgs& g, const float2* rowstat, const float* csum, uint16_t* out, int ldo, hipStream_t s);
// geometry of one attention problem: D = H*hd features, T tokens per cell; Q/K rows padded to TP = 16*NT tokens and STORED with
// hdq = round8(hd) dims (compact: whole PS groups only); the MFMA K dimension is hdp = round32(hd), the groups beyond hdq are
// zero registers, never memory.  V^T rows = hdv head dims x KP keys (KP = 32*ceil(NT/2)); rows >= hd are never read.
struct AttnGeom {
  int D, H, hd, hdp, hdv, T, NT, TP, KP, hdq;
};
AttnGeom make_attn_geom(int D, int H, int T);
bool attention_supported(const AttnGeom& a);
// true: the attention kernel for this geometry reads V ROW-MAJOR, [cell][head][TP][2 * hdq] like K (the 101-token classifiers: LDS-staged
// kernel with transposing LDS reads); false: V^T [cell][head][hdv][2 * KP] in fragment order (the imputer's <= 16-token problems)
bool attention_v_rowmajor(const AttnGeom& a);
size_t attention_v_elems(const AttnGeom& a, int cells);   // 16-bit elements of the V operand buffer for `cells` cells
// qkv: scatter into per-head attention operands (Q pre-scaled by hd^-0.5, V transposed + key-permuted)
void launch_gemm_qkv(const GemmArgs& g, uint16_t* q, uint16_t* k, uint16_t* vt, const AttnGeom& a, float scale, hipStream_t s);
// fp32 output through a per-cell row map (imputer embeddings / predictions), see EpiRowMap
// n_off / cls_rows / rs_stride: a window of the full product (EpiQKVT): g.W, g.bias and csum already point at output column n_off,
// g.N columns are computed; cls_rows = 1: GEMM row m is token 0 of cell m (g.A addressed with a row stride of T rows) and its
// statistics are rowstat[m * rs_stride]
void launch_gemm_qkv_ln(const GemmArgs& g, const float2* rowstat, const float* csum, uint16_t* q, uint16_t* k, uint16_t* vt, const AttnGeom& a,
                        float scale, hipStream_t s, int n_off = 0, int cls_rows = 0, int rs_stride = 1);
void launch_gemm_rowmap(const GemmArgs& g, float* out, int ldo, const float* add, int ldadd, const int* slot, const int* addrow, int R,
                        int dst_per_cell, hipStream_t s);

// ----- MX GEMM (gemm_mx.hip): fp16 hi * hi + two block-scaled correction products; activations in the three-plane "MX3" format
// (hi fp16 permuted inside every 128 columns, lo as e4m3 bytes, one E8M0 scale byte per 32 columns -- see gemm_mx.hip), Kp % 128 == 0.
// The scale plane is TRANSPOSED, [Kp / 128][M][4]: the 128 rows x 4 bytes a K step needs are 512 contiguous bytes (four lines), not 128
// pieces of four bytes one row pitch apart.
struct MxWeight { const uint16_t* wh; const unsigned char* wx; };     // mx_pack_w image of a packed-split weight [Np][2 Kp]
size_t mx_wh_bytes(int Np, int Kp);
size_t mx_wx_bytes(int Np, int Kp);
inline size_t mx_act_hi_elems(size_t M, int Kp128) { return M * (size_t)Kp128; }          // fp16 elements; l8: the same count of bytes; sc: / 32
// W: packed-split [Np][ldw] with Ksrc columns (a multiple of 32); Kp = Ksrc rounded up to 128 (the pad is written as zeros)
void launch_mx_pack_w(const uint16_t* W, int ldw, int Np, int Ksrc, int Kp, uint16_t* WH, unsigned char* WX, hipStream_t s);
// packed-split rows [M][2 Kp] -> MX3 (a.Kp = Kp rounded up to 128): the reference producer of the format
void launch_mx_pack_act(const uint16_t* ps, int ldps, int M, int Kp, const MxAct& a, hipStream_t s);
bool gemm_mx_supported(int N, int Kp);
// out (MX3, out.Kp == g.N) = gelu(rstd acc + (-mean rstd csum + bias)): mlp.fc1 in front of an MX fc2 (gemm_duo.hip); false = not launched
bool launch_gemm_gelu_mx(const GemmArgs& g, const float2* rowstat, const float* csum, const MxAct& out, hipStream_t s);
// z_ps = (z_ps - prev mean) + A W^T + bias, statistics per 48-column wave block (as launch_gemm_resid_ps on the duo kernel)
// zmx: as for launch_gemm_resid_ps (N % 192 == 0)
ResidStatGeom launch_gemm_mx_resid(const MxAct& A, const MxWeight& W, int M, int N, const float* bias, uint16_t* z, int ldz, float2* part,
                                   const float2* prev, int prev_stride, hipStream_t s, int abl = 0, const MxAct* zmx = nullptr);
// attn.qkv / mlp.fc1 with the LayerNorm fold on the MX kernel: A = the residual rows in MX3 (N % 192 == 0); the fc1 form writes its GELU
// output in MX3 (out.Kp == N).  Arguments as launch_gemm_qkv_ln / launch_gemm_gelu_mx.
void launch_gemm_mx_qkv_ln(const MxAct& A, const MxWeight& W, int M, int N, const float* bias, const float2* rowstat, const float* csum, uint16_t* q,
                           uint16_t* k, uint16_t* vt, const AttnGeom& a, float scale, hipStream_t s);
void launch_gemm_mx_gelu(const MxAct& A, const MxWeight& W, int M, int N, const float* bias, const float2* rowstat, const float* csum, const MxAct& out,
                         hipStream_t s);

// ----- attention (attention.hip) ---------------------------------------------------------------------------
// q,k: [cells][H][TP][2*hdq]  vt: V operand, layout per attention_v_rowmajor()  out: packed-split [cells*T][ldo]
// q_tiles > 0 restricts the QUERY rows to the first q_tiles 16-token tiles (keys/values are always complete)
void launch_attention(const uint16_t* q, const uint16_t* k, const uint16_t* vt, uint16_t* out, int ldo, int cells, const AttnGeom& a,
                      hipStream_t s, int q_tiles = 0);

// ----- per-cell fusion norm1 -> qkv -> attention (cell_attention.hip): D = 144 / 288 / 384 (head dims 12 / 24 / 32), 101 tokens, 12 heads.
// z: packed-split residual rows, W / bias2 / csum: the folded qkv weight (row-major packed-split) and its vectors, rowstat: (rstd, mean)
// per row; out: packed-split attention output rows.  Replaces launch_gemm_qkv_ln + launch_attention for whole blocks.
bool cell_attention_supported(int D, int H, int T);
void launch_cell_qkv_attention(const uint16_t* z, int ldz, const uint16_t* W, int ldw, const float* bias2, const float* csum, const float2* rowstat,
                               uint16_t* out, int ldo, int cells, int D, float scale, hipStream_t s);

// ----- small ViT kernels (vit_misc.hip) ----------------------------------------------------------------------
void launch_layernorm_ps(const float* z, int ldz, const float* gamma, const float* beta, uint16_t* out, int ldo, int M, int D,
                         hipStream_t s);
void launch_embed_f32(const float* patches, int c_img, const int* src_chan, int C, const float* w, const float* bias, const float* pos,
                      float* z, int ldz, int D, int cells, hipStream_t s);
void launch_cls_rows(float* z, int ldz, const float* cls, const float* pos, int D, int cells, int tokens_per_cell, hipStream_t s);
// out_ps row (cell*S + j) = LayerNorm(z row (cell*T + sel[j])): the rows a following GEMM actually needs
void launch_layernorm_gather_ps(const float* z, int ldz, const float* gamma, const float* beta, uint16_t* out, int ldo, int cells, int T,
                                int S, const int* sel, int D, hipStream_t s);
// the same from a packed-split residual stream (the imputer's folded blocks): z_ps rows [cells*T][ldz] of 16-bit elements, D <= 768
void launch_layernorm_gather_ps_from_ps(const uint16_t* z_ps, int ldz, const float* gamma, const float* beta, uint16_t* out, int ldo, int cells,
                                        int T, int S, const int* sel, int D, hipStream_t s);
// out_ps row (cell*S + j) = split(src fp32 row (cell*T + sel[j])) of K values, zero padded to Kp
void launch_rows_to_ps(const float* src, int K, uint16_t* out, int ldo, int Kp, int cells, int T, int S, const int* sel, hipStream_t s);
// z row (cell*T + sel[j]) = a[:] + table[sel[j]][:]   (mask tokens + positional embedding)
void launch_fill_rows(float* z, int ldz, const float* a, const float* table, int D, int cells, int T, int S, const int* sel, hipStream_t s);
void launch_head_softmax(const float* z, int ldz, const float* gamma, const float* beta, const float* hw, const float* hb,
                         float* probs, int D, int K, int cells, hipStream_t s);
void launch_pack_weight(const float* w, int N, int K, uint16_t* out, int Np, int Kp, hipStream_t s);
// packed-split residual stream of the classifiers
void launch_embed_ps(const float* patches, int c_img, const int* src_chan, int C, const float* w, const float* bias, const float* pos,
                     uint16_t* z, int ldz, int D, int cells, hipStream_t s);
void launch_cls_rows_ps(uint16_t* z, int ldz, const float* cls, const float* pos, int D, int cells, int tokens_per_cell, hipStream_t s);
void launch_head_softmax_ps(const uint16_t* z, int ldz, const float* gamma, const float* beta, const float* hw, const float* hb,
                            float* probs, int D, int K, int cells, hipStream_t s);
// rowstat[m] = (rstd, mean) of row m of a packed-split z (LayerNorm eps 1e-6, biased variance); recentre: the row is first rewritten
// as z - mean (every reader is a LayerNorm: unobservable) and the statistics describe the rewritten row
void launch_row_stats_ps(uint16_t* z, int ldz, int M, int D, float2* rowstat, bool recentre, hipStream_t s);
// the same from the residual epilogue's per-tile pairs: part [T][M], tiles bn wide over N columns
void launch_ln_finalize(const float2* part, int T, int M, int bn, int N, float2* rowstat, hipStream_t s);
// gamma o W packed-split + csum[n] + bias2[n] = bias[n] + W[n] . beta   (LayerNorm folded into the Linear)
void launch_pack_weight_fold(const float* w, int N, int K, const float* gamma, const float* beta, const float* bias, uint16_t* out, int Np,
                             int Kp, float* csum, float* bias2, hipStream_t s);

// ----- pre-processing (preprocess.hip) -----------------------------------------------------------------------
void launch_mask_max(const int32_t* mask, long long n, int32_t* out_max, hipStream_t s);
void launch_label_table_init(int32_t* tab_i32, unsigned long long* tab_u64, int L, hipStream_t s);
void launch_label_table(const int32_t* mask, int H, int W, int L, int32_t* tab_i32, unsigned long long* tab_u64, hipStream_t s);
void launch_channel_min(const float* img, int C, long long hw, float* out_min, hipStream_t s);
struct PatchArgs {
  const float* img; int C, H, W;     // normalised image (C,H,W) fp32
  const int32_t* mask;               // (H,W)
  const float* chan_min;             // (C)  per-channel minimum (reference _move_image_range)
  const int32_t* cell_id;            // (n)
  const int32_t* bbox;               // (n,4) rmin,rmax,cmin,cmax
  const double* taps;                // Gaussian taps sigma=1,2,3: [9 + 17 + 25] doubles, centre-out order documented in .hip
  float* patches;                    // (n, C, 40, 40)
  double* avg_int;                   // (n, C) or nullptr
  int n;
};
void launch_extract_patches(const PatchArgs& a, hipStream_t s);
// cell_size != 30: window ps x ps, anti-alias taps (aa_radius + 1 doubles, 0 = no filter), 40 nearest-neighbour source indices.
// Returns non-zero if ps is outside what one workgroup can hold (ps*ps <= 8192).
int launch_extract_patches_scaled(const PatchArgs& a, int ps, const double* aa_taps, int aa_radius, const int32_t* src_index, hipStream_t s);

// ----- label painting (colorize.hip) ----------------------------------------------------------------------------
void launch_colorize(const int32_t* mask, long long npx, const int32_t* label_to_cell, int L, const uint8_t* type_rgb, const uint8_t* conf_rgb,
                     const uint8_t* type_idx, uint8_t* out_type, uint8_t* out_conf, uint8_t* out_idx, hipStream_t s);

// ----- k nearest neighbours + cell-type co-occurrence (knn.hip); non-zero return = unsupported k / T
int launch_knn_cooccurrence(const double* x, const double* y, const int32_t* type, int n, int k, int T, unsigned long long* matrix,
                            hipStream_t s);

// per cell: counts of each type among its nearest list[l] other cells (k = list[last] + 1 <= 256 neighbours incl. itself)
int launch_knn_compositions(const double* x, const double* y, const int32_t* type, int n, int k, int T, int n_lists, const int* list_dev,
                            uint16_t* counts, hipStream_t s);

// ----- whole-image normalisation (normalize.hip) ---------------------------------------------------------------
void launch_u16_to_f32(const uint16_t* in, float* out, long long n, hipStream_t s);
void launch_gauss1d(const float* in, float* out, int planes, int H, int W, int axis, const double* w, int R, int mode, hipStream_t s);
void launch_bg_subtract(float* x, const float* bg, long long n, float cap, hipStream_t s);
void launch_plane_max(const float* x, int planes, long long hw, float* out, hipStream_t s);
void launch_radix_hist(const float* x, int planes, long long hw, const unsigned int* prefix, unsigned int mask_hi, int shift, int bits,
                       unsigned int* hist, hipStream_t s);
void launch_norm_finalize(float* x, int planes, long long hw, const int* mode, const float* clip, const float* denom, hipStream_t s);

// ----- vote (vote.hip) -----------------------------------------------------------------------------------------
struct VoteArgs {
  const float* p_a; int k_a; const int8_t* map_a;   // first model: probs (n,k_a), class -> global id (17 = Others)
  const float* p_b; int k_b; const int8_t* map_b;   // second model or nullptr
  const float* type_conf;                           // [18] per-type thresholds (global id order), negative = unset
  float conf;                                       // global confidence threshold
  int n;
  int8_t* label;                                    // out: global class id
  float* out_conf;                                  // out: confidence, -1 when thresholded to Others
};
void launch_vote(const VoteArgs& a, hipStream_t s);

}  // namespace ribca
