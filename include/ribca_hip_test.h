/* libribca_hip_test.so -- kernel-level hooks of tests/ and tools/ for the library of include/ribca_hip.h.
 *
 * NOT part of the product ABI: libribca_hip.so exports none of these.  The hooks live in a library of their own that is linked against
 * libribca_hip.so and calls the same host launchers -- hence the same kernels -- as ribca_vit_forward / ribca_mae_impute; they exist to
 * localise a mismatch to one kernel (tests/test_gpu_kernels.py, tests/test_gpu_mx.py) and to time one kernel (tools/).  Conventions as in
 * ribca_hip.h: device pointers, a hipStream_t as void*, 0 = ok, error text via ribca_last_error() of libribca_hip.so.
 */
#ifndef RIBCA_HIP_TEST_H
#define RIBCA_HIP_TEST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* built with -fvisibility=hidden: these hooks are the library's whole dynamic symbol table */

/* Kernel-level hooks used by tests/ to localise a mismatch (same kernels the forward launches). */
int ribca_test_pack_weight(const float* w, int32_t N, int32_t K, uint16_t* out, int32_t Np, int32_t Kp, void* stream);
int ribca_test_layernorm(const float* z, int32_t ldz, const float* gamma, const float* beta, uint16_t* out, int32_t ldo, int32_t M,
                         int32_t D, void* stream);
int ribca_test_gemm(int32_t kind, const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp,
                    const float* bias, void* out, int32_t ldo, void* stream); /* kind 0: z += ..., 1: gelu -> PS */
int ribca_test_qkv_attention(const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t cells, int32_t D, int32_t Kp,
                             const float* bias, uint16_t* q, uint16_t* k, uint16_t* vt, uint16_t* out, int32_t ldo, void* stream);
/* The classifiers' blocks run with LayerNorm FOLDED into the Linear behind it (timm Block.norm1 -> attn.qkv, norm2 -> mlp.fc1,
 * reached from model.py:54-55): the residual stream is packed-split fp16 hi + lo, z_ps [M][ldz], which the qkv / fc1 GEMM reads as
 * its operand; the weight is gamma o W, and the epilogue applies x = rstd * acc + (-mean * rstd) * csum[n] + bias2[n] with the
 * (rstd, mean) pair of the row in `rowstat`.  Hooks:
 *   fold_weight:   w [N][K] fp32 + gamma, beta [K] + bias [N] -> packed weight [Np][2*Kp], csum [N], bias2 [N]
 *   row_stats:     rowstat [M] float2 = (rstd, mean) of every row of z_ps (eps 1e-6, biased variance over D); recentre != 0: the
 *                  rows are first rewritten as z - mean (every reader of the stream is a LayerNorm, so a per-row constant is
 *                  unobservable) and the statistics are those of the rewritten rows
 *   gemm_resid_ps: z_ps = (z_ps - prev[m].mean) + A W^T + bias in place (prev: rowstat of the stored rows, or NULL); rowstat
 *                  (optional) = statistics of the NEW rows through the epilogue's per-tile pairs (part: scratch of
 *                  ribca_test_resid_tiles(N) * M float2)
 *   gemm_fold:     kind 1: out_ps = gelu(folded x), as ribca_test_gemm kind 1
 *   qkv_attention_fold: as ribca_test_qkv_attention with the folded qkv epilogue */
int ribca_test_fold_weight(const float* w, int32_t N, int32_t K, const float* gamma, const float* beta, const float* bias, uint16_t* out,
                           int32_t Np, int32_t Kp, float* csum, float* bias2, void* stream);
int ribca_test_row_stats(uint16_t* z_ps, int32_t ldz, int32_t M, int32_t D, float* rowstat, int32_t recentre, void* stream);
int32_t ribca_test_resid_tiles(int32_t N);
int ribca_test_gemm_resid_ps(const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp,
                             const float* bias, uint16_t* z_ps, int32_t ldz, float* part, float* rowstat, const float* prev, void* stream);
/* gemm_resid_ps as the classifiers' full blocks run it for the weight shapes where it measured faster (any M): two workgroups per CU (gemm_duo.hip), the weight in
 * MFMA fragment order (wf_scratch: Np * 2 * Kp uint16, filled here from W), the residual tile through the A ring as extra K steps against
 * an identity fragment, a load-free epilogue; statistics per wave column block (part: ribca_test_resid_part_rows(N) * M float2). */
int32_t ribca_test_resid_part_rows(int32_t N);
int ribca_test_gemm_resid_ps_duo(const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp,
                                 const float* bias, uint16_t* wf_scratch, uint16_t* z_ps, int32_t ldz, float* part, float* rowstat,
                                 const float* prev, void* stream);
/* The MX form of the residual GEMM (csrc/gemm_mx.hip; replaces timm Mlp.fc2 reached from reference model.py:54-55): A given as
 * packed-split rows is first converted to the three-plane MX3 format (hi_out [M][Kp128] fp16, l8_out [M][Kp128] bytes, sc_out
 * [Kp128 / 128][M][4] bytes, Kp128 = Kp rounded up to 128; all three are outputs the tests inspect), W to the MX weight image
 * (wh_scratch / wx_scratch: ribca_test_mx_weight_bytes), then z_ps = (z_ps - prev mean) + A W^T + bias with statistics per 48-column
 * block (part: (N / 48) * M float2).  N % 48 == 0. */
int64_t ribca_test_mx_weight_bytes(int32_t N, int32_t Kp, int32_t which);
int ribca_test_mx_pack_act(const uint16_t* A, int32_t lda, int32_t M, int32_t Kp, uint16_t* hi_out, uint8_t* l8_out, uint8_t* sc_out, void* stream);
int ribca_test_gemm_mx_resid(const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp, const float* bias,
                             uint16_t* hi_out, uint8_t* l8_out, uint8_t* sc_out, uint16_t* wh_scratch, uint8_t* wx_scratch, uint16_t* z_ps,
                             int32_t ldz, float* part, float* rowstat, const float* prev, void* stream);
/* the same GEMM on operands that are already in the MX formats (what the forward does; tools/bench_mx.py times it) */
int ribca_test_gemm_mx_resid_packed(const uint16_t* hi, const uint8_t* l8, const uint8_t* sc, int32_t Kp, const uint16_t* wh, const uint8_t* wx,
                                    int32_t M, int32_t N, const float* bias, uint16_t* z_ps, int32_t ldz, float* part, float* rowstat,
                                    const float* prev, void* stream);
/* mlp.fc1 with the LayerNorm fold writing its GELU output straight in the MX3 format (csrc/gemm_duo.hip, EpiGeluMx): N % 128 == 0,
 * hi_out [M][N] fp16 (permuted inside every 128 columns), l8_out [M][N] bytes, sc_out [N / 128][M][4] bytes */
int ribca_test_gemm_gelu_mx(const uint16_t* z_ps, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp, const float* bias2,
                            const float* csum, const float* rowstat, uint16_t* wf_scratch, uint16_t* hi_out, uint8_t* l8_out, uint8_t* sc_out,
                            void* stream);
/* mlp.fc1 on the MX kernel (csrc/gemm_mx.hip, EpiGeluMx with 48-column wave blocks): z_ps (packed-split, Kp columns, Kp % 32 == 0) is first
 * converted to MX3 (a_hi / a_l8 / a_sc: scratch planes with Kp rounded up to 128 columns) and W to the MX weight image of that padded K
 * (ribca_test_mx_weight_bytes(N, Kp128, .)); outputs as ribca_test_gemm_gelu_mx.  N % 192 == 0. */
int ribca_test_gemm_mx_fc1(const uint16_t* z_ps, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp, const float* bias2,
                           const float* csum, const float* rowstat, uint16_t* a_hi, uint8_t* a_l8, uint8_t* a_sc, uint16_t* wh_scratch,
                           uint8_t* wx_scratch, uint16_t* hi_out, uint8_t* l8_out, uint8_t* sc_out, void* stream);
/* ribca_test_qkv_attention_fold with the qkv product on the MX kernel (scratch planes / images as above, N = 3 D, (3 D) % 192 == 0) */
int ribca_test_qkv_attention_mx(const uint16_t* z_ps, int32_t lda, const uint16_t* W, int32_t ldw, int32_t cells, int32_t D, int32_t Kp,
                                const float* bias2, const float* csum, const float* rowstat, uint16_t* a_hi, uint8_t* a_l8, uint8_t* a_sc,
                                uint16_t* wh_scratch, uint8_t* wx_scratch, uint16_t* q, uint16_t* k, uint16_t* vt, uint16_t* out, int32_t ldo,
                                void* stream);
/* attn.proj / mlp.fc2 writing the NEW residual rows twice: packed-split into z_ps (as ribca_test_gemm_resid_ps_duo / ribca_test_gemm_mx_resid) and
 * in MX3 into z_hi / z_l8 / z_sc (row pitch z_Kp, a multiple of 128 >= N; columns >= N are not written).  kind 0: packed-split A on the
 * two-workgroups kernel's 128 x 192 tile (w_scratch: fragment-order copy of W; a_*, wx_scratch unused); kind 1: the MX kernel (a_*: MX3 image
 * of A, Kp % 128 == 0; w_scratch / wx_scratch: MX weight image).  N % 192 == 0; part: (N / 48) * M float2. */
int ribca_test_gemm_resid_zmx(int32_t kind, const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp,
                              const float* bias, uint16_t* a_hi, uint8_t* a_l8, uint8_t* a_sc, uint16_t* w_scratch, uint8_t* wx_scratch,
                              uint16_t* z_ps, int32_t ldz, float* part, float* rowstat, const float* prev, uint16_t* z_hi, uint8_t* z_l8,
                              uint8_t* z_sc, int32_t z_Kp, void* stream);
int ribca_test_gemm_fold(int32_t kind, const uint16_t* z_ps, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp,
                         const float* bias2, const float* csum, const float* rowstat, void* out, int32_t ldo, void* stream);
int ribca_test_qkv_attention_fold(const uint16_t* z_ps, int32_t lda, const uint16_t* W, int32_t ldw, int32_t cells, int32_t D, int32_t Kp,
                                  const float* bias2, const float* csum, const float* rowstat, uint16_t* q, uint16_t* k, uint16_t* vt,
                                  uint16_t* out, int32_t ldo, void* stream);
/* mlp.fc1 as the forward runs it for M >= 4096: the GELU epilogue on the two-workgroups-per-CU kernel (gemm_duo.hip), which reads the
 * weight in MFMA fragment order from wf_scratch (Np * 2 * Kp uint16, filled here from W).  csum / rowstat NULL: plain epilogue
 * (bit-identical to ribca_test_gemm kind 1), else the folded one (bit-identical to ribca_test_gemm_fold kind 1). */
int ribca_test_gemm_duo_gelu(const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp,
                             const float* bias, const float* csum, const float* rowstat, uint16_t* wf_scratch, uint16_t* out, int32_t ldo,
                             void* stream);
/* norm1 -> attn.qkv -> attention of `cells` cells in ONE per-cell kernel (cell_attention.hip; D = 144, 288 or 384): same inputs as
 * ribca_test_qkv_attention_fold, q / k / v never leave the CU.  Non-zero return: geometry not supported. */
int ribca_test_cell_attention(const uint16_t* z_ps, int32_t lda, const uint16_t* W, int32_t ldw, int32_t cells, int32_t D,
                              const float* bias2, const float* csum, const float* rowstat, uint16_t* out, int32_t ldo, void* stream);
int32_t ribca_gemm_padded_n(int32_t N);
/* Measurement hooks of the DIAGNOSTIC library (libribca_hip_diag.so, built with -DRIBCA_DIAG; tools/ only).  In the product library
 * every variant runs the production kernel.  0 = production GEMM; 3 = without the half-step stagger; 4/5/7/9/20-24 = timing
 * ablations (no loads / loads only / no loads + stagger / no epilogue / fewer MFMA passes) whose RESULTS ARE WRONG by construction;
 * 12 = production kernel + time stamps; 30 = persistent workgroups; 40-49 = the two-workgroups-per-CU kernel and its ablations. */
int ribca_set_gemm_variant(int32_t v);
/* Diagnostics for variants 12 / 48: device buffer of 20 x uint64 per workgroup receiving 100 MHz time stamps (entry, first stage
 * landed, K loop done, epilogue stores accepted), the XCC / HW id the workgroup ran on, and in [6..17] the time each of the 12
 * waves had its epilogue stores accepted.  capacity_blocks = workgroups the buffer has room for (workgroups beyond it do not
 * stamp).  NULL disables. */
int ribca_set_gemm_stamps(void* dev_buffer, int64_t capacity_blocks);
/* 1 if this library carries the diagnostic kernel forms (-DRIBCA_DIAG), else 0 */
int ribca_is_diag_build(void);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* RIBCA_HIP_TEST_H */
