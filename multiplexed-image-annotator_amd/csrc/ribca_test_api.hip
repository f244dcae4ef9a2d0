// Kernel-level hooks of tests/ and tools/ (include/ribca_hip_test.h) -- a library of their own, libribca_hip_test.so, linked against
// libribca_hip.so: every hook calls the SAME host launchers and kernels the product's forward uses (they live in libribca_hip.so; nothing
// is compiled twice), while the product library exports none of them.  A caller of the product ABI never loads this file.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <string>

#include <utility>

#include "../../include/ribca_hip.h"
#include "../../include/ribca_hip_test.h"
#include "ribca_common.h"
#include "ribca_internal.h"

// Types and constants come from the shared headers; FUNCTIONS do not: this library is linked with --no-undefined against the C entry
// points of libribca_hip.so alone, and every launcher below is reached through the product library's versioned table (ribca_internal.h).
// No `using namespace ribca`: an unqualified call must find the forwarder, never the (hidden, unlinkable) declaration.
using ribca::AttnGeom;
using ribca::GemmArgs;
using ribca::kHeads;
using ribca::kTokens;
using ribca::MxAct;
using ribca::MxWeight;
using ribca::ResidStatGeom;

namespace {
const ribca::InternalTable* table() {
  static const ribca::InternalTable* t = static_cast<const ribca::InternalTable*>(ribca_internal_table(RIBCA_INTERNAL_VERSION));
  return t;
}
// one forwarder per launcher, same name.  Function OBJECTS, not functions: ordinary lookup then finds a variable, which switches
// argument-dependent lookup off -- a call with a ribca:: argument would otherwise prefer the hidden declaration of ribca_kernels.h.
// (Default arguments are spelled out at the call sites: a call through a pointer has none.)
#define RIBCA_X(name)                                                                                      \
  struct name##_fwd {                                                                                      \
    template <class... A>                                                                                  \
    auto operator()(A&&... a) const -> decltype(table()->name(std::forward<A>(a)...)) {                    \
      return table()->name(std::forward<A>(a)...);                                                         \
    }                                                                                                      \
  };                                                                                                       \
  constexpr name##_fwd name{};
RIBCA_INTERNAL_FUNCS(RIBCA_X)
#undef RIBCA_X
int fail(const std::string& msg) { return api_fail(msg.c_str()); }
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
}  // namespace

#undef RIBCA_FINISH
#define RIBCA_FINISH()                   \
  do {                                   \
    if (api_finish() != 0) return 1;     \
  } while (0)
// every hook starts here: a product library of another build hands out no table
#define RIBCA_NEED_TABLE()                                                                                                          \
  do {                                                                                                                              \
    if (table() == nullptr) return 1;      /* (no record possible: the error text lives behind the table) */                        \
  } while (0)

extern "C" {

int32_t ribca_gemm_padded_n(int32_t N) { if (table() == nullptr) return 0; return gemm_padded_n(N); }
int ribca_set_gemm_variant(int32_t v) {
  RIBCA_NEED_TABLE(); gemm_set_variant(v); return 0; }
int ribca_set_gemm_stamps(void* dev_buffer, int64_t capacity_blocks) {
  RIBCA_NEED_TABLE(); return gemm_set_stamp_buffer(dev_buffer, capacity_blocks); }

int ribca_test_pack_weight(const float* w, int32_t N, int32_t K, uint16_t* out, int32_t Np, int32_t Kp, void* stream) {
  RIBCA_NEED_TABLE();
  launch_pack_weight(w, N, K, out, Np, Kp, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}
int ribca_test_layernorm(const float* z, int32_t ldz, const float* gamma, const float* beta, uint16_t* out, int32_t ldo, int32_t M,
                         int32_t D, void* stream) {
  RIBCA_NEED_TABLE();
  launch_layernorm_ps(z, ldz, gamma, beta, out, ldo, M, D, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}
int ribca_test_gemm(int32_t kind, const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp,
                    const float* bias, void* out, int32_t ldo, void* stream) {
  RIBCA_NEED_TABLE();
  GemmArgs g{A, lda, W, ldw, M, N, Kp, bias};
  if (kind == 0) launch_gemm_resid(g, (float*)out, ldo, (hipStream_t)stream);
  else if (kind == 1) launch_gemm_gelu(g, (uint16_t*)out, ldo, (hipStream_t)stream);
  else return fail("ribca_test_gemm: kind must be 0 or 1");
  RIBCA_FINISH();
  return 0;
}
int ribca_test_fold_weight(const float* w, int32_t N, int32_t K, const float* gamma, const float* beta, const float* bias, uint16_t* out,
                           int32_t Np, int32_t Kp, float* csum, float* bias2, void* stream) {
  RIBCA_NEED_TABLE();
  if (Np != gemm_padded_n(N)) return fail("ribca_test_fold_weight: Np must be ribca_gemm_padded_n(N)");
  launch_pack_weight_fold(w, N, K, gamma, beta, bias, out, Np, Kp, csum, bias2, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}
int ribca_test_row_stats(uint16_t* z_ps, int32_t ldz, int32_t M, int32_t D, float* rowstat, int32_t recentre, void* stream) {
  RIBCA_NEED_TABLE();
  launch_row_stats_ps(z_ps, ldz, M, D, reinterpret_cast<float2*>(rowstat), recentre != 0, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}
int32_t ribca_test_resid_tiles(int32_t N) { if (table() == nullptr) return 0; return gemm_resid_tiles(N); }
int ribca_test_gemm_resid_ps(const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp,
                             const float* bias, uint16_t* z_ps, int32_t ldz, float* part, float* rowstat, const float* prev, void* stream) {
  RIBCA_NEED_TABLE();
  if (N % 8 != 0) return fail("ribca_test_gemm_resid_ps: N must be a multiple of 8");
  if ((rowstat != nullptr) != (part != nullptr)) return fail("ribca_test_gemm_resid_ps: part and rowstat go together");
  GemmArgs g{A, lda, W, ldw, M, N, Kp, bias};
  const ResidStatGeom sg = launch_gemm_resid_ps(g, z_ps, ldz, reinterpret_cast<float2*>(part), reinterpret_cast<const float2*>(prev), 1, (hipStream_t)stream, false, nullptr);
  if (rowstat) launch_ln_finalize(reinterpret_cast<const float2*>(part), sg.tiles, M, sg.bn, N, reinterpret_cast<float2*>(rowstat), (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}
int32_t ribca_test_resid_part_rows(int32_t N) { if (table() == nullptr) return 0; return gemm_resid_part_rows(N); }
// the same update on the two-workgroups-per-CU kernel with the residual tile riding the A ring (EpiResidZK): what the classifiers' full
// blocks run for proj / fc2 where it measured faster (any M: the choice depends on the shape of the weight alone).  part needs ribca_test_resid_part_rows(N) x M pairs; wf_scratch the size of the packed weight.
int ribca_test_gemm_resid_ps_duo(const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp,
                                 const float* bias, uint16_t* wf_scratch, uint16_t* z_ps, int32_t ldz, float* part, float* rowstat,
                                 const float* prev, void* stream) {
  RIBCA_NEED_TABLE();
  if (!wf_scratch) return fail("ribca_test_gemm_resid_ps_duo: wf_scratch is NULL");
  if (N % 8 != 0) return fail("ribca_test_gemm_resid_ps_duo: N must be a multiple of 8");
  if ((rowstat != nullptr) != (part != nullptr)) return fail("ribca_test_gemm_resid_ps_duo: part and rowstat go together");
  launch_pack_wf(W, ldw, gemm_padded_n(N), Kp, wf_scratch, (hipStream_t)stream);
  GemmArgs g{A, lda, W, ldw, M, N, Kp, bias, wf_scratch};
  const ResidStatGeom sg = launch_gemm_resid_ps(g, z_ps, ldz, reinterpret_cast<float2*>(part), reinterpret_cast<const float2*>(prev), 1, (hipStream_t)stream,
                                                true, nullptr);
  if (sg.bn == gemm_resid_bn(N))      // the duo kernel's wave blocks are 16 / 32 / 48 columns wide, never a tile width
    return fail("ribca_test_gemm_resid_ps_duo: the two-workgroups-per-CU kernel did not take this shape");
  if (rowstat) launch_ln_finalize(reinterpret_cast<const float2*>(part), sg.tiles, M, sg.bn, N, reinterpret_cast<float2*>(rowstat), (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}
int64_t ribca_test_mx_weight_bytes(int32_t N, int32_t Kp, int32_t which) { if (table() == nullptr) return 0;
  const int Np = (N + 15) / 16 * 16;
  return (int64_t)(which == 0 ? mx_wh_bytes(Np, Kp) : mx_wx_bytes(Np, Kp));
}
int ribca_test_mx_pack_act(const uint16_t* A, int32_t lda, int32_t M, int32_t Kp, uint16_t* hi_out, uint8_t* l8_out, uint8_t* sc_out, void* stream) {
  RIBCA_NEED_TABLE();
  if (!A || !hi_out || !l8_out || !sc_out) return fail("ribca_test_mx_pack_act: NULL buffer");
  if (Kp % 32 != 0) return fail("ribca_test_mx_pack_act: Kp must be a multiple of 32");
  const MxAct a{hi_out, l8_out, sc_out, round_up(Kp, 128), M};
  launch_mx_pack_act(A, lda, M, Kp, a, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}
int ribca_test_gemm_mx_resid(const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp, const float* bias,
                             uint16_t* hi_out, uint8_t* l8_out, uint8_t* sc_out, uint16_t* wh_scratch, uint8_t* wx_scratch, uint16_t* z_ps,
                             int32_t ldz, float* part, float* rowstat, const float* prev, void* stream) {
  RIBCA_NEED_TABLE();
  if (!gemm_mx_supported(N, Kp)) return fail("ribca_test_gemm_mx_resid: N must be a multiple of 48 and Kp of 128");
  if ((rowstat != nullptr) != (part != nullptr)) return fail("ribca_test_gemm_mx_resid: part and rowstat go together");
  hipStream_t s = (hipStream_t)stream;
  const MxAct a{hi_out, l8_out, sc_out, Kp, M};
  launch_mx_pack_act(A, lda, M, Kp, a, s);
  launch_mx_pack_w(W, ldw, (N + 15) / 16 * 16, Kp, Kp, wh_scratch, wx_scratch, s);
  const MxWeight w{wh_scratch, wx_scratch};
  const ResidStatGeom sg = launch_gemm_mx_resid(a, w, M, N, bias, z_ps, ldz, reinterpret_cast<float2*>(part), reinterpret_cast<const float2*>(prev), 1, s, 0, nullptr);
  if (rowstat) launch_ln_finalize(reinterpret_cast<const float2*>(part), sg.tiles, M, sg.bn, N, reinterpret_cast<float2*>(rowstat), s);
  RIBCA_FINISH();
  return 0;
}
int ribca_test_gemm_mx_resid_packed(const uint16_t* hi, const uint8_t* l8, const uint8_t* sc, int32_t Kp, const uint16_t* wh, const uint8_t* wx,
                                    int32_t M, int32_t N, const float* bias, uint16_t* z_ps, int32_t ldz, float* part, float* rowstat,
                                    const float* prev, void* stream) {
  RIBCA_NEED_TABLE();
  if (!gemm_mx_supported(N, Kp)) return fail("ribca_test_gemm_mx_resid_packed: N must be a multiple of 48 and Kp of 128");
  if ((rowstat != nullptr) != (part != nullptr)) return fail("ribca_test_gemm_mx_resid_packed: part and rowstat go together");
  hipStream_t s = (hipStream_t)stream;
  const MxAct a{const_cast<uint16_t*>(hi), const_cast<uint8_t*>(l8), const_cast<uint8_t*>(sc), Kp, M};
  const MxWeight w{wh, wx};
  const ResidStatGeom sg = launch_gemm_mx_resid(a, w, M, N, bias, z_ps, ldz, reinterpret_cast<float2*>(part), reinterpret_cast<const float2*>(prev), 1, s, 0, nullptr);
  if (rowstat) launch_ln_finalize(reinterpret_cast<const float2*>(part), sg.tiles, M, sg.bn, N, reinterpret_cast<float2*>(rowstat), s);
  RIBCA_FINISH();
  return 0;
}
int ribca_test_gemm_gelu_mx(const uint16_t* z_ps, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp, const float* bias2,
                            const float* csum, const float* rowstat, uint16_t* wf_scratch, uint16_t* hi_out, uint8_t* l8_out, uint8_t* sc_out,
                            void* stream) {
  RIBCA_NEED_TABLE();
  if (!wf_scratch || !csum || !rowstat) return fail("ribca_test_gemm_gelu_mx: NULL buffer");
  launch_pack_wf(W, ldw, gemm_padded_n(N), Kp, wf_scratch, (hipStream_t)stream);
  GemmArgs g{z_ps, lda, W, ldw, M, N, Kp, bias2, wf_scratch};
  const MxAct out{hi_out, l8_out, sc_out, N, M};
  if (!launch_gemm_gelu_mx(g, reinterpret_cast<const float2*>(rowstat), csum, out, (hipStream_t)stream))
    return fail("ribca_test_gemm_gelu_mx: N must be a multiple of 128");
  RIBCA_FINISH();
  return 0;
}
// mlp.fc1 on the MX kernel: z (packed-split, Kp columns) -> MX3 (a_*: scratch planes, Kp rounded up to 128) -> gelu(LN-folded product) in MX3
int ribca_test_gemm_mx_fc1(const uint16_t* z_ps, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp, const float* bias2,
                           const float* csum, const float* rowstat, uint16_t* a_hi, uint8_t* a_l8, uint8_t* a_sc, uint16_t* wh_scratch,
                           uint8_t* wx_scratch, uint16_t* hi_out, uint8_t* l8_out, uint8_t* sc_out, void* stream) {
  RIBCA_NEED_TABLE();
  if (!csum || !rowstat || !a_hi || !a_l8 || !a_sc || !wh_scratch || !wx_scratch) return fail("ribca_test_gemm_mx_fc1: NULL buffer");
  if (N % 192 != 0 || Kp % 32 != 0) return fail("ribca_test_gemm_mx_fc1: N must be a multiple of 192 and Kp of 32");
  hipStream_t s = (hipStream_t)stream;
  const int Kz = round_up(Kp, 128);
  const MxAct a{a_hi, a_l8, a_sc, Kz, M};
  launch_mx_pack_act(z_ps, lda, M, Kp, a, s);
  launch_mx_pack_w(W, ldw, N, Kp, Kz, wh_scratch, wx_scratch, s);
  const MxAct out{hi_out, l8_out, sc_out, N, M};
  launch_gemm_mx_gelu(a, MxWeight{wh_scratch, wx_scratch}, M, N, bias2, reinterpret_cast<const float2*>(rowstat), csum, out, s);
  RIBCA_FINISH();
  return 0;
}
// ribca_test_qkv_attention_fold with the qkv product on the MX kernel
int ribca_test_qkv_attention_mx(const uint16_t* z_ps, int32_t lda, const uint16_t* W, int32_t ldw, int32_t cells, int32_t D, int32_t Kp,
                                const float* bias2, const float* csum, const float* rowstat, uint16_t* a_hi, uint8_t* a_l8, uint8_t* a_sc,
                                uint16_t* wh_scratch, uint8_t* wx_scratch, uint16_t* q, uint16_t* k, uint16_t* vt, uint16_t* out, int32_t ldo,
                                void* stream) {
  RIBCA_NEED_TABLE();
  if (!csum || !rowstat || !a_hi || !a_l8 || !a_sc || !wh_scratch || !wx_scratch) return fail("ribca_test_qkv_attention_mx: NULL buffer");
  if ((3 * D) % 192 != 0 || Kp % 32 != 0) return fail("ribca_test_qkv_attention_mx: 3 D must be a multiple of 192 and Kp of 32");
  hipStream_t s = (hipStream_t)stream;
  const AttnGeom g = make_attn_geom(D, kHeads, kTokens);
  const int M = cells * kTokens, Kz = round_up(Kp, 128);
  const MxAct a{a_hi, a_l8, a_sc, Kz, M};
  launch_mx_pack_act(z_ps, lda, M, Kp, a, s);
  launch_mx_pack_w(W, ldw, 3 * D, Kp, Kz, wh_scratch, wx_scratch, s);
  launch_gemm_mx_qkv_ln(a, MxWeight{wh_scratch, wx_scratch}, M, 3 * D, bias2, reinterpret_cast<const float2*>(rowstat), csum, q, k, vt, g,
                        1.0f / sqrtf((float)g.hd), s);
  launch_attention(q, k, vt, out, ldo, cells, g, s, 0);
  RIBCA_FINISH();
  return 0;
}
// proj / fc2 writing the new rows a second time in MX3 (z_hi / z_l8 / z_sc, row pitch z_Kp).  kind 0: the packed-split operand on the
// two-workgroups kernel (w_scratch: fragment-order copy of W; a_* and wx_scratch unused); kind 1: the MX kernel (a_*: MX3 image of A,
// w_scratch / wx_scratch: the MX weight image)
int ribca_test_gemm_resid_zmx(int32_t kind, const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp,
                              const float* bias, uint16_t* a_hi, uint8_t* a_l8, uint8_t* a_sc, uint16_t* w_scratch, uint8_t* wx_scratch,
                              uint16_t* z_ps, int32_t ldz, float* part, float* rowstat, const float* prev, uint16_t* z_hi, uint8_t* z_l8,
                              uint8_t* z_sc, int32_t z_Kp, void* stream) {
  RIBCA_NEED_TABLE();
  if (!w_scratch || !z_hi || !z_l8 || !z_sc || !part || !rowstat) return fail("ribca_test_gemm_resid_zmx: NULL buffer");
  if (N % 192 != 0 || z_Kp % 128 != 0 || z_Kp < N) return fail("ribca_test_gemm_resid_zmx: N must be a multiple of 192, z_Kp of 128 and >= N");
  hipStream_t s = (hipStream_t)stream;
  const MxAct zmx{z_hi, z_l8, z_sc, z_Kp, M};
  ResidStatGeom sg;
  if (kind == 0) {
    launch_pack_wf(W, ldw, gemm_padded_n(N), Kp, w_scratch, s);
    GemmArgs g{A, lda, W, ldw, M, N, Kp, bias, w_scratch};
    sg = launch_gemm_resid_ps(g, z_ps, ldz, reinterpret_cast<float2*>(part), reinterpret_cast<const float2*>(prev), 1, s, false, &zmx);
  } else if (kind == 1) {
    if (!a_hi || !a_l8 || !a_sc || !wx_scratch || Kp % 128 != 0) return fail("ribca_test_gemm_resid_zmx: kind 1 needs the MX3 scratch planes and Kp % 128 == 0");
    const MxAct a{a_hi, a_l8, a_sc, Kp, M};
    launch_mx_pack_act(A, lda, M, Kp, a, s);
    launch_mx_pack_w(W, ldw, N, Kp, Kp, w_scratch, wx_scratch, s);
    sg = launch_gemm_mx_resid(a, MxWeight{w_scratch, wx_scratch}, M, N, bias, z_ps, ldz, reinterpret_cast<float2*>(part), reinterpret_cast<const float2*>(prev), 1,
                              s, 0, &zmx);
  } else {
    return fail("ribca_test_gemm_resid_zmx: kind must be 0 or 1");
  }
  launch_ln_finalize(reinterpret_cast<const float2*>(part), sg.tiles, M, sg.bn, N, reinterpret_cast<float2*>(rowstat), s);
  RIBCA_FINISH();
  return 0;
}
int ribca_test_gemm_fold(int32_t kind, const uint16_t* z_ps, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp,
                         const float* bias2, const float* csum, const float* rowstat, void* out, int32_t ldo, void* stream) {
  RIBCA_NEED_TABLE();
  if (kind != 1) return fail("ribca_test_gemm_fold: kind must be 1");
  GemmArgs g{z_ps, lda, W, ldw, M, N, Kp, bias2};
  launch_gemm_gelu_ln(g, reinterpret_cast<const float2*>(rowstat), csum, (uint16_t*)out, ldo, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}
int ribca_test_qkv_attention_fold(const uint16_t* z_ps, int32_t lda, const uint16_t* W, int32_t ldw, int32_t cells, int32_t D, int32_t Kp,
                                  const float* bias2, const float* csum, const float* rowstat, uint16_t* q, uint16_t* k, uint16_t* vt,
                                  uint16_t* out, int32_t ldo, void* stream) {
  RIBCA_NEED_TABLE();
  const AttnGeom a = make_attn_geom(D, kHeads, kTokens);
  GemmArgs g{z_ps, lda, W, ldw, cells * kTokens, 3 * D, Kp, bias2};
  launch_gemm_qkv_ln(g, reinterpret_cast<const float2*>(rowstat), csum, q, k, vt, a, 1.0f / sqrtf((float)a.hd), (hipStream_t)stream, 0, 0, 1);
  launch_attention(q, k, vt, out, ldo, cells, a, (hipStream_t)stream, 0);
  RIBCA_FINISH();
  return 0;
}
int ribca_test_gemm_duo_gelu(const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp,
                             const float* bias, const float* csum, const float* rowstat, uint16_t* wf_scratch, uint16_t* out, int32_t ldo,
                             void* stream) {
  RIBCA_NEED_TABLE();
  if (M < 4096) return fail("ribca_test_gemm_duo_gelu: the forward uses this kernel for M >= 4096 only");
  if (!wf_scratch) return fail("ribca_test_gemm_duo_gelu: wf_scratch is NULL");
  if ((csum != nullptr) != (rowstat != nullptr)) return fail("ribca_test_gemm_duo_gelu: csum and rowstat go together");
  launch_pack_wf(W, ldw, gemm_padded_n(N), Kp, wf_scratch, (hipStream_t)stream);
  GemmArgs g{A, lda, W, ldw, M, N, Kp, bias, wf_scratch};
  if (csum) launch_gemm_gelu_ln(g, reinterpret_cast<const float2*>(rowstat), csum, out, ldo, (hipStream_t)stream);
  else launch_gemm_gelu(g, out, ldo, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}
int ribca_test_cell_attention(const uint16_t* z_ps, int32_t lda, const uint16_t* W, int32_t ldw, int32_t cells, int32_t D,
                              const float* bias2, const float* csum, const float* rowstat, uint16_t* out, int32_t ldo, void* stream) {
  RIBCA_NEED_TABLE();
  if (!cell_attention_supported(D, kHeads, kTokens)) return fail("ribca_test_cell_attention: D must be 144, 288 or 384");
  launch_cell_qkv_attention(z_ps, lda, W, ldw, bias2, csum, reinterpret_cast<const float2*>(rowstat), out, ldo, cells, D,
                            1.0f / sqrtf((float)(D / kHeads)), (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}
int ribca_is_diag_build(void) {
#ifdef RIBCA_DIAG
  return 1;
#else
  return 0;
#endif
}
int ribca_test_qkv_attention(const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t cells, int32_t D, int32_t Kp,
                             const float* bias, uint16_t* q, uint16_t* k, uint16_t* vt, uint16_t* out, int32_t ldo, void* stream) {
  RIBCA_NEED_TABLE();
  const AttnGeom a = make_attn_geom(D, kHeads, kTokens);
  GemmArgs g{A, lda, W, ldw, cells * kTokens, 3 * D, Kp, bias};
  launch_gemm_qkv(g, q, k, vt, a, 1.0f / sqrtf((float)a.hd), (hipStream_t)stream);
  launch_attention(q, k, vt, out, ldo, cells, a, (hipStream_t)stream, 0);
  RIBCA_FINISH();
  return 0;
}

}  // extern "C"
