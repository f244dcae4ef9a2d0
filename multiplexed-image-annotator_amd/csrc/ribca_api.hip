// C ABI of libribca_hip.so (see include/ribca_hip.h): handle management, workspace carving and the launch sequence of
// the ViT forward.  No torch types; the caller (Python via ctypes) owns all buffers except the packed-weight handle.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ribca_hip.h"
#include "ribca_common.h"
#include "ribca_internal.h"
#include "ribca_kernels.h"
#include "ribca_status.h"

using namespace ribca;

namespace {

// (status plumbing: ribca_status.h -- shared with the test-hook library)
int fail(const std::string& msg) { return api_fail(msg.c_str()); }
int hip_fail(hipError_t e, const char* what) { return api_hip_fail(e, what); }

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// ----------------------------------------------------------------------------------------------- profiling
enum ProfClass { P_QKV = 0, P_PROJ, P_FC1, P_FC2, P_EMBED, P_ATTN, P_LN, P_CELL, P_HEAD, P_OTHER, P_COUNT };
const char* kProfNames[P_COUNT] = {"gemm_qkv", "gemm_proj", "gemm_fc1", "gemm_fc2", "gemm_embed", "attention", "layernorm", "cell_qkv_attention",
                                   "head", "other"};
struct ProfRec { hipEvent_t a, b; int cls; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
double g_prof_ms[P_COUNT] = {0};
long long g_prof_n[P_COUNT] = {0};

void prof_drain();

struct ProfScope {
  hipStream_t s; int idx = -1;
  ProfScope(int cls, hipStream_t st) : s(st) {
    if (!g_prof_on) return;
    if (g_prof.size() >= 2048) prof_drain();     // bound the number of live HIP events (profiling mode only)
    ProfRec r; r.cls = cls;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
    idx = (int)g_prof.size() - 1;
  }
  ~ProfScope() { if (idx >= 0) (void)hipEventRecord(g_prof[idx].b, s); }
};

void prof_drain() {
  for (auto& r : g_prof) {
    (void)hipEventSynchronize(r.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { g_prof_ms[r.cls] += ms; g_prof_n[r.cls] += 1; }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof.clear();
}

}  // namespace

namespace ribca {
namespace {
thread_local std::string g_err;
// what a launcher recorded with launch_error() since the calling thread's last api_finish() (the first record wins: later ones are
// usually its consequences)
thread_local std::string g_launch_err;
}  // namespace
void launch_error(const char* fmt, ...) {
  if (!g_launch_err.empty()) return;
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_launch_err = buf;
}
int api_fail(const char* msg) {
  g_err = msg ? msg : "unknown error";
  g_launch_err.clear();
  return 1;
}
int api_hip_fail(hipError_t e, const char* what) { return api_fail((std::string(what) + ": " + hipGetErrorString(e)).c_str()); }
int api_finish() {
  if (!g_launch_err.empty()) {
    const std::string m = g_launch_err;
    (void)hipGetLastError();
    return api_fail(m.c_str());
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : api_hip_fail(e, "kernel launch");
}
const char* api_last_error() { return g_err.c_str(); }
}  // namespace ribca

// ----------------------------------------------------------------------------------------------- model handles
// weights of one pre-LN transformer block (timm Block): fp32 vectors + packed-split matrices
struct BlockW {
  const float *ln1w, *ln1b, *qkvb, *projb, *ln2w, *ln2b, *fc1b, *fc2b;
  const uint16_t *qkvw, *projw, *fc1w, *fc2w;
  const uint16_t* qkvwf = nullptr;   // folded qkv weight in fragment order: the qkv product runs on the duo kernel as well
  const uint16_t* fc1wf;   // fc1 weight again, in MFMA fragment order (gemm_duo.hip): the GELU GEMMs run on the two-workgroups-per-CU kernel
  const uint16_t *projwf = nullptr, *fc2wf = nullptr;   // classifiers: proj / fc2 in fragment order too (residual through the ring, EpiResidZK)
  // LayerNorm folded into qkv / fc1 (classifiers): qkvw / fc1w then hold gamma o W, and per output column the sum of the packed row
  // and the bias with beta folded in (vit_misc.hip pack_weight_fold_kernel)
  const float *qkvc = nullptr, *qkvb2 = nullptr, *fc1c = nullptr, *fc1b2 = nullptr;
  // mlp.fc2 in the MX weight image (gemm_mx.hip) where the width allows (4 D % 128 == 0, D % 48 == 0): fc1 then writes its GELU output in
  // the MX3 format and fc2 runs as fp16 hi * hi + two block-scaled corrections
  const uint16_t* fc2mxh = nullptr; const unsigned char* fc2mxx = nullptr;
  // attn.qkv / mlp.fc1 (gamma o W) in the MX weight image, K padded to a multiple of 128, where the residual rows are kept in MX3 as well
  // (mx_z_on): proj and fc2 then write the new rows twice -- packed-split for the residual tile of the next proj / fc2, MX3 for these
  const uint16_t *qkvmxh = nullptr, *fc1mxh = nullptr; const unsigned char *qkvmxx = nullptr, *fc1mxx = nullptr;
};

// RIBCA_MX=0: the fp16x3 kernels everywhere (A/B).  The choice depends on the model's width alone, never on the chunk: a cell's bits must
// not depend on the size of the chunk it was computed in.
bool mx_on(int D) {
  static const bool on = !(getenv("RIBCA_MX") && atoi(getenv("RIBCA_MX")) == 0);
  return on && (4 * D) % 128 == 0 && gemm_mx_supported(D, 4 * D);
}

// RIBCA_MXZ=0: qkv / fc1 stay on the fp16x3 kernels (A/B).  D % 192 == 0: proj / fc2 emit the MX3 copy from 128 x 192 tiles only, and the
// 32-column scale blocks need D % 96 == 0 (gemm_epi.h mx3_emit_wave48); 3 D and 4 D are then multiples of 192 as well.
bool mx_z_on(int D) {
  static const bool on = !(getenv("RIBCA_MXZ") && atoi(getenv("RIBCA_MXZ")) == 0);
  return on && mx_on(D) && D % 192 == 0;
}

struct ribca_vit {
  int D, C, K, depth, hd, hdp, hdv, Dp, H4;
  bool fold = true;        // LayerNorm folded into the qkv / fc1 GEMMs, residual stream packed-split (RIBCA_LN_FOLD=0 at create: round-2 path)
  char* arena = nullptr;
  size_t arena_bytes = 0;
  const float *cls, *pos, *pe_b, *norm_w, *norm_b, *head_w, *head_b;
  const float* pe_w;   // fp32 [D][16*C]: the patch embedding stays in fp32 (vit_misc.hip embed_f32_kernel)
  std::vector<BlockW> layers;
};

// marker imputer (reference markerImputer.py:69-329): encoder 768 / 12 heads, decoder 512 / 8 heads, tokens = channels
struct ribca_mae {
  int L, enc_depth, dec_depth;
  // round 5: the blocks run on the classifiers' folded path (LayerNorm folded into qkv / fc1, packed-split residual stream, residual tile
  // through the operand ring; the 768-wide encoder on the MX kernel as well).  RIBCA_MAE_FOLD=0 at create: the round-2 path (A/B)
  bool fold = true;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  const float *cls, *pos, *pe_b, *norm_w, *norm_b, *de_b, *mask_tok, *dpos, *dnorm_w, *dnorm_b, *pred_b;
  const uint16_t *pe_w, *de_w, *pred_w;
  std::vector<BlockW> enc, dec;
};

namespace {

constexpr int kEncD = 768, kEncH = 12, kDecD = 512, kDecH = 8, kTokPix = 1600;

struct Carver {
  char* base; size_t off = 0;
  explicit Carver(char* b) : base(b) {}
  template <class T> T* take(size_t count) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off = align256(off + count * sizeof(T));
    return p;
  }
};

void layout_block(Carver& c, BlockW& L, int D, bool fold = false) {
  const int Dp = round_up(D, 32), H4 = 4 * D;
  if (fold) {
    L.qkvc = c.take<float>(3 * D); L.qkvb2 = c.take<float>(3 * D);
    L.fc1c = c.take<float>(4 * D); L.fc1b2 = c.take<float>(4 * D);
  }
  L.ln1w = c.take<float>(D); L.ln1b = c.take<float>(D);
  L.qkvb = c.take<float>(3 * D); L.projb = c.take<float>(D);
  L.ln2w = c.take<float>(D); L.ln2b = c.take<float>(D);
  L.fc1b = c.take<float>(4 * D); L.fc2b = c.take<float>(D);
  L.qkvw = c.take<uint16_t>((size_t)gemm_padded_n(3 * D) * 2 * Dp);
  if (fold) L.qkvwf = c.take<uint16_t>((size_t)gemm_padded_n(3 * D) * 2 * Dp);
  L.projw = c.take<uint16_t>((size_t)gemm_padded_n(D) * 2 * Dp);
  if (fold) L.projwf = c.take<uint16_t>((size_t)gemm_padded_n(D) * 2 * Dp);
  L.fc1w = c.take<uint16_t>((size_t)gemm_padded_n(4 * D) * 2 * Dp);
  L.fc1wf = c.take<uint16_t>((size_t)gemm_padded_n(4 * D) * 2 * Dp);
  L.fc2w = c.take<uint16_t>((size_t)gemm_padded_n(D) * 2 * H4);
  if (fold) L.fc2wf = c.take<uint16_t>((size_t)gemm_padded_n(D) * 2 * H4);
  if (fold && mx_on(D)) {
    L.fc2mxh = c.take<uint16_t>(mx_wh_bytes(round_up(D, 16), H4) / 2);
    L.fc2mxx = c.take<unsigned char>(mx_wx_bytes(round_up(D, 16), H4));
  }
  if (fold && mx_z_on(D)) {
    const int Kz = round_up(D, 128);
    L.qkvmxh = c.take<uint16_t>(mx_wh_bytes(3 * D, Kz) / 2); L.qkvmxx = c.take<unsigned char>(mx_wx_bytes(3 * D, Kz));
    L.fc1mxh = c.take<uint16_t>(mx_wh_bytes(4 * D, Kz) / 2); L.fc1mxx = c.take<unsigned char>(mx_wx_bytes(4 * D, Kz));
  }
}

// lays the arena out; with base == nullptr only measures
size_t layout(ribca_vit* m, char* base) {
  Carver c(base);
  const int D = m->D;
  m->cls = c.take<float>(D);
  m->pos = c.take<float>((size_t)kTokens * D);
  m->pe_b = c.take<float>(D);
  m->pe_w = c.take<float>((size_t)D * 16 * m->C);
  m->layers.resize(m->depth);
  for (auto& L : m->layers) layout_block(c, L, D, m->fold);
  m->norm_w = c.take<float>(D); m->norm_b = c.take<float>(D);
  m->head_w = c.take<float>((size_t)m->K * D); m->head_b = c.take<float>(m->K);
  return c.off;
}

size_t layout_mae(ribca_mae* m, char* base) {
  Carver c(base);
  m->cls = c.take<float>(kEncD);
  m->pos = c.take<float>((size_t)(m->L + 1) * kEncD);
  m->pe_w = c.take<uint16_t>((size_t)gemm_padded_n(kEncD) * 2 * kTokPix);
  m->pe_b = c.take<float>(kEncD);
  m->enc.resize(m->enc_depth);
  for (auto& L : m->enc) layout_block(c, L, kEncD, m->fold);
  m->norm_w = c.take<float>(kEncD); m->norm_b = c.take<float>(kEncD);
  m->de_w = c.take<uint16_t>((size_t)gemm_padded_n(kDecD) * 2 * kEncD);
  m->de_b = c.take<float>(kDecD);
  m->mask_tok = c.take<float>(kDecD);
  m->dpos = c.take<float>((size_t)(m->L + 1) * kDecD);
  m->dec.resize(m->dec_depth);
  for (auto& L : m->dec) layout_block(c, L, kDecD, m->fold);
  m->dnorm_w = c.take<float>(kDecD); m->dnorm_b = c.take<float>(kDecD);
  m->pred_w = c.take<uint16_t>((size_t)gemm_padded_n(kTokPix) * 2 * kDecD);
  m->pred_b = c.take<float>(kTokPix);
  return c.off;
}

// copies / packs one block's parameters from the flat fp32 blob (state-dict order), advancing p
struct BlobReader {
  const float* p; hipStream_t s; hipError_t err = hipSuccess;
  void copy(const float* dst, size_t n) {
    hipError_t r = hipMemcpyAsync((void*)dst, p, n * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (err == hipSuccess) err = r;
    p += n;
  }
  void pack(const uint16_t* dst, int N, int K, int Kp) {
    launch_pack_weight(p, N, K, const_cast<uint16_t*>(dst), gemm_padded_n(N), Kp, s);
    p += (size_t)N * K;
  }
  // weight [N][K] and its bias [N] follow each other in the blob; gamma / beta were copied to the arena just before (same stream)
  void pack_fold(const uint16_t* dst, int N, int K, int Kp, const float* gamma, const float* beta, const float* csum, const float* bias2) {
    launch_pack_weight_fold(p, N, K, gamma, beta, p + (size_t)N * K, const_cast<uint16_t*>(dst), gemm_padded_n(N), Kp, const_cast<float*>(csum),
                            const_cast<float*>(bias2), s);
    p += (size_t)N * K;
  }
  void block(const BlockW& L, int D, bool fold = false) {
    const int Dp = round_up(D, 32);
    copy(L.ln1w, D); copy(L.ln1b, D);
    if (fold) {
      pack_fold(L.qkvw, 3 * D, D, Dp, L.ln1w, L.ln1b, L.qkvc, L.qkvb2);
      launch_pack_wf(L.qkvw, 2 * Dp, gemm_padded_n(3 * D), Dp, const_cast<uint16_t*>(L.qkvwf), s);
    } else pack(L.qkvw, 3 * D, D, Dp);
    copy(L.qkvb, 3 * D);
    pack(L.projw, D, D, Dp); copy(L.projb, D);
    if (fold) launch_pack_wf(L.projw, 2 * Dp, gemm_padded_n(D), Dp, const_cast<uint16_t*>(L.projwf), s);
    copy(L.ln2w, D); copy(L.ln2b, D);
    if (fold) pack_fold(L.fc1w, 4 * D, D, Dp, L.ln2w, L.ln2b, L.fc1c, L.fc1b2);
    else pack(L.fc1w, 4 * D, D, Dp);
    copy(L.fc1b, 4 * D);
    launch_pack_wf(L.fc1w, 2 * Dp, gemm_padded_n(4 * D), Dp, const_cast<uint16_t*>(L.fc1wf), s);
    pack(L.fc2w, D, 4 * D, 4 * D); copy(L.fc2b, D);
    if (fold) launch_pack_wf(L.fc2w, 2 * 4 * D, gemm_padded_n(D), 4 * D, const_cast<uint16_t*>(L.fc2wf), s);
    if (L.fc2mxh) launch_mx_pack_w(L.fc2w, 2 * 4 * D, round_up(D, 16), 4 * D, 4 * D, const_cast<uint16_t*>(L.fc2mxh), const_cast<unsigned char*>(L.fc2mxx), s);
    if (L.qkvmxh) {
      const int Kz = round_up(D, 128);
      launch_mx_pack_w(L.qkvw, 2 * Dp, 3 * D, Dp, Kz, const_cast<uint16_t*>(L.qkvmxh), const_cast<unsigned char*>(L.qkvmxx), s);
      launch_mx_pack_w(L.fc1w, 2 * Dp, 4 * D, Dp, Kz, const_cast<uint16_t*>(L.fc1mxh), const_cast<unsigned char*>(L.fc1mxx), s);
    }
  }
};
int64_t block_params(int64_t d) { return 2 * d + 3 * d * d + 3 * d + d * d + d + 2 * d + 4 * d * d + 4 * d + 4 * d * d + d; }

// scratch for a run of transformer blocks over `cells` cells of T tokens, width D
struct BlockWs {
  float* z; uint16_t* xa; uint16_t* q; uint16_t* k; uint16_t* vt; uint16_t* h;
  size_t qk_bytes, vt_bytes, xa_bytes;
  // folded-LayerNorm blocks: residual stream packed-split [rows][2 * Dp], per-tile row statistics of the last residual GEMM and
  // the (rstd, -mean rstd) pairs the next qkv / fc1 epilogue reads
  uint16_t* zps = nullptr; float2* part = nullptr; float2* rs = nullptr;
  // the residual rows again in MX3 (mx_z_on; Kp = D rounded up to 128, M set per chunk by the caller) and the bytes of its planes
  MxAct zmx = MxAct{nullptr, nullptr, nullptr, 0, 0};
  size_t zmx_rows = 0;
};
BlockWs carve_blocks(Carver& c, int cells, const AttnGeom& a, bool fold = false) {
  BlockWs w;
  const size_t Mc = (size_t)cells * a.T;
  const int Dp = round_up(a.D, 32);
  if (fold) {
    w.z = nullptr;
    w.zps = c.take<uint16_t>(Mc * 2 * Dp);
    w.part = c.take<float2>(Mc * gemm_resid_part_rows(a.D));
    w.rs = c.take<float2>(Mc);
    if (mx_z_on(a.D)) {
      const int Kz = round_up(a.D, 128);
      w.zmx.hi = c.take<uint16_t>(Mc * Kz);
      w.zmx.l8 = c.take<unsigned char>(Mc * Kz);
      w.zmx.sc = c.take<unsigned char>(Mc * (Kz / 32));
      w.zmx.Kp = Kz;
      w.zmx_rows = Mc;
    }
  } else {
    w.z = c.take<float>(Mc * a.D);
  }
  w.xa_bytes = Mc * 2 * Dp * sizeof(uint16_t);
  w.xa = c.take<uint16_t>(Mc * 2 * Dp);
  const size_t qk = (size_t)cells * a.H * a.TP * 2 * a.hdq;
  w.qk_bytes = qk * sizeof(uint16_t);
  w.q = c.take<uint16_t>(qk);
  w.k = c.take<uint16_t>(qk);
  const size_t vt = attention_v_elems(a, cells);
  w.vt_bytes = vt * sizeof(uint16_t);
  w.vt = c.take<uint16_t>(vt);
  w.h = c.take<uint16_t>(Mc * 2 * 4 * a.D);
  return w;
}
// pads (tokens >= T, head dims >= hd, feature columns >= D) are never written by any kernel: zero them once per call
int zero_pads(const BlockWs& w, hipStream_t s) {
  HIP_TRY(hipMemsetAsync(w.xa, 0, w.xa_bytes, s));
  if (w.zps) HIP_TRY(hipMemsetAsync(w.zps, 0, w.xa_bytes, s));     // same shape as xa: the K pad of the next GEMM must read as zeros
  HIP_TRY(hipMemsetAsync(w.q, 0, w.qk_bytes, s));
  HIP_TRY(hipMemsetAsync(w.k, 0, w.qk_bytes, s));
  HIP_TRY(hipMemsetAsync(w.vt, 0, w.vt_bytes, s));
  if (w.zmx.hi) {      // the K pad of the MX3 residual rows (D = 576: columns 576 .. 639) is never written either; scale byte 0 = 2^-127
    HIP_TRY(hipMemsetAsync(w.zmx.hi, 0, w.zmx_rows * w.zmx.Kp * 2, s));
    HIP_TRY(hipMemsetAsync(w.zmx.l8, 0, w.zmx_rows * w.zmx.Kp, s));
    HIP_TRY(hipMemsetAsync(w.zmx.sc, 0, w.zmx_rows * (w.zmx.Kp / 32), s));
  }
  return 0;
}
// one pre-LN block on z (cells*T rows): z += proj(attn(LN1 z)); z += fc2(gelu(fc1(LN2 z)))
void run_block(const BlockW& L, const BlockWs& w, int cells, const AttnGeom& a, hipStream_t s) {
  const int D = a.D, Dp = round_up(D, 32), ld_x = 2 * Dp, ld_h = 2 * 4 * D, Mc = cells * a.T;
  const float scale = 1.0f / sqrtf((float)a.hd);
  { ProfScope ps(P_LN, s); launch_layernorm_ps(w.z, D, L.ln1w, L.ln1b, w.xa, ld_x, Mc, D, s); }
  {
    ProfScope ps(P_QKV, s);
    GemmArgs g{w.xa, ld_x, L.qkvw, ld_x, Mc, 3 * D, Dp, L.qkvb};
    launch_gemm_qkv(g, w.q, w.k, w.vt, a, scale, s);
  }
  { ProfScope ps(P_ATTN, s); launch_attention(w.q, w.k, w.vt, w.xa, ld_x, cells, a, s); }
  {
    ProfScope ps(P_PROJ, s);
    GemmArgs g{w.xa, ld_x, L.projw, ld_x, Mc, D, Dp, L.projb};
    launch_gemm_resid(g, w.z, D, s);
  }
  { ProfScope ps(P_LN, s); launch_layernorm_ps(w.z, D, L.ln2w, L.ln2b, w.xa, ld_x, Mc, D, s); }
  {
    ProfScope ps(P_FC1, s);
    GemmArgs g{w.xa, ld_x, L.fc1w, ld_x, Mc, 4 * D, Dp, L.fc1b, L.fc1wf};
    launch_gemm_gelu(g, w.h, ld_h, s);
  }
  {
    ProfScope ps(P_FC2, s);
    GemmArgs g{w.h, ld_h, L.fc2w, ld_h, Mc, D, 4 * D, L.fc2b};
    launch_gemm_resid(g, w.z, D, s);
  }
}

// Last block of a classifier: only the CLS row reaches the head (reference model.py:61-62 takes x[:, 0] after the final norm),
// so after the attention (which still needs every token's K and V) the projection, the MLP and both residual updates run
// on the CLS rows only: they are addressed in place with a row stride of T rows (M = cells).  Same values as the full block
// on those rows; the other 100 rows of the last block are never read again.
void run_last_block_cls(const BlockW& L, const BlockWs& w, int cells, const AttnGeom& a, hipStream_t s) {
  const int D = a.D, Dp = round_up(D, 32), ld_x = 2 * Dp, ld_h = 2 * 4 * D, Mc = cells * a.T;
  const float scale = 1.0f / sqrtf((float)a.hd);
  { ProfScope ps(P_LN, s); launch_layernorm_ps(w.z, D, L.ln1w, L.ln1b, w.xa, ld_x, Mc, D, s); }
  {
    ProfScope ps(P_QKV, s);
    GemmArgs g{w.xa, ld_x, L.qkvw, ld_x, Mc, 3 * D, Dp, L.qkvb};
    launch_gemm_qkv(g, w.q, w.k, w.vt, a, scale, s);
  }
  { ProfScope ps(P_ATTN, s); launch_attention(w.q, w.k, w.vt, w.xa, ld_x, cells, a, s, 1); }
  {
    ProfScope ps(P_PROJ, s);
    GemmArgs g{w.xa, a.T * ld_x, L.projw, ld_x, cells, D, Dp, L.projb};
    launch_gemm_resid(g, w.z, a.T * D, s);
  }
  { ProfScope ps(P_LN, s); launch_layernorm_ps(w.z, a.T * D, L.ln2w, L.ln2b, w.xa, a.T * ld_x, cells, D, s); }
  {
    ProfScope ps(P_FC1, s);
    GemmArgs g{w.xa, a.T * ld_x, L.fc1w, ld_x, cells, 4 * D, Dp, L.fc1b};
    launch_gemm_gelu(g, w.h, ld_h, s);
  }
  {
    ProfScope ps(P_FC2, s);
    GemmArgs g{w.h, ld_h, L.fc2w, ld_h, cells, D, 4 * D, L.fc2b};
    launch_gemm_resid(g, w.z, a.T * D, s);
  }
}

// ---- the classifiers' blocks with LayerNorm folded into the GEMM behind it (gemm_epi.h).  Five launches + two finalisers per block
// instead of seven + no LayerNorm pass over the rows: qkv and fc1 read the packed-split residual stream itself, the residual GEMMs
// (proj, fc2) update it in place and leave the row statistics of the NEW rows behind.
// Precondition: w.rs holds the statistics of z for norm1 (row_stats after the embedding, or the previous block's fc2).
// prev_stride: w.rs[m * prev_stride] = (rstd, mean) of the stored row that GEMM row m updates (re-centring, EpiResidPS)
void resid_ps_and_stats(const GemmArgs& g, const BlockWs& w, int ldz_rows, int D, bool want_stats, int prev_stride, hipStream_t s) {
  const ResidStatGeom sg = launch_gemm_resid_ps(g, w.zps, ldz_rows, want_stats ? w.part : nullptr, w.rs, prev_stride, s);
  if (want_stats) launch_ln_finalize(w.part, sg.tiles, g.M, sg.bn, D, w.rs, s);
}
// norm1 -> qkv -> attention of a whole block in ONE per-cell kernel (cell_attention.hip) where the geometry allows (D = 144, 288, 384):
// 13.9 -> 14.4 k cells/s in a same-box A/B (profiles/r3/ab_cell_attention.txt).  RIBCA_CELL_ATTN=0: the unfused pair, for A/B.
bool cell_attn_on(const AttnGeom& a) {
  static const int v = getenv("RIBCA_CELL_ATTN") ? atoi(getenv("RIBCA_CELL_ATTN")) : 1;
  return v != 0 && cell_attention_supported(a.D, a.H, a.T);
}
// next_full: the block after this one is a full block too (its qkv reads what this block's fc2 writes)
void run_block_fold(const BlockW& L, const BlockWs& w, int cells, const AttnGeom& a, hipStream_t s, bool precise = false, bool next_full = true) {
  const int D = a.D, Dp = round_up(D, 32), ld_x = 2 * Dp, ld_h = 2 * 4 * D, Mc = cells * a.T;
  const float scale = 1.0f / sqrtf((float)a.hd);
  // the residual rows in MX3 beside the packed-split ones: qkv (where it is a GEMM of its own) and fc1 run on the MX kernel
  const bool mz = w.zmx.hi != nullptr && L.fc1mxh != nullptr && !precise;
  MxAct zmx = w.zmx;
  zmx.M = Mc;
  const bool fused_attn = cell_attn_on(a);
  if (fused_attn) {
    ProfScope ps(P_CELL, s);
    launch_cell_qkv_attention(w.zps, ld_x, L.qkvw, ld_x, L.qkvb2, L.qkvc, w.rs, w.xa, ld_x, cells, D, scale, s);
  } else {
    {
      ProfScope ps(P_QKV, s);
      if (mz) {
        launch_gemm_mx_qkv_ln(zmx, MxWeight{L.qkvmxh, L.qkvmxx}, Mc, 3 * D, L.qkvb2, w.rs, L.qkvc, w.q, w.k, w.vt, a, scale, s);
      } else {
        GemmArgs g{w.zps, ld_x, L.qkvw, ld_x, Mc, 3 * D, Dp, L.qkvb2, L.qkvwf};
        launch_gemm_qkv_ln(g, w.rs, L.qkvc, w.q, w.k, w.vt, a, scale, s);
      }
    }
    { ProfScope ps(P_ATTN, s); launch_attention(w.q, w.k, w.vt, w.xa, ld_x, cells, a, s); }
  }
  {
    ProfScope ps(P_PROJ, s);
    GemmArgs g{w.xa, ld_x, L.projw, ld_x, Mc, D, Dp, L.projb, L.projwf};
    if (mz) {
      const ResidStatGeom sg = launch_gemm_resid_ps(g, w.zps, ld_x, w.part, w.rs, 1, s, false, &zmx);
      launch_ln_finalize(w.part, sg.tiles, Mc, sg.bn, D, w.rs, s);
    } else {
      resid_ps_and_stats(g, w, ld_x, D, true, 1, s);
    }
  }
  if (L.fc2mxh != nullptr && !precise) {
    // the MX pair: fc1's GELU epilogue emits the three-plane operand (3 bytes per element, carved out of the h buffer), fc2 multiplies it
    // as fp16 hi * hi + two block-scaled corrections (gemm_mx.hip)
    const size_t hn = (size_t)Mc * 4 * D;
    const MxAct hmx{w.h, reinterpret_cast<unsigned char*>(w.h + hn), reinterpret_cast<unsigned char*>(w.h + hn) + hn, 4 * D, Mc};
    {
      ProfScope ps(P_FC1, s);
      if (mz) {
        launch_gemm_mx_gelu(zmx, MxWeight{L.fc1mxh, L.fc1mxx}, Mc, 4 * D, L.fc1b2, w.rs, L.fc1c, hmx, s);
      } else {
        GemmArgs g{w.zps, ld_x, L.fc1w, ld_x, Mc, 4 * D, Dp, L.fc1b2, L.fc1wf};
        (void)launch_gemm_gelu_mx(g, w.rs, L.fc1c, hmx, s);
      }
    }
    {
      ProfScope ps(P_FC2, s);
      // (the MX3 copy of the new rows is read by the next block's qkv GEMM: not wanted where that runs inside the fused per-cell kernel, which
      // reads the packed-split rows, nor in front of the last block)
      const bool emit = mz && !fused_attn && next_full;
      const ResidStatGeom sg = launch_gemm_mx_resid(hmx, MxWeight{L.fc2mxh, L.fc2mxx}, Mc, D, L.fc2b, w.zps, ld_x, w.part, w.rs, 1, s, 0, emit ? &zmx : nullptr);
      launch_ln_finalize(w.part, sg.tiles, Mc, sg.bn, D, w.rs, s);
    }
    return;
  }
  {
    ProfScope ps(P_FC1, s);
    GemmArgs g{w.zps, ld_x, L.fc1w, ld_x, Mc, 4 * D, Dp, L.fc1b2, L.fc1wf};
    launch_gemm_gelu_ln(g, w.rs, L.fc1c, w.h, ld_h, s);
  }
  {
    ProfScope ps(P_FC2, s);
    GemmArgs g{w.h, ld_h, L.fc2w, ld_h, Mc, D, 4 * D, L.fc2b, L.fc2wf};
    resid_ps_and_stats(g, w, ld_x, D, true, 1, s);
  }
}
// last block, CLS rows only behind the attention (see run_last_block_cls): GEMM row m = cell, addressed with a row stride of T rows;
// the statistics of norm2 are indexed by GEMM row too.  Nothing reads statistics after the last fc2 (the head normalises itself).
void run_last_block_cls_fold(const BlockW& L, const BlockWs& w, int cells, const AttnGeom& a, hipStream_t s) {
  const int D = a.D, Dp = round_up(D, 32), ld_x = 2 * Dp, ld_h = 2 * 4 * D, Mc = cells * a.T;
  const float scale = 1.0f / sqrtf((float)a.hd);
  {
    // only the CLS query is used (launch_attention below: one query tile, and only its row 0 is read afterwards): K and V for every
    // token (output columns D .. 3D of the product), Q for the CLS rows alone -- a third of this block's qkv work is never done.
    // (Rows 1 .. 15 of the first Q tile keep the previous block's values: finite, feeding output rows nothing reads.)
    ProfScope ps(P_QKV, s);
    GemmArgs gkv{w.zps, ld_x, L.qkvw + (size_t)D * ld_x, ld_x, Mc, 2 * D, Dp, L.qkvb2 + D};
    launch_gemm_qkv_ln(gkv, w.rs, L.qkvc + D, w.q, w.k, w.vt, a, scale, s, D, 0, 1);
    GemmArgs gq{w.zps, a.T * ld_x, L.qkvw, ld_x, cells, D, Dp, L.qkvb2};
    launch_gemm_qkv_ln(gq, w.rs, L.qkvc, w.q, w.k, w.vt, a, scale, s, 0, 1, a.T);
  }
  { ProfScope ps(P_ATTN, s); launch_attention(w.q, w.k, w.vt, w.xa, ld_x, cells, a, s, 1); }
  {
    ProfScope ps(P_PROJ, s);
    GemmArgs g{w.xa, a.T * ld_x, L.projw, ld_x, cells, D, Dp, L.projb};
    resid_ps_and_stats(g, w, a.T * ld_x, D, true, a.T, s);      // the stored CLS rows still carry norm1's (all-rows) statistics
  }
  {
    ProfScope ps(P_FC1, s);
    GemmArgs g{w.zps, a.T * ld_x, L.fc1w, ld_x, cells, 4 * D, Dp, L.fc1b2};
    launch_gemm_gelu_ln(g, w.rs, L.fc1c, w.h, ld_h, s);
  }
  {
    ProfScope ps(P_FC2, s);
    GemmArgs g{w.h, ld_h, L.fc2w, ld_h, cells, D, 4 * D, L.fc2b};
    resid_ps_and_stats(g, w, a.T * ld_x, D, false, 1, s);
  }
}

}  // namespace

extern "C" {

int ribca_version(void) { return 100; }
const char* ribca_last_error(void) { return api_last_error(); }

// the launcher table of libribca_hip_test.so (csrc/ribca_internal.h): the only way into the library besides the C entry points
const void* ribca_internal_table(int32_t version) {
  static const InternalTable table = {RIBCA_INTERNAL_VERSION,
#define RIBCA_X(name) +1
                                      0 RIBCA_INTERNAL_FUNCS(RIBCA_X),
#undef RIBCA_X
#define RIBCA_X(name) &ribca::name,
                                      RIBCA_INTERNAL_FUNCS(RIBCA_X)
#undef RIBCA_X
  };
  return version == RIBCA_INTERNAL_VERSION ? &table : nullptr;
}

int32_t ribca_mx_enabled(int32_t D) { return mx_on(D) ? 1 : 0; }
int32_t ribca_mxz_enabled(int32_t D) { return mx_z_on(D) ? 1 : 0; }

int64_t ribca_vit_blob_len(int32_t D, int32_t C, int32_t K, int32_t depth) {
  const int64_t d = D;
  return d + (int64_t)kTokens * d + d * C * 16 + d + (int64_t)depth * (2 * d + 3 * d * d + 3 * d + d * d + d + 2 * d + 4 * d * d + 4 * d + 4 * d * d + d) +
         2 * d + (int64_t)K * d + K;
}

int ribca_vit_create(const float* blob, int64_t blob_len, int32_t D, int32_t C, int32_t K, int32_t depth, void* stream, ribca_vit_t** out) {
  if (!out) return fail("ribca_vit_create: out is NULL");
  *out = nullptr;
  if (D <= 0 || D % 48 != 0 || D > 768) return fail("ribca_vit_create: D must be a multiple of 48 and <= 768");
  if (C <= 0 || C > 64 || K <= 0 || K > 16 || depth <= 0) return fail("ribca_vit_create: bad C/K/depth");
  if (!attention_supported(make_attn_geom(D, kHeads, kTokens))) return fail("ribca_vit_create: unsupported head dimension");
  if (blob_len != ribca_vit_blob_len(D, C, K, depth)) return fail("ribca_vit_create: blob length does not match (D, C, K, depth)");
  hipStream_t s = (hipStream_t)stream;
  ribca_vit* m = new ribca_vit();
  m->D = D; m->C = C; m->K = K; m->depth = depth;
  m->hd = D / kHeads;
  m->hdp = round_up(m->hd, 32);
  m->hdv = round_up(m->hd, 16);
  m->Dp = round_up(D, 32);
  m->H4 = 4 * D;  // multiple of 32 because D % 8 == 0
  m->fold = !(getenv("RIBCA_LN_FOLD") && atoi(getenv("RIBCA_LN_FOLD")) == 0);
  m->arena_bytes = layout(m, nullptr);
  hipError_t e = hipMalloc((void**)&m->arena, m->arena_bytes);
  if (e != hipSuccess) { delete m; return hip_fail(e, "hipMalloc(weights)"); }
  layout(m, m->arena);
  BlobReader r{blob, s};
  r.copy(m->cls, D);
  r.copy(m->pos, (size_t)kTokens * D);
  r.copy(m->pe_w, (size_t)D * 16 * C);
  r.copy(m->pe_b, D);
  for (auto& L : m->layers) r.block(L, D, m->fold);
  r.copy(m->norm_w, D); r.copy(m->norm_b, D);
  r.copy(m->head_w, (size_t)K * D); r.copy(m->head_b, K);
  e = r.err != hipSuccess ? r.err : hipGetLastError();
  if (e != hipSuccess) { ribca_vit_destroy(m); return hip_fail(e, "weight packing"); }
  if (api_finish() != 0) { ribca_vit_destroy(m); return 1; }
  *out = m;
  return 0;
}

void ribca_vit_destroy(ribca_vit_t* m) {
  if (!m) return;
  if (m->arena) (void)hipFree(m->arena);
  delete m;
}

double ribca_vit_flops_per_cell(const ribca_vit_t* m) {
  if (!m) return 0.0;
  const double d = m->D, n = kTokens;
  return 2.0 * 100 * 16 * m->C * d + m->depth * (24.0 * n * d * d + 4.0 * n * n * d) + 2.0 * d * m->K;
}

int64_t ribca_vit_workspace_bytes(const ribca_vit_t* m, int32_t chunk_cells) {
  if (!m || chunk_cells <= 0) return 0;
  Carver c(nullptr);
  carve_blocks(c, chunk_cells, make_attn_geom(m->D, kHeads, kTokens), m->fold);
  return (int64_t)c.off;
}

static int vit_forward_impl(const ribca_vit_t* m, const float* patches, int32_t c_img, const int32_t* src_chan, int32_t n_cells, float* probs,
                            void* workspace, int64_t workspace_bytes, int32_t chunk_cells, void* stream, bool precise);
int ribca_vit_forward(const ribca_vit_t* m, const float* patches, int32_t c_img, const int32_t* src_chan, int32_t n_cells, float* probs,
                      void* workspace, int64_t workspace_bytes, int32_t chunk_cells, void* stream) {
  return vit_forward_impl(m, patches, c_img, src_chan, n_cells, probs, workspace, workspace_bytes, chunk_cells, stream, false);
}
int ribca_vit_forward_precise(const ribca_vit_t* m, const float* patches, int32_t c_img, const int32_t* src_chan, int32_t n_cells, float* probs,
                              void* workspace, int64_t workspace_bytes, int32_t chunk_cells, void* stream) {
  return vit_forward_impl(m, patches, c_img, src_chan, n_cells, probs, workspace, workspace_bytes, chunk_cells, stream, true);
}
static int vit_forward_impl(const ribca_vit_t* m, const float* patches, int32_t c_img, const int32_t* src_chan, int32_t n_cells, float* probs,
                            void* workspace, int64_t workspace_bytes, int32_t chunk_cells, void* stream, bool precise) {
  if (!m) return fail("ribca_vit_forward: model is NULL");
  if (n_cells < 0 || chunk_cells <= 0) return fail("ribca_vit_forward: bad cell counts");
  if (n_cells == 0) return 0;
  if (!patches || !src_chan || !probs || !workspace) return fail("ribca_vit_forward: NULL buffer");
  if (c_img <= 0) return fail("ribca_vit_forward: c_img (channels per patch in `patches`) must be positive");
  if (((uintptr_t)workspace & 255) != 0) return fail("ribca_vit_forward: workspace must be 256-byte aligned");
  const AttnGeom geom = make_attn_geom(m->D, kHeads, kTokens);
  Carver c((char*)workspace);
  const BlockWs w = carve_blocks(c, chunk_cells, geom, m->fold);
  if ((int64_t)c.off > workspace_bytes) return fail("ribca_vit_forward: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int D = m->D;
  {
    ProfScope ps(P_OTHER, s);
    if (zero_pads(w, s)) return 1;
  }
  for (int c0 = 0; c0 < n_cells; c0 += chunk_cells) {
    const int bc = n_cells - c0 < chunk_cells ? n_cells - c0 : chunk_cells;
    if (m->fold) {
      const int ld_z = 2 * m->Dp;
      {
        ProfScope ps(P_EMBED, s);
        launch_embed_ps(patches + (size_t)c0 * c_img * 1600, c_img, src_chan, m->C, m->pe_w, m->pe_b, m->pos, w.zps, ld_z, D, bc, s);
      }
      {
        ProfScope ps(P_OTHER, s);
        launch_cls_rows_ps(w.zps, ld_z, m->cls, m->pos, D, bc, kTokens, s);
      }
      { ProfScope ps(P_LN, s); launch_row_stats_ps(w.zps, ld_z, bc * kTokens, D, w.rs, true, s); }
      if (w.zmx.hi != nullptr && !precise && !cell_attn_on(geom) && m->layers.size() > 1) {      // the first block's qkv operand
        ProfScope ps(P_OTHER, s);
        MxAct zmx = w.zmx;
        zmx.M = bc * kTokens;
        launch_mx_pack_act(w.zps, ld_z, bc * kTokens, m->Dp, zmx, s);
      }
      for (size_t li = 0; li + 1 < m->layers.size(); ++li) run_block_fold(m->layers[li], w, bc, geom, s, precise, li + 2 < m->layers.size());
      run_last_block_cls_fold(m->layers.back(), w, bc, geom, s);
      {
        ProfScope ps(P_HEAD, s);
        launch_head_softmax_ps(w.zps, ld_z, m->norm_w, m->norm_b, m->head_w, m->head_b, probs + (size_t)c0 * m->K, D, m->K, bc, s);
      }
      continue;
    }
    {
      ProfScope ps(P_EMBED, s);
      launch_embed_f32(patches + (size_t)c0 * c_img * 1600, c_img, src_chan, m->C, m->pe_w, m->pe_b, m->pos, w.z, D, D, bc, s);
    }
    {
      ProfScope ps(P_OTHER, s);
      launch_cls_rows(w.z, D, m->cls, m->pos, D, bc, kTokens, s);
    }
    for (size_t li = 0; li + 1 < m->layers.size(); ++li) run_block(m->layers[li], w, bc, geom, s);
    run_last_block_cls(m->layers.back(), w, bc, geom, s);
    {
      ProfScope ps(P_HEAD, s);
      launch_head_softmax(w.z, D, m->norm_w, m->norm_b, m->head_w, m->head_b, probs + (size_t)c0 * m->K, D, m->K, bc, s);
    }
  }
  RIBCA_FINISH();
  return 0;
}

// ------------------------------------------------------------------------------------------- marker imputer
int64_t ribca_mae_blob_len(int32_t L, int32_t enc_depth, int32_t dec_depth) {
  const int64_t e = kEncD, d = kDecD;
  return e + (int64_t)(L + 1) * e + e * kTokPix + e + enc_depth * block_params(e) + 2 * e + d * e + d + d + (int64_t)(L + 1) * d +
         dec_depth * block_params(d) + 2 * d + (int64_t)kTokPix * d + kTokPix;
}

int ribca_mae_create(const float* blob, int64_t blob_len, int32_t L, int32_t enc_depth, int32_t dec_depth, void* stream, ribca_mae_t** out) {
  if (!out) return fail("ribca_mae_create: out is NULL");
  *out = nullptr;
  if (L < 2 || L > 15 || enc_depth <= 0 || dec_depth <= 0) return fail("ribca_mae_create: L must be in [2, 15] (tokens incl. CLS <= 16)");
  if (blob_len != ribca_mae_blob_len(L, enc_depth, dec_depth)) return fail("ribca_mae_create: blob length does not match (L, depths)");
  hipStream_t s = (hipStream_t)stream;
  ribca_mae* m = new ribca_mae();
  m->L = L; m->enc_depth = enc_depth; m->dec_depth = dec_depth;
  m->fold = !(getenv("RIBCA_MAE_FOLD") && atoi(getenv("RIBCA_MAE_FOLD")) == 0);
  m->arena_bytes = layout_mae(m, nullptr);
  hipError_t e = hipMalloc((void**)&m->arena, m->arena_bytes);
  if (e != hipSuccess) { delete m; return hip_fail(e, "hipMalloc(imputer weights)"); }
  layout_mae(m, m->arena);
  BlobReader r{blob, s};
  r.copy(m->cls, kEncD);
  r.copy(m->pos, (size_t)(L + 1) * kEncD);
  r.pack(m->pe_w, kEncD, kTokPix, kTokPix);
  r.copy(m->pe_b, kEncD);
  for (auto& B : m->enc) r.block(B, kEncD, m->fold);
  r.copy(m->norm_w, kEncD); r.copy(m->norm_b, kEncD);
  r.pack(m->de_w, kDecD, kEncD, kEncD);
  r.copy(m->de_b, kDecD);
  r.copy(m->mask_tok, kDecD);
  r.copy(m->dpos, (size_t)(L + 1) * kDecD);
  for (auto& B : m->dec) r.block(B, kDecD, m->fold);
  r.copy(m->dnorm_w, kDecD); r.copy(m->dnorm_b, kDecD);
  r.pack(m->pred_w, kTokPix, kDecD, kDecD);
  r.copy(m->pred_b, kTokPix);
  e = r.err != hipSuccess ? r.err : hipGetLastError();
  if (e != hipSuccess) { ribca_mae_destroy(m); return hip_fail(e, "imputer weight packing"); }
  if (api_finish() != 0) { ribca_mae_destroy(m); return 1; }
  *out = m;
  return 0;
}

void ribca_mae_destroy(ribca_mae_t* m) {
  if (!m) return;
  if (m->arena) (void)hipFree(m->arena);
  delete m;
}

namespace {
struct MaeWs {
  BlockWs enc, dec;
  uint16_t* tok_ps;     // present channel tiles as packed-split rows [cells*P][2*1600]; reused for decoder_norm rows [cells*Mi][2*512]
  int* tables;          // device int tables (kMaeTables x 16)
  // folded path: the token rows are assembled in fp32 (embedding GEMM through its row map, cls / mask rows, positional embeddings) and then
  // split once into the packed-split residual stream the blocks run on
  float *enc_zf = nullptr, *dec_zf = nullptr;
  size_t total;
};
constexpr int kMaeTables = 7;
MaeWs carve_mae(const ribca_mae* m, int chunk, int P, char* base) {
  Carver c(base);
  MaeWs w;
  w.enc = carve_blocks(c, chunk, make_attn_geom(kEncD, kEncH, P + 1), m->fold);
  w.dec = carve_blocks(c, chunk, make_attn_geom(kDecD, kDecH, m->L + 1), m->fold);
  if (m->fold) {
    w.enc_zf = c.take<float>((size_t)chunk * (P + 1) * kEncD);
    w.dec_zf = c.take<float>((size_t)chunk * (m->L + 1) * kDecD);
  }
  w.tok_ps = c.take<uint16_t>((size_t)chunk * m->L * 2 * kTokPix);
  w.tables = c.take<int>(kMaeTables * 16);
  w.total = c.off;
  return w;
}
}  // namespace

int64_t ribca_mae_workspace_bytes(const ribca_mae_t* m, int32_t chunk_cells, int32_t n_present) {
  if (!m || chunk_cells <= 0 || n_present <= 0 || n_present >= m->L) return 0;
  return (int64_t)carve_mae(m, chunk_cells, n_present, nullptr).total;
}

int ribca_mae_impute(const ribca_mae_t* m, float* patches, const int32_t* present_host, int32_t n_present, int32_t n_cells, void* workspace,
                     int64_t workspace_bytes, int32_t chunk_cells, void* stream) {
  if (!m) return fail("ribca_mae_impute: model is NULL");
  if (n_cells < 0 || chunk_cells <= 0) return fail("ribca_mae_impute: bad cell counts");
  const int L = m->L, P = n_present, Mi = L - P;
  if (!present_host || P <= 0 || P >= L) return fail("ribca_mae_impute: need 1 <= n_present < L");
  if (n_cells == 0) return 0;
  if (!patches || !workspace) return fail("ribca_mae_impute: NULL buffer");
  if (((uintptr_t)workspace & 255) != 0) return fail("ribca_mae_impute: workspace must be 256-byte aligned");
  // host tables: [0] present, [1] missing, [2] enc slot (1+j), [3] enc pos row (1+present[j]), [4] dec slot/pos (0, 1+present[..]),
  // [5] dec mask rows (1+missing[j])
  int tab[kMaeTables][16] = {};      // [6]: identity (every token row of a cell, in order)
  bool seen[16] = {};
  for (int j = 0; j < 16; ++j) tab[6][j] = j;
  for (int j = 0; j < P; ++j) {
    const int c = present_host[j];
    if (c < 0 || c >= L || seen[c] || (j > 0 && c <= present_host[j - 1])) return fail("ribca_mae_impute: present must be strictly increasing in [0, L)");
    seen[c] = true;
    tab[0][j] = c; tab[2][j] = 1 + j; tab[3][j] = 1 + c; tab[4][1 + j] = 1 + c;
  }
  for (int c = 0, j = 0; c < L; ++c)
    if (!seen[c]) { tab[1][j] = c; tab[5][j] = 1 + c; ++j; }
  const MaeWs w = carve_mae(m, chunk_cells, P, (char*)workspace);
  if ((int64_t)w.total > workspace_bytes) return fail("ribca_mae_impute: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemcpyAsync(w.tables, tab, sizeof(tab), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));   // `tab` lives on this stack frame
  const int *t_present = w.tables, *t_missing = w.tables + 16, *t_eslot = w.tables + 32, *t_epos = w.tables + 48, *t_dslot = w.tables + 64,
            *t_dmask = w.tables + 80, *t_all = w.tables + 96;
  const AttnGeom ge = make_attn_geom(kEncD, kEncH, P + 1), gd = make_attn_geom(kDecD, kDecH, L + 1);
  if (zero_pads(w.enc, s) || zero_pads(w.dec, s)) return 1;
  const bool fold = m->fold;
  // fp32 token rows -> the packed-split residual stream of a run of folded blocks + the statistics its first LayerNorm reads (+ the MX3
  // copy where that width's qkv runs on the MX kernel): what vit_forward_impl does behind the patch embedding
  auto enter_fold = [&](const float* zf, const BlockWs& bw, int cells, const AttnGeom& a, size_t depth) {
    const int D = a.D, ld = 2 * round_up(D, 32), Mc = cells * a.T;
    launch_rows_to_ps(zf, D, bw.zps, ld, round_up(D, 32), cells, a.T, a.T, t_all, s);
    launch_row_stats_ps(bw.zps, ld, Mc, D, bw.rs, true, s);
    if (bw.zmx.hi != nullptr && !cell_attn_on(a) && depth > 0) {
      MxAct zmx = bw.zmx;
      zmx.M = Mc;
      launch_mx_pack_act(bw.zps, ld, Mc, round_up(D, 32), zmx, s);
    }
  };
  for (int c0 = 0; c0 < n_cells; c0 += chunk_cells) {
    const int bc = n_cells - c0 < chunk_cells ? n_cells - c0 : chunk_cells;
    float* pch = patches + (size_t)c0 * L * kTokPix;
    float* ez = fold ? w.enc_zf : w.enc.z;
    float* dz = fold ? w.dec_zf : w.dec.z;
    // encoder input: embed the present channel tiles (markerImputer.py:186-199)
    launch_rows_to_ps(pch, kTokPix, w.tok_ps, 2 * kTokPix, kTokPix, bc, L, P, t_present, s);
    {
      GemmArgs g{w.tok_ps, 2 * kTokPix, m->pe_w, 2 * kTokPix, bc * P, kEncD, kTokPix, m->pe_b};
      launch_gemm_rowmap(g, ez, kEncD, m->pos, kEncD, t_eslot, t_epos, P, P + 1, s);
    }
    launch_cls_rows(ez, kEncD, m->cls, m->pos, kEncD, bc, P + 1, s);
    if (fold) {
      enter_fold(ez, w.enc, bc, ge, m->enc.size());
      for (size_t li = 0; li < m->enc.size(); ++li) run_block_fold(m->enc[li], w.enc, bc, ge, s, false, li + 1 < m->enc.size());
      // decoder input (markerImputer.py:208-219): final encoder norm (its own statistics from the packed-split rows), project latents
      launch_layernorm_gather_ps_from_ps(w.enc.zps, 2 * kEncD, m->norm_w, m->norm_b, w.enc.xa, 2 * kEncD, bc, P + 1, P + 1, t_all, kEncD, s);
    } else {
      for (const auto& B : m->enc) run_block(B, w.enc, bc, ge, s);
      launch_layernorm_ps(w.enc.z, kEncD, m->norm_w, m->norm_b, w.enc.xa, 2 * kEncD, bc * (P + 1), kEncD, s);
    }
    {
      GemmArgs g{w.enc.xa, 2 * kEncD, m->de_w, 2 * kEncD, bc * (P + 1), kDecD, kEncD, m->de_b};
      launch_gemm_rowmap(g, dz, kDecD, m->dpos, kDecD, t_dslot, t_dslot, P + 1, L + 1, s);
    }
    // mask tokens at the missing positions, + decoder pos
    launch_fill_rows(dz, kDecD, m->mask_tok, m->dpos, kDecD, bc, L + 1, Mi, t_dmask, s);
    // predict only the missing channels and write them into the patch tensor (blend, markerImputer.py:312-326)
    if (fold) {
      enter_fold(dz, w.dec, bc, gd, m->dec.size());
      for (size_t li = 0; li < m->dec.size(); ++li) run_block_fold(m->dec[li], w.dec, bc, gd, s, false, li + 1 < m->dec.size());
      launch_layernorm_gather_ps_from_ps(w.dec.zps, 2 * kDecD, m->dnorm_w, m->dnorm_b, w.tok_ps, 2 * kDecD, bc, L + 1, Mi, t_dmask, kDecD, s);
    } else {
      for (const auto& B : m->dec) run_block(B, w.dec, bc, gd, s);
      launch_layernorm_gather_ps(w.dec.z, kDecD, m->dnorm_w, m->dnorm_b, w.tok_ps, 2 * kDecD, bc, L + 1, Mi, t_dmask, kDecD, s);
    }
    {
      GemmArgs g{w.tok_ps, 2 * kDecD, m->pred_w, 2 * kDecD, bc * Mi, kTokPix, kDecD, m->pred_b};
      launch_gemm_rowmap(g, pch, kTokPix, nullptr, 0, t_missing, t_missing, Mi, L, s);
    }
  }
  RIBCA_FINISH();
  return 0;
}

// ------------------------------------------------------------------------------------------- pre-processing
int ribca_mask_minmax(const int32_t* mask, int64_t n, int32_t* out2, void* stream) {
  if (!out2 || (n > 0 && !mask)) return fail("ribca_mask_minmax: NULL buffer");
  launch_mask_max(mask, n, out2, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}

int ribca_label_table(const int32_t* mask, int32_t H, int32_t W, int32_t L, int32_t* tab_i32, uint64_t* tab_u64, void* stream) {
  if (!mask || !tab_i32 || !tab_u64) return fail("ribca_label_table: NULL buffer");
  if (H <= 0 || W <= 0 || L <= 0) return fail("ribca_label_table: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  launch_label_table_init(tab_i32, (unsigned long long*)tab_u64, L, s);
  launch_label_table(mask, H, W, L, tab_i32, (unsigned long long*)tab_u64, s);
  RIBCA_FINISH();
  return 0;
}

int ribca_channel_min(const float* image, int32_t C, int64_t hw, float* out_min, void* stream) {
  if (!image || !out_min) return fail("ribca_channel_min: NULL buffer");
  if (C <= 0 || C > 64) return fail("ribca_channel_min: C must be in [1, 64]");
  launch_channel_min(image, C, hw, out_min, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}

int ribca_extract_patches(const float* image, int32_t C, int32_t H, int32_t W, const int32_t* mask, const float* chan_min,
                          const int32_t* cell_id, const int32_t* bbox, const double* taps, int32_t n, float* patches, double* avg,
                          void* stream) {
  if (n == 0) return 0;
  if (!image || !mask || !chan_min || !cell_id || !bbox || !taps || !patches) return fail("ribca_extract_patches: NULL buffer");
  if (C <= 0 || H <= 0 || W <= 0 || n < 0) return fail("ribca_extract_patches: bad sizes");
  PatchArgs a{image, C, H, W, mask, chan_min, cell_id, bbox, taps, patches, avg, n};
  launch_extract_patches(a, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}

int ribca_extract_patches_scaled(const float* image, int32_t C, int32_t H, int32_t W, const int32_t* mask, const float* chan_min,
                                 const int32_t* cell_id, const int32_t* bbox, const double* taps, int32_t n, int32_t patch_size,
                                 const double* aa_taps, int32_t aa_radius, const int32_t* src_index, float* patches, double* avg,
                                 void* stream) {
  if (n == 0) return 0;
  if (!image || !mask || !chan_min || !cell_id || !bbox || !taps || !patches || !src_index) return fail("ribca_extract_patches_scaled: NULL buffer");
  if (aa_radius > 0 && !aa_taps) return fail("ribca_extract_patches_scaled: aa_taps is NULL");
  if (C <= 0 || H <= 0 || W <= 0 || n < 0) return fail("ribca_extract_patches_scaled: bad sizes");
  PatchArgs a{image, C, H, W, mask, chan_min, cell_id, bbox, taps, patches, avg, n};
  if (launch_extract_patches_scaled(a, patch_size, aa_taps, aa_radius, src_index, (hipStream_t)stream) != 0)
    return fail("ribca_extract_patches_scaled: patch_size must be in [4, 90] (cell_size up to 67) and aa_radius in [0, 15]");
  RIBCA_FINISH();
  return 0;
}

int ribca_vote(const float* p_a, int32_t k_a, const int8_t* map_a, const float* p_b, int32_t k_b, const int8_t* map_b,
               const float* type_conf, float conf, int32_t n, int8_t* label, float* out_conf, void* stream) {
  if (n == 0) return 0;
  if (!p_a || !map_a || !type_conf || !label || !out_conf) return fail("ribca_vote: NULL buffer");
  if (p_b && !map_b) return fail("ribca_vote: map_b is NULL");
  if (k_a <= 0 || k_a > 16 || (p_b && (k_b <= 0 || k_b > 16))) return fail("ribca_vote: class counts must be in [1, 16]");
  VoteArgs a{p_a, k_a, map_a, p_b, k_b, map_b, type_conf, conf, n, label, out_conf};
  launch_vote(a, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}

int ribca_colorize(const int32_t* mask, int64_t n_pixels, const int32_t* label_to_cell, int32_t L, const uint8_t* cell_type_rgb,
                   const uint8_t* cell_conf_rgb, const uint8_t* cell_type_idx, uint8_t* out_type_rgb, uint8_t* out_conf_rgb,
                   uint8_t* out_type_idx, void* stream) {
  if (n_pixels == 0) return 0;
  if (!mask || !label_to_cell || !cell_type_rgb || !cell_conf_rgb || !cell_type_idx || !out_type_rgb || !out_conf_rgb || !out_type_idx)
    return fail("ribca_colorize: NULL buffer");
  if (n_pixels < 0 || L <= 0) return fail("ribca_colorize: bad sizes");
  launch_colorize(mask, n_pixels, label_to_cell, L, cell_type_rgb, cell_conf_rgb, cell_type_idx, out_type_rgb, out_conf_rgb, out_type_idx,
                  (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}

int ribca_knn_cooccurrence(const double* x, const double* y, const int32_t* cell_type, int32_t n_cells, int32_t n_neighbors, int32_t n_types,
                           uint64_t* matrix, void* stream) {
  if (!x || !y || !cell_type || !matrix) return fail("ribca_knn_cooccurrence: NULL buffer");
  if (n_cells <= 0) return fail("ribca_knn_cooccurrence: no cells");
  if (n_neighbors > n_cells) return fail("ribca_knn_cooccurrence: n_neighbors exceeds the number of cells");
  if (launch_knn_cooccurrence(x, y, cell_type, n_cells, n_neighbors, n_types, reinterpret_cast<unsigned long long*>(matrix), (hipStream_t)stream))
    return fail("ribca_knn_cooccurrence: n_neighbors must be in [1, 32] and n_types in [1, 32]");
  RIBCA_FINISH();
  return 0;
}

int ribca_knn_compositions(const double* x, const double* y, const int32_t* cell_type, int32_t n_cells, int32_t n_types, const int32_t* sizes,
                           int32_t n_sizes, uint16_t* counts, void* stream) {
  if (!x || !y || !cell_type || !sizes || !counts) return fail("ribca_knn_compositions: NULL buffer");
  if (n_cells <= 0 || n_sizes <= 0) return fail("ribca_knn_compositions: bad sizes");
  int32_t host[8];
  if (n_sizes > 8) return fail("ribca_knn_compositions: at most 8 neighbourhood sizes");
  HIP_TRY(hipMemcpyAsync(host, sizes, sizeof(int32_t) * n_sizes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  for (int i = 0; i < n_sizes; ++i)
    if (host[i] < 1 || (i > 0 && host[i] <= host[i - 1])) return fail("ribca_knn_compositions: sizes must be positive and strictly increasing");
  const int k = host[n_sizes - 1] + 1;
  if (k > n_cells) return fail("ribca_knn_compositions: more neighbours requested than cells");
  if (launch_knn_compositions(x, y, cell_type, n_cells, k, n_types, n_sizes, sizes, counts, (hipStream_t)stream))
    return fail("ribca_knn_compositions: needs max(sizes) <= 255 and n_types <= 32");
  RIBCA_FINISH();
  return 0;
}

// ------------------------------------------------------------------------------------------- normalisation primitives
int ribca_u16_to_f32(const uint16_t* in, float* out, int64_t n, void* stream) {
  if (n > 0 && (!in || !out)) return fail("ribca_u16_to_f32: NULL buffer");
  launch_u16_to_f32(in, out, n, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}
int ribca_gauss1d(const float* in, float* out, int32_t planes, int32_t H, int32_t W, int32_t axis, const double* taps, int32_t R,
                  int32_t mode, void* stream) {
  if (!in || !out || !taps) return fail("ribca_gauss1d: NULL buffer");
  if (in == out) return fail("ribca_gauss1d: in-place filtering is not supported");
  if (axis < 0 || axis > 1 || mode < 0 || mode > 1 || R < 0) return fail("ribca_gauss1d: bad axis/mode/radius");
  launch_gauss1d(in, out, planes, H, W, axis, taps, R, mode, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}
int ribca_bg_subtract(float* x, const float* bg, int64_t n, float cap, void* stream) {
  if (n > 0 && (!x || !bg)) return fail("ribca_bg_subtract: NULL buffer");
  launch_bg_subtract(x, bg, n, cap, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}
int ribca_plane_max(const float* x, int32_t planes, int64_t hw, float* out, void* stream) {
  if (!x || !out) return fail("ribca_plane_max: NULL buffer");
  launch_plane_max(x, planes, hw, out, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}
int ribca_radix_hist(const float* x, int32_t planes, int64_t hw, const uint32_t* prefix, uint32_t mask_hi, int32_t shift, int32_t bits,
                     uint32_t* hist, void* stream) {
  if (!x || !prefix || !hist) return fail("ribca_radix_hist: NULL buffer");
  if (bits < 1 || bits > 11 || shift < 0 || shift + bits > 32) return fail("ribca_radix_hist: bad shift/bits");
  launch_radix_hist(x, planes, hw, prefix, mask_hi, shift, bits, hist, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}
int ribca_norm_finalize(float* x, int32_t planes, int64_t hw, const int32_t* mode, const float* clip, const float* denom, void* stream) {
  if (!x || !mode || !clip || !denom) return fail("ribca_norm_finalize: NULL buffer");
  launch_norm_finalize(x, planes, hw, mode, clip, denom, (hipStream_t)stream);
  RIBCA_FINISH();
  return 0;
}

// ------------------------------------------------------------------------------------------- profiling
int ribca_prof_enable(int32_t on) {
  prof_drain();
  g_prof_on = on != 0;
  for (int i = 0; i < P_COUNT; ++i) { g_prof_ms[i] = 0; g_prof_n[i] = 0; }
  return 0;
}
int ribca_prof_read(double* ms_out10, int64_t* count_out10) {
  prof_drain();
  for (int i = 0; i < P_COUNT; ++i) {
    if (ms_out10) ms_out10[i] = g_prof_ms[i];
    if (count_out10) count_out10[i] = g_prof_n[i];
  }
  return 0;
}
const char* ribca_prof_name(int32_t cls) { return (cls >= 0 && cls < P_COUNT) ? kProfNames[cls] : ""; }

}  // extern "C"
