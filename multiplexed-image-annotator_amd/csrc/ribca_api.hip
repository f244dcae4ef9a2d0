// C ABI of libribca_hip.so (see include/ribca_hip.h): handle management, workspace carving and the launch sequence of
// the ViT forward.  No torch types; the caller (Python via ctypes) owns all buffers except the packed-weight handle.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ribca_hip.h"
#include "ribca_common.h"
#include "ribca_kernels.h"

using namespace ribca;

namespace {

thread_local std::string g_err;

int fail(const std::string& msg) {
  g_err = msg;
  return 1;
}
int hip_fail(hipError_t e, const char* what) { return fail(std::string(what) + ": " + hipGetErrorString(e)); }
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return hip_fail(e_, #expr);   \
  } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// ----------------------------------------------------------------------------------------------- profiling
enum ProfClass { P_QKV = 0, P_PROJ, P_FC1, P_FC2, P_EMBED, P_ATTN, P_LN, P_IM2COL, P_HEAD, P_OTHER, P_COUNT };
const char* kProfNames[P_COUNT] = {"gemm_qkv", "gemm_proj", "gemm_fc1", "gemm_fc2", "gemm_embed", "attention", "layernorm", "im2col",
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

// ----------------------------------------------------------------------------------------------- model handle
struct ribca_vit {
  int D, C, K, depth, hd, hdp, hdv, Dp, Kpe, H4;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  const float *cls, *pos, *pe_b, *norm_w, *norm_b, *head_w, *head_b;
  const float* pe_w;   // fp32 [D][16*C]: the patch embedding stays in fp32 (vit_misc.hip embed_f32_kernel)
  struct Layer {
    const float *ln1w, *ln1b, *qkvb, *projb, *ln2w, *ln2b, *fc1b, *fc2b;
    const uint16_t *qkvw, *projw, *fc1w, *fc2w;
  };
  std::vector<Layer> layers;
};

namespace {

struct Carver {
  char* base; size_t off = 0;
  explicit Carver(char* b) : base(b) {}
  template <class T> T* take(size_t count) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off = align256(off + count * sizeof(T));
    return p;
  }
};

// lays the arena out; with base == nullptr only measures
size_t layout(ribca_vit* m, char* base) {
  Carver c(base);
  const int D = m->D;
  m->cls = c.take<float>(D);
  m->pos = c.take<float>((size_t)kTokens * D);
  m->pe_b = c.take<float>(D);
  m->pe_w = c.take<float>((size_t)D * 16 * m->C);
  m->layers.resize(m->depth);
  for (auto& L : m->layers) {
    L.ln1w = c.take<float>(D); L.ln1b = c.take<float>(D);
    L.qkvb = c.take<float>(3 * D); L.projb = c.take<float>(D);
    L.ln2w = c.take<float>(D); L.ln2b = c.take<float>(D);
    L.fc1b = c.take<float>(4 * D); L.fc2b = c.take<float>(D);
    L.qkvw = c.take<uint16_t>((size_t)gemm_padded_n(3 * D) * 2 * m->Dp);
    L.projw = c.take<uint16_t>((size_t)gemm_padded_n(D) * 2 * m->Dp);
    L.fc1w = c.take<uint16_t>((size_t)gemm_padded_n(4 * D) * 2 * m->Dp);
    L.fc2w = c.take<uint16_t>((size_t)gemm_padded_n(D) * 2 * m->H4);
  }
  m->norm_w = c.take<float>(D); m->norm_b = c.take<float>(D);
  m->head_w = c.take<float>((size_t)m->K * D); m->head_b = c.take<float>(m->K);
  return c.off;
}

struct Workspace {
  float* z; uint16_t* xa; uint16_t* q; uint16_t* k; uint16_t* vt; uint16_t* h;
  size_t qk_bytes, vt_bytes, xa_bytes, total;
};
Workspace carve_ws(const ribca_vit* m, int chunk, char* base) {
  Carver c(base);
  Workspace w;
  const size_t Mc = (size_t)chunk * kTokens;
  w.z = c.take<float>(Mc * m->D);
  w.xa_bytes = Mc * 2 * m->Dp * sizeof(uint16_t);
  w.xa = c.take<uint16_t>(Mc * 2 * m->Dp);
  w.qk_bytes = (size_t)chunk * kHeads * kTokPad * 2 * m->hdp * sizeof(uint16_t);
  w.q = c.take<uint16_t>((size_t)chunk * kHeads * kTokPad * 2 * m->hdp);
  w.k = c.take<uint16_t>((size_t)chunk * kHeads * kTokPad * 2 * m->hdp);
  w.vt_bytes = (size_t)chunk * kHeads * m->hdv * 2 * kKeyPad * sizeof(uint16_t);
  w.vt = c.take<uint16_t>((size_t)chunk * kHeads * m->hdv * 2 * kKeyPad);
  w.h = c.take<uint16_t>(Mc * 2 * m->H4);
  w.total = c.off;
  return w;
}

}  // namespace

extern "C" {

int ribca_version(void) { return 100; }
const char* ribca_last_error(void) { return g_err.c_str(); }
int32_t ribca_gemm_padded_n(int32_t N) { return gemm_padded_n(N); }
int ribca_set_gemm_variant(int32_t v) { gemm_set_variant(v); return 0; }

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
  if (blob_len != ribca_vit_blob_len(D, C, K, depth)) return fail("ribca_vit_create: blob length does not match (D, C, K, depth)");
  hipStream_t s = (hipStream_t)stream;
  ribca_vit* m = new ribca_vit();
  m->D = D; m->C = C; m->K = K; m->depth = depth;
  m->hd = D / kHeads;
  m->hdp = round_up(m->hd, 32);
  m->hdv = round_up(m->hd, 16);
  m->Dp = round_up(D, 32);
  m->Kpe = round_up(16 * C, 32);
  m->H4 = 4 * D;  // multiple of 32 because D % 8 == 0
  m->arena_bytes = layout(m, nullptr);
  hipError_t e = hipMalloc((void**)&m->arena, m->arena_bytes);
  if (e != hipSuccess) { delete m; return hip_fail(e, "hipMalloc(weights)"); }
  layout(m, m->arena);

  const float* p = blob;
  auto copyf = [&](const float* dst, size_t n) -> hipError_t {
    hipError_t r = hipMemcpyAsync((void*)dst, p, n * sizeof(float), hipMemcpyDeviceToDevice, s);
    p += n;
    return r;
  };
  auto pack = [&](const uint16_t* dst, int N, int Kdim, int Kp) {
    launch_pack_weight(p, N, Kdim, const_cast<uint16_t*>(dst), gemm_padded_n(N), Kp, s);
    p += (size_t)N * Kdim;
  };
#define CP(dst, n) do { hipError_t r_ = copyf(dst, n); if (r_ != hipSuccess) { ribca_vit_destroy(m); return hip_fail(r_, "hipMemcpyAsync(param)"); } } while (0)
  CP(m->cls, D);
  CP(m->pos, (size_t)kTokens * D);
  CP(m->pe_w, (size_t)D * 16 * C);
  CP(m->pe_b, D);
  for (auto& L : m->layers) {
    CP(L.ln1w, D); CP(L.ln1b, D);
    pack(L.qkvw, 3 * D, D, m->Dp);
    CP(L.qkvb, 3 * D);
    pack(L.projw, D, D, m->Dp);
    CP(L.projb, D);
    CP(L.ln2w, D); CP(L.ln2b, D);
    pack(L.fc1w, 4 * D, D, m->Dp);
    CP(L.fc1b, 4 * D);
    pack(L.fc2w, D, 4 * D, m->H4);
    CP(L.fc2b, D);
  }
  CP(m->norm_w, D); CP(m->norm_b, D);
  CP(m->head_w, (size_t)K * D); CP(m->head_b, K);
#undef CP
  e = hipGetLastError();
  if (e != hipSuccess) { ribca_vit_destroy(m); return hip_fail(e, "weight packing launch"); }
  *out = m;
  return 0;
}

void ribca_vit_destroy(ribca_vit_t* m) {
  if (!m) return;
  if (m->arena) (void)hipFree(m->arena);
  delete m;
}

double ribca_vit_flops_per_cell(const ribca_vit_t* m) {
  const double d = m->D, n = kTokens;
  return 2.0 * 100 * 16 * m->C * d + m->depth * (24.0 * n * d * d + 4.0 * n * n * d) + 2.0 * d * m->K;
}

int64_t ribca_vit_workspace_bytes(const ribca_vit_t* m, int32_t chunk_cells) {
  if (!m || chunk_cells <= 0) return 0;
  return (int64_t)carve_ws(m, chunk_cells, nullptr).total;
}

int ribca_vit_forward(const ribca_vit_t* m, const float* patches, int32_t c_img, const int32_t* src_chan, int32_t n_cells, float* probs,
                      void* workspace, int64_t workspace_bytes, int32_t chunk_cells, void* stream) {
  if (!m) return fail("ribca_vit_forward: model is NULL");
  if (n_cells < 0 || chunk_cells <= 0) return fail("ribca_vit_forward: bad cell counts");
  if (n_cells == 0) return 0;
  if (!patches || !src_chan || !probs || !workspace) return fail("ribca_vit_forward: NULL buffer");
  if (((uintptr_t)workspace & 255) != 0) return fail("ribca_vit_forward: workspace must be 256-byte aligned");
  const Workspace w = carve_ws(m, chunk_cells, (char*)workspace);
  if ((int64_t)w.total > workspace_bytes) return fail("ribca_vit_forward: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int D = m->D, Dp = m->Dp, ld_x = 2 * Dp, ld_h = 2 * m->H4;
  const float scale = 1.0f / sqrtf((float)m->hd);

  // pads (tokens >= 101, head dims >= hd, feature columns >= D) are never written by any kernel: zero them once per call
  {
    ProfScope ps(P_OTHER, s);
    HIP_TRY(hipMemsetAsync(w.xa, 0, w.xa_bytes, s));
    HIP_TRY(hipMemsetAsync(w.q, 0, w.qk_bytes, s));
    HIP_TRY(hipMemsetAsync(w.k, 0, w.qk_bytes, s));
    HIP_TRY(hipMemsetAsync(w.vt, 0, w.vt_bytes, s));
  }
  for (int c0 = 0; c0 < n_cells; c0 += chunk_cells) {
    const int bc = n_cells - c0 < chunk_cells ? n_cells - c0 : chunk_cells;
    const int Mc = bc * kTokens;
    {
      ProfScope ps(P_EMBED, s);
      launch_embed_f32(patches + (size_t)c0 * c_img * 1600, c_img, src_chan, m->C, m->pe_w, m->pe_b, m->pos, w.z, D, D, bc, s);
    }
    {
      ProfScope ps(P_OTHER, s);
      launch_cls_rows(w.z, D, m->cls, m->pos, D, bc, s);
    }
    for (const auto& L : m->layers) {
      { ProfScope ps(P_LN, s); launch_layernorm_ps(w.z, D, L.ln1w, L.ln1b, w.xa, ld_x, Mc, D, s); }
      {
        ProfScope ps(P_QKV, s);
        GemmArgs g{w.xa, ld_x, L.qkvw, ld_x, Mc, 3 * D, Dp, L.qkvb};
        launch_gemm_qkv(g, w.q, w.k, w.vt, D, m->hd, m->hdp, m->hdv, scale, s);
      }
      { ProfScope ps(P_ATTN, s); launch_attention(w.q, w.k, w.vt, w.xa, ld_x, bc, m->hd, m->hdp, m->hdv, s); }
      {
        ProfScope ps(P_PROJ, s);
        GemmArgs g{w.xa, ld_x, L.projw, ld_x, Mc, D, Dp, L.projb};
        launch_gemm_resid(g, w.z, D, s);
      }
      { ProfScope ps(P_LN, s); launch_layernorm_ps(w.z, D, L.ln2w, L.ln2b, w.xa, ld_x, Mc, D, s); }
      {
        ProfScope ps(P_FC1, s);
        GemmArgs g{w.xa, ld_x, L.fc1w, ld_x, Mc, 4 * D, Dp, L.fc1b};
        launch_gemm_gelu(g, w.h, ld_h, s);
      }
      {
        ProfScope ps(P_FC2, s);
        GemmArgs g{w.h, ld_h, L.fc2w, ld_h, Mc, D, m->H4, L.fc2b};
        launch_gemm_resid(g, w.z, D, s);
      }
    }
    {
      ProfScope ps(P_HEAD, s);
      launch_head_softmax(w.z, D, m->norm_w, m->norm_b, m->head_w, m->head_b, probs + (size_t)c0 * m->K, D, m->K, bc, s);
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------- pre-processing
int ribca_mask_minmax(const int32_t* mask, int64_t n, int32_t* out2, void* stream) {
  if (!out2 || (n > 0 && !mask)) return fail("ribca_mask_minmax: NULL buffer");
  launch_mask_max(mask, n, out2, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

int ribca_label_table(const int32_t* mask, int32_t H, int32_t W, int32_t L, int32_t* tab_i32, uint64_t* tab_u64, void* stream) {
  if (!mask || !tab_i32 || !tab_u64) return fail("ribca_label_table: NULL buffer");
  if (H <= 0 || W <= 0 || L <= 0) return fail("ribca_label_table: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  launch_label_table_init(tab_i32, (unsigned long long*)tab_u64, L, s);
  launch_label_table(mask, H, W, L, tab_i32, (unsigned long long*)tab_u64, s);
  HIP_TRY(hipGetLastError());
  return 0;
}

int ribca_channel_min(const float* image, int32_t C, int64_t hw, float* out_min, void* stream) {
  if (!image || !out_min) return fail("ribca_channel_min: NULL buffer");
  if (C <= 0 || C > 64) return fail("ribca_channel_min: C must be in [1, 64]");
  launch_channel_min(image, C, hw, out_min, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
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
  HIP_TRY(hipGetLastError());
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
  HIP_TRY(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------- normalisation primitives
int ribca_u16_to_f32(const uint16_t* in, float* out, int64_t n, void* stream) {
  if (n > 0 && (!in || !out)) return fail("ribca_u16_to_f32: NULL buffer");
  launch_u16_to_f32(in, out, n, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}
int ribca_gauss1d(const float* in, float* out, int32_t planes, int32_t H, int32_t W, int32_t axis, const double* taps, int32_t R,
                  int32_t mode, void* stream) {
  if (!in || !out || !taps) return fail("ribca_gauss1d: NULL buffer");
  if (in == out) return fail("ribca_gauss1d: in-place filtering is not supported");
  if (axis < 0 || axis > 1 || mode < 0 || mode > 1 || R < 0) return fail("ribca_gauss1d: bad axis/mode/radius");
  launch_gauss1d(in, out, planes, H, W, axis, taps, R, mode, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}
int ribca_bg_subtract(float* x, const float* bg, int64_t n, float cap, void* stream) {
  if (n > 0 && (!x || !bg)) return fail("ribca_bg_subtract: NULL buffer");
  launch_bg_subtract(x, bg, n, cap, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}
int ribca_plane_max(const float* x, int32_t planes, int64_t hw, float* out, void* stream) {
  if (!x || !out) return fail("ribca_plane_max: NULL buffer");
  launch_plane_max(x, planes, hw, out, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}
int ribca_radix_hist(const float* x, int32_t planes, int64_t hw, const uint32_t* prefix, uint32_t mask_hi, int32_t shift, int32_t bits,
                     uint32_t* hist, void* stream) {
  if (!x || !prefix || !hist) return fail("ribca_radix_hist: NULL buffer");
  if (bits < 1 || bits > 11 || shift < 0 || shift + bits > 32) return fail("ribca_radix_hist: bad shift/bits");
  launch_radix_hist(x, planes, hw, prefix, mask_hi, shift, bits, hist, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}
int ribca_norm_finalize(float* x, int32_t planes, int64_t hw, const int32_t* mode, const float* clip, const float* denom, void* stream) {
  if (!x || !mode || !clip || !denom) return fail("ribca_norm_finalize: NULL buffer");
  launch_norm_finalize(x, planes, hw, mode, clip, denom, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
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

// ------------------------------------------------------------------------------------------- test hooks
int ribca_test_pack_weight(const float* w, int32_t N, int32_t K, uint16_t* out, int32_t Np, int32_t Kp, void* stream) {
  launch_pack_weight(w, N, K, out, Np, Kp, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}
int ribca_test_layernorm(const float* z, int32_t ldz, const float* gamma, const float* beta, uint16_t* out, int32_t ldo, int32_t M,
                         int32_t D, void* stream) {
  launch_layernorm_ps(z, ldz, gamma, beta, out, ldo, M, D, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}
int ribca_test_gemm(int32_t kind, const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t M, int32_t N, int32_t Kp,
                    const float* bias, void* out, int32_t ldo, void* stream) {
  GemmArgs g{A, lda, W, ldw, M, N, Kp, bias};
  if (kind == 0) launch_gemm_resid(g, (float*)out, ldo, (hipStream_t)stream);
  else if (kind == 1) launch_gemm_gelu(g, (uint16_t*)out, ldo, (hipStream_t)stream);
  else return fail("ribca_test_gemm: kind must be 0 or 1");
  HIP_TRY(hipGetLastError());
  return 0;
}
int ribca_test_qkv_attention(const uint16_t* A, int32_t lda, const uint16_t* W, int32_t ldw, int32_t cells, int32_t D, int32_t Kp,
                             const float* bias, uint16_t* q, uint16_t* k, uint16_t* vt, uint16_t* out, int32_t ldo, void* stream) {
  const int hd = D / kHeads, hdp = round_up(hd, 32), hdv = round_up(hd, 16);
  GemmArgs g{A, lda, W, ldw, cells * kTokens, 3 * D, Kp, bias};
  launch_gemm_qkv(g, q, k, vt, D, hd, hdp, hdv, 1.0f / sqrtf((float)hd), (hipStream_t)stream);
  launch_attention(q, k, vt, out, ldo, cells, hd, hdp, hdv, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"
