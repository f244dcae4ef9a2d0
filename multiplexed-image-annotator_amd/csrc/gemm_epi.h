// Epilogue functors, the two-sweep register epilogue driver and the LDS tile addressing shared by the GEMM kernels
// (gemm_split16.hip: one 768-thread workgroup per CU; gemm_duo.hip: two 256-thread workgroups per CU).
#pragma once
#include <type_traits>
#include <utility>

#include "ribca_common.h"
#include "ribca_kernels.h"

namespace ribca {

constexpr int BK = 32;
constexpr int ROWB = 128;  // bytes per LDS tile row

__device__ __forceinline__ int lds_off(int row, int chunk) {
  const int f = ((row >> 1) & 7) ^ ((((row + 12) & 15) < 8) ? 2 : 0);
  return row * ROWB + ((chunk ^ f) << 4);
}

// ---------------------------------------------------------------------------------------------- epilogues
// Every epilogue runs in two sweeps over the wave's 4 x TN accumulator tiles: `fetch` issues ALL the loads it needs
// (bias once per column group, the residual z tile) back to back, `apply` then computes and stores.  A single sweep
// that loads, waits and stores per tile costs one L2/HBM round trip per tile (16 dependent round trips ~ 11 us per
// 256 x 128 tile, as much as 14 K-steps of MFMAs).
// Interface of an epilogue functor (the drivers below and the LDS drain of gemm_split16.hip call exactly these):
//   fetch_bias(n) / fetch_csum(n)  per 4-column chunk: bias and (folded LayerNorm only) column sums of the weight
//   fetch_row(m)                   per output row: (rstd, -mean * rstd) of the LayerNorm folded into this GEMM, else empty
//   fetch(m, n, ctx)               per (row, chunk): residual / table values
//   apply<PX>(m, n, acc, bias, csum, rowstat, ctx [, Row, Col])
struct NoRow {};
struct LnRow { float rstd, nm; };   // nm = -mean * rstd  (memory holds (rstd, mean) per row: the residual epilogue wants the mean itself)
__device__ __forceinline__ void settle_row(NoRow&) {}
__device__ __forceinline__ void settle_row(LnRow& r) { asm volatile("" : "+v"(r.rstd), "+v"(r.nm)); }

struct EpiResid {
  float* z; int ldz; const float* bias; int M, N; int nt = 0;
  struct Ctx { float4 zv; };
  typedef NoRow RowS;
  __device__ __forceinline__ RowS fetch_row(int) const { return RowS{}; }
  __device__ __forceinline__ float4 fetch_csum(int) const { return float4{0.f, 0.f, 0.f, 0.f}; }
  // The residual tile z[m0 .. m0 + 255][n0 .. n0 + BN) was written one or two launches ago and has left the L2: the drain's z loads
  // are HBM / Infinity-Cache misses, ~3 us of every tile's epilogue (profiles/r2/resid_epilogue_l2_resident_z_experiment.txt).  The
  // loader waves therefore TOUCH it (one dword per 128-byte line, result discarded) right behind the last ring stage they issue,
  // two K steps before the drain reads it.
  static constexpr bool kTouch = true, kFold = false;
  __device__ __forceinline__ bool touch_on() const { return (nt & 2) == 0; }      // nt bit 1: RIBCA_GEMM_TOUCH=0 (A/B)
  __device__ __forceinline__ const float* touch_ptr(int m, int n) const {      // always a valid address: the touch is issued by every
    return z + (size_t)(m < M ? m : M - 1) * ldz + (n < N ? n : N - 1);         // wave, whatever its rows (the caller counts vmcnt)
  }
  __device__ __forceinline__ float4 fetch_bias(int n) const {
    return n < N ? *reinterpret_cast<const float4*>(bias + n) : float4{0.f, 0.f, 0.f, 0.f};
  }
  __device__ __forceinline__ void fetch(int m, int n, Ctx& c) const {
    c.zv = (m < M && n < N) ? *reinterpret_cast<const float4*>(z + (size_t)m * ldz + n) : float4{0.f, 0.f, 0.f, 0.f};
  }
  template <int PX = 16>
  __device__ __forceinline__ void apply(int m, int n, const f32x4& v, const float4& b, const float4&, const RowS&, const Ctx& c) const {
    if (m >= M || n >= N) return;
    f32x4 o;
    o[0] = c.zv.x + (v[0] + b.x); o[1] = c.zv.y + (v[1] + b.y); o[2] = c.zv.z + (v[2] + b.z); o[3] = c.zv.w + (v[3] + b.w);
    f32x4* dst = reinterpret_cast<f32x4*>(z + (size_t)m * ldz + n);
    if (nt & 1) __builtin_nontemporal_store(o, dst);
    else *dst = o;
  }
};

// Residual update on a PACKED-SPLIT residual stream (the classifiers' blocks): z[m][n] += acc + bias with z stored as fp16 hi + lo
// (22 bits), which is exactly the A operand the next qkv / fc1 GEMM reads -- no LayerNorm kernel, no second copy of the row.
// The drain (lds_drain_resid_ps, gemm_split16.hip) also emits per (row, column tile) the mean and the centred sum of squares of the
// NEW row segment; ln_finalize_kernel combines a row's tiles in tile order (Chan's update) into (rstd, -mean rstd).
// Re-centring: every reader of the residual stream of a pre-LN block is a LayerNorm (norm1, norm2, the final norm), so a constant
// added to a whole row is unobservable.  The epilogue therefore subtracts the mean the STORED row had before this update
// (prev[m * prev_stride].y, the statistics the previous LayerNorm used): stored rows keep |mean| <= one update's common mode,
// whatever offset the weights put on the stream, and the fold's cancellation rstd (acc - mean c) stays benign
// (tests/test_gpu_kernels.py::test_vit_forward_large_row_mean).
// Only the LDS-drain form of the kernel supports it (fetch / apply are never called).
struct EpiResidPS {
  static constexpr bool kTouch = true, kFold = false;
  uint16_t* z; int ldz; const float* bias; int M, N; int nt = 0;
  float2* part = nullptr;      // [column tiles][M] (mean, M2), or nullptr: no statistics wanted
  const float2* prev = nullptr; int prev_stride = 1;     // (rstd, mean) of the stored rows, or nullptr: no re-centring
  struct Ctx {};
  typedef NoRow RowS;
  __device__ __forceinline__ bool touch_on() const { return (nt & 2) == 0; }
  __device__ __forceinline__ const float* touch_ptr(int m, int n) const {      // one dword of the 128-byte line holding columns n .. n + 31
    return reinterpret_cast<const float*>(z + (size_t)(m < M ? m : M - 1) * ldz + 2 * (n < N ? n : N - 8));
  }
  __device__ __forceinline__ RowS fetch_row(int) const { return RowS{}; }
  __device__ __forceinline__ float4 fetch_csum(int) const { return float4{0.f, 0.f, 0.f, 0.f}; }
  __device__ __forceinline__ float4 fetch_bias(int n) const {
    return n < N ? *reinterpret_cast<const float4*>(bias + n) : float4{0.f, 0.f, 0.f, 0.f};
  }
  __device__ __forceinline__ void fetch(int, int, Ctx&) const {}
  template <int PX = 16>
  __device__ __forceinline__ void apply(int, int, const f32x4&, const float4&, const float4&, const RowS&, const Ctx&) const {}
};

// The same residual update for the two-workgroups-per-CU kernel (gemm_duo.hip), with NO load in the epilogue.  Round 2 measured what
// sinks that kernel on proj / fc2: beside a neighbour workgroup that keeps two ring stages of LDS-DMA in flight
// even L2-hit loads of an epilogue take ~6 us, the residual tile's z loads 13 of a tile's 17 us -- without them the kernel was 26 % under
// the one-workgroup kernel.  The stored residual stream IS an A operand (packed-split rows), so the z tile rides the A ring as BN / 32
// extra K steps behind the product's own:  z + A W^T = [A | z] [W | I]^T, the identity as a constant register fragment (one 1.0 per
// lane, no weight traffic), two MFMAs per (row tile, column tile) -- lo then hi, both exact products -- on the K step that holds the
// tile's columns.  The asynchronous loader that hides the operands' latency now hides the residual's; bias and the stored rows' means
// are read before the first ring stage.  What is left behind the K loop is VALU and stores: x = (acc - previous mean) + bias, split,
// 16-byte pair stores, and the statistics of the new row segment per (row, WAVE column block of 16 TN columns) -- a lane sums its own
// 4 TN values, two lane-half swaps add the four lanes of a row in a fixed order; no LDS, no barrier, nothing shared between waves.
// ln_finalize_kernel combines the blocks exactly as it combines the one-workgroup kernel's column tiles (more of them, same update).
struct EpiResidZK {
  static constexpr bool kTouch = false, kFold = false, kZK = true;
  uint16_t* z; int ldz; const float* bias; int M, N;
  float2* part = nullptr;      // [N / (16 TN)][M] (mean, M2) per wave column block, or nullptr: no statistics wanted
  const float2* prev = nullptr; int prev_stride = 1;     // (rstd, mean) of the stored rows, or nullptr: no re-centring
  // optional second copy of the NEW rows in the MX3 format (gemm_mx.hip): the operand the MX forms of the next qkv / fc1 read (3 bytes per
  // element instead of 4, and fp16 hi * hi + block-scaled corrections instead of three fp16 passes); hi == nullptr: not wanted.  Only with
  // 48-column wave blocks (TN = 3) and N % 192 == 0: a 32-column scale block then lies in one wave or is shared by a wave and its neighbour.
  MxAct zmx = MxAct{nullptr, nullptr, nullptr, 0, 0};
  int nt = 0;      // A/B switch (RIBCA_MX_NT bit 3): the new rows -- both copies -- stored non-temporal
  struct Ctx {};
  typedef NoRow RowS;
};

// LayerNorm folded into the GEMM that follows it (timm Block.norm1 -> attn.qkv, norm2 -> mlp.fc1):
//   LN(z) W^T + b = rstd (z (gamma o W)^T) - rstd mean c + b',   c[n] = sum_k (gamma o W)[n][k],  b' = b + W beta
// The GEMM reads the residual stream z itself (packed-split), its weight is gamma o W (ribca_vit_create folds it), and the
// epilogue applies the per-row (rstd, nm = -mean rstd) that the preceding residual epilogue produced:  x = rstd acc + (nm c + b').
template <bool FOLD>
__device__ __forceinline__ float4 ln_fold4(const f32x4& v, const float4& b, const float4& c, const std::conditional_t<FOLD, LnRow, NoRow>& r) {
  if constexpr (FOLD) {
    return float4{fmaf(r.rstd, v[0], fmaf(r.nm, c.x, b.x)), fmaf(r.rstd, v[1], fmaf(r.nm, c.y, b.y)), fmaf(r.rstd, v[2], fmaf(r.nm, c.z, b.z)),
                  fmaf(r.rstd, v[3], fmaf(r.nm, c.w, b.w))};
  } else {
    return float4{v[0] + b.x, v[1] + b.y, v[2] + b.z, v[3] + b.w};
  }
}

template <bool FOLD>
struct EpiGeluT {
  static constexpr bool kTouch = false, kFold = FOLD;
  uint16_t* out; int ldo; const float* bias; int M, N; int nt = 0;
  const float2* rowstat = nullptr; const float* csum = nullptr;     // FOLD only
  int rs_stride = 1;                                                 // rowstat[m * rs_stride] belongs to GEMM row m
  struct Ctx {};
  typedef std::conditional_t<FOLD, LnRow, NoRow> RowS;
  // IN = the caller's whole tile lies inside M x N (gemm_duo.hip tests it once per workgroup): no per-access guards, so the register
  // epilogue is one basic block the scheduler can interleave across its 4-value groups instead of 24 exec-masked islands
  static constexpr bool kInside = true;
  template <bool IN = false>
  __device__ __forceinline__ RowS fetch_row(int m) const {
    if constexpr (FOLD) { const float2 r = this->rowstat[(size_t)(IN || m < M ? m : M - 1) * rs_stride]; return RowS{r.x, -r.y * r.x}; }
    else return RowS{};
  }
  template <bool IN = false>
  __device__ __forceinline__ float4 fetch_csum(int n) const {
    if constexpr (FOLD) return (IN || n < N) ? *reinterpret_cast<const float4*>(this->csum + n) : float4{0.f, 0.f, 0.f, 0.f};
    else return float4{0.f, 0.f, 0.f, 0.f};
  }
  template <bool IN = false>
  __device__ __forceinline__ float4 fetch_bias(int n) const {      // bias == nullptr: already added (plain())
    return (bias != nullptr && (IN || n < N)) ? *reinterpret_cast<const float4*>(bias + n) : float4{0.f, 0.f, 0.f, 0.f};
  }
  // the same epilogue for a tile whose fold (and bias) has already been applied to the accumulators (gemm_split16.hip does it before
  // parking the tile: the drain then needs neither row statistics nor column sums)
  __device__ __forceinline__ EpiGeluT<false> plain() const { return EpiGeluT<false>{out, ldo, nullptr, M, N, nt}; }
  __device__ __forceinline__ void fetch(int, int, Ctx&) const {}
  template <int PX = 16, bool IN = false>
  __device__ __forceinline__ void apply(int m, int n, const f32x4& v, const float4& b, const float4& c, const RowS& r, const Ctx&) const {
    if (!IN && (m >= M || n >= N)) return;
    const float4 x = ln_fold4<FOLD>(v, b, c, r);
    const f32x2v u0 = gelu_erf2(f32x2v{x.x, x.y}), u1 = gelu_erf2(f32x2v{x.z, x.w});
    float t[4] = {u0.x, u0.y, u1.x, u1.y};
    ps_store4_pair<PX>(out + (size_t)m * ldo, n, t, nt != 0);      // N % 8 == 0: the partner lane (n ^ 4, same m) passed the same guard
  }
};
typedef EpiGeluT<false> EpiGelu;
typedef EpiGeluT<true> EpiGeluLn;

template <bool FOLD>
struct EpiQKVT {
  static constexpr bool kTouch = false, kFold = FOLD;
  typedef std::conditional_t<FOLD, LnRow, NoRow> RowS;
  static constexpr bool kInside = true;      // see EpiGeluT
  template <bool IN = false>
  __device__ __forceinline__ RowS fetch_row(int m) const {
    if constexpr (FOLD) { const float2 r = this->rowstat[(size_t)(IN || m < M ? m : M - 1) * rs_stride]; return RowS{r.x, -r.y * r.x}; }
    else return RowS{};
  }
  template <bool IN = false>
  __device__ __forceinline__ float4 fetch_csum(int n) const {
    if constexpr (FOLD) return (IN || n < N) ? *reinterpret_cast<const float4*>(this->csum + n) : float4{0.f, 0.f, 0.f, 0.f};
    else return float4{0.f, 0.f, 0.f, 0.f};
  }
  uint16_t* q; uint16_t* k; uint16_t* vt; const float* bias; int D, hd, hdp /* stored dims per Q/K row = AttnGeom::hdq */, hdv; float scale; int M, N;
  int T, TP, H, KP;   // tokens per cell, padded token rows of Q/K, heads, padded keys per V^T row
  int nt = 0;
  unsigned magicT = 0;   // ceil(2^32 / T): m / T as one v_mul_hi + a fix-up (the epilogue does one such division per output row)
  const float2* rowstat = nullptr; const float* csum = nullptr;     // FOLD only
  // vrow != 0: V is stored like K -- row-major packed-split rows [cell][head][TP][2 * hdq] in `vt` -- and the attention kernel
  // transposes it on the way out of LDS (ds_read_b64_tr_b16): every tile of the product then takes the row-contiguous LDS drain,
  // none the sixteen 2-byte scattered stores per value of the V^T form (15 % of the qkv launch)
  int vrow = 0;
  // a column / row window of the full qkv product (the classifiers' LAST block: only the CLS query is ever used, so K and V are
  // computed for every token -- columns D .. 3D, n_off = D -- and Q for the CLS rows alone -- cls_rows = 1: GEMM row m is token 0 of
  // cell m)
  int n_off = 0, cls_rows = 0, rs_stride = 1;
  struct Ctx {};
  // row / column decompositions are computed once per accumulator row (4) and column group (TN), not once per tile
  struct Row { int cell, t, vpos; };
  struct Col { int which, head, d; };
  __device__ __forceinline__ Row row(int m) const {
    Row r;
    if (cls_rows) { r.cell = m; r.t = 0; r.vpos = 0; return r; }
    // m * ceil(2^32 / T) >> 32 is m / T or one more (m < 2^32): a runtime integer division costs ~25 VALU instructions, and the
    // drain evaluates it 8-11 times per thread
    r.cell = (int)__umulhi((unsigned)m, magicT);
    r.t = m - r.cell * T;
    if (r.t < 0) { r.cell -= 1; r.t += T; }
    // V^T key order permuted inside each 32-key block so that the 8 keys a lane group owns after the K*Q^T MFMA (two
    // 16-key tiles, rows 4g..4g+3 of each) are contiguous: key = 32s+16u+4g+r -> 32s+8g+4u+r
    r.vpos = ps_off((r.t & ~31) | (((r.t >> 2) & 3) << 3) | (((r.t >> 4) & 1) << 2) | (r.t & 3));
    return r;
  }
  __device__ __forceinline__ Col col(int n) const {
    Col c;
    const int nn = n + n_off;
    c.which = nn / D;
    const int f = nn - c.which * D;
    c.head = f / hd;
    c.d = f - c.head * hd;   // multiple of 4, d+3 < hd (hd % 4 == 0)
    return c;
  }
  template <bool IN = false>
  __device__ __forceinline__ float4 fetch_bias(int n) const {      // bias == nullptr: already added (plain())
    return (bias != nullptr && (IN || n < N)) ? *reinterpret_cast<const float4*>(bias + n) : float4{0.f, 0.f, 0.f, 0.f};
  }
  __device__ __forceinline__ EpiQKVT<false> plain() const {
    return EpiQKVT<false>{q, k, vt, nullptr, D, hd, hdp, hdv, scale, M, N, T, TP, H, KP, nt, magicT, nullptr, nullptr, vrow, n_off, cls_rows, 1};
  }
  __device__ __forceinline__ void fetch(int, int, Ctx&) const {}
  template <int PX = 16, bool IN = false>
  __device__ __forceinline__ void apply(int m, int n, const f32x4& v, const float4& b, const float4& cs, const RowS& rs, const Ctx&, const Row& r,
                                        const Col& c) const {
    if (!IN && (m >= M || n >= N)) return;
    const float4 xf = ln_fold4<FOLD>(v, b, cs, rs);
    float x[4] = {xf.x, xf.y, xf.z, xf.w};
    const size_t ch = (size_t)r.cell * H + c.head;
    if (c.which < 2 || vrow) {
      if (c.which == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] *= scale;
      }
      uint16_t* rowp = (c.which == 0 ? q : c.which == 1 ? k : vt) + (ch * TP + r.t) * (size_t)(2 * hdp);
      if ((hd & 7) == 0) ps_store4_pair<PX>(rowp, c.d, x, nt != 0);   // the partner lane's 4 columns are in the same head
      else ps_store4(rowp, c.d, x);
    } else {
      uint2 hi, lo;
      split4(x, hi, lo);
      uint16_t* base = vt + (ch * hdv + c.d) * (size_t)(2 * KP) + r.vpos;
      const size_t st = (size_t)(2 * KP);
      base[0] = (uint16_t)hi.x;          base[8] = (uint16_t)lo.x;
      base[st] = (uint16_t)(hi.x >> 16); base[st + 8] = (uint16_t)(lo.x >> 16);
      base[2 * st] = (uint16_t)hi.y;     base[2 * st + 8] = (uint16_t)lo.y;
      base[3 * st] = (uint16_t)(hi.y >> 16); base[3 * st + 8] = (uint16_t)(lo.y >> 16);
    }
  }
};
typedef EpiQKVT<false> EpiQKV;
typedef EpiQKVT<true> EpiQKVLn;

// fp32 output with a per-cell row map (marker imputer): GEMM row m = cell * R + j is written to row
// cell * dst_per_cell + slot[j] of `out`, plus bias and an optional table row add[addrow[j]] (positional embeddings).
struct EpiRowMap {
  static constexpr bool kTouch = false, kFold = false;
  float* out; int ldo; const float* bias; const float* add; int ldadd; const int* slot; const int* addrow; int R, dst_per_cell; int M, N;
  struct Ctx { float4 a; };
  typedef NoRow RowS;
  __device__ __forceinline__ RowS fetch_row(int) const { return RowS{}; }
  __device__ __forceinline__ float4 fetch_csum(int) const { return float4{0.f, 0.f, 0.f, 0.f}; }
  __device__ __forceinline__ float4 fetch_bias(int n) const {
    return n < N ? *reinterpret_cast<const float4*>(bias + n) : float4{0.f, 0.f, 0.f, 0.f};
  }
  __device__ __forceinline__ void fetch(int m, int n, Ctx& c) const {
    c.a = float4{0.f, 0.f, 0.f, 0.f};
    if (add != nullptr && m < M && n < N) {
      const int cell = m / R, j = m - cell * R;
      c.a = *reinterpret_cast<const float4*>(add + (size_t)addrow[j] * ldadd + n);
    }
  }
  template <int PX = 16>
  __device__ __forceinline__ void apply(int m, int n, const f32x4& v, const float4& b, const float4&, const RowS&, const Ctx& c) const {
    if (m >= M || n >= N) return;
    const int cell = m / R, j = m - cell * R;
    float4 o;
    o.x = v[0] + b.x + c.a.x; o.y = v[1] + b.y + c.a.y; o.z = v[2] + b.z + c.a.z; o.w = v[3] + b.w + c.a.w;
    *reinterpret_cast<float4*>(out + ((size_t)cell * dst_per_cell + slot[j]) * ldo + n) = o;
  }
};

// Values that came from global loads are passed through an empty asm statement before the store loop of the LDS drain: the
// compiler's wait-count pass then settles them once (one s_waitcnt), instead of re-emitting s_waitcnt vmcnt(0) at their first use
// in every exec-masked iteration -- where it would also wait for the store issued by the previous iteration.
__device__ __forceinline__ void settle(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
template <class Ctx> __device__ __forceinline__ void settle_ctx(Ctx&) {}
template <> __device__ __forceinline__ void settle_ctx<EpiResid::Ctx>(EpiResid::Ctx& c) { settle(c.zv); }
template <> __device__ __forceinline__ void settle_ctx<EpiRowMap::Ctx>(EpiRowMap::Ctx& c) { settle(c.a); }

// two-sweep driver shared by the kernels: lane owns rows m_i = mbase + 16 i and column groups n_j = nbase + 16 j
template <class Epi, class = void> struct has_rowcol : std::false_type {};
template <class Epi> struct has_rowcol<Epi, std::void_t<typename Epi::Row>> : std::true_type {};

template <class Epi, class = void> struct has_inside : std::false_type {};
template <class Epi> struct has_inside<Epi, std::enable_if_t<Epi::kInside>> : std::true_type {};

// the residual tile rides the A ring as extra K steps (EpiResidZK, gemm_duo.hip)
template <class Epi, class = void> struct is_zk : std::false_type {};
template <class Epi> struct is_zk<Epi, std::enable_if_t<Epi::kZK>> : std::true_type {};

// IN: the whole workgroup tile is inside M x N (only honoured by the epilogues that have unguarded forms: kInside)
template <int TN, class Epi, int R = 4, bool IN = false>
__device__ __forceinline__ void run_epilogue(const Epi& epi, int mbase, int nbase, f32x4 (&acc)[R][TN]) {
  constexpr bool INF = IN && has_inside<Epi>::value;
  float4 b4[TN], c4[TN];
  typename Epi::Ctx ctx[R][TN];
  typename Epi::RowS rs[R];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    if constexpr (INF) { b4[j] = epi.template fetch_bias<true>(nbase + 16 * j); c4[j] = epi.template fetch_csum<true>(nbase + 16 * j); }
    else { b4[j] = epi.fetch_bias(nbase + 16 * j); c4[j] = epi.fetch_csum(nbase + 16 * j); }
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
    if constexpr (INF) rs[i] = epi.template fetch_row<true>(mbase + 16 * i);
    else rs[i] = epi.fetch_row(mbase + 16 * i);
  }
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) epi.fetch(mbase + 16 * i, nbase + 16 * j, ctx[i][j]);
  // one wait for everything that was loaded, before the first store (see settle())
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    settle(b4[j]);
    if constexpr (Epi::kFold) settle(c4[j]);
  }
#pragma unroll
  for (int i = 0; i < R; ++i) settle_row(rs[i]);
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) settle_ctx(ctx[i][j]);
  if constexpr (has_rowcol<Epi>::value) {
    typename Epi::Row rows[R];
    typename Epi::Col cols[TN];
#pragma unroll
    for (int i = 0; i < R; ++i) rows[i] = epi.row(mbase + 16 * i);
#pragma unroll
    for (int j = 0; j < TN; ++j) cols[j] = epi.col(nbase + 16 * j);
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (INF) epi.template apply<16, true>(mbase + 16 * i, nbase + 16 * j, acc[i][j], b4[j], c4[j], rs[i], ctx[i][j], rows[i], cols[j]);
        else epi.apply(mbase + 16 * i, nbase + 16 * j, acc[i][j], b4[j], c4[j], rs[i], ctx[i][j], rows[i], cols[j]);
      }
  } else {
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (INF) epi.template apply<16, true>(mbase + 16 * i, nbase + 16 * j, acc[i][j], b4[j], c4[j], rs[i], ctx[i][j]);
        else epi.apply(mbase + 16 * i, nbase + 16 * j, acc[i][j], b4[j], c4[j], rs[i], ctx[i][j]);
      }
  }
}

// ---------------------------------------------------------------------------------------------- MX3 activation format (gemm_mx.hip)
// exponent byte shared by the producers: ef = exponent field of the block's largest |hi| (fp16 bits)
__device__ __host__ __forceinline__ int mx_sl_byte(int ef) { return (ef < 1 ? 1 : ef) + 93; }     // E - 19 + 127, E = max(ef, 1) - 15
constexpr int kMxShDelta = 17;                                                                       // sh = sl + 17  (E - 2 + 127)
__device__ __forceinline__ float e8m0_float(int byte) { return __builtin_bit_cast(float, (unsigned)byte << 23); }
// position of logical column c in the permuted hi plane
__device__ __host__ __forceinline__ int mx_hi_pos(int c) { return (c & ~127) | (((c >> 3) & 3) << 5) | (((c >> 5) & 3) << 3) | (c & 7); }

// mlp.fc1 with the LayerNorm fold whose output IS the MX3 operand of the MX fc2 (gemm_mx.hip): gelu(rstd acc + (nm c + b')) written as
// fp16 hi (permuted plane) + e4m3 lo + one scale byte per (row, 32 columns) -- 3 bytes per element instead of the 4 of the packed-split
// form, and nothing left to convert in front of fc2.  Exists on the two-workgroups-per-CU kernel with 4 waves as 1 x 4 and 128-wide tiles
// only (gemm_duo.hip): a wave then owns exactly ONE 32-column block of each of its rows, so the block's scale is a four-lane maximum.
struct EpiGeluMx {
  static constexpr bool kTouch = false, kFold = true, kMxOut = true;
  MxAct out; const float* bias; int M, N;
  const float2* rowstat; const float* csum; int rs_stride = 1;
  int nt = 0;      // 1: the emitted planes are stored non-temporal (A/B switch RIBCA_MX_NT: h is read once, by the next launch)
  struct Ctx {};
  typedef LnRow RowS;
};
template <class Epi, class = void> struct is_mx_out : std::false_type {};
template <class Epi> struct is_mx_out<Epi, std::enable_if_t<Epi::kMxOut>> : std::true_type {};

// maximum over the four lanes (r16, g = 0 .. 3) that share an output row (see g4_sum below for the inline asm)
__device__ __forceinline__ float g4_max(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  float c = fmaxf(a, b), d;
  asm volatile("v_mov_b32 %0, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %1, %0\n\ts_nop 1" : "=&v"(d), "+v"(c));
  return fmaxf(c, d);
}
typedef short mx_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
// acc: the wave's MT x 2 tiles (lane: row r16 of every row tile, columns 4 g .. 4 g + 3 of both column tiles); n32 = first column of the
// wave's 32-column block.  IN: every row of the tile is inside M (N is a multiple of the tile width by construction).
template <int MT, bool IN>
__device__ __forceinline__ void gelu_mx_epilogue(const EpiGeluMx& epi, int mbase, int n32, int g, f32x4 (&acc)[1][MT][2]) {
  const int nb0 = n32 + 4 * g;
  float4 b4[2], c4[2];
  LnRow rs[MT];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    b4[j] = *reinterpret_cast<const float4*>(epi.bias + nb0 + 16 * j);
    c4[j] = *reinterpret_cast<const float4*>(epi.csum + nb0 + 16 * j);
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = mbase + 16 * i;
    const float2 r = epi.rowstat[(size_t)(IN || m < epi.M ? m : epi.M - 1) * epi.rs_stride];
    rs[i] = LnRow{r.x, -r.y * r.x};
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) { settle(b4[j]); settle(c4[j]); }
#pragma unroll
  for (int i = 0; i < MT; ++i) settle_row(rs[i]);
  const int Kp = epi.out.Kp;
  // this lane's store columns: the pair (g, g ^ 1) owns 8 consecutive columns of each column tile; the even lane stores tile 0's, the odd lane tile 1's
  const int c8 = n32 + 16 * (g & 1) + 8 * (g >> 1);
  const int hpos = mx_hi_pos(c8);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = mbase + 16 * i;
    const bool ok = IN || m < epi.M;
    float x[2][4];
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4 f = ln_fold4<true>(acc[0][i][j], b4[j], c4[j], rs[i]);
      const f32x2v u0 = gelu_erf2(f32x2v{f.x, f.y}), u1 = gelu_erf2(f32x2v{f.z, f.w});
      x[j][0] = clamp_f16_range(u0.x); x[j][1] = clamp_f16_range(u0.y); x[j][2] = clamp_f16_range(u1.x); x[j][3] = clamp_f16_range(u1.y);
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(x[j][0]), fabsf(x[j][1])), fmaxf(fabsf(x[j][2]), fabsf(x[j][3]))));
    }
    mx = g4_max(mx);
    // the block's exponent byte from fp16(max |x|) (rounding is monotone: the largest |hi| is the hi of the largest |x|)
    const int ef = (int)(f16_bits(mx) >> 10);
    const int sl = mx_sl_byte(ef);
    const float scale = e8m0_float(sl);
    uint2 hi[2];
    unsigned l8[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      hi[j].x = cvt_pk_f16(x[j][0], x[j][1]);
      hi[j].y = cvt_pk_f16(x[j][2], x[j][3]);
      mx_s16x2 r = {0, 0};
      r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, f32_minus_f16lo(x[j][0], hi[j].x), f32_minus_f16hi(x[j][1], hi[j].x), scale, false);
      r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, f32_minus_f16lo(x[j][2], hi[j].y), f32_minus_f16hi(x[j][3], hi[j].y), scale, true);
      l8[j] = __builtin_bit_cast(unsigned, r);
    }
    // even lane: its own 4 columns of tile 0 and the partner's; odd lane: the partner's 4 columns of tile 1 and its own
    const auto rx = __builtin_amdgcn_permlane16_swap(hi[0].x, hi[1].x, false, false);
    const auto ry = __builtin_amdgcn_permlane16_swap(hi[0].y, hi[1].y, false, false);
    const auto rl = __builtin_amdgcn_permlane16_swap(l8[0], l8[1], false, false);
    if (ok) {
      u32x4* hp = reinterpret_cast<u32x4*>(epi.out.hi + (size_t)m * Kp + hpos);
      u32x2s* lp = reinterpret_cast<u32x2s*>(epi.out.l8 + (size_t)m * Kp + c8);
      if (epi.nt) {      // h is read once, by the next launch: stores that do not displace the weight from the L2 (RIBCA_MX_NT)
        __builtin_nontemporal_store(u32x4{rx[0], ry[0], rx[1], ry[1]}, hp);
        __builtin_nontemporal_store(u32x2s{rl[0], rl[1]}, lp);
      } else {
        *hp = u32x4{rx[0], ry[0], rx[1], ry[1]};
        *lp = u32x2s{rl[0], rl[1]};
      }
      if (g == 0) epi.out.sc[((size_t)(n32 >> 7) * epi.out.M + m) * 4 + ((n32 >> 5) & 3)] = (unsigned char)sl;
    }
  }
}

// ---- MX3 emission of a 128 x 192 workgroup tile whose four waves own 8 row tiles x 3 column tiles each (48 columns starting at a multiple
// of 48; lane: row r16 of every row tile, columns 4 g .. 4 g + 3 of every column tile).  32-column scale blocks against 48-column wave
// blocks: an EVEN wave block owns its columns 0 .. 31 (tiles 0, 1) and shares 32 .. 47 (tile 2) with tile 0 of the next (odd) wave block,
// which owns its columns 16 .. 47.  The shared block's maximum goes through 2 KB of LDS (xch: [4 waves][128 rows] floats outside the
// staging image) and a workgroup barrier; the converted pieces through the staging image (stg) and a second barrier.
// EVERY wave of the workgroup must call this (the launchers take N % 192 == 0: no wave of a tile lies beyond N).
// x: the values to emit, already clamped to the fp16 range.  The pair (g, g ^ 1) owns 8 consecutive columns of a tile: the even lane stages
// their 8 hi halves (16 bytes), the odd lane their 8 lo bytes.
template <int... Is, class F>
__device__ __forceinline__ void epi_sfor_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void epi_sfor(F&& f) { epi_sfor_impl(std::make_integer_sequence<int, N>{}, f); }

// (inline asm with lambda-local operands does not compile inside a generic lambda: through functions)
template <int OFF> __device__ __forceinline__ void epi_lds_wr32(unsigned addr, float v) {
  asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void epi_lds_rd32(float& dst, unsigned addr) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int CNT> __device__ __forceinline__ void epi_wait_lgkm2(float& v, float& w) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(v), "+v"(w) : "n"(CNT) : "memory");
}

template <int OFF> __device__ __forceinline__ void epi_lds_wr128(unsigned addr, const u32x4& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void epi_lds_wr64(unsigned addr, const u32x2s& v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void epi_lds_rd128(u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int CNT> __device__ __forceinline__ void epi_wait_lgkm128(u32x4& v) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(CNT) : "memory"); }
// staging image of a 128 x 192 tile's MX3 planes: hi rows of 24 chunks in plane order (384 bytes, pitch 400: the 16 rows a wave instruction
// writes then fall into 16 different bank groups), lo rows of 192 bytes (pitch 208)
constexpr int kMx3StageHiRow = 400, kMx3StageL8Row = 208, kMx3StageBytes = 128 * (kMx3StageHiRow + kMx3StageL8Row);

template <int MT, bool IN>
__device__ __forceinline__ void mx3_emit_wave48(const MxAct& out, int M, int m0, int ncol0, int g_in, int r16_in, int wave, unsigned stg_lds,
                                                unsigned xch_lds, const f32x4 (&x)[MT][3], bool nt = false) {
  static_assert(MT == 8, "the staging image is that of a 128-row tile");
  // (opaque copies: everything below that derives from the lane's position is then formed HERE, not shared with the kernel's prologue and
  // carried -- or spilled -- across the K loop)
  int g = g_in, r16 = r16_in;
  asm volatile("" : "+v"(g), "+v"(r16));
  const bool odd = ((ncol0 / 48) & 1) != 0;
  // the shared block's partial maxima change hands with the neighbour wave (wave ^ 1) through LDS; a wave reads its own back as well rather
  // than keeping eight more values in registers across the barrier
  const unsigned mine = xch_lds + (unsigned)((wave * 128 + r16) * 4), theirs = xch_lds + (unsigned)(((wave ^ 1) * 128 + r16) * 4);
  // (a row's two exponent bytes are kept PACKED, four rows to a register: 4 registers instead of 16 beside the 96 values to emit)
  static_assert(MT % 4 == 0, "exponent bytes are packed four rows to a register");
  unsigned so_pk[MT / 4], ss_pk[MT / 4];
#pragma unroll
  for (int q = 0; q < MT / 4; ++q) so_pk[q] = ss_pk[q] = 0u;
  __builtin_amdgcn_sched_barrier(0);      // (keeps the maxima out of the caller's last stores: registers)
  epi_sfor<MT>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    float m3[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) m3[j] = fmaxf(fmaxf(fabsf(x[i][j][0]), fabsf(x[i][j][1])), fmaxf(fabsf(x[i][j][2]), fabsf(x[i][j][3])));
    // exponent byte of a block from fp16(max |x|): rounding is monotone, so that is the largest |hi|
    const float mo = g4_max(odd ? fmaxf(m3[1], m3[2]) : fmaxf(m3[0], m3[1]));
    so_pk[i / 4] |= (unsigned)mx_sl_byte((int)(f16_bits(mo) >> 10)) << (8 * (i % 4));
    asm volatile("" : "+v"(so_pk[i / 4]));      // (HERE: left to itself the end of the reduction sinks to its use and both halves stay live)
    const float ms = g4_max(odd ? m3[0] : m3[2]);
    if (g == 0) epi_lds_wr32<i * 64>(mine, ms);
    __builtin_amdgcn_sched_barrier(0);
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  epi_sfor<MT / 4>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    float mp[4], mq[4];
    epi_sfor<4>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      epi_lds_rd32<(4 * q + i) * 64>(mp[i], theirs);
      epi_lds_rd32<(4 * q + i) * 64>(mq[i], mine);
    });
    epi_sfor<4>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      epi_wait_lgkm2<2 * (3 - i)>(mp[i], mq[i]);
      ss_pk[q] |= (unsigned)mx_sl_byte((int)(f16_bits(fmaxf(mp[i], mq[i])) >> 10)) << (8 * i);
    });
    asm volatile("" : "+v"(ss_pk[q]));
  });
  // The converted pieces go through LDS (stg_lds: kMx3StageBytes that no wave reads any more once every wave has passed the barrier above -- the
  // operand ring) and leave it as whole runs: stored straight from the accumulator layout a wave instruction writes 32 isolated 16-byte
  // pieces of the permuted hi plane and 32 8-byte pieces of the lo plane -- the same bytes in contiguous runs take 4 of the 7 ms that the
  // stores of 44 fc1 launches at D = 576 cost (profiles/r4/mx_emit_store_ablation.txt).
  const int Kp = out.Kp;
  const int n0 = ncol0 / 192 * 192, wcol = ncol0 - n0;      // the tile's first column (a multiple of 192) and the wave's first column in it
  const bool half_first = (n0 & 64) != 0;                   // the tile starts in the middle of a 128-column group of the hi plane
  const int c8 = wcol + 8 * (g >> 1);                       // + 16 j: the 8 columns the pair (g, g ^ 1) owns in column tile j, relative to the tile
  // slot (16-byte chunk of the staged 384-byte hi row) of an 8-column piece: the row's chunks in PLANE order -- a whole group is 16 chunks
  // at the plane's positions, a half group its 8 pieces (sub-step s = 0 .. 3, two adjacent 8-column pieces each)
  unsigned hi_st[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int c = n0 + c8 + 16 * j, s_ = (c >> 3) & 3, gg = (c >> 5) & 3;
    const bool in_first = (c >> 7) == (n0 >> 7);
    int q;
    if (!half_first) q = in_first ? 4 * s_ + gg : 16 + 2 * s_ + gg;      // whole group, then the first half (gg = 0, 1) of the next
    else q = in_first ? 2 * s_ + gg - 2 : 8 + 4 * s_ + gg;               // second half (gg = 2, 3), then a whole group
    hi_st[j] = stg_lds + (unsigned)(r16 * kMx3StageHiRow + q * 16);
  }
  const unsigned l8_st = stg_lds + (unsigned)(128 * kMx3StageHiRow + r16 * kMx3StageL8Row + c8);
  const int rows = IN ? 16 * MT : ((M - m0) < 16 * MT ? (M - m0) : 16 * MT);
  const int blk_own = (ncol0 + (odd ? 16 : 0)) >> 5, blk_sh = (ncol0 + (odd ? 0 : 32)) >> 5;      // 32-column block indices
  unsigned char* sc_own = out.sc + ((size_t)(blk_own >> 2) * out.M + m0 + r16) * 4 + (blk_own & 3);
  unsigned char* sc_sh = out.sc + ((size_t)(blk_sh >> 2) * out.M + m0 + r16) * 4 + (blk_sh & 3);
  epi_sfor<MT>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const int sl_o = (int)((so_pk[i / 4] >> (8 * (i % 4))) & 0xffu), sl_s = (int)((ss_pk[i / 4] >> (8 * (i % 4))) & 0xffu);
    epi_sfor<3>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const bool shared = odd ? j == 0 : j == 2;
      const float scale = e8m0_float(shared ? sl_s : sl_o);
      uint2 hi;
      hi.x = cvt_pk_f16(x[i][j][0], x[i][j][1]);
      hi.y = cvt_pk_f16(x[i][j][2], x[i][j][3]);
      mx_s16x2 r = {0, 0};
      r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, f32_minus_f16lo(x[i][j][0], hi.x), f32_minus_f16hi(x[i][j][1], hi.x), scale, false);
      r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, f32_minus_f16lo(x[i][j][2], hi.y), f32_minus_f16hi(x[i][j][3], hi.y), scale, true);
      const unsigned l8 = __builtin_bit_cast(unsigned, r);
      // even lane: own hi.x / hi.y and the partner's; odd lane: the partner's lo bytes and its own (the two operands of a swap differ on
      // purpose: handed the same value twice hipcc folds the swap's two results into one, tools/swap_probe.hip)
      const auto rx = __builtin_amdgcn_permlane16_swap(hi.x, l8, false, false);
      const auto ry = __builtin_amdgcn_permlane16_swap(hi.y, l8, false, false);
      if ((g & 1) == 0) epi_lds_wr128<i * 16 * kMx3StageHiRow>(hi_st[j], u32x4{rx[0], ry[0], rx[1], ry[1]});
      else epi_lds_wr64<i * 16 * kMx3StageL8Row + 16 * j>(l8_st, u32x2s{rx[0], rx[1]});
    });
    if ((IN || r16 + 16 * i < rows) && g == 0) {
      sc_own[i * 64] = (unsigned char)sl_o;
      if (!odd) sc_sh[i * 64] = (unsigned char)sl_s;
    }
    __builtin_amdgcn_sched_barrier(0);
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // ---- out of LDS in runs: rows beyond M are dropped by the descriptors' range check
  const __amdgpu_buffer_rsrc_t hi_rsrc = __builtin_amdgcn_make_buffer_rsrc(out.hi + (size_t)m0 * Kp, 0, rows * Kp * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t l8_rsrc = __builtin_amdgcn_make_buffer_rsrc(out.l8 + (size_t)m0 * Kp, 0, rows * Kp, 0x00020000);
  const int tid = wave * 64 + r16 + 16 * g;
  {
    // hi: 8 consecutive lanes take 8 consecutive chunks (128 bytes) of a row, three times per row (24 chunks), 32 rows per instruction
    const int l8i = tid & 7, row0 = tid >> 3;
    const int gbase = (n0 >> 7) * 128;      // first column of the 128-column group the tile starts in
    int pos[3];                             // plane position (fp16 elements from the row start) of chunk q = l8i + 8 t
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int q = l8i + 8 * t;
      if (!half_first) pos[t] = q < 16 ? gbase + 8 * q : gbase + 128 + 32 * ((q - 16) >> 1) + 8 * ((q - 16) & 1);
      else pos[t] = q < 8 ? gbase + 32 * (q >> 1) + 8 * (2 + (q & 1)) : gbase + 128 + 8 * (q - 8);
    }
    const unsigned rd = stg_lds + (unsigned)(row0 * kMx3StageHiRow + l8i * 16);
    const int voff = row0 * Kp * 2;
    u32x4 v[12];
    epi_sfor<4>([&](auto rc) {
      constexpr int rr = decltype(rc)::value;
      epi_sfor<3>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        epi_lds_rd128<rr * 32 * kMx3StageHiRow + t * 128>(v[rr * 3 + t], rd);
      });
    });
    epi_sfor<12>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      epi_wait_lgkm128<11 - k>(v[k]);
      if (nt) __builtin_amdgcn_raw_buffer_store_b128(v[k], hi_rsrc, voff + pos[k % 3] * 2, (k / 3) * 32 * Kp * 2, 2);
      else __builtin_amdgcn_raw_buffer_store_b128(v[k], hi_rsrc, voff + pos[k % 3] * 2, (k / 3) * 32 * Kp * 2, 0);
    });
  }
  {
    // lo: a row is 192 bytes = 12 chunks; 4 consecutive lanes take 4 consecutive chunks, three times per row, 64 rows per instruction
    const int l4 = tid & 3, row0 = tid >> 2;
    const unsigned rd = stg_lds + (unsigned)(128 * kMx3StageHiRow + row0 * kMx3StageL8Row + l4 * 16);
    const int voff = row0 * Kp + n0 + l4 * 16;
    u32x4 v[6];
    epi_sfor<2>([&](auto rc) {
      constexpr int rr = decltype(rc)::value;
      epi_sfor<3>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        epi_lds_rd128<rr * 64 * kMx3StageL8Row + t * 64>(v[rr * 3 + t], rd);
      });
    });
    epi_sfor<6>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      epi_wait_lgkm128<5 - k>(v[k]);
      if (nt) __builtin_amdgcn_raw_buffer_store_b128(v[k], l8_rsrc, voff + (k % 3) * 64, (k / 3) * 64 * Kp, 2);
      else __builtin_amdgcn_raw_buffer_store_b128(v[k], l8_rsrc, voff + (k % 3) * 64, (k / 3) * 64 * Kp, 0);
    });
  }
}

// ---------------------------------------------------------------------------------------------- EpiResidZK pieces shared by gemm_duo.hip and gemm_mx.hip
// ---- EpiResidZK (gemm_epi.h): descriptor over the residual tile's rows, the four-lane sum and the load-free epilogue
template <class Epi>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t zk_rsrc(const Epi& epi, int m0, int rows_here, __amdgpu_buffer_rsrc_t other) {
  if constexpr (is_zk<Epi>::value) {
    // the last row ends with its own 2 * Dp values, not with the row pitch (ldz can be a multiple of it): reads behind it return zeros
    const int row_bytes = ((epi.N + 31) / 32 * 32) * 4;
    return __builtin_amdgcn_make_buffer_rsrc(epi.z + (size_t)m0 * epi.ldz, 0, (rows_here - 1) * epi.ldz * 2 + row_bytes, 0x00020000);
  } else {
    return other;
  }
}
// sum over the four lanes (r16, g = 0 .. 3) that share an output row: two lane-half swaps (gfx950), the same bits in all four, fixed order
// (Written as inline asm: handed the same value twice, hipcc folds the two results of the swap builtins into one -- the ISA then adds a
// register to itself; tools/swap_probe.hip shows it.  s_nop: the swaps read VALU results of the instruction just before.)
__device__ __forceinline__ float g4_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  float c = a + b, d;
  asm volatile("v_mov_b32 %0, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %1, %0\n\ts_nop 1" : "=&v"(d), "+v"(c));
  return c + d;
}
// acc holds z + A W^T of the wave's MT x TN tiles (lane: row r16 of every row tile, columns 4 g .. 4 g + 3 of every column tile);
// b4 / pm were read before the K loop.  IN: the whole workgroup tile lies inside M x N.
template <int TN, int MT, int RB, bool IN>
__device__ __forceinline__ void resid_zk_epilogue(const EpiResidZK& epi, int mbase, int nbase, int blk, int g, f32x4 (&acc)[MT / RB][RB][TN],
                                                  const float4 (&b4)[TN], const float (&pm)[MT]) {
  float sum[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    f32x4(&a)[TN] = acc[i / RB][i % RB];
    const int m = mbase + 16 * i;
    const bool ok = IN || m < epi.M;
    uint16_t* zr = epi.z + (size_t)(ok ? m : 0) * epi.ldz;
    const f32x2p pm2 = {pm[i], pm[i]};
    f32x2p tot = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f32x2p x0 = (f32x2p{a[j][0], a[j][1]} - pm2) + f32x2p{b4[j].x, b4[j].y};
      f32x2p x1 = (f32x2p{a[j][2], a[j][3]} - pm2) + f32x2p{b4[j].z, b4[j].w};
      x0.x = clamp_f16_range(x0.x); x0.y = clamp_f16_range(x0.y); x1.x = clamp_f16_range(x1.x); x1.y = clamp_f16_range(x1.y);
      a[j] = f32x4{x0.x, x0.y, x1.x, x1.y};
      tot += x0; tot += x1;
      // split (values already inside the fp16 range) and pair store: the partner lane (g ^ 1) holds the other 4 columns of the PS group
      uint2 hi, lo;
      hi.x = cvt_pk_f16(x0.x, x0.y); hi.y = cvt_pk_f16(x1.x, x1.y);
      lo.x = cvt_pk_f16(f32_minus_f16lo(x0.x, hi.x), f32_minus_f16hi(x0.y, hi.x));
      lo.y = cvt_pk_f16(f32_minus_f16lo(x1.x, hi.y), f32_minus_f16hi(x1.y, hi.y));
      const auto rx = __builtin_amdgcn_permlane16_swap(hi.x, lo.x, false, false);
      const auto ry = __builtin_amdgcn_permlane16_swap(hi.y, lo.y, false, false);
      const u32x4 o = {rx[0], ry[0], rx[1], ry[1]};      // even g: 8 x hi, odd g: 8 x lo
      const int k = nbase + 16 * j;
      if (ok) {
        u32x4* zp = reinterpret_cast<u32x4*>(zr + ps_off(k & ~7) + ((k & 4) ? 8 : 0));
        if (epi.nt) __builtin_nontemporal_store(o, zp);
        else *zp = o;
      }
    }
    sum[i] = tot.x + tot.y;
  }
  if (epi.part == nullptr) return;      // (acc keeps the new values x either way: the MX3 copy is emitted from it by the caller)
  constexpr float inv = 1.0f / (float)(16 * TN);
#pragma unroll
  for (int i = 0; i < MT; ++i) sum[i] = g4_sum(sum[i]) * inv;      // block mean of the row
  float q[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    f32x4(&a)[TN] = acc[i / RB][i % RB];
    const f32x2p mu = {sum[i], sum[i]};
    f32x2p qq = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const f32x2p d0 = f32x2p{a[j][0], a[j][1]} - mu, d1 = f32x2p{a[j][2], a[j][3]} - mu;
      qq += d0 * d0; qq += d1 * d1;
    }
    q[i] = qq.x + qq.y;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) q[i] = g4_sum(q[i]);
  if (g == 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = mbase + 16 * i;
      if (IN || m < epi.M) epi.part[(size_t)blk * epi.M + m] = float2{sum[i], q[i]};
    }
  }
}

// mlp.fc1 on the MX kernel (gemm_mx.hip: a wave owns MT x 3 tiles): LayerNorm fold + GELU in place, then the MX3 planes of the result
template <int MT, bool IN>
__device__ __forceinline__ void gelu_mx48_epilogue(const EpiGeluMx& epi, int m0, int ncol0, int g, int r16, int wave, unsigned stg_lds,
                                                   unsigned xch_lds, f32x4 (&acc)[1][MT][3]) {
  const int nb0 = ncol0 + 4 * g, mbase = m0 + r16;
  float4 b4[3], c4[3];
  LnRow rs[MT];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    b4[j] = *reinterpret_cast<const float4*>(epi.bias + nb0 + 16 * j);
    c4[j] = *reinterpret_cast<const float4*>(epi.csum + nb0 + 16 * j);
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = mbase + 16 * i;
    const float2 r = epi.rowstat[(size_t)(IN || m < epi.M ? m : epi.M - 1) * epi.rs_stride];
    rs[i] = LnRow{r.x, -r.y * r.x};
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) { settle(b4[j]); settle(c4[j]); }
#pragma unroll
  for (int i = 0; i < MT; ++i) settle_row(rs[i]);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float4 f = ln_fold4<true>(acc[0][i][j], b4[j], c4[j], rs[i]);
#ifdef MXDBG_NOGELU      // (timing ablation, tools/build_mx_variant.py)
      const f32x2v u0 = f32x2v{f.x, f.y}, u1 = f32x2v{f.z, f.w};
#else
      const f32x2v u0 = gelu_erf2(f32x2v{f.x, f.y}), u1 = gelu_erf2(f32x2v{f.z, f.w});
#endif
      acc[0][i][j] = f32x4{clamp_f16_range(u0.x), clamp_f16_range(u0.y), clamp_f16_range(u1.x), clamp_f16_range(u1.y)};
      // (one tile at a time: interleaving the 24 erf evaluations costs more registers than the wave has beside its accumulators)
      if ((j & 1) == 1 || j == 2) __builtin_amdgcn_sched_barrier(0);
    }
#ifdef MXDBG_NOEMIT
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) asm volatile("" ::"v"(acc[0][i][j]));
#else
  mx3_emit_wave48<MT, IN>(epi.out, epi.M, m0, ncol0, g, r16, wave, stg_lds, xch_lds, acc[0], epi.nt != 0);
#endif
}


// ---------------------------------------------------------------------------------------------- kernel
__device__ __forceinline__ int swz_f(int row) { return ((row >> 1) & 7) ^ ((((row + 12) & 15) < 8) ? 2 : 0); }

template <int CNT> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(CNT >= 0 && CNT <= 63, "vmcnt is a 6-bit field on gfx9");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
}

// gemm_duo.hip: two 256-thread workgroups per CU, W in fragment order straight to registers (abl: timing ablations, 0 = none)
template <int BN, class Epi>
bool launch_duo(const GemmArgs& g, const Epi& epi, hipStream_t s, int abl);
int duo_set_stamp_buffer(void* dev_ptr, unsigned int capacity_blocks);

}  // namespace ribca
