// Multi-head attention  softmax(q k^T) v  per (cell, head) with N = 101 tokens, hd in {12, 24, 32, 48}
// (timm Attention inside each Block; reached from reference cell_type_annotation/model.py:54-55, 402).
//
// One wave owns one (cell, head).  There is no LDS and no barrier: every MFMA operand is loaded straight from HBM/L2
// into the lane that needs it, because the producer (the qkv GEMM epilogue) already wrote Q, K and V^T in fragment order:
//   Q, K : [cell][head][112 tokens][2*hdq] packed-split rows, hdq = round8(hd) (no padding to the MFMA K of 32: a 16-token
//          tile is one contiguous 16*4*hdq-byte block; lane (r = lane&15, g = lane>>4) reads the 32 contiguous bytes (hi|lo)
//          of k-group 4*ks+g of row r, or uses zeros when that group is beyond hdq -- hd = 24 moves 96 instead of 128 bytes
//          per row, hd = 48 192 instead of 256, hd = 12 64 instead of 128).
//   V^T  : [cell][head][hdv][2*128 keys], keys permuted so that the 8 keys lane-group g holds after K*Q^T are contiguous.
//
// Per 16-query tile:  S^T = K * Q^T (keys on MFMA rows, queries on lanes) -> each lane holds, for ITS query, keys
// {16*kt + 4g + r}: the softmax reduction over keys is 28 in-register values + two xor-shuffles (lanes 16/32 apart);
// the normalised probabilities are split to fp16 hi/lo IN PLACE and are already the B operand of  O^T = V^T * P^T
// (k index = key, column = query) -- no cross-lane movement, no LDS round trip.  O^T tiles put 4 consecutive head
// dims of one query in a lane: one 8-byte hi + one 8-byte lo store into the packed-split attention output.
// All three products use the fp16x3 split (ribca_common.h).  Q is pre-scaled by hd^-0.5 by the producer.
#include <cstdlib>

#include "ribca_common.h"
#include "ribca_kernels.h"

namespace ribca {

template <int HD /* head dim */, int NT /* 16-token tiles: 7 for the 101-token classifiers, 1 for the imputer */>
__global__ __launch_bounds__(256) void attention_kernel(const uint16_t* __restrict__ Q, const uint16_t* __restrict__ K,
                                                        const uint16_t* __restrict__ Vt, uint16_t* __restrict__ out, int ldo,
                                                        int n_pairs, int H, int T, int q_tiles) {
  constexpr int KS = (HD + 31) / 32;        // MFMA K steps of the Q K^T product
  constexpr int DT = (HD + 15) / 16;        // 16-row tiles of V^T
  constexpr int hd = HD;
  constexpr int hdq = (HD + 7) / 8 * 8;
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);  // (cell, head) index, wave-uniform
  if (pair >= n_pairs) return;
  const int cell = pair / H, head = pair - cell * H;
  const int r16 = lane & 15, g = lane >> 4;
  constexpr int KST = (NT + 1) / 2;         // 32-key steps of the P*V product
  constexpr int TP = 16 * NT;
  constexpr int ROW = 2 * hdq;              // 16-bit elements per Q/K row
  constexpr int ngrp = hdq >> 3;            // stored k-groups per row
  constexpr int VROW = 2 * 32 * KST;        // 16-bit elements per V^T row
  const uint16_t* qb = Q + (size_t)pair * TP * ROW;
  const uint16_t* kb = K + (size_t)pair * TP * ROW;
  const uint16_t* vb = Vt + (size_t)pair * (DT * 16) * VROW;

  // K fragments stay in registers for all query tiles
  f16x8 khi[NT][KS], klo[NT][KS];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 h4 = {0u, 0u, 0u, 0u}, l4 = {0u, 0u, 0u, 0u};
      if (ks * 4 + 3 < ngrp || ks * 4 + g < ngrp) {
        const uint4* p = reinterpret_cast<const uint4*>(kb + (size_t)(kt * 16 + r16) * ROW + (ks * 4 + g) * 16);
        h4 = p[0]; l4 = p[1];
      }
      khi[kt][ks] = __builtin_bit_cast(f16x8, h4);
      klo[kt][ks] = __builtin_bit_cast(f16x8, l4);
    }

  for (int qt = 0; qt < q_tiles; ++qt) {
    f16x8 qhi[KS], qlo[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 h4 = {0u, 0u, 0u, 0u}, l4 = {0u, 0u, 0u, 0u};
      if ((ks * 4 + 3 < ngrp || ks * 4 + g < ngrp) && qt * 16 + r16 < T) {      // (pad query rows: zeros, never fetched; their output is not stored)
        const uint4* p = reinterpret_cast<const uint4*>(qb + (size_t)(qt * 16 + r16) * ROW + (ks * 4 + g) * 16);
        h4 = p[0]; l4 = p[1];
      }
      qhi[ks] = __builtin_bit_cast(f16x8, h4);
      qlo[ks] = __builtin_bit_cast(f16x8, l4);
    }
    f32x4 s[2 * KST];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s[kt] = mfma_f16(klo[kt][ks], qhi[ks], s[kt]);
        s[kt] = mfma_f16(khi[kt][ks], qlo[ks], s[kt]);
        s[kt] = mfma_f16(khi[kt][ks], qhi[ks], s[kt]);
      }
    }
    // keys >= T (only possible in the last tile) are padding
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * (NT - 1) + 4 * g + r >= T) s[NT - 1][r] = -INFINITY;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // exp(s - mx) = exp2(s log2e - mx log2e): one packed fma per two scores in front of v_exp_f32 (cell_attention.hip does the same)
    const f32x2v l2 = {1.44269504089f, 1.44269504089f}, moff = {-mx * 1.44269504089f, -mx * 1.44269504089f};
    f32x2v sum2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      const f32x2v a0 = fma2(f32x2v{s[kt][0], s[kt][1]}, l2, moff);
      const f32x2v a1 = fma2(f32x2v{s[kt][2], s[kt][3]}, l2, moff);
      const f32x2v e0 = {__builtin_amdgcn_exp2f(a0.x), __builtin_amdgcn_exp2f(a0.y)}, e1 = {__builtin_amdgcn_exp2f(a1.x), __builtin_amdgcn_exp2f(a1.y)};
      s[kt] = f32x4{e0.x, e0.y, e1.x, e1.y};
      sum2 += e0;
      sum2 += e1;
    }
    float sum = sum2.x + sum2.y;
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (NT & 1) s[NT] = f32x4{0.f, 0.f, 0.f, 0.f};
    // P fragments: k-step t covers key tiles 2t (elements 0..3) and 2t+1 (elements 4..7)
    f16x8 phi[KST], plo[KST];
#pragma unroll
    for (int t = 0; t < KST; ++t) {
      const f32x2v inv2 = {inv, inv};
      const f32x2v pa0 = f32x2v{s[2 * t][0], s[2 * t][1]} * inv2, pa1 = f32x2v{s[2 * t][2], s[2 * t][3]} * inv2;
      const f32x2v pb0 = f32x2v{s[2 * t + 1][0], s[2 * t + 1][1]} * inv2, pb1 = f32x2v{s[2 * t + 1][2], s[2 * t + 1][3]} * inv2;
      const float pa[4] = {pa0.x, pa0.y, pa1.x, pa1.y}, pb[4] = {pb0.x, pb0.y, pb1.x, pb1.y};
      uint2 ha, la, hb, lb;
      split4_unit(pa, ha, la);      // probabilities: inside the fp16 range by construction
      split4_unit(pb, hb, lb);
      phi[t] = __builtin_bit_cast(f16x8, uint4{ha.x, ha.y, hb.x, hb.y});
      plo[t] = __builtin_bit_cast(f16x8, uint4{la.x, la.y, lb.x, lb.y});
    }
    const int qtok = qt * 16 + r16;
    uint16_t* orow = out + ((size_t)cell * T + qtok) * ldo;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < KST; ++t) {
        uint4 h4 = {0u, 0u, 0u, 0u}, l4 = {0u, 0u, 0u, 0u};
        if (dt * 16 + 15 < hd || dt * 16 + r16 < hd) {   // V^T rows beyond the head dim are padding: never read
          const uint4* p = reinterpret_cast<const uint4*>(vb + (size_t)(dt * 16 + r16) * VROW + (4 * t + g) * 16);
          h4 = p[0]; l4 = p[1];
        }
        const f16x8 vhi = __builtin_bit_cast(f16x8, h4);
        const f16x8 vlo = __builtin_bit_cast(f16x8, l4);
        o = mfma_f16(vlo, phi[t], o);
        o = mfma_f16(vhi, plo[t], o);
        o = mfma_f16(vhi, phi[t], o);
      }
      const int d = dt * 16 + 4 * g;  // o[r] = O[query = qtok][head dim d + r]
      if (qtok < T && d < hd) {
        float v[4] = {o[0], o[1], o[2], o[3]};
        ps_store4(orow, head * hd + d, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- LDS-staged form (classifiers)
// The kernel above is bound by its dependent chain (Q load -> MFMA -> softmax -> V^T loads from L2 -> MFMA -> store) at 1-3
// waves/SIMD: K fragments of all 112 keys live in registers (114-296 VGPRs).  Here K and V of one (cell, head) are copied
// ONCE into LDS (direct-to-LDS loads, 13-45 KB), WPP waves share them and split the query tiles (tile qt belongs to wave
// qt % WPP), every fragment read is an LDS read of ~100 cycles, and the next query tile's Q fragments are prefetched while
// the current tile is processed.  Same arithmetic, same operation order per query row as above -> identical results.
//
// V arrives ROW-MAJOR, [token][2 * hdq] packed-split rows exactly like K (the qkv epilogue then writes whole row segments instead of
// sixteen 2-byte pieces per value), and is transposed on the way out of LDS: O^T = V^T P^T needs, in lane (d = lane & 15, g), the 8
// keys {32 t + 4 g + r, 32 t + 16 + 4 g + r} of head dim 16 dt + d.  ds_read_b64_tr_b16 hands the 16 lanes of group g a 4-row x
// 16-column block of 16-bit elements column-major (lane 4q + p supplies the address of row q, columns 4p .. 4p + 3; lane i receives
// column i of the four rows): one read per key quartet, hi and lo halves separately = 4 reads per (dt, t) where the V^T form had two
// 16-byte ones.  Rows 101 .. 111 lie beyond the staging descriptor's range and arrive as zeros (never fetched), rows 112 .. 127 are zeroed in LDS: their P is exactly 0.
typedef __attribute__((__vector_size__(4 * sizeof(_Float16)))) _Float16 f16x4;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
__device__ __forceinline__ f16x4 lds_read_tr16(const char* p) {
  return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p)));
}
template <int HD, int NT, int WPP>
__global__ __launch_bounds__(64 * WPP) void attention_lds_kernel(const uint16_t* __restrict__ Q, const uint16_t* __restrict__ K,
                                                                 const uint16_t* __restrict__ V, uint16_t* __restrict__ out, int ldo,
                                                                 int H, int T, int q_tiles) {
  constexpr int KS = (HD + 31) / 32;
  constexpr int DT = (HD + 15) / 16;
  constexpr int hd = HD;
  constexpr int hdq = (HD + 7) / 8 * 8;
  constexpr int KST = (NT + 1) / 2;
  constexpr int TP = 16 * NT;
  constexpr int ROW = 2 * hdq;                       // 16-bit elements per Q/K row
  constexpr int ngrp = hdq >> 3;
  constexpr int ROWB = ROW * 2;                      // bytes per Q / K / V row
  constexpr int K_BYTES = TP * ROWB;
  constexpr int K_LDS = (K_BYTES + 1023) / 1024 * 1024;
  constexpr int V_DMA = K_LDS;                       // V has K's shape: the same 1 KB pieces
  constexpr int V_ROWS = 32 * KST;                   // keys the P V product runs over (128): rows >= TP are zeroed here
  constexpr int V_LDS = ((V_ROWS * ROWB > V_DMA ? V_ROWS * ROWB : V_DMA) + 64 * DT + 15) / 16 * 16;   // + the overshoot of dims >= hdq in the last row
  __shared__ __attribute__((aligned(16))) char lds[K_LDS + V_LDS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.x;
  const int cell = pair / H, head = pair - cell * H;
  const int r16 = lane & 15, g = lane >> 4;
  const uint16_t* qb = Q + (size_t)pair * TP * ROW;
  const char* kb = reinterpret_cast<const char*>(K + (size_t)pair * TP * ROW);
  const char* vb = reinterpret_cast<const char*>(V + (size_t)pair * TP * ROW);

  // ---- stage K and V: 1 KB per wave instruction, linear, through a buffer descriptor that ends behind token row T - 1: the pad rows
  // T .. TP - 1 (101 .. 111: a tenth of the operand) are out of its range, so they arrive in LDS as zeros WITHOUT being fetched -- this
  // kernel runs at its HBM roof (5.6 TB/s at head dim 48), bytes are its only lever.  (Behind V's rows: zeros, written before the copies
  // are issued; the tail piece of a copy lies wholly or partly beyond the range and leaves zeros as well.)
  for (int o = K_BYTES / 1024 * 1024 + threadIdx.x * 16; o < V_LDS; o += 64 * WPP * 16)
    *reinterpret_cast<uint4*>(lds + K_LDS + o) = uint4{0u, 0u, 0u, 0u};
  __syncthreads();
  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kb), 0, T * ROWB, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vb), 0, T * ROWB, 0x00020000);
  for (int i = wave; i < K_LDS / 1024; i += WPP)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (__attribute__((address_space(3))) void*)(lds + i * 1024), 16, i * 1024 + lane * 16, 0, 0, 0);
  // Bank swizzle of the V image (HD = 32 only: 128-byte rows put every second row on the same banks, and a transposed read spans 8
  // rows per half wave: SQ_LDS_BANK_CONFLICT 0.31 of the kernel's cycles without it).  The copy is linear in LDS, so the XOR is applied
  // to the GLOBAL source address: LDS chunk p of row r holds the row's 16-byte chunk p ^ vswz(r).
  auto vswz = [](int row) { return HD == 32 ? (((row >> 1) & 1) | (((row >> 2) & 1) << 2)) : 0; };
  for (int i = wave; i < V_DMA / 1024; i += WPP) {
    int off = i * 1024 + lane * 16;
    if (HD == 32) { const int row = off / ROWB, ch = (off % ROWB) >> 4; off = row * ROWB + ((ch ^ vswz(row)) << 4); }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (__attribute__((address_space(3))) void*)(lds + K_LDS + i * 1024), 16, off, 0, 0, 0);
  }
  // first Q tile of this wave while the copies fly
  auto load_q = [&](int qt, f16x8 (&qh)[KS], f16x8 (&ql)[KS]) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 h4 = {0u, 0u, 0u, 0u}, l4 = {0u, 0u, 0u, 0u};
      if (ks * 4 + 3 < ngrp || ks * 4 + g < ngrp) {
        const uint4* p = reinterpret_cast<const uint4*>(qb + (size_t)(qt * 16 + r16) * ROW + (ks * 4 + g) * 16);
        h4 = p[0]; l4 = p[1];
      }
      qh[ks] = __builtin_bit_cast(f16x8, h4);
      ql[ks] = __builtin_bit_cast(f16x8, l4);
    }
  };
  f16x8 qhi[KS], qlo[KS], qhi_n[KS], qlo_n[KS];
  if (wave < q_tiles) load_q(wave, qhi, qlo);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int qt = wave; qt < q_tiles; qt += WPP) {
    if (qt + WPP < q_tiles) load_q(qt + WPP, qhi_n, qlo_n);
    int opaque = 0;
    asm volatile("" : "+v"(opaque));           // K / V^T fragments are re-read from LDS every tile, not hoisted into registers
    const char* kl = lds + opaque;
    const char* vl = lds + K_LDS + opaque;
    f32x4 s[2 * KST];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        uint4 h4 = {0u, 0u, 0u, 0u}, l4 = {0u, 0u, 0u, 0u};
        if (ks * 4 + 3 < ngrp || ks * 4 + g < ngrp) {
          const uint4* p = reinterpret_cast<const uint4*>(kl + ((kt * 16 + r16) * ROW + (ks * 4 + g) * 16) * 2);
          h4 = p[0]; l4 = p[1];
        }
        const f16x8 kh = __builtin_bit_cast(f16x8, h4), klo_ = __builtin_bit_cast(f16x8, l4);
        s[kt] = mfma_f16(klo_, qhi[ks], s[kt]);
        s[kt] = mfma_f16(kh, qlo[ks], s[kt]);
        s[kt] = mfma_f16(kh, qhi[ks], s[kt]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * (NT - 1) + 4 * g + r >= T) s[NT - 1][r] = -INFINITY;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // exp(s - mx) = exp2(s log2e - mx log2e): one packed fma per two scores in front of v_exp_f32 (cell_attention.hip does the same)
    const f32x2v l2 = {1.44269504089f, 1.44269504089f}, moff = {-mx * 1.44269504089f, -mx * 1.44269504089f};
    f32x2v sum2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      const f32x2v a0 = fma2(f32x2v{s[kt][0], s[kt][1]}, l2, moff);
      const f32x2v a1 = fma2(f32x2v{s[kt][2], s[kt][3]}, l2, moff);
      const f32x2v e0 = {__builtin_amdgcn_exp2f(a0.x), __builtin_amdgcn_exp2f(a0.y)}, e1 = {__builtin_amdgcn_exp2f(a1.x), __builtin_amdgcn_exp2f(a1.y)};
      s[kt] = f32x4{e0.x, e0.y, e1.x, e1.y};
      sum2 += e0;
      sum2 += e1;
    }
    float sum = sum2.x + sum2.y;
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (NT & 1) s[NT] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 phi[KST], plo[KST];
#pragma unroll
    for (int t = 0; t < KST; ++t) {
      const f32x2v inv2 = {inv, inv};
      const f32x2v pa0 = f32x2v{s[2 * t][0], s[2 * t][1]} * inv2, pa1 = f32x2v{s[2 * t][2], s[2 * t][3]} * inv2;
      const f32x2v pb0 = f32x2v{s[2 * t + 1][0], s[2 * t + 1][1]} * inv2, pb1 = f32x2v{s[2 * t + 1][2], s[2 * t + 1][3]} * inv2;
      const float pa[4] = {pa0.x, pa0.y, pa1.x, pa1.y}, pb[4] = {pb0.x, pb0.y, pb1.x, pb1.y};
      uint2 ha, la, hb, lb;
      split4_unit(pa, ha, la);      // probabilities: inside the fp16 range by construction
      split4_unit(pb, hb, lb);
      phi[t] = __builtin_bit_cast(f16x8, uint4{ha.x, ha.y, hb.x, hb.y});
      plo[t] = __builtin_bit_cast(f16x8, uint4{la.x, la.y, lb.x, lb.y});
    }
    const int qtok = qt * 16 + r16;
    uint16_t* orow = out + ((size_t)cell * T + qtok) * ldo;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < KST; ++t) {
        // lane 4q + p of its 16-lane group: row (key) 32 t + 4 g + q, columns (dims) 16 dt + 4p .. + 3 of the hi halves; + 16 bytes: lo
        const int vrow = 32 * t + 4 * g + (r16 >> 2);                 // vswz(vrow + 16) == vswz(vrow)
        const int vch = 2 * (2 * dt + ((r16 & 3) >> 1));             // 16-byte chunk of the hi half; the lo half is the next one
        const char* va = vl + vrow * ROWB + (r16 & 1) * 8;
        const int oh = ((vch ^ vswz(vrow)) << 4), ol = (((vch | 1) ^ vswz(vrow)) << 4);
        const f16x4 h0 = lds_read_tr16(va + oh), l0 = lds_read_tr16(va + ol);
        const f16x4 h1 = lds_read_tr16(va + 16 * ROWB + oh), l1 = lds_read_tr16(va + 16 * ROWB + ol);
        const f16x8 vhi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        const f16x8 vlo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        o = mfma_f16(vlo, phi[t], o);
        o = mfma_f16(vhi, plo[t], o);
        o = mfma_f16(vhi, phi[t], o);
      }
      const int d = dt * 16 + 4 * g;
      if (qtok < T && d < hd) {
        float v[4] = {o[0], o[1], o[2], o[3]};
        ps_store4(orow, head * hd + d, v);
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { qhi[ks] = qhi_n[ks]; qlo[ks] = qlo_n[ks]; }
  }
}

AttnGeom make_attn_geom(int D, int H, int T) {
  AttnGeom a;
  a.D = D; a.H = H; a.hd = D / H; a.T = T;
  a.hdp = (a.hd + 31) / 32 * 32;
  a.hdq = (a.hd + 7) / 8 * 8;
  a.hdv = (a.hd + 15) / 16 * 16;
  a.NT = (T + 15) / 16;
  a.TP = 16 * a.NT;
  a.KP = 32 * ((a.NT + 1) / 2);
  return a;
}

void launch_attention(const uint16_t* q, const uint16_t* k, const uint16_t* vt, uint16_t* out, int ldo, int cells, const AttnGeom& a,
                      hipStream_t s, int q_tiles) {
  if (q_tiles <= 0 || q_tiles > a.NT) q_tiles = a.NT;
  const int pairs = cells * a.H;
  if (pairs <= 0) return;
  if (a.NT == 7) {
#define RIBCA_ATT_LDS(HD_, WPP_) \
  hipLaunchKernelGGL((attention_lds_kernel<HD_, 7, WPP_>), dim3(pairs), dim3(64 * WPP_), 0, s, q, k, vt, out, ldo, a.H, a.T, q_tiles)
    // 4 waves per (cell, head) measured 2 % faster than 2 (871.6 vs 887.6 ms per pass over the five classifiers)
    if (a.hd == 12) { RIBCA_ATT_LDS(12, 4); return; }
    if (a.hd == 24) { RIBCA_ATT_LDS(24, 4); return; }
    if (a.hd == 32) { RIBCA_ATT_LDS(32, 4); return; }
    if (a.hd == 48) { RIBCA_ATT_LDS(48, 4); return; }
#undef RIBCA_ATT_LDS
  }
  const dim3 grid((pairs + 3) / 4), block(256);
#define RIBCA_ATT(HD_, NT_) \
  hipLaunchKernelGGL((attention_kernel<HD_, NT_>), grid, block, 0, s, q, k, vt, out, ldo, pairs, a.H, a.T, q_tiles)
  if (a.NT == 1 && a.hd == 64) RIBCA_ATT(64, 1);
  else launch_error("launch_attention: no kernel for %d tokens x head dim %d (attention_supported() says which exist)", a.T, a.hd);
#undef RIBCA_ATT
}

bool attention_v_rowmajor(const AttnGeom& a) { return a.NT == 7; }
size_t attention_v_elems(const AttnGeom& a, int cells) {
  return attention_v_rowmajor(a) ? (size_t)cells * a.H * a.TP * 2 * a.hdq : (size_t)cells * a.H * a.hdv * 2 * a.KP;
}
bool attention_supported(const AttnGeom& a) {
  return (a.NT == 7 && (a.hd == 12 || a.hd == 24 || a.hd == 32 || a.hd == 48)) || (a.NT == 1 && a.hd == 64);
}

}  // namespace ribca
