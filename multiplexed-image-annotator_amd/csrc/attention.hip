// Multi-head attention  softmax(q k^T) v  per (cell, head) with N = 101 tokens, hd in {12, 24, 32, 48}
// (timm Attention inside each Block; reached from reference cell_type_annotation/model.py:54-55, 402).
//
// One wave owns one (cell, head).  There is no LDS and no barrier: every MFMA operand is loaded straight from HBM/L2
// into the lane that needs it, because the producer (the qkv GEMM epilogue) already wrote Q, K and V^T in fragment order:
//   Q, K : [cell][head][112 tokens][2*hdp] packed-split rows  -> a 16-token tile is one contiguous 16*4*hdp-byte block and
//          lane (r = lane&15, g = lane>>4) reads 32 contiguous bytes (hi|lo of k-group g) of row r: fully coalesced.
//   V^T  : [cell][head][hdv][2*128 keys], keys permuted so that the 8 keys lane-group g holds after K*Q^T are contiguous.
//
// Per 16-query tile:  S^T = K * Q^T (keys on MFMA rows, queries on lanes) -> each lane holds, for ITS query, keys
// {16*kt + 4g + r}: the softmax reduction over keys is 28 in-register values + two xor-shuffles (lanes 16/32 apart);
// the normalised probabilities are split to bf16 hi/lo IN PLACE and are already the B operand of  O^T = V^T * P^T
// (k index = key, column = query) -- no cross-lane movement, no LDS round trip.  O^T tiles put 4 consecutive head
// dims of one query in a lane: one 8-byte hi + one 8-byte lo store into the packed-split attention output.
// All three products use the bf16x3 split (ribca_common.h).  Q is pre-scaled by hd^-0.5 by the producer.
#include "ribca_common.h"
#include "ribca_kernels.h"

namespace ribca {

template <int KS /* hdp/32 */, int DT /* hdv/16 */>
__global__ __launch_bounds__(256) void attention_kernel(const uint16_t* __restrict__ Q, const uint16_t* __restrict__ K,
                                                        const uint16_t* __restrict__ Vt, uint16_t* __restrict__ out, int ldo,
                                                        int n_pairs, int hd) {
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);  // (cell, head) index, wave-uniform
  if (pair >= n_pairs) return;
  const int cell = pair / kHeads, head = pair - cell * kHeads;
  const int r16 = lane & 15, g = lane >> 4;
  constexpr int ROW = 2 * KS * 32;          // bf16 per Q/K row
  constexpr int VROW = 2 * kKeyPad;         // bf16 per V^T row
  const uint16_t* qb = Q + (size_t)pair * kTokPad * ROW;
  const uint16_t* kb = K + (size_t)pair * kTokPad * ROW;
  const uint16_t* vb = Vt + (size_t)pair * (DT * 16) * VROW;

  // K fragments stay in registers for all 7 query tiles
  bf16x8 khi[7][KS], klo[7][KS];
#pragma unroll
  for (int kt = 0; kt < 7; ++kt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint4* p = reinterpret_cast<const uint4*>(kb + (size_t)(kt * 16 + r16) * ROW + ks * 64 + g * 16);
      khi[kt][ks] = __builtin_bit_cast(bf16x8, p[0]);
      klo[kt][ks] = __builtin_bit_cast(bf16x8, p[1]);
    }

  for (int qt = 0; qt < 7; ++qt) {
    bf16x8 qhi[KS], qlo[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint4* p = reinterpret_cast<const uint4*>(qb + (size_t)(qt * 16 + r16) * ROW + ks * 64 + g * 16);
      qhi[ks] = __builtin_bit_cast(bf16x8, p[0]);
      qlo[ks] = __builtin_bit_cast(bf16x8, p[1]);
    }
    f32x4 s[8];
#pragma unroll
    for (int kt = 0; kt < 7; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s[kt] = mfma_bf16(klo[kt][ks], qhi[ks], s[kt]);
        s[kt] = mfma_bf16(khi[kt][ks], qlo[ks], s[kt]);
        s[kt] = mfma_bf16(khi[kt][ks], qhi[ks], s[kt]);
      }
    }
    // keys 101..111 (tile 6, 4g+r >= 5) are padding
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * g + r >= kTokens - 96) s[6][r] = -INFINITY;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 7; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 7; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[kt][r] - mx);
        s[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    s[7] = f32x4{0.f, 0.f, 0.f, 0.f};
    // P fragments: k-step t covers key tiles 2t (elements 0..3) and 2t+1 (elements 4..7)
    bf16x8 phi[4], plo[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float pa[4], pb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { pa[r] = s[2 * t][r] * inv; pb[r] = s[2 * t + 1][r] * inv; }
      uint2 ha, la, hb, lb;
      split4(pa, ha, la);
      split4(pb, hb, lb);
      phi[t] = __builtin_bit_cast(bf16x8, uint4{ha.x, ha.y, hb.x, hb.y});
      plo[t] = __builtin_bit_cast(bf16x8, uint4{la.x, la.y, lb.x, lb.y});
    }
    const int qtok = qt * 16 + r16;
    uint16_t* orow = out + ((size_t)cell * kTokens + qtok) * ldo;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint4* p = reinterpret_cast<const uint4*>(vb + (size_t)(dt * 16 + r16) * VROW + (4 * t + g) * 16);
        const bf16x8 vhi = __builtin_bit_cast(bf16x8, p[0]);
        const bf16x8 vlo = __builtin_bit_cast(bf16x8, p[1]);
        o = mfma_bf16(vlo, phi[t], o);
        o = mfma_bf16(vhi, plo[t], o);
        o = mfma_bf16(vhi, phi[t], o);
      }
      const int d = dt * 16 + 4 * g;  // o[r] = O[query = qtok][head dim d + r]
      if (qtok < kTokens && d < hd) {
        float v[4] = {o[0], o[1], o[2], o[3]};
        ps_store4(orow, head * hd + d, v);
      }
    }
  }
}

void launch_attention(const uint16_t* q, const uint16_t* k, const uint16_t* vt, uint16_t* out, int ldo, int cells, int hd, int hdp,
                      int hdv, hipStream_t s) {
  const int pairs = cells * kHeads;
  if (pairs <= 0) return;
  const dim3 grid((pairs + 3) / 4), block(256);
  const int ks = hdp / 32, dt = hdv / 16;
#define RIBCA_ATT(KS_, DT_) \
  hipLaunchKernelGGL((attention_kernel<KS_, DT_>), grid, block, 0, s, q, k, vt, out, ldo, pairs, hd)
  if (ks == 1 && dt == 1) RIBCA_ATT(1, 1);
  else if (ks == 1 && dt == 2) RIBCA_ATT(1, 2);
  else if (ks == 2 && dt == 3) RIBCA_ATT(2, 3);
  else if (ks == 2 && dt == 4) RIBCA_ATT(2, 4);
  else if (ks == 1 && dt == 3) RIBCA_ATT(1, 3);
  else if (ks == 1 && dt == 4) RIBCA_ATT(1, 4);
  else abort();
#undef RIBCA_ATT
}

}  // namespace ribca
