// Shared device helpers for the RIBCA hot-path kernels (gfx950 / CDNA4 only).
//
// Numerics ("bf16x3"): every matrix product on the path runs on the bf16 matrix cores with each fp32
// operand split as x = hi + lo (hi = bf16(x), lo = bf16(x - hi)) and three MFMA passes
//     acc += hi_a*hi_b ; acc += lo_a*hi_b ; acc += hi_a*lo_b        (fp32 accumulate)
// which carries ~16 mantissa bits per operand.  A single bf16 or fp16 pass misses the reference's 1e-3
// confidence tolerance by 3-16x after 12 blocks (DESIGN.md "precision study"); the split form meets it with
// 30x margin at 3x the MFMA work.
//
// "Packed split" (PS) layout: a logical row of Kp elements (Kp % 32 == 0) is stored as 2*Kp bf16:
//     group g = k / 8 occupies 16 consecutive bf16: [hi(8g..8g+7) | lo(8g..8g+7)]
// so one lane's MFMA fragment (8 consecutive k of one row) is a 16-byte hi vector followed by a 16-byte lo
// vector, and a 32-deep K step of one row is one 128-byte line.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ribca {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

constexpr int kTokens = 101;   // 10x10 patches + CLS (reference model.py:66-88 with img_size=40, patch 4)
constexpr int kHeads = 12;
constexpr int kTokPad = 112;   // tokens padded to 7 MFMA tiles of 16 for attention operands
constexpr int kKeyPad = 128;   // keys padded to 4 MFMA K-steps of 32 for the P*V product

__device__ __forceinline__ uint16_t bf16_bits(float x) {
  return __builtin_bit_cast(uint16_t, (__bf16)x);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t b) {
  return __uint_as_float(((uint32_t)b) << 16);
}
// x = hi + lo split, both as bf16 bit patterns
__device__ __forceinline__ void split_bf16(float x, uint16_t& hi, uint16_t& lo) {
  hi = bf16_bits(x);
  lo = bf16_bits(x - bf16_to_f32(hi));
}
// four consecutive values -> 8 bytes of hi and 8 bytes of lo
__device__ __forceinline__ void split4(const float v[4], uint2& hi, uint2& lo) {
  uint16_t h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split_bf16(v[i], h[i], l[i]);
  hi.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
  hi.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
  lo.x = (uint32_t)l[0] | ((uint32_t)l[1] << 16);
  lo.y = (uint32_t)l[2] | ((uint32_t)l[3] << 16);
}
// element offset (in bf16 units) of the hi part of logical column k in a PS row; lo part is +8
__device__ __host__ __forceinline__ int ps_off(int k) { return ((k >> 3) << 4) + (k & 7); }

// store 4 consecutive logical columns k..k+3 (k % 4 == 0) of a PS row
__device__ __forceinline__ void ps_store4(uint16_t* row, int k, const float v[4]) {
  uint2 hi, lo;
  split4(v, hi, lo);
  uint16_t* p = row + ps_off(k);
  *reinterpret_cast<uint2*>(p) = hi;
  *reinterpret_cast<uint2*>(p + 8) = lo;
}

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact (erf) GELU, as nn.GELU() default used by timm Mlp
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

}  // namespace ribca
