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
// four consecutive values -> 8 bytes of hi and 8 bytes of lo.  Packed conversions: one v_cvt_pk_bf16_f32 per value pair.
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2;
typedef __attribute__((__vector_size__(2 * sizeof(float)))) float f32x2;
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split4(const float v[4], uint2& hi, uint2& lo) {
  hi.x = cvt_pk_bf16(v[0], v[1]);
  hi.y = cvt_pk_bf16(v[2], v[3]);
  const float r0 = v[0] - __uint_as_float(hi.x << 16), r1 = v[1] - __uint_as_float(hi.x & 0xFFFF0000u);
  const float r2 = v[2] - __uint_as_float(hi.y << 16), r3 = v[3] - __uint_as_float(hi.y & 0xFFFF0000u);
  lo.x = cvt_pk_bf16(r0, r1);
  lo.y = cvt_pk_bf16(r2, r3);
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

// same, for a lane pair (lane, lane ^ PX) that together owns the 8 columns of one PS group (k % 4 == 0; the even lane holds
// k % 8 == 0): the pair swaps halves so that each lane writes ONE 16-byte vector (all 8 hi, or all 8 lo) instead of two
// 8-byte pieces -- half the store instructions and 64 contiguous bytes per row and instruction.  Both lanes must be active.
typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t u32x4;
template <int PX>
__device__ __forceinline__ void ps_store4_pair(uint16_t* row, int k, const float v[4], bool nt = false) {
  uint2 hi, lo;
  split4(v, hi, lo);
  const bool odd = (k & 4) != 0;
  const uint2 send = odd ? hi : lo;
  uint2 recv;
  recv.x = __shfl_xor(send.x, PX, 64);
  recv.y = __shfl_xor(send.y, PX, 64);
  const u32x4 o = odd ? u32x4{recv.x, recv.y, lo.x, lo.y} : u32x4{hi.x, hi.y, recv.x, recv.y};
  u32x4* dst = reinterpret_cast<u32x4*>(row + ps_off(k & ~7) + (odd ? 8 : 0));
  if (nt) __builtin_nontemporal_store(o, dst);      // streamed once, read by a later launch: keep it out of this XCD's L2
  else *dst = o;
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

// erf to ~2e-7 absolute (Abramowitz & Stegun 7.1.26 evaluated in fp32): 1 rcp + 1 exp2 + 7 fma, branch-free.  The GELU
// output is re-quantised to a bf16 hi/lo pair (2^-17 relative) right after, so the libm erff's last bits would be
// discarded anyway while costing ~3x the instructions in the fc1 epilogue.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float r = fmaf(-p, e, 1.0f);
  return copysignf(r, x);
}
// exact-erf GELU (nn.GELU() default used by timm Mlp): 0.5 x (1 + erf(x / sqrt(2)))
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

}  // namespace ribca
