// Shared device helpers for the RIBCA hot-path kernels (gfx950 / CDNA4 only).
//
// Numerics ("fp16x3"): every matrix product on the path runs on the 16-bit matrix cores with each fp32 operand split as
// x = hi + lo (hi = fp16(x), lo = fp16(x - hi)) and three MFMA passes
//     acc += hi_a*hi_b ; acc += lo_a*hi_b ; acc += hi_a*lo_b        (fp32 accumulate)
// which carries ~22 mantissa bits per operand WHILE lo IS A NORMAL fp16, i.e. for |x| >= 2^-3: hi keeps 11 bits, lo (|lo| <= 2^-11 |x|)
// another 11.  Below that lo is subnormal (quantum 2^-24) and the split's error is ABSOLUTE, <= 2^-25 -- about 20 bits at the 0.02
// scale of typical weights, still 2^-25 / |x| relative.  The dropped lo*lo term is 2^-22 relative.  A single bf16 or fp16 pass misses
// the reference's 1e-3 confidence tolerance by 3-16x after 12 blocks; the round-1 bf16 split (16 bits) met it with 30x margin;
// the fp16 split costs the same three v_mfma_f32_16x16x32 passes and the same 4 bytes per element and is ~5x closer to the
// fp32 reference (tests/precision_study.py): max |dp| 4-6e-6 instead of 2-2.5e-5.
// Range: fp16 saturates at 65504, so values are clamped to +-65504 before the split: no inf can enter an MFMA, but an operand beyond
// that range SATURATES SILENTLY (and v_med3 maps a NaN operand to a finite value): the result is then finite and wrong rather than
// inf / NaN as in the fp32 reference.  Residual-stream values, LayerNorm outputs, GELU outputs and weights of a trained ViT sit
// orders of magnitude below (tests/test_gpu_kernels.py::test_split_operand_range pins the behaviour at 3e4 and beyond 65504).  lo is
// usually SUBNORMAL in fp16 (|lo| <= 2^-12 |x|): v_mfma_f32_16x16x32_f16 and the VALU keep fp16 subnormals on gfx950
// (tools/f16_probe.hip: exact), and lo of a subnormal-range hi is exactly zero.
//
// "Packed split" (PS) layout: a logical row of Kp elements (Kp % 32 == 0) is stored as 2*Kp fp16:
//     group g = k / 8 occupies 16 consecutive fp16: [hi(8g..8g+7) | lo(8g..8g+7)]
// so one lane's MFMA fragment (8 consecutive k of one row) is a 16-byte hi vector followed by a 16-byte lo
// vector, and a 32-deep K step of one row is one 128-byte line.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>

// Launch errors.  A host-side launcher that cannot run its kernel (a shape it has no form for, an attribute the runtime refuses) launches
// NOTHING and records why with launch_error(); the exported entry point that called it turns the record into a non-zero status whose text
// ribca_last_error() returns (ribca_api.hip: RIBCA_FINISH).  The library never ends the process itself: it is loaded into the caller's process
// (a napari worker, a Python interpreter), and the reference's own convention for a bad request is an exception the caller can catch
// (cell_type_annotation/model.py:636, 770), not process death.
namespace ribca {
void launch_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
}

// Raises a kernel's dynamic-LDS limit, once per (call site, device): a flag per call site alone would leave the second device of a process
// without the attribute (the launch would then fail with "invalid argument" -- or, worse, be dropped).  false = the runtime refused (recorded
// with launch_error): the caller must not launch -- a launch that silently does not happen leaves stale results behind, so the entry point
// reports the failure instead.  `done`: one bit per device ordinal (< 64), owned by the call site.
[[nodiscard]] inline bool ensure_dynamic_lds(const void* kernel, int bytes, unsigned long long& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev >= 0 && dev < 64 && ((done >> dev) & 1ull)) return true;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    ribca::launch_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed on device %d: %s", bytes, dev, hipGetErrorString(e));
    return false;
  }
  if (dev >= 0 && dev < 64) done |= 1ull << dev;
  return true;
}


namespace ribca {

typedef __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16 f16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

constexpr int kTokens = 101;   // 10x10 patches + CLS (reference model.py:66-88 with img_size=40, patch 4)
constexpr int kHeads = 12;
constexpr int kTokPad = 112;   // tokens padded to 7 MFMA tiles of 16 for attention operands
constexpr int kKeyPad = 128;   // keys padded to 4 MFMA K-steps of 32 for the P*V product

constexpr float kF16Max = 65504.0f;
__device__ __forceinline__ float clamp_f16_range(float x) { return __builtin_amdgcn_fmed3f(x, -kF16Max, kF16Max); }
__device__ __forceinline__ uint16_t f16_bits(float x) {
  return __builtin_bit_cast(uint16_t, (_Float16)x);
}
__device__ __forceinline__ float f16_to_f32(uint16_t b) {
  return (float)__builtin_bit_cast(_Float16, b);
}
// x = hi + lo split, both as fp16 bit patterns (round to nearest even; x clamped to the fp16 range first)
__device__ __forceinline__ void split_f16(float x, uint16_t& hi, uint16_t& lo) {
  x = clamp_f16_range(x);
  hi = f16_bits(x);
  lo = f16_bits(x - f16_to_f32(hi));
}
// four consecutive values -> 8 bytes of hi and 8 bytes of lo
typedef __attribute__((__vector_size__(2 * sizeof(_Float16)))) _Float16 f16x2;
typedef __attribute__((__vector_size__(2 * sizeof(float)))) float f32x2;
__device__ __forceinline__ uint32_t cvt_pk_f16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));     // round to nearest even
}
__device__ __forceinline__ f32x2 unpack_f16(uint32_t p) {
  return __builtin_convertvector(__builtin_bit_cast(f16x2, p), f32x2);
}
// Mixed-precision FMA (v_fma_mix_f32: every source is an fp32 register or one fp16 half of a register, the arithmetic is one fp32
// fma): "fp32 minus an fp16 half" and "fp16 half plus fp16 half" in ONE instruction where v_cvt_f32_f16 + v_sub / v_add took two or
// three.  Same values bit for bit: h * (-1) + x and h * 1 + l are exact products and a single rounding, as the separate convert +
// add is.  The split and the packed-split residual epilogue are VALU-bound on exactly these conversions.
__device__ __forceinline__ float f32_minus_f16lo(float x, uint32_t h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}
__device__ __forceinline__ float f32_minus_f16hi(float x, uint32_t h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}
__device__ __forceinline__ float f16lo_plus_f16lo(uint32_t a, uint32_t b) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float f16hi_plus_f16hi(uint32_t a, uint32_t b) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void split4(const float v[4], uint2& hi, uint2& lo) {
  const float c0 = clamp_f16_range(v[0]), c1 = clamp_f16_range(v[1]), c2 = clamp_f16_range(v[2]), c3 = clamp_f16_range(v[3]);
  hi.x = cvt_pk_f16(c0, c1);
  hi.y = cvt_pk_f16(c2, c3);
  lo.x = cvt_pk_f16(f32_minus_f16lo(c0, hi.x), f32_minus_f16hi(c1, hi.x));
  lo.y = cvt_pk_f16(f32_minus_f16lo(c2, hi.y), f32_minus_f16hi(c3, hi.y));
}
// the same split for values known to lie inside the fp16 range (softmax probabilities in [0, 1]): no clamp -- a quarter of split4's
// instructions
__device__ __forceinline__ void split4_unit(const float v[4], uint2& hi, uint2& lo) {
  hi.x = cvt_pk_f16(v[0], v[1]);
  hi.y = cvt_pk_f16(v[2], v[3]);
  lo.x = cvt_pk_f16(f32_minus_f16lo(v[0], hi.x), f32_minus_f16hi(v[1], hi.x));
  lo.y = cvt_pk_f16(f32_minus_f16lo(v[2], hi.y), f32_minus_f16hi(v[3], hi.y));
}
// element offset (in fp16 units) of the hi part of logical column k in a PS row; lo part is +8
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
  if constexpr (PX == 1) {
    // neighbour lane inside a quad: one DPP move (quad_perm [1,0,3,2]) instead of a ds_bpermute round trip through the LDS unit
    recv.x = (uint32_t)__builtin_amdgcn_mov_dpp((int)send.x, 0xB1, 0xF, 0xF, true);
    recv.y = (uint32_t)__builtin_amdgcn_mov_dpp((int)send.y, 0xB1, 0xF, 0xF, true);
  } else if constexpr (PX == 16) {
    // partner = the same lane of the neighbouring 16-lane row: v_permlane16_swap_b32 (gfx950) exchanges the odd rows of its first
    // operand with the even rows of its second, which is this hi/lo exchange in ONE VALU instruction per register -- no
    // ds_bpermute round trip through the LDS unit (two per tile, each waited for, in the register epilogues)
    const auto rx = __builtin_amdgcn_permlane16_swap(hi.x, lo.x, false, false);
    const auto ry = __builtin_amdgcn_permlane16_swap(hi.y, lo.y, false, false);
    const u32x4 o = {rx[0], ry[0], rx[1], ry[1]};      // even row: 8 x hi, odd row: 8 x lo
    u32x4* dst = reinterpret_cast<u32x4*>(row + ps_off(k & ~7) + (odd ? 8 : 0));
    if (nt) __builtin_nontemporal_store(o, dst);
    else *dst = o;
    return;
  } else {
    recv.x = __shfl_xor(send.x, PX, 64);
    recv.y = __shfl_xor(send.y, PX, 64);
  }
  const u32x4 o = odd ? u32x4{recv.x, recv.y, lo.x, lo.y} : u32x4{hi.x, hi.y, recv.x, recv.y};
  u32x4* dst = reinterpret_cast<u32x4*>(row + ps_off(k & ~7) + (odd ? 8 : 0));
  if (nt) __builtin_nontemporal_store(o, dst);      // streamed once, read by a later launch: keep it out of this XCD's L2
  else *dst = o;
}

__device__ __forceinline__ f32x4 mfma_f16(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
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

// exact-erf GELU (nn.GELU() default used by timm Mlp): 0.5 x (1 + erf(x / sqrt(2))).
// erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7) on z = |x| / sqrt(2):  erf(z) = 1 - q,  q = poly(t) exp(-z^2),
// t = 1 / (1 + 0.3275911 z).  With a = |x|:   gelu = 0.5 x + a (0.5 - 0.5 q)  -- no sign select, no branch.
// The output is re-quantised to an fp16 hi/lo pair right after, so libm erff's last bits would be discarded anyway.
// One fp32 instruction per value and step (round 6).  Rounds 2-5 evaluated two values at a time in PACKED fp32 (v_pk_fma_f32 /
// v_pk_mul_f32: 8.5 instructions per value instead of 15), which was faster while the GELU epilogue ran on a CU of its own; since the
// GELU GEMMs run two workgroups per CU, one workgroup's epilogue shares its SIMDs with the other's MFMA stream, and a packed fp32
// operation beside MFMAs costs more than the two plain ones it replaces (MI355X_MICROARCH.md, per-instruction table: "an anti-lever
// beside MFMAs").  Same-box A/B, per 4096 cells (profiles/r6/ab_gelu_scalar.txt): fc1 at D = 288 11.9 -> 11.3 ms, at D = 384 15.6 ->
// 15.1-15.4, at D = 576 29.5 -> 28.9-29.3.  RIBCA_GELU_PACKED restores the packed form (A/B); the library is also built with
// -fno-slp-vectorize so that the compiler does not pack the scalar form again.
// A pair of fp32 values.  Rounds 2-5 made it a hardware vector (ext_vector_type: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, two values per
// instruction) in the softmax, fold and statistics code; beside an MFMA stream -- the wave's own in the attention kernels, the other
// workgroup's in the two-workgroups-per-CU GEMMs -- a packed fp32 operation costs more than the two plain ones it replaces, so the pair is
// a plain struct now and every operation on it one scalar instruction per value (same IEEE operations, same results; -DRIBCA_PACKED_F32
// restores the vector form for A/B, profiles/r6/ab_packed_f32.txt).
// (f32x2p: always the hardware vector -- the residual epilogues' statistics code keeps it: as a struct it costs the 128 x 192 residual kernel
// four spilled registers, which a kernel that fills registers with inline-asm loads must not have, tests/test_kernel_resources.py)
typedef float f32x2p __attribute__((ext_vector_type(2)));
#if defined(RIBCA_PACKED_F32) || defined(RIBCA_GELU_PACKED)
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2v fma2(f32x2v a, f32x2v b, f32x2v c) { return __builtin_elementwise_fma(a, b, c); }
#else
struct f32x2v {
  float x, y;
  __device__ __forceinline__ f32x2v& operator+=(const f32x2v& o) { x += o.x; y += o.y; return *this; }
};
__device__ __forceinline__ f32x2v operator+(const f32x2v& a, const f32x2v& b) { return f32x2v{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f32x2v operator-(const f32x2v& a, const f32x2v& b) { return f32x2v{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ f32x2v operator*(const f32x2v& a, const f32x2v& b) { return f32x2v{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f32x2v fma2(const f32x2v& a, const f32x2v& b, const f32x2v& c) {
  return f32x2v{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)};
}
#endif
#ifndef RIBCA_GELU_PACKED
__device__ __forceinline__ float gelu_erf1(float x) {
  const float a = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(a, 0.23164188814f, 1.0f));      // 0.3275911 / sqrt(2)
  float p = __builtin_fmaf(t, 0.5307027145f, -0.7265760135f);      // 0.5 x (1.061405429, -1.453152027, 1.421413741, -0.284496736, 0.254829592)
  p = __builtin_fmaf(p, t, 0.7107068705f);
  p = __builtin_fmaf(p, t, -0.142248368f);
  p = __builtin_fmaf(p, t, 0.127414796f);
  p = p * t;
  const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044f);      // -log2(e) / 2
  const float s = __builtin_fmaf(-p, e, 0.5f);                           // 0.5 erfc-complement: 0.5 - 0.5 q
  return __builtin_fmaf(a, s, x * 0.5f);
}
__device__ __forceinline__ f32x2v gelu_erf2(f32x2v x) { return f32x2v{gelu_erf1(x.x), gelu_erf1(x.y)}; }
#else
__device__ __forceinline__ f32x2v gelu_erf2(f32x2v x) {
  f32x2v a;
  a.x = fabsf(x.x); a.y = fabsf(x.y);
  const f32x2v den = a * 0.23164188814f + 1.0f;
  f32x2v t;
  t.x = __builtin_amdgcn_rcpf(den.x); t.y = __builtin_amdgcn_rcpf(den.y);
  f32x2v p = t * 0.5307027145f + -0.7265760135f;
  p = p * t + 0.7107068705f;
  p = p * t + -0.142248368f;
  p = p * t + 0.127414796f;
  p = p * t;
  const f32x2v ea = (x * x) * -0.72134752044f;
  f32x2v e;
  e.x = __builtin_amdgcn_exp2f(ea.x); e.y = __builtin_amdgcn_exp2f(ea.y);
  const f32x2v s = 0.5f - p * e;
  return a * s + x * 0.5f;
}
#endif
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf2(f32x2v{x, x}).x; }

}  // namespace ribca
