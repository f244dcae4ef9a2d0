// Round-4 hardware probe behind the "fp16 hi*hi + block-scaled correction passes" GEMM (gfx950 only).
//
// tools/mx_probe.hip (round 2) pinned the fp6 MFMA's lane map, field order and scale byte.  This probe pins what the production
// kernel additionally relies on, and prices the candidate instruction mixes of one wave's 128-deep K step:
//   A. the whole scheme on one 16 x 16 tile over K = 256, every operand in the layout the kernel uses:
//        acc  = sum_s mfma_f16(Wh[s], Ah[s])                       4 x v_mfma_f32_16x16x32_f16 per 128 k, lane (r, g) holding
//                                                                   k = 128 b + 32 g + 8 s + j  (32 CONSECUTIVE k per lane over the 4 sub-steps)
//        acc += mfma_scale(W h' [fp8 e4m3], A lo [fp8 e4m3])       block-scaled, lane (r, g) = the same 32 k
//        acc += mfma_scale(W lo [fp6 e2m3], A h' [fp6 e2m3])       A h' made IN REGISTERS from the 16 dwords of Ah by v_cvt_scalef32_pk32_fp6_f16
//      against (i) a host emulation of exactly these roundings (pins formats, k order, scale direction, op_sel bytes: must agree to
//      fp32 summation noise) and (ii) the fp64 product (the error the scheme really has);
//      (Part A multiplies fp8 x fp8 and fp6 x fp6, where both operands of an instruction share one lane map, so any consistent k order
//      passes.  The production kernel MIXES an fp6 and an fp8 operand in one instruction; their lane maps differ -- an fp8 operand's lane
//      (r, g) holds k = 16 g .. + 15 and 64 + 16 g .. + 15 -- which tools/mx_kmap_probe.hip pinned after the first kernel was 3.5e-5 off.)
//   B. v_cvt_scalef32_pk_fp8_f32 (what a producing epilogue uses for the lo byte): RNE(x / scale) onto e4m3, word select; measured: it does
//      NOT saturate (|x / scale| > 448 gives the NaN code 0x7f / 0xff: 2 of 256 probe values), hence the format's scale rule (lo / scale <= 256);
//   C. issue cost of one wave-step (8 row tiles x 3 column tiles x 128 k) for the candidate mixes, 1 and 2 waves per SIMD, whole chip,
//      random operands: today's 12 f16 MFMAs per tile; 4 f16 + 2 fp6; 4 f16 + fp8 + fp6; the same with the 8 (A) and 11 (A + W) in-loop
//      pk32 conversions; the bf8-by-v_perm alternative for A h'.
//   hipcc -O3 --offload-arch=gfx950 tools/mx_mix_probe.hip -o gpurun_out/mx_mix_probe && gpurun_out/mx_mix_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(6))) unsigned u32x6;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(32))) _Float16 h32;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(2))) short s16x2;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------------------------------------ host codecs
static double e2m3_value(int code) {
  const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
  const double v = e == 0 ? m / 8.0 : (1.0 + m / 8.0) * (double)(1 << (e - 1));
  return s ? -v : v;
}
static int e2m3_encode(double x) {   // saturating RNE
  const int s = x < 0;
  double a = fabs(x);
  if (a > 7.5) a = 7.5;
  int best = 0; double bd = 1e30;
  for (int c = 0; c < 32; ++c) {
    const double d = fabs(e2m3_value(c) - a);
    if (d < bd || (d == bd && (c & 1) == 0)) { bd = d; best = c; }
  }
  return (s << 5) | best;
}
static double e4m3_value(int code) {   // OCP e4m3fn: bias 7, no inf, 0x7f = NaN, max 448
  const int s = (code >> 7) & 1, e = (code >> 3) & 15, m = code & 7;
  const double v = e == 0 ? ldexp(m / 8.0, -6) : ldexp(1.0 + m / 8.0, e - 7);
  return s ? -v : v;
}
static int e4m3_encode(double x) {   // saturating RNE (to 448)
  const int s = x < 0;
  double a = fabs(x);
  if (a > 448.0) a = 448.0;
  int best = 0; double bd = 1e30;
  for (int c = 0; c < 127; ++c) {
    const double d = fabs(e4m3_value(c) - a);
    if (d < bd || (d == bd && (c & 1) == 0)) { bd = d; best = c; }
  }
  return (s << 7) | best;
}
static float f16_round(float x) { return (float)(_Float16)x; }
static int floor_log2(double a) { int e; frexp(a, &e); return e - 1; }   // a > 0

// ------------------------------------------------------------------------------------------------ A: the scheme on one tile
// per lane and 128-k block b: Ah[4] / Wh[4] f16 fragments, A lo fp8 (8 dwords), W h' fp8 (8 dwords), W lo fp6 (6 dwords),
// scale word per operand side: byte 0 = lo scale, byte 1 = h' scale (E8M0)
struct LaneOps {
  h8 ah[2][4], wh[2][4];
  i32x8 al8[2], wh8[2], wl6[2], wh6[2];
  int a_sc[2], w_sc[2];
};
template <int MIXED>
__global__ void scheme_kernel(const LaneOps* ops, f32x4* out, u32x6* h6_out) {
  const LaneOps o = ops[threadIdx.x];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < 2; ++b) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.wh[b][s], o.ah[b][s], acc, 0, 0, 0);
    h32 hv;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) hv[8 * s + j] = o.ah[b][s][j];
    const float sh = __builtin_bit_cast(float, (unsigned)((o.a_sc[b] >> 8) & 0xff) << 23);
    const u32x6 h6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hv, sh);
    if (b == 0) h6_out[threadIdx.x] = h6;
    i32x8 ah6 = {(int)h6[0], (int)h6[1], (int)h6[2], (int)h6[3], (int)h6[4], (int)h6[5], 0, 0};
    // Term 2: W h' (fp8, scale byte 1 of w_sc) x A lo (fp8, scale byte 0 of a_sc)
    if (MIXED) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(o.wh6[b], o.al8[b], acc, 2, 0, 2, o.w_sc[b], 0, o.a_sc[b]);      // W hi as fp6 (scale byte 2) x A lo fp8
    else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(o.wh8[b], o.al8[b], acc, 0, 0, 1, o.w_sc[b], 0, o.a_sc[b]);
    // Term 3: W lo (fp6, scale byte 0 of w_sc) x A h' (fp6, scale byte 1 of a_sc)
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(o.wl6[b], ah6, acc, 2, 2, 0, o.w_sc[b], 1, o.a_sc[b]);
  }
  out[threadIdx.x] = acc;
}

// ------------------------------------------------------------------------------------------------ B: the epilogue's fp8 conversion
__global__ void cvt8_kernel(const float* x, const float* scale, unsigned* out) {
  const int t = threadIdx.x;
  s16x2 r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x[4 * t], x[4 * t + 1], scale[t], false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x[4 * t + 2], x[4 * t + 3], scale[t], true);
  out[t] = __builtin_bit_cast(unsigned, r);
}

// ------------------------------------------------------------------------------------------------ C: rates
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
  return x;
}
template <int MODE>
__global__ __launch_bounds__(512) void mix_kernel(unsigned long long* stamps, float* sink, int iters) {
  constexpr int MT = 8, NT = 3;
  // operands: hash-random, fp16 values in [-2, 2), 8-bit / 6-bit fields random.  One register set per operand kind (the production kernel
  // streams them): every use passes through an empty asm so that nothing derived from them can be hoisted out of the loops.
  h8 wh[4], ahp[4];
  i32x8 w8, w6, a8;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      wh[s][e] = (_Float16)(((int)(hash32(threadIdx.x * 131u + s * 31u + e) & 0xffffu) - 32768) * (1.0f / 16384.0f));
      ahp[s][e] = (_Float16)(((int)(hash32(threadIdx.x * 257u + s * 61u + e + 77u) & 0xffffu) - 32768) * (1.0f / 16384.0f));
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    w8[e] = (int)(hash32(threadIdx.x * 17u + e + 1000u) & 0x3f3f3f3fu);     // small fp8 magnitudes (no NaN code)
    w6[e] = (int)hash32(threadIdx.x * 19u + e + 2000u);
    a8[e] = (int)(hash32(threadIdx.x * 23u + e + 3000u) & 0x3f3f3f3fu);
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int sc = 0x7f7f7f7f;
  unsigned cvsink = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    i32x8 w6c[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) w6c[j] = w6;
    if (MODE == 4) {      // W h' made in the loop as well: 3 more conversions per wave-step
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(wh[s]));
        h32 hv;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e) hv[8 * s + e] = wh[s][e];
        const u32x6 c = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hv, 1.0f);
        w6c[j] = i32x8{(int)c[0], (int)c[1], (int)c[2], (int)c[3], (int)c[4], (int)c[5], 0, 0};
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      h8 ah[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) { asm volatile("" : "+v"(ahp[s])); ah[s] = ahp[s]; }      // "a new fragment from LDS"
      asm volatile("" : "+v"(a8));
      i32x8 ah6 = a8;
      if (MODE == 3 || MODE == 4 || MODE == 7) {
        h32 hv;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e) hv[8 * s + e] = ah[s][e];
        const u32x6 c = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hv, 1.0f);
        ah6 = i32x8{(int)c[0], (int)c[1], (int)c[2], (int)c[3], (int)c[4], (int)c[5], 0, 0};
        if (MODE == 7) { asm volatile("" ::"v"(ah6)); continue; }      // the conversion alone
      }
      if (MODE == 5) {      // bf8 = rounded high byte of each fp16: v_pk_add_u16 + v_perm_b32 per two dwords
        typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const uint4 v = __builtin_bit_cast(uint4, ah[s]);
          unsigned q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] = __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, q[e]) + (u16x2){0x80, 0x80}));
          ah6[2 * s] = (int)__builtin_amdgcn_perm(q[1], q[0], 0x07050301u);
          ah6[2 * s + 1] = (int)__builtin_amdgcn_perm(q[3], q[2], 0x07050301u);
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (MODE == 0) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], ah[s], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(s + 1) & 3], ah[s], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], ah[(s + 1) & 3], acc[i][j], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(s + j) & 3], ah[s], acc[i][j], 0, 0, 0);
          if (MODE == 1) {
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w6c[j], a8, acc[i][j], 2, 2, 0, sc, 0, sc);
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w6c[j], ah6, acc[i][j], 2, 2, 0, sc, 0, sc);
          } else if (MODE == 6) {
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8, a8, acc[i][j], 0, 0, 0, sc, 0, sc);
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8, ah6, acc[i][j], 0, 0, 0, sc, 0, sc);
          } else if (MODE == 5) {
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8, a8, acc[i][j], 0, 0, 0, sc, 0, sc);
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w6c[j], ah6, acc[i][j], 2, 1, 0, sc, 0, sc);
          } else {      // 2, 3, 4: fp8 x fp8 + fp6 x fp6
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8, a8, acc[i][j], 0, 0, 0, sc, 0, sc);
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w6c[j], ah6, acc[i][j], 2, 2, 0, sc, 0, sc);
          }
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  float r = (float)cvsink;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) r += acc[i][j][0] + acc[i][j][3];
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = c1 - c0;
  }
  if (r == 12345.678f) sink[0] = r;
}

template <int MODE>
static void run_mix(const char* name, int waves_per_simd, int mfma_cycles_ideal) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int threads = 256 * waves_per_simd;
  const int waves = cus * threads / 64;
  unsigned long long* stamps; float* sink;
  CK(hipMalloc(&stamps, sizeof(unsigned long long) * 2 * waves));
  CK(hipMalloc(&sink, 64));
  const int iters = MODE == 7 ? 20000 : 3000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mix_kernel<MODE>, dim3(cus), dim3(threads), 0, 0, stamps, sink, 50);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(mix_kernel<MODE>, dim3(cus), dim3(threads), 0, 0, stamps, sink, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * waves);
  CK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost));
  double rt = 0, cy = 0;
  for (int i = 0; i < waves; ++i) { rt += (double)h[2 * i]; cy += (double)h[2 * i + 1]; }
  rt /= waves; cy /= waves;
  const double ghz = cy / (rt * 10.0);          // s_memrealtime ticks are 10 ns
  const double cyc_step = cy / iters;           // shader cycles per wave-step as seen by one wave
  const double per_simd = cyc_step / waves_per_simd;
  printf("  %-44s %d w/SIMD  %8.1f cyc per wave-step (%7.1f per SIMD and step; MFMA-only ideal %5d)  %5.3f GHz  %7.3f ms  %6.3f us per step and SIMD\n", name,
         waves_per_simd, cyc_step, per_simd, mfma_cycles_ideal, ghz, ms, ms * 1e3 / iters / waves_per_simd);
  CK(hipFree(stamps)); CK(hipFree(sink));
}

int main() {
  srand(20261004);
  // ---------------------------------------------------------------- A
  {
    const int K = 256;
    std::vector<float> A(16 * K), W(16 * K);
    for (int r = 0; r < 16; ++r) {
      const float rs = ldexpf(1.0f, (r % 7) - 3);
      for (int k = 0; k < K; ++k) {
        // activations: GELU-like mix of small and large values; weights ~ 0.05
        float u = (rand() % 20001 - 10000) / 10000.0f, v = (rand() % 20001 - 10000) / 10000.0f;
        A[r * K + k] = rs * (u * u * u * 4.0f + 0.01f * v);
        W[r * K + k] = 0.05f * (float)((rand() % 20001 - 10000) / 10000.0) + 0.003f * v;
      }
    }
    std::vector<LaneOps> ops(64);
    // host emulation accumulators
    std::vector<double> emu(16 * 16, 0.0), ref(16 * 16, 0.0), t1only(16 * 16, 0.0), emu6(16 * 16, 0.0);
    std::vector<float> Ah(16 * K), Al(16 * K), Wh(16 * K), Wl(16 * K);
    for (int i = 0; i < 16 * K; ++i) {
      Ah[i] = f16_round(A[i]); Al[i] = A[i] - Ah[i];
      Wh[i] = f16_round(W[i]); Wl[i] = W[i] - Wh[i];
    }
    // dequantised correction operands for the emulation
    std::vector<double> Al_q(16 * K), Ah_q(16 * K), Wl_q(16 * K), Wh_q(16 * K), Wh6_q(16 * K);
    for (int l = 0; l < 64; ++l) {
      const int r = l & 15, g = l >> 4;
      LaneOps& o = ops[l];
      memset(&o, 0, sizeof(o));
      for (int b = 0; b < 2; ++b) {
        const int k0 = 128 * b + 32 * g;
        for (int s = 0; s < 4; ++s)
          for (int j = 0; j < 8; ++j) {
            o.ah[b][s][j] = (_Float16)Ah[r * K + k0 + 8 * s + j];
            o.wh[b][s][j] = (_Float16)Wh[r * K + k0 + 8 * s + j];
          }
        // block exponents from the hi parts (what a producer has at hand): E = floor(log2 max|hi|)
        double ma = 0, mw = 0;
        for (int f = 0; f < 32; ++f) { ma = fmax(ma, fabs((double)Ah[r * K + k0 + f])); mw = fmax(mw, fabs((double)Wh[r * K + k0 + f])); }
        const int Ea = ma > 0 ? floor_log2(ma) : -20, Ew = mw > 0 ? floor_log2(mw) : -20;
        // fp6 h': max / 2^(E-2) in [4, 8) (saturates at 7.5);  fp8 lo: |lo| <= 2^(E-11) -> / 2^(E-19) <= 256 < 448;  fp6 lo: / 2^(E-13) <= 4
        // fp8 h' (W): max / 2^(E-7) in [128, 256) < 448
        const int a_sh = Ea - 2 + 127, a_sl = Ea - 19 + 127, w_sh = Ew - 7 + 127, w_sl = Ew - 13 + 127;
        o.a_sc[b] = (a_sl & 0xff) | ((a_sh & 0xff) << 8) | (0x55 << 16) | (0x66 << 24);
        const int w_sh6 = Ew - 2 + 127;
        o.w_sc[b] = (w_sl & 0xff) | ((w_sh & 0xff) << 8) | ((w_sh6 & 0xff) << 16) | (0x22 << 24);
        unsigned al8[8] = {0}, wh8[8] = {0}, wl6[8] = {0}, wh6[8] = {0};
        for (int f = 0; f < 32; ++f) {
          const int k = k0 + f;
          const int ca = e4m3_encode((double)Al[r * K + k] / ldexp(1.0, a_sl - 127));
          al8[f >> 2] |= (unsigned)ca << (8 * (f & 3));
          Al_q[r * K + k] = e4m3_value(ca) * ldexp(1.0, a_sl - 127);
          const int cw = e4m3_encode((double)Wh[r * K + k] / ldexp(1.0, w_sh - 127));
          wh8[f >> 2] |= (unsigned)cw << (8 * (f & 3));
          Wh_q[r * K + k] = e4m3_value(cw) * ldexp(1.0, w_sh - 127);
          const int cl = e2m3_encode((double)Wl[r * K + k] / ldexp(1.0, w_sl - 127));
          const int bit = 6 * f;
          wl6[bit >> 5] |= (unsigned)cl << (bit & 31);
          if ((bit & 31) > 26) wl6[(bit >> 5) + 1] |= (unsigned)cl >> (32 - (bit & 31));
          Wl_q[r * K + k] = e2m3_value(cl) * ldexp(1.0, w_sl - 127);
          const int ch6 = e2m3_encode((double)Wh[r * K + k] / ldexp(1.0, w_sh6 - 127));
          wh6[bit >> 5] |= (unsigned)ch6 << (bit & 31);
          if ((bit & 31) > 26) wh6[(bit >> 5) + 1] |= (unsigned)ch6 >> (32 - (bit & 31));
          Wh6_q[r * K + k] = e2m3_value(ch6) * ldexp(1.0, w_sh6 - 127);
          const int ch = e2m3_encode((double)Ah[r * K + k] / ldexp(1.0, a_sh - 127));
          Ah_q[r * K + k] = e2m3_value(ch) * ldexp(1.0, a_sh - 127);
        }
        for (int e = 0; e < 8; ++e) { o.al8[b][e] = (int)al8[e]; o.wh8[b][e] = (int)wh8[e]; o.wl6[b][e] = (int)wl6[e]; o.wh6[b][e] = (int)wh6[e]; }
      }
    }
    for (int n = 0; n < 16; ++n)
      for (int m = 0; m < 16; ++m) {
        double e = 0, rr = 0, t1 = 0, e6 = 0;
        for (int k = 0; k < K; ++k) {
          e += (double)Wh[n * K + k] * Ah[m * K + k] + Wh_q[n * K + k] * Al_q[m * K + k] + Wl_q[n * K + k] * Ah_q[m * K + k];
          t1 += (double)Wh[n * K + k] * Ah[m * K + k];
          e6 += (double)Wh[n * K + k] * Ah[m * K + k] + Wh6_q[n * K + k] * Al_q[m * K + k] + Wl_q[n * K + k] * Ah_q[m * K + k];
          rr += (double)W[n * K + k] * (double)A[m * K + k];
        }
        emu[n * 16 + m] = e; ref[n * 16 + m] = rr; t1only[n * 16 + m] = t1; emu6[n * 16 + m] = e6;
      }
    LaneOps* dops; f32x4* dout; u32x6* dh6;
    CK(hipMalloc(&dops, 64 * sizeof(LaneOps))); CK(hipMalloc(&dout, 64 * sizeof(f32x4))); CK(hipMalloc(&dh6, 64 * sizeof(u32x6)));
    CK(hipMemcpy(dops, ops.data(), 64 * sizeof(LaneOps), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(scheme_kernel<0>, dim3(1), dim3(64), 0, 0, dops, dout, dh6);
    CK(hipDeviceSynchronize());
    std::vector<f32x4> hc(64);
    CK(hipMemcpy(hc.data(), dout, 64 * sizeof(f32x4), hipMemcpyDeviceToHost));
    double e_emu = 0, e_ref = 0, e_t1 = 0, mag = 0;
    for (int l = 0; l < 64; ++l)
      for (int reg = 0; reg < 4; ++reg) {
        // acc = mfma(W as the first operand, A as the second): row index = W row (output column n), column = A row m
        const int m = l & 15, n = 4 * (l >> 4) + reg;
        const double got = (double)hc[l][reg];
        e_emu = fmax(e_emu, fabs(got - emu[n * 16 + m]));
        e_ref = fmax(e_ref, fabs(got - ref[n * 16 + m]));
        e_t1 = fmax(e_t1, fabs(t1only[n * 16 + m] - ref[n * 16 + m]));
        mag = fmax(mag, fabs(ref[n * 16 + m]));
      }
    printf("A: scheme on a 16x16 tile, K = 256 (max |ref| %.4f):\n", mag);
    printf("   device vs host emulation of the same roundings : max abs %.3e  (fp32 summation noise expected, ~1e-6 x |ref|)\n", e_emu);
    printf("   device vs fp64 product                         : max abs %.3e  = %.3e of max |ref|\n", e_ref, e_ref / mag);
    printf("   hi*hi alone vs fp64 product                    : max abs %.3e  = %.3e of max |ref|\n", e_t1, e_t1 / mag);
    // the kernel's own Term 2: W hi as fp6 (6 dwords) x A lo as fp8 (8 dwords) -- mixed operand formats
    hipLaunchKernelGGL(scheme_kernel<1>, dim3(1), dim3(64), 0, 0, dops, dout, dh6);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hc.data(), dout, 64 * sizeof(f32x4), hipMemcpyDeviceToHost));
    double e_emu6 = 0, e_ref6 = 0;
    for (int l = 0; l < 64; ++l)
      for (int reg = 0; reg < 4; ++reg) {
        const int m = l & 15, n = 4 * (l >> 4) + reg;
        e_emu6 = fmax(e_emu6, fabs((double)hc[l][reg] - emu6[n * 16 + m]));
        e_ref6 = fmax(e_ref6, fabs((double)hc[l][reg] - ref[n * 16 + m]));
      }
    printf("   mixed formats (W hi fp6 x A lo fp8): device vs its emulation max abs %.3e, vs fp64 %.3e = %.3e of max |ref|\n", e_emu6, e_ref6, e_ref6 / mag);
    CK(hipFree(dops)); CK(hipFree(dout)); CK(hipFree(dh6));
  }
  // ---------------------------------------------------------------- B
  {
    std::vector<float> x(64 * 4), sc(64);
    for (int t = 0; t < 64; ++t) {
      sc[t] = ldexpf(1.0f, (t % 11) - 5);
      for (int i = 0; i < 4; ++i) x[4 * t + i] = ((rand() % 20001) - 10000) / 10000.0f * 300.0f * sc[t] * ((i & 1) ? 0.01f : 1.0f);
      if (t == 0) { x[0] = 1000.0f * sc[t]; x[1] = -1000.0f * sc[t]; x[2] = 0.0009765625f * sc[t]; x[3] = 17.0f * sc[t]; }   // saturation, subnormal tie, tie 16|18
    }
    float *dx, *ds; unsigned* dout;
    CK(hipMalloc(&dx, 256 * 4)); CK(hipMalloc(&ds, 256)); CK(hipMalloc(&dout, 256));
    CK(hipMemcpy(dx, x.data(), 256 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, sc.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt8_kernel, dim3(1), dim3(64), 0, 0, dx, ds, dout);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> ho(64);
    CK(hipMemcpy(ho.data(), dout, 256, hipMemcpyDeviceToHost));
    int bad = 0, shown = 0;
    for (int t = 0; t < 64; ++t)
      for (int i = 0; i < 4; ++i) {
        const int c = (ho[t] >> (8 * i)) & 0xff;
        const int want = e4m3_encode((double)x[4 * t + i] / sc[t]);
        const bool zero_ok = e4m3_value(c) == 0 && e4m3_value(want) == 0;
        if (c != want && !zero_ok) {
          ++bad;
          if (shown++ < 8) printf("   cvt8: x=%g scale=%g -> 0x%02x (%g), RNE(x/scale) would be 0x%02x (%g)\n", x[4 * t + i], sc[t], c, e4m3_value(c), want, e4m3_value(want));
        }
      }
    printf("B: v_cvt_scalef32_pk_fp8_f32 (byte i of the dword = value i; word select false/true = low/high half): mismatches vs saturating RNE(x / scale) %d of 256\n", bad);
    CK(hipFree(dx)); CK(hipFree(ds)); CK(hipFree(dout));
  }
  // ---------------------------------------------------------------- C
  printf("C: one wave-step = 8 row tiles x 3 column tiles x 128 k; cycles from s_memtime, clock = s_memtime / s_memrealtime\n");
  for (int w = 1; w <= 2; ++w) {
    run_mix<0>("12 x f16 per tile (today)", w, 24 * 12 * 16);
    run_mix<1>("4 x f16 + 2 x (fp6 x fp6)", w, 24 * 96);
    run_mix<2>("4 x f16 + fp8 x fp8 + fp6 x fp6", w, 24 * 112);
    run_mix<3>("  + 8 x cvt_pk32_fp6_f16 (A h' in the loop)", w, 24 * 112);
    run_mix<4>("  + 11 x cvt_pk32_fp6_f16 (A and W h')", w, 24 * 112);
    run_mix<5>("4 x f16 + fp8 x fp8 + fp6 x bf8(v_perm of hi)", w, 24 * 128);
    run_mix<6>("4 x f16 + 2 x (fp8 x fp8)", w, 24 * 128);
    run_mix<7>("8 x cvt_pk32_fp6_f16 alone", w, 0);
  }
  return 0;
}
