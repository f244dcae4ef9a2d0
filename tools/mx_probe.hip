// Hardware probe behind DESIGN.md's "fp16 + MX-fp6 correction" numerics (gfx950 only; no vendor ISA text is available in
// the image, so operand layouts and conversion semantics are pinned here by experiment):
//   A. v_mfma_scale_f32_16x16x128_f8f6f4 with e2m3 operands: lane -> (row, k) map, 6-bit field order, E8M0 scale byte
//   B. v_cvt_scalef32_pk32_fp6_f16: rounding, scale direction, field order (must equal the MFMA's)
//   C. issue rates: f16 16x16x32, MX fp6 / fp8 16x16x128, the conversion, and the production mix 4 x f16 + 2 x fp6 per tile
//   hipcc -O3 --offload-arch=gfx950 tools/mx_probe.hip -o gpurun_out/mx_probe && gpurun_out/mx_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(6))) unsigned u32x6;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(32))) _Float16 h32;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static double e2m3_value(int code) {   // s eem mm m
  const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
  const double v = e == 0 ? m / 8.0 : (1.0 + m / 8.0) * (double)(1 << (e - 1));
  return s ? -v : v;
}
static int e2m3_encode_rne(double x) {   // saturating round-to-nearest-even onto the e2m3 grid
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

__global__ void mfma_fp6_kernel(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* c) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 2, 2, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
  c[threadIdx.x] = acc;
}
__global__ void mfma_fp6_opsel_kernel(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* c) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 2, 2, 2, sa[threadIdx.x], 1, sb[threadIdx.x]);
  c[threadIdx.x] = acc;
}
__global__ void cvt_kernel(const h32* in, const float* scale, u32x6* out) {
  out[threadIdx.x] = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(in[threadIdx.x], scale[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------ timing
// one wave per SIMD (256 threads), every CU busy; cycles per instruction from s_memtime around an unrolled loop
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(unsigned long long* out, int iters, const h32* src) {
  h32 hv = src[threadIdx.x & 63];
  h8 ha, hb;
  i32x8 ia, ib;
#pragma unroll
  for (int i = 0; i < 8; ++i) { ha[i] = hv[i]; hb[i] = hv[8 + i]; ia[i] = (int)threadIdx.x * 77 + i; ib[i] = (int)threadIdx.x * 31 + i * 5; }
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x6 cv = {0, 0, 0, 0, 0, 0};
  const int sc = 0x7f7f7f7f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) {            // f16 16x16x32
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
      } else if (MODE == 1) {     // MX fp6 x fp6
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ia, ib, acc[i], 2, 2, 0, sc, 0, sc);
      } else if (MODE == 2) {     // MX fp8 x fp8
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ia, ib, acc[i], 0, 0, 0, sc, 0, sc);
      } else if (MODE == 3) {     // MX fp8 x fp6
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ia, ib, acc[i], 0, 2, 0, sc, 0, sc);
      } else if (MODE == 4) {     // production mix per tile and 128 k: 4 x f16 + 2 x fp6
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, ha, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, ha, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, hb, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ia, ib, acc[i], 2, 2, 0, sc, 0, sc);
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ib, ia, acc[i], 2, 2, 0, sc, 0, sc);
      } else if (MODE == 5) {     // the conversion alone (dependent on itself through one register so it is not hoisted)
        hv[0] = (_Float16)(float)(cv[0] & 3);
        cv = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hv, 1.0f);
      } else if (MODE == 6) {     // bf16 16x16x32 (round-1 path), for the same-binary comparison
        typedef __attribute__((ext_vector_type(8))) __bf16 b8;
        b8 ba, bb;
#pragma unroll
        for (int q = 0; q < 8; ++q) { ba[q] = (__bf16)(float)ha[q]; bb[q] = (__bf16)(float)hb[q]; }
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[i], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  s += (float)cv[0] + (float)cv[5];
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = (unsigned long long)s; }
}

template <int MODE>
static void run_rate(const char* name, int per_iter, const h32* dsrc) {
  unsigned long long* d;
  CK(hipMalloc(&d, 256 * 2 * sizeof(unsigned long long)));
  const int iters = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(256), 0, 0, d, 10, dsrc);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(256), 0, 0, d, iters, dsrc);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(512);
  CK(hipMemcpy(h.data(), d, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double cyc = 0;
  for (int i = 0; i < 256; ++i) cyc += (double)h[2 * i];
  cyc /= 256.0;
  const double n = (double)iters * 16 * per_iter;
  printf("  %-34s %7.2f s_memtime ticks per instruction (wave), %8.3f ms wall, %6.2f ns per instruction\n", name, cyc / n, ms, ms * 1e6 / n);
  CK(hipFree(d));
}

int main() {
  srand(1234);
  // ---------------------------------------------------------------- A: MFMA layout
  std::vector<int> acode(16 * 128), bcode(16 * 128);       // a[i][k], bt[j][k]
  for (auto& v : acode) v = rand() & 63;
  for (auto& v : bcode) v = rand() & 63;
  std::vector<int> sa(64), sb(64);                          // per lane (row, k block) exponent bytes
  for (int l = 0; l < 64; ++l) { sa[l] = 120 + rand() % 15; sb[l] = 120 + rand() % 15; }
  auto pack = [&](const std::vector<int>& code, std::vector<i32x8>& regs) {
    regs.assign(64, i32x8{0, 0, 0, 0, 0, 0, 0, 0});
    for (int l = 0; l < 64; ++l) {
      const int r = l & 15, g = l >> 4;
      unsigned w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int f = 0; f < 32; ++f) {
        const unsigned c = (unsigned)code[r * 128 + 32 * g + f];
        const int bit = 6 * f;
        w[bit >> 5] |= c << (bit & 31);
        if ((bit & 31) > 26) w[(bit >> 5) + 1] |= c >> (32 - (bit & 31));
      }
      for (int i = 0; i < 8; ++i) regs[l][i] = (int)w[i];
    }
  };
  std::vector<i32x8> areg, breg;
  pack(acode, areg); pack(bcode, breg);
  i32x8 *da, *db; int *dsa, *dsb; f32x4* dc;
  CK(hipMalloc(&da, 64 * sizeof(i32x8))); CK(hipMalloc(&db, 64 * sizeof(i32x8)));
  CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dc, 64 * sizeof(f32x4)));
  CK(hipMemcpy(da, areg.data(), 64 * sizeof(i32x8), hipMemcpyHostToDevice));
  CK(hipMemcpy(db, breg.data(), 64 * sizeof(i32x8), hipMemcpyHostToDevice));
  for (int variant = 0; variant < 2; ++variant) {
    // variant 0: exponent in byte 0, opsel 0;  variant 1: A's exponent in byte 2 (opsel 2), B's in byte 1 (opsel 1), junk elsewhere
    std::vector<int> ha(64), hb(64);
    for (int l = 0; l < 64; ++l) {
      ha[l] = variant == 0 ? sa[l] : (int)(0x11u | (0x22u << 8) | ((unsigned)sa[l] << 16) | (0x33u << 24));
      hb[l] = variant == 0 ? sb[l] : (int)(0x44u | ((unsigned)sb[l] << 8) | (0x55u << 16) | (0x66u << 24));
    }
    CK(hipMemcpy(dsa, ha.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsb, hb.data(), 256, hipMemcpyHostToDevice));
    if (variant == 0) hipLaunchKernelGGL(mfma_fp6_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
    else hipLaunchKernelGGL(mfma_fp6_opsel_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
    CK(hipDeviceSynchronize());
    std::vector<f32x4> hc(64);
    CK(hipMemcpy(hc.data(), dc, 64 * sizeof(f32x4), hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int l = 0; l < 64; ++l)
      for (int reg = 0; reg < 4; ++reg) {
        const int n = l & 15, m = 4 * (l >> 4) + reg;
        double ref = 0;
        for (int k = 0; k < 128; ++k) {
          const int g = k >> 5;
          ref += e2m3_value(acode[m * 128 + k]) * ldexp(1.0, sa[g * 16 + m] - 127) * e2m3_value(bcode[n * 128 + k]) * ldexp(1.0, sb[g * 16 + n] - 127);
        }
        maxerr = fmax(maxerr, fabs(ref - (double)hc[l][reg]));
        maxref = fmax(maxref, fabs(ref));
      }
    printf("A%d: fp6 MFMA vs hypothesis (lane r=l&15,g=l>>4 holds k=32g..32g+31, field f at bit 6f, E8M0 byte via opsel): max err %.3e (max |ref| %.3e)\n",
           variant, maxerr, maxref);
  }
  // ---------------------------------------------------------------- B: conversion
  {
    std::vector<_Float16> hin(64 * 32);
    std::vector<float> hs(64);
    for (int l = 0; l < 64; ++l) {
      hs[l] = ldexpf(1.0f, (l % 9) - 4);
      for (int i = 0; i < 32; ++i) {
        float v = ((rand() % 20001) - 10000) / 10000.0f * 9.0f * hs[l];
        if (i == 0) v = 0.0625f * hs[l];             // exact tie between 0 and the first subnormal 0.125
        if (i == 1) v = 0.1875f * hs[l];             // tie between 0.125 and 0.25
        if (i == 2) v = 100.0f * hs[l];              // saturates
        if (i == 3) v = -1.0625f * hs[l];            // tie between 1.0 and 1.125
        hin[l * 32 + i] = (_Float16)v;
      }
    }
    h32* din; float* dscale; u32x6* dout;
    CK(hipMalloc(&din, 64 * sizeof(h32))); CK(hipMalloc(&dscale, 256)); CK(hipMalloc(&dout, 64 * sizeof(u32x6)));
    CK(hipMemcpy(din, hin.data(), 64 * sizeof(h32), hipMemcpyHostToDevice));
    CK(hipMemcpy(dscale, hs.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, din, dscale, dout);
    CK(hipDeviceSynchronize());
    std::vector<u32x6> ho(64);
    CK(hipMemcpy(ho.data(), dout, 64 * sizeof(u32x6), hipMemcpyDeviceToHost));
    int bad_div = 0, bad_mul = 0, shown = 0;
    for (int l = 0; l < 64; ++l)
      for (int i = 0; i < 32; ++i) {
        const int bit = 6 * i;
        unsigned c = ho[l][bit >> 5] >> (bit & 31);
        if ((bit & 31) > 26) c |= ho[l][(bit >> 5) + 1] << (32 - (bit & 31));
        c &= 63;
        const double x = (double)(float)hin[l * 32 + i];
        const int ediv = e2m3_encode_rne(x / hs[l]), emul = e2m3_encode_rne(x * hs[l]);
        const bool zero_ok = (e2m3_value(c) == 0 && e2m3_value(ediv) == 0);
        if ((int)c != ediv && !zero_ok) {
          ++bad_div;
          if (shown < 12 && hs[l] == 1.0f) { printf("   cvt: x=%g scale=%g -> code %u (%g), x/s RNE would be %d (%g)\n", x, hs[l], c, e2m3_value(c), ediv, e2m3_value(ediv)); ++shown; }
        }
        if ((int)c != emul && !(e2m3_value(c) == 0 && e2m3_value(emul) == 0)) ++bad_mul;
      }
    printf("B: v_cvt_scalef32_pk32_fp6_f16 (element i at bit 6i): mismatches vs RNE(x / scale) %d, vs RNE(x * scale) %d of 2048\n", bad_div, bad_mul);
    // ---------------------------------------------------------------- C: rates
    printf("C: issue rates, one wave per SIMD on every CU (s_memtime ticks are 100 MHz-domain? compare ratios and ns):\n");
    run_rate<6>("bf16 16x16x32", 1, din);
    run_rate<0>("f16 16x16x32", 1, din);
    run_rate<1>("MX fp6 x fp6 16x16x128", 1, din);
    run_rate<2>("MX fp8 x fp8 16x16x128", 1, din);
    run_rate<3>("MX fp8 x fp6 16x16x128", 1, din);
    run_rate<4>("mix 4 x f16 + 2 x fp6 (per 6)", 6, din);
    run_rate<5>("cvt_scalef32_pk32_fp6_f16", 1, din);
  }
  return 0;
}
