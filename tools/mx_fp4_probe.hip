// Round 5: what would an fp4 (e2m1) image of A lo cost / need?  (a <= 1.5-unit form of the MX mix.)
//   (1) lane map: which (lane, element) of an fp4 second operand of v_mfma_scale_f32_16x16x128_f8f6f4 meets which element of an fp6 first
//       operand, and whose scale byte multiplies it (as tools/mx_kmap_probe.hip did for fp8 against fp6);
//   (2) v_cvt_scalef32_pk_fp4_f32: rounding, saturation, which byte the selector writes, nibble order;
//   (3) issue cycles of the scaled instruction by operand format (fp6 x fp8, fp6 x fp6, fp6 x fp4), one wave per SIMD, independent accumulators.
//   hipcc -O3 --offload-arch=gfx950 tools/mx_fp4_probe.hip -o /tmp/mx_fp4_probe && /tmp/mx_fp4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int FA, int FB>
__global__ void pair_kernel(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* c) {
  const int ia = blockIdx.x >> 3, ib = blockIdx.x & 7;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[ia * 64 + threadIdx.x], b[ib * 64 + threadIdx.x], acc, FA, FB, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
  c[blockIdx.x * 64 + threadIdx.x] = acc;
}

// one-hot operand set `it`: row r (lane r + 16 g for every g) has its single 1.0 at position p = 16 it + r = 32 g + f
static void one_hot(int fmt, std::vector<i32x8>& regs) {
  regs.assign(8 * 64, i32x8{0, 0, 0, 0, 0, 0, 0, 0});
  for (int it = 0; it < 8; ++it)
    for (int r = 0; r < 16; ++r) {
      const int p = 16 * it + r, g = p >> 5, f = p & 31;
      unsigned w[8] = {0};
      if (fmt == 0) w[f >> 2] = 0x38u << (8 * (f & 3));                       // e4m3 1.0 = 0x38
      else if (fmt == 4) w[f >> 3] = 0x2u << (4 * (f & 7));                   // e2m1 1.0 = 0b0010, element f in nibble f
      else { const int bit = 6 * f; const unsigned code = 0x08;               // e2m3 1.0 = 0b001000
        w[bit >> 5] |= code << (bit & 31); if ((bit & 31) > 26) w[(bit >> 5) + 1] |= code >> (32 - (bit & 31)); }
      for (int e = 0; e < 8; ++e) regs[it * 64 + g * 16 + r][e] = (int)w[e];
    }
}

template <int FA, int FB>
static void run(const char* name) {
  std::vector<i32x8> ha, hb;
  one_hot(FA, ha); one_hot(FB, hb);
  std::vector<int> sa(64), sb(64);
  for (int l = 0; l < 64; ++l) { sa[l] = 127 + (l >> 4); sb[l] = 127 + 4 * (l >> 4); }
  i32x8 *da, *db; int *dsa, *dsb; f32x4* dc;
  CK(hipMalloc(&da, ha.size() * sizeof(i32x8))); CK(hipMalloc(&db, hb.size() * sizeof(i32x8)));
  CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dc, 64 * 64 * sizeof(f32x4)));
  CK(hipMemcpy(da, ha.data(), ha.size() * sizeof(i32x8), hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), hb.size() * sizeof(i32x8), hipMemcpyHostToDevice));
  CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL((pair_kernel<FA, FB>), dim3(64), dim3(64), 0, 0, da, db, dsa, dsb, dc);
  CK(hipDeviceSynchronize());
  std::vector<f32x4> hc(64 * 64);
  CK(hipMemcpy(hc.data(), dc, hc.size() * sizeof(f32x4), hipMemcpyDeviceToHost));
  printf("%s: first-operand position p = 32 g + f  ->  second-operand position q it meets (value = product of the two scale factors)\n", name);
  int identity = 1, scales_ok = 1;
  for (int p = 0; p < 128; ++p) {
    int found = -1, count = 0; float val = 0;
    for (int q = 0; q < 128; ++q) {
      const int ia = p >> 4, ra = p & 15, ib = q >> 4, rb = q & 15;
      const float v = hc[(ia * 8 + ib) * 64 + rb + 16 * (ra >> 2)][ra & 3];
      if (v != 0.f) { found = q; val = v; ++count; }
    }
    if (found != p || count != 1) identity = 0;
    const int g = p >> 5;
    if (val != (float)(1 << g) * (float)(1 << (4 * g))) scales_ok = 0;
    if (p % 8 == 0) printf("\n  ");
    printf("%3d->%3d(x%g,n%d) ", p, found, val, count);
  }
  printf("\n  => %s; %s\n", identity ? "IDENTITY: the two formats share one k map" : "NOT the identity",
         scales_ok ? "each element takes the scale bytes of ITS lane group on both sides" : "scale bytes do NOT follow the lane group");
  CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dsa)); CK(hipFree(dsb)); CK(hipFree(dc));
}

// ---- (2) the conversion
__global__ void cvt_kernel(const float* x, float scale, unsigned* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // four calls, selectors 0..3, values (x, -x) / (2x, x/2) / (x, 0) / (0, x): which byte, which nibble
  unsigned r = 0xdeadbeefu;
  r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, x[i], -x[i], scale, 0);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, 2.f * x[i], 0.5f * x[i], scale, 1);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, x[i], 0.f, scale, 2);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, 0.f, x[i], scale, 3);
  out[i] = r;
}

// ---- (3) issue cycles
template <int FA, int FB>
__global__ void time_kernel(const i32x8* a, const i32x8* b, f32x4* c, unsigned long long* cyc, int iters) {
  const i32x8 va = a[threadIdx.x & 63], vb = b[threadIdx.x & 63];
  f32x4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, acc[j], FA, FB, 0, 127, 0, 127);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = acc[0];
#pragma unroll
  for (int j = 1; j < 8; ++j) s += acc[j];
  c[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int FA, int FB>
static void time_it(const char* name) {
  i32x8 *da, *db; f32x4* dc; unsigned long long* dcy;
  CK(hipMalloc(&da, 64 * sizeof(i32x8))); CK(hipMalloc(&db, 64 * sizeof(i32x8))); CK(hipMalloc(&dc, 256 * 512 * sizeof(f32x4))); CK(hipMalloc(&dcy, 8));      // 256 blocks x up to 512 threads
  std::vector<i32x8> h(64);
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) h[l][e] = (int)(0x12492492u * (unsigned)(l + 3 * e + 1));      // arbitrary finite codes
  CK(hipMemcpy(da, h.data(), 64 * sizeof(i32x8), hipMemcpyHostToDevice)); CK(hipMemcpy(db, h.data(), 64 * sizeof(i32x8), hipMemcpyHostToDevice));
  const int iters = 2000;
  for (int waves = 1; waves <= 2; ++waves) {      // 1 or 2 waves per SIMD (256 / 512 threads per CU-sized block), every CU busy
    hipLaunchKernelGGL((time_kernel<FA, FB>), dim3(256), dim3(256 * waves), 0, 0, da, db, dc, dcy, iters);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((time_kernel<FA, FB>), dim3(256), dim3(256 * waves), 0, 0, da, db, dc, dcy, iters);
    CK(hipDeviceSynchronize());
    unsigned long long cy = 0;
    CK(hipMemcpy(&cy, dcy, 8, hipMemcpyDeviceToHost));
    printf("  %-10s %d wave(s) per SIMD: %.1f shader cycles per instruction and wave\n", name, waves, (double)cy / (iters * 8.0));
  }
  CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dc)); CK(hipFree(dcy));
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  run<2, 4>("fp6 x fp4");
  run<4, 2>("fp4 x fp6");
  run<4, 4>("fp4 x fp4");
  {
    const float xs[] = {0.f, 0.2f, 0.25f, 0.26f, 0.5f, 0.74f, 0.75f, 0.76f, 1.f, 1.24f, 1.25f, 1.26f, 1.5f, 1.75f, 2.f, 2.5f, 3.f, 3.5f, 4.f, 5.f, 6.f, 7.f, 100.f, -0.3f};
    const int n = sizeof(xs) / sizeof(float);
    float* dx; unsigned* dout;
    CK(hipMalloc(&dx, sizeof(xs))); CK(hipMalloc(&dout, n * 4));
    CK(hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice));
    for (float scale : {1.0f, 4.0f}) {
      hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, dx, scale, dout, n);
      CK(hipDeviceSynchronize());
      std::vector<unsigned> r(n);
      CK(hipMemcpy(r.data(), dout, n * 4, hipMemcpyDeviceToHost));
      printf("v_cvt_scalef32_pk_fp4_f32, scale %g: x -> word (sel 0: (x, -x) | sel 1: (2x, x/2) | sel 2: (x, 0) | sel 3: (0, x); e2m1 codes 0 .5 1 1.5 2 3 4 6 = 0..7, sign bit 8)\n", scale);
      for (int i = 0; i < n; ++i) printf("  x = %7.3f -> 0x%08x\n", xs[i], r[i]);
    }
    CK(hipFree(dx)); CK(hipFree(dout));
  }
  printf("issue cycles of v_mfma_scale_f32_16x16x128_f8f6f4 by operand formats (8 independent accumulators per wave):\n");
  time_it<2, 0>("fp6 x fp8");
  time_it<2, 2>("fp6 x fp6");
  time_it<2, 4>("fp6 x fp4");
  time_it<4, 4>("fp4 x fp4");
  return 0;
}
