// Which (lane, element) of an fp8 operand of v_mfma_scale_f32_16x16x128_f8f6f4 meets which (lane, element) of an fp6 operand?
// Products of equal formats cannot tell (any k permutation applied to both operands cancels); the MX GEMM mixes formats (A lo fp8 x W hi fp6),
// so the RELATIVE k maps of the formats matter.  One-hot rows against one-hot columns: position p = 32 g + f (lane group g, element f) of the
// first operand, q likewise of the second; C[n][m] != 0 iff p meets q.  Also: whose scale byte multiplies an element.
//   hipcc -O3 --offload-arch=gfx950 tools/mx_kmap_probe.hip -o /tmp/mx_kmap_probe && /tmp/mx_kmap_probe
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
  // 64 (operand sets) x 64 lanes: set = 8 * ia + ib; a indexed by ia, b by ib
  const int ia = blockIdx.x >> 3, ib = blockIdx.x & 7;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[ia * 64 + threadIdx.x], b[ib * 64 + threadIdx.x], acc, FA, FB, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
  c[blockIdx.x * 64 + threadIdx.x] = acc;
}

// one-hot operand set `it`: row r (lane r + 16 g for every g) has its single 1.0 at position p = 16 it + r
static void one_hot(int fmt, std::vector<i32x8>& regs) {
  regs.assign(8 * 64, i32x8{0, 0, 0, 0, 0, 0, 0, 0});
  for (int it = 0; it < 8; ++it)
    for (int r = 0; r < 16; ++r) {
      const int p = 16 * it + r, g = p >> 5, f = p & 31;
      unsigned w[8] = {0};
      if (fmt == 0) w[f >> 2] = 0x38u << (8 * (f & 3));                       // e4m3 1.0 = 0x38
      else { const int bit = 6 * f; const unsigned code = 0x08;               // e2m3 1.0 = 0b001000
        w[bit >> 5] |= code << (bit & 31); if ((bit & 31) > 26) w[(bit >> 5) + 1] |= code >> (32 - (bit & 31)); }
      for (int e = 0; e < 8; ++e) regs[it * 64 + g * 16 + r][e] = (int)w[e];
    }
}

template <int FA, int FB>
static void run(const char* name) {
  std::vector<i32x8> ha, hb;
  one_hot(FA, ha); one_hot(FB, hb);
  // scales: first operand lane group g -> 2^g, second operand -> 2^(4 g') : the product's exponent tells whose scale bytes were applied
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
  int identity = 1;
  for (int p = 0; p < 128; ++p) {
    int found = -1, count = 0; float val = 0;
    for (int q = 0; q < 128; ++q) {
      const int ia = p >> 4, ra = p & 15, ib = q >> 4, rb = q & 15;
      // C row index = first operand's row (ra), column = second operand's row (rb): lane = rb + 16 * (ra / 4), reg = ra % 4
      const float v = hc[(ia * 8 + ib) * 64 + rb + 16 * (ra >> 2)][ra & 3];
      if (v != 0.f) { found = q; val = v; ++count; }
    }
    if (found != p) identity = 0;
    if (p % 8 == 0) printf("\n  ");
    printf("%3d->%3d(x%g,n%d) ", p, found, val, count);
  }
  printf("\n  => %s\n", identity ? "IDENTITY: the two formats share one k map" : "NOT the identity");
  CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dsa)); CK(hipFree(dsb)); CK(hipFree(dc));
}

int main() {
  run<2, 2>("fp6 x fp6");
  run<0, 0>("fp8 x fp8");
  run<2, 0>("fp6 x fp8");
  run<0, 2>("fp8 x fp6");
  return 0;
}
