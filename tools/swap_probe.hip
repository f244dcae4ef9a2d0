#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float g4_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  float c = a + b, d;
  asm volatile("v_mov_b32 %0, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %1, %0\n\ts_nop 1" : "=&v"(d), "+v"(c));
  return c + d;
}
__global__ void k(float* o, unsigned* p) {
  const int l = threadIdx.x;
  o[l] = g4_sum((float)(1 << (l >> 4)) * 1000.f + (float)(l & 15));
  unsigned a = l, b = 100 + l;
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  p[l] = r[0]; p[64 + l] = r[1];
  const auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  p[128 + l] = q[0]; p[192 + l] = q[1];
}
int main() {
  float* o; unsigned* p;
  hipMalloc(&o, 256); hipMalloc(&p, 1024);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, p);
  float h[64]; unsigned hp[256];
  hipMemcpy(h, o, 256, hipMemcpyDeviceToHost); hipMemcpy(hp, p, 1024, hipMemcpyDeviceToHost);
  printf("g4_sum (expect 15000 + 4*r16 in every lane):\n");
  for (int i = 0; i < 64; ++i) printf("%g%c", h[i], (i & 15) == 15 ? '\n' : ' ');
  const char* nm[4] = {"swap16 r0", "swap16 r1", "swap32 r0", "swap32 r1"};
  for (int t = 0; t < 4; ++t) { printf("%s:", nm[t]); for (int i = 0; i < 64; i += 8) printf(" %u", hp[64 * t + i]); printf("\n"); }
  return 0;
}
