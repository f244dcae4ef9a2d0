// Does v_mfma_f32_16x16x32_f16 keep fp16 SUBNORMAL inputs (the low part l = x - fp16(x) of small activations is subnormal)?
// And is v_pk_mul_f16 by 2^-12 exact into the subnormal range (default HIP denorm mode)?   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const uint16_t* abits, const uint16_t* bbits, float* out, const uint16_t* mul_in, uint16_t* mul_out) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = __builtin_bit_cast(_Float16, abits[threadIdx.x * 8 + i]); b[i] = __builtin_bit_cast(_Float16, bbits[threadIdx.x * 8 + i]); }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = acc[i];
  h2 v = {__builtin_bit_cast(_Float16, mul_in[2 * threadIdx.x]), __builtin_bit_cast(_Float16, mul_in[2 * threadIdx.x + 1])};
  const h2 s = {(_Float16)0.000244140625f, (_Float16)0.000244140625f};   // 2^-12
  v = v * s;
  mul_out[2 * threadIdx.x] = __builtin_bit_cast(uint16_t, v[0]);
  mul_out[2 * threadIdx.x + 1] = __builtin_bit_cast(uint16_t, v[1]);
}
static float h2f(uint16_t h) { int s = h >> 15, e = (h >> 10) & 31, m = h & 1023; float v = e == 0 ? ldexpf((float)m, -24) : (e == 31 ? INFINITY : ldexpf((float)(m + 1024), e - 25)); return s ? -v : v; }
int main() {
  uint16_t ha[512], hb[512], hm[128], hmo[128];
  // A[row][k]: subnormal pattern: bits = 1 + ((row * 32 + k) % 1023)  (values m * 2^-24);  B[k][col] = 1024.0 (0x6400) * (1 + col % 3)
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 8; ++i) {
    const int r = l & 15, kk = 8 * (l >> 4) + i;
    ha[l * 8 + i] = (uint16_t)(1 + ((r * 32 + kk) * 7) % 1023);
    const float bv = 1024.0f * (1 + (r % 3));
    hb[l * 8 + i] = bv == 1024.0f ? 0x6400 : (bv == 2048.0f ? 0x6800 : 0x6A00);
  }
  for (int i = 0; i < 128; ++i) hm[i] = (uint16_t)(0x3C00 + i * 97 % 8192 - 0x2000 + (i & 1 ? 0x8000 : 0));   // values around 2^-8 .. 2^0
  uint16_t *da, *db, *dm, *dmo; float* dout;
  hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dout, 1024); hipMalloc(&dm, 256); hipMalloc(&dmo, 256);
  hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice); hipMemcpy(dm, hm, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout, dm, dmo);
  float ho[256]; hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost); hipMemcpy(hmo, dmo, 256, hipMemcpyDeviceToHost);
  double maxrel = 0; int zeros = 0;
  for (int l = 0; l < 64; ++l) for (int reg = 0; reg < 4; ++reg) {
    const int n = l & 15, m = 4 * (l >> 4) + reg;      // transposed roles do not matter for the check: compute C[m][n] = sum_k A[m][k] B[k][n]
    double ref = 0;
    for (int kk = 0; kk < 32; ++kk) ref += (double)h2f((uint16_t)(1 + ((m * 32 + kk) * 7) % 1023)) * (1024.0 * (1 + (n % 3)));
    const double got = ho[l * 4 + reg];
    if (got == 0) ++zeros;
    maxrel = fmax(maxrel, fabs(got - ref) / ref);
  }
  printf("f16 MFMA with subnormal A inputs: max rel err %.3e, zero outputs %d of 256  (flushed inputs would give all zeros)\n", maxrel, zeros);
  int bad = 0;
  for (int i = 0; i < 128; ++i) {
    const float want = h2f(hm[i]) * 0.000244140625f;   // exact in fp32; round to f16 (subnormal step 2^-24) RNE
    const float q = ldexpf(rintf(ldexpf(want, 24)), -24);
    const float got = h2f(hmo[i]);
    const bool sub = fabsf(want) < 6.103515625e-05f;
    if (sub ? got != q : fabsf(got - want) > fabsf(want) * 0.001f) { if (bad < 5) printf("  mul: in %g want %g got %g\n", h2f(hm[i]), sub ? q : want, got); ++bad; }
  }
  printf("v_pk_mul_f16 by 2^-12 into the subnormal range: %d mismatches of 128\n", bad);
  return 0;
}
