// What the matrix cores SUSTAIN on this chip, whole GPU, tens of milliseconds (the power-managed clock, not the 2.4 GHz label):
// register-resident MFMA loops -- no memory traffic at all -- for the instruction the GEMMs use (v_mfma_f32_16x16x32_f16) and the
// 32x32x16 shape (half the operand reads per FLOP), with 1, 2 and 4 waves per SIMD.  For each: wall time (HIP events), achieved
// dense TFLOP/s, and the in-kernel shader clock (s_memtime cycles per s_memrealtime tick of 10 ns).
// The roofline in bench.py prices against 2.5 PFLOP/s (MI355X_MICROARCH.md); this probe says how much of that any kernel can see
// while the chip holds its power budget, which is the number the K-loop duty figures of DESIGN.md section 6 should be read against.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_sustain_probe.hip -o gpurun_out/mfma_sustain_probe && gpurun_out/mfma_sustain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// MODE 0: 16x16x32 f16, 8 independent accumulator chains; MODE 1: 32x32x16 f16, 4 chains (same accumulator registers: 32 vs 64)
template <int MODE>
__global__ __launch_bounds__(1024) void sustain_kernel(unsigned long long* stamps, float* sink, int iters) {
  // four operand sets of hash-random fp16 bit patterns in [-2, 2): successive MFMAs see different operands, so the multiplier
  // array toggles as it does on real data (constant operands would draw less power and flatter the clock)
  h8 av[4], bv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      unsigned x = (threadIdx.x * 8u + i) * 2654435761u + s * 40503u + blockIdx.x * 97u;
      x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
      av[s][i] = (_Float16)(((int)(x & 0xffffu) - 32768) * (1.0f / 16384.0f));
      bv[s][i] = (_Float16)(((int)(x >> 16) - 32768) * (1.0f / 16384.0f));
    }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  if (MODE == 0) {
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[(j + u) & 3], bv[(j >> 1) & 3], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) r += acc[j][0] + acc[j][3];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[(j + u) & 3], bv[(j + 2 * u) & 3], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) r += acc[j][0] + acc[j][15];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = c1 - c0;
  }
  if (r == 12345.678f) sink[0] = r;      // keeps the chains alive
}

template <int MODE>
static void run(const char* name, int waves_per_simd, int iters) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int threads = 256 * waves_per_simd;        // one workgroup per CU: 4 SIMDs x waves_per_simd waves
  const int waves = cus * threads / 64;
  unsigned long long* stamps;
  float* sink;
  CK(hipMalloc(&stamps, sizeof(unsigned long long) * 2 * waves));
  CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  sustain_kernel<MODE><<<cus, threads>>>(stamps, sink, iters / 8);      // warm-up
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  sustain_kernel<MODE><<<cus, threads>>>(stamps, sink, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long* h = (unsigned long long*)malloc(sizeof(unsigned long long) * 2 * waves);
  CK(hipMemcpy(h, stamps, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost));
  double ticks = 0, cyc = 0;
  for (int w = 0; w < waves; ++w) { ticks += (double)h[2 * w]; cyc += (double)h[2 * w + 1]; }
  const double ghz = cyc / ticks / 10.0;            // shader cycles per 10 ns tick
  // FLOP per MFMA: 2 * 16*16*32 = 16384 (MODE 0) and 2 * 32*32*16 = 32768 (MODE 1); per iteration 32 (MODE 0) / 16 (MODE 1) of them
  const double flop = (double)waves * (double)iters * (MODE == 0 ? 32.0 * 16384.0 : 16.0 * 32768.0);
  const double tf = flop / (ms * 1e-3) / 1e12;
  // cycles per MFMA per SIMD from the wall time (the SIMD's waves share its matrix core, and the issue arbiter serves the oldest wave
  // first: per-wave stamps of co-resident waves end at 1/w, 2/w, ... of the kernel, so they cannot be averaged into a rate)
  const double per_mfma = (ms * 1e-3 * ghz * 1e9) / ((double)iters * (MODE == 0 ? 32.0 : 16.0) * waves_per_simd);
  printf("%-22s %d wave(s)/SIMD: %7.2f ms  %7.1f TFLOP/s dense  clock %.2f GHz  %.1f cycles per MFMA per SIMD  (%.0f%% of 2.5 PF)\n", name,
         waves_per_simd, ms, tf, ghz, per_mfma, tf / 2500.0 * 100.0);
  free(h);
  CK(hipFree(stamps));
  CK(hipFree(sink));
}

int main() {
  const int iters = 60000;      // x 32 (16) MFMAs of 16 (32) passes... tens of milliseconds per launch
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("mfma_f32_16x16x32_f16", 1, iters);
    run<0>("mfma_f32_16x16x32_f16", 2, iters / 2);
    run<0>("mfma_f32_16x16x32_f16", 4, iters / 4);
    run<1>("mfma_f32_32x32x16_f16", 1, iters);
    run<1>("mfma_f32_32x32x16_f16", 2, iters / 2);
    run<1>("mfma_f32_32x32x16_f16", 4, iters / 4);
  }
  return 0;
}
