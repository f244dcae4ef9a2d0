// What can one CU request from an L2-resident buffer, and do requests for the SAME lines by several waves of a workgroup cost as much
// as requests for different lines?  (the GEMM tile forms are ranked by "bytes a CU must fetch per unit of
// matrix work"; the 2 x 2 wave layout of gemm_duo.hip requests every W fragment twice.)
//   hipcc --offload-arch=gfx950 -O2 tools/l2_fetch_probe.hip -o tools/l2_fetch_probe && ./tools/l2_fetch_probe
// One 256-thread workgroup per CU (grid = CUs, or 2 x CUs with "2" as argv[1]); every wave streams 1 KB per instruction
// (global_load_dwordx4, 64 lanes x 16 B, fully coalesced) over a private window of a buffer of argv[2] MB (default 16: 2 MB per XCD,
// L2-resident after the first pass; 128: Infinity-Cache resident) that the grid keeps re-reading, 8 loads in flight per wave.
//   share = 1: the 4 waves of a workgroup read 4 different 1 KB blocks per step
//   share = 2: waves {0,1} and {2,3} read the same block (what the 2 x 2 layout does with W fragments)
//   share = 4: all 4 waves read the same block
// Reported: requested bytes per second per CU (what the waves asked for) and unique bytes per second per CU (what had to come from L2).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHARE>
__global__ __launch_bounds__(256) void fetch_kernel(const char* __restrict__ buf, size_t window, int steps, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int stream = wave / SHARE;                    // waves with the same stream id read the same blocks
  constexpr int NS = 4 / SHARE;                       // distinct streams per workgroup
  const char* base = buf + (size_t)blockIdx.x * window;
  u32x4 acc = {0u, 0u, 0u, 0u};
  const size_t blocks = window / 1024;
  size_t b = stream;
  for (int s = 0; s < steps; s += 8) {
    u32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const char* p = base + (b % blocks) * 1024 + lane * 16;
      v[i] = *reinterpret_cast<const u32x4*>(p);
      b += NS;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= v[i];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;      // keep the loads alive
}

template <int SHARE>
static int run(const char* buf, size_t window, int grid, int steps, unsigned* sink, int n_cu) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(fetch_kernel<SHARE>, dim3(grid), dim3(256), 0, 0, buf, window, steps, sink);      // warm the caches
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(fetch_kernel<SHARE>, dim3(grid), dim3(256), 0, 0, buf, window, steps, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double req = (double)grid * 4.0 * steps * 1024.0, uniq = req / SHARE;
  printf("share %d: %8.3f ms  requested %7.1f GB/s per CU (%6.2f TB/s chip)  unique %7.1f GB/s per CU\n", SHARE, best, req / best / 1e6 / n_cu,
         req / best / 1e9, uniq / best / 1e6 / n_cu);
  return 0;
}

int main(int argc, char** argv) {
  int dev = 0, n_cu = 0;
  CHECK(hipGetDevice(&dev));
  CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  const int per_cu = argc > 1 ? atoi(argv[1]) : 1;
  const int grid = n_cu * per_cu;
  const size_t total_mb = argc > 2 ? (size_t)atoi(argv[2]) : 16;
  const size_t window = (total_mb << 20) / (size_t)grid / 1024 * 1024;      // the grid's windows tile the buffer
  char* buf = nullptr;
  unsigned* sink = nullptr;
  CHECK(hipMalloc(&buf, (size_t)grid * window));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 1, (size_t)grid * window));
  const int steps = 16384;      // 16 MB requested per wave
  printf("%d CUs, %d workgroup(s) of 4 waves per CU, %zu KB window per workgroup, %d x 1 KB loads per wave\n", n_cu, per_cu, window / 1024, steps);
  if (run<1>(buf, window, grid, steps, sink, n_cu)) return 1;
  if (run<2>(buf, window, grid, steps, sink, n_cu)) return 1;
  if (run<4>(buf, window, grid, steps, sink, n_cu)) return 1;
  return 0;
}
