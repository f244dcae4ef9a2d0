// Micro-benchmark behind DESIGN.md's epilogue analysis: how fast can N workgroups (one per CU) drain a GEMM-tile-like
// store stream, as a function of the per-instruction address shape?   hipcc -O3 --offload-arch=gfx950 store_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// each workgroup = 8 waves; each wave writes `reps` blocks of 16 rows x 256 B (a 16 x 64 fp32 accumulator slice)
// shape 0: lane -> 16 B, instruction = 1 KB contiguous (4 rows x 256 B when rows are contiguous)
// shape 1: instruction = 16 rows x 64 B, row stride `ld` bytes            (fp32 residual / paired PS store)
// shape 2: two instructions of 16 rows x 2 x 16 B pieces (8 B per lane)    (unpaired PS store)
// shape 3: instruction = 4 rows x 256 B, row stride `ld` bytes            (LDS-transposed epilogue)
template <int SHAPE>
__global__ __launch_bounds__(512) void store_kernel(char* out, size_t wg_stride, int ld, int reps, int rmw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* base = out + (size_t)blockIdx.x * wg_stride;
  const int r16 = lane & 15, g = lane >> 4;
  for (int rep = 0; rep < reps; ++rep) {
    // this wave's 16-row block: rows (rep * 8 + wave) * 16 ..
    char* blk = base + (size_t)(rep * 8 + wave) * 16 * ld;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (SHAPE == 0) {
        uint4 v = {(unsigned)lane, (unsigned)rep, (unsigned)j, 1u};
        char* p = blk + (size_t)j * 1024 + lane * 16;   // treated as contiguous
        if (rmw) { uint4 o = *reinterpret_cast<uint4*>(p); v.x += o.x; v.y += o.y; }
        *reinterpret_cast<uint4*>(p) = v;
      } else if (SHAPE == 1) {
        uint4 v = {(unsigned)lane, (unsigned)rep, (unsigned)j, 1u};
        char* p = blk + (size_t)r16 * ld + j * 64 + g * 16;
        if (rmw) { uint4 o = *reinterpret_cast<uint4*>(p); v.x += o.x; v.y += o.y; }
        *reinterpret_cast<uint4*>(p) = v;
      } else if (SHAPE == 2) {
        uint2 v = {(unsigned)lane, (unsigned)rep};
        char* p = blk + (size_t)r16 * ld + j * 64 + (g >> 1) * 32 + (g & 1) * 8;
        *reinterpret_cast<uint2*>(p) = v;
        *reinterpret_cast<uint2*>(p + 16) = v;
      } else {
        uint4 v = {(unsigned)lane, (unsigned)rep, (unsigned)j, 1u};
        char* p = blk + (size_t)(4 * j + (lane >> 4)) * ld + (lane & 15) * 16;
        if (rmw) { uint4 o = *reinterpret_cast<uint4*>(p); v.x += o.x; v.y += o.y; }
        *reinterpret_cast<uint4*>(p) = v;
      }
    }
  }
}

int main(int argc, char** argv) {
  const size_t total = (size_t)3 << 30;
  char* buf;
  if (hipMalloc(&buf, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 0, total);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int ld = 2304 * 4;                 // fp32 row of D = 2304? no: output row pitch of the fc1 activation (bytes)
  const int grids[] = {1, 8, 32, 64, 128, 256, 512, 1024};
  printf("%6s %6s | %10s %10s %10s %10s | %10s %10s\n", "WGs", "MB", "contig", "16x64B", "16x2x16B", "4x256B", "rmw16x64", "rmw4x256");
  for (int gi = 0; gi < 8; ++gi) {
    const int wgs = grids[gi];
    // per workgroup: reps * 8 waves * 16 rows of pitch ld
    int reps = 16;
    size_t wg_stride = (size_t)reps * 8 * 16 * ld;
    while ((size_t)wgs * wg_stride > total) { reps /= 2; wg_stride = (size_t)reps * 8 * 16 * ld; }
    const double bytes = (double)wgs * reps * 8 * 4096.0;
    double gbps[6];
    for (int sh = 0; sh < 6; ++sh) {
      float best = 1e30f;
      for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0, 0);
        const int rmw = sh >= 4;
        switch (sh) {
          case 0: hipLaunchKernelGGL(store_kernel<0>, dim3(wgs), dim3(512), 0, 0, buf, wg_stride, ld, reps, 0); break;
          case 1: hipLaunchKernelGGL(store_kernel<1>, dim3(wgs), dim3(512), 0, 0, buf, wg_stride, ld, reps, 0); break;
          case 2: hipLaunchKernelGGL(store_kernel<2>, dim3(wgs), dim3(512), 0, 0, buf, wg_stride, ld, reps, 0); break;
          case 3: hipLaunchKernelGGL(store_kernel<3>, dim3(wgs), dim3(512), 0, 0, buf, wg_stride, ld, reps, 0); break;
          case 4: hipLaunchKernelGGL(store_kernel<1>, dim3(wgs), dim3(512), 0, 0, buf, wg_stride, ld, reps, 1); break;
          case 5: hipLaunchKernelGGL(store_kernel<3>, dim3(wgs), dim3(512), 0, 0, buf, wg_stride, ld, reps, 1); break;
        }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
      }
      gbps[sh] = bytes / (best * 1e-3) / 1e9;
    }
    printf("%6d %6.0f | %10.0f %10.0f %10.0f %10.0f | %10.0f %10.0f   GB/s (rmw counts written bytes only)\n", wgs, bytes / 1e6, gbps[0], gbps[1], gbps[2],
           gbps[3], gbps[4], gbps[5]);
  }
  return 0;
}
