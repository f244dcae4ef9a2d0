// k nearest centroids per cell and the cell-type co-occurrence count built from them
// (reference spatial_methods.neighborhood_analysis, cell_type_annotation/spatial_methods.py:13-130, which asks scikit-learn's
// ball tree once per cell from Python).  Brute force in fp64, exactly the squared distances the ball tree compares
// (d = dx*dx; d += dy*dy, no contraction), ties broken towards the lower index:
//   one thread per query cell, candidates streamed through LDS in tiles, the 32 best kept sorted in registers (a new candidate
//   enters at the bottom and bubbles up through statically indexed compare-exchanges), then neighbours 1..k-1 (0 is the cell
//   itself) are counted into a per-workgroup LDS matrix and flushed with one atomic per non-zero entry.
// n = 1e5 cells -> 1e10 distance evaluations, tens of milliseconds; the reference needs minutes.
#include <algorithm>

#include "ribca_common.h"
#include "ribca_kernels.h"

namespace ribca {

constexpr int KNN_MAX = 32;
constexpr int KNN_TILE = 1024;
constexpr int KNN_TYPES = 32;

__device__ __forceinline__ bool knn_less(double da, int ia, double db, int ib) { return da < db || (da == db && ia < ib); }

__global__ __launch_bounds__(256) void knn_cooccurrence_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                               const int32_t* __restrict__ type, int n, int k, int T,
                                                               unsigned long long* __restrict__ matrix) {
  __shared__ double sx[KNN_TILE], sy[KNN_TILE];
  __shared__ unsigned int hist[KNN_TYPES * KNN_TYPES];
  const int tid = threadIdx.x;
  const int q = blockIdx.x * blockDim.x + tid;
  for (int i = tid; i < T * T; i += blockDim.x) hist[i] = 0;
  const double qx = q < n ? x[q] : 0.0, qy = q < n ? y[q] : 0.0;
  double bd[KNN_MAX];
  int bi[KNN_MAX];
#pragma unroll
  for (int p = 0; p < KNN_MAX; ++p) { bd[p] = INFINITY; bi[p] = 0x7FFFFFFF; }
  for (int base = 0; base < n; base += KNN_TILE) {
    __syncthreads();
    for (int i = tid; i < KNN_TILE; i += blockDim.x) {
      const int j = base + i;
      sx[i] = j < n ? x[j] : 0.0;
      sy[i] = j < n ? y[j] : 0.0;
    }
    __syncthreads();
    const int lim = n - base < KNN_TILE ? n - base : KNN_TILE;
    if (q < n) {
      for (int i = 0; i < lim; ++i) {
        const double dx = sx[i] - qx, dy = sy[i] - qy;
        const double d = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
        const int j = base + i;
        if (knn_less(d, j, bd[KNN_MAX - 1], bi[KNN_MAX - 1])) {
          bd[KNN_MAX - 1] = d;
          bi[KNN_MAX - 1] = j;
#pragma unroll
          for (int p = KNN_MAX - 1; p >= 1; --p) {
            if (knn_less(bd[p], bi[p], bd[p - 1], bi[p - 1])) {
              const double td = bd[p]; bd[p] = bd[p - 1]; bd[p - 1] = td;
              const int ti = bi[p]; bi[p] = bi[p - 1]; bi[p - 1] = ti;
            }
          }
        }
      }
    }
  }
  if (q < n) {
    const int tq = type[q];
#pragma unroll
    for (int p = 1; p < KNN_MAX; ++p)
      if (p < k) atomicAdd(&hist[tq * T + type[bi[p]]], 1u);
  }
  __syncthreads();
  for (int i = tid; i < T * T; i += blockDim.x)
    if (hist[i]) atomicAdd(&matrix[i], (unsigned long long)hist[i]);
}

int launch_knn_cooccurrence(const double* x, const double* y, const int32_t* type, int n, int k, int T, unsigned long long* matrix,
                            hipStream_t s) {
  if (k < 1 || k > KNN_MAX || T < 1 || T > KNN_TYPES || k > n) return 1;
  hipLaunchKernelGGL(knn_cooccurrence_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, y, type, n, k, T, matrix);
  return 0;
}

}  // namespace ribca
