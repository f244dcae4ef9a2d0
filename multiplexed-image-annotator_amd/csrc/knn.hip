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

// ---- neighbourhood compositions for the tissue-region step (reference spatial_methods.tissue_region_partition,
// spatial_methods.py:133-180): for every cell the counts of each cell type among its nearest n_1 < n_2 < ... other cells (the
// reference uses 10 ... 200 of 201 neighbours).  k = 201 does not fit a per-thread register list, so one 256-thread workgroup
// serves one query: candidates pass a shrinking distance threshold into an LDS buffer of 1024 (d, index) pairs; whenever fewer
// than 768 slots are free the buffer is bitonic-sorted, cut to the k best, and the threshold becomes the k-th distance.  The
// final sort leaves the k nearest in (distance, index) order: rank 0 is the cell itself, ranks 1..n_j feed the j-th histogram.
constexpr int KC_BUF = 1024;
constexpr int KC_ROUND = 768;      // candidates examined between two capacity checks (3 per thread)
constexpr int KC_MAXK = 256;
constexpr int KC_LISTS = 8;

__device__ __forceinline__ void kc_sort(double* bd, int* bi, int tid) {
  for (int size = 2; size <= KC_BUF; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = tid; t < KC_BUF / 2; t += 256) {
        const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
        const int hi = lo | stride;
        const bool up = (lo & size) == 0;
        const double dl = bd[lo], dh = bd[hi];
        const int il = bi[lo], ih = bi[hi];
        const bool lo_gt = knn_less(dh, ih, dl, il);
        if (lo_gt == up) { bd[lo] = dh; bd[hi] = dl; bi[lo] = ih; bi[hi] = il; }
      }
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void knn_composition_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                              const int32_t* __restrict__ type, int n, int k, int T, int n_lists,
                                                              const int* __restrict__ list /* n_lists ascending, each < k */,
                                                              uint16_t* __restrict__ counts /* [n][n_lists][T] */) {
  __shared__ double bd[KC_BUF];
  __shared__ int bi[KC_BUF];
  __shared__ int s_cnt;
  __shared__ double s_tau;
  __shared__ unsigned int hist[KC_LISTS * KNN_TYPES];
  const int tid = threadIdx.x;
  const int q = blockIdx.x;
  const double qx = x[q], qy = y[q];
  if (tid == 0) { s_cnt = 0; s_tau = INFINITY; }
  for (int i = tid; i < n_lists * T; i += 256) hist[i] = 0;
  __syncthreads();
  for (int base = 0; base < n; base += KC_ROUND) {
    const double tau = s_tau;
#pragma unroll
    for (int u = 0; u < KC_ROUND / 256; ++u) {
      const int j = base + tid + 256 * u;
      if (j < n) {
        const double dx = x[j] - qx, dy = y[j] - qy;
        const double d = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
        if (d <= tau) {
          const int pos = atomicAdd(&s_cnt, 1);
          if (pos < KC_BUF) {          // cannot trigger (<= 256 kept + 768 new); guards the LDS buffer against a logic error
            bd[pos] = d;
            bi[pos] = j;
          }
        }
      }
    }
    __syncthreads();
    const int cnt = s_cnt < KC_BUF ? s_cnt : KC_BUF;
    __syncthreads();               // every thread has read the count before anyone appends again (the branch below must be uniform)
    if (cnt > KC_BUF - KC_ROUND || base + KC_ROUND >= n) {       // uniform: make room (or finish)
      for (int i = cnt + tid; i < KC_BUF; i += 256) { bd[i] = INFINITY; bi[i] = 0x7FFFFFFF; }
      kc_sort(bd, bi, tid);
      if (tid == 0) {
        s_cnt = cnt < k ? cnt : k;
        s_tau = cnt >= k ? bd[k - 1] : INFINITY;
      }
      __syncthreads();
    }
  }
  // bd / bi[0 .. k-1] sorted; rank 0 is the query itself (distance 0, or the lower index among exact duplicates)
  for (int r = tid + 1; r < k; r += 256) {
    const int nb = bi[r];
    if (nb < 0 || nb >= n) continue;
    const int ty = type[nb];
    if (ty < 0 || ty >= T) continue;
    for (int l = 0; l < n_lists; ++l)
      if (r <= list[l]) atomicAdd(&hist[l * T + ty], 1u);
  }
  __syncthreads();
  for (int i = tid; i < n_lists * T; i += 256) counts[(size_t)q * n_lists * T + i] = (uint16_t)hist[i];
}

int launch_knn_compositions(const double* x, const double* y, const int32_t* type, int n, int k, int T, int n_lists, const int* list_dev,
                            uint16_t* counts, hipStream_t s) {
  if (k < 2 || k > KC_MAXK || k > n || T < 1 || T > KNN_TYPES || n_lists < 1 || n_lists > KC_LISTS) return 1;
  hipLaunchKernelGGL(knn_composition_kernel, dim3(n), dim3(256), 0, s, x, y, type, n, k, T, n_lists, list_dev, counts);
  return 0;
}

int launch_knn_cooccurrence(const double* x, const double* y, const int32_t* type, int n, int k, int T, unsigned long long* matrix,
                            hipStream_t s) {
  if (k < 1 || k > KNN_MAX || T < 1 || T > KNN_TYPES || k > n) return 1;
  hipLaunchKernelGGL(knn_cooccurrence_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, y, type, n, k, T, matrix);
  return 0;
}

}  // namespace ribca
