// Whole-image normalisation primitives (reference cell_type_annotation/preprocess.py:214-239, ImageProcessor._normalize).
//
// Per channel the reference does:  bg = gaussian_filter(x, 20); bg = min(bg, 125); x = max(x - bg, 0);
// [x = gaussian_filter(x, blur)]; no positive pixel -> -1; t = percentile(x, amax); t > 20 -> x = min(x, t);
// x = 2 * (x / max(25, max x)) - 1.
// The kernels below reproduce that arithmetic operation for operation so the result is bit-identical to the CPU path:
//   * gauss1d: scipy.ndimage correlate1d, symmetric branch: fp32 line -> fp64, t = x0*w0; for k = R..1:
//     t += (x[-k] + x[+k]) * w[k]; result rounded to fp32 (the output array of gaussian_filter on fp32 input is fp32, so the
//     intermediate between the axis-0 and axis-1 passes is fp32 too).  'reflect' = (d c b a | a b c d | d c b a), any radius.
//   * radix-select histograms give the exact order statistics np.percentile interpolates between (values are >= 0 here,
//     so the fp32 bit pattern orders like the value); the interpolation itself is done by the host with numpy's own code.
//   * finalize: fp32 min, fp32 divide, exact *2, fp32 subtract -- no FMA contraction.
// All kernels are HBM/L2 streaming passes over (C, H, W); lanes run along W so every load is coalesced.
#include <algorithm>

#include "ribca_common.h"
#include "ribca_kernels.h"

namespace ribca {

__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i >= 0 && i < n) return i;
  const int p = 2 * n;
  int m = i % p;
  if (m < 0) m += p;
  return m < n ? m : p - 1 - m;
}
__device__ __forceinline__ int clamp_idx(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// planes x [H][W] fp32; axis 0 = along H (rows), axis 1 = along W.  w[k] = tap at distance k (fp64), k = 0..R.
template <int AXIS, int MODE /*0 reflect, 1 nearest*/>
__global__ __launch_bounds__(256) void gauss1d_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                      const double* __restrict__ w, int R) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= W) return;
  const size_t plane = (size_t)blockIdx.z * H * W;
  const float* p = in + plane;
  double t = __dmul_rn((double)p[(size_t)y * W + x], w[0]);
  for (int k = R; k >= 1; --k) {
    float a, b;
    if (AXIS == 0) {
      const int ya = MODE == 0 ? reflect_idx(y - k, H) : clamp_idx(y - k, H);
      const int yb = MODE == 0 ? reflect_idx(y + k, H) : clamp_idx(y + k, H);
      a = p[(size_t)ya * W + x];
      b = p[(size_t)yb * W + x];
    } else {
      const int xa = MODE == 0 ? reflect_idx(x - k, W) : clamp_idx(x - k, W);
      const int xb = MODE == 0 ? reflect_idx(x + k, W) : clamp_idx(x + k, W);
      a = p[(size_t)y * W + xa];
      b = p[(size_t)y * W + xb];
    }
    t = __dadd_rn(t, __dmul_rn(__dadd_rn((double)a, (double)b), w[k]));
  }
  out[plane + (size_t)y * W + x] = __double2float_rn(t);
}

void launch_gauss1d(const float* in, float* out, int planes, int H, int W, int axis, const double* w, int R, int mode, hipStream_t s) {
  if (planes <= 0 || H <= 0 || W <= 0) return;
  const dim3 grid((W + 255) / 256, H, planes), block(256);
  if (axis == 0 && mode == 0) hipLaunchKernelGGL((gauss1d_kernel<0, 0>), grid, block, 0, s, in, out, H, W, w, R);
  else if (axis == 1 && mode == 0) hipLaunchKernelGGL((gauss1d_kernel<1, 0>), grid, block, 0, s, in, out, H, W, w, R);
  else if (axis == 0) hipLaunchKernelGGL((gauss1d_kernel<0, 1>), grid, block, 0, s, in, out, H, W, w, R);
  else hipLaunchKernelGGL((gauss1d_kernel<1, 1>), grid, block, 0, s, in, out, H, W, w, R);
}

__global__ void u16_to_f32_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = (float)in[i];
}
void launch_u16_to_f32(const uint16_t* in, float* out, long long n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(u16_to_f32_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 16384)), dim3(256), 0, s, in, out, n);
}

// x = max(x - min(bg, cap), 0)       (preprocess.py:219-222)
__global__ void bg_subtract_kernel(float* __restrict__ x, const float* __restrict__ bg, long long n, float cap) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float b = bg[i];
    b = b > cap ? cap : b;
    const float d = __fsub_rn(x[i], b);
    x[i] = d < 0.f ? 0.f : d;     // np.clip(v, 0, None); NaN propagates like numpy
  }
}
void launch_bg_subtract(float* x, const float* bg, long long n, float cap, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(bg_subtract_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 16384)), dim3(256), 0, s, x, bg, n, cap);
}

// per-plane maximum of non-negative data via integer atomicMax on the bit pattern (out must be zeroed = 0.0f)
__global__ __launch_bounds__(256) void plane_max_kernel(const float* __restrict__ x, long long hw, unsigned int* __restrict__ out) {
  const float* p = x + (size_t)blockIdx.y * hw;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long long)gridDim.x * blockDim.x) m = fmaxf(m, p[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(&out[blockIdx.y], __float_as_uint(m));
}
void launch_plane_max(const float* x, int planes, long long hw, float* out, hipStream_t s) {
  (void)hipMemsetAsync(out, 0, planes * sizeof(float), s);
  if (planes <= 0 || hw <= 0) return;
  hipLaunchKernelGGL(plane_max_kernel, dim3((unsigned)std::min<long long>((hw + 255) / 256, 1024), planes), dim3(256), 0, s, x, hw,
                     reinterpret_cast<unsigned int*>(out));
}

// Radix-select histogram: for each plane counts keys whose bits above `shift + bits` equal prefix[plane] (mask_hi selects
// those bits; mask_hi == 0 -> every key), binned by (key >> shift) & ((1 << bits) - 1).  hist: [planes][2048] uint32, zeroed here.
__global__ __launch_bounds__(256) void radix_hist_kernel(const float* __restrict__ x, long long hw, const unsigned int* __restrict__ prefix,
                                                         unsigned int mask_hi, int shift, unsigned int bin_mask,
                                                         unsigned int* __restrict__ hist) {
  __shared__ unsigned int h[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) h[i] = 0;
  __syncthreads();
  const unsigned int* p = reinterpret_cast<const unsigned int*>(x + (size_t)blockIdx.y * hw);
  const unsigned int pre = prefix[blockIdx.y];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long long)gridDim.x * blockDim.x) {
    const unsigned int k = p[i];
    if ((k & mask_hi) == pre) atomicAdd(&h[(k >> shift) & bin_mask], 1u);
  }
  __syncthreads();
  unsigned int* o = hist + (size_t)blockIdx.y * 2048;
  for (int i = threadIdx.x; i < 2048; i += 256)
    if (h[i]) atomicAdd(&o[i], h[i]);
}
void launch_radix_hist(const float* x, int planes, long long hw, const unsigned int* prefix, unsigned int mask_hi, int shift, int bits,
                       unsigned int* hist, hipStream_t s) {
  (void)hipMemsetAsync(hist, 0, (size_t)planes * 2048 * sizeof(unsigned int), s);
  if (planes <= 0 || hw <= 0) return;
  hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)std::min<long long>((hw + 255) / 256, 512), planes), dim3(256), 0, s, x, hw, prefix,
                     mask_hi, shift, (1u << bits) - 1u, hist);
}

// per plane: mode[p] == 0 -> fill -1 (no positive pixel); else x = 2 * (min(x, clip[p]) / denom[p]) - 1 (clip = +inf disables)
__global__ void finalize_kernel(float* __restrict__ x, long long hw, const int* __restrict__ mode, const float* __restrict__ clip,
                                const float* __restrict__ denom) {
  float* p = x + (size_t)blockIdx.y * hw;
  const int md = mode[blockIdx.y];
  const float c = clip[blockIdx.y], d = denom[blockIdx.y];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long long)gridDim.x * blockDim.x) {
    if (md == 0) { p[i] = -1.0f; continue; }
    float v = p[i];
    v = v > c ? c : v;
    v = v < 0.f ? 0.f : v;
    p[i] = __fsub_rn(__fmul_rn(2.0f, __fdiv_rn(v, d)), 1.0f);
  }
}
void launch_norm_finalize(float* x, int planes, long long hw, const int* mode, const float* clip, const float* denom, hipStream_t s) {
  if (planes <= 0 || hw <= 0) return;
  hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)std::min<long long>((hw + 255) / 256, 2048), planes), dim3(256), 0, s, x, hw, mode, clip,
                     denom);
}

}  // namespace ribca
