// crop_cell + smooth + skimage.transform.resize for cell_size != 30  (reference preprocess.py:78,106 with utils.py:226-270):
//   patch_size ps = int(40 * cell_size / 30); the soft-masked fp64 (C, ps, ps) patch is built exactly as in preprocess.hip,
//   then  resize(patch, (C, 40, 40), order=0, anti_aliasing=True, preserve_range=True):
//     * ps > 40: Gaussian pre-filter per channel plane, sigma = (ps/40 - 1)/2, truncate 4, mode 'mirror', fp64, axis y then x
//       (scipy.ndimage.gaussian_filter on the fp64 array: no fp32 rounding between the passes);
//     * nearest-neighbour grid sampling  src = floor(((o + 0.5) * (ps/40) - 0.5) + 0.5)  (scipy.ndimage.zoom, order 0,
//       grid_mode=True) -- the 40 source indices are computed on the host with the same fp64 operations and passed in;
//     * the fp64 result is rounded to fp32 once (torch.tensor(temp, dtype=float32), preprocess.py:124).
//   Only the 40 sampled rows / columns of the filtered plane are ever evaluated (the filter is separable: the x pass needs the
//   y pass on the sampled rows only), with the same operation order as scipy's correlate1d, so the result is bit-identical.
//   skimage's final clip to the input range can only change a value by an fp64 rounding error of the convex filter sum; it
//   disappears in the fp32 rounding and is not replayed.
// One 1024-thread workgroup per cell, up to 8 pixels per thread (ps <= 90), planes in dynamic LDS (<= 154 KB).
#include "ribca_common.h"
#include "ribca_kernels.h"

namespace ribca {

constexpr int SC_THREADS = 1024;
constexpr int SC_PPT = 8;
constexpr int SC_OUT = 40;

__device__ __forceinline__ int mirror_idx(int i, int n) {   // ndimage 'mirror': d c b | a b c d | c b a
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// separable fp64 Gaussian of the 0/1 plane {dmin <= r2}, edge-replicated ('nearest'), via tmp into dst
__device__ __forceinline__ void gauss2d_n(const uint8_t* dmin, int r2, const double* __restrict__ w, int R, double* tmp, double* dst, int ps) {
  const int npix = ps * ps;
  for (int p = threadIdx.x; p < npix; p += SC_THREADS) {
    const int y = p / ps, x = p - y * ps;
    double t = __dmul_rn(dmin[p] <= r2 ? 1.0 : 0.0, w[0]);
    for (int k = R; k >= 1; --k) {
      const int ya = y - k < 0 ? 0 : y - k, yb = y + k > ps - 1 ? ps - 1 : y + k;
      const double a = dmin[ya * ps + x] <= r2 ? 1.0 : 0.0;
      const double b = dmin[yb * ps + x] <= r2 ? 1.0 : 0.0;
      t = __dadd_rn(t, __dmul_rn(__dadd_rn(a, b), w[k]));
    }
    tmp[p] = t;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < npix; p += SC_THREADS) {
    const int y = p / ps, x = p - y * ps;
    double t = __dmul_rn(tmp[p], w[0]);
    for (int k = R; k >= 1; --k) {
      const int xa = x - k < 0 ? 0 : x - k, xb = x + k > ps - 1 ? ps - 1 : x + k;
      t = __dadd_rn(t, __dmul_rn(__dadd_rn(tmp[y * ps + xa], tmp[y * ps + xb]), w[k]));
    }
    dst[p] = t;
  }
  __syncthreads();
}

__global__ __launch_bounds__(SC_THREADS) void extract_patches_scaled_kernel(PatchArgs a, int ps, const double* __restrict__ aa_taps, int aa_radius,
                                                                            const int32_t* __restrict__ src_index) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int npix = ps * ps;
  const int npad = (npix + 15) & ~15;
  double* tmp = reinterpret_cast<double*>(lds);
  double* gout = tmp + npad;
  double* taps = gout + npad;                 // 27 smooth taps + up to 16 anti-alias taps
  double* aat = taps + 27;
  double* red_d = aat + 16;                   // 16
  float* red_f = reinterpret_cast<float*>(red_d + 16);   // 16
  int* red_i = reinterpret_cast<int*>(red_f + 16);       // 16
  int* sidx = red_i + 16;                     // 40
  uint8_t* own = reinterpret_cast<uint8_t*>(sidx + SC_OUT);
  uint8_t* lab_any = own + npad;
  uint8_t* dmin = lab_any + npad;

  const int cell = blockIdx.x;
  const int tid = threadIdx.x;
  const int id = a.cell_id[cell];
  const int rmin = a.bbox[4 * cell + 0], rmax = a.bbox[4 * cell + 1], cmin = a.bbox[4 * cell + 2], cmax = a.bbox[4 * cell + 3];
  // utils.py:227-235 with x_mean - ps/2 in floating point, truncated by int()
  const int rc = (rmin + rmax) >> 1, cc = (cmin + cmax) >> 1;
  const double hr = (double)rc - (double)ps / 2.0, hc = (double)cc - (double)ps / 2.0;
  const int r0 = hr > 0.0 ? (int)hr : 0;
  const int r1 = r0 + ps < a.H ? r0 + ps : a.H;
  const int c0 = hc > 0.0 ? (int)hc : 0;
  const int c1 = c0 + ps < a.W ? c0 + ps : a.W;
  const int wh = r1 - r0, ww = c1 - c0;

  if (tid < 27) taps[tid] = a.taps[tid];
  if (tid >= 32 && tid < 32 + 16) aat[tid - 32] = (aa_radius > 0 && tid - 32 <= aa_radius) ? aa_taps[tid - 32] : 0.0;   // aa_taps may be NULL
  if (tid >= 64 && tid < 64 + SC_OUT) sidx[tid - 64] = src_index[tid - 64];
  for (int p = tid; p < npix; p += SC_THREADS) {
    const int y = p / ps, x = p - y * ps;
    int m = 0;
    if (y < wh && x < ww) m = a.mask[(size_t)(r0 + y) * a.W + (c0 + x)];
    own[p] = (m == id) ? 1 : 0;
    lab_any[p] = (m > 0) ? 1 : 0;
  }
  __syncthreads();
  for (int p = tid; p < npix; p += SC_THREADS) {
    const int y = p / ps, x = p - y * ps;
    int best = 255;
    for (int dy = -4; dy <= 4; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= ps) continue;
      for (int dx = -4; dx <= 4; ++dx) {
        const int xx = x + dx;
        if (xx < 0 || xx >= ps) continue;
        const int d2 = dy * dy + dx * dx;
        if (d2 < best && own[yy * ps + xx]) best = d2;
      }
    }
    dmin[p] = (uint8_t)best;
  }
  __syncthreads();

  float S[SC_PPT];
#pragma unroll
  for (int i = 0; i < SC_PPT; ++i) {
    const int p = tid + SC_THREADS * i;
    S[i] = (p < npix && own[p]) ? 1.0f : 0.0f;
  }
  for (int j = 1; j <= 4; ++j) {
    const int r2 = j * j;
#pragma unroll
    for (int i = 0; i < SC_PPT; ++i) {
      const int p = tid + SC_THREADS * i;
      if (p < npix) S[i] = __fadd_rn(S[i], dmin[p] <= r2 ? 1.0f : 0.0f);
    }
    for (int sg = 1; sg < j; ++sg) {
      const double* w = taps + (sg == 1 ? 0 : (sg == 2 ? 5 : 14));
      gauss2d_n(dmin, r2, w, 4 * sg, tmp, gout, ps);
#pragma unroll
      for (int i = 0; i < SC_PPT; ++i) {
        const int p = tid + SC_THREADS * i;
        if (p < npix) S[i] = __double2float_rn(__dadd_rn((double)S[i], gout[p]));
      }
      __syncthreads();
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < SC_PPT; ++i) {
    const int p = tid + SC_THREADS * i;
    if (p < npix) {
      S[i] = __fdiv_rn(S[i], 11.0f);
      mx = fmaxf(mx, __fadd_rn(S[i], 1e-6f));
    }
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red_f[tid >> 6] = mx;
  __syncthreads();
  mx = red_f[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) mx = fmaxf(mx, red_f[i]);
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < SC_PPT; ++i) {
    const int p = tid + SC_THREADS * i;
    if (p < npix) {
      S[i] = __fdiv_rn(S[i], mx);
      cnt += lab_any[p];
    }
  }
  if (a.avg_int) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((tid & 63) == 0) red_i[tid >> 6] = cnt;
    __syncthreads();
    cnt = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) cnt += red_i[i];
  }

  for (int c = 0; c < a.C; ++c) {
    const float mn = a.chan_min[c];
    const float* src = a.img + (size_t)c * a.H * a.W;
    float* dst = a.patches + ((size_t)cell * a.C + c) * (SC_OUT * SC_OUT);
    double part = 0.0;
    // the fp64 soft-masked plane of this channel (crop_cell's marker_a)
#pragma unroll
    for (int i = 0; i < SC_PPT; ++i) {
      const int p = tid + SC_THREADS * i;
      if (p < npix) {
        const int y = p / ps, x = p - y * ps;
        float t = 0.0f;
        if (y < wh && x < ww) t = __fsub_rn(src[(size_t)(r0 + y) * a.W + (c0 + x)], mn);
        const double v = __dadd_rn(__dmul_rn((double)t, (double)S[i]), (double)mn);
        gout[p] = v;
        if (lab_any[p]) part += v;
      }
    }
    __syncthreads();
    if (aa_radius > 0) {
      // y pass on the 40 sampled rows
      for (int q = tid; q < SC_OUT * ps; q += SC_THREADS) {
        const int oy = q / ps, x = q - oy * ps;
        const int iy = sidx[oy];
        double t = __dmul_rn(gout[iy * ps + x], aat[0]);
        for (int k = aa_radius; k >= 1; --k) {
          const int ya = mirror_idx(iy - k, ps), yb = mirror_idx(iy + k, ps);
          t = __dadd_rn(t, __dmul_rn(__dadd_rn(gout[ya * ps + x], gout[yb * ps + x]), aat[k]));
        }
        tmp[q] = t;
      }
      __syncthreads();
      // x pass on the 40 sampled columns of those rows
      for (int q = tid; q < SC_OUT * SC_OUT; q += SC_THREADS) {
        const int oy = q / SC_OUT, ox = q - oy * SC_OUT;
        const int ix = sidx[ox];
        const double* row = tmp + oy * ps;
        double t = __dmul_rn(row[ix], aat[0]);
        for (int k = aa_radius; k >= 1; --k) {
          const int xa = mirror_idx(ix - k, ps), xb = mirror_idx(ix + k, ps);
          t = __dadd_rn(t, __dmul_rn(__dadd_rn(row[xa], row[xb]), aat[k]));
        }
        dst[q] = __double2float_rn(t);
      }
    } else {
      for (int q = tid; q < SC_OUT * SC_OUT; q += SC_THREADS) {
        const int oy = q / SC_OUT, ox = q - oy * SC_OUT;
        dst[q] = __double2float_rn(gout[sidx[oy] * ps + sidx[ox]]);
      }
    }
    if (a.avg_int) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
      if ((tid & 63) == 0) red_d[tid >> 6] = part;
      __syncthreads();
      if (tid == 0) {
        double s = 0.0;
        for (int i = 0; i < 16; ++i) s += red_d[i];
        a.avg_int[(size_t)cell * a.C + c] = s / (double)cnt;
      }
    }
    __syncthreads();   // gout / tmp / red_d are reused by the next channel
  }
}

size_t patches_scaled_lds_bytes(int ps) {
  const int npix = ps * ps;
  const int npad = (npix + 15) & ~15;
  return (size_t)npad * 16 + (27 + 16 + 16) * 8 + 16 * 4 + 16 * 4 + SC_OUT * 4 + (size_t)npad * 3;
}

int launch_extract_patches_scaled(const PatchArgs& a, int ps, const double* aa_taps, int aa_radius, const int32_t* src_index, hipStream_t s) {
  if (a.n <= 0) return 0;
  const size_t lds = patches_scaled_lds_bytes(ps);
  if (ps < 4 || ps * ps > SC_THREADS * SC_PPT || lds > 160 * 1024 || aa_radius < 0 || aa_radius > 15) return 1;
  static size_t attr = 0;
  if (lds > attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&extract_patches_scaled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return 2;
    attr = lds;
  }
  hipLaunchKernelGGL(extract_patches_scaled_kernel, dim3(a.n), dim3(SC_THREADS), lds, s, a, ps, aa_taps, aa_radius, src_index);
  return 0;
}

}  // namespace ribca
