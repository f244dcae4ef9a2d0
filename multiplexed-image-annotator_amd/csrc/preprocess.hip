// Mask -> per-label table, per-channel minimum, and the per-cell crop / soft-mask / patch kernel.
//
// Reference rows (paths under cell_type_annotation/):
//   label table      <- preprocess.py:159-181 (_cell_pos_dict), reduced to what the hot path consumes: per label the row/col
//                       min, max, sum and the pixel count (bbox centre for the crop, centroid for the CSV)
//   channel minimum  <- preprocess.py:153-157 (_move_image_range)
//   extract_patches  <- utils.py:226-253 (crop_cell) + utils.py:255-270 (smooth) for patch_size 40
//
// extract_patches reproduces the reference's arithmetic operation for operation (same fp32 / fp64 types, same summation
// order inside scipy's correlate1d, no FMA contraction), so patches are bit-identical to the CPU path:
//   S  = fp32 accumulator over  own, dil_1..dil_4  and fp64 Gaussians G_sigma(dil_j) (sigma = 1..j-1), each added as
//        S = (float)((double)S + g);   S /= 11;   S /= max(S + 1e-6)                       [all fp32, round-to-nearest]
//   dil_j(p) = [ min_{q in cell, q in window} |p-q|^2 <= j^2 ]     (binary dilation by the Euclidean disk, zero border)
//   G_sigma  = separable fp64 filter, axis 0 then axis 1, edge-replicated, taps exp(-k^2/(2 sigma^2)) / sum for |k| <= 4 sigma,
//              evaluated as  t = x0*w0;  for k = R..1:  t += (x[-k] + x[+k]) * w[k]          (scipy NI_Correlate1D, symmetric case)
//   patch[c] = (float)( (double)(img[c] - min_c) * (double)S + (double)min_c ),  zero-padded window => min_c outside
//   avg[c]   = mean of the fp64 patch over every labelled pixel of the window (any label > 0: reference quirk C.3)
// One 256-thread workgroup per cell; the 40x40 tile, its dilations and the two fp64 filter planes live in LDS (~31 KB).
#include <algorithm>

#include "ribca_common.h"
#include "ribca_kernels.h"

namespace ribca {

// ---------------------------------------------------------------------------------------------- mask min / max
__global__ void mask_minmax_kernel(const int32_t* __restrict__ mask, long long n, int32_t* __restrict__ out) {
  int mx = INT32_MIN, mn = INT32_MAX;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int v = mask[i];
    mx = max(mx, v);
    mn = min(mn, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mx = max(mx, __shfl_xor(mx, o, 64));
    mn = min(mn, __shfl_xor(mn, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&out[0], mx);
    atomicMin(&out[1], mn);
  }
}
__global__ void minmax_init_kernel(int32_t* out) { out[0] = INT32_MIN; out[1] = INT32_MAX; }

void launch_mask_max(const int32_t* mask, long long n, int32_t* out2, hipStream_t s) {
  hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, s, out2);
  if (n <= 0) return;
  const int blocks = (int)std::min<long long>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(mask_minmax_kernel, dim3(blocks), dim3(256), 0, s, mask, n, out2);
}

// ---------------------------------------------------------------------------------------------- label table
// tab_i32: [5][L] = rmin, rmax, cmin, cmax, count ; tab_u64: [2][L] = sum_r, sum_c
__global__ void label_table_init_kernel(int32_t* ti, unsigned long long* tu, int L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  ti[i] = INT32_MAX; ti[L + i] = -1; ti[2 * L + i] = INT32_MAX; ti[3 * L + i] = -1; ti[4 * L + i] = 0;
  tu[i] = 0ull; tu[L + i] = 0ull;
}
void launch_label_table_init(int32_t* ti, unsigned long long* tu, int L, hipStream_t s) {
  hipLaunchKernelGGL(label_table_init_kernel, dim3((L + 255) / 256), dim3(256), 0, s, ti, tu, L);
}

__device__ __forceinline__ void flush_run(int lab, int r, int c_first, int c_last, int cnt, unsigned long long sumc, int L, int32_t* ti,
                                          unsigned long long* tu) {
  if (lab <= 0 || lab >= L) return;
  atomicMin(&ti[lab], r);
  atomicMax(&ti[L + lab], r);
  atomicMin(&ti[2 * L + lab], c_first);
  atomicMax(&ti[3 * L + lab], c_last);
  atomicAdd(&ti[4 * L + lab], cnt);
  atomicAdd(&tu[lab], (unsigned long long)r * (unsigned long long)cnt);
  atomicAdd(&tu[L + lab], sumc);
}

// Each thread scans 8 consecutive pixels of one row and merges equal-label runs before touching the table.
__global__ __launch_bounds__(256) void label_table_kernel(const int32_t* __restrict__ mask, int H, int W, int L, int32_t* ti,
                                                          unsigned long long* tu) {
  const int gpr = (W + 7) >> 3;
  const long long total = (long long)H * gpr;
  const bool vec = (W & 7) == 0;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(idx / gpr);
    const int c0 = (int)(idx - (long long)r * gpr) << 3;
    int lab[8];
    if (vec) {
      const int4* p = reinterpret_cast<const int4*>(mask + (size_t)r * W + c0);
      const int4 a = p[0], b = p[1];
      lab[0] = a.x; lab[1] = a.y; lab[2] = a.z; lab[3] = a.w; lab[4] = b.x; lab[5] = b.y; lab[6] = b.z; lab[7] = b.w;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) lab[i] = (c0 + i < W) ? mask[(size_t)r * W + c0 + i] : 0;
    }
    int cur = 0, first = 0, last = 0, cnt = 0;
    unsigned long long sc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (lab[i] != cur) {
        flush_run(cur, r, first, last, cnt, sc, L, ti, tu);
        cur = lab[i]; first = c0 + i; cnt = 0; sc = 0;
      }
      last = c0 + i; ++cnt; sc += (unsigned long long)(c0 + i);
    }
    flush_run(cur, r, first, last, cnt, sc, L, ti, tu);
  }
}
void launch_label_table(const int32_t* mask, int H, int W, int L, int32_t* ti, unsigned long long* tu, hipStream_t s) {
  const long long total = (long long)H * ((W + 7) >> 3);
  if (total <= 0) return;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(label_table_kernel, dim3(blocks), dim3(256), 0, s, mask, H, W, L, ti, tu);
}

// ---------------------------------------------------------------------------------------------- channel minimum
// monotone uint encoding of fp32 so that unsigned atomicMin orders like the floats
__device__ __forceinline__ uint32_t f32_key(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_unkey(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
__global__ void chan_min_init_kernel(uint32_t* keys, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) keys[i] = 0xFFFFFFFFu;
}
__global__ __launch_bounds__(256) void chan_min_kernel(const float* __restrict__ img, long long hw, uint32_t* keys) {
  const int c = blockIdx.y;
  const float* p = img + (size_t)c * hw;
  float mn = INFINITY;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long long)gridDim.x * blockDim.x) mn = fminf(mn, p[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMin(&keys[c], f32_key(mn));
}
__global__ void chan_min_final_kernel(uint32_t* keys, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) reinterpret_cast<float*>(keys)[i] = f32_unkey(keys[i]);
}
void launch_channel_min(const float* img, int C, long long hw, float* out_min, hipStream_t s) {
  uint32_t* keys = reinterpret_cast<uint32_t*>(out_min);
  hipLaunchKernelGGL(chan_min_init_kernel, dim3(1), dim3(64), 0, s, keys, C);
  if (hw > 0) {
    const int bx = (int)std::min<long long>((hw + 255) / 256, 512);
    hipLaunchKernelGGL(chan_min_kernel, dim3(bx, C), dim3(256), 0, s, img, hw, keys);
  }
  hipLaunchKernelGGL(chan_min_final_kernel, dim3(1), dim3(64), 0, s, keys, C);
}

// ---------------------------------------------------------------------------------------------- crop + soft mask + patches
constexpr int PS = 40;
constexpr int NPIX = PS * PS;
constexpr int PPT = 7;  // pixels per thread: 7 * 256 >= 1600

// one separable fp64 Gaussian of the 0/1 plane {dmin <= r2} into dst (LDS), via tmp (LDS).  w[k], k = 0..R.
__device__ __forceinline__ void gauss2d(const uint8_t* dmin, int r2, const double* __restrict__ w, int R, double* tmp, double* dst) {
  // axis 0 (rows)
  for (int p = threadIdx.x; p < NPIX; p += 256) {
    const int y = p / PS, x = p - y * PS;
    double t = __dmul_rn(dmin[p] <= r2 ? 1.0 : 0.0, w[0]);
    for (int k = R; k >= 1; --k) {
      const int ya = y - k < 0 ? 0 : y - k, yb = y + k > PS - 1 ? PS - 1 : y + k;
      const double a = dmin[ya * PS + x] <= r2 ? 1.0 : 0.0;
      const double b = dmin[yb * PS + x] <= r2 ? 1.0 : 0.0;
      t = __dadd_rn(t, __dmul_rn(__dadd_rn(a, b), w[k]));
    }
    tmp[p] = t;
  }
  __syncthreads();
  // axis 1 (columns)
  for (int p = threadIdx.x; p < NPIX; p += 256) {
    const int y = p / PS, x = p - y * PS;
    double t = __dmul_rn(tmp[p], w[0]);
    for (int k = R; k >= 1; --k) {
      const int xa = x - k < 0 ? 0 : x - k, xb = x + k > PS - 1 ? PS - 1 : x + k;
      t = __dadd_rn(t, __dmul_rn(__dadd_rn(tmp[y * PS + xa], tmp[y * PS + xb]), w[k]));
    }
    dst[p] = t;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void extract_patches_kernel(PatchArgs a) {
  __shared__ uint8_t own[NPIX];
  __shared__ uint8_t lab_any[NPIX];
  __shared__ uint8_t dmin[NPIX];
  __shared__ __attribute__((aligned(16))) double tmp[NPIX];
  __shared__ __attribute__((aligned(16))) double gout[NPIX];
  __shared__ double taps[27];
  __shared__ float red_f[4];
  __shared__ double red_d[4];
  __shared__ int red_i[4];

  const int cell = blockIdx.x;
  const int tid = threadIdx.x;
  const int id = a.cell_id[cell];
  const int rmin = a.bbox[4 * cell + 0], rmax = a.bbox[4 * cell + 1], cmin = a.bbox[4 * cell + 2], cmax = a.bbox[4 * cell + 3];
  // utils.py:227-235
  const int rc = (rmin + rmax) >> 1, cc = (cmin + cmax) >> 1;
  const int r0 = rc - PS / 2 > 0 ? rc - PS / 2 : 0;
  const int r1 = r0 + PS < a.H ? r0 + PS : a.H;
  const int c0 = cc - PS / 2 > 0 ? cc - PS / 2 : 0;
  const int c1 = c0 + PS < a.W ? c0 + PS : a.W;
  const int wh = r1 - r0, ww = c1 - c0;

  if (tid < 27) taps[tid] = a.taps[tid];
  for (int p = tid; p < NPIX; p += 256) {
    const int y = p / PS, x = p - y * PS;
    int m = 0;
    if (y < wh && x < ww) m = a.mask[(size_t)(r0 + y) * a.W + (c0 + x)];
    own[p] = (m == id) ? 1 : 0;
    lab_any[p] = (m > 0) ? 1 : 0;
  }
  __syncthreads();
  // squared distance to the nearest own pixel inside the window, capped (only <= 16 matters)
  for (int p = tid; p < NPIX; p += 256) {
    const int y = p / PS, x = p - y * PS;
    int best = 255;
    for (int dy = -4; dy <= 4; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= PS) continue;
      for (int dx = -4; dx <= 4; ++dx) {
        const int xx = x + dx;
        if (xx < 0 || xx >= PS) continue;
        const int d2 = dy * dy + dx * dx;
        if (d2 < best && own[yy * PS + xx]) best = d2;
      }
    }
    dmin[p] = (uint8_t)best;
  }
  __syncthreads();

  // fp32 accumulator per owned pixel, reference order (utils.py:257-266)
  float S[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int p = tid + 256 * i;
    S[i] = (p < NPIX && own[p]) ? 1.0f : 0.0f;
  }
  for (int j = 1; j <= 4; ++j) {
    const int r2 = j * j;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int p = tid + 256 * i;
      if (p < NPIX) S[i] = __fadd_rn(S[i], dmin[p] <= r2 ? 1.0f : 0.0f);
    }
    for (int sg = 1; sg < j; ++sg) {
      const double* w = taps + (sg == 1 ? 0 : (sg == 2 ? 5 : 14));
      gauss2d(dmin, r2, w, 4 * sg, tmp, gout);
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const int p = tid + 256 * i;
        if (p < NPIX) S[i] = __double2float_rn(__dadd_rn((double)S[i], gout[p]));
      }
      __syncthreads();
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int p = tid + 256 * i;
    if (p < NPIX) {
      S[i] = __fdiv_rn(S[i], 11.0f);
      mx = fmaxf(mx, __fadd_rn(S[i], 1e-6f));
    }
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red_f[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int p = tid + 256 * i;
    if (p < NPIX) {
      S[i] = __fdiv_rn(S[i], mx);
      cnt += lab_any[p];
    }
  }
  if (a.avg_int) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((tid & 63) == 0) red_i[tid >> 6] = cnt;
    __syncthreads();
    cnt = red_i[0] + red_i[1] + red_i[2] + red_i[3];
  }

  for (int c = 0; c < a.C; ++c) {
    const float mn = a.chan_min[c];
    const float* src = a.img + (size_t)c * a.H * a.W;
    float* dst = a.patches + ((size_t)cell * a.C + c) * NPIX;
    double part = 0.0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int p = tid + 256 * i;
      if (p < NPIX) {
        const int y = p / PS, x = p - y * PS;
        float t = 0.0f;
        if (y < wh && x < ww) t = __fsub_rn(src[(size_t)(r0 + y) * a.W + (c0 + x)], mn);
        const double v = __dadd_rn(__dmul_rn((double)t, (double)S[i]), (double)mn);
        dst[p] = __double2float_rn(v);
        if (lab_any[p]) part += v;
      }
    }
    if (a.avg_int) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
      __syncthreads();
      if ((tid & 63) == 0) red_d[tid >> 6] = part;
      __syncthreads();
      if (tid == 0) a.avg_int[(size_t)cell * a.C + c] = ((red_d[0] + red_d[1]) + (red_d[2] + red_d[3])) / (double)cnt;
    }
  }
}

void launch_extract_patches(const PatchArgs& a, hipStream_t s) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(extract_patches_kernel, dim3(a.n), dim3(256), 0, s, a);
}

}  // namespace ribca
