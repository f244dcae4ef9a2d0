// Small kernels of the ViT forward: LayerNorm -> packed-split, the fp32 patch-embed conv (im2col on the fly),
// CLS rows, final LayerNorm + head + softmax, and the one-off weight packer.
// Reference semantics: timm Block.norm1/norm2 and VisionTransformer.norm are nn.LayerNorm(eps=1e-6) (model.py:66-88);
// patch embedding is Conv2d(C, D, 4, 4) with K order (c, ky, kx) and token order py*10+px; head + softmax at model.py:402-404.
#include "ribca_common.h"
#include "ribca_kernels.h"

namespace ribca {

constexpr float kLnEps = 1e-6f;

// One wave per row; D <= 768 (<= 3 float4 per lane).  Two-pass mean / biased variance in registers.
__global__ __launch_bounds__(256) void layernorm_ps_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, uint16_t* __restrict__ out, int ldo, int M,
                                                           int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nv = D >> 2;
  const float4* zr = reinterpret_cast<const float4*>(z + (size_t)row * ldz);
  float4 x[3];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int v = lane + 64 * i;
    x[i] = v < nv ? zr[v] : float4{0.f, 0.f, 0.f, 0.f};
    sum += (x[i].x + x[i].y) + (x[i].z + x[i].w);
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int v = lane + 64 * i;
    if (v < nv) {
      const float a = x[i].x - mean, b = x[i].y - mean, c = x[i].z - mean, d = x[i].w - mean;
      sq += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + kLnEps);
  uint16_t* orow = out + (size_t)row * ldo;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int v = lane + 64 * i;
    if (v < nv) {
      const float4 gm = reinterpret_cast<const float4*>(gamma)[v];
      const float4 bt = reinterpret_cast<const float4*>(beta)[v];
      float y[4] = {(x[i].x - mean) * rstd * gm.x + bt.x, (x[i].y - mean) * rstd * gm.y + bt.y,
                    (x[i].z - mean) * rstd * gm.z + bt.z, (x[i].w - mean) * rstd * gm.w + bt.w};
      ps_store4(orow, 4 * v, y);
    }
  }
}

void launch_layernorm_ps(const float* z, int ldz, const float* gamma, const float* beta, uint16_t* out, int ldo, int M, int D,
                         hipStream_t s) {
  if (M <= 0) return;
  hipLaunchKernelGGL(layernorm_ps_kernel, dim3((M + 3) / 4), dim3(256), 0, s, z, ldz, gamma, beta, out, ldo, M, D);
}

// Patch embedding in plain fp32 (timm PatchEmbed: Conv2d(C, D, k=4, s=4), model.py:47):
//   z[cell*101 + 1 + t][n] = sum_{c,ky,kx} W[n][c][ky][kx] * x[cell][src[c]][4py+ky][4px+kx] + bias[n] + pos[1+t][n]
// It is 1-2 % of the FLOPs but feeds the first LayerNorm directly: background tokens embed to ~pos_embed (|z| ~ 0.02) by
// cancellation of O(1) terms, and LayerNorm rescales that row to unit variance, so a 2^-16-relative product error
// (split 16-bit operands) is amplified ~50x there.  fp32 FMA keeps this stage at the reference's own precision.
// Classic 64x64 LDS-tiled SGEMM, one input channel (16 taps) per K step, im2col done on the fly from the fp32 patches.
// PS = true: z is the packed-split residual stream of the classifiers (uint16 [rows][ldz], fp16 hi + lo), else fp32.
template <bool PS>
__global__ __launch_bounds__(256) void embed_f32_kernel(const float* __restrict__ patches, int c_img, const int* __restrict__ src_chan,
                                                        int C, const float* __restrict__ w /*[D][C*16]*/, const float* __restrict__ bias,
                                                        const float* __restrict__ pos, void* __restrict__ zv, int ldz, int D, int M) {
  __shared__ float As[16][64 + 4];  // [k][row]
  __shared__ float Ws[16][64 + 4];  // [k][col]
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;  // 16 x 16 threads, 4 x 4 outputs each
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  // staging: thread -> (row = tid>>2, ky = tid&3) loads 4 kx for A; (col = tid>>2, 4 taps) for W
  const int srow = tid >> 2, sq = tid & 3;
  const int gm = m0 + srow;
  const bool a_ok = gm < M;
  const int cell = a_ok ? gm / 100 : 0;
  const int t = a_ok ? gm - cell * 100 : 0;
  const int py = t / 10, px = t - py * 10;
  const float* abase = patches + ((size_t)cell * c_img * 40 + (4 * py + sq)) * 40 + 4 * px;  // + src*1600
  const int gn = n0 + srow;
  const bool w_ok = gn < D;
  const float* wbase = w + (size_t)(w_ok ? gn : 0) * (C * 16) + sq * 4;
  float acc[4][4] = {};
  for (int c = 0; c < C; ++c) {
    const int sc = src_chan[c];
    float4 av = {0.f, 0.f, 0.f, 0.f};
    if (a_ok) {
      if (sc < 0) av = float4{-1.f, -1.f, -1.f, -1.f};
      else av = *reinterpret_cast<const float4*>(abase + (size_t)sc * 1600);
    }
    float4 wv = {0.f, 0.f, 0.f, 0.f};
    if (w_ok) wv = *reinterpret_cast<const float4*>(wbase + c * 16);
    __syncthreads();
    As[sq * 4 + 0][srow] = av.x; As[sq * 4 + 1][srow] = av.y; As[sq * 4 + 2][srow] = av.z; As[sq * 4 + 3][srow] = av.w;
    Ws[sq * 4 + 0][srow] = wv.x; Ws[sq * 4 + 1][srow] = wv.y; Ws[sq * 4 + 2][srow] = wv.z; Ws[sq * 4 + 3][srow] = wv.w;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 a4 = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
      const float4 b4 = *reinterpret_cast<const float4*>(&Ws[k][tx * 4]);
      const float a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
  }
  const int n = n0 + tx * 4;
  if (n >= D) return;
  const float4 bv = *reinterpret_cast<const float4*>(bias + n);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
    const int cl = m / 100, tt = m - cl * 100;
    const float4 pe = *reinterpret_cast<const float4*>(pos + (size_t)(1 + tt) * D + n);
    float4 o;
    o.x = acc[i][0] + bv.x + pe.x; o.y = acc[i][1] + bv.y + pe.y; o.z = acc[i][2] + bv.z + pe.z; o.w = acc[i][3] + bv.w + pe.w;
    const size_t zrow = ((size_t)cl * kTokens + 1 + tt) * ldz;
    if constexpr (PS) {
      const float v4[4] = {o.x, o.y, o.z, o.w};
      ps_store4(static_cast<uint16_t*>(zv) + zrow, n, v4);
    } else {
      *reinterpret_cast<float4*>(static_cast<float*>(zv) + zrow + n) = o;
    }
  }
}

void launch_embed_f32(const float* patches, int c_img, const int* src_chan, int C, const float* w, const float* bias, const float* pos,
                      float* z, int ldz, int D, int cells, hipStream_t s) {
  const int M = cells * 100;
  if (M <= 0) return;
  hipLaunchKernelGGL(embed_f32_kernel<false>, dim3((M + 63) / 64, (D + 63) / 64), dim3(256), 0, s, patches, c_img, src_chan, C, w, bias, pos,
                     (void*)z, ldz, D, M);
}
void launch_embed_ps(const float* patches, int c_img, const int* src_chan, int C, const float* w, const float* bias, const float* pos,
                     uint16_t* z, int ldz, int D, int cells, hipStream_t s) {
  const int M = cells * 100;
  if (M <= 0) return;
  hipLaunchKernelGGL(embed_f32_kernel<true>, dim3((M + 63) / 64, (D + 63) / 64), dim3(256), 0, s, patches, c_img, src_chan, C, w, bias, pos,
                     (void*)z, ldz, D, M);
}

// packed-split residual stream: CLS rows (model.py:49-51), one thread per 4 columns
__global__ void cls_rows_ps_kernel(uint16_t* __restrict__ z, int ldz, const float* __restrict__ cls, const float* __restrict__ pos, int D,
                                   int cells, int T) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int dq = D >> 2;
  if (idx >= cells * dq) return;
  const int cell = idx / dq, d = 4 * (idx - cell * dq);
  const float4 a = *reinterpret_cast<const float4*>(cls + d), b = *reinterpret_cast<const float4*>(pos + d);
  const float v4[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
  ps_store4(z + (size_t)cell * T * ldz, d, v4);
}
void launch_cls_rows_ps(uint16_t* z, int ldz, const float* cls, const float* pos, int D, int cells, int tokens_per_cell, hipStream_t s) {
  if (cells <= 0) return;
  hipLaunchKernelGGL(cls_rows_ps_kernel, dim3((cells * (D >> 2) + 255) / 256), dim3(256), 0, s, z, ldz, cls, pos, D, cells, tokens_per_cell);
}

// Row statistics of a packed-split residual stream, as the folded-LayerNorm GEMM epilogues read them: rowstat[m] = (rstd, mean),
// two-pass mean / biased variance over hi + lo (exact in fp32), eps 1e-6.  One wave per row, D <= 1024.  Used once per forward chunk,
// behind the patch embedding; every later LayerNorm's statistics come out of the residual GEMM epilogues (gemm_split16.hip).
// RECENTRE: the row is rewritten as z - mean first (see EpiResidPS: a per-row constant is unobservable) and the statistics are those
// of the rewritten row.
template <bool RECENTRE>
__global__ __launch_bounds__(256) void row_stats_ps_kernel(uint16_t* __restrict__ z, int ldz, int M, int D, float2* __restrict__ rowstat) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int ng = D >> 3;
  uint16_t* zr = z + (size_t)row * ldz;
  float x[2][8];
  auto load_sum = [&]() {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = lane + 64 * i;
      if (g < ng) {
        const uint4 hi = *reinterpret_cast<const uint4*>(zr + 16 * g), lo = *reinterpret_cast<const uint4*>(zr + 16 * g + 8);
        const uint32_t hw[4] = {hi.x, hi.y, hi.z, hi.w}, lw[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2 h = unpack_f16(hw[j]), l = unpack_f16(lw[j]);
          x[i][2 * j] = h[0] + l[0]; x[i][2 * j + 1] = h[1] + l[1];
        }
        sum += ((x[i][0] + x[i][1]) + (x[i][2] + x[i][3])) + ((x[i][4] + x[i][5]) + (x[i][6] + x[i][7]));
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[i][j] = 0.f;
      }
    }
    return sum;
  };
  float mean = wave_sum(load_sum()) / (float)D;
  if constexpr (RECENTRE) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = lane + 64 * i;
      if (g < ng) {
        float a[4] = {x[i][0] - mean, x[i][1] - mean, x[i][2] - mean, x[i][3] - mean};
        float b[4] = {x[i][4] - mean, x[i][5] - mean, x[i][6] - mean, x[i][7] - mean};
        uint2 ah, al, bh, bl;
        split4(a, ah, al);
        split4(b, bh, bl);
        *reinterpret_cast<uint4*>(zr + 16 * g) = uint4{ah.x, ah.y, bh.x, bh.y};
        *reinterpret_cast<uint4*>(zr + 16 * g + 8) = uint4{al.x, al.y, bl.x, bl.y};
        const uint32_t hw[4] = {ah.x, ah.y, bh.x, bh.y}, lw[4] = {al.x, al.y, bl.x, bl.y};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2 h = unpack_f16(hw[j]), l = unpack_f16(lw[j]);
          x[i][2 * j] = h[0] + l[0]; x[i][2 * j + 1] = h[1] + l[1];
        }
        sum += ((x[i][0] + x[i][1]) + (x[i][2] + x[i][3])) + ((x[i][4] + x[i][5]) + (x[i][6] + x[i][7]));
      }
    }
    mean = wave_sum(sum) / (float)D;
  }
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (lane + 64 * i < ng) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = x[i][j] - mean; sq += d * d; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + kLnEps);
  if (lane == 0) rowstat[row] = float2{rstd, mean};
}
void launch_row_stats_ps(uint16_t* z, int ldz, int M, int D, float2* rowstat, bool recentre, hipStream_t s) {
  if (M <= 0) return;
  if (recentre) hipLaunchKernelGGL(row_stats_ps_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, s, z, ldz, M, D, rowstat);
  else hipLaunchKernelGGL(row_stats_ps_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, s, z, ldz, M, D, rowstat);
}

// rowstat[m] = (rstd, mean) from the per-column-tile (mean, centred sum of squares) pairs the residual epilogue wrote:
// tiles are combined in tile order with Chan's update, so the result does not depend on which workgroup finished first.
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float2* __restrict__ part, int T, int M, int bn, int N, float2* __restrict__ rowstat) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float2 p = part[m];
  float mean = p.x, m2 = p.y, cnt = (float)(N < bn ? N : bn);
  for (int t = 1; t < T; ++t) {
    const int rem = N - t * bn;
    const float nb = (float)(rem < bn ? rem : bn);
    p = part[(size_t)t * M + m];
    const float delta = p.x - mean, tot = cnt + nb;
    mean += delta * (nb / tot);
    m2 += p.y + delta * delta * (cnt * nb / tot);
    cnt = tot;
  }
  const float rstd = 1.0f / sqrtf(m2 / (float)N + kLnEps);
  rowstat[m] = float2{rstd, mean};
}
void launch_ln_finalize(const float2* part, int T, int M, int bn, int N, float2* rowstat, hipStream_t s) {
  if (M <= 0) return;
  hipLaunchKernelGGL(ln_finalize_kernel, dim3((M + 255) / 256), dim3(256), 0, s, part, T, M, bn, N, rowstat);
}

// z[cell*T + 0][:] = cls_token + pos_embed[0]   (model.py:49-51; markerImputer.py:197-199)
__global__ void cls_rows_kernel(float* __restrict__ z, int ldz, const float* __restrict__ cls, const float* __restrict__ pos, int D,
                                int cells, int T) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= cells * D) return;
  const int cell = idx / D, d = idx - cell * D;
  z[(size_t)cell * T * ldz + d] = cls[d] + pos[d];
}
void launch_cls_rows(float* z, int ldz, const float* cls, const float* pos, int D, int cells, int tokens_per_cell, hipStream_t s) {
  if (cells <= 0) return;
  hipLaunchKernelGGL(cls_rows_kernel, dim3((cells * D + 255) / 256), dim3(256), 0, s, z, ldz, cls, pos, D, cells, tokens_per_cell);
}

// LayerNorm of selected rows: out_ps row (cell*S + j) = LN(z row (cell*T + sel[j])).  Same arithmetic as layernorm_ps_kernel.
// PS_IN: z is a packed-split residual stream (fp16 hi + lo, ldz in 16-bit elements: the imputer's blocks on the folded path, round 5) --
// a row is decoded as hi + lo in fp32 (exact) and normalised with its own statistics, whatever constant the re-centring of the residual
// epilogues has taken off it (LayerNorm is shift-invariant)
template <bool PS_IN>
__global__ __launch_bounds__(256) void layernorm_gather_ps_kernel(const void* __restrict__ zv, int ldz, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, uint16_t* __restrict__ out, int ldo,
                                                                  int rows_out, int T, int S, const int* __restrict__ sel, int D) {
  const int lane = threadIdx.x & 63;
  const int ro = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ro >= rows_out) return;
  const int cell = ro / S, j = ro - cell * S;
  const int nv = D >> 2;
  const size_t row_in = (size_t)cell * T + sel[j];
  float4 x[3];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int v = lane + 64 * i;
    x[i] = float4{0.f, 0.f, 0.f, 0.f};
    if (v < nv) {
      if constexpr (PS_IN) {
        const uint16_t* zr = static_cast<const uint16_t*>(zv) + row_in * ldz + ps_off(4 * v);
        const uint2 h = *reinterpret_cast<const uint2*>(zr), l = *reinterpret_cast<const uint2*>(zr + 8);
        const f32x2 h0 = unpack_f16(h.x), h1 = unpack_f16(h.y), l0 = unpack_f16(l.x), l1 = unpack_f16(l.y);
        x[i] = float4{h0[0] + l0[0], h0[1] + l0[1], h1[0] + l1[0], h1[1] + l1[1]};
      } else {
        x[i] = reinterpret_cast<const float4*>(static_cast<const float*>(zv) + row_in * ldz)[v];
      }
    }
    sum += (x[i].x + x[i].y) + (x[i].z + x[i].w);
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int v = lane + 64 * i;
    if (v < nv) {
      const float a = x[i].x - mean, b = x[i].y - mean, c = x[i].z - mean, d = x[i].w - mean;
      sq += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + kLnEps);
  uint16_t* orow = out + (size_t)ro * ldo;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int v = lane + 64 * i;
    if (v < nv) {
      const float4 gm = reinterpret_cast<const float4*>(gamma)[v];
      const float4 bt = reinterpret_cast<const float4*>(beta)[v];
      float y[4] = {(x[i].x - mean) * rstd * gm.x + bt.x, (x[i].y - mean) * rstd * gm.y + bt.y,
                    (x[i].z - mean) * rstd * gm.z + bt.z, (x[i].w - mean) * rstd * gm.w + bt.w};
      ps_store4(orow, 4 * v, y);
    }
  }
}
void launch_layernorm_gather_ps(const float* z, int ldz, const float* gamma, const float* beta, uint16_t* out, int ldo, int cells, int T,
                                int S, const int* sel, int D, hipStream_t s) {
  const int rows = cells * S;
  if (rows <= 0) return;
  hipLaunchKernelGGL(layernorm_gather_ps_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, s, z, ldz, gamma, beta, out, ldo, rows, T, S, sel, D);
}
void launch_layernorm_gather_ps_from_ps(const uint16_t* z_ps, int ldz, const float* gamma, const float* beta, uint16_t* out, int ldo, int cells,
                                        int T, int S, const int* sel, int D, hipStream_t s) {
  const int rows = cells * S;
  if (rows <= 0) return;
  if (D > 768 || D % 4 != 0) { launch_error("launch_layernorm_gather_ps_from_ps: D = %d (a multiple of 4, at most 768)", D); return; }
  hipLaunchKernelGGL(layernorm_gather_ps_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, s, z_ps, ldz, gamma, beta, out, ldo, rows, T, S, sel, D);
}

// out_ps row (cell*S + j) = hi/lo split of fp32 src row (cell*T + sel[j]) (K values, zero padded to Kp).  One thread per 4 values.
__global__ __launch_bounds__(256) void rows_to_ps_kernel(const float* __restrict__ src, int K, uint16_t* __restrict__ out, int ldo, int Kp,
                                                         long long total, int T, int S, const int* __restrict__ sel) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int kq = Kp >> 2;
  const long long ro = idx / kq;
  const int q = (int)(idx - ro * kq);
  const long long cell = ro / S;
  const int j = (int)(ro - cell * S);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (4 * q < K) {
    const float4 p = *reinterpret_cast<const float4*>(src + ((size_t)cell * T + sel[j]) * K + 4 * q);
    v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w;
  }
  ps_store4(out + (size_t)ro * ldo, 4 * q, v);
}
void launch_rows_to_ps(const float* src, int K, uint16_t* out, int ldo, int Kp, int cells, int T, int S, const int* sel, hipStream_t s) {
  const long long total = (long long)cells * S * (Kp >> 2);
  if (total <= 0) return;
  hipLaunchKernelGGL(rows_to_ps_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, K, out, ldo, Kp, total, T, S, sel);
}

// z row (cell*T + sel[j]) = a + table[sel[j]]     (decoder mask tokens: mask_token + decoder_pos_embed, markerImputer.py:213-219)
__global__ void fill_rows_kernel(float* __restrict__ z, int ldz, const float* __restrict__ a, const float* __restrict__ table, int D,
                                 long long total, int T, int S, const int* __restrict__ sel) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long long ro = idx / D;
  const int d = (int)(idx - ro * D);
  const long long cell = ro / S;
  const int t = sel[(int)(ro - cell * S)];
  z[((size_t)cell * T + t) * ldz + d] = a[d] + table[(size_t)t * D + d];
}
void launch_fill_rows(float* z, int ldz, const float* a, const float* table, int D, int cells, int T, int S, const int* sel, hipStream_t s) {
  const long long total = (long long)cells * S * D;
  if (total <= 0) return;
  hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, z, ldz, a, table, D, total, T, S, sel);
}

// final LayerNorm of the CLS row -> Linear(D, K) -> softmax(dim=1), all fp32.  One wave per cell, K <= 16.
template <bool PS>
__global__ __launch_bounds__(256) void head_softmax_kernel(const void* __restrict__ zv, int ldz, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ hw,
                                                           const float* __restrict__ hb, float* __restrict__ probs, int D, int K,
                                                           int cells) {
  const int lane = threadIdx.x & 63;
  const int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cell >= cells) return;
  float x[12];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int d = lane + 64 * i;
    if constexpr (PS) {
      const uint16_t* zr = static_cast<const uint16_t*>(zv) + (size_t)cell * kTokens * ldz;
      x[i] = d < D ? f16_to_f32(zr[ps_off(d)]) + f16_to_f32(zr[ps_off(d) + 8]) : 0.f;
    } else {
      const float* zr = static_cast<const float*>(zv) + (size_t)cell * kTokens * ldz;
      x[i] = d < D ? zr[d] : 0.f;
    }
    sum += x[i];
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int d = lane + 64 * i;
    if (d < D) { const float a = x[i] - mean; sq += a * a; }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + kLnEps);
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int d = lane + 64 * i;
    x[i] = d < D ? (x[i] - mean) * rstd * gamma[d] + beta[d] : 0.f;
  }
  float logit[16];
  float mx = -INFINITY;
  for (int k = 0; k < K; ++k) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int d = lane + 64 * i;
      if (d < D) acc += x[i] * hw[(size_t)k * D + d];
    }
    acc = wave_sum(acc) + hb[k];
    logit[k] = acc;
    mx = fmaxf(mx, acc);
  }
  float den = 0.f;
  for (int k = 0; k < K; ++k) { logit[k] = expf(logit[k] - mx); den += logit[k]; }
  if (lane == 0)
    for (int k = 0; k < K; ++k) probs[(size_t)cell * K + k] = logit[k] / den;
}

void launch_head_softmax(const float* z, int ldz, const float* gamma, const float* beta, const float* hw, const float* hb, float* probs,
                         int D, int K, int cells, hipStream_t s) {
  if (cells <= 0) return;
  hipLaunchKernelGGL(head_softmax_kernel<false>, dim3((cells + 3) / 4), dim3(256), 0, s, (const void*)z, ldz, gamma, beta, hw, hb, probs, D, K, cells);
}
void launch_head_softmax_ps(const uint16_t* z, int ldz, const float* gamma, const float* beta, const float* hw, const float* hb, float* probs,
                            int D, int K, int cells, hipStream_t s) {
  if (cells <= 0) return;
  hipLaunchKernelGGL(head_softmax_kernel<true>, dim3((cells + 3) / 4), dim3(256), 0, s, (const void*)z, ldz, gamma, beta, hw, hb, probs, D, K, cells);
}

// fp32 nn.Linear weight [N][K] -> packed-split fp16 [Np][2*Kp], zero padded.  One thread per 4 consecutive k.
__global__ void pack_weight_kernel(const float* __restrict__ w, int N, int K, uint16_t* __restrict__ out, int Np, int Kp) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int kq = Kp >> 2;
  if (idx >= (long long)Np * kq) return;
  const int n = (int)(idx / kq), q = (int)(idx - (long long)n * kq);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < N) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (4 * q + i < K) v[i] = w[(size_t)n * K + 4 * q + i];
  }
  ps_store4(out + (size_t)n * (2 * Kp), 4 * q, v);
}
void launch_pack_weight(const float* w, int N, int K, uint16_t* out, int Np, int Kp, hipStream_t s) {
  const long long total = (long long)Np * (Kp >> 2);
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, N, K, out, Np, Kp);
}


// LayerNorm folded into the Linear that follows it (gemm_epi.h, EpiGeluT / EpiQKVT with FOLD): packs gamma o W as the packed-split
// weight and emits, per output row n, csum[n] = sum_k (hi + lo of the PACKED element) -- what the MFMAs will actually multiply the
// row mean with -- and bias2[n] = bias[n] + sum_k beta[k] W[n][k]; both sums in fp64.  One wave per output row.
__global__ __launch_bounds__(256) void pack_weight_fold_kernel(const float* __restrict__ w, int N, int K, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ bias,
                                                               uint16_t* __restrict__ out, int Np, int Kp, float* __restrict__ csum,
                                                               float* __restrict__ bias2) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= Np) return;
  double cq = 0.0, bb = 0.0;
  for (int q = lane; q < (Kp >> 2); q += 64) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 4 * q + i;
        if (k < K) {
          const float wv = w[(size_t)n * K + k];
          v[i] = gamma[k] * wv;
          bb += (double)beta[k] * (double)wv;
        }
      }
    }
    uint2 hi, lo;
    split4(v, hi, lo);
    uint16_t* p = out + (size_t)n * (2 * Kp) + ps_off(4 * q);
    *reinterpret_cast<uint2*>(p) = hi;
    *reinterpret_cast<uint2*>(p + 8) = lo;
    const f32x2 h01 = unpack_f16(hi.x), h23 = unpack_f16(hi.y), l01 = unpack_f16(lo.x), l23 = unpack_f16(lo.y);
    cq += ((double)h01[0] + (double)l01[0]) + ((double)h01[1] + (double)l01[1]) + ((double)h23[0] + (double)l23[0]) + ((double)h23[1] + (double)l23[1]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { cq += __shfl_xor(cq, o, 64); bb += __shfl_xor(bb, o, 64); }
  if (lane == 0 && n < N) { csum[n] = (float)cq; bias2[n] = (float)((double)bias[n] + bb); }
}
void launch_pack_weight_fold(const float* w, int N, int K, const float* gamma, const float* beta, const float* bias, uint16_t* out, int Np,
                             int Kp, float* csum, float* bias2, hipStream_t s) {
  hipLaunchKernelGGL(pack_weight_fold_kernel, dim3((Np + 3) / 4), dim3(256), 0, s, w, N, K, gamma, beta, bias, out, Np, Kp, csum, bias2);
}

}  // namespace ribca
