// bf16x3 MFMA GEMM for the ViT linears (timm Attention.qkv / Attention.proj / Mlp.fc1 / Mlp.fc2 / PatchEmbed.proj,
// reached from reference cell_type_annotation/model.py:402 ``model(x)``).
//
//   C[m][n] = sum_k A[m][k] * W[n][k]        A: activations  [M ][2*Kp] packed-split bf16 (ribca_common.h)
//                                            W: nn.Linear wt [Np][2*Kp] packed-split bf16, Np = N padded to the tile
//
// Shape regime: M = cells*101 is huge (1e4..1e6), N in {144..2304}, K in {64..2304}: short K loops, so the tile is
// 128 x {64,96,128} with BK = 32 and two blocks per CU covering each other's prologue/epilogue.
//
// * 256 threads = 4 waves as 2(M) x 2(N); each wave owns 64 x BN/2 outputs = 4 x TN tiles of 16x16.
// * Tiles are computed TRANSPOSED: acc = mfma(Wfrag, Afrag) so a lane holds 4 consecutive output columns n of one row m
//   (C/D map: col = lane&15 -> m, row = 4*(lane>>4)+r -> n).  Epilogues then issue one 8/16-byte store per tile
//   instead of four 2-byte ones (residual RMW on fp32 z is one float4).
// * Each (Afrag, Wfrag) pair feeds three MFMAs (hi*hi, lo*hi, hi*lo): LDS bytes per MFMA are 2/3 of a plain bf16 GEMM.
// * LDS tile = rows of 128 B (one 32-deep K step of a PS row: 4 x [16 B hi | 16 B lo]).  16-byte chunk c of row r lives at
//   chunk c ^ f(r),  f(r) = ((r>>1)&7) ^ (4 <= (r&15) < 12 ? 2 : 0):  every hardware ds_read_b128 lane group
//   ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) then touches 16 distinct 16-byte slots of the 256-byte bank row.
// * global -> register -> LDS staging, double buffered: loads of K-step k+1 are issued before the MFMAs of step k and
//   written to the other LDS stage after them; one barrier per K step.
// * block id -> tile map is XCD-aware (blocks b, b+8, ... share an L2): every XCD walks whole rows of n-tiles of one
//   m-tile, so an A tile is fetched into one L2 once and reused by all its n-tiles.
#include "ribca_common.h"
#include "ribca_kernels.h"

namespace ribca {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int ROWB = 128;  // bytes per LDS tile row

__device__ __forceinline__ int lds_off(int row, int chunk) {
  const int f = ((row >> 1) & 7) ^ ((((row + 12) & 15) < 8) ? 2 : 0);
  return row * ROWB + ((chunk ^ f) << 4);
}

// ---------------------------------------------------------------------------------------------- epilogues
struct EpiResid {
  float* z; int ldz; const float* bias; int M, N;
  __device__ __forceinline__ void operator()(int m, int n, const f32x4& v) const {
    if (m >= M || n >= N) return;
    float4* p = reinterpret_cast<float4*>(z + (size_t)m * ldz + n);
    const float4 b = *reinterpret_cast<const float4*>(bias + n);
    float4 o = *p;
    o.x += v[0] + b.x; o.y += v[1] + b.y; o.z += v[2] + b.z; o.w += v[3] + b.w;
    *p = o;
  }
};

struct EpiGelu {
  uint16_t* out; int ldo; const float* bias; int M, N;
  __device__ __forceinline__ void operator()(int m, int n, const f32x4& v) const {
    if (m >= M || n >= N) return;
    const float4 b = *reinterpret_cast<const float4*>(bias + n);
    float t[4] = {gelu_erf(v[0] + b.x), gelu_erf(v[1] + b.y), gelu_erf(v[2] + b.z), gelu_erf(v[3] + b.w)};
    ps_store4(out + (size_t)m * ldo, n, t);
  }
};

struct EpiEmbed {
  float* z; int ldz; const float* bias; const float* pos; int D; int M, N;
  __device__ __forceinline__ void operator()(int m, int n, const f32x4& v) const {
    if (m >= M || n >= N) return;
    const int cell = m / 100, t = m - cell * 100;
    const float4 b = *reinterpret_cast<const float4*>(bias + n);
    const float4 pe = *reinterpret_cast<const float4*>(pos + (size_t)(1 + t) * D + n);
    float4 o;
    o.x = v[0] + b.x + pe.x; o.y = v[1] + b.y + pe.y; o.z = v[2] + b.z + pe.z; o.w = v[3] + b.w + pe.w;
    *reinterpret_cast<float4*>(z + ((size_t)cell * kTokens + 1 + t) * ldz + n) = o;
  }
};

struct EpiQKV {
  uint16_t* q; uint16_t* k; uint16_t* vt; const float* bias; int D, hd, hdp, hdv; float scale; int M, N;
  __device__ __forceinline__ void operator()(int m, int n, const f32x4& v) const {
    if (m >= M || n >= N) return;
    const int which = n / D;
    const int f = n - which * D;
    const int head = f / hd;
    const int d = f - head * hd;  // multiple of 4, d+3 < hd (hd % 4 == 0)
    const int cell = m / kTokens, t = m - cell * kTokens;
    const float4 b = *reinterpret_cast<const float4*>(bias + n);
    float x[4] = {v[0] + b.x, v[1] + b.y, v[2] + b.z, v[3] + b.w};
    const size_t ch = (size_t)cell * kHeads + head;
    if (which < 2) {
      if (which == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] *= scale;
      }
      uint16_t* row = (which == 0 ? q : k) + (ch * kTokPad + t) * (size_t)(2 * hdp);
      ps_store4(row, d, x);
    } else {
      // V^T[d][key], key order permuted inside each 32-key block so that the 8 keys a lane group owns after the
      // K*Q^T MFMA (two 16-key tiles, rows 4g..4g+3 of each) are contiguous: key = 32s+16u+4g+r -> 32s+8g+4u+r
      const int pos = (t & ~31) | (((t >> 2) & 3) << 3) | (((t >> 4) & 1) << 2) | (t & 3);
      uint16_t* base = vt + ch * (size_t)hdv * (2 * kKeyPad) + ps_off(pos);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint16_t hi, lo;
        split_bf16(x[i], hi, lo);
        uint16_t* p = base + (size_t)(d + i) * (2 * kKeyPad);
        p[0] = hi;
        p[8] = lo;
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------- kernel
template <int BN, class Epi>
__global__ __launch_bounds__(256, 2) void gemm_ps_kernel(const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ W, int ldw,
                                                         int M, int Kp, int mtiles, int ntiles, Epi epi) {
  constexpr int TN = BN / 32;
  constexpr int A_CH = BM * 8 / 256;
  constexpr int W_CH = BN * 8 / 256;
  constexpr int STAGE = (BM + BN) * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware bijective remap (cdna guide T1): blocks with equal (bid % 8) share an L2
  const int nblk = mtiles * ntiles;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, loc = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int mt = bid / ntiles, nt = bid - mt * ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, g = lane >> 4;

  // staging assignment: chunk q = tid + 256*i -> (row = q>>3, 16-byte chunk = q&7)
  const uint16_t* a_src[A_CH];
  int a_dst[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int qd = tid + 256 * i, row = qd >> 3, ch = qd & 7;
    int gm = m0 + row;
    gm = gm < M ? gm : M - 1;  // tail rows re-read the last valid row; their outputs are never stored
    a_src[i] = A + (size_t)gm * lda + ch * 8;
    a_dst[i] = lds_off(row, ch);
  }
  const uint16_t* w_src[W_CH];
  int w_dst[W_CH];
#pragma unroll
  for (int i = 0; i < W_CH; ++i) {
    const int qd = tid + 256 * i, row = qd >> 3, ch = qd & 7;
    w_src[i] = W + (size_t)(n0 + row) * ldw + ch * 8;
    w_dst[i] = BM * ROWB + lds_off(row, ch);
  }

  int a_rd[4], w_rd[TN];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_rd[i] = lds_off(wm * 64 + i * 16 + r16, 2 * g);
#pragma unroll
  for (int i = 0; i < TN; ++i) w_rd[i] = BM * ROWB + lds_off(wn * (BN / 2) + i * 16 + r16, 2 * g);
  // chunk 2g+1 (the lo half) differs from chunk 2g only in bit 4 of the swizzled offset
  f32x4 acc[4][TN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 ra[A_CH], rw[W_CH];
  const int nk = Kp / BK;

#pragma unroll
  for (int i = 0; i < A_CH; ++i) ra[i] = *reinterpret_cast<const uint4*>(a_src[i]);
#pragma unroll
  for (int i = 0; i < W_CH; ++i) rw[i] = *reinterpret_cast<const uint4*>(w_src[i]);
#pragma unroll
  for (int i = 0; i < A_CH; ++i) *reinterpret_cast<uint4*>(smem + a_dst[i]) = ra[i];
#pragma unroll
  for (int i = 0; i < W_CH; ++i) *reinterpret_cast<uint4*>(smem + w_dst[i]) = rw[i];
  __syncthreads();

  auto compute = [&](const char* st) {
    bf16x8 ahi[4], alo[4], whi[TN], wlo[TN];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ahi[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + a_rd[i]));
      alo[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + (a_rd[i] ^ 16)));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      whi[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + w_rd[j]));
      wlo[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + (w_rd[j] ^ 16)));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = mfma_bf16(wlo[j], ahi[i], acc[i][j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = mfma_bf16(whi[j], alo[i], acc[i][j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = mfma_bf16(whi[j], ahi[i], acc[i][j]);
  };

  // steady state: loads of step kk+1 in flight under the MFMAs of step kk, written to the other stage afterwards
  for (int kk = 0; kk + 1 < nk; ++kk) {
    const int ko = (kk + 1) * (2 * BK);  // bf16 elements per K step in a PS row
#pragma unroll
    for (int i = 0; i < A_CH; ++i) ra[i] = *reinterpret_cast<const uint4*>(a_src[i] + ko);
#pragma unroll
    for (int i = 0; i < W_CH; ++i) rw[i] = *reinterpret_cast<const uint4*>(w_src[i] + ko);
    compute(smem + (kk & 1) * STAGE);
    char* nx = smem + ((kk + 1) & 1) * STAGE;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) *reinterpret_cast<uint4*>(nx + a_dst[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < W_CH; ++i) *reinterpret_cast<uint4*>(nx + w_dst[i]) = rw[i];
    __syncthreads();
  }
  compute(smem + ((nk - 1) & 1) * STAGE);

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + r16;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 16 + 4 * g;
      epi(m, n, acc[i][j]);
    }
  }
}

// ---------------------------------------------------------------------------------------------- host side
int gemm_pick_bn(int N) {
  if (N % 128 == 0) return 128;
  if (N % 96 == 0) return 96;
  if (N % 64 == 0) return 64;
  return 96;
}
int gemm_padded_n(int N) {
  const int bn = gemm_pick_bn(N);
  return (N + bn - 1) / bn * bn;
}

template <int BN, class Epi>
static void launch_bn(const GemmArgs& g, const Epi& epi, hipStream_t s) {
  const int mtiles = (g.M + BM - 1) / BM;
  const int ntiles = gemm_padded_n(g.N) / BN;
  const size_t lds = 2 * (size_t)(BM + BN) * ROWB;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ps_kernel<BN, Epi>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_ps_kernel<BN, Epi>), dim3(mtiles * ntiles), dim3(256), lds, s, g.A, g.lda, g.W, g.ldw, g.M, g.Kp, mtiles,
                     ntiles, epi);
}

template <class Epi>
static void launch_any(const GemmArgs& g, const Epi& epi, hipStream_t s) {
  if (g.M <= 0) return;
  switch (gemm_pick_bn(g.N)) {
    case 128: launch_bn<128>(g, epi, s); break;
    case 64: launch_bn<64>(g, epi, s); break;
    default: launch_bn<96>(g, epi, s); break;
  }
}

void launch_gemm_resid(const GemmArgs& g, float* z, int ldz, hipStream_t s) {
  launch_any(g, EpiResid{z, ldz, g.bias, g.M, g.N}, s);
}
void launch_gemm_gelu(const GemmArgs& g, uint16_t* out, int ldo, hipStream_t s) {
  launch_any(g, EpiGelu{out, ldo, g.bias, g.M, g.N}, s);
}
void launch_gemm_embed(const GemmArgs& g, float* z, int ldz, const float* pos, int D, hipStream_t s) {
  launch_any(g, EpiEmbed{z, ldz, g.bias, pos, D, g.M, g.N}, s);
}
void launch_gemm_qkv(const GemmArgs& g, uint16_t* q, uint16_t* k, uint16_t* vt, int D, int hd, int hdp, int hdv, float scale,
                     hipStream_t s) {
  launch_any(g, EpiQKV{q, k, vt, g.bias, D, hd, hdp, hdv, scale, g.M, g.N}, s);
}

}  // namespace ribca
