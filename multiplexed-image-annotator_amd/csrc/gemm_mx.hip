// "MX" form of the split-operand GEMM: fp16 hi * hi on v_mfma_f32_16x16x32_f16 plus TWO block-scaled correction products on
// v_mfma_scale_f32_16x16x128_f8f6f4 -- 1.75 matrix units per product instead of the three fp16 passes of gemm_split16.hip /
// gemm_duo.hip, and 3 bytes per activation element instead of 4 (timm Mlp.fc2 of every full Block reached from reference
// cell_type_annotation/model.py:54-55 ``for blk in self.blocks: x = blk(x)``; tolerance: north_star's 1e-3 on
// softmax(model(x), dim=1), model.py:401-404).
//
//   x = hi + lo,  hi = fp16(x),  lo = x - hi                      (as everywhere on this path: ribca_common.h)
//   A W^T  ~=  Ah Wh^T  +  Al' Wh'^T  +  Ah' Wl'^T                Ah Wh^T : 4 x v_mfma_f32_16x16x32_f16 per 128 k   (64 cycles)
//                                                                 Al' Wh'^T: A lo as fp8 e4m3 x W hi as fp6 e2m3    (32 cycles)
//                                                                 Ah' Wl'^T: A hi as fp6 e2m3 x W lo as fp6 e2m3    (16 cycles)
//   every primed operand block-scaled (E8M0 per 32 k).  The corrections are 2^-11 of the product and need 4 bits each: the scheme's
//   error is 2^-16 class (tools/mx_mix_probe.hip: 7.8e-6 of max |C| on hardware, the roundings emulated bit for bit;
//   tests/precision_study.py: 4-9e-5 on softmax outputs after 12 blocks).  What it buys (profiles/r4/mx_mix_probe.txt): one wave's 128-deep
//   step over 8 x 3 tiles takes 1.5 us instead of 2.5 us of matrix time at two waves per SIMD (the chip is power-managed: fewer matrix
//   passes also run at a higher clock), and the A stream through L2 -> LDS shrinks by a quarter.
//
// Operand formats ("MX3" activations; written by mx_pack_act_kernel here and by the GELU epilogue of gemm_duo.hip):
//   * 32 consecutive columns form a block; E = max(exponent field of the block's largest |hi|, 1) - 15;
//     the block's scale byte is  sl = E - 19 + 127  (lo / 2^(E-19) <= 256 < 448, the e4m3 maximum: v_cvt_scalef32_pk_fp8_f32 does NOT
//     saturate, it returns the NaN code); the fp6 image of hi uses sh = sl + 17 (hi / 2^(E-2) < 8, saturating at 7.5);
//   * hi plane  [M][Kp] fp16, PERMUTED inside every 128 columns: column c = 32 g + 8 s + j sits at position 32 s + 8 g + j, so that the
//     f16 MFMA of sub-step s reads one contiguous 64-byte piece per row AND lane (row, g) holds 32 CONSECUTIVE columns over the four
//     sub-steps -- exactly the block the scaled MFMA wants in one lane, so the fp6 image of hi is ONE v_cvt_scalef32_pk32_fp6_f16 of
//     registers the kernel has anyway (never stored);
//   * lo plane  [M][Kp] bytes (e4m3, natural column order: a 128-deep step of a row is one 128-byte line);
//   * scale plane [Kp / 128][M][4] bytes (transposed: a K step's 128 rows x 4 bytes are contiguous).
//   Weights (mx_pack_w_kernel, once per model): fragment order as in gemm_duo.hip, grouped by the 48 output columns (3 column tiles) one
//   wave owns -- per group and 128 k: twelve 1 KB hi fragments [sub-step][tile], then the fp6 images of lo and hi (24 bytes per lane and
//   tile, as 8 + 16) and 8 bytes of scale bytes per lane -- so that ONE scalar base per stream reaches a step's operands.
//
// Kernel: two 256-thread workgroups per CU (77 KB of LDS each), tile 128 x 192, 4 waves as 1 x 4 (a wave owns all 128 rows x 48 columns
// = 8 x 3 accumulator tiles), W straight from L2 into registers, A through LDS by buffer_load ... lds.  One 128-deep step:
//     [B1: the step's four hi units have landed]  f16 phase: 4 sub-steps x 8 row tiles x 3 MFMAs, each sub-step's W hi fragments in a register set of their own
//     that is refilled for the NEXT step right behind its last use (a whole MX phase ahead: with one MFMA per tile a sub-step lasts a third
//     of an fp16x3 K step, two of them are shorter than an L2 round trip)  ->  conversion: every wave turns the hi rows of TWO row tiles into fp6 (2 v_cvt_scalef32_pk32_fp6_f16 per wave instead of 8)
//     and leaves them in LDS  ->  [B2]  the next step's hi units are requested into the slots just freed  ->  MX phase: 8 row tiles x
//     (lo, fp6 hi, scale from LDS) x 3 x 2 scaled MFMAs.
// Two barriers per 128 k (the fp16x3 kernels: four).  The residual tile rides the ring behind the product's own steps exactly as in
// gemm_duo.hip (EpiResidZK: identity fragments, exact), and the epilogue is that kernel's.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "gemm_epi.h"

#ifdef MX_BIG      // (see the kernel section: 256-row tiles, accumulators in AGPRs)
#define MX_ACC_CONSTRAINT "+a"
#else
#define MX_ACC_CONSTRAINT "+v"
#endif

namespace ribca {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef unsigned u32v4 __attribute__((ext_vector_type(4)));
typedef unsigned u32v2 __attribute__((ext_vector_type(2)));
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

namespace {

template <int... Is, class F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(std::make_integer_sequence<int, N>{}, f);
}

template <int OFF>
__device__ __forceinline__ void lds_rd128(u32v4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_rd128h(f16x8& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_rd64(u32v2& dst, unsigned addr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_rd8(unsigned& dst, unsigned addr) {
  asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_wr128(unsigned addr, const u32v4& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_wr64(unsigned addr, const u32v2& v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ unsigned long long uniform_ptr(const void* p) {
  const unsigned long long b = (unsigned long long)(uintptr_t)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
template <int OFF>
__device__ __forceinline__ void gld16h(f16x8& dst, unsigned voff, unsigned long long sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void gld16(u32v4& dst, unsigned voff, unsigned long long sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void gld8(u32v2& dst, unsigned voff, unsigned long long sbase) {
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void gld4(unsigned& dst, unsigned voff, unsigned long long sbase) {
  asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// fp6 operands occupy 6 dwords of the 8 the builtin takes: the upper two are never read by the instruction
__device__ __forceinline__ i32x8 op6(const u32v4& a, const u32v2& b) {
  const u32v4 bx = __builtin_shufflevector(b, b, 0, 1, -1, -1);
  return __builtin_bit_cast(i32x8, __builtin_shufflevector(a, bx, 0, 1, 2, 3, 4, 5, -1, -1));
}
__device__ __forceinline__ i32x8 op8(const u32v4& a, const u32v4& b) {
  return __builtin_bit_cast(i32x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// Pins every MFMA issued so far in front of this point (no instruction): the products are pure values to the compiler, which otherwise
// sinks them below the conditional operand requests that follow a phase -- keeping COPIES of the fragments those requests overwrite
// (measured: 400 registers instead of 230).
template <int MT, int TN>
__device__ __forceinline__ void pin_acc(f32x4 (&acc)[1][MT][TN]) {
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : MX_ACC_CONSTRAINT(acc[0][i][j]));
}

}  // namespace

// ----------------------------------------------------------------------------------------------------------- operand packers
// bytes of the weight image per (48 output columns, 128 k): WH 12 x 1 KB of fp16 hi fragments [sub-step][tile];
// WX (behind a 512-byte header per 48 columns): scale bytes 64 x 8 (sl0 sh0' sl1 sh1' | sl2 sh2' 0 0) | l6b [tile][64 x 8] | l6a [tile][64 x 16]
// -- the fp6 image of W lo with its scale bytes sl, and the scale bytes sh' of the NEXT 128 k's hi image (the header holds those of the first):
// a step's hi scale is needed at the top of the step, one MX phase before the words of its own block are waited for.  The fp6 image of W HI is not stored: the kernel holds a step's four hi fragments of a tile in registers anyway and
// converts them itself (3 v_cvt_scalef32_pk32_fp6_f16 per wave and step instead of 4.6 KB of loads: the K loop is bound by the CU's
// fetch path, profiles/r4/mx_kernel_ablations.txt).  Every piece lies within the 13-bit signed instruction offset of ONE scalar base
// (block start + 1024).
constexpr int kMxWhBytes = 12288, kMxWxBytes = 5120, kMxWxHeader = 512;
constexpr int kWxBase = 1024, kWxSc = 0, kWxL6b = 512, kWxL6a = 2048;
// (sized for whole 192-column tiles: the waves of a last tile that lie beyond N still stream their -- zero -- fragments)
size_t mx_wh_bytes(int Np, int Kp) { return (size_t)((Np + 191) / 192 * 4) * (Kp / 128) * kMxWhBytes; }
size_t mx_wx_bytes(int Np, int Kp) { return (size_t)((Np + 191) / 192 * 4) * ((size_t)(Kp / 128) * kMxWxBytes + kMxWxHeader); }

// packed-split weight [Np][>= 2 Ksrc] (rows n >= N read as zeros by the caller's padding; Kp = Ksrc rounded up to 128: zeros) -> WH / WX; one thread per
// (output column, 32-k block); N48 = columns rounded up to whole 192-column tiles: columns beyond Np are written as zeros
__global__ void mx_pack_w_kernel(const uint16_t* __restrict__ W, int ldw, int Np, int N48, int Ksrc, int Kp, uint16_t* __restrict__ WH,
                                 unsigned char* __restrict__ WX) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nq = Kp / 32;
  if (idx >= (long long)N48 * nq) return;
  const int n = (int)(idx / nq), q = (int)(idx - (long long)n * nq);
  const int jb = n / 48, j = (n - 48 * jb) >> 4, r16 = n & 15, b = q >> 2, g = q & 3, lane = g * 16 + r16, nb = Kp / 128;
  h32 lv;
  unsigned mh = 0, ml = 0;
  uint4 hraw[4] = {};
  if (n < Np && 32 * q < Ksrc) {
    const uint16_t* src = W + (size_t)n * ldw + (size_t)q * 64;      // 4 PS groups of [hi 8 | lo 8]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      hraw[t] = *reinterpret_cast<const uint4*>(src + 16 * t);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint16_t hb = src[16 * t + e], lb = src[16 * t + 8 + e];
        lv[8 * t + e] = __builtin_bit_cast(_Float16, lb);
        mh = max(mh, (unsigned)(hb & 0x7fff)); ml = max(ml, (unsigned)(lb & 0x7fff));
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < 32; ++e) lv[e] = (_Float16)0.f;
  }
  // both images scaled so that the block's largest magnitude lands in [4, 8): the weights are packed once, so lo gets its own exponent
  int eh = (int)(mh >> 10), el = (int)(ml >> 10);
  eh = eh < 1 ? 1 : eh; el = el < 1 ? 1 : el;
  const int sh = eh + 110, sl = el + 110;      // E - 2 + 127, E = ef - 15
  const u32x6 l6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(lv, e8m0_float(sl));
  const size_t blk = (size_t)jb * nb + b;
  unsigned char* wh = reinterpret_cast<unsigned char*>(WH) + blk * kMxWhBytes + j * 1024 + lane * 16;
#pragma unroll
  for (int t = 0; t < 4; ++t) *reinterpret_cast<uint4*>(wh + t * 3072) = hraw[t];
  unsigned char* wxj = WX + (size_t)jb * ((size_t)nb * kMxWxBytes + kMxWxHeader);      // this 48-column group: header, then nb blocks
  unsigned char* wx = wxj + kMxWxHeader + (size_t)b * kMxWxBytes;
  *reinterpret_cast<uint2*>(wx + kWxL6b + j * 512 + lane * 8) = uint2{l6[4], l6[5]};
  *reinterpret_cast<uint4*>(wx + kWxL6a + j * 1024 + lane * 16) = uint4{l6[0], l6[1], l6[2], l6[3]};
  wx[kWxSc + lane * 8 + j * 2] = (unsigned char)sl;
  // the hi image's scale byte goes into the PREVIOUS block's word (the header for the first block)
  unsigned char* prev = b == 0 ? wxj : wxj + kMxWxHeader + (size_t)(b - 1) * kMxWxBytes + kWxSc;
  prev[lane * 8 + j * 2 + 1] = (unsigned char)sh;
  if (b == nb - 1) wx[kWxSc + lane * 8 + j * 2 + 1] = 0;
  if (b == 0) wxj[lane * 8 + j * 2] = 0;
  if (j == 2) {
    *reinterpret_cast<uint16_t*>(wx + kWxSc + lane * 8 + 6) = 0;
    if (b == 0) *reinterpret_cast<uint16_t*>(wxj + lane * 8 + 6) = 0;
  }
}
void launch_mx_pack_w(const uint16_t* W, int ldw, int Np, int Ksrc, int Kp, uint16_t* WH, unsigned char* WX, hipStream_t s) {
  const int N48 = (Np + 191) / 192 * 192;
  const long long total = (long long)N48 * (Kp / 32);
  hipLaunchKernelGGL(mx_pack_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, ldw, Np, N48, Ksrc, Kp, WH, WX);
}

// packed-split activation rows [M][2 Kp] -> the three MX3 planes (Kp128 = Kp rounded up to 128: the pad reads as zeros).  The reference
// producer of the format: tests compare the fused producers with it, and operands no kernel emits in MX3 yet go through it.
__global__ void mx_pack_act_kernel(const uint16_t* __restrict__ ps, int ldps, int M, int Kp, int Kp128, uint16_t* __restrict__ hi,
                                   unsigned char* __restrict__ l8, unsigned char* __restrict__ sc) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nq = Kp128 / 32;
  if (idx >= (long long)M * nq) return;
  const int m = (int)(idx / nq), q = (int)(idx - (long long)m * nq);
  uint4 hv[4] = {}, lv[4] = {};
  if (32 * q < Kp) {
    const uint4* src = reinterpret_cast<const uint4*>(ps + (size_t)m * ldps + (size_t)q * 64);
#pragma unroll
    for (int t = 0; t < 4; ++t) { hv[t] = src[2 * t]; lv[t] = src[2 * t + 1]; }
  }
  unsigned mh = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const unsigned w[4] = {hv[t].x, hv[t].y, hv[t].z, hv[t].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) mh = max(mh, max(w[e] & 0x7fffu, (w[e] >> 16) & 0x7fffu));
  }
  const int sl = mx_sl_byte((int)(mh >> 10));
  const float scale = e8m0_float(sl);
  uint16_t* hrow = hi + (size_t)m * Kp128 + (q >> 2) * 128 + (q & 3) * 8;      // + 32 t: mx_hi_pos of column 32 q + 8 t
  unsigned out8[8];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    *reinterpret_cast<uint4*>(hrow + 32 * t) = hv[t];
    const unsigned w[4] = {lv[t].x, lv[t].y, lv[t].z, lv[t].w};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const f32x2 a = unpack_f16(w[2 * e]), b = unpack_f16(w[2 * e + 1]);
      s16x2 r = {0, 0};
      r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, a[0], a[1], scale, false);
      r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, b[0], b[1], scale, true);
      out8[2 * t + e] = __builtin_bit_cast(unsigned, r);
    }
  }
  uint4* lrow = reinterpret_cast<uint4*>(l8 + (size_t)m * Kp128 + 32 * q);
  lrow[0] = uint4{out8[0], out8[1], out8[2], out8[3]};
  lrow[1] = uint4{out8[4], out8[5], out8[6], out8[7]};
  sc[((size_t)(q >> 2) * M + m) * 4 + (q & 3)] = (unsigned char)sl;
}
void launch_mx_pack_act(const uint16_t* ps, int ldps, int M, int Kp, const MxAct& a, hipStream_t s) {
  const long long total = (long long)M * (a.Kp / 32);
  hipLaunchKernelGGL(mx_pack_act_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ps, ldps, M, Kp, a.Kp, a.hi, a.l8, a.sc);
}

// ----------------------------------------------------------------------------------------------------------- kernel
#ifndef MXDBG_PF
#define MXDBG_PF 3
#endif
namespace {
// MX_BIG (experiment of round 5, tools/build_ab_lib.py ... -DMX_BIG; NOT the product): 256-row tiles, ONE 4-wave workgroup per CU, a wave owns 256 rows x
// 48 columns (16 x 3 accumulator tiles in AGPRs) -- every W fragment then serves twice the rows: 0.027 operand bytes per multiply-add instead of
// 0.038.  No MX3 emission in this form (the staging image is that of a 128-row tile).
#ifdef MX_BIG
constexpr int MX_BM = 256, MX_MT = 16;
#else
constexpr int MX_BM = 128, MX_MT = 8;
#endif
constexpr int MX_BN = 192, MX_TN = 3;
// LDS map: hi as two halves of BM rows x 128 bytes (sub-steps 0-1 | 2-3), the fp6 rows, two scale slots, two lo slots
constexpr int L_HALF = MX_BM * 128;
constexpr int L_HI = 0, L_H6A = 2 * L_HALF, L_H6B = L_H6A + MX_BM * 64, L_SC = L_H6B + MX_BM * 32, L_SC_SLOT = MX_BM * 4 + 128, L_L8 = L_SC + 2 * L_SC_SLOT,
              L_XCH = L_L8 + 2 * L_HALF;
static_assert(MX_BM != 128 || (L_H6A == 32768 && L_H6B == 40960 && L_SC == 45056 && L_SC_SLOT == 640), "the 128-row map is the one the tests pinned");
// (L_XCH: 2 KB outside every ring slot for the MX3-emitting epilogues, gemm_epi.h mx3_emit_wave48; their staging image overlays the ring)
constexpr int L_TOTAL = L_XCH + 2048;
static_assert(L_TOTAL <= 160 * 1024, "LDS of a CU");
static_assert(MX_BM != 128 || kMx3StageBytes <= L_XCH, "the staging image of the MX3 emission must not reach the exchange space");

static_assert(L_L8 % 16 == 0, "LDS-DMA destination alignment");
__device__ __forceinline__ int mx_f4(int row) { return (4 - ((row >> 2) & 3)) & 3; }      // chunk swizzle of the 64-byte hi rows
// chunk swizzle of the 128-byte lo rows: an fp8 operand's lane (row, g) holds k = 16 g .. + 15 and 64 + 16 g .. + 15 (tools/mx_kmap_probe.hip:
// NOT the 32 consecutive k of the fp6 formats; the block scales follow the logical k), i.e. chunks g and g + 4 of the row -- with this
// swizzle every ds_read_b128 lane group touches 16 distinct 16-byte slots for both
__device__ __forceinline__ int mx_f8(int row) { return (row >> 1) & 7; }
}  // namespace

// ABL (diagnostic library only): 1 = no epilogue (results dropped)
// HALF (round 6; the launcher sets it where N is not a multiple of the 192-column tile: mlp.fc2 at D = 288, whose second column tile holds 96
// columns): the waves of a tile that lie wholly beyond N -- their accumulators are never stored -- skip their W requests, their fragment
// reads and their MFMAs, and keep what the workgroup needs from them: their share of the A requests, the hi -> fp6 conversion of their two row
// tiles, the residual units' requests and every barrier.  A separate instantiation: with HALF = false `dead` is a compile-time false and the
// code of every other launch is the round-5 kernel's, instruction for instruction.
template <class Epi, int ABL, bool HALF = false>
#ifdef MX_BIG
__global__ __launch_bounds__(256, 1) void gemm_mx_duo_kernel(
#else
__global__ __launch_bounds__(256, 2) void gemm_mx_duo_kernel(
#endif
MxAct A, const uint16_t* __restrict__ WH, const unsigned char* __restrict__ WX, int M, int nb,
                                                              int mtiles, int ntiles, Epi epi, int panel) {
  // epilogues: EpiResidZK (proj / fc2: the residual tile through the ring, optionally a second copy of the new rows in MX3), EpiQKVLn,
  // EpiGeluMx (fc1: GELU output in MX3)
  constexpr bool ZK = is_zk<Epi>::value;
  constexpr int MT = MX_MT, TN = MX_TN, BM = MX_BM, BN = MX_BN, ZS = BN / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nblk = mtiles * ntiles;
  int bid = blockIdx.x;
  {
    // blocks b, b + 8, ... run on one XCD (one L2): XCD x walks a contiguous tile range with n fastest (as gemm_duo.hip).  panel > 0
    // (n-tiles per panel, chosen by the launcher where the weight image is larger than an XCD's L2 can keep beside the A and output
    // streams): the whole m-tile rows of the range are walked panel by panel -- every m-tile's n-tiles of panel 0, then of panel 1, ... --
    // so a panel of W stays resident while the A rows stream past it (A is then fetched once per panel); the partial rows at the two
    // ends of the range keep the plain order.
    const int xcd = bid & 7, loc = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int cnt = xcd < r ? q + 1 : q;
    bid = first + loc;
    if (panel > 0 && panel < ntiles) {
      const int r0 = (first + ntiles - 1) / ntiles, r1 = (first + cnt) / ntiles;     // whole m-tile rows [r0, r1)
      const int head = r0 * ntiles - first;
      if (r1 > r0 && loc >= head) {
        int l = loc - head;
        const int rows = r1 - r0;
        if (l < rows * ntiles) {
          int p0 = 0, w = panel;
          while (l >= rows * w) {          // at most ntiles / panel iterations, uniform over the workgroup
            l -= rows * w;
            p0 += w;
            w = ntiles - p0 < panel ? ntiles - p0 : panel;
          }
          const int rr = l / w;
          bid = (r0 + rr) * ntiles + p0 + (l - rr * w);
        }
      }
    }
  }
  const int mt = bid / ntiles, nt = bid - mt * ntiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int rows_here = (M - m0) < BM ? (M - m0) : BM;
  const int Kp = A.Kp;

  // ---- descriptors over the tile's rows of the three planes (rows beyond M read as zeros) and of the residual stream
  const __amdgpu_buffer_rsrc_t hi_rsrc = __builtin_amdgcn_make_buffer_rsrc(A.hi + (size_t)m0 * Kp, 0, rows_here * Kp * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t l8_rsrc = __builtin_amdgcn_make_buffer_rsrc(A.l8 + (size_t)m0 * Kp, 0, rows_here * Kp, 0x00020000);
  // (plane b of the transposed scale bytes starts b * M * 4 bytes further: rows beyond the tile read the next rows' bytes or, at the very end of the
  // buffer, zeros -- they belong to rows that are never stored)
  const __amdgpu_buffer_rsrc_t sc_rsrc = __builtin_amdgcn_make_buffer_rsrc(A.sc + (size_t)m0 * 4, 0, ((nb - 1) * A.M + (A.M - m0)) * 4, 0x00020000);
  // (a copy: naming epi inside a lambda that a generic lambda calls makes hipcc drop the kernel's HOST stub without a diagnostic -- the library
  // then fails to load with an undefined symbol)
  int ldz_ = 0;
  if constexpr (ZK) ldz_ = epi.ldz;
  int hi_voff, l8_voff, sc_voff;
  {
    const int hrow = wave * 8 + (lane >> 3);                       // + 32 i for the wave's other groups (same swizzle): 8 rows x 128 bytes per piece
    hi_voff = hrow * Kp * 2 + (((lane & 7) ^ mx_f8(hrow)) << 4);
    const int drow = wave * 8 + (lane >> 3);                       // + 32 i for the wave's other groups (same swizzle)
    l8_voff = drow * Kp + (((lane & 7) ^ mx_f8(drow)) << 4);
    sc_voff = (wave * (BM / 4) + lane) * 4;
  }
  auto issue_hi = [&](int b) {      // the hi columns of step b as two halves of 64 columns (sub-steps 0-1, 2-3): 8 operations per wave
#ifdef MXDBG_NOA
    return;
#endif
    // whole 128-byte lines per row and piece: a sub-step's 64 bytes alone would request every line twice, half at a time
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < BM / 32; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(hi_rsrc, (__attribute__((address_space(3))) void*)(smem + L_HI + h * L_HALF + wave * 1024 + i * 4096), 16,
                                                 hi_voff, i * 32 * Kp * 2 + b * 256 + h * 128, 0, 0);
  };
  auto issue_l8 = [&](int b) {      // lo unit + scale unit of step b: 5 operations per wave
#ifdef MXDBG_NOA
    return;
#endif
    char* st = smem + L_L8 + (b & 1) * L_HALF + wave * 1024;
#ifdef MXDBG_LO4      // timing variant: half the lo bytes (what an fp4 image of A lo would move)
#pragma unroll
    for (int i = 0; i < BM / 64; ++i)
#else
#pragma unroll
    for (int i = 0; i < BM / 32; ++i)
#endif
      __builtin_amdgcn_raw_ptr_buffer_load_lds(l8_rsrc, (__attribute__((address_space(3))) void*)(st + i * 4096), 16, l8_voff, i * 32 * Kp + b * 128, 0, 0);
    // 64 rows per wave from row 32 w: the upper half repeats what the next wave writes (the same bytes) and the last wave's spills into the slot's pad
    __builtin_amdgcn_raw_ptr_buffer_load_lds(sc_rsrc, (__attribute__((address_space(3))) void*)(smem + L_SC + (b & 1) * L_SC_SLOT + wave * BM), 4, sc_voff,
                                             b * A.M * 4, 0, 0);
  };
  auto issue_z = [&](int t, int slot_off) {      // 32 columns of the residual tile (packed-split: 128 bytes per row): 4 operations per wave
    char* st = smem + slot_off + wave * 1024;
    const int ko = n0 * 4 + t * 128;
    const __amdgpu_buffer_rsrc_t z_rsrc = zk_rsrc(epi, m0, rows_here, hi_rsrc);
    const int drow = wave * 8 + (lane >> 3);
    const int z_voff = drow * ldz_ * 2 + (((lane & 7) ^ swz_f(drow)) << 4);      // (recomputed: not worth a register across the K loop)
#pragma unroll
    for (int i = 0; i < BM / 32; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(z_rsrc, (__attribute__((address_space(3))) void*)(st + i * 4096), 16, z_voff, i * 32 * ldz_ * 2 + ko, 0, 2);
  };

  // ---- W: the wave's three column tiles; hi fragments stream through three register sets, the fp6 images are single-buffered
  const int wn = wave;
  // (MX_DEAD folds to a literal false in the front end where HALF is false: those instantiations never see the branch)
  bool dead_rt = false;
  if constexpr (HALF && ZK) dead_rt = __builtin_amdgcn_readfirstlane((int)(n0 + wn * (16 * MX_TN) >= epi.N)) != 0;
#define MX_DEAD (HALF && ZK && dead_rt)
  const unsigned wvoff16 = (unsigned)lane * 16u, wvoff8 = (unsigned)lane * 8u;
  const size_t jb = (size_t)(n0 / 48) + (size_t)wn;
  const char* whb = reinterpret_cast<const char*>(WH) + jb * (size_t)nb * kMxWhBytes;
  const char* wxh = reinterpret_cast<const char*>(WX) + jb * ((size_t)nb * kMxWxBytes + kMxWxHeader);      // header: hi scale bytes of step 0
  const char* wxb = wxh + kMxWxHeader;
  f16x8 whi[4][TN];      // one set per sub-step, refilled for the NEXT 128-deep step right behind its last use
  u32v4 wl6a[TN];
  u32v2 wl6b[TN];
  u32x6 wh6r[TN];      // fp6 images of W hi, converted from the four resident hi sets at the top of every step
  u32v2 wsc;      // bytes: sl0 sh0' sl1 sh1' | sl2 sh2' - -   (sl: this step's lo image, sh': the NEXT step's hi image)
  u32v2 wsh;      // the hi scale bytes of the current step (taken out of wsc before the next block's words are requested into it)
  auto issue_whi = [&](int t, f16x8 (&dst)[TN]) {      // sub-step t = 4 b + s of the K loop: 3 operations
#ifdef MXDBG_NOW
    return;
#endif
    const unsigned long long p = uniform_ptr(whb + (size_t)t * 3072);
    gld16h<0>(dst[0], wvoff16, p);
    gld16h<1024>(dst[1], wvoff16, p);
    gld16h<2048>(dst[2], wvoff16, p);
  };
  auto issue_wx = [&](int b) {                          // 7 operations
#ifdef MXDBG_NOW
    return;
#endif
    const unsigned long long p = uniform_ptr(wxb + (size_t)b * kMxWxBytes + kWxBase);
    gld8<kWxL6b - kWxBase>(wl6b[0], wvoff8, p); gld8<kWxL6b - kWxBase + 512>(wl6b[1], wvoff8, p); gld8<kWxL6b - kWxBase + 1024>(wl6b[2], wvoff8, p);
    gld16<kWxL6a - kWxBase>(wl6a[0], wvoff16, p); gld16<kWxL6a - kWxBase + 1024>(wl6a[1], wvoff16, p); gld16<kWxL6a - kWxBase + 2048>(wl6a[2], wvoff16, p);
    gld8<kWxSc - kWxBase>(wsc, wvoff8, p);
  };
  constexpr int NWX = 7;

  f32x4 acc[1][MT][TN];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[0][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // per-lane read addresses (row tile i adds 16 rows)
  // hi rows are 128 bytes = two sub-steps: sub-step s reads chunk 4 (s & 1) + g of half s >> 1   (+ 2048 i + 16384 (s >> 1), second sub-step at ^ 64)
  const unsigned rd_hi = lds_base + L_HI + (unsigned)(r16 * 128 + ((g ^ mx_f8(r16)) << 4));
  const unsigned rd_h6a = lds_base + L_H6A + (unsigned)(r16 * 64 + ((g ^ mx_f4(r16)) << 4));               // + 1024 i
  const unsigned rd_l8 = lds_base + (unsigned)(r16 * 128 + ((g ^ mx_f8(r16)) << 4));      // 128-byte lo rows: + 2048 i + slot; chunk g + 4 at ^ 64
  const unsigned rd_h6b = lds_base + L_H6B + (unsigned)(r16 * 32 + ((g ^ (2 * ((r16 >> 3) & 1))) << 3));  // + 512 i
  const unsigned rd_sc = lds_base + L_SC + (unsigned)(r16 * 4 + g);                                        // + 64 i + slot

  // ---- prologue
  if (!MX_DEAD) {
#ifndef MXDBG_NOW
    gld8<0>(wsh, wvoff8, uniform_ptr(wxh));
#endif
    issue_whi(0, whi[0]);
    issue_whi(1, whi[1]);
    issue_whi(2, whi[2]);
    issue_whi(3, whi[3]);
  } else {      // (defined values in the registers the skipped phases would have filled: nothing reads them, the compiler still wants them initialised)
    wsh = u32v2{0u, 0u}; wsc = u32v2{0u, 0u};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int j = 0; j < TN; ++j) whi[s4][j] = f16x8{};
#pragma unroll
    for (int j = 0; j < TN; ++j) { wl6a[j] = u32v4{0u, 0u, 0u, 0u}; wl6b[j] = u32v2{0u, 0u}; }
  }
  issue_l8(0);
  issue_hi(0);
  if (!MX_DEAD) issue_wx(0);

#ifdef MXDBG_STAMP      // timing variant (tools/build_mx_variant.py): shader-clock cycles per phase, summed over the K loop, per wave
  unsigned long long st_prev = __builtin_amdgcn_s_memtime(), st_b1 = 0, st_f16 = 0, st_cv = 0, st_b2 = 0, st_mx = 0;
  const unsigned long long st_start = st_prev, rt_start = __builtin_amdgcn_s_memrealtime();      // (100 MHz: the in-kernel clock = d memtime / d memrealtime)
#define MX_STAMP(acc_) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_ += t_ - st_prev; st_prev = t_; }
#else
#define MX_STAMP(acc_)
#endif
  // one 128-deep step
  // (the last step is its own instance: "more" is a compile-time constant, so no phase is cut into basic blocks by the requests for the next step)
  auto step = [&](auto more_c, int b) {
    constexpr bool more = decltype(more_c)::value;
    // B1: hi units, lo / scale unit and the four W hi sets of this step have landed (only the 7 operations of W's fp6 lo image and scale bytes are younger)
    // (a dead wave has no W operation in flight: its youngest operations ARE the hi / lo units, so it waits for all of them)
    if (MX_DEAD) wait_vmcnt<0>();
    else wait_vmcnt<NWX>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    MX_STAMP(st_b1)
    if (!MX_DEAD) {
    // the fp6 image of this step's W hi, from the four resident fragment sets (they are refilled for the next step right behind their last use,
    // so now is the moment); scale byte = the packer's sh of the tile, which travelled with the PREVIOUS block's words (wsh)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(whi[s4][j]));
    asm volatile("" : "+v"(wsh));
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      h32 hv;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[8 * s4 + e] = whi[s4][j][e];
      const unsigned shb = (j == 0 ? (wsh[0] >> 8) : j == 1 ? (wsh[0] >> 24) : (wsh[1] >> 8)) & 0xffu;
      wh6r[j] = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hv, e8m0_float((int)shb));
    }
    sfor<4>([&](auto s_c) {
      constexpr int S = decltype(s_c)::value;
      __builtin_amdgcn_sched_barrier(0);
      // A hi fragments PF row tiles ahead of their MFMAs: a row tile is 3 MFMAs = 48 cycles here (the fp16x3 kernels: 9), less than one LDS
      // round trip -- with the next tile's read as the only one in flight every iteration waited for it (timing ablation without any
      // global load: 2.2 x the matrix time)
      constexpr int PF = MXDBG_PF;
      f16x8 ah[PF + 1];
      const unsigned rd_s = (S & 1) ? (rd_hi ^ 64u) : rd_hi;
      constexpr int HOFF = (S >> 1) * L_HALF;
      sfor<PF>([&](auto pc) {
        constexpr int q = decltype(pc)::value;
        lds_rd128h<HOFF + q * 2048>(ah[q], rd_s);
      });
      sfor<MT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int cur = i % (PF + 1);
        if constexpr (i + PF < MT) {
          lds_rd128h<HOFF + (i + PF) * 2048>(ah[(i + PF) % (PF + 1)], rd_s);
          asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ah[cur]) : "n"(PF) : "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ah[cur]) : "n"(MT - 1 - i) : "memory");
        }
#ifndef MXDBG_NOF16
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[0][i][j] = mfma_f16(whi[S][j], ah[cur], acc[0][i][j]);
#endif
        __builtin_amdgcn_sched_barrier(0);
      });
      pin_acc(acc);
      // this sub-step's W hi set for the next 128-deep step: a whole MX phase (and more) ahead of its use
      if constexpr (more) issue_whi(4 * b + 4 + S, whi[S]);
      __builtin_amdgcn_sched_barrier(0);
    });
    }
    // the next step's lo and scale units (their slot was last read in the MX phase of step b - 1, which every wave left before B1)
    if constexpr (more) issue_l8(b + 1);
    __builtin_amdgcn_sched_barrier(0);
    MX_STAMP(st_f16)
#ifndef MXDBG_NOCONV
    // ---- conversion: this wave turns the hi rows of row tiles 2 w, 2 w + 1 into fp6 for everybody
    {
      constexpr int CVT = MT / 4;      // row tiles this wave converts
      const unsigned cv_hi = rd_hi + (unsigned)(wave * CVT * 2048), cv_hi1 = cv_hi ^ 64u, cv_sc = rd_sc + (unsigned)(wave * CVT * 64 + (b & 1) * L_SC_SLOT);
      const unsigned cv_a = rd_h6a + (unsigned)(wave * CVT * 1024), cv_b = rd_h6b + (unsigned)(wave * CVT * 512);
      sfor<CVT>([&](auto u_c) {
        constexpr int U = decltype(u_c)::value;
        f16x8 h[4];
        unsigned sb;
        lds_rd128h<U * 2048>(h[0], cv_hi);
        lds_rd128h<U * 2048>(h[1], cv_hi1);
        lds_rd128h<U * 2048 + L_HALF>(h[2], cv_hi);
        lds_rd128h<U * 2048 + L_HALF>(h[3], cv_hi1);
        lds_rd8<U * 64>(sb, cv_sc);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(sb)::"memory");
        h32 hv;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e) hv[8 * s + e] = h[s][e];
        const u32x6 c = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hv, e8m0_float((int)sb + kMxShDelta));
        lds_wr128<U * 1024>(cv_a, u32v4{c[0], c[1], c[2], c[3]});
        lds_wr64<U * 512>(cv_b, u32v2{c[4], c[5]});
      });
    }
#endif
    // B2: fp6 rows visible to every wave, hi slots free.  W's fp6 images of THIS step (requested behind the previous MX phase) must have
    // landed before the MX phase reads them: only the requests made during this step's f16 phase are younger -- 4 x 3 W hi fragments and
    // the 5 pieces of the next lo / scale unit; none in the last step.  (A first version of the four-set W hi schedule had dropped the
    // wait that used to cover them: tests/test_gpu_e2e.py::test_classifier_bitwise_repeatable caught it.)
    MX_STAMP(st_cv)
    if constexpr (more) wait_vmcnt<4 * TN + 5>();
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    MX_STAMP(st_b2)
    if constexpr (more) {
      issue_hi(b + 1);
    } else if constexpr (ZK) {
      // the residual tile's first three 32-column units: two into the hi slots, one into the lo slot of the other parity
      issue_z(0, L_HI);
      issue_z(1, L_HI + L_HALF);
      issue_z(2, L_L8 + ((b + 1) & 1) * L_HALF);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!MX_DEAD) {
    // ---- MX phase
#ifndef MXDBG_NOMX
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(wl6a[j]), "+v"(wl6b[j]));
    asm volatile("" : "+v"(wsc));
    i32x8 wl6[TN], wh6[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      wl6[j] = op6(wl6a[j], wl6b[j]);
      wh6[j] = __builtin_bit_cast(i32x8, __builtin_shufflevector(wh6r[j], wh6r[j], 0, 1, 2, 3, 4, 5, -1, -1));
    }
    {
      const unsigned m_l8a = rd_l8 + (unsigned)(L_L8 + (b & 1) * L_HALF), m_l8b = m_l8a ^ 64u;
      const unsigned m_sc = rd_sc + (unsigned)((b & 1) * L_SC_SLOT);
      u32v4 la[2], lb[2], ha[2];
      u32v2 hb[2];
      unsigned sb[2];
#ifdef MXDBG_LO4
#define MX_LB_READ(dst, off, addr) dst = la[0]
#define MX_LGKM_PER_TILE 4
#define MX_LO_FMT 2
#else
#define MX_LB_READ(dst, off, addr) lds_rd128<off>(dst, addr)
#define MX_LGKM_PER_TILE 5
#define MX_LO_FMT 0
#endif
      lds_rd128<0>(la[0], m_l8a); MX_LB_READ(lb[0], 0, m_l8b); lds_rd128<0>(ha[0], rd_h6a); lds_rd64<0>(hb[0], rd_h6b); lds_rd8<0>(sb[0], m_sc);
      sfor<MT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int cur = i & 1, nxt = cur ^ 1;
        if constexpr (i + 1 < MT) {
          lds_rd128<(i + 1) * 2048>(la[nxt], m_l8a); MX_LB_READ(lb[nxt], (i + 1) * 2048, m_l8b);
          lds_rd128<(i + 1) * 1024>(ha[nxt], rd_h6a); lds_rd64<(i + 1) * 512>(hb[nxt], rd_h6b); lds_rd8<(i + 1) * 64>(sb[nxt], m_sc);
          asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(la[cur]), "+v"(lb[cur]), "+v"(ha[cur]), "+v"(hb[cur]), "+v"(sb[cur]) : "n"(MX_LGKM_PER_TILE) : "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(la[cur]), "+v"(lb[cur]), "+v"(ha[cur]), "+v"(hb[cur]), "+v"(sb[cur])::"memory");
        }
        const i32x8 al8 = op8(la[cur], lb[cur]), ah6 = op6(ha[cur], hb[cur]);
        const int asc = (int)(sb[cur] | ((sb[cur] + kMxShDelta) << 8));      // byte 0: lo scale, byte 1: fp6-hi scale
        // W hi' (fp6, its sh byte) x A lo (fp8, byte 0);  W lo' (fp6, its sl byte) x A hi' (fp6, byte 1)
        acc[0][i][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wh6[0], al8, acc[0][i][0], 2, MX_LO_FMT, 1, (int)wsh[0], 0, asc);
        acc[0][i][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wh6[1], al8, acc[0][i][1], 2, MX_LO_FMT, 3, (int)wsh[0], 0, asc);
        acc[0][i][2] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wh6[2], al8, acc[0][i][2], 2, MX_LO_FMT, 1, (int)wsh[1], 0, asc);
        acc[0][i][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wl6[0], ah6, acc[0][i][0], 2, 2, 0, (int)wsc[0], 1, asc);
        acc[0][i][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wl6[1], ah6, acc[0][i][1], 2, 2, 2, (int)wsc[0], 1, asc);
        acc[0][i][2] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wl6[2], ah6, acc[0][i][2], 2, 2, 0, (int)wsc[1], 1, asc);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    pin_acc(acc);
#endif
    wsh = wsc;      // the next step's hi scale bytes, before the next block's words are requested into wsc
    asm volatile("" : "+v"(wsh));
    if constexpr (more) issue_wx(b + 1);
    __builtin_amdgcn_sched_barrier(0);
    }
    MX_STAMP(st_mx)
  };

#ifdef RIBCA_KLOOP_PRIO      // A/B: see gemm_duo.hip
  __builtin_amdgcn_s_setprio(2);
#endif
  for (int b = 0; b + 1 < nb; ++b) step(std::true_type{}, b);
  step(std::false_type{}, nb - 1);
#ifdef RIBCA_KLOOP_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif

  // the lane's position again, from the hardware: r16 / g of the prologue then end with the address registers formed from them instead of
  // occupying registers (or scratch) across the K loop for the epilogue's sake
  int lane_e;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
  const int r16e = lane_e & 15, ge = lane_e >> 4;
  const int mbase = m0 + r16e, nbase = n0 + wn * (16 * TN) + 4 * ge;
  const unsigned xch = lds_base + L_XCH;
  if constexpr (ZK) {
    // everything the epilogue needs from memory: requested behind the last operand batch, in front of the residual units, whose six
    // barriers cover the round trip (20 registers that the K loop does not have to carry)
    float4 zb4[TN];
    float zpm[MT];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (16 * TN) + 16 * j + 4 * ge;
      zb4[j] = n < epi.N ? *reinterpret_cast<const float4*>(epi.bias + n) : float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + 16 * i + r16e;
      zpm[i] = (epi.prev != nullptr && m < M) ? epi.prev[(size_t)m * epi.prev_stride].y : 0.f;
    }
#ifndef MXDBG_NOZ
    // ---- the residual tile: unit t holds columns n0 + 32 t .. + 31 of the stored rows (gemm_duo.hip, EpiResidZK)
    {
      const int last = nb - 1;
      const unsigned rd_ps_hi = lds_base + (unsigned)lds_off(r16e, 2 * ge);      // packed-split rows: hi chunk 2 g, lo chunk at ^ 16
      const int zslot[4] = {L_HI, L_HI + L_HALF, L_L8 + ((last + 1) & 1) * L_HALF, L_L8 + (last & 1) * L_HALF};
      const int p8 = r16e & 7;
      const unsigned one = (p8 & 1) ? 0x3C000000u : 0x00003C00u;
      const u32x4 pat = {(p8 >> 1) == 0 ? one : 0u, (p8 >> 1) == 1 ? one : 0u, (p8 >> 1) == 2 ? one : 0u, (p8 >> 1) == 3 ? one : 0u};
      const u32x4 zero4 = {0u, 0u, 0u, 0u};
      const f16x8 id0 = __builtin_bit_cast(f16x8, ge == (r16e >> 3) ? pat : zero4), id16 = __builtin_bit_cast(f16x8, ge == 2 + (r16e >> 3) ? pat : zero4);
      int zsel[TN];
      f16x8 idj[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int c0 = wn * (16 * TN) + 16 * j;
        zsel[j] = n0 + c0 < epi.N ? (c0 >> 5) : -1;
        idj[j] = (c0 & 16) ? id16 : id0;
      }
      sfor<ZS>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        constexpr int younger = (ZS - 1 - t) < 2 ? (ZS - 1 - t) : 2;      // units requested behind unit t
        wait_vmcnt<4 * younger>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (t + 3 < ZS) issue_z(t + 3, zslot[(t + 3) & 3]);
        __builtin_amdgcn_sched_barrier(0);
        bool mine = false;
#pragma unroll
        for (int j = 0; j < TN; ++j) mine = mine || zsel[j] == t;
        if (mine) {      // wave-uniform
          const unsigned a_hi_s = rd_ps_hi + (unsigned)zslot[t & 3], a_lo_s = a_hi_s ^ 16u;
          f16x8 ah[2], al[2];
          lds_rd128h<0>(ah[0], a_hi_s);
          lds_rd128h<0>(al[0], a_lo_s);
          sfor<MT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int cur = i & 1, nxt = cur ^ 1;
            if constexpr (i + 1 < MT) {
              lds_rd128h<(i + 1) * 2048>(ah[nxt], a_hi_s);
              lds_rd128h<(i + 1) * 2048>(al[nxt], a_lo_s);
              asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[cur]), "+v"(al[cur])::"memory");
            } else {
              asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[cur]), "+v"(al[cur])::"memory");
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if (zsel[j] == t) {
                acc[0][i][j] = mfma_f16(idj[j], al[cur], acc[0][i][j]);
                acc[0][i][j] = mfma_f16(idj[j], ah[cur], acc[0][i][j]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          });
        }
      });
    }
#endif
#ifdef MXDBG_NOEPI
    if (true) {
#else
    if (ABL & 1) {
#endif
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[0][i][j]));
      return;
    }
    if (n0 + wn * (16 * TN) >= epi.N) return;      // (never with an MX3 copy: its launcher takes whole tiles only)
    const int blk = nt * 4 + wn;
#ifdef MXDBG_STAMP
    if (epi.part != nullptr && lane_e == 0) {      // behind the statistics: [N / 48][M] pairs, then 4 pairs per wave of every workgroup
      float2* o = epi.part + (size_t)(epi.N / 48) * epi.M + ((size_t)blockIdx.x * 4 + wave) * 4;
      o[0] = float2{(float)st_b1, (float)st_f16};
      o[1] = float2{(float)st_cv, (float)st_b2};
      o[2] = float2{(float)st_mx, (float)nb};
      const unsigned long long st_now = __builtin_amdgcn_s_memtime(), rt_now = __builtin_amdgcn_s_memrealtime();
      // everything behind the K loop; the shader clock this wave saw from its first instruction to here, in MHz (MI355X_MICROARCH.md, DVFS give-back item 6)
      o[3] = float2{(float)(st_now - st_prev), rt_now > rt_start ? 100.0f * (float)(st_now - st_start) / (float)(rt_now - rt_start) : 0.f};
    }
#endif
    const bool emit = epi.zmx.hi != nullptr;
    if (m0 + BM <= M) {
      resid_zk_epilogue<TN, MT, MT, true>(epi, mbase, nbase, blk, ge, acc, zb4, zpm);
      if constexpr (MT == 8) { if (emit) mx3_emit_wave48<MT, true>(epi.zmx, M, m0, n0 + wn * (16 * TN), ge, r16e, wave, lds_base, xch, acc[0], epi.nt != 0); }
    } else {
      resid_zk_epilogue<TN, MT, MT, false>(epi, mbase, nbase, blk, ge, acc, zb4, zpm);
      if constexpr (MT == 8) { if (emit) mx3_emit_wave48<MT, false>(epi.zmx, M, m0, n0 + wn * (16 * TN), ge, r16e, wave, lds_base, xch, acc[0], epi.nt != 0); }
    }
  } else {
    // N is a multiple of the tile width for these (gemm_mx_supported + the launchers): only the rows need guards
#ifdef MXDBG_NOEPI
    if (true) {
#else
    if (ABL & 1) {
#endif
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[0][i][j]));
      return;
    }
    if constexpr (is_mx_out<Epi>::value) {
      if constexpr (MT == 8) {
        if (m0 + BM <= M) gelu_mx48_epilogue<MT, true>(epi, m0, n0 + wn * (16 * TN), ge, r16e, wave, lds_base, xch, acc);
        else gelu_mx48_epilogue<MT, false>(epi, m0, n0 + wn * (16 * TN), ge, r16e, wave, lds_base, xch, acc);
      }
    } else {
      if (m0 + BM <= M) run_epilogue<TN, Epi, MT, true>(epi, mbase, nbase, acc[0]);
      else run_epilogue<TN, Epi, MT>(epi, mbase, nbase, acc[0]);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------- host side
bool gemm_mx_supported(int N, int Kp) { return N % 48 == 0 && N % 8 == 0 && Kp % 128 == 0 && Kp >= 128; }

template <class Epi, int ABL, bool HALF = false>
static void launch_mx_impl(const MxAct& A, const MxWeight& W, int M, int N, const Epi& epi, hipStream_t s) {
  const int mtiles = (M + MX_BM - 1) / MX_BM, ntiles = (N + MX_BN - 1) / MX_BN;
  const int nb = A.Kp / 128;
  void (*kernel)(MxAct, const uint16_t*, const unsigned char*, int, int, int, int, Epi, int) = gemm_mx_duo_kernel<Epi, ABL, HALF>;
  static unsigned long long attr_done = 0ull;
  if (!ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), L_TOTAL, attr_done)) return;
  // W-panel walk (see the kernel's tile map): RIBCA_MX_PANEL_KB = the most weight image (KB) a panel may hold, applied only where the
  // whole image is larger (qkv / fc1 at D = 576: 3.1 / 4.2 MB against the XCD's 4 MB L2); 0 = off
  static const int panel_kb = getenv("RIBCA_MX_PANEL_KB") ? atoi(getenv("RIBCA_MX_PANEL_KB")) : 0;
  int panel = 0;
  {
    const size_t tile_bytes = (size_t)4 * nb * (kMxWhBytes + kMxWxBytes), w_bytes = tile_bytes * ntiles;
    if (panel_kb > 0 && w_bytes > (size_t)panel_kb * 1024) {
      const int fit = (int)((size_t)panel_kb * 1024 / tile_bytes);
      if (fit >= 1) {
        const int np = (ntiles + fit - 1) / fit;
        panel = (ntiles + np - 1) / np;
      }
    }
  }
  kernel<<<dim3(mtiles * ntiles), dim3(256), L_TOTAL, s>>>(A, W.wh, W.wx, M, nb, mtiles, ntiles, epi, panel);
}

// z (packed-split) = (z - prev mean) + A W^T + bias with A in MX3, W in the mx_pack_w image; statistics per 48-column wave block
ResidStatGeom launch_gemm_mx_resid(const MxAct& A, const MxWeight& W, int M, int N, const float* bias, uint16_t* z, int ldz, float2* part, const float2* prev,
                                   int prev_stride, hipStream_t s, int abl, const MxAct* zmx) {
  EpiResidZK epi{z, ldz, bias, M, N, part, prev, prev_stride};
  static const int mx_nt = getenv("RIBCA_MX_NT") ? atoi(getenv("RIBCA_MX_NT")) : 2;
  epi.nt = (mx_nt >> 3) & 1;
  if (zmx != nullptr) {
    // (whole 192-column tiles: every wave of a workgroup then reaches the barrier of the emission)
    if (N % MX_BN != 0) { launch_error("launch_gemm_mx_resid: an MX3 copy of the residual rows needs N %% 192 == 0 (N = %d)", N); return ResidStatGeom{N / 48, 48}; }
    epi.zmx = *zmx;
  }
#ifdef RIBCA_DIAG
  if (abl == 1) { launch_mx_impl<EpiResidZK, 1>(A, W, M, N, epi, s); return ResidStatGeom{N / 48, 48}; }
#endif
  (void)abl;
  // RIBCA_MX_HALF=0: the whole-tile kernel for the ragged last column tile as well (A/B; the results are the same bits either way)
  static const bool half_on = !(getenv("RIBCA_MX_HALF") && atoi(getenv("RIBCA_MX_HALF")) == 0);
  if (N % MX_BN != 0 && half_on) launch_mx_impl<EpiResidZK, 0, true>(A, W, M, N, epi, s);
  else launch_mx_impl<EpiResidZK, 0>(A, W, M, N, epi, s);
  return ResidStatGeom{N / 48, 48};
}

// qkv with the LayerNorm fold (EpiQKVLn exactly as launch_gemm_qkv_ln builds it, gemm_split16.hip)
void launch_gemm_mx_qkv_ln(const MxAct& A, const MxWeight& W, int M, int N, const float* bias, const float2* rowstat, const float* csum, uint16_t* q,
                           uint16_t* k, uint16_t* vt, const AttnGeom& a, float scale, hipStream_t s) {
  if (N % MX_BN != 0) { launch_error("launch_gemm_mx_qkv_ln needs N %% 192 == 0 (N = %d)", N); return; }
  // RIBCA_MX_NT: bit 0 = the q / k / v rows stored non-temporal (A/B: no effect measured), bit 1 = the MX3 planes of h (default on, below)
  static const int mx_nt = getenv("RIBCA_MX_NT") ? atoi(getenv("RIBCA_MX_NT")) : 2;
  const EpiQKVLn epi{q, k, vt, bias, a.D, a.hd, a.hdq, a.hdv, scale, M, N, a.T, a.TP, a.H, a.KP, mx_nt & 1,
                     (unsigned)((0x100000000ull + (unsigned long long)a.T - 1) / (unsigned long long)a.T), rowstat, csum, attention_v_rowmajor(a) ? 1 : 0,
                     0, 0, 1};
  launch_mx_impl<EpiQKVLn, 0>(A, W, M, N, epi, s);
}
void launch_gemm_mx_gelu(const MxAct& A, const MxWeight& W, int M, int N, const float* bias, const float2* rowstat, const float* csum, const MxAct& out,
                         hipStream_t s) {
  if (N % MX_BN != 0 || out.Kp != N) { launch_error("launch_gemm_mx_gelu needs N %% 192 == 0 and out.Kp == N (N = %d, out.Kp = %d)", N, out.Kp); return; }
  // The emitted planes of h (715 MB per launch at D = 576) are read once, by the next launch: stored non-temporal they do not displace
  // the weight image from the XCD's L2 -- fetch 1689 -> 1125 MB per launch, fc1 -2 %, fc2 -1 % (profiles/r4/ab_mx_nt_stores.txt).
  // RIBCA_MX_NT bit 1 = 0 for A/B.
  static const int mx_nt = getenv("RIBCA_MX_NT") ? atoi(getenv("RIBCA_MX_NT")) : 2;
  const EpiGeluMx epi{out, bias, M, N, rowstat, csum, 1, (mx_nt >> 1) & 1};
  launch_mx_impl<EpiGeluMx, 0>(A, W, M, N, epi, s);
}

}  // namespace ribca
