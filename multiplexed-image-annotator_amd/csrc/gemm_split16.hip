// fp16x3 (split-operand) MFMA GEMM for the ViT linears (timm Attention.qkv / Attention.proj / Mlp.fc1 / Mlp.fc2; the imputer's embeddings and
// prediction head too), reached from reference cell_type_annotation/model.py:402 ``model(x)``).
//
//   C[m][n] = sum_k A[m][k] * W[n][k]        A: activations  [M ][2*Kp] packed-split fp16 (ribca_common.h)
//                                            W: nn.Linear wt [Np][2*Kp] packed-split fp16, Np = N padded to the tile
//
// Shape regime: M = cells*101 is huge (1e4..1e6), N in {144..2304}, K in {160..2304}: short K loops and a weight matrix
// that lives in L2 / Infinity Cache, so the kernel is built around keeping the MFMA pipe fed from a deep LDS ring:
//
// * tile 256 x {64,96,128}, BK = 32; 768 threads = 8 CONSUMER waves as 4(M) x 2(N) (each owns 64 x BN/2 outputs = 4 x TN
//   tiles of 16x16) + 4 LOADER waves that only issue the direct-to-LDS loads.  (A 128-row tile was measured 10-15 % slower:
//   it needs 1/3 more L2->LDS bytes per MFMA.)
// * Tiles are computed TRANSPOSED: acc = mfma(Wfrag, Afrag) so a lane holds 4 consecutive output columns n of one row m
//   (C/D map: col = lane&15 -> m, row = 4*(lane>>4)+r -> n).  Epilogues then issue one 8/16-byte store per tile
//   instead of four 2-byte ones (residual RMW on fp32 z is one float4).
// * Each (Afrag, Wfrag) pair feeds three MFMAs (hi*hi, lo*hi, hi*lo): LDS bytes per MFMA are 2/3 of a plain bf16 GEMM.
// * 3-stage LDS ring (3 x 48 KB) filled by global_load_lds_dwordx4 (no staging registers, no ds_write): K-step k+2 is issued
//   right after the barrier that opens step k, so two steps of loads are always in flight; completion is tracked with a
//   counted s_waitcnt vmcnt(G) on the loader waves + raw s_barrier.
// * Half-step stagger: two barriers per K step split it into a fragment-read phase and an MFMA phase, and consumer waves 4-7
//   (the SIMD partners of waves 0-3) run one phase late, so each SIMD always has one wave multiplying while the other reads.
// * LDS tile = rows of 128 B (one 32-deep K step of a PS row: 4 x [16 B hi | 16 B lo]).  16-byte chunk c of row r lives at
//   chunk c ^ f(r),  f(r) = ((r>>1)&7) ^ (4 <= (r&15) < 12 ? 2 : 0):  every hardware ds_read_b128 lane group
//   ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) then touches 16 distinct 16-byte slots of the 256-byte bank row
//   (SQ_LDS_BANK_CONFLICT measured 0).  The LDS-DMA destination is linear (wave base + lane*16), so the swizzle is applied to
//   each lane's GLOBAL source address.
// * block id -> tile map is XCD-aware (blocks b, b+8, ... share an L2): every XCD walks whole rows of n-tiles of one
//   m-tile, so an A tile is fetched into one L2 once and reused by all its n-tiles.
// * Epilogue through LDS: the finished tile is parked in the (dead) ring and all 12 waves write it out in row-major 16-byte
//   chunks -- whole 256-512 byte row segments per instruction (what a CU can drain depends on the address shape of a store,
//   tools/store_bench.hip).  It is not overlapped with the next tile's K loop; DESIGN.md section 6 lists the persistent /
//   streaming / relay forms that were built for that and why they did not pay (output traffic, not the latency chain).
//   gemm_duo.hip is the two-workgroups-per-CU form of this kernel (weights out of LDS, in fragment order): bit-identical, reached
//   through variants 40-49 / RIBCA_GEMM_DUO=1, not faster end to end.
#include <cstdlib>
#include <type_traits>

#include "gemm_epi.h"
#include "ribca_common.h"
#include "ribca_kernels.h"

namespace ribca {

// Diagnostic stamps (measurement builds only: variant 12): wave 0 of every workgroup writes s_memrealtime (100 MHz) at kernel
// entry, after the first stage has landed, after the K loop and after its epilogue, plus the XCC/CU it ran on, into a
// buffer nothing else reads.
__device__ unsigned long long* g_stamps = nullptr;
__device__ unsigned int g_stamp_cap = 0;      // workgroups the buffer has room for: larger grids do not stamp
constexpr int kStampStride = 20;   // per workgroup: t0..t3, xcc, hw id, then the epilogue end of each of the 12 waves
__device__ __forceinline__ unsigned long long stamp_now() { return __builtin_amdgcn_s_memrealtime(); }

// ---------------------------------------------------------------------------------------------- epilogue through LDS
// tools/store_bench.hip: what one CU can push to memory depends on the ADDRESS SHAPE of each store instruction, not on its
// bytes: 16 rows x 64 B (what the accumulator layout gives: 16 rows m per lane group) drains at ~27 GB/s per CU, 16 rows x
// 2 x 16 B pieces at ~15 GB/s, whole 256-512 B row segments at ~67 GB/s (the chip-wide write limit is ~5 TB/s, i.e. ~75 CUs
// at that rate).  After the K loop the ring is dead, so the tile's 256 x BN fp32 accumulators are parked there (row pitch
// 4 BN + 16 bytes: the 16 lanes of a ds_write_b128 group hit 16 different bank quads) and ALL 12 waves then walk the tile
// in row-major 16-byte chunks: a wave instruction covers 1 KB = 2-4 whole row segments, loads of z are equally contiguous,
// the GELU / split work is spread over 768 threads, and a chunk's PS partner (the other 4 columns of its 8-group) is the
// neighbouring lane.  A V tile of the qkv product scatters 2-byte elements along V^T rows and keeps the register path.
template <class Epi> __device__ __forceinline__ bool lepi_tile_uses_registers(const Epi&, int, int) { return false; }
template <> __device__ __forceinline__ bool lepi_tile_uses_registers<EpiQKV>(const EpiQKV& e, int n0, int bn) { return !e.vrow && n0 + bn > 2 * e.D; }


template <class Epi> __device__ __forceinline__ auto lepi_plain(const Epi& e) {
  if constexpr (Epi::kFold) return e.plain();
  else return e;
}

template <int BN>
__device__ __forceinline__ void lds_park(char* smem, const f32x4 (&acc)[4][BN / 32], int wm, int wn, int r16, int g) {
  constexpr int SROW = BN * 4 + 16;          // staging row pitch in bytes
  char* sb = smem + (wm * 64 + r16) * SROW + (wn * (BN / 2) + 4 * g) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < BN / 32; ++j) *reinterpret_cast<f32x4*>(sb + i * 16 * SROW + j * 64) = acc[i][j];
}

template <int BN, class Epi>
__device__ __forceinline__ void lds_drain(const Epi& epi, const char* smem, int m0, int n0, int tid) {
  constexpr int SROW = BN * 4 + 16;
  constexpr int CPR = BN / 4;                // 16-byte chunks per row
  constexpr int RSTEP = 768 / CPR;           // rows covered by one pass of the workgroup
  constexpr int NIT = (256 + RSTEP - 1) / RSTEP;
  static_assert(768 % CPR == 0, "a thread keeps its column chunk across passes");
  const int c = tid % CPR, row0 = tid / CPR;
  const int n = n0 + 4 * c;
  float4 b4 = epi.fetch_bias(n), c4 = epi.fetch_csum(n);
  typename Epi::Ctx ctx[NIT];
  typename Epi::RowS rs[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = row0 + it * RSTEP;
    if (NIT * RSTEP == 256 || row < 256) {
      epi.fetch(m0 + row, n, ctx[it]);
      rs[it] = epi.fetch_row(m0 + row);
    } else {
      rs[it] = typename Epi::RowS{};
    }
  }
  // ALL of this thread's chunks leave LDS before the first store is issued.  The kernel contains LDS-DMA, so hipcc guards every
  // use of a ds_read result with s_waitcnt vmcnt(0) (an LDS-DMA could be pending for all it knows) -- interleaved with the stores
  // (read, compute, store, read, ...) that wait also drained the PREVIOUS iteration's global store, i.e. one HBM write round
  // trip per iteration, 8-11 times per tile: most of the 6-9 us the epilogue took (tools/epilogue_scaling.py).  With the reads
  // hoisted the single wait sits before any store and the stores stream out back to back.
  const char* src = smem + row0 * SROW + c * 16;
  f32x4 vals[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = row0 + it * RSTEP;
    vals[it] = (NIT * RSTEP == 256 || row < 256) ? *reinterpret_cast<const f32x4*>(src + it * RSTEP * SROW) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // Passing the values through an (empty) asm statement makes the compiler's wait-count pass settle every LDS read HERE; without
  // it each first use inside the store loop re-emits s_waitcnt vmcnt(0) and serialises on the store issued just before.
#pragma unroll
  for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(vals[it]));
  settle(b4);
  if constexpr (Epi::kFold) settle(c4);
#pragma unroll
  for (int it = 0; it < NIT; ++it) { settle_ctx(ctx[it]); settle_row(rs[it]); }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (has_rowcol<Epi>::value) {
    const typename Epi::Col col = epi.col(n);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = row0 + it * RSTEP;
      if (NIT * RSTEP == 256 || row < 256) epi.template apply<1>(m0 + row, n, vals[it], b4, c4, rs[it], ctx[it], epi.row(m0 + row), col);
    }
  } else {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = row0 + it * RSTEP;
      if (NIT * RSTEP == 256 || row < 256) epi.template apply<1>(m0 + row, n, vals[it], b4, c4, rs[it], ctx[it]);
    }
  }
}


// ---------------------------------------------------------------------------------------------- residual drain, packed-split z
template <int CTRL> __device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row, the same bits in every lane of the row, fixed order: four VALU steps, no LDS round trip
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_get<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_get<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_get<0x141>(v);     // row_half_mirror: the other quad of the 8
  v += dpp_get<0x140>(v);     // row_mirror: the other 8 of the 16
  return v;
}
// EpiResidPS: z (fp16 hi + lo) = (z - previous mean) + tile + bias, and the row statistics of the new segment.  A lane owns one whole PS
// group of a tile row -- 8 columns: two 16-byte chunks of the parked tile, one 16-byte hi and one 16-byte lo vector of z -- so
// nothing is exchanged between lanes; the 16 lanes of a DPP row share a tile row (BN / 8 <= 16 of them active), a wave covers 4 rows
// per pass, 12 waves 48, six passes the tile.  Statistics are taken over the fp32 values before the split (the stored row differs by
// 2^-23 relative per element: 1e-9 on a mean that the fold multiplies by O(1)) with DPP reductions in a fixed order.
template <int BN>
__device__ __forceinline__ void lds_drain_resid_ps(const EpiResidPS& epi, const char* smem, const char* stage, int m0, int n0, int nt, int tid) {
  constexpr int SROW = BN * 4 + 16, GPR = BN / 8, NIT = 6;     // 12 waves x 4 rows x 6 passes = 288 >= 256 rows
  static_assert(GPR <= 16, "a tile row fits one DPP row of lanes");
  const int lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15;
  const int n = n0 + 8 * c;
  const bool col_ok = c < GPR && n < epi.N;      // N % 8 == 0
  const int rbase = 4 * wave + (lane >> 4);
  uint4 zh[NIT], zl[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = rbase + 48 * it;
    const bool ok = col_ok && row < 256 && m0 + row < epi.M;
    const uint16_t* zp = epi.z + (size_t)(m0 + row) * epi.ldz + 2 * n;
    zh[it] = ok ? *reinterpret_cast<const uint4*>(zp) : uint4{0u, 0u, 0u, 0u};
    zl[it] = ok ? *reinterpret_cast<const uint4*>(zp + 8) : uint4{0u, 0u, 0u, 0u};
  }
  f32x4 va[NIT], vb[NIT];
  float pmean[NIT];
  const char* src = smem + rbase * SROW + c * 32;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = rbase + 48 * it;
    const bool rd = c < GPR && row < 256;
    va[it] = rd ? *reinterpret_cast<const f32x4*>(src + it * 48 * SROW) : f32x4{0.f, 0.f, 0.f, 0.f};
    vb[it] = rd ? *reinterpret_cast<const f32x4*>(src + it * 48 * SROW + 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    pmean[it] = row < 256 ? reinterpret_cast<const float2*>(stage + row * 8)->y : 0.f;      // mean of the stored row (0: no re-centring)
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(va[it]), "+v"(vb[it]), "+v"(pmean[it]));
#pragma unroll
  for (int it = 0; it < NIT; ++it)
    asm volatile("" : "+v"(zh[it].x), "+v"(zh[it].y), "+v"(zh[it].z), "+v"(zh[it].w), "+v"(zl[it].x), "+v"(zl[it].y), "+v"(zl[it].z), "+v"(zl[it].w));
  __builtin_amdgcn_sched_barrier(0);
  f32x2p xs[NIT][4];
  float s[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = rbase + 48 * it;
    const bool ok = col_ok && row < 256 && m0 + row < epi.M;
    const uint32_t hw[4] = {zh[it].x, zh[it].y, zh[it].z, zh[it].w}, lw[4] = {zl[it].x, zl[it].y, zl[it].z, zl[it].w};
    const f32x2p acc2[4] = {f32x2p{va[it][0], va[it][1]}, f32x2p{va[it][2], va[it][3]}, f32x2p{vb[it][0], vb[it][1]}, f32x2p{vb[it][2], vb[it][3]}};
    const f32x2p pm = {pmean[it], pmean[it]};
    uint32_t nh[4], nl[4];
    f32x2p tot = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {      // two columns at a time: packed fp32 adds
      const f32x2p zo = {f16lo_plus_f16lo(hw[j], lw[j]), f16hi_plus_f16hi(hw[j], lw[j])};      // hi + lo: exact
      f32x2p x = (zo - pm) + acc2[j];              // the tile already carries the bias (consumers, before parking)
      x.x = clamp_f16_range(x.x); x.y = clamp_f16_range(x.y);
      nh[j] = cvt_pk_f16(x.x, x.y);
      nl[j] = cvt_pk_f16(f32_minus_f16lo(x.x, nh[j]), f32_minus_f16hi(x.y, nh[j]));
      xs[it][j] = x;
      tot += x;
    }
    if (ok) {
      uint16_t* zp = epi.z + (size_t)(m0 + row) * epi.ldz + 2 * n;
      *reinterpret_cast<u32x4*>(zp) = u32x4{nh[0], nh[1], nh[2], nh[3]};
      *reinterpret_cast<u32x4*>(zp + 8) = u32x4{nl[0], nl[1], nl[2], nl[3]};
    }
    s[it] = ok ? tot.x + tot.y : 0.f;
  }
  if (epi.part == nullptr) return;
  const int ncols = (epi.N - n0) < BN ? (epi.N - n0) : BN;
  const float inv = 1.0f / (float)ncols;
#pragma unroll
  for (int it = 0; it < NIT; ++it) s[it] = row16_sum(s[it]) * inv;       // tile mean of the row
  float q[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = rbase + 48 * it;
    const bool ok = col_ok && row < 256 && m0 + row < epi.M;
    const f32x2p mu = {s[it], s[it]};
    f32x2p qq = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) { const f32x2p d = xs[it][j] - mu; qq += d * d; }
    q[it] = ok ? qq.x + qq.y : 0.f;
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) q[it] = row16_sum(q[it]);
  if (c == 0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = rbase + 48 * it;
      if (row < 256 && m0 + row < epi.M) epi.part[(size_t)nt * epi.M + m0 + row] = float2{s[it], q[it]};
    }
  }
}
template <int BN, class Epi>
__device__ __forceinline__ void lds_drain_any(const Epi& epi, const char* smem, int m0, int n0, int nt, int tid) {
  if constexpr (std::is_same<Epi, EpiResidPS>::value) lds_drain_resid_ps<BN>(epi, smem, smem + 3 * (256 + BN) * ROWB, m0, n0, nt, tid);
  else if constexpr (Epi::kFold) lds_drain<BN>(epi.plain(), smem, m0, n0, tid);      // the consumers folded before parking
  else lds_drain<BN>(epi, smem, m0, n0, tid);
}

// ---------------------------------------------------------------------------------------------- folded LayerNorm, in registers
// Epi::kFold: behind the ring the workgroup keeps (rstd, -mean rstd) of its 256 rows, the column sums and the folded bias of its BN
// columns (2 KB + 8 BN bytes, written by the loader waves while the first ring stages are in flight).  After the K loop every
// consumer lane turns its accumulators into  x = rstd acc + (nm c + b')  -- 4 rows and TN column groups per lane -- so that the
// epilogue proper (park, drain, or the register path of the V tiles) is the plain one with nothing left to add.
constexpr int kFoldRowBytes = 256 * 8;
template <int BN> constexpr int fold_lds_bytes() { return kFoldRowBytes + 8 * BN; }
template <int BN, int TN>
__device__ __forceinline__ void fold_accumulators(f32x4 (&acc)[4][TN], const char* xs, int wm, int wn, int r16, int g) {
  float2 rs[4];
  float4 c4[TN], b4[TN];
#pragma unroll
  for (int i = 0; i < 4; ++i) rs[i] = *reinterpret_cast<const float2*>(xs + (wm * 64 + i * 16 + r16) * 8);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = wn * (BN / 2) + j * 16 + 4 * g;
    c4[j] = *reinterpret_cast<const float4*>(xs + kFoldRowBytes + col * 4);
    b4[j] = *reinterpret_cast<const float4*>(xs + kFoldRowBytes + 4 * BN + col * 4);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      acc[i][j][0] = fmaf(rs[i].x, acc[i][j][0], fmaf(rs[i].y, c4[j].x, b4[j].x));
      acc[i][j][1] = fmaf(rs[i].x, acc[i][j][1], fmaf(rs[i].y, c4[j].y, b4[j].y));
      acc[i][j][2] = fmaf(rs[i].x, acc[i][j][2], fmaf(rs[i].y, c4[j].z, b4[j].z));
      acc[i][j][3] = fmaf(rs[i].x, acc[i][j][3], fmaf(rs[i].y, c4[j].w, b4[j].w));
    }
}

// ---------------------------------------------------------------------------------------------- loader/consumer split
// Same tile (256 x BN, BK 32, 3-deep ring) with ROLES: waves 0-7 only read LDS and issue MFMAs; waves 8-11 only issue the
// direct-to-LDS loads.  An LDS-DMA instruction costs its wave ~100 issue cycles, and 6 of them per wave right after each
// barrier (every wave at once, both SIMD partners) left the matrix pipe idle for a third of each K step; on dedicated waves
// that cost overlaps the consumers' MFMAs (3 waves per SIMD: 2 consumers + 1 loader).  Consumers never touch vmcnt in the
// loop, so their epilogue loads/stores cannot drain the ring.
// timing ablations (measurement builds only; results are wrong on purpose): 1 = no loads, 2 = no MFMA/LDS reads, 3 = no epilogue,
// 4 = stamps, 5 = no loads and no epilogue, 6 / 7 = two / one MFMA per operand pair, 8 / 9 = the same without the epilogue
constexpr bool abl_no_loads(int a) { return a == 1 || a == 5; }
constexpr bool abl_no_epi(int a) { return a == 3 || a == 5 || a == 8 || a == 9; }
constexpr int abl_passes(int a) { return (a == 6 || a == 8) ? 2 : (a == 7 || a == 9) ? 1 : 3; }
template <int BN, class Epi, int ABL = 0,
          bool STAG = false, bool LEPI = false /* epilogue through LDS: row-contiguous stores by all 12 waves */>
__global__ __launch_bounds__(768) void gemm_ps_split_kernel(const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ W, int ldw,
                                                            int M, int Kp, int mtiles, int ntiles, Epi epi, int deph) {
  constexpr int BM = 256, NST = 3, NLW = 4;
  constexpr int TN = BN / 32;
  constexpr int ROWS = BM + BN;
  constexpr int STAGE = ROWS * ROWB;
  constexpr int NGRP = ROWS / 8;
  constexpr int GPL = (NGRP + NLW - 1) / NLW;   // DMA groups per loader wave per stage (10..12)
  extern __shared__ __attribute__((aligned(16))) char smem[];

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
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = Kp / BK;

  // De-phasing: the first workgroup on every CU starts 0..31/32 of a tile time late (spread within each XCD), so that the
  // CUs' epilogue bursts interleave instead of all hitting the memory system together (chip-wide write limit ~5 TB/s = ~20
  // GB/s per CU when all 256 store at once, against ~67 GB/s that one CU can sustain with row-contiguous stores).  Later
  // workgroups inherit the phase of the CU they land on.  Consumers wait; loaders prefetch and meet them at the barrier.
  if (deph > 0 && blockIdx.x < 256 && wave < 8) {
    const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(((blockIdx.x >> 3) & 31) * deph) / 32;
    while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(8);
  }

  if (wave >= 8) {
    // ------------------------------------------------------------------ loader
    const int lw = wave - 8;
    const uint16_t* src[GPL];
    int dst[GPL];
#pragma unroll
    for (int i = 0; i < GPL; ++i) {
      int grp = lw + NLW * i;
      grp = grp < NGRP ? grp : NGRP - 1;
      const int row = grp * 8 + (lane >> 3);
      const int ch = (lane & 7) ^ swz_f(row);
      if (row < BM) {
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        src[i] = A + (size_t)gm * lda + ch * 8;
      } else {
        src[i] = W + (size_t)(n0 + row - BM) * ldw + ch * 8;
      }
      dst[i] = grp * 1024;
    }
    auto issue = [&](int kk, int stage) {
      char* st = smem + stage * STAGE;
      const int ko = kk * (2 * BK);
#pragma unroll
      for (int i = 0; i < GPL; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + ko),
                                         (__attribute__((address_space(3))) void*)(st + dst[i]), 16, 0, 0);
    };
    // z touch (EpiResid, LDS epilogue): lane t of the 256 loader lanes owns row t of the tile and reads one dword of each of its
    // BN / 32 lines, issued right BEHIND the last ring stage: vmcnt retires in order, so the two remaining stage waits simply
    // leave these NT youngest operations in flight.  The results are never used; the destination registers are kept allocated
    // until the drain's own waits have passed (empty asm at the end of this branch).
    constexpr int NT = (Epi::kTouch && LEPI && STAG && (ABL == 0 || ABL == 4)) ? BN / 32 : 0;
    unsigned int touched[NT > 0 ? NT : 1];
    auto touch = [&]() {
      if constexpr (NT > 0) {
        const int trow = lw * 64 + lane;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          // (switched off for A/B, the same instruction reads the start of z: an L2 hit that keeps the vmcnt arithmetic unchanged)
          const float* p = epi.touch_on() ? epi.touch_ptr(m0 + trow, n0 + 32 * i) : epi.touch_ptr(0, 0);
          asm volatile("global_load_dword %0, %1, off" : "=v"(touched[i]) : "v"(p) : "memory");
        }
      }
    };
    // folded LayerNorm: this lane's row statistics and (loader waves 0 / 1) a chunk of the column sums / folded bias, requested
    // in FRONT of the first ring stage -- vmcnt retires in order, so the wait for stage 0 covers them -- and copied behind the ring
    unsigned long long fold_rs = 0;
    u32x4 fold_cb = {0u, 0u, 0u, 0u};
    constexpr bool kResidPS = std::is_same<Epi, EpiResidPS>::value;
    if constexpr (Epi::kFold) {
      int gm = m0 + lw * 64 + lane;
      gm = gm < M ? gm : M - 1;
      const float2* rp = epi.rowstat + (size_t)gm * epi.rs_stride;
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(fold_rs) : "v"(rp) : "memory");
      {      // every loader wave issues it (no branch around an asynchronous load, see below); waves 0 / 1 keep theirs
        const int n = n0 + 4 * lane;
        const float* cp = (lw == 0 ? epi.csum : epi.bias) + ((lane < BN / 4 && n < epi.N) ? n : 0);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(fold_cb) : "v"(cp) : "memory");
      }
    }
    // packed-split residual epilogue: the mean each stored row had (re-centring) and the tile's bias travel the same way: the consumers
    // add the bias to their accumulators before parking, the drain reads the row's mean from LDS instead of global memory
    // ((z - mean) is kept as its own, nearly exact, subtraction: that is what makes the re-centring free of rounding)
    if constexpr (kResidPS) {
      int gm = m0 + lw * 64 + lane;
      gm = gm < M ? gm : M - 1;
      // (unconditional: a load whose destination the compiler may merge with another value behind a branch would be copied before
      // its data has arrived -- the asm statement only ISSUES it; without re-centring it reads the bias and the value is dropped below)
      const float2* rp = epi.prev != nullptr ? epi.prev + (size_t)gm * epi.prev_stride : reinterpret_cast<const float2*>(epi.bias);
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(fold_rs) : "v"(rp) : "memory");
      {
        const int n = n0 + 4 * lane;
        const float* cp = epi.bias + ((lane < BN / 4 && n < epi.N) ? n : 0);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(fold_cb) : "v"(cp) : "memory");
      }
    }
    if (!abl_no_loads(ABL)) {
      issue(0, 0);
      if (nk > 1) issue(1, 1);
    }
    if (nk <= 2) touch();
    int cur = 0;
    for (int kk = 0; kk < nk; ++kk) {
      // stage kk has landed; one younger stage (and, behind the last stage, the touches) may stay in flight
      const bool after_touch = NT > 0 && kk + 2 >= nk;
      if (kk + 1 < nk) {
        if (after_touch) wait_vmcnt<GPL + NT>();
        else wait_vmcnt<GPL>();
      } else {
        if (after_touch) wait_vmcnt<NT>();
        else wait_vmcnt<0>();
      }
      if constexpr (kResidPS) {
        if (kk == 0) {
          asm volatile("" : "+v"(fold_rs), "+v"(fold_cb));
          char* xs = smem + NST * STAGE;
          const float2 sm = __builtin_bit_cast(float2, fold_rs);      // (rstd, mean) of the stored row, or zeros
          *reinterpret_cast<float2*>(xs + (lw * 64 + lane) * 8) = float2{0.f, epi.prev != nullptr ? sm.y : 0.f};
          if (lw == 1 && lane < BN / 4) {
            const bool ok = n0 + 4 * lane < epi.N;
            *reinterpret_cast<u32x4*>(xs + kFoldRowBytes + 4 * BN + lane * 16) = ok ? fold_cb : u32x4{0u, 0u, 0u, 0u};
          }
        }
      }
      if constexpr (Epi::kFold) {
        if (kk == 0) {      // consumers read this after the K loop, many barriers from here
          asm volatile("" : "+v"(fold_rs), "+v"(fold_cb));      // (the values exist from here on: the wait above covered their loads)
          char* xs = smem + NST * STAGE;
          const float2 sm = __builtin_bit_cast(float2, fold_rs);      // memory: (rstd, mean); the fold wants (rstd, -mean rstd)
          *reinterpret_cast<float2*>(xs + (lw * 64 + lane) * 8) = float2{sm.x, -sm.y * sm.x};
          if (lw < 2 && lane < BN / 4) {
            const bool ok = n0 + 4 * lane < epi.N;
            *reinterpret_cast<u32x4*>(xs + kFoldRowBytes + lw * (4 * BN) + lane * 16) = ok ? fold_cb : u32x4{0u, 0u, 0u, 0u};
          }
        }
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (!abl_no_loads(ABL) && kk + 2 < nk) {
        int nxt = cur + 2;
        nxt = nxt >= NST ? nxt - NST : nxt;
        issue(kk + 2, nxt);
        if (kk + 3 == nk) touch();              // that was the last stage
      }
      if (STAG) __builtin_amdgcn_s_barrier();   // phase 2kk+1
      cur = cur + 1 == NST ? 0 : cur + 1;
    }
    if (STAG) __builtin_amdgcn_s_barrier();     // phase 2nk
    if constexpr (LEPI && STAG && !abl_no_epi(ABL)) {
      if (!lepi_tile_uses_registers(lepi_plain(epi), n0, BN)) {
        __syncthreads();                          // consumers have parked the tile
        lds_drain_any<BN>(epi, smem, m0, n0, nt, tid);
      }
    }
    if (ABL == 4 && g_stamps != nullptr && blockIdx.x < g_stamp_cap && lane == 0) {      // per-wave end of the epilogue (stores accepted)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      g_stamps[(size_t)blockIdx.x * kStampStride + 6 + wave] = stamp_now();
    }
    if constexpr (NT > 0) {
#pragma unroll
      for (int i = 0; i < NT; ++i) asm volatile("" ::"v"(touched[i]));
    }
    return;
  }

  // -------------------------------------------------------------------- consumer
  unsigned long long t0 = 0, t1 = 0, t2 = 0, tc1 = 0, tc2 = 0;
  if (ABL == 4) t0 = stamp_now();
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, g = lane >> 4;
  int a_rd[4], w_rd[TN];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_rd[i] = lds_off(wm * 64 + i * 16 + r16, 2 * g);
#pragma unroll
  for (int i = 0; i < TN; ++i) w_rd[i] = BM * ROWB + lds_off(wn * (BN / 2) + i * 16 + r16, 2 * g);
  f32x4 acc[4][TN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  f16x8 ahi[4], alo[4], whi[TN], wlo[TN];
  auto read_frags = [&](const char* st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ahi[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(st + a_rd[i]));
      alo[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(st + (a_rd[i] ^ 16)));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      whi[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(st + w_rd[j]));
      wlo[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(st + (w_rd[j] ^ 16)));
    }
  };
  auto mfmas = [&]() {
    if constexpr (abl_passes(ABL) >= 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_f16(wlo[j], ahi[i], acc[i][j]);
    }
    if constexpr (abl_passes(ABL) >= 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_f16(whi[j], alo[i], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = mfma_f16(whi[j], ahi[i], acc[i][j]);
  };

  int cur = 0;
  if (!STAG) {
    for (int kk = 0; kk < nk; ++kk) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (ABL != 2) {
        read_frags(smem + cur * STAGE);
        mfmas();
      }
      cur = cur + 1 == NST ? 0 : cur + 1;
    }
  } else {
    // Two barriers per K step split it into a fragment-read phase and an MFMA phase; waves 4-7 (the SIMD partners of
    // waves 0-3) run half a step late, so on every SIMD one wave's LDS burst overlaps the other wave's 48 MFMAs.
    //   phase 2s   : waves 0-3 read(s)   | waves 4-7 mfma(s-1)
    //   phase 2s+1 : waves 0-3 mfma(s)   | waves 4-7 read(s)
    const bool late = wave >= 4;
    if (late) __builtin_amdgcn_s_barrier();          // phase 0: nothing to multiply yet
    for (int kk = 0; kk < nk; ++kk) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (ABL == 4 && kk == 0) { t1 = stamp_now(); tc1 = __builtin_amdgcn_s_memtime(); }
      if (ABL != 2) read_frags(smem + cur * STAGE);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (ABL != 2) mfmas();
      cur = cur + 1 == NST ? 0 : cur + 1;
    }
    if (!late) __builtin_amdgcn_s_barrier();         // phase 2nk: partners finish their last MFMAs
  }
  if (ABL == 4) { t2 = stamp_now(); tc2 = __builtin_amdgcn_s_memtime(); }
  if (abl_no_epi(ABL)) {   // timing ablation: no epilogue (accumulators kept alive so the MFMAs are not dead code)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  if constexpr (Epi::kFold) {
    static_assert(LEPI && STAG, "the folded epilogues exist in the production form only");
    fold_accumulators<BN, TN>(acc, smem + NST * STAGE, wm, wn, r16, g);
    const auto pe = epi.plain();
    if (lepi_tile_uses_registers(pe, n0, BN)) run_epilogue<TN>(pe, m0 + wm * 64 + r16, n0 + wn * (BN / 2) + 4 * g, acc);
    else {
      lds_park<BN>(smem, acc, wm, wn, r16, g);
      __syncthreads();
      lds_drain<BN>(pe, smem, m0, n0, tid);
    }
  } else if constexpr (LEPI && STAG) {
    if constexpr (std::is_same<Epi, EpiResidPS>::value) {      // acc += bias (staged behind the ring by the loaders): one add the drain need not do
      const char* xs = smem + NST * STAGE;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float4 b4 = *reinterpret_cast<const float4*>(xs + kFoldRowBytes + 4 * BN + (wn * (BN / 2) + j * 16 + 4 * g) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i][j][0] += b4.x; acc[i][j][1] += b4.y; acc[i][j][2] += b4.z; acc[i][j][3] += b4.w; }
      }
    }
    if (lepi_tile_uses_registers(epi, n0, BN)) run_epilogue<TN>(epi, m0 + wm * 64 + r16, n0 + wn * (BN / 2) + 4 * g, acc);
    else {
      lds_park<BN>(smem, acc, wm, wn, r16, g);
      __syncthreads();
      lds_drain_any<BN>(epi, smem, m0, n0, nt, tid);
    }
  } else {
    run_epilogue<TN>(epi, m0 + wm * 64 + r16, n0 + wn * (BN / 2) + 4 * g, acc);
  }
  if (ABL == 4 && g_stamps != nullptr && blockIdx.x < g_stamp_cap && lane == 0 && tid != 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    g_stamps[(size_t)blockIdx.x * kStampStride + 6 + wave] = stamp_now();
  }
  if (ABL == 4 && g_stamps != nullptr && blockIdx.x < g_stamp_cap && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // epilogue stores of this wave have been accepted
    const unsigned long long t3 = stamp_now();
    unsigned long long* o = g_stamps + (size_t)blockIdx.x * kStampStride;
    o[6] = t3;
    o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[18] = tc1; o[19] = tc2;
    unsigned int xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    o[4] = xcc; o[5] = hwid;
  }
}

#ifdef RIBCA_DIAG
// ---------------------------------------------------------------------------------------------- persistent form
// Same tile, ring, roles and stagger as gemm_ps_split_kernel, but ONE workgroup per CU walks a sequence of tiles (tile ids
// blockIdx.x, blockIdx.x + gridDim.x, ...; the XCD-aware id -> (m, n) map is unchanged, so a workgroup's tiles stay on its XCD and
// mostly share their A tile) and the K steps of consecutive tiles form one continuous stream through the ring: while the
// consumers write tile t out of their accumulator registers, the loader waves have already issued the first two K steps of
// tile t + 1, so the 2 us "first stage in flight" prologue and the 0.5-3 us workgroup-launch gap that every tile of the
// one-tile-per-workgroup kernel pays (tools/stamp_gemm.py) are hidden behind the epilogue.  The epilogue cannot park the tile
// in the ring (it is live), so it is the register form: each lane writes its 4 x TN accumulator tiles directly.
// Barrier accounting (raw s_barrier counts arrivals of all 12 waves): every wave executes 2 barriers per K step of every tile
// of this workgroup, plus one: late consumers (waves 4-7) take theirs before the first step, everyone else after the last.
template <int BN, class Epi>
__global__ __launch_bounds__(768) void gemm_ps_persist_kernel(const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ W, int ldw,
                                                              int M, int Kp, int mtiles, int ntiles, Epi epi) {
  constexpr int BM = 256, NST = 3, NLW = 4;
  constexpr int TN = BN / 32;
  constexpr int ROWS = BM + BN;
  constexpr int STAGE = ROWS * ROWB;
  constexpr int NGRP = ROWS / 8;
  constexpr int GPL = (NGRP + NLW - 1) / NLW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nblk = mtiles * ntiles;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = Kp / BK;
  const int my_tiles = (int)blockIdx.x < nblk ? (nblk - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  auto tile_origin = [&](int seq, int& m0, int& n0) {      // seq-th tile of this workgroup
    int bid = (int)blockIdx.x + seq * (int)gridDim.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int mt = bid / ntiles, nt = bid - mt * ntiles;
    m0 = mt * BM; n0 = nt * BN;
  };

  if (wave >= 8) {
    // ------------------------------------------------------------------ loader: issue iterator runs two K steps ahead
    const int lw = wave - 8;
    const uint16_t* src[GPL];
    int dst[GPL];
    auto set_tile = [&](int seq) {
      int m0, n0;
      tile_origin(seq, m0, n0);
#pragma unroll
      for (int i = 0; i < GPL; ++i) {
        int grp = lw + NLW * i;
        grp = grp < NGRP ? grp : NGRP - 1;
        const int row = grp * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ swz_f(row);
        if (row < BM) {
          int gm = m0 + row;
          gm = gm < M ? gm : M - 1;
          src[i] = A + (size_t)gm * lda + ch * 8;
        } else {
          src[i] = W + (size_t)(n0 + row - BM) * ldw + ch * 8;
        }
        dst[i] = grp * 1024;
      }
    };
    int iss_seq = 0, iss_kk = 0, iss_stage = 0;            // next (tile, K step) to issue and the ring slot it goes to
    const long long total = (long long)my_tiles * nk;
    long long issued = 0;
    auto issue_next = [&]() {
      if (issued >= total) return;
      if (iss_kk == 0) set_tile(iss_seq);
      char* st = smem + iss_stage * STAGE;
      const int ko = iss_kk * (2 * BK);
#pragma unroll
      for (int i = 0; i < GPL; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + ko),
                                         (__attribute__((address_space(3))) void*)(st + dst[i]), 16, 0, 0);
      ++issued;
      iss_stage = iss_stage + 1 == NST ? 0 : iss_stage + 1;
      if (++iss_kk == nk) { iss_kk = 0; ++iss_seq; }
    };
    issue_next();
    issue_next();
    for (long long g = 0; g < total; ++g) {
      if (issued > g + 1) wait_vmcnt<GPL>();               // one younger stage may stay in flight
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();                        // opens step g: slot (g + 2) % 3 was read in step g - 1 and is free
      asm volatile("" ::: "memory");
      issue_next();
      __builtin_amdgcn_s_barrier();                        // phase 2g + 1
    }
    __builtin_amdgcn_s_barrier();                          // the "+ 1"
    return;
  }

  // -------------------------------------------------------------------- consumer
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, g = lane >> 4;
  int a_rd[4], w_rd[TN];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_rd[i] = lds_off(wm * 64 + i * 16 + r16, 2 * g);
#pragma unroll
  for (int i = 0; i < TN; ++i) w_rd[i] = BM * ROWB + lds_off(wn * (BN / 2) + i * 16 + r16, 2 * g);
  // Barrier schedule (B0, B1, ... counted over the whole workgroup; step g of the continuous K-step stream):
  //   loaders      : B(2g)  issue(g + 2)  B(2g+1)                                   ... final B(2 total)
  //   waves 0-3    : B0 | read(g)  B(2g+1)  mfma(g)  B(2g+2)                         -> epilogue AFTER the barrier that ends the tile
  //   waves 4-7    : B0 | B(2g+1)  read(g)  B(2g+2)  mfma(g)                         -> epilogue after the tile's last mfma
  // so an early wave has already released its partner's last MFMA phase when it starts writing its tile out, exactly as in
  // the one-tile kernel, and every wave executes 2 total + 1 barriers.
  const bool late = wave >= 4;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  __builtin_amdgcn_s_barrier();                              // B0
  int cur = 0;
  for (int seq = 0; seq < my_tiles; ++seq) {
    f32x4 acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kk = 0; kk < nk; ++kk) {
      f16x8 ahi[4], alo[4], whi[TN], wlo[TN];
      if (late) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // Fragment reads as inline asm: for a plain LDS load hipcc would put s_waitcnt vmcnt(0) in front of the first MFMA of every
      // tile (the kernel contains LDS-DMA), i.e. wait for the previous tile's epilogue stores -- the very overlap this form exists for.
      const unsigned st = lds_base + (unsigned)(cur * STAGE);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(ahi[i]) : "v"(st + (unsigned)a_rd[i]) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(alo[i]) : "v"(st + (unsigned)(a_rd[i] ^ 16)) : "memory");
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(whi[j]) : "v"(st + (unsigned)w_rd[j]) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(wlo[j]) : "v"(st + (unsigned)(w_rd[j] ^ 16)) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_f16(wlo[j], ahi[i], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_f16(whi[j], alo[i], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_f16(whi[j], ahi[i], acc[i][j]);
      if (!late) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      cur = cur + 1 == NST ? 0 : cur + 1;
    }
    int m0, n0;
    tile_origin(seq, m0, n0);
    run_epilogue<TN>(epi, m0 + wm * 64 + r16, n0 + wn * (BN / 2) + 4 * g, acc);
  }
}

#endif  // RIBCA_DIAG

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

static int g_variant = 0;   // 0 = production kernel; 3/4/5/7/9 = A/B and timing-ablation forms (tools/bench_gemm.py)
void gemm_set_variant(int v) { g_variant = v; }
int gemm_set_stamp_buffer(void* dev_ptr, long long capacity_blocks) {
  const unsigned int cap = dev_ptr == nullptr || capacity_blocks <= 0 ? 0u : (unsigned int)(capacity_blocks > 0xffffffffll ? 0xffffffffll : capacity_blocks);
  if (duo_set_stamp_buffer(dev_ptr, cap) != 0) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_cap), &cap, sizeof(cap)) != hipSuccess) return 1;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dev_ptr, sizeof(dev_ptr));
}

template <int BN, class Epi, int ABL = 0, bool STAG = false, bool LEPI = false>
static void launch_split(const GemmArgs& g, const Epi& epi, hipStream_t s) {
  const int mtiles = (g.M + 255) / 256;
  const int ntiles = gemm_padded_n(g.N) / BN;
  const size_t lds = (size_t)3 * (256 + BN) * ROWB + ((Epi::kFold || std::is_same<Epi, EpiResidPS>::value) ? fold_lds_bytes<BN>() : 0);
  static unsigned long long attr_done = 0ull;
  if (!ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_ps_split_kernel<BN, Epi, ABL, STAG, LEPI>), (int)lds, attr_done)) return;
  // tile time in 10 ns ticks: K steps of ~0.65/0.87/1.15 us (BN 64/96/128) + pipeline fill + epilogue
  int deph = 0;
  static const bool env_deph = getenv("RIBCA_GEMM_DEPH") != nullptr;
  if ((g_variant == 15 || env_deph) && mtiles * ntiles >= 1024) deph = (g.Kp / BK) * (BN == 128 ? 115 : BN == 96 ? 87 : 65) + 400;
  hipLaunchKernelGGL((gemm_ps_split_kernel<BN, Epi, ABL, STAG, LEPI>), dim3(mtiles * ntiles), dim3(768), lds, s, g.A, g.lda, g.W, g.ldw, g.M, g.Kp, mtiles,
                     ntiles, epi, deph);
}

#ifdef RIBCA_DIAG
static int persist_mode() {       // RIBCA_GEMM_PERSIST=1 (or variant 30): persistent workgroups with cross-tile prefetch
  static const int m = getenv("RIBCA_GEMM_PERSIST") ? atoi(getenv("RIBCA_GEMM_PERSIST")) : 0;
  return m;
}
template <int BN, class Epi>
static void launch_persist(const GemmArgs& g, const Epi& epi, hipStream_t s) {
  const int mtiles = (g.M + 255) / 256;
  const int ntiles = gemm_padded_n(g.N) / BN;
  const size_t lds = (size_t)3 * (256 + BN) * ROWB;
  static bool attr_set = false;
  static int n_cu = 0;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ps_persist_kernel<BN, Epi>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
    attr_set = true;
  }
  const int grid = mtiles * ntiles < n_cu ? mtiles * ntiles : n_cu;
  hipLaunchKernelGGL((gemm_ps_persist_kernel<BN, Epi>), dim3(grid), dim3(768), lds, s, g.A, g.lda, g.W, g.ldw, g.M, g.Kp, mtiles, ntiles, epi);
}
#endif

// Product build: ONE kernel form per (tile width, epilogue) -- loader waves + half-step stagger + LDS epilogue -- plus the
// two-workgroups-per-CU kernel for mlp.fc1.  The A/B and timing-ablation forms of DESIGN.md section 6 (no stagger, no loads, no
// epilogue, fewer MFMA passes, stamps, persistent workgroups, duo variants 40-49) exist only in the diagnostic library
// (-DRIBCA_DIAG: `python -m multiplexed_image_annotator_amd.build --diag` -> libribca_hip_diag.so, used by tools/).
// RIBCA_DUO_BN192 (bit 0: qkv, bit 1: fc1; default 3, 0 for A/B): 128 x 192 tiles with 4 waves as 1 x 4 on the two-workgroups-per-CU
// kernel where N % 192 == 0 -- every W fragment is requested by ONE wave: 1.67 bytes from L2 per (row, column, K step) against 2.67 for
// the 2 x 2 waves of the 192 x 96 tile (qkv at D = 576, fc1 at D = 144) and the same 1.67 as the 192 x 128 tile.  Same bits (the
// accumulation order per element does not change); qkv launches 840 -> 785 ms per pass, fc1 1950 -> 1928, +0.2-0.3 % end to end in an
// interleaved same-box A/B (profiles/r3/ab_duo_128x192.txt).
static int duo_bn192() {
  static const int v = getenv("RIBCA_DUO_BN192") ? atoi(getenv("RIBCA_DUO_BN192")) : 3;
  return v;
}
template <int BN, class Epi>
static void launch_bn(const GemmArgs& g, const Epi& epi, hipStream_t s) {
  // GELU GEMMs whose weight carries a fragment-order copy (GemmArgs::WF, set by the block runner for mlp.fc1) run on the
  // two-workgroups-per-CU kernel: same bits, 4-12 % less time on those launches and 1.4 % end to end
  // (profiles/r2/duo_kernel/ab_end_to_end_fc1_on_duo_6rounds.txt).  The residual and QKV epilogues stay here: there the duo form is
  // 5-25 % slower.  RIBCA_GEMM_DUO=0: off.
  if constexpr (std::is_same<Epi, EpiGelu>::value || std::is_same<Epi, EpiGeluLn>::value) {
    static const bool duo_on = !(getenv("RIBCA_GEMM_DUO") && atoi(getenv("RIBCA_GEMM_DUO")) == 0);
    if constexpr (std::is_same<Epi, EpiGeluLn>::value) {
      if (g_variant == 0 && duo_on && (duo_bn192() & 2) && g.N % 192 == 0 && g.WF != nullptr && g.M >= 4096 && launch_duo<192, Epi>(g, epi, s, 0)) return;
    }
    if (g_variant == 0 && duo_on && g.WF != nullptr && g.M >= 4096 && launch_duo<BN, Epi>(g, epi, s, 0)) return;
  }
  // The folded qkv product too, now that V is stored row-major (round 2 measured this epilogue 5-25 % slower on the duo kernel because
  // of its eight 2-byte V^T stores per value on one wave per SIMD; with 16-byte row stores it is the GELU epilogue's shape): qkv family
  // 1939 -> 1808 ms per pass, +0.65 % end to end in an interleaved same-box A/B (profiles/r3/ab_qkv_on_duo.txt).  RIBCA_QKV_DUO=0: off.
  if constexpr (std::is_same<Epi, EpiQKVLn>::value) {
    static const bool qkv_duo = !(getenv("RIBCA_QKV_DUO") && atoi(getenv("RIBCA_QKV_DUO")) == 0);
    if (g_variant == 0 && qkv_duo && (duo_bn192() & 1) && g.N % 192 == 0 && g.WF != nullptr && g.M >= 4096 && epi.vrow &&
        launch_duo<192, Epi>(g, epi, s, 0))
      return;
    if (g_variant == 0 && qkv_duo && g.WF != nullptr && g.M >= 4096 && epi.vrow && launch_duo<BN, Epi>(g, epi, s, 0)) return;
  }
#ifdef RIBCA_DIAG
  if constexpr (!std::is_same<Epi, EpiResidPS>::value) {
    if (g_variant >= 40 && g_variant <= 49) {   // two workgroups per CU (gemm_duo.hip); 41-47 = its timing ablations (bit mask), 48 = stamps
      if (launch_duo<BN, Epi>(g, epi, s, g_variant - 40)) return;   // 41 = no epilogue, 48 = stamps, 49 = both
    }
  }
  if constexpr (std::is_same<Epi, EpiResidPS>::value || Epi::kFold) {      // these epilogues exist in the LDS-drain forms only
    switch (g_variant) {
      case 21: launch_split<BN, Epi, 6, true, true>(g, epi, s); return;
      case 22: launch_split<BN, Epi, 7, true, true>(g, epi, s); return;
      case 12: launch_split<BN, Epi, 4, true, true>(g, epi, s); return;
      case 9: launch_split<BN, Epi, 3, true, true>(g, epi, s); return;
      default: break;
    }
  } else {
    if ((g_variant == 30 || (g_variant == 0 && persist_mode())) && (g.M + 255) / 256 * (gemm_padded_n(g.N) / BN) >= 512) {
      launch_persist<BN, Epi>(g, epi, s);
      return;
    }
    switch (g_variant) {
      case 3: launch_split<BN, Epi, 0, false>(g, epi, s); return;   // no stagger (A/B reference)
      case 4: launch_split<BN, Epi, 1, false>(g, epi, s); return;   // ablation: no loads
      case 5: launch_split<BN, Epi, 2, false>(g, epi, s); return;   // ablation: loads only
      case 7: launch_split<BN, Epi, 1, true>(g, epi, s); return;    // ablation: no loads, staggered
      case 9: launch_split<BN, Epi, 3, true>(g, epi, s); return;    // ablation: no epilogue
      case 20: launch_split<BN, Epi, 5, true>(g, epi, s); return;   // ablation: no loads, no epilogue (the K-loop structure alone)
      case 21: launch_split<BN, Epi, 6, true, true>(g, epi, s); return;   // ablation: 2 MFMAs per operand pair
      case 22: launch_split<BN, Epi, 7, true, true>(g, epi, s); return;   // ablation: 1 MFMA per operand pair
      case 23: launch_split<BN, Epi, 8, true>(g, epi, s); return;   // ablation: 2 MFMAs, no epilogue
      case 24: launch_split<BN, Epi, 9, true>(g, epi, s); return;   // ablation: 1 MFMA, no epilogue
      case 12: launch_split<BN, Epi, 4, true, true>(g, epi, s); return;   // production kernel + diagnostic time stamps
      case 14: launch_split<BN, Epi, 0, true>(g, epi, s); return;   // A/B: epilogue straight from the accumulator registers
      default: break;
    }
  }
#endif
  launch_split<BN, Epi, 0, true, true>(g, epi, s);   // production: loader waves + half-step stagger + LDS epilogue
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

// RIBCA_NT bit 0: fc1 (GELU) output, bit 1: Q / K rows, bit 2: residual z  -- non-temporal epilogue stores (A/B switch)
static int nt_mask() {
  static const int m = getenv("RIBCA_NT") ? atoi(getenv("RIBCA_NT")) : 0;
  return m;
}
void launch_gemm_resid(const GemmArgs& g, float* z, int ldz, hipStream_t s) {
  static const int no_touch = (getenv("RIBCA_GEMM_TOUCH") && atoi(getenv("RIBCA_GEMM_TOUCH")) == 0) ? 2 : 0;
  launch_any(g, EpiResid{z, ldz, g.bias, g.M, g.N, ((nt_mask() >> 2) & 1) | no_touch}, s);
}
void launch_gemm_gelu(const GemmArgs& g, uint16_t* out, int ldo, hipStream_t s) {
  launch_any(g, EpiGelu{out, ldo, g.bias, g.M, g.N, nt_mask() & 1}, s);
}
// Tile width of the two-workgroups-per-CU form a residual GEMM would take (gemm_duo.hip launch_duo), and the columns of a wave's block
// in it: 4 waves as 1 x 4 for 64- / 128- / 192-wide tiles, 2 x 2 for 96-wide ones.
// RIBCA_RESID_DUO192 (default 1; 0 for A/B): where N % 192 == 0 the 128 x 192 tile (every W fragment requested once: 1.67 bytes from L2
// per unit instead of 2.67) replaces the 192 x 96 one, and with it the long K loops pay too (fc2 at D = 384 / 576).
static bool resid_duo192() {
  static const bool on = !(getenv("RIBCA_RESID_DUO192") && atoi(getenv("RIBCA_RESID_DUO192")) == 0);
  return on;
}
static int resid_duo_bn(int N) { return (resid_duo192() && N % 192 == 0) ? 192 : gemm_pick_bn(N); }
static int resid_duo_block(int N) {
  const int bn = resid_duo_bn(N);
  return (bn == 96 || bn == 192) ? 48 : bn / 4;
}
// Where the two-workgroups-per-CU form pays (profiles/r3/resid_through_ring_bench.txt, M = 103 424, cache-cold operands, row statistics
// included on both sides): the narrow classifier (N = 144: proj -15 %, fc2 -10 %) and the short-K products of the 96-wide tiles (proj at
// D = 288 / 576: -2 % with the test hook's weight repack inside the figure).  Long K loops lose on the 192 x 96 tile (fc2 at D = 288
// +11 %, D = 576 +4 %): with 4 waves as 2 x 2 every W fragment is requested by two waves, 2.7 bytes from L2 per (row, column, K step)
// against 1.8 for the 256 x 96 tile of the one-workgroup kernel, and the K loop runs into the CU's L2 fetch rate (61-70 GB/s, DESIGN.md
// section 6.2) before the load-free epilogue can matter; the 192 x 128 form (1 x 4 waves) ties (+1 ... +5 %).
// The 128 x 192 tile (1.67 bytes per unit) pays wherever the one-workgroup kernel would run its 96-wide tile (1.83), long K loops
// included -- proj at D = 576 -12 %, fc2 at D = 576 -7 % -- and loses against the 128-wide tile (1.50): D = 384 +1 ... +10 %
// (profiles/r3/resid_through_ring_bench_128x192.txt).
static bool resid_duo_pays(int N, int Kp) {
  if (N <= 192) return true;
  if (resid_duo_bn(N) == 192) return gemm_pick_bn(N) == 96;
  return gemm_pick_bn(N) == 96 && Kp <= (N + 31) / 32 * 32;
}
ResidStatGeom launch_gemm_resid_ps(const GemmArgs& g, uint16_t* z, int ldz, float2* part, const float2* prev, int prev_stride, hipStream_t s,
                                   bool force_duo, const MxAct* zmx) {
  static const int no_touch = (getenv("RIBCA_GEMM_TOUCH") && atoi(getenv("RIBCA_GEMM_TOUCH")) == 0) ? 2 : 0;
  // proj / fc2 of the classifiers' full blocks: two workgroups per CU, residual through the ring, load-free epilogue (EpiResidZK).
  // RIBCA_RESID_DUO = 0: never, 1 (default): the shapes where it measured faster, 2: every shape it supports (A/B)
  static const int duo_mode = getenv("RIBCA_RESID_DUO") ? atoi(getenv("RIBCA_RESID_DUO")) : 1;
  const int blk = resid_duo_block(g.N);
  const bool want = force_duo || duo_mode == 2 || (duo_mode == 1 && resid_duo_pays(g.N, g.Kp));
  // (No threshold on M: this form is not bit-identical to the one-workgroup kernel -- the residual is added inside the accumulation,
  // the statistics are combined per wave block -- and a cell's result must not depend on the size of the chunk it was computed in.)
  // zmx: the new rows also in the MX3 format (gemm_mx.hip) -- the 128 x 192 tile of the two-workgroups form emits it; the caller asks for
  // it only where that form exists (N % 192 == 0, a fragment-order weight), and the choice is the model's, never the chunk's
  if (zmx != nullptr) {
    if (g.WF == nullptr || g.N % 192 != 0 || g_variant != 0) {
      launch_error("launch_gemm_resid_ps: an MX3 copy of the residual rows needs the 128 x 192 form (N %% 192 == 0, a fragment-order weight, "
                   "gemm variant 0): N = %d, WF %s, variant %d", g.N, g.WF ? "given" : "missing", g_variant);
      return ResidStatGeom{g.N / 48, 48};
    }
    EpiResidZK epi{z, ldz, g.bias, g.M, g.N, part, prev, prev_stride};
    epi.zmx = *zmx;
    if (!launch_duo<192, EpiResidZK>(g, epi, s, 0)) launch_error("launch_gemm_resid_ps: the 128 x 192 two-workgroups form refused N = %d, Kp = %d", g.N, g.Kp);
    return ResidStatGeom{g.N / 48, 48};
  }
  if (want && g_variant == 0 && g.WF != nullptr && g.N % blk == 0 && g.N % 8 == 0) {
    const EpiResidZK epi{z, ldz, g.bias, g.M, g.N, part, prev, prev_stride};
    bool done = false;
    switch (resid_duo_bn(g.N)) {
      case 192: done = launch_duo<192, EpiResidZK>(g, epi, s, 0); break;
      case 128: done = launch_duo<128, EpiResidZK>(g, epi, s, 0); break;
      case 64: done = launch_duo<64, EpiResidZK>(g, epi, s, 0); break;
      default: done = launch_duo<96, EpiResidZK>(g, epi, s, 0); break;
    }
    if (done) return ResidStatGeom{g.N / blk, blk};
  }
  launch_any(g, EpiResidPS{z, ldz, g.bias, g.M, g.N, no_touch, part, prev, prev_stride}, s);
  return ResidStatGeom{gemm_resid_tiles(g.N), gemm_resid_bn(g.N)};
}
int gemm_resid_part_rows(int N) {
  const int blk = resid_duo_block(N);
  const int fine = (N + blk - 1) / blk, coarse = gemm_padded_n(N) / gemm_pick_bn(N);
  return fine > coarse ? fine : coarse;
}
void launch_gemm_gelu_ln(const GemmArgs& g, const float2* rowstat, const float* csum, uint16_t* out, int ldo, hipStream_t s) {
  launch_any(g, EpiGeluLn{out, ldo, g.bias, g.M, g.N, nt_mask() & 1, rowstat, csum}, s);
}
void launch_gemm_qkv_ln(const GemmArgs& g, const float2* rowstat, const float* csum, uint16_t* q, uint16_t* k, uint16_t* vt, const AttnGeom& a,
                        float scale, hipStream_t s, int n_off, int cls_rows, int rs_stride) {
  launch_any(g, EpiQKVLn{q, k, vt, g.bias, a.D, a.hd, a.hdq, a.hdv, scale, g.M, g.N, a.T, a.TP, a.H, a.KP, (nt_mask() >> 1) & 1,
                         (unsigned)((0x100000000ull + (unsigned long long)a.T - 1) / (unsigned long long)a.T), rowstat, csum, attention_v_rowmajor(a) ? 1 : 0,
                         n_off, cls_rows, rs_stride}, s);
}
int gemm_resid_tiles(int N) { return gemm_padded_n(N) / gemm_pick_bn(N); }
int gemm_resid_bn(int N) { return gemm_pick_bn(N); }
void launch_gemm_qkv(const GemmArgs& g, uint16_t* q, uint16_t* k, uint16_t* vt, const AttnGeom& a, float scale, hipStream_t s) {
  launch_any(g, EpiQKV{q, k, vt, g.bias, a.D, a.hd, a.hdq /* Q/K row pitch: compact */, a.hdv, scale, g.M, g.N, a.T, a.TP, a.H, a.KP, (nt_mask() >> 1) & 1,
                       (unsigned)((0x100000000ull + (unsigned long long)a.T - 1) / (unsigned long long)a.T), nullptr, nullptr, attention_v_rowmajor(a) ? 1 : 0}, s);
}
void launch_gemm_rowmap(const GemmArgs& g, float* out, int ldo, const float* add, int ldadd, const int* slot, const int* addrow, int R,
                        int dst_per_cell, hipStream_t s) {
  launch_any(g, EpiRowMap{out, ldo, g.bias, add, ldadd, slot, addrow, R, dst_per_cell, g.M, g.N}, s);
}

}  // namespace ribca
