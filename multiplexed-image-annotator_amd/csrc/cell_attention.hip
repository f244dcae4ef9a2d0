// Per-cell fusion of  norm1 -> attn.qkv -> softmax(q k^T) v  for the classifiers with head dims <= 24 (D = 144: hd 12, D = 288: hd 24;
// timm Block.norm1 / Attention reached from reference cell_type_annotation/model.py:54-55, 402).
//
// Why: per row and layer the unfused pair writes Q, K, V (12 D bytes) from the qkv GEMM and reads them back in the attention kernel
// -- 24 D of the 88 D bytes a block moves -- and the qkv GEMM streams BOTH operands through L2 -> LDS (57 bytes per kMAC at 256 x 96
// tiles).  Here one workgroup owns one cell: its 101 token rows of the packed-split residual stream are the MFMA operand of seven waves
// for the whole kernel (registers: 16 rows x D each), only the weight streams (36 bytes per kMAC), and q, k, v never leave the CU.
//
// * grid = cells, 512 threads: waves 0-6 own tokens 16 w .. 16 w + 15 (rows >= 101 are clamped copies whose results are dropped),
//   wave 7 only issues the direct-to-LDS weight loads (3-stage ring, one 32-deep K step of 144 weight rows = 18 KB per stage,
//   counted vmcnt, one s_barrier per step, XOR-swizzled 128-byte rows exactly as gemm_split16.hip).
// * heads are processed in GROUPS of 48 feature dims (2 heads of 24 or 4 heads of 12 = three 16-column MFMA tiles each for q, k
//   and v): per group a 9-tile x K = D product per wave (transposed tiles: a lane holds 4 consecutive output columns of its token),
//   then the folded LayerNorm  x = rstd acc + (-mean rstd c + b')  (gemm_epi.h), then attention for the group's heads.
// * q stays in the registers of the wave that owns the queries; k is published to LDS in MFMA FRAGMENT order (a common permutation
//   of the head's dims on q and k leaves q.k unchanged, so the accumulator layout IS the operand layout: lane (token, g) contributes
//   columns 4g .. 4g+3 of the one or two 16-column tiles the head overlaps, the rest of the 32-deep K block is zero); v is published
//   row-major (packed-split rows of 48 dims) and transposed on the way out of LDS with ds_read_b64_tr_b16, as attention.hip does.
// * S^T = K Q^T per 16-key tile, softmax over keys in registers + two xor-shuffles, P in place as the next operand, O^T = V^T P^T:
//   the arithmetic and its order per (query, key, dim) are those of gemm + attention.hip, so results agree to rounding with the
//   unfused path (tests/test_gpu_kernels.py::test_cell_attention_fused compares with fp64 and with the unfused kernels).
#include <cstdlib>

#include "gemm_epi.h"
#include "ribca_common.h"
#include "ribca_kernels.h"

namespace ribca {

namespace {

typedef __attribute__((__vector_size__(4 * sizeof(_Float16)))) _Float16 f16x4c;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4c;
__device__ __forceinline__ f16x4c lds_read_tr16c(const char* p) {
  return __builtin_bit_cast(f16x4c, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4c __attribute__((address_space(3)))*)(p)));
}

constexpr int kTP = 112, kVRows = 128;

template <int D, int HD> struct CellGeom {
  static constexpr int Dp = (D + 31) / 32 * 32;
  static constexpr int NK = Dp / 32;
  static constexpr int GD = HD == 32 ? 64 : 48;  // feature dims per head group: two heads of 24 or 32, four of 12 -> 3 or 4 MFMA tiles each of q, k, v
  static constexpr int NTP = GD / 16;          // 16-column tiles per part (q, k, v)
  static constexpr int WROWS = 3 * GD;         // weight rows per ring stage
  static constexpr int WSTAGE = WROWS * ROWB;  // 18 or 24 KB
  static constexpr int GROUPS = D / GD;
  static constexpr int HPG = GD / HD;          // heads per group
  static constexpr int KIMG = kTP * ROWB;      // one head's K image: 112 tokens x 128 B (fragment order, swizzled like a GEMM tile)
  static constexpr int VROWB = 4 * GD;         // 192 / 256 B: the group's dims packed-split
  static constexpr int VIMG = kVRows * VROWB + 64;
  static constexpr int CB = 2 * 3 * D * 4;     // column sums and folded bias of the whole qkv product
  static constexpr int RING = 3;               // stage s + 1 is written while stage s is read; the slot of s - 1 is free by then
  static constexpr int OFF_K = RING * WSTAGE;
  static constexpr int OFF_V = OFF_K + HPG * KIMG;
  static constexpr int OFF_CB = OFF_V + (VIMG + 15) / 16 * 16;
  static constexpr int LDS = OFF_CB + CB;
  static_assert(D % GD == 0 && GD % HD == 0 && HD % 4 == 0 && HD <= 32 && HPG % 2 == 0 && WROWS % 8 == 0, "head geometry");
};

}  // namespace

template <int D, int HD>
__global__ __launch_bounds__(512) void cell_qkv_attention_kernel(const uint16_t* __restrict__ z, int ldz, const uint16_t* __restrict__ W, int ldw,
                                                                 const float* __restrict__ bias2, const float* __restrict__ csum,
                                                                 const float2* __restrict__ rowstat, uint16_t* __restrict__ out, int ldo, int T,
                                                                 float scale, int dbg_arg) {
  // timing ablations exist in the diagnostic library only (RIBCA_CELL_DBG, tools/bench_cell_attention.py); the product kernel has none
#ifdef RIBCA_DIAG
  const int dbg = dbg_arg;
#else
  constexpr int dbg = 0;
  (void)dbg_arg;
#endif
  using G = CellGeom<D, HD>;
  constexpr int NK = G::NK, GROUPS = G::GROUPS, HPG = G::HPG, kRing = G::RING, kGroupDims = G::GD, NTP = G::NTP, kWStage = G::WSTAGE;
  constexpr int NTILES = 3 * NTP;              // accumulator tiles per wave and group: q | k | v
  constexpr int TOTAL = GROUPS * NK;           // K steps of the whole cell
#ifdef CELLDBG_MXBOUND      // timing-only bound (results wrong on purpose; tools/build_ab_lib.py ... -DCELLDBG_MXBOUND): what the weight stream in an
  // MX image (fp16 hi + fp6 lo: ~2/3 of the bytes) and one correction on the block-scaled instruction could buy the QKV phase at most --
  // 2/3 of the DMA pieces per stage, no lo fragment reads, 2 of the 3 MFMAs per tile and step
  constexpr int GPL = G::WROWS / 8 * 2 / 3;
#else
  constexpr int GPL = G::WROWS / 8;            // 18 / 24 DMA instructions (1 KB each) per stage
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cell = blockIdx.x;
  const int r16 = lane & 15, g = lane >> 4;

  // ---- once per cell: column sums / folded bias of all 3 D outputs, zero rows behind V
  for (int i = tid; i < 3 * D / 4; i += 512) {
    reinterpret_cast<float4*>(smem + G::OFF_CB)[i] = reinterpret_cast<const float4*>(csum)[i];
    reinterpret_cast<float4*>(smem + G::OFF_CB + 3 * D * 4)[i] = reinterpret_cast<const float4*>(bias2)[i];
  }
  for (int o = kTP * G::VROWB + tid * 16; o < G::VIMG; o += 512 * 16) *reinterpret_cast<uint4*>(smem + G::OFF_V + o) = uint4{0u, 0u, 0u, 0u};
  __syncthreads();

  if (wave == 7) {
    // ------------------------------------------------------------------ loader: the weight stream of all head groups, two steps ahead.
    // (ONE wave issues an LDS-DMA instruction every ~67 cycles = 30 GB/s, and that stream is the kernel's floor: 131 of 300 us at
    // D = 288 with neither MFMAs nor attention, tools/bench_cell_attention.py.  A deeper ring does not help -- issue-bound, not
    // latency-bound -- and a register-staged loader (global_load_dwordx4 + ds_write_b128, one or two stages in flight) was 2.4-5x
    // slower: profiles/r3/cell_attention_loader_experiments.txt.  The GEMMs use four loader waves for this reason.)
    // Later experiment, NOT kept: the same stream through one buffer descriptor with scalar / precomputed offsets (4 instructions per DMA
    // instead of a dependent 64-bit address chain, 40 GB/s from this one wave) bought 0.15 % end to end and was NOT deterministic at
    // 18 DMAs per stage (D = 144, 288): a few rows per thousand cells differed by up to 5e-4 between identical launches
    // (tools/check_determinism.py; tests/test_gpu_e2e.py::test_config3_full_size_properties caught it).  Cause (round 4, from the ISA): the
    // CONSUMERS left fragment reads of the slot in flight across the step barrier -- see the wait in front of it below; with that wait the
    // protocol no longer depends on how soon behind the barrier the first piece is issued.
    auto issue = [&](int step) {
      const int hg = step / NK, s = step - hg * NK;
      char* st = smem + (step % kRing) * kWStage;
#pragma unroll
      for (int i = 0; i < GPL; ++i) {
        const int row = i * 8 + (lane >> 3);                   // [q GD | k GD | v GD] rows of this group
        const int which = row / kGroupDims, c = row - which * kGroupDims;
        const int n = which * D + hg * kGroupDims + c;
        const int ch = (lane & 7) ^ swz_f(row);
        const uint16_t* src = W + (size_t)n * ldw + s * (2 * BK) + ch * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(st + i * 1024), 16,
                                         0, 0);
      }
    };
    issue(0);
    issue(1);
    for (int step = 0; step < TOTAL; ++step) {
#ifdef CELLDBG_HALFBAR      // timing-only (results racy on purpose): what half the step barriers (64-deep ring stages) would buy
      if ((step & 1) == 0 || step + 1 == TOTAL) {
#endif
      if (step + 1 < TOTAL) wait_vmcnt<GPL>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();                            // stage `step` landed; the slot of step - 1 has been read by everyone
      asm volatile("" ::: "memory");
#ifdef CELLDBG_HALFBAR
      }
#endif
      if (step + 2 < TOTAL) issue(step + 2);
      if ((step + 1) % NK == 0) __builtin_amdgcn_s_barrier();  // the group's "k, v published" barrier
    }
    return;
  }

  // -------------------------------------------------------------------- consumers: wave w owns tokens 16 w .. 16 w + 15
  const int tok = wave * 16 + r16;
  const int trow = tok < T ? tok : T - 1;                      // pad rows compute on a copy of the last token; nothing of theirs is stored
  const uint16_t* zr = z + ((size_t)cell * T + trow) * ldz;
  f16x8 ahi[NK], alo[NK];
#pragma unroll
  for (int s = 0; s < NK; ++s) {
    const uint4* p = reinterpret_cast<const uint4*>(zr + (4 * s + g) * 16);
    ahi[s] = __builtin_bit_cast(f16x8, p[0]);
    alo[s] = __builtin_bit_cast(f16x8, p[1]);
  }
  const float2 rs = rowstat[(size_t)cell * T + trow];
  const float rstd = rs.x, nm = -rs.y * rs.x;
  const char* cb = smem + G::OFF_CB;
  uint16_t* orow = out + ((size_t)cell * T + tok) * ldo;
  int w_rd[NTILES];
#pragma unroll
  for (int j = 0; j < NTILES; ++j) w_rd[j] = lds_off(16 * j + r16, 2 * g);
  const int k_wr = lds_off(tok, 2 * g);                        // this lane's 32 bytes (hi | lo) of a K image row
  constexpr int KST = 4, NT = 7;

  int step = 0;
  for (int hg = 0; hg < GROUPS; ++hg) {
    f32x4 acc[NTILES];
#pragma unroll
    for (int j = 0; j < NTILES; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NK; ++s, ++step) {
      // Every fragment read of the PREVIOUS stage must have RETURNED before this barrier: behind it the loader refills that very slot
      // (ring of 3, two stages ahead).  Without the wait hipcc software-pipelines the loop -- the last ds_read_b128s of a stage are issued
      // in front of the barrier and waited for behind it, their MFMAs sunk below it (11 of the 19 barriers of the D = 384 kernel, found
      // in the ISA) -- and the slot is only protected by the DMA's latency exceeding an LDS read's.  That is the non-repeatability the
      // round-3 buffer-descriptor loader showed (a few rows per thousand cells at D = 144 / 288): it issues its first piece a few
      // cycles behind the barrier instead of ~70 and so closed the window.
#ifdef CELLDBG_HALFBAR
      if ((step & 1) == 0 || step + 1 == TOTAL) {
#endif
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef CELLDBG_HALFBAR
      }
#endif
      const char* st = smem + (step % kRing) * kWStage;
      if (dbg & 2) continue;                                   // timing ablation: barriers and the weight stream only
#pragma unroll
      for (int jb = 0; jb < NTILES / 3; ++jb) {
        f16x8 whi[3], wlo[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          whi[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(st + w_rd[3 * jb + j]));
#ifndef CELLDBG_MXBOUND
          wlo[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(st + (w_rd[3 * jb + j] ^ 16)));
#else
          wlo[j] = whi[j];
#endif
        }
#ifndef CELLDBG_MXBOUND
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[3 * jb + j] = mfma_f16(wlo[j], ahi[s], acc[3 * jb + j]);
#endif
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[3 * jb + j] = mfma_f16(whi[j], alo[s], acc[3 * jb + j]);
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[3 * jb + j] = mfma_f16(whi[j], ahi[s], acc[3 * jb + j]);
      }
    }
    // ---- folded LayerNorm + bias: tile j of part `which` holds output columns which D + 48 hg + 16 (j % 3) + 4 g .. + 3 of this token
#pragma unroll
    for (int j = 0; j < NTILES; ++j) {
      const int n = (j / NTP) * D + hg * kGroupDims + 16 * (j % NTP) + 4 * g;
      const float4 c4 = *reinterpret_cast<const float4*>(cb + n * 4), b4 = *reinterpret_cast<const float4*>(cb + 3 * D * 4 + n * 4);
      // two columns per instruction (v_pk_fma_f32); the same two fused multiply-adds per value as ln_fold4 (gemm_epi.h)
      const f32x2v r2 = {rstd, rstd}, n2 = {nm, nm};
      const f32x2v lo2 = fma2(r2, f32x2v{acc[j][0], acc[j][1]}, fma2(n2, f32x2v{c4.x, c4.y}, f32x2v{b4.x, b4.y}));
      const f32x2v hi2 = fma2(r2, f32x2v{acc[j][2], acc[j][3]}, fma2(n2, f32x2v{c4.z, c4.w}, f32x2v{b4.z, b4.w}));
      acc[j] = f32x4{lo2.x, lo2.y, hi2.x, hi2.y};
    }
    // ---- publish k (fragment order, one image per head) and v (row-major packed-split rows of 48 dims)
    f16x8 qhi[HPG], qlo[HPG];
#pragma unroll
    for (int h = 0; h < HPG; ++h) {
      constexpr int dummy = 0; (void)dummy;
      const int d_lo = h * HD, d_hi = d_lo + HD;               // dims of this head inside the group
      const int tA = d_lo / 16, tB = (d_hi - 1) / 16;
      // a lane's 4 columns of a tile lie wholly inside or outside the head (HD % 4 == 0)
      const bool inA = 16 * tA + 4 * g >= d_lo && 16 * tA + 4 * g < d_hi;
      const bool inB = tB != tA && 16 * tB + 4 * g >= d_lo && 16 * tB + 4 * g < d_hi;
      float qa[4], qb[4], ka[4], kb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        qa[r] = inA ? acc[tA][r] * scale : 0.f;
        qb[r] = inB ? acc[tB][r] * scale : 0.f;
        ka[r] = inA ? acc[NTP + tA][r] : 0.f;
        kb[r] = inB ? acc[NTP + tB][r] : 0.f;
      }
      uint2 h0, l0, h1, l1;
      split4(qa, h0, l0);
      split4(qb, h1, l1);
      qhi[h] = __builtin_bit_cast(f16x8, uint4{h0.x, h0.y, h1.x, h1.y});
      qlo[h] = __builtin_bit_cast(f16x8, uint4{l0.x, l0.y, l1.x, l1.y});
      split4(ka, h0, l0);
      split4(kb, h1, l1);
      char* kimg = smem + G::OFF_K + h * G::KIMG;
      *reinterpret_cast<uint4*>(kimg + k_wr) = uint4{h0.x, h0.y, h1.x, h1.y};
      *reinterpret_cast<uint4*>(kimg + (k_wr ^ 16)) = uint4{l0.x, l0.y, l1.x, l1.y};
    }
    {
      uint16_t* vrow = reinterpret_cast<uint16_t*>(smem + G::OFF_V + tok * G::VROWB);
#pragma unroll
      for (int j = 0; j < NTP; ++j) {
        const float v4[4] = {acc[2 * NTP + j][0], acc[2 * NTP + j][1], acc[2 * NTP + j][2], acc[2 * NTP + j][3]};
        // 256-byte V rows (64-dim groups) would put the 8 rows of a transposed read on the same banks: the 16-dim pair index is
        // XORed with the low two bits of the row on both sides (write here, read below)
        const int jj = kGroupDims == 64 ? (j ^ (tok & 3)) : j;
        ps_store4_pair<16>(vrow, 16 * jj + 4 * g, v4);         // lanes g and g ^ 1 complete an 8-dim group: 16 bytes each
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // k, v of every token are in LDS
    asm volatile("" ::: "memory");

    if (dbg & 1) continue;                                     // timing ablation (RIBCA_CELL_DBG=1): no attention phase, results wrong
    // ---- attention of the group's heads for this wave's 16 queries, TWO heads at a time: the S^T MFMAs of one head cover the softmax
    // VALU of the other, and the two P V products interleave (a wave is in-order: without a second independent chain every
    // exp / split waits behind the MFMAs that feed it)
    constexpr int DT = (HD + 15) / 16;
    const char* vimg = smem + G::OFF_V;
#pragma unroll
    for (int hp = 0; hp < HPG; hp += 2) {
      f32x4 s[2][2 * KST];
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        const int ko = lds_off(16 * kt + r16, 2 * g);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const char* kimg = smem + G::OFF_K + (hp + e) * G::KIMG;
          const f16x8 kh = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(kimg + ko));
          const f16x8 kl = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(kimg + (ko ^ 16)));
          s[e][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
          s[e][kt] = mfma_f16(kl, qhi[hp + e], s[e][kt]);
          s[e][kt] = mfma_f16(kh, qlo[hp + e], s[e][kt]);
          s[e][kt] = mfma_f16(kh, qhi[hp + e], s[e][kt]);
        }
      }
      f16x8 phi[2][KST], plo[2][KST];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * (NT - 1) + 4 * g + r >= T) s[e][NT - 1][r] = -INFINITY;
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[e][kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // exp(s - mx) = exp2(s log2e - mx log2e): ONE packed fma per two scores in front of v_exp_f32 (a subtract and a multiply per
        // score before); probabilities are normalised two at a time and split without the fp16 range clamp (they lie in [0, 1])
        const f32x2v l2 = {1.44269504089f, 1.44269504089f}, moff = {-mx * 1.44269504089f, -mx * 1.44269504089f};
        f32x2v sum2 = {0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          const f32x2v a0 = fma2(f32x2v{s[e][kt][0], s[e][kt][1]}, l2, moff);
          const f32x2v a1 = fma2(f32x2v{s[e][kt][2], s[e][kt][3]}, l2, moff);
          const f32x2v e0 = {__builtin_amdgcn_exp2f(a0.x), __builtin_amdgcn_exp2f(a0.y)}, e1 = {__builtin_amdgcn_exp2f(a1.x), __builtin_amdgcn_exp2f(a1.y)};
          s[e][kt] = f32x4{e0.x, e0.y, e1.x, e1.y};
          sum2 += e0;
          sum2 += e1;
        }
        float sum = sum2.x + sum2.y;
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        const f32x2v inv2 = {inv, inv};
        s[e][NT] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KST; ++t) {
          const f32x2v pa0 = f32x2v{s[e][2 * t][0], s[e][2 * t][1]} * inv2, pa1 = f32x2v{s[e][2 * t][2], s[e][2 * t][3]} * inv2;
          const f32x2v pb0 = f32x2v{s[e][2 * t + 1][0], s[e][2 * t + 1][1]} * inv2, pb1 = f32x2v{s[e][2 * t + 1][2], s[e][2 * t + 1][3]} * inv2;
          const float pa[4] = {pa0.x, pa0.y, pa1.x, pa1.y}, pb[4] = {pb0.x, pb0.y, pb1.x, pb1.y};
          uint2 ha, la, hb, lb;
          split4_unit(pa, ha, la);
          split4_unit(pb, hb, lb);
          phi[e][t] = __builtin_bit_cast(f16x8, uint4{ha.x, ha.y, hb.x, hb.y});
          plo[e][t] = __builtin_bit_cast(f16x8, uint4{la.x, la.y, lb.x, lb.y});
        }
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int t = 0; t < KST; ++t) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int dcol0 = (hp + e) * HD + 16 * dt + 4 * (r16 & 3);    // first of the 4 dims this lane addresses for the transposed read
            const int dcol = kGroupDims == 64 ? ((((dcol0 >> 4) ^ (r16 >> 2)) << 4) | (dcol0 & 15)) : dcol0;      // (row & 3 == r16 >> 2)
            const int voff = (dcol >> 3) * 32 + (dcol & 7) * 2;           // byte offset of their hi halves inside a V row; + 16: lo halves
            const char* va = vimg + (32 * t + 4 * g + (r16 >> 2)) * G::VROWB + voff;
            const f16x4c h0 = lds_read_tr16c(va), l0 = lds_read_tr16c(va + 16);
            const f16x4c h1 = lds_read_tr16c(va + 16 * G::VROWB), l1 = lds_read_tr16c(va + 16 * G::VROWB + 16);
            const f16x8 vhi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            const f16x8 vlo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            o[e] = mfma_f16(vlo, phi[e][t], o[e]);
            o[e] = mfma_f16(vhi, plo[e][t], o[e]);
            o[e] = mfma_f16(vhi, phi[e][t], o[e]);
          }
        }
        const int d = 16 * dt + 4 * g;                         // o[r] = O[query tok][head dim d + r]
        if (tok < T && d < HD) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float v4[4] = {o[e][0], o[e][1], o[e][2], o[e][3]};
            ps_store4(orow, (hg * HPG + hp + e) * HD + d, v4);
          }
        }
      }
    }
  }
}

bool cell_attention_supported(int D, int H, int T) { return H == kHeads && T == kTokens && (D == 144 || D == 288 || D == 384); }

void launch_cell_qkv_attention(const uint16_t* z, int ldz, const uint16_t* W, int ldw, const float* bias2, const float* csum, const float2* rowstat,
                               uint16_t* out, int ldo, int cells, int D, float scale, hipStream_t s) {
  if (cells <= 0) return;
  auto go = [&](auto kern, int lds) {
    static unsigned long long attr_done[3] = {0ull, 0ull, 0ull};      // per width (one kernel instantiation each) and device
    const int slot = D == 288 ? 1 : D == 384 ? 2 : 0;
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_done[slot])) return;
#ifdef RIBCA_DIAG
    static const int dbg = getenv("RIBCA_CELL_DBG") ? atoi(getenv("RIBCA_CELL_DBG")) : 0;      // timing ablations of tools/bench_cell_attention.py
#else
    const int dbg = 0;
#endif
    hipLaunchKernelGGL(kern, dim3(cells), dim3(512), lds, s, z, ldz, W, ldw, bias2, csum, rowstat, out, ldo, (int)kTokens, scale, dbg);
  };
  if (D == 288) go(cell_qkv_attention_kernel<288, 24>, CellGeom<288, 24>::LDS);
  else if (D == 384) go(cell_qkv_attention_kernel<384, 32>, CellGeom<384, 32>::LDS);
  else go(cell_qkv_attention_kernel<144, 12>, CellGeom<144, 12>::LDS);
}

}  // namespace ribca
