// "Duo" form of the fp16x3 GEMM: TWO 256-thread workgroups per CU, each a whole 256 x BN tile (same numerics, operand formats and
// epilogue functors as gemm_split16.hip; reached from reference cell_type_annotation/model.py:402 ``model(x)`` through the timm
// Linear layers of every Block).
//
// Why: round-2 counters and ablations show the one-workgroup-per-CU kernel spending a third of a launch in
// phases in which the matrix pipe is idle by construction -- the tile's epilogue (park, GELU / residual, 128 KB of stores), the
// next workgroup's start and its first ring stage in flight -- because a 144 KB ring leaves no LDS for a second workgroup.  Here
// the LDS footprint of a tile is cut to 64 KB so that two independent workgroups share a CU: while one writes its tile out or
// waits for its first operands, the other owns the matrix pipe; the hardware scheduler does the overlap, no persistent-kernel
// hand-offs.
//
// * A (activations, 256 rows x 128 B per 32-deep K step) goes through a 2-stage LDS ring filled by global_load_lds_dwordx4, issued
//   by the four waves themselves (8 per wave and step), XOR-swizzled exactly as in gemm_split16.hip.
// * W never touches LDS: the weights are static, so ribca keeps a second copy in FRAGMENT ORDER (pack_wf_kernel): for n-tile jt (16
//   output columns) and K step s a 2 KB block [hi: 64 lanes x 16 B | lo: 64 lanes x 16 B], lane (r16, g) holding W[16 jt + r16][32 s +
//   8 g .. + 7].  A wave's W fragment is then ONE fully coalesced 1 KB global_load_dwordx4 straight into the MFMA operand
//   registers, prefetched one K step ahead into a second register set.  The two waves that share columns request the same lines
//   back to back (L1 / L2 hits).
// * 4 waves as 2 (M) x 2 (N): a wave owns 128 x BN/2 outputs = 8 x TN accumulator tiles (128 registers at BN = 128).  A fragments
//   are streamed from LDS one m-tile ahead of their 3 TN MFMAs (inline-asm ds_read_b128 + counted lgkmcnt: with LDS-DMA in the
//   kernel hipcc would guard every LDS read with s_waitcnt vmcnt(0), i.e. wait for the prefetch just issued).
// * One s_barrier per K step (4 waves): "stage k landed for everyone, stage k - 1 read by everyone".
// * Accumulation order per output element is the one of gemm_split16.hip (per K step: lo*hi, hi*lo, hi*hi), so results are
//   bit-identical to that kernel's.
// * Epilogue straight from the accumulator registers (run_epilogue): its latency chain and store shape now overlap the other
//   workgroup's K loop.
#include <cstdlib>
#include <map>
#include <tuple>
#include <type_traits>
#include <utility>

#include "gemm_epi.h"

namespace ribca {

namespace {

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

template <int OFF>
__device__ __forceinline__ void lds_read16(f16x8& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void gload16(f16x8& dst, unsigned voff, const void* sbase) {
  // the base is wave-uniform by construction; readfirstlane states it (under SGPR pressure hipcc otherwise hands the "s" operand a
  // VGPR pair, which the saddr form of the instruction does not take)
  const unsigned long long b = (unsigned long long)(uintptr_t)sbase;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const unsigned long long u = ((unsigned long long)hi << 32) | lo;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(u), "n"(OFF) : "memory");
}

}  // namespace

// weight [Np][2*Kp] packed-split  ->  fragment order (see the header comment); one thread per 16-byte vector
__global__ void pack_wf_kernel(const uint16_t* __restrict__ W, int ldw, int Np, int nk, uint16_t* __restrict__ WF) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)(Np / 16) * nk * 128;
  if (idx >= total) return;
  const int lane = (int)(idx & 63), hl = (int)((idx >> 6) & 1);
  const long long blk = idx >> 7;
  const int s = (int)(blk % nk), jt = (int)(blk / nk);
  const int r16 = lane & 15, g = lane >> 4;
  const uint4* src = reinterpret_cast<const uint4*>(W + (size_t)(16 * jt + r16) * ldw + (4 * s + g) * 16 + hl * 8);
  reinterpret_cast<uint4*>(WF)[idx] = *src;
}
void launch_pack_wf(const uint16_t* W, int ldw, int Np, int Kp, uint16_t* WF, hipStream_t s) {
  const int nk = Kp / BK;
  const long long total = (long long)(Np / 16) * nk * 128;
  hipLaunchKernelGGL(pack_wf_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, ldw, Np, nk, WF);
}

// Diagnostic stamps (ABL = 4, variant 44): same record as gemm_split16.hip's variant 12 -- per workgroup t0 entry, t1 first stage
// landed, t2 K loop done, t3 stores accepted (wave 0), XCC id, HW id, then each wave's "stores accepted" -- into a buffer nothing reads.
static __device__ unsigned long long* g_duo_stamps = nullptr;
static __device__ unsigned int g_duo_stamp_cap = 0;      // workgroups the buffer has room for: larger grids do not stamp
int duo_set_stamp_buffer(void* dev_ptr, unsigned int capacity_blocks) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_duo_stamp_cap), &capacity_blocks, sizeof(capacity_blocks)) != hipSuccess) return 1;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_duo_stamps), &dev_ptr, sizeof(dev_ptr));
}

// ABL bit mask (timing ablations, results wrong on purpose): 1 = no epilogue, 2 = no A loads, 4 = no W loads; 8 = stamps (results correct)
// BM = rows of the tile: 256 with a 2-stage A ring (64 KB; operands of step k + 1 in flight during step k), or 192 with a 3-stage ring
//      (72 KB) and three W register sets: operands of steps k + 1 and k + 2 in flight.  One K step of one workgroup is 0.55-0.75 us
//      of MFMA issue, an L2 round trip under load 1-1.5 us: with one step of prefetch a workgroup that has the CU's matrix pipe to
//      itself (its neighbour is in its epilogue) is latency-bound (measured 1.6 us per step), with two steps it is not.
// WM = waves along M: 2 -> waves as 2 (M) x 2 (N), a wave owns BM/2 rows x BN/2 columns, the two waves of a column half request the
//                          same W fragments;
//                     1 -> waves as 1 x 4, a wave owns ALL rows x BN/4 columns: every W fragment is requested once per workgroup,
//                          every wave reads the whole A stage from LDS (96-128 KB of LDS reads per workgroup and step, 1/3 of the array).
// NW = waves per workgroup (4 or 8: two workgroups per CU either way, by LDS), WM x WN = NW their layout, NWS = W register sets.
// With 4 waves a workgroup has ONE wave per SIMD: its epilogue (GELU + split, ~80 VALU instructions per 16 x 16 tile) then issues
// from a single in-order wave beside the neighbour workgroup's MFMA stream -- measured 14.5 cycles per VALU instruction, 13 us per
// tile against 5.4 us with the CU to itself, and the neighbour's K loop a third slower.  8 waves (128 registers each) put two
// waves per SIMD on every phase.
template <int BM, int NW, int WM, int TN, int NWS, class Epi, int ABL>
__global__ __launch_bounds__(64 * NW, NW / 2) void gemm_ps_duo_kernel(const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ WF, int M,
                                                                       int Kp, int mtiles, int ntiles, Epi epi, int mode, int delay) {
  static_assert(BM == 256 || BM == 192 || BM == 128, "tile rows");
  constexpr int NST = BM == 256 ? 2 : 3;      // ring stages
  static_assert(NWS == NST || (NST == 3 && NWS == 2), "W register sets");
  constexpr int WN = NW / WM, BN = 16 * TN * WN;
  constexpr int MT = BM / 16 / WM;            // m-tiles per wave
  // epilogue blocks of RB row tiles: the whole wave tile at once where its residual / table loads fit beside the accumulators
  // (all of them in flight together: one round trip instead of one per block), else 4 or 3 row tiles at a time
  constexpr int RB = MT * TN <= (NW == 8 ? 12 : 24) ? MT : (MT % 4 == 0 ? 4 : 3), NB = MT / RB;
  static_assert(NB * RB == MT, "row tiles per wave");
  constexpr int STAGE = BM * ROWB;            // A only
  constexpr int GPW = BM / 8 / NW;            // 8-row DMA groups per wave and stage
  static_assert(GPW * 8 * NW == BM, "DMA groups");
  constexpr int NA = (ABL & 2) ? 0 : GPW, NWL = (ABL & 4) ? 0 : 2 * TN;   // vector-memory operations per K step and wave: A, W
  // issue order inside a step: W (for step k + NWS - 1), then A (for step k + NST - 1); what may stay in flight at the top of step k
  constexpr int FLY = NST == 2 ? 0 : (NWS == 3 ? NA + NWL : NA);
  constexpr int PERIOD = NST == NWS ? NST : NST * NWS;
  // z touch (EpiResid: the drain's z loads are HBM / Infinity-Cache misses, 13 of a tile's 17 us of epilogue here): every thread reads
  // one dword of each 128-byte line of "its" row of the residual tile right behind the LAST operand batch; vmcnt retires in order,
  // so the remaining waits leave these NT youngest operations in flight and nothing waits for them before the epilogue does.
  constexpr int NT = (Epi::kTouch && !(ABL & 1)) ? BN / 32 : 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nblk = mtiles * ntiles;
  int bid = blockIdx.x;
  {
    // blocks b, b + 8, ... run on one XCD (one L2): XCD x walks the contiguous tile range [first, first + cnt) with n fastest, so
    // the n-tiles of an m-tile run together and its A rows are fetched once.  When the weight is larger than the L2 can keep
    // beside the A / output streams (panel > 0: n-tiles per panel, chosen by the launcher), the whole m-tile rows of the range are
    // walked panel by panel -- every m-tile's n-tiles of panel 0, then of panel 1, ... -- so a panel of W stays resident while the
    // A rows stream past it (A is then fetched once per panel); the partial rows at the two ends of the range keep the plain order.
    const int xcd = bid & 7, loc = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int cnt = xcd < r ? q + 1 : q;
    bid = first + loc;
    const int panel = mode >> 8;
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
  const int wm = wave / WN, wn = wave - wm * WN;
  const int r16 = lane & 15, g = lane >> 4;
  const int nk = Kp / BK;
  // EpiResidZK: the residual tile's BN / 32 K steps follow the product's own through the same ring
  constexpr bool ZK = is_zk<Epi>::value;
  constexpr int ZS = ZK ? BN / 32 : 0;
  static_assert(!ZK || (NST == 3 && ABL == 0), "the residual-through-the-ring form exists for the production tile only");
  const int nkz = nk + ZS;
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, tc1 = 0, tc2 = 0;      // tc: shader-clock stamps around the K loop (in-kernel clock)
  if (ABL & 8) ts0 = __builtin_amdgcn_s_memrealtime();
  // the scheduling A/B switches below (RIBCA_DUO_MODE) exist in the diagnostic library only; none of them paid
#ifdef RIBCA_DIAG
  const int lab = mode & 0xff;
#else
  constexpr int lab = 0;
#endif
  // mode bit 0: static wave priority by the CU's workgroup slot (TG_ID of HW_ID, bits 19:16); bit 1: the first round's odd-slot
  // workgroups start `delay` x 10 ns late; bit 2: epilogue at raised priority (A/B switches: none paid)
  if (lab & 3) {
    unsigned int hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const bool odd_slot = ((hwid >> 16) & 1u) != 0;
    if (lab & 1) {
      if (odd_slot) __builtin_amdgcn_s_setprio(0);
      else __builtin_amdgcn_s_setprio(2);
    }
    if ((lab & 2) && odd_slot && (int)blockIdx.x < 1024) {
      const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)delay;
      while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(16);
    }
  }

  if (lab & 8) __builtin_amdgcn_s_setprio(3);
  // ---- A ring: wave w loads the 8-row groups w, w + NW, ... of every stage (1 KB per instruction) through a buffer descriptor
  // over the tile's rows: ONE per-lane offset register, everything that varies (group, K step) in the scalar offset, and rows beyond
  // M read as zeros by the descriptor's range check (their outputs are dropped by the epilogue guards).
  const int rows_here = (M - m0) < BM ? (M - m0) : BM;
  const __amdgpu_buffer_rsrc_t a_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A + (size_t)m0 * lda), 0, rows_here * lda * 2, 0x00020000);
  int a_voff;
  {
    const int drow = wave * 8 + (lane >> 3);
    const int dch = (lane & 7) ^ swz_f(drow);   // swz_f(row + 32 i) == swz_f(row): the stride between a wave's groups is 8 NW rows
    a_voff = drow * lda * 2 + dch * 16;
  }
  const int a_gstride = 8 * NW * lda * 2;       // bytes between the groups of one wave
  // the residual tile (ZK): rows of the packed-split stream, 128 bytes = 32 columns per K step, same swizzle
  const __amdgpu_buffer_rsrc_t z_rsrc = zk_rsrc(epi, m0, rows_here, a_rsrc);
  int z_voff = 0, z_gstride = 0;
  if constexpr (ZK) {
    const int drow = wave * 8 + (lane >> 3);
    z_voff = drow * epi.ldz * 2 + (((lane & 7) ^ swz_f(drow)) * 16);
    z_gstride = 8 * NW * epi.ldz * 2;
  }
  auto issue_a = [&](int kk, int slot) {
    if (ABL & 2) return;
    char* st = smem + slot * STAGE + wave * 1024;
    if (ZK && kk >= nk) {
      const int ko = n0 * 4 + (kk - nk) * (4 * BK);
      // 128 x 192 form: the residual rows are read ONCE, by this workgroup alone -- a non-temporal request.  (It also keeps the two
      // DMA paths apart for hipcc: with identical instructions behind the branch it merges them into one load whose DESCRIPTOR is
      // picked from a table in scratch and re-read every K step -- vector-memory operations the counted waits know nothing of.)
      constexpr int ZAUX = BN == 192 ? 2 : 0;
#pragma unroll
      for (int i = 0; i < GPW; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(z_rsrc, (__attribute__((address_space(3))) void*)(st + i * (1024 * NW)), 16, z_voff, i * z_gstride + ko,
                                                 0, ZAUX);
      return;
    }
    const int ko = kk * (4 * BK);
#pragma unroll
    for (int i = 0; i < GPW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(st + i * (1024 * NW)), 16, a_voff, i * a_gstride + ko, 0,
                                               0);
  };
  // ---- W fragments: n-tiles jt0 .. jt0 + TN - 1 of this wave, 2 KB per (n-tile, K step)
  const unsigned wvoff = (unsigned)lane * 16u;
  const char* wbase = reinterpret_cast<const char*>(WF) + (size_t)((n0 >> 4) + wn * TN) * (size_t)nk * 2048;
  const size_t wjstride = (size_t)nk * 2048;
  f16x8 whi[NWS][TN], wlo[NWS][TN];
  auto issue_w = [&](int kk, f16x8 (&hi)[TN], f16x8 (&lo)[TN]) {
    if (ABL & 4) return;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const char* p = wbase + (size_t)j * wjstride + (size_t)kk * 2048;
      gload16<0>(hi[j], wvoff, p);
      gload16<1024>(lo[j], wvoff, p);
    }
  };

  // ZK: everything the epilogue needs from memory is requested here, in front of the first ring stage (vmcnt retires in order: the
  // first operand wait covers these) -- bias of the lane's 4 TN columns, mean of the stored row for each of its MT rows
  float4 zb4[ZK ? TN : 1];
  float zpm[ZK ? MT : 1];
  if constexpr (ZK) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (16 * TN) + 16 * j + 4 * g;
      zb4[j] = n < epi.N ? *reinterpret_cast<const float4*>(epi.bias + n) : float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * (16 * MT) + 16 * i + r16;
      zpm[i] = (epi.prev != nullptr && m < M) ? epi.prev[(size_t)m * epi.prev_stride].y : 0.f;
    }
  }
  f32x4 acc[NB][RB][TN];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (ABL & 4) {
#pragma unroll
    for (int s = 0; s < NWS; ++s)
#pragma unroll
      for (int j = 0; j < TN; ++j) whi[s][j] = wlo[s][j] = f16x8{};
  }

  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned a_hi = lds_base + (unsigned)lds_off(wm * (16 * MT) + r16, 2 * g);     // + 2048 per m-tile, + STAGE per slot
  const unsigned a_lo = a_hi ^ 16u;

  unsigned int touched[NT > 0 ? NT : 1];
  auto touch = [&]() {
    if constexpr (NT > 0) {
      const int trow = tid < BM ? tid : BM - 1;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const float* p = epi.touch_on() ? epi.touch_ptr(m0 + trow, n0 + 32 * i) : epi.touch_ptr(0, 0);
        asm volatile("global_load_dword %0, %1, off" : "=v"(touched[i]) : "v"(p) : "memory");
      }
    }
  };
  const int last_issue = nk - NST;            // the step whose issue is the final operand batch (< 0: all issued in the prologue)
  // K step kk on ring slot SA = kk % NST with W set SW = kk % NWS
  auto step = [&](auto sa_c, auto sw_c, int kk) {
    constexpr int SA = decltype(sa_c)::value, SW = decltype(sw_c)::value;
    // the operands of step kk have landed; younger ones (FLY operations: issued one step ago) may stay in flight
    if constexpr (ZK) {                         // behind the last W batch only the next stage's A (or z) rows stay in flight
      if (kk + 1 < nk) wait_vmcnt<FLY>();
      else wait_vmcnt<NA>();
    } else if (NT > 0 && kk > last_issue) {     // the touches sit behind every operand batch
      if (FLY > 0 && kk + 1 < nk) wait_vmcnt<FLY + NT>();
      else wait_vmcnt<NT>();
    } else {
      if (FLY > 0 && kk + 1 < nk) wait_vmcnt<FLY>();
      else wait_vmcnt<0>();
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(whi[SW][j]), "+v"(wlo[SW][j]));
    __builtin_amdgcn_s_barrier();             // stage kk landed for every wave; the slot of stage kk - 1 read by all
    asm volatile("" ::: "memory");
    if ((ABL & 8) && kk == 0) { ts1 = __builtin_amdgcn_s_memrealtime(); tc1 = __builtin_amdgcn_s_memtime(); }
    if (kk + NWS - 1 < nk) {
      constexpr int SN = (SW + NWS - 1) % NWS;
      issue_w(kk + NWS - 1, whi[SN], wlo[SN]);
    }
    if (kk + NST - 1 < nkz) issue_a(kk + NST - 1, (SA + NST - 1) % NST);
    if (NT > 0 && kk == last_issue) touch();
    __builtin_amdgcn_sched_barrier(0);
    f16x8 ah[2], al[2];
    const unsigned a_hi_s = a_hi + (unsigned)(SA * STAGE), a_lo_s = a_lo + (unsigned)(SA * STAGE);   // (the 16-bit offset field cannot hold slot 2)
    lds_read16<0>(ah[0], a_hi_s);
    lds_read16<0>(al[0], a_lo_s);
    static_for<MT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int cur = i & 1, nxt = cur ^ 1;
      if constexpr (i + 1 < MT) {
        lds_read16<(i + 1) * 2048>(ah[nxt], a_hi_s);
        lds_read16<(i + 1) * 2048>(al[nxt], a_lo_s);
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[cur]), "+v"(al[cur])::"memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[cur]), "+v"(al[cur])::"memory");
      }
      f32x4(&a)[TN] = acc[i / RB][i % RB];
#pragma unroll
      for (int j = 0; j < TN; ++j) a[j] = mfma_f16(wlo[SW][j], ah[cur], a[j]);
#pragma unroll
      for (int j = 0; j < TN; ++j) a[j] = mfma_f16(whi[SW][j], al[cur], a[j]);
#pragma unroll
      for (int j = 0; j < TN; ++j) a[j] = mfma_f16(whi[SW][j], ah[cur], a[j]);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // prologue, in the steady-state order (W of step s before A of step s + (NST - NWS) ...): steps 0 .. NST - 2 of A, 0 .. NWS - 2 of W
  static_for<NST - 1>([&](auto s_c) {
    constexpr int S = decltype(s_c)::value;
    if (S < NWS - 1 && S < nk) issue_w(S, whi[S], wlo[S]);
    if (S < nkz) issue_a(S, S);
  });
  if (NT > 0 && last_issue < 0) touch();
#ifdef RIBCA_KLOOP_PRIO      // A/B (tools/build_ab_lib.py ... -DRIBCA_KLOOP_PRIO): the K loop's waves ahead of the co-resident workgroup's epilogue waves at issue
  __builtin_amdgcn_s_setprio(2);
#endif
  int kk = 0;
  for (; kk + PERIOD <= nk; kk += PERIOD)
    static_for<PERIOD>([&](auto p_c) {
      constexpr int P = decltype(p_c)::value;
      step(std::integral_constant<int, P % NST>{}, std::integral_constant<int, P % NWS>{}, kk + P);
    });
  static_for<PERIOD - 1>([&](auto p_c) {
    constexpr int P = decltype(p_c)::value;
    if (kk + P < nk) step(std::integral_constant<int, P % NST>{}, std::integral_constant<int, P % NWS>{}, kk + P);
  });
#ifdef RIBCA_KLOOP_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif

  if constexpr (ZK) {
    // ---- the residual tile: K step t holds columns n0 + 32 t .. + 31 of the stored rows; against the identity, the column tile whose
    // 16 columns sit at offset 0 or 16 of that step receives z_lo and z_hi (exact products, fp32 accumulate)
    const int p8 = r16 & 7;
    const unsigned one = (p8 & 1) ? 0x3C000000u : 0x00003C00u;       // 1.0 in the lane's element r16 % 8
    const u32x4 pat = {(p8 >> 1) == 0 ? one : 0u, (p8 >> 1) == 1 ? one : 0u, (p8 >> 1) == 2 ? one : 0u, (p8 >> 1) == 3 ? one : 0u};
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const f16x8 id0 = __builtin_bit_cast(f16x8, g == (r16 >> 3) ? pat : zero4), id16 = __builtin_bit_cast(f16x8, g == 2 + (r16 >> 3) ? pat : zero4);
    int zsel[TN];
    f16x8 idj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c0 = wn * (16 * TN) + 16 * j;
      zsel[j] = n0 + c0 < epi.N ? (c0 >> 5) : -1;       // column tiles beyond N take nothing (their K step may lie behind the row)
      idj[j] = (c0 & 16) ? id16 : id0;
    }
    const int nk3 = nk % NST;
    static_for<ZS>([&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      if constexpr (t + 1 < ZS) wait_vmcnt<NA>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr (t + NST - 1 < ZS) issue_a(nk + t + NST - 1, (nk3 + t + NST - 1) % NST);
      __builtin_amdgcn_sched_barrier(0);
      bool mine = false;
#pragma unroll
      for (int j = 0; j < TN; ++j) mine = mine || zsel[j] == t;
      if (mine) {      // wave-uniform
        const int slot = (nk3 + t) % NST;
        f16x8 ah[2], al[2];
        const unsigned a_hi_s = a_hi + (unsigned)(slot * STAGE), a_lo_s = a_lo + (unsigned)(slot * STAGE);
        lds_read16<0>(ah[0], a_hi_s);
        lds_read16<0>(al[0], a_lo_s);
        static_for<MT>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          constexpr int cur = i & 1, nxt = cur ^ 1;
          if constexpr (i + 1 < MT) {
            lds_read16<(i + 1) * 2048>(ah[nxt], a_hi_s);
            lds_read16<(i + 1) * 2048>(al[nxt], a_lo_s);
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[cur]), "+v"(al[cur])::"memory");
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[cur]), "+v"(al[cur])::"memory");
          }
          f32x4(&a)[TN] = acc[i / RB][i % RB];
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if (zsel[j] == t) {
              a[j] = mfma_f16(idj[j], al[cur], a[j]);
              a[j] = mfma_f16(idj[j], ah[cur], a[j]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    });
  }

  if (ABL & 8) { ts2 = __builtin_amdgcn_s_memrealtime(); tc2 = __builtin_amdgcn_s_memtime(); }
  if (ABL & 1) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[b][i][j]));
    if ((ABL & 8) && g_duo_stamps != nullptr && blockIdx.x < g_duo_stamp_cap && lane == 0) {
      unsigned long long* o = g_duo_stamps + (size_t)blockIdx.x * 20;
      o[6 + wave] = ts2;
      if (wave == 0) {
        o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts2; o[18] = tc1; o[19] = tc2;
        unsigned int xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        o[4] = xcc; o[5] = hwid;
      }
    }
    return;
  }
  if (lab & 4) __builtin_amdgcn_s_setprio(3);
  if (lab & 8) __builtin_amdgcn_s_setprio(0);     // bit 3: K loop at priority 3 (set below the prologue), epilogue back at 0
  const int mbase = m0 + wm * (16 * MT) + r16, nbase = n0 + wn * (16 * TN) + 4 * g;
  // a tile wholly inside the product (all but the last row of m-tiles, every n-tile when N is a multiple of BN) takes the unguarded
  // form of the epilogue: workgroup-uniform branch
  if constexpr (ZK) {
    if (n0 + wn * (16 * TN) >= epi.N) return;      // a wave column block wholly beyond N (N is a multiple of the block): nothing to write
    const int blk = nt * WN + wn;
    if (m0 + BM <= M) resid_zk_epilogue<TN, MT, RB, true>(epi, mbase, nbase, blk, g, acc, zb4, zpm);
    else resid_zk_epilogue<TN, MT, RB, false>(epi, mbase, nbase, blk, g, acc, zb4, zpm);
    // second copy of the new rows in the MX3 format (the operand of the MX forms of the next qkv / fc1, gemm_mx.hip): the 128 x 192 tile only
    // (48-column wave blocks; its launcher adds the 2 KB of exchange space behind the ring)
    if constexpr (TN == 3 && WM == 1 && RB == MT) {
      if (epi.zmx.hi != nullptr) {      // workgroup-uniform
        // (the launcher sized the LDS for it: the staging image over the ring -- no wave reads the ring once all have reached the emission's
        // first barrier -- and the exchange space behind it)
        const unsigned stg = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem, xch = stg + (unsigned)kMx3StageBytes;
        if (m0 + BM <= M) mx3_emit_wave48<MT, true>(epi.zmx, M, m0, n0 + wn * 48, g, r16, wave, stg, xch, acc[0]);
        else mx3_emit_wave48<MT, false>(epi.zmx, M, m0, n0 + wn * 48, g, r16, wave, stg, xch, acc[0]);
      }
    }
  } else if constexpr (is_mx_out<Epi>::value) {
    static_assert(WM == 1 && TN == 2 && NB == 1, "the MX3-emitting GELU epilogue: 4 waves as 1 x 4 over a 128-wide tile");
    if (m0 + BM <= M) gelu_mx_epilogue<MT, true>(epi, m0 + r16, n0 + wn * 32, g, acc);
    else gelu_mx_epilogue<MT, false>(epi, m0 + r16, n0 + wn * 32, g, acc);
  } else if (m0 + BM <= M && n0 + BN <= epi.N && !(mode & 0x10)) {      // mode bit 4 (RIBCA_DUO_GUARDED=1): A/B switch, always the guarded form
#pragma unroll
    for (int b = 0; b < NB; ++b) run_epilogue<TN, Epi, RB, true>(epi, mbase + 16 * RB * b, nbase, acc[b]);
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) run_epilogue<TN, Epi, RB>(epi, mbase + 16 * RB * b, nbase, acc[b]);
  }
  if constexpr (NT > 0) {
#pragma unroll
    for (int i = 0; i < NT; ++i) asm volatile("" ::"v"(touched[i]));
  }
  if ((ABL & 8) && g_duo_stamps != nullptr && blockIdx.x < g_duo_stamp_cap && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long ts3 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* o = g_duo_stamps + (size_t)blockIdx.x * 20;
    o[6 + wave] = ts3;
    if (wave == 0) {
      o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3; o[18] = tc1; o[19] = tc2;
      unsigned int xcc, hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      o[4] = xcc; o[5] = hwid;
    }
  }
}

// ---------------------------------------------------------------------------------------------- host side
// Fragment-order copies made on demand for weights that were not created through ribca_vit_create (tests, tools): keyed by the
// packed weight's device address.  Model weights carry their own copy (GemmArgs::WF).
#ifndef RIBCA_DIAG
static const uint16_t* wf_for(const GemmArgs& g, hipStream_t) { return g.WF; }     // product build: the caller owns the fragment-order copy
#else
// (diagnostic library only: single-threaded tools; the copy is remade on the caller's stream on every call)
static const uint16_t* wf_for(const GemmArgs& g, hipStream_t s) {
  if (g.WF != nullptr) return g.WF;
  static std::map<std::tuple<const void*, int, int>, uint16_t*> cache;
  const int Np = gemm_padded_n(g.N);
  const auto key = std::make_tuple((const void*)g.W, Np, g.Kp);
  auto it = cache.find(key);
  if (it != cache.end()) {
    // the same address can be reused for different contents (test tensors): repack every time in this fallback path
    launch_pack_wf(g.W, g.ldw, Np, g.Kp, it->second, s);
    return it->second;
  }
  uint16_t* wf = nullptr;
  if (hipMalloc(&wf, (size_t)Np * 2 * g.Kp * sizeof(uint16_t)) != hipSuccess) return nullptr;
  launch_pack_wf(g.W, g.ldw, Np, g.Kp, wf, s);
  cache[key] = wf;
  return wf;
}
#endif

// false = not launched (no fragment-order weight): the caller runs the one-workgroup-per-CU kernel instead
template <int BM, int NW, int WM, int NWS, int BN, class Epi>
static bool launch_duo_impl(const GemmArgs& g, const Epi& epi, hipStream_t s, int abl) {
  static const int mode = getenv("RIBCA_DUO_MODE") ? atoi(getenv("RIBCA_DUO_MODE")) : 0;
  static const int delay_per_step = getenv("RIBCA_DUO_DELAY") ? atoi(getenv("RIBCA_DUO_DELAY")) : 40;   // 10 ns ticks per K step
  static const int lds_pad = getenv("RIBCA_DUO_SOLO") ? 1 : 0;      // diagnostics: one workgroup per CU (LDS padded past half)
  constexpr int TN = BN / (16 * (NW / WM));
  constexpr int NST = BM == 256 ? 2 : 3;
  const int mtiles = (g.M + BM - 1) / BM;
  const int ntiles = gemm_padded_n(g.N) / BN;
  size_t lds = lds_pad ? (size_t)100 * 1024 : (size_t)NST * BM * ROWB;
  if constexpr (is_zk<Epi>::value && TN == 3 && WM == 1) {      // an MX3 copy of the new rows: staging image + exchange space (gemm_epi.h mx3_emit_wave48)
    if (epi.zmx.hi != nullptr && lds < (size_t)kMx3StageBytes + 2048) lds = (size_t)kMx3StageBytes + 2048;
  }
  const uint16_t* wf = wf_for(g, s);
  if (wf == nullptr) return false;
  const dim3 grid(mtiles * ntiles), block(64 * NW);
  const int delay = delay_per_step * (g.Kp / BK);
  // W panel order (see the kernel's tile map), OFF by default: RIBCA_DUO_PANEL_KB = the most W (KB) a panel may hold, applied only where
  // the whole weight is larger.  Measured at D = 576 (W = 4.0 / 5.3 MB against the XCD's 4 MB L2; profiles/r3/duo_w_panel_experiment.txt):
  // 2816 KB halves the counter traffic of those launches (fc1 3.7 -> 2.5 GB, qkv 2.7 -> 1.7 GB; whole pass 21.7 -> 19.4 TB) and costs
  // 0.4-0.9 % of throughput, more with smaller panels -- the misses it removes are served by the Infinity Cache at no cost in time,
  // while every panel re-reads the A rows.
  static const int panel_kb = getenv("RIBCA_DUO_PANEL_KB") ? atoi(getenv("RIBCA_DUO_PANEL_KB")) : 0;
  int panel = 0;
  {
    const size_t tile_bytes = (size_t)BN * g.Kp * 4, w_bytes = tile_bytes * ntiles;
    if (panel_kb > 0 && w_bytes > (size_t)panel_kb * 1024) {
      const int fit = (int)((size_t)panel_kb * 1024 / tile_bytes);
      if (fit >= 1) {
        const int np = (ntiles + fit - 1) / fit;
        panel = (ntiles + np - 1) / np;
      }
    }
  }
  static const int guarded = (getenv("RIBCA_DUO_GUARDED") && atoi(getenv("RIBCA_DUO_GUARDED")) != 0) ? 0x10 : 0;
  const int mode_p = (mode & 0xef) | guarded | (panel << 8);
  auto go = [&](auto abl_c) {
    constexpr int ABL = decltype(abl_c)::value;
    static unsigned long long attr_done = 0ull;
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_ps_duo_kernel<BM, NW, WM, TN, NWS, Epi, ABL>), (int)(100 * 1024), attr_done)) return;
    hipLaunchKernelGGL((gemm_ps_duo_kernel<BM, NW, WM, TN, NWS, Epi, ABL>), grid, block, lds, s, g.A, g.lda, wf, g.M, g.Kp, mtiles, ntiles, epi, mode_p, delay);
  };
  // the diagnostic forms (no epilogue / stamps) exist for the epilogues tools/bench_gemm.py and tools/stamp_duo.py drive
#ifdef RIBCA_DIAG
  constexpr bool diag = std::is_same<Epi, EpiGelu>::value || std::is_same<Epi, EpiResid>::value;
#else
  constexpr bool diag = false;
#endif
  if constexpr (diag) {
    switch (abl) {
      case 1: go(std::integral_constant<int, 1>{}); return true;
      case 8: go(std::integral_constant<int, 8>{}); return true;
      case 9: go(std::integral_constant<int, 9>{}); return true;
      default: break;
    }
  }
  go(std::integral_constant<int, 0>{});
  return true;
}

template <int BN, class Epi>
bool launch_duo(const GemmArgs& g, const Epi& epi, hipStream_t s, int abl) {
  // RIBCA_DUO_FORM (A/B): 0 = 192-row tiles, 3-stage ring, 4 waves as 1 x 4 (2 x 2 for 96-wide tiles), 3 W sets (production);
  // 1 = the same tile with 8 waves as 2 x 4 (4 x 2), 2 W sets; 2 = 256 rows, 2-stage ring, 4 waves
  static const int form = getenv("RIBCA_DUO_FORM") ? atoi(getenv("RIBCA_DUO_FORM")) : 0;
  (void)form;
  if constexpr (BN == 192) {
    // 128 x 192 tiles, 4 waves as 1 x 4: a wave owns all 128 rows x 48 columns (96 accumulator registers) and every W fragment is
    // requested by ONE wave.  (192 rows x 192 columns needs 144 + 48 + 16 registers beside the addresses: hipcc spills fragment
    // registers inside the K loop, and a spilled register that an inline-asm load is still filling holds garbage.)
    return launch_duo_impl<128, 4, 1, 3, BN, Epi>(g, epi, s, abl);
  } else if constexpr (BN % 64 == 0) {
#ifdef RIBCA_DIAG
    if constexpr (!is_zk<Epi>::value) {      // the A/B tile forms do not exist for the residual-through-the-ring epilogue
      if (form == 2) return launch_duo_impl<256, 4, 1, 2, BN, Epi>(g, epi, s, abl);
      if (form == 1) return launch_duo_impl<192, 8, 2, 2, BN, Epi>(g, epi, s, abl);
    }
#endif
    return launch_duo_impl<192, 4, 1, 3, BN, Epi>(g, epi, s, abl);
  } else {   // BN = 96
#ifdef RIBCA_DIAG
    if constexpr (!is_zk<Epi>::value) {
      if (form == 2) return launch_duo_impl<256, 4, 2, 2, BN, Epi>(g, epi, s, abl);
      if (form == 1) return launch_duo_impl<192, 8, 4, 2, BN, Epi>(g, epi, s, abl);
    }
#endif
    return launch_duo_impl<192, 4, 2, 3, BN, Epi>(g, epi, s, abl);
  }
}

// mlp.fc1 (LayerNorm folded) writing its GELU output in the MX3 format of gemm_mx.hip: 192 x 128 tiles, 4 waves as 1 x 4 (one 32-column
// block per wave and row).  N % 128 == 0 and a fragment-order weight are required.
bool launch_gemm_gelu_mx(const GemmArgs& g, const float2* rowstat, const float* csum, const MxAct& out, hipStream_t s) {
  if (g.N % 128 != 0 || g.WF == nullptr || out.Kp != g.N) return false;
  // RIBCA_MX_NT bit 2 (default off): the MX3 planes of h stored non-temporal from THIS kernel (fc1 at D = 288).  The MX kernel's fc1
  // (D = 384 / 576, bit 1, gemm_mx.hip) gains 2-5 % from it; here it measured nothing (profiles/r4/ab_mx_nt_stores.txt)
  static const int mx_nt = getenv("RIBCA_MX_NT") ? atoi(getenv("RIBCA_MX_NT")) : 2;
  const EpiGeluMx epi{out, g.bias, g.M, g.N, rowstat, csum, 1, (mx_nt >> 2) & 1};
  return launch_duo_impl<192, 4, 1, 3, 128, EpiGeluMx>(g, epi, s, 0);
}

#define RIBCA_DUO_INST(BN, EPI) template bool launch_duo<BN, EPI>(const GemmArgs&, const EPI&, hipStream_t, int);
RIBCA_DUO_INST(128, EpiGelu) RIBCA_DUO_INST(96, EpiGelu) RIBCA_DUO_INST(64, EpiGelu)
RIBCA_DUO_INST(128, EpiGeluLn) RIBCA_DUO_INST(96, EpiGeluLn) RIBCA_DUO_INST(64, EpiGeluLn)
RIBCA_DUO_INST(128, EpiQKVLn) RIBCA_DUO_INST(96, EpiQKVLn) RIBCA_DUO_INST(64, EpiQKVLn)
RIBCA_DUO_INST(128, EpiResidZK) RIBCA_DUO_INST(96, EpiResidZK) RIBCA_DUO_INST(64, EpiResidZK)
RIBCA_DUO_INST(192, EpiQKVLn) RIBCA_DUO_INST(192, EpiGeluLn) RIBCA_DUO_INST(192, EpiResidZK)
#ifdef RIBCA_DIAG
RIBCA_DUO_INST(128, EpiResid) RIBCA_DUO_INST(96, EpiResid) RIBCA_DUO_INST(64, EpiResid)
RIBCA_DUO_INST(128, EpiQKV) RIBCA_DUO_INST(96, EpiQKV) RIBCA_DUO_INST(64, EpiQKV)
RIBCA_DUO_INST(128, EpiRowMap) RIBCA_DUO_INST(96, EpiRowMap) RIBCA_DUO_INST(64, EpiRowMap)
#endif
#undef RIBCA_DUO_INST

}  // namespace ribca
