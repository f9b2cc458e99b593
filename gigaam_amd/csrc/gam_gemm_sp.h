// gam_gemm_sp.h -- the large-M split-fp16 GEMM: (64 MT) x (64 NW) block tiles fed entirely by
// LDS-DMA (global_load_lds_dwordx4), two LDS stages (<= 64 KB each), one barrier per k-tile.
//
// Same arithmetic as gam_gemm16.h (a.w ~= (a_hi.w_hi + a_hi.w_lo + a_lo.w_hi) 2^-s on
// v_mfma_f32_32x32x16_f16, fp32 accumulate) but both operands arrive already split, in the
// "sp32" layout that the producing kernels write instead of fp32:
//
//   element (row, k)  ->  halfs  row*2K + (k/32)*64 + (k%32)        hi = fp16(x)
//                                row*2K + (k/32)*64 + 32 + (k%32)   lo = fp16(x - hi)
//
// i.e. the 4 bytes an fp32 element would occupy hold its (hi, lo) pair, regrouped so that one
// row's share of a 32-deep k-tile is ONE 128-byte line [hi x32 | lo x32].  A k-tile of a block is
// then (BM + 256) full lines; a wave-wide global_load_lds_dwordx4 moves 8 of them (1 KiB) straight
// into LDS -- no staging registers, no ds_write pass, no conversion in the GEMM.  (The 128x128
// register-staged kernel measured TD/TCP-bound on exactly that traffic: profiles/r01_f16x3_*.)
//
// LDS image: rows of 128 B (8 slots of 16 B), slot' = slot ^ ((row >> 1) & 7).  LDS-DMA writes
// lane-linear, so the XOR is applied to the per-lane SOURCE address; fragment reads apply the same
// XOR and every 16-lane service group of a ds_read_b128 (rows distinct mod 16) covers all 16 slots
// of the 256-byte bank row exactly once: conflict-free without padding.
//
// 2 (M) x NW (N) waves; a wave owns (32 MT) x 64 outputs = MT x 2 MFMA tiles, 3 MFMAs per tile per
// k16-step.  At NW = 4 (256-wide tiles, one workgroup per CU) the bytes moved per FLOP are half
// those of the 128x128 kernel; the k-tile after next lands while the current one is multiplied.
// MT in {2,3,4} and NW in {2,4} are picked per launch (gam_gemm_sp_pick) to minimise the tail of the
// last round of tiles.
#pragma once
#include "gam_gemm16.h"
#if defined(GAM_SP_INSTRUMENT) && GAM_SP_INSTRUMENT
#include <algorithm>
#include <vector>
#endif

// -DGAM_SP_INSTRUMENT=1 compiles the GAM_SP_DBG experiment switches in (1: skip the epilogue, 2: one
// k-tile only, 4: per-phase clock64 counters written into the last C row, 8: drop the in-loop barrier).
// They produced profiles/r01_gemm_sp_clock_random_vs_zero.txt; production builds carry none of it.
#ifndef GAM_SP_INSTRUMENT
#define GAM_SP_INSTRUMENT 0
#endif
#define GAM_SP_DBG(g) (GAM_SP_INSTRUMENT ? (g).dbg : 0)
// GAM_SP_DBG & 16 (instrumented builds): per-workgroup timeline -- wall clock (100 MHz) at entry / operands of the first two
// k-tiles landed / main loop done / epilogue done, shader clock around the loop -- into g.tlog[tile * 8 ..]; the launcher
// prints the distribution (tools/gemm_sp_test.py; profiles/r03_gemm_timeline.txt).
#if GAM_SP_INSTRUMENT
#define GAM_SP_TL(i)                                                                  \
  if ((GAM_SP_DBG(g) & 16) && g.tlog != nullptr && threadIdx.x == 0) {                        \
    g.tlog[(size_t)lid * 8 + (i)] = wall_clock64();                                   \
    if ((i) == 1 || (i) == 2) g.tlog[(size_t)lid * 8 + 4 + (i)] = clock64();          \
    if ((i) == 0) g.tlog[(size_t)lid * 8 + 7] = blockIdx.x;                            \
  }
#else
#define GAM_SP_TL(i)
#endif

#define GAM_SP_MIN_M 1      // rows from which the encoder runs on this kernel family: all (with split-K for small grids it beats the
                            // 128x128 register-staged kernels from one 5 s clip up, profiles/r03_smallm_sweep.txt); GAM_SP_MIN_M overrides

template <int MT, int NW, int NS = 2>
struct GamGemmSpCfg {
  static constexpr int BM = 64 * MT;
  static constexpr int BN = 64 * NW;
  static constexpr int NWAVES = 2 * NW;                 // 2 (M) x NW (N) waves
  static constexpr int NT = 64 * NWAVES;
  static constexpr int A_BYTES = BM * 128;
  static constexpr int W_BYTES = BN * 128;
  static constexpr int STAGE = A_BYTES + W_BYTES;
  static constexpr int SMEM = NS * STAGE;      // NS LDS stages: the k-tiles kt .. kt + NS - 1 are resident / in flight
  static constexpr int NAI = BM / 8 / NWAVES;   // A DMA pieces per wave per k-tile (8 rows x 128 B each)
  static constexpr int NWI = BN / 8 / NWAVES;   // W DMA pieces per wave per k-tile
};

// NS = LDS stages.  2: the k-tile after next lands while the current one is multiplied -- enough for the 8-wave tiles, whose
// k-tile takes 1.4-1.8 us of MFMA work.  3 (r04, the 4-wave tiles of small grids): a 128 x 128 k-tile is 0.3 us of MFMA work but
// its 32 KB take ~0.8 us from issue to landed, and with two stages exactly ONE k-tile of fetch is in flight, so the loop ran at
// the fetch latency (per-workgroup timeline, profiles/r04_gemm_timeline_m2008.txt: 0.78 us per k-tile at M = 2008 whatever the
// shape).  With three stages TWO k-tiles are in flight: the in-loop wait is a counted vmcnt (the newest tile's pieces may still
// be outstanding) in front of a bare s_barrier -- __syncthreads() would drain the queue (cdna_hip_programming.md, glds rule).
// H16 (r04, the opt-in GAM_GEMM_F16 speed mode): ONE fp16 MFMA per product -- the arithmetic contract of the reference's own GPU
// default (fp16 autocast, /root/reference/gigaam/model.py:34-37), fp32 accumulation.  Both operands are then PLAIN fp16 rows
// (the producing kernels store format 2 of gam_store4: the fp16 of each value, row-major), and the kernel is told HALF the
// reduction length: a row's 128-byte share of a "k-tile" holds 64 consecutive fp16 values instead of 32 (hi, lo) pairs, so the
// DMA, the LDS image, the swizzle and the fragment reads are exactly those of the three-term kernel -- what used to be the lo
// half of the line is simply the next 32 k-values -- and the MFMA stream multiplies (first half x first half) + (second half x
// second half): two MFMAs per 32 x 32 x 32 block instead of three per 32 x 32 x 16.  Per unit of K: a third of the matrix
// work, half the operand bytes.  (A first version kept the sp32 operands and fetched only the hi half of every line: a 64-byte
// piece of each 128-byte line costs the memory path the whole line, and the kernel ran at half the delivery rate --
// profiles/r04_fastmode.txt.)
template <int ACT, int MT, int NW, int NS = 2, bool H16 = false>
__global__ __launch_bounds__(128 * NW, NW == 2 ? 2 : 1) void gam_gemm_sp_kernel(GamGemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gam_smem_sp[];
  using Cfg = GamGemmSpCfg<MT, NW, NS>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, NAI = Cfg::NAI, NWI = Cfg::NWI, NWAVES = Cfg::NWAVES;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NW, wn = wave % NW;

  const int nbn = (g.N + BN - 1) / BN;
  const int total = gridDim.x;
  const int q8 = total >> 3, r8 = total & 7;
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  int tm = lid / nbn, tn = lid % nbn;
  if (g.skip_pad) {
    // An XCD's CONTIGUOUS share of the tiles is a run of utterances: the XCDs of the short ones would finish early and wait for the one that
    // holds the longest (measured: no gain at all).  The tiles are dealt round robin over the XCDs instead.  (Keeping the column tiles of a
    // row tile on one XCD -- they share the A patch -- measured worse: conv2 0.955 of the padded launch against 0.905, r06_packed_rows_ab.txt.)
    tm = bid / nbn;
    tn = bid % nbn;
  }
  if (GAM_SP_INSTRUMENT && g.tile_order > 0 && nbn % g.tile_order == 0 && !g.skip_pad) {   // (r05 experiment, instrumented builds only)
    // r05 experiment (GAM_SP_ORDER = G): groups of G column tiles OUTERMOST -- an XCD's contiguous share of the tile sequence
    // then stays inside one group, i.e. G x BN rows of W (G = 3, BN = 256, K = 768: 2.4 MB, resident in the XCD's 4 MB L2)
    // while the row tiles stream past; same tiles, same arithmetic: bit-identical results (profiles/r05_gemm_tile_order.txt)
    const int G = g.tile_order, nbm = (g.M + BM - 1) / BM;
    const int grp = lid / (nbm * G), rem = lid - grp * (nbm * G);
    tm = rem / G;
    tn = grp * G + rem % G;
  }
  const int m0 = tm * BM;
  const int n0 = tn * BN;
  if (g.skip_pad && g.lens != nullptr) {   // (uniform: the whole workgroup leaves before its first DMA and barrier)
    const int mlast = m0 + BM - 1 < g.M ? m0 + BM - 1 : g.M - 1;
    const int b0 = m0 / g.rpb;
    if (b0 == mlast / g.rpb && (m0 - b0 * g.rpb) / g.fdiv >= g.lens[b0]) return;
  }
  GAM_SP_TL(0);

  // ---- DMA sources.  Piece q of an operand = tile rows 8q .. 8q+7; this wave moves pieces
  //      q = wave + NWAVES i.  Lane l -> row 8q + (l>>3), LDS slot' l&7, source slot (l&7) ^ ((row>>1)&7).
  //      Addresses = wave-uniform tile base (SGPR pair) + 32-bit lane offset.
  auto a_row_off = [&](int m) -> size_t {
    m = m < g.M ? m : g.M - 1;
    if (g.a_mode == 0) return (size_t)m * (size_t)g.lda;
    const int fr = m / g.conv_f2, ff = m - fr * g.conv_f2;
    return ((size_t)fr * 2 * g.conv_fp + 2 * ff) * (size_t)g.conv_c;
  };
  const size_t a_tile0 = a_row_off(m0);   // offsets grow with m, so every row offset is >= this one
  const unsigned char* Ab = reinterpret_cast<const unsigned char*>(g.n_switch > 0 && n0 >= g.n_switch ? g.Asp2 : g.Asp) + a_tile0 * 4;
  const unsigned char* Wb = reinterpret_cast<const unsigned char*>(g.Wsp) + (size_t)n0 * (size_t)g.K * 4;
  unsigned a_src[NAI], w_src[NWI];
#pragma unroll
  for (int i = 0; i < NAI; ++i) {
    const int row = 8 * (wave + NWAVES * i) + (lane >> 3);
    const int slot = (lane & 7) ^ ((row >> 1) & 7);
    a_src[i] = (unsigned)((a_row_off(m0 + row) - a_tile0) * 4) + slot * 16;
  }
#pragma unroll
  for (int i = 0; i < NWI; ++i) {
    const int row = 8 * (wave + NWAVES * i) + (lane >> 3);
    const int slot = (lane & 7) ^ ((row >> 1) & 7);
    int n = n0 + row;
    n = n < g.N ? n : g.N - 1;
    w_src[i] = (unsigned)(n - n0) * (unsigned)g.K * 4u + slot * 16;
  }

  f32x16 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // split-K (grid.y = g.splitk slices, small grids: a few utterances per GPU): this workgroup contracts k-tiles
  // [kt_first, kt_first + nk) and leaves raw partial sums in g.partial[slice]; gam_splitk_reduce_kernel sums the slices in
  // a fixed order (bit-reproducible) and applies the epilogue.  One slice = the whole K = the plain kernel.
  const int nslice = g.splitk > 1 ? g.splitk : 1;
  const int nk = g.K / 32 / nslice;
  const int kt_first = (nslice > 1 ? (int)blockIdx.y : 0) * nk, kt_end = kt_first + nk;
  // next k-tile to fetch, tracked incrementally (no divisions in the loop); it stops at the slice's last tile
  int d_kt = kt_first, d_c0 = 0, d_kw = 0, d_kh = 0;
  if (g.a_mode != 0) {   // taps innermost: k-tile = 9 * (32-channel block) + 3 kh + kw
    const int cb = kt_first / 9, t9 = kt_first - cb * 9;
    d_c0 = 32 * cb; d_kh = t9 / 3; d_kw = t9 - d_kh * 3;
  }
  auto dma_ka = [&]() -> size_t {   // byte offset of tile d_kt along an A row
    if (g.a_mode == 0) return (size_t)d_kt * 128;
    return (((size_t)d_kh * g.conv_fp + d_kw) * (size_t)g.conv_c + d_c0) * 4;
  };
  // conv mode walks the taps INNERMOST: the nine k-tiles of a 32-channel block re-read the same (2 rows + 1) x (2 cols
  // + 1) pixel patch of the tile, shifted -- 139 KB per workgroup, L2-resident -- instead of streaming the whole
  // 768-channel image once per tap (the weight's sp32 copy is laid out in the same k-tile order, make_split)
  auto dma_advance = [&]() {
    if (d_kt + 1 >= kt_end) return;
    ++d_kt;
    if (g.a_mode == 0) return;
    if (++d_kw == 3) {
      d_kw = 0;
      if (++d_kh == 3) { d_kh = 0; d_c0 += 32; }
    }
  };
  auto issue = [&](int stage) {
    unsigned char* sb = gam_smem_sp + stage * Cfg::STAGE + wave * 1024;
    const unsigned char* ab = Ab + dma_ka();
    const unsigned char* wb = Wb + (size_t)d_kt * 128;
#pragma unroll
    for (int i = 0; i < NAI; ++i)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(ab + a_src[i]), (lds_ptr_t)(sb + i * (NWAVES * 1024)), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NWI; ++i)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(wb + w_src[i]), (lds_ptr_t)(sb + Cfg::A_BYTES + i * (NWAVES * 1024)), 16, 0, 0);
    dma_advance();
  };

  // ---- fragment addressing: lane -> row (lane&31), k-half kg = lane>>5; slot = plane*4 + 2*ks + kg
  const int sx = (lane >> 1) & 7;
  const int kg = lane >> 5;
  const int rowb = (lane & 31) * 128;
  const int o_h0 = rowb + (((0 + kg) ^ sx) << 4), o_h1 = rowb + (((2 + kg) ^ sx) << 4);
  const int o_l0 = rowb + (((4 + kg) ^ sx) << 4), o_l1 = rowb + (((6 + kg) ^ sx) << 4);
  const int a_base = wm * (BM / 2) * 128;
  const int w_base = Cfg::A_BYTES + wn * 64 * 128;

  // Two fragment sets (k16-step 0 / 1 of a k-tile), two MFMA phases per k-tile:
  //   phase 0: MFMAs on set 0, reading set 1 (same tile) in between
  //   vmcnt(0) + lgkmcnt(0) + barrier      -- tile kt+1 landed everywhere, stage kt&1 has no reader left
  //   phase 1: MFMAs on set 1, in between: read set 0 of tile kt+1, then DMA tile kt+2 -> stage kt&1
  // Everything that is not an MFMA is issued one or two items at a time BETWEEN MFMAs: an LDS-DMA piece
  // costs the issuing wave ~60+ cycles, and with the whole refill issued in one block after the barrier
  // both waves of a SIMD sat in it together and the matrix pipe idled ~600 of every 3500 cycles
  // (clock64 instrumentation, GAM_SP_INSTRUMENT build with GAM_SP_DBG=4).  Fragment reads get >= 2/3 of a phase to land.
  gam_half8 fah[2][MT], fal[2][MT], fbh[2][2], fbl[2][2];
  constexpr int NTERM = H16 ? 2 : 3;
  constexpr int NM = 2 * NTERM * MT;  // MFMAs per phase: terms x MT x 2 tiles
  constexpr int NR = 2 * MT + 4;    // fragment reads per set
  constexpr int NG = NAI + NWI;     // DMA pieces per wave per k-tile
  // first MFMA slot that carries a DMA piece: behind the reads where the phase is long enough (three terms), beside them otherwise
  constexpr int G0 = ((NR + 1) / 2 + NG <= NM) ? (NR + 1) / 2 : 0;
  static_assert((NR + 1) / 2 <= NM && G0 + NG <= NM, "phase too short for its reads + DMA pieces");
#define GAM_SPLD(P) (*reinterpret_cast<const gam_half8*>(P))
  // (plain ifs on unrolled loop counters, not nested generic lambdas: those push the fragment arrays to scratch)
#define GAM_SP_RDITEM(S, Q, ST, OH, OL)                                                           \
  {                                                                                               \
    const int q_ = (Q);                                                                           \
    if (q_ < 2) fbh[S][q_ & 1] = GAM_SPLD((ST) + w_base + (q_ & 1) * 4096 + (OH));                 \
    else if (q_ < MT + 2) fah[S][(q_ + MT - 2) % MT] = GAM_SPLD((ST) + a_base + ((q_ + MT - 2) % MT) * 4096 + (OH)); \
    else if (q_ < MT + 4) fbl[S][(q_ - MT) & 1] = GAM_SPLD((ST) + w_base + ((q_ - MT) & 1) * 4096 + (OL)); \
    else fal[S][(q_ + MT - 4) % MT] = GAM_SPLD((ST) + a_base + ((q_ + MT - 4) % MT) * 4096 + (OL)); \
  }
#define GAM_SP_MFMA(S, T)                                                                         \
  {                                                                                               \
    const int term_ = (T) / (2 * MT), i_ = ((T) % (2 * MT)) / 2, j_ = (T) & 1;                    \
    /* three terms: hi.hi, lo_w.hi_a, hi_w.lo_a;  H16: first-half x first-half, second-half x second-half */ \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term_ == 1 ? fbl[S][j_] : fbh[S][j_],    \
                                                         (H16 ? term_ == 1 : term_ == 2) ? fal[S][i_] : fah[S][i_], acc[i_][j_], 0, 0, 0); \
  }
  // phase<S, DMA>: MFMAs of set S; reads of set 1-S from (rst, roh, rol); DMA pieces of the next tile -> stage istage
  auto phase = [&](auto setc, auto dmac, const unsigned char* rst, int roh, int rol, int istage) {
    constexpr int S = decltype(setc)::value, R = 1 - S;
    constexpr bool DMA = decltype(dmac)::value;
    const unsigned char* ab = Ab;
    const unsigned char* wb = Wb;
    unsigned char* sb = gam_smem_sp;
    if constexpr (DMA) {
      ab = Ab + dma_ka();
      wb = Wb + (size_t)d_kt * 128;
      sb = gam_smem_sp + istage * Cfg::STAGE + wave * 1024;
      dma_advance();
    }
#pragma unroll
    for (int t = 0; t < NM; ++t) {
      GAM_SP_MFMA(S, t);
      if (2 * t < NR) GAM_SP_RDITEM(R, 2 * t, rst, roh, rol);
      if (2 * t + 1 < NR) GAM_SP_RDITEM(R, 2 * t + 1, rst, roh, rol);
      if (DMA && t >= G0 && t - G0 < NG) {
        const int gi = t - G0;
        if (gi < NAI)
          __builtin_amdgcn_global_load_lds((glb_ptr_t)(ab + a_src[gi % NAI]), (lds_ptr_t)(sb + gi * (NWAVES * 1024)), 16, 0, 0);
        else
          __builtin_amdgcn_global_load_lds((glb_ptr_t)(wb + w_src[(gi - NAI + NWI) % NWI]),
                                           (lds_ptr_t)(sb + Cfg::A_BYTES + (gi - NAI) * (NWAVES * 1024)), 16, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using C0_ = std::integral_constant<int, 0>;
  using C1_ = std::integral_constant<int, 1>;
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;

  long long t_bar = 0, t_mm0 = 0, t_mm1 = 0, t_start = 0, w_start = 0;
  if (GAM_SP_DBG(g) & 4) { t_start = clock64(); w_start = wall_clock64(); }
  // static priority for the later-dispatched half of the workgroup's waves (the arbitration loser on every phase:
  // MI355X_MICROARCH.md "Two waves per SIMD", item 4)
  if (GAM_SP_INSTRUMENT && g.prio && wave >= NWAVES / 2) __builtin_amdgcn_s_setprio(1);   // (r02 experiment, instrumented builds only)
  // in-loop wait: everything but the newest (NS - 2) k-tiles' DMA pieces of this wave has landed (vmcnt counts in issue order)
  constexpr int VMW = (NS - 2) * NG;
  static_assert(VMW < 64, "vmcnt field");
#pragma unroll
  for (int st_ = 0; st_ < NS; ++st_) issue(st_);   // (fewer k-tiles than stages: the last one is fetched again, unused)
  if constexpr (NS == 2) {
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();   // both tiles have landed for every wave
  } else {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VMW) : "memory");   // tiles 0 and 1 have landed for every wave
  }
  GAM_SP_TL(1);
#pragma unroll
  for (int q = 0; q < NR; ++q) GAM_SP_RDITEM(0, q, gam_smem_sp, o_h0, o_l0);
  const int nk_run = (GAM_SP_DBG(g) & 2) ? 1 : nk;
  int s_cur = 0;       // LDS stage of k-tile kt
  for (int kt = 0; kt < nk_run; ++kt) {
    const int s_nxt = s_cur + 1 == NS ? 0 : s_cur + 1;
    const unsigned char* st = gam_smem_sp + s_cur * Cfg::STAGE;
    const unsigned char* sn = gam_smem_sp + s_nxt * Cfg::STAGE;
    long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (GAM_SP_DBG(g) & 4) c0 = clock64();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): set 0 landed (read >= 2/3 of a phase ago)
    phase(C0_{}, F_{}, st, o_h1, o_l1, 0);
    if (GAM_SP_DBG(g) & 4) c1 = clock64();
    // vmcnt by hand: this wave's DMA pieces of tile kt+1 have landed (hipcc does not order an LDS-DMA against the
    // ds_reads behind a later barrier); lgkmcnt(0): set 1 is in registers
    if constexpr (NS == 2) {
      __builtin_amdgcn_s_waitcnt(0x0070);
      if (!(GAM_SP_DBG(g) & 8)) __syncthreads();
    } else {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VMW) : "memory");
    }
    if (GAM_SP_DBG(g) & 4) c2 = clock64();
    // Unconditional (one copy of the MFMA stream; a second, DMA-less copy behind a branch made hipcc
    // double-buffer the accumulators): past the end the set-0 reads fetch stale LDS that is never used
    // and the DMA re-fetches the last k-tile into a stage nobody reads again (drained before the epilogue).
    // The refill goes to the stage of tile kt: every wave has read its last fragments before the barrier above.
    phase(C1_{}, T_{}, sn, o_h0, o_l0, s_cur);
    if (GAM_SP_DBG(g) & 4) { c3 = clock64(); t_mm0 += c1 - c0; t_bar += c2 - c1; t_mm1 += c3 - c2; }
    s_cur = s_nxt;
  }
  if ((GAM_SP_DBG(g) & 4) && lane == 0 && (lid == 0 || lid == (int)gridDim.x - 1)) {
    // [total clk, total wall(100 MHz), phase 0, barrier, phase 1] of one wave
    float* d = g.C + (size_t)(g.M - 1) * g.ldc + (lid == 0 ? 0 : 64) + wave * 8;
    d[0] = (float)(clock64() - t_start); d[1] = (float)(wall_clock64() - w_start);
    d[2] = (float)t_mm0; d[3] = (float)t_bar; d[4] = (float)t_mm1;
    return;
  }
#undef GAM_SPLD
#undef GAM_SP_RDITEM
#undef GAM_SP_MFMA
  // every LDS-DMA piece of this wave has landed (a workgroup must not retire with DMA writes in flight: its LDS
  // could already belong to the next one); no barrier -- the epilogue below touches no shared memory
  __builtin_amdgcn_s_waitcnt(0x0070);
  GAM_SP_TL(2);

  const int mw = m0 + wm * (BM / 2), nw = n0 + wn * 64;
  if ((GAM_SP_DBG(g) & 1) && acc[0][0][0] != 123.456f) return;

  // ---- epilogue, straight from the accumulators.  The MFMAs run with the operands SWAPPED (W fragment as the A
  //      operand, activation fragment as B), i.e. they accumulate C^T tiles: lane l then owns ONE row of C
  //      (m = l & 31 of the 32 x 32 tile) and, per register quad g = r >> 2, FOUR CONSECUTIVE columns
  //      n = 8 g + 4 (l >> 5) + (r & 3).  Bias / residual / C therefore move as 16-byte vectors with no transpose:
  //      the former per-wave LDS round trip (96 ds_write_b32 + 24 ds_read_b128 per wave, 8 waves on one LDS) cost
  //      5.0 of the 7.3 us a 192 x 256 tile spent here (profiles/r03_gemm_timeline.txt); lanes l and l + 32 write
  //      adjacent 16-byte pieces of a row, the four quads complete its 128-byte line.
  // (r04, tried and dropped: requesting the bias pieces and the row scales of the 4-wave tiles in front of the first k-tile, so
  //  that the epilogue of a small-grid launch does not start with an L2 round trip -- 32 + MT more live registers, same-box A/B
  //  6.22-6.24 vs 6.24-6.25 ms at 4 x 20 s, 2.85 vs 2.89 ms for one clip: nothing; profiles/r04_ab_experiments.txt)
  // r04: EIGHT consecutive columns per lane.  Lanes l and l + 32 hold the two halves of every 8-column group of their (common) row;
  // one v_permlane32_swap per register trades the second half of the even group against the first half of the odd one, after
  // which lane l < 32 owns columns 16 p .. 16 p + 7 and lane l + 32 columns 16 p + 8 .. 16 p + 15 of each 32-column block: bias /
  // residual / C move as pairs of adjacent 16-byte pieces (64 contiguous bytes per row per instruction instead of 32), and an
  // sp32 / fp16 result as ONE 16-byte store per plane instead of two 8-byte ones (the store path, not the arithmetic, is what an
  // epilogue costs: profiles/r03_gemm_timeline.txt).  (semantics probed on the device, tools/permlane_probe.hip: after the instruction
  // vdst = [vdst(0..31) | vsrc(0..31)], vsrc = [vdst(32..63) | vsrc(32..63)].)  Same values, same roundings: bit-identical results.
  // Inline asm, not __builtin_amdgcn_permlane32_swap: hipcc 7.2 miscompiles a SEQUENCE of the builtin on vector elements (every
  // element comes back as element 0 -- reproduced in isolation by the probe).  The asm reads MFMA results and the hazard
  // recognizer does not look inside asm: two s_nop 15 cover the XDL-write -> VALU-read wait states of the last MFMAs.
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int pq = 0; pq < 2; ++pq)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[i][j][8 * pq + r], y = acc[i][j][8 * pq + 4 + r];
          asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
          acc[i][j][8 * pq + r] = x;
          acc[i][j][8 * pq + 4 + r] = y;
        }
  const int lrow = lane & 31, lq8 = (lane >> 5) * 8;
  const float accscale = g.wscale_inv;
  // piece (j, pq, hf): columns nw + 32 j + 16 pq + lq8 + 4 hf .. + 3  =  accumulator registers 8 pq + 4 hf .. + 3
  f32x4 bv[2][2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int pq = 0; pq < 2; ++pq)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int col = nw + 32 * j + 16 * pq + lq8 + 4 * hf;
        bv[j][pq][hf] = (g.bias != nullptr && col < g.N) ? *reinterpret_cast<const f32x4*>(g.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};   // N % 4 == 0
      }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = mw + 32 * i + lrow;
    if (row >= g.M) continue;
    bool masked = false, skip = false;
    long orow = row;
    if (g.lens != nullptr || g.remap) {
      const int bb = row / g.rpb, tt = row - bb * g.rpb;
      if (g.lens != nullptr) masked = (tt / g.fdiv) >= g.lens[bb];
      if (g.remap) {
        skip = tt >= g.rows_valid;
        orow = (long)bb * g.out_rpb + tt + g.out_shift;
      }
    }
    if (skip && g.partial == nullptr) continue;
    // per-row factor (weight scale x the A operand's row scale)
    const float rsc = accscale * (g.a_rs != nullptr ? g.a_rs[row] : 1.0f);
    if (g.partial != nullptr) {   // split-K slice: scaled raw sums; bias / activation / residual belong to the reduce pass
      float* P = g.partial + ((size_t)blockIdx.y * (size_t)g.M + (size_t)row) * (size_t)g.N;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pq = 0; pq < 2; ++pq)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int ecol = nw + 32 * j + 16 * pq + lq8 + 4 * hf, r0 = 8 * pq + 4 * hf;
            if (ecol < g.N)
              *reinterpret_cast<f32x4*>(P + ecol) = (f32x4){acc[i][j][r0], acc[i][j][r0 + 1], acc[i][j][r0 + 2], acc[i][j][r0 + 3]} * rsc;
          }
      continue;
    }
    f32x4 rv[2][2][2];
    if (g.R != nullptr) {   // all eight residual pieces of the row in flight before the first use
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pq = 0; pq < 2; ++pq)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int col = nw + 32 * j + 16 * pq + lq8 + 4 * hf;
            rv[j][pq][hf] = col < g.N ? *reinterpret_cast<const f32x4*>(g.R + orow * g.ldr + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
          }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int pq = 0; pq < 2; ++pq) {
        const int ecol = nw + 32 * j + 16 * pq + lq8;      // first of this lane's eight columns
        if (ecol >= g.N) continue;
        const bool both = ecol + 4 < g.N;                    // (N % 8 == 4: the last group has its first half only)
        f32x4 v[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int r0 = 8 * pq + 4 * hf;
          f32x4 t = (f32x4){acc[i][j][r0], acc[i][j][r0 + 1], acc[i][j][r0 + 2], acc[i][j][r0 + 3]};
          t = t * rsc + bv[j][pq][hf];
          if (ACT == GAM_ACT_SILU) { t.x = gam_silu(t.x); t.y = gam_silu(t.y); t.z = gam_silu(t.z); t.w = gam_silu(t.w); }
          if (ACT == GAM_ACT_RELU) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
          if (masked) t = (f32x4){0.f, 0.f, 0.f, 0.f};
          t = t * g.alpha;
          if (g.R != nullptr) t += rv[j][pq][hf];
          if (!both && hf == 1) t = (f32x4){0.f, 0.f, 0.f, 0.f};
          v[hf] = t;
        }
        if (g.c_guard) { gam_range_note(g.range_flag, v[0].x, v[0].y, v[0].z, v[0].w); gam_range_note(g.range_flag, v[1].x, v[1].y, v[1].z, v[1].w); }
        if ((GAM_SP_DBG(g) & 32) && v[0].x != 123.456f) continue;   // (experiment: the whole epilogue except its global stores)
        if (g.c_split == 2) {   // plain fp16 rows (the one-term mode's operand format): 8 halfs = one 16-byte store
          _Float16* cp = reinterpret_cast<_Float16*>(g.C) + orow * g.ldc + ecol;
          const gam_half4 h0 = __builtin_convertvector(v[0], gam_half4), h1 = __builtin_convertvector(v[1], gam_half4);
          if (both) *reinterpret_cast<gam_half8*>(cp) = (gam_half8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
          else *reinterpret_cast<gam_half4*>(cp) = h0;
        } else if (g.c_split) {
          _Float16* cp = reinterpret_cast<_Float16*>(g.C) + orow * (2 * g.ldc) + (ecol >> 5) * 64 + (ecol & 31);
          gam_half4 hi0, lo0, hi1, lo1;
          gam_split4(v[0], hi0, lo0);
          gam_split4(v[1], hi1, lo1);
          if (both) {
            *reinterpret_cast<gam_half8*>(cp) = (gam_half8){hi0[0], hi0[1], hi0[2], hi0[3], hi1[0], hi1[1], hi1[2], hi1[3]};
            *reinterpret_cast<gam_half8*>(cp + 32) = (gam_half8){lo0[0], lo0[1], lo0[2], lo0[3], lo1[0], lo1[1], lo1[2], lo1[3]};
          } else {
            *reinterpret_cast<gam_half4*>(cp) = hi0;
            *reinterpret_cast<gam_half4*>(cp + 32) = lo0;
          }
        } else {
          *reinterpret_cast<f32x4*>(g.C + orow * g.ldc + ecol) = v[0];
          if (both) *reinterpret_cast<f32x4*>(g.C + orow * g.ldc + ecol + 4) = v[1];
        }
      }
  }
#if GAM_SP_INSTRUMENT
  if (GAM_SP_DBG(g) & 16) { __builtin_amdgcn_s_waitcnt(0x0070); __syncthreads(); }   // stores issued AND acknowledged by every wave
  GAM_SP_TL(3);
#endif
}

// the vectorised epilogue moves 16-byte pieces of bias / R / C rows
static inline bool gam_gemm_sp_epilogue_ok(const GamGemmArgs& a) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return a.N % 4 == 0 && a.ldc % 4 == 0 && al16(a.C) && (a.bias == nullptr || al16(a.bias)) &&
         (a.R == nullptr || (a.ldr % 4 == 0 && al16(a.R)));
}

template <int ACT, int MT, int NW, int NS = 2, bool H16 = false>
static inline void gam_launch_gemm_sp_t(const GamGemmArgs& a, int grid, hipStream_t stream) {
  static std::atomic<unsigned long long> attr_devs{0};
  constexpr int smem = GamGemmSpCfg<MT, NW, NS>::SMEM;
  static_assert(smem <= 160 * 1024, "LDS stages do not fit a CU");
  auto kern = gam_gemm_sp_kernel<ACT, MT, NW, NS, H16>;
  if (gam_set_max_lds(reinterpret_cast<const void*>(kern), smem, attr_devs) != hipSuccess) return;
  hipLaunchKernelGGL(kern, dim3(grid, a.splitk > 1 ? a.splitk : 1), dim3(128 * NW), smem, stream, a);
}

// Tile shape and split-K factor.  NW = 4: (64 MT) x 256 tiles, 8 waves, one workgroup per CU -- fewest bytes per FLOP.
// NW = 2: (64 MT) x 128 tiles, 4 waves, up to two workgroups per CU.  S > 1: the K range is cut into S slices (grid.y) whose
// partial sums a second pass adds up -- for grids that would leave most of the chip idle (a few utterances per GPU: the
// strong-scaling points, single clips).  The plan minimises a time model fitted to the sweep of tools/smallm_sweep.py
// (profiles/r03_smallm_sweep.txt): the busiest CU runs ceil(workgroups / slots) workgroups of
//   nk / S k-tiles x t_kt(MT, NW, sharing) + t_fix(NW, S > 1),    plus, for S > 1, the reduce pass over (S + 1) M N floats.
struct GamSpPlan { int mt, nw, s, ns; };   // tile rows / 64, tile columns / 64, split-K slices, LDS stages
// tuning hook (gam_tune_sp / GAM_SP_MT, GAM_SP_NW, GAM_SP_SPLITK): 0 = free.  Process-wide by design (a sweep tool's knob), so it
// is race-free by construction: the environment is read once (thread-safe function-local static), the fields are atomics,
// and every handle's hipGraph key carries the three values (gam_api.hip), so a plan forced after a shape was captured
// cannot replay the old tiling.
struct GamSpForce {
  std::atomic<int> mt{0}, nw{0}, s{0}, ns{0};
  GamSpForce() {
    const char* e0 = getenv("GAM_SP_MT"); const char* e1 = getenv("GAM_SP_NW"); const char* e2 = getenv("GAM_SP_SPLITK");
    const char* e3 = getenv("GAM_SP_STAGES");
    mt = e0 ? atoi(e0) : 0; nw = e1 ? atoi(e1) : 0; s = e2 ? atoi(e2) : 0; ns = e3 ? atoi(e3) : 0;
  }
};
// tile classes with a three-stage instantiation: the 4-wave tiles (128-wide) and the 128 x 256 8-wave tile (3 x 48 KB)
static inline bool gam_sp_has_ns3(int mt, int nw) { return (nw == 2 && (mt == 2 || mt == 3)) || (nw == 4 && mt == 2); }
static inline GamSpForce& gam_sp_force() { static GamSpForce f; return f; }
static inline int gam_env_int_once(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
static inline double gam_gemm_sp_model(int M, int N, int K, int a_mode, int t, int w, int S, int ncu, int ns = 2) {
  // least-squares fit (log error) to the 810 points of profiles/r03_smallm_sweep.txt -- five layer shapes x six row counts x
  // every (MT, NW, S) -- r.m.s. 10 %, the planned configuration within 1 % of the best measured one on average (worst 11 %).
  // r04: the three-stage classes (ns = 3: one workgroup per CU whatever the tile, their LDS image is 96-144 KB) fitted to the
  // 1370-point sweep of profiles/r04_smallm_sweep_stages.txt (tools/fit_sp_model.py): r.m.s. 8 %, mean regret of the plan over
  // the 30 shapes 1.7 % (worst 12 %: M = 8032, N = 2304).
  const long wgs = (long)gam_cdiv(M, 64 * t) * gam_cdiv(N, 64 * w) * S;
  const int nkt = K / 32 / S;
  double t_kt, t_fix, rounds;
  if (ns == 3) {
    if (w == 4) { t_kt = 0.953; t_fix = S > 1 ? 6.62 : 12.8; }                 // 128 x 256, 8 waves
    else { t_kt = 0.229 * t + 0.115; t_fix = S > 1 ? 4.76 : 12.0; }            // (64 t) x 128, 4 waves
    rounds = (double)((wgs + ncu - 1) / ncu);
  } else if (w == 4) {
    t_kt = 0.335 * t + 0.425;                     // us per k-tile, one 8-wave workgroup per CU
    t_fix = S > 1 ? 6.2 : 13.2;                   // launch + prologue + epilogue (partial sums: no bias / residual / split)
    rounds = (double)((wgs + ncu - 1) / ncu);
  } else {
    const bool shared = wgs > ncu;                // two 4-wave workgroups on one CU
    const double tk_u = 0.256 * t + 0.142, tf_u = S > 1 ? 4.9 : 14.1;
    t_kt = tk_u * (shared ? 1.82 : 1.0);
    t_fix = tf_u * (shared ? 0.73 : 1.0);
    rounds = (double)((wgs + (shared ? 2 : 1) * ncu - 1) / ((shared ? 2 : 1) * ncu));
    // r06: a last round of at most ONE workgroup per CU runs unshared (756 tiles = one shared round of 512 + 244 alone): counted as a
    // whole shared round it made the three-stage class look better than it is -- the model's three worst picks (M = 8032 / N = 2304,
    // M = 12 017 / N = 1536, M = 5681 / N = 3072: 12-14 % slower than the best measured) were this; mean regret of the plan over the 36
    // shapes of profiles/r06_gemm_sweep_*.txt 1.4 % -> 0.3 %, worst 14 % -> 2 %; no pick changes at the headline's row count.
    const long cap2 = 2L * ncu, rem = wgs % cap2;
    if (shared && wgs > cap2 && rem != 0 && rem <= ncu) {
      (void)a_mode;
      double us2 = (double)(wgs / cap2) * (nkt * t_kt + t_fix) + (nkt * tk_u + tf_u);
      if (S > 1) us2 += 7.1 + (double)(S + 2) * M * N * 4.0 / 3.87e6;
      return us2;
    }
  }
  (void)a_mode;
  double us = rounds * (nkt * t_kt + t_fix);
  if (S > 1) us += 7.1 + (double)(S + 2) * M * N * 4.0 / 3.87e6;   // reduce pass: launch + partials, residual, C at ~3.9 TB/s
  return us;
}
static inline GamSpPlan gam_gemm_sp_plan(int M, int N, int K, int a_mode = 0, int ncu = 256) {
  GamSpForce& frc = gam_sp_force();
  const int f_mt = frc.mt.load(std::memory_order_relaxed), f_nw = frc.nw.load(std::memory_order_relaxed), f_s = frc.s.load(std::memory_order_relaxed);
  const int f_ns = frc.ns.load(std::memory_order_relaxed);
  GamSpPlan best = {3, 4, 1, 2};
  double bt = 1e30;
  const int nk = K / 32;
  for (int w = 4; w >= 2; w -= 2) {
    if ((f_nw == 2 || f_nw == 4) && w != f_nw) continue;
    for (int t = (w == 2 ? 3 : 4); t >= 2; --t) {
      if (f_mt >= 2 && f_mt <= 4 && t != f_mt && !(w == 2 && f_mt == 4)) continue;
      for (int S = 1; S <= 8; ++S) {
        if (f_s >= 1 && S != f_s) continue;
        if (S > 1 && (nk % S != 0 || nk / S < 4)) continue;     // whole k-tiles, and enough of them to fill the pipeline
        if (f_s < 1 && S > 1 && (long)gam_cdiv(M, 64 * t) * gam_cdiv(N, 64 * w) * 2 > ncu) continue;   // only for grids under half the chip
        for (int ns = 2; ns <= 3; ++ns) {
          if (ns == 3 && (!gam_sp_has_ns3(t, w) || a_mode != 0)) continue;   // (the implicit-GEMM conv runs the big two-stage tiles)
          if ((f_ns == 2 || f_ns == 3) && ns != f_ns && !(f_ns == 3 && !gam_sp_has_ns3(t, w))) continue;
          const double us = gam_gemm_sp_model(M, N, K, a_mode, t, w, S, ncu, ns);
          if (us < bt - 1e-9) { bt = us; best = {t, w, S, ns}; }
        }
      }
    }
  }
  return best;
}

static inline hipError_t gam_launch_gemm_sp(const GamGemmArgs& a_in, int act, hipStream_t stream) {
  GamGemmArgs a = a_in;
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  if (a.K <= 0 || a.K % 32 != 0 || a.Asp == nullptr || a.Wsp == nullptr) return hipErrorInvalidValue;
  if (a.a_mode == 0 ? (a.lda % 32 != 0) : (a.conv_c % 32 != 0)) return hipErrorInvalidValue;
  if (!gam_gemm_sp_epilogue_ok(a)) return hipErrorInvalidValue;
  // the caller made the plan (it owns the split-K workspace): a.sp_mt / a.sp_nw / a.splitk (+ a.partial when > 1)
  int mt = a.sp_mt, nw = a.sp_nw, ns = a.sp_ns;
  if (mt < 2 || mt > 4 || (nw != 2 && nw != 4) || (nw == 2 && mt == 4)) {
    const GamSpPlan p = gam_gemm_sp_plan(a.M, a.N, a.K, a.a_mode);
    mt = p.mt; nw = p.nw; ns = p.ns;
    if (a.splitk > 1 && a.partial == nullptr) return hipErrorInvalidValue;
  }
  if (ns != 3 || !gam_sp_has_ns3(mt, nw)) ns = 2;
  if (a.splitk > 1 && (a.partial == nullptr || (a.K / 32) % a.splitk != 0)) return hipErrorInvalidValue;
  if (a.n_switch > 0 && (a.Asp2 == nullptr || a.a_mode != 0 || a.n_switch % (64 * nw) != 0)) return hipErrorInvalidValue;
  if (a.splitk <= 1) { a.splitk = 0; a.partial = nullptr; }
  const int grid = gam_cdiv(a.M, 64 * mt) * gam_cdiv(a.N, 64 * nw);
  a.ntiles = grid;
  // the experiment switches GAM_SP_DBG / GAM_SP_PRIO / GAM_SP_ORDER exist in -DGAM_SP_INSTRUMENT=1 builds only: the production kernel
  // carries none of their branches and the launcher reads no environment variable for them
#if GAM_SP_INSTRUMENT
  static const int dbg = gam_env_int_once("GAM_SP_DBG"), prio = gam_env_int_once("GAM_SP_PRIO"), order = gam_env_int_once("GAM_SP_ORDER");   // (thread-safe one-time init)
#else
  constexpr int dbg = 0, prio = 0, order = 0;
#endif
  a.dbg = dbg;
  a.prio = prio;
  a.tile_order = order;
#if GAM_SP_INSTRUMENT
  static long long* tl_dev = nullptr;
  static int tl_cap = 0;
  if (dbg & 16) {
    if (grid > tl_cap) { if (tl_dev) (void)hipFree(tl_dev); (void)hipMalloc(&tl_dev, (size_t)grid * 64); tl_cap = grid; }
    (void)hipMemsetAsync(tl_dev, 0, (size_t)grid * 64, stream);
    a.tlog = tl_dev;
  }
#endif
#define GAM_LSP(ACTV)                                                            \
  if (a.h16) {   /* (K, lda, conv_c arrive halved: gam_api.hip gemm()) */        \
    if (ns == 3) {                                                               \
      if (nw == 4) gam_launch_gemm_sp_t<ACTV, 2, 4, 3, true>(a, grid, stream);   \
      else if (mt == 2) gam_launch_gemm_sp_t<ACTV, 2, 2, 3, true>(a, grid, stream); \
      else gam_launch_gemm_sp_t<ACTV, 3, 2, 3, true>(a, grid, stream);           \
    } else if (nw == 2) {                                                        \
      if (mt == 2) gam_launch_gemm_sp_t<ACTV, 2, 2, 2, true>(a, grid, stream);   \
      else gam_launch_gemm_sp_t<ACTV, 3, 2, 2, true>(a, grid, stream);           \
    } else switch (mt) {                                                         \
      case 2: gam_launch_gemm_sp_t<ACTV, 2, 4, 2, true>(a, grid, stream); break; \
      case 3: gam_launch_gemm_sp_t<ACTV, 3, 4, 2, true>(a, grid, stream); break; \
      default: gam_launch_gemm_sp_t<ACTV, 4, 4, 2, true>(a, grid, stream); break; \
    }                                                                            \
  } else if (ns == 3) {                                                          \
    if (nw == 4) gam_launch_gemm_sp_t<ACTV, 2, 4, 3>(a, grid, stream);           \
    else if (mt == 2) gam_launch_gemm_sp_t<ACTV, 2, 2, 3>(a, grid, stream);      \
    else gam_launch_gemm_sp_t<ACTV, 3, 2, 3>(a, grid, stream);                   \
  } else if (nw == 2) switch (mt) {                                              \
    case 2: gam_launch_gemm_sp_t<ACTV, 2, 2>(a, grid, stream); break;            \
    default: gam_launch_gemm_sp_t<ACTV, 3, 2>(a, grid, stream); break;           \
  } else switch (mt) {                                                           \
    case 2: gam_launch_gemm_sp_t<ACTV, 2, 4>(a, grid, stream); break;            \
    case 3: gam_launch_gemm_sp_t<ACTV, 3, 4>(a, grid, stream); break;            \
    default: gam_launch_gemm_sp_t<ACTV, 4, 4>(a, grid, stream); break;           \
  }
  switch (act) {
    case GAM_ACT_SILU: GAM_LSP(GAM_ACT_SILU); break;
    case GAM_ACT_RELU: GAM_LSP(GAM_ACT_RELU); break;
    default: GAM_LSP(GAM_ACT_NONE); break;
  }
#undef GAM_LSP
#if GAM_SP_INSTRUMENT
  if ((dbg & 16) && getenv("GAM_SP_TLOG")) {
    std::vector<long long> t((size_t)grid * 8);
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(t.data(), tl_dev, t.size() * 8, hipMemcpyDeviceToHost);
    long long t_first = t[0], t_last = 0;
    for (int i = 0; i < grid; ++i) { t_first = std::min(t_first, t[i * 8]); t_last = std::max(t_last, t[i * 8 + 3]); }
    auto stat = [&](auto f, const char* name) {
      std::vector<double> v(grid);
      for (int i = 0; i < grid; ++i) v[i] = f(i);
      std::sort(v.begin(), v.end());
      fprintf(stderr, "    %-26s min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us\n", name, v[0], v[grid / 2], v[(size_t)(grid * 0.9)], v[grid - 1]);
    };
    fprintf(stderr, "[tlog] M=%d N=%d K=%d MT=%d NW=%d tiles=%d: first entry -> last exit %.2f us\n", a.M, a.N, a.K, mt, nw, grid, (t_last - t_first) * 0.01);
    stat([&](int i) { return (t[i * 8] - t_first) * 0.01; }, "entry after first entry");
    stat([&](int i) { return (t[i * 8 + 1] - t[i * 8]) * 0.01; }, "prologue (2 k-tiles land)");
    stat([&](int i) { return (t[i * 8 + 2] - t[i * 8 + 1]) * 0.01; }, "main loop");
    stat([&](int i) { return (t[i * 8 + 2] - t[i * 8 + 1]) * 0.01 / (a.K / 32); }, "  per k-tile");
    stat([&](int i) { return (double)(t[i * 8 + 6] - t[i * 8 + 5]) / std::max<long long>(1, t[i * 8 + 2] - t[i * 8 + 1]) * 100.0; }, "  shader MHz in loop");
    stat([&](int i) { return (t[i * 8 + 3] - t[i * 8 + 2]) * 0.01; }, "epilogue");
    stat([&](int i) { return (t_last - t[i * 8 + 3]) * 0.01; }, "idle after exit");
    double x_loop[8] = {0}; int x_n[8] = {0};
    for (int i = 0; i < grid; ++i) { const int x = (int)(t[i * 8 + 7] & 7); x_loop[x] += (t[i * 8 + 2] - t[i * 8 + 1]) * 0.01; ++x_n[x]; }
    fprintf(stderr, "    mean loop us by XCD:");
    for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f", x_n[x] ? x_loop[x] / x_n[x] : 0.0);
    fprintf(stderr, "\n");
  }
#endif
  return hipGetLastError();
}

// fp32 [rows, K] -> sp32 (row pitch 2*K halfs); 4 elements per thread.  rs (optional, [rows]): the row's 2^-e from
// gam_rowscale_kernel -- the values are stored multiplied by 2^e (gam_common.h gam_row_scale).
__global__ __launch_bounds__(256) void gam_to_sp32_kernel(const float* __restrict__ x, _Float16* __restrict__ y,
                                                          size_t n4, const float* __restrict__ rs, int K) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    const size_t e = i * 4;                       // flat element index; 32-element blocks are row-aligned
    if (rs != nullptr) v = v * (1.0f / rs[e / (size_t)K]);
    gam_half4 h, l;
    gam_split4(v, h, l);
    _Float16* p = y + (e >> 5) * 64 + (e & 31);
    *reinterpret_cast<gam_half4*>(p) = h;
    *reinterpret_cast<gam_half4*>(p + 32) = l;
  }
}

// fp32 [rows, K] -> plain fp16 rows (format 2 of gam_store4; the one-term mode's operand), row-scaled like gam_to_sp32_kernel
__global__ __launch_bounds__(256) void gam_to_h16_kernel(const float* __restrict__ x, _Float16* __restrict__ y, size_t n4,
                                                         const float* __restrict__ rs, int K) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    if (rs != nullptr) v = v * (1.0f / rs[(i * 4) / (size_t)K]);
    reinterpret_cast<gam_half4*>(y)[i] = __builtin_convertvector(v, gam_half4);
  }
}

// fp32 [rows, K] -> fp32 copy with row r multiplied by 2^e_r = 1 / rs[r] (the 128x128 kernel's scaled operand)
__global__ __launch_bounds__(256) void gam_scale_rows_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n4,
                                                             const float* __restrict__ rs, int K) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
    reinterpret_cast<f32x4*>(y)[i] = reinterpret_cast<const f32x4*>(x)[i] * (1.0f / rs[(i * 4) / (size_t)K]);
}

// rs[row] = 2^-e with max|row| 2^e in [2^7, 2^8): one wave per row of an fp32 [rows, K] matrix (row pitch lda)
__global__ __launch_bounds__(256) void gam_rowscale_kernel(const float* __restrict__ x, float* __restrict__ rs, int rows,
                                                           int K, long lda) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * lda;
  float m = 0.f;
  for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(xr[k]));
  m = gam_wave_max(m);
  float s_, inv_;
  gam_row_scale(m, s_, inv_);
  if (lane == 0) rs[row] = inv_;
}
