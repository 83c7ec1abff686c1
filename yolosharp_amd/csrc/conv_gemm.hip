// conv_gemm.hip -- GEMM-grade implicit-GEMM convolution for the wide layers (Cin >= 128) of the bf16 path, gfx950 (CDNA4).
//
// Same arithmetic and argument contract as conv_p2_kernel (conv.hip): Conv2d of Modules/Convs.cs:36-62 / Head.cs:47-50 in NHWC with
// weights [Cout][kh][kw][Cin], forward (stride 1, 2), stride-1 dgrad and the 1x1..2x2 phase convolutions of a stride-2 dgrad.
// The whole-Cin LDS patch of conv_p2_kernel is the right shape for <= 128 channels (YOLOv8n/s); from 160-640 channels (YOLOv8x,
// YOLOv11m: BASELINE configs 4 and 5) the patch no longer fits a useful tile and the layer is MFMA-bound, so it wants the
// classical blocked GEMM instead:
//
//   D[pixel m][cout n] = sum_k A[m][k] * W[n][k],   k = (kh, kw, ci),  A gathered on the fly from the NHWC activation
//
//  * workgroup tile BM x BN = (WM*MR*16) x (WN*NR*16) pixels x channels (128x160 for the 80-multiples of YOLOv8x, 128x128 for the
//    64-multiples of v11m, 256x80 / 256x64 for the narrow outputs), 4 waves, each wave MR x NR MFMA 16x16x32 tiles: 20 MFMAs per
//    9 fragment reads (460 B of LDS per MFMA -- conv_p2_kernel's wide-layer tiles read 1.2 KB per MFMA);
//  * K in tiles of 64 (one 128-byte line per tile row): both operands go global -> LDS by LDS DMA (global_load_lds_dwordx4),
//    16 bytes per lane and no VGPRs in between; every lane computes its own source offset into a buffer descriptor, which is where the
//    implicit-GEMM gather (tap offset; zero padding = an out-of-range offset, the hardware returns zeros) and the bank swizzle live.  2-4 LDS stages (GemmArgs::nstage): the DMA of tiles k+1 .. k+nstage-1 is in flight
//    while tile k is multiplied (s_waitcnt vmcnt(N) on the wave's own requests), one barrier per K-tile;
//  * LDS rows are 128 B, so a fragment read (16 rows x 16 B per lane quarter) would hit 2 of 16 bank groups; unit u of row r is
//    stored at slot u ^ ((r >> 1) & 7) -- applied to the SOURCE address of the DMA and to the ds_read address alike -> conflict-free;
//  * persistent grid (gx, N-tiles): a workgroup keeps its output-channel tile and walks M-tiles of its XCD's contiguous share, so
//    the workgroups that share an A tile (same x, different y) and the neighbouring tiles' halo rows meet in one XCD's L2; the BN
//    batch statistics stay in registers across tiles and leave as one row per workgroup (p2 contract);
//  * epilogue, statistics, bias / eval-BN / SiLU, residual and gradient accumulation: the shared wide-store stage (conv_epi.h).
#include "conv_epi.h"
#include "conv_halo.h"
#include <atomic>
#ifndef YS_GEMM_EPI_DIRECT
#define YS_GEMM_EPI_DIRECT 0   // 1: this kernel's epilogue goes straight from the accumulator registers (16-byte stores after a 16-lane row swap, conv_epi.h
                               // p2_epilogue_direct) and the NEXT tile's first operand requests are issued before it (no LDS staging to alias the stages).
                               // Built and measured in round 4 (same box, alternating, config 2): conv_gemm_kernel 1.23 -> 1.27-1.28 ms/step -- a wave's
                               // direct store covers 16 pixel rows x 64 B (half lines), the staged form 8 rows x 128 B, and the store path is what this
                               // epilogue is bound by -- so the staged form stays
#endif
#ifndef YS_EPI_BATCH_GEMM
#define YS_EPI_BATCH_GEMM 16
#endif
#ifndef YS_GEMM_READ_AHEAD
#define YS_GEMM_READ_AHEAD 0   // measured (round 3, MI355X): no difference -- config 5 bf16 88.51 vs 88.57 ms/step, config 4 32.62 vs 32.71, config 2 10.24 vs 10.22; the K-tile is bound by LDS bytes (operand DMA + fragment reads ~220 KB per K-tile pair and CU), not by the exposed round trip
#endif
#include <cstdlib>

// ablation switches for performance triage (YS_GEMM_DBG bits; compiled in only with -DYS_GEMM_ABLATE = `build.py ablate`):
// 1 = A operand requests out of range, 2 = B operand requests out of range (zeros, no L2 traffic), 4 = no MFMAs, 8 = no epilogue;
// conv_halo_kernel also: 16 = no fragment reads and no MFMAs, 32 = no K-step barrier, 64 = no operand requests at all
#ifdef YS_GEMM_ABLATE
#define GEMM_DBG(bit) ((a.dbg & (bit)) != 0)
#else
#define GEMM_DBG(bit) false
#endif
struct GemmTap { int x, y; };   // per 16-byte K unit: offset (in units) from the tap-(0,0) pixel; kh << 8 | kw, or -1 = padding
struct GemmArgs {
  int nkt;          // K-tiles of 64
  int kunits;       // real 16-byte K units = KH*KW*Cin / 8
  int mtiles;       // ceil(M / BM)
  int off_stage;    // LDS byte offset of the two operand stages (the tap table sits at 0)
  int stage_bytes;  // (BM + BN) * 128
  int nstage;       // LDS stages of the operand pipeline (2..4): tile kt + nstage - 1 is requested while tile kt is multiplied
  int HoWo;
  YsFastDiv dHoWo, dWout;   // m -> (image, row, column) without integer divisions
  unsigned abytes;  // bytes of the input view from its first channel to the end of the last image (descriptor range, < 2^31)
};

// F8 = 1 (e4m3 input) / 2 (e5m2 input: a gradient): both operands are fp8 in memory (the input as the dense image ys_conv_launch quantises into a.q8 with the tensor's
// delayed scale, e4m3 or -- for a gradient -- e5m2; the weights as the e4m3 shadow a.w8).  Rows are still 128 B, now 128 K values:
// one v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales) per tile pair and K-tile, twice the bf16 MFMA rate at the same LDS and
// L2 bytes per instruction.  The fp32 accumulators are scaled back by a.deq in front of the shared epilogue.
template <int WM, int WN, int MR, int NR, int F8, int RED = 0>
__global__ void __launch_bounds__(256, 2)
conv_gemm_kernel(ConvArgs a, GemmArgs g) {
  typedef bf16_t T;
  constexpr int EPU = F8 ? 16 : 8;            // K elements per 16-byte unit
  constexpr int NT = 256;
  constexpr int BM = WM * MR * 16, BN = WN * NR * 16;
  constexpr int NA = BM / 32;                 // A pieces (8 rows x 128 B = one DMA instruction) per wave and K-tile
  constexpr int NBP = BN / 8;                 // B pieces per K-tile in the workgroup
  constexpr int NB = (NBP + 3) / 4;           // per wave (the last one may be absent: BN = 80)
  static_assert(WM * WN == 4 && BM % 32 == 0 && BN % 8 == 0, "tile");
#ifdef YS_P2_TIMELINE
  // s_memtime stamps of wave 0 / lane 0 of every 37th workgroup (as conv_p2_kernel): entry, tap table written, then per tile
  // (addresses done + first K-tile requested, K loop done, epilogue done), last = exit
  int tl_n = 0;
  unsigned long long* tl_p = (a.tl && (blockIdx.x % 37) == 0 && blockIdx.y == 0 && threadIdx.x == 0) ? a.tl + (blockIdx.x / 37) * 64 : nullptr;
#define GTL_STAMP() do { if (tl_p && tl_n < 63) tl_p[1 + tl_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define GTL_STAMP() ((void)0)
#endif
  GTL_STAMP();
  YS_DYN_LDS(lds);
  char* lb = (char*)lds;
  GemmTap* sTab = (GemmTap*)lb;                     // [nkt * 8] per K unit: (16-byte unit offset from the tap-(0,0) pixel, kh << 8 | kw) or (0, -1)
  char* sStage = lb + g.off_stage;

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, q = lane >> 4;
#ifdef YS_EMU_BUILD
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int wm = wave / WN, wn = wave - wm * WN;
  const int n0 = blockIdx.y * BN;
  const char* xb = (const char*)(F8 ? a.x8 : a.x);
  const char* wb = (const char*)(F8 ? a.w8 : a.w);
  const int cu = a.Cin / EPU;
  const int ldu = F8 ? cu : (a.in_ldc >> 3), cofu = F8 ? 0 : (a.in_coff >> 3);     // the fp8 image is dense: [pixel][Cin]
  const long bstride = F8 ? (long)a.Hin * a.Win : a.in_bstride;
  const long Kbytes = (long)g.kunits * 16;    // bytes per weight row

  for (int e = tid; e < g.nkt * 8; e += NT) {
    GemmTap t;
    if (e < g.kunits) {
      const int tap = e / cu, c8 = e - tap * cu;
      const int kh = tap / a.KW, kw = tap - kh * a.KW;
      t.x = ((kh * a.Win + kw) * ldu + c8) * 16;       // byte offset from the tap-(0,0) pixel
      t.y = (kh << 8) | kw;
    } else { t.x = 0; t.y = -1; }
    sTab[e] = t;
  }
  GTL_STAMP();

  // DMA roles.  Piece p of a stage = rows 8p .. 8p+7 (1 KB, one wave-wide instruction); wave w issues pieces w, w+4, ...  Lane l
  // of a piece lands at row 8p + (l >> 3), slot l & 7, and therefore fetches unit (l & 7) ^ ((row >> 1) & 7) of that row -- the
  // same for all of this thread's pieces since 8 * 4j / 2 is a multiple of 8.
  const int rsub = lane >> 3;
  const int kunit = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  // Requests go through buffer descriptors (ys_bufld_lds16): the weight rows need no per-lane work at all -- a lane's offset
  // (row n, unit kunit) never changes, the K-tile is the scalar offset, rows past Cout and the overrun of the last row are out of
  // range = zeros; an activation unit is one 32-bit add of the tap offset, padding = the out-of-range offset.  (With 64-bit
  // per-lane addresses and a zero line a request cost ~14 VALU instructions, ~100 cycles of the wave's time.)
  const ys_rsrc_t rsB = ys_make_rsrc(wb, (unsigned)((long)a.Cout * Kbytes));
  const ys_rsrc_t rsA = ys_make_rsrc(xb + ((long)cofu << 4), g.abytes);
  unsigned boff[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) {
    const int n = n0 + 8 * (wave + 4 * j) + rsub;
    boff[j] = (wave + 4 * j < NBP && n < a.Cout && !GEMM_DBG(2)) ? (unsigned)((long)n * Kbytes) + (unsigned)kunit * 16u : YS_BUF_OOB;
  }
  int abase[NA];                              // byte offset of the row's tap-(0,0) pixel from the descriptor base (views < 2 GB)
  int aiy[NA], aix[NA];

  auto issue = [&](int st, int kt) {
    char* sb = sStage + st * g.stage_bytes;
    const int ku = kt * 8 + kunit;
    const GemmTap te = sTab[ku];
    const int kh = te.y >> 8, kw = te.y & 255;
#pragma unroll
    for (int j = 0; j < NA; j++) {
      const bool ok = (bool)((int)(te.y >= 0) & (int)((unsigned)(aiy[j] + kh) < (unsigned)a.Hin) & (int)((unsigned)(aix[j] + kw) < (unsigned)a.Win) & (int)!GEMM_DBG(1));
      ys_bufld_lds16(rsA, ok ? (unsigned)(abase[j] + te.x) : YS_BUF_OOB, 0u, sb + (wave + 4 * j) * 1024);
    }
#pragma unroll
    for (int j = 0; j < NB; j++)
      if (wave + 4 * j < NBP) ys_bufld_lds16(rsB, boff[j], (unsigned)kt * 128u, sb + BM * 128 + (wave + 4 * j) * 1024);
  };
  constexpr int PER = NA + NBP / 4;           // requests per K-tile that EVERY wave issues (BN = 80: two waves issue one more; waiting for fewer is conservative)

  // fragment read offsets: row (16-row fragment base + li); bf16: unit (ks * 4 + q) ^ (li >> 1) of K-step ks
  // fp8: a lane quarter's 32 bytes are units q and 4 + q as well (not 2q, 2q + 1): which 32 of the tile's 128 K values a quarter
  // multiplies is free as long as both operands agree, and with 2q / 2q + 1 the eight lanes of quarter q + 1 that the LDS serves together
  // with eight of quarter q landed on the same 16-byte slots (SQ_LDS_BANK_CONFLICT 0.36 of LDS-active cycles in the fp8 kernel, 0.08 in bf16)
  const int koff0 = (q ^ (li >> 1)) << 4, koff1 = ((4 + q) ^ (li >> 1)) << 4;
  const int arow0 = ((wm * MR) * 16 + li) * 128;
  const int brow0 = BM * 128 + ((wn * NR) * 16 + li) * 128;

  // tile order: as conv_p2_kernel -- workgroup i runs on XCD i % 8 and walks that XCD's contiguous share of the M-tiles
  const bool xcd_order = (gridDim.x & 7) == 0;
  const int t_per_xcd = (g.mtiles + 7) >> 3;
  const int t_step = xcd_order ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int t_first = xcd_order ? (int)(blockIdx.x & 7) * t_per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_end = xcd_order ? (((int)(blockIdx.x & 7) + 1) * t_per_xcd < g.mtiles ? ((int)(blockIdx.x & 7) + 1) * t_per_xcd : g.mtiles) : g.mtiles;

  constexpr int NS = YS_GEMM_EPI_DIRECT ? 4 * NR : 8;
  float st1[NS], st2[NS];
#pragma unroll
  for (int e = 0; e < NS; e++) { st1[e] = 0.f; st2[e] = 0.f; }

  // row coordinates of a tile's A pieces + the requests of its first nstage - 1 K-tiles
  auto open_tile = [&](const int tile) {
    const int m0 = tile * BM;
#pragma unroll
    for (int j = 0; j < NA; j++) {
      const int m = m0 + 8 * (wave + 4 * j) + rsub;
      if (m < a.M) {
        const int b = (int)ys_fastdiv((unsigned)m, g.dHoWo), rem = m - b * g.HoWo;
        const int oy = (int)ys_fastdiv((unsigned)rem, g.dWout), ox = rem - oy * a.Wout;
        aiy[j] = oy * a.SA - a.PAD; aix[j] = ox * a.SA - a.PAD - a.pad_w_delta;
        abase[j] = (int)((((long)b * bstride + (long)aiy[j] * a.Win + aix[j]) * ldu) << 4);
      } else { aiy[j] = -(1 << 20); aix[j] = 0; abase[j] = 0; }
    }
    ys_barrier_lds();                         // the tap table is written; every wave is done with the stages (and, staged epilogue, with the staging area)
    for (int p = 0; p < g.nstage - 1 && p < g.nkt; p++) issue(p, p);
  };
  // Direct epilogue (YS_GEMM_EPI_DIRECT): a tile is OPENED before the previous tile's epilogue runs -- its first K-tiles are in flight
  // while the accumulators are rounded, reduced into the statistics and stored (~8 thousand cycles per 128 x 128 tile in the round-3
  // stamps, during which the operand pipeline was empty, and then another DMA round trip before the first MFMA).  The epilogue's stores
  // are younger than those requests, so the first waits of the next K loop also wait for them -- which the staged form did anyway.
  constexpr bool EARLY = YS_GEMM_EPI_DIRECT != 0;
  if (EARLY && t_first < t_end) open_tile(t_first);
  for (int tile = t_first; tile < t_end; tile += t_step) {
    const int m0 = tile * BM;
    if (!EARLY) open_tile(tile);
    GTL_STAMP();
    f32x4 acc[MR][NR];
#pragma unroll
    for (int mf = 0; mf < MR; mf++)
#pragma unroll
      for (int nf = 0; nf < NR; nf++) acc[mf][nf] = f32x4_zero();

    int stage_cur = 0, stage_next = (g.nstage - 1 < g.nkt ? g.nstage - 1 : g.nkt) % g.nstage;
#pragma unroll 1
    for (int kt = 0; kt < g.nkt; kt++) {
      // this wave's DMA pieces of tile kt have landed: at most the requests of the `ahead` younger tiles are still in flight
      {
        const int left = g.nkt - 1 - kt, ahead = left < g.nstage - 2 ? left : g.nstage - 2;
        if (ahead >= 2) ys_wait_vm<2 * PER>(); else if (ahead == 1) ys_wait_vm<PER>(); else YS_WAIT_VM0();
      }
      ys_barrier_lds();                       // ... everybody's have, and everybody is done reading tile kt - 1's stage, which is requested next
      if (kt + g.nstage - 1 < g.nkt) { issue(stage_next, kt + g.nstage - 1); stage_next = stage_next + 1 == g.nstage ? 0 : stage_next + 1; }
      const char* sb = sStage + stage_cur * g.stage_bytes;
      stage_cur = stage_cur + 1 == g.nstage ? 0 : stage_cur + 1;
      if (GEMM_DBG(4)) continue;
      if (F8) {
        // 32-byte fragments: the pixel fragments stay live, the weight fragments come in two groups (all at once is 72 registers
        // of fragments next to 80 accumulators: spills)
        constexpr int NG = (NR + 1) / 2;
        uint4 fx[MR][2];
#pragma unroll
        for (int mf = 0; mf < MR; mf++) { fx[mf][0] = *(const uint4*)(sb + arow0 + mf * 2048 + koff0); fx[mf][1] = *(const uint4*)(sb + arow0 + mf * 2048 + koff1); }
#pragma unroll
        for (int gq = 0; gq < 2; gq++) {
          const int nlo = gq * NG, nhi = gq ? NR : NG;
          uint4 fw[NG][2];
#pragma unroll
          for (int nf = nlo; nf < nhi; nf++) { fw[nf - nlo][0] = *(const uint4*)(sb + brow0 + nf * 2048 + koff0); fw[nf - nlo][1] = *(const uint4*)(sb + brow0 + nf * 2048 + koff1); }
          YS_SCHED_FENCE();
          // the input format is a template parameter: a run-time branch around the MFMAs (even a uniform one) made the compiler keep
          // two copies of the 80 accumulator registers and spill the DMA addresses -- and every scratch reload carries an
          // s_waitcnt vmcnt(0) that drains the DMA in flight
#pragma unroll
          for (int nf = nlo; nf < nhi; nf++)
#pragma unroll
            for (int mf = 0; mf < MR; mf++) acc[mf][nf] = mfma_scale_16x16x128_f8<(F8 == 2)>(fw[nf - nlo][0], fw[nf - nlo][1], fx[mf][0], fx[mf][1], acc[mf][nf]);
          YS_SCHED_FENCE();
        }
        continue;
      }
      constexpr bool AHEAD = YS_GEMM_READ_AHEAD && !(WM == 4 && NR == 5 && !RED);   // 256x80 forward: 36 more live registers spill
      if constexpr (AHEAD) {
      // both K-steps' fragments are requested before the first MFMA: the second step's LDS round trip runs under the first step's
      // MFMAs (ds_reads return in order, the compiler waits with lgkmcnt(N)), one exposed round trip per K-tile instead of two
      uint4 fw[2][NR], fx[2][MR];
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const int ko = ks ? koff1 : koff0;
#pragma unroll
        for (int nf = 0; nf < NR; nf++) fw[ks][nf] = *(const uint4*)(sb + brow0 + nf * 2048 + ko);
#pragma unroll
        for (int mf = 0; mf < MR; mf++) fx[ks][mf] = *(const uint4*)(sb + arow0 + mf * 2048 + ko);
      }
      YS_SCHED_FENCE();
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int nf = 0; nf < NR; nf++)
#pragma unroll
          for (int mf = 0; mf < MR; mf++) acc[mf][nf] = ys_mma<T>(fw[ks][nf], fx[ks][mf], acc[mf][nf]);
      } else {
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const int ko = ks ? koff1 : koff0;
        uint4 fw[NR], fx[MR];
#pragma unroll
        for (int nf = 0; nf < NR; nf++) fw[nf] = *(const uint4*)(sb + brow0 + nf * 2048 + ko);
#pragma unroll
        for (int mf = 0; mf < MR; mf++) fx[mf] = *(const uint4*)(sb + arow0 + mf * 2048 + ko);
        YS_SCHED_FENCE();
#pragma unroll
        for (int nf = 0; nf < NR; nf++)
#pragma unroll
          for (int mf = 0; mf < MR; mf++) acc[mf][nf] = ys_mma<T>(fw[nf], fx[mf], acc[mf][nf]);
      }
      }
    }
    GTL_STAMP();
    if (EARLY) { if (tile + t_step < t_end) open_tile(tile + t_step); }   // (its barrier: every wave finished reading this tile's stages)
    else ys_barrier_lds();                    // every wave finished reading the stages: they become the epilogue staging area

    int orow[MR];                             // row indices / byte offsets of a launch fit 31 bits (conv_gemm_plan)
    bool pv[MR];
#pragma unroll
    for (int mf = 0; mf < MR; mf++) {
      const int m = m0 + (wm * MR + mf) * 16 + li;
      pv[mf] = m < a.M;
      const int mm = pv[mf] ? m : 0;
      const int b = (int)ys_fastdiv((unsigned)mm, g.dHoWo), rem = mm - b * g.HoWo;
      const int oy = (int)ys_fastdiv((unsigned)rem, g.dWout), ox = rem - oy * a.Wout;
      orow[mf] = b * (int)a.out_bstride + (a.out_rh ? (oy * a.out_rh + ox * a.out_rw + (int)a.out_r0) : (oy * a.Wout + ox));
    }
    if (F8) {                                 // back to real units: 1 / (input scale * weight scale)
      const float dq = a.deq[0];
#pragma unroll
      for (int mf = 0; mf < MR; mf++)
#pragma unroll
        for (int nf = 0; nf < NR; nf++)
#pragma unroll
          for (int r = 0; r < 4; r++) acc[mf][nf][r] *= dq;
    }
    char* stg = sStage + wave * (16 * MR * (NR * 16 + 8) * 2 + 16 * MR * 16);
#if YS_GEMM_EPI_DIRECT
    (void)stg;
    if (!GEMM_DBG(8)) p2_epilogue_direct<MR, NR, RED>(a, acc, orow, pv, n0 + wn * NR * 16, st1, st2);
#else
    if (!GEMM_DBG(8)) p2_epilogue<MR, NR, RED, YS_EPI_BATCH_GEMM>(a, acc, orow, pv, n0 + wn * NR * 16, stg, st1, st2);
#endif
    GTL_STAMP();
  }
#if YS_GEMM_EPI_DIRECT
  if (RED ? a.nred > 0 : a.stats != nullptr) p2_stats_flush_direct<NR, WM, WN>(a, n0, st1, st2, (float*)sStage, (long)blockIdx.x);
#else
  if (RED ? a.nred > 0 : a.stats != nullptr) conv_stats_flush_grid<NR, WM, WN>(a, n0, st1, st2, (float*)sStage, (long)blockIdx.x);
#endif
  GTL_STAMP();
#ifdef YS_P2_TIMELINE
  if (tl_p) tl_p[0] = (unsigned long long)tl_n;
#endif
}

// ------------------------------------------------------------------ host side
struct GemmPlan { int ok, wm, wn, mr, nr, gx, gy; size_t lds; GemmArgs g; int halo; HaloArgs h; };

// halo-patch plan of a 3x3 stride-1 bf16 layer (conv_halo_kernel); p.ok = 0 when the layer is not eligible.  p.nr = channel fragments per wave:
// workgroup tile 16 x 16 pixels x (2 * nr * 16) channels, one workgroup of 8 waves per CU
static GemmPlan conv_halo_plan(const ConvArgs& a) {
  GemmPlan p{};
  const bool off = YS_OPT_INT("GEMM_HALO", 1) == 0;   // A/B switch against the blocked kernel
  if (off || a.f8) return p;
  if (!(a.KH == 3 && a.KW == 3 && a.SA == 1 && a.PAD == 1 && a.DIVM == 0 && a.out_rh == 0 && a.pad_w_delta == 0 && a.Hout == a.Hin && a.Wout == a.Win)) return p;
  if (a.Cin < 64 || a.Cout < 64) return p;
  // channel tile 160 (80-multiples: YOLOv8x) or 128, least padding first; more than a quarter padding: the blocked kernel has 64 / 80-wide tiles
  const long pad5 = (long)ys_cdiv(a.Cout, 160) * 160, pad4 = (long)ys_cdiv(a.Cout, 128) * 128;
  p.nr = pad5 <= pad4 ? 5 : 4;
  p.wm = 2; p.wn = 2; p.mr = 8;
  const int bn = p.wn * p.nr * 16;
  p.gy = ys_cdiv(a.Cout, bn);
  if ((long)p.gy * bn * 4 > (long)a.Cout * 5) return p;
  HaloArgs h{};
  h.nchunk = ys_cdiv(a.Cin, 64);
  // (The plan must depend on the geometry alone: the fused BN-backward row counts are planned before the segments are attached -- a gate on a.nred / a.accumulate
  // sent planning and launch to different kernels with different grids.  Launches that accumulate / carry the reduction pay the LDS-staged epilogue once per tile
  // with nothing to hide it behind; with three chunks per tile -- YOLOv8x 160 -> 160 at 160 x 160 -- they are 293 us against the blocked kernel's 270.)
  // the pixel tiles must cover the maps without much waste and give every CU work: 16 x 16 tiles (MR = 8) when they fill; else 16 x 8 tiles (MR = 4: half the MFMAs per
  // weight byte landed, but 40 x 40 maps fill 15 of them to 83 % -- 480 tile-jobs per 32 images -- where nine 16 x 16 tiles fill 69 % and make 288)
  const int min_fill = (int)YS_OPT_INT("HALO_MIN_FILL", 75);   // per cent
  const int mr4_opt = (int)YS_OPT_INT("HALO_MR4", 1);          // 0: never, 1: when 16 x 16 tiles do not fill, 2: whenever 16 x 8 tiles fill (tests)
  const bool mr4_on = mr4_opt != 0;
  h.tiles_x = ys_cdiv(a.Wout, 16); h.tiles_y = ys_cdiv(a.Hout, 16);
  if (mr4_opt == 2 || (long)a.Hout * a.Wout * 100 < (long)h.tiles_x * h.tiles_y * 256 * min_fill) {
    h.tiles_y = ys_cdiv(a.Hout, 8); p.mr = 4;
    if (!mr4_on || (long)a.Hout * a.Wout * 100 < (long)h.tiles_x * h.tiles_y * 128 * min_fill) return p;
  }
  const long mt = (long)a.B * h.tiles_x * h.tiles_y;
  if (mt >= (1L << 24)) return p;
  h.mtiles = (int)mt;
  h.dTpi = ys_fastdiv_make((unsigned)(h.tiles_x * h.tiles_y)); h.dTx = ys_fastdiv_make((unsigned)h.tiles_x);
  {
    const long pix = (long)(a.B - 1) * a.in_bstride + (long)a.Hin * a.Win;
    const long ab = (pix * a.in_ldc - a.in_coff) * 2L;
    if (ab <= 0 || ab >= (1L << 31) || (long)a.Cout * 9 * a.Cin * 2 >= (1L << 31)) return p;   // 32-bit request offsets
    if ((long)a.B * a.out_bstride * a.out_ldc * 2L >= (1L << 31)) return p;                      // the epilogue's store offsets
    h.abytes = (unsigned)ab;
  }
  p.lds = halo_lds_bytes(p.nr, p.mr);
  const int ncu = ys_cu_count();
  long gx = (ncu / p.gy) & ~7L;               // one workgroup per CU; the XCD-ordered tile walk needs a multiple of 8
  if (gx < 8) gx = 8;
  if (gx > mt) gx = mt;
  if (mt * p.gy <= ncu) gx = mt;
  { const long cap = YS_OPT_INT("HALO_MAX_GRID", 0); if (cap > 0 && gx > cap) gx = cap; }   // tests: workgroups that walk several tiles on oracle-sized shapes
  p.gx = (int)gx;
  p.h = h; p.halo = 1; p.ok = 1;
  return p;
}

static GemmPlan conv_gemm_plan(const ConvArgs& a) {
  GemmPlan p{};
  // 160 (was 128 until the end of round 4): with 128 <= Cin < 160 only the wide-output 3x3 layers (below) stay here.  What moved to the patch kernel, per-launch
  // records of YOLOv8n B = 64: the dgrad of the fused tower input at P3 (cin144 -> cout64, M = 409600: 168 -> 110 us -- a 256 x 64 tile leaves this kernel one
  // workgroup per CU) and the 2x2 / 1x2 / 2x1 phases of the stride-2 dgrads with 128 gradient channels, which join their 1x1 phase in ONE grouped patch-kernel
  // launch (141 -> 78 us, 92 -> 49 us); step 8.85 -> 8.76 ms
  const int min_cin = (int)YS_OPT_INT("GEMM_MIN_CIN", 160);
  const int min_k = 256;
  const bool f8 = a.f8 != 0;
  if (f8 && (!a.x8 || !a.w8 || !a.deq || a.Cin % 16)) return p;
  const bool k3 = a.KH == 3 && a.KW == 3 && a.out_rh == 0;
  const bool phase = a.KH >= 1 && a.KH <= 2 && a.KW >= 1 && a.KW <= 2 && a.SA == 1 && a.out_rh != 0 && a.PAD == 0;
  const bool k1 = a.KH == 1 && a.KW == 1 && a.PAD == 0 && a.SA == 1 && a.out_rh == 0;
  // input-gradient phases of ConvTranspose2d(2, 2) (Proto.upsample, Block.cs:69): a strided 1x1 gather at (2 oh + dh, 2 ow + dw)
  const bool ctd = a.KH == 1 && a.KW == 1 && a.SA == 2 && a.out_rh == 0;
  if (!((k3 || phase || k1 || ctd) && a.DIVM == 0 && (a.SA == 1 || a.SA == 2) && (a.pad_w_delta == 0 || ctd))) return p;
  if (a.Cin % 8 || a.in_ldc % 8 || a.in_coff % 8 || a.out_ldc % 8 || a.out_coff % 8 || a.Cout % 8) return p;
  if (a.res && (a.res_ldc % 8 || a.res_coff % 8)) return p;
  const int taps = a.KH * a.KW;
  const long Ktot = (long)taps * a.Cin;
  const int min_m = (int)YS_OPT_INT("GEMM_MIN_M", 1024);
  // stride-2 3x3 layers with 64 <= Cin < 128 come here too (round 4): the whole-Cin patch of a stride-2 tile is four times the tile's area, so the patch
  // kernel falls back to 2 x 22-pixel tiles whose patch every output-channel column re-reads (64 -> 128 at 80 x 80: 71 us, 9.4x its floor)
  const int s2_min_cin = 64;
  const bool s2_narrow = k3 && a.SA == 2 && a.Cin >= s2_min_cin && !f8;
  // ... and stride-1 3x3 layers with 64 <= Cin < 128 whose output is wide (the fused Detect / Segment tower input, 64 -> 144): the patch kernel streams
  // 166 KB of weights per 256-pixel tile there
  const int wide_cout = 128;
  const bool wide_out = k3 && a.SA == 1 && a.Cin >= 64 && a.Cout >= wide_cout && !f8;
  // (round 6: 1x1 layers with 160 <= Cin <= 256 / 384 / 512 on the patch kernel instead -- config 2 8.58 -> 8.63 / 8.64 / 8.63 ms, config 3 9.37 -> 9.51 / 9.55 / 9.78, config 4 29.5 -> 29.8 / 29.8 / 31.2: the gate stays)
  if ((a.Cin < min_cin && !s2_narrow && !wide_out) || Ktot < min_k || a.Cout < 64 || a.M < min_m) return p;
  if (k3 && a.SA == 1 && !f8) { const GemmPlan hp = conv_halo_plan(a); if (hp.ok) return hp; }
  // output-channel tile: least padding first, then the widest (most reuse of the A tile)
  static const int cand[4][4] = {{2, 2, 4, 5}, {2, 2, 4, 4}, {4, 1, 4, 5}, {4, 1, 4, 4}};   // WM, WN, MR, NR
  int best = -1; long best_pad = 0;
  for (int i = 0; i < 4; i++) {
    const int bn = cand[i][1] * cand[i][3] * 16;
    const long pad = (long)ys_cdiv(a.Cout, bn) * bn;
    if (best < 0 || pad < best_pad) { best = i; best_pad = pad; }
  }
  p.wm = cand[best][0]; p.wn = cand[best][1]; p.mr = cand[best][2]; p.nr = cand[best][3];
  const int bm = p.wm * p.mr * 16, bn = p.wn * p.nr * 16;
  GemmArgs g{};
  g.kunits = (int)(Ktot / (f8 ? 16 : 8));
  g.nkt = (int)((Ktot + (f8 ? 127 : 63)) / (f8 ? 128 : 64));
  g.mtiles = ys_cdiv(a.M, bm);
  g.HoWo = a.Hout * a.Wout;
  g.dHoWo = ys_fastdiv_make((unsigned)g.HoWo); g.dWout = ys_fastdiv_make((unsigned)a.Wout);
  {
    const long pix = f8 ? (long)a.B * a.Hin * a.Win : ((long)(a.B - 1) * a.in_bstride + (long)a.Hin * a.Win);
    const long ab = f8 ? pix * a.Cin : (pix * a.in_ldc - a.in_coff) * 2L;
    if (ab <= 0 || ab >= (1L << 31) || (long)a.Cout * Ktot * (f8 ? 1 : 2) >= (1L << 31)) return p;   // 32-bit request offsets
    if ((long)a.B * a.out_bstride * a.out_ldc * 2L >= (1L << 31)) return p;                             // the epilogue's store offsets
    g.abytes = (unsigned)ab;
  }
  const size_t tab = (size_t)g.nkt * 8 * sizeof(GemmTap);
  g.off_stage = (int)((tab + 1023) / 1024 * 1024);
  g.stage_bytes = (bm + bn) * 128;
  const size_t stage2 = (size_t)2 * g.stage_bytes;
  const size_t epi = (size_t)4 * (16 * p.mr * (p.nr * 16 + 8) * 2 + 16 * p.mr * 16);
  if (epi > stage2 || (size_t)16 * 256 * 4 > stage2) return p;
  if (g.off_stage + stage2 > 160 * 1024) return p;
  p.gy = ys_cdiv(a.Cout, bn);
  // Pipeline depth.  Two stages leave a request one K-tile of MFMA time (~500-1300 cycles) to land against ~1600 cycles of memory
  // latency (s_memtime stamps, round 3: 1600-1750 cycles per K-tile whatever the tile): a second workgroup on the CU covers the gap
  // when the grid has one; a launch with at most one workgroup per CU (the deep YOLOv8n layers at B = 64: 200 tiles) has nothing
  // else to run and takes as many stages as the LDS holds instead.
  g.nstage = 2;
  {
    int fit = (int)((160 * 1024 - (size_t)g.off_stage) / g.stage_bytes);
    if (fit > 4) fit = 4;
    if ((long)g.mtiles * p.gy <= ys_cu_count() && g.nkt > 2) g.nstage = fit;
    if (g.nstage > g.nkt) g.nstage = g.nkt < 2 ? 2 : g.nkt;
  }
  p.lds = g.off_stage + (size_t)g.nstage * g.stage_bytes;
  const int per_cu = p.lds <= 80 * 1024 ? 2 : 1;
  long gx = ((long)ys_cu_count() * per_cu) / p.gy;
  gx &= ~7L;                                  // XCD-ordered tile walk needs a multiple of 8
  if (gx < 8) gx = 8;
  if (gx > g.mtiles) gx = g.mtiles;
  if ((long)g.mtiles * p.gy <= ys_cu_count()) gx = g.mtiles;   // the whole launch fits the chip at one workgroup per CU: one tile each
  p.gx = (int)gx;
  p.g = g;
  p.ok = 1;
  return p;
}

template <int WM, int WN, int MR, int NR, int F8, int RED = 0>
static int conv_gemm_launch_t(hipStream_t st, ConvArgs a, const GemmPlan& p) {
  a.red_koff = (int)offsetof(ConvArgs, red);       // ConvArgs is the kernel's first argument (conv_epi.h ys_red_table)
#ifdef YS_GEMM_ABLATE
  a.dbg = getenv("YS_GEMM_DBG") ? atoi(getenv("YS_GEMM_DBG")) : 0;
#endif
  static std::atomic<unsigned> attr_done{0};      // per device: the attribute belongs to the device's code object
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  if (!(attr_done.load(std::memory_order_relaxed) & (1u << (dev_id & 31)))) {
    hipFuncSetAttribute((const void*)conv_gemm_kernel<WM, WN, MR, NR, F8, RED>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.fetch_or(1u << (dev_id & 31), std::memory_order_relaxed);
  }
  char lab[192] = "";
  if (ys_kprof_enabled()) snprintf(lab, sizeof(lab), F8 ? "gemmf8 k%d s%d div1 cin%d cout%d M%d acc%d tile%dx%d grid%dx%d lds%d" : "gemm k%d s%d div1 cin%d cout%d M%d acc%d tile%dx%d grid%dx%d lds%d", a.KH * 10 + a.KW, a.SA, a.Cin, a.Cout, a.M, a.accumulate, WM * MR * 16, WN * NR * 16, p.gx, p.gy, (int)p.lds);
  YsKprofScope prof(st, "conv_igemm", lab);
#ifdef YS_P2_TIMELINE
  static unsigned long long* tl_buf = nullptr;
  const char* tl_path = getenv("YS_P2_TL");
  if (tl_path) {
    if (!tl_buf) hipMalloc(&tl_buf, 64 * 64 * 8);
    hipMemsetAsync(tl_buf, 0, 64 * 64 * 8, st);
    a.tl = tl_buf;
  }
#endif
  YS_LAUNCH_LDS((conv_gemm_kernel<WM, WN, MR, NR, F8, RED>), dim3(p.gx, p.gy), 256, p.lds, st, a, p.g);
#ifdef YS_P2_TIMELINE
  if (tl_path) {
    static unsigned long long h[64 * 64];
    hipStreamSynchronize(st);
    hipMemcpy(h, tl_buf, sizeof(h), hipMemcpyDeviceToHost);
    FILE* f = fopen(tl_path, "a");
    if (f) {
      fprintf(f, "# gemm k%d%d s%d cin%d cout%d M%d acc%d tile%dx%d grid%dx%d lds%d mtiles%d nkt%d (stamps: table, then per tile: first request, K loop done, epilogue done; exit)\n", a.KH, a.KW, a.SA, a.Cin, a.Cout, a.M, a.accumulate, WM * MR * 16, WN * NR * 16, p.gx, p.gy, (int)p.lds, p.g.mtiles, p.g.nkt);
      for (int w = 0; w < 64 && w * 37 < p.gx; w++) {
        const int n = (int)h[w * 64];
        if (n <= 0) continue;
        fprintf(f, "wg%d:", w * 37);
        for (int i = 1; i < n; i++) fprintf(f, " %llu", h[w * 64 + 1 + i] - h[w * 64 + 1]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
#endif
  return YS_OK;
}

static int conv_halo_launch(hipStream_t st, const ConvArgs& a, const GemmPlan& p) {
  HaloLaunch l{p.gx, p.gy, p.nr, p.mr, p.lds, p.h};
  if (p.mr == 4) return p.nr == 5 ? ys_conv_halo_launch_nr5m(st, a, l) : ys_conv_halo_launch_nr4m(st, a, l);
  return p.nr == 5 ? ys_conv_halo_launch_nr5(st, a, l) : ys_conv_halo_launch_nr4(st, a, l);
}

int ys_conv_gemm_rows(const ConvArgs& a) {
  const GemmPlan p = conv_gemm_plan(a);
  return p.ok ? p.gx : 0;
}

int ys_conv_gemm_launch(hipStream_t st, const ConvArgs& a) {
  const GemmPlan p = conv_gemm_plan(a);
  if (!p.ok) return YS_ERR_UNSUPPORTED;
  if (p.halo) return conv_halo_launch(st, a, p);
  if (a.f8 == 1 && a.nred > 0) { ys_set_error("conv gemm: the fused BN-backward reduction belongs to dgrad launches (e5m2 input)"); return YS_ERR_UNSUPPORTED; }
  // RED variants: bf16 and e5m2-input (fp8-mode dgrad) launches that carry BN-backward segments
#define GM(A_, B_, C_, D_) if (p.wm == A_ && p.wn == B_ && p.mr == C_ && p.nr == D_) return a.f8 == 2 ? (a.nred > 0 ? conv_gemm_launch_t<A_, B_, C_, D_, 2, 1>(st, a, p) : conv_gemm_launch_t<A_, B_, C_, D_, 2>(st, a, p)) : (a.f8 ? conv_gemm_launch_t<A_, B_, C_, D_, 1>(st, a, p) : ((a.nred > 0 || (a.accumulate && YS_GEMM_EPI_DIRECT)) ? conv_gemm_launch_t<A_, B_, C_, D_, 0, 1>(st, a, p) : conv_gemm_launch_t<A_, B_, C_, D_, 0>(st, a, p)));
  GM(2, 2, 4, 5) GM(2, 2, 4, 4) GM(4, 1, 4, 5) GM(4, 1, 4, 4)
#undef GM
  return YS_ERR_UNSUPPORTED;
}
