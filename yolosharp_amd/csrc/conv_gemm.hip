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

// ------------------------------------------------------------------ halo-patch form for the 3x3 stride-1 layers (round 5)
// conv_gemm_kernel gathers the A operand of a 3x3 layer NINE times (once per tap: 36 KB of LDS DMA per K-tile of its 128 x 160 tile against 640 cycles
// of MFMA), every wave both requests and multiplies, and the ring is two K-tiles deep.  Here ONE workgroup of 8 waves per CU owns a 2-D tile of 16 x 16
// output pixels x BN = 160 / 128 output channels (4 x 2 waves, each the blocked kernel's 64 x 80 / 64 x 64 register tile) and stages the INPUT PATCH of
// the tile -- 18 x 18 pixels, one 64-channel chunk at a time, 128-byte rows -- ONCE per chunk; the nine taps of the chunk read their pixel fragments
// from that one patch at shifted rows.  Only the weights stream per tap (BN rows x 128 B).  Per tap (= 2 K-steps = 40 MFMAs per wave, 1280 MFMA cycles
// per SIMD) 5 KB of patch + 20 KB of weights land: ~20 B/clk/CU against the blocked kernel's 56, a wave issues ~3 one-KB requests per 40 MFMAs
// instead of 18.
//   * 128-byte rows on purpose: round-5 probe (tools/probe/probe_dma_rate.hip): LDS-DMA pieces that gather 64-byte row segments (16 cache lines per
//     instruction, half of each used) saturate at 13-21 B/clk/CU with 8 waves per CU requesting, full-line pieces reach 54; a first version of
//     this kernel with 32-channel chunks (64-byte rows, two 256-thread workgroups per CU) sat exactly on that limit.
//   * LDS: two patch buffers (chunk c is multiplied while chunk c + 1 lands, 2 x 45 KB) + a three-stage weight ring (stage = tap, slot = tap % 3);
//     everything by LDS DMA through buffer descriptors (zero padding = out-of-range offsets), counted vmcnt waits with compile-time counts, one
//     LDS-only barrier per tap.
//   * the two waves of a SIMD run the same phase (one workgroup), so each wave overlaps its own LDS reads with its own MFMAs: two fragment sets,
//     the reads of K-step k + 1 are issued before the MFMAs of K-step k, and the barrier of tap t + 1 sits BETWEEN the two K-steps of tap t.
//   * bank conflicts: a tap shift moves a fragment's 16 pixels to an ARBITRARY patch offset, so the blocked kernel's swizzle (which relies on
//     16-row-aligned fragments) does not carry over.  Layout (tools/dev/r05/halo_bank_model.py checks every offset against the ds_read_b128 lane
//     groups of MI355X_MICROARCH.md): patch pixel p (row pitch 20 pixels) stores K-unit u at slot u ^ ((p >> 1) & 7); lane quarter q multiplies
//     K-unit ((q & 1) << 2) | ((q >> 1) << 1) | ks of the row in K-step ks (both operands agree, so any assignment is a valid dot product); MFMA
//     column li holds tile column li ^ ((li >> 1) & 4) (the third and fourth groups of four swap).  With these three the 16 lanes the LDS serves
//     together always hit 16 different 16-byte bank columns.  Weight rows (fragments 16-row aligned): slot u ^ ((n >> 1) & 7) ^ (4 * (((n >> 2) ^ (n >> 3)) & 1)).
//   * fragment addresses: with a 20-pixel pitch (p >> 1) & 7 of a shifted pixel depends only on (row & 3, column), so a lane keeps 12 offsets
//     (3 column shifts x 4 row classes); a read is offset + immediate (K-step 1: offset ^ 16).
//   * epilogue / statistics / fused BN-backward reduction: the shared wide-store stage (conv_epi.h), two fragment rows at a time, staged in the ring.
struct HaloArgs {
  int nchunk;                 // 64-channel chunks of the input: ceil(Cin / 64)
  int tiles_x, tiles_y, mtiles;
  YsFastDiv dTpi, dTx;        // tile index -> (image, tile row, tile column)
  unsigned abytes;            // descriptor range of the input view
};
#define HALO_PW 18            // patch columns: 16 + halo
#define HALO_PWP 20           // row pitch of the patch in pixels (see above)
#define HALO_PH 18            // patch rows
#define HALO_MR 8             // fragment rows (tile rows) per wave
#define HALO_EMR 1            // fragment rows per epilogue call (one: 3 store iterations per call, whose accumulate / BN-reduction operands are all requested ahead)
#define HALO_NPP (HALO_PH * HALO_PWP / 8)       // 1 KB DMA pieces (8 pixels x 128 B) per patch: 45
#define HALO_PATCH (HALO_NPP * 1024)
__host__ __device__ constexpr int halo_stage_bytes(int nr) { return 2 * nr * 16 * 128; }
__host__ __device__ constexpr int halo_wstg(int nr) { return 16 * HALO_EMR * (nr * 16 + 8) * 2 + 16 * HALO_EMR * 16; }
__host__ __device__ constexpr size_t halo_lds_bytes(int nr) { return (size_t)2 * HALO_PATCH + (size_t)3 * halo_stage_bytes(nr); }

template <int NR, int RED = 0>
__global__ void __launch_bounds__(256, 1)
conv_halo_kernel(ConvArgs a, HaloArgs g) {
  typedef bf16_t T;
  constexpr int WM = 2, WN = 2, MR = HALO_MR, EMR = HALO_EMR, NWV = 4;
  constexpr int TH = WM * MR, PH = HALO_PH;
  constexpr int NPP = HALO_NPP, NPW = (NPP + NWV - 1) / NWV, NPMIN = NPP / NWV;   // patch pieces per chunk: workgroup, wave (most / least): 45, 12, 11
  constexpr int PATCH = HALO_PATCH;
  constexpr int BN = WN * NR * 16;
  constexpr int NBP = BN / 8, NBW = NBP / NWV;                 // weight pieces (8 rows x 128 B) per tap: workgroup, wave
  constexpr int STAGE = halo_stage_bytes(NR);
  constexpr int NMF = MR * NR;                                 // MFMAs per K-step and wave
  static_assert(TH + 2 == PH && NPW == 12 && NPMIN == 11 && NBP % NWV == 0 && MR % EMR == 0 && (HALO_PH * HALO_PWP) % 8 == 0 && NMF >= 2 * (NR + MR) + 8, "halo pipeline");
  static_assert(NWV * halo_wstg(NR) <= 3 * STAGE && 16 * 256 * 4 <= 3 * STAGE, "epilogue staging / statistics scratch inside the ring");
#ifdef YS_P2_TIMELINE
  int tl_n = 0;
  unsigned long long* tl_p = (a.tl && (blockIdx.x % 37) == 0 && blockIdx.y == 0 && threadIdx.x == 0) ? a.tl + (blockIdx.x / 37) * 64 : nullptr;
#define HTL_STAMP() do { if (tl_p && tl_n < 63) tl_p[1 + tl_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define HTL_STAMP() ((void)0)
#endif
  HTL_STAMP();
  YS_DYN_LDS(lds);
  char* lb = (char*)lds;
  char* sRing = lb + 2 * PATCH;

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, q = lane >> 4;
#ifdef YS_EMU_BUILD
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int wm = wave / WN, wn = wave - wm * WN;
  const int n0 = blockIdx.y * BN;
  const int ldu = a.in_ldc >> 3;                      // 16-byte units per input pixel
  const long Kbytes = (long)9 * a.Cin * 2;            // bytes per weight row

  const ys_rsrc_t rsB = ys_make_rsrc(a.w, (unsigned)((long)a.Cout * Kbytes));
  const ys_rsrc_t rsA = ys_make_rsrc((const char*)a.x + ((long)(a.in_coff >> 3) << 4), g.abytes);

  // weight requests: piece bp = wave + 4j covers rows 8bp .. 8bp + 7 of the stage, lane l -> row 8bp + (l >> 3), slot l & 7
  unsigned boff[NBW];
#pragma unroll
  for (int j = 0; j < NBW; j++) {
    const int row = (wave + NWV * j) * 8 + (lane >> 3), n = n0 + row;
    const int u = (lane & 7) ^ ((row >> 1) & 7) ^ ((((row >> 2) ^ (row >> 3)) & 1) << 2);
    boff[j] = (n < a.Cout && !GEMM_DBG(2)) ? (unsigned)((long)n * Kbytes) + (unsigned)u * 16u : YS_BUF_OOB;
  }
  // fragment read offsets (K-step 0; K-step 1 = the same ^ 16).  Pixels: MFMA column li holds tile column xm; a shifted pixel (row r, column xm + kx) sits at
  // padded index r * 20 + xm + kx, whose swizzle term (2 (r & 3) + ((xm + kx) >> 1)) & 7 depends on the row only through r & 3 (the wave's first row 8 wm
  // is a multiple of 4)
  const int xm = li ^ ((li >> 1) & 4);
  const int uq0 = ((q & 1) << 2) | ((q >> 1) << 1);
  int offA[3][4];
#pragma unroll
  for (int kx = 0; kx < 3; kx++)
#pragma unroll
    for (int rc = 0; rc < 4; rc++) {
      const int xx = xm + kx;
      offA[kx][rc] = (wm * MR * HALO_PWP + xx) * 128 + ((uq0 ^ ((2 * rc + (xx >> 1)) & 7)) << 4);
    }
  const int offB = (wn * NR * 16 + li) * 128 + ((uq0 ^ ((li >> 1) & 7) ^ ((((li >> 2) ^ (li >> 3)) & 1) << 2)) << 4);

  // tile order: as conv_gemm_kernel -- workgroup i runs on XCD i % 8 and walks that XCD's contiguous share of the tiles
  const bool xcd_order = (gridDim.x & 7) == 0;
  const int t_per_xcd = (g.mtiles + 7) >> 3;
  const int t_step = xcd_order ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int t_first = xcd_order ? (int)(blockIdx.x & 7) * t_per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_end = xcd_order ? (((int)(blockIdx.x & 7) + 1) * t_per_xcd < g.mtiles ? ((int)(blockIdx.x & 7) + 1) * t_per_xcd : g.mtiles) : g.mtiles;

  float st1[8], st2[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { st1[e] = 0.f; st2[e] = 0.f; }

  for (int tile = t_first; tile < t_end; tile += t_step) {
    const int b = (int)ys_fastdiv((unsigned)tile, g.dTpi), trem = tile - b * (g.tiles_x * g.tiles_y);
    const int ty = (int)ys_fastdiv((unsigned)trem, g.dTx), tx = trem - ty * g.tiles_x;
    const int y0 = ty * TH, x0 = tx * 16;
    const int iy0 = y0 - 1, ix0 = x0 - 1;     // image coordinates of patch pixel (0, 0); its byte offset from the descriptor base (may be negative at the border)
    const int tb = (int)((((long)b * a.in_bstride + (long)iy0 * a.Win + ix0) * ldu) << 4);
    // patch piece j of this wave for chunk c into buffer buf: piece pp = wave + 4j covers padded patch pixels 8pp .. 8pp + 7, lane l -> pixel 8pp + (l >> 3), slot l & 7
    auto issue_p = [&](const int buf, const int c, const int j) {
      const int pp = wave + NWV * j;
      if (pp < NPP) {
        const int pl = pp * 8 + (lane >> 3);
        const int py = (pl * 3277) >> 16, px = pl - py * HALO_PWP;    // pl / 20 for pl < 400
        const int u = (lane & 7) ^ ((pl >> 1) & 7);
        const bool ok = (bool)((int)(px < HALO_PW) & (int)((unsigned)(iy0 + py) < (unsigned)a.Hin) & (int)((unsigned)(ix0 + px) < (unsigned)a.Win) &
                               (int)((c * 8 + u) * 8 < a.Cin) & (int)!GEMM_DBG(1));
        if (GEMM_DBG(64)) return;
        const int vo = tb + (((py * a.Win + px) * ldu + c * 8 + u) << 4);
        ys_bufld_lds16(rsA, ok ? (unsigned)vo : YS_BUF_OOB, 0u, lb + buf * PATCH + pp * 1024);
      }
    };
    auto issue_w1 = [&](const int slot, const int c, const int tap, const int j) {   // weight piece j of this wave for tap (c, tap)
      if (GEMM_DBG(64)) return;
      ys_bufld_lds16(rsB, boff[j], (unsigned)(tap * a.Cin + c * 64) * 2u, sRing + slot * STAGE + (wave + NWV * j) * 1024);
    };
    ys_barrier_lds();                         // every wave is done with the previous tile's staging area (the ring)
    // prologue: patch of chunk 0, weights of taps 0, 1, 2 (the whole ring)
#pragma unroll
    for (int j = 0; j < NPW; j++) issue_p(0, 0, j);
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
      for (int j = 0; j < NBW; j++) issue_w1(t, 0, t, j);
    HTL_STAMP();
    f32x4 acc[MR][NR];
#pragma unroll
    for (int mf = 0; mf < MR; mf++)
#pragma unroll
      for (int nf = 0; nf < NR; nf++) acc[mf][nf] = f32x4_zero();

    // ONE wave per SIMD (up to 512 registers): the wave hides its own LDS reads, operand requests and waits in the issue slots between its MFMAs.  Two fragment
    // sets: set A holds K-step 0 of a tap, set B K-step 1.  Iteration of tap s:
    //   K-step 0: 40 (32) MFMAs on A; between them the 13 (12) reads of (s, K-step 1) -> B
    //   s_waitcnt vmcnt(N) + barrier of tap s + 1 (its requests were issued two iterations ago); frees ring slot s % 3 and, at the last tap of a chunk, the patch buffer
    //   K-step 1: MFMAs on B; between them this tap's requests (patch pieces of the next chunk, weights of tap s + 3 -> slot s % 3), then the reads of (s + 1, K-step 0) -> A
    uint4 fwA[NR], fxA[MR], fwB[NR], fxB[MR];
    // MFMA i of a K-step = (nf, mf) = (i / MR, i % MR); hook(i) runs after it
    auto kstep = [&](const uint4 (&fw)[NR], const uint4 (&fx)[MR], auto hook) {
      ys_static_for<0, NMF>([&](auto ic) {
        constexpr int i = decltype(ic)::value, nf = i / MR, mf = i % MR;
        if (!GEMM_DBG(4)) acc[mf][nf] = ys_mma<T>(fw[nf], fx[mf], acc[mf][nf]);
        else acc[mf][nf][0] += ys_u2f(fw[nf].x) + ys_u2f(fx[mf].x);
        YS_SCHED_FENCE();
        hook(ic);
        YS_SCHED_FENCE();
      });
    };
    // read r (0 .. NR + MR - 1) of the fragments of (chunk patch offset table po, tap, K-step ks) into (fw, fx): weights first, then pixels
    auto frag_read = [&](auto rc_, auto tapc, auto ksc, uint4 (&fw)[NR], uint4 (&fx)[MR], const int pbo) {
      constexpr int r = decltype(rc_)::value, tap = decltype(tapc)::value, ks = decltype(ksc)::value;
      constexpr int ky = tap / 3, kx = tap - ky * 3;
      if (GEMM_DBG(16)) return;
      if constexpr (r < NR) fw[r] = *(const uint4*)(sRing + (tap % 3) * STAGE + (ks ? (offB ^ 16) : offB) + r * 2048);
      else {
        constexpr int mf = r - NR;
        const int o = offA[kx][(mf + ky) & 3];
        fx[mf] = *(const uint4*)(lb + pbo + (ks ? (o ^ 16) : o) + (mf + ky) * (HALO_PWP * 128));
      }
    };
    // before the loop: tap 0 landed, its K-step 0 fragments -> A
    ys_wait_vm<2 * NBW>();
    ys_barrier_lds();
    ys_static_for<0, NR + MR>([&](auto rc_) { frag_read(rc_, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fwA, fxA, 0); });
    HTL_STAMP();
#pragma unroll 1
    for (int c = 0; c < g.nchunk; c++) {
      const bool more = c + 1 < g.nchunk;
      const int pbo = (c & 1) * PATCH;        // patch buffer of this chunk
      ys_static_for<0, 9>([&](auto tc) {
        constexpr int tap = decltype(tc)::value;
        // ---- K-step 0 on A; reads (tap, K-step 1) -> B
        kstep(fwA, fxA, [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if constexpr (i < NR + MR) frag_read(ic, tc, std::integral_constant<int, 1>{}, fwB, fxB, pbo);
        });
        // ---- tap + 1 has landed (this wave's pieces: everything older than the requests of the previous iteration), everybody's have
        const bool last = !more && tap == 8;
        if (!last) {
          constexpr int np_prev = tap == 0 ? 0 : (tap - 1 < 3 ? 2 : 1);     // patch pieces every wave issued in the previous iteration (chunks with a successor)
          if (more) ys_wait_vm<np_prev + NBW>();
          else if constexpr (tap < 7) ys_wait_vm<NBW>(); else YS_WAIT_VM0();
          if (c == 1 && tile == t_first) HTL_STAMP();   // (triage builds: fine stamps of the second chunk of the first tile)
          if (!GEMM_DBG(32)) ys_barrier_lds();
          if (c == 1 && tile == t_first) HTL_STAMP();
        }
        // ---- K-step 1 on B; this tap's requests, then reads (tap + 1, K-step 0) -> A
        constexpr int tap3 = (tap + 3) % 9;
        const int c3 = c + (tap + 3) / 9;
        const bool w_on = c3 < g.nchunk;
        kstep(fwB, fxB, [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          // requests at MFMAs 0, 2, 4, ...: [patch piece(s) of chunk c + 1: two at taps 0 - 2, one after], weights of tap + 3
          constexpr int NPT = tap < 3 ? 2 : 1, JP0 = tap < 3 ? 2 * tap : tap + 3;        // patch pieces of this tap: JP0 .. JP0 + NPT - 1 (12 per chunk)
          if constexpr ((i & 1) == 0 && i / 2 < NPT) { if (more) issue_p((c + 1) & 1, c + 1, JP0 + i / 2); }
          else if constexpr ((i & 1) == 0 && i / 2 < NPT + NBW) { if (w_on) issue_w1(tap % 3, c3, tap3, i / 2 - NPT); }
          else if constexpr (i >= 2 * (NPT + NBW) && i < 2 * (NPT + NBW) + NR + MR) {
            constexpr int r = i - 2 * (NPT + NBW);
            if constexpr (tap < 8) frag_read(std::integral_constant<int, r>{}, std::integral_constant<int, tap + 1>{}, std::integral_constant<int, 0>{}, fwA, fxA, pbo);
            else { if (more) frag_read(std::integral_constant<int, r>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fwA, fxA, PATCH - pbo); }
          }
        });
        if (c == 1 && tile == t_first) HTL_STAMP();
      });
    }
    HTL_STAMP();
    ys_barrier_lds();                         // every wave finished reading the ring: it becomes the epilogue staging area

    char* stg = sRing + wave * halo_wstg(NR);
#pragma unroll
    for (int h = 0; h < MR / EMR; h++) {
      int orow[EMR];
      bool pv[EMR];
#pragma unroll
      for (int e = 0; e < EMR; e++) {
        const int oy = y0 + wm * MR + h * EMR + e, ox = x0 + xm;
        pv[e] = (bool)((int)(oy < a.Hout) & (int)(ox < a.Wout));
        orow[e] = pv[e] ? b * (int)a.out_bstride + oy * a.Wout + ox : 0;
      }
      f32x4 sub[EMR][NR];                     // (register moves the allocator coalesces; no address of acc is taken)
#pragma unroll
      for (int e = 0; e < EMR; e++)
#pragma unroll
        for (int nf = 0; nf < NR; nf++) sub[e][nf] = acc[h * EMR + e][nf];
      if (!GEMM_DBG(8)) p2_epilogue<EMR, NR, RED, 4>(a, sub, orow, pv, n0 + wn * NR * 16, stg, st1, st2);
    }
    HTL_STAMP();
  }
  if (RED ? a.nred > 0 : a.stats != nullptr) conv_stats_flush_grid<NR, WM, WN>(a, n0, st1, st2, (float*)sRing, (long)blockIdx.x);
  HTL_STAMP();
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
  static const bool on = getenv("YS_GEMM_HALO") && atoi(getenv("YS_GEMM_HALO")) != 0;   // (work in progress: opt-in until it beats the blocked kernel)
  if (!on || a.f8) return p;
  if (!(a.KH == 3 && a.KW == 3 && a.SA == 1 && a.PAD == 1 && a.DIVM == 0 && a.out_rh == 0 && a.pad_w_delta == 0 && a.Hout == a.Hin && a.Wout == a.Win)) return p;
  if (a.Cin < 64 || a.Cout < 64) return p;
  // channel tile 160 (80-multiples: YOLOv8x) or 128, least padding first; more than a quarter padding: the blocked kernel has 64 / 80-wide tiles
  const long pad5 = (long)ys_cdiv(a.Cout, 160) * 160, pad4 = (long)ys_cdiv(a.Cout, 128) * 128;
  p.nr = pad5 <= pad4 ? 5 : 4;
  p.wm = 2; p.wn = 2; p.mr = HALO_MR;
  const int bn = p.wn * p.nr * 16;
  p.gy = ys_cdiv(a.Cout, bn);
  if ((long)p.gy * bn * 4 > (long)a.Cout * 5) return p;
  HaloArgs h{};
  h.nchunk = ys_cdiv(a.Cin, 64);
  h.tiles_x = ys_cdiv(a.Wout, 16); h.tiles_y = ys_cdiv(a.Hout, 16);
  const long mt = (long)a.B * h.tiles_x * h.tiles_y;
  if (mt >= (1L << 24)) return p;
  // the pixel tiles must cover the maps without much waste (40 x 40: 9 tiles of 256 for 1600 pixels) and give every CU work
  static const int min_fill = getenv("YS_HALO_MIN_FILL") ? atoi(getenv("YS_HALO_MIN_FILL")) : 75;   // per cent
  if ((long)a.Hout * a.Wout * 100 < (long)h.tiles_x * h.tiles_y * 256 * min_fill) return p;
  h.mtiles = (int)mt;
  h.dTpi = ys_fastdiv_make((unsigned)(h.tiles_x * h.tiles_y)); h.dTx = ys_fastdiv_make((unsigned)h.tiles_x);
  {
    const long pix = (long)(a.B - 1) * a.in_bstride + (long)a.Hin * a.Win;
    const long ab = (pix * a.in_ldc - a.in_coff) * 2L;
    if (ab <= 0 || ab >= (1L << 31) || (long)a.Cout * 9 * a.Cin * 2 >= (1L << 31)) return p;   // 32-bit request offsets
    if ((long)a.B * a.out_bstride * a.out_ldc * 2L >= (1L << 31)) return p;                      // the epilogue's store offsets
    h.abytes = (unsigned)ab;
  }
  p.lds = halo_lds_bytes(p.nr);
  long gx = (256 / p.gy) & ~7L;               // one workgroup per CU; the XCD-ordered tile walk needs a multiple of 8
  if (gx < 8) gx = 8;
  if (gx > mt) gx = mt;
  if (mt * p.gy <= 256) gx = mt;
  p.gx = (int)gx;
  p.h = h; p.halo = 1; p.ok = 1;
  return p;
}

static GemmPlan conv_gemm_plan(const ConvArgs& a) {
  GemmPlan p{};
  static const bool off = getenv("YS_NO_GEMM") != nullptr;
  // 160 (was 128 until the end of round 4): with 128 <= Cin < 160 only the wide-output 3x3 layers (below) stay here.  What moved to the patch kernel, per-launch
  // records of YOLOv8n B = 64: the dgrad of the fused tower input at P3 (cin144 -> cout64, M = 409600: 168 -> 110 us -- a 256 x 64 tile leaves this kernel one
  // workgroup per CU) and the 2x2 / 1x2 / 2x1 phases of the stride-2 dgrads with 128 gradient channels, which join their 1x1 phase in ONE grouped patch-kernel
  // launch (141 -> 78 us, 92 -> 49 us); step 8.85 -> 8.76 ms
  static const int min_cin = getenv("YS_GEMM_MIN_CIN") ? atoi(getenv("YS_GEMM_MIN_CIN")) : 160;
  static const int min_k = getenv("YS_GEMM_MIN_K") ? atoi(getenv("YS_GEMM_MIN_K")) : 256;
  const bool f8 = a.f8 != 0;
  static const bool f8_off = getenv("YS_NO_GEMM_F8") != nullptr;
  if (off || (f8 && (f8_off || !a.x8 || !a.w8 || !a.deq || a.Cin % 16))) return p;
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
  static const int min_m = getenv("YS_GEMM_MIN_M") ? atoi(getenv("YS_GEMM_MIN_M")) : 1024;
  // stride-2 3x3 layers with 64 <= Cin < 128 come here too (round 4): the whole-Cin patch of a stride-2 tile is four times the tile's area, so the patch
  // kernel falls back to 2 x 22-pixel tiles whose patch every output-channel column re-reads (64 -> 128 at 80 x 80: 71 us, 9.4x its floor)
  static const int s2_min_cin = getenv("YS_GEMM_S2_MIN_CIN") ? atoi(getenv("YS_GEMM_S2_MIN_CIN")) : 64;
  const bool s2_narrow = k3 && a.SA == 2 && a.Cin >= s2_min_cin && !f8;
  // ... and stride-1 3x3 layers with 64 <= Cin < 128 whose output is wide (the fused Detect / Segment tower input, 64 -> 144): the patch kernel streams
  // 166 KB of weights per 256-pixel tile there
  static const int wide_cout = getenv("YS_GEMM_WIDE_COUT") ? atoi(getenv("YS_GEMM_WIDE_COUT")) : 128;
  const bool wide_out = k3 && a.SA == 1 && a.Cin >= 64 && a.Cout >= wide_cout && !f8;
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
  // else to run and takes as many stages as the LDS holds instead.  YS_GEMM_STAGES forces a depth (A/B runs).
  static const int force_st = getenv("YS_GEMM_STAGES") ? atoi(getenv("YS_GEMM_STAGES")) : 0;
  g.nstage = 2;
  {
    int fit = (int)((160 * 1024 - (size_t)g.off_stage) / g.stage_bytes);
    if (fit > 4) fit = 4;
    if ((long)g.mtiles * p.gy <= 256 && g.nkt > 2) g.nstage = fit;
    if (force_st >= 2 && force_st <= 4) g.nstage = force_st < fit ? force_st : fit;
    if (g.nstage > g.nkt) g.nstage = g.nkt < 2 ? 2 : g.nkt;
  }
  p.lds = g.off_stage + (size_t)g.nstage * g.stage_bytes;
  const int per_cu = p.lds <= 80 * 1024 ? 2 : 1;
  long gx = (256L * per_cu) / p.gy;
  gx &= ~7L;                                  // XCD-ordered tile walk needs a multiple of 8
  if (gx < 8) gx = 8;
  if (gx > g.mtiles) gx = g.mtiles;
  if ((long)g.mtiles * p.gy <= 256) gx = g.mtiles;   // the whole launch fits the chip at one workgroup per CU: one tile each
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

template <int NR, int RED>
static int conv_halo_launch_t(hipStream_t st, ConvArgs a, const GemmPlan& p) {
  a.red_koff = (int)offsetof(ConvArgs, red);       // ConvArgs is the kernel's first argument (conv_epi.h ys_red_table)
#ifdef YS_GEMM_ABLATE
  a.dbg = getenv("YS_GEMM_DBG") ? atoi(getenv("YS_GEMM_DBG")) : 0;
#endif
  static std::atomic<unsigned> attr_done{0};      // per device: the attribute belongs to the device's code object
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  if (!(attr_done.load(std::memory_order_relaxed) & (1u << (dev_id & 31)))) {
    hipFuncSetAttribute((const void*)conv_halo_kernel<NR, RED>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.fetch_or(1u << (dev_id & 31), std::memory_order_relaxed);
  }
  char lab[192] = "";
  if (ys_kprof_enabled()) snprintf(lab, sizeof(lab), "halo k33 s1 div1 cin%d cout%d M%d acc%d tile%dx16x%d grid%dx%d lds%d", a.Cin, a.Cout, a.M, a.accumulate, 16, 2 * NR * 16, p.gx, p.gy, (int)p.lds);
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
  YS_LAUNCH_LDS((conv_halo_kernel<NR, RED>), dim3(p.gx, p.gy), 256, p.lds, st, a, p.h);
#ifdef YS_P2_TIMELINE
  if (tl_path) {
    static unsigned long long h[64 * 64];
    hipStreamSynchronize(st);
    hipMemcpy(h, tl_buf, sizeof(h), hipMemcpyDeviceToHost);
    FILE* f = fopen(tl_path, "a");
    if (f) {
      fprintf(f, "# halo k33 cin%d cout%d M%d acc%d tile%dx16x%d grid%dx%d lds%d mtiles%d nchunk%d (stamps: entry, then per tile: prologue issued, K loop done, epilogue done; exit)\n", a.Cin, a.Cout, a.M, a.accumulate, 16, 2 * NR * 16, p.gx, p.gy, (int)p.lds, p.h.mtiles, p.h.nchunk);
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
#define HL(R_) if (p.nr == R_) return a.nred > 0 ? conv_halo_launch_t<R_, 1>(st, a, p) : conv_halo_launch_t<R_, 0>(st, a, p);
  HL(5) HL(4)
#undef HL
  return YS_ERR_UNSUPPORTED;
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
