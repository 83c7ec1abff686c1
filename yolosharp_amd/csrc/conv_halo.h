// conv_halo.h -- conv_halo_kernel: the halo-patch form of the 3x3 stride-1 wide bf16 layers.  Planned and routed by conv_gemm.hip (conv_halo_plan); the kernel
// templates are instantiated in conv_halo4.hip / conv_halo5.hip (one translation unit per channel-tile width: 24 kernels with 720-MFMA loop bodies
// compile for minutes, so they build side by side).
#pragma once
#include "conv_epi.h"
#include <atomic>
#include <cstdlib>
// ------------------------------------------------------------------ halo-patch form for the 3x3 stride-1 layers (round 5)
// conv_gemm_kernel gathers the A operand of a 3x3 layer NINE times (once per tap: 36 KB of LDS DMA per K-tile of its 128 x 160 tile against 640 cycles
// of MFMA), every wave both requests and multiplies, and its ring is two K-tiles deep.  Here ONE workgroup of FOUR waves per CU -- one wave per SIMD, up to
// 512 registers each -- owns a 2-D tile of 16 x 16 output pixels x BN = 160 / 128 output channels (2 x 2 waves, each a 128 x 80 / 128 x 64 register tile:
// 160 / 128 accumulators) and stages the INPUT PATCH of the tile -- 18 x 18 pixels, one 64-channel chunk at a time, 128-byte rows -- ONCE per chunk; the
// nine taps of the chunk read their pixel fragments from that one patch at shifted rows.  Only the weights stream per tap (BN rows x 128 B).  Per tap (= 2
// K-steps = 80 MFMAs per wave, 1280 MFMA cycles per SIMD) 5 KB of patch + 20 KB of weights land: ~20 B/clk/CU against the blocked kernel's 56, and a wave
// issues ~7 one-KB requests per 80 MFMAs instead of 36.
//   * one wave per SIMD: nothing on the SIMD hides a wave's stalls, so the wave hides them itself -- two fragment sets (K-step 0 / 1 of a tap), the 13 reads of
//     the next K-step and the tap's 6-7 operand requests sit ONE AT A TIME in the issue slots between MFMAs (kstep hooks), per-lane request offsets are computed
//     once per tile (a request is 2-3 VALU + the DMA instruction), requests are unconditional (an out-of-range offset when there is nothing to fetch), every
//     s_waitcnt count and ring slot is a compile-time constant.  Measured (s_memtime, 320 -> 320 at 80 x 80 x 16): 1500 cycles per tap against the 1360 of 80
//     back-to-back 16x16x32 MFMAs.  How it got there (each step measured, profiles/README.md round 5): two 256-thread workgroups per CU with 32-channel chunks sat
//     on the gather rate of 64-byte row segments (13-21 B/clk/CU, tools/probe/probe_dma_rate.hip; full 128-byte lines reach 40-54); 8 waves per CU in one
//     workgroup spent ~1000 of 2250 cycles per tap with every wave in its request burst or at the barrier; with one wave per SIMD the barrier is 4 waves wide and
//     the requests ride under the MFMAs.
//   * the taps of ALL the tiles of a workgroup form one stream (the weights of a tap are the same bytes for every tile; 9 taps = 3 ring turns): the last chunk
//     of a tile requests chunk 0 of the next tile and its first taps, patch buffers alternate across the tile boundary -- only the first tile has a prologue
//     (per tile it was ~20 thousand cycles: every CU asking for 108 KB at once), and the next tile's operands land under the epilogue.
//   * LDS: two patch buffers (chunk c is multiplied while chunk c + 1 lands, 2 x 45 KB) + a three-stage weight ring (stage = tap, slot = tap % 3, 3 x 20 KB);
//     everything by LDS DMA through buffer descriptors (zero padding = out-of-range offsets), one LDS-only barrier per tap, between its two K-steps.
//   * 128-byte rows on purpose (see the probe above); a last chunk with at most 32 real channels (Cin = 160) runs a compile-time copy of the tap body without
//     the MFMAs of K-step 1 (HALF).
//   * bank conflicts: a tap shift moves a fragment's 16 pixels to an ARBITRARY patch offset, so the blocked kernel's swizzle (which relies on
//     16-row-aligned fragments) does not carry over.  Layout (tools/dev/r05/halo_bank_model.py checks every offset against the ds_read_b128 lane
//     groups of MI355X_MICROARCH.md; measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.000): patch pixel p (row pitch 20 pixels) stores K-unit u at slot
//     u ^ ((p >> 1) & 7); lane quarter q multiplies K-unit (ks << 2) | ((q & 1) << 1) | (q >> 1) of the row in K-step ks (both operands agree, so any
//     assignment is a valid dot product; K-step 0 = channels 0-31 of the chunk); MFMA columns 0-3, 12-15 hold tile columns 0 1 4 5 8 9 12 13 and columns
//     4-11 hold 2 3 6 7 10 11 14 15.  With these three the 16 lanes the LDS serves together always hit 16 different 16-byte bank columns.  Weight
//     rows (fragments 16-row aligned): slot u ^ ((n >> 1) & 7) ^ (2 * (((n >> 2) ^ (n >> 3)) & 1)).
//   * fragment addresses: with a 20-pixel pitch (p >> 1) & 7 of a shifted pixel depends only on (row & 3, column), so a lane keeps 12 offsets
//     (3 column shifts x 4 row classes); a read is offset + immediate (K-step 1: offset ^ 64).
//   * epilogue: a compile-time class of the launch (RED).  Overwriting launches store straight from the accumulator registers (p2_epilogue_direct; the training
//     forward of a BatchNorm unit as its FMODE 0: with the options tested at run time the epilogue was ~450 scalar branches and ~1900 reloads of spilled scalar
//     registers, 23 thousand cycles per tile; 6.5 thousand now); accumulating / fused-reduction launches use the LDS-staged wide form in the released patch buffer.
struct HaloArgs {
  int nchunk;                 // 64-channel chunks of the input: ceil(Cin / 64)
  int tiles_x, tiles_y, mtiles;
  YsFastDiv dTpi, dTx;        // tile index -> (image, tile row, tile column)
  unsigned abytes;            // descriptor range of the input view
};
#define HALO_PW 18            // patch columns: 16 + halo
#define HALO_PWP 20           // row pitch of the patch in pixels (see above)
#define HALO_EMR 2            // fragment rows per staged epilogue call (accumulate / reduction launches)
// MR = fragment rows (tile rows) per wave: 8 (tile 16 x 16 pixels, patch 18 x 18) or -- round 5, for maps that 16-row tiles cover badly (40 x 40: 69 %, 288 tile-jobs
// on 256 CUs) -- 4 (tile 16 columns x 8 rows, patch 18 x 10)
__host__ __device__ constexpr int halo_ph(int mr) { return 2 * mr + 2; }                        // patch rows
__host__ __device__ constexpr int halo_npp(int mr) { return halo_ph(mr) * HALO_PWP / 8; }       // 1 KB DMA pieces (8 pixels x 128 B) per patch: 45 / 25
__host__ __device__ constexpr int halo_patch(int mr) { return halo_npp(mr) * 1024; }
__host__ __device__ constexpr int halo_stage_bytes(int nr) { return 2 * nr * 16 * 128; }
__host__ __device__ constexpr int halo_wstg(int nr) { return 16 * HALO_EMR * (nr * 16 + 8) * 2 + 16 * HALO_EMR * 16; }
__host__ __device__ constexpr size_t halo_lds_bytes(int nr, int mr) { return (size_t)2 * halo_patch(mr) + (size_t)3 * halo_stage_bytes(nr); }


// launch description handed from the plan (conv_gemm.hip) to the instantiating translation units
struct HaloLaunch { int gx, gy, nr, mr; size_t lds; HaloArgs h; };
int ys_conv_halo_launch_nr4(hipStream_t st, const ConvArgs& a, const HaloLaunch& p);
int ys_conv_halo_launch_nr5(hipStream_t st, const ConvArgs& a, const HaloLaunch& p);
int ys_conv_halo_launch_nr4m(hipStream_t st, const ConvArgs& a, const HaloLaunch& p);     // MR = 4 (conv_halo4m.hip / conv_halo5m.hip)
int ys_conv_halo_launch_nr5m(hipStream_t st, const ConvArgs& a, const HaloLaunch& p);

#ifdef HALO_INSTANTIATE_NR
template <int MR> struct HaloSched {
  static constexpr int npt(int tap) { return MR == 8 ? (tap <= 4 ? 2 : (tap <= 6 ? 1 : 0)) : (tap <= 6 ? 1 : 0); }   // patch pieces a wave requests at this tap
  static constexpr int jp0(int tap) { return MR == 8 ? (tap <= 4 ? 2 * tap : tap + 5) : tap; }                      // ... the first of them
};
// HALF: Cin mod 64 in 1 .. 32 (the last chunk's K-step 1 is all padding).  RED = epilogue class, a compile-time fact of the launch: 0 = forward with every run-time
// option; 1 (round 6) = the EVAL forward of a BatchNorm unit (folded scale / shift + SiLU, nothing else); 2 = the training forward of a BatchNorm unit (raw output + statistics only); 4 .. 7 = dgrad: 4 + (accumulate) + 2 (fused BN-backward reduction)
template <int NR, int RED = 0, int HALF = 0, int MR = 8>
__global__ void __launch_bounds__(256, 1)
conv_halo_kernel(ConvArgs a, HaloArgs g) {
  typedef bf16_t T;
  constexpr int WM = 2, WN = 2, EMR = HALO_EMR, NWV = 4;
  constexpr int TH = WM * MR, PH = halo_ph(MR);
  constexpr int NPP = halo_npp(MR), NPW = (NPP + NWV - 1) / NWV, NPMIN = NPP / NWV;   // patch pieces per chunk: workgroup, wave (most / least): 45, 12, 11 (MR = 4: 25, 7, 6)
  constexpr int PATCH = halo_patch(MR);
  // request schedule of a chunk's NPW patch pieces over the taps (all of them by tap 6, see the wait below): MR = 8: two at taps 0 - 4, one at taps 5 - 6; MR = 4: one at taps 0 - 6.
  // Requests sit in every second MFMA slot of K-step 1 (MR = 8: 40 / 32 slots) or in every slot (MR = 4: 20 / 16 slots), the fragment reads of the next tap behind them.
  constexpr int RSTRIDE = MR == 8 ? 2 : 1;
  typedef HaloSched<MR> Sched;
  constexpr int BN = WN * NR * 16;
  constexpr int NBP = BN / 8, NBW = NBP / NWV;                 // weight pieces (8 rows x 128 B) per tap: workgroup, wave
  constexpr int STAGE = halo_stage_bytes(NR);
  constexpr int NMF = MR * NR;                                 // MFMAs per K-step and wave
  static_assert((MR == 8 || MR == 4) && TH + 2 == PH && NPW == (MR == 8 ? 12 : 7) && NPMIN == NPW - 1 && NBP % NWV == 0 && MR % EMR == 0 && (PH * HALO_PWP) % 8 == 0 &&
                NMF >= RSTRIDE * ((MR == 8 ? 2 : 1) + NBW) + NR + MR && PH * HALO_PWP <= 400, "halo pipeline");
  static_assert(2 * WM * BN * 4 <= 3 * STAGE && 16 * 256 * 4 <= 3 * STAGE && NWV * halo_wstg(NR) <= PATCH, "statistics scratch inside the ring, epilogue staging inside one patch buffer");
#ifdef YS_P2_TIMELINE
  int tl_n = 0;
  unsigned long long* tl_p = (a.tl && (blockIdx.x % 37) == 0 && blockIdx.y == 0 && threadIdx.x == 0) ? a.tl + (blockIdx.x / 37) * 64 : nullptr;
  // (s_memtime through volatile asm with a memory clobber + its own wait: hipcc otherwise moves the counter read across the MFMA stream)
#define HTL_STAMP() do { if (tl_p && tl_n < 63) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory"); tl_p[1 + tl_n++] = t_; } } while (0)
#else
#define HTL_STAMP() ((void)0)
#endif
  HTL_STAMP();
  YS_DYN_LDS(lds);
  char* lb = (char*)lds;
  char* sRing = lb + 2 * PATCH;

  const int tid = threadIdx.x, lane = tid & 63;
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

  // tile order: as conv_gemm_kernel -- workgroup i runs on XCD i % 8 and walks that XCD's contiguous share of the tiles
  const bool xcd_order = (gridDim.x & 7) == 0;
  const int t_per_xcd = (g.mtiles + 7) >> 3;
  const int t_step = xcd_order ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int t_first = xcd_order ? (int)(blockIdx.x & 7) * t_per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_end = xcd_order ? (((int)(blockIdx.x & 7) + 1) * t_per_xcd < g.mtiles ? ((int)(blockIdx.x & 7) + 1) * t_per_xcd : g.mtiles) : g.mtiles;

  // Epilogue form.  Overwriting launches go straight from the accumulator registers (p2_epilogue_direct); launches that ACCUMULATE into the output or carry the fused
  // BN-backward reduction read 80-160 KB of old gradient / producer output per tile, and the direct form's 8-byte-per-lane loads are quarter-line gathers
  // (29-48 thousand cycles per tile, measured) -- those use the LDS-staged wide form (p2_epilogue: 16-byte vectors, all operands of a call requested up front),
  // two fragment rows per call, staged in the patch buffer the last chunk has just released.
  constexpr bool STAGED = RED >= 5;
  constexpr int NST = STAGED ? 8 : 4 * NR;
  float st1[NST], st2[NST];                   // BatchNorm statistics / fused BN-backward sums (per-lane layout of the epilogue form)
#pragma unroll
  for (int e = 0; e < NST; e++) { st1[e] = 0.f; st2[e] = 0.f; }

  // tile -> (image, first row, first column, byte offset of patch pixel (0, 0) from the descriptor base: may be negative at the border)
  struct TileAt { int b, y0, x0, tb; };
  auto tile_at = [&](const int tile) {
    TileAt t;
    t.b = (int)ys_fastdiv((unsigned)tile, g.dTpi);
    const int trem = tile - t.b * (g.tiles_x * g.tiles_y);
    const int ty = (int)ys_fastdiv((unsigned)trem, g.dTx), tx = trem - ty * g.tiles_x;
    t.y0 = ty * TH; t.x0 = tx * 16;
    t.tb = (int)((((long)t.b * a.in_bstride + (long)(t.y0 - 1) * a.Win + (t.x0 - 1)) * ldu) << 4);
    return t;
  };
  // ONE wave per SIMD (up to 512 registers): the wave hides its own LDS reads, operand requests and waits in the issue slots between its MFMAs.  Two fragment
  // sets: set A holds K-step 0 of a tap, set B K-step 1.  The taps of all the tiles of the workgroup form ONE stream (the weights of a tap are the same bytes
  // for every tile, 9 taps = 3 ring turns): the last chunk of a tile requests chunk 0 of the NEXT tile and its first taps, so only the first tile has a prologue
  // (per tile it was ~20 thousand cycles: every CU asking for 108 KB at once) and the next tile's operands land under the epilogue.  Iteration of tap s:
  //   K-step 0: 40 (32) MFMAs on A; between them the 13 (12) reads of (s, K-step 1) -> B
  //   s_waitcnt vmcnt(N) + barrier of tap s + 1 (its requests were issued two iterations ago); frees ring slot s % 3 and, at the last tap of a chunk, the patch buffer
  //   K-step 1: MFMAs on B; between them this tap's requests (patch pieces of the next chunk, weights of tap s + 3 -> slot s % 3), then the reads of (s + 1, K-step 0) -> A
  int par = 0;                                // patch buffer of the current chunk
  bool primed = false;                        // the current tile's chunk 0 and taps 0 .. 2 were requested (and its tap-0 barrier passed) by the previous tile
  for (int tile = t_first; tile < t_end; tile += t_step) {
    const TileAt tc_ = tile_at(tile);
    const bool has_next = tile + t_step < t_end;
    const TileAt tn_ = tile_at(has_next ? tile + t_step : tile);
    // Per-lane request / fragment offsets are re-derived per tile from a laundered lane id: kept live across the epilogue (the register peak of the kernel)
    // they were spilled to scratch and reloaded inside the K loop, and every scratch reload carries an s_waitcnt vmcnt(0) that drains the operand pipeline.
    int ln = lane;
#ifndef YS_EMU_BUILD
    asm volatile("" : "+v"(ln));
#endif
    // weight requests: piece bp = wave + 4j covers rows 8bp .. 8bp + 7 of the stage, lane l -> row 8bp + (l >> 3), slot l & 7
    unsigned boff[NBW];
#pragma unroll
    for (int j = 0; j < NBW; j++) {
      const int row = (wave + NWV * j) * 8 + (ln >> 3), n = n0 + row;
      const int u = (ln & 7) ^ ((row >> 1) & 7) ^ ((((row >> 2) ^ (row >> 3)) & 1) << 1);
      boff[j] = n < a.Cout ? (unsigned)((long)n * Kbytes) + (unsigned)u * 16u : YS_BUF_OOB;
    }
    // fragment read offsets (K-step 0; K-step 1 = the same ^ 64).  Pixels: MFMA column li holds tile column xm; a shifted pixel (row r, column xm + kx) sits at
    // padded index r * 20 + xm + kx, whose swizzle term (2 (r & 3) + ((xm + kx) >> 1)) & 7 depends on the row only through r & 3 (the wave's first row 8 wm
    // is a multiple of 4)
    const int li = ln & 15, q = ln >> 4;
    const int xt = (li >= 4 && li < 12) ? li - 4 : (li & 3);
    const int xm = ((xt & 1) | ((xt >> 1) << 2)) + ((li >= 4 && li < 12) ? 2 : (li >= 12 ? 8 : 0));   // 0 1 4 5 | 2 3 6 7 10 11 14 15 | 8 9 12 13
    const int uq0 = ((q & 1) << 1) | (q >> 1);
    int offA[3][4];
#pragma unroll
    for (int kx = 0; kx < 3; kx++)
#pragma unroll
      for (int rc = 0; rc < 4; rc++) {
        const int xx = xm + kx;
        offA[kx][rc] = (wm * MR * HALO_PWP + xx) * 128 + ((uq0 ^ ((2 * rc + (xx >> 1)) & 7)) << 4);
      }
    const int offB = (wn * NR * 16 + li) * 128 + ((uq0 ^ ((li >> 1) & 7) ^ ((((li >> 2) ^ (li >> 3)) & 1) << 1)) << 4);

    // Patch requests.  Piece pp = wave + 4j (j < 12) of a chunk covers padded patch pixels 8pp .. 8pp + 7, lane l -> pixel 8pp + (l >> 3), slot l & 7.  The per-lane part
    // of the 12 source offsets -- pixel position inside the patch, K-unit, and whether the pixel lies inside the image for tile t -- is computed ONCE per tile (prq_set)
    // into prq[j] (+ a bit mask for the last chunk when Cin is not a multiple of 64: units past Cin out of range); a request inside the MFMA stream is
    // then two or three VALU instructions + the DMA instruction.  (Computed at the request, ~20 VALU instructions sat between two MFMAs, seven times per tap: the matrix pipe idled ~600 cycles per tap.)
    unsigned prq[NPW], prl_bad = 0u;            // prl_bad bit j: piece j's K-unit of this lane lies past Cin in the last chunk
    const bool part_last = (a.Cin & 63) != 0;
    auto prq_set = [&](const TileAt& t) {
#pragma unroll
      for (int j = 0; j < NPW; j++) {
        const int pl = (wave + NWV * j) * 8 + (ln >> 3);
        const int py = (pl * 3277) >> 16, px = pl - py * HALO_PWP;    // pl / 20 for pl < 400
        const int u = (ln & 7) ^ ((pl >> 1) & 7);
        const bool ok = (bool)((int)(wave + NWV * j < NPP) & (int)(px < HALO_PW) & (int)(py < PH) & (int)((unsigned)(t.y0 - 1 + py) < (unsigned)a.Hin) & (int)((unsigned)(t.x0 - 1 + px) < (unsigned)a.Win));
        const unsigned v = (unsigned)(t.tb + (((py * a.Win + px) * ldu + u) << 4));
        prq[j] = ok ? v : YS_BUF_OOB;
        if (j == 0) prl_bad = 0u;
        prl_bad |= (((g.nchunk - 1) * 8 + u) * 8 < a.Cin ? 0u : 1u) << j;
      }
    };
    // piece j of chunk c into buffer buf (offsets of the tile prq_set was last called for); an out-of-range lane stays out of range: YS_BUF_OOB + c * 128 < 2^32
    auto issue_p = [&](const int buf, const int c, auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j < NPMIN) {              // (the last piece exists for one wave only: a wave-uniform branch there)
        const unsigned bad = (part_last && c == g.nchunk - 1) ? (prl_bad >> j) & 1u : 0u;
        ys_bufld_lds16_nom0(rsA, prq[j] | (bad << 31), (unsigned)c * 128u, lb + buf * PATCH + (wave + NWV * j) * 1024);   // (valid offsets are < 2^31; YS_BUF_OOB = bit 31)
      } else if (wave + NWV * j < NPP) {
        const unsigned bad = (part_last && c == g.nchunk - 1) ? (prl_bad >> j) & 1u : 0u;
        ys_bufld_lds16_nom0(rsA, prq[j] | (bad << 31), (unsigned)c * 128u, lb + buf * PATCH + (wave + NWV * j) * 1024);
      }
    };
    // In the K loop the requests are UNCONDITIONAL: a request that has nothing to fetch (no next chunk, no tap + 3) carries the out-of-range offset in its scalar part
    // and lands zeros in a buffer nobody reads -- a scalar branch around each of the seven requests of a tap cost more than the request, and the wait counts become constants.
    auto issue_w1 = [&](const int slot, const unsigned so, const int j) {   // weight piece j of this wave for the tap at scalar offset so (the same bytes for every tile)
      ys_bufld_lds16_nom0(rsB, boff[j], so, sRing + slot * STAGE + (wave + NWV * j) * 1024);
    };

    // read r (0 .. NR + MR - 1) of the fragments of (patch buffer offset pbo, tap, K-step ks) into (fw, fx): weights first, then pixels
    uint4 fwA[NR], fxA[MR], fwB[NR], fxB[MR];
    auto frag_read = [&](auto rc_, auto tapc, auto ksc, uint4 (&fw)[NR], uint4 (&fx)[MR], const int pbo) {
      constexpr int r = decltype(rc_)::value, tap = decltype(tapc)::value, ks = decltype(ksc)::value;
      constexpr int ky = tap / 3, kx = tap - ky * 3;
      if constexpr (r < NR) fw[r] = *(const uint4*)(sRing + (tap % 3) * STAGE + (ks ? (offB ^ 64) : offB) + r * 2048);
      else {
        constexpr int mf = r - NR;
        const int o = offA[kx][(mf + ky) & 3];
        fx[mf] = *(const uint4*)(lb + pbo + (ks ? (o ^ 64) : o) + (mf + ky) * (HALO_PWP * 128));
      }
    };
    prq_set(tc_);                              // (per tile, like the other per-lane offsets: the previous tile's copy for this tile died with its registers)
    if (!primed) {
      // prologue (first tile of the workgroup): patch of chunk 0, weights of taps 0, 1, 2 (the whole ring); tap 0 landed
      ys_static_for<0, NPW>([&](auto jc) { issue_p(par, 0, jc); });
#pragma unroll
      for (int t = 0; t < 3; t++)
#pragma unroll
        for (int j = 0; j < NBW; j++) issue_w1(t, (unsigned)(t * a.Cin) * 2u, j);
      ys_wait_vm<2 * NBW>();
      ys_barrier_lds();
    }
    // (tap 0, K-step 0) -> A.  (Read by the previous tile's last K-step the 52 fragment registers would be live across the epilogue: one exposed LDS round trip per tile instead.)
    ys_static_for<0, NR + MR>([&](auto rc_) { frag_read(rc_, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fwA, fxA, par * PATCH); });
    HTL_STAMP();
    f32x4 acc[MR][NR];
#pragma unroll
    for (int mf = 0; mf < MR; mf++)
#pragma unroll
      for (int nf = 0; nf < NR; nf++) acc[mf][nf] = f32x4_zero();

    // MFMA i of a K-step = (nf, mf) = (i / MR, i % MR); hook(i) runs after it
    auto kstep = [&](const uint4 (&fw)[NR], const uint4 (&fx)[MR], auto hook) {
      ys_static_for<0, NMF>([&](auto ic) {
        constexpr int i = decltype(ic)::value, nf = i / MR, mf = i % MR;
        acc[mf][nf] = ys_mma<T>(fw[nf], fx[mf], acc[mf][nf]);
        YS_SCHED_FENCE();
        hook(ic);
        YS_SCHED_FENCE();
      });
    };
    // the hooks of a K-step without its MFMAs: K-step 1 of the last chunk when at most 32 of its 64 channels are real (Cin = 160: 2.5 chunks -- the zero half
    // would be a sixth of the layer's MFMAs)
    auto kstep_hooks_only = [&](auto hook) { ys_static_for<0, NMF>([&](auto ic) { hook(ic); }); };
    // one chunk = nine taps.  SKIP1: no MFMAs for K-step 1 (the last chunk of a layer whose Cin leaves at most 32 of its 64 channels real -- a compile-time
    // copy of the body for that one chunk: a run-time branch inside every tap doubled the loop's code and its register pressure)
    auto chunk_body = [&](auto skipc, const int c) {
      constexpr bool SKIP1 = decltype(skipc)::value;
      const bool in_tile = c + 1 < g.nchunk;
      const bool more = in_tile || has_next;  // the stream has a next chunk: chunk c + 1 of this tile, or chunk 0 of the next
      const int cq = in_tile ? c + 1 : 0;
      if (!in_tile && has_next) prq_set(tn_); // the next patch requests are the next tile's chunk 0
      const int pbo = par * PATCH;            // patch buffer of this chunk
      ys_static_for<0, 9>([&](auto tc) {
        constexpr int tap = decltype(tc)::value;
        // ---- K-step 0 on A; reads (tap, K-step 1) -> B
        kstep(fwA, fxA, [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if constexpr (i < NR + MR && !SKIP1) frag_read(ic, tc, std::integral_constant<int, 1>{}, fwB, fxB, pbo);
        });
        // ---- tap + 1 has landed (this wave's pieces: everything older than the requests of the previous iteration), everybody's have
        {
          // patch pieces every wave issued in the previous iteration (unconditional requests: constants).  Schedule of a chunk's 12 pieces: two at taps 0 - 4, one at
          // taps 5 - 6, NONE at taps 7 - 8: this wait covers what was issued up to TWO iterations ago, and the next chunk's patch is first read right behind the
          // barrier of tap 8 -- so its last piece must be issued by tap 6.  (The first version issued one piece per tap up to tap 8: the last two were read
          // without a covering wait -- correct in the interpreter, where a request lands at once, and almost always on the device; it showed as a loss that
          // differed in the fourth digit between two runs of bench.py.)
          constexpr int np_prev = tap == 0 ? 0 : Sched::npt(tap - 1);
          // (piece 11, issued at tap 6, exists for wave 0 only -- 45 pieces over 4 waves: the count must be what THIS wave issued, one too many leaves the oldest
          // weight piece of this tap uncovered)
          if constexpr (tap == 7) { if (wave + NWV * (NPW - 1) < NPP) ys_wait_vm<np_prev + NBW>(); else ys_wait_vm<np_prev - 1 + NBW>(); }
          else ys_wait_vm<np_prev + NBW>();
          ys_barrier_lds();
        }
        // ---- K-step 1 on B; this tap's requests, then reads (tap + 1, K-step 0) -> A
        constexpr int tap3 = (tap + 3) % 9;
        const int c3 = c + (tap + 3) / 9;     // chunk of tap + 3: c, c + 1 (<= nchunk: the next tile's chunk 0)
        const unsigned so3 = (c3 < g.nchunk || has_next) ? (unsigned)(tap3 * a.Cin + (c3 < g.nchunk ? c3 : 0) * 64) * 2u : YS_BUF_OOB;
        const unsigned cqs = more ? (unsigned)cq : (YS_BUF_OOB >> 7);     // chunk index of the patch requests; out of range when the stream ends
        auto hook1 = [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          // requests at MFMAs 0, 2, 4, ...: [patch piece(s) of the next chunk: two at taps 0 - 4, one at taps 5 - 6], weights of tap + 3
          constexpr int NPT = Sched::npt(tap), JP0 = Sched::jp0(tap);   // patch pieces of this tap: JP0 .. JP0 + NPT - 1 (NPW per chunk)
          if constexpr ((i % RSTRIDE) == 0 && i / RSTRIDE < NPT) issue_p(par ^ 1, (int)cqs, std::integral_constant<int, JP0 + i / RSTRIDE>{});
          else if constexpr ((i % RSTRIDE) == 0 && i / RSTRIDE < NPT + NBW) issue_w1(tap % 3, so3, i / RSTRIDE - NPT);
          else if constexpr (i >= RSTRIDE * (NPT + NBW) && i < RSTRIDE * (NPT + NBW) + NR + MR) {
            constexpr int r = i - RSTRIDE * (NPT + NBW);
            if constexpr (tap < 8) frag_read(std::integral_constant<int, r>{}, std::integral_constant<int, tap + 1>{}, std::integral_constant<int, 0>{}, fwA, fxA, pbo);
            else { if (in_tile) frag_read(std::integral_constant<int, r>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fwA, fxA, PATCH - pbo); }
          }
        };
        if constexpr (!SKIP1) kstep(fwB, fxB, hook1); else kstep_hooks_only(hook1);
      });
      par ^= 1;
      HTL_STAMP();                            // (triage builds: one stamp per chunk)
    };
    {
      const int nfull = HALF ? g.nchunk - 1 : g.nchunk;
#pragma unroll 1
      for (int c = 0; c < nfull; c++) chunk_body(std::integral_constant<bool, false>{}, c);
      if constexpr (HALF) chunk_body(std::integral_constant<bool, true>{}, g.nchunk - 1);
    }
    primed = has_next;
    HTL_STAMP();
    // epilogue straight from the accumulator registers (conv_epi.h p2_epilogue_direct: 16-byte stores after a 16-lane row swap, no LDS -- the ring and one
    // patch buffer already hold the next tile's operands, and a lone wave per SIMD cannot hide the staged form's dependent LDS round trips)
    if constexpr (!STAGED) {
      int orow[MR];
      bool pv[MR];
#pragma unroll
      for (int mf = 0; mf < MR; mf++) {
        const int oy = tc_.y0 + wm * MR + mf, ox = tc_.x0 + xm;
        pv[mf] = (bool)((int)(oy < a.Hout) & (int)(ox < a.Wout));
        orow[mf] = pv[mf] ? tc_.b * (int)a.out_bstride + oy * a.Wout + ox : 0;
      }
      p2_epilogue_direct<MR, NR, RED >= 4 ? 1 : 0, YsNoStamp, RED == 2 ? 0 : (RED == 1 ? 1 : (RED >= 4 ? RED : 2))>(a, acc, orow, pv, n0 + wn * NR * 16, *reinterpret_cast<float (*)[4 * NR]>(&st1), *reinterpret_cast<float (*)[4 * NR]>(&st2));
    } else {
      ys_barrier_lds();                       // every wave has read the last chunk's patch: its buffer (par was flipped past it) is the staging area
      char* stg = lb + (par ^ 1) * PATCH + wave * halo_wstg(NR);
#pragma unroll
      for (int h = 0; h < MR / EMR; h++) {
        int orow[EMR];
        bool pv[EMR];
#pragma unroll
        for (int e = 0; e < EMR; e++) {
          const int oy = tc_.y0 + wm * MR + h * EMR + e, ox = tc_.x0 + xm;
          pv[e] = (bool)((int)(oy < a.Hout) & (int)(ox < a.Wout));
          orow[e] = pv[e] ? tc_.b * (int)a.out_bstride + oy * a.Wout + ox : 0;
        }
        f32x4 sub[EMR][NR];                   // (register moves the allocator coalesces; no address of acc is taken)
#pragma unroll
        for (int e = 0; e < EMR; e++)
#pragma unroll
          for (int nf = 0; nf < NR; nf++) sub[e][nf] = acc[h * EMR + e][nf];
        p2_epilogue<EMR, NR, 1, 8>(a, sub, orow, pv, n0 + wn * NR * 16, stg, *reinterpret_cast<float (*)[8]>(&st1), *reinterpret_cast<float (*)[8]>(&st2));
      }
    }
    HTL_STAMP();
  }
  YS_WAIT_VM0();                              // the stream's last (empty) requests still land zeros in the ring, which is the flush's scratch (its first barrier follows)
  if constexpr (STAGED) { if (RED & 2) conv_stats_flush_grid<NR, WM, WN>(a, n0, *reinterpret_cast<float (*)[8]>(&st1), *reinterpret_cast<float (*)[8]>(&st2), (float*)sRing, (long)blockIdx.x); }
  else if (RED >= 4 ? (RED & 2) != 0 : a.stats != nullptr) p2_stats_flush_direct<NR, WM, WN>(a, n0, *reinterpret_cast<float (*)[4 * NR]>(&st1), *reinterpret_cast<float (*)[4 * NR]>(&st2), (float*)sRing, (long)blockIdx.x);
  HTL_STAMP();
#ifdef YS_P2_TIMELINE
  if (tl_p) tl_p[0] = (unsigned long long)tl_n;
#endif
}


template <int NR, int RED, int HALF, int MR = 8>
static int conv_halo_launch_t(hipStream_t st, ConvArgs a, const HaloLaunch& p) {
  a.red_koff = (int)offsetof(ConvArgs, red);       // ConvArgs is the kernel's first argument (conv_epi.h ys_red_table)
  static std::atomic<unsigned> attr_done{0};      // per device: the attribute belongs to the device's code object
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  if (!(attr_done.load(std::memory_order_relaxed) & (1u << (dev_id & 31)))) {
    hipFuncSetAttribute((const void*)conv_halo_kernel<NR, RED, HALF, MR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.fetch_or(1u << (dev_id & 31), std::memory_order_relaxed);
  }
  char lab[192] = "";
  if (ys_kprof_enabled()) snprintf(lab, sizeof(lab), "halo k33 s1 div1 cin%d cout%d M%d acc%d tile%dx16x%d grid%dx%d lds%d", a.Cin, a.Cout, a.M, a.accumulate, 2 * MR, 2 * NR * 16, p.gx, p.gy, (int)p.lds);
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
  YS_LAUNCH_LDS((conv_halo_kernel<NR, RED, HALF, MR>), dim3(p.gx, p.gy), 256, p.lds, st, a, p.h);
#ifdef YS_P2_TIMELINE
  if (tl_path) {
    static unsigned long long h[64 * 64];
    hipStreamSynchronize(st);
    hipMemcpy(h, tl_buf, sizeof(h), hipMemcpyDeviceToHost);
    FILE* f = fopen(tl_path, "a");
    if (f) {
      fprintf(f, "# halo k33 cin%d cout%d M%d acc%d tile%dx16x%d grid%dx%d lds%d mtiles%d nchunk%d (stamps: entry, then per tile: prologue issued, K loop done, epilogue done; exit)\n", a.Cin, a.Cout, a.M, a.accumulate, 2 * MR, 2 * NR * 16, p.gx, p.gy, (int)p.lds, p.h.mtiles, p.h.nchunk);
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


static int conv_halo_launch_nr(hipStream_t st, const ConvArgs& a, const HaloLaunch& p) {
  const bool half = (a.Cin & 63) != 0 && (a.Cin & 63) <= 32;
  // epilogue class (conv_halo_kernel's RED parameter)
  const bool plain = !a.stats && !a.scale && !a.shift && !a.res && !a.act;
  const bool eval_bn = a.scale && a.shift && !a.stats && !a.res && !a.accumulate && a.nred == 0;     // (round 6: the eval forward ran class 0 -- 195 us against 112 for the training forward of the same 64 -> 144 layer)
  const int ec = (a.nred > 0 || (plain && a.accumulate)) ? 4 + (a.accumulate ? 1 : 0) + (a.nred > 0 ? 2 : 0) : ((a.stats && !a.scale && !a.shift && !a.res && !a.accumulate && !a.act) ? 2 : (plain ? 4 : (eval_bn ? 1 : 0)));
#ifndef HALO_INSTANTIATE_MR
#define HALO_INSTANTIATE_MR 8
#endif
#define HL(R_, M_) if (p.nr == R_ && p.mr == M_) { \
    if (ec == 4) return half ? conv_halo_launch_t<R_, 4, 1, M_>(st, a, p) : conv_halo_launch_t<R_, 4, 0, M_>(st, a, p); \
    if (ec == 5) return half ? conv_halo_launch_t<R_, 5, 1, M_>(st, a, p) : conv_halo_launch_t<R_, 5, 0, M_>(st, a, p); \
    if (ec == 6) return half ? conv_halo_launch_t<R_, 6, 1, M_>(st, a, p) : conv_halo_launch_t<R_, 6, 0, M_>(st, a, p); \
    if (ec == 7) return half ? conv_halo_launch_t<R_, 7, 1, M_>(st, a, p) : conv_halo_launch_t<R_, 7, 0, M_>(st, a, p); \
    if (ec == 2) return half ? conv_halo_launch_t<R_, 2, 1, M_>(st, a, p) : conv_halo_launch_t<R_, 2, 0, M_>(st, a, p); \
    if (ec == 1) return half ? conv_halo_launch_t<R_, 1, 1, M_>(st, a, p) : conv_halo_launch_t<R_, 1, 0, M_>(st, a, p); \
    return half ? conv_halo_launch_t<R_, 0, 1, M_>(st, a, p) : conv_halo_launch_t<R_, 0, 0, M_>(st, a, p); }
  HL(HALO_INSTANTIATE_NR, HALO_INSTANTIATE_MR)
#undef HL
  return YS_ERR_UNSUPPORTED;
}
#endif
