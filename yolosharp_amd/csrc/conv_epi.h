// conv_epi.h -- output stage shared by the LDS-staged bf16 convolution kernels (conv_p2_kernel in conv.hip, conv_gemm_kernel in
// conv_gemm.hip): accumulator fragments -> bf16 NHWC rows with the BN batch statistics, bias / eval-BN / SiLU, residual and
// gradient accumulation on the wide (16-byte) path.
#pragma once
#include "ys_internal.h"
#include "ys_kernels.h"

// Epilogue of the P2 kernel.  Phase 1: the wave rounds its whole 16*MR x BN accumulator tile into a wave-private LDS
// slice (fragment layout -> pixel rows) and records each pixel's output row.  Phase 2: the wave streams the slice back
// out as full 16-byte vectors -- consecutive lanes cover consecutive channels of one pixel row -- applying the residual /
// gradient accumulation there, and takes the BN batch statistics per 8-channel column (fixed per lane).  Compared with
// conv_epilogue the two phases keep few values live, which leaves the registers to the next tile's patch prefetch.
// ablation switches of the epilogue (triage builds -DYS_P2_ABLATE, YS_DBG bits): 256 = no global stores, 512 = no statistics /
// fused-reduction arithmetic, 1024 = phase 1 only (no store loop at all)
#ifdef YS_P2_ABLATE
#define EPI_DBG(bit) ((a.dbg & (bit)) != 0)
#else
#define EPI_DBG(bit) false
#endif

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N) -- fragment indices must be constants for the register allocator
#include <type_traits>
template <int I, int N, class F> __device__ inline void ys_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); ys_static_for<I + 1, N>(f); }
}

// The segment table of a backward launch, re-read from the kernel-argument segment where it is used.  Read through `a.red[k]` the
// compiler loads the whole table (36 scalar registers) at kernel entry and keeps it live across the tile loop -- the kernels sit at
// the SGPR limit, the overflow spills into VGPR lanes and from there into scratch (128-472 bytes per lane in the RED variants).  The
// laundered pointer makes every use a fresh scalar load inside the epilogue.
#ifdef YS_EMU_BUILD
typedef const BnRedSeg* ys_redp_t;
__device__ inline ys_redp_t ys_red_table(const ConvArgs& a) { return a.red; }
#else
typedef const BnRedSeg __attribute__((address_space(4)))* ys_redp_t;
__device__ inline ys_redp_t ys_red_table(const ConvArgs& a) {
  // a.red_koff = byte offset of THIS ConvArgs' segment table inside the launch's kernel-argument segment, set by the host launcher:
  // offsetof(ConvArgs, red) where ConvArgs is the first kernel argument (conv_p2_kernel, conv_gemm_kernel), the element's offset in the
  // problem array of a grouped launch (conv_p2_group_kernel)
  const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  ys_redp_t p = (ys_redp_t)(ka + a.red_koff);
  asm volatile("" : "+s"(p));
  return p;
}
#endif

// destination of one fused BN-backward partial sum (output-view channel c of the launch, `which` = 0: sum du / 1: sum du * y): a row entry of the covering
// producer's partial buffer
__device__ inline void ys_bnred_put(const ConvArgs& a, int c, int which, long row, float t) {
  const ys_redp_t rt = ys_red_table(a);
#pragma unroll
  for (int k = 0; k < YS_BNRED_MAXSEG; k++)
    if (k < a.nred && c >= rt[k].c0 && c < rt[k].c1) {
      const int pc = c - rt[k].c0 + rt[k].yc0;
      rt[k].part[(((long)a.red_row0 + row) * 2 + which) * rt[k].C + pc] = t;
    }
}

// RED = 1: the dgrad form -- gradient accumulation and the fused BN-backward reduction only (no bias / eval-BN / SiLU / residual:
// a dgrad launch never carries them), compiled as its own kernel variants so that the forward kernels' register allocation is
// untouched by the reduction's live values (y vectors, coefficients).
// Store instructions one wave issues per p2_epilogue call (NITER below): the static count behind the tile loop's counted waits --
// conv_p2_body waits for "all but the last p2_epi_stores() vector-memory operations" at the top of a tile, i.e. for the prefetched
// patch but NOT for the previous tile's output stores (gfx9-family parts count loads and stores on one in-order vmcnt).
__host__ __device__ constexpr int p2_epi_stores(int mr, int nr) { return (16 * mr + 64 / (nr * 2) - 1) / (64 / (nr * 2)); }
#ifndef YS_EPI_ACC_PREFETCH
#define YS_EPI_ACC_PREFETCH 1   // 1: the dgrad (RED) variants; 2: every variant that may accumulate (measured equal: forward variants do not accumulate in a training step)
#endif
#ifndef YS_EPI_SCALAR_STATS
#define YS_EPI_SCALAR_STATS 0
#endif
#ifndef YS_EPI_LDS_NOALIAS
#define YS_EPI_LDS_NOALIAS 0
#endif
#ifndef YS_EPI_FULL_UNROLL
#define YS_EPI_FULL_UNROLL 1   // forward store loop fully unrolled: hipcc then SEES the NITER stores in a row, which is what lets conv_p2_body wait with vmcnt(NITER)
                               // (behind a rolled loop it credits one trip and re-drains).  REQUIRES -fno-slp-vectorize (yolosharp_amd/build.py): with SLP-packed
                               // statistics the unrolled form is the reproducer of the run-to-run nondeterminism (profiles/README.md, round 4); 0 = `unroll 2`
#endif
template <int M> struct EpiMode { static constexpr int value = M; };
struct YsNoStamp { __device__ inline void operator()() const {} };   // timeline hook of triage builds (-DYS_P2_TIMELINE): nothing in the product
// after_stage: called once the accumulators have been rounded into the staging rows (they are dead from there on) -- conv_p2_body's
// streamed-weight variants request the next tile's patch there, into the registers the accumulators just freed
template <int MR, int NR, int RED = 0, int BMAX = 4 /* most store-loop iterations whose accumulate operands are requested ahead */, class SF = YsNoStamp, class AF = YsNoStamp>
__device__ inline void p2_epilogue(const ConvArgs& a, f32x4 (&acc)[MR][NR], const int (&orow)[MR], const bool (&pv)[MR],
                                   int n0, char* stg, float (&s1)[8], float (&s2)[8], SF stamp = SF(), AF after_stage = AF()) {
  typedef bf16_t T;
  constexpr int BN = NR * 16;
  constexpr int PITCH = (BN + 8) * 2;         // bytes per staged pixel row
  constexpr int NPX = 16 * MR;
  constexpr int VPP = BN / 8;                  // 16-byte vectors per pixel
  constexpr int PPI = 64 / VPP;                // pixels per wave iteration
  constexpr int NITER = (NPX + PPI - 1) / PPI;
  static_assert(NITER == p2_epi_stores(MR, NR), "p2_epi_stores");
  const int lane = threadIdx.x & 63, li = lane & 15, q = lane >> 4;
  unsigned* rowtab = (unsigned*)(stg + NPX * PITCH);   // [NPX] byte offset of the pixel row (YS_BUF_OOB = outside), [NPX] row index; views < 2^31 bytes (launch plans)
  const int cv = lane % VPP, pl = lane / VPP;
  const bool active = lane < PPI * VPP;
  const int c = n0 + cv * 8;
  if (q == 0) {
#pragma unroll
    for (int mf = 0; mf < MR; mf++) {
      rowtab[mf * 16 + li] = pv[mf] ? ((unsigned)orow[mf] * (unsigned)a.out_ldc + (unsigned)a.out_coff) * 2u : YS_BUF_OOB;
      rowtab[NPX + mf * 16 + li] = (unsigned)orow[mf];
    }
  }
  // ---- fused BN-backward reduction (BnRedSeg): this lane's 8-channel column belongs to at most one producer.  Its y vectors for
  // all NITER iterations are requested now -- right after the row table, before the accumulators are staged -- so that their
  // latency overlaps phase 1 (a load issued inside the store loop was one dependent HBM round trip per iteration).
  const bool red_on = RED != 0 && a.nred > 0;
  const char* ry = nullptr; const float* rscp = nullptr; const float* rshp = nullptr; int rC = 0, rcol = 0; bool ract = false;
  uint4 yv[RED ? NITER : 1];
  float rsc[8], rsh[8];
  if (RED && red_on) {
#pragma unroll
    for (int k = 0; k < YS_BNRED_MAXSEG; k++)
      if (k < a.nred && active && c >= a.red[k].c0 && c < a.red[k].c1) {
        ry = (const char*)a.red[k].y; rscp = a.red[k].scale; rshp = a.red[k].shift; rC = a.red[k].C;
        rcol = c - a.red[k].c0 + a.red[k].yc0; ract = a.red[k].act != 0;
      }
    ys_wave_sync();                           // the row table is visible
#pragma unroll
    for (int it = 0; it < NITER; it++) {
      const int px = it * PPI + pl;
      yv[it] = ys_zero16();
      if (ry && px < NPX && rowtab[px] != YS_BUF_OOB) yv[it] = ys_ld16(ry + ((long)rowtab[NPX + px] * rC + rcol) * 2L);
    }
#pragma unroll
    for (int e = 0; e < 8; e++) { rsc[e] = 0.f; rsh[e] = 0.f; }
    if (ry) { ys_ldcoef<8>(rscp + rcol, rsc); ys_ldcoef<8>(rshp + rcol, rsh); }
  }
  // ---- gradient accumulation (dgrad into a view that already holds another consumer's contribution): the old values of all NITER
  // iterations are requested here too (round 4).  Loaded inside the store loop each iteration was load -> s_waitcnt vmcnt(0) -> add ->
  // store, i.e. NITER dependent memory round trips per tile, each also draining the previous iteration's store: accumulate launches ran
  // 1.15-1.9x their overwrite twins (per-launch records: 3x3 32 -> 32 at 160 x 160: 40.6 us against 26.6).  Every lane re-reads exactly the
  // 16 bytes it will overwrite, so the order of load and store per address is program order.
  constexpr bool ACC_PRE = (RED != 0 || YS_EPI_ACC_PREFETCH == 2) && YS_EPI_ACC_PREFETCH != 0 && NITER <= BMAX;   // BMAX: what the caller's register budget affords (conv_p2: 4 vectors; conv_gemm: all)
  uint4 ov[ACC_PRE ? NITER : 1];
  if (ACC_PRE && a.accumulate) {
    const ys_rsrcv_t rsO = ys_make_rsrcv(a.y, 0x7ffffff0u);
    if (!red_on) ys_wave_sync();              // (the reduction path above has already made the row table visible)
#pragma unroll
    for (int it = 0; it < NITER; it++) {
      const int px = it * PPI + pl;
      const unsigned ro = (active && px < NPX && c < a.Cout) ? rowtab[px] : YS_BUF_OOB;
      ov[it] = ys_bufld16(rsO, ro != YS_BUF_OOB ? ro + (unsigned)c * 2u : YS_BUF_OOB);
    }
  }
  stamp();
#pragma unroll
  for (int mf = 0; mf < MR; mf++) {
#pragma unroll
    for (int nf = 0; nf < NR; nf++) {
      const int c = n0 + nf * 16 + 4 * q;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = acc[mf][nf][r];
      if (!RED && !a.scale && a.shift) {     // plain conv bias (heads): added to the fp32 accumulators
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int cc = (c + r) < a.Cout ? (c + r) : 0;
          v[r] += a.shift[cc];
        }
        if (a.act) {
#pragma unroll
          for (int r = 0; r < 4; r++) v[r] = ys_silu(v[r]);
        }
        // padded channels of the output row stay zero.  Only the bias can make them anything else: the weight rows past Cout reach
        // LDS as zeros (out-of-range DMA / masked fetch), so their accumulators ARE zero -- the four selects per fragment (and the
        // SGPR pairs holding their lane masks) are not part of the BatchNorm / gradient launches
#pragma unroll
        for (int r = 0; r < 4; r++) if (c + r >= a.Cout) v[r] = 0.f;
      }
      uint2 pk;
      pk.x = ys_pack_bf16x2(v[0], v[1]);
      pk.y = ys_pack_bf16x2(v[2], v[3]);
      *(uint2*)(stg + (mf * 16 + li) * PITCH + (nf * 16 + 4 * q) * 2) = pk;
    }
  }
  stamp();
  ys_wave_sync_lds();                          // the staged rows (written by other lanes of this wave) are IN LDS before the reads below are issued
  after_stage();
  stamp();
  const bool do_stats_rt = !RED && a.stats != nullptr && !EPI_DBG(512);
  // eval-mode BatchNorm folded into the conv (Convs.cs:48 with running statistics): applied on the wide path to the
  // bf16-rounded conv output -- the same value the training path normalises -- with the lane's 8 coefficients loaded once
  const bool bn_eval_rt = !RED && a.scale != nullptr;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { sc[e] = 1.f; sh[e] = 0.f; }
  if (bn_eval_rt && active && c < a.Cout) {     // coefficient arrays are padded to a multiple of 4 floats; Cout % 8 == 0 for BN convs
    ys_ldcoef<8>(a.scale + c, sc);
    if (a.shift) ys_ldcoef<8>(a.shift + c, sh);
  }
  char* yb = (char*)a.y;
  const char* rb_rt = RED ? nullptr : (const char*)a.res;
  // Stores go through a buffer descriptor of the output buffer: masked lanes (pixel outside the image, channel tile past Cout,
  // idle lanes of the wave) carry the out-of-range offset and the hardware drops them, so the store is issued unconditionally,
  // NITER times per tile -- a static count the compiler can keep in flight (vmcnt(N)) across the next tile's loads instead of the
  // vmcnt(0) that exec-masked stores forced.  Valid offsets are < 2^31 (checked by the launch plans).
  const ys_rsrcv_t rsY = ys_make_rsrcv(yb, 0x7ffffff0u);
  // One iteration at a time: row offset, then the staged vector, then the store.  (Round 3 tried batches of up to four iterations with
  // all LDS reads -- and the accumulate operand -- issued up front: 0.3 % faster on config 2, but the blocked-GEMM kernel's BatchNorm sums
  // then differed from run to run on wide layers (cin400 -> cout160 1x1 at 320x320: one pixel's vectors in ~10^6 read stale, element 1 / 3 / 5
  // of every vector of that pixel; caught by tests/test_configs.py::test_c5_v8x_1280_bs16_fp8_train_steps, reproducer
  // tools/dev/determinism_layer.py).  A wait between the staging writes and the reads removed one instance, a full wait plus idle
  // cycles after the reads lowered the rate of the other but did not remove it -- reverted to this form, which the row-table
  // dependency serialises: every read is consumed before the next is issued.)
  // MODE (compile-time view of the run-time flags, so that the common launches get a branch-free, fully unrolled loop -- the compiler
  // then SEES the NITER stores in a row, which is what lets the tile loop wait with vmcnt(NITER) instead of draining them):
  //   0 training forward (statistics only), 1 eval BatchNorm (+ SiLU), 2 everything (residual, accumulate), RED always 2
  auto store_iter = [&](const int it, auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    const bool do_stats = MODE != 1 && do_stats_rt;
    const bool bn_eval = MODE != 0 && bn_eval_rt;
    const char* const rb = MODE == 2 ? rb_rt : nullptr;
    const bool accum = MODE == 2 && a.accumulate;
    const int px = it * PPI + pl;
    const bool lane_ok = active && px < NPX && c < a.Cout;
    const unsigned rofs = lane_ok ? rowtab[px] : YS_BUF_OOB;
    uint4 val = ys_zero16();
    {
      if (rofs != YS_BUF_OOB) {
#if YS_EPI_LDS_NOALIAS && !defined(YS_EMU_BUILD)
        {   // triage (round-4 determinism bisection): the staged vector through an LDS read whose destination registers cannot be the address register
          const unsigned sa = (unsigned)(uintptr_t)(stg + px * PITCH + cv * 16);
#if YS_EPI_LDS_NOALIAS == 2
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "=&v"(val) : "v"(sa) : "memory");
#else
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(val) : "v"(sa) : "memory");
#endif
        }
#else
        val = *(const uint4*)(stg + px * PITCH + cv * 16);
#endif
        float f[8];
        ys_unpack<T>(val, f);
        if (do_stats) {
#if YS_EPI_SCALAR_STATS && !defined(YS_EMU_BUILD)
          // triage (round-4 determinism bisection): the sums through single v_add_f32 / v_fmac_f32 (inline asm: hipcc cannot pair them into v_pk_*_f32)
#pragma unroll
          for (int e = 0; e < 8; e++) {
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(s1[e]) : "v"(f[e]));
            asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(s2[e]) : "v"(f[e]));
          }
#else
#pragma unroll
          for (int e = 0; e < 8; e++) { s1[e] += f[e]; s2[e] += f[e] * f[e]; }
#endif
        }
        if (bn_eval) {
#pragma unroll
          for (int e = 0; e < 8; e++) f[e] = f[e] * sc[e] + sh[e];
          if (a.act) {                        // one uniform branch around the unrolled loop, not one per element
#pragma unroll
            for (int e = 0; e < 8; e++) f[e] = ys_silu(f[e]);
          }
#pragma unroll
          for (int e = 0; e < 8; e++) if (c + e >= a.Cout) f[e] = 0.f;
          if (!(rb || accum)) val = ys_pack<T>(f);
        }
        T* yp = (T*)(yb + rofs + c * 2);
        if (rb || accum) {
          float gq[8];
          if (rb) {
            const long row = (long)rowtab[NPX + px];                              // eval-only path (Bottleneck shortcut)
            ys_unpack<T>(ys_ld16(rb + (row * a.res_ldc + a.res_coff + c) * 2L), gq);
#pragma unroll
            for (int e = 0; e < 8; e++) f[e] += gq[e];
          }
          if (accum) {
            if constexpr (ACC_PRE) ys_unpack<T>(ov[ACC_PRE ? it : 0], gq); else ys_unpack<T>(ys_ld16(yp), gq);
#pragma unroll
            for (int e = 0; e < 8; e++) f[e] += gq[e];
          }
          val = ys_pack<T>(f);
        }
        if (RED && red_on && ry && !EPI_DBG(512)) {
          // dz as every later reader sees it (bf16-rounded, all contributions in), against the producer's raw output y
          float g[8], yf[8];
          ys_unpack<T>(val, g);
          ys_unpack<T>(yv[RED ? it : 0], yf);
          if (ract) {
#pragma unroll
            for (int e = 0; e < 8; e++) g[e] *= ys_silu_grad(yf[e] * rsc[e] + rsh[e]);
          }
#pragma unroll
          for (int e = 0; e < 8; e++) { s1[e] += g[e]; s2[e] += g[e] * yf[e]; }
        }
      }
    }
    ys_bufst16(rsY, (rofs != YS_BUF_OOB && !EPI_DBG(256)) ? rofs + (unsigned)c * 2u : YS_BUF_OOB, val);
  };
  if (EPI_DBG(1024)) { ys_wave_sync(); return; }
  if (RED) {
#pragma unroll                                // fully: yv[it] must be a register, not an indexed (scratch) array
    for (int it = 0; it < NITER; it++) store_iter(it, EpiMode<2>{});
  } else {
#if YS_EPI_FULL_UNROLL
#pragma unroll
    for (int it = 0; it < NITER; it++) store_iter(it, EpiMode<2>{});
#else
#pragma unroll 2
    for (int it = 0; it < NITER; it++) store_iter(it, EpiMode<2>{});
#endif
  }
  stamp();
  ys_wave_sync();
}

// ---------------------------------------------------------------------------------------------------------------------------
// Direct epilogue (round 3, default: YS_P2_EPI_DIRECT = 1).  After the MFMA a lane (li, q) already owns 4 CONSECUTIVE channels
// (n0 + nf*16 + 4q ..) of pixel mf*16 + li for every (mf, nf) fragment, i.e. 8 contiguous bytes of the bf16 NHWC row.  The staged
// epilogue above transposes through LDS to reach 16-byte stores: s_memtime stamps put that at 3.0-4.9 thousand cycles per tile
// (row table ~1k, staging writes 0.5-0.9k, two LDS round trips + the store loop 1.3-2.3k) against 2.5-3.5k for the whole K loop,
// and phase ablation at 1.8 of conv_p2_kernel's 3.9 ms per YOLOv8n step, of which the stores themselves are 0.43 ms
// (profiles/README.md, round 3).  Here the accumulators are rounded, reduced into the statistics and stored straight from registers
// as 8-byte buffer stores -- no LDS, no row table, no wave rendezvous; the four q-lanes of a pixel write one contiguous 32-byte
// sector per fragment.  Statistics are per-lane sums for the lane's own 4*NR channels (st[nf*4 + r]); the 16 pixel lanes of a DPP row
// are combined once per launch (p2_stats_flush_direct).
// RED = 1 (backward form): gradient accumulation and the fused BN-backward reduction, operands prefetched for all fragments before
// the first is consumed.  RED = 0 (forward / eval form): bias, eval-BN, SiLU, residual, BN statistics (+ an in-loop accumulate for
// the launches that have no RED variant).
#ifndef YS_P2_EPI_DIRECT
#define YS_P2_EPI_DIRECT 0     // measured (round 3, MI355X, config 2): staged 10.19-10.21 ms/step; direct with 8-byte stores 10.93 (against 10.65 then); direct with the 16-lane row swap (YS_EPI_SWAP16, 16-byte stores from registers) 10.58-10.60 -- the 4*NR-per-lane statistics and the masked selects cost the 168-register variants more (12-160 B of scratch) than the LDS round trip
#endif
#ifndef YS_EPI_SWAP16
#define YS_EPI_SWAP16 1        // direct epilogue: fragment pairs trade 16-lane rows (v_permlane16_swap_b32) so that every lane stores 16 contiguous bytes
#endif
__device__ inline uint2 ys_ld8(const void* p) { return *(const uint2*)p; }
// v_permlane16_swap_b32 (gfx950): the odd 16-lane rows of `a` trade places with the even rows of `b`:
//   a' = [a.row0, b.row0, a.row2, b.row2],  b' = [a.row1, b.row1, a.row3, b.row3]      (probe: tools/dev/permlane_swap_probe.hip)
// With a = lane (li, q)'s four packed channels 4q.. of fragment column n and b = the same of column n + 1, lane q = 0 ends up with
// channels 0-7 of column n (its own four + q = 1's), q = 1 with channels 0-7 of column n + 1, q = 2 / 3 with channels 8-15: the 8-byte
// pieces of the MFMA layout become 16-byte pieces of the NHWC row without LDS.
__device__ inline void ys_row_swap(unsigned& a, unsigned& b) {
#ifdef YS_EMU_BUILD
  const int lane = threadIdx.x & 63;
  const bool odd = (lane >> 4) & 1;
  const unsigned pa = __shfl(a, lane ^ 16), pb = __shfl(b, lane ^ 16);
  const unsigned na = odd ? pb : a, nb = odd ? b : pa;
  a = na; b = nb;
#else
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0]; b = r[1];
#endif
}
__device__ inline void ys_unpack4_bf16(const uint2& v, float* f) {
  f[0] = ys_u2f(v.x << 16); f[1] = ys_u2f(v.x & 0xffff0000u); f[2] = ys_u2f(v.y << 16); f[3] = ys_u2f(v.y & 0xffff0000u);
}
// FMODE (forward form only): 2 = every run-time option (bias, eval BatchNorm, SiLU, residual, accumulate); 0 = the training forward of a BatchNorm unit -- raw
// output + statistics, nothing else -- as a compile-time fact: with the options tested per fragment at run time the forward form was ~450 scalar branches and
// ~1900 v_readlane reloads of spilled kernel-argument SGPRs per call (conv_halo_kernel, 40 fragments: 23 thousand cycles per 256 x 160 tile).
template <int MR, int NR, int RED = 0, class SF = YsNoStamp, int FMODE = 2, int DEPTH = 2>
__device__ inline void p2_epilogue_direct(const ConvArgs& a, f32x4 (&acc)[MR][NR], const int (&orow)[MR], const bool (&pv)[MR],
                                          int n0, float (&s1)[4 * NR], float (&s2)[4 * NR], SF stamp = SF()) {
  const int lane = threadIdx.x & 63, q = lane >> 4;
  const ys_rsrcv_t rsY = ys_make_rsrcv(a.y, 0x7ffffff0u);
  unsigned roff[MR];                           // byte offset of this lane's pixel row in the output view (plans: < 2^31)
#pragma unroll
  for (int mf = 0; mf < MR; mf++)
    roff[mf] = pv[mf] ? ((unsigned)orow[mf] * (unsigned)a.out_ldc + (unsigned)a.out_coff) * 2u : YS_BUF_OOB;
  // 16-byte stores of two 8-byte pieces A (even-q lanes keep it) and B (odd-q lanes keep it): after the row swap a lane holds 8
  // consecutive channels starting at c8 = column base + (q >> 1) * 8 of ITS piece; rowE / rowO = byte offset of the pixel row the even /
  // odd lanes write (YS_BUF_OOB = pixel outside), colE / colO = first channel of the fragment column of A / B.
  const bool q_odd = (q & 1) != 0;
  auto store_pair = [&](uint2 A, uint2 B, unsigned rowE, unsigned rowO, int colE, int colO) {
    ys_row_swap(A.x, B.x);
    ys_row_swap(A.y, B.y);
    const unsigned row = q_odd ? rowO : rowE;
    const int c8 = (q_odd ? colO : colE) + (q >> 1) * 8;
    ys_bufst16(rsY, (row != YS_BUF_OOB && c8 < a.Cout && !EPI_DBG(256)) ? row + (unsigned)c8 * 2u : YS_BUF_OOB, make_uint4(A.x, A.y, B.x, B.y));
  };
  uint2 pkE[MR];                               // even fragment columns wait here for their odd neighbour
  if (RED) {
    // ---- backward form.  Operands of fragment column nf + 1 (old dz, the producer's y, its BN coefficients) are requested before
    // column nf is consumed: two batches of 4*MR + 8 registers in flight instead of every fragment's (which spilled), one exposed
    // memory latency per tile instead of one per column.
    // FMODE >= 4: (accumulate, fused reduction) = (FMODE & 1, FMODE & 2) are compile-time facts of the launch (conv_halo_kernel)
    constexpr bool CT = FMODE >= 4;
    const bool accum_rt = CT ? (FMODE & 1) != 0 : a.accumulate != 0;
    const bool red_on = CT ? (FMODE & 2) != 0 : (a.nred > 0 && !EPI_DBG(512));
    auto issue = [&](auto nfc, uint2 (&b_old)[MR], uint2 (&b_y)[MR], float (&b_sc)[4], float (&b_sh)[4], bool& b_has, bool& b_act) {
      constexpr int nf = decltype(nfc)::value;
      const int c = n0 + nf * 16 + 4 * q;
      const char* ry = nullptr; const float* scp = nullptr; const float* shp = nullptr; int rC = 0, rcol = 0; bool ract = false;
      const ys_redp_t rt = ys_red_table(a);
#pragma unroll
      for (int k = 0; k < YS_BNRED_MAXSEG; k++)
        if (red_on && k < a.nred && c >= rt[k].c0 && c < rt[k].c1) {
          ry = (const char*)rt[k].y; rC = rt[k].C; rcol = c - rt[k].c0 + rt[k].yc0; ract = rt[k].act != 0;
          scp = rt[k].scale + rcol; shp = rt[k].shift + rcol;
        }
      b_has = ry != nullptr; b_act = ract;
#pragma unroll
      for (int mf = 0; mf < MR; mf++) {
        const bool ok = (bool)((int)pv[mf] & (int)(c < a.Cout));
        b_old[mf] = accum_rt ? ys_bufld8(rsY, ok ? roff[mf] + (unsigned)c * 2u : YS_BUF_OOB) : make_uint2(0u, 0u);
        // unconditional load: a lane without a segment / pixel reads a valid dummy address and is masked when consumed
        const char* yp = (ry && ok) ? ry + ((long)orow[mf] * rC + rcol) * 2L : (const char*)a.y;
        b_y[mf] = red_on ? ys_ld8(yp) : make_uint2(0u, 0u);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) { b_sc[r] = 0.f; b_sh[r] = 0.f; }
      if (ry) { ys_ldcoef<4>(scp, b_sc); ys_ldcoef<4>(shp, b_sh); }
    };
    auto consume = [&](auto nfc, const uint2 (&b_old)[MR], const uint2 (&b_y)[MR], const float (&b_sc)[4], const float (&b_sh)[4], const bool b_has, const bool b_act) {
      constexpr int nf = decltype(nfc)::value;
      const int c = n0 + nf * 16 + 4 * q;
#pragma unroll
      for (int mf = 0; mf < MR; mf++) {
        const bool ok = (bool)((int)pv[mf] & (int)(c < a.Cout));
        float v[4], o[4];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = (c + r < a.Cout) ? acc[mf][nf][r] : 0.f;   // channels past Cout inside the last 4-group stay zero
        if (accum_rt) {
          ys_unpack4_bf16(b_old[mf], o);
#pragma unroll
          for (int r = 0; r < 4; r++) v[r] += o[r];
        }
        uint2 pk;
        pk.x = ys_pack_bf16x2(v[0], v[1]); pk.y = ys_pack_bf16x2(v[2], v[3]);
        if (b_has) {
          // dz as every later reader sees it (bf16-rounded, all contributions in), against the producer's raw output y
          float g[4], yf[4];
          ys_unpack4_bf16(pk, g);
          ys_unpack4_bf16(b_y[mf], yf);
          if (b_act) {
#pragma unroll
            for (int r = 0; r < 4; r++) g[r] *= ys_silu_grad(yf[r] * b_sc[r] + b_sh[r]);
          }
#pragma unroll
          for (int r = 0; r < 4; r++) { const float du = ok ? g[r] : 0.f; s1[nf * 4 + r] += du; s2[nf * 4 + r] += ok ? du * yf[r] : 0.f; }
        }
        if constexpr (!YS_EPI_SWAP16) ys_bufst8(rsY, (ok && !EPI_DBG(256)) ? roff[mf] + (unsigned)c * 2u : YS_BUF_OOB, pk);
        else if constexpr ((nf & 1) == 0 && nf + 1 < NR) pkE[mf] = pk;
        else if constexpr (nf & 1) store_pair(pkE[mf], pk, roff[mf], roff[mf], n0 + (nf - 1) * 16, n0 + nf * 16);
        else {                                 // last, unpaired column: pixel rows mf / mf + 1 pair up instead
          if ((mf & 1) == 0 && mf + 1 < MR) pkE[mf] = pk;
          else if (mf & 1) store_pair(pkE[mf - 1], pk, roff[mf - 1], roff[mf], n0 + nf * 16, n0 + nf * 16);
          else ys_bufst8(rsY, (ok && !EPI_DBG(256)) ? roff[mf] + (unsigned)c * 2u : YS_BUF_OOB, pk);
        }
      }
    };
    // DEPTH operand batches in rotation: column nf's operands are requested DEPTH - 1 columns ahead.  2 (default): the register budget of two waves per SIMD;
    // NR: everything up front -- one exposed memory latency per tile -- for a kernel with one wave per SIMD and registers to spare (conv_halo_kernel: with two batches
    // its accumulate launches spent ~35 thousand cycles per 256 x 160 tile in five dependent round trips).
    uint2 oldv[DEPTH][MR], yv[DEPTH][MR];
    float scv[DEPTH][4], shv[DEPTH][4];
    bool hasv[DEPTH], actv[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) { hasv[d] = false; actv[d] = false; }
    ys_static_for<0, (DEPTH - 1 < NR ? DEPTH - 1 : NR)>([&](auto nfc) {
      constexpr int nf = decltype(nfc)::value;
      issue(nfc, oldv[nf % DEPTH], yv[nf % DEPTH], scv[nf % DEPTH], shv[nf % DEPTH], hasv[nf % DEPTH], actv[nf % DEPTH]);
    });
    stamp();
    ys_static_for<0, NR>([&](auto nfc) {
      constexpr int nf = decltype(nfc)::value;
      // the scheduling fences keep hipcc from hoisting every column's loads to the top (all batches live at once: spills)
      if constexpr (nf + DEPTH - 1 < NR) {
        constexpr int nx = nf + DEPTH - 1;
        issue(std::integral_constant<int, nx>{}, oldv[nx % DEPTH], yv[nx % DEPTH], scv[nx % DEPTH], shv[nx % DEPTH], hasv[nx % DEPTH], actv[nx % DEPTH]);
      }
      YS_SCHED_FENCE();
      consume(nfc, oldv[nf % DEPTH], yv[nf % DEPTH], scv[nf % DEPTH], shv[nf % DEPTH], hasv[nf % DEPTH], actv[nf % DEPTH]);
      YS_SCHED_FENCE();
    });
    stamp(); stamp(); stamp();
    return;
  }
  // ---- forward / eval form
  // FMODE 0: training forward of a BatchNorm unit (raw output + statistics); 1 (round 6): eval forward of a BatchNorm unit -- folded scale / shift, SiLU when a.act,
  // nothing else (no statistics, bias, residual or accumulation) -- as compile-time facts; 2: every option tested at run time
  constexpr bool TRAIN = FMODE == 0, EVAL = FMODE == 1;
  const bool do_stats = TRAIN ? true : (EVAL ? false : (a.stats != nullptr && !EPI_DBG(512)));
  const bool bn_eval = TRAIN ? false : (EVAL ? true : a.scale != nullptr);
  const bool bias = (TRAIN || EVAL) ? false : (!a.scale && a.shift);
  const char* rb = (TRAIN || EVAL) ? nullptr : (const char*)a.res;
  const bool accum = (TRAIN || EVAL) ? false : a.accumulate != 0;
  const bool shift_on = TRAIN ? false : (EVAL ? true : a.shift != nullptr);
  const bool act_on = TRAIN ? false : a.act != 0;
  stamp();
#pragma unroll
  for (int nf = 0; nf < NR; nf++) {
    const int c = n0 + nf * 16 + 4 * q;
    const int cc = c < a.Cout ? c : 0;         // coefficient arrays are padded to a multiple of 4 floats; Cout % 4 == 0 on this path
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (bn_eval) ys_ldcoef<4>(a.scale + cc, sc);
    if (shift_on) ys_ldcoef<4>(a.shift + cc, sh);
#pragma unroll
    for (int mf = 0; mf < MR; mf++) {
      const bool ok = (bool)((int)pv[mf] & (int)(c < a.Cout));
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = acc[mf][nf][r];
      if (bias) {                              // plain conv bias (heads): added to the fp32 accumulators
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] += sh[r];
        if (act_on) {
#pragma unroll
          for (int r = 0; r < 4; r++) v[r] = ys_silu(v[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; r++) if (c + r >= a.Cout) v[r] = 0.f;   // channels past Cout inside the last 4-group stay zero (Pose: 51 outputs)
      uint2 pk;
      pk.x = ys_pack_bf16x2(v[0], v[1]); pk.y = ys_pack_bf16x2(v[2], v[3]);
      if (do_stats || bn_eval || rb || accum) {
        float f[4];
        ys_unpack4_bf16(pk, f);                // the rounded conv output: what BN normalises / what the statistics describe
        if (do_stats) {
#pragma unroll
          for (int r = 0; r < 4; r++) { const float t = ok ? f[r] : 0.f; s1[nf * 4 + r] += t; s2[nf * 4 + r] += t * t; }
        }
        if (bn_eval) {
#pragma unroll
          for (int r = 0; r < 4; r++) f[r] = f[r] * sc[r] + sh[r];
          if (act_on) {
#pragma unroll
            for (int r = 0; r < 4; r++) f[r] = ys_silu(f[r]);
          }
        }
        if (rb || accum) {
          float gq[4];
          if (rb) {                            // eval-only path (Bottleneck shortcut)
            const char* rp = ok ? rb + ((long)orow[mf] * a.res_ldc + a.res_coff + c) * 2L : rb;
            ys_unpack4_bf16(ys_ld8(rp), gq);
#pragma unroll
            for (int r = 0; r < 4; r++) f[r] += gq[r];
          }
          if (accum) {
            ys_unpack4_bf16(ys_bufld8(rsY, ok ? roff[mf] + (unsigned)c * 2u : YS_BUF_OOB), gq);
#pragma unroll
            for (int r = 0; r < 4; r++) f[r] += gq[r];
          }
        }
        if (bn_eval || rb || accum) {
#pragma unroll
          for (int r = 0; r < 4; r++) if (c + r >= a.Cout) f[r] = 0.f;
          pk.x = ys_pack_bf16x2(f[0], f[1]); pk.y = ys_pack_bf16x2(f[2], f[3]);
        }
      }
      if (!YS_EPI_SWAP16) ys_bufst8(rsY, (ok && !EPI_DBG(256)) ? roff[mf] + (unsigned)c * 2u : YS_BUF_OOB, pk);
      else if ((nf & 1) == 0 && nf + 1 < NR) pkE[mf] = pk;
      else if (nf & 1) store_pair(pkE[mf], pk, roff[mf], roff[mf], n0 + (nf - 1) * 16, n0 + nf * 16);
      else {                                   // last, unpaired column: pixel rows mf / mf + 1 pair up instead
        if ((mf & 1) == 0 && mf + 1 < MR) pkE[mf] = pk;
        else if (mf & 1) store_pair(pkE[mf - 1], pk, roff[mf - 1], roff[mf], n0 + nf * 16, n0 + nf * 16);
        else ys_bufst8(rsY, (ok && !EPI_DBG(256)) ? roff[mf] + (unsigned)c * 2u : YS_BUF_OOB, pk);
      }
    }
  }
  stamp(); stamp(); stamp();
}

// One statistics row per workgroup for the direct layout: lane (li, q) of wave w holds the sums of channels nf*16 + 4q + r over its
// own pixels.  The 16 pixel lanes of a DPP row are combined in registers (ys_row16_sum: 4 VALU adds per value), lane li == 0 of every
// row parks its 8*NR totals in LDS ([2][waves][BN] floats), and one thread per (sum, channel) adds the waves' entries in a fixed
// order.  WM x WN wave grids (conv_gemm_kernel): wave w = wm * WN + wn owns channels wn*NR*16 .. of the tile; WN = 1 for conv_p2_kernel.
template <int NR, int WM, int WN>
__device__ inline void p2_stats_flush_direct(const ConvArgs& a, int n0, float (&s1)[4 * NR], float (&s2)[4 * NR], float* scr, long stat_row) {
  constexpr int BNW = NR * 16, BN = WN * BNW, NT = WM * WN * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
  const int wm = wave / WN, wn = wave - wm * WN;
#pragma unroll
  for (int i = 0; i < 4 * NR; i++) { s1[i] = ys_row16_sum(s1[i]); s2[i] = ys_row16_sum(s2[i]); }
  ys_barrier_lds();                           // the LDS region's previous use (patch / operand stages) is over
  if (li == 0) {
#pragma unroll
    for (int nf = 0; nf < NR; nf++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int ch = wn * BNW + nf * 16 + 4 * q + r;
        scr[(0 * WM + wm) * BN + ch] = s1[nf * 4 + r];
        scr[(1 * WM + wm) * BN + ch] = s2[nf * 4 + r];
      }
  }
  ys_barrier_lds();
  for (int o = tid; o < 2 * BN; o += NT) {
    const int which = o / BN, c = o - which * BN;
    float t = scr[(which * WM) * BN + c];
#pragma unroll
    for (int w = 1; w < WM; w++) t += scr[(which * WM + w) * BN + c];
    if (a.nred) ys_bnred_put(a, n0 + c, which, stat_row, t);
    else if (n0 + c < a.Cout) { if (a.stat_acc) ys_stat_acc_add(a.stat_acc, stat_row, a.Cout, n0 + c, which, t); else a.stats[(stat_row * 2 + which) * a.Cout + n0 + c] = t; }
  }
}

// One statistics row per workgroup: the per-lane column sums gathered over all of its tiles go through LDS ([16][NT] floats in
// the patch region, lane-contiguous -> conflict-free) and one thread per (sum, channel) adds its NW * PPI entries in a fixed
// order.  (The earlier form -- 16 values x log2(64 / VPP) rounds of cross-lane shuffles per lane -- was a 4-5 thousand cycle
// chain of dependent ds_bpermutes at the end of every forward launch.)
template <int NR, int NW>
__device__ inline void p2_stats_flush(const ConvArgs& a, int n0, float (&s1)[8], float (&s2)[8], float* scr, long stat_row) {
  constexpr int BN = NR * 16;
  constexpr int VPP = BN / 8;
  constexpr int PPI = 64 / VPP;
  constexpr int NT = NW * 64;
  constexpr int NE = NW * PPI;                // lane entries per (sum, channel)
  constexpr int P = NT / (2 * BN) >= 1 ? NT / (2 * BN) : 1;   // threads sharing one output: the serial chain of NE dependent LDS reads
  constexpr int LEN = (NE + P - 1) / P;       // (a 3-4 thousand cycle tail of every forward launch) becomes NE / P + P
  const int tid = threadIdx.x;
  ys_barrier_lds();                           // the last tile's epilogue staging (same LDS region) is consumed
#pragma unroll
  for (int e = 0; e < 8; e++) { scr[e * NT + tid] = s1[e]; scr[(8 + e) * NT + tid] = s2[e]; }
  ys_barrier_lds();
  float* part = scr + 16 * NT;                // [P][2 * BN]
  {
    const int o = tid % (2 * BN), pi = tid / (2 * BN);
    if (pi < P) {
      const int which = o / BN, c = o - which * BN;
      const int cv = c >> 3, e = c & 7;
      const float* col = scr + (which * 8 + e) * NT + cv;
      float t = 0.f;
#pragma unroll 4
      for (int k = pi * LEN; k < (pi + 1) * LEN && k < NE; k++) {   // lanes cv, cv + VPP, ... of wave 0, then wave 1, ...: lane index = (k / PPI) * 64 + (k % PPI) * VPP
        const int w = k / PPI, j = k - w * PPI;
        t += col[w * 64 + j * VPP];
      }
      part[pi * 2 * BN + o] = t;
    }
  }
  ys_barrier_lds();
  for (int o = tid; o < 2 * BN; o += NT) {
    const int which = o / BN, c = o - which * BN;
    float t = part[o];
#pragma unroll
    for (int pi = 1; pi < P; pi++) t += part[pi * 2 * BN + o];      // fixed order -> deterministic
    if (a.nred) ys_bnred_put(a, n0 + c, which, stat_row, t);
    else if (n0 + c < a.Cout) { if (a.stat_acc) ys_stat_acc_add(a.stat_acc, stat_row, a.Cout, n0 + c, which, t); else a.stats[(stat_row * 2 + which) * a.Cout + n0 + c] = t; }
  }
}


// Same flush for a workgroup whose waves form a WM x WN grid over (pixels, channels): wave w = wm * WN + wn covers channels
// n0 + wn * NR * 16 .. of the workgroup's tile, so a column's partial sums live in the WM waves of one wave column.
template <int NR, int WM, int WN>
__device__ inline void conv_stats_flush_grid(const ConvArgs& a, int n0, float (&s1)[8], float (&s2)[8], float* scr, long stat_row) {
  constexpr int BNW = NR * 16, BN = WN * BNW;
  constexpr int VPP = BNW / 8;
  constexpr int PPI = 64 / VPP;
  constexpr int NT = WM * WN * 64;
  const int tid = threadIdx.x;
  ys_barrier_lds();                           // the last tile's epilogue staging (same LDS region) is consumed
#pragma unroll
  for (int e = 0; e < 8; e++) { scr[e * NT + tid] = s1[e]; scr[(8 + e) * NT + tid] = s2[e]; }
  ys_barrier_lds();
  for (int o = tid; o < 2 * BN; o += NT) {
    const int which = o / BN, c = o - which * BN;
    const int wn = c / BNW, cw = c - wn * BNW;
    const int cv = cw >> 3, e = cw & 7;
    const float* col = scr + (which * 8 + e) * NT + cv;
    float t = 0.f;
    for (int wm = 0; wm < WM; wm++)
      for (int j = 0; j < PPI; j++) t += col[(wm * WN + wn) * 64 + j * VPP];
    if (a.nred) ys_bnred_put(a, n0 + c, which, stat_row, t);
    else if (n0 + c < a.Cout) { if (a.stat_acc) ys_stat_acc_add(a.stat_acc, stat_row, a.Cout, n0 + c, which, t); else a.stats[(stat_row * 2 + which) * a.Cout + n0 + c] = t; }
  }
}
