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
// destination of one fused BN-backward partial sum: output-view channel c of the launch, `which` = 0 (sum du) / 1 (sum du * y)
__device__ inline float* ys_bnred_dst(const ConvArgs& a, int c, int which, long row) {
  float* d = nullptr;
#pragma unroll
  for (int k = 0; k < YS_BNRED_MAXSEG; k++)     // static indices: the segment table stays in scalar registers
    if (k < a.nred && c >= a.red[k].c0 && c < a.red[k].c1)
      d = a.red[k].part + (((long)a.red_row0 + row) * 2 + which) * a.red[k].C + (c - a.red[k].c0 + a.red[k].yc0);
  return d;
}

// RED = 1: the dgrad form -- gradient accumulation and the fused BN-backward reduction only (no bias / eval-BN / SiLU / residual:
// a dgrad launch never carries them), compiled as its own kernel variants so that the forward kernels' register allocation is
// untouched by the reduction's live values (y vectors, coefficients).
template <int MR, int NR, int RED = 0>
__device__ inline void p2_epilogue(const ConvArgs& a, f32x4 (&acc)[MR][NR], const long (&orow)[MR], const bool (&pv)[MR],
                                   int n0, char* stg, float (&s1)[8], float (&s2)[8]) {
  typedef bf16_t T;
  constexpr int BN = NR * 16;
  constexpr int PITCH = (BN + 8) * 2;         // bytes per staged pixel row
  constexpr int NPX = 16 * MR;
  constexpr int VPP = BN / 8;                  // 16-byte vectors per pixel
  constexpr int PPI = 64 / VPP;                // pixels per wave iteration
  constexpr int NITER = (NPX + PPI - 1) / PPI;
  const int lane = threadIdx.x & 63, li = lane & 15, q = lane >> 4;
  long* rowtab = (long*)(stg + NPX * PITCH);
  const int cv = lane % VPP, pl = lane / VPP;
  const bool active = lane < PPI * VPP;
  const int c = n0 + cv * 8;
  if (q == 0) {
#pragma unroll
    for (int mf = 0; mf < MR; mf++) {
      rowtab[mf * 16 + li] = pv[mf] ? (orow[mf] * a.out_ldc + a.out_coff) * 2L : -1L;   // byte offset of the pixel row
      rowtab[NPX + mf * 16 + li] = orow[mf];                                              // row index (residual view, producer's y)
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
      if (ry && px < NPX && rowtab[px] >= 0) yv[it] = ys_ld16(ry + (rowtab[NPX + px] * rC + rcol) * 2L);
    }
#pragma unroll
    for (int e = 0; e < 8; e++) { rsc[e] = 0.f; rsh[e] = 0.f; }
    if (ry) { ys_ldcoef<8>(rscp + rcol, rsc); ys_ldcoef<8>(rshp + rcol, rsh); }
  }
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
      }
#pragma unroll
      for (int r = 0; r < 4; r++) if (c + r >= a.Cout) v[r] = 0.f;   // padded channels of the output row stay zero
      uint2 pk;
      pk.x = ys_pack_bf16x2(v[0], v[1]);
      pk.y = ys_pack_bf16x2(v[2], v[3]);
      *(uint2*)(stg + (mf * 16 + li) * PITCH + (nf * 16 + 4 * q) * 2) = pk;
    }
  }
  ys_wave_sync();
  const bool do_stats = !RED && a.stats != nullptr;
  // eval-mode BatchNorm folded into the conv (Convs.cs:48 with running statistics): applied on the wide path to the
  // bf16-rounded conv output -- the same value the training path normalises -- with the lane's 8 coefficients loaded once
  const bool bn_eval = !RED && a.scale != nullptr;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { sc[e] = 1.f; sh[e] = 0.f; }
  if (bn_eval && active && c < a.Cout) {     // coefficient arrays are padded to a multiple of 4 floats; Cout % 8 == 0 for BN convs
    ys_ldcoef<8>(a.scale + c, sc);
    if (a.shift) ys_ldcoef<8>(a.shift + c, sh);
  }
  char* yb = (char*)a.y;
  const char* rb = RED ? nullptr : (const char*)a.res;
  // Stores go through a buffer descriptor of the output buffer: masked lanes (pixel outside the image, channel tile past Cout,
  // idle lanes of the wave) carry the out-of-range offset and the hardware drops them, so the store is issued unconditionally,
  // NITER times per tile -- a static count the compiler can keep in flight (vmcnt(N)) across the next tile's loads instead of the
  // vmcnt(0) that exec-masked stores forced.  Valid offsets are < 2^31 (checked by the launch plans).
  const ys_rsrcv_t rsY = ys_make_rsrcv(yb, 0x7ffffff0u);
  auto store_iter = [&](const int it) {
    const int px = it * PPI + pl;
    const bool lane_ok = active && px < NPX && c < a.Cout;
    const long rofs = lane_ok ? rowtab[px] : -1L;
    uint4 val = ys_zero16();
    {
      if (rofs >= 0) {
        val = *(const uint4*)(stg + px * PITCH + cv * 16);
        float f[8];
        ys_unpack<T>(val, f);
        if (do_stats) {
#pragma unroll
          for (int e = 0; e < 8; e++) { s1[e] += f[e]; s2[e] += f[e] * f[e]; }
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
          if (!(rb || a.accumulate)) val = ys_pack<T>(f);
        }
        T* yp = (T*)(yb + rofs + c * 2);
        if (rb || a.accumulate) {
          float gq[8];
          if (rb) {
            const long row = rowtab[NPX + px];                                    // eval-only path (Bottleneck shortcut)
            ys_unpack<T>(ys_ld16(rb + (row * a.res_ldc + a.res_coff + c) * 2L), gq);
#pragma unroll
            for (int e = 0; e < 8; e++) f[e] += gq[e];
          }
          if (a.accumulate) {
            ys_unpack<T>(ys_ld16(yp), gq);
#pragma unroll
            for (int e = 0; e < 8; e++) f[e] += gq[e];
          }
          val = ys_pack<T>(f);
        }
        if (RED && red_on && ry) {
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
    ys_bufst16(rsY, rofs >= 0 ? (unsigned)(rofs + (long)c * 2L) : YS_BUF_OOB, val);
  };
  if (RED) {
#pragma unroll                                // fully: yv[it] must be a register, not an indexed (scratch) array
    for (int it = 0; it < NITER; it++) store_iter(it);
  } else {
#pragma unroll 2
    for (int it = 0; it < NITER; it++) store_iter(it);
  }
  ys_wave_sync();
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
    if (a.nred) { float* d = ys_bnred_dst(a, n0 + c, which, stat_row); if (d) *d = t; }
    else if (n0 + c < a.Cout) a.stats[(stat_row * 2 + which) * a.Cout + n0 + c] = t;
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
    if (a.nred) { float* d = ys_bnred_dst(a, n0 + c, which, stat_row); if (d) *d = t; }
    else if (n0 + c < a.Cout) a.stats[(stat_row * 2 + which) * a.Cout + n0 + c] = t;
  }
}
