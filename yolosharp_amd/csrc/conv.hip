// conv.hip -- NHWC implicit-GEMM convolution for gfx950 (CDNA4), forward / dgrad / wgrad.
//
// Replaces the arithmetic of torch.nn.Conv2d inside Modules/Convs.cs:36-62 (Conv = conv2d, bias=False,
// p = k/2) and the plain biased Conv2d heads of Modules/Head.cs:47-50, plus their autograd
// (Amp.cs:348,370).  k in {1,3}, stride in {1,2}, groups = 1.
//
// Layout: activations NHWC (channel stride `ldc`, channel offset `coff` so producers write straight
// into concat buffers and chunk() is a view); weights [Cout][kh][kw][Cin] (K-contiguous per cout).
// GEMM view per launch:  D[cout][pixel] = sum_k W[cout][k] * X[pixel][k],  k = (kh,kw,ci).
// MFMA orientation is "weights as the row operand": a lane ends up with 4 consecutive output
// channels of ONE pixel -> 8-byte (bf16) / 16-byte (f32) NHWC stores, and BN batch statistics
// reduce across lanes with 4 xor-shuffles.
//   T = bf16_t : v_mfma_f32_16x16x32_bf16, one 16-byte fragment load feeds one MFMA (K = 32)
//   T = float  : v_mfma_f32_16x16x4_f32 x4 (exact f32 fma chain) -- the parity path
// Activations stream HBM -> VGPR fragments directly (no reuse across waves: waves split pixels);
// the weight tile of the current tap is staged in LDS once per workgroup and shared by 4 waves.
//
// The same kernel computes dgrad:  ih*DIV = oh*SA + kh - PAD  describes both the forward gather
// (SA = stride, DIV = 1, PAD = k/2) and the input-gradient gather of a strided conv
// (SA = 1, DIV = stride, PAD = k-1-k/2, spatially flipped + transposed weights).
#include "ys_internal.h"
#include "ys_kernels.h"
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

// ------------------------------------------------------------------ shared epilogue
// Lane (li,q) holds acc[mf][nf][r] = D[channel n0+nf*16+4q+r][pixel mf*16+li].  Per pixel fragment the wave rounds its
// 16 x BN tile to T into a wave-private LDS slice and writes it back out as full 16-byte vectors, consecutive lanes
// covering consecutive channels of one pixel row (whole 64-256 B rows per pixel instead of 8-byte fragments); the
// residual add (eval Bottleneck) and gradient accumulation (dgrad into an already written view) happen on that
// wide path.  BN batch statistics are taken from the rounded values in the fragment layout.
template <class T, int MR, int NR, int NW>
__device__ inline void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[MR][NR], const long (&orow)[MR], const bool (&mv)[MR],
                                     int n0, T* wst, float* sStat, long stat_row) {
  constexpr int EPL = Elem<T>::EPL;
  constexpr int BN = NR * 16;
  constexpr int BNP = BN + EPL;        // staging row pitch (elements), rows stay 16-byte aligned
  constexpr int VPP = BN / EPL;        // 16-byte vectors per pixel
  constexpr int NIT = (16 * VPP + 63) / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
  const bool do_stats = a.stats != nullptr;
  float s1[NR][4], s2[NR][4];
#pragma unroll
  for (int nf = 0; nf < NR; nf++)
#pragma unroll
    for (int r = 0; r < 4; r++) { s1[nf][r] = 0.f; s2[nf][r] = 0.f; }
  char* yb = (char*)a.y;
  const char* rb = (const char*)a.res;
#pragma unroll
  for (int mf = 0; mf < MR; mf++) {
#pragma unroll
    for (int nf = 0; nf < NR; nf++) {
      const int c = n0 + nf * 16 + 4 * q;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = acc[mf][nf][r];
      if (a.scale || a.shift) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int cc = (c + r) < a.Cout ? (c + r) : 0;
          const float sc = a.scale ? a.scale[cc] : 1.0f;
          const float sh = a.shift ? a.shift[cc] : 0.0f;
          v[r] = v[r] * sc + sh;
          if (a.act) v[r] = ys_silu(v[r]);
        }
      }
      T o[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (c + r >= a.Cout) v[r] = 0.f;     // padded channels of the output row stay zero
        o[r] = Elem<T>::from_f(v[r]);
      }
      if (do_stats && mv[mf]) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float f = Elem<T>::to_f(o[r]);
          s1[nf][r] += f;
          s2[nf][r] += f * f;
        }
      }
      T* wp = wst + li * BNP + nf * 16 + 4 * q;
      if (sizeof(T) == 2) {
        uint2 pk;
        pk.x = ys_pack_bf16x2(v[0], v[1]);
        pk.y = ys_pack_bf16x2(v[2], v[3]);
        *(uint2*)wp = pk;
      } else {
        *(uint4*)wp = make_uint4(ys_f2u(v[0]), ys_f2u(v[1]), ys_f2u(v[2]), ys_f2u(v[3]));
      }
    }
    ys_wave_sync();
    const unsigned rlo = (unsigned)(unsigned long)orow[mf], rhi = (unsigned)((unsigned long)orow[mf] >> 32);
    const int mvi = mv[mf] ? 1 : 0;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int vv = lane + 64 * it;
      const int px = vv < 16 * VPP ? vv / VPP : 0;
      const int cv = vv - px * VPP;
      const unsigned plo = __shfl(rlo, px), phi = __shfl(rhi, px);
      const int pok = __shfl(mvi, px);
      const int c = n0 + cv * EPL;
      if (vv < 16 * VPP && pok && c < a.Cout) {
        const long row = (long)(((unsigned long)phi << 32) | plo);
        uint4 val = *(const uint4*)(wst + px * BNP + cv * EPL);
        T* yp = (T*)(yb + (row * a.out_ldc + a.out_coff + c) * (long)sizeof(T));
        if (rb || a.accumulate) {
          float f[EPL], g[EPL];
          ys_unpack<T>(val, f);
          if (rb) {
            ys_unpack<T>(ys_ld16(rb + (row * a.res_ldc + a.res_coff + c) * (long)sizeof(T)), g);
#pragma unroll
            for (int e = 0; e < EPL; e++) f[e] += g[e];
          }
          if (a.accumulate) {
            ys_unpack<T>(ys_ld16(yp), g);
#pragma unroll
            for (int e = 0; e < EPL; e++) f[e] += g[e];
          }
          val = ys_pack<T>(f);
        }
        ys_st16(yp, val);
      }
    }
    ys_wave_sync();
  }
  if (do_stats) {
#pragma unroll
    for (int nf = 0; nf < NR; nf++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float x1 = s1[nf][r], x2 = s2[nf][r];
        for (int msk = 1; msk <= 8; msk <<= 1) {
          x1 += __shfl_xor(x1, msk);
          x2 += __shfl_xor(x2, msk);
        }
        if (li == 0) {
          sStat[((wave * BN) + nf * 16 + 4 * q + r) * 2 + 0] = x1;
          sStat[((wave * BN) + nf * 16 + 4 * q + r) * 2 + 1] = x2;
        }
      }
    __syncthreads();
    if (tid < BN && n0 + tid < a.Cout) {
      float t1 = 0.f, t2 = 0.f;
      for (int w = 0; w < NW; w++) { t1 += sStat[(w * BN + tid) * 2 + 0]; t2 += sStat[(w * BN + tid) * 2 + 1]; }
      if (a.stat_acc) { ys_stat_acc_add(a.stat_acc, stat_row, a.Cout, n0 + tid, 0, t1); ys_stat_acc_add(a.stat_acc, stat_row, a.Cout, n0 + tid, 1, t2); }
      else {
        a.stats[(stat_row * 2 + 0) * a.Cout + n0 + tid] = t1;
        a.stats[(stat_row * 2 + 1) * a.Cout + n0 + tid] = t2;
      }
    }
  }
}

template <class T, int MR, int NR>
__global__ void __launch_bounds__(256)
conv_igemm_kernel(ConvArgs a) {
  constexpr int EPL = Elem<T>::EPL;
  constexpr int BN = NR * 16;
  constexpr int LDS_UNITS = NR <= 2 ? 1024 : 3072;      // 16 / 48 KiB of staged weights
  constexpr int PITCH = ((LDS_UNITS / BN) - 1) | 1;     // odd row pitch in 16-byte units (bank spread)
  constexpr int GSTEPS = PITCH / 4;                     // k-steps whose weights fit in LDS at once
  __shared__ uint4 sW[BN * PITCH];
  __shared__ float sStat[4][BN][2];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 15;
  const int q = lane >> 4;
  const int m0 = blockIdx.x * (4 * MR * 16) + wave * (MR * 16);
  const int n0 = blockIdx.y * BN;
  const int HWo = a.Hout * a.Wout;
  const char* xb = (const char*)a.x;
  const char* wb = (const char*)a.w;
  const int taps = a.KH * a.KW;
  const int Ktot = taps * a.Cin;
  const int cu = a.Cin / EPL;        // 16-byte units per tap
  const int spt = (cu + 3) >> 2;     // k-steps per tap (4 units each, tail predicated off)
  const int total = taps * spt;

  // per m-fragment pixel coordinates of this lane's column (pixel j = li)
  int pb[MR], poh[MR], pow_[MR];
  bool pv[MR];
#pragma unroll
  for (int mf = 0; mf < MR; mf++) {
    const int m = m0 + mf * 16 + li;
    pv[mf] = m < a.M;
    const int mm = pv[mf] ? m : 0;
    const int b = mm / HWo;
    const int r = mm - b * HWo;
    pb[mf] = b;
    poh[mf] = r / a.Wout;
    pow_[mf] = r - poh[mf] * a.Wout;
  }

  f32x4 acc[MR][NR];
#pragma unroll
  for (int mf = 0; mf < MR; mf++)
#pragma unroll
    for (int nf = 0; nf < NR; nf++) acc[mf][nf] = f32x4_zero();

  // ---- activation stream: software-pipelined one k-step ahead (HBM/L2 -> VGPR fragments)
  const char* px[MR];
  bool pvt[MR];
  auto set_tap = [&](int kh, int kw) {
#pragma unroll
    for (int mf = 0; mf < MR; mf++) {
      const int ihn = poh[mf] * a.SA + kh - a.PAD;
      const int iwn = pow_[mf] * a.SA + kw - a.PAD - a.pad_w_delta;
      const int ih = ihn >> a.DIVS, iw = iwn >> a.DIVS;
      const bool ok = pv[mf] && ihn >= 0 && iwn >= 0 && ((ihn | iwn) & a.DIVM) == 0 && ih < a.Hin && iw < a.Win;
      pvt[mf] = ok;
      const long pix = ok ? ((long)pb[mf] * a.in_bstride + (long)ih * a.Win + iw) : 0;
      px[mf] = xb + (pix * a.in_ldc + a.in_coff) * (long)sizeof(T);
    }
  };
  // Loads are issued unconditionally (padding pixels point at pixel 0, a channel tail re-reads channel 0) and masked when the
  // fragment is consumed: a load under an exec mask makes the compiler wait with vmcnt(0) BEFORE the MFMAs of the current step,
  // which serialises the one-step-ahead prefetch with the arithmetic it is meant to hide behind.
  auto load_step = [&](uint4* xf, int s) -> unsigned {
    const int c = (4 * s + q) * EPL;
    const bool cok = c < a.Cin;
    const long cb = (long)(cok ? c : 0) * (long)sizeof(T);
    unsigned okm = 0;
#pragma unroll
    for (int mf = 0; mf < MR; mf++) {
      xf[mf] = ys_ld16(px[mf] + cb);
      okm |= (pvt[mf] && cok) ? (1u << mf) : 0u;
    }
    return okm;
  };
  uint4 xc[MR], xn[MR];
  int ltap = 0, ls = 0, lkh = 0, lkw = 0;   // position of the NEXT load
  set_tap(0, 0);
  {
    const unsigned m0k = load_step(xc, 0);
#pragma unroll
    for (int mf = 0; mf < MR; mf++) if (!((m0k >> mf) & 1u)) xc[mf] = ys_zero16();
  }
  int lt = 0;                                // k-step index inside the staged weight group
  for (int t = 0; t < total; t++) {
    // advance the load position and prefetch step t+1
    ls++;
    if (ls == spt) {
      ls = 0; ltap++; lkw++;
      if (lkw == a.KW) { lkw = 0; lkh++; }
      if (ltap < taps) set_tap(lkh, lkw);
    }
    const unsigned nmask = load_step(xn, ls);   // (past the last step: a harmless re-read)
    if (lt == 0) {
      // stage the weights of k-steps [t, t+GSTEPS) for this workgroup's BN output channels
      __syncthreads();
      const int gs = (total - t) < GSTEPS ? (total - t) : GSTEPS;
      const int gu = gs * 4;
      for (int idx = tid; idx < BN * gu; idx += 256) {
        const int n = idx / gu, lu = idx - n * gu;
        const int tt = t + (lu >> 2);
        const int tp = tt / spt;
        const int c = (((tt - tp * spt) << 2) + (lu & 3)) * EPL;
        uint4 v = ys_zero16();
        if (n0 + n < a.Cout && c < a.Cin)
          v = ys_ld16(wb + ((long)(n0 + n) * Ktot + (long)tp * a.Cin + c) * (long)sizeof(T));
        sW[n * PITCH + lu] = v;
      }
      __syncthreads();
    }
    const int u = 4 * lt + q;
#pragma unroll
    for (int nf = 0; nf < NR; nf++) {
      const uint4 wf = sW[(nf * 16 + li) * PITCH + u];
#pragma unroll
      for (int mf = 0; mf < MR; mf++) acc[mf][nf] = ys_mma<T>(wf, xc[mf], acc[mf][nf]);
    }
#pragma unroll
    for (int mf = 0; mf < MR; mf++) xc[mf] = ((nmask >> mf) & 1u) ? xn[mf] : ys_zero16();
    lt++;
    if (lt == GSTEPS) lt = 0;
  }

  // ------------------------------------------------------------------ epilogue (weights in LDS are dead: reuse as staging)
  __syncthreads();
  long orow[MR];
  bool mvv[MR];
#pragma unroll
  for (int mf = 0; mf < MR; mf++) {
    const int m = m0 + mf * 16 + li;
    mvv[mf] = m < a.M;
    const int mm = mvv[mf] ? m : 0;
    const int bb = mm / HWo;
    const int rr = mm - bb * HWo;
    orow[mf] = a.out_rh ? (long)bb * a.out_bstride + (long)(rr / a.Wout) * a.out_rh + (long)(rr % a.Wout) * a.out_rw + a.out_r0
                        : (long)bb * a.out_bstride + rr;
  }
  T* wst = (T*)sW + wave * (16 * (BN + EPL));
  conv_epilogue<T, MR, NR, 4>(a, acc, orow, mvv, n0, wst, &sStat[0][0][0], (long)blockIdx.x);
}

// ===================================================================================== 3x3: LDS patch kernel
// 3x3 convolutions (forward s1/s2 and dgrad s1/s2) re-use every input pixel up to 9 times.  Instead of fetching the
// operand fragments of each tap from L1/L2 (latency- and TA-bound), one workgroup owns a 2-D output tile TH x TW
// (<= 64*MR pixels) and stages, per 32-channel (bf16) / 16-channel (f32) chunk, the input PATCH = tile + halo once
// in LDS ([PH][PW] pixels x 4 sixteen-byte units, pitch 5 units) together with the chunk's weights of all 9 taps
// ([BN][9*4] units, pitch 37).  All global loads of a chunk are issued back-to-back by the 256 threads (deep
// memory-level parallelism), then the 9 taps x MR x NR MFMAs run from LDS only.  Zero padding and stride-2 dgrad
// holes are handled by zero-filled patch pixels / predicated fragments.
#define PATCH_UNITS 1664
#define C3_THREADS 512
// Persistent variant: grid.x workgroups (about one per CU) walk the tiles; when the weights of this workgroup's BN
// output channels fit in LDS for the whole K range (wres) they are staged ONCE per workgroup, otherwise per chunk.
// The global loads of the next (tile, chunk) are issued into registers before the MFMAs of the current one.
template <class T, int MR, int NR>
__global__ void __launch_bounds__(C3_THREADS)
conv3x3_tile_kernel(ConvArgs a, int ntiles, int nchunks, int wres, int wpitch, int patch_units) {
  constexpr int EPL = Elem<T>::EPL;
  constexpr int BN = NR * 16;
  constexpr int PP = 5;                                     // patch pixel pitch: 4 units + 1
  constexpr int PATCH_DATA = (PATCH_UNITS / PP) * 4;          // data units of the largest patch (pitch 5 holds 4)
  constexpr int NPU = (PATCH_DATA + C3_THREADS - 1) / C3_THREADS;    // patch units fetched per thread
  constexpr int NWU = (BN * 36 + C3_THREADS - 1) / C3_THREADS;       // weight units per thread (non-resident mode)
  YS_DYN_LDS(lds);
  uint4* sW = lds;                                          // [BN][wpitch]
  uint4* sP = lds + BN * wpitch;                            // [PATCH_UNITS]
  float* sStat = (float*)(sP + patch_units);                // [8][BN][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.y * BN;
  const char* xb = (const char*)a.x;
  const char* wb = (const char*)a.w;
  const int Ktot = 9 * a.Cin;

  if (wres) {
    const int per_row = nchunks * 36;
    for (int idx = tid; idx < BN * per_row; idx += C3_THREADS) {
      const int n = idx / per_row, cu = idx - n * per_row;
      const int chunk = cu / 36, tu = cu - chunk * 36;
      const int ch = chunk * 4 * EPL + (tu & 3) * EPL;
      uint4 v = ys_zero16();
      if (n0 + n < a.Cout && ch < a.Cin)
        v = ys_ld16(wb + ((long)(n0 + n) * Ktot + (long)(tu >> 2) * a.Cin + ch) * (long)sizeof(T));
      sW[n * wpitch + cu] = v;
    }
  }

  // All loads of a fetch are issued unconditionally (clamped to a valid address) and masked when the registers are written to
  // LDS: loads under exec masks make the compiler wait for them with vmcnt(0) before the MFMAs they are meant to overlap.
  uint4 rp[NPU], rw[NWU];
  unsigned pmask = 0, wmask = 0;
  const long safe_x = (long)a.in_coff * (long)sizeof(T);
  auto fetch = [&](int tile, int chunk) {
    int t = tile;
    const int txi = t % a.tiles_x; t /= a.tiles_x;
    const int tyi = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int oy0 = tyi * a.TH, ox0 = txi * a.TW;
    const int iy_lo = (oy0 * a.SA - a.PAD) >> a.DIVS, ix_lo = (ox0 * a.SA - a.PAD) >> a.DIVS;
    const int iy_hi = ((oy0 + a.TH - 1) * a.SA + 2 - a.PAD) >> a.DIVS, ix_hi = ((ox0 + a.TW - 1) * a.SA + 2 - a.PAD) >> a.DIVS;
    const int PH = iy_hi - iy_lo + 1, PW = ix_hi - ix_lo + 1;
    const int npatch = PH * PW * 4;
    const int c0 = chunk * 4 * EPL;
    pmask = 0;
#pragma unroll
    for (int k = 0; k < NPU; k++) {
      const int idx = tid + C3_THREADS * k;
      const int pix = idx >> 2, u = idx & 3;
      const int r = pix / PW, cc = pix - r * PW;
      const int iy = iy_lo + r, ix = ix_lo + cc;
      const int ch = c0 + u * EPL;
      const bool ok = (bool)((int)(idx < npatch) & (int)((unsigned)iy < (unsigned)a.Hin) & (int)((unsigned)ix < (unsigned)a.Win) &
                             (int)(ch < a.Cin) & (int)!(a.dbg & 1));
      long off = ((((long)b * a.in_bstride + (long)iy * a.Win + ix) * a.in_ldc) + a.in_coff + ch) * (long)sizeof(T);
      off = ok ? off : safe_x;
      rp[k] = ys_ld16(xb + off);
      pmask |= (unsigned)ok << k;
    }
    if (!wres) {
      wmask = 0;
#pragma unroll
      for (int k = 0; k < NWU; k++) {
        const int idx = tid + C3_THREADS * k;
        const int n = idx / 36, tu = idx - n * 36;
        const int ch = c0 + (tu & 3) * EPL;
        const bool ok = (bool)((int)(idx < BN * 36) & (int)(n0 + n < a.Cout) & (int)(ch < a.Cin));
        long off = ((long)(n0 + n) * Ktot + (long)(tu >> 2) * a.Cin + ch) * (long)sizeof(T);
        off = ok ? off : 0;
        rw[k] = ys_ld16(wb + off);
        wmask |= (unsigned)ok << k;
      }
    }
  };

  int tile = blockIdx.x, chunk = 0;
  if (tile < ntiles) fetch(tile, 0);
  // geometry of the tile being computed
  int b = 0, oy0 = 0, ox0 = 0, iy_lo = 0, ix_lo = 0, PW = 1;
  int poy[MR], pox[MR];
  bool pv[MR];
  f32x4 acc[MR][NR];
  while (tile < ntiles) {
    __syncthreads();                       // every wave finished reading the previous chunk
#pragma unroll
    for (int k = 0; k < NPU; k++) {
      const int idx = tid + C3_THREADS * k;
      if (idx < PATCH_DATA) sP[(idx >> 2) * PP + (idx & 3)] = ((pmask >> k) & 1u) ? rp[k] : ys_zero16();
    }
    if (!wres) {
#pragma unroll
      for (int k = 0; k < NWU; k++) {
        const int idx = tid + C3_THREADS * k;
        if (idx < BN * 36) { const int n = idx / 36; sW[n * wpitch + (idx - n * 36)] = ((wmask >> k) & 1u) ? rw[k] : ys_zero16(); }
      }
    }
    __syncthreads();
    // ---- prefetch the next (tile, chunk); past the end it re-fetches the current one (unconditional, see above)
    int ntile = tile, nchunk = chunk + 1;
    if (nchunk == nchunks) { nchunk = 0; ntile = tile + gridDim.x; }
    {
      const bool more = ntile < ntiles;
      fetch(more ? ntile : tile, more ? nchunk : chunk);
    }
    // ---- compute this chunk from LDS
    if (chunk == 0) {
      int t = tile;
      const int txi = t % a.tiles_x; t /= a.tiles_x;
      const int tyi = t % a.tiles_y;
      b = t / a.tiles_y;
      oy0 = tyi * a.TH; ox0 = txi * a.TW;
      iy_lo = (oy0 * a.SA - a.PAD) >> a.DIVS; ix_lo = (ox0 * a.SA - a.PAD) >> a.DIVS;
      const int ix_hi = ((ox0 + a.TW - 1) * a.SA + 2 - a.PAD) >> a.DIVS;
      PW = ix_hi - ix_lo + 1;
#pragma unroll
      for (int mf = 0; mf < MR; mf++) {
        const int p = wave * (MR * 16) + mf * 16 + li;
        const int ty = p / a.TW, tx = p - ty * a.TW;
        poy[mf] = oy0 + ty; pox[mf] = ox0 + tx;
        pv[mf] = p < a.TH * a.TW && poy[mf] < a.Hout && pox[mf] < a.Wout;
#pragma unroll
        for (int nf = 0; nf < NR; nf++) acc[mf][nf] = f32x4_zero();
      }
    }
    const int wbase = wres ? chunk * 36 : 0;
    if (!(a.dbg & 2))
#pragma unroll
    for (int kh = 0; kh < 3; kh++) {
#pragma unroll
      for (int kw = 0; kw < 3; kw++) {
        uint4 xf[MR];
#pragma unroll
        for (int mf = 0; mf < MR; mf++) {
          const int ihn = poy[mf] * a.SA + kh - a.PAD, iwn = pox[mf] * a.SA + kw - a.PAD;
          const bool ok = pv[mf] && ((ihn | iwn) & a.DIVM) == 0;
          const int r = (ihn >> a.DIVS) - iy_lo, cc = (iwn >> a.DIVS) - ix_lo;
          xf[mf] = ys_zero16();
          if (ok) xf[mf] = sP[(r * PW + cc) * PP + q];
        }
#pragma unroll
        for (int nf = 0; nf < NR; nf++) {
          const uint4 wf = sW[(nf * 16 + li) * wpitch + wbase + (kh * 3 + kw) * 4 + q];
#pragma unroll
          for (int mf = 0; mf < MR; mf++) acc[mf][nf] = ys_mma<T>(wf, xf[mf], acc[mf][nf]);
        }
      }
    }
    if (chunk == nchunks - 1 && !(a.dbg & 4)) {
      // ---------------------------------------------------------------- epilogue (patch region becomes the staging area)
      __syncthreads();
      long orow[MR];
#pragma unroll
      for (int mf = 0; mf < MR; mf++) orow[mf] = (long)b * a.out_bstride + (pv[mf] ? ((long)poy[mf] * a.Wout + pox[mf]) : 0);
      T* wst = (T*)sP + wave * (16 * (BN + EPL));
      conv_epilogue<T, MR, NR, C3_THREADS / 64>(a, acc, orow, pv, n0, wst, sStat, (long)tile);
    }  // last chunk of the tile
    tile = ntile; chunk = nchunk;
  }
}

// ===================================================================================== 3x3, bf16: whole-Cin LDS patch ("P2")
// Round-1 follow-up of the patch kernel above for bf16 (the f32 parity path keeps the chunked kernel).  Differences:
//  * the patch holds ALL input channels of the tile (+halo) in natural NHWC rows, so a tile needs one load phase and one
//    barrier pair instead of two barriers per 32-channel chunk;
//  * K runs over (tap, channel) in steps of 32 regardless of Cin (Cin = 16 packs two taps into one MFMA); the LDS offset
//    of every (K-step, q) fragment is tile-independent and comes from a small table built once per workgroup, and each
//    lane's pixel offsets are computed once per kernel -> the inner loop is 1 table read + adds + ds_read_b128 + MFMA;
//  * small weight sets stay resident (persistent grid); large ones stream through a double-buffered LDS ring, one barrier
//    per K-group, so the footprint stays <= ~75 KB and two workgroups share a CU (one loads while the other computes);
//  * 256 threads, <= 128 VGPRs: register pressure no longer caps occupancy.
// Handles forward (stride 1 and 2) and stride-1 dgrad (same gather, flipped weights); DIV = 2 dgrads stay on the old path.
#include "conv_epi.h"
#ifndef YS_EPI_BATCH_P2
#define YS_EPI_BATCH_P2(NPU_) 4
#endif

// ablation switches (YS_DBG bits) cost scalar checks in the hot loops: compiled in only for triage builds (-DYS_P2_ABLATE)
#ifdef YS_P2_ABLATE
#define P2_DBG(bit) ((a.dbg & (bit)) != 0)
#else
#define P2_DBG(bit) false
#endif
#ifndef YS_P2_GROUP_COPY
#define YS_P2_GROUP_COPY 1
#endif
#ifndef YS_P2_G1_MIN
#define YS_P2_G1_MIN 12        // register tiles of >= this many MFMAs per K-step run one K-step per LDS wait (12: two K-steps for the 2 x 5 tile -- spills at 256 registers)
#endif
#ifndef YS_P2_LATE_PFETCH
#define YS_P2_LATE_PFETCH 0    // 1: streamed-weight dgrad variants request the next patch inside the epilogue and run two K-steps per LDS wait (2 x 5 tile).
                               // Built, verified and measured (round 4, same box, alternating): grouped head launches 0.68 -> 0.67 ms, step 9.51 = 9.51 -- off
#endif
#ifndef YS_P2_RING3
#define YS_P2_RING3 0          // 1: streamed weights fetched three groups ahead, next patch requested after the K loop.  Built and measured (round 4, same box,
                               // alternating): grouped head launches 1.00 -> 0.96 ms, plain launches 2.96 -> 3.00 (more spills in the non-grouped variants), step 9.75 = 9.75:
                               // the K loop of the streamed layers is bound by its LDS round trips per K-step, not by the weight fetch -- off
#endif
#ifndef YS_P2_DMA_RING
#define YS_P2_DMA_RING 1       // 1: streamed bf16 weights go global -> LDS by DMA into a ring of three group slots, two groups ahead (no registers, no ds_write,
                               // no wait for the fetch inside the group) -- see DMAW in conv_p2_body
#endif
#ifndef YS_P2_DMA_G2
#define YS_P2_DMA_G2 1
#endif
#ifndef YS_P2_COUNTED_WAIT
#define YS_P2_COUNTED_WAIT 1   // the tile loop opens with s_waitcnt vmcnt(N), N = the epilogue's stores (0: vmcnt(0), rounds 1-3).  Relies on loads and stores
                               // retiring in issue order on the shared counter -- what hipcc's own wait insertion assumes on gfx9-family parts (it derives
                               // vmcnt(3), (2) from store counts itself when the store loop is rolled).  This switch was suspected and withdrawn once in
                               // round 4, when the config-5 determinism test failed with it -- the bisection (profiles/README.md) cleared it: counted wait +
                               // rolled store loop is bit-stable, full unroll WITHOUT it is not (packed-FP32 statistics), and with the convolution sources
                               // compiled -fno-slp-vectorize the combination is stable in 8 / 8 model-level rounds and every layer rerun.
#endif
#ifndef P2_KG
#define P2_KG 2            // K-steps (32 K each) per streamed weight group
#endif
#define P2_NPU 12          // max patch units (16 B) a thread keeps in flight: 12 x 256 x 16 B = 48 KB per workgroup (small layers: 6)
struct P2Tag0 { static constexpr int value = 0; };
struct P2Tag1 { static constexpr int value = 1; };
// K-steps per register group of the pipelined K loop: enough MFMAs per group (>= 8) to cover an LDS round trip
// G K-steps share one LDS wait: about 16 MFMAs per group, so that a group's MFMA time matches the LDS round trip the SIMD's other
// waves have to cover -- bounded by the fragment registers a group keeps live, 4 * G * (MR + NR): <= 64 in the 256-register
// variants, <= 32 in the `tight` ones (compiled for three waves per SIMD, 168 registers).  G is 1, 2 or 4 (one table read).
__host__ __device__ constexpr int p2_reg_group(int mr, int nr, bool tight = false, bool late = false) {
  const int want = mr * nr >= (late ? YS_P2_G1_MIN : 10) ? 1 : (mr * nr >= 6 ? 2 : 4);
  const int cap = (tight ? 8 : 16) / (mr + nr);
  const int g = want < cap ? want : cap;
  return g >= 4 ? 4 : (g >= 2 ? 2 : 1);
}
struct P2Args {
  int TH, TW, tiles_x, tiles_y, ntiles, PH, PW;
  int ppb;       // patch pixel pitch (bytes)
  int prb;       // patch row pitch (bytes) = PW * ppb + row padding (p2_pick_rowpad)
  int wpitch;    // weight row pitch in LDS (16-byte units)
  int nsteps;    // K-steps (32 K each) = ceil(KH*KW*Cin / 32)
  int nsp;       // row length of the q-major offset table [4][nsp]: nsteps rounded up to 4, plus slack for the pipelined over-read
  int kg;        // K-steps per streamed weight group
  int off_w, off_p, off_stat;   // LDS byte offsets (offset table sits at 0)
  unsigned xbytes;              // bytes of the input view from its first channel to the end of the last image (descriptor range, < 2^31)
};

// F8 = 1: fp8 mode.  Global activations stay bf16 (same HBM bytes); the patch is quantised to fp8 (e4m3, or e5m2 for a
// gradient input) with a per-tensor scale as it is written to LDS, the weights arrive pre-quantised (e4m3), and the K loop
// runs v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales: 128 K per instruction at twice the bf16 MFMA rate, and half
// the LDS bytes per K (the K loop of this kernel is LDS-bandwidth-bound for small register tiles).  A K-step is then 128 K =
// four 32-channel pieces (one per lane quarter); Cin % 32 == 0.  The fp32 accumulators are scaled back in the epilogue.
// RED = 1: dgrad launches that also take the BN-backward sums of the producers whose gradient they complete (conv_epi.h, BnRedSeg)
// The kernel body takes its workgroup index and grid size as arguments (vbx of vgx): conv_p2_kernel passes blockIdx.x / gridDim.x,
// conv_p2_group_kernel (below) the workgroup's index inside the problem its blockIdx.x range belongs to.
template <int MR, int NR, int WRES, int NPU, int NT, int F8, int RED = 0>
__device__ __forceinline__ void conv_p2_body(const ConvArgs& a, const P2Args& g, const int* __restrict__ tab, const int vbx, const int vgx) {
  typedef bf16_t T;
  constexpr int WES = F8 ? 1 : 2;             // bytes per weight element
  constexpr int UPS = F8 ? 8 : 4;             // 16-byte LDS units per weight row and K-step
  if (P2_DBG(64)) return;                     // ablation: launch + dispatch cost only
  constexpr int BN = NR * 16;
  constexpr int NWV = NT / 64;
#ifdef YS_P2_TIMELINE
  // s_memtime stamps of wave 0 / lane 0 of every 37th workgroup: [slot 0] entry, [1] prologue issued, then per tile
  // (top, patch in LDS, MFMA loop done, epilogue done), last = exit.  64 slots per recorded workgroup.
  int tl_n = 0;
  unsigned long long* tl_p = (a.tl && (vbx % 37) == 0 && blockIdx.y == 0 && threadIdx.x == 0) ? a.tl + (vbx / 37) * 64 : nullptr;
#define TL_STAMP() do { if (tl_p && tl_n < 63) tl_p[1 + tl_n++] = __builtin_readcyclecounter(); } while (0)
#ifdef YS_P2_TIMELINE_FINE
#define TL_STAMP2() TL_STAMP()
#else
#define TL_STAMP2() ((void)0)
#endif
#else
#define TL_STAMP() ((void)0)
#define TL_STAMP2() ((void)0)
#endif
  TL_STAMP();
  constexpr int NWU = WRES ? 1 : (BN * P2_KG * UPS + NT - 1) / NT;   // streamed weight units per thread
  YS_DYN_LDS(lds);
  char* lb = (char*)lds;
  int* sOff = (int*)lb;                       // [nsteps][4]
  uint4* sW = (uint4*)(lb + g.off_w);         // WRES: [BN][wpitch]; else [2][BN][wpitch]
  char* sPb = lb + g.off_p;                   // [PH*PW][ppb]
  float* sStat = (float*)(lb + g.off_stat);   // [NWV][BN][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.y * BN;
  const char* xb = (const char*)a.x;
  const char* wb = (const char*)(F8 ? a.w8 : a.w);
  const int taps = a.KH * a.KW;
  const int Ktot = taps * a.Cin;
  const int cu = a.Cin >> 3;                  // 16-byte units per patch pixel in GLOBAL memory (bf16)
  const float qs = F8 ? a.qscale[0] : 1.0f;   // quantisation multiplier of the input tensor
  float amx = 0.f;                            // fp8: running amax(|input|) of what this thread stages (next step's scale)
  const int npatch = g.PH * g.PW * cu;

  // ---- tile-independent tables, computed once per layer geometry on the host (p2_tables): per-(quarter, K-step) patch
  // offsets (q-major so that a lane fetches the offsets of consecutive K-steps with one LDS read), per-thread pixel offsets,
  // per-thread patch unit descriptors
  // the offset table goes global -> LDS by DMA as well (nsp 16-byte units, <= 2 requests): the copy through registers made every
  // workgroup wait out one L2 round trip before it could even request its patch
  {
    const ys_rsrc_t rsT = ys_make_rsrc(tab, (unsigned)g.nsp * 16u);
    if (tid < g.nsp) ys_bufld_lds16(rsT, (unsigned)tid * 16u, 0u, lb + wave * 1024);
  }
  const int* tpx = tab + g.nsp * 4;
  int pixbase[MR], pty[MR], ptx[MR], prow[MR];
#pragma unroll
  for (int mf = 0; mf < MR; mf++) {
    pixbase[mf] = tpx[(mf * 4 + 0) * NT + tid];
    pty[mf] = tpx[(mf * 4 + 1) * NT + tid];
    ptx[mf] = tpx[(mf * 4 + 2) * NT + tid];
    prow[mf] = tpx[(mf * 4 + 3) * NT + tid];    // output row of the lane's pixel relative to the tile's first pixel (epilogue)
  }
  constexpr int KG = P2_KG;                    // K-steps per streamed weight group
  constexpr int GU = KG * UPS;                 // 16-byte units per weight row and group
  constexpr bool TIGHT = NT == 256 && NPU <= 6 && MR * NR <= 8 && !F8;                    // 168-register variants
  constexpr int G0 = F8 ? 1 : p2_reg_group(MR, NR, TIGHT, !WRES && RED != 0 && YS_P2_LATE_PFETCH != 0 && !YS_P2_RING3);   // fp8: one K-step is already 128 K (32-byte fragments)
  // streamed bf16 weights by DMA leave the dgrad variants (no bias / BatchNorm / residual epilogue paths) room for both K-steps of a weight
  // group per LDS wait: with two waves per SIMD the one-step form is a chain of two LDS round trips per 10-16 MFMAs
  constexpr bool DMAW_G2 = !WRES && !F8 && NT == 256 && !YS_P2_RING3 && YS_P2_DMA_RING != 0 && RED != 0 && MR * NR <= 12 && YS_P2_DMA_G2 != 0;   // (4 x 4: spills)
  constexpr int G = DMAW_G2 ? KG : ((WRES || G0 < KG) ? G0 : KG);   // K-steps per register group of the K loop
  const int ngroups = WRES ? 1 : (g.nsteps + KG - 1) / KG;
  // RING3 (round 4): streamed weights are fetched THREE groups ahead (three register sets in rotation, two LDS slots as before).  One
  // group ahead -- the rounds 1-3 form: fetch at the top of a group, store at its end -- gives the L2 round trip one group's MFMAs to
  // hide behind (20-32 MFMAs = 320-512 cycles against ~1.5 thousand): s_memtime stamps of the 80 -> 80 and 64 -> 144 3x3 layers of the
  // Detect towers put their K loop at 21-24 thousand cycles per tile for 3.7 thousand cycles of MFMA.  The next tile's patch is then
  // requested AFTER the K loop (its registers would not fit next to three weight sets) and lands under the epilogue.
  constexpr bool RING3 = !WRES && YS_P2_RING3 != 0;
  // LATE (round 4): streamed-weight variants request the next tile's patch after the K loop (it lands under the epilogue) -- the 48
  // registers of a 12-unit patch are then free during the K loop, which pays for two K-steps of fragments per LDS wait (YS_P2_G1_MIN)
  // (dgrad variants only: the forward variants' epilogue -- bias, eval BatchNorm, residual paths -- leaves no room and spilled the patch)
  constexpr bool LATE = RING3 || (!WRES && !F8 && RED != 0 && YS_P2_LATE_PFETCH != 0);
  // DMAW (round 4): streamed bf16 weights by LDS DMA.  A weight group (KG = 2 K-steps = 64 K = 128 bytes per output channel) is BN rows of
  // eight 16-byte units; a wave instruction moves eight rows (1 KB), lane l -> row l / 8, LDS unit l % 8, which holds the row's logical unit
  // (l % 8) ^ ((row / 2) % 8) -- the blocked-GEMM kernel's swizzle: the K loop's fragment reads (16 rows x 4 units) are conflict-free.  Three
  // slots: at group g the workgroup waits for its own requests of g (issued two groups earlier), meets at ONE barrier, requests g + 2 into
  // the slot group g - 1 was read from, and multiplies.  The register form costs 12 registers, three ds_write_b128 and -- the expensive
  // part -- a wait for loads issued only one group (~1 thousand cycles) earlier, per thread and group.
  constexpr bool DMAW = !WRES && !F8 && NT == 256 && !RING3 && YS_P2_DMA_RING != 0;
  static_assert(!DMAW || (KG == 2 && UPS == 4), "the DMA weight ring hard-codes 128-byte rows per group (two K-steps of four 16-byte units): offsets, slot stride and the host plan's wbytes");
  constexpr int NBP = BN / 8;                  // 1 KB requests per weight group
  constexpr int NPW = (NBP + NWV - 1) / NWV;   // ... per wave (the last round may be partial)
  uint4 rwA[NWU], rwB[RING3 ? NWU : 1], rwC[RING3 ? NWU : 1];
  // this thread's (row, unit-in-group) of the streamed weight tile never changes: keep the row's byte offset (32 bits, through a buffer
  // descriptor of the weight shadow: a unit past the row's real K, a row past Cout or an idle thread carries the out-of-range offset and
  // arrives as zeros -- no 64-bit row pointers to keep (they were spilled next to three register sets), no select after the load)
  const ys_rsrcv_t rsWv = ys_make_rsrcv(wb, (unsigned)((long)a.Cout * Ktot * WES));
  unsigned wro[NWU];
  int wun[NWU];
#pragma unroll
  for (int k = 0; k < NWU; k++) {
    const int idx = tid + NT * k;
    const int n = idx / GU;
    wun[k] = (idx < BN * GU && n0 + n < a.Cout) ? idx - n * GU : -1;
    wro[k] = (unsigned)((long)(n0 + (wun[k] >= 0 ? n : 0)) * Ktot * WES);
  }
  auto wfetch = [&](uint4 (&rw)[NWU], int grp) {   // global -> registers: weights of K-steps [grp*KG, grp*KG + KG); unconditional loads
#pragma unroll
    for (int k = 0; k < NWU; k++) {
      const int u = grp * GU + wun[k];
      const bool ok = (bool)((int)(wun[k] >= 0) & (int)(u * (16 / WES) < Ktot) & (int)!P2_DBG(16));
      rw[k] = ys_bufld16(rsWv, ok ? wro[k] + (unsigned)u * 16u : YS_BUF_OOB);
    }
  };
  auto wstore = [&](const uint4 (&rw)[NWU], int buf) {
#pragma unroll
    for (int k = 0; k < NWU; k++) {
      const int idx = tid + NT * k;
      if (idx < BN * GU) { const int n = idx / GU; sW[(buf * BN + n) * g.wpitch + (idx - n * GU)] = rw[k]; }
    }
  };

#ifdef YS_EMU_BUILD
  const int wvu = wave;
#else
  const int wvu = __builtin_amdgcn_readfirstlane(wave);
#endif
  const int nmine = DMAW ? (NBP - wvu + NWV - 1) / NWV : 0;     // requests of this wave per weight group
  const int kunits = Ktot >> 3;
  const ys_rsrc_t rsWd = ys_make_rsrc(wb, (unsigned)((long)a.Cout * Ktot * WES));
  unsigned dro[DMAW ? NPW : 1];
  int dun[DMAW ? NPW : 1];
  if constexpr (DMAW) {
#pragma unroll
    for (int j = 0; j < NPW; j++) {
      const int row = (wvu + NWV * j) * 8 + (lane >> 3);
      dun[j] = (lane & 7) ^ ((row >> 1) & 7);
      dro[j] = (row < BN && n0 + row < a.Cout) ? (unsigned)((long)(n0 + row) * Ktot * 2) + (unsigned)dun[j] * 16u : YS_BUF_OOB;
    }
  }
  auto wdma = [&](int grp, int slot) {           // weight group grp -> ring slot; units past the row's real K, rows past Cout: zeros
    if constexpr (DMAW) {
#pragma unroll
      for (int j = 0; j < NPW; j++)
        if (wvu + NWV * j < NBP) {
          const bool ok = (bool)((int)(grp * 8 + dun[j] < kunits) & (int)!P2_DBG(16));
          ys_bufld_lds16(rsWd, ok ? dro[j] : YS_BUF_OOB, (unsigned)grp * 128u, (char*)sW + slot * (BN * 128) + (wvu + NWV * j) * 1024);
        }
    }
  };
  const int wko0 = (q ^ ((li >> 1) & 3)) * 16 + ((((li >> 1) & 7) >> 2) << 6);   // ring slot: byte offset of (K-step 0, quarter q) in this lane's rows

  // ---- patch of a tile: global -> registers (all loads back-to-back); the NEXT tile's patch is in flight while the
  // current one is consumed, so every resident workgroup always has a whole patch outstanding (HBM needs ~40 KB per CU
  // in flight to reach its bandwidth).  Per patch unit a thread keeps two tile-independent words: the element offset from
  // the tile's patch origin in global memory, and (row, column, LDS slot) packed -- so a tile costs an add, two compares
  // and a shift per unit instead of divisions and multiplies.
  uint4 rp[NPU];
  unsigned pdesc[NPU];                      // patch row (9 bits) | patch column (10) | LDS offset / 16 (13)
  int goff[NPU];                            // (row * Win + column) * in_ldc + unit * 8
  {
    const int* tpd = tpx + MR * 4 * NT;
#pragma unroll
    for (int k = 0; k < NPU; k++) {
      pdesc[k] = (unsigned)tpd[(2 * k + 0) * NT + tid];
      goff[k] = tpd[(2 * k + 1) * NT + tid];
    }
  }
  // Tile order.  Workgroup i runs on XCD i % 8 (round-robin dispatch); when the grid is a multiple of 8 each XCD walks its own
  // contiguous eighth of the tile list, so the tiles resident on an XCD at any time are spatial neighbours and their shared
  // halo rows hit that XCD's L2.  Otherwise plain interleaving.  Tile coordinates advance incrementally (no per-tile divisions).
  const bool xcd_order = (vgx & 7) == 0 && !P2_DBG(32);
  const int t_per_xcd = (g.ntiles + 7) >> 3;
  const int t_step = xcd_order ? (vgx >> 3) : vgx;
  const int t_first = xcd_order ? (vbx & 7) * t_per_xcd + (vbx >> 3) : vbx;
  const int t_end = xcd_order ? (((vbx & 7) + 1) * t_per_xcd < g.ntiles ? ((vbx & 7) + 1) * t_per_xcd : g.ntiles) : g.ntiles;
  int stx, sty, sb;
  {
    const int G = t_step;
    stx = G % g.tiles_x; const int rem = G / g.tiles_x;
    sty = rem % g.tiles_y; sb = rem / g.tiles_y;
  }
  auto advance = [&](int& txi, int& tyi, int& b) {
    txi += stx; if (txi >= g.tiles_x) { txi -= g.tiles_x; tyi++; }
    tyi += sty; if (tyi >= g.tiles_y) { tyi -= g.tiles_y; b++; }
    b += sb;
  };
  // Every unit issues its load unconditionally (loads inside exec-masked branches make the compiler lose count of what is in
  // flight and fall back to s_waitcnt vmcnt(0) BEFORE the MFMA loop, i.e. no overlap).  The loads go through a buffer descriptor
  // of the input view: a unit that is unused or falls into the zero padding gets the out-of-range offset and the hardware
  // returns zeros -- one 32-bit add and one select per unit instead of a 64-bit address, a 64-bit select against a safe address
  // and a second select when the patch is written to LDS.
  const ys_rsrcv_t rsP = ys_make_rsrcv(xb + (long)a.in_coff * 2L, g.xbytes);
  auto pfetch = [&](int txi, int tyi, int b) -> unsigned {
    const int iy0 = tyi * g.TH * a.SA - a.PAD, ix0 = txi * g.TW * a.SA - a.PAD;
    const int toff = (int)((((long)b * a.in_bstride + (long)iy0 * a.Win + ix0) * a.in_ldc) * 2L);   // may be negative for a tile on the border: only valid units use it
    unsigned okm = 0;
#pragma unroll
    for (int k = 0; k < NPU; k++) {
      unsigned d = pdesc[k];
      int go = goff[k];
#ifndef YS_EMU_BUILD
      asm volatile("" : "+v"(d), "+v"(go));
#endif
      // branch-free validity (bitwise, not short-circuit): unit in use, row and column inside the image
      const unsigned iy = (unsigned)(iy0 + (int)(d >> 23)), ix = (unsigned)(ix0 + (int)((d >> 13) & 1023u));
      const bool ok = (bool)((int)(d != 0xffffffffu) & (int)(iy < (unsigned)a.Hin) & (int)(ix < (unsigned)a.Win) & (int)!P2_DBG(1));
      rp[k] = ys_bufld16(rsP, ok ? (unsigned)(toff + go * 2) : YS_BUF_OOB);
      okm |= (unsigned)ok << k;
    }
    return okm;
  };
  int txi, tyi, b;
  {
    int t = t_first;
    txi = t % g.tiles_x; t /= g.tiles_x;
    tyi = t % g.tiles_y; b = t / g.tiles_y;
  }
  int ntx = txi, nty = tyi, nb = b;
  unsigned okm_next = 0;
  TL_STAMP();                                            // tables requested
  if (t_first < t_end) okm_next = pfetch(txi, tyi, b);   // the first patch is in flight while the weights are staged
  TL_STAMP();                                            // first patch requested
  if constexpr (RING3) { wfetch(rwA, 0); wfetch(rwB, 1); wfetch(rwC, 2); }
  if (WRES) {
    // resident weights: rows padded with zeros to a multiple of 4 K-steps (the pipelined K loop runs whole register groups).
    // LDS DMA (buffer_load ... lds, 1 KB per wave instruction): all of a wave's requests are in flight at once and no VGPR is
    // touched.  The register form -- eight loads, then eight LDS stores, per round -- was 4-8 thousand cycles of every workgroup's
    // ~10 thousand cycle prologue (s_memtime stamps, round 3).  LDS slot j of the weight region is (row j / wpitch, unit j % wpitch);
    // units past the real K (zero padding of the row, the pitch's spare slots) and rows past Cout carry the out-of-range offset =
    // zeros.  The region is rounded up to whole 1 KB requests by the plan.
    const int per_row = ((g.nsteps + 3) & ~3) * UPS;
    const int nslots = BN * g.wpitch;
    const ys_rsrc_t rsW = ys_make_rsrc(wb, (unsigned)((long)a.Cout * Ktot * WES));
    int n = tid / g.wpitch, u = tid - n * g.wpitch;
    const int dn = NT / g.wpitch, du = NT - dn * g.wpitch;       // one round of the workgroup = NT slots further
    char* dst = (char*)sW + wave * 1024;
    for (int j = wave * 64; j < nslots; j += NT) {
      const bool ok = (bool)((int)(u < per_row) & (int)(u * (16 / WES) < Ktot) & (int)(n0 + n < a.Cout) & (int)!P2_DBG(16));
      ys_bufld_lds16(rsW, ok ? (unsigned)((long)(n0 + n) * Ktot * WES) + (unsigned)u * 16u : YS_BUF_OOB, 0u, dst);
      dst += NT * 16;
      n += dn; u += du;
      if (u >= g.wpitch) { u -= g.wpitch; n++; }
    }
  }
  TL_STAMP();
  if (P2_DBG(128)) return;                     // ablation: prologue only (tables, resident weights, first patch fetch)
  constexpr int P2_NS = YS_P2_EPI_DIRECT ? 4 * NR : 8;
  float st1[P2_NS], st2[P2_NS];                // BN statistics of this workgroup's tiles (per-lane column sums)
#pragma unroll
  for (int e = 0; e < P2_NS; e++) { st1[e] = 0.f; st2[e] = 0.f; }

  // Counted waits (round 4).  vmcnt counts loads AND stores on this part and retires them in issue order.  The tile loop used to open
  // with s_waitcnt vmcnt(0): besides the prefetched patch that also drained the previous tile's output stores, which a CU retires at
  // 7-10 bytes per clock -- about two thousand cycles for a 16 KB tile, every tile, with nothing of this workgroup overlapping it (the
  // "patch in LDS" phase of the round-3 stamps: 3.4-3.6 thousand cycles whatever the epilogue form).  The patch loads are issued BEFORE
  // the epilogue, so "at most NST operations outstanding", NST = the epilogue's store instructions per wave (a static count: the stores
  // are unconditional, masked lanes carry the out-of-range offset), means the patch has landed while the stores keep draining under the
  // LDS writes, the next prefetch and the K loop.  Anything else the epilogue issued (accumulate / residual / y loads) only makes the
  // number outstanding larger, i.e. the wait longer -- never too short.  The weight / table DMA (inline asm, invisible to the
  // compiler) is waited for once, in front of the loop.
  constexpr int NST = (YS_P2_EPI_DIRECT || !YS_P2_COUNTED_WAIT) ? 0 : p2_epi_stores(MR, NR);
  YS_WAIT_VM0();
  for (int tile = t_first; tile < t_end; tile += t_step) {
    const int oy0 = tyi * g.TH, ox0 = txi * g.TW;
    TL_STAMP();
    ys_wait_vm<NST>();                        // this tile's patch has landed (the previous tile's stores may still be in flight)
    TL_STAMP2();
    ys_barrier_lds();                         // previous tile's epilogue staging (patch region) and tables are settled
    TL_STAMP2();
    if (!WRES && !RING3 && !DMAW) wfetch(rwA, 0);
#pragma unroll
    for (int k = 0; k < NPU; k++) {
      unsigned d = pdesc[k];
#ifndef YS_EMU_BUILD
      asm volatile("" : "+v"(d));
#endif
      if (F8) {
        if (d != 0xffffffffu && !P2_DBG(8)) {
          uint2 v8; v8.x = 0u; v8.y = 0u;
          if ((okm_next >> k) & 1u) {
            float f[8];
            ys_unpack<T>(rp[k], f);
#pragma unroll
            for (int e = 0; e < 8; e++) { amx = fmaxf(amx, fabsf(f[e])); f[e] *= qs; }
            v8 = a.f8 == 2 ? ys_pack_f8x8<1>(f) : ys_pack_f8x8<0>(f);
          }
          *(uint2*)(sPb + ((d & 8191u) << 3)) = v8;
        }
      } else {
        if (d != 0xffffffffu && !P2_DBG(8)) *(uint4*)(sPb + ((d & 8191u) << 4)) = rp[k];     // padding units arrived as zeros
      }
    }
    if (!WRES && !DMAW) wstore(rwA, 0);
    ys_barrier_lds();
    TL_STAMP();
    advance(ntx, nty, nb);
    // everything older (the patch just consumed, the previous epilogue's conditional loads / stores) has already been waited
    // for above; saying so explicitly resets the compiler's "may still be in flight" state for the accumulator registers
    ys_wait_vm<NST>();                        // (without it -- the epilogue's stores are unconditional now -- the class is 3 % slower: 5.10 -> 5.25 ms)
    const bool more = tile + t_step < t_end;
    if constexpr (DMAW) { wdma(0, 0); if (ngroups > 1) wdma(1, 1); }   // BEFORE the patch requests: the first groups' waits then leave the patch in flight
    if (!LATE && more) okm_next = pfetch(ntx, nty, nb);

    // resident weights: the accumulators start as the first K-step's products (MFMA with a zero C operand -- an inline constant, no
    // registers cleared: 4 * MR * NR v_mov per tile in a kernel whose busiest pipe is the VALU); streamed weights enter the K loop
    // once per weight group and keep the cleared accumulators
    f32x4 acc[MR][NR];
    if (!WRES || F8) {                        // (fp8 variants keep the cleared accumulators: peeling their 128-deep first group costs them registers)
#pragma unroll
      for (int mf = 0; mf < MR; mf++)
#pragma unroll
        for (int nf = 0; nf < NR; nf++) acc[mf][nf] = f32x4_zero();
    }

    // ---- K loop in register groups of G K-steps.  The one-step-at-a-time form (table read -> wait -> operand reads -> wait ->
    // MFMAs) cost ~350 cycles per K-step whatever the number of MFMAs in it (s_memtime stamps: 90-180 cycles per MFMA against 16
    // of issue time): two dependent LDS round trips per step with only 2-3 waves per SIMD to cover them.  Now one LDS wait serves
    // G steps (about 16 MFMAs), and the next group's table entries are fetched while the MFMAs issue, so the chain per group is
    // one round trip + the MFMAs -- which the SIMD's other waves overlap.  (Double-buffered fragment sets cost 50-100 registers
    // and spilled in most variants.)  Reads past the last real step land in valid LDS; their weights are zero.
    constexpr int FU = F8 ? 2 : 1;             // 16-byte reads per fragment
    struct Frags { uint4 w[G][NR][FU]; uint4 x[G][MR][FU]; };
    auto read_offs = [&](int (&off)[G], int s) {
      const int* pq = sOff + q * g.nsp + s;
      if (G == 4) { const uint4 v = *(const uint4*)pq; off[0] = (int)v.x; off[G > 1 ? 1 : 0] = (int)v.y; off[G > 2 ? 2 : 0] = (int)v.z; off[G > 3 ? 3 : 0] = (int)v.w; }
      else if (G == 2) { const uint2 v = *(const uint2*)pq; off[0] = (int)v.x; off[G > 1 ? 1 : 0] = (int)v.y; }
      else off[0] = pq[0];
    };
    // ng register groups starting at table step sbase, weight units from 0 in wbuf
    auto kloop = [&](const uint4* wbuf, int sbase, int ng, auto bf8_tag) {
      constexpr int B_BF8 = decltype(bf8_tag)::value;
      if (P2_DBG(2)) {                          // ablation build only: no K loop, defined accumulators
        if (WRES && !F8) {
#pragma unroll
          for (int mf = 0; mf < MR; mf++)
#pragma unroll
            for (int nf = 0; nf < NR; nf++) acc[mf][nf] = f32x4_zero();
        }
        return;
      }
      int off[G];
      read_offs(off, sbase);
      auto group = [&](const int gi, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value != 0;   // the tile's first group: C = 0 for its first K-step
        Frags f;
#pragma unroll
        for (int gs = 0; gs < G; gs++) {
#pragma unroll
          for (int nf = 0; nf < NR; nf++)
#pragma unroll
            for (int h = 0; h < FU; h++) {
              if constexpr (DMAW) f.w[gs][nf][h] = *(const uint4*)((const char*)wbuf + (nf * 16 + li) * 128 + (wko0 ^ (((gi * G + gs) & 1) << 6)));
              else f.w[gs][nf][h] = wbuf[(nf * 16 + li) * g.wpitch + ((gi * G + gs) * 4 + q) * FU + h];
            }
#pragma unroll
          for (int mf = 0; mf < MR; mf++)
#pragma unroll
            for (int h = 0; h < FU; h++) f.x[gs][mf][h] = *(const uint4*)(sPb + pixbase[mf] + off[gs] + h * 16);
        }
        read_offs(off, sbase + (gi + 1) * G);
        YS_SCHED_FENCE();                        // every read of the group is issued before its first MFMA
#pragma unroll
        for (int gs = 0; gs < G; gs++)
#pragma unroll
          for (int nf = 0; nf < NR; nf++)
#pragma unroll
            for (int mf = 0; mf < MR; mf++) {
              const f32x4 c0 = (FIRST && gs == 0) ? f32x4_zero() : acc[mf][nf];
              if (F8) acc[mf][nf] = mfma_scale_16x16x128_f8<B_BF8>(f.w[gs][nf][0], f.w[gs][nf][FU - 1], f.x[gs][mf][0], f.x[gs][mf][FU - 1], c0);
              else acc[mf][nf] = ys_mma<T>(f.w[gs][nf][0], f.x[gs][mf][0], c0);
            }
      };
      int gi0 = 0;
      if (WRES && !F8) { group(0, P2Tag1{}); gi0 = 1; }
#pragma unroll 1
      for (int gi = gi0; gi < ng; gi++) group(gi, P2Tag0{});
    };
    const bool in_bf8 = F8 && a.f8 == 2;      // uniform: the input operand is a gradient quantised to e5m2
    if (WRES) {
      if (in_bf8) kloop(sW, 0, (g.nsteps + G - 1) / G, P2Tag1{}); else kloop(sW, 0, (g.nsteps + G - 1) / G, P2Tag0{});
    } else if constexpr (RING3) {
      constexpr int NGS = KG / G;               // register groups per streamed weight slot
      // invariant at a group g = 0 (mod 3): rwA holds group g (already in LDS slot g & 1), rwB group g + 1, rwC group g + 2
      auto ring_step = [&](uint4 (&rf)[NWU], uint4 (&rs)[NWU], const int grp) {
        wfetch(rf, grp + 3);                    // unconditional (past the end: zeros); three groups of MFMAs to land
        if (in_bf8) kloop(sW + (grp & 1) * BN * g.wpitch, grp * KG, NGS, P2Tag1{}); else kloop(sW + (grp & 1) * BN * g.wpitch, grp * KG, NGS, P2Tag0{});
        wstore(rs, (grp + 1) & 1);
        ys_barrier_lds();
      };
      int grp = 0;
#pragma unroll 1
      for (; grp + 3 <= ngroups; grp += 3) { ring_step(rwA, rwB, grp); ring_step(rwB, rwC, grp + 1); ring_step(rwC, rwA, grp + 2); }
      if (grp < ngroups) { ring_step(rwA, rwB, grp); if (grp + 1 < ngroups) ring_step(rwB, rwC, grp + 1); }
      // the next tile: its patch and its first three weight groups are requested now and land under the epilogue
      if (tile + t_step < t_end) { wfetch(rwA, 0); wfetch(rwB, 1); wfetch(rwC, 2); }
    } else if constexpr (DMAW) {
      constexpr int NGS = KG / G;
      // Counted waits: vmcnt(N) with N = the requests of THIS wave known to be younger than group grp's -- group grp + 1's (issued one
      // iteration earlier) and, for the first two groups, the next tile's patch (NPU unconditional loads issued after the prologue's two
      // groups).  Anything else in flight (older stores, spills) only makes the wait stricter.
      int slot = 0;
#pragma unroll 1
      for (int grp = 0; grp < ngroups; grp++) {
        ys_wait_vm_dyn((grp + 1 < ngroups ? nmine : 0) + ((grp < 2 && more && !LATE) ? NPU : 0));
        if (grp < 3) TL_STAMP2();
        ys_barrier_lds();                       // group grp is in LDS for every wave; nobody still reads the slot of group grp - 1
        if (grp < 3) TL_STAMP2();
        int s2 = slot + 2; if (s2 >= 3) s2 -= 3;
        if (grp + 2 < ngroups) wdma(grp + 2, s2);
        if (in_bf8) kloop(sW + slot * (BN * 8), grp * KG, NGS, P2Tag1{}); else kloop(sW + slot * (BN * 8), grp * KG, NGS, P2Tag0{});
        if (grp < 3) TL_STAMP2();
        slot = slot == 2 ? 0 : slot + 1;
      }
      ys_barrier_lds();                         // every wave finished reading the patch: it becomes the staging area
    } else {
      constexpr int NGS = KG / G;               // register groups per streamed weight slot
#pragma unroll 1
      for (int grp = 0; grp < ngroups; grp++) {
        wfetch(rwA, grp + 1);                   // unconditional (past the end: zeros); lands during this group's MFMAs
        if (grp < 3) TL_STAMP2();               // (fine timeline: weight fetch issued / K-steps done / weights stored / barrier passed, first three groups)
        if (in_bf8) kloop(sW + (grp & 1) * BN * g.wpitch, grp * KG, NGS, P2Tag1{}); else kloop(sW + (grp & 1) * BN * g.wpitch, grp * KG, NGS, P2Tag0{});
        if (grp < 3) TL_STAMP2();
        wstore(rwA, (grp + 1) & 1);
        if (grp < 3) TL_STAMP2();
        ys_barrier_lds();
        if (grp < 3) TL_STAMP2();
      }
    }
    if (WRES) ys_barrier_lds();               // every wave finished reading the patch: it becomes the staging area
    TL_STAMP();

    // output rows: tile base (scalar) + the lane's table entry -- the 64-bit per-lane form cost ~1-1.5 thousand cycles per tile
    // (s_memtime stamps, round 3); every row index / byte offset of a P2 launch fits 31 bits (conv_p2_plan)
    const int rh = a.out_rh ? a.out_rh : a.Wout, rw = a.out_rh ? a.out_rw : 1;
    const int tbase = b * (int)a.out_bstride + oy0 * rh + ox0 * rw + (int)a.out_r0;
    int orow[MR];
    bool pv[MR];
#pragma unroll
    for (int mf = 0; mf < MR; mf++) {
      pv[mf] = (bool)((int)(pty[mf] < g.TH) & (int)(oy0 + pty[mf] < a.Hout) & (int)(ox0 + ptx[mf] < a.Wout));
      orow[mf] = tbase + prow[mf];
    }
    char* stg = sPb + wave * (16 * MR * (BN + 8) * 2 + 16 * MR * 16);
    if (F8) {                                 // back to real units: 1 / (input scale * weight scale)
      const float dq = a.deq[0];
#pragma unroll
      for (int mf = 0; mf < MR; mf++)
#pragma unroll
        for (int nf = 0; nf < NR; nf++)
#pragma unroll
          for (int r = 0; r < 4; r++) acc[mf][nf][r] *= dq;
    }
#if YS_P2_EPI_DIRECT
    (void)stg;
#ifdef YS_P2_TIMELINE
    if (!P2_DBG(4)) p2_epilogue_direct<MR, NR, RED>(a, acc, orow, pv, n0, st1, st2, [&]() { TL_STAMP(); });
#else
    if (!P2_DBG(4)) p2_epilogue_direct<MR, NR, RED>(a, acc, orow, pv, n0, st1, st2);
#endif
#else
    // LATE: the next tile's patch is requested inside the epilogue, right after the accumulators have been staged (their registers are
    // free then; requested straight after the K loop the forward variants spilled five patch units to scratch -- each spill a wait
    // for its own load, five exposed HBM round trips per tile)
    auto late_fetch = [&]() { if (LATE && tile + t_step < t_end) okm_next = pfetch(ntx, nty, nb); };
#ifdef YS_P2_TIMELINE
    if (!P2_DBG(4)) p2_epilogue<MR, NR, RED, YS_EPI_BATCH_P2(NPU)>(a, acc, orow, pv, n0, stg, st1, st2, [&]() { TL_STAMP(); }, late_fetch);
#else
    if (!P2_DBG(4)) p2_epilogue<MR, NR, RED, YS_EPI_BATCH_P2(NPU)>(a, acc, orow, pv, n0, stg, st1, st2, YsNoStamp(), late_fetch);
#endif
#endif
    TL_STAMP();
    txi = ntx; tyi = nty; b = nb;
  }
  YS_WAIT_VM0();                               // a workgroup without tiles still has its table / weight DMA in flight: it must land before the LDS is released
#if YS_P2_EPI_DIRECT
  if (RED ? a.nred > 0 : a.stats != nullptr) p2_stats_flush_direct<NR, NWV, 1>(a, n0, st1, st2, (float*)sPb, (long)vbx);
#else
  if (RED ? a.nred > 0 : a.stats != nullptr) p2_stats_flush<NR, NWV>(a, n0, st1, st2, (float*)sPb, (long)vbx);
#endif
  if (F8 && a.amax && blockIdx.y == 0) ys_amax_update(a.amax, amx);
  TL_STAMP();
#ifdef YS_P2_TIMELINE
  if (tl_p) tl_p[0] = (unsigned long long)tl_n;
#endif
}

template <int MR, int NR, int WRES, int NPU, int NT, int F8, int RED = 0>
__global__ void __launch_bounds__(NT, (NT == 512 ? 4 : (NPU <= 6 && MR * NR <= 8 && !F8 ? 3 : 2)))   // TIGHT variants: 3 waves / SIMD
conv_p2_kernel(ConvArgs a, P2Args g, const int* __restrict__ tab) {
  conv_p2_body<MR, NR, WRES, NPU, NT, F8, RED>(a, g, tab, (int)blockIdx.x, (int)gridDim.x);
}

// ---- grouped (multi-problem) launch.  Several INDEPENDENT convolutions that share a kernel variant -- the tower layers of the three
// pyramid levels of Detect / Segment (Head.cs:81-82: the same module applied to x[0], x[1], x[2] with per-level weights) -- run as ONE
// persistent grid: workgroups [end[i-1], end[i]) belong to problem i and see themselves as workgroup vbx of a grid of end[i] - end[i-1].
// The P4 / P5 launches of a YOLOv8n step are 100-400 tiles each, i.e. one latency-bound round of workgroups (12-27 us against a byte
// floor of 1-4 us); inside the P3 level's grid they are just more tiles.  Side streams were measured twice (-5 %: co-running persistent
// grids take each other's workgroup slots); one grid with a static split does not have that problem.  Range starts and sizes are
// multiples of 8 whenever a problem has >= 8 workgroups, so workgroup vbx still runs on XCD vbx % 8 (the tile order relies on it).
struct P2Prob { ConvArgs a; P2Args g; const int* tab; };
struct P2Group { int n; int end[YS_GROUP_MAX]; P2Prob p[YS_GROUP_MAX]; };
template <int MR, int NR, int WRES, int NPU, int NT, int RED>
__global__ void __launch_bounds__(NT, (NT == 512 ? 4 : (NPU <= 6 && MR * NR <= 8 ? 3 : 2)))
conv_p2_group_kernel(P2Group grp) {
  const int bx = (int)blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int k = 0; k + 1 < YS_GROUP_MAX; k++) pi += (int)(k + 1 < grp.n && bx >= grp.end[k]);
  const int start = pi ? grp.end[pi - 1] : 0;
  const P2Prob& pr = grp.p[pi];
#if YS_P2_GROUP_COPY
  // The problem's arguments are copied into registers once (round 4).  Read in place -- a dynamically indexed slot of the kernel-argument
  // segment -- hipcc re-loads fields where they are used: 81-109 s_load instructions per variant against 36-44 in conv_p2_kernel, many
  // of them inside the K loop, and every one is followed by s_waitcnt lgkmcnt(0), which drains the wave's LDS reads as well (the
  // counter is shared): per-launch records put the grouped 80 -> 80 tower layer at 228 us for 1.31x the pixels of a 113 us single launch.
  const ConvArgs a = pr.a;
  const P2Args g = pr.g;
  const int* const tab = pr.tab;
  conv_p2_body<MR, NR, WRES, NPU, NT, 0, RED>(a, g, tab, bx - start, grp.end[pi] - start);
#else
  conv_p2_body<MR, NR, WRES, NPU, NT, 0, RED>(pr.a, pr.g, pr.tab, bx - start, grp.end[pi] - start);
#endif
}


// ---- LDS bank model of the K loop's ds_read_b128 fragment reads (MI355X: a wave's b128 read is served in four fixed 16-lane groups
// over 64 four-byte banks, i.e. 16 slots of 16 bytes; lanes of a group that hit the same slot at different addresses serialise).
// Lane (li, q) of a patch fragment reads pixel(li) * ppb + q * 16 (+ a tap / channel offset common to the wave), and each group holds
// eight lanes of quarter q and eight of quarter q + 1.  With an ODD number of slots per pixel (the round-1/2 rule) the eight pixels of
// one quarter and the eight of the next always collide pairwise: every fragment read cost 8 LDS cycles instead of 4 (measured:
// SQ_LDS_BANK_CONFLICT = 44 % of LDS-active cycles).  A pitch of 2 (mod 4) slots puts the pixels of a quarter on even slots and the
// next quarter's on odd ones -> conflict-free for 16 consecutive pixels; a tile row shorter than 16 pixels needs the row pitch
// adjusted as well, which p2_pick_rowpad searches with this model.  Returns the mean LDS cycles per fragment read (4 = conflict-free).
static double p2_read_cycles(int cin, int kh, int kw, int sa, int th, int tw, int mr, int nwv, int ppb, int prb) {
  static const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
  const int ktot = kh * kw * cin;
  const int nsteps = cin >= 16 ? 1 : (ktot + 31) / 32;   // cin >= 16: the quarters of a lane group share a tap, so every K-step adds one constant to all lanes -> same pattern
  long tot = 0, n = 0;
  for (int wave = 0; wave < nwv; wave++)
    for (int mf = 0; mf < mr; mf++) {
      int pb[16];
      for (int li = 0; li < 16; li++) {
        const int px = wave * mr * 16 + mf * 16 + li;
        int ty = px / tw, tx = px - ty * tw;
        if (ty >= th) { ty = 0; tx = 0; }
        pb[li] = ty * sa * prb + tx * sa * ppb;
      }
      for (int st = 0; st < nsteps; st++) {
        int qo[4];
        for (int q = 0; q < 4; q++) {
          int k0 = st * 32 + q * 8; if (k0 >= ktot) k0 = 0;
          const int tap = k0 / cin, ch = k0 - tap * cin;
          const int a = tap / kw, b = tap - a * kw;
          qo[q] = a * prb + b * ppb + ch * 2;
        }
        for (int half = 0; half < 2; half++)
          for (int g = 0; g < 2; g++) {
            int addr[16], mx = 1;
            for (int i = 0; i < 16; i++) { const int l = grp[g][i] + half * 32; addr[i] = pb[l & 15] + qo[l >> 4]; }
            for (int i = 0; i < 16; i++) {
              int c = 1;
              for (int j = 0; j < i; j++) if (addr[j] == addr[i]) { c = 0; break; }       // identical addresses broadcast
              if (!c) continue;
              for (int j = i + 1; j < 16; j++) {
                if (((addr[j] >> 4) & 15) != ((addr[i] >> 4) & 15) || addr[j] == addr[i]) continue;
                bool dup = false;
                for (int k = i + 1; k < j; k++) if (addr[k] == addr[j]) { dup = true; break; }
                if (!dup) c++;
              }
              if (c > mx) mx = c;
            }
            tot += mx;
          }
        n++;
      }
    }
  return n ? (double)tot / (double)n : 4.0;
}
// pixel pitch of the bf16 patch: stride 1 -> smallest slot count >= Cin / 8 that is 2 (mod 4); stride 2 (lanes two pixels apart) -> odd.
// (Round 3 measured the odd rule everywhere as the slower alternative.)
static int p2_pixel_pitch(int cin, int sa) {
  const int cu = cin / 8;
  if (sa != 1) return cin * 2 + ((cu & 1) ? 32 : 16);
  int p = cu;
  while ((p & 3) != 2) p++;
  return p * 16;
}
// row padding (in 16-byte slots, 0..15) of the patch that minimises the modelled read cycles of the chosen tile within `room` bytes
static int p2_pick_rowpad(int cin, int kh, int kw, int sa, int th, int tw, int mr, int nwv, int ppb, int pw, int ph, size_t room) {
  static std::map<std::vector<int>, int> cache;
  static std::mutex mu;
  const std::vector<int> key = {cin, kh, kw, sa, th, tw, mr, nwv, ppb, pw, ph, (int)(room / (16 * (size_t)(ph > 0 ? ph : 1)) > 15 ? 15 : (int)(room / (16 * (size_t)(ph > 0 ? ph : 1))))};
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  double best = p2_read_cycles(cin, kh, kw, sa, th, tw, mr, nwv, ppb, pw * ppb);
  int pad = 0;
  for (int r = 1; r < 16 && best > 4.0 + 1e-9; r++) {
    if ((size_t)ph * r * 16 > room) break;
    const double c = p2_read_cycles(cin, kh, kw, sa, th, tw, mr, nwv, ppb, pw * ppb + r * 16);
    if (c < best - 1e-9) { best = c; pad = r; }
  }
  cache[key] = pad;
  return pad;
}

struct P2Plan { int ok, mr, nr, wres, npu, nt, gx, gy, per_cu; size_t lds; P2Args g; };
// fp8 variants whose 32-byte fragments fit the 256-register budget without spilling (hipcc -Rpass-analysis=kernel-resource-usage)
static bool p2_f8_tile_ok(int mr, int nr, int wres, int npu) {
  if (wres) return npu == 6 ? !(mr == 4 && nr == 4) : (mr * nr <= 8 && !(mr == 2 && nr == 5));
  return npu == 6 ? (mr * nr <= 6 || (mr == 4 && nr == 2)) && !(mr == 1 && nr == 5) : (mr * nr <= 4 && mr + nr <= 5);
}
// force_mr / force_npu (0 = free): grouped launches need every problem on the kernel variant of the group's largest problem
// plan sweep (tools/dev/r06/p2_sweep.py through ys_debug_p2_force): a register-tile height, a tile width and an output-channel split imposed on every plan of the
// process; 0 = the cost model's own choice.  Triage only -- nothing in the package sets them.
static int g_p2_force_mr = 0, g_p2_force_tw = 0, g_p2_force_nr = 0;
extern "C" __attribute__((visibility("default"))) int ys_debug_p2_force(int mr, int tw, int nr) { g_p2_force_mr = mr; g_p2_force_tw = tw; g_p2_force_nr = nr; return 0; }
static P2Plan conv_p2_plan(const ConvArgs& a, int force_mr = 0, int force_npu = 0, bool want_full = true, int force_nr = 0) {
  P2Plan p{};
  if (g_p2_force_mr && !force_mr) force_mr = g_p2_force_mr;
  if (g_p2_force_nr && !force_nr) force_nr = g_p2_force_nr;
  // 3x3 forward / stride-1 dgrad, 1x1 forward / dgrad, and the 1x1 .. 2x2 phase convolutions of a stride-2 dgrad (strided
  // output-row map)
  const bool k3 = a.KH == 3 && a.KW == 3 && a.out_rh == 0;
  const bool phase = a.KH >= 1 && a.KH <= 2 && a.KW >= 1 && a.KW <= 2 && a.SA == 1 && a.out_rh != 0 && a.PAD == 0;
  // 1x1 layers use the same kernel (patch = tile, no halo): measured 4 % faster per step than the direct-fragment kernel
  const bool k1 = a.KH == 1 && a.KW == 1 && a.PAD == 0 && a.SA == 1 && a.out_rh == 0;
  if (!((k3 || phase || k1) && a.DIVM == 0 && (a.SA == 1 || a.SA == 2) && a.pad_w_delta == 0)) return p;
  if (a.Cin % 8) return p;
  long xbytes_l;
  {
    const long pix = (long)(a.B - 1) * a.in_bstride + (long)a.Hin * a.Win;
    xbytes_l = (pix * a.in_ldc - a.in_coff) * 2L;
    if (xbytes_l <= 0 || xbytes_l >= (1L << 31)) return p;     // 32-bit descriptor offsets (larger views: the round-1 kernels)
    if ((long)a.B * a.out_bstride * a.out_ldc * 2L >= (1L << 31)) return p;                      // the epilogue's store offsets
  }
  const bool f8 = a.f8 != 0;
  // fp8 pays where the K loop dominates (LDS / MFMA bound layers); the HBM-bound small-channel layers gain nothing from it and pay
  // the on-the-fly quantisation (measured: 32-channel layers 1.6x slower in fp8) -> they keep the bf16 kernel
  const int f8_min_cin = (int)YS_OPT_INT("F8_MIN_CIN", 128);
  if (f8 && (a.Cin % 32 || a.Cin < f8_min_cin || !a.w8 || !a.qscale || !a.deq)) return p;
  const int ups = f8 ? 8 : 4;                        // 16-byte LDS units per weight row and K-step
  const int nfr = (a.Cout + 15) / 16;
  int nr = nfr <= 4 ? nfr : (nfr % 5 == 0 ? 5 : 4);
  const int cu = a.Cin / 8;
  P2Args g{};
  g.xbytes = (unsigned)xbytes_l;
  g.ppb = f8 ? a.Cin + (((a.Cin / 16) & 1) ? 32 : 16) : p2_pixel_pitch(a.Cin, a.SA);   // bf16: bank-model rule (p2_pixel_pitch); fp8: odd number of 16-byte slots per pixel
  const int taps = a.KH * a.KW;
  g.nsteps = f8 ? (taps * a.Cin + 127) / 128 : (taps * a.Cin + 31) / 32;
  // resident up to 40 KB (measured: 20 KB 14.52, 40 KB 14.48, 80 KB 14.96 ms/step -- larger resident sets cost the second
  // workgroup per CU)
  // rows padded to 4 K-steps; fp8: 52 KB (the streamed fp8 variants are limited to small register tiles; YOLOv8x 1280 step 165.3 -> 161.7 ms)
  const size_t wresmax = f8 ? 52 * 1024 : 44 * 1024;
  // Streamed weights cost one L2 round trip per K-group on the critical path of every tile.  When half the output channels
  // would make the weight set resident, split the channels over two workgroup columns instead (the patch is then read
  // twice, from L2).
  const double tileconst = 3000.0;
  const int nsteps4 = (g.nsteps + 3) & ~3;           // resident weight rows are zero-padded to whole register groups (<= 4 K-steps)
  // weight rows: 16 consecutive rows per fragment read, same lane-group structure as the patch -> a pitch of 2 (mod 4) slots is
  // conflict-free (the odd pitch was 2-way); fp8 rows (two reads per fragment, quarters two slots apart) keep the odd pitch
  auto wp = [&](int units) { if (f8) return units | 1; int p = units; while ((p & 3) != 2) p++; return p; };
  g.nsp = nsteps4 + 12;                              // + slack: the pipelined loop reads table entries up to two groups ahead
  if ((size_t)nr * 16 * wp(nsteps4 * ups) * 16 > wresmax && nr % 2 == 0 &&   // (measured 13.00 -> 12.86 ms/step)
      (size_t)(nr / 2) * 16 * wp(nsteps4 * ups) * 16 <= wresmax && nfr % (nr / 2) == 0)
    nr /= 2;
  if (f8 && (size_t)nr * 16 * wp(nsteps4 * ups) * 16 > wresmax && nr > 2) nr = 2;   // streamed fp8 weights: small register tiles only
  if (force_nr && force_nr <= nr && nfr % force_nr == 0) nr = force_nr;   // grouped launch: the output-channel split of the group's first member (never wider than this problem's own choice)
  const int bn = nr * 16;
  const size_t wres_bytes = (size_t)bn * wp(nsteps4 * ups) * 16;
  const int wres = wres_bytes <= wresmax ? 1 : 0;
  g.kg = wres ? g.nsteps : P2_KG;     // conv_p2_kernel::KG
  if (!wres && g.kg > g.nsteps) g.kg = g.nsteps;
  g.wpitch = wres ? wp(nsteps4 * ups) : wp(g.kg * ups);
  const bool dmaw = !wres && !f8 && YS_P2_DMA_RING != 0;   // conv_p2_body::DMAW (every streamed bf16 variant has 256 threads): three ring slots of bn 128-byte rows
  const size_t wbytes = wres ? (wres_bytes + 1023) / 1024 * 1024 : (dmaw ? (size_t)3 * bn * 128 : (size_t)2 * bn * g.wpitch * 16);   // resident set: whole 1 KB LDS-DMA requests
  const size_t tab = ((size_t)g.nsp * 16 + 15) / 16 * 16;
  const int gy = ys_cdiv(a.Cout, bn);
  // 3x3 layers with >= 256 input channels do not fit a useful whole-Cin patch (<= 64-pixel tiles, the full weight set streamed
  // per tile): the chunked round-1 kernel handles them better (YOLOv11m-seg step 65.9 -> 63.4 ms)
  const int maxcin3 = 255;
  if (k3 && a.SA == 1 && a.Cin > (f8 ? 640 : maxcin3)) return p;   // an fp8 patch pixel is half the bytes
  for (size_t budget = (wres && wres_bytes > 52 * 1024 ? 152 : 76) * 1024; budget <= 152 * 1024 && !p.ok; budget *= 2) {   // two workgroups per CU; one if nothing else fits
  // tile = (4 waves x 16*mr pixels, th x tw): minimise the bytes a layer moves through the CU (patch incl. halo, streamed
  // weights, output) plus a per-tile constant; among shapes that give the chip >= 512 workgroups when the layer is large
  // enough.  (512-thread workgroups -- 8 waves x 2 fragments, same LDS footprint -- were measured 13 % slower: the 128-register
  // budget of 4 waves/SIMD spills for >= 48 output channels.)
  double best = 1e30; bool best_full = false;
  for (int nt = 256; nt >= 256; nt >>= 1) {
    const int nwv = nt / 64;
    const size_t stat = (size_t)nwv * bn * 2 * 4;
    const int npu_max = nt == 512 ? 6 : P2_NPU;
    for (int mr = (nt == 512 ? 2 : (nr <= 4 ? 4 : 2)); mr >= 1; mr >>= 1) {
      if (force_mr && mr != force_mr) continue;
      const int npx = 16 * nwv * mr;
      const size_t stage = (size_t)nwv * (16 * mr * (bn + 8) * 2 + 16 * mr * 16);
      for (int tw = 1; tw <= npx && tw <= a.Wout; tw++) {
        if (g_p2_force_tw && tw != g_p2_force_tw) continue;
        int th = npx / tw; if (th > a.Hout) th = a.Hout;
        const int ph = (th - 1) * a.SA + a.KH, pw = (tw - 1) * a.SA + a.KW;
        size_t pbytes = (size_t)ph * pw * g.ppb; if (pbytes < stage) pbytes = stage;
        if (pbytes < (size_t)17 * nt * 4) pbytes = (size_t)17 * nt * 4;   // statistics scratch of p2_stats_flush ([16][NT] lane sums + [P][2 BN] partials)
        const size_t lds = tab + wbytes + pbytes + stat;
        if (f8 && !p2_f8_tile_ok(mr, nr, wres, ph * pw * cu <= 6 * 256 ? 6 : 12)) continue;
        if (lds > budget || ph * pw * cu > npu_max * nt || (size_t)ph * pw * g.ppb > (size_t)8192 * (f8 ? 8 : 16)) continue;   // 13-bit LDS slot field
        if (force_npu == 6 && ph * pw * cu > 6 * nt) continue;
        const int tx = ys_cdiv(a.Wout, tw), ty = ys_cdiv(a.Hout, th);
        const long ntiles = (long)tx * ty * a.B;
        const double per_tile = (double)ph * pw * a.Cin + (wres ? 0.0 : 0.5 * bn * (double)taps * a.Cin) + 1.0 * npx * (a.Cin + bn) + tileconst;
        // Round 6 (plan sweep on the MI355X, tools/dev/r06/p2_sweep.py): the byte model above prefers the largest tile, but a plan of the three-workgroups-per-CU class
        // (6-unit patch, register tile of <= 8 MFMAs, <= 50 KB of LDS: conv_p2_kernel's TIGHT variants) beats a cheaper-looking two-per-CU plan by more than the
        // bytes say -- 1x1 64 -> 64 at M = 409600: 4 x 4 tiles 37.0 us, 1 x 4 tiles 28.2 us; 3x3 s2 16 -> 32 at M = 1.6 M: 68.0 -> 59.0 us.  The class gets a discount.
        const bool cls3 = !f8 && nt == 256 && mr * nr <= 8 && ph * pw * cu <= 6 * nt && lds <= (size_t)50 * 1024;
        const double cost = (double)tx * ty * per_tile * (cls3 ? 0.75 : 1.0);   // (0.85 / 0.75 / 0.65 / 0.55 measured alike end to end: conv_p2 time 2.16 -> 2.11 ms per config-2 step, config 3 1.44 -> 1.40, config 4 3.27 -> 3.16)
        const bool full = !want_full || ntiles * gy >= 512;   // (a small member of a grouped launch does not have to fill the chip by itself: cheapest tiles)
        if ((full && !best_full) || (full == best_full && cost < best)) {
          best = cost; best_full = full;
          P2Args cur = g;
          cur.TH = th; cur.TW = tw; cur.tiles_x = tx; cur.tiles_y = ty; cur.PH = ph; cur.PW = pw; cur.ntiles = (int)ntiles;
          cur.off_w = (int)tab; cur.off_p = (int)(tab + wbytes); cur.off_stat = (int)(lds - stat);
          p.ok = 1; p.mr = mr; p.nr = nr; p.wres = wres; p.g = cur; p.lds = lds; p.gy = gy; p.nt = nt;
          p.npu = nt == 512 ? 6 : (ph * pw * cu <= 6 * 256 ? 6 : 12);
          if (force_npu) p.npu = force_npu;
        }
      }
    }
  }
  }
  if (!p.ok) return p;
  if ((long)p.g.ntiles > 2L * ys_cdiv(a.M, 64)) { p.ok = 0; return p; }   // stats workspace bound (model.hip stat_max)
  // three workgroups per CU (TIGHT register variants) when three footprints fit the 160 KB
  const size_t lds3 = (size_t)50 * 1024;
  p.g.prb = p.g.PW * p.g.ppb;
  if (!f8) {
    // row padding of the patch (bank model, p2_pick_rowpad) inside the occupancy class the tile search settled on
    const int nwv = p.nt / 64;
    const size_t stat = (size_t)nwv * bn * 2 * 4, stage = (size_t)nwv * (16 * p.mr * (bn + 8) * 2 + 16 * p.mr * 16);
    size_t floor_b = stage > (size_t)17 * p.nt * 4 ? stage : (size_t)17 * p.nt * 4;
    const size_t pb0 = (size_t)p.g.PH * p.g.prb;
    const size_t cap = (p.lds <= lds3 && p.npu == 6 && p.mr * p.nr <= 8) ? lds3 : (p.lds <= 76 * 1024 ? 76 * 1024 : 152 * 1024);
    const size_t base = p.lds - (pb0 > floor_b ? pb0 : floor_b);           // tables + weights + statistics
    size_t room = cap > base + pb0 ? cap - base - pb0 : 0;
    if (pb0 + room > (size_t)8192 * 16) room = (size_t)8192 * 16 > pb0 ? (size_t)8192 * 16 - pb0 : 0;   // 13-bit LDS slot field
    const int pad = p2_pick_rowpad(a.Cin, a.KH, a.KW, a.SA, p.g.TH, p.g.TW, p.mr, nwv, p.g.ppb, p.g.PW, p.g.PH, room);
    p.g.prb += pad * 16;
    const size_t pb1 = (size_t)p.g.PH * p.g.prb;
    p.lds = base + (pb1 > floor_b ? pb1 : floor_b);
    p.g.off_stat = (int)(p.lds - stat);
  }
  // three per CU needs the TIGHT register variant too (conv_p2_kernel's launch bounds: NPU <= 6, MR * NR <= 8, bf16 -- 168 registers): the
  // 4 x 3, 2 x 5 and 4 x 4 tiles compile to 189-236 registers, i.e. two waves per SIMD, and a 768-workgroup grid of theirs left 256
  // workgroups waiting for a slot (round 4: the same pathology as the grouped launches' rounding)
  const bool tight_regs = p.npu == 6 && p.mr * p.nr <= 8 && !f8;
  const int per_cu = (p.lds <= lds3 && tight_regs) ? 3 : (p.lds <= 76 * 1024 ? 2 : 1);
  p.per_cu = per_cu;
  // (measured, round 4: a 168-register variant that LDS limits to two workgroups per CU anyway is NOT better off on the 12-unit variant of the
  // same tile with 2-4 K-steps per LDS wait: config 2 patch-kernel time 2.58 -> 2.65 ms, config 5 6.8 -> 6.9 -- the K loop of these tiles is
  // bound by LDS read bandwidth / issue, not by the exposed wait)
  long gx = ((long)ys_cu_count() * per_cu) / p.gy;            // persistent grid: the next tile's patch is prefetched
  if (gx > p.g.ntiles) gx = p.g.ntiles;
  if (gx < 1) gx = 1;
  p.gx = (int)gx;
  return p;
}

// Tile-independent index tables of a P2 launch (see conv_p2_kernel), built on the host once per layer geometry and cached
// on the device for the life of the process (a few KB per distinct layer shape).
static const int* p2_tables(const ConvArgs& a, const P2Plan& p) {
  static std::map<std::vector<int>, int*> cache;
  static std::mutex cache_mu;                   // distinct ys_ctx may launch from different threads
  std::lock_guard<std::mutex> lock(cache_mu);
  int dev = 0;
  hipGetDevice(&dev);
  const P2Args& g = p.g;
  const bool f8 = a.f8 != 0;
  const int rh = a.out_rh ? a.out_rh : a.Wout, rw = a.out_rh ? a.out_rw : 1;
  std::vector<int> key = {dev, a.Cin, a.KH, a.KW, a.SA, a.Win, a.in_ldc, g.TH, g.TW, g.PH, g.PW, g.ppb, g.prb, g.nsteps, g.nsp, p.mr, p.npu, p.nt, (int)f8, rh, rw};
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  const int NT = p.nt;
  const int cu = a.Cin / 8, Ktot = a.KH * a.KW * a.Cin, npatch = g.PH * g.PW * cu;
  std::vector<int> h((size_t)g.nsp * 4 + (size_t)p.mr * 4 * NT + (size_t)p.npu * 2 * NT, 0);
  for (int qq = 0; qq < 4; qq++)            // q-major: [quarter][K-step]; steps past the last real one keep offset 0 (their weights are zero)
    for (int st = 0; st < g.nsteps; st++) {
      const int kp = f8 ? 32 : 8;             // channels per (K-step, quarter) piece
      const int k0 = st * 4 * kp + qq * kp;
      if (k0 < Ktot) {
        const int tap = k0 / a.Cin, ch = k0 - tap * a.Cin;
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        h[(size_t)qq * g.nsp + st] = kh * g.prb + kw * g.ppb + ch * (f8 ? 1 : 2);
      }
    }
  int* tpx = h.data() + g.nsp * 4;
  for (int tid = 0; tid < NT; tid++) {
    const int wave = tid >> 6, li = tid & 15;
    for (int mf = 0; mf < p.mr; mf++) {
      const int px = wave * (p.mr * 16) + mf * 16 + li;
      int ty = px / g.TW, tx = px - ty * g.TW;
      if (ty >= g.TH) { ty = g.TH; tx = 0; }    // idle lane of a ragged tile: marked by ty == TH, reads pixel (0,0)
      tpx[(mf * 4 + 0) * NT + tid] = ty < g.TH ? (ty * a.SA) * g.prb + (tx * a.SA) * g.ppb : 0;
      tpx[(mf * 4 + 1) * NT + tid] = ty;
      tpx[(mf * 4 + 2) * NT + tid] = tx;
      tpx[(mf * 4 + 3) * NT + tid] = ty < g.TH ? ty * rh + tx * rw : 0;
    }
  }
  int* tpd = tpx + p.mr * 4 * NT;
  for (int tid = 0; tid < NT; tid++)
    for (int k = 0; k < p.npu; k++) {
      const int idx = tid + NT * k;
      unsigned d = 0xffffffffu; int go = 0;
      if (idx < npatch) {
        const int pix = idx / cu, u = idx - pix * cu;
        const int r = pix / g.PW, cc = pix - r * g.PW;
        d = ((unsigned)r << 23) | ((unsigned)cc << 13) | (f8 ? (unsigned)((r * g.prb + cc * g.ppb + u * 8) >> 3) : (unsigned)((r * g.prb + cc * g.ppb + u * 16) >> 4));
        go = (r * a.Win + cc) * a.in_ldc + u * 8;
      }
      tpd[(2 * k + 0) * NT + tid] = (int)d;
      tpd[(2 * k + 1) * NT + tid] = go;
    }
  int* dptr = nullptr;
  if (hipMalloc(&dptr, h.size() * sizeof(int)) != hipSuccess) return nullptr;
  if (hipMemcpy(dptr, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { hipFree(dptr); return nullptr; }
  cache[key] = dptr;
  return dptr;
}

template <int MR, int NR, int WRES, int NPU, int NT, int F8, int RED = 0>
static int conv_p2_launch_t(hipStream_t st, ConvArgs a, const P2Plan& p) {
#ifdef YS_P2_ABLATE
  a.dbg = (int)YS_OPT_INT("DBG", 0);   // ablation switches (triage build only: build.py p2ablate)
#endif
  a.red_koff = (int)offsetof(ConvArgs, red);       // ConvArgs is the kernel's first argument (conv_epi.h ys_red_table)
  static std::atomic<unsigned> attr_done{0};      // per device: the attribute belongs to the device's code object
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  if (!(attr_done.load(std::memory_order_relaxed) & (1u << (dev_id & 31)))) {
    hipFuncSetAttribute((const void*)conv_p2_kernel<MR, NR, WRES, NPU, NT, F8, RED>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.fetch_or(1u << (dev_id & 31), std::memory_order_relaxed);
  }
  char lab[192] = "";
  if (ys_kprof_enabled()) snprintf(lab, sizeof(lab), F8 ? "p2f8 k%d s%d div1 cin%d cout%d M%d acc%d nt%d mr%d nr%d wres%d npu%d tile%dx%d grid%dx%d lds%d" : "p2 k%d s%d div1 cin%d cout%d M%d acc%d nt%d mr%d nr%d wres%d npu%d tile%dx%d grid%dx%d lds%d", a.KH * 10 + a.KW, a.SA, a.Cin, a.Cout, a.M, a.accumulate, NT, MR, NR, WRES, NPU, p.g.TH, p.g.TW, p.gx, p.gy, (int)p.lds);
  YsKprofScope prof(st, "conv_igemm", lab);
  const int* tab = p2_tables(a, p);
  if (!tab) { ys_set_error("conv p2: cannot allocate the index tables"); return YS_ERR_OOM; }
#ifdef YS_P2_TIMELINE
  static unsigned long long* tl_buf = nullptr;
  const char* tl_path = getenv("YS_P2_TL");
  if (tl_path) {
    if (!tl_buf) hipMalloc(&tl_buf, 64 * 64 * 8);
    hipMemsetAsync(tl_buf, 0, 64 * 64 * 8, st);
    a.tl = tl_buf;
  }
#endif
  YS_LAUNCH_LDS((conv_p2_kernel<MR, NR, WRES, NPU, NT, F8, RED>), dim3(p.gx, p.gy), NT, p.lds, st, a, p.g, tab);
#ifdef YS_P2_TIMELINE
  if (tl_path) {
    static unsigned long long h[64 * 64];
    hipStreamSynchronize(st);
    hipMemcpy(h, tl_buf, sizeof(h), hipMemcpyDeviceToHost);
    FILE* f = fopen(tl_path, "a");
    if (f) {
      fprintf(f, "# k%d%d s%d cin%d cout%d M%d mr%d nr%d wres%d npu%d tile%dx%d grid%dx%d lds%d ntiles%d nsteps%d\n", a.KH, a.KW, a.SA, a.Cin, a.Cout, a.M, MR, NR, WRES, NPU, p.g.TH, p.g.TW, p.gx, p.gy, (int)p.lds, p.g.ntiles, p.g.nsteps);
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
static int conv_p2_group_dispatch(hipStream_t st, const ConvArgs* a, const P2Plan* p, int n, const int* gxs, size_t lds);
static int conv_p2_dispatch(hipStream_t st, const ConvArgs& a, const P2Plan& p) {
  // (Single launches through conv_p2_group_kernel with one problem -- arguments read from the kernel-argument segment instead of held in scalar registers --
  // were built and measured in round 4 as no faster; the switch is gone.)
  {
#define P2F(M_, N_, F_, R_) { \
    if (p.wres) return p.npu == 6 ? conv_p2_launch_t<M_, N_, 1, 6, 256, F_, R_>(st, a, p) : conv_p2_launch_t<M_, N_, 1, 12, 256, F_, R_>(st, a, p); \
    return p.npu == 6 ? conv_p2_launch_t<M_, N_, 0, 6, 256, F_, R_>(st, a, p) : conv_p2_launch_t<M_, N_, 0, 12, 256, F_, R_>(st, a, p); }
#define P2(M_, N_) if (p.mr == M_ && p.nr == N_) { if (a.f8) P2F(M_, N_, 1, 0) else if (a.nred > 0 || (a.accumulate && YS_P2_EPI_DIRECT)) P2F(M_, N_, 0, 1) else P2F(M_, N_, 0, 0) }
    if (a.f8 && a.nred > 0) { ys_set_error("conv p2: the fused BN-backward reduction has no fp8 variant"); return YS_ERR_UNSUPPORTED; }
#ifdef YS_P2_ONE          // compile-time triage: a single register tile (seconds instead of minutes per resource-usage experiment)
#ifndef YS_P2_ONE_M
#define YS_P2_ONE_M 1
#define YS_P2_ONE_N 5
#endif
    P2(YS_P2_ONE_M, YS_P2_ONE_N)
#else
    P2(1, 1) P2(2, 1) P2(4, 1) P2(1, 2) P2(2, 2) P2(4, 2) P2(1, 3) P2(2, 3) P2(4, 3) P2(1, 4) P2(2, 4) P2(4, 4) P2(1, 5) P2(2, 5)
#endif
#undef P2
#undef P2F
  }
  ys_set_error("conv p2: no kernel for NT=%d MR=%d NR=%d", p.nt, p.mr, p.nr);
  return YS_ERR_UNSUPPORTED;
}


// ---- grouped launch of n <= YS_GROUP_MAX independent P2 convolutions on one kernel variant (conv_p2_group_kernel).  Returns YS_OK
// and the statistics rows (= workgroups) each problem got in rows[], or YS_ERR_UNSUPPORTED when the problems cannot share a variant
// (the caller then launches them one by one).  row_cap[i] > 0 bounds problem i's workgroups (partial-row regions sized elsewhere).
template <int MR, int NR, int WRES, int NPU, int NT, int RED>
static int conv_p2_group_launch_t(hipStream_t st, const ConvArgs* a, const P2Plan* p, int n, const int* gxs, size_t lds) {
#ifdef YS_P2_ABLATE
  const int dbg = (int)YS_OPT_INT("DBG", 0);
#else
  const int dbg = 0;
#endif
  static std::atomic<unsigned> attr_done{0};
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  if (!(attr_done.load(std::memory_order_relaxed) & (1u << (dev_id & 31)))) {
    hipFuncSetAttribute((const void*)conv_p2_group_kernel<MR, NR, WRES, NPU, NT, RED>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.fetch_or(1u << (dev_id & 31), std::memory_order_relaxed);
  }
  P2Group grp{};
  grp.n = n;
  int total = 0;
  char lab[320] = "";
  for (int i = 0; i < n; i++) {
    P2Prob& pr = grp.p[i];
    pr.a = a[i]; pr.a.dbg = dbg; pr.g = p[i].g;
    pr.a.red_koff = (int)(offsetof(P2Group, p) + (size_t)i * sizeof(P2Prob) + offsetof(P2Prob, a) + offsetof(ConvArgs, red));
    pr.tab = p2_tables(a[i], p[i]);
    if (!pr.tab) { ys_set_error("conv p2: cannot allocate the index tables"); return YS_ERR_OOM; }
    total += gxs[i];
    grp.end[i] = total;
  }
  for (int i = n; i < YS_GROUP_MAX; i++) grp.end[i] = total;
  if (ys_kprof_enabled()) {
    int o = snprintf(lab, sizeof(lab), "p2grp%d k%d s%d div1 cin%d cout%d M", n, a[0].KH * 10 + a[0].KW, a[0].SA, a[0].Cin, a[0].Cout);
    long Msum = 0;
    for (int i = 0; i < n; i++) Msum += a[i].M;
    o += snprintf(lab + o, sizeof(lab) - o, "%ld acc%d nt%d mr%d nr%d wres%d npu%d tile%dx%d grid%dx%d lds%d", Msum, a[0].accumulate, NT, MR, NR, WRES, NPU, p[0].g.TH, p[0].g.TW, total, p[0].gy, (int)lds);
    for (int i = 0; i < n && o < (int)sizeof(lab) - 24; i++) o += snprintf(lab + o, sizeof(lab) - o, " [%dx%d t%d g%d]", p[i].g.TH, p[i].g.TW, p[i].g.ntiles, gxs[i]);
  }
  YsKprofScope prof(st, "conv_igemm", lab);
  YS_LAUNCH_LDS((conv_p2_group_kernel<MR, NR, WRES, NPU, NT, RED>), dim3(total, p[0].gy), NT, lds, st, grp);
  return YS_OK;
}

// variant dispatch of a grouped launch (n >= 1 problems already planned on ONE variant; gxs = workgroups per problem)
static int conv_p2_group_dispatch(hipStream_t st, const ConvArgs* a, const P2Plan* p, int n, const int* gxs, size_t lds) {
  const bool red = a[0].nred > 0 || (a[0].accumulate && YS_P2_EPI_DIRECT);
#define P2GF(M_, N_, R_) { \
    if (p[0].wres) return p[0].npu == 6 ? conv_p2_group_launch_t<M_, N_, 1, 6, 256, R_>(st, a, p, n, gxs, lds) : conv_p2_group_launch_t<M_, N_, 1, 12, 256, R_>(st, a, p, n, gxs, lds); \
    return p[0].npu == 6 ? conv_p2_group_launch_t<M_, N_, 0, 6, 256, R_>(st, a, p, n, gxs, lds) : conv_p2_group_launch_t<M_, N_, 0, 12, 256, R_>(st, a, p, n, gxs, lds); }
#define P2G(M_, N_) if (p[0].mr == M_ && p[0].nr == N_) { if (red) P2GF(M_, N_, 1) else P2GF(M_, N_, 0) }
#ifdef YS_P2_ONE
  P2G(YS_P2_ONE_M, YS_P2_ONE_N)
#else
  P2G(1, 1) P2G(2, 1) P2G(4, 1) P2G(1, 2) P2G(2, 2) P2G(4, 2) P2G(1, 3) P2G(2, 3) P2G(4, 3) P2G(1, 4) P2G(2, 4) P2G(4, 4) P2G(1, 5) P2G(2, 5)
#endif
#undef P2G
#undef P2GF
  ys_set_error("conv p2 group: no kernel for MR=%d NR=%d", p[0].mr, p[0].nr);
  return YS_ERR_UNSUPPORTED;
}

int ys_conv_p2_group_launch(hipStream_t st, const ConvArgs* a, int n, const int* row_cap, int* rows, bool plan_only) {
  const bool off = YS_OPT_INT("GROUP", 1) == 0;
  if (off || n < 2 || n > YS_GROUP_MAX) return YS_ERR_UNSUPPORTED;
  P2Plan p[YS_GROUP_MAX];
  int big = 0;
  for (int i = 0; i < n; i++) {
    const ConvArgs& x = a[i];
    // (measured, round 4: the stride-2 dgrad phases of the 128- and 256-channel layers, which the blocked-GEMM kernel takes one by one, run as one
    // patch-kernel grid within noise of the four GEMM launches: 9.01-9.02 -> 8.92-9.00 ms/step; they stay on the GEMM kernel)
    if (x.f8 || ys_conv_gemm_rows(x)) return YS_ERR_UNSUPPORTED;
    if (ys_conv_dgrad_uses_phases(YS_BF16, x.KH, x.DIVM + 1) && x.KW == x.KH) return YS_ERR_UNSUPPORTED;
    if (x.Cin != a[0].Cin || x.Cout != a[0].Cout || x.SA != a[0].SA || x.PAD != a[0].PAD ||   // (kernel sizes may differ: the four phases of a stride-2 dgrad)
        (x.nred > 0) != (a[0].nred > 0) || (x.accumulate != 0) != (a[0].accumulate != 0) || (x.stats != nullptr) != (a[0].stats != nullptr)) return YS_ERR_UNSUPPORTED;
    if (x.M > a[big].M) big = i;
  }
  p[big] = conv_p2_plan(a[big]);
  if (!p[big].ok) return YS_ERR_UNSUPPORTED;
  for (int i = 0; i < n; i++) {
    if (i == big) continue;
    p[i] = conv_p2_plan(a[i], p[big].mr, p[big].npu, false, p[big].nr);
    if (!p[i].ok || p[i].nr != p[big].nr || p[i].wres != p[big].wres || p[i].nt != p[big].nt || p[i].gy != p[big].gy) return YS_ERR_UNSUPPORTED;
  }
  // one persistent grid: the slots of the largest LDS footprint's occupancy class, split in proportion to the tile counts
  size_t lds = 0; int per_cu = 3;
  long tiles = 0;
  for (int i = 0; i < n; i++) { lds = lds > p[i].lds ? lds : p[i].lds; per_cu = per_cu < p[i].per_cu ? per_cu : p[i].per_cu; tiles += p[i].g.ntiles; }
  const long slots = ((long)ys_cu_count() * per_cu) / p[0].gy > 0 ? ((long)ys_cu_count() * per_cu) / p[0].gy : 1;
  int gxs[YS_GROUP_MAX];
  long used = 0;
  for (int i = 0; i < n; i++) {
    long gx = (slots * p[i].g.ntiles + tiles / 2) / tiles;
    if (gx >= 8) gx = (gx + 4) / 8 * 8;          // multiples of 8: workgroup vbx keeps running on XCD vbx % 8
    if (gx < 1) gx = 1;
    if (gx > p[i].g.ntiles) gx = p[i].g.ntiles;
    if (row_cap && row_cap[i] > 0 && gx > row_cap[i]) gx = row_cap[i];
    gxs[i] = (int)gx; used += gx;
  }
  // The grid must FIT: rounding every range up to a multiple of 8 made it 520-528 workgroups for 512 slots (776 for 768) on the Detect
  // towers -- the last 8 wait for a slot, start when the first workgroups leave and then walk their whole tile share alone (round 4,
  // per-launch records: the grouped 80 -> 80 layer 215 us against 174 us for its three levels launched one by one).  Take the excess
  // from the largest ranges, 8 at a time (1 at a time below 16).
  while (used > slots) {
    int big = 0;
    for (int i = 1; i < n; i++) if (gxs[i] > gxs[big]) big = i;
    const int dec = gxs[big] >= 16 ? 8 : 1;
    if (gxs[big] - dec < 1) break;
    gxs[big] -= dec; used -= dec;
  }
  // a range that is not a multiple of 8 shifts the XCD phase of the ranges behind it: order the problems so that only the last may be ragged
  // (ranges of >= 8 workgroups are multiples of 8 unless a cap cut them; a shifted phase only costs L2 locality, never correctness)
  for (int i = 0; i < n; i++) if (rows) rows[i] = gxs[i];
  if (plan_only) return YS_OK;
  return conv_p2_group_dispatch(st, a, p, n, gxs, lds);
}

// host-side tile choice for the patch kernel: largest pixel tile (64*MR) whose patch fits PATCH_UNITS, shaped to
// minimise (patch pixels loaded) + (idle tile pixels), keeping a few workgroups per CU
struct TileChoice { int mr, th, tw, tx, ty; };
static TileChoice conv3x3_pick_tile(const ConvArgs& a, int nr) {
  const int gy = ys_cdiv(a.Cout, nr * 16);
  const int div = a.DIVM + 1;
  TileChoice best{0, 0, 0, 0, 0};
  for (int mr = 2; mr >= 1; mr >>= 1) {
    const int npx = 128 * mr;   // 8 waves x MR fragments x 16 pixels
    double best_cost = 1e30;
    TileChoice cur{0, 0, 0, 0, 0};
    for (int tw = 1; tw <= npx && tw <= a.Wout; tw++) {
      const int th_max = (npx / tw) < a.Hout ? (npx / tw) : a.Hout;
      for (int th = th_max; th >= 1; th--) {
        const int ph = div == 1 ? (th - 1) * a.SA + 3 : (th + 1) / 2 + 2, pw = div == 1 ? (tw - 1) * a.SA + 3 : (tw + 1) / 2 + 2;
        if ((long)ph * pw * 5 > PATCH_UNITS) continue;
        const int tx = ys_cdiv(a.Wout, tw), ty = ys_cdiv(a.Hout, th);
        const double cost = (double)tx * ty * ((double)ph * pw + 0.5 * npx);
        if (cost < best_cost) { best_cost = cost; cur = TileChoice{mr, th, tw, tx, ty}; }
      }
    }
    if (cur.mr == 0) continue;
    if (best.mr == 0) best = cur;
    if ((long)cur.tx * cur.ty * a.B * gy >= 512) return cur;   // enough tiles to feed every CU: take the largest
    best = cur;                                                // otherwise keep shrinking
  }
  return best;
}
struct C3Plan { int nr, wres, nchunks, wpitch, patch_units; size_t lds_bytes; };
static C3Plan conv3x3_plan(const ConvArgs& a, int epl) {
  const int nfr = (a.Cout + 15) / 16;
  const int nchunks = (a.Cin + 4 * epl - 1) / (4 * epl);
  C3Plan p{};
  p.nchunks = nchunks;
  const int es = epl == 8 ? 2 : 4;
  auto patch_units_for = [&](int nr) {   // patch region doubles as the epilogue staging area (8 waves x 16 x (BN+EPL) elements)
    const int need = (8 * 16 * (nr * 16 + epl) * es + 15) / 16;
    return need > PATCH_UNITS ? need : PATCH_UNITS;
  };
  for (int nr = nfr < 5 ? nfr : 5; nr >= 1; nr--) {       // weights of all chunks resident in LDS?
    const size_t fixed = (size_t)patch_units_for(nr) * 16;
    const size_t bytes = (size_t)nr * 16 * (nchunks * 36 + 1) * 16 + fixed + (size_t)nr * 16 * 64;
    if (bytes <= 144 * 1024 && (nr == nfr || nr * 2 >= (nfr < 5 ? nfr : 5))) {   // do not shrink NR below half for residency
      p.nr = nr; p.wres = 1; p.wpitch = nchunks * 36 + 1; p.lds_bytes = bytes; p.patch_units = patch_units_for(nr);
      return p;
    }
  }
  p.nr = nfr <= 5 ? nfr : (nfr % 5 == 0 ? 5 : 4);
  p.wres = 0; p.wpitch = 37; p.patch_units = patch_units_for(p.nr);
  p.lds_bytes = (size_t)p.nr * 16 * 37 * 16 + (size_t)p.patch_units * 16 + (size_t)p.nr * 16 * 64;
  return p;
}
// stride-2 forward convs have a 4x larger input patch per output pixel: their tiles would be patch-limited to ~64 pixels,
// so they stay on the direct-fragment kernel; stride-1 forward and all dgrads (SA == 1) use the LDS patch kernel
static bool conv_use_patch(const ConvArgs& a) { return a.KH == 3 && a.KW == 3 && a.SA == 1; }

template <class T, int MR, int NR>
static void conv_launch_t(hipStream_t st, const ConvArgs& a) {
  dim3 grid(ys_cdiv(a.M, 4 * MR * 16), ys_cdiv(a.Cout, NR * 16));
  char lab[128] = "";
  if (ys_kprof_enabled()) snprintf(lab, sizeof(lab), "direct k%d s%d div%d cin%d cout%d M%d acc%d mr%d nr%d", a.KH, a.SA, a.DIVM + 1, a.Cin, a.Cout, a.M, a.accumulate, MR, NR);
  YsKprofScope prof(st, "conv_igemm", lab);
  YS_LAUNCH((conv_igemm_kernel<T, MR, NR>), grid, 256, st, a);
}

// Tile selection.  NR = output-channel fragments per workgroup (all of Cout when it fits, so activations are read
// once); MR = pixel fragments per wave: 2 (occupancy: three workgroups per CU overlap each other's load and
// epilogue phases; larger MR measured slower), 1 on small feature maps (20x20, 40x40 at the deep end of the net)
// so that the grid still covers the 256 CUs a few times.
static int conv_pick_nr(int cout, int M = 1 << 30) {
  const int nfr = (cout + 15) / 16;
  if (nfr <= 6) return nfr;
  // small feature maps: narrower column tiles give the chip more workgroups (Cin = 128 -> 256 stride-2 layer at 20x20: 119 -> 83 us)
  const int smallm = 30000;
  if (M <= smallm && nfr % 4 == 0) return 4;
  if (nfr % 8 == 0 || nfr > 10) return 8;
  if (nfr % 5 == 0) return 5;
  return 4;
}
static int conv_pick_mr(int M, int cout) {
  const int nr = conv_pick_nr(cout, M);
  const int gy = ys_cdiv(cout, nr * 16);
  int mr = 2;   // measured: 2 fragments per wave (<= ~130 registers, 3 workgroups per CU) beats 4-8 fragments at 2 per CU
  while (mr > 1 && (long)ys_cdiv(M, 64 * mr) * gy < 768) mr >>= 1;
  return mr;
}

// fp8 request on the blocked-GEMM kernel: possible when the caller lent a quantisation scratch (or the image is already there) and
// the images of the batch are contiguous in the input view; b = a with the fp8 image in place
static bool conv_f8_gemm_args(const ConvArgs& a, ConvArgs& b) {
  b = a;
  if (!a.f8 || !(a.x8 || a.q8) || a.in_bstride != (long)a.Hin * a.Win) return false;
  if (!b.x8) b.x8 = a.q8;
  return ys_conv_gemm_rows(b) != 0;
}

// fp8 pays where the quantisation pass of the input is amortised over several taps: measured on YOLOv8x 1280 (config 5), every 1x1
// layer is faster on the bf16 kernels once that pass is counted (e.g. 1280 -> 320 @ 160x160: 480 us bf16 vs 306 + 393 us), every
// 3x3 layer is 1.3-1.5x faster in fp8.  YS_F8_MIN_TAPS overrides (the tests run 1x1 layers in fp8 too).
static bool conv_f8_declined(const ConvArgs& a) {
  const int min_taps = (int)YS_OPT_INT("F8_MIN_TAPS", 4);
  return a.f8 && !a.x8 && a.KH * a.KW < min_taps;
}

int ys_conv_grid_m(const ConvArgs& a, int dtype) {
  if (conv_f8_declined(a)) { ConvArgs b = a; b.f8 = 0; return ys_conv_grid_m(b, dtype); }
  if (a.f8) {                       // mirrors ys_conv_launch: the fp8 plan, else the bf16 kernels
    ConvArgs g8;
    if (dtype == YS_BF16 && conv_f8_gemm_args(a, g8)) return ys_conv_gemm_rows(g8);
    const P2Plan pf = dtype == YS_BF16 ? conv_p2_plan(a) : P2Plan{};
    if (pf.ok) return pf.gx;
    ConvArgs b = a; b.f8 = 0;
    return ys_conv_grid_m(b, dtype);
  }
  if (dtype == YS_BF16) {
    const int gr = ys_conv_gemm_rows(a);   // wide layers: the blocked GEMM kernel (conv_gemm.hip), same one-row-per-workgroup contract
    if (gr) return gr;
    const P2Plan p2 = conv_p2_plan(a);
    if (p2.ok) return p2.gx;   // one statistics row per (persistent) workgroup
  }
  if (conv_use_patch(a)) {
    // tile shape does not depend on dtype-specific chunking; NR only enters through the workgroup-count heuristic
    const TileChoice t = conv3x3_pick_tile(a, conv3x3_plan(a, 8).nr);
    return t.tx * t.ty * a.B;
  }
  return ys_cdiv(a.M, 64 * conv_pick_mr(a.M, a.Cout));
}

// channel tiles (gridDim.y) of the conv_p2_kernel launch a bf16, non-fp8 forward convolution would get; 0 = another kernel runs it
int ys_conv_is_p2(const ConvArgs& a) {
  if (a.f8 || ys_conv_gemm_rows(a)) return 0;
  const P2Plan p2 = conv_p2_plan(a);
  return p2.ok ? p2.gy : 0;
}

template <class T, int MR, int NR>
static int conv3x3_launch_t(hipStream_t st, ConvArgs a, const TileChoice& t, const C3Plan& p) {
  a.TH = t.th; a.TW = t.tw; a.tiles_x = t.tx; a.tiles_y = t.ty;
#ifdef YS_P2_ABLATE
  a.dbg = (int)YS_OPT_INT("DBG", 0);
#endif
  const int ntiles = t.tx * t.ty * a.B;
  const int gy = ys_cdiv(a.Cout, NR * 16);
  const int per_cu = p.lds_bytes <= 76 * 1024 ? 2 : 1;
  int gx = (ys_cu_count() * per_cu + gy - 1) / gy;
  if (gx > ntiles) gx = ntiles;
  static std::atomic<unsigned> attr_done{0};      // per device: the attribute belongs to the device's code object
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  if (!(attr_done.load(std::memory_order_relaxed) & (1u << (dev_id & 31)))) {
    hipFuncSetAttribute((const void*)conv3x3_tile_kernel<T, MR, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.fetch_or(1u << (dev_id & 31), std::memory_order_relaxed);
  }
  char lab[160] = "";
  if (ys_kprof_enabled()) snprintf(lab, sizeof(lab), "patch k3 s%d div%d cin%d cout%d M%d acc%d mr%d nr%d tile%dx%d grid%dx%d lds%d", a.SA, a.DIVM + 1, a.Cin, a.Cout, a.M, a.accumulate, MR, NR, a.TH, a.TW, gx, gy, (int)p.lds_bytes);
  YsKprofScope prof(st, "conv_igemm", lab);
  YS_LAUNCH_LDS((conv3x3_tile_kernel<T, MR, NR>), dim3(gx, gy), C3_THREADS, p.lds_bytes, st, a, ntiles, p.nchunks, p.wres, p.wpitch, p.patch_units);
  return YS_OK;
}

template <class T>
static int conv_launch_dtype(hipStream_t st, const ConvArgs& a) {
  if (conv_use_patch(a)) {
    const C3Plan p = conv3x3_plan(a, Elem<T>::EPL);
    const TileChoice t = conv3x3_pick_tile(a, conv3x3_plan(a, 8).nr);   // same tile for both dtypes (stats grid)
    if (t.mr == 0) { ys_set_error("conv3x3: no tile fits"); return YS_ERR_UNSUPPORTED; }
#define C3(M_, N_) if (t.mr == M_ && p.nr == N_) return conv3x3_launch_t<T, M_, N_>(st, a, t, p);
    C3(1, 1) C3(2, 1) C3(1, 2) C3(2, 2) C3(1, 3) C3(2, 3) C3(1, 4) C3(2, 4) C3(1, 5) C3(2, 5)
#undef C3
    ys_set_error("conv3x3: no kernel for tile MR=%d NR=%d", t.mr, p.nr);
    return YS_ERR_UNSUPPORTED;
  }
  const int nr = conv_pick_nr(a.Cout, a.M), mr = conv_pick_mr(a.M, a.Cout);
#define CV(M_, N_) if (mr == M_ && nr == N_) { conv_launch_t<T, M_, N_>(st, a); return YS_OK; }
  CV(1, 1) CV(2, 1) CV(4, 1) CV(8, 1)
  CV(1, 2) CV(2, 2) CV(4, 2) CV(8, 2)
  CV(1, 3) CV(2, 3) CV(4, 3)
  CV(1, 4) CV(2, 4) CV(4, 4)
  CV(1, 5) CV(2, 5)
  CV(1, 6) CV(2, 6)
  CV(1, 8) CV(2, 8)
#undef CV
  ys_set_error("conv: no kernel for tile MR=%d NR=%d", mr, nr);
  return YS_ERR_UNSUPPORTED;
}

// ---- stride-2 3x3 dgrad as four phase convolutions (bf16).  dx[ih, iw] only receives the taps whose parity matches
// (ih, iw): phase (a, b) = (ih & 1, iw & 1) is a stride-1 correlation of dy with KH_a x KW_b taps (1 if the parity is 0,
// else 2: row offsets {0, +1} <-> kh = {2, 0}) written to the rows 2i+a / columns 2j+b of dx.  9/4 taps per output pixel
// instead of 9 with 3/4 of them predicated off.  The dgrad weights of such a layer are laid out phase-major by the weight
// prep: [phase][Cin_real][taps_p][Cout_pad], phases in the order (0,0) (0,1) (1,0) (1,1) -> tap offsets 0, 1, 3, 5.
bool ys_conv_dgrad_uses_phases(int dtype, int k, int stride) { return dtype == YS_BF16 && k == 3 && stride == 2; }

// arguments of phase ph of a stride-2 3x3 dgrad; false when the phase has no output pixels
static bool conv_dgrad_s2_phase_args(const ConvArgs& a, int ph, ConvArgs& q) {
  static const int toff[4] = {0, 1, 3, 5};
  const int Hx = a.Hout, Wx = a.Wout;                     // dx grid (the layer's input)
  const int pa = ph >> 1, pb = ph & 1;
  q = a;
  q.KH = pa ? 2 : 1; q.KW = pb ? 2 : 1;
  q.SA = 1; q.DIVS = 0; q.DIVM = 0; q.PAD = 0;
  q.Hout = (Hx - pa + 1) / 2; q.Wout = (Wx - pb + 1) / 2;
  if (q.Hout <= 0 || q.Wout <= 0) return false;
  q.M = a.B * q.Hout * q.Wout;
  q.w = (const char*)a.w + (size_t)toff[ph] * a.Cout * a.Cin * 2;   // [phase][Cout = layer Cin_real][taps_p][Cin = layer Cout_pad]
  if (a.w8) q.w8 = (const char*)a.w8 + (size_t)toff[ph] * a.Cout * a.Cin;
  q.out_rh = 2 * Wx; q.out_rw = 2; q.out_r0 = (long)pa * Wx + pb;
  return true;
}

// rows_only: no launch, returns the partial rows the fused BN-backward reduction of the four launches would write (0 = a phase runs a
// kernel without that epilogue); otherwise launches and returns a status
static int conv_dgrad_s2_phases(hipStream_t st, const ConvArgs& a, bool rows_only = false) {
  int row0 = a.red_row0;
  // The four phases as ONE persistent grid (conv_p2_group_kernel) when every phase is a bf16 patch-kernel launch of one variant: they read the
  // same dy tiles -- side by side on an XCD the second to fourth read hit its L2 -- and three launches' fixed costs go.  The 2x2-tap phase
  // first: the group runs on the variant of its first largest member, and that phase has the largest patch.
  if (!a.f8) {
    ConvArgs qs[4];
    int n = 0;
    for (int ph = 3; ph >= 0; ph--) if (conv_dgrad_s2_phase_args(a, ph, qs[n])) n++; else break;
    int rows[4];
    if (n == 4 && ys_conv_p2_group_launch(nullptr, qs, 4, nullptr, rows, true) == YS_OK) {
      for (int i = 0; i < 4; i++) { qs[i].red_row0 = row0; row0 += rows[i]; }
      if (rows_only) return row0 - a.red_row0;
      return ys_conv_p2_group_launch(st, qs, 4, nullptr, rows);
    }
    row0 = a.red_row0;
  }
  for (int ph = 0; ph < 4; ph++) {
    ConvArgs q;
    if (!conv_dgrad_s2_phase_args(a, ph, q)) continue;
    q.red_row0 = row0;                                                // the phases write disjoint pixels of dx: their statistics rows stack
    if (q.f8 && q.x8 && ys_conv_gemm_rows(q)) {
      if (rows_only && q.f8 != 2) return 0;                           // only the e5m2-input (dgrad) form of the fp8 GEMM kernel carries the fused reduction
      row0 += ys_conv_gemm_rows(q);
      if (!rows_only) { const int rc = ys_conv_gemm_launch(st, q); if (rc != YS_OK) return rc; }
      continue;
    }
    P2Plan p2 = conv_p2_plan(q);
    if (p2.ok && q.f8 && rows_only) return 0;
    if (!p2.ok && q.f8) { q.f8 = 0; p2 = conv_p2_plan(q); }          // no fp8 plan for this shape: bf16 kernel, same result type
    if (!q.f8 && ys_conv_gemm_rows(q)) {
      row0 += ys_conv_gemm_rows(q);
      if (!rows_only) { const int rc = ys_conv_gemm_launch(st, q); if (rc != YS_OK) return rc; }
      continue;
    }
    if (p2.ok) { row0 += p2.gx; if (!rows_only) { const int rc = conv_p2_dispatch(st, q, p2); if (rc != YS_OK) return rc; } }
    else { if (rows_only) return 0; const int rc = conv_launch_dtype<bf16_t>(st, q); if (rc != YS_OK) return rc; }
  }
  return rows_only ? row0 - a.red_row0 : YS_OK;
}

// mirrors ys_conv_launch's routing (bf16 storage): the fused reduction lives in conv_epi.h, i.e. in conv_p2_kernel and conv_gemm_kernel
int ys_conv_bnred_rows(const ConvArgs& a, int dtype) {
  if (dtype != YS_BF16) return 0;
  if (conv_f8_declined(a)) { ConvArgs b = a; b.f8 = 0; return ys_conv_bnred_rows(b, dtype); }
  const bool phases = ys_conv_dgrad_uses_phases(dtype, a.KH, a.DIVM + 1) && a.KW == a.KH;
  if (a.f8 && !a.x8 && a.q8 && ys_conv_wants_x8(a)) { ConvArgs b = a; b.x8 = a.q8; return ys_conv_bnred_rows(b, dtype); }
  if (phases) return conv_dgrad_s2_phases(nullptr, a, true);
  if (a.f8) {                                   // fp8 routing: the blocked-GEMM kernel has a fused variant for e5m2 (dgrad) input, conv_p2_kernel<F8> has none
    if (a.x8 && ys_conv_gemm_rows(a)) return a.f8 == 2 ? ys_conv_gemm_rows(a) : 0;
    const P2Plan pf = conv_p2_plan(a);
    if (pf.ok) return 0;
    ConvArgs b = a; b.f8 = 0;
    return ys_conv_bnred_rows(b, dtype);
  }
  if (ys_conv_gemm_rows(a)) return ys_conv_gemm_rows(a);
  const P2Plan p2 = conv_p2_plan(a);
  return p2.ok ? p2.gx : 0;
}

bool ys_conv_wants_x8(const ConvArgs& a) {
  if (!a.f8 || a.x8 || a.in_bstride != (long)a.Hin * a.Win) return false;
  ConvArgs b = a; b.x8 = a.x;                 // any non-null pointer: the plans only test for presence
  if (ys_conv_dgrad_uses_phases(YS_BF16, a.KH, a.DIVM + 1) && a.KW == a.KH) {
    for (int ph = 0; ph < 4; ph++) { ConvArgs q; if (conv_dgrad_s2_phase_args(b, ph, q) && ys_conv_gemm_rows(q) != 0) return true; }
    return false;
  }
  return ys_conv_gemm_rows(b) != 0;
}

int ys_conv_launch(hipStream_t st, int dtype, const ConvArgs& a) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  if (a.Cin % epl || a.in_ldc % epl || a.in_coff % epl) {
    ys_set_error("conv: Cin/ldc/coff (%d,%d,%d) must be multiples of %d", a.Cin, a.in_ldc, a.in_coff, epl);
    return YS_ERR_INVALID_ARG;
  }
  if (dtype == YS_BF16) {
    if (conv_f8_declined(a)) { ConvArgs b = a; b.f8 = 0; return ys_conv_launch(st, dtype, b); }
    const bool phases = ys_conv_dgrad_uses_phases(dtype, a.KH, a.DIVM + 1) && a.KW == a.KH;
    if (a.f8 && !a.x8 && a.q8) {
      // fp8 on the blocked-GEMM kernel: its operand tiles reach LDS by DMA, so the input view is quantised into the caller's
      // scratch first (one pass that also records amax(|input|) for the next step's scale) -- when the layer, or a phase of its
      // stride-2 dgrad, will run that kernel
      ConvArgs b = a; b.x8 = a.q8;
      if (ys_conv_wants_x8(a)) {
        const int rc = ys_f8_quant_view_launch(st, a.f8 == 2 ? 1 : 0, a.x, (long)a.B * a.Hin * a.Win, a.Cin, a.in_ldc, a.in_coff, a.qscale, a.q8, a.amax);
        if (rc != YS_OK) return rc;
        b.amax = nullptr;                                               // recorded by the quantisation pass
        return ys_conv_launch(st, dtype, b);
      }
    }
    if (phases) return conv_dgrad_s2_phases(st, a);
    if (a.f8) {                                                       // fp8 request: the blocked-GEMM kernel on the fp8 image, else the P2 kernel's fp8 mode
      if (a.x8 && ys_conv_gemm_rows(a)) return ys_conv_gemm_launch(st, a);
      const P2Plan pf = conv_p2_plan(a);
      if (pf.ok) return conv_p2_dispatch(st, a, pf);
      ConvArgs b = a; b.f8 = 0;
      return ys_conv_launch(st, dtype, b);
    }
    if (ys_conv_gemm_rows(a)) return ys_conv_gemm_launch(st, a);
    const P2Plan p2 = conv_p2_plan(a);
    if (p2.ok) return conv_p2_dispatch(st, a, p2);
    return conv_launch_dtype<bf16_t>(st, a);
  }
  return conv_launch_dtype<float>(st, a);
}

// Development aid (tools/dev/p2_plans.py): the P2 plan of a forward convolution geometry as text -- no device needed.
extern "C" __attribute__((visibility("default"))) int ys_debug_p2_plan(int B, int Hin, int Win, int Cin, int Cout, int k, int s, int in_ldc, char* buf, int cap) {
  ConvArgs a{};
  a.B = B; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Cout = Cout; a.KH = a.KW = k; a.SA = s; a.PAD = k / 2;
  a.Hout = (Hin + 2 * (k / 2) - k) / s + 1; a.Wout = (Win + 2 * (k / 2) - k) / s + 1;
  a.in_ldc = in_ldc; a.in_bstride = (long)Hin * Win; a.out_ldc = Cout; a.out_bstride = (long)a.Hout * a.Wout; a.M = B * a.Hout * a.Wout;
  if (ys_conv_gemm_rows(a)) { snprintf(buf, cap, "gemm"); return 2; }
  const P2Plan p = conv_p2_plan(a);
  if (!p.ok) { snprintf(buf, cap, "none"); return 0; }
  const double cyc = p2_read_cycles(a.Cin, a.KH, a.KW, a.SA, p.g.TH, p.g.TW, p.mr, p.nt / 64, p.g.ppb, p.g.prb);
  snprintf(buf, cap, "mr%d nr%d wres%d npu%d tile%dx%d grid%dx%d lds%zu ppb%d prb%d(pad%d) wpitch%d readcyc%.2f", p.mr, p.nr, p.wres, p.npu, p.g.TH, p.g.TW,
           p.gx, p.gy, p.lds, p.g.ppb, p.g.prb, (p.g.prb - p.g.PW * p.g.ppb) / 16, p.g.wpitch, cyc);
  return 1;
}
