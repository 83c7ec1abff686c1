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

template <class T, int MR, int NR>
__global__ void __launch_bounds__(256)
conv_igemm_kernel(ConvArgs a) {
  constexpr int EPL = Elem<T>::EPL;
  constexpr int BN = NR * 16;
  constexpr int LDS_UNITS = 3072;                       // 48 KiB of staged weights (3 workgroups per CU)
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
      const int iwn = pow_[mf] * a.SA + kw - a.PAD;
      const int ih = ihn >> a.DIVS, iw = iwn >> a.DIVS;
      const bool ok = pv[mf] && ihn >= 0 && iwn >= 0 && ((ihn | iwn) & a.DIVM) == 0 && ih < a.Hin && iw < a.Win;
      pvt[mf] = ok;
      const long pix = ok ? ((long)pb[mf] * a.in_bstride + (long)ih * a.Win + iw) : 0;
      px[mf] = xb + (pix * a.in_ldc + a.in_coff) * (long)sizeof(T);
    }
  };
  auto load_step = [&](uint4* xf, int s) {
    const int c = (4 * s + q) * EPL;
#pragma unroll
    for (int mf = 0; mf < MR; mf++) {
      xf[mf] = ys_zero16();
      if (pvt[mf] && c < a.Cin) xf[mf] = ys_ld16(px[mf] + (long)c * (long)sizeof(T));
    }
  };
  uint4 xc[MR], xn[MR];
#pragma unroll
  for (int mf = 0; mf < MR; mf++) xn[mf] = ys_zero16();
  int ltap = 0, ls = 0, lkh = 0, lkw = 0;   // position of the NEXT load
  set_tap(0, 0);
  load_step(xc, 0);
  int lt = 0;                                // k-step index inside the staged weight group
  for (int t = 0; t < total; t++) {
    // advance the load position and prefetch step t+1
    ls++;
    if (ls == spt) {
      ls = 0; ltap++; lkw++;
      if (lkw == a.KW) { lkw = 0; lkh++; }
      if (ltap < taps) set_tap(lkh, lkw);
    }
    if (t + 1 < total) load_step(xn, ls);
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
    for (int mf = 0; mf < MR; mf++) xc[mf] = xn[mf];
    lt++;
    if (lt == GSTEPS) lt = 0;
  }

  // ------------------------------------------------------------------ epilogue
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
    const int m = m0 + mf * 16 + li;
    const bool mv = m < a.M;
    const int mm = mv ? m : 0;
    const int b = mm / HWo;
    const int pix = mm - b * HWo;
    const long orow = (long)b * a.out_bstride + pix;
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
      if (rb && mv) {
        const T* rp = (const T*)(rb + (orow * a.res_ldc + a.res_coff + c) * (long)sizeof(T));
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (c + r < a.Cout) v[r] += Elem<T>::to_f(rp[r]);
      }
      T* yp = (T*)(yb + (orow * a.out_ldc + a.out_coff + c) * (long)sizeof(T));
      if (a.accumulate && mv) {
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (c + r < a.Cout) v[r] += Elem<T>::to_f(yp[r]);
      }
      T o[4];
#pragma unroll
      for (int r = 0; r < 4; r++) o[r] = Elem<T>::from_f(v[r]);
      if (do_stats) {
        // statistics of the values actually stored (what BN will normalise)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float f = Elem<T>::to_f(o[r]);
          s1[nf][r] += f;
          s2[nf][r] += f * f;
        }
      }
      if (mv) {
        if (a.vec_ok && c + 3 < a.Cout) {
          if (sizeof(T) == 2) {
            uint2 pk;
            pk.x = ys_pack_bf16x2(v[0], v[1]);
            pk.y = ys_pack_bf16x2(v[2], v[3]);
            *(uint2*)yp = pk;
          } else {
            *(uint4*)yp = make_uint4(ys_f2u(v[0]), ys_f2u(v[1]), ys_f2u(v[2]), ys_f2u(v[3]));
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (c + r < a.Cout) yp[r] = o[r];
        }
      }
    }
  }

  if (do_stats) {
    // reduce over the 16 pixels held by lanes with equal q (xor 1,2,4,8), then over waves via LDS
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
          sStat[wave][nf * 16 + 4 * q + r][0] = x1;
          sStat[wave][nf * 16 + 4 * q + r][1] = x2;
        }
      }
    __syncthreads();
    if (tid < BN && n0 + tid < a.Cout) {
      const float t1 = sStat[0][tid][0] + sStat[1][tid][0] + sStat[2][tid][0] + sStat[3][tid][0];
      const float t2 = sStat[0][tid][1] + sStat[1][tid][1] + sStat[2][tid][1] + sStat[3][tid][1];
      a.stats[((long)blockIdx.x * 2 + 0) * a.Cout + n0 + tid] = t1;
      a.stats[((long)blockIdx.x * 2 + 1) * a.Cout + n0 + tid] = t2;
    }
  }
}

template <class T, int MR, int NR>
static void conv_launch_t(hipStream_t st, const ConvArgs& a) {
  dim3 grid(ys_cdiv(a.M, 4 * MR * 16), ys_cdiv(a.Cout, NR * 16));
  YsKprofScope prof(st, "conv_igemm");
  YS_LAUNCH((conv_igemm_kernel<T, MR, NR>), grid, 256, st, a);
}

int ys_conv_grid_m(const ConvArgs& a) { return ys_cdiv(a.M, 4 * 2 * 16); }

template <class T>
static int conv_launch_dtype(hipStream_t st, const ConvArgs& a) {
  const int nfr = (a.Cout + 15) / 16;
  // MR is fixed at 2 (128 pixels per workgroup) so that the BN partial-statistics grid is known
  if (nfr <= 1) conv_launch_t<T, 2, 1>(st, a);
  else if (nfr == 2) conv_launch_t<T, 2, 2>(st, a);
  else if (nfr == 3) conv_launch_t<T, 2, 3>(st, a);
  else if (nfr == 4) conv_launch_t<T, 2, 4>(st, a);
  else if (nfr == 5) conv_launch_t<T, 2, 5>(st, a);
  else if (nfr == 6) conv_launch_t<T, 2, 6>(st, a);
  else if (nfr % 8 == 0 || nfr > 10) conv_launch_t<T, 2, 8>(st, a);
  else if (nfr % 5 == 0) conv_launch_t<T, 2, 5>(st, a);
  else conv_launch_t<T, 2, 4>(st, a);
  return YS_OK;
}

int ys_conv_launch(hipStream_t st, int dtype, const ConvArgs& a) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  if (a.Cin % epl || a.in_ldc % epl || a.in_coff % epl) {
    ys_set_error("conv: Cin/ldc/coff (%d,%d,%d) must be multiples of %d", a.Cin, a.in_ldc, a.in_coff, epl);
    return YS_ERR_INVALID_ARG;
  }
  if (dtype == YS_BF16) return conv_launch_dtype<bf16_t>(st, a);
  return conv_launch_dtype<float>(st, a);
}

// ===================================================================================== wgrad
// dW[co][tap][ci] = sum_p dy[p][co] * x[pix(p,tap)][ci].  Both operands have the reduction dim
// (pixels) as the slow axis in NHWC, so tiles are staged pixel-major in LDS and read transposed.
// Grid: (pixel splits, co-tile x ci-tile, taps).  Each wave owns a quarter of the workgroup's pixel
// range; waves are combined through LDS and the workgroup writes ONE fp32 partial tile, which
// wgrad_reduce_kernel sums over splits in a fixed order (deterministic, no atomics).
template <class T, int MRA, int NRB>
__global__ void __launch_bounds__(256)
conv_wgrad_kernel(WgradArgs a) {
  constexpr int EPL = Elem<T>::EPL;
  constexpr int KS = 4 * EPL;  // pixels per wave-step
  constexpr int COT = MRA * 16, CIT = NRB * 16;
  constexpr int PD = COT + 16 / (int)sizeof(T);  // LDS pitches in elements (rows stay 16-byte aligned)
  constexpr int PX = CIT + 16 / (int)sizeof(T);
  constexpr int STAGE_BYTES = 4 * KS * (PD + PX) * (int)sizeof(T);
  constexpr int RED_BYTES = 4 * COT * CIT * 4;
  constexpr int LDS_BYTES = STAGE_BYTES > RED_BYTES ? STAGE_BYTES : RED_BYTES;
  __shared__ uint4 smem[LDS_BYTES / 16];
  T* sD = (T*)smem;                       // [4][KS][PD]
  T* sX = sD + 4 * KS * PD;               // [4][KS][PX]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
  const int ci_tiles = (a.Cin + CIT - 1) / CIT;
  const int co0 = (blockIdx.y / ci_tiles) * COT;
  const int ci0 = (blockIdx.y % ci_tiles) * CIT;
  const int tap = blockIdx.z;
  const int kh = tap / a.KW, kw = tap % a.KW;
  const int HWo = a.Hout * a.Wout;
  // pixel range of this workgroup, in wave-steps
  const long steps_total = ((long)a.M + KS - 1) / KS;
  const long steps_per_blk = (steps_total + gridDim.x - 1) / gridDim.x;
  const long sb = (long)blockIdx.x * steps_per_blk;
  long se = sb + steps_per_blk;
  if (se > steps_total) se = steps_total;
  const long iters = (steps_per_blk + 3) / 4;  // uniform trip count for all waves/blocks

  f32x4 acc[MRA][NRB];
#pragma unroll
  for (int i = 0; i < MRA; i++)
#pragma unroll
    for (int j = 0; j < NRB; j++) acc[i][j] = f32x4_zero();

  T* myD = sD + wave * KS * PD;
  T* myX = sX + wave * KS * PX;
  const char* dyb = (const char*)a.dy;
  const char* xb = (const char*)a.x;
  constexpr int DV = COT / EPL;  // 16-byte vectors per pixel row of the dy tile
  constexpr int XV = CIT / EPL;

  // Each wave stages and consumes its own LDS region: no workgroup barrier inside the loop, so the 4 waves (and the
  // other resident workgroups) drift apart and overlap each other's load / transpose / MFMA phases.  The global loads
  // of wave-step it+1 are issued into registers before the MFMAs of wave-step it.
  constexpr int ND = (KS * DV + 63) / 64, NX = (KS * XV + 63) / 64;
  uint4 rd[ND], rx[NX];
  auto fetch = [&](long it) {
    const long step = sb + it * 4 + wave;
    const bool active = step < se;
    const long p0 = step * KS;
#pragma unroll
    for (int k = 0; k < ND; k++) {
      const int v = lane + 64 * k;
      const int pr = v / DV, cv = v % DV;
      const long p = p0 + pr;
      const int c = co0 + cv * EPL;
      uint4 val = ys_zero16();
      if (v < KS * DV && active && p < a.M && c < a.Cout) {
        const long bb = p / HWo;
        const long drow = bb * a.dy_bstride + (p - bb * HWo);
        val = ys_ld16(dyb + ((drow * a.dy_ldc) + a.dy_coff + c) * (long)sizeof(T));
      }
      rd[k] = val;
    }
#pragma unroll
    for (int k = 0; k < NX; k++) {
      const int v = lane + 64 * k;
      const int pr = v / XV, cv = v % XV;
      const long p = p0 + pr;
      const int c = ci0 + cv * EPL;
      uint4 val = ys_zero16();
      if (v < KS * XV && active && p < a.M && c < a.Cin) {
        const int b = (int)(p / HWo);
        const int r = (int)(p - (long)b * HWo);
        const int oh = r / a.Wout, ow = r - oh * a.Wout;
        const int ih = oh * a.stride + kh - a.pad, iw = ow * a.stride + kw - a.pad;
        if (ih >= 0 && ih < a.Hin && iw >= 0 && iw < a.Win)
          val = ys_ld16(xb + ((((long)b * a.in_bstride + (long)ih * a.Win + iw) * a.in_ldc) + a.in_coff + c) * (long)sizeof(T));
      }
      rx[k] = val;
    }
  };
  fetch(0);
  for (long it = 0; it < iters; it++) {
    // ---- registers -> this wave's LDS tiles: dy [KS][COT], gathered x [KS][CIT]
#pragma unroll
    for (int k = 0; k < ND; k++) {
      const int v = lane + 64 * k;
      if (v < KS * DV) *(uint4*)(myD + (v / DV) * PD + (v % DV) * EPL) = rd[k];
    }
#pragma unroll
    for (int k = 0; k < NX; k++) {
      const int v = lane + 64 * k;
      if (v < KS * XV) *(uint4*)(myX + (v / XV) * PX + (v % XV) * EPL) = rx[k];
    }
    ys_wave_sync();
    if (it + 1 < iters) fetch(it + 1);
    // ---- transposed fragment reads + MFMA
    uint4 fa[MRA], fb[NRB];
#pragma unroll
    for (int i = 0; i < MRA; i++) {
      alignas(16) T tmp[EPL];
#pragma unroll
      for (int e = 0; e < EPL; e++) tmp[e] = myD[(q * EPL + e) * PD + i * 16 + li];
      fa[i] = ys_pack_elems(tmp);
    }
#pragma unroll
    for (int j = 0; j < NRB; j++) {
      alignas(16) T tmp[EPL];
#pragma unroll
      for (int e = 0; e < EPL; e++) tmp[e] = myX[(q * EPL + e) * PX + j * 16 + li];
      fb[j] = ys_pack_elems(tmp);
    }
#pragma unroll
    for (int i = 0; i < MRA; i++)
#pragma unroll
      for (int j = 0; j < NRB; j++) acc[i][j] = ys_mma<T>(fa[i], fb[j], acc[i][j]);
    ys_wave_sync();
  }
  __syncthreads();
  // ---- combine the 4 waves, write the partial tile
  float* sR = (float*)smem;  // [4][COT][CIT]
#pragma unroll
  for (int i = 0; i < MRA; i++)
#pragma unroll
    for (int j = 0; j < NRB; j++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        sR[(wave * COT + i * 16 + 4 * q + r) * CIT + j * 16 + li] = acc[i][j][r];
  __syncthreads();
  float* outp = a.partial + (long)blockIdx.x * a.Cout * a.KH * a.KW * a.Cin;
  for (int e = tid; e < COT * CIT; e += 256) {
    const int co = co0 + e / CIT, ci = ci0 + e % CIT;
    if (co < a.Cout && ci < a.Cin) {
      const float v = sR[e] + sR[COT * CIT + e] + sR[2 * COT * CIT + e] + sR[3 * COT * CIT + e];
      outp[((long)co * a.KH * a.KW + tap) * a.Cin + ci] = v;
    }
  }
}

__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ partial, int splits, long n, int cin_pad, int cin_real,
                    float* __restrict__ grad) {
  // grad[(row)*cin_real + ci] += sum_s partial[s][row*cin_pad + ci]   (drops padded input channels)
  // 64 outputs x 4 split lanes per workgroup: the sum over splits is 4 interleaved chains combined in a fixed order
  __shared__ float sred[4][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + o;
  float s = 0.f;
  if (i < n)
    for (int k = sl; k < splits; k += 4) s += partial[(long)k * n + i];
  sred[sl][o] = s;
  __syncthreads();
  if (sl == 0 && i < n) {
    const long row = i / cin_pad;
    const int ci = (int)(i - row * cin_pad);
    if (ci < cin_real) grad[row * cin_real + ci] += (sred[0][o] + sred[1][o]) + (sred[2][o] + sred[3][o]);
  }
}

template <class T, int MRA, int NRB>
static void wgrad_launch_t(hipStream_t st, const WgradArgs& a, int splits) {
  const int co_tiles = ys_cdiv(a.Cout, MRA * 16), ci_tiles = ys_cdiv(a.Cin, NRB * 16);
  dim3 grid(splits, co_tiles * ci_tiles, a.KH * a.KW);
  YsKprofScope prof(st, "conv_wgrad");
  YS_LAUNCH((conv_wgrad_kernel<T, MRA, NRB>), grid, 256, st, a);
}

template <class T>
static void wgrad_dispatch(hipStream_t st, const WgradArgs& a, int splits) {
  const int cof = (a.Cout + 15) / 16, cif = (a.Cin + 15) / 16;
  const int mra = cof >= 4 ? ((cof % 5 == 0) ? 5 : 4) : cof;  // 1,2,3,4,5
  const int nrb = cif >= 4 ? 4 : cif;
#define WG(M_, N_) if (mra == M_ && nrb == N_) { wgrad_launch_t<T, M_, N_>(st, a, splits); return; }
  WG(1, 1) WG(1, 2) WG(1, 3) WG(1, 4)
  WG(2, 1) WG(2, 2) WG(2, 3) WG(2, 4)
  WG(3, 1) WG(3, 2) WG(3, 3) WG(3, 4)
  WG(4, 1) WG(4, 2) WG(4, 3) WG(4, 4)
  WG(5, 1) WG(5, 2) WG(5, 3) WG(5, 4)
#undef WG
}

// number of pixel splits used for a layer (also sizes the partial workspace)
int ys_wgrad_splits(const WgradArgs& a, int dtype) {
  const int ks = dtype == YS_BF16 ? 32 : 16;
  const int cof = (a.Cout + 15) / 16, cif = (a.Cin + 15) / 16;
  const int mra = cof >= 4 ? ((cof % 5 == 0) ? 5 : 4) : cof;
  const int nrb = cif >= 4 ? 4 : cif;
  const long tiles = (long)ys_cdiv(a.Cout, mra * 16) * ys_cdiv(a.Cin, nrb * 16) * a.KH * a.KW;
  const long steps = ((long)a.M + ks - 1) / ks;
  long s = (2048 + tiles - 1) / tiles;          // aim for ~2k workgroups (256 CUs x 8)
  const long smax = (steps + 15) / 16;          // at least 16 wave-steps (4 iterations) per workgroup
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  if (s > 128) s = 128;
  return (int)s;
}

int ys_wgrad_launch(hipStream_t st, int dtype, const WgradArgs& a, int splits, int cin_real, float* grad) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  if (a.Cin % epl || a.in_ldc % epl || a.in_coff % epl || a.dy_ldc % epl || a.dy_coff % epl) {
    ys_set_error("wgrad: channel counts/strides must be multiples of %d (Cin %d Cout %d)", epl, a.Cin, a.Cout);
    return YS_ERR_INVALID_ARG;
  }
  if (dtype == YS_BF16) wgrad_dispatch<bf16_t>(st, a, splits);
  else wgrad_dispatch<float>(st, a, splits);
  const long n = (long)a.Cout * a.KH * a.KW * a.Cin;
  YS_LAUNCH(wgrad_reduce_kernel, ys_cdiv(n, 64), 256, st, (const float*)a.partial, splits, n, a.Cin, cin_real, grad);
  return YS_OK;
}

// ===================================================================================== weight prep
// master fp32 weights [Cout][taps][Cin_real] -> forward weights T [Cout][taps][Cin_pad]
//                                            -> dgrad weights  T [Cin_real][taps flipped][Cout_pad]
template <class T>
__global__ void __launch_bounds__(256)
weight_prep_kernel(const float* __restrict__ w, int Cout, int taps, int cin_real, int cin_pad, int cout_pad,
                   T* __restrict__ wf, T* __restrict__ wd) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nf = (long)Cout * taps * cin_pad;
  if (i < nf) {
    const int ci = (int)(i % cin_pad);
    const long r = i / cin_pad;  // co*taps + tap
    wf[i] = Elem<T>::from_f(ci < cin_real ? w[r * cin_real + ci] : 0.f);
  }
  if (wd) {
    const long nd = (long)cin_real * taps * cout_pad;
    if (i < nd) {
      const int co = (int)(i % cout_pad);
      const long r = i / cout_pad;
      const int tapf = (int)(r % taps);
      const int ci = (int)(r / taps);
      const int tap = taps - 1 - tapf;  // spatial flip of a square kernel
      wd[i] = Elem<T>::from_f(co < Cout ? w[((long)co * taps + tap) * cin_real + ci] : 0.f);
    }
  }
}

int ys_weight_prep_launch(hipStream_t st, int dtype, const float* w, int Cout, int taps, int cin_real, int cin_pad,
                          int cout_pad, void* wf, void* wd) {
  const long nf = (long)Cout * taps * cin_pad;
  const long nd = wd ? (long)cin_real * taps * cout_pad : 0;
  const long n = nf > nd ? nf : nd;
  if (dtype == YS_BF16)
    YS_LAUNCH((weight_prep_kernel<bf16_t>), ys_cdiv(n, 256), 256, st, w, Cout, taps, cin_real, cin_pad, cout_pad, (bf16_t*)wf, (bf16_t*)wd);
  else
    YS_LAUNCH((weight_prep_kernel<float>), ys_cdiv(n, 256), 256, st, w, Cout, taps, cin_real, cin_pad, cout_pad, (float*)wf, (float*)wd);
  return YS_OK;
}
