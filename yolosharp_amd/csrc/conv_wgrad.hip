// conv_wgrad.hip -- weight gradients of the NHWC convolutions (see conv.hip for the forward / dgrad kernels and the layout),
// the fixed-order split reduction, and the weight-shadow preparation kernels.
#include "ys_internal.h"
#include "ys_kernels.h"
#include <atomic>
#include <cstdlib>

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
        const long rr = p - bb * HWo;
        const long drow = a.dy_rh ? bb * a.dy_bstride + (rr / a.Wout) * a.dy_rh + (rr % a.Wout) * a.dy_rw + a.dy_r0
                                  : bb * a.dy_bstride + rr;
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

__global__ void __launch_bounds__(512)
wgrad_reduce_kernel(const float* __restrict__ partial, int splits, long n, int cin_pad, int cin_real,
                    float* __restrict__ grad) {
  // grad[(row)*cin_real + ci] += sum_s partial[s][row*cin_pad + ci]   (drops padded input channels)
  // 32 outputs x 16 split lanes per workgroup; every lane walks its splits (k = sl, sl+16, ...) with four loads in flight,
  // and the 16 lane sums are combined in a fixed order -> deterministic
  __shared__ float sred[16][32];
  const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const long i = (long)blockIdx.x * 32 + o;
  // the read-modify-write of grad is issued up front so its latency overlaps the partial loads (it was a second dependent
  // round trip at the tail of a ~5 us kernel)
  float gprev = 0.f; long gidx = -1;
  if (sl == 0 && i < n) {
    const long row = i / cin_pad;
    const int ci = (int)(i - row * cin_pad);
    if (ci < cin_real) { gidx = row * cin_real + ci; gprev = grad[gidx]; }
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int k = sl;
    for (; k + 48 < splits; k += 64) {
      s0 += partial[(long)k * n + i];
      s1 += partial[(long)(k + 16) * n + i];
      s2 += partial[(long)(k + 32) * n + i];
      s3 += partial[(long)(k + 48) * n + i];
    }
    for (; k < splits; k += 16) s0 += partial[(long)k * n + i];
  }
  sred[sl][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (gidx >= 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; w++) t += sred[w][o];
    grad[gidx] = gprev + t;
  }
}

// ------------------------------------------------------------------ wgrad, bf16: LDS tiles + hardware transpose reads
// One workgroup owns a 2-D tile of TH x TW output pixels (TW a power of two).  It stages the dy tile [pixel][COT] and the
// input PATCH [(TH-1)*S+KH][(TW-1)*S+KW] x [CIT] once in LDS in their natural NHWC row form, and every tap reads its
// operands from that one patch: ds_read_b64_tr_b16 delivers 4 consecutive-K (pixel) values of one channel per lane from
// four freely addressed rows, so the tap shift (and stride 2) is just a different row address -- no per-tap re-read of
// dy / x from L2 (the round-1 kernel re-fetched both for each of the 9 taps) and no 2-byte transposing LDS traffic.
//   TAPS == 9: 9 waves, wave w accumulates tap w over all pixels of every tile the workgroup walks (persistent);
//   TAPS == 1: 4 waves split the 32-pixel K-blocks and are combined through LDS at the end.
// K order inside a tile is the tile-local pixel index p = ty*TW + tx for both operands; pixels outside the image (or the
// tile's 32-pixel rounding) read a zero dy row.  Partials go to partial[blockIdx.x][Cout][taps][Cin] (fixed-order reduce).
#ifndef YS_WG_TWO_MAX
#define YS_WG_TWO_MAX 2        // 9-wave weight-gradient tiles of up to this many MFMA fragments run two workgroups per CU (96 registers).  4 measured (round 4): the 2 x 2 tile spills 32 B and the 32 -> 32 layers go 29.4 -> 42.0 us (400 partial slabs instead of 229) -- stays 2
#endif
template <int MRA, int NRB, int TAPS>
__global__ void __launch_bounds__(TAPS == 9 ? 576 : 256, (TAPS == 9 && MRA * NRB <= YS_WG_TWO_MAX) ? 5 : 1)   // small 9-wave tiles: 96 registers = two workgroups per CU
conv_wgrad_tr_kernel(WgradArgs a) {
  constexpr int NT = TAPS == 9 ? 576 : 256;
  constexpr int NW = NT / 64;
  constexpr int COT = MRA * 16, CIT = NRB * 16;
  constexpr int DV = COT / 8, XV = CIT / 8;            // 16-byte units per pixel row
  constexpr int TPMAX = TAPS == 9 ? 256 : 128;
  constexpr int ND = (TPMAX * DV + NT - 1) / NT;       // dy units fetched per thread
  constexpr int NX = TAPS == 9 ? 5 : (TPMAX * XV + NT - 1) / NT;   // patch units per thread (host keeps PH*PW*XV <= NX*NT)
  YS_DYN_LDS(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
  const int TW = 1 << a.TWS, TP = a.TH * TW;
  const int nkb = (TP + 31) >> 5;
  const int ZR = nkb * 32;                              // zero row of the dy image
  char* sDb = (char*)lds;                               // [ZR + 1][pdb]
  char* sXb = sDb + (size_t)(ZR + 1) * a.pdb;           // [PH*PW][pxb]
  const int ci_tiles = (a.Cin + CIT - 1) / CIT;
  const int co0 = (blockIdx.y / ci_tiles) * COT;
  const int ci0 = (blockIdx.y % ci_tiles) * CIT;
  const char* dyb = (const char*)a.dy;
  const char* xb = (const char*)a.x;
  const int S = a.stride;
  const int npatch = a.PH * a.PW * XV;

  for (int i = tid; i < a.pdb / 4; i += NT) ((unsigned*)(sDb + (size_t)ZR * a.pdb))[i] = 0u;

  // both operands through buffer descriptors: every unit's load is unconditional, padding / unused units carry the out-of-range
  // offset and arrive as zeros (the exec-masked `if (valid) load` form made the compiler wait vmcnt(0) at the join)
  const ys_rsrcv_t rsD = ys_make_rsrcv(dyb + (long)a.dy_coff * 2L, a.dybytes);
  const ys_rsrcv_t rsX = ys_make_rsrcv(xb + (long)a.in_coff * 2L, a.xbytes);
  uint4 rd[ND], rx[NX];
  auto fetch = [&](int tile) {
    int t = tile;
    const int txi = t % a.tiles_x; t /= a.tiles_x;
    const int tyi = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int oy0 = tyi * a.TH, ox0 = txi * TW;
#pragma unroll
    for (int k = 0; k < ND; k++) {
      const int idx = tid + NT * k;
      const int p = idx / DV, u = idx - p * DV;
      const int oy = oy0 + (p >> a.TWS), ox = ox0 + (p & (TW - 1));
      const int c = co0 + u * 8;
      const bool ok = (bool)((int)(idx < TP * DV) & (int)(oy < a.Hout) & (int)(ox < a.Wout) & (int)(c < a.Cout));
      rd[k] = ys_bufld16(rsD, ok ? (unsigned)(((((long)b * a.dy_bstride + (long)oy * a.Wout + ox) * a.dy_ldc) + c) * 2L) : YS_BUF_OOB);
    }
    const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
#pragma unroll
    for (int k = 0; k < NX; k++) {
      const int idx = tid + NT * k;
      const int pix = idx / XV, u = idx - pix * XV;
      const int r = pix / a.PW, cc = pix - r * a.PW;
      const int iy = iy0 + r, ix = ix0 + cc;
      const int c = ci0 + u * 8;
      const bool ok = (bool)((int)(idx < npatch) & (int)((unsigned)iy < (unsigned)a.Hin) & (int)((unsigned)ix < (unsigned)a.Win) & (int)(c < a.Cin));
      rx[k] = ys_bufld16(rsX, ok ? (unsigned)(((((long)b * a.in_bstride + (long)iy * a.Win + ix) * a.in_ldc) + c) * 2L) : YS_BUF_OOB);
    }
  };

  f32x4 acc[MRA][NRB];
#pragma unroll
  for (int i = 0; i < MRA; i++)
#pragma unroll
    for (int j = 0; j < NRB; j++) acc[i][j] = f32x4_zero();

  const int tap = TAPS == 9 ? wave : 0;
  const int kh = tap / 3, kw = tap - kh * 3;
  const int rr = li >> 2, c4 = li & 3;                  // this lane's row / 4-column group inside a [4][16] block
  // K order: MFMA k = 8q + 4h + j  <->  tile pixel p = 32*kb + 16h + 4q + j (any bijection works as long as both operands
  // use it); this one makes the 32 lanes the LDS services together (q, q+1) read 8 consecutive rows -> distinct banks with
  // the odd-slot row pitch.  Per lane and h the pixel advances by 32 per K-block: constant LDS strides, no per-step divides.
  unsigned dof[2], xof[2];
  int ty0[2], tx0[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int p0 = 16 * h + 4 * q + rr;
    ty0[h] = p0 >> a.TWS; tx0[h] = p0 & (TW - 1);
    dof[h] = (unsigned)(p0 * a.pdb + c4 * 8);
    xof[h] = (unsigned)((size_t)(ZR + 1) * a.pdb) + (unsigned)(((ty0[h] * S + kh) * a.PW + tx0[h] * S + kw) * a.pxb + c4 * 8);
  }
  const unsigned dstep = 32u * (unsigned)a.pdb;                          // 32 pixels further down the dy image
  const unsigned xstep = (unsigned)((32 >> a.TWS) * S * a.PW * a.pxb);   // 32/TW tile rows further down the patch
  const unsigned zof = (unsigned)(ZR * a.pdb + c4 * 8);
  const unsigned xzero = (unsigned)((ZR + 1) * a.pdb + c4 * 8);
  const int rows_kb = 32 >> a.TWS;
  const char* lb = (const char*)lds;

  int tile = blockIdx.x;
  if (tile < a.ntiles) fetch(tile);
  while (tile < a.ntiles) {
    ys_barrier_lds();                                   // every wave finished reading the previous tile
#pragma unroll
    for (int k = 0; k < ND; k++) {
      const int idx = tid + NT * k;
      if (idx < TP * DV) { const int p = idx / DV; *(uint4*)(sDb + (size_t)p * a.pdb + (idx - p * DV) * 16) = rd[k]; }
    }
#pragma unroll
    for (int k = 0; k < NX; k++) {
      const int idx = tid + NT * k;
      if (idx < npatch) { const int pix = idx / XV; *(uint4*)(sXb + (size_t)pix * a.pxb + (idx - pix * XV) * 16) = rx[k]; }
    }
    ys_barrier_lds();
    int t = tile;
    const int txi = t % a.tiles_x; t /= a.tiles_x;
    const int tyi = t % a.tiles_y;
    const int oy0 = tyi * a.TH, ox0 = txi * TW;
    const int ntile = tile + gridDim.x;
    if (ntile < a.ntiles) fetch(ntile);                 // in flight while this tile is consumed
    // rows of the tile that hold image pixels (tile-local), and this lane's column validity
    const int tylim = (a.Hout - oy0) < a.TH ? (a.Hout - oy0) : a.TH;
    const bool colok0 = (ox0 + tx0[0]) < a.Wout, colok1 = (ox0 + tx0[1]) < a.Wout;
    for (int kb = (TAPS == 9 ? 0 : wave); kb < nkb; kb += (TAPS == 9 ? 1 : NW)) {
      // the transpose reads land asynchronously: their destination registers must not be touched before ys_lds_tr_wait
      uint2 ra[2][MRA], rb[2][NRB];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int ty = ty0[h] + kb * rows_kb;
        const bool ok = (h == 0 ? colok0 : colok1) && ty < tylim;
        const unsigned d = ok ? dof[h] + (unsigned)kb * dstep : zof;
        const unsigned x = ty < a.TH ? xof[h] + (unsigned)kb * xstep : xzero;   // rounding pixels: any initialised row
#pragma unroll
        for (int i = 0; i < MRA; i++) ra[h][i] = ys_lds_tr_b64(lb + d + i * 32);
#pragma unroll
        for (int j = 0; j < NRB; j++) rb[h][j] = ys_lds_tr_b64(lb + x + j * 32);
      }
#pragma unroll
      for (int i = 0; i < MRA; i++) ys_lds_tr_wait(ra[0][i], ra[1][i]);
#pragma unroll
      for (int j = 0; j < NRB; j++) ys_lds_tr_wait(rb[0][j], rb[1][j]);
      uint4 fa[MRA], fb[NRB];
#pragma unroll
      for (int i = 0; i < MRA; i++) fa[i] = make_uint4(ra[0][i].x, ra[0][i].y, ra[1][i].x, ra[1][i].y);
#pragma unroll
      for (int j = 0; j < NRB; j++) fb[j] = make_uint4(rb[0][j].x, rb[0][j].y, rb[1][j].x, rb[1][j].y);
#pragma unroll
      for (int i = 0; i < MRA; i++)
#pragma unroll
        for (int j = 0; j < NRB; j++) acc[i][j] = ys_mma<bf16_t>(fa[i], fb[j], acc[i][j]);
    }
    tile = ntile;
  }

  float* outp = a.partial + (long)blockIdx.x * a.Cout * a.KH * a.KW * a.Cin;
  const int taps = a.KH * a.KW;
  if (TAPS == 9) {
#pragma unroll
    for (int i = 0; i < MRA; i++)
#pragma unroll
      for (int j = 0; j < NRB; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int co = co0 + i * 16 + 4 * q + r, ci = ci0 + j * 16 + li;
          if (co < a.Cout && ci < a.Cin) outp[((long)co * taps + tap) * a.Cin + ci] = acc[i][j][r];
        }
  } else {
    __syncthreads();
    float* sR = (float*)lds;                            // [NW][COT][CIT]
#pragma unroll
    for (int i = 0; i < MRA; i++)
#pragma unroll
      for (int j = 0; j < NRB; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) sR[(wave * COT + i * 16 + 4 * q + r) * CIT + j * 16 + li] = acc[i][j][r];
    __syncthreads();
    for (int e = tid; e < COT * CIT; e += NT) {
      const int co = co0 + e / CIT, ci = ci0 + e % CIT;
      if (co < a.Cout && ci < a.Cin) {
        float v = 0.f;
        for (int w = 0; w < NW; w++) v += sR[w * COT * CIT + e];
        outp[(long)co * a.Cin + ci] = v;
      }
    }
  }
}

// host-side plan of the LDS-tile wgrad kernel
struct WgPlan { int ok, taps9, mra, nrb, th, tws, tx, ty, ph, pw, pdb, pxb, gx; size_t lds; };
static int wg_pitch_bytes(int ch) {       // smallest row pitch >= ch*2 bytes whose 32-byte slot count is odd (8 rows -> 8 bank slots)
  int s = (ch * 2 + 31) / 32;
  if (!(s & 1)) s++;
  return s * 32;
}
// descriptor ranges of the dy / x views (bytes from the first channel to the end of the last image); false if either is >= 2 GB
static bool wgrad_view_bytes(const WgradArgs& a, unsigned& dyb, unsigned& xb) {
  const long dpix = (long)(a.B - 1) * a.dy_bstride + (long)a.Hout * a.Wout, xpix = (long)(a.B - 1) * a.in_bstride + (long)a.Hin * a.Win;
  const long d = (dpix * a.dy_ldc - a.dy_coff) * 2L, x = (xpix * a.in_ldc - a.in_coff) * 2L;
  if (d <= 0 || x <= 0 || d >= (1L << 31) || x >= (1L << 31)) return false;
  dyb = (unsigned)d; xb = (unsigned)x;
  return true;
}
// plan sweep (tools/dev/r06/wg_sweep.py through ys_debug_wgrad_force): tile height / log2 width / workgroups per CU imposed on every plan of the process; 0 = the
// cost model's own choice.  Triage only.
static int g_wg_force_th = 0, g_wg_force_tws = 0, g_wg_force_percu = 0;
extern "C" __attribute__((visibility("default"))) int ys_debug_wgrad_force(int th, int tws, int per_cu) { g_wg_force_th = th; g_wg_force_tws = tws; g_wg_force_percu = per_cu; return 0; }
static WgPlan wgrad_tr_plan(const WgradArgs& a) {
  WgPlan p{};
  { unsigned d0, x0; if (!wgrad_view_bytes(a, d0, x0)) return p; }
  const bool k3 = a.KH == 3 && a.KW == 3 && a.pad == 1 && (a.stride == 1 || a.stride == 2);
  const bool k1 = a.KH == 1 && a.KW == 1 && a.pad == 0 && a.stride == 1;
  if (a.dy_rh || !(k3 || k1)) return p;
  const int cof = (a.Cout + 15) / 16, cif = (a.Cin + 15) / 16;
  p.mra = cof >= 4 ? ((cof % 5 == 0) ? 5 : 4) : cof;
  p.nrb = cif >= 4 ? 4 : cif;
  p.taps9 = k3 ? 1 : 0;
  if (k3) {
    // the 9-wave kernel has 168 registers per lane (3 waves/SIMD): fragment tiles that fit without spilling, cheapest re-read
    static const int ok_pairs[][2] = {{1, 1}, {1, 2}, {1, 3}, {1, 4}, {2, 1}, {2, 2}, {2, 3}, {2, 4}, {3, 1}, {3, 2}, {3, 3}, {3, 4}, {4, 1}, {4, 2}, {5, 1}};
    int best_cost = 1 << 30;
    for (auto& pr : ok_pairs) {
      const int m = pr[0], n = pr[1];
      if (m > cof || n > cif) continue;
      const int cost = ys_cdiv(cof, m) * ys_cdiv(cif, n) * (m + n) * 16 + (ys_cdiv(cof, m) * m - cof) + (ys_cdiv(cif, n) * n - cif) + (n > m ? 1 : 0);
      if (cost < best_cost) { best_cost = cost; p.mra = m; p.nrb = n; }
    }
  }
  const int nt = k3 ? 576 : 256, tpmax = k3 ? 256 : 128;
  const int cot = p.mra * 16, cit = p.nrb * 16, xv = cit / 8;
  const int nxu = (k3 ? 5 : (tpmax * xv + nt - 1) / nt) * nt;
  const int S = a.stride;
  p.pdb = wg_pitch_bytes(cot);
  // stride-2 patches are read every other row: a pitch of 16 (mod 32) bytes makes the 2-row step an odd slot count
  p.pxb = S == 1 ? wg_pitch_bytes(cit) : ((cit * 2 + 15) / 32) * 32 + 16;
  double best = 1e30;
  for (int tws = 2; tws <= 5; tws++) {
    const int tw = 1 << tws;
    if (tw > a.Wout && tws > 2 && (tw >> 1) >= a.Wout) continue;
    if (g_wg_force_tws && tws != g_wg_force_tws) continue;
    for (int th = 1; th <= a.Hout && th * tw <= tpmax; th++) {
      if (g_wg_force_th && th != g_wg_force_th) continue;
      const int ph = (th - 1) * S + a.KH, pw = (tw - 1) * S + a.KW;
      if (ph * pw * xv > nxu) continue;
      const int nkb = (th * tw + 31) / 32;
      size_t lds = (size_t)(nkb * 32 + 1) * p.pdb + (size_t)ph * pw * p.pxb;
      if (!k3 && lds < (size_t)4 * cot * cit * 4) lds = (size_t)4 * cot * cit * 4;
      if (lds > 150 * 1024) continue;
      const int tx = ys_cdiv(a.Wout, tw), ty = ys_cdiv(a.Hout, th);
      const double cost = (double)tx * ty * ((double)ph * pw * cit + (double)nkb * 32 * cot + 0.5 * nkb * 32 * (cot + cit) + 3000.0);
      if (cost < best) { best = cost; p.ok = 1; p.th = th; p.tws = tws; p.tx = tx; p.ty = ty; p.ph = ph; p.pw = pw; p.lds = lds; }
    }
  }
  if (!p.ok) return p;
  const int gy = ys_cdiv(a.Cout, cot) * ys_cdiv(a.Cin, cit);
  const long ntiles = (long)p.tx * p.ty * a.B;
  int per_cu = (int)((150 * 1024) / p.lds);
  const int per_cu_max = k3 ? (p.mra * p.nrb <= YS_WG_TWO_MAX ? 2 : 1) : 4;   // 9-wave workgroups are register-limited to 1-2 per CU (launch bounds of conv_wgrad_tr_kernel)
  if (per_cu > per_cu_max) per_cu = per_cu_max;
  if (g_wg_force_percu && g_wg_force_percu < per_cu) per_cu = g_wg_force_percu;
  if (per_cu < 1) per_cu = 1;
  long gx = ((long)ys_cu_count() * per_cu) / gy;          // (round 6: 1.5x / 2x / 3x the resident slots -- shorter-lived workgroups, so that the main stream's kernels find free slots sooner -- measured 8.78 -> 8.89 / 8.99 / 9.27 ms on config 2)   (round 5: half / quarter grids -- fewer partial slabs for the split reduction -- measured 8.69 -> 8.73 / 10.01 ms on config 2)
  gx = gx / 8 * 8;                                       // same-x workgroups (same pixels, other channel tiles) share an XCD
  if (gx < 8) gx = 8;
  const long wsmax = (48L << 20) / ((long)a.Cout * a.KH * a.KW * a.Cin * 4);   // bound the partial workspace to 48 MB per layer
  if (gx > wsmax) gx = wsmax > 0 ? wsmax : 1;
  if (gx > ntiles) gx = ntiles;
  const long per = (ntiles + gx - 1) / gx;                // equal tile counts: no idle tail workgroups, fewer partials
  gx = (ntiles + per - 1) / per;
  p.gx = (int)gx;
  return p;
}

template <int MRA, int NRB, int TAPS>
static void wgrad_tr_launch_t(hipStream_t st, WgradArgs a, const WgPlan& p) {
  a.TH = p.th; a.TWS = p.tws; a.tiles_x = p.tx; a.tiles_y = p.ty; a.ntiles = p.tx * p.ty * a.B;
  wgrad_view_bytes(a, a.dybytes, a.xbytes);
  a.PH = p.ph; a.PW = p.pw; a.pdb = p.pdb; a.pxb = p.pxb;
  const int gy = ys_cdiv(a.Cout, MRA * 16) * ys_cdiv(a.Cin, NRB * 16);
  static std::atomic<unsigned> attr_done{0};      // per device: the attribute belongs to the device's code object
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  if (!(attr_done.load(std::memory_order_relaxed) & (1u << (dev_id & 31)))) {
    hipFuncSetAttribute((const void*)conv_wgrad_tr_kernel<MRA, NRB, TAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.fetch_or(1u << (dev_id & 31), std::memory_order_relaxed);
  }
  char lab[160] = "";
  if (ys_kprof_enabled()) snprintf(lab, sizeof(lab), "wgrad_tr k%d s%d cin%d cout%d M%d tile%dx%d grid%dx%d lds%d", a.KH, a.stride, a.Cin, a.Cout, a.M, p.th, 1 << p.tws, p.gx, gy, (int)p.lds);
  YsKprofScope prof(st, "conv_wgrad", lab);
  YS_LAUNCH_LDS((conv_wgrad_tr_kernel<MRA, NRB, TAPS>), dim3(p.gx, gy), (TAPS == 9 ? 576 : 256), p.lds, st, a);
}

static bool wgrad_tr_dispatch(hipStream_t st, const WgradArgs& a, const WgPlan& p) {
#define WT(M_, N_) if (p.mra == M_ && p.nrb == N_) { if (p.taps9) wgrad_tr_launch_t<M_, N_, 9>(st, a, p); else wgrad_tr_launch_t<M_, N_, 1>(st, a, p); return true; }
  WT(1, 1) WT(1, 2) WT(1, 3) WT(1, 4)
  WT(2, 1) WT(2, 2) WT(2, 3) WT(2, 4)
  WT(3, 1) WT(3, 2) WT(3, 3) WT(3, 4)
  WT(4, 1) WT(4, 2) WT(4, 3) WT(4, 4)
  WT(5, 1) WT(5, 2) WT(5, 3) WT(5, 4)
#undef WT
  return false;
}

template <class T, int MRA, int NRB>
static void wgrad_launch_t(hipStream_t st, const WgradArgs& a, int splits) {
  const int co_tiles = ys_cdiv(a.Cout, MRA * 16), ci_tiles = ys_cdiv(a.Cin, NRB * 16);
  dim3 grid(splits, co_tiles * ci_tiles, a.KH * a.KW);
  char lab[128] = "";
  if (ys_kprof_enabled()) snprintf(lab, sizeof(lab), "wgrad k%d s%d cin%d cout%d M%d splits%d tiles%d", a.KH, a.stride, a.Cin, a.Cout, a.M, splits, co_tiles * ci_tiles);
  YsKprofScope prof(st, "conv_wgrad", lab);
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
  if (dtype == YS_BF16) {
    const int gs = ys_wgrad_gemm_splits(a);   // wide layers: blocked-GEMM kernel (conv_wgrad_gemm.hip)
    if (gs) return gs;
    const WgPlan p = wgrad_tr_plan(a);
    if (p.ok) return p.gx;
  }
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

int ys_wgrad_launch(hipStream_t st, int dtype, const WgradArgs& a, int splits, int cin_real, float* grad, int* used_splits) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  if (a.Cin % epl || a.in_ldc % epl || a.in_coff % epl || a.dy_ldc % epl || a.dy_coff % epl) {
    ys_set_error("wgrad: channel counts/strides must be multiples of %d (Cin %d Cout %d)", epl, a.Cin, a.Cout);
    return YS_ERR_INVALID_ARG;
  }
  bool done = false;
  if (dtype == YS_BF16) {
    const int used = ys_wgrad_gemm_launch(st, a, splits);
    if (used) { splits = used; done = true; }
  }
  if (dtype == YS_BF16 && !done) {
    WgPlan p = wgrad_tr_plan(a);
    if (p.ok) {
      if (splits < p.gx) p.gx = splits;     // never exceed the caller's partial workspace
      splits = p.gx;
      done = wgrad_tr_dispatch(st, a, p);
    }
  }
  if (!done) {
    if (dtype == YS_BF16) wgrad_dispatch<bf16_t>(st, a, splits);
    else wgrad_dispatch<float>(st, a, splits);
  }
  const long n = (long)a.Cout * a.KH * a.KW * a.Cin;
  if (used_splits) { *used_splits = splits; return YS_OK; }
  YS_LAUNCH(wgrad_reduce_kernel, ys_cdiv(n, 32), 512, st, (const float*)a.partial, splits, n, a.Cin, cin_real, grad);
  return YS_OK;
}

// The same reduction for a list of layers in ONE launch (the per-layer form was 63 launches of ~6 us + a kernel boundary each per
// YOLOv8n step, none of them on the dependency chain of the backward pass -- the optimizer is the only reader).  A workgroup finds
// its layer by binary search over the block prefix; summation order per output is the per-layer kernel's.
__global__ void __launch_bounds__(512)
wgrad_reduce_batched_kernel(const WgRedDesc* __restrict__ descs, int nd) {
  // a thread owns FOUR consecutive outputs (16-byte loads of the partial rows: the 4-byte form moved 128 B per half-wave request
  // and ran at ~0.9 TB/s, 0.30 ms per YOLOv8n step); every output keeps the per-layer kernel's summation order, component by
  // component.  n and cin_pad are multiples of 4 (channel padding), so the four share one weight row.
  // Round 5: a workgroup covers NSUB groups of 128 outputs instead of one, the loads of all of them requested before the first is reduced.  With one group a
  // workgroup was 8 dependent descriptor loads, ONE 16-byte load per thread, a barrier and 128 stores: 530 thousand such workgroups per YOLOv8x step, 2.4 TB/s
  // (1.64 ms; 0.37 ms = 3.6 % of the YOLOv8s step).  Same sums in the same order per output.
  constexpr int NSUB = YS_WGRED_OUT_PER_BLOCK / 128;
  __shared__ float4 sred[NSUB][16][32];
  int lo = 0, hi = nd - 1;
  const long blk = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (descs[mid].blk0 <= blk) lo = mid; else hi = mid - 1; }
  const WgRedDesc d = descs[lo];
  const float* __restrict__ partial = d.partial;
  const int splits = d.splits;
  const long n = d.n;
  const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const long i0 = (blk - d.blk0) * YS_WGRED_OUT_PER_BLOCK + o * 4;
  auto add4 = [](float4& acc, const float4& v) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; };
  float4 tot[NSUB];
  if (splits <= 16) {          // the common case: one partial row per split lane -> NSUB independent loads in flight per thread
    float4 v[NSUB];
#pragma unroll
    for (int u = 0; u < NSUB; u++) {
      const long i = i0 + u * 128;
      v[u] = (i < n && sl < splits) ? *(const float4*)(partial + (long)sl * n + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < NSUB; u++) tot[u] = make_float4((v[u].x + 0.f) + (0.f + 0.f), (v[u].y + 0.f) + (0.f + 0.f), (v[u].z + 0.f) + (0.f + 0.f), (v[u].w + 0.f) + (0.f + 0.f));
  } else {
#pragma unroll
    for (int u = 0; u < NSUB; u++) {
      const long i = i0 + u * 128;
      float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
      if (i < n) {
        int k = sl;
        for (; k + 48 < splits; k += 64) {
          const float4 v0 = *(const float4*)(partial + (long)k * n + i);
          const float4 v1 = *(const float4*)(partial + (long)(k + 16) * n + i);
          const float4 v2 = *(const float4*)(partial + (long)(k + 32) * n + i);
          const float4 v3 = *(const float4*)(partial + (long)(k + 48) * n + i);
          add4(s0, v0); add4(s1, v1); add4(s2, v2); add4(s3, v3);
        }
        for (; k < splits; k += 16) add4(s0, *(const float4*)(partial + (long)k * n + i));
      }
      tot[u] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
    }
  }
#pragma unroll
  for (int u = 0; u < NSUB; u++) sred[u][sl][o] = tot[u];
  __syncthreads();
  // split lane u finishes group u: 32 threads x 4 outputs each
  if (sl < NSUB) {
    const long i = i0 + sl * 128;
    if (i < n) {
      const long row = i / d.cin_pad;
      const int ci = (int)(i - row * d.cin_pad);
      int nreal = d.cin_real - ci; if (nreal > 4) nreal = 4;
      if (nreal > 0) {
        const long gidx = row * d.cin_real + ci;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 16; w++) add4(t, sred[sl][w][o]);
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; e++) if (e < nreal) d.grad[gidx + e] = d.grad[gidx + e] + tv[e];
      }
    }
  }
}
int ys_wgrad_reduce_batched_launch(hipStream_t st, const WgRedDesc* descs_dev, int n_desc, long total_blocks) {
  if (n_desc <= 0 || total_blocks <= 0) return YS_OK;
  YS_LAUNCH(wgrad_reduce_batched_kernel, (int)total_blocks, 512, st, descs_dev, n_desc);
  return YS_OK;
}

// ===================================================================================== weight prep
// master fp32 weights [Cout][taps][Cin_real] -> forward weights T [Cout][taps][Cin_pad]
//                                            -> dgrad weights  T [Cin_real][taps flipped][Cout_pad]
template <class T>
__global__ void __launch_bounds__(256)
weight_prep_kernel(const float* __restrict__ w, int Cout, int taps, int cin_real, int cin_pad, int cout_pad,
                   T* __restrict__ wf, T* __restrict__ wd, int phase) {
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
      int co, ci, tap;
      if (phase) {
        ys_phase_wd_index(i, cin_real, cout_pad, ci, tap, co);
      } else {
        co = (int)(i % cout_pad);
        const long r = i / cout_pad;
        const int tapf = (int)(r % taps);
        ci = (int)(r / taps);
        tap = taps - 1 - tapf;  // spatial flip of a square kernel
      }
      wd[i] = Elem<T>::from_f(co < Cout ? w[((long)co * taps + tap) * cin_real + ci] : 0.f);
    }
  }
}

int ys_weight_prep_launch(hipStream_t st, int dtype, const float* w, int Cout, int taps, int cin_real, int cin_pad,
                          int cout_pad, void* wf, void* wd, int phase) {
  const long nf = (long)Cout * taps * cin_pad;
  const long nd = wd ? (long)cin_real * taps * cout_pad : 0;
  const long n = nf > nd ? nf : nd;
  if (dtype == YS_BF16)
    YS_LAUNCH((weight_prep_kernel<bf16_t>), ys_cdiv(n, 256), 256, st, w, Cout, taps, cin_real, cin_pad, cout_pad, (bf16_t*)wf, (bf16_t*)wd, phase);
  else
    YS_LAUNCH((weight_prep_kernel<float>), ys_cdiv(n, 256), 256, st, w, Cout, taps, cin_real, cin_pad, cout_pad, (float*)wf, (float*)wd, phase);
  return YS_OK;
}
