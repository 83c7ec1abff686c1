// conv_wgrad_gemm.hip -- weight gradient of the wide bf16 convolutions as a blocked GEMM, gfx950 (CDNA4).
//
//   dW[co][tap][ci] = sum_m dy[m][co] * x[pix(m, tap)][ci]                 (autograd of Convs.cs:36-62 / Amp.cs:348,370)
//
// The contraction runs over pixels m, the slow axis of both NHWC operands.  conv_wgrad_tr_kernel (conv_wgrad.hip) stages a 2-D
// pixel tile with its halo once and gives each of 9 waves one tap; that is the right trade for <= 128 channels, but its
// register tiles are small (<= 12 MFMAs per 14 transposing reads), one 9-wave workgroup fills a CU, and at 160-640 channels it
// sits at 380 TFLOP/s.  This kernel is the GEMM form for those layers:
//
//  * one workgroup = one (cout tile, cin tile, tap) output block of 160x160 or 128x128 (or mixed) channels and one contiguous share of
//    the pixels; 4 waves as 2 x 2, each MR x NR MFMA 16x16x32 tiles (25 MFMAs per 20 transposing reads at 160x160);
//  * K-tiles of KT = 64 (or 32) pixels: dy rows [pixel][cout tile] and x rows [tap-shifted pixel][cin tile] go global -> LDS by
//    LDS DMA, two stages, one barrier per K-tile.  Four lanes fetch one pixel row, 64 B per DMA instruction and row, so the LDS
//    image of an operand is [64-byte chunk][16-row block][row][64 B]; the tap shift, the stride and the zero padding are
//    per-lane source offsets into buffer descriptors (an out-of-range offset for pixels outside the image: hardware zeros);
//  * fragments by ds_read_b64_tr_b16 (4 consecutive pixels of one channel per lane).  The 32 lanes the LDS serves together read 8
//    consecutive rows x 32 B; rows are 64 B apart inside a block, so the two 32-byte halves of a chunk are swapped on rows 4-7
//    of every 8 (source address of the DMA and read address alike) -> conflict-free;
//  * workgroups are numbered so that the blocks working on the same pixels (other taps / channel tiles) run on the same XCD at
//    the same time: dy and x come from HBM once per XCD and from L2 for the other 8 .. 35 blocks;
//  * output: fp32 partial[split][Cout][taps][Cin], summed in a fixed order by wgrad_reduce_kernel (deterministic; same contract
//    as the other wgrad kernels).
#include "ys_internal.h"
#include "ys_kernels.h"
#include <atomic>
#include <cstdlib>

struct WgGemmArgs {
  int gy;           // (cout tile, cin tile, tap) blocks
  int co_tiles, ci_tiles, taps;
  int nkt;          // K-tiles of KT pixels
  int per;          // K-tiles per pixel split
  int HoWo;
  float inv_howo, inv_wo;
  unsigned dybytes, xbytes;   // bytes of the dy / x views from their first channel to the end of the last image (descriptor ranges)
};

// floor(n / d) for 0 <= n < 2^24 with a precomputed float reciprocal (one multiply + a +-1 fix-up instead of a 32-bit division)
__device__ inline int ys_div24(int n, int d, float inv) {
  int qv = (int)((float)n * inv);
  const int r = n - qv * d;
  qv += (r >= d) ? 1 : 0;
  qv -= (r < 0) ? 1 : 0;
  return qv;
}

template <int MR, int NR, int KT>
__global__ void __launch_bounds__(256, 2)
conv_wgrad_gemm_kernel(WgradArgs a, WgGemmArgs g) {
  typedef bf16_t T;
  constexpr int BCO = 2 * MR * 16, BCI = 2 * NR * 16;
  constexpr int CD = BCO / 32, CX = BCI / 32;          // 64-byte chunks per row
  constexpr int RW = KT / 16;                          // 16-row blocks per K-tile
  constexpr int DY_BYTES = KT * BCO * 2, X_BYTES = KT * BCI * 2, STAGE = DY_BYTES + X_BYTES;
  static_assert(KT == 32 || KT == 64, "K-tile");
  YS_DYN_LDS(lds);
  char* lb = (char*)lds;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, q = lane >> 4, rr = li >> 2, c4 = li & 3;
#ifdef YS_EMU_BUILD
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int wm = wave >> 1, wn = wave & 1;

  // block -> (pixel split, cout tile, cin tile, tap).  Workgroup i runs on XCD i % 8; renumbering so that each XCD owns a contiguous
  // range of the split-major list keeps the gy blocks of one split (same dy / x rows) on one XCD (bijective for any grid size).
  int item;
  {
    const int nwg = (int)gridDim.x, orig = (int)blockIdx.x, xcd = orig & 7, qn = nwg >> 3, rn = nwg & 7;
    item = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (orig >> 3);
  }
  const int split = item / g.gy, yb = item - split * g.gy;
  const int tap = yb % g.taps, tt = yb / g.taps;
  const int ci_t = tt % g.ci_tiles, co_t = tt / g.ci_tiles;
  const int co0 = co_t * BCO, ci0 = ci_t * BCI;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int k_begin = split * g.per;
  const int k_end = k_begin + g.per < g.nkt ? k_begin + g.per : g.nkt;

  // loader roles: four lanes per pixel row.  KT = 64: every thread loads its row of both operands; KT = 32: waves 0-1 load dy,
  // waves 2-3 load x.
  const int ldrow = KT == 64 ? (tid >> 2) : ((tid & 127) >> 2);
  const bool ld_dy = KT == 64 || wave < 2, ld_x = KT == 64 || wave >= 2;
  const int rblk = KT == 64 ? wave : (wave & 1);       // 16-row block of this wave's rows
  const int sl = (tid & 3) ^ (((ldrow >> 2) & 1) << 1);   // source 16-byte unit inside a 64-byte chunk (halves swapped on rows 4-7 of 8)
  // requests through buffer descriptors (ys_bufld_lds16): per K-tile a thread computes ONE 32-bit row offset per operand; the 64-byte
  // chunks of the row are scalar offsets, a padded row (outside the image / past M) is the out-of-range offset -> zeros.  Channels
  // past Cout / Cin of a ragged last tile read whatever follows in the row (finite activations) or zeros past the buffer end: those
  // output rows / columns are never stored.
  const ys_rsrc_t rsD = ys_make_rsrc((const char*)a.dy + ((long)a.dy_coff + co0) * 2L, g.dybytes > (unsigned)co0 * 2u ? g.dybytes - (unsigned)co0 * 2u : 0u);
  const ys_rsrc_t rsX = ys_make_rsrc((const char*)a.x + ((long)a.in_coff + ci0) * 2L, g.xbytes > (unsigned)ci0 * 2u ? g.xbytes - (unsigned)ci0 * 2u : 0u);

  auto issue = [&](int st, int kt) {
    char* sb = lb + st * STAGE;
    const int m = kt * KT + ldrow;
    const bool mok = m < a.M;
    const int mm = mok ? m : 0;
    const int b = ys_div24(mm, g.HoWo, g.inv_howo), rem = mm - b * g.HoWo;
    const int oy = ys_div24(rem, a.Wout, g.inv_wo), ox = rem - oy * a.Wout;
    if (ld_dy) {
      // dy_rh != 0: strided row map (the phases of ConvTranspose2d(2, 2): dy lives on the 2x upsampled grid)
      const long drow = (long)b * a.dy_bstride + (a.dy_rh ? (long)oy * a.dy_rh + (long)ox * a.dy_rw + a.dy_r0 : (long)oy * a.Wout + ox);
      const unsigned vo = mok ? (unsigned)((drow * a.dy_ldc) * 2L) + (unsigned)sl * 16u : YS_BUF_OOB;
#pragma unroll
      for (int i = 0; i < CD; i++) ys_bufld_lds16(rsD, vo, (unsigned)i * 64u, sb + (i * RW + rblk) * 1024);
    }
    if (ld_x) {
      const int iy = oy * a.stride + kh - a.pad, ix = ox * a.stride + kw - a.pad;
      const bool pok = (bool)((int)mok & (int)((unsigned)iy < (unsigned)a.Hin) & (int)((unsigned)ix < (unsigned)a.Win));
      const unsigned vo = pok ? (unsigned)((((long)b * a.in_bstride + (long)iy * a.Win + ix) * a.in_ldc) * 2L) + (unsigned)sl * 16u : YS_BUF_OOB;
#pragma unroll
      for (int i = 0; i < CX; i++) ys_bufld_lds16(rsX, vo, (unsigned)i * 64u, sb + DY_BYTES + (i * RW + rblk) * 1024);
    }
  };

  // transposing fragment reads: MFMA k = 8q + 4h + j  <->  K-tile pixel p = 32 kb + 16 h + 4q + j (row block 2 kb + h, row 4q + rr)
  int dof[MR], xof[NR];
  {
    const int lc = (4 * q + rr) * 64 + c4 * 8;
#pragma unroll
    for (int i = 0; i < MR; i++) { const int f = wm * MR + i; dof[i] = (f >> 1) * RW * 1024 + lc + (((f & 1) ^ (q & 1)) * 32); }
#pragma unroll
    for (int j = 0; j < NR; j++) { const int f = wn * NR + j; xof[j] = DY_BYTES + (f >> 1) * RW * 1024 + lc + (((f & 1) ^ (q & 1)) * 32); }
  }

  f32x4 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; i++)
#pragma unroll
    for (int j = 0; j < NR; j++) acc[i][j] = f32x4_zero();

  if (k_begin < k_end) issue(0, k_begin);
#pragma unroll 1
  for (int kt = k_begin; kt < k_end; kt++) {
    const int st = (kt - k_begin) & 1;
    YS_WAIT_VM0();                            // this wave's DMA pieces of tile kt have landed ...
    ys_barrier_lds();                         // ... everybody's have, and everybody is done reading the other stage
    if (kt + 1 < k_end) issue(st ^ 1, kt + 1);
    const char* sb = lb + st * STAGE;
#pragma unroll
    for (int kb = 0; kb < KT / 32; kb++) {
      uint2 ra[2][MR], rb[2][NR];
#pragma unroll
      for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int i = 0; i < MR; i++) ra[h][i] = ys_lds_tr_b64(sb + dof[i] + (2 * kb + h) * 1024);
#pragma unroll
        for (int j = 0; j < NR; j++) rb[h][j] = ys_lds_tr_b64(sb + xof[j] + (2 * kb + h) * 1024);
      }
#pragma unroll
      for (int i = 0; i < MR; i++) ys_lds_tr_wait(ra[0][i], ra[1][i]);
#pragma unroll
      for (int j = 0; j < NR; j++) ys_lds_tr_wait(rb[0][j], rb[1][j]);
      uint4 fa[MR], fb[NR];
#pragma unroll
      for (int i = 0; i < MR; i++) fa[i] = make_uint4(ra[0][i].x, ra[0][i].y, ra[1][i].x, ra[1][i].y);
#pragma unroll
      for (int j = 0; j < NR; j++) fb[j] = make_uint4(rb[0][j].x, rb[0][j].y, rb[1][j].x, rb[1][j].y);
#pragma unroll
      for (int i = 0; i < MR; i++)
#pragma unroll
        for (int j = 0; j < NR; j++) acc[i][j] = ys_mma<T>(fa[i], fb[j], acc[i][j]);
    }
  }

  float* outp = a.partial + (long)split * a.Cout * g.taps * a.Cin;
#pragma unroll
  for (int i = 0; i < MR; i++)
#pragma unroll
    for (int j = 0; j < NR; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int co = co0 + (wm * MR + i) * 16 + 4 * q + r, ci = ci0 + (wn * NR + j) * 16 + li;
        if (co < a.Cout && ci < a.Cin) outp[((long)co * g.taps + tap) * a.Cin + ci] = acc[i][j][r];
      }
}

// ------------------------------------------------------------------ host side
struct WgGemmPlan { int ok, mr, nr, kt, gx; size_t lds; WgGemmArgs g; };

static int wgemm_pick_tile(int c) {           // 160 or 128 channels: least padding, then the wider tile
  const long p160 = (long)ys_cdiv(c, 160) * 160, p128 = (long)ys_cdiv(c, 128) * 128;
  return p160 <= p128 ? 160 : 128;
}

static WgGemmPlan wgrad_gemm_plan(const WgradArgs& a) {
  WgGemmPlan p{};
  const int min_c = 128;
  const int min_m = (int)YS_OPT_INT("WGEMM_MIN_M", 4096);
  const long kt_opt = YS_OPT_INT("WGEMM_KT", 0);   // (the tests switch K-tile variants inside one process)
  const int kt_env = (int)kt_opt;
  const bool k3 = a.KH == 3 && a.KW == 3 && a.pad == 1 && (a.stride == 1 || a.stride == 2);
  const bool k1 = a.KH == 1 && a.KW == 1 && a.pad == 0 && a.stride == 1;
  if (!(k3 || k1) || (a.dy_rh && !k1)) return p;
  if (a.Cin % 8 || a.Cout % 8 || a.in_ldc % 8 || a.in_coff % 8 || a.dy_ldc % 8 || a.dy_coff % 8) return p;
  if (a.Cin < min_c || a.Cout < min_c || a.M < min_m || a.M >= (1 << 24)) return p;
  const int bco = wgemm_pick_tile(a.Cout), bci = wgemm_pick_tile(a.Cin);
  p.mr = bco / 32; p.nr = bci / 32;
  p.kt = kt_env == 32 ? 32 : 64;
  if ((size_t)2 * p.kt * (bco + bci) * 2 > 80 * 1024) p.kt = 32;
  p.lds = (size_t)2 * p.kt * (bco + bci) * 2;
  WgGemmArgs g{};
  g.co_tiles = ys_cdiv(a.Cout, bco); g.ci_tiles = ys_cdiv(a.Cin, bci); g.taps = a.KH * a.KW;
  g.gy = g.co_tiles * g.ci_tiles * g.taps;
  g.nkt = ys_cdiv(a.M, p.kt);
  g.HoWo = a.Hout * a.Wout;
  g.inv_howo = 1.0f / (float)g.HoWo; g.inv_wo = 1.0f / (float)a.Wout;
  {
    const long dpix = a.dy_rh ? (long)a.B * a.dy_bstride : (long)(a.B - 1) * a.dy_bstride + (long)a.Hout * a.Wout, xpix = (long)(a.B - 1) * a.in_bstride + (long)a.Hin * a.Win;
    const long db = (dpix * a.dy_ldc - a.dy_coff) * 2L, xbts = (xpix * a.in_ldc - a.in_coff) * 2L;
    if (db <= 0 || xbts <= 0 || db >= (1L << 31) || xbts >= (1L << 31)) return p;      // 32-bit request offsets
    g.dybytes = (unsigned)db; g.xbytes = (unsigned)xbts;
  }
  const int wpc = 2;     // 256-register waves: two workgroups per CU
  long gx = ((long)ys_cu_count() * wpc) / g.gy;
  if (gx < 1) gx = 1;
  const long wsmax = (48L << 20) / ((long)a.Cout * g.taps * a.Cin * 4);   // bound the partial workspace to 48 MB per layer
  if (gx > wsmax) gx = wsmax > 0 ? wsmax : 1;
  if (gx > g.nkt) gx = g.nkt;
  g.per = ys_cdiv(g.nkt, gx);
  gx = ys_cdiv(g.nkt, g.per);                 // equal shares, no empty split
  p.gx = (int)gx;
  p.g = g;
  p.ok = 1;
  return p;
}

template <int MR, int NR, int KT>
static int wgrad_gemm_launch_t(hipStream_t st, const WgradArgs& a, const WgGemmPlan& p) {
  static std::atomic<unsigned> attr_done{0};      // per device: the attribute belongs to the device's code object
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  if (!(attr_done.load(std::memory_order_relaxed) & (1u << (dev_id & 31)))) {
    hipFuncSetAttribute((const void*)conv_wgrad_gemm_kernel<MR, NR, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done.fetch_or(1u << (dev_id & 31), std::memory_order_relaxed);
  }
  char lab[176] = "";
  if (ys_kprof_enabled()) snprintf(lab, sizeof(lab), "wgemm k%d s%d cin%d cout%d M%d tile%dx%d kt%d splits%d blocks%d lds%d", a.KH, a.stride, a.Cin, a.Cout, a.M, MR * 32, NR * 32, KT, p.gx, p.g.gy, (int)p.lds);
  YsKprofScope prof(st, "conv_wgrad", lab);
  YS_LAUNCH_LDS((conv_wgrad_gemm_kernel<MR, NR, KT>), dim3(p.gx * p.g.gy), 256, p.lds, st, a, p.g);
  return YS_OK;
}

int ys_wgrad_gemm_splits(const WgradArgs& a) {
  const WgGemmPlan p = wgrad_gemm_plan(a);
  return p.ok ? p.gx : 0;
}

// launches with at most `splits` pixel splits (the caller's partial workspace); returns the number used, 0 = not eligible
int ys_wgrad_gemm_launch(hipStream_t st, const WgradArgs& a, int splits) {
  WgGemmPlan p = wgrad_gemm_plan(a);
  if (!p.ok) return 0;
  if (p.gx > splits) {
    p.g.per = ys_cdiv(p.g.nkt, splits);
    p.gx = ys_cdiv(p.g.nkt, p.g.per);
  }
#define WGM(M_, N_) if (p.mr == M_ && p.nr == N_) { if (p.kt == 64) wgrad_gemm_launch_t<M_, N_, 64>(st, a, p); else wgrad_gemm_launch_t<M_, N_, 32>(st, a, p); return p.gx; }
  WGM(5, 5) WGM(5, 4) WGM(4, 5) WGM(4, 4)
#undef WGM
  return 0;
}
