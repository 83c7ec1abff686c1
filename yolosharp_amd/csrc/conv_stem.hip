// conv_stem.hip -- the first convolution of the YOLO graphs (model.0 = Conv(3, c, k = 3, s = 2): Models/Yolo.cs:43, Modules/Convs.cs:36-62)
// read STRAIGHT from the image tensor the reference hands over: fp32 NCHW [B][3][H][W] (Data/YoloDataLoader.cs:18-44), gfx950 (CDNA4).
//
// Rounds 1-3 converted the image into the engine's bf16 NHWC layout first (pack_input4_kernel: 315 MB read + 420 MB written per B = 64 step
// at 640 x 640, the 3 channels padded to one 16-byte unit of 8) and then ran the generic whole-Cin patch kernel and the generic weight-
// gradient kernel on the padded copy (another 420 MB read each): 0.47 ms of a 9.8 ms step spent on a layer with 432 weights.  Here the
// forward and the weight gradient stage their input patch from the fp32 planes themselves (rounded to bf16 on the way into LDS, the
// same rounding the packed copy had), so the packed copy and its pass disappear: the layer reads 315 MB and writes 210 MB in the
// forward, reads 315 + 210 MB in the backward.  Both kernels are plain streaming kernels -- small workgroups, 7-15 KB of LDS, eight
// workgroups per CU, no persistence tricks: the layer is 5.7 GFLOP against 0.5 GB, i.e. HBM-bound by two orders of magnitude.
//
// GEMM view (one v_mfma_f32_16x16x32_bf16 per 16 pixels x 16 output channels): K = 27 = (tap, channel) padded to 32,
//   forward   D[cout][pixel] = sum_k W[cout][k] X[pixel][k]        (weights as the row operand, like every other conv kernel here)
//   wgrad     D[cout][k]     = sum_pixels dy[pixel][cout] X[pixel][k]   (K = 32 consecutive pixels of one output row)
// A workgroup tile is 8 x 32 output pixels; its input patch is 17 x 65 x 3 values.
#include "ys_internal.h"
#include "ys_kernels.h"
#include <cstdlib>

#define STEM_TH 8
#define STEM_TW 32
#define STEM_PH (2 * STEM_TH + 1)      // 17 input rows
#define STEM_PW (2 * STEM_TW + 1)      // 65 input columns
#define STEM_NPATCH (3 * STEM_PH * STEM_PW)
#define STEM_NLD ((STEM_NPATCH + 255) / 256)   // patch values per thread (13)
#define STEM_THREADS 256

struct StemFwdArgs {
  const float* x;            // [B][3][H][W] fp32
  const bf16_t* wf;          // forward weight shadow [Cout][9][8] (channels 3..7 zero)
  bf16_t* y;                 // [B][out_bstride][out_ldc] + out_coff
  float* stats;              // training: one row [2][Cout] per workgroup (sum, sum of squares of the bf16-rounded outputs)
  unsigned long long* stat_acc;   // ... or, when set, fixed-point integer accumulators (ys_kernels.h ys_stat_acc_add)
  const float* scale;        // eval: BatchNorm folded to scale / shift (+ SiLU when act)
  const float* shift;
  int B, H, W, Hout, Wout, Cout, out_ldc, out_coff, act;
  long out_bstride;          // rows per image of the output buffer
  int tiles_x, tiles_y, ntiles;
};

// fp32 image patch of tile (b, oy0, ox0) -> registers (all loads in flight), zero outside the image
__device__ inline void stem_patch_fetch(const float* __restrict__ x, int b, int iy0, int ix0, int H, int W, float (&v)[STEM_NLD]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < STEM_NLD; k++) {
    const int e = tid + STEM_THREADS * k;
    const int c = e / (STEM_PH * STEM_PW), rem = e - c * (STEM_PH * STEM_PW);
    const int r = rem / STEM_PW, col = rem - r * STEM_PW;
    const int iy = iy0 + r, ix = ix0 + col;
    const bool ok = (bool)((int)(e < STEM_NPATCH) & (int)((unsigned)iy < (unsigned)H) & (int)((unsigned)ix < (unsigned)W));
    const long off = ok ? (((long)b * 3 + c) * H + iy) * (long)W + ix : 0;
    const float f = x[off];
    v[k] = ok ? f : 0.f;
  }
}

// Round 6: the same patch as 16-byte vectors.  A patch row is input columns ix0 .. ix0 + 64 with ix0 = 2 ox0 - 1: the left halo column, then 64 columns that start
// at a multiple of 64 -- sixteen aligned float4 when W % 4 == 0.  816 vectors + 51 halo scalars per tile instead of 3315 scalars: the element form spent ~35 integer
// instructions per 4-byte load (two divisions by constants, bounds, 64-bit address) and ~15 per 2-byte LDS store, 13 of each per thread and tile -- both kernels
// were instruction-bound at 2.2-2.5 TB/s (8 workgroups per CU already cover the latency).  Vector u = tid + 256 k: row u / 16 = (c, r), columns 1 + 4 (u % 16) .. + 3.
#define STEM_NVEC (3 * STEM_PH * 16)            // 816
#define STEM_NVL ((STEM_NVEC + STEM_THREADS - 1) / STEM_THREADS)   // 4 per thread (the last round partial)
struct StemPatchV { float4 v[STEM_NVL]; float h; };
__device__ inline void stem_patch_fetch_v(const float* __restrict__ x, int b, int iy0, int ix0, int H, int W, StemPatchV& p) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < STEM_NVL; k++) {
    const int u = tid + STEM_THREADS * k;
    const int row = u >> 4, j = u & 15;
    const int c = row / STEM_PH, r = row - c * STEM_PH;
    const int iy = iy0 + r, ix = ix0 + 1 + 4 * j;
    const bool ok = (bool)((int)(u < STEM_NVEC) & (int)((unsigned)iy < (unsigned)H) & (int)(ix + 3 < W));
    const long off = ok ? (((long)b * 3 + c) * H + iy) * (long)W + ix : 0;
    const float4 f = *(const float4*)(x + off);
    p.v[k] = ok ? f : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  {
    const int row = tid < 3 * STEM_PH ? tid : 0;
    const int c = row / STEM_PH, r = row - c * STEM_PH;
    const int iy = iy0 + r;
    const bool ok = (bool)((int)(tid < 3 * STEM_PH) & (int)((unsigned)iy < (unsigned)H) & (int)(ix0 >= 0));
    const long off = ok ? (((long)b * 3 + c) * H + iy) * (long)W + ix0 : 0;
    const float f = x[off];
    p.h = ok ? f : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------------------------ forward
// LDS patch: planar bf16 [3][17][66].  Lane (li, q) of a pixel fragment supplies K values k = 8q .. 8q+7 of pixel li, k = tap * 3 + c.
template <int NR, int EVAL, int VEC>
__global__ void __launch_bounds__(STEM_THREADS)
stem_fwd_kernel(StemFwdArgs a) {
  constexpr int PWP = STEM_PW + 1;
  __shared__ unsigned short sP[3 * STEM_PH * PWP + 2];
  __shared__ float sStat[4][NR * 16][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;

  // weight fragments (rows = output channels) and this lane's patch offsets: both tile-independent
  uint4 wfr[NR];
  int offp[8];
  unsigned kmask = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int k = q * 8 + j;
    const int tap = k / 3, c = k - tap * 3;
    const int kh = tap / 3, kw = tap - kh * 3;
    const bool ok = k < 27;
    offp[j] = ok ? (c * STEM_PH + kh) * PWP + kw : 0;
    kmask |= (unsigned)ok << j;
  }
#pragma unroll
  for (int nf = 0; nf < NR; nf++) {
    unsigned short e[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = q * 8 + j;
      const int tap = k / 3, c = k - tap * 3;
      const int co = nf * 16 + li;
      const bool ok = (bool)((int)(k < 27) & (int)(co < a.Cout));
      const unsigned short w = a.wf[ok ? ((long)co * 9 + tap) * 8 + c : 0].v;
      e[j] = ok ? w : (unsigned short)0;
    }
    wfr[nf] = make_uint4((unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                         (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16));
  }
  float sc[NR][4], sh[NR][4];
  if (EVAL) {
#pragma unroll
    for (int nf = 0; nf < NR; nf++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int c = nf * 16 + q * 4 + r;
        sc[nf][r] = c < a.Cout ? a.scale[c] : 0.f;
        sh[nf][r] = c < a.Cout ? a.shift[c] : 0.f;
      }
  }
  float st1[NR][4], st2[NR][4];
#pragma unroll
  for (int nf = 0; nf < NR; nf++)
#pragma unroll
    for (int r = 0; r < 4; r++) { st1[nf][r] = 0.f; st2[nf][r] = 0.f; }

  // VEC: the NEXT tile's patch is requested as soon as the current one sits in LDS, so every workgroup has 13 KB in flight all the time (round 6: without it a
  // workgroup's loads were only outstanding during ~40 % of its tile period -- 8 workgroups x 13 KB x 0.4 per CU is half of what 5 TB/s needs at ~3 us latency)
  StemPatchV pv;
  if (VEC && (int)blockIdx.x < a.ntiles) {
    const int t0 = blockIdx.x, tx = t0 % a.tiles_x, trem = t0 / a.tiles_x;
    stem_patch_fetch_v(a.x, trem / a.tiles_y, 2 * (trem % a.tiles_y) * STEM_TH - 1, 2 * tx * STEM_TW - 1, a.H, a.W, pv);
  }
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int tx = tile % a.tiles_x, trem = tile / a.tiles_x;
    const int ty = trem % a.tiles_y, b = trem / a.tiles_y;
    const int oy0 = ty * STEM_TH, ox0 = tx * STEM_TW;
    if (VEC) {
      __syncthreads();                        // the previous tile's fragments are read
#pragma unroll
      for (int k = 0; k < STEM_NVL; k++) {
        const int u = tid + STEM_THREADS * k;
        if (u < STEM_NVEC) {
          unsigned short* d = sP + (u >> 4) * PWP + 1 + 4 * (u & 15);      // odd element: 2-byte, 4-byte (PWP is even), 2-byte stores
          const unsigned lo = ys_pack_bf16x2(pv.v[k].x, pv.v[k].y), hi = ys_pack_bf16x2(pv.v[k].z, pv.v[k].w);
          d[0] = (unsigned short)(lo & 0xffffu);
          *(unsigned*)(d + 1) = (lo >> 16) | (hi << 16);
          d[3] = (unsigned short)(hi >> 16);
        }
      }
      if (tid < 3 * STEM_PH) sP[tid * PWP] = Elem<bf16_t>::from_f(pv.h).v;
      const int nt = tile + gridDim.x;
      if (nt < a.ntiles) {
        const int ntx = nt % a.tiles_x, ntrem = nt / a.tiles_x;
        stem_patch_fetch_v(a.x, ntrem / a.tiles_y, 2 * (ntrem % a.tiles_y) * STEM_TH - 1, 2 * ntx * STEM_TW - 1, a.H, a.W, pv);
      }
    } else {
      float v[STEM_NLD];
      stem_patch_fetch(a.x, b, 2 * oy0 - 1, 2 * ox0 - 1, a.H, a.W, v);
      __syncthreads();                        // the previous tile's fragments are read
#pragma unroll
      for (int k = 0; k < STEM_NLD; k++) {
        const int e = tid + STEM_THREADS * k;
        if (e < STEM_NPATCH) {
          const int c = e / (STEM_PH * STEM_PW), rem = e - c * (STEM_PH * STEM_PW);
          const int r = rem / STEM_PW, col = rem - r * STEM_PW;
          sP[(c * STEM_PH + r) * PWP + col] = Elem<bf16_t>::from_f(v[k]).v;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int mf = 0; mf < 4; mf++) {
      const int oyl = 2 * wave + (mf >> 1), oxl = (mf & 1) * 16 + li;
      const unsigned short* pp = sP + (2 * oyl) * PWP + 2 * oxl;
      unsigned short e[8];
#pragma unroll
      for (int j = 0; j < 8; j++) { const unsigned short t = pp[offp[j]]; e[j] = ((kmask >> j) & 1u) ? t : (unsigned short)0; }
      const uint4 xf = make_uint4((unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                                  (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16));
      const int oy = oy0 + oyl, ox = ox0 + oxl;
      const bool valid = (bool)((int)(oy < a.Hout) & (int)(ox < a.Wout));
      bf16_t* yp = a.y + (((long)b * a.out_bstride + (long)oy * a.Wout + ox) * a.out_ldc + a.out_coff + q * 4);
#pragma unroll
      for (int nf = 0; nf < NR; nf++) {
        const f32x4 d = ys_mma<bf16_t>(wfr[nf], xf, f32x4_zero());
        uint2 pk;
        pk.x = ys_pack_bf16x2(d[0], d[1]);
        pk.y = ys_pack_bf16x2(d[2], d[3]);
        // everything downstream sees the bf16-rounded output: statistics (training) and the folded BatchNorm (eval) use it too
        float f[4] = {ys_u2f(pk.x << 16), ys_u2f(pk.x & 0xffff0000u), ys_u2f(pk.y << 16), ys_u2f(pk.y & 0xffff0000u)};
        if (EVAL) {
#pragma unroll
          for (int r = 0; r < 4; r++) { f[r] = f[r] * sc[nf][r] + sh[nf][r]; }
          if (a.act) {
#pragma unroll
            for (int r = 0; r < 4; r++) f[r] = ys_silu(f[r]);
          }
          pk.x = ys_pack_bf16x2(f[0], f[1]);
          pk.y = ys_pack_bf16x2(f[2], f[3]);
        } else if (valid) {
#pragma unroll
          for (int r = 0; r < 4; r++) { st1[nf][r] += f[r]; st2[nf][r] += f[r] * f[r]; }
        }
        if (valid && nf * 16 + q * 4 < a.Cout) *(uint2*)(yp + nf * 16) = pk;
      }
    }
  }
  if (!EVAL && a.stats) {
    // one statistics row per workgroup: the 16 pixel lanes of a quarter, then the four waves in a fixed order
#pragma unroll
    for (int nf = 0; nf < NR; nf++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float s1 = st1[nf][r], s2 = st2[nf][r];
#pragma unroll
        for (int msk = 1; msk < 16; msk <<= 1) { s1 += __shfl_xor(s1, msk); s2 += __shfl_xor(s2, msk); }
        if (li == 0) { sStat[wave][nf * 16 + q * 4 + r][0] = s1; sStat[wave][nf * 16 + q * 4 + r][1] = s2; }
      }
    __syncthreads();
    if (tid < NR * 16 && tid < a.Cout) {
      const float s1 = (sStat[0][tid][0] + sStat[1][tid][0]) + (sStat[2][tid][0] + sStat[3][tid][0]);
      const float s2 = (sStat[0][tid][1] + sStat[1][tid][1]) + (sStat[2][tid][1] + sStat[3][tid][1]);
      if (a.stat_acc) { ys_stat_acc_add(a.stat_acc, (long)blockIdx.x, a.Cout, tid, 0, s1); ys_stat_acc_add(a.stat_acc, (long)blockIdx.x, a.Cout, tid, 1, s2); }
      else {
        a.stats[((long)blockIdx.x * 2 + 0) * a.Cout + tid] = s1;
        a.stats[((long)blockIdx.x * 2 + 1) * a.Cout + tid] = s2;
      }
    }
  }
}

bool ys_stem_eligible(int dtype, int cin, int cout, int k, int s) {
  return dtype == YS_BF16 && cin == 3 && k == 3 && s == 2 && cout % 16 == 0 && cout >= 16 && cout <= 80;
}

static int stem_grid(long ntiles) { return (int)(ntiles < 2048 ? ntiles : 2048); }   // eight workgroups per CU; rows of the statistics region

int ys_stem_fwd_rows(int B, int Hout, int Wout) {
  return stem_grid((long)B * ys_cdiv(Hout, STEM_TH) * ys_cdiv(Wout, STEM_TW));
}

// training (stats != nullptr): raw bf16 output + one statistics row per workgroup (*rows); eval: scale / shift (+ SiLU) applied
int ys_stem_fwd_launch(hipStream_t st, const float* x, int B, int H, int W, const void* wf, int Cout, void* y, int out_ldc, int out_coff,
                       long out_bstride, float* stats, const float* scale, const float* shift, int act, int* rows, unsigned long long* stat_acc) {
  StemFwdArgs a{};
  a.x = x; a.wf = (const bf16_t*)wf; a.y = (bf16_t*)y; a.stats = stats; a.stat_acc = stat_acc; a.scale = scale; a.shift = shift; a.act = act;
  a.B = B; a.H = H; a.W = W; a.Hout = (H - 1) / 2 + 1; a.Wout = (W - 1) / 2 + 1; a.Cout = Cout;
  a.out_ldc = out_ldc; a.out_coff = out_coff; a.out_bstride = out_bstride;
  a.tiles_x = ys_cdiv(a.Wout, STEM_TW); a.tiles_y = ys_cdiv(a.Hout, STEM_TH); a.ntiles = B * a.tiles_x * a.tiles_y;
  if ((long)B * 3 * H * W >= (1L << 31) || (out_ldc & 3) || (out_coff & 3)) { ys_set_error("stem conv: unsupported view"); return YS_ERR_UNSUPPORTED; }
  const int grid = stem_grid(a.ntiles);
  if (rows) *rows = grid;
  const int nr = Cout / 16;
  const bool eval = stats == nullptr;
  if (eval && (!scale || !shift)) { ys_set_error("stem conv: eval launch without BatchNorm coefficients"); return YS_ERR_INVALID_ARG; }
  char lab[160] = "";
  if (ys_kprof_enabled()) snprintf(lab, sizeof(lab), "stem k33 s2 div1 cin3 cout%d M%ld acc0 tile%dx%d grid%dx1", Cout, (long)B * a.Hout * a.Wout, STEM_TH, STEM_TW, grid);
  YsKprofScope prof(st, "conv_igemm", lab);
  const bool vec = (W & 3) == 0 && ((size_t)x & 15) == 0;       // aligned float4 rows (every BASELINE shape: W is a multiple of 32)
#define STEM_F(N_) if (nr == N_) { \
    if (eval) { if (vec) YS_LAUNCH((stem_fwd_kernel<N_, 1, 1>), grid, STEM_THREADS, st, a); else YS_LAUNCH((stem_fwd_kernel<N_, 1, 0>), grid, STEM_THREADS, st, a); } \
    else { if (vec) YS_LAUNCH((stem_fwd_kernel<N_, 0, 1>), grid, STEM_THREADS, st, a); else YS_LAUNCH((stem_fwd_kernel<N_, 0, 0>), grid, STEM_THREADS, st, a); } \
    return YS_OK; }
  STEM_F(1) STEM_F(2) STEM_F(3) STEM_F(4) STEM_F(5)
#undef STEM_F
  ys_set_error("stem conv: %d output channels", Cout);
  return YS_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------------------------ weight gradient
// dW[cout][tap][c] = sum over output pixels of dy[pixel][cout] * x[c][2 oy + kh - 1][2 ox + kw - 1].  The MFMA's K dimension is 32
// consecutive pixels of one output row, so both operands want eight CONSECUTIVE pixels per lane:
//  * x: the patch is kept in LDS split by column parity -- input column 2 ox + kw - 1 of consecutive ox is a unit-stride run of the even
//    plane (kw = 0: entries ox, kw = 2: entries ox + 1) or of the odd plane (kw = 1); a second copy of the even plane shifted by one
//    entry keeps every 16-byte fragment read aligned.  Planes [3][17][40] bf16.
//  * dy: the tile's rows [256 pixels][Cout] sit in LDS as they are in memory; a lane gathers its output channel's eight pixels.
// A workgroup accumulates its tiles in registers and leaves ONE partial slab [Cout][9][8] (the generic kernel's split layout, so the
// batched split reduction of the backward segment takes it as it is).
struct StemWgArgs {
  const float* x;
  const bf16_t* dy;          // [B][dy_bstride][dy_ldc] + dy_coff
  float* partial;            // [gridDim.x][Cout][9][8]
  int B, H, W, Hout, Wout, Cout, dy_ldc, dy_coff;
  long dy_bstride;
  int tiles_x, tiles_y, ntiles;
};

template <int NR, int VEC>
__global__ void __launch_bounds__(STEM_THREADS)
stem_wgrad_kernel(StemWgArgs a) {
  constexpr int PL = 40;                                // plane row pitch (entries): a multiple of 8 -> aligned 16-byte reads
  constexpr int PLANE = 3 * STEM_PH * PL;
  constexpr int DP = NR * 16 + 2;                       // dy row pitch in LDS (entries)
  // one LDS block: the three parity planes, then the dy rows -- which become the cross-wave reduction scratch after the last tile
  constexpr int DY_BYTES = STEM_TH * STEM_TW * DP * 2, RED_BYTES = 4 * NR * 2 * 64 * 4 * 4;
  constexpr int R2_BYTES = DY_BYTES > RED_BYTES ? DY_BYTES : RED_BYTES;
  __shared__ uint4 sMem[(3 * PLANE * 2 + R2_BYTES + 15) / 16];
  unsigned short* sE = (unsigned short*)sMem;           // even patch columns: entry i = patch column 2 i      (input column 2 (ox0 + i) - 1)
  unsigned short* sO = sE + PLANE;                      // odd patch columns:  entry i = patch column 2 i + 1
  unsigned short* sE1 = sO + PLANE;                     // even plane shifted: entry i = patch column 2 i + 2
  unsigned short* sDy = sE1 + PLANE;                    // PLANE * 2 bytes is a multiple of 16
  float (*sRed)[NR][2][64][4] = (float (*)[NR][2][64][4])(sE1 + PLANE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;

  // this lane's column of the B operand: k = kb * 16 + li -> (tap, c) -> plane and row offset
  int boff[2]; int bsel[2];
#pragma unroll
  for (int kb = 0; kb < 2; kb++) {
    const int k = kb * 16 + li;
    const int tap = k / 3, c = k - tap * 3;
    const int kh = tap / 3, kw = tap - kh * 3;
    bsel[kb] = k < 27 ? (kw == 1 ? 1 : (kw == 2 ? 2 : 0)) : 3;
    boff[kb] = k < 27 ? (c * STEM_PH + kh) * PL : 0;
  }
  f32x4 acc[NR][2];
#pragma unroll
  for (int nf = 0; nf < NR; nf++) { acc[nf][0] = f32x4_zero(); acc[nf][1] = f32x4_zero(); }

  // dy rows of a tile: 256 pixels x NR*16 channels, 16-byte units (8 channels), zero outside the image
  constexpr int DU = STEM_TH * STEM_TW * NR * 2;      // 16-byte units of the tile
  constexpr int NDU = (DU + STEM_THREADS - 1) / STEM_THREADS;
  float v[STEM_NLD];                          // (the form not taken is dead code: VEC is a template parameter)
  StemPatchV pv;
  uint4 dv[NDU];
  auto fetch_tile = [&](int t) {              // everything tile t needs from memory, all loads back to back
    const int tx = t % a.tiles_x, trem = t / a.tiles_x;
    const int ty = trem % a.tiles_y, b = trem / a.tiles_y;
    const int oy0 = ty * STEM_TH, ox0 = tx * STEM_TW;
    if (VEC) stem_patch_fetch_v(a.x, b, 2 * oy0 - 1, 2 * ox0 - 1, a.H, a.W, pv);
    else stem_patch_fetch(a.x, b, 2 * oy0 - 1, 2 * ox0 - 1, a.H, a.W, v);
#pragma unroll
    for (int k = 0; k < NDU; k++) {
      const int u = tid + STEM_THREADS * k;
      const int px = u / (NR * 2), cu = u - px * (NR * 2);
      const int oyl = px / STEM_TW, oxl = px - oyl * STEM_TW;
      const int oy = oy0 + oyl, ox = ox0 + oxl;
      const bool ok = (bool)((int)(u < DU) & (int)(oy < a.Hout) & (int)(ox < a.Wout) & (int)(cu * 8 < a.Cout));
      const long off = ok ? (((long)b * a.dy_bstride + (long)oy * a.Wout + ox) * a.dy_ldc + a.dy_coff + cu * 8) : 0;
      const uint4 t4 = *(const uint4*)(a.dy + off);
      dv[k] = ok ? t4 : ys_zero16();
    }
  };
  // the NEXT tile's operands are requested as soon as the current ones sit in LDS (round 6, as in stem_fwd_kernel)
  if ((int)blockIdx.x < a.ntiles) fetch_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();                          // the previous tile's fragments are read
    if (VEC) {
      // vector u holds patch columns 1 + 4j .. 4 + 4j of row u / 16: the odd ones (1 + 4j, 3 + 4j) are entries 2j, 2j + 1 of the odd plane, the even ones
      // (2 + 4j, 4 + 4j) entries 2j + 1, 2j + 2 of the even plane and 2j, 2j + 1 of its shifted copy
#pragma unroll
      for (int k = 0; k < STEM_NVL; k++) {
        const int u = tid + STEM_THREADS * k;
        if (u < STEM_NVEC) {
          const int row = (u >> 4) * PL, j2 = 2 * (u & 15);
          const unsigned od = ys_pack_bf16x2(pv.v[k].x, pv.v[k].z), ev = ys_pack_bf16x2(pv.v[k].y, pv.v[k].w);
          *(unsigned*)(sO + row + j2) = od;
          *(unsigned*)(sE1 + row + j2) = ev;
          sE[row + j2 + 1] = (unsigned short)(ev & 0xffffu);
          sE[row + j2 + 2] = (unsigned short)(ev >> 16);
        }
      }
      if (tid < 3 * STEM_PH) sE[tid * PL] = Elem<bf16_t>::from_f(pv.h).v;      // patch column 0 (the left halo): entry 0 of the even plane
    } else {
#pragma unroll
      for (int k = 0; k < STEM_NLD; k++) {
        const int e = tid + STEM_THREADS * k;
        if (e < STEM_NPATCH) {
          const int c = e / (STEM_PH * STEM_PW), rem = e - c * (STEM_PH * STEM_PW);
          const int r = rem / STEM_PW, col = rem - r * STEM_PW;
          const unsigned short h = Elem<bf16_t>::from_f(v[k]).v;
          const int row = (c * STEM_PH + r) * PL;
          if (col & 1) sO[row + (col >> 1)] = h;
          else { sE[row + (col >> 1)] = h; if (col >= 2) sE1[row + (col >> 1) - 1] = h; }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NDU; k++) {
      const int u = tid + STEM_THREADS * k;
      if (u < DU) {
        const int px = u / (NR * 2), cu = u - px * (NR * 2);
        unsigned* d = (unsigned*)(sDy + px * DP + cu * 8);     // DP is even: 4-byte aligned
        d[0] = dv[k].x; d[1] = dv[k].y; d[2] = dv[k].z; d[3] = dv[k].w;
      }
    }
    if (tile + (int)gridDim.x < a.ntiles) fetch_tile(tile + gridDim.x);
    __syncthreads();
    // wave w: output rows 2w, 2w + 1 of the tile; one K = 32 step per row
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
      const int oyl = 2 * wave + rr;
      uint4 xf[2];
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        const unsigned short* pl = bsel[kb] == 1 ? sO : (bsel[kb] == 2 ? sE1 : sE);
        const uint4 t = *(const uint4*)(pl + boff[kb] + (2 * oyl) * PL + q * 8);
        xf[kb] = bsel[kb] == 3 ? ys_zero16() : t;
      }
#pragma unroll
      for (int nf = 0; nf < NR; nf++) {
        const unsigned short* dp = sDy + (oyl * STEM_TW + q * 8) * DP + nf * 16 + li;
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; j++) e[j] = dp[j * DP];
        const uint4 df = make_uint4((unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                                    (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16));
        acc[nf][0] = ys_mma<bf16_t>(df, xf[0], acc[nf][0]);
        acc[nf][1] = ys_mma<bf16_t>(df, xf[1], acc[nf][1]);
      }
    }
  }
  // the four waves' accumulators, summed in a fixed order -> this workgroup's slab.  Lane (li, q) holds D[cout = 4q + r][k = kb*16 + li].
  // The scratch aliases the dy rows: every wave must be through its last fragment reads first (round 5: without this barrier a wave that finished early wrote
  // its sums over rows a slower wave was still reading -- fp32 bit patterns read as bf16 pairs, an occasional wrong or NaN stem weight gradient, mostly on the
  // first launch of a process or model when the waves of a workgroup are furthest apart; found by tools/dev/r05/det_phase.py).
  __syncthreads();
#pragma unroll
  for (int nf = 0; nf < NR; nf++)
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 4; r++) sRed[wave][nf][kb][lane][r] = acc[nf][kb][r];
  __syncthreads();
  float* slab = a.partial + (long)blockIdx.x * a.Cout * 72;
  for (int o = tid; o < a.Cout * 72; o += STEM_THREADS) {
    const int co = o / 72, rem = o - co * 72;
    const int tap = rem >> 3, c = rem & 7;
    float s = 0.f;
    if (c < 3) {
      const int k = tap * 3 + c;
      const int nf = co >> 4, qq = (co & 15) >> 2, r = co & 3, kb = k >> 4, l = (k & 15) + 16 * qq;
      s = (sRed[0][nf][kb][l][r] + sRed[1][nf][kb][l][r]) + (sRed[2][nf][kb][l][r] + sRed[3][nf][kb][l][r]);
    }
    slab[o] = s;
  }
}

// writes min(max_splits, 1024) partial slabs [Cout][9][8] into `partial`; *used = the number written
int ys_stem_wgrad_launch(hipStream_t st, const float* x, int B, int H, int W, const void* dy, int dy_ldc, int dy_coff, long dy_bstride,
                         int Cout, float* partial, int max_splits, int* used) {
  StemWgArgs a{};
  a.x = x; a.dy = (const bf16_t*)dy; a.partial = partial;
  a.B = B; a.H = H; a.W = W; a.Hout = (H - 1) / 2 + 1; a.Wout = (W - 1) / 2 + 1; a.Cout = Cout;
  a.dy_ldc = dy_ldc; a.dy_coff = dy_coff; a.dy_bstride = dy_bstride;
  a.tiles_x = ys_cdiv(a.Wout, STEM_TW); a.tiles_y = ys_cdiv(a.Hout, STEM_TH); a.ntiles = B * a.tiles_x * a.tiles_y;
  if ((long)B * 3 * H * W >= (1L << 31) || (dy_ldc & 7) || (dy_coff & 7) || max_splits < 1) { ys_set_error("stem wgrad: unsupported view"); return YS_ERR_UNSUPPORTED; }
  int grid = max_splits < 1024 ? max_splits : 1024;
  if (grid > a.ntiles) grid = a.ntiles;
  if (used) *used = grid;
  const int nr = Cout / 16;
  char lab[160] = "";
  if (ys_kprof_enabled()) snprintf(lab, sizeof(lab), "wstem k3 s2 cin3 cout%d M%ld tile%dx%d grid%dx1", Cout, (long)B * a.Hout * a.Wout, STEM_TH, STEM_TW, grid);
  YsKprofScope prof(st, "conv_wgrad", lab);
  const bool vec = (W & 3) == 0 && ((size_t)x & 15) == 0;
#define STEM_W(N_) if (nr == N_) { if (vec) YS_LAUNCH((stem_wgrad_kernel<N_, 1>), grid, STEM_THREADS, st, a); else YS_LAUNCH((stem_wgrad_kernel<N_, 0>), grid, STEM_THREADS, st, a); return YS_OK; }
  STEM_W(1) STEM_W(2) STEM_W(3) STEM_W(4) STEM_W(5)
#undef STEM_W
  ys_set_error("stem wgrad: %d output channels", Cout);
  return YS_ERR_UNSUPPORTED;
}
