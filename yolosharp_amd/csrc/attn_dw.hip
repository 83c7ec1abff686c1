// attn_dw.hip -- YOLOv11-only operators:
//   * depthwise 3x3/s1 convolution (Convs.DWConv, Modules/Convs.cs:108-114: groups = gcd(c1,c2) = C) used by the
//     legacy=false cls towers of Detect (Modules/Head.cs:50) and by Attention.pe (Modules/Block.cs:746)
//   * the attention core of C2PSA/PSABlock/Attention (Modules/Block.cs:763-805): per head
//       attn = softmax(q^T k * key_dim^-0.5) ; x = v @ attn^T        (N = H*W tokens, 400 at 640x640)
//     forward, and the two-pass backward (dS/dq per query row, dk/dv per key column).
// All of it is HBM/latency-bound small-tensor work (the 20x20 level): no MFMA, fp32 math, T storage, NHWC views.
#include "ys_internal.h"
#include "ys_kernels.h"

#define AD_THREADS 256

// ------------------------------------------------------------------ depthwise 3x3, stride 1, pad 1
// FLIP = 0: y[p,c] = sum_t w[t][c] * x[p + off(t), c]            (forward)
// FLIP = 1: dx[p,c] = sum_t w[t][c] * dy[p - off(t), c]          (input gradient)
// Round 5: a thread owns DW_PX consecutive pixels of a row for its channel vector: the 3 x (DW_PX + 2) input window is loaded once (4.5 vector loads per output
// instead of 9) and the 9 x EPL weights once per thread instead of once per pixel (the one-pixel form issued 81 load instructions per 16-byte output: 59 us per
// launch at 3.8 TB/s on YOLOv11m-seg's 80 x 80 towers).  Same products in the same order per output (taps outside the image contribute 0 * w instead of being skipped).
#define DW_PX 4
template <class T, int FLIP>
__global__ void __launch_bounds__(AD_THREADS)
dwconv3x3_kernel(const T* __restrict__ x, int x_ldc, int x_coff, int B, int H, int W, int C, const float* __restrict__ w,
                 T* __restrict__ y, int y_ldc, int y_coff, int accumulate) {
  constexpr int EPL = Elem<T>::EPL, PX = DW_PX;
  const int CG = C / EPL, WS = (W + PX - 1) / PX;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * H * WS * CG) return;
  const int c = (int)(i % CG) * EPL;
  const long seg = i / CG;
  const int w0 = (int)(seg % WS) * PX, hh = (int)((seg / WS) % H);
  const long b = seg / ((long)WS * H);
  float wt[9][EPL];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int e = 0; e < EPL; e++) wt[t][e] = w[t * C + c + e];   // flat-buffer offsets are not 16-byte aligned in general
  float acc[PX][EPL];
#pragma unroll
  for (int p = 0; p < PX; p++)
#pragma unroll
    for (int e = 0; e < EPL; e++) acc[p][e] = 0.f;
#pragma unroll
  for (int kh = 0; kh < 3; kh++) {
    const int ih = FLIP ? hh - kh + 1 : hh + kh - 1;
    const bool rok = ih >= 0 && ih < H;
    float f[PX + 2][EPL];
#pragma unroll
    for (int q = 0; q < PX + 2; q++) {
      const int iw = w0 - 1 + q;
      const bool ok = (bool)((int)rok & (int)(iw >= 0) & (int)(iw < W));
      const uint4 v = ys_ld16(x + ((b * H + (rok ? ih : 0)) * W + (ok ? iw : 0)) * x_ldc + x_coff + c);    // clamped address, masked value: no exec-masked loads
      ys_unpack<T>(ok ? v : ys_zero16(), f[q]);
    }
#pragma unroll
    for (int kw = 0; kw < 3; kw++)
#pragma unroll
      for (int p = 0; p < PX; p++) {
        const int q = FLIP ? p + 2 - kw : p + kw;            // input column w0 + p -+ (kw - 1)
#pragma unroll
        for (int e = 0; e < EPL; e++) acc[p][e] += f[q][e] * wt[kh * 3 + kw][e];
      }
  }
#pragma unroll
  for (int p = 0; p < PX; p++) {
    if (w0 + p >= W) continue;
    T* yp = y + ((b * H + hh) * (long)W + w0 + p) * y_ldc + y_coff + c;
    if (accumulate) {
      float o[EPL];
      ys_unpack<T>(ys_ld16(yp), o);
#pragma unroll
      for (int e = 0; e < EPL; e++) acc[p][e] += o[e];
    }
    ys_st16(yp, ys_pack<T>(acc[p]));
  }
}

int ys_dwconv_launch(hipStream_t st, int dtype, int flip, const void* x, int x_ldc, int x_coff, int B, int H, int W, int C,
                     const float* w, void* y, int y_ldc, int y_coff, int accumulate) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  if (C % epl) { ys_set_error("dwconv: C=%d must be a multiple of %d", C, epl); return YS_ERR_UNSUPPORTED; }
  const long n = (long)B * H * ((W + DW_PX - 1) / DW_PX) * (C / epl);
  const int g = ys_cdiv(n, AD_THREADS);
#define DW(TT, FL) YS_LAUNCH((dwconv3x3_kernel<TT, FL>), g, AD_THREADS, st, (const TT*)x, x_ldc, x_coff, B, H, W, C, w, (TT*)y, y_ldc, y_coff, accumulate)
  if (dtype == YS_BF16) { if (flip) DW(bf16_t, 1); else DW(bf16_t, 0); }
  else { if (flip) DW(float, 1); else DW(float, 0); }
#undef DW
  return YS_OK;
}

// dW[t][c] = sum_p dy[p,c] * x[p + off(t), c]: workgroup partials [nblk][9][C], summed in order by the finalize kernel
template <class T>
__global__ void __launch_bounds__(AD_THREADS)
dwconv_wgrad_kernel(const T* __restrict__ x, int x_ldc, int x_coff, const T* __restrict__ dy, int B, int H, int W, int C,
                    float* __restrict__ partial, unsigned xbytes) {
  constexpr int EPL = Elem<T>::EPL;
  __shared__ float sAcc[AD_THREADS][EPL];
  const int CG = C / EPL;
  const int RP = AD_THREADS / CG;
  const int tid = threadIdx.x;
  const int cv = tid % CG, rl = tid / CG;
  const int c = cv * EPL;
  const long rows = (long)B * H * W;
  const long rows_per_blk = (rows + gridDim.x - 1) / gridDim.x;
  const long r0 = (long)blockIdx.x * rows_per_blk;
  long r1 = r0 + rows_per_blk;
  if (r1 > rows) r1 = rows;
  float acc[9][EPL];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int e = 0; e < EPL; e++) acc[t][e] = 0.f;
  // the nine taps of a pixel are nine unconditional 16-byte loads through a buffer descriptor of the x view (a tap outside the image
  // = the out-of-range offset = zeros), all in flight together; the exec-masked `if (inside) load` form serialised them
  // (1.15 TB/s on the Detect cv3 depthwise layers of YOLOv11m)
  const ys_rsrcv_t rsX = ys_make_rsrcv((const char*)(x + x_coff), xbytes);
  if (rl < RP) {
    for (long row = r0 + rl; row < r1; row += RP) {
      const int ww = (int)(row % W), hh = (int)((row / W) % H);
      const long b = row / ((long)W * H);
      float g[EPL];
      ys_unpack<T>(ys_ld16(dy + row * C + c), g);
      uint4 xv[9];
#pragma unroll
      for (int kh = 0; kh < 3; kh++) {
        const int ih = hh + kh - 1;
#pragma unroll
        for (int kw = 0; kw < 3; kw++) {
          const int iw = ww + kw - 1;
          const bool ok = (bool)((int)((unsigned)ih < (unsigned)H) & (int)((unsigned)iw < (unsigned)W));
          xv[kh * 3 + kw] = ys_bufld16(rsX, ok ? (unsigned)((((b * H + ih) * W + iw) * x_ldc + c) * (long)sizeof(T)) : YS_BUF_OOB);
        }
      }
#pragma unroll
      for (int t = 0; t < 9; t++) {
        float f[EPL];
        ys_unpack<T>(xv[t], f);
#pragma unroll
        for (int e = 0; e < EPL; e++) acc[t][e] += g[e] * f[e];
      }
    }
  }
  for (int t = 0; t < 9; t++) {
#pragma unroll
    for (int e = 0; e < EPL; e++) sAcc[tid][e] = acc[t][e];
    __syncthreads();
    if (tid < CG) {
#pragma unroll
      for (int e = 0; e < EPL; e++) {
        float s = 0.f;
        for (int k = 0; k < RP; k++) s += sAcc[k * CG + tid][e];
        partial[((long)blockIdx.x * 9 + t) * C + c + e] = s;
      }
    }
    __syncthreads();
  }
}

// grad[i] += sum_k partial[k][i]: 32 outputs x 8 split lanes per workgroup (consecutive threads = consecutive outputs: coalesced
// rows), every lane walks its share of the nblk partials, the eight lane sums are combined in a fixed order -> deterministic.
// (One thread per output walking all partials serially: 155 us per layer at 1024 partials.)
__global__ void __launch_bounds__(AD_THREADS)
dwconv_wgrad_finalize_kernel(const float* __restrict__ partial, int nblk, int n, float* __restrict__ grad) {
  __shared__ float sred[AD_THREADS / 32][32];
  const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
  constexpr int NS = AD_THREADS / 32;
  const int i = blockIdx.x * 32 + o;
  float s0 = 0.f, s1 = 0.f;
  if (i < n) {
    int k = sl;
    for (; k + NS < nblk; k += 2 * NS) { s0 += partial[(long)k * n + i]; s1 += partial[(long)(k + NS) * n + i]; }
    if (k < nblk) s0 += partial[(long)k * n + i];
  }
  sred[sl][o] = s0 + s1;
  __syncthreads();
  if (sl == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < NS; q++) t += sred[q][o];
    grad[i] += t;
  }
}

int ys_dwconv_wgrad_blocks(long rows, int C, int dtype) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  const int rp = AD_THREADS / (C / epl);
  long nb = (rows + (long)rp * 8 - 1) / ((long)rp * 8);
  if (nb > 1024) nb = 1024;     // four workgroups per CU: a streaming pass wants the bytes in flight (256: 1.15 TB/s)
  if (nb < 1) nb = 1;
  return (int)nb;
}

int ys_dwconv_wgrad_launch(hipStream_t st, int dtype, const void* x, int x_ldc, int x_coff, const void* dy, int B, int H, int W,
                           int C, float* partial, float* grad) {
  const int epl = dtype == YS_BF16 ? 8 : 4;
  if (C % epl || C / epl > AD_THREADS) { ys_set_error("dwconv wgrad: unsupported C=%d", C); return YS_ERR_UNSUPPORTED; }
  const int nb = ys_dwconv_wgrad_blocks((long)B * H * W, C, dtype);
  const long xb = ((long)B * H * W * x_ldc - x_coff) * (dtype == YS_BF16 ? 2L : 4L);     // bytes of the x view from its first channel
  if (xb <= 0 || xb >= (1L << 31)) { ys_set_error("dwconv wgrad: input view of %ld bytes exceeds the 2 GB descriptor range", xb); return YS_ERR_UNSUPPORTED; }
  if (dtype == YS_BF16) YS_LAUNCH((dwconv_wgrad_kernel<bf16_t>), nb, AD_THREADS, st, (const bf16_t*)x, x_ldc, x_coff, (const bf16_t*)dy, B, H, W, C, partial, (unsigned)xb);
  else YS_LAUNCH((dwconv_wgrad_kernel<float>), nb, AD_THREADS, st, (const float*)x, x_ldc, x_coff, (const float*)dy, B, H, W, C, partial, (unsigned)xb);
  YS_LAUNCH(dwconv_wgrad_finalize_kernel, ys_cdiv(9 * C, 32), AD_THREADS, st, (const float*)partial, nb, 9 * C, grad);
  return YS_OK;
}

// ------------------------------------------------------------------ attention core
// qkv: [B][N][ldq] with channel layout head*(2*kd+hd) + {q: 0..kd | k: kd..2kd | v: 2kd..2kd+hd}  (Block.cs:772-775)
// ao : [B][N][ldo] at channel head*hd + d                                                        (view(B,C,H,W), :782)
// P  : [B*heads][N][N] fp32 softmax probabilities (kept for the backward)
#define ATT_NMAX 1600
template <class T>
__global__ void __launch_bounds__(AD_THREADS)
attn_fwd_kernel(const T* __restrict__ qkv, int ldq, int B, int N, int heads, int kd, int hd, float scale,
                T* __restrict__ ao, int ldo, float* __restrict__ P) {
  __shared__ float sP[AD_THREADS / 64][ATT_NMAX];
  __shared__ float sQ[AD_THREADS / 64][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.x * (AD_THREADS / 64) + wave;
  const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
  const int hs = 2 * kd + hd;
  if (n >= N) return;                       // whole wave (no workgroup barrier below)
  float* pr = sP[wave];
  float* sq = sQ[wave];
  const T* base = qkv + (long)b * N * ldq + h * hs;
  for (int d = lane; d < kd; d += 64) sq[d] = Elem<T>::to_f(base[(long)n * ldq + d]);
  ys_wave_sync();
  float mx = -INFINITY;
  for (int m = lane; m < N; m += 64) {
    const T* kr = base + (long)m * ldq + kd;
    float s = 0.f;
    for (int d = 0; d < kd; d++) s += sq[d] * Elem<T>::to_f(kr[d]);
    s *= scale;
    pr[m] = s;
    mx = fmaxf(mx, s);
  }
  mx = ys_wave_max(mx);
  float sum = 0.f;
  for (int m = lane; m < N; m += 64) { const float e = __expf(pr[m] - mx); pr[m] = e; sum += e; }
  sum = ys_wave_sum(sum);
  const float inv = 1.0f / sum;
  float* Prow = P + ((long)bh * N + n) * N;
  for (int m = lane; m < N; m += 64) { const float p = pr[m] * inv; pr[m] = p; Prow[m] = p; }
  ys_wave_sync();
  for (int d = lane; d < hd; d += 64) {
    float acc = 0.f;
    for (int m = 0; m < N; m++) acc += pr[m] * Elem<T>::to_f(base[(long)m * ldq + 2 * kd + d]);
    ao[((long)b * N + n) * ldo + h * hd + d] = Elem<T>::from_f(acc);
  }
}

// pass 1 (per query row n): dP = dO^T v ; dS = P o (dP - sum(dP o P)) ; dq = scale * dS k^T
template <class T>
__global__ void __launch_bounds__(AD_THREADS)
attn_bwd_q_kernel(const T* __restrict__ qkv, int ldq, int B, int N, int heads, int kd, int hd, float scale,
                  const T* __restrict__ dao, int ldo, const float* __restrict__ P, float* __restrict__ dS,
                  T* __restrict__ dqkv) {
  __shared__ float sS[AD_THREADS / 64][ATT_NMAX];
  __shared__ float sO[AD_THREADS / 64][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.x * (AD_THREADS / 64) + wave;
  const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
  const int hs = 2 * kd + hd;
  if (n >= N) return;
  float* ds = sS[wave];
  float* so = sO[wave];
  const T* base = qkv + (long)b * N * ldq + h * hs;
  for (int d = lane; d < hd; d += 64) so[d] = Elem<T>::to_f(dao[((long)b * N + n) * ldo + h * hd + d]);
  ys_wave_sync();
  const float* Prow = P + ((long)bh * N + n) * N;
  float t = 0.f;
  for (int m = lane; m < N; m += 64) {
    const T* vr = base + (long)m * ldq + 2 * kd;
    float dp = 0.f;
    for (int d = 0; d < hd; d++) dp += so[d] * Elem<T>::to_f(vr[d]);
    ds[m] = dp;
    t += dp * Prow[m];
  }
  t = ys_wave_sum(t);
  float* dSrow = dS + ((long)bh * N + n) * N;
  for (int m = lane; m < N; m += 64) { const float v = Prow[m] * (ds[m] - t); ds[m] = v; dSrow[m] = v; }
  ys_wave_sync();
  for (int d = lane; d < kd; d += 64) {
    float acc = 0.f;
    for (int m = 0; m < N; m++) acc += ds[m] * Elem<T>::to_f(base[(long)m * ldq + kd + d]);
    dqkv[((long)b * N + n) * ldq + h * hs + d] = Elem<T>::from_f(acc * scale);
  }
}

// ---- round 5: bf16, kd = 32, hd = 64 (every C2PSA of the YOLOv11 graphs at 640 x 640).  (A four-query-rows-per-wave scalar generation -- attn_fwd4_kernel /
// attn_bwd_q4_kernel: K / V of a head staged once per workgroup in LDS, bit-identical to the one-row kernels above, 412 -> 290 us forward -- was deleted in round 6
// once the MFMA kernels below had replaced it everywhere it ran; the one-row kernels remain for fp32, other head sizes and > ATT_NM tokens.)
#define ATT_QB 64            // query rows per workgroup (16 per wave)
#define ATT_KP 80            // LDS pitch of a K row: 64 B + 16 (consecutive rows shift by four banks: the 16-byte row reads of a wave do not collide)
#define ATT_VP 144           // ... of a V row: 128 B + 16
// ---- round 5, second step: the same two passes on the matrix cores (bf16, kd = 32, hd = 64, N <= 416 tokens = every C2PSA at 640 x 640).  One wave owns 16 query
// rows: S = Q K^T as one 16x16x32 MFMA per 16 keys (K rows straight from the staged LDS copy), softmax in the accumulator layout (a query row lives in the 16 lanes of
// a DPP row: four shuffles per reduction), the probabilities go to global memory in fp32 (the backward reads them) and as bf16 into a per-wave LDS tile that feeds the
// second product O = P V against a TRANSPOSED staged copy of V (the MFMA's K dimension is the key index).  The backward pass mirrors it: dP = dO V^T, dS = P (dP - t),
// dq = dS K against a transposed K.  The probabilities / dS enter the second product rounded to bf16 (the scalar kernels multiply them in fp32): a 2^-9 relative
// perturbation per term, below what the bf16 storage of the operands already costs; the parity tests of the block and the model hold their tolerances.
#define ATT_NM 416
__host__ __device__ inline int att_np16(int N) { return (N + 15) & ~15; }
__host__ __device__ inline int att_np32(int N) { return (N + 31) & ~31; }
__host__ __device__ inline int att_tp(int N) { return att_np32(N) * 2 + 16; }       // pitch (bytes) of a transposed row / a probability row: + 16 keeps 16-row fragment reads off one bank group
__device__ inline float att_row16_max(float v) { for (int m = 8; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m)); return v; }
__device__ inline float att_row16_sum(float v) { for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m); return v; }
// rows [0, np16) of a [N][W bf16] operand -> LDS rows of `pitch` bytes (rows >= N zero); src = first element of row 0, W / 8 16-byte units per row
__device__ inline void att_stage_rows(const bf16_t* src, int ldq, int N, int np16, int units, int pitch, char* dst) {
  for (int u = threadIdx.x; u < np16 * units; u += AD_THREADS) {
    const int m = u / units, c = u - m * units;
    *(uint4*)(dst + m * pitch + c * 16) = m < N ? *(const uint4*)(src + (long)m * ldq + c * 8) : ys_zero16();
  }
}
// the same operand transposed: LDS row d (0 .. 8 * units - 1) holds element d of keys 0 .. np32 - 1 (keys >= N zero)
__device__ inline void att_stage_transposed(const bf16_t* src, int ldq, int N, int np32, int units, int tp, char* dst) {
  for (int u = threadIdx.x; u < np32 * units; u += AD_THREADS) {
    const int c = u / np32, m = u - c * np32;
    const uint4 v = m < N ? *(const uint4*)(src + (long)m * ldq + c * 8) : ys_zero16();
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; j++) *(unsigned short*)(dst + (c * 8 + j) * tp + m * 2) = (unsigned short)(w[j >> 1] >> (16 * (j & 1)));
  }
}
#define ATT_TMAX (ATT_NM / 16)
__global__ void __launch_bounds__(AD_THREADS, 1)
attn_fwd_mfma_kernel(const bf16_t* __restrict__ qkv, int ldq, int B, int N, int heads, float scale, bf16_t* __restrict__ ao, int ldo, float* __restrict__ P) {
  constexpr int KD = 32, HD = 64, hs = 2 * KD + HD;
  YS_DYN_LDS(lds);
  const int np16 = att_np16(N), np32 = att_np32(N), tp = att_tp(N), T = np16 >> 4, KS = np32 >> 5;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, q4 = lane >> 4;
  const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
  const bf16_t* base = qkv + (long)b * N * ldq + h * hs;
  char* sK = (char*)lds;
  char* sVt = sK + (size_t)np16 * ATT_KP;
  char* sP = sVt + (size_t)HD * tp + (size_t)wave * 16 * tp;
  att_stage_rows(base + KD, ldq, N, np16, KD / 8, ATT_KP, sK);
  att_stage_transposed(base + 2 * KD, ldq, N, np32, HD / 8, tp, sVt);
  __syncthreads();
  const int n0 = blockIdx.x * ATT_QB + wave * 16;
  if (n0 >= N) return;                        // wave-uniform; no workgroup barrier below
  const int arow = n0 + li < N ? n0 + li : N - 1;
  const uint4 aq = *(const uint4*)(base + (long)arow * ldq + q4 * 8);
  f32x4 sacc[ATT_TMAX];
#pragma unroll
  for (int t = 0; t < ATT_TMAX; t++) {
    sacc[t] = f32x4_zero();
    if (t < T) sacc[t] = mfma_16x16x32_bf16(aq, *(const uint4*)(sK + (16 * t + li) * ATT_KP + q4 * 16), sacc[t]);
  }
  float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int t = 0; t < ATT_TMAX; t++) {
    if (t < T) {
      const bool colok = 16 * t + li < N;
#pragma unroll
      for (int r = 0; r < 4; r++) { const float v = colok ? sacc[t][r] * scale : -INFINITY; sacc[t][r] = v; mx[r] = fmaxf(mx[r], v); }
    }
  }
  float sum[4];
#pragma unroll
  for (int r = 0; r < 4; r++) { mx[r] = att_row16_max(mx[r]); sum[r] = 0.f; }
#pragma unroll
  for (int t = 0; t < ATT_TMAX; t++) {
    if (t < T) {
#pragma unroll
      for (int r = 0; r < 4; r++) { const float e = __expf(sacc[t][r] - mx[r]); sacc[t][r] = e; sum[r] += e; }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) sum[r] = 1.0f / att_row16_sum(sum[r]);
#pragma unroll
  for (int t = 0; t < ATT_TMAX; t++) {
    if (t < T) {
      const int col = 16 * t + li;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float pv = sacc[t][r] * sum[r];
        const int row = n0 + 4 * q4 + r;
        if (row < N && col < N) P[((long)bh * N + row) * N + col] = pv;
        *(unsigned short*)(sP + (4 * q4 + r) * tp + col * 2) = Elem<bf16_t>::from_f(pv).v;
      }
    }
  }
  if (np32 > np16) *(uint2*)(sP + li * tp + (np16 + 4 * q4) * 2) = make_uint2(0u, 0u);     // columns np16 .. np32 - 1 of the last K-step
  ys_wave_sync();
  f32x4 oacc[HD / 16];
#pragma unroll
  for (int dt = 0; dt < HD / 16; dt++) oacc[dt] = f32x4_zero();
  for (int ks = 0; ks < KS; ks++) {
    const uint4 ap = *(const uint4*)(sP + li * tp + (32 * ks + 8 * q4) * 2);
#pragma unroll
    for (int dt = 0; dt < HD / 16; dt++)
      oacc[dt] = mfma_16x16x32_bf16(ap, *(const uint4*)(sVt + (16 * dt + li) * tp + (32 * ks + 8 * q4) * 2), oacc[dt]);
  }
#pragma unroll
  for (int dt = 0; dt < HD / 16; dt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = n0 + 4 * q4 + r;
      if (row < N) ao[((long)b * N + row) * ldo + h * HD + 16 * dt + li] = Elem<bf16_t>::from_f(oacc[dt][r]);
    }
}

__global__ void __launch_bounds__(AD_THREADS, 1)
attn_bwd_q_mfma_kernel(const bf16_t* __restrict__ qkv, int ldq, int B, int N, int heads, float scale, const bf16_t* __restrict__ dao, int ldo,
                       const float* __restrict__ P, float* __restrict__ dS, bf16_t* __restrict__ dqkv) {
  constexpr int KD = 32, HD = 64, hs = 2 * KD + HD;
  YS_DYN_LDS(lds);
  const int np16 = att_np16(N), np32 = att_np32(N), tp = att_tp(N), T = np16 >> 4, KS = np32 >> 5;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, q4 = lane >> 4;
  const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
  const bf16_t* base = qkv + (long)b * N * ldq + h * hs;
  char* sV = (char*)lds;
  char* sKt = sV + (size_t)np16 * ATT_VP;
  char* sS = sKt + (size_t)KD * tp + (size_t)wave * 16 * tp;
  att_stage_rows(base + 2 * KD, ldq, N, np16, HD / 8, ATT_VP, sV);
  att_stage_transposed(base + KD, ldq, N, np32, KD / 8, tp, sKt);
  __syncthreads();
  const int n0 = blockIdx.x * ATT_QB + wave * 16;
  if (n0 >= N) return;
  const int arow = n0 + li < N ? n0 + li : N - 1;
  const bf16_t* dorow = dao + ((long)b * N + arow) * ldo + h * HD;
  const uint4 ao0 = *(const uint4*)(dorow + q4 * 8), ao1 = *(const uint4*)(dorow + 32 + q4 * 8);
  f32x4 dp[ATT_TMAX];
  float pv[ATT_TMAX][4];
#pragma unroll
  for (int t = 0; t < ATT_TMAX; t++) {
    dp[t] = f32x4_zero();
    if (t < T) {
      const char* vr = sV + (16 * t + li) * ATT_VP;
      dp[t] = mfma_16x16x32_bf16(ao0, *(const uint4*)(vr + q4 * 16), dp[t]);
      dp[t] = mfma_16x16x32_bf16(ao1, *(const uint4*)(vr + 64 + q4 * 16), dp[t]);
      const int col = 16 * t + li;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = n0 + 4 * q4 + r;
        pv[t][r] = (row < N && col < N) ? P[((long)bh * N + row) * N + col] : 0.f;
      }
    }
  }
  float tr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < ATT_TMAX; t++)
    if (t < T) {
#pragma unroll
      for (int r = 0; r < 4; r++) tr[r] += dp[t][r] * pv[t][r];
    }
#pragma unroll
  for (int r = 0; r < 4; r++) tr[r] = att_row16_sum(tr[r]);
#pragma unroll
  for (int t = 0; t < ATT_TMAX; t++) {
    if (t < T) {
      const int col = 16 * t + li;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float v = pv[t][r] * (dp[t][r] - tr[r]);
        const int row = n0 + 4 * q4 + r;
        if (row < N && col < N) dS[((long)bh * N + row) * N + col] = v;
        *(unsigned short*)(sS + (4 * q4 + r) * tp + col * 2) = Elem<bf16_t>::from_f(v).v;
      }
    }
  }
  if (np32 > np16) *(uint2*)(sS + li * tp + (np16 + 4 * q4) * 2) = make_uint2(0u, 0u);
  ys_wave_sync();
  f32x4 qacc[KD / 16];
#pragma unroll
  for (int dt = 0; dt < KD / 16; dt++) qacc[dt] = f32x4_zero();
  for (int ks = 0; ks < KS; ks++) {
    const uint4 as = *(const uint4*)(sS + li * tp + (32 * ks + 8 * q4) * 2);
#pragma unroll
    for (int dt = 0; dt < KD / 16; dt++)
      qacc[dt] = mfma_16x16x32_bf16(as, *(const uint4*)(sKt + (16 * dt + li) * tp + (32 * ks + 8 * q4) * 2), qacc[dt]);
  }
#pragma unroll
  for (int dt = 0; dt < KD / 16; dt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = n0 + 4 * q4 + r;
      if (row < N) dqkv[((long)b * N + row) * ldq + h * hs + 16 * dt + li] = Elem<bf16_t>::from_f(qacc[dt][r] * scale);
    }
}
// pass 2 on the matrix cores: a wave owns 16 key columns; dk = scale dS^T q and dv += P^T dO walk the query rows 32 at a time.  The A operand (rows = keys, K = query
// index) is gathered from the fp32 dS / P matrices in global memory -- eight 4-byte loads per lane and K-step, 64-byte segments per row -- and rounded to bf16; the B
// operands are q and dO staged TRANSPOSED in LDS.
__global__ void __launch_bounds__(AD_THREADS, 1)
attn_bwd_kv_mfma_kernel(const bf16_t* __restrict__ qkv, int ldq, int B, int N, int heads, float scale, const bf16_t* __restrict__ dao, int ldo,
                        const float* __restrict__ P, const float* __restrict__ dS, bf16_t* __restrict__ dqkv) {
  constexpr int KD = 32, HD = 64, hs = 2 * KD + HD;
  YS_DYN_LDS(lds);
  const int np32 = att_np32(N), tp = att_tp(N), KS = np32 >> 5;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, q4 = lane >> 4;
  const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
  const bf16_t* base = qkv + (long)b * N * ldq + h * hs;
  char* sQt = (char*)lds;
  char* sOt = sQt + (size_t)KD * tp;
  att_stage_transposed(base, ldq, N, np32, KD / 8, tp, sQt);
  att_stage_transposed(dao + (long)b * N * ldo + h * HD, ldo, N, np32, HD / 8, tp, sOt);
  __syncthreads();
  const int m0 = blockIdx.x * ATT_QB + wave * 16;
  if (m0 >= N) return;
  const int mcol = m0 + li;                    // this lane's key column of the A operands
  const float* Sb = dS + (long)bh * N * N + mcol;
  const float* Pb = P + (long)bh * N * N + mcol;
  f32x4 kacc[KD / 16], vacc[HD / 16];
#pragma unroll
  for (int dt = 0; dt < KD / 16; dt++) kacc[dt] = f32x4_zero();
#pragma unroll
  for (int dt = 0; dt < HD / 16; dt++) vacc[dt] = f32x4_zero();
  for (int ks = 0; ks < KS; ks++) {
    float fs[8], fp[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int n = 32 * ks + 8 * q4 + j;
      const bool ok = (bool)((int)(n < N) & (int)(mcol < N));
      const long o = ok ? (long)n * N : 0;
      const float a = Sb[ok ? o : -(long)mcol], c = Pb[ok ? o : -(long)mcol];      // (clamped to element 0 of the matrix: an unconditional load)
      fs[j] = ok ? a : 0.f; fp[j] = ok ? c : 0.f;
    }
    const uint4 as = ys_pack<bf16_t>(fs), ap = ys_pack<bf16_t>(fp);
#pragma unroll
    for (int dt = 0; dt < KD / 16; dt++)
      kacc[dt] = mfma_16x16x32_bf16(as, *(const uint4*)(sQt + (16 * dt + li) * tp + (32 * ks + 8 * q4) * 2), kacc[dt]);
#pragma unroll
    for (int dt = 0; dt < HD / 16; dt++)
      vacc[dt] = mfma_16x16x32_bf16(ap, *(const uint4*)(sOt + (16 * dt + li) * tp + (32 * ks + 8 * q4) * 2), vacc[dt]);
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int m = m0 + 4 * q4 + r;
    if (m < N) {
      bf16_t* row = dqkv + ((long)b * N + m) * ldq + h * hs;
#pragma unroll
      for (int dt = 0; dt < KD / 16; dt++) row[KD + 16 * dt + li] = Elem<bf16_t>::from_f(kacc[dt][r] * scale);
#pragma unroll
      for (int dt = 0; dt < HD / 16; dt++) { bf16_t* dst = row + 2 * KD + 16 * dt + li; *dst = Elem<bf16_t>::from_f(vacc[dt][r] + Elem<bf16_t>::to_f(*dst)); }
    }
  }
}
static bool attn_mfma_ok(int dtype, int ldq, int ldo, int N, int heads, int kd, int hd) {
  return dtype == YS_BF16 && kd == 32 && hd == 64 && (ldq & 7) == 0 && (ldo & 7) == 0 && N >= 1 && N <= ATT_NM && YS_OPT_INT("ATTN_MFMA", 1) != 0;
}
// dynamic LDS above 64 KB needs the attribute once per (kernel, device): cached like the convolution launchers' (it was set on every launch and its result dropped)
template <class K>
static int attn_lds_attr(K kernel, size_t bytes, std::atomic<unsigned>& done) {
  if (bytes <= 64 * 1024) return YS_OK;
  int dev = 0;
  YS_CHECK_HIP(hipGetDevice(&dev));
  const unsigned bit = 1u << (dev & 31);
  if (done.load(std::memory_order_acquire) & bit) return YS_OK;
  YS_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  done.fetch_or(bit, std::memory_order_acq_rel);
  return YS_OK;
}

// pass 2 (per key column m): dk = scale * dS^T q ; dv = dO P  (added to the gradient that arrived through pe(v)).
// A workgroup owns 64 key columns of one (image, head) and walks the query rows in chunks of 32: the dS / P tiles [32][64] are
// read as full rows (coalesced) into LDS together with the q and dO rows of the chunk, then thread (m, d-slice) accumulates its
// outputs over the chunk -- rows in ascending order, the same summation order as one thread per output walking a column.
// (The round-1 form read dS and P column-wise, one 4-byte element per 1600-byte stride: 1.59 ms on BASELINE config 4.)
#define AKV_TM 64
#define AKV_TN 32
template <class T, int KD4, int HD4>         // accumulators per thread: kd / 4 and hd / 4 (4 waves split the d axis)
__global__ void __launch_bounds__(AD_THREADS)
attn_bwd_kv_kernel(const T* __restrict__ qkv, int ldq, int B, int N, int heads, int kd, int hd, float scale,
                   const T* __restrict__ dao, int ldo, const float* __restrict__ P, const float* __restrict__ dS,
                   T* __restrict__ dqkv) {
  __shared__ float sS[AKV_TN][AKV_TM], sPt[AKV_TN][AKV_TM];
  __shared__ float sQ[AKV_TN][KD4 * 4], sO[AKV_TN][HD4 * 4];
  const int tid = threadIdx.x, wave = tid >> 6, ml = tid & 63;
  const int m0 = blockIdx.x * AKV_TM, m = m0 + ml;
  const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
  const int hs = 2 * kd + hd;
  const T* base = qkv + (long)b * N * ldq + h * hs;
  const float* Pb = P + (long)bh * N * N;
  const float* Sb = dS + (long)bh * N * N;
  float ak[KD4], av[HD4];
#pragma unroll
  for (int j = 0; j < KD4; j++) ak[j] = 0.f;
#pragma unroll
  for (int j = 0; j < HD4; j++) av[j] = 0.f;
  for (int n0 = 0; n0 < N; n0 += AKV_TN) {
    __syncthreads();
    for (int i = tid; i < AKV_TN * AKV_TM; i += AD_THREADS) {
      const int r = i / AKV_TM, c = i - r * AKV_TM;
      const bool ok = n0 + r < N && m0 + c < N;
      sS[r][c] = ok ? Sb[(long)(n0 + r) * N + m0 + c] : 0.f;
      sPt[r][c] = ok ? Pb[(long)(n0 + r) * N + m0 + c] : 0.f;
    }
    for (int i = tid; i < AKV_TN * kd; i += AD_THREADS) {
      const int r = i / kd, d = i - r * kd;
      sQ[r][d] = n0 + r < N ? Elem<T>::to_f(base[(long)(n0 + r) * ldq + d]) : 0.f;
    }
    for (int i = tid; i < AKV_TN * hd; i += AD_THREADS) {
      const int r = i / hd, d = i - r * hd;
      sO[r][d] = n0 + r < N ? Elem<T>::to_f(dao[((long)b * N + n0 + r) * ldo + h * hd + d]) : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < AKV_TN; r++) {
      const float sv = sS[r][ml], pv = sPt[r][ml];
#pragma unroll
      for (int j = 0; j < KD4; j++) ak[j] += sv * sQ[r][wave + 4 * j];
#pragma unroll
      for (int j = 0; j < HD4; j++) av[j] += pv * sO[r][wave + 4 * j];
    }
  }
  if (m < N) {
#pragma unroll
    for (int j = 0; j < KD4; j++) {
      const int d = wave + 4 * j;
      if (d < kd) dqkv[((long)b * N + m) * ldq + h * hs + kd + d] = Elem<T>::from_f(ak[j] * scale);
    }
#pragma unroll
    for (int j = 0; j < HD4; j++) {
      const int d = wave + 4 * j;
      if (d < hd) {
        T* dst = dqkv + ((long)b * N + m) * ldq + h * hs + 2 * kd + d;
        *dst = Elem<T>::from_f(av[j] + Elem<T>::to_f(*dst));
      }
    }
  }
}

template <class T>
static int attn_bwd_kv_dispatch(hipStream_t st, const T* qkv, int ldq, int B, int N, int heads, int kd, int hd, float scale, const T* dao,
                                int ldo, const float* P, const float* dS, T* dqkv) {
  dim3 grid(ys_cdiv(N, AKV_TM), B * heads);
#define AKV(K_, H_) if (kd <= 4 * K_ && hd <= 4 * H_) { YS_LAUNCH((attn_bwd_kv_kernel<T, K_, H_>), grid, AD_THREADS, st, qkv, ldq, B, N, heads, kd, hd, scale, dao, ldo, P, dS, dqkv); return YS_OK; }
  AKV(8, 16) AKV(16, 32) AKV(32, 64)
#undef AKV
  ys_set_error("attention backward: kd=%d hd=%d outside the supported range", kd, hd);
  return YS_ERR_UNSUPPORTED;
}

int ys_attn_fwd_launch(hipStream_t st, int dtype, const void* qkv, int ldq, int B, int N, int heads, int kd, int hd,
                       void* ao, int ldo, float* P) {
  if (N > ATT_NMAX || kd > 128 || hd > 256) { ys_set_error("attention: N=%d kd=%d hd=%d outside the supported range", N, kd, hd); return YS_ERR_UNSUPPORTED; }
  const float scale = 1.0f / sqrtf((float)kd);   // Math.Pow(key_dim, -0.5) (Block.cs:733)
  if (attn_mfma_ok(dtype, ldq, ldo, N, heads, kd, hd)) {
    const size_t lb = (size_t)att_np16(N) * ATT_KP + (size_t)(64 + 64) * att_tp(N);
    static std::atomic<unsigned> attr_f{0};
    YS_TRY(attn_lds_attr(attn_fwd_mfma_kernel, lb, attr_f));
    YS_LAUNCH_LDS(attn_fwd_mfma_kernel, dim3(ys_cdiv(N, ATT_QB), B * heads), AD_THREADS, lb, st, (const bf16_t*)qkv, ldq, B, N, heads, scale, (bf16_t*)ao, ldo, P);
    return YS_OK;
  }
  dim3 grid(ys_cdiv(N, AD_THREADS / 64), B * heads);
  if (dtype == YS_BF16) YS_LAUNCH((attn_fwd_kernel<bf16_t>), grid, AD_THREADS, st, (const bf16_t*)qkv, ldq, B, N, heads, kd, hd, scale, (bf16_t*)ao, ldo, P);
  else YS_LAUNCH((attn_fwd_kernel<float>), grid, AD_THREADS, st, (const float*)qkv, ldq, B, N, heads, kd, hd, scale, (float*)ao, ldo, P);
  return YS_OK;
}

int ys_attn_bwd_launch(hipStream_t st, int dtype, const void* qkv, int ldq, int B, int N, int heads, int kd, int hd,
                       const void* dao, int ldo, const float* P, float* dS, void* dqkv) {
  const float scale = 1.0f / sqrtf((float)kd);
  if (attn_mfma_ok(dtype, ldq, ldo, N, heads, kd, hd)) {
    const size_t lb = (size_t)att_np16(N) * ATT_VP + (size_t)(32 + 64) * att_tp(N);
    static std::atomic<unsigned> attr_q{0}, attr_kv{0};
    YS_TRY(attn_lds_attr(attn_bwd_q_mfma_kernel, lb, attr_q));
    YS_LAUNCH_LDS(attn_bwd_q_mfma_kernel, dim3(ys_cdiv(N, ATT_QB), B * heads), AD_THREADS, lb, st, (const bf16_t*)qkv, ldq, B, N, heads, scale, (const bf16_t*)dao, ldo, P, dS, (bf16_t*)dqkv);
    const size_t lk = (size_t)(32 + 64) * att_tp(N);
    YS_TRY(attn_lds_attr(attn_bwd_kv_mfma_kernel, lk, attr_kv));
    YS_LAUNCH_LDS(attn_bwd_kv_mfma_kernel, dim3(ys_cdiv(N, ATT_QB), B * heads), AD_THREADS, lk, st, (const bf16_t*)qkv, ldq, B, N, heads, scale, (const bf16_t*)dao, ldo, P, (const float*)dS, (bf16_t*)dqkv);
    return YS_OK;
  }
  dim3 grid(ys_cdiv(N, AD_THREADS / 64), B * heads);
  if (dtype == YS_BF16) {
    YS_LAUNCH((attn_bwd_q_kernel<bf16_t>), grid, AD_THREADS, st, (const bf16_t*)qkv, ldq, B, N, heads, kd, hd, scale, (const bf16_t*)dao, ldo, P, dS, (bf16_t*)dqkv);
    YS_TRY(attn_bwd_kv_dispatch<bf16_t>(st, (const bf16_t*)qkv, ldq, B, N, heads, kd, hd, scale, (const bf16_t*)dao, ldo, P, (const float*)dS, (bf16_t*)dqkv));
  } else {
    YS_LAUNCH((attn_bwd_q_kernel<float>), grid, AD_THREADS, st, (const float*)qkv, ldq, B, N, heads, kd, hd, scale, (const float*)dao, ldo, P, dS, (float*)dqkv);
    YS_TRY(attn_bwd_kv_dispatch<float>(st, (const float*)qkv, ldq, B, N, heads, kd, hd, scale, (const float*)dao, ldo, P, (const float*)dS, (float*)dqkv));
  }
  return YS_OK;
}

// v channels of qkv -> contiguous [B*N][C] (input of Attention.pe) and back (gradient of that view)
template <class T>
__global__ void __launch_bounds__(AD_THREADS)
attn_v_copy_kernel(const T* __restrict__ src, T* __restrict__ dst, long rows, int ldq, int heads, int kd, int hd, int ldv, int to_qkv) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int C = heads * hd;
  if (i >= rows * C) return;
  const long row = i / C;
  const int c = (int)(i - row * C);
  const int h = c / hd, d = c - h * hd;
  const long qi = row * ldq + h * (2 * kd + hd) + 2 * kd + d;
  if (to_qkv) dst[qi] = src[row * ldv + c];
  else dst[row * ldv + c] = src[qi];
}
int ys_attn_v_copy_launch(hipStream_t st, int dtype, const void* src, void* dst, long rows, int ldq, int heads, int kd, int hd,
                          int ldv, int to_qkv) {
  const long n = rows * heads * hd;
  if (dtype == YS_BF16) YS_LAUNCH((attn_v_copy_kernel<bf16_t>), ys_cdiv(n, AD_THREADS), AD_THREADS, st, (const bf16_t*)src, (bf16_t*)dst, rows, ldq, heads, kd, hd, ldv, to_qkv);
  else YS_LAUNCH((attn_v_copy_kernel<float>), ys_cdiv(n, AD_THREADS), AD_THREADS, st, (const float*)src, (float*)dst, rows, ldq, heads, kd, hd, ldv, to_qkv);
  return YS_OK;
}
