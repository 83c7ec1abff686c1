// ys_kernels.h -- argument blocks and host launchers of the device kernels.
#pragma once
#include "ys_hip.h"

// Fused BN-backward reduction (round 3).  When a dgrad launch is the LAST writer of the gradient dz of a BN Conv unit's output
// (the unit's first consumer in forward order), its epilogue -- which holds the finished dz values -- also accumulates that
// unit's per-channel sums  sum(du), sum(du * y),  du = dz * SiLU'(scale * y + shift)  (what chan_reduce_kernel<MODE 0> computes
// in a pass of its own over dz and y): one statistics row per workgroup, like the forward BN statistics.  A segment maps the
// channels [c0, c1) of the launch's OUTPUT view onto channels [yc0, ...) of one producer.
#define YS_BNRED_MAXSEG 3
#define YS_GROUP_MAX 4        // problems of one grouped convolution launch (the three pyramid levels of a head; the four phases of a stride-2 dgrad)
#define YS_EW_GROUP_MAX 9     // problems of one grouped elementwise launch (BN passes of up to three towers x three levels)
struct BnRedSeg {
  const void* y;        // producer's raw conv output, dense [M][C]
  const float* scale;   // producer's BN scale / shift (forward, training statistics), [C]
  const float* shift;
  float* part;          // partial rows [rows][2][C] of this (producer, launch) pair
  int c0, c1;           // output-view channel range of the launch covered by this producer (multiples of 8)
  int yc0;              // producer channel of c0
  int C;                // producer channel count
  int act;              // producer applies SiLU
  int pad_;
};
struct ConvArgs {
  const void* x;        // input activations (NHWC view)
  const void* w;        // weights [Cout][KH*KW][Cin], storage type T
  void* y;              // output (NHWC view)
  const float* scale;   // optional per-cout multiplier (eval BN)          } v = acc*scale + shift
  const float* shift;   // optional per-cout offset (eval BN shift / bias) }
  const void* res;      // optional residual added after the activation (same row mapping as y)
  float* stats;         // optional [gridM][2][Cout] partial (sum, sumsq) of the stored values
  unsigned long long* stat_acc;   // round 5: when set (with stats != null as the "take statistics" flag), a workgroup ADDS its sums -- as 2^-20 fixed point, 64-bit integer
                        // atomics, order-independent = deterministic -- into [YS_STAT_SHARDS][Cout][2] accumulators instead of writing a row; the BN + SiLU apply pass
                        // finalizes from them (ys_bn_fin_apply_launch) and the bn_finalize launch of the unit goes
  int B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW;
  int SA, DIVS, DIVM, PAD;   // ih*DIV = oh*SA + kh - PAD ; DIVS = log2(DIV), DIVM = DIV-1
  int in_ldc, in_coff;
  long in_bstride;           // pixels between consecutive images of the input buffer
  int out_ldc, out_coff;
  long out_bstride;          // rows between consecutive images in the output addressing
  int res_ldc, res_coff;
  int act;                   // 1 = SiLU after scale/shift
  int accumulate;            // y += result (gradient accumulation)
  int vec_ok;                // out_ldc/out_coff (and res) allow 4-channel vector stores
  int M;                     // B*Hout*Wout
  int TH, TW, tiles_x, tiles_y;  // 2-D output tile of the 3x3 LDS-patch kernel (filled by the launcher)
  int dbg;                       // ablation switches for performance triage (YS_DBG env; 0 in production)
  // ConvTranspose2d(2,2) phases (Proto.upsample, Block.cs:69) reuse the 1x1 direct kernel with two generalisations:
  int pad_w_delta;               // width gather uses PAD + pad_w_delta (phase (dh,dw): PAD=-dh, delta=dh-dw)
  int out_rh, out_rw;            // out_rh != 0: output row = b*out_bstride + oh*out_rh + ow*out_rw + out_r0
  long out_r0;                   //   (writes phase (dh,dw) of a 2x upsampled grid: rh = 4W, rw = 2, r0 = dh*2W + dw)
  // fp8 convolution path (conv_p2_kernel<F8 = 1>): the bf16 input is quantised while its patch is staged in LDS
  int f8;                        // 0 = off; 1 = e4m3 input (forward); 2 = e5m2 input (dgrad: the input is dy)
  const void* w8;                // fp8 (e4m3) weights, same element order as w
  const float* qscale;           // device scalar: quantisation multiplier of the input tensor
  const float* deq;              // device scalar: 1 / (input scale * weight scale), applied to the fp32 accumulators
  unsigned* amax;                // optional: YS_AMAX_WAYS slots receiving amax(|input|) of this launch (next step's scale)
  // fp8 blocked-GEMM path (conv_gemm_kernel<F8 = 1>): the operand tiles go to LDS by DMA, so the input must already be fp8 in memory
  void* q8;                      // optional scratch of >= B*Hin*Win*Cin bytes: ys_conv_launch quantises the input view into it first
  const void* x8;                // set by ys_conv_launch: the dense [B*Hin*Win][Cin] fp8 image of the input (inside q8)
  unsigned long long* tl;        // triage builds (-DYS_P2_TIMELINE): per-workgroup s_memtime stamps of the tile phases; null otherwise
  // fused BN-backward reduction (dgrad launches through conv_epi.h only; see BnRedSeg)
  int nred;                      // segments in use (0 = off)
  int red_row0;                  // first partial row of this launch (the four phase launches of a stride-2 dgrad stack their rows)
  int red_koff;                  // set by the launchers: byte offset of `red` in the launch's kernel-argument segment (conv_epi.h ys_red_table)
  BnRedSeg red[YS_BNRED_MAXSEG];
};

struct WgradArgs {
  const void* x;   // layer input (NHWC view)
  const void* dy;  // gradient w.r.t. the conv output, [M][dy_ldc]
  float* partial;  // [splits][Cout][KH*KW][Cin] fp32
  int B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad;
  int in_ldc, in_coff;
  long in_bstride;
  int dy_ldc, dy_coff;
  long dy_bstride;  // rows between consecutive images of dy (Hout*Wout when dense)
  int M;
  int dy_rh, dy_rw; // dy_rh != 0: dy row = b*dy_bstride + oh*dy_rh + ow*dy_rw + dy_r0 (ConvTranspose phases)
  long dy_r0;
  // tile geometry of the bf16 LDS-tile kernel (filled by the launcher)
  int TH, TWS, tiles_x, tiles_y, ntiles, PH, PW, pdb, pxb;
  unsigned dybytes, xbytes;   // descriptor ranges of the dy / x views (filled by the launcher)
};

int ys_conv_launch(hipStream_t st, int dtype, const ConvArgs& a);
// blocked-GEMM kernel for the wide bf16 layers (conv_gemm.hip): statistics rows of its launch, 0 when the layer is not eligible
int ys_f8_quant_view_launch(hipStream_t st, int fmt, const void* x, long rows, int C, int ldc, int coff, const float* qscale,
                            void* out, unsigned* slots);
int ys_conv_gemm_rows(const ConvArgs& a);
// true when an fp8 request (a.f8 set, a.x8 not yet) would run conv_gemm_kernel<F8> -- on the layer itself or on a phase of its
// stride-2 dgrad -- if the dense fp8 image of the input were supplied in a.x8: the caller may then produce that image in the pass
// that writes the bf16 tensor (ys_bn_bwd_apply_q8_launch) instead of leaving it to ys_conv_launch's quantisation pass
bool ys_conv_wants_x8(const ConvArgs& a);
int ys_bn_act_apply_q8_launch(hipStream_t st, const void* y, long rows, int C, const float* scale, const float* shift, int act,
                              const void* res, int res_ldc, int res_coff, void* z, int z_ldc, int z_coff, void* q8,
                              const float* qscale, unsigned* amax);
int ys_bn_bwd_apply_q8_launch(hipStream_t st, const void* dz, int dz_ldc, int dz_coff, const void* y, long rows, int C,
                              const float* scale, const float* shift, const float* k2, const float* k3, int act, void* dy,
                              void* q8, const float* qscale, unsigned* amax, void* rg = nullptr, int rg_ldc = 0, int rg_coff = 0);
int ys_conv_gemm_launch(hipStream_t st, const ConvArgs& a);
// model.0 read straight from the fp32 NCHW image (conv_stem.hip): forward (training: raw output + statistics rows; eval: folded BatchNorm
// + SiLU) and weight gradient (partial slabs [Cout][9][8] in the generic kernel's split layout)
bool ys_stem_eligible(int dtype, int cin, int cout, int k, int s);
int ys_stem_fwd_rows(int B, int Hout, int Wout);
int ys_stem_fwd_launch(hipStream_t st, const float* x, int B, int H, int W, const void* wf, int Cout, void* y, int out_ldc, int out_coff,
                       long out_bstride, float* stats, const float* scale, const float* shift, int act, int* rows, unsigned long long* stat_acc = nullptr);
int ys_stem_wgrad_launch(hipStream_t st, const float* x, int B, int H, int W, const void* dy, int dy_ldc, int dy_coff, long dy_bstride,
                         int Cout, float* partial, int max_splits, int* used);
int ys_conv_grid_m(const ConvArgs& a, int dtype);
int ys_conv_is_p2(const ConvArgs& a);
// partial rows a launch of `a` (a dgrad with a.nred segments) writes per segment when every kernel it dispatches to supports the fused
// BN-backward reduction (conv_p2_kernel / conv_gemm_kernel, incl. the phase launches of a stride-2 dgrad); 0 = not supported
int ys_conv_bnred_rows(const ConvArgs& a, int dtype);
int ys_wgrad_splits(const WgradArgs& a, int dtype);
// used_splits != nullptr: the split reduction is left to the caller (ys_wgrad_reduce_batched_launch over the layers of a backward
// segment); *used_splits = number of partial slabs [Cout][taps][Cin] written to a.partial
int ys_wgrad_launch(hipStream_t st, int dtype, const WgradArgs& a, int splits, int cin_real, float* grad, int* used_splits = nullptr);
// one launch reducing the partial slabs of several layers: grad[row * cin_real + ci] += sum_s partial[s][row * cin_pad + ci]
#define YS_WGRED_OUT_PER_BLOCK 512   // outputs per workgroup (a multiple of 128) of wgrad_reduce_batched_kernel (blk0 prefix of the descriptors)
struct WgRedDesc { const float* partial; float* grad; long n; long blk0; int splits, cin_pad, cin_real, pad_; };
int ys_wgrad_reduce_batched_launch(hipStream_t st, const WgRedDesc* descs_dev, int n_desc, long total_blocks);
// blocked-GEMM weight-gradient kernel of the wide bf16 layers (conv_wgrad_gemm.hip): pixel splits it wants (0 = not eligible);
// the launch uses at most `splits` and returns the number used (0 = not eligible, nothing launched)
int ys_wgrad_gemm_splits(const WgradArgs& a);
int ys_wgrad_gemm_launch(hipStream_t st, const WgradArgs& a, int splits);
int ys_weight_prep_launch(hipStream_t st, int dtype, const float* w, int Cout, int taps, int cin_real, int cin_pad,
                          int cout_pad, void* wf, void* wd, int phase);
// true when the dgrad of a (k, stride) layer runs as four phase convolutions and wants the phase-major dgrad weights
bool ys_conv_dgrad_uses_phases(int dtype, int k, int stride);
// element e of a phase-major dgrad weight block [phase][Cin_real][taps_p][Cout_pad] (3x3, stride 2) -> (ci, tap, co)
__host__ __device__ inline void ys_phase_wd_index(long e, int cin_real, int cout_pad, int& ci, int& tap, int& co) {
  const long blk = (long)cin_real * cout_pad;
  const int ph = e < blk ? 0 : (e < 3 * blk ? 1 : (e < 5 * blk ? 2 : 3));
  const long start = (ph == 0 ? 0 : (ph == 1 ? 1 : (ph == 2 ? 3 : 5))) * blk;
  const int pa = ph >> 1, pb = ph & 1;
  const int kwn = pb ? 2 : 1, tp_n = (pa ? 2 : 1) * kwn;
  const long el = e - start;
  co = (int)(el % cout_pad);
  const long r = el / cout_pad;
  const int tp = (int)(r % tp_n);
  ci = (int)(r / tp_n);
  const int khp = tp / kwn, kwp = tp - khp * kwn;
  const int kh = pa == 0 ? 1 : (khp == 0 ? 2 : 0);
  const int kw = pb == 0 ? 1 : (kwp == 0 ? 2 : 0);
  tap = kh * 3 + kw;
}


// ---- elementwise.hip
// NCHW fp32 -> NHWC T with channels zero-padded to cpad
int ys_pack_input_launch(hipStream_t st, int dtype, const float* x_nchw, int B, int C, int H, int W, int cpad, void* y);
// uint8 NCHW [B,C,h,w] in 0..255 -> NHWC T [B,H,W,cpad]: / 255, bottom / right padding to (H, W) with 114 / 255
int ys_pack_input_u8_launch(hipStream_t st, int dtype, const unsigned char* x, int B, int C, int h, int w, int H, int W, int cpad, void* y);
// LetterBox / Rectangle resize + pad (Augment.cs:698-857): planes [C][h][w] -> [C][H][W]; element = uint8 or fp32
int ys_letterbox_launch(hipStream_t st, int is_float, const void* x, int C, int h, int w, void* y, int H, int W, int new_h, int new_w,
                        int pad_u, int pad_l, float color);
// NHWC view T -> NCHW fp32
int ys_unpack_nchw_launch(hipStream_t st, int dtype, const void* x, int ldc, int coff, int B, int C, long rows_per_b,
                          float* y_nchw);
// BN (training) statistics finalize: partial [nblk][2][C] -> scale/shift (+ saved mean, rstd) and running-stat update
int ys_bn_finalize_launch(hipStream_t st, const float* partial, int nblk, int C, long count, const float* gamma,
                          const float* beta, float eps, float momentum, float* run_mean, float* run_var,
                          float* nbt, float* scale, float* shift, float* mean, float* rstd);
// eval: scale/shift from running stats
int ys_bn_eval_coeffs_launch(hipStream_t st, int C, const float* gamma, const float* beta, const float* run_mean,
                             const float* run_var, float eps, float* scale, float* shift);
// Statistics as fixed-point integer sums (round 5).  Scale 2^20: a workgroup's partial sum is quantised to 2^-20 ~ 1e-6 absolute (rounded; <= 768 partials per
// launch), the 64-bit accumulator holds |sum| up to 2^42 = 4.4e12 -- e.g. 6.5 million pixels (the largest map of the BASELINE configs) at a mean square of 6.7e5 (a single
// workgroup partial is bounded at 2^32, see the poison rule below).
#define YS_STAT_SHARDS 8          // accumulator copies (workgroup index modulo): ~64 arrivals per cache line and launch instead of ~512
#define YS_STAT_FIX 1048576.0f
// A partial that is not finite, or too large for the fixed point (|t| >= 2^32: 768 such partials would pass 2^62), POISONS the channel: bit 62 of its sum-of-squares
// word is raised (legitimate sums of squares are non-negative and below 2^62, so later adds cannot clear it) and bn_fin_apply_kernel turns the flag into NaN mean /
// variance -- what the row path and the reference do on divergence (an unguarded fptosi of NaN / Inf is a garbage FINITE integer that would be folded into
// run_mean / run_var: ADVICE r5).
#define YS_STAT_POISON (1ull << 62)
__device__ inline void ys_stat_acc_add(unsigned long long* acc, long stat_row, int C, int c, int which, float t) {
  unsigned long long* a = acc + (((stat_row & (YS_STAT_SHARDS - 1)) * C + c) * 2);
  if (!(fabsf(t) < 4294967296.0f)) { atomicOr(a + 1, YS_STAT_POISON); return; }     // NaN fails the comparison too
#ifdef YS_EMU_BUILD
  const long long q = (long long)llrintf(t * YS_STAT_FIX);
#else
  const long long q = __float2ll_rn(t * YS_STAT_FIX);
#endif
  if (q != 0) atomicAdd(a + which, (unsigned long long)q);
}
// finalize-inside-apply operands: the accumulators of one BatchNorm unit + what bn_finalize_kernel reads and writes
struct BnAccFin {
  const unsigned long long* acc;   // [YS_STAT_SHARDS][C][2]
  double count;
  const float* gamma; const float* beta;
  float* run_mean; float* run_var; float* nbt;
  float* scale; float* shift; float* mean; float* rstd;   // written by workgroup 0 (the backward pass reads them)
  float eps, momentum;
};
// z[out view] = act(BN(y)) (+ residual view) with the batch statistics finalized from the accumulators by every workgroup (identical arithmetic, identical result);
// workgroup 0 stores the coefficients and updates the running statistics
int ys_bn_fin_apply_launch(hipStream_t st, int dtype, const void* y, long rows, int C, const BnAccFin& f, int act, const void* res, int res_ldc, int res_coff,
                           void* z, int z_ldc, int z_coff);
// z[out view] = act(y*scale+shift) (+ residual view)
int ys_bn_act_apply_launch(hipStream_t st, int dtype, const void* y, long rows, int C, const float* scale,
                           const float* shift, int act, const void* res, int res_ldc, int res_coff, void* z,
                           int z_ldc, int z_coff, unsigned* amax = nullptr);
// backward of z = act(BN(y)): pass 1 partial sums of du and du*xhat; optionally res_grad += dz
int ys_bn_bwd_reduce_launch(hipStream_t st, int dtype, const void* dz, int dz_ldc, int dz_coff, const void* y, long rows,
                            int C, const float* scale, const float* shift, const float* mean, const float* rstd,
                            int act, void* res_grad, int rg_ldc, int rg_coff, float* partial, int* nblk_out);
// pass 1b: dgamma += sum(du*xhat), dbeta += sum(du), coefficients for pass 2
int ys_bn_bwd_finalize_launch(hipStream_t st, const float* partial, int nblk, int C, long count, float* dgamma,
                              float* dbeta, float* k2, float* k3, const float* scale, const float* mean, const float* rstd);
// pass 2: dy = gamma*rstd*(du - mean(du) - xhat*mean(du*xhat)) = scale*du - k2 - y*k3
// rg != nullptr: also rg[view] += dz (the residual-input gradient of a Bottleneck shortcut)
int ys_bn_bwd_apply_launch(hipStream_t st, int dtype, const void* dz, int dz_ldc, int dz_coff, const void* y, long rows,
                           int C, const float* scale, const float* shift, const float* k2, const float* k3, int act, void* dy,
                           unsigned* amax = nullptr, void* rg = nullptr, int rg_ldc = 0, int rg_coff = 0);
// finalize reading each channel's partial rows from the source that covers it (fused BN-backward reduction)
struct FinSrc { const float* p[YS_BNRED_MAXSEG]; int nblk[YS_BNRED_MAXSEG]; int c1[YS_BNRED_MAXSEG]; int n; };
int ys_bn_bwd_finalize_src_launch(hipStream_t st, const FinSrc& src, int C, long count, float* dgamma, float* dbeta, float* k2,
                                  float* k3, const float* scale, const float* mean, const float* rstd);
// column sums of a [rows][ldc] view into grad[C] (+=)   (bias gradients)
// ---- grouped (multi-problem) launches: independent Conv units of the same kind side by side in one grid (elementwise.hip, conv.hip)
struct BnFinProb { const float* partial; int nblk, C; double count; const float *gamma, *beta; float *run_mean, *run_var, *nbt, *scale, *shift, *mean, *rstd; };
struct BnFinGroup { int n; int end[YS_EW_GROUP_MAX]; BnFinProb p[YS_EW_GROUP_MAX]; };
struct BnApplyProb { const void* y; long rows; int C; const float *scale, *shift; const void* res; int res_ldc, res_coff; void* z; int z_ldc, z_coff; };
struct BnApplyGroup { int n; int end[YS_EW_GROUP_MAX]; BnApplyProb p[YS_EW_GROUP_MAX]; };
struct ChanFinProb { FinSrc src; int C; double count; float *g0, *g1, *c1, *c2; const float *scale, *mean, *rstd; };
struct ChanFinGroup { int n; int end[YS_EW_GROUP_MAX]; ChanFinProb p[YS_EW_GROUP_MAX]; };
struct BnBwdProb { const void* dz; int dz_ldc, dz_coff; const void* y; long rows; int C; const float *scale, *shift, *k2, *k3; void* dy; void* rg; int rg_ldc, rg_coff; };
struct BnBwdGroup { int n; int end[YS_EW_GROUP_MAX]; BnBwdProb p[YS_EW_GROUP_MAX]; };
int ys_bn_finalize_group_launch(hipStream_t st, const BnFinProb* probs, int n, float eps, float momentum);
int ys_bn_act_apply_group_launch(hipStream_t st, int dtype, const BnApplyProb* probs, int n, int act);
int ys_bn_bwd_finalize_group_launch(hipStream_t st, const ChanFinProb* probs, int n);
int ys_bn_bwd_apply_group_launch(hipStream_t st, int dtype, const BnBwdProb* probs, int n, int act);
// n <= YS_GROUP_MAX independent P2 convolutions of one shape class as ONE persistent grid; YS_ERR_UNSUPPORTED = launch them one by one.
// row_cap[i] > 0 bounds problem i's workgroups; rows[i] = workgroups (= statistics / BN-reduction partial rows) problem i got
// plan_only: no launch, only the row counts (a dgrad's BN-reduction segments need them before the launch)
int ys_conv_p2_group_launch(hipStream_t st, const ConvArgs* a, int n, const int* row_cap, int* rows, bool plan_only = false);
int ys_colsum_launch(hipStream_t st, int dtype, const void* x, int ldc, int coff, long rows, long rows_per_b,
                     long bstride, int C, float* partial, float* grad);
int ys_colsum_blocks(long rows, int C, int dtype);
// 5x5/s1/p2 max-pool on an NHWC view (+ argmax byte per output element for the backward)
int ys_maxpool5_fwd_launch(hipStream_t st, int dtype, const void* x, int x_ldc, int x_coff, int B, int H, int W, int C,
                           void* y, int y_ldc, int y_coff, unsigned char* argmax);
int ys_maxpool5_bwd_launch(hipStream_t st, int dtype, const void* dy, int dy_ldc, int dy_coff, int B, int H, int W,
                           int C, const unsigned char* argmax, void* dx, int dx_ldc, int dx_coff, int accumulate);
// nearest 2x upsample into a (concat) view, and its backward (2x2 sum)
int ys_upsample2x_fwd_launch(hipStream_t st, int dtype, const void* x, int x_ldc, int x_coff, int B, int H, int W, int C,
                             void* y, int y_ldc, int y_coff);
int ys_upsample2x_bwd_launch(hipStream_t st, int dtype, const void* dy, int dy_ldc, int dy_coff, int B, int H, int W,
                             int C, void* dx, int dx_ldc, int dx_coff, int accumulate);
// dst view (+)= src view
int ys_copy_view_launch(hipStream_t st, int dtype, const void* src, int s_ldc, int s_coff, long rows, int C, void* dst,
                        int d_ldc, int d_coff, int accumulate);
// AdamW over a flat range
#define YS_ADAMW_MAX_RANGES 12
struct AdamwRanges { int n; long off[YS_ADAMW_MAX_RANGES]; long count[YS_ADAMW_MAX_RANGES]; float lr[YS_ADAMW_MAX_RANGES]; };
// parameters listed in two optimizer groups (mask[i] != 0): a second update with lr_second; bias corrections of the two step indices
struct AdamwDup { const unsigned char* mask; float lr_second, bc1_first, bc2s_first, bc1_second, bc2s_second; };
int ys_adamw_ranges_launch(hipStream_t st, float* p, const float* g, float* m, float* v, long n, const AdamwRanges& rg,
                           float beta1, float beta2, float eps, float wd, float bc1, float bc2, const AdamwDup* dup = nullptr);
int ys_adamw_launch(hipStream_t st, float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                    float beta2, float eps, float wd, float bc1, float bc2);
int ys_fill_launch(hipStream_t st, float* p, long n, float v);

int ys_chan_stats_launch(hipStream_t st, int dtype, const void* y, long rows, int C, float* partial, int* nblk_out);

// ---- f8.hip (scaling machinery of the fp8 convolution path)
struct F8Layer { long w_off, count; };                       // fp32 master weights of one conv layer in the flat parameter buffer
struct F8Conv { int layer; };                                // index of the layer's weight-amax slot
int ys_f8_weight_amax_launch(hipStream_t st, const float* params, const F8Layer* layers, int n, float* amax_w);
int ys_f8_scales_launch(hipStream_t st, const F8Conv* convs, int n, const float* amax_w, unsigned* amax_x, unsigned* amax_dy,
                        float* out /*[n][4]: s_x, deq_fwd, s_g, deq_dgrad*/);
// amax(|x|) of a [rows][C] bf16 view (channel stride ldc, offset coff) into YS_AMAX_WAYS slots (bootstrap of the delayed scales)
int ys_f8_view_amax_launch(hipStream_t st, const void* x_bf16, long rows, int C, int ldc, int coff, unsigned* slots);
int ys_f8_quant_weights_launch(hipStream_t st, const void* w_bf16, long n, const float* amax_w, void* w8);

// ---- attn_dw.hip (YOLOv11 operators)
int ys_dwconv_launch(hipStream_t st, int dtype, int flip, const void* x, int x_ldc, int x_coff, int B, int H, int W, int C,
                     const float* w /*[9][C]*/, void* y, int y_ldc, int y_coff, int accumulate);
int ys_dwconv_wgrad_blocks(long rows, int C, int dtype);
int ys_dwconv_wgrad_launch(hipStream_t st, int dtype, const void* x, int x_ldc, int x_coff, const void* dy, int B, int H, int W,
                           int C, float* partial /*[blocks][9][C]*/, float* grad /*[9][C] +=*/);
int ys_attn_fwd_launch(hipStream_t st, int dtype, const void* qkv, int ldq, int B, int N, int heads, int kd, int hd,
                       void* ao, int ldo, float* P);
int ys_attn_bwd_launch(hipStream_t st, int dtype, const void* qkv, int ldq, int B, int N, int heads, int kd, int hd,
                       const void* dao, int ldo, const float* P, float* dS, void* dqkv);
int ys_attn_v_copy_launch(hipStream_t st, int dtype, const void* src, void* dst, long rows, int ldq, int heads, int kd, int hd,
                          int ldv, int to_qkv);

// ---- loss.hip
struct LossArgs {
  const void* pd;  // box logits   [B][A][ld_pd]  (4*reg_max used)
  const void* ps;  // class logits [B][A][ld_ps]  (nc used)
  void* dpd;       // gradients, same layouts
  void* dps;
  int ld_pd, ld_ps;
  int B, A, nc, reg_max;
  int H, W;            // input image size (imgsz = feats[0].shape[2:]*stride[0])
  int nl;              // levels
  int lvl_off[4], lvl_w[4], lvl_h[4], lvl_stride[4];
  int gmax;   // largest per-image label count when the host knows it (host labels), else gcap: the (box, image) grid of tal_metrics_kernel
  // labels (device): raw collate arrays
  const float* batch_idx; const float* cls; const float* bboxes; int n_labels;
  int gcap;            // GT capacity per image
  // workspace (device)
  int* gt_count;       // [B]
  float* gt_box;       // [B][gcap][4] xyxy pixels
  int* gt_cls;         // [B][gcap]   (followed by gt_valid [B][gcap] and gt_src [B][gcap] in the same allocation)
  float* pbox;         // [B][A][4] decoded pred box, grid units (xyxy)
  float* ov;           // [B][gcap][A]
  float* align;        // [B][gcap][A]
  unsigned char* mpos; // [B][gcap][A]
  unsigned* pos_align; // [B][gcap] float bits (non-negative -> uint order)
  unsigned* pos_ov;    // [B][gcap]
  int* fg_gt;          // [B][A]  assigned gt index or -1
  float* tnorm;        // [B][A]  normalised alignment (target score value)
  float* partial;      // [nblk][4] partial sums
  float* scalars;      // [8]: tss, loss_box, loss_cls, loss_dfl, total
  float hyp_box, hyp_cls, hyp_dfl;
  int topk;
  // v8OBBLoss (Loss.cs:486-684): rot = 1 -> labels carry (cx, cy, w, h, angle), gt_box / pbox rows are 5 floats (xywh + angle),
  // the assigner uses probiou and the rotated in-box test, the box term is 1 - probiou, plus the angle term (scalars[12..13])
  int rot;
  const void* pa;      // angle logits [B][A][ld_pa] (channel 0); angle = (sigmoid - 0.25) * pi (Head.cs:429)
  void* dpa;           // gradient w.r.t. the logit
  int ld_pa;
  float hyp_angle;
};
int ys_loss_detect_launch(hipStream_t st, int dtype, const LossArgs& a);
size_t ys_loss_partial_floats(int B, int A);

// ---- decode (Head.cs:204-223)
int ys_detect_decode_launch(hipStream_t st, int dtype, const void* pd, int ld_pd, const void* ps, int ld_ps, int B, int A,
                            int nc, int reg_max, int nl, const int* lvl_off, const int* lvl_w, const int* lvl_stride,
                            float* pred, int pred_C, const void* px, int ld_px, int xkind, int nx, int kdim);
// xkind 2: Obb decode (dist2rbox with the angle logit px[.][0], Head.cs:435-438, Tal.cs:389-408); 3: Pose.kpts_decode of the nx
// keypoint outputs (Head.cs:590-605); 0/1: plain Detect decode.  In place (sigmoid(p) - 0.25) * pi (Head.cs:429):
int ys_obb_angle_launch(hipStream_t st, float* p, long n);
int ys_unpack_nchw_strided_launch(hipStream_t st, int dtype, const void* x, int ldc, int coff, int B, int C, long rpb, float* y,
                                  long y_bstride, long y_off);
// ---- segloss.hip
int ys_loss_segment_launch(hipStream_t st, int dtype, const void* mc, void* dmc, int ld_mc, const void* proto, void* dproto, int ld_pr,
                           const float* masks, const int* fg_gt, const float* gt_box, int* cnt, int* off, int* list, float* ent,
                           float* part, float* scalars, int B, int A, int nm, int mh, int mw, int gcap, int H, int W, int trunc_crop);
// foreground anchors of the last assignment, ordered by image then anchor: cnt [B], off [B+1], list [off[B]]
int ys_fg_list_launch(hipStream_t st, const int* fg_gt, int B, int A, int* cnt, int* off, int* list);
// ---- poseloss.hip: keypoint terms of v8PoseLoss (Loss.cs:870-1071) after ys_loss_detect_launch
#define YS_POSE_KMAX 64
struct PoseArgs {
  const void* kp;          // raw keypoint outputs [B][A][ld] (K*D used)
  void* dkp;               // gradient, same layout
  int ld;
  const int* fg_gt;        // [B][A] assigned GT slot or -1
  const float* gt_box;     // [B][gcap][4] xyxy pixels
  const int* gt_src;       // [B][gcap] label row of the slot
  const float* keypoints;  // [N][K][D] normalised (x, y[, visibility])
  const int* off; const int* list;
  float* part;             // [grid][2]
  float* scalars;          // [10] pose, [11] kobj, [4] total
  int B, A, K, D, gcap, H, W, nl;
  int lvl_off[4], lvl_w[4], lvl_stride[4];
  float hyp_pose, hyp_kobj;
  float sigma[YS_POSE_KMAX];
};
int ys_loss_pose_grid(int B, int A);
int ys_loss_pose_launch(hipStream_t st, int dtype, const PoseArgs& a, int* cnt, int* off, int* list);
int ys_process_mask_launch(hipStream_t st, const float* protos, const float* masks_in, const float* boxes, int n, int nm, int mh,
                           int mw, int ih, int iw, int upsample, int trunc_crop, unsigned char* out);
