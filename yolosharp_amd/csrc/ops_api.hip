// ops_api.hip -- per-operator entry points of the C ABI (unit parity; a TorchSharp-free Conv wrapper).
#include "ys_internal.h"
#include "ys_kernels.h"
#include <vector>

namespace {
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  int alloc(size_t n) { hipError_t e = hipMalloc(&p, n ? n : 16); return e == hipSuccess ? YS_OK : YS_ERR_OOM; }
};
}  // namespace

extern "C" int ys_conv_bn_act_fwd(ys_ctx* ctx, int dtype, const float* x_nchw, int B, int Cin, int H, int W,
                                  const float* w_oihw, int Cout, int k, int stride, const float* bn_gamma,
                                  const float* bn_beta, float* bn_mean, float* bn_var, const float* bias,
                                  int act_silu, int training, float* y_nchw) {
  YS_REQUIRE(ctx && x_nchw && w_oihw && y_nchw, "ys_conv_bn_act_fwd: null argument");
  YS_REQUIRE(dtype == YS_F32 || dtype == YS_BF16 || dtype == YS_FP8, "ys_conv_bn_act_fwd: bad dtype %d", dtype);
  YS_REQUIRE((k == 1 || k == 3) && (stride == 1 || stride == 2), "ys_conv_bn_act_fwd: k=%d stride=%d unsupported", k, stride);
  // YS_FP8: bf16 storage; the convolution runs the fp8 MFMA kernel with CURRENT per-tensor scales of this call's own tensors
  // (s_w = 448 / amax|W|, s_x = 0.5 * 448 / amax|bf16(x)|, the recipe of f8.hip) when Cin % 32 == 0, else the bf16 kernel
  const bool want_f8 = dtype == YS_FP8;
  if (want_f8) dtype = YS_BF16;
  const bool has_bn = bn_gamma != nullptr;
  YS_REQUIRE(!has_bn || (bn_beta && bn_mean && bn_var), "ys_conv_bn_act_fwd: incomplete BN arguments");
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int epl = dtype == YS_BF16 ? 8 : 4;
  const size_t es = dtype == YS_BF16 ? 2 : 4;
  const int cpad = (Cin + epl - 1) / epl * epl;
  const int pad = k / 2;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const int taps = k * k;
  const long M = (long)B * Ho * Wo;
  const int cout_ld = (Cout + epl - 1) / epl * epl;

  // host: OIHW -> [Cout][taps][Cin]
  std::vector<float> wint((size_t)Cout * taps * Cin);
  for (int co = 0; co < Cout; co++)
    for (int ci = 0; ci < Cin; ci++)
      for (int t = 0; t < taps; t++) wint[((size_t)co * taps + t) * Cin + ci] = w_oihw[((size_t)co * Cin + ci) * taps + t];

  DevBuf dx, dxn, dwm, dwf, dy, dz, dstat, dpar, dout;
  YS_TRY(dx.alloc((size_t)B * Cin * H * W * 4));
  YS_TRY(dxn.alloc((size_t)B * H * W * cpad * es));
  YS_TRY(dwm.alloc(wint.size() * 4));
  YS_TRY(dwf.alloc((size_t)Cout * taps * cpad * es));
  YS_TRY(dy.alloc((size_t)M * cout_ld * es));
  YS_TRY(dz.alloc((size_t)M * cout_ld * es));
  YS_TRY(dout.alloc((size_t)M * Cout * 4));
  YS_CHECK_HIP(hipMemsetAsync(dy.p, 0, (size_t)M * cout_ld * es, st));
  YS_CHECK_HIP(hipMemsetAsync(dz.p, 0, (size_t)M * cout_ld * es, st));
  YS_CHECK_HIP(hipMemcpyAsync(dx.p, x_nchw, (size_t)B * Cin * H * W * 4, hipMemcpyHostToDevice, st));
  YS_CHECK_HIP(hipMemcpyAsync(dwm.p, wint.data(), wint.size() * 4, hipMemcpyHostToDevice, st));
  YS_TRY(ys_pack_input_launch(st, dtype, (const float*)dx.p, B, Cin, H, W, cpad, dxn.p));
  YS_TRY(ys_weight_prep_launch(st, dtype, (const float*)dwm.p, Cout, taps, Cin, cpad, cout_ld, dwf.p, nullptr, 0));

  // per-channel parameter block: gamma, beta, rmean, rvar, scale, shift, mean, rstd, bias, nbt
  YS_TRY(dpar.alloc((size_t)Cout * 10 * 4));
  float* par = (float*)dpar.p;
  float *g = par, *bt = par + Cout, *rm = par + 2 * Cout, *rv = par + 3 * Cout, *sc = par + 4 * Cout, *sh = par + 5 * Cout,
        *mu = par + 6 * Cout, *rs = par + 7 * Cout, *bs = par + 8 * Cout, *nbt = par + 9 * Cout;
  if (has_bn) {
    YS_CHECK_HIP(hipMemcpyAsync(g, bn_gamma, Cout * 4, hipMemcpyHostToDevice, st));
    YS_CHECK_HIP(hipMemcpyAsync(bt, bn_beta, Cout * 4, hipMemcpyHostToDevice, st));
    YS_CHECK_HIP(hipMemcpyAsync(rm, bn_mean, Cout * 4, hipMemcpyHostToDevice, st));
    YS_CHECK_HIP(hipMemcpyAsync(rv, bn_var, Cout * 4, hipMemcpyHostToDevice, st));
  }
  if (bias) YS_CHECK_HIP(hipMemcpyAsync(bs, bias, Cout * 4, hipMemcpyHostToDevice, st));
  YS_CHECK_HIP(hipMemsetAsync(nbt, 0, Cout * 4, st));

  ConvArgs a{};
  a.x = dxn.p; a.w = dwf.p;
  DevBuf dw8, dq, dq8;
  if (want_f8 && Cin % 32 == 0) {
    YS_TRY(dw8.alloc((size_t)Cout * taps * cpad));
    YS_TRY(dq8.alloc((size_t)B * H * W * cpad));   // fp8 image of the input for the blocked-GEMM kernel (quantised by ys_conv_launch)
    YS_TRY(dq.alloc(1024));                     // floats: [0] amax_w, [16..80) amax_x ways, [96..160) amax_dy ways, [200..204) scales
    YS_CHECK_HIP(hipMemsetAsync(dq.p, 0, 1024, st));
    float* qf = (float*)dq.p;
    F8Layer hl{0, (long)Cout * taps * Cin};
    F8Conv hc{0};
    DevBuf dl, dc;
    YS_TRY(dl.alloc(sizeof hl)); YS_TRY(dc.alloc(sizeof hc));
    YS_CHECK_HIP(hipMemcpyAsync(dl.p, &hl, sizeof hl, hipMemcpyHostToDevice, st));
    YS_CHECK_HIP(hipMemcpyAsync(dc.p, &hc, sizeof hc, hipMemcpyHostToDevice, st));
    YS_TRY(ys_f8_weight_amax_launch(st, (const float*)dwm.p, (const F8Layer*)dl.p, 1, qf));
    YS_TRY(ys_f8_view_amax_launch(st, dxn.p, (long)B * H * W, cpad, cpad, 0, (unsigned*)(qf + 16)));
    YS_TRY(ys_f8_scales_launch(st, (const F8Conv*)dc.p, 1, qf, (unsigned*)(qf + 16), (unsigned*)(qf + 96), qf + 200));
    YS_TRY(ys_f8_quant_weights_launch(st, dwf.p, (long)Cout * taps * cpad, qf, dw8.p));
    YS_CHECK_HIP(hipStreamSynchronize(st));     // hl / hc / dl / dc are temporaries of this scope
    a.f8 = 1; a.w8 = dw8.p; a.qscale = qf + 200; a.deq = qf + 201; a.q8 = dq8.p;
  }
  a.B = B; a.Hin = H; a.Win = W; a.Cin = cpad; a.Hout = Ho; a.Wout = Wo; a.Cout = Cout; a.KH = k; a.KW = k;
  a.SA = stride; a.DIVS = 0; a.DIVM = 0; a.PAD = pad;
  a.in_ldc = cpad; a.in_coff = 0; a.in_bstride = (long)H * W;
  a.out_ldc = cout_ld; a.out_coff = 0; a.out_bstride = (long)Ho * Wo;
  a.vec_ok = 1; a.M = (int)M;
  const void* result = nullptr;
  if (has_bn && training) {
    const int gm = ys_conv_grid_m(a, dtype);
    YS_TRY(dstat.alloc((size_t)gm * 2 * Cout * 4));
    a.y = dy.p; a.stats = (float*)dstat.p;
    YS_TRY(ys_conv_launch(st, dtype, a));
    YS_TRY(ys_bn_finalize_launch(st, (const float*)dstat.p, gm, Cout, M, g, bt, 1e-3f, 0.03f, rm, rv, nbt, sc, sh, mu, rs));
    YS_REQUIRE(Cout % epl == 0, "ys_conv_bn_act_fwd: training BN needs Cout %% %d == 0", epl);
    YS_TRY(ys_bn_act_apply_launch(st, dtype, dy.p, M, Cout, sc, sh, act_silu, nullptr, 0, 0, dz.p, cout_ld, 0));
    result = dz.p;
    YS_CHECK_HIP(hipMemcpyAsync(bn_mean, rm, Cout * 4, hipMemcpyDeviceToHost, st));
    YS_CHECK_HIP(hipMemcpyAsync(bn_var, rv, Cout * 4, hipMemcpyDeviceToHost, st));
  } else {
    if (has_bn) {
      YS_TRY(ys_bn_eval_coeffs_launch(st, Cout, g, bt, rm, rv, 1e-3f, sc, sh));
      a.scale = sc; a.shift = sh;
    } else if (bias) {
      a.shift = bs;
    }
    a.act = act_silu && (a.scale || a.shift) ? 1 : 0;
    YS_REQUIRE(!(act_silu && !a.scale && !a.shift), "ys_conv_bn_act_fwd: activation without BN/bias is not a reference configuration");
    a.y = dz.p;
    YS_TRY(ys_conv_launch(st, dtype, a));
    result = dz.p;
  }
  YS_TRY(ys_unpack_nchw_launch(st, dtype, result, cout_ld, 0, B, Cout, (long)Ho * Wo, (float*)dout.p));
  YS_CHECK_HIP(hipMemcpyAsync(y_nchw, dout.p, (size_t)M * Cout * 4, hipMemcpyDeviceToHost, st));
  YS_CHECK_HIP(hipStreamSynchronize(st));
  YS_CHECK_HIP(hipGetLastError());
  return YS_OK;
}

// Gradients of y = conv2d(x, w, stride, padding=k/2) given dy (autograd of torch.nn.Conv2d, reached from
// Amp.cs:348,370): dx [B,Cin,H,W] (may be null) and dw [Cout,Cin,k,k], all fp32 NCHW/OIHW host arrays.
extern "C" int ys_conv_bwd(ys_ctx* ctx, int dtype, const float* x_nchw, int B, int Cin, int H, int W,
                           const float* w_oihw, int Cout, int k, int stride, const float* dy_nchw,
                           float* dx_nchw, float* dw_oihw) {
  YS_REQUIRE(ctx && x_nchw && w_oihw && dy_nchw && dw_oihw, "ys_conv_bwd: null argument");
  YS_REQUIRE(dtype == YS_F32 || dtype == YS_BF16 || dtype == YS_FP8, "ys_conv_bwd: bad dtype %d", dtype);
  const bool want_f8 = dtype == YS_FP8;           // dgrad on the fp8 kernel (dy -> e5m2, weights -> e4m3, current scales); wgrad stays bf16
  if (want_f8) dtype = YS_BF16;
  YS_REQUIRE((k == 1 || k == 3) && (stride == 1 || stride == 2), "ys_conv_bwd: k=%d stride=%d unsupported", k, stride);
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int epl = dtype == YS_BF16 ? 8 : 4;
  const size_t es = dtype == YS_BF16 ? 2 : 4;
  const int cpad = (Cin + epl - 1) / epl * epl, copad = (Cout + epl - 1) / epl * epl;
  const int pad = k / 2, taps = k * k;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const long M = (long)B * Ho * Wo;
  std::vector<float> wint((size_t)Cout * taps * Cin);
  for (int co = 0; co < Cout; co++)
    for (int ci = 0; ci < Cin; ci++)
      for (int t = 0; t < taps; t++) wint[((size_t)co * taps + t) * Cin + ci] = w_oihw[((size_t)co * Cin + ci) * taps + t];
  DevBuf dx, dxn, ddy, ddyn, dwm, dwf, dwd, dgx, dgxo, dgw, dpart;
  YS_TRY(dx.alloc((size_t)B * Cin * H * W * 4));
  YS_TRY(dxn.alloc((size_t)B * H * W * cpad * es));
  YS_TRY(ddy.alloc((size_t)M * Cout * 4));
  YS_TRY(ddyn.alloc((size_t)M * copad * es));
  YS_TRY(dwm.alloc(wint.size() * 4));
  YS_TRY(dwf.alloc((size_t)Cout * taps * cpad * es));
  YS_TRY(dwd.alloc((size_t)Cin * taps * copad * es));
  YS_TRY(dgx.alloc((size_t)B * H * W * cpad * es));
  YS_TRY(dgxo.alloc((size_t)B * Cin * H * W * 4));
  YS_TRY(dgw.alloc(wint.size() * 4));
  YS_CHECK_HIP(hipMemsetAsync(dgx.p, 0, (size_t)B * H * W * cpad * es, st));
  YS_CHECK_HIP(hipMemsetAsync(dgw.p, 0, wint.size() * 4, st));
  YS_CHECK_HIP(hipMemcpyAsync(dx.p, x_nchw, (size_t)B * Cin * H * W * 4, hipMemcpyHostToDevice, st));
  YS_CHECK_HIP(hipMemcpyAsync(ddy.p, dy_nchw, (size_t)M * Cout * 4, hipMemcpyHostToDevice, st));
  YS_CHECK_HIP(hipMemcpyAsync(dwm.p, wint.data(), wint.size() * 4, hipMemcpyHostToDevice, st));
  YS_TRY(ys_pack_input_launch(st, dtype, (const float*)dx.p, B, Cin, H, W, cpad, dxn.p));
  YS_TRY(ys_pack_input_launch(st, dtype, (const float*)ddy.p, B, Cout, Ho, Wo, copad, ddyn.p));
  YS_TRY(ys_weight_prep_launch(st, dtype, (const float*)dwm.p, Cout, taps, Cin, cpad, copad, dwf.p, dwd.p, ys_conv_dgrad_uses_phases(dtype, k, stride) ? 1 : 0));
  WgradArgs wa{};
  wa.x = dxn.p; wa.dy = ddyn.p;
  wa.B = B; wa.Hin = H; wa.Win = W; wa.Cin = cpad; wa.Hout = Ho; wa.Wout = Wo; wa.Cout = Cout; wa.KH = wa.KW = k;
  wa.stride = stride; wa.pad = pad; wa.in_ldc = cpad; wa.in_coff = 0; wa.in_bstride = (long)H * W;
  wa.dy_ldc = copad; wa.dy_coff = 0; wa.dy_bstride = (long)Ho * Wo; wa.M = (int)M;
  const int splits = ys_wgrad_splits(wa, dtype);
  YS_TRY(dpart.alloc((size_t)splits * Cout * taps * cpad * 4));
  wa.partial = (float*)dpart.p;
  YS_TRY(ys_wgrad_launch(st, dtype, wa, splits, Cin, (float*)dgw.p));
  std::vector<float> gw(wint.size());
  YS_CHECK_HIP(hipMemcpyAsync(gw.data(), dgw.p, gw.size() * 4, hipMemcpyDeviceToHost, st));
  if (dx_nchw) {
    ConvArgs a{};
    a.x = ddyn.p; a.w = dwd.p; a.y = dgx.p;
    DevBuf dw8, dq, dq8;
    if (want_f8 && Cout % 32 == 0) {
      YS_TRY(dw8.alloc((size_t)Cin * taps * copad));
      YS_TRY(dq8.alloc((size_t)M * copad));
      YS_TRY(dq.alloc(1024));
      YS_CHECK_HIP(hipMemsetAsync(dq.p, 0, 1024, st));
      float* qf = (float*)dq.p;
      F8Layer hl{0, (long)Cout * taps * Cin};
      F8Conv hc{0};
      DevBuf dl, dc;
      YS_TRY(dl.alloc(sizeof hl)); YS_TRY(dc.alloc(sizeof hc));
      YS_CHECK_HIP(hipMemcpyAsync(dl.p, &hl, sizeof hl, hipMemcpyHostToDevice, st));
      YS_CHECK_HIP(hipMemcpyAsync(dc.p, &hc, sizeof hc, hipMemcpyHostToDevice, st));
      YS_TRY(ys_f8_weight_amax_launch(st, (const float*)dwm.p, (const F8Layer*)dl.p, 1, qf));
      YS_TRY(ys_f8_view_amax_launch(st, ddyn.p, M, copad, copad, 0, (unsigned*)(qf + 96)));
      YS_TRY(ys_f8_scales_launch(st, (const F8Conv*)dc.p, 1, qf, (unsigned*)(qf + 16), (unsigned*)(qf + 96), qf + 200));
      YS_TRY(ys_f8_quant_weights_launch(st, dwd.p, (long)Cin * taps * copad, qf, dw8.p));
      YS_CHECK_HIP(hipStreamSynchronize(st));
      a.f8 = 2; a.w8 = dw8.p; a.qscale = qf + 202; a.deq = qf + 203; a.q8 = dq8.p;
    }
    a.B = B; a.Hin = Ho; a.Win = Wo; a.Cin = copad; a.Hout = H; a.Wout = W; a.Cout = Cin; a.KH = a.KW = k;
    a.SA = 1; a.DIVS = stride == 2 ? 1 : 0; a.DIVM = stride - 1; a.PAD = k - 1 - pad;
    a.in_ldc = copad; a.in_coff = 0; a.in_bstride = (long)Ho * Wo;
    a.out_ldc = cpad; a.out_coff = 0; a.out_bstride = (long)H * W; a.vec_ok = 1; a.M = B * H * W;
    YS_TRY(ys_conv_launch(st, dtype, a));
    YS_TRY(ys_unpack_nchw_launch(st, dtype, dgx.p, cpad, 0, B, Cin, (long)H * W, (float*)dgxo.p));
    YS_CHECK_HIP(hipMemcpyAsync(dx_nchw, dgxo.p, (size_t)B * Cin * H * W * 4, hipMemcpyDeviceToHost, st));
  }
  YS_CHECK_HIP(hipStreamSynchronize(st));
  for (int co = 0; co < Cout; co++)
    for (int ci = 0; ci < Cin; ci++)
      for (int t = 0; t < taps; t++) dw_oihw[((size_t)co * Cin + ci) * taps + t] = gw[((size_t)co * taps + t) * Cin + ci];
  YS_CHECK_HIP(hipGetLastError());
  return YS_OK;
}
