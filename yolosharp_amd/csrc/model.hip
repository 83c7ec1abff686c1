// model.hip -- graph builder and step orchestration behind the ys_model_* / ys_loss_* / ys_optim_* ABI.
//
// Builds the reference's Yolov8 detect graph (Models/Yolo.cs:41-89: widths/depths table :43-55, layer list
// :56-87, skip router :92-134) out of Conv units (Modules/Convs.cs:36-62), C2f/Bottleneck
// (Modules/Block.cs:371-399, 572-608), SPPF (Block.cs:236-285), Upsample+Concat (Yolo.cs:70-75) and the
// Detect head (Modules/Head.cs:35-53,71-106,204-223), as a flat list of device ops over NHWC buffers:
//   * every Concat is a shared buffer: producers write their channel slice, chunk(2,1) is a view
//   * Bottleneck's shortcut add is fused into the BN/SiLU apply pass of its cv2
//   * Detect towers write straight into [B, A, C] (= the permuted layout the loss consumes)
// Training forward keeps the raw conv output y per Conv unit (BN backward needs it); backward walks the op
// list in reverse, gradient buffers mirror activation buffers, "first writer writes, later writers accumulate".
#include "ys_internal.h"
#include "ys_kernels.h"
#include <cmath>
#include <cstring>
#include <algorithm>

namespace {

struct View { int buf = -1; int coff = 0; int C = 0; };

struct Buf {
  int H = 0, W = 0, ldc = 0;
  long rows_per_b = 0;   // H*W (or A for head outputs)
  void* act = nullptr;
  void* grad = nullptr;
  bool need_grad = true;
  std::vector<char> gw;  // per-channel "gradient already written in this backward pass"
};

struct ConvL {
  std::string name;      // state_dict prefix
  int cin = 0, cout = 0, k = 1, s = 1;
  int cin_pad = 0;       // channels of the input view (first layer: 3 -> EPL)
  int cout_ld = 0;       // channels incl. padding in the dgrad weight matrix / dy rows
  int cout_real = 0;     // output channels of the reference module.  cout > cout_real only for the Pose towers (51 -> next 16-byte
                         // multiple): the extra rows of every parameter stay zero, so the extra channels are exactly 0 in both BN modes,
                         // receive zero gradients, and the state_dict surface lists the cout_real prefix
  bool bn = true, act = true;
  View in, out, res;
  bool has_res = false;
  int Hin = 0, Win = 0, Hout = 0, Wout = 0;
  long out_rowoff = 0;   // head outputs: first row of this level inside [B][A]
  long w_off = -1, g_off = -1, b_off = -1;   // flat parameter offsets (floats): weight, bn.weight|bias, bn.bias
  long rm_off = -1, rv_off = -1, nbt_off = -1;  // running stats in `state`
  long wf_off = 0, wd_off = 0;               // element offsets into wf_all / wd_all
  long y_off = 0;                             // element offset into y_all (bn layers)
  long acc_off = -1;                          // BatchNorm unit: offset (64-bit words) of its statistics accumulators in ys_model::stat_acc_all
  long ch_off = 0;                            // offset into per-channel scratch (scale.. c2), floats
  int seg = 0;
  bool first = false;
  bool dw = false;       // depthwise 3x3 (groups = channels): weights [9][C] fp32, no MFMA path
  bool f8_fwd = false, f8_bwd = false;   // fp8 mode: forward / dgrad of this layer may run the fp8 kernel (f8.hip recipe)
  int idx = -1, prep_idx = -1;           // own index in ys_model::convs; first PrepDesc (weight-amax slot)
  long wgp_off = -1; int wgp_splits = 0; // own region of the weight-gradient partial workspace (floats) and the splits it holds; -1 = shared scratch + immediate reduce
  int red_slot = -1;                     // index into ys_model::red_host (deferred split reduction)
  bool ct = false;       // ConvTranspose2d(k=2,s=2,bias) = four 1x1 phase GEMMs (Proto.upsample, Block.cs:69); weights [4][Cout][Cin]
  // fused BN-backward reduction (BnRedSeg, ys_kernels.h).  As a consumer: the producers whose dz this layer's dgrad completes
  // (it is their first reader in forward order = last gradient writer in backward order).  As a producer: where its sums come from.
  struct RedFeed { int prod; int c0, c1, yc0; long part_off; int rows_cap, rows; };
  struct RedSrc { int cons, feed; };
  std::vector<RedFeed> feeds;
  std::vector<RedSrc> red_src;           // sorted by producer channel
  bool red_ok = false; int red_seen = 0; // every source is a supported dgrad launch / sources attached in the current backward pass
  // Several reference modules executed as ONE convolution (shared-input fusion, add_detect): member k owns output rows
  // [row0, row0 + rows) of this layer's weight / BN vectors and appears in the state_dict under its own module name.  Empty = one module.
  struct Member { std::string name; int row0, rows; };
  std::vector<Member> members;
  // Level-parallel execution of the head (round 4): `stage` orders the head's ops stage-major (all pyramid levels of one tower layer next
  // to each other); ops of one `group` (>= 0) are consecutive in ys_model::ops, mutually independent, and run as grouped launches
  // (run_conv_fwd_group / run_conv_bwd_group).  Grouped units own their statistics rows and dy buffer (no shared scratch between problems).
  int stage = -1, group = -1;
  long gstat_off = -1;                   // floats into ys_model::stat_group
  void* dy_own = nullptr;
};

enum OpType { OP_CONV = 0, OP_MAXPOOL = 1, OP_UPSAMPLE = 2, OP_ATTN = 3, OP_VCOPY = 4, OP_COPY = 5 };
struct Op { int type; int conv = -1; View in, out; int H = 0, W = 0; long aux_off = 0; int seg = 0; int heads = 0, kd = 0, hd = 0; };

struct TensorRec {
  std::string name;
  int ndim = 1; int64_t shape[4] = {1, 1, 1, 1};
  bool is_param = true;
  int kind = 0;     // 0 conv weight (OIHW at the edge), 1 vector in flat params, 2 vector in state, 3 dfl weight
  int conv = -1;
  long off = 0, count = 0;
};

struct PrepDesc { long w_off, wf_off, wd_off, nf_start, nd_start; int cout, taps, cin_real, cin_pad, cout_pad, has_wd, phase;
                  long tile_start; int tiles_ci, tiles_co, layer, pad_; };   // round 5 (weight_prep_fast_kernel): 64 x 64 transpose tiles of the dgrad shadow, prefix over the table; layer = index in the full table (weight-amax slot)

}  // namespace

struct ys_model {
  ys_ctx* ctx = nullptr;
  ys_model_desc d{};
  int dtype = 0, epl = 4; size_t es = 4;
  int maxB = 0, B = 0;
  int A = 0, nl = 3;
  int lvl_off[4] = {0}, lvl_w[4] = {0}, lvl_h[4] = {0}, lvl_stride[4] = {8, 16, 32, 64};
  bool training = true;
  std::vector<Buf> bufs;
  std::vector<ConvL> convs;
  std::vector<Op> ops;
  std::vector<TensorRec> tensors;
  std::vector<int> reg;          // conv indices in the reference's module REGISTRATION order (state_dict order)
  std::string head_prefix;       // "model.22" (v8) / "model.23" (v11)
  float* attn_ws = nullptr; long n_attn = 0;   // softmax probabilities + dS of the C2PSA attention ops
  // segmentation (Head.cs:238-324): mask coefficients [B][A][ld_mc], prototypes [B][mh*mw][ld_pr]
  bool segment = false; int nm = 0, mc_buf = -1, pr_buf = -1, ld_mc = 0, ld_pr = 0, mh = 0, mw = 0;
  // Obb (Head.cs:376-482) / Pose (Head.cs:484-606) reuse the cv4 output buffer: nm = ne (1) / nk (kpt_num * kpt_dim) channels
  float* kp_dev = nullptr;       // Pose: staged keypoint labels [max_labels][K][D] (grows with the label workspace)
  int xkind = 0, kdim = 3;       // 0 none, 1 mask coefficients, 2 angle logit, 3 keypoints (argument of ys_detect_decode_launch)
  float* masks_dev = nullptr; int *seg_cnt = nullptr, *seg_off = nullptr, *seg_list = nullptr; float *seg_ent = nullptr, *seg_part = nullptr;
  int n_items = 3; bool have_seg_loss = false;
  int dfl_after_conv = -1;   // the DFL weight registers right after Detect's cv2/cv3 (Head.cs:52-56), before Segment's proto/cv4
  int in_buf = -1, pd_buf = -1, ps_buf = -1;
  bool is_block = false; int blk_out = -1, blk_c1 = 3, blk_c2 = 0;   // standalone block handle (ys_block_create)
  bool is_head = false; int head_in[3] = {-1, -1, -1}, head_ch[3] = {0, 0, 0};   // standalone head handle (ys_head_create): P3 / P4 / P5 input buffers
  int ld_pd = 0, ld_ps = 0;
  // flat fp32 parameter state
  long n_params = 0, n_params_real = 0;          // flat length incl. the zero rows of padded towers / the reference's parameter count
  float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr;
  float* state = nullptr; long n_state = 0;     // running_mean / running_var / num_batches_tracked
  float dfl_w[64];
  struct Range { long off, count; };
  static constexpr int NSEG = 4;                 // backward segments: head, neck, late backbone, stem (the last, exposed all-reduce is the smallest)
  Range seg_group[NSEG][3];                      // [segment][adamw group]
  long step = 0;
  // fp8 mode (ys_dtype YS_FP8: bf16 storage + fp8 MFMA convolutions, f8.hip)
  bool f8 = false, f8_sx_valid = false, f8_sg_valid = false, f8_bwd_done = false;
  unsigned char *wf8_all = nullptr, *wd8_all = nullptr;
  int q8_fwd_ready = -1;                  // forward: conv index whose input image already sits in q8 (written by its producer's BN pass)
  unsigned char* q8 = nullptr;            // scratch: fp8 image of one convolution input (blocked-GEMM fp8 kernel, quantised by ys_conv_launch)
  float *amax_w = nullptr, *f8_scales = nullptr; unsigned *amax_act = nullptr, *amax_dy = nullptr;
  F8Layer* f8_layers = nullptr; F8Conv* f8_convs = nullptr; int n_f8_convs = 0; long n_wf_pending = 0, n_wd_pending = 0;
  int group_mode = 0;                            // 0 = disjoint groups, 1 = the reference's overlapping groups as written
  unsigned char* bn_mask = nullptr;              // [n_params] 1 = BatchNorm weight / bias (listed twice in the reference's groups)
  // T weights
  void *wf_all = nullptr, *wd_all = nullptr; long n_wf = 0, n_wd = 0;
  PrepDesc* prep_dev = nullptr; int n_prep = 0; long prep_nf = 0, prep_nd = 0;
  bool prep_fast = false;
  PrepDesc* prep_tile_dev = nullptr; int n_prep_tile = 0; long prep_tiles = 0;          // layers whose dgrad shadow is a plain transpose (tile_start prefix)
  PrepDesc* prep_phase_dev = nullptr; int n_prep_phase = 0; long prep_nd_phase = 0;     // stride-2 layers with phase-major dgrad shadows (nd_start = compact prefix)
  bool weights_dirty = true, eval_coeffs_dirty = true;
  // activations
  void* y_all = nullptr; long n_y = 0;
  void* dy_scratch = nullptr; long n_dy = 0;
  // weight gradients run on a second stream, concurrently with the BN-backward / dgrad chain of the following layers
  // (both mostly latency-bound); dy lives in a ring of DY_RING buffers guarded by events
  static constexpr int DY_RING = 4;
  // round 6: weight gradients are handed to the second stream in BATCHES.  Every BatchNorm unit keeps its own dy buffer (ConvL::dy_own: no ring slot to wait for),
  // a unit's weight-gradient launch is queued instead of issued, and ONE event record / wait pair hands a whole batch over (flush_wgrads).  A rocprofv3 trace of
  // config 2 showed what the per-layer hand-off cost: every hipEventRecord between two kernels of the main stream is a ~6.6 us bubble (29 + 10 + 5 of them per step
  // between bn_bwd_apply and the dgrad that follows) and every ring-slot wait another ~6 us (21 per step): 0.6 ms of an 8.7 ms step with no kernel running on
  // the main stream, all of it in the backward pass (the forward has none).
  struct PendWg { int conv; const void* dy; int ldc, coff; long bstride; };
  std::vector<PendWg> pend_wg; double pend_mb = 0.0; int ev_hand = 0;
  bool hold_stem = false;                      // one-call backward: model.0's weight gradient stays queued until the segment end (backward_range: stem_split)
  // head lanes (round 3): the towers of the three pyramid levels are independent chains (own buffers, own rows of the prediction buffers);
  // the P4 / P5 chains are short, latency-bound launches (100-400 workgroups) that run beside the P3 chain on two side streams
  // asynchronous segment ends (data-parallel step): the weight-gradient stream is NOT joined into the main stream when a backward segment
  // ends; the segment's completion is two events (main stream, weight-gradient stream) a communication stream waits on (ys_model_segment_fence)
  hipEvent_t ev_seg_m[NSEG] = {nullptr, nullptr, nullptr, nullptr}, ev_seg_w[NSEG] = {nullptr, nullptr, nullptr, nullptr};
  bool seg_on_st2[NSEG] = {false, false, false, false};
  bool overlap = false, overlap_built = false; hipStream_t st2 = nullptr;   // overlap_built: second stream / dy ring exist; overlap: in use (ys_model_set_overlap)
  hipEvent_t ev_dy[DY_RING + 1] = {nullptr}, ev_join = nullptr;     // hand-over events (rotated; a wait binds to the record that preceded it)
  bool st2_dirty = false;
  float* chan = nullptr; long n_chan = 0;       // per conv: scale, shift, mean, rstd, c1, c2 (6*cout)
  float* stat_partial = nullptr; long n_stat = 0;
  unsigned long long* stat_acc_all = nullptr; long n_stat_acc = 0;   // round 5: [unit][YS_STAT_SHARDS][cout][2] fixed-point statistics sums, cleared by ONE memset per training forward
  bool bn_atomic = false;                                            // BatchNorm units take their statistics through them and finalize inside the apply pass
  float* stat_group = nullptr;                  // statistics rows of the grouped head stages (one region per unit, ConvL::gstat_off)
  float* wg_partial = nullptr; long n_wgp = 0;   // [shared scratch (ConvTranspose phases) | one region per convolution]
  // deferred split reduction of the weight gradients: one batched launch per backward_range call instead of one per layer
  std::vector<WgRedDesc> red_host, red_uploaded; WgRedDesc* red_dev = nullptr; int red_first[NSEG + 1] = {0, 0, 0, 0, 0};
  bool defer_wgred = true;
  // fused BN-backward reduction: planned per batch size (plan_bnred), partial rows of every (producer, consumer) pair
  bool bnred_on = true; int bnred_B = -1; float* bnred_part = nullptr; long n_bnred = 0;
  unsigned char* argmax = nullptr; long n_argmax = 0;
  float* img_dev = nullptr;                      // staging for host images
  bool stem_on = true;                           // YS_STEM_DIRECT=0 at creation: model.0 reads the packed bf16 copy like every other layer
  const float* in_f32 = nullptr;                 // the fp32 NCHW image of the current step when model.0 reads it directly (conv_stem.hip); null = the packed input buffer holds it
  float* pred = nullptr;                         // [B][4+nc][A] fp32 (eval)
  float* out_stage = nullptr; long n_out_stage = 0;
  // loss
  int gcap = 64; int max_labels = 0;
  float *lab_bidx = nullptr, *lab_cls = nullptr, *lab_box = nullptr;
  int* gt_count = nullptr; float* gt_box = nullptr; int* gt_cls = nullptr; float* pbox = nullptr;
  float *ov = nullptr, *align = nullptr; unsigned char* mpos = nullptr; unsigned *pos_align = nullptr, *pos_ov = nullptr;
  int* fg_gt = nullptr; float* tnorm = nullptr; float* loss_partial = nullptr; float* scalars = nullptr;
  bool have_fwd = false, have_loss = false;
  bool fwd_training = false;   // the last forward kept what backward needs (training-mode BN statistics, pre-BN outputs)
  std::vector<void*> allocs;
};

namespace {

int dev_alloc(ys_model* m, void** p, size_t bytes, bool zero = true) {
  if (bytes == 0) bytes = 16;
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) { ys_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); return YS_ERR_OOM; }
  m->allocs.push_back(*p);
  if (zero) { e = hipMemsetAsync(*p, 0, bytes, m->ctx->stream); if (e != hipSuccess) { ys_set_error("hipMemset failed"); return YS_ERR_HIP; } }
  return YS_OK;
}

int new_buf(ys_model* m, int H, int W, int C) {
  Buf b; b.H = H; b.W = W; b.ldc = C; b.rows_per_b = (long)H * W; b.gw.assign(C, 0);
  m->bufs.push_back(b);
  return (int)m->bufs.size() - 1;
}

void add_tensor(ys_model* m, const std::string& name, int kind, int conv, long off, std::vector<int64_t> shape, bool is_param) {
  TensorRec t; t.name = name; t.kind = kind; t.conv = conv; t.off = off; t.is_param = is_param;
  t.ndim = (int)shape.size(); t.count = 1;
  for (int i = 0; i < t.ndim; i++) { t.shape[i] = shape[i]; t.count *= shape[i]; }
  m->tensors.push_back(t);
}

// one Conv unit (conv+BN+act) or plain biased Conv2d; returns conv index and appends the op
int add_conv(ys_model* m, const std::string& name, View in, View out, int cin, int cout, int k, int s, bool bn, bool act,
             int Hin, int Win, int seg, const View* res = nullptr, bool dw = false) {
  ConvL c; c.name = name; c.in = in; c.out = out; c.cin = cin; c.cout = cout; c.k = k; c.s = s; c.bn = bn; c.act = act; c.dw = dw;
  c.cin_pad = in.C; c.Hin = Hin; c.Win = Win;
  const int p = k / 2;
  c.Hout = (Hin + 2 * p - k) / s + 1; c.Wout = (Win + 2 * p - k) / s + 1;
  c.cout_ld = (cout + m->epl - 1) / m->epl * m->epl;
  c.cout_real = cout;
  c.seg = seg;
  if (res) { c.res = *res; c.has_res = true; }
  c.idx = (int)m->convs.size();
  m->convs.push_back(c);
  Op op; op.type = OP_CONV; op.conv = (int)m->convs.size() - 1; op.in = in; op.out = out; op.H = Hin; op.W = Win; op.seg = seg;
  m->ops.push_back(op);
  return op.conv;
}

// C2f (Block.cs:371-399): cv1 1x1 -> chunk -> n Bottlenecks (3x3,3x3, e=1.0) -> cat -> cv2 1x1
void add_c2f(ys_model* m, const std::string& name, View xin, View xout, int c1, int c2, int n, bool shortcut, int H, int W, int seg) {
  const int c = (int)(c2 * 0.5f);
  const int cat = new_buf(m, H, W, (2 + n) * c);
  const int i1 = add_conv(m, name + ".cv1", xin, View{cat, 0, 2 * c}, c1, 2 * c, 1, 1, true, true, H, W, seg);
  std::vector<int> mids;
  for (int i = 0; i < n; i++) {
    const View bin{cat, (1 + i) * c, c};
    const View bout{cat, (2 + i) * c, c};
    const int tmp = new_buf(m, H, W, c);
    const std::string bp = name + ".m." + std::to_string(i);
    mids.push_back(add_conv(m, bp + ".cv1", bin, View{tmp, 0, c}, c, c, 3, 1, true, true, H, W, seg));
    mids.push_back(add_conv(m, bp + ".cv2", View{tmp, 0, c}, bout, c, c, 3, 1, true, true, H, W, seg, shortcut ? &bin : nullptr));
  }
  const int i2 = add_conv(m, name + ".cv2", View{cat, 0, (2 + n) * c}, xout, (2 + n) * c, c2, 1, 1, true, true, H, W, seg);
  m->reg.push_back(i1); m->reg.push_back(i2);            // registration order: cv1, cv2, m.* (Block.cs:373-376)
  for (int i : mids) m->reg.push_back(i);
}

// registration-list entry of member `mem` of a fused convolution (plain convolutions: the index itself)
inline int reg_entry(int conv, int mem) { return conv | (mem << 20); }
int add_conv_reg(ys_model* m, const std::string& name, View in, View out, int cin, int cout, int k, int s, bool bn, bool act,
                 int Hin, int Win, int seg, const View* res = nullptr, bool dw = false) {
  const int i = add_conv(m, name, in, out, cin, cout, k, s, bn, act, Hin, Win, seg, res, dw);
  m->reg.push_back(i);
  return i;
}

// Bottleneck (Block.cs:572-608) 3x3,3x3 with hidden = int(c*e); returns conv indices {cv1, cv2}
std::pair<int, int> add_bottleneck(ys_model* m, const std::string& name, View in, View out, int c, float e, bool shortcut, int H, int W, int seg) {
  const int c_ = (int)(c * e);
  const int tmp = new_buf(m, H, W, c_);
  const int a = add_conv(m, name + ".cv1", in, View{tmp, 0, c_}, c, c_, 3, 1, true, true, H, W, seg);
  const int b = add_conv(m, name + ".cv2", View{tmp, 0, c_}, out, c_, c, 3, 1, true, true, H, W, seg, shortcut ? &in : nullptr);
  return {a, b};
}

// C3k (Block.cs:611-620 over C3 :404-442): cv3(cat(m(cv1(x)), cv2(x))), m = n Bottlenecks(c_, c_, 3x3, e=1.0); registration
// order cv1, cv2, cv3, m.* ; appends its conv indices (registration order) to `regs`
void add_c3k(ys_model* m, const std::string& name, View in, View out, int c1, int c2, int n, bool shortcut, int H, int W, int seg,
             std::vector<int>& regs) {
  const int c_ = (int)(c2 * 0.5f);
  const int cat = new_buf(m, H, W, 2 * c_);
  int cur = new_buf(m, H, W, c_);
  const int i1 = add_conv(m, name + ".cv1", in, View{cur, 0, c_}, c1, c_, 1, 1, true, true, H, W, seg);
  std::vector<int> mids;
  for (int i = 0; i < n; i++) {
    const View bin{cur, 0, c_};
    View bout;
    if (i == n - 1) bout = View{cat, 0, c_};
    else { cur = new_buf(m, H, W, c_); bout = View{cur, 0, c_}; }
    auto pr = add_bottleneck(m, name + ".m." + std::to_string(i), bin, bout, c_, 1.0f, shortcut, H, W, seg);
    mids.push_back(pr.first); mids.push_back(pr.second);
  }
  const int i2 = add_conv(m, name + ".cv2", in, View{cat, c_, c_}, c1, c_, 1, 1, true, true, H, W, seg);
  const int i3 = add_conv(m, name + ".cv3", View{cat, 0, 2 * c_}, out, 2 * c_, c2, 1, 1, true, true, H, W, seg);
  regs.push_back(i1); regs.push_back(i2); regs.push_back(i3);
  for (int i : mids) regs.push_back(i);
}

// C3k2 (Block.cs:623-662): like C2f but m[i] = C3k(c,c,2) or Bottleneck(c,c,e=0.5); shortcut defaults to true
void add_c3k2(ys_model* m, const std::string& name, View xin, View xout, int c1, int c2, int n, bool c3k, float e, int H, int W, int seg) {
  const int c = (int)(c2 * e);
  const int cat = new_buf(m, H, W, (2 + n) * c);
  const int i1 = add_conv(m, name + ".cv1", xin, View{cat, 0, 2 * c}, c1, 2 * c, 1, 1, true, true, H, W, seg);
  std::vector<int> mids;
  for (int i = 0; i < n; i++) {
    const View bin{cat, (1 + i) * c, c};
    const View bout{cat, (2 + i) * c, c};
    const std::string bp = name + ".m." + std::to_string(i);
    if (c3k) add_c3k(m, bp, bin, bout, c, c, 2, true, H, W, seg, mids);
    else { auto pr = add_bottleneck(m, bp, bin, bout, c, 0.5f, true, H, W, seg); mids.push_back(pr.first); mids.push_back(pr.second); }
  }
  const int i2 = add_conv(m, name + ".cv2", View{cat, 0, (2 + n) * c}, xout, (2 + n) * c, c2, 1, 1, true, true, H, W, seg);
  m->reg.push_back(i1); m->reg.push_back(i2);
  for (int i : mids) m->reg.push_back(i);
}

// C2PSA (Block.cs:664-810).  Reference quirks kept: qkv / proj / pe and both ffn convs all use the default SiLU.
void add_c2psa(ys_model* m, const std::string& name, View xin, View xout, int c1, int n, int H, int W, int seg) {
  const int c = (int)(c1 * 0.5f);
  const int cat = new_buf(m, H, W, 2 * c);       // cv1 output: a | b
  const int cat2 = new_buf(m, H, W, 2 * c);      // cv2 input:  a | b_final
  const int i1 = add_conv(m, name + ".cv1", xin, View{cat, 0, 2 * c}, c1, 2 * c, 1, 1, true, true, H, W, seg);
  { Op op; op.type = OP_COPY; op.in = View{cat, 0, c}; op.out = View{cat2, 0, c}; op.H = H; op.W = W; op.seg = seg; m->ops.push_back(op); }
  std::vector<int> mids;
  View bcur{cat, c, c};
  const int heads = c / 64, hd = c / heads, kd = (int)(hd * 0.5f);
  const int hq = c + 2 * kd * heads;
  for (int i = 0; i < n; i++) {
    const std::string bp = name + ".m." + std::to_string(i);
    const int qb = new_buf(m, H, W, hq), ao = new_buf(m, H, W, c), vb = new_buf(m, H, W, c), sb = new_buf(m, H, W, c);
    const int b1 = new_buf(m, H, W, c), f1 = new_buf(m, H, W, 2 * c);
    const int iq = add_conv(m, bp + ".attn.qkv", bcur, View{qb, 0, hq}, c, hq, 1, 1, true, true, H, W, seg);
    { Op op; op.type = OP_ATTN; op.in = View{qb, 0, hq}; op.out = View{ao, 0, c}; op.H = H; op.W = W; op.seg = seg; op.heads = heads; op.kd = kd; op.hd = hd; m->ops.push_back(op); }
    { Op op; op.type = OP_VCOPY; op.in = View{qb, 0, hq}; op.out = View{vb, 0, c}; op.H = H; op.W = W; op.seg = seg; op.heads = heads; op.kd = kd; op.hd = hd; m->ops.push_back(op); }
    const View aov{ao, 0, c};
    const int ipe = add_conv(m, bp + ".attn.pe", View{vb, 0, c}, View{sb, 0, c}, c, c, 3, 1, true, true, H, W, seg, &aov, true);
    const int ipr = add_conv(m, bp + ".attn.proj", View{sb, 0, c}, View{b1, 0, c}, c, c, 1, 1, true, true, H, W, seg, &bcur);
    const int if0 = add_conv(m, bp + ".ffn.0", View{b1, 0, c}, View{f1, 0, 2 * c}, c, 2 * c, 1, 1, true, true, H, W, seg);
    const View b1v{b1, 0, c};
    View bnext;
    if (i == n - 1) bnext = View{cat2, c, c};
    else { const int nb = new_buf(m, H, W, c); bnext = View{nb, 0, c}; }
    const int if1 = add_conv(m, bp + ".ffn.1", View{f1, 0, 2 * c}, bnext, 2 * c, c, 1, 1, true, true, H, W, seg, &b1v);
    mids.push_back(iq); mids.push_back(ipr); mids.push_back(ipe); mids.push_back(if0); mids.push_back(if1);   // qkv, proj, pe (Block.cs:744-746)
    bcur = bnext;
  }
  const int i2 = add_conv(m, name + ".cv2", View{cat2, 0, 2 * c}, xout, 2 * c, c1, 1, 1, true, true, H, W, seg);
  m->reg.push_back(i1); m->reg.push_back(i2);
  for (int i : mids) m->reg.push_back(i);
}

void add_sppf(ys_model* m, const std::string& name, View xin, View xout, int c1, int H, int W, int seg) {
  // SPPF (Block.cs:236-285): cv1 has NO activation (:257); three chained 5x5 pools
  const int c_ = c1 / 2;
  const int catS = new_buf(m, H, W, 4 * c_);
  add_conv_reg(m, name + ".cv1", xin, View{catS, 0, c_}, c1, c_, 1, 1, true, false, H, W, seg);
  for (int i = 0; i < 3; i++) {
    Op op; op.type = OP_MAXPOOL; op.in = View{catS, i * c_, c_}; op.out = View{catS, (i + 1) * c_, c_}; op.H = H; op.W = W; op.seg = seg;
    m->ops.push_back(op);
  }
  add_conv_reg(m, name + ".cv2", View{catS, 0, 4 * c_}, xout, 4 * c_, c1, 1, 1, true, true, H, W, seg);
}

// Proto (Block.cs:51-84): cv1 3x3 -> ConvTranspose2d(2,2,bias) -> cv2 3x3 -> cv3 1x1; field / registration order cv1, cv2, cv3,
// upsample (:53-56).  Returns the output buffer [2H][2W][pad(nm)].
int add_proto(ys_model* m, const std::string& name, View in, int c1, int npr, int nm, int H, int W, int seg) {
  const int ld = (nm + m->epl - 1) / m->epl * m->epl;
  const int pa = new_buf(m, H, W, npr), pu = new_buf(m, 2 * H, 2 * W, npr), pb = new_buf(m, 2 * H, 2 * W, npr);
  const int out = new_buf(m, 2 * H, 2 * W, ld);
  const int p1 = add_conv(m, name + ".cv1", in, View{pa, 0, npr}, c1, npr, 3, 1, true, true, H, W, seg);
  const int pup = add_conv(m, name + ".upsample", View{pa, 0, npr}, View{pu, 0, npr}, npr, npr, 1, 1, false, false, H, W, seg);
  m->convs[pup].ct = true; m->convs[pup].Hout = 2 * H; m->convs[pup].Wout = 2 * W;
  const int p2 = add_conv(m, name + ".cv2", View{pu, 0, npr}, View{pb, 0, npr}, npr, npr, 3, 1, true, true, 2 * H, 2 * W, seg);
  const int p3 = add_conv(m, name + ".cv3", View{pb, 0, npr}, View{out, 0, nm}, npr, nm, 1, 1, true, true, 2 * H, 2 * W, seg);
  m->reg.push_back(p1); m->reg.push_back(p2); m->reg.push_back(p3); m->reg.push_back(pup);
  return out;
}

// Detect (Head.cs:35-53): c2 = max(16, ch0/4, 4*reg_max), c3 = max(ch0, min(nc,100)); strides fixed {8,16,32} (:43).
// legacy=false (v11, Head.cs:50): each 3x3 Conv of the cls tower becomes DWConv3x3(x->x) + Conv1x1(x->c3).
int add_detect(ys_model* m, const std::string& hp, const int* pv, const int* ch, const int* hh, const int* ww, bool legacy, int seg) {
  const ys_model_desc& d = m->d;
  const int c2 = std::max(16, std::max(ch[0] / 4, d.reg_max * 4));
  const int c3 = std::max(ch[0], std::min(d.nc, 100));
  if (c2 % m->epl || c3 % m->epl) { ys_set_error("model: head widths c2=%d c3=%d must be multiples of %d", c2, c3, m->epl); return YS_ERR_UNSUPPORTED; }
  m->head_prefix = hp;
  m->nl = 3; m->A = 0;
  for (int i = 0; i < 3; i++) { m->lvl_off[i] = m->A; m->lvl_w[i] = ww[i]; m->lvl_h[i] = hh[i]; m->A += hh[i] * ww[i]; }
  m->ld_pd = (4 * d.reg_max + m->epl - 1) / m->epl * m->epl;
  m->ld_ps = (d.nc + m->epl - 1) / m->epl * m->epl;
  m->pd_buf = new_buf(m, 1, m->A, m->ld_pd);
  m->ps_buf = new_buf(m, 1, m->A, m->ld_ps);
  // Shared-input fusion (round 4).  cv2[i][0] = Conv(x, c2, 3) and cv3[i][0] = Conv(x, c3, 3) -- and Segment's cv4[i][0] = Conv(x, c4, 3) --
  // read the SAME x[i] (Head.cs:47-48, 254-259, consumed at :81-82): they run as ONE convolution with Cout = c2 + c3 (+ c4) into one
  // buffer whose channel slices the second tower layers read.  BatchNorm and SiLU are per channel, so this is exact; the input is
  // read once in the forward and in the weight gradient, and the input gradient is ONE dgrad with K = 9 (c2 + c3 + c4) instead of an
  // overwrite followed by read-modify-write accumulations.  The state_dict keeps the reference's module names (ConvL::members).
  // v11 heads (legacy = false) start their class tower with a depthwise unit: nothing to fuse with there.  YS_HEAD_FUSE=0: one launch each.
  const int c4s = d.task == YS_SEGMENT ? std::max(ch[0] / 4, 32) : 0;
  const int cf = c2 + c3 + c4s;
  bool fuse = legacy && (YS_OPT_INT("HEAD_FUSE", 1) != 0) && (c4s % m->epl) == 0;
  if (m->f8 && (cf % 32 || c2 % 32 || c3 % 32)) fuse = false;   // fp8 mode: the dgrad of a fused layer must still qualify for the fp8 kernel (K units of 32)
  int fbuf[3] = {-1, -1, -1}, fconv[3] = {-1, -1, -1};
  const size_t op0 = m->ops.size();
  auto tag = [&](int idx, int tower, int depth) { m->convs[idx].stage = tower * 4 + depth; };   // towers: 0 cv2, 1 cv3, 2 proto, 3 cv4
  for (int t = 0; t < 2; t++) {   // cv2 towers for all levels, then cv3 towers (registration order cv2.*, cv3.*)
    for (int i = 0; i < 3; i++) {
      const int cm = t == 0 ? c2 : c3;
      const int co = t == 0 ? 4 * d.reg_max : d.nc;
      const int ob = t == 0 ? m->pd_buf : m->ps_buf;
      const std::string tp = hp + (t == 0 ? ".cv2." : ".cv3.") + std::to_string(i);
      const int t1 = new_buf(m, hh[i], ww[i], cm);
      if (fuse) {
        if (t == 0) {
          fbuf[i] = new_buf(m, hh[i], ww[i], cf);
          fconv[i] = add_conv(m, tp + ".0+cv3" + (c4s ? "+cv4" : ""), View{pv[i], 0, ch[i]}, View{fbuf[i], 0, cf}, ch[i], cf, 3, 1, true, true, hh[i], ww[i], seg);
          ConvL& fc = m->convs[fconv[i]];
          fc.members.push_back({tp + ".0", 0, c2});
          fc.members.push_back({hp + ".cv3." + std::to_string(i) + ".0", c2, c3});
          if (c4s) fc.members.push_back({hp + ".cv4." + std::to_string(i) + ".0", c2 + c3, c4s});
          tag(fconv[i], 0, 0);
        }
        m->reg.push_back(reg_entry(fconv[i], t));
        tag(add_conv_reg(m, tp + ".1", View{fbuf[i], t == 0 ? 0 : c2, cm}, View{t1, 0, cm}, cm, cm, 3, 1, true, true, hh[i], ww[i], seg), t, 1);
      } else if (t == 0 || legacy) {
        const int t0 = new_buf(m, hh[i], ww[i], cm);
        tag(add_conv_reg(m, tp + ".0", View{pv[i], 0, ch[i]}, View{t0, 0, cm}, ch[i], cm, 3, 1, true, true, hh[i], ww[i], seg), t, 0);
        tag(add_conv_reg(m, tp + ".1", View{t0, 0, cm}, View{t1, 0, cm}, cm, cm, 3, 1, true, true, hh[i], ww[i], seg), t, 1);
      } else {
        const int t0 = new_buf(m, hh[i], ww[i], cm);
        const int d0 = new_buf(m, hh[i], ww[i], ch[i]), d1 = new_buf(m, hh[i], ww[i], cm);
        add_conv_reg(m, tp + ".0.0", View{pv[i], 0, ch[i]}, View{d0, 0, ch[i]}, ch[i], ch[i], 3, 1, true, true, hh[i], ww[i], seg, nullptr, true);
        add_conv_reg(m, tp + ".0.1", View{d0, 0, ch[i]}, View{t0, 0, cm}, ch[i], cm, 1, 1, true, true, hh[i], ww[i], seg);
        add_conv_reg(m, tp + ".1.0", View{t0, 0, cm}, View{d1, 0, cm}, cm, cm, 3, 1, true, true, hh[i], ww[i], seg, nullptr, true);
        add_conv_reg(m, tp + ".1.1", View{d1, 0, cm}, View{t1, 0, cm}, cm, cm, 1, 1, true, true, hh[i], ww[i], seg);
      }
      const int cc = add_conv_reg(m, tp + ".2", View{t1, 0, cm}, View{ob, 0, co}, cm, co, 1, 1, false, false, hh[i], ww[i], seg);
      m->convs[cc].out_rowoff = m->lvl_off[i];
      tag(cc, t, 2);
    }
  }
  m->dfl_after_conv = m->reg.back();
  if (d.task == YS_SEGMENT) {
    // Segment (Head.cs:238-324): Proto(ch0, npr = ch0, nm = 32) on P3 (Yolo.cs:348,366) and the cv4 towers (c4 = max(ch0/4, nm))
    m->segment = true; m->nm = 32; m->n_items = 5;
    const int nm = m->nm, npr = ch[0];
    const int c4 = std::max(ch[0] / 4, nm);
    if (c4 % m->epl || npr % m->epl) { ys_set_error("model: segment widths c4=%d npr=%d must be multiples of %d", c4, npr, m->epl); return YS_ERR_UNSUPPORTED; }
    m->ld_mc = (nm + m->epl - 1) / m->epl * m->epl;
    m->ld_pr = m->ld_mc;
    m->mh = 2 * hh[0]; m->mw = 2 * ww[0];
    {
      const size_t pc0 = m->convs.size();
      m->pr_buf = add_proto(m, hp + ".proto", View{pv[0], 0, ch[0]}, ch[0], npr, nm, hh[0], ww[0], seg);
      for (size_t k = pc0; k < m->convs.size(); k++) m->convs[k].stage = 2 * 4 + (int)(k - pc0);   // a chain: one unit per stage
    }
    m->mc_buf = new_buf(m, 1, m->A, m->ld_mc);
    for (int i = 0; i < 3; i++) {
      const std::string tp = hp + ".cv4." + std::to_string(i);
      const int t1 = new_buf(m, hh[i], ww[i], c4);
      if (fuse) {                 // cv4[i][0] ran inside the level's fused first convolution (channels [c2 + c3, c2 + c3 + c4) of its output)
        m->reg.push_back(reg_entry(fconv[i], 2));
        tag(add_conv_reg(m, tp + ".1", View{fbuf[i], c2 + c3, c4}, View{t1, 0, c4}, c4, c4, 3, 1, true, true, hh[i], ww[i], seg), 4, 1);
      } else {
        const int t0 = new_buf(m, hh[i], ww[i], c4);
        tag(add_conv_reg(m, tp + ".0", View{pv[i], 0, ch[i]}, View{t0, 0, c4}, ch[i], c4, 3, 1, true, true, hh[i], ww[i], seg), 4, 0);
        tag(add_conv_reg(m, tp + ".1", View{t0, 0, c4}, View{t1, 0, c4}, c4, c4, 3, 1, true, true, hh[i], ww[i], seg), 4, 1);
      }
      const int cc = add_conv_reg(m, tp + ".2", View{t1, 0, c4}, View{m->mc_buf, 0, nm}, c4, nm, 1, 1, false, false, hh[i], ww[i], seg);
      m->convs[cc].out_rowoff = m->lvl_off[i];
      tag(cc, 4, 2);
    }
    m->xkind = 1;
  } else if (d.task == YS_OBB || d.task == YS_POSE) {
    // Obb: ne = 1 angle logit per anchor (Head.cs:384-390); Pose: nk = kpt_num * kpt_dim outputs (Head.cs:492-499).  Both add
    // only the cv4 towers (c4 = max(ch0/4, ne|nk)) after Detect's own modules
    const bool obb = d.task == YS_OBB;
    m->kdim = d.kpt_dim > 0 ? d.kpt_dim : 3;
    const int nx = obb ? 1 : (d.kpt_num > 0 ? d.kpt_num : 17) * m->kdim;
    if (!obb && m->kdim != 2 && m->kdim != 3) { ys_set_error("model: keypoint dim %d (2 or 3)", m->kdim); return YS_ERR_INVALID_ARG; }
    m->nm = nx; m->xkind = obb ? 2 : 3;
    m->n_items = obb ? 4 : 5;      // v8OBBLoss: box, cls, dfl, angle (Loss.cs:546); v8PoseLoss: box, pose, kobj, cls, dfl (Loss.cs:923)
    const int c4 = std::max(ch[0] / 4, nx);                      // Pose n/s/m: 51, not a multiple of the 16-byte unit
    const int c4p = (c4 + m->epl - 1) / m->epl * m->epl;         // tower buffers are padded; the pad channels stay zero
    m->ld_mc = (nx + m->epl - 1) / m->epl * m->epl;
    m->mc_buf = new_buf(m, 1, m->A, m->ld_mc);
    for (int i = 0; i < 3; i++) {
      const std::string tp = hp + ".cv4." + std::to_string(i);
      const int t0 = new_buf(m, hh[i], ww[i], c4p), t1 = new_buf(m, hh[i], ww[i], c4p);
      const int k0 = add_conv_reg(m, tp + ".0", View{pv[i], 0, ch[i]}, View{t0, 0, c4p}, ch[i], c4p, 3, 1, true, true, hh[i], ww[i], seg);
      const int k1 = add_conv_reg(m, tp + ".1", View{t0, 0, c4p}, View{t1, 0, c4p}, c4, c4p, 3, 1, true, true, hh[i], ww[i], seg);
      m->convs[k0].cout_real = m->convs[k1].cout_real = c4;
      const int cc = add_conv_reg(m, tp + ".2", View{t1, 0, c4p}, View{m->mc_buf, 0, nx}, c4, nx, 1, 1, false, false, hh[i], ww[i], seg);
      m->convs[cc].out_rowoff = m->lvl_off[i];
      tag(k0, 4, 0); tag(k1, 4, 1); tag(cc, 4, 2);
    }
  }
  // Level-parallel schedule (v8 heads; YS_GROUP=0 keeps the level-major order and single launches): the head's ops stage-major -- the same
  // tower layer of P3, P4, P5 next to each other -- and every stage of >= 2 units is one group.  Units of a stage read and write disjoint
  // buffers (per-level towers, per-level rows of the prediction buffers); a stage only depends on earlier stages of its own tower and on
  // the (fused) first layers, which sort first.
  const bool group_on = legacy && (YS_OPT_INT("GROUP", 1) != 0) && !m->f8;
  if (group_on) {
    bool all = true;
    for (size_t k = op0; k < m->ops.size(); k++) all = all && m->ops[k].type == OP_CONV && m->convs[m->ops[k].conv].stage >= 0;
    if (all) {
      std::stable_sort(m->ops.begin() + op0, m->ops.end(), [&](const Op& x, const Op& y) { return m->convs[x.conv].stage < m->convs[y.conv].stage; });
      int gid = 0;
      for (size_t k = op0; k < m->ops.size();) {
        size_t e = k + 1;
        while (e < m->ops.size() && m->convs[m->ops[e].conv].stage == m->convs[m->ops[k].conv].stage) e++;
        const ConvL& c0 = m->convs[m->ops[k].conv];
        if (e - k >= 2 && (int)(e - k) <= YS_GROUP_MAX && !c0.dw && !c0.ct) { for (size_t j = k; j < e; j++) m->convs[m->ops[j].conv].group = gid; gid++; }
        k = e;
      }
    }
  }
  return YS_OK;
}

int build_v8_detect(ys_model* m) {
  const ys_model_desc& d = m->d;
  static const float dm[5] = {0.34f, 0.34f, 0.67f, 1.0f, 1.0f};
  static const float wm[5] = {0.25f, 0.5f, 0.75f, 1.0f, 1.25f};
  static const int mc[5] = {1024, 1024, 576, 512, 640};
  const int base_w[5] = {64, 128, 256, 512, 1024};
  int w[5];
  for (int i = 0; i < 5; i++) w[i] = std::min((int)(base_w[i] * wm[d.size]), mc[d.size]);   // Yolo.cs:53
  const int dep[3] = {(int)(3 * dm[d.size]), (int)(6 * dm[d.size]), (int)(9 * dm[d.size])};  // Yolo.cs:54
  const int H = d.height, W = d.width;
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8, H16 = H / 16, W16 = W / 16, H32 = H / 32, W32 = W / 32;
  for (int i = 0; i < 5; i++)
    if (w[i] % m->epl || ((int)(w[i] * 0.5f)) % m->epl) { ys_set_error("model: width %d not a multiple of %d", w[i], m->epl); return YS_ERR_UNSUPPORTED; }

  m->in_buf = new_buf(m, H, W, m->epl);
  m->bufs[m->in_buf].need_grad = false;
  // concat buffers of the neck (Yolo.cs:70-84; concat order [x, skip], Yolo.cs:107)
  const int cat11 = new_buf(m, H16, W16, w[4] + w[3]);
  const int cat14 = new_buf(m, H8, W8, w[3] + w[2]);
  const int cat17 = new_buf(m, H16, W16, w[2] + w[3]);
  const int cat20 = new_buf(m, H32, W32, w[3] + w[4]);
  const int b0 = new_buf(m, H2, W2, w[0]), b1 = new_buf(m, H4, W4, w[1]), b2 = new_buf(m, H4, W4, w[1]);
  const int b3 = new_buf(m, H8, W8, w[2]), b5 = new_buf(m, H16, W16, w[3]), b7 = new_buf(m, H32, W32, w[4]);
  const int b8 = new_buf(m, H32, W32, w[4]);
  const int b15 = new_buf(m, H8, W8, w[2]), b18 = new_buf(m, H16, W16, w[3]), b21 = new_buf(m, H32, W32, w[4]);
  const View v4{cat14, w[3], w[2]}, v6{cat11, w[4], w[3]}, v9{cat20, w[3], w[4]}, v12{cat17, w[2], w[3]};
  // backward segments: head first, stem last.  The backbone is cut after model.4: the all-reduce of the LAST segment has no
  // backward left to hide behind, so it should be the smallest (model.0-4: 2.5 % of YOLOv8n's parameters, 4 % of YOLOv8s')
  const int SB = 2, SN = 1, SH = 0, SS = 3;

  int ci = add_conv_reg(m, "model.0", View{m->in_buf, 0, m->epl}, View{b0, 0, w[0]}, 3, w[0], 3, 2, true, true, H, W, SS);
  m->convs[ci].first = true;
  add_conv_reg(m, "model.1", View{b0, 0, w[0]}, View{b1, 0, w[1]}, w[0], w[1], 3, 2, true, true, H2, W2, SS);
  add_c2f(m, "model.2", View{b1, 0, w[1]}, View{b2, 0, w[1]}, w[1], w[1], dep[0], true, H4, W4, SS);
  add_conv_reg(m, "model.3", View{b2, 0, w[1]}, View{b3, 0, w[2]}, w[1], w[2], 3, 2, true, true, H4, W4, SS);
  add_c2f(m, "model.4", View{b3, 0, w[2]}, v4, w[2], w[2], dep[1], true, H8, W8, SS);
  add_conv_reg(m, "model.5", v4, View{b5, 0, w[3]}, w[2], w[3], 3, 2, true, true, H8, W8, SB);
  add_c2f(m, "model.6", View{b5, 0, w[3]}, v6, w[3], w[3], dep[1], true, H16, W16, SB);
  add_conv_reg(m, "model.7", v6, View{b7, 0, w[4]}, w[3], w[4], 3, 2, true, true, H16, W16, SB);
  add_c2f(m, "model.8", View{b7, 0, w[4]}, View{b8, 0, w[4]}, w[4], w[4], dep[0], true, H32, W32, SB);
  add_sppf(m, "model.9", View{b8, 0, w[4]}, v9, w[4], H32, W32, SB);
  { Op op; op.type = OP_UPSAMPLE; op.in = v9; op.out = View{cat11, 0, w[4]}; op.H = H32; op.W = W32; op.seg = SN; m->ops.push_back(op); }
  add_c2f(m, "model.12", View{cat11, 0, w[4] + w[3]}, v12, w[4] + w[3], w[3], dep[0], false, H16, W16, SN);
  { Op op; op.type = OP_UPSAMPLE; op.in = v12; op.out = View{cat14, 0, w[3]}; op.H = H16; op.W = W16; op.seg = SN; m->ops.push_back(op); }
  add_c2f(m, "model.15", View{cat14, 0, w[3] + w[2]}, View{b15, 0, w[2]}, w[3] + w[2], w[2], dep[0], false, H8, W8, SN);
  add_conv_reg(m, "model.16", View{b15, 0, w[2]}, View{cat17, 0, w[2]}, w[2], w[2], 3, 2, true, true, H8, W8, SN);
  add_c2f(m, "model.18", View{cat17, 0, w[2] + w[3]}, View{b18, 0, w[3]}, w[2] + w[3], w[3], dep[0], false, H16, W16, SN);
  add_conv_reg(m, "model.19", View{b18, 0, w[3]}, View{cat20, 0, w[3]}, w[3], w[3], 3, 2, true, true, H16, W16, SN);
  add_c2f(m, "model.21", View{cat20, 0, w[3] + w[4]}, View{b21, 0, w[4]}, w[3] + w[4], w[4], dep[0], false, H32, W32, SN);
  const int ch[3] = {w[2], w[3], w[4]};
  const int hh[3] = {H8, H16, H32}, ww[3] = {W8, W16, W32};
  const int pv[3] = {b15, b18, b21};
  return add_detect(m, "model.22", pv, ch, hh, ww, true, SH);
}

// Yolov11 detect (Yolo.cs:200-258): C3k2 backbone/neck (shortcut=true everywhere: C3k2's default), SPPF, C2PSA, Detect(legacy=false)
int build_v11_detect(ys_model* m) {
  const ys_model_desc& d = m->d;
  static const float dm[5] = {0.5f, 0.5f, 0.5f, 1.0f, 1.0f};
  static const float wm[5] = {0.25f, 0.5f, 1.0f, 1.0f, 1.5f};
  static const int mc[5] = {1024, 1024, 512, 512, 768};
  static const bool c3k_sz[5] = {false, false, true, true, true};
  const int base_w[5] = {64, 128, 256, 512, 1024};
  int w[5];
  for (int i = 0; i < 5; i++) w[i] = std::min((int)(base_w[i] * wm[d.size]), mc[d.size]);
  const int n = (int)(2 * dm[d.size]);
  const bool uc = c3k_sz[d.size];
  const int H = d.height, W = d.width;
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8, H16 = H / 16, W16 = W / 16, H32 = H / 32, W32 = W / 32;
  if (((int)(w[2] * 0.25f) / 2) % m->epl) {
    ys_set_error("model: YOLOv11 size %d needs hidden widths that are multiples of %d (use the f32 model for n)", d.size, m->epl);
    return YS_ERR_UNSUPPORTED;
  }
  if ((w[4] / 2) % 64) { ys_set_error("model: C2PSA needs c %% 64 == 0"); return YS_ERR_UNSUPPORTED; }
  m->in_buf = new_buf(m, H, W, m->epl);
  m->bufs[m->in_buf].need_grad = false;
  const int cat12 = new_buf(m, H16, W16, w[4] + w[3]);   // [up(L10) | L6]
  const int cat15 = new_buf(m, H8, W8, w[3] + w[3]);     // [up(L13) | L4]
  const int cat18 = new_buf(m, H16, W16, w[2] + w[3]);   // [L17 | L13]
  const int cat21 = new_buf(m, H32, W32, w[3] + w[4]);   // [L20 | L10]
  const int b0 = new_buf(m, H2, W2, w[0]), b1 = new_buf(m, H4, W4, w[1]), b2 = new_buf(m, H4, W4, w[2]);
  const int b3 = new_buf(m, H8, W8, w[2]), b5 = new_buf(m, H16, W16, w[3]), b7 = new_buf(m, H32, W32, w[4]);
  const int b8 = new_buf(m, H32, W32, w[4]), b9 = new_buf(m, H32, W32, w[4]);
  const int b16 = new_buf(m, H8, W8, w[2]), b19 = new_buf(m, H16, W16, w[3]), b22 = new_buf(m, H32, W32, w[4]);
  const View v4{cat15, w[3], w[3]}, v6{cat12, w[4], w[3]}, v10{cat21, w[3], w[4]}, v13{cat18, w[2], w[3]};
  const int SB = 2, SN = 1, SH = 0, SS = 3;   // stem segment: see build_v8_detect
  int ci = add_conv_reg(m, "model.0", View{m->in_buf, 0, m->epl}, View{b0, 0, w[0]}, 3, w[0], 3, 2, true, true, H, W, SS);
  m->convs[ci].first = true;
  add_conv_reg(m, "model.1", View{b0, 0, w[0]}, View{b1, 0, w[1]}, w[0], w[1], 3, 2, true, true, H2, W2, SS);
  add_c3k2(m, "model.2", View{b1, 0, w[1]}, View{b2, 0, w[2]}, w[1], w[2], n, uc, 0.25f, H4, W4, SS);
  add_conv_reg(m, "model.3", View{b2, 0, w[2]}, View{b3, 0, w[2]}, w[2], w[2], 3, 2, true, true, H4, W4, SS);
  add_c3k2(m, "model.4", View{b3, 0, w[2]}, v4, w[2], w[3], n, uc, 0.25f, H8, W8, SS);
  add_conv_reg(m, "model.5", v4, View{b5, 0, w[3]}, w[3], w[3], 3, 2, true, true, H8, W8, SB);
  add_c3k2(m, "model.6", View{b5, 0, w[3]}, v6, w[3], w[3], n, true, 0.5f, H16, W16, SB);
  add_conv_reg(m, "model.7", v6, View{b7, 0, w[4]}, w[3], w[4], 3, 2, true, true, H16, W16, SB);
  add_c3k2(m, "model.8", View{b7, 0, w[4]}, View{b8, 0, w[4]}, w[4], w[4], n, true, 0.5f, H32, W32, SB);
  add_sppf(m, "model.9", View{b8, 0, w[4]}, View{b9, 0, w[4]}, w[4], H32, W32, SB);
  add_c2psa(m, "model.10", View{b9, 0, w[4]}, v10, w[4], n, H32, W32, SB);
  { Op op; op.type = OP_UPSAMPLE; op.in = v10; op.out = View{cat12, 0, w[4]}; op.H = H32; op.W = W32; op.seg = SN; m->ops.push_back(op); }
  add_c3k2(m, "model.13", View{cat12, 0, w[4] + w[3]}, v13, w[4] + w[3], w[3], n, uc, 0.5f, H16, W16, SN);
  { Op op; op.type = OP_UPSAMPLE; op.in = v13; op.out = View{cat15, 0, w[3]}; op.H = H16; op.W = W16; op.seg = SN; m->ops.push_back(op); }
  add_c3k2(m, "model.16", View{cat15, 0, w[3] + w[3]}, View{b16, 0, w[2]}, w[3] + w[3], w[2], n, uc, 0.5f, H8, W8, SN);
  add_conv_reg(m, "model.17", View{b16, 0, w[2]}, View{cat18, 0, w[2]}, w[2], w[2], 3, 2, true, true, H8, W8, SN);
  add_c3k2(m, "model.19", View{cat18, 0, w[2] + w[3]}, View{b19, 0, w[3]}, w[2] + w[3], w[3], n, uc, 0.5f, H16, W16, SN);
  add_conv_reg(m, "model.20", View{b19, 0, w[3]}, View{cat21, 0, w[3]}, w[3], w[3], 3, 2, true, true, H16, W16, SN);
  add_c3k2(m, "model.22", View{cat21, 0, w[3] + w[4]}, View{b22, 0, w[4]}, w[3] + w[4], w[4], n, true, 0.5f, H32, W32, SN);
  const int ch[3] = {w[2], w[3], w[4]};
  const int hh[3] = {H8, H16, H32}, ww[3] = {W8, W16, W32};
  const int pv[3] = {b16, b19, b22};
  return add_detect(m, "model.23", pv, ch, hh, ww, false, SH);
}

// Standalone block graph (ys_block_create): input buffer -> one module -> output buffer.  Modules are built with the
// prefix "block" and the tensor names are made module-relative afterwards (layout_block_names).
int build_block(ys_model* m, const ys_block_desc& bd) {
  const int epl = m->epl, H = bd.height, W = bd.width, c1 = bd.c1, c2 = bd.c2;
  auto bad = [&](int v, const char* what) {
    if (v > 0 && v % epl == 0) return false;
    ys_set_error("ys_block_create: %s = %d must be a positive multiple of %d for this dtype", what, v, epl);
    return true;
  };
  if (bad(c1, "c1") || bad(c2, "c2")) return YS_ERR_UNSUPPORTED;
  const float e = bd.e > 0.f ? bd.e : 0.5f;
  m->in_buf = new_buf(m, H, W, c1);
  const View vin{m->in_buf, 0, c1};
  int Ho = H, Wo = W;
  if (bd.kind == YS_BLOCK_CONV) {
    if ((bd.k != 1 && bd.k != 3) || (bd.s != 1 && bd.s != 2)) { ys_set_error("ys_block_create: Conv k=%d s=%d unsupported (k in {1,3}, s in {1,2})", bd.k, bd.s); return YS_ERR_UNSUPPORTED; }
    const int p = bd.k / 2;
    Ho = (H + 2 * p - bd.k) / bd.s + 1; Wo = (W + 2 * p - bd.k) / bd.s + 1;
  } else if (bd.kind == YS_BLOCK_PROTO) {
    Ho = 2 * H; Wo = 2 * W;
  }
  m->blk_out = new_buf(m, Ho, Wo, c2);
  const View vout{m->blk_out, 0, c2};
  const std::string nm = "block";
  switch (bd.kind) {
    case YS_BLOCK_CONV:
      add_conv_reg(m, nm, vin, vout, c1, c2, bd.k, bd.s, true, bd.act != 0, H, W, 0);
      break;
    case YS_BLOCK_BOTTLENECK: {
      if (c1 != c2) { ys_set_error("ys_block_create: Bottleneck needs c1 == c2 (the graphs' only use)"); return YS_ERR_UNSUPPORTED; }
      if (bad((int)(c2 * e), "Bottleneck hidden width")) return YS_ERR_UNSUPPORTED;
      auto pr = add_bottleneck(m, nm, vin, vout, c2, e, bd.shortcut != 0, H, W, 0);
      m->reg.push_back(pr.first); m->reg.push_back(pr.second);
      break;
    }
    case YS_BLOCK_C2F:
      if (bd.n < 1 || bad((int)(c2 * 0.5f), "C2f hidden width")) { if (bd.n < 1) ys_set_error("ys_block_create: n = %d", bd.n); return YS_ERR_UNSUPPORTED; }
      add_c2f(m, nm, vin, vout, c1, c2, bd.n, bd.shortcut != 0, H, W, 0);
      break;
    case YS_BLOCK_C3K2: {
      const int c = (int)(c2 * e);
      if (bd.n < 1) { ys_set_error("ys_block_create: n = %d", bd.n); return YS_ERR_UNSUPPORTED; }
      if (bad(c, "C3k2 hidden width") || bad((int)(c * 0.5f), "C3k2 inner hidden width")) return YS_ERR_UNSUPPORTED;
      add_c3k2(m, nm, vin, vout, c1, c2, bd.n, bd.c3k != 0, e, H, W, 0);
      break;
    }
    case YS_BLOCK_SPPF:
      if (c1 != c2) { ys_set_error("ys_block_create: SPPF needs c1 == c2 (the graphs' only use)"); return YS_ERR_UNSUPPORTED; }
      if (bad(c1 / 2, "SPPF hidden width")) return YS_ERR_UNSUPPORTED;
      add_sppf(m, nm, vin, vout, c1, H, W, 0);
      break;
    case YS_BLOCK_C2PSA:
      if (c1 != c2 || (c1 / 2) % 64 || bd.n < 1) { ys_set_error("ys_block_create: C2PSA needs c1 == c2, (c1/2) %% 64 == 0 and n >= 1"); return YS_ERR_UNSUPPORTED; }
      add_c2psa(m, nm, vin, vout, c1, bd.n, H, W, 0);
      break;
    case YS_BLOCK_PROTO: {
      if (bad(bd.n, "Proto hidden width (n)")) return YS_ERR_UNSUPPORTED;
      // add_proto allocates its own output buffer: use it as the block output
      m->bufs.pop_back();
      m->blk_out = add_proto(m, nm, vin, c1, bd.n, c2, H, W, 0);
      break;
    }
    default:
      ys_set_error("ys_block_create: unknown block kind %d", bd.kind);
      return YS_ERR_INVALID_ARG;
  }
  return YS_OK;
}

// ---- parameter layout: [segment][group] contiguous ranges; group rule of YoloBaseTaskModel.cs:144-151 made disjoint:
//      0 = "bias" (conv bias, bn.bias), 1 = conv "weight", 2 = "bn" weight
int layout_params(ys_model* m) {
  long off = 0;
  for (int seg = 0; seg < ys_model::NSEG; seg++)
    for (int grp = 0; grp < 3; grp++) {
      const long start = off;
      for (auto& c : m->convs) {
        if (c.seg != seg) continue;
        if (grp == 0) { if (c.bn) { c.b_off = off; off += c.cout; } else { c.g_off = off; off += c.cout; } }   // bn.bias | conv bias (stored in g_off for plain convs)
        if (grp == 1) { c.w_off = off; off += c.dw ? (long)c.cout * 9 : c.ct ? 4L * c.cout * c.cin : (long)c.cout * c.k * c.k * c.cin; }
        if (grp == 2 && c.bn) { c.g_off = off; off += c.cout; }
      }
      m->seg_group[seg][grp] = {start, off - start};
    }
  m->n_params = off;
  m->n_params_real = off;
  for (auto& c : m->convs)
    if (c.cout_real != c.cout) m->n_params_real -= (long)(c.cout - c.cout_real) * ((long)c.k * c.k * c.cin + (c.bn ? 2 : 1));
  long so = 0;
  for (auto& c : m->convs)
    if (c.bn) { c.rm_off = so; so += c.cout; c.rv_off = so; so += c.cout; c.nbt_off = so; so += 1; }
  m->n_state = so;
  // state_dict listing: parameters in module REGISTRATION order, then buffers (TorchSharp named_parameters + named_buffers)
  std::vector<int> reg2;
  {
    std::vector<int> seen_e;
    std::vector<int> covered(m->convs.size(), 0);
    for (int e : m->reg) {
      if (std::find(seen_e.begin(), seen_e.end(), e) != seen_e.end()) continue;
      seen_e.push_back(e); reg2.push_back(e);
      covered[e & 0xfffff] += 1;
    }
    for (size_t i = 0; i < m->convs.size(); i++) {
      const int want = m->convs[i].members.empty() ? 1 : (int)m->convs[i].members.size();
      if (covered[i] != want) { ys_set_error("internal: registration list covers %d of %d modules of conv %zu (%s)", covered[i], want, i, m->convs[i].name.c_str()); return YS_ERR_STATE; }
    }
  }
  // (module name, first output row, rows) of a registration entry
  auto module_of = [&](int e, std::string& name, int& row0, int& rows) {
    const ConvL& c = m->convs[e & 0xfffff];
    if (c.members.empty()) { name = c.name; row0 = 0; rows = c.cout_real; }
    else { const ConvL::Member& mb = c.members[e >> 20]; name = mb.name; row0 = mb.row0; rows = mb.rows; }
  };
  for (int e : reg2) {
    const int i = e & 0xfffff;
    const ConvL& c = m->convs[i];
    std::string nm; int r0, nr;
    module_of(e, nm, r0, nr);
    const long wrow = c.dw ? 9 : (long)c.k * c.k * c.cin;
    if (c.bn) {
      if (c.dw) add_tensor(m, nm + ".conv.weight", 4, i, c.w_off, {nr, 1, 3, 3}, true);
      else add_tensor(m, nm + ".conv.weight", 0, i, c.w_off + r0 * wrow, {nr, c.cin, c.k, c.k}, true);
      add_tensor(m, nm + ".bn.weight", 1, i, c.g_off + r0, {nr}, true);
      add_tensor(m, nm + ".bn.bias", 1, i, c.b_off + r0, {nr}, true);
    } else {
      if (c.ct) add_tensor(m, nm + ".weight", 5, i, c.w_off, {c.cin, c.cout, 2, 2}, true);   // ConvTranspose2d: [Cin][Cout][kh][kw]
      else add_tensor(m, nm + ".weight", 0, i, c.w_off + r0 * wrow, {nr, c.cin, c.k, c.k}, true);
      add_tensor(m, nm + ".bias", 1, i, c.g_off + r0, {nr}, true);
    }
    if (e == m->dfl_after_conv)
      add_tensor(m, m->head_prefix + ".dfl.conv.weight", 3, -1, 0, {1, m->d.reg_max, 1, 1}, true);   // Block.cs:28-30 (never trained, Head.cs:221)
  }
  for (int e : reg2) {
    const int i = e & 0xfffff;
    const ConvL& c = m->convs[i];
    if (!c.bn) continue;
    std::string nm; int r0, nr;
    module_of(e, nm, r0, nr);
    add_tensor(m, nm + ".bn.running_mean", 2, i, c.rm_off + r0, {nr}, false);
    add_tensor(m, nm + ".bn.running_var", 2, i, c.rv_off + r0, {nr}, false);
    add_tensor(m, nm + ".bn.num_batches_tracked", 2, i, c.nbt_off, {1}, false);   // members of a fused unit share the counter: they always step together
  }
  return YS_OK;
}

// ---- batched weight preparation: fp32 master [Cout][taps][Cin] -> T forward [Cout][taps][Cin_pad]
//                                                                  -> T dgrad   [Cin][taps flipped][Cout_pad]
template <class T>
__global__ void __launch_bounds__(256)
weight_prep_all_kernel(const float* __restrict__ params, const PrepDesc* __restrict__ desc, int n, long total_f, long total_d,
                       T* __restrict__ wf_all, T* __restrict__ wd_all, const float* __restrict__ amax_w,
                       unsigned char* __restrict__ wf8_all, unsigned char* __restrict__ wd8_all) {
  // fp8 mode: e4m3 copies of both shadows (same element order) with the layer's current scale 448 / amax(|W|)
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total_f) {
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].nf_start <= i) lo = mid; else hi = mid - 1; }
    const PrepDesc d = desc[lo];
    const long e = i - d.nf_start;
    const int ci = (int)(e % d.cin_pad);
    const long r = e / d.cin_pad;
    const T tv = Elem<T>::from_f(ci < d.cin_real ? params[d.w_off + r * d.cin_real + ci] : 0.f);
    wf_all[d.wf_off + e] = tv;
    if (wf8_all) { const float aw = amax_w[lo]; wf8_all[d.wf_off + e] = ys_f32_to_e4m3_dev(Elem<T>::to_f(tv) * (aw > 0.f ? YS_E4M3_MAX / aw : 1.0f)); }
  }
  if (i < total_d) {
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].nd_start <= i) lo = mid; else hi = mid - 1; }
    const PrepDesc d = desc[lo];
    if (d.has_wd) {
      const long e = i - d.nd_start;
      int co, ci, tap;
      if (d.phase) {          // stride-2 3x3 layer on the bf16 path: phase-major dgrad weights (conv_dgrad_s2_phases)
        ys_phase_wd_index(e, d.cin_real, d.cout_pad, ci, tap, co);
      } else {
        co = (int)(e % d.cout_pad);
        const long r = e / d.cout_pad;
        const int tapf = (int)(r % d.taps);
        ci = (int)(r / d.taps);
        tap = d.taps - 1 - tapf;
      }
      const T tv = Elem<T>::from_f(co < d.cout && ci < d.cin_real ? params[d.w_off + ((long)co * d.taps + tap) * d.cin_real + ci] : 0.f);
      wd_all[d.wd_off + e] = tv;
      if (wd8_all) { const float aw = amax_w[lo]; wd8_all[d.wd_off + e] = ys_f32_to_e4m3_dev(Elem<T>::to_f(tv) * (aw > 0.f ? YS_E4M3_MAX / aw : 1.0f)); }
    }
  }
}

// Round 5: the bf16 form of the above as three block ranges of ONE launch.  The element-per-thread kernel spent its time in two 8-step binary searches over the
// descriptor table and three 64-bit divisions PER ELEMENT, stored 2 bytes per lane, and read the master weights of the dgrad shadow with a stride of taps * Cin floats
// between neighbouring lanes: 48 us per YOLOv8n step, 1.11 ms per YOLOv8x step -- a tenth of the HBM rate for a pass that moves 8 bytes per parameter.
//   blocks [0, nbF):        forward shadow, 8 consecutive elements per thread (one search, 32-bit index arithmetic, two 16-byte loads, one 16-byte store)
//   blocks [nbF, nbF+nbT):  dgrad shadow of the plain layers as 64 (cout) x 64 (cin) tiles of one tap through LDS: rows of 256 B read along cin, rows of 128 B written along cout
//   blocks [nbF+nbT, ...):  dgrad shadow of the stride-2 layers (phase-major order, conv_dgrad_s2_phases): the element form, on their own compact table
// Same values as weight_prep_all_kernel<bf16_t> element for element (round-to-nearest-even; e4m3 copies from the ROUNDED bf16 value).
__global__ void __launch_bounds__(256)
weight_prep_fast_kernel(const float* __restrict__ params, const PrepDesc* __restrict__ desc, int n, long total_f8,
                        const PrepDesc* __restrict__ tdesc, int nt, long nbF, long nbT,
                        const PrepDesc* __restrict__ pdesc, int np, long total_p,
                        bf16_t* __restrict__ wf_all, bf16_t* __restrict__ wd_all, const float* __restrict__ amax_w,
                        unsigned char* __restrict__ wf8_all, unsigned char* __restrict__ wd8_all) {
  __shared__ float sT[64][65];
  const long blk = blockIdx.x;
  const int tid = threadIdx.x;
  if (blk < nbF) {
    const long u = blk * 256 + tid;            // 8-element unit
    if (u >= total_f8) return;
    const long i = u * 8;
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].nf_start <= i) lo = mid; else hi = mid - 1; }
    const PrepDesc d = desc[lo];
    const unsigned e = (unsigned)(i - d.nf_start);
    const unsigned r = e / (unsigned)d.cin_pad, ci = e - r * (unsigned)d.cin_pad;     // cin_pad is a multiple of 8: the unit stays inside one row
    const long src = d.w_off + (long)r * d.cin_real + ci;
    float f[8];
    if ((int)ci + 8 <= d.cin_real && (src & 3) == 0) {
      const float4 a = *(const float4*)(params + src), b = *(const float4*)(params + src + 4);
      f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) f[k] = (int)ci + k < d.cin_real ? params[src + k] : 0.f;
    }
    const uint4 pk = ys_pack<bf16_t>(f);
    *(uint4*)(wf_all + d.wf_off + e) = pk;
    if (wf8_all) {
      const float aw = amax_w[lo], sc = aw > 0.f ? YS_E4M3_MAX / aw : 1.0f;
      float g[8];
      ys_unpack<bf16_t>(pk, g);
      unsigned w0 = 0, w1 = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) { w0 |= (unsigned)ys_f32_to_e4m3_dev(g[k] * sc) << (8 * k); w1 |= (unsigned)ys_f32_to_e4m3_dev(g[4 + k] * sc) << (8 * k); }
      *(uint2*)(wf8_all + d.wf_off + e) = make_uint2(w0, w1);
    }
    return;
  }
  if (blk < nbF + nbT) {
    const long t = blk - nbF;
    int lo = 0, hi = nt - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tdesc[mid].tile_start <= t) lo = mid; else hi = mid - 1; }
    const PrepDesc d = tdesc[lo];
    const int tl = (int)(t - d.tile_start);
    const int per_tap = d.tiles_ci * d.tiles_co;
    const int tapf = tl / per_tap, rem = tl - tapf * per_tap;
    const int tci = rem / d.tiles_co, tco = rem - tci * d.tiles_co;
    const int tap = d.taps - 1 - tapf;                 // spatial flip of a square kernel
    const int co0 = tco * 64, ci0 = tci * 64;
    // load: thread -> (row = tid / 16 + 16 rr, 4 consecutive cin)
    const int lr = tid >> 4, lc = (tid & 15) * 4;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const int co = co0 + lr + 16 * rr, ci = ci0 + lc;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (co < d.cout && ci < d.cin_real) {
        const long src = d.w_off + ((long)co * d.taps + tap) * d.cin_real + ci;
        if (ci + 4 <= d.cin_real && (src & 3) == 0) { const float4 a = *(const float4*)(params + src); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
        else {
#pragma unroll
          for (int k = 0; k < 4; k++) if (ci + k < d.cin_real) v[k] = params[src + k];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) sT[lr + 16 * rr][lc + k] = v[k];
    }
    __syncthreads();
    // store: thread -> (cin row = tid / 4, 16 consecutive cout)
    const int sr = tid >> 2, sc0 = (tid & 3) * 16;
    const int ci = ci0 + sr;
    if (ci < d.cin_pad) {
      const float aw = wd8_all ? amax_w[d.layer] : 0.f, sc = aw > 0.f ? YS_E4M3_MAX / aw : 1.0f;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int co = co0 + sc0 + 8 * h;
        if (co < d.cout_pad) {                           // cout_pad is a multiple of 8
          float f[8];
#pragma unroll
          for (int k = 0; k < 8; k++) f[k] = sT[sc0 + 8 * h + k][sr];
          const uint4 pk = ys_pack<bf16_t>(f);
          const long dst = d.wd_off + ((long)ci * d.taps + tapf) * d.cout_pad + co;
          *(uint4*)(wd_all + dst) = pk;
          if (wd8_all) {
            float g[8];
            ys_unpack<bf16_t>(pk, g);
            unsigned w0 = 0, w1 = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { w0 |= (unsigned)ys_f32_to_e4m3_dev(g[k] * sc) << (8 * k); w1 |= (unsigned)ys_f32_to_e4m3_dev(g[4 + k] * sc) << (8 * k); }
            *(uint2*)(wd8_all + dst) = make_uint2(w0, w1);
          }
        }
      }
    }
    return;
  }
  {
    const long i = (blk - nbF - nbT) * 256 + tid;
    if (i >= total_p) return;
    int lo = 0, hi = np - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pdesc[mid].nd_start <= i) lo = mid; else hi = mid - 1; }
    const PrepDesc d = pdesc[lo];
    const long e = i - d.nd_start;
    int co, ci, tap;
    ys_phase_wd_index(e, d.cin_real, d.cout_pad, ci, tap, co);
    const bf16_t tv = Elem<bf16_t>::from_f(co < d.cout && ci < d.cin_real ? params[d.w_off + ((long)co * d.taps + tap) * d.cin_real + ci] : 0.f);
    wd_all[d.wd_off + e] = tv;
    if (wd8_all) { const float aw = amax_w[d.layer]; wd8_all[d.wd_off + e] = ys_f32_to_e4m3_dev(Elem<bf16_t>::to_f(tv) * (aw > 0.f ? YS_E4M3_MAX / aw : 1.0f)); }
  }
}

int prep_weights(ys_model* m) {
  if (!m->weights_dirty) return YS_OK;
  const long total = std::max(m->prep_nf, m->prep_nd);
  if (m->f8) YS_TRY(ys_f8_weight_amax_launch(m->ctx->stream, m->params, m->f8_layers, m->n_prep, m->amax_w));
  if (m->dtype == YS_BF16 && m->prep_fast) {
    const long nbF = ys_cdiv(m->prep_nf / 8, 256), nbT = m->prep_tiles, nbP = ys_cdiv(m->prep_nd_phase, 256);
    YS_LAUNCH(weight_prep_fast_kernel, (unsigned)(nbF + nbT + nbP), 256, m->ctx->stream, (const float*)m->params, (const PrepDesc*)m->prep_dev, m->n_prep, m->prep_nf / 8,
              (const PrepDesc*)m->prep_tile_dev, m->n_prep_tile, nbF, nbT, (const PrepDesc*)m->prep_phase_dev, m->n_prep_phase, m->prep_nd_phase,
              (bf16_t*)m->wf_all, (bf16_t*)m->wd_all, (const float*)m->amax_w, m->wf8_all, m->wd8_all);
  } else if (m->dtype == YS_BF16)
    YS_LAUNCH((weight_prep_all_kernel<bf16_t>), ys_cdiv(total, 256), 256, m->ctx->stream, (const float*)m->params, (const PrepDesc*)m->prep_dev, m->n_prep, m->prep_nf, m->prep_nd, (bf16_t*)m->wf_all, (bf16_t*)m->wd_all, (const float*)m->amax_w, m->wf8_all, m->wd8_all);
  else
    YS_LAUNCH((weight_prep_all_kernel<float>), ys_cdiv(total, 256), 256, m->ctx->stream, (const float*)m->params, (const PrepDesc*)m->prep_dev, m->n_prep, m->prep_nf, m->prep_nd, (float*)m->wf_all, (float*)m->wd_all, (const float*)nullptr, (unsigned char*)nullptr, (unsigned char*)nullptr);
  m->weights_dirty = false;
  return YS_OK;
}

inline char* view_ptr(ys_model* m, void* base, const Buf& b, long row0) { return (char*)base + (size_t)row0 * b.ldc * m->es; }

float* chan_ptr(ys_model* m, const ConvL& c, int which) { return m->chan + c.ch_off + (long)which * ((c.cout + 3) / 4 * 4); }

void dev_free_tracked(ys_model* m, void* p) {
  if (!p) return;
  for (size_t i = 0; i < m->allocs.size(); i++) if (m->allocs[i] == p) { m->allocs.erase(m->allocs.begin() + i); break; }
  hipFree(p);
}

// Ground-truth workspace for `gcap` labels PER IMAGE (the reference pads every image to the batch's largest label count,
// Loss.cs:363-390, without a cap): raw label staging, padded GT arrays and the [B][gcap][A] assignment matrices.
int alloc_label_ws(ys_model* m, int gcap) {
  const int B = m->maxB;
  void* old[] = {m->lab_bidx, m->lab_cls, m->lab_box, m->gt_count, m->gt_box, m->gt_cls, m->ov, m->align, m->mpos, m->pos_align, m->pos_ov, m->kp_dev};
  if (m->lab_bidx) YS_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
  for (void* p : old) dev_free_tracked(m, p);
  m->lab_bidx = m->lab_cls = m->lab_box = nullptr; m->gt_count = nullptr; m->gt_box = nullptr; m->gt_cls = nullptr;
  m->ov = m->align = nullptr; m->mpos = nullptr; m->pos_align = m->pos_ov = nullptr; m->kp_dev = nullptr;
  m->gcap = gcap;
  m->max_labels = gcap * B;
  const size_t GA = (size_t)B * gcap * m->A;
  YS_TRY(dev_alloc(m, (void**)&m->lab_bidx, (size_t)m->max_labels * 4));
  YS_TRY(dev_alloc(m, (void**)&m->lab_cls, (size_t)m->max_labels * 4));
  YS_TRY(dev_alloc(m, (void**)&m->lab_box, (size_t)m->max_labels * 20));   // 4 floats per label, 5 for oriented boxes
  YS_TRY(dev_alloc(m, (void**)&m->gt_count, (size_t)B * 4));
  YS_TRY(dev_alloc(m, (void**)&m->gt_box, (size_t)B * gcap * 20));        // xyxy, or xywh + angle (OBB)
  YS_TRY(dev_alloc(m, (void**)&m->gt_cls, (size_t)B * gcap * 12));  // gt_cls + gt_valid + gt_src
  YS_TRY(dev_alloc(m, (void**)&m->ov, GA * 4));
  YS_TRY(dev_alloc(m, (void**)&m->align, GA * 4));
  YS_TRY(dev_alloc(m, (void**)&m->mpos, GA));
  YS_TRY(dev_alloc(m, (void**)&m->pos_align, (size_t)B * gcap * 4));
  YS_TRY(dev_alloc(m, (void**)&m->pos_ov, (size_t)B * gcap * 4));
  if (m->xkind == 3) YS_TRY(dev_alloc(m, (void**)&m->kp_dev, (size_t)m->max_labels * m->nm * 4));
  return YS_OK;
}

// fp8 mode: which convolutions may run the fp8 kernel, the e4m3 weight shadows and the amax / scale slots (recipe: f8.hip)
int alloc_f8(ys_model* m, const std::vector<PrepDesc>& pd) {
  hipStream_t st = m->ctx->stream;
  const int nb = (int)m->bufs.size(); (void)nb;
  std::vector<F8Layer> layers(pd.size());
  for (size_t i = 0; i < pd.size(); i++) { layers[i].w_off = pd[i].w_off; layers[i].count = (long)pd[i].cout * pd[i].taps * pd[i].cin_real; }
  std::vector<F8Conv> fc(m->convs.size());
  size_t q8_bytes = 0;
  for (auto& c : m->convs) {
    F8Conv& f = fc[c.idx];
    f.layer = c.prep_idx >= 0 ? c.prep_idx : 0;
    const bool dense = !c.first && !c.dw && !c.ct && c.bn;       // the plain biased head outputs feed the loss directly: kept in bf16
    c.f8_fwd = dense && c.cin_pad % 32 == 0 && c.cin == c.cin_pad;
    c.f8_bwd = dense && c.cout % 32 == 0 && c.cout_ld == c.cout;
    if (c.f8_fwd) q8_bytes = std::max(q8_bytes, (size_t)m->maxB * c.Hin * c.Win * c.cin_pad);
    if (c.f8_bwd) q8_bytes = std::max(q8_bytes, (size_t)m->maxB * c.Hout * c.Wout * c.cout_ld);
  }
  if (q8_bytes) YS_TRY(dev_alloc(m, (void**)&m->q8, q8_bytes));
  m->n_f8_convs = (int)fc.size();
  YS_TRY(dev_alloc(m, (void**)&m->wf8_all, (size_t)m->n_wf_pending));
  YS_TRY(dev_alloc(m, (void**)&m->wd8_all, (size_t)m->n_wd_pending));
  YS_TRY(dev_alloc(m, (void**)&m->amax_w, pd.size() * 4));
  YS_TRY(dev_alloc(m, (void**)&m->amax_act, fc.size() * YS_AMAX_WAYS * 4));
  YS_TRY(dev_alloc(m, (void**)&m->amax_dy, fc.size() * YS_AMAX_WAYS * 4));
  YS_TRY(dev_alloc(m, (void**)&m->f8_scales, fc.size() * 16));
  YS_TRY(dev_alloc(m, (void**)&m->f8_layers, layers.size() * sizeof(F8Layer)));
  YS_TRY(dev_alloc(m, (void**)&m->f8_convs, fc.size() * sizeof(F8Conv)));
  YS_CHECK_HIP(hipMemcpyAsync(m->f8_layers, layers.data(), layers.size() * sizeof(F8Layer), hipMemcpyHostToDevice, st));
  YS_CHECK_HIP(hipMemcpyAsync(m->f8_convs, fc.data(), fc.size() * sizeof(F8Conv), hipMemcpyHostToDevice, st));
  YS_CHECK_HIP(hipStreamSynchronize(st));
  return YS_OK;
}

int allocate(ys_model* m) {
  const int B = m->maxB;
  hipStream_t st = m->ctx->stream; (void)st;
  for (auto& b : m->bufs) {
    const size_t bytes = (size_t)B * b.rows_per_b * b.ldc * m->es;
    YS_TRY(dev_alloc(m, &b.act, bytes));
    if (b.need_grad) YS_TRY(dev_alloc(m, &b.grad, bytes));
  }
  YS_TRY(dev_alloc(m, (void**)&m->params, (size_t)m->n_params * 4));
  YS_TRY(dev_alloc(m, (void**)&m->grads, (size_t)m->n_params * 4));
  YS_TRY(dev_alloc(m, (void**)&m->adam_m, (size_t)m->n_params * 4));
  YS_TRY(dev_alloc(m, (void**)&m->adam_v, (size_t)m->n_params * 4));
  YS_TRY(dev_alloc(m, (void**)&m->state, (size_t)m->n_state * 4));
  long nf = 0, nd = 0, ny = 0, nch = 0, dy_max = 0, stat_max = 0, amax = 0;
  std::vector<PrepDesc> pd;
  for (auto& c : m->convs) {
    const int taps = c.k * c.k;
    if (c.dw) {   // depthwise: fp32 master weights are used directly
      const long M = (long)B * c.Hout * c.Wout;
      c.y_off = ny; ny += M * c.cout;
      c.ch_off = nch; nch += 6L * ((c.cout + 3) / 4 * 4);
      dy_max = std::max(dy_max, M * c.cout_ld);
      stat_max = std::max(stat_max, 2048L * 2 * c.cout_ld);
      stat_max = std::max(stat_max, 1024L * 9 * c.cout);   // depthwise weight-gradient partials (ys_dwconv_wgrad_blocks <= 1024)
      continue;
    }
    c.wf_off = nf;
    c.wd_off = nd;
    c.prep_idx = (int)pd.size();
    for (int ph = 0; ph < (c.ct ? 4 : 1); ph++) {   // ConvTranspose: one 1x1 weight matrix per output phase
      PrepDesc d{}; d.w_off = c.w_off + (long)ph * c.cout * c.cin; d.wf_off = nf; d.wd_off = nd; d.cout = c.cout; d.taps = taps;
      d.cin_real = c.cin; d.cin_pad = c.cin_pad; d.cout_pad = c.cout_ld; d.has_wd = c.first ? 0 : 1;
      d.phase = (!c.ct && ys_conv_dgrad_uses_phases(m->dtype, c.k, c.s)) ? 1 : 0;
      d.nf_start = nf; d.nd_start = nd;
      nf += (long)c.cout * taps * c.cin_pad;
      if (!c.first) nd += (long)c.cin_pad * taps * c.cout_ld;   // cin_pad > cin (Pose towers): zero rows -> zero input-gradient pad channels
      pd.push_back(d);
    }
    const long M = (long)B * c.Hout * c.Wout;
    if (c.bn) { c.y_off = ny; ny += M * c.cout; }
    c.ch_off = nch; nch += 6L * ((c.cout + 3) / 4 * 4);   // 16-byte aligned coefficient vectors
    dy_max = std::max(dy_max, M * c.cout_ld);
    stat_max = std::max(stat_max, 2L * ys_cdiv(M, 64) * 2 * c.cout);   // conv epilogue partials (smallest pixel tile, ragged 2-D tiles)
    stat_max = std::max(stat_max, 2048L * 2 * c.cout_ld);   // channel-reduction partials (<= 2048 workgroups)
  }
  for (auto& op : m->ops) if (op.type == OP_MAXPOOL) { op.aux_off = amax; amax += (long)B * op.H * op.W * op.in.C; }
  long nattn = 0;
  for (auto& op : m->ops) if (op.type == OP_ATTN) { op.aux_off = nattn; nattn += 2L * B * op.heads * (long)(op.H * op.W) * (op.H * op.W); }
  m->n_attn = nattn;
  YS_TRY(dev_alloc(m, (void**)&m->attn_ws, (size_t)nattn * 4));
  m->n_wf = nf; m->n_wd = nd; m->n_y = ny; m->n_chan = nch; m->n_dy = dy_max; m->n_stat = stat_max; m->n_argmax = amax;
  m->prep_nf = nf; m->prep_nd = nd; m->n_prep = (int)pd.size();
  YS_TRY(dev_alloc(m, &m->wf_all, (size_t)nf * m->es));
  YS_TRY(dev_alloc(m, &m->wd_all, (size_t)nd * m->es));
  YS_TRY(dev_alloc(m, (void**)&m->prep_dev, pd.size() * sizeof(PrepDesc)));
  YS_CHECK_HIP(hipMemcpyAsync(m->prep_dev, pd.data(), pd.size() * sizeof(PrepDesc), hipMemcpyHostToDevice, m->ctx->stream));
  {
    // weight_prep_fast_kernel's tables (bf16): the plain layers' transpose tiles and the phase-major layers' elements, each as a compact prefix
    std::vector<PrepDesc> pt, pp;
    long tiles = 0, pe = 0;
    bool ok = m->dtype == YS_BF16;
    for (size_t i = 0; i < pd.size(); i++) {
      PrepDesc d = pd[i];
      d.layer = (int)i;
      if ((d.cin_pad & 7) || (d.cout_pad & 7) || (d.nf_start & 7) || (d.wf_off & 7) || (d.wd_off & 7)) ok = false;
      if (!d.has_wd) continue;
      if (d.phase) { d.nd_start = pe; pe += (long)d.cin_pad * d.taps * d.cout_pad; pp.push_back(d); }
      else {
        d.tiles_ci = ys_cdiv(d.cin_pad, 64); d.tiles_co = ys_cdiv(d.cout_pad, 64); d.tile_start = tiles;
        tiles += (long)d.taps * d.tiles_ci * d.tiles_co; pt.push_back(d);
      }
    }
    if ((nf & 7) || tiles + ys_cdiv(nf / 8, 256) + ys_cdiv(pe, 256) >= (1L << 31)) ok = false;
    m->prep_fast = ok;      // (the element-per-thread kernel stays for fp32 and for tables that miss the 8-element alignment; bit-identity of the two was a round-5 test, profiles/README.md)
    if (m->prep_fast) {
      m->n_prep_tile = (int)pt.size(); m->prep_tiles = tiles; m->n_prep_phase = (int)pp.size(); m->prep_nd_phase = pe;
      if (pt.empty()) pt.push_back(PrepDesc{});
      if (pp.empty()) pp.push_back(PrepDesc{});
      YS_TRY(dev_alloc(m, (void**)&m->prep_tile_dev, pt.size() * sizeof(PrepDesc)));
      YS_TRY(dev_alloc(m, (void**)&m->prep_phase_dev, pp.size() * sizeof(PrepDesc)));
      YS_CHECK_HIP(hipMemcpyAsync(m->prep_tile_dev, pt.data(), pt.size() * sizeof(PrepDesc), hipMemcpyHostToDevice, m->ctx->stream));
      YS_CHECK_HIP(hipMemcpyAsync(m->prep_phase_dev, pp.data(), pp.size() * sizeof(PrepDesc), hipMemcpyHostToDevice, m->ctx->stream));
      YS_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));   // pt / pp are host temporaries
    }
  }
  YS_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));   // pd is a host temporary
  m->n_wf_pending = nf; m->n_wd_pending = nd;
  if (m->f8) YS_TRY(alloc_f8(m, pd));
  YS_TRY(dev_alloc(m, &m->y_all, (size_t)ny * m->es));
  YS_TRY(dev_alloc(m, &m->dy_scratch, (size_t)dy_max * m->es));
  // measured on MI355X (YOLOv8n B=64): +1.4 % step throughput, but both streams' kernels fill the CUs' LDS, so they mostly
  // time-slice and every per-kernel duration inflates; off by default (YS_OVERLAP=1 enables it)
  // Round 3: on by default.  With the split reduction deferred to one launch per segment the weight-gradient kernels are independent
  // of everything until the segment ends, and both chains are latency-bound: measured 10.50 -> 10.30 ms/step (+2 %) on config 2
  // (round 2, before the deferral: +1.4 %).  Co-running kernels time-slice the CUs, so PER-KERNEL durations inflate (conv class +7 %):
  // bench.py switches the overlap off (ys_model_set_overlap) for its per-kernel profile steps.  YS_OVERLAP=0 disables it.
  m->overlap = (YS_OPT_INT("OVERLAP", 1) != 0);
  m->overlap_built = m->overlap;
  if (m->overlap) {
    // lowest priority: the weight gradients are off the critical path (the optimizer is their only reader), and a priority class of
    // its own is a hardware queue of its own -- streams of one class share a handful of queues in creation order, and with the main
    // stream and this one on the SAME queue nothing overlaps (seen with an eagerly initialised RCCL communicator, whose streams
    // shifted the assignment: 11.6 instead of 10.2 ms/step).
    {
      int least = 0, greatest = 0;
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
        YS_CHECK_HIP(hipStreamCreateWithPriority(&m->st2, hipStreamNonBlocking, least));     // (round 6: the HIGHEST priority instead: 8.61 / 8.61 / 8.63 -> 8.65 / 8.65 / 8.66 ms)
      else
        YS_CHECK_HIP(hipStreamCreateWithFlags(&m->st2, hipStreamNonBlocking));
    }
    // (rounds 3-5: a ring of DY_RING dy_max-sized buffers guarded by events; round 6: one buffer per unit, sized for that unit -- sum of the BN outputs, 1.9 GB for
    // YOLOv8n at B = 64, ~10 GB for YOLOv8x at 1280 x 1280 x 16)
    for (auto& c : m->convs) {
      if (!c.bn || c.dw || c.ct || c.dy_own) continue;
      YS_TRY(dev_alloc(m, &c.dy_own, (size_t)B * c.Hout * c.Wout * c.cout_ld * m->es));
    }
    for (int k = 0; k <= ys_model::DY_RING; k++) YS_CHECK_HIP(hipEventCreateWithFlags(&m->ev_dy[k], hipEventDisableTiming));
    YS_CHECK_HIP(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
  }
  for (int k = 0; k < ys_model::NSEG; k++) {
    YS_CHECK_HIP(hipEventCreateWithFlags(&m->ev_seg_m[k], hipEventDisableTiming));
    YS_CHECK_HIP(hipEventCreateWithFlags(&m->ev_seg_w[k], hipEventDisableTiming));
  }
  {
    // grouped head stages: every unit keeps its own statistics rows (conv -> grouped finalize) and its own dy (BN backward -> wgrad / dgrad)
    long gst = 0;
    for (auto& c : m->convs) {
      if (c.group < 0) continue;
      const long M = (long)B * c.Hout * c.Wout;
      c.gstat_off = gst; gst += (2L * ys_cdiv(M, 64) * 2 * c.cout_ld + 63) / 64 * 64;
      if (c.bn && !c.dy_own) YS_TRY(dev_alloc(m, &c.dy_own, (size_t)M * c.cout_ld * m->es));
    }
    if (gst) YS_TRY(dev_alloc(m, (void**)&m->stat_group, (size_t)gst * 4));
  }
  YS_TRY(dev_alloc(m, (void**)&m->chan, (size_t)nch * 4));
  YS_TRY(dev_alloc(m, (void**)&m->stat_partial, (size_t)stat_max * 4));
  // statistics accumulators of the BatchNorm units (round 5, ys_kernels.h ys_stat_acc_add).  Off in fp8 mode (its apply pass also writes the e4m3 image)
  // and by BN_ATOMIC=0 (A/B switch against the row + bn_finalize form)
  m->bn_atomic = YS_OPT_INT("BN_ATOMIC", 1) != 0 && !m->f8;
  if (m->bn_atomic) {
    long off = 0;
    for (auto& c : m->convs) if (c.bn && !c.dw && !c.ct) { c.acc_off = off; off += (long)YS_STAT_SHARDS * c.cout * 2; }
    m->n_stat_acc = off;
    if (off > 0) YS_TRY(dev_alloc(m, (void**)&m->stat_acc_all, (size_t)off * 8));
    else m->bn_atomic = false;
  }
  // (BN-backward sums through exact integer accumulators + a finalize inside the apply pass -- BNB_ATOMIC, round 5: identical results, 51 launches fewer, NOT faster
  // (8.94-8.97 against 8.83-8.94 ms) -- was deleted in round 6; the record is in profiles/README.md round 5.)
  // (Two experiments lived here through round 4 and are gone: side-stream "head lanes" for the P4 / P5 towers -- 10.52-10.55 ms/step against 9.98-10.01 without,
  // a side-stream kernel that takes CU slots turns the persistent P3 kernels' equal tile shares into a tail -- and the last-arriver "ticket" BatchNorm finalize inside
  // the producing convolution, +0.9 ms/step: DESIGN.md 6b / 6d.)
  YS_TRY(dev_alloc(m, (void**)&m->argmax, (size_t)amax));
  // wgrad partial workspace: a shared scratch (max over layers of splits * |W|: ConvTranspose phases, immediate reduction) followed by
  // one region per convolution, so that the split reduction of a whole backward segment can run as ONE launch after it
  m->defer_wgred = true;
  m->stem_on = (YS_OPT_INT("STEM_DIRECT", 1) != 0);
  m->bnred_on = (YS_OPT_INT("BNRED", 1) != 0);           // YS_BNRED=0: every BN backward runs its own reduction pass
  long wgp = 0, wgp_regions = 0;
  std::vector<long> need(m->convs.size(), 0);
  for (auto& c : m->convs) {
    if (c.dw) continue;
    WgradArgs a{}; a.Cin = c.cin_pad; a.Cout = c.cout; a.KH = a.KW = c.k; a.M = (int)((long)B * c.Hout * c.Wout);
    a.B = B; a.Hin = c.Hin; a.Win = c.Win; a.Hout = c.Hout; a.Wout = c.Wout; a.stride = c.s; a.pad = c.k / 2;
    {                                        // the view geometry the launch will see (the blocked-GEMM plan depends on it)
      const Buf& ib = m->bufs[c.in.buf]; const Buf& ob = m->bufs[c.out.buf];
      a.in_ldc = ib.ldc; a.in_coff = c.in.coff; a.in_bstride = ib.rows_per_b;
      if (c.bn) { a.dy_ldc = c.cout; a.dy_coff = 0; a.dy_bstride = (long)c.Hout * c.Wout; }
      else { a.dy_ldc = ob.ldc; a.dy_coff = c.out.coff; a.dy_bstride = ob.rows_per_b; }
    }
    if (c.ct) { a.Hout = c.Hin; a.Wout = c.Win; a.M = (int)((long)B * c.Hin * c.Win); a.dy_rh = 1; }
    const int sp = ys_wgrad_splits(a, m->dtype);
    need[c.idx] = (long)sp * c.cout * c.k * c.k * c.cin_pad;
    wgp = std::max(wgp, need[c.idx]);
    if (m->defer_wgred && !c.ct) { c.wgp_splits = sp; wgp_regions += (need[c.idx] + 63) / 64 * 64; }
  }
  {
    long off = (wgp + 63) / 64 * 64;
    // descriptor order = backward-segment order, so that a backward_range call reduces one contiguous run of descriptors
    for (int seg = 0; seg < ys_model::NSEG; seg++) {
      m->red_first[seg] = (int)m->red_host.size();
      for (auto& c : m->convs) {
        if (c.seg != seg || c.dw || c.ct || !m->defer_wgred) continue;
        c.wgp_off = off; off += (need[c.idx] + 63) / 64 * 64;
        c.red_slot = (int)m->red_host.size();
        m->red_host.push_back(WgRedDesc{});
      }
    }
    m->red_first[ys_model::NSEG] = (int)m->red_host.size();
    m->n_wgp = off;
  }
  YS_TRY(dev_alloc(m, (void**)&m->wg_partial, (size_t)m->n_wgp * 4));
  if (!m->red_host.empty()) YS_TRY(dev_alloc(m, (void**)&m->red_dev, m->red_host.size() * sizeof(WgRedDesc)));
  const ys_model_desc& d = m->d;
  YS_TRY(dev_alloc(m, (void**)&m->img_dev, (size_t)B * std::max(3, m->blk_c1) * d.height * d.width * 4));
  YS_TRY(dev_alloc(m, (void**)&m->pred, (size_t)B * (4 + d.nc + m->nm) * m->A * 4));
  m->n_out_stage = std::max((long)B * m->A * std::max(std::max(m->ld_pd, m->ld_ps), 4 + d.nc + m->nm), (long)B * m->mh * m->mw * std::max(m->ld_pr, 1));
  if (m->is_block) { const Buf& ob = m->bufs[m->blk_out]; m->n_out_stage = std::max(m->n_out_stage, (long)B * ob.rows_per_b * ob.ldc); }
  if (m->xkind == 3) {   // v8PoseLoss: foreground list + per-workgroup partials
    YS_TRY(dev_alloc(m, (void**)&m->seg_cnt, (size_t)B * 4));
    YS_TRY(dev_alloc(m, (void**)&m->seg_off, (size_t)(B + 1) * 4));
    YS_TRY(dev_alloc(m, (void**)&m->seg_list, (size_t)B * m->A * 4));
    YS_TRY(dev_alloc(m, (void**)&m->seg_part, (size_t)2 * ys_loss_pose_grid(B, m->A) * 4));
  }
  if (m->segment) {
    YS_TRY(dev_alloc(m, (void**)&m->masks_dev, (size_t)B * m->mh * m->mw * 4));
    YS_TRY(dev_alloc(m, (void**)&m->seg_cnt, (size_t)B * 4));
    YS_TRY(dev_alloc(m, (void**)&m->seg_off, (size_t)(B + 1) * 4));
    YS_TRY(dev_alloc(m, (void**)&m->seg_list, (size_t)B * m->A * 4));
    YS_TRY(dev_alloc(m, (void**)&m->seg_ent, (size_t)B * m->A * 8 * 4));
    YS_TRY(dev_alloc(m, (void**)&m->seg_part, (size_t)B * m->A * 4));
  }
  YS_TRY(dev_alloc(m, (void**)&m->out_stage, (size_t)m->n_out_stage * 4));
  // loss workspace (label-capacity dependent part: alloc_label_ws; grown on demand by ys_loss_detect / ys_model_reserve_labels)
  YS_TRY(alloc_label_ws(m, d.max_labels > 0 ? d.max_labels : 64));
  YS_TRY(dev_alloc(m, (void**)&m->pbox, (size_t)B * m->A * 20));
  YS_TRY(dev_alloc(m, (void**)&m->fg_gt, (size_t)B * m->A * 4));
  YS_TRY(dev_alloc(m, (void**)&m->tnorm, (size_t)B * m->A * 4));
  YS_TRY(dev_alloc(m, (void**)&m->loss_partial, ys_loss_partial_floats(B, m->A) * 4));
  YS_TRY(dev_alloc(m, (void**)&m->scalars, 64 * 4 + 64 * 8 * 8));     // 64 float scalars + the criterion's sharded integer accumulators (loss.hip LOSS_ACC: [64][8] u64)
  return YS_OK;
}

// ------------------------------------------------------------------ forward
// depthwise Conv unit: y = dwconv(x) (dense), then BN statistics / apply exactly like the dense path
int run_dwconv_fwd(ys_model* m, const ConvL& c, int B) {
  hipStream_t st = m->ctx->stream;
  const Buf& ib = m->bufs[c.in.buf];
  const Buf& ob = m->bufs[c.out.buf];
  const long M = (long)B * c.Hout * c.Wout;
  void* y = (char*)m->y_all + (size_t)c.y_off * m->es;
  YS_TRY(ys_dwconv_launch(st, m->dtype, 0, ib.act, ib.ldc, c.in.coff, B, c.Hin, c.Win, c.cout, m->params + c.w_off, y, c.cout, 0, 0));
  if (m->training) {
    int nblk = 0;
    YS_TRY(ys_chan_stats_launch(st, m->dtype, y, M, c.cout, m->stat_partial, &nblk));
    YS_TRY(ys_bn_finalize_launch(st, m->stat_partial, nblk, c.cout, M, m->params + c.g_off, m->params + c.b_off, 1e-3f, 0.03f,
                                 m->state + c.rm_off, m->state + c.rv_off, m->state + c.nbt_off, chan_ptr(m, c, 0),
                                 chan_ptr(m, c, 1), chan_ptr(m, c, 2), chan_ptr(m, c, 3)));
  }
  const void* res = nullptr; int rl = 0, rc = 0;
  if (c.has_res) { res = m->bufs[c.res.buf].act; rl = m->bufs[c.res.buf].ldc; rc = c.res.coff; }
  YS_TRY(ys_bn_act_apply_launch(st, m->dtype, y, M, c.cout, chan_ptr(m, c, 0), chan_ptr(m, c, 1), c.act ? 1 : 0, res, rl, rc,
                                ob.act, ob.ldc, c.out.coff));
  return YS_OK;
}

// ConvTranspose2d(k=2, s=2, bias): out[b, 2h+dh, 2w+dw, :] = W[dh,dw] x[b,h,w,:] + bias  -> four 1x1 GEMMs with a strided output map
int run_convT_fwd(ys_model* m, const ConvL& c, int B) {
  hipStream_t st = m->ctx->stream;
  const Buf& ib = m->bufs[c.in.buf];
  const Buf& ob = m->bufs[c.out.buf];
  for (int ph = 0; ph < 4; ph++) {
    const int dh = ph >> 1, dw = ph & 1;
    ConvArgs a{};
    a.x = ib.act; a.w = (char*)m->wf_all + (size_t)(c.wf_off + (long)ph * c.cout * c.cin_pad) * m->es; a.y = ob.act;
    a.B = B; a.Hin = c.Hin; a.Win = c.Win; a.Cin = c.cin_pad; a.Hout = c.Hin; a.Wout = c.Win; a.Cout = c.cout; a.KH = a.KW = 1;
    a.SA = 1; a.PAD = 0;
    a.in_ldc = ib.ldc; a.in_coff = c.in.coff; a.in_bstride = ib.rows_per_b;
    a.out_ldc = ob.ldc; a.out_coff = c.out.coff; a.out_bstride = ob.rows_per_b;
    a.out_rh = 4 * c.Win; a.out_rw = 2; a.out_r0 = (long)dh * 2 * c.Win + dw;
    a.vec_ok = (ob.ldc % 4 == 0 && c.out.coff % 4 == 0) ? 1 : 0;
    a.shift = m->params + c.g_off;
    a.M = B * c.Hin * c.Win;
    YS_TRY(ys_conv_launch(st, m->dtype, a));
  }
  return YS_OK;
}

// model.0 of a whole-model handle on the bf16 path, with its own weight-gradient partial region (deferred split reduction)
static const ConvL* stem_conv(const ys_model* m) {
  if (!m->stem_on || m->is_block || m->is_head || m->dtype != YS_BF16 || m->convs.empty()) return nullptr;
  const ConvL& c = m->convs[0];
  if (!c.first || !c.bn || c.dw || c.ct || c.has_res || c.wgp_off < 0 || c.in.buf != m->in_buf) return nullptr;
  if (!ys_stem_eligible(m->dtype, c.cin, c.cout, c.k, c.s)) return nullptr;
  const Buf& ob = m->bufs[c.out.buf];
  if ((ob.ldc & 3) || (c.out.coff & 3)) return nullptr;
  return &c;
}
static bool stem_direct(const ys_model* m) { return stem_conv(m) != nullptr; }

// geometry / operand part of the forward convolution arguments of layer c
static ConvArgs fwd_args(ys_model* m, const ConvL& c, int B) {
  const Buf& ib = m->bufs[c.in.buf];
  ConvArgs a{};
  a.x = ib.act; a.w = (char*)m->wf_all + (size_t)c.wf_off * m->es;
  a.B = B; a.Hin = c.Hin; a.Win = c.Win; a.Cin = c.cin_pad; a.Hout = c.Hout; a.Wout = c.Wout; a.Cout = c.cout; a.KH = a.KW = c.k;
  a.SA = c.s; a.DIVS = 0; a.DIVM = 0; a.PAD = c.k / 2;
  a.in_ldc = ib.ldc; a.in_coff = c.in.coff; a.in_bstride = ib.rows_per_b;
  a.M = B * c.Hout * c.Wout;
  return a;
}

// finalize-inside-apply operands of BatchNorm unit c (ys_bn_fin_apply_launch): its accumulators + everything bn_finalize_kernel reads and writes
static BnAccFin bn_acc_fin(ys_model* m, const ConvL& c, long count) {
  BnAccFin f{};
  f.acc = m->stat_acc_all + c.acc_off; f.count = (double)count;
  f.gamma = m->params + c.g_off; f.beta = m->params + c.b_off;
  f.run_mean = m->state + c.rm_off; f.run_var = m->state + c.rv_off; f.nbt = m->state + c.nbt_off;
  f.scale = chan_ptr(m, c, 0); f.shift = chan_ptr(m, c, 1); f.mean = chan_ptr(m, c, 2); f.rstd = chan_ptr(m, c, 3);
  f.eps = 1e-3f; f.momentum = 0.03f;
  return f;
}

// `next`: the convolution that runs right after this one, when it reads exactly the view this one writes (else null)
int run_conv_fwd(ys_model* m, const ConvL& c, int B, const ConvL* next = nullptr) {
  if (c.dw) return run_dwconv_fwd(m, c, B);
  if (c.ct) return run_convT_fwd(m, c, B);
  hipStream_t st = m->ctx->stream;
  float* stat_partial = m->stat_partial;
  const Buf& ib = m->bufs[c.in.buf];
  const Buf& ob = m->bufs[c.out.buf];
  if (c.first && m->in_f32) {      // model.0 straight from the fp32 image planes (conv_stem.hip)
    const long Ms = (long)B * c.Hout * c.Wout;
    const void* wf = (char*)m->wf_all + (size_t)c.wf_off * m->es;
    if (m->training) {
      void* y = (char*)m->y_all + (size_t)c.y_off * m->es;
      int gm = 0;
      const bool atomic = m->bn_atomic && c.acc_off >= 0;
      YS_TRY(ys_stem_fwd_launch(st, m->in_f32, B, c.Hin, c.Win, wf, c.cout, y, c.cout, 0, (long)c.Hout * c.Wout, stat_partial, nullptr, nullptr, 0, &gm,
                                atomic ? m->stat_acc_all + c.acc_off : nullptr));
      if (atomic) {
        YS_TRY(ys_bn_fin_apply_launch(st, m->dtype, y, Ms, c.cout, bn_acc_fin(m, c, Ms), c.act ? 1 : 0, nullptr, 0, 0, ob.act, ob.ldc, c.out.coff));
      } else {
      YS_TRY(ys_bn_finalize_launch(st, stat_partial, gm, c.cout, Ms, m->params + c.g_off, m->params + c.b_off, 1e-3f, 0.03f,
                                   m->state + c.rm_off, m->state + c.rv_off, m->state + c.nbt_off, chan_ptr(m, c, 0),
                                   chan_ptr(m, c, 1), chan_ptr(m, c, 2), chan_ptr(m, c, 3)));
      YS_TRY(ys_bn_act_apply_launch(st, m->dtype, y, Ms, c.cout, chan_ptr(m, c, 0), chan_ptr(m, c, 1), c.act ? 1 : 0, nullptr, 0, 0,
                                    ob.act, ob.ldc, c.out.coff, nullptr));
      }
    } else {
      YS_TRY(ys_stem_fwd_launch(st, m->in_f32, B, c.Hin, c.Win, wf, c.cout, ob.act, ob.ldc, c.out.coff, ob.rows_per_b, nullptr,
                                chan_ptr(m, c, 0), chan_ptr(m, c, 1), c.act ? 1 : 0, nullptr));
    }
    return YS_OK;
  }
  ConvArgs a = fwd_args(m, c, B);
  const long M = a.M;
  const bool vec = (ob.ldc % 4 == 0) && (c.out.coff % 4 == 0);
  if (m->f8 && c.f8_fwd) {
    unsigned* slots = m->amax_act + (size_t)c.idx * YS_AMAX_WAYS;
    if (m->f8_sx_valid) {   // fp8 MFMA kernel where one is planned (ys_conv_launch falls back to bf16 otherwise); it records amax(|x|) itself
      a.f8 = 1; a.w8 = m->wf8_all + c.wf_off; a.qscale = m->f8_scales + 4L * c.idx; a.deq = m->f8_scales + 4L * c.idx + 1; a.amax = slots;
      a.q8 = m->q8;
      if (m->q8_fwd_ready == c.idx) { a.x8 = m->q8; a.amax = nullptr; }   // image and maximum written by the producer's BN / SiLU pass
    } else {                // first pass: no scale yet -> bf16 kernels, and a bootstrap pass records the input maximum
      YS_TRY(ys_f8_view_amax_launch(st, ib.act, (long)B * c.Hin * c.Win, c.cin_pad, ib.ldc, c.in.coff, slots));
    }
  }
  unsigned* amax_slot = nullptr;
  if (c.bn && m->training) {
    void* y = (char*)m->y_all + (size_t)c.y_off * m->es;
    a.y = y; a.out_ldc = c.cout; a.out_coff = 0; a.out_bstride = (long)c.Hout * c.Wout; a.vec_ok = (c.cout % 4 == 0);
    a.stats = stat_partial;
    const bool atomic = m->bn_atomic && c.acc_off >= 0 && !a.f8;
    if (atomic) a.stat_acc = m->stat_acc_all + c.acc_off;
    YS_TRY(ys_conv_launch(st, m->dtype, a));
    const void* res = nullptr; int rl = 0, rc = 0;
    if (c.has_res) { res = m->bufs[c.res.buf].act; rl = m->bufs[c.res.buf].ldc; rc = c.res.coff; }
    if (atomic) {   // the statistics left the convolution already reduced (integer sums): finalize + BN + SiLU in ONE pass, no bn_finalize launch
      YS_TRY(ys_bn_fin_apply_launch(st, m->dtype, y, M, c.cout, bn_acc_fin(m, c, M), c.act ? 1 : 0, res, rl, rc, ob.act, ob.ldc, c.out.coff));
      return YS_OK;
    }
    const int gm = ys_conv_grid_m(a, m->dtype);
    YS_TRY(ys_bn_finalize_launch(st, stat_partial, gm, c.cout, M, m->params + c.g_off, m->params + c.b_off, 1e-3f, 0.03f,
                                 m->state + c.rm_off, m->state + c.rv_off, m->state + c.nbt_off, chan_ptr(m, c, 0),
                                 chan_ptr(m, c, 1), chan_ptr(m, c, 2), chan_ptr(m, c, 3)));
    // fp8 mode: the next convolution of the schedule reads exactly this output and will run the fp8 blocked-GEMM kernel -> this pass also writes
    // the e4m3 image it consumes (the consumer's delayed scale) into the scratch and records its maximum
    bool q8_out = false;
    if (next && m->f8 && m->f8_sx_valid && next->f8_fwd && m->q8 && m->dtype == YS_BF16 && !next->dw && !next->ct) {
      ConvArgs q = fwd_args(m, *next, B);
      q.f8 = 1; q.w8 = m->wf8_all + next->wf_off; q.qscale = m->f8_scales + 4L * next->idx; q.deq = m->f8_scales + 4L * next->idx + 1;
      q8_out = ys_conv_wants_x8(q);
    }
    if (q8_out) {
      YS_TRY(ys_bn_act_apply_q8_launch(st, y, M, c.cout, chan_ptr(m, c, 0), chan_ptr(m, c, 1), c.act ? 1 : 0, res, rl, rc, ob.act, ob.ldc,
                                       c.out.coff, m->q8, m->f8_scales + 4L * next->idx, m->amax_act + (size_t)next->idx * YS_AMAX_WAYS));
      m->q8_fwd_ready = next->idx;
    } else {
      YS_TRY(ys_bn_act_apply_launch(st, m->dtype, y, M, c.cout, chan_ptr(m, c, 0), chan_ptr(m, c, 1), c.act ? 1 : 0, res, rl, rc,
                                    ob.act, ob.ldc, c.out.coff, amax_slot));
    }
  } else {
    a.y = view_ptr(m, ob.act, ob, c.out_rowoff);
    a.out_ldc = ob.ldc; a.out_coff = c.out.coff; a.out_bstride = ob.rows_per_b; a.vec_ok = vec ? 1 : 0;
    if (c.bn) {
      a.scale = chan_ptr(m, c, 0); a.shift = chan_ptr(m, c, 1); a.act = c.act ? 1 : 0;
      if (c.has_res) { const Buf& rb = m->bufs[c.res.buf]; a.res = rb.act; a.res_ldc = rb.ldc; a.res_coff = c.res.coff; }
    } else {
      a.shift = m->params + c.g_off;  // conv bias
    }
    YS_TRY(ys_conv_launch(st, m->dtype, a));
  }
  return YS_OK;
}

// ---- grouped forward of n independent Conv units / plain convolutions of one head stage (ConvL::group): one grouped convolution
// launch when the problems share a kernel variant (same channels and taps: the .1 / .2 tower layers of the three levels), else one
// launch each; then ONE BatchNorm finalize and ONE BN + SiLU apply launch for the whole stage.  Same arithmetic as run_conv_fwd.
int run_conv_fwd_group(ys_model* m, ConvL* const* cs, int n, int B) {
  hipStream_t st = m->ctx->stream;
  bool ok = n >= 2 && n <= YS_GROUP_MAX && !m->f8;
  for (int i = 0; i < n && ok; i++) {
    const ConvL& c = *cs[i];
    ok = !c.dw && !c.ct && !c.has_res && c.gstat_off >= 0 && c.bn == cs[0]->bn && c.act == cs[0]->act;
  }
  if (!ok) { for (int i = 0; i < n; i++) YS_TRY(run_conv_fwd(m, *cs[i], B)); return YS_OK; }
  ConvArgs a[YS_GROUP_MAX];
  int rows[YS_GROUP_MAX] = {0, 0, 0};
  const bool bn_train = cs[0]->bn && m->training;
  for (int i = 0; i < n; i++) {
    const ConvL& c = *cs[i];
    const Buf& ob = m->bufs[c.out.buf];
    a[i] = fwd_args(m, c, B);
    if (bn_train) {
      a[i].y = (char*)m->y_all + (size_t)c.y_off * m->es; a[i].out_ldc = c.cout; a[i].out_coff = 0; a[i].out_bstride = (long)c.Hout * c.Wout;
      a[i].vec_ok = (c.cout % 4 == 0);
      a[i].stats = m->stat_group + c.gstat_off;
    } else {
      a[i].y = view_ptr(m, ob.act, ob, c.out_rowoff);
      a[i].out_ldc = ob.ldc; a[i].out_coff = c.out.coff; a[i].out_bstride = ob.rows_per_b;
      a[i].vec_ok = ((ob.ldc % 4 == 0) && (c.out.coff % 4 == 0)) ? 1 : 0;
      if (c.bn) { a[i].scale = chan_ptr(m, c, 0); a[i].shift = chan_ptr(m, c, 1); a[i].act = c.act ? 1 : 0; }
      else a[i].shift = m->params + c.g_off;  // conv bias
    }
  }
  int rc = m->dtype == YS_BF16 ? ys_conv_p2_group_launch(st, a, n, nullptr, rows) : YS_ERR_UNSUPPORTED;
  if (rc == YS_ERR_UNSUPPORTED) {
    for (int i = 0; i < n; i++) { YS_TRY(ys_conv_launch(st, m->dtype, a[i])); rows[i] = ys_conv_grid_m(a[i], m->dtype); }
  } else if (rc != YS_OK) return rc;
  if (!bn_train) return YS_OK;
  BnFinProb fp[YS_GROUP_MAX];
  BnApplyProb ap[YS_GROUP_MAX];
  for (int i = 0; i < n; i++) {
    const ConvL& c = *cs[i];
    const Buf& ob = m->bufs[c.out.buf];
    const long M = a[i].M;
    fp[i] = BnFinProb{m->stat_group + c.gstat_off, rows[i], c.cout, (double)M, m->params + c.g_off, m->params + c.b_off, m->state + c.rm_off,
                      m->state + c.rv_off, m->state + c.nbt_off, chan_ptr(m, c, 0), chan_ptr(m, c, 1), chan_ptr(m, c, 2), chan_ptr(m, c, 3)};
    ap[i] = BnApplyProb{a[i].y, M, c.cout, chan_ptr(m, c, 0), chan_ptr(m, c, 1), nullptr, 0, 0, ob.act, ob.ldc, c.out.coff};
  }
  YS_TRY(ys_bn_finalize_group_launch(st, fp, n, 1e-3f, 0.03f));
  YS_TRY(ys_bn_act_apply_group_launch(st, m->dtype, ap, n, cs[0]->act ? 1 : 0));
  return YS_OK;
}

static int join_wgrad_stream(ys_model* m);
int forward_impl(ys_model* m, int B) {
  hipStream_t st = m->ctx->stream;
  YS_TRY(join_wgrad_stream(m));              // asynchronous segment ends: the previous step's weight-gradient kernels still read activations / dy slots
  YS_TRY(prep_weights(m));
  if (m->training && m->bn_atomic) YS_CHECK_HIP(hipMemsetAsync(m->stat_acc_all, 0, (size_t)m->n_stat_acc * 8, st));   // every unit's statistics accumulators: one clear per forward
  if (m->f8)   // delayed scaling: this pass quantises with the maxima the previous passes recorded (consumed and cleared here)
    YS_TRY(ys_f8_scales_launch(st, m->f8_convs, m->n_f8_convs, m->amax_w, m->amax_act, m->amax_dy, m->f8_scales));
  if (m->f8 && m->f8_bwd_done) m->f8_sg_valid = true;
  if (!m->training && m->eval_coeffs_dirty) {
    for (auto& c : m->convs)
      if (c.bn) YS_TRY(ys_bn_eval_coeffs_launch(st, c.cout, m->params + c.g_off, m->params + c.b_off, m->state + c.rm_off,
                                                m->state + c.rv_off, 1e-3f, chan_ptr(m, c, 0), chan_ptr(m, c, 1)));
    m->eval_coeffs_dirty = false;
  }
  m->q8_fwd_ready = -1;
  for (size_t oi = 0; oi < m->ops.size(); oi++) {
    const Op& op = m->ops[oi];
    const Buf& ib = m->bufs[op.in.buf];
    const Buf& ob = m->bufs[op.out.buf];
    if (op.type == OP_CONV && m->convs[op.conv].group >= 0) {
      // a head stage: the consecutive ops of one group run as grouped launches
      ConvL* gc[YS_GROUP_MAX]; int gn = 0;
      const int gid = m->convs[op.conv].group;
      while (oi + gn < m->ops.size() && gn < YS_GROUP_MAX && m->ops[oi + gn].type == OP_CONV && m->convs[m->ops[oi + gn].conv].group == gid) { gc[gn] = &m->convs[m->ops[oi + gn].conv]; gn++; }
      YS_TRY(run_conv_fwd_group(m, gc, gn, B));
      oi += gn - 1;
      continue;
    }
    if (op.type == OP_CONV) {
      const ConvL& cc = m->convs[op.conv];
      const ConvL* next = nullptr;               // the very next op, if it is a convolution reading exactly what this one writes
      if (oi + 1 < m->ops.size() && m->ops[oi + 1].type == OP_CONV) {
        const ConvL& nc = m->convs[m->ops[oi + 1].conv];
        if (nc.in.buf == cc.out.buf && nc.in.coff == cc.out.coff && nc.in.C == cc.out.C && nc.cin_pad == cc.cout && nc.cin == cc.cout &&
            nc.Hin == cc.Hout && nc.Win == cc.Wout) next = &nc;
      }
      YS_TRY(run_conv_fwd(m, cc, B, next));
    } else if (op.type == OP_MAXPOOL) {
      YS_TRY(ys_maxpool5_fwd_launch(st, m->dtype, ib.act, ib.ldc, op.in.coff, B, op.H, op.W, op.in.C, ob.act, ob.ldc,
                                    op.out.coff, m->training ? m->argmax + op.aux_off : nullptr));
    } else if (op.type == OP_UPSAMPLE) {
      YS_TRY(ys_upsample2x_fwd_launch(st, m->dtype, ib.act, ib.ldc, op.in.coff, B, op.H, op.W, op.in.C, ob.act, ob.ldc, op.out.coff));
    } else if (op.type == OP_ATTN) {
      YS_TRY(ys_attn_fwd_launch(st, m->dtype, ib.act, ib.ldc, B, op.H * op.W, op.heads, op.kd, op.hd, ob.act, ob.ldc, m->attn_ws + op.aux_off));
    } else if (op.type == OP_VCOPY) {
      YS_TRY(ys_attn_v_copy_launch(st, m->dtype, ib.act, ob.act, (long)B * op.H * op.W, ib.ldc, op.heads, op.kd, op.hd, ob.ldc, 0));
    } else if (op.type == OP_COPY) {
      YS_TRY(ys_copy_view_launch(st, m->dtype, ib.act, ib.ldc, op.in.coff, (long)B * op.H * op.W, op.in.C, ob.act, ob.ldc, op.out.coff, 0));
    }
  }
  if (m->f8) m->f8_sx_valid = true;        // every fp8 candidate has recorded an input maximum (bootstrap pass or its own kernel)
  if (!m->training && m->pd_buf >= 0) {
    YS_TRY(ys_detect_decode_launch(st, m->dtype, m->bufs[m->pd_buf].act, m->ld_pd, m->bufs[m->ps_buf].act, m->ld_ps, B, m->A,
                                   m->d.nc, m->d.reg_max, m->nl, m->lvl_off, m->lvl_w, m->lvl_stride, m->pred, 4 + m->d.nc + m->nm,
                                   m->xkind >= 2 ? m->bufs[m->mc_buf].act : nullptr, m->ld_mc, m->xkind, m->nm, m->kdim));
    if (m->segment)   // Segment._inference: cat(preds, mask_coefficient) (Head.cs:309-313), raw coefficients
      YS_TRY(ys_unpack_nchw_strided_launch(st, m->dtype, m->bufs[m->mc_buf].act, m->ld_mc, 0, B, m->nm, m->A, m->pred,
                                           (long)(4 + m->d.nc + m->nm) * m->A, (long)(4 + m->d.nc) * m->A));
  }
  YS_CHECK_HIP(hipGetLastError());
  return YS_OK;
}

// ------------------------------------------------------------------ backward
// returns 0 = first writer (overwrite), 1 = accumulate; -1 = inconsistent slice state
int grad_mode(ys_model* m, const View& v) {
  Buf& b = m->bufs[v.buf];
  int nw = 0;
  for (int c = v.coff; c < v.coff + v.C; c++) nw += b.gw[c] ? 1 : 0;
  if (nw != 0 && nw != v.C) return -1;
  for (int c = v.coff; c < v.coff + v.C; c++) b.gw[c] = 1;
  return nw ? 1 : 0;
}

static int flush_wgrads(ys_model* m, int B);
int run_convT_bwd(ys_model* m, const ConvL& c, int B) {
  hipStream_t st = m->ctx->stream;
  YS_TRY(flush_wgrads(m, B));            // queued weight gradients go first: this path drains the second stream and then uses the shared workspace
  if (m->overlap && m->st2_dirty) {      // this path uses the shared wgrad workspace on `st`: drain the weight-gradient stream first
    YS_CHECK_HIP(hipEventRecord(m->ev_join, m->st2));
    YS_CHECK_HIP(hipStreamWaitEvent(st, m->ev_join, 0));
    m->st2_dirty = false;
  }
  const Buf& ib = m->bufs[c.in.buf];
  const Buf& ob = m->bufs[c.out.buf];
  const long Mup = (long)B * c.Hout * c.Wout;
  YS_TRY(ys_colsum_launch(st, m->dtype, ob.grad, ob.ldc, c.out.coff, Mup, Mup, Mup, c.cout, m->stat_partial, m->grads + c.g_off));
  const int mode0 = grad_mode(m, c.in);
  if (mode0 < 0) { ys_set_error("backward: inconsistent gradient slice state at %s", c.name.c_str()); return YS_ERR_STATE; }
  for (int ph = 0; ph < 4; ph++) {
    const int dh = ph >> 1, dw = ph & 1;
    WgradArgs wa{};
    wa.x = ib.act; wa.dy = ob.grad; wa.partial = m->wg_partial;
    wa.B = B; wa.Hin = c.Hin; wa.Win = c.Win; wa.Cin = c.cin_pad; wa.Hout = c.Hin; wa.Wout = c.Win; wa.Cout = c.cout; wa.KH = wa.KW = 1;
    wa.stride = 1; wa.pad = 0; wa.in_ldc = ib.ldc; wa.in_coff = c.in.coff; wa.in_bstride = ib.rows_per_b;
    wa.dy_ldc = ob.ldc; wa.dy_coff = c.out.coff; wa.dy_bstride = ob.rows_per_b;
    wa.dy_rh = 4 * c.Win; wa.dy_rw = 2; wa.dy_r0 = (long)dh * 2 * c.Win + dw;
    wa.M = B * c.Hin * c.Win;
    const int splits = ys_wgrad_splits(wa, m->dtype);
    if ((long)splits * c.cout * c.cin_pad > m->n_wgp) { ys_set_error("wgrad workspace too small"); return YS_ERR_STATE; }
    YS_TRY(ys_wgrad_launch(st, m->dtype, wa, splits, c.cin, m->grads + c.w_off + (long)ph * c.cout * c.cin));
    // dx[h,w] (+)= W[dh,dw]^T dy[2h+dh, 2w+dw]: 1x1 gather with stride 2 and offsets (dh, dw)
    ConvArgs a{};
    a.x = ob.grad; a.w = (char*)m->wd_all + (size_t)(c.wd_off + (long)ph * c.cin * c.cout_ld) * m->es; a.y = ib.grad;
    a.B = B; a.Hin = c.Hout; a.Win = c.Wout; a.Cin = c.cout_ld; a.Hout = c.Hin; a.Wout = c.Win; a.Cout = c.cin; a.KH = a.KW = 1;
    a.SA = 2; a.PAD = -dh; a.pad_w_delta = dh - dw;
    a.in_ldc = ob.ldc; a.in_coff = c.out.coff; a.in_bstride = ob.rows_per_b;
    a.out_ldc = ib.ldc; a.out_coff = c.in.coff; a.out_bstride = ib.rows_per_b;
    a.vec_ok = (ib.ldc % 4 == 0 && c.in.coff % 4 == 0) ? 1 : 0;
    a.accumulate = (ph == 0) ? mode0 : 1;
    a.M = B * c.Hin * c.Win;
    YS_TRY(ys_conv_launch(st, m->dtype, a));
  }
  if (m->overlap) {                      // later weight-gradient launches must not overtake this layer's use of the workspace
    YS_CHECK_HIP(hipEventRecord(m->ev_dy[ys_model::DY_RING], st));
    YS_CHECK_HIP(hipStreamWaitEvent(m->st2, m->ev_dy[ys_model::DY_RING], 0));
  }
  return YS_OK;
}

// input-gradient convolution of layer c (gather form with flipped / transposed weights) reading dy as a [M][dy_ldc] view
static ConvArgs dgrad_args(ys_model* m, const ConvL& c, int B, const void* dy, int dy_ldc, int dy_coff, long dy_bstride) {
  const Buf& ib = m->bufs[c.in.buf];
  ConvArgs a{};
  a.x = dy; a.w = (char*)m->wd_all + (size_t)c.wd_off * m->es;
  a.y = ib.grad;
  a.B = B; a.Hin = c.Hout; a.Win = c.Wout; a.Cin = c.cout_ld; a.Hout = c.Hin; a.Wout = c.Win; a.Cout = c.cin_pad; a.KH = a.KW = c.k;
  a.SA = 1; a.DIVS = c.s == 2 ? 1 : 0; a.DIVM = c.s - 1; a.PAD = c.k - 1 - c.k / 2;
  a.in_ldc = dy_ldc; a.in_coff = dy_coff; a.in_bstride = dy_bstride;
  a.out_ldc = ib.ldc; a.out_coff = c.in.coff; a.out_bstride = ib.rows_per_b;
  a.vec_ok = (ib.ldc % 4 == 0 && c.in.coff % 4 == 0) ? 1 : 0;
  a.M = B * c.Hin * c.Win;
  return a;
}


// Fused BN-backward reduction, host side.  For every BN Conv unit L find, per output channel, the FIRST op in forward order that
// reads it: that op's backward is the last writer of the channel's gradient dz.  When all of L's channels are completed by the
// dgrad launches of plain convolutions (reading them as their input view, not as a residual) whose kernels carry the fused
// epilogue (ys_conv_bnred_rows), those launches produce L's sums and L's backward skips chan_reduce_kernel.
static ConvArgs dgrad_args(ys_model* m, const ConvL& c, int B, const void* dy, int dy_ldc, int dy_coff, long dy_bstride);
static ConvArgs dgrad_plan_args(ys_model* m, const ConvL& c, int B, bool f8) {
  const Buf& ob = m->bufs[c.out.buf];
  ConvArgs a = c.bn ? dgrad_args(m, c, B, m->dy_scratch, c.cout, 0, (long)c.Hout * c.Wout)
                    : dgrad_args(m, c, B, view_ptr(m, ob.grad, ob, c.out_rowoff), ob.ldc, c.out.coff, ob.rows_per_b);
  if (f8 && m->f8 && c.f8_bwd) {
    a.f8 = 2; a.w8 = m->wd8_all + c.wd_off; a.qscale = m->f8_scales + 4L * c.idx + 2; a.deq = m->f8_scales + 4L * c.idx + 3; a.q8 = m->q8;
    if (c.bn && m->q8 && c.cout_ld == c.cout && ys_conv_wants_x8(a)) a.x8 = m->q8;
  }
  return a;
}
int plan_bnred(ys_model* m, int B) {
  if (m->bnred_B == B) return YS_OK;
  m->bnred_B = B;
  for (auto& c : m->convs) { c.feeds.clear(); c.red_src.clear(); c.red_ok = false; c.red_seen = 0; }
  // fp8 mode: a producer is fused only when its consumers' dgrads carry the fused epilogue under BOTH routings (the scale-less
  // first step runs the bf16 kernels, later steps the fp8 ones): conv_gemm_kernel<e5m2> has the variant, conv_p2_kernel<F8> does not
  if (!m->bnred_on || m->dtype != YS_BF16) return YS_OK;
  const int epl = m->epl;
  // first reader per (buffer, channel): op index and kind (0 = input of a plain convolution, 1 = residual / unsupported reader)
  std::vector<std::vector<int>> fop(m->bufs.size()), fkind(m->bufs.size());
  for (size_t b = 0; b < m->bufs.size(); b++) { fop[b].assign(m->bufs[b].ldc, -1); fkind[b].assign(m->bufs[b].ldc, 1); }
  auto touch = [&](const View& v, int op, int kind) {
    for (int ch = v.coff; ch < v.coff + v.C; ch++) {
      if (fop[v.buf][ch] == -1) { fop[v.buf][ch] = op; fkind[v.buf][ch] = kind; }
      else if (fop[v.buf][ch] == op) fkind[v.buf][ch] = 1;      // read twice by the same op (input and residual): not fusable
    }
  };
  for (size_t oi = 0; oi < m->ops.size(); oi++) {
    const Op& op = m->ops[oi];
    if (op.type == OP_CONV) {
      const ConvL& x = m->convs[op.conv];
      touch(x.in, (int)oi, (x.dw || x.ct || x.first) ? 1 : 0);
      if (x.has_res) touch(x.res, (int)oi, 1);
    } else {
      touch(op.in, (int)oi, 1);
    }
  }
  // candidate feeds
  for (auto& l : m->convs) {
    if (!l.bn || l.ct || l.cout_ld != l.cout) continue;
    const Buf& ob = m->bufs[l.out.buf]; (void)ob;
    bool ok = true;
    std::vector<ConvL::RedSrc> srcs;
    std::vector<std::pair<int, ConvL::RedFeed>> add;           // (consumer conv, feed)
    int ch = l.out.coff;
    while (ok && ch < l.out.coff + l.out.C) {
      const int op = fop[l.out.buf][ch];
      if (op < 0 || fkind[l.out.buf][ch] != 0) { ok = false; break; }
      int e = ch;
      while (e < l.out.coff + l.out.C && fop[l.out.buf][e] == op && fkind[l.out.buf][e] == 0) e++;
      const ConvL& x = m->convs[m->ops[op].conv];
      ConvL::RedFeed f{};
      f.prod = l.idx; f.c0 = ch - x.in.coff; f.c1 = e - x.in.coff; f.yc0 = ch - l.out.coff;
      if (f.c0 % epl || f.c1 % epl || f.yc0 % epl || x.Hin != l.Hout || x.Win != l.Wout) { ok = false; break; }
      add.push_back({x.idx, f});
      ch = e;
    }
    if (!ok || add.empty() || (int)add.size() > YS_BNRED_MAXSEG) continue;
    for (auto& pr : add) m->convs[pr.first].feeds.push_back(pr.second);
    l.red_ok = true;                                           // provisional: consumers are checked below
  }
  // consumers: segment capacity and kernel support; a failing consumer disqualifies the producers it would have served
  bool changed = true;
  while (changed) {
    changed = false;
    for (auto& x : m->convs) {
      if (x.feeds.empty()) continue;
      int rows = 0;
      if ((int)x.feeds.size() <= YS_BNRED_MAXSEG) {
        ConvArgs a = dgrad_plan_args(m, x, B, false);
        rows = ys_conv_bnred_rows(a, m->dtype);
        if (rows > 0 && m->f8) { ConvArgs a8 = dgrad_plan_args(m, x, B, true); const int r8 = ys_conv_bnred_rows(a8, m->dtype); rows = r8 > 0 ? std::max(rows, r8) : 0; }
      }
      for (auto& f : x.feeds) f.rows_cap = rows;
      if (rows == 0) {
        for (auto& f : x.feeds) m->convs[f.prod].red_ok = false;
        x.feeds.clear();
        changed = true;
      }
    }
    for (auto& x : m->convs) {                                  // drop feeds of disqualified producers
      const size_t n0 = x.feeds.size();
      x.feeds.erase(std::remove_if(x.feeds.begin(), x.feeds.end(), [&](const ConvL::RedFeed& f) { return !m->convs[f.prod].red_ok; }), x.feeds.end());
      if (x.feeds.size() != n0) changed = true;
    }
  }
  // partial-row regions and the producers' source lists
  long off = 0;
  for (auto& x : m->convs)
    for (size_t k = 0; k < x.feeds.size(); k++) {
      ConvL::RedFeed& f = x.feeds[k];
      f.part_off = off; off += (long)f.rows_cap * 2 * m->convs[f.prod].cout;
      m->convs[f.prod].red_src.push_back(ConvL::RedSrc{x.idx, (int)k});
    }
  for (auto& l : m->convs) {
    std::sort(l.red_src.begin(), l.red_src.end(), [&](const ConvL::RedSrc& p, const ConvL::RedSrc& q) {
      return m->convs[p.cons].feeds[p.feed].yc0 < m->convs[q.cons].feeds[q.feed].yc0; });
    if (l.red_src.empty()) l.red_ok = false;
  }
  if (off > m->n_bnred) {
    if (m->bnred_part) { YS_CHECK_HIP(hipStreamSynchronize(m->ctx->stream)); dev_free_tracked(m, m->bnred_part); m->bnred_part = nullptr; }
    YS_TRY(dev_alloc(m, (void**)&m->bnred_part, (size_t)off * 4));
    m->n_bnred = off;
  }
  if (YS_OPT_INT("BNRED_LOG", 0) != 0) {        // plan report (tests/test_bnred.py reads the count)
    int nf = 0, nb = 0;
    for (auto& l : m->convs) { if (l.bn) nb++; if (l.red_ok) nf++; }
    fprintf(stderr, "[ys] fused BN-backward reduction: %d of %d BN conv units (B=%d, %.1f MB of partial rows)\n", nf, nb, B, off * 4.0 / 1e6);
    for (auto& l : m->convs) if (l.bn && !l.red_ok) fprintf(stderr, "[ys]   not fused: %s\n", l.name.c_str());
  }
  return YS_OK;
}

// weight gradient of one convolution on stream `sw` (split partials into the layer's own region when the reduction is deferred)
static int launch_wgrad(ys_model* m, ConvL& c, int B, const void* dy, int dy_ldc, int dy_coff, long dy_bstride, hipStream_t sw) {
  const Buf& ib = m->bufs[c.in.buf];
  const long M = (long)B * c.Hout * c.Wout;
  WgradArgs a{};
  a.x = ib.act; a.dy = dy; a.partial = m->wg_partial;
  a.B = B; a.Hin = c.Hin; a.Win = c.Win; a.Cin = c.cin_pad; a.Hout = c.Hout; a.Wout = c.Wout; a.Cout = c.cout;
  a.KH = a.KW = c.k; a.stride = c.s; a.pad = c.k / 2;
  a.in_ldc = ib.ldc; a.in_coff = c.in.coff; a.in_bstride = ib.rows_per_b;
  a.dy_ldc = dy_ldc; a.dy_coff = dy_coff; a.dy_bstride = dy_bstride; a.M = (int)M;
  int splits = ys_wgrad_splits(a, m->dtype);
  const bool defer = c.wgp_off >= 0;
  if (c.first && m->in_f32 && defer) {     // model.0: x is the fp32 image itself (conv_stem.hip); slabs in the generic split layout
    int used = 0;
    YS_TRY(ys_stem_wgrad_launch(sw, m->in_f32, B, c.Hin, c.Win, dy, dy_ldc, dy_coff, dy_bstride, c.cout, m->wg_partial + c.wgp_off, c.wgp_splits, &used));
    WgRedDesc& d = m->red_host[c.red_slot];
    d.partial = m->wg_partial + c.wgp_off; d.grad = m->grads + c.w_off; d.n = (long)c.cout * c.k * c.k * c.cin_pad; d.splits = used;
    d.cin_pad = c.cin_pad; d.cin_real = c.cin;
    return YS_OK;
  }
  if (defer) { a.partial = m->wg_partial + c.wgp_off; if (splits > c.wgp_splits) splits = c.wgp_splits; }   // own region (sized at max_batch)
  else if ((long)splits * c.cout * c.k * c.k * c.cin_pad > m->n_wgp) { ys_set_error("wgrad workspace too small"); return YS_ERR_STATE; }
  int used = 0;
  YS_TRY(ys_wgrad_launch(sw, m->dtype, a, splits, c.cin, m->grads + c.w_off, defer ? &used : nullptr));
  if (defer) {
    WgRedDesc& d = m->red_host[c.red_slot];
    d.partial = a.partial; d.grad = m->grads + c.w_off; d.n = (long)c.cout * c.k * c.k * c.cin_pad; d.splits = used;
    d.cin_pad = c.cin_pad; d.cin_real = c.cin;
    YS_REQUIRE((c.cin_pad & 3) == 0, "wgrad reduce: input channel pitch must be a multiple of 4");
  }
  return YS_OK;
}

// Hand the queued weight-gradient launches to the second stream behind ONE event (everything they read -- their units' dy, written by bn_bwd_apply launches
// already issued on the main stream -- is complete there once the event fires).  Without the second stream nothing is ever queued.
static int flush_wgrads(ys_model* m, int B) {
  if (m->pend_wg.empty()) return YS_OK;
  hipStream_t sw = m->ctx->stream;
  if (m->overlap) {
    hipEvent_t ev = m->ev_dy[m->ev_hand];
    m->ev_hand = (m->ev_hand + 1) % (ys_model::DY_RING + 1);
    YS_CHECK_HIP(hipEventRecord(ev, m->ctx->stream));
    YS_CHECK_HIP(hipStreamWaitEvent(m->st2, ev, 0));
    sw = m->st2;
    m->st2_dirty = true;
  }
  for (const auto& p : m->pend_wg) YS_TRY(launch_wgrad(m, m->convs[p.conv], B, p.dy, p.ldc, p.coff, p.bstride, sw));
  m->pend_wg.clear(); m->pend_mb = 0.0;
  return YS_OK;
}
// queue layer c's weight gradient; flush when the batch is full: 6 launches or 40 megabytes of (input + dy) tensors (
// the P1 / P2 / P3 layers, whose kernels run 50-150 us, go over one or two at a time -- the bubble is small next to them and the second stream should not start
// them late; the P4 / P5 layers, 15-30 us each, go over in fours to sixes)
static int queue_wgrad(ys_model* m, ConvL& c, int B, const void* dy, int ldc, int coff, long bstride) {
  if (!m->overlap) return launch_wgrad(m, c, B, dy, ldc, coff, bstride, m->ctx->stream);
  if (c.first && m->hold_stem) {               // everything queued so far goes now; the stem's own launch waits for the segment end
    YS_TRY(flush_wgrads(m, B));
    m->pend_wg.push_back(ys_model::PendWg{c.idx, dy, ldc, coff, bstride});
    return YS_OK;
  }
  m->pend_wg.push_back(ys_model::PendWg{c.idx, dy, ldc, coff, bstride});
  m->pend_mb += ((double)B * c.Hout * c.Wout * c.cout + (double)B * c.Hin * c.Win * c.cin_pad) * m->es * 1e-6;
  // (measured on config 2, same box, two rounds each: one per launch 8.71 / 8.71 ms, 4 / 24 MB 8.74 / 8.72, 6 / 40 MB 8.65 / 8.66, 12 / 80 MB 8.81 / 8.75, 8 / 200 MB 8.84 / 8.81,
  // everything in one batch per segment 8.93 / 8.87; round 5's ring + per-launch events 8.79 / 8.73)
  if (m->pend_wg.size() >= 6 || m->pend_mb >= 40.0) return flush_wgrads(m, B);
  return YS_OK;
}

// the BN-backward segments of the producers whose dz this dgrad launch completes (c.feeds) -> a.red[]; `rows` = partial rows it will write
static void attach_bnred_feeds(ys_model* m, ConvL& c, ConvArgs& a, int rows) {
  a.nred = (int)c.feeds.size(); a.red_row0 = 0;
  for (int k = 0; k < a.nred; k++) {
    ConvL::RedFeed& f = c.feeds[k];
    ConvL& l = m->convs[f.prod];
    BnRedSeg& sg = a.red[k];
    sg.y = (char*)m->y_all + (size_t)l.y_off * m->es; sg.scale = chan_ptr(m, l, 0); sg.shift = chan_ptr(m, l, 1);
    sg.part = m->bnred_part + f.part_off; sg.c0 = f.c0; sg.c1 = f.c1; sg.yc0 = f.yc0; sg.C = l.cout; sg.act = l.act ? 1 : 0;
    f.rows = rows;
    l.red_seen++;
  }
}

int run_conv_bwd(ys_model* m, ConvL& c, int B) {
  if (c.ct) return run_convT_bwd(m, c, B);
  hipStream_t st = m->ctx->stream;
  const Buf& ib = m->bufs[c.in.buf];
  const Buf& ob = m->bufs[c.out.buf];
  const long M = (long)B * c.Hout * c.Wout;
  const void* dy = nullptr; int dy_ldc = 0, dy_coff = 0; long dy_bstride = (long)c.Hout * c.Wout;
  bool dy_q8 = false;                       // m->q8 already holds the e5m2 image of dy (written by the BN backward pass)
  if (c.bn) {
    const void* y = (char*)m->y_all + (size_t)c.y_off * m->es;
    void* rg = nullptr; int rgl = 0, rgc = 0;
    if (c.has_res) {
      // d(residual input) += dz.  First contribution to that view: plain copy; later ones accumulate inside the reduce pass.
      Buf& rb = m->bufs[c.res.buf];
      const int rmode = grad_mode(m, c.res);
      if (rmode < 0) { ys_set_error("backward: inconsistent residual gradient state at %s", c.name.c_str()); return YS_ERR_STATE; }
      if (rmode == 0) {
        YS_TRY(ys_copy_view_launch(st, m->dtype, ob.grad, ob.ldc, c.out.coff, M, c.cout, rb.grad, rb.ldc, c.res.coff, 0));
      } else {
        rg = rb.grad; rgl = rb.ldc; rgc = c.res.coff;
      }
    }
    // sum(du), sum(du * y): from the epilogues of the dgrad launches that completed dz (fused BN-backward reduction), else a pass of its own
    const bool fused = c.red_ok && c.red_seen == (int)c.red_src.size() && !c.red_src.empty();
    c.red_seen = 0;
    void* rg_apply = nullptr;                    // the shortcut's residual-gradient accumulation rides on the reduction pass; without one, on the apply pass
    if (fused) {
      FinSrc src{};
      src.n = (int)c.red_src.size();
      for (int k = 0; k < src.n; k++) {
        const ConvL::RedFeed& f = m->convs[c.red_src[k].cons].feeds[c.red_src[k].feed];
        src.p[k] = m->bnred_part + f.part_off; src.nblk[k] = f.rows; src.c1[k] = f.yc0 + (f.c1 - f.c0);
      }
      YS_TRY(ys_bn_bwd_finalize_src_launch(st, src, c.cout, M, m->grads + c.g_off, m->grads + c.b_off, chan_ptr(m, c, 4), chan_ptr(m, c, 5),
                                           chan_ptr(m, c, 0), chan_ptr(m, c, 2), chan_ptr(m, c, 3)));
      rg_apply = rg;
    } else {
      int nblk = 0;
      YS_TRY(ys_bn_bwd_reduce_launch(st, m->dtype, ob.grad, ob.ldc, c.out.coff, y, M, c.cout, chan_ptr(m, c, 0), chan_ptr(m, c, 1),
                                     chan_ptr(m, c, 2), chan_ptr(m, c, 3), c.act ? 1 : 0, rg, rgl, rgc, m->stat_partial, &nblk));
      YS_TRY(ys_bn_bwd_finalize_launch(st, m->stat_partial, nblk, c.cout, M, m->grads + c.g_off, m->grads + c.b_off,
                                       chan_ptr(m, c, 4), chan_ptr(m, c, 5), chan_ptr(m, c, 0), chan_ptr(m, c, 2), chan_ptr(m, c, 3)));
    }
    void* dyb = m->dy_scratch;
    if (m->overlap && !c.dw && c.dy_own) dyb = c.dy_own;   // the unit's own buffer: its weight gradient reads it later, from the second stream (queue_wgrad)
    // fp8 mode: when this layer's dgrad will run the fp8 blocked-GEMM kernel, the same pass writes the e5m2 image of dy it consumes
    // (and records amax(|dy|)) -- no separate quantisation pass over dy
    if (m->f8 && c.f8_bwd && m->f8_sg_valid && !c.first && !c.dw && m->q8 && m->dtype == YS_BF16 && c.cout_ld == c.cout) {
      ConvArgs q = dgrad_args(m, c, B, dyb, c.cout, 0, (long)c.Hout * c.Wout);
      q.f8 = 2; q.w8 = m->wd8_all + c.wd_off; q.qscale = m->f8_scales + 4L * c.idx + 2; q.deq = m->f8_scales + 4L * c.idx + 3;
      dy_q8 = ys_conv_wants_x8(q);
    }
    if (dy_q8) {
      YS_TRY(ys_bn_bwd_apply_q8_launch(st, ob.grad, ob.ldc, c.out.coff, y, M, c.cout, chan_ptr(m, c, 0), chan_ptr(m, c, 1), chan_ptr(m, c, 4),
                                       chan_ptr(m, c, 5), c.act ? 1 : 0, dyb, m->q8, m->f8_scales + 4L * c.idx + 2,
                                       m->amax_dy + (size_t)c.idx * YS_AMAX_WAYS, rg_apply, rgl, rgc));
    } else {
      YS_TRY(ys_bn_bwd_apply_launch(st, m->dtype, ob.grad, ob.ldc, c.out.coff, y, M, c.cout, chan_ptr(m, c, 0), chan_ptr(m, c, 1),
                                    chan_ptr(m, c, 4), chan_ptr(m, c, 5), c.act ? 1 : 0, dyb, nullptr, rg_apply, rgl, rgc));
    }
    dy = dyb; dy_ldc = c.cout; dy_coff = 0;
  } else {
    // plain Conv2d with bias (head outputs): dy is the loss gradient itself
    dy = view_ptr(m, ob.grad, ob, c.out_rowoff); dy_ldc = ob.ldc; dy_coff = c.out.coff; dy_bstride = ob.rows_per_b;
    YS_TRY(ys_colsum_launch(st, m->dtype, dy, dy_ldc, dy_coff, M, (long)c.Hout * c.Wout, dy_bstride, c.cout, m->stat_partial,
                            m->grads + c.g_off));
  }
  if (c.dw) {
    YS_TRY(ys_dwconv_wgrad_launch(st, m->dtype, ib.act, ib.ldc, c.in.coff, dy, B, c.Hin, c.Win, c.cout, m->stat_partial, m->grads + c.w_off));
    const int mode = grad_mode(m, c.in);
    if (mode < 0) { ys_set_error("backward: inconsistent gradient slice state at %s", c.name.c_str()); return YS_ERR_STATE; }
    YS_TRY(ys_dwconv_launch(st, m->dtype, 1, dy, c.cout, 0, B, c.Hin, c.Win, c.cout, m->params + c.w_off, ib.grad, ib.ldc, c.in.coff, mode));
    return YS_OK;
  }
  // ---- wgrad: queued for the second stream (issued at once without it)
  if (c.bn && m->overlap && dy != c.dy_own) { ys_set_error("backward: %s has no dy buffer of its own", c.name.c_str()); return YS_ERR_STATE; }
  YS_TRY(queue_wgrad(m, c, B, dy, dy_ldc, dy_coff, dy_bstride));
  // ---- dgrad (gather form with flipped/transposed weights)
  if (!c.first) {
    const int mode = grad_mode(m, c.in);
    if (mode < 0) { ys_set_error("backward: inconsistent gradient slice state at %s", c.name.c_str()); return YS_ERR_STATE; }
    ConvArgs a = dgrad_args(m, c, B, dy, dy_ldc, dy_coff, dy_bstride);
    a.accumulate = mode;
    if (m->f8 && c.f8_bwd) {
      unsigned* slots = m->amax_dy + (size_t)c.idx * YS_AMAX_WAYS;
      if (m->f8_sg_valid) {   // dgrad with the gradient quantised to e5m2 and the e4m3 dgrad weights; records amax(|dy|) itself
        a.f8 = 2; a.w8 = m->wd8_all + c.wd_off; a.qscale = m->f8_scales + 4L * c.idx + 2; a.deq = m->f8_scales + 4L * c.idx + 3; a.amax = slots;
        a.q8 = m->q8;
        if (dy_q8) { a.x8 = m->q8; a.amax = nullptr; }     // image and maximum already produced by the BN backward pass
      } else {
        YS_TRY(ys_f8_view_amax_launch(st, dy, M, c.cout, dy_ldc, dy_coff, slots));
      }
    }
    if (c.bn && c.cout_ld != c.cout) { ys_set_error("backward: padded BN conv unsupported"); return YS_ERR_UNSUPPORTED; }
    if (!c.feeds.empty()) {
      // this launch completes dz of the producers in c.feeds: its epilogue takes their BN-backward sums (BnRedSeg)
      const int rows = ys_conv_bnred_rows(a, m->dtype);
      bool fits = rows > 0;
      for (auto& f : c.feeds) fits = fits && rows <= f.rows_cap;
      if (fits) attach_bnred_feeds(m, c, a, rows);
    }
    YS_TRY(ys_conv_launch(st, m->dtype, a));
  }
  return YS_OK;
}

// ---- grouped backward of one head stage (mirror of run_conv_fwd_group): BN backward finalize + apply as one launch each, the weight
// gradients handed to the weight-gradient stream behind ONE event, the input gradients as one grouped dgrad launch when the problems
// share a kernel variant.  Falls back to run_conv_bwd per unit whenever a unit needs something the grouped form does not carry.
int run_conv_bwd_group(ys_model* m, ConvL* const* cs, int n, int B) {
  hipStream_t st = m->ctx->stream;
  bool ok = n >= 2 && n <= YS_GROUP_MAX && !m->f8;
  for (int i = 0; i < n && ok; i++) {
    const ConvL& c = *cs[i];
    ok = !c.dw && !c.ct && !c.has_res && !c.first && c.gstat_off >= 0 && c.bn == cs[0]->bn && c.act == cs[0]->act && (!c.bn || (c.dy_own && c.cout_ld == c.cout));
    if (ok && c.bn) ok = c.red_ok && c.red_seen == (int)c.red_src.size() && !c.red_src.empty();   // sums already produced by the consumers' dgrads
  }
  if (!ok) { for (int i = 0; i < n; i++) YS_TRY(run_conv_bwd(m, *cs[i], B)); return YS_OK; }
  const void* dy[YS_GROUP_MAX]; int dy_ldc[YS_GROUP_MAX], dy_coff[YS_GROUP_MAX]; long dy_bs[YS_GROUP_MAX];
  if (cs[0]->bn) {
    ChanFinProb fp[YS_GROUP_MAX];
    BnBwdProb bp[YS_GROUP_MAX];
    for (int i = 0; i < n; i++) {
      ConvL& c = *cs[i];
      const Buf& ob = m->bufs[c.out.buf];
      const long M = (long)B * c.Hout * c.Wout;
      FinSrc src{};
      src.n = (int)c.red_src.size();
      for (int k = 0; k < src.n; k++) {
        const ConvL::RedFeed& f = m->convs[c.red_src[k].cons].feeds[c.red_src[k].feed];
        src.p[k] = m->bnred_part + f.part_off; src.nblk[k] = f.rows; src.c1[k] = f.yc0 + (f.c1 - f.c0);
      }
      c.red_seen = 0;
      fp[i] = ChanFinProb{src, c.cout, (double)M, m->grads + c.g_off, m->grads + c.b_off, chan_ptr(m, c, 4), chan_ptr(m, c, 5), chan_ptr(m, c, 0), chan_ptr(m, c, 2), chan_ptr(m, c, 3)};
      bp[i] = BnBwdProb{ob.grad, ob.ldc, c.out.coff, (char*)m->y_all + (size_t)c.y_off * m->es, M, c.cout, chan_ptr(m, c, 0), chan_ptr(m, c, 1), chan_ptr(m, c, 4), chan_ptr(m, c, 5), c.dy_own, nullptr, 0, 0};
      dy[i] = c.dy_own; dy_ldc[i] = c.cout; dy_coff[i] = 0; dy_bs[i] = (long)c.Hout * c.Wout;
    }
    YS_TRY(ys_bn_bwd_finalize_group_launch(st, fp, n));
    YS_TRY(ys_bn_bwd_apply_group_launch(st, m->dtype, bp, n, cs[0]->act ? 1 : 0));
  } else {
    for (int i = 0; i < n; i++) {
      ConvL& c = *cs[i];
      const Buf& ob = m->bufs[c.out.buf];
      const long M = (long)B * c.Hout * c.Wout;
      dy[i] = view_ptr(m, ob.grad, ob, c.out_rowoff); dy_ldc[i] = ob.ldc; dy_coff[i] = c.out.coff; dy_bs[i] = ob.rows_per_b;
      YS_TRY(ys_colsum_launch(st, m->dtype, dy[i], dy_ldc[i], dy_coff[i], M, (long)c.Hout * c.Wout, dy_bs[i], c.cout, m->stat_partial, m->grads + c.g_off));
    }
  }
  // ---- weight gradients: every dy of the stage is complete on `st` here -- queued, and handed over together
  for (int i = 0; i < n; i++) YS_TRY(queue_wgrad(m, *cs[i], B, dy[i], dy_ldc[i], dy_coff[i], dy_bs[i]));
  YS_TRY(flush_wgrads(m, B));
  // ---- input gradients
  ConvArgs a[YS_GROUP_MAX];
  int cap[YS_GROUP_MAX] = {0, 0, 0}, rows[YS_GROUP_MAX] = {0, 0, 0};
  for (int i = 0; i < n; i++) {
    ConvL& c = *cs[i];
    const int mode = grad_mode(m, c.in);
    if (mode < 0) { ys_set_error("backward: inconsistent gradient slice state at %s", c.name.c_str()); return YS_ERR_STATE; }
    a[i] = dgrad_args(m, c, B, dy[i], dy_ldc[i], dy_coff[i], dy_bs[i]);
    a[i].accumulate = mode;
    for (auto& f : c.feeds) cap[i] = cap[i] ? std::min(cap[i], f.rows_cap) : f.rows_cap;
    if (!c.feeds.empty()) { a[i].nred = (int)c.feeds.size(); }   // (segments are attached once the row counts are known)
  }
  // a grouped launch needs the segment tables before it starts and its row counts are only known after planning: plan first (dry run)
  int rc = m->dtype == YS_BF16 ? ys_conv_p2_group_launch(st, a, n, cap, rows, true) : YS_ERR_UNSUPPORTED;
  if (rc == YS_OK) {
    for (int i = 0; i < n; i++) { a[i].nred = 0; if (!cs[i]->feeds.empty()) attach_bnred_feeds(m, *cs[i], a[i], rows[i]); }
    YS_TRY(ys_conv_p2_group_launch(st, a, n, cap, rows));
  } else if (rc == YS_ERR_UNSUPPORTED) {
    for (int i = 0; i < n; i++) {
      ConvL& c = *cs[i];
      a[i].nred = 0;
      if (!c.feeds.empty()) {
        const int r = ys_conv_bnred_rows(a[i], m->dtype);
        bool fits = r > 0;
        for (auto& f : c.feeds) fits = fits && r <= f.rows_cap;
        if (fits) attach_bnred_feeds(m, c, a[i], r);
      }
      YS_TRY(ys_conv_launch(st, m->dtype, a[i]));
    }
  } else return rc;
  return YS_OK;
}

// the weight-gradient stream has work the main stream has not waited for: order the main stream behind it
static int join_wgrad_stream(ys_model* m) {
  if (m->st2 && m->st2_dirty) {
    YS_CHECK_HIP(hipEventRecord(m->ev_join, m->st2));
    YS_CHECK_HIP(hipStreamWaitEvent(m->ctx->stream, m->ev_join, 0));
    m->st2_dirty = false;
  }
  return YS_OK;
}

// async_end: leave the weight-gradient stream unjoined (its split reduction runs there too) and record the segment's completion events
int backward_range(ys_model* m, int seg_lo, int seg_hi, bool async_end = false, bool seg_events = true) {
  hipStream_t st = m->ctx->stream;
  const int B = m->B;
  YS_TRY(plan_bnred(m, B));
  for (int i = (int)m->ops.size() - 1; i >= 0; i--) {
    const Op& op = m->ops[i];
    if (op.seg < seg_lo || op.seg > seg_hi) continue;
    if (op.type == OP_CONV && m->convs[op.conv].group >= 0) {
      ConvL* gc[YS_GROUP_MAX]; int gn = 0;
      const int gid = m->convs[op.conv].group;
      int j = i;
      while (j >= 0 && gn < YS_GROUP_MAX && m->ops[j].type == OP_CONV && m->convs[m->ops[j].conv].group == gid) { gn++; j--; }
      for (int k = 0; k < gn; k++) gc[k] = &m->convs[m->ops[i - gn + 1 + k].conv];     // forward (level) order
      YS_TRY(run_conv_bwd_group(m, gc, gn, B));
      i -= gn - 1;
    } else if (op.type == OP_CONV) {
      YS_TRY(run_conv_bwd(m, m->convs[op.conv], B));
    } else if (op.type == OP_ATTN) {
      // d(qkv) = [dq | dk | dv]; the v part already holds the gradient that arrived through pe(v) (OP_VCOPY backward)
      const Buf& ib = m->bufs[op.in.buf];
      const Buf& ob = m->bufs[op.out.buf];
      const long nn = (long)B * op.heads * (long)(op.H * op.W) * (op.H * op.W);
      YS_TRY(ys_attn_bwd_launch(st, m->dtype, ib.act, ib.ldc, B, op.H * op.W, op.heads, op.kd, op.hd, ob.grad, ob.ldc,
                                m->attn_ws + op.aux_off, m->attn_ws + op.aux_off + nn, ib.grad));
      Buf& qb = m->bufs[op.in.buf];
      std::fill(qb.gw.begin(), qb.gw.end(), 1);
    } else if (op.type == OP_VCOPY) {
      const Buf& ib = m->bufs[op.in.buf];
      const Buf& ob = m->bufs[op.out.buf];
      YS_TRY(ys_attn_v_copy_launch(st, m->dtype, ob.grad, ib.grad, (long)B * op.H * op.W, ib.ldc, op.heads, op.kd, op.hd, ob.ldc, 1));
    } else if (op.type == OP_COPY) {
      const Buf& ib = m->bufs[op.in.buf];
      const Buf& ob = m->bufs[op.out.buf];
      const int mode = grad_mode(m, op.in);
      if (mode < 0) { ys_set_error("backward: inconsistent gradient slice state at op %d", i); return YS_ERR_STATE; }
      YS_TRY(ys_copy_view_launch(st, m->dtype, ob.grad, ob.ldc, op.out.coff, (long)B * op.H * op.W, op.in.C, ib.grad, ib.ldc, op.in.coff, mode));
    } else {
      const Buf& ib = m->bufs[op.in.buf];
      const Buf& ob = m->bufs[op.out.buf];
      const int mode = grad_mode(m, op.in);
      if (mode < 0) { ys_set_error("backward: inconsistent gradient slice state at op %d", i); return YS_ERR_STATE; }
      if (op.type == OP_MAXPOOL)
        YS_TRY(ys_maxpool5_bwd_launch(st, m->dtype, ob.grad, ob.ldc, op.out.coff, B, op.H, op.W, op.in.C, m->argmax + op.aux_off,
                                      ib.grad, ib.ldc, op.in.coff, mode));
      else
        YS_TRY(ys_upsample2x_bwd_launch(st, m->dtype, ob.grad, ob.ldc, op.out.coff, B, op.H, op.W, op.in.C, ib.grad, ib.ldc,
                                        op.in.coff, mode));
    }
  }
  // the segment's gradients are complete only when the weight-gradient stream has drained: the main stream waits for it here, unless the
  // caller asked for an asynchronous end -- then the split reduction below goes to that stream as well, nothing on the main stream waits,
  // and whoever consumes the segment's gradients (the all-reduce) waits on the two events recorded at the end
  // split reduction of weight gradients [lo, hi) of the descriptor table in one launch on stream sr (the partial slabs sit in per-layer regions)
  auto reduce_range = [&](int lo, int hi, hipStream_t sr) -> int {
    if (hi <= lo) return YS_OK;
    long blk = 0;
    for (int i = lo; i < hi; i++) { m->red_host[i].blk0 = blk; blk += (m->red_host[i].n + YS_WGRED_OUT_PER_BLOCK - 1) / YS_WGRED_OUT_PER_BLOCK; }
    if (m->red_uploaded.size() != m->red_host.size()) m->red_uploaded.assign(m->red_host.size(), WgRedDesc{});
    if (memcmp(&m->red_uploaded[lo], &m->red_host[lo], (size_t)(hi - lo) * sizeof(WgRedDesc)) != 0) {   // first step / batch size / launch partition changed
      YS_CHECK_HIP(hipMemcpyAsync(m->red_dev + lo, &m->red_host[lo], (size_t)(hi - lo) * sizeof(WgRedDesc), hipMemcpyHostToDevice, st));
      YS_CHECK_HIP(hipStreamSynchronize(st));
      memcpy(&m->red_uploaded[lo], &m->red_host[lo], (size_t)(hi - lo) * sizeof(WgRedDesc));
    }
    return ys_wgrad_reduce_batched_launch(sr, m->red_dev + lo, hi - lo, blk);
  };
  const int rlo = m->red_first[seg_lo], rhi = m->red_first[seg_hi + 1];
  // The step's LAST dependency chain is dgrad(model.1) -> BN backward of model.0 -> stem weight gradient -> split reduction -> AdamW (round-6 trace: 200 us with the
  // main stream idle).  In the one-call backward the stem's weight gradient is therefore handed over on its own, AFTER the reduction of everything else in its
  // segment has been queued on the second stream: that reduction (48 us) then runs while the main stream is still in the BatchNorm backward of model.0, and only
  // the stem's own slabs are reduced behind its weight gradient.
  bool stem_split = false;
  ys_model::PendWg stem_job{};
  if (async_end && !seg_events && m->overlap && seg_hi == ys_model::NSEG - 1 && !m->pend_wg.empty() && m->convs[m->pend_wg.back().conv].first &&
      m->convs[m->pend_wg.back().conv].red_slot == rlo && m->defer_wgred && rhi - rlo > 1) {
    stem_split = true; stem_job = m->pend_wg.back(); m->pend_wg.pop_back();
  }
  YS_TRY(flush_wgrads(m, B));
  const bool on_st2 = async_end && m->overlap && (m->st2_dirty || stem_split);
  hipStream_t sr = on_st2 ? m->st2 : st;
  if (!on_st2) YS_TRY(join_wgrad_stream(m));
  if (m->f8 && seg_hi == ys_model::NSEG - 1) m->f8_bwd_done = true;     // every gradient maximum of the step is recorded (last segment = stem)
  if (stem_split) {
    YS_TRY(reduce_range(rlo + 1, rhi, sr));
    m->pend_wg.push_back(stem_job);
    YS_TRY(flush_wgrads(m, B));
    YS_TRY(reduce_range(rlo, rlo + 1, sr));
  } else {
    YS_TRY(reduce_range(rlo, rhi, sr));
  }
  if (async_end && seg_events) {
    for (int sgi = seg_lo; sgi <= seg_hi; sgi++) {
      m->seg_on_st2[sgi] = on_st2;
      YS_CHECK_HIP(hipEventRecord(m->ev_seg_m[sgi], st));
      if (on_st2) YS_CHECK_HIP(hipEventRecord(m->ev_seg_w[sgi], m->st2));
    }
  }
  YS_CHECK_HIP(hipGetLastError());
  return YS_OK;
}

void reset_grad_state(ys_model* m) {
  for (auto& b : m->bufs) std::fill(b.gw.begin(), b.gw.end(), 0);
  for (auto& c : m->convs) c.red_seen = 0;
  if (m->is_block) { std::fill(m->bufs[m->blk_out].gw.begin(), m->bufs[m->blk_out].gw.end(), 1); return; }   // the caller's dy
  // the loss wrote the head gradients
  std::fill(m->bufs[m->pd_buf].gw.begin(), m->bufs[m->pd_buf].gw.end(), 1);
  std::fill(m->bufs[m->ps_buf].gw.begin(), m->bufs[m->ps_buf].gw.end(), 1);
  if (m->segment) {
    std::fill(m->bufs[m->mc_buf].gw.begin(), m->bufs[m->mc_buf].gw.end(), 1);
    std::fill(m->bufs[m->pr_buf].gw.begin(), m->bufs[m->pr_buf].gw.end(), 1);
  }
  if (m->xkind >= 2) std::fill(m->bufs[m->mc_buf].gw.begin(), m->bufs[m->mc_buf].gw.end(), 1);   // angle-logit / keypoint gradients
}

TensorRec* find_tensor(ys_model* m, const char* name) {
  for (auto& t : m->tensors) if (t.name == name) return &t;
  return nullptr;
}

// splitmix64 -> uniform(-bound, bound)
struct Rng {
  uint64_t s;
  uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  float uni(float b) { return ((float)((next() >> 40) & 0xFFFFFF) / 16777216.0f * 2.0f - 1.0f) * b; }
};

}  // namespace

extern "C" {

int ys_model_create(ys_ctx* ctx, const ys_model_desc* desc, ys_model** out) {
  YS_REQUIRE(ctx && desc && out, "ys_model_create: null argument");
  YS_REQUIRE(desc->dtype == YS_F32 || desc->dtype == YS_BF16 || desc->dtype == YS_FP8, "ys_model_create: dtype %d unsupported", desc->dtype);
  if ((desc->family != YS_YOLOV8 && desc->family != YS_YOLOV11) || desc->task < YS_DETECT || desc->task > YS_POSE) {
    ys_set_error("ys_model_create: YOLOv8 / YOLOv11 detect, segment, obb and pose are built (family %d task %d)", desc->family, desc->task);
    return YS_ERR_UNSUPPORTED;
  }
  YS_REQUIRE(desc->size >= 0 && desc->size <= 4, "ys_model_create: size %d out of range", desc->size);
  YS_REQUIRE(desc->nc > 0 && desc->reg_max > 1 && desc->reg_max <= 32, "ys_model_create: nc=%d reg_max=%d", desc->nc, desc->reg_max);
  YS_REQUIRE(desc->height > 0 && desc->width > 0 && desc->height % 32 == 0 && desc->width % 32 == 0,
             "ys_model_create: image size %dx%d must be a positive multiple of 32", desc->height, desc->width);
  YS_REQUIRE(desc->max_batch > 0, "ys_model_create: max_batch %d", desc->max_batch);
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  ys_model* m = new ys_model();
  // YS_FP8 = the bf16 engine (storage, BN, losses, weight gradients, optimizer) with fp8 MFMA forward / dgrad convolutions
  m->f8 = desc->dtype == YS_FP8;
  const int store = m->f8 ? YS_BF16 : desc->dtype;
  m->ctx = ctx; m->d = *desc; m->dtype = store; m->epl = store == YS_BF16 ? 8 : 4; m->es = store == YS_BF16 ? 2 : 4;
  m->maxB = desc->max_batch;
  for (int j = 0; j < 64; j++) m->dfl_w[j] = (float)j;
  int st = desc->family == YS_YOLOV11 ? build_v11_detect(m) : build_v8_detect(m);
  if (st == YS_OK) st = layout_params(m);
  if (st == YS_OK) st = allocate(m);
  if (st != YS_OK) { ys_model_destroy(m); return st; }
  st = ys_model_init_weights(m, 0);
  if (st != YS_OK) { ys_model_destroy(m); return st; }
  *out = m;
  return YS_OK;
}

int ys_model_destroy(ys_model* m) {
  if (!m) return YS_OK;
  hipSetDevice(m->ctx->device);
  hipStreamSynchronize(m->ctx->stream);
  if (m->st2) {
    hipStreamSynchronize(m->st2);
    for (int k = 0; k <= ys_model::DY_RING; k++) if (m->ev_dy[k]) hipEventDestroy(m->ev_dy[k]);
    if (m->ev_join) hipEventDestroy(m->ev_join);
    hipStreamDestroy(m->st2);
  }
  for (int k = 0; k < ys_model::NSEG; k++) { if (m->ev_seg_m[k]) hipEventDestroy(m->ev_seg_m[k]); if (m->ev_seg_w[k]) hipEventDestroy(m->ev_seg_w[k]); }
  for (void* p : m->allocs) hipFree(p);
  delete m;
  return YS_OK;
}

}  // extern "C"
ys_ctx* ys_model_ctx(ys_model* m) { return m->ctx; }   // for dist.hip
extern "C" {
int ys_model_num_tensors(ys_model* m) { return m ? (int)m->tensors.size() : 0; }
int ys_model_num_anchors(ys_model* m) { return m ? m->A : 0; }
int64_t ys_model_num_params(ys_model* m) { return m ? (int64_t)m->n_params_real : 0; }

int ys_model_tensor_info(ys_model* m, int index, char* name, int name_cap, int32_t* ndim, int64_t shape[4], int32_t* is_param) {
  YS_REQUIRE(m && index >= 0 && index < (int)m->tensors.size(), "ys_model_tensor_info: index %d out of range", index);
  const TensorRec& t = m->tensors[index];
  if (name && name_cap > 0) { strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (ndim) *ndim = t.ndim;
  if (shape) for (int i = 0; i < 4; i++) shape[i] = i < t.ndim ? t.shape[i] : 1;
  if (is_param) *is_param = t.is_param ? 1 : 0;
  return YS_OK;
}

static int tensor_io(ys_model* m, const char* name, float* host, size_t count, int what /*0 set,1 get,2 get grad*/) {
  YS_REQUIRE(m && name && host, "tensor io: null argument");
  if (what == 2) YS_TRY(join_wgrad_stream(m));   // asynchronous segment ends: gradients are complete once the weight-gradient stream has drained
  TensorRec* t = find_tensor(m, name);
  YS_REQUIRE(t != nullptr, "unknown tensor '%s'", name);
  YS_REQUIRE((long)count == t->count, "tensor '%s' has %ld elements, caller passed %zu", name, t->count, count);
  hipStream_t st = m->ctx->stream;
  if (t->kind == 3) {
    YS_REQUIRE(what != 2, "'%s' has no gradient (DFL is only used in eval, Head.cs:221)", name);
    if (what == 0) memcpy(m->dfl_w, host, count * 4); else memcpy(host, m->dfl_w, count * 4);
    return YS_OK;
  }
  float* dev = t->kind == 2 ? m->state + t->off : (what == 2 ? m->grads + t->off : m->params + t->off);
  YS_REQUIRE(!(what == 2 && t->kind == 2), "'%s' is a buffer and has no gradient", name);
  if (t->kind == 5) {   // ConvTranspose2d weight: edge [Cin][Cout][2][2] <-> internal [phase][Cout][Cin]
    const ConvL& c = m->convs[t->conv];
    std::vector<float> tmp(count);
    if (what == 0) {
      for (int ci = 0; ci < c.cin; ci++) for (int co = 0; co < c.cout; co++) for (int ph = 0; ph < 4; ph++)
        tmp[((size_t)ph * c.cout + co) * c.cin + ci] = host[((size_t)ci * c.cout + co) * 4 + ph];
      YS_CHECK_HIP(hipMemcpyAsync(dev, tmp.data(), count * 4, hipMemcpyHostToDevice, st));
      YS_CHECK_HIP(hipStreamSynchronize(st));
      m->weights_dirty = true;
    } else {
      YS_CHECK_HIP(hipMemcpyAsync(tmp.data(), dev, count * 4, hipMemcpyDeviceToHost, st));
      YS_CHECK_HIP(hipStreamSynchronize(st));
      for (int ci = 0; ci < c.cin; ci++) for (int co = 0; co < c.cout; co++) for (int ph = 0; ph < 4; ph++)
        host[((size_t)ci * c.cout + co) * 4 + ph] = tmp[((size_t)ph * c.cout + co) * c.cin + ci];
    }
    return YS_OK;
  }
  if (t->kind == 4) {
    const ConvL& c = m->convs[t->conv];
    std::vector<float> tmp(count);
    if (what == 0) {
      for (int ch = 0; ch < c.cout; ch++) for (int tp = 0; tp < 9; tp++) tmp[(size_t)tp * c.cout + ch] = host[(size_t)ch * 9 + tp];
      YS_CHECK_HIP(hipMemcpyAsync(dev, tmp.data(), count * 4, hipMemcpyHostToDevice, st));
      YS_CHECK_HIP(hipStreamSynchronize(st));
    } else {
      YS_CHECK_HIP(hipMemcpyAsync(tmp.data(), dev, count * 4, hipMemcpyDeviceToHost, st));
      YS_CHECK_HIP(hipStreamSynchronize(st));
      for (int ch = 0; ch < c.cout; ch++) for (int tp = 0; tp < 9; tp++) host[(size_t)ch * 9 + tp] = tmp[(size_t)tp * c.cout + ch];
    }
    return YS_OK;
  }
  if (t->kind == 0) {
    // OIHW at the edge <-> [Cout][taps][Cin] inside
    const ConvL& c = m->convs[t->conv];
    const int taps = c.k * c.k;
    const int rows = (int)t->shape[0];      // the module's own rows (a member of a fused convolution owns a row range)
    std::vector<float> tmp(count);
    if (what == 0) {
      for (int co = 0; co < rows; co++) for (int ci = 0; ci < c.cin; ci++) for (int tp = 0; tp < taps; tp++)
        tmp[((size_t)co * taps + tp) * c.cin + ci] = host[((size_t)co * c.cin + ci) * taps + tp];
      YS_CHECK_HIP(hipMemcpyAsync(dev, tmp.data(), count * 4, hipMemcpyHostToDevice, st));
      YS_CHECK_HIP(hipStreamSynchronize(st));
      m->weights_dirty = true;
    } else {
      YS_CHECK_HIP(hipMemcpyAsync(tmp.data(), dev, count * 4, hipMemcpyDeviceToHost, st));
      YS_CHECK_HIP(hipStreamSynchronize(st));
      for (int co = 0; co < rows; co++) for (int ci = 0; ci < c.cin; ci++) for (int tp = 0; tp < taps; tp++)
        host[((size_t)co * c.cin + ci) * taps + tp] = tmp[((size_t)co * taps + tp) * c.cin + ci];
    }
  } else {
    if (what == 0) YS_CHECK_HIP(hipMemcpyAsync(dev, host, count * 4, hipMemcpyHostToDevice, st));
    else YS_CHECK_HIP(hipMemcpyAsync(host, dev, count * 4, hipMemcpyDeviceToHost, st));
    YS_CHECK_HIP(hipStreamSynchronize(st));
    if (what == 0) m->eval_coeffs_dirty = true;
  }
  return YS_OK;
}
int ys_model_set_tensor(ys_model* m, const char* name, const float* host, size_t count) { return tensor_io(m, name, (float*)host, count, 0); }
int ys_model_get_tensor(ys_model* m, const char* name, float* host, size_t count) { return tensor_io(m, name, host, count, 1); }
int ys_model_get_grad(ys_model* m, const char* name, float* host, size_t count) { return tensor_io(m, name, host, count, 2); }

int ys_model_init_weights(ys_model* m, uint64_t seed) {
  YS_REQUIRE(m, "null model");
  std::vector<float> p(m->n_params, 0.f), s(m->n_state, 0.f);
  Rng rng{seed * 0x9E3779B97F4A7C15ull + 0x1234567ull};
  for (const auto& c : m->convs) {
    const long nw = c.dw ? (long)c.cout * 9 : c.ct ? 4L * c.cout * c.cin : (long)c.cout * c.k * c.k * c.cin;
    const float bound = c.ct ? 1.0f / sqrtf((float)(c.cout * 4)) : 1.0f / sqrtf((float)((c.dw ? 1 : c.cin) * c.k * c.k));   // kaiming_uniform(a=sqrt 5): 1/sqrt(fan_in)
    const long nwr = c.cout_real == c.cout ? nw : (long)c.cout_real * c.k * c.k * c.cin;   // padded rows stay zero
    for (long i = 0; i < nwr; i++) p[c.w_off + i] = rng.uni(bound);
    if (c.bn) {
      for (int i = 0; i < c.cout; i++) { p[c.g_off + i] = 1.f; p[c.b_off + i] = 0.f; s[c.rm_off + i] = 0.f; s[c.rv_off + i] = 1.f; }
      s[c.nbt_off] = 0.f;
    } else {
      for (int i = 0; i < c.cout_real; i++) p[c.g_off + i] = rng.uni(bound);   // Conv2d bias: U(-1/sqrt(fan_in), +)
    }
  }
  hipStream_t st = m->ctx->stream;
  YS_CHECK_HIP(hipMemcpyAsync(m->params, p.data(), p.size() * 4, hipMemcpyHostToDevice, st));
  YS_CHECK_HIP(hipMemcpyAsync(m->state, s.data(), s.size() * 4, hipMemcpyHostToDevice, st));
  YS_CHECK_HIP(hipMemsetAsync(m->grads, 0, (size_t)m->n_params * 4, st));
  YS_CHECK_HIP(hipMemsetAsync(m->adam_m, 0, (size_t)m->n_params * 4, st));
  YS_CHECK_HIP(hipMemsetAsync(m->adam_v, 0, (size_t)m->n_params * 4, st));
  YS_CHECK_HIP(hipStreamSynchronize(st));
  m->step = 0; m->weights_dirty = true; m->eval_coeffs_dirty = true;
  return YS_OK;
}

int ys_model_set_training(ys_model* m, int training) {
  YS_REQUIRE(m, "null model");
  m->training = training != 0;
  if (!m->training) m->eval_coeffs_dirty = true;
  return YS_OK;
}

int ys_model_forward(ys_model* m, const float* images, int on_device, int batch) {
  YS_REQUIRE(m && images, "ys_model_forward: null argument");
  YS_REQUIRE(!m->is_block && !m->is_head, "ys_model_forward: this handle is a block / head (use ys_block_forward or ys_head_forward)");
  YS_REQUIRE(batch > 0 && batch <= m->maxB, "ys_model_forward: batch %d outside (0, %d]", batch, m->maxB);
  YS_CHECK_HIP(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  const float* src = images;
  YS_TRY(join_wgrad_stream(m));   // the previous backward's weight-gradient kernels (model.0 reads the input buffer / the staged image) may still run on the second stream
  if (!on_device) {
    YS_CHECK_HIP(hipMemcpyAsync(m->img_dev, images, (size_t)batch * 3 * m->d.height * m->d.width * 4, hipMemcpyHostToDevice, st));
    src = m->img_dev;
  }
  YsTimer timer(m->ctx, "forward");
  m->B = batch;
  // model.0 reads the fp32 planes itself when it can (conv_stem.hip): no packed bf16 NHWC copy of the image.  The weight gradient of
  // this step reads the same planes again, so a device-resident image must stay unchanged until the step's backward has run (the
  // autograd rule of the reference: Utils/Amp.cs:348 differentiates through the tensor it was handed).
  m->in_f32 = nullptr;
  if (stem_direct(m)) m->in_f32 = src;
  else YS_TRY(ys_pack_input_launch(st, m->dtype, src, batch, 3, m->d.height, m->d.width, m->epl, m->bufs[m->in_buf].act));
  YS_TRY(forward_impl(m, batch));
  m->have_fwd = true; m->fwd_training = m->training; m->have_loss = false; m->have_seg_loss = false;
  return YS_OK;
}

// Predict-side input path (Models/Detector.cs:31-41): uint8 RGB planes [B,3,h,w] -> zero-pad mode with value 114 on the bottom /
// right up to the model's (H, W) -> / 255 -> the NHWC input buffer, in one kernel (no fp32 image is materialised).
int ys_model_forward_u8(ys_model* m, const uint8_t* images, int on_device, int batch, int h, int w) {
  YS_REQUIRE(m && images, "ys_model_forward_u8: null argument");
  YS_REQUIRE(!m->is_block && !m->is_head, "ys_model_forward_u8: this handle is a block / head (use ys_block_forward or ys_head_forward)");
  YS_REQUIRE(batch > 0 && batch <= m->maxB, "ys_model_forward_u8: batch %d outside (0, %d]", batch, m->maxB);
  YS_REQUIRE(h > 0 && w > 0 && h <= m->d.height && w <= m->d.width, "ys_model_forward_u8: image %dx%d does not fit the model's %dx%d", h, w, m->d.height, m->d.width);
  YS_CHECK_HIP(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  const unsigned char* src = images;
  if (!on_device) {
    YS_TRY(join_wgrad_stream(m)); // the staging buffer may be the fp32 image the previous step's model.0 weight gradient still reads
    YS_CHECK_HIP(hipMemcpyAsync(m->img_dev, images, (size_t)batch * 3 * h * w, hipMemcpyHostToDevice, st));
    src = (const unsigned char*)m->img_dev;
  }
  YsTimer timer(m->ctx, "forward");
  m->B = batch;
  YS_TRY(join_wgrad_stream(m));   // as in ys_model_forward: the input buffer is rewritten below
  m->in_f32 = nullptr;            // uint8 planes: the packed input buffer is the layer's input
  YS_TRY(ys_pack_input_u8_launch(st, m->dtype, src, batch, 3, h, w, m->d.height, m->d.width, m->epl, m->bufs[m->in_buf].act));
  YS_TRY(forward_impl(m, batch));
  m->have_fwd = true; m->fwd_training = m->training; m->have_loss = false; m->have_seg_loss = false;
  return YS_OK;
}

int ys_model_get_output(ys_model* m, const char* key, float* host, size_t count) {
  YS_REQUIRE(m && key && host, "ys_model_get_output: null argument");
  YS_REQUIRE(!m->is_block, "ys_model_get_output: this handle is a block (use ys_block_forward / ys_block_backward)");
  YS_REQUIRE(m->have_fwd, "ys_model_get_output: no forward has run");
  hipStream_t st = m->ctx->stream;
  const int B = m->B;
  const std::string k(key);
  if (k == "boxes" || k == "scores") {
    const bool bx = k == "boxes";
    const int C = bx ? 4 * m->d.reg_max : m->d.nc;
    YS_REQUIRE(count == (size_t)B * C * m->A, "ys_model_get_output(%s): expected %zu elements", key, (size_t)B * C * m->A);
    const Buf& b = m->bufs[bx ? m->pd_buf : m->ps_buf];
    YS_TRY(ys_unpack_nchw_launch(st, m->dtype, b.act, b.ldc, 0, B, C, m->A, m->out_stage));
    YS_CHECK_HIP(hipMemcpyAsync(host, m->out_stage, count * 4, hipMemcpyDeviceToHost, st));
  } else if (k == "dboxes" || k == "dscores") {
    const bool bx = k == "dboxes";
    const int C = bx ? 4 * m->d.reg_max : m->d.nc;
    YS_REQUIRE(m->have_loss, "ys_model_get_output(%s): no loss has run", key);
    YS_REQUIRE(count == (size_t)B * C * m->A, "ys_model_get_output(%s): expected %zu elements", key, (size_t)B * C * m->A);
    const Buf& b = m->bufs[bx ? m->pd_buf : m->ps_buf];
    YS_TRY(ys_unpack_nchw_launch(st, m->dtype, b.grad, b.ldc, 0, B, C, m->A, m->out_stage));
    YS_CHECK_HIP(hipMemcpyAsync(host, m->out_stage, count * 4, hipMemcpyDeviceToHost, st));
  } else if (m->segment && (k == "mask_coefficient" || k == "dmask_coefficient")) {   // Head.cs:290-296: [B][nm][A]
    const bool g = k[0] == 'd';
    YS_REQUIRE(!g || m->have_seg_loss, "ys_model_get_output(%s): no segment loss has run", key);
    YS_REQUIRE(count == (size_t)B * m->nm * m->A, "ys_model_get_output(%s): expected %zu elements", key, (size_t)B * m->nm * m->A);
    const Buf& b = m->bufs[m->mc_buf];
    YS_TRY(ys_unpack_nchw_launch(st, m->dtype, g ? b.grad : b.act, b.ldc, 0, B, m->nm, m->A, m->out_stage));
    YS_CHECK_HIP(hipMemcpyAsync(host, m->out_stage, count * 4, hipMemcpyDeviceToHost, st));
  } else if (m->segment && (k == "proto" || k == "dproto")) {                           // Head.cs:289: [B][nm][mh][mw]
    const bool g = k[0] == 'd';
    const long np = (long)m->mh * m->mw;
    YS_REQUIRE(!g || m->have_seg_loss, "ys_model_get_output(%s): no segment loss has run", key);
    YS_REQUIRE(count == (size_t)B * m->nm * np, "ys_model_get_output(%s): expected %zu elements", key, (size_t)B * m->nm * np);
    const Buf& b = m->bufs[m->pr_buf];
    YS_TRY(ys_unpack_nchw_launch(st, m->dtype, g ? b.grad : b.act, b.ldc, 0, B, m->nm, np, m->out_stage));
    YS_CHECK_HIP(hipMemcpyAsync(host, m->out_stage, count * 4, hipMemcpyDeviceToHost, st));
  } else if ((m->xkind == 3 && k == "dkpts") || (m->xkind == 2 && k == "dangle")) {   // d(sum(loss * B)) / d(raw kpts | angle LOGIT) [B][nk|1][A]
    YS_REQUIRE(m->have_seg_loss, "ys_model_get_output(%s): the model's criterion has not run", key);
    YS_REQUIRE(count == (size_t)B * m->nm * m->A, "ys_model_get_output(%s): expected %zu elements", key, (size_t)B * m->nm * m->A);
    const Buf& b = m->bufs[m->mc_buf];
    YS_TRY(ys_unpack_nchw_launch(st, m->dtype, b.grad, b.ldc, 0, B, m->nm, m->A, m->out_stage));
    YS_CHECK_HIP(hipMemcpyAsync(host, m->out_stage, count * 4, hipMemcpyDeviceToHost, st));
  } else if ((m->xkind == 2 && k == "angle") || (m->xkind == 3 && k == "kpts")) {
    // Obb.forward_head: angle = (sigmoid(cat cv4) - 0.25) * pi [B][ne][A] (Head.cs:421-433); Pose.forward_head: raw kpts [B][nk][A] (:531-543)
    YS_REQUIRE(count == (size_t)B * m->nm * m->A, "ys_model_get_output(%s): expected %zu elements", key, (size_t)B * m->nm * m->A);
    const Buf& b = m->bufs[m->mc_buf];
    YS_TRY(ys_unpack_nchw_launch(st, m->dtype, b.act, b.ldc, 0, B, m->nm, m->A, m->out_stage));
    if (m->xkind == 2) YS_TRY(ys_obb_angle_launch(st, m->out_stage, (long)count));
    YS_CHECK_HIP(hipMemcpyAsync(host, m->out_stage, count * 4, hipMemcpyDeviceToHost, st));
  } else if (k == "pred") {
    const size_t pc = (size_t)(4 + m->d.nc + m->nm);
    YS_REQUIRE(!m->training, "ys_model_get_output(pred): model is in training mode (Detect returns preds only, Head.cs:103-106)");
    YS_REQUIRE(count == (size_t)B * pc * m->A, "ys_model_get_output(pred): expected %zu elements", (size_t)B * pc * m->A);
    YS_CHECK_HIP(hipMemcpyAsync(host, m->pred, count * 4, hipMemcpyDeviceToHost, st));
  } else {
    ys_set_error("ys_model_get_output: unknown key '%s'", key);
    return YS_ERR_INVALID_ARG;
  }
  YS_CHECK_HIP(hipStreamSynchronize(st));
  return YS_OK;
}

// The criterion's `preds` argument supplied by the caller (Loss.cs:411 `forward(preds, batch)`): head outputs in the reference layout
// -- boxes [B, 4*reg_max, A], scores [B, nc, A] and, for Segment models, mask_coefficient [B, nm, A] and proto [B, nm, H/4, W/4] --
// are packed into the engine's head buffers as if a forward had produced them.  ys_loss_detect / ys_loss_segment and the
// "dboxes" / "dscores" / ... gradient outputs then work on them; ys_model_backward is refused (no graph state behind these preds).
int ys_model_set_preds(ys_model* m, int batch, const float* boxes, const float* scores, const float* mask_coefficient, const float* proto) {
  YS_REQUIRE(m && !m->is_block && boxes && scores, "ys_model_set_preds: null argument or block handle");
  YS_REQUIRE(batch > 0 && batch <= m->maxB, "ys_model_set_preds: batch %d outside (0, %d]", batch, m->maxB);
  YS_REQUIRE(!m->segment || (mask_coefficient && proto), "ys_model_set_preds: a Segment model needs mask_coefficient and proto");
  YS_REQUIRE(m->xkind < 2 || mask_coefficient, "ys_model_set_preds: a Pose / Obb model takes its raw kpts [B,nk,A] / angle LOGITS [B,1,A] in the mask_coefficient argument");
  YS_CHECK_HIP(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  struct Item { const float* src; int buf; int C; long rows; } items[4] = {
    {boxes, m->pd_buf, 4 * m->d.reg_max, m->A}, {scores, m->ps_buf, m->d.nc, m->A},
    {m->segment || m->xkind >= 2 ? mask_coefficient : nullptr, m->mc_buf, m->nm, m->A}, {m->segment ? proto : nullptr, m->pr_buf, m->nm, (long)m->mh * m->mw}};
  for (const Item& it : items) {
    if (!it.src) continue;
    const Buf& b = m->bufs[it.buf];
    const size_t cnt = (size_t)batch * it.C * it.rows;
    YS_REQUIRE((long)cnt <= m->n_out_stage, "ys_model_set_preds: staging buffer too small");
    YS_CHECK_HIP(hipMemcpyAsync(m->out_stage, it.src, cnt * 4, hipMemcpyHostToDevice, st));
    YS_TRY(ys_pack_input_launch(st, m->dtype, m->out_stage, batch, it.C, 1, (int)it.rows, b.ldc, b.act));
    YS_CHECK_HIP(hipStreamSynchronize(st));   // out_stage is reused by the next item
  }
  m->B = batch; m->have_fwd = true; m->fwd_training = false; m->have_loss = false; m->have_seg_loss = false;
  return YS_OK;
}

int ys_model_pred_device(ys_model* m, float** dptr) {
  YS_REQUIRE(m && dptr, "null argument");
  *dptr = m->pred;
  return YS_OK;
}

int ys_model_reserve_labels(ys_model* m, int per_image) {
  YS_REQUIRE(m && !m->is_block, "ys_model_reserve_labels: needs a full model");
  YS_REQUIRE(per_image > 0, "ys_model_reserve_labels: per_image = %d", per_image);
  if (per_image <= m->gcap) return YS_OK;
  YS_CHECK_HIP(hipSetDevice(m->ctx->device));
  m->have_loss = false; m->have_seg_loss = false;
  return alloc_label_ws(m, (per_image + 15) / 16 * 16);
}

static int loss_detect_core(ys_model* m, const float* batch_idx, const float* cls, const float* bboxes, int n, int on_device, bool aux_follows);

int ys_loss_detect(ys_model* m, const float* batch_idx, const float* cls, const float* bboxes, int n, int on_device) {
  return loss_detect_core(m, batch_idx, cls, bboxes, n, on_device, false);
}

static int loss_detect_core(ys_model* m, const float* batch_idx, const float* cls, const float* bboxes, int n, int on_device, bool aux_follows) {
  YS_REQUIRE(m, "null model");
  // Training forward -> the criterion feeds backward (Amp.cs:338-348).  Eval forward -> validation loss on the eval-mode preds
  // (Detector.cs:94-97): the head logits are produced in both modes; only backward needs the training-mode state.
  YS_REQUIRE(!m->is_block && m->have_fwd, "ys_loss_detect: needs a forward of a full model first");
  YS_REQUIRE(m->xkind != 2 || aux_follows, "ys_loss_detect: an OBB model's criterion is ys_loss_obb (oriented labels, Loss.cs:486-684)");
  YS_REQUIRE(m->xkind != 3 || aux_follows, "ys_loss_detect: a Pose model's criterion is ys_loss_pose (keypoint terms, Loss.cs:870-1071)");
  const bool rot = m->xkind == 2;
  const size_t lbytes = rot ? 20 : 16;
  YS_REQUIRE(n >= 0, "ys_loss_detect: n_labels = %d", n);
  YS_REQUIRE(n == 0 || (batch_idx && cls && bboxes), "ys_loss_detect: null label arrays");
  YS_CHECK_HIP(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  const float *bi = batch_idx, *cl = cls, *bb = bboxes;
  int host_cmax = 0;                           // host labels: the largest per-image count (0 = unknown)
  if (!on_device && n > 0) {
    // host labels: size the padded GT workspace from the batch itself, like the reference's counts.max() (Loss.cs:376-380)
    std::vector<int> cnt(m->B, 0);
    int mx = 0;
    for (int i = 0; i < n; i++) { const int b = (int)batch_idx[i]; if (b >= 0 && b < m->B) mx = std::max(mx, ++cnt[b]); }
    // every label row is staged (rows whose batch_idx lies outside [0, B) are ignored by the kernels, like the reference's
    // `batch_idx == j` matches): the staging arrays hold gcap * max_batch rows, so n itself bounds the capacity too
    host_cmax = mx > 0 ? mx : 1;
    const int per_rows = (n + m->maxB - 1) / m->maxB;
    if (per_rows > mx) mx = per_rows;
    if (mx > m->gcap) YS_TRY(alloc_label_ws(m, (mx + 15) / 16 * 16));
    YS_REQUIRE(n <= m->max_labels, "ys_loss_detect: %d label rows exceed the staging capacity %d", n, m->max_labels);
    YS_CHECK_HIP(hipMemcpyAsync(m->lab_bidx, batch_idx, (size_t)n * 4, hipMemcpyHostToDevice, st));
    YS_CHECK_HIP(hipMemcpyAsync(m->lab_cls, cls, (size_t)n * 4, hipMemcpyHostToDevice, st));
    YS_CHECK_HIP(hipMemcpyAsync(m->lab_box, bboxes, (size_t)n * lbytes, hipMemcpyHostToDevice, st));
    bi = m->lab_bidx; cl = m->lab_cls; bb = m->lab_box;
  }
  YsTimer timer(m->ctx, "loss");
  LossArgs a{};
  a.pd = m->bufs[m->pd_buf].act; a.ps = m->bufs[m->ps_buf].act; a.dpd = m->bufs[m->pd_buf].grad; a.dps = m->bufs[m->ps_buf].grad;
  a.ld_pd = m->ld_pd; a.ld_ps = m->ld_ps; a.B = m->B; a.A = m->A; a.nc = m->d.nc; a.reg_max = m->d.reg_max;
  a.H = m->d.height; a.W = m->d.width; a.nl = m->nl;
  for (int i = 0; i < 4; i++) { a.lvl_off[i] = m->lvl_off[i]; a.lvl_w[i] = m->lvl_w[i]; a.lvl_h[i] = m->lvl_h[i]; a.lvl_stride[i] = m->lvl_stride[i]; }
  a.batch_idx = bi; a.cls = cl; a.bboxes = bb; a.n_labels = n; a.gcap = m->gcap;
  a.gmax = (host_cmax > 0 && host_cmax < m->gcap) ? host_cmax : m->gcap;
  a.gt_count = m->gt_count; a.gt_box = m->gt_box; a.gt_cls = m->gt_cls; a.pbox = m->pbox; a.ov = m->ov; a.align = m->align;
  a.mpos = m->mpos; a.pos_align = m->pos_align; a.pos_ov = m->pos_ov; a.fg_gt = m->fg_gt; a.tnorm = m->tnorm;
  a.partial = m->loss_partial; a.scalars = m->scalars;
  a.hyp_box = 7.5f; a.hyp_cls = 0.5f; a.hyp_dfl = 1.5f; a.topk = 10;   // Loss.cs:344,357
  if (rot) {                                                             // Loss.cs:489: hyp_angle = 1
    const Buf& ab = m->bufs[m->mc_buf];
    a.rot = 1; a.pa = ab.act; a.dpa = ab.grad; a.ld_pa = m->ld_mc; a.hyp_angle = 1.0f;
  }
  YS_TRY(ys_loss_detect_launch(st, m->dtype, a));
  YS_CHECK_HIP(hipGetLastError());
  m->have_loss = true;
  return YS_OK;
}

// v8SegmentationLoss (Loss.cs:711-780): detection part + assignment (loss.hip), then the mask term (segloss.hip).
// masks: [B][mh][mw] fp32, overlap-encoded instance ids (0 = background, g+1 = the image's g-th label; YoloDataset.cs:265-267).
int ys_loss_segment(ys_model* m, const float* batch_idx, const float* cls, const float* bboxes, int n, const float* masks, int on_device,
                    int crop_mode) {
  YS_REQUIRE(m && m->segment, "ys_loss_segment: model has no Segment head");
  YS_REQUIRE(masks, "ys_loss_segment: null masks");
  YS_TRY(ys_loss_detect(m, batch_idx, cls, bboxes, n, on_device));
  m->have_loss = false;
  hipStream_t st = m->ctx->stream;
  const float* mk = masks;
  if (!on_device) {
    YS_CHECK_HIP(hipMemcpyAsync(m->masks_dev, masks, (size_t)m->B * m->mh * m->mw * 4, hipMemcpyHostToDevice, st));
    mk = m->masks_dev;
  }
  YsTimer timer(m->ctx, "loss_seg");
  const Buf& mc = m->bufs[m->mc_buf];
  const Buf& pr = m->bufs[m->pr_buf];
  YS_TRY(ys_loss_segment_launch(st, m->dtype, mc.act, mc.grad, m->ld_mc, pr.act, pr.grad, m->ld_pr, mk, m->fg_gt, m->gt_box, m->seg_cnt,
                                m->seg_off, m->seg_list, m->seg_ent, m->seg_part, m->scalars, m->B, m->A, m->nm, m->mh, m->mw, m->gcap,
                                m->d.height, m->d.width, crop_mode));
  YS_CHECK_HIP(hipGetLastError());
  m->have_loss = true; m->have_seg_loss = true;
  return YS_OK;
}

// v8OBBLoss (Loss.cs:486-684): the loss.hip pipeline in its rotated mode (probiou assigner and box term, rbox2dist DFL targets,
// angle term).  bboxes: fp32 [n][5] = normalised cx, cy, w, h + angle in radians.
int ys_loss_obb(ys_model* m, const float* batch_idx, const float* cls, const float* bboxes, int n, int on_device) {
  YS_REQUIRE(m && m->xkind == 2, "ys_loss_obb: model has no Obb head");
  YS_TRY(loss_detect_core(m, batch_idx, cls, bboxes, n, on_device, true));
  m->have_seg_loss = true;
  return YS_OK;
}

// v8PoseLoss (Loss.cs:870-1071): detection part + assignment (loss.hip), then the keypoint terms (poseloss.hip).
// keypoints: fp32 [n][kpt_num][kpt_dim] normalised to the image like bboxes (x, y[, visibility]); row i belongs to label i.
int ys_loss_pose(ys_model* m, const float* batch_idx, const float* cls, const float* bboxes, int n, const float* keypoints, int on_device) {
  YS_REQUIRE(m && m->xkind == 3, "ys_loss_pose: model has no Pose head");
  YS_REQUIRE(n == 0 || keypoints, "ys_loss_pose: null keypoints");
  if (!on_device && batch_idx)                 // keypoint rows are addressed by a label's rank within its image: collate order only
    for (int i = 1; i < n; i++)
      YS_REQUIRE(batch_idx[i] >= batch_idx[i - 1], "ys_loss_pose: labels must be grouped by image in collate order (batch_idx[%d] = %g < batch_idx[%d] = %g)",
                 i, (double)batch_idx[i], i - 1, (double)batch_idx[i - 1]);
  YS_TRY(loss_detect_core(m, batch_idx, cls, bboxes, n, on_device, true));
  m->have_loss = false;
  hipStream_t st = m->ctx->stream;
  const float* kp = keypoints;
  if (!on_device && n > 0) {
    YS_REQUIRE(n <= m->max_labels, "ys_loss_pose: %d labels exceed the staging capacity %d", n, m->max_labels);
    YS_CHECK_HIP(hipMemcpyAsync(m->kp_dev, keypoints, (size_t)n * m->nm * 4, hipMemcpyHostToDevice, st));
    kp = m->kp_dev;
  }
  YsTimer timer(m->ctx, "loss_pose");
  const Buf& kb = m->bufs[m->mc_buf];
  PoseArgs a{};
  a.kp = kb.act; a.dkp = kb.grad; a.ld = m->ld_mc; a.fg_gt = m->fg_gt; a.gt_box = m->gt_box;
  a.gt_src = m->gt_cls + 2L * m->B * m->gcap;
  a.keypoints = kp; a.part = m->seg_part; a.scalars = m->scalars;
  a.B = m->B; a.A = m->A; a.K = m->nm / m->kdim; a.D = m->kdim; a.gcap = m->gcap; a.H = m->d.height; a.W = m->d.width; a.nl = m->nl;
  for (int i = 0; i < 4; i++) { a.lvl_off[i] = m->lvl_off[i]; a.lvl_w[i] = m->lvl_w[i]; a.lvl_stride[i] = m->lvl_stride[i]; }
  a.hyp_pose = 12.0f; a.hyp_kobj = 1.0f;                                                  // Loss.cs:896
  static const float oks[17] = {0.026f, 0.025f, 0.025f, 0.035f, 0.035f, 0.079f, 0.079f, 0.072f, 0.072f, 0.062f, 0.062f, 0.107f, 0.107f,
                                0.087f, 0.087f, 0.089f, 0.089f};                          // OKS_SIGMA (Loss.cs:9-16)
  const bool coco = a.K == 17 && a.D == 3;                                                // Loss.cs:903-905
  for (int k = 0; k < a.K && k < YS_POSE_KMAX; k++) a.sigma[k] = coco ? oks[k] : 1.0f / (float)a.K;
  YS_TRY(ys_loss_pose_launch(st, m->dtype, a, m->seg_cnt, m->seg_off, m->seg_list));
  YS_CHECK_HIP(hipGetLastError());
  m->have_loss = true; m->have_seg_loss = true;
  return YS_OK;
}

// device-resident labels cannot size the workspace without a host sync: the prep kernel records the batch's largest per-image
// label count and the first synchronising read refuses a truncated assignment instead of returning it
static int check_label_overflow(ys_model* m, float max_count) {
  if ((int)max_count <= m->gcap) return YS_OK;
  m->have_loss = false; m->have_seg_loss = false;
  ys_set_error("loss: an image of this batch has %d labels but the workspace holds %d per image (the reference pads to the batch maximum, "
               "Loss.cs:363-390): call ys_model_reserve_labels(model, %d) or pass max_labels at creation, then repeat the step",
               (int)max_count, m->gcap, (int)max_count);
  return YS_ERR_INVALID_ARG;
}

// loss items in the reference's order: detect [box, cls, dfl] (Loss.cs:414); segment [box, seg, cls, dfl, semseg] (Loss.cs:719)
int ys_loss_read_items(ys_model* m, float* items, int n_items, float* loss_sum) {
  YS_REQUIRE(m && m->have_loss, "ys_loss_read_items: no loss has run");
  YS_REQUIRE(items && n_items == m->n_items, "ys_loss_read_items: this model's criterion has %d items", m->n_items);
  float h[16];
  YS_CHECK_HIP(hipMemcpyAsync(h, m->scalars, sizeof(h), hipMemcpyDeviceToHost, m->ctx->stream));
  YS_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
  YS_TRY(check_label_overflow(m, h[15]));
  if (m->segment) {
    YS_REQUIRE(m->have_seg_loss, "ys_loss_read_items: the Segment model needs ys_loss_segment");
    items[0] = h[1]; items[1] = h[8]; items[2] = h[2]; items[3] = h[3]; items[4] = 0.f;
  } else if (m->xkind == 2) {
    items[0] = h[1]; items[1] = h[2]; items[2] = h[3]; items[3] = h[13];                     // box, cls, dfl, angle (Loss.cs:619)
  } else if (m->xkind == 3) {
    YS_REQUIRE(m->have_seg_loss, "ys_loss_read_items: the Pose model needs ys_loss_pose");
    items[0] = h[1]; items[1] = h[10]; items[2] = h[11]; items[3] = h[2]; items[4] = h[3];   // box, pose, kobj, cls, dfl (Loss.cs:965)
  } else {
    items[0] = h[1]; items[1] = h[2]; items[2] = h[3];
  }
  if (loss_sum) *loss_sum = h[4];
  return YS_OK;
}

int ys_loss_read(ys_model* m, float loss_items[3], float* loss_sum) {
  YS_REQUIRE(m && m->have_loss, "ys_loss_read: no loss has run");
  float h[16];
  YS_CHECK_HIP(hipMemcpyAsync(h, m->scalars, sizeof(h), hipMemcpyDeviceToHost, m->ctx->stream));
  YS_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
  YS_TRY(check_label_overflow(m, h[15]));
  if (loss_items) { loss_items[0] = h[1]; loss_items[1] = h[2]; loss_items[2] = h[3]; }
  if (loss_sum) *loss_sum = h[4];
  return YS_OK;
}

int ys_model_backward_segments(ys_model* m) { (void)m; return ys_model::NSEG; }

int ys_model_backward_segment(ys_model* m, int seg) {
  YS_REQUIRE(m && m->have_loss, "ys_model_backward: needs forward + loss first");
  YS_REQUIRE(m->fwd_training, "ys_model_backward: the last forward ran in eval mode (no batch statistics / pre-BN outputs were kept)");
  YS_REQUIRE(seg >= 0 && seg < ys_model::NSEG, "ys_model_backward_segment: segment %d out of range", seg);
  YS_CHECK_HIP(hipSetDevice(m->ctx->device));
  if (seg == 0) reset_grad_state(m);
  return backward_range(m, seg, seg);
}

int ys_model_backward_segment_async(ys_model* m, int seg) {
  YS_REQUIRE(m && m->have_loss, "ys_model_backward: needs forward + loss first");
  YS_REQUIRE(m->fwd_training, "ys_model_backward: the last forward ran in eval mode (no batch statistics / pre-BN outputs were kept)");
  YS_REQUIRE(seg >= 0 && seg < ys_model::NSEG, "ys_model_backward_segment_async: segment %d out of range", seg);
  YS_CHECK_HIP(hipSetDevice(m->ctx->device));
  if (seg == 0) reset_grad_state(m);
  return backward_range(m, seg, seg, true);
}

int ys_model_segment_fence(ys_model* m, int seg, void* stream) {
  YS_REQUIRE(m && seg >= 0 && seg < ys_model::NSEG, "ys_model_segment_fence: bad argument");
  YS_REQUIRE(m->ev_seg_m[seg], "ys_model_segment_fence: model not built");
  YS_CHECK_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = (hipStream_t)stream;
  YS_CHECK_HIP(hipStreamWaitEvent(s, m->ev_seg_m[seg], 0));
  if (m->seg_on_st2[seg]) YS_CHECK_HIP(hipStreamWaitEvent(s, m->ev_seg_w[seg], 0));
  return YS_OK;
}

int ys_model_backward(ys_model* m) {
  YS_REQUIRE(m && m->have_loss, "ys_model_backward: needs forward + loss first");
  YS_REQUIRE(m->fwd_training, "ys_model_backward: the last forward ran in eval mode (no batch statistics / pre-BN outputs were kept)");
  YS_REQUIRE(!m->segment || m->have_seg_loss, "ys_model_backward: the Segment model needs ys_loss_segment (mask gradients)");
  YS_CHECK_HIP(hipSetDevice(m->ctx->device));
  YsTimer timer(m->ctx, "backward");
  reset_grad_state(m);
  // Round 6: the one-call backward ends every segment asynchronously as well -- a segment's split reduction (wgrad_reduce_batched_kernel: 1.08 GB of partial slabs per
  // YOLOv8n step, 0.21 ms when it runs alone at the end) goes to the weight-gradient stream behind that segment's weight gradients and runs beside the next segment's
  // BN-backward / dgrad chain; only the last (stem) segment's share is left for the end.  Nothing on the main stream waits: AdamW, zero_grad, gradient reads and the
  // next forward order themselves behind the second stream (join_wgrad_stream), as they do after ys_model_backward_segment_async.  Same kernels, same operands, same
  // order per stream: bit-identical gradients (tests/test_dist.py::test_async_segment_ends_give_the_same_gradients compares the two forms).
  if (m->overlap) {
    m->hold_stem = true;
    int rc = YS_OK;
    for (int sg = 0; sg < ys_model::NSEG && rc == YS_OK; sg++) rc = backward_range(m, sg, sg, true, false);
    m->hold_stem = false;
    return rc;
  }
  return backward_range(m, 0, ys_model::NSEG - 1);
}

int ys_model_segment_grad_range(ys_model* m, int seg, int64_t* offset, int64_t* count) {
  YS_REQUIRE(m && offset && count && seg >= 0 && seg < ys_model::NSEG, "ys_model_segment_grad_range: bad argument");
  *offset = m->seg_group[seg][0].off;
  *count = m->seg_group[seg][0].count + m->seg_group[seg][1].count + m->seg_group[seg][2].count;
  return YS_OK;
}

int ys_model_set_overlap(ys_model* m, int on) {
  YS_REQUIRE(m, "null model");
  if (m->st2_dirty) {                     // drain the weight-gradient stream before the mode changes
    YS_CHECK_HIP(hipEventRecord(m->ev_join, m->st2));
    YS_CHECK_HIP(hipStreamWaitEvent(m->ctx->stream, m->ev_join, 0));
    m->st2_dirty = false;
  }
  m->overlap = on != 0 && m->overlap_built;
  return YS_OK;
}

int ys_model_zero_grad(ys_model* m) {
  YS_REQUIRE(m, "null model");
  YS_TRY(join_wgrad_stream(m));              // asynchronous segment ends: the weight-gradient stream may still be adding into the buffer
  YS_CHECK_HIP(hipMemsetAsync(m->grads, 0, (size_t)m->n_params * 4, m->ctx->stream));
  return YS_OK;
}

int ys_model_grad_buffer(ys_model* m, float** dptr, int64_t* count) {
  YS_REQUIRE(m && dptr && count, "null argument");
  *dptr = m->grads; *count = m->n_params;
  return YS_OK;
}
int ys_model_param_buffer(ys_model* m, float** dptr, int64_t* count) {
  YS_REQUIRE(m && dptr && count, "null argument");
  *dptr = m->params; *count = m->n_params;
  return YS_OK;
}

// Parameter-group construction of the optimizer.  mode 0 (default): three DISJOINT groups (bias | conv weight | bn weight).
// mode 1: the reference's groups exactly as written (YoloBaseTaskModel.cs:144-151) -- name.Contains("bias") | Contains("weight") |
// Contains("bn") -- in which every BatchNorm weight and bias is listed twice; TorchSharp keys the optimizer state by parameter, so
// such a parameter receives two AdamW updates per step() from one shared state.  Must be chosen before the
// first optimizer step.
int ys_optim_set_param_groups(ys_model* m, int mode) {
  YS_REQUIRE(m && (mode == 0 || mode == 1), "ys_optim_set_param_groups: mode %d", mode);
  YS_REQUIRE(m->step == 0 || mode == m->group_mode, "ys_optim_set_param_groups: the optimizer has already stepped");
  if (mode == 1 && !m->bn_mask) {
    YS_CHECK_HIP(hipSetDevice(m->ctx->device));
    std::vector<unsigned char> h((size_t)m->n_params, 0);
    for (const auto& c : m->convs)
      if (c.bn) for (int i = 0; i < c.cout; i++) { h[c.g_off + i] = 1; h[c.b_off + i] = 1; }
    YS_TRY(dev_alloc(m, (void**)&m->bn_mask, h.size(), false));
    YS_CHECK_HIP(hipMemcpyAsync(m->bn_mask, h.data(), h.size(), hipMemcpyHostToDevice, m->ctx->stream));
    YS_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
  }
  m->group_mode = mode;
  return YS_OK;
}

int ys_optim_adamw_step(ys_model* m, const float* lr_per_group, int ngroups, float beta1, float beta2, float eps, float wd) {
  YS_REQUIRE(m && lr_per_group && ngroups >= 1, "ys_optim_adamw_step: bad argument");
  YS_CHECK_HIP(hipSetDevice(m->ctx->device));
  YS_TRY(join_wgrad_stream(m));              // asynchronous segment ends (ys_model_backward_segment_async): every gradient is in before the update
  YsTimer timer(m->ctx, "optim");
  m->step += 1;
  const float bc1 = 1.0f - powf(beta1, (float)m->step), bc2 = 1.0f - powf(beta2, (float)m->step);
  AdamwRanges rg{};                          // one launch for the 3 segments x 3 groups (was nine ~5 us launches)
  auto lr_of = [&](int g) { return lr_per_group[g < ngroups ? g : ngroups - 1]; };
  for (int seg = 0; seg < ys_model::NSEG; seg++)
    for (int g = 0; g < 3; g++) {
      const auto r = m->seg_group[seg][g];
      if (r.count <= 0) continue;
      // reference mode: bn.weight is met first in the "weight" group (lr of group 1), then again in the "bn" group
      rg.off[rg.n] = r.off; rg.count[rg.n] = r.count; rg.lr[rg.n] = lr_of(m->group_mode == 1 && g == 2 ? 1 : g);
      rg.n++;
    }
  AdamwDup dup{};
  if (m->group_mode == 1) {
    const double s1 = 2.0 * (double)m->step - 1.0, s2 = 2.0 * (double)m->step;
    dup.mask = m->bn_mask; dup.lr_second = lr_of(2);
    dup.bc1_first = (float)(1.0 - pow((double)beta1, s1)); dup.bc2s_first = sqrtf((float)(1.0 - pow((double)beta2, s1)));
    dup.bc1_second = (float)(1.0 - pow((double)beta1, s2)); dup.bc2s_second = sqrtf((float)(1.0 - pow((double)beta2, s2)));
  }
  YS_TRY(ys_adamw_ranges_launch(m->ctx->stream, m->params, m->grads, m->adam_m, m->adam_v, m->n_params, rg, beta1, beta2, eps, wd, bc1, bc2,
                                m->group_mode == 1 ? &dup : nullptr));
  m->weights_dirty = true; m->eval_coeffs_dirty = true;
  YS_CHECK_HIP(hipGetLastError());
  return YS_OK;
}


// ------------------------------------------------------------------ standalone blocks (include/yolosharp_hip.h "per-block entry points")
int ys_block_create(ys_ctx* ctx, const ys_block_desc* bd, ys_model** out) {
  YS_REQUIRE(ctx && bd && out, "ys_block_create: null argument");
  YS_REQUIRE(bd->dtype == YS_F32 || bd->dtype == YS_BF16 || bd->dtype == YS_FP8, "ys_block_create: dtype %d unsupported", bd->dtype);
  YS_REQUIRE(bd->height > 0 && bd->width > 0 && bd->max_batch > 0, "ys_block_create: geometry %dx%d batch %d", bd->height, bd->width, bd->max_batch);
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  ys_model* m = new ys_model();
  m->f8 = bd->dtype == YS_FP8;
  const int bstore = m->f8 ? YS_BF16 : bd->dtype;
  m->ctx = ctx; m->dtype = bstore; m->epl = bstore == YS_BF16 ? 8 : 4; m->es = bstore == YS_BF16 ? 2 : 4;
  m->maxB = bd->max_batch; m->is_block = true; m->blk_c1 = bd->c1; m->blk_c2 = bd->c2; m->A = 0; m->nl = 0;
  m->d = ys_model_desc{}; m->d.height = bd->height; m->d.width = bd->width; m->d.max_batch = bd->max_batch; m->d.dtype = bd->dtype;
  m->d.reg_max = 1; m->d.max_labels = 1;
  int st = build_block(m, *bd);
  if (st == YS_OK) st = layout_params(m);
  if (st == YS_OK) {
    for (auto& t : m->tensors) if (t.name.compare(0, 6, "block.") == 0) t.name.erase(0, 6);   // module-relative names
    st = allocate(m);
  }
  if (st == YS_OK) st = ys_model_init_weights(m, 0);
  if (st != YS_OK) { ys_model_destroy(m); return st; }
  *out = m;
  return YS_OK;
}

int ys_block_output_shape(ys_model* m, int32_t shape_chw[3]) {
  YS_REQUIRE(m && m->is_block && shape_chw, "ys_block_output_shape: not a block handle");
  const Buf& ob = m->bufs[m->blk_out];
  shape_chw[0] = m->blk_c2; shape_chw[1] = ob.H; shape_chw[2] = ob.W;
  return YS_OK;
}

int ys_block_forward(ys_model* m, const float* x, int on_device, int batch, float* y) {
  YS_REQUIRE(m && m->is_block && x && y, "ys_block_forward: null argument or not a block handle");
  YS_REQUIRE(batch > 0 && batch <= m->maxB, "ys_block_forward: batch %d outside (0, %d]", batch, m->maxB);
  YS_CHECK_HIP(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  const Buf& ib = m->bufs[m->in_buf];
  const Buf& ob = m->bufs[m->blk_out];
  const size_t nx = (size_t)batch * m->blk_c1 * ib.rows_per_b, ny = (size_t)batch * m->blk_c2 * ob.rows_per_b;
  const float* src = x;
  if (!on_device) { YS_CHECK_HIP(hipMemcpyAsync(m->img_dev, x, nx * 4, hipMemcpyHostToDevice, st)); src = m->img_dev; }
  m->B = batch;
  YS_TRY(join_wgrad_stream(m));
  YS_TRY(ys_pack_input_launch(st, m->dtype, src, batch, m->blk_c1, ib.H, ib.W, ib.ldc, ib.act));
  YS_TRY(forward_impl(m, batch));
  m->have_fwd = m->training;   // only a training-mode forward keeps what backward needs (pre-BN outputs, batch statistics)
  float* dst = on_device ? y : m->out_stage;
  YS_TRY(ys_unpack_nchw_launch(st, m->dtype, ob.act, ob.ldc, 0, batch, m->blk_c2, ob.rows_per_b, dst));
  if (!on_device) { YS_CHECK_HIP(hipMemcpyAsync(y, m->out_stage, ny * 4, hipMemcpyDeviceToHost, st)); YS_CHECK_HIP(hipStreamSynchronize(st)); }
  return YS_OK;
}

int ys_block_backward(ys_model* m, const float* dy, int on_device, float* dx) {
  YS_REQUIRE(m && m->is_block && dy, "ys_block_backward: null argument or not a block handle");
  YS_REQUIRE(m->have_fwd && m->training, "ys_block_backward: needs a training-mode ys_block_forward first");
  YS_CHECK_HIP(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  const int B = m->B;
  const Buf& ib = m->bufs[m->in_buf];
  const Buf& ob = m->bufs[m->blk_out];
  const size_t nx = (size_t)B * m->blk_c1 * ib.rows_per_b, ny = (size_t)B * m->blk_c2 * ob.rows_per_b;
  const float* src = dy;
  if (!on_device) { YS_CHECK_HIP(hipMemcpyAsync(m->out_stage, dy, ny * 4, hipMemcpyHostToDevice, st)); src = m->out_stage; }
  YS_TRY(ys_pack_input_launch(st, m->dtype, src, B, m->blk_c2, ob.H, ob.W, ob.ldc, ob.grad));
  reset_grad_state(m);
  YS_TRY(backward_range(m, 0, ys_model::NSEG - 1));
  if (dx) {
    float* dst = on_device ? dx : m->img_dev;
    YS_TRY(ys_unpack_nchw_launch(st, m->dtype, ib.grad, ib.ldc, 0, B, m->blk_c1, ib.rows_per_b, dst));
    if (!on_device) YS_CHECK_HIP(hipMemcpyAsync(dx, m->img_dev, nx * 4, hipMemcpyDeviceToHost, st));
  }
  YS_CHECK_HIP(hipStreamSynchronize(st));
  return YS_OK;
}


// ---- heads as standalone modules (Head.cs:8-236 Detect, :238-374 Segment, :376-482 Obb, :484-606 Pose): three input buffers + add_detect
int ys_head_create(ys_ctx* ctx, const ys_head_desc* hd, ys_model** out) {
  YS_REQUIRE(ctx && hd && out, "ys_head_create: null argument");
  YS_REQUIRE(hd->dtype == YS_F32 || hd->dtype == YS_BF16 || hd->dtype == YS_FP8, "ys_head_create: dtype %d unsupported", hd->dtype);
  YS_REQUIRE((hd->family == YS_YOLOV8 || hd->family == YS_YOLOV11) && hd->task >= YS_DETECT && hd->task <= YS_POSE, "ys_head_create: family %d task %d", hd->family, hd->task);
  YS_REQUIRE(hd->nc > 0 && hd->reg_max > 1 && hd->reg_max <= 32, "ys_head_create: nc=%d reg_max=%d", hd->nc, hd->reg_max);
  YS_REQUIRE(hd->height > 0 && hd->width > 0 && hd->height % 32 == 0 && hd->width % 32 == 0, "ys_head_create: image size %dx%d must be a positive multiple of 32", hd->height, hd->width);
  YS_REQUIRE(hd->max_batch > 0, "ys_head_create: max_batch %d", hd->max_batch);
  YS_CHECK_HIP(hipSetDevice(ctx->device));
  ys_model* m = new ys_model();
  m->f8 = hd->dtype == YS_FP8;
  const int store = m->f8 ? YS_BF16 : hd->dtype;
  m->ctx = ctx; m->dtype = store; m->epl = store == YS_BF16 ? 8 : 4; m->es = store == YS_BF16 ? 2 : 4;
  m->maxB = hd->max_batch; m->is_head = true;
  m->d = ys_model_desc{}; m->d.family = hd->family; m->d.task = hd->task; m->d.nc = hd->nc; m->d.reg_max = hd->reg_max;
  m->d.height = hd->height; m->d.width = hd->width; m->d.max_batch = hd->max_batch; m->d.dtype = hd->dtype;
  m->d.kpt_num = hd->kpt_num; m->d.kpt_dim = hd->kpt_dim;
  for (int j = 0; j < 64; j++) m->dfl_w[j] = (float)j;
  int st = YS_OK;
  int hh[3], ww[3], ch[3];
  for (int i = 0; i < 3 && st == YS_OK; i++) {
    hh[i] = hd->height / (8 << i); ww[i] = hd->width / (8 << i); ch[i] = hd->ch[i];
    if (ch[i] <= 0 || ch[i] % m->epl) { ys_set_error("ys_head_create: ch[%d] = %d must be a positive multiple of %d for this dtype", i, ch[i], m->epl); st = YS_ERR_UNSUPPORTED; break; }
    m->head_in[i] = new_buf(m, hh[i], ww[i], ch[i]);
    m->head_ch[i] = ch[i];
  }
  m->in_buf = m->head_in[0];
  if (st == YS_OK) st = add_detect(m, "head", m->head_in, ch, hh, ww, hd->family == YS_YOLOV8, 0);
  if (st == YS_OK) st = layout_params(m);
  if (st == YS_OK) {
    for (auto& t : m->tensors) if (t.name.compare(0, 5, "head.") == 0) t.name.erase(0, 5);   // module-relative names (Detect's own state_dict)
    int cst = 1;                                                                            // host staging (allocate: B * max(3, blk_c1) * H * W floats)
    for (int i = 0; i < 3; i++) cst = std::max(cst, (ch[i] + (64 << (2 * i)) - 1) / (64 << (2 * i)));   // ch_i * (H / s_i) * (W / s_i) <= cst * H * W
    m->blk_c1 = cst;
    st = allocate(m);
  }
  if (st == YS_OK) st = ys_model_init_weights(m, 0);
  if (st != YS_OK) { ys_model_destroy(m); return st; }
  *out = m;
  return YS_OK;
}

int ys_head_forward(ys_model* m, const float* const x[3], int on_device, int batch) {
  YS_REQUIRE(m && m->is_head && x && x[0] && x[1] && x[2], "ys_head_forward: null argument or not a head handle");
  YS_REQUIRE(batch > 0 && batch <= m->maxB, "ys_head_forward: batch %d outside (0, %d]", batch, m->maxB);
  YS_CHECK_HIP(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  YsTimer timer(m->ctx, "forward");
  m->B = batch;
  YS_TRY(join_wgrad_stream(m));   // the level inputs are rewritten below; the previous backward's weight-gradient kernels read them
  for (int i = 0; i < 3; i++) {
    const Buf& ib = m->bufs[m->head_in[i]];
    const float* src = x[i];
    if (!on_device) {
      YS_CHECK_HIP(hipMemcpyAsync(m->img_dev, x[i], (size_t)batch * m->head_ch[i] * ib.rows_per_b * 4, hipMemcpyHostToDevice, st));
      src = m->img_dev;
    }
    YS_TRY(ys_pack_input_launch(st, m->dtype, src, batch, m->head_ch[i], ib.H, ib.W, ib.ldc, ib.act));
    if (!on_device) YS_CHECK_HIP(hipStreamSynchronize(st));          // the staging buffer is reused by the next level
  }
  YS_TRY(forward_impl(m, batch));
  m->have_fwd = true; m->fwd_training = m->training; m->have_loss = false; m->have_seg_loss = false;
  return YS_OK;
}

int ys_head_set_grads(ys_model* m, const float* dboxes, const float* dscores, const float* dextra, const float* dproto) {
  YS_REQUIRE(m && m->is_head && dboxes && dscores, "ys_head_set_grads: null argument or not a head handle");
  YS_REQUIRE(m->have_fwd && m->fwd_training, "ys_head_set_grads: needs a training-mode ys_head_forward first");
  YS_REQUIRE(!m->segment || (dextra && dproto), "ys_head_set_grads: a Segment head needs dmask_coefficient and dproto");
  YS_REQUIRE(m->xkind < 2 || dextra, "ys_head_set_grads: an Obb / Pose head needs the gradient of its angle logits / keypoints");
  YS_CHECK_HIP(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  const int B = m->B;
  struct Item { const float* src; int buf; int C; long rows; } items[4] = {
    {dboxes, m->pd_buf, 4 * m->d.reg_max, m->A}, {dscores, m->ps_buf, m->d.nc, m->A},
    {m->segment || m->xkind >= 2 ? dextra : nullptr, m->mc_buf, m->nm, m->A}, {m->segment ? dproto : nullptr, m->pr_buf, m->nm, (long)m->mh * m->mw}};
  for (const Item& it : items) {
    if (!it.src) continue;
    const Buf& b = m->bufs[it.buf];
    const size_t cnt = (size_t)B * it.C * it.rows;
    YS_REQUIRE((long)cnt <= m->n_out_stage, "ys_head_set_grads: staging buffer too small");
    YS_CHECK_HIP(hipMemcpyAsync(m->out_stage, it.src, cnt * 4, hipMemcpyHostToDevice, st));
    YS_TRY(ys_pack_input_launch(st, m->dtype, m->out_stage, B, it.C, 1, (int)it.rows, b.ldc, b.grad));
    YS_CHECK_HIP(hipStreamSynchronize(st));   // out_stage is reused by the next item
  }
  m->have_loss = true; m->have_seg_loss = true;
  return YS_OK;
}

int ys_head_backward(ys_model* m, int on_device, float* const dx[3]) {
  YS_REQUIRE(m && m->is_head, "ys_head_backward: not a head handle");
  YS_REQUIRE(m->have_loss && m->fwd_training, "ys_head_backward: needs a training-mode ys_head_forward and a criterion (ys_loss_*) or ys_head_set_grads first");
  YS_REQUIRE(!m->segment || m->have_seg_loss, "ys_head_backward: the Segment head needs ys_loss_segment (mask gradients)");
  YS_CHECK_HIP(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  YsTimer timer(m->ctx, "backward");
  reset_grad_state(m);
  YS_TRY(backward_range(m, 0, ys_model::NSEG - 1));
  for (int i = 0; i < 3 && dx; i++) {
    if (!dx[i]) continue;
    const Buf& ib = m->bufs[m->head_in[i]];
    const size_t n = (size_t)m->B * m->head_ch[i] * ib.rows_per_b;
    float* dst = on_device ? dx[i] : m->img_dev;
    YS_TRY(ys_unpack_nchw_launch(st, m->dtype, ib.grad, ib.ldc, 0, m->B, m->head_ch[i], ib.rows_per_b, dst));
    if (!on_device) { YS_CHECK_HIP(hipMemcpyAsync(dx[i], m->img_dev, n * 4, hipMemcpyDeviceToHost, st)); YS_CHECK_HIP(hipStreamSynchronize(st)); }
  }
  YS_CHECK_HIP(hipStreamSynchronize(st));
  return YS_OK;
}

}  // extern "C"
