/* yolosharp_hip.h -- C ABI of libyolosharp_hip.so: an MI355X (gfx950) native engine for the
 * YOLO forward / loss / backward / AdamW / NMS hot path of IntptrMax/YoloSharp.
 *
 * The reference has no FFI of its own (it is 100 % managed C# on TorchSharp); the "operator API"
 * it exposes is the TorchSharp module surface.  Each entry point below names the reference
 * interface (file:line under /root/reference/YoloSharp) whose arithmetic it replaces.  A C# host
 * binds these with P/Invoke; tests bind them with ctypes.
 *
 * Conventions: extern "C", blittable arguments only, opaque handles, int32 status return
 * (YS_OK == 0), thread-local error text via ys_last_error(), no callbacks, no exceptions across
 * the boundary.  Tensors crossing the edge are plain fp32 in the reference's own layouts
 * (images NCHW, weights OIHW, predictions [B,C,A]); internal layouts (NHWC bf16/f32) never leak.
 * Pointers flagged `on_device` are HIP device pointers that are already resident in HBM;
 * otherwise they are host pointers and the call copies (and, for outputs, synchronises).
 * One HIP stream per ys_ctx; calls on one ctx must be serialised by the caller.
 */
#ifndef YOLOSHARP_HIP_H
#define YOLOSHARP_HIP_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define YS_API __attribute__((visibility("default")))

typedef struct ys_ctx ys_ctx;
typedef struct ys_model ys_model;

enum ys_status {
  YS_OK = 0,
  YS_ERR_INVALID_ARG = 1, /* mirrors the reference's ArgumentException (e.g. Ops.cs:248-255) */
  YS_ERR_HIP = 2,
  YS_ERR_OOM = 3,
  YS_ERR_UNSUPPORTED = 4,
  YS_ERR_STATE = 5
};
/* compute/storage type of activations+weights.  YS_FP8 (BASELINE config 5): bf16 storage, statistics and optimizer with the
 * forward / input-gradient convolutions on the fp8 MFMA path -- e4m3 weights (current per-tensor scale) and activations, e5m2
 * gradients (delayed per-tensor scales), fp32 accumulation; layers whose channel count is not a multiple of 32 stay in bf16. */
enum ys_dtype { YS_F32 = 0, YS_BF16 = 1, YS_FP8 = 2 };
enum ys_family { YS_YOLOV8 = 8, YS_YOLOV11 = 11 };   /* Models/Yolo.cs:10-135, :200-258 */
enum ys_size { YS_N = 0, YS_S = 1, YS_M = 2, YS_L = 3, YS_X = 4 }; /* Types/YoloTypes.cs YoloSize; Yolo.cs:43-51 */
enum ys_task { YS_DETECT = 0, YS_SEGMENT = 1, YS_OBB = 2, YS_POSE = 3 };   /* Config.cs TaskType */

YS_API const char* ys_last_error(void);
YS_API int ys_version(void);
/* 1 when built by hipcc for gfx950, 0 for the test-only CPU interpreter build (never shipped). */
YS_API int ys_is_device_build(void);

/* ---- context: one device + one stream ------------------------------------------------------ */
YS_API int ys_ctx_create(int device, ys_ctx** out);
/* Same, but run on a caller-owned hipStream_t (e.g. torch's current stream) so that RCCL
 * collectives issued by the host order correctly against the engine's kernels. */
YS_API int ys_ctx_create_on_stream(int device, void* hip_stream, ys_ctx** out);
YS_API int ys_ctx_destroy(ys_ctx* ctx);
YS_API int ys_ctx_synchronize(ys_ctx* ctx);
YS_API void* ys_ctx_stream(ys_ctx* ctx);

/* ---- model: replaces Models/Yolo.cs Yolov8 (:10-135) built from Modules/Convs.cs Conv (:36-62),
 *      Modules/Block.cs C2f (:371-399) / Bottleneck (:572-608) / SPPF (:236-285),
 *      Modules/Head.cs Detect (:8-236) ---------------------------------------------------------- */
typedef struct ys_model_desc {
  int32_t family;     /* ys_family */
  int32_t size;       /* ys_size */
  int32_t task;       /* ys_task */
  int32_t nc;         /* number of classes (Config.NumberClass) */
  int32_t reg_max;    /* DFL bins (16) */
  int32_t height;     /* input H (multiple of 32) */
  int32_t width;      /* input W (multiple of 32) */
  int32_t max_batch;  /* capacity; forward may use any B <= max_batch */
  int32_t dtype;      /* ys_dtype: YS_F32 = parity path, YS_BF16 = performance path, YS_FP8 = bf16 + fp8 MFMA convolutions */
  int32_t max_labels; /* initial capacity of ground-truth rows PER IMAGE for the loss (0 = 64).  Not a limit: host-label loss calls
                         grow the workspace to the batch's largest per-image count (the reference pads to counts.max(),
                         Utils/Loss.cs:363-390); device-label callers reserve with ys_model_reserve_labels */
  int32_t kpt_num;    /* YS_POSE: keypoints per object (0 = 17, Models/Yolo.cs:473) */
  int32_t kpt_dim;    /* YS_POSE: 2 = (x, y), 3 = (x, y, visibility) (0 = 3) */
} ys_model_desc;

YS_API int ys_model_create(ys_ctx* ctx, const ys_model_desc* desc, ys_model** out);
YS_API int ys_model_destroy(ys_model* m);

/* state_dict surface (names and order as TorchSharp's named_parameters()+buffers, e.g.
 * "model.0.conv.weight" [16,3,3,3]; Yolo.cs:10-39, YoloBaseTaskModel.cs:31-32,470-490). */
YS_API int ys_model_num_tensors(ys_model* m);
YS_API int ys_model_tensor_info(ys_model* m, int index, char* name, int name_cap,
                                int32_t* ndim, int64_t shape[4], int32_t* is_param);
/* OIHW / [C] fp32 host arrays at the edge. */
YS_API int ys_model_set_tensor(ys_model* m, const char* name, const float* host, size_t count);
YS_API int ys_model_get_tensor(ys_model* m, const char* name, float* host, size_t count);
YS_API int ys_model_get_grad(ys_model* m, const char* name, float* host, size_t count);
/* PyTorch-default initialisation of every tensor (kaiming-uniform(a=sqrt 5) conv weights and
 * biases, BN gamma=1 beta=0, running stats 0/1) from a 64-bit seed (deterministic, host-side). */
YS_API int ys_model_init_weights(ys_model* m, uint64_t seed);
YS_API int ys_model_set_training(ys_model* m, int training);   /* Module.train()/eval() */
YS_API int ys_model_num_anchors(ys_model* m);                  /* A = sum_l (H/s_l)(W/s_l) */
YS_API int64_t ys_model_num_params(ys_model* m);

/* Yolov8.forward (Yolo.cs:92-134).  images: fp32 NCHW [B,3,H,W] in [0,1].
 * training: Detect returns preds only (Head.cs:89-106): "boxes" [B,4*reg_max,A], "scores" [B,nc,A].
 * eval: additionally Detect._inference (Head.cs:204-223): "pred" [B,4+nc,A] (xywh*stride, sigmoid). */
YS_API int ys_model_forward(ys_model* m, const float* images, int on_device, int batch);
/* Predict-side input path (Models/Detector.cs:31-41): uint8 RGB planes [B,3,h,w] (0..255) are padded on the bottom / right
 * with 114 up to the model's (height, width), divided by 255 and packed on the device; then the forward above. */
YS_API int ys_model_forward_u8(ys_model* m, const uint8_t* images, int on_device, int batch, int h, int w);
/* Copy an output to a host fp32 array in the reference layout: key = "boxes" | "scores" | "pred";
 * Segment models (Head.cs:283-313) add "mask_coefficient" [B,nm,A] and "proto" [B,nm,H/4,W/4], and their eval
 * "pred" is [B,4+nc+nm,A] (raw mask coefficients appended).  "dboxes" | "dscores" | "dmask_coefficient" | "dproto"
 * return the loss gradients w.r.t. those outputs after a loss call. */
YS_API int ys_model_get_output(ys_model* m, const char* key, float* host, size_t count);
/* The criterion's `preds` argument supplied by the caller (Utils/Loss.cs:411 `forward(preds, batch)`): host fp32 head outputs in the
 * reference layout -- boxes [B,4*reg_max,A], scores [B,nc,A], and for Segment models mask_coefficient [B,nm,A] + proto
 * [B,nm,H/4,W/4] (NULL otherwise) -- become the engine's head buffers, as if a forward had produced them.  ys_loss_detect /
 * ys_loss_segment and the "dboxes" / "dscores" / "dmask_coefficient" / "dproto" outputs then work on them; ys_model_backward
 * is refused (there is no graph state behind such preds). */
YS_API int ys_model_set_preds(ys_model* m, int batch, const float* boxes, const float* scores,
                              const float* mask_coefficient, const float* proto);
/* Device pointer of the eval prediction [B,4+nc,A] fp32 (input of ys_nms_batched). */
YS_API int ys_model_pred_device(ys_model* m, float** dptr);

/* v8DetectionLoss.forward (Utils/Loss.cs:411-484) + TaskAlignedAssigner (Utils/Tal.cs:13-258)
 * + BboxLoss/DFLoss (Loss.cs:94-167) + bbox_iou CIoU (Utils/Metrics.cs:36-111), and d(sum(loss*B))/d(preds).
 * Labels use the collate contract (Data/YoloDataLoader.cs:18-44): batch_idx[n], cls[n], bboxes[n,4]
 * normalised cxcywh, all fp32.  Asynchronous: results stay on the device until ys_loss_read.
 * Works after a training-mode forward (the step's criterion, Amp.cs:338-348) and after an eval-mode forward (the validation loss
 * on the eval preds, Models/Detector.cs:94-97); only ys_model_backward needs the training-mode forward.
 * Label capacity: with host labels the padded ground-truth workspace grows to the batch's largest per-image label count, like
 * the reference's counts.max() (Loss.cs:376-380).  With device labels (no host sync) an image holding more labels than the
 * workspace makes ys_loss_read* return YS_ERR_INVALID_ARG -- never a silently truncated assignment; reserve first. */
YS_API int ys_loss_detect(ys_model* m, const float* batch_idx, const float* cls, const float* bboxes,
                          int n_labels, int on_device);
/* loss_items[3] = (box, cls, dfl) un-multiplied (the reference's loss_detach), *loss_sum = sum(items)*B
 * (the scalar the reference calls backward() on, Amp.cs:340).  Synchronises the stream. */
YS_API int ys_loss_read(ys_model* m, float loss_items[3], float* loss_sum);
/* Grow the per-image ground-truth capacity of the loss workspace ahead of device-label calls (no-op when already large enough). */
YS_API int ys_model_reserve_labels(ys_model* m, int per_image);

/* v8SegmentationLoss.forward (Utils/Loss.cs:688-863) for task = YS_SEGMENT models: the detection terms and
 * assignment above, then calculate_segmentation_loss / single_mask_loss (:794-863) with Ops.crop_mask
 * (Utils/Ops.cs:409-449) and the gradients w.r.t. "mask_coefficient" and "proto".
 * masks: fp32 [B, H/4, W/4] overlap-encoded instance ids (0 = background, k = the image's k-th label, 1-based;
 * Data/YoloDataset.cs:265-267, overlap_mask = true).  crop_mode 0 = crop_mask's broadcast form (:437-447, the
 * branch an accelerator run takes); 1 = its CPU "n < 50" integer-truncation branch (:421-435). */
YS_API int ys_loss_segment(ys_model* m, const float* batch_idx, const float* cls, const float* bboxes,
                           int n_labels, const float* masks, int on_device, int crop_mode);
/* v8OBBLoss.forward (Utils/Loss.cs:486-684) + RotatedTaskAlignedAssigner (Utils/Tal.cs:260-310) + RotatedBboxLoss
 * (Loss.cs:190-228) for task = YS_OBB models.  bboxes: fp32 [n_labels, 5] = normalised cx, cy, w, h + angle (radians).
 * Labels thinner than 2 pixels are dropped (Loss.cs:563).  Items (box, cls, dfl, angle) via ys_loss_read_items(n_items = 4).
 * The angle head output is kept as the LOGIT: ys_model_get_output("angle") applies (sigmoid - 0.25) * pi (Head.cs:429),
 * "dangle" is the gradient w.r.t. the logit, and ys_model_set_preds takes angle logits [B,1,A] in its mask_coefficient argument. */
YS_API int ys_loss_obb(ys_model* m, const float* batch_idx, const float* cls, const float* bboxes,
                       int n_labels, int on_device);
/* v8PoseLoss.forward (Utils/Loss.cs:870-1071, KeypointLoss :169-188) for task = YS_POSE models: the detection terms and
 * assignment of ys_loss_detect, then the keypoint location (OKS-style, hyp_pose 12) and visibility (BCE, hyp_kobj 1) terms and
 * d(sum(loss*B))/d(raw kpts).  keypoints: fp32 [n_labels, kpt_num, kpt_dim] normalised like bboxes (x, y[, visibility]), row i
 * belongs to label i (YoloDataset keypoints, batch["keypoints"]).  Labels must be grouped by image in collate order (batch_idx
 * non-decreasing), as YoloDataset's collate produces them: the reference's _select_target_keypoints (Loss.cs:1040-1071) indexes
 * the keypoint rows by each label's rank within its image and is only defined for that order; host labels are validated
 * (YS_ERR_INVALID_ARG otherwise), device labels are the caller's responsibility.  A Pose model's ys_model_set_preds takes its raw kpts
 * [B, nk, A] in the mask_coefficient argument; ys_model_get_output("dkpts") returns the gradient. */
YS_API int ys_loss_pose(ys_model* m, const float* batch_idx, const float* cls, const float* bboxes,
                        int n_labels, const float* keypoints, int on_device);
/* Loss items in the criterion's own order: detect n_items = 3 (box, cls, dfl; Loss.cs:414);
 * segment n_items = 5 (box, seg, cls, dfl, semseg = 0; Loss.cs:719); pose n_items = 5 (box, pose, kobj, cls, dfl; Loss.cs:923);
 * obb n_items = 4 (box, cls, dfl, angle; Loss.cs:546).
 * *loss_sum = sum(items)*B. */
YS_API int ys_loss_read_items(ys_model* m, float* loss_items, int n_items, float* loss_sum);

/* autograd backward of sum(loss*B) through the whole graph (Amp.cs:348,370). Gradients accumulate
 * into the flat fp32 gradient buffer like torch's .grad (call ys_model_zero_grad between steps). */
YS_API int ys_model_backward(ys_model* m);
/* Backward in `nseg` consecutive segments (0 = head ... nseg-1 = stem) so the host can launch the
 * RCCL all-reduce of a finished gradient bucket while later segments still run. */
YS_API int ys_model_backward_segments(ys_model* m);
YS_API int ys_model_backward_segment(ys_model* m, int seg);
/* The same segment WITHOUT ordering the context stream behind the weight-gradient stream when it ends (ys_model_backward_segment
 * waits there, which stalls the next segment's kernels for the weight gradients still queued -- measured 1.1 ms of an 11.4 ms
 * data-parallel step on one MI355X).  The segment's gradients are complete once ys_model_segment_fence's events have fired:
 * ys_model_segment_fence(m, seg, stream) makes `stream` (a hipStream_t, e.g. the stream the all-reduce is issued on) wait for them.
 * ys_optim_adamw_step, ys_model_zero_grad, ys_model_get_grad and the next forward order the context stream behind the
 * weight-gradient stream themselves.  No reference counterpart (single-device autograd, Amp.cs:348). */
YS_API int ys_model_backward_segment_async(ys_model* m, int seg);
YS_API int ys_model_segment_fence(ys_model* m, int seg, void* stream);
/* [offset,count) (in floats) of the flat gradient buffer completed by segment `seg`. */
YS_API int ys_model_segment_grad_range(ys_model* m, int seg, int64_t* offset, int64_t* count);
YS_API int ys_model_zero_grad(ys_model* m);
/* Weight-gradient kernels of a backward pass run on a second stream, concurrently with the BatchNorm-backward / input-gradient chain
 * of the following layers (default on; results are identical either way -- every reduction has a fixed order).  on = 0 runs
 * everything on the context stream (per-kernel profiling: co-running kernels time-slice the CUs and inflate each other's durations).
 * No reference counterpart (TorchSharp's autograd engine schedules its own streams). */
YS_API int ys_model_set_overlap(ys_model* m, int on);
/* flat fp32 device buffers (all parameters / all gradients, same order) for collectives. */
YS_API int ys_model_grad_buffer(ys_model* m, float** dptr, int64_t* count);
YS_API int ys_model_param_buffer(ys_model* m, float** dptr, int64_t* count);

/* ---- data-parallel exchange (the reference has no multi-GPU path of its own) ------------------------------
 * One process (and one ys_ctx) per GPU; gradients of a step are SUM-all-reduced over RCCL / xGMI before AdamW.
 * ys_dist_unique_id: rank 0 creates the 128-byte RCCL id and ships it to the other ranks by any host channel.
 * ys_dist_init: joins the communicator (collective) and runs one 4-byte all-reduce so that RCCL's lazily created streams exist when
 * it returns.  CALL IT BEFORE ys_model_create: hardware queues are handed out in stream-creation order, and a communicator built after
 * the model can push the engine's main and weight-gradient streams onto one queue (no overlap; 11.6 vs 10.3 ms/step measured).
 * ys_dist_allreduce_grads(m, seg): asynchronous SUM all-reduce of a
 * backward segment's gradient range (seg < 0: the whole buffer) on a communication stream ordered after the engine stream.
 * ys_dist_wait: the engine stream waits for the outstanding all-reduces.  ys_model_backward_allreduce = the four backward
 * segments with each finished segment's all-reduce overlapped with the next (the loop bench.py runs through torch.distributed).
 * RCCL is dlopen'ed on first use; single-GPU processes never load it. */
/* Tuning / routing options (process-wide; the kernel plans read them at every launch).  Keys are the historical environment names without the YS_ prefix
 * (e.g. "GEMM_MIN_M", "GEMM_HALO", "BNRED"); at library load every YS_<KEY>=<number> environment variable seeds the table, after that only these calls
 * change it.  Production hosts never need them: the defaults are the measured optimum; the test-suite lowers size gates so that oracle-sized shapes reach
 * the wide-layer kernels (tests/conftest.py), the A/B scripts under tools/ flip one switch per run. */
YS_API int ys_set_option(const char* key, double value);
YS_API int ys_unset_option(const char* key);
/* the table's entry for `key` (set by ys_set_option or seeded from the environment at load): *is_set = 0 and *value = 0 when the built-in default applies.
 * Lets a caller change an option temporarily and put back exactly what was there (Engine.options in the Python host). */
YS_API int ys_get_option(const char* key, double* value, int* is_set);

YS_API int ys_dist_unique_id(void* id128);
YS_API int ys_dist_init(ys_ctx* ctx, int rank, int world, const void* id128);
YS_API int ys_dist_destroy(ys_ctx* ctx);
YS_API int ys_dist_allreduce_grads(ys_model* m, int segment);
YS_API int ys_dist_wait(ys_model* m);
YS_API int ys_model_backward_allreduce(ys_model* m);

/* torch.optim.AdamW.step (built YoloBaseTaskModel.cs:144-153, stepped Amp.cs:355-356,371-372):
 * decoupled weight decay, bias-corrected.  Parameter groups follow the reference's name rule:
 * group 0 = names containing "bias", 1 = "weight" (non-BN), 2 = "bn" weights.  lr comes from the host
 * (warm-up / LambdaLR stay in C#, YoloBaseTaskModel.cs:306-319). */
YS_API int ys_optim_adamw_step(ys_model* m, const float* lr_per_group, int ngroups,
                               float beta1, float beta2, float eps, float weight_decay);
/* Parameter-group construction.  mode 0 (default): the name rule above made DISJOINT.  mode 1: the reference's three groups exactly
 * as written (YoloBaseTaskModel.cs:144-151: Contains("bias") | Contains("weight") | Contains("bn")), in which every BatchNorm
 * weight / bias is listed twice: TorchSharp keys optimizer state by parameter, so such a parameter receives two consecutive AdamW
 * updates per step() from one shared state -- first with its first group's learning rate, then with the bn group's, its step
 * counter advancing by two.  Choose before the first optimizer step. */
YS_API int ys_optim_set_param_groups(ys_model* m, int mode);

/* Ops.non_max_suppression (Utils/Ops.cs:239-371) incl. torchvision.ops.nms (:357).
 * pred: [B, C=4+nc+extra, A] fp32 xywh + class probabilities; boxes are converted to xyxy IN PLACE
 * like the reference (:288-291).  out_rows [B,max_det,6+extra] = (x1,y1,x2,y2,conf,cls,extra...),
 * out_keep [B,max_det] = original anchor index, out_count [B].  `on_device` applies to all four
 * pointers.  conf/iou outside [0,1] -> YS_ERR_INVALID_ARG (ArgumentException in the reference). */
YS_API int ys_nms_batched(ys_ctx* ctx, float* pred, int on_device, int batch, int channels, int anchors,
                          float conf_thres, float iou_thres, int max_det, int nc, int max_nms, int max_wh,
                          float* out_rows, int64_t* out_keep, int32_t* out_count);

/* Ops.non_max_suppression(rotated: true) (Utils/Ops.cs:286,349-353): oriented boxes.  pred [B, 4+nc+extra, A] with the angle
 * (radians) as the LAST channel; boxes stay xywh (no in-place conversion); candidates are ordered by score and candidate j is
 * dropped iff any earlier candidate i has Metrics.batch_probiou(i, j) >= iou_thres (Ops.nms_rotated, :373-401 -- not the greedy
 * rule of ys_nms_batched).  Outputs as ys_nms_batched with rows (x, y, w, h, conf, cls, extra..., angle). */
YS_API int ys_nms_rotated_batched(ys_ctx* ctx, float* pred, int on_device, int batch, int channels, int anchors,
                                  float conf_thres, float iou_thres, int max_det, int nc, int max_nms, int max_wh,
                                  float* out_rows, int64_t* out_keep, int32_t* out_count);

/* Metrics.probiou (Utils/Metrics.cs:137-177): probabilistic IoU of n pairs of oriented boxes xywhr [n,5] -> out [n]; ciou != 0 adds
 * the aspect-ratio term of the reference's CIoU option.  Metrics.batch_probiou (:223-258): all pairs, out [n, m].  eps: the
 * reference's 1e-7.  `on_device` applies to all pointers. */
YS_API int ys_probiou(ys_ctx* ctx, const float* obb1, const float* obb2, int on_device, int n, int ciou, float eps, float* out);
YS_API int ys_batch_probiou(ys_ctx* ctx, const float* obb1, int n, const float* obb2, int m, int on_device, float eps, float* out);

/* Validation, per-image part of Detector.Val (Models/Detector.cs:103-120), batched on the device:
 * GT boxes = bboxes[batch_idx == b] * (W,H,W,H) -> xyxy (Utils/Ops.cs:68-81); iou = Metrics.box_iou(gt, pred[:,0:4])
 * (Utils/Metrics.cs:16-34); correct = match_predictions(pred[:,5], cls, iou) (Models/YoloBaseTaskModel.cs:377-446) for the
 * IoU thresholds linspace(0.5, 0.95, 10).  rows [B,max_det,row_stride] / count [B] are ys_nms_batched outputs;
 * correct: uint8 [B, max_det, 10].  `on_device` applies to rows, count, the label arrays and correct. */
YS_API int ys_val_match_batched(ys_ctx* ctx, const float* rows, const int32_t* count, int on_device, int batch, int max_det,
                                int row_stride, const float* batch_idx, const float* cls, const float* bboxes, int n_labels,
                                float img_w, float img_h, uint8_t* correct);
/* Metrics.box_iou (Utils/Metrics.cs:16-34): iou [n, m] of xyxy boxes, fp32, eps as given (reference default 1e-7). */
YS_API int ys_box_iou(ys_ctx* ctx, const float* box1, int n, const float* box2, int m, float eps, int on_device, float* iou);

/* Metrics.mask_iou (Utils/Metrics.cs:120-125) as Segmenter.Val uses it (Models/Segmenter.cs:131-143): mask1[k] = (gt_ids == k+1),
 * k < nl, from the image's overlap-encoded instance-id map [npix] (fp32 ids, YoloDataset.cs:265-267); mask2 = pred_masks uint8
 * [n, npix] (ys_process_mask output) -> iou fp32 [nl, n].  Bit-exact: the reference's float matmul of 0/1 masks is an integer count. */
YS_API int ys_mask_iou(ys_ctx* ctx, const float* gt_ids, int nl, const uint8_t* pred_masks, int n, int npix, float eps,
                       int on_device, float* iou);
/* Metrics.kpt_iou (Utils/Metrics.cs:186-212) as PoseDetector.Val uses it (Models/PoseDetector.cs:150-158): object keypoint similarity
 * [n, m] of ground-truth keypoints kpt1 [n, K, 3] (pixels, visibility; 2-D labels get a visibility column of ones on the host,
 * PoseDetector.cs:144-148) against predicted keypoints kpt2 [m, K, kpt_dim]; area [n] = w * h * 0.53 of the labels' boxes;
 * sigmas = the COCO OKS sigmas when K == 17, else 1/K. */
YS_API int ys_kpt_iou(ys_ctx* ctx, const float* kpt1, int n, const float* kpt2, int m, const float* area, int kpt_num, int kpt_dim,
                      float eps, int on_device, float* iou);
/* match_predictions (Models/YoloBaseTaskModel.cs:377-446) on a caller-supplied IoU matrix [nl, n] (box, mask, OBB or keypoint
 * IoU alike): correct uint8 [n, 10] for the thresholds linspace(0.5, 0.95, 10). */
YS_API int ys_match_predictions(ys_ctx* ctx, const float* pred_cls, int n, const float* true_cls, int nl, const float* iou,
                                int on_device, uint8_t* correct);

/* Ops.process_mask (Utils/Ops.cs:462-489), used by Segmenter post-processing (Models/Segmenter.cs:131-160 region):
 * masks = masks_in[n,nm] @ protos[nm,mh,mw], cropped to boxes (xyxy, image pixels) scaled to the mask grid,
 * optionally bilinearly upsampled (align_corners = false) to (ih, iw), thresholded > 0.
 * out: uint8 [n, oh, ow], (oh, ow) = upsample ? (ih, iw) : (mh, mw).  `on_device` applies to all pointers. */
YS_API int ys_process_mask(ys_ctx* ctx, const float* protos, const float* masks_in, const float* boxes, int on_device,
                           int n, int nm, int mh, int mw, int ih, int iw, int upsample, int crop_mode, uint8_t* out);

/* Input side: Augment.LetterBox.LetterboxImage (Data/Augment.cs:757-778) and Augment.Rectangle.RectangleImage
 * (:836-857): ratio = min(fit_w / w, fit_h / h), new = (int)(size * ratio), torchvision resize (TorchVision.NET default: nearest) and
 * constant padding with `color` (114 for images, 0 for masks), centred on an out_w x out_h canvas.  LetterBox: fit = out;
 * Rectangle: fit = the label's resized shape, out = its rectangle shape.  Planes [C,h,w] -> [C,out_h,out_w], uint8 or fp32
 * (is_float).  *pad_l / *pad_u = the offsets the reference adds to boxes / keypoints / OBB corners. */
YS_API int ys_letterbox(ys_ctx* ctx, const void* src, int is_float, int on_device, int C, int h, int w, int fit_w, int fit_h,
                        int out_w, int out_h, int color, void* dst, int32_t* pad_l, int32_t* pad_u);

/* ---- per-operator entry points (unit parity; a TorchSharp-free C# Conv wrapper) ---------------
 * Convs.Conv.forward (Convs.cs:36-62): y = act(BN(conv2d(x))) on fp32 NCHW / OIHW HOST arrays at the
 * edge (the call stages them through HBM).  training != 0 uses batch statistics and updates running stats (momentum 0.03, eps 1e-3). */
YS_API int ys_conv_bn_act_fwd(ys_ctx* ctx, int dtype, const float* x_nchw, int B, int Cin, int H, int W,
                              const float* w_oihw, int Cout, int k, int stride,
                              const float* bn_gamma, const float* bn_beta, float* bn_mean, float* bn_var,
                              const float* bias, int act_silu, int training, float* y_nchw);

/* autograd of the same conv2d (Amp.cs:348,370): given dy [B,Cout,Ho,Wo] returns dx [B,Cin,H,W] (optional, may be
 * NULL) and dw [Cout,Cin,k,k]; all fp32 host arrays. */
YS_API int ys_conv_bwd(ys_ctx* ctx, int dtype, const float* x_nchw, int B, int Cin, int H, int W,
                       const float* w_oihw, int Cout, int k, int stride, const float* dy_nchw,
                       float* dx_nchw, float* dw_oihw);

/* ---- per-block entry points: a TorchSharp-free body for the reference's block modules, one stateful handle per
 *      module instance (SURVEY 8b "ys_c2f_fwd/bwd, ys_c3k2_fwd/bwd, ys_sppf_fwd/bwd, ys_proto_fwd/bwd").  The handle IS a
 *      ys_model: ys_model_num_tensors / tensor_info / set_tensor / get_tensor / get_grad / init_weights / set_training /
 *      zero_grad / optim_adamw_step / destroy all apply, with the module-relative state_dict names TorchSharp gives a
 *      standalone module ("cv1.conv.weight", "m.0.cv2.bn.running_var", "upsample.bias", ...).
 *        YS_BLOCK_CONV        Convs.Conv        (Convs.cs:36-62)    c1->c2, k in {1,3}, s in {1,2}, act
 *        YS_BLOCK_BOTTLENECK  Block.Bottleneck  (Block.cs:572-608)  c1 == c2, 3x3/3x3, hidden int(c2*e), shortcut
 *        YS_BLOCK_C2F         Block.C2f         (Block.cs:371-399)  n Bottlenecks (e = 1.0), shortcut
 *        YS_BLOCK_C3K2        Block.C3k2        (Block.cs:623-662)  n x (C3k(c,c,2) if c3k else Bottleneck(c,c)), e, shortcut = true
 *        YS_BLOCK_SPPF        Block.SPPF        (Block.cs:236-285)  c1 == c2
 *        YS_BLOCK_C2PSA       Block.C2PSA       (Block.cs:664-810)  c1 == c2, (c1/2) % 64 == 0, n PSABlocks
 *        YS_BLOCK_PROTO       Block.Proto       (Block.cs:51-84)    c1 -> n (= c_, hidden) -> c2 masks, output 2H x 2W
 *      Channel counts (c1, c2 and every hidden width) must be multiples of 4 (f32) / 8 (bf16): the widths the YOLO graphs use. */
typedef enum ys_block_kind {
  YS_BLOCK_CONV = 0, YS_BLOCK_BOTTLENECK = 1, YS_BLOCK_C2F = 2, YS_BLOCK_C3K2 = 3, YS_BLOCK_SPPF = 4, YS_BLOCK_C2PSA = 5,
  YS_BLOCK_PROTO = 6
} ys_block_kind;

typedef struct ys_block_desc {
  int32_t kind;       /* ys_block_kind */
  int32_t c1, c2;     /* input / output channels */
  int32_t n;          /* repeats (C2f, C3k2, C2PSA); hidden width c_ (Proto) */
  int32_t shortcut;   /* Bottleneck / C2f */
  int32_t c3k;        /* C3k2: use C3k inner blocks */
  float e;            /* expansion (Bottleneck, C3k2); 0 = the module's default (0.5) */
  int32_t k, s, act;  /* Conv */
  int32_t height;     /* input H */
  int32_t width;      /* input W */
  int32_t max_batch;
  int32_t dtype;      /* ys_dtype */
} ys_block_desc;

YS_API int ys_block_create(ys_ctx* ctx, const ys_block_desc* desc, ys_model** out);
/* output geometry of the block: [c, h, w] */
YS_API int ys_block_output_shape(ys_model* block, int32_t shape_chw[3]);
/* y = module.forward(x): x fp32 NCHW [batch,c1,H,W], y fp32 NCHW [batch,c2,Ho,Wo]; host arrays (on_device = 0) or device
 * pointers (1).  Training mode uses batch statistics and updates the running ones, exactly like the full model. */
YS_API int ys_block_forward(ys_model* block, const float* x_nchw, int on_device, int batch, float* y_nchw);
/* autograd of the last training-mode forward: given dy (shape of y) accumulates parameter gradients (read with
 * ys_model_get_grad) and writes dx (shape of x; may be NULL). */
YS_API int ys_block_backward(ys_model* block, const float* dy_nchw, int on_device, float* dx_nchw);

/* ---- the heads as standalone modules (Detect / Segment handles; round 3).
 * Modules/Head.cs:8-236 Detect, :238-374 Segment, :376-482 Obb, :484-606 Pose: the three neck feature maps in, the criterion's
 * `preds` (training) / the decoded predictions (eval) out.  The handle is a ys_model without a backbone: the state_dict surface
 * (module-relative names "cv2.0.0.conv.weight", "cv3...", "dfl.conv.weight", "proto...", "cv4..."; Head.cs registration order),
 * ys_model_set_training, ys_model_get_output ("boxes", "scores", "pred", "mask_coefficient", "proto", "kpts", "angle" and their
 * "d..." gradients), ys_loss_detect / _segment / _obb / _pose, ys_model_zero_grad and the optimizer calls work on it unchanged. */
typedef struct ys_head_desc {
  int32_t family;        /* ys_family: YS_YOLOV8 = Detect(legacy: 3x3 cls tower), YS_YOLOV11 = depthwise + 1x1 cls tower (Head.cs:50) */
  int32_t task;          /* ys_task: which head */
  int32_t nc, reg_max;
  int32_t ch[3];         /* channels of P3, P4, P5 (multiples of 4 for f32, 8 for bf16) */
  int32_t height, width; /* INPUT IMAGE size (multiple of 32): level i sees [height / s_i, width / s_i], s = 8, 16, 32 (Head.cs:43) */
  int32_t max_batch, dtype;
  int32_t kpt_num, kpt_dim;   /* YS_POSE (0 = 17 x 3) */
} ys_head_desc;
YS_API int ys_head_create(ys_ctx* ctx, const ys_head_desc* desc, ys_model** out);
/* x[i]: fp32 NCHW [batch, ch[i], height / s_i, width / s_i], host arrays (on_device = 0) or device pointers (1) */
YS_API int ys_head_forward(ys_model* head, const float* const x[3], int on_device, int batch);
/* Gradients of the head outputs supplied by the caller instead of a ys_loss_* call: [B, 4*reg_max, A], [B, nc, A], the task's
 * extra output ([B, nm | 1 | nk, A]; NULL for Detect) and the prototypes ([B, nm, mh, mw]; Segment only); fp32 host arrays. */
YS_API int ys_head_set_grads(ys_model* head, const float* dboxes, const float* dscores, const float* dextra, const float* dproto);
/* autograd of the last training-mode ys_head_forward from the gradients the criterion (or ys_head_set_grads) left on the head outputs:
 * accumulates the parameter gradients and writes dx[i] (shape of x[i]; entries / the array may be NULL). */
YS_API int ys_head_backward(ys_model* head, int on_device, float* const dx[3]);

/* device memory helpers for hosts without a HIP binding of their own */
YS_API int ys_device_malloc(ys_ctx* ctx, size_t bytes, void** dptr);
YS_API int ys_device_free(ys_ctx* ctx, void* dptr);
YS_API int ys_memcpy_h2d(ys_ctx* ctx, void* dst, const void* src, size_t bytes);
YS_API int ys_memcpy_d2h(ys_ctx* ctx, void* dst, const void* src, size_t bytes);

/* timing of the most recent calls on this ctx's stream, measured with HIP events on that stream:
 * name = "forward" | "loss" | "backward" | "optim" | "nms"; returns ms of the last completed call. */
YS_API int ys_ctx_profile_enable(ys_ctx* ctx, int enable);
YS_API int ys_ctx_last_ms(ys_ctx* ctx, const char* name, float* ms);

/* per-kernel-class timing (HIP events on this ctx's stream around every launch of a class, e.g. "conv_igemm",
 * "conv_wgrad", "bn_act", "nms"); enable for a few UNTIMED steps, then read the launch count and summed duration. */
YS_API int ys_ctx_kernel_profile(ys_ctx* ctx, int enable);
YS_API int ys_ctx_kernel_profile_read(ys_ctx* ctx, const char* name, int32_t* launches, float* total_ms);
/* per-launch CSV (class,label,us) of everything recorded since profiling was enabled (triage tool) */
YS_API int ys_ctx_kernel_profile_dump(ys_ctx* ctx, const char* path);

#ifdef __cplusplus
}
#endif
#endif /* YOLOSHARP_HIP_H */
