// YoloSharpHip.cs -- P/Invoke declarations for libyolosharp_hip.so (C ABI: include/yolosharp_hip.h).
// Drop this file into YoloSharp (e.g. YoloSharp/Native/YoloSharpHip.cs); INTEGRATION.md section 2 maps the reference's call sites
// (Models/YoloBaseTaskModel.cs:142-160, Models/Detector.cs:27-72, Utils/Amp.cs:260-286, Utils/Ops.cs:239-371) onto these entry points.
// Not compiled in this repository's image (no dotnet); tests/test_abi.py checks that every symbol declared here is exported by the
// library and declared in include/yolosharp_hip.h.
using System;
using System.Runtime.InteropServices;

namespace YoloSharp.Native
{
    [StructLayout(LayoutKind.Sequential)]
    internal struct YsHeadDesc
    {
        public int family;     // 8 = Detect(legacy), 11 = depthwise + 1x1 class tower (Modules/Head.cs:50)
        public int task;       // 0 Detect, 1 Segment, 2 Obb, 3 Pose
        public int nc, reg_max;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public int[] ch;   // channels of P3, P4, P5
        public int height, width;   // input IMAGE size; level i sees [height / s_i, width / s_i], s = 8, 16, 32 (Head.cs:43)
        public int max_batch, dtype, kpt_num, kpt_dim;
    }

    [StructLayout(LayoutKind.Sequential)]
    internal struct YsModelDesc
    {
        public int family;     // 8 = Yolov8 (Models/Yolo.cs:10), 11 = Yolov11
        public int size;       // YoloSize n,s,m,l,x = 0..4 (Types/YoloTypes.cs)
        public int task;       // 0 detect, 1 segment, 2 obb, 3 pose
        public int nc, reg_max, height, width, max_batch;
        public int dtype;      // 0 = Float32 (parity path), 1 = BFloat16 (performance path), 2 = bf16 + fp8 MFMA convolutions
        public int max_labels; // INITIAL ground-truth capacity per image (0 -> 64); host-label loss calls grow it, see ys_model_reserve_labels
        public int kpt_num, kpt_dim; // task 3: Yolov8Pose(kpt_num: 17, kpt_dim: 3) (Models/Yolo.cs:473); 0 -> defaults
    }

    internal static class Hip
    {
        const string Lib = "yolosharp_hip";   // libyolosharp_hip.so next to the assembly / on LD_LIBRARY_PATH

        [DllImport(Lib)] internal static extern IntPtr ys_last_error();
        [DllImport(Lib)] internal static extern int ys_ctx_create(int device, out IntPtr ctx);
        [DllImport(Lib)] internal static extern int ys_ctx_destroy(IntPtr ctx);
        [DllImport(Lib)] internal static extern int ys_ctx_synchronize(IntPtr ctx);

        [DllImport(Lib)] internal static extern int ys_model_create(IntPtr ctx, ref YsModelDesc desc, out IntPtr model);
        [DllImport(Lib)] internal static extern int ys_model_destroy(IntPtr model);
        [DllImport(Lib)] internal static extern int ys_model_num_tensors(IntPtr model);
        [DllImport(Lib, CharSet = CharSet.Ansi)]
        internal static extern int ys_model_tensor_info(IntPtr model, int index, byte[] name, int nameCap,
                                                        out int ndim, [Out] long[] shape4, out int isParam);
        [DllImport(Lib, CharSet = CharSet.Ansi)]
        internal static extern int ys_model_set_tensor(IntPtr model, string name, float[] host, UIntPtr count);
        [DllImport(Lib, CharSet = CharSet.Ansi)]
        internal static extern int ys_model_get_tensor(IntPtr model, string name, [Out] float[] host, UIntPtr count);
        [DllImport(Lib)] internal static extern int ys_model_set_training(IntPtr model, int training);
        [DllImport(Lib)] internal static extern int ys_model_num_anchors(IntPtr model);

        [DllImport(Lib)] internal static extern int ys_model_forward(IntPtr model, float[] imagesNchw, int onDevice, int batch);
        [DllImport(Lib, CharSet = CharSet.Ansi)]
        internal static extern int ys_model_get_output(IntPtr model, string key, [Out] float[] host, UIntPtr count);
        [DllImport(Lib)] internal static extern int ys_model_pred_device(IntPtr model, out IntPtr dptr);

        [DllImport(Lib)] internal static extern int ys_loss_detect(IntPtr model, float[] batchIdx, float[] cls, float[] bboxes, int n, int onDevice);
        [DllImport(Lib)] internal static extern int ys_loss_read(IntPtr model, [Out] float[] items3, out float lossSum);
        // criterion on caller-supplied preds (Loss.cs:411 forward(preds, batch)); device-label callers reserve the per-image label capacity
        [DllImport(Lib)] internal static extern int ys_model_set_preds(IntPtr model, int batch, float[] boxes, float[] scores, float[] maskCoefficient, float[] proto);
        [DllImport(Lib)] internal static extern int ys_model_reserve_labels(IntPtr model, int perImage);
        // 0 = disjoint AdamW groups, 1 = the overlapping groups of YoloBaseTaskModel.cs:144-151 exactly as written
        [DllImport(Lib)] internal static extern int ys_optim_set_param_groups(IntPtr model, int mode);
        // Augment.LetterBox / Augment.Rectangle (Data/Augment.cs:698-857) on the device; uint8 planes (isFloat = 0) or fp32 masks
        [DllImport(Lib)] internal static extern int ys_letterbox(IntPtr ctx, byte[] src, int isFloat, int onDevice, int C, int h, int w, int fitW, int fitH,
                                                                 int outW, int outH, int color, [Out] byte[] dst, out int padL, out int padU);
        // Obb / Pose tasks (Head.Obb / v8OBBLoss, Head.Pose / v8PoseLoss)
        [DllImport(Lib)] internal static extern int ys_loss_obb(IntPtr model, float[] batchIdx, float[] cls, float[] bboxes5, int n, int onDevice);
        [DllImport(Lib)] internal static extern int ys_loss_pose(IntPtr model, float[] batchIdx, float[] cls, float[] bboxes, int n, float[] keypoints, int onDevice);
        // Segment task (Head.Segment / v8SegmentationLoss / Ops.process_mask)
        [DllImport(Lib)] internal static extern int ys_loss_segment(IntPtr model, float[] batchIdx, float[] cls, float[] bboxes, int n,
                                                                    float[] masks /* [B,H/4,W/4] overlap-encoded ids */, int onDevice, int cropMode);
        [DllImport(Lib)] internal static extern int ys_loss_read_items(IntPtr model, [Out] float[] items, int nItems, out float lossSum);
        [DllImport(Lib)] internal static extern int ys_process_mask(IntPtr ctx, float[] protos, float[] masksIn, float[] boxes, int onDevice,
                                                                    int n, int nm, int mh, int mw, int ih, int iw, int upsample, int cropMode,
                                                                    [Out] byte[] outMasks);
        [DllImport(Lib)] internal static extern int ys_model_backward(IntPtr model);
        [DllImport(Lib)] internal static extern int ys_model_zero_grad(IntPtr model);
        [DllImport(Lib)] internal static extern int ys_model_set_overlap(IntPtr model, int on);   // weight-gradient kernels on a second stream (default on)
        [DllImport(Lib)] internal static extern int ys_optim_adamw_step(IntPtr model, float[] lrPerGroup, int nGroups,
                                                                        float beta1, float beta2, float eps, float weightDecay);

        [DllImport(Lib)] internal static extern int ys_mask_iou(IntPtr ctx, float[] gtIds, int nl, byte[] predMasks, int n, int npix, float eps,
                                                                int onDevice, [Out] float[] iou);
        [DllImport(Lib)] internal static extern int ys_match_predictions(IntPtr ctx, float[] predCls, int n, float[] trueCls, int nl, float[] iou,
                                                                         int onDevice, [Out] byte[] correct);
        // per-block handles (Modules.Conv / Bottleneck / C2f / C3k2 / SPPF / C2PSA / Proto as standalone Module<Tensor,Tensor>)
        [StructLayout(LayoutKind.Sequential)] internal struct BlockDesc {
            public int kind, c1, c2, n, shortcut, c3k; public float e; public int k, s, act, height, width, maxBatch, dtype; }
        [DllImport(Lib)] internal static extern int ys_block_create(IntPtr ctx, ref BlockDesc desc, out IntPtr block);
        [DllImport(Lib)] internal static extern int ys_block_output_shape(IntPtr block, [Out] int[] chw);
        [DllImport(Lib)] internal static extern int ys_block_forward(IntPtr block, float[] xNchw, int onDevice, int batch, [Out] float[] yNchw);
        [DllImport(Lib)] internal static extern int ys_block_backward(IntPtr block, float[] dyNchw, int onDevice, [Out] float[] dxNchw);
        // Modules/Head.cs Detect / Segment / Obb / Pose as standalone modules (a ys_model without a backbone: state_dict, set_training,
        // get_output, the ys_loss_* criteria and the optimizer calls work on the handle unchanged)
        [DllImport(Lib)] internal static extern int ys_head_create(IntPtr ctx, ref YsHeadDesc desc, out IntPtr head);
        [DllImport(Lib)] internal static extern int ys_head_forward(IntPtr head, IntPtr[] xNchw3, int onDevice, int batch);   // pinned float[] or device pointers
        [DllImport(Lib)] internal static extern int ys_head_set_grads(IntPtr head, float[] dBoxes, float[] dScores, float[] dExtra, float[] dProto);
        [DllImport(Lib)] internal static extern int ys_head_backward(IntPtr head, int onDevice, IntPtr[] dxNchw3);

        [DllImport(Lib)] internal static extern int ys_nms_batched(IntPtr ctx, float[] pred, int onDevice, int batch, int channels, int anchors,
                                                                   float conf, float iou, int maxDet, int nc, int maxNms, int maxWh,
                                                                   [Out] float[] rows, [Out] long[] keep, [Out] int[] count);
        [DllImport(Lib)] internal static extern int ys_nms_rotated_batched(IntPtr ctx, float[] pred, int onDevice, int batch, int channels, int anchors,
                                                                   float conf, float iou, int maxDet, int nc, int maxNms, int maxWh,
                                                                   [Out] float[] rows, [Out] long[] keep, [Out] int[] count);
        [DllImport(Lib)] internal static extern int ys_probiou(IntPtr ctx, float[] obb1, float[] obb2, int onDevice, int n, int ciou, float eps, float[] output);
        [DllImport(Lib)] internal static extern int ys_batch_probiou(IntPtr ctx, float[] obb1, int n, float[] obb2, int m, int onDevice, float eps, float[] output);
        // data-parallel step without torch (include/yolosharp_hip.h "multi-GPU"): one process per GPU, rank 0 makes the 128-byte RCCL id and
        // ships it over any host channel.  ORDER MATTERS: call ys_dist_init right after ys_ctx_create and BEFORE ys_model_create -- the
        // communicator's streams must exist before the engine creates its weight-gradient stream (otherwise both engine streams can land
        // on one hardware queue and nothing overlaps: 11.6 instead of 10.3 ms/step measured at one rank; ys_dist_init also runs one
        // all-reduce so that RCCL's lazily created resources exist when it returns).
        // routing / tuning options (include/yolosharp_hip.h: process-wide table, seeded from YS_<KEY>=<number> environment variables at load)
        [DllImport(Lib)] internal static extern int ys_set_option([MarshalAs(UnmanagedType.LPStr)] string key, double value);
        [DllImport(Lib)] internal static extern int ys_unset_option([MarshalAs(UnmanagedType.LPStr)] string key);
        [DllImport(Lib)] internal static extern int ys_get_option([MarshalAs(UnmanagedType.LPStr)] string key, out double value, out int isSet);
        [DllImport(Lib)] internal static extern int ys_dist_unique_id([Out] byte[] id128);
        [DllImport(Lib)] internal static extern int ys_dist_init(IntPtr ctx, int rank, int world, byte[] id128);
        [DllImport(Lib)] internal static extern int ys_dist_destroy(IntPtr ctx);
        [DllImport(Lib)] internal static extern int ys_dist_allreduce_grads(IntPtr model, int segment);
        [DllImport(Lib)] internal static extern int ys_dist_wait(IntPtr model);
        [DllImport(Lib)] internal static extern int ys_model_backward_allreduce(IntPtr model);   // segmented backward, each segment's SUM all-reduce overlapped with the next
        // hosts with a collective of their own: asynchronous segment ends + a fence for their communication stream
        [DllImport(Lib)] internal static extern int ys_model_backward_segments(IntPtr model);
        [DllImport(Lib)] internal static extern int ys_model_backward_segment_async(IntPtr model, int seg);
        [DllImport(Lib)] internal static extern int ys_model_segment_fence(IntPtr model, int seg, IntPtr hipStream);
        [DllImport(Lib)] internal static extern int ys_model_segment_grad_range(IntPtr model, int seg, out long offset, out long count);

        internal static void Check(int status)
        {
            if (status == 0) return;
            string msg = Marshal.PtrToStringAnsi(ys_last_error()) ?? "yolosharp_hip error";
            if (status == 1) throw new ArgumentException(msg);   // same exception type as Ops.cs:248-255
            throw new InvalidOperationException($"yolosharp_hip status {status}: {msg}");
        }
    }
}
