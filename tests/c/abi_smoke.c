/* abi_smoke.c -- a compiled, non-Python consumer of include/yolosharp_hip.h ("C/C++ test binaries").
 *
 * Plain C11, no HIP / torch headers: everything it knows about the engine comes from the public header, so struct layouts,
 * int64_t shape arrays, enum values and the pointer / size conventions are checked by a real compiler, the way a C# P/Invoke
 * or cgo binding would consume them.  One YOLOv8n pass of the whole hot path at 64x64, B = 2:
 *   ctx -> model -> state_dict listing -> train forward -> v8DetectionLoss -> backward -> AdamW -> zero_grad
 *       -> eval forward -> pred -> ys_nms_batched (host buffers) -> destroy
 * Exit codes: 0 ok, 77 no device / library cannot create a context (skip), anything else = failure (message on stderr).
 * Built by yolosharp_amd/build.py against libyolosharp_hip.so (GPU) and against the test-only interpreter build (CPU CI). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "yolosharp_hip.h"

#define CHECK(call) do { int st_ = (call); if (st_ != YS_OK) { fprintf(stderr, "%s -> status %d: %s\n", #call, st_, ys_last_error()); return 1; } } while (0)

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xFFFFFF) / 16777216.0f; }

int main(void) {
  enum { B = 2, H = 64, W = 64, NC = 80, MAXDET = 300 };
  ys_ctx* ctx = NULL;
  if (ys_ctx_create(0, &ctx) != YS_OK) { fprintf(stderr, "abi_smoke: no context (%s) -- skipped\n", ys_last_error()); return 77; }
  ys_model_desc d;
  memset(&d, 0, sizeof d);
  d.family = YS_YOLOV8; d.size = YS_N; d.task = YS_DETECT; d.nc = NC; d.reg_max = 16; d.height = H; d.width = W; d.max_batch = B;
  d.dtype = YS_F32; d.max_labels = 0;
  ys_model* m = NULL;
  CHECK(ys_model_create(ctx, &d, &m));
  CHECK(ys_model_init_weights(m, 7));
  const int A = ys_model_num_anchors(m);
  if (A != 8 * 8 + 4 * 4 + 2 * 2) { fprintf(stderr, "anchors %d\n", A); return 1; }
  /* state_dict surface through the blittable listing */
  const int nt = ys_model_num_tensors(m);
  int64_t total = 0;
  for (int i = 0; i < nt; i++) {
    char name[128]; int32_t nd = 0, isp = 0; int64_t shape[4] = {0, 0, 0, 0};
    CHECK(ys_model_tensor_info(m, i, name, (int)sizeof name, &nd, shape, &isp));
    int64_t n = 1;
    for (int k = 0; k < nd; k++) n *= shape[k];
    if (i == 0 && (strcmp(name, "model.0.conv.weight") != 0 || nd != 4 || shape[0] != 16 || shape[1] != 3 || shape[2] != 3 || shape[3] != 3)) {
      fprintf(stderr, "tensor 0 is %s nd=%d [%lld,%lld,%lld,%lld]\n", name, nd, (long long)shape[0], (long long)shape[1], (long long)shape[2], (long long)shape[3]);
      return 1;
    }
    if (isp && strstr(name, "dfl") == NULL) total += n;
  }
  if (total != ys_model_num_params(m) || total != 3157184) { fprintf(stderr, "parameter count %lld vs %lld\n", (long long)total, (long long)ys_model_num_params(m)); return 1; }
  /* one training step */
  unsigned seed = 12345u;
  float* img = (float*)malloc(sizeof(float) * B * 3 * H * W);
  for (int i = 0; i < B * 3 * H * W; i++) img[i] = frand(&seed);
  const float bidx[3] = {0.f, 0.f, 1.f}, cls[3] = {3.f, 17.f, 42.f};
  const float box[12] = {0.5f, 0.5f, 0.5f, 0.4f, 0.3f, 0.6f, 0.2f, 0.3f, 0.6f, 0.4f, 0.5f, 0.6f};
  CHECK(ys_model_set_training(m, 1));
  CHECK(ys_model_forward(m, img, 0, B));
  CHECK(ys_loss_detect(m, bidx, cls, box, 3, 0));
  float items[3] = {0, 0, 0}, lsum = 0.f;
  CHECK(ys_loss_read(m, items, &lsum));
  if (!(isfinite(items[0]) && isfinite(items[1]) && isfinite(items[2]) && items[1] > 0.f) || fabsf(lsum - (items[0] + items[1] + items[2]) * B) > 1e-3f * fabsf(lsum)) {
    fprintf(stderr, "loss items %g %g %g sum %g\n", items[0], items[1], items[2], lsum); return 1;
  }
  CHECK(ys_model_zero_grad(m));
  CHECK(ys_model_backward(m));
  float g0[16 * 3 * 3 * 3];
  CHECK(ys_model_get_grad(m, "model.0.conv.weight", g0, sizeof g0 / sizeof g0[0]));
  double gn = 0.0;
  for (size_t i = 0; i < sizeof g0 / sizeof g0[0]; i++) gn += (double)g0[i] * g0[i];
  if (!(gn > 0.0) || !isfinite(gn)) { fprintf(stderr, "stem gradient norm %g\n", gn); return 1; }
  const float lrs[3] = {1e-3f, 1e-3f, 1e-3f};
  float w_before[16 * 3 * 3 * 3], w_after[16 * 3 * 3 * 3];
  CHECK(ys_model_get_tensor(m, "model.0.conv.weight", w_before, 432));
  CHECK(ys_optim_adamw_step(m, lrs, 3, 0.9f, 0.999f, 1e-8f, 5e-4f));
  CHECK(ys_model_zero_grad(m));
  CHECK(ys_model_get_tensor(m, "model.0.conv.weight", w_after, 432));
  if (memcmp(w_before, w_after, sizeof w_before) == 0) { fprintf(stderr, "AdamW did not move the weights\n"); return 1; }
  /* misuse is reported, not ignored */
  if (ys_model_get_tensor(m, "no.such.tensor", w_after, 1) != YS_ERR_INVALID_ARG) { fprintf(stderr, "unknown tensor accepted\n"); return 1; }
  /* eval forward + NMS on host buffers (int64_t keep indices, int32_t counts) */
  CHECK(ys_model_set_training(m, 0));
  CHECK(ys_model_forward(m, img, 0, B));
  const int Cc = 4 + NC;
  float* pred = (float*)malloc(sizeof(float) * B * Cc * A);
  CHECK(ys_model_get_output(m, "pred", pred, (size_t)B * Cc * A));
  float* rows = (float*)calloc((size_t)B * MAXDET * 6, sizeof(float));
  int64_t* keep = (int64_t*)calloc((size_t)B * MAXDET, sizeof(int64_t));
  int32_t cnt[B];
  CHECK(ys_nms_batched(ctx, pred, 0, B, Cc, A, 0.001f, 0.7f, MAXDET, 0, 30000, 7680, rows, keep, cnt));
  for (int b = 0; b < B; b++) {
    if (cnt[b] <= 0 || cnt[b] > MAXDET) { fprintf(stderr, "image %d: %d detections\n", b, cnt[b]); return 1; }
    for (int i = 0; i < cnt[b]; i++) {
      const int64_t k = keep[(size_t)b * MAXDET + i];
      const float* r = rows + ((size_t)b * MAXDET + i) * 6;
      if (k < 0 || k >= A || !(r[2] >= r[0]) || !(r[4] > 0.001f) || r[5] < 0.f || r[5] >= (float)NC) { fprintf(stderr, "bad row %d/%d\n", b, i); return 1; }
      if (i > 0 && r[4] > r[-6 + 4]) { fprintf(stderr, "rows not in descending confidence\n"); return 1; }
    }
  }
  if (ys_nms_batched(ctx, pred, 0, B, Cc, A, 1.5f, 0.7f, MAXDET, 0, 30000, 7680, rows, keep, cnt) != YS_ERR_INVALID_ARG) {
    fprintf(stderr, "conf 1.5 accepted (the reference throws ArgumentException, Ops.cs:248-255)\n"); return 1;
  }
  printf("abi_smoke OK: device_build=%d tensors=%d params=%lld loss=(%.4f %.4f %.4f) kept=(%d %d)\n", ys_is_device_build(), nt,
         (long long)total, items[0], items[1], items[2], cnt[0], cnt[1]);
  free(img); free(pred); free(rows); free(keep);
  CHECK(ys_model_destroy(m));
  CHECK(ys_ctx_destroy(ctx));
  return 0;
}
