// probe_f8.hip -- measured semantics of the gfx950 fp8 path used by the fp8 convolution kernels:
//   * v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales (E8M0 = 127): operand lane map and result layout
//   * v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32: rounding, overflow and subnormal behaviour (OCP e4m3fn / e5m2)
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 tools/probe/probe_f8.hip -o /tmp/probe_f8 && /tmp/probe_f8
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const unsigned char* A, const unsigned char* B, float* D, int fmt_a, int assume) {
  // A: [16 rows][128 k] bytes, B: [16 cols][128 k] bytes.  Assumed map: lane (li = l & 15, q = l >> 4) holds k = 32q .. 32q+31.
  const int l = threadIdx.x, li = l & 15, q = l >> 4;
  v8i a, b;
  for (int w = 0; w < 8; w++) {
    unsigned wa = 0, wb = 0;
    for (int t = 0; t < 4; t++) {
      const int kk = assume == 0 ? 32 * q + 4 * w + t : (w < 4 ? 16 * q + 4 * w + t : 64 + 16 * q + 4 * (w - 4) + t);
      wa |= (unsigned)A[li * 128 + kk] << (8 * t);
      wb |= (unsigned)B[li * 128 + kk] << (8 * t);
    }
    a[w] = (int)wa; b[w] = (int)wb;
  }
  v4f c = {0.f, 0.f, 0.f, 0.f};
  if (fmt_a == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
  else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 1, 0, 0, 127, 0, 127);   // A = bf8 (e5m2), B = fp8
  for (int r = 0; r < 4; r++) D[(4 * q + r) * 16 + li] = c[r];     // assumed: D[row 4q+r][col li], rows <- A lanes, cols <- B lanes
}

__global__ void k_cvt(const float* x, int n, unsigned char* o8, unsigned char* o5, float* back8, float* back5) {
  const int i = threadIdx.x;
  if (i >= n) return;
  unsigned w = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], 0.f, 0u, false);
  unsigned v = __builtin_amdgcn_cvt_pk_bf8_f32(x[i], 0.f, 0u, false);
  o8[i] = (unsigned char)(w & 255u); o5[i] = (unsigned char)(v & 255u);
  back8[i] = __builtin_amdgcn_cvt_f32_fp8((int)w, 0);
  back5[i] = __builtin_amdgcn_cvt_f32_bf8((int)v, 0);
}

static float e4m3_to_f(unsigned char b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? ldexpf((float)m, -9) : (e == 15 && m == 7 ? NAN : ldexpf(1.f + m / 8.f, e - 7));
  return s ? -v : v;
}
static float e5m2_to_f(unsigned char b) {
  const int s = b >> 7, e = (b >> 2) & 31, m = b & 3;
  float v = e == 0 ? ldexpf((float)m, -16) : (e == 31 ? (m ? NAN : INFINITY) : ldexpf(1.f + m / 4.f, e - 15));
  return s ? -v : v;
}

int main() {
  unsigned char hA[16 * 128], hB[16 * 128], *dA, *dB;
  float hD[256], *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  srand(1);
  for (int fmt = 0; fmt < 2; fmt++)
    for (int assume = 0; assume < 2; assume++) {
      // small exactly representable values; asymmetric patterns so that row / column swaps and K permutations show up
      static const unsigned char v8[8] = {0x00, 0x38, 0x40, 0x44, 0xB8, 0xC0, 0x30, 0x48};   // e4m3: 0, 1, 2, 3, -1, -2, .5, 4
      static const unsigned char v5[8] = {0x00, 0x3C, 0x40, 0x42, 0xBC, 0xC0, 0x38, 0x44};   // e5m2: 0, 1, 2, 3, -1, -2, .5, 4
      for (int i = 0; i < 16 * 128; i++) { hA[i] = (fmt ? v5 : v8)[rand() & 7]; hB[i] = v8[rand() & 7]; }
      hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
      k_mfma<<<1, 64>>>(dA, dB, dD, fmt, assume);
      hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
      double worst = 0;
      for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
          double ref = 0;
          for (int k = 0; k < 128; k++) ref += (double)(fmt ? e5m2_to_f(hA[i * 128 + k]) : e4m3_to_f(hA[i * 128 + k])) * e4m3_to_f(hB[j * 128 + k]);
          worst = fmax(worst, fabs(ref - hD[i * 16 + j]));
        }
      printf("mfma_scale 16x16x128 A=%s B=fp8, K map %s: max |D - ref| = %g %s\n", fmt ? "bf8" : "fp8",
             assume == 0 ? "lane q holds k=32q..32q+31" : "lane q holds k=16q..+15 and 64+16q..+15", worst, worst == 0 ? "(exact)" : "");
    }
  const float xs[] = {0.f, 1.f, 1.0625f, 1.125f, 1.1875f, 3.9f, 447.f, 448.f, 464.f, 480.f, 500.f, 1e6f, -1e6f, INFINITY, NAN, 0.0146f,
                      0.001953125f, 0.0009765625f, 0.0012f, 57344.f, 60000.f, 65536.f, 1e-5f, 7.6e-6f, -0.3f, 0.017f};
  const int n = sizeof xs / sizeof xs[0];
  float *dx, *db8, *db5, hb8[64], hb5[64]; unsigned char *d8, *d5, h8[64], h5[64];
  hipMalloc(&dx, sizeof xs); hipMalloc(&d8, 64); hipMalloc(&d5, 64); hipMalloc(&db8, 256); hipMalloc(&db5, 256);
  hipMemcpy(dx, xs, sizeof xs, hipMemcpyHostToDevice);
  k_cvt<<<1, 64>>>(dx, n, d8, d5, db8, db5);
  hipMemcpy(h8, d8, 64, hipMemcpyDeviceToHost); hipMemcpy(h5, d5, 64, hipMemcpyDeviceToHost);
  hipMemcpy(hb8, db8, 256, hipMemcpyDeviceToHost); hipMemcpy(hb5, db5, 256, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; i++)
    printf("x=%-14g fp8 0x%02X -> %-12g (host decode %-12g) | bf8 0x%02X -> %-12g (host decode %g)\n", xs[i], h8[i], hb8[i], e4m3_to_f(h8[i]), h5[i], hb5[i], e5m2_to_f(h5[i]));
  return 0;
}
