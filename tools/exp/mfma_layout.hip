// Probe the lane layouts of v_mfma_f64_4x4x4 and v_mfma_f64_16x16x4 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void probe(double *out) {
  const int l = threadIdx.x;
  // (1) a = 2^lane-ish unique tags are too big; use a = lane, b = 1 -> D = sum over k of A
  double d1 = __builtin_amdgcn_mfma_f64_4x4x4f64((double)l, 1.0, 0.0, 0, 0, 0);
  // (2) a = 1, b = lane -> D = sum over k of B
  double d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, (double)l, 0.0, 0, 0, 0);
  // (3) chain: A := d1, B = 1
  double d3 = __builtin_amdgcn_mfma_f64_4x4x4f64(d1, 1.0, 0.0, 0, 0, 0);
  // (4) chain: A = 1, B := d1
  double d4 = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, d1, 0.0, 0, 0, 0);
  v4d z = {0, 0, 0, 0};
  v4d e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, (double)l, z, 0, 0, 0);  // A=1: sum_k B[k][j]
  v4d e2 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)l, 1.0, z, 0, 0, 0);  // B=1: sum_k A[i][k]
  out[l * 12 + 0] = d1; out[l * 12 + 1] = d2; out[l * 12 + 2] = d3; out[l * 12 + 3] = d4;
  for (int r = 0; r < 4; ++r) { out[l * 12 + 4 + r] = e1[r]; out[l * 12 + 8 + r] = e2[r]; }
}
__global__ void timing(double *out, int iters) {
  const int l = threadIdx.x;
  double a = l * 0.001;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    double p = __builtin_amdgcn_mfma_f64_4x4x4f64(a, 1.0, 0.0, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f64_4x4x4f64(p, 1.0, 0.0, 0, 0, 0) * 1e-3 + 0.5;
  }
  long long t1 = __builtin_readcyclecounter();
  v4d z = {0, 0, 0, 0};
  double b = l * 0.001;
  for (int i = 0; i < iters; ++i) {
    v4d e = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, b, z, 0, 0, 0);
    b = e[0] * 1e-3 + 0.5;
  }
  long long t2 = __builtin_readcyclecounter();
  if (l == 0) { out[0] = double(t1 - t0) / iters; out[1] = double(t2 - t1) / iters; out[2] = a + b; }
}
int main() {
  double *d; hipMalloc(&d, 64 * 12 * 8);
  probe<<<1, 64>>>(d);
  double h[64 * 12]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *nm[4] = {"4x4x4 A=lane,B=1", "4x4x4 A=1,B=lane", "chain A:=d1", "chain B:=d1"};
  for (int c = 0; c < 4; ++c) { printf("%s:\n", nm[c]); for (int l = 0; l < 64; ++l) printf("%5.0f%s", h[l * 12 + c], (l % 16 == 15) ? "\n" : " "); }
  printf("16x16x4 A=1,B=lane regs0..3 (lanes 0..63):\n");
  for (int r = 0; r < 4; ++r) { for (int l = 0; l < 64; ++l) printf("%4.0f%s", h[l * 12 + 4 + r], (l % 16 == 15) ? "\n" : " "); printf("--\n"); }
  printf("16x16x4 A=lane,B=1 reg0:\n");
  for (int r = 0; r < 4; ++r) { for (int l = 0; l < 64; ++l) printf("%4.0f%s", h[l * 12 + 8 + r], (l % 16 == 15) ? "\n" : " "); printf("--\n"); }
  timing<<<1, 64>>>(d, 100000); hipMemcpy(h, d, 3 * 8, hipMemcpyDeviceToHost);
  printf("cycles: 2x mfma4x4x4 + fma chain = %.1f ; 1x mfma16x16x4 + fma = %.1f\n", h[0], h[1]);
  return 0;
}
