// Validate MFMA-based wave reductions (values + timing) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CTRL, int BANK = 0xf>
__device__ inline double dppm(double old, double v) {
  int lo = __double2loint(v), hi = __double2hiint(v), olo = __double2loint(old), ohi = __double2hiint(old);
  lo = __builtin_amdgcn_update_dpp(olo, lo, CTRL, 0xf, BANK, false);
  hi = __builtin_amdgcn_update_dpp(ohi, hi, CTRL, 0xf, BANK, false);
  return __hiloint2double(hi, lo);
}
__device__ inline double rl(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ inline double blocksum(double v) {
  double p = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
  return __builtin_amdgcn_mfma_f64_4x4x4f64(p, 1.0, 0.0, 0, 0, 0);
}
__device__ inline double wsum1(double x) {
  double q = blocksum(x);
  double t = (q + dppm<0x124>(q, q)) + (dppm<0x128>(q, q) + dppm<0x12C>(q, q));
  return rl(t, 0);
}
__device__ inline double xor4(double x) {
  double t = dppm<0x104, 0x5>(x, x);   // row_shl:4 into banks 0,2
  return dppm<0x114, 0xA>(t, x);       // row_shr:4 into banks 1,3
}
__device__ inline void wsum4(double &v0, double &v1, double &v2, double &v3, int lane) {
  const bool h8 = lane & 8, h4 = lane & 4;
  double u0 = (h8 ? v1 : v0) + dppm<0x128>(0.0, h8 ? v0 : v1);
  double u1 = (h8 ? v3 : v2) + dppm<0x128>(0.0, h8 ? v2 : v3);
  double w = (h4 ? u1 : u0) + xor4(h4 ? u0 : u1);
  double q = blocksum(w);
  v0 = rl(q, 0); v2 = rl(q, 4); v1 = rl(q, 8); v3 = rl(q, 12);
}
__global__ void probe(double *out, int iters) {
  const int l = threadIdx.x;
  double a = 1.0 + 0.001 * l, b = sin(0.37 * l), c = 1.0 / (1 + l), d = (l % 7) - 3.0;
  double s1 = wsum1(a);
  double v0 = a, v1 = b, v2 = c, v3 = d;
  wsum4(v0, v1, v2, v3, l);
  if (l == 0) { out[0] = s1; out[1] = v0; out[2] = v1; out[3] = v2; out[4] = v3; }
  out[8 + l] = xor4((double)l);
  // timing
  double acc = a;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) acc = 0.5 + 1e-3 * wsum1(acc);
  long long t1 = __builtin_readcyclecounter();
  double x0 = a, x1 = b, x2 = c, x3 = d;
  for (int i = 0; i < iters; ++i) {
    double y0 = x0, y1 = x1, y2 = x2, y3 = x3;
    wsum4(y0, y1, y2, y3, l);
    x0 = a + 1e-3 * y0; x1 = b + 1e-3 * y1; x2 = c + 1e-3 * y2; x3 = d + 1e-3 * y3;
  }
  long long t2 = __builtin_readcyclecounter();
  if (l == 0) { out[5] = double(t1 - t0) / iters; out[6] = double(t2 - t1) / iters; out[7] = acc + x0 + x1 + x2 + x3; }
}
int main() {
  double *d; hipMalloc(&d, 80 * 8);
  probe<<<1, 64>>>(d, 100000);
  double h[80]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double r[5] = {0, 0, 0, 0, 0};
  for (int l = 0; l < 64; ++l) { r[0] += 1.0 + 0.001 * l; r[1] += 1.0 + 0.001 * l; r[2] += sin(0.37 * l); r[3] += 1.0 / (1 + l); r[4] += (l % 7) - 3.0; }
  for (int i = 0; i < 5; ++i) printf("sum%d gpu %.15g ref %.15g diff %.2e\n", i, h[i], r[i], h[i] - r[i]);
  printf("xor4 of lane id:"); for (int l = 0; l < 32; ++l) printf(" %.0f", h[8 + l]); printf("\n");
  printf("cycles: wsum1 %.1f  wsum4 %.1f\n", h[5], h[6]);
  return 0;
}
