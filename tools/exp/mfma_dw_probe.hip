// Probe for the clique (D w) product on v_mfma_f64_4x4x4: operand layouts with general A and B,
// and the cost of a chain of accumulating MFMAs fed from LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
__global__ void probe(const double *a, const double *b, double *out) {
  const int l = threadIdx.x;
  out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
template <int NC>
__global__ void chain(double *out, int iters, int n) {
  __shared__ double w[128 * 4];
  const int l = threadIdx.x;
  for (int t = l; t < 512; t += 64) w[t] = 1e-3 * t;
  __syncthreads();
  double dr[32];
#pragma unroll
  for (int m = 0; m < 32; ++m) dr[m] = 1e-3 * (l + m);
  const double *wb = w + (l >> 4) * 4 + (l & 3);
  double s = 0.0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    double c[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) c[q] = 0.0;
#pragma unroll
    for (int m = 0; m < 28; ++m) c[m % NC] = __builtin_amdgcn_mfma_f64_4x4x4f64(dr[m], wb[m * 16], c[m % NC], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NC; ++q) s += c[q];
    w[l] = s * 1e-9;   // keep the loads inside the loop
    __builtin_amdgcn_wave_barrier();
  }
  long long t1 = __builtin_readcyclecounter();
  out[l] = s;
  if (l == 0) out[64] = double(t1 - t0) / iters;
}
int main() {
  double ha[64], hb[64], hd[65], *a, *b, *d;
  srand(1);
  for (int i = 0; i < 64; ++i) { ha[i] = rand() / (double)RAND_MAX; hb[i] = rand() / (double)RAND_MAX; }
  hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 65 * 8);
  hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(a, b, d);
  hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
  // hypothesis: A[blk][i][k] = a[4 blk + i + 16 k], B[blk][k][j] = b[4 blk + j + 16 k], D[blk][i][j] at 4 blk + j + 16 i
  double worst = 0;
  for (int blk = 0; blk < 4; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
    double s = 0; for (int k = 0; k < 4; ++k) s += ha[4 * blk + i + 16 * k] * hb[4 * blk + j + 16 * k];
    worst = fmax(worst, fabs(s - hd[4 * blk + j + 16 * i]));
  }
  printf("layout hypothesis A[i][k]=lane(4b+i+16k) B[k][j]=lane(4b+j+16k) D[i][j]=lane(4b+j+16i): max err %.3e\n", worst);
  chain<1><<<1, 64>>>(d, 20000, 28); hipMemcpy(hd, d, 65 * 8, hipMemcpyDeviceToHost);
  printf("28 accumulating mfma_f64_4x4x4 fed from LDS, 1 chain: %.0f cycles (one wave alone)\n", hd[64]);
  chain<2><<<1, 64>>>(d, 20000, 28); hipMemcpy(hd, d, 65 * 8, hipMemcpyDeviceToHost);
  printf("  2 chains: %.0f cycles\n", hd[64]);
  chain<4><<<1, 64>>>(d, 20000, 28); hipMemcpy(hd, d, 65 * 8, hipMemcpyDeviceToHost);
  printf("  4 chains: %.0f cycles\n", hd[64]);
  chain<7><<<1, 64>>>(d, 20000, 28); hipMemcpy(hd, d, 65 * 8, hipMemcpyDeviceToHost);
  printf("  7 chains: %.0f cycles\n", hd[64]);
  chain<4><<<1, 128>>>(d, 20000, 28); hipMemcpy(hd, d, 65 * 8, hipMemcpyDeviceToHost);
  printf("  4 chains, two waves (different SIMDs probably): %.0f cycles\n", hd[64]);
  chain<4><<<1, 512>>>(d, 20000, 28); hipMemcpy(hd, d, 65 * 8, hipMemcpyDeviceToHost);
  printf("  4 chains, eight waves (two per SIMD): %.0f cycles\n", hd[64]);
  return 0;
}
