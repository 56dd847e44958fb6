// dev check: accuracy of the reciprocal / reciprocal-square-root helpers of gik_wave.hip.h
// build: hipcc --offload-arch=gfx950 -O3 -I include -I graphik_amd/csrc tools/exp/rcp_accuracy.hip -o /tmp/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include "gik_wave.hip.h"
__global__ void k(const double *x, double *o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    o[i] = gik::frcp1(x[i]);
    o[n + i] = gik::frcp(x[i]);
    o[2 * n + i] = gik::frsqrt(fabs(x[i]));
    o[3 * n + i] = __builtin_amdgcn_rcp(x[i]);
  }
}
int main() {
  const int n = 1 << 20;
  double *hx = (double *)malloc(8 * n), *ho = (double *)malloc(8 * 4 * n), *dx, *dout;
  srand(1);
  for (int i = 0; i < n; ++i) hx[i] = ldexp(1.0 + rand() / (double)RAND_MAX, rand() % 120 - 60) * ((rand() & 1) ? 1 : -1);
  hipMalloc(&dx, 8 * n); hipMalloc(&dout, 8 * 4 * n);
  hipMemcpy(dx, hx, 8 * n, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(ho, dout, 8 * 4 * n, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0, e3 = 0, e0 = 0;
  for (int i = 0; i < n; ++i) {
    const long double r = 1.0L / hx[i], s = 1.0L / sqrtl(fabsl((long double)hx[i]));
    e1 = fmax(e1, (double)fabsl((ho[i] - r) / r));
    e2 = fmax(e2, (double)fabsl((ho[n + i] - r) / r));
    e3 = fmax(e3, (double)fabsl((ho[2 * n + i] - s) / s));
    e0 = fmax(e0, (double)fabsl((ho[3 * n + i] - r) / r));
  }
  printf("max relative error over %d samples: v_rcp_f64 %.2e | frcp1 %.2e | frcp %.2e | frsqrt %.2e  (eps = 1.1e-16)\n", n, e0, e1, e2, e3);
  return 0;
}
