// dev experiment: LDS read throughput of one 512-thread workgroup for the access patterns of the
// workgroup-per-problem path (row gathers with heavy address sharing).
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/lds_probe.hip -o /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(512) probe(double *out, int iters) {
  __shared__ double lds[128 * 4 * 3 + 6000];  // rows of 4 doubles; second array at +512 doubles
  const int tid = threadIdx.x;
  for (int i = tid; i < 128 * 4 * 3 + 6000; i += 512) lds[i] = 1.0 + i;
  __syncthreads();
  const int node = tid >> 2, part = tid & 3;
  double acc0 = 0, acc1 = 0, acc2 = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll 4
    for (int s = 0; s < 28; ++s) {
      int j;
      if (MODE == 0) j = part * 28 + s;                       // 4 rows per wave, shared by nodes
      else if (MODE == 1) j = (part * 28 + s + node) % 116;   // every node a different row
      else if (MODE == 2) j = s;                              // one row for the whole wave
      else if (MODE == 3) j = (part * 29 + s) % 116;          // 4 rows, odd row stride between parts
      else {                                                  // obstacle-clique lists of the table scene
        const int me = 16 + (node % 100);
        const int e = part * 27 + s;
        j = e < 4 ? e : (e < 6 ? e + 10 : e + 10);
        if (j >= me) j += 1;
        if (j > 115) j = me;
      }
      const double2 a = *reinterpret_cast<const double2 *>(&lds[j * 4]);
      const double b = lds[j * 4 + 2];
      acc0 += a.x; acc1 += a.y; acc2 += b;
      if (MODE >= 5) {                                        // second array 4 KB further (the W rows)
        const double2 c = *reinterpret_cast<const double2 *>(&lds[512 + j * 4]);
        const double e2 = lds[512 + j * 4 + 2];
        acc0 = fma(c.x, a.x, acc0); acc1 = fma(c.y, a.y, acc1); acc2 = fma(e2, b, acc2);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (tid == 0) out[0] = (double)(t1 - t0) / iters;
  out[1 + tid] = acc0 + acc1 + acc2;
}

int main() {
  double *d, h[2];
  hipMalloc(&d, 8 * 600);
  const char *nm[6] = {"4 rows/wave (parts at 28-row stride)", "distinct row per node", "1 row per wave",
                       "4 rows/wave (29-row stride)", "table-scene clique lists", "clique lists, Y and W rows"};
  for (int m = 0; m < 6; ++m) {
    if (m == 4) probe<4><<<1, 512>>>(d, 200);
    if (m == 5) probe<5><<<1, 512>>>(d, 200);
    if (m == 0) probe<0><<<1, 512>>>(d, 200);
    if (m == 1) probe<1><<<1, 512>>>(d, 200);
    if (m == 2) probe<2><<<1, 512>>>(d, 200);
    if (m == 3) probe<3><<<1, 512>>>(d, 200);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-40s %8.0f cycles per 28 row reads (b128+b64) x 8 waves -> %.1f cycles per DS instr\n", nm[m], h[0],
           h[0] / (28 * (m == 5 ? 4 : 2) * 8));
  }
  return 0;
}
