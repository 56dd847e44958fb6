// dev experiment: closer model of BlockCtx::ehess -- slot table in LDS, dependent row addresses,
// two row arrays, per-slot fp64 work -- to find what makes the real loop 5x slower per DS
// instruction than the plain address-stream probe (lds_probe.hip).
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/lds_probe2.hip -o /tmp/lds_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ void __launch_bounds__(512) probe(const uint32_t *g_slots, double *out, int iters, int SL) {
  extern __shared__ double smem[];
  double *Y = smem, *W = smem + 512, *tgt = smem + 1536;          // rows of 4 doubles
  uint32_t *slots = reinterpret_cast<uint32_t *>(tgt + 5612);
  const int tid = threadIdx.x;
  for (int i = tid; i < 1536 + 5612; i += 512) smem[i] = 1.0 + 1e-3 * i;
  for (int i = tid; i < SL * 512; i += 512) slots[i] = g_slots[i];
  __syncthreads();
  const int node = tid >> 2;
  const double yi0 = Y[node * 4], yi1 = Y[node * 4 + 1], yi2 = Y[node * 4 + 2];
  double wi0 = W[node * 4], wi1 = W[node * 4 + 1], wi2 = W[node * 4 + 2];
  double a0 = 0, a1 = 0, a2 = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll 4
    for (int s = 0; s < SL; ++s) {
      const uint32_t m = slots[s * 512 + tid];
      const int j = m & 0xff;
      const double2 ya = *reinterpret_cast<const double2 *>(&Y[j * 4]);
      const double yb = Y[j * 4 + 2];
      double2 wa = {0.5, 0.25};
      double wb = 0.125;
      if (MODE != 1) {
        wa = *reinterpret_cast<const double2 *>(&W[j * 4]);
        wb = W[j * 4 + 2];
      }
      const double y0 = yi0 - ya.x, y1 = yi1 - ya.y, y2 = yi2 - yb;
      const double w0 = wi0 - wa.x, w1 = wi1 - wa.y, w2 = wi2 - wb;
      if (MODE == 2) {   // no arithmetic beyond keeping the loads alive
        a0 += y0 + w0; a1 += y1 + w1; a2 += y2 + w2;
      } else {
        const double d = fma(y2, y2, fma(y1, y1, y0 * y0));
        const double sd = fma(y2, w2, fma(y1, w1, y0 * w0));
        const double c = d - tgt[(m >> 8) & 0xffff];
        const double a2s = 2.0 * sd;
        a0 = fma(a2s, y0, fma(c, w0, a0));
        a1 = fma(a2s, y1, fma(c, w1, a1));
        a2 = fma(a2s, y2, fma(c, w2, a2));
      }
    }
    wi0 += 1e-9 * a0;
  }
  const long long t1 = __builtin_readcyclecounter();
  if (tid == 0) out[0] = (double)(t1 - t0) / iters;
  out[1 + tid] = a0 + a1 + a2;
}

int main() {
  const int SL = 28;
  static uint32_t h_slots[28 * 512];
  for (int tid = 0; tid < 512; ++tid) {
    const int node = tid >> 2, part = tid & 3;
    for (int s = 0; s < SL; ++s) {
      int j = node < 116 ? node : 0, term = 0;
      if (node >= 16 && node < 116 && s < 27) {          // obstacle node: clique + 6 robot nodes
        const int e = part * 27 + s;
        if (e < 105) {
          j = e < 4 ? e : (e < 6 ? e + 10 : e + 10);
          if (j >= node) j += 1;
          term = (node * 131 + j * 17) % 5612;
        }
      }
      h_slots[s * 512 + tid] = (uint32_t)j | ((uint32_t)term << 8);
    }
  }
  uint32_t *d_slots;
  double *d, h[2];
  hipMalloc(&d_slots, sizeof(h_slots));
  hipMemcpy(d_slots, h_slots, sizeof(h_slots), hipMemcpyHostToDevice);
  hipMalloc(&d, 8 * 600);
  const size_t smem = 8 * (1536 + 5612) + 4 * SL * 512;
  hipFuncSetAttribute((const void *)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipFuncSetAttribute((const void *)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipFuncSetAttribute((const void *)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const char *nm[3] = {"full slot (Y row, W row, target, 20 fp64 ops)", "no W row reads", "all reads, no arithmetic"};
  for (int m = 0; m < 3; ++m) {
    if (m == 0) probe<0><<<1, 512, smem>>>(d_slots, d, 100, SL);
    if (m == 1) probe<1><<<1, 512, smem>>>(d_slots, d, 100, SL);
    if (m == 2) probe<2><<<1, 512, smem>>>(d_slots, d, 100, SL);
    hipError_t e = hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-48s %8.0f cycles per sweep of %d slots (8 waves)%s\n", nm[m], h[0], SL, e == hipSuccess ? "" : "  [error]");
  }
  return 0;
}
