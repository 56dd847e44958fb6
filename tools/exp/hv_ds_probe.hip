// dev experiment: what a tCG iteration of the wavefront kernel pays for its DS instructions, lone
// wave and saturated SIMD.  One wave per workgroup, 19 KB of dynamic LDS (8 waves per CU like the
// solve kernel).  Per iteration: 1 ds_write_b64, the gathers, the product's FMAs, and FILL more
// dependent-ish fp64 ops + 16 DPP moves standing in for the rest of the iteration.
//   MODE 0: 9 ds_read_b64 at the kernel's addresses (48-byte rows, chain neighbours)
//   MODE 1: 5 ds_read2_b64 (rows j, j+1 of one base) + 30 FMAs
//   MODE 2: 9 ds_read_b64, conflict-free addresses
//   MODE 3: 9 x (ds_read_b128 + ds_read_b64) (the row form's traffic)
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/hv_ds_probe.hip -o /tmp/hv_ds_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(64, 2) probe(double *out, int iters, int fill) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2400; i += 64) smem[i] = 1.0 + 1e-6 * i;
  __builtin_amdgcn_wave_barrier();
  const int node = lane < 54 ? lane / 3 : 32, comp = lane < 54 ? lane % 3 : 0;
  int off[9];
#pragma unroll
  for (int s = 0; s < 9; ++s) {
    int j = node + s - 4;                       // chain-like neighbourhood
    j = j < 0 ? j + 18 : (j >= 18 ? j - 18 : j);
    if (lane >= 54) j = 32;
    off[s] = (MODE == 2) ? (lane + 7 * s) % 64 : j * 6 + comp;
  }
  double c[10][3];
#pragma unroll
  for (int s = 0; s < 10; ++s)
#pragma unroll
    for (int t = 0; t < 3; ++t) c[s][t] = 1e-3 * (s + 1) + 1e-4 * t + 1e-6 * lane;
  double w = 1.0 + lane * 1e-3, acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = 1e-3 * q;
  for (int it = 0; it < iters; ++it) {
    smem[node * 6 + comp] = w;
    __builtin_amdgcn_wave_barrier();
    double p[3] = {c[9][0] * w, c[9][1] * w, c[9][2] * w};
    if (MODE == 0 || MODE == 2) {
      double v[9];
#pragma unroll
      for (int s = 0; s < 9; ++s) v[s] = smem[off[s]];
#pragma unroll
      for (int s = 0; s < 9; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) p[t] = fma(c[s][t], v[s], p[t]);
    } else if (MODE == 1) {
      double v[5][2];
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const double *b = smem + off[2 * s];
        v[s][0] = b[0];
        v[s][1] = b[6];
      }
#pragma unroll
      for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) p[t] = fma(c[2 * s + 1][t], v[s][1], fma(c[2 * s][t], v[s][0], p[t]));
    } else {
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        const double2 a = *reinterpret_cast<const double2 *>(smem + ((off[s] - comp) & ~1));
        const double b3 = smem[off[s] - comp + 2];
        p[0] = fma(c[s][0], a.x, p[0]);
        p[1] = fma(c[s][1], a.y, p[1]);
        p[2] = fma(c[s][2], b3, p[2]);
      }
    }
    double h = p[0] + p[1] + p[2];
    for (int f = 0; f < fill; ++f) {           // stand-in for reduction / scalar chain / updates
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = fma(acc[q], 1.0000001, h);
      h += __shfl_xor(acc[f & 7], 1);
    }
    w = fma(1e-9, h, w);
  }
  out[blockIdx.x * 64 + lane] = w + acc[0] + acc[3];
}

template <int MODE>
static void run(const char *name, int grid, int iters, int fill, double *d_out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 19456);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(64), 19456, 0, d_out, 100, fill);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(64), 19456, 0, d_out, iters, fill);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s grid %5d: %7.1f ns per iteration per wave (%.0f cycles at 2.4 GHz)\n", name, grid, ms * 1e6 / iters,
         ms * 1e6 / iters * 2.4);
}

int main() {
  double *d_out; hipMalloc(&d_out, 8 * 64 * 4096);
  const int iters = 20000, fill = 12;   // 12 x (8 fma + shuffle) ~ 110 VALU
  for (int grid : {1, 1024, 2048}) {
    run<0>("9 ds_read_b64 (kernel addresses)", grid, iters, fill, d_out);
    run<2>("9 ds_read_b64 (conflict free)", grid, iters, fill, d_out);
    run<1>("5 ds_read2_b64", grid, iters, fill, d_out);
    run<3>("9 x (b128 + b64) row form", grid, iters, fill, d_out);
  }
  return 0;
}
