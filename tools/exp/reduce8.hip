// dev experiment: 8-value wavefront reduction with gfx950 permlane swaps vs. the MFMA scheme
// build: hipcc --offload-arch=gfx950 -O3 -I include -I graphik_amd/csrc tools/exp/reduce8.hip -o /tmp/reduce8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "gik_wave.hip.h"
using namespace gik;

__device__ inline void swap32(double &a, double &b) {  // a.hi-lanes <-> b.lo-lanes
  unsigned al = __double2loint(a), ah = __double2hiint(a), bl = __double2loint(b), bh = __double2hiint(b);
  auto r = __builtin_amdgcn_permlane32_swap(al, bl, false, false); al = r[0]; bl = r[1];
  auto s = __builtin_amdgcn_permlane32_swap(ah, bh, false, false); ah = s[0]; bh = s[1];
  a = __hiloint2double(ah, al); b = __hiloint2double(bh, bl);
}
__device__ inline void swap16(double &a, double &b) {
  unsigned al = __double2loint(a), ah = __double2hiint(a), bl = __double2loint(b), bh = __double2hiint(b);
  auto r = __builtin_amdgcn_permlane16_swap(al, bl, false, false); al = r[0]; bl = r[1];
  auto s = __builtin_amdgcn_permlane16_swap(ah, bh, false, false); ah = s[0]; bh = s[1];
  a = __hiloint2double(ah, al); b = __hiloint2double(bh, bl);
}

__global__ void layout(double *o) {
  double a = threadIdx.x, b = 100 + threadIdx.x;
  double a2 = a, b2 = b;
  swap32(a, b);
  swap16(a2, b2);
  o[threadIdx.x] = a; o[64 + threadIdx.x] = b; o[128 + threadIdx.x] = a2; o[192 + threadIdx.x] = b2;
}

// 8 values -> wave-uniform totals
__device__ inline void wave_sum8_swap(double (&v)[8]) {
  // stage bit 5: pairs (0,1)(2,3)(4,5)(6,7)
#pragma unroll
  for (int q = 0; q < 8; q += 2) { swap32(v[q], v[q + 1]); v[q] += v[q + 1]; }
  // v[0],v[2],v[4],v[6]: lanes<32 hold value q, lanes>=32 value q+1
  // stage bit 4: pairs (0,2) (4,6)
  swap16(v[0], v[2]); v[0] += v[2];
  swap16(v[4], v[6]); v[4] += v[6];
  // stage bit 3: pair (0,4) via row_ror:8 select exchange
  const bool h8 = threadIdx.x & 8;
  double w = (h8 ? v[4] : v[0]) + dpp_f64<0x128>(h8 ? v[0] : v[4]);
  // butterfly over bits 0,1,2
  w += dpp_f64<0xB1>(w);
  w += dpp_f64<0x4E>(w);
  w += dpp_f64<0x141>(w);  // row_half_mirror
  // readout: value index bits: (lane bit5 -> +1), (bit4 -> +2), (bit3 -> +4)
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int lane = ((q & 1) ? 32 : 0) + ((q & 2) ? 16 : 0) + ((q & 4) ? 8 : 0);
    v[q] = readlane_f64(w, lane);
  }
}

__global__ void check(const double *in, double *o, int mode) {
  double v[8];
  for (int q = 0; q < 8; ++q) v[q] = in[q * 64 + threadIdx.x];
  if (mode == 0) wave_sum8_swap(v); else wave_sum_n<8>(v);
  if (threadIdx.x == 0) for (int q = 0; q < 8; ++q) o[q] = v[q];
}

template <int MODE>
__global__ void timing(const double *in, double *o, int iters) {
  double x = in[threadIdx.x];
  double c[8];
  for (int q = 0; q < 8; ++q) c[q] = in[q * 64 + threadIdx.x];
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = c[q] * x;
    if (MODE == 0) wave_sum8_swap(v);
    else if (MODE == 1) wave_sum_n<8>(v);
    else { double u[4] = {v[0], v[1], v[2], v[3]}; wave_sum_n<4>(u); v[0] = u[0]; v[1] = u[1]; v[2] = u[2]; v[3] = u[3]; }
    x = x * 0.5 + 1e-3 * (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { o[0] = (double)(t1 - t0) / iters; o[1] = x; }
}

int main() {
  double *d, h[512];
  hipMalloc(&d, sizeof(h));
  layout<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(double) * 256, hipMemcpyDeviceToHost);
  const char *nm[4] = {"swap32 a'", "swap32 b'", "swap16 a'", "swap16 b'"};
  for (int r = 0; r < 4; ++r) { printf("%s:", nm[r]); for (int l = 0; l < 64; ++l) printf(" %g", h[r * 64 + l]); printf("\n"); }
  double in[512], ref[8] = {0};
  for (int q = 0; q < 8; ++q) for (int l = 0; l < 64; ++l) { in[q * 64 + l] = sin(1.0 + q * 64 + l) * (1 + q); ref[q] += in[q * 64 + l]; }
  double *din; hipMalloc(&din, sizeof(in)); hipMemcpy(din, in, sizeof(in), hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    check<<<1, 64>>>(din, d, mode);
    hipMemcpy(h, d, sizeof(double) * 8, hipMemcpyDeviceToHost);
    double e = 0; for (int q = 0; q < 8; ++q) e = fmax(e, fabs(h[q] - ref[q]));
    printf("mode %d max err %.3e\n", mode, e);
  }
  timing<0><<<1, 64>>>(din, d, 20000); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("swap scheme      : %.1f cycles / 8 values (incl 8 mul + 8 add)\n", h[0]);
  timing<1><<<1, 64>>>(din, d, 20000); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("2 x wave_sum4    : %.1f cycles\n", h[0]);
  timing<2><<<1, 64>>>(din, d, 20000); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("1 x wave_sum4    : %.1f cycles (same overhead)\n", h[0]);
  return 0;
}
