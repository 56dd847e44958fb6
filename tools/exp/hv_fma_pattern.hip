// dev experiment: the 27 v_fmac_f64 of one column-form Hessian product (fresh coefficient register per op,
// one gathered value per three ops, three accumulators) with explicit registers, for a lone wavefront.
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/hv_fma_pattern.hip -o /tmp/hv_fma_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
template <int P>
__global__ void __launch_bounds__(64) k(long long *out, int iters) {
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (P == 0) asm volatile(
    "v_fmac_f64 v[200:201], v[20:21], v[100:101]\n"
    "v_fmac_f64 v[202:203], v[22:23], v[100:101]\n"
    "v_fmac_f64 v[204:205], v[24:25], v[100:101]\n"
    "v_fmac_f64 v[200:201], v[26:27], v[102:103]\n"
    "v_fmac_f64 v[202:203], v[28:29], v[102:103]\n"
    "v_fmac_f64 v[204:205], v[30:31], v[102:103]\n"
    "v_fmac_f64 v[200:201], v[32:33], v[104:105]\n"
    "v_fmac_f64 v[202:203], v[34:35], v[104:105]\n"
    "v_fmac_f64 v[204:205], v[36:37], v[104:105]\n"
    "v_fmac_f64 v[200:201], v[38:39], v[106:107]\n"
    "v_fmac_f64 v[202:203], v[40:41], v[106:107]\n"
    "v_fmac_f64 v[204:205], v[42:43], v[106:107]\n"
    "v_fmac_f64 v[200:201], v[44:45], v[108:109]\n"
    "v_fmac_f64 v[202:203], v[46:47], v[108:109]\n"
    "v_fmac_f64 v[204:205], v[48:49], v[108:109]\n"
    "v_fmac_f64 v[200:201], v[50:51], v[110:111]\n"
    "v_fmac_f64 v[202:203], v[52:53], v[110:111]\n"
    "v_fmac_f64 v[204:205], v[54:55], v[110:111]\n"
    "v_fmac_f64 v[200:201], v[56:57], v[112:113]\n"
    "v_fmac_f64 v[202:203], v[58:59], v[112:113]\n"
    "v_fmac_f64 v[204:205], v[60:61], v[112:113]\n"
    "v_fmac_f64 v[200:201], v[62:63], v[114:115]\n"
    "v_fmac_f64 v[202:203], v[64:65], v[114:115]\n"
    "v_fmac_f64 v[204:205], v[66:67], v[114:115]\n"
    "v_fmac_f64 v[200:201], v[68:69], v[116:117]\n"
    "v_fmac_f64 v[202:203], v[70:71], v[116:117]\n"
    "v_fmac_f64 v[204:205], v[72:73], v[116:117]\n"
    ::: "v200","v201","v202","v203","v204","v205");
    if (P == 1) asm volatile(
    "v_fmac_f64 v[200:201], v[20:21], v[102:103]\n"
    "v_fmac_f64 v[204:205], v[24:25], v[102:103]\n"
    "v_fmac_f64 v[208:209], v[28:29], v[102:103]\n"
    "v_fmac_f64 v[200:201], v[32:33], v[106:107]\n"
    "v_fmac_f64 v[204:205], v[36:37], v[106:107]\n"
    "v_fmac_f64 v[208:209], v[40:41], v[106:107]\n"
    "v_fmac_f64 v[200:201], v[44:45], v[110:111]\n"
    "v_fmac_f64 v[204:205], v[48:49], v[110:111]\n"
    "v_fmac_f64 v[208:209], v[52:53], v[110:111]\n"
    "v_fmac_f64 v[200:201], v[56:57], v[114:115]\n"
    "v_fmac_f64 v[204:205], v[60:61], v[114:115]\n"
    "v_fmac_f64 v[208:209], v[64:65], v[114:115]\n"
    "v_fmac_f64 v[200:201], v[68:69], v[118:119]\n"
    "v_fmac_f64 v[204:205], v[72:73], v[118:119]\n"
    "v_fmac_f64 v[208:209], v[76:77], v[118:119]\n"
    "v_fmac_f64 v[200:201], v[80:81], v[122:123]\n"
    "v_fmac_f64 v[204:205], v[84:85], v[122:123]\n"
    "v_fmac_f64 v[208:209], v[88:89], v[122:123]\n"
    "v_fmac_f64 v[200:201], v[92:93], v[126:127]\n"
    "v_fmac_f64 v[204:205], v[96:97], v[126:127]\n"
    "v_fmac_f64 v[208:209], v[100:101], v[126:127]\n"
    "v_fmac_f64 v[200:201], v[104:105], v[130:131]\n"
    "v_fmac_f64 v[204:205], v[108:109], v[130:131]\n"
    "v_fmac_f64 v[208:209], v[112:113], v[130:131]\n"
    "v_fmac_f64 v[200:201], v[116:117], v[134:135]\n"
    "v_fmac_f64 v[204:205], v[120:121], v[134:135]\n"
    "v_fmac_f64 v[208:209], v[124:125], v[134:135]\n"
    ::: "v200","v201","v202","v203","v204","v205");
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[P] = t1 - t0;
}
int main() {
  long long *d, h[2];
  (void)hipMalloc(&d, sizeof(h));
  const int iters = 400;
  k<0><<<1, 64>>>(d, iters); k<1><<<1, 64>>>(d, iters);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // (odd-aligned 64-bit tuples are rejected by the assembler on gfx950: only two bank pairs exist)
  const char *names[2] = {"coefficients v[20,22,..], gathered v[100,102,..], accumulators v[200,202,204]",
                          "coefficients pair-parity 0, gathered pair-parity 1, accumulators pair-parity 0"};
  for (int p = 0; p < 2; ++p) printf("%-90s %.2f cycles per v_fmac_f64 (27 per product: %.0f)\n", names[p], (double)h[p] / (iters * 27.0), (double)h[p] / iters);
  return 0;
}
