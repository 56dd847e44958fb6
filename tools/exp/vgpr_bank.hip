// dev experiment: issue cost of independent v_fma_f64 for a lone wavefront as a function of which VGPR
// banks its three 64-bit source operands sit in (explicit registers through inline asm).
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/vgpr_bank.hip -o /tmp/vgpr_bank
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
// 8 independent destinations v[40:41]..v[54:55]; sources per pattern
#define FMA_BLOCK(S0, S1, S2)                                  \
  "v_fma_f64 v[40:41], " S0 ", " S1 ", " S2 "\n"            \
  "v_fma_f64 v[42:43], " S0 ", " S1 ", " S2 "\n"            \
  "v_fma_f64 v[44:45], " S0 ", " S1 ", " S2 "\n"            \
  "v_fma_f64 v[46:47], " S0 ", " S1 ", " S2 "\n"            \
  "v_fma_f64 v[48:49], " S0 ", " S1 ", " S2 "\n"            \
  "v_fma_f64 v[50:51], " S0 ", " S1 ", " S2 "\n"            \
  "v_fma_f64 v[52:53], " S0 ", " S1 ", " S2 "\n"            \
  "v_fma_f64 v[54:55], " S0 ", " S1 ", " S2 "\n"

#define CLOBBERS "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23", \
  "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55"

template <int P>
__global__ void k(long long *out, int iters) {
  asm volatile("v_mov_b32 v8, 0\n v_mov_b32 v9, 0x3ff00000\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0x3ff00000\n"
               "v_mov_b32 v12, 0\n v_mov_b32 v13, 0x3ff00000\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0x3ff00000\n"
               "v_mov_b32 v16, 0\n v_mov_b32 v17, 0x3ff00000\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0x3ff00000\n" ::: CLOBBERS);
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (P == 0) asm volatile(REP8(FMA_BLOCK("v[8:9]", "v[12:13]", "v[16:17]")) ::: CLOBBERS);    // banks 0,0,0
    if (P == 1) asm volatile(REP8(FMA_BLOCK("v[8:9]", "v[10:11]", "v[12:13]")) ::: CLOBBERS);    // banks 0,2,0
    if (P == 2) asm volatile(REP8(FMA_BLOCK("v[8:9]", "v[10:11]", "v[14:15]")) ::: CLOBBERS);    // banks 0,2,2
    if (P == 3) asm volatile(REP8(FMA_BLOCK("v[8:9]", "v[8:9]", "v[10:11]")) ::: CLOBBERS);      // two distinct, 0,0,2
    if (P == 4) asm volatile(REP8(FMA_BLOCK("v[8:9]", "v[8:9]", "v[12:13]")) ::: CLOBBERS);      // two distinct, 0,0,0
    if (P == 5) asm volatile(REP8(FMA_BLOCK("v[8:9]", "v[10:11]", "1.0")) ::: CLOBBERS);          // two VGPR + constant
    if (P == 6) asm volatile(REP8(FMA_BLOCK("v[8:9]", "v[12:13]", "1.0")) ::: CLOBBERS);          // same banks + constant
    if (P == 7) asm volatile(REP8(FMA_BLOCK("v[8:9]", "s[4:5]", "v[10:11]")) ::: CLOBBERS);       // SGPR operand
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[P] = t1 - t0;
}

int main() {
  long long *d, h[8];
  hipMalloc(&d, sizeof(h));
  const int iters = 200;
  k<0><<<1, 64>>>(d, iters); k<1><<<1, 64>>>(d, iters); k<2><<<1, 64>>>(d, iters); k<3><<<1, 64>>>(d, iters);
  k<4><<<1, 64>>>(d, iters); k<5><<<1, 64>>>(d, iters); k<6><<<1, 64>>>(d, iters); k<7><<<1, 64>>>(d, iters);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[8] = {"v,v,v banks 0/0/0", "v,v,v banks 0/2/0", "v,v,v banks 0/2/2", "v,same v,v banks 0/0/2",
                          "v,same v,v banks 0/0/0", "v,v,const banks 0/2", "v,v,const banks 0/0", "v,s,v banks 0/2"};
  for (int p = 0; p < 8; ++p) printf("%-28s %.2f cycles per v_fma_f64\n", names[p], (double)h[p] / (iters * 64.0));
  return 0;
}
