// graphik_amd/csrc/gik_instances.h -- the list of compiled kernel instantiations, by translation unit.
//
// GIK_KERNELS_<GROUP>(X) calls X(<function signature>) for every instantiation of the group; gik_k_<group>.hip expands
// it with GIK_INSTANTIATE, gik_host.hip expands all groups with GIK_EXTERN_TEMPLATE (so that taking a kernel's address
// there -- the variant tables, hipLaunchKernelGGL -- refers to the other file's symbol instead of compiling the kernel
// a second time).  The four non-template kernels (prep_wave_kernel, recover_kernel, anch_init_kernel,
// anch_gather_kernel) are defined where GIK_DEFINE_PLAIN_KERNELS is set: gik_k_prep.hip.
#pragma once

#define GIK_INSTANTIATE(...) template __global__ __VA_ARGS__;
#define GIK_EXTERN_TEMPLATE(...) extern template __global__ __VA_ARGS__;

// one unknown per lane, k = 3: column-form product (+ ConjugateGradient, known answers, tail spreading)
#define GIK_KERNELS_WAVE3(X)                                   \
  X(void rtr_wave_kernel<3, 9, true>(SolveArgs))               \
  X(void rtr_wave_kernel<3, 9, false>(SolveArgs))              \
  X(void rtr_wave_kernel<3, 9, true, false, true>(SolveArgs))  \
  X(void rcg_wave_kernel<3, 9>(SolveArgs))                     \
  X(void kat_wave_kernel<3, 9>(KatArgs))                       \
  X(void rtr_wave_kernel<3, 10, true>(SolveArgs))              \
  X(void rtr_wave_kernel<3, 10, false>(SolveArgs))             \
  X(void rcg_wave_kernel<3, 10>(SolveArgs))                    \
  X(void kat_wave_kernel<3, 10>(KatArgs))
// ... per-edge product form (gik_wave_strict.hip.h)
#define GIK_KERNELS_WAVE3_STRICT(X)                                  \
  X(void rtr_wave_kernel<3, 9, true, false, false, true>(SolveArgs)) \
  X(void rtr_wave_kernel<3, 9, true, false, true, true>(SolveArgs))  \
  X(void rtr_wave_kernel<3, 9, false, false, false, true>(SolveArgs)) \
  X(void kat_wave_kernel<3, 9, false, true>(KatArgs))                \
  X(void rtr_wave_kernel<3, 10, true, false, false, true>(SolveArgs)) \
  X(void rtr_wave_kernel<3, 10, true, false, true, true>(SolveArgs)) \
  X(void rtr_wave_kernel<3, 10, false, false, false, true>(SolveArgs)) \
  X(void kat_wave_kernel<3, 10, false, true>(KatArgs))
// ... fixed-anchor formulation
#define GIK_KERNELS_ANCH(X)                             \
  X(void rtr_wave_kernel<3, 9, true, true>(SolveArgs))  \
  X(void kat_wave_kernel<3, 9, true>(KatArgs))          \
  X(void rtr_wave_kernel<3, 20, true, true>(SolveArgs)) \
  X(void kat_wave_kernel<3, 20, true>(KatArgs))
// one unknown per lane, k = 2
#define GIK_KERNELS_WAVE2(X)                       \
  X(void rtr_wave_kernel<2, 6, true>(SolveArgs))   \
  X(void rtr_wave_kernel<2, 6, false>(SolveArgs))  \
  X(void rcg_wave_kernel<2, 6>(SolveArgs))         \
  X(void kat_wave_kernel<2, 6>(KatArgs))           \
  X(void rtr_wave_kernel<2, 16, true>(SolveArgs))  \
  X(void rtr_wave_kernel<2, 16, false>(SolveArgs)) \
  X(void rcg_wave_kernel<2, 16>(SolveArgs))        \
  X(void kat_wave_kernel<2, 16>(KatArgs))          \
  X(void rtr_wave_kernel<2, 31, true>(SolveArgs))  \
  X(void rtr_wave_kernel<2, 31, false>(SolveArgs)) \
  X(void rcg_wave_kernel<2, 31>(SolveArgs))        \
  X(void kat_wave_kernel<2, 31>(KatArgs))
// workgroup per problem
#define GIK_KERNELS_BLOCK(X)                    \
  X(void rtr_block_kernel<2>(SolveArgs, int))   \
  X(void rtr_block_kernel<3>(SolveArgs, int))   \
  X(void rcg_block_kernel<2>(SolveArgs, int))   \
  X(void rcg_block_kernel<3>(SolveArgs, int))   \
  X(void kat_block_kernel<2>(KatArgs, int))     \
  X(void kat_block_kernel<3>(KatArgs, int))
// node per lane
#define GIK_KERNELS_NPT(X)                              \
  X(void rtr_npt_kernel<1, 1, 2, false>(SolveArgs))     \
  X(void kat_npt_kernel<1, 1, 2, false>(KatArgs))       \
  X(void rtr_npt_kernel<4, 1, 2, false>(SolveArgs))     \
  X(void kat_npt_kernel<4, 1, 2, false>(KatArgs))       \
  X(void rtr_npt_kernel<1, 2, 1, false>(SolveArgs))     \
  X(void kat_npt_kernel<1, 2, 1, false>(KatArgs))       \
  X(void rtr_npt_kernel<4, 2, 1, false>(SolveArgs))     \
  X(void kat_npt_kernel<4, 2, 1, false>(KatArgs))
#define GIK_KERNELS_NPT4(X)                             \
  X(void rtr_npt_kernel<1, 1, 4, true>(SolveArgs))      \
  X(void kat_npt_kernel<1, 1, 4, true>(KatArgs))        \
  X(void rtr_npt_kernel<4, 1, 4, true>(SolveArgs))      \
  X(void kat_npt_kernel<4, 1, 4, true>(KatArgs))
// four planar problems / goals per wavefront
#define GIK_KERNELS_QUAD(X)                 \
  X(void rtr_quad_kernel<6>(SolveArgs))     \
  X(void kat_quad_kernel<6>(KatArgs))       \
  X(void prep_quad_kernel<13>(PrepArgs))    \
  X(void prep_quad_kernel<0>(PrepArgs))
// prepare (workgroup per goal); the plain kernels of gik_prep.hip.h are defined next to these
#define GIK_KERNELS_PREP(X)                            \
  X(void prep_block_kernel<true>(PrepArgs, double *))  \
  X(void prep_block_kernel<false>(PrepArgs, double *)) \
  X(void prep_block_kernel<false, PREP_BIGN>(PrepArgs, double *))

#define GIK_ALL_KERNELS(X)                                                                             \
  GIK_KERNELS_WAVE3(X) GIK_KERNELS_WAVE3_STRICT(X) GIK_KERNELS_ANCH(X) GIK_KERNELS_WAVE2(X) GIK_KERNELS_BLOCK(X) \
  GIK_KERNELS_NPT(X) GIK_KERNELS_NPT4(X) GIK_KERNELS_QUAD(X) GIK_KERNELS_PREP(X)
