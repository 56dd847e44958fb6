// graphik_amd/csrc/gik_k_npt.hip -- device code of the GIK_KERNELS_NPT group (gik_instances.h)
#include "gik_kernels.hip.h"
#include "gik_instances.h"

namespace gik {
GIK_KERNELS_NPT(GIK_INSTANTIATE)
}  // namespace gik
