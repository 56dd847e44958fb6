// graphik_amd/csrc/gik_k_npt4.hip -- device code of the GIK_KERNELS_NPT4 group (gik_instances.h)
#include "gik_kernels.hip.h"
#include "gik_instances.h"

namespace gik {
GIK_KERNELS_NPT4(GIK_INSTANTIATE)
}  // namespace gik
