// graphik_amd/csrc/gik_k_anch.hip -- device code of the GIK_KERNELS_ANCH group (gik_instances.h)
#include "gik_kernels.hip.h"
#include "gik_instances.h"

namespace gik {
GIK_KERNELS_ANCH(GIK_INSTANTIATE)
}  // namespace gik
