// graphik_amd/csrc/gik_k_block.hip -- device code of the GIK_KERNELS_BLOCK group (gik_instances.h)
#include "gik_kernels.hip.h"
#include "gik_instances.h"

namespace gik {
GIK_KERNELS_BLOCK(GIK_INSTANTIATE)
}  // namespace gik
