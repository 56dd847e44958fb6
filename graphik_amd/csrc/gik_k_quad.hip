// graphik_amd/csrc/gik_k_quad.hip -- device code of the GIK_KERNELS_QUAD group (gik_instances.h)
#include "gik_kernels.hip.h"
#include "gik_instances.h"

namespace gik {
GIK_KERNELS_QUAD(GIK_INSTANTIATE)
}  // namespace gik
