// graphik_amd/csrc/gik_k_prep.hip -- device code of the GIK_KERNELS_PREP group (gik_instances.h)
#define GIK_DEFINE_PLAIN_KERNELS 1
#include "gik_kernels.hip.h"
#include "gik_instances.h"

namespace gik {
GIK_KERNELS_PREP(GIK_INSTANTIATE)
}  // namespace gik
