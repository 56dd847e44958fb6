// graphik_amd/csrc/gik_k_wave3.hip -- device code of the GIK_KERNELS_WAVE3 group (gik_instances.h)
#include "gik_kernels.hip.h"
#include "gik_instances.h"

namespace gik {
GIK_KERNELS_WAVE3(GIK_INSTANTIATE)
}  // namespace gik
