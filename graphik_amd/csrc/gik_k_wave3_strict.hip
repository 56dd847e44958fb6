// graphik_amd/csrc/gik_k_wave3_strict.hip -- device code of the GIK_KERNELS_WAVE3_STRICT group (gik_instances.h)
#include "gik_kernels.hip.h"
#include "gik_instances.h"

namespace gik {
GIK_KERNELS_WAVE3_STRICT(GIK_INSTANTIATE)
}  // namespace gik
