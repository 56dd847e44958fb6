"""graphik_amd -- MI355X-native batched distance-geometric inverse kinematics.

Drop-in backend for the RiemannianSolver hot path of utiasSTARS/GraphIK: hand-written HIP
kernels for gfx950 behind a C ABI (include/graphik_amd.h), loaded with ctypes.  PyTorch-ROCm is
used for device buffers / streams / torch.distributed only.
"""
__version__ = "0.1.0"
