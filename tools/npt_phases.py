"""Cycles per phase of one Hessian product of the node-per-lane kernel (developer build with
-DGIK_DEV -DGIK_NPT_PROF: python -m graphik_amd.build --dev; GIK_LIB_PATH=graphik_amd/lib/exp/libgraphik_amd_dev.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GIK_DBG"] = os.environ.get("GIK_DBG", "8")
import numpy as np, torch
from graphik_amd.engine import Template
from graphik_amd import _ffi
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/ur10_table.npz"))
T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True, params=dict(maxiter=60))
r = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"])); torch.cuda.synchronize()
buf = np.zeros(64); L = C.CDLL(_ffi.LIB_PATH); L.gik_debug_fetch(buf.ctypes.data_as(C.c_void_p), 64)
its = int(r["iterations"][0])
print(T.info)
print("%.0f cycles per tCG iteration (%d counted, %d executed, %d outer); outside tCG: %.0f cycles per outer iteration" % (
    buf[0] / buf[3], buf[1], buf[3], its, (buf[2] - buf[0]) / its))
n = max(buf[25], 1)
names = ["W rows + moment contributions + term vectors", "reduction network (24)", "gather", "exchange (barrier)",
         "scalars + closed form", "inner products + reduction (8) + barrier", "scalar step + updates"]
for i, nm in enumerate(names):
    print("   %-48s %7.0f" % (nm, buf[16 + i] / n))
print("   sum %.0f over %d products" % (sum(buf[16:23]) / n, n))
