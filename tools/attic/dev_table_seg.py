import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
os.environ["GIK_DBG"] = "8"
from graphik_amd.engine import Template
from graphik_amd import _ffi
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests/golden/ur10_table.npz"))
T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True, params=dict(maxiter=30))
r = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"])); torch.cuda.synchronize()
buf = np.zeros(64); L = C.CDLL(_ffi.LIB_PATH); L.gik_debug_fetch(buf.ctypes.data_as(C.c_void_p), 64)
its = int(r["iterations"][0])
print("ur10_table (block path): %.0f cycles per tCG iteration (%d inner, %d outer); outside tCG: %.0f cycles per outer iteration" % (
    buf[0] / buf[1], buf[1], its, (buf[2] - buf[0]) / its))
print("maxdeg/SL:", T.maxdeg)
if buf[12] > 0:
    for w in range(8):
        n = buf[12 + 6 * w]
        print("   wave %d: W + moments + barrier %.0f, closed form %.0f, D w %.0f, slots + combine %.0f cycles per ehess (%d calls)"
              % (w, buf[8 + 6 * w] / n, buf[9 + 6 * w] / n, buf[10 + 6 * w] / n, buf[11 + 6 * w] / n, n))
