import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
os.environ["GIK_DBG"] = str(8 | int(os.environ.get("GIK_DBG", "0")))   # 16: no checkpoint resume
from graphik_amd.engine import Template
from graphik_amd import _ffi
d = np.load("tests/golden/lwa4d.npz")
T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True, params=dict(maxiter=400))
r = T.solve(d["Y_init"][3:4], T.targets_from_D(d["D_goal"][3:4])); torch.cuda.synchronize()
buf = np.zeros(8); L = C.CDLL(_ffi.LIB_PATH); L.gik_debug_fetch(buf.ctypes.data_as(C.c_void_p), 8)
its = int(r["iterations"][0])
print("%s: %.1f cycles per executed tCG iteration (%d executed, %d counted as the reference does, %d outer); outside tCG: %.0f cycles per outer iteration; total %.3f ms" % (
    os.environ.get("GIK_LIB_PATH", "default").split("/")[-1], buf[0] / buf[3], buf[3], buf[1], its, (buf[2] - buf[0]) / its, buf[2] / 2.4e6))
