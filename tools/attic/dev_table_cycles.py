"""dev: table scene (workgroup kernel), problem 0 of a small batch under the developer build: cycles
per executed tCG iteration and cycles outside tCG per outer iteration (GIK_DBG=8 counters).
    GIK_LIB_PATH=<dev build> python tools/attic/dev_table_cycles.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
os.environ["GIK_DBG"] = str(8 | int(os.environ.get("GIK_DBG", "0")))
from graphik_amd.engine import Template
from graphik_amd import _ffi
d = np.load("tests/golden/ur10_table.npz")
T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True)
B = int(os.environ.get("B", "1"))
Yi = np.tile(d["Y_init"][:1], (B, 1, 1)); tg = np.tile(T.targets_from_D(d["D_goal"][:1]), (B, 1))
r = T.solve(Yi, tg); torch.cuda.synchronize()
buf = np.zeros(8); L = C.CDLL(_ffi.LIB_PATH); L.gik_debug_fetch(buf.ctypes.data_as(C.c_void_p), 8)
its = int(r["iterations"][0])
print("%s B=%d: %.1f cycles per executed tCG iteration (%d executed, %d outer); outside tCG: %.0f cycles per outer iteration; total %.3f ms" % (
    os.environ.get("GIK_LIB_PATH", "default").split("/")[-1], B, buf[0] / buf[3], buf[3], its, (buf[2] - buf[0]) / its, buf[2] / 2.4e6))
