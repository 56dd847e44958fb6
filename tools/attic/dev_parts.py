import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphik_amd.engine import Template
from graphik_amd import _ffi
d = np.load("tests/golden/lwa4d.npz")
T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True)
L = C.CDLL(_ffi.LIB_PATH); L.gik_debug_parts.restype = C.c_double; L.gik_debug_parts.argtypes = [C.c_void_p, C.c_int, C.c_int]
names = {0: "ehess", 1: "wave_sum_n<3>", 2: "wave_sum<1>", 3: "fp64 div", 4: "proj(ehess)", 5: "8 dependent fma", 6: "sqrt", 7: "64 fma (8 indep chains)", 8: "LDS put+read", 9: "dpp+mul+add", 10: "readlane+mul+add", 11: "8 dep f32 fma (+cvt)", 12: "64 indep add_f64", 13: "64 indep mul_f64", 14: "64 indep fma_f64 vvv", 15: "192 indep 32-bit ALU", 16: "64 v_mov_dpp (+8 add)", 17: "64 v_cndmask (+8 add)", 18: "put + 11 gathers, no math", 19: "cost()", 20: "commit()", 21: "proj_setup()", 22: "sum1 + sqrt", 23: "1-value DPP butterfly"}
for mode in range(24):
    L.gik_debug_parts(T._h, mode, 1000)
    c = L.gik_debug_parts(T._h, mode, 20000)
    ns = L.gik_debug_parts(T._h, 100 + mode, 200000)
    print("%-24s %8.1f ticks  %8.1f ns  -> tick = %.3f ns" % (names[mode], c, ns, ns / c), flush=True)
