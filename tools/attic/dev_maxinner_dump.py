"""dev: dump the inner iterations of a tCG call that runs into maxinner (needs the -DGIK_TCGDUMP build:
GIK_LIB_PATH=graphik_amd/lib/exp/libgik_dump.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphik_amd import _ffi
from graphik_amd.utils.roboturdf import load_ur10
from graphik_amd.solvers.riemannian_solver import BatchProblem
from graphik_amd.engine import Template
robot, graph = load_ur10()
prob = BatchProblem(graph, use_limits=True)
B = 1024
rng = np.random.RandomState(3)
Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(B, robot.n))
targets, Y0 = prob.prepare(Tg)
b, k = int(sys.argv[1]), int(sys.argv[2])
tk = Template.from_matrices(prob.omega, prob.psi_L, prob.psi_U, k=3, use_limits=True, params=dict(maxiter=k))
r = tk.solve(Y0[b:b + 1], targets[b:b + 1], trace_cap=3000); torch.cuda.synchronize()
print("after %d its: f %.3e gn %.3e  last stops %s numit %s Delta %s" % (k, r["f"][0], r["gradnorm"][0], r["trace"]["stop"][0, k - 3:k].tolist(), r["trace"]["numit"][0, k - 3:k].tolist(), r["trace"]["Delta"][0, k - 3:k].tolist()))
Y = r["x"].reshape(1, -1)
os.environ["GIK_DBG"] = "4"
t1 = Template.from_matrices(prob.omega, prob.psi_L, prob.psi_U, k=3, use_limits=True, params=dict(maxiter=1))
r1 = t1.solve(Y, targets[b:b + 1], trace_cap=4); torch.cuda.synchronize()
print("one more it: numit %d stop %d" % (r1["trace"]["numit"][0, 0], r1["trace"]["stop"][0, 0]))
buf = np.zeros(1024 * 8)
_ffi.lib().gik_debug_fetch(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
buf = buf.reshape(1024, 8)
for s in list(range(0, 256, 8)) + list(range(256, 256 + 313, 8)):
    j = s if s < 256 else (s - 256) << 5
    q = buf[s]
    if q[0] == 0: continue
    print("j %5d r_r %.3e d_Hd %.3e alpha %.3e model %.17e  r.H/d_Hd %.6f  HdHd %.3e  <delta,Q0>/|delta| %.2e <r,Q0>/|r| %.2e" % (
        j, q[0], q[1], q[2], q[3], q[4] / q[1], q[5], q[6], q[7] / np.sqrt(q[0])))
