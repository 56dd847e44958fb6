"""dev: prepare-kernel time vs the Jacobi sweep cap (how much of the kernel is Jacobi)."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import make_graph
from graphik_amd.solvers import riemannian_solver as rs
name, B = sys.argv[1], int(sys.argv[2])
robot, graph = make_graph(name)
rng = np.random.RandomState(0)
lb, ub = robot.limits_arrays()
Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rng.rand(B, robot.n))).cuda()
for sweeps in (0, 4, 2, 1):
    orig = rs.Template.attach_pipeline
    def patched(self, **kw):
        kw["jacobi_sweeps"] = sweeps
        return orig(self, **kw)
    rs.Template.attach_pipeline = patched
    prob = rs.BatchProblem(graph, use_limits=True)
    rs.Template.attach_pipeline = orig
    for _ in range(2):
        prob.template.prepare(Tg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        tg, Y0, K = prob.template.prepare(Tg, return_K=True)
    torch.cuda.synchronize()
    print(name, B, "sweeps cap", sweeps or 10, "prepare ms", (time.perf_counter() - t0) / 5 * 1e3, "K mean", K.float().mean().item(), flush=True)
