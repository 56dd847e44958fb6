import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import make_graph
from graphik_amd.solvers.riemannian_solver import BatchProblem
robot, graph = make_graph("kuka")
rs = np.random.RandomState(0)
lb, ub = robot.limits_arrays()
Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rs.rand(65536, robot.n))).cuda()
prob = BatchProblem(graph, use_limits=True)
tg, Y0 = prob.template.prepare(Tg)
r = prob.template.solve(Y0, tg)
inner = r["inner_total"].cpu().numpy().astype(np.int64); st = r["stop"].cpu().numpy()
print("maxiter share of goals %.4f, of products %.4f; mean products maxiter %.0f others %.0f" % (np.mean(st == 1), inner[st == 1].sum() / inner.sum(), inner[st == 1].mean(), inner[st != 1].mean()))
