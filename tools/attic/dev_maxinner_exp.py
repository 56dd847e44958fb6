import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphik_amd.utils.roboturdf import load_ur10
from graphik_amd.solvers.riemannian_solver import BatchProblem
from graphik_amd.engine import Template
robot, graph = load_ur10()
prob = BatchProblem(graph, use_limits=True)
B = 1024
rng = np.random.RandomState(3)
Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(B, robot.n))
targets, Y0 = prob.prepare(Tg)
r = prob.template.solve(Y0, targets, trace_cap=3000); torch.cuda.synchronize()
st = r["trace"]["stop"].cpu().numpy(); its = r["iterations"].cpu().numpy()
n4 = sum(int((st[b, :its[b]] == 4).sum()) for b in range(B))
print(os.environ.get("GIK_LIB_PATH", "default").split("/")[-1], "maxinner calls", n4, "mean Hv", r["inner_total"].double().mean().item(), "mean its", its.mean(), "converged", (r["stop"] == 0).double().mean().item())
