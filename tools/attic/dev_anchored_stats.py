"""dev: who converges in the fixed-anchor formulation of UR10 + table (4096 random goals)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import make_graph
from graphik_amd.solvers.riemannian_solver import AnchoredProblem, solve_batch
robot, graph = make_graph("ur10_table")
ap = AnchoredProblem(graph)
rng = np.random.RandomState(0)
B = 4096
lb, ub = robot.limits_arrays()
Q = lb + (ub - lb) * rng.rand(B, robot.n)
Tg = robot.fk_batch(Q)
r = ap.solve(Tg)
f = r["f"].cpu().numpy(); Y = r["x"].cpu().numpy(); pos = r["pos_err"].cpu().numpy(); rot = r["rot_err"].cpu().numpy()
conv = f < 1e-9
P_goal = np.stack([robot.fk_batch(Q, i)[:, :3, 3] for i in range(1, robot.n + 1)], axis=1)
d = np.linalg.norm(P_goal[:, :, None, :] - ap.obstacles[None, None, :, :3], axis=-1) - ap.obstacles[None, None, :, 3]
clear_goal = d.min(axis=(1, 2))                      # clearance of the generating configuration (p1..p6)
ee_clear = d[:, -1].min(axis=1)                      # ... of the end effector alone (a constant of the problem)
free_goal = clear_goal > 0
print("goals whose generating configuration is collision free: %.3f" % free_goal.mean())
print("  converged | collision-free generating configuration: %.3f" % conv[free_goal].mean())
print("  converged | generating configuration collides:       %.3f" % conv[~free_goal].mean())
print("  converged | EE itself inside a sphere:                %.3f  (share of goals %.3f)" % (conv[ee_clear < 0].mean(), (ee_clear < 0).mean()))
print("overall converged %.3f; collision free among converged %.3f; success (pos,rot<0.01) %.3f" % (
    conv.mean(), (ap.clearance(Y)[conv] > -1e-4).mean(), ((pos < 0.01) & (rot < 0.01)).mean()))
# the reference-semantics pipeline on the bare arm ignores the table: how many of ITS solutions collide?
_, bare = make_graph("ur10")
q, Yb, info = solve_batch(bare, Tg)
Pb = np.stack([robot.fk_batch(q, i)[:, :3, 3] for i in range(1, robot.n)], axis=1)      # p1..p5
db = (np.linalg.norm(Pb[:, :, None, :] - ap.obstacles[None, None, :, :3], axis=-1) - ap.obstacles[None, None, :, 3]).min(axis=(1, 2))
okb = (info["pos_err"] < 0.01) & (info["rot_err"] < 0.01)
print("reference semantics (obstacles ignored by the cost): success %.3f, of which collision free %.3f" % (okb.mean(), (db[okb] > -1e-4).mean()))
