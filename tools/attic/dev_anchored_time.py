"""dev: anchored pipeline time with and without the obstacle hinges (same goals)."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import make_graph
from graphik_amd.solvers.riemannian_solver import AnchoredProblem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
robot, g_table = make_graph("ur10_table")
_, g_bare = make_graph("ur10")
rng = np.random.RandomState(0)
lb, ub = robot.limits_arrays()
Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rng.rand(B, robot.n))).cuda()
for name, g in (("bare", g_bare), ("table", g_table)):
    ap = AnchoredProblem(g)
    for _ in range(2):
        r = ap.template.anchored_ik(ap.base.template, Tg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r = ap.template.anchored_ik(ap.base.template, Tg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    its = r["iterations"].double()
    print(name, "ms/batch %.1f" % (dt * 1e3), "solve kernel ms %.1f" % ap.template.lib.gik_anchored_last_solve_ms(ap.template._h),
          "outer its median %.0f max %.0f  hv/outer %.1f  maxiter frac %.4f" % (its.median().item(), its.max().item(),
          (r["inner_total"].double().sum() / its.sum()).item(), (r["stop"] == 1).double().mean().item()), flush=True)
