"""dev: wavefront prepare kernel, time up to each phase (dev build, GIK_PREP_STOP)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
os.environ["GIK_LIB_PATH"] = os.path.join(R, "graphik_amd/lib/exp/libgraphik_amd_dev.so")
import numpy as np, torch
from graphik_amd.solvers.riemannian_solver import BatchProblem
sys.path.insert(0, os.path.join(R, "tests"))
from conftest import make_graph
name, B = os.environ.get("ROBOT", "planar10"), int(os.environ.get("B", "65536"))
robot, graph = make_graph(name)
rs = np.random.RandomState(0)
lb, ub = robot.limits_arrays()
Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rs.rand(B, robot.n))).cuda()
names = ["all", "setup+targets", "floyd-warshall", "lower bounds", "gram", "jacobi 1", "factor", "eig count", "scatter", "jacobi 2"]
prev = 0.0
for ph in list(range(1, 10)) + [0]:
    os.environ["GIK_PREP_STOP"] = str(ph)
    prob = BatchProblem(graph, use_limits=True)
    prob.template.prepare(Tg); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); prob.template.prepare(Tg); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    t = min(ts)
    print(f"{name} B={B}: up to and including {names[ph]:>15s}: {t:7.3f} ms (+{t - prev:6.3f})", flush=True); prev = t
