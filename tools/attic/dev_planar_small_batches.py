import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import make_graph
from graphik_amd.solvers.riemannian_solver import BatchProblem
robot, graph = make_graph("planar10_limits_pi")
rs = np.random.RandomState(0)
lb, ub = robot.limits_arrays()
Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rs.rand(4096, robot.n))).cuda()
tag = "wave" if os.environ.get("GIK_NO_PREP_QUAD") else "quad"
prob = BatchProblem(graph, use_limits=True, params={"debug_flags": 8192} if tag == "wave" else None)
tpl = prob.template
for B in (1, 4, 16, 64, 256, 1024, 4096):
    T = Tg[:B].contiguous()
    tp, ts = [], []
    for rep in range(12):
        e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e0.record(); tg, Y0 = tpl.prepare(T); e1.record(); r = tpl.solve(Y0, tg); e2.record(); torch.cuda.synchronize()
        tp.append(e0.elapsed_time(e1)); ts.append(e1.elapsed_time(e2))
    print(f"{tag} B={B:5d}: prepare {np.median(tp)*1e3:7.1f} us  solve {np.median(ts)*1e3:7.1f} us", flush=True)
