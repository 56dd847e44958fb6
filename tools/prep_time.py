"""Time of the planar prepare kernel (c5: 65536 planar-10 goals) by torch events over N launches, plus a digest
of its outputs -- the quick A/B loop of the round-5 LDS-layout work (NOTEBOOK 10.1).
    python tools/prep_time.py [robot] [B] [reps]"""
import hashlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from conftest import make_graph
from graphik_amd.solvers.riemannian_solver import BatchProblem
name = sys.argv[1] if len(sys.argv) > 1 else "planar10_limits_pi"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
robot, graph = make_graph(name)
prob = BatchProblem(graph, use_limits=not name.endswith("nolimits"))
rs = np.random.RandomState(0)
lb, ub = robot.limits_arrays()
Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rs.rand(B, robot.n))).cuda()
tpl = prob.template
ms = []
for rep in range(reps + 3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    tg, Y0, K = tpl.prepare(Tg, return_K=True)
    e1.record(); torch.cuda.synchronize()
    if rep >= 3: ms.append(e0.elapsed_time(e1))
G = Y0 @ Y0.transpose(1, 2)
print("%s B=%d prepare: min %.3f median %.3f ms | K median %d | sum|Gram| %.12e | tg sha %s" % (
    name, B, min(ms), float(np.median(ms)), int(K.median()), float(G.abs().sum()),
    hashlib.sha256(tg.cpu().numpy().tobytes()).hexdigest()[:12]))
