"""dev: row-form vs column-form Hessian product (two builds of the library, GIK_LIB_PATH): sha256 of
the solver outputs on 2048 KUKA goals (must be equal across builds) and kernel ms at several batch sizes."""
import sys, os, hashlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphik_amd.utils.roboturdf import load_kuka, load_schunk_lwa4d
from graphik_amd.solvers.riemannian_solver import BatchProblem
tag = os.environ.get("GIK_LIB_PATH", "default").split("/")[-1]
for name, ld, sizes in (("kuka", load_kuka, (2048, 8192, 65536)), ("lwa4d", load_schunk_lwa4d, (4096, 16384))):
    robot, graph = ld()
    prob = BatchProblem(graph, use_limits=True)
    rs = np.random.RandomState(0)
    lb, ub = robot.limits_arrays()
    U = rs.rand(max(sizes), robot.n)
    for B in sizes:
        Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * U[:B])).cuda()
        tg, Y0 = prob.template.prepare(Tg)
        r = prob.template.solve(Y0, tg); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = prob.template.solve(Y0, tg); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        h = hashlib.sha256(r["x"].cpu().numpy().tobytes() + r["iterations"].cpu().numpy().tobytes()
                           + r["inner_total"].cpu().numpy().tobytes()).hexdigest()[:16]
        print(f"{tag} {name} B={B}: kernel ms {[round(t, 1) for t in ts]} sha {h}", flush=True)
