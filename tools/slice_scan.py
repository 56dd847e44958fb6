"""Round-robin slice length of the wavefront kernel against time and hand-overs (HBM traffic): KUKA, B goals.
   python tools/slice_scan.py [B] [slices...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import make_graph
from graphik_amd.solvers.riemannian_solver import BatchProblem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
slices = [int(a) for a in sys.argv[2:]] or [256, 384, 512, 768, 1024]
robot, graph = make_graph(os.environ.get("ROBOT", "kuka"))
rs = np.random.RandomState(0)
lb, ub = robot.limits_arrays()
Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rs.rand(B, robot.n))).cuda()
for sl in slices:
    prob = BatchProblem(graph, use_limits=True, params={"slice_outer_its": sl})
    tpl = prob.template
    tg, Y0 = tpl.prepare(Tg)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        r = tpl.solve(Y0, tg)
        torch.cuda.synchronize(); best = min(best, time.time() - t0)
    ho = int((r["flags"].cpu().numpy().astype(np.int64) >> 8).sum())
    alg = B * (8 * (tpl.T + 2 * tpl.N * tpl.k) + 48)
    per = 8 * (tpl.T + 2 * tpl.N * tpl.k) + 2 * 32
    print(f"slice {sl:5d}: {best*1e3:7.1f} ms  {B/best:9.0f} solves/s  hand-overs {ho:7d}  est. traffic {(alg + ho*per)/1e6:6.0f} MB = {(alg + ho*per)/alg:.2f} x algorithmic", flush=True)
