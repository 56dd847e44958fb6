"""dev: throughput of B = 4096 batches when several are in flight on separate HIP streams (the straggler tail
of one batch overlaps with the bulk of the next).  Not the bench line: bench.py runs its steps back to back."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphik_amd.utils.roboturdf import load_schunk_lwa4d
from graphik_amd.solvers.riemannian_solver import BatchProblem
robot, graph = load_schunk_lwa4d()
dev = torch.device("cuda", 0)
prob = BatchProblem(graph, use_limits=True, device=dev)
tpl = prob.template
B, K = 4096, 16
lb, ub = robot.limits_arrays()
goals = []
for s in range(K):
    rs = np.random.RandomState(100 + s)
    goals.append(torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rs.rand(B, robot.n))).to(dev))
def one(Tg):
    targets, Y0 = tpl.prepare(Tg)
    res = tpl.solve(Y0, targets)
    q, pe, re = tpl.recover(res["x"], Tg)
    return pe
one(goals[0]); torch.cuda.synchronize()
for S in (1, 2, 3, 4, 8):
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = []
    for i in range(K):
        with torch.cuda.stream(streams[i % S]):
            outs.append(one(goals[i]))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = np.mean([(o < 0.01).double().mean().item() for o in outs])
    print("streams %d: %d batches of %d in %.1f ms -> %.0f solves/s (pos<0.01: %.3f)" % (S, K, B, dt * 1e3, K * B / dt, ok), flush=True)
