"""dev: round-robin slicing at one wave per SIMD (batches of 1025..6144 goals): kernel ms with and without, several seeds."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from graphik_amd.solvers.riemannian_solver import BatchProblem
from conftest import make_graph
for name, B in (("lwa4d", 4096), ("kuka", 4096), ("ur10", 4096), ("lwa4d", 2048), ("kuka", 6144)):
    robot, graph = make_graph(name)
    lb, ub = robot.limits_arrays()
    probs = {lab: BatchProblem(graph, use_limits=True, params=pr) for lab, pr in (("rr", None), ("plain", {"debug_flags": 512}))}
    for seed in range(4):
        rs = np.random.RandomState(seed)
        Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rs.rand(B, robot.n))).cuda()
        line = f"{name} {B} seed {seed}:"
        outs = {}
        for lab, prob in probs.items():
            tg, Y0 = prob.template.prepare(Tg)
            prob.template.solve(Y0, tg); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); r = prob.template.solve(Y0, tg); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            outs[lab] = r["x"].cpu().numpy()
            line += f"  {lab} {min(ts):7.2f} ms (max {max(ts):7.2f})"
        assert np.array_equal(outs["rr"], outs["plain"])
        print(line, flush=True)
