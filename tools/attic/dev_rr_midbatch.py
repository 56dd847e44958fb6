"""dev: mid-size batches (1025..6144 goals): one wave per SIMD (default) against two waves per SIMD with tail
spreading (debug_flags 1024) and with round-robin slicing on top (0); kernel ms, several seeds."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from graphik_amd.solvers.riemannian_solver import BatchProblem
from conftest import make_graph
cfgs = (("1w", {"waves_per_cu": 4, "debug_flags": 512}), ("2w+spread", {"waves_per_cu": 8, "debug_flags": 1024}), ("2w+spread+rr", {"waves_per_cu": 8}))
for name, B in [tuple(x.split(":")) for x in os.environ.get("CASES", "lwa4d:4096 kuka:4096 ur10:4096 lwa4d:2048 lwa4d:3000 kuka:6144").split()]:
    B = int(B)
    robot, graph = make_graph(name)
    lb, ub = robot.limits_arrays()
    probs = {lab: BatchProblem(graph, use_limits=True, params=pr) for lab, pr in cfgs}
    tot = {lab: 0.0 for lab, _ in cfgs}
    for seed in range(6):
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
            line += f"  {lab} {np.median(ts):7.2f}"
            tot[lab] += np.median(ts)
        assert all(np.array_equal(outs["1w"], v) for v in outs.values())
        print(line, flush=True)
    print(f"{name} {B} mean:", {k: round(v / 6, 2) for k, v in tot.items()}, flush=True)
