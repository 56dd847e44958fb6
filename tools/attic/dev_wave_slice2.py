"""dev: round-robin slicing, kernel ms by (GIK_SLICE, GIK_SLICE_CYCLES)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphik_amd.solvers.riemannian_solver import BatchProblem
from graphik_amd.utils.roboturdf import load_schunk_lwa4d, load_ur10, load_kuka
name, B = os.environ.get("ROBOT", "kuka"), int(os.environ.get("B", "8192"))
robot, graph = {"kuka": load_kuka, "lwa4d": load_schunk_lwa4d, "ur10": load_ur10}[name]()
rs = np.random.RandomState(0)
lb, ub = robot.limits_arrays()
Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rs.rand(B, robot.n))).cuda()
print(name, B)
for cfg in os.environ.get("CFGS", "0:0 32:16000000 32:8000000 32:4000000").split():
    sl, cyc, *age = cfg.split(":")
    os.environ["GIK_YIELD_AGE"] = age[0] if age else str(1 << 30)
    os.environ["GIK_SLICE"] = sl; os.environ["GIK_SLICE_CYCLES"] = cyc
    prob = BatchProblem(graph, use_limits=True, params={"slice_outer_its": int(sl)})
    tg, Y0 = prob.template.prepare(Tg)
    r = prob.template.solve(Y0, tg); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = prob.template.solve(Y0, tg); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    fl = r["flags"].cpu().numpy()
    print(f"slice {sl:>4s} cycles {cyc:>9s} age {os.environ['GIK_YIELD_AGE']:>10s}: {min(ts):7.1f} ms (median {sorted(ts)[1]:7.1f}); hand-overs {int((fl >> 8).sum())}", flush=True)
