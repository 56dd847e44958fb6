"""dev: wavefront kernel, round-robin slicing: kernel ms and share of problems that were ever resumed,
by slice length (GIK_SLICE) on KUKA 8192."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphik_amd.utils.roboturdf import load_kuka
from graphik_amd.solvers.riemannian_solver import BatchProblem
from graphik_amd.utils.roboturdf import load_schunk_lwa4d, load_ur10
name, B = os.environ.get("ROBOT", "kuka"), int(os.environ.get("B", "8192"))
robot, graph = {"kuka": load_kuka, "lwa4d": load_schunk_lwa4d, "ur10": load_ur10}[name]()
rs = np.random.RandomState(0)
lb, ub = robot.limits_arrays()
Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rs.rand(B, robot.n))).cuda()
print(name, B)
for sl in os.environ.get("SLICES", "0 2048 512 128 64").split():
    os.environ["GIK_SLICE"] = sl
    prob = BatchProblem(graph, use_limits=True, params={"slice_outer_its": int(sl), "debug_flags": int(os.environ.get("DBG", "0"))})
    tg, Y0 = prob.template.prepare(Tg)
    r = prob.template.solve(Y0, tg); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = prob.template.solve(Y0, tg); e1.record(); torch.cuda.synchronize()
    fl = r["flags"].cpu().numpy(); its = r["iterations"].cpu().numpy()
    nres = fl >> 8
    print(f"slice {sl:>5s}: {e0.elapsed_time(e1):7.1f} ms; resumed {((fl & 2) != 0).mean():.3f} of problems; "
          f"hand-overs total {int(nres.sum())} (expected from slices {int((its // max(int(sl), 1)).sum()) if int(sl) else 0}), max per problem {int(nres.max())}; "
          f"executed/total products {r['inner_executed'].sum().item() / r['inner_total'].sum().item():.3f}", flush=True)
