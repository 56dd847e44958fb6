"""dev: per-iteration traces (f, |grad|, Delta, numit, accept) of the first CAP outer iterations together with the
final iteration counts: raw material for a scheduling predictor of the problems that run to maxiter."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from graphik_amd.solvers.riemannian_solver import BatchProblem
from conftest import make_graph
name, B, CAP = os.environ.get("ROBOT", "kuka"), int(os.environ.get("B", "8192")), int(os.environ.get("CAP", "300"))
robot, graph = make_graph(name)
rs = np.random.RandomState(0)
lb, ub = robot.limits_arrays()
Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rs.rand(B, robot.n))).cuda()
prob = BatchProblem(graph, use_limits=True, params={"debug_flags": 512})
tg, Y0 = prob.template.prepare(Tg)
r = prob.template.solve(Y0, tg, trace_cap=CAP); torch.cuda.synchronize()
tr = r["trace"]
out = {k: v.cpu().numpy().astype(np.float32) if v.dtype.is_floating_point else v.cpu().numpy().astype(np.int16) for k, v in tr.items()}
out["iterations"] = r["iterations"].cpu().numpy(); out["inner_total"] = r["inner_total"].cpu().numpy(); out["stop"] = r["stop"].cpu().numpy()
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed(f"gpurun_out/predict_{name}_{B}.npz", **out)
print(name, B, {k: v.shape for k, v in out.items()}, "maxiter frac", (out["stop"] == 1).mean())
