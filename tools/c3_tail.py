"""The TAIL of the table scene (BASELINE configs[2], UR10 + table_environment(), N = 116) against the CPU oracle on the
scene itself: from a 4096-goal run of the device pipeline the NL longest goals and NR random ones are solved by the
oracle from the device's start points (1-25 s each per thread).   python tools/c3_tail.py [NL] [NR] [path: 0 | 1 | 2]
-> gpurun_out/c3_tail.json"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from conftest import make_graph
from oracle import c_oracle as co
from graphik_amd.solvers.riemannian_solver import BatchProblem

NL = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 64
path = int(sys.argv[3]) if len(sys.argv) > 3 else 0
robot, graph = make_graph("ur10_table")
prob = BatchProblem(graph, use_limits=True, params=({"force_block_path": path} if path else None))
B = 4096
rs = np.random.RandomState(0)
lb, ub = robot.limits_arrays()
Tg = robot.fk_batch(lb + (ub - lb) * rs.rand(B, robot.n))
tpl = prob.template
tg, Y0 = tpl.prepare(Tg)
r = tpl.solve(Y0, tg)
torch.cuda.synchronize()
its = r["iterations"].cpu().numpy(); hv = r["inner_total"].cpu().numpy().astype(np.int64); f = r["f"].cpu().numpy()
order = np.argsort(-its, kind="stable")
longest = order[:NL]
rest = np.setdiff1d(np.arange(B), longest)
rnd = np.random.RandomState(1).choice(rest, NR, replace=False)
idx = np.concatenate([longest, rnd])
D, _, _ = prob.assemble(Tg[idx])
t0 = time.time()
o = co.rtr_solve_batch(Y0.cpu().numpy()[idx], D, prob.omega, prob.psi_L, prob.psi_U, True, fast=False)
dt = time.time() - t0
oi, oh = o["iterations"], o["inner_total"].astype(np.int64)


def stats(sel):
    a, b = its[idx][sel], oi[sel]
    mx, mxo = a >= 3000, b >= 3000
    cv = (f[idx][sel] < 1e-9) == (o["f(x)"][sel] < 1e-9)
    both = ~mx & ~mxo
    return {"n": int(sel.sum()), "p90": [float(np.percentile(a, 90)), float(np.percentile(b, 90))],
            "median": [float(np.median(a)), float(np.median(b))], "at_maxiter": [int(mx.sum()), int(mxo.sum())],
            "to_maxiter": int((mx & ~mxo).sum()), "from_maxiter": int((~mx & mxo).sum()),
            "same_maxiter_class": float(np.mean(mx == mxo)), "same_convergence_class": float(cv.mean()),
            "hv_ratio": float(hv[idx][sel].sum() / oh[sel].sum()),
            "hv_ratio_neither_at_maxiter": float(hv[idx][sel][both].sum() / max(1, oh[sel][both].sum())),
            "outer_ratio_neither_at_maxiter": float(a[both].sum() / max(1, b[both].sum()))}


sel_all = np.ones(len(idx), bool); sel_l = np.arange(len(idx)) < NL
out = {"kernel": tpl.info, "goals": B, "oracle_seconds": dt, "all": stats(sel_all), "longest": stats(sel_l), "random": stats(~sel_l),
       "device_at_maxiter_of_4096": int((its >= 3000).sum())}
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(R, "gpurun_out", f"c3_tail_path{path}.json"), "w"), indent=1)
