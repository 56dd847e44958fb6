"""dev: per-problem excess of Hessian products, GPU vs oracle, and where it sits."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as co
from graphik_amd.utils.roboturdf import load_schunk_lwa4d, load_kuka, load_ur10
from graphik_amd.solvers.riemannian_solver import BatchProblem
name = sys.argv[1] if len(sys.argv) > 1 else "ur10"
robot, graph = {"lwa4d": load_schunk_lwa4d, "kuka": load_kuka, "ur10": load_ur10}[name]()
prob = BatchProblem(graph, use_limits=True)
B = 1024
rng = np.random.RandomState(3)
Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(B, robot.n))
targets, Y0 = prob.prepare(Tg)
r = prob.template.solve(Y0, targets, trace_cap=3000); torch.cuda.synchronize()
D, _, _ = prob.assemble(Tg)
its_g = r["iterations"].cpu().numpy(); hv_g = r["inner_total"].cpu().numpy()
numit = r["trace"]["numit"].cpu().numpy(); stop = r["trace"]["stop"].cpu().numpy(); fb = r["trace"]["f_before"].cpu().numpy()
names = ["negcurv", "exceedTR", "lin", "superlin", "maxinner", "model_inc"]
G = {k: [0, 0] for k in range(6)}; O = {k: [0, 0] for k in range(6)}
tot_o_its = 0
for g in range(256):
    o = co.rtr_solve(Y0[g], D[g], prob.omega, prob.psi_L, prob.psi_U, True, traj_cap=3000)
    m = int(o["iterations"]); tot_o_its += m
    for k in range(6):
        sel = np.asarray(o["traj"]["stop"][:m]) == k
        O[k][0] += sel.sum(); O[k][1] += (np.asarray(o["traj"]["numit"][:m])[sel] + 1).sum()
    n = its_g[g]
    for k in range(6):
        sel = stop[g][:n] == k
        G[k][0] += sel.sum(); G[k][1] += (numit[g][:n][sel] + 1).sum()
print(name, "first 256 goals: outer its GPU %d oracle %d" % (its_g[:256].sum(), tot_o_its))
for k in range(6):
    print("   %-10s GPU calls %6d inner %8d (%.1f)   oracle calls %6d inner %8d (%.1f)" % (
        names[k], G[k][0], G[k][1], G[k][1] / max(G[k][0], 1), O[k][0], O[k][1], O[k][1] / max(O[k][0], 1)))
