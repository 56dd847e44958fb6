"""dev: mean tCG iteration count per call, binned by the cost at the start of the outer iteration,
GPU against the oracle (same 16 golden LWA4D goals)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as co
from graphik_amd.engine import Template
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests/golden/lwa4d.npz"))
T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True)
r = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"]), trace_cap=3000)
its = r["iterations"].cpu().numpy()
G = {"f": [], "n": [], "s": []}; O = {"f": [], "n": [], "s": []}
for g in range(len(its)):
    n = its[g]
    G["f"] += r["trace"]["f_before"][g][:n].cpu().numpy().tolist(); G["n"] += (r["trace"]["numit"][g][:n].cpu().numpy() + 1).tolist(); G["s"] += r["trace"]["stop"][g][:n].cpu().numpy().tolist()
    o = co.rtr_solve(d["Y_init"][g], d["D_goal"][g], d["omega"], d["psi_L"], d["psi_U"], True, traj_cap=3000)
    m = int(o["iterations"])
    O["f"] += list(o["traj"]["f_before"][:m]); O["n"] += list(np.asarray(o["traj"]["numit"][:m]) + 1); O["s"] += list(o["traj"]["stop"][:m])
for X in (G, O):
    for k in X: X[k] = np.asarray(X[k], dtype=float)
edges = [1e3, 1, 1e-2, 1e-4, 1e-6, 1e-8, 1e-10, 1e-12, 1e-14, 1e-16, 1e-18, 1e-22, 0]
print("%-22s %28s %28s" % ("f at start of outer it", "GPU calls / mean inner", "oracle calls / mean inner"))
for hi, lo in zip(edges[:-1], edges[1:]):
    a = (G["f"] <= hi) & (G["f"] > lo); b = (O["f"] <= hi) & (O["f"] > lo)
    print("(%7.0e, %7.0e]  %10d / %6.1f (superlin %4d / %6.1f)   %10d / %6.1f (superlin %4d / %6.1f)" % (
        lo, hi, a.sum(), G["n"][a].mean() if a.any() else 0, (a & (G["s"] == 3)).sum(), G["n"][a & (G["s"] == 3)].mean() if (a & (G["s"] == 3)).any() else 0,
        b.sum(), O["n"][b].mean() if b.any() else 0, (b & (O["s"] == 3)).sum(), O["n"][b & (O["s"] == 3)].mean() if (b & (O["s"] == 3)).any() else 0))

# second view: per bin, the gradient norm after the step and the trust-region radius
Gg, Go, Gd, Od = [], [], [], []
for g in range(len(its)):
    n = its[g]
    Gg += r["trace"]["gradnorm_after"][g][:n].cpu().numpy().tolist(); Gd += r["trace"]["Delta"][g][:n].cpu().numpy().tolist()
    o = co.rtr_solve(d["Y_init"][g], d["D_goal"][g], d["omega"], d["psi_L"], d["psi_U"], True, traj_cap=3000)
    m = int(o["iterations"])
    Go += list(o["traj"]["gradnorm_after"][:m]); Od += list(o["traj"]["Delta"][:m])
Gg, Go, Gd, Od = map(np.asarray, (Gg, Go, Gd, Od))
print("%-22s %30s %30s" % ("f bin", "GPU median |g| after / Delta", "oracle median |g| after / Delta"))
for hi, lo in zip(edges[:-1], edges[1:]):
    a = (G["f"] <= hi) & (G["f"] > lo); b = (O["f"] <= hi) & (O["f"] > lo)
    if a.any() and b.any():
        print("(%7.0e, %7.0e]   %10.2e / %8.2e      %10.2e / %8.2e" % (lo, hi, np.median(Gg[a]), np.median(Gd[a]), np.median(Go[b]), np.median(Od[b])))
