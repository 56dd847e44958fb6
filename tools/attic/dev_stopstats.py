import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphik_amd.engine import Template
d = np.load("tests/golden/lwa4d.npz")
T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True)
r = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"]), trace_cap=3000)
tr = r["trace"]
its = r["iterations"].cpu().numpy()
names = ["negcurv", "exceedTR", "lin", "superlin", "maxinner", "model_inc"]
tot = np.zeros(6, int); nit = np.zeros(6, int)
for g in range(len(its)):
    st = tr["stop"][g][:its[g]].cpu().numpy(); nu = tr["numit"][g][:its[g]].cpu().numpy()
    for k in range(6):
        tot[k] += (st == k).sum(); nit[k] += (nu[st == k] + 1).sum()
print(os.environ.get("GIK_LIB_PATH", "default").split("/")[-1], "outer its", its.sum(), "inner", int(r["inner_total"].sum()))
for k in range(6):
    print("   %-10s outer %6d  inner %8d  (%.1f per call)" % (names[k], tot[k], nit[k], nit[k] / max(tot[k], 1)))
