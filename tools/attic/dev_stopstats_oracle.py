import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import c_oracle as co
d = np.load("tests/golden/lwa4d.npz")
names = ["negcurv", "exceedTR", "lin", "superlin", "maxinner", "model_inc"]
tot = np.zeros(6, int); nit = np.zeros(6, int); outer = 0
for g in range(len(d["Y_init"])):
    o = co.rtr_solve(d["Y_init"][g], d["D_goal"][g], d["omega"], d["psi_L"], d["psi_U"], True, traj_cap=3000)
    tr = o["traj"]; n = int(o["iterations"])
    st = np.asarray(tr["stop"][:n]); nu = np.asarray(tr["numit"][:n])
    outer += n
    for k in range(6):
        tot[k] += (st == k).sum(); nit[k] += (nu[st == k] + 1).sum()
print("oracle outer its", outer, "inner", nit.sum())
for k in range(6):
    print("   %-10s outer %6d  inner %8d  (%.1f per call)" % (names[k], tot[k], nit[k], nit[k] / max(tot[k], 1)))
