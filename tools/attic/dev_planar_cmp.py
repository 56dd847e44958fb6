import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphik_amd.engine import Template
from oracle import c_oracle as co
np.set_printoptions(linewidth=200)
for nm in ["planar10_nolimits", "planar10_limits_halfpi"]:
    d = np.load(f"tests/golden/{nm}.npz"); use_lim = bool(int(d["use_limits"]))
    T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=2, use_limits=use_lim)
    r = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"]), trace_cap=48)
    tr = {k: v.cpu().numpy() for k, v in r["trace"].items()}
    its = r["iterations"].cpu().numpy()
    for g in range(len(d["seed"])):
        o = co.rtr_solve(d["Y_init"][g], d["D_goal"][g], d["omega"], d["psi_L"], d["psi_U"], use_lim, traj_cap=48)
        n = o["iterations"]
        if int(its[g]) != n or not np.array_equal(tr["numit"][g][:n], o["traj"]["numit"]):
            print(nm, g, "gpu its", its[g], "oracle", n)
            print("  numit gpu", tr["numit"][g][:its[g]]); print("  numit ora", o["traj"]["numit"])
            print("  stop  gpu", tr["stop"][g][:its[g]]); print("  stop  ora", o["traj"]["stop"])
            print("  acc   gpu", tr["accept"][g][:its[g]]); print("  acc   ora", o["traj"]["accept"])
            print("  f gpu", tr["f_before"][g][:its[g]]); print("  f ora", o["traj"]["f_before"])
            print("  gn gpu", tr["gradnorm_after"][g][:its[g]]); print("  gn ora", o["traj"]["gradnorm_after"])
