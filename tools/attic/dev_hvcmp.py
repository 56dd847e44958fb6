import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import c_oracle as co
from graphik_amd.engine import Template
for name in ("lwa4d", "kuka", "ur10"):
    d = np.load(f"tests/golden/{name}.npz")
    T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True)
    r = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"]))
    o = co.rtr_solve_batch(d["Y_init"], d["D_goal"], d["omega"], d["psi_L"], d["psi_U"], True, fast=False)
    print(name, "reference: its", int(d["iterations"].sum()), "hv", int(d["hv_total"].sum()),
          "| oracle: its", int(o["iterations"].sum()), "inner", int(o["inner_total"].sum()),
          "| GPU: its", int(r["iterations"].sum()), "inner", int(r["inner_total"].sum()))
    print("   per goal ref its ", d["iterations"].tolist())
    print("   per goal GPU its ", r["iterations"].cpu().numpy().tolist())
    print("   per goal ref hv/its ", np.round(d["hv_total"] / d["iterations"], 1).tolist())
    print("   per goal orc hv/its ", np.round(o["inner_total"] / o["iterations"], 1).tolist())
    print("   per goal GPU hv/its ", np.round(r["inner_total"].cpu().numpy() / r["iterations"].cpu().numpy(), 1).tolist())
