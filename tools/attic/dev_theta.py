import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from graphik_amd.engine import Template
from oracle import c_oracle as co
d = np.load("tests/golden/lwa4d.npz")
for theta in (1.0, 0.5):
    T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True, params=dict(theta=theta, maxiter=40))
    r = T.solve(d["Y_init"][:4], T.targets_from_D(d["D_goal"][:4]), trace_cap=40)
    print("theta", theta, "its", r["iterations"].cpu().numpy(), "inner", r["inner_total"].cpu().numpy(), "numit[0][:8]", r["trace"]["numit"][0][:8].cpu().numpy())
