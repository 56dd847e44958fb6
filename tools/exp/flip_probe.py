import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from conftest import load_golden
from oracle import c_oracle as co
from graphik_amd.engine import Template
d = load_golden("kuka")
g = 8
o = co.rtr_solve(d["Y_init"][g], d["D_goal"][g], d["omega"], d["psi_L"], d["psi_U"], True, traj_cap=16)
print("oracle", {k: np.asarray(v)[5:10].tolist() for k, v in o["traj"].items()})
for path, params in (("wave", {}), ("wave_column", {"hessian_form": "column"}), ("block", {"force_block_path": 1}), ("npt", {"force_block_path": 2})):
    T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True, params=params)
    r = T.solve(d["Y_init"][g:g+1], T.targets_from_D(d["D_goal"][g:g+1]), trace_cap=16)
    print(path, {k: v[0].cpu().numpy()[5:10].tolist() for k, v in r["trace"].items()})
