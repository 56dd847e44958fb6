import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphik_amd.engine import Template
nm = sys.argv[1]
d = np.load(f"tests/golden/{nm}.npz")
use_lim = bool(int(d["use_limits"])); k = int(d["dim"])
T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=k, use_limits=use_lim, params=dict(maxiter=6, maxinner=40))
r = T.solve(d["Y_init"][:2], T.targets_from_D(d["D_goal"][:2]), trace_cap=8); torch.cuda.synchronize()
print(nm, "DBG", os.environ.get("GIK_DBG"), "solve:", r["iterations"].tolist(), r["inner_total"].tolist(), r["f"].tolist(), r["stop"].tolist(), flush=True)
print("  numit", r["trace"]["numit"][:, :6].tolist(), "ref", d["loop_traj_numit"][:2,:6].tolist(), flush=True)
