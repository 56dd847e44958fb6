import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphik_amd.engine import Template
def rel(a, b): return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
for nm in ["planar10_nolimits", "lwa4d"]:
    d = np.load(f"tests/golden/{nm}.npz")
    use_lim = bool(int(d["use_limits"])); k = int(d["dim"])
    key = "lim" if use_lim else "nolim"
    T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=k, use_limits=use_lim, params=dict(maxiter=6, maxinner=40))
    print(nm, "waves/cu", "created", flush=True)
    tg = T.targets_from_D(d["D_goal"][0]); Y, W = d["kat_Y"], d["kat_W"]
    c = T.cost(Y, tg).cpu().numpy(); print(" cost", rel(c, d[f"kat_{key}_loop_cost"]), flush=True)
    g = T.grad(Y, tg).cpu().numpy(); print(" grad", rel(g, d[f"kat_{key}_loop_grad"]), flush=True)
    h = T.hess(Y, W, tg).cpu().numpy(); print(" hess", rel(h, d[f"kat_{key}_loop_hess"]), flush=True)
    pj = T.proj(Y, W).cpu().numpy(); print(" proj", rel(pj, d["kat_proj"]), flush=True)
    r = T.solve(d["Y_init"][:2], T.targets_from_D(d["D_goal"][:2]), trace_cap=8); torch.cuda.synchronize()
    print(" solve(maxiter=6):", r["iterations"].tolist(), r["inner_total"].tolist(), r["f"].tolist(), flush=True)
    print("  numit", r["trace"]["numit"].tolist(), "ref", d["loop_traj_numit"][:2,:6].tolist(), flush=True)
    print("  stop", r["trace"]["stop"].tolist(), "ref", d["loop_traj_stop"][:2,:6].tolist(), flush=True)
