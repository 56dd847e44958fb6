import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphik_amd.engine import Template
d = np.load("tests/golden/lwa4d.npz")
T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True, params=dict(maxiter=400))
i = 3
out = []
for B in [1, 1024]:
    Yi = torch.from_numpy(np.tile(d["Y_init"][i:i+1], (B, 1, 1))).cuda()
    tg = torch.from_numpy(np.tile(T.targets_from_D(d["D_goal"][i:i+1]), (B, 1))).cuda()
    r = T.solve(Yi, tg); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.time(); r = T.solve(Yi, tg); torch.cuda.synchronize(); ts.append(time.time() - t0)
    inner = int(r["inner_total"][0]); dt = min(ts)
    out.append("B=%d %.3f us/iter (inner %d)" % (B, dt / inner * 1e6, inner))
print(os.environ.get("GIK_LIB_PATH", "default").split("/")[-1], " | ".join(out), flush=True)
# quick parity check of this build
def rel(a, b): return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
tg0 = T.targets_from_D(d["D_goal"][0])
print("   KAT rel: cost %.1e grad %.1e hess %.1e proj %.1e" % (
    rel(T.cost(d["kat_Y"], tg0).cpu().numpy(), d["kat_lim_loop_cost"]), rel(T.grad(d["kat_Y"], tg0).cpu().numpy(), d["kat_lim_loop_grad"]),
    rel(T.hess(d["kat_Y"], d["kat_W"], tg0).cpu().numpy(), d["kat_lim_loop_hess"]), rel(T.proj(d["kat_Y"], d["kat_W"]).cpu().numpy(), d["kat_proj"])), flush=True)
T2 = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True)
r = T2.solve(d["Y_init"], T2.targets_from_D(d["D_goal"]))
print("   its", r["iterations"].cpu().numpy().tolist(), flush=True)
