"""dev: UR10 + table scene on the workgroup-per-problem kernel, clique closed form on (default) or
off (GIK_DBG=128): time per Hessian product, result agreement.  Usage: dev_table_ab.py [B]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphik_amd.engine import Template
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = np.load("tests/golden/ur10_table.npz")
out = {}
for flags in (128, 256, 0):
    T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True,
                               params={"debug_flags": flags})
    G = len(d["Y_init"])
    Yi = np.tile(d["Y_init"], (B // G, 1, 1)); tg = np.tile(T.targets_from_D(d["D_goal"]), (B // G, 1))
    r = T.solve(Yi, tg); torch.cuda.synchronize()
    t0 = time.time(); r = T.solve(Yi, tg); torch.cuda.synchronize(); dt = time.time() - t0
    ex = r["inner_executed"].cpu().numpy().astype(float)
    print("flags %s B=%d: %.3f s, executed products %.3g total, max %d -> %.2f us per product per CU-resident problem; %.1f solves/s"
          % (str(flags).rjust(3), len(Yi), dt, ex.sum(), ex.max(), dt * min(len(Yi), 256) / ex.sum() * 1e6, len(Yi) / dt), flush=True)
    out[flags] = {k: r[k].cpu().numpy() for k in ("x", "f", "iterations", "inner_total", "stop")}
a, b = out[128], out[0]
print("iterations off/on:", a["iterations"][:G], b["iterations"][:G])
print("f off/on:", a["f"][:G], b["f"][:G])
print("max |x_off - x_on| per goal:", np.abs(a["x"][:G] - b["x"][:G]).reshape(G, -1).max(1))
