import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphik_amd.engine import Template
d = np.load("tests/golden/ur10_table.npz")
T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True)
for B in (1, 256, 1024):
    Yi = np.tile(d["Y_init"], (B, 1, 1)); tg = np.tile(T.targets_from_D(d["D_goal"]), (B, 1))
    r = T.solve(Yi, tg); torch.cuda.synchronize()
    t0 = time.time(); r = T.solve(Yi, tg); torch.cuda.synchronize(); dt = time.time() - t0
    inner = int(r["inner_total"][0])
    print("ur10_table B=%d: %.3f s, its %d inner %d -> %.2f us per Hv per problem; %.1f solves/s" % (B, dt, int(r["iterations"][0]), inner, dt / inner * 1e6, B / dt), flush=True)
