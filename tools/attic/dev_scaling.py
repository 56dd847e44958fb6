import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphik_amd.engine import Template
d = np.load("tests/golden/lwa4d.npz")
T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True)
i = 2  # ~15k inner
for B in [1, 64, 256, 512, 1024, 2048, 4096, 8192]:
    Yi = torch.from_numpy(np.tile(d["Y_init"][i:i+1], (B, 1, 1))).cuda()
    tg = torch.from_numpy(np.tile(T.targets_from_D(d["D_goal"][i:i+1]), (B, 1))).cuda()
    r = T.solve(Yi, tg); torch.cuda.synchronize()
    t0 = time.time(); r = T.solve(Yi, tg); torch.cuda.synchronize(); dt = time.time() - t0
    inner = int(r["inner_total"][0])
    print("B=%5d time %.4fs inner %d -> %.3f us/iter/wave ; agg %.1f MHv/s" % (B, dt, inner, dt / inner * 1e6, B * inner / dt / 1e6), flush=True)
