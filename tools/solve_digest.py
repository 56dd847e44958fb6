"""Digest of the solve kernel's outputs on a fixed batch (golden start points, jittered) + its time by events:
run once per library (GIK_LIB_PATH) to show that a change of the kernel left every bit where it was.
    python tools/solve_digest.py [golden name] [B] [debug_flags]"""
import hashlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from graphik_amd.engine import Template
name = sys.argv[1] if len(sys.argv) > 1 else "planar10_limits_pi"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
d = np.load(os.path.join(R, "tests", "golden", name + ".npz"))
T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=int(d["dim"]), use_limits=bool(int(d["use_limits"])),
                           params={"debug_flags": flags})
tg = T.targets_from_D(d["D_goal"])
G = len(d["Y_init"])
idx = np.arange(B) % G
Y0 = d["Y_init"][idx] + 1e-3 * np.random.RandomState(0).randn(B, *d["Y_init"].shape[1:])
tgb = tg[torch.as_tensor(idx, device=tg.device)]
ms = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = T.solve(Y0, tgb); e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
h = hashlib.sha256()
for k in ("x", "f", "gradnorm", "iterations", "inner_total", "stop", "n_accept"):
    h.update(r[k].cpu().numpy().tobytes())
print("%s B=%d lib=%s: solve min %.3f ms | iterations %d inner %d | sha %s" % (
    name, B, os.path.basename(os.environ.get("GIK_LIB_PATH", "libgraphik_amd.so")), min(ms),
    int(r["iterations"].sum()), int(r["inner_total"].sum()), h.hexdigest()[:16]))
