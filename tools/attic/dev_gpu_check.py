"""Developer check (run on the GPU box): HIP kernels vs oracle + quick timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from graphik_amd.engine import Template
from oracle import c_oracle as co

def rel(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))

for nm in ["lwa4d", "planar10_limits_halfpi", "planar10_nolimits", "ur10", "kuka"]:
    d = np.load(f"tests/golden/{nm}.npz")
    use_lim = bool(int(d["use_limits"]))
    om, pL, pU = d["omega"], d["psi_L"], d["psi_U"]
    k = int(d["dim"])
    T = Template.from_matrices(om, pL, pU, k=k, use_limits=use_lim)
    D0 = d["D_goal"][0]
    tg = T.targets_from_D(D0)
    Y, W = d["kat_Y"], d["kat_W"]
    key = "lim" if use_lim else "nolim"
    c = T.cost(Y, tg).cpu().numpy(); g = T.grad(Y, tg).cpu().numpy(); h = T.hess(Y, W, tg).cpu().numpy()
    pj = T.proj(Y, W).cpu().numpy()
    print(nm, "KAT rel: cost %.1e grad %.1e hess %.1e proj %.1e" % (
        rel(c, d[f"kat_{key}_loop_cost"]), rel(g, d[f"kat_{key}_loop_grad"]),
        rel(h, d[f"kat_{key}_loop_hess"]), rel(pj, d["kat_proj"])))
    # trajectories from captured Y_init
    G = len(d["seed"])
    tgs = T.targets_from_D(d["D_goal"])
    t0 = time.time()
    r = T.solve(d["Y_init"], tgs, trace_cap=48)
    torch.cuda.synchronize()
    dt = time.time() - t0
    it = r["iterations"].cpu().numpy(); inner = r["inner_total"].cpu().numpy()
    print("   iters gpu", it.tolist())
    print("   iters ref", d["iterations"].tolist())
    print("   inner gpu", inner.tolist())
    print("   inner ref", d["hv_total"].tolist(), "time %.3fs" % dt)
    print("   f gpu", ["%.1e" % v for v in r["f"].cpu().numpy()])
    # oracle trajectory compare for goal 0
    o = co.rtr_solve(d["Y_init"][0], d["D_goal"][0], om, pL, pU, use_lim, traj_cap=48)
    tr = r["trace"]
    n = min(len(o["traj"]["numit"]), 48)
    gn = tr["numit"][0].cpu().numpy()[:n]
    same = int(np.argmax(np.concatenate([gn != o["traj"]["numit"][:n], [True]])))
    print("   goal0: oracle iters", o["iterations"], "numit prefix equal for", same, "outer its")

# throughput: LWA4D batch 4096 random goals -> reuse golden Y_init/targets tiled (timing only)
d = np.load("tests/golden/lwa4d.npz")
T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True)
B = 4096
reps = B // len(d["seed"])
Yi = torch.from_numpy(np.tile(d["Y_init"], (reps, 1, 1))).cuda()
tg = torch.from_numpy(np.tile(T.targets_from_D(d["D_goal"]), (reps, 1))).cuda()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    r = T.solve(Yi, tg)
    torch.cuda.synchronize(); dt = time.time() - t0
    inner = r["inner_total"].cpu().numpy().astype(np.int64)
    print("B=%d time %.3fs -> %.0f solves/s ; total inner %d ; %.3f us per Hv (aggregate) ; max inner %d -> %.3f us/iter bound" % (
        B, dt, B / dt, inner.sum(), dt / inner.sum() * 1e6, inner.max(), dt / inner.max() * 1e6))
# single-wave latency: B=1 straggler
i = int(np.argmax(d["hv_total"]))
torch.cuda.synchronize(); t0 = time.time()
r = T.solve(d["Y_init"][i:i+1], T.targets_from_D(d["D_goal"][i:i+1]))
torch.cuda.synchronize(); dt = time.time() - t0
print("single problem: inner %d time %.3fs -> %.3f us per tCG iteration" % (int(r["inner_total"][0]), dt, dt / int(r["inner_total"][0]) * 1e6))
