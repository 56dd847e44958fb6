"""dev: textbook tCG (numpy) from mid/late-phase points with oracle operators and with the GPU kernels as
operators, next to the GPU solver's own inner-iteration count from the same point and radius."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as co
from graphik_amd.utils.roboturdf import load_ur10
from graphik_amd.solvers.riemannian_solver import BatchProblem
from graphik_amd.engine import Template
from tools.attic.dev_maxinner_numpy_lib import tcg
robot, graph = load_ur10()
prob = BatchProblem(graph, use_limits=True)
B = 64
rng = np.random.RandomState(3)
Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(B, robot.n))
targets, Y0 = prob.prepare(Tg)
D_all, _, _ = prob.assemble(Tg)
om, pL, pU = prob.omega, prob.psi_L, prob.psi_U
il = co.limit_inds(om, pL, pU)
r = prob.template.solve(Y0, targets, trace_cap=3000); torch.cuda.synchronize()
its = r["iterations"].cpu().numpy(); tr = {k: v.cpu().numpy() for k, v in r["trace"].items()}
tot = np.zeros(4)
for b in range(24):
    cand = [k for k in range(int(0.5 * its[b]), its[b]) if tr["stop"][b, k] == 3 and tr["numit"][b, k] > 40]
    if not cand: continue
    k = cand[len(cand) // 2]
    tk = Template.from_matrices(om, pL, pU, k=3, use_limits=True, params=dict(maxiter=k))
    rk = tk.solve(Y0[b:b + 1], targets[b:b + 1]); torch.cuda.synchronize()
    Y = rk["x"][0].cpu().numpy(); D = D_all[b]; tg = targets[b:b + 1]
    Gg = tk.grad(Y[None], tg)[0].cpu().numpy().reshape(Y.shape)
    G = co.lgrad(Y, D, om, pL, pU, il)
    ho = lambda Y_, W: co.lhess(Y_, W, D, om, pL, pU, il)
    hg = lambda Y_, W: tk.hess(Y_[None], W[None], tg)[0].cpu().numpy().reshape(Y.shape)
    pg = lambda Y_, Z: tk.proj(Y_[None], Z[None])[0].cpu().numpy().reshape(Y.shape)
    Dl = float(tr["Delta"][b, k])
    a = tcg(Y, G, Dl, ho, co.proj); c = tcg(Y, Gg, Dl, hg, pg); d = tcg(Y, Gg, Dl, ho, co.proj)
    print("b %2d it %3d/%3d |g| %.2e Delta %.2e: oracle ops %s | oracle ops, GPU grad %s | GPU ops %s | GPU solver numit %d stop %d" % (
        b, k, its[b], np.linalg.norm(G), Dl, a, d, c, tr["numit"][b, k], tr["stop"][b, k]), flush=True)
    tot += [a[0], d[0], c[0], tr["numit"][b, k]]
print("totals: oracle ops %d | oracle ops GPU grad %d | GPU ops %d | GPU solver %d" % tuple(tot))
