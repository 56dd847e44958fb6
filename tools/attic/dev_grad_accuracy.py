"""dev: accuracy of the GPU gradient vs the oracle's, against extended precision, at mid-phase points."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as co
from graphik_amd.utils.roboturdf import load_ur10
from graphik_amd.solvers.riemannian_solver import BatchProblem
from graphik_amd.engine import Template
from tools.attic.dev_maxinner_numpy_lib import tcg
LD = np.longdouble
robot, graph = load_ur10()
prob = BatchProblem(graph, use_limits=True)
B = 64
rng = np.random.RandomState(3)
Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(B, robot.n))
targets, Y0 = prob.prepare(Tg)
D_all, _, _ = prob.assemble(Tg)
om, pL, pU = prob.omega, prob.psi_L, prob.psi_U
il = co.limit_inds(om, pL, pU)
def lgrad_ld(Y, D):
    Y = Y.astype(LD); G = np.zeros_like(Y)
    for i, j in zip(*il):
        y = Y[i] - Y[j]; nrm = (y * y).sum(); c = LD(0)
        if om[i, j] > 0: c += nrm - LD(D[i, j])
        if pL[i, j] > 0 and pL[i, j] - nrm > 0: c += nrm - LD(pL[i, j])
        if pU[i, j] > 0 and nrm - pU[i, j] > 0: c += nrm - LD(pU[i, j])
        G[i] += 2 * c * y; G[j] -= 2 * c * y
    return G
for b, k in [(0, 53), (3, 57), (8, 69), (11, 301), (21, 130), (17, 58)]:
    tk = Template.from_matrices(om, pL, pU, k=3, use_limits=True, params=dict(maxiter=k))
    rk = tk.solve(Y0[b:b + 1], targets[b:b + 1]); torch.cuda.synchronize()
    Y = rk["x"][0].cpu().numpy(); D = D_all[b]; tg = targets[b:b + 1]
    Gg = tk.grad(Y[None], tg)[0].cpu().numpy().reshape(Y.shape)
    G = co.lgrad(Y, D, om, pL, pU, il)
    Gx = lgrad_ld(Y, D)
    # targets as the GPU sees them vs D
    tgD = tk.targets_from_D(D) if hasattr(tk, "targets_from_D") else None
    dt = np.abs(np.asarray(tgD) - np.asarray(tg[0])).max() if tgD is not None else -1
    print("b %d it %d |g| %.3e: |G_oracle - exact| %.2e  |G_gpu - exact| %.2e  |G_gpu - G_oracle| %.2e   max |targets(prepare) - targets(D)| %.2e" % (
        b, k, np.linalg.norm(G), float(np.sqrt(((G - Gx) ** 2).sum())), float(np.sqrt(((Gg - Gx) ** 2).sum())), np.linalg.norm(G - Gg), dt))
    Gt = tk.grad(Y[None], np.asarray(tgD)[None])[0].cpu().numpy().reshape(Y.shape) if tgD is not None else Gg
    print("      GPU grad with targets from D: |. - exact| %.2e" % float(np.sqrt(((Gt - Gx) ** 2).sum())))
