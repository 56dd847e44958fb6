"""dev: which part of (GPU gradient - oracle gradient) delays the textbook tCG run with the oracle's operators?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as co
from graphik_amd.utils.roboturdf import load_ur10
from graphik_amd.solvers.riemannian_solver import BatchProblem
from graphik_amd.engine import Template
from tools.attic.dev_maxinner_numpy_lib import tcg
from tools.attic.dev_grad_accuracy_lib import lgrad_ld
robot, graph = load_ur10()
prob = BatchProblem(graph, use_limits=True)
B = 64
rng = np.random.RandomState(3)
Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(B, robot.n))
targets, Y0 = prob.prepare(Tg)
D_all, _, _ = prob.assemble(Tg)
om, pL, pU = prob.omega, prob.psi_L, prob.psi_U
il = co.limit_inds(om, pL, pU)
E = [np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 0.]]), np.array([[0, 0, 1], [0, 0, 0], [-1, 0, 0.]]), np.array([[0, 0, 0], [0, 0, 1], [0, -1, 0.]])]
tot = np.zeros(8)
r = prob.template.solve(Y0, targets, trace_cap=3000); torch.cuda.synchronize()
tr = {k: v.cpu().numpy() for k, v in r["trace"].items()}
for b, k in [(0, 53), (3, 57), (8, 69), (11, 301), (21, 130), (17, 58), (4, 50), (9, 134), (14, 131), (6, 255)]:
    tk = Template.from_matrices(om, pL, pU, k=3, use_limits=True, params=dict(maxiter=k))
    rk = tk.solve(Y0[b:b + 1], targets[b:b + 1]); torch.cuda.synchronize()
    Y = rk["x"][0].cpu().numpy(); D = D_all[b]; tg = targets[b:b + 1]
    Gg = tk.grad(Y[None], tg)[0].cpu().numpy().reshape(Y.shape)
    G = co.lgrad(Y, D, om, pL, pU, il)
    Gx = lgrad_ld(Y, D, il, om, pL, pU).astype(np.float64)
    e = Gg - G
    N = Y.shape[0]
    basis = [np.tile(np.eye(3)[c], (N, 1)) / np.sqrt(N) for c in range(3)]      # translations
    e_t = sum((e * t).sum() * t for t in basis)
    V = np.stack([(Y @ m).ravel() for m in E], axis=1); Qv, _ = np.linalg.qr(V)   # vertical space
    e_v = (Qv @ (Qv.T @ e.ravel())).reshape(Y.shape)
    e_r = e - e_t - e_v
    noise = np.random.RandomState(b).randn(*Y.shape); noise *= np.linalg.norm(e) / np.linalg.norm(noise)
    ho = lambda Y_, W: co.lhess(Y_, W, D, om, pL, pU, il)
    Dl = float(tr["Delta"][b, k])
    res = [tcg(Y, g_, Dl, ho, co.proj)[0] for g_ in (G, Gg, G + e_t, G + e_v, G + e_r, G + noise, Gx, co.proj(Y, G))]
    print("b %2d it %3d |e| %.1e (transl %.1e, vertical %.1e, rest %.1e) | G oracle %d | G gpu %d | +transl %d | +vertical %d | +rest %d | +random noise %d | exact grad %d | proj(G oracle) %d" % (
        b, k, np.linalg.norm(e), np.linalg.norm(e_t), np.linalg.norm(e_v), np.linalg.norm(e_r), *res), flush=True)
    tot += res
print("totals", tot)
