"""dev: accuracy of the Hessian-vector product near a solution, GPU vs oracle, against an
extended-precision evaluation (np.longdouble) of lhess (costs.py:175-207)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as co
from graphik_amd.engine import Template
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests/golden/lwa4d.npz"))
om, pL, pU = d["omega"], d["psi_L"], d["psi_U"]
T = Template.from_matrices(om, pL, pU, k=3, use_limits=True)
il = co.limit_inds(om, pL, pU)
LD = np.longdouble
def lhess_ld(Y, W, D):
    Y = Y.astype(LD); W = W.astype(LD); D = D.astype(LD); H = np.zeros_like(Y)
    for i, j in zip(*il):
        y = Y[i] - Y[j]; w = W[i] - W[j]; nrm = (y * y).sum(); sc = (y * w).sum()
        a = LD(0); c = LD(0)
        if om[i, j] > 0: a += 1; c += nrm - D[i, j]
        if pL[i, j] > 0 and pL[i, j] - nrm > 0: a += 1; c += nrm - LD(pL[i, j])
        if pU[i, j] > 0 and nrm - pU[i, j] > 0: a += 1; c += nrm - LD(pU[i, j])
        t = 2 * (2 * sc * a * y + c * w)
        H[i] += t; H[j] -= t
    return H
rng = np.random.RandomState(0)
for g in (0, 3, 5):
    D = d["D_goal"][g]; tg = T.targets_from_D(D)
    for scale in (1e-3, 1e-6, 1e-9):
        Y = d["Y_sol"][g] + scale * rng.randn(*d["Y_sol"][g].shape)
        # a direction from the solver's world: the projected gradient and a random one
        G = co.lgrad(Y, D, om, pL, pU, il)
        for nm, W in (("grad", G / np.linalg.norm(G)), ("random", rng.randn(*Y.shape))):
            Hx = lhess_ld(Y, W, D)
            Hg = T.hess(Y, W, tg)[0].cpu().numpy(); Ho = co.lhess(Y, W, D, om, pL, pU, il)
            nx = float(np.sqrt((Hx * Hx).sum()))
            eg = float(np.sqrt(((Hg.astype(LD) - Hx) ** 2).sum())) / nx; eo = float(np.sqrt(((Ho.astype(LD) - Hx) ** 2).sum())) / nx
            print("goal %d |Y-Ysol| %.0e W=%-6s |H W| %.2e  rel err GPU %.2e  oracle %.2e" % (g, scale, nm, nx, eg, eo))
