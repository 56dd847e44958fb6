"""dev: accuracy of P(H delta) on the directions a late-phase tCG solve actually produces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as co
from graphik_amd.engine import Template
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests/golden/lwa4d.npz"))
om, pL, pU = d["omega"], d["psi_L"], d["psi_U"]
il = co.limit_inds(om, pL, pU)
T1 = Template.from_matrices(om, pL, pU, k=3, use_limits=True, params=dict(maxiter=1))
LD = np.longdouble
E = [np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 0.]]), np.array([[0, 0, 1], [0, 0, 0], [-1, 0, 0.]]), np.array([[0, 0, 0], [0, 0, 1], [0, -1, 0.]])]
def lhess_ld(Y, W, D):
    Y = Y.astype(LD); W = W.astype(LD); D = D.astype(LD); H = np.zeros_like(Y)
    for i, j in zip(*il):
        y = Y[i] - Y[j]; w = W[i] - W[j]; nrm = (y * y).sum(); sc = (y * w).sum()
        a = LD(0); c = LD(0)
        if om[i, j] > 0: a += 1; c += nrm - D[i, j]
        if pL[i, j] > 0 and pL[i, j] - nrm > 0: a += 1; c += nrm - LD(pL[i, j])
        if pU[i, j] > 0 and nrm - pU[i, j] > 0: a += 1; c += nrm - LD(pU[i, j])
        t = 2 * (2 * sc * a * y + c * w); H[i] += t; H[j] -= t
    return H
def proj_ld(Y, Z):
    V = np.stack([(Y.astype(LD) @ e.astype(LD)).ravel() for e in E], axis=1)
    M = (V.T @ V); rhs = V.T @ Z.astype(LD).ravel()
    coef = np.linalg.solve(M.astype(np.float64), rhs.astype(np.float64)).astype(LD)
    coef = coef + np.linalg.solve(M.astype(np.float64), (rhs - M @ coef).astype(np.float64)).astype(LD)   # one refinement
    return (Z.astype(LD).ravel() - V @ coef).reshape(Y.shape)
g = 3; D = d["D_goal"][g]; tg = T1.targets_from_D(D)
o = co.rtr_solve(d["Y_init"][g], D, om, pL, pU, True, maxiter=444); Y = o["x"]
G = co.lgrad(Y, D, om, pL, pU, il)
r = G.copy(); delta = -r; z_r = float((r * r).sum())
print("tCG from goal 3 outer 444 (f %.1e |g| %.1e); per iteration: |r|, rel err of P(H delta) [GPU, oracle], rel err of <delta,Hdelta> [GPU, oracle]" % (o["f(x)"], np.linalg.norm(G)))
for j in range(100):
    Hx = proj_ld(Y, lhess_ld(Y, delta, D).astype(LD))
    Ho = co.proj(Y, co.lhess(Y, delta, D, om, pL, pU, il))
    Hg = T1.proj(Y, T1.hess(Y, delta, tg)[0].cpu().numpy())[0].cpu().numpy()
    nx = float(np.sqrt((Hx * Hx).sum()))
    dx = float((delta.astype(LD) * Hx).sum())
    if j % 8 == 0 or j > 92:
        print("  j %3d |r| %.2e |Hd|/|d| %.2e  vec err GPU %.1e oracle %.1e | curvature err GPU %.1e oracle %.1e" % (
            j, np.sqrt(z_r), nx / np.linalg.norm(delta),
            float(np.sqrt(((Hg.astype(LD) - Hx) ** 2).sum())) / nx, float(np.sqrt(((Ho.astype(LD) - Hx) ** 2).sum())) / nx,
            abs(float((delta * Hg).sum()) - dx) / dx, abs(float((delta * Ho).sum()) - dx) / dx))
    Hd = Ho; d_Hd = float((delta * Hd).sum()); alpha = z_r / d_Hd
    r = r + alpha * Hd; r_r = float((r * r).sum()); beta = r_r / z_r; z_r = r_r; delta = -r + beta * delta
