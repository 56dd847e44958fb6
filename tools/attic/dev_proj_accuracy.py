"""dev: how horizontal is proj(Z)?  <Y E_m, proj(Z)> / (|Y||proj Z|) for the three skew generators,
GPU projector vs the oracle's (reference algorithm), at points near a solution."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as co
from graphik_amd.engine import Template
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests/golden/lwa4d.npz"))
T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True)
E = [np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 0.]]), np.array([[0, 0, 1], [0, 0, 0], [-1, 0, 0.]]), np.array([[0, 0, 0], [0, 0, 1], [0, -1, 0.]])]
rng = np.random.RandomState(0)
LD = np.longdouble
def vert(Y, P):
    return max(abs(float(((Y.astype(LD) @ e.astype(LD)) * P.astype(LD)).sum())) for e in E) / (np.linalg.norm(Y) * np.linalg.norm(P))
for g in (0, 3, 5, 9):
    Y = d["Y_sol"][g] + 1e-6 * rng.randn(*d["Y_sol"][g].shape)
    Z = rng.randn(*Y.shape)
    Pg = T.proj(Y, Z)[0].cpu().numpy(); Po = co.proj(Y, Z)
    # exact projector in extended precision: least squares on the vertical basis
    V = np.stack([(Y.astype(LD) @ e.astype(LD)).ravel() for e in E], axis=1)
    coef = np.linalg.solve((V.T @ V).astype(np.float64), (V.T @ Z.astype(LD).ravel()).astype(np.float64))
    Px = (Z.astype(LD).ravel() - V @ coef.astype(LD)).reshape(Y.shape)
    print("goal %d: vertical residual GPU %.2e oracle %.2e | |P - P_exact|/|P| GPU %.2e oracle %.2e | idempotence GPU %.2e" % (
        g, vert(Y, Pg), vert(Y, Po), float(np.linalg.norm((Pg - Px).astype(np.float64))) / np.linalg.norm(Pg),
        float(np.linalg.norm((Po - Px).astype(np.float64))) / np.linalg.norm(Po),
        np.linalg.norm(T.proj(Y, Pg)[0].cpu().numpy() - Pg) / np.linalg.norm(Pg)))
