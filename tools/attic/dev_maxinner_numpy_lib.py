"""dev: textbook tCG loop shared by the dev_*_numpy probes (trust_region.py:436-599)."""
import numpy as np

def tcg(Y, g, Delta, hess, proj, maxinner=10000, kappa=0.1, fused_beta=False):
    eta = np.zeros_like(Y); Heta = np.zeros_like(Y); r = g.copy()
    r_r = float((r * r).sum()); norm_r0 = np.sqrt(r_r); e_Pe = 0.0; e_Pd = 0.0; d_Pd = r_r; z_r = r_r
    delta = -r; model = 0.0; target = norm_r0 * min(norm_r0, kappa)
    for j in range(maxinner):
        Hd = proj(Y, hess(Y, delta)); d_Hd = float((delta * Hd).sum()); alpha = z_r / d_Hd
        e_Pe_new = e_Pe + 2 * alpha * e_Pd + alpha * alpha * d_Pd
        if d_Hd <= 0 or e_Pe_new >= Delta ** 2: return j, "TR"
        e_Pe = e_Pe_new; ne = eta + alpha * delta; nH = Heta + alpha * Hd
        nm = float((ne * g).sum()) + 0.5 * float((ne * nH).sum())
        if nm >= model: return j, "model"
        eta, Heta, model = ne, nH, nm
        if fused_beta:
            rH = float((r * Hd).sum()); HH = float((Hd * Hd).sum())
            beta = 1.0 + (2 * rH + alpha * HH) / d_Hd
        r = r + alpha * Hd; r_r = float((r * r).sum())
        if j >= 1 and np.sqrt(r_r) <= target: return j, "target"
        if not fused_beta or beta < 1e-3:
            beta = r_r / z_r
        z_r = r_r; delta = -r + beta * delta
        e_Pd = beta * (e_Pd + alpha * d_Pd); d_Pd = z_r + beta * beta * d_Pd
    return maxinner, "max"

