"""dev: where do the extra tCG iterations of the GPU come from?  The textbook loop
(trust_region.py:436-599) is run in numpy twice from the same late-phase points -- once with the
oracle's operators (lhess, proj), once with the GPU kernels (gik_hess, gik_proj) -- and compared
with the inner-iteration count of the GPU solver's own first outer iteration from that point."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as co
from graphik_amd.engine import Template
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests/golden/lwa4d.npz"))
om, pL, pU = d["omega"], d["psi_L"], d["psi_U"]
il = co.limit_inds(om, pL, pU)
T1 = Template.from_matrices(om, pL, pU, k=3, use_limits=True, params=dict(maxiter=1))

def tcg(Y, g, Delta, hess, proj, maxinner=10000, kappa=0.1):
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
        r = r + alpha * Hd; r_r = float((r * r).sum())
        if j >= 1 and np.sqrt(r_r) <= target: return j, "target"
        beta = r_r / z_r; z_r = r_r; delta = -r + beta * delta
        e_Pd = beta * (e_Pd + alpha * d_Pd); d_Pd = z_r + beta * beta * d_Pd
    return maxinner, "max"

for g in (0, 3, 5, 12, 13):
    D = d["D_goal"][g]; tg = T1.targets_from_D(D)
    # late-phase point: run the oracle to f ~ 1e-10 by limiting iterations
    for frac in (0.5, 0.8):
        n = int(frac * d["iterations"][g])
        o = co.rtr_solve(d["Y_init"][g], D, om, pL, pU, True, traj_cap=4000, maxiter=n)
        Y = o["x"]; Delta = float(o["traj"]["Delta"][n - 1]) if n > 0 else 1.0
        G = co.lgrad(Y, D, om, pL, pU, il)
        ho = lambda Y_, W: co.lhess(Y_, W, D, om, pL, pU, il)
        hg = lambda Y_, W: T1.hess(Y_, W, tg)[0].cpu().numpy()
        pg = lambda Y_, Z: T1.proj(Y_, Z)[0].cpu().numpy()
        jo = tcg(Y, G, 1e9, ho, co.proj); jg = tcg(Y, G, 1e9, hg, pg)
        jhg = tcg(Y, G, 1e9, hg, co.proj); jpg = tcg(Y, G, 1e9, ho, pg)
        print("   (no TR bound) oracle/oracle %s | GPU hess + oracle proj %s | oracle hess + GPU proj %s | GPU/GPU %s" % (jo, jhg, jpg, jg))
        r = T1.solve(Y[None], tg[None] if tg.ndim == 1 else tg, trace_cap=2)
        print("goal %2d at outer %4d (f %.1e, |g| %.1e, Delta %.1e): numpy tCG with oracle ops %s | with GPU ops %s | GPU solver numit %d stop %d (its own Delta)" % (
            g, n, o["f(x)"], np.linalg.norm(G), Delta, jo, jg, int(r["trace"]["numit"][0][0]), int(r["trace"]["stop"][0][0])))
