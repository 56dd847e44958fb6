"""dev: random 3-D graphs on the workgroup-per-problem kernels -- rigid cliques of random size and
position in the node numbering, hinges on clique pairs, other nodes tied to random subsets --
cost / gradient / Hessian product against the CPU oracle (closed form, threshold lowered to 4
nodes, and the direct sum where it fits).  Usage: dev_clique_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import c_oracle as co
from graphik_amd.engine import Template
from graphik_amd._ffi import GikError

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(cases):
    N = int(rng.randint(6, 129))
    n_clique = int(rng.randint(4, N + 1))
    P = rng.randn(N, 3) * rng.uniform(0.3, 2.0, 3) + rng.choice([0.0, 50.0]) * rng.randn(3)   # scenes far from the origin too
    Dtrue = ((P[:, None] - P[None]) ** 2).sum(-1)
    perm = rng.permutation(N)
    clique, other = perm[:n_clique], perm[n_clique:]
    om = np.zeros((N, N)); pL = np.zeros((N, N)); pU = np.zeros((N, N))
    hinge_mod = int(rng.randint(3, 40))
    for a in range(n_clique):
        for b in range(a + 1, n_clique):
            i, j = clique[a], clique[b]
            om[i, j] = om[j, i] = 1.0
            if (a * 7 + b) % hinge_mod == 0:
                pL[i, j] = pL[j, i] = 0.9 * Dtrue[i, j]
                pU[i, j] = pU[j, i] = 1.2 * Dtrue[i, j]
    for q, i in enumerate(other):
        nb = rng.choice(N, size=min(N - 1, int(rng.randint(1, 9))), replace=False)
        for j in nb:
            if j == i:
                continue
            if rng.rand() < 0.6:
                om[i, j] = om[j, i] = 1.0
            else:
                pL[i, j] = pL[j, i] = 0.5 * Dtrue[i, j]
                pU[i, j] = pU[j, i] = 1.5 * Dtrue[i, j]
    il = co.limit_inds(om, pL, pU)
    D = Dtrue * om
    pert = rng.choice([1e-7, 1e-3, 0.3])
    # near a solution every residual c = d - D is a difference of O(1) numbers: cost and gradient
    # carry eps * d / c relative to their own size in ANY summation order (the oracle's included)
    tol = 1e-11 * max(1.0, 1e-3 / pert)
    Y = P + pert * rng.randn(N, 3)
    W = rng.randn(N, 3)
    want = (co.lcost(Y, D, om, pL, pU, il), co.lgrad(Y, D, om, pL, pU, il), co.lhess(Y, W, D, om, pL, pU, il))
    line = f"case {case}: N {N} clique {n_clique} pert {pert:g}"
    for flags in (64, 128):
        try:
            T = Template.from_matrices(om, pL, pU, k=3, use_limits=True,
                                       params={"force_block_path": 1, "debug_flags": flags})
        except GikError as e:
            line += f" | flags {flags}: {e}"
            continue
        tg = T.targets_from_D(D)
        ec = abs(float(T.cost(Y, tg)[0]) - want[0]) / max(abs(want[0]), 1e-300)
        eg = np.abs(T.grad(Y, tg)[0].cpu().numpy() - want[1]).max() / max(np.abs(want[1]).max(), 1e-300)
        eh = np.abs(T.hess(Y, W, tg)[0].cpu().numpy() - want[2]).max() / max(np.abs(want[2]).max(), 1e-300)
        worst = max(worst, ec, eg, eh)
        line += f" | flags {flags}: cost {ec:.1e} grad {eg:.1e} hess {eh:.1e}"
        assert max(ec, eg) < tol and eh < 1e-11, line
    print(line, flush=True)
print("worst relative error", worst)
