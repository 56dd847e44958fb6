"""dev: sweeps of the round-robin cyclic Jacobi (the prepare kernels' schedule and threshold) on planar-10 Gram
matrices, as they are (13 x 13, one exactly-zero eigenvalue: the centring vector) and with that vector deflated by
a Householder similarity (12 x 12).  CPU only (numpy).  GIK_CPU_ONLY-safe: no device handle is created."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from conftest import make_graph
from graphik_amd.utils import dgp


def rr_pairs(n):
    ne = n + (n & 1)
    out = []
    for r in range(ne - 1):
        rnd = []
        for m in range(ne // 2):
            if m == 0: p, q = ne - 1, r
            else: p, q = (r + m) % (ne - 1), (r - m + ne - 1) % (ne - 1)
            p, q = min(p, q), max(p, q)
            if q < n: rnd.append((p, q))
        out.append(rnd)
    return out


def jacobi_sweeps(A, max_sweeps=12):
    A = A.copy(); n = len(A)
    thr = 1e-16 * np.linalg.norm(A)
    rounds = rr_pairs(n)
    nrot = 0
    for sw in range(max_sweeps):
        rotated = False
        for rnd in rounds:
            J = np.eye(n)
            for p, q in rnd:
                apq = A[p, q]
                if abs(apq) > thr:
                    th = (A[q, q] - A[p, p]) / (2 * apq)
                    t = np.sign(th if th != 0 else 1.0) / (abs(th) + np.sqrt(th * th + 1))
                    c = 1 / np.sqrt(t * t + 1); s = t * c
                    J[p, p] = c; J[q, q] = c; J[p, q] = s; J[q, p] = -s
                    rotated = True; nrot += 1
            A = J.T @ A @ J
        if not rotated:
            return sw + 1, nrot, np.sort(np.diag(A))
    return max_sweeps, nrot, np.sort(np.diag(A))


name = sys.argv[1] if len(sys.argv) > 1 else "planar10_limits_pi"
robot, graph = make_graph(name)
from graphik_amd.solvers.riemannian_solver import BatchProblem
prob = BatchProblem(graph, use_limits=True, host_only=True)
rs = np.random.RandomState(0)
lbq, ubq = robot.limits_arrays()
Tg = robot.fk_batch(lbq + (ubq - lbq) * rs.rand(300, robot.n))
D, lo, up = prob.assemble(Tg)
lb, ub = dgp.floyd_warshall_bounds(lo, up)
G = dgp.gram_from_distance_matrix((lb + 0.9 * (ub - lb)) ** 2)
N = G.shape[1]
u = np.ones(N) / np.sqrt(N); u[-1] -= 1.0           # Householder vector: ones / sqrt(N) -> e_N
H = np.eye(N) - 2 * np.outer(u, u) / (u @ u)
full, defl, rot_f, rot_d, err = [], [], [], [], []
for g in range(len(G)):
    sf, rf, ef = jacobi_sweeps(G[g])
    B2 = H @ G[g] @ H
    sd, rd, ed = jacobi_sweeps(B2[:N - 1, :N - 1])
    full.append(sf); defl.append(sd); rot_f.append(rf); rot_d.append(rd)
    ref = np.linalg.eigvalsh(G[g])
    err.append(max(np.abs(ef - ref).max(), np.abs(np.sort(np.append(ed, 0.0)) - ref).max()) / np.abs(ref).max())
    if g == 0: print("spectrum of goal 0:", np.round(ref, 6), " coupling |B2[:-1,-1]| %.1e" % np.abs(B2[:N - 1, N - 1]).max())
full, defl = np.array(full), np.array(defl)
print(f"{name}: sweeps (incl. the confirming one) full 13 x 13: mean {full.mean():.2f} max {full.max()}   deflated 12 x 12: mean {defl.mean():.2f} max {defl.max()}")
print(f"   max over groups of four goals: full {np.max(full[:len(full)//4*4].reshape(-1, 4), 1).mean():.2f}  deflated {np.max(defl[:len(defl)//4*4].reshape(-1, 4), 1).mean():.2f}")
print(f"   rotations: full {np.mean(rot_f):.0f}  deflated {np.mean(rot_d):.0f};  rounds per sweep 13 vs 11, pairs per round 6 vs 6;  eigenvalue error {max(err):.1e}")
