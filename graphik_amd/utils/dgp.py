"""Distance-geometry helpers on the host (graphik/utils/dgp.py), numpy only.

Single-problem functions keep the reference's names and semantics; `*_batch` variants are the
vectorised forms used to prepare / post-process whole batches when the corresponding device
kernels are not used.
"""
import math

import numpy as np

from ..graphs.graph_base import DistanceGraph
from ..utils.constants import POS


def gram_from_distance_matrix(D):
    """-1/2 J D J  (dgp.py:28-31)"""
    n = D.shape[-1]
    J = np.identity(n) - (1 / n) * np.ones((n, n))
    return -0.5 * J @ D @ J


def distance_matrix_from_gram(X):
    """dgp.py:34-35"""
    return (X.diagonal()[:, np.newaxis] + X.diagonal()) - 2 * X


def distance_matrix_from_pos(Y):
    """dgp.py:38-39"""
    return distance_matrix_from_gram(Y @ Y.T)


def distance_matrix_from_graph(G, label=None, nonedge=0):
    """Squared DIST per edge; edges without DIST read as weight 1 (networkx default), non-edges
    as `nonedge` (dgp.py:42-50)."""
    W = np.where(G.edge, np.where(np.isnan(G.dist), 1.0, G.dist), float(nonedge))
    return W ** 2


def adjacency_matrix_from_graph(G, label=None, nodelist=None):
    """1 where the edge carries DIST (dgp.py:53-65)."""
    return (G.edge & ~np.isnan(G.dist)).astype(float)


def pos_from_graph(G, node_ids=None):
    """dgp.py:68-82"""
    ids = node_ids or G.node_ids
    return np.array([list(G.nodes[n][POS]) for n in ids])


def graph_from_pos(P, node_ids=None, dist=True):
    """dgp.py:85-103"""
    P = np.asarray(P, dtype=float)
    ids = node_ids or ["p" + str(i) for i in range(P.shape[0])]
    G = DistanceGraph(ids, P.shape[1])
    for i, n in enumerate(ids):
        G.nodes[n][POS] = P[i, :]
    if dist:
        G.complete_edges()
    return G


def graph_from_pos_dict(P, dist=True):
    ids = list(P.keys())
    return graph_from_pos(np.array([P[k] for k in ids]), ids, dist)


def graph_complete_edges(G, overwrite=False):
    """dgp.py:124-147: add the distance between every two nodes with known positions (in place)."""
    G.complete_edges(overwrite=overwrite)
    return G


def normalize_positions(Y, scale=False):
    """dgp.py:233-242: centre the points and rotate them into the eigenbasis of their scatter matrix
    (numpy's eig, in its order and with its signs, as the reference's tests rely on); `scale`
    multiplies by the largest coordinate magnitude, as the reference does."""
    Y = np.asarray(Y, dtype=float)
    Yc = Y - Y.mean(0)
    _, v = np.linalg.eig(Yc.T.dot(Yc))
    Ycr = Yc.dot(v)
    return Ycr * np.abs(Ycr).max() if scale else Ycr


def factor(A):
    """eigh, clip negatives, scale columns by sqrt(eigenvalue), flip (dgp.py:150-159)."""
    n = A.shape[0]
    evals, evecs = np.linalg.eigh(A)
    evals[evals < 0] = 0
    sq = np.eye(n)
    for i in range(n):
        sq[i, i] = math.sqrt(evals[i])
    return np.fliplr(evecs.dot(sq))


def MDS(B, eps=1e-5):
    """Classical MDS; the column count K is the number of eigenvalues > eps of eigh applied to
    the (non-symmetric) factor itself, i.e. of its lower triangle (dgp.py:163-171)."""
    n = B.shape[0]
    x = factor(B)
    evals, _ = np.linalg.eigh(x)
    K = len(evals[evals > eps])
    return x[:, 0:K] if K < n else x


def linear_projection(P, F, dim):
    """Project onto the top-`dim` eigenvectors of sum over nonzeros of F of the difference outer
    products (dgp.py:174-183)."""
    I, J = np.nonzero(F)
    d = P[I, :] - P[J, :]
    S = d.T @ d
    _, eigvec = np.linalg.eigh(S)
    return P @ np.fliplr(eigvec)[:, :dim]


def floyd_warshall_bounds(lower, upper):
    """bound_smoothing on dense LOWER/UPPER matrices (NaN = no edge), batched over leading axes.

    The reference (dgp.py:192-231) runs all-pairs Bellman-Ford on the doubled graph
    {u, u'}: u->u' 0; u->v', v->u' -LOWER; u<->v UPPER; u'<->v' UPPER.  Shortest paths between
    unprimed nodes only use UPPER arcs, and a shortest u -> v' path is  u ~> a -> b' ~> v' with
    exactly one crossing arc, so
        ub = APSP(UPPER),    lb[u,v] = max(0, max_{a,b} (LOWER_ab - ub[u,a] - ub[b,v]))
    with LOWER_aa = 0.  Same values as the doubled-graph sweep up to fp64 association order.
    """
    lower = np.asarray(lower, dtype=float)
    upper = np.asarray(upper, dtype=float)
    N = upper.shape[-1]
    ub = np.where(np.isnan(upper), np.inf, upper)
    eye = np.arange(N)
    ub[..., eye, eye] = 0.0
    for m in range(N):
        ub = np.minimum(ub, ub[..., :, m:m + 1] + ub[..., m:m + 1, :])
    lo = np.where(np.isnan(lower), -np.inf, lower)
    lo[..., eye, eye] = 0.0
    # t[u,b] = max_a (lo[a,b] - ub[u,a]) ; lb[u,v] = max_b (t[u,b] - ub[b,v])
    t = np.max(lo[..., None, :, :] - ub[..., :, :, None], axis=-2)
    lb = np.max(t[..., :, :, None] - ub[..., None, :, :], axis=-2)
    return np.maximum(lb, 0.0), ub


def bound_smoothing(G):
    """Triangle-inequality bound smoothing of a goal graph -> (lb, ub)  (dgp.py:192-231)."""
    return floyd_warshall_bounds(np.where(G.edge, G.lower, np.nan),
                                 np.where(G.edge, G.upper, np.nan))


def generate_initialization(bounds, dim, omega):
    """RiemannianSolver.generate_initialization (riemannian_solver.py:67-75)."""
    lb, ub = bounds
    D_rand = (lb + 0.9 * (ub - lb)) ** 2
    X_rand = MDS(gram_from_distance_matrix(D_rand), eps=1e-8)
    return linear_projection(X_rand, omega, dim)


def _canonical_signs(V):
    """Flip each eigenvector (column) so that its entry of largest magnitude is positive."""
    idx = np.argmax(np.abs(V), axis=-2)
    sgn = np.sign(np.take_along_axis(V, idx[..., None, :], axis=-2))
    return V * np.where(sgn == 0, 1.0, sgn)


def generate_initialization_batch(lb, ub, dim, omega, canonical=False, return_info=False):
    """Vectorised generate_initialization: lb, ub [B,N,N] -> Y_init [B,N,dim].

    canonical=False keeps LAPACK's eigenvector signs (what the reference gets).  The column count
    K of MDS() is the number of positive eigenvalues of a matrix built from the LOWER TRIANGLE of
    the eigenvector factor, so it depends on those arbitrary signs; canonical=True fixes them by
    a rule (largest-magnitude entry positive), which is what the device kernel implements.
    return_info=True also returns {"K", "ev_gram" (ascending), "ev_rank" (ascending)}."""
    B, N, _ = lb.shape
    D = (lb + 0.9 * (ub - lb)) ** 2
    G = gram_from_distance_matrix(D)
    ev, V = np.linalg.eigh(G)
    if canonical:
        V = _canonical_signs(V)
    ev = np.where(ev < 0, 0.0, ev)
    X = (V * np.sqrt(ev)[:, None, :])[:, :, ::-1]
    ev2 = np.linalg.eigvalsh(X)  # lower triangle, like numpy's default UPLO='L'
    K = np.sum(ev2 > 1e-8, axis=1)
    X = X * (np.arange(N)[None, None, :] < K[:, None, None])
    I, J = np.nonzero(omega)
    # S_b = sum_e d_e d_e^T (linear_projection, dgp.py:174-183) as one GEMM per goal, in slabs that
    # keep the [slab, E, N] edge differences under ~256 MB (N = 116 has 11208 ordered pairs)
    S = np.empty((B, N, N))
    slab = max(1, min(B, int(1.6e7 // max(1, len(I) * N))))
    di = np.empty((slab, len(I), N))       # reused across slabs (first-touch page faults of fresh
    dj = np.empty((slab, len(I), N))       # multi-GB temporaries dominated this function at N = 116)
    for s0 in range(0, B, slab):
        n = min(slab, B - s0)
        np.take(X[s0:s0 + n], I, axis=1, out=di[:n])
        np.take(X[s0:s0 + n], J, axis=1, out=dj[:n])
        np.subtract(di[:n], dj[:n], out=di[:n])
        np.matmul(di[:n].transpose(0, 2, 1), di[:n], out=S[s0:s0 + n])
    _, W = np.linalg.eigh(S)
    if canonical:
        W = _canonical_signs(W)
    Y = X @ W[:, :, ::-1][:, :, :dim]
    if return_info:
        return Y, {"K": K, "ev_gram": ev, "ev_rank": ev2}
    return Y
