"""SURVEY 8(f)3 pinned to reference code.  tests/golden/ur10_table_intended.npz is produced by
tools/capture_golden_intended.py: the reference's own add_spherical_obstacle / from_pose /
distance_bound_matrices / costs.py loops, run with the list-vs-string comparison of
graph_base.py:207 made to succeed.  Here (no GPU): the term set AnchoredProblem derives equals the
reference's edge set entry for entry, and the fixed-anchor cost / gradient / Hessian (plain numpy
evaluation over that term set, and the C oracle's anchored driver) reproduce the reference loops'
values on the free rows."""
import numpy as np
import pytest

from conftest import load_golden, make_graph
from parity_util import anchored_numpy, anchored_terms


@pytest.fixture(scope="module")
def setup():
    from graphik_amd.solvers.riemannian_solver import AnchoredProblem
    d = load_golden("ur10_table_intended")
    robot, graph = make_graph("ur10_table")
    ap = AnchoredProblem(graph, host_only=True)
    assert list(d["node_ids"]) == list(graph.node_ids)
    return d, robot, graph, ap


def test_hinge_set_equals_the_reference_edges(setup):
    d, robot, graph, ap = setup
    ids = list(d["node_ids"])
    obs = d["obstacle_index"]
    psi_L, psi_U = d["psi_L"], d["psi_U"]
    assert int(d["n_hinge_edges"]) == 700                       # p0..p6 x 100 (graph_base.py:205-211)
    # obstacle table: same spheres, same order
    assert [ids[i] for i in obs] == ap.obstacle_names
    assert np.array_equal(d["obstacle_pos"], ap.obstacles[:, :3])
    assert np.array_equal(d["obstacle_radius"], ap.obstacles[:, 3])
    # free node x obstacle: lower hinge psi_L = radius^2 exactly on the masked (p-) nodes, nothing else
    free = np.array(ap.free)
    want = ap.obs_mask[:, None] * (ap.obstacles[:, 3] ** 2)[None, :]
    assert np.array_equal(psi_L[np.ix_(free, obs)], want)
    assert np.array_equal(psi_L[np.ix_(obs, free)], want.T)
    assert not psi_U[np.ix_(free, obs)].any() and not d["omega"][:, free][:, :, obs].any()
    # the constants of the formulation (p0, p6) carry the same hinge in the reference; nothing else does
    hinge_rows = {ids[i] for i in np.nonzero(psi_L[:, obs].any(axis=1))[0]}
    assert hinge_rows == {f"p{i}" for i in range(robot.n + 1)}
    # free-free terms and pinned (free x base / goal anchor) terms: entry for entry
    from graphik_amd.engine import build_terms
    sub = np.ix_(free, free)
    for g in range(len(d["T_goal"])):
        om, D = d["omega"][g], d["D_goal"][g]
        ti, tj, tk, tv = build_terms(om[sub], psi_L[sub], psi_U[sub], True)
        assert np.array_equal(ti, ap.free_terms[0]) and np.array_equal(tj, ap.free_terms[1])
        assert np.array_equal(tk, ap.free_terms[2])
        tgt = np.where(np.isnan(tv), D[sub][ti, tj], tv)
        assert np.allclose(tgt, ap.free_terms[3], rtol=1e-13, atol=0)
        pins = set()
        for fi, i in enumerate(ap.free):
            for r, a in enumerate(ap.anchors):
                if om[i, a] != 0:
                    pins.add((fi, r, 1, D[i, a]))
                if psi_L[i, a] != 0:
                    pins.add((fi, r, 2, psi_L[i, a]))
                if psi_U[i, a] != 0:
                    pins.add((fi, r, 3, psi_U[i, a]))
        mine = {(p[0], p[1], p[2]): p[3] for p in ap.pin}
        assert {(p[0], p[1], p[2]) for p in pins} == set(mine)
        for p in pins:
            assert abs(mine[p[:3]] - p[3]) <= 1e-13 * abs(p[3])


def test_anchored_cost_grad_hess_equal_the_reference_loops(setup):
    """Free rows of lcost / lgrad / lhess (costs.py:80-207) on the N = 116 graph with every anchor at
    its true position and W = 0 there == the fixed-anchor evaluation over AnchoredProblem's terms."""
    d, robot, graph, ap = setup
    free = np.array(ap.free)
    Nf = len(free)
    ga_all = ap.goal_anchors(d["T_goal"])
    for t in range(len(d["kat_cost"])):
        g = int(d["kat_goal"][t])
        Y, W = d["kat_Y"][t], d["kat_W"][t]
        # the fixture's anchor rows ARE the positions AnchoredProblem uses
        assert np.allclose(Y[ap.anchors[:-2]], ap.base.anchor_pos, atol=1e-15)
        assert np.allclose(Y[ap.anchors[-2:]].ravel(), ga_all[g], atol=1e-14)
        f, G, H = anchored_numpy(ap, Nf, Y[free], W[free], ga_all[g])
        assert int(d["kat_active_hinges"][t]) >= 5
        assert abs(f - d["kat_cost"][t]) <= 1e-12 * abs(d["kat_cost"][t])
        Gr, Hr = d["kat_grad"][t][free], d["kat_hess"][t][free]
        assert np.abs(G - Gr).max() <= 1e-12 * np.abs(Gr).max()
        assert np.abs(H - Hr).max() <= 1e-12 * np.abs(Hr).max()


def test_oracle_anchored_driver_on_the_reference_points(setup):
    """The C twin of the anchored kernels (gik_o_rtr_solve_anchored) evaluated at the fixture's
    points for ONE iteration whose step is rejected (rho_prime = 1e300; like the reference, the driver
    tests its stopping rules after an iteration): the f and |grad| it reports are those of the start
    point, i.e. the reference loops' cost and free-row gradient norm."""
    from oracle import c_oracle as co
    d, robot, graph, ap = setup
    free = np.array(ap.free)
    Nf = len(free)
    ti, tj, tk, target = ap.free_terms
    om = np.zeros((Nf, Nf)); pL = np.zeros((Nf, Nf)); pU = np.zeros((Nf, Nf)); D = np.zeros((Nf, Nf))
    for i, j, k_, t in zip(ti, tj, tk, target):
        if k_ == 1:
            om[i, j] = om[j, i] = 1.0; D[i, j] = D[j, i] = t
        elif k_ == 2:
            pL[i, j] = pL[j, i] = t
        else:
            pU[i, j] = pU[j, i] = t
    ga_all = ap.goal_anchors(d["T_goal"])
    for t in range(len(d["kat_cost"])):
        g = int(d["kat_goal"][t])
        node, pos, tgt, kind = anchored_terms(ap, ga_all[g])
        o = co.rtr_solve_anchored(d["kat_Y"][t][free], D, om, pL, pU, node, pos, tgt, kind, maxiter=1, rho_prime=1e300)
        assert abs(o["f(x)"] - d["kat_cost"][t]) <= 1e-12 * d["kat_cost"][t]
        gn = np.linalg.norm(d["kat_grad"][t][free])
        assert abs(o["gradnorm"] - gn) <= 1e-12 * gn
