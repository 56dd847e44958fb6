"""Host-side mirror of the reference API (graphs, robots, dgp, loaders) against golden data
captured from the reference, plus the reference's own round-trip properties
(tests/test_joint_variables.py, test_bound_smoothing.py, test_distance_matrix.py).  CPU only."""
import os

import numpy as np
import pytest

from conftest import REPO, SCENARIOS, SCENARIOS_3D, load_golden, make_graph
from graphik_amd.utils import dgp
from graphik_amd.utils.lie import SE2, SE3
from graphik_amd.utils.utils import table_environment


def _same(a, b, tol=0.0):
    return np.array_equal(np.isnan(a), np.isnan(b)) and \
        np.nanmax(np.abs(np.nan_to_num(a) - np.nan_to_num(b))) <= tol


@pytest.mark.parametrize("name", SCENARIOS + ["ur10_table"])
def test_template_matches_reference(name):
    d = load_golden(name)
    robot, graph = make_graph(name)
    assert graph.node_ids == list(d["node_ids"])
    assert _same(graph.dist, d["G_dist"]) and _same(graph.lower, d["G_lower"])
    assert _same(graph.upper, d["G_upper"])
    assert np.array_equal(np.where(graph.bounded == 5, 0, graph.bounded), d["G_bounded"])
    L, U = graph.distance_bound_matrices()
    assert np.array_equal(L, d["psi_L"]) and np.array_equal(U, d["psi_U"])
    assert np.abs(robot.T0_array() - d["T0"]).max() == 0.0


@pytest.mark.parametrize("name", SCENARIOS + ["ur10_table"])
def test_goal_assembly_bounds_init(name):
    d = load_golden(name)
    robot, graph = make_graph(name)
    SE = SE3 if graph.dim == 3 else SE2
    for g in range(min(3, len(d["seed"]))):
        G = graph.from_pose(SE.from_matrix(d["T_goal"][g]))
        assert np.abs(dgp.distance_matrix_from_graph(G) - d["D_goal"][g]).max() < 1e-14 * max(1.0, np.abs(d["D_goal"][g]).max())
        om = dgp.adjacency_matrix_from_graph(G)
        assert np.array_equal(om, d["omega"])
        lb, ub = dgp.bound_smoothing(G)
        assert np.abs(lb - d["lb"][g]).max() < 1e-12 and np.abs(ub - d["ub"][g]).max() < 1e-12
        Y0 = dgp.generate_initialization((lb, ub), graph.dim, om)
        # eigenvector signs are free: compare up to per-column sign
        assert np.abs(np.abs(Y0) - np.abs(d["Y_init"][g])).max() < 1e-9
        assert np.abs(Y0 @ Y0.T - d["Y_init"][g] @ d["Y_init"][g].T).max() < 1e-9


@pytest.mark.parametrize("name", SCENARIOS)
def test_batched_preprocessing_equals_single(name):
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    d = load_golden(name)
    robot, graph = make_graph(name)
    use_lim = bool(int(d["use_limits"]))
    prob = BatchProblem(graph, use_limits=use_lim, host_only=True)   # host part only: no device template
    D, lo, up = prob.assemble(d["T_goal"])
    assert np.array_equal(prob.omega, d["omega"])
    assert np.abs(D - d["D_goal"]).max() < 1e-14 * max(1.0, np.abs(d["D_goal"]).max())
    lb, ub = dgp.floyd_warshall_bounds(lo, up)
    assert np.abs(lb - d["lb"]).max() < 1e-12 and np.abs(ub - d["ub"]).max() < 1e-12
    Y0 = dgp.generate_initialization_batch(lb, ub, graph.dim, prob.omega)
    assert np.abs(np.abs(Y0) - np.abs(d["Y_init"])).max() < 1e-9


@pytest.mark.parametrize("name", SCENARIOS)
def test_joint_variables_and_pose_errors_match_reference(name):
    d = load_golden(name)
    robot, graph = make_graph(name)
    SE = SE3 if graph.dim == 3 else SE2
    for g in range(len(d["seed"])):
        T = SE.from_matrix(d["T_goal"][g])
        q = graph.joint_variables(d["Y_sol"][g], {f"p{robot.n}": T})
        assert np.abs(robot.q_to_array(q) - d["q_sol"][g]).max() < 1e-9
        Ts = robot.pose(q, f"p{robot.n}")
        assert abs(np.linalg.norm(T.trans - Ts.trans) - d["pos_err"][g]) < 1e-9


@pytest.mark.parametrize("name", ["lwa4d", "ur10", "kuka", "planar10"])
def test_fk_realization_roundtrip(name):
    """FK -> realization -> joint_variables recovers q (reference tests/test_joint_variables.py
    :55-78), and FK / realization equal the reference's numbers."""
    h = load_golden("host_kats")
    robot, graph = make_graph("planar10_nolimits" if name == "planar10" else name)
    Q = h[f"{name}_fk_q"]
    for b in range(len(Q)):
        q = robot.array_to_q(Q[b])
        for i in range(1, robot.n + 1):
            assert np.abs(robot.pose(q, f"p{i}").as_matrix() - h[f"{name}_fk_T"][b, i - 1]).max() < 1e-12
        G = graph.realization(q)
        assert np.abs(G.positions() - h[f"{name}_realization"][b]).max() < 1e-12
        qr = graph.joint_variables(G, {f"p{robot.n}": robot.pose(q, f"p{robot.n}")})
        assert np.abs(robot.q_to_array(qr) - h[f"{name}_jointvars"][b]).max() < 1e-9
        assert np.abs(np.mod(robot.q_to_array(qr) - Q[b] + np.pi, 2 * np.pi) - np.pi).max() < 1e-6
    assert np.abs(robot.fk_batch(Q) - h[f"{name}_fk_T"][:, robot.n - 1]).max() < 1e-12


def test_bound_smoothing_brackets_true_distances():
    """lb^2 - tol <= D_true <= ub^2 + tol (reference tests/test_bound_smoothing.py:99-117)."""
    robot, graph = make_graph("ur10")
    rng = np.random.RandomState(3)
    for _ in range(5):
        q = robot.array_to_q(rng.uniform(-np.pi, np.pi, robot.n))
        T = robot.pose(q, f"p{robot.n}")
        lb, ub = dgp.bound_smoothing(graph.from_pose(T))
        D = dgp.distance_matrix_from_pos(graph.realization(q).positions())
        assert np.all(lb ** 2 - 1e-8 <= D) and np.all(D <= ub ** 2 + 1e-8)


def test_table_environment_and_obstacles():
    h = load_golden("host_kats")
    obs = table_environment()
    assert np.array_equal(np.array([o[0] for o in obs]), h["table_centers"])
    assert np.array_equal(np.array([o[1] for o in obs]), h["table_radii"])
    robot, graph = make_graph("ur10_table")
    assert graph.number_of_nodes() == 116
    # reference quirk: no robot<->obstacle lower-bound edges, limit check inert
    L, U = graph.distance_bound_matrices()
    assert not np.any(L[16:, :16]) and graph.check_distance_limits(graph.realization(
        robot.zero_configuration())) == []
    graph.clear_obstacles()
    assert graph.number_of_nodes() == 16


def test_api_surface():
    """Names and call shapes the reference's callers rely on."""
    import graphik_amd.solvers.riemannian_solver as rs
    import inspect
    sig = inspect.signature(rs.solve_with_riemannian)
    assert list(sig.parameters)[:3] == ["graph", "T_goal", "use_jit"] and "jit" in sig.parameters
    sig = inspect.signature(rs.RiemannianSolver.solve)
    assert list(sig.parameters) == ["self", "D_goal", "omega", "use_limits", "bounds", "Y_init",
                                    "jit", "output_log"]
    robot, graph = make_graph("lwa4d")
    for attr in ("from_pose", "node_ids", "robot", "dim", "axis_length", "number_of_nodes",
                 "distance_bound_matrices", "joint_variables", "realization",
                 "check_distance_limits", "add_spherical_obstacle", "clear_obstacles"):
        assert hasattr(graph, attr)
    for attr in ("n", "dim", "random_configuration", "pose", "end_effectors", "lb", "ub"):
        assert hasattr(robot, attr)
    np.random.seed(0)
    q = robot.random_configuration()
    np.random.seed(0)
    u = np.random.rand(robot.n)
    assert np.allclose(robot.q_to_array(q), -np.pi + 2 * np.pi * u)  # robot_base.py:76-85
    with pytest.raises(Exception):
        rs.RiemannianSolver(graph, {"solver": "nope"})
    # static in the reference (riemannian_solver.py:67-78, 121-122): callable on the class, no instance
    for name, params in (("create_cost", ["D_goal", "omega", "jit"]),
                         ("create_cost_limits", ["D_goal", "omega", "psi_L", "psi_U", "jit"]),
                         ("generate_initialization", None)):
        assert isinstance(inspect.getattr_static(rs.RiemannianSolver, name), staticmethod), name
        if params:
            assert list(inspect.signature(getattr(rs.RiemannianSolver, name)).parameters) == params
    om = np.ones((4, 4)) - np.eye(4)
    triple = rs.RiemannianSolver.create_cost(np.zeros((4, 4)), om)      # closures only: nothing touches a GPU yet
    assert len(triple) == 3 and all(callable(f) for f in triple)
    assert len(rs.RiemannianSolver.create_cost_limits(np.zeros((4, 4)), om, 0 * om, 0 * om)) == 3


@pytest.mark.skipif(not os.path.isdir("/root/reference/graphik/robots/urdfs"),
                    reason="reference URDF data only exists in the build container")
@pytest.mark.parametrize("name,urdf", [("lwa4d", "lwa4d.urdf"), ("ur10", "ur10_mod.urdf"),
                                       ("kuka", "kuka_iiwr.urdf")])
def test_urdf_reader_reproduces_packaged_constants(name, urdf):
    from graphik_amd.utils.roboturdf import RobotURDF
    u = RobotURDF("/root/reference/graphik/robots/urdfs/" + urdf)
    n = u.n_q_joints
    r = u.make_Revolute3d(np.pi * np.ones(n), -np.pi * np.ones(n))
    assert np.abs(r.T0_array() - load_golden(name)["T0"]).max() < 1e-12


def test_bench_algorithmic_work_matches_survey():
    """bench.py's roofline figures use SURVEY 8(d)'s per-unit work: F_hv = 4383 flop per
    Hessian-vector product and 1512 B of HBM traffic per solve for LWA4D (N=18, k=3, 75 terms:
    targets + Y_init in, Y_sol + the 48-byte gik_stats record out)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.algorithmic_flops(18, 3, 75, 1, 0, 0) == 4383
    assert bench.algorithmic_flops(13, 2, 19, 1, 0, 0) == 1116      # planar chain, C5
    assert bench.algorithmic_bytes(18, 3, 75) == 8 * (75 + 2 * 54) + 48 == 1512


def test_problem_cache_key_follows_graph_content():
    """solve_batch caches device problems by the CONTENT of the graph (the reference re-reads the
    graph on every call): in-place mutations that keep the node count -- moving an obstacle,
    intended obstacle semantics, other joint limits -- must change the key; an equal graph built
    separately must not."""
    from graphik_amd.solvers.riemannian_solver import graph_fingerprint
    from graphik_amd.utils.roboturdf import load_ur10
    _, g = load_ur10()
    _, g_same = load_ur10()
    base = graph_fingerprint(g)
    assert base == graph_fingerprint(g_same)
    g.add_spherical_obstacle("o0", np.array([0.5, 0.1, 0.3]), 0.1)
    one = graph_fingerprint(g)
    assert one != base
    g.clear_obstacles()
    assert graph_fingerprint(g) == base
    g.add_spherical_obstacle("o0", np.array([0.5, 0.1, 0.4]), 0.1)       # same count, elsewhere
    moved = graph_fingerprint(g)
    assert moved != one
    g.clear_obstacles()
    g.add_spherical_obstacle("o0", np.array([0.5, 0.1, 0.3]), 0.1, intended=True)
    assert graph_fingerprint(g) not in (one, moved)
    _, g_lim = load_ur10(limits=(-np.pi / 2 * np.ones(6), np.pi / 2 * np.ones(6)))
    assert graph_fingerprint(g_lim) != base


# ---- tree-structured robots (several end effectors) ---------------------------------------------
def tree_robot():
    """The robot of the reference's tests/test_joint_variables.py:192-226."""
    from graphik_amd.robots import RobotRevolute
    from graphik_amd.graphs import ProblemGraphRevolute
    pi = np.pi
    params = {"a": {"p1": 0, "p2": -0.612, "p3": -0.612, "p4": -0.5732, "p5": -0.5732},
              "alpha": {"p1": pi / 2, "p2": 0, "p3": 0, "p4": 0, "p5": 0},
              "d": {"p1": 0.1237, "p2": 0, "p3": 0, "p4": 0, "p5": 0},
              "theta": {"p1": 0, "p2": 0, "p3": 0, "p4": 0, "p5": 0}, "modified_dh": False,
              "parents": {"p0": ["p1"], "p1": ["p2", "p3"], "p2": ["p4"], "p3": ["p5"]}, "num_joints": 5}
    robot = RobotRevolute(params)
    return robot, ProblemGraphRevolute(robot)


def test_tree_robot_graph_matches_reference():
    """Topology, node order, zero-configuration frames and every edge attribute of the problem
    graph of a two-end-effector tree against tests/golden/tree5.npz (tools/capture_golden_tree.py:
    the reference's RobotRevolute / ProblemGraphRevolute on the same parameters)."""
    from graphik_amd.graphs.graph_base import B_ABSENT, B_EMPTY, B_NOEDGE
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "tree5.npz"))
    robot, graph = tree_robot()
    assert robot.joint_ids == list(d["joint_ids"]) and robot.end_effectors == list(d["end_effectors"])
    assert graph.node_ids == list(d["node_ids"]) and not robot.is_chain
    T0 = np.stack([robot.nodes[j]["T0"].as_matrix() for j in robot.joint_ids])
    assert np.abs(T0 - d["T0"]).max() < 1e-14
    for key, M in (("G_dist", graph.dist), ("G_lower", graph.lower), ("G_upper", graph.upper)):
        assert np.array_equal(np.isnan(M), np.isnan(d[key])), key
        assert np.nanmax(np.abs(M - d[key])) < 1e-14, key
    code = np.where((graph.bounded == B_ABSENT) | (graph.bounded == B_EMPTY), 0, graph.bounded)
    assert np.array_equal(np.where(graph.bounded == B_NOEDGE, -1, code), d["G_bounded"])
    pL, pU = graph.distance_bound_matrices()
    assert np.abs(pL - d["psi_L"]).max() < 1e-14 and np.abs(pU - d["psi_U"]).max() < 1e-14


def test_tree_robot_realization_and_joint_variables():
    """realization(q) and joint_variables(realization(q), T_goal) against the captured reference
    output, and the reference's own round-trip test (test_joint_variables.py:216-225: 100 random
    configurations, rtol 1e-5) on this implementation."""
    from graphik_amd.utils import dgp
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "tree5.npz"))
    robot, graph = tree_robot()
    for s in range(len(d["q_goal"])):
        np.random.seed(s)
        q = robot.random_configuration()
        assert np.allclose([q[j] for j in robot.joint_ids[1:]], d["q_goal"][s], rtol=0, atol=0)
        T_goal = {ee: robot.pose(q, ee) for ee in robot.end_effectors}
        for i, ee in enumerate(robot.end_effectors):
            assert np.abs(T_goal[ee].as_matrix() - d["T_goal"][s][i]).max() < 1e-13
        G = graph.realization(q)
        assert np.abs(G.positions() - d["X"][s]).max() < 1e-13
        q_rec = graph.joint_variables(G, T_goal)
        assert np.abs(np.array([q_rec[j] for j in robot.joint_ids[1:]]) - d["q_rec"][s]).max() < 1e-10
        Gd = graph.from_pose(T_goal)
        if s < len(d["sol_D_goal"]):
            assert np.abs(dgp.distance_matrix_from_graph(Gd) - d["sol_D_goal"][s]).max() < 1e-13
            assert np.array_equal(dgp.adjacency_matrix_from_graph(Gd), d["omega"])
            lb, ub = dgp.bound_smoothing(Gd)
            assert np.abs(lb - d["sol_lb"][s]).max() < 1e-12 and np.abs(ub - d["sol_ub"][s]).max() < 1e-12
    np.random.seed(100)
    for _ in range(100):
        q_goal = robot.random_configuration()
        T_goal = {ee: robot.pose(q_goal, ee) for ee in robot.end_effectors}
        q_rec = graph.joint_variables(graph.realization(q_goal), T_goal)
        np.testing.assert_allclose([q_goal[k] for k in sorted(q_goal)], [q_rec[k] for k in sorted(q_rec)], rtol=1e-5)


def test_randomized_links_match_reference():
    """load_*(randomized_links=True) (roboturdf.py:236-244: one np.random.rand() per link scales
    the translation between consecutive zero-configuration frames) against frames captured from
    the reference's loaders with the same seed (host_kats.npz)."""
    from graphik_amd.utils.roboturdf import load_kuka, load_schunk_lwa4d, load_ur10
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "host_kats.npz"))
    for nm, ld in (("lwa4d", load_schunk_lwa4d), ("ur10", load_ur10), ("kuka", load_kuka)):
        np.random.seed(5)
        robot, graph = ld(randomized_links=True, randomize_percentage=0.3)
        T = np.stack([robot.nodes[f"p{i}"]["T0"].as_matrix() for i in range(robot.n + 1)])
        assert np.abs(T - d[f"{nm}_randomized_T0"]).max() < 1e-14
        plain, _ = ld()
        assert np.abs(T - plain.T0_array()).max() > 1e-3                 # it does change the arm
        assert graph.number_of_nodes() == 2 * robot.n + 4


# ---- planar trees (graph_planar.py:50-88; ADVICE r2: they used to build a silently wrong model) -----
@pytest.mark.parametrize("which", ["y5", "bin2"])
def test_planar_tree_graph_matches_reference(which):
    """Topology, node order, zero-configuration frames, every edge attribute, psi_L / psi_U, and per
    seed the random configuration, poses, realization, joint_variables, the goal graph's D_goal /
    omega and bound_smoothing of two planar trees against the reference's RobotPlanar /
    ProblemGraphPlanar (tests/golden/planar_tree.npz, tools/capture_golden_planar_tree.py)."""
    from conftest import planar_tree
    from graphik_amd.graphs.graph_base import B_ABSENT, B_EMPTY, B_NOEDGE
    d = load_golden("planar_tree")
    g = lambda k: d[f"{which}_{k}"]   # noqa: E731
    robot, graph = planar_tree(which)
    assert robot.joint_ids == list(g("joint_ids")) and robot.end_effectors == list(g("end_effectors"))
    assert graph.node_ids == list(g("node_ids")) and not robot.is_chain
    T0 = np.stack([robot.nodes[j]["T0"].as_matrix() for j in robot.joint_ids])
    assert np.abs(T0 - g("T0")).max() < 1e-14
    for key, M in (("G_dist", graph.dist), ("G_lower", graph.lower), ("G_upper", graph.upper)):
        assert np.array_equal(np.isnan(M), np.isnan(g(key))), key
        assert np.nanmax(np.abs(M - g(key))) < 1e-14, key
    code = np.where((graph.bounded == B_ABSENT) | (graph.bounded == B_EMPTY), 0, graph.bounded)
    assert np.array_equal(np.where(graph.bounded == B_NOEDGE, -1, code), g("G_bounded"))
    pL, pU = graph.distance_bound_matrices()
    assert np.abs(pL - g("psi_L")).max() < 1e-14 and np.abs(pU - g("psi_U")).max() < 1e-14
    for s in range(len(g("q_goal"))):
        np.random.seed(s)
        q = robot.random_configuration()
        assert np.array_equal([q[j] for j in robot.joint_ids[1:]], g("q_goal")[s])
        T_goal = {ee: robot.pose(q, ee) for ee in robot.end_effectors}
        for i, ee in enumerate(robot.end_effectors):
            assert np.abs(T_goal[ee].as_matrix() - g("T_goal")[s][i]).max() < 1e-13
        G = graph.realization(q)
        assert np.abs(G.positions() - g("X")[s]).max() < 1e-13
        q_rec = graph.joint_variables(G)
        assert list(q_rec) == [v for _, v in graph.structure_edges]
        assert np.abs(np.array([q_rec[j] for j in robot.joint_ids[1:]]) - g("q_rec")[s]).max() < 1e-12
        Gd = graph.from_pose(T_goal)
        assert np.abs(dgp.distance_matrix_from_graph(Gd) - g("D_goal")[s]).max() < 1e-13
        assert np.array_equal(dgp.adjacency_matrix_from_graph(Gd), g("omega"))
        lb, ub = dgp.bound_smoothing(Gd)
        assert np.abs(lb - g("lb")[s]).max() < 1e-12 and np.abs(ub - g("ub")[s]).max() < 1e-12


def test_planar_tree_round_trip_and_batch_assembly():
    """The reference's round-trip property (tests/test_joint_variables.py:139-156, this time on real
    trees: balanced binary trees of height 2..4 built as there, WITH their parents) and the batched
    goal assembly of BatchProblem against the per-goal graph path."""
    import networkx as nx
    from conftest import planar_tree
    from graphik_amd.robots import RobotPlanar
    from graphik_amd.graphs import ProblemGraphPlanar
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    from graphik_amd.utils import list_to_variable_dict
    np.random.seed(3)
    for height in (2, 3, 4):
        gen = nx.balanced_tree(2, height, create_using=nx.DiGraph)
        gen = nx.relabel_nodes(gen, {node: f"p{node}" for node in gen})
        n = gen.number_of_edges()
        params = {"link_lengths": list_to_variable_dict(np.ones(n)), "num_joints": n,
                  "parents": {k: v for k, v in nx.to_dict_of_lists(gen).items() if v}}
        robot = RobotPlanar(params)
        graph = ProblemGraphPlanar(robot)
        assert len(robot.end_effectors) == 2 ** height
        for _ in range(10):
            q_goal = robot.random_configuration()
            q_rec = graph.joint_variables(graph.realization(q_goal))
            np.testing.assert_allclose([q_goal[k] for k in sorted(q_goal)], [q_rec[k] for k in sorted(q_goal)],
                                       rtol=1e-5)
    for which in ("y5", "bin2"):
        robot, graph = planar_tree(which)
        prob = BatchProblem(graph, use_limits=True, host_only=True)
        rng = np.random.RandomState(1)
        lb, ub = robot.limits_arrays()
        Q = lb + (ub - lb) * rng.rand(6, robot.n)
        Tg = np.stack([[robot.pose(robot.array_to_q(q), ee).as_matrix() for ee in robot.end_effectors] for q in Q])
        D, lo, up = prob.assemble(Tg)
        for b in range(len(Q)):
            Gd = graph.from_pose({ee: Tg[b, i] for i, ee in enumerate(robot.end_effectors)})
            assert np.abs(dgp.distance_matrix_from_graph(Gd) - D[b])[prob.omega != 0].max() < 1e-13
            assert np.array_equal(dgp.adjacency_matrix_from_graph(Gd), prob.omega)
            l1, u1 = dgp.bound_smoothing(Gd)
            l2, u2 = dgp.floyd_warshall_bounds(lo[b:b + 1], up[b:b + 1])
            assert np.abs(l1 - l2[0]).max() < 1e-12 and np.abs(u1 - u2[0]).max() < 1e-12
        qr = prob.joint_variables(np.stack([graph.realization(robot.array_to_q(q)).positions() for q in Q]), Tg)
        assert np.abs(np.mod(qr - Q + np.pi, 2 * np.pi) - np.pi).max() < 1e-9


@pytest.mark.parametrize("name", ["lwa4p", "panda"])
def test_remaining_loaders_match_reference(name):
    """load_schunk_lwa4p / load_panda (roboturdf.py:299-312, 343-356): template, frames, and
    configuration -> pose -> realization -> joint_variables tuples against the reference's loaders
    (tests/golden/loaders_extra.npz, tools/capture_golden_loaders.py)."""
    from graphik_amd.utils.roboturdf import load_panda, load_schunk_lwa4p
    d = load_golden("loaders_extra")
    g = lambda k: d[f"{name}_{k}"]   # noqa: E731
    robot, graph = {"lwa4p": load_schunk_lwa4p, "panda": load_panda}[name]()
    assert robot.n == int(g("n_joints")) and graph.node_ids == list(g("node_ids"))
    assert np.abs(robot.T0_array() - g("T0")).max() == 0.0
    assert _same(graph.dist, g("G_dist")) and _same(graph.lower, g("G_lower")) and _same(graph.upper, g("G_upper"))
    assert np.array_equal(np.where(graph.bounded == 5, 0, graph.bounded), g("G_bounded"))
    L, U = graph.distance_bound_matrices()
    assert np.array_equal(L, g("psi_L")) and np.array_equal(U, g("psi_U"))
    n = robot.n
    for s_ in range(len(g("q_goal"))):
        np.random.seed(s_)
        q = robot.random_configuration()
        assert np.array_equal(robot.q_to_array(q), g("q_goal")[s_])
        T = robot.pose(q, f"p{n}")
        assert np.abs(T.as_matrix() - g("T_goal")[s_]).max() < 1e-13
        G = graph.realization(q)
        assert np.abs(G.positions() - g("X")[s_]).max() < 1e-13
        qr = graph.joint_variables(G, {f"p{n}": T})
        assert np.abs(robot.q_to_array(qr) - g("q_rec")[s_]).max() < 1e-9


# ---- the rest of the ProblemGraph surface (graph_base.py:57-137, graph_revolute.py:325-349) ----------
@pytest.mark.parametrize("name", ["ur10", "planar10"])
def test_graph_api_members_match_reference(name):
    """distance_matrix_from_joints, end_effector_nodes, the base / structure subgraphs and the callable
    nodes view against what the reference's own objects return (tools/capture_golden_api.py)."""
    d = np.load(os.path.join(REPO, "tests", "golden", "graph_api.npz"))
    robot, graph = make_graph(name)
    assert list(graph.node_ids) == list(d[f"{name}_ids"]) == graph.nodes() == list(d[f"{name}_nodes_call"])
    for q, D in zip(d[f"{name}_q"], d[f"{name}_D"]):
        got = graph.distance_matrix_from_joints(robot.array_to_q(q))
        assert np.allclose(got, D, rtol=0, atol=1e-12 * max(1.0, np.abs(D).max()))
    assert list(graph.end_effector_nodes) == list(d[f"{name}_ee_nodes"])
    from graphik_amd.utils.constants import TYPE
    assert ["|".join(t) for _, t in graph.nodes(data=TYPE)] == list(d[f"{name}_types"])
    assert dict(graph.nodes(data=True)).keys() == dict.fromkeys(graph.node_ids).keys()
    for sub in ("base", "structure"):
        S = getattr(graph, sub)
        assert sorted(S.nodes()) == sorted(d[f"{name}_{sub}_nodes"])     # (networkx filters through a set: its order means nothing)
        # (the reference's subgraph views are directed; this graph stores undirected edges)
        want = {frozenset(e.split(">")) for e in d[f"{name}_{sub}_edges"]}
        assert {frozenset(e) for e in S.edges()} == want
        u, v = next(iter(S.edges()))
        assert S[u][v] == graph[u][v]


def test_distance_bounds_from_sampling_matches_reference():
    """graph_revolute.py:325-349 under a fixed numpy seed: the same 2001 random configurations give
    the same LOWER / UPPER for every node pair, and DIST = |D_max - D_min| on the rigid pairs."""
    from graphik_amd.utils.constants import DIST, LOWER, UPPER
    d = np.load(os.path.join(REPO, "tests", "golden", "graph_api.npz"))
    robot, graph = make_graph("ur10")
    assert robot.spherical is False and isinstance(robot.limited_joints, list)
    np.random.seed(11)
    graph.distance_bounds_from_sampling()
    ids = graph.node_ids
    off = ~np.eye(len(ids), dtype=bool)
    for key, M in ((LOWER, graph.lower), (UPPER, graph.upper)):
        ref = d[f"ur10_sampled_{key}"]
        ref = np.where(np.isnan(ref), ref.T, ref)          # (directed there, symmetric here)
        assert np.allclose(M[off], ref[off], rtol=0, atol=1e-9), key
    ref = d[f"ur10_sampled_{DIST}"]
    ref = np.where(np.isnan(ref), ref.T, ref)
    rigid = np.abs(graph.upper ** 2 - graph.lower ** 2) < 1e-5
    assert np.all(np.abs(graph.dist[rigid & off]) < 1e-5) and rigid[off].sum() >= 20


def test_normalize_positions_and_graph_complete_edges():
    """dgp.normalize_positions as the reference's joint-variable tests use it (test_joint_variables.py:49):
    centred, principal axes, invariant under rigid motions up to column signs; graph_complete_edges is the
    module-level spelling of DistanceGraph.complete_edges."""
    rs = np.random.RandomState(3)
    Y = rs.randn(9, 3) * [3.0, 2.0, 0.5]
    Z = dgp.normalize_positions(Y)
    assert np.allclose(Z.mean(0), 0, atol=1e-12)
    C = Z.T.dot(Z)
    assert np.allclose(C - np.diag(np.diag(C)), 0, atol=1e-9)
    R, _ = np.linalg.qr(rs.randn(3, 3))
    Z2 = dgp.normalize_positions(Y.dot(R.T) + [1.0, -2.0, 0.3])
    # same points in the principal frame: equal up to the order / sign of the axes
    assert np.allclose(np.sort(np.abs(Z), axis=1), np.sort(np.abs(Z2), axis=1), atol=1e-9)
    S = dgp.normalize_positions(Y, scale=True)
    assert np.allclose(S, Z * np.abs(Z).max())
    robot, graph = make_graph("planar10")
    G = graph.realization(robot.zero_configuration())
    D = dgp.distance_matrix_from_graph(dgp.graph_complete_edges(G, overwrite=True))
    P = dgp.pos_from_graph(G)
    assert np.allclose(D, ((P[:, None] - P[None]) ** 2).sum(-1), atol=1e-12)


def test_reference_spellings_of_the_robot_constructors():
    """RobotRevolute.from_dh_params / RobotPlanar.from_params under the reference's public names."""
    robot, _ = make_graph("planar10")
    T = robot.from_params()
    assert set(T) == set(robot.joint_ids) and np.allclose(T["p10"].as_matrix()[:2, 2], [10.0, 0.0])
    from graphik_amd.robots import RobotRevolute
    n = 3
    params = {"a": [0.0, 0.5, 0.5], "alpha": [np.pi / 2, 0.0, 0.0], "d": [0.3, 0.0, 0.0], "theta": [0.0, 0.0, 0.0],
              "modified_dh": False, "num_joints": n, "joint_limits_lower": n * [-np.pi], "joint_limits_upper": n * [np.pi]}
    r = RobotRevolute(params)
    T = r.from_dh_params(params)
    for name in r.joint_ids:
        assert np.allclose(T[name].as_matrix(), r.nodes[name]["T0"].as_matrix())
