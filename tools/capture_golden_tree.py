#!/usr/bin/env python3
"""Golden vectors for tree-structured (multi end-effector) robots (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden_tree.py

Runs the reference on the robot of tests/test_joint_variables.py:192-226 (a 5-joint tree with two
end effectors, DH parameters) and records: node order, the edge attribute matrices of
ProblemGraphRevolute, zero-configuration frames, and for a set of seeds the random configuration,
the end-effector poses, the realization (node positions), joint_variables() of it, and one full
solve through RiemannianSolver.solve (distance matrix, omega, bounds, initial point, solution,
recovered angles).  Only numbers are written.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "ref_shims"))

import refcompat  # noqa: E402
import numpy as np  # noqa: E402
from numpy import pi  # noqa: E402

refcompat.patch_skew()
import graphik.solvers.riemannian_solver as rs  # noqa: E402
import graphik.solvers.costs as costs  # noqa: E402
from graphik.solvers.riemannian_solver import RiemannianSolver  # noqa: E402
from graphik.robots import RobotRevolute  # noqa: E402
from graphik.graphs import ProblemGraphRevolute  # noqa: E402
from graphik.utils.dgp import (adjacency_matrix_from_graph, bound_smoothing,  # noqa: E402
                               distance_matrix_from_graph, graph_from_pos, pos_from_graph)
from graphik.utils.utils import list_to_variable_dict  # noqa: E402
from graphik.utils.constants import DIST, LOWER, UPPER, BOUNDED, BELOW, ABOVE  # noqa: E402

for _n in ("jcost", "jgrad", "jhess", "lcost", "lgrad", "lhess"):
    setattr(rs, _n, getattr(costs, _n))

TREE = dict(
    num_joints=5,
    parents={"p0": ["p1"], "p1": ["p2", "p3"], "p2": ["p4"], "p3": ["p5"]},
    a={"p1": 0, "p2": -0.612, "p3": -0.612, "p4": -0.5732, "p5": -0.5732},
    d={"p1": 0.1237, "p2": 0, "p3": 0, "p4": 0, "p5": 0},
    alpha={"p1": pi / 2, "p2": 0, "p3": 0, "p4": 0, "p5": 0},
    theta={"p1": 0, "p2": 0, "p3": 0, "p4": 0, "p5": 0},
    modified_dh=False,
)


def bounded_code(data):
    if BOUNDED not in data:
        return 0
    b = data[BOUNDED]
    if len(b) == 0:
        return 0
    if b[0] is False:
        return 1
    return 2 if b[0] == BELOW else (3 if b[0] == ABOVE else 4)


if __name__ == "__main__":
    robot = RobotRevolute(dict(TREE))
    graph = ProblemGraphRevolute(robot)
    ids = list(graph.node_ids)
    N = len(ids)
    out = {"node_ids": np.array(ids), "end_effectors": np.array(robot.end_effectors),
           "joint_ids": np.array(robot.joint_ids)}
    for key, attr in (("G_dist", DIST), ("G_lower", LOWER), ("G_upper", UPPER)):
        M = np.full((N, N), np.nan)
        for u, v, data in graph.edges(data=True):
            if attr in data:
                M[ids.index(u), ids.index(v)] = M[ids.index(v), ids.index(u)] = data[attr]
        out[key] = M
    Bd = np.full((N, N), -1, dtype=np.int8)
    for u, v, data in graph.edges(data=True):
        Bd[ids.index(u), ids.index(v)] = Bd[ids.index(v), ids.index(u)] = bounded_code(data)
    out["G_bounded"] = Bd
    out["T0"] = np.stack([robot.nodes[j]["T0"].as_matrix() for j in robot.joint_ids])
    psi_L, psi_U = graph.distance_bound_matrices()
    out["psi_L"], out["psi_U"] = psi_L, psi_U
    Q, X, QR, TG = [], [], [], []
    for seed in range(12):
        np.random.seed(seed)
        q = robot.random_configuration()
        T_goal = {ee: robot.pose(list_to_variable_dict(q), ee) if False else robot.pose(q, ee)
                  for ee in robot.end_effectors}
        G = graph.realization(q)
        q_rec = graph.joint_variables(G, T_goal)
        Q.append([q[j] for j in robot.joint_ids[1:]])
        QR.append([q_rec[j] for j in robot.joint_ids[1:]])
        X.append(pos_from_graph(G, ids))
        TG.append(np.stack([T_goal[ee].as_matrix() for ee in robot.end_effectors]))
    out.update(q_goal=np.array(Q), q_rec=np.array(QR), X=np.array(X), T_goal=np.array(TG))
    # full solves through RiemannianSolver.solve (the reference has no solve_with_riemannian for trees)
    sol = {k: [] for k in ("D_goal", "lb", "ub", "Y_init", "Y_sol", "f", "iterations", "q_sol", "pos_err")}
    for g in range(4):
        q = {j: out["q_goal"][g][i] for i, j in enumerate(robot.joint_ids[1:])}
        T_goal = {ee: robot.pose(q, ee) for ee in robot.end_effectors}
        G = graph.from_pose(T_goal)
        D_goal = distance_matrix_from_graph(G)
        omega = adjacency_matrix_from_graph(G)
        lb, ub = bound_smoothing(G)
        Y_init = RiemannianSolver.generate_initialization((lb, ub), 3, omega, psi_L, psi_U)
        solver = RiemannianSolver(graph)
        info = solver.solve(D_goal, omega, use_limits=True, Y_init=Y_init.copy(), jit=False)
        q_sol = graph.joint_variables(graph_from_pos(info["x"], ids), T_goal)
        err = max(np.linalg.norm(robot.pose(q_sol, ee).trans - T_goal[ee].trans) for ee in robot.end_effectors)
        print(f"goal {g}: it={info['iterations']} f={info['f(x)']:.2e} pos_err={err:.2e}", flush=True)
        out["omega"] = omega
        for k, v in (("D_goal", D_goal), ("lb", lb), ("ub", ub), ("Y_init", Y_init), ("Y_sol", info["x"]),
                     ("f", info["f(x)"]), ("iterations", info["iterations"]),
                     ("q_sol", [q_sol[j] for j in robot.joint_ids[1:]]), ("pos_err", err)):
            sol[k].append(v)
    out.update({"sol_" + k: np.array(v) for k, v in sol.items()})
    path = os.path.join(REPO, "tests", "golden", "tree5.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; nodes", ids, "ee", robot.end_effectors)
