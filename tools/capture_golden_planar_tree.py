#!/usr/bin/env python3
"""Golden vectors for PLANAR tree robots (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden_planar_tree.py

The reference's RobotPlanar / ProblemGraphPlanar accept params["parents"]
(graph_planar.py:50-88, robot_planar.py:51-60); its own tree test
(tests/test_joint_variables.py:139-156) builds `parents` but never passes it, so it only ever
exercises chains.  This script runs the reference on two real planar trees -- a 5-joint tree with two
end effectors (one branch point) and the balanced binary tree of height 2 of that test WITH its
parents (6 joints, four end effectors, two of which share their predecessor) -- and records node
order, end effectors, zero-configuration frames, every edge attribute, psi_L / psi_U, and for a set
of seeds the random configuration, end-effector poses, realization, joint_variables() of it, the
goal graph's D_goal / omega and bound_smoothing.  Only numbers are written
(tests/golden/planar_tree.npz).
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
import networkx as nx  # noqa: E402

refcompat.patch_skew()
import graphik.solvers.riemannian_solver as rs  # noqa: E402
import graphik.solvers.costs as costs  # noqa: E402
from graphik.solvers.riemannian_solver import RiemannianSolver  # noqa: E402
from graphik.robots import RobotPlanar  # noqa: E402
from graphik.graphs import ProblemGraphPlanar  # noqa: E402
from graphik.utils.dgp import (adjacency_matrix_from_graph, bound_smoothing,  # noqa: E402
                               distance_matrix_from_graph, graph_from_pos, pos_from_graph)

for _n in ("jcost", "jgrad", "jhess", "lcost", "lgrad", "lhess"):
    setattr(rs, _n, getattr(costs, _n))
from graphik.utils.utils import list_to_variable_dict  # noqa: E402
from graphik.utils.constants import DIST, LOWER, UPPER, BOUNDED, BELOW  # noqa: E402


def tree_params(which):
    if which == "y5":       # p0 - p1 - {p2 - p3, p4 - p5}
        parents = {"p0": ["p1"], "p1": ["p2", "p4"], "p2": ["p3"], "p4": ["p5"]}
        n = 5
        lengths = [1.0, 0.8, 0.6, 0.9, 0.7]
        lim = [np.pi, 2.5, 2.0, np.pi / 2, 3.0]
    else:                   # balanced binary tree, height 2 (test_joint_variables.py:141-146)
        gen = nx.balanced_tree(2, 2, create_using=nx.DiGraph)
        gen = nx.relabel_nodes(gen, {node: f"p{node}" for node in gen})
        parents = {k: v for k, v in nx.to_dict_of_lists(gen).items() if v}
        n = gen.number_of_edges()
        lengths = list(np.ones(n))
        lim = list(np.pi * np.ones(n))
    return {"link_lengths": list_to_variable_dict(lengths), "num_joints": n, "parents": parents,
            "theta": list_to_variable_dict(np.zeros(n)),
            "joint_limits_upper": list_to_variable_dict(lim),
            "joint_limits_lower": list_to_variable_dict([-x for x in lim])}


def bounded_code(data):
    if BOUNDED not in data:
        return 0
    b = data[BOUNDED]
    if isinstance(b, str):
        return 2 if b == BELOW else 3
    return 0 if len(b) == 0 else (1 if b[0] is False else 4)


if __name__ == "__main__":
    out = {}
    for which in ("y5", "bin2"):
        params = tree_params(which)
        robot = RobotPlanar(params)
        graph = ProblemGraphPlanar(robot)
        ids = list(graph.node_ids)
        N = len(ids)
        o = {"node_ids": np.array(ids), "end_effectors": np.array(robot.end_effectors),
             "joint_ids": np.array(robot.joint_ids),
             "link_lengths": np.array([params["link_lengths"][f"p{i}"] for i in range(1, robot.n + 1)]),
             "limits": np.array([params["joint_limits_upper"][f"p{i}"] for i in range(1, robot.n + 1)]),
             "parents_flat": np.array([f"{u}>{v}" for u, kids in params["parents"].items() for v in kids])}
        for key, attr in (("G_dist", DIST), ("G_lower", LOWER), ("G_upper", UPPER)):
            M = np.full((N, N), np.nan)
            for u, v, data in graph.edges(data=True):
                if attr in data:
                    M[ids.index(u), ids.index(v)] = M[ids.index(v), ids.index(u)] = data[attr]
            o[key] = M
        Bd = np.full((N, N), -1, dtype=np.int8)
        for u, v, data in graph.edges(data=True):
            Bd[ids.index(u), ids.index(v)] = Bd[ids.index(v), ids.index(u)] = bounded_code(data)
        o["G_bounded"] = Bd
        o["T0"] = np.stack([robot.nodes[j]["T0"].as_matrix() for j in robot.joint_ids])
        o["psi_L"], o["psi_U"] = graph.distance_bound_matrices()
        Q, X, QR, TG, DG, LB, UB = [], [], [], [], [], [], []
        for seed in range(8):
            np.random.seed(seed)
            q = robot.random_configuration()
            T_goal = {ee: robot.pose(q, ee) for ee in robot.end_effectors}
            G = graph.realization(q)
            q_rec = graph.joint_variables(G)
            Q.append([q[j] for j in robot.joint_ids[1:]])
            QR.append([q_rec[j] for j in robot.joint_ids[1:]])
            X.append(pos_from_graph(G, ids))
            TG.append(np.stack([T_goal[ee].as_matrix() for ee in robot.end_effectors]))
            Gd = graph.from_pose(T_goal)
            DG.append(distance_matrix_from_graph(Gd))
            o["omega"] = adjacency_matrix_from_graph(Gd)
            lb, ub = bound_smoothing(Gd)
            LB.append(lb); UB.append(ub)
        o.update(q_goal=np.array(Q), q_rec=np.array(QR), X=np.array(X), T_goal=np.array(TG),
                 D_goal=np.array(DG), lb=np.array(LB), ub=np.array(UB))
        # full solves through RiemannianSolver.solve (the reference has no solve_with_riemannian for
        # trees): initial point, solution, cost, iteration count, recovered angles, worst EE error
        sol = {k: [] for k in ("Y_init", "Y_sol", "f", "iterations", "q_sol", "pos_err")}
        psi_L, psi_U = o["psi_L"], o["psi_U"]
        for g in range(4):
            q = {j: Q[g][i] for i, j in enumerate(robot.joint_ids[1:])}
            T_goal = {ee: robot.pose(q, ee) for ee in robot.end_effectors}
            Gd = graph.from_pose(T_goal)
            Y_init = RiemannianSolver.generate_initialization((LB[g], UB[g]), 2, o["omega"], psi_L, psi_U)
            info = RiemannianSolver(graph).solve(DG[g], o["omega"], use_limits=True, Y_init=Y_init.copy(), jit=False)
            q_sol = graph.joint_variables(graph_from_pos(info["x"], ids))
            err = max(np.linalg.norm(robot.pose(q_sol, ee).trans - T_goal[ee].trans) for ee in robot.end_effectors)
            print(f"  {which} goal {g}: it={info['iterations']} f={info['f(x)']:.2e} pos_err={err:.2e}", flush=True)
            for k, v in (("Y_init", Y_init), ("Y_sol", info["x"]), ("f", info["f(x)"]),
                         ("iterations", info["iterations"]), ("q_sol", [q_sol[j] for j in robot.joint_ids[1:]]),
                         ("pos_err", err)):
                sol[k].append(v)
        o.update({"sol_" + k: np.array(v) for k, v in sol.items()})
        out.update({f"{which}_{k}": v for k, v in o.items()})
        print(which, "nodes", ids, "ee", robot.end_effectors, "joints", robot.joint_ids)
    path = os.path.join(REPO, "tests", "golden", "planar_tree.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
