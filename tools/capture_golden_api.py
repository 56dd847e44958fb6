#!/usr/bin/env python3
"""The remaining members of the reference's ProblemGraph surface (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden_api.py

Runs the reference's own objects (graph_base.py:57-137, graph_revolute.py:325-349) and records what
the host mirror has to reproduce -> tests/golden/graph_api.npz:
  * distance_matrix_from_joints(q) for seeded configurations (UR10, planar-10, the 5-joint tree),
  * end_effector_nodes, the node and edge sets of the `base` and `structure` subgraph views,
  * nodes(), nodes(data=TYPE) in networkx' order,
  * distance_bounds_from_sampling() under a fixed numpy seed (UR10): LOWER / UPPER / DIST after it.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import capture_golden as cg  # noqa: E402  (sets up the shims and patches, imports the reference)
import numpy as np  # noqa: E402
from graphik.utils.constants import TYPE, LOWER, UPPER, DIST  # noqa: E402
from graphik.utils.roboturdf import load_ur10  # noqa: E402
from graphik.robots import RobotPlanar  # noqa: E402
from graphik.graphs import ProblemGraphPlanar  # noqa: E402
from graphik.utils.utils import list_to_variable_dict  # noqa: E402


def planar10():
    n = 10
    robot = RobotPlanar({"link_lengths": list_to_variable_dict(np.ones(n)), "theta": list_to_variable_dict(np.zeros(n)),
                         "joint_limits_upper": list_to_variable_dict(np.pi * np.ones(n)),
                         "joint_limits_lower": list_to_variable_dict(-np.pi * np.ones(n)), "num_joints": n})
    return robot, ProblemGraphPlanar(robot)


def edge_matrix(G, ids, key):
    M = np.full((len(ids), len(ids)), np.nan)
    for u, v, d in G.edges(data=True):
        if key in d:
            M[ids.index(u), ids.index(v)] = d[key]
    return M


if __name__ == "__main__":
    out = {}
    for name, make in (("ur10", load_ur10), ("planar10", planar10)):
        robot, graph = make()
        ids = list(graph.node_ids)
        Q, D = [], []
        for seed in range(5):
            np.random.seed(seed)
            q = robot.random_configuration()
            Q.append([q[f"p{i}"] for i in range(1, robot.n + 1)])
            D.append(graph.distance_matrix_from_joints(q))
        out[f"{name}_q"] = np.array(Q)
        out[f"{name}_D"] = np.array(D)
        out[f"{name}_ids"] = np.array(ids)
        out[f"{name}_ee_nodes"] = np.array(list(graph.end_effector_nodes))
        for sub in ("base", "structure"):
            S = getattr(graph, sub)
            out[f"{name}_{sub}_nodes"] = np.array(list(S.nodes()))
            out[f"{name}_{sub}_edges"] = np.array(sorted(f"{u}>{v}" for u, v in S.edges()))
        out[f"{name}_nodes_call"] = np.array(list(graph.nodes()))
        out[f"{name}_types"] = np.array(["|".join(t) for _, t in graph.nodes(data=TYPE)])
        print(name, len(ids), "nodes; end effector nodes", list(graph.end_effector_nodes),
              "; base", len(out[f"{name}_base_edges"]), "structure", len(out[f"{name}_structure_edges"]), "edges")
    robot, graph = load_ur10()
    ids = list(graph.node_ids)
    np.random.seed(11)
    graph.distance_bounds_from_sampling()
    for key in (LOWER, UPPER, DIST):
        out[f"ur10_sampled_{key}"] = edge_matrix(graph, ids, key)
    path = os.path.join(cg.OUT, "graph_api.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
