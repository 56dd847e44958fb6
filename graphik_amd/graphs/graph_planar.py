"""Distance-geometric problem graph of a planar revolute chain
(graphik/graphs/graph_planar.py).  Node order p0, x, y, p1, ..., pn."""
from math import sqrt

import numpy as np

from .graph_base import ProblemGraph, B_BELOW, B_EMPTY
from ..utils.constants import BASE, END_EFFECTOR, POS, ROBOT, TYPE
from ..utils.lie import as_matrix
from ..utils.utils import wraptopi


def best_fit_transform(A, B):
    """Least-squares rigid fit A -> B without reflection handling
    (graphik/utils/geometry.py:60-100)."""
    cA, cB = A.mean(axis=0), B.mean(axis=0)
    H = (A - cA).T @ (B - cB)
    U, S, Vt = np.linalg.svd(H)
    R = Vt.T @ U.T
    return R, cB - R @ cA


class ProblemGraphPlanar(ProblemGraph):
    planar_bounded = True

    def __init__(self, robot, params={}):
        n = robot.n
        super().__init__(robot, params, ["p0", "x", "y"] + [f"p{i}" for i in range(1, n + 1)])
        # base (graph_planar.py:30-48); x axis mirrored as in the reference
        for name, pos, typ in (("p0", [0, 0], [BASE, ROBOT]), ("x", [-1, 0], [BASE]),
                               ("y", [0, 1], [BASE])):
            self.nodes[name][POS] = np.array(pos, dtype=float)
            self.nodes[name][TYPE] = typ
        for u, v in (("p0", "x"), ("p0", "y"), ("x", "y")):
            d = np.linalg.norm(self.nodes[u][POS] - self.nodes[v][POS])
            self.set_edge(u, v, dist=d, lower=d, upper=d, bounded=B_EMPTY)
        # structure (:50-88)
        for i in range(1, n + 1):
            d = np.linalg.norm(robot.nodes[f"p{i}"]["T0"].trans - robot.nodes[f"p{i - 1}"]["T0"].trans)
            self.set_edge(f"p{i - 1}", f"p{i}", dist=d, lower=d, upper=d, bounded=B_EMPTY)
            self.nodes[f"p{i}"][TYPE] = [ROBOT]
        self.nodes[f"p{n}"][TYPE] += [END_EFFECTOR]
        self.nodes[f"p{n - 1}"][TYPE] = self.nodes[f"p{n - 1}"].get(TYPE, []) + [END_EFFECTOR]
        self.set_limits()
        self.root_angle_limits()

    def _limit(self, l1, l2, node):
        lim = max(abs(self.robot.ub[node]), abs(self.robot.lb[node]))
        return l1 + l2, sqrt(l1 ** 2 + l2 ** 2 - 2 * l1 * l2 * np.cos(np.pi - lim))

    def set_limits(self):
        """two-apart pairs p_{i-2}, p_i (graph_planar.py:110-134)"""
        for i in range(2, self.robot.n + 1):
            l1, l2 = self.robot.l[f"p{i - 1}"], self.robot.l[f"p{i}"]
            up, lo = self._limit(l1, l2, f"p{i}")
            self.set_edge(f"p{i - 2}", f"p{i}", lower=lo, upper=up, bounded=B_BELOW)

    def root_angle_limits(self):
        """x -- p1 (graph_planar.py:90-108)"""
        l1 = np.linalg.norm(self.nodes["x"][POS])
        l2 = self.dist[self.index("p0"), self.index("p1")]
        up, lo = self._limit(l1, l2, "p1")
        self.set_edge("x", "p1", lower=lo, upper=up, bounded=B_BELOW)

    def _pose_goal(self, T_goal):
        """graph_planar.py:136-145: p_n and its predecessor are pinned by an SE(2) goal."""
        pos = {}
        for u, T in T_goal.items():
            i = int(u[1:])
            if i == 0:
                continue
            M = as_matrix(T)
            v = f"p{i - 1}"
            d = self.dist[self.index(v), self.index(u)]
            pos[u] = M[:2, 2]
            pos[v] = M[:2, 2] - M[:2, 0] * d
        return pos

    def joint_variables(self, G, T_final=None):
        P = G if isinstance(G, np.ndarray) else G.positions()
        return self.robot.array_to_q(joint_variables_planar_batch(self, P[None])[0])

    def get_pose(self, joint_angles, query_node):
        return self.robot.pose(joint_angles, query_node)


def joint_variables_planar_batch(graph, P):
    """graph_planar.py:147-176 over B realisations.  P [B,N,2] -> q [B,n]."""
    n = graph.robot.n
    ix = graph.index
    B = P.shape[0]
    q = np.zeros((B, n))
    target = np.array([[0.0, 0.0], [-1.0, 0.0], [0.0, 1.0]])
    for b in range(B):
        R_, _ = best_fit_transform(np.vstack((P[b, ix("p0")], P[b, ix("x")], P[b, ix("y")])), target)
        R = np.identity(2)
        for i in range(1, n + 1):
            diff = R_ @ (P[b, ix(f"p{i}")] - P[b, ix(f"p{i - 1}")])
            sol = R.T @ (diff / np.linalg.norm(diff))
            th = np.arctan2(sol[1], sol[0])
            q[b, i - 1] = wraptopi(th)
            c, s = np.cos(th), np.sin(th)
            R = R @ np.array([[c, -s], [s, c]])
    return q
