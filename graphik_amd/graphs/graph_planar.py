"""Distance-geometric problem graph of a planar revolute chain or tree
(graphik/graphs/graph_planar.py).  Node order of a chain: p0, x, y, p1, ..., pn."""
from math import sqrt

import numpy as np

from .graph_base import ProblemGraph, B_BELOW, B_EMPTY
from ..utils.constants import BASE, END_EFFECTOR, POS, ROBOT, ROOT, TYPE
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
        # node order of nx.compose(base, structure) (graph_planar.py:17-23): p0, x, y, then the joints
        # in the order the end effectors' paths visit them (a chain: p1, ..., pn)
        ids = ["p0", "x", "y"]
        for ee in robot.end_effectors:
            for node in robot.kinematic_map[ROOT][ee]:
                if node not in ids:
                    ids.append(node)
        super().__init__(robot, params, ids)
        # base (graph_planar.py:30-48); x axis mirrored as in the reference
        for name, pos, typ in (("p0", [0, 0], [BASE, ROBOT]), ("x", [-1, 0], [BASE]),
                               ("y", [0, 1], [BASE])):
            self.nodes[name][POS] = np.array(pos, dtype=float)
            self.nodes[name][TYPE] = typ
        for u, v in (("p0", "x"), ("p0", "y"), ("x", "y")):
            d = np.linalg.norm(self.nodes[u][POS] - self.nodes[v][POS])
            self.set_edge(u, v, dist=d, lower=d, upper=d, bounded=B_EMPTY)
        # structure (:50-88): one pass per end effector over its path from the root.  Re-adding a
        # node resets its TYPE (networkx add_nodes_from updates the attribute dict), exactly as there.
        self.structure_edges = []            # (pred, cur) in insertion order: parents before children
        for ee in robot.end_effectors:
            k_map = robot.kinematic_map[ROOT][ee]
            for idx, cur in enumerate(k_map):
                self.nodes[cur][TYPE] = [ROBOT] + ([BASE] if cur == ROOT else [])
                if idx:
                    pred = k_map[idx - 1]
                    d = np.linalg.norm(robot.nodes[cur]["T0"].trans - robot.nodes[pred]["T0"].trans)
                    self.set_edge(pred, cur, dist=d, lower=d, upper=d, bounded=B_EMPTY)
                    if (pred, cur) not in self.structure_edges:
                        self.structure_edges.append((pred, cur))
                    if cur in robot.end_effectors:
                        self.nodes[cur][TYPE] += [END_EFFECTOR]
                        self.nodes[pred][TYPE] += [END_EFFECTOR]
        self.set_limits()
        self.root_angle_limits()

    def _limit(self, l1, l2, node):
        lim = max(abs(self.robot.ub[node]), abs(self.robot.lb[node]))
        return l1 + l2, sqrt(l1 ** 2 + l2 ** 2 - 2 * l1 * l2 * np.cos(np.pi - lim))

    def set_limits(self):
        """two-apart pairs u -> mid -> v along the tree (graph_planar.py:110-134): UPPER = l1 + l2,
        LOWER from the (symmetric) limit of joint v, BOUNDED = "below"."""
        robot = self.robot
        for u in [n for n in self.node_ids if n in robot.children]:
            for mid in robot.children[u]:
                for v in robot.children[mid]:
                    up, lo = self._limit(robot.l[mid], robot.l[v], v)
                    self.set_edge(u, v, lower=lo, upper=up, bounded=B_BELOW)

    def root_angle_limits(self):
        """x -- every child of the root (graph_planar.py:90-108)"""
        l1 = np.linalg.norm(self.nodes["x"][POS])
        for node in self.robot.children[ROOT]:
            l2 = self.dist[self.index(ROOT), self.index(node)]
            up, lo = self._limit(l1, l2, node)
            self.set_edge("x", node, lower=lo, upper=up, bounded=B_BELOW)

    def _pose_goal(self, T_goal):
        """graph_planar.py:136-145: a goal pose pins its node and the node's parent."""
        pos = {}
        for u, T in T_goal.items():
            v = self.robot.parent.get(u)
            if v is None:                     # the root has no predecessor: nothing is pinned
                continue
            M = as_matrix(T)
            d = self.dist[self.index(v), self.index(u)]
            pos[u] = M[:2, 2]
            pos[v] = M[:2, 2] - M[:2, 0] * d
        return pos

    def joint_variables(self, G, T_final=None):
        P = G if isinstance(G, np.ndarray) else G.positions()
        q = joint_variables_planar_batch(self, P[None])[0]
        # keys in the order the reference's loop over the structure edges creates them
        return {v: float(q[int(v[1:]) - 1]) for _, v in self.structure_edges}

    def get_pose(self, joint_angles, query_node):
        return self.robot.pose(joint_angles, query_node)


def joint_variables_planar_batch(graph, P):
    """graph_planar.py:147-176 over B realisations (chains and trees).  P [B,N,2] -> q [B,n], column
    i - 1 = joint p_i."""
    n = graph.robot.n
    ix = graph.index
    B = P.shape[0]
    q = np.zeros((B, n))
    target = np.array([[0.0, 0.0], [-1.0, 0.0], [0.0, 1.0]])
    for b in range(B):
        R_, _ = best_fit_transform(np.vstack((P[b, ix("p0")], P[b, ix("x")], P[b, ix("y")])), target)
        R = {ROOT: np.identity(2)}
        for u, v in graph.structure_edges:
            diff = R_ @ (P[b, ix(v)] - P[b, ix(u)])
            sol = R[u].T @ (diff / np.linalg.norm(diff))
            th = np.arctan2(sol[1], sol[0])
            q[b, int(v[1:]) - 1] = wraptopi(th)
            c, s = np.cos(th), np.sin(th)
            R[v] = R[u] @ np.array([[c, -s], [s, c]])
    return q
