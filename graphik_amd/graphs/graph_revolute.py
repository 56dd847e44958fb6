"""Distance-geometric problem graph of a 3-D revolute chain
(graphik/graphs/graph_revolute.py).  Node order p0, x, y, q0, p1, q1, ..., pn, qn."""
import numpy as np

from .graph_base import ProblemGraph, B_ABOVE, B_BELOW, B_EMPTY, B_FALSE, B_NONE
from ..utils.constants import BASE, POS, ROBOT, ROOT, TYPE, AUX_PREFIX
from ..utils.lie import SE3, SO3, as_matrix, rot_axis, trans_axis
from ..utils.utils import wraptopi


def max_min_distance_revolute(r, P, C, N):
    """Extreme distances between point P and a circle of radius r, centre C, normal N
    (graphik/utils/geometry.py:45-58)."""
    delta = P - C
    ax = N.dot(delta) ** 2
    rad = np.linalg.norm(np.cross(N, delta))
    d_min_s = ax + (rad - r) ** 2
    d_max_s = ax + (rad + r) ** 2
    return (np.sqrt(d_max_s) if d_max_s > 0 else 0), (np.sqrt(d_min_s) if d_min_s > 0 else 0)


def _classify(d, d_max, d_min):
    """Exact float comparisons, as the reference (graph_revolute.py:134-141, 210-217)."""
    if d_max == d_min:
        return B_FALSE
    if d == d_max:
        return B_BELOW
    if d == d_min:
        return B_ABOVE
    return B_NONE


class ProblemGraphRevolute(ProblemGraph):
    def __init__(self, robot, params={}):
        n = robot.n
        ids = ["p0", "x", "y", "q0"] + [f"{c}{i}" for i in range(1, n + 1) for c in "pq"]
        super().__init__(robot, params, ids)
        self._base_subgraph()
        self._structure_graph()
        self.set_limits()
        self.root_angle_limits()

    # graph_revolute.py:32-57
    def _base_subgraph(self):
        a = self.axis_length
        for name, pos, typ in (("p0", [0, 0, 0], [ROBOT, BASE]), ("x", [a, 0, 0], [BASE]),
                               ("y", [0, -a, 0], [BASE]), ("q0", [0, 0, a], [ROBOT, BASE])):
            self.nodes[name][POS] = np.array(pos, dtype=float)
            self.nodes[name][TYPE] = typ
        for u, v in (("p0", "x"), ("p0", "y"), ("p0", "q0"), ("x", "y"), ("y", "q0"), ("q0", "x")):
            d = np.linalg.norm(self.nodes[u][POS] - self.nodes[v][POS])
            self.set_edge(u, v, dist=d, lower=d, upper=d, bounded=B_EMPTY)

    # graph_revolute.py:59-106
    def _structure_graph(self):
        tz = trans_axis(self.axis_length, "z")
        robot = self.robot
        pos = {}
        for i in range(robot.n + 1):
            cur, aux = f"p{i}", f"q{i}"
            T0 = robot.nodes[cur]["T0"]
            pos[cur], pos[aux] = T0.trans, T0.dot(tz).trans
            d = np.linalg.norm(pos[cur] - pos[aux])
            self.set_edge(cur, aux, dist=d, lower=d, upper=d, bounded=B_EMPTY)
            if i:
                for u in (f"p{i - 1}", f"q{i - 1}"):
                    for v in (cur, aux):
                        d = np.linalg.norm(pos[u] - pos[v])
                        self.set_edge(u, v, dist=d, lower=d, upper=d, bounded=B_EMPTY)
            if i:
                self.nodes[cur][TYPE] = [ROBOT]
                self.nodes[aux][TYPE] = [ROBOT]

    def _limit_edge(self, u, v, T0, T1, T2, T_rel, ub):
        """Shared body of set_limits / root_angle_limits (graph_revolute.py:120-165, 196-239)."""
        N = T1.as_matrix()[0:3, 2]
        C = T1.trans + (N.dot(T2.trans - T1.trans)) * N
        r = np.linalg.norm(T2.trans - C)
        P = T0.trans
        d_max, d_min = max_min_distance_revolute(r, P, C, N)
        d = np.linalg.norm(T2.trans - T0.trans)
        code = _classify(d, d_max, d_min)
        if code in (B_BELOW, B_ABOVE):
            d_limit = np.linalg.norm(T1.dot(rot_axis(ub, "z")).dot(T_rel).trans - T0.trans)
            if code == B_ABOVE:
                d_max = d_limit
            else:
                d_min = d_limit
        self.set_edge(u, v, dist=(d_max if d_max == d_min else None), lower=d_min, upper=d_max,
                      bounded=code)
        return code

    def set_limits(self):
        robot, tz = self.robot, trans_axis(self.axis_length, "z")
        limited = []
        for idx in range(2, robot.n + 1):
            cur, mid, prev = f"p{idx}", f"p{idx - 1}", f"p{idx - 2}"
            for a0 in "pq":
                for a1 in "pq":
                    T0, T1, T2 = (robot.nodes[k]["T0"] for k in (prev, mid, cur))
                    if a0 == AUX_PREFIX:
                        T0 = T0.dot(tz)
                    if a1 == AUX_PREFIX:
                        T2 = T2.dot(tz)
                    code = self._limit_edge(f"{a0}{idx - 2}", f"{a1}{idx}", T0, T1, T2,
                                            T1.inv().dot(T2), robot.ub[cur])
                    if code in (B_BELOW, B_ABOVE):
                        limited.append(cur)
        self.limited_joints = limited

    def root_angle_limits(self):
        robot, tz = self.robot, trans_axis(self.axis_length, "z")
        T1 = robot.nodes[ROOT]["T0"]
        for base_node in ("x", "y"):
            for node in ("p1", "q1"):
                T0 = SE3(SO3.identity(), np.asarray(self.nodes[base_node][POS], dtype=float))
                T2 = robot.nodes["p1"]["T0"] if node[0] == "p" else robot.nodes["p1"]["T0"].dot(tz)
                code = self._limit_edge(base_node, node, T0, T1, T2, T1.inv().dot(T2),
                                        robot.ub["p1"])
                if code in (B_BELOW, B_ABOVE):
                    self.limited_joints += ["p1"]

    # graph_revolute.py:243-249
    def _pose_goal(self, T_goal):
        pos = {}
        tz = trans_axis(self.axis_length, "z")
        for u, T in T_goal.items():
            T = SE3.from_matrix(as_matrix(T))
            pos[u] = T.trans
            pos[AUX_PREFIX + u[1:]] = T.dot(tz).trans
        return pos

    def joint_variables(self, G, T_final=None):
        """Joint angles of a realisation (graph_revolute.py:251-318).  G: graph with POS on every
        node, or an N x 3 array in node order."""
        P = G if isinstance(G, np.ndarray) else G.positions()
        T_fin = None
        if T_final is not None:
            T_fin = as_matrix(T_final[self.robot.end_effectors[0]] if isinstance(T_final, dict)
                              else T_final)
        q = joint_variables_revolute_batch(self, P[None], None if T_fin is None else T_fin[None])[0]
        return self.robot.array_to_q(q)

    def get_pose(self, joint_angles, query_node):
        T = self.robot.pose(joint_angles, "p" + query_node[1:])
        return T.dot(trans_axis(self.axis_length, "z")) if query_node[0] == AUX_PREFIX else T


def joint_variables_revolute_batch(graph, P, T_final=None, tol=1e-10):
    """Vectorised restatement of ProblemGraphRevolute.joint_variables over B realisations.
    P [B,N,3] (node order of `graph`), T_final [B,4,4] or None  ->  q [B,n]."""
    robot = graph.robot
    n, a = robot.n, graph.axis_length
    ix = graph.index
    B = P.shape[0]
    unit = lambda v: v / np.where(np.linalg.norm(v, axis=-1, keepdims=True) == 0, 1.0,
                                  np.linalg.norm(v, axis=-1, keepdims=True))
    p0 = P[:, ix("p0")]
    x, y, z = (unit(P[:, ix(k)] - p0) for k in ("x", "y", "q0"))
    R = np.stack((x, -y, z), axis=-1)                      # columns x, -y, z  (:270-279)
    Rt = np.swapaxes(R, 1, 2)
    to_base = lambda v: np.einsum("bij,bj->bi", Rt, v - p0)  # B.inv().dot(v)
    T0 = robot.T0_array()
    Tz = np.identity(4)
    Tz[2, 3] = a
    T_prev = np.broadcast_to(as_matrix(robot.T_base), (B, 4, 4)).copy()
    theta = np.zeros((B, n))
    T_rel = None
    for idx in range(1, n + 1):
        inv_prev0 = np.linalg.inv(T0[idx - 1])
        T_rel = inv_prev0 @ T0[idx]
        qs_0 = (inv_prev0 @ T0[idx] @ Tz)[:3, 3]
        pc, qc = P[:, ix(f"p{idx}")], P[:, ix(f"q{idx}")]
        qn = to_base(pc + unit(qc - pc))
        qs = np.einsum("bji,bj->bi", T_prev[:, :3, :3], qn - T_prev[:, :3, 3])
        theta[:, idx - 1] = np.arctan2(qs_0[0] * qs[:, 1] - qs_0[1] * qs[:, 0],
                                       qs_0[0] * qs[:, 0] + qs_0[1] * qs[:, 1])   # :308
        c, s = np.cos(theta[:, idx - 1]), np.sin(theta[:, idx - 1])
        Rz = np.zeros((B, 4, 4))
        Rz[:, 0, 0], Rz[:, 0, 1], Rz[:, 1, 0], Rz[:, 1, 1] = c, -s, s, c
        Rz[:, 2, 2] = Rz[:, 3, 3] = 1.0
        T_prev = T_prev @ Rz @ T_rel                                                   # :310
    if T_final is not None and np.linalg.norm(np.cross(T_rel[:3, 3], [0, 0, 1])) < tol:  # :314
        T_th = np.linalg.inv(T_prev) @ T_final
        theta[:, n - 1] = wraptopi(theta[:, n - 1] + np.arctan2(T_th[:, 1, 0], T_th[:, 0, 0]))
    return theta
