"""Distance-geometric problem graph of a 3-D revolute chain or tree
(graphik/graphs/graph_revolute.py).  Node order of a chain: p0, x, y, q0, p1, q1, ..., pn, qn."""
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
        # node order of nx.compose(base, structure) (graph_revolute.py:21-27): the base nodes, then
        # (p, q) of every joint in the order the end effectors' paths visit them -- for a chain
        # p0, x, y, q0, p1, q1, ..., pn, qn
        ids = ["p0", "x", "y", "q0"]
        for ee in robot.end_effectors:
            for node in robot.kinematic_map[ROOT][ee]:
                for name in (node, AUX_PREFIX + node[1:]):
                    if name not in ids:
                        ids.append(name)
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
        for ee in robot.end_effectors:
            k_map = robot.kinematic_map[ROOT][ee]
            for idx, cur in enumerate(k_map):
                aux = AUX_PREFIX + cur[1:]
                T0 = robot.nodes[cur]["T0"]
                pos[cur], pos[aux] = T0.trans, T0.dot(tz).trans
                d = np.linalg.norm(pos[cur] - pos[aux])
                self.set_edge(cur, aux, dist=d, lower=d, upper=d, bounded=B_EMPTY)
                if idx:
                    pred = k_map[idx - 1]
                    for u in (pred, AUX_PREFIX + pred[1:]):
                        for v in (cur, aux):
                            d = np.linalg.norm(pos[u] - pos[v])
                            self.set_edge(u, v, dist=d, lower=d, upper=d, bounded=B_EMPTY)
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
        for ee in robot.end_effectors:                     # graph_revolute.py:180-241
            k_map = robot.kinematic_map[ROOT][ee]
            for idx in range(2, len(k_map)):
                cur, prev = k_map[idx], k_map[idx - 2]
                mid = robot.kinematic_map[prev][cur][1]
                for a0 in "pq":
                    for a1 in "pq":
                        T0, T1, T2 = (robot.nodes[k]["T0"] for k in (prev, mid, cur))
                        if a0 == AUX_PREFIX:
                            T0 = T0.dot(tz)
                        if a1 == AUX_PREFIX:
                            T2 = T2.dot(tz)
                        code = self._limit_edge(f"{a0}{prev[1:]}", f"{a1}{cur[1:]}", T0, T1, T2,
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
        node, or an N x 3 array in node order.  T_final: pose of the end effector, or a dict
        {end effector: pose} (robots with several end effectors)."""
        P = G if isinstance(G, np.ndarray) else G.positions()
        T_fin = None
        if T_final is not None:
            if not isinstance(T_final, dict):
                T_final = {self.robot.end_effectors[0]: T_final}
            T_fin = {ee: as_matrix(T)[None] for ee, T in T_final.items()}
        q = joint_variables_revolute_batch(self, P[None], T_fin)[0]
        return self.robot.array_to_q(q)

    def distance_bounds_from_sampling(self, samples=2000):
        """graph_revolute.py:325-349: LOWER / UPPER of EVERY node pair from the extremes of
        distance_matrix_from_joints over random configurations (one np.random.rand() per joint and
        sample, as the reference draws them).  Kept: DIST is set to |D_max - D_min| -- not to the
        distance -- where the squared extremes differ by less than 1e-5.  Not kept: the self-loops
        the reference adds on the diagonal (this graph has no representation for them)."""
        robot = self.robot
        D_min = self.distance_matrix_from_joints(robot.random_configuration())
        D_max = D_min.copy()
        for _ in range(samples):
            D = self.distance_matrix_from_joints(robot.random_configuration())
            np.maximum(D_max, D, out=D_max)
            np.minimum(D_min, D, out=D_min)
        ids = self.node_ids
        for i in range(len(ids)):
            for j in range(i + 1, len(ids)):
                rigid = abs(D_max[i, j] - D_min[i, j]) < 1e-5
                self.set_edge(ids[i], ids[j], lower=np.sqrt(D_min[i, j]), upper=np.sqrt(D_max[i, j]),
                              dist=abs(D_max[i, j] - D_min[i, j]) if rigid else None)

    def get_pose(self, joint_angles, query_node):
        T = self.robot.pose(joint_angles, "p" + query_node[1:])
        return T.dot(trans_axis(self.axis_length, "z")) if query_node[0] == AUX_PREFIX else T


def joint_variables_revolute_batch(graph, P, T_final=None, tol=1e-10):
    """Vectorised restatement of ProblemGraphRevolute.joint_variables over B realisations.
    P [B,N,3] (node order of `graph`); T_final: [B,4,4] poses of the (first) end effector, a dict
    {end effector: [B,4,4]}, or None  ->  q [B,n] (columns p1..pn)."""
    robot = graph.robot
    n, a = robot.n, graph.axis_length
    ix = graph.index
    B = P.shape[0]
    if T_final is not None and not isinstance(T_final, dict):
        T_final = {robot.end_effectors[0]: T_final}
    unit = lambda v: v / np.where(np.linalg.norm(v, axis=-1, keepdims=True) == 0, 1.0,
                                  np.linalg.norm(v, axis=-1, keepdims=True))
    p0 = P[:, ix("p0")]
    x, y, z = (unit(P[:, ix(k)] - p0) for k in ("x", "y", "q0"))
    R = np.stack((x, -y, z), axis=-1)                      # columns x, -y, z  (:270-279)
    Rt = np.swapaxes(R, 1, 2)
    to_base = lambda v: np.einsum("bij,bj->bi", Rt, v - p0)  # B.inv().dot(v)
    Tz = np.identity(4)
    Tz[2, 3] = a
    T = {ROOT: np.broadcast_to(as_matrix(robot.T_base), (B, 4, 4)).copy()}
    theta = np.zeros((B, n))
    for ee in robot.end_effectors:                         # :285-316, one path per end effector
        k_map = robot.kinematic_map[ROOT][ee]
        T_rel = None
        for idx in range(1, len(k_map)):
            cur, pred = k_map[idx], k_map[idx - 1]
            T0_prev, T0_cur = robot.nodes[pred]["T0"].as_matrix(), robot.nodes[cur]["T0"].as_matrix()
            inv_prev0 = np.linalg.inv(T0_prev)
            T_rel = inv_prev0 @ T0_cur
            qs_0 = (inv_prev0 @ T0_cur @ Tz)[:3, 3]
            pc, qc = P[:, ix(cur)], P[:, ix(AUX_PREFIX + cur[1:])]
            qn = to_base(pc + unit(qc - pc))
            T_prev = T[pred]
            qs = np.einsum("bji,bj->bi", T_prev[:, :3, :3], qn - T_prev[:, :3, 3])
            col = int(cur[1:]) - 1
            theta[:, col] = np.arctan2(qs_0[0] * qs[:, 1] - qs_0[1] * qs[:, 0],
                                       qs_0[0] * qs[:, 0] + qs_0[1] * qs[:, 1])   # :308
            c, s = np.cos(theta[:, col]), np.sin(theta[:, col])
            Rz = np.zeros((B, 4, 4))
            Rz[:, 0, 0], Rz[:, 0, 1], Rz[:, 1, 0], Rz[:, 1, 1] = c, -s, s, c
            Rz[:, 2, 2] = Rz[:, 3, 3] = 1.0
            T[cur] = T_prev @ Rz @ T_rel                                               # :310
        if T_final is not None and ee in T_final and \
                np.linalg.norm(np.cross(T_rel[:3, 3], [0, 0, 1])) < tol:               # :314
            T_th = np.linalg.inv(T[ee]) @ T_final[ee]
            col = int(ee[1:]) - 1
            theta[:, col] = wraptopi(theta[:, col] + np.arctan2(T_th[:, 1, 0], T_th[:, 0, 0]))
    return theta
