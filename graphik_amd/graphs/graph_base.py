"""Array-backed problem graphs (host side).

`DistanceGraph` stores what the reference keeps in networkx node/edge attribute dicts
(graphik/graphs/graph_base.py, graphik/utils/constants.py) as dense N x N matrices: DIST
("weight"), LOWER, UPPER (NaN = attribute absent) and a small integer code for BOUNDED.  The
dict-style accessors the reference's callers use (G.nodes[n][POS], G[u][v][DIST],
G.edges(data=True), number_of_nodes()) are provided as light views.
"""
import numpy as np

from ..utils.constants import (ABOVE, BASE, BELOW, BOUNDED, DIST, END_EFFECTOR, LOWER, OBSTACLE, POS, ROBOT,
                               TYPE, UPPER, MAIN_PREFIX)
from ..utils.lie import as_matrix

# BOUNDED codes
B_NOEDGE, B_EMPTY, B_FALSE, B_BELOW, B_ABOVE, B_NONE, B_ABSENT = -1, 0, 1, 2, 3, 4, 5


class _Adj:
    def __init__(self, g, i):
        self._g, self._i = g, i

    def __getitem__(self, v):
        g, i, j = self._g, self._i, self._g.index(v)
        if not g.edge[i, j]:
            raise KeyError(v)
        d = {}
        if not np.isnan(g.dist[i, j]):
            d[DIST] = g.dist[i, j]
        if not np.isnan(g.lower[i, j]):
            d[LOWER] = g.lower[i, j]
        if not np.isnan(g.upper[i, j]):
            d[UPPER] = g.upper[i, j]
        code = g.bounded[i, j]
        if code != B_ABSENT:
            d[BOUNDED] = g._bounded_value(code)
        return d

    def __contains__(self, v):
        return v in self._g._idx and bool(self._g.edge[self._i, self._g.index(v)])


class _Nodes(dict):
    """graph.nodes: name -> attribute dict, and callable like networkx' NodeView --
    nodes() lists the names, nodes(data=True) (name, attributes), nodes(data=KEY) (name, value)."""

    def __call__(self, data=False, default=None):
        if data is False:
            return list(self)
        if data is True:
            return list(self.items())
        return [(n, a.get(data, default)) for n, a in self.items()]


class DistanceGraph:
    planar_bounded = False  # planar graphs store BOUNDED as a plain string (graph_planar.py:108)

    def __init__(self, node_ids, dim):
        self.node_ids = list(node_ids)
        self._idx = {n: i for i, n in enumerate(self.node_ids)}
        N = len(self.node_ids)
        self.dim = dim
        self.edge = np.zeros((N, N), dtype=bool)
        self.dist = np.full((N, N), np.nan)
        self.lower = np.full((N, N), np.nan)
        self.upper = np.full((N, N), np.nan)
        self.bounded = np.full((N, N), B_NOEDGE, dtype=np.int8)
        self.nodes = _Nodes({n: {} for n in self.node_ids})

    # -- container protocol ----------------------------------------------------------------------
    def index(self, name):
        return self._idx[name]

    def number_of_nodes(self):
        return len(self.node_ids)

    def number_of_edges(self):
        return int(np.count_nonzero(np.triu(self.edge)))

    def __len__(self):
        return len(self.node_ids)

    def __iter__(self):
        return iter(self.node_ids)

    def __contains__(self, name):
        return name in self._idx

    def __getitem__(self, u):
        return _Adj(self, self.index(u))

    def _bounded_value(self, code):
        if self.planar_bounded and code == B_BELOW:
            return BELOW
        return {B_EMPTY: [], B_FALSE: [False], B_BELOW: [BELOW], B_ABOVE: [ABOVE],
                B_NONE: [None]}[int(code)]

    def edges(self, data=False):
        N = len(self.node_ids)
        for i in range(N):
            for j in range(i + 1, N):
                if self.edge[i, j]:
                    u, v = self.node_ids[i], self.node_ids[j]
                    yield (u, v, self[u][v]) if data else (u, v)

    def set_edge(self, u, v, dist=None, lower=None, upper=None, bounded=None):
        i, j = self.index(u), self.index(v)
        if not self.edge[i, j]:
            self.edge[i, j] = self.edge[j, i] = True
            self.bounded[i, j] = self.bounded[j, i] = B_ABSENT
        for M, val in ((self.dist, dist), (self.lower, lower), (self.upper, upper)):
            if val is not None:
                M[i, j] = M[j, i] = val
        if bounded is not None:
            self.bounded[i, j] = self.bounded[j, i] = bounded

    def _copy_into(self, G):
        for name in ("edge", "dist", "lower", "upper", "bounded"):
            setattr(G, name, getattr(self, name).copy())
        G.nodes = _Nodes({n: dict(a) for n, a in self.nodes.items()})
        return G

    def subgraph(self, names):
        """The graph induced by `names` (in this graph's node order), as a copy."""
        keep = [n for n in self.node_ids if n in set(names)]
        idx = np.array([self._idx[n] for n in keep], dtype=int)
        G = DistanceGraph(keep, self.dim)
        G.planar_bounded = self.planar_bounded
        for name in ("edge", "dist", "lower", "upper", "bounded"):
            setattr(G, name, getattr(self, name)[np.ix_(idx, idx)].copy())
        G.nodes = _Nodes({n: dict(self.nodes[n]) for n in keep})
        return G

    def positions(self):
        """N x dim array of POS (NaN rows where unknown)."""
        P = np.full((len(self.node_ids), self.dim), np.nan)
        for i, n in enumerate(self.node_ids):
            if POS in self.nodes[n]:
                P[i] = self.nodes[n][POS]
        return P

    def complete_edges(self, overwrite=False):
        """graph_complete_edges (graphik/utils/dgp.py:124-147): every pair of nodes that both
        carry POS and share no DIST edge gets DIST = LOWER = UPPER = their distance."""
        have = [i for i, n in enumerate(self.node_ids) if POS in self.nodes[n]]
        for a, i in enumerate(have):
            for j in have[a + 1:]:
                if overwrite or np.isnan(self.dist[i, j]):
                    d = np.linalg.norm(np.asarray(self.nodes[self.node_ids[i]][POS], dtype=float)
                                       - np.asarray(self.nodes[self.node_ids[j]][POS], dtype=float))
                    self.set_edge(self.node_ids[i], self.node_ids[j], dist=d, lower=d, upper=d)
        return self


class ProblemGraph(DistanceGraph):
    """graphik/graphs/graph_base.py:19-279 (the parts the Riemannian path uses)."""

    def __init__(self, robot, params, node_ids):
        super().__init__(node_ids, robot.dim)
        self.robot = robot
        self.axis_length = params.get("axis_length", 1)

    # node classes (graph_base.py:31-69)
    def _nodes_of(self, tag):
        return [n for n in self.node_ids if tag in self.nodes[n].get(TYPE, [])]

    @property
    def base_nodes(self):
        return self._nodes_of(BASE)

    @property
    def structure_nodes(self):
        return self._nodes_of(ROBOT)

    @property
    def end_effector_nodes(self):
        """graph_base.py:57-68: the nodes tagged END_EFFECTOR (only ProblemGraphPlanar tags any), cached."""
        if not hasattr(self, "_end_effector_nodes"):
            self._end_effector_nodes = self._nodes_of(END_EFFECTOR)
        return self._end_effector_nodes

    @property
    def base(self):
        """graph_base.py:70-75: the base coordinate system's subgraph (a copy; edges are undirected here)."""
        return self.subgraph(self.base_nodes)

    @property
    def structure(self):
        """graph_base.py:77-82: the robot structure's subgraph."""
        return self.subgraph(self.structure_nodes)

    def _instance(self):
        G = DistanceGraph(self.node_ids, self.dim)
        G.planar_bounded = self.planar_bounded
        return self._copy_into(G)

    def from_pos(self, P, dist=True, overwrite=False):
        """graph_base.py:146-165"""
        G = self._instance()
        for name, pos in P.items():
            if name in G:
                G.nodes[name][POS] = np.asarray(pos, dtype=float)
        if dist:
            G.complete_edges(overwrite=overwrite)
        return G

    def _pose_goal(self, T_goal):
        raise NotImplementedError

    def from_pose(self, T_goal):
        """graph_base.py:171-180 (a bare pose means the first end effector)."""
        if not isinstance(T_goal, dict):
            T_goal = {self.robot.end_effectors[0]: T_goal}
        return self.from_pos(self._pose_goal(T_goal))

    def realization(self, joint_angles):
        """graph_base.py:112-120"""
        return self.from_pos(self._pose_goal(self.robot.get_all_poses(joint_angles)))

    def distance_matrix(self):
        from ..utils.dgp import distance_matrix_from_graph
        return distance_matrix_from_graph(self)

    def distance_matrix_from_joints(self, joint_angles):
        """graph_base.py:129-136: squared distances between all nodes at the given configuration."""
        from ..utils.dgp import distance_matrix_from_graph
        return distance_matrix_from_graph(self.realization(joint_angles))

    def adjacency_matrix(self):
        from ..utils.dgp import adjacency_matrix_from_graph
        return adjacency_matrix_from_graph(self)

    def add_anchor_node(self, name, data):
        """graph_base.py:182-199: a node with known position, tied to every node with POS."""
        if POS not in data:
            raise KeyError("Node needs to gave a position to be added.")
        old = self
        N = len(self.node_ids)
        self.node_ids.append(name)
        self._idx[name] = N
        for attr, fill in (("dist", np.nan), ("lower", np.nan), ("upper", np.nan)):
            M = np.full((N + 1, N + 1), fill)
            M[:N, :N] = getattr(old, attr)
            setattr(self, attr, M)
        E = np.zeros((N + 1, N + 1), dtype=bool)
        E[:N, :N] = self.edge
        self.edge = E
        Bd = np.full((N + 1, N + 1), B_NOEDGE, dtype=np.int8)
        Bd[:N, :N] = self.bounded
        self.bounded = Bd
        self.nodes[name] = dict(data)
        p = np.asarray(data[POS], dtype=float)
        for other in self.node_ids[:-1]:
            if POS in self.nodes[other]:
                d = np.linalg.norm(np.asarray(self.nodes[other][POS], dtype=float) - p)
                self.set_edge(other, name, dist=d, lower=d, upper=d, bounded=B_EMPTY)

    def add_spherical_obstacle(self, name, position, radius, intended=False):
        """graph_base.py:201-211.  The reference compares a node's TYPE (a list) with the string
        ROBOT, so it never creates robot<->obstacle lower-bound edges; that observable behaviour
        is the default here.  intended=True creates them."""
        self.add_anchor_node(name, {POS: np.asarray(position, dtype=float), TYPE: OBSTACLE,
                                    "radius": radius})
        if intended:
            for node in self.node_ids:
                if ROBOT in self.nodes[node].get(TYPE, []) and node[0] == MAIN_PREFIX \
                        and np.isnan(self.dist[self.index(node), self.index(name)]):
                    self.set_edge(node, name, lower=radius, upper=100, bounded=B_BELOW)

    def clear_obstacles(self):
        """graph_base.py:213-217"""
        keep = [i for i, n in enumerate(self.node_ids) if self.nodes[n].get(TYPE) != OBSTACLE]
        if len(keep) == len(self.node_ids):
            return
        ix = np.ix_(keep, keep)
        for attr in ("edge", "dist", "lower", "upper", "bounded"):
            setattr(self, attr, getattr(self, attr)[ix].copy())
        for n in [n for i, n in enumerate(self.node_ids) if i not in keep]:
            del self.nodes[n]
        self.node_ids = [self.node_ids[i] for i in keep]
        self._idx = {n: i for i, n in enumerate(self.node_ids)}

    def check_distance_limits(self, G, tol=1e-10, intended=False):
        """graph_base.py:219-260.  In the reference `typ[u] == ROBOT` compares a list with a
        string, so no violation is ever reported for graphs built by these classes; that is the
        default (returns []).  intended=True performs the check the code describes."""
        if not intended:
            return []
        broken = []
        for u, v, data in self.edges(data=True):
            b = data.get(BOUNDED, [])
            if BELOW in b or ABOVE in b:
                d = G[u][v][DIST] if v in G[u] and DIST in G[u][v] else np.linalg.norm(
                    np.asarray(G.nodes[u][POS]) - np.asarray(G.nodes[v][POS]))
                tu, tv = self.nodes[u].get(TYPE, []), self.nodes[v].get(TYPE, [])
                kind = OBSTACLE if OBSTACLE in (tu, tv) else "joint"
                if d < data[LOWER] - tol:
                    broken.append({"edge": (u, v), "value": d - data[LOWER], "type": kind,
                                   "side": LOWER})
                if d > data[UPPER] + tol:
                    broken.append({"edge": (u, v), "value": d - data[UPPER], "type": kind,
                                   "side": UPPER})
        return broken

    def distance_bound_matrices(self):
        """psi_L, psi_U (graph_base.py:262-279): squared LOWER where BOUNDED contains 'below',
        squared UPPER where it contains 'above'."""
        L = np.where(self.bounded == B_BELOW, self.lower ** 2, 0.0)
        U = np.where(self.bounded == B_ABOVE, self.upper ** 2, 0.0)
        return np.nan_to_num(L), np.nan_to_num(U)

    @staticmethod
    def _pose_matrix(T):
        return as_matrix(T)
