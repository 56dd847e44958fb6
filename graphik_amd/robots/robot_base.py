"""Kinematic-chain robot description (host side).

Mirrors the surface of graphik/robots/robot_base.py that the Riemannian-solver path and its
callers use: n, dim, lb/ub dicts, joint_ids, end_effectors, kinematic_map, nodes[...]["T0"],
random_configuration(), zero_configuration(), pose(), get_all_poses().  Chains and trees (several
end effectors, params["parents"]); array-backed instead of a networkx.DiGraph.
"""
import numpy as np

from ..utils.constants import ROOT
from ..utils.utils import list_to_variable_dict, flatten


class _NodeView(dict):
    """robot.nodes[name] -> attribute dict, like networkx' NodeView for the keys we keep."""


class Robot:
    def __init__(self, params):
        self.params = params
        self.n = int(params["num_joints"])
        # Topology (robot_base.py:29-41): `parents` maps a joint to its children; a chain if absent.
        # Node order = insertion order of networkx.DiGraph(dict of lists): every key followed by its
        # children, first appearance counts.
        if "parents" in params:
            order, children = [], {}
            for u, kids in params["parents"].items():
                if u not in order:
                    order.append(u)
                for v in kids:
                    if v not in order:
                        order.append(v)
                children.setdefault(u, []).extend(kids)
        else:
            order = [f"p{i}" for i in range(self.n + 1)]
            children = {order[i]: [order[i + 1]] for i in range(self.n)}
        self.joint_ids = order
        self.children = {u: list(children.get(u, [])) for u in order}
        self.parent = {v: u for u, kids in self.children.items() for v in kids}
        if sorted(order, key=lambda s: int(s[1:])) != [f"p{i}" for i in range(self.n + 1)] or ROOT in self.parent \
                or len(self.parent) != self.n:
            raise ValueError("joints must be named p0..pn and form a tree rooted at p0")
        self.is_chain = all(len(k) <= 1 for k in self.children.values())
        self.nodes = _NodeView({name: {} for name in self.joint_ids})
        # shortest paths between joints (robot_base.py:41): kinematic_map[a][b] for every b below a
        self.kinematic_map = {a: self._paths_from(a) for a in self.joint_ids}
        lb = params.get("joint_limits_lower", self.n * [-np.pi])
        ub = params.get("joint_limits_upper", self.n * [np.pi])
        self.lb = lb if isinstance(lb, dict) else list_to_variable_dict(flatten([list(lb)]))
        self.ub = ub if isinstance(ub, dict) else list_to_variable_dict(flatten([list(ub)]))

    def _paths_from(self, a):
        paths, todo = {a: [a]}, [a]
        while todo:                      # breadth first, children in insertion order
            u = todo.pop(0)
            for v in self.children[u]:
                paths[v] = paths[u] + [v]
                todo.append(v)
        return paths

    @property
    def end_effectors(self):
        """Leaves of the tree in node order (robot_base.py:100-106)."""
        return [j for j in self.joint_ids if not self.children[j]]

    @property
    def T_base(self):
        return self.nodes[ROOT]["T0"]

    @property
    def limited_joints(self):
        """robot_base.py:141-150: joints whose limits a graph could express (set by the graph's set_limits)."""
        return getattr(self, "_limited_joints", [])

    @limited_joints.setter
    def limited_joints(self, lim):
        self._limited_joints = list(lim)

    @property
    def spherical(self):
        return False

    def random_configuration(self):
        """One np.random.rand() per joint in p1..pn order (robot_base.py:76-85)."""
        q = {}
        for key in self.joint_ids:
            if key != ROOT:
                q[key] = self.lb[key] + (self.ub[key] - self.lb[key]) * np.random.rand()
        return q

    def zero_configuration(self):
        return {key: 0 for key in self.joint_ids if key != ROOT}

    def get_all_poses(self, joint_angles):
        """robot_base.py:185-193"""
        T = {ROOT: self.T_base}
        for ee in self.end_effectors:
            for node in self.kinematic_map[ROOT][ee][1:]:
                T[node] = self.pose(joint_angles, node)
        return T

    # -- array views used by the batched engine ------------------------------------------------
    def limits_arrays(self):
        lb = np.array([self.lb[f"p{i}"] for i in range(1, self.n + 1)], dtype=float)
        ub = np.array([self.ub[f"p{i}"] for i in range(1, self.n + 1)], dtype=float)
        return lb, ub

    def q_to_array(self, q):
        return np.array([q[f"p{i}"] for i in range(1, self.n + 1)], dtype=float)

    def array_to_q(self, a):
        return {f"p{i + 1}": float(a[i]) for i in range(self.n)}
