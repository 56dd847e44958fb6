"""Kinematic-chain robot description (host side).

Mirrors the surface of graphik/robots/robot_base.py that the Riemannian-solver path and its
callers use: n, dim, lb/ub dicts, joint_ids, end_effectors, kinematic_map, nodes[...]["T0"],
random_configuration(), zero_configuration(), pose(), get_all_poses().  Chains only (every
BASELINE config is a serial arm); array-backed instead of a networkx.DiGraph.
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
        if "parents" in params:
            for k, ch in params["parents"].items():
                if len(ch) > 1:
                    raise NotImplementedError("tree-structured robots are outside the hot path")
        self.joint_ids = [f"p{i}" for i in range(self.n + 1)]
        self.nodes = _NodeView({name: {} for name in self.joint_ids})
        # shortest paths between joints of a chain (robot_base.py:41)
        self.kinematic_map = {
            a: {b: self.joint_ids[i:j + 1] for j, b in enumerate(self.joint_ids) if j >= i}
            for i, a in enumerate(self.joint_ids)}
        lb = params.get("joint_limits_lower", self.n * [-np.pi])
        ub = params.get("joint_limits_upper", self.n * [np.pi])
        self.lb = lb if isinstance(lb, dict) else list_to_variable_dict(flatten([list(lb)]))
        self.ub = ub if isinstance(ub, dict) else list_to_variable_dict(flatten([list(ub)]))

    @property
    def end_effectors(self):
        return [self.joint_ids[-1]]

    @property
    def T_base(self):
        return self.nodes[ROOT]["T0"]

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
        for node in self.joint_ids[1:]:
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
