"""Planar revolute chains and trees (graphik/robots/robot_planar.py)."""
import numpy as np

from .robot_base import Robot
from ..utils.constants import ROOT
from ..utils.lie import SE2, SO2, as_matrix


class RobotPlanar(Robot):
    def __init__(self, params):
        super().__init__(params)
        self.dim = 2
        if "T_zero" in params:
            T_zero = {k: SE2.from_matrix(as_matrix(v)) for k, v in params["T_zero"].items()}
        else:
            try:
                T_zero = self._from_params()
            except KeyError:
                raise Exception("Robot description not provided.")
        for name in self.joint_ids:
            self.nodes[name]["T0"] = T_zero[name]
        for name in self.joint_ids:  # robot_planar.py:40-42
            M = self.nodes[name]["T0"].as_matrix()
            q = np.array([M[0, 2], M[1, 2], 0.0])
            w = np.array([0.0, 0.0, 1.0])
            self.nodes[name]["S"] = np.hstack((np.cross(-w, q), w))[[0, 1, 5]]

    def _from_params(self):
        """Zero-configuration frames from the link lengths, accumulated along the path from the root
        to every joint -- chains and trees (robot_planar.py:51-60, kinematics.py:21-35,
        geometry.py:8-17).  As in the reference, the zero configuration stands in for the link
        offsets: fk_tree_2d is called with theta = q = zero_configuration(), so params["theta"] is
        not read."""
        self.l = self.params["link_lengths"]
        T = {ROOT: SE2.identity()}
        for node in self.joint_ids:          # a parent precedes its children in joint_ids
            if node == ROOT:
                continue
            T[node] = T[self.parent[node]].dot(SE2(SO2.identity(), np.array([self.l[node], 0.0])))
        return T

    def from_params(self):
        """robot_planar.py:51-60 (the reference's public name): zero-configuration frames {joint: SE2}."""
        return self._from_params()

    def pose(self, joint_angles, query_node):
        path = self.kinematic_map[ROOT][query_node]
        T = self.nodes[ROOT]["T0"]
        for pred, cur in zip(path[:-1], path[1:]):
            T = T.dot(SE2.exp(self.nodes[pred]["S"] * joint_angles[cur]))
        return T.dot(self.nodes[query_node]["T0"])

    def T0_array(self):
        return np.stack([self.nodes[f"p{i}"]["T0"].as_matrix() for i in range(self.n + 1)])

    def fk_batch(self, Q, node_index=None):
        """Q [B,n] -> T [B,3,3]."""
        Q = np.atleast_2d(np.asarray(Q, dtype=float))
        m = self.n if node_index is None else int(node_index)
        out = np.empty((Q.shape[0], 3, 3))
        for b in range(Q.shape[0]):
            out[b] = self.pose({f"p{i + 1}": Q[b, i] for i in range(self.n)}, f"p{m}").as_matrix()
        return out
