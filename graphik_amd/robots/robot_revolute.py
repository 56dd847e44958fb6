"""3-D revolute chain (graphik/robots/robot_revolute.py): frames at zero configuration `T0`,
screw axes `S`, forward kinematics by product of exponentials."""
import numpy as np

from .robot_base import Robot
from ..utils.constants import ROOT
from ..utils.lie import SE3, as_matrix, rot_axis, trans_axis
from ..utils.utils import list_to_variable_dict, flatten


def _dh(a, alpha, d, theta, modified):
    """One (modified) DH link (graphik/utils/kinematics.py:43-84)."""
    TX, RX, TZ, RZ = trans_axis(a, "x"), rot_axis(alpha, "x"), trans_axis(d, "z"), rot_axis(theta, "z")
    if modified:
        return TX.dot(RX.dot(TZ.dot(RZ)))
    return TZ.dot(RZ.dot(TX.dot(RX)))


class RobotRevolute(Robot):
    def __init__(self, params):
        super().__init__(params)
        self.dim = 3
        if "T_zero" in params:
            T_zero = {k: SE3.from_matrix(as_matrix(v)) for k, v in params["T_zero"].items()}
        elif all(k in params for k in ("a", "d", "alpha", "theta", "modified_dh")):
            T_zero = self._from_dh(params)
        else:
            raise Exception("Robot description not provided.")
        for name in self.joint_ids:
            self.nodes[name]["T0"] = T_zero[name]
        # twists of the joint axes (robot_revolute.py:33-44): S = [-w x q ; w]
        for name in self.joint_ids:
            M = self.nodes[name]["T0"].as_matrix()
            w, q = M[:3, 2], M[:3, 3]
            self.nodes[name]["S"] = np.hstack((np.cross(-w, q), w))

    def _from_dh(self, params):
        as_dict = lambda v: v if isinstance(v, dict) else list_to_variable_dict(flatten([list(v)]))
        a, d, al, th = (as_dict(params[k]) for k in ("a", "d", "alpha", "theta"))
        T = {ROOT: SE3.identity()}
        for node in self.joint_ids:            # robot_revolute.py:66-83: links along the path root -> node
            if node == ROOT:
                continue
            acc = None
            for link_node in self.kinematic_map[ROOT][node][1:][::-1]:   # right-to-left product
                link = _dh(a[link_node], al[link_node], d[link_node], th[link_node], params["modified_dh"])
                acc = link if acc is None else link.dot(acc)
            T[node] = acc
        return T

    def from_dh_params(self, params):
        """robot_revolute.py:53-83 (the reference's public name): zero-configuration frames {joint: SE3}."""
        return self._from_dh(params)

    def pose(self, joint_angles, query_node):
        """T0[root] * prod exp(S_pred * q_cur) * T0[node]  (robot_revolute.py:85-103)."""
        path = self.kinematic_map[ROOT][query_node]
        T = self.nodes[ROOT]["T0"]
        for pred, cur in zip(path[:-1], path[1:]):
            T = T.dot(SE3.exp(self.nodes[pred]["S"] * joint_angles[cur]))
        return T.dot(self.nodes[query_node]["T0"])

    # -- batched FK for goal generation / pose-error metrics ------------------------------------
    def T0_array(self):
        return np.stack([self.nodes[f"p{i}"]["T0"].as_matrix() for i in range(self.n + 1)])

    def fk_batch(self, Q, node_index=None):
        """Q [B,n] (columns p1..pn) -> T [B,4,4] of frame `node_index` (default: the first end
        effector), along its path from the root."""
        Q = np.atleast_2d(np.asarray(Q, dtype=float))
        node = self.end_effectors[0] if node_index is None else f"p{int(node_index)}"
        path = self.kinematic_map[ROOT][node]
        B = Q.shape[0]
        T = np.broadcast_to(self.nodes[ROOT]["T0"].as_matrix(), (B, 4, 4)).copy()
        for pred, cur in zip(path[:-1], path[1:]):
            S = self.nodes[pred]["S"]
            v, w = S[:3], S[3:]
            W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=float)
            th = Q[:, int(cur[1:]) - 1][:, None, None]
            s, c = np.sin(th), np.cos(th)
            I = np.identity(3)
            R = I + s * W + (1 - c) * (W @ W)                       # unit axis
            J = th * I + (1 - c) * W + (th - s) * (W @ W)           # theta * left Jacobian
            E = np.zeros((B, 4, 4))
            E[:, :3, :3] = R
            E[:, :3, 3] = (J @ v)
            E[:, 3, 3] = 1.0
            T = T @ E
        return T @ self.nodes[node]["T0"].as_matrix()
