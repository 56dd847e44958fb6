"""Robot loaders (graphik/utils/roboturdf.py).

`load_schunk_lwa4d / load_ur10 / load_kuka / load_schunk_lwa4p / load_panda` return (robot, graph)
like the reference's loaders (roboturdf.py:299-371) from kinematic constants packaged under
graphik_amd/data/robots (frames at zero configuration extracted from the reference's URDF data; see
tools/export_robot_data.py, tools/capture_golden_loaders.py).
`RobotURDF` is an own kinematics-only URDF reader for user-supplied files.
"""
import json
import os
import xml.etree.ElementTree as ET

import numpy as np

from ..graphs import ProblemGraphRevolute
from ..robots import RobotRevolute
from ..utils.lie import SE3

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "robots")


def _packaged(name):
    with open(os.path.join(_DATA, name + ".json")) as f:
        rec = json.load(f)
    T = np.array([[[float.fromhex(v) for v in row] for row in M] for M in rec["T_zero"]])
    return rec["num_joints"], T


def _randomize_links(T, pct):
    """randomized_links of make_Revolute3d (roboturdf.py:236-244): the translation between
    consecutive frames is scaled by (1 - pct) + 2 pct U, one np.random.rand() per link, entries below
    1e-6 zeroed.  As in the reference the list is modified in place while it is read, so link idx is
    measured from the ALREADY MODIFIED frame idx to the original frame idx + 1.  (Applied to the
    frames re-based on p0: relative transforms do not change under the common left factor.)"""
    T = [np.array(M, dtype=float) for M in T]
    for idx in range(len(T) - 1):
        D = np.linalg.inv(T[idx]) @ T[idx + 1]
        t = D[:3, 3] * ((1 - pct) + 2 * pct * np.random.rand())
        t[np.abs(t) < 1e-6] = 0
        D[:3, 3] = t
        T[idx + 1] = T[idx] @ D
    return np.array(T)


def _make(n, T_zero, limits, randomized_links=False, randomize_percentage=0.4):
    if randomized_links:
        T_zero = _randomize_links(T_zero, randomize_percentage)
    if limits is None:  # roboturdf.py:318-320
        ub = np.ones(n) * np.pi
        lb = -ub
    else:
        lb, ub = limits[0], limits[1]
    params = {"T_zero": {f"p{i}": T_zero[i] for i in range(n + 1)}, "num_joints": n,
              "joint_limits_upper": list(ub), "joint_limits_lower": list(lb),
              "parents": {f"p{i}": ([f"p{i + 1}"] if i < n else []) for i in range(n + 1)}}
    robot = RobotRevolute(params)
    return robot, ProblemGraphRevolute(robot)


def load_schunk_lwa4d(limits=None, randomized_links=False, randomize_percentage=0.4):
    return _make(*_packaged("lwa4d"), limits, randomized_links, randomize_percentage)


def load_ur10(limits=None, randomized_links=False, randomize_percentage=0.4):
    return _make(*_packaged("ur10"), limits, randomized_links, randomize_percentage)


def load_kuka(limits=None, randomized_links=False, randomize_percentage=0.4):
    return _make(*_packaged("kuka"), limits, randomized_links, randomize_percentage)


def load_schunk_lwa4p(limits=None, randomized_links=False, randomize_percentage=0.4):
    """Schunk LWA4P, 6 joints (roboturdf.py:299-312)."""
    return _make(*_packaged("lwa4p"), limits, randomized_links, randomize_percentage)


def load_panda(limits=None, randomized_links=False, randomize_percentage=0.4):
    """Franka Panda arm, 7 joints (roboturdf.py:343-356)."""
    return _make(*_packaged("panda"), limits, randomized_links, randomize_percentage)


def load_truncated_ur10(n):
    """First n links of a UR10 from DH parameters (roboturdf.py:374-402)."""
    a = [0, -0.612, -0.5723, 0, 0, 0][:n]
    d = [0.1273, 0, 0, 0.1639, 0.1157, 0.0922][:n]
    al = [np.pi / 2, 0, 0, np.pi / 2, -np.pi / 2, 0][:n]
    params = {"a": a, "alpha": al, "d": d, "theta": [0] * n, "modified_dh": False,
              "num_joints": n}
    robot = RobotRevolute(params)
    return robot, ProblemGraphRevolute(robot)


# ---------------------------------------------------------------------------------------------
def _rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return (np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]]) @
            np.array([[cp, 0, sp], [0, 1.0, 0], [-sp, 0, cp]]) @
            np.array([[1.0, 0, 0], [0, cr, -sr], [0, sr, cr]]))


def _axis_frame(axis):
    """Rotation taking the joint axis onto +z (get_T_from_joint_axis, roboturdf.py:266-297)."""
    z = np.array([0.0, 0.0, 1.0])
    axis = np.asarray(axis, dtype=float)
    if np.all(np.isclose(axis, -z)):
        return np.diag([1.0, -1.0, -1.0])
    if np.all(np.isclose(axis, z)):
        return np.identity(3)
    k = np.cross(axis, z)
    ang = -np.arcsin(np.linalg.norm(k) / np.linalg.norm(axis))
    k = k / np.linalg.norm(k)
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) * np.cos(ang) + (1 - np.cos(ang)) * np.outer(k, k) - np.sin(ang) * K


class RobotURDF:
    """Kinematics-only URDF reader for serial chains (RobotURDF, roboturdf.py:11-264)."""

    def __init__(self, fname):
        self.fname = fname
        root = ET.parse(fname).getroot()
        joints = []
        for e in root.findall("joint"):
            o = e.find("origin")
            xyz = [float(t) for t in (o.get("xyz", "0 0 0") if o is not None else "0 0 0").split()]
            rpy = [float(t) for t in (o.get("rpy", "0 0 0") if o is not None else "0 0 0").split()]
            M = np.identity(4)
            M[:3, :3] = _rpy(*rpy)
            M[:3, 3] = xyz
            a = e.find("axis")
            axis = np.array([float(t) for t in (a.get("xyz") if a is not None else "1 0 0").split()])
            lim = e.find("limit")
            joints.append(dict(name=e.get("name"), type=e.get("type"),
                               parent=e.find("parent").get("link"),
                               child=e.find("child").get("link"), origin=M,
                               axis=axis / np.linalg.norm(axis),
                               lower=float(lim.get("lower", 0)) if lim is not None else 0.0,
                               upper=float(lim.get("upper", 0)) if lim is not None else 0.0))
        self.joints = joints
        children = {j["child"] for j in joints}
        link = [l.get("name") for l in root.findall("link") if l.get("name") not in children][0]
        # walk the chain from the base link, accumulating link frames at zero configuration
        by_parent = {}
        for j in joints:
            by_parent.setdefault(j["parent"], []).append(j)
        T = np.identity(4)
        chain = []
        while link in by_parent:
            if len(by_parent[link]) != 1:
                raise NotImplementedError("only serial chains are supported")
            j = by_parent[link][0]
            T = T @ j["origin"]
            chain.append((j, T.copy()))
            link = j["child"]
        self.actuated = [(j, T) for j, T in chain if j["type"] != "fixed"]
        self.n_q_joints = len(self.actuated)
        frames = []
        for j, Tl in self.actuated:  # z along the joint axis (roboturdf.py:122-147)
            A = np.identity(4)
            A[:3, :3] = _axis_frame(j["axis"])
            frames.append(Tl @ np.linalg.inv(A))
        # end-effector joints = joints without actuated descendants: the last actuated joint and
        # everything after it; their frame is the child-link frame itself (:149-176)
        last = max(i for i, (j, _) in enumerate(chain) if j["type"] != "fixed")
        frames[-1] = chain[last][1]
        tail = [T for _, T in chain[last + 1:]]
        self.T_zero_list = frames + tail

    def joint_limits(self):
        ub = {f"p{i + 1}": float(np.clip(j["upper"], -np.pi, np.pi)) for i, (j, _) in
              enumerate(self.actuated)}
        lb = {f"p{i + 1}": float(np.clip(j["lower"], -np.pi, np.pi)) for i, (j, _) in
              enumerate(self.actuated)}
        return ub, lb

    def make_Revolute3d(self, ub, lb, randomized_links=False, randomize_percentage=0.4):
        """Frames labelled p0..pn and re-based on p0 (roboturdf.py:226-264)."""
        T = self.T_zero_list
        if randomized_links:
            T = list(_randomize_links(T, randomize_percentage))
        if len(T) != self.n_q_joints + 1:
            raise NotImplementedError("expected exactly one fixed end-effector joint after the chain")
        T0inv = np.linalg.inv(T[0])
        T_zero = {f"p{i}": SE3.from_matrix(T0inv @ T[i]) for i in range(len(T))}
        return RobotRevolute({"T_zero": T_zero, "num_joints": self.n_q_joints,
                              "joint_limits_upper": list(ub), "joint_limits_lower": list(lb)})
