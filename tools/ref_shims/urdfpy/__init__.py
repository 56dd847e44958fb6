"""Stand-in for urdfpy: a kinematics-only URDF reader (links, joints, actuated_joints, link_fk)."""
import xml.etree.ElementTree as ET

import numpy as np


def _floats(text, default):
    if text is None:
        return np.array(default, dtype=float)
    return np.array([float(t) for t in text.split()], dtype=float)


def _rpy_to_matrix(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]], dtype=float)
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]], dtype=float)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]], dtype=float)
    return Rz.dot(Ry).dot(Rx)


def _axis_angle(axis, angle):
    a = axis / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.identity(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K.dot(K)
    T = np.identity(4)
    T[:3, :3] = R
    return T


class Link:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "Link(%s)" % self.name


class JointLimit:
    def __init__(self, lower, upper):
        self.lower = lower
        self.upper = upper


class Joint:
    def __init__(self, name, joint_type, parent, child, origin, axis, limit):
        self.name = name
        self.joint_type = joint_type
        self.parent = parent
        self.child = child
        self.origin = origin
        self.axis = axis
        self.limit = limit

    def get_child_pose(self, cfg=None):
        if cfg is None or self.joint_type == "fixed":
            return self.origin
        if self.joint_type in ("revolute", "continuous"):
            return self.origin.dot(_axis_angle(self.axis, float(cfg)))
        if self.joint_type == "prismatic":
            T = np.identity(4)
            T[:3, 3] = self.axis * float(cfg)
            return self.origin.dot(T)
        raise NotImplementedError(self.joint_type)

    def __repr__(self):
        return "Joint(%s)" % self.name


class URDF:
    def __init__(self, links, joints):
        self.links = links
        self.joints = joints
        self._link_map = {l.name: l for l in links}
        self._parent_joint = {j.child: j for j in joints}
        children = set(j.child for j in joints)
        self.base_link = [l for l in links if l.name not in children][0]
        act = [j for j in joints if j.joint_type != "fixed"]
        depth = [len(self._path_to_base(self._link_map[j.child])) for j in act]
        self.actuated_joints = [act[i] for i in np.argsort(depth, kind="stable")]

    def _path_to_base(self, link):
        path = [link]
        while path[-1].name in self._parent_joint:
            path.append(self._link_map[self._parent_joint[path[-1].name].parent])
        return path

    @staticmethod
    def load(fname):
        root = ET.parse(fname).getroot()
        links = [Link(e.get("name")) for e in root.findall("link")]
        joints = []
        for e in root.findall("joint"):
            o = e.find("origin")
            xyz = _floats(o.get("xyz") if o is not None else None, [0, 0, 0])
            rpy = _floats(o.get("rpy") if o is not None else None, [0, 0, 0])
            origin = np.identity(4)
            origin[:3, :3] = _rpy_to_matrix(rpy)
            origin[:3, 3] = xyz
            a = e.find("axis")
            axis = _floats(a.get("xyz") if a is not None else None, [1, 0, 0])
            axis = axis / np.linalg.norm(axis)
            lim = e.find("limit")
            limit = None
            if lim is not None:
                limit = JointLimit(float(lim.get("lower", 0.0)), float(lim.get("upper", 0.0)))
            joints.append(Joint(e.get("name"), e.get("type"), e.find("parent").get("link"),
                                e.find("child").get("link"), origin, axis, limit))
        return URDF(links, joints)

    def link_fk(self, cfg=None):
        cfg = cfg or {}
        fk = {}
        order = sorted(self.links, key=lambda l: len(self._path_to_base(l)))
        for lnk in order:
            if lnk.name not in self._parent_joint:
                fk[lnk] = np.identity(4)
                continue
            j = self._parent_joint[lnk.name]
            pose = j.get_child_pose(cfg.get(j.name, None)).dot(np.identity(4))
            fk[lnk] = fk[self._link_map[j.parent]].dot(pose)
        return fk
