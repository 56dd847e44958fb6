"""Minimal SE(2)/SE(3) pose types with the liegroups-style surface GraphIK callers use
(.rot, .trans, .dot, .inv, .as_matrix, exp, log, from_matrix, identity).  Any object exposing
`.as_matrix()` (e.g. a liegroups SE3Matrix) or a raw (dim+1)x(dim+1) array is accepted wherever
a pose is expected -- see `as_matrix()`."""
import numpy as np

_SMALL = 1e-12


def as_matrix(T):
    """Homogeneous matrix of a pose given as SE2/SE3 (ours or liegroups') or ndarray."""
    if hasattr(T, "as_matrix"):
        return np.asarray(T.as_matrix(), dtype=float)
    return np.asarray(T, dtype=float)


def hat3(w):
    w = np.asarray(w, dtype=float).ravel()
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def rodrigues(phi):
    """exp of a rotation vector and the matching left Jacobian (both 3x3)."""
    phi = np.asarray(phi, dtype=float).ravel()
    th = np.linalg.norm(phi)
    if th < _SMALL:
        W = hat3(phi)
        return np.identity(3) + W, np.identity(3) + 0.5 * W
    a = phi / th
    s, c = np.sin(th), np.cos(th)
    aa, A = np.outer(a, a), hat3(a)
    R = c * np.identity(3) + (1 - c) * aa + s * A
    J = (s / th) * np.identity(3) + (1 - s / th) * aa + ((1 - c) / th) * A
    return R, J


def so3_log(R):
    c = np.clip(0.5 * np.trace(R) - 0.5, -1.0, 1.0)
    th = np.arccos(c)
    if np.isclose(th, 0.0):
        M = R - np.identity(3)
    else:
        M = (0.5 * th / np.sin(th)) * (R - R.T)
    return np.array([M[2, 1], M[0, 2], M[1, 0]])


class _Rot:
    def __init__(self, mat):
        self.mat = np.asarray(mat, dtype=float)

    def as_matrix(self):
        return self.mat

    def inv(self):
        return type(self)(self.mat.T.copy())

    def dot(self, other):
        if isinstance(other, _Rot):
            return type(self)(self.mat @ other.mat)
        return self.mat @ np.asarray(other, dtype=float)


class SO3(_Rot):
    @classmethod
    def identity(cls):
        return cls(np.identity(3))

    @classmethod
    def rotz(cls, a):
        c, s = np.cos(a), np.sin(a)
        return cls(np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]]))

    @classmethod
    def roty(cls, a):
        c, s = np.cos(a), np.sin(a)
        return cls(np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]]))

    @classmethod
    def rotx(cls, a):
        c, s = np.cos(a), np.sin(a)
        return cls(np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]]))

    def log(self):
        return so3_log(self.mat)


class SO2(_Rot):
    @classmethod
    def identity(cls):
        return cls(np.identity(2))

    @classmethod
    def from_angle(cls, a):
        c, s = np.cos(a), np.sin(a)
        return cls(np.array([[c, -s], [s, c]]))

    def to_angle(self):
        return float(np.arctan2(self.mat[1, 0], self.mat[0, 0]))

    def log(self):
        return self.to_angle()


class _Pose:
    Rot = None
    d = None

    def __init__(self, rot, trans):
        self.rot = rot if isinstance(rot, _Rot) else self.Rot(rot)
        self.trans = np.asarray(trans, dtype=float)

    @classmethod
    def identity(cls):
        return cls(cls.Rot(np.identity(cls.d)), np.zeros(cls.d))

    @classmethod
    def from_matrix(cls, M, normalize=False):
        M = as_matrix(M)
        return cls(cls.Rot(M[: cls.d, : cls.d].copy()), M[: cls.d, cls.d].copy())

    def as_matrix(self):
        M = np.identity(self.d + 1)
        M[: self.d, : self.d] = self.rot.mat
        M[: self.d, self.d] = self.trans
        return M

    def inv(self):
        Rt = self.rot.mat.T
        return type(self)(self.Rot(Rt.copy()), -(Rt @ self.trans))

    def dot(self, other):
        if hasattr(other, "as_matrix") and not isinstance(other, _Rot):
            M = as_matrix(other)
            return type(self)(self.Rot(self.rot.mat @ M[: self.d, : self.d]),
                              self.rot.mat @ M[: self.d, self.d] + self.trans)
        v = np.asarray(other, dtype=float)
        if v.shape[-1] == self.d:
            return (self.rot.mat @ v.T).T + self.trans
        return self.as_matrix() @ v

    def __repr__(self):
        return f"<{type(self).__name__}>\n{self.as_matrix()}"


class SE3(_Pose):
    Rot = SO3
    d = 3

    @classmethod
    def exp(cls, xi):
        xi = np.asarray(xi, dtype=float).ravel()
        R, J = rodrigues(xi[3:6])
        return cls(SO3(R), J @ xi[0:3])

    def log(self):
        phi = so3_log(self.rot.mat)
        _, J = rodrigues(phi)
        return np.hstack([np.linalg.solve(J, self.trans), phi])


class SE2(_Pose):
    Rot = SO2
    d = 2

    @classmethod
    def exp(cls, xi):
        xi = np.asarray(xi, dtype=float).ravel()
        phi = xi[2]
        if abs(phi) < _SMALL:
            J = np.array([[1.0, -0.5 * phi], [0.5 * phi, 1.0]])
        else:
            s, c = np.sin(phi), np.cos(phi)
            J = np.array([[s / phi, -(1 - c) / phi], [(1 - c) / phi, s / phi]])
        return cls(SO2.from_angle(phi), J @ xi[0:2])

    def log(self):
        phi = self.rot.to_angle()
        if abs(phi) < _SMALL:
            J = np.array([[1.0, -0.5 * phi], [0.5 * phi, 1.0]])
        else:
            s, c = np.sin(phi), np.cos(phi)
            J = np.array([[s / phi, -(1 - c) / phi], [(1 - c) / phi, s / phi]])
        return np.hstack([np.linalg.solve(J, self.trans), phi])


def trans_axis(t, axis="z"):
    """Pure translation along a coordinate axis (graphik/utils/geometry.py:27-34)."""
    v = np.zeros(3)
    v["xyz".index(axis)] = t
    return SE3(SO3.identity(), v)


def rot_axis(theta, axis="z"):
    """Pure rotation about a coordinate axis (graphik/utils/geometry.py:37-44)."""
    R = {"x": SO3.rotx, "y": SO3.roty, "z": SO3.rotz}[axis](theta)
    return SE3(R, np.zeros(3))
