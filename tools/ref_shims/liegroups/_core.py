"""Stand-in for the numpy backend of `liegroups` (SO2/SO3/SE2/SE3 matrix Lie groups).

Own restatement of the API surface the reference touches: constructors, identity,
from_matrix, exp/log (xi = [rho; phi], translation first), wedge, left_jacobian, inv, dot
(group*group and group*point), adjoint, as_matrix, .rot/.trans/.mat.
"""
import numpy as np

_EPS = 1e-12


def _is_group(x, cls):
    return isinstance(x, cls)


class SO2Matrix:
    dim = 2
    dof = 1

    def __init__(self, mat):
        self.mat = np.asarray(mat, dtype=float)

    @classmethod
    def identity(cls):
        return cls(np.identity(2))

    @classmethod
    def from_matrix(cls, mat, normalize=False):
        return cls(np.array(mat, dtype=float))

    @classmethod
    def from_angle(cls, angle):
        c, s = np.cos(angle), np.sin(angle)
        return cls(np.array([[c, -s], [s, c]]))

    @classmethod
    def exp(cls, phi):
        return cls.from_angle(phi)

    @staticmethod
    def wedge(phi):
        phi = float(np.squeeze(phi))
        return np.array([[0.0, -phi], [phi, 0.0]])

    @classmethod
    def left_jacobian(cls, phi):
        phi = float(np.squeeze(phi))
        if abs(phi) < _EPS:
            return np.identity(2) + 0.5 * cls.wedge(phi)
        s, c = np.sin(phi), np.cos(phi)
        return (s / phi) * np.identity(2) + ((1 - c) / phi) * cls.wedge(1.0)

    @classmethod
    def inv_left_jacobian(cls, phi):
        phi = float(np.squeeze(phi))
        if abs(phi) < _EPS:
            return np.identity(2) - 0.5 * cls.wedge(phi)
        half = 0.5 * phi
        cot = 1.0 / np.tan(half)
        return half * cot * np.identity(2) - half * cls.wedge(1.0)

    def to_angle(self):
        return np.arctan2(self.mat[1, 0], self.mat[0, 0])

    def log(self):
        return self.to_angle()

    def inv(self):
        return self.__class__(self.mat.T.copy())

    def as_matrix(self):
        return self.mat

    def dot(self, other):
        if _is_group(other, SO2Matrix):
            return self.__class__(self.mat.dot(other.mat))
        other = np.asarray(other, dtype=float)
        return np.squeeze(self.mat.dot(other.T).T) if other.ndim == 2 else self.mat.dot(other)

    def __repr__(self):
        return "<SO2Matrix>\n" + str(self.mat)


class SO3Matrix:
    dim = 3
    dof = 3

    def __init__(self, mat):
        self.mat = np.asarray(mat, dtype=float)

    @classmethod
    def identity(cls):
        return cls(np.identity(3))

    @classmethod
    def from_matrix(cls, mat, normalize=False):
        return cls(np.array(mat, dtype=float))

    @classmethod
    def rotx(cls, a):
        c, s = np.cos(a), np.sin(a)
        return cls(np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]]))

    @classmethod
    def roty(cls, a):
        c, s = np.cos(a), np.sin(a)
        return cls(np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]]))

    @classmethod
    def rotz(cls, a):
        c, s = np.cos(a), np.sin(a)
        return cls(np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]]))

    @staticmethod
    def wedge(phi):
        phi = np.asarray(phi, dtype=float).ravel()
        return np.array([[0.0, -phi[2], phi[1]], [phi[2], 0.0, -phi[0]], [-phi[1], phi[0], 0.0]])

    @staticmethod
    def vee(Phi):
        return np.array([Phi[2, 1], Phi[0, 2], Phi[1, 0]])

    @classmethod
    def exp(cls, phi):
        phi = np.asarray(phi, dtype=float).ravel()
        angle = np.linalg.norm(phi)
        if angle < _EPS:
            return cls(np.identity(3) + cls.wedge(phi))
        axis = phi / angle
        s, c = np.sin(angle), np.cos(angle)
        return cls(c * np.identity(3) + (1 - c) * np.outer(axis, axis) + s * cls.wedge(axis))

    def log(self):
        cos_angle = 0.5 * np.trace(self.mat) - 0.5
        cos_angle = np.clip(cos_angle, -1.0, 1.0)
        angle = np.arccos(cos_angle)
        if np.isclose(angle, 0.0):
            return self.vee(self.mat - np.identity(3))
        return self.vee((0.5 * angle / np.sin(angle)) * (self.mat - self.mat.T))

    @classmethod
    def left_jacobian(cls, phi):
        phi = np.asarray(phi, dtype=float).ravel()
        angle = np.linalg.norm(phi)
        if angle < _EPS:
            return np.identity(3) + 0.5 * cls.wedge(phi)
        axis = phi / angle
        s, c = np.sin(angle), np.cos(angle)
        return ((s / angle) * np.identity(3) + (1 - s / angle) * np.outer(axis, axis)
                + ((1 - c) / angle) * cls.wedge(axis))

    @classmethod
    def inv_left_jacobian(cls, phi):
        phi = np.asarray(phi, dtype=float).ravel()
        angle = np.linalg.norm(phi)
        if angle < _EPS:
            return np.identity(3) - 0.5 * cls.wedge(phi)
        axis = phi / angle
        half = 0.5 * angle
        cot = 1.0 / np.tan(half)
        return (half * cot * np.identity(3) + (1 - half * cot) * np.outer(axis, axis)
                - half * cls.wedge(axis))

    def inv(self):
        return self.__class__(self.mat.T.copy())

    def as_matrix(self):
        return self.mat

    def dot(self, other):
        if _is_group(other, SO3Matrix):
            return self.__class__(self.mat.dot(other.mat))
        other = np.asarray(other, dtype=float)
        return np.squeeze(self.mat.dot(other.T).T) if other.ndim == 2 else self.mat.dot(other)

    def __repr__(self):
        return "<SO3Matrix>\n" + str(self.mat)


class _SEBase:
    RotationType = None
    dim = None
    dof = None

    def __init__(self, rot, trans):
        self.rot = rot
        self.trans = np.asarray(trans, dtype=float)

    @classmethod
    def identity(cls):
        return cls(cls.RotationType.identity(), np.zeros(cls.dim - 1))

    @classmethod
    def from_matrix(cls, mat, normalize=False):
        mat = np.asarray(mat, dtype=float)
        d = cls.dim - 1
        return cls(cls.RotationType(mat[:d, :d].copy()), mat[:d, d].copy())

    def as_matrix(self):
        d = self.dim - 1
        M = np.identity(self.dim)
        M[:d, :d] = self.rot.as_matrix()
        M[:d, d] = self.trans
        return M

    def inv(self):
        inv_rot = self.rot.inv()
        return self.__class__(inv_rot, -(inv_rot.dot(self.trans)))

    def dot(self, other):
        d = self.dim - 1
        if isinstance(other, _SEBase):
            return self.__class__(self.rot.dot(other.rot), self.rot.dot(other.trans) + self.trans)
        other = np.asarray(other, dtype=float)
        if other.ndim == 1 and other.shape[0] == d:
            return self.rot.dot(other) + self.trans
        if other.ndim == 1 and other.shape[0] == d + 1:
            return self.as_matrix().dot(other)
        if other.ndim == 2 and other.shape[1] == d:
            return (self.rot.as_matrix().dot(other.T)).T + self.trans
        raise ValueError("unsupported operand for dot")

    def __repr__(self):
        return "<%s>\n%s" % (type(self).__name__, self.as_matrix())


class SE2Matrix(_SEBase):
    RotationType = SO2Matrix
    dim = 3
    dof = 3

    @classmethod
    def exp(cls, xi):
        xi = np.asarray(xi, dtype=float).ravel()
        rho, phi = xi[0:2], xi[2]
        return cls(SO2Matrix.exp(phi), SO2Matrix.left_jacobian(phi).dot(rho))

    def log(self):
        phi = self.rot.log()
        rho = SO2Matrix.inv_left_jacobian(phi).dot(self.trans)
        return np.hstack([rho, phi])

    def adjoint(self):
        rot_part = self.rot.as_matrix()
        trans_part = np.array([self.trans[1], -self.trans[0]]).reshape((2, 1))
        return np.vstack([np.hstack([rot_part, trans_part]), [0, 0, 1]])


class SE3Matrix(_SEBase):
    RotationType = SO3Matrix
    dim = 4
    dof = 6

    @classmethod
    def exp(cls, xi):
        xi = np.asarray(xi, dtype=float).ravel()
        rho, phi = xi[0:3], xi[3:6]
        return cls(SO3Matrix.exp(phi), SO3Matrix.left_jacobian(phi).dot(rho))

    def log(self):
        phi = self.rot.log()
        rho = SO3Matrix.inv_left_jacobian(phi).dot(self.trans)
        return np.hstack([rho, phi])

    def adjoint(self):
        R = self.rot.as_matrix()
        return np.vstack([np.hstack([R, SO3Matrix.wedge(self.trans).dot(R)]),
                          np.hstack([np.zeros((3, 3)), R])])
