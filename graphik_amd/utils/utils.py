"""Small helpers mirrored from graphik/utils/utils.py (only what the hot path's callers use)."""
import numpy as np
from numpy import pi


def wraptopi(e):
    """graphik/utils/utils.py:38-39"""
    return np.mod(e + pi, 2 * pi) - pi


def flatten(l):
    return [item for sub in l for item in sub]


def list_to_variable_dict(l, label="p", index_start=1):
    """graphik/utils/utils.py:46-52"""
    if isinstance(l, dict):
        return l
    return {label + str(index_start + i): v for i, v in enumerate(l)}


def variable_dict_to_list(d, order=None):
    return [d[k] for k in (order if order is not None else d)]


def normalize(v):
    n = np.linalg.norm(v)
    return v if n == 0 else v / n


def table_environment(height=0.9, width=0.8, n_height=9, n_width=8, obs_inflation=2.0):
    """Sphere-approximated table: list of (centre, radius) (graphik/utils/utils.py:179-191)."""
    radius = 0.5 * height / n_height
    half = n_width // 2
    rng = range(-half, half)
    top = [(np.asarray([2 * (i + 0.5) * radius, 2 * (j + 0.5) * radius, height + radius]),
            obs_inflation * radius) for i in rng for j in rng]
    legs = []
    for sx, sy in ((-1, -1), (-1, 1), (1, -1), (1, 1)):
        legs += [(np.asarray([sx * (width / 2 - radius), sy * (width / 2 - radius),
                              (2 * i + 1) * radius]), obs_inflation * radius)
                 for i in range(n_height)]
    return top + legs
