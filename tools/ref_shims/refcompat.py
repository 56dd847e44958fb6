"""Compatibility patches applied to the *imported* reference (numpy>=2, networkx>=3).

Import this before any `graphik` module.  No file under /root/reference is modified.
"""
import math
import warnings

import numpy as np

warnings.filterwarnings("ignore")
if not hasattr(np, "math"):
    np.math = math  # graph_planar.py:171 uses np.math.atan2 (removed in numpy 2.0)


def patch_skew():
    """geometry.skew() is handed a (3,1) column at roboturdf.py:282-287 -> ragged array on
    numpy>=1.24.  Wrap it so the argument is flattened first (same values)."""
    import graphik.utils.geometry as geo
    import graphik.utils as gu
    import graphik.utils.roboturdf as ru
    import graphik.graphs.graph_revolute as gr

    orig = geo.skew

    def skew(x):
        return orig(np.asarray(x, dtype=float).ravel())

    for mod in (geo, gu, ru, gr):
        if hasattr(mod, "skew"):
            mod.skew = skew
