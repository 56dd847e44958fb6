import numpy as np


class Manifold(object):
    """Base class: the reference's PSDFixedRank only inherits zerovec()."""

    def zerovec(self, X):
        return np.zeros(np.shape(X))
