"""dev: extended-precision restatement of lgrad (costs.py:98-123) used by the accuracy probes."""
import numpy as np
LD = np.longdouble


def lgrad_ld(Y, D, il, om, pL, pU):
    Y = Y.astype(LD); G = np.zeros_like(Y)
    for i, j in zip(*il):
        y = Y[i] - Y[j]; nrm = (y * y).sum(); c = LD(0)
        if om[i, j] > 0: c += nrm - LD(D[i, j])
        if pL[i, j] > 0 and pL[i, j] - nrm > 0: c += nrm - LD(pL[i, j])
        if pU[i, j] > 0 and nrm - pU[i, j] > 0: c += nrm - LD(pU[i, j])
        G[i] += 2 * c * y; G[j] -= 2 * c * y
    return G
