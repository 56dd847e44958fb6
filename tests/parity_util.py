"""Shared helpers of the trajectory-parity tests (identical discrete decisions and
f, |grad| to 1e-8 relative for the outer iterations k <= 20 -- as far as the trajectory itself is
reproducible; see stable_prefix)."""
import json
import os

import numpy as np

from conftest import REPO

TRAJ_KEYS = ("Delta", "numit", "stop", "f_before", "gradnorm_after", "accept")
CONTRACT_K = 20          # prefix pinned for k <= 20
CONTRACT_RTOL = 1e-8     # ... f, |grad| to 1e-8 relative


def first_divergence(a, b, n, rtol=CONTRACT_RTOL):
    """First outer iteration k < n at which two traces differ in a discrete decision (inner
    iteration count, tCG stop reason, accept flag, radius) or in f / |grad| by more than rtol
    (n if they never do)."""
    for k in range(n):
        for key in ("numit", "stop", "accept", "Delta"):
            if a[key][k] != b[key][k]:
                return k
        for key in ("f_before", "gradnorm_after"):
            if not abs(a[key][k] - b[key][k]) <= rtol * abs(b[key][k]):
                return k
    return n


def golden_traj(d, prefix, g):
    return {k: d[f"{prefix}_traj_{k}"][g] for k in TRAJ_KEYS}


def stable_prefix(a, b, n):
    """Number of leading outer iterations over which two renderings of the reference algorithm that
    differ only in summation order (the numpy closures vs the costs.py loops; the oracle vs either)
    are still the SAME computation: identical decisions, f and |grad| equal to 1e-12.  3-D solves
    amplify round-off by ~10x every few outer iterations, so beyond this index the
    reference does not reproduce itself and nothing else can be asked to."""
    return first_divergence(a, b, n, rtol=1e-12)


def assert_prefix_equal(t, o, m, rtol=CONTRACT_RTOL):
    for key in ("numit", "stop", "accept", "Delta"):
        assert np.array_equal(t[key][:m], o[key][:m]), (key, t[key][:m], o[key][:m])
    for key in ("f_before", "gradnorm_after"):
        assert np.allclose(t[key][:m], o[key][:m], rtol=rtol, atol=0), (key, t[key][:m], o[key][:m])


def wrap_abs(a):
    return np.abs(np.mod(a + np.pi, 2 * np.pi) - np.pi)


def report(section, payload):
    """Append measured parity distributions to gpurun_out/parity_report.json (copied to profiles/
    by hand); never fails a test."""
    try:
        path = os.path.join(REPO, "gpurun_out", "parity_report.json")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[section] = payload
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except Exception:
        pass
    print(section, json.dumps(payload))


# ---- fixed-anchor formulation: explicit term lists and a plain numpy evaluation ----
def anchored_terms(ap, goal_anchor):
    """Explicit point-to-anchor term list (node, position, squared target, kind) of one problem:
    pinned terms + one lower hinge per (p-node, obstacle)."""
    pos_tab = np.zeros((len(ap.anchors), 3))
    nb = len(ap.anchors) - 2
    pos_tab[:nb] = ap.base.anchor_pos
    pos_tab[nb:] = goal_anchor.reshape(2, 3)
    node = [p[0] for p in ap.pin]
    pos = [pos_tab[p[1]] for p in ap.pin]
    tgt = [p[3] for p in ap.pin]
    kind = [p[2] for p in ap.pin]
    for i in np.nonzero(ap.obs_mask)[0]:
        for o in ap.obstacles:
            node.append(int(i)); pos.append(o[:3]); tgt.append(o[3] ** 2); kind.append(2)
    return np.array(node, dtype=np.int32), np.array(pos), np.array(tgt), np.array(kind, dtype=np.int32)


def anchored_numpy(ap, Nf, Y, W, goal_anchor):
    """Plain fp64 reference of the anchored cost, egrad (= 1/2 grad f, costs.py convention) and
    ehess: free-free terms + point-to-anchor terms."""
    ti, tj, tk, target = ap.free_terms
    f, G, H = 0.0, np.zeros_like(Y), np.zeros_like(Y)

    def term(yi, wi, yj, wj, tgt, kind):
        y, w = yi - yj, wi - wj
        d = y @ y
        u = tgt - d
        act = kind == 1 or (kind == 2 and u > 0) or (kind == 3 and u < 0)
        if not act:
            return 0.0, 0 * y, 0 * y
        c = d - tgt
        return u * u, 2 * c * y, 2 * (2 * (y @ w) * y + c * w)
    for i, j, k_, t in zip(ti, tj, tk, target):
        df, dg, dh = term(Y[i], W[i], Y[j], W[j], t, k_)
        f += df; G[i] += dg; G[j] -= dg; H[i] += dh; H[j] -= dh
    node, pos, tgt, kind = anchored_terms(ap, goal_anchor)
    for i, a, t, k_ in zip(node, pos, tgt, kind):
        df, dg, dh = term(Y[i], W[i], a, 0 * a, t, k_)
        f += df; G[i] += dg; H[i] += dh
    return f, G, H
