"""ctypes binding of oracle/gik_oracle.c (CPU restatement; test infrastructure only).

The function names/arguments mirror the reference's `costgrd` exports (graphik/solvers/costs.py)
and `TrustRegions.solve` (graphik/solvers/trust_region.py) so tests read like calls into the
reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


class Params(C.Structure):
    _fields_ = [("mingradnorm", C.c_double), ("maxiter", C.c_int), ("maxinner", C.c_int),
                ("mininner", C.c_int), ("theta", C.c_double), ("kappa", C.c_double),
                ("rho_prime", C.c_double), ("rho_regularization", C.c_double),
                ("use_limits", C.c_int)]


class Result(C.Structure):
    _fields_ = [("f", C.c_double), ("gradnorm", C.c_double), ("iterations", C.c_int),
                ("inner_total", C.c_int), ("stop", C.c_int)]


class Traj(C.Structure):
    _fields_ = [("cap", C.c_int), ("len", C.c_int), ("Delta", C.POINTER(C.c_double)),
                ("numit", C.POINTER(C.c_int)), ("stop", C.POINTER(C.c_int)),
                ("f_before", C.POINTER(C.c_double)), ("gradnorm_after", C.POINTER(C.c_double)),
                ("accept", C.POINTER(C.c_int))]


class CgParams(C.Structure):
    _fields_ = [("mingradnorm", C.c_double), ("maxiter", C.c_int), ("minstepsize", C.c_double),
                ("orth_value", C.c_double), ("beta_type", C.c_int), ("use_limits", C.c_int),
                ("ls_contraction", C.c_double), ("ls_suff_decr", C.c_double),
                ("ls_initial_stepsize", C.c_double), ("ls_maxiter", C.c_int)]


class CgTraj(C.Structure):
    _fields_ = [("cap", C.c_int), ("len", C.c_int), ("f", C.POINTER(C.c_double)),
                ("gradnorm", C.POINTER(C.c_double)), ("stepsize", C.POINTER(C.c_double)),
                ("costevals", C.POINTER(C.c_int))]


class AnchorTerms(C.Structure):
    _fields_ = [("n", C.c_int), ("node", C.POINTER(C.c_int)), ("pos", C.POINTER(C.c_double)),
                ("target", C.POINTER(C.c_double)), ("kind", C.POINTER(C.c_int))]


def build(force=False):
    """Compile oracle/_build/*.so with gcc (idempotent)."""
    so = os.path.join(_HERE, "_build", "libgik_oracle.so")
    if force or not os.path.exists(so) or \
            os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "gik_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


_libs = {}


def lib(fast=False):
    key = "fast" if fast else "strict"
    if key not in _libs:
        build()
        name = "libgik_oracle_fast.so" if fast else "libgik_oracle.so"
        L = C.CDLL(os.path.join(_HERE, "_build", name))
        L.gik_o_jcost.restype = C.c_double
        L.gik_o_jcost.argtypes = [_dp, _dp, _ip, _ip, C.c_int64, C.c_int, C.c_int]
        L.gik_o_jgrad.argtypes = [_dp, _dp, _ip, _ip, C.c_int64, C.c_int, C.c_int, _dp]
        L.gik_o_jhess.argtypes = [_dp, _dp, _dp, _ip, _ip, C.c_int64, C.c_int, C.c_int, _dp]
        L.gik_o_jcost_and_grad.restype = C.c_double
        L.gik_o_jcost_and_grad.argtypes = [_dp, _dp, _ip, _ip, C.c_int64, C.c_int, C.c_int, _dp]
        L.gik_o_lcost_and_grad.restype = C.c_double
        L.gik_o_lcost_and_grad.argtypes = [_dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int64, C.c_int, C.c_int, _dp]
        L.gik_o_lcost.restype = C.c_double
        L.gik_o_lcost.argtypes = [_dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int64, C.c_int, C.c_int]
        L.gik_o_lgrad.argtypes = [_dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int64, C.c_int, C.c_int,
                                  _dp]
        L.gik_o_lhess.argtypes = [_dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int64, C.c_int,
                                  C.c_int, _dp]
        L.gik_o_proj.argtypes = [_dp, _dp, C.c_int, C.c_int, _dp]
        L.gik_o_default_params.argtypes = [C.POINTER(Params)]
        L.gik_o_rtr_solve.argtypes = [_dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int64, C.c_int,
                                      C.c_int, C.POINTER(Params), C.POINTER(Result),
                                      C.POINTER(Traj)]
        L.gik_o_rtr_solve_anchored.argtypes = [_dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int64, C.c_int,
                                               C.c_int, C.POINTER(AnchorTerms), C.POINTER(Params),
                                               C.POINTER(Result), C.POINTER(Traj)]
        L.gik_o_cg_default_params.argtypes = [C.POINTER(CgParams)]
        L.gik_o_cg_solve.argtypes = [_dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int64, C.c_int, C.c_int,
                                     C.POINTER(CgParams), C.POINTER(Result), C.POINTER(CgTraj)]
        L.gik_o_rtr_solve_batch.argtypes = [_dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int64, C.c_int,
                                            C.c_int, C.c_int, C.POINTER(Params),
                                            C.POINTER(Result), C.c_int]
        L.gik_o_bound_smoothing.argtypes = [_dp, _dp, C.c_int, _dp, _dp]
        _libs[key] = L
    return _libs[key]


def _c(a, dt=np.float64):
    return np.ascontiguousarray(a, dtype=dt)


def _inds(inds):
    return _c(inds[0], np.int64), _c(inds[1], np.int64)


# ---- costgrd twins (graphik/solvers/costs.py) ------------------------------------------------
def jcost(Y, D_goal, inds):
    ii, jj = _inds(inds)
    Y = _c(Y)
    return lib().gik_o_jcost(Y, _c(D_goal), ii, jj, len(ii), Y.shape[0], Y.shape[1])


def jgrad(Y, D_goal, inds):
    ii, jj = _inds(inds)
    Y = _c(Y)
    out = np.empty_like(Y)
    lib().gik_o_jgrad(Y, _c(D_goal), ii, jj, len(ii), Y.shape[0], Y.shape[1], out)
    return out


def jhess(Y, w, D_goal, inds):
    ii, jj = _inds(inds)
    Y = _c(Y)
    out = np.empty_like(Y)
    lib().gik_o_jhess(Y, _c(w), _c(D_goal), ii, jj, len(ii), Y.shape[0], Y.shape[1], out)
    return out


def lcost(Y, D_goal, omega, psi_L, psi_U, inds):
    ii, jj = _inds(inds)
    Y = _c(Y)
    return lib().gik_o_lcost(Y, _c(D_goal), _c(omega), _c(psi_L), _c(psi_U), ii, jj, len(ii),
                             Y.shape[0], Y.shape[1])


def lgrad(Y, D_goal, omega, psi_L, psi_U, inds):
    ii, jj = _inds(inds)
    Y = _c(Y)
    out = np.empty_like(Y)
    lib().gik_o_lgrad(Y, _c(D_goal), _c(omega), _c(psi_L), _c(psi_U), ii, jj, len(ii),
                      Y.shape[0], Y.shape[1], out)
    return out


def jcost_and_grad(Y, D_goal, inds):
    """costs.py:61-77 -> (f, G)."""
    ii, jj = _inds(inds)
    Y = _c(Y)
    out = np.empty_like(Y)
    f = lib().gik_o_jcost_and_grad(Y, _c(D_goal), ii, jj, len(ii), Y.shape[0], Y.shape[1], out)
    return f, out


def lcost_and_grad(Y, D_goal, omega, psi_L, psi_U, inds):
    """costs.py:126-169 -> (f, G)."""
    ii, jj = _inds(inds)
    Y = _c(Y)
    out = np.empty_like(Y)
    f = lib().gik_o_lcost_and_grad(Y, _c(D_goal), _c(omega), _c(psi_L), _c(psi_U), ii, jj, len(ii),
                                   Y.shape[0], Y.shape[1], out)
    return f, out


def lhess(Y, w, D_goal, omega, psi_L, psi_U, inds):
    ii, jj = _inds(inds)
    Y = _c(Y)
    out = np.empty_like(Y)
    lib().gik_o_lhess(Y, _c(w), _c(D_goal), _c(omega), _c(psi_L), _c(psi_U), ii, jj, len(ii),
                      Y.shape[0], Y.shape[1], out)
    return out


def proj(Y, Z):
    """PSDFixedRank.proj (fixed_rank_psd_sym.py:91-113)."""
    Y = _c(Y)
    out = np.empty_like(Y)
    rc = lib().gik_o_proj(Y, _c(Z), Y.shape[0], Y.shape[1], out)
    if rc != 0:
        raise np.linalg.LinAlgError("singular system in proj")
    return out


def default_params(**kw):
    p = Params()
    lib().gik_o_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def limit_inds(omega, psi_L, psi_U):
    """Index pairs of create_cost_limits (riemannian_solver.py:122-124)."""
    diff = psi_L != psi_U
    return np.nonzero(np.triu(omega) + np.triu(diff * (psi_L > 0)) + np.triu(diff * (psi_U > 0)))


def rtr_solve(Y_init, D_goal, omega, psi_L=None, psi_U=None, use_limits=True, traj_cap=0, **kw):
    """TrustRegions.solve on create_cost_limits / create_cost (loop form)."""
    Y = _c(Y_init).copy()
    N, k = Y.shape
    omega = _c(omega)
    if use_limits:
        psi_L, psi_U = _c(psi_L), _c(psi_U)
        inds = limit_inds(omega, psi_L, psi_U)
    else:
        psi_L, psi_U = np.zeros_like(omega), np.zeros_like(omega)
        inds = np.nonzero(np.triu(omega))  # riemannian_solver.py:79
    ii, jj = _inds(inds)
    p = default_params(use_limits=int(use_limits), **kw)
    res = Result()
    tr = None
    keep = {}
    if traj_cap > 0:
        tr = Traj()
        tr.cap = traj_cap
        for name, ct, dt in (("Delta", C.c_double, np.float64), ("numit", C.c_int, np.int32),
                             ("stop", C.c_int, np.int32), ("f_before", C.c_double, np.float64),
                             ("gradnorm_after", C.c_double, np.float64),
                             ("accept", C.c_int, np.int32)):
            keep[name] = np.zeros(traj_cap, dtype=dt)
            setattr(tr, name, keep[name].ctypes.data_as(C.POINTER(ct)))
    rc = lib().gik_o_rtr_solve(Y, _c(D_goal), omega, psi_L, psi_U, ii, jj, len(ii), N, k,
                               C.byref(p), C.byref(res), C.byref(tr) if tr else None)
    assert rc == 0
    info = {"x": Y, "f(x)": res.f, "gradnorm": res.gradnorm, "iterations": res.iterations,
            "inner_total": res.inner_total, "stop": res.stop}
    if tr:
        info["traj"] = {k_: v[: tr.len] for k_, v in keep.items()}
    return info


def rtr_solve_batch(Y_init, D_goal, omega, psi_L, psi_U, use_limits=True, nthreads=0, fast=True,
                    **kw):
    Y = _c(Y_init).copy()
    B, N, k = Y.shape
    omega = _c(omega)
    if use_limits:
        inds = limit_inds(omega, psi_L, psi_U)
    else:
        psi_L, psi_U = np.zeros_like(omega), np.zeros_like(omega)
        inds = np.nonzero(np.triu(omega))
    ii, jj = _inds(inds)
    p = default_params(use_limits=int(use_limits), **kw)
    res = (Result * B)()
    L = lib(fast=fast)
    rc = L.gik_o_rtr_solve_batch(Y, _c(D_goal), omega, _c(psi_L), _c(psi_U), ii, jj, len(ii), N, k,
                                 B, C.byref(p), res, nthreads)
    assert rc == 0
    out = {"x": Y, "f(x)": np.array([r.f for r in res]),
           "gradnorm": np.array([r.gradnorm for r in res]),
           "iterations": np.array([r.iterations for r in res]),
           "inner_total": np.array([r.inner_total for r in res])}
    return out


def rtr_solve_anchored(Y_init, D_ff, omega_ff, psi_L_ff, psi_U_ff, at_node, at_pos, at_target, at_kind,
                       traj_cap=0, **kw):
    """Trust-region solve of the fixed-anchor formulation (gik_o_rtr_solve_anchored): Y_init
    [Nf,3] free nodes, dense free-free matrices, point-to-anchor terms (node, position, squared
    target, kind)."""
    Y = _c(Y_init).copy()
    N, k = Y.shape
    omega, psi_L, psi_U = _c(omega_ff), _c(psi_L_ff), _c(psi_U_ff)
    ii, jj = _inds(limit_inds(omega, psi_L, psi_U))
    p = default_params(use_limits=1, **kw)
    node = np.ascontiguousarray(at_node, dtype=np.int32)
    pos = np.ascontiguousarray(at_pos, dtype=np.float64)
    tgt = np.ascontiguousarray(at_target, dtype=np.float64)
    kind = np.ascontiguousarray(at_kind, dtype=np.int32)
    at = AnchorTerms(len(node), node.ctypes.data_as(C.POINTER(C.c_int)), pos.ctypes.data_as(C.POINTER(C.c_double)),
                     tgt.ctypes.data_as(C.POINTER(C.c_double)), kind.ctypes.data_as(C.POINTER(C.c_int)))
    res = Result()
    tr, keep = None, {}
    if traj_cap > 0:
        tr = Traj()
        tr.cap = traj_cap
        for name, ct, dt in (("Delta", C.c_double, np.float64), ("numit", C.c_int, np.int32),
                             ("stop", C.c_int, np.int32), ("f_before", C.c_double, np.float64),
                             ("gradnorm_after", C.c_double, np.float64), ("accept", C.c_int, np.int32)):
            keep[name] = np.zeros(traj_cap, dtype=dt)
            setattr(tr, name, keep[name].ctypes.data_as(C.POINTER(ct)))
    rc = lib().gik_o_rtr_solve_anchored(Y, _c(D_ff), omega, psi_L, psi_U, ii, jj, len(ii), N, k, C.byref(at),
                                        C.byref(p), C.byref(res), C.byref(tr) if tr else None)
    assert rc == 0
    info = {"x": Y, "f(x)": res.f, "gradnorm": res.gradnorm, "iterations": res.iterations,
            "inner_total": res.inner_total, "stop": res.stop}
    if tr:
        info["traj"] = {k_: v[: tr.len] for k_, v in keep.items()}
    return info


def cg_solve(Y_init, D_goal, omega, psi_L=None, psi_U=None, use_limits=True, traj_cap=0, **kw):
    """RiemannianSolver(graph, {"solver": "ConjugateGradient"}).solve (riemannian_solver.py:51-59):
    pymanopt 0.2.5 ConjugateGradient (HagerZhang) + LineSearchAdaptive, restated in C."""
    Y = _c(Y_init).copy()
    N, k = Y.shape
    omega = _c(omega)
    if use_limits:
        psi_L, psi_U = _c(psi_L), _c(psi_U)
        inds = limit_inds(omega, psi_L, psi_U)
    else:
        psi_L, psi_U = np.zeros_like(omega), np.zeros_like(omega)
        inds = np.nonzero(np.triu(omega))
    ii, jj = _inds(inds)
    p = CgParams()
    lib().gik_o_cg_default_params(C.byref(p))
    p.use_limits = int(use_limits)
    for k_, v in kw.items():
        setattr(p, k_, v)
    res = Result()
    tr, keep = None, {}
    if traj_cap > 0:
        tr = CgTraj()
        tr.cap = traj_cap
        for name, ct, dt in (("f", C.c_double, np.float64), ("gradnorm", C.c_double, np.float64),
                             ("stepsize", C.c_double, np.float64), ("costevals", C.c_int, np.int32)):
            keep[name] = np.zeros(traj_cap, dtype=dt)
            setattr(tr, name, keep[name].ctypes.data_as(C.POINTER(ct)))
    rc = lib().gik_o_cg_solve(Y, _c(D_goal), omega, psi_L, psi_U, ii, jj, len(ii), N, k, C.byref(p),
                              C.byref(res), C.byref(tr) if tr else None)
    assert rc == 0
    info = {"x": Y, "f(x)": res.f, "gradnorm": res.gradnorm, "iterations": res.iterations,
            "costevals": res.inner_total, "stop": res.stop}
    if tr:
        info["traj"] = {k_: v[: tr.len] for k_, v in keep.items()}
    return info


def bound_smoothing(lower, upper):
    """dgp.py:192-231 on dense LOWER/UPPER matrices (NaN = no edge)."""
    lower, upper = _c(lower), _c(upper)
    N = lower.shape[0]
    lb = np.empty((N, N))
    ub = np.empty((N, N))
    lib().gik_o_bound_smoothing(lower, upper, N, lb, ub)
    return lb, ub
