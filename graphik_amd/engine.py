"""Device-side engine objects: a problem-graph template + batched kernels over HBM buffers.

`Template` plays the role of the closures returned by RiemannianSolver.create_cost_limits /
create_cost (graphik/solvers/riemannian_solver.py:77-176): it fixes the index pairs and which
of omega / psi_L / psi_U apply to each, and exposes batched cost / egrad / ehess / proj and the
trust-region solve on the GPU.  All array arguments are torch tensors on the HIP device
(fp64, contiguous); numpy inputs are copied to the device for convenience.
"""
import ctypes as C

import numpy as np
import torch

from . import _ffi


def build_terms(omega, psi_L=None, psi_U=None, use_limits=True):
    """Residual terms in the order the reference's loops visit them.

    Index pairs follow riemannian_solver.py:122-124 (limits) / :79 (no limits): row-major
    nonzeros of the upper triangles; per pair the loops of costs.py:80-207 apply an equality
    term if omega != 0, a lower hinge if psi_L != 0 and an upper hinge if psi_U != 0.
    Returns (term_i, term_j, term_kind, targets_static) where targets_static holds psi_L / psi_U
    for hinge terms and NaN for equality terms (filled per goal from D_goal).
    """
    omega = np.asarray(omega, dtype=float)
    N = omega.shape[0]
    if use_limits:
        psi_L = np.asarray(psi_L, dtype=float)
        psi_U = np.asarray(psi_U, dtype=float)
        diff = psi_L != psi_U
        inds = np.nonzero(np.triu(omega) + np.triu(diff * (psi_L > 0)) + np.triu(diff * (psi_U > 0)))
    else:
        psi_L = np.zeros((N, N))
        psi_U = np.zeros((N, N))
        inds = np.nonzero(np.triu(omega))
    ti, tj, tk, tv = [], [], [], []
    for i, j in zip(*inds):
        if omega[i, j] != 0:
            ti.append(i); tj.append(j); tk.append(_ffi.TERM_EQ); tv.append(np.nan)
        if psi_L[i, j] != 0:
            ti.append(i); tj.append(j); tk.append(_ffi.TERM_LOWER); tv.append(psi_L[i, j])
        if psi_U[i, j] != 0:
            ti.append(i); tj.append(j); tk.append(_ffi.TERM_UPPER); tv.append(psi_U[i, j])
    return (np.array(ti, dtype=np.int32), np.array(tj, dtype=np.int32),
            np.array(tk, dtype=np.int32), np.array(tv, dtype=np.float64))


def _dev(x, device):
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
    t = t.to(device=device, dtype=torch.float64)
    return t.contiguous()


class Template:
    """Goal-independent part of an IK problem family, resident on one GPU."""

    def __init__(self, N, k, term_i, term_j, term_kind, targets_static=None, device=None,
                 params=None):
        self.lib = _ffi.lib()
        if not torch.cuda.is_available():
            raise _ffi.GikError("no HIP device visible: graphik_amd needs an AMD GPU (gfx950)")
        self.device = torch.device(device if device is not None else
                                   f"cuda:{torch.cuda.current_device()}")
        self.N, self.k = int(N), int(k)
        self.term_i = np.ascontiguousarray(term_i, dtype=np.int32)
        self.term_j = np.ascontiguousarray(term_j, dtype=np.int32)
        self.term_kind = np.ascontiguousarray(term_kind, dtype=np.int32)
        self.T = len(self.term_i)
        self.targets_static = None if targets_static is None else \
            np.ascontiguousarray(targets_static, dtype=np.float64)
        d = _ffi.TemplateDesc()
        self.lib.gik_default_params(C.byref(d))
        d.N, d.k, d.n_terms = self.N, self.k, self.T
        d.term_i = self.term_i.ctypes.data_as(C.POINTER(C.c_int32))
        d.term_j = self.term_j.ctypes.data_as(C.POINTER(C.c_int32))
        d.term_kind = self.term_kind.ctypes.data_as(C.POINTER(C.c_int32))
        for key, val in (params or {}).items():
            if not hasattr(d, key):
                raise KeyError(f"unknown solver parameter {key!r}")
            setattr(d, key, val)
        self.params = {f: getattr(d, f) for f in ("mingradnorm", "maxiter", "maxinner", "mininner",
                                                   "theta", "kappa", "rho_prime",
                                                   "rho_regularization", "planar_proj_exact")}
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.gik_template_create(C.byref(d), C.byref(h)))
        self._h = h

    @classmethod
    def from_matrices(cls, omega, psi_L=None, psi_U=None, k=3, use_limits=True, **kw):
        ti, tj, tk, tv = build_terms(omega, psi_L, psi_U, use_limits)
        return cls(np.asarray(omega).shape[0], k, ti, tj, tk, tv, **kw)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.gik_template_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- helpers ------------------------------------------------------------------------------
    def targets_from_D(self, D_goal):
        """[B,N,N] (or [N,N]) squared-distance matrices -> [B,T] per-term targets."""
        D = np.asarray(D_goal, dtype=np.float64)
        if D.ndim == 2:
            D = D[None]
        tg = D[:, self.term_i, self.term_j].copy()
        if self.targets_static is not None:
            hinge = self.term_kind != _ffi.TERM_EQ
            tg[:, hinge] = self.targets_static[hinge]
        return tg

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _vec(self, Y):
        Y = _dev(Y, self.device)
        if Y.dim() == 2:
            Y = Y[None]
        B = Y.shape[0]
        return Y.reshape(B, self.N * self.k).contiguous(), B

    def _tg(self, targets, B):
        t = _dev(targets, self.device)
        if t.dim() == 1:
            t = t[None]
        if t.shape[0] == 1 and B > 1:
            t = t.expand(B, -1).contiguous()
        assert t.shape == (B, self.T), (t.shape, (B, self.T))
        return t

    # -- costgrd twins ------------------------------------------------------------------------
    def cost(self, Y, targets):
        Y, B = self._vec(Y)
        t = self._tg(targets, B)
        out = torch.empty(B, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.gik_cost(self._h, Y.data_ptr(), t.data_ptr(), B, out.data_ptr(),
                                         self._stream()))
        return out

    def grad(self, Y, targets):
        Y, B = self._vec(Y)
        t = self._tg(targets, B)
        out = torch.empty_like(Y)
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.gik_grad(self._h, Y.data_ptr(), t.data_ptr(), B, out.data_ptr(),
                                         self._stream()))
        return out.reshape(B, self.N, self.k)

    def hess(self, Y, W, targets):
        Y, B = self._vec(Y)
        W, _ = self._vec(W)
        t = self._tg(targets, B)
        out = torch.empty_like(Y)
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.gik_hess(self._h, Y.data_ptr(), W.data_ptr(), t.data_ptr(), B,
                                         out.data_ptr(), self._stream()))
        return out.reshape(B, self.N, self.k)

    def proj(self, Y, Z):
        Y, B = self._vec(Y)
        Z, _ = self._vec(Z)
        out = torch.empty_like(Y)
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.gik_proj(self._h, Y.data_ptr(), Z.data_ptr(), B, out.data_ptr(),
                                         self._stream()))
        return out.reshape(B, self.N, self.k)

    # -- trust-region solve -------------------------------------------------------------------
    def solve(self, Y_init, targets, trace_cap=0):
        """Batched TrustRegions.solve.  Returns dict of device tensors:
        x [B,N,k], f, gradnorm, iterations, inner_total, stop, n_accept (+ trace arrays)."""
        Y, B = self._vec(Y_init)
        t = self._tg(targets, B)
        out = torch.empty_like(Y)
        stats = torch.zeros(B, 4, dtype=torch.float64, device=self.device)  # 32 B / problem
        tr = None
        keep = {}
        if trace_cap > 0:
            tr = _ffi.Trace()
            tr.cap = trace_cap
            for name, dt in (("Delta", torch.float64), ("numit", torch.int32),
                             ("stop", torch.int32), ("f_before", torch.float64),
                             ("gradnorm_after", torch.float64), ("accept", torch.int32)):
                fill = float("nan") if dt == torch.float64 else -9
                keep[name] = torch.full((B, trace_cap), fill, dtype=dt, device=self.device)
                setattr(tr, "d_" + name, keep[name].data_ptr())
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.gik_solve_batch(self._h, Y.data_ptr(), t.data_ptr(), B,
                                                out.data_ptr(), stats.data_ptr(),
                                                C.byref(tr) if tr is not None else None,
                                                self._stream()))
        ints = stats.view(torch.int32)  # [B, 8]
        res = {"x": out.reshape(B, self.N, self.k), "f": stats[:, 0], "gradnorm": stats[:, 1],
               "iterations": ints[:, 4], "inner_total": ints[:, 5], "stop": ints[:, 6],
               "n_accept": ints[:, 7]}
        if tr is not None:
            res["trace"] = keep
        return res
