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


def _alloc_stats(B, device):
    """[B] gik_stats records as a [B, sizeof(gik_stats) / 8] fp64 buffer."""
    assert _ffi.STATS_BYTES % 8 == 0
    return torch.zeros(B, _ffi.STATS_BYTES // 8, dtype=torch.float64, device=device)


def _decode_stats(stats):
    """Views of the gik_stats fields (layout taken from _ffi.Stats, i.e. from the header)."""
    ints = stats.view(torch.int32)
    out = {name: stats[:, col] for name, col in _ffi.STATS_F64.items()}      # f, gradnorm, stepsize
    for name in ("iterations", "inner_total", "stop", "n_accept", "inner_executed", "flags"):
        out[name] = ints[:, _ffi.STATS_I32[name]]
    return out


class Template:
    """Goal-independent part of an IK problem family, resident on one GPU."""

    def __init__(self, N, k, term_i, term_j, term_kind, targets_static=None, device=None,
                 params=None, anchored=None):
        """anchored: None, or the fixed-anchor data of gik_anchored_desc as a dict (anchor_pos
        [A,3], n_goal_anchor, term_target [T], pin_node / pin_anchor / pin_kind / pin_target,
        obs [n_obs,4] (x, y, z, r^2), obs_node_mask [N], full_N, free_full_index,
        anchor_full_index, axis_length) -- see include/graphik_amd.h."""
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
        params = dict(params or {})
        self.solver = params.pop("solver", "TrustRegions")
        if self.solver == "ConjugateGradient":      # riemannian_solver.py:51-59
            self.lib.gik_default_cg_params(C.byref(d))
        elif self.solver == "TrustRegions":
            self.lib.gik_default_params(C.byref(d))
        else:
            raise ValueError("params[\"solver\"] must be one of 'ConjugateGradient', 'TrustRegions'")
        d.N, d.k, d.n_terms = self.N, self.k, self.T
        d.term_i = self.term_i.ctypes.data_as(C.POINTER(C.c_int32))
        d.term_j = self.term_j.ctypes.data_as(C.POINTER(C.c_int32))
        d.term_kind = self.term_kind.ctypes.data_as(C.POINTER(C.c_int32))
        alias = {"minstepsize": "cg_minstepsize", "orth_value": "cg_orth_value", "beta_type": "cg_beta_type"}
        for key, val in params.items():
            key = alias.get(key, key)
            if not hasattr(d, key):
                raise KeyError(f"unknown solver parameter {key!r}")
            if key == "clique_closed_form" and isinstance(val, str):
                val = {"auto": _ffi.CLIQUE_AUTO, "off": _ffi.CLIQUE_OFF, "dense": _ffi.CLIQUE_DENSE}[val]
            if key == "hessian_form" and isinstance(val, str):
                val = {"column": _ffi.HESS_COLUMN, "per_edge": _ffi.HESS_PER_EDGE, "auto": _ffi.HESS_AUTO}[val]
            setattr(d, key, int(val) if key in ("maxiter", "cg_beta_type", "clique_closed_form", "hessian_form") else val)
        self.params = {f: getattr(d, f) for f in ("mingradnorm", "maxiter", "maxinner", "mininner",
                                                   "theta", "kappa", "rho_prime",
                                                   "rho_regularization", "planar_proj_exact",
                                                   "force_block_path", "waves_per_cu",
                                                   "slice_outer_its", "debug_flags", "cg_minstepsize",
                                                   "cg_orth_value", "cg_beta_type", "clique_closed_form", "hessian_form")}
        self.params["solver"] = self.solver
        h = C.c_void_p()
        self.anchored = anchored is not None
        with torch.cuda.device(self.device):
            if anchored is None:
                _ffi.check(self.lib.gik_template_create(C.byref(d), C.byref(h)))
            else:
                ad, keep = _ffi.AnchoredDesc(), {}

                def arr(name, dt):
                    keep[name] = np.ascontiguousarray(anchored[name], dtype=dt)
                    return keep[name].ctypes.data_as(C.POINTER(C.c_double if dt == np.float64 else C.c_int32))

                ad.anchor_pos = arr("anchor_pos", np.float64)
                ad.n_anchor = len(keep["anchor_pos"])
                ad.n_goal_anchor = int(anchored["n_goal_anchor"])
                ad.term_target = arr("term_target", np.float64)
                assert len(keep["term_target"]) == self.T
                ad.pin_node, ad.pin_anchor = arr("pin_node", np.int32), arr("pin_anchor", np.int32)
                ad.pin_kind, ad.pin_target = arr("pin_kind", np.int32), arr("pin_target", np.float64)
                ad.n_pin = len(keep["pin_node"])
                ad.obs = arr("obs", np.float64)
                ad.n_obs = len(keep["obs"])
                ad.obs_node_mask = arr("obs_node_mask", np.int32)
                ad.full_N = int(anchored["full_N"])
                ad.free_full_index = arr("free_full_index", np.int32)
                ad.anchor_full_index = arr("anchor_full_index", np.int32)
                ad.axis_length = float(anchored["axis_length"])
                self.n_goal_anchor, self.full_N = ad.n_goal_anchor, ad.full_N
                _ffi.check(self.lib.gik_template_create_anchored(C.byref(d), C.byref(ad), C.byref(h)))
        self._h = h
        self._read_info()
        deg = np.bincount(np.concatenate([self.term_i, self.term_j]), minlength=self.N).max()
        # compiled slot count of the wavefront variant the library chose (or the raw degree: workgroup / node-per-lane paths)
        self.maxdeg = int(self.info["max_terms_per_node"]) if not self.info["is_block"] else int(deg)

    def _read_info(self):
        """What the library decided for this handle (gik_template_get_info); read again after attach_pipeline,
        which decides the prepare kernel."""
        info = _ffi.TemplateInfo()
        _ffi.check(self.lib.gik_template_get_info(self._h, C.byref(info)))
        self.info = {f: getattr(info, f) for f, _ in _ffi.TemplateInfo._fields_ if f != "reserved"}

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
        width = self.T if not self.anchored else 3 * self.n_goal_anchor   # anchored: goal anchors
        assert t.shape == (B, width), (t.shape, (B, width))
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

    def cost_and_grad(self, Y, targets):
        """lcost_and_grad / jcost_and_grad (costs.py:126-169, 61-77): (f [B], G [B,N,k]) in one pass."""
        Y, B = self._vec(Y)
        t = self._tg(targets, B)
        f = torch.empty(B, dtype=torch.float64, device=Y.device)
        out = torch.empty_like(Y)
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.gik_cost_and_grad(self._h, Y.data_ptr(), t.data_ptr(), B, f.data_ptr(),
                                                  out.data_ptr(), self._stream()))
        return f, out.reshape(B, self.N, self.k)

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

    # -- device pre/post-processing ----------------------------------------------------------
    def attach_pipeline(self, *, T0, p_index, q_index, x_index, y_index, axis_length, goal_nodes,
                        goal_len, base_lower, base_upper, anchor_index, anchor_pos, pair_i, pair_j,
                        term_src, term_static, last_link_along_z, jacobi_sweeps=0,
                        force_block_prepare=False, ee_goal_nodes=None, ee_path=None,
                        goal_pair_a=(), goal_pair_b=(), ee_goal_len=None):
        """Give the handle what it needs to run from_pose + bound_smoothing +
        generate_initialization and joint_variables on the device (gik_pipeline_attach)."""
        keep = {}

        def arr(name, a, dt):
            keep[name] = np.ascontiguousarray(a, dtype=dt)
            ct = C.c_double if dt == np.float64 else C.c_int32
            return keep[name].ctypes.data_as(C.POINTER(ct))

        d = _ffi.PipelineDesc()
        T0 = np.asarray(T0, dtype=np.float64)
        d.n_joints = T0.shape[0] - 1
        d.T0 = arr("T0", T0, np.float64)
        d.p_index = arr("p", p_index, np.int32)
        d.q_index = arr("q", q_index if q_index is not None else p_index, np.int32)
        d.x_index, d.y_index = int(x_index), int(y_index)
        d.axis_length = float(axis_length)
        d.goal_node0, d.goal_node1 = int(goal_nodes[0]), int(goal_nodes[1])
        d.goal_len = float(goal_len)
        d.base_lower = arr("lo", base_lower, np.float64)
        d.base_upper = arr("up", base_upper, np.float64)
        d.n_anchor = len(anchor_index)
        d.anchor_index = arr("ai", anchor_index, np.int32)
        d.anchor_pos = arr("ap", anchor_pos, np.float64)
        d.n_pairs = len(pair_i)
        d.pair_i = arr("pi", pair_i, np.int32)
        d.pair_j = arr("pj", pair_j, np.int32)
        d.term_src = arr("ts", term_src, np.int32)
        d.term_static = arr("tv", term_static, np.float64)
        d.last_link_along_z = int(last_link_along_z)       # one bit per end effector
        d.jacobi_sweeps = int(jacobi_sweeps)
        d.force_block_prepare = int(bool(force_block_prepare))
        self.n_ee = 1
        if ee_goal_nodes is not None and len(ee_goal_nodes) > 2:     # several end effectors
            self.n_ee = len(ee_goal_nodes) // 2
            d.n_ee = self.n_ee
            d.ee_goal_nodes = arr("eg", ee_goal_nodes, np.int32)
            d.ee_path = arr("ep", ee_path, np.int32)
            d.n_goal_pairs = len(goal_pair_a)
            d.goal_pair_a = arr("ga", goal_pair_a, np.int32)
            d.goal_pair_b = arr("gb", goal_pair_b, np.int32)
            if ee_goal_len is not None:                  # planar trees: the link parent(e) -> e per end effector
                d.ee_goal_len = arr("el", ee_goal_len, np.float64)
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.gik_pipeline_attach(self._h, C.byref(d)))
        self.n_joints = int(d.n_joints)
        self.has_pipeline = True
        self._read_info()

    def _poses(self, T_goal):
        T = _dev(T_goal, self.device)
        B = T.shape[0]
        width = getattr(self, "n_ee", 1) * (self.k + 1) ** 2
        assert T.numel() == B * width, (tuple(T.shape), width)
        T = T.reshape(B, width).contiguous()      # [B][n_ee][(k+1)^2]
        return T, B

    def prepare(self, T_goal, return_K=False):
        """goal poses [B,k+1,k+1] -> (targets [B,T], Y_init [B,N,k]) on the device."""
        T, B = self._poses(T_goal)
        targets = torch.empty(B, self.T, dtype=torch.float64, device=self.device)
        Y0 = torch.empty(B, self.N * self.k, dtype=torch.float64, device=self.device)
        Kc = torch.zeros(B, dtype=torch.int32, device=self.device) if return_K else None
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.gik_prepare_batch(self._h, T.data_ptr(), B, targets.data_ptr(),
                                                  Y0.data_ptr(),
                                                  Kc.data_ptr() if return_K else None,
                                                  self._stream()))
        Y0 = Y0.reshape(B, self.N, self.k)
        return (targets, Y0, Kc) if return_K else (targets, Y0)

    def prepare_debug(self, T_goal):
        """prepare() plus the intermediate results the reference computes on the way: dict with
        targets, Y_init, K (MDS column count), lb, ub [B,N,N] (bound_smoothing) and eig [B,3,N]
        (spectra of the Gram matrix, of MDS's rank matrix and of the scatter matrix)."""
        T, B = self._poses(T_goal)
        f64 = dict(dtype=torch.float64, device=self.device)
        out = {"targets": torch.empty(B, self.T, **f64), "Y_init": torch.empty(B, self.N * self.k, **f64),
               "K": torch.zeros(B, dtype=torch.int32, device=self.device),
               "lb": torch.empty(B, self.N, self.N, **f64), "ub": torch.empty(B, self.N, self.N, **f64),
               "eig": torch.empty(B, 3, self.N, **f64)}
        dg = _ffi.PrepareDiag(out["lb"].data_ptr(), out["ub"].data_ptr(), out["eig"].data_ptr())
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.gik_prepare_batch_debug(self._h, T.data_ptr(), B, out["targets"].data_ptr(),
                                                        out["Y_init"].data_ptr(), out["K"].data_ptr(),
                                                        C.byref(dg), self._stream()))
        out["Y_init"] = out["Y_init"].reshape(B, self.N, self.k)
        return out

    def recover(self, Y, T_goal):
        """points + goal poses -> (q [B,n], pos_err [B], rot_err [B]) on the device."""
        Y, B = self._vec(Y)
        T, _ = self._poses(T_goal)
        q = torch.empty(B, self.n_joints, dtype=torch.float64, device=self.device)
        pe = torch.empty(B, dtype=torch.float64, device=self.device)
        re = torch.empty(B, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.gik_recover_batch(self._h, Y.data_ptr(), T.data_ptr(), B,
                                                  q.data_ptr(), pe.data_ptr(), re.data_ptr(),
                                                  self._stream()))
        return q, pe, re

    def ik(self, T_goal, out=None):
        """Whole solve_with_riemannian pipeline for a batch of goal poses, one stream, no host
        round trip: prepare -> solve -> recover.  Returns a dict of device tensors."""
        T, B = self._poses(T_goal)
        if out is None:
            out = self.alloc_ik_buffers(B)
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.gik_ik_batch(self._h, T.data_ptr(), B, out["targets"].data_ptr(),
                                             out["Y"].data_ptr(), out["stats"].data_ptr(),
                                             out["q"].data_ptr(), out["pos_err"].data_ptr(),
                                             out["rot_err"].data_ptr(), self._stream()))
        res = {"x": out["Y"].reshape(B, self.N, self.k), "q": out["q"], "pos_err": out["pos_err"],
               "rot_err": out["rot_err"]}
        res.update(_decode_stats(out["stats"]))
        return res

    def anchored_ik(self, base, T_goal):
        """Whole pipeline through the fixed-anchor solve (gik_anchored_ik_batch): `base` is the
        robot graph's Template (no obstacles) with its pipeline attached.  Returns device tensors:
        x [B, full_N, 3] (all robot-graph nodes, anchors included), q, pos_err, rot_err + stats."""
        assert self.anchored and base.has_pipeline
        T, B = base._poses(T_goal)
        f64 = dict(dtype=torch.float64, device=self.device)
        nws = int(self.lib.gik_anchored_ws_doubles(self._h, base._h, B))
        ws = torch.empty(max(nws, 1), **f64)
        out = {"Y": torch.empty(B, self.full_N * 3, **f64), "stats": _alloc_stats(B, self.device),
               "q": torch.empty(B, base.n_joints, **f64), "pos_err": torch.empty(B, **f64),
               "rot_err": torch.empty(B, **f64)}
        with torch.cuda.device(self.device):
            _ffi.check(self.lib.gik_anchored_ik_batch(self._h, base._h, T.data_ptr(), B, ws.data_ptr(),
                                                      out["Y"].data_ptr(), out["stats"].data_ptr(),
                                                      out["q"].data_ptr(), out["pos_err"].data_ptr(),
                                                      out["rot_err"].data_ptr(), self._stream()))
        res = {"x": out["Y"].reshape(B, self.full_N, 3), "q": out["q"], "pos_err": out["pos_err"],
               "rot_err": out["rot_err"], "_ws": ws}
        res.update(_decode_stats(out["stats"]))
        return res

    def alloc_ik_buffers(self, B):
        f64 = dict(dtype=torch.float64, device=self.device)
        return {"targets": torch.empty(B, self.T, **f64), "Y": torch.empty(B, self.N * self.k, **f64),
                "stats": _alloc_stats(B, self.device), "q": torch.empty(B, self.n_joints, **f64),
                "pos_err": torch.empty(B, **f64), "rot_err": torch.empty(B, **f64)}

    # -- trust-region solve -------------------------------------------------------------------
    def solve(self, Y_init, targets, trace_cap=0):
        """Batched TrustRegions.solve.  Returns dict of device tensors:
        x [B,N,k], f, gradnorm, iterations, inner_total, stop, n_accept (+ trace arrays)."""
        Y, B = self._vec(Y_init)
        t = self._tg(targets, B)
        out = torch.empty_like(Y)
        stats = _alloc_stats(B, self.device)
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
        res = {"x": out.reshape(B, self.N, self.k)}
        res.update(_decode_stats(stats))
        if tr is not None:
            res["trace"] = keep
        return res
