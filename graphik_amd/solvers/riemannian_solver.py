"""Drop-in front end for GraphIK's Riemannian solver
(graphik/solvers/riemannian_solver.py), running on the MI355X engine.

    from graphik_amd.solvers.riemannian_solver import solve_with_riemannian, RiemannianSolver

`solve_with_riemannian(graph, T_goal, use_jit=True)` and `RiemannianSolver(graph, params)
.solve(D_goal, omega, use_limits, bounds, Y_init, jit, output_log)` keep the reference's
signatures and return shapes; `jit` / `use_jit` are accepted and ignored (there is exactly one
implementation: the HIP kernels -- no CPU fallback).  `solve_batch` is the batched entry point
the engine is built for.
"""
import collections
import hashlib
import time

import numpy as np
import torch

from ..engine import Template, build_terms
from ..utils import dgp
from ..utils.constants import POS
from ..utils.lie import as_matrix

_STOP_REASONS = {0: "Terminated - min grad norm reached", 1: "Terminated - max iterations reached",
                 2: "Terminated - NaN encountered", 3: "Terminated - min stepsize reached"}
# riemannian_solver.py:24-26 (indexable like pymanopt's make_enum: BetaTypes[3] == "HagerZhang")
BetaTypes = ["FletcherReeves", "PolakRibiere", "HestenesStiefel", "HagerZhang"]


# Templates behind the create_cost / create_cost_limits closures: least-recently-used, bounded like
# _PROBLEM_CACHE (a Template owns device tables and a HIP handle), keyed by the device as well, and
# emptied by clear_problem_cache()
_closure_templates = collections.OrderedDict()
_CLOSURE_TEMPLATES_MAX = 8


def _cost_closures(D_goal, omega, psi_L, psi_U, use_limits):
    """The closure triple of create_cost / create_cost_limits on the default device, reference
    solver defaults; one cached Template per (k, masks)."""
    omega = np.asarray(omega, dtype=float)
    psi_L = None if psi_L is None else np.asarray(psi_L, dtype=float)
    psi_U = None if psi_U is None else np.asarray(psi_U, dtype=float)
    D_goal = np.asarray(D_goal, dtype=float)
    state = {}

    def tpl(Y):
        k = int(np.asarray(Y).shape[-1])
        if state.get("k") != k:
            dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
            key = (dev, k, use_limits, omega.tobytes(), None if psi_L is None else psi_L.tobytes(),
                   None if psi_U is None else psi_U.tobytes())
            if key not in _closure_templates:
                _closure_templates[key] = Template.from_matrices(omega, psi_L, psi_U, k=k, use_limits=use_limits)
                while len(_closure_templates) > _CLOSURE_TEMPLATES_MAX:
                    _closure_templates.popitem(last=False)
            else:
                _closure_templates.move_to_end(key)
            state["k"], state["T"] = k, _closure_templates[key]
            state["tg"] = state["T"].targets_from_D(D_goal)
        return state["T"], state["tg"]

    def cost(Y):
        T, tg = tpl(Y)
        return float(T.cost(Y, tg)[0])

    def egrad(Y):
        T, tg = tpl(Y)
        return T.grad(Y, tg)[0].cpu().numpy()

    def ehess(Y, Z):
        T, tg = tpl(Y)
        return T.hess(Y, Z, tg)[0].cpu().numpy()

    return cost, egrad, ehess


class RiemannianSolver:
    def __init__(self, graph, params={}):
        self.params = params
        self.graph = graph
        self.dim = graph.dim
        self.N = graph.number_of_nodes()
        solver_type = params.get("solver", "TrustRegions")
        if solver_type == "TrustRegions":
            # riemannian_solver.py:44-50
            self.tr_params = {"mingradnorm": params.get("mingradnorm", 0.5 * 1e-9),
                              "maxiter": int(params.get("maxiter", 3000)),
                              "theta": params.get("theta", 1.0), "kappa": params.get("kappa", 0.1)}
        elif solver_type == "ConjugateGradient":
            # riemannian_solver.py:51-59: pymanopt's ConjugateGradient (HagerZhang, adaptive line
            # search), on the device like the trust-region solver (rcg_* kernels)
            beta = params.get("beta_type", 3)
            if isinstance(beta, str):
                beta = BetaTypes.index(beta)
            self.tr_params = {"solver": "ConjugateGradient",
                              "mingradnorm": params.get("mingradnorm", 1e-9),
                              "maxiter": int(params.get("maxiter", 10e4)),
                              "minstepsize": params.get("minstepsize", 1e-10),
                              "orth_value": params.get("orth_value", 10e10), "beta_type": int(beta)}
        else:
            raise ValueError("params[\"solver\"] must be one of 'ConjugateGradient', 'TrustRegions'")
        for k in ("maxinner", "mininner", "rho_prime", "rho_regularization", "planar_proj_exact",
                  "force_block_path", "waves_per_cu", "slice_outer_its", "debug_flags",
                  "clique_closed_form", "hessian_form"):
            if k in params:
                self.tr_params[k] = params[k]
        self.device = params.get("device", None)
        self._templates = {}

    # -- statics kept for API parity -----------------------------------------------------------
    @staticmethod
    def generate_initialization(bounds, dim, omega, psi_L=None, psi_U=None):
        """riemannian_solver.py:67-75"""
        return dgp.generate_initialization(bounds, dim, omega)

    def _template(self, omega, psi_L, psi_U, use_limits):
        key = (use_limits, omega.tobytes(), None if psi_L is None else psi_L.tobytes(),
               None if psi_U is None else psi_U.tobytes())
        if key not in self._templates:
            self._templates[key] = Template.from_matrices(
                omega, psi_L, psi_U, k=self.dim, use_limits=use_limits, device=self.device,
                params=self.tr_params)
        return self._templates[key]

    @staticmethod
    def create_cost(D_goal, omega, jit=True):
        """(cost, egrad, ehess) closures like riemannian_solver.py:77-119 (a @staticmethod there too:
        callable on the class), evaluated on the GPU.  The embedding dimension is taken from the
        first point the closures see (Y.shape[1]); templates are cached per process."""
        return _cost_closures(D_goal, omega, None, None, False)

    @staticmethod
    def create_cost_limits(D_goal, omega, psi_L, psi_U, jit=True):
        """riemannian_solver.py:121-176"""
        return _cost_closures(D_goal, omega, psi_L, psi_U, True)

    # -- solve ----------------------------------------------------------------------------------
    def solve(self, D_goal, omega, use_limits=False, bounds=None, Y_init=None, jit=True,
              output_log=True):
        """riemannian_solver.py:178-218.  D_goal / Y_init / bounds may carry a leading batch
        axis; the return value is then a dict of arrays instead of a single final_values dict."""
        omega = np.asarray(omega, dtype=float)
        D_goal = np.asarray(D_goal, dtype=float)
        batched = D_goal.ndim == 3
        if use_limits:
            psi_L, psi_U = self.graph.distance_bound_matrices()
        else:
            psi_L, psi_U = None, None
        if bounds is not None:  # bounds take precedence over Y_init (:197-198)
            lb, ub = np.asarray(bounds[0], dtype=float), np.asarray(bounds[1], dtype=float)
            if lb.ndim == 2:
                Y_init = dgp.generate_initialization((lb, ub), self.dim, omega)
            else:
                Y_init = dgp.generate_initialization_batch(lb, ub, self.dim, omega)
        elif Y_init is None:
            raise Exception("If not using bounds, provide an initialization!")
        T = self._template(omega, psi_L, psi_U, use_limits)
        t0 = time.time()
        res = T.solve(Y_init, T.targets_from_D(D_goal))
        torch.cuda.synchronize(T.device)
        dt = time.time() - t0
        x = res["x"].cpu().numpy()
        B = x.shape[0]
        info = {"x": x, "f(x)": res["f"].cpu().numpy(), "time": np.full(B, dt / B),
                "gradnorm": res["gradnorm"].cpu().numpy(),
                "iterations": res["iterations"].cpu().numpy(),
                "inner_iterations": res["inner_total"].cpu().numpy(),
                "stop": res["stop"].cpu().numpy()}
        if T.solver == "ConjugateGradient":      # pymanopt's final_values carry the last step size
            info["costevals"] = info.pop("inner_iterations")
            info["stepsize"] = res["stepsize"].cpu().numpy()
        if not batched:
            info = {k: (v[0] if k == "x" else v[0].item()) for k, v in info.items()}
            info["stop_reason"] = _STOP_REASONS[int(info["stop"])]
        return info if output_log else info["x"]


# ---------------------------------------------------------------------------------------------
class BatchProblem:
    """Goal-independent data of solve_with_riemannian for one problem graph, prepared once:
    edge template, which squared distances depend on the goal, anchors, limits."""

    def __init__(self, graph, use_limits=True, params=None, device=None, force_block_prepare=False,
                 host_only=False):
        """host_only: only the goal-independent host data (edge pattern, anchors, limits); no device
        handle is created (term-set construction can then be inspected without a GPU)."""
        self.force_block_prepare = force_block_prepare
        self.graph = graph
        self.robot = graph.robot
        self.dim = graph.dim
        self.use_limits = use_limits
        N = graph.number_of_nodes()
        n = self.robot.n
        self.end_effectors = list(self.robot.end_effectors)
        self.multi_ee = len(self.end_effectors) > 1
        ee = self.end_effectors[0]
        # goal nodes and how their positions follow from the goal pose (_pose_goal); a robot with
        # several end effectors (3-D trees) has a (p, q) pair per end effector and takes goals
        # [B, n_ee, 4, 4] in the order of robot.end_effectors
        if self.dim == 3:
            self.goal_nodes = [graph.index(c + e[1:]) for e in self.end_effectors for c in "pq"]
        else:
            # planar (graph_planar.py:136-145): a goal pose pins its end effector and the end effector's
            # parent; two end effectors of a tree may share that parent (one goal node then)
            self.goal_nodes, self._goal_src = [], []
            for e_i, e in enumerate(self.end_effectors):
                for is_parent, name in ((0, e), (1, self.robot.parent[e])):
                    if graph.index(name) not in self.goal_nodes:
                        self.goal_nodes.append(graph.index(name))
                        self._goal_src.append((e_i, is_parent, graph.dist[graph.index(self.robot.parent[e]),
                                                                            graph.index(e)]))
        self.anchor_nodes = [i for i, name in enumerate(graph.node_ids)
                             if POS in graph.nodes[name] and i not in self.goal_nodes]
        self.anchor_pos = np.array([graph.nodes[graph.node_ids[i]][POS] for i in self.anchor_nodes],
                                   dtype=float)
        # a goal instance with a dummy pose gives the complete edge pattern (omega)
        q0 = self.robot.zero_configuration()
        G0 = graph.from_pose({e: self.robot.pose(q0, e) for e in self.end_effectors})
        self.omega = dgp.adjacency_matrix_from_graph(G0)
        self.base_D = dgp.distance_matrix_from_graph(G0)
        self.base_lower = np.where(G0.edge, G0.lower, np.nan)
        self.base_upper = np.where(G0.edge, G0.upper, np.nan)
        if use_limits:
            self.psi_L, self.psi_U = graph.distance_bound_matrices()
        else:
            self.psi_L = self.psi_U = None
        self.N = N
        self.terms = build_terms(self.omega, self.psi_L, self.psi_U, use_limits)   # (i, j, kind, static target)
        if host_only:
            self.template, self.device_pipeline = None, False
            return
        self.template = Template.from_matrices(self.omega, self.psi_L, self.psi_U, k=self.dim,
                                               use_limits=use_limits, device=device, params=params)
        self._attach_device_pipeline()

    def _attach_device_pipeline(self):
        """Hand the goal-independent pre/post-processing data to the device handle."""
        g, T = self.graph, self.template
        n = self.robot.n
        if len(self.anchor_nodes) > 256 or self.N > (255 if self.dim == 3 else 128) or len(self.end_effectors) > 8:
            # beyond the device prepare / recover kernels (N <= 128, 3-D graphs 255; <= 8 end effectors):
            # host pre/post-processing around the device solve
            self.device_pipeline = False
            return
        # Device goal slots: two per end effector.  3-D: (p_e, q_e) = self.goal_nodes in order.  Planar (round 6: trees
        # too): (the end effector, its parent); a parent that an earlier end effector's pose pins already gets an inert
        # slot (-1) -- _pose_goal / BatchProblem keep the first definition.
        if self.dim == 3:
            dev_slots = list(self.goal_nodes)
        else:
            dev_slots, seen = [], set()
            for e in self.end_effectors:
                for name in (e, self.robot.parent[e]):
                    node = g.index(name)
                    dev_slots.append(node if node not in seen else -1)
                    seen.add(node)
            if any(dev_slots[2 * e] < 0 for e in range(len(self.end_effectors))):
                self.device_pipeline = False       # (an end effector that is another one's parent: host path)
                return
        goal_slot = {node: sl for sl, node in enumerate(dev_slots) if node >= 0}
        goalset = set(self.goal_nodes)
        slot = {a: s for s, a in enumerate(self.anchor_nodes)}
        G = len(dev_slots)                           # 2 per end effector
        # goal nodes of different end effectors that the goal graph ties by an equality edge
        live = sorted(goal_slot.values())
        pairs = [(a, b) for ia, a in enumerate(live) for b in live[ia + 1:]
                 if self.omega[dev_slots[a], dev_slots[b]] != 0 and np.isnan(g.dist[dev_slots[a], dev_slots[b]])]
        pair_slot = {(dev_slots[a], dev_slots[b]): q for q, (a, b) in enumerate(pairs)}
        term_src = np.full(T.T, -1, dtype=np.int32)
        for t in range(T.T):
            i, j = int(T.term_i[t]), int(T.term_j[t])
            if T.term_kind[t] != 1:
                continue
            for a, gnode in ((i, j), (j, i)):
                if gnode in goalset and a in slot:
                    term_src[t] = slot[a] * G + goal_slot[gnode]
            if (i, j) in pair_slot or (j, i) in pair_slot:
                term_src[t] = G * len(self.anchor_nodes) + pair_slot.get((i, j), pair_slot.get((j, i)))
        static = np.where(np.isnan(T.targets_static), self.base_D[T.term_i, T.term_j],
                          T.targets_static)
        lower = self.base_lower.copy()
        upper = self.base_upper.copy()
        for a in self.anchor_nodes:   # goal edges are re-created per goal on the device
            for gnode in self.goal_nodes:
                lower[a, gnode] = lower[gnode, a] = np.nan
                upper[a, gnode] = upper[gnode, a] = np.nan
        for a, b in pairs:
            ga, gb = dev_slots[a], dev_slots[b]
            lower[ga, gb] = lower[gb, ga] = upper[ga, gb] = upper[gb, ga] = np.nan
        I, J = np.nonzero(np.triu(self.omega))
        T0 = self.robot.T0_array()
        ee_path = np.full((len(self.end_effectors), n + 1), -1, dtype=np.int32)
        for e, ee in enumerate(self.end_effectors):
            path = [int(name[1:]) for name in self.robot.kinematic_map["p0"][ee]]
            ee_path[e, :len(path)] = path
        ee_len = None
        if self.dim == 3:
            p_idx = [g.index(f"p{i}") for i in range(n + 1)]
            q_idx = [g.index(f"q{i}") for i in range(n + 1)]
            along_z = 0                                   # one bit per end effector (:314)
            for e, ee in enumerate(self.end_effectors):
                path = [int(name[1:]) for name in self.robot.kinematic_map["p0"][ee]]
                rel_last = np.linalg.inv(T0[path[-2]]) @ T0[path[-1]]
                if np.linalg.norm(np.cross(rel_last[:3, 3], [0, 0, 1])) < 1e-10:
                    along_z |= 1 << e
            goal_len = g.axis_length
        else:
            p_idx = [g.index(f"p{i}") for i in range(n + 1)]
            q_idx = None
            along_z = 0
            goal_len = g.dist[self.goal_nodes[1], self.goal_nodes[0]]
            ee_len = [g.dist[g.index(self.robot.parent[e]), g.index(e)] for e in self.end_effectors]
        T.attach_pipeline(T0=T0, p_index=p_idx, q_index=q_idx, x_index=g.index("x"),
                          y_index=g.index("y"), axis_length=g.axis_length,
                          goal_nodes=self.goal_nodes, goal_len=goal_len, base_lower=lower,
                          base_upper=upper, anchor_index=self.anchor_nodes,
                          anchor_pos=self.anchor_pos, pair_i=I, pair_j=J, term_src=term_src,
                          term_static=static, last_link_along_z=along_z,
                          force_block_prepare=self.force_block_prepare,
                          ee_goal_nodes=dev_slots if self.multi_ee else None, ee_path=ee_path, ee_goal_len=ee_len,
                          goal_pair_a=[a for a, _ in pairs], goal_pair_b=[b for _, b in pairs])
        self.device_pipeline = True

    def goal_positions(self, T_goals):
        """[B,d+1,d+1] poses -> positions of the goal nodes [B,2,d]  (_pose_goal)."""
        T = np.asarray(T_goals, dtype=float)
        d = self.dim
        if d == 3:
            if T.ndim == 3:
                T = T[:, None]                           # one end effector
            p = T[:, :, :3, 3]
            q = p + T[:, :, :3, 2] * self.graph.axis_length
            return np.stack((p, q), axis=2).reshape(T.shape[0], -1, 3)   # p_e, q_e per end effector
        if T.ndim == 3:
            T = T[:, None]                               # one end effector
        return np.stack([T[:, e_i, :2, 2] - (T[:, e_i, :2, 0] * dist if is_parent else 0.0)
                         for e_i, is_parent, dist in self._goal_src], axis=1)

    def assemble(self, T_goals):
        """D_goal, LOWER, UPPER [B,N,N] for a batch of goals (from_pose + graph_complete_edges)."""
        gp = self.goal_positions(T_goals)
        B = gp.shape[0]
        D = np.broadcast_to(self.base_D, (B, self.N, self.N)).copy()
        lo = np.broadcast_to(self.base_lower, (B, self.N, self.N)).copy()
        up = np.broadcast_to(self.base_upper, (B, self.N, self.N)).copy()
        for gi, g in enumerate(self.goal_nodes):
            dist = np.linalg.norm(gp[:, gi, None, :] - self.anchor_pos[None], axis=-1)  # [B,A]
            for ai, a in enumerate(self.anchor_nodes):
                D[:, a, g] = D[:, g, a] = dist[:, ai] ** 2
                lo[:, a, g] = lo[:, g, a] = dist[:, ai]
                up[:, a, g] = up[:, g, a] = dist[:, ai]
            for hi in range(gi + 1, len(self.goal_nodes)):   # goal nodes of DIFFERENT end effectors
                h = self.goal_nodes[hi]
                if self.omega[g, h] != 0 and np.isnan(self.graph.dist[g, h]):
                    dd = np.linalg.norm(gp[:, gi] - gp[:, hi], axis=-1)
                    D[:, g, h] = D[:, h, g] = dd ** 2
                    lo[:, g, h] = lo[:, h, g] = dd
                    up[:, g, h] = up[:, h, g] = dd
        return D, lo, up

    def prepare(self, T_goals, chunk=None, workers=None):
        """Host pre-processing for a batch: targets [B,T] and Y_init [B,N,k].  Chunks of goals are
        smoothed and initialised on a thread pool (numpy releases the GIL in its loops and in
        LAPACK); for graphs the device prepare kernel covers (N <= 32) this path is only the mirror
        the tests compare it with."""
        import os
        from concurrent.futures import ThreadPoolExecutor
        D, lo, up = self.assemble(T_goals)
        B = D.shape[0]
        if chunk is None:
            chunk = 512 if self.N <= 32 else 8
        spans = [(s, min(s + chunk, B)) for s in range(0, B, chunk)]

        def one(span):
            lb, ub = dgp.floyd_warshall_bounds(lo[span[0]:span[1]], up[span[0]:span[1]])
            return dgp.generate_initialization_batch(lb, ub, self.dim, self.omega)

        workers = workers or min(len(spans), max(1, (os.cpu_count() or 1) // 2))
        if workers <= 1 or len(spans) == 1:
            Ys = [one(sp) for sp in spans]
        else:
            with ThreadPoolExecutor(workers) as ex:
                Ys = list(ex.map(one, spans))
        return self.targets_from_D(D), np.concatenate(Ys, axis=0)

    def targets_from_D(self, D):
        """[B,N,N] squared-distance matrices -> [B,T] per-term targets (Template.targets_from_D without
        a device handle)."""
        ti, tj, tk, tv = self.terms
        D = np.asarray(D, dtype=np.float64)
        tg = D[:, ti, tj].copy()
        hinge = tk != 1
        tg[:, hinge] = tv[hinge]
        return tg

    def joint_variables(self, Y, T_goals):
        from ..graphs.graph_revolute import joint_variables_revolute_batch
        from ..graphs.graph_planar import joint_variables_planar_batch
        if self.dim == 3:
            T = np.asarray(T_goals, dtype=float)
            if T.ndim == 4:
                T = {e: T[:, i] for i, e in enumerate(self.end_effectors)}
            return joint_variables_revolute_batch(self.graph, Y, T)
        return joint_variables_planar_batch(self.graph, Y)

    def pose_errors(self, q, T_goals):
        """EE position / rotation error of FK(q) against the goals (the metric of
        experiments/simple_ik_examples/test_chain_2d_new.py:62-66)."""
        T_goals = np.asarray(T_goals, dtype=float)
        if T_goals.ndim == 4:        # several end effectors: the worst of them
            errs = [BatchProblem._pose_err(self.robot.fk_batch(q, int(e[1:])), T_goals[:, i], self.dim)
                    for i, e in enumerate(self.end_effectors)]
            return np.max([e[0] for e in errs], axis=0), np.max([e[1] for e in errs], axis=0)
        return BatchProblem._pose_err(self.robot.fk_batch(q), T_goals, self.dim)

    @staticmethod
    def _pose_err(T_sol, T_goals, d):
        pos = np.linalg.norm(T_goals[:, :d, d] - T_sol[:, :d, d], axis=1)
        Rrel = T_goals[:, :d, :d] @ np.swapaxes(T_sol[:, :d, :d], 1, 2)
        if d == 3:
            c = np.clip(0.5 * np.trace(Rrel, axis1=1, axis2=2) - 0.5, -1.0, 1.0)
            rot = np.arccos(c)
        else:
            rot = np.abs(np.arctan2(Rrel[:, 1, 0], Rrel[:, 0, 0]))
        return pos, rot


class AnchoredProblem:
    """Opt-in "intended" obstacle semantics.  graph_base.py:182-211 ties every
    node with a known position -- base frame, goal nodes, obstacle centres -- to the others by
    equality edges and means to add robot<->obstacle lower-bound hinges (:205-211; the TYPE
    comparison at :207 never fires, so the reference creates none, and that observable behaviour
    stays the default everywhere else in this package).  Here those nodes are CONSTANTS instead of
    rows of Y and the hinges exist: UR10 + table_environment() becomes a 10-node problem with 100
    point-to-obstacle hinges per p-node on the wavefront kernel, instead of N = 116 / 5612 terms on
    the workgroup kernel.  The anchors fix the gauge, so the search space is Euclidean.

    graph: a ProblemGraphRevolute (chain) with its obstacles added (either spelling of
    add_spherical_obstacle).  The initial point is the robot graph's own (bound smoothing + MDS
    without obstacles) fitted to the world frame by its anchors."""

    def __init__(self, graph, params=None, device=None, host_only=False):
        """host_only: derive the term set only (no device handles) -- what tests/test_host_layer.py
        compares with the reference's own edge construction (tests/golden/ur10_table_intended.npz)."""
        import copy
        from ..utils.constants import OBSTACLE, ROBOT, TYPE, MAIN_PREFIX
        from .. import _ffi
        if graph.dim != 3 or not graph.robot.is_chain:
            raise NotImplementedError("the fixed-anchor formulation covers 3-D chains")
        self.graph, self.robot = graph, graph.robot
        obstacles = [n for n in graph.node_ids if graph.nodes[n].get(TYPE) == OBSTACLE]
        bare = copy.deepcopy(graph)
        bare.clear_obstacles()
        self.base = BatchProblem(bare, use_limits=True, params=params, device=device, host_only=host_only)
        if not host_only and not self.base.device_pipeline:
            raise NotImplementedError("robot graph beyond the device pipeline")
        bp = self.base
        N = bp.N
        goal = list(bp.goal_nodes)
        anchors = list(bp.anchor_nodes) + goal               # constant rows first, goal rows last
        free = [i for i in range(N) if i not in anchors]
        fidx = {n: f for f, n in enumerate(free)}
        om, pL, pU, D = bp.omega, bp.psi_L, bp.psi_U, bp.base_D
        sub = np.ix_(free, free)
        ti, tj, tk, tv = build_terms(om[sub], pL[sub], pU[sub], True)
        Dff = D[sub]
        target = np.where(np.isnan(tv), Dff[ti, tj], tv)
        pin = []
        for i in free:
            for r, a in enumerate(anchors):
                if om[i, a] != 0:
                    pin.append((fidx[i], r, _ffi.TERM_EQ, D[i, a]))
                if pL[i, a] != 0:
                    pin.append((fidx[i], r, _ffi.TERM_LOWER, pL[i, a]))
                if pU[i, a] != 0:
                    pin.append((fidx[i], r, _ffi.TERM_UPPER, pU[i, a]))
        self.obstacles = np.array([[*np.asarray(graph.nodes[o]["pos"], dtype=float), float(graph.nodes[o]["radius"])]
                                   for o in obstacles], dtype=float).reshape(-1, 4)
        self.obstacle_names = obstacles
        obs = self.obstacles.copy()
        obs[:, 3] = obs[:, 3] ** 2                            # LOWER = radius  ->  psi_L = radius^2
        names = [bare.node_ids[i] for i in free]
        # graph_base.py:205-211 as written: every node whose TYPE holds ROBOT and whose name starts
        # with MAIN_PREFIX gets [BELOW], LOWER = radius towards the obstacle -- pinned to the
        # reference's own lines by tools/capture_golden_intended.py (700 edges for UR10 + table: p0..p6
        # x 100; p0 and p6 are constants here, so 5 x 100 hinges act on unknowns)
        mask = [int(n[0] == MAIN_PREFIX and ROBOT in bare.nodes[n].get(TYPE, [])) for n in names]
        pos = np.zeros((len(anchors), 3))
        pos[:len(bp.anchor_nodes)] = bp.anchor_pos
        self.free, self.anchors, self.pin = free, anchors, pin
        self.free_names = names
        self.free_terms = (ti, tj, tk, target)
        self.obs_mask = np.array(mask, dtype=np.int32)
        if host_only:
            self.template = None
            return
        self.template = Template(
            len(free), 3, ti, tj, tk, None, device=device, params=params,
            anchored=dict(anchor_pos=pos, n_goal_anchor=len(goal), term_target=target,
                          pin_node=[p[0] for p in pin], pin_anchor=[p[1] for p in pin],
                          pin_kind=[p[2] for p in pin], pin_target=[p[3] for p in pin],
                          obs=obs, obs_node_mask=mask, full_N=N, free_full_index=free,
                          anchor_full_index=anchors, axis_length=graph.axis_length))

    def goal_anchors(self, T_goals):
        """[B,4,4] -> [B, 2*3]: p_n, q_n world positions (graph_revolute.py:243-249)."""
        return self.base.goal_positions(T_goals).reshape(len(T_goals), -1)

    def solve(self, T_goals):
        """Goal poses -> dict of device tensors (x [B, N_robot, 3], q, pos_err, rot_err, stats)."""
        return self.template.anchored_ik(self.base.template, np.asarray(T_goals, dtype=float))

    def clearance(self, Y_full, include_goal=False):
        """min over (p-node of the robot, obstacle) of |p - centre| - radius per goal (>= 0:
        collision free in the sense of graph_base.py:205-211).  Y_full [B, N_robot, 3] (numpy).
        The end effector p_n sits where the goal puts it (a constant of the problem), so it only
        counts with include_goal=True."""
        g = self.base.graph
        pn = [g.index(f"p{i}") for i in range(1, self.robot.n + (1 if include_goal else 0))]
        P = np.asarray(Y_full)[:, pn]                                            # [B, n, 3]
        d = np.linalg.norm(P[:, :, None, :] - self.obstacles[None, None, :, :3], axis=-1)
        return (d - self.obstacles[None, None, :, 3]).min(axis=(1, 2))


# BatchProblem objects (device handles) of recent solve_batch / solve_with_riemannian calls.  The
# reference re-reads the graph on every call (riemannian_solver.py:220-234), so the key is the
# CONTENT a BatchProblem is built from -- edge pattern, distances, limits, anchor positions, robot
# frames -- not the identity of the graph object: a graph mutated in place (clear_obstacles +
# add_spherical_obstacle, set_limits, ...) gets a fresh problem.  Least recently used entries are
# dropped (their device handles are freed by Template.__del__).
_PROBLEM_CACHE = collections.OrderedDict()
_PROBLEM_CACHE_MAX = 8


def graph_fingerprint(graph):
    """Content hash of everything BatchProblem reads from a problem graph."""
    h = hashlib.blake2b(digest_size=16)
    h.update(repr((type(graph).__name__, graph.dim, graph.number_of_nodes(), tuple(graph.node_ids),
                   float(getattr(graph, "axis_length", 0.0)))).encode())
    for a in (graph.edge, graph.dist, graph.lower, graph.upper):
        h.update(np.ascontiguousarray(a).tobytes())
    if hasattr(graph, "bounded"):
        h.update(repr(graph.bounded).encode() if not isinstance(graph.bounded, np.ndarray)
                 else np.ascontiguousarray(graph.bounded).tobytes())
    for name in graph.node_ids:
        pos = graph.nodes[name].get(POS)
        h.update(b"-" if pos is None else np.asarray(pos, dtype=float).tobytes())
    h.update(np.ascontiguousarray(graph.robot.T0_array()).tobytes())
    lb, ub = graph.robot.limits_arrays()
    h.update(np.asarray(lb, dtype=float).tobytes() + np.asarray(ub, dtype=float).tobytes())
    return h.hexdigest()


def _problem_for(graph, use_limits=True, params=None, device=None):
    key = (graph_fingerprint(graph), bool(use_limits),
           None if not params else tuple(sorted(params.items())), None if device is None else str(device))
    prob = _PROBLEM_CACHE.get(key)
    if prob is None:
        prob = _PROBLEM_CACHE[key] = BatchProblem(graph, use_limits, params, device)
        while len(_PROBLEM_CACHE) > _PROBLEM_CACHE_MAX:
            _PROBLEM_CACHE.popitem(last=False)
    else:
        _PROBLEM_CACHE.move_to_end(key)
        prob.graph = graph      # same content, possibly another object: recover through the caller's
    return prob


def clear_problem_cache():
    _PROBLEM_CACHE.clear()
    _closure_templates.clear()


def solve_batch(graph, T_goals, use_limits=True, params=None, device=None, Y_init=None):
    """Batched solve_with_riemannian.  T_goals: [B,d+1,d+1] array or list of poses.
    Returns (q [B,n], Y [B,N,k], info dict of arrays)."""
    T = np.stack([as_matrix(t) for t in T_goals]) if not isinstance(T_goals, np.ndarray) \
        else np.asarray(T_goals, dtype=float)
    prob = _problem_for(graph, use_limits, params, device)
    t0 = time.time()
    if prob.device_pipeline and Y_init is None:
        # everything on the device: prepare -> solve -> recover (gik_ik_batch)
        res = prob.template.ik(T)
        torch.cuda.synchronize(prob.template.device)
        dt = time.time() - t0
        Y = res["x"].cpu().numpy()
        q = res["q"].cpu().numpy()
        pos, rot = res["pos_err"].cpu().numpy(), res["rot_err"].cpu().numpy()
    else:
        targets, Y0 = prob.prepare(T)
        if Y_init is not None:
            Y0 = np.asarray(Y_init, dtype=float)
        t0 = time.time()
        res = prob.template.solve(Y0, targets)
        torch.cuda.synchronize(prob.template.device)
        dt = time.time() - t0
        Y = res["x"].cpu().numpy()
        q = prob.joint_variables(Y, T)
        pos, rot = prob.pose_errors(q, T)
    info = {"x": Y, "f(x)": res["f"].cpu().numpy(), "gradnorm": res["gradnorm"].cpu().numpy(),
            "iterations": res["iterations"].cpu().numpy(),
            "inner_iterations": res["inner_total"].cpu().numpy(),
            "stop": res["stop"].cpu().numpy(), "time": np.full(len(Y), dt / max(len(Y), 1)),
            "solve_time": dt, "pos_err": pos, "rot_err": rot}
    return q, Y, info


def solve_with_riemannian(graph, T_goal, use_jit=True, jit=None):
    """riemannian_solver.py:220-234 on the GPU engine (B = 1).  `jit=` is accepted as an alias of
    `use_jit=` because the reference's README spells it that way (README.md:45)."""
    if isinstance(T_goal, dict):     # several end effectors: {end effector: pose}
        T = np.stack([as_matrix(T_goal[e]) for e in graph.robot.end_effectors])[None]
    else:
        T = as_matrix(T_goal)[None]
    q, Y, info = solve_batch(graph, T)
    q_sol = graph.robot.array_to_q(q[0])
    broken = graph.check_distance_limits(graph.realization(q_sol), tol=1e-6)
    if len(broken) > 0:
        return None, None
    return q_sol, Y[0]
