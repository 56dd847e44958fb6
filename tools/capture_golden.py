#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden.py [scenario ...]

Imports /root/reference (utiasSTARS/GraphIK) with the third-party stand-ins of tools/ref_shims
(see its README) and records, per scenario (robot + goal set):

  * the problem-graph *template* (node order, edge attribute matrices, psi_L/psi_U, zero-config
    frames T0, joint limits)                      -> pins graphik_amd.graphs / robots
  * per goal: q_goal, T_goal, D_goal, omega, bound_smoothing (lb, ub), Y_init
                                                  -> pins host/device pre-processing
  * kernel known-answer vectors: cost / egrad / ehess / proj on random (Y, W), limits and
    no-limits, numpy closures and costs.py loops  -> pins oracle + HIP kernels
  * trust-region trajectory prefix from Y_init (f, |grad|, Delta, inner iterations, tCG stop
    reason, accept flag per outer iteration)      -> pins oracle RTR/tCG + HIP solve kernel
  * finals: Y_sol, f, gradnorm, iterations, q_sol, EE position / rotation error.

Only data (numbers) is written; no reference source text is stored.  The fixtures are small
compressed .npz files; this script is committed next to them so they can be regenerated.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "ref_shims"))

import refcompat  # noqa: E402  (must precede graphik imports)
import numpy as np  # noqa: E402
from graphik.utils.roboturdf import load_schunk_lwa4d, load_ur10, load_kuka  # noqa: E402

refcompat.patch_skew()

import graphik.solvers.riemannian_solver as rs  # noqa: E402
import graphik.solvers.costs as costs  # noqa: E402
from graphik.solvers.riemannian_solver import RiemannianSolver  # noqa: E402
from graphik.utils.manifolds.fixed_rank_psd_sym import PSDFixedRank  # noqa: E402
from graphik.utils.dgp import (  # noqa: E402
    adjacency_matrix_from_graph, bound_smoothing, distance_matrix_from_graph, graph_from_pos,
    pos_from_graph)
from graphik.utils.utils import table_environment, list_to_variable_dict  # noqa: E402
from graphik.utils.constants import DIST, LOWER, UPPER, BOUNDED, BELOW, ABOVE  # noqa: E402
from graphik.robots.robot_planar import RobotPlanar  # noqa: E402
from graphik.graphs.graph_planar import ProblemGraphPlanar  # noqa: E402

# the AOT module `costgrd` does not exist here: expose the same loops under the names
# riemannian_solver.py:18-21 would have imported (they run as plain Python under the numba shim)
for _n in ("jcost", "jgrad", "jhess", "lcost", "lgrad", "lhess"):
    setattr(rs, _n, getattr(costs, _n))

OUT = os.path.join(REPO, "tests", "golden")
MAX_TRAJ = 48  # outer iterations of trajectory prefix kept


# ------------------------------------------------------------------------------------------
def bounded_code(data):
    """0: key absent or [] ; 1: [False] ; 2: below ; 3: above ; 4: [None]"""
    if BOUNDED not in data:
        return 0
    b = data[BOUNDED]
    if isinstance(b, str):
        return 2 if b == BELOW else (3 if b == ABOVE else 0)
    if len(b) == 0:
        return 0
    if b[0] is False:
        return 1
    if b[0] == BELOW:
        return 2
    if b[0] == ABOVE:
        return 3
    return 4


def template_arrays(graph, robot):
    ids = graph.node_ids
    N = len(ids)
    idx = {n: i for i, n in enumerate(ids)}
    Gd = np.full((N, N), np.nan)
    Gl = np.full((N, N), np.nan)
    Gu = np.full((N, N), np.nan)
    Gb = np.full((N, N), -1, dtype=np.int8)
    for u, v, d in graph.edges(data=True):
        i, j = idx[u], idx[v]
        for M, key in ((Gd, DIST), (Gl, LOWER), (Gu, UPPER)):
            if key in d:
                M[i, j] = M[j, i] = d[key]
        Gb[i, j] = Gb[j, i] = bounded_code(d)
    psi_L, psi_U = graph.distance_bound_matrices()
    n = robot.n
    dim = robot.dim
    T0 = np.stack([robot.nodes[f"p{i}"]["T0"].as_matrix() for i in range(n + 1)])
    lbq = np.array([robot.lb[f"p{i}"] for i in range(1, n + 1)], dtype=float)
    ubq = np.array([robot.ub[f"p{i}"] for i in range(1, n + 1)], dtype=float)
    pos = np.full((N, dim), np.nan)
    for name, data in graph.nodes(data=True):
        if "pos" in data:
            pos[idx[name]] = data["pos"]
    return dict(node_ids=np.array(ids), dim=dim, n_joints=n, axis_length=float(graph.axis_length),
                G_dist=Gd, G_lower=Gl, G_upper=Gu, G_bounded=Gb, psi_L=psi_L, psi_U=psi_U,
                T0=T0, lb_q=lbq, ub_q=ubq, anchor_pos=pos)


class Recorder:
    """Hooks into one TrustRegions instance; records the outer-iteration trajectory."""

    def __init__(self, tr):
        self.tr = tr
        self.rows = []
        self._orig_tcg = tr._truncated_conjugate_gradient
        self._orig_chk = tr._check_stopping_criterion
        tr._truncated_conjugate_gradient = self.tcg
        tr._check_stopping_criterion = self.chk
        self.prev_x = None
        self.hv = 0

    def tcg(self, problem, x, fgradx, eta, Delta, theta, kappa, mininner, maxinner):
        if self.rows:
            self.rows[-1]["accept"] = int(x is not self.prev_x)
        self.prev_x = x
        eta, Heta, numit, stop = self._orig_tcg(problem, x, fgradx, eta, Delta, theta, kappa,
                                                mininner, maxinner)
        self.hv += numit + 1
        self.rows.append(dict(Delta=Delta, numit=numit, stop=stop, fx_before=float(problem.cost(x)),
                              accept=-1))
        return eta, Heta, numit, stop

    def chk(self, time0, **kw):
        self.rows[-1]["gradnorm"] = float(kw.get("gradnorm"))
        return self._orig_chk(time0, **kw)

    def restore(self):
        self.tr._truncated_conjugate_gradient = self._orig_tcg
        self.tr._check_stopping_criterion = self._orig_chk

    def arrays(self, x_final):
        if self.rows:
            self.rows[-1]["accept"] = int(x_final is not self.prev_x)
        r = self.rows[:MAX_TRAJ]
        return dict(
            traj_Delta=np.array([a["Delta"] for a in r]),
            traj_numit=np.array([a["numit"] for a in r], dtype=np.int32),
            traj_stop=np.array([a["stop"] for a in r], dtype=np.int32),
            traj_f_before=np.array([a["fx_before"] for a in r]),
            traj_gradnorm_after=np.array([a["gradnorm"] for a in r]),
            traj_accept=np.array([a["accept"] for a in r], dtype=np.int32),
        )


def ee_errors(robot, graph, q_sol, T_goal):
    ee = f"p{robot.n}"
    T_sol = robot.pose(q_sol, ee)
    pos = float(np.linalg.norm(T_goal.trans - T_sol.trans))
    if robot.dim == 3:
        rot = float(np.linalg.norm(T_goal.rot.dot(T_sol.rot.inv()).log()))
    else:
        rot = float(abs((T_goal.dot(T_sol.inv())).log()[2]))
    return pos, rot


def solve_one(robot, graph, T_goal, use_limits, jit, record=True, Y_init_in=None):
    """Mirror of solve_with_riemannian (riemannian_solver.py:220-234) with recording hooks."""
    G = graph.from_pose(T_goal)
    solver = RiemannianSolver(graph)
    D_goal = distance_matrix_from_graph(G)
    omega = adjacency_matrix_from_graph(G)
    t0 = time.time()
    lb, ub = bound_smoothing(G)
    t_bs = time.time() - t0
    if use_limits:
        psi_L, psi_U = graph.distance_bound_matrices()
    else:
        psi_L, psi_U = 0 * omega, 0 * omega
    # same call RiemannianSolver.solve makes (:197-198); deterministic, so calling it here and
    # passing the result as Y_init (bounds=None) reproduces solve(bounds=(lb,ub)) exactly
    Y_init = RiemannianSolver.generate_initialization((lb, ub), graph.dim, omega, psi_L, psi_U)
    rec = Recorder(solver.solver) if record else None
    t0 = time.time()
    info = solver.solve(D_goal, omega, use_limits=use_limits, Y_init=Y_init.copy(), jit=jit)
    t_solve = time.time() - t0
    out = dict(D_goal=D_goal, omega=omega, lb=lb, ub=ub, Y_init=Y_init, Y_sol=info["x"],
               f_sol=float(info["f(x)"]), gradnorm=float(info["gradnorm"]),
               iterations=int(info["iterations"]), t_solve=t_solve, t_bs=t_bs)
    if rec:
        out.update(rec.arrays(info["x"]))
        out["hv_total"] = rec.hv
        rec.restore()
    G_sol = graph_from_pos(info["x"], graph.node_ids)
    q_sol = graph.joint_variables(G_sol, {f"p{robot.n}": T_goal})
    out["q_sol"] = np.array([q_sol[f"p{i}"] for i in range(1, robot.n + 1)], dtype=float)
    out["pos_err"], out["rot_err"] = ee_errors(robot, graph, q_sol, T_goal)
    broken = graph.check_distance_limits(graph.realization(q_sol), tol=1e-6)
    out["n_broken"] = len(broken)
    return out


def kernel_kats(graph, D_goal, omega, seed, M=6):
    """cost/egrad/ehess/proj known answers on random (Y, W)."""
    rng = np.random.RandomState(seed)
    N = omega.shape[0]
    k = graph.dim
    psi_L, psi_U = graph.distance_bound_matrices()
    fn = {}
    fn["nolim_np"] = RiemannianSolver.create_cost(D_goal, omega, jit=False)
    fn["nolim_loop"] = RiemannianSolver.create_cost(D_goal, omega, jit=True)
    fn["lim_np"] = RiemannianSolver.create_cost_limits(D_goal, omega, psi_L, psi_U, jit=False)
    fn["lim_loop"] = RiemannianSolver.create_cost_limits(D_goal, omega, psi_L, psi_U, jit=True)
    Ys, Ws = [], []
    res = {f"kat_{k_}_{q}": [] for k_ in fn for q in ("cost", "grad", "hess")}
    projs = []
    for m in range(M):
        scale = [1.0, 0.3, 2.0][m % 3]
        Y = scale * rng.randn(N, k)
        W = rng.randn(N, k)
        Ys.append(Y)
        Ws.append(W)
        for k_, (c, g, h) in fn.items():
            res[f"kat_{k_}_cost"].append(float(c(Y)))
            res[f"kat_{k_}_grad"].append(np.asarray(g(Y), dtype=float))
            res[f"kat_{k_}_hess"].append(np.asarray(h(Y, W), dtype=float))
        projs.append(PSDFixedRank.proj(Y, W))
    out = {k_: np.array(v) for k_, v in res.items()}
    out.update(kat_Y=np.array(Ys), kat_W=np.array(Ws), kat_proj=np.array(projs))
    inds = np.nonzero(np.triu(omega) + np.triu((psi_L != psi_U) * (psi_L > 0))
                      + np.triu((psi_L != psi_U) * (psi_U > 0)))
    out["inds_limits"] = np.array(inds, dtype=np.int32)
    out["inds_nolimits"] = np.array(np.nonzero(np.triu(omega)), dtype=np.int32)
    return out


def T_of(T):
    return T.as_matrix()


def run_scenario(name, robot, graph, seeds, use_limits=True, traj_goals=4, loop_goals=2,
                 kat=True):
    print(f"== {name}: N={graph.number_of_nodes()} dim={graph.dim}", flush=True)
    data = template_arrays(graph, robot)
    data["use_limits"] = int(use_limits)
    per_goal = {}

    def add(key, val):
        per_goal.setdefault(key, []).append(val)

    first = None
    for gi, seed in enumerate(seeds):
        np.random.seed(seed)
        q_goal = robot.random_configuration()
        T_goal = robot.pose(q_goal, f"p{robot.n}")
        t0 = time.time()
        r = solve_one(robot, graph, T_goal, use_limits, jit=False, record=True)
        print(f"  goal {gi} seed {seed}: it={r['iterations']} hv={r['hv_total']} f={r['f_sol']:.2e} "
              f"|g|={r['gradnorm']:.2e} pos={r['pos_err']:.2e} rot={r['rot_err']:.2e} "
              f"t={time.time()-t0:.1f}s", flush=True)
        if first is None:
            first = r
        add("seed", seed)
        add("q_goal", np.array([q_goal[f"p{i}"] for i in range(1, robot.n + 1)]))
        add("T_goal", T_of(T_goal))
        for key in ("D_goal", "lb", "ub", "Y_init", "Y_sol", "f_sol", "gradnorm", "iterations",
                    "q_sol", "pos_err", "rot_err", "hv_total", "n_broken", "t_solve", "t_bs"):
            add(key, r[key])
        if gi < traj_goals:
            for key in [k_ for k_ in r if k_.startswith("traj_")]:
                pad = np.full(MAX_TRAJ, np.nan) if r[key].dtype.kind == "f" else \
                    np.full(MAX_TRAJ, -9, dtype=np.int32)
                pad[: len(r[key])] = r[key]
                add("np_" + key, pad)
        if gi < loop_goals:
            # same goal through the costs.py loops (the path the AOT module would run)
            rl = solve_one(robot, graph, T_goal, use_limits, jit=True, record=True)
            print(f"     loops: it={rl['iterations']} hv={rl['hv_total']} f={rl['f_sol']:.2e} "
                  f"pos={rl['pos_err']:.2e} max|dq|={np.max(np.abs(rl['q_sol']-r['q_sol'])):.2e}",
                  flush=True)
            for key in ("Y_sol", "f_sol", "gradnorm", "iterations", "q_sol", "pos_err", "rot_err",
                        "hv_total"):
                add("loop_" + key, rl[key])
            for key in [k_ for k_ in rl if k_.startswith("traj_")]:
                pad = np.full(MAX_TRAJ, np.nan) if rl[key].dtype.kind == "f" else \
                    np.full(MAX_TRAJ, -9, dtype=np.int32)
                pad[: len(rl[key])] = rl[key]
                add("loop_" + key, pad)
    data["omega"] = first["omega"]
    for k_, v in per_goal.items():
        data[k_] = np.array(v)
    if kat:
        data.update(kernel_kats(graph, first["D_goal"], first["omega"], seed=1234))
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **data)
    print(f"  wrote {path} ({os.path.getsize(path)/1024:.1f} KiB)", flush=True)


def planar_chain(n=10, lim=np.pi):
    a = list_to_variable_dict(np.ones(n))
    th = list_to_variable_dict(np.zeros(n))
    if np.isscalar(lim):
        lims = lim * np.ones(n)
    else:
        lims = np.asarray(lim, dtype=float)
    params = {"link_lengths": a, "theta": th,
              "joint_limits_upper": list_to_variable_dict(lims),
              "joint_limits_lower": list_to_variable_dict(-lims), "num_joints": n}
    robot = RobotPlanar(params)
    graph = ProblemGraphPlanar(robot)
    return robot, graph


def host_kats():
    """Host-layer known answers: table_environment(), FK of given q, realization round trip."""
    out = {}
    obs = table_environment()
    out["table_centers"] = np.array([o[0] for o in obs])
    out["table_radii"] = np.array([o[1] for o in obs])
    for nm, ld in (("lwa4d", load_schunk_lwa4d), ("ur10", load_ur10), ("kuka", load_kuka)):
        robot, graph = ld()
        rng = np.random.RandomState(7)
        Q = rng.uniform(-np.pi, np.pi, size=(5, robot.n))
        Ts, Ps, Qr = [], [], []
        for q in Q:
            qd = {f"p{i+1}": q[i] for i in range(robot.n)}
            Ts.append(np.stack([robot.pose(qd, f"p{i}").as_matrix() for i in range(1, robot.n + 1)]))
            G = graph.realization(qd)
            Ps.append(pos_from_graph(G, graph.node_ids))
            qr = graph.joint_variables(G, {f"p{robot.n}": robot.pose(qd, f"p{robot.n}")})
            Qr.append([qr[f"p{i}"] for i in range(1, robot.n + 1)])
        out[f"{nm}_fk_q"] = Q
        out[f"{nm}_fk_T"] = np.array(Ts)
        out[f"{nm}_realization"] = np.array(Ps)
        out[f"{nm}_jointvars"] = np.array(Qr)
    for nm, ld in (("lwa4d", load_schunk_lwa4d), ("ur10", load_ur10), ("kuka", load_kuka)):
        np.random.seed(5)                       # randomized_links (roboturdf.py:236-244)
        robot, _ = ld(randomized_links=True, randomize_percentage=0.3)
        out[f"{nm}_randomized_T0"] = np.stack([robot.nodes[f"p{i}"]["T0"].as_matrix() for i in range(robot.n + 1)])
    robot, graph = planar_chain(10)
    rng = np.random.RandomState(7)
    Q = rng.uniform(-np.pi, np.pi, size=(5, 10))
    Ts, Ps, Qr = [], [], []
    for q in Q:
        qd = {f"p{i+1}": q[i] for i in range(10)}
        Ts.append(np.stack([robot.pose(qd, f"p{i}").as_matrix() for i in range(1, 11)]))
        G = graph.realization(qd)
        Ps.append(pos_from_graph(G, graph.node_ids))
        qr = graph.joint_variables(G, {"p10": robot.pose(qd, "p10")})
        Qr.append([qr[f"p{i}"] for i in range(1, 11)])
    out["planar10_fk_q"] = Q
    out["planar10_fk_T"] = np.array(Ts)
    out["planar10_realization"] = np.array(Ps)
    out["planar10_jointvars"] = np.array(Qr)
    path = os.path.join(OUT, "host_kats.npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {path} ({os.path.getsize(path)/1024:.1f} KiB)", flush=True)


SCENARIOS = {}


def scenario(f):
    SCENARIOS[f.__name__] = f
    return f


@scenario
def host():
    host_kats()


@scenario
def lwa4d():
    robot, graph = load_schunk_lwa4d()
    run_scenario("lwa4d", robot, graph, seeds=list(range(16)), traj_goals=16, loop_goals=16)


@scenario
def ur10():
    robot, graph = load_ur10()
    run_scenario("ur10", robot, graph, seeds=list(range(12)), traj_goals=12, loop_goals=12)


@scenario
def kuka():
    robot, graph = load_kuka()
    run_scenario("kuka", robot, graph, seeds=list(range(12)), traj_goals=12, loop_goals=12)


@scenario
def planar10():
    robot, graph = planar_chain(10, np.pi)
    run_scenario("planar10_nolimits", robot, graph, seeds=list(range(21, 37)), use_limits=False,
                 traj_goals=8, loop_goals=4)
    run_scenario("planar10_limits_pi", robot, graph, seeds=list(range(21, 29)), use_limits=True,
                 traj_goals=4, loop_goals=2)
    robot, graph = planar_chain(10, np.array(9 * [np.pi / 2] + [np.pi]))
    run_scenario("planar10_limits_halfpi", robot, graph, seeds=list(range(22, 38)), use_limits=True,
                 traj_goals=8, loop_goals=4)


@scenario
def ur10_table():
    robot, graph = load_ur10()
    for idx, obs in enumerate(table_environment()):
        graph.add_spherical_obstacle(f"o{idx}", obs[0], obs[1])
    run_scenario("ur10_table", robot, graph, seeds=list(range(8)), traj_goals=8, loop_goals=0)


if __name__ == "__main__":
    todo = sys.argv[1:] or list(SCENARIOS)
    for s in todo:
        SCENARIOS[s]()
