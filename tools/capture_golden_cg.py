#!/usr/bin/env python3
"""Golden vectors of the reference's ConjugateGradient option (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden_cg.py

Runs the reference's RiemannianSolver(graph, {"solver": "ConjugateGradient"}).solve(...)
(graphik/solvers/riemannian_solver.py:51-59, :178-218) from the initial points already captured in
tests/golden/<scenario>.npz and records, per goal: every iteration's cost, gradient norm, step size
and line-search cost evaluations (first MAX_TRAJ iterations), the final point, cost, gradient norm,
iteration count and stopping reason.  pymanopt is not installed: its ConjugateGradient and
LineSearchAdaptive are the restatements under tools/ref_shims/pymanopt/solvers (see the README
there), so these vectors pin "reference source + those restatements".  Only numbers are written.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "ref_shims"))

import refcompat  # noqa: E402
import numpy as np  # noqa: E402
from graphik.utils.roboturdf import load_schunk_lwa4d, load_ur10  # noqa: E402

refcompat.patch_skew()
import graphik.solvers.riemannian_solver as rs  # noqa: E402
import graphik.solvers.costs as costs  # noqa: E402
from graphik.solvers.riemannian_solver import RiemannianSolver  # noqa: E402
from graphik.utils.utils import list_to_variable_dict  # noqa: E402
from graphik.robots.robot_planar import RobotPlanar  # noqa: E402
from graphik.graphs.graph_planar import ProblemGraphPlanar  # noqa: E402

for _n in ("jcost", "jgrad", "jhess", "lcost", "lgrad", "lhess"):
    setattr(rs, _n, getattr(costs, _n))

MAX_TRAJ = 64
GOLD = os.path.join(REPO, "tests", "golden")
STOP = {"max iterations": 1, "min grad norm": 0, "min stepsize": 3, "max time": 4}


def planar_chain(lim):
    n = 10
    lims = lim * np.ones(n) if np.isscalar(lim) else np.asarray(lim, dtype=float)
    robot = RobotPlanar({"link_lengths": list_to_variable_dict(np.ones(n)),
                         "theta": list_to_variable_dict(np.zeros(n)),
                         "joint_limits_upper": list_to_variable_dict(lims),
                         "joint_limits_lower": list_to_variable_dict(-lims), "num_joints": n})
    return robot, ProblemGraphPlanar(robot)


def run(name, graph, goals, params):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    use_limits = bool(int(d["use_limits"]))
    out = {k: [] for k in ("goal", "x", "f", "gradnorm", "iterations", "stop", "stepsize", "t_solve",
                           "traj_f", "traj_gradnorm", "traj_stepsize", "traj_costevals")}
    for g in goals:
        solver = RiemannianSolver(graph, dict(params, solver="ConjugateGradient"))
        cg = solver.solver
        steps, evals = [], []
        ls = cg._linesearch
        orig = type(ls).search

        def search(self, *a, **k):
            s, newx = orig(self, *a, **k)
            steps.append(s)
            evals.append(self.last_cost_evaluations)
            return s, newx

        type(ls).search = search
        try:
            t0 = time.time()
            cg._logverbosity = 2
            info = solver.solve(d["D_goal"][g], d["omega"], use_limits=use_limits,
                                Y_init=d["Y_init"][g].copy(), jit=False)
            dt = time.time() - t0
        finally:
            type(ls).search = orig
        log = cg._optlog
        reason = log["stoppingreason"]
        code = next(v for k, v in STOP.items() if k in reason)
        it = log["iterations"]
        m = min(MAX_TRAJ, len(steps))

        def pad(a, dtype=float):
            p = np.full(MAX_TRAJ, np.nan if dtype is float else -9, dtype=dtype)
            p[:m] = a[:m]
            return p

        out["goal"].append(g)
        out["x"].append(info["x"])
        out["f"].append(float(info["f(x)"]))
        out["gradnorm"].append(float(info["gradnorm"]))
        out["iterations"].append(int(info["iterations"]))
        out["stepsize"].append(float(info.get("stepsize", np.nan)))
        out["stop"].append(code)
        out["t_solve"].append(dt)
        out["traj_f"].append(pad(np.array(it["f(x)"], dtype=float)))          # cost BEFORE step k
        out["traj_gradnorm"].append(pad(np.array(it["gradnorm"], dtype=float)))
        out["traj_stepsize"].append(pad(np.array(steps, dtype=float)))
        out["traj_costevals"].append(pad(np.array(evals), dtype=np.int32))
        print(f"  {name} goal {g}: it={info['iterations']} f={info['f(x)']:.2e} |g|={info['gradnorm']:.2e} "
              f"stop={code} ({reason[:40]}) t={dt:.1f}s", flush=True)
    return {f"{name}__{k}": np.array(v) for k, v in out.items()}


if __name__ == "__main__":
    data = {}
    params = {}
    cap = {"maxiter": 2000}          # bounded run time for the 3-D arms (CG needs > 1e4 iterations there)
    data.update(run("planar10_nolimits", planar_chain(np.pi)[1], range(6), params))
    data.update(run("planar10_limits_halfpi", planar_chain(np.array(9 * [np.pi / 2] + [np.pi]))[1], range(6), params))
    data.update(run("lwa4d", load_schunk_lwa4d()[1], range(4), cap))
    data.update(run("ur10", load_ur10()[1], range(4), cap))
    data["maxiter_3d"] = np.int64(cap["maxiter"])
    path = os.path.join(GOLD, "cg.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
