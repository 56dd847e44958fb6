"""The KUKA tail (p90 of the outer iterations, share of goals at maxiter, Hessian products) of the wavefront kernel's
per-edge product form against the CPU oracle from the same start points, for several libraries (GIK_LIB_PATH) and/or
kernel paths -- the measurement behind NOTEBOOK 11.2.

    python tools/tail_bisect.py [B] [--libs a.so b.so ...] [--paths wave_per_edge block npt ...] [--robot kuka]

The oracle runs once (cached in /tmp); every library is loaded in a child process of its own."""
import hashlib, json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np

PATHS = {"wave_column": {"hessian_form": 0}, "wave_per_edge": {"hessian_form": 1}, "block": {"force_block_path": 1},
         "npt": {"force_block_path": 2}}


def problem(robot_name, B):
    from conftest import make_graph
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph(robot_name)
    prob = BatchProblem(graph, use_limits=True)
    rng = np.random.RandomState(3)
    Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(B, robot.n))
    targets, Y0 = prob.prepare(Tg)
    D, _, _ = prob.assemble(Tg)
    return prob, targets, Y0, D


def child(robot_name, B, path, cache):
    import torch
    from graphik_amd.engine import Template
    prob, targets, Y0, D = problem(robot_name, B)
    o = np.load(cache)
    T = Template.from_matrices(prob.omega, prob.psi_L, prob.psi_U, k=3, use_limits=True, params=PATHS[path])
    r = T.solve(Y0, targets)
    its, hv = r["iterations"].cpu().numpy(), r["inner_total"].cpu().numpy().astype(np.int64)
    oi = o["iterations"]
    mx, mxo = its >= 3000, oi >= 3000
    rec = {"lib": os.path.basename(os.environ.get("GIK_LIB_PATH", "product")), "path": path,
           "p90": float(np.percentile(its, 90)), "p90_oracle": float(np.percentile(oi, 90)),
           "maxiter": int(mx.sum()), "maxiter_oracle": int(mxo.sum()),
           "to_maxiter": int((mx & ~mxo).sum()), "from_maxiter": int((~mx & mxo).sum()),
           "same_class": float(np.mean(mx == mxo)), "hv_ratio": float(hv.sum() / o["inner_total"].sum()),
           "hv_ratio_converged_both": float(hv[~mx & ~mxo].sum() / o["inner_total"][~mx & ~mxo].sum()),
           "median": [float(np.median(its)), float(np.median(oi))],
           "sha": hashlib.sha256(its.tobytes() + hv.tobytes()).hexdigest()[:12]}
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--child":
        child(a[1], int(a[2]), a[3], a[4])
        sys.exit(0)
    B = int(a[0]) if a and a[0].isdigit() else 512
    libs = a[a.index("--libs") + 1:] if "--libs" in a else [None]
    libs = [l for l in libs if not (isinstance(l, str) and l.startswith("--"))] or [None]
    if "--paths" in a:
        i = a.index("--paths")
        paths = [p for p in a[i + 1:] if not p.startswith("--") and not p.endswith(".so")]
    else:
        paths = ["wave_per_edge"]
    robot_name = a[a.index("--robot") + 1] if "--robot" in a else "kuka"
    cache = f"/tmp/tail_oracle_{robot_name}_{B}.npz"
    if not os.path.exists(cache):
        from oracle import c_oracle as co
        prob, targets, Y0, D = problem(robot_name, B)
        o = co.rtr_solve_batch(np.asarray(Y0), D, prob.omega, prob.psi_L, prob.psi_U, True, fast=False)
        np.savez(cache, iterations=o["iterations"], inner_total=o["inner_total"])
    for lib in libs:
        for path in paths:
            env = dict(os.environ)
            if lib:
                env["GIK_LIB_PATH"] = os.path.abspath(lib)
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child", robot_name, str(B), path, cache], env=env)
