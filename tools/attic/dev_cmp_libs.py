"""dev: two builds of the library give bit-identical solver outputs (GIK_LIB_PATH=<other build> vs the default)."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if len(sys.argv) > 2 and sys.argv[1] == "child":
    import torch
    from graphik_amd.utils.roboturdf import load_schunk_lwa4d, load_kuka, load_ur10
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    from graphik_amd.engine import Template
    out = {}
    for name, ld in (("lwa4d", load_schunk_lwa4d), ("kuka", load_kuka), ("ur10", load_ur10)):
        robot, graph = ld()
        prob = BatchProblem(graph, use_limits=True)
        rng = np.random.RandomState(1)
        Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(1024, robot.n))
        targets, Y0 = prob.prepare(Tg)
        r = prob.template.solve(Y0, targets, trace_cap=64); torch.cuda.synchronize()
        out[name] = {k: r[k].cpu().numpy() for k in ("x", "f", "gradnorm", "iterations", "inner_total", "stop", "n_accept", "inner_executed")}
        out[name]["numit"] = r["trace"]["numit"].cpu().numpy(); out[name]["tstop"] = r["trace"]["stop"].cpu().numpy()
        if name == "lwa4d":     # workgroup-per-problem path and a small maxinner on the same problems
            tb = Template.from_matrices(prob.omega, prob.psi_L, prob.psi_U, k=3, use_limits=True, params={"force_block_path": 1})
            rb = tb.solve(Y0[:64], targets[:64]); torch.cuda.synchronize()
            out["lwa4d_block"] = {k: rb[k].cpu().numpy() for k in ("x", "f", "iterations", "inner_total", "inner_executed")}
            tm = Template.from_matrices(prob.omega, prob.psi_L, prob.psi_U, k=3, use_limits=True, params={"maxinner": 37, "maxiter": 200})
            rm = tm.solve(Y0[:256], targets[:256]); torch.cuda.synchronize()
            out["lwa4d_maxinner37"] = {k: rm[k].cpu().numpy() for k in ("x", "f", "iterations", "inner_total", "inner_executed")}
    np.save(sys.argv[2], out, allow_pickle=True)
else:
    other = sys.argv[1]
    env = dict(os.environ); env.pop("GIK_LIB_PATH", None)
    subprocess.check_call([sys.executable, __file__, "child", "/tmp/cmp_a.npy"], env=env)
    env["GIK_LIB_PATH"] = other
    subprocess.check_call([sys.executable, __file__, "child", "/tmp/cmp_b.npy"], env=env)
    a = np.load("/tmp/cmp_a.npy", allow_pickle=True).item(); b = np.load("/tmp/cmp_b.npy", allow_pickle=True).item()
    for name in a:
        bad = [k for k in a[name] if not np.array_equal(a[name][k], b[name][k], equal_nan=True)]
        print(name, "bit-identical" if not bad else "DIFFERS in %s" % bad, "| executed", a[name]["inner_executed"].sum(), "vs", b[name]["inner_executed"].sum())
