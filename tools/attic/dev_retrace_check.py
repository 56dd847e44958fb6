"""dev: the checkpoint resume after rejected steps gives bit-identical results to rerunning tCG (GIK_DBG=16)."""
import sys, os, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from graphik_amd.utils.roboturdf import load_schunk_lwa4d, load_kuka, load_ur10
    from graphik_amd.solvers.riemannian_solver import BatchProblem
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
    np.save(sys.argv[2], out, allow_pickle=True)
else:
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, "child", "/tmp/retrace_on.npy"], env=env)
    env["GIK_DBG"] = "16"
    subprocess.check_call([sys.executable, __file__, "child", "/tmp/retrace_off.npy"], env=env)
    a = np.load("/tmp/retrace_on.npy", allow_pickle=True).item(); b = np.load("/tmp/retrace_off.npy", allow_pickle=True).item()
    for name in a:
        same = all(np.array_equal(a[name][k], b[name][k], equal_nan=True) for k in a[name] if k != "inner_executed")
        print(name, "bit-identical:", same, "| products executed %.4g of %.4g counted (%.1f %% saved); without: %.4g" % (
            a[name]["inner_executed"].sum(), a[name]["inner_total"].sum(),
            100 * (1 - a[name]["inner_executed"].sum() / a[name]["inner_total"].sum()), b[name]["inner_executed"].sum()))
        if not same:
            for k in a[name]:
                if k != "inner_executed" and not np.array_equal(a[name][k], b[name][k], equal_nan=True):
                    d = np.nonzero(np.any(np.reshape(a[name][k] != b[name][k], (1024, -1)), axis=1))[0]
                    print("   differs:", k, "problems", d[:10], "of", len(d))
