"""dev: time slicing (long problems re-queued every GIK_SLICE outer iterations) gives bit-identical results to
running every problem to completion in one go (GIK_SLICE=0)."""
import sys, os, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if len(sys.argv) > 2 and sys.argv[1] == "child":
    import torch
    from graphik_amd.utils.roboturdf import load_schunk_lwa4d, load_kuka
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    from graphik_amd.engine import Template
    out = {}
    for name, ld, B in (("lwa4d", load_schunk_lwa4d, 6000), ("kuka", load_kuka, 3000)):
        robot, graph = ld()
        prob = BatchProblem(graph, use_limits=True)
        rng = np.random.RandomState(1)
        Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(B, robot.n))
        targets, Y0 = prob.template.prepare(torch.from_numpy(Tg).cuda())
        torch.cuda.synchronize(); t0 = time.time()
        r = prob.template.solve(Y0, targets, trace_cap=32); torch.cuda.synchronize()
        out[name + "_ms"] = (time.time() - t0) * 1e3
        out[name] = {k: r[k].cpu().numpy() for k in ("x", "f", "gradnorm", "iterations", "inner_total", "stop", "n_accept", "inner_executed")}
        out[name]["numit"] = r["trace"]["numit"].cpu().numpy()
        if name == "lwa4d":
            tb = Template.from_matrices(prob.omega, prob.psi_L, prob.psi_U, k=3, use_limits=True, params={"force_block_path": 1})
            t0 = time.time(); rb = tb.solve(Y0[:700], targets[:700]); torch.cuda.synchronize()
            out["lwa4d_block_ms"] = (time.time() - t0) * 1e3
            out["lwa4d_block"] = {k: rb[k].cpu().numpy() for k in ("x", "f", "iterations", "inner_total", "inner_executed", "stop")}
    np.save(sys.argv[2], out, allow_pickle=True)
else:
    env = dict(os.environ); env.pop("GIK_SLICE", None)
    subprocess.check_call(["timeout", "240", sys.executable, __file__, "child", "/tmp/sl_a.npy"], env=env)
    env["GIK_SLICE"] = "0"
    subprocess.check_call(["timeout", "240", sys.executable, __file__, "child", "/tmp/sl_b.npy"], env=env)
    a = np.load("/tmp/sl_a.npy", allow_pickle=True).item(); b = np.load("/tmp/sl_b.npy", allow_pickle=True).item()
    for name in a:
        if name.endswith("_ms"):
            print(name, "sliced %.1f ms, unsliced %.1f ms" % (a[name], b[name])); continue
        bad = [k for k in a[name] if not np.array_equal(a[name][k], b[name][k], equal_nan=True)]
        print(name, "bit-identical" if not bad else "DIFFERS in %s" % bad, "| maxiter problems", int((a[name]["stop"] == 1).sum()))
