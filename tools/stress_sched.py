"""Stress of the wavefront kernel's schedulers (round-robin slicing, tail spreading): batch sizes around
the number of resident waves, slices from 1 iteration up, several batches in flight on one handle; every
result must equal the plain run (debug_flags = 512) bit for bit and nothing may hang (run under timeout).
    timeout 600 python tools/stress_sched.py            (on the GPU box)
    STRESS_MAXITER=3000 STRESS_BATCHES="8192 16384" STRESS_SLICES="1 7 64" ...   a longer soak"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["GIK_SLICE_CYCLES"] = "0"
from graphik_amd.solvers.riemannian_solver import BatchProblem
from graphik_amd.utils.roboturdf import load_schunk_lwa4d, load_ur10, load_kuka
keys = ("x", "f", "gradnorm", "iterations", "inner_total", "stop", "n_accept", "stepsize")
t_start = time.time()
n_runs = 0
for name, load in (("kuka", load_kuka), ("ur10", load_ur10), ("lwa4d", load_schunk_lwa4d)):
    robot, graph = load()
    rs = np.random.RandomState(1)
    lb, ub = robot.limits_arrays()
    Tg_all = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rs.rand(16384, robot.n))).cuda()
    MAXITER = int(os.environ.get("STRESS_MAXITER", "600"))
    plain = BatchProblem(graph, use_limits=True, params={"debug_flags": 512, "maxiter": MAXITER})
    waves = None
    for B in [int(b) for b in os.environ.get("STRESS_BATCHES", "2049 2100 2560 4096 4097 6000 16384").split()]:
        Tg = Tg_all[:B]
        tg, Y0 = plain.template.prepare(Tg)
        ref = plain.template.solve(Y0, tg)
        torch.cuda.synchronize()
        ref = {k: ref[k].cpu().numpy() for k in keys}
        for sl in [int(x) for x in os.environ.get("STRESS_SLICES", "1 3 17 64 256").split()]:
            prob = BatchProblem(graph, use_limits=True, params={"slice_outer_its": sl, "maxiter": MAXITER})
            streams = [torch.cuda.Stream() for _ in range(3)]
            outs = []
            for s in streams:             # three batches in flight on one handle
                with torch.cuda.stream(s):
                    outs.append(prob.template.solve(Y0, tg))
            torch.cuda.synchronize()
            for r in outs:
                for k in keys:
                    assert np.array_equal(r[k].cpu().numpy(), ref[k], equal_nan=True), (name, B, sl, k)
            n_runs += 3
            ho = int((outs[0]["flags"].cpu().numpy() >> 8).sum())
            print(f"{name} B={B} slice={sl}: ok, hand-overs {ho}, {time.time() - t_start:.0f} s", flush=True)
print("stress passed:", n_runs, "launches")
