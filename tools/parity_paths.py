"""Parity residuals by kernel path (round 4): for the three 3-D arms and the solve
kernels -- wavefront (one unknown per lane: the default per-edge product form and, "wave_column", the column form with per-slot Hessian rows cached per tCG solve),
workgroup (terms recomputed from the point rows), node-per-lane (s = y . w formed once per edge) --
  (a) finals on the golden goals: median / upper-quartile max |dq| against the reference's numpy path,
      next to the oracle's and the reference's own two paths;
  (b) effort on random goals from the same start points: Hessian products and outer iterations against
      the CPU oracle.
Writes gpurun_out/parity_paths.json.   python tools/parity_paths.py [B]"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from conftest import load_golden, make_graph
from parity_util import wrap_abs
from oracle import c_oracle as co
from graphik_amd.engine import Template
from graphik_amd.graphs.graph_revolute import joint_variables_revolute_batch
from graphik_amd.solvers.riemannian_solver import BatchProblem

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
PATHS = {"wave": {}, "wave_column": {"hessian_form": "column"}, "block": {"force_block_path": 1}, "npt": {"force_block_path": 2}}
out = {}
for name in ("kuka", "lwa4d", "ur10"):
    d = load_golden(name)
    robot, graph = make_graph(name)
    conv = d["f_sol"] < 1e-9
    o = co.rtr_solve_batch(d["Y_init"], d["D_goal"], d["omega"], d["psi_L"], d["psi_U"], True, fast=False)
    dq_orc = wrap_abs(joint_variables_revolute_batch(graph, o["x"], d["T_goal"]) - d["q_sol"]).max(axis=1)
    dq_ref = wrap_abs(d["q_sol"] - d["loop_q_sol"]).max(axis=1)
    both = conv & (d["loop_f_sol"] < 1e-9)
    rec = {"finals": {"reference_pair": [float(np.percentile(dq_ref[both], q)) for q in (50, 75)],
                      "oracle": [float(np.percentile(dq_orc[conv], q)) for q in (50, 75)]}, "effort": {}}
    # effort: random goals, oracle from the device's start points
    prob = BatchProblem(graph, use_limits=True)
    rng = np.random.RandomState(3)
    Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(B, robot.n))
    targets, Y0 = prob.prepare(Tg)
    D, _, _ = prob.assemble(Tg)
    oo = co.rtr_solve_batch(np.asarray(Y0), D, prob.omega, prob.psi_L, prob.psi_U, True, fast=False)
    for path, params in PATHS.items():
        T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True, params=params)
        r = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"]))
        q = joint_variables_revolute_batch(graph, r["x"].cpu().numpy(), d["T_goal"])
        dq = wrap_abs(q - d["q_sol"]).max(axis=1)
        rec["finals"][path] = [float(np.percentile(dq[conv], q_)) for q_ in (50, 75)]
        Tp = Template.from_matrices(prob.omega, prob.psi_L, prob.psi_U, k=3, use_limits=True, params=params)
        rr = Tp.solve(Y0, targets)
        ms = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); Tp.solve(Y0, targets); e1.record(); torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        hv, its = rr["inner_total"].cpu().numpy().astype(np.int64), rr["iterations"].cpu().numpy()
        ex = rr["inner_executed"].cpu().numpy().astype(np.int64)
        same = (its < 3000) == (oo["iterations"] < 3000)
        rec["effort"][path] = {"hv_ratio": float(hv.sum() / oo["inner_total"].sum()),
                               "hv_executed_ratio": float(ex.sum() / oo["inner_total"].sum()),
                               "median_its": [float(np.median(its)), float(np.median(oo["iterations"]))],
                               "p90_its": [float(np.percentile(its, 90)), float(np.percentile(oo["iterations"], 90))],
                               "same_convergence_class": float(same.mean()), "solve_ms": float(min(ms))}
    out[name] = rec
    print(name, json.dumps(rec), flush=True)
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
json.dump({"goals_effort": B, "results": out}, open(os.path.join(R, "gpurun_out", "parity_paths.json"), "w"), indent=1)
