#!/usr/bin/env python3
"""Wavefront-kernel variants that do not fit the register file without scratch (<3,10> 48 B, <3,20>
796 B, <2,16> 168 B per lane) against the workgroup kernels on graphs that select them: the 3-D
tree of tests/golden/tree5.npz (13 terms at its busiest node), the planar trees (8-9 terms), and
LWA4D with one extra hinge at its busiest node (10 terms).  Run on the GPU box:
    python tools/gpu_variants.py            -> solves/s per (graph, kernel path)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from conftest import make_graph, planar_tree
from test_host_layer import tree_robot
from graphik_amd.engine import Template, build_terms
from graphik_amd.solvers.riemannian_solver import BatchProblem

def bench(tag, prob, Tg, B):
    targets, Y0 = prob.prepare(Tg)                     # host prepare (trees beyond the device pipeline too)
    ti, tj, tk, tv = build_terms(prob.omega, prob.psi_L, prob.psi_U, True)
    for path, params in (("wave", {}), ("block", {"force_block_path": 1})):
        T = Template.from_matrices(prob.omega, prob.psi_L, prob.psi_U, k=prob.dim, use_limits=True, params=params)
        tg = torch.from_numpy(np.asarray(targets)).cuda(); y = torch.from_numpy(Y0).cuda()
        r = T.solve(y, tg); torch.cuda.synchronize()
        t0 = time.time(); r = T.solve(y, tg); torch.cuda.synchronize(); dt = time.time() - t0
        print(f"{tag:28s} {path:5s} maxdeg {T.maxdeg:2d} is_block {T.info['is_block']} B={B}: {B / dt:9.0f} solves/s "
              f"({dt * 1e3:.1f} ms; median its {int(np.median(r['iterations'].cpu().numpy()))}, "
              f"maxiter frac {(r['stop'].cpu().numpy() == 1).mean():.3f})", flush=True)

B = 4096
rng = np.random.RandomState(0)
robot, graph = tree_robot()
lb, ub = robot.limits_arrays()
Q = lb + (ub - lb) * rng.rand(B, robot.n)
Tg = np.stack([[robot.pose(robot.array_to_q(q), ee).as_matrix() for ee in robot.end_effectors] for q in Q])
bench("tree5 (3-D, 2 EE)", BatchProblem(graph, use_limits=True, host_only=True), Tg, B)
for w in ("y5", "bin2"):
    robot, graph = planar_tree(w)
    lb, ub = robot.limits_arrays()
    Q = lb + (ub - lb) * rng.rand(B, robot.n)
    Tg = np.stack([[robot.pose(robot.array_to_q(q), ee).as_matrix() for ee in robot.end_effectors] for q in Q])
    bench(f"planar tree {w}", BatchProblem(graph, use_limits=True, host_only=True), Tg, B)
robot, graph = make_graph("lwa4d")
lb, ub = robot.limits_arrays()
Tg = robot.fk_batch(lb + (ub - lb) * rng.rand(B, robot.n))
prob = BatchProblem(graph, use_limits=True, host_only=True)
bench("lwa4d (9 terms)", prob, Tg, B)
ti, tj, tk, tv = build_terms(prob.omega, prob.psi_L, prob.psi_U, True)
deg = np.bincount(np.concatenate([ti, tj]), minlength=prob.N)
i = int(np.argmax(deg)); j = next(j for j in range(prob.N) if j != i and prob.omega[i, j] == 0 and prob.psi_L[i, j] == 0 and prob.psi_U[i, j] == 0)
prob.psi_U = prob.psi_U.copy(); prob.psi_U[i, j] = prob.psi_U[j, i] = 100.0     # an upper hinge that never binds
prob.terms = build_terms(prob.omega, prob.psi_L, prob.psi_U, True)
bench("lwa4d + 1 inert hinge (10)", prob, Tg, B)
