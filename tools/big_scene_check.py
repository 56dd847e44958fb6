"""A scene beyond 128 nodes (round 5): UR10 + table_environment(n_width=12, n_height=14) = 200 spheres, N = 216 --
the reference takes any number of spheres (graph_base.py:182-211); until round 5 gik_template_create refused N > 128.
Known answers of the four-wavefront node-per-lane kernel against the CPU oracle, trajectories of a few goals,
a batch through solve_batch (since round 6 prepare / recover run on the device too: prep_block_kernel<false, 256>), throughput.
    python tools/big_scene_check.py [B]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from graphik_amd.utils import table_environment
from graphik_amd.utils.roboturdf import load_ur10
from graphik_amd.solvers.riemannian_solver import BatchProblem, solve_batch
from oracle import c_oracle as co

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
robot, graph = load_ur10()
for idx, obs in enumerate(table_environment(n_width=12, n_height=14)):
    graph.add_spherical_obstacle(f"o{idx}", obs[0], obs[1])
N = graph.number_of_nodes()
prob = BatchProblem(graph, use_limits=True)
T = prob.template
print("N", N, "terms", T.T, "info", {k: T.info[k] for k in ("is_block", "node_per_lane", "n_clique", "n_slot_terms", "lds_bytes", "waves_per_cu", "has_pipeline")}, flush=True)
assert N == 216 and T.info["node_per_lane"] == 4 and T.info["n_clique"] == 206
rs = np.random.RandomState(0)
Tg = robot.fk_batch(-np.pi + 2 * np.pi * rs.rand(B, robot.n))
t0 = time.time(); targets, Y0 = prob.prepare(Tg[:4]); print("host prepare of 4 goals %.2f s" % (time.time() - t0), flush=True)
D, _, _ = prob.assemble(Tg[:4])
om, pL, pU = prob.omega, prob.psi_L, prob.psi_U
inds = co.limit_inds(om, pL, pU)
# known answers at random points and near the start points
for scale in (1.0, 1e-3):
    Y = np.asarray(Y0[:2]) + scale * rs.randn(2, N, 3)
    W = rs.randn(2, N, 3)
    c, g, h = T.cost(Y, targets[:2]).cpu().numpy(), T.grad(Y, targets[:2]).cpu().numpy(), T.hess(Y, W, targets[:2]).cpu().numpy()
    for m in range(2):
        rc, rg, rh = co.lcost(Y[m], D[m], om, pL, pU, inds), co.lgrad(Y[m], D[m], om, pL, pU, inds), co.lhess(Y[m], W[m], D[m], om, pL, pU, inds)
        e = (abs(c[m] - rc) / abs(rc), np.abs(g[m] - rg).max() / np.abs(rg).max(), np.abs(h[m] - rh).max() / np.abs(rh).max())
        print("KAT scale %g goal %d: cost %.1e grad %.1e hess %.1e" % (scale, m, *e), flush=True)
        assert max(e) < 1e-12, e
P = T.proj(Y, W).cpu().numpy()
print("proj horizontal:", float(np.abs(Y[0].T @ P[0] - P[0].T @ Y[0]).max()))
# trajectories of 2 goals against the oracle
r = T.solve(Y0[:2], targets[:2], trace_cap=8)
for gidx in range(2):
    o = co.rtr_solve(np.asarray(Y0[gidx]), D[gidx], om, pL, pU, True, traj_cap=8)
    print("goal", gidx, "numit gpu", r["trace"]["numit"][gidx].cpu().numpy()[:6].tolist(), "oracle", o["traj"]["numit"][:6].tolist(),
          "its", int(r["iterations"][gidx]), o["iterations"], "f", float(r["f"][gidx]), o["f(x)"], flush=True)
    assert np.array_equal(r["trace"]["numit"][gidx].cpu().numpy()[:4], o["traj"]["numit"][:4])
    assert (float(r["f"][gidx]) < 1e-9) == (o["f(x)"] < 1e-9)
# the batch through the drop-in entry point
t0 = time.time()
q, Yb, info = solve_batch(graph, Tg, use_limits=True)
dt = time.time() - t0
Ts = robot.fk_batch(q)
pos = np.linalg.norm(Ts[:, :3, 3] - Tg[:, :3, 3], axis=1)
print("solve_batch %d goals: %.2f s wall (device prepare + solve + recover: gik_ik_batch), success %.3f, median pos err %.2e, outer its median %d max %d"
      % (B, dt, float(np.mean(pos < 0.01)), float(np.median(pos)), int(np.median(info["iterations"])), int(np.max(info["iterations"]))), flush=True)
# device solve alone
targets, Y0 = prob.prepare(Tg)
for rep in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); rr = T.solve(Y0, targets); e1.record(); torch.cuda.synchronize()
print("device solve of %d goals: %.1f ms -> %.0f solves/s; Hessian products %d" % (B, e0.elapsed_time(e1), B / e0.elapsed_time(e1) * 1e3, int(rr["inner_total"].sum())))
