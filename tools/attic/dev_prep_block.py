"""dev: the workgroup-per-goal prepare kernel against the wave kernel (same graph, forced) and against the host
mirror on the table scene."""
import sys, os, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from graphik_amd.utils.roboturdf import load_schunk_lwa4d, load_ur10
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    out = {}
    for name, ld in (("lwa4d", load_schunk_lwa4d), ("ur10", load_ur10)):
        robot, graph = ld()
        prob = BatchProblem(graph, use_limits=True)
        rng = np.random.RandomState(2)
        Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(300, robot.n))
        tg, Y0 = prob.template.prepare(torch.from_numpy(Tg).cuda())
        torch.cuda.synchronize()
        out[name] = (tg.cpu().numpy(), Y0.cpu().numpy())
    np.save(sys.argv[2], out, allow_pickle=True)
    sys.exit(0)
env = dict(os.environ)
subprocess.check_call([sys.executable, __file__, "child", "/tmp/prep_wave.npy"], env=env)
env["GIK_PREP_FORCE_BLOCK"] = "1"
subprocess.check_call([sys.executable, __file__, "child", "/tmp/prep_block.npy"], env=env)
a = np.load("/tmp/prep_wave.npy", allow_pickle=True).item(); b = np.load("/tmp/prep_block.npy", allow_pickle=True).item()
for k in a:
    print(k, "targets identical", np.array_equal(a[k][0], b[k][0]), "| Y_init identical", np.array_equal(a[k][1], b[k][1]), "max |dY|", np.abs(a[k][1] - b[k][1]).max())
# table scene: device (block kernel) vs host mirror with canonical signs
import torch
from graphik_amd.utils.roboturdf import load_ur10
from graphik_amd.utils import table_environment, dgp
from graphik_amd.solvers.riemannian_solver import BatchProblem
robot, graph = load_ur10()
for idx, obs in enumerate(table_environment()):
    graph.add_spherical_obstacle(f"o{idx}", obs[0], obs[1])
prob = BatchProblem(graph, use_limits=True)
print("table scene: device pipeline", prob.device_pipeline)
rng = np.random.RandomState(0)
lb, ub = robot.limits_arrays()
Tg = robot.fk_batch(lb + (ub - lb) * rng.rand(64, robot.n))
torch.cuda.synchronize(); t0 = time.time()
tg, Y0 = prob.template.prepare(torch.from_numpy(Tg).cuda()); torch.cuda.synchronize()
print("device prepare of 64 goals: %.3f s" % (time.time() - t0))
D, lo, up = prob.assemble(Tg)
lbm, ubm = dgp.floyd_warshall_bounds(lo, up)
Yh = dgp.generate_initialization_batch(lbm, ubm, 3, prob.omega, canonical=True)
tgh = prob.template.targets_from_D(D)
Yd = Y0.cpu().numpy().reshape(Yh.shape)
G_d = Yd @ Yd.transpose(0, 2, 1); G_h = Yh @ Yh.transpose(0, 2, 1)
print("targets max rel diff %.2e" % (np.abs(tg.cpu().numpy() - np.asarray(tgh)) / (1e-30 + np.abs(np.asarray(tgh)))).max())
print("Y_init: max |Y_d - Y_h| %.2e ; Gram max rel diff %.2e (per goal max: %s)" % (
    np.abs(Yd - Yh).max(), np.abs(G_d - G_h).max() / np.abs(G_h).max(), np.array2string(np.abs(G_d - G_h).reshape(64, -1).max(axis=1)[:8], precision=2)))
for B in (256, 1024):
    Tg = robot.fk_batch(lb + (ub - lb) * rng.rand(B, robot.n)); Tgd = torch.from_numpy(Tg).cuda()
    torch.cuda.synchronize(); t0 = time.time(); prob.template.prepare(Tgd); torch.cuda.synchronize()
    print("device prepare B=%d: %.3f s -> %.2f ms/goal" % (B, time.time() - t0, 1e3 * (time.time() - t0) / B))
