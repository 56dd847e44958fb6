"""Planar prepare kernel, four goals per wavefront against one (GIK_NO_PREP_QUAD): time per 65536 goals, agreement
of targets (bitwise), MDS column counts and initial points (Y_init itself and its Gram matrix)."""
import os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
    import numpy as np, torch
    from conftest import make_graph
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    name, B = sys.argv[3], int(sys.argv[4])
    robot, graph = make_graph(name)
    prob = BatchProblem(graph, use_limits=not name.endswith("nolimits"))
    rs = np.random.RandomState(0)
    lb, ub = robot.limits_arrays()
    Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rs.rand(B, robot.n))).cuda()
    tpl = prob.template
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.time()
        tg, Y0, K = tpl.prepare(Tg, return_K=True)
        torch.cuda.synchronize(); dt = time.time() - t0
    np.savez(sys.argv[2], tg=tg.cpu().numpy(), Y0=Y0.cpu().numpy(), K=K.cpu().numpy(), ms=dt * 1e3)
    sys.exit(0)
import numpy as np
for name, B in (("planar10_limits_pi", 65536), ("planar10_nolimits", 4099), ("planar10_limits_halfpi", 3)):
    outs = {}
    for tag, env in (("quad", {}), ("wave", {"GIK_NO_PREP_QUAD": "1"})):
        f = f"/tmp/prep_quad_ab_{tag}.npz"
        subprocess.run([sys.executable, __file__, "child", f, name, str(B)], check=True, env=dict(os.environ, **env))
        outs[tag] = dict(np.load(f))
    a, b = outs["quad"], outs["wave"]
    print(f"{name} B={B}: prepare quad %.2f ms, wave %.2f ms (host wall)" % (a["ms"], b["ms"]))
    print("  targets bitwise equal:", np.array_equal(a["tg"], b["tg"]), " finite:", np.isfinite(a["Y0"]).all())
    dY = np.abs(a["Y0"] - b["Y0"]).reshape(len(a["Y0"]), -1).max(1)
    print("  Y_init: bitwise equal on %d of %d goals; |diff| median %.1e, max %.1e" % ((dY == 0).sum(), len(dY), np.median(dY), dY.max()))
    Ga, Gb = a["Y0"] @ a["Y0"].transpose(0, 2, 1), b["Y0"] @ b["Y0"].transpose(0, 2, 1)
    err = np.abs(Ga - Gb).reshape(len(Ga), -1).max(1) / np.abs(Gb).reshape(len(Gb), -1).max(1)
    print("  Gram(Y_init) relative difference: median %.1e, max %.1e; goals above 1e-9: %d" % (np.median(err), err.max(), (err > 1e-9).sum()))
    print("  K equal on %d of %d goals; K median %d" % ((a["K"] == b["K"]).sum(), len(a["K"]), np.median(a["K"])), flush=True)
