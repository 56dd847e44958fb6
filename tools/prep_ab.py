"""Table-scene prepare kernel with and without the range compression (GIK_PREP_NO_COMPRESS): time per 4096
goals, agreement of targets / bounds (bitwise) and of the initial points (Gram matrices), MDS column counts."""
import os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
    import numpy as np, torch
    from conftest import make_graph
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph("ur10_table")
    prob = BatchProblem(graph, use_limits=True)
    rs = np.random.RandomState(0)
    lb, ub = robot.limits_arrays()
    Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rs.rand(4096, robot.n))).cuda()
    tpl = prob.template
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        tg, Y0, K = tpl.prepare(Tg, return_K=True)
        torch.cuda.synchronize(); dt = time.time() - t0
    dbg = tpl.prepare_debug(Tg[:64])
    np.savez(sys.argv[2], tg=tg.cpu().numpy(), Y0=Y0.cpu().numpy(), K=K.cpu().numpy(), ms=dt * 1e3,
             eig=dbg["eig"].cpu().numpy())
    sys.exit(0)
import numpy as np
outs = {}
for tag, env in (("compressed", {}), ("full", {"GIK_PREP_NO_COMPRESS": "1"})):
    f = f"/tmp/prep_ab_{tag}.npz"
    subprocess.run([sys.executable, __file__, "child", f], check=True, env=dict(os.environ, **env))
    outs[tag] = dict(np.load(f))
a, b = outs["compressed"], outs["full"]
print("prepare 4096 table-scene goals: compressed %.1f ms, full Jacobi %.1f ms" % (a["ms"], b["ms"]))
print("targets bitwise equal:", np.array_equal(a["tg"], b["tg"]))
Ga, Gb = a["Y0"] @ a["Y0"].transpose(0, 2, 1), b["Y0"] @ b["Y0"].transpose(0, 2, 1)
err = np.abs(Ga - Gb).reshape(len(Ga), -1).max(1) / np.abs(Gb).reshape(len(Gb), -1).max(1)
print("Gram(Y_init) relative difference: median %.1e, max %.1e; goals above 1e-8: %d" % (np.median(err), err.max(), (err > 1e-8).sum()))
print("K equal on %d of %d goals; K compressed median %d, full median %d" % ((a["K"] == b["K"]).sum(), len(a["K"]), np.median(a["K"]), np.median(b["K"])))
ea, eb = np.sort(a["eig"][:, 0], 1), np.sort(b["eig"][:, 0], 1)
print("Gram spectrum (64 goals): max |diff| / max |ev| = %.1e" % (np.abs(ea - eb).max() / np.abs(eb).max()))
