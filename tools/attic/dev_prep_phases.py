"""dev: cumulative time of the phases of the workgroup-per-goal prepare kernel (developer build:
GIK_LIB_PATH=graphik_amd/lib/exp/libgraphik_amd_dev.so; GIK_PREP_STOP=p leaves every goal after
phase p).  Usage: dev_prep_phases.py [robot] [B]"""
import os, sys, time, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    import numpy as np, torch
    from conftest import make_graph
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    name, B = sys.argv[2], int(sys.argv[3])
    robot, graph = make_graph(name)
    rng = np.random.RandomState(0)
    lb, ub = robot.limits_arrays()
    Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rng.rand(B, robot.n))).cuda()
    prob = BatchProblem(graph, use_limits=True)
    for _ in range(2):
        prob.template.prepare(Tg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        prob.template.prepare(Tg)
    torch.cuda.synchronize()
    print((time.perf_counter() - t0) / 3 * 1e3)
    sys.exit(0)
name = sys.argv[1] if len(sys.argv) > 1 else "ur10_table"
B = sys.argv[2] if len(sys.argv) > 2 else "4096"
phases = ["goal distances, targets", "Floyd-Warshall (upper bounds)", "max-plus passes (lower bounds)", "Gram matrix",
          "Jacobi N x N with eigenvectors", "factor(), MDS matrix", "rank count (Householder + Sturm)",
          "scatter matrix", "Jacobi Kc x Kc", "Y_init (everything)"]
prev = 0.0
for p in list(range(1, 10)) + [0]:
    env = dict(os.environ, GIK_PREP_STOP=str(p))
    out = subprocess.run([sys.executable, __file__, "--one", name, B], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    t = float(out)
    print(f"up to phase {p or 10:2d} ({phases[(p or 10) - 1]:34s}): {t:7.1f} ms  (+{t - prev:6.1f})", flush=True)
    prev = t
