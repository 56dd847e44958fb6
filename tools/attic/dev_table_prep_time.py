"""dev: host pre-processing time of the UR10 + table scene (N = 116) on the GPU box's host cores."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from graphik_amd.utils.roboturdf import load_ur10
from graphik_amd.utils import table_environment
from graphik_amd.solvers.riemannian_solver import BatchProblem
robot, graph = load_ur10()
for idx, obs in enumerate(table_environment()):
    graph.add_spherical_obstacle(f"o{idx}", obs[0], obs[1])
prob = BatchProblem(graph, use_limits=True)
rng = np.random.RandomState(0)
lb, ub = robot.limits_arrays()
print("host cores", os.cpu_count())
for B, w in ((64, 1), (256, 16), (256, 64), (512, 128)):
    Tg = robot.fk_batch(lb + (ub - lb) * rng.rand(B, robot.n))
    t0 = time.time(); tg, Y0 = prob.prepare(Tg, workers=w); dt = time.time() - t0
    print("B %d workers %d: %.2f s -> %.4f s/goal" % (B, w, dt, dt / B), flush=True)
