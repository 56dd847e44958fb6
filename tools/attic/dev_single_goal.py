"""dev: wall time of the single-goal drop-in call (BASELINE configs[0])."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from graphik_amd.utils.roboturdf import load_schunk_lwa4d
from graphik_amd.solvers.riemannian_solver import solve_with_riemannian
robot, graph = load_schunk_lwa4d()
np.random.seed(0)
ts = []
for i in range(40):
    q = robot.random_configuration()
    T = robot.pose(q, f"p{robot.n}")
    t0 = time.perf_counter(); qs, Y = solve_with_riemannian(graph, T); dt = time.perf_counter() - t0
    if i >= 4: ts.append(dt)
ts = np.array(ts) * 1e3
print("solve_with_riemannian, one goal per call, LWA4D, 36 calls: median %.1f ms, p10 %.1f, p90 %.1f, max %.1f" % (np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), ts.max()))
