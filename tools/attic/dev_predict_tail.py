"""dev: can the long problems of a batch be told apart after a short probe?  Solve every goal with
maxiter = P (probe), then fully; rank by what the probe saw (f, gradnorm, trust radius is not
visible) and report how many of the `maxiter` goals / of the total work the top-S predictions
hold, and simulated makespans: first come first served vs longest-predicted-first after the probe.
Usage: dev_predict_tail.py robot B [probe ...]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import make_graph
from graphik_amd.solvers.riemannian_solver import BatchProblem
from graphik_amd.engine import Template

name, B = sys.argv[1], int(sys.argv[2])
probes = [int(x) for x in sys.argv[3:]] or [40, 80, 160]
robot, graph = make_graph(name)
rng = np.random.RandomState(0)
lb, ub = robot.limits_arrays()
Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rng.rand(B, robot.n))).cuda()
prob = BatchProblem(graph, use_limits=True)
tg, Y0 = prob.template.prepare(Tg)
full = prob.template.solve(Y0, tg)
work = full["inner_executed"].cpu().numpy().astype(float) + 8.0 * full["iterations"].cpu().numpy()   # ~cycles / 972
its = full["iterations"].cpu().numpy()
maxit = its >= 3000
print(f"{name} B={B}: maxiter {maxit.mean():.4f}, total work {work.sum():.3g}, max {work.max():.3g}, mean {work.mean():.3g}")
S = int(os.environ.get("SIM_SERVERS", "1024"))   # SIMDs (wavefront kernel) or CUs (workgroup kernel)
cyc = 972 / 2.4e6   # ms per product-unit

def fast_sim(order, w, slots=2, slow=1.35):
    """List scheduling on S SIMDs with `slots` waves each, problems claimed in `order`; a wave runs at
    1 / slow of its speed while its SIMD has two busy waves.  Event-driven; work in product units."""
    import heapq
    n = len(order); nxt = 0
    # state per SIMD: list of remaining works
    simd = [[] for _ in range(S)]
    for s in range(S):
        for k in range(slots):
            if nxt < n: simd[s].append(w[order[nxt]]); nxt += 1
    tnow = [0.0] * S
    def next_done(s):
        m = len(simd[s])
        if m == 0: return None
        rate = 1.0 if m == 1 else 1.0 / slow
        return tnow[s] + min(simd[s]) / rate
    heap = []
    for s in range(S):
        nd = next_done(s)
        if nd is not None: heapq.heappush(heap, (nd, s))
    tend = 0.0
    while heap:
        t, s = heapq.heappop(heap)
        m = len(simd[s])
        rate = 1.0 if m == 1 else 1.0 / slow
        dt = t - tnow[s]
        simd[s] = [x - dt * rate for x in simd[s]]
        tnow[s] = t
        simd[s] = [x for x in simd[s] if x > 1e-6]
        while len(simd[s]) < slots and nxt < n:
            simd[s].append(w[order[nxt]]); nxt += 1
        tend = max(tend, t)
        nd = next_done(s)
        if nd is not None: heapq.heappush(heap, (nd, s))
    return tend

fcfs = fast_sim(np.arange(B), work)
lpt = fast_sim(np.argsort(-work), work)
# ideal processor sharing (every unfinished problem advances at the same rate, at most one SIMD
# each): the service level s grows at min(1, S / n(s)), n(s) = number of problems longer than s
ws = np.sort(work)
edges = np.concatenate(([0.0], ws))
n_longer = len(ws) - np.arange(len(ws))          # problems longer than edges[k], k = 0..n-1
t_ps = float(np.sum(np.diff(edges) * np.maximum(n_longer / S, 1.0)))
print(f"ideal processor sharing (time slicing with migration, no overhead): {t_ps * cyc:.1f} ms")
print(f"simulated makespan: FCFS {fcfs * cyc:.1f} ms, clairvoyant longest-first {lpt * cyc:.1f} ms, lower bound max(job, total/S) {max(work.max(), work.sum() / S) * cyc:.1f} ms")
for P in probes:
    T = Template.from_matrices(prob.omega, prob.psi_L, prob.psi_U, k=3, use_limits=True, params={"maxiter": P})
    pr = T.solve(Y0, tg)
    f = pr["f"].cpu().numpy(); gn = pr["gradnorm"].cpu().numpy(); done = pr["iterations"].cpu().numpy() < P
    spent = pr["inner_executed"].cpu().numpy().astype(float) + 8.0 * pr["iterations"].cpu().numpy()
    rest = np.maximum(work - spent, 0.0)
    for label, key in (("f", f), ("gradnorm", gn), ("f*gn", f * gn)):
        key = np.where(done, -1.0, key)
        order = np.argsort(-key)
        top = order[:S]
        cap_max = maxit[top].sum() / max(1, maxit.sum())
        cap_work = rest[top].sum() / rest.sum()
        from scipy.stats import spearmanr
        rho = spearmanr(key[~done], rest[~done]).correlation
        # two-phase: probe phase (all problems, P iterations, FCFS) then predicted-longest-first on the rest
        t1 = fast_sim(np.arange(B), spent)
        t2 = fast_sim(order[: int((~done).sum())], rest)
        print(f"probe {P:4d} key {label:9s}: unfinished {1 - done.mean():.3f}, probe work {spent.sum() / work.sum():.3f} of total; Spearman(key, rest) {rho:.2f}; "
              f"top-{S} holds {cap_max:.3f} of the maxiter goals, {cap_work:.3f} of the remaining work; two-phase makespan {(t1 + t2) * cyc:.1f} ms")
