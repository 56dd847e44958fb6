"""dev: time-slicing policies of the workgroup-per-problem kernel, simulated on the per-problem work
of a real batch.  `dump` (on the GPU box): solve 4096 table-scene goals, save outer iterations and
executed products per problem.  `sim` (anywhere): S workgroups, slice length L outer iterations,
requeue policy fifo (behind everything that waits), oldest (requeued problems by iterations so far,
after the fresh ones) or aged (problems past an age threshold before everything else, oldest first),
makespan in units of products.
Usage: dev_slice_sim.py dump out.npz | dev_slice_sim.py sim in.npz"""
import os, sys, heapq
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))

def dump(path, B=4096):
    import torch
    from conftest import make_graph
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph("ur10_table")
    rng = np.random.RandomState(0)
    lb, ub = robot.limits_arrays()
    Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rng.rand(B, robot.n))).cuda()
    prob = BatchProblem(graph, use_limits=True)
    tg, Y0 = prob.template.prepare(Tg)
    r = prob.template.solve(Y0, tg)
    np.savez(path, its=r["iterations"].cpu().numpy(), work=r["inner_executed"].cpu().numpy())
    print("saved", path)

def simulate(its, work, S, L, policy, age=0):
    """Each problem: `its` outer iterations of work/its products each.  A workgroup runs a problem for
    L iterations (or to its end), then requeues it.  Fresh problems first, in order."""
    n = len(its)
    per_it = work / np.maximum(its, 1)
    done_its = np.zeros(n, int)
    fresh = list(range(n))[::-1]                 # pop() gives 0, 1, 2, ...
    fifo = []                                    # (seq, id) heap for requeued, fifo order
    old = []                                     # (-its_done, seq, id)
    seq = 0
    free = [(0.0, s) for s in range(S)]          # (time free, server)
    heapq.heapify(free)
    t_end = 0.0
    pending = []                                 # (finish time, id) of running slices -> requeue at that time
    # event-driven: process servers in order of free time; requeued problems become available at their finish time
    avail = []                                   # heap of (time available, key..., id)
    while True:
        t, s = heapq.heappop(free)
        # move problems whose slice finished by time t into the queues
        while pending and pending[0][0] <= t:
            tf, i = heapq.heappop(pending)
            if policy == "fifo" or (policy == "aged" and done_its[i] < age):
                heapq.heappush(fifo, (seq, i))
            else:
                heapq.heappush(old, (-done_its[i], seq, i))
            seq += 1
        if policy == "aged" and old:               # problems past the age threshold: before everything
            _, _, i = heapq.heappop(old)
        elif fresh:
            i = fresh.pop()
        elif fifo:
            _, i = heapq.heappop(fifo)
        elif old:
            _, _, i = heapq.heappop(old)
        elif pending:
            # nothing available now: wait for the next slice to finish
            heapq.heappush(free, (pending[0][0], s))
            continue
        else:
            break
        k = min(L if L > 0 else 10 ** 9, its[i] - done_its[i])
        tf = t + k * per_it[i]
        done_its[i] += k
        if done_its[i] < its[i]:
            heapq.heappush(pending, (tf, i))
        t_end = max(t_end, tf)
        heapq.heappush(free, (tf, s))
    return t_end

if sys.argv[1] == "dump":
    dump(sys.argv[2])
else:
    d = np.load(sys.argv[2])
    its, work = d["its"].astype(int), d["work"].astype(float)
    S = 256
    print(f"{len(its)} problems, total work {work.sum():.4g}, max {work.max():.4g}; bound max(job, total / S) = {max(work.max(), work.sum() / S):.4g}")
    for L in (0, 256, 96, 32):
        for pol in ("fifo", "oldest"):
            if L == 0 and pol == "oldest":
                continue
            print(f"slice {L:4d} {pol:7s}: makespan {simulate(its, work, S, L, pol):.4g}")
        if L:
            for age in (L, 2 * L, 4 * L, 8 * L):
                print(f"slice {L:4d} aged >= {age:4d} first: makespan {simulate(its, work, S, L, 'aged', age):.4g}")
