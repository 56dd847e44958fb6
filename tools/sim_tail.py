#!/usr/bin/env python3
"""Makespan of one batch on the wavefront kernel under different scheduling policies (CPU model).

    python tools/sim_tail.py work.npz [--t1 0.405] [--slots 2]

work.npz: per-problem outer iterations `its` and Hessian products `hv` (e.g. from the CPU oracle on
the goals of a bench config).  Model: 1024 SIMDs x 2 wave slots; a lone wave needs t1 us per
Hessian product (972 cycles) plus 4.6 products' worth per outer iteration; two waves on a SIMD run at
(r_eq, r_eq) of that speed when they have the same age priority and (r_hi, r_lo) when they differ
(s_setprio by outer-iteration count: 32 / 128 / 512), calibrated on the measured kernel (bulk rate of
a saturated SIMD, speed of an old wave next to a young one).  Policies:
  fcfs    problems are claimed in index order by whichever wave is free (the kernel as it is)
  spread  + tail spreading: an empty SIMD takes over one of two problems that share a SIMD
  oracle  longest-first order (needs knowledge nobody has): the bound
"""
import argparse
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("work")
ap.add_argument("--t1", type=float, default=0.405)     # us per product, lone wave
ap.add_argument("--simds", type=int, default=1024)
ap.add_argument("--r-eq", type=float, default=0.66)
ap.add_argument("--r-hi", type=float, default=0.93)
ap.add_argument("--r-lo", type=float, default=0.38)
ap.add_argument("--single-below", type=int, default=0,
                help="policy knob: when fewer than this many fresh problems remain, a wave whose SIMD "
                     "partner is busy stops claiming (late problems start alone)")
args = ap.parse_args()
d = np.load(args.work)
its, hv = d["its"].astype(float), d["hv"].astype(float)
work = (hv + 4.6 * its) * args.t1 * 1e-3          # ms alone
per_it = work / its                                 # ms per outer iteration (alone)
B, S = len(work), args.simds


def prio(done_frac, i):
    k = done_frac * its[i]
    return 0 if k < 32 else 1 if k < 128 else 2 if k < 512 else 3


def simulate(order, spread, single_below=0):
    slot = -np.ones((S, 2), dtype=int)              # problem index per (simd, slot)
    rem = work.copy()
    nxt, t, moved = 0, 0.0, 0
    queue = list(order)

    def refill():
        nonlocal nxt
        for s in range(S):
            for w in range(2):
                if slot[s, w] < 0 and nxt < B:
                    if single_below and B - nxt < single_below and slot[s, 1 - w] >= 0:
                        continue
                    slot[s, w] = queue[nxt]
                    nxt += 1

    def rebalance():
        nonlocal moved
        if nxt < B or not spread:
            return
        empty = [s for s in range(S) if slot[s, 0] < 0 and slot[s, 1] < 0]
        double = [s for s in range(S) if slot[s, 0] >= 0 and slot[s, 1] >= 0]
        for e, dsimd in zip(empty, double):
            slot[e, 0] = slot[dsimd, 1]
            slot[dsimd, 1] = -1
            moved += 1

    refill()
    while True:
        rebalance()
        run = slot >= 0
        if not run.any():
            break
        rate = np.zeros((S, 2))
        for s in range(S):
            a, b = slot[s]
            if a >= 0 and b >= 0:
                pa, pb = prio(1 - rem[a] / work[a], a), prio(1 - rem[b] / work[b], b)
                if pa == pb:
                    rate[s] = args.r_eq
                else:
                    rate[s] = (args.r_hi, args.r_lo) if pa > pb else (args.r_lo, args.r_hi)
            elif a >= 0:
                rate[s, 0] = 1.0
            elif b >= 0:
                rate[s, 1] = 1.0
        idx = slot[run]
        dt = min((rem[idx] / rate[run]).min(), 2.0)          # (rates are re-evaluated at least every 2 ms)
        rem[idx] -= rate[run] * dt
        t += dt
        fin = run & (rem[np.where(run, slot, 0)] <= 1e-9)
        slot[fin] = -1
        refill()
    return t, moved


idx = np.arange(B)
print(f"{B} problems, work alone: total {work.sum() / S:.1f} ms per SIMD, longest {work.max():.1f} ms, "
      f"maxiter fraction {(its >= 3000).mean():.3f}")
for name, order, spread, sb in (("fcfs", idx, False, 0), ("spread", idx, True, 0),
                                ("fcfs + late problems alone", idx, False, args.single_below or 1536),
                                ("spread + late alone", idx, True, args.single_below or 1536),
                                ("oracle (longest first)", np.argsort(-work), True, 0)):
    t, moved = simulate(order, spread, sb)
    print(f"{name:32s} makespan {t:7.1f} ms  ({B / t:6.1f} k solves/s)  hand-overs {moved}")
