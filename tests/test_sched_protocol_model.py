"""Exhaustive interleaving check of the wavefront kernel's yield-queue protocol (round-robin time
slicing; graphik_amd/csrc/gik_kernels.hip.h: rtr_wave_kernel<.., MIG> and mig_wait, SolveArgs::y_*).

The device code takes yield-queue entries by fetch-add tickets (no compare-and-swap), which is only
safe if a ticket's entry is guaranteed to exist.  This file restates the protocol as a state machine
-- one atomic memory operation per step, the operations of the kernel in the kernel's order -- and
explores EVERY interleaving of a few waves over a few problems:

  * claims are conserved: y_avail + (fetch-adds about to be undone) + (waves that hold a claim)
    == entries pushed - tickets taken, at every step; so a ticket is always below the pushes,
  * no problem is ever held by two waves, none is lost, every one finishes,
  * no reachable state is stuck (problems unfinished and no wave able to take a step).

Tail spreading's donor / helper hand-over is represented by its effect on this protocol only: a wave
that finds nothing to claim may, once unfinished <= waves, COMMIT to a hand-over ticket, after which
it no longer looks at the yield queue (it waits for a donor or for the end of the batch; a donation
moves a running problem between waves and does not touch the queue).
No GPU, no library: pure Python."""
import pytest

# program counters of a wave
(CLAIM, SEM_LOAD, SEM_FAA, SEM_UNDO, POP, POP_WAIT, RUN, YIELD, PUBLISH, WAIT, COMMITTED, EXIT) = range(12)


def explore(n_waves, slices, cap=10 ** 9):
    """slices[i] = number of time slices problem i needs.  Returns the number of states visited."""
    B = len(slices)
    # shared: fresh, y_head, y_tail, published bitmask, ids, y_avail, done, remaining slices
    # wave: (pc, problem, tail, owns_entry, ticket)
    idle = (CLAIM, -1, False, False, -1)
    init = (0, 0, 0, 0, (), 0, 0, tuple(slices), tuple(idle for _ in range(n_waves)))
    seen, stack = {init}, [init]
    while stack:
        st = stack.pop()
        fresh, yh, yt, pub, ids, avail, done, rem, waves = st
        running = [w[1] for w in waves if w[0] in (RUN, YIELD)]      # (PUBLISH: the entry is already in the queue)
        assert len(running) == len(set(running)), ("a problem is held by two waves", st)
        holders = sum(1 for w in waves if w[0] in (POP, PUBLISH) or (w[0] == CLAIM and w[3]))
        undo = sum(1 for w in waves if w[0] == SEM_UNDO)
        assert avail + undo + holders == yt - yh, ("claims are not conserved", st)
        queued = [ids[k] for k in range(yh, yt)] + [ids[w[4]] for w in waves if w[0] == POP_WAIT]
        alive = sorted(running + queued + list(range(min(fresh, B), B)))
        assert alive == [i for i in range(B) if rem[i] > 0], ("a problem was lost or duplicated", st)
        successors = []
        for i, (pc, b, tail, owns, tk) in enumerate(waves):
            def put(nw, **kw):
                s = dict(fresh=fresh, yh=yh, yt=yt, pub=pub, ids=ids, avail=avail, done=done, rem=rem)
                s.update(kw)
                successors.append((s["fresh"], s["yh"], s["yt"], s["pub"], s["ids"], s["avail"], s["done"],
                                   s["rem"], waves[:i] + (nw,) + waves[i + 1:]))

            if pc == CLAIM:
                if not tail:                       # atomicAdd(work_counter)
                    if fresh < B:                  # a fresh problem; a wave that has just pushed passes its claim on
                        put((RUN, fresh, False, False, -1), fresh=fresh + 1, avail=avail + (1 if owns else 0))
                    else:
                        put((POP if owns else SEM_LOAD, -1, True, False, -1), fresh=fresh + 1)
                else:
                    put((POP if owns else SEM_LOAD, -1, True, False, -1))
            elif pc == SEM_LOAD:                   # if (load(y_avail) > 0) ... else mig_wait()
                put((SEM_FAA if avail > 0 else WAIT, -1, True, False, -1))
            elif pc == SEM_FAA:                    # fetch_add(y_avail, -1) > 0 ?
                put((POP if avail > 0 else SEM_UNDO, -1, True, False, -1), avail=avail - 1)
            elif pc == SEM_UNDO:                   # fetch_add(y_avail, +1), then mig_wait()
                put((WAIT, -1, True, False, -1), avail=avail + 1)
            elif pc == POP:                        # h = fetch_add(y_head, 1)
                assert yh < yt, ("ticket without an entry", st)
                put((POP_WAIT, -1, True, False, yh), yh=yh + 1)
            elif pc == POP_WAIT:                   # while (y_seq[h] != h + 1) sleep
                if (pub >> tk) & 1:
                    put((RUN, ids[tk], True, False, -1))
            elif pc == RUN:                        # one time slice of rtr_solve_one
                r = rem[b] - 1
                nrem = rem[:b] + (r,) + rem[b + 1:]
                if r == 0:
                    put((CLAIM, -1, tail, False, -1), rem=nrem, done=done + 1)
                elif B > done + n_waves and yt < cap:      # mig_anyone_waiting(): loads of y_tail, q_done
                    put((YIELD, b, tail, False, -1), rem=nrem)
                else:
                    put((RUN, b, tail, False, -1), rem=nrem)
            elif pc == YIELD:                      # (state saved) k = atomicAdd(y_tail, 1)
                put((PUBLISH, b, tail, False, yt), yt=yt + 1, ids=ids + (b,))
            elif pc == PUBLISH:                    # y_ids[k] = b; y_seq[k] = k + 1 (release); owns_entry = true
                put((CLAIM, -1, tail, True, -1), pub=pub | (1 << tk))
            elif pc == WAIT:                       # mig_wait(): one pass of its loop
                if done >= B:
                    put((EXIT, -1, True, False, -1))
                elif avail > 0:
                    put((SEM_LOAD, -1, True, False, -1))       # returns -2: back to the claim
                elif B <= done + n_waves:
                    put((COMMITTED, -1, True, False, -1))      # (if its SIMD is empty: may or may not happen)
            elif pc == COMMITTED:                  # waits for a donor's problem or the end of the batch
                if done >= B:
                    put((EXIT, -1, True, False, -1))
        if not successors:
            assert done == B and all(w[0] == EXIT for w in waves), ("stuck", st)
        for s in successors:
            if s not in seen:
                seen.add(s)
                stack.append(s)
    return len(seen)


@pytest.mark.parametrize("n_waves,slices", [
    (2, (1, 1, 1)), (2, (3, 1, 1)), (2, (2, 2, 2)), (2, (1, 3, 1, 2)), (2, (4, 1, 1, 1)), (2, (2, 1, 2, 1, 1)),
    (3, (2, 1, 3, 1)), (3, (3, 3, 1, 1)), (3, (2, 2, 2, 2)), (3, (3, 1, 2, 1, 2)), (3, (4, 4, 1, 1, 1)),
])
def test_every_interleaving_finishes(n_waves, slices):
    """(3 waves x (3, 3, 2, 2, 1): 1.5 M states, 22 s -- passes too, left out of the suite.)"""
    assert explore(n_waves, slices) > 10


def test_full_queue_stops_yielding():
    """With room for two entries only, the waves stop yielding and run their problems to the end."""
    explore(2, (3, 3, 3), cap=2)
    explore(3, (3, 2, 3, 2), cap=1)


def test_the_model_catches_a_broken_protocol():
    """Sanity of the checker: a pop without a claim (the compare-free scheme WITHOUT the semaphore)
    must trip the `ticket without an entry` / conservation assertions."""
    import types
    src = open(__file__).read().replace("put((POP if owns else SEM_LOAD, -1, True, False, -1))",
                                        "put((POP, -1, True, False, -1))")
    mod = types.ModuleType("broken")
    mod.__dict__["__file__"] = __file__
    exec(compile(src.split("@pytest.mark.parametrize")[0], "broken", "exec"), mod.__dict__)
    with pytest.raises(AssertionError):
        mod.explore(2, (2, 2, 2))
