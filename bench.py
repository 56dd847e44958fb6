#!/usr/bin/env python3
"""bench.py -- IK solves/s of the MI355X-native GraphIK RiemannianSolver path.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5]

`--gpus N` with N > 1 brings up N ranks itself (it re-executes this file under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, one
process per GPU, RCCL); started under an external torch.distributed.run it uses the environment
that launcher provides.

Workloads (BASELINE.json `configs`; goals = FK of uniform random configurations within the joint
limits, drawn exactly as the reference's examples draw them -- robot.random_configuration()):
  c2 (default, the bench line)  Schunk LWA4D, 4096 goals per GPU (weak scaling)
  c3  UR10 + table_environment(), 4096 goals per GPU (weak scaling)
  c4  KUKA iiwa, 65536 goals sharded contiguously over the GPUs (strong scaling)
  c5  10-link planar chain, 65536 goals sharded contiguously over the GPUs (strong scaling)
One "step" = one pass of the hot path over the rank's shard with its inputs resident in HBM.
Goals are independent: no data-path collective, ONE gather of the per-problem results at the end.
Rank 0 prints ONE JSON line.

`--dry-solve` (CPU, `--backend gloo`) replaces the device solve by a deterministic per-goal stand-in
so that the rank / shard / gather logic can be tested without a GPU; its line carries
"dry_solve": true and no throughput.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# HIP maps streams onto 4 hardware queues by default, so of more than 3 side streams some share a
# queue and serialise; the serving measurement keeps 16 batches in flight (read at HIP start-up;
# measured at 4096 LWA4D goals per batch: 4 queues 65 k, 8 queues / 8 streams 110 k, 16 / 12 190 k,
# 24 / 16 203 k solves/s)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

FP64_PEAK_TFLOPS = 78.6   # MI355X vector = matrix fp64 peak (spec; fp32 vector 157.3 / 2)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec

CONFIGS = {   # name -> (robot, goals, "per_gpu" | "total", BASELINE.json configs index)
    "c2": ("lwa4d", 4096, "per_gpu", 1),
    "c3": ("ur10_table", 4096, "per_gpu", 2),
    "c4": ("kuka", 65536, "total", 3),
    "c5": ("planar10", 65536, "total", 4),
}


def algorithmic_flops(N, k, T, n_inner, n_outer, n_accept):
    """F_solve = n_inner*F_hv + n_outer*F_out (+ F_acc per accepted step)."""
    F_hv = 12 * k * T + (4 * N * k * k + k * k + 2 * k ** 4) + 16 * N * k
    F_cost = (3 * k + 6) * T + 5 * N * k
    F_acc = 9 * k * T + N * k * (k + 1) + (2.0 / 3.0) * k ** 6
    return n_inner * F_hv + n_outer * F_cost + n_accept * F_acc


def library_digest():
    """Digest of the sources the loaded libgraphik_amd.so was built from (graphik_amd/build.py writes it next to the
    library): the key that ties a profile under profiles/ to a binary."""
    try:
        from graphik_amd import _ffi
        return open(_ffi.LIB_PATH + ".digest").read().strip()
    except Exception:
        return None


def prepare_flops(N, K):
    """F_init ~ 2 * 9 N^3 + 9 K^3 (two N x N and one K x K Jacobi-class eigendecomposition) per
    goal, plus bound smoothing as the kernels run it: Floyd-Warshall on UPPER (N^3 add + min = 2 N^3) and
    the two-stage max-plus pass for LOWER (2 N^3).  K: the MDS column count the kernel reports per goal."""
    return 2 * 9 * N ** 3 + 9 * np.asarray(K, dtype=float) ** 3 + 4 * N ** 3


def algorithmic_bytes(N, k, T):
    """HBM bytes the solve kernel must move per IK problem: T targets + Y_init in, Y_sol + stats
    out (SURVEY 8(d) counts the pipeline-level 616 B/solve; the dominant kernel alone sees this)."""
    return 8 * (T + 2 * N * k) + 48          # (+ the 48-byte gik_stats record)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="BASELINE.json workload (default c2 = configs[1], the bench line)")
    ap.add_argument("--robot", default=None,
                    choices=["lwa4d", "ur10", "kuka", "planar10", "planar10_halfpi", "ur10_table"],
                    help="parity configs outside BASELINE's list (overrides --config's robot)")
    ap.add_argument("--batch", type=int, default=0, help="goals per GPU (overrides the config's size)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="problems for the CPU baseline (0=auto)")
    ap.add_argument("--streams", type=int, default=1,
                    help="batches in flight in the TIMED region (default 1: steps back to back, the "
                         "headline configuration)")
    ap.add_argument("--serving-streams", type=int, default=16,
                    help="extra, separately labelled measurement after the timed region: the same "
                         "batches issued round-robin on this many HIP streams (0 = skip)")
    ap.add_argument("--hessian-form", choices=["auto", "column", "per_edge"], default="auto",
                    help="3-D arms on the wavefront kernel: how lhess is rendered (gik_template_desc.hessian_form); "
                         "auto (the library's default, the BASELINE line) = per_edge: s = y.w once per edge as "
                         "costs.py:186-203 writes it; column = the cached-rows form that was the default until round 5 "
                         "(+5-8 % Hessian products against the reference's arithmetic)")
    ap.add_argument("--intended", action="store_true",
                    help="ur10_table only: the opt-in fixed-anchor formulation with the robot<->obstacle "
                         "hinges the reference means to create (SURVEY 8(f)3); NOT the reference's observable "
                         "semantics and not the BASELINE line")
    ap.add_argument("--headline-only", action="store_true",
                    help="default invocation: skip the c3 / c4 / c5 measurements that follow the headline")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"])
    ap.add_argument("--dry-solve", action="store_true",
                    help="no GPU: deterministic stand-in for the solve (tests of the N>1 logic)")
    return ap.parse_args(argv)


def self_launch(args):
    """--gpus N > 1 outside a launcher: one rank per GPU under torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def build_graph(robot_name):
    from graphik_amd.utils.roboturdf import load_schunk_lwa4d, load_ur10, load_kuka
    if robot_name.startswith("planar10"):
        from graphik_amd.robots import RobotPlanar
        from graphik_amd.graphs import ProblemGraphPlanar
        from graphik_amd.utils import list_to_variable_dict
        nl = 10
        lim = np.array(9 * [np.pi / 2] + [np.pi]) if robot_name.endswith("halfpi") else np.pi * np.ones(nl)
        robot = RobotPlanar({"link_lengths": list_to_variable_dict(np.ones(nl)),
                             "theta": list_to_variable_dict(np.zeros(nl)),
                             "joint_limits_upper": list_to_variable_dict(lim),
                             "joint_limits_lower": list_to_variable_dict(-lim), "num_joints": nl})
        return robot, ProblemGraphPlanar(robot)
    if robot_name == "ur10_table":
        # BASELINE configs[2]: UR10 + table_environment() (N = 116, 5612 terms): the
        # workgroup-per-goal prepare kernel and the workgroup-per-problem solve kernel
        from graphik_amd.utils import table_environment
        robot, graph = load_ur10()
        for idx, obs in enumerate(table_environment()):
            graph.add_spherical_obstacle(f"o{idx}", obs[0], obs[1])
        return robot, graph
    return {"lwa4d": load_schunk_lwa4d, "ur10": load_ur10, "kuka": load_kuka}[robot_name]()


def workload(args, world):
    """(config name, robot name, goals of the whole job, scaling)"""
    cfg = args.config or ("c2" if args.robot is None else None)
    if cfg is not None:
        robot_name, goals, mode, _ = CONFIGS[cfg]
    else:
        robot_name, goals, mode = args.robot, 4096, "per_gpu"
    if args.robot is not None:
        robot_name = args.robot
    if args.batch:
        goals, mode = args.batch, "per_gpu"
    total = goals * world if mode == "per_gpu" else goals
    return cfg, robot_name, total, ("weak" if mode == "per_gpu" else "strong")


def dry_solve(T_goal, n):
    """Deterministic per-goal stand-in for the device solve (--dry-solve): q and every statistic of
    graphik_amd.distributed.RESULT_STATS as a function of the goal pose alone, so the table gathered
    from N ranks must equal the single-process one row for row."""
    flat = T_goal.reshape(len(T_goal), -1)
    key = np.abs(flat).sum(axis=1)
    its = 1.0 + np.floor(997.0 * (key - np.floor(key)))
    q = np.sin(key[:, None] * (1.0 + np.arange(n)[None, :]))
    return {"q": q, "pos_err": 1e-4 * key, "rot_err": 1e-4 * (key + 1.0), "f": 1e-12 * key, "gradnorm": 1e-10 * key,
            "iterations": its, "inner_total": 40.0 * its, "n_accept": np.floor(0.8 * its),
            "stop": np.zeros_like(key), "inner_executed": 38.0 * its}


def cpu_baseline(prob, T_goal, Y0_h, B, args):
    """The oracle (C restatement, -O3 AVX2+FMA, OpenMP over problems) timed on this host: one
    thread on a small sample and every usable core on 64 goals per thread (capped at the batch).
    Hessian products per second per thread are tail-free and are what to compare machines by."""
    from oracle import c_oracle as co
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:   # a cgroup CPU quota can be far below the affinity mask
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    cores = max(1, min(usable, int(quota)) if quota else usable)

    def run(ns, nthreads, D=None):
        if D is None:
            D, _, _ = prob.assemble(T_goal[:ns])
        t0 = time.perf_counter()
        o = co.rtr_solve_batch(Y0_h[:ns], D, prob.omega, prob.psi_L, prob.psi_U, prob.psi_L is not None,
                               nthreads=nthreads, fast=True)
        dt = time.perf_counter() - t0
        return dt, int(o["inner_total"].sum())

    big = prob.N * prob.dim > 64            # table scene: ~1 s per solve per thread
    planar = prob.dim == 2                  # ~50 products per solve
    n1 = min(B, 2 if big else (256 if planar else 24))
    t1, hv1 = run(n1, 1)
    # a bounded sample (>= 256 problems per config, the table scene >= 16): ~5-20 s of CPU work
    ns = args.cpu_sample or min(B, max(16, 2 * cores) if big else (65536 if planar else max(256, 64 * cores)))
    Ds, _, _ = prob.assemble(T_goal[:ns])
    tc, hvc = run(ns, cores, Ds)
    reps = 1
    if tc < 0.3:      # planar goals: ~2 us each -- repeat the sample until the clock has something to measure
        reps = int(np.ceil(0.3 / max(tc, 1e-4))) + 1
        tc = sum(run(ns, cores, Ds)[0] for _ in range(reps)) / reps
    return {
        "value": ns / tc, "unit": "solves/s", "cores": cores, "kind": "port",
        "sample": f"first {ns} goals of rank 0's batch ({ns / cores:.1f} per thread), "
                  f"oracle/gik_oracle.c (-O3 AVX2+FMA), OpenMP dynamic schedule over problems, "
                  f"{tc * reps:.2f} s wall" + (f" ({reps} repeats of the sample)" if reps > 1 else ""),
        "hv_total": hvc, "hv_per_s_per_thread": hvc / tc / cores,
        "single_thread": {"value": n1 / t1, "unit": "solves/s", "sample": f"first {n1} goals, {t1:.1f} s",
                          "hv_per_s": hv1 / t1},
        "effective_parallelism": (hvc / tc) / (hv1 / t1),
        "host": {"os_cpu_count": os.cpu_count(), "sched_affinity": usable, "cgroup_cpu_quota": quota},
    }


def executed_flops(N, k, T, info, lowrank_frac, n_inner, n_outer, n_accept):
    """Flops the solve kernel EXECUTES (NOTEBOOK 4.1).  Wavefront kernel: the algorithmic count
    (every term is evaluated once per product).  Workgroup kernel with a rigid clique of n nodes
    in closed form: per Hessian product the T_rest terms left in the per-term loops (12 k each),
    the moments (18, or 27 with Euclidean targets: one multiply + one add per clique node each),
    the closed form itself (58 flops per clique unknown, 49 without the low-rank D w part) and,
    without Euclidean targets, the dense D w product (2 n flops per clique unknown); cost and
    gradient walk every term as the algorithmic count says."""
    n = int(info.get("n_clique", 0)) if info else 0
    if n == 0:
        return algorithmic_flops(N, k, T, n_inner, n_outer, n_accept)
    T_rest = int(info["n_slot_terms"])
    if info.get("node_per_lane"):
        # node-per-lane kernel (gik_npt.hip.h): every slot term once (12 k, as the reference's edge loop),
        # 24 moments (15 without Euclidean targets), ~110 (83) flops of closed form per clique NODE
        hv_low = 12 * k * T_rest + 24 * 2 * n + 110 * n
        hv_dense = 12 * k * T_rest + 15 * 2 * n + 83 * n + 2 * n * 3 * n
    else:
        hv_low = 12 * k * T_rest + 27 * 2 * n + 58 * 3 * n
        hv_dense = 12 * k * T_rest + 18 * 2 * n + 49 * 3 * n + 2 * n * 3 * n
    hv = lowrank_frac * hv_low + (1.0 - lowrank_frac) * hv_dense
    F_hv = hv + (4 * N * k * k + k * k + 2 * k ** 4) + 16 * N * k
    F_cost = (3 * k + 6) * T + 5 * N * k
    F_acc = 9 * k * T + N * k * (k + 1) + (2.0 / 3.0) * k ** 6
    return n_inner * F_hv + n_outer * F_cost + n_accept * F_acc


class Bench:
    """One process = one rank = one GPU.  `measure()` times one workload (a BASELINE config) on this
    rank's shard and returns the report dict on rank 0 (None elsewhere)."""

    def __init__(self, args):
        import torch
        from graphik_amd import distributed as gd
        self.args, self.torch, self.gd = args, torch, gd
        backend = args.backend or ("gloo" if args.dry_solve else None)
        self.rank, self.local_rank, self.world = gd.init_process_group(backend=backend)
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {self.world} rank(s); "
                             f"run `python bench.py --gpus {args.gpus}` (self-launching) or "
                             f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py "
                             f"--gpus {args.gpus}`")
        self.dry = args.dry_solve
        if not self.dry:
            assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device("cuda", self.local_rank)
        else:
            self.dev = torch.device("cpu")

    # -- goals: rank r draws rows shard_range(total, r, world) of ONE global random stream --------
    def goals(self, robot, total, seed):
        lo, hi = self.gd.shard_range(total, self.rank, self.world)
        rs = np.random.RandomState(seed)
        U = rs.rand(total, robot.n)[lo:hi]
        lb, ub = robot.limits_arrays()
        return robot.fk_batch(lb + (ub - lb) * U), hi - lo

    def measure_dry(self, cfg, robot_name, total, scaling, steps, warmup):
        gd, torch = self.gd, self.torch
        robot, graph = build_graph(robot_name)
        T_goal, B = self.goals(robot, total, self.args.seed)
        for _ in range(warmup):
            st = dry_solve(T_goal, robot.n)
        gd.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            st = dry_solve(T_goal, robot.n)
        gd.barrier()
        dt_local = time.perf_counter() - t0
        dt = gd.max_over_ranks(dt_local, self.dev)
        per_rank = gd.gather_vectors([dt_local / steps * 1e3, float(st["iterations"].max()) if B else 0.0,
                                      float(np.mean(st["stop"] == 1)) if B else 0.0, float(B)], self.dev)
        # the library's result table (q + statistics per problem) and its ONE gather
        allstats = gd.gather_rows(gd.pack_results(st), total, dst=0)
        if self.rank != 0:
            return None
        a = allstats.numpy()
        return {
            "metric": "IK solves/sec (batched random goals)", "value": None, "unit": "solves/s",
            "dry_solve": True, "n_gpus": self.world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "scaling": scaling,
            "config": {"workload": f"{robot_name}, {total} goals over {self.world} rank(s)",
                       "config": cfg, "robot": robot_name, "goals_total": total},
            "per_rank": [dict(zip(("ms", "max_outer", "frac_maxiter", "goals"), v)) for v in per_rank],
            "rows": int(a.shape[0]), "row_bytes": int(a.shape[1]) * 8, "checksum": float(a.sum()),
            "rows_sha": __import__("hashlib").sha256(np.ascontiguousarray(a).tobytes()).hexdigest(),
        }

    def measure(self, cfg, robot_name, total, scaling, steps, warmup, serving_streams=0,
                cpu=False, n_streams=1, intended=False, seed=None, use_limits=True, hessian_form=None):
        if self.dry:
            return self.measure_dry(cfg, robot_name, total, scaling, steps, warmup)
        args, torch, gd, dev, rank, world = self.args, self.torch, self.gd, self.dev, self.rank, self.world
        seed = args.seed if seed is None else seed
        robot, graph = build_graph(robot_name)
        T_goal, B = self.goals(robot, total, seed)
        from graphik_amd.solvers.riemannian_solver import AnchoredProblem, BatchProblem
        anch = None
        if intended:
            if robot_name != "ur10_table":
                raise SystemExit("--intended applies to --config c3 / --robot ur10_table")
            anch = AnchoredProblem(graph, device=dev)
            prob = anch.base
            N, k = len(anch.free), 3
            T = anch.template.T + len(anch.pin)              # terms the Hessian product sees
        else:
            form = hessian_form or self.args.hessian_form
            prob = BatchProblem(graph, use_limits=use_limits, device=dev,
                                params=(None if form == "auto" else {"hessian_form": form}))
            N, k, T = graph.number_of_nodes(), graph.dim, prob.template.T
        tpl = prob.template
        on_device = prob.device_pipeline
        Tg_dev = torch.from_numpy(T_goal).to(dev)        # inputs resident in HBM
        if not on_device:
            tg_h, Y0_h0 = prob.prepare(T_goal)
            tg_dev, Y0_dev = torch.from_numpy(tg_h).to(dev), torch.from_numpy(Y0_h0).to(dev)
        torch.cuda.synchronize(dev)
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        evp = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]     # before the prepare kernel
        evr = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]     # after the recover kernel

        anch_ms = []

        def step(i=None):
            """goal poses -> joint angles + pose errors, entirely on the device: prepare
            (from_pose, bound smoothing, MDS init) -> RTR solve -> recover (joint_variables, FK)."""
            if anch is not None:        # fixed-anchor pipeline: one C call (prepare, fit, solve, gather, recover)
                res = anch.template.anchored_ik(tpl, Tg_dev)
                if i is not None:
                    anch_ms.append(anch.template)
                res.update(Y0=res["x"])
                return res
            if i is not None:
                evp[i].record()      # all kernels are launched on torch's current stream
            Kc = None
            if on_device:
                targets, Y0, Kc = tpl.prepare(Tg_dev, return_K=True)
            else:
                targets, Y0 = tg_dev, Y0_dev
            if i is not None:
                ev0[i].record()
            res = tpl.solve(Y0, targets)
            if i is not None:
                ev1[i].record()
            if on_device:
                q, pe, re = tpl.recover(res["x"], Tg_dev)
                res.update(q=q, pos_err=pe, rot_err=re)
            if i is not None:
                evr[i].record()
            res.update(Y0=Y0, K=Kc)
            return res

        for _ in range(warmup):
            res = step()
        torch.cuda.synchronize(dev)
        gd.barrier()
        torch.cuda.synchronize(dev)
        streams = [torch.cuda.Stream(dev) for _ in range(n_streams)] if n_streams > 1 else None
        t0 = time.perf_counter()
        for i in range(steps):
            if streams is None:
                res = step(i)
            else:                      # every step is the same full batch; S of them are in flight
                with torch.cuda.stream(streams[i % n_streams]):
                    res = step(i)
        torch.cuda.synchronize(dev)
        gd.barrier()
        dt_local = time.perf_counter() - t0
        dt = gd.max_over_ranks(dt_local, dev)
        # what every rank saw: a shard is as long as ITS longest problem, so the first hardware scaling run
        # can be read against DESIGN 5's straggler model rank by rank (a few doubles per rank, outside the timed region)
        its_l = res["iterations"].double()
        per_rank = gd.gather_vectors([dt_local / steps * 1e3, float(its_l.max()) if B else 0.0,
                                      float((res["stop"] == 1).double().mean()) if B else 0.0, float(B)], dev)
        if anch is not None:
            # The events around the anchored solve kernel live inside the C call and reading them waits
            # for it, so the per-step kernel time is taken in a pass of its own AFTER the timed region
            # (same inputs, `steps` launches, mean).
            ks = []
            for _ in range(steps):
                step()
                ks.append(float(anch.template.lib.gik_anchored_last_solve_ms(anch.template._h)))
            kernel_ms = float(np.mean(ks))
        else:
            ks = [float(a.elapsed_time(b)) for a, b in zip(ev0, ev1)]
            kernel_ms = float(np.mean(ks))
        prep_ms = recover_ms = None
        if anch is None and on_device:
            prep_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(evp, ev0)]))
            recover_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev1, evr)]))

        # Serving configuration, reported separately (NOT `value`): the same batches, S in flight on
        # separate HIP streams, so the straggler tail of one batch overlaps with the bulk of the next.
        serving = None
        S = serving_streams
        if S > 1 and n_streams == 1 and dt / steps < 0.5:   # (skipped for multi-second batches)
            nb = 4 * S
            sv_streams = [torch.cuda.Stream(dev) for _ in range(S)]
            for s_ in sv_streams:                        # warm the per-stream allocations
                with torch.cuda.stream(s_):
                    step()
            torch.cuda.synchronize(dev)
            gd.barrier()
            ts = time.perf_counter()
            for i in range(nb):
                with torch.cuda.stream(sv_streams[i % S]):
                    step()
            torch.cuda.synchronize(dev)
            gd.barrier()
            dts = gd.max_over_ranks(time.perf_counter() - ts, dev)
            serving = {"value": total * nb / dts, "unit": "solves/s", "batches_in_flight": S, "seconds": dts,
                       "batches": nb, "ms_per_batch": dts / nb * 1e3,
                       "note": "same workload and kernels as `value`; S batches in flight on separate HIP "
                               "streams (a serving configuration: independent requests overlap, the "
                               "straggler tail of one batch runs beside the bulk of the next)"}

        if not on_device:   # host post-processing, after the timed region
            qh = prob.joint_variables(res["x"].cpu().numpy(), T_goal)
            pe, re = prob.pose_errors(qh, T_goal)
            res.update(pos_err=torch.from_numpy(pe).to(dev), rot_err=torch.from_numpy(re).to(dev),
                       q=torch.from_numpy(np.asarray(qh, dtype=float)).to(dev))

        # single gather of the per-problem results at the end (RCCL over xGMI when N > 1): the table of
        # graphik_amd.distributed.solve_batch_sharded -- joint angles + statistics per problem
        allstats = gd.gather_rows(gd.pack_results(res, device=dev), total, dst=0)
        Y0_h = res["Y0"].cpu().numpy()

        flags_local = res["flags"] if "flags" in res else None
        lowrank_local = float((flags_local & 1).double().mean()) if flags_local is not None else 0.0
        if rank != 0:
            return None
        st = allstats.cpu().numpy()
        import hashlib
        import torch.distributed as dist
        gather = {"rows": int(st.shape[0]), "bytes_per_problem": int(st.shape[1]) * 8,
                  "bytes": int(st.shape[0]) * int(st.shape[1]) * 8,
                  "backend": dist.get_backend() if dist.is_initialized() else None,
                  "sha256": hashlib.sha256(np.ascontiguousarray(st).tobytes()).hexdigest(),
                  "note": "per-problem results of all ranks on rank 0, the ONE collective of the job (gather to rank 0 over "
                          "RCCL): q [n] + " + ", ".join(gd.RESULT_STATS) + " -- the table of "
                          "graphik_amd.distributed.solve_batch_sharded (the points Y travel on request: with_Y)"}
        _, _, ginfo = gd.unpack_results(st, robot.n)
        pos, rot, its, inner, nacc, stop, execd = (ginfo[k].astype(float) for k in (
            "pos_err", "rot_err", "iterations", "inner_total", "n_accept", "stop", "inner_executed"))
        inner_local = float(res["inner_total"].double().sum())        # as the reference counts them
        exec_local = float(res["inner_executed"].double().sum())      # Hessian products evaluated
        outer_local = float(res["iterations"].double().sum())
        acc_local = float(res["n_accept"].double().sum())
        flops = algorithmic_flops(N, k, T, exec_local, outer_local, acc_local)   # executed products only
        achieved_tf = flops / (kernel_ms * 1e-3) / 1e12
        info = getattr(prob.template if anch is None else anch.template, "info", None)
        # which solve kernel the library chose (gik_template_get_info), not a re-derivation of its rules
        if info and info.get("node_per_lane"):
            kernel_name = "rtr_npt_kernel (node per lane, %d wavefront(s) per problem)" % int(info["node_per_lane"])
        elif info and info.get("is_block"):
            kernel_name = f"rtr_block_kernel<{k}>"
        elif info and info.get("problems_per_wave", 1) == 4:
            kernel_name = f"rtr_quad_kernel<{info['max_terms_per_node']}> (four planar problems per wavefront)"
        else:
            kernel_name = f"rtr_wave_kernel<{k},{info['max_terms_per_node'] if info else prob.template.maxdeg}>"
            if info and k == 3:
                kernel_name += " (per-edge product form)" if info.get("hessian_form") else " (column-form product)"
        flops_exec = executed_flops(N, k, T, info if anch is None else None, lowrank_local,
                                    exec_local, outer_local, acc_local)
        executed_tf = flops_exec / (kernel_ms * 1e-3) / 1e12
        hbm_bytes = algorithmic_bytes(N, k, T) * B
        value = total * steps / dt
        base_idx = CONFIGS[cfg][3] if cfg else None
        out = {
            "metric": "IK solves/sec (batched random goals)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{robot_name} N={N} k={k} terms={T}, {total} random goals "
                                   f"({B} on rank 0; " +
                                   (f"BASELINE configs[{base_idx}]" if base_idx is not None else "parity config")
                                   + "), reference solver defaults (mingradnorm 5e-10, maxiter 3000)",
                       "config": cfg, "robot": robot_name, "goals_total": total, "batch_per_gpu": B,
                       "seed": seed,
                       "step": ("goal poses (HBM) -> prepare kernel (goal distances, bound smoothing, "
                                "MDS init) -> RTR solve kernel -> recover kernel (joint angles, FK "
                                "pose error); no host work inside the timed region") if on_device else
                               ("RTR solve kernel (workgroup per problem) on targets / Y_init resident in "
                                "HBM; goal assembly and joint recovery run on the host outside the "
                                "timed region (graphs beyond the device pipeline, N > 128)"),
                       "parallelism": f"shard{world}", "batches_in_flight": n_streams},
            "median_pos_err_m": float(np.median(pos)), "median_rot_err_rad": float(np.median(rot)),
            "p90_pos_err_m": float(np.percentile(pos, 90)),
            "success_rate": float(np.mean((pos < 0.01) & (rot < 0.01))),
            "outer_iterations": {"median": float(np.median(its)), "max": float(its.max())},
            "hv_products": {"median": float(np.median(inner)), "max": float(inner.max()),
                            "total_per_gpu": inner_local, "executed_per_gpu": exec_local,
                            "note": "median/max/total: tCG iterations as the reference counts them; "
                                    "executed: Hessian products evaluated (a tCG solve after a rejected "
                                    "step resumes from a checkpoint, bit-identical result); roofline "
                                    "flops use executed"},
            "frac_maxiter": float(np.mean(stop == 1)),
            "roofline": {"bound": "fp64-valu", "achieved": achieved_tf, "peak": FP64_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved_tf / FP64_PEAK_TFLOPS,
                         "frac_executed": executed_tf / FP64_PEAK_TFLOPS, "traffic": None,
                         "kernel": kernel_name, "kernel_ms": kernel_ms, "kernel_ms_per_step": ks,
                         "kernel_share_of_step": kernel_ms / (dt_local / steps * 1e3),
                         "flops_per_launch": flops, "flops_executed_per_launch": flops_exec,
                         "note": "fp64 vector ALU (the contract's \"mfma\" class: compute-bound, not HBM); "
                                 "the solve is LDS/register resident and bound by the instruction issue "
                                 "rate of one wavefront per problem (and, at this batch size, by the "
                                 "slowest problem), not by HBM or MFMA (SURVEY 8(d), DESIGN 4.1); "
                                 "peak = MI355X fp64 vector/matrix spec"
                                 + ("" if not (info and info.get("is_block")) else
                                    "; graphs beyond one wavefront: `achieved` / `frac` use the ALGORITHMIC "
                                    "flop count of SURVEY 8(d) (12 k |E| per product); a rigid anchor clique "
                                    "is evaluated in closed form with far fewer operations -- "
                                    "`frac_executed` prices the flops the kernel actually executes "
                                    "(bench.py: executed_flops, DESIGN 4.2)")},
            "roofline_hbm": {"bound": "hbm", "achieved": hbm_bytes / (kernel_ms * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": hbm_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "bytes_per_launch": hbm_bytes},
        }
        out["gather"] = gather
        out["per_rank"] = [dict(zip(("ms", "max_outer", "frac_maxiter", "goals"), v)) for v in per_rank]
        if serving is not None:
            # the same flops per batch as the timed region's, S batches in flight: what the chip does on this config
            serving["frac"] = flops * serving["batches"] / serving.pop("seconds") / 1e12 / FP64_PEAK_TFLOPS
            out["serving"] = serving
        # every kernel of the step (HIP events on the launch stream), which of them dominates, and -- when it
        # is the prepare kernel (planar-10: c5) -- its own roofline entry next to the solve kernel's
        if prep_ms is not None:
            K_local = res["K"].double().cpu().numpy()
            pf = float(prepare_flops(N, K_local).sum())
            prep_name = ("prep_block_kernel" if info and info.get("prepare_is_block") else
                         "prep_quad_kernel (four goals per wavefront)" if info and info.get("goals_per_wave") == 4 else
                         "prep_wave_kernel")
            out["kernels"] = {"prepare_ms": prep_ms, "solve_ms": kernel_ms, "recover_ms": recover_ms,
                              "dominant": prep_name if prep_ms > kernel_ms else kernel_name}
            out["roofline_prepare"] = {
                "bound": "fp64-valu", "kernel": prep_name, "kernel_ms": prep_ms,
                "achieved": pf / (prep_ms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": pf / (prep_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "flops_per_launch": pf,
                "mds_columns_median": float(np.median(K_local)),
                # SURVEY 8(d) bytes of this kernel: goal pose in, per-term targets + initial point (+ the MDS column
                # count) out; `traffic` (HBM bytes per launch from the PMC passes, null if unprofiled) is filled below
                "bytes_per_launch": float(B * (8 * (k + 1) ** 2 + 8 * T + 8 * N * k + 4)),
                "traffic": None,
                "note": "goal distances + bound smoothing + MDS initial point, one wavefront (workgroup) per goal; "
                        "SURVEY 8(d): F_init = 2 * 9 N^3 + 9 K^3 per goal with the K the kernel reports, + 4 N^3 for "
                        "Floyd-Warshall (UPPER) and the max-plus pass (LOWER); LDS resident, bound by instruction "
                        "issue of the cyclic Jacobi sweeps (DESIGN 4.3)"}
        else:
            out["kernels"] = {"prepare_ms": None, "solve_ms": kernel_ms, "recover_ms": None, "dominant": out["roofline"]["kernel"]}
        # HBM traffic of the solve kernel from the PMC passes of the SAME workload (tools/profile.sh:
        # rocprofv3 cannot run inside this process), newest round first; null for unprofiled workloads
        tags = {("lwa4d", 4096): ["r06", "r05", "r04", "r03", "r02"], ("ur10_table", 4096): ["r06_c3", "r05_c3", "r04_c3"],
                ("kuka", 65536): ["r06_c4", "r05_c4", "r04_c4", "r03_c4"],
                ("kuka", 8192): ["r06_c4share", "r05_c4share", "r04_c4share", "r03_c4share"],
                ("planar10", 65536): ["r06_c5", "r05_c5", "r04_c5", "r03_c5", "r02_c5"]}
        if hessian_form == "column":      # (the column-form comparison lines: their own profiles, if any)
            tags = {kk: [t + "_column" for t in v] for kk, v in tags.items()}
        out["roofline"]["traffic_measured"] = None
        lib_digest = library_digest()
        for tag in ([] if intended else tags.get((robot_name, B), [])):
            traffic_file = os.path.join(REPO, "profiles", "hbm_traffic.json" if tag == "r02" else f"{tag}_hbm_traffic.json")
            if not os.path.exists(traffic_file):
                continue
            try:
                tj = json.load(open(traffic_file))
                src = (f"profiles/{os.path.basename(traffic_file)}: rocprofv3 --pmc passes of this workload in a "
                       "separate run (tools/profile.sh + tools/summarize_prof.py), not measured by the process "
                       "that printed this line")
                out["roofline"]["traffic"] = tj.get("bytes_per_launch")
                out["roofline"]["traffic_ratio"] = tj.get("bytes_per_launch") / hbm_bytes if hbm_bytes else None
                out["roofline"]["traffic_source"] = src
                out["roofline"]["traffic_measured"] = "profiles (separate rocprofv3 run), not in-process"
                # counters are a property of the BINARY they were read from: the profile carries the digest of the
                # library's sources (tools/summarize_prof.py); anything else is an old measurement and says so
                out["roofline"]["traffic_stale"] = tj.get("source_digest") != lib_digest
                pj = tj.get("prepare")
                if pj and "roofline_prepare" in out:
                    rp = out["roofline_prepare"]
                    rp["traffic"] = pj.get("bytes_per_launch")
                    rp["traffic_ratio"] = pj.get("bytes_per_launch") / rp["bytes_per_launch"]
                    rp["lds_bank_conflict_ratio"] = pj.get("lds_bank_conflict_ratio")
                    rp["traffic_source"] = src
                    rp["traffic_stale"] = out["roofline"]["traffic_stale"]
                break
            except Exception:
                pass

        if anch is not None:
            Yh = res["x"].cpu().numpy()
            clear = anch.clearance(Yh)
            conv = (res["f"].cpu().numpy() < 1e-9)
            out["intended"] = {
                "formulation": "fixed anchors (base, goal nodes, obstacle centres are constants) + robot<->obstacle "
                               "lower hinges; opt-in, NOT the reference's observable semantics (SURVEY 8(f)3)",
                "free_nodes": N, "obstacles": int(len(anch.obstacles)),
                "converged_frac": float(conv.mean()),
                "collision_free_frac_of_converged": float((clear[conv] > -1e-4).mean()) if conv.any() else None,
                "min_clearance_of_converged_m": float(clear[conv].min()) if conv.any() else None}
            out["config"]["workload"] += " [--intended: fixed-anchor formulation, not the BASELINE semantics]"
            out["roofline"]["kernel"] = "rtr_wave_kernel<3,9,anchored>"
        if not use_limits:
            out["config"]["workload"] += " [use_limits=False: the j* kernels, costs.py:8-58]"
        if cpu and not args.no_cpu_baseline and world == 1 and anch is None:   # rank 0, single-GPU runs
            out["cpu_baseline"] = cpu_baseline(prob, T_goal, Y0_h, B, args)
        return out


def brief(o):
    """The sub-record of a BASELINE config inside the default line's "configs" object."""
    if o is None:
        return None
    if o.get("dry_solve"):
        return {k: o[k] for k in ("rows", "rows_sha", "scaling", "ms_per_step")} | \
            {"goals_total": o["config"]["goals_total"]}
    r = o["roofline"]
    return {"value": o["value"], "unit": "solves/s", "ms_per_step": o["ms_per_step"], "steps": o["steps"],
            "warmup": o["warmup"], "scaling": o["scaling"], "workload": o["config"]["workload"],
            "goals_total": o["config"]["goals_total"], "batch_per_gpu": o["config"]["batch_per_gpu"],
            "kernel": r["kernel"], "kernel_ms": r["kernel_ms"],
            "roofline": {"bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"],
                         "frac": r["frac"], "frac_executed": r["frac_executed"], "traffic": r["traffic"],
                         "traffic_ratio": r.get("traffic_ratio"), "traffic_measured": r.get("traffic_measured")},
            "success_rate": o["success_rate"], "frac_maxiter": o["frac_maxiter"],
            "median_pos_err_m": o["median_pos_err_m"], "median_rot_err_rad": o["median_rot_err_rad"],
            "outer_iterations": o["outer_iterations"],
            "hv_products_executed_per_gpu": o["hv_products"]["executed_per_gpu"],
            "kernels": o.get("kernels"), "roofline_prepare": o.get("roofline_prepare"),
            "cpu_baseline": ({k: o["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "sample",
                                                                 "hv_per_s_per_thread")}
                             if "cpu_baseline" in o else None)}


def main():
    args = parse_args()
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(self_launch(args))

    b = Bench(args)
    cfg, robot_name, total, scaling = workload(args, b.world)
    out = b.measure(cfg, robot_name, total, scaling, args.steps, args.warmup,
                    serving_streams=args.serving_streams, cpu=True, n_streams=args.streams,
                    intended=args.intended)
    # The default invocation (`python bench.py [--gpus N]`, what the driver runs) also times the other
    # BASELINE workloads after the headline, a few steps each, and reports them under "configs" in
    # the same line: c3 (UR10 + table, 4096 goals per GPU), c4 (KUKA, 65536 goals sharded over the
    # GPUs: strong scaling; on one GPU also its 8192-goal per-GPU share of an 8-GPU run) and c5
    # (planar-10, 65536 sharded).  --headline-only skips them.
    default_run = (args.config is None and args.robot is None and not args.batch and not args.intended
                   and args.streams == 1 and not args.headline_only)
    if default_run:
        extra = {}
        plan = [("c3", "ur10_table", 4096 * b.world, "weak", 2, 1),
                ("c4", "kuka", 65536, "strong", 3, 1),
                ("c5", "planar10", 65536, "strong", 5, 1),
                ("c5_nolimits", "planar10", 65536, "strong", 5, 1)]      # use_limits=False & True
        if b.world == 1:
            plan.insert(2, ("c4_share_of_8", "kuka", 8192, "strong", 5, 1))
        # the column-form product beside the default (per-edge) lines of the two configs it concerns
        plan += [("c2_column", robot_name, total, scaling, 3, 1), ("c4_column", "kuka", 65536, "strong", 3, 1)]
        for name, rb, tot, sc, st, wu in plan:
            o = b.measure(name[:2], rb, tot, sc, st, wu, cpu=(name != "c4_share_of_8" and not name.endswith("_column")),
                          use_limits=(name != "c5_nolimits"), hessian_form=("column" if name.endswith("_column") else None))
            if b.rank == 0:
                extra[name] = brief(o)
        # the headline on seeds 0-3: `value` stays seed 0; a 4096-goal batch is as long as
        # the longest problem of its draw
        per_seed = [out["ms_per_step"] if b.rank == 0 else None]
        for sd in (1, 2, 3):
            o = b.measure(cfg, robot_name, total, scaling, 3, 1, seed=sd)
            per_seed.append(o["ms_per_step"] if b.rank == 0 else None)
        if b.rank == 0:
            out["configs"] = extra
            out["seeds"] = {"seeds": [0, 1, 2, 3], "ms_per_step": per_seed, "median_ms_per_step": float(np.median(per_seed)),
                            "value_at_median": total / (float(np.median(per_seed)) * 1e-3),
                            "note": "same workload, goal streams of seeds 0-3 (seed 0 = `value`, steps as given; seeds 1-3: "
                                    "3 steps after 1 warm-up)"}
    b.gd.shutdown()
    if b.rank == 0:
        # LAST key of the line: one compact record per BASELINE config (the driver keeps the tail of a long line)
        def tiny(o, r=None):
            r = r or o.get("roofline") or {}
            rp = o.get("roofline_prepare") or {}
            kn = o.get("kernels") or {}
            sig = lambda v: None if v is None else float("%.4g" % v)
            return {"value": sig(o.get("value")), "ms": sig(o.get("ms_per_step")), "frac": sig(r.get("frac")),
                    "frac_exec": sig(r.get("frac_executed")), "traffic_x": sig(r.get("traffic_ratio")),
                    "prep_ms": sig(kn.get("prepare_ms")), "solve_ms": sig(kn.get("solve_ms")),
                    "prep_frac": sig(rp.get("frac")), "prep_traffic_x": sig(rp.get("traffic_ratio")),
                    "cpu": sig((o.get("cpu_baseline") or {}).get("value"))}
        summ = {cfg if cfg else "custom": tiny(out)}
        if out.get("serving"):      # sixteen batches in flight: what the chip does on this config (NOT `value`)
            summ[cfg if cfg else "custom"].update(serving=float("%.4g" % out["serving"]["value"]),
                                                  serving_frac=float("%.4g" % out["serving"]["frac"]))
        for name, o in (out.get("configs") or {}).items():
            if o and "value" in o:
                summ[name] = tiny(o)
        out["summary"] = {"unit": "solves/s; ms per step; roofline fractions of the fp64 vector peak (algorithmic / executed); "
                                  "traffic_x = HBM bytes per launch (PMC, profiles/) / algorithmic bytes; cpu = oracle solves/s; "
                                  "serving = solves/s with 16 batches in flight; *_column = the same config with "
                                  "hessian_form = column (the default is the per-edge product form)",
                          "traffic_stale": bool((out.get("roofline") or {}).get("traffic_stale")),
                          "traffic_measured": "profiles/ (separate rocprofv3 --pmc runs), not in-process", **summ}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
