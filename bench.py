#!/usr/bin/env python3
"""bench.py -- IK solves/s of the MI355X-native GraphIK RiemannianSolver path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): Schunk LWA4D (N=18 graph nodes, k=3, 75 residual terms),
4096 random goals per GPU (weak scaling), goals = FK of uniform random configurations in
[-pi, pi]^7 exactly as the reference's examples draw them (robot.random_configuration()).
One "step" = one pass of the hot path over one batch with its inputs resident in HBM.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402

from graphik_amd import distributed as gd  # noqa: E402

FP64_PEAK_TFLOPS = 78.6   # MI355X vector = matrix fp64 peak (spec; fp32 vector 157.3 / 2)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_flops(N, k, T, n_inner, n_outer, n_accept):
    """F_solve = n_inner*F_hv + n_outer*F_out (+ F_acc per accepted step)."""
    F_hv = 12 * k * T + (4 * N * k * k + k * k + 2 * k ** 4) + 16 * N * k
    F_cost = (3 * k + 6) * T + 5 * N * k
    F_acc = 9 * k * T + N * k * (k + 1) + (2.0 / 3.0) * k ** 6
    return n_inner * F_hv + n_outer * F_cost + n_accept * F_acc


def algorithmic_bytes(N, k, T):
    """HBM bytes the solve kernel must move per IK problem: T targets + Y_init in, Y_sol + stats
    out (SURVEY 8(d) counts the pipeline-level 616 B/solve; the dominant kernel alone sees this)."""
    return 8 * (T + 2 * N * k) + 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0,
                    help="goals per GPU (default 4096; 256 for ur10_table, whose goal assembly runs "
                         "on the host)")
    ap.add_argument("--robot", default="lwa4d",
                    choices=["lwa4d", "ur10", "kuka", "planar10", "planar10_halfpi", "ur10_table"],
                    help="lwa4d = BASELINE configs[1] (the bench line); the others are the parity "
                         "configs, runnable here for reference numbers")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="problems for the CPU baseline (0=auto)")
    ap.add_argument("--streams", type=int, default=1,
                    help="batches in flight: steps are issued round-robin on this many HIP streams, so the "
                         "straggler tail of one batch overlaps with the bulk of the next (serving mode; "
                         "the default 1 runs the steps back to back and is the headline configuration)")
    args = ap.parse_args()

    rank, local_rank, world = gd.init_process_group()
    assert world == args.gpus or world == 1 and args.gpus == 1, (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from graphik_amd.utils.roboturdf import load_schunk_lwa4d, load_ur10, load_kuka
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    if args.robot.startswith("planar10"):
        from graphik_amd.robots import RobotPlanar
        from graphik_amd.graphs import ProblemGraphPlanar
        from graphik_amd.utils import list_to_variable_dict
        nl = 10
        lim = np.array(9 * [np.pi / 2] + [np.pi]) if args.robot.endswith("halfpi") else np.pi * np.ones(nl)
        robot = RobotPlanar({"link_lengths": list_to_variable_dict(np.ones(nl)),
                             "theta": list_to_variable_dict(np.zeros(nl)),
                             "joint_limits_upper": list_to_variable_dict(lim),
                             "joint_limits_lower": list_to_variable_dict(-lim), "num_joints": nl})
        graph = ProblemGraphPlanar(robot)
    elif args.robot == "ur10_table":
        # BASELINE configs[2]: UR10 + table_environment() (N = 116, 5612 terms): the
        # workgroup-per-goal prepare kernel and the workgroup-per-problem solve kernel
        from graphik_amd.utils import table_environment
        robot, graph = load_ur10()
        for idx, obs in enumerate(table_environment()):
            graph.add_spherical_obstacle(f"o{idx}", obs[0], obs[1])
    else:
        robot, graph = {"lwa4d": load_schunk_lwa4d, "ur10": load_ur10, "kuka": load_kuka}[args.robot]()
    prob = BatchProblem(graph, use_limits=True, device=dev)
    B = args.batch or (256 if args.robot == "ur10_table" else 4096)
    N, k, T, n = graph.number_of_nodes(), graph.dim, prob.template.T, robot.n

    # synthetic goals: rank r draws rows [r*B, (r+1)*B) of one global stream
    rs = np.random.RandomState(args.seed)
    U = rs.rand(world * B, n)[rank * B:(rank + 1) * B]
    lb, ub = robot.limits_arrays()
    Q = lb + (ub - lb) * U
    T_goal = robot.fk_batch(Q)
    tpl = prob.template
    on_device = prob.device_pipeline
    Tg_dev = torch.from_numpy(T_goal).to(dev)        # inputs resident in HBM
    if on_device:
        bufs = tpl.alloc_ik_buffers(B)
    else:
        tg_h, Y0_h0 = prob.prepare(T_goal)
        tg_dev, Y0_dev = torch.from_numpy(tg_h).to(dev), torch.from_numpy(Y0_h0).to(dev)
    torch.cuda.synchronize(dev)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(i=None):
        """goal poses -> joint angles + pose errors, entirely on the device: prepare
        (from_pose, bound smoothing, MDS init) -> RTR solve -> recover (joint_variables, FK)."""
        targets, Y0 = tpl.prepare(Tg_dev) if on_device else (tg_dev, Y0_dev)
        if i is not None:
            ev0[i].record()      # all kernels are launched on torch's current stream
        res = tpl.solve(Y0, targets)
        if i is not None:
            ev1[i].record()
        if on_device:
            q, pe, re = tpl.recover(res["x"], Tg_dev)
            res.update(q=q, pos_err=pe, rot_err=re)
        res.update(Y0=Y0)
        return res

    for _ in range(args.warmup):
        res = step()
    torch.cuda.synchronize(dev)
    gd.barrier()
    torch.cuda.synchronize(dev)
    streams = [torch.cuda.Stream(dev) for _ in range(args.streams)] if args.streams > 1 else None
    t0 = time.perf_counter()
    for i in range(args.steps):
        if streams is None:
            res = step(i)
        else:                      # every step is the same full batch; S of them are in flight
            with torch.cuda.stream(streams[i % args.streams]):
                res = step(i)
    torch.cuda.synchronize(dev)
    gd.barrier()
    dt_local = time.perf_counter() - t0
    dt = gd.max_over_ranks(dt_local, dev)
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    if not on_device:   # host post-processing, after the timed region
        qh = prob.joint_variables(res["x"].cpu().numpy(), T_goal)
        pe, re = prob.pose_errors(qh, T_goal)
        res.update(pos_err=torch.from_numpy(pe).to(dev), rot_err=torch.from_numpy(re).to(dev))

    # single gather of the per-problem results at the end (RCCL over xGMI when N > 1)
    stats_local = torch.stack([res["pos_err"], res["rot_err"], res["iterations"].double(),
                               res["inner_total"].double(), res["n_accept"].double(),
                               res["stop"].double()], dim=1)
    allstats = gd.gather_rows(stats_local, world * B, dst=0)
    Y0_h = res["Y0"].cpu().numpy()

    if rank != 0:
        gd.shutdown()
        return
    st = allstats.cpu().numpy()
    pos, rot, its, inner, nacc, stop = st.T
    inner_local = float(res["inner_total"].double().sum())        # as the reference counts them
    exec_local = float(res["inner_executed"].double().sum())      # Hessian products evaluated
    outer_local = float(res["iterations"].double().sum())
    acc_local = float(res["n_accept"].double().sum())
    flops = algorithmic_flops(N, k, T, exec_local, outer_local, acc_local)   # executed work only
    achieved_tf = flops / (kernel_ms * 1e-3) / 1e12
    hbm_bytes = algorithmic_bytes(N, k, T) * B
    value = world * B * args.steps / dt
    out = {
        "metric": "IK solves/sec (batched random goals)",
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.robot} N={N} k={k} terms={T}, {B} random goals per GPU "
                               "(" + ("BASELINE configs[1]" if args.robot == "lwa4d" else "parity config") + "), reference solver defaults "
                               "(mingradnorm 5e-10, maxiter 3000)",
                   "robot": args.robot, "batch_per_gpu": B, "seed": args.seed,
                   "step": ("goal poses (HBM) -> prepare kernel (goal distances, bound smoothing, "
                            "MDS init) -> RTR solve kernel -> recover kernel (joint angles, FK "
                            "pose error); no host work inside the timed region") if on_device else
                           ("RTR solve kernel (workgroup per problem) on targets / Y_init resident in "
                            "HBM; goal assembly and joint recovery run on the host outside the "
                            "timed region (graphs beyond the device pipeline, N > 128)"),
                   "parallelism": f"shard{world}", "batches_in_flight": args.streams},
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
        "roofline": {"bound": "mfma", "achieved": achieved_tf, "peak": FP64_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved_tf / FP64_PEAK_TFLOPS, "traffic": None,
                     "kernel": (f"rtr_wave_kernel<{k},{prob.template.maxdeg}>" if N * k <= 64
                                else f"rtr_block_kernel<{k}>"), "kernel_ms": kernel_ms,
                     "kernel_share_of_step": kernel_ms / (dt_local / args.steps * 1e3),
                     "flops_per_launch": flops,
                     "note": "fp64; the solve is LDS/register resident and bound by the instruction "
                             "issue rate of one wavefront per problem (and, at this batch size, by "
                             "the slowest problem), not by HBM or MFMA (SURVEY 8(d), DESIGN 4.1); "
                             "peak = MI355X fp64 vector/matrix spec"},
        "roofline_hbm": {"bound": "hbm", "achieved": hbm_bytes / (kernel_ms * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": hbm_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "bytes_per_launch": hbm_bytes},
    }
    traffic_file = os.path.join(REPO, "profiles", "hbm_traffic.json")
    if os.path.exists(traffic_file) and args.robot == "lwa4d" and B == 4096:   # profiled workload only
        try:
            out["roofline"]["traffic"] = json.load(open(traffic_file)).get("bytes_per_launch")
        except Exception:
            pass

    if not args.no_cpu_baseline and world == 1:   # rank 0, single-GPU runs only
        from oracle import c_oracle as co
        cores = os.cpu_count() or 1
        ns = args.cpu_sample or min(B, max(32, 8 * cores))
        D, _, _ = prob.assemble(T_goal[:ns])
        t0 = time.perf_counter()
        o = co.rtr_solve_batch(Y0_h[:ns], D, prob.omega, prob.psi_L, prob.psi_U, True,
                               nthreads=cores, fast=True)
        tc = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": ns / tc, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": f"first {ns} goals of rank 0's batch, oracle/gik_oracle.c (-O3 AVX2+FMA), "
                      f"OpenMP over problems, {tc:.1f} s wall",
            "hv_total": int(o["inner_total"].sum())}
    gd.shutdown()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
