"""BASELINE.json workloads at BASELINE size through the whole device pipeline (configs[2]: UR10 +
table, 4096 goals; configs[3]: KUKA, the 8192-goal per-GPU share of 65536 over 8 GPUs; configs[4]:
planar-10, 8192 and 65536), concurrent batches on one handle, and the RCCL gather.  Size-independent
properties in the mould of test_full_size_batch_properties: every problem ends by a legal stopping
rule, results are finite and bit-identical across reruns, the recovered configurations realise the
goal pose, and a sub-sample agrees with the CPU oracle started from the same initial points."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, make_graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _goals(robot, B, seed):
    """bench.py's goal stream: FK of uniform random configurations within the joint limits."""
    rs = np.random.RandomState(seed)
    lb, ub = robot.limits_arrays()
    return robot.fk_batch(lb + (ub - lb) * rs.rand(B, robot.n))


def _pipeline_twice(torch, prob, Tg):
    """prepare -> solve -> recover on the device, twice; returns numpy results of run 1 after
    asserting that run 2 reproduced every output bit for bit."""
    tpl = prob.template
    runs = []
    for _ in range(2):
        tg, Y0 = tpl.prepare(Tg)
        r = tpl.solve(Y0, tg)
        q, pe, re = tpl.recover(r["x"], Tg)
        torch.cuda.synchronize()
        runs.append({"tg": tg.cpu().numpy(), "Y0": Y0.cpu().numpy(), "q": q.cpu().numpy(),
                     "pos": pe.cpu().numpy(), "rot": re.cpu().numpy(),
                     **{k: r[k].cpu().numpy() for k in ("x", "f", "gradnorm", "iterations", "inner_total",
                                                         "inner_executed", "stop", "n_accept", "flags")}})
    for k in runs[0]:
        # (inner_executed / flags of a batch beyond the resident waves depend on which problems the
        # tail spreading moves, i.e. on timing; every result and every reference-defined counter is
        # reproducible)
        if k not in ("inner_executed", "flags"):
            assert np.array_equal(runs[0][k], runs[1][k], equal_nan=True), k
    return runs[0]


def _legal_stops(r, maxiter=3000, mingradnorm=0.5e-9, sliced=False):
    stop, its, gn = r["stop"], r["iterations"], r["gradnorm"]
    assert np.all(np.isfinite(r["x"])) and np.all(np.isfinite(r["q"]))
    assert np.all((stop == 0) | (stop == 1))                       # gradnorm or maxiter, never NaN
    assert np.all(gn[stop == 0] < mingradnorm) and np.all(its[stop == 1] == maxiter)
    assert np.all(its[stop == 0] <= maxiter)
    assert np.all(r["n_accept"] <= r["iterations"])
    # executed Hessian products: fewer than the reference counts overall (checkpoint resume), although a
    # single problem may run a few more ("model increased" exits cost one product more than they
    # report; a time-slice boundary drops the checkpoint)
    assert r["inner_executed"].sum() <= r["inner_total"].sum()
    if not sliced:
        assert np.all(r["inner_executed"] <= r["inner_total"] + r["iterations"])


def test_full_size_c3_ur10_table(torch_cuda):
    """BASELINE configs[2]: UR10 + table_environment() (N = 116, 5612 terms), 4096 random goals on
    the workgroup-per-goal prepare kernel and the node-per-lane solve kernel (two wavefronts per problem,
    rigid clique in closed form, time slicing: 4096 problems on 512 resident workgroups)."""
    from oracle import c_oracle as co
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph("ur10_table")
    prob = BatchProblem(graph, use_limits=True)
    assert prob.device_pipeline and prob.template.info["is_block"] == 1
    assert prob.template.info["node_per_lane"] == 2     # the node-per-lane kernel, two wavefronts per problem
    assert prob.template.info["n_clique"] == 106 and prob.template.info["n_slot_terms"] == 47
    B = 4096
    Tg = _goals(robot, B, 0)
    r = _pipeline_twice(torch_cuda, prob, Tg)
    _legal_stops(r, sliced=True)
    assert np.all(r["flags"] & 1)            # Euclidean targets: (D w) by moments for every goal
    ok = (r["pos"] < 0.01) & (r["rot"] < 0.01)
    assert ok.mean() > 0.88 and np.median(r["pos"]) < 1e-3          # measured 0.929, 2.7e-4
    assert (r["stop"] == 1).mean() < 0.02                            # measured 0.0054
    # the recover kernel's pose error is the error of FK(q) (host FK of the device's angles)
    pos_h, rot_h = prob.pose_errors(r["q"][:256], Tg[:256])
    assert np.allclose(pos_h, r["pos"][:256], atol=1e-9) and np.allclose(rot_h, r["rot"][:256], atol=1e-7)
    # The TAIL on the scene itself (the batch time of this config IS its tail): the 64 longest goals of the run
    # -- every one of the 23 that stop at maxiter among them -- and 64 random ones, solved by the oracle from the device's
    # start points (1-25 s each per thread: ~2 min on the box).  Measured round 6 (tools/c3_tail.py): all 23 maxiter goals
    # are maxiter goals of the oracle, none other; convergence class equal on 128 / 128; outer iterations of the goals
    # that converge in both 1.003x (longest) / 0.997x (random); Hessian products +4.7 % / +3.5 % -- the closed form of the
    # 106-anchor clique sums the same terms through 24 moments, and tCG's iteration count feels the different round-off
    # (a graph without a clique on the same kernel: +1.1 %, test_effort_parity_kuka_tail[npt]).
    its, hv = r["iterations"], r["inner_total"].astype(np.int64)
    longest = np.argsort(-its, kind="stable")[:64]
    rnd = np.random.RandomState(1).choice(np.setdiff1d(np.arange(B), longest), 64, replace=False)
    idx = np.concatenate([longest, rnd])
    D, _, _ = prob.assemble(Tg[idx])
    o = co.rtr_solve_batch(r["Y0"][idx], D, prob.omega, prob.psi_L, prob.psi_U, True, fast=False)
    oi, oh = o["iterations"], o["inner_total"].astype(np.int64)
    mx, mxo = its[idx] >= 3000, oi >= 3000
    assert (its >= 3000).sum() <= 64 and mxo.sum() >= 0.9 * mx.sum()        # the sample holds the whole maxiter tail
    assert np.mean(mx == mxo) >= 0.95 and np.mean((r["f"][idx] < 1e-9) == (o["f(x)"] < 1e-9)) >= 0.95
    assert np.percentile(its[idx], 90) <= 1.10 * np.percentile(oi, 90)
    both = ~mx & ~mxo
    assert 0.97 < its[idx][both].sum() / oi[both].sum() < 1.03, its[idx][both].sum() / oi[both].sum()
    assert 0.9 < np.median(its[idx][64:]) / np.median(oi[64:]) < 1.1
    assert 0.97 < hv[idx].sum() / oh.sum() < 1.06, hv[idx].sum() / oh.sum()


def test_full_size_c4_kuka_share(torch_cuda):
    """BASELINE configs[3]: KUKA iiwa, the 8192-goal share one GPU gets of 65536 goals over 8 GPUs
    (rows 0..8191 of the global goal stream = rank 0's shard in bench.py)."""
    from oracle import c_oracle as co
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph("kuka")
    prob = BatchProblem(graph, use_limits=True)
    rs = np.random.RandomState(0)
    lb, ub = robot.limits_arrays()
    Tg = robot.fk_batch(lb + (ub - lb) * rs.rand(65536, robot.n)[:8192])
    r = _pipeline_twice(torch_cuda, prob, Tg)
    _legal_stops(r)
    ok = (r["pos"] < 0.01) & (r["rot"] < 0.01)
    assert ok.mean() > 0.95 and np.median(r["pos"]) < 5e-4          # measured 0.981, 2.0e-4
    assert 0.04 < (r["stop"] == 1).mean() < 0.12                     # measured 0.081 (the reference: 2 of 8)
    # oracle sub-sample from the device's start points: 256 goals (round 6: the 0.6-1.6 / 0.8-1.25 bands of rounds 2-5
    # dated from before the causes of the effort differences were known; ~3.5 % of KUKA goals change their maxiter class
    # between any two correct renderings, in both directions: test_effort_parity_kuka_tail)
    n = 256
    D, _, _ = prob.assemble(Tg[:n])
    o = co.rtr_solve_batch(r["Y0"][:n], D, prob.omega, prob.psi_L, prob.psi_U, True, fast=False)
    assert np.mean((r["f"][:n] < 1e-9) == (o["f(x)"] < 1e-9)) >= 0.94
    assert 0.9 < np.median(r["iterations"][:n]) / np.median(o["iterations"]) < 1.1
    both = (r["iterations"][:n] < 3000) & (o["iterations"] < 3000)
    assert 0.96 < r["inner_total"][:n][both].sum() / o["inner_total"][both].sum() < 1.06


def test_full_size_c4_kuka_whole_batch(torch_cuda):
    """BASELINE configs[3] as one GPU sees it when it is alone: all 65536 KUKA goals (round-robin slicing and
    tail spreading on, ~5 hand-overs per long problem), with an oracle sub-sample of its own: 96 goals spread
    over the batch (every 683rd), solved on the CPU from the device's initial points."""
    from oracle import c_oracle as co
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph("kuka")
    prob = BatchProblem(graph, use_limits=True)
    B = 65536
    Tg = _goals(robot, B, 0)
    r = _pipeline_twice(torch_cuda, prob, Tg)
    _legal_stops(r, sliced=True)
    ok = (r["pos"] < 0.01) & (r["rot"] < 0.01)
    assert ok.mean() > 0.95 and np.median(r["pos"]) < 5e-4
    assert 0.04 < (r["stop"] == 1).mean() < 0.12
    idx = np.arange(0, B, 683)[:96]
    D, _, _ = prob.assemble(Tg[idx])
    o = co.rtr_solve_batch(r["Y0"][idx], D, prob.omega, prob.psi_L, prob.psi_U, True, fast=False)
    assert np.mean((r["f"][idx] < 1e-9) == (o["f(x)"] < 1e-9)) >= 0.92
    assert 0.85 < np.median(r["iterations"][idx]) / np.median(o["iterations"]) < 1.15
    both = (r["iterations"][idx] < 3000) & (o["iterations"] < 3000)
    assert 0.95 < r["inner_total"][idx][both].sum() / o["inner_total"][both].sum() < 1.07
    # the 8192-goal share of an 8-GPU run is rows 0..8191 of the same stream: same answers in either batch
    r8 = prob.template.ik(torch_cuda.from_numpy(Tg[:8192]).cuda())
    assert np.array_equal(r8["x"].cpu().numpy(), r["x"][:8192]) and np.array_equal(r8["q"].cpu().numpy(), r["q"][:8192])


@pytest.mark.parametrize("B", [8192, 65536])
@pytest.mark.parametrize("name", ["planar10_limits_pi", "planar10_limits_halfpi", "planar10_nolimits"])
def test_full_size_c5_planar(torch_cuda, name, B):
    """BASELINE configs[4]: 10-link planar chain (limits +-pi as in test_chain_2d_new.py, and the
    +-pi/2 variant of test_chain_2d_limits_new.py), the per-GPU share of an 8-GPU run and the whole
    65536-goal batch on one GPU.  Planar solves do not amplify round-off, so the oracle sub-sample
    is compared decision for decision (iteration and Hessian-product counts)."""
    from oracle import c_oracle as co
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph(name)
    use_lim = not name.endswith("nolimits")     # nolimits: the j* kernels (jcost / jgrad / jhess, costs.py:8-58), the
    prob = BatchProblem(graph, use_limits=use_lim)      # default of test_chain_2d_new.py:59
    assert (prob.psi_L is None) == (not use_lim)
    Tg = _goals(robot, B, 0)
    r = _pipeline_twice(torch_cuda, prob, Tg)
    _legal_stops(r)
    ok = (r["pos"] < 0.01) & (r["rot"] < 0.01)
    if name.endswith("_pi") or not use_lim:
        # the reference's own acceptance test (test_chain_2d_new.py:82): every EE position error < 1e-4
        assert ok.mean() > 0.999 and np.percentile(r["pos"], 99) < 1e-4
    else:
        assert ok.mean() > 0.9
    n = 128
    D, _, _ = prob.assemble(Tg[:n])
    o = co.rtr_solve_batch(r["Y0"][:n], D, prob.omega, prob.psi_L, prob.psi_U, use_lim, fast=False)
    same = (r["iterations"][:n] == o["iterations"]) & (r["inner_total"][:n] == o["inner_total"])
    assert same.mean() > 0.9, same.mean()
    assert np.mean((r["f"][:n] < 1e-9) == (o["f(x)"] < 1e-9)) > 0.97


def test_tail_spreading_and_round_robin_are_bit_identical(torch_cuda, monkeypatch):
    """Wavefront kernel, batches beyond the resident waves (two waves per SIMD).  Two schedulers
    ride on the exactly resumable state (x, Delta, counters): round-robin time slicing while more
    problems are unfinished than there are waves (a problem gives up its slot after a slice and
    queues behind the others), and in the tail a wave whose SIMD hosts two long-running problems
    hands its problem to a wave that waits on an empty SIMD.  Every output must equal the run with
    debug_flags = 512 (neither) bit for bit -- also with slices of 4 outer iterations, i.e. ~10^5
    hand-overs --, hand-overs must actually happen, and a batch in which every problem is long
    (maxiter-bound start points) must terminate."""
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph("kuka")
    rs = np.random.RandomState(0)
    lb, ub = robot.limits_arrays()
    Tg = robot.fk_batch(lb + (ub - lb) * rs.rand(65536, robot.n)[:8192])
    keys = ("x", "f", "gradnorm", "iterations", "inner_total", "stop", "n_accept", "stepsize")
    out = {}
    monkeypatch.setenv("GIK_SLICE_CYCLES", "0")      # (read at creation: no lower bound on a slice's duration)
    for name, params in (("default", None), ("tiny", {"slice_outer_its": 4}), ("spread", {"debug_flags": 1024}),
                         ("plain", {"debug_flags": 512})):
        prob = BatchProblem(graph, use_limits=True, params=params)
        tg, Y0 = prob.template.prepare(Tg)
        r = prob.template.solve(Y0, tg)
        torch_cuda.cuda.synchronize()
        out[name] = {k: r[k].cpu().numpy() for k in keys + ("flags", "inner_executed")}
    assert not (out["plain"]["flags"] & 2).any()
    for name, least in (("default", 2000), ("tiny", 100000), ("spread", 20)):   # measured 6.5 k / 135 k / 450
        for k in keys:
            assert np.array_equal(out[name][k], out["plain"][k], equal_nan=True), (name, k)
        moved = (out[name]["flags"] & 2) != 0
        # (a hand-over drops the tCG checkpoint: a moved problem may execute a few products more)
        assert np.array_equal(out[name]["inner_executed"][~moved], out["plain"]["inner_executed"][~moved])
        assert np.all(out[name]["inner_executed"][moved] >= out["plain"]["inner_executed"][moved])
        handovers = int((out[name]["flags"] >> 8).sum())
        assert handovers >= least and moved.sum() >= 20, (name, handovers, moved.sum())
    monkeypatch.delenv("GIK_SLICE_CYCLES")
    # 4096 copies of three slow goals: nothing finishes early, every wave stays busy to the end
    slow = np.argsort(-out["plain"]["iterations"])[:3]
    prob = BatchProblem(graph, use_limits=True, params={"maxiter": 300})
    tg, Y0 = prob.template.prepare(Tg[np.tile(slow, 1400)])
    r = prob.template.solve(Y0, tg)
    torch_cuda.cuda.synchronize()
    x = r["x"].cpu().numpy().reshape(1400, 3, -1)
    assert np.array_equal(x, np.broadcast_to(x[0], x.shape))


# ---- concurrent batches on ONE handle (the "serving" figure of bench.py) ----------------------------
def _serial_and_concurrent(torch, run, n_batches, n_streams):
    """run(i) issues batch i on torch's current stream and returns a dict of device tensors.
    Serial reference first (default stream, synchronised), then all batches round-robin on
    n_streams HIP streams without any synchronisation in between."""
    serial = []
    for i in range(n_batches):
        out = run(i)
        torch.cuda.synchronize()
        serial.append({k: v.cpu().numpy() for k, v in out.items()})
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    torch.cuda.synchronize()
    live = []
    for rep in range(2):                       # two waves of submissions, nothing waits in between
        for i in range(n_batches):
            with torch.cuda.stream(streams[(i + rep) % n_streams]):
                live.append((i, run(i)))
    torch.cuda.synchronize()
    for i, out in live:
        for k, v in out.items():
            assert np.array_equal(serial[i][k], v.cpu().numpy(), equal_nan=True), (i, k)


_KEYS = ("x", "f", "gradnorm", "iterations", "inner_total", "inner_executed", "stop", "n_accept")


def test_concurrent_batches_wave_path(torch_cuda):
    """16 different 1024-goal LWA4D batches, 8 streams, one template: prepare -> solve -> recover of
    every batch must reproduce its serial result bit for bit (work-queue counters, per-call buffers)."""
    torch = torch_cuda
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph("lwa4d")
    prob = BatchProblem(graph, use_limits=True)
    tpl = prob.template
    Tgs = [torch.from_numpy(_goals(robot, 1024, 100 + i)).cuda() for i in range(16)]

    def run(i):
        r = tpl.ik(Tgs[i])
        return {k: r[k] for k in _KEYS + ("q", "pos_err", "rot_err")}

    _serial_and_concurrent(torch, run, 16, 8)


@pytest.mark.parametrize("path", [1, 2])      # workgroup kernels / node-per-lane kernel
def test_concurrent_batches_block_path_with_time_slicing(torch_cuda, monkeypatch, path):
    """Workgroup-per-problem kernels on one handle from 8 streams: the solve kernel with time slicing
    (more problems than resident workgroups: re-queue rings from the handle's pool of workspaces,
    shrunk to 4 here -- GIK_SLICE_POOL, read at create -- so that 24 launches in flight reuse every
    slot several times behind its event) and the workgroup-per-goal prepare kernel (launches share one
    scratch slab and are chained by an event)."""
    torch = torch_cuda
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph("lwa4d")
    monkeypatch.setenv("GIK_SLICE_POOL", "4")
    prob = BatchProblem(graph, use_limits=True, force_block_prepare=True,
                        params={"force_block_path": path, "slice_outer_its": 24})
    monkeypatch.delenv("GIK_SLICE_POOL")
    tpl = prob.template
    assert tpl.info["is_block"] == 1 and (tpl.info["node_per_lane"] != 0) == (path == 2)
    B = tpl.info["n_cu"] * tpl.info["waves_per_cu"] + 192        # more than fit at once -> slicing on
    Tgs = [torch.from_numpy(_goals(robot, B, 200 + i)).cuda() for i in range(12)]

    def run(i):
        r = tpl.ik(Tgs[i])
        return {k: r[k] for k in _KEYS + ("q", "pos_err")}

    _serial_and_concurrent(torch, run, 12, 8)


def test_more_calls_in_flight_than_counter_slots(torch_cuda, monkeypatch):
    """The handle keeps a ring of work-queue counters, one per in-flight solve call; a slot is reused
    only behind the event recorded after its previous launch.  With the ring shrunk to 3 slots
    (GIK_COUNTER_RING, read at create) 48 calls on 8 streams wrap it 16 times while earlier kernels
    are still running; and 600 small calls wrap the full 256-slot ring of a default handle."""
    torch = torch_cuda
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph("lwa4d")
    monkeypatch.setenv("GIK_COUNTER_RING", "3")
    small_ring = BatchProblem(graph, use_limits=True)
    monkeypatch.delenv("GIK_COUNTER_RING")
    Tgs = [torch.from_numpy(_goals(robot, 2048, 300 + i)).cuda() for i in range(24)]

    def run(i):
        r = small_ring.template.ik(Tgs[i])
        return {k: r[k] for k in _KEYS}

    _serial_and_concurrent(torch, run, 24, 8)

    full_ring = BatchProblem(graph, use_limits=True, params={"maxiter": 40})
    Tg = torch.from_numpy(_goals(robot, 300 * 64, 7)).cuda().reshape(300, 64, 4, 4)

    def run_small(i):
        r = full_ring.template.ik(Tg[i])
        return {k: r[k] for k in ("x", "iterations", "inner_total")}

    _serial_and_concurrent(torch, run_small, 300, 8)      # 600 launches without a synchronisation


def test_clique_closed_form_is_an_explicit_parameter(torch_cuda):
    """gik_template_desc.clique_closed_form replaces the debug bits 128 / 256: AUTO (closed form,
    moments for Euclidean targets), DENSE (closed form, dense D w), OFF (per-term loops, the
    reference's summation order).  All three agree to round-off on the golden table-scene points and
    give the same iteration-count distribution; gik_template_get_info / gik_stats.flags say what ran."""
    from conftest import load_golden
    from graphik_amd.engine import Template
    d = load_golden("ur10_table")
    res = {}
    for mode in ("auto", "dense", "off"):
        T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True,
                                   params={"clique_closed_form": mode})
        assert T.info["clique_closed_form"] == {"auto": 0, "dense": 2, "off": 1}[mode]
        assert T.info["n_clique"] == (0 if mode == "off" else 106)
        tg = T.targets_from_D(d["D_goal"])
        h = T.hess(d["kat_Y"][:4], d["kat_W"][:4], tg[:1]).cpu().numpy()
        r = T.solve(d["Y_init"], tg)
        res[mode] = (h, r["iterations"].cpu().numpy(), r["flags"].cpu().numpy(), r["f"].cpu().numpy())
    for mode in ("dense", "off"):
        assert np.abs(res[mode][0] - res["auto"][0]).max() <= 1e-12 * np.abs(res["auto"][0]).max()
        assert np.array_equal(res[mode][3] < 1e-9, res["auto"][3] < 1e-9)
        conv = res["auto"][3] < 1e-9
        assert 0.8 < np.median(res[mode][1][conv]) / np.median(res["auto"][1][conv]) < 1.25
    assert np.all(res["auto"][2] & 1) and not np.any(res["dense"][2] & 1) and not np.any(res["off"][2] & 1)


# ---- RCCL: the gather of bench.py under torch.distributed.run -----------------------------------------
def _bench(cmd_prefix, *extra):
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run(cmd_prefix + [os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                                     "--no-cpu-baseline", "--serving-streams", "0", *extra],
                       env=env, timeout=900, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def _free_port():
    """A rendezvous port nobody holds (a fixed one may still be in TIME_WAIT from the previous run)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_rccl_gather_under_torchrun(torch_cuda):
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 --backend nccl`: RCCL comes up, the
    barrier / max-over-ranks / all_gather path runs on the device, and the gathered per-problem
    table equals (sha256) the table of a plain single-process run without a process group -- the
    only RCCL coverage a one-GPU lease allows."""
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    a = _bench(launcher, "--backend", "nccl", "--batch", "256")
    b = _bench([sys.executable], "--batch", "256")
    assert a["gather"]["backend"] == "nccl" and b["gather"]["backend"] is None
    assert a["gather"]["rows"] == b["gather"]["rows"] == 256
    assert a["gather"]["sha256"] == b["gather"]["sha256"]
    assert a["n_gpus"] == 1 and a["value"] > 0 and a["success_rate"] > 0.9
    # strong-scaling workload through the same path (c5: cheap)
    c = _bench(launcher, "--backend", "nccl", "--config", "c5")
    assert c["gather"]["rows"] == 65536 and c["scaling"] == "strong" and c["success_rate"] > 0.99
    # the driver's own invocation (no --config) under the launcher: the headline plus every other
    # BASELINE workload under "configs", all through the same process group
    d = _bench(launcher, "--backend", "nccl")
    assert d["config"]["config"] == "c2" and d["gather"]["backend"] == "nccl" and d["gather"]["rows"] == 4096
    assert set(d["configs"]) == {"c3", "c4", "c4_share_of_8", "c5", "c5_nolimits", "c2_column", "c4_column"}
    # the default lines run the per-edge product form, the *_column lines the column form -- and say so
    assert "per-edge" in d["roofline"]["kernel"] and "per-edge" in d["configs"]["c4"]["kernel"]
    assert "column" in d["configs"]["c2_column"]["kernel"] and "column" in d["configs"]["c4_column"]["kernel"]
    assert d["summary"]["traffic_stale"] in (True, False) and "traffic_stale" in d["roofline"]
    assert len(d["per_rank"]) == 1 and d["per_rank"][0]["goals"] == 4096 and d["per_rank"][0]["max_outer"] == 3000
    assert d["seeds"]["seeds"] == [0, 1, 2, 3] and len(d["seeds"]["ms_per_step"]) == 4
    assert d["gather"]["bytes_per_problem"] == 8 * (7 + 9) and d["kernels"]["dominant"].startswith("rtr_wave_kernel")
    assert d["configs"]["c5"]["kernels"]["dominant"].startswith("prep_quad_kernel") and d["configs"]["c5"]["roofline_prepare"]["frac"] > 0
    assert d["configs"]["c3"]["kernel"].startswith("rtr_npt_kernel")
    for name, lo in (("c3", 1300), ("c4", 80e3), ("c4_share_of_8", 35e3), ("c5", 5e6), ("c5_nolimits", 5e6)):
        cf = d["configs"][name]
        assert cf["value"] > lo and 0 < cf["roofline"]["frac_executed"] <= cf["roofline"]["frac"] < 1, (name, cf)
    assert d["configs"]["c3"]["success_rate"] > 0.88 and d["configs"]["c5"]["success_rate"] > 0.999


SHARDED_RCCL = r'''
import os, sys
sys.path.insert(0, os.environ["GIK_REPO"]); sys.path.insert(0, os.path.join(os.environ["GIK_REPO"], "tests"))
import numpy as np, torch
from graphik_amd import distributed as gd
from conftest import make_graph
rank, local_rank, world = gd.init_process_group(backend="nccl")
robot, graph = make_graph("lwa4d")
T = robot.fk_batch(-np.pi + 2 * np.pi * np.random.RandomState(9).rand(512, robot.n))
q, Y, info = gd.solve_batch_sharded(graph, T, with_Y=True)
import torch.distributed as dist
np.savez(os.environ["GIK_OUT"], q=q, Y=Y, backend=np.array(dist.get_backend()), **info)
gd.shutdown()
'''


def test_solve_batch_sharded_over_rccl(torch_cuda, tmp_path):
    """The library call of SURVEY 8(e): graphik_amd.distributed.solve_batch_sharded under
    `torch.distributed.run --nproc-per-node 1` with the RCCL backend (all a one-GPU lease allows) returns,
    through its one all_gather of q + statistics + Y, exactly what solve_batch returns in a process
    without a process group -- joint angles, points and counters bit for bit."""
    from graphik_amd import distributed as gd
    from graphik_amd.solvers.riemannian_solver import solve_batch
    script = tmp_path / "sharded_rccl.py"
    script.write_text(SHARDED_RCCL)
    out = tmp_path / "sharded_rccl.npz"
    env = dict(os.environ, GIK_REPO=REPO, GIK_OUT=str(out), MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, timeout=600, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    got = dict(np.load(out))
    assert str(got["backend"]) == "nccl"
    robot, graph = make_graph("lwa4d")
    T = robot.fk_batch(-np.pi + 2 * np.pi * np.random.RandomState(9).rand(512, robot.n))
    q, Y, info = solve_batch(graph, T)
    assert np.array_equal(got["q"], q) and np.array_equal(got["Y"], Y)
    assert np.array_equal(got["iterations"], info["iterations"]) and np.array_equal(got["f"], info["f(x)"])
    assert np.array_equal(got["pos_err"], info["pos_err"])
    # ... and without a process group it is solve_batch itself
    q1, Y1, info1 = gd.solve_batch_sharded(graph, T, with_Y=True)
    assert np.array_equal(q1, q) and np.array_equal(Y1, Y) and np.array_equal(info1["stop"], info["stop"])
    assert gd.result_row_bytes(robot.n, 18, 3, with_Y=True) == 560


def test_integration_stub_runs(torch_cuda):
    """The ctypes stub INTEGRATION.md tells a GraphIK maintainer to add is executed as written (only
    the library path is filled in) and must give, bit for bit, what the package's own engine gives."""
    import re
    from conftest import load_golden
    from graphik_amd import _ffi
    from graphik_amd.engine import Template
    md = open(os.path.join(REPO, "INTEGRATION.md")).read()
    block = next(b for b in re.findall(r"```python\n(.*?)```", md, flags=re.S) if "hip_backend.py" in b)
    block = block.replace('C.CDLL("libgraphik_amd.so")', f'C.CDLL({_ffi.LIB_PATH!r})')
    ns = {}
    exec(compile(block, "INTEGRATION.md:hip_backend", "exec"), ns)
    d = load_golden("lwa4d")
    stub = ns["HipTemplate"](d["omega"], d["psi_L"], d["psi_U"], 3, use_limits=True)
    Y, stats = stub.solve(d["Y_init"], d["D_goal"])
    torch_cuda.cuda.synchronize()
    T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True)
    r = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"]))
    assert np.array_equal(Y.cpu().numpy(), r["x"].cpu().numpy())
    assert np.array_equal(stats[:, 0].cpu().numpy(), r["f"].cpu().numpy())
    assert np.array_equal(stats.view(torch_cuda.int32)[:, 4].cpu().numpy(), r["iterations"].cpu().numpy())
