"""N>1 path on CPU: two gloo ranks shard a batch, run a deterministic per-row function on their
shard, and gather; the result must equal the single-process answer."""
import os
import subprocess
import sys

import numpy as np

from conftest import REPO
from graphik_amd.distributed import shard_range

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["GIK_REPO"])
import numpy as np, torch
from graphik_amd import distributed as gd
rank, local_rank, world = gd.init_process_group(backend="gloo")
total = 37
lo, hi = gd.shard_range(total, rank, world)
rows = torch.arange(total, dtype=torch.float64)[lo:hi, None] * torch.tensor([[1.0, 2.0, 3.0]], dtype=torch.float64)
gd.barrier()
full = gd.gather_rows(rows, total, dst=0)
mx = gd.max_over_ranks(float(rank + 1), torch.device("cpu"))
sm = gd.sum_over_ranks(float(hi - lo), torch.device("cpu"))
if rank == 0:
    np.save(os.environ["GIK_OUT"], full.numpy())
    assert mx == world and sm == total
else:
    assert full is None
'''


def test_shard_ranges_cover_batch():
    for total in (0, 1, 7, 4096, 65536 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _free_port():
    """A rendezvous port nobody holds (a fixed one may still be in TIME_WAIT from the previous run)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "full.npy"
    env = dict(os.environ, GIK_REPO=REPO, GIK_OUT=str(out), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    subprocess.run(cmd, check=True, env=env, timeout=300, capture_output=True)
    full = np.load(out)
    assert np.array_equal(full, np.arange(37.0)[:, None] * np.array([[1.0, 2.0, 3.0]]))


def _bench_line(*extra):
    """Run bench.py the way the driver does (`python bench.py --gpus N ...`, no launcher) and
    return its one JSON line."""
    import json
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--dry-solve", "--backend", "gloo",
                        "--steps", "2", "--warmup", "1", *extra], env=env, timeout=600,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout        # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_self_launches_ranks_and_gathers():
    """`python bench.py --gpus 2` must bring up its two ranks itself (round 1 asserted on the world
    size instead) and the table gathered from the two contiguous shards must be, row for row, the
    single-process table -- through bench.py's own rank / shard / barrier / gather code, with a
    deterministic stand-in for the device solve (--dry-solve)."""
    one = _bench_line("--gpus", "1", "--config", "c4", "--batch", "0")
    two = _bench_line("--gpus", "2", "--config", "c4")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["dry_solve"] and two["dry_solve"] and two["value"] is None
    assert one["config"]["goals_total"] == two["config"]["goals_total"] == 65536    # strong scaling
    assert one["rows"] == two["rows"] == 65536
    assert one["rows_sha"] == two["rows_sha"]
    assert two["scaling"] == "strong"
    # weak-scaling default (BASELINE configs[1]): every rank brings its own 4096 goals
    w2 = _bench_line("--gpus", "2")
    assert w2["config"]["robot"] == "lwa4d" and w2["rows"] == 2 * 4096 and w2["scaling"] == "weak"
    # ... and the default line (what the driver runs at every N) carries the other BASELINE workloads
    # under "configs": the c4 / c5 tables gathered from two shards equal the one-rank tables
    cf = w2["configs"]
    assert set(cf) == {"c3", "c4", "c5", "c5_nolimits", "c2_column", "c4_column"}     # (the 8192-goal share line is single-GPU only)
    assert cf["c4_column"]["rows_sha"] == one["rows_sha"]      # (the dry solve knows no product form: the same table)
    # every rank's own time, longest problem, share of maxiter goals and goal count (a shard is as long as
    # ITS longest problem -- the record the first hardware scaling run is read against)
    for line, world in ((one, 1), (two, 2), (w2, 2)):
        pr = line["per_rank"]
        assert len(pr) == world and all(set(r) == {"ms", "max_outer", "frac_maxiter", "goals"} for r in pr), pr
        assert sum(r["goals"] for r in pr) == line["rows"]
    assert [r["goals"] for r in two["per_rank"]] == [32768, 32768]
    assert cf["c4"]["rows_sha"] == one["rows_sha"] and cf["c4"]["scaling"] == "strong"
    assert cf["c3"]["rows"] == 2 * 4096 and cf["c3"]["scaling"] == "weak"
    # c5 shards the planar chain the same way (uneven world sizes included)
    p1, p3 = _bench_line("--gpus", "1", "--config", "c5"), _bench_line("--gpus", "3", "--config", "c5")
    assert p1["rows_sha"] == p3["rows_sha"] and p3["rows"] == 65536
    assert cf["c5"]["rows_sha"] == p1["rows_sha"]
    assert "configs" not in p1                       # explicit --config: that workload only


def test_bench_refuses_a_wrong_world_size():
    """Started under a launcher whose world size is not --gpus, bench.py must say so (not assert)."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--dry-solve", "--gpus", "2"],
                       env=env, timeout=300, capture_output=True, text=True)
    assert r.returncode != 0 and "launcher started 1 rank" in (r.stderr + r.stdout)


SHARDED_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["GIK_REPO"])
sys.path.insert(0, os.path.join(os.environ["GIK_REPO"], "tests"))
import numpy as np, torch
from graphik_amd import distributed as gd
from conftest import make_graph
sys.path.insert(0, os.environ["GIK_REPO"])
import bench
rank, local_rank, world = gd.init_process_group(backend="gloo")
robot, graph = make_graph("lwa4d")
rs = np.random.RandomState(5)
T = robot.fk_batch(-np.pi + 2 * np.pi * rs.rand(203, robot.n))          # every rank holds the full batch
N, k, n = graph.number_of_nodes(), graph.dim, robot.n

def solve_fn(T_local):            # deterministic stand-in for the device solve, rows = functions of the goal
    res = bench.dry_solve(T_local, n)
    key = np.abs(T_local.reshape(len(T_local), -1)).sum(axis=1)
    res["x"] = np.cos(key[:, None, None] * (1.0 + np.arange(N * k).reshape(1, N, k)))
    return res

q, Y, info = gd.solve_batch_sharded(graph, T, with_Y=True, solve_fn=solve_fn)
q2, Y2, info2 = gd.solve_batch_sharded(graph, T, with_Y=False, solve_fn=solve_fn)
# SURVEY 8(e): ONE gather to dst -- a rank other than dst sends its rows and receives NOTHING
if world > 1:
    assert gd.LAST_GATHER["collective"] == "gather", gd.LAST_GATHER
    if rank != 0:
        assert gd.LAST_GATHER["recv_bytes"] == 0 and gd.LAST_GATHER["sent_bytes"] > 0, gd.LAST_GATHER
    else:
        assert gd.LAST_GATHER["recv_bytes"] > 0
# fewer goals than ranks: the ranks with an empty shard skip the solve and still enter the gather
calls = []
def counting(T_local):
    calls.append(len(T_local))
    return solve_fn(T_local)
class RowsOnly:                       # a batch that refuses to be materialised as a whole
    def __init__(self, a): self.a = a
    def __len__(self): return len(self.a)
    def __getitem__(self, s):
        assert isinstance(s, slice) and (s.stop - s.start) <= -(-len(self.a) // world), "rank read other ranks' goals"
        return self.a[s]
q3, Y3, info3 = gd.solve_batch_sharded(graph, RowsOnly(T[:3]), with_Y=True, solve_fn=counting)
lo3, hi3 = gd.shard_range(3, rank, world)
assert calls == ([hi3 - lo3] if hi3 > lo3 else []), (rank, calls)
q0, Y0, info0 = gd.solve_batch_sharded(graph, T[:0], with_Y=True, solve_fn=counting)      # an empty batch
if rank == 0:
    assert Y2 is None and np.array_equal(q, q2)
    assert q0.shape == (0, n) and Y0.shape == (0, N, k)
    np.savez(os.environ["GIK_OUT"], q=q, Y=Y, q3=q3, Y3=Y3, it3=info3["iterations"], **info)
else:
    assert q is None and Y is None and info is None and q2 is None and q3 is None
'''


def _run_sharded(tmp_path, world, port):
    script = tmp_path / f"sharded_{world}.py"
    script.write_text(SHARDED_WORKER)
    out = tmp_path / f"sharded_{world}.npz"
    env = dict(os.environ, GIK_REPO=REPO, GIK_OUT=str(out), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    if world == 1:
        subprocess.run([sys.executable, str(script)], check=True, env=env, timeout=300, capture_output=True)
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
        subprocess.run(cmd, check=True, env=env, timeout=600, capture_output=True)
    return dict(np.load(out))


def test_solve_batch_sharded_gathers_q_and_Y(tmp_path):
    """graphik_amd.distributed.solve_batch_sharded -- the library's multi-GPU solve_batch -- over 2 and 8
    gloo ranks: the ONE gather returns, row for row, what a single process returns: q [B,n], the
    statistics and (with_Y) the points Y [B,N,k]; ranks other than dst get None."""
    from graphik_amd.distributed import RESULT_STATS, result_row_bytes
    one = _run_sharded(tmp_path, 1, 0)
    assert one["q"].shape == (203, 7) and one["Y"].shape == (203, 18, 3)
    assert set(RESULT_STATS) <= set(one) and one["iterations"].dtype == np.int64
    # (B = 3 < world = 8: ADVICE r4 -- empty shards used to raise before the gather and hang the others)
    assert one["q3"].shape == (3, 7) and np.array_equal(one["q3"], one["q"][:3]) and np.array_equal(one["Y3"], one["Y"][:3])
    assert result_row_bytes(7, 18, 3, with_Y=True) == 8 * (7 + len(RESULT_STATS) + 54)     # 560 B per problem
    for world in (2, 8):
        many = _run_sharded(tmp_path, world, _free_port())
        assert set(many) == set(one)
        for key in one:
            assert np.array_equal(one[key], many[key]), (world, key)
