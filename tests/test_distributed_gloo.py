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


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "full.npy"
    env = dict(os.environ, GIK_REPO=REPO, GIK_OUT=str(out), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)]
    subprocess.run(cmd, check=True, env=env, timeout=300, capture_output=True)
    full = np.load(out)
    assert np.array_equal(full, np.arange(37.0)[:, None] * np.array([[1.0, 2.0, 3.0]]))
