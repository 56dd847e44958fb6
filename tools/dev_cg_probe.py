"""dev: smallest CG launches, each under its own timeout by the caller."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_golden
from graphik_amd.engine import Template
import torch
name, path, maxiter, B = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
d = load_golden(name)
T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=int(d["dim"]), use_limits=bool(int(d["use_limits"])),
                           params={"solver": "ConjugateGradient", "force_block_path": int(path == "block"), "maxiter": maxiter})
print("template ok", T.params, flush=True)
r = T.solve(d["Y_init"][:B], T.targets_from_D(d["D_goal"][:B]), trace_cap=16)
torch.cuda.synchronize()
print(name, path, "its", r["iterations"].cpu().numpy(), "stop", r["stop"].cpu().numpy(), "f", r["f"].cpu().numpy(), flush=True)
print("trace f", r["trace"]["f_before"][0].cpu().numpy()[:6], "evals", r["trace"]["numit"][0].cpu().numpy()[:6], flush=True)
