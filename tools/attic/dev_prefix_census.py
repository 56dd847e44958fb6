"""dev: per golden goal, how long the HIP trace follows the oracle (1e-8) vs how long the reference's
numpy path follows the oracle at 1e-12 / 1e-8 (both kernel paths)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_golden
from parity_util import first_divergence, golden_traj, stable_prefix
from oracle import c_oracle as co
from graphik_amd.engine import Template
for name in ("lwa4d", "ur10", "kuka"):
    d = load_golden(name)
    for path in ("wave", "block"):
        T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True,
                                   params={"force_block_path": int(path == "block")})
        r = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"]), trace_cap=48)
        tr = {k: v.cpu().numpy() for k, v in r["trace"].items()}
        its = r["iterations"].cpu().numpy()
        rows = []
        for g in range(len(d["seed"])):
            o = co.rtr_solve(d["Y_init"][g], d["D_goal"][g], d["omega"], d["psi_L"], d["psi_U"], True, traj_cap=48)
            ref = golden_traj(d, "np", g)
            n = min(48, int(its[g]), o["iterations"], int(d["iterations"][g]))
            t = {k: tr[k][g] for k in tr}
            rows.append((stable_prefix(o["traj"], ref, n), first_divergence(o["traj"], ref, n), first_divergence(t, o["traj"], n),
                         first_divergence(t, o["traj"], n, rtol=1e-6)))
        print(name, path, "(K12 ref, K8 ref, K8 hip, K6 hip):", rows, flush=True)
