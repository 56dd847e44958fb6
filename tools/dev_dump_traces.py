"""dev: HIP trust-region traces + finals from the golden start points (wave and forced block path),
written to gpurun_out/traces_<name>.npz for offline comparison with the oracle and the fixtures."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_golden, make_graph
from graphik_amd.engine import Template
from graphik_amd.graphs.graph_revolute import joint_variables_revolute_batch

os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
for name in ("lwa4d", "ur10", "kuka"):
    d = load_golden(name)
    robot, graph = make_graph(name)
    out = {}
    for tag, params in (("wave", None), ("block", {"force_block_path": 1})):
        T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True, params=params)
        r = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"]), trace_cap=48)
        for k, v in r["trace"].items():
            out[f"{tag}_t_{k}"] = v.cpu().numpy()
        for k in ("x", "f", "gradnorm", "iterations", "inner_total", "stop"):
            out[f"{tag}_{k}"] = r[k].cpu().numpy()
        out[f"{tag}_q"] = joint_variables_revolute_batch(graph, out[f"{tag}_x"], d["T_goal"])
    np.savez(os.path.join(REPO, "gpurun_out", f"traces_{name}.npz"), **out)
    print(name, out["wave_iterations"].tolist())
