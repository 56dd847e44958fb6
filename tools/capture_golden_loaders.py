#!/usr/bin/env python3
"""Templates of the remaining URDF loaders (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden_loaders.py

Runs the reference's load_schunk_lwa4p / load_panda (roboturdf.py:299-312, 343-356) and records, per
arm, the problem-graph template (node order, edge attribute matrices, psi_L / psi_U,
zero-configuration frames, limits -- the same arrays tools/capture_golden.py records for the
BASELINE arms) plus a few (configuration, pose, realization, joint_variables) tuples ->
tests/golden/loaders_extra.npz.  `--export` also writes graphik_amd/data/robots/<arm>.json (the
frames, as hex floats) so that the loaders work where /root/reference does not exist.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import capture_golden as cg  # noqa: E402  (sets up the shims and patches, imports the reference)
import numpy as np  # noqa: E402
from graphik.utils.roboturdf import load_panda, load_schunk_lwa4p  # noqa: E402
from graphik.utils.dgp import pos_from_graph  # noqa: E402

ARMS = {"lwa4p": (load_schunk_lwa4p, "lwa4p.urdf"), "panda": (load_panda, "panda_arm.urdf")}

if __name__ == "__main__":
    out = {}
    for name, (loader, urdf) in ARMS.items():
        robot, graph = loader()
        t = cg.template_arrays(graph, robot)
        n = robot.n
        Q, TG, X, QR = [], [], [], []
        for seed in range(6):
            np.random.seed(seed)
            q = robot.random_configuration()
            T = robot.pose(q, f"p{n}")
            G = graph.realization(q)
            qr = graph.joint_variables(G, {f"p{n}": T})
            Q.append([q[f"p{i}"] for i in range(1, n + 1)])
            TG.append(T.as_matrix())
            X.append(pos_from_graph(G, list(graph.node_ids)))
            QR.append([qr[f"p{i}"] for i in range(1, n + 1)])
        t.update(q_goal=np.array(Q), T_goal=np.array(TG), X=np.array(X), q_rec=np.array(QR))
        out.update({f"{name}_{k}": v for k, v in t.items()})
        print(name, "n =", n, "N =", len(graph.node_ids), "edges", graph.number_of_edges())
        if "--export" in sys.argv:
            rec = {"name": name, "source_urdf": urdf, "num_joints": int(n),
                   "T_zero": [[[float.hex(float(v)) for v in row] for row in T] for T in t["T0"]]}
            with open(os.path.join(cg.REPO, "graphik_amd", "data", "robots", name + ".json"), "w") as f:
                json.dump(rec, f, indent=0)
    path = os.path.join(cg.OUT, "loaders_extra.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
