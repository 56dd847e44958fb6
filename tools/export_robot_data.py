#!/usr/bin/env python3
"""Write graphik_amd/data/robots/<name>.json from the captured golden templates.

The kinematic constants (frames at zero configuration, re-based on the first joint, z along the
joint axis -- what RobotURDF.make_Revolute3d hands to RobotRevolute, roboturdf.py:226-264) were
extracted from the reference's URDF data files by tools/capture_golden.py; this script only
re-packages them so the loaders work where /root/reference does not exist.
"""
import json
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = {"lwa4d": ("lwa4d.npz", "lwa4d.urdf"), "ur10": ("ur10.npz", "ur10_mod.urdf"),
       "kuka": ("kuka.npz", "kuka_iiwr.urdf")}
out_dir = os.path.join(REPO, "graphik_amd", "data", "robots")
os.makedirs(out_dir, exist_ok=True)
for name, (npz, urdf) in SRC.items():
    d = np.load(os.path.join(REPO, "tests", "golden", npz))
    rec = {"name": name, "source_urdf": urdf, "num_joints": int(d["n_joints"]),
           "T_zero": [[[float.hex(float(v)) for v in row] for row in T] for T in d["T0"]]}
    with open(os.path.join(out_dir, name + ".json"), "w") as f:
        json.dump(rec, f, indent=0)
    print("wrote", name)
