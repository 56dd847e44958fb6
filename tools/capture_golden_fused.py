#!/usr/bin/env python3
"""Supplementary golden vectors for the fused cost+gradient twins (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden_fused.py

Runs the reference's `jcost_and_grad` (graphik/solvers/costs.py:61-77) and `lcost_and_grad`
(:126-169) -- plain Python loops under the numba stand-in of tools/ref_shims -- on the known-answer
inputs already stored in tests/golden/<scenario>.npz (kat_Y, goal 0's D_goal, omega, psi_L, psi_U,
edge lists) and writes tests/golden/fused_kat.npz: per scenario `<name>_{lim,nolim}_{cost,grad}`.
Only numbers are written.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "ref_shims"))

import refcompat  # noqa: E402,F401  (must precede graphik imports)
import numpy as np  # noqa: E402
import graphik.solvers.costs as costs  # noqa: E402

G = os.path.join(REPO, "tests", "golden")
out = {}
for fn in sorted(os.listdir(G)):
    if not fn.endswith(".npz") or fn == "fused_kat.npz":
        continue
    d = np.load(os.path.join(G, fn))
    if "kat_Y" not in d.files:
        continue
    name = fn[:-4]
    D, om, pL, pU = d["D_goal"][0], d["omega"], d["psi_L"], d["psi_U"]
    il = tuple(np.asarray(a, dtype=np.uint64) for a in d["inds_limits"])
    inl = tuple(np.asarray(a, dtype=np.uint64) for a in d["inds_nolimits"])
    fl, gl, fn_, gn = [], [], [], []
    for Y in d["kat_Y"]:
        f, g = costs.lcost_and_grad(Y, D, om, pL, pU, il)
        fl.append(float(f)); gl.append(np.asarray(g, float))
        f, g = costs.jcost_and_grad(Y, D, inl)
        fn_.append(float(f)); gn.append(np.asarray(g, float))
    out[f"{name}_lim_cost"] = np.array(fl); out[f"{name}_lim_grad"] = np.array(gl)
    out[f"{name}_nolim_cost"] = np.array(fn_); out[f"{name}_nolim_grad"] = np.array(gn)
    # the reference's own claim: the fused loops return what lcost/lgrad, jcost/jgrad return
    print(name, "max |fused - separate|: lim cost %.1e grad %.1e | nolim cost %.1e grad %.1e" % (
        np.max(np.abs(out[f"{name}_lim_cost"] - d["kat_lim_loop_cost"])),
        np.max(np.abs(out[f"{name}_lim_grad"] - d["kat_lim_loop_grad"])),
        np.max(np.abs(out[f"{name}_nolim_cost"] - d["kat_nolim_loop_cost"])),
        np.max(np.abs(out[f"{name}_nolim_grad"] - d["kat_nolim_loop_grad"]))))
np.savez_compressed(os.path.join(G, "fused_kat.npz"), **out)
print("wrote", os.path.join(G, "fused_kat.npz"), len(out), "arrays")
