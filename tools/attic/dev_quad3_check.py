"""Developer check of the four-3-D-problems-per-wavefront kernel (GIK_QUAD3_MIN_BATCH=1 in a child process)
against the one-problem-per-wavefront kernel: traces on the golden goals, statistics and time of a batch."""
import os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, R)
    import numpy as np, torch
    from graphik_amd.engine import Template
    name, B = sys.argv[3], int(sys.argv[4])
    d = np.load(os.path.join(R, "tests", "golden", name + ".npz"))
    T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=bool(int(d["use_limits"])))
    tg = np.asarray(T.targets_from_D(d["D_goal"]))
    r = T.solve(d["Y_init"], tg, trace_cap=32)
    out = {k: r[k].cpu().numpy() for k in ("x", "f", "gradnorm", "iterations", "inner_total", "stop", "n_accept")}
    for k in ("numit", "stop", "accept", "Delta", "f_before"):
        out["tr_" + k] = r["trace"][k].cpu().numpy()
    G = len(d["Y_init"])
    rng = np.random.RandomState(0)
    idx = np.arange(B) % G
    Y0 = d["Y_init"][idx] + 1e-3 * rng.randn(B, *d["Y_init"].shape[1:])
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        rb = T.solve(Y0, tg[idx])
        torch.cuda.synchronize(); dt = time.time() - t0
    out.update(ms=dt * 1e3, b_it=rb["iterations"].cpu().numpy(), b_inner=rb["inner_total"].cpu().numpy(),
               b_f=rb["f"].cpu().numpy(), b_stop=rb["stop"].cpu().numpy())
    np.savez(sys.argv[2], **out)
    sys.exit(0)
import numpy as np
for name, B in ((os.environ.get("Q3_ROBOT", "ur10"), int(os.environ.get("Q3_B", "65536"))),):
    outs = {}
    for tag, env in (("quad3", {"GIK_QUAD3_MIN_BATCH": "1"}), ("wave", {})):
        f = f"/tmp/quad3_{tag}.npz"
        subprocess.run([sys.executable, __file__, "child", f, name, str(B)], check=True, env=dict(os.environ, **env))
        outs[tag] = dict(np.load(f))
    a, b = outs["quad3"], outs["wave"]
    n = np.minimum(a["iterations"], b["iterations"])
    same = [all(np.array_equal(a["tr_" + k][g, :min(n[g], 5)], b["tr_" + k][g, :min(n[g], 5)]) for k in ("numit", "stop", "accept", "Delta"))
            for g in range(len(n))]
    print(f"{name}: first 5 outer iterations identical on {int(np.sum(same))} of {len(same)} golden goals")
    print("   iterations quad3", a["iterations"][:8], "wave", b["iterations"][:8])
    print("   f quad3", a["f"][:4], "wave", b["f"][:4], " converged classes equal:", np.array_equal(a["f"] < 1e-9, b["f"] < 1e-9))
    print(f"   B={B}: quad3 {a['ms']:.1f} ms, wave {b['ms']:.1f} ms; products {a['b_inner'].sum()/1e6:.1f} M / {b['b_inner'].sum()/1e6:.1f} M; "
          f"maxiter {np.mean(a['b_stop'] == 1):.3f} / {np.mean(b['b_stop'] == 1):.3f}; converged {np.mean(a['b_f'] < 1e-9):.3f} / {np.mean(b['b_f'] < 1e-9):.3f}", flush=True)
