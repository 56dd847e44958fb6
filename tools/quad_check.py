"""Developer check of the four-problems-per-wavefront kernel (planar graphs) against the one-problem-per-
wavefront kernel (debug_flags = 8192): traces decision for decision on the golden goals, statistics of a
larger jittered batch, and solve-kernel timings."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphik_amd.engine import Template


def load_golden(name):
    return np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name + ".npz"))


def templates(d):
    kw = dict(k=2, use_limits=bool(int(d["use_limits"])))
    Tq = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], params={"debug_flags": 16384}, **kw)
    Tw = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], params={"debug_flags": 8192}, **kw)
    assert Tq.info["problems_per_wave"] == 4 and Tw.info["problems_per_wave"] == 1, (Tq.info, Tw.info)
    return Tq, Tw


def traces():
    for name in ("planar10_nolimits", "planar10_limits_pi", "planar10_limits_halfpi"):
        d = load_golden(name)
        Tq, Tw = templates(d)
        tg = Tq.targets_from_D(d["D_goal"])
        rq = Tq.solve(d["Y_init"], tg, trace_cap=32)
        rw = Tw.solve(d["Y_init"], tg, trace_cap=32)
        torch.cuda.synchronize()
        print(name, "goals", len(d["Y_init"]))
        for key in ("numit", "stop", "accept"):
            a, b = rq["trace"][key].cpu().numpy(), rw["trace"][key].cpu().numpy()
            n = np.minimum(rq["iterations"].cpu().numpy(), rw["iterations"].cpu().numpy())
            same = [np.array_equal(a[g, :min(n[g], 8)], b[g, :min(n[g], 8)]) for g in range(len(n))]
            print("  ", key, "first 8 equal on", int(np.sum(same)), "of", len(same))
        for key in ("iterations", "inner_total", "stop", "n_accept"):
            print("  ", key, "quad", rq[key].cpu().numpy()[:12], "wave", rw[key].cpu().numpy()[:12])
        print("   f quad", rq["f"].cpu().numpy()[:6], "\n   f wave", rw["f"].cpu().numpy()[:6])
        dY = np.abs(rq["x"].cpu().numpy() - rw["x"].cpu().numpy()).max()
        print("   max |Y_quad - Y_wave|", dY, flush=True)


def timing(B):
    for name in ("planar10_limits_pi", "planar10_nolimits"):
        d = load_golden(name)
        Tq, Tw = templates(d)
        tg = Tq.targets_from_D(d["D_goal"])
        G = len(d["Y_init"])
        rng = np.random.RandomState(0)
        idx = np.arange(B) % G
        Y0 = d["Y_init"][idx] + 1e-3 * rng.randn(B, *d["Y_init"].shape[1:])
        tgb = tg[torch.as_tensor(idx, device=tg.device)] if torch.is_tensor(tg) else np.asarray(tg)[idx]
        out = {}
        for tag, T in (("quad", Tq), ("wave", Tw)):
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.time()
                r = T.solve(Y0, tgb)
                torch.cuda.synchronize()
                dt = time.time() - t0
            out[tag] = r
            it = r["iterations"].cpu().numpy()
            print(f"{name} {tag}: B={B} {dt*1e3:.2f} ms, products {r['inner_total'].cpu().numpy().sum()/1e6:.2f} M, "
                  f"iterations mean {it.mean():.2f} max {it.max()}, converged {np.mean(r['f'].cpu().numpy() < 1e-9):.4f}", flush=True)
        for key in ("iterations", "inner_total", "stop", "n_accept"):
            a, b = out["quad"][key].cpu().numpy(), out["wave"][key].cpu().numpy()
            print(f"   {key}: equal on {np.mean(a == b):.4f} of the problems; sums {a.sum()} / {b.sum()}")
        fq, fw = out["quad"]["f"].cpu().numpy(), out["wave"]["f"].cpu().numpy()
        dY = np.abs(out["quad"]["x"].cpu().numpy() - out["wave"]["x"].cpu().numpy()).max()
        print(f"   f: max quad {fq.max():.3e} wave {fw.max():.3e}; max |Y diff| {dY:.3e}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["traces", "timing"]
    if "traces" in what: traces()
    if "timing" in what: timing(int(os.environ.get("QUAD_B", "65536")))
