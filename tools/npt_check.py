"""Developer check of the node-per-lane kernel (force_block_path = 2) against the golden known answers,
the workgroup kernel and itself (time slicing); prints timings of a table-scene batch on both kernels."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from graphik_amd.engine import Template


def load_golden(name):
    return np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name + ".npz"))


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def kats():
    for name, flags in [("ur10_table", 0), ("ur10_table", 256), ("lwa4d", 0), ("lwa4d", 64), ("kuka", 64), ("ur10", 64)]:
        d = load_golden(name)
        use_lim = bool(int(d["use_limits"]))
        T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=use_lim,
                                   params={"force_block_path": 2, "debug_flags": flags})
        assert T.info["node_per_lane"] == 1, T.info
        key = "lim" if use_lim else "nolim"
        tg = T.targets_from_D(d["D_goal"][0])
        Y, W = d["kat_Y"], d["kat_W"]
        e = (rel_err(T.cost(Y, tg).cpu().numpy(), d[f"kat_{key}_loop_cost"]),
             rel_err(T.grad(Y, tg).cpu().numpy(), d[f"kat_{key}_loop_grad"]),
             rel_err(T.hess(Y, W, tg).cpu().numpy(), d[f"kat_{key}_loop_hess"]),
             rel_err(T.proj(Y, W).cpu().numpy(), d["kat_proj"]))
        print(f"KAT {name} flags {flags}: cost {e[0]:.1e} grad {e[1]:.1e} hess {e[2]:.1e} proj {e[3]:.1e}", T.info, flush=True)


def solves():
    for name, flags in [("lwa4d", 0), ("lwa4d", 64), ("ur10_table", 0)]:
        d = load_golden(name)
        kw = dict(k=3, use_limits=bool(int(d["use_limits"])))
        Tb = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], params={"force_block_path": 1, "debug_flags": flags}, **kw)
        Tn = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], params={"force_block_path": 2, "debug_flags": flags}, **kw)
        tg = Tb.targets_from_D(d["D_goal"])
        rb = Tb.solve(d["Y_init"], tg, trace_cap=32)
        rn = Tn.solve(d["Y_init"], tg, trace_cap=32)
        torch.cuda.synchronize()
        for key in ("numit", "stop", "accept"):
            a, b = rb["trace"][key].cpu().numpy()[:, :8], rn["trace"][key].cpu().numpy()[:, :8]
            print(name, flags, key, "first 8 equal:", np.array_equal(a, b))
        print(" f block", rb["f"].cpu().numpy(), "\n f npt  ", rn["f"].cpu().numpy())
        print(" its block", rb["iterations"].cpu().numpy(), "\n its npt  ", rn["iterations"].cpu().numpy())
        print(" inner block", rb["inner_total"].cpu().numpy(), "\n inner npt  ", rn["inner_total"].cpu().numpy())
        print(" flags", rn["flags"].cpu().numpy(), " stop", rn["stop"].cpu().numpy(), flush=True)


def timing(B=512):
    """B problems (the 8 golden table-scene goals, jittered): with B <= resident problems the kernel time is
    the longest problem's, i.e. (its executed products) x (latency of one product of a lone problem)."""
    d = load_golden("ur10_table")
    kw = dict(k=3, use_limits=True)
    G = len(d["Y_init"])
    rng = np.random.RandomState(0)
    for path, flags in ((1, 0), (2, 0), (2, 2048)):
        T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], params={"force_block_path": path, "debug_flags": flags}, **kw)
        tg = T.targets_from_D(d["D_goal"])
        idx = np.arange(B) % G
        Y0 = d["Y_init"][idx] + 1e-3 * rng.randn(B, *d["Y_init"].shape[1:])
        tgb = np.asarray(tg)[idx] if not torch.is_tensor(tg) else tg[torch.as_tensor(idx, device=tg.device)]
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.time()
            r = T.solve(Y0, tgb)
            torch.cuda.synchronize()
            dt = time.time() - t0
        it = r["iterations"].cpu().numpy()
        ex = r["inner_executed"].cpu().numpy().astype(np.int64)
        print(f"path {path} flags {flags}: B={B} {dt*1e3:.1f} ms, {B/dt:.0f} solves/s, products {ex.sum()/1e6:.1f} M, longest {ex.max()/1e3:.0f} k "
              f"-> {dt*2.4e9/ex.max():.0f} cycles/product if tail-bound; maxiter {np.mean(it >= 3000):.3f}; "
              f"converged {np.mean(r['f'].cpu().numpy() < 1e-9):.3f}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["kats", "solves", "timing"]
    if "kats" in what: kats()
    if "solves" in what: solves()
    if "timing" in what: timing(int(os.environ.get("NPT_B", "512")))
