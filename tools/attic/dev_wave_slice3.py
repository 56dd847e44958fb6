"""dev: round-robin slicing, where the waves' time goes (dev build, debug flag 4096)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["GIK_LIB_PATH"] = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "graphik_amd/lib/exp/libgraphik_amd_dev.so")
import numpy as np, torch
from graphik_amd import _ffi
from graphik_amd.solvers.riemannian_solver import BatchProblem
from graphik_amd.utils.roboturdf import load_schunk_lwa4d, load_ur10, load_kuka
name, B = os.environ.get("ROBOT", "kuka"), int(os.environ.get("B", "8192"))
robot, graph = {"kuka": load_kuka, "lwa4d": load_schunk_lwa4d, "ur10": load_ur10}[name]()
rs = np.random.RandomState(0)
lb, ub = robot.limits_arrays()
Tg = torch.from_numpy(robot.fk_batch(lb + (ub - lb) * rs.rand(B, robot.n))).cuda()
L = C.CDLL(_ffi.LIB_PATH)
print(name, B)
for cfg in os.environ.get("CFGS", "0:0 32:16000000 32:8000000").split():
    sl, cyc = cfg.split(":")
    os.environ["GIK_SLICE"] = sl; os.environ["GIK_SLICE_CYCLES"] = cyc
    prob = BatchProblem(graph, use_limits=True, params={"slice_outer_its": int(sl), "debug_flags": 4096})
    tg, Y0 = prob.template.prepare(Tg)
    r = prob.template.solve(Y0, tg); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = prob.template.solve(Y0, tg); e1.record(); torch.cuda.synchronize()
    buf = np.zeros(1 << 20); L.gik_debug_fetch(buf.ctypes.data_as(C.c_void_p), 1 << 20)
    n = int(min(buf[8], 250000)); ev = buf[16:16 + 4 * n].reshape(n, 4).copy(); ev[:, 0] -= ev[:, 0].min()
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez_compressed(f"gpurun_out/events_{name}_{B}_{sl}_{cyc}.npz", ev=ev.astype(np.float32), its=r["iterations"].cpu().numpy())
    fl = r["flags"].cpu().numpy()
    print(f"slice {sl:>4s} cycles {cyc:>9s}: {e0.elapsed_time(e1):7.1f} ms; hand-overs {int((fl >> 8).sum())}; wave cycles: "
          f"solve {buf[4]:.3e} claim/wait {buf[5]:.3e} alive {buf[6]:.3e} claims {int(buf[7])}; outer its {int(r['iterations'].sum())} products {int(r['inner_executed'].sum())}", flush=True)
