"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the golden
vectors captured from the reference.  Run on the GPU box with `-m gpu`."""
import os

import numpy as np
import pytest

from conftest import SCENARIOS, SCENARIOS_2D, SCENARIOS_3D, load_golden, make_graph, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from graphik_amd import _ffi
    assert _ffi.lib().gik_device_count() >= 1
    return torch


def _template(d, **params):
    from graphik_amd.engine import Template
    use_lim = bool(int(d["use_limits"]))
    return Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=int(d["dim"]),
                                  use_limits=use_lim, params=params or None)


# ---- kernel-level known answers (costgrd twins + proj): 1e-12 relative ----------------------
@pytest.mark.parametrize("name", SCENARIOS)
def test_cost_grad_hess_proj_known_answers(torch_cuda, name):
    """(Planar scenarios twice: the one-problem-per-wavefront context and, with debug_flags 16384, the
    four-problems-per-wavefront context QuadCtx through kat_quad_kernel -- six points: two wavefronts, the second
    half empty.)"""
    from oracle import c_oracle as co
    d = load_golden(name)
    templates = [_template(d)] + ([_template(d, debug_flags=16384)] if int(d["dim"]) == 2 else
                                  [_template(d, hessian_form="column")])      # (3-D: + the column-form product, WaveCtx)
    if int(d["dim"]) == 3:
        # the default of a 3-D wavefront graph is the reference's per-edge arithmetic (WaveCtxStrict); graphs on the
        # workgroup / node-per-lane kernels (the table scene) form s per edge whatever the field says
        blk = templates[0].info["is_block"]
        assert templates[0].info["hessian_form"] == 1 and templates[1].info["hessian_form"] == (1 if blk else 0)
    for T in templates:
        key = "lim" if int(d["use_limits"]) else "nolim"
        tg = T.targets_from_D(d["D_goal"][0])
        Y, W = d["kat_Y"], d["kat_W"]
        assert rel_err(T.cost(Y, tg).cpu().numpy(), d[f"kat_{key}_loop_cost"]) < 1e-12
        assert rel_err(T.grad(Y, tg).cpu().numpy(), d[f"kat_{key}_loop_grad"]) < 1e-12
        assert rel_err(T.hess(Y, W, tg).cpu().numpy(), d[f"kat_{key}_loop_hess"]) < 1e-12
        assert rel_err(T.proj(Y, W).cpu().numpy(), d["kat_proj"]) < 1e-12
        # and against the oracle on fresh random points, including points near a solution where the
        # hinge terms switch on and off
        rng = np.random.RandomState(5)
        om, pL, pU, D = d["omega"], d["psi_L"], d["psi_U"], d["D_goal"][0]
        use_lim = bool(int(d["use_limits"]))
        inds = co.limit_inds(om, pL, pU) if use_lim else np.nonzero(np.triu(om))
        Ys = np.stack([d["Y_sol"][0] + s * rng.randn(*d["Y_sol"][0].shape)
                       for s in (0.0, 1e-6, 1e-3, 0.05, 0.3, 1.0)])
        Ws = rng.randn(*Ys.shape)
        g = T.grad(Ys, tg).cpu().numpy()
        h = T.hess(Ys, Ws, tg).cpu().numpy()
        c = T.cost(Ys, tg).cpu().numpy()
        # fused twin (lcost_and_grad / jcost_and_grad): one pass, bit-identical to the separate calls
        cf, gf = T.cost_and_grad(Ys, tg)
        assert np.array_equal(cf.cpu().numpy(), c) and np.array_equal(gf.cpu().numpy(), g)
        for m in range(len(Ys)):
            if use_lim:
                rc = co.lcost(Ys[m], D, om, pL, pU, inds)
                rg = co.lgrad(Ys[m], D, om, pL, pU, inds)
                rh = co.lhess(Ys[m], Ws[m], D, om, pL, pU, inds)
            else:
                rc, rg, rh = co.jcost(Ys[m], D, inds), co.jgrad(Ys[m], D, inds), co.jhess(Ys[m], Ws[m], D, inds)
            # near a solution the residuals D - d cancel to ~1e-8 of their operands, so the
            # achievable accuracy is eps*|D| per residual: absolute floors on top of 1e-12 relative
            assert abs(c[m] - rc) <= 1e-12 * abs(rc) + 1e-14 * np.sqrt(abs(rc))
            assert np.abs(g[m] - rg).max() <= 1e-12 * np.abs(rg).max() + 1e-13
            assert np.abs(h[m] - rh).max() <= 1e-12 * np.abs(rh).max() + 1e-13


@pytest.mark.parametrize("name", ["lwa4d", "planar10_limits_halfpi"])
def test_operator_properties(torch_cuda, name):
    """Size-independent properties: ehess is linear and symmetric, egrad is exactly half the
    finite-difference gradient of cost, proj is idempotent for k=3."""
    d = load_golden(name)
    T = _template(d)
    tg = T.targets_from_D(d["D_goal"][0])
    rng = np.random.RandomState(1)
    shape = d["kat_Y"][0].shape
    Y = rng.randn(*shape)
    W1, W2 = rng.randn(*shape), rng.randn(*shape)
    H = lambda W: T.hess(Y, W, tg)[0].cpu().numpy()
    assert rel_err(H(2.5 * W1 - 0.75 * W2), 2.5 * H(W1) - 0.75 * H(W2)) < 1e-12
    assert abs(np.sum(W1 * H(W2)) - np.sum(W2 * H(W1))) < 1e-10 * abs(np.sum(W1 * H(W2)))
    g = T.grad(Y, tg)[0].cpu().numpy()
    eps = 1e-6
    fd = (float(T.cost(Y + eps * W1, tg)[0]) - float(T.cost(Y - eps * W1, tg)[0])) / (2 * eps)
    assert abs(fd / np.sum(g * W1) - 2.0) < 1e-6
    if shape[1] == 3:
        P = T.proj(Y, W1)[0].cpu().numpy()
        assert rel_err(T.proj(Y, P)[0].cpu().numpy(), P) < 1e-12
        assert np.abs(Y.T @ P - P.T @ Y).max() < 1e-10  # horizontal


# ---- trust-region trajectories -----------------------------------------------------------------
@pytest.mark.parametrize("per_wave", [4, 1])
@pytest.mark.parametrize("name", SCENARIOS_2D)
def test_trajectory_planar_identical_to_oracle(torch_cuda, name, per_wave):
    """(Both planar solve kernels: four problems per wavefront -- the default -- and one, debug_flags 8192.)
    Planar solves are well conditioned: every discrete decision (inner-iteration count, tCG
    stop reason, accept flag, radius) equals the oracle's, and f agrees to 1e-7, for as long as
    the iterate is above the round-off floor (f >= 1e-14).  Below it the outcome of a tCG call
    hinges on whether CG's finite-termination collapse survives round-off (the reference's
    superlinear target asks for |r| <= |r0|^2 ~ 1e-17), so only the answer is compared there."""
    from oracle import c_oracle as co
    from graphik_amd.graphs.graph_planar import joint_variables_planar_batch
    d = load_golden(name)
    robot, graph = make_graph(name)
    T = _template(d, debug_flags=16384 if per_wave == 4 else 8192)     # (16384: the four-problem kernel at any batch size)
    assert T.info["problems_per_wave"] == per_wave
    use_lim = bool(int(d["use_limits"]))
    r = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"]), trace_cap=48)
    tr = {k: v.cpu().numpy() for k, v in r["trace"].items()}
    its = r["iterations"].cpu().numpy()
    x = r["x"].cpu().numpy()
    exact_tail = 0
    for g in range(len(d["seed"])):
        o = co.rtr_solve(d["Y_init"][g], d["D_goal"][g], d["omega"], d["psi_L"], d["psi_U"],
                         use_lim, traj_cap=48)
        fo = o["traj"]["f_before"]
        m = int(np.argmax(fo < 1e-14)) if np.any(fo < 1e-14) else len(fo)
        m = min(m, int(its[g]))
        assert m >= 6
        assert np.array_equal(tr["numit"][g][:m], o["traj"]["numit"][:m])
        assert np.array_equal(tr["stop"][g][:m], o["traj"]["stop"][:m])
        assert np.array_equal(tr["accept"][g][:m], o["traj"]["accept"][:m])
        assert np.array_equal(tr["Delta"][g][:m], o["traj"]["Delta"][:m])
        assert np.allclose(tr["f_before"][g][:m], o["traj"]["f_before"][:m], rtol=1e-7)
        assert np.allclose(tr["gradnorm_after"][g][:m - 1], o["traj"]["gradnorm_after"][:m - 1],
                           rtol=1e-6)
        # A 10-link chain is redundant and Y is only defined up to a rigid motion, so below the
        # floor the iterate may drift along the solution set: compare what is determined -- the
        # cost and the end-effector pose of the recovered configuration.
        qg = joint_variables_planar_batch(graph, x[g][None])
        Tg = robot.fk_batch(qg)[0]
        assert np.linalg.norm(Tg[:2, 2] - d["T_goal"][g][:2, 2]) < 1e-5   # reference rule: 1e-4
        assert float(r["f"][g]) < 1e-11
        if int(its[g]) == o["iterations"]:
            exact_tail += 1
            qo = joint_variables_planar_batch(graph, o["x"][None])
            assert np.abs(qg - qo).max() < 1e-6 and np.abs(qg[0] - d["q_sol"][g]).max() < 1e-6
    assert exact_tail >= len(d["seed"]) // 2


# the 3-D solve kernels: wavefront (the default: the per-edge product form, costs.py:186-203 term by term), wavefront with
# the column-form product (gik_template_desc.hessian_form = GIK_HESS_COLUMN, the default until round 5), workgroup,
# node-per-lane
_PATH_PARAMS = {"wave": {"force_block_path": 0}, "wave_column": {"force_block_path": 0, "hessian_form": "column"},
                "block": {"force_block_path": 1}, "npt": {"force_block_path": 2}}


def _hip_traces(d, path):
    from graphik_amd.engine import Template
    T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True, params=_PATH_PARAMS[path])
    r = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"]), trace_cap=48)
    tr = {k: v.cpu().numpy() for k, v in r["trace"].items()}
    return r, [{k: tr[k][g] for k in tr} for g in range(len(d["seed"]))]


@pytest.mark.parametrize("path", ["wave", "wave_column", "block", "npt"])
@pytest.mark.parametrize("name", SCENARIOS_3D)
def test_trajectory_prefix_3d(torch_cuda, name, path):
    """identical discrete decisions and f, |grad| to 1e-8 for the outer iterations
    k <= 20 -- on both kernel paths, for every robot.  3-D solves amplify round-off (~10x every few
    outer iterations), so each golden goal is pinned as far as the REFERENCE reproduces itself:
      (1) strictly -- decisions identical, f / |grad| to 1e-8 against the oracle -- up to
          min(20, K12), K12 = first iteration at which the reference's numpy path (fixture) and the
          oracle (the costs.py loops restated) differ by more than 1e-12, i.e. while they are still
          the same computation; never fewer than 5 iterations.  On the default kernel NO goal may
          flip a discrete decision earlier.  (Until round 5 one per robot was tolerated everywhere
          and blamed on the predicted |r'|^2 of the single-reduction loop.  Round 6 (a) decides a
          near tie of the residual test on the re-summed value, as trust_region.py:560-572 does
          (rtr_solve_one: target2_lo / target2_hi), and (b) looked at the one flip there is -- KUKA
          goal 8, outer iteration 7, column-form and workgroup kernels: the RADIUS test
          |eta + alpha delta|^2 >= Delta^2 (:506-509), which the reference itself takes on a
          recurrence, falls one inner iteration earlier (27: exceeded the radius, instead of 28:
          negative curvature); both end on the same boundary point, the step is rejected either
          way, f / |grad| / Delta stay equal.  No arithmetic that differs from numpy's in the last
          bit can promise the reference's side of such a tie, so on the other kernels ONE flip per
          robot is accepted IF it is exactly that: both solves stop at the trust-region boundary,
          one inner iteration apart, same acceptance);
      (2) in distribution: the HIP trajectory leaves the oracle's (1e-8) no earlier than the
          reference's own numpy path does, for at least 2/3 of the goals, and the summed prefix
          lengths (capped at 20) reach 85 % of the reference pair's."""
    from oracle import c_oracle as co
    from parity_util import (CONTRACT_K, assert_prefix_equal, first_divergence, golden_traj, report,
                             stable_prefix)
    d = load_golden(name)
    r, traces = _hip_traces(d, path)
    its = r["iterations"].cpu().numpy()
    assert d["np_traj_numit"].shape[0] == len(d["seed"])      # every goal has a recorded numpy trajectory
    k_hip, k_ref, pinned, early = [], [], [], []
    for g in range(len(d["seed"])):
        o = co.rtr_solve(d["Y_init"][g], d["D_goal"][g], d["omega"], d["psi_L"], d["psi_U"], True,
                         traj_cap=48)
        ref = golden_traj(d, "np", g)
        n = min(48, int(its[g]), o["iterations"], int(d["iterations"][g]))
        m = max(5, min(CONTRACT_K, stable_prefix(o["traj"], ref, n)))
        kh = first_divergence(traces[g], o["traj"], n)
        k_ref.append(min(CONTRACT_K, first_divergence(o["traj"], ref, n)))
        k_hip.append(min(CONTRACT_K, kh))
        pinned.append(m)
        if kh < m:
            # a discrete decision that flipped inside the stable prefix (refused below; recorded with the evidence
            # that it was a near-tie -- everything before it, and f / |grad| AT it, agree)
            early.append((g, kh, m))
            assert_prefix_equal(traces[g], o["traj"], kh)
            assert kh >= 5 and abs(traces[g]["f_before"][kh] - o["traj"]["f_before"][kh]) <= \
                1e-8 * abs(o["traj"]["f_before"][kh]), (g, kh)
            # ... and only as a near-tie of the radius test: boundary exits on both sides, one inner iteration apart
            assert int(traces[g]["stop"][kh]) in (0, 1) and int(o["traj"]["stop"][kh]) in (0, 1), (g, kh)
            assert abs(int(traces[g]["numit"][kh]) - int(o["traj"]["numit"][kh])) == 1, (g, kh)
            assert int(traces[g]["accept"][kh]) == int(o["traj"]["accept"][kh]), (g, kh)
        else:
            assert_prefix_equal(traces[g], o["traj"], m)
            assert np.array_equal(traces[g]["numit"][:m], ref["numit"][:m])      # the reference itself
    k_hip, k_ref = np.array(k_hip), np.array(k_ref)
    report(f"trajectory_prefix/{name}/{path}", {
        "strictly_pinned_iterations": pinned, "hip_leaves_oracle_at": k_hip.tolist(),
        "reference_np_leaves_oracle_at": k_ref.tolist(), "early_decision_flips": early})
    assert len(early) <= (0 if path == "wave" else 1), early
    assert np.mean(k_hip >= k_ref) >= 2.0 / 3.0, (k_hip, k_ref)
    assert k_hip.sum() >= 0.85 * k_ref.sum(), (k_hip, k_ref)


@pytest.mark.parametrize("path", ["wave", "wave_column", "block", "npt"])
@pytest.mark.parametrize("name", SCENARIOS_3D)
def test_finals_statistical_3d(torch_cuda, name, path):
    """End-to-end parity of the recovered joint configurations (same IK branch,
    max |dq| < 1e-3 rad after wrapping).  Pointwise that bar only exists where the reference meets
    it against itself: its own two code paths (numpy closures vs costs.py loops, fixtures) end
    1e-2 rad apart on the 7-DOF arms, whose solution sets are self-motion curves.  So:
      * UR10 (6 DOF, isolated solutions), pointwise: every converged goal to 1e-3 (block path: 90 %
        to 1e-3, all to 5e-3 -- one slowly converging goal, f = 1e-13, sits at 1.3e-3);
      * the 7-DOF arms in distribution, MEDIAN against MEDIAN over all captured goals (the loop path
        is captured for every goal since round 3): median |dq| of HIP against the reference's numpy
        path no larger than 3x the larger of (the oracle's median |dq| against it, the median |dq|
        between the reference's own two paths) -- KUKA: reference pair 3.0e-3, oracle 2.0e-3,
        wavefront kernel 8.4e-3, workgroup kernel 2.6e-3; LWA4D: 5.7e-4 / 6.6e-4 / 5.0e-4 -- and
        nothing beyond the self-motion scale (0.2 rad) except where the oracle leaves the
        reference's branch as well;
      * convergence class per goal, iteration counts and EE errors as before.
    Round 4 (tools/parity_paths.py -> profiles/r04_parity_by_kernel_path.json) pinned where the wavefront
    kernel's larger drift comes from: the workgroup and node-per-lane kernels form s = y . (W_i - W_j) per
    edge and then s y, as costs.py:186-203 does, so the Gauss-Newton part's round-off lies along y; the
    wavefront kernel's column form multiplies by the rows of B = 2 a y y^T + c I and never forms s.  KUKA
    median |dq|: reference pair 3.0e-3, oracle 2.0e-3, workgroup 2.6e-3, node-per-lane 2.5e-3, wavefront
    8.4e-3.  So the bar is 2x on the two kernels that form s and stays 3x on the wavefront kernel, where
    forming s costs more than it returns (three lanes share a node: +10 % per product measured for the row
    form, rounds 1 and 3, against -5 % products).
    The measured distributions are written to gpurun_out/parity_report.json."""
    from oracle import c_oracle as co
    from parity_util import report, wrap_abs
    from graphik_amd.graphs.graph_revolute import joint_variables_revolute_batch
    d = load_golden(name)
    robot, graph = make_graph(name)
    r, _ = _hip_traces(d, path)
    f = r["f"].cpu().numpy()
    its = r["iterations"].cpu().numpy()
    assert np.array_equal(f < 1e-9, d["f_sol"] < 1e-9)
    assert 0.5 < np.median(its) / np.median(d["iterations"]) < 2.0
    q = joint_variables_revolute_batch(graph, r["x"].cpu().numpy(), d["T_goal"])
    Ts = robot.fk_batch(q)
    pos = np.linalg.norm(Ts[:, :3, 3] - d["T_goal"][:, :3, 3], axis=1)
    conv = d["f_sol"] < 1e-9
    assert np.median(pos[conv]) < 3 * np.median(d["pos_err"][conv]) + 1e-6
    assert np.all(pos[conv] < 5e-3)
    dq = wrap_abs(q - d["q_sol"]).max(axis=1)
    o = co.rtr_solve_batch(d["Y_init"], d["D_goal"], d["omega"], d["psi_L"], d["psi_U"], True, fast=False)
    dq_orc = wrap_abs(joint_variables_revolute_batch(graph, o["x"], d["T_goal"]) - d["q_sol"]).max(axis=1)
    nl = d["loop_q_sol"].shape[0]
    assert nl == len(d["q_sol"])                     # loop path captured for every goal
    dq_ref = wrap_abs(d["q_sol"] - d["loop_q_sol"]).max(axis=1)
    report(f"finals_dq/{name}/{path}", {
        "reference_np_vs_loops": dq_ref.tolist(), "hip_vs_reference_np": dq.tolist(),
        "oracle_vs_reference_np": dq_orc.tolist(), "converged": conv.astype(int).tolist(),
        "iterations_hip": its.tolist(), "iterations_reference": d["iterations"].tolist()})
    if robot.n == 6:
        # isolated solutions: pointwise.  Wherever the reference's two paths agree to 1e-3, so does HIP
        for g in range(nl):
            if dq_ref[g] < 1e-3 and conv[g]:
                assert dq[g] < (1e-3 if path == "wave_column" else 5e-3), (g, dq[g], dq_ref[g])
        if path == "wave_column":
            assert np.all(dq[conv] < 1e-3), dq
        else:
            assert np.mean(dq[conv] < 1e-3) >= 0.9 and np.all(dq[conv] < 5e-3), dq
    else:
        # 7-DOF arms: the solution set of a goal is a self-motion curve and where a solve stops on it
        # is decided by round-off -- with every goal's loop path captured, the reference's own two
        # paths are 3.4e-4 apart on a KUKA goal on which the oracle, restating the loop path line by
        # line, ends 2.2e-2 away: no pointwise statement survives.  Median against median, and the
        # upper quartile against the upper quartile:
        both = conv & (d["loop_f_sol"] < 1e-9)
        for qt in (50, 75):
            bar = (3 if path == "wave_column" else 2) * max(np.percentile(dq_orc[conv], qt), np.percentile(dq_ref[both], qt))
            assert np.percentile(dq[conv], qt) <= bar, (qt, np.percentile(dq[conv], qt), bar)
        # nothing beyond the self-motion scale, except on goals where the oracle or the reference's OWN second path
        # leaves the numpy path's branch as well (LWA4D goal 14: reference pair 0.195 rad apart, node-per-lane
        # kernel 0.200, oracle 0.035)
        assert np.all((dq[conv] < 0.2) | (dq_orc[conv] > 0.05) | (dq_ref[conv] > 0.05)), (dq, dq_orc, dq_ref)


@pytest.mark.parametrize("name", SCENARIOS_3D)
def test_gradient_roundoff_is_horizontal(torch_cuda, name):
    """The exact gradient of a rotation-invariant cost is horizontal, and so is the reference's in
    floating point: G[i] += t, G[j] -= t with t parallel to Y_i - Y_j (costs.py:98-123) leaves a
    vertical part of ~1e-22 near a solution.  tCG never removes a vertical part of its start
    residual (every Hdelta is projected), and 1e-16 there already costs ~12 % more iterations, so
    the device gradient has to have the same property: one residual per term, not per lane."""
    from oracle import c_oracle as co
    d = load_golden(name)
    T = _template(d)
    il = co.limit_inds(d["omega"], d["psi_L"], d["psi_U"])
    rng = np.random.RandomState(5)
    conv = np.nonzero(d["f_sol"] < 1e-12)[0][:6]
    Y = d["Y_sol"][conv] + 1e-7 * rng.randn(len(conv), *d["Y_sol"].shape[1:])
    tg = T.targets_from_D(d["D_goal"][conv])
    G = T.grad(Y, tg).cpu().numpy()
    E = [np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 0.]]), np.array([[0, 0, 1], [0, 0, 0], [-1, 0, 0.]]),
         np.array([[0, 0, 0], [0, 0, 1], [0, -1, 0.]])]
    for b, g in enumerate(conv):
        Qv, _ = np.linalg.qr(np.stack([(Y[b] @ m).ravel() for m in E], axis=1))
        vert = np.linalg.norm(Qv.T @ G[b].ravel())
        Go = co.lgrad(Y[b], d["D_goal"][g], d["omega"], d["psi_L"], d["psi_U"], il)
        vert_o = np.linalg.norm(Qv.T @ Go.ravel())
        nrm = np.linalg.norm(G[b])
        assert 1e-8 < nrm < 1e-4                       # near a solution: residuals ~1e-7
        assert vert < 1e-12 * nrm, (vert, vert_o, nrm)  # per-lane residuals give ~1e-9 * nrm
        assert np.abs(G[b].sum(axis=0)).max() < 1e-12 * nrm   # translation-free as well


_PARAM_SETS = {"maxinner5": {"maxinner": 5}, "mininner4": {"mininner": 4}, "rho": {"rho_prime": 0.2, "rho_regularization": 10.0},
               "maxiter25": {"maxiter": 25}, "maxinner1": {"maxinner": 1}}


@pytest.mark.parametrize("pset", sorted(_PARAM_SETS))
@pytest.mark.parametrize("path", ["wave", "wave_column", "block", "npt", "planar_wave", "planar_quad"])
def test_solver_parameters_against_the_oracle(torch_cuda, path, pset):
    """The trust-region parameters of the descriptor (trust_region.py:85-121: maxinner, mininner, rho_prime,
    rho_regularization, maxiter) away from the reference's defaults, on every solve kernel, against the oracle with the
    same parameters: the cold paths of the single-reduction tCG loop -- inner iterations exhausted with the model test
    pending (maxinner = 5, 1), the residual test held back (mininner = 4) -- and the acceptance rule.  First outer
    iterations decision for decision (inner iteration counts, stopping reasons, acceptance), iteration counts equal where
    maxiter cuts every solve."""
    from oracle import c_oracle as co
    from graphik_amd.engine import Template
    planar = path.startswith("planar")
    d = load_golden("planar10_limits_halfpi" if planar else "lwa4d")
    kw = _PARAM_SETS[pset]
    params = dict({"planar_wave": {"debug_flags": 8192}, "planar_quad": {"debug_flags": 16384}}.get(path) or _PATH_PARAMS[path], **kw)
    use_lim = bool(int(d["use_limits"]))
    T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=int(d["dim"]), use_limits=use_lim, params=params)
    r = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"]), trace_cap=12)
    its = r["iterations"].cpu().numpy()
    for g in range(len(d["seed"])):
        o = co.rtr_solve(d["Y_init"][g], d["D_goal"][g], d["omega"], d["psi_L"], d["psi_U"], use_lim, traj_cap=12, **kw)
        # (planar solves end at the round-off floor, where the last two iterations' stopping reasons are noise: the
        #  planar trajectory tests compare up to the last two as well)
        m = min(12, int(its[g]) - 2, o["iterations"] - 2) if planar else min(5, int(its[g]), o["iterations"])
        for key in ("numit", "stop", "accept"):
            assert np.array_equal(r["trace"][key][g].cpu().numpy()[:m], o["traj"][key][:m]), (g, key)
        if "maxiter" in kw:
            assert int(its[g]) == o["iterations"] or min(int(its[g]), o["iterations"]) < kw["maxiter"]
        if "maxinner" in kw:
            assert r["trace"]["numit"][g].cpu().numpy()[:m].max() <= kw["maxinner"] - 1


@pytest.mark.parametrize("form", ["auto", "column"])
def test_ten_terms_per_node_variant(torch_cuda, form):
    """rtr_wave_kernel<3, 10, ...>: every packaged arm has at most nine terms per node, so the ten-slot variants (four
    owned slots per lane in the per-edge context, one of them padding on most lanes) only run on graphs like this one --
    UR10 + ONE spherical obstacle (N = 17: the obstacle ties to the four base anchors and the two goal nodes).  Known
    answers against the oracle at 1e-12, the first outer iterations decision for decision, convergence class."""
    from oracle import c_oracle as co
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    from graphik_amd.utils.roboturdf import load_ur10
    robot, graph = load_ur10()
    graph.add_spherical_obstacle("o0", np.array([0.6, 0.1, 0.4]), 0.15)
    prob = BatchProblem(graph, use_limits=True, params={"hessian_form": form})
    T = prob.template
    assert T.info["is_block"] == 0 and T.info["max_terms_per_node"] == 10 and T.info["hessian_form"] == int(form == "auto")
    rs = np.random.RandomState(11)
    B = 24
    lb, ub = robot.limits_arrays()
    Tg = robot.fk_batch(lb + (ub - lb) * rs.rand(B, robot.n))
    targets, Y0 = prob.prepare(Tg)
    D, _, _ = prob.assemble(Tg)
    om, pL, pU = prob.omega, prob.psi_L, prob.psi_U
    inds = co.limit_inds(om, pL, pU)
    N = prob.N
    for scale in (1.0, 1e-3):
        Y = np.asarray(Y0[:4]) + scale * rs.randn(4, N, 3)
        W = rs.randn(4, N, 3)
        c, g, h = (T.cost(Y, targets[:4]).cpu().numpy(), T.grad(Y, targets[:4]).cpu().numpy(),
                   T.hess(Y, W, targets[:4]).cpu().numpy())
        for m in range(4):
            assert abs(c[m] - co.lcost(Y[m], D[m], om, pL, pU, inds)) <= 1e-12 * abs(c[m])
            assert rel_err(g[m], co.lgrad(Y[m], D[m], om, pL, pU, inds)) < 1e-12
            assert rel_err(h[m], co.lhess(Y[m], W[m], D[m], om, pL, pU, inds)) < 1e-12
    r = T.solve(Y0, targets, trace_cap=8)
    q = prob.joint_variables(r["x"].cpu().numpy(), Tg)
    pos, rot = prob.pose_errors(q, Tg)
    same = 0
    for gi in range(B):
        o = co.rtr_solve(np.asarray(Y0[gi]), D[gi], om, pL, pU, True, traj_cap=8)
        m = min(5, int(r["iterations"][gi]), o["iterations"])
        assert np.array_equal(r["trace"]["numit"][gi].cpu().numpy()[:m], o["traj"]["numit"][:m]), gi
        same += (float(r["f"][gi]) < 1e-9) == (o["f(x)"] < 1e-9)
    assert same >= B - 1 and np.mean((pos < 0.01) & (rot < 0.01)) > 0.8


@pytest.mark.parametrize("path", ["wave", "wave_column", "block"])
@pytest.mark.parametrize("theta,kappa", [(0.5, 0.1), (2.0, 0.5)])
def test_theta_and_kappa_against_the_oracle(torch_cuda, path, theta, kappa):
    """The tCG stopping rule |r| <= |r0| min(|r0|^theta, kappa) (trust_region.py:572; the reference passes theta = 1,
    kappa = 0.1) with other parameters: the kernels compiled for any theta (pow() in the per-solve setup; per-edge and
    column-form wavefront kernels, the workgroup kernel) against the oracle with the same parameters -- the first outer
    iterations decision for decision, the convergence class, the outer iteration counts in distribution."""
    from oracle import c_oracle as co
    from graphik_amd.engine import Template
    d = load_golden("lwa4d")
    T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True,
                               params=dict(_PATH_PARAMS[path], theta=theta, kappa=kappa))
    assert T.info["hessian_form"] == int(path != "wave_column") and T.info["node_per_lane"] == 0
    r = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"]), trace_cap=16)
    its = r["iterations"].cpu().numpy()
    f = r["f"].cpu().numpy()
    its_o = []
    for g in range(len(d["seed"])):
        o = co.rtr_solve(d["Y_init"][g], d["D_goal"][g], d["omega"], d["psi_L"], d["psi_U"], True, traj_cap=16,
                         theta=theta, kappa=kappa)
        its_o.append(o["iterations"])
        m = min(5, int(its[g]), o["iterations"])
        assert np.array_equal(r["trace"]["numit"][g].cpu().numpy()[:m], o["traj"]["numit"][:m]), (g, theta, kappa)
        assert np.array_equal(r["trace"]["stop"][g].cpu().numpy()[:m], o["traj"]["stop"][:m]), (g, theta, kappa)
        assert (f[g] < 1e-9) == (o["f(x)"] < 1e-9)
    assert 0.8 < np.median(its) / np.median(its_o) < 1.25
    # and the parameters do something: the default's inner iteration counts differ (late in a solve for theta < 1: the
    # superlinear target only binds once |r0| is small)
    r1 = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True,
                                params=_PATH_PARAMS[path]).solve(d["Y_init"], T.targets_from_D(d["D_goal"]), trace_cap=16)
    assert not np.array_equal(r1["inner_total"].cpu().numpy(), r["inner_total"].cpu().numpy())


@pytest.mark.parametrize("path", ["wave", "wave_column", "block", "npt"])
def test_effort_parity_ur10(torch_cuda, path):
    """Same work as the reference's algorithm, not only the same answers: on random UR10 goals no
    tCG solve runs into maxinner (the reference's never do; a search direction that keeps the
    vertical round-off of the gradient does, late in a solve, in one solve out of five), outer
    iterations agree in distribution and the Hessian products stay within 12 % of the oracle's
    (measured +8 %: the column-form product puts its round-off outside range(J^T), NOTEBOOK 2; the
    bound was 15 % in round 2).  The kernels that form s = y . w per edge like the reference -- workgroup and
    node-per-lane -- measure +2 % (1024 goals per arm, profiles/r04_parity_by_kernel_path.json: wavefront
    +6.6 / +7.1 / +7.9 %, workgroup +5.1 / +1.9 / +2.2 %, node-per-lane +3.5 / +1.6 / +2.1 % on KUKA / LWA4D /
    UR10) and are held to 6 %; the wavefront kernel keeps 12 %: forming s there needs a three-lane sum per
    term (+25 % instructions) or whole-row gathers (+10 % cycles, measured) to save 5 % of the products."""
    from oracle import c_oracle as co
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph("ur10")
    prob = BatchProblem(graph, use_limits=True, params=_PATH_PARAMS[path])
    assert prob.template.info["is_block"] == int(not path.startswith("wave"))
    assert prob.template.info["hessian_form"] == int(path != "wave_column")
    B = 192
    rng = np.random.RandomState(3)
    Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(B, robot.n))
    targets, Y0 = prob.prepare(Tg)
    r = prob.template.solve(Y0, targets, trace_cap=3000)
    its = r["iterations"].cpu().numpy()
    hv = r["inner_total"].cpu().numpy()
    stop = r["trace"]["stop"].cpu().numpy()
    assert not any((stop[b, :its[b]] == 4).any() for b in range(B))
    D, _, _ = prob.assemble(Tg)
    o = [co.rtr_solve(np.asarray(Y0[b]), D[b], prob.omega, prob.psi_L, prob.psi_U, True) for b in range(B)]
    its_o = np.array([x["iterations"] for x in o])
    hv_o = np.array([x["inner_total"] for x in o])
    assert np.array_equal(its < 3000, its_o < 3000)
    assert 0.9 < np.median(its) / np.median(its_o) < 1.1
    assert 0.95 < hv.sum() / hv_o.sum() < (1.12 if path == "wave_column" else 1.06), (hv.sum(), hv_o.sum())


_TAIL_ORACLE = {}


@pytest.mark.parametrize("path", ["wave", "wave_column", "block", "npt"])
def test_effort_parity_kuka_tail(torch_cuda, path):
    """The TAIL of the effort distribution on the redundant arm: 8.3 % of KUKA goals run to maxiter and
    hold a third of c4's Hessian products, so the share of such goals is where parity and throughput meet.
    8192 random goals, oracle from the device's start points.

    Rounds 4-5 measured p90 of the outer iterations on 512 goals and found 1.25x on the workgroup, node-per-lane and
    (round 6, first run) the new per-edge wavefront kernels against 1.01x for round 5's per-edge form.  Round 6 bisected
    it (tools/tail_bisect.py, tools/exp/strict_bisect.sh, NOTEBOOK 11.2): with p90 sitting 1.7 % of the goals away
    from the 8.3 % that stop at 3000, a surplus of nine goals in 512 moves it by 450 iterations -- and between any two
    correct renderings of the same arithmetic ~3.5 % of the goals change class, in BOTH directions.  On 8192 goals
    every summation grouping of gradient, product and cost (fourteen builds) and every kernel path has the oracle's
    share of maxiter goals to within the binomial noise of the flips: wavefront default 668 / 677 (to / from maxiter
    142 / 151), column 695 (152 / 134), workgroup 691 (145 / 131), node-per-lane 696 (154 / 135); p90 0.99x / 1.005x /
    1.05x / 1.02x.  So the statistics here are the ones that have a noise model: the flips must be balanced to three
    standard deviations, p90 and p75 within the bars on a sample where p90 is resolved (+-4 %), Hessian products
    within +4 % (column form: +8 %; measured +1.7 / +5.2 / +1.9 / +1.1 %)."""
    from oracle import c_oracle as co
    from graphik_amd.engine import Template
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    from parity_util import report
    robot, graph = make_graph("kuka")
    prob = BatchProblem(graph, use_limits=True)
    B = 8192
    rng = np.random.RandomState(3)
    Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(B, robot.n))
    targets, Y0 = prob.prepare(Tg)
    if "o" not in _TAIL_ORACLE:
        D, _, _ = prob.assemble(Tg)
        _TAIL_ORACLE["o"] = co.rtr_solve_batch(np.asarray(Y0), D, prob.omega, prob.psi_L, prob.psi_U, True, fast=False)
    o = _TAIL_ORACLE["o"]
    T = Template.from_matrices(prob.omega, prob.psi_L, prob.psi_U, k=3, use_limits=True, params=_PATH_PARAMS[path])
    r = T.solve(Y0, targets)
    its, hv = r["iterations"].cpu().numpy(), r["inner_total"].cpu().numpy().astype(np.int64)
    mx, mxo = its >= 3000, o["iterations"] >= 3000
    to, frm = int((mx & ~mxo).sum()), int((~mx & mxo).sum())
    p90, p90_o = np.percentile(its, 90), np.percentile(o["iterations"], 90)
    p75, p75_o = np.percentile(its, 75), np.percentile(o["iterations"], 75)
    hvr = hv.sum() / o["inner_total"].sum()
    report(f"effort_tail/kuka/{path}", {"goals": B, "p90_outer": [float(p90), float(p90_o)], "p75_outer": [float(p75), float(p75_o)],
                                        "at_maxiter": [int(mx.sum()), int(mxo.sum())], "to_maxiter": to, "from_maxiter": frm,
                                        "same_class": float(np.mean(mx == mxo)), "hv_ratio": float(hvr)})
    assert abs(to - frm) <= 3.0 * np.sqrt(to + frm) + 1, (to, frm)        # no systematic drift into (or out of) maxiter
    assert np.mean(mx == mxo) >= 0.95
    assert p90 <= {"wave": 1.06, "wave_column": 1.08}.get(path, 1.10) * p90_o, (p90, p90_o)
    assert 0.95 * p75_o <= p75 <= 1.05 * p75_o, (p75, p75_o)
    assert 0.97 < hvr < (1.08 if path == "wave_column" else 1.04), hvr


# ---- batched pipeline -------------------------------------------------------------------------
@pytest.mark.parametrize("name,B", [("lwa4d", 256), ("planar10_limits_halfpi", 256)])
def test_solve_batch_random_goals(torch_cuda, name, B):
    from graphik_amd.solvers.riemannian_solver import solve_batch
    robot, graph = make_graph(name)
    rng = np.random.RandomState(11)
    lb, ub = robot.limits_arrays()
    Q = lb + (ub - lb) * rng.rand(B, robot.n)
    Tg = robot.fk_batch(Q)
    q, Y, info = solve_batch(graph, Tg, use_limits=True)
    assert np.all(np.isfinite(Y)) and np.all(np.isfinite(q))
    assert np.all(info["stop"] != 2)
    ok = (info["pos_err"] < 0.01) & (info["rot_err"] < 0.01)   # the reference's success rule
    if graph.dim == 3:
        assert ok.mean() > 0.85 and np.median(info["pos_err"]) < 1e-3
    else:
        assert ok.mean() > 0.95 and np.median(info["pos_err"]) < 1e-6


def test_full_size_batch_properties(torch_cuda):
    """BASELINE configs[1]: LWA4D, 4096 random goals.  Size-independent checks: every problem
    terminates by a legal stopping rule, results are finite, reruns are bit-identical, the
    recovered configurations realise the goal pose, and statistics match the oracle on a
    sub-sample."""
    import torch
    from oracle import c_oracle as co
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph("lwa4d")
    prob = BatchProblem(graph, use_limits=True)
    B = 4096
    rng = np.random.RandomState(0)
    Q = -np.pi + 2 * np.pi * rng.rand(B, robot.n)
    Tg = robot.fk_batch(Q)
    targets, Y0 = prob.prepare(Tg)
    r1 = prob.template.solve(Y0, targets)
    r2 = prob.template.solve(Y0, targets)
    torch.cuda.synchronize()
    x1, x2 = r1["x"].cpu().numpy(), r2["x"].cpu().numpy()
    assert np.array_equal(x1, x2)
    stop = r1["stop"].cpu().numpy()
    its = r1["iterations"].cpu().numpy()
    gn = r1["gradnorm"].cpu().numpy()
    assert np.all(np.isfinite(x1)) and np.all((stop == 0) | (stop == 1))
    assert np.all(gn[stop == 0] < 0.5e-9) and np.all(its[stop == 1] == 3000)
    q = prob.joint_variables(x1, Tg)
    pos, rot = prob.pose_errors(q, Tg)
    assert np.median(pos) < 5e-4 and np.mean((pos < 0.01) & (rot < 0.01)) > 0.9
    # oracle on the first 256 problems from the same start points: convergence class, outer iterations and Hessian
    # products (round 6: the default kernel forms the reference's product -- LWA4D products +1.6 % on 512 goals,
    # class 100 %; the 48-goal sample and the 0.6-1.6 band of rounds 1-5 dated from the column form)
    n = 256
    D, _, _ = prob.assemble(Tg[:n])
    o = co.rtr_solve_batch(np.asarray(Y0[:n]), D, prob.omega, prob.psi_L, prob.psi_U, True, fast=False)
    f = r1["f"].cpu().numpy()[:n]
    hv = r1["inner_total"].cpu().numpy()[:n].astype(np.int64)
    assert np.mean((f < 1e-9) == (o["f(x)"] < 1e-9)) >= 0.97
    assert 0.9 < np.median(its[:n]) / np.median(o["iterations"]) < 1.1
    assert 0.97 < hv.sum() / o["inner_total"].sum() < 1.05, hv.sum() / o["inner_total"].sum()


@pytest.mark.parametrize("name", ["lwa4d", "kuka", "lwa4d_block", "lwa4d_npt2", "lwa4d_npt1"])
def test_checkpoint_resume_is_bit_identical(torch_cuda, monkeypatch, name):
    """After a rejected step the reference's next tCG solve repeats the previous one up to the
    smaller radius; the engine resumes from a checkpoint instead.  Same arithmetic, so the results
    (points, costs, every counter the reference would report, the per-iteration trace) must equal
    those of actually rerunning tCG (debug_flags = 16), bit for bit; only the executed work differs."""
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    from graphik_amd.engine import Template
    block = "_" in name                      # the workgroup and node-per-lane kernels keep their checkpoint too
    path = {"block": 1, "npt2": 2, "npt1": 2}.get(name.split("_")[-1], 0)     # gik_template_desc.force_block_path
    npt_flags = 2048 if name.endswith("npt1") else 0                            # one wavefront per problem
    robot, graph = make_graph(name.split("_")[0])
    prob = BatchProblem(graph, use_limits=True)
    rng = np.random.RandomState(21)
    Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(96 if block else 384, robot.n))
    targets, Y0 = prob.prepare(Tg)
    def template(debug_flags):
        return Template.from_matrices(prob.omega, prob.psi_L, prob.psi_U, k=3, use_limits=True,
                                      params={"force_block_path": path, "debug_flags": debug_flags | npt_flags})

    def run(debug_flags=0):
        r = template(debug_flags).solve(Y0, targets, trace_cap=96)
        out = {k: r[k].cpu().numpy() for k in ("x", "f", "gradnorm", "iterations", "inner_total",
                                              "stop", "n_accept", "inner_executed")}
        out.update({"t_" + k: v.cpu().numpy() for k, v in r["trace"].items()})
        return out

    a = run()
    b = run(debug_flags=16)
    for k in a:
        if k != "inner_executed":
            assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert np.all(b["inner_executed"] >= b["inner_total"])       # rerun: every counted product runs
    assert np.all(a["inner_executed"] <= b["inner_executed"])
    assert a["inner_executed"].sum() < 0.95 * b["inner_executed"].sum()   # measured: -11 ... -14 %


@pytest.mark.parametrize("path,flags", [(1, 0), (1, 64), (2, 0), (2, 64), (2, 64 | 2048)])
#                          path 1: workgroup kernel, 2: node-per-lane kernel (2048: one wavefront per problem);
#                          64: with the clique closed form (base + goal nodes)
def test_time_slicing_is_bit_identical(torch_cuda, monkeypatch, path, flags):
    """The workgroup-per-problem kernel re-queues a problem that has not met a stopping rule after
    slice_outer_its outer iterations (default 96) behind everything that is waiting, so that the long
    problems of a batch do not start last.  A solve is exactly resumable from (x, Delta,
    counters), so the results must not depend on the slice length; only the executed work may (the
    tCG checkpoint is dropped at a slice boundary)."""
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    from graphik_amd.engine import Template
    robot, graph = make_graph("lwa4d")
    prob = BatchProblem(graph, use_limits=True)
    rng = np.random.RandomState(12)
    Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(600, robot.n))     # more problems than workgroups
    targets, Y0 = prob.prepare(Tg)
    keys = ("x", "f", "gradnorm", "iterations", "inner_total", "stop", "n_accept")
    runs = {}
    for sl in ("0", "256", "24"):
        tpl = Template.from_matrices(prob.omega, prob.psi_L, prob.psi_U, k=3, use_limits=True,
                                     params={"force_block_path": path, "slice_outer_its": int(sl),
                                             "debug_flags": flags})
        assert (tpl.info["node_per_lane"] != 0) == (path == 2)
        r = tpl.solve(Y0, targets, trace_cap=40)
        runs[sl] = {k: r[k].cpu().numpy() for k in keys + ("inner_executed",)}
        runs[sl]["numit"] = r["trace"]["numit"].cpu().numpy()
    assert (runs["0"]["iterations"] > 256).sum() > 20          # problems that do get re-queued
    for sl in ("256", "24"):
        for k in keys + ("numit",):
            assert np.array_equal(runs["0"][k], runs[sl][k], equal_nan=True), (sl, k)
        assert runs[sl]["inner_executed"].sum() >= runs["0"]["inner_executed"].sum()


def test_results_independent_of_persistent_grid(torch_cuda, monkeypatch):
    """The solve kernel is persistent (waves claim problems from a queue); the number of resident
    waves is a scheduling choice (one or two per SIMD, gik_solve_batch) and must not change a
    single bit of any result."""
    d = load_golden("lwa4d")
    reps = 40
    Y0 = np.tile(d["Y_init"], (reps, 1, 1))
    outs = []
    for wpc in (1, 4, 8):
        T = _template(d, maxiter=60, waves_per_cu=wpc)
        tg = np.tile(T.targets_from_D(d["D_goal"]), (reps, 1))
        r = T.solve(Y0, tg)
        outs.append((r["x"].cpu().numpy(), r["iterations"].cpu().numpy(), r["inner_total"].cpu().numpy()))
    for o in outs[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(outs[0], o))
    # and every replica of a goal gives the same answer
    x = outs[0][0].reshape(reps, -1, *outs[0][0].shape[1:])
    assert np.array_equal(x, np.broadcast_to(x[0], x.shape))


def test_edge_cases(torch_cuda):
    import torch
    d = load_golden("lwa4d")
    T = _template(d, maxiter=50)
    tg = T.targets_from_D(d["D_goal"][:3])
    empty = T.solve(d["Y_init"][:0], tg[:0])
    assert empty["x"].shape[0] == 0
    one = T.solve(d["Y_init"][:1], tg[:1])
    assert int(one["iterations"][0]) == 50 and int(one["stop"][0]) == 1
    bad = d["Y_init"][:3].copy()
    bad[1, 0, 0] = np.nan
    r = T.solve(bad, tg)
    assert r["stop"].cpu().numpy().tolist() == [1, 2, 1]
    # odd batch sizes through the persistent work queue
    for B in (2, 63, 65, 1025):
        reps = -(-B // 16)
        Yi = np.tile(d["Y_init"], (reps, 1, 1))[:B]
        tt = np.tile(T.targets_from_D(d["D_goal"]), (reps, 1))[:B]
        rr = T.solve(Yi, tt)
        ref = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"]))
        n = min(B, 16)
        # which wave picks which problem off the queue must not matter
        assert np.array_equal(rr["x"].cpu().numpy()[:n], ref["x"].cpu().numpy()[:n])
        assert torch.isfinite(rr["x"]).all()


def test_drop_in_single_goal(torch_cuda):
    """solve_with_riemannian(graph, T_goal) as the reference's README / example call it."""
    from graphik_amd.solvers.riemannian_solver import solve_with_riemannian, RiemannianSolver
    from graphik_amd.utils import dgp
    robot, graph = make_graph("lwa4d")
    np.random.seed(1)
    q_goal = robot.random_configuration()
    T_goal = robot.pose(q_goal, f"p{robot.n}")
    q_sol, Y = solve_with_riemannian(graph, T_goal, use_jit=False)
    assert set(q_sol) == {f"p{i}" for i in range(1, 8)} and Y.shape == (18, 3)
    T_sol = robot.pose(q_sol, "p7")
    # the same call was captured from the reference (tests/golden/lwa4d.npz, seed 1): same goal, the EE error
    # within the reference's own band on this goal (3x its 2.1e-4 m / 2.0e-4 rad), and the same IK branch --
    # joint angles within the self-motion scale of the captured solution (7-DOF: where on the self-motion curve
    # a solve stops is decided by its start point, and the device's MDS start differs from LAPACK's by the
    # eigenvector sign rule)
    from parity_util import wrap_abs
    d = load_golden("lwa4d")
    g = int(np.nonzero(d["seed"] == 1)[0][0])
    assert np.allclose(T_goal.as_matrix(), d["T_goal"][g], atol=1e-12)
    pos_err = np.linalg.norm(T_sol.trans - T_goal.trans)
    R_err = T_goal.as_matrix()[:3, :3].T @ T_sol.as_matrix()[:3, :3]
    rot_err = np.arccos(np.clip((np.trace(R_err) - 1.0) / 2.0, -1.0, 1.0))
    assert pos_err < 3 * d["pos_err"][g] + 1e-5 and rot_err < 3 * d["rot_err"][g] + 1e-5, (pos_err, rot_err)
    dq = wrap_abs(robot.q_to_array(q_sol) - d["q_sol"][g]).max()
    assert dq < 0.2, dq
    q2, _ = solve_with_riemannian(graph, T_goal, jit=False)   # README spelling
    assert q2 is not None
    # RiemannianSolver.solve as the planar example scripts call it
    probot, pgraph = make_graph("planar10_nolimits")
    np.random.seed(21)
    qg = probot.random_configuration()
    Tg = probot.pose(qg, "p10")
    G = pgraph.from_pose(Tg)
    solver = RiemannianSolver(pgraph)
    lb, ub = dgp.bound_smoothing(G)
    info = solver.solve(dgp.distance_matrix_from_graph(G), dgp.adjacency_matrix_from_graph(G),
                        bounds=(lb, ub), jit=False)
    assert set(info) >= {"x", "f(x)", "time", "gradnorm", "iterations"}
    qs = pgraph.joint_variables(dgp.graph_from_pos(info["x"], pgraph.node_ids), {"p10": Tg})
    assert np.linalg.norm(probot.pose(qs, "p10").trans - Tg.trans) < 1e-4  # test_chain_2d_new.py:86
    with pytest.raises(Exception):
        solver.solve(dgp.distance_matrix_from_graph(G), dgp.adjacency_matrix_from_graph(G))
    cost, egrad, ehess = solver.create_cost(dgp.distance_matrix_from_graph(G),
                                            dgp.adjacency_matrix_from_graph(G))
    assert cost(info["x"]) < 1e-20 and egrad(info["x"]).shape == (13, 2)
    # ... and on the class, as the reference's static methods are called (riemannian_solver.py:77-78)
    c2, g2, h2 = RiemannianSolver.create_cost(dgp.distance_matrix_from_graph(G), dgp.adjacency_matrix_from_graph(G))
    assert c2(info["x"]) == cost(info["x"]) and np.array_equal(g2(info["x"]), egrad(info["x"]))
    W = np.random.RandomState(0).randn(13, 2)
    assert np.array_equal(h2(info["x"], W), ehess(info["x"], W))


# ---- device pre/post-processing (gik_prepare_batch / gik_recover_batch / gik_ik_batch) ----------
@pytest.mark.parametrize("name", ["lwa4d", "ur10", "kuka", "planar10_limits_halfpi", "planar10_nolimits",
                                  "ur10_table"])
def test_device_prepare_matches_host(torch_cuda, name):
    """from_pose + bound_smoothing + generate_initialization on the device against the host
    restatement.  Targets and bounds-derived data must agree to 1e-12; Y_init is compared through
    the sign-canonical host variant (eigenvector signs are LAPACK accidents in the reference, and
    its MDS rank rule depends on them -- see dgp.generate_initialization_batch)."""
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    from graphik_amd.utils import dgp
    d = load_golden(name)
    robot, graph = make_graph(name)
    use_lim = bool(int(d["use_limits"]))
    prob = BatchProblem(graph, use_limits=use_lim)
    assert prob.device_pipeline
    rng = np.random.RandomState(3)
    lb_q, ub_q = robot.limits_arrays()
    n_rand = 112 if graph.number_of_nodes() <= 32 else 23      # N = 116 runs the workgroup-per-goal kernel
    Tg = np.concatenate([d["T_goal"], robot.fk_batch(lb_q + (ub_q - lb_q) * rng.rand(n_rand, robot.n))])
    tg_d, Y_d, K_d = prob.template.prepare(Tg, return_K=True)
    tg_d, Y_d, K_d = tg_d.cpu().numpy(), Y_d.cpu().numpy(), K_d.cpu().numpy()
    D, lo, up = prob.assemble(Tg)
    tg_h = prob.template.targets_from_D(D)
    assert np.abs(tg_d - tg_h).max() <= 1e-12 * np.abs(tg_h).max()
    lb, ub = dgp.floyd_warshall_bounds(lo, up)
    Y_h, info_h = dgp.generate_initialization_batch(lb, ub, graph.dim, prob.omega, canonical=True,
                                                    return_info=True)
    cls = _classify_init(Y_d, K_d, Y_h, info_h)
    assert cls["unexplained"] == 0, cls
    assert cls["identical"] >= 0.97 * len(Tg), cls
    ok = cls["mask_identical"]
    assert np.abs(np.abs(Y_d[ok]) - np.abs(Y_h[ok])).max() < 1e-7
    assert np.all(K_d >= graph.dim) and np.all(K_d <= graph.number_of_nodes())


def _classify_init(Y_d, K_d, Y_h, info_h, tol=1e-8):
    """Every goal's device initial point against a host rendering of generate_initialization:
    identical (Gram matrices equal to 1e-8 -- Y_init is defined up to the signs of its columns),
    or explained by a DIFFERENT MDS column count K whose cause is visible in the data -- an
    eigenvalue of the rank matrix within a factor 4 of MDS's 1e-8 threshold ("near_threshold"), or,
    when comparing with the reference's LAPACK-sign rendering, the sign dependence of that matrix
    ("sign_rule": K differs although no eigenvalue is near the threshold) -- or unexplained."""
    B = len(Y_d)
    G_d = Y_d @ np.swapaxes(Y_d, 1, 2)
    G_h = Y_h @ np.swapaxes(Y_h, 1, 2)
    err = np.abs(G_d - G_h).reshape(B, -1).max(axis=1) / np.abs(G_h).reshape(B, -1).max(axis=1)
    same = err < tol
    near = np.any((info_h["ev_rank"] > 0.25e-8) & (info_h["ev_rank"] < 4e-8), axis=1)
    kdiff = K_d != info_h["K"]
    out = {"goals": B, "identical": int(same.sum()),
           "near_threshold": int((~same & near).sum()),
           "sign_rule": int((~same & ~near & kdiff).sum()),
           "unexplained": int((~same & ~near & ~kdiff).sum()),
           "K_differs": int(kdiff.sum()), "max_err_identical": float(err[same].max()) if same.any() else 0.0}
    out["mask_identical"] = same
    return out


@pytest.mark.parametrize("name", ["lwa4d", "ur10", "kuka", "planar10_nolimits", "planar10_limits_pi",
                                  "planar10_limits_halfpi", "ur10_table"])
def test_device_prepare_against_reference_fixtures(torch_cuda, name):
    """a10 / a11 against data captured from the reference itself (tests/golden, one hop):
      * bound_smoothing (dgp.py:192-231): device lb, ub == the reference's networkx output, 1e-12;
      * the Gram matrix's spectrum == eigvalsh of the Gram built from the reference's lb, ub, 1e-10;
      * generate_initialization (riemannian_solver.py:67-75): Gram(device Y_init) == Gram(reference
        Y_init) for EVERY goal whose MDS column count K equals the reference's.  K is the one thing
        that cannot be pinned: dgp.py:166-167 takes it from eigh() of the non-symmetric eigenvector
        factor (its lower triangle), which depends on the signs LAPACK happens to return.  Each goal
        is classified (identical / K differs by the sign rule / eigenvalue at the threshold /
        unexplained) and the test fails on any unexplained one; the census goes to the report."""
    from parity_util import report
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    from graphik_amd.utils import dgp
    d = load_golden(name)
    robot, graph = make_graph(name)
    prob = BatchProblem(graph, use_limits=bool(int(d["use_limits"])))
    assert prob.device_pipeline
    r = prob.template.prepare_debug(d["T_goal"])
    lb_d, ub_d = r["lb"].cpu().numpy(), r["ub"].cpu().numpy()
    assert np.abs(lb_d - d["lb"]).max() < 1e-12 and np.abs(ub_d - d["ub"]).max() < 1e-12
    Dr = (d["lb"] + 0.9 * (d["ub"] - d["lb"])) ** 2
    ev_ref = np.linalg.eigvalsh(dgp.gram_from_distance_matrix(Dr))
    ev_dev = np.sort(r["eig"].cpu().numpy()[:, 0, :], axis=1)
    assert np.abs(ev_dev - ev_ref).max() <= 1e-10 * np.abs(ev_ref).max()
    # the reference's own K: LAPACK signs, same numpy/LAPACK build as the capture
    Y_ref, info_ref = dgp.generate_initialization_batch(d["lb"], d["ub"], graph.dim, d["omega"],
                                                        canonical=False, return_info=True)
    assert np.abs(Y_ref @ np.swapaxes(Y_ref, 1, 2) - d["Y_init"] @ np.swapaxes(d["Y_init"], 1, 2)).max() < 1e-8
    Y_d, K_d = r["Y_init"].cpu().numpy(), r["K"].cpu().numpy()
    cls = _classify_init(Y_d, K_d, d["Y_init"], info_ref)
    mask = cls.pop("mask_identical")
    report(f"device_init_vs_reference/{name}", dict(cls, K_device=K_d.tolist(), K_reference=info_ref["K"].tolist()))
    assert cls["unexplained"] == 0, cls
    assert np.all(mask[K_d == info_ref["K"]] | np.any((info_ref["ev_rank"] > 0.25e-8) &
                                                     (info_ref["ev_rank"] < 4e-8), axis=1)[K_d == info_ref["K"]])


@pytest.mark.parametrize("name", SCENARIOS_3D)
def test_end_to_end_statistics_from_device_init(torch_cuda, name):
    """The device's initial point differs from the reference's for the goals where MDS's column
    count depends on LAPACK's eigenvector signs (test above), so end-to-end parity is checked in
    distribution on 1024 random goals per robot: the whole device pipeline (gik_ik_batch: device
    init -> solve -> recover) against the oracle started from the REFERENCE-rule initial point
    (host mirror with LAPACK signs, which reproduces the captured Y_init): convergence rate,
    outer-iteration quantiles, Hessian products, EE error and success rate (the reference's
    pos < 0.01 and rot < 0.01)."""
    from oracle import c_oracle as co
    from parity_util import report
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    from graphik_amd.utils import dgp
    robot, graph = make_graph(name)
    prob = BatchProblem(graph, use_limits=True)
    B = 1024
    rng = np.random.RandomState(17)
    lb_q, ub_q = robot.limits_arrays()
    Tg = robot.fk_batch(lb_q + (ub_q - lb_q) * rng.rand(B, robot.n))
    r = prob.template.ik(Tg)
    its = r["iterations"].cpu().numpy()
    hv = r["inner_total"].cpu().numpy()
    stop = r["stop"].cpu().numpy()
    pos, rot = r["pos_err"].cpu().numpy(), r["rot_err"].cpu().numpy()
    D, lo, up = prob.assemble(Tg)
    lbs, ubs = dgp.floyd_warshall_bounds(lo, up)
    Y_ref = dgp.generate_initialization_batch(lbs, ubs, 3, prob.omega, canonical=False)
    o = co.rtr_solve_batch(Y_ref, D, prob.omega, prob.psi_L, prob.psi_U, True, fast=True)
    q_o = prob.joint_variables(o["x"], Tg)
    pos_o, rot_o = prob.pose_errors(q_o, Tg)
    its_o, hv_o = o["iterations"], o["inner_total"]
    conv, conv_o = stop == 0, its_o < 3000
    succ, succ_o = (pos < 0.01) & (rot < 0.01), (pos_o < 0.01) & (rot_o < 0.01)
    stats = {
        "converged": [float(conv.mean()), float(conv_o.mean())],
        "outer_median": [float(np.median(its)), float(np.median(its_o))],
        "outer_p90": [float(np.percentile(its, 90)), float(np.percentile(its_o, 90))],
        "hv_total": [float(hv.sum()), float(hv_o.sum())],
        "pos_err_median": [float(np.median(pos)), float(np.median(pos_o))],
        "pos_err_p90": [float(np.percentile(pos, 90)), float(np.percentile(pos_o, 90))],
        "success": [float(succ.mean()), float(succ_o.mean())],
        "order": "[device pipeline from device init, oracle from reference-rule init]"}
    report(f"end_to_end_statistics/{name}", stats)
    assert abs(conv.mean() - conv_o.mean()) < 0.03
    assert abs(succ.mean() - succ_o.mean()) < 0.03
    assert 0.93 < np.median(its) / np.median(its_o) < 1.07      # measured 0.997 .. 1.000
    assert 0.75 < np.percentile(its, 90) / np.percentile(its_o, 90) < 1.33
    assert 0.92 < hv.sum() / hv_o.sum() < 1.12      # measured 1.055 (KUKA), 1.069 (LWA4D), 1.079 (UR10)
    assert 0.7 < np.median(pos) / np.median(pos_o) < 1.4


@pytest.mark.parametrize("name,a_global", [("lwa4d", False), ("lwa4d", True), ("planar10_limits_halfpi", False)])
def test_block_prepare_kernel_equals_wave_kernel(torch_cuda, monkeypatch, name, a_global):
    """The workgroup-per-goal prepare kernel (graphs beyond one wavefront's LDS) performs the same
    operations in the same order per matrix element as the wavefront kernel, so on a graph both can
    take, targets, initial points and MDS ranks agree bit for bit -- with its work matrix in LDS
    (N <= 123) and with it in the global slab (GIK_PREP_A_GLOBAL, read at attach: the variant
    graphs of 124..128 nodes get)."""
    if a_global:
        monkeypatch.setenv("GIK_PREP_A_GLOBAL", "1")
    # (graphs of at most 16 nodes default to the four-goals-per-wavefront kernel, which sums two of its inner
    # products in another order -- test_quad_prepare_kernel_against_wave_kernel; this test is about the other two)
    monkeypatch.setenv("GIK_NO_PREP_QUAD", "1")
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph(name)
    rng = np.random.RandomState(8)
    lb_q, ub_q = robot.limits_arrays()
    Tg = robot.fk_batch(lb_q + (ub_q - lb_q) * rng.rand(700, robot.n))    # more goals than workgroups
    out = []
    for force in (False, True):
        prob = BatchProblem(graph, use_limits=True, force_block_prepare=force)
        tg, Y, K = prob.template.prepare(Tg, return_K=True)
        out.append((tg.cpu().numpy(), Y.cpu().numpy(), K.cpu().numpy()))
    for a, b in zip(*out):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("name", ["planar10_limits_pi", "planar10_nolimits", "planar10_limits_halfpi"])
def test_quad_prepare_kernel_against_wave_kernel(torch_cuda, monkeypatch, name):
    """Four goals per wavefront (prep_quad_kernel, the default for graphs of at most 16 nodes) against one
    (prep_wave_kernel, GIK_NO_PREP_QUAD read at attach): targets bit for bit (no inner product in them), MDS
    column counts equal, initial points to round-off (the Jacobi threshold and the Householder reflectors sum in
    another order; the K x K Jacobi runs the schedule of the wavefront's largest K) -- for batch sizes that leave
    slots of the last wavefront empty, and the solves that start from either end on the same iteration."""
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph(name)
    use_lim = not name.endswith("nolimits")
    rng = np.random.RandomState(11)
    lb_q, ub_q = robot.limits_arrays()
    Tg = robot.fk_batch(lb_q + (ub_q - lb_q) * rng.rand(2051, robot.n))
    quad = BatchProblem(graph, use_limits=use_lim)
    monkeypatch.setenv("GIK_NO_PREP_QUAD", "1")
    wave = BatchProblem(graph, use_limits=use_lim)
    assert quad.template.info["goals_per_wave"] == 4 and wave.template.info["goals_per_wave"] == 1
    for B in (1, 2, 3, 5, 2051):
        tq, Yq, Kq = [x.cpu().numpy() for x in quad.template.prepare(Tg[:B], return_K=True)]
        tw, Yw, Kw = [x.cpu().numpy() for x in wave.template.prepare(Tg[:B], return_K=True)]
        assert np.array_equal(tq, tw)
        assert np.all(np.isfinite(Yq))
        # MDS counts the eigenvalues above 1e-8 of a matrix whose small columns ARE of size 1e-8 (the square roots
        # of Gram eigenvalues that are zero up to round-off, 1e-16): on 0.5-2.5 % of the goals the count depends on the
        # last bit -- in the reference too -- and the columns it adds or drops weigh 1e-16 in the result below
        assert np.mean(Kq == Kw) >= 0.95 and np.abs(Kq - Kw).max() <= 2
        assert np.abs(Yq - Yw).max() < 1e-10
        Gq, Gw = Yq @ Yq.transpose(0, 2, 1), Yw @ Yw.transpose(0, 2, 1)
        assert np.abs(Gq - Gw).max() < 1e-10 * np.abs(Gw).max()
    rq = quad.template.solve(Yq[:256], tq[:256])
    rw = wave.template.solve(Yw[:256], tw[:256])
    # (below the round-off floor a redundant chain drifts along its solution set: the points are compared where
    # that has not started, the answers by their cost)
    assert np.mean(rq["iterations"].cpu().numpy() == rw["iterations"].cpu().numpy()) > 0.95
    fq, fw = rq["f"].cpu().numpy(), rw["f"].cpu().numpy()
    assert np.mean((fq < 1e-11) == (fw < 1e-11)) > 0.99 and np.mean(fq < 1e-11) > 0.9   # (tight limits: a few local minima)


@pytest.mark.parametrize("links", [4, 7, 13])
def test_planar_chains_of_other_sizes_on_the_quad_kernels(torch_cuda, monkeypatch, links):
    """Chains of 4 / 7 / 13 links (N = 7 / 10 / 16 nodes: the generic prep_quad_kernel<0>, and 16 = the largest
    graph the four-per-wavefront kernels take) through the whole device pipeline, against the same pipeline on
    the one-per-wavefront kernels: targets bit for bit, initial points to round-off, every goal reached by both,
    iteration counts equal on nearly all goals, and every solution reproduces its goal position."""
    from graphik_amd.robots import RobotPlanar
    from graphik_amd.graphs import ProblemGraphPlanar
    from graphik_amd.utils import list_to_variable_dict
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    lim = np.pi * np.ones(links)
    robot = RobotPlanar({"link_lengths": list_to_variable_dict(0.5 + 0.1 * np.arange(links)),
                         "theta": list_to_variable_dict(np.zeros(links)),
                         "joint_limits_upper": list_to_variable_dict(lim),
                         "joint_limits_lower": list_to_variable_dict(-lim), "num_joints": links})
    graph = ProblemGraphPlanar(robot)
    rng = np.random.RandomState(links)
    Tg = robot.fk_batch(-lim + 2 * lim * rng.rand(301, links))
    quad = BatchProblem(graph, use_limits=True, params={"debug_flags": 16384})      # (at any batch size)
    assert quad.template.info["problems_per_wave"] == 4 and quad.template.info["goals_per_wave"] == 4
    monkeypatch.setenv("GIK_NO_PREP_QUAD", "1")
    wave = BatchProblem(graph, use_limits=True, params={"debug_flags": 8192})
    assert wave.template.info["problems_per_wave"] == 1 and wave.template.info["goals_per_wave"] == 1
    tq, Yq = [x.cpu().numpy() for x in quad.template.prepare(Tg)]
    tw, Yw = [x.cpu().numpy() for x in wave.template.prepare(Tg)]
    assert np.array_equal(tq, tw) and np.abs(Yq - Yw).max() < 1e-10
    rq, rw = quad.template.ik(Tg), wave.template.ik(Tg)
    fq, fw = rq["f"].cpu().numpy(), rw["f"].cpu().numpy()
    assert np.mean((fq < 1e-11) == (fw < 1e-11)) > 0.99 and np.mean(fq < 1e-11) > 0.95
    assert np.mean(rq["iterations"].cpu().numpy() == rw["iterations"].cpu().numpy()) > 0.95
    ok = fq < 1e-11
    assert np.all(rq["pos_err"].cpu().numpy()[ok] < 1e-5)
    assert np.all(np.isfinite(rq["q"].cpu().numpy()))


def test_lds_allowance_survives_later_templates(torch_cuda):
    """The dynamic-LDS allowance is a property of a kernel, not of a launch: a template that needs
    less (LWA4D on the workgroup kernels: 3 KB work matrix, 10 KB of solver state) created AFTER one
    that needs more (table scene: 105 KB work matrix in the prepare kernel) must not take it away."""
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph("ur10_table")
    rng = np.random.RandomState(2)
    lb_q, ub_q = robot.limits_arrays()
    Tg = robot.fk_batch(lb_q + (ub_q - lb_q) * rng.rand(8, robot.n))
    big = BatchProblem(graph, use_limits=True)
    before = [x.cpu().numpy() for x in big.template.prepare(Tg)]
    r0 = big.template.solve(before[1], before[0])["x"].cpu().numpy()
    robot2, graph2 = make_graph("lwa4d")
    small = BatchProblem(graph2, use_limits=True, force_block_prepare=True, params={"force_block_path": 1})
    lb2, ub2 = robot2.limits_arrays()
    Tg2 = robot2.fk_batch(lb2 + (ub2 - lb2) * rng.rand(8, robot2.n))
    tg2, Y2 = small.template.prepare(Tg2)
    assert np.all(np.isfinite(small.template.solve(Y2, tg2)["x"].cpu().numpy()))
    after = [x.cpu().numpy() for x in big.template.prepare(Tg)]
    for a, b in zip(before, after):
        assert np.array_equal(a, b)
    assert np.array_equal(r0, big.template.solve(after[1], after[0])["x"].cpu().numpy())


@pytest.mark.parametrize("name", ["lwa4d", "ur10", "kuka", "planar10_limits_halfpi"])
def test_device_recover_matches_host(torch_cuda, name):
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    d = load_golden(name)
    robot, graph = make_graph(name)
    prob = BatchProblem(graph, use_limits=bool(int(d["use_limits"])))
    q_d, pe_d, re_d = prob.template.recover(d["Y_sol"], d["T_goal"])
    q_d = q_d.cpu().numpy()
    assert np.abs(q_d - d["q_sol"]).max() < 1e-9          # the reference's joint_variables
    q_h = prob.joint_variables(d["Y_sol"], d["T_goal"])
    pe_h, re_h = prob.pose_errors(q_h, d["T_goal"])
    assert np.abs(pe_d.cpu().numpy() - pe_h).max() < 1e-9
    assert np.abs(pe_d.cpu().numpy() - d["pos_err"]).max() < 1e-9
    assert np.abs(re_d.cpu().numpy() - re_h).max() < 1e-6
    assert np.abs(re_d.cpu().numpy() - d["rot_err"]).max() < 1e-6
    # perturbed / mirrored realisations as well (reference tests/test_joint_variables.py:30-53)
    rng = np.random.RandomState(0)
    Yp = d["Y_sol"] + 1e-3 * rng.randn(*d["Y_sol"].shape)
    q_d2 = prob.template.recover(Yp, d["T_goal"])[0].cpu().numpy()
    assert np.abs(q_d2 - prob.joint_variables(Yp, d["T_goal"])).max() < 1e-9


@pytest.mark.parametrize("name,B", [("lwa4d", 512), ("planar10_limits_halfpi", 512)])
def test_device_pipeline_end_to_end(torch_cuda, name, B):
    """gik_ik_batch = prepare -> solve -> recover without leaving the device."""
    import torch
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    robot, graph = make_graph(name)
    prob = BatchProblem(graph, use_limits=True)
    rng = np.random.RandomState(21)
    lb_q, ub_q = robot.limits_arrays()
    Tg = robot.fk_batch(lb_q + (ub_q - lb_q) * rng.rand(B, robot.n))
    r = prob.template.ik(Tg)
    torch.cuda.synchronize()
    tg, Y0 = prob.template.prepare(Tg)
    r2 = prob.template.solve(Y0, tg)
    assert np.array_equal(r["x"].cpu().numpy(), r2["x"].cpu().numpy())   # same stages, same bits
    pos, rot = r["pos_err"].cpu().numpy(), r["rot_err"].cpu().numpy()
    q = r["q"].cpu().numpy()
    ph, rh = prob.pose_errors(q, Tg)
    assert np.abs(pos - ph).max() < 1e-9
    ok = (pos < 0.01) & (rot < 0.01)
    if graph.dim == 3:
        assert ok.mean() > 0.9 and np.median(pos) < 1e-3
    else:
        assert ok.mean() > 0.97 and np.median(pos) < 1e-6


# ---- workgroup-per-problem path (graphs with N*k > 64) -------------------------------------------
# debug_flags of the workgroup-per-problem path: 64 = closed form for rigid cliques from 4 nodes up
# (default 16: the small robots' base + goal nodes then exercise it), 128 = closed form off,
# 256 = closed form with the dense D w product even when the targets are distances of points
@pytest.mark.parametrize("path", [1, 2])      # gik_template_desc.force_block_path: workgroup kernels / node-per-lane kernel
@pytest.mark.parametrize("name,flags", [("ur10_table", 0), ("ur10_table", 256), ("ur10_table", 128), ("lwa4d", 0), ("lwa4d", 64),
                                        ("kuka", 64), ("planar10_limits_halfpi", 0), ("ur10_table", 2048), ("ur10", 64 | 2048)])
def test_block_path_known_answers(torch_cuda, name, flags, path):
    """UR10 + table_environment(): 116 nodes, 5612 residual terms (BASELINE configs[2]) runs on
    the workgroup-per-problem kernels; the small graphs are forced onto the same kernels so both
    code paths are checked against the same golden vectors.  The table scene's 106 anchors form a
    rigid clique whose Hessian-vector product is evaluated in closed form (gik_block.hip.h); the
    same golden values pin it, and the direct sum (flags 128)."""
    from graphik_amd.engine import Template
    d = load_golden(name)
    use_lim = bool(int(d["use_limits"]))
    if path == 2 and (int(d["dim"]) == 2 or (name == "ur10_table" and flags == 128)):
        pytest.skip("node-per-lane kernel: 3-D graphs with at most 256 terms outside a rigid clique")
    if path == 1 and flags & 2048:
        pytest.skip("2048 selects the one-wavefront layout of the node-per-lane kernel")
    T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=int(d["dim"]),
                               use_limits=use_lim, params={"force_block_path": path, "debug_flags": flags})
    assert T.info["is_block"] == 1 and T.info["node_per_lane"] == (0 if path == 1 else (1 if flags & 2048 else 2))
    key = "lim" if use_lim else "nolim"
    tg = T.targets_from_D(d["D_goal"][0])
    Y, W = d["kat_Y"], d["kat_W"]
    assert rel_err(T.cost(Y, tg).cpu().numpy(), d[f"kat_{key}_loop_cost"]) < 1e-12
    assert rel_err(T.grad(Y, tg).cpu().numpy(), d[f"kat_{key}_loop_grad"]) < 1e-12
    assert rel_err(T.hess(Y, W, tg).cpu().numpy(), d[f"kat_{key}_loop_hess"]) < 1e-12
    assert rel_err(T.proj(Y, W).cpu().numpy(), d["kat_proj"]) < 1e-12


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("name,flags", [("lwa4d", 0), ("lwa4d", 64), ("ur10", 64), ("planar10_limits_halfpi", 0),
                                        ("lwa4d", 64 | 2048)])
def test_block_path_trajectories_match_wave_path(torch_cuda, name, flags, path):
    """The two kernel paths sum the same terms in different orders, so on 3-D goals they part where
    round-off is amplified to 1e-8 -- like any two renderings of the algorithm (stable_prefix).  Strict:
    decisions identical and f to 1e-7 for 5 (3-D) / 8 (planar) outer iterations on every goal; and the
    iteration at which each goal's traces really part (decisions or f, |grad| at 1e-8) is reported and
    bounded from below in the median."""
    from graphik_amd.engine import Template
    d = load_golden(name)
    use_lim = bool(int(d["use_limits"]))
    if (path == 2 and int(d["dim"]) == 2) or (path == 1 and flags & 2048):
        pytest.skip("node-per-lane kernel: 3-D graphs; 2048 is its one-wavefront layout")
    kw = dict(k=int(d["dim"]), use_limits=use_lim)
    Tw = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], **kw)
    Tb = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"],
                                params={"force_block_path": path, "debug_flags": flags}, **kw)
    tg = Tw.targets_from_D(d["D_goal"])
    from parity_util import first_divergence, report
    rw = Tw.solve(d["Y_init"], tg, trace_cap=32)
    rb = Tb.solve(d["Y_init"], tg, trace_cap=32)
    m = 5 if int(d["dim"]) == 3 else 8
    for key in ("numit", "stop", "accept", "Delta"):
        assert np.array_equal(rw["trace"][key].cpu().numpy()[:, :m], rb["trace"][key].cpu().numpy()[:, :m])
    assert np.allclose(rw["trace"]["f_before"].cpu().numpy()[:, :m],
                       rb["trace"]["f_before"].cpu().numpy()[:, :m], rtol=1e-7)
    fw, fb = rw["f"].cpu().numpy(), rb["f"].cpu().numpy()
    assert np.array_equal(fw < 1e-9, fb < 1e-9)
    tw = {k: v.cpu().numpy() for k, v in rw["trace"].items()}
    tb = {k: v.cpu().numpy() for k, v in rb["trace"].items()}
    iw, ib = rw["iterations"].cpu().numpy(), rb["iterations"].cpu().numpy()
    n = [min(32, int(iw[g]), int(ib[g])) for g in range(len(iw))]      # (planar solves end after 7-13 iterations)
    part = [first_divergence({k: tw[k][g] for k in tw}, {k: tb[k][g] for k in tb}, n[g]) for g in range(len(iw))]
    report(f"trajectory_prefix/{name}/{'block' if path == 1 else 'npt'}_vs_wave/flags{flags}", {"paths_part_at": part, "compared": n})
    if int(d["dim"]) == 3:      # (planar solves reach f ~ 1e-28 within 8-14 iterations: relative 1e-8 on f and
        #                         |grad| loses its meaning there; the strict check above is the planar bar)
        assert all(p >= min(m, q) for p, q in zip(part, n)), (part, n)
        assert np.median(part) >= m + 2, part


# ---- four planar problems per wavefront (rtr_quad_kernel) ---------------------------------------
@pytest.mark.parametrize("name", SCENARIOS_2D)
def test_quad_kernel_against_one_problem_per_wavefront(torch_cuda, name):
    """The same solves on both planar kernels.  They sum inner products in different orders (per node
    first / per unknown), so the bar is that of two renderings of the algorithm: every decision of the
    first 8 outer iterations identical, the iteration counts equal, the points equal to 1e-9; and a slot
    of the wavefront must not care what its three neighbours do: every problem of a batch of 37 (ragged:
    the last wavefront holds one problem) ends bit-identical to the same problem solved alone."""
    from graphik_amd.engine import Template
    d = load_golden(name)
    kw = dict(k=2, use_limits=bool(int(d["use_limits"])))
    # (debug_flags 16384: the four-problem kernel at any batch size -- by default batches below 12 problems per CU
    # run the one-problem kernel, which is faster there)
    Tq = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], params={"debug_flags": 16384}, **kw)
    Tw = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], params={"debug_flags": 8192}, **kw)
    assert Tq.info["problems_per_wave"] == 4 and Tw.info["problems_per_wave"] == 1
    tg = Tq.targets_from_D(d["D_goal"])
    rq = Tq.solve(d["Y_init"], tg, trace_cap=32)
    rw = Tw.solve(d["Y_init"], tg, trace_cap=32)
    for key in ("numit", "stop", "accept", "Delta"):
        assert np.array_equal(rq["trace"][key].cpu().numpy()[:, :8], rw["trace"][key].cpu().numpy()[:, :8]), key
    for key in ("iterations", "stop", "n_accept"):
        assert np.array_equal(rq[key].cpu().numpy(), rw[key].cpu().numpy()), key
    assert np.array_equal(rq["inner_total"].cpu().numpy(), rq["inner_executed"].cpu().numpy())
    assert np.abs(rq["x"].cpu().numpy() - rw["x"].cpu().numpy()).max() < 1e-9
    assert float(rq["f"].max()) < 1e-11
    # ragged batch, every problem against itself alone
    G = len(d["Y_init"])
    rng = np.random.RandomState(5)
    idx = rng.randint(0, G, size=37)
    Y0 = d["Y_init"][idx] + 1e-3 * rng.randn(37, *d["Y_init"].shape[1:])
    tgb = np.asarray(tg)[idx]
    rb = Tq.solve(Y0, tgb)
    xb = rb["x"].cpu().numpy()
    for g in (0, 1, 17, 35, 36):
        r1 = Tq.solve(Y0[g:g + 1], tgb[g:g + 1])
        assert np.array_equal(r1["x"].cpu().numpy()[0], xb[g]), g
        for key in ("f", "gradnorm", "iterations", "inner_total", "stop", "n_accept", "stepsize"):
            assert r1[key].cpu().numpy()[0] == rb[key].cpu().numpy()[g], (g, key)


def test_quad_kernel_stopping_rules(torch_cuda):
    """maxiter (stop 1, counters as the reference leaves them), a NaN start point (stop 2, no iteration, the
    other slots of the wavefront unaffected) and solves pushed far below the round-off floor."""
    from graphik_amd.engine import Template
    d = load_golden("planar10_limits_pi")
    kw = dict(k=2, use_limits=True)
    tg = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], **kw).targets_from_D(d["D_goal"])
    G = len(d["Y_init"])
    for flags in (16384, 8192):
        T3 = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], params={"maxiter": 3, "debug_flags": flags}, **kw)
        r = T3.solve(d["Y_init"], tg, trace_cap=8)
        assert np.all(r["iterations"].cpu().numpy() == 3) and np.all(r["stop"].cpu().numpy() == 1)
        if flags == 16384:
            ref = r
        else:
            assert np.array_equal(r["inner_total"].cpu().numpy(), ref["inner_total"].cpu().numpy())
            assert np.allclose(r["x"].cpu().numpy(), ref["x"].cpu().numpy(), atol=1e-11)
    T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], params={"debug_flags": 16384}, **kw)
    Y0 = np.concatenate([d["Y_init"]] * 3)[:21].copy()
    tgb = np.concatenate([np.asarray(tg)] * 3)[:21]
    clean = T.solve(Y0, tgb)
    Y0n = Y0.copy()
    Y0n[5, 3, 1] = np.nan
    r = T.solve(Y0n, tgb)
    st = r["stop"].cpu().numpy()
    assert st[5] == 2 and int(r["iterations"][5]) == 0
    keep = np.arange(21) != 5
    assert np.array_equal(st[keep], clean["stop"].cpu().numpy()[keep])
    assert np.array_equal(r["x"].cpu().numpy()[keep], clean["x"].cpu().numpy()[keep])
    # far below the round-off floor (mingradnorm = 0: every slot runs to maxiter = 40, steps rejected for dozens
    # of passes, radii shrinking to 1e-20): the masked loops must neither hang nor produce a NaN
    Tm = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"],
                                params={"maxiter": 40, "mingradnorm": 0.0, "debug_flags": 16384}, **kw)
    rm = Tm.solve(Y0, tgb)
    assert np.all(rm["iterations"].cpu().numpy() == 40) and np.all(rm["stop"].cpu().numpy() == 1)
    assert float(rm["f"].max()) < 1e-20 and np.all(np.isfinite(rm["x"].cpu().numpy()))

def test_clique_closed_form_against_direct_sum(torch_cuda):
    """Table scene, random points and directions (not only the golden ones): the closed form of
    the 106-anchor clique against the direct sum of the same kernel, cost / gradient / Hessian
    product at 1e-12, near a solution too (where every c_ij of the clique is a difference of O(1)
    numbers); the solves end in the same class with similar effort."""
    from graphik_amd.engine import Template
    d = load_golden("ur10_table")
    kw = dict(k=3, use_limits=True)
    Tc = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], params={"debug_flags": 0}, **kw)
    Td = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], params={"debug_flags": 128}, **kw)
    tg = Tc.targets_from_D(d["D_goal"])
    G = len(tg)
    rng = np.random.RandomState(3)
    rd = Td.solve(d["Y_init"], tg)
    rc = Tc.solve(d["Y_init"], tg)
    xs = rd["x"].cpu().numpy()
    for Y in (d["Y_init"], xs, xs + 1e-6 * rng.randn(*xs.shape), d["Y_init"] + 0.3 * rng.randn(*xs.shape)):
        W = rng.randn(*Y.shape)
        assert rel_err(Tc.cost(Y, tg).cpu().numpy(), Td.cost(Y, tg).cpu().numpy()) < 1e-12
        gd = Td.grad(Y, tg).cpu().numpy()
        assert np.abs(Tc.grad(Y, tg).cpu().numpy() - gd).max() < 1e-12 * max(1.0, np.abs(gd).max())
        hd = Td.hess(Y, W, tg).cpu().numpy()
        assert np.abs(Tc.hess(Y, W, tg).cpu().numpy() - hd).max() < 1e-12 * np.abs(hd).max()
    fd, fc = rd["f"].cpu().numpy(), rc["f"].cpu().numpy()
    assert np.array_equal(fd < 1e-9, fc < 1e-9)
    # the scene's targets are distances of points: the closed form runs without the dense product
    assert np.all(rc["flags"].cpu().numpy() == 1) and np.all(rd["flags"].cpu().numpy() == 0)
    itd, itc = rd["iterations"].cpu().numpy().astype(float), rc["iterations"].cpu().numpy().astype(float)
    assert np.all(np.abs(itc / itd - 1.0) < 0.25), (itd, itc)
    hvd, hvc = float(rd["inner_total"].sum()), float(rc["inner_total"].sum())
    assert abs(hvc / hvd - 1.0) < 0.15, (hvd, hvc)


@pytest.mark.parametrize("n_clique,n_other,euclid", [(21, 19, True), (16, 3, True), (45, 0, True), (106, 10, True),
                                                    (21, 19, False), (64, 8, False), (40, 5, "planar"),
                                                    (190, 10, True), (150, 12, False), (160, 9, "planar")])
def test_clique_detection_on_synthetic_graphs(torch_cuda, n_clique, n_other, euclid):
    """Synthetic 3-D graphs on the workgroup path: a rigid clique whose size is not a multiple of
    four, with lower / upper hinges ON TOP of some clique pairs (they stay in the slot tables), other
    nodes chained to the clique and to each other, the clique's nodes scattered over the node
    numbering -- cost, gradient and Hessian product against the CPU oracle at 1e-12, closed form and
    direct sum, and one solve ending at the same cost.  With Euclidean targets (the squared distances
    of a point set, as every scene gives) the kernel finds coordinates for the clique and replaces
    its dense D w product by 12 more moments; targets that are NOT distances of points in space (each
    scaled by its own factor), or whose points lie in a plane (no frame for the trilateration), must
    take the dense product and give the same answers."""
    from oracle import c_oracle as co
    from graphik_amd.engine import Template
    rng = np.random.RandomState(11 + n_clique)
    N = n_clique + n_other
    P = rng.randn(N, 3) * np.array([1.0, 0.8, 0.5])
    Dtrue = ((P[:, None] - P[None]) ** 2).sum(-1)
    perm = rng.permutation(N)
    clique, other = perm[:n_clique], perm[n_clique:]
    om = np.zeros((N, N)); pL = np.zeros((N, N)); pU = np.zeros((N, N))
    for a in range(n_clique):
        for b in range(a + 1, n_clique):
            i, j = clique[a], clique[b]
            om[i, j] = om[j, i] = 1.0
            # hinges on a clique pair as well (graphs beyond 128 nodes only run on the node-per-lane kernel: at most
            # 256 terms outside the clique, so there the hinged pairs stay among the clique's first 40 nodes)
            if (a + b) % 11 == 0 and (N <= 128 or b < 40):
                pL[i, j] = pL[j, i] = 0.9 * Dtrue[i, j]
                pU[i, j] = pU[j, i] = 1.2 * Dtrue[i, j]
    for q, i in enumerate(other):
        for j in list(clique[(3 * q) % n_clique:][:4]) + list(other[:q][-2:]):
            if rng.rand() < 0.7:
                om[i, j] = om[j, i] = 1.0
            else:
                pL[i, j] = pL[j, i] = 0.5 * Dtrue[i, j]
                pU[i, j] = pU[j, i] = 1.5 * Dtrue[i, j]
    il = co.limit_inds(om, pL, pU)
    D = Dtrue * om
    if euclid == "planar":
        P[clique, 2] = 0.25          # the clique's points in one plane (other nodes off it)
        D = ((P[:, None] - P[None]) ** 2).sum(-1) * om
    elif not euclid:
        S = 1.0 + 0.1 * rng.rand(N, N)
        D = D * (S + S.T)
    Y = P + 0.3 * rng.randn(N, 3)
    W = rng.randn(N, 3)
    want = (co.lcost(Y, D, om, pL, pU, il), co.lgrad(Y, D, om, pL, pU, il), co.lhess(Y, W, D, om, pL, pU, il))
    fs = []
    # (the largest case only fits the LDS with the clique taken out of the slot tables)
    # both kernel families: (force_block_path, debug_flags); the node-per-lane kernel needs the clique
    # taken out (at most 256 terms outside it, at most 16 per node), in both of its layouts
    cases = [(1, f) for f in ((0, 256, 128) if n_clique < 64 else (0, 256))] + [(2, 0), (2, 256), (2, 2048)]
    if N > 128:
        # beyond 128 nodes (round 5): the node-per-lane kernel on FOUR wavefronts, the clique's target triangle in
        # global memory; force_block_path 0 (automatic) must pick it, the 512-thread kernels must refuse
        cases = [(0, 0), (2, 0), (2, 256)]
        with pytest.raises(RuntimeError, match="128 nodes"):
            Template.from_matrices(om, pL, pU, k=3, use_limits=True, params={"force_block_path": 1})
    for path, flags in cases:
        try:
            T = Template.from_matrices(om, pL, pU, k=3, use_limits=True,
                                       params={"force_block_path": path, "debug_flags": flags})
        except RuntimeError as e:
            assert N <= 128 and path == 2 and "node-per-lane" in str(e), e      # a node with more than 16 terms outside the clique
            continue
        assert (T.info["node_per_lane"] != 0) == (path == 2 or N > 128)
        if N > 128:
            assert T.info["node_per_lane"] == 4
        tg = T.targets_from_D(D)
        assert rel_err(float(T.cost(Y, tg)[0]), want[0]) < 1e-12
        assert rel_err(T.grad(Y, tg)[0].cpu().numpy(), want[1]) < 1e-12
        assert rel_err(T.hess(Y, W, tg)[0].cpu().numpy(), want[2]) < 1e-12
        r = T.solve(Y[None], tg[None] if tg.ndim == 1 else tg)
        fs.append(float(r["f"][0]))
        # which rendering of the clique's D w product ran (gik_stats.flags bit 0)
        assert int(r["flags"][0]) == (1 if (euclid is True and not flags & (128 | 256)) else 0)
    if euclid is True:
        assert max(fs) < 1e-9
    else:                       # inconsistent targets / hinges: a positive minimum, the same for both renderings
        assert max(fs) > 1e-6 and abs(fs[0] - fs[-1]) < 1e-6 * fs[0]


def test_busy_nodes_fall_back_to_block_path(torch_cuda):
    """A graph that fits a wavefront (N*k <= 64) but whose busiest node carries more residual
    terms than the largest compiled slot count is solved by the workgroup-per-problem kernels
    instead of being refused: planar, 20 points, an equality on every pair of neighbours and a
    lower + upper hinge on every other pair (up to 2 * 17 + 2 terms per node > 31)."""
    from oracle import c_oracle as co
    from graphik_amd.engine import Template
    rng = np.random.RandomState(7)
    N = 20
    P = rng.randn(N, 2) * 2.0
    Dtrue = ((P[:, None] - P[None]) ** 2).sum(-1)
    om = np.zeros((N, N)); pL = np.zeros((N, N)); pU = np.zeros((N, N))
    for i in range(N):
        for j in range(i + 1, N):
            if j - i <= 1:
                om[i, j] = om[j, i] = 1.0
            else:
                pL[i, j] = pL[j, i] = 0.5 * Dtrue[i, j]
                pU[i, j] = pU[j, i] = 1.5 * Dtrue[i, j]
    T = Template.from_matrices(om, pL, pU, k=2, use_limits=True)
    assert T.maxdeg > 31
    il = co.limit_inds(om, pL, pU)
    D = Dtrue * om
    tg = T.targets_from_D(D)
    Y = P + 0.8 * rng.randn(N, 2)       # hinges partly active
    W = rng.randn(N, 2)
    assert rel_err(float(T.cost(Y, tg)[0]), co.lcost(Y, D, om, pL, pU, il)) < 1e-12
    assert rel_err(T.grad(Y, tg)[0].cpu().numpy(), co.lgrad(Y, D, om, pL, pU, il)) < 1e-12
    assert rel_err(T.hess(Y, W, tg)[0].cpu().numpy(), co.lhess(Y, W, D, om, pL, pU, il)) < 1e-12
    r = T.solve(Y[None], tg[None] if tg.ndim == 1 else tg, trace_cap=8)
    o = co.rtr_solve(Y, D, om, pL, pU, True, traj_cap=8)
    m = 4
    assert np.array_equal(r["trace"]["numit"][0][:m].cpu().numpy(), o["traj"]["numit"][:m])
    assert np.array_equal(r["trace"]["stop"][0][:m].cpu().numpy(), o["traj"]["stop"][:m])
    assert float(r["f"][0]) < 1e-9 and int(r["stop"][0]) == 0


@pytest.mark.parametrize("path", [0, 1])      # 0: automatic = the node-per-lane kernel; 1: the workgroup kernel
def test_ur10_table_solve(torch_cuda, path):
    """BASELINE configs[2] on the node-per-lane kernel (the default for this graph) and on the
    workgroup-per-problem kernel: the 8 captured goals from the
    reference's own Y_init.  Per goal: trajectory prefix against the oracle at the contract's
    tolerance (as far as the reference's numpy path and the oracle are the same computation, see
    test_trajectory_prefix_3d), the reference's convergence class (one of the goals ends in a local
    minimum, f = 1e-3), iteration count, and the EE error of the recovered configuration."""
    from oracle import c_oracle as co
    from parity_util import (CONTRACT_K, assert_prefix_equal, first_divergence, golden_traj, report,
                             stable_prefix)
    from graphik_amd.engine import Template
    d = load_golden("ur10_table")
    robot, graph = make_graph("ur10_table")
    G = len(d["seed"])
    assert G >= 8
    T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=3, use_limits=True, params={"force_block_path": path})
    assert T.info["is_block"] == 1 and T.info["node_per_lane"] == (2 if path == 0 else 0)
    tg = T.targets_from_D(d["D_goal"])
    r = T.solve(np.concatenate([d["Y_init"], d["Y_init"]]), np.concatenate([tg, tg]), trace_cap=48)
    x = r["x"].cpu().numpy()
    assert np.array_equal(x[:G], x[G:])                           # deterministic across workgroups
    tr = {k: v.cpu().numpy() for k, v in r["trace"].items()}
    f, its, stop = r["f"].cpu().numpy()[:G], r["iterations"].cpu().numpy()[:G], r["stop"].cpu().numpy()[:G]
    pinned, k_hip, k_ref = [], [], []
    for g in range(G):
        o = co.rtr_solve(d["Y_init"][g], d["D_goal"][g], d["omega"], d["psi_L"], d["psi_U"], True,
                         traj_cap=48)
        ref = golden_traj(d, "np", g)
        n = min(48, int(its[g]), o["iterations"], int(d["iterations"][g]))
        m = max(5, min(CONTRACT_K, stable_prefix(o["traj"], ref, n)))
        assert_prefix_equal({k: tr[k][g] for k in tr}, o["traj"], m)
        pinned.append(m)
        k_ref.append(min(CONTRACT_K, first_divergence(o["traj"], ref, n)))
        k_hip.append(min(CONTRACT_K, first_divergence({k: tr[k][g] for k in tr}, o["traj"], n)))
        assert (f[g] < 1e-9) == (d["f_sol"][g] < 1e-9)
        if d["f_sol"][g] < 1e-9:
            assert stop[g] == 0 and 0.5 < its[g] / int(d["iterations"][g]) < 2.0
            q = graph.joint_variables(x[g], d["T_goal"][g])
            T_sol = robot.pose(q, "p6")
            assert np.linalg.norm(T_sol.trans - d["T_goal"][g][:3, 3]) < 3 * d["pos_err"][g] + 1e-4
    report("trajectory_prefix/ur10_table/" + ("npt" if path == 0 else "block"), {"strictly_pinned_iterations": pinned,
           "hip_leaves_oracle_at": k_hip, "reference_np_leaves_oracle_at": k_ref,
           "iterations_hip": its.tolist(), "iterations_reference": d["iterations"].tolist()})
    assert np.mean(np.array(k_hip) >= np.array(k_ref)) >= 2.0 / 3.0
    assert sum(k_hip) >= 0.85 * sum(k_ref)


def test_scene_with_200_spheres_beyond_128_nodes(torch_cuda):
    """graph_base.py:182-211 takes any number of spheres; until round 5 gik_template_create refused graphs of more
    than 128 nodes (a scene with 113 obstacles raised).  UR10 + table_environment(n_width=12, n_height=14) -- the
    reference's own generator, 200 spheres, N = 216, 21 162 terms, a rigid clique of 206 anchors -- runs on the
    node-per-lane kernel with FOUR wavefronts per problem (clique targets in global memory): known answers against the
    oracle at 1e-12, the first outer iterations decision for decision, and a batch through the drop-in entry point
    solve_batch with the reference's success rule.  Round 6: prepare and recover of such graphs run on the device too
    (prep_block_kernel<false, 256>: work matrix in a six-matrix slab, always range-compressed; until then 0.13 s of
    host work per goal around a 35 ms solve): device bounds against the host's bound smoothing at 1e-12, Gram(Y_init)
    at 1e-8, the device's joint angles against the host recovery, and the whole pipeline (gik_ik_batch) at more than
    25 goals per second end to end."""
    from oracle import c_oracle as co
    from graphik_amd.solvers.riemannian_solver import BatchProblem, solve_batch, solve_with_riemannian
    from graphik_amd.utils import table_environment
    from graphik_amd.utils.roboturdf import load_ur10
    robot, graph = load_ur10()
    for idx, obs in enumerate(table_environment(n_width=12, n_height=14)):
        graph.add_spherical_obstacle(f"o{idx}", obs[0], obs[1])
    N = graph.number_of_nodes()
    assert N == 216
    prob = BatchProblem(graph, use_limits=True)
    T = prob.template
    assert T.info["node_per_lane"] == 4 and T.info["n_clique"] == 206 and prob.device_pipeline
    assert T.info["prepare_is_block"] == 1
    rs = np.random.RandomState(0)
    B = 24
    Tg = robot.fk_batch(-np.pi + 2 * np.pi * rs.rand(B, robot.n))
    targets, Y0 = prob.prepare(Tg[:3])
    D, _, _ = prob.assemble(Tg[:3])
    om, pL, pU = prob.omega, prob.psi_L, prob.psi_U
    inds = co.limit_inds(om, pL, pU)
    for scale in (1.0, 1e-3):
        Y = np.asarray(Y0[:2]) + scale * rs.randn(2, N, 3)
        W = rs.randn(2, N, 3)
        c, g, h = (T.cost(Y, targets[:2]).cpu().numpy(), T.grad(Y, targets[:2]).cpu().numpy(),
                   T.hess(Y, W, targets[:2]).cpu().numpy())
        for m in range(2):
            assert abs(c[m] - co.lcost(Y[m], D[m], om, pL, pU, inds)) <= 1e-12 * abs(c[m])
            assert rel_err(g[m], co.lgrad(Y[m], D[m], om, pL, pU, inds)) < 1e-12
            assert rel_err(h[m], co.lhess(Y[m], W[m], D[m], om, pL, pU, inds)) < 1e-12
    r = T.solve(Y0[:3], targets[:3], trace_cap=8)
    assert np.all(r["flags"].cpu().numpy() & 1)           # the scene's targets are distances of points: closed form
    for gi in range(3):
        o = co.rtr_solve(np.asarray(Y0[gi]), D[gi], om, pL, pU, True, traj_cap=8)
        assert np.array_equal(r["trace"]["numit"][gi].cpu().numpy()[:4], o["traj"]["numit"][:4])
        assert (float(r["f"][gi]) < 1e-9) == (o["f(x)"] < 1e-9)
    # device prepare against the host path: bounds, targets, the initial point up to the Gram matrix (the eigenvectors of
    # a degenerate pair are free to rotate; MDS column counts differ where they are taken from 1e-8-sized noise columns)
    from graphik_amd.utils import dgp
    dbg = T.prepare_debug(Tg[:3])
    _, lo, up = prob.assemble(Tg[:3])
    lb_h, ub_h = dgp.floyd_warshall_bounds(lo, up)
    assert np.abs(dbg["lb"].cpu().numpy() - lb_h).max() < 1e-12 and np.abs(dbg["ub"].cpu().numpy() - ub_h).max() < 1e-12
    assert np.abs(dbg["targets"].cpu().numpy() - np.asarray(targets)).max() < 1e-13
    tg_d, Y0_d = T.prepare(Tg[:3])
    Yd, Yh = Y0_d.cpu().numpy(), np.asarray(Y0)
    assert np.all(np.isfinite(Yd))
    for gi in range(3):
        Gd, Gh = Yd[gi] @ Yd[gi].T, Yh[gi] @ Yh[gi].T
        assert np.abs(Gd - Gh).max() < 1e-8 * max(1.0, np.abs(Gh).max()), (gi, np.abs(Gd - Gh).max())
    # device recovery of the solved points against the host's
    q_d, pe_d, re_d = T.recover(r["x"], Tg[:3])
    q_h = prob.joint_variables(r["x"].cpu().numpy(), Tg[:3])
    assert np.abs(np.mod(q_d.cpu().numpy() - np.asarray(q_h, dtype=float) + np.pi, 2 * np.pi) - np.pi).max() < 1e-9
    pos_h, rot_h = prob.pose_errors(q_h, Tg[:3])
    assert np.allclose(pos_h, pe_d.cpu().numpy(), atol=1e-9) and np.allclose(rot_h, re_d.cpu().numpy(), atol=1e-7)
    # the whole pipeline on the device, end to end (one C call: gik_ik_batch), timed
    import time
    T.ik(Tg[:2])
    torch_cuda.cuda.synchronize()
    t0 = time.perf_counter()
    q, Yb, info = solve_batch(graph, Tg, use_limits=True)
    dt = time.perf_counter() - t0
    Ts = robot.fk_batch(q)
    pos = np.linalg.norm(Ts[:, :3, 3] - Tg[:, :3, 3], axis=1)
    assert Yb.shape == (B, N, 3) and np.mean(pos < 0.01) >= 0.75          # (table scene with 100 spheres: 93 % of 4096)
    assert np.allclose(pos, info["pos_err"], atol=1e-8)
    from parity_util import report
    report("big_scene/ur10_200_spheres", {"goals": B, "seconds_end_to_end": dt, "goals_per_second": B / dt})
    assert B / dt > 25.0, (B, dt)          # (until round 5: ~8 goals/s, 0.13 s of host prepare / recover per goal)
    # ... and the single-goal drop-in call on the same graph
    from graphik_amd.utils.lie import SE3
    q1, Y1 = solve_with_riemannian(graph, SE3.from_matrix(Tg[0]))
    assert Y1.shape == (N, 3) and q1 is not None


def test_chain_with_more_than_31_joints_through_the_device_pipeline(torch_cuda):
    """gik_pipeline_attach took at most 31 joints until round 6 ("graphs the reference takes and the library
    still refuses"); nothing in the prepare / recover kernels depends on the count.  A 40-joint 3-D chain (N = 84: the
    node-per-lane solve kernel, the workgroup prepare kernel): the first outer iterations against the oracle decision
    for decision, the device's joint angles and pose errors against the host recovery."""
    from oracle import c_oracle as co
    from graphik_amd.robots import RobotRevolute
    from graphik_amd.graphs import ProblemGraphRevolute
    from graphik_amd.solvers.riemannian_solver import BatchProblem, solve_batch
    n = 40
    rs = np.random.RandomState(0)
    names = [f"p{i}" for i in range(1, n + 1)]
    params = {"a": dict(zip(names, 0.1 + 0.2 * rs.rand(n))), "alpha": dict(zip(names, rs.choice([0, np.pi / 2, -np.pi / 2], n))),
              "d": dict(zip(names, 0.1 * rs.rand(n))), "theta": dict(zip(names, np.zeros(n))), "modified_dh": False,
              "num_joints": n}
    robot = RobotRevolute(params)
    graph = ProblemGraphRevolute(robot)
    prob = BatchProblem(graph, use_limits=True)
    assert graph.number_of_nodes() == 84 and prob.device_pipeline and prob.template.info["prepare_is_block"] == 1
    B = 16
    lb, ub = robot.limits_arrays()
    Tg = robot.fk_batch(lb + (ub - lb) * rs.rand(B, n))
    q, Y, info = solve_batch(graph, Tg)
    assert np.all(info["stop"] != 2) and np.median(info["f(x)"]) < 1e-12
    qh = np.asarray(prob.joint_variables(Y, Tg), dtype=float)
    assert np.abs(np.mod(q - qh + np.pi, 2 * np.pi) - np.pi).max() < 1e-9
    pos_h, rot_h = prob.pose_errors(q, Tg)
    assert np.allclose(pos_h, info["pos_err"], atol=1e-9) and np.allclose(rot_h, info["rot_err"], atol=1e-7)
    targets, Y0 = prob.prepare(Tg[:2])
    D, _, _ = prob.assemble(Tg[:2])
    r = prob.template.solve(Y0, targets, trace_cap=8)
    for gi in range(2):
        o = co.rtr_solve(np.asarray(Y0[gi]), D[gi], prob.omega, prob.psi_L, prob.psi_U, True, traj_cap=8)
        assert np.array_equal(r["trace"]["numit"][gi].cpu().numpy()[:5], o["traj"]["numit"][:5])
        assert (float(r["f"][gi]) < 1e-9) == (o["f(x)"] < 1e-9)


def test_host_prepare_thread_pool_is_deterministic(torch_cuda):
    """Graphs beyond the device prepare kernel (N > 32) are pre-processed on a host thread pool:
    same values whatever the number of workers."""
    from graphik_amd.solvers.riemannian_solver import BatchProblem
    from graphik_amd.utils import table_environment
    robot, graph = make_graph("ur10")
    for idx, obs in enumerate(table_environment()):
        graph.add_spherical_obstacle(f"o{idx}", obs[0], obs[1])
    prob = BatchProblem(graph, use_limits=True)
    rng = np.random.RandomState(4)
    lb, ub = robot.limits_arrays()
    Tg = robot.fk_batch(lb + (ub - lb) * rng.rand(24, robot.n))
    t1, y1 = prob.prepare(Tg, workers=1)
    t4, y4 = prob.prepare(Tg, workers=3)
    assert np.array_equal(np.asarray(t1), np.asarray(t4)) and np.array_equal(y1, y4)
    assert y1.shape == (24, graph.number_of_nodes(), 3) and np.all(np.isfinite(y1))


def test_ur10_table_drop_in(torch_cuda):
    """experiments/riemannian_example.py flow: load_ur10 + table obstacles + solve_with_riemannian."""
    from graphik_amd.solvers.riemannian_solver import solve_with_riemannian
    robot, graph = make_graph("ur10_table")
    np.random.seed(0)
    q_goal = robot.random_configuration()
    T_goal = robot.pose(q_goal, f"p{robot.n}")
    q_sol, Y = solve_with_riemannian(graph, T_goal, use_jit=False)
    assert Y.shape == (116, 3)
    assert np.linalg.norm(robot.pose(q_sol, "p6").trans - T_goal.trans) < 5e-3


# ---- the reference's ConjugateGradient option (riemannian_solver.py:51-59) on the device --------
@pytest.mark.parametrize("name,path", [(n, p) for n in ("planar10_nolimits", "planar10_limits_halfpi", "lwa4d", "ur10")
                                       for p in ("wave", "block", "block_clique")
                                       if not (p == "block_clique" and n.startswith("planar"))])   # clique path: k = 3
def test_conjugate_gradient_against_oracle(torch_cuda, name, path):
    """rcg_wave_kernel / rcg_block_kernel against the CPU twin (which tests/test_oracle_golden.py
    pins to trajectories captured from the reference's own solve()): the first 12 iterations agree
    in cost, |grad| and step size to 1e-8 and in the line search's cost evaluations exactly; the
    runs end in the same regime (planar: round-off floor by the step-size / gradient rule; 3-D,
    capped at 2000 iterations like the fixture: maxiter with a comparable cost); and against the
    fixture itself for the goals it holds.
    NOTE on what the fixture pins: pymanopt 0.2.5 is not installed here, so tests/golden/cg.npz was
    captured through our own restatement of its ConjugateGradient / LineSearchAdaptive
    (tools/ref_shims/pymanopt/solvers), driven by the reference's RiemannianSolver.  These tests pin
    "reference source + that restatement", NOT pymanopt's own code."""
    from oracle import c_oracle as co
    from graphik_amd.engine import Template
    cg = np.load(os.path.join(os.path.dirname(__file__), "golden", "cg.npz"))
    d = load_golden(name)
    use_lim = bool(int(d["use_limits"]))
    planar = name.startswith("planar")
    params = {"solver": "ConjugateGradient", "force_block_path": int(path != "wave")}
    if path == "block_clique":       # cost / gradient loops of the clique path (base + goal nodes)
        params["debug_flags"] = 64
    kw = {}
    if not planar:
        params["maxiter"] = kw["maxiter"] = int(cg["maxiter_3d"])
    T = Template.from_matrices(d["omega"], d["psi_L"], d["psi_U"], k=int(d["dim"]), use_limits=use_lim,
                               params=params)
    G = len(d["seed"])
    r = T.solve(d["Y_init"], T.targets_from_D(d["D_goal"]), trace_cap=64)
    tr = {k: v.cpu().numpy() for k, v in r["trace"].items()}
    f, its, stop = r["f"].cpu().numpy(), r["iterations"].cpu().numpy(), r["stop"].cpu().numpy()
    goals = {int(g): q for q, g in enumerate(cg[name + "__goal"])}
    m = 12
    its_o, f_o = [], []
    for g in range(G):
        o = co.cg_solve(d["Y_init"][g], d["D_goal"][g], d["omega"], d["psi_L"], d["psi_U"], use_lim,
                        traj_cap=64, **kw)
        for key, okey in (("f_before", "f"), ("gradnorm_after", "gradnorm"), ("Delta", "stepsize")):
            assert np.allclose(tr[key][g][:m], o["traj"][okey][:m], rtol=1e-8, atol=0), (g, key)
        assert np.array_equal(tr["numit"][g][:m], o["traj"]["costevals"][:m])
        if g in goals:
            q = goals[g]
            assert np.allclose(tr["f_before"][g][:m], cg[name + "__traj_f"][q][:m], rtol=1e-8, atol=0)
            assert np.array_equal(tr["numit"][g][:m], cg[name + "__traj_costevals"][q][:m])
        its_o.append(o["iterations"])
        f_o.append(o["f(x)"])
        if planar:
            assert stop[g] in (0, 3) and f[g] < 1e-13
        else:
            assert stop[g] == 1 and its[g] == o["iterations"]
    # CG with an inexact line search amplifies round-off: how long the tail at the round-off floor
    # lasts (planar; until a line search fails to move) and where a capped run stands (3-D) vary by
    # an order of magnitude per goal between any two renderings, so both are compared in distribution
    if planar:
        assert 0.5 < np.median(its) / np.median(its_o) < 2.0, (its, its_o)
    else:
        assert 0.2 < np.median(f) / np.median(f_o) < 5.0, (f, f_o)


def test_conjugate_gradient_drop_in(torch_cuda):
    """RiemannianSolver(graph, {"solver": "ConjugateGradient"}).solve(...) as the reference
    exposes it: same call, pymanopt's final_values keys, a solution that realises the goal; an
    unknown solver name raises ValueError (the reference raises a malformed tuple there)."""
    from graphik_amd.solvers.riemannian_solver import RiemannianSolver
    from graphik_amd.utils import dgp
    probot, pgraph = make_graph("planar10_nolimits")
    np.random.seed(21)
    qg = probot.random_configuration()
    Tg = probot.pose(qg, "p10")
    G = pgraph.from_pose(Tg)
    solver = RiemannianSolver(pgraph, {"solver": "ConjugateGradient"})
    lb, ub = dgp.bound_smoothing(G)
    info = solver.solve(dgp.distance_matrix_from_graph(G), dgp.adjacency_matrix_from_graph(G),
                        bounds=(lb, ub), jit=False)
    assert set(info) >= {"x", "f(x)", "time", "gradnorm", "iterations", "stepsize"}     # pymanopt's CG final_values
    assert info["f(x)"] < 1e-13 and info["stop"] in (0, 3) and 0 <= info["stepsize"] < 1.0
    qs = pgraph.joint_variables(dgp.graph_from_pos(info["x"], pgraph.node_ids), {"p10": Tg})
    assert np.linalg.norm(probot.pose(qs, "p10").trans - Tg.trans) < 1e-4
    with pytest.raises(ValueError):
        RiemannianSolver(pgraph, {"solver": "SteepestDescent"})


def test_tree_robot_solve(torch_cuda):
    """A robot with two end effectors (the tree of the reference's test_joint_variables.py:192-226)
    through RiemannianSolver.solve on the device -- the reference has no solve_with_riemannian for
    trees either -- from the captured initial points: cost at the round-off floor like the
    reference's, and joint_variables() of the solution puts BOTH end effectors on their goals and
    reproduces the captured angles."""
    from test_host_layer import tree_robot
    from graphik_amd.solvers.riemannian_solver import RiemannianSolver
    from graphik_amd.utils import dgp
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "tree5.npz"))
    robot, graph = tree_robot()
    solver = RiemannianSolver(graph)
    G = len(d["sol_f"])
    info = solver.solve(d["sol_D_goal"], d["omega"], use_limits=True, Y_init=d["sol_Y_init"])
    assert np.all(info["f(x)"] < 1e-18) and np.all(d["sol_f"] < 1e-18) and np.all(info["stop"] == 0)
    assert np.all(np.abs(info["iterations"] - d["sol_iterations"]) <= 3)
    for g in range(G):
        q = {j: d["q_goal"][g][i] for i, j in enumerate(robot.joint_ids[1:])}
        T_goal = {ee: robot.pose(q, ee) for ee in robot.end_effectors}
        q_sol = graph.joint_variables(info["x"][g], T_goal)
        for ee in robot.end_effectors:
            assert np.linalg.norm(robot.pose(q_sol, ee).trans - T_goal[ee].trans) < 1e-8
        qa = np.array([q_sol[j] for j in robot.joint_ids[1:]])
        assert np.abs(np.mod(qa - d["sol_q_sol"][g] + np.pi, 2 * np.pi) - np.pi).max() < 1e-6
    # same through bounds (device-independent host initialisation) and the no-limits cost
    Gd = graph.from_pose({ee: robot.pose({j: d["q_goal"][0][i] for i, j in enumerate(robot.joint_ids[1:])}, ee)
                          for ee in robot.end_effectors})
    r2 = solver.solve(dgp.distance_matrix_from_graph(Gd), dgp.adjacency_matrix_from_graph(Gd),
                      bounds=dgp.bound_smoothing(Gd))
    assert r2["f(x)"] < 1e-18


# ---- fixed-anchor formulation: "intended" obstacle semantics (opt-in) --------------
def _anchored_setup():
    from graphik_amd.solvers.riemannian_solver import AnchoredProblem
    robot, graph = make_graph("ur10_table")
    ap = AnchoredProblem(graph)
    ti, tj, tk, target = ap.free_terms
    Nf = len(ap.free)
    return robot, graph, ap, Nf


from parity_util import anchored_numpy as _anchored_numpy, anchored_terms as _anchored_terms  # noqa: E402


def test_anchored_kernel_known_answers(torch_cuda):
    """cost / egrad / ehess of the fixed-anchor formulation (UR10 + table_environment(): 10 free
    nodes, pinned terms to base and goal anchors, 5 x 100 obstacle hinges) against a plain fp64
    numpy evaluation: 1e-12, on points where a good share of the hinges is active."""
    robot, graph, ap, Nf = _anchored_setup()
    rng = np.random.RandomState(4)
    B = 6
    Tg = robot.fk_batch(-np.pi + 2 * np.pi * rng.rand(B, robot.n))
    ga = ap.goal_anchors(Tg)
    Y = 0.6 * rng.randn(B, Nf, 3) + np.array([0.0, 0.0, 0.9])        # around the table top
    for b in range(B):                                                  # p-nodes inside / next to spheres
        for i in np.nonzero(ap.obs_mask)[0]:
            Y[b, i] = ap.obstacles[rng.randint(len(ap.obstacles)), :3] + 0.07 * rng.randn(3)
    W = rng.randn(B, Nf, 3)
    T = ap.template
    f = T.cost(Y, ga).cpu().numpy()
    G = T.grad(Y, ga).cpu().numpy()
    H = T.hess(Y, W, ga).cpu().numpy()
    f2, G2 = T.cost_and_grad(Y, ga)
    assert np.array_equal(f2.cpu().numpy(), f) and np.array_equal(G2.cpu().numpy(), G)
    active = 0
    for b in range(B):
        fr, Gr, Hr = _anchored_numpy(ap, Nf, Y[b], W[b], ga[b])
        assert abs(f[b] - fr) <= 1e-12 * abs(fr)
        assert np.abs(G[b] - Gr).max() <= 1e-12 * np.abs(Gr).max()
        assert np.abs(H[b] - Hr).max() <= 1e-12 * np.abs(Hr).max()
        node, pos, tgt, kind = _anchored_terms(ap, ga[b])
        d = ((Y[b][node] - pos) ** 2).sum(axis=1)
        active += int(((kind == 2) & (tgt - d > 0)).sum())
    assert active > 20                                                  # the obstacle hinges are exercised
    assert np.array_equal(T.proj(Y, W).cpu().numpy(), W)                # Euclidean: proj is the identity


def test_anchored_kernel_against_reference_fixture(torch_cuda):
    """The anchored HIP kernels against REFERENCE code: tests/golden/ur10_table_intended.npz holds
    lcost / lgrad / lhess of the reference's own loops on the N = 116 graph the reference builds when
    the comparison of graph_base.py:207 is made to succeed (tools/capture_golden_intended.py), at
    points whose anchor rows sit at their true positions.  Free rows: 1e-12."""
    robot, graph, ap, Nf = _anchored_setup()
    d = load_golden("ur10_table_intended")
    assert list(d["node_ids"]) == list(graph.node_ids)
    free = np.array(ap.free)
    ga = ap.goal_anchors(d["T_goal"])[d["kat_goal"]]
    Y, W = d["kat_Y"][:, free], d["kat_W"][:, free]
    T = ap.template
    f = T.cost(Y, ga).cpu().numpy()
    G = T.grad(Y, ga).cpu().numpy()
    H = T.hess(Y, W, ga).cpu().numpy()
    assert d["kat_active_hinges"].min() >= 5
    for t in range(len(f)):
        Gr, Hr = d["kat_grad"][t][free], d["kat_hess"][t][free]
        assert abs(f[t] - d["kat_cost"][t]) <= 1e-12 * d["kat_cost"][t]
        assert np.abs(G[t] - Gr).max() <= 1e-12 * np.abs(Gr).max()
        assert np.abs(H[t] - Hr).max() <= 1e-12 * np.abs(Hr).max()


def test_anchored_trajectory_against_oracle(torch_cuda):
    """The anchored trust-region solve against its CPU twin (gik_o_rtr_solve_anchored) from the same
    start points: identical decisions and f, |grad| to 1e-8 for the first 5 outer iterations, same
    convergence class, and the solutions realise the goal."""
    from oracle import c_oracle as co
    robot, graph, ap, Nf = _anchored_setup()
    rng = np.random.RandomState(9)
    B = 12
    lbq, ubq = robot.limits_arrays()
    Tg = robot.fk_batch(lbq + (ubq - lbq) * rng.rand(B, robot.n))
    ga = ap.goal_anchors(Tg)
    Y0 = 0.5 * rng.randn(B, Nf, 3) + np.array([0.0, 0.0, 0.6])
    T = ap.template
    r = T.solve(Y0, ga, trace_cap=32)
    tr = {k: v.cpu().numpy() for k, v in r["trace"].items()}
    f, its = r["f"].cpu().numpy(), r["iterations"].cpu().numpy()
    ti, tj, tk, target = ap.free_terms
    om = np.zeros((Nf, Nf)); pL = np.zeros((Nf, Nf)); pU = np.zeros((Nf, Nf)); D = np.zeros((Nf, Nf))
    for i, j, k_, t in zip(ti, tj, tk, target):
        if k_ == 1:
            om[i, j] = om[j, i] = 1.0; D[i, j] = D[j, i] = t
        elif k_ == 2:
            pL[i, j] = pL[j, i] = t
        else:
            pU[i, j] = pU[j, i] = t
    from parity_util import assert_prefix_equal, first_divergence, report
    same_class, k_hip = 0, []
    for b in range(B):
        node, pos, tgt, kind = _anchored_terms(ap, ga[b])
        o = co.rtr_solve_anchored(Y0[b], D, om, pL, pU, node, pos, tgt, kind, traj_cap=32)
        n = min(32, int(its[b]), o["iterations"])
        hip = {k: tr[k][b] for k in tr}
        assert_prefix_equal(hip, o["traj"], min(5, n))                    # strict: every goal
        k_hip.append(first_divergence(hip, o["traj"], n))                 # ... and how far it really goes
        same_class += int((f[b] < 1e-9) == (o["f(x)"] < 1e-9))
    report("trajectory_prefix/ur10_table_intended/anchored", {"hip_leaves_oracle_at": k_hip,
           "iterations_hip": its.tolist()})
    assert same_class >= B - 2
    # the fixed-anchor problem is better conditioned than the quotient formulation (no gauge
    # freedom): the two renderings stay the same computation (decisions identical, f and |grad| to
    # 1e-8) well beyond the strict five iterations
    assert np.median(k_hip) >= 12 and min(k_hip) >= 8, k_hip      # measured: 11 .. 32, median 21


def test_anchored_pipeline(torch_cuda):
    """UR10 + table_environment() end to end through the fixed-anchor pipeline
    (gik_anchored_ik_batch: robot-graph init -> Procrustes fit to the anchors -> anchored solve ->
    recover): every problem stops by a legal rule; a converged problem (f < 1e-9) realises its goal
    pose AND keeps every p-node outside every sphere (|p - centre| >= radius - 1e-4: the hinges the
    reference means to create, graph_base.py:205-211); goals whose own configuration is collision
    free converge at a rate comparable with the reference-semantics path on the bare arm."""
    robot, graph, ap, Nf = _anchored_setup()
    rng = np.random.RandomState(2)
    B = 512
    lbq, ubq = robot.limits_arrays()
    Q = lbq + (ubq - lbq) * rng.rand(B, robot.n)
    Tg = robot.fk_batch(Q)
    r = ap.solve(Tg)
    f, stop = r["f"].cpu().numpy(), r["stop"].cpu().numpy()
    pos, rot = r["pos_err"].cpu().numpy(), r["rot_err"].cpu().numpy()
    Y = r["x"].cpu().numpy()
    assert np.all(np.isfinite(Y)) and np.all((stop == 0) | (stop == 1))
    conv = f < 1e-9
    clear = ap.clearance(Y)
    assert np.all(clear[conv] > -1e-4), clear[conv].min()
    assert np.median(pos[conv]) < 1e-3 and np.all(pos[conv] < 2e-2)
    # goals drawn from collision-free configurations are reachable without touching a sphere
    P_goal = np.stack([robot.fk_batch(Q, i)[:, :3, 3] for i in range(1, robot.n + 1)], axis=1)
    d = np.linalg.norm(P_goal[:, :, None, :] - ap.obstacles[None, None, :, :3], axis=-1) - ap.obstacles[None, None, :, 3]
    free_goal = d.min(axis=(1, 2)) > 0.02
    assert free_goal.sum() > 50
    assert conv[free_goal].mean() > 0.8, conv[free_goal].mean()
    # the anchors of the returned point matrix are the constants they were given
    g = ap.base.graph
    assert np.abs(Y[:, g.index("p0")]).max() == 0.0
    assert np.abs(Y[:, g.index(f"p{robot.n}")] - Tg[:, :3, 3]).max() < 1e-15


def test_tree_robot_solve_batch(torch_cuda):
    """solve_batch / solve_with_riemannian for a robot with two end effectors (goals [B, n_ee, 4, 4]
    in the order of robot.end_effectors, or a dict for the single call): the whole device pipeline
    (prepare with goal nodes of both end effectors and the edges between them, solve, recover by
    one walk per end effector).  Assembled matrices, bounds and recovered angles equal the captured
    reference's; random goals are reached by BOTH end effectors."""
    from test_host_layer import tree_robot
    from graphik_amd.solvers.riemannian_solver import BatchProblem, solve_batch, solve_with_riemannian
    from graphik_amd.utils import dgp
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "tree5.npz"))
    robot, graph = tree_robot()
    prob = BatchProblem(graph, use_limits=True)
    assert prob.multi_ee and prob.device_pipeline        # prepare / recover kernels take several end effectors
    G = len(d["sol_f"])
    # device goal assembly + bound smoothing against the captured reference (one hop)
    dbg = prob.template.prepare_debug(d["T_goal"][:G])
    assert np.abs(dbg["lb"].cpu().numpy() - d["sol_lb"]).max() < 1e-12
    assert np.abs(dbg["ub"].cpu().numpy() - d["sol_ub"]).max() < 1e-12
    tg_ref = prob.template.targets_from_D(d["sol_D_goal"])
    assert np.abs(dbg["targets"].cpu().numpy() - tg_ref).max() < 1e-13
    # device joint recovery of the captured solutions against the captured angles
    q_d, pe, re = prob.template.recover(d["sol_Y_sol"], d["T_goal"][:G])
    dq = np.abs(np.mod(q_d.cpu().numpy() - d["sol_q_sol"] + np.pi, 2 * np.pi) - np.pi)
    assert dq.max() < 1e-9 and pe.max().item() < 1e-9
    D, lo, up = prob.assemble(d["T_goal"][:G])
    assert np.abs(D - d["sol_D_goal"]).max() < 1e-13
    lb, ub = dgp.floyd_warshall_bounds(lo, up)
    assert np.abs(lb - d["sol_lb"]).max() < 1e-12 and np.abs(ub - d["sol_ub"]).max() < 1e-12
    rng = np.random.RandomState(6)
    B = 64
    lbq, ubq = robot.limits_arrays()
    Q = lbq + (ubq - lbq) * rng.rand(B, robot.n)
    Tg = np.stack([robot.fk_batch(Q, int(e[1:])) for e in robot.end_effectors], axis=1)      # [B, 2, 4, 4]
    q, Y, info = solve_batch(graph, Tg)
    assert Y.shape == (B, graph.number_of_nodes(), 3) and q.shape == (B, robot.n)
    assert np.mean(info["pos_err"] < 1e-6) > 0.9 and np.all(info["stop"] != 2)
    q1, Y1 = solve_with_riemannian(graph, {e: Tg[0, i] for i, e in enumerate(robot.end_effectors)})
    assert set(q1) == {f"p{i}" for i in range(1, 6)}
    for i, e in enumerate(robot.end_effectors):
        assert np.linalg.norm(robot.pose(q1, e).trans - Tg[0, i][:3, 3]) < 1e-6


def test_anchored_without_obstacles_agrees_with_reference_semantics(torch_cuda):
    """With no obstacle the fixed-anchor formulation and the reference's quotient formulation
    describe the same feasibility problem (the anchors' mutual distances are constants either way),
    so on a bare UR10 both pipelines must reach the goal poses at the same rate, and -- UR10
    solutions are isolated -- mostly on the same IK branch.  Also: B = 0, and a graph whose
    obstacles were cleared."""
    from graphik_amd.solvers.riemannian_solver import AnchoredProblem, solve_batch
    robot, graph = make_graph("ur10")
    ap = AnchoredProblem(graph)
    assert len(ap.obstacles) == 0 and len(ap.free) == 10
    rng = np.random.RandomState(13)
    B = 512
    lbq, ubq = robot.limits_arrays()
    Tg = robot.fk_batch(lbq + (ubq - lbq) * rng.rand(B, robot.n))
    r = ap.solve(Tg)
    pos_a, rot_a = r["pos_err"].cpu().numpy(), r["rot_err"].cpu().numpy()
    q_a = r["q"].cpu().numpy()
    q_s, _, info = solve_batch(graph, Tg)
    ok_a = (pos_a < 0.01) & (rot_a < 0.01)
    ok_s = (info["pos_err"] < 0.01) & (info["rot_err"] < 0.01)
    assert abs(ok_a.mean() - ok_s.mean()) < 0.05 and ok_a.mean() > 0.85, (ok_a.mean(), ok_s.mean())
    both = ok_a & ok_s
    assert np.median(pos_a[both]) < 2 * np.median(info["pos_err"][both]) + 1e-5
    dq = np.abs(np.mod(q_a - q_s + np.pi, 2 * np.pi) - np.pi).max(axis=1)
    assert np.mean(dq[both] < 1e-2) > 0.5          # same start point (robot-graph MDS), mostly the same branch
    assert ap.solve(Tg[:0])["x"].shape[0] == 0
    robot2, graph2 = make_graph("ur10_table")
    graph2.clear_obstacles()
    assert len(AnchoredProblem(graph2).obstacles) == 0


def test_anchored_near_lists_are_bit_identical(torch_cuda):
    """The anchored kernel walks only the obstacles near a node while the node stays within the
    clearance of all others measured at its last full walk (conservative bound: a skipped hinge is
    provably inactive and contributes exactly zero).  Results must equal, bit for bit, those of
    walking all 100 obstacles in every cost / gradient evaluation (debug_flags = 128)."""
    from graphik_amd.solvers.riemannian_solver import AnchoredProblem
    robot, graph = make_graph("ur10_table")
    rng = np.random.RandomState(31)
    B = 384
    lbq, ubq = robot.limits_arrays()
    Tg = robot.fk_batch(lbq + (ubq - lbq) * rng.rand(B, robot.n))
    outs = []
    for flags in (0, 128):
        ap = AnchoredProblem(graph, params={"debug_flags": flags})
        r = ap.solve(Tg)
        outs.append({k: r[k].cpu().numpy() for k in ("x", "f", "gradnorm", "iterations", "inner_total", "stop", "q")})
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), k
    assert (outs[0]["f"] < 1e-9).mean() > 0.7


@pytest.mark.parametrize("which", ["y5", "bin2"])
def test_planar_tree_solve_batch(torch_cuda, which):
    """Planar TREES (graph_planar.py:50-88; several end effectors, possibly sharing their parent): since round 6 the
    whole pipeline on the device (goal assembly + bound smoothing + initial point: prep_wave_kernel with an inert goal
    slot for a shared parent, or prep_quad_kernel; joint_variables + FK per end effector: recover_kernel's path walk),
    checked against the host layer -- every end effector of the recovered configuration reaches its goal pose; and
    the drop-in call with a {end effector: pose} dict."""
    from conftest import planar_tree
    from graphik_amd.solvers.riemannian_solver import BatchProblem, solve_batch, solve_with_riemannian
    from graphik_amd.utils import dgp
    robot, graph = planar_tree(which)
    prob = BatchProblem(graph, use_limits=True)
    assert prob.multi_ee and prob.device_pipeline
    rng = np.random.RandomState(6)
    lb, ub = robot.limits_arrays()
    B = 64
    Q = lb + (ub - lb) * rng.rand(B, robot.n)
    Tg = np.stack([[robot.pose(robot.array_to_q(q), ee).as_matrix() for ee in robot.end_effectors] for q in Q])
    q, Y, info = solve_batch(graph, Tg, use_limits=True)
    assert Y.shape == (B, graph.number_of_nodes(), 2) and q.shape == (B, robot.n)
    assert np.all(np.isfinite(Y)) and np.all(info["stop"] != 2)
    ok = (info["pos_err"] < 0.01) & (info["rot_err"] < 0.01)     # worst end effector per goal
    assert ok.mean() > 0.9, ok.mean()
    assert np.median(info["pos_err"]) < 1e-4
    # device goal assembly + bound smoothing against the host's, the initial point up to its Gram matrix
    dbg = prob.template.prepare_debug(Tg[:8])
    D_h, lo, up = prob.assemble(Tg[:8])
    lb_h, ub_h = dgp.floyd_warshall_bounds(lo, up)
    assert np.abs(dbg["lb"].cpu().numpy() - lb_h).max() < 1e-12 and np.abs(dbg["ub"].cpu().numpy() - ub_h).max() < 1e-12
    assert np.abs(dbg["targets"].cpu().numpy() - prob.targets_from_D(D_h)).max() < 1e-13
    _, Y0_h = prob.prepare(Tg[:8])
    Y0_d = dbg["Y_init"].cpu().numpy()
    for gi in range(8):
        assert np.abs(Y0_d[gi] @ Y0_d[gi].T - Y0_h[gi] @ Y0_h[gi].T).max() < 1e-8
    # device joint recovery and pose errors (the worst end effector's) against the host's on the solved points
    q_h = np.asarray(prob.joint_variables(Y, Tg), dtype=float)
    assert np.abs(np.mod(q - q_h + np.pi, 2 * np.pi) - np.pi).max() < 1e-9
    pos_h, rot_h = prob.pose_errors(q, Tg)
    assert np.allclose(pos_h, info["pos_err"], atol=1e-9) and np.allclose(rot_h, info["rot_err"], atol=1e-7)
    T_goal = {ee: robot.pose(robot.array_to_q(Q[0]), ee) for ee in robot.end_effectors}
    q_sol, Y1 = solve_with_riemannian(graph, T_goal)
    for ee in robot.end_effectors:
        assert np.linalg.norm(robot.pose(q_sol, ee).trans - T_goal[ee].trans) < 1e-3
    # against the reference's own solves of the same trees (RiemannianSolver.solve from its Y_init,
    # tests/golden/planar_tree.npz): same iteration count, same point, same angles
    from graphik_amd.solvers.riemannian_solver import RiemannianSolver
    d = load_golden("planar_tree")
    g = lambda k: d[f"{which}_{k}"]   # noqa: E731
    solver = RiemannianSolver(graph)
    for s_ in range(len(g("sol_f"))):
        info = solver.solve(g("D_goal")[s_], g("omega"), use_limits=True, Y_init=g("sol_Y_init")[s_], jit=False)
        assert info["iterations"] == int(g("sol_iterations")[s_]) and info["f(x)"] < 1e-20
        assert np.abs(info["x"] - g("sol_Y_sol")[s_]).max() < 1e-9
        q = graph.joint_variables(info["x"])
        dq = np.array([q[j] for j in robot.joint_ids[1:]]) - g("sol_q_sol")[s_]
        assert np.abs(np.mod(dq + np.pi, 2 * np.pi) - np.pi).max() < 1e-8


def test_planar_tree_with_eight_end_effectors(torch_cuda):
    """More than four end effectors (the device pipeline's limit until round 6): the balanced binary tree of height 3 --
    14 joints, 8 end effectors, every pair of them sharing its parent -- through the device pipeline against the host
    layer: bounds, targets, recovered angles, worst-end-effector pose errors."""
    from graphik_amd.robots import RobotPlanar
    from graphik_amd.graphs import ProblemGraphPlanar
    from graphik_amd.utils import dgp, list_to_variable_dict
    from graphik_amd.solvers.riemannian_solver import BatchProblem, solve_batch
    n = 14
    parents = {f"p{i}": [f"p{2 * i + 1}", f"p{2 * i + 2}"] for i in range(7)}
    lim = np.full(n, np.pi / 2)
    robot = RobotPlanar({"link_lengths": list_to_variable_dict(np.ones(n)), "num_joints": n, "parents": parents,
                         "joint_limits_upper": list_to_variable_dict(lim), "joint_limits_lower": list_to_variable_dict(-lim)})
    graph = ProblemGraphPlanar(robot)
    assert len(robot.end_effectors) == 8
    prob = BatchProblem(graph, use_limits=True)
    assert prob.device_pipeline
    rng = np.random.RandomState(2)
    B = 32
    Q = -lim + 2 * lim * rng.rand(B, n)
    Tg = np.stack([[robot.pose(robot.array_to_q(q), ee).as_matrix() for ee in robot.end_effectors] for q in Q])
    dbg = prob.template.prepare_debug(Tg[:4])
    D_h, lo, up = prob.assemble(Tg[:4])
    lb_h, ub_h = dgp.floyd_warshall_bounds(lo, up)
    assert np.abs(dbg["lb"].cpu().numpy() - lb_h).max() < 1e-12 and np.abs(dbg["ub"].cpu().numpy() - ub_h).max() < 1e-12
    assert np.abs(dbg["targets"].cpu().numpy() - prob.targets_from_D(D_h)).max() < 1e-13
    q, Y, info = solve_batch(graph, Tg, use_limits=True)
    assert np.all(info["stop"] != 2)
    q_h = np.asarray(prob.joint_variables(Y, Tg), dtype=float)
    assert np.abs(np.mod(q - q_h + np.pi, 2 * np.pi) - np.pi).max() < 1e-9
    pos_h, rot_h = prob.pose_errors(q, Tg)
    assert np.allclose(pos_h, info["pos_err"], atol=1e-9) and np.allclose(rot_h, info["rot_err"], atol=1e-7)
    assert np.mean((info["pos_err"] < 0.01) & (info["rot_err"] < 0.01)) > 0.8


def test_stream_capture_is_refused(torch_cuda):
    """A batch call on a capturing stream is refused with a message BEFORE anything that is illegal under capture
    happens (event queries / waits, workspace growth, the counter reset a replay would share: ADVICE r5), the capture
    itself survives, and the handle works afterwards.  (The C entry point is called directly: nothing of torch's may
    run between capture_begin and capture_end either.)"""
    import ctypes as C
    from graphik_amd import _ffi
    from graphik_amd.engine import _alloc_stats
    d = load_golden("lwa4d")
    T = _template(d)
    B = 4
    dev = T.device
    tg = torch_cuda.as_tensor(np.ascontiguousarray(T.targets_from_D(d["D_goal"][:B])), device=dev)
    Y0 = torch_cuda.as_tensor(d["Y_init"][:B], device=dev).contiguous()       # [B, N, 3]
    ref = T.solve(Y0, tg)["x"].clone()
    out, stats = torch_cuda.empty_like(Y0), _alloc_stats(B, dev)
    torch_cuda.cuda.synchronize()
    s = torch_cuda.cuda.Stream()
    g = torch_cuda.cuda.CUDAGraph()
    with torch_cuda.cuda.stream(s):
        g.capture_begin()
        try:
            rc = T.lib.gik_solve_batch(T._h, Y0.data_ptr(), tg.data_ptr(), B, out.data_ptr(), stats.data_ptr(), None,
                                       C.c_void_p(s.cuda_stream))
            msg = T.lib.gik_last_error().decode()
        finally:
            g.capture_end()
    assert rc != 0 and "capturing" in msg, (rc, msg)
    torch_cuda.cuda.synchronize()
    assert torch_cuda.equal(T.solve(Y0, tg)["x"], ref)


def test_c_abi_error_behaviour(torch_cuda):
    """Every entry point refuses bad arguments with a non-zero return and a message in
    gik_last_error() -- no crash, no silent success (include/graphik_amd.h; the Python layer turns
    these into GikError)."""
    import ctypes as C
    from graphik_amd import _ffi
    L = _ffi.lib()
    ti = np.array([0, 0, 1], dtype=np.int32)           # terms sorted by (i, j, kind)
    tj = np.array([1, 2, 2], dtype=np.int32)
    tk = np.full(3, _ffi.TERM_EQ, dtype=np.int32)

    def desc(**kw):
        d = _ffi.TemplateDesc()
        L.gik_default_params(C.byref(d))
        d.N, d.k, d.n_terms = 3, 3, 3
        d.term_i = ti.ctypes.data_as(C.POINTER(C.c_int32))
        d.term_j = tj.ctypes.data_as(C.POINTER(C.c_int32))
        d.term_kind = tk.ctypes.data_as(C.POINTER(C.c_int32))
        for k_, v in kw.items():
            setattr(d, k_, v)
        return d

    def refused(rc, fragment):
        msg = L.gik_last_error().decode()
        assert rc != 0 and fragment in msg, (rc, msg)

    h = C.c_void_p()
    refused(L.gik_template_create(None, C.byref(h)), "null")
    refused(L.gik_template_create(C.byref(desc(abi_version=_ffi.ABI_VERSION - 1)), C.byref(h)), "ABI")
    refused(L.gik_template_create(C.byref(desc(k=4)), C.byref(h)), "k must be")
    refused(L.gik_template_create(C.byref(desc(N=1)), C.byref(h)), "N must be")
    refused(L.gik_template_create(C.byref(desc(N=256)), C.byref(h)), "N must be")      # (round 5: up to 255)
    refused(L.gik_template_create(C.byref(desc(N=129, k=2)), C.byref(h)), "128 nodes")  # beyond 128: 3-D TrustRegions only
    refused(L.gik_template_create(C.byref(desc(hessian_form=3)), C.byref(h)), "hessian_form")
    # an explicit per-edge product where no such kernel exists (ConjugateGradient: it takes no Hessian products) is refused,
    # not ignored ...
    refused(L.gik_template_create(C.byref(desc(hessian_form=_ffi.HESS_PER_EDGE, solver=_ffi.SOLVER_CONJUGATE_GRADIENT)), C.byref(h)), "GIK_HESS_PER_EDGE")
    # ... while the default (GIK_HESS_AUTO) resolves to what the template's kernel does, and says so
    info = _ffi.TemplateInfo()
    for kw, form in (({}, _ffi.HESS_PER_EDGE), ({"theta": 0.5}, _ffi.HESS_PER_EDGE), ({"theta": 0.5, "hessian_form": _ffi.HESS_COLUMN}, _ffi.HESS_COLUMN),
                     ({"solver": _ffi.SOLVER_CONJUGATE_GRADIENT}, _ffi.HESS_COLUMN),
                     ({"hessian_form": _ffi.HESS_COLUMN}, _ffi.HESS_COLUMN), ({"force_block_path": 1}, _ffi.HESS_PER_EDGE)):
        assert L.gik_template_create(C.byref(desc(**kw)), C.byref(h)) == 0, L.gik_last_error()
        assert L.gik_template_get_info(h, C.byref(info)) == 0 and info.hessian_form == form, (kw, info.hessian_form)
        L.gik_template_destroy(h)
    refused(L.gik_template_create(C.byref(desc(n_terms=0)), C.byref(h)), "n_terms")
    refused(L.gik_template_create(C.byref(desc(solver=7)), C.byref(h)), "solver")
    refused(L.gik_template_create(C.byref(desc(clique_closed_form=9)), C.byref(h)), "clique_closed_form")
    bad_j = np.array([1, 2, 1], dtype=np.int32)          # a term from a node to itself
    d = desc()
    d.term_j = bad_j.ctypes.data_as(C.POINTER(C.c_int32))
    refused(L.gik_template_create(C.byref(d), C.byref(h)), "bad term indices")
    bad_k = np.array([1, 5, 1], dtype=np.int32)
    d = desc()
    d.term_kind = bad_k.ctypes.data_as(C.POINTER(C.c_int32))
    refused(L.gik_template_create(C.byref(d), C.byref(h)), "bad term kind")
    # a good handle, bad calls on it
    assert L.gik_template_create(C.byref(desc()), C.byref(h)) == 0
    Y = torch_cuda.zeros(2, 3, 3, dtype=torch_cuda.float64, device="cuda")
    tg = torch_cuda.ones(2, 3, dtype=torch_cuda.float64, device="cuda")
    st = torch_cuda.zeros(2, _ffi.STATS_BYTES // 8, dtype=torch_cuda.float64, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    refused(L.gik_solve_batch(h, None, p(tg), 2, p(Y), p(st), None, None), "null buffer")
    refused(L.gik_solve_batch(h, p(Y), p(tg), -1, p(Y), p(st), None, None), "bad argument")
    refused(L.gik_solve_batch(None, p(Y), p(tg), 2, p(Y), p(st), None, None), "bad argument")
    refused(L.gik_prepare_batch(h, p(Y), 2, p(tg), p(Y), None, None), "no pipeline attached")
    refused(L.gik_recover_batch(h, p(Y), p(Y), 2, p(Y), p(Y), p(Y), None), "no pipeline attached")
    refused(L.gik_cost(h, None, p(tg), 2, p(st), None), "null")
    refused(L.gik_template_get_info(h, None), "null")
    refused(L.gik_pipeline_attach(h, None), "null")
    assert L.gik_solve_batch(h, p(Y), p(tg), 0, p(Y), p(st), None, None) == 0      # an empty batch is not an error
    L.gik_template_destroy(h)
    L.gik_template_destroy(None)                                                    # and destroying nothing is harmless
    # the Python layer: unknown parameters and solvers are refused before the library is called
    d0 = load_golden("lwa4d")
    with pytest.raises(KeyError):
        _template(d0, no_such_parameter=1)
    from graphik_amd.engine import Template
    with pytest.raises(ValueError):
        Template.from_matrices(d0["omega"], d0["psi_L"], d0["psi_U"], k=3, use_limits=True, params={"solver": "Newton"})
    with pytest.raises(_ffi.GikError):
        Template.from_matrices(d0["omega"], k=5, use_limits=False)
