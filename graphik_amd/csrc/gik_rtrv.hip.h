// graphik_amd/csrc/gik_rtrv.hip.h -- the Riemannian trust-region driver over a SMALL VECTOR PER
// THREAD: rtr_solve_one (gik_rtr.hip.h, one entry of every tangent vector per thread) generalised
// to NE entries per thread, for contexts in which a thread owns whole graph nodes (NptCtx,
// gik_npt.hip.h: two nodes x three components per lane, one wavefront per problem).
//
// rtr_solve_vec<>() is TrustRegions.solve (graphik/solvers/trust_region.py:112-434) with
// _truncated_conjugate_gradient (:436-599) inlined -- the same single-reduction formulation, the
// same order of events and the same checkpoint resume as rtr_solve_one's k = 3 path (see the comments
// there and docs/NOTEBOOK.md 4.1); only the granularity differs: every per-thread product
// becomes a short dot product over the thread's entries before the reduction, every axpy a short
// loop.  Context interface:
//     cost(x) -> f, commit(x, g), proj_setup(x), ehess(delta, H),
//     vert_dots(Z, r0, r1, r2) / vert_coords(r, u) / vert_apply(u, Z, out): the horizontal projector
//       Z - Q Q^T Z through the generators of the vertical space (u = Q^T Z, Q orthonormal),
//     ck_put(i, v) / ck_get(i, e), sum_n<NV>(v), sum1(x), lead().
#pragma once

#include <hip/hip_runtime.h>

#include "gik_rtr.hip.h"

namespace gik {

template <int NE>
__device__ inline double vdot(const double (&a)[NE], const double (&b)[NE]) {
  double s = a[0] * b[0];
#pragma unroll
  for (int e = 1; e < NE; ++e) s = fma(a[e], b[e], s);
  return s;
}

// ||g||_F together with the components of g along the orthonormal vertical basis, one reduction
template <typename Ctx>
__device__ inline double grad_norm_and_rho_vec(Ctx &cx, const double (&g)[Ctx::NE], double (&rho0)[3]) {
  double v[4];
  v[0] = vdot(g, g);
  cx.vert_dots(g, v[1], v[2], v[3]);
  cx.template sum_n<4>(v);
  const double r[3] = {v[1], v[2], v[3]};
  cx.vert_coords(r, rho0);
  return sqrt(v[0]);
}

// THETA_ONE / SLICE as in rtr_solve_one.  k = 3 only (the literal planar projector of
// fixed_rank_psd_sym.py:107-110 is not orthogonal; planar graphs fit the wavefront kernel anyway).
template <bool THETA_ONE, bool SLICE, typename Ctx>
__device__ inline void rtr_solve_vec(Ctx &cx, const Params &p, const gik_trace &trace, int has_trace, int dbg,
                                     double *dbg_buf, int b, double (&x)[Ctx::NE], RtrOut &out,
                                     const RtrResume &rs, int slice_its) {
  constexpr int NE = Ctx::NE;
  constexpr int K = 3;
  const double Delta_bar = 10.0 + K;  // typicaldist (fixed_rank_psd_sym.py:71-73)
  const bool lead = cx.lead();
  double Delta = (SLICE && rs.resumed) ? rs.Delta : Delta_bar / 8.0;   // trust_region.py:134-135,164
  double fx = cx.cost(x);                 // :159
  double g[NE];
  cx.commit(x, g);                        // :160
  cx.proj_setup(x);
  double rho0[3];
  double norm_grad = grad_norm_and_rho_vec(cx, g, rho0);   // :161
  int kiter = SLICE ? rs.kiter : 0, inner_total = SLICE ? rs.inner_total : 0,
      inner_exec = SLICE ? rs.inner_exec : 0, n_accept = SLICE ? rs.n_accept : 0, stop = 1;
  int slice_count = 0;
  int paused = PAUSE_NONE;
  // Retrace (rtr_solve_one): after a rejected step the next tCG solve repeats the previous one up to
  // the smaller radius; it is resumed from a checkpoint instead, bit for bit (dbg & 16 disables).
  const bool retrace_on = !(dbg & 16);
  bool prev_rejected = false, ck_set = false, ck_neg = false;
  int ck_j = 0;
  double ck_T = 0.0, ck_e_Pe = 0.0, ck_e_Pd2 = 0.0, ck_d_Pd = 0.0;
  double last_Delta2 = 0.0, last_e_Pe = 0.0;
  double eta[NE], Heta[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) eta[e] = Heta[e] = 0.0;
  int stop_tCG = TCG_MAX_INNER_ITER, j = 0;
  const bool prof = (dbg & 8) && b == 0 && dbg_buf;   // cycle counters (developer aid)
  long long prof_tcg = 0;
  const long long prof_t0 = prof ? (long long)__builtin_readcyclecounter() : 0;
  bool bad = UNI(!(fx == fx) || !(norm_grad == norm_grad));
  if (dbg & 2) bad = true;

  while (!bad) {
    // -------------- _truncated_conjugate_gradient (trust_region.py:436-599) -------------
    const double Delta2 = Delta * Delta;
    bool reuse = false;
    if (retrace_on && prev_rejected) {
      if (Delta2 == last_Delta2) {
        reuse = true;                                  // same radius: the identical solve
      } else if (ck_set && Delta2 == ck_T) {           // the rerun stops at the checkpoint
        const double tau = boundary_tau(ck_e_Pd2, ck_d_Pd, Delta2, ck_e_Pe);       // :514
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          eta[e] = fma(tau, cx.ck_get(2, e), cx.ck_get(0, e));                     // :516
          Heta[e] = fma(tau, cx.ck_get(3, e), cx.ck_get(1, e));                    // :521
        }
        stop_tCG = ck_neg ? TCG_NEGATIVE_CURVATURE : TCG_EXCEEDED_TR;
        j = ck_j;
        ck_set = false;
        reuse = true;
      } else if (stop_tCG != TCG_NEGATIVE_CURVATURE && stop_tCG != TCG_EXCEEDED_TR && last_e_Pe < Delta2) {
        reuse = true;                                  // never met the smaller radius either
      }
    }
    last_Delta2 = Delta2;
    const long long prof_t1 = prof ? (long long)__builtin_readcyclecounter() : 0;
    int executed = 0;
    if (!reuse) {
      double eta_l[NE], Heta_l[NE];              // :444-445
#pragma unroll
      for (int e = 0; e < NE; ++e) eta_l[e] = Heta_l[e] = 0.0;
      stop_tCG = TCG_MAX_INNER_ITER;             // :491
      int extra = 0;
      double r[NE], w[NE], delta[NE];
      const double r0_r0 = norm_grad * norm_grad;   // :455 (r = grad: same sum as ||grad||^2)
      const double nr0_theta = (THETA_ONE || p.theta == 1.0) ? norm_grad : pow(norm_grad, p.theta);
      const double target = norm_grad * fmin(nr0_theta, p.kappa);  // rhs of :572
      const double target2 = target * target;
      const double target2_hi = target2 * (1.0 + 1e-9), target2_lo = target2 * (1.0 - 1e-9);   // (see rtr_solve_one)
      const double Tq = 0.0625 * Delta2;         // radius the plain path tests against until the checkpoint
      double T_cur = Tq;
      ck_set = false;
      // w = -(horizontal part of r): the search direction is built from it (rtr_solve_one)
      {
        double mg[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          r[e] = g[e];                           // :448
          mg[e] = g[e];
        }
        cx.vert_apply(rho0, mg, w);              // g - Q Q^T g
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          w[e] = -w[e];
          delta[e] = w[e];                       // :469 (horizontal part)
        }
      }
      double e_Pd2 = 0.0, d_Pd = r0_r0;          // :464-471 (precon = identity)
      double model_prev = __builtin_inf();       // model value before the last step (:485: 0)
      // eta, Heta and <eta, eta> live in two register sets; a step reads one and writes the other,
      // which until then holds the PREVIOUS eta / Heta -- what the deferred model test rolls back to
      double ea[NE], ha[NE], eb[NE], hb[NE], pa = 0.0, pb = 0.0;
#pragma unroll
      for (int e = 0; e < NE; ++e) ea[e] = ha[e] = eb[e] = hb[e] = 0.0;
      double e_Pe_end = 0.0;                     // <eta, eta> of the last step that passed the radius test
      __builtin_amdgcn_s_waitcnt(0);
      // One tCG iteration (:495-597).  (ec, hc, pc): current eta, Heta, <eta,eta>; (en, hn): the
      // previous eta, Heta on entry, the new ones on a plain return; pn: the new <eta,eta>.
      auto step = [&](const double (&ec)[NE], const double (&hc)[NE], const double pc, double (&en)[NE],
                      double (&hn)[NE], double &pn) __attribute__((always_inline)) -> bool {
        double H[NE];
#ifdef GIK_NPT_PROF
        long long pt = cx.pf_now();
#endif
        cx.ehess(delta, H);                      // :497
#ifdef GIK_NPT_PROF
        pt = cx.pf_now();
#endif
        double v[8];
        cx.vert_dots(H, v[0], v[1], v[2]);
        v[3] = vdot(delta, H);
        v[4] = vdot(w, H);
        v[5] = vdot(H, H);
        {
          double s = ec[0] * fma(0.5, hc[0], g[0]);
#pragma unroll
          for (int e = 1; e < NE; ++e) s = fma(ec[e], fma(0.5, hc[e], g[e]), s);
          v[6] = s;
        }
        v[7] = vdot(r, r);
        cx.template sum_n<8>(v);
#ifdef GIK_NPT_PROF
        { const long long t_ = cx.pf_now(); cx.pf[5] += t_ - pt; pt = t_; }      // inner products + reduction
#endif
        double Hdelta[NE], uv[3];
        {
          const double rv[3] = {v[0], v[1], v[2]};
          cx.vert_coords(rv, uv);                // u = Q^T H
        }
        cx.vert_apply(uv, H, Hdelta);            // H - Q u
        const double d_Hd = v[3];                // :500
        const double Hd_Hd = fma(-uv[2], uv[2], fma(-uv[1], uv[1], fma(-uv[0], uv[0], v[5])));
        const double model_value = v[6];         // :551 evaluated at the current eta
        const double r_r = v[7];                 // :564 exact
        const double rho = frcp1(d_Hd);
        const double alpha = r_r * rho;          // :503
        const double e_Pe_new = fma(alpha, fma(alpha, d_Pd, e_Pd2), pc);             // :506
        const double beta_p = fma(fma(alpha, Hd_Hd, -(v[4] + v[4])), rho, 1.0);      // :592 predicted
        double new_r_r = beta_p * r_r;                                               // :564 predicted
        const bool plain = (model_value < model_prev) & (d_Hd > 0.0) & (e_Pe_new < T_cur) &
                           (beta_p >= 1e-3) & !((j >= p.mininner) & (new_r_r <= target2_hi)) &
                           (j + 1 < p.maxinner);
        double beta = beta_p;
        double rr_test = new_r_r;      // what the residual test sees (the recurrences keep the prediction)
        if (__builtin_expect(UNI(!plain), 0)) {   // any exit, a NaN, or the accuracy guard
          if (!(d_Hd == d_Hd) || !(new_r_r == new_r_r) || !(model_value == model_value)) {
            bad = true;
            return true;
          }
          if (model_value >= model_prev) {                    // :552 of step j-1
#pragma unroll
            for (int e = 0; e < NE; ++e) {
              eta_l[e] = en[e];
              Heta_l[e] = hn[e];
            }
            e_Pe_end = pc;
            stop_tCG = TCG_MODEL_INCREASED;
            extra = 1;
            j = j - 1;
            return true;
          }
          if (!ck_set && (d_Hd <= 0.0 || e_Pe_new >= Tq)) {   // first meeting with radius / 4
            cx.ck_put(0, ec);
            cx.ck_put(1, hc);
            cx.ck_put(2, delta);
            cx.ck_put(3, Hdelta);
            ck_e_Pe = pc;
            ck_e_Pd2 = e_Pd2;
            ck_d_Pd = d_Pd;
            ck_j = j;
            ck_neg = d_Hd <= 0.0;
            ck_T = Tq;
            ck_set = true;
            T_cur = Delta2;
          }
          if (d_Hd <= 0.0 || e_Pe_new >= Delta2) {           // :509
            const double tau = boundary_tau(e_Pd2, d_Pd, Delta2, pc);             // :514
#pragma unroll
            for (int e = 0; e < NE; ++e) {
              eta_l[e] = fma(tau, delta[e], ec[e]);           // :516
              Heta_l[e] = fma(tau, Hdelta[e], hc[e]);         // :521
            }
            e_Pe_end = pc;
            stop_tCG = (d_Hd <= 0.0) ? TCG_NEGATIVE_CURVATURE : TCG_EXCEEDED_TR;  // :531-534
            return true;
          }
          double alpha_c = alpha;
          asm volatile("" : "+v"(alpha_c));
          if (beta_p < 1e-3) {
            double s = 0.0;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
              const double nr = fma(alpha_c, Hdelta[e], r[e]);  // :561
              s = fma(nr, nr, s);
            }
            new_r_r = cx.sum1(s);
            beta = new_r_r / r_r;
            rr_test = new_r_r;
          } else if (j >= p.mininner && new_r_r >= target2_lo && new_r_r <= target2_hi) {
            double s = 0.0;      // a near tie is decided on the sum itself (:560-572)
#pragma unroll
            for (int e = 0; e < NE; ++e) {
              const double nr = fma(alpha_c, Hdelta[e], r[e]);
              s = fma(nr, nr, s);
            }
            rr_test = cx.sum1(s);
          }
          const bool at_target = j >= p.mininner && rr_test <= target2;           // :572
          const bool at_maxinner = j + 1 >= p.maxinner;                           // :495
          if (at_target || at_maxinner) {
            e_Pe_end = e_Pe_new;   // this step passed the radius test (what a rerun has to pass again)
            // the reference tests the model of this step first (:552)
            double ne[NE], nh[NE], s = 0.0;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
              ne[e] = fma(alpha_c, delta[e], ec[e]);          // :538
              nh[e] = fma(alpha_c, Hdelta[e], hc[e]);         // :542
              s = fma(ne[e], fma(0.5, nh[e], g[e]), s);
            }
            const double model_new = cx.sum1(s);
            if (model_new >= model_value) {
#pragma unroll
              for (int e = 0; e < NE; ++e) {
                eta_l[e] = ec[e];
                Heta_l[e] = hc[e];
              }
              stop_tCG = TCG_MODEL_INCREASED;
            } else {
#pragma unroll
              for (int e = 0; e < NE; ++e) {
                eta_l[e] = ne[e];
                Heta_l[e] = nh[e];
              }
              if (at_target)
                stop_tCG = (p.kappa < nr0_theta) ? TCG_REACHED_TARGET_LINEAR : TCG_REACHED_TARGET_SUPERLINEAR;
            }
            if (!at_target) j = p.maxinner;
            return true;
          }
        }
        ++j;
        pn = e_Pe_new;                                    // :537
        model_prev = model_value;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          en[e] = fma(alpha, delta[e], ec[e]);            // :538, :556-558 (over the previous eta)
          hn[e] = fma(alpha, Hdelta[e], hc[e]);           // :542
          r[e] = fma(alpha, Hdelta[e], r[e]);             // :561
          w[e] = fma(-alpha, Hdelta[e], w[e]);
          delta[e] = fma(beta, delta[e], w[e]);           // :593
        }
        e_Pd2 = beta * fma(alpha + alpha, d_Pd, e_Pd2);   // :596 (carried as 2 <eta, delta>)
        d_Pd = fma(beta * beta, d_Pd, new_r_r);           // :597
#ifdef GIK_NPT_PROF
        { const long long t_ = cx.pf_now(); cx.pf[6] += t_ - pt; }               // scalar step + vector updates
#endif
        return false;
      };
      j = 0;
      if (p.maxinner > 0)
        for (;;) {                               // :495
          if (step(ea, ha, pa, eb, hb, pb)) break;
          if (step(eb, hb, pb, ea, ha, pa)) break;
        }
      last_e_Pe = e_Pe_end;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        eta[e] = eta_l[e];
        Heta[e] = Heta_l[e];
      }
      executed = (j >= p.maxinner ? p.maxinner : j + 1) + extra;
    }
    if (prof) prof_tcg += (long long)__builtin_readcyclecounter() - prof_t1;
    if (bad) break;
    if (j >= p.maxinner) j = p.maxinner - 1;  // Python leaves j at the last index
    inner_total += j + 1;
    inner_exec += executed;

    // -------------- outer iteration (trust_region.py:248-422) ---------------------------
    if (has_trace && kiter < trace.cap && lead) {
      const size_t q = (size_t)b * trace.cap + kiter;
      trace.d_Delta[q] = Delta;
      trace.d_numit[q] = j;
      trace.d_stop[q] = stop_tCG;
      trace.d_f_before[q] = fx;
    }
    double x_prop[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) x_prop[e] = x[e] + eta[e];   // :248 retr
    const double fx_prop = cx.cost(x_prop);            // :251
    double rhonum = fx - fx_prop;                      // :255
    double gd[2] = {vdot(g, eta), vdot(eta, Heta)};
    cx.template sum_n<2>(gd);
    double rhoden = -gd[0] - 0.5 * gd[1];              // :256
    const double rho_reg = fmax(1.0, fabs(fx)) * 2.220446049250313e-16 * p.rho_regularization;  // :287
    rhonum += rho_reg;                                 // :288
    rhoden += rho_reg;                                 // :289
    const bool model_decreased = rhoden >= 0.0;        // :311
    const double rho = rhonum / rhoden;                // :317
    if (rho < 0.25 || !model_decreased || !(rho == rho)) {  // :336
      Delta = Delta / 4.0;                             // :338
    } else if (rho > 0.75 && (stop_tCG == TCG_NEGATIVE_CURVATURE || stop_tCG == TCG_EXCEEDED_TR)) {
      Delta = fmin(2.0 * Delta, Delta_bar);            // :357-361
    }
    int accept = 0;
    prev_rejected = true;
    if (UNI(model_decreased && rho > p.rho_prime)) {   // :382
      accept = 1;
      prev_rejected = false;
      ++n_accept;
#pragma unroll
      for (int e = 0; e < NE; ++e) x[e] = x_prop[e];   // :385
      fx = fx_prop;                                    // :386
      cx.commit(x, g);                                 // :387 (rows of x_prop are in LDS)
      cx.proj_setup(x);
      norm_grad = grad_norm_and_rho_vec(cx, g, rho0);  // :388
    }
    if (has_trace && kiter < trace.cap && lead) {
      const size_t q = (size_t)b * trace.cap + kiter;
      trace.d_gradnorm_after[q] = norm_grad;
      trace.d_accept[q] = accept;
    }
    kiter = kiter + 1;                                 // :394
    // :414-416 stopping criterion (pymanopt 0.2.5 order: maxiter before gradnorm; no wall-clock test)
    if (kiter >= p.maxiter) { stop = 1; break; }
    if (UNI(norm_grad < p.mingradnorm)) { stop = 0; break; }
    if (UNI(!(norm_grad == norm_grad) || !(fx == fx))) { bad = true; break; }
    if constexpr (SLICE) {
      if (slice_its > 0 && ++slice_count >= slice_its) { paused = PAUSE_YIELD; break; }
    }
  }
  if (bad) stop = 2;
#ifdef GIK_NPT_PROF
  if (prof && lead)
    for (int i = 0; i < 10; ++i) dbg_buf[16 + i] = (double)cx.pf[i];
#endif
  if (prof && lead) {
    dbg_buf[0] = (double)prof_tcg;
    dbg_buf[1] = (double)inner_total;
    dbg_buf[2] = (double)((long long)__builtin_readcyclecounter() - prof_t0);
    dbg_buf[3] = (double)inner_exec;
  }
  out.f = fx;
  out.gradnorm = norm_grad;
  out.iterations = kiter;
  out.inner_total = inner_total;
  out.inner_executed = inner_exec;
  out.stop = stop;
  out.n_accept = n_accept;
  out.Delta = Delta;
  out.paused = paused;
}

}  // namespace gik
