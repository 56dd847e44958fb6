// graphik_amd/csrc/gik_rtr.hip.h -- the Riemannian trust-region driver shared by the
// wave-per-problem and the workgroup-per-problem kernels.
//
// rtr_solve_one<Ctx>() is TrustRegions.solve (graphik/solvers/trust_region.py:112-434) with
// _truncated_conjugate_gradient (:436-599) inlined, written against a context `Ctx` that owns the
// problem data and provides
//     cost(x), commit() -> egrad entry, proj_setup(flag), hess_proj_dot(delta, s, d_Hd, hd),
//     static sum_n<NV>(v)   (reduction over all unknowns, result uniform in every thread),
//     pk2[NC], lead()       (the one thread that writes per-problem scalars).
// Every thread holds one entry of each tangent vector (or zero if it owns none).
#pragma once

#include <hip/hip_runtime.h>

#include "gik_wave.hip.h"

namespace gik {

// Branch conditions on solver scalars are identical in every thread of the problem; routing them
// through a ballot makes that explicit (the predicate lands in an SGPR pair, the branch is scalar)
// so the structurizer never builds exec-masked loops around the reductions.
#define UNI(cond) (__builtin_amdgcn_ballot_w64(cond) != 0ull)

struct RtrOut {
  double f, gradnorm;
  int iterations, inner_total, stop, n_accept;
};

// ||g||_F together with <g, pk2_m> in one reduction
template <typename Ctx>
__device__ inline double grad_norm_and_rho(Ctx &cx, double g, double (&rho0)[Ctx::NC]) {
  double v[Ctx::NC + 1];
  v[0] = g * g;
#pragma unroll
  for (int m = 0; m < Ctx::NC; ++m) v[m + 1] = g * cx.pk2[m];
  cx.template sum_n<Ctx::NC + 1>(v);
#pragma unroll
  for (int m = 0; m < Ctx::NC; ++m) rho0[m] = v[m + 1];
  return sqrt(v[0]);
}

template <int K, typename Ctx>
__device__ inline void rtr_solve_one(Ctx &cx, const Params &p, const gik_trace &trace, int has_trace,
                                     int dbg, double *dbg_buf, int b, double &x, RtrOut &out) {
  const double Delta_bar = 10.0 + K;  // typicaldist (fixed_rank_psd_sym.py:71-73)
  const bool lead = cx.lead();
    double Delta = Delta_bar / 8.0;         // trust_region.py:134-135,164
    double fx = cx.cost(x);                 // :159
    double g = cx.commit();                 // :160  (also loads the slot constants at x)
    cx.proj_setup(p.planar_proj_exact);
    // ||grad|| (:161) and rho0_m = <grad, pk2_m> (start values of the tCG recurrences)
    double rho0[Ctx::NC];
    double norm_grad = grad_norm_and_rho(cx, g, rho0);
    int kiter = 0, inner_total = 0, n_accept = 0, stop = 1;
    bool bad = UNI(!(fx == fx) || !(norm_grad == norm_grad));
    if (dbg & 2) bad = true;

    while (!bad) {
      // -------------- _truncated_conjugate_gradient (trust_region.py:436-599) -------------
      double eta = 0.0, Heta = 0.0, r = g;       // :444-448
      double e_Pe = 0.0;
      double r_r = cx.sum1(r * r);              // :455
      const double norm_r0 = sqrt(r_r);
      const double nr0_theta = (p.theta == 1.0) ? norm_r0 : pow(norm_r0, p.theta);
      const double target = norm_r0 * fmin(nr0_theta, p.kappa);  // rhs of :572
      const double target2 = target * target;
      double z_r = r_r, d_Pd = r_r;              // :464-466 (precon = identity)
#if GIK_FASTDIV
      double inv_z_r = frcp(z_r);
#endif
      double delta = -r;                         // :469
      double e_Pd = 0.0, model_value = 0.0;      // :471,485
      int stop_tCG = TCG_MAX_INNER_ITER;         // :491
      double rho_pk[Ctx::NC], s_pk[Ctx::NC], hd_pk[Ctx::NC];  // <r,pk2>, <delta,pk2>, <Hdelta,pk2>
#pragma unroll
      for (int m = 0; m < Ctx::NC; ++m) {
        rho_pk[m] = rho0[m];
        s_pk[m] = -rho0[m];
      }
      int j = 0;
      for (j = 0; j < p.maxinner; ++j) {         // :495
        double d_Hd;
        const double Hdelta = cx.hess_proj_dot(delta, s_pk, d_Hd, hd_pk);  // :497-500
        if (UNI(!(d_Hd == d_Hd))) { bad = true; break; }
#if GIK_FASTDIV
        const double alpha = z_r * frcp(d_Hd);            // :503
#else
        const double alpha = z_r / d_Hd;                  // :503
#endif
        const double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd;  // :506
        if (UNI(d_Hd <= 0.0 || e_Pe_new >= Delta * Delta)) {   // :509
          const double tau =
              (-e_Pd + sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd;  // :514
          eta = eta + tau * delta;                        // :516
          Heta = Heta + tau * Hdelta;                     // :521
          stop_tCG = (d_Hd <= 0.0) ? TCG_NEGATIVE_CURVATURE : TCG_EXCEEDED_TR;  // :531-534
          break;
        }
        if ((dbg & 4) && b == 0 && lead && dbg_buf && kiter < 64 && j < 128) {
          double *q = dbg_buf + ((size_t)kiter * 128 + j) * 4;
          q[0] = r_r; q[1] = d_Hd; q[2] = alpha; q[3] = model_value;
        }
        e_Pe = e_Pe_new;                                  // :537
        const double new_eta = eta + alpha * delta;       // :538
        const double new_Heta = Heta + alpha * Hdelta;    // :542
        const double new_r = r + alpha * Hdelta;          // :561 (speculative; same value)
        double m[3] = {new_eta * g, new_eta * new_Heta, new_r * new_r};
        cx.template sum_n<3>(m);
        const double new_model_value = m[0] + 0.5 * m[1]; // :551
        if (UNI(new_model_value >= model_value)) {        // :552
          stop_tCG = TCG_MODEL_INCREASED;
          break;
        }
        eta = new_eta;                                    // :556-558
        Heta = new_Heta;
        model_value = new_model_value;
        r = new_r;                                        // :561
        r_r = m[2];                                       // :564
        // :572  norm_r <= norm_r0*min(norm_r0^theta, kappa), compared on the squares
        if (UNI(j >= p.mininner && r_r <= target2)) {
          stop_tCG = (p.kappa < nr0_theta) ? TCG_REACHED_TARGET_LINEAR
                                           : TCG_REACHED_TARGET_SUPERLINEAR;
          break;
        }
        const double zold_rold = z_r;                     // :587
        z_r = r_r;                                        // :589
#if GIK_FASTDIV
        const double beta = z_r * inv_z_r;                // :592 (1/z_r_old, formed off the critical path)
        inv_z_r = frcp(z_r);
#else
        const double beta = z_r / zold_rold;              // :592
#endif
        delta = -r + beta * delta;                        // :593
        if constexpr (K == 2) {                           // the same two updates seen through pk2
#pragma unroll
          for (int m = 0; m < Ctx::NC; ++m) {
            rho_pk[m] = fma(alpha, hd_pk[m], rho_pk[m]);
            s_pk[m] = fma(beta, s_pk[m], -rho_pk[m]);
          }
        }
        e_Pd = beta * (e_Pd + alpha * d_Pd);              // :596
        d_Pd = z_r + beta * beta * d_Pd;                  // :597
      }
      if (bad) break;
      if (j >= p.maxinner) j = p.maxinner - 1;  // Python leaves j at the last index
      inner_total += j + 1;

      // -------------- outer iteration (trust_region.py:248-422) ---------------------------
      if (has_trace && kiter < trace.cap && lead) {
        const size_t q = (size_t)b * trace.cap + kiter;
        trace.d_Delta[q] = Delta;
        trace.d_numit[q] = j;
        trace.d_stop[q] = stop_tCG;
        trace.d_f_before[q] = fx;
      }
      const double x_prop = x + eta;                     // :248 retr
      const double fx_prop = cx.cost(x_prop);            // :251
      double rhonum = fx - fx_prop;                      // :255
      double gd[2] = {g * eta, eta * Heta};
      cx.template sum_n<2>(gd);
      double rhoden = -gd[0] - 0.5 * gd[1];              // :256
      const double rho_reg =
          fmax(1.0, fabs(fx)) * 2.220446049250313e-16 * p.rho_regularization;  // :287
      rhonum += rho_reg;                                 // :288
      rhoden += rho_reg;                                 // :289
      const bool model_decreased = rhoden >= 0.0;        // :311
      const double rho = rhonum / rhoden;                // :317
      if (rho < 0.25 || !model_decreased || !(rho == rho)) {  // :336
        Delta = Delta / 4.0;                             // :338
      } else if (rho > 0.75 &&
                 (stop_tCG == TCG_NEGATIVE_CURVATURE || stop_tCG == TCG_EXCEEDED_TR)) {
        Delta = fmin(2.0 * Delta, Delta_bar);            // :357-361
      }
      int accept = 0;
      if (UNI(model_decreased && rho > p.rho_prime)) {   // :382
        accept = 1;
        ++n_accept;
        x = x_prop;                                      // :385
        fx = fx_prop;                                    // :386
        g = cx.commit();                                 // :387 (rows of x_prop are in LDS)
        cx.proj_setup(p.planar_proj_exact);
        norm_grad = grad_norm_and_rho(cx, g, rho0);      // :388
      }
      if (has_trace && kiter < trace.cap && lead) {
        const size_t q = (size_t)b * trace.cap + kiter;
        trace.d_gradnorm_after[q] = norm_grad;
        trace.d_accept[q] = accept;
      }
      kiter = kiter + 1;                                 // :394
      // :414-416 stopping criterion (pymanopt 0.2.5 order: maxiter before gradnorm; the
      // wall-clock maxtime test is not reproduced -- it is non-deterministic)
      if (kiter >= p.maxiter) { stop = 1; break; }
      if (UNI(norm_grad < p.mingradnorm)) { stop = 0; break; }
      if (UNI(!(norm_grad == norm_grad) || !(fx == fx))) { bad = true; break; }
    }
    if (bad) stop = 2;

    out.f = fx;
    out.gradnorm = norm_grad;
    out.iterations = kiter;
    out.inner_total = inner_total;
    out.stop = stop;
    out.n_accept = n_accept;
}

}  // namespace gik
