// graphik_amd/csrc/gik_rcg.hip.h -- Riemannian conjugate gradients, the reference's alternative
// solver: RiemannianSolver(graph, {"solver": "ConjugateGradient"})
// (graphik/solvers/riemannian_solver.py:51-59) = pymanopt 0.2.5 ConjugateGradient with
// beta_type HagerZhang, orth_value 10e10, maxiter 10e4, mingradnorm 1e-9, minstepsize 1e-10 and
// the default LineSearchAdaptive, on the reference's manifold (fixed_rank_psd_sym.py: Frobenius
// metric, retr(Y, U) = Y + U, transp(Y, Z, U) = proj(Z, U), egrad2rgrad = identity).  pymanopt is a
// third-party dependency outside /root/reference; its algorithm is restated here and tested against
// a CPU twin that is pinned to vectors captured from the reference's own solve() (tests/golden/cg.npz).
//
// Written against the same context interface as rtr_solve_one (WaveCtx / BlockCtx): every thread
// holds one entry of each vector; cost(x) publishes x, commit() returns the gradient entry there,
// proj_setup() factors the projector at the committed point, proj(Z) applies it.
#pragma once

#include <hip/hip_runtime.h>

#include "gik_rtr.hip.h"

namespace gik {

struct CgParams {
  double mingradnorm, minstepsize, orth_value;
  int maxiter, beta_type, planar_proj_exact;
};

enum { CG_FLETCHER_REEVES = 0, CG_POLAK_RIBIERE = 1, CG_HESTENES_STIEFEL = 2, CG_HAGER_ZHANG = 3 };

// trace columns (gik_trace, per iteration q): d_f_before = cost before the step, d_gradnorm_after
// = |grad| before the step, d_Delta = step size returned by the line search, d_numit = its cost
// evaluations, d_stop = 1 if the direction was reset to -grad (not a descent direction),
// d_accept = 1 if the line search moved (alpha != 0)
template <int K, typename Ctx>
__device__ inline void rcg_solve_one(Ctx &cx, const CgParams &p, const gik_trace &trace, int has_trace,
                                     int b, double &x, RtrOut &out) {
  const bool lead = cx.lead();
  int iter = 0, stop = 1, costevals = 0, moved = 0;
  double stepsize = __builtin_nan("");
  double cost = cx.cost(x);                          // objective(x)
  double g = cx.commit();                            // gradient(x) (egrad2rgrad = identity)
  double gradPgrad = cx.sum1(g * g);                 // man.inner(x, grad, Pgrad), precon = identity
  double gradnorm = sqrt(gradPgrad);                 // man.norm(x, grad)
  double desc = -g;                                  // initial descent direction
  double oldalpha = 0.0;
  bool have_old = false;                             // LineSearchAdaptive._oldalpha is None
  for (;;) {
    // _check_stopping_criterion(iter = iter + 1): maxiter, gradnorm, stepsize (NaN compares false);
    // the wall-clock maxtime test is not reproduced
    if (iter + 1 >= p.maxiter) { stop = 1; break; }
    if (UNI(gradnorm < p.mingradnorm)) { stop = 0; break; }
    if (UNI(stepsize < p.minstepsize)) { stop = 3; break; }
    if (UNI(!(cost == cost) || !(gradnorm == gradnorm))) { stop = 2; break; }
    double v2[2] = {g * desc, desc * desc};
    cx.template sum_n<2>(v2);
    double df0 = v2[0];
    int restarted = 0;
    if (UNI(df0 >= 0.0)) {                           // not a descent direction: restart
      desc = -g;
      df0 = -gradPgrad;
      v2[1] = gradPgrad;
      restarted = 1;
    }
    // ---- LineSearchAdaptive.search ----
    const double norm_d = sqrt(v2[1]);
    double alpha = have_old ? oldalpha : 1.0 / norm_d;
    double newx = fma(alpha, desc, x);               // man.retr(x, alpha * d)
    double newf = cx.cost(newx);
    int evals = 1;
    while (UNI(newf > cost + 0.5 * alpha * df0 && evals <= 10)) {
      alpha *= 0.5;
      newx = fma(alpha, desc, x);
      newf = cx.cost(newx);
      ++evals;
    }
    if (UNI(newf > cost)) {
      alpha = 0.0;
      newx = x;
    }
    stepsize = alpha * norm_d;
    oldalpha = (evals == 2) ? alpha : alpha + alpha;
    have_old = true;
    costevals += evals;
    const bool did_move = UNI(alpha != 0.0);
    moved += did_move ? 1 : 0;
    if (has_trace && iter < trace.cap) {   // every thread stores the same values (no divergent branch inside the solver loop)
      const size_t q = (size_t)b * trace.cap + iter;
      trace.d_f_before[q] = cost;
      trace.d_gradnorm_after[q] = gradnorm;
      trace.d_Delta[q] = stepsize;
      trace.d_numit[q] = evals;
      trace.d_stop[q] = restarted;
      trace.d_accept[q] = did_move ? 1 : 0;
    }
    // ---- quantities at the new point ----
    const double newcost = cx.cost(newx);            // objective(newx) (publishes newx)
    const double ng = cx.commit();                   // gradient(newx)
    cx.proj_setup(p.planar_proj_exact);              // projector at newx (man.transp = proj(newx, .))
    const double oldgrad = cx.proj(g);               // man.transp(x, newx, grad)
    const double tdesc = cx.proj(desc);              // man.transp(x, newx, desc_dir)
    const double diff = ng - oldgrad;                // (also Pdiff: precon = identity)
    double v[8] = {ng * ng, oldgrad * ng, diff * tdesc, diff * ng, diff * diff, tdesc * ng,
                   tdesc * tdesc, 0.0};
    cx.template sum_n<8>(v);
    const double newgradPnewgrad = v[0];
    const double orth_grads = v[1] / newgradPnewgrad;
    if (UNI(fabs(orth_grads) >= p.orth_value)) {     // Powell's restart strategy
      desc = -ng;
    } else {
      double beta;
      if (p.beta_type == CG_FLETCHER_REEVES) {
        beta = newgradPnewgrad / gradPgrad;
      } else if (p.beta_type == CG_POLAK_RIBIERE) {
        beta = fmax(0.0, v[3] / gradPgrad);
      } else if (p.beta_type == CG_HESTENES_STIEFEL) {
        // numpy scalars never raise ZeroDivisionError, so pymanopt's `except: beta = 1` arm is dead
        // code on this path: a zero denominator gives inf / nan and Python's max(0, .) keeps inf,
        // maps -inf and nan to 0 -- which is what fmax does
        beta = fmax(0.0, v[3] / v[2]);
      } else {                                        // Hager-Zhang
        const double deno = v[2];
        double numo = v[3];
        numo -= 2.0 * v[4] * v[5] / deno;
        beta = numo / deno;
        const double eta_HZ = -1.0 / (sqrt(v[6]) * fmin(0.01, gradnorm));
        beta = (eta_HZ > beta) ? eta_HZ : beta;       // Python's max(beta, eta_HZ): a NaN beta stays NaN
      }
      desc = fma(beta, tdesc, -ng);
    }
    x = newx;
    cost = newcost;
    g = ng;
    gradPgrad = newgradPnewgrad;
    gradnorm = sqrt(newgradPnewgrad);
    ++iter;
  }
  out.f = cost;
  out.gradnorm = gradnorm;
  out.iterations = iter;
  out.inner_total = costevals;
  out.inner_executed = costevals;
  out.stop = stop;
  out.n_accept = moved;
  out.Delta = stepsize;
  out.paused = 0;
}

}  // namespace gik
