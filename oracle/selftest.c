/* oracle/selftest.c -- sanitizer harness for the CPU restatement (test infrastructure).
 *
 *   selftest <problem.bin>  -> prints one checksum line per entry point
 *
 * Built twice by the Makefile from the same sources: plain (-O2 -ffp-contract=off) and with
 * -fsanitize=address,undefined -fno-sanitize-recover=all.  tests/test_oracle_sanitizers.py writes
 * problems taken from the golden fixtures, runs both and requires identical output and a clean
 * sanitizer exit -- out-of-bounds indexing, use of uninitialised scratch and signed overflow in the
 * oracle would otherwise only show up as wrong parity verdicts.
 *
 * problem.bin: int32 N, k, use_limits, n_at;  then doubles: Y[N*k], W[N*k], D[N*N], omega[N*N],
 * psi_L[N*N], psi_U[N*N], lower[N*N], upper[N*N] (NaN = no edge), at_pos[n_at*3], at_target[n_at];
 * then int32 at_node[n_at], at_kind[n_at].
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gik_oracle.h"

static double *rd(FILE *f, size_t n) {
  double *p = (double *)malloc(sizeof(double) * (n ? n : 1));
  if (fread(p, sizeof(double), n, f) != n) exit(3);
  return p;
}
static double sum(const double *a, size_t n) {
  double s = 0;
  for (size_t i = 0; i < n; ++i) s += a[i] * (1.0 + 1e-3 * (double)(i % 7));
  return s;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t hdr[4];
  if (fread(hdr, sizeof(int32_t), 4, f) != 4) return 3;
  const int N = hdr[0], k = hdr[1], use_limits = hdr[2], n_at = hdr[3];
  const size_t n = (size_t)N * k, NN = (size_t)N * N;
  double *Y = rd(f, n), *W = rd(f, n), *D = rd(f, NN), *om = rd(f, NN), *pL = rd(f, NN), *pU = rd(f, NN);
  double *lower = rd(f, NN), *upper = rd(f, NN), *apos = rd(f, (size_t)n_at * 3), *atgt = rd(f, n_at);
  int *anode = (int *)malloc(sizeof(int) * (n_at ? n_at : 1)), *akind = (int *)malloc(sizeof(int) * (n_at ? n_at : 1));
  if (fread(anode, sizeof(int), n_at, f) != (size_t)n_at || fread(akind, sizeof(int), n_at, f) != (size_t)n_at) return 3;
  fclose(f);
  /* index pairs of create_cost_limits / create_cost */
  int64_t *ii = (int64_t *)malloc(sizeof(int64_t) * NN), *jj = (int64_t *)malloc(sizeof(int64_t) * NN);
  int64_t ni = 0;
  for (int i = 0; i < N; ++i)
    for (int j = i; j < N; ++j) {
      const size_t e = (size_t)i * N + j;
      const int hinge = use_limits && pL[e] != pU[e] && (pL[e] > 0 || pU[e] > 0);
      if (om[e] != 0 || hinge) {
        ii[ni] = i;
        jj[ni] = j;
        ++ni;
      }
    }
  double *out = (double *)malloc(sizeof(double) * n);
  printf("lcost %.17g\n", gik_o_lcost(Y, D, om, pL, pU, ii, jj, ni, N, k));
  gik_o_lgrad(Y, D, om, pL, pU, ii, jj, ni, N, k, out);
  printf("lgrad %.17g\n", sum(out, n));
  gik_o_lhess(Y, W, D, om, pL, pU, ii, jj, ni, N, k, out);
  printf("lhess %.17g\n", sum(out, n));
  printf("jcost %.17g\n", gik_o_jcost(Y, D, ii, jj, ni, N, k));
  printf("lcost_and_grad %.17g\n", gik_o_lcost_and_grad(Y, D, om, pL, pU, ii, jj, ni, N, k, out));
  if (gik_o_proj(Y, W, N, k, out) == 0) printf("proj %.17g\n", sum(out, n));
  double *lb = (double *)malloc(sizeof(double) * NN), *ub = (double *)malloc(sizeof(double) * NN);
  gik_o_bound_smoothing(lower, upper, N, lb, ub);
  printf("bounds %.17g %.17g\n", sum(lb, NN), sum(ub, NN));
  gik_o_params p;
  gik_o_default_params(&p);
  p.use_limits = use_limits;
  p.maxiter = 60;
  gik_o_result res;
  double *x = (double *)malloc(sizeof(double) * n);
  double tD[16], tf[16], tg[16];
  int tn[16], ts[16], ta[16];
  gik_o_traj tr = {16, 0, tD, tn, ts, tf, tg, ta};
  for (size_t t = 0; t < n; ++t) x[t] = Y[t];
  gik_o_rtr_solve(x, D, om, pL, pU, ii, jj, ni, N, k, &p, &res, &tr);
  printf("rtr %.17g %.17g %d %d %d traj %d %.17g\n", res.f, sum(x, n), res.iterations, res.inner_total, res.stop,
         tr.len, tr.len ? tf[tr.len - 1] : 0.0);
  gik_o_cg_params cp;
  gik_o_cg_default_params(&cp);
  cp.use_limits = use_limits;
  cp.maxiter = 200;
  double cf[32], cg[32], cs[32];
  int ce[32];
  gik_o_cg_traj ct = {32, 0, cf, cg, cs, ce};
  for (size_t t = 0; t < n; ++t) x[t] = Y[t];
  gik_o_cg_solve(x, D, om, pL, pU, ii, jj, ni, N, k, &cp, &res, &ct);
  printf("cg %.17g %.17g %d %d %d traj %d\n", res.f, sum(x, n), res.iterations, res.inner_total, res.stop, ct.len);
  if (n_at > 0) {
    gik_o_anchor_terms at = {n_at, anode, apos, atgt, akind};
    for (size_t t = 0; t < n; ++t) x[t] = Y[t];
    gik_o_rtr_solve_anchored(x, D, om, pL, pU, ii, jj, ni, N, k, &at, &p, &res, 0);
    printf("anchored %.17g %.17g %d %d\n", res.f, sum(x, n), res.iterations, res.inner_total);
  }
  free(Y); free(W); free(D); free(om); free(pL); free(pU); free(lower); free(upper); free(apos); free(atgt);
  free(anode); free(akind); free(ii); free(jj); free(out); free(lb); free(ub); free(x);
  return 0;
}
