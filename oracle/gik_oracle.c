/*
 * oracle/gik_oracle.c -- CPU restatement (plain C, fp64) of GraphIK's RiemannianSolver hot path.
 *
 * TEST INFRASTRUCTURE ONLY (checker + CPU baseline); see gik_oracle.h.  Every function cites the
 * reference file:line it follows.  Build with -ffp-contract=off so that, like the Python loops,
 * every multiply and add rounds separately.
 */
#include "gik_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define IDX(i, j, N) ((size_t)(i) * (size_t)(N) + (size_t)(j))

/* ------------------------------------------------------------------------------------------
 * graphik/solvers/costs.py:8-16   jcost
 * ------------------------------------------------------------------------------------------ */
double gik_o_jcost(const double *Y, const double *D_goal, const int64_t *ii, const int64_t *jj,
                   int64_t n_inds, int N, int k) {
  double cost = 0.0;
  for (int64_t e = 0; e < n_inds; ++e) {
    int64_t idx = ii[e], jdx = jj[e];
    double nrm = 0.0;
    for (int kdx = 0; kdx < k; ++kdx) {
      double d = Y[idx * k + kdx] - Y[jdx * k + kdx];
      nrm += d * d;
    }
    double r = D_goal[IDX(idx, jdx, N)] - nrm;
    cost += 2.0 * (r * r);
  }
  return 0.5 * cost;
}

/* costs.py:20-35   jgrad */
void gik_o_jgrad(const double *Y, const double *D_goal, const int64_t *ii, const int64_t *jj,
                 int64_t n_inds, int N, int k, double *grad) {
  memset(grad, 0, sizeof(double) * (size_t)N * k);
  for (int64_t e = 0; e < n_inds; ++e) {
    int64_t idx = ii[e], jdx = jj[e];
    double nrm = 0.0;
    for (int kdx = 0; kdx < k; ++kdx) {
      double d = Y[idx * k + kdx] - Y[jdx * k + kdx];
      nrm += d * d;
    }
    for (int kdx = 0; kdx < k; ++kdx) {
      grad[idx * k + kdx] +=
          -4.0 * (D_goal[IDX(idx, jdx, N)] - nrm) * (Y[idx * k + kdx] - Y[jdx * k + kdx]);
      grad[jdx * k + kdx] +=
          -4.0 * (D_goal[IDX(jdx, idx, N)] - nrm) * (Y[jdx * k + kdx] - Y[idx * k + kdx]);
    }
  }
  for (int t = 0; t < N * k; ++t) grad[t] *= 0.5;
}

/* costs.py:39-58   jhess */
void gik_o_jhess(const double *Y, const double *w, const double *D_goal, const int64_t *ii,
                 const int64_t *jj, int64_t n_inds, int N, int k, double *hess) {
  memset(hess, 0, sizeof(double) * (size_t)N * k);
  for (int64_t e = 0; e < n_inds; ++e) {
    int64_t idx = ii[e], jdx = jj[e];
    double nrm = 0.0, sc = 0.0;
    for (int kdx = 0; kdx < k; ++kdx) {
      double dy = Y[idx * k + kdx] - Y[jdx * k + kdx];
      sc += dy * (w[idx * k + kdx] - w[jdx * k + kdx]);
      nrm += dy * dy;
    }
    for (int kdx = 0; kdx < k; ++kdx) {
      hess[idx * k + kdx] +=
          4.0 * (2.0 * sc * (Y[idx * k + kdx] - Y[jdx * k + kdx]) +
                 (nrm - D_goal[IDX(idx, jdx, N)]) * (w[idx * k + kdx] - w[jdx * k + kdx]));
      hess[jdx * k + kdx] +=
          4.0 * (2.0 * sc * (Y[jdx * k + kdx] - Y[idx * k + kdx]) +
                 (nrm - D_goal[IDX(jdx, idx, N)]) * (w[jdx * k + kdx] - w[idx * k + kdx]));
    }
  }
  for (int t = 0; t < N * k; ++t) hess[t] *= 0.5;
}

/* costs.py:61-77   jcost_and_grad : one pass, returns (0.5 * sum 2 r^2, 0.5 * grad) */
double gik_o_jcost_and_grad(const double *Y, const double *D_goal, const int64_t *ii,
                            const int64_t *jj, int64_t n_inds, int N, int k, double *grad) {
  double cost = 0.0;
  memset(grad, 0, sizeof(double) * (size_t)N * k);
  for (int64_t e = 0; e < n_inds; ++e) {
    int64_t idx = ii[e], jdx = jj[e];
    double nrm = 0.0;
    for (int kdx = 0; kdx < k; ++kdx) {
      double d = Y[idx * k + kdx] - Y[jdx * k + kdx];
      nrm += d * d;
    }
    for (int kdx = 0; kdx < k; ++kdx) {
      grad[idx * k + kdx] +=
          -4.0 * (D_goal[IDX(idx, jdx, N)] - nrm) * (Y[idx * k + kdx] - Y[jdx * k + kdx]);
      grad[jdx * k + kdx] +=
          -4.0 * (D_goal[IDX(jdx, idx, N)] - nrm) * (Y[jdx * k + kdx] - Y[idx * k + kdx]);
    }
    double r = D_goal[IDX(idx, jdx, N)] - nrm;
    cost += 2.0 * (r * r);
  }
  for (int t = 0; t < N * k; ++t) grad[t] *= 0.5;
  return 0.5 * cost;
}

/* costs.py:126-169   lcost_and_grad */
double gik_o_lcost_and_grad(const double *Y, const double *D_goal, const double *omega,
                            const double *psi_L, const double *psi_U, const int64_t *ii,
                            const int64_t *jj, int64_t n_inds, int N, int k, double *grad) {
  double cost = 0.0;
  memset(grad, 0, sizeof(double) * (size_t)N * k);
  for (int64_t e = 0; e < n_inds; ++e) {
    int64_t idx = ii[e], jdx = jj[e];
    size_t ij = IDX(idx, jdx, N), ji = IDX(jdx, idx, N);
    double nrm = 0.0;
    for (int kdx = 0; kdx < k; ++kdx) {
      double d = Y[idx * k + kdx] - Y[jdx * k + kdx];
      nrm += d * d;
    }
    if (omega[ij] > 0) {
      double r = D_goal[ij] - nrm;
      cost += 2.0 * (r * r);
      for (int kdx = 0; kdx < k; ++kdx) {
        grad[idx * k + kdx] += 4.0 * (nrm - D_goal[ij]) * (Y[idx * k + kdx] - Y[jdx * k + kdx]);
        grad[jdx * k + kdx] += 4.0 * (nrm - D_goal[ji]) * (Y[jdx * k + kdx] - Y[idx * k + kdx]);
      }
    }
    if (psi_L[ij] > 0) {
      double r = fmax(psi_L[ij] - nrm, 0.0);
      cost += 2.0 * (r * r);
      if (fmax(psi_L[ij] - nrm, 0.0) > 0) {
        for (int kdx = 0; kdx < k; ++kdx) {
          grad[idx * k + kdx] += 4.0 * (nrm - psi_L[ij]) * (Y[idx * k + kdx] - Y[jdx * k + kdx]);
          grad[jdx * k + kdx] += 4.0 * (nrm - psi_L[ji]) * (Y[jdx * k + kdx] - Y[idx * k + kdx]);
        }
      }
    }
    if (psi_U[ij] > 0) {
      double r = fmax(-psi_U[ij] + nrm, 0.0);
      cost += 2.0 * (r * r);
      if (fmax(-psi_U[ij] + nrm, 0.0) > 0) {
        for (int kdx = 0; kdx < k; ++kdx) {
          grad[idx * k + kdx] += 4.0 * (nrm - psi_U[ij]) * (Y[idx * k + kdx] - Y[jdx * k + kdx]);
          grad[jdx * k + kdx] += 4.0 * (nrm - psi_U[ji]) * (Y[jdx * k + kdx] - Y[idx * k + kdx]);
        }
      }
    }
  }
  for (int t = 0; t < N * k; ++t) grad[t] *= 0.5;
  return 0.5 * cost;
}

/* costs.py:80-93   lcost */
double gik_o_lcost(const double *Y, const double *D_goal, const double *omega,
                   const double *psi_L, const double *psi_U, const int64_t *ii,
                   const int64_t *jj, int64_t n_inds, int N, int k) {
  double cost = 0.0;
  for (int64_t e = 0; e < n_inds; ++e) {
    int64_t idx = ii[e], jdx = jj[e];
    size_t ij = IDX(idx, jdx, N);
    double nrm = 0.0;
    for (int kdx = 0; kdx < k; ++kdx) {
      double d = Y[idx * k + kdx] - Y[jdx * k + kdx];
      nrm += d * d;
    }
    if (omega[ij] > 0) {
      double r = D_goal[ij] - nrm;
      cost += r * r;
    }
    if (psi_L[ij] > 0) {
      double r = fmax(psi_L[ij] - nrm, 0.0);
      cost += r * r;
    }
    if (psi_U[ij] > 0) {
      double r = fmax(-psi_U[ij] + nrm, 0.0);
      cost += r * r;
    }
  }
  return cost;
}

/* costs.py:98-123   lgrad */
void gik_o_lgrad(const double *Y, const double *D_goal, const double *omega, const double *psi_L,
                 const double *psi_U, const int64_t *ii, const int64_t *jj, int64_t n_inds,
                 int N, int k, double *grad) {
  memset(grad, 0, sizeof(double) * (size_t)N * k);
  for (int64_t e = 0; e < n_inds; ++e) {
    int64_t idx = ii[e], jdx = jj[e];
    size_t ij = IDX(idx, jdx, N);
    double nrm = 0.0;
    for (int kdx = 0; kdx < k; ++kdx) {
      double d = Y[idx * k + kdx] - Y[jdx * k + kdx];
      nrm += d * d;
    }
    if (omega[ij] != 0.0) {
      for (int kdx = 0; kdx < k; ++kdx) {
        double a = (nrm - D_goal[ij]) * (Y[idx * k + kdx] - Y[jdx * k + kdx]);
        grad[idx * k + kdx] += a;
        grad[jdx * k + kdx] += -a;
      }
    }
    if (psi_L[ij] != 0.0) {
      if (fmax(psi_L[ij] - nrm, 0.0) > 0) {
        for (int kdx = 0; kdx < k; ++kdx) {
          double a = (nrm - psi_L[ij]) * (Y[idx * k + kdx] - Y[jdx * k + kdx]);
          grad[idx * k + kdx] += a;
          grad[jdx * k + kdx] += -a;
        }
      }
    }
    if (psi_U[ij] != 0.0) {
      if (fmax(-psi_U[ij] + nrm, 0.0) > 0) {
        for (int kdx = 0; kdx < k; ++kdx) {
          double a = (nrm - psi_U[ij]) * (Y[idx * k + kdx] - Y[jdx * k + kdx]);
          grad[idx * k + kdx] += a;
          grad[jdx * k + kdx] += -a;
        }
      }
    }
  }
  for (int t = 0; t < N * k; ++t) grad[t] *= 2.0;
}

/* costs.py:175-207   lhess */
void gik_o_lhess(const double *Y, const double *w, const double *D_goal, const double *omega,
                 const double *psi_L, const double *psi_U, const int64_t *ii, const int64_t *jj,
                 int64_t n_inds, int N, int k, double *hess) {
  memset(hess, 0, sizeof(double) * (size_t)N * k);
  for (int64_t e = 0; e < n_inds; ++e) {
    int64_t idx = ii[e], jdx = jj[e];
    size_t ij = IDX(idx, jdx, N);
    double nrm = 0.0, sc = 0.0;
    for (int kdx = 0; kdx < k; ++kdx) {
      double dy = Y[idx * k + kdx] - Y[jdx * k + kdx];
      nrm += dy * dy;
      sc += dy * (w[idx * k + kdx] - w[jdx * k + kdx]);
    }
    if (omega[ij] != 0.0) {
      for (int kdx = 0; kdx < k; ++kdx) {
        double a = 2.0 * sc * (Y[idx * k + kdx] - Y[jdx * k + kdx]);
        double b = (nrm - D_goal[ij]) * (w[idx * k + kdx] - w[jdx * k + kdx]);
        double c = a + b;
        hess[idx * k + kdx] += c;
        hess[jdx * k + kdx] += -c;
      }
    }
    if (psi_L[ij] != 0.0 && fmax(psi_L[ij] - nrm, 0.0) > 0) {
      for (int kdx = 0; kdx < k; ++kdx) {
        double a = 2.0 * sc * (Y[idx * k + kdx] - Y[jdx * k + kdx]);
        double b = (nrm - psi_L[ij]) * (w[idx * k + kdx] - w[jdx * k + kdx]);
        double c = a + b;
        hess[idx * k + kdx] += c;
        hess[jdx * k + kdx] += -c;
      }
    }
    if (psi_U[ij] != 0.0 && fmax(-psi_U[ij] + nrm, 0.0) > 0) {
      for (int kdx = 0; kdx < k; ++kdx) {
        double a = 2.0 * sc * (Y[idx * k + kdx] - Y[jdx * k + kdx]);
        double b = (nrm - psi_U[ij]) * (w[idx * k + kdx] - w[jdx * k + kdx]);
        double c = a + b;
        hess[idx * k + kdx] += c;
        hess[jdx * k + kdx] += -c;
      }
    }
  }
  for (int t = 0; t < N * k; ++t) hess[t] *= 2.0;
}

/* ------------------------------------------------------------------------------------------
 * dense LU solve with partial pivoting (np.linalg.solve == LAPACK dgesv); n <= 9
 * ------------------------------------------------------------------------------------------ */
static int lu_solve(double *A, double *b, int n) {
  for (int c = 0; c < n; ++c) {
    int piv = c;
    double best = fabs(A[c * n + c]);
    for (int r = c + 1; r < n; ++r) {
      double v = fabs(A[r * n + c]);
      if (v > best) {
        best = v;
        piv = r;
      }
    }
    if (best == 0.0) return -1;
    if (piv != c) {
      for (int t = 0; t < n; ++t) {
        double tmp = A[c * n + t];
        A[c * n + t] = A[piv * n + t];
        A[piv * n + t] = tmp;
      }
      double tb = b[c];
      b[c] = b[piv];
      b[piv] = tb;
    }
    for (int r = c + 1; r < n; ++r) {
      double f = A[r * n + c] / A[c * n + c];
      A[r * n + c] = f;
      for (int t = c + 1; t < n; ++t) A[r * n + t] -= f * A[c * n + t];
      b[r] -= f * b[c];
    }
  }
  for (int r = n - 1; r >= 0; --r) {
    double s = b[r];
    for (int t = r + 1; t < n; ++t) s -= A[r * n + t] * b[t];
    b[r] = s / A[r * n + r];
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * graphik/utils/manifolds/fixed_rank_psd_sym.py:91-113   PSDFixedRank.proj
 * ------------------------------------------------------------------------------------------ */
int gik_o_proj(const double *Y, const double *Z, int N, int k, double *out) {
  double X[9], C[9], A[81], rhs[9];
  /* X = Y.T.dot(Y) (:93) */
  for (int a = 0; a < k; ++a)
    for (int b = 0; b < k; ++b) {
      double s = 0.0;
      for (int i = 0; i < N; ++i) s += Y[i * k + a] * Y[i * k + b];
      X[a * k + b] = s;
    }
#define XX(a, b) X[(a) * k + (b)]
  if (k == 3) { /* :94-105, literal */
    const double rows[9][9] = {
        {XX(0, 0) + XX(0, 0), XX(0, 1), XX(0, 2), XX(1, 0), 0, 0, XX(2, 0), 0, 0},
        {XX(1, 0), XX(1, 1) + XX(0, 0), XX(1, 2), 0, XX(1, 0), 0, 0, XX(2, 0), 0},
        {XX(2, 0), XX(2, 1), XX(2, 2) + XX(0, 0), 0, 0, XX(1, 0), 0, 0, XX(2, 0)},
        {XX(0, 1), 0, 0, XX(0, 0) + XX(1, 1), XX(0, 1), XX(0, 2), XX(2, 1), 0, 0},
        {0, XX(0, 1), 0, XX(1, 0), XX(1, 1) + XX(1, 1), XX(1, 2), 0, XX(2, 1), 0},
        {0, 0, XX(0, 1), XX(2, 0), XX(2, 1), XX(2, 2) + XX(1, 1), 0, 0, XX(2, 1)},
        {XX(0, 2), 0, 0, XX(1, 2), 0, 0, XX(0, 0) + XX(2, 2), XX(0, 1), XX(0, 2)},
        {0, XX(0, 2), 0, 0, XX(1, 2), 0, XX(1, 0), XX(1, 1) + XX(2, 2), XX(1, 2)},
        {0, 0, XX(0, 2), 0, 0, XX(1, 2), XX(2, 0), XX(2, 1), XX(2, 2) + XX(2, 2)}};
    memcpy(A, rows, sizeof(rows));
  } else if (k == 2) { /* :106-110, literal -- entry [1][1] is X[0,1]+X[0,0] as written */
    const double rows[4][4] = {{XX(0, 0) + XX(0, 0), XX(0, 1), XX(0, 1), 0},
                               {XX(1, 0), XX(0, 1) + XX(0, 0), 0, XX(0, 1)},
                               {XX(0, 1), 0, XX(0, 0) + XX(1, 1), XX(0, 1)},
                               {0, XX(0, 1), XX(1, 0), XX(1, 1) + XX(1, 1)}};
    memcpy(A, rows, sizeof(rows));
  } else {
    return -2;
  }
#undef XX
  /* C = Y.T Z - Z.T Y (:111) */
  for (int a = 0; a < k; ++a)
    for (int b = 0; b < k; ++b) {
      double s1 = 0.0, s2 = 0.0;
      for (int i = 0; i < N; ++i) {
        s1 += Y[i * k + a] * Z[i * k + b];
        s2 += Z[i * k + a] * Y[i * k + b];
      }
      C[a * k + b] = s1 - s2;
    }
  int kk = k * k;
  for (int t = 0; t < kk; ++t) rhs[t] = C[t];
  if (lu_solve(A, rhs, kk) != 0) return -1; /* :112 Omega = solve(A, C.ravel()).reshape */
  /* :113 Z - Y.dot(Omega) */
  for (int i = 0; i < N; ++i)
    for (int b = 0; b < k; ++b) {
      double s = 0.0;
      for (int a = 0; a < k; ++a) s += Y[i * k + a] * rhs[a * k + b];
      out[i * k + b] = Z[i * k + b] - s;
    }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * solver context
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const double *D_goal, *omega, *psi_L, *psi_U;
  const int64_t *ii, *jj;
  int64_t n_inds;
  int N, k, use_limits;
  double *tmp; /* N*k scratch for ehess before projection */
  /* fixed-anchor variant (gik_o_rtr_solve_anchored): point-to-anchor terms, no quotient */
  const gik_o_anchor_terms *at;
} ctx_t;

/* point-to-anchor residual terms ("intended" obstacle semantics with anchors as
 * constants): same residuals as costs.py:80-93 with Y_j replaced by a fixed position:
 *   kind 1: (target - d)^2    kind 2: max(target - d, 0)^2    kind 3: max(d - target, 0)^2 */
static double anch_cost(const gik_o_anchor_terms *at, const double *Y, int k) {
  double f = 0.0;
  for (int t = 0; t < at->n; ++t) {
    double d = 0.0;
    for (int c = 0; c < k; ++c) {
      const double y = Y[at->node[t] * k + c] - at->pos[t * 3 + c];
      d += y * y;
    }
    double u = at->target[t] - d;
    if (at->kind[t] == 2) u = u > 0 ? u : 0.0;
    if (at->kind[t] == 3) u = u < 0 ? u : 0.0;
    f += u * u;
  }
  return f;
}
static void anch_grad(const gik_o_anchor_terms *at, const double *Y, int k, double *g) {
  for (int t = 0; t < at->n; ++t) {
    double y[3] = {0, 0, 0}, d = 0.0;
    for (int c = 0; c < k; ++c) {
      y[c] = Y[at->node[t] * k + c] - at->pos[t * 3 + c];
      d += y[c] * y[c];
    }
    double cc = d - at->target[t];                 /* egrad = 1/2 grad f, like costs.py:98-123 */
    if (at->kind[t] == 2 && !(at->target[t] - d > 0)) cc = 0.0;
    if (at->kind[t] == 3 && !(d - at->target[t] > 0)) cc = 0.0;
    for (int c = 0; c < k; ++c) g[at->node[t] * k + c] += 2 * cc * y[c];
  }
}
static void anch_hess(const gik_o_anchor_terms *at, const double *Y, const double *w, int k, double *h) {
  for (int t = 0; t < at->n; ++t) {
    double y[3] = {0, 0, 0}, d = 0.0, s = 0.0;
    for (int c = 0; c < k; ++c) {
      y[c] = Y[at->node[t] * k + c] - at->pos[t * 3 + c];
      d += y[c] * y[c];
      s += y[c] * w[at->node[t] * k + c];
    }
    int act = 1;
    if (at->kind[t] == 2) act = at->target[t] - d > 0;
    if (at->kind[t] == 3) act = d - at->target[t] > 0;
    if (!act) continue;
    const double cc = d - at->target[t];           /* costs.py:175-207 with w_j = 0 */
    for (int c = 0; c < k; ++c)
      h[at->node[t] * k + c] += 2 * (2 * s * y[c] + cc * w[at->node[t] * k + c]);
  }
}

static double ctx_cost(const ctx_t *c, const double *Y) {
  /* riemannian_solver.py:84-85 (K * jcost) / :131-132 (K * lcost), K = 1 */
  if (c->use_limits)
    return gik_o_lcost(Y, c->D_goal, c->omega, c->psi_L, c->psi_U, c->ii, c->jj, c->n_inds, c->N,
                       c->k) + (c->at ? anch_cost(c->at, Y, c->k) : 0.0);
  return gik_o_jcost(Y, c->D_goal, c->ii, c->jj, c->n_inds, c->N, c->k) +
         (c->at ? anch_cost(c->at, Y, c->k) : 0.0);
}

static void ctx_grad(const ctx_t *c, const double *Y, double *g) {
  /* problem.grad = egrad2rgrad(egrad) = egrad (fixed_rank_psd_sym.py:123-124) */
  if (c->use_limits)
    gik_o_lgrad(Y, c->D_goal, c->omega, c->psi_L, c->psi_U, c->ii, c->jj, c->n_inds, c->N, c->k,
                g);
  else
    gik_o_jgrad(Y, c->D_goal, c->ii, c->jj, c->n_inds, c->N, c->k, g);
  if (c->at) anch_grad(c->at, Y, c->k, g);
}

static void ctx_hess(const ctx_t *c, const double *Y, const double *w, double *out) {
  /* problem.hess(x, a) = ehess2rhess(x, egrad(x), ehess(x, a), a) = proj(x, ehess(x, a))
   * (fixed_rank_psd_sym.py:126-127).  The egrad(x) pymanopt evaluates and discards is omitted. */
  if (c->use_limits)
    gik_o_lhess(Y, w, c->D_goal, c->omega, c->psi_L, c->psi_U, c->ii, c->jj, c->n_inds, c->N,
                c->k, c->tmp);
  else
    gik_o_jhess(Y, w, c->D_goal, c->ii, c->jj, c->n_inds, c->N, c->k, c->tmp);
  if (c->at) { /* anchors fix the gauge: Euclidean space, no horizontal projection */
    anch_hess(c->at, Y, w, c->k, c->tmp);
    for (int t = 0; t < c->N * c->k; ++t) out[t] = c->tmp[t];
    return;
  }
  gik_o_proj(Y, c->tmp, c->N, c->k, out);
}

static double dot(const double *a, const double *b, int n) { /* PSDFixedRank.inner (:75-79) */
  double s = 0.0;
  for (int t = 0; t < n; ++t) s += a[t] * b[t];
  return s;
}

void gik_o_default_params(gik_o_params *p) {
  p->mingradnorm = 0.5 * 1e-9;
  p->maxiter = 3000;
  p->maxinner = 10000;
  p->mininner = 1;
  p->theta = 1.0;
  p->kappa = 0.1;
  p->rho_prime = 0.1;
  p->rho_regularization = 1e3;
  p->use_limits = 1;
}

enum {
  NEGATIVE_CURVATURE = 0,
  EXCEEDED_TR = 1,
  REACHED_TARGET_LINEAR = 2,
  REACHED_TARGET_SUPERLINEAR = 3,
  MAX_INNER_ITER = 4,
  MODEL_INCREASED = 5
}; /* trust_region.py:68-75 */

/* trust_region.py:436-599  _truncated_conjugate_gradient  (use_rand=False, precon=identity)
 * work: 5 vectors of n (r, delta, Hdelta, new_eta, new_Heta).  eta/Heta: out.           */
static int tcg(const ctx_t *c, const double *x, const double *fgradx, double *eta, double *Heta,
               double Delta, const gik_o_params *p, double *work, int *numit_out) {
  const int n = c->N * c->k;
  double *r = work, *delta = work + n, *Hdelta = work + 2 * n, *new_eta = work + 3 * n,
         *new_Heta = work + 4 * n;
  /* :444-448 eta = 0 (caller), Heta = zerovec, r = fgradx, e_Pe = 0 */
  for (int t = 0; t < n; ++t) {
    eta[t] = 0.0;
    Heta[t] = 0.0;
    r[t] = fgradx[t];
  }
  double e_Pe = 0.0;
  double r_r = dot(r, r, n); /* :455 */
  double norm_r = sqrt(r_r);
  const double norm_r0 = norm_r;
  /* :460-466 z = precon(x, r) = r ; z_r = <z, r> ; d_Pd = z_r */
  double z_r = dot(r, r, n);
  double d_Pd = z_r;
  for (int t = 0; t < n; ++t) delta[t] = -r[t]; /* :469 */
  double e_Pd = 0.0;                            /* :471 */
  double model_value = 0.0;                     /* :484-485 */
  int stop_tCG = MAX_INNER_ITER;                /* :491 */
  int j = 0;
  for (j = 0; j < p->maxinner; ++j) { /* :495 */
    ctx_hess(c, x, delta, Hdelta);    /* :497 */
    double d_Hd = dot(delta, Hdelta, n); /* :500 */
    double alpha = z_r / d_Hd;           /* :503 */
    double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd; /* :506 */
    if (d_Hd <= 0 || e_Pe_new >= Delta * Delta) {                        /* :509 */
      double tau = (-e_Pd + sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd; /* :514 */
      for (int t = 0; t < n; ++t) {
        eta[t] = eta[t] + tau * delta[t];    /* :516 */
        Heta[t] = Heta[t] + tau * Hdelta[t]; /* :521 */
      }
      stop_tCG = (d_Hd <= 0) ? NEGATIVE_CURVATURE : EXCEEDED_TR; /* :531-534 */
      break;
    }
    e_Pe = e_Pe_new; /* :537 */
    for (int t = 0; t < n; ++t) {
      new_eta[t] = eta[t] + alpha * delta[t];    /* :538 */
      new_Heta[t] = Heta[t] + alpha * Hdelta[t]; /* :542 */
    }
    /* :551 */
    double new_model_value = dot(new_eta, fgradx, n) + 0.5 * dot(new_eta, new_Heta, n);
    if (new_model_value >= model_value) { /* :552 */
      stop_tCG = MODEL_INCREASED;
      break;
    }
    for (int t = 0; t < n; ++t) { /* :556-557 */
      eta[t] = new_eta[t];
      Heta[t] = new_Heta[t];
    }
    model_value = new_model_value;                           /* :558 */
    for (int t = 0; t < n; ++t) r[t] = r[t] + alpha * Hdelta[t]; /* :561 */
    r_r = dot(r, r, n);                                      /* :564 */
    norm_r = sqrt(r_r);
    /* :572 */
    if (j >= p->mininner && norm_r <= norm_r0 * fmin(pow(norm_r0, p->theta), p->kappa)) {
      stop_tCG = (p->kappa < pow(norm_r0, p->theta)) ? REACHED_TARGET_LINEAR
                                                     : REACHED_TARGET_SUPERLINEAR; /* :574-577 */
      break;
    }
    double zold_rold = z_r; /* :587 */
    z_r = dot(r, r, n);     /* :589 (z = r) */
    double beta = z_r / zold_rold; /* :592 */
    for (int t = 0; t < n; ++t) delta[t] = -r[t] + beta * delta[t]; /* :593 */
    e_Pd = beta * (e_Pd + alpha * d_Pd); /* :596 */
    d_Pd = z_r + beta * beta * d_Pd;     /* :597 */
  }
  /* Python's `for j in range(maxinner)` leaves j = maxinner-1 when the loop is exhausted */
  if (j >= p->maxinner) j = p->maxinner - 1;
  *numit_out = j; /* :599 */
  return stop_tCG;
}

/* trust_region.py:112-434   TrustRegions.solve */
int gik_o_rtr_solve(double *Y, const double *D_goal, const double *omega, const double *psi_L,
                    const double *psi_U, const int64_t *ii, const int64_t *jj, int64_t n_inds,
                    int N, int k, const gik_o_params *p, gik_o_result *res, gik_o_traj *traj) {
  return gik_o_rtr_solve_anchored(Y, D_goal, omega, psi_L, psi_U, ii, jj, n_inds, N, k, 0, p, res, traj);
}

/* The same trust-region solver on the fixed-anchor formulation: Y holds the FREE nodes only, `at`
 * the point-to-anchor terms; the metric is Euclidean (at == NULL: the reference's quotient). */
int gik_o_rtr_solve_anchored(double *Y, const double *D_goal, const double *omega, const double *psi_L,
                             const double *psi_U, const int64_t *ii, const int64_t *jj, int64_t n_inds,
                             int N, int k, const gik_o_anchor_terms *at, const gik_o_params *p,
                             gik_o_result *res, gik_o_traj *traj) {
  const int n = N * k;
  double *buf = (double *)malloc(sizeof(double) * (size_t)n * 10);
  if (!buf) return -1;
  double *fgradx = buf, *eta = buf + n, *Heta = buf + 2 * n, *x_prop = buf + 3 * n,
         *work = buf + 4 * n, *tmp = buf + 9 * n;
  ctx_t c = {D_goal, omega, psi_L, psi_U, ii, jj, n_inds, N, k, p->use_limits, tmp, at};
  double *x = Y;

  const double Delta_bar = 10.0 + k;     /* :128-131 typicaldist (fixed_rank_psd_sym.py:71-73) */
  const double Delta0 = Delta_bar / 8.0; /* :134-135 */
  int kiter = 0;                          /* :156 */
  double fx = ctx_cost(&c, x);            /* :159 */
  ctx_grad(&c, x, fgradx);                /* :160 */
  double norm_grad = sqrt(dot(fgradx, fgradx, n)); /* :161 man.norm */
  double Delta = Delta0;                  /* :164 */
  int inner_total = 0, stop = 1;
  if (traj) traj->len = 0;

  for (;;) { /* :179 */
    int numit = 0;
    int stop_inner = tcg(&c, x, fgradx, eta, Heta, Delta, p, work, &numit); /* :196-206 */
    inner_total += numit + 1;
    if (traj && traj->len < traj->cap) {
      int q = traj->len;
      if (traj->Delta) traj->Delta[q] = Delta;
      if (traj->numit) traj->numit[q] = numit;
      if (traj->stop) traj->stop[q] = stop_inner;
      if (traj->f_before) traj->f_before[q] = fx;
    }
    for (int t = 0; t < n; ++t) x_prop[t] = x[t] + eta[t]; /* :248 retr (:137-138) */
    double fx_prop = ctx_cost(&c, x_prop);                 /* :251 */
    double rhonum = fx - fx_prop;                          /* :255 */
    double rhoden = -dot(fgradx, eta, n) - 0.5 * dot(eta, Heta, n); /* :256 */
    double rho_reg = fmax(1.0, fabs(fx)) * 2.220446049250313e-16 * p->rho_regularization; /* :287 */
    rhonum = rhonum + rho_reg; /* :288 */
    rhoden = rhoden + rho_reg; /* :289 */
    int model_decreased = rhoden >= 0; /* :311 */
    double rho = rhonum / rhoden;      /* :317 (IEEE: inf/nan instead of ZeroDivisionError) */
    if (rho < 1.0 / 4 || !model_decreased || isnan(rho)) { /* :336 */
      Delta = Delta / 4;                                    /* :338 */
    } else if (rho > 3.0 / 4 && (stop_inner == NEGATIVE_CURVATURE || stop_inner == EXCEEDED_TR)) {
      Delta = fmin(2 * Delta, Delta_bar); /* :357-361 */
    }
    int accept = 0;
    if (model_decreased && rho > p->rho_prime) { /* :382 */
      accept = 1;
      for (int t = 0; t < n; ++t) x[t] = x_prop[t]; /* :385 */
      fx = fx_prop;                                 /* :386 */
      ctx_grad(&c, x, fgradx);                      /* :387 */
      norm_grad = sqrt(dot(fgradx, fgradx, n));     /* :388 */
    }
    kiter = kiter + 1; /* :394 */
    if (traj && traj->len < traj->cap) {
      int q = traj->len;
      if (traj->gradnorm_after) traj->gradnorm_after[q] = norm_grad;
      if (traj->accept) traj->accept[q] = accept;
      traj->len++;
    }
    /* :414-416 _check_stopping_criterion (pymanopt 0.2.5 Solver): maxtime (disabled here --
     * wall clock is not deterministic), then iter >= maxiter, then gradnorm < mingradnorm */
    if (kiter >= p->maxiter) {
      stop = 1;
      break;
    }
    if (norm_grad < p->mingradnorm) {
      stop = 0;
      break;
    }
  }
  if (res) {
    res->f = fx;
    res->gradnorm = norm_grad;
    res->iterations = kiter;
    res->inner_total = inner_total;
    res->stop = stop;
  }
  free(buf);
  return 0;
}

int gik_o_rtr_solve_batch(double *Y, const double *D_goal, const double *omega,
                          const double *psi_L, const double *psi_U, const int64_t *ii,
                          const int64_t *jj, int64_t n_inds, int N, int k, int B,
                          const gik_o_params *p, gik_o_result *res, int nthreads) {
  int rc = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int b = 0; b < B; ++b) {
    int r = gik_o_rtr_solve(Y + (size_t)b * N * k, D_goal + (size_t)b * N * N, omega, psi_L, psi_U,
                            ii, jj, n_inds, N, k, p, res ? res + b : 0, 0);
    if (r != 0) rc = r;
  }
  return rc;
}

/* ------------------------------------------------------------------------------------------
 * Riemannian conjugate gradients: the reference's params["solver"] = "ConjugateGradient"
 * (graphik/solvers/riemannian_solver.py:51-59) is pymanopt 0.2.5's ConjugateGradient
 * (pymanopt/solvers/conjugate_gradient.py, a port of Manopt's conjugategradient.m) with
 * LineSearchAdaptive (pymanopt/solvers/linesearch.py) -- a THIRD-PARTY dependency that is not under
 * /root/reference (setup.py:20 pins pymanopt == 0.2.5).  Restated here from the published
 * algorithm; manifold methods are the reference's (fixed_rank_psd_sym.py): inner/norm = Frobenius,
 * retr(Y, U) = Y + U, transp(Y, Z, U) = proj(Z, U), egrad2rgrad = identity, precon = identity.
 * Pinned by tests/golden/cg.npz (tools/capture_golden_cg.py: the reference's solve() driving the
 * same restatement in Python, tools/ref_shims/pymanopt/solvers).
 * ------------------------------------------------------------------------------------------ */
void gik_o_cg_default_params(gik_o_cg_params *p) {
  p->mingradnorm = 1e-9;   /* riemannian_solver.py:53 */
  p->maxiter = 100000;     /* :55  (10e4)             */
  p->minstepsize = 1e-10;  /* :56                     */
  p->orth_value = 10e10;   /* :57                     */
  p->beta_type = 3;        /* :58  BetaTypes[3] = HagerZhang */
  p->use_limits = 1;
  /* LineSearchAdaptive defaults */
  p->ls_contraction = 0.5;
  p->ls_suff_decr = 0.5;
  p->ls_maxiter = 10;
  p->ls_initial_stepsize = 1.0;
}

int gik_o_cg_solve(double *Y, const double *D_goal, const double *omega, const double *psi_L,
                   const double *psi_U, const int64_t *ii, const int64_t *jj, int64_t n_inds,
                   int N, int k, const gik_o_cg_params *p, gik_o_result *res, gik_o_cg_traj *traj) {
  const int n = N * k;
  double *buf = (double *)malloc(sizeof(double) * (size_t)n * 9);
  if (!buf) return -1;
  double *grad = buf, *desc = buf + n, *newx = buf + 2 * n, *newgrad = buf + 3 * n,
         *oldgrad = buf + 4 * n, *tdesc = buf + 5 * n, *diff = buf + 6 * n, *tmp = buf + 7 * n,
         *trial = buf + 8 * n;
  ctx_t c = {D_goal, omega, psi_L, psi_U, ii, jj, n_inds, N, k, p->use_limits, tmp, 0};
  double *x = Y;
  int iter = 0, stop = 1, costevals_total = 0;
  double stepsize = NAN;
  double cost = ctx_cost(&c, x);
  ctx_grad(&c, x, grad);
  double gradnorm = sqrt(dot(grad, grad, n));      /* man.norm */
  double gradPgrad = dot(grad, grad, n);            /* precon = identity: Pgrad = grad */
  for (int t = 0; t < n; ++t) desc[t] = -grad[t];
  double oldalpha = 0.0;
  int have_oldalpha = 0;                            /* LineSearchAdaptive._oldalpha = None */
  if (traj) traj->len = 0;
  for (;;) {
    /* _check_stopping_criterion(time0, gradnorm=gradnorm, iter=iter + 1, stepsize=stepsize):
     * maxtime (not reproduced), iter >= maxiter, gradnorm < mingradnorm, stepsize < minstepsize */
    if (iter + 1 >= p->maxiter) { stop = 1; break; }
    if (gradnorm < p->mingradnorm) { stop = 0; break; }
    if (stepsize < p->minstepsize) { stop = 3; break; }      /* NaN compares false */
    if (isnan(cost) || isnan(gradnorm)) { stop = 2; break; } /* (pymanopt would spin to maxiter) */
    double df0 = dot(grad, desc, n);
    if (df0 >= 0) {                                 /* not a descent direction: restart */
      for (int t = 0; t < n; ++t) desc[t] = -grad[t];
      df0 = -gradPgrad;
    }
    /* ---- LineSearchAdaptive.search(objective, man, x, d, f0, df0) ---- */
    const double norm_d = sqrt(dot(desc, desc, n));
    double alpha = have_oldalpha ? oldalpha : p->ls_initial_stepsize / norm_d;
    for (int t = 0; t < n; ++t) trial[t] = alpha * desc[t];
    for (int t = 0; t < n; ++t) newx[t] = x[t] + trial[t];   /* man.retr(x, alpha * d) */
    double newf = ctx_cost(&c, newx);
    int cost_evaluations = 1;
    while (newf > cost + p->ls_suff_decr * alpha * df0 && cost_evaluations <= p->ls_maxiter) {
      alpha *= p->ls_contraction;
      for (int t = 0; t < n; ++t) trial[t] = alpha * desc[t];
      for (int t = 0; t < n; ++t) newx[t] = x[t] + trial[t];
      newf = ctx_cost(&c, newx);
      cost_evaluations += 1;
    }
    if (newf > cost) {
      alpha = 0;
      for (int t = 0; t < n; ++t) newx[t] = x[t];
    }
    stepsize = alpha * norm_d;
    oldalpha = (cost_evaluations == 2) ? alpha : 2 * alpha;
    have_oldalpha = 1;
    costevals_total += cost_evaluations;
    if (traj && traj->len < traj->cap) {
      int q = traj->len++;
      if (traj->f) traj->f[q] = cost;
      if (traj->gradnorm) traj->gradnorm[q] = gradnorm;
      if (traj->stepsize) traj->stepsize[q] = stepsize;
      if (traj->costevals) traj->costevals[q] = cost_evaluations;
    }
    /* ---- new point ---- */
    const double newcost = ctx_cost(&c, newx);
    ctx_grad(&c, newx, newgrad);
    const double newgradnorm = sqrt(dot(newgrad, newgrad, n));
    const double newgradPnewgrad = dot(newgrad, newgrad, n);
    gik_o_proj(newx, grad, N, k, oldgrad);           /* oldgrad = man.transp(x, newx, grad) */
    const double orth_grads = dot(oldgrad, newgrad, n) / newgradPnewgrad;
    if (fabs(orth_grads) >= p->orth_value) {         /* Powell restart */
      for (int t = 0; t < n; ++t) desc[t] = -newgrad[t];
    } else {
      gik_o_proj(newx, desc, N, k, tdesc);           /* desc_dir = man.transp(x, newx, desc_dir) */
      double beta;
      if (p->beta_type == 0) {                       /* FletcherReeves */
        beta = newgradPnewgrad / gradPgrad;
      } else if (p->beta_type == 1) {                /* PolakRibiere */
        for (int t = 0; t < n; ++t) diff[t] = newgrad[t] - oldgrad[t];
        beta = fmax(0.0, dot(newgrad, diff, n) / gradPgrad);
      } else if (p->beta_type == 2) {                /* HestenesStiefel */
        for (int t = 0; t < n; ++t) diff[t] = newgrad[t] - oldgrad[t];
        const double den = dot(diff, tdesc, n);
        /* pymanopt: `try: beta = max(0, ip_diff / inner(diff, desc_dir)) except ZeroDivisionError:
         * beta = 1` -- numpy scalars never raise, so den == 0 gives inf / nan and Python's
         * max(0, .) keeps inf and maps -inf and nan to 0: exactly fmax */
        beta = fmax(0.0, dot(newgrad, diff, n) / den);
      } else {                                       /* HagerZhang */
        for (int t = 0; t < n; ++t) diff[t] = newgrad[t] - oldgrad[t];
        /* Poldgrad = man.transp(x, newx, Pgrad) = oldgrad; Pdiff = Pnewgrad - Poldgrad = diff */
        const double deno = dot(diff, tdesc, n);
        double numo = dot(diff, newgrad, n);
        numo -= 2 * dot(diff, diff, n) * dot(tdesc, newgrad, n) / deno;
        beta = numo / deno;
        const double desc_dir_norm = sqrt(dot(tdesc, tdesc, n));
        const double eta_HZ = -1 / (desc_dir_norm * fmin(0.01, gradnorm));
        beta = (eta_HZ > beta) ? eta_HZ : beta; /* Python max(beta, eta_HZ): a NaN beta stays NaN */
      }
      for (int t = 0; t < n; ++t) desc[t] = -newgrad[t] + beta * tdesc[t];
    }
    for (int t = 0; t < n; ++t) x[t] = newx[t];
    cost = newcost;
    for (int t = 0; t < n; ++t) grad[t] = newgrad[t];
    gradnorm = newgradnorm;
    gradPgrad = newgradPnewgrad;
    iter += 1;
  }
  if (res) {
    res->f = cost;
    res->gradnorm = gradnorm;
    res->iterations = iter;
    res->inner_total = costevals_total;
    res->stop = stop;
  }
  free(buf);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * graphik/utils/dgp.py:192-231   bound_smoothing
 * The reference runs networkx all-pairs Bellman-Ford on the doubled graph H (u, u' = "us"):
 *   u->u' 0 ; u->v', v->u' -LOWER ; u<->v UPPER ; u'<->v' UPPER            (:203-211)
 * There are no negative cycles (LOWER <= UPPER), so shortest-path lengths are unique and a
 * Floyd-Warshall sweep over the same 2N-node graph gives the same values up to the association
 * order of the fp64 path sums.
 * ------------------------------------------------------------------------------------------ */
void gik_o_bound_smoothing(const double *lower, const double *upper, int N, double *lb,
                           double *ub) {
  const int M = 2 * N;
  double *sp = (double *)malloc(sizeof(double) * (size_t)M * M);
  for (int a = 0; a < M * M; ++a) sp[a] = INFINITY;
  for (int a = 0; a < M; ++a) sp[IDX(a, a, M)] = 0.0;
  for (int u = 0; u < N; ++u)
    for (int v = 0; v < N; ++v) {
      if (u == v) continue;
      double lo = lower[IDX(u, v, N)], up = upper[IDX(u, v, N)];
      if (isnan(lo) || isnan(up)) continue;
      sp[IDX(u, u + N, M)] = fmin(sp[IDX(u, u + N, M)], 0.0);
      sp[IDX(v, v + N, M)] = fmin(sp[IDX(v, v + N, M)], 0.0);
      sp[IDX(u, v + N, M)] = fmin(sp[IDX(u, v + N, M)], -lo);
      sp[IDX(v, u + N, M)] = fmin(sp[IDX(v, u + N, M)], -lo);
      sp[IDX(u, v, M)] = fmin(sp[IDX(u, v, M)], up);
      sp[IDX(v, u, M)] = fmin(sp[IDX(v, u, M)], up);
      sp[IDX(u + N, v + N, M)] = fmin(sp[IDX(u + N, v + N, M)], up);
      sp[IDX(v + N, u + N, M)] = fmin(sp[IDX(v + N, u + N, M)], up);
    }
  for (int m = 0; m < M; ++m)
    for (int a = 0; a < M; ++a) {
      double am = sp[IDX(a, m, M)];
      if (am == INFINITY) continue;
      for (int b = 0; b < M; ++b) {
        double cand = am + sp[IDX(m, b, M)];
        if (cand < sp[IDX(a, b, M)]) sp[IDX(a, b, M)] = cand;
      }
    }
  for (int u = 0; u < N; ++u)
    for (int v = 0; v < N; ++v) {
      double s = sp[IDX(u, v + N, M)];
      lb[IDX(u, v, N)] = (s < 0) ? -s : 0.0; /* :222-225 */
      ub[IDX(u, v, N)] = sp[IDX(u, v, M)];   /* :227 */
    }
  free(sp);
}
