/*
 * oracle/gik_oracle.h -- CPU restatement (plain C, fp64) of GraphIK's RiemannianSolver hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under graphik_amd/ may include, link or load this; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it -- as the checker and
 * the CPU baseline, never as the product path.
 *
 * Each function cites the reference lines it restates (paths relative to the GraphIK repo).
 * Pinning: tests/test_oracle_golden.py checks every function against the .npz files of tests/golden, which
 * were produced by running the reference itself (tools/capture_golden.py).
 */
#ifndef GIK_ORACLE_H
#define GIK_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- cost / gradient / Hessian-vector loops: graphik/solvers/costs.py ------------------- */
/* Dense N x N row-major matrices + index pairs `inds` exactly as the numba signatures
 * f8(f8[:,:], f8[:,:], UniTuple(u8[:],2)) etc. take them.  Y, w, out: N x k row-major.     */
double gik_o_jcost(const double *Y, const double *D_goal, const int64_t *ii, const int64_t *jj,
                   int64_t n_inds, int N, int k);                     /* costs.py:8-16   */
void gik_o_jgrad(const double *Y, const double *D_goal, const int64_t *ii, const int64_t *jj,
                 int64_t n_inds, int N, int k, double *grad);         /* costs.py:20-35  */
void gik_o_jhess(const double *Y, const double *w, const double *D_goal, const int64_t *ii,
                 const int64_t *jj, int64_t n_inds, int N, int k, double *hess); /* :39-58 */
double gik_o_jcost_and_grad(const double *Y, const double *D_goal, const int64_t *ii,
                            const int64_t *jj, int64_t n_inds, int N, int k, double *grad);
double gik_o_lcost_and_grad(const double *Y, const double *D_goal, const double *omega,
                            const double *psi_L, const double *psi_U, const int64_t *ii,
                            const int64_t *jj, int64_t n_inds, int N, int k, double *grad);
double gik_o_lcost(const double *Y, const double *D_goal, const double *omega,
                   const double *psi_L, const double *psi_U, const int64_t *ii,
                   const int64_t *jj, int64_t n_inds, int N, int k);  /* costs.py:80-93  */
void gik_o_lgrad(const double *Y, const double *D_goal, const double *omega, const double *psi_L,
                 const double *psi_U, const int64_t *ii, const int64_t *jj, int64_t n_inds,
                 int N, int k, double *grad);                         /* costs.py:98-123 */
void gik_o_lhess(const double *Y, const double *w, const double *D_goal, const double *omega,
                 const double *psi_L, const double *psi_U, const int64_t *ii, const int64_t *jj,
                 int64_t n_inds, int N, int k, double *hess);         /* costs.py:175-207 */

/* ---- manifold: graphik/utils/manifolds/fixed_rank_psd_sym.py ----------------------------- */
/* proj (:91-113): literal 9x9 (k=3) / 4x4 (k=2, including the [1,1] entry as written)
 * system solved by LU with partial pivoting (np.linalg.solve == LAPACK dgesv).            */
int gik_o_proj(const double *Y, const double *Z, int N, int k, double *out);

/* ---- trust-region solver: graphik/solvers/trust_region.py -------------------------------- */
typedef struct {
  double mingradnorm;        /* riemannian_solver.py:45  (0.5e-9)                      */
  int maxiter;               /* riemannian_solver.py:47  (3000)                        */
  int maxinner;              /* trust_region.py:118      (10000)                       */
  int mininner;              /* trust_region.py:116      (1)                           */
  double theta, kappa;       /* riemannian_solver.py:48-49 (1.0, 0.1)                  */
  double rho_prime;          /* trust_region.py:90       (0.1)                         */
  double rho_regularization; /* trust_region.py:92       (1e3)                         */
  int use_limits;            /* create_cost_limits (1) vs create_cost (0)              */
} gik_o_params;

void gik_o_default_params(gik_o_params *p);

typedef struct {
  double f;          /* final cost                              */
  double gradnorm;   /* final ||grad||_F                        */
  int iterations;    /* outer iterations k                      */
  int inner_total;   /* total Hessian-vector products           */
  int stop;          /* 0 gradnorm, 1 maxiter                   */
} gik_o_result;

/* Optional per-outer-iteration trace (arrays of length traj_cap, may be NULL). */
typedef struct {
  int cap;
  int len;
  double *Delta;          /* radius handed to tCG            */
  int *numit;             /* tCG returned j                  */
  int *stop;              /* tCG stop reason 0..5            */
  double *f_before;       /* f(x) before the step            */
  double *gradnorm_after; /* ||grad|| after accept/reject    */
  int *accept;
} gik_o_traj;

/* TrustRegions.solve (trust_region.py:112-434) + _truncated_conjugate_gradient (:436-599)
 * on the cost of create_cost_limits / create_cost (riemannian_solver.py:77-176, loop form),
 * manifold PSDFixedRank(N,k).  Y is N x k: in = Y_init, out = Y_sol.                    */
int gik_o_rtr_solve(double *Y, const double *D_goal, const double *omega, const double *psi_L,
                    const double *psi_U, const int64_t *ii, const int64_t *jj, int64_t n_inds,
                    int N, int k, const gik_o_params *p, gik_o_result *res, gik_o_traj *traj);

/* Fixed-anchor formulation (opt-in "intended" obstacle semantics): nodes with a
 * known position are constants, not rows of Y, and the robot<->obstacle lower-bound hinges that
 * graph_base.py:205-211 means to create exist.  n point-to-anchor terms: free node index, anchor
 * position, squared target, kind (1 equality, 2 lower hinge, 3 upper hinge).  No reference
 * counterpart to pin it to: it is the twin the HIP anchored kernels are tested against.        */
typedef struct {
  int n;
  const int *node;        /* [n] */
  const double *pos;      /* [n][3] */
  const double *target;   /* [n] */
  const int *kind;        /* [n] */
} gik_o_anchor_terms;
int gik_o_rtr_solve_anchored(double *Y, const double *D_goal, const double *omega, const double *psi_L,
                             const double *psi_U, const int64_t *ii, const int64_t *jj, int64_t n_inds,
                             int N, int k, const gik_o_anchor_terms *at, const gik_o_params *p,
                             gik_o_result *res, gik_o_traj *traj);

/* Batch driver (OpenMP over problems) -- used as bench.py's cpu_baseline.  D_goal is per
 * problem (B x N x N); omega/psi shared.  Y: B x N x k in/out.                           */
int gik_o_rtr_solve_batch(double *Y, const double *D_goal, const double *omega,
                          const double *psi_L, const double *psi_U, const int64_t *ii,
                          const int64_t *jj, int64_t n_inds, int N, int k, int B,
                          const gik_o_params *p, gik_o_result *res, int nthreads);

/* ---- Riemannian conjugate gradients: pymanopt 0.2.5 ConjugateGradient + LineSearchAdaptive as
 * configured by graphik/solvers/riemannian_solver.py:51-59 (third-party, restated; see the .c) ---- */
typedef struct {
  double mingradnorm;   /* 1e-9   */
  int maxiter;          /* 100000 */
  double minstepsize;   /* 1e-10  */
  double orth_value;    /* 10e10  */
  int beta_type;        /* 0 FletcherReeves, 1 PolakRibiere, 2 HestenesStiefel, 3 HagerZhang */
  int use_limits;
  double ls_contraction, ls_suff_decr, ls_initial_stepsize;   /* 0.5, 0.5, 1 */
  int ls_maxiter;                                             /* 10 */
} gik_o_cg_params;
void gik_o_cg_default_params(gik_o_cg_params *p);
typedef struct {
  int cap, len;
  double *f;          /* cost before step q                         */
  double *gradnorm;   /* ||grad|| before step q                     */
  double *stepsize;   /* alpha * ||d|| the line search returned     */
  int *costevals;     /* cost evaluations of that line search       */
} gik_o_cg_traj;
/* res->inner_total = cost evaluations of all line searches; res->stop: 0 gradnorm, 1 maxiter,
 * 2 NaN, 3 minstepsize */
int gik_o_cg_solve(double *Y, const double *D_goal, const double *omega, const double *psi_L,
                   const double *psi_U, const int64_t *ii, const int64_t *jj, int64_t n_inds,
                   int N, int k, const gik_o_cg_params *p, gik_o_result *res, gik_o_cg_traj *traj);

/* ---- pre-processing: graphik/utils/dgp.py ------------------------------------------------- */
/* bound_smoothing (dgp.py:192-231): all-pairs shortest paths on the doubled graph.  lower /
 * upper are N x N with NaN where the goal graph has no edge.  lb, ub: N x N out.         */
void gik_o_bound_smoothing(const double *lower, const double *upper, int N, double *lb,
                           double *ub);

#ifdef __cplusplus
}
#endif
#endif
