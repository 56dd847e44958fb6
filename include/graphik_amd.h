/*
 * include/graphik_amd.h -- C ABI of the MI355X-native batched distance-geometric IK engine.
 *
 * This is the drop-in boundary for GraphIK's RiemannianSolver hot path.  The reference has no
 * FFI of its own (it is pure Python); the interfaces these entry points replace are
 *
 *   (1) the numba-AOT extension `graphik.solvers.costgrd`
 *       (graphik/solvers/costs.py:5-207, imported at graphik/solvers/riemannian_solver.py:18-21):
 *           jcost/jgrad/jhess, lcost/lgrad/lhess            -> gik_cost / gik_grad / gik_hess
 *           jcost_and_grad (:61-77), lcost_and_grad (:126-169)       -> gik_cost_and_grad
 *   (2) the numba-jitted manifold methods
 *       (graphik/utils/manifolds/fixed_rank_psd_sym.py:75-113): proj            -> gik_proj
 *   (3) pymanopt-style `TrustRegions.solve(problem, x)` as called from
 *       RiemannianSolver.solve (graphik/solvers/riemannian_solver.py:178-218,
 *       graphik/solvers/trust_region.py:112-599)                         -> gik_solve_batch
 *   (4) the per-goal pre/post-processing of solve_with_riemannian
 *       (riemannian_solver.py:220-234; dgp.py:42-65,150-183,192-231;
 *        graph_revolute.py:243-318, graph_planar.py:136-176)               -> gik_ik_batch
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  No torch / numpy types.
 *   - All `d_*` pointers are DEVICE pointers (HBM) owned by the caller (e.g. the data_ptr() of
 *     a PyTorch-ROCm tensor); the library never allocates inside a batch call except for the
 *     handle's own persistent workspace, which grows monotonically with the largest B seen.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are
 *     asynchronous with respect to the host; synchronise the stream before reading results.
 *   - Every function returns 0 on success, <0 on error; gik_last_error() returns a
 *     thread-local message for the last failure.
 *   - All floating point data is fp64 (IEEE double); vectors are N x k row-major like the
 *     reference's numpy arrays.
 */
#ifndef GRAPHIK_AMD_H
#define GRAPHIK_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GIK_ABI_VERSION 6

/* Residual-term kinds: one "term" per (index pair, kind) exactly as the loops of
 * costs.py:80-207 visit them: equality (omega != 0), lower hinge (psi_L != 0), upper hinge
 * (psi_U != 0).                                                                           */
enum { GIK_TERM_EQ = 1, GIK_TERM_LOWER = 2, GIK_TERM_UPPER = 3 };

/* Problem-graph template: everything that does not depend on the goal pose.  Mirrors the
 * arguments create_cost_limits() closes over (riemannian_solver.py:121-128): the index pairs
 * `inds` (row-major upper triangle), and which of omega / psi_L / psi_U is set on each.   */
typedef struct {
  int32_t abi_version;   /* GIK_ABI_VERSION                                               */
  int32_t N;             /* number of graph nodes (rows of Y): 2 .. 255; beyond 128 only k = 3, TrustRegions,
                            theta = 1 graphs with at most 256 terms outside a rigid clique (obstacle scenes)  */
  int32_t k;             /* embedding dimension: 3 (revolute) or 2 (planar)               */
  int32_t n_terms;       /* number of residual terms T                                    */
  const int32_t *term_i; /* [T] first node index  (i < j)                                 */
  const int32_t *term_j; /* [T] second node index                                         */
  const int32_t *term_kind; /* [T] GIK_TERM_*; terms sorted by (i, j, kind)               */
  /* trust-region parameters (riemannian_solver.py:44-50, trust_region.py:85-121)          */
  double mingradnorm;    /* 0.5e-9                                                        */
  int32_t maxiter;       /* 3000                                                          */
  int32_t maxinner;      /* 10000                                                         */
  int32_t mininner;      /* 1                                                             */
  double theta;          /* 1.0                                                           */
  double kappa;          /* 0.1                                                           */
  double rho_prime;      /* 0.1                                                           */
  double rho_regularization; /* 1e3                                                       */
  int32_t planar_proj_exact; /* 0: reproduce fixed_rank_psd_sym.py:107-110 literally (k=2) */
  int32_t force_block_path;  /* graphs with N*k > 64 (or a node busier than any wavefront variant) leave the
                                one-unknown-per-lane kernel; 0: automatic (k = 3, N*k > 64, TrustRegions: the
                                node-per-lane wavefront kernel, else the workgroup kernels), 1: the
                                workgroup-per-problem kernels even if N*k <= 64, 2: the node-per-lane kernel
                                (k = 3, TrustRegions, theta = 1, <= 256 terms outside a rigid clique)        */
  /* scheduling knobs, fixed for the life of the handle (results never depend on them; the
   * environment variables GIK_WAVES_PER_CU / GIK_SLICE / GIK_DBG are read ONCE, at
   * gik_template_create, as developer overrides of these fields)                             */
  int32_t waves_per_cu;      /* persistent solve waves (workgroups) per CU; 0 = automatic      */
  int32_t slice_outer_its;   /* time slice in outer iterations: a problem yields its slot to whatever
                                waits after that many; -1 = default (256; node-per-lane kernel 192; on the
                                wavefront kernel only for batches beyond the resident waves), 0 = off */
  int32_t debug_flags;       /* developer flags (gik_kernels.hip.h: SolveArgs::dbg); 16 = rerun tCG
                                after a rejected step instead of resuming from the checkpoint;
                                workgroup path: 64 = closed form for rigid cliques from 4 nodes
                                up (default: 16 nodes); 128 / 256 = older spellings of
                                clique_closed_form = GIK_CLIQUE_OFF / GIK_CLIQUE_DENSE; 2048 = node-per-lane
                                kernel with one wavefront per problem (two nodes per lane); 8192 = planar graphs: one
                                problem per wavefront instead of four, 16384 = four at any batch size (default: from
                                12 problems per CU on); 512 = neither round-robin
                                slicing nor tail spreading on the wavefront kernel, 1024 = no slicing
                                (scheduling measures of large batches, bit-neutral: tests compare) */
  /* which of the reference's two solvers gik_solve_batch runs (riemannian_solver.py:40-65):
   * GIK_SOLVER_TRUST_REGIONS (default) or GIK_SOLVER_CONJUGATE_GRADIENT = pymanopt 0.2.5
   * ConjugateGradient + LineSearchAdaptive as configured at :51-59.  The CG defaults of
   * mingradnorm / maxiter differ (1e-9 / 100000): gik_default_cg_params() sets them.           */
  int32_t solver;
  double cg_minstepsize;     /* 1e-10                                                         */
  double cg_orth_value;      /* 10e10                                                         */
  int32_t cg_beta_type;      /* 0 FletcherReeves, 1 PolakRibiere, 2 HestenesStiefel, 3 HagerZhang
                                (the reference passes BetaTypes[3])                           */
  /* Workgroup-per-problem path only (N*k > 64): a rigid clique -- >= 16 nodes every pair of which
   * is tied by an equality term, i.e. the anchors of a scene with obstacles (graph_base.py:182-199)
   * -- is taken out of the per-term loops and its share of lhess (costs.py:175-207) is evaluated
   * in closed form from 18 moments (+ 9 when the clique's targets are distances of points).  Same
   * function, different SUMMATION ORDER: Hessian products agree with the reference's edge loop to
   * round-off (<= 2e-15 relative), not bit for bit, so trust-region iteration counts near an
   * accept / reject threshold can differ from a run with GIK_CLIQUE_OFF (which sums the terms in
   * the reference's order).  gik_stats.flags bit 0 and gik_template_get_info report what ran.   */
  int32_t clique_closed_form; /* GIK_CLIQUE_AUTO (default) | GIK_CLIQUE_OFF | GIK_CLIQUE_DENSE     */
  /* One-unknown-per-lane wavefront kernel, k = 3 (graphs of N * k <= 64 unknowns: the arms): how lhess
   * (costs.py:175-207) is rendered.
   *   GIK_HESS_PER_EDGE: the reference's arithmetic -- per edge  s = y . (W_i - W_j),  t = 2 s a y + c w,  +t to one end
   *     and -t to the other (costs.py:186-203), the form every other kernel of the library uses.  The three lanes of a
   *     node split its term list, each evaluates its terms completely from whole rows and the three partial vectors are
   *     added (gik_wave_strict.hip.h).  Exists for TrustRegions (any theta), free-free (not anchored) graphs.
   *     Against the CPU oracle from the same start points (8192 random KUKA goals): Hessian products +1.7 %, share of
   *     goals at maxiter 668 / 677, p90 of the outer iterations 0.99 x; 7-DOF end configurations 2.9e-3 rad from the
   *     reference in the median (the reference's own two code paths: 3.0e-3).
   *   GIK_HESS_COLUMN: rows of the 3 x 3 blocks 2 a y y^T + c I, cached per accepted point, times the neighbour's
   *     entries.  s is never formed: the Gauss-Newton part's round-off leaves range(J^T) and truncated CG needs 5-8 %
   *     more Hessian products than the reference's arithmetic (7-DOF end configurations 8.4e-3 rad from the reference,
   *     2.8 x the reference pair's own spread).  Until round 5 the default; since round 6 NOT faster either (c2 34.5 k
   *     against 35.3 k solves/s, c4 125.6 k against 136.0 k: DESIGN.md 4.1, 6) -- kept as the form of the ConjugateGradient
   *     (which takes no Hessian products), anchored and planar wavefront kernels, and as an explicit choice for comparisons.
   *   GIK_HESS_AUTO (default): PER_EDGE where that kernel exists (3-D, TrustRegions, not anchored), else
   *     COLUMN.  An explicit GIK_HESS_PER_EDGE on a wavefront template without such a kernel is refused.
   * Graphs on the workgroup / node-per-lane kernels form s per edge whatever this field says (gik_template_info
   * reports GIK_HESS_PER_EDGE for them).                                                                          */
  int32_t hessian_form;
} gik_template_desc;
enum { GIK_HESS_COLUMN = 0, GIK_HESS_PER_EDGE = 1, GIK_HESS_AUTO = 2 };

enum { GIK_SOLVER_TRUST_REGIONS = 0, GIK_SOLVER_CONJUGATE_GRADIENT = 1 };
enum {
  GIK_CLIQUE_AUTO = 0,  /* closed form; (D w) by moments when the targets are Euclidean (checked per problem) */
  GIK_CLIQUE_OFF = 1,   /* every term in the per-term loops, the reference's summation order               */
  GIK_CLIQUE_DENSE = 2  /* closed form with the dense (D w) product always                                 */
};

/* Opaque handle.  The problem description in it is immutable after gik_template_create /
 * gik_pipeline_attach; what a batch call mutates is bookkeeping only, guarded inside the library:
 *   - a ring of work-queue counters and a pool of time-slicing workspaces, handed out per call
 *     under a mutex; a slot is reused only behind the event recorded after its previous launch, so
 *     any number of calls may be in flight on any number of streams and host threads;
 *   - the workgroup-per-goal prepare kernel's scratch slab (N > 32): launches on different streams
 *     are chained by an event (they serialise; results are unaffected);
 *   - anchored templates: the event pair read by gik_anchored_last_solve_ms (diagnostic; with
 *     concurrent callers it reports whichever call recorded last).
 * Because of that bookkeeping a batch call cannot be captured into a HIP graph: on a stream that is capturing
 * (hipStreamBeginCapture) gik_solve_batch / gik_ik_batch / gik_anchored_ik_batch -- and gik_prepare_batch where it
 * uses the workgroup kernel -- return an error before touching the stream.  (A batch is one persistent launch; there
 * is no launch overhead for a graph to remove.)
 * Results never depend on that bookkeeping, on the stream, on the number of calls in flight or on how the problems of
 * a batch are scheduled.  ONE property of a call does select arithmetic, and it is the number of goals in it: planar
 * graphs of at most 16 nodes run four problems to a wavefront (rtr_quad_kernel: the k = 2 projector by substitution,
 * inner products summed per node first) in batches of at least 12 problems per CU and one problem per wavefront below
 * that, so the same goal can come back with different last bits (x agrees to ~1e-9, iteration counts on 100 %,
 * inner_total on 99.3 % of 65536 goals) depending on how many goals share the call.  debug_flags 16384 / 8192 pin the
 * four-problem / one-problem kernel at every batch size; graphik_amd.distributed.solve_batch_sharded applies the rule
 * to the GLOBAL batch and pins the result, so what it gathers does not depend on the world size.  Destroying a handle
 * while calls on it are in flight is undefined; synchronise first.                                          */
typedef struct gik_template gik_template;

/* Per-problem solver statistics (final_values of the reference's optlog + counters). */
typedef struct {
  double f;            /* final cost  f(x)                                                */
  double gradnorm;     /* final ||grad||_F                                                */
  int32_t iterations;  /* outer (trust-region) iterations                                 */
  int32_t inner_total; /* tCG iterations as the reference counts them (sum of numit+1); a
                          "model increased" exit costs one product more than it reports  */
  int32_t stop;        /* 0: gradnorm < mingradnorm, 1: maxiter, 2: NaN encountered,
                          3: step size < cg_minstepsize (ConjugateGradient only)           */
  int32_t n_accept;    /* accepted steps                                                  */
  int32_t inner_executed; /* Hessian products actually evaluated: after a rejected step the
                          reference's next tCG solve repeats the previous one up to the smaller
                          radius, and the engine resumes from a checkpoint instead (same result,
                          bit for bit); inner_total counts what the reference would have run   */
  int32_t flags;       /* bit 0: workgroup kernels, rigid clique: the clique's target distances were
                          those of a point set in R^3 and its dense D w product was replaced by
                          moments (informational; results agree to round-off either way);
                          bit 1: wavefront kernel, large batch: the problem was paused at least once
                          (round-robin time slice, or handed to a wave on an idle SIMD in the tail)
                          and resumed by another wave (same result bit for bit); bits 8..31: how
                          often.  Bit 1, bits 8.. and inner_executed depend on timing            */
  double stepsize;     /* ConjugateGradient: step of the last line search (the `stepsize` entry of
                          pymanopt's final_values); TrustRegions: trust-region radius at return  */
} gik_stats;            /* 48 bytes                                                         */

/* Optional per-outer-iteration trace (device arrays of B x cap; pass NULL to disable). */
typedef struct {
  int32_t cap;
  double *d_Delta;
  int32_t *d_numit;
  int32_t *d_stop;
  double *d_f_before;
  double *d_gradnorm_after;
  int32_t *d_accept;
} gik_trace;

const char *gik_last_error(void);
int gik_abi_version(void);
int gik_device_count(void);

/* Set `desc->` trust-region fields to the reference defaults. */
void gik_default_params(gik_template_desc *desc);
/* ... and for params["solver"] = "ConjugateGradient" (riemannian_solver.py:51-59): solver,
 * mingradnorm 1e-9, maxiter 10e4, cg_minstepsize 1e-10, cg_orth_value 10e10, cg_beta_type 3.
 * With this solver gik_stats reports: iterations = CG iterations, inner_total = inner_executed =
 * cost evaluations of the line searches, n_accept = line searches that moved, stop = 0 gradnorm,
 * 1 maxiter, 2 NaN, 3 step size below cg_minstepsize; gik_trace columns: d_f_before / d_gradnorm_after
 * = cost / |grad| before step q, d_Delta = step size, d_numit = cost evaluations of the line
 * search, d_stop = direction reset to -grad, d_accept = the line search moved.               */
void gik_default_cg_params(gik_template_desc *desc);

int gik_template_create(const gik_template_desc *desc, gik_template **out);
void gik_template_destroy(gik_template *t);

/* What gik_template_create decided (informational; e.g. bench.py's executed-flop count).        */
typedef struct {
  int32_t is_block;            /* 1: workgroup-per-problem kernels (N*k > 64 or forced)             */
  int32_t max_terms_per_node;  /* compiled slot count of the wavefront kernel variant (0 on the block path) */
  int32_t n_clique;            /* nodes of the rigid clique handled in closed form (0 = none)       */
  int32_t n_slot_terms;        /* terms left in the per-term loops (= T without a clique)           */
  int32_t slots_per_thread;    /* block path: padded per-thread slot count                          */
  int32_t waves_per_cu;        /* resident solve wavefronts (workgroups) per CU                     */
  int32_t n_cu;
  int32_t lds_bytes;           /* dynamic LDS per wavefront / workgroup of the solve kernel         */
  int32_t clique_closed_form;  /* GIK_CLIQUE_* in effect                                            */
  int32_t anchored;
  int32_t has_pipeline;
  int32_t prepare_is_block;    /* workgroup-per-goal prepare kernel                                 */
  int32_t node_per_lane;       /* != 0: an is_block graph solved by the node-per-lane kernel instead of the
                                  512-thread workgroup kernel; the value is its wavefronts per problem (2: one
                                  node per lane, 1: two nodes per lane, 4: one node per lane, graphs of 129 .. 255
                                  nodes -- the only kernel that takes them)                               */
  int32_t problems_per_wave;   /* 4: planar graph (k = 2, <= 16 nodes, <= 6 terms per node) whose trust-region
                                  solves run four problems to a wavefront (rtr_quad_kernel) in batches of at
                                  least 12 problems per CU (smaller batches: one per wavefront, see the note on
                                  the handle above); else 1 (0: block) */
  int32_t goals_per_wave;      /* prepare kernel: 4 = graph of at most 16 nodes, four goals to a wavefront
                                  (prep_quad_kernel); 1 = one (prep_wave_kernel); 0 = workgroup per goal / no pipeline */
  int32_t hessian_form;        /* what the solve kernel of this template does: GIK_HESS_PER_EDGE or GIK_HESS_COLUMN
                                  (never _AUTO; workgroup and node-per-lane kernels: always _PER_EDGE)              */
} gik_template_info;
int gik_template_get_info(const gik_template *t, gik_template_info *info);

/* costgrd twins, batched over B problems.  d_Y, d_W, d_out: [B][N*k]; d_targets: [B][T]
 * (squared goal distance for EQ terms, psi_L / psi_U for hinge terms); d_f: [B].          */
int gik_cost(const gik_template *t, const double *d_Y, const double *d_targets, int B,
             double *d_f, void *stream);                      /* lcost / jcost            */
int gik_grad(const gik_template *t, const double *d_Y, const double *d_targets, int B,
             double *d_out, void *stream);                    /* lgrad / jgrad            */
int gik_cost_and_grad(const gik_template *t, const double *d_Y, const double *d_targets, int B,
                      double *d_f, double *d_grad, void *stream); /* lcost_and_grad / jcost_and_grad:
                                                 one pass over the residual terms, same (f, G) */
int gik_hess(const gik_template *t, const double *d_Y, const double *d_W,
             const double *d_targets, int B, double *d_out, void *stream); /* lhess/jhess */
int gik_proj(const gik_template *t, const double *d_Y, const double *d_Z, int B,
             double *d_out, void *stream);                    /* PSDFixedRank.proj        */

/* TrustRegions.solve for B problems: d_Y_init -> d_Y_out ([B][N*k]), d_stats [B].          */
int gik_solve_batch(const gik_template *t, const double *d_Y_init, const double *d_targets,
                    int B, double *d_Y_out, gik_stats *d_stats, const gik_trace *trace,
                    void *stream);

/* ---- whole solve_with_riemannian pipeline on the device ------------------------------------
 * Goal-independent description of the per-goal pre/post-processing of
 * solve_with_riemannian (riemannian_solver.py:220-234): how a goal pose pins the goal nodes
 * (graph_revolute.py:243-249 / graph_planar.py:136-145), the LOWER/UPPER attributes bound
 * smoothing runs on (dgp.py:192-231), the omega pairs of linear_projection (dgp.py:174-183),
 * and the robot frames joint_variables needs (graph_revolute.py:251-318).                   */
typedef struct {
  int32_t n_joints;          /* n (1 .. 125)                                               */
  const double *T0;          /* [(n+1)][(k+1)*(k+1)] frames at zero configuration, row-major */
  const int32_t *p_index;    /* [n+1] node index of p_i                                    */
  const int32_t *q_index;    /* [n+1] node index of q_i (k=3; ignored for k=2)             */
  int32_t x_index, y_index;  /* base anchor nodes "x", "y"                                 */
  double axis_length;        /* graph.axis_length                                          */
  int32_t goal_node0;        /* node pinned at the goal position (p_n)                     */
  int32_t goal_node1;        /* q_n (k=3: p_n + axis_length*z) or p_{n-1} (k=2: p_n - len*x) */
  double goal_len;           /* axis_length (k=3) / DIST(p_{n-1}, p_n) (k=2)               */
  const double *base_lower;  /* [N*N] LOWER of the goal-independent edges, NaN = no edge   */
  const double *base_upper;  /* [N*N] UPPER                                                */
  int32_t n_anchor;          /* nodes with a known position besides the goal nodes (<= 256;
                                graphs with N > 32 or more than 32 anchors, up to N = 128, are
                                prepared by a workgroup-per-goal kernel, same arithmetic)     */
  const int32_t *anchor_index; /* [n_anchor]                                               */
  const double *anchor_pos;  /* [n_anchor][k]                                              */
  int32_t n_pairs;           /* omega pairs (i<j) of the goal graph                        */
  const int32_t *pair_i, *pair_j;
  const int32_t *term_src;   /* [T] -1: term_static[t]; else anchor_slot*2 + goal_slot     */
  const double *term_static; /* [T] psi_L / psi_U / goal-independent squared distances     */
  int32_t last_link_along_z; /* the T_final correction of graph_revolute.py:314-316 applies */
  int32_t jacobi_sweeps;     /* 0 = default (10)                                           */
  int32_t force_block_prepare; /* 1: workgroup-per-goal prepare kernel even for small graphs
                                (tests; GIK_PREP_FORCE_BLOCK is read once, at attach)         */
  /* Robots with several end effectors (k = 3 trees, robot_base.py:29-41; round 6: planar trees,
   * graph_planar.py:50-88; at most 8): goal poses are [B][n_ee][(k+1)^2] in the order of `ee_goal_nodes`;
   * n_ee <= 1 (or 0) is the chain above and the arrays below may be NULL.  last_link_along_z then
   * holds one bit per end effector.                                                             */
  int32_t n_ee;
  const int32_t *ee_goal_nodes; /* [2 * n_ee] graph nodes pinned by goal pose e: (p_e, q_e); k = 2: (the end effector,
                                   its parent -- or -1 where an earlier end effector's pose pins that parent already:
                                   graph_planar.py:136-145 keeps the first)                      */
  const int32_t *ee_path;       /* [n_ee][n + 1] joint numbers from the root to end effector e
                                   (path[0] = 0), -1 padded; a parent's number precedes its
                                   children's nowhere assumed                                   */
  int32_t n_goal_pairs;         /* goal-node pairs of DIFFERENT end effectors tied by an edge   */
  int32_t reserved1;
  const int32_t *goal_pair_a, *goal_pair_b;  /* [n_goal_pairs] slots into ee_goal_nodes; term_src
                                   of such a term = 2 n_ee n_anchor + pair; of an anchor<->goal
                                   term = anchor_slot * 2 n_ee + goal slot                     */
  const double *ee_goal_len;    /* [n_ee] k = 2 trees: length of the link parent(e) -> e (goal_len of a chain)      */
} gik_pipeline_desc;

int gik_pipeline_attach(gik_template *t, const gik_pipeline_desc *desc);

/* goal poses [B][n_ee][(k+1)^2] (row-major homogeneous matrices) -> per-term targets [B][T] and the
 * initial point [B][N*k] (from_pose + bound_smoothing + generate_initialization).
 * d_K_out (optional, [B] int32): the MDS column count chosen per goal.                     */
int gik_prepare_batch(const gik_template *t, const double *d_T_goal, int B, double *d_targets,
                      double *d_Y_init, int32_t *d_K_out, void *stream);

/* Diagnostic twin of gik_prepare_batch: also returns what the reference computes on the way, so
 * that each stage can be checked against captured reference data on its own -- bound_smoothing's
 * output (dgp.py:192-231) and the eigenvalue spectra behind generate_initialization
 * (riemannian_solver.py:67-75; dgp.py:150-183): row 0 the Gram matrix of the interpolated bounds,
 * row 1 the matrix MDS() takes its rank from (dgp.py:166-167), row 2 the scatter matrix of
 * linear_projection (zero beyond the MDS rank); all unsorted.  Any pointer may be NULL.        */
typedef struct {
  double *d_lb, *d_ub;   /* [B][N*N]  (both or neither)                                       */
  double *d_eig;         /* [B][3][N]                                                         */
} gik_prepare_diag;
int gik_prepare_batch_debug(const gik_template *t, const double *d_T_goal, int B,
                            double *d_targets, double *d_Y_init, int32_t *d_K_out,
                            const gik_prepare_diag *diag, void *stream);

/* points [B][N*k] + goal poses -> joint angles [B][n], EE position / rotation error of
 * FK(q) against the goal [B] (joint_variables + robot.pose + the examples' error metric).  */
int gik_recover_batch(const gik_template *t, const double *d_Y, const double *d_T_goal, int B,
                      double *d_q, double *d_pos_err, double *d_rot_err, void *stream);

/* prepare + solve + recover on one stream.  d_Y [B][N*k] receives the solution points and
 * doubles as the Y_init buffer; d_targets [B][T] is caller-provided scratch.               */
int gik_ik_batch(const gik_template *t, const double *d_T_goal, int B, double *d_targets,
                 double *d_Y, gik_stats *d_stats, double *d_q, double *d_pos_err,
                 double *d_rot_err, void *stream);

/* ---- fixed-anchor formulation: "intended" obstacle semantics (opt-in) -----------------------
 * graph_base.py:182-211 ties every node with a known position (base frame, goal nodes, obstacle
 * centres) to the others by equality edges and MEANS to add robot<->obstacle lower-bound hinges
 * (:205-211; the TYPE comparison at :207 never fires, so the reference creates none -- that
 * observable behaviour stays the default of this library).  Here those nodes are constants
 * instead of rows of Y, and the hinges exist: UR10 + table_environment() is a 10-node problem with
 * 100 point-to-obstacle hinges per p-node instead of N = 116 / 5612 terms.  The free nodes are the
 * template's N nodes; its terms are the free-free terms (targets are template constants:
 * `term_target`).  With an anchored template
 *   - gik_solve_batch's `d_targets` argument carries the per-problem goal anchors [B][n_goal*3],
 *     and so does the `d_targets` argument of gik_cost / gik_grad / gik_cost_and_grad / gik_hess;
 *   - gik_proj is the identity (the anchors fix the gauge; the search space is Euclidean).     */
typedef struct {
  int32_t n_anchor;              /* rows of the pinned-anchor table (<= 16): base + goal anchors */
  int32_t n_goal_anchor;         /* the LAST n_goal_anchor rows are per-problem (goal nodes)     */
  const double *anchor_pos;      /* [n_anchor][3] world positions (goal rows ignored)            */
  const double *term_target;     /* [T] squared targets of the template's (free-free) terms      */
  int32_t n_pin;                 /* point-to-anchor terms, at most 8 per free node               */
  const int32_t *pin_node;       /* [n_pin] free node                                            */
  const int32_t *pin_anchor;     /* [n_pin] anchor row                                           */
  const int32_t *pin_kind;       /* [n_pin] GIK_TERM_*                                           */
  const double *pin_target;      /* [n_pin] squared distance / psi_L / psi_U                     */
  int32_t n_obs;                 /* spherical obstacles                                          */
  int32_t reserved0;
  const double *obs;             /* [n_obs][4] centre x, y, z and SQUARED radius                 */
  const int32_t *obs_node_mask;  /* [N] 1: the free node keeps |Y_i - centre| >= radius          */
  /* mapping to the robot graph (gik_anchored_ik_batch) */
  int32_t full_N;                /* nodes of the robot graph                                     */
  int32_t reserved1;
  const int32_t *free_full_index;    /* [N]                                                      */
  const int32_t *anchor_full_index;  /* [n_anchor]; goal rows: p_n then q_n                      */
  double axis_length;
} gik_anchored_desc;

int gik_template_create_anchored(const gik_template_desc *desc, const gik_anchored_desc *adesc,
                                 gik_template **out);

/* goal poses -> joint angles through the anchored solve, one stream, no host round trip:
 * gik_prepare_batch on `base` (the robot graph WITHOUT obstacles, pipeline attached: bound
 * smoothing + MDS initial point) -> orthogonal Procrustes fit of that point's anchor rows onto the
 * world frame -> anchored trust-region solve -> full point matrix d_Y_full [B][full_N*3] ->
 * gik_recover_batch on `base`.  d_ws: scratch of gik_anchored_ws_doubles(anch, base, B) doubles. */
size_t gik_anchored_ws_doubles(const gik_template *anch, const gik_template *base, int B);
int gik_anchored_ik_batch(const gik_template *anch, const gik_template *base, const double *d_T_goal,
                          int B, double *d_ws, double *d_Y_full, gik_stats *d_stats, double *d_q,
                          double *d_pos_err, double *d_rot_err, void *stream);

/* Duration (ms, HIP events on the call's stream) of the anchored solve kernel inside the most
 * recent gik_anchored_ik_batch on this handle; waits for it.  < 0 if there was none.           */
double gik_anchored_last_solve_ms(const gik_template *anch);

#ifdef __cplusplus
}
#endif
#endif
