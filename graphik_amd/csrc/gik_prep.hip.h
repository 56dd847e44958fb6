// graphik_amd/csrc/gik_prep.hip.h -- per-goal pre- and post-processing on the device.
//
//   prep_wave_kernel : one wavefront per goal, everything in LDS:
//       goal pose -> anchor/goal distances (ProblemGraph.from_pose + graph_complete_edges,
//       graph_base.py:146-180, dgp.py:124-147) -> bound smoothing (dgp.py:192-231)
//       -> RiemannianSolver.generate_initialization (riemannian_solver.py:67-75: Gram of the
//       0.9-interpolated bounds, MDS with the reference's rank rule, linear projection)
//       -> per-term targets + Y_init for the solve kernel.
//     The three symmetric eigendecompositions (N x N Gram, the N x N "rank" matrix, K x K
//     scatter) are cyclic Jacobi sweeps in round-robin order: the N/2 rotations of a round act on
//     disjoint index pairs, so a round is three conflict-free passes over LDS rows/columns.
//   recover_kernel   : one thread per goal: joint_variables (graph_revolute.py:251-318 /
//       graph_planar.py:147-176) + forward kinematics of the recovered angles -> EE pose error.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gik_wave.hip.h"

namespace gik {

struct PipeConst {
  // graph template (device pointers)
  const double *base_lower;   // [N*N]  NaN = no edge; goal edges are added per problem
  const double *base_upper;   // [N*N]
  const int *anchor_idx;      // [A]   nodes with a fixed position (base frame, obstacles)
  const double *anchor_pos;   // [A][K]
  const int *pair_i, *pair_j; // [P]   omega pairs (i<j), goal edges included
  const int *term_src;        // [T]   -1: static target, else anchor_slot*2 + goal_slot
  const double *term_static;  // [T]
  // robot
  const double *T0;           // [n+1][(K+1)^2] frames at zero configuration (row-major)
  const int *p_idx, *q_idx;   // [n+1] node index of p_i / q_i
  int N, K, T, n_anchor, n_pairs, n_joints;
  int goal0, goal1;           // ee node, and q_n (k=3) or p_{n-1} (k=2)   (first end effector)
  int x_idx, y_idx;
  double goal_len;            // axis_length (k=3) / |p_{n-1} p_n| (k=2)
  double axis_length;
  int last_along_z;           // joint_variables: last link offset parallel to z (:314), bit e = end effector e
  // several end effectors (k = 3 trees; round 6: planar trees): goal poses are [B][n_ee][(K+1)^2]; goal node 2e / 2e+1 is
  // (p, q) of end effector e (k = 2: the end effector and its parent).  Distances: gd[ai * 2 n_ee + g] anchor ai <-> goal node g, then the
  // n_gg goal-node pairs of DIFFERENT end effectors.  A chain has n_ee = 1, n_gg = 0 (the layout
  // and arithmetic of the single-end-effector kernels, bit for bit).
  int n_ee, n_gg;
  int goal_node[2 * 8];       // graph node of goal node g; -1: inert slot (k = 2 trees: a parent that an earlier
                              // end effector's pose already pins -- graph_planar.py:136-145 through BatchProblem)
  double ee_len[8];           // goal_len per end effector: axis_length (k = 3), |parent(e) e| (k = 2)
  const int *gg_a, *gg_b;     // [n_gg] goal-node slots of each pair
  const int *ee_path;         // [n_ee][n+1] joints from the root to end effector e, -1 padded
};
constexpr int PREP_MAX_EE = 8;      // (4 until round 6)

// position of goal node g = 2 e + s of problem `Tg` ([n_ee][(K+1)^2]): p_e, or q_e = p_e + len z_e
// (graph_revolute.py:243-249); k = 2: p_n, p_{n-1} = p_n - len x_n (graph_planar.py:136-145)
__device__ inline void goal_node_pos(const PipeConst &pc, const double *Tg, int g, double (&w)[3]) {
  const int K = pc.K, D = K + 1;
  const double *T = Tg + (size_t)(g >> 1) * D * D;
  for (int c = 0; c < K; ++c) {
    const double p = T[c * D + K];
    const double len = pc.ee_len[g >> 1];
    w[c] = !(g & 1) ? p : ((K == 3) ? p + T[c * D + 2] * len : p - T[c * D + 0] * len);
  }
}

// anchor <-> goal and goal <-> goal distances of one problem: entry idx of gd (see PipeConst)
__device__ inline double goal_distance(const PipeConst &pc, const double *Tg, int idx, int &na, int &nb) {
  const int K = pc.K, G = 2 * pc.n_ee;
  double a[3], b[3];
  if (idx < G * pc.n_anchor) {
    const int ai = idx / G, g = idx - ai * G;
    for (int c = 0; c < K; ++c) a[c] = pc.anchor_pos[ai * K + c];
    goal_node_pos(pc, Tg, g, b);
    na = pc.anchor_idx[ai];
    nb = pc.goal_node[g];
  } else {
    const int q = idx - G * pc.n_anchor;
    goal_node_pos(pc, Tg, pc.gg_a[q], a);
    goal_node_pos(pc, Tg, pc.gg_b[q], b);
    na = pc.goal_node[pc.gg_a[q]];
    nb = pc.goal_node[pc.gg_b[q]];
  }
  double d2 = 0.0;
  for (int c = 0; c < K; ++c) {
    const double df = a[c] - b[c];
    d2 += df * df;
  }
  return sqrt(d2);   // np.linalg.norm
}

// A rotation is skipped when |a_pq| <= 1e-16 |A|_F, and the sweeps stop after the first one without a
// rotation.  (Relaxing the threshold buys nothing -- measured and emulated: the Gram matrices here
// are rank deficient and cyclic Jacobi spends ~6 sweeps in its linear phase whatever the threshold
// between 1e-16 and 1e-12, then collapses within one sweep.)
// round-robin (chess tournament) pairing: n_even players, round r, table m
__device__ inline void rr_pair(int n_even, int r, int m, int &p, int &q) {
  if (m == 0) {
    p = n_even - 1;
    q = r;
  } else {
    p = (r + m) % (n_even - 1);
    q = (r - m + n_even - 1) % (n_even - 1);
  }
  if (p > q) {
    const int t = p;
    p = q;
    q = t;
  }
}

// Cyclic Jacobi on the symmetric N x N matrix A (LDS, row stride N); V (optional) accumulates
// the eigenvectors as columns.  cs: 2*16 doubles, pq: 16 ints of LDS scratch.  At most `sweeps`
// sweeps; stops after the first sweep in which no off-diagonal entry exceeded 1e-16 |A|_F
// (rotations below that are skipped, so the confirming sweep is cheap; typically 6-7 sweeps).
// Only the leading n x n block is decomposed (row stride stays N): the scatter matrix of
// linear_projection is non-zero in its leading K x K block only (K ~ 6 of N = 18).
// tab: (ne - 1) * np ints of LDS scratch (ne = n rounded up to even, np = ne / 2): the round-robin
// pairs of every round, worked out once per call instead of two integer modulos per lane and round.
__device__ inline void jacobi_lds(double *A, double *V, int N, int sweeps, double *cs, int *pq, int *tab,
                                  int lane, int n = -1) {
  if (n < 0) n = N;
  const int stride = N;
  N = n;
  const int ne = N + (N & 1), np = ne / 2;
  for (int e = lane; e < (ne - 1) * np; e += WAVE) {
    const int r = e / np;
    int p, q;
    rr_pair(ne, r, e - r * np, p, q);
    tab[e] = p | (q << 8);
  }
  __builtin_amdgcn_wave_barrier();
  double fro = 0.0;
  if (lane < N)
    for (int j = 0; j < N; ++j) fro = fma(A[lane * stride + j], A[lane * stride + j], fro);
  const double thr = 1e-16 * sqrt(wave_sum(fro));
  for (int sw = 0; sw < sweeps; ++sw) {
    bool rotated = false;
    for (int r = 0; r < ne - 1; ++r) {
      bool sig = false;
      if (lane < np) {
        const int pair = tab[r * np + lane], p = pair & 0xff, q = pair >> 8;
        double c = 1.0, s = 0.0;
        int code = -1;
        if (q < N) {
          const double apq = A[p * stride + q];
          if (fabs(apq) > thr) {
            sig = true;
            // t = sgn(th) / (|th| + sqrt(th^2 + 1)), c = 1 / sqrt(t^2 + 1), s = t c  with
            // reciprocal / reciprocal-square-root Newton steps (~1 ulp; Jacobi is self-correcting)
            // instead of two IEEE square roots and two divisions per rotation
            const double th = (A[q * stride + q] - A[p * stride + p]) * (0.5 * frcp(apq));
            const double h2 = fma(th, th, 1.0);
            const double t = (th >= 0.0 ? 1.0 : -1.0) * frcp(fabs(th) + h2 * frsqrt(h2));
            c = frsqrt(fma(t, t, 1.0));
            s = t * c;
            code = pair;
          }
        }
        cs[2 * lane] = c;
        cs[2 * lane + 1] = s;
        pq[lane] = code;
      }
      if (__builtin_amdgcn_ballot_w64(sig) == 0ull) continue;  // nothing to rotate in this round
      rotated = true;
      __builtin_amdgcn_wave_barrier();
      // The rotations of a round touch disjoint index pairs, so several are applied at once:
      // the wavefront is cut into groups of GW lanes (GW = 16 if the matrix has at most 16 rows,
      // else 32), each group takes one rotation per step and its lanes take the rows / columns.
      const int gsh = (N <= 16) ? 4 : 5, GW = 1 << gsh, idx = lane & (GW - 1);
      // column phase: the lower half of the wave rotates A's columns, the upper half V's
      {
        const int half_groups = 32 >> gsh;                 // groups per matrix: 2 or 1
        const int grp = (lane & 31) >> gsh;
        double *M = (lane < 32) ? A : V;
        for (int m0 = 0; m0 < np; m0 += half_groups) {
          const int m = m0 + grp;
          const int code = (m < np) ? pq[m] : -1;
          if (code >= 0 && idx < N && M != nullptr) {
            const int p = code & 0xff, q = code >> 8;
            const double c = cs[2 * m], s = cs[2 * m + 1];
            const double ap = M[idx * stride + p], aq = M[idx * stride + q];
            M[idx * stride + p] = c * ap - s * aq;
            M[idx * stride + q] = s * ap + c * aq;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      // row phase on A: all 64 lanes, 64 / GW rotations per step, lanes take the columns
      {
        const int groups = 64 >> gsh, grp = lane >> gsh;   // 4 or 2
        for (int m0 = 0; m0 < np; m0 += groups) {
          const int m = m0 + grp;
          const int code = (m < np) ? pq[m] : -1;
          if (code >= 0 && idx < N) {
            const int p = code & 0xff, q = code >> 8;
            const double c = cs[2 * m], s = cs[2 * m + 1];
            const double ap = A[p * stride + idx], aq = A[q * stride + idx];
            A[p * stride + idx] = c * ap - s * aq;
            A[q * stride + idx] = s * ap + c * aq;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (!rotated) break;
  }
}

// Number of eigenvalues > tau of the symmetric N x N matrix A (LDS, row stride N, destroyed):
// Householder reduction to tridiagonal form + a Sturm count.  MDS() only takes its column count
// from this spectrum (dgp.py:166-167: `len(evals[evals > eps])`), so the eigenvalues themselves
// are not needed: ~1.6 k instructions at N = 13 against 8.3 k for a cyclic-Jacobi decomposition.
// The count differs from eigh's only if an eigenvalue lies within ~N eps |A| of tau.
// v, w: 32 doubles of LDS scratch each.
__device__ inline int count_eigs_above_lds(double *A, int N, double tau, double *v, double *w, int lane) {
  for (int k = 0; k + 2 < N; ++k) {
    const int m = N - k - 1;                       // trailing block (k+1 .. N-1)
    const bool mine = lane < m;
    const double x = mine ? A[(k + 1 + lane) * N + k] : 0.0;
    const double x0 = A[(k + 1) * N + k];
    double sg[2] = {x * x, (mine && lane > 0) ? x * x : 0.0};
    wave_sum_n<2>(sg);
    if (sg[1] == 0.0) continue;                    // column already tridiagonal (uniform)
    const double alpha = x0 > 0.0 ? -sqrt(sg[0]) : sqrt(sg[0]);
    const double vj = mine ? (lane == 0 ? x - alpha : x) : 0.0;
    const double beta = 1.0 / (sg[0] - alpha * x0);           // 2 / v'v
    if (lane < 32) v[lane] = vj;
    __builtin_amdgcn_wave_barrier();
    double pj = 0.0;
    if (mine)
      for (int i = 0; i < m; ++i) pj = fma(A[(k + 1 + lane) * N + (k + 1 + i)], v[i], pj);
    pj *= beta;
    const double Kc = 0.5 * beta * wave_sum(vj * pj);
    const double wj = pj - Kc * vj;
    if (lane < 32) w[lane] = mine ? wj : 0.0;
    __builtin_amdgcn_wave_barrier();
    if (mine) {
      for (int i = 0; i < m; ++i) {
        const int e = (k + 1 + lane) * N + (k + 1 + i);
        A[e] = A[e] - vj * w[i] - wj * v[i];
      }
      if (lane == 0) A[(k + 1) * N + k] = alpha;   // sub-diagonal entry of the tridiagonal form
    }
    __builtin_amdgcn_wave_barrier();
  }
  // Sturm count at tau: q_i = a_i - tau - b_{i-1}^2 / q_{i-1}; #(q_i < 0) = #eigenvalues < tau
  int below = 0;
  double q = A[0] - tau;
  below += q < 0.0;
  for (int i = 1; i < N; ++i) {
    const double bb = A[i * N + i - 1];
    if (q == 0.0) q = 1e-300;
    q = A[i * N + i] - tau - bb * bb / q;
    below += q < 0.0;
  }
  return N - below;
}

// rank of eigenvalue c in DESCENDING order (ties broken by index), for lanes c < N
__device__ inline int desc_rank(const double *ev, int N, int c) {
  int rk = 0;
  const double v = ev[c];
  for (int j = 0; j < N; ++j) rk += (ev[j] > v) || (ev[j] == v && j < c);
  return rk;
}

struct PrepArgs {
  PipeConst pc;
  const double *T_goal;  // [B][(K+1)^2]
  double *targets;       // [B][T]
  double *Y_init;        // [B][N*K]
  int *K_out;            // [B] MDS column count (diagnostic), may be null
  // diagnostics (gik_prepare_batch_debug), each may be null: bound_smoothing's output and the three
  // eigenvalue spectra of generate_initialization (Gram matrix, MDS's rank matrix, scatter matrix)
  double *dbg_lb, *dbg_ub;  // [B][N*N]
  double *dbg_eig;          // [B][3][N]
  int B, sweeps;
  int stop_phase;        // developer build only (GIK_PREP_STOP): leave a goal after phase p (timing)
  int no_compress;       // workgroup kernel: 1 = full N x N Jacobi even for rank-deficient Gram matrices (tests, A/B)
};

__global__ void __launch_bounds__(WAVE) prep_wave_kernel(PrepArgs a)
#ifndef GIK_DEFINE_PLAIN_KERNELS
    ;      // (defined in gik_k_prep.hip)
#else
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const PipeConst &pc = a.pc;
  const int N = pc.N, K = pc.K, NN = N * N, lane = threadIdx.x;
  double *U = smem;            // upper bounds -> ub
  double *L = U + NN;          // lower bounds
  double *A = L + NN;          // work matrix
  double *V = A + NN;          // eigenvectors / temp
  double *X = V + NN;          // MDS factor
  double *gd = X + NN;         // [2 n_ee n_anchor + n_gg] anchor<->goal and goal<->goal distances
  double *cs = gd + 2 * pc.n_ee * pc.n_anchor + pc.n_gg;
  double *ev = cs + 32;        // [32] eigenvalues / row means
  double *sg = ev + 32;        // [32] signs*scale
  int *pq = reinterpret_cast<int *>(sg + 32);  // [16]
  int *rk = pq + 16;                           // [32]
  int *jtab = rk + 32;                         // [(ne - 1) ne / 2], ne = N rounded up to even
  const int D = K + 1;

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const double *Tg = a.T_goal + (size_t)b * D * D * pc.n_ee;
    const int n_gd = 2 * pc.n_ee * pc.n_anchor + pc.n_gg;
    for (int e = lane; e < NN; e += WAVE) {
      const double lo = pc.base_lower[e], up = pc.base_upper[e];
      const bool diag = (e / N) == (e % N);
      U[e] = diag ? 0.0 : (up == up ? up : INFINITY);
      L[e] = diag ? 0.0 : (lo == lo ? lo : -INFINITY);
    }
    __builtin_amdgcn_wave_barrier();
    // goal nodes (_pose_goal) and the distances graph_complete_edges gives them (dgp.py:124-147)
    for (int idx = lane; idx < n_gd; idx += WAVE) {
      int an, gn;
      const double d = goal_distance(pc, Tg, idx, an, gn);
      gd[idx] = d;
      if (an >= 0 && gn >= 0) {      // (an inert goal slot pins nothing)
        U[an * N + gn] = U[gn * N + an] = d;
        L[an * N + gn] = L[gn * N + an] = d;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // per-term targets: squared goal distances for the goal edges, template constants otherwise
    for (int t = lane; t < pc.T; t += WAVE) {
      const int src = pc.term_src[t];
      const double g = src >= 0 ? gd[src] : 0.0;
      a.targets[(size_t)b * pc.T + t] = src >= 0 ? g * g : pc.term_static[t];
    }
#ifdef GIK_DEV
    if (a.stop_phase == 1) continue;
#endif
    // ---- bound smoothing: ub = APSP(UPPER) (Floyd-Warshall), then
    //      lb[u][v] = max(0, max_{a,b} LOWER[a][b] - ub[u][a] - ub[b][v])   (see dgp.py)
    for (int m = 0; m < N; ++m) {
      for (int e = lane; e < NN; e += WAVE) {
        const int i = e / N, j = e - i * N;
        const double cand = U[i * N + m] + U[m * N + j];
        if (cand < U[e]) U[e] = cand;
      }
      __builtin_amdgcn_wave_barrier();
    }
#ifdef GIK_DEV
    if (a.stop_phase == 2) continue;
#endif
    for (int e = lane; e < NN; e += WAVE) {  // A[u][b] = max_a (L[a][b] - U[u][a])
      const int u = e / N, bb = e - u * N;
      double best = -INFINITY;
      for (int q = 0; q < N; ++q) best = fmax(best, L[q * N + bb] - U[u * N + q]);
      A[e] = best;
    }
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < NN; e += WAVE) {  // V[u][v] = lb
      const int u = e / N, v = e - u * N;
      double best = 0.0;
      for (int q = 0; q < N; ++q) best = fmax(best, A[u * N + q] - U[q * N + v]);
      V[e] = best;
    }
    __builtin_amdgcn_wave_barrier();
#ifdef GIK_DEV
    if (a.stop_phase == 3) continue;
#endif
    if (a.dbg_lb)
      for (int e = lane; e < NN; e += WAVE) {
        a.dbg_lb[(size_t)b * NN + e] = V[e];
        a.dbg_ub[(size_t)b * NN + e] = U[e];
      }
    // ---- generate_initialization: D_rand = (lb + 0.9 (ub - lb))^2, Gram = -1/2 J D J
    for (int e = lane; e < NN; e += WAVE) {
      const double lbv = V[e], d = lbv + 0.9 * (U[e] - lbv);
      X[e] = d * d;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < N) {
      double s = 0.0;
      for (int j = 0; j < N; ++j) s += X[lane * N + j];
      ev[lane] = s / N;
    }
    __builtin_amdgcn_wave_barrier();
    double mean = 0.0;
    for (int j = 0; j < N; ++j) mean += ev[j];
    mean /= N;
    for (int e = lane; e < NN; e += WAVE) {
      const int i = e / N, j = e - i * N;
      A[e] = -0.5 * (X[e] - ev[i] - ev[j] + mean);
      V[e] = (i == j) ? 1.0 : 0.0;
    }
    __builtin_amdgcn_wave_barrier();
#ifdef GIK_DEV
    if (a.stop_phase == 4) continue;
#endif
    jacobi_lds(A, V, N, a.sweeps, cs, pq, jtab, lane);
#ifdef GIK_DEV
    if (a.stop_phase == 5) continue;
#endif
    // ---- factor(): clip, scale by sqrt(lambda), order descending (fliplr of ascending)
    if (lane < N) ev[lane] = A[lane * N + lane];
    if (a.dbg_eig && lane < N) a.dbg_eig[((size_t)b * 3 + 0) * N + lane] = A[lane * N + lane];
    __builtin_amdgcn_wave_barrier();
    if (lane < N) {
      rk[lane] = desc_rank(ev, N, lane);
      // canonical sign: the entry of largest magnitude (first on ties) is positive.  LAPACK's
      // sign is an implementation accident; the reference's rank rule below depends on it.
      double big = 0.0, sgn = 1.0;
      for (int r = 0; r < N; ++r) {
        const double v = V[r * N + lane];
        if (fabs(v) > big) {
          big = fabs(v);
          sgn = v < 0.0 ? -1.0 : 1.0;
        }
      }
      sg[lane] = sgn * sqrt(fmax(ev[lane], 0.0));
    }
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < NN; e += WAVE) {
      const int r = e / N, c = e - r * N;
      X[r * N + rk[c]] = V[e] * sg[c];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- MDS(): K = #eigenvalues > eps of eigh(x), i.e. of the symmetric matrix read from the
    //      LOWER triangle of the non-symmetric factor x (dgp.py:166-167, numpy UPLO='L')
    for (int e = lane; e < NN; e += WAVE) {
      const int i = e / N, j = e - i * N;
      A[e] = (i >= j) ? X[i * N + j] : X[j * N + i];
    }
    __builtin_amdgcn_wave_barrier();
    if (a.dbg_eig) {   // diagnostics only: the spectrum itself, then A is rebuilt for the count
      jacobi_lds(A, nullptr, N, a.sweeps, cs, pq, jtab, lane);
      if (lane < N) a.dbg_eig[((size_t)b * 3 + 1) * N + lane] = A[lane * N + lane];
      __builtin_amdgcn_wave_barrier();
      for (int e = lane; e < NN; e += WAVE) {
        const int i = e / N, j = e - i * N;
        A[e] = (i >= j) ? X[i * N + j] : X[j * N + i];
      }
      __builtin_amdgcn_wave_barrier();
    }
#ifdef GIK_DEV
    if (a.stop_phase == 6) continue;
#endif
    const int Kc = count_eigs_above_lds(A, N, 1e-8, cs, ev, lane);
    if (a.K_out && lane == 0) a.K_out[b] = Kc;
    __builtin_amdgcn_wave_barrier();
#ifdef GIK_DEV
    if (a.stop_phase == 7) continue;
#endif
    // ---- linear_projection (dgp.py:174-183): scatter of the edge differences of the first Kc
    //      columns, its top-`dim` eigenvectors
    for (int e = lane; e < NN; e += WAVE) {
      const int r = e / N, c = e - r * N;
      if (c >= Kc) X[e] = 0.0;
    }
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < NN; e += WAVE) {
      const int r = e / N, c = e - r * N;
      double s = 0.0;
      if (r < Kc && c < Kc) {
        for (int p = 0; p < pc.n_pairs; ++p) {
          const int i = pc.pair_i[p], j = pc.pair_j[p];
          s = fma(X[i * N + r] - X[j * N + r], X[i * N + c] - X[j * N + c], s);
        }
      }
      A[e] = 2.0 * s;  // the reference sums both (i,j) and (j,i)
      V[e] = (r == c) ? 1.0 : 0.0;
    }
    __builtin_amdgcn_wave_barrier();
#ifdef GIK_DEV
    if (a.stop_phase == 8) continue;
#endif
    jacobi_lds(A, V, N, a.sweeps, cs, pq, jtab, lane, Kc > 1 ? Kc : 2);
#ifdef GIK_DEV
    if (a.stop_phase == 9) continue;
#endif
    if (lane < N) ev[lane] = (lane < Kc) ? A[lane * N + lane] : -INFINITY;
    if (a.dbg_eig && lane < N) a.dbg_eig[((size_t)b * 3 + 2) * N + lane] = (lane < Kc) ? A[lane * N + lane] : 0.0;
    __builtin_amdgcn_wave_barrier();
    if (lane < N) {
      rk[lane] = desc_rank(ev, N, lane);
      double big = 0.0, sgn = 1.0;
      for (int r = 0; r < N; ++r) {
        const double v = V[r * N + lane];
        if (fabs(v) > big) {
          big = fabs(v);
          sgn = v < 0.0 ? -1.0 : 1.0;
        }
      }
      sg[lane] = sgn;
    }
    __builtin_amdgcn_wave_barrier();
    // Y = X * W, W = the K eigenvectors of largest eigenvalue
    if (lane < N * K) {
      const int r = lane / K, dcol = lane - r * K;
      int col = 0;
      for (int c = 0; c < N; ++c) col = (rk[c] == dcol) ? c : col;
      double s = 0.0;
      for (int c = 0; c < N; ++c) s += X[r * N + c] * V[c * N + col];
      a.Y_init[(size_t)b * N * K + lane] = s * sg[col];
    }
    __builtin_amdgcn_wave_barrier();
  }
}
#endif

// ---------------------------------------------------------------------------------------------
// prep_block_kernel: the same pre-processing for graphs beyond one wavefront's LDS (N <= 128, up to
// 256 anchors: UR10 + table_environment(), N = 116): one 512-thread workgroup per goal, the
// N x N matrices in a per-workgroup slab of global memory (L2 / Infinity-Cache resident) except the
// one being worked on, which sits in LDS when it fits (prep_block_kernel<true>), the small vectors
// in LDS.  Same operations in the same order per matrix element as prep_wave_kernel -- the
// rotations of a Jacobi round act on disjoint index pairs, every other phase is element-wise -- so
// on a graph both kernels can take the results agree bit for bit (tests).
constexpr int PREP_NT = 512;
constexpr int PREP_MAXN = 128;
constexpr int PREP_MAXA = 256;
constexpr int PREP_PC = 32;      // pairs per LDS tile of the scatter matrix

__device__ inline double prep_block_sum(double x, double *red, int tid) {
  x = wave_sum(x);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = x;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < PREP_NT / 64; ++w) s += red[w];
  return s;
}

// cyclic Jacobi, round-robin order, block-wide (see jacobi_lds for the arithmetic).
// Eigenvectors: V lives in the global slab, and rotating its columns round by round reads and
// writes the whole matrix 115 times per sweep (0.8 TB through the L2 per 4096 table-scene goals,
// the bulk of the kernel).  With a log buffer (`vlog`, 28 KB of LDS) the rounds only rotate A and
// record their (c, s, p, q); every JLOG_ROUNDS rotated rounds the log is applied to V row by row
// -- a wavefront keeps two rows in LDS, runs the logged rounds over it (the pairs of a round are
// disjoint: 58 lanes in parallel, rounds in order) and writes it back -- so V is read and written
// once per 12 rounds.  Every element still goes through the same rotations in the same order:
// bit-identical to the round-by-round version (and to the wavefront kernel).
constexpr int JLOG_ROUNDS = 12;
constexpr int JLOG_DOUBLES = JLOG_ROUNDS * PREP_MAXN + JLOG_ROUNDS * (PREP_MAXN / 2) / 2 + 2 * (PREP_NT / 64) * PREP_MAXN;

__device__ inline void jacobi_apply_log(double *V, int stride, int n, int np, int nlog, const double *cs_log,
                                        const int *pq_log, double *rows, int tid) {
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  // two rows per wavefront at a time (the rounds of a row are a chain of dependent LDS round
  // trips, two independent chains overlap), and the next two rows are requested from the slab
  // into registers before the rounds of the current ones start: a row costs two trips through
  // the L2 / Infinity Cache, which would otherwise sit between any two row pairs of a wavefront
  double *rb0 = rows + (2 * wave) * PREP_MAXN, *rb1 = rb0 + PREP_MAXN;
  const int c0 = lane, c1 = lane + 64;
  const bool h0 = c0 < n, h1 = c1 < n;
  constexpr int STEP = 2 * (PREP_NT / 64);
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  auto fetch = [&](int row) {
    if (row < n) {
      if (h0) a0 = V[row * stride + c0];
      if (h1) a1 = V[row * stride + c1];
      if (row + 1 < n) {
        if (h0) b0 = V[(row + 1) * stride + c0];
        if (h1) b1 = V[(row + 1) * stride + c1];
      }
    }
  };
  fetch(2 * wave);
  for (int row = 2 * wave; row < n; row += STEP) {
    const bool two = row + 1 < n;
    if (h0) { rb0[c0] = a0; rb1[c0] = b0; }
    if (h1) { rb0[c1] = a1; rb1[c1] = b1; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    fetch(row + STEP);
    for (int k = 0; k < nlog; ++k) {
      if (lane < np) {
        const int code = pq_log[k * (PREP_MAXN / 2) + lane];
        if (code >= 0) {
          const int p = code & 0xff, q = code >> 8;
          const double c = cs_log[k * PREP_MAXN + 2 * lane], s = cs_log[k * PREP_MAXN + 2 * lane + 1];
          const double ap = rb0[p], aq = rb0[q], bp = rb1[p], bq = rb1[q];
          rb0[p] = c * ap - s * aq;
          rb0[q] = s * ap + c * aq;
          rb1[p] = c * bp - s * bq;
          rb1[q] = s * bp + c * bq;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (h0) V[row * stride + c0] = rb0[c0];
    if (h1) V[row * stride + c1] = rb0[c1];
    if (two) {
      if (h0) V[(row + 1) * stride + c0] = rb1[c0];
      if (h1) V[(row + 1) * stride + c1] = rb1[c1];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // rb is rewritten at the top of the loop
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
}

__device__ inline void jacobi_blk(double *A, double *V, int N, int sweeps, double *cs, int *pq,
                                  double *red, int tid, int n = -1, double *vlog = nullptr) {
  if (n < 0) n = N;
  const int stride = N;
  N = n;
  const int ne = N + (N & 1), np = ne / 2;
  double *cs_log = vlog;
  int *pq_log = reinterpret_cast<int *>(vlog + JLOG_ROUNDS * PREP_MAXN);
  double *rows = vlog + JLOG_ROUNDS * PREP_MAXN + JLOG_ROUNDS * (PREP_MAXN / 2) / 2;   // two rows per wavefront
  const bool logged = V != nullptr && vlog != nullptr && N <= PREP_MAXN;   // (the log's rows hold 2 x 64 columns)
  int nlog = 0;
  // quotient / remainder of the thread's first item and of the stride, for both item orders
  const int col_i0 = tid / np, col_m0 = tid - col_i0 * np, col_dq = PREP_NT / np, col_dr = PREP_NT - col_dq * np;
  const int row_m0 = tid / N, row_i0 = tid - row_m0 * N, row_dq = PREP_NT / N, row_dr = PREP_NT - row_dq * N;
  double fro = 0.0;
  for (int e = tid; e < N * N; e += PREP_NT) {
    const int i = e / N, j = e - i * N;
    const double v = A[i * stride + j];
    fro = fma(v, v, fro);
  }
  const double thr = 1e-16 * sqrt(prep_block_sum(fro, red, tid));
  for (int sw = 0; sw < sweeps; ++sw) {
    bool rotated = false;
    for (int r = 0; r < ne - 1; ++r) {
      int sig = 0;
      if (tid < np) {
        int p, q;
        rr_pair(ne, r, tid, p, q);
        double c = 1.0, s = 0.0;
        int code = -1;
        if (q < N) {
          const double apq = A[p * stride + q];
          if (fabs(apq) > thr) {
            sig = 1;
            const double th = (A[q * stride + q] - A[p * stride + p]) * (0.5 * frcp(apq));
            const double h2 = fma(th, th, 1.0);
            const double t = (th >= 0.0 ? 1.0 : -1.0) * frcp(fabs(th) + h2 * frsqrt(h2));
            c = frsqrt(fma(t, t, 1.0));
            s = t * c;
            code = p | (q << 8);
          }
        }
        cs[2 * tid] = c;
        cs[2 * tid + 1] = s;
        pq[tid] = code;
        if (logged) {   // slot nlog is overwritten by the next round if this one rotates nothing
          cs_log[nlog * PREP_MAXN + 2 * tid] = c;
          cs_log[nlog * PREP_MAXN + 2 * tid + 1] = s;
          pq_log[nlog * (PREP_MAXN / 2) + tid] = code;
        }
      }
      if (!__syncthreads_or(sig)) continue;   // nothing to rotate in this round
      rotated = true;
      // column phase: A's and V's columns p, q of every row
      // (consecutive threads take the rotations of ONE row: its 2 np entries lie in a few cache
      // lines, whereas consecutive rows of one column are N doubles apart -- 64 lines per wave)
      // (A and V in loops of their own: A may live in LDS, and each loop then has one address space)
      // (item `it` = (row idx, pair m) = (it / np, it % np): the quotient and remainder are
      // stepped, not divided -- an integer division by a run-time divisor is ~40 instructions, more
      // than the rotation itself, and there were 26 of them per thread and round)
      const int per = np * N;
      {
        int idx = col_i0, m = col_m0;
        for (int it = tid; it < per; it += PREP_NT) {
          const int code = pq[m];
          if (code >= 0) {
            const int p = code & 0xff, q = code >> 8;
            const double c = cs[2 * m], s = cs[2 * m + 1];
            const double ap = A[idx * stride + p], aq = A[idx * stride + q];
            A[idx * stride + p] = c * ap - s * aq;
            A[idx * stride + q] = s * ap + c * aq;
          }
          idx += col_dq;
          m += col_dr;
          if (m >= np) {
            m -= np;
            ++idx;
          }
        }
      }
      if (V != nullptr && !logged) {
        int idx = col_i0, m = col_m0;
#pragma unroll 2
        for (int it = tid; it < per; it += PREP_NT) {
          const int code = pq[m];
          if (code >= 0) {
            const int p = code & 0xff, q = code >> 8;
            const double c = cs[2 * m], s = cs[2 * m + 1];
            const double ap = V[idx * stride + p], aq = V[idx * stride + q];
            V[idx * stride + p] = c * ap - s * aq;
            V[idx * stride + q] = s * ap + c * aq;
          }
          idx += col_dq;
          m += col_dr;
          if (m >= np) {
            m -= np;
            ++idx;
          }
        }
      }
      __syncthreads();
      // row phase on A: rows p, q of every column
      {
        int m = row_m0, idx = row_i0;
        for (int it = tid; it < per; it += PREP_NT) {
          const int code = pq[m];
          if (code >= 0) {
            const int p = code & 0xff, q = code >> 8;
            const double c = cs[2 * m], s = cs[2 * m + 1];
            const double ap = A[p * stride + idx], aq = A[q * stride + idx];
            A[p * stride + idx] = c * ap - s * aq;
            A[q * stride + idx] = s * ap + c * aq;
          }
          m += row_dq;
          idx += row_dr;
          if (idx >= N) {
            idx -= N;
            ++m;
          }
        }
      }
      __syncthreads();
      if (logged && ++nlog == JLOG_ROUNDS) {
        jacobi_apply_log(V, stride, N, np, nlog, cs_log, pq_log, rows, tid);
        nlog = 0;
      }
    }
    if (!rotated) break;
  }
  if (logged && nlog) jacobi_apply_log(V, stride, N, np, nlog, cs_log, pq_log, rows, tid);
}

// count_eigs_above_lds for the workgroup-per-goal kernel: A in global memory (row stride N), v / w in
// LDS ([PREP_MAXN] each), thread j takes row j of the trailing block.  Same operations per element in
// the same order as the wavefront version, so both give the same count on a graph both can take.
__device__ inline int count_eigs_above_blk(double *A, int N, double tau, double *v, double *w, double *red,
                                           int tid) {
  for (int k = 0; k + 2 < N; ++k) {
    const int m = N - k - 1;
    const bool mine = tid < m;
    const double x = mine ? A[(k + 1 + tid) * N + k] : 0.0;
    const double x0 = A[(k + 1) * N + k];
    const double s_all = prep_block_sum(x * x, red, tid);
    const double s_tail = prep_block_sum((mine && tid > 0) ? x * x : 0.0, red, tid);
    if (s_tail == 0.0) continue;
    const double alpha = x0 > 0.0 ? -sqrt(s_all) : sqrt(s_all);
    const double vj = mine ? (tid == 0 ? x - alpha : x) : 0.0;
    const double beta = 1.0 / (s_all - alpha * x0);
    if (tid < N) v[tid] = vj;
    __syncthreads();
    double pj = 0.0;
    if (mine)
      for (int i = 0; i < m; ++i) pj = fma(A[(k + 1 + tid) * N + (k + 1 + i)], v[i], pj);
    pj *= beta;
    const double Kc = 0.5 * beta * prep_block_sum(vj * pj, red, tid);
    const double wj = pj - Kc * vj;
    if (tid < N) w[tid] = mine ? wj : 0.0;
    __syncthreads();
    if (mine) {
      for (int i = 0; i < m; ++i) {
        const int e = (k + 1 + tid) * N + (k + 1 + i);
        A[e] = A[e] - vj * w[i] - wj * v[i];
      }
      if (tid == 0) A[(k + 1) * N + k] = alpha;
    }
    __syncthreads();
  }
  int below = 0;
  double q = A[0] - tau;
  below += q < 0.0;
  for (int i = 1; i < N; ++i) {
    const double bb = A[i * N + i - 1];
    if (q == 0.0) q = 1e-300;
    q = A[i * N + i] - tau - bb * bb / q;
    below += q < 0.0;
  }
  return N - below;
}

// ---- range compression of a rank-deficient symmetric matrix (workgroup-per-goal kernel, N >= 48) ----
// The Gram matrix of a scene with many anchors has a numerical rank far below N: the anchors' mutual
// distances are exact, so their block of the distance matrix is a true EDM (rank 5) and the rest is
// the 2 x (free nodes) rows and columns -- rank ~ 30 at N = 116 (UR10 + table_environment()).  Cyclic
// Jacobi on the full matrix spends 9 sweeps x 115 rounds x 58 rotations on 86 eigenvalues that are
// zero to round-off (124 of the kernel's 181 ms per 4096 goals).  Instead: an orthonormal basis Q of
// range(B) by column-pivoted Gram-Schmidt with re-orthogonalisation (stops when the largest
// residual column is below 1e-13 |B|_F), M = Q^T B Q (r x r), Jacobi on M, eigenvectors Q W.  The
// nonzero eigenpairs are those of B to round-off (eps |B|, like LAPACK's); the null space comes out
// as exact zeros instead of +-1e-16 noise -- generate_initialization() keeps eigenvalues > 1e-8 only
// (dgp.py:150-171), so nothing downstream sees the difference beyond 1e-8-sized noise columns.
// Returns the rank r, or -1 if it exceeds PREP_RMAX (A is then restored from the copy and the caller
// runs the full decomposition).
constexpr int PREP_COMPRESS_MIN_N = 48;
constexpr int PREP_RMAX = 64;

// A: N x N (row stride N; LDS or global), destroyed.  Bc: global copy of B (out).  Qt: global, row m =
// basis vector m (row stride N).  nrm, qv, cj: LDS, N doubles each (the kernel's MAXN).
__device__ inline int range_basis_blk(double *A, int N, double *Bc, double *Qt, double *nrm, double *qv, double *cj,
                                      double *red, int tid) {
  const int NN = N * N;
  for (int e = tid; e < NN; e += PREP_NT) Bc[e] = A[e];
  if (tid < N) {
    double s = 0.0;
    for (int i = 0; i < N; ++i) s = fma(A[i * N + tid], A[i * N + tid], s);
    nrm[tid] = s;
  }
  __syncthreads();
  double fro2 = 0.0;
  for (int j = 0; j < N; ++j) fro2 += nrm[j];
  const double tol2 = 1e-26 * fro2;
  const int e_i0 = tid / N, e_j0 = tid - e_i0 * N, e_dq = PREP_NT / N, e_dr = PREP_NT - e_dq * N;   // element stepping
  int r = 0;
  for (;;) {
    double best = -1.0;
    int p = 0;
    for (int j = 0; j < N; ++j) {
      const double v = nrm[j];
      if (v > best) {
        best = v;
        p = j;
      }
    }
    if (!(best > tol2)) break;
    if (r >= PREP_RMAX) {      // not a low-rank matrix: hand the original back
      __syncthreads();
      for (int e = tid; e < NN; e += PREP_NT) A[e] = Bc[e];
      __syncthreads();
      return -1;
    }
    // candidate: the residual column, re-orthogonalised twice against the basis so far (the deflated column
    // is orthogonal to it up to eps x the growth of the pivots, 1e11 here)
    double qi = tid < N ? A[tid * N + p] : 0.0;
    for (int pass = 0; pass < 2 && r > 0; ++pass) {
      if (tid < N) qv[tid] = qi;
      __syncthreads();
      {   // d_m = <q_m, q>: eight threads per basis vector
        const int m = tid >> 3, part = tid & 7;
        double d = 0.0;
        if (m < r)
          for (int i = part; i < N; i += 8) d = fma(Qt[m * N + i], qv[i], d);
        d += dpp_f64<0xB1>(d);
        d += dpp_f64<0x4E>(d);
        d += dpp_f64<0x141>(d);   // row_half_mirror: lanes 0..7 of a group of 8
        if (part == 0 && m < PREP_RMAX) cj[m] = m < r ? d : 0.0;
      }
      __syncthreads();
      if (tid < N)
        for (int m = 0; m < r; ++m) qi = fma(-cj[m], Qt[m * N + tid], qi);
      __syncthreads();
    }
    const double nn = prep_block_sum(qi * qi, red, tid);
    if (!(nn > tol2)) {        // numerically dependent on the basis: drop the column
      if (tid == 0) nrm[p] = 0.0;
      __syncthreads();
      continue;
    }
    qi *= 1.0 / sqrt(nn);
    if (tid < N) {
      qv[tid] = qi;
      Qt[r * N + tid] = qi;
    }
    __syncthreads();
    for (int j0 = 0; j0 < N; j0 += PREP_NT / 4) {   // c_j = <q, A[:, j]>: four threads per column, 128 columns at a time
      const int j = j0 + (tid >> 2), part = tid & 3;
      double c = 0.0;
      if (j < N)
        for (int i = part; i < N; i += 4) c = fma(qv[i], A[i * N + j], c);
      c += dpp_f64<0xB1>(c);
      c += dpp_f64<0x4E>(c);
      if (part == 0 && j < N) cj[j] = c;
    }
    __syncthreads();
    {   // deflate: A -= q c^T
      int i = e_i0, j = e_j0;
      for (int e = tid; e < NN; e += PREP_NT) {
        A[i * N + j] = fma(-qv[i], cj[j], A[i * N + j]);
        i += e_dq;
        j += e_dr;
        if (j >= N) {
          j -= N;
          ++i;
        }
      }
    }
    if (tid < N) nrm[tid] = tid == p ? 0.0 : fmax(fma(-cj[tid], cj[tid], nrm[tid]), 0.0);
    ++r;
    __syncthreads();
  }
  return r;
}

// B (global copy Bc) in the basis Qt: Tt[l] = B q_l (global, row stride N), M = Q^T B Q symmetrised into the
// leading r x r block of A (row stride N), W = identity (global, row stride N).
__device__ inline void compress_to_basis_blk(double *A, int N, int r, const double *Bc, const double *Qt, double *Tt,
                                             double *W, int tid) {
  for (int e = tid; e < r * N; e += PREP_NT) {      // e = l * N + i
    const int l = e / N, i = e - l * N;
    double s = 0.0;
    for (int j = 0; j < N; ++j) s = fma(Bc[i * N + j], Qt[l * N + j], s);
    Tt[e] = s;
  }
  __syncthreads();
  for (int e = tid; e < r * r; e += PREP_NT) {
    const int k = e / r, l = e - k * r;
    double s = 0.0, t = 0.0;
    for (int i = 0; i < N; ++i) {
      s = fma(Qt[k * N + i], Tt[l * N + i], s);
      t = fma(Qt[l * N + i], Tt[k * N + i], t);
    }
    A[k * N + l] = 0.5 * (s + t);
    W[k * N + l] = k == l ? 1.0 : 0.0;
  }
  __syncthreads();
}

// eigenvectors of B from those of M: V[i][k] = sum_m Q[i][m] W[m][k] (k < r); the columns beyond are zero and are
// NOT written; the eigenvalues stay on A's diagonal (zeros beyond r)
__device__ inline void expand_from_basis_blk(double *A, double *V, int N, int r, const double *Qt, const double *W,
                                             int tid) {
  for (int e = tid; e < r * N; e += PREP_NT) {      // e = k * N + i: consecutive threads take consecutive rows i
    const int k = e / N, i = e - k * N;                // (columns k >= r are zero and are not stored: the caller's nc)
    double s = 0.0;
    for (int m = 0; m < r; ++m) s = fma(Qt[m * N + i], W[m * N + k], s);
    V[i * N + k] = s;
  }
  if (tid >= r && tid < N) A[tid * N + tid] = 0.0;
  __syncthreads();
}

// A_LDS: the matrix being worked on (the upper bounds during the N Floyd-Warshall rounds, then the
// Gram matrix = Jacobi / Householder work matrix, then the scatter matrix) lives in the dynamic LDS
// segment instead of the global slab -- the kernel is bound by the L2 traffic of the Jacobi rounds,
// two thirds of which are the work matrix's (N <= 123: 8 N^2 bytes next to the 41 KB of static LDS)
// MAXN: rows the small LDS arrays hold -- 128 (PREP_MAXN), or 256 (PREP_BIGN, round 6) for graphs of 129 .. 255 nodes
// (UR10 + a scene of more than 112 spheres: N = 216 with table_environment(12, 14)), whose work matrix lives in the
// slab (A_LDS = false) and which ALWAYS go through the range compression (their slab has a sixth matrix for the copy
// of the Gram matrix the compression needs; with the work matrix in LDS that copy takes the work matrix's slot).
// Such a graph has few nodes outside its rigid anchor clique -- the solve kernel that takes it asks for that -- so
// the rank of its Gram matrix is small (27 at N = 216) and every Jacobi of the kernel runs on <= 64 rows; the paths
// that would need more (a fallback to the full N x N decomposition, more than 128 MDS columns) exist in slow form
// (rotations applied round by round) or are refused (K_out = -1, Y_init = NaN: the solve reports stop = 2).
constexpr int PREP_BIGN = 256;
template <bool A_LDS, int MAXN = PREP_MAXN>
__global__ void __launch_bounds__(PREP_NT) prep_block_kernel(PrepArgs a, double *ws) {
  static_assert(MAXN == PREP_MAXN || (MAXN == PREP_BIGN && !A_LDS), "");
  constexpr int SLAB = MAXN > PREP_MAXN ? 6 : 5;      // N x N matrices per workgroup
  extern __shared__ __attribute__((aligned(16))) double sh_A[];
  __shared__ double gd[2 * PREP_MAXA + 16];     // (several end effectors: checked at attach)
  __shared__ double cs[2 * (MAXN / 2)];
  __shared__ double ev[MAXN], sg[MAXN], red[PREP_NT / 64];
  __shared__ int pq[MAXN / 2], rk[MAXN], cinv[MAXN];
  __shared__ double dl[PREP_PC * PREP_MAXN];    // edge differences of PREP_PC pairs; Jacobi rotation log
  static_assert(JLOG_DOUBLES <= PREP_PC * PREP_MAXN, "the rotation log shares the scatter phase's tile buffer");
  const PipeConst &pc = a.pc;
  const int N = pc.N, K = pc.K, NN = N * N, tid = threadIdx.x;
  double *Ug = ws + (size_t)blockIdx.x * SLAB * NN;
  double *L = Ug + NN;                           // lower bounds
  double *Ag = L + NN;
  // the LDS buffer holds the upper bounds during bound smoothing (N Floyd-Warshall rounds over the
  // whole matrix) and becomes the work matrix once the Gram matrix is formed
  double *U = A_LDS ? sh_A : Ug;                 // upper bounds -> ub
  double *A1 = Ag;                               // max-plus intermediate
  double *A = A_LDS ? sh_A : Ag;                 // work matrix (from the Gram matrix on)
  double *V = L + 2 * NN;                        // eigenvectors / temp
  double *X = V + NN;                            // MDS factor
  const int D = K + 1;
  // The LOWER table differs between goals in the goal-node entries only, and those sit at the same places for every
  // goal: this workgroup's copy is written ONCE per launch and patched per goal (until round 5 it was rewritten
  // whole for every goal, and its buffer then reused as scratch: 107 KB of stores per table-scene goal).
  for (int e = tid; e < NN; e += PREP_NT) {
    const double lo = pc.base_lower[e];
    L[e] = ((e / N) == (e % N)) ? 0.0 : (lo == lo ? lo : -INFINITY);
  }

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const double *Tg = a.T_goal + (size_t)b * D * D * pc.n_ee;
    const int n_gd = 2 * pc.n_ee * pc.n_anchor + pc.n_gg;
    for (int e = tid; e < NN; e += PREP_NT) {
      const double up = pc.base_upper[e];
      U[e] = ((e / N) == (e % N)) ? 0.0 : (up == up ? up : INFINITY);
    }
    __syncthreads();
    for (int idx = tid; idx < n_gd; idx += PREP_NT) {
      int an, gn;
      const double d = goal_distance(pc, Tg, idx, an, gn);
      gd[idx] = d;
      if (an >= 0 && gn >= 0) {      // (an inert goal slot pins nothing)
        U[an * N + gn] = U[gn * N + an] = d;
        L[an * N + gn] = L[gn * N + an] = d;
      }
    }
    __syncthreads();
    for (int t = tid; t < pc.T; t += PREP_NT) {
      const int src = pc.term_src[t];
      const double g = src >= 0 ? gd[src] : 0.0;
      a.targets[(size_t)b * pc.T + t] = src >= 0 ? g * g : pc.term_static[t];
    }
#ifdef GIK_DEV
    if (a.stop_phase == 1) continue;
#endif
    // ---- bound smoothing (see prep_wave_kernel)
    for (int m = 0; m < N; ++m) {
      for (int e = tid; e < NN; e += PREP_NT) {
        const int i = e / N, j = e - i * N;
        const double cand = U[i * N + m] + U[m * N + j];
        if (cand < U[e]) U[e] = cand;
      }
      __syncthreads();
    }
#ifdef GIK_DEV
    if (a.stop_phase == 2) continue;
#endif
    // A1[u][b] = max_a (L[a][b] - ub[u][a]), then lb[u][v] = max(0, max_q A1[u][q] - ub[q][v]) and, from it,
    // D_rand = (lb + 0.9 (ub - lb))^2 -- by BLOCKS OF ROWS u: the rows of the max-plus intermediate live in the LDS
    // tile buffer (4096 doubles: 35 rows at N = 116) and never reach the slab, and lb goes straight into D_rand
    // (round 5: two N x N stores and loads per goal less; same operations per element; the diagnostics still get lb)
    {
      const int RB = (PREP_PC * PREP_MAXN) / N;      // rows per block
      for (int u0 = 0; u0 < N; u0 += RB) {
        const int nr = min(RB, N - u0);
        for (int idx = tid; idx < nr * N; idx += PREP_NT) {
          const int ur = idx / N, bb = idx - ur * N, u = u0 + ur;
          double best = -INFINITY;
          for (int q = 0; q < N; ++q) best = fmax(best, L[q * N + bb] - U[u * N + q]);
          dl[idx] = best;
        }
        __syncthreads();
        for (int idx = tid; idx < nr * N; idx += PREP_NT) {
          const int ur = idx / N, v = idx - ur * N, e = (u0 + ur) * N + v;
          double best = 0.0;
          for (int q = 0; q < N; ++q) best = fmax(best, dl[ur * N + q] - U[q * N + v]);
          if (a.dbg_lb) {
            a.dbg_lb[(size_t)b * NN + e] = best;
            a.dbg_ub[(size_t)b * NN + e] = U[e];
          }
          const double d = best + 0.9 * (U[e] - best);
          X[e] = d * d;
        }
        __syncthreads();
      }
    }
#ifdef GIK_DEV
    if (a.stop_phase == 3) continue;
#endif
    __syncthreads();
    if (tid < N) {
      double s = 0.0;
      for (int j = 0; j < N; ++j) s += X[tid * N + j];
      ev[tid] = s / N;
    }
    __syncthreads();
    double mean = 0.0;
    for (int j = 0; j < N; ++j) mean += ev[j];
    mean /= N;
    for (int e = tid; e < NN; e += PREP_NT) {
      const int i = e / N, j = e - i * N;
      A[e] = -0.5 * (X[e] - ev[i] - ev[j] + mean);
    }
    __syncthreads();
#ifdef GIK_DEV
    if (a.stop_phase == 4) continue;
#endif
    // nc: columns of V / X that are STORED from here on.  With the range compression only the r eigenvectors of the
    // range exist -- the other N - r columns are exact zeros -- and round 5 no longer writes, scales, reads or clears
    // them (four N x N stores and three loads per goal): whoever reads X asks cinv[] < nc first.
    int nc = N;
    {
      int rnk = -1;
      if ((A_LDS || MAXN > PREP_MAXN) && N >= PREP_COMPRESS_MIN_N && (!a.no_compress || MAXN > PREP_MAXN)) {
        // (global buffers free at this point: the slab's upper bounds Ug -- their last reader was D_rand --, the
        // work-matrix slot A1 when the work matrix is in LDS, V and X; NOT the lower-bound table L, which stays for the
        // next goal; graphs of 124 .. 128 nodes, whose work matrix sits in a five-matrix slab, keep the full
        // decomposition; graphs beyond 128 nodes have a sixth matrix for the copy)
        double *Bc = A_LDS ? A1 : Ug + 5 * NN, *Qt = X, *Tt = V, *W = Ug;   // (V is written by expand_from_basis_blk, after Tt's last use)
        rnk = range_basis_blk(A, N, Bc, Qt, ev, sg, cs, red, tid);
        if (rnk >= 0) {
          compress_to_basis_blk(A, N, rnk, Bc, Qt, Tt, W, tid);
          if (rnk > 1) jacobi_blk(A, W, N, a.sweeps, cs, pq, red, tid, rnk, dl);
          __syncthreads();
          expand_from_basis_blk(A, V, N, rnk, Qt, W, tid);
          nc = rnk;
        }
      }
      if (rnk < 0) {
        for (int e = tid; e < NN; e += PREP_NT) V[e] = ((e / N) == (e % N)) ? 1.0 : 0.0;
        __syncthreads();
        jacobi_blk(A, V, N, a.sweeps, cs, pq, red, tid, -1, dl);
      }
    }
#ifdef GIK_DEV
    if (a.stop_phase == 5) continue;
#endif
    // ---- factor(): clip, scale by sqrt(lambda), order descending
    __syncthreads();
    if (tid < N) ev[tid] = A[tid * N + tid];
    if (a.dbg_eig && tid < N) a.dbg_eig[((size_t)b * 3 + 0) * N + tid] = A[tid * N + tid];
    __syncthreads();
    if (tid < N) {
      rk[tid] = desc_rank(ev, N, tid);
      double big = 0.0, sgn = 1.0;
      if (tid < nc)                      // (a column that is not stored is zero: no entry, sign +)
        for (int r = 0; r < N; ++r) {
          const double v = V[r * N + tid];
          if (fabs(v) > big) {
            big = fabs(v);
            sgn = v < 0.0 ? -1.0 : 1.0;
          }
        }
      sg[tid] = sgn * sqrt(fmax(ev[tid], 0.0));
    }
    __syncthreads();
    if (tid < N) cinv[rk[tid]] = tid;    // column j of X came from eigenvector cinv[j]: stored iff cinv[j] < nc
    for (int e = tid; e < N * nc; e += PREP_NT) {
      const int r = e / nc, c = e - r * nc;
      X[r * N + rk[c]] = V[r * N + c] * sg[c];
    }
    __syncthreads();
    // ---- MDS(): K = #eigenvalues > eps of the symmetric matrix read from the LOWER triangle of x
    for (int e = tid; e < NN; e += PREP_NT) {
      const int i = e / N, j = e - i * N;
      const int rr = i >= j ? i : j, cj = i >= j ? j : i;
      A[e] = cinv[cj] < nc ? X[rr * N + cj] : 0.0;
    }
    __syncthreads();
#ifdef GIK_DEV
    if (a.stop_phase == 6) continue;
#endif
    if (a.dbg_eig) {   // diagnostics only: the spectrum itself, then A is rebuilt for the count
      jacobi_blk(A, nullptr, N, a.sweeps, cs, pq, red, tid);
      __syncthreads();
      if (tid < N) a.dbg_eig[((size_t)b * 3 + 1) * N + tid] = A[tid * N + tid];
      __syncthreads();
      for (int e = tid; e < NN; e += PREP_NT) {
        const int i = e / N, j = e - i * N;
        const int rr = i >= j ? i : j, cj = i >= j ? j : i;
        A[e] = cinv[cj] < nc ? X[rr * N + cj] : 0.0;
      }
      __syncthreads();
    }
    const int Kc = count_eigs_above_blk(A, N, 1e-8, ev, sg, red, tid);
    if (MAXN > PREP_MAXN && Kc > PREP_MAXN) {      // (more MDS columns than the scatter phase stages: see the kernel's head)
      if (a.K_out && tid == 0) a.K_out[b] = -1;
      for (int t = tid; t < N * K; t += PREP_NT) a.Y_init[(size_t)b * N * K + t] = __builtin_nan("");
      __syncthreads();
      continue;
    }
    if (a.K_out && tid == 0) a.K_out[b] = Kc;
    __syncthreads();
#ifdef GIK_DEV
    if (a.stop_phase == 7) continue;
#endif
    // ---- linear_projection (dgp.py:174-183): X[:, Kc:] = 0 -- not cleared in memory: every read below stops at Kc
    {
      // S[r][c] = sum_p d_p[r] d_p[c], d_p = X[i_p] - X[j_p]: 5604 pairs x Kc^2 at N = 116.  The
      // edge differences of 32 pairs at a time are staged in LDS; a thread owns row r = tid / 4 and
      // every fourth column from tid % 4, so that one LDS read of d_p[r] serves all its entries
      // and the running sums sit in registers with constant indices (same order over p per entry
      // as the wave kernel; columns >= Kc of X are zero, so the whole N x N matrix is formed)
      constexpr int CPT = PREP_MAXN / 4;   // columns per thread
      const int r = tid >> 2, c0 = tid & 3;
      double acc[CPT];
#pragma unroll
      for (int q = 0; q < CPT; ++q) acc[q] = 0.0;
      // Columns >= Kc of X are zero, hence so are the edge differences there and every entry of S
      // outside its leading Kc x Kc block (Kc ~ 30 of 116 on the table scene): only that block is
      // accumulated -- same products in the same order for the entries that are not zero.
      const int nq = (Kc + 3) >> 2;                  // column groups c0 + 4 q that reach below Kc
      const bool row_live = r < Kc;
      const int kw = 4 * nq;                         // staged columns: [0, Kc) and the zero pad up to 4 nq
      // The work-matrix buffer is free until S is written: it holds X's first kw columns (row
      // stride kw) for the edge differences -- out of the slab these were two dependent loads
      // through the L2 / Infinity Cache per staged value, 55 ms of the 236 per 4096 table-scene goals.
      double *XL = A;
      for (int e = tid; e < N * kw; e += PREP_NT) {
        const int i = e / kw, c = e - i * kw;
        XL[e] = (c < Kc && cinv[c] < nc) ? X[i * N + c] : 0.0;
      }
      __syncthreads();
      for (int p0 = 0; p0 < pc.n_pairs; p0 += PREP_PC) {
        const int np_ = min(PREP_PC, pc.n_pairs - p0);
        for (int it = tid; it < np_ * kw; it += PREP_NT) {
          const int pp = it / kw, c = it - pp * kw;
          const int i = pc.pair_i[p0 + pp], j = pc.pair_j[p0 + pp];
          dl[pp * PREP_MAXN + c] = XL[i * kw + c] - XL[j * kw + c];
        }
        __syncthreads();
        if (row_live) {
          // column groups outermost (one uniform branch per group of four columns, the pair loop
          // inside it): a guard per column made every product wait for its own LDS read
#pragma unroll
          for (int g = 0; g < CPT / 4; ++g) {
            if (4 * g >= nq) continue;
#pragma unroll 4
            for (int pp = 0; pp < np_; ++pp) {
              const double dr = dl[pp * PREP_MAXN + r];
#pragma unroll
              for (int q = 4 * g; q < 4 * g + 4; ++q) acc[q] = fma(dr, dl[pp * PREP_MAXN + c0 + 4 * q], acc[q]);
            }
          }
        }
        __syncthreads();
      }
#pragma unroll
      for (int q = 0; q < CPT; ++q) {
        const int c = c0 + 4 * q;
        if (r < N && c < N) A[r * N + c] = 2.0 * acc[q];  // the reference sums both (i,j) and (j,i)
      }
    }
    // the K x K Jacobi touches the leading n2 x n2 block of V only (outside it V is the identity, implicitly)
    const int n2 = Kc > 1 ? Kc : 2;
    __syncthreads();
    for (int e = tid; e < n2 * n2; e += PREP_NT) {
      const int i = e / n2, j = e - i * n2;
      V[i * N + j] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
#ifdef GIK_DEV
    if (a.stop_phase == 8) continue;
#endif
    jacobi_blk(A, V, N, a.sweeps, cs, pq, red, tid, n2, dl);
    __syncthreads();
#ifdef GIK_DEV
    if (a.stop_phase == 9) continue;
#endif
    if (tid < N) ev[tid] = (tid < Kc) ? A[tid * N + tid] : -INFINITY;
    if (a.dbg_eig && tid < N) a.dbg_eig[((size_t)b * 3 + 2) * N + tid] = (tid < Kc) ? A[tid * N + tid] : 0.0;
    __syncthreads();
    if (tid < N) {
      rk[tid] = desc_rank(ev, N, tid);
      double big = 0.0, sgn = 1.0;
      if (tid < n2)                      // (outside the block column tid of V is e_tid: sign +)
        for (int r = 0; r < n2; ++r) {
          const double v = V[r * N + tid];
          if (fabs(v) > big) {
            big = fabs(v);
            sgn = v < 0.0 ? -1.0 : 1.0;
          }
        }
      sg[tid] = sgn;
    }
    __syncthreads();
    for (int t = tid; t < N * K; t += PREP_NT) {
      const int r = t / K, dcol = t - r * K;
      int col = 0;
      for (int c = 0; c < N; ++c) col = (rk[c] == dcol) ? c : col;
      // Y = X[:, :Kc] V[:Kc, col]: the columns of X beyond Kc are zero, V is the identity outside its n2 x n2 block
      double s = 0.0;
      if (col < n2)
        for (int c = 0; c < Kc; ++c)
          if (cinv[c] < nc) s += X[r * N + c] * V[c * N + col];
      a.Y_init[(size_t)b * N * K + t] = s * sg[col];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
struct RecoverArgs {
  PipeConst pc;
  const double *Y;       // [B][N*K]
  const double *T_goal;  // [B][(K+1)^2]
  double *q;             // [B][n]
  double *pos_err;       // [B]
  double *rot_err;       // [B]
  int B;
};

__device__ inline void mat4_mul(const double *A, const double *B, double *C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = (j == 3) ? A[i * 4 + 3] : 0.0;
      for (int t = 0; t < 3; ++t) s += A[i * 4 + t] * B[t * 4 + j];
      C[i * 4 + j] = s;
    }
  C[12] = C[13] = C[14] = 0.0;
  C[15] = 1.0;
}
__device__ inline void mat4_inv_rigid(const double *A, double *C) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) C[i * 4 + j] = A[j * 4 + i];
    C[i * 4 + 3] = -(A[0 * 4 + i] * A[3] + A[1 * 4 + i] * A[7] + A[2 * 4 + i] * A[11]);
  }
  C[12] = C[13] = C[14] = 0.0;
  C[15] = 1.0;
}
__device__ inline double wrap_pi(double e) {
  const double twopi = 6.283185307179586;
  double m = fmod(e + 3.141592653589793, twopi);
  if (m < 0.0) m += twopi;  // numpy mod takes the sign of the divisor
  return m - 3.141592653589793;
}

// (launched with 64 threads per block; without the bound the compiler budgets for 1024 and 128 VGPRs: 252 B of scratch)
__global__ void __launch_bounds__(64) recover_kernel(RecoverArgs a)
#ifndef GIK_DEFINE_PLAIN_KERNELS
    ;      // (defined in gik_k_prep.hip)
#else
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const PipeConst &pc = a.pc;
  const int N = pc.N, K = pc.K, n = pc.n_joints;
  const double *P = a.Y + (size_t)b * N * K;
  double *q = a.q + (size_t)b * n;
  if (K == 3) {
    const double *p0 = P + pc.p_idx[0] * 3;
    // base frame from the recovered anchors: R = [x^, -y^, z^]  (graph_revolute.py:263-279)
    double R[9];
    const int src[3] = {pc.x_idx, pc.y_idx, pc.q_idx[0]};
    for (int c = 0; c < 3; ++c) {
      double v[3], nr = 0.0;
      for (int t = 0; t < 3; ++t) {
        v[t] = P[src[c] * 3 + t] - p0[t];
        nr += v[t] * v[t];
      }
      nr = sqrt(nr);
      const double sc = (nr == 0.0 ? 1.0 : 1.0 / nr) * (c == 1 ? -1.0 : 1.0);
      for (int t = 0; t < 3; ++t) R[t * 3 + c] = v[t] * sc;
    }
    double worst_p = 0.0, worst_r = 0.0;
    // one walk from the root per end effector (graph_revolute.py:285-316; joints shared by several
    // paths get the same angle from each)
    for (int e = 0; e < pc.n_ee; ++e) {
      const double *Tg = a.T_goal + ((size_t)b * pc.n_ee + e) * 16;
      const int *path = pc.ee_path + e * (n + 1);
      double Tp[16], Trel[16], tmp[16], inv[16];
      for (int t = 0; t < 16; ++t) Tp[t] = pc.T0[t];  // T[ROOT] = robot.T_base
      double th = 0.0;
      int cur = 0, last = 0;
      for (int k = 1; k <= n && path[k] >= 0; ++k) last = k;
      for (int k = 1; k <= last; ++k) {
        const int pred = path[k - 1];
        cur = path[k];
        mat4_inv_rigid(pc.T0 + pred * 16, inv);
        mat4_mul(inv, pc.T0 + cur * 16, Trel);  // T_rel = T_prev_0^-1 T_0
        // qs_0 = (T_prev_0^-1 T_0 trans_z(a)).trans
        double qs0[2];
        for (int t = 0; t < 2; ++t) qs0[t] = Trel[t * 4 + 3] + Trel[t * 4 + 2] * pc.axis_length;
        const double *pc_ = P + pc.p_idx[cur] * 3, *qc = P + pc.q_idx[cur] * 3;
        double dq[3], nr = 0.0;
        for (int t = 0; t < 3; ++t) {
          dq[t] = qc[t] - pc_[t];
          nr += dq[t] * dq[t];
        }
        nr = sqrt(nr);
        double qn[3], qb[3];
        for (int t = 0; t < 3; ++t) qn[t] = pc_[t] + dq[t] / nr - p0[t];
        for (int t = 0; t < 3; ++t) qb[t] = R[0 * 3 + t] * qn[0] + R[1 * 3 + t] * qn[1] + R[2 * 3 + t] * qn[2];
        double qs[2];
        for (int t = 0; t < 2; ++t)
          qs[t] = Tp[0 * 4 + t] * (qb[0] - Tp[3]) + Tp[1 * 4 + t] * (qb[1] - Tp[7]) +
                  Tp[2 * 4 + t] * (qb[2] - Tp[11]);
        th = atan2(qs0[0] * qs[1] - qs0[1] * qs[0], qs0[0] * qs[0] + qs0[1] * qs[1]);  // :308
        q[cur - 1] = th;
        if (k == last) break;  // keep T_prev of the last joint for the T_final correction
        const double c = cos(th), s = sin(th);
        double Rz[16] = {c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        mat4_mul(Tp, Rz, tmp);
        mat4_mul(tmp, Trel, Tp);  // :310
      }
      // T[ee] with the uncorrected last angle
      double Tee[16];
      {
        const double c = cos(th), s = sin(th);
        double Rz[16] = {c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        mat4_mul(Tp, Rz, tmp);
        mat4_mul(tmp, Trel, Tee);
      }
      if ((pc.last_along_z >> e) & 1) {  // :314-316
        // T_final expressed in the recovered base frame is the goal itself (T_base = identity
        // re-basing happened at load time); T_th = T[ee]^-1 T_final
        mat4_inv_rigid(Tee, inv);
        mat4_mul(inv, Tg, tmp);
        th = wrap_pi(th + atan2(tmp[4], tmp[0]));
        q[cur - 1] = th;
        const double c = cos(th), s = sin(th);
        double Rz[16] = {c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        mat4_mul(Tp, Rz, tmp);
        mat4_mul(tmp, Trel, Tee);
      }
      // pose error of FK(q) = T[ee] against the goal
      double dp = 0.0, tr = 0.0;
      for (int t = 0; t < 3; ++t) {
        const double df = Tg[t * 4 + 3] - Tee[t * 4 + 3];
        dp += df * df;
        for (int u = 0; u < 3; ++u) tr += Tg[t * 4 + u] * Tee[t * 4 + u];  // trace(Rg Rs^T)
      }
      worst_p = fmax(worst_p, sqrt(dp));
      worst_r = fmax(worst_r, acos(fmin(1.0, fmax(-1.0, 0.5 * tr - 0.5))));
    }
    a.pos_err[b] = worst_p;     // (several end effectors: the worst of them)
    a.rot_err[b] = worst_r;
  } else {
    const double *Tg = a.T_goal + (size_t)b * 9 * pc.n_ee;
    // best_fit_transform of (p0, x, y) onto ((0,0), (-1,0), (0,1)) without reflection handling
    const double *A0 = P + pc.p_idx[0] * 2, *A1 = P + pc.x_idx * 2, *A2 = P + pc.y_idx * 2;
    const double ca[2] = {(A0[0] + A1[0] + A2[0]) / 3.0, (A0[1] + A1[1] + A2[1]) / 3.0};
    const double Bp[3][2] = {{0, 0}, {-1, 0}, {0, 1}};
    const double cb[2] = {-1.0 / 3.0, 1.0 / 3.0};
    const double *Ap[3] = {A0, A1, A2};
    double H[4] = {0, 0, 0, 0};  // H = AA^T BB
    for (int t = 0; t < 3; ++t)
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) H[i * 2 + j] += (Ap[t][i] - ca[i]) * (Bp[t][j] - cb[j]);
    // R = V U^T (H = U S V^T) = orthogonal polar factor of H^T; M = H^T
    const double M00 = H[0], M01 = H[2], M10 = H[1], M11 = H[3];
    const double det = M00 * M11 - M01 * M10;
    double Rm[4];
    if (det >= 0.0) {
      const double ang = atan2(M10 - M01, M00 + M11);
      Rm[0] = cos(ang); Rm[1] = -sin(ang); Rm[2] = sin(ang); Rm[3] = cos(ang);
    } else {
      const double ang = atan2(M10 + M01, M00 - M11);
      Rm[0] = cos(ang); Rm[1] = sin(ang); Rm[2] = sin(ang); Rm[3] = -cos(ang);
    }
    // One walk from the root per end effector (a chain: the one path 0, 1, ..., n; planar trees, round 6: joints shared
    // by several paths get the same angle from each -- graph_planar.py:147-176 walks the structure edges with R[u] per
    // node, which along a path is the running product kept here).  FK of the recovered angles along the same path:
    // T_i = T_{i-1} Rz(q_i) T0_{i-1}^-1 T0_i  (robot_planar.py:62-80); pose error: the worst end effector's.
    double worst_p = 0.0, worst_r = 0.0;
    for (int e = 0; e < pc.n_ee; ++e) {
      const double *Te = Tg + (size_t)e * 9;
      const int *path = pc.ee_path + e * (n + 1);
      double Rc[4] = {1, 0, 0, 1};
      double Fr[4] = {pc.T0[0], pc.T0[1], pc.T0[3], pc.T0[4]}, Ft[2] = {pc.T0[2], pc.T0[5]};
      for (int k = 1; k <= n && path[k] >= 0; ++k) {
        const int pred = path[k - 1], i = path[k];
        const double *pu = P + pc.p_idx[pred] * 2, *pv = P + pc.p_idx[i] * 2;
        const double d0 = pv[0] - pu[0], d1 = pv[1] - pu[1];
        double f0 = Rm[0] * d0 + Rm[1] * d1, f1 = Rm[2] * d0 + Rm[3] * d1;
        const double len = sqrt(f0 * f0 + f1 * f1);
        f0 /= len;
        f1 /= len;
        const double s0 = Rc[0] * f0 + Rc[2] * f1, s1 = Rc[1] * f0 + Rc[3] * f1;  // R[u]^T f
        const double th = atan2(s1, s0);
        q[i - 1] = wrap_pi(th);
        const double c = cos(th), s = sin(th);
        const double n0 = Rc[0] * c + Rc[1] * s, n1 = -Rc[0] * s + Rc[1] * c;
        const double n2 = Rc[2] * c + Rc[3] * s, n3 = -Rc[2] * s + Rc[3] * c;
        Rc[0] = n0; Rc[1] = n1; Rc[2] = n2; Rc[3] = n3;
        // T_rel = T0_pred^-1 T0_i
        const double *Ta = pc.T0 + pred * 9, *Tb = pc.T0 + i * 9;
        const double a00 = Ta[0], a01 = Ta[1], a10 = Ta[3], a11 = Ta[4];
        const double dxr = Tb[2] - Ta[2], dyr = Tb[5] - Ta[5];
        const double rr[4] = {a00 * Tb[0] + a10 * Tb[3], a00 * Tb[1] + a10 * Tb[4],
                              a01 * Tb[0] + a11 * Tb[3], a01 * Tb[1] + a11 * Tb[4]};
        const double rt[2] = {a00 * dxr + a10 * dyr, a01 * dxr + a11 * dyr};
        const double cq = cos(q[i - 1]), sq = sin(q[i - 1]);
        // F <- F * Rz(q) * T_rel
        const double g0 = Fr[0] * cq + Fr[1] * sq, g1 = -Fr[0] * sq + Fr[1] * cq;
        const double g2 = Fr[2] * cq + Fr[3] * sq, g3 = -Fr[2] * sq + Fr[3] * cq;
        Ft[0] += g0 * rt[0] + g1 * rt[1];
        Ft[1] += g2 * rt[0] + g3 * rt[1];
        Fr[0] = g0 * rr[0] + g1 * rr[2]; Fr[1] = g0 * rr[1] + g1 * rr[3];
        Fr[2] = g2 * rr[0] + g3 * rr[2]; Fr[3] = g2 * rr[1] + g3 * rr[3];
      }
      const double px = Ft[0], py = Ft[1], phi = atan2(Fr[2], Fr[0]);
      const double dx = Te[2] - px, dy = Te[5] - py;
      worst_p = fmax(worst_p, sqrt(dx * dx + dy * dy));
      const double gphi = atan2(Te[3], Te[0]);
      worst_r = fmax(worst_r, fabs(wrap_pi(gphi - phi)));
    }
    a.pos_err[b] = worst_p;
    a.rot_err[b] = worst_r;
  }
}
#endif


// ---------------------------------------------------------------------------------------------
// Fixed-anchor formulation: glue between the robot-graph pipeline and the anchored
// solve.  One thread per goal, a few hundred flops each.
//   anch_init_kernel   : the initial point of the robot graph (prep_wave_kernel, any rigid frame,
//                        possibly mirrored) is mapped onto the world frame by the orthogonal
//                        Procrustes fit of its anchor rows to the anchors' world positions; the
//                        free rows become the anchored start point, the goal anchors are written out.
//   anch_gather_kernel : free rows + anchors -> full robot-graph point matrix (for recover_kernel).
struct AnchGlueArgs {
  const double *T_goal;        // [B][16]
  const double *Y_full_in;     // [B][full_N*3]   (init: prepare output)
  double *Y_free;              // [B][Nf*3]       (init: out, gather: in)
  double *anchor_goal;         // [B][n_goal*3]   (init: out, gather: in)
  double *Y_full_out;          // [B][full_N*3]   (gather: out)
  const double *anch_const;    // [ANCH_MAXA][4]
  const int *free_full;        // [Nf]
  const int *anchor_full;      // [n_anchor]
  int B, Nf, full_N, n_anchor, n_goal, goal_row0;
  double axis_length;
};

__device__ inline void anchor_world(const AnchGlueArgs &a, const double *Tg, int r, double (&w)[3]) {
  if (r < a.goal_row0) {
    for (int c = 0; c < 3; ++c) w[c] = a.anch_const[r * 4 + c];
  } else {   // goal anchors: p_n = t, q_n = t + axis_length * z  (graph_revolute.py:243-249)
    const double s = (r - a.goal_row0) ? a.axis_length : 0.0;
    for (int c = 0; c < 3; ++c) w[c] = Tg[c * 4 + 3] + s * Tg[c * 4 + 2];
  }
}

__global__ void anch_init_kernel(AnchGlueArgs a)
#ifndef GIK_DEFINE_PLAIN_KERNELS
    ;      // (defined in gik_k_prep.hip)
#else
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const double *Tg = a.T_goal + (size_t)b * 16;
  const double *Y = a.Y_full_in + (size_t)b * a.full_N * 3;
  // centroids and cross-covariance M = sum_i (w_i - cw)(p_i - cp)^T over the anchors
  double cp[3] = {0, 0, 0}, cw[3] = {0, 0, 0};
  for (int r = 0; r < a.n_anchor; ++r) {
    double w[3];
    anchor_world(a, Tg, r, w);
    for (int c = 0; c < 3; ++c) {
      cp[c] += Y[a.anchor_full[r] * 3 + c];
      cw[c] += w[c];
    }
  }
  for (int c = 0; c < 3; ++c) {
    cp[c] /= a.n_anchor;
    cw[c] /= a.n_anchor;
  }
  double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < a.n_anchor; ++r) {
    double w[3];
    anchor_world(a, Tg, r, w);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) M[i * 3 + j] += (w[i] - cw[i]) * (Y[a.anchor_full[r] * 3 + j] - cp[j]);
  }
  // orthogonal polar factor of M (nearest orthogonal matrix, reflections allowed: an MDS embedding
  // is defined up to a mirror image) by the scaled Newton iteration X <- (g X + X^-T / g) / 2
  double X[9];
  for (int t = 0; t < 9; ++t) X[t] = M[t];
  for (int it = 0; it < 30; ++it) {
    const double c00 = X[4] * X[8] - X[5] * X[7], c01 = X[5] * X[6] - X[3] * X[8], c02 = X[3] * X[7] - X[4] * X[6];
    const double c10 = X[2] * X[7] - X[1] * X[8], c11 = X[0] * X[8] - X[2] * X[6], c12 = X[1] * X[6] - X[0] * X[7];
    const double c20 = X[1] * X[5] - X[2] * X[4], c21 = X[2] * X[3] - X[0] * X[5], c22 = X[0] * X[4] - X[1] * X[3];
    const double det = X[0] * c00 + X[1] * c01 + X[2] * c02;
    if (!(fabs(det) > 1e-300)) break;   // degenerate anchors: keep what we have
    const double id = 1.0 / det;
    const double IT[9] = {c00 * id, c01 * id, c02 * id, c10 * id, c11 * id, c12 * id, c20 * id, c21 * id, c22 * id};  // X^-T
    double nx = 0.0, ni = 0.0;
    for (int t = 0; t < 9; ++t) {
      nx += X[t] * X[t];
      ni += IT[t] * IT[t];
    }
    const double g = sqrt(sqrt(ni / nx));
    double diff = 0.0;
    for (int t = 0; t < 9; ++t) {
      const double v = 0.5 * (g * X[t] + IT[t] / g);
      diff += (v - X[t]) * (v - X[t]);
      X[t] = v;
    }
    if (diff < 1e-30) break;
  }
  double *Yf = a.Y_free + (size_t)b * a.Nf * 3;
  for (int f = 0; f < a.Nf; ++f) {
    const double *p = Y + a.free_full[f] * 3;
    for (int i = 0; i < 3; ++i)
      Yf[f * 3 + i] = cw[i] + X[i * 3 + 0] * (p[0] - cp[0]) + X[i * 3 + 1] * (p[1] - cp[1]) + X[i * 3 + 2] * (p[2] - cp[2]);
  }
  for (int r = a.goal_row0; r < a.goal_row0 + a.n_goal; ++r) {
    double w[3];
    anchor_world(a, Tg, r, w);
    for (int c = 0; c < 3; ++c) a.anchor_goal[((size_t)b * a.n_goal + (r - a.goal_row0)) * 3 + c] = w[c];
  }
}
#endif

__global__ void anch_gather_kernel(AnchGlueArgs a)
#ifndef GIK_DEFINE_PLAIN_KERNELS
    ;      // (defined in gik_k_prep.hip)
#else
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const double *Tg = a.T_goal + (size_t)b * 16;
  double *Y = a.Y_full_out + (size_t)b * a.full_N * 3;
  const double *Yf = a.Y_free + (size_t)b * a.Nf * 3;
  for (int f = 0; f < a.Nf; ++f)
    for (int c = 0; c < 3; ++c) Y[a.free_full[f] * 3 + c] = Yf[f * 3 + c];
  for (int r = 0; r < a.n_anchor; ++r) {
    double w[3];
    anchor_world(a, Tg, r, w);
    for (int c = 0; c < 3; ++c) Y[a.anchor_full[r] * 3 + c] = w[c];
  }
}
#endif

}  // namespace gik
