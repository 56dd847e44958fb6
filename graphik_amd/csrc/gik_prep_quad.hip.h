// graphik_amd/csrc/gik_prep_quad.hip.h -- per-goal pre-processing, four goals per wavefront
//
// prep_wave_kernel (gik_prep.hip.h) gives a goal a whole wavefront.  For a graph of at most 16 nodes
// (the planar chains: N = 13) that is 64 lanes for matrices of 13 rows, and every rotation parameter,
// Householder scalar and loop counter is a wave-wide instruction for one number: measured, round 4,
// 34 k instructions per goal, 3.6 of the 5.1 ms of BASELINE configs[4].  Here a wavefront holds FOUR
// goals in the lane layout of the planar solve kernel (gik_quad.hip.h):
//
//   lane l = 16 b + n   ->   goal slot b (0..3), matrix row n (0..15)
//
// (round 5; until then the lane map of the planar SOLVE kernel, l = 16 r + 4 b + i, so that per-goal sums were
// two MFMAs: with it a 16-lane store group held four rows of all four goals and a 32-lane load group eight
// rows of all four, and no placement of the goals makes both conflict-free -- measured SQ_LDS_BANK_CONFLICT /
// SQ_LDS_IDX_ACTIVE = 0.50 with the LDS pipe 69 % busy, the kernel's bound.  With a goal per 16-lane DPP row a
// store group is ONE goal's 16 rows -- odd row stride: 16 different bank pairs -- and a load group two goals'
// rows, disjoint when the goals lie 16 doubles apart modulo 32 (prep_quad_goal_stride); per-goal sums are four
// row_ror steps (goal_sum), which the kernel needs ~40 times per group of goals.)
//
// so a lane owns a ROW of its goal's N x N matrices (LDS, odd row stride), and one instruction stream serves
// four goals.  Same phases and -- per matrix element -- the same
// operations in the same order as prep_wave_kernel (ProblemGraph.from_pose + graph_complete_edges,
// graph_base.py:146-180, dgp.py:124-147; bound smoothing, dgp.py:192-231;
// RiemannianSolver.generate_initialization, riemannian_solver.py:67-75), with differences that stay at
// round-off: the per-goal sums of the Jacobi threshold and of the Householder reflectors are taken in
// another order; the K x K Jacobi of linear_projection runs the round-robin schedule of the LARGEST K
// among the wavefront's four goals (a goal with a smaller K sees extra pairs whose off-diagonal entry
// is exactly zero: skipped); and where the matrix size is a compile-time constant a round forms
// (J^T A) J from rows held in registers instead of J^T (A J) in LDS (quad_jacobi_fixed).  Three
// matrices per goal instead of five (each phase overwrites what the previous one no longer needs):
// 19.6 KB of LDS per wavefront at N = 13, eight wavefronts per CU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gik_prep.hip.h"
#include "gik_quad.hip.h"

namespace gik {

constexpr int PREPQ_MAXN = QUAD_NODES;
// The small per-goal arrays -- ev, sg, hv, hw (16 doubles each), the rotations (c, s) (16 double2) and the ranks (16
// ints) -- form ONE block of 104 doubles per goal slot.  104 = 8 mod 32: what the four goals of a half-wave read at
// the same index (broadcast reads, most of the traffic to these arrays; for the rotations a ds_read_b128) falls into
// four different bank groups.  Until round 5 every array had a slot stride of 16 or 32 doubles: the four goals' (c, s)
// sat on the SAME banks (every read of a rotation 4-way conflicted), the others two by two.
constexpr int PREPQ_VEC_STRIDE = 104;

// sum over the 16 lanes of a goal (a DPP row); every lane gets the total, bit-identical in all 16 (each level adds
// two partial sums that are equal in the lanes exchanging them)
__device__ inline double goal_sum(double v) {
  v += dpp_f64<0x128>(v);   // row_ror:8
  v += dpp_f64<0x124>(v);   // row_ror:4
  v += dpp_f64<0x122>(v);   // row_ror:2
  v += dpp_f64<0x121>(v);   // row_ror:1
  return v;
}

// LDS byte address of a pointer into the kernel's dynamic shared memory
__device__ inline unsigned lds_addr(const void *p) {
  return (unsigned)(__UINTPTR_TYPE__)(const __attribute__((address_space(3))) void *)p;
}
// N consecutive doubles from LDS as N ds_read_b64.  Left to itself the compiler pairs neighbouring loads into
// ds_read2_b64, which the LDS serves at HALF the rate of ds_read_b64 (MI355X_MICROARCH.md, LDS table: 8 cycles per
// 16 bytes and lane against 2 x 2) -- and this kernel is bound by the LDS pipe.
// The loads AND the wait for them are ONE asm statement (early-clobber outputs: none of them may share a register with
// the address, which the later loads still read): the compiler's own waitcnt insertion does not see loads issued from
// inline asm, so with one statement per load it would have been free to copy or spill a destination between "its"
// load and a separate wait -- nothing can be scheduled into the middle of a single statement.
template <int N>
__device__ inline void lds_read_row(double (&v)[N], const double *row);      // written out below for the sizes in use
template <>
__device__ inline void lds_read_row<6>(double (&v)[6], const double *row) {
  const unsigned a = lds_addr(row);
  asm volatile(
      "ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:8\n\tds_read_b64 %2, %6 offset:16\n\t"
      "ds_read_b64 %3, %6 offset:24\n\tds_read_b64 %4, %6 offset:32\n\tds_read_b64 %5, %6 offset:40\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
      : "v"(a)
      : "memory");
}
template <>
__device__ inline void lds_read_row<7>(double (&v)[7], const double *row) {
  const unsigned a = lds_addr(row);
  asm volatile(
      "ds_read_b64 %0, %7\n\tds_read_b64 %1, %7 offset:8\n\tds_read_b64 %2, %7 offset:16\n\t"
      "ds_read_b64 %3, %7 offset:24\n\tds_read_b64 %4, %7 offset:32\n\tds_read_b64 %5, %7 offset:40\n\t"
      "ds_read_b64 %6, %7 offset:48\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6])
      : "v"(a)
      : "memory");
}
template <>
__device__ inline void lds_read_row<8>(double (&v)[8], const double *row) {
  const unsigned a = lds_addr(row);
  asm volatile(
      "ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\t"
      "ds_read_b64 %3, %8 offset:24\n\tds_read_b64 %4, %8 offset:32\n\tds_read_b64 %5, %8 offset:40\n\t"
      "ds_read_b64 %6, %8 offset:48\n\tds_read_b64 %7, %8 offset:56\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(a)
      : "memory");
}
template <>
__device__ inline void lds_read_row<13>(double (&v)[13], const double *row) {
  const unsigned a = lds_addr(row);
  asm volatile(
      "ds_read_b64 %0, %13\n\tds_read_b64 %1, %13 offset:8\n\tds_read_b64 %2, %13 offset:16\n\t"
      "ds_read_b64 %3, %13 offset:24\n\tds_read_b64 %4, %13 offset:32\n\tds_read_b64 %5, %13 offset:40\n\t"
      "ds_read_b64 %6, %13 offset:48\n\tds_read_b64 %7, %13 offset:56\n\tds_read_b64 %8, %13 offset:64\n\t"
      "ds_read_b64 %9, %13 offset:72\n\tds_read_b64 %10, %13 offset:80\n\tds_read_b64 %11, %13 offset:88\n\t"
      "ds_read_b64 %12, %13 offset:96\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12])
      : "v"(a)
      : "memory");
}

// doubles between the matrices of neighbouring goals: N rows of odd stride S = N | 1, padded to 16 mod 32 doubles.
// A ds_read_b64 is served per half-wave = the rows of TWO goals.  Lane = row, one column: goal 0 touches the bank
// pairs S * {0..N-1} (mod 32; distinct: S is odd), goal 1 the set S * {0..N-1} + 16 = S * {16..16+N-1} -- disjoint.
// Lane = column, one row: two runs of N <= 16 consecutive doubles, 16 apart -- disjoint as well.
__host__ __device__ inline int prep_quad_goal_stride(int N) {
  const int NS = N * (N | 1);
  return NS + ((16 - NS % 32) + 32) % 32;
}
__host__ __device__ inline size_t prep_quad_lds_bytes(int N, int n_gd) {
  return sizeof(double) * ((size_t)12 * prep_quad_goal_stride(N) + 4 * (size_t)((n_gd + 1) & ~1) + 4 * PREPQ_VEC_STRIDE);
}

// round-robin (chess tournament) pair m of round r among ne (even) players, rr_pair without its divisions
// (r + m < 2 (ne - 1)); p < q, q = ne - 1 is the bye of an odd matrix size
__host__ __device__ constexpr int quad_rr_p(int ne, int r, int m) {
  const int a = m == 0 ? ne - 1 : (r + m >= ne - 1 ? r + m - (ne - 1) : r + m);
  const int b = m == 0 ? r : (r - m < 0 ? r - m + (ne - 1) : r - m);
  return a < b ? a : b;
}
__host__ __device__ constexpr int quad_rr_q(int ne, int r, int m) {
  const int a = m == 0 ? ne - 1 : (r + m >= ne - 1 ? r + m - (ne - 1) : r + m);
  const int b = m == 0 ? r : (r - m < 0 ? r - m + (ne - 1) : r - m);
  return a < b ? b : a;
}

// rotation of the pair (p, q) of a symmetric matrix, as in jacobi_lds: t = sgn(th) / (|th| + sqrt(th^2 + 1)),
// c = 1 / sqrt(t^2 + 1), s = t c with reciprocal / reciprocal-square-root Newton steps
__device__ inline bool quad_rotation(double app, double aqq, double apq, double thr, double &c, double &s) {
  c = 1.0;
  s = 0.0;
  if (!(fabs(apq) > thr)) return false;
  const double th = (aqq - app) * (0.5 * frcp(apq));
  const double h2 = fma(th, th, 1.0);
  const double t = (th >= 0.0 ? 1.0 : -1.0) * frcp(fabs(th) + h2 * frsqrt(h2));
  c = frsqrt(fma(t, t, 1.0));
  s = t * c;
  return true;
}

// Cyclic Jacobi (see jacobi_lds) on the leading n x n block of this goal's symmetric matrix A (row
// stride S), V accumulates the eigenvectors as columns.  n is wave-uniform; a goal whose matrix is
// non-zero in a smaller leading block only passes the larger n as well.  Lane `i` is row / column i
// of its goal; cs: (c, s) of this goal's pairs.
__device__ inline void quad_jacobi(double *A, double *V, int S, int n, int sweeps, double2 *cs, int slot, int i) {
  const int ne = n + (n & 1), np = ne / 2;
  __builtin_amdgcn_wave_barrier();
  double fro = 0.0;
  if (i < n)
    for (int j = 0; j < n; ++j) fro = fma(A[i * S + j], A[i * S + j], fro);
  const double thr = 1e-16 * sqrt(goal_sum(fro));
  const unsigned long long mine = 0xFFFFull << (16 * slot);
  for (int sw = 0; sw < sweeps; ++sw) {
    bool rotated = false;
    for (int r = 0; r < ne - 1; ++r) {
      bool sig = false;
      if (i < np) {
        const int p = quad_rr_p(ne, r, i), q = quad_rr_q(ne, r, i);
        double c = 1.0, s = 0.0;
        if (q < n) sig = quad_rotation(A[p * S + p], A[q * S + q], A[p * S + q], thr, c, s);
        cs[i] = make_double2(c, s);
      }
      const unsigned long long anysig = __builtin_amdgcn_ballot_w64(sig);
      if (anysig == 0ull) continue;                      // nothing to rotate in this round, in any goal
      rotated = rotated || (anysig & mine) != 0ull;
      __builtin_amdgcn_wave_barrier();
      // (a goal without a significant pair in this round multiplies by c = 1, s = 0: exact)
      for (int m = 0; m < np; ++m) {                     // column phase: lane = row, A and V
        const int p = quad_rr_p(ne, r, m), q = quad_rr_q(ne, r, m);   // (wave-uniform)
        if (q >= n) continue;
        const double2 r2 = cs[m];
        const double c = r2.x, s = r2.y;
        if (i < n) {
          const double ap = A[i * S + p], aq = A[i * S + q];
          A[i * S + p] = c * ap - s * aq;
          A[i * S + q] = s * ap + c * aq;
          const double vp = V[i * S + p], vq = V[i * S + q];
          V[i * S + p] = c * vp - s * vq;
          V[i * S + q] = s * vp + c * vq;
        }
      }
      __builtin_amdgcn_wave_barrier();
      for (int m = 0; m < np; ++m) {                     // row phase on A: lane = column
        const int p = quad_rr_p(ne, r, m), q = quad_rr_q(ne, r, m);
        if (q >= n) continue;
        const double2 r2 = cs[m];
        const double c = r2.x, s = r2.y;
        if (i < n) {
          const double ap = A[p * S + i], aq = A[q * S + i];
          A[p * S + i] = c * ap - s * aq;
          A[q * S + i] = s * ap + c * aq;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (__builtin_amdgcn_ballot_w64(rotated) == 0ull) break;   // every goal of the wavefront has converged
  }
}

// The same for a matrix size N and a row stride S known at compile time (V must come in as the identity), with
// the rows in REGISTERS.  What bounds the loop above is the LDS pipe -- one per CU, shared by all its wavefronts
// -- at 49 KB per wavefront and round (every element of A read and written twice, of V once).  Here a lane keeps
// row i of A and of V in registers for the whole decomposition; the rounds are unrolled, every pair a constant,
// so the column rotations (A J, V J: a row's own entries) are plain register arithmetic.  The row rotations
// (J^T A) need the partner's row: a round is
//   write the own row to LDS -> the smaller index of each pair reads a_pp, a_qq, a_pq from that copy, forms the
//   rotation and publishes (c, -s) under its row index -> every lane reads its pair's entry (the larger index flips
//   the sign of s), the seven (c, s) of the round and the partner's row (plain ds_read_b64: lds_read_row) ->
//   own' = c own -/+ s partner -> own' J in registers,
// 19 KB of LDS traffic per wavefront and round.  A' = (J^T A) J where jacobi_lds forms J^T (A J): the same
// matrix up to round-off.
template <int N, int S>
__device__ inline void quad_jacobi_fixed(double *A, double *V, int sweeps, double2 *cs, int slot, int i) {   // (i by value: made opaque per round)
  constexpr int NE = N + (N & 1), NP = NE / 2;
  __builtin_amdgcn_wave_barrier();
  double ar[N], vr[N];
  {
    const bool row = i < N;
    const double *Ar = A + i * S;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      ar[j] = row ? Ar[j] : 0.0;
      vr[j] = (i == j) ? 1.0 : 0.0;
    }
  }
  cs[i] = make_double2(1.0, 0.0);
  double fro = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) fro = fma(ar[j], ar[j], fro);
  const double thr = 1e-16 * sqrt(goal_sum(fro));
  const unsigned long long mine = 0xFFFFull << (16 * slot);
  bool stale = false;            // (wave-uniform) the LDS copy of A is behind the registers
  for (int sw = 0; sw < sweeps; ++sw) {
    bool rotated = false;
#pragma unroll
    for (int r = 0; r < NE - 1; ++r) {
      // this lane's partner in round r (quad_rr_p / quad_rr_q seen from a row): index NE - 1 meets r, everybody
      // else 2 r - i modulo NE - 1; a partner >= N is the bye of an odd N
      // (the row index goes through an empty asm per round: what a round derives from it -- partner, flags, LDS
      //  addresses -- is then recomputed there, a dozen integer instructions, instead of being hoisted out of the
      //  sweep loop for all 13 rounds at once and spilled: round 4's build kept 123 VGPRs in scratch)
      asm volatile("" : "+v"(i));
      const bool row = i < N;
      double *Ar = A + i * S;
      int j = 2 * r - i;
      j = j < 0 ? j + (NE - 1) : j;
      j = j >= NE - 1 ? j - (NE - 1) : j;
      j = (i == r) ? NE - 1 : j;
      j = (i == NE - 1) ? r : j;
      const bool paired = row && j < N;
      if (stale) {
        if (row) {
#pragma unroll
          for (int c = 0; c < N; ++c) Ar[c] = ar[c];
        }
        stale = false;
      }
      __builtin_amdgcn_wave_barrier();
      bool sig = false;
      if (paired && i < j) {          // the smaller index of a pair forms the rotation and publishes (c, -s) under its row index
        double c, s;
        sig = quad_rotation(Ar[i], A[j * S + j], Ar[j], thr, c, s);
        cs[i] = make_double2(c, -s);
      }
      const unsigned long long anysig = __builtin_amdgcn_ballot_w64(sig);
      if (anysig == 0ull) continue;                        // nothing to rotate in this round, in any goal
      rotated = rotated || (anysig & mine) != 0ull;
      __builtin_amdgcn_wave_barrier();
      // own rotation: the pair's entry, s with the other sign for the larger index; a bye multiplies by (1, 0)
      double2 own = cs[paired ? (i < j ? i : j) : i];
      own.y = (i < j) ? own.y : -own.y;
      if (!paired) own = make_double2(1.0, 0.0);
      double pr[N];
      double2 rc[NP];
      lds_read_row<N>(pr, A + (paired ? j : i) * S);
#pragma unroll
      for (int m = 0; m < NP; ++m)
        if (quad_rr_q(NE, r, m) < N) rc[m] = cs[quad_rr_p(NE, r, m)];
      // rows: p' = c p - s q, q' = c q + s p
#pragma unroll
      for (int c = 0; c < N; ++c) ar[c] = own.x * ar[c] + own.y * pr[c];
      // columns of A' and of V
#pragma unroll
      for (int m = 0; m < NP; ++m)
        if (quad_rr_q(NE, r, m) < N) {
          const int P = quad_rr_p(NE, r, m), Q = quad_rr_q(NE, r, m);   // (constants once the loops are unrolled)
          const double c = rc[m].x, s = -rc[m].y;
          const double ap = ar[P], aq = ar[Q];
          ar[P] = c * ap - s * aq;
          ar[Q] = s * ap + c * aq;
          const double vp = vr[P], vq = vr[Q];
          vr[P] = c * vp - s * vq;
          vr[Q] = s * vp + c * vq;
        }
      stale = true;
    }
    if (__builtin_amdgcn_ballot_w64(rotated) == 0ull) break;
  }
  __builtin_amdgcn_wave_barrier();
  if (i < N) {
    double *Ar = A + i * S, *Vr = V + i * S;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      Ar[j] = ar[j];
      Vr[j] = vr[j];
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// Number of eigenvalues > tau of this goal's symmetric N x N matrix A (destroyed): Householder
// reduction to tridiagonal form + a Sturm count (see count_eigs_above_lds).  hv, hw: 16 doubles each.
__device__ inline int quad_count_eigs_above(double *A, int S, int N, double tau, double *hv, double *hw, int i) {
#pragma unroll 1
  for (int k = 0; k + 2 < N; ++k) {
    const bool mine = i > k && i < N;
    const double x = mine ? A[i * S + k] : 0.0;
    const double x0 = A[(k + 1) * S + k];
    const double s0 = goal_sum(x * x), s1 = goal_sum((mine && i > k + 1) ? x * x : 0.0);
    const bool go = s1 != 0.0;                       // else: column already tridiagonal (goal-uniform)
    const double alpha = x0 > 0.0 ? -sqrt(s0) : sqrt(s0);
    const double vj = mine ? (i == k + 1 ? x - alpha : x) : 0.0;
    const double beta = 1.0 / (s0 - alpha * x0);     // 2 / v'v
    __builtin_amdgcn_wave_barrier();
    hv[i] = vj;
    __builtin_amdgcn_wave_barrier();
    double pj = 0.0;
    if (mine)
      for (int c = k + 1; c < N; ++c) pj = fma(A[i * S + c], hv[c], pj);
    pj *= beta;
    const double Kc = 0.5 * beta * goal_sum(vj * pj);
    const double wj = pj - Kc * vj;
    hw[i] = mine ? wj : 0.0;
    __builtin_amdgcn_wave_barrier();
    if (mine && go) {
      for (int c = k + 1; c < N; ++c) {
        const int e = i * S + c;
        A[e] = A[e] - vj * hw[c] - wj * hv[c];
      }
      if (i == k + 1) A[(k + 1) * S + k] = alpha;    // sub-diagonal entry of the tridiagonal form
    }
    __builtin_amdgcn_wave_barrier();
  }
  int below = 0;
  double q = A[0] - tau;
  below += q < 0.0;
  for (int r = 1; r < N; ++r) {
    const double bb = A[r * S + r - 1];
    if (q == 0.0) q = 1e-300;
    q = A[r * S + r] - tau - bb * bb / q;
    below += q < 0.0;
  }
  return N - below;
}

// NT: the graph's node count where the kernel is compiled for it (13: the 10-link planar chains of BASELINE
// configs[4] -- every loop bound and LDS offset a constant, the Jacobi rounds unrolled), 0: any N <= 16
template <int NT>
__global__ void __launch_bounds__(WAVE, 2) prep_quad_kernel(PrepArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const PipeConst &pc = a.pc;
  const int N = NT ? NT : pc.N, K = pc.K, S = N | 1, D = K + 1;
  const int lane = threadIdx.x, slot = lane >> 4;
  int i = lane & 15;
  const int n_gd = 2 * pc.n_ee * pc.n_anchor + pc.n_gg, n_gd_pad = (n_gd + 1) & ~1;
  // three matrices per goal: M0 = upper bounds -> eigenvectors; M1 = lower-bound table -> lower bounds
  // -> D_rand -> MDS factor X; M2 = work matrix
  const int GS = prep_quad_goal_stride(N);
  double *M0 = smem + (size_t)slot * GS;
  double *M1 = smem + (size_t)(4 + slot) * GS;
  double *M2 = smem + (size_t)(8 + slot) * GS;
  double *gd = smem + (size_t)12 * GS + (size_t)slot * n_gd_pad;
  double *vec = smem + (size_t)12 * GS + 4 * (size_t)n_gd_pad + (size_t)slot * PREPQ_VEC_STRIDE;
  double *ev = vec, *sg = vec + 16, *hv = vec + 32, *hw = vec + 48;
  double2 *cs = reinterpret_cast<double2 *>(vec + 64);                // 16 (c, s) per goal: by pair, or by row
  int *rk = reinterpret_cast<int *>(vec + 96);

  const int groups = (a.B + QUAD_SLOTS - 1) / QUAD_SLOTS;
  for (int g4 = blockIdx.x; g4 < groups; g4 += gridDim.x) {
    asm volatile("" : "+v"(i));     // (row-derived addresses are recomputed per group, not kept across the loop: see quad_jacobi_fixed)
    const bool has = i < N;
    const int b_raw = g4 * QUAD_SLOTS + slot;
    const bool valid = b_raw < a.B;
    const int b = valid ? b_raw : a.B - 1;           // (an empty slot of the last group repeats the last goal, unstored)
    const double *Tg = a.T_goal + (size_t)b * D * D * pc.n_ee;
    double *U = M0, *L = M1, *A = M2;
    __builtin_amdgcn_wave_barrier();
    if (has)
      for (int j = 0; j < N; ++j) {
        const double lo = pc.base_lower[i * N + j], up = pc.base_upper[i * N + j];
        U[i * S + j] = (i == j) ? 0.0 : (up == up ? up : INFINITY);
        L[i * S + j] = (i == j) ? 0.0 : (lo == lo ? lo : -INFINITY);
      }
    __builtin_amdgcn_wave_barrier();
    // goal nodes (_pose_goal) and the distances graph_complete_edges gives them (dgp.py:124-147)
    for (int idx = i; idx < n_gd; idx += QUAD_NODES) {
      int an, gn;
      const double d = goal_distance(pc, Tg, idx, an, gn);
      gd[idx] = d;
      U[an * S + gn] = U[gn * S + an] = d;
      L[an * S + gn] = L[gn * S + an] = d;
    }
    __builtin_amdgcn_wave_barrier();
    // per-term targets: squared goal distances for the goal edges, template constants otherwise
    if (valid)
      for (int t = i; t < pc.T; t += QUAD_NODES) {
        const int src = pc.term_src[t];
        const double g = src >= 0 ? gd[src] : 0.0;
        a.targets[(size_t)b * pc.T + t] = src >= 0 ? g * g : pc.term_static[t];
      }
#ifdef GIK_DEV
    if (a.stop_phase == 1) continue;
#endif
    // ---- bound smoothing: ub = APSP(UPPER) (Floyd-Warshall; row m and column m do not change in step m), then
    //      lb[u][v] = max(0, max_{a,b} LOWER[a][b] - ub[u][a] - ub[b][v])   (see dgp.py)
#pragma unroll 1
    for (int m = 0; m < N; ++m) {                    // (rolled, like the Householder steps: no faster unrolled, 5 k instructions less)
      if (has) {
        const double uim = U[i * S + m];
        for (int j = 0; j < N; ++j) {
          const double cand = uim + U[m * S + j];
          if (cand < U[i * S + j]) U[i * S + j] = cand;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
#ifdef GIK_DEV
    if (a.stop_phase == 2) continue;
#endif
    if (has)
      for (int c = 0; c < N; ++c) {                  // A[u][b] = max_a (L[a][b] - U[u][a])
        double best = -INFINITY;
        for (int q = 0; q < N; ++q) best = fmax(best, L[q * S + c] - U[i * S + q]);
        A[i * S + c] = best;
      }
    __builtin_amdgcn_wave_barrier();
    if (has)
      for (int c = 0; c < N; ++c) {                  // lb[u][v] (over L: the table is no longer needed)
        double best = 0.0;
        for (int q = 0; q < N; ++q) best = fmax(best, A[i * S + q] - U[q * S + c]);
        L[i * S + c] = best;
      }
    __builtin_amdgcn_wave_barrier();
#ifdef GIK_DEV
    if (a.stop_phase == 3) continue;
#endif
    // ---- generate_initialization: D_rand = (lb + 0.9 (ub - lb))^2, Gram = -1/2 J D J
    double *X = M1, *V = M0;
    if (has) {
      for (int j = 0; j < N; ++j) {
        const double lbv = L[i * S + j], d = lbv + 0.9 * (U[i * S + j] - lbv);
        X[i * S + j] = d * d;
      }
      double s = 0.0;                                // (a second pass, as in prep_wave_kernel: the squares are
      for (int j = 0; j < N; ++j) s += X[i * S + j]; //  rounded before they are summed, no fused multiply-add)
      ev[i] = s / N;
    }
    __builtin_amdgcn_wave_barrier();
    double mean = 0.0;
    for (int j = 0; j < N; ++j) mean += ev[j];
    mean /= N;
    if (has)
      for (int j = 0; j < N; ++j) {
        A[i * S + j] = -0.5 * (X[i * S + j] - ev[i] - ev[j] + mean);
        V[i * S + j] = (i == j) ? 1.0 : 0.0;
      }
#ifdef GIK_DEV
    if (a.stop_phase == 4) continue;
#endif
    if constexpr (NT > 0) quad_jacobi_fixed<NT, (NT | 1)>(A, V, a.sweeps, cs, slot, i);
    else quad_jacobi(A, V, S, N, a.sweeps, cs, slot, i);
#ifdef GIK_DEV
    if (a.stop_phase == 5) continue;
#endif
    // ---- factor(): clip, scale by sqrt(lambda), order descending (fliplr of ascending)
    if (has) ev[i] = A[i * S + i];
    __builtin_amdgcn_wave_barrier();
    if (has) {
      rk[i] = desc_rank(ev, N, i);
      // canonical sign: the entry of largest magnitude (first on ties) is positive (see prep_wave_kernel)
      double big = 0.0, sgn = 1.0;
      for (int r = 0; r < N; ++r) {
        const double v = V[r * S + i];
        if (fabs(v) > big) {
          big = fabs(v);
          sgn = v < 0.0 ? -1.0 : 1.0;
        }
      }
      sg[i] = sgn * sqrt(fmax(ev[i], 0.0));
    }
    __builtin_amdgcn_wave_barrier();
    if (has)
      for (int c = 0; c < N; ++c) X[i * S + rk[c]] = V[i * S + c] * sg[c];
    __builtin_amdgcn_wave_barrier();
    // ---- MDS(): K = #eigenvalues > eps of eigh(x), i.e. of the symmetric matrix read from the
    //      LOWER triangle of the non-symmetric factor x (dgp.py:166-167, numpy UPLO='L')
    if (has)
      for (int j = 0; j < N; ++j) A[i * S + j] = (i >= j) ? X[i * S + j] : X[j * S + i];
    __builtin_amdgcn_wave_barrier();
#ifdef GIK_DEV
    if (a.stop_phase == 6) continue;
#endif
    const int Kc = quad_count_eigs_above(A, S, N, 1e-8, hv, hw, i);
    if (a.K_out && valid && i == 0) a.K_out[b] = Kc;
    __builtin_amdgcn_wave_barrier();
#ifdef GIK_DEV
    if (a.stop_phase == 7) continue;
#endif
    // ---- linear_projection (dgp.py:174-183): scatter of the edge differences of the first Kc
    //      columns, its top-`dim` eigenvectors
    if (has)
      for (int c = Kc; c < N; ++c) X[i * S + c] = 0.0;
    __builtin_amdgcn_wave_barrier();
    if (has)
      for (int c = 0; c < N; ++c) {
        double s = 0.0;
        if (i < Kc && c < Kc) {
          for (int p = 0; p < pc.n_pairs; ++p) {
            const int pi = pc.pair_i[p], pj = pc.pair_j[p];
            s = fma(X[pi * S + i] - X[pj * S + i], X[pi * S + c] - X[pj * S + c], s);
          }
        }
        A[i * S + c] = 2.0 * s;  // the reference sums both (i,j) and (j,i)
        V[i * S + c] = (i == c) ? 1.0 : 0.0;
      }
#ifdef GIK_DEV
    if (a.stop_phase == 8) continue;
#endif
    // the largest block among the four goals sets the schedule (wave-uniform)
    const int n2 = Kc > 1 ? Kc : 2;
    const int n2max = max(max(__builtin_amdgcn_readlane(n2, 0), __builtin_amdgcn_readlane(n2, 16)),
                          max(__builtin_amdgcn_readlane(n2, 32), __builtin_amdgcn_readlane(n2, 48)));
    if constexpr (NT >= 8) {   // (the usual block sizes of a chain: K = 6..8)
      if (n2max == 7) quad_jacobi_fixed<7, (NT | 1)>(A, V, a.sweeps, cs, slot, i);
      else if (n2max == 8) quad_jacobi_fixed<8, (NT | 1)>(A, V, a.sweeps, cs, slot, i);
      else if (n2max == 6) quad_jacobi_fixed<6, (NT | 1)>(A, V, a.sweeps, cs, slot, i);
      else quad_jacobi(A, V, S, n2max, a.sweeps, cs, slot, i);
    } else {
      quad_jacobi(A, V, S, n2max, a.sweeps, cs, slot, i);
    }
#ifdef GIK_DEV
    if (a.stop_phase == 9) continue;
#endif
    if (has) ev[i] = (i < Kc) ? A[i * S + i] : -INFINITY;
    __builtin_amdgcn_wave_barrier();
    if (has) {
      rk[i] = desc_rank(ev, N, i);
      double big = 0.0, sgn = 1.0;
      for (int r = 0; r < N; ++r) {
        const double v = V[r * S + i];
        if (fabs(v) > big) {
          big = fabs(v);
          sgn = v < 0.0 ? -1.0 : 1.0;
        }
      }
      sg[i] = sgn;
    }
    __builtin_amdgcn_wave_barrier();
    // Y = X * W, W = the K eigenvectors of largest eigenvalue
    if (has && valid)
      for (int dcol = 0; dcol < K; ++dcol) {
        int col = 0;
        for (int c = 0; c < N; ++c) col = (rk[c] == dcol) ? c : col;
        double s = 0.0;
        for (int c = 0; c < N; ++c) s += X[i * S + c] * V[c * S + col];
        a.Y_init[(size_t)b * N * K + i * K + dcol] = s * sg[col];
      }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace gik
