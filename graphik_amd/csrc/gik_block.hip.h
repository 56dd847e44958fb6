// graphik_amd/csrc/gik_block.hip.h -- workgroup-per-IK-problem context for graphs that do not fit
// one wavefront (N*k > 64): e.g. UR10 + table_environment() with N = 116 nodes and 5612 residual
// terms (BASELINE configs[2]; every obstacle is a free point tied to all anchors).
//
// 512 threads, four per node ("parts" = one DPP quad).  Thread (node i, part p < k) owns the
// unknown (i, p); all four parts of a node share the node's residual terms (a quarter each).  A
// node with ~100 neighbours cannot cache per-slot Hessian blocks in registers, so the
// Hessian-vector product recomputes y = Y_i - Y_j, d, c from the point rows kept in LDS; each term
// is evaluated from both endpoints (owner-computes: no scatter, no atomics, fixed summation
// order), the four partial 3-vectors of a node are combined with two quad_perm DPP stages.
// Reductions: per-wave MFMA reduction, then eight wave totals through a double-buffered LDS
// scratch (one barrier per reduction).  The solver logic is the shared rtr_solve_one<>.
//
// Rigid cliques (k = 3).  The table scene's 5612 terms are 5565 equalities among 106 anchors (base,
// goal nodes, obstacle centres: every pair is tied) and 47 others.  gik_template_create finds such
// a clique A (n = |A| >= 16), gives its nodes the LDS rows 0..n-1 and takes its terms out of the
// slot tables.  Their share of the Hessian-vector product is evaluated in closed form: with the
// rows centred on the clique's centroid (y~_j, sum_j y~_j = 0) and the moments
//   Sw = sum w_j, M = sum y~_j w_j^T, T3 = sum (y~_j.w_j) y~_j, U3 = sum |y~_j|^2 w_j   (18 sums),
//   sum_j [2 (y_ij.w_ij) y_ij + c_ij w_ij]
//     = 2 [ y~_i (n a_i - y~_i.Sw + tr M) + M y~_i + Syy w_i - T3 ]
//       + w_i (n |y~_i|^2 + tr Syy - sum_j D_ij) - |y~_i|^2 Sw + 2 M^T y~_i - U3 + (D w)_i,
// a_i = y~_i.w_i, Syy = sum y~_j y~_j^T (refreshed per accepted step).  Only (D w)_i is O(n) per
// node, three FMAs per pair with D_ij held in registers (27 per thread at n = 106), instead of a
// 20-instruction term evaluation from both ends.  The round-off is of the same size as the direct
// sum's (both carry eps * |y|^2 |w| per pair: c_ij = d_ij - D_ij is a difference of O(1) numbers
// either way; measured 2e-13 against 2e-13 relative to a long-double sum at n = 104).  Cost and
// gradient (once per outer iteration) walk the clique pairs directly.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gik_wave.hip.h"

namespace gik {

constexpr int BLOCK_NT = 512;
constexpr int BLOCK_WAVES = BLOCK_NT / WAVE;
constexpr int BLOCK_MAXN = BLOCK_NT / 4;  // 128 nodes
constexpr int CLQ_M = BLOCK_MAXN / 4;     // clique partners per thread (row 4m + part, m < CLQ_M)
constexpr int CLQ_NCQ = 12;               // per accepted point: Syy (3 x 3, row-major), tr Syy, n, pad
constexpr int CLQ_NMOM = 27;              // Sw[3], M[3][3], T3[3], U3[3] (Euclidean targets: U3 - R3), P[3][3]

// launch-invariant tables of the workgroup-per-problem path (device pointers; gik_template_create)
struct BlockTabs {
  const int *nc_term;      // [Tc] index in the caller's target row of each slot-table term; null = identity
  const int *clq_term;     // [M][512] target index of clique pair (row tid >> 2, row 4m + (tid & 3)); -1 = none
  const int *clq_pair_term;        // [n_pairs] target index of clique pair p (each pair once)
  const unsigned short *clq_pid_t; // [M][512] pair id of (row tid >> 2, row 4m + (tid & 3)), like clq_term; 0xffff = none
  int n_pairs;
  const int *node_of_row;  // [128] node (the caller's numbering) of each LDS row; null = identity
  const int *wave_sl;      // [8][2] slot-loop bounds {SLE, SL} of each wavefront
  int Tc;                  // terms kept in the slot tables
  int n_clq;               // clique size (rows 0..n_clq-1), 0 = none
  int clq_euclid;          // 1: look for point coordinates behind the clique's target distances
};

template <int K>
struct BlockCtx {
  static constexpr int NC = (K == 3) ? 3 : 1;
  static constexpr int RS = 4;  // LDS row stride (doubles): 32-byte rows
  static constexpr bool HAS_CK = (K == 3);   // tCG checkpoint (rtr_solve_one "Retrace"): [4][512] doubles
  static constexpr bool AGE_PRIORITY = false;  // one workgroup per CU: nothing shares its SIMDs
  double *sh_ck;
  __device__ inline void ck_put(int i, double v) { sh_ck[i * BLOCK_NT + tid] = v; }   // own thread only
  __device__ inline double ck_get(int i) const { return sh_ck[i * BLOCK_NT + tid]; }

  int tid, lane, wave, node, part, N, SL, SLE;  // slots [0, SLE) hold equality terms or padding only
  int gnode;         // the caller's index of LDS row `node`
  bool live;         // LDS row `node` holds a node
  bool active;       // owns an unknown: live && part < K
  // rigid clique (rows 0..n_clq-1), k = 3 only
  int n_clq, M_clq;
  bool wclq;                   // this wavefront owns clique rows (wave-uniform)
  uint32_t clq_valid;          // bit m: (node, 4m + part) is a clique pair
  // its squared target distance is NOT kept in registers (28 + 4 doubles per thread used to sit
  // there for the whole solve and were spilled around proj_setup: 4.4 GB of scratch traffic per
  // 4096-goal launch).  The clique's targets -- each pair once, 44.5 KB on the table scene -- are
  // staged in LDS per problem (sh_ctg), the pair ids of a thread's partners once per kernel
  // (sh_pid, 16 bit): the once-per-outer-iteration walks of cost() / commit() gather from there.
  double *sh_ctg;              // [n_pairs] squared target distances of the clique pairs
  const unsigned short *sh_pid;// [CLQ_M rows used][512] pair id of (node, 4m + part), 0xffff = none
  __device__ inline double dr(int m) const {
    if (m >= M_clq) return 0.0;       // (the dense D w product runs whole groups of four partners)
    const unsigned p = sh_pid[m * BLOCK_NT + tid];
    return p != 0xffffu ? sh_ctg[p] : 0.0;
  }
  double rD, n_count;          // sum_j D_ij of the node; (double)n_clq
  double yt[3], ytp, y2t;      // centred row of the node, own entry, squared norm
  // Euclidean targets: if the clique's D_ij are the squared distances of points X_j in R^3 (any
  // rigid scene), D = r 1^T + 1 r^T - 2 X X^T has rank 5 and (D w)_i = r_i Sw + R3 - 2 P X_i with
  // R3 = sum r_j w_j, P = sum w_j X_j^T: 9 more moments (R3 rides with U3) replace the O(n)
  // product per node
  bool lowrank;                // this problem's clique targets passed the check (block-uniform)
  double Xr[3], rr;            // centred coordinates of the node, squared norm
  double *sh_mom;              // [CLQ_NMOM][BLOCK_WAVES] per-wave partial moments, then
                               // [8] Syy (xx xy xz yy yz zz), tr Syy, n of the committed point
  double *sh_Y;      // [128][4] accepted point
  double *sh_P;      // [128][4] proposal (cost) / point being committed
  double *sh_W;      // [128][4] direction being differentiated
  const double *sh_tgt;
  const uint32_t *sh_slots;  // [SL][512]
  double *sh_red;            // [2][8][BLOCK_WAVES]
  int red_buf;
  double pk[NC], pk2[NC], Pm[NC * NC], G2[NC * NC], Q[NC];
#ifdef GIK_BLK_PROF
  double *prof = nullptr;
#endif

  __host__ __device__ static constexpr size_t lds_bytes(int T, int SL, int n_pairs = 0, int n_clq = 0) {
    return sizeof(double) * ((size_t)3 * BLOCK_MAXN * RS + (size_t)((T + 1) & ~1) + 2 * 8 * BLOCK_WAVES +
                             CLQ_NMOM * BLOCK_WAVES + CLQ_NCQ + 32 * BLOCK_WAVES) +
           sizeof(uint32_t) * (size_t)SL * BLOCK_NT + (HAS_CK ? sizeof(double) * 4 * BLOCK_NT : 0) +
           sizeof(double) * (size_t)((n_pairs + 1) & ~1) + sizeof(unsigned short) * (size_t)((n_clq + 3) / 4) * BLOCK_NT;
  }

  __device__ inline bool lead() const { return tid == 0; }

  // T: terms in the slot tables (bt.Tc); SL_: slot rows staged in LDS (the busiest wavefront's)
  __device__ inline void init(int N_, int SL_, const BlockTabs &bt, double *base, const uint32_t *slots_lds,
                              int T) {
    tid = threadIdx.x;
    lane = tid & 63;
    wave = tid >> 6;
    node = tid >> 2;
    part = tid & 3;
    N = N_;
    SLE = __builtin_amdgcn_readfirstlane(bt.wave_sl[2 * wave]);
    SL = __builtin_amdgcn_readfirstlane(bt.wave_sl[2 * wave + 1]);
    gnode = bt.node_of_row ? bt.node_of_row[node] : (node < N ? node : -1);
    live = gnode >= 0;
    active = live && part < K;
    sh_Y = base;
    sh_P = sh_Y + BLOCK_MAXN * RS;
    sh_W = sh_P + BLOCK_MAXN * RS;
    double *tg = sh_W + BLOCK_MAXN * RS;
    sh_tgt = tg;
    sh_red = tg + ((T + 1) & ~1);
    sh_mom = sh_red + 2 * 8 * BLOCK_WAVES;
    sh_slots = slots_lds;
    sh_ck = reinterpret_cast<double *>(const_cast<uint32_t *>(slots_lds) + (size_t)SL_ * BLOCK_NT);
    red_buf = 0;
    n_clq = (K == 3) ? bt.n_clq : 0;
    M_clq = __builtin_amdgcn_readfirstlane((n_clq + 3) >> 2);
    n_count = (double)n_clq;
    wclq = __builtin_amdgcn_readfirstlane(wave * (WAVE / 4)) < n_clq;
    clq_valid = 0u;
    rD = 0.0;
    lowrank = false;
    rr = 0.0;
    Xr[0] = Xr[1] = Xr[2] = 0.0;
    sh_ctg = sh_ck + (HAS_CK ? 4 * BLOCK_NT : 0);
    sh_pid = reinterpret_cast<const unsigned short *>(sh_ctg + ((bt.n_pairs + 1) & ~1));
  }

  // per problem: the slot-table targets into LDS (sh_tgt, writable alias `tgw`), the clique's
  // target distances into registers
  __device__ inline void load_problem(const double *tg_b, const BlockTabs &bt, double *tgw) {
    for (int t = tid; t < bt.Tc; t += BLOCK_NT)
      tgw[t] = tg_b ? tg_b[bt.nc_term ? bt.nc_term[t] : t] : 0.0;
    if constexpr (K == 3) {
      if (n_clq) {
        double s = 0.0;
        uint32_t v = 0u;
        for (int p = tid; p < bt.n_pairs; p += BLOCK_NT) sh_ctg[p] = tg_b ? tg_b[bt.clq_pair_term[p]] : 0.0;
        __syncthreads();
#pragma unroll
        for (int m = 0; m < CLQ_M; ++m) {
          if (m >= M_clq) continue;
          const unsigned p = sh_pid[m * BLOCK_NT + tid];
          v |= (p != 0xffffu ? 1u : 0u) << m;
          s += dr(m);
        }
        clq_valid = v;
        s += dpp_f64<0xB1>(s);
        s += dpp_f64<0x4E>(s);
        rD = s;
        lowrank = false;
        if (bt.clq_euclid && tg_b) {
          __syncthreads();
          lowrank = clique_coordinates(tg_b, bt);
        }
      }
    }
    __syncthreads();
  }

  // block-wide argmax over the clique rows of a per-row value (held by every lane of the row);
  // scratch: column `col` of sh_W.  Returns the row, the value in `best`.
  __device__ inline int clique_argmax(double val, int col, double &best) {
    if (part == 0) sh_W[node * RS + col] = (node < n_clq) ? val : -1.0;
    __syncthreads();
    double b = -1.0;
    int arg = 0;
    for (int j = 0; j < n_clq; ++j) {
      const double v = sh_W[j * RS + col];
      if (v > b) {
        b = v;
        arg = j;
      }
    }
    __syncthreads();
    best = b;
    return arg;
  }

  // Are the clique's target distances those of a point set in R^3?  Coordinates by trilateration
  // from four of its nodes -- row 0, the row farthest from it, the row farthest from their line, the
  // row farthest from their plane -- then every pair is checked against its target at 1e-12 of
  // the largest distance.  On success the centred coordinates of this thread's node are in Xr / rr
  // and those of all rows in sh_P (x, y, z, |X|^2) until the solve overwrites them.
  __device__ inline bool clique_coordinates(const double *tg_b, const BlockTabs &bt) {
    const bool in = node < n_clq;
    auto dist_to = [&](int a) -> double {   // D(node, a), a block-uniform
      if (!in) return 0.0;
      const int idx = bt.clq_term[(a >> 2) * BLOCK_NT + (node << 2) + (a & 3)];
      return idx >= 0 ? tg_b[idx] : 0.0;
    };
    double m1, m2, m3;
    const double d0 = dist_to(0);
    const int a1 = clique_argmax(d0, 0, m1);
    if (!(m1 > 0.0)) return false;
    const double d01 = sqrt(m1);
    const double x = (d0 + m1 - dist_to(a1)) / (2.0 * d01);
    const int a2 = clique_argmax(d0 - x * x, 0, m2);
    if (!(m2 > 1e-6 * m1)) return false;    // collinear
    if (part == 0) sh_P[node * RS] = x;
    __syncthreads();
    const double x2 = sh_P[a2 * RS], y2 = sqrt(m2);
    const double y = (d0 - dist_to(a2) + (x2 * x2 + m2) - 2.0 * x * x2) / (2.0 * y2);
    const int a3 = clique_argmax(d0 - x * x - y * y, 0, m3);
    if (!(m3 > 1e-6 * m1)) return false;    // coplanar
    if (part == 0) sh_P[node * RS + 1] = y;
    __syncthreads();
    const double x3 = sh_P[a3 * RS], y3 = sh_P[a3 * RS + 1], z3 = sqrt(m3);
    const double z = (d0 - dist_to(a3) + (x3 * x3 + y3 * y3 + m3) - 2.0 * x * x3 - 2.0 * y * y3) / (2.0 * z3);
    const double lm = (in && part == 0) ? 1.0 : 0.0;
    double c[3] = {lm * x, lm * y, lm * z};
    sum_n<3>(c);
    const double inv_n = 1.0 / n_count;
    Xr[0] = fma(-c[0], inv_n, x);
    Xr[1] = fma(-c[1], inv_n, y);
    Xr[2] = fma(-c[2], inv_n, z);
    rr = fma(Xr[2], Xr[2], fma(Xr[1], Xr[1], Xr[0] * Xr[0]));
    if (in) sh_P[node * RS + part] = part == 0 ? Xr[0] : (part == 1 ? Xr[1] : (part == 2 ? Xr[2] : rr));
    __syncthreads();
    double bad = 0.0;
    const double tol = 1e-12 * m1;
    const double *pb = sh_P + part * RS;
#pragma unroll
    for (int m = 0; m < CLQ_M; ++m) {
      if (m >= M_clq) continue;
      double r[K];
      row(pb, 4 * m, r);
      const double e0 = Xr[0] - r[0], e1 = Xr[1] - r[1], e2 = Xr[2] - r[2];
      const double e = fma(e2, e2, fma(e1, e1, e0 * e0)) - dr(m);
      bad += (((clq_valid >> m) & 1u) && !(fabs(e) <= tol)) ? 1.0 : 0.0;
      if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    const double nbad = sum1(bad);
    __syncthreads();   // sh_P is the solver's from here on
    return nbad == 0.0;
  }

  template <int NV>
  __device__ inline void sum_n(double (&v)[NV]) {
    static_assert(NV <= 8, "reduction scratch holds 8 values");
    double *buf = sh_red + red_buf * 8 * BLOCK_WAVES;
    if constexpr (NV == 8) {
      // the lanes that hold the eight wave totals store them: one exec-masked write instead of
      // sixteen v_readlane and eight writes from lane 0
      const double w = wave_sum8_distributed(v);
      const int q = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2);
      if ((lane & 7) == 0) buf[q * BLOCK_WAVES + wave] = w;
    } else {
      wave_sum_n<NV>(v);  // wave totals, uniform within the wave
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) buf[q * BLOCK_WAVES + wave] = v[q];
      }
    }
    __syncthreads();
    // cross-wave step: lane 8q + w fetches the total of wave w for value q (one LDS read per
    // thread), a 3-stage butterfly inside each group of 8 lanes adds the eight waves in the
    // order ((0+1)+(2+3))+((4+5)+(6+7)), and value q is read back from lane 8q
    static_assert(BLOCK_WAVES == 8, "butterfly below assumes 8 waves");
    double t = (lane < NV * BLOCK_WAVES) ? buf[lane] : 0.0;
    t += dpp_f64<0xB1>(t);   // quad_perm [1,0,3,2]
    t += dpp_f64<0x4E>(t);   // quad_perm [2,3,0,1]
    t += dpp_f64<0x141>(t);  // row_half_mirror
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = readlane_f64(t, 8 * q);
    red_buf ^= 1;  // the next reduction writes the other buffer: one barrier per reduction
  }
  __device__ inline double sum1(double x) {
    double v[1] = {x};
    sum_n<1>(v);
    return v[0];
  }

  __device__ inline void row(const double *M, int r, double (&o)[K]) const {
    const double2 a = *reinterpret_cast<const double2 *>(M + r * RS);
    o[0] = a.x;
    o[1] = a.y;
    if constexpr (K == 3) o[2] = M[r * RS + 2];
  }

  // sum the four parts of a node (one DPP quad) and return entry `part` of the result
  __device__ inline double quad_pick(double (&acc)[K]) const {
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] += dpp_f64<0xB1>(acc[q]);
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] += dpp_f64<0x4E>(acc[q]);
    if constexpr (K == 3)
      return part == 0 ? acc[0] : (part == 1 ? acc[1] : acc[2]);
    else
      return part == 0 ? acc[0] : acc[1];
  }

  // f(Yv) (lcost / jcost, costs.py:80-93, 8-16); leaves Yv in sh_P
  __device__ inline double cost(double Yv) {
    if (active) sh_P[node * RS + part] = Yv;
    __syncthreads();
    double own[K];
    row(sh_P, live ? node : 0, own);
    double f = 0.0;
    // equality-only slots first (no kind decoding: 5604 of the table scene's 5612 terms), then
    // the few slots that may hold hinge terms
#pragma unroll 2
    for (int s = 0; s < SLE; ++s) {
      const uint32_t m = sh_slots[s * BLOCK_NT + tid];
      double r[K];
      row(sh_P, meta_j(m), r);
      double d = 0.0;
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const double y = own[q] - r[q];
        d = fma(y, y, d);
      }
      const double u = sh_tgt[meta_term(m)] - d;
      f = fma(meta_owner(m) ? u : 0.0, u, f);   // padding slots are never owners
    }
    for (int s = SLE; s < SL; ++s) {
      const uint32_t m = sh_slots[s * BLOCK_NT + tid];
      double r[K];
      row(sh_P, meta_j(m), r);
      double d = 0.0;
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const double y = own[q] - r[q];
        d = fma(y, y, d);
      }
      const int kind = meta_kind(m);
      const double u = sh_tgt[meta_term(m)] - d;
      const double wp = (meta_owner(m) && (kind == GIK_TERM_EQ || kind == GIK_TERM_LOWER)) ? 1.0 : 0.0;
      const double wn = (meta_owner(m) && (kind == GIK_TERM_EQ || kind == GIK_TERM_UPPER)) ? 1.0 : 0.0;
      const double pp = fmax(u, 0.0), nn = fmax(-u, 0.0);
      f = fma(wp * pp, pp, f);
      f = fma(wn * nn, nn, f);
    }
    if constexpr (K == 3) {
      if (n_clq && wclq) {   // clique pairs, each counted by its lower row
        const double *pb = sh_P + part * RS;
#pragma unroll
        for (int m = 0; m < CLQ_M; ++m) {
          if (m >= M_clq) continue;
          double r[K];
          row(pb, 4 * m, r);
          double d = 0.0;
#pragma unroll
          for (int q = 0; q < K; ++q) {
            const double y = own[q] - r[q];
            d = fma(y, y, d);
          }
          const double u = dr(m) - d;
          const bool mine = ((clq_valid >> m) & 1u) && (4 * m + part > node);
          f = fma(mine ? u : 0.0, u, f);
          if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound the loads in flight (registers)
        }
      }
    }
    return sum1(f);
  }

  // accept the point in sh_P: copy it to sh_Y and return this thread's egrad entry
  __device__ inline double commit() {
    double own[K];
    row(sh_P, live ? node : 0, own);
    double acc[K];
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = 0.0;
#pragma unroll 2
    for (int s = 0; s < SLE; ++s) {   // equality terms (padding: y = 0)
      const uint32_t m = sh_slots[s * BLOCK_NT + tid];
      double r[K], y[K];
      row(sh_P, meta_j(m), r);
      double d = 0.0;
#pragma unroll
      for (int q = 0; q < K; ++q) {
        y[q] = own[q] - r[q];
        d = fma(y[q], y[q], d);
      }
      const double c = d - sh_tgt[meta_term(m)];
#pragma unroll
      for (int q = 0; q < K; ++q) acc[q] = fma(c, y[q], acc[q]);
    }
    for (int s = SLE; s < SL; ++s) {
      const uint32_t m = sh_slots[s * BLOCK_NT + tid];
      double r[K], y[K];
      row(sh_P, meta_j(m), r);
      double d = 0.0;
#pragma unroll
      for (int q = 0; q < K; ++q) {
        y[q] = own[q] - r[q];
        d = fma(y[q], y[q], d);
      }
      const int kind = meta_kind(m);
      const double c0 = d - sh_tgt[meta_term(m)];
      const bool act = (kind == GIK_TERM_EQ) || (kind == GIK_TERM_LOWER && c0 < 0.0) ||
                       (kind == GIK_TERM_UPPER && c0 > 0.0);
      const double c = act ? c0 : 0.0;
#pragma unroll
      for (int q = 0; q < K; ++q) acc[q] = fma(c, y[q], acc[q]);
    }
    if constexpr (K == 3) {
      if (n_clq && wclq) {
        const double *pb = sh_P + part * RS;
#pragma unroll
        for (int m = 0; m < CLQ_M; ++m) {
          if (m >= M_clq) continue;
          double r[K], y[K];
          row(pb, 4 * m, r);
          double d = 0.0;
#pragma unroll
          for (int q = 0; q < K; ++q) {
            y[q] = own[q] - r[q];
            d = fma(y[q], y[q], d);
          }
          const double c = ((clq_valid >> m) & 1u) ? d - dr(m) : 0.0;
#pragma unroll
          for (int q = 0; q < K; ++q) acc[q] = fma(c, y[q], acc[q]);
          if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    const double G = 2.0 * quad_pick(acc);
    if (active) sh_Y[node * RS + part] = sh_P[node * RS + part];
    __syncthreads();
    if constexpr (K == 3) {
      if (n_clq) clique_refresh(own);
    }
    return active ? G : 0.0;
  }

  // moments of the committed point over the clique: centroid, centred row of this node, Syy
  __device__ inline void clique_refresh(const double (&own)[K]) {
    const double lm = (node < n_clq && part == 0) ? 1.0 : 0.0;
    double c[3] = {lm * own[0], lm * own[1], lm * own[2]};
    sum_n<3>(c);
    const double inv_n = 1.0 / n_count;
#pragma unroll
    for (int q = 0; q < 3; ++q) yt[q] = fma(-c[q], inv_n, own[q]);
    ytp = part == 0 ? yt[0] : (part == 1 ? yt[1] : yt[2]);
    y2t = fma(yt[2], yt[2], fma(yt[1], yt[1], yt[0] * yt[0]));
    double x[6] = {lm * yt[0] * yt[0], lm * yt[0] * yt[1], lm * yt[0] * yt[2],
                   lm * yt[1] * yt[1], lm * yt[1] * yt[2], lm * yt[2] * yt[2]};
    sum_n<6>(x);
    if (tid == 0) {   // uniform; read back (broadcast) by clique_closed_form after the next barrier
      double *cq = sh_mom + CLQ_NMOM * BLOCK_WAVES;
      cq[0] = x[0]; cq[1] = x[1]; cq[2] = x[2];
      cq[3] = x[1]; cq[4] = x[3]; cq[5] = x[4];
      cq[6] = x[2]; cq[7] = x[4]; cq[8] = x[5];
      cq[9] = (x[0] + x[3]) + x[5];
      cq[10] = n_count;
      cq[11] = 0.0;
    }
  }

  // Per-wave partial sums of the 18 clique moments of the direction W, from registers: thread
  // (j, p < 3) holds w_j[p] and adds to {Sw[p], M[0][p], M[1][p], M[2][p], T3[p], U3[p]}.  One
  // v_mfma_f64_4x4x4 with B = 1 adds the four 16-lane rows of every lane column and leaves, in
  // lane row i, the totals of part i (layout: wave_sum_n, gik_wave.hip.h); two row rotations add
  // the four quads of a row.  Lane 16 i then holds the wave's total for part i.
  __device__ inline void clique_moment_partials(double W) {
    if (!wclq) {
      if ((lane & 15) == 0 && lane < 48) {
#pragma unroll
        for (int r = 0; r < 9; ++r) sh_mom[(r * 3 + (lane >> 4)) * BLOCK_WAVES + wave] = 0.0;
      }
      return;
    }
    const double cm = (node < n_clq && part < 3) ? 1.0 : 0.0;
    const double w = cm * W;
    double aw = ytp * w;
    aw += dpp_f64<0xB1>(aw);
    aw += dpp_f64<0x4E>(aw);   // a_j = y~_j . w_j in the four lanes of the node
    // Registers: 0 Sw, 1..3 M[r-1][.], 4 U3 (with Euclidean targets U3 and R3 = sum r_j w_j only
    // enter as R3 - U3: one register), 5..7 P[.][r-5]; lane part p holds the entry of column p.
    // Eight of them are folded by transposing exchanges, as in wave_sum_n<8>: v_permlane32_swap
    // puts the upper half of one register next to the lower half of another, so one add folds
    // lane rows (0, 2) and (1, 3) of TWO registers; v_permlane16_swap does the same for the
    // remaining pair of rows, after which lane row rho of a register holds the complete
    // cross-row sums of original register {0, 2, 1, 3}[rho].  Lane positions (part, quad) are
    // preserved throughout; two row rotations add the four quads.  30 instructions for eight
    // registers instead of eight MFMAs + 48 (an f64 MFMA occupies the matrix pipe ~16 cycles and
    // the two waves of a SIMD share it).
    // (T3[a] = sum a_j y~_j[a] rides in the idle fourth lane of register 1 + a)
    const double wsel = part == 3 ? (node < n_clq ? aw : 0.0) : w;
    double v[8];
    v[0] = w; v[1] = yt[0] * wsel; v[2] = yt[1] * wsel; v[3] = yt[2] * wsel;
    v[4] = (lowrank ? y2t - rr : y2t) * w;
    v[5] = lowrank ? Xr[0] * w : 0.0;
    v[6] = lowrank ? Xr[1] * w : 0.0;
    v[7] = lowrank ? Xr[2] * w : 0.0;
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
      lane_swap32(v[q], v[q + 1]);
      v[q] += v[q + 1];
    }
    lane_swap16(v[0], v[2]);
    v[0] += v[2];
    lane_swap16(v[4], v[6]);
    v[4] += v[6];
#pragma unroll
    for (int q = 0; q < 8; q += 4) {
      v[q] += dpp_f64<0x128>(v[q]);   // row_ror:8
      v[q] += dpp_f64<0x124>(v[q]);   // row_ror:4
    }
    if ((lane & 12) == 0) {   // quad 0 of every lane row: row rho holds register {0, 2, 1, 3}[rho] (+ 4)
      const int rho = lane >> 4;
      const int reg = ((rho & 1) << 1) | (rho >> 1);
      // moment index: Sw[p] = p, M[a][p] = 3 + 3a + p, T3[a] = 12 + a, U3[p] = 15 + p, P[p][a] = 18 + 3a + p
      if (part < 3) {
        sh_mom[(reg * 3 + part) * BLOCK_WAVES + wave] = v[0];
        sh_mom[((5 + reg) * 3 + part) * BLOCK_WAVES + wave] = v[4];
      } else if (reg > 0) {
        sh_mom[(12 + reg - 1) * BLOCK_WAVES + wave] = v[0];
      }
    }
  }

  __device__ inline void clique_dw(double (&acc)[K]) {
    // (D w)_i, dense: partner rows 4m + part, a quarter of the clique per thread, in groups of four
    // (D = 0 beyond the clique and rows up to 127 exist, so whole groups are run: one scalar
    // branch per group).  Only for targets that are not distances of points (clique_coordinates),
    // so it is kept small: its registers add to the budget of the loop every problem runs.
    const double *wb = sh_W + part * RS;
#pragma unroll
    for (int g = 0; g < CLQ_M / 4; ++g) {
      if (4 * g >= M_clq) continue;
      double buf[4][K], dg[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) dg[u] = dr(4 * g + u);
#pragma unroll
      for (int u = 0; u < 4; ++u) row(wb, 4 * (4 * g + u), buf[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = fma(dg[u], buf[u][q], acc[q]);
      }
      __builtin_amdgcn_sched_barrier(0);   // do not hoist later groups' loads (registers)
    }
  }

  // the O(1)-per-node part of the clique's Hessian-vector product (see the file header), entry
  // `part` of node `node`; call after the barrier that publishes sh_W and sh_mom.
  // Lanes 0..26 add the eight partials of one moment each; the totals go through a 32-double LDS
  // strip of the wave's own and come back as LDS reads instead of 60 v_readlane into scalar
  // registers.
  __device__ inline double clique_closed_form(const double (&wi)[K]) {
    static_assert(BLOCK_WAVES == 8, "eight partials per moment");
    const double *p = sh_mom + (lane < CLQ_NMOM ? lane : 0) * BLOCK_WAVES;
    const double2 p01 = *reinterpret_cast<const double2 *>(p), p23 = *reinterpret_cast<const double2 *>(p + 2);
    const double2 p45 = *reinterpret_cast<const double2 *>(p + 4), p67 = *reinterpret_cast<const double2 *>(p + 6);
    const double tot = ((p01.x + p01.y) + (p23.x + p23.y)) + ((p45.x + p45.y) + (p67.x + p67.y));
    double *strip = sh_mom + CLQ_NMOM * BLOCK_WAVES + CLQ_NCQ + 32 * wave;
    if (lane < 32) strip[lane] = tot;
    __builtin_amdgcn_wave_barrier();
    // Each lane evaluates ITS component q = part: the moments it needs are picked by address
    // (Sw_q, row q and column q of M, T3_q, U3_q, R3_q, column q of P^T), the ones every lane needs
    // (Sw for y~ . Sw, the diagonal of M) are broadcast reads.
    const int q = part < 3 ? part : 0;
    const double *cq = sh_mom + CLQ_NMOM * BLOCK_WAVES;
    const double *sq = strip + q;
    const double Sq0 = cq[3 * q], Sq1 = cq[3 * q + 1], Sq2 = cq[3 * q + 2];
    const double s_yy = cq[9], nn = cq[10];
    const double wq = part == 0 ? wi[0] : (part == 1 ? wi[1] : wi[2]);
    const double ty[3] = {yt[0] + yt[0], yt[1] + yt[1], yt[2] + yt[2]};
    const double a_i = fma(yt[2], wi[2], fma(yt[1], wi[1], yt[0] * wi[0]));
    const double ySw = fma(yt[2], strip[2], fma(yt[1], strip[1], yt[0] * strip[0]));
    const double s_yw = (strip[3] + strip[7]) + strip[11];
    const double g = fma(nn, a_i, s_yw) - ySw;
    const double cw = fma(nn, y2t, s_yy) - rD;
    const double Syw = fma(Sq2, wi[2], fma(Sq1, wi[1], Sq0 * wi[0]));
    double h = fma(wq, cw, Syw + Syw);                                        // w_q cw + 2 (Syy w)_q
    h = fma(lowrank ? rr - y2t : -y2t, sq[0], h);                             // (r_i - |y~|^2) Sw_q
    const double *mr = strip + 3 + 3 * q;                                     // row q of M: 2 (M y~)_q
    h = fma(ty[2], mr[2], fma(ty[1], mr[1], fma(ty[0], mr[0], h)));
    h = fma(ty[2], sq[9], fma(ty[1], sq[6], fma(ty[0], sq[3], h)));           // column q: 2 (M^T y~)_q
    h = fma(-2.0, sq[12], h) - sq[15];                                        // - 2 T3_q - U3_q (+ R3_q)
    h = fma(ytp + ytp, g, h);                                                 // 2 y~_q g
    if (lowrank)     // - 2 (P X_i)_q,  P[q][a] at 18 + 3 a + q
      h = fma(-(Xr[2] + Xr[2]), sq[24], fma(-(Xr[1] + Xr[1]), sq[21], fma(-(Xr[0] + Xr[0]), sq[18], h)));
    return (node < n_clq && part < 3) ? h : 0.0;
  }

  // ehess(Y, W) (lhess / jhess, costs.py:175-207, 39-58) at the committed point sh_Y
  __device__ inline double ehess(double W) {
#ifdef GIK_BLK_PROF
    const long long pt0 = __builtin_readcyclecounter();
#endif
    if (active) sh_W[node * RS + part] = W;
    if constexpr (K == 3) {
      if (n_clq) clique_moment_partials(W);
    }
    __syncthreads();
#ifdef GIK_BLK_PROF
    const long long pt1 = __builtin_readcyclecounter();
#endif
    const int me = live ? node : 0;
    double yi[K], wi[K];
    row(sh_Y, me, yi);
    row(sh_W, me, wi);
    double acc[K];
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = 0.0;
    double hclq = 0.0;
#ifdef GIK_BLK_PROF
    long long pt1a = pt1, pt1b = pt1;
#endif
    if constexpr (K == 3) {
      if (n_clq && wclq) {
        hclq = clique_closed_form(wi);
#ifdef GIK_BLK_PROF
        pt1a = __builtin_readcyclecounter();
#endif
        if (!lowrank) clique_dw(acc);
#ifdef GIK_BLK_PROF
        pt1b = __builtin_readcyclecounter();
#endif
      }
    }
#pragma unroll 2
    for (int s = 0; s < SLE; ++s) {   // equality terms: always active (padding: y = w = 0)
      const uint32_t m = sh_slots[s * BLOCK_NT + tid];
      const int j = meta_j(m);
      double yj[K], wj[K], y[K], w[K];
      row(sh_Y, j, yj);
      row(sh_W, j, wj);
      double d = 0.0, sd = 0.0;
#pragma unroll
      for (int q = 0; q < K; ++q) {
        y[q] = yi[q] - yj[q];
        w[q] = wi[q] - wj[q];
        d = fma(y[q], y[q], d);
        sd = fma(y[q], w[q], sd);
      }
      const double c = d - sh_tgt[meta_term(m)];
      const double a2s = sd + sd;
#pragma unroll
      for (int q = 0; q < K; ++q) acc[q] = fma(a2s, y[q], fma(c, w[q], acc[q]));
    }
    for (int s = SLE; s < SL; ++s) {
      const uint32_t m = sh_slots[s * BLOCK_NT + tid];
      const int j = meta_j(m);
      double yj[K], wj[K], y[K], w[K];
      row(sh_Y, j, yj);
      row(sh_W, j, wj);
      double d = 0.0, sd = 0.0;
#pragma unroll
      for (int q = 0; q < K; ++q) {
        y[q] = yi[q] - yj[q];
        w[q] = wi[q] - wj[q];
        d = fma(y[q], y[q], d);
        sd = fma(y[q], w[q], sd);
      }
      const int kind = meta_kind(m);
      const double c0 = d - sh_tgt[meta_term(m)];
      const bool act = (kind == GIK_TERM_EQ) || (kind == GIK_TERM_LOWER && c0 < 0.0) ||
                       (kind == GIK_TERM_UPPER && c0 > 0.0);
      const double c = act ? c0 : 0.0;
      const double a2s = act ? 2.0 * sd : 0.0;
#pragma unroll
      for (int q = 0; q < K; ++q) acc[q] = fma(a2s, y[q], fma(c, w[q], acc[q]));
    }
    const double H = 2.0 * (quad_pick(acc) + hclq);
#ifdef GIK_BLK_PROF
    if (prof && (tid & 63) == 0) {
      const long long pt2 = __builtin_readcyclecounter();
      prof[8 + 6 * wave] += (double)(pt1 - pt0);     // store W, moment partials, barrier
      prof[9 + 6 * wave] += (double)(pt1a - pt1);    // closed form
      prof[10 + 6 * wave] += (double)(pt1b - pt1a);  // D w
      prof[11 + 6 * wave] += (double)(pt2 - pt1b);   // slot loops, quad combine
      prof[12 + 6 * wave] += 1.0;
    }
#endif
    return active ? H : 0.0;
  }

  // horizontal-space projector at the committed point (same algebra as WaveCtx::proj_setup)
  __device__ inline void proj_setup(int planar_proj_exact) {
    double own[K];
    row(sh_Y, live ? node : 0, own);
    const double lm = (live && part == 0) ? 1.0 : 0.0;
    const double am = active ? 1.0 : 0.0;
    if constexpr (K == 3) {
      const double y0 = own[0], y1 = own[1], y2 = own[2];
      double x[6] = {lm * y0 * y0, lm * y0 * y1, lm * y0 * y2, lm * y1 * y1, lm * y1 * y2,
                     lm * y2 * y2};
      sum_n<6>(x);
      const double X00 = x[0], X01 = x[1], X02 = x[2], X11 = x[3], X12 = x[4], X22 = x[5];
      const double a = X00 + X11, b = X12, c = -X02, d = X00 + X22, e = X01, f = X11 + X22;
      const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
      const double c11 = a * f - c * c, c12 = b * c - a * e, c22 = a * d - b * b;
      const double idet = 1.0 / (a * c00 + b * c01 + c * c02);
      Pm[0] = c00 * idet; Pm[1] = c01 * idet; Pm[2] = c02 * idet;
      Pm[3] = c01 * idet; Pm[4] = c11 * idet; Pm[5] = c12 * idet;
      Pm[6] = c02 * idet; Pm[7] = c12 * idet; Pm[8] = c22 * idet;
      G2[0] = a; G2[1] = b; G2[2] = c; G2[3] = b; G2[4] = d; G2[5] = e;
      G2[6] = c; G2[7] = e; G2[8] = f;
      const double e0 = part == 0 ? 1.0 : 0.0, e1 = part == 1 ? 1.0 : 0.0,
                   e2 = part == 2 ? 1.0 : 0.0;
      pk[0] = pk2[0] = am * (e1 * y0 - e0 * y1);
      pk[1] = pk2[1] = am * (e2 * y0 - e0 * y2);
      pk[2] = pk2[2] = am * (e2 * y1 - e1 * y2);
      vertical_basis(a, b, c, d, e, f, pk, Q);
    } else {
      const double y0 = own[0], y1 = own[1];
      double x[3] = {lm * y0 * y0, lm * y0 * y1, lm * y1 * y1};
      sum_n<3>(x);
      const double X00 = x[0], X01 = x[1], X11 = x[2];
      double u0, u1, u2, u3;
      if (planar_proj_exact) {
        const double it = 1.0 / (X00 + X11);
        u0 = 0.0; u1 = it; u2 = -it; u3 = 0.0;
      } else {
        double A[4][5] = {{X00 + X00, X01, X01, 0.0, 0.0},
                          {X01, X01 + X00, 0.0, X01, 1.0},
                          {X01, 0.0, X00 + X11, X01, -1.0},
                          {0.0, X01, X01, X11 + X11, 0.0}};
#pragma unroll
        for (int col = 0; col < 4; ++col) {
#pragma unroll
          for (int r = col + 1; r < 4; ++r) {
            const bool sw = fabs(A[r][col]) > fabs(A[col][col]);
#pragma unroll
            for (int t = 0; t < 5; ++t) {
              const double p = A[col][t], q = A[r][t];
              A[col][t] = sw ? q : p;
              A[r][t] = sw ? p : q;
            }
          }
          const double ip = 1.0 / A[col][col];
#pragma unroll
          for (int r = col + 1; r < 4; ++r) {
            const double fct = A[r][col] * ip;
#pragma unroll
            for (int t = col; t < 5; ++t) A[r][t] = fma(-fct, A[col][t], A[r][t]);
          }
        }
        u3 = A[3][4] / A[3][3];
        u2 = (A[2][4] - A[2][3] * u3) / A[2][2];
        u1 = (A[1][4] - A[1][2] * u2 - A[1][3] * u3) / A[1][1];
        u0 = (A[0][4] - A[0][1] * u1 - A[0][2] * u2 - A[0][3] * u3) / A[0][0];
      }
      const double e0 = part == 0 ? 1.0 : 0.0, e1 = 1.0 - e0;
      pk[0] = am * (e1 * y0 - e0 * y1);
      pk2[0] = am * (e0 * (y0 * u0 + y1 * u2) + e1 * (y0 * u1 + y1 * u3));
      Pm[0] = 1.0;
      G2[0] = sum1(pk2[0] * pk2[0]);
      Q[0] = 0.0;
    }
  }

  __device__ inline double proj(double Z) {
    double v[NC];
    if constexpr (K == 3) {
#pragma unroll
      for (int m = 0; m < NC; ++m) v[m] = Q[m] * Z;
      sum_n<NC>(v);
      return fma(-Q[2], v[2], fma(-Q[1], v[1], fma(-Q[0], v[0], Z)));
    }
#pragma unroll
    for (int m = 0; m < NC; ++m) v[m] = pk[m] * Z;
    sum_n<NC>(v);
    double out = Z;
#pragma unroll
    for (int m = 0; m < NC; ++m) {
      double o = 0.0;
#pragma unroll
      for (int q = 0; q < NC; ++q) o = fma(Pm[m * NC + q], v[q], o);
      out = fma(-pk2[m], o, out);
    }
    return out;
  }

  // same contract as WaveCtx::hess_proj_dot
  __device__ inline double hess_proj_dot(double delta, const double (&s_dpk)[NC], double &d_Hd,
                                         double (&hd_pk)[NC]) {
    return proj_dot(ehess(delta), delta, s_dpk, d_Hd, hd_pk);
  }
  __device__ inline double proj_dot(double H, double delta, const double (&s_dpk)[NC], double &d_Hd,
                                    double (&hd_pk)[NC]) {
    constexpr int NV = (K == 3) ? NC + 1 : NC + 2;
    double v[NV];
#pragma unroll
    for (int m = 0; m < NC; ++m) v[m] = pk[m] * H;
    v[NC] = delta * H;
    if constexpr (K == 2) v[NC + 1] = pk2[0] * H;
    sum_n<NV>(v);
    double out = H, dot = v[NC];
    double o[NC];
#pragma unroll
    for (int m = 0; m < NC; ++m) {
      o[m] = 0.0;
#pragma unroll
      for (int q = 0; q < NC; ++q) o[m] = fma(Pm[m * NC + q], v[q], o[m]);
      out = fma(-pk2[m], o[m], out);
      if constexpr (K == 2) dot = fma(-o[m], s_dpk[m], dot);
    }
    if constexpr (K == 2) {
      hd_pk[0] = fma(-o[0], G2[0], v[NC + 1]);
    } else {
#pragma unroll
      for (int m = 0; m < NC; ++m) hd_pk[m] = 0.0;  // see WaveCtx::hess_proj_dot
    }
    d_Hd = dot;
    return out;
  }
};

}  // namespace gik
