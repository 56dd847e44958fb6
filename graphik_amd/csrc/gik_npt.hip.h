// graphik_amd/csrc/gik_npt.hip.h -- node-per-lane context: one WAVEFRONT per IK problem for graphs
// that do not fit the one-unknown-per-lane layout (N*k > 64, N <= 128, k = 3): UR10 +
// table_environment(), N = 116 nodes and 5612 residual terms (BASELINE configs[2]).
//
// Layout.  A lane owns up to TWO graph nodes (LDS rows 2*lane and 2*lane + 1) with all three
// components of each: every tangent vector of the solver is six fp64 registers per lane, an axpy
// six v_fma_f64, an inner product six multiply-adds and ONE wave reduction.  No k-lane exchange, no
// workgroup barrier anywhere: a single wavefront executes its DS instructions in order.  The
// workgroup kernel this replaces for such graphs (gik_block.hip.h: 512 threads, one unknown each,
// two block barriers per Hessian product, every wave repeating the solver's scalar chain) issued
// ~2900 wave instructions per product; this one ~650, so three problems share a CU (LDS bound) and
// a CU finishes a product every ~1.2 k cycles instead of every 3.5-4 k.
//
//   * Rigid clique (rows 0..n-1; the scene's anchors, every pair tied by an equality):
//     its share of lhess (costs.py:175-207) in closed form from 27 moments of the direction, exactly
//     the algebra of gik_block.hip.h (header there); each lane adds the contributions of its two
//     nodes, one 28-register transposing wave reduction (wave_sum28) folds them, the totals come
//     back as scalars (v_readlane).  cost() / commit() (once per outer iteration) walk the clique
//     pairs directly from both ends, partner row wave-uniform (LDS broadcast), the pair's target
//     from the per-problem triangle `sh_ctg` (each pair once).
//   * The other terms ("slot terms": 47 on the table scene) are evaluated ONCE per term, one term
//     per lane (TL per lane), as the reference's edge loop does: y = Y_i - Y_j and the residual c
//     are cached per accepted point, s = y . (W_i - W_j) is formed per edge -- the Gauss-Newton
//     part's round-off stays in range(J^T) (docs/NOTEBOOK.md 2) -- the term's vector goes through a
//     small LDS buffer and the two end nodes add it with opposite signs (gather lists per lane,
//     fixed order: the reference's accumulation order per row).  One t per term at both ends keeps
//     the gradient's round-off horizontal (NOTEBOOK 2, finding 1).
//
// Drives rtr_solve_vec<> (gik_rtrv.hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gik_wave.hip.h"

namespace gik {

constexpr int NPT_MAXN = 2 * WAVE;     // rows
constexpr int NPT_RS = 4;              // LDS row stride (doubles)

// launch-invariant tables (device pointers; gik_template_create)
struct NptTabs {
  const int *node_of_row;          // [128] the caller's node held by thread slot NS * thread + s, -1 = none
  const unsigned char *prow_of_slot;   // [128] its row in the point table sh_P (clique nodes: cbase + rank)
  const int *clq_pair_term;        // [n_pairs] target index of clique pair (a < b: clique ranks), p = a n - a (a + 1) / 2 + b - a - 1
  const uint32_t *term_rec;        // [TL][64] row_i | row_j << 8 | kind << 16 | wslot_i << 18 | wslot_j << 25; kind 0 = padding
  const int *term_tgt;             // [TL][64] target index of the lane's term, -1 = padding
  const unsigned short *gather;    // [DEG0 + DEG1][threads] row of the +-t table the thread adds: 2 slot (+t) or 2 slot + 1 (-t);
                                   // padding = the zero row 2 * n_terms
  const unsigned char *wslot_of_row;   // [128] (per thread slot) row of the compact direction table that publishes the node, 255 = none
  int n_clq, n_pairs, DEG0, DEG1, n_wrows, TL;
  int clq_euclid;                  // 1: look for point coordinates behind the clique's target distances
  int cbase;                       // first row of the clique (its nodes take rows cbase .. cbase + n_clq - 1, rank = row - cbase)
  int n_rows;                      // rows in use (even)
  int n_terms;                     // slot terms (the first n_terms of the TL * 64 term slots, all evaluated by wavefront 0)
  int term_sync;                   // two waves per problem: 1 = end nodes of slot terms sit in both wavefronts
  int n_helped;                    // one node per lane: lanes 2 i (i < n_helped) hold the busiest nodes, lane 2 i + 1 gathers the
                                   // second half of lane 2 i's list (the gather is as long as the longest list of any lane)
};

// Sum 24 independent per-lane values over the wavefront by transposing exchanges (the scheme of
// wave_sum_n<8>, five levels deep): each level pairs the registers, a lane keeps one value of the
// pair and receives the partner lane's share of it -- v_permlane32_swap / v_permlane16_swap on lane
// bits 5 and 4, select exchanges on bits 3, 2, 1 -- and a final add folds bit 0.  Returns ONE
// register: value q's total in the lanes with (bit5, bit4, bit3, bit2, bit1) = (q & 1, q & 2, q & 4,
// q & 8, q & 16), i.e. lane npt_lane_of(q) and its neighbour.  25 adds (a tree needs 24 per lane
// column), fixed data flow => bit-reproducible.
__host__ __device__ constexpr int npt_lane_of(int q) {
  return ((q & 1) << 5) | (((q >> 1) & 1) << 4) | (((q >> 2) & 1) << 3) | (((q >> 3) & 1) << 2) | (((q >> 4) & 1) << 1);
}
__device__ inline double wave_sum24_distributed(double (&v)[24]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < 24; q += 2) {     // lane bit 5
    lane_swap32(v[q], v[q + 1]);
    v[q] += v[q + 1];
  }
#pragma unroll
  for (int q = 0; q < 24; q += 4) {     // lane bit 4
    lane_swap16(v[q], v[q + 2]);
    v[q] += v[q + 2];
  }
  // six registers left: v[0], v[4], ..., v[20]
  const bool h8 = lane & 8, h4 = lane & 4, h2 = lane & 2;
  double u[3];
#pragma unroll
  for (int q = 0; q < 3; ++q)           // lane bit 3 (row_ror:8 == lane ^ 8)
    u[q] = (h8 ? v[8 * q + 4] : v[8 * q]) + dpp_f64<0x128>(h8 ? v[8 * q] : v[8 * q + 4]);
  const double t0 = (h4 ? u[1] : u[0]) + lane_xor4(h4 ? u[0] : u[1]);   // lane bit 2
  const double t1 = (h4 ? 0.0 : u[2]) + lane_xor4(h4 ? u[2] : 0.0);
  double w = (h2 ? t1 : t0) + dpp_f64<0x4E>(h2 ? t0 : t1);              // lane bit 1 (quad_perm [2,3,0,1])
  w += dpp_f64<0xB1>(w);                                                // lane bit 0 (quad_perm [1,0,3,2])
  return w;
}

// TL: slot terms per lane of wavefront 0.  NS: graph nodes per lane.  NW: wavefronts per problem.
//   <TL, 2, 1>: one wavefront per problem, two nodes per lane (no barrier at all);
//   <TL, 1, 2>: one node per lane, two wavefronts (128 threads) per problem: half the per-lane state and
//               per-node work, every reduction finished through LDS behind a workgroup barrier (two
//               per Hessian product: the moments, and the solver's eight inner products).
//   <TL, 1, 4, true>: one node per lane, FOUR wavefronts per problem -- graphs of 129 .. 255 nodes (round 5: a scene
//               with more than 112 obstacles used to be refused).  CTG_GLOBAL: the clique's target triangle --
//               n (n - 1) / 2 doubles, 169 KB at 206 anchors -- lives in a per-workgroup slice of global memory
//               instead of LDS; it is read by the once-per-outer-iteration walks of cost() / clique_dw() only,
//               never by a Hessian product.
template <int TL, int NS, int NW, bool CTG_GLOBAL = false>
struct NptCtx {
  static_assert((NS == 2 && NW == 1) || NS == 1, "two nodes per lane: the one-wavefront layout only");
  static_assert(NS * NW == 2 || (NS == 1 && NW == 4), "128 rows, or 256 on four wavefronts");
  static constexpr int ROWS = WAVE * NS * NW;   // thread slots = rows of the tables below
  static constexpr int NE = 3 * NS;  // entries per lane
  static constexpr int NT = WAVE * NW;   // threads per problem
  static constexpr int NC = 3;
  static constexpr bool HAS_CK = true;
  static constexpr bool AGE_PRIORITY = false;

  int tid, lane, wave;
  bool live[NS];
  double lm[NS];             // 1.0 / 0.0
  int gnode[NS];
  int prow[NS];              // element offset of the node's row in sh_P
  double g_own, g_recv;      // gather: 1.0 if the lane's list is its own node's / if lane + 1 holds the rest of it
  // rigid clique (rows cbase .. cbase + n_clq - 1)
  int n_clq, n_pairs, cbase;
  double n_count;
  bool inq[NS];
  int crank[NS], offc[NS];   // clique rank of the slot's node (0 if not in the clique), offset of its triangle row
  double yt[NS][3], y2t[NS], cw[NS], cS[NS], rD[NS];
  double Xr[NS][3], rr[NS];
  double Syy[6], trS;        // uniform (kept in vector registers): sum y~ y~^T of the committed point
  double mscale;             // per-lane scale of the distributed moment totals (ehess)
  bool lowrank;
  bool dense_dw;             // closed form needs the O(n) product (targets are not distances of points)
  // slot terms, TL per lane (wavefront 0)
  int t_pi[TL], t_pj[TL], t_wi[TL], t_wj[TL], t_kind[TL];
  bool t_on[TL];
  double t_tgt[TL], t_y[TL][3], t_c[TL], t_a2[TL];
  int w_addr[NS];            // element offset of the slot's row in the compact direction table, -1 = not published
  int DEG0, DEG1, n_wrows, n_terms, term_sync;
  // LDS
  double *sh_P;              // [n_rows][4] proposal (cost) / committed point
  double *sh_W;              // [n_wrows + 1][4] direction rows of the nodes that carry slot terms (+ a zero row)
  double *sh_T;              // [2 n_terms + 1][4] per-term vectors: row 2 q = +t_q, row 2 q + 1 = -t_q (+ a zero row)
  double *sh_ctg;            // [n_pairs] clique target distances, each pair once (CTG_GLOBAL: global memory)
  double *sh_red;            // NW > 1: [2][NW][32] cross-wave reduction scratch, double buffered
  int red_buf;
  static constexpr int NG = 8;   // packed gather words per lane: 16 entries (two 16-bit rows each)
  uint32_t gat[NG];          // rows of the +-t table this lane's nodes add, first node's list first
  // vertical space at the committed point: generators pk_m = Y E_m, Gram matrix M = L L^T
  // (vertical_basis, gik_wave.hip.h); yc = the lane's rows of the committed point
  double yc[NS][3];
  double L_i00, L_l10, L_l20, L_i11, L_l21, L_i22;
  double ck[4][NE];          // tCG checkpoint (register file)
  double gq[NS][3];          // clique share of egrad at the point of the last cost() call (see cost())
#ifdef GIK_NPT_PROF
  long long pf[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // developer build: cycles per phase of a product (wavefront 0 of problem 0)
  __device__ inline long long pf_now() const { return (long long)__builtin_readcyclecounter(); }
#define NPT_PF(i, t_prev) { const long long t_ = pf_now(); pf[i] += t_ - t_prev; t_prev = t_; }
#else
#define NPT_PF(i, t_prev)
#endif

  __host__ __device__ static constexpr size_t lds_bytes(int n_pairs, int n_wrows, int n_rows, int n_terms) {
    return sizeof(double) * ((size_t)n_rows * NPT_RS + (size_t)(n_wrows + 1) * NPT_RS +
                             (size_t)(2 * n_terms + 1) * NPT_RS + (CTG_GLOBAL ? 0 : (size_t)((n_pairs + 1) & ~1)) +
                             (NW > 1 ? 2 * NW * 32 : 0));
  }
  // doubles of global workspace per resident workgroup (CTG_GLOBAL)
  __host__ __device__ static constexpr size_t ctg_doubles(int n_pairs) { return CTG_GLOBAL ? (size_t)((n_pairs + 7) & ~7) : 0; }

  __device__ inline bool lead() const { return tid == 0; }
  __device__ inline void ck_put(int i, const double (&v)[NE]) {
#pragma unroll
    for (int e = 0; e < NE; ++e) ck[i][e] = v[e];
  }
  __device__ inline double ck_get(int i, int e) const { return ck[i][e]; }

  // all threads of the problem: LDS writes before, LDS reads after
  __device__ static inline void block_sync() {
    if constexpr (NW > 1) __syncthreads();
    else __builtin_amdgcn_wave_barrier();
  }

  // Sum over all threads of the problem; result uniform (identical bits in every wavefront: the wave
  // totals are added in wavefront order by everybody).
  template <int NV>
  __device__ inline void sum_n(double (&v)[NV]) {
    static_assert(NV <= 8, "reduction scratch");
    if constexpr (NW == 1) {
      wave_sum_n<NV>(v);
    } else {
      double *buf = sh_red + red_buf * (NW * 32);
      if constexpr (NV == 8) {
        // the lanes that hold the eight wave totals store them (one exec-masked write instead of
        // sixteen v_readlane); value q sits in the lanes with (bit5, bit4, bit3) = (q & 1, q & 2, q & 4)
        const double w = wave_sum8_distributed(v);
        const int q = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2);
        if ((lane & 7) == 0) buf[wave * 32 + q] = w;
      } else {
        wave_sum_n<NV>(v);
        if (lane == 0) {
#pragma unroll
          for (int q = 0; q < NV; ++q) buf[wave * 32 + q] = v[q];
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        double t = buf[q];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) t += buf[w2 * 32 + q];
        v[q] = t;
      }
      red_buf ^= 1;    // the next reduction writes the other buffer: one barrier per reduction
    }
  }
  __device__ inline double sum1(double x) {
    double v[1] = {x};
    sum_n<1>(v);
    return v[0];
  }
  // the 24 moment totals of ehess(), distributed (value q in lane npt_lane_of(q) and its neighbour).
  // Two wavefronts: publish the wavefront's totals, then -- behind work that does not depend on them --
  // sum24_finish() adds the other wavefront's (wavefront order: the same bits in both).
  __device__ inline void sum24_publish(double tot) {
    if constexpr (NW > 1) {
      double *buf = sh_red + red_buf * (NW * 32);
      const int q = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3) |
                    (((lane >> 1) & 1) << 4);
      buf[wave * 32 + q] = tot;        // (both lanes of a pair store the same value)
    }
  }
  __device__ inline double sum24_finish(double tot) {
    if constexpr (NW > 1) {
      const double *buf = sh_red + red_buf * (NW * 32);
      const int q = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3) |
                    (((lane >> 1) & 1) << 4);
      __syncthreads();
      tot = buf[q] + buf[32 + q];
#pragma unroll
      for (int w2 = 2; w2 < NW; ++w2) tot += buf[w2 * 32 + q];
      red_buf ^= 1;
    }
    return tot;
  }

  // once per kernel: LDS carve-up, launch-invariant tables (ctg_ws: this workgroup's slice of the global
  // workspace, CTG_GLOBAL only)
  __device__ inline void init(const NptTabs &nt, double *smem, double *ctg_ws = nullptr) {
    tid = threadIdx.x;
    lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    n_clq = nt.n_clq;
    n_pairs = nt.n_pairs;
    cbase = nt.cbase;
    n_count = (double)n_clq;
    DEG0 = __builtin_amdgcn_readfirstlane(nt.DEG0);
    DEG1 = __builtin_amdgcn_readfirstlane(nt.DEG1);
    n_wrows = nt.n_wrows;
    n_terms = nt.n_terms;
    term_sync = nt.term_sync;
    sh_P = smem;
    sh_W = sh_P + nt.n_rows * NPT_RS;
    sh_T = sh_W + (n_wrows + 1) * NPT_RS;
    if constexpr (CTG_GLOBAL) {
      sh_ctg = ctg_ws;
      sh_red = sh_T + (2 * n_terms + 1) * NPT_RS;
    } else {
      sh_ctg = sh_T + (2 * n_terms + 1) * NPT_RS;
      sh_red = sh_ctg + ((n_pairs + 1) & ~1);
    }
    red_buf = 0;
    for (int t = tid; t < nt.n_rows * NPT_RS; t += NT) sh_P[t] = 0.0;
    for (int t = tid; t < (n_wrows + 1) * NPT_RS; t += NT) sh_W[t] = 0.0;
    for (int t = tid; t < (2 * n_terms + 1) * NPT_RS; t += NT) sh_T[t] = 0.0;
    {   // the thread's gather lists, two entries per register
      const int zero_row = 2 * n_terms;
#pragma unroll
      for (int e = 0; e < 2 * NG; e += 2) {
        const unsigned lo = e < DEG0 + DEG1 ? nt.gather[e * NT + tid] : zero_row;
        const unsigned hi = e + 1 < DEG0 + DEG1 ? nt.gather[(e + 1) * NT + tid] : zero_row;
        gat[e >> 1] = lo | (hi << 16);
      }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int r = NS * tid + s;
      gnode[s] = r < ROWS ? nt.node_of_row[r] : -1;
      live[s] = gnode[s] >= 0;
      lm[s] = live[s] ? 1.0 : 0.0;
      const int pr = live[s] ? (int)nt.prow_of_slot[r] : 0;
      prow[s] = pr * NPT_RS;
      inq[s] = live[s] && pr >= cbase && pr < cbase + n_clq;
      crank[s] = inq[s] ? pr - cbase : 0;
      offc[s] = crank[s] * n_clq - crank[s] * (crank[s] + 1) / 2 - crank[s] - 1;
      const int ws = live[s] ? nt.wslot_of_row[r] : 255;
      w_addr[s] = ws == 255 ? -1 : ws * NPT_RS;
      rD[s] = 0.0;
      rr[s] = 0.0;
      y2t[s] = cw[s] = cS[s] = 0.0;
#pragma unroll
      for (int q = 0; q < 3; ++q) Xr[s][q] = yt[s][q] = yc[s][q] = 0.0;
    }
#pragma unroll
    for (int u = 0; u < TL; ++u) {
      // (wavefront 0 evaluates every slot term; the records of the others are inert padding)
      const uint32_t pad = ((uint32_t)n_wrows << 18) | ((uint32_t)n_wrows << 25);
      const uint32_t rec = wave == 0 ? nt.term_rec[u * WAVE + lane] : pad;
      t_pi[u] = (int)(rec & 0xffu) * NPT_RS;
      t_pj[u] = (int)((rec >> 8) & 0xffu) * NPT_RS;
      t_kind[u] = (int)((rec >> 16) & 3u);
      t_wi[u] = (int)((rec >> 18) & 0x7fu) * NPT_RS;
      t_wj[u] = (int)((rec >> 25) & 0x7fu) * NPT_RS;
      t_on[u] = wave == 0 && u * WAVE + lane < n_terms;
      t_tgt[u] = 0.0;
      t_c[u] = t_a2[u] = 0.0;
#pragma unroll
      for (int q = 0; q < 3; ++q) t_y[u][q] = 0.0;
    }
    {   // helper lanes of the gather (one node per lane)
      const bool helped = NS == 1 && wave == 0 && (lane >> 1) < nt.n_helped;
      g_recv = (helped && !(lane & 1)) ? 1.0 : 0.0;
      g_own = (helped && (lane & 1)) ? 0.0 : 1.0;
    }
    mscale = 1.0;
#pragma unroll
    for (int q = 0; q < 24; ++q) {
      const double sc = (q == 3 || q == 6 || q == 8) ? 2.0 : ((q >= 9 && q < 12) ? -2.0 : ((q >= 12 && q < 15) ? -1.0 : 1.0));
      if ((lane & ~1) == npt_lane_of(q)) mscale = sc;
    }
    lowrank = false;
    dense_dw = false;
#pragma unroll
    for (int q = 0; q < 6; ++q) Syy[q] = 0.0;
    trS = 0.0;
    L_i00 = L_l10 = L_l20 = L_i11 = L_l21 = L_i22 = 0.0;
    block_sync();
  }

  __device__ inline void row3(const double *M, int off, double (&o)[3]) const {
    const double2 a = *reinterpret_cast<const double2 *>(M + off);
    o[0] = a.x;
    o[1] = a.y;
    o[2] = M[off + 2];
  }
  __device__ inline void put3(double *M, int off, const double *v) const {
    *reinterpret_cast<double2 *>(M + off) = make_double2(v[0], v[1]);
    M[off + 2] = v[2];
  }
  // index of clique pair (slot's node, rank m) in the triangle (a valid address for m == own rank
  // and for slots outside the clique too; callers mask)
  __device__ inline int pair_index(int s, int m, int offm) const {
    const int i = crank[s] < m ? offc[s] + m : offm + crank[s];
    return i < 0 ? 0 : i;
  }
  __device__ static inline int tri_off(int m, int n) { return m * n - m * (m + 1) / 2 - m - 1; }

  // per problem: targets of the lane's terms into registers, the clique's target distances into
  // LDS; row sums of D; point coordinates behind the clique's targets (clique_coordinates)
  __device__ inline void load_problem(const double *tg_b, const NptTabs &nt) {
#pragma unroll
    for (int u = 0; u < TL; ++u) {
      const int ti = t_on[u] ? nt.term_tgt[u * WAVE + lane] : -1;
      t_tgt[u] = (tg_b && ti >= 0) ? tg_b[ti] : 0.0;
    }
    lowrank = false;
    dense_dw = false;
    if (n_clq) {
      for (int p = tid; p < n_pairs; p += NT) sh_ctg[p] = tg_b ? tg_b[nt.clq_pair_term[p]] : 0.0;
      block_sync();
      double sD[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) sD[s] = 0.0;
      for (int m = 0; m < n_clq; ++m) {
        const int offm = tri_off(m, n_clq);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const double d = sh_ctg[pair_index(s, m, offm)];
          sD[s] += (inq[s] && m != crank[s]) ? d : 0.0;
        }
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) rD[s] = sD[s];
      if (nt.clq_euclid && tg_b) lowrank = clique_coordinates();
      dense_dw = !lowrank;
    }
    block_sync();
  }

  // argmax over the clique rows of a per-node value; scratch: the pad column of sh_P
  __device__ inline int clique_argmax(const double (&val)[NS], double &best) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (live[s]) sh_P[prow[s] + 3] = inq[s] ? val[s] : -1.0;
    block_sync();
    double b = -1.0;
    int arg = 0;
    for (int j = 0; j < n_clq; ++j) {
      const double v = sh_P[(cbase + j) * NPT_RS + 3];
      if (v > b) {
        b = v;
        arg = j;
      }
    }
    block_sync();
    best = b;
    return __builtin_amdgcn_readfirstlane(arg);
  }
  __device__ inline double dist_to(int s, int a) const {   // D(slot's node, rank a), a uniform
    if (!inq[s] || a == crank[s]) return 0.0;
    return sh_ctg[pair_index(s, a, tri_off(a, n_clq))];
  }

  // Are the clique's target distances those of a point set in R^3?  Same construction as
  // BlockCtx::clique_coordinates: trilateration from rank 0, the node farthest from it, the node
  // farthest from their line, the node farthest from their plane; then every pair against its
  // target at 1e-12 of the largest distance.  On success Xr / rr hold the centred coordinates.
  __device__ inline bool clique_coordinates() {
    const bool ok = clique_coordinates_impl();
    // sh_P is the solver's from here on
    block_sync();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const double zero[3] = {0.0, 0.0, 0.0};
      if (live[s]) put3(sh_P, prow[s], zero);
      if (!ok) {
        rr[s] = 0.0;
        Xr[s][0] = Xr[s][1] = Xr[s][2] = 0.0;
      }
    }
    block_sync();
    return ok;
  }
  __device__ inline bool clique_coordinates_impl() {
    double m1, m2, m3;
    double d0[NS], x[NS], y[NS], z[NS], v[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) d0[s] = dist_to(s, 0);
    const int a1 = clique_argmax(d0, m1);
    if (!(m1 > 0.0)) return false;
    const double d01 = sqrt(m1);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      x[s] = (d0[s] + m1 - dist_to(s, a1)) / (2.0 * d01);
      v[s] = d0[s] - x[s] * x[s];
    }
    const int a2 = clique_argmax(v, m2);
    if (!(m2 > 1e-6 * m1)) return false;    // collinear
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (live[s]) sh_P[prow[s]] = x[s];
    block_sync();
    const double x2 = sh_P[(cbase + a2) * NPT_RS], y2 = sqrt(m2);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      y[s] = (d0[s] - dist_to(s, a2) + (x2 * x2 + m2) - 2.0 * x[s] * x2) / (2.0 * y2);
      v[s] = d0[s] - x[s] * x[s] - y[s] * y[s];
    }
    const int a3 = clique_argmax(v, m3);
    if (!(m3 > 1e-6 * m1)) return false;    // coplanar
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (live[s]) sh_P[prow[s] + 1] = y[s];
    block_sync();
    const double x3 = sh_P[(cbase + a3) * NPT_RS], y3 = sh_P[(cbase + a3) * NPT_RS + 1], z3 = sqrt(m3);
    double c[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      z[s] = (d0[s] - dist_to(s, a3) + (x3 * x3 + y3 * y3 + m3) - 2.0 * x[s] * x3 - 2.0 * y[s] * y3) / (2.0 * z3);
      const double q = inq[s] ? 1.0 : 0.0;
      c[0] += q * x[s];
      c[1] += q * y[s];
      c[2] += q * z[s];
    }
    sum_n<3>(c);
    const double inv_n = 1.0 / n_count;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const double q = inq[s] ? 1.0 : 0.0;     // slots outside the clique carry zeros
      Xr[s][0] = q * fma(-c[0], inv_n, x[s]);
      Xr[s][1] = q * fma(-c[1], inv_n, y[s]);
      Xr[s][2] = q * fma(-c[2], inv_n, z[s]);
      rr[s] = fma(Xr[s][2], Xr[s][2], fma(Xr[s][1], Xr[s][1], Xr[s][0] * Xr[s][0]));
      if (inq[s]) put3(sh_P, prow[s], Xr[s]);
    }
    block_sync();
    double bad = 0.0;
    const double tol = 1e-12 * m1;
    for (int m = 0; m < n_clq; ++m) {
      double r[3];
      row3(sh_P, (cbase + m) * NPT_RS, r);
      const int offm = tri_off(m, n_clq);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const double e0 = Xr[s][0] - r[0], e1 = Xr[s][1] - r[1], e2 = Xr[s][2] - r[2];
        const double e = fma(e2, e2, fma(e1, e1, e0 * e0)) - sh_ctg[pair_index(s, m, offm)];
        bad += (inq[s] && m != crank[s] && !(fabs(e) <= tol)) ? 1.0 : 0.0;
      }
    }
    const double nbad = sum1(bad);
    return nbad == 0.0;
  }

  // residual of a slot term at squared distance d: EQ d - D; hinges only where violated
  __device__ static inline double term_resid(int kind, double d, double tgt, bool &act) {
    const double c0 = d - tgt;
    act = (kind == GIK_TERM_EQ) || (kind == GIK_TERM_LOWER && c0 < 0.0) || (kind == GIK_TERM_UPPER && c0 > 0.0);
    return act ? c0 : 0.0;
  }

  // f(x) (lcost / jcost, costs.py:80-93, 8-16); leaves the rows of x in sh_P
  __device__ inline double cost(const double (&x)[NE]) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (live[s]) put3(sh_P, prow[s], &x[3 * s]);
    block_sync();
    double f = 0.0;
    if (wave == 0) {
#pragma unroll
      for (int u = 0; u < TL; ++u) {
        double a[3], b[3];
        row3(sh_P, t_pi[u], a);
        row3(sh_P, t_pj[u], b);
        const double y0 = a[0] - b[0], y1 = a[1] - b[1], y2 = a[2] - b[2];
        const double d = fma(y2, y2, fma(y1, y1, y0 * y0));
        bool act;
        const double c = term_resid(t_kind[u], d, t_tgt[u], act);   // padding: kind 0 -> 0
        f = fma(c, c, f);
      }
    }
    // Clique pairs: the walk of cost() also sums the clique's share of the gradient at this point, c_ij y_ij over
    // all partners (three multiply-adds more per pair): if the step is accepted, commit() is called for the SAME
    // point and would repeat the whole walk -- 23 k of the ~50 k cycles an accepted outer iteration costs outside
    // tCG.  Same pairs, same order, same arithmetic as the separate walk (c^2 == u^2 bit for bit); each pair is
    // counted in f by its lower rank.
#pragma unroll
    for (int s = 0; s < NS; ++s) gq[s][0] = gq[s][1] = gq[s][2] = 0.0;
    if (n_clq) {
#pragma unroll 2      // (4: the walk itself 5 % shorter, the kernel 0.5 % slower -- register allocation of the tCG loop)
      for (int m = 0; m < n_clq; ++m) {
        double r[3];
        row3(sh_P, (cbase + m) * NPT_RS, r);
        const int offm = tri_off(m, n_clq);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const double y0 = x[3 * s] - r[0], y1 = x[3 * s + 1] - r[1], y2 = x[3 * s + 2] - r[2];
          const double d = fma(y2, y2, fma(y1, y1, y0 * y0));
          const double c = (inq[s] && m != crank[s]) ? d - sh_ctg[pair_index(s, m, offm)] : 0.0;
          const bool mine = m > crank[s];      // (c == 0 outside the clique)
          f = fma(mine ? c : 0.0, c, f);
          gq[s][0] = fma(c, y0, gq[s][0]);
          gq[s][1] = fma(c, y1, gq[s][1]);
          gq[s][2] = fma(c, y2, gq[s][2]);
        }
      }
    }
    return sum1(f);
  }

  // +t and -t of the lane's term u into the table the end nodes gather from
  __device__ inline void put_term(int u, const double (&t)[3]) {
    if (t_on[u]) {
      const int off = 2 * (u * WAVE + lane) * NPT_RS;
      const double nt[3] = {-t[0], -t[1], -t[2]};
      put3(sh_T, off, t);
      put3(sh_T, off + NPT_RS, nt);
    }
  }

  // accept the point of the LAST cost() call (its rows are in sh_P; x: the lane's own entries): egrad (lgrad /
  // jgrad, costs.py:98-123, 19-35) into g, per-term and per-node constants of the Hessian refreshed
  __device__ inline void commit(const double (&x)[NE], double (&g)[NE]) {
    double acc[NS][3];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s][0] = acc[s][1] = acc[s][2] = 0.0;
    if (wave == 0) {
#pragma unroll
      for (int u = 0; u < TL; ++u) {
        double a[3], b[3];
        row3(sh_P, t_pi[u], a);
        row3(sh_P, t_pj[u], b);
        const double y0 = a[0] - b[0], y1 = a[1] - b[1], y2 = a[2] - b[2];
        const double d = fma(y2, y2, fma(y1, y1, y0 * y0));
        bool act;
        const double c = term_resid(t_kind[u], d, t_tgt[u], act);
        t_y[u][0] = y0;
        t_y[u][1] = y1;
        t_y[u][2] = y2;
        t_c[u] = c;
        t_a2[u] = act ? 2.0 : 0.0;     // a = omega + [L active] + [U active] (costs.py:185-199), doubled
        const double t[3] = {c * y0, c * y1, c * y2};
        put_term(u, t);
      }
    }
    // the clique's share was summed by the cost() call that published this point (commit() always follows the
    // cost() of the same point: rtr_solve_vec, kat_npt_kernel)
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[s][q] = gq[s][q];
    if constexpr (NW > 1) {
      if (term_sync) __syncthreads();
    }
    gather_terms(acc);
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int q = 0; q < 3; ++q) g[3 * s + q] = lm[s] * 2.0 * acc[s][q];
    if (n_clq) clique_refresh(x);
  }

  // add the vectors of the slot terms to their end nodes: + at the term's first node, - at its
  // second (G[i] += t, G[j] -= t; costs.py:120-121, 204-205), in the order of the lane's list (the
  // reference's accumulation order per row).  The lists sit in registers, two 16-bit rows of the
  // +-t table per word; list lengths are uniform bounds.
  __device__ inline void gather_terms(double (&acc)[NS][3]) const {
    const int n0 = DEG0, n = DEG0 + DEG1;
    // Groups of GG entries behind ONE uniform test each; the LDS reads of ALL groups are issued before
    // the first sum (a test per entry serialises their latencies -- measured nine round trips instead
    // of one; a DS instruction costs a lone wavefront 10-16 issue cycles, so the last group is as
    // short as the list allows: GG = 3 reads 9 entries for the table scene's 9, GG = 4 would read 12).
    // (the one-wavefront layout has no registers to spare for that: it reads and sums group by group)
    double ga[3] = {0.0, 0.0, 0.0};        // (one node per lane) the lane's list, which may be half of its neighbour's
    constexpr int GG = 3, NGRP = (2 * NG + GG - 1) / GG;
    constexpr bool BATCH = false;      // (measured on the two-wavefront layout: all reads up front 4.5 k cycles per product, group by group 4.1 k)
    double t[NGRP * GG][3];
    auto load_group = [&](int g) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < GG; ++i) {
        const int e = g * GG + i;
        if (e < 2 * NG) {
          const unsigned row = (e & 1) ? (gat[e >> 1] >> 16) : (gat[e >> 1] & 0xffffu);
          row3(sh_T, (int)row * NPT_RS, t[e]);
        }
      }
    };
    if constexpr (BATCH) {
#pragma unroll
      for (int g = 0; g < NGRP; ++g)
        if (g * GG < n) load_group(g);
    }
#pragma unroll
    for (int g = 0; g < NGRP; ++g) {
      if (g * GG < n) {
        if constexpr (!BATCH) load_group(g);
#pragma unroll
        for (int i = 0; i < GG; ++i) {
          const int e = g * GG + i;
          if (e < 2 * NG) {
            if constexpr (NS == 1) {
#pragma unroll
              for (int q = 0; q < 3; ++q) ga[q] += t[e][q];
            } else {
              const double m0 = e < n0 ? 1.0 : 0.0, m1 = 1.0 - m0;     // (uniform)
#pragma unroll
              for (int q = 0; q < 3; ++q) {
                acc[0][q] = fma(m0, t[e][q], acc[0][q]);
                acc[NS - 1][q] = fma(m1, t[e][q], acc[NS - 1][q]);
              }
            }
          }
        }
      }
    }
    if constexpr (NS == 1) {
      // an even lane's node also owns what the odd lane next to it gathered (quad_perm [1,1,3,3]); a helper
      // lane's own node carries no slot terms
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[0][q] += fma(g_recv, dpp_f64<0xF5>(ga[q]), g_own * ga[q]);
    }
  }

  // moments of the committed point over the clique: centroid, centred rows, Syy
  __device__ inline void clique_refresh(const double (&x)[NE]) {
    double c[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const double q = inq[s] ? 1.0 : 0.0;
#pragma unroll
      for (int a = 0; a < 3; ++a) c[a] = fma(q, x[3 * s + a], c[a]);
    }
    sum_n<3>(c);
    const double inv_n = 1.0 / n_count;
    double m[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const double q = inq[s] ? 1.0 : 0.0;
#pragma unroll
      for (int a = 0; a < 3; ++a) yt[s][a] = q * fma(-c[a], inv_n, x[3 * s + a]);
      y2t[s] = fma(yt[s][2], yt[s][2], fma(yt[s][1], yt[s][1], yt[s][0] * yt[s][0]));
      m[0] = fma(yt[s][0], yt[s][0], m[0]);
      m[1] = fma(yt[s][0], yt[s][1], m[1]);
      m[2] = fma(yt[s][0], yt[s][2], m[2]);
      m[3] = fma(yt[s][1], yt[s][1], m[3]);
      m[4] = fma(yt[s][1], yt[s][2], m[4]);
      m[5] = fma(yt[s][2], yt[s][2], m[5]);
    }
    sum_n<6>(m);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      Syy[q] = m[q];
      asm volatile("" : "+v"(Syy[q]));      // a vector register each: the scalar file is needed elsewhere
    }
    trS = (m[0] + m[3]) + m[5];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const double q = inq[s] ? 1.0 : 0.0;
      cw[s] = q * (fma(n_count, y2t[s], trS) - rD[s]);
      cS[s] = lowrank ? rr[s] - y2t[s] : -y2t[s];
    }
  }

  // Horizontal-space projector at the committed point (fixed_rank_psd_sym.py:91-113).  The vertical
  // space is spanned by pk_m = Y E_m; with their Gram matrix M = L L^T, Q = pk L^-T is orthonormal
  // (vertical_basis) and proj Z = Z - Q Q^T Z.  Q is never formed here: the lane keeps its rows of Y,
  //     vert_dots(Z)      -> the lane's share of r_m = <pk_m, Z>     (to be reduced),
  //     vert_coords(r, u) -> u = Q^T Z = L^-1 r                       (orthonormal components),
  //     vert_apply(u, Z)  -> Z - pk c with L^T c = u                   (= Z - Q u).
  __device__ inline void proj_setup(const double (&x)[NE]) {
    double m[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const double y0 = lm[s] * x[3 * s], y1 = lm[s] * x[3 * s + 1], y2 = lm[s] * x[3 * s + 2];
      yc[s][0] = y0;
      yc[s][1] = y1;
      yc[s][2] = y2;
      m[0] = fma(y0, y0, m[0]);
      m[1] = fma(y0, y1, m[1]);
      m[2] = fma(y0, y2, m[2]);
      m[3] = fma(y1, y1, m[3]);
      m[4] = fma(y1, y2, m[4]);
      m[5] = fma(y2, y2, m[5]);
    }
    sum_n<6>(m);
    const double X00 = m[0], X01 = m[1], X02 = m[2], X11 = m[3], X12 = m[4], X22 = m[5];
    // M = [[X00+X11, X12, -X02], [X12, X00+X22, X01], [-X02, X01, X11+X22]]
    const double a = X00 + X11, b = X12, c = -X02, d = X00 + X22, e = X01, f = X11 + X22;
    L_i00 = frsqrt(a);
    L_l10 = b * L_i00;
    L_l20 = c * L_i00;
    L_i11 = frsqrt(fma(-L_l10, L_l10, d));
    L_l21 = fma(-L_l20, L_l10, e) * L_i11;
    L_i22 = frsqrt(fma(-L_l21, L_l21, fma(-L_l20, L_l20, f)));
  }
  // generators: comp 0: (-y1, -y2, 0)   comp 1: (y0, 0, -y2)   comp 2: (0, y0, y1)
  __device__ inline void vert_dots(const double (&Z)[NE], double &r0, double &r1, double &r2) const {
    r0 = fma(yc[0][0], Z[1], -(yc[0][1] * Z[0]));
    r1 = fma(yc[0][0], Z[2], -(yc[0][2] * Z[0]));
    r2 = fma(yc[0][1], Z[2], -(yc[0][2] * Z[1]));
    if constexpr (NS == 2) {
      r0 = fma(yc[NS - 1][0], Z[NE - 2], fma(-yc[NS - 1][1], Z[NE - 3], r0));
      r1 = fma(yc[NS - 1][0], Z[NE - 1], fma(-yc[NS - 1][2], Z[NE - 3], r1));
      r2 = fma(yc[NS - 1][1], Z[NE - 1], fma(-yc[NS - 1][2], Z[NE - 2], r2));
    }
  }
  __device__ inline void vert_coords(const double (&r)[3], double (&u)[3]) const {
    u[0] = r[0] * L_i00;
    u[1] = fma(-L_l10, u[0], r[1]) * L_i11;
    u[2] = fma(-L_l21, u[1], fma(-L_l20, u[0], r[2])) * L_i22;
  }
  __device__ inline void vert_apply(const double (&u)[3], const double (&Z)[NE], double (&out)[NE]) const {
    const double c2 = u[2] * L_i22;
    const double c1 = fma(-L_l21, c2, u[1]) * L_i11;
    const double c0 = fma(-L_l20, c2, fma(-L_l10, c1, u[0])) * L_i00;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      out[3 * s] = fma(yc[s][2], c1, fma(yc[s][1], c0, Z[3 * s]));
      out[3 * s + 1] = fma(yc[s][2], c2, fma(-yc[s][0], c0, Z[3 * s + 1]));
      out[3 * s + 2] = fma(-yc[s][1], c2, fma(-yc[s][0], c1, Z[3 * s + 2]));
    }
  }

  // ehess(Y, W) (lhess / jhess, costs.py:175-207, 39-58) at the committed point.
  // Order of work (a lone wavefront cannot overlap latencies with another's instructions): the
  // direction rows of the term end nodes go to LDS first, the moment contributions are formed while
  // they land, the term vectors are computed and written, the reduction network runs while THEY
  // land, and the gather comes last.  No scheduling fences: the DS queue of a wavefront is in order
  // and the compiler keeps the LDS dependences (two wavefronts: the slot terms and their end nodes
  // all sit in wavefront 0 unless term_sync says otherwise).
  __device__ inline void ehess(const double (&W)[NE], double (&H)[NE]) {
#ifdef GIK_NPT_PROF
    long long pt = pf_now();
#endif
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (w_addr[s] >= 0) put3(sh_W, w_addr[s], &W[3 * s]);
    if constexpr (NW > 1) {
      if (term_sync) __syncthreads();
    }
    double hq[NS][3], acc[NS][3];
#pragma unroll
    for (int s = 0; s < NS; ++s) hq[s][0] = hq[s][1] = hq[s][2] = acc[s][0] = acc[s][1] = acc[s][2] = 0.0;
    double v[24];
    double wm[NS][3], ai[NS];
    if (n_clq) {
      // ---- 24 moments of the direction over the clique (header of gik_block.hip.h):
      //   0..2 Sw, 3..8 Ms = M + M^T (xx xy xz yy yz zz; diagonal halved), 9..11 T3, 12..14 U3 (- R3),
      //   15..23 P[p][a] at 15 + 3 a + p
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const double cm = inq[s] ? 1.0 : 0.0;
#pragma unroll
        for (int p = 0; p < 3; ++p) wm[s][p] = cm * W[3 * s + p];
        ai[s] = fma(yt[s][2], wm[s][2], fma(yt[s][1], wm[s][1], yt[s][0] * wm[s][0]));
      }
      {
        const double *y0 = yt[0], *w0 = wm[0];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          v[p] = w0[p];
          v[9 + p] = ai[0] * y0[p];
          v[12 + p] = -(cS[0] * w0[p]);                            // (|y~|^2 - r) w
#pragma unroll
          for (int a = 0; a < 3; ++a) v[15 + 3 * a + p] = Xr[0][a] * w0[p];
        }
        v[3] = y0[0] * w0[0];
        v[6] = y0[1] * w0[1];
        v[8] = y0[2] * w0[2];
        v[4] = fma(y0[1], w0[0], y0[0] * w0[1]);
        v[5] = fma(y0[2], w0[0], y0[0] * w0[2]);
        v[7] = fma(y0[2], w0[1], y0[1] * w0[2]);
      }
      if constexpr (NS == 2) {
        const double *y1 = yt[NS - 1], *w1 = wm[NS - 1];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          v[p] += w1[p];
          v[9 + p] = fma(ai[NS - 1], y1[p], v[9 + p]);
          v[12 + p] = fma(-cS[NS - 1], w1[p], v[12 + p]);
#pragma unroll
          for (int a = 0; a < 3; ++a) v[15 + 3 * a + p] = fma(Xr[NS - 1][a], w1[p], v[15 + 3 * a + p]);
        }
        v[3] = fma(y1[0], w1[0], v[3]);
        v[6] = fma(y1[1], w1[1], v[6]);
        v[8] = fma(y1[2], w1[2], v[8]);
        v[4] = fma(y1[1], w1[0], fma(y1[0], w1[1], v[4]));
        v[5] = fma(y1[2], w1[0], fma(y1[0], w1[2], v[5]));
        v[7] = fma(y1[2], w1[1], fma(y1[1], w1[2], v[7]));
      }
    }
    // ---- slot terms: t = 2 a (y . w) y + c w per term (costs.py:186-203), once per term ----
    if (wave == 0) {
#pragma unroll
      for (int u = 0; u < TL; ++u) {
        double a[3], b[3];
        row3(sh_W, t_wi[u], a);
        row3(sh_W, t_wj[u], b);
        const double w0 = a[0] - b[0], w1 = a[1] - b[1], w2 = a[2] - b[2];
        const double sd = fma(t_y[u][2], w2, fma(t_y[u][1], w1, t_y[u][0] * w0));
        const double a2s = t_a2[u] * sd;
        const double t[3] = {fma(a2s, t_y[u][0], t_c[u] * w0), fma(a2s, t_y[u][1], t_c[u] * w1),
                             fma(a2s, t_y[u][2], t_c[u] * w2)};
        put_term(u, t);
      }
    }
    // The end nodes add the term vectors while the moment totals cross over to the other wavefront:
    // the gather's LDS round trips hide behind the exchange (and its own behind the reduction network).
    NPT_PF(0, pt)      // direction rows, moment contributions, term vectors
    double tot_w = 0.0;
    if (n_clq) {
      tot_w = wave_sum24_distributed(v);
      sum24_publish(tot_w);
    }
    NPT_PF(1, pt)      // reduction network
    if constexpr (NW > 1) {
      if (term_sync) __syncthreads();
    }
    gather_terms(acc);
    NPT_PF(2, pt)      // gather
    if (n_clq) {
      // the 24 totals stay in ONE distributed register; its lanes are pre-scaled (diagonal of M + M^T
      // was reduced halved: x 2; T3: x -2; U3: x -1) and every total is fetched (v_readlane) where it is
      // used, by all of the lane's nodes at once -- groups fenced, so that no more than nine of them
      // occupy scalar registers at a time (all 24 up front overflow the SGPR file into spill lanes)
      const double tot = sum24_finish(tot_w) * mscale;
      NPT_PF(3, pt)    // exchange with the other wavefront
      auto mo = [&](int q) { return readlane_f64(tot, npt_lane_of(q)); };
      double z[NS][3];
      {   // Sw: y~ . Sw, (r_i - |y~|^2) Sw_q, w_q cw
        const double S0 = mo(0), S1 = mo(1), S2 = mo(2);
        const double Sv[3] = {S0, S1, S2};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          ai[s] = fma(n_count, ai[s], -fma(yt[s][2], S2, fma(yt[s][1], S1, yt[s][0] * S0)));   // n a_i - y~ . Sw
#pragma unroll
          for (int q = 0; q < 3; ++q) hq[s][q] = fma(cS[s], Sv[q], wm[s][q] * cw[s]);
        }
      }
      if constexpr (NS > 1) __builtin_amdgcn_sched_barrier(0);
      {   // - 2 T3_q - U3_q (+ R3_q): the same vector for every node of the clique
        const double c0 = mo(9) + mo(12), c1 = mo(10) + mo(13), c2 = mo(11) + mo(14);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const double cmk = inq[s] ? 1.0 : 0.0;
          hq[s][0] = fma(cmk, c0, hq[s][0]);
          hq[s][1] = fma(cmk, c1, hq[s][1]);
          hq[s][2] = fma(cmk, c2, hq[s][2]);
        }
      }
      if constexpr (NS > 1) __builtin_amdgcn_sched_barrier(0);
      {   // Ms = M + M^T (xx xy xz yy yz zz): (Ms y~)_q, tr M; and (Syy w)_q, y~_q g
        const double M0 = mo(3), M1 = mo(4), M2 = mo(5), M3 = mo(6), M4 = mo(7), M5 = mo(8);
        const double s_yw = 0.5 * ((M0 + M3) + M5);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const double *y = yt[s], *w = wm[s];
          const double gg = ai[s] + s_yw;                       // n a_i + tr M - y~ . Sw
          double z0 = fma(Syy[2], w[2], fma(Syy[1], w[1], Syy[0] * w[0]));
          double z1 = fma(Syy[4], w[2], fma(Syy[3], w[1], Syy[1] * w[0]));
          double z2 = fma(Syy[5], w[2], fma(Syy[4], w[1], Syy[2] * w[0]));
          z0 = fma(M2, y[2], fma(M1, y[1], fma(M0, y[0], z0)));
          z1 = fma(M4, y[2], fma(M3, y[1], fma(M1, y[0], z1)));
          z2 = fma(M5, y[2], fma(M4, y[1], fma(M2, y[0], z2)));
          z[s][0] = fma(y[0], gg, z0);
          z[s][1] = fma(y[1], gg, z1);
          z[s][2] = fma(y[2], gg, z2);
        }
      }
      if constexpr (NS > 1) __builtin_amdgcn_sched_barrier(0);
      if (lowrank) {      // - (P X_i)_q,  P[q][a] at 15 + 3 a + q
        const double P0 = mo(15), P1 = mo(16), P2 = mo(17), P3 = mo(18), P4 = mo(19), P5 = mo(20), P6 = mo(21),
                     P7 = mo(22), P8 = mo(23);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const double *X = Xr[s];
          z[s][0] = fma(-X[2], P6, fma(-X[1], P3, fma(-X[0], P0, z[s][0])));
          z[s][1] = fma(-X[2], P7, fma(-X[1], P4, fma(-X[0], P1, z[s][1])));
          z[s][2] = fma(-X[2], P8, fma(-X[1], P5, fma(-X[0], P2, z[s][2])));
        }
      }
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int q = 0; q < 3; ++q) hq[s][q] = fma(2.0, z[s][q], hq[s][q]);
      if (dense_dw) clique_dw(W, acc);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int q = 0; q < 3; ++q) H[3 * s + q] = lm[s] * 2.0 * (acc[s][q] + hq[s][q]);
    NPT_PF(4, pt)      // totals -> scalars, closed form
#ifdef GIK_NPT_PROF
    pf[9] += 1;
#endif
  }

  // (D w)_i, dense: only for targets that are not distances of points (an arbitrary D_goal through
  // gik_solve_batch, collinear or coplanar cliques).  O(n) per node and product.  The direction rows
  // of all nodes are published in the point table's place -- sh_P is only read between a cost() and the
  // commit() that follows it, and every cost() rewrites it -- and walked like the clique in cost().
  __device__ inline void clique_dw(const double (&W)[NE], double (&acc)[NS][3]) {
    block_sync();
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (live[s]) put3(sh_P, prow[s], &W[3 * s]);
    block_sync();
    for (int m = 0; m < n_clq; ++m) {
      const int offm = tri_off(m, n_clq);
      double wm[3];
      row3(sh_P, (cbase + m) * NPT_RS, wm);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const double dg = (inq[s] && m != crank[s]) ? sh_ctg[pair_index(s, m, offm)] : 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[s][q] = fma(dg, wm[q], acc[s][q]);
      }
    }
  }
};

}  // namespace gik
