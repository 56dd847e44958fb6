// tools/exp/gik_quad3.hip.h -- four 3-D IK problems per wavefront (gfx950 / CDNA4): an EXPERIMENT of round 4,
// measured slower than rtr_wave_kernel and therefore not part of the library (docs/NOTEBOOK.md 9.9 has the numbers
// and how it was wired into gik_kernels.hip.h; tools/attic/dev_quad3_check.py is the check that was run).
//
// The throughput regime of the wavefront kernel (rtr_wave_kernel, one unknown per lane) is bound by
// instruction issue: 162 VALU instructions per Hessian product, two waves per SIMD at 82 % of the
// issue capacity (BASELINE configs[3]: 65536 KUKA goals, 1.7 G Hessian products).  This kernel applies the layout of the planar kernel
// (gik_quad.hip.h) to the 3-D arms: a problem is one 4-lane block column of the wave's four rows --
// the 16 lanes v_mfma_f64_4x4x4 sums over -- and a lane owns whole NODES: node i and, for graphs of
// more than 16 nodes (the 7-DOF arms have 18), node 16 + i.  Every solver scalar is a per-lane value,
// one instruction stream serves four problems, an inner product is a few multiply-adds and two MFMAs.
//
// The four slots are independent: each is either inside truncated CG or waiting for its outer
// step.  One turn of the kernel's loop runs the outer step (cost of the proposal, acceptance,
// gradient / Hessian constants / projector, stopping rules, start of the next tCG solve) for the
// slots that wait for it -- masked, the others idle -- and then ONE tCG iteration for every slot
// that is inside tCG.  A slot whose problem has met a stopping rule claims the next one at once.
// 3-D tCG solves take 5 to 150 iterations, so running them in step (as the planar kernel does with
// its 5 to 10) would leave most slots waiting for the longest.
//
// Arithmetic: trust_region.py's, as in rtr_solve_one's k = 3 branch (orthonormal vertical basis Q,
// delta kept horizontal through w = -P r, <delta, Hdelta> = <delta, H>) without its deferred model
// test and its residual prediction: reductions are cheap here, so every inner product the reference
// forms is reduced where the reference forms it.  The Hessian product forms s = y.w per term end
// (like the workgroup and node-per-lane kernels).  No retrace checkpoint: after a rejected
// step tCG is rerun, as the reference does.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gik_quad.hip.h"

namespace gik {

template <int DEG, int NS>
struct Quad3Ctx {
  static constexpr int ROWS = QUAD_NODES * NS;      // rows of a problem's table
  static constexpr int ROW_BYTES = 32;              // x, y, z, pad
  int lane, slot, li;
  bool has[NS];
  char *sh_P;        // [4][ROWS] rows of the point last given to cost()
  char *sh_W;        // [4][ROWS] rows of the direction given to ehess()
  int own[NS];       // byte offset of this lane's rows
  // slot (s, e): [11:0] byte offset of the neighbour's row (own row: padding), [12] the residual has no
  // lower clamp, [13] no upper clamp (see QuadCtx), [31:16] term index
  uint32_t sl[NS][DEG];
  double tg[NS][DEG];             // per problem: squared target distances
  // per committed point: ys = 2 a (Y_i - Y_j) (a = 1 where the term is active, else 0), cc = 2 c
  double ys[NS][DEG][3], cc[NS][DEG];
  double Q[NS][3][3];             // orthonormal vertical basis at x: [node][component][m]

  __host__ __device__ static constexpr size_t lds_bytes() {
    return 2 * (size_t)QUAD_SLOTS * ROWS * ROW_BYTES + sizeof(int) * 2 * QUAD_SLOTS;
  }

  // g_meta: the wavefront kernel's slot table [DEG][64] (lane = 3 node + component)
  __device__ inline void init(int lane_, int N, char *P, char *W, const uint32_t *g_meta) {
    lane = lane_;
    slot = (lane >> 2) & 3;
    li = ((lane >> 4) << 2) | (lane & 3);
    sh_P = P;
    sh_W = W;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int node = li + QUAD_NODES * s;
      has[s] = node < N;
      own[s] = (slot * ROWS + node) * ROW_BYTES;
#pragma unroll
      for (int e = 0; e < DEG; ++e) {
        const uint32_t m = has[s] ? g_meta[e * WAVE + 3 * node] : meta_pack(node, 0, 0, 0);
        const int kind = meta_kind(m);
        sl[s][e] = (uint32_t)((slot * ROWS + meta_j(m)) * ROW_BYTES) |
                   ((kind == GIK_TERM_EQ || kind == GIK_TERM_UPPER) ? 0x1000u : 0u) |
                   ((kind == GIK_TERM_EQ || kind == GIK_TERM_LOWER) ? 0x2000u : 0u) | ((uint32_t)meta_term(m) << 16);
        tg[s][e] = 0.0;
        cc[s][e] = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) ys[s][e][c] = 0.0;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int m = 0; m < 3; ++m) Q[s][c][m] = 0.0;
      put_row(sh_P, own[s], 0.0, 0.0, 0.0);
      put_row(sh_W, own[s], 0.0, 0.0, 0.0);
    }
    __builtin_amdgcn_wave_barrier();
  }

  __device__ static inline void put_row(char *base, int off, double a, double b, double c) {
    *reinterpret_cast<double2 *>(base + off) = make_double2(a, b);
    *reinterpret_cast<double *>(base + off + 16) = c;
  }
  __device__ static inline void get_row(const char *base, int off, double (&v)[3]) {
    const double2 ab = *reinterpret_cast<const double2 *>(base + off);
    v[0] = ab.x;
    v[1] = ab.y;
    v[2] = *reinterpret_cast<const double *>(base + off + 16);
  }
  __device__ inline int nb(int s, int e) const { return (int)(sl[s][e] & 0xfffu); }
  __device__ inline double residual(int s, int e, double u) const {
    const double lo = __hiloint2double((sl[s][e] & 0x1000u) ? (int)0xfff00000 : 0, 0);
    const double hi = __hiloint2double((sl[s][e] & 0x2000u) ? 0x7ff00000 : 0, 0);
    return fmin(fmax(u, lo), hi);
  }

  // per problem (divergent: only the lanes of the slot that starts problem b)
  __device__ inline void load_targets(const double *targets_b) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int e = 0; e < DEG; ++e) tg[s][e] = targets_b[sl[s][e] >> 16];
  }

  // f(x): lcost (costs.py:80-93) / jcost (:8-16); leaves the rows of x in sh_P.  Every term sits in the
  // slot lists of both of its nodes: counted twice, halved (exact).  Wave-uniform call.
  __device__ inline double cost(const double (&x)[NS][3]) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < NS; ++s) put_row(sh_P, own[s], x[s][0], x[s][1], x[s][2]);
    __builtin_amdgcn_wave_barrier();
    double f = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      double fs = 0.0;
#pragma unroll
      for (int e = 0; e < DEG; ++e) {
        double r[3];
        get_row(sh_P, nb(s, e), r);
        const double a0 = x[s][0] - r[0], a1 = x[s][1] - r[1], a2 = x[s][2] - r[2];
        const double d = fma(a2, a2, fma(a1, a1, a0 * a0));
        const double cl = residual(s, e, tg[s][e] - d);
        fs = fma(cl, cl, fs);
      }
      f += has[s] ? fs : 0.0;
    }
    return 0.5 * quad_sum(f);
  }

  // egrad at the point whose rows are in sh_P (lgrad / jgrad, costs.py:98-123, 20-35) and the per-slot
  // constants of the Hessian there.  No cross-lane step: may be called by some slots only.
  __device__ inline void commit(double (&g)[NS][3]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      double o[3];
      get_row(sh_P, own[s], o);
      double G0 = 0.0, G1 = 0.0, G2 = 0.0;
#pragma unroll
      for (int e = 0; e < DEG; ++e) {
        double r[3];
        get_row(sh_P, nb(s, e), r);
        const double a0 = o[0] - r[0], a1 = o[1] - r[1], a2 = o[2] - r[2];
        const double d = fma(a2, a2, fma(a1, a1, a0 * a0));
        const double cl = residual(s, e, tg[s][e] - d);
        const bool act = ((sl[s][e] & 0x3000u) == 0x3000u) || (cl != 0.0);
        const double c = -cl;
        ys[s][e][0] = act ? a0 + a0 : 0.0;
        ys[s][e][1] = act ? a1 + a1 : 0.0;
        ys[s][e][2] = act ? a2 + a2 : 0.0;
        cc[s][e] = c + c;
        G0 = fma(c, a0, G0);
        G1 = fma(c, a1, G1);
        G2 = fma(c, a2, G2);
      }
      g[s][0] = G0 + G0;
      g[s][1] = G1 + G1;
      g[s][2] = G2 + G2;
    }
  }

  // ehess(Y, W) (lhess / jhess, costs.py:175-207, 39-58) at the last commit():
  //   H_i = sum_j [ 4 a (y.w) y + 2 c w ],  y = Y_i - Y_j,  w = W_i - W_j     (4 a y y^T = ys ys^T)
  __device__ inline void ehess(const double (&w)[NS][3], double (&h)[NS][3]) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < NS; ++s) put_row(sh_W, own[s], w[s][0], w[s][1], w[s][2]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
      for (int e = 0; e < DEG; ++e) {
        double r[3];
        get_row(sh_W, nb(s, e), r);
        const double u0 = w[s][0] - r[0], u1 = w[s][1] - r[1], u2 = w[s][2] - r[2];
        const double t = fma(ys[s][e][2], u2, fma(ys[s][e][1], u1, ys[s][e][0] * u0));
        a0 = fma(t, ys[s][e][0], fma(cc[s][e], u0, a0));
        a1 = fma(t, ys[s][e][1], fma(cc[s][e], u1, a1));
        a2 = fma(t, ys[s][e][2], fma(cc[s][e], u2, a2));
      }
      h[s][0] = a0;
      h[s][1] = a1;
      h[s][2] = a2;
    }
  }

  // sum over the problem's nodes of the per-node inner product <a_i, b_i>
  __device__ inline double dot(const double (&a)[NS][3], const double (&b)[NS][3]) const {
    double v = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) v = fma(a[s][2], b[s][2], fma(a[s][1], b[s][1], fma(a[s][0], b[s][0], v)));
    return quad_sum(v);
  }
  __device__ inline double dotQ(int m, const double (&b)[NS][3]) const {
    double v = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) v = fma(Q[s][2][m], b[s][2], fma(Q[s][1][m], b[s][1], fma(Q[s][0][m], b[s][0], v)));
    return quad_sum(v);
  }

  // Orthonormal basis Q of the vertical space at x (PSDFixedRank.proj, fixed_rank_psd_sym.py:91-113; see
  // WaveCtx::proj_setup / vertical_basis): spanned by pk_m = Y E_m, Gram matrix M = [[a,b,c],[b,d,e],[c,e,f]],
  // Q = pk L^-T with M = L L^T.  A function of x alone (a slot whose point did not change gets the same bits).
  __device__ inline void proj_setup(const double (&x)[NS][3]) {
    double p00 = 0.0, p01 = 0.0, p02 = 0.0, p11 = 0.0, p12 = 0.0, p22 = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const double m = has[s] ? 1.0 : 0.0;
      const double y0 = m * x[s][0], y1 = m * x[s][1], y2 = m * x[s][2];
      p00 = fma(y0, y0, p00);
      p01 = fma(y0, y1, p01);
      p02 = fma(y0, y2, p02);
      p11 = fma(y1, y1, p11);
      p12 = fma(y1, y2, p12);
      p22 = fma(y2, y2, p22);
    }
    const double X00 = quad_sum(p00), X01 = quad_sum(p01), X02 = quad_sum(p02), X11 = quad_sum(p11),
                 X12 = quad_sum(p12), X22 = quad_sum(p22);
    const double a = X00 + X11, b = X12, c = -X02, d = X00 + X22, e = X01, f = X11 + X22;
    const double i00 = frsqrt(a);
    const double l10 = b * i00, l20 = c * i00;
    const double i11 = frsqrt(fma(-l10, l10, d));
    const double l21 = fma(-l20, l10, e) * i11;
    const double i22 = frsqrt(fma(-l21, l21, fma(-l20, l20, f)));
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const double m = has[s] ? 1.0 : 0.0;
      const double y0 = m * x[s][0], y1 = m * x[s][1], y2 = m * x[s][2];
      // pk_0 = (-y1, y0, 0), pk_1 = (-y2, 0, y0), pk_2 = (0, -y2, y1) per node (components 0, 1, 2)
      const double pk[3][3] = {{-y1, -y2, 0.0}, {y0, 0.0, -y2}, {0.0, y0, y1}};
#pragma unroll
      for (int cmp = 0; cmp < 3; ++cmp) {
        const double q0 = pk[cmp][0] * i00;
        const double q1 = fma(-l10, q0, pk[cmp][1]) * i11;
        const double q2 = fma(-l21, q1, fma(-l20, q0, pk[cmp][2])) * i22;
        Q[s][cmp][0] = q0;
        Q[s][cmp][1] = q1;
        Q[s][cmp][2] = q2;
      }
    }
  }
};

// The kernel body (instantiated in gik_kernels.hip.h: SolveArgs lives there).
template <int DEG, int NS, typename Args>
__device__ inline void rtr_quad3_body(const Args &a, double *smem) {
  using Ctx = Quad3Ctx<DEG, NS>;
  const int lane = threadIdx.x;
  char *sh_P = reinterpret_cast<char *>(smem);
  char *sh_W = sh_P + QUAD_SLOTS * Ctx::ROWS * Ctx::ROW_BYTES;
  int *sh_claim = reinterpret_cast<int *>(sh_W + QUAD_SLOTS * Ctx::ROWS * Ctx::ROW_BYTES);
  Ctx cx;
  cx.init(lane, a.N, sh_P, sh_W, a.slot_meta);
  const Params &p = a.p;
  const int NK = a.N * 3;
  const bool lead = cx.li == 0;
  const double Delta_bar = 10.0 + 3;            // typicaldist (fixed_rank_psd_sym.py:71-73), k = 3

  // per-slot state (equal in the 16 lanes of a slot)
  bool alive = false, fresh = false, more = true, in_tcg = false;
  int b = -1, kiter = 0, inner_total = 0, n_accept = 0;
  double x[NS][3], g[NS][3], eta[NS][3], Heta[NS][3], r[NS][3], w[NS][3], dl[NS][3];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int c = 0; c < 3; ++c) x[s][c] = g[s][c] = eta[s][c] = Heta[s][c] = r[s][c] = w[s][c] = dl[s][c] = 0.0;
  double fx = 0.0, Delta = 0.0, norm_grad = 0.0;
  // truncated CG (one solve at a time per slot)
  double r_r = 0.0, e_Pe = 0.0, e_Pd = 0.0, d_Pd = 0.0, model_value = 0.0, target2 = 0.0, Delta2 = 0.0;
  int j = 0, jx = 0, stop_tCG = TCG_MAX_INNER_ITER, stop_target = 0;
  bool bad = false;

  for (;;) {
    // ---------------- refill ----------------
    const bool want = !alive && more;
    if (quad_any(want)) {
      if (want && lead) {
        const unsigned int t = atomicAdd(a.work_counter, 1u);
        sh_claim[cx.slot] = t < (unsigned)a.B ? (int)t : -1;
      }
      __builtin_amdgcn_wave_barrier();
      if (want) {
        const int nb = sh_claim[cx.slot];
        if (nb >= 0) {
          b = nb;
          alive = fresh = true;
          in_tcg = false;
          cx.load_targets(a.targets + (size_t)b * a.T);
#pragma unroll
          for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int c = 0; c < 3; ++c)
              x[s][c] = cx.has[s] ? a.Y_init[(size_t)b * NK + (cx.li + QUAD_NODES * s) * 3 + c] : 0.0;
          kiter = inner_total = n_accept = 0;
          Delta = Delta_bar / 8.0;                   // trust_region.py:134-135,164
          bad = false;
        } else {
          more = false;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (!quad_any(alive)) break;

    // ---------------- outer step for the slots that are not inside tCG (:248-422; fresh: :159-161) -------------
    const bool outer = alive && !in_tcg;
    if (quad_any(outer)) {
      const bool step = outer && !fresh && !bad;
      if (step) inner_total += jx + 1;
      const bool tr = a.has_trace && step && lead && kiter < a.trace.cap;
      if (tr) {
        const size_t q = (size_t)b * a.trace.cap + kiter;
        a.trace.d_Delta[q] = Delta;
        a.trace.d_numit[q] = jx;
        a.trace.d_stop[q] = stop_tCG;
        a.trace.d_f_before[q] = fx;
      }
      // proposal; the slots inside tCG pass their own point (cost() rewrites every row of the table, and a
      // slot's commit() reads only its own problem's rows: nothing of theirs is disturbed)
      double xp[NS][3];
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int c = 0; c < 3; ++c) xp[s][c] = step ? x[s][c] + eta[s][c] : x[s][c];          // :248 retr
      const double fx_prop = cx.cost(xp);                    // :251 (fresh: :159)
      const double gd0 = cx.dot(g, eta), gd1 = cx.dot(eta, Heta);
      double rhonum = fx - fx_prop;                          // :255
      double rhoden = -gd0 - 0.5 * gd1;                      // :256
      const double rho_reg = fmax(1.0, fabs(fx)) * 2.220446049250313e-16 * p.rho_regularization;   // :287
      rhonum += rho_reg;                                     // :288
      rhoden += rho_reg;                                     // :289
      const bool model_decreased = rhoden >= 0.0;            // :311
      const double rho = rhonum / rhoden;                    // :317
      if (step) {
        if (rho < 0.25 || !model_decreased || !(rho == rho)) {                      // :336
          Delta = Delta / 4.0;                               // :338
        } else if (rho > 0.75 && (stop_tCG == TCG_NEGATIVE_CURVATURE || stop_tCG == TCG_EXCEEDED_TR)) {
          Delta = fmin(2.0 * Delta, Delta_bar);              // :357-361
        }
      }
      const bool accept = step && model_decreased && rho > p.rho_prime;             // :382
      if (accept || (outer && fresh)) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int c = 0; c < 3; ++c) x[s][c] = xp[s][c];    // :385
        fx = fx_prop;                                        // :386
        cx.commit(g);                                        // :387 (fresh: :160)
      }
      if (accept) ++n_accept;
      // projector, ||grad|| and the vertical part of grad are functions of (x, grad): recomputed for every
      // slot, the same bits again where nothing was accepted and for the slots inside tCG
      cx.proj_setup(x);
      const double gg = cx.dot(g, g);
      const double rq0 = cx.dotQ(0, g), rq1 = cx.dotQ(1, g), rq2 = cx.dotQ(2, g);
      if (outer) norm_grad = sqrt(gg);                       // :388 (fresh: :161)
      if (tr) {
        const size_t q = (size_t)b * a.trace.cap + kiter;
        a.trace.d_gradnorm_after[q] = norm_grad;
        a.trace.d_accept[q] = accept ? 1 : 0;
      }
      if (step) ++kiter;                                     // :394
      // :414-416 stopping criterion (pymanopt 0.2.5 order: maxiter before gradnorm)
      const bool isnan = !(norm_grad == norm_grad) || !(fx == fx);
      int stop = -1;
      if (outer && bad) stop = 2;
      else if (step && kiter >= p.maxiter) stop = 1;
      else if (step && norm_grad < p.mingradnorm) stop = 0;
      else if (outer && (isnan || (a.dbg & 2))) stop = 2;
      const bool fin = outer && stop >= 0;
      if (fin) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
          if (cx.has[s]) {
#pragma unroll
            for (int c = 0; c < 3; ++c) a.Y_out[(size_t)b * NK + (cx.li + QUAD_NODES * s) * 3 + c] = x[s][c];
          }
        if (lead) {
          gik_stats st;
          st.f = fx;
          st.gradnorm = norm_grad;
          st.iterations = kiter;
          st.inner_total = inner_total;
          st.stop = stop;
          st.n_accept = n_accept;
          st.inner_executed = inner_total;
          st.flags = 0;
          st.stepsize = Delta;
          a.stats[b] = st;
        }
        alive = false;
      }
      // start of the next truncated-CG solve (:436-491)
      const bool go = outer && !fin;
      if (go) {
        Delta2 = Delta * Delta;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            eta[s][c] = 0.0;                                 // :444
            Heta[s][c] = 0.0;                                // :445
            r[s][c] = g[s][c];                               // :448
            // w = -(horizontal part of r) (see rtr_solve_one: the reference never projects the gradient)
            w[s][c] = fma(rq2, cx.Q[s][c][2], fma(rq1, cx.Q[s][c][1], fma(rq0, cx.Q[s][c][0], -g[s][c])));
            dl[s][c] = w[s][c];                              // :469
          }
        r_r = gg;                                            // :455 (r = grad)
        const double target = norm_grad * fmin(norm_grad, p.kappa);   // rhs of :572 (theta = 1)
        target2 = target * target;
        stop_target = (p.kappa < norm_grad) ? TCG_REACHED_TARGET_LINEAR : TCG_REACHED_TARGET_SUPERLINEAR;
        e_Pe = 0.0;
        e_Pd = 0.0;
        d_Pd = gg;                                           // :464-471 (precon = identity)
        model_value = 0.0;                                   // :485
        stop_tCG = TCG_MAX_INNER_ITER;                       // :491
        jx = p.maxinner - 1;                                 // Python leaves j at the last index
        j = 0;
        in_tcg = p.maxinner > 0;
      }
      if (outer) fresh = false;
    }

    // ---------------- one iteration of truncated CG for the slots inside it (:495-597) ----------------
    const bool act0 = alive && in_tcg;
    if (quad_any(act0)) {
      bool act = act0;
      double H[NS][3];
      cx.ehess(dl, H);                                       // :497
      const double u0 = cx.dotQ(0, H), u1 = cx.dotQ(1, H), u2 = cx.dotQ(2, H);
      const double d_Hd = cx.dot(dl, H);                     // :500 (delta is horizontal: <delta, P H> = <delta, H>)
      double Hd[NS][3];
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          Hd[s][c] = fma(-cx.Q[s][c][2], u2, fma(-cx.Q[s][c][1], u1, fma(-cx.Q[s][c][0], u0, H[s][c])));
      const bool nan = act && !(d_Hd == d_Hd);
      if (nan) {
        bad = true;
        in_tcg = false;
      }
      act = act && !nan;
      const double alpha = r_r * frcp(d_Hd);                 // :503
      const double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd;     // :506
      const bool exb = act && (d_Hd <= 0.0 || e_Pe_new >= Delta2);                  // :509
      if (exb) {
        const double tau = (-e_Pd + sqrt(e_Pd * e_Pd + d_Pd * (Delta2 - e_Pe))) / d_Pd;   // :514
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            eta[s][c] = eta[s][c] + tau * dl[s][c];          // :516
            Heta[s][c] = Heta[s][c] + tau * Hd[s][c];        // :521
          }
        stop_tCG = (d_Hd <= 0.0) ? TCG_NEGATIVE_CURVATURE : TCG_EXCEEDED_TR;        // :531-534
        jx = j;
        in_tcg = false;
      }
      act = act && !exb;
      if (quad_any(act)) {
        double ne[NS][3], nH[NS][3], nr[NS][3], gh[NS][3];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            ne[s][c] = eta[s][c] + alpha * dl[s][c];         // :538
            nH[s][c] = Heta[s][c] + alpha * Hd[s][c];        // :542
            nr[s][c] = r[s][c] + alpha * Hd[s][c];           // :561
            gh[s][c] = fma(0.5, nH[s][c], g[s][c]);
          }
        const double new_model_value = cx.dot(ne, gh);       // :551 <eta, grad> + 1/2 <eta, Heta>
        const double new_r_r = cx.dot(nr, nr);               // :564
        const bool exm = act && (new_model_value >= model_value);                   // :552
        if (exm) {
          stop_tCG = TCG_MODEL_INCREASED;
          jx = j;
          in_tcg = false;
        }
        act = act && !exm;
        if (act) {
          e_Pe = e_Pe_new;                                   // :537
          model_value = new_model_value;
#pragma unroll
          for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              eta[s][c] = ne[s][c];                          // :556-558
              Heta[s][c] = nH[s][c];
              r[s][c] = nr[s][c];                            // :561
            }
        }
        const bool ext = act && (j >= p.mininner && new_r_r <= target2);            // :572
        if (ext) {
          stop_tCG = stop_target;
          jx = j;
          in_tcg = false;
        }
        act = act && !ext;
        const bool exi = act && (j + 1 >= p.maxinner);       // :495 exhausted: stop stays MAX_INNER_ITER
        if (exi) {
          jx = p.maxinner - 1;
          in_tcg = false;
        }
        act = act && !exi;
        if (act) {
          const double beta = new_r_r * frcp(r_r);           // :592
          r_r = new_r_r;                                     // :589
#pragma unroll
          for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              w[s][c] = fma(-alpha, Hd[s][c], w[s][c]);
              dl[s][c] = fma(beta, dl[s][c], w[s][c]);       // :593
            }
          e_Pd = beta * (e_Pd + alpha * d_Pd);               // :596
          d_Pd = r_r + beta * beta * d_Pd;                   // :597
          ++j;
        }
      }
    }
  }
}

}  // namespace gik
