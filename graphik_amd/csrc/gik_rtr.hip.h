// graphik_amd/csrc/gik_rtr.hip.h -- the Riemannian trust-region driver shared by the
// wave-per-problem and the workgroup-per-problem kernels.
//
// rtr_solve_one<Ctx>() is TrustRegions.solve (graphik/solvers/trust_region.py:112-434) with
// _truncated_conjugate_gradient (:436-599) inlined, written against a context `Ctx` that owns the
// problem data and provides
//     cost(x), commit() -> egrad entry, proj_setup(flag),
//     k = 3: ehess(delta) -> raw Hessian product entry, Q[3] (orthonormal vertical basis),
//            ck_put / ck_get (per-thread checkpoint slots in LDS, HAS_CK);
//     k = 2: hess_proj_dot(delta, s, d_Hd, hd), pk2[NC];
//     sum_n<NV>(v), sum1(x) (reductions over all unknowns, result uniform in every thread),
//     lead()                (the one thread that writes per-problem scalars).
// Every thread holds one entry of each tangent vector (or zero if it owns none).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "gik_wave.hip.h"

namespace gik {

// Branch conditions on solver scalars are identical in every thread of the problem; routing them
// through a ballot makes that explicit (the predicate lands in an SGPR pair, the branch is scalar)
// so the structurizer never builds exec-masked loops around the reductions.
#define UNI(cond) (__builtin_amdgcn_ballot_w64(cond) != 0ull)

struct RtrOut {
  double f, gradnorm;
  int iterations, inner_total, stop, n_accept;
  int inner_executed;   // Hessian products actually evaluated (<= the reference's count, see "Retrace")
  double Delta;         // trust-region radius at return
  int paused;           // returned after `slice_its` outer iterations without meeting a stopping rule
};

// A solve is exactly resumable from (x, Delta, counters): cost, gradient and projector are
// functions of x alone and are recomputed, the tCG checkpoint is dropped (resuming from it is
// bit-identical to rerunning tCG anyway).  The persistent kernels use this to time-slice long
// problems (see rtr_wave_kernel).
struct RtrResume {
  double Delta;
  int kiter, inner_total, inner_exec, n_accept;
  int resumed;
  int resumes;    // (bookkeeping of the kernels, not read by the solver)
};

// ||g||_F together with <g, pk2_m> in one reduction
template <typename Ctx>
__device__ inline double grad_norm_and_rho(Ctx &cx, double g, double (&rho0)[Ctx::NC]) {
  if constexpr (Ctx::NC == 3) {   // k = 3: the components of g along the orthonormal vertical basis
    double v[4] = {g * g, g * cx.Q[0], g * cx.Q[1], g * cx.Q[2]};
    cx.template sum_n<4>(v);
#pragma unroll
    for (int m = 0; m < 3; ++m) rho0[m] = v[m + 1];
    return sqrt(v[0]);
  }
  double v[Ctx::NC + 1];
  v[0] = g * g;
#pragma unroll
  for (int m = 0; m < Ctx::NC; ++m) v[m + 1] = g * cx.pk2[m];
  cx.template sum_n<Ctx::NC + 1>(v);
#pragma unroll
  for (int m = 0; m < Ctx::NC; ++m) rho0[m] = v[m + 1];
  return sqrt(v[0]);
}

// tau of trust_region.py:514 with <eta, delta> carried doubled
__device__ inline double boundary_tau(double e_Pd2, double d_Pd, double Delta2, double e_Pe) {
  const double e_Pd = 0.5 * e_Pd2;
  return (sqrt(fma(e_Pd, e_Pd, d_Pd * (Delta2 - e_Pe))) - e_Pd) / d_Pd;   // no contraction left open
}

// Tail spreading (wavefront kernel, two waves per SIMD).  Once the queue of fresh problems is empty
// the batch is a set of long-running problems scattered over the SIMDs: some SIMDs host two (each at
// 0.5-0.7 of a lone wave's speed), others none.  A wave that runs out of work and finds its SIMD
// empty becomes a HELPER: it reserves the SIMD (simd_run[sid] 0 -> 1), takes a hand-over ticket and
// posts a credit.  A running wave polls the credits every fourth outer iteration; if there is one
// and its own SIMD hosts two problems it takes the credit, pauses its solve -- a solve is exactly
// resumable from (x, Delta, counters) -- publishes it under the next ticket and turns helper itself.
// A SIMD's count only ever rises from 0 to 1 in this phase, so a problem moves at most once, and
// every published problem has a helper waiting for exactly that ticket.  Results are bit-identical
// to an unmigrated run (tests/test_full_size_gpu.py::test_tail_spreading_and_round_robin_are_bit_identical).
// Round-robin time slicing rides on the same hooks: while more problems are unfinished than the launch
// has waves, a problem yields its slot after `slice_its` outer iterations, so that the long problems
// -- unknown in advance -- are not the last to START; the yielder queues behind everything that
// waits (fresh problems first, then the yield queue in FIFO order; SolveArgs::y_*).
struct MigCtl {
  int *credits;      // helpers waiting on an empty SIMD, not yet matched with a donor
  int *simd_run;     // [MIG_SIMDS] problems running or reserved per physical SIMD
  int sid;           // this wave's SIMD (XCC, SE, SH, CU, SIMD bits of the hardware id registers)
  const unsigned int *fresh;    // ticket counter of the fresh problems (tickets < B)
  const int *y_avail;           // yield queue: entries nobody has a claim on yet (see SolveArgs)
  const unsigned int *y_tail;   // ... entries pushed so far
  int B;
  unsigned int y_cap;           // entries of the yield queue (no yield once it is nearly full)
  const unsigned int *done;     // problems finished so far
  int waves;                    // persistent waves of the launch: with B - done <= waves every unfinished
                                // problem can have a slot -- no more yields, helpers may commit to hand-overs
  int slice_cycles;             // shortest slice in cycles of the shader clock counter
};
enum { PAUSE_NONE = 0, PAUSE_DONATE = 1, PAUSE_YIELD = 2 };
constexpr int MIG_SIMDS = 1 << 14;

template <typename Ctx>
__device__ inline bool mig_poll(const Ctx &cx, const MigCtl &m) {
  int go = 0;
  if (cx.lead()) {
    if (__hip_atomic_load(m.credits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0 &&
        __hip_atomic_load(&m.simd_run[m.sid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 2) {
      if (__hip_atomic_fetch_add(m.credits, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 1)
        go = 1;
      else
        __hip_atomic_fetch_add(m.credits, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  return __builtin_amdgcn_readfirstlane(go) != 0;
}

// are there more unfinished problems than slots?  (one thread's loads, broadcast.)  Only then does a
// yield give a waiting problem a turn; with fewer, an entry in the yield queue is just another yielder
// on its way back to a slot, and yielding to it would go on for ever.
template <typename Ctx>
__device__ inline bool mig_anyone_waiting(const Ctx &cx, const MigCtl &m) {
  int w = 0;
  if (cx.lead()) {
    const unsigned int yt = __hip_atomic_load(m.y_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int dn = __hip_atomic_load(m.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (every wave of the launch may be between this test and its push: the margin is the wave count)
    w = yt + (unsigned)m.waves + 64u < m.y_cap && (unsigned)m.B > dn + (unsigned)m.waves;
  }
  return __builtin_amdgcn_readfirstlane(w) != 0;
}

// contexts whose Hessian product comes in two halves (ehess_begin / ehess_end: WaveCtxStrict)
template <typename Ctx, typename = void>
struct split_ehess : std::false_type {};
template <typename Ctx>
struct split_ehess<Ctx, std::enable_if_t<Ctx::SPLIT_EHESS>> : std::true_type {};

// THETA_ONE: compiled for the reference default theta = 1 (trust_region.py:92), where
// norm_r0 ** theta needs no pow().  The generic build evaluates pow() when theta != 1; inlined, its
// polynomial constants are hoisted to the per-problem setup and spilled to scratch by every problem
// (measured: +15 MB of HBM writes per 4096-goal launch), so the default path must not contain it.
// SLICE: compiled with the resume / pause hooks (workgroup-per-problem kernel).  The wave kernel
// does without: the extra scalar state costs it its last free registers (256 VGPRs + scratch,
// 972 -> 1114 cycles per iteration), more than time slicing returns there.
// MIG: SLICE with the pause decided by mig_poll() instead of a fixed slice length.
template <int K, bool THETA_ONE, bool SLICE, typename Ctx, bool MIG = false>
__device__ inline void rtr_solve_one(Ctx &cx, const Params &p, const gik_trace &trace, int has_trace,
                                     int dbg, double *dbg_buf, int b, double &x, RtrOut &out,
                                     const RtrResume &rs, int slice_its, const MigCtl *mig = nullptr) {
  static_assert(!MIG || SLICE, "tail spreading needs the resume / pause hooks");
  // The two-halves product (the next product's LDS round trip behind this step's updates) costs 22 VGPRs -- the gathered
  // rows live across the loop's back edge: 153 -> 175 -- which a lone wavefront has and three per SIMD do not.  The
  // MIG build is the one large batches run (two or three waves per SIMD, bound by the vector ALU, not by latency): it
  // keeps the one-piece product.  Same values either way.
  constexpr bool SPLIT = split_ehess<Ctx>::value && !MIG;
  const double Delta_bar = 10.0 + K;  // typicaldist (fixed_rank_psd_sym.py:71-73)
  const bool lead = cx.lead();
    double Delta = (SLICE && rs.resumed) ? rs.Delta : Delta_bar / 8.0;   // trust_region.py:134-135,164
    double fx = cx.cost(x);                 // :159
    double g = cx.commit();                 // :160  (also loads the slot constants at x)
    cx.proj_setup(p.planar_proj_exact);
    // ||grad|| (:161) and rho0_m = <grad, pk2_m> (start values of the tCG recurrences)
    double rho0[Ctx::NC];
    double norm_grad = grad_norm_and_rho(cx, g, rho0);
    int kiter = SLICE ? rs.kiter : 0, inner_total = SLICE ? rs.inner_total : 0,
        inner_exec = SLICE ? rs.inner_exec : 0, n_accept = SLICE ? rs.n_accept : 0, stop = 1;
    int slice_count = 0;
    long long slice_t0 = MIG ? (long long)__builtin_readcyclecounter() : 0;
    int paused = PAUSE_NONE;
    // ---- Retrace (k = 3 wave path) --------------------------------------------------------
    // A rejected step leaves x, g and the Hessian unchanged and divides the radius by 4
    // (:336-338, :382), so the reference's next tCG solve repeats the previous one operation for
    // operation until |eta| passes the smaller radius -- 11 % of all Hessian products on random
    // LWA4D goals.  Every tCG solve therefore leaves a checkpoint where it first meets its own
    // radius / 4 (vectors in LDS, scalars in registers); after a rejection the boundary step is
    // formed from the checkpoint with the same arithmetic the rerun would end in, bit for bit
    // (tests: results identical with GIK_DBG=16, which disables this).  A solve that ended inside
    // the smaller radius without meeting it (model / target / maxinner exits) is its own rerun.
    // inner_total keeps counting what the reference would have executed.
    constexpr bool RETRACE = (K == 3) && Ctx::HAS_CK;
    const bool retrace_on = RETRACE && !(dbg & 16);
    bool prev_rejected = false, ck_set = false, ck_neg = false;
    int ck_j = 0;
    double ck_T = 0.0, ck_e_Pe = 0.0, ck_e_Pd2 = 0.0, ck_d_Pd = 0.0;
    double last_Delta2 = 0.0, last_e_Pe = 0.0;
    double eta = 0.0, Heta = 0.0;
    int stop_tCG = TCG_MAX_INNER_ITER, j = 0;
    const bool prof = (dbg & 8) && b == 0 && dbg_buf;   // cycle counters (developer aid)
    long long prof_tcg = 0;
    const long long prof_t0 = prof ? (long long)__builtin_readcyclecounter() : 0;
    bool bad = UNI(!(fx == fx) || !(norm_grad == norm_grad));
    if (dbg & 2) bad = true;

    while (!bad) {
      // Wave priority grows with the age of the problem (s_setprio, 0..3).  Two waves share a SIMD's
      // fp64 pipe; the arbiter issues the ready wave of highest priority, so a long-running problem
      // -- the batch's critical path -- keeps close to the speed of a lone wave while the younger
      // co-resident problem fills the issue slots it leaves.  Results do not depend on it.
      if constexpr (Ctx::AGE_PRIORITY) {
        if (kiter == 32) __builtin_amdgcn_s_setprio(1);
        else if (kiter == 128) __builtin_amdgcn_s_setprio(2);
        else if (kiter == 512) __builtin_amdgcn_s_setprio(3);
        if constexpr (SLICE) {      // a resumed problem keeps the priority of its age
          if (rs.resumed && slice_count == 0) {
            if (kiter > 512) __builtin_amdgcn_s_setprio(3);
            else if (kiter > 128) __builtin_amdgcn_s_setprio(2);
            else if (kiter > 32) __builtin_amdgcn_s_setprio(1);
          }
        }
      }
      // -------------- _truncated_conjugate_gradient (trust_region.py:436-599) -------------
      const double Delta2 = Delta * Delta;
      bool reuse = false;
      if constexpr (RETRACE) {
        if (retrace_on && prev_rejected) {
          if (Delta2 == last_Delta2) {
            reuse = true;                                  // same radius: the identical solve
          } else if (ck_set && Delta2 == ck_T) {           // the rerun stops at the checkpoint
            const double tau = boundary_tau(ck_e_Pd2, ck_d_Pd, Delta2, ck_e_Pe);       // :514
            eta = fma(tau, cx.ck_get(2), cx.ck_get(0));                                // :516
            Heta = fma(tau, cx.ck_get(3), cx.ck_get(1));                               // :521
            stop_tCG = ck_neg ? TCG_NEGATIVE_CURVATURE : TCG_EXCEEDED_TR;
            j = ck_j;
            ck_set = false;
            reuse = true;
          } else if (stop_tCG != TCG_NEGATIVE_CURVATURE && stop_tCG != TCG_EXCEEDED_TR &&
                     last_e_Pe < Delta2) {
            reuse = true;                                  // never met the smaller radius either
          }
        }
      }
      last_Delta2 = Delta2;
      const long long prof_t1 = prof ? (long long)__builtin_readcyclecounter() : 0;
      int executed = 0;
      if (reuse) {
        // eta, Heta, j, stop_tCG are the rerun's
      } else if constexpr (K == 3) {
        double eta_l = 0.0, Heta_l = 0.0;          // :444-445 (loop-local: the outer copies must
                                                   // not constrain the loop's register allocation)
        stop_tCG = TCG_MAX_INNER_ITER;             // :491
        int extra = 0;
        // One reduction per inner iteration.  The projector P = I - Q Q^T is orthogonal, so for
        // HORIZONTAL delta and w (= -P r, see below), with H = ehess(delta), u = Q^T H,
        // Hdelta = H - Q u:
        //     <delta, Hdelta> = <delta, H>     <r, Hdelta> = -<w, H>     |Hdelta|^2 = |H|^2 - u.u
        // Everything the textbook loop reduces AFTER its vector updates is either predicted from
        // these, or reduced one iteration late next to them:
        //   * <r', r'> = <r, r> + 2 alpha <r, Hdelta> + alpha^2 |Hdelta|^2 gives beta (:592) and
        //     the residual test (:572) at once; the exact <r', r'> arrives with the next
        //     iteration's reduction and is what alpha (:503) uses, as in the reference.  The
        //     prediction loses digits when the residual drops sharply; below 1e-3 <r, r> it is
        //     re-reduced directly (rare).
        //   * the model value <eta, g> + 1/2 <eta, Heta> (:551) of the CURRENT eta rides along as
        //     one more summand, so the "model increased" test (:552) of step j-1 is taken at the top
        //     of step j -- before anything of step j is used, which is the reference's order --
        //     and rolls eta back one step.  It is a fresh inner product, not a recurrence: the
        //     test exists to catch round-off stagnation and must see the same noise as the
        //     reference's.  Where the reference would test the model before leaving (residual
        //     target reached, inner iterations exhausted) the pending test is reduced on the spot.
        // One branch covers all exits.  Reported inner iteration counts are the reference's (a
        // rolled-back step has cost one extra Hessian product that is not counted).
        double r = g;                              // :448
        const double r0_r0 = norm_grad * norm_grad;   // :455 (r = grad: same sum as ||grad||^2)
        const double nr0_theta = (THETA_ONE || p.theta == 1.0) ? norm_grad : pow(norm_grad, p.theta);
        const double target = norm_grad * fmin(nr0_theta, p.kappa);  // rhs of :572
        const double target2 = target * target;
        // The residual test (:572) is taken on the PREDICTED <r', r'>.  Where the prediction comes within 1e-9 of the
        // threshold the decision would hang on the prediction's own round-off, so the hot path only continues on a
        // clear miss (> target2_hi); anything nearer goes through the cold block, which re-sums <r', r'> as the
        // reference does (:560-572) when the value lies inside the band -- a second reduction on a few steps in a
        // thousand, no instruction in the hot path.  (Until round 5 this was the one tolerated early flip.)
        const double target2_hi = target2 * (1.0 + 1e-9), target2_lo = target2 * (1.0 - 1e-9);
        // radius the plain path tests against: Delta2 / 16 until the checkpoint is taken
        const double Tq = 0.0625 * Delta2;
        double T_cur = RETRACE ? Tq : Delta2;
        ck_set = !RETRACE;
        // w = -(horizontal part of r).  The reference never projects the gradient
        // (fixed_rank_psd_sym.py:123-124), so r carries the vertical round-off of egrad for the
        // whole solve (every Hdelta is horizontal) and delta = -r + beta delta accumulates it.  Its
        // <delta, Hdelta> (:500) does not see that part -- Hdelta is projected -- but <delta, H>
        // with the raw H would, and late in a solve, where |r| falls to the size of that round-off,
        // the difference keeps tCG from ever leaving through the model test (measured on UR10:
        // one solve in five ran a 10000-iteration tCG the reference ends after ~150).  So delta
        // is built from w, which follows the same recurrence as -r from a projected start, and the
        // identities above hold to rounding; r itself, <r, r>, alpha and beta stay the reference's.
        double w = fma(rho0[2], cx.Q[2], fma(rho0[1], cx.Q[1], fma(rho0[0], cx.Q[0], -g)));
        double delta = w;                          // :469 (horizontal part)
        double e_Pd2 = 0.0, d_Pd = r0_r0;          // :464-471 (precon = identity); <eta,eta> below
        double model_prev = __builtin_inf();       // model value before the last step (:485: 0)
        // eta, Heta and <eta, eta> live in two register sets, A and B.  A step reads the current
        // set and writes the new values into the other one -- which until then holds the PREVIOUS
        // eta / Heta, exactly what the deferred model test rolls back to -- and the loop body is two
        // steps with the roles swapped.  Written with one set and `prev = cur; cur = new`, every
        // iteration pays seven v_mov_b64 register rotations (~10 cycles each for a lone wavefront;
        // `#pragma unroll` is ignored on this loop).
        double ea = 0.0, ha = 0.0, pa = 0.0;       // :444-445, :464
        double eb = 0.0, hb = 0.0, pb = 0.0;
        double e_Pe_end = 0.0;                     // <eta, eta> of the last step that passed the radius test
        // Drain every outstanding memory operation before the loop: a pending FLAT access (the
        // trace stores of the outer iteration) may return out of order with LDS, and as long as
        // the compiler has to assume one at the loop header it turns the first staged
        // s_waitcnt lgkmcnt(n) of every Hessian product into lgkmcnt(0).
        __builtin_amdgcn_s_waitcnt(0);
        // One tCG iteration (:495-597).  (ec, hc, pc): current eta, Heta, <eta,eta>; (en, hn): the
        // previous eta, Heta on entry, the new ones on a plain return; pn: the new <eta,eta>.
        // Returns true when tCG ends (result in eta_l / Heta_l, or `bad`).
        auto step = [&](const double ec, const double hc, const double pc, double &en, double &hn,
                        double &pn) __attribute__((always_inline)) -> bool {
          double H;                                // :497
          if constexpr (SPLIT) H = cx.ehess_end();     // (its gathers were issued at the end of the last step)
          else H = cx.ehess(delta);
          double v[8] = {cx.Q[0] * H, cx.Q[1] * H, cx.Q[2] * H,      delta * H,
                         w * H,       H * H,       ec * fma(0.5, hc, g), r * r};
          cx.template sum_n<8>(v);
          const double Hdelta = fma(-cx.Q[2], v[2], fma(-cx.Q[1], v[1], fma(-cx.Q[0], v[0], H)));
          const double d_Hd = v[3];                // :500
          const double Hd_Hd = fma(-v[2], v[2], fma(-v[1], v[1], fma(-v[0], v[0], v[5])));
          const double model_value = v[6];         // :551 evaluated at the current eta
          const double r_r = v[7];                 // :564 exact
          const double rho = frcp1(d_Hd);
          const double alpha = r_r * rho;          // :503
#ifdef GIK_TCGDUMP
          if ((dbg & 4) && b == 0 && dbg_buf && kiter == 0) {
            const int slot = j < 256 ? j : 256 + (j >> 5);
            const double dq = cx.sum1(delta * cx.Q[0]), rq = cx.sum1(r * cx.Q[0]);
            if (slot < 1024 && lead) {
              double *q = dbg_buf + (size_t)slot * 8;
              q[0] = r_r; q[1] = d_Hd; q[2] = alpha; q[3] = model_value; q[4] = v[4]; q[5] = Hd_Hd;
              q[6] = dq; q[7] = rq;
            }
          }
#endif
          const double e_Pe_new = fma(alpha, fma(alpha, d_Pd, e_Pd2), pc);             // :506
          // <r',r'>/<r,r> = 1 + (2 <r,Hdelta> + alpha |Hdelta|^2) / <delta,Hdelta>   (alpha/<r,r> = rho)
          const double beta_p = fma(fma(alpha, Hd_Hd, -(v[4] + v[4])), rho, 1.0);      // :592 predicted
          double new_r_r = beta_p * r_r;                                               // :564 predicted
          // & rather than &&, and the cold block marked unlikely: five compares and scalar ands
          // with the hot path laid out as a fall-through chain.  Short-circuit evaluation put a
          // taken branch into every iteration, the default block placement two more (measured
          // 1072 -> 1048 -> 1004 cycles per iteration)
          // (the last permitted iteration, :495, is routed through the cold block as well, so
          // that the loop has a single exit edge)
          const bool plain = (model_value < model_prev) & (d_Hd > 0.0) & (e_Pe_new < T_cur) &
                             (beta_p >= 1e-3) & !((j >= p.mininner) & (new_r_r <= target2_hi)) &
                             (j + 1 < p.maxinner);
          double beta = beta_p;
          double rr_test = new_r_r;      // what the residual test sees (the recurrences keep the prediction)
          if (__builtin_expect(UNI(!plain), 0)) {   // any exit, a NaN, or the accuracy guard
            if (!(d_Hd == d_Hd) || !(new_r_r == new_r_r) || !(model_value == model_value)) {
              bad = true;
              return true;
            }
            if (model_value >= model_prev) {                    // :552 of step j-1
              eta_l = en;
              Heta_l = hn;
              e_Pe_end = pc;
              stop_tCG = TCG_MODEL_INCREASED;
              extra = 1;
              j = j - 1;
              return true;
            }
            if constexpr (RETRACE) {
              if (!ck_set && (d_Hd <= 0.0 || e_Pe_new >= Tq)) {   // first meeting with radius / 4
                cx.ck_put(0, ec);
                cx.ck_put(1, hc);
                cx.ck_put(2, delta);
                cx.ck_put(3, Hdelta);
                ck_e_Pe = pc;
                ck_e_Pd2 = e_Pd2;
                ck_d_Pd = d_Pd;
                ck_j = j;
                ck_neg = d_Hd <= 0.0;
                ck_T = Tq;
                ck_set = true;
                T_cur = Delta2;
              }
            }
            if (d_Hd <= 0.0 || e_Pe_new >= Delta2) {           // :509
              const double tau = boundary_tau(e_Pd2, d_Pd, Delta2, pc);             // :514
              eta_l = fma(tau, delta, ec);                      // :516
              Heta_l = fma(tau, Hdelta, hc);                    // :521
              e_Pe_end = pc;
              stop_tCG = (d_Hd <= 0.0) ? TCG_NEGATIVE_CURVATURE : TCG_EXCEEDED_TR;  // :531-534
              return true;
            }
            // the step's new vectors, for this block only: formed from an opaque copy of alpha so
            // that they are not merged with (and hoisted above the branch for) the hot path's
            double alpha_c = alpha;
            asm volatile("" : "+v"(alpha_c));
            if (beta_p < 1e-3) {
              const double new_r = fma(alpha_c, Hdelta, r);     // :561
              new_r_r = cx.sum1(new_r * new_r);
              beta = new_r_r / r_r;
              rr_test = new_r_r;
            } else if (j >= p.mininner && new_r_r >= target2_lo && new_r_r <= target2_hi) {
              const double new_r = fma(alpha_c, Hdelta, r);     // :561: a near tie is decided on the sum itself (:564)
              rr_test = cx.sum1(new_r * new_r);
            }
            if (j >= p.mininner && rr_test <= target2) {        // :572
              e_Pe_end = e_Pe_new;   // this step passed the radius test (what a rerun has to pass again)
              // the reference tests the model of this step first (:552)
              const double new_eta = fma(alpha_c, delta, ec);   // :538
              const double new_Heta = fma(alpha_c, Hdelta, hc); // :542
              const double model_new = cx.sum1(new_eta * fma(0.5, new_Heta, g));
              if (model_new >= model_value) {
                eta_l = ec;
                Heta_l = hc;
                stop_tCG = TCG_MODEL_INCREASED;
              } else {
                eta_l = new_eta;
                Heta_l = new_Heta;
                stop_tCG = (p.kappa < nr0_theta) ? TCG_REACHED_TARGET_LINEAR
                                                 : TCG_REACHED_TARGET_SUPERLINEAR;
              }
              return true;
            }
            if (j + 1 >= p.maxinner) {
              // inner iterations exhausted (:495) with the model test of this last step pending:
              // take the step and reduce its model value on the spot
              e_Pe_end = e_Pe_new;
              const double new_eta = fma(alpha_c, delta, ec);   // :538
              const double new_Heta = fma(alpha_c, Hdelta, hc); // :542
              const double model_last = cx.sum1(new_eta * fma(0.5, new_Heta, g));
              if (model_last >= model_value) {
                eta_l = ec;
                Heta_l = hc;
                stop_tCG = TCG_MODEL_INCREASED;
              } else {
                eta_l = new_eta;
                Heta_l = new_Heta;
              }
              j = p.maxinner;
              return true;
            }
          }
          ++j;
          pn = e_Pe_new;                                    // :537
          model_prev = model_value;
          if constexpr (SPLIT) {
            // the next direction first, published and its gathers issued at once: the LDS round trip then runs behind
            // the rest of this step's updates instead of in front of the next product (same values, another order)
            en = fma(alpha, delta, ec);                     // :538 (reads the old delta)
            w = fma(-alpha, Hdelta, w);
            delta = fma(beta, delta, w);                    // :593
            cx.ehess_begin(delta);
            __builtin_amdgcn_sched_barrier(0);
            hn = fma(alpha, Hdelta, hc);                    // :542
            r = fma(alpha, Hdelta, r);                      // :561
            // (forming the next step's two H-free summands here as well was tried: four register moves more between the
            //  role-swapped steps, and the gain of the split gone -- NOTEBOOK 11.9)
          } else {
            en = fma(alpha, delta, ec);                     // :538, :556-558 (over the previous eta)
            hn = fma(alpha, Hdelta, hc);                    // :542
            r = fma(alpha, Hdelta, r);                      // :561
            w = fma(-alpha, Hdelta, w);
            delta = fma(beta, delta, w);                    // :593
          }
          e_Pd2 = beta * fma(alpha + alpha, d_Pd, e_Pd2);   // :596 (carried as 2 <eta, delta>)
          d_Pd = fma(beta * beta, d_Pd, new_r_r);           // :597
          return false;
        };
        j = 0;
        if constexpr (SPLIT) {
          if (p.maxinner > 0) cx.ehess_begin(delta);
        }
        if (p.maxinner > 0)
          for (;;) {                               // :495
            if (step(ea, ha, pa, eb, hb, pb)) break;
            if (step(eb, hb, pb, ea, ha, pa)) break;
          }
        const double e_Pe = e_Pe_end;
        last_e_Pe = e_Pe;
        eta = eta_l;
        Heta = Heta_l;
        executed = (j >= p.maxinner ? p.maxinner : j + 1) + extra;
      } else {
        eta = 0.0;                                 // :444-445
        Heta = 0.0;
        stop_tCG = TCG_MAX_INNER_ITER;             // :491
        double r = g;                              // :448
        double e_Pe = 0.0;
        double r_r = cx.sum1(r * r);              // :455
        const double norm_r0 = sqrt(r_r);
        const double nr0_theta = (THETA_ONE || p.theta == 1.0) ? norm_r0 : pow(norm_r0, p.theta);
        const double target = norm_r0 * fmin(nr0_theta, p.kappa);  // rhs of :572
        const double target2 = target * target;
        double z_r = r_r, d_Pd = r_r;              // :464-466 (precon = identity)
        double inv_z_r = frcp(z_r);
        double delta = -r;                         // :469
        double e_Pd = 0.0, model_value = 0.0;      // :471,485
        double rho_pk[Ctx::NC], s_pk[Ctx::NC], hd_pk[Ctx::NC];  // <r,pk2>, <delta,pk2>, <Hdelta,pk2>
#pragma unroll
        for (int m = 0; m < Ctx::NC; ++m) {
          rho_pk[m] = rho0[m];
          s_pk[m] = -rho0[m];
        }
        for (j = 0; j < p.maxinner; ++j) {         // :495
          double d_Hd;
          const double Hdelta = cx.hess_proj_dot(delta, s_pk, d_Hd, hd_pk);  // :497-500
          if (UNI(!(d_Hd == d_Hd))) { bad = true; break; }
          const double alpha = z_r * frcp(d_Hd);            // :503
          const double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd;  // :506
          if (UNI(d_Hd <= 0.0 || e_Pe_new >= Delta * Delta)) {   // :509
            const double tau =
                (-e_Pd + sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd;  // :514
            eta = eta + tau * delta;                        // :516
            Heta = Heta + tau * Hdelta;                     // :521
            stop_tCG = (d_Hd <= 0.0) ? TCG_NEGATIVE_CURVATURE : TCG_EXCEEDED_TR;  // :531-534
            break;
          }
          if ((dbg & 4) && b == 0 && lead && dbg_buf && kiter < 64 && j < 128) {
            double *q = dbg_buf + ((size_t)kiter * 128 + j) * 4;
            q[0] = r_r; q[1] = d_Hd; q[2] = alpha; q[3] = model_value;
          }
          e_Pe = e_Pe_new;                                  // :537
          const double new_eta = eta + alpha * delta;       // :538
          const double new_Heta = Heta + alpha * Hdelta;    // :542
          const double new_r = r + alpha * Hdelta;          // :561 (speculative; same value)
          double m[3] = {new_eta * g, new_eta * new_Heta, new_r * new_r};
          cx.template sum_n<3>(m);
          const double new_model_value = m[0] + 0.5 * m[1]; // :551
          if (UNI(new_model_value >= model_value)) {        // :552
            stop_tCG = TCG_MODEL_INCREASED;
            break;
          }
          eta = new_eta;                                    // :556-558
          Heta = new_Heta;
          model_value = new_model_value;
          r = new_r;                                        // :561
          r_r = m[2];                                       // :564
          // :572  norm_r <= norm_r0*min(norm_r0^theta, kappa), compared on the squares
          if (UNI(j >= p.mininner && r_r <= target2)) {
            stop_tCG = (p.kappa < nr0_theta) ? TCG_REACHED_TARGET_LINEAR
                                             : TCG_REACHED_TARGET_SUPERLINEAR;
            break;
          }
          z_r = r_r;                                        // :589
          const double beta = z_r * inv_z_r;                // :592 (1/z_r_old, formed off the critical path)
          inv_z_r = frcp(z_r);
          delta = -r + beta * delta;                        // :593
          if constexpr (K == 2) {                           // the same two updates seen through pk2
#pragma unroll
            for (int m = 0; m < Ctx::NC; ++m) {
              rho_pk[m] = fma(alpha, hd_pk[m], rho_pk[m]);
              s_pk[m] = fma(beta, s_pk[m], -rho_pk[m]);
            }
          }
          e_Pd = beta * (e_Pd + alpha * d_Pd);              // :596
          d_Pd = z_r + beta * beta * d_Pd;                  // :597
        }
        executed = j >= p.maxinner ? p.maxinner : j + 1;
      }
      if (prof) prof_tcg += (long long)__builtin_readcyclecounter() - prof_t1;
      if (bad) break;
      if (j >= p.maxinner) j = p.maxinner - 1;  // Python leaves j at the last index
      inner_total += j + 1;
      inner_exec += executed;

      // -------------- outer iteration (trust_region.py:248-422) ---------------------------
      if (has_trace && kiter < trace.cap && lead) {
        const size_t q = (size_t)b * trace.cap + kiter;
        trace.d_Delta[q] = Delta;
        trace.d_numit[q] = j;
        trace.d_stop[q] = stop_tCG;
        trace.d_f_before[q] = fx;
      }
      const double x_prop = x + eta;                     // :248 retr
      const double fx_prop = cx.cost(x_prop);            // :251
      double rhonum = fx - fx_prop;                      // :255
      double gd[2] = {g * eta, eta * Heta};
      cx.template sum_n<2>(gd);
      double rhoden = -gd[0] - 0.5 * gd[1];              // :256
      const double rho_reg =
          fmax(1.0, fabs(fx)) * 2.220446049250313e-16 * p.rho_regularization;  // :287
      rhonum += rho_reg;                                 // :288
      rhoden += rho_reg;                                 // :289
      const bool model_decreased = rhoden >= 0.0;        // :311
      const double rho = rhonum / rhoden;                // :317
      if (rho < 0.25 || !model_decreased || !(rho == rho)) {  // :336
        Delta = Delta / 4.0;                             // :338
      } else if (rho > 0.75 &&
                 (stop_tCG == TCG_NEGATIVE_CURVATURE || stop_tCG == TCG_EXCEEDED_TR)) {
        Delta = fmin(2.0 * Delta, Delta_bar);            // :357-361
      }
      int accept = 0;
      prev_rejected = true;
      if (UNI(model_decreased && rho > p.rho_prime)) {   // :382
        accept = 1;
        prev_rejected = false;
        ++n_accept;
        x = x_prop;                                      // :385
        fx = fx_prop;                                    // :386
        g = cx.commit();                                 // :387 (rows of x_prop are in LDS)
        cx.proj_setup(p.planar_proj_exact);
        norm_grad = grad_norm_and_rho(cx, g, rho0);      // :388
      }
      if (has_trace && kiter < trace.cap && lead) {
        const size_t q = (size_t)b * trace.cap + kiter;
        trace.d_gradnorm_after[q] = norm_grad;
        trace.d_accept[q] = accept;
      }
      kiter = kiter + 1;                                 // :394
      // :414-416 stopping criterion (pymanopt 0.2.5 order: maxiter before gradnorm; the
      // wall-clock maxtime test is not reproduced -- it is non-deterministic)
      if (kiter >= p.maxiter) { stop = 1; break; }
      if (UNI(norm_grad < p.mingradnorm)) { stop = 0; break; }
      if (UNI(!(norm_grad == norm_grad) || !(fx == fx))) { bad = true; break; }
      if constexpr (MIG) {
        ++slice_count;
        if ((kiter & 3) == 0 && mig_poll(cx, *mig)) { paused = PAUSE_DONATE; break; }
        // a slice is at least `slice_its` outer iterations AND mig->slice_cycles long (a hand-over
        // costs the ~10 us of saving / reloading the point and re-evaluating cost and gradient:
        // the lower bound in time keeps that negligible for problems with very cheap iterations)
        if (slice_its > 0 && slice_count >= slice_its &&
            (long long)__builtin_readcyclecounter() - slice_t0 > (long long)mig->slice_cycles) {
          if (mig_anyone_waiting(cx, *mig)) { paused = PAUSE_YIELD; break; }
          slice_count = 0;        // nobody waits: keep the slot for another slice
          slice_t0 = (long long)__builtin_readcyclecounter();
        }
      } else if constexpr (SLICE) {
        if (slice_its > 0 && ++slice_count >= slice_its) { paused = PAUSE_YIELD; break; }
      }
    }
    if (bad) stop = 2;
    if constexpr (Ctx::AGE_PRIORITY) __builtin_amdgcn_s_setprio(0);
    if (prof && lead) {
      dbg_buf[0] = (double)prof_tcg;
      dbg_buf[1] = (double)inner_total;
      dbg_buf[2] = (double)((long long)__builtin_readcyclecounter() - prof_t0);
      dbg_buf[3] = (double)inner_exec;
    }

    out.f = fx;
    out.gradnorm = norm_grad;
    out.iterations = kiter;
    out.inner_total = inner_total;
    out.inner_executed = inner_exec;
    out.stop = stop;
    out.n_accept = n_accept;
    out.Delta = Delta;
    out.paused = paused;
}

}  // namespace gik
