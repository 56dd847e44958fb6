// graphik_amd/csrc/gik_kernels.hip.h -- every __global__ entry of the library (gfx950), as templates / prototypes.
//
// The kernels are INSTANTIATED in the gik_k_*.hip translation units (explicit instantiations, one group of
// kernels per file so that the groups compile in parallel and a change to one context recompiles one file);
// gik_host.hip -- templates, scheduling slots, the C ABI -- sees them through `extern template` declarations
// (gik_instances.h) and compiles no device code of its own.
//
//   rtr_wave_kernel : whole Riemannian trust-region solve (TrustRegions.solve +
//                     _truncated_conjugate_gradient, graphik/solvers/trust_region.py:112-599)
//                     of one IK problem per wavefront, one launch per batch.
//   kat_wave_kernel : the same device functions exposed one call at a time, batched
//                     (costgrd twins + PSDFixedRank.proj) for known-answer parity tests.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <string>
#include <vector>

#include "gik_block.hip.h"
#include "gik_npt.hip.h"
#include "gik_quad.hip.h"
#include "gik_prep.hip.h"
#include "gik_prep_quad.hip.h"
#include "gik_rcg.hip.h"
#include "gik_rtr.hip.h"
#include "gik_rtrv.hip.h"
#include "gik_wave.hip.h"
#ifdef GIK_STRICT_HEADER      // developer experiments (tools/exp/strict_bisect.sh): another rendering of WaveCtxStrict
#include GIK_STRICT_HEADER
#else
#include "gik_wave_strict.hip.h"
#endif
#include "graphik_amd.h"

namespace gik {

// ------------------------------------------------------------------------------------------
struct SliceState {
  double Delta;
  int kiter, inner_total, inner_exec, n_accept;
  int resumes, pad;    // times the problem changed hands (reported in gik_stats.flags >> 8)
};

// Claim the next piece of work for this wave / workgroup (called by one thread).  Returns the
// problem index, or -1 when every problem of the launch has finished.
__device__ inline int claim_work(unsigned int *ticket_counter, const unsigned int *q_seq, const int *q_ids,
                                 const unsigned int *q_done, int B, int slicing, int &resumed) {
  const unsigned int ticket = atomicAdd(ticket_counter, 1u);
  resumed = 0;
  if (ticket < (unsigned)B) return (int)ticket;
  if (!slicing) return -1;
  const unsigned int t = ticket - (unsigned)B;
  for (;;) {
    if (__hip_atomic_load(&q_seq[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == ticket) break;
    if (__hip_atomic_load(q_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)B) return -1;
    __builtin_amdgcn_s_sleep(32);
  }
  resumed = 1;
  return __hip_atomic_load(&q_ids[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (cache-bypassing loads: the state was written by another CU; the acquire in claim_work was one
// thread's)
// The values are the same in every lane; readfirstlane moves them to scalar registers, where the
// solver's counters live (as vector loads they would occupy VGPRs for the whole solve -- enough
// to push the 9-slot kernel over 256 registers and into scratch).
__device__ inline int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline double uniform_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                          __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ inline RtrResume load_slice_state(const SliceState *st) {
  RtrResume rs;
  rs.Delta = uniform_f64(__builtin_nontemporal_load(&st->Delta));
  rs.kiter = uniform_i32(__builtin_nontemporal_load(&st->kiter));
  rs.inner_total = uniform_i32(__builtin_nontemporal_load(&st->inner_total));
  rs.inner_exec = uniform_i32(__builtin_nontemporal_load(&st->inner_exec));
  rs.n_accept = uniform_i32(__builtin_nontemporal_load(&st->n_accept));
  rs.resumes = uniform_i32(__builtin_nontemporal_load(&st->resumes));
  rs.resumed = 1;
  return rs;
}

// publish a paused problem (one thread; the state stores of all threads must be complete and
// fenced before the call)
__device__ inline void requeue_work(unsigned int *q_tail, int *q_ids, unsigned int *q_seq, int B, int b) {
  const unsigned int t = atomicAdd(q_tail, 1u);
  __hip_atomic_store(&q_ids[t], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&q_seq[t], (unsigned)B + t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

struct AnchArgs {
  const double *anch_const;     // [ANCH_MAXA][4] pinned anchor table (goal rows are overwritten per problem)
  const double *anchor_goal;    // [B][n_goal * 3] per-problem anchor positions
  const uint32_t *pin_meta;     // [ANCH_PMAX][64] anchor row | kind << 8
  const double *pin_tgt;        // [ANCH_PMAX][64]
  const double *obs;            // [n_obs][4] x, y, z, r^2
  unsigned long long obs_mask;  // bit i: free node i carries the obstacle hinges
  int n_obs, n_goal, goal_row0; // goal anchors occupy rows goal_row0 .. goal_row0 + n_goal - 1
};

struct SolveArgs {
  const uint32_t *slot_meta;  // [MAXDEG][64]
  const double *targets;      // [B][T]
  const double *Y_init;       // [B][N*K]
  double *Y_out;              // [B][N*K]
  gik_stats *stats;           // [B]
  unsigned int *work_counter; // zeroed before launch; problems are claimed with atomicAdd
  gik_trace trace;
  int has_trace;
  int N, T, B;
  int dbg;  // debug flags (env GIK_DBG): 1 = one block per problem, 2 = skip the TR loop,
            // 4 = dump (r_r, d_Hd, alpha, model) of every inner iteration of problem 0 to dbg_buf,
            // 8 = cycle counters of problem 0: dbg_buf = {cycles in tCG loops, tCG iterations, all cycles},
            // 16 = rerun tCG after every rejected step instead of resuming from the checkpoint;
            // at template creation: 32 = print the kernel choice, 64 / 128 = clique closed form of the
            // workgroup path from 4 nodes up / off
  double *dbg_buf;
  Params p;
  CgParams cg;   // solver == GIK_SOLVER_CONJUGATE_GRADIENT (rcg_* kernels)
  // fixed-anchor formulation (anchored templates; see WaveCtx<.., ANCH>)
  AnchArgs an;
  // Time slicing (slice_its > 0): a problem that has not met a stopping rule after slice_its outer
  // iterations is written back (x in Y_out, SliceState) and re-queued behind everything that is
  // waiting, so that all problems advance at about the same rate and the long ones -- unknown in
  // advance -- are not the last to START.  Tickets < B are the fresh problems themselves;
  // ticket B + t is the t-th re-queued problem, published in q_ids[t] / q_seq[t] (no slot is ever
  // reused: the ring has room for every possible re-queue of the launch).
  int slice_its;
  int slice_cycles;   // wavefront kernel: shortest round-robin slice (MigCtl::slice_cycles)
  // tail spreading (wavefront kernel, MIG variant): q_head = hand-over tickets taken by helpers,
  // mig_credits / mig_simd_run as in MigCtl; q_tail / q_seq / q_ids / q_state / q_done as for slicing
  unsigned int *q_head;
  int *mig_credits, *mig_simd_run;
  // round-robin slicing of the wavefront kernel: yield queue (y_seq[k] == k + 1 once entry k is published)
  // Entries are taken by fetch-add tickets on y_head, never by compare-and-swap (2048 waves that
  // retry a CAS on one word serve ~50 k claims per second -- measured, the whole batch then waits for
  // its queue).  That needs a guarantee that a ticket's entry exists: a wave that yields pushes
  // first, so it owns one entry's worth of claim (it either pops at once, or, if it got a fresh
  // problem instead, passes the claim on by y_avail += 1); a wave that comes from a finished
  // problem has to win one from y_avail (fetch-add -1, undone if it went negative).
  unsigned int *y_head, *y_tail, *y_seq;
  int *y_ids, *y_avail;
  unsigned int y_cap;
  unsigned int *q_tail, *q_done;   // next to work_counter (= the ticket counter)
  int *q_ids;                      // [cap]
  unsigned int *q_seq;             // [cap], 0xffffffff = not published
  SliceState *q_state;             // [B]
  BlockTabs bt;                    // workgroup-per-problem path
  NptTabs nt;                      // node-per-lane path (rtr_npt_kernel)
  double *npt_ctg_ws = nullptr;    // graphs beyond 128 nodes: [grid][ctg_doubles] clique target triangles (global memory)
};

// This wave's physical SIMD: XCC_ID[3:0] and the SIMD / CU / SH / SE fields of HW_ID (bits 4-5 and
// 8-15; the wave slot, pipe and queue fields in between and above say nothing about the place).
__device__ inline int hw_simd_id() {
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11));    // HW_REG_HW_ID[15:0]
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID[3:0]
  return (int)(((hw >> 4) & 3u) | (((hw >> 8) & 0xffu) << 2) | (xcc << 10));
}

// Helper side of the tail spreading (MigCtl, gik_rtr.hip.h), one thread: wait until every problem is
// done (-> -1) or a paused problem is handed to this wave (-> its index).  The wave first has to
// find its SIMD empty and reserve it; while the SIMD is busy with the partner wave's problem it
// polls slowly.
// (-2: the yield queue of the round-robin slicing has an entry again -- the caller goes back to it; a
// helper only commits to a hand-over ticket while that queue is empty.)
__device__ inline int mig_wait(const MigCtl &m, unsigned int *q_head, const unsigned int *q_seq, const int *q_ids,
                               const unsigned int *q_done, int B) {
  for (;;) {
    if (__hip_atomic_load(q_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)B) return -1;
    if (__hip_atomic_load(m.y_avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) return -2;
    int expect = 0;
    // (commit to a hand-over only in the tail: while there are more unfinished problems than waves the
    // yield queue refills at once)
    if ((unsigned)B <= __hip_atomic_load(q_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (unsigned)m.waves &&
        __hip_atomic_load(&m.simd_run[m.sid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 &&
        __hip_atomic_compare_exchange_strong(&m.simd_run[m.sid], &expect, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT)) {
      const unsigned int t = __hip_atomic_fetch_add(q_head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(m.credits, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        if (__hip_atomic_load(&q_seq[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)B + t)
          return __hip_atomic_load(&q_ids[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_load(q_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)B) return -1;
        __builtin_amdgcn_s_sleep(64);
      }
    }
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_s_sleep(127);   // ~7 us
  }
}

#ifdef GIK_DEV
// event log of the scheduling probes: dbg_buf[8] = entries, entry e at 16 + 4 e = {time in 10 ns, wave, type, problem}
__device__ inline void dev_log(double *buf, int type, int b) {
  const int e = (int)atomicAdd(&buf[8], 1.0);
  if (e >= 250000) return;
  double *p = buf + 16 + 4 * (size_t)e;
  p[0] = (double)__builtin_amdgcn_s_memrealtime();
  p[1] = (double)blockIdx.x;
  p[2] = (double)type;
  p[3] = (double)b;
}
#endif

// Stage the launch-invariant slot table into LDS and zero the gather tiles (idle lanes and
// padding slots read the never-written dump row, which must hold finite zeros).
template <typename Ctx>
__device__ inline void stage_lds(double *tiles, uint32_t *meta, const uint32_t *g_meta, int lane,
                                 int maxdeg, int N) {
  for (int t = lane; t < Ctx::NTILE * Ctx::TILE; t += WAVE) tiles[t] = 0.0;
  if constexpr (Ctx::SLIM_LAYOUT) {
    // slim layout (per-edge context): table row k of a lane is node slot comp + 3 k, the slots the lane OWNS; beyond the
    // compiled slot count: the padding word of the host's table (own row -- idle lanes: the dump row --, kind none)
    const bool active = lane < N * 3;
    const int comp = active ? lane % 3 : 0;
    const uint32_t pad = meta_pack(active ? lane / 3 : TILE_ROWS - 1, 0, 0, 0);
    for (int k = 0; k < Ctx::NSL; ++k) {
      const int s = comp + 3 * k;
      meta[k * WAVE + lane] = s < maxdeg ? g_meta[s * WAVE + lane] : pad;
    }
  } else {
    for (int s = 0; s < maxdeg; ++s) meta[s * WAVE + lane] = g_meta[s * WAVE + lane];
  }
  __builtin_amdgcn_wave_barrier();
}

// Persistent kernel: grid = (resident waves), each wavefront claims IK problems from a global
// counter until the batch is exhausted.  Iteration counts differ by >50x between goals and the
// hardware hands workgroups to XCDs round-robin, so a static block->problem map leaves whole
// XCDs idle behind a few stragglers; the queue keeps every SIMD busy until the end.
// (the fixed-anchor variant holds 34 KB of LDS per wave: four waves per CU, one per SIMD, so it may
// as well have that SIMD's whole register file -- at two waves per SIMD it spilled into the hot loop)
// STRICT: the Hessian product term by term as costs.py:186-203 forms it (gik_wave_strict.hip.h;
// gik_template_desc.hessian_form = GIK_HESS_PER_EDGE), k = 3 free-free graphs
template <int K, int MAXDEG, bool THETA_ONE, bool ANCH = false, bool MIG = false, bool STRICT = false>
__global__ void __launch_bounds__(WAVE, ANCH ? 1 : 2) rtr_wave_kernel(SolveArgs a) {
  static_assert(!ANCH || K == 3, "the fixed-anchor formulation is 3-D");
  static_assert(!MIG || !ANCH, "tail spreading: two waves per SIMD, i.e. not the anchored variant");
  static_assert(!STRICT || (K == 3 && !ANCH), "the per-edge product form: 3-D free-free graphs");
  using Ctx = std::conditional_t<STRICT, WaveCtxStrict<MAXDEG>, WaveCtx<K, MAXDEG, ANCH>>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int NK = a.N * K;
  double *sh_tiles = smem;
  double *sh_tgt = smem + Ctx::NTILE * Ctx::TILE;
  uint32_t *sh_meta = reinterpret_cast<uint32_t *>(sh_tgt + ((a.T + 1) & ~1));
  stage_lds<Ctx>(sh_tiles, sh_meta, a.slot_meta, lane, MAXDEG, a.N);

  Ctx cx;
  cx.init(lane, a.N, sh_tiles, sh_tgt, sh_meta);
  if constexpr (ANCH) {
    // every target is a template constant here: records, pinned records and the constant rows of
    // the anchor table are staged once per wave; only the goal anchors change per problem
    cx.init_anchored(a.an.obs_mask, a.an.obs, a.an.n_obs, !(a.dbg & 128));
    for (int t = lane; t < a.T; t += WAVE) sh_tgt[t] = a.targets[t];
    for (int t = lane; t < 4 * ANCH_MAXA; t += WAVE) cx.sh_anch[t] = a.an.anch_const[t];
    __builtin_amdgcn_wave_barrier();
    cx.load_slot_records();
    cx.load_pinned_records(a.an.pin_meta, a.an.pin_tgt);
  }
  const Params &p = a.p;
  int pass = 0;
  MigCtl mig = {a.mig_credits, a.mig_simd_run, 0, a.work_counter, a.y_avail, a.y_tail, a.B, a.y_cap, a.q_done,
                (int)gridDim.x, a.slice_cycles};
  bool tail = false;
  bool owns_entry = false;    // (lane 0) this wave has just pushed a yielded problem
  if constexpr (MIG) mig.sid = hw_simd_id();
#ifdef GIK_DEV
  long long dev_t_start = (long long)__builtin_readcyclecounter(), dev_t_solve = 0, dev_t_claim = 0;
  int dev_n_claim = 0;
#endif
  for (;;) {
    int b = 0, resumed = 0;
#ifdef GIK_DEV
    const long long dev_tc = (long long)__builtin_readcyclecounter();
#endif
    if constexpr (MIG) {
      // (the static block -> problem alternative is never selected together with MIG on the host; it
      // is what keeps the compiler's divergence analysis from treating this loop's exit as divergent
      // and wrapping the solver loops in exec masks -- see the note at rcg_wave_kernel)
      if (a.dbg & 1) {
        b = (int)blockIdx.x + pass * (int)gridDim.x;
        ++pass;
        if (b >= a.B) b = -1;
      } else {
        if (lane == 0) {
          b = -1;
          if (!tail) {
            const unsigned int t = atomicAdd(a.work_counter, 1u);
            if (t < (unsigned)a.B) {
              b = (int)t;
              __hip_atomic_fetch_add(&mig.simd_run[mig.sid], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          if (b >= 0 && owns_entry)      // this wave's yield stays in the queue: somebody else's to take
            __hip_atomic_fetch_add(a.y_avail, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          if (b < 0) {      // no fresh problem left: the oldest yielder, if any, else a hand-over
            resumed = 1;
            bool pop = owns_entry;
            while (!pop) {
              if (__hip_atomic_load(a.y_avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) {
                if (__hip_atomic_fetch_add(a.y_avail, -1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) > 0) {
                  pop = true;
                  break;
                }
                __hip_atomic_fetch_add(a.y_avail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
              b = mig_wait(mig, a.q_head, a.q_seq, a.q_ids, a.q_done, a.B);
              if (b != -2) break;       // a handed-over problem, or -1: everything is done
            }
            if (pop) {
              const unsigned int h = __hip_atomic_fetch_add(a.y_head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              while (__hip_atomic_load(&a.y_seq[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != h + 1u)
                __builtin_amdgcn_s_sleep(2);      // (its publisher is between the tail increment and this store)
              b = __hip_atomic_load(&a.y_ids[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_fetch_add(&mig.simd_run[mig.sid], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          owns_entry = false;
        }
        b = __builtin_amdgcn_readlane(b, 0);  // lane 0 explicitly, independent of exec
        resumed = __builtin_amdgcn_readlane(resumed, 0);
      }
      tail = tail || resumed;
#ifdef GIK_DEV
      dev_t_claim += (long long)__builtin_readcyclecounter() - dev_tc;
      ++dev_n_claim;
      if ((a.dbg & 4096) && a.dbg_buf && lane == 0) dev_log(a.dbg_buf, resumed ? 1 : 0, b);
#endif
      if (UNI(b < 0)) break;
    } else {
      if (a.dbg & 1) {
        b = (int)blockIdx.x + pass * (int)gridDim.x;
        ++pass;
      } else {
        if (lane == 0) b = (int)atomicAdd(a.work_counter, 1u);
        b = __builtin_amdgcn_readlane(b, 0);  // lane 0 explicitly, independent of exec
      }
      if (UNI(b >= a.B)) break;
    }

    if constexpr (ANCH) {
      if (lane < 3 * a.an.n_goal)
        cx.sh_anch[(a.an.goal_row0 + lane / 3) * 4 + lane % 3] =
            a.an.anchor_goal[(size_t)b * 3 * a.an.n_goal + lane];
      cx.obs_reset();
    } else {
      for (int t = lane; t < a.T; t += WAVE) sh_tgt[t] = a.targets[(size_t)b * a.T + t];
      __builtin_amdgcn_wave_barrier();
      cx.load_slot_records();
    }
    RtrOut ro;
    double x;
    int rs_resumes = 0;
    if constexpr (MIG) {
      // Branch-free on purpose: q_state is zeroed at launch, so a fresh problem reads zeros (a
      // conditional load here made the compiler treat the solver's counters as divergent and wrap
      // its loops in exec masks).  (State and point of a resumed problem were written by a wave on
      // another CU: cache-bypassing loads.)
      RtrResume rs = load_slice_state(&a.q_state[b]);
      rs.resumed = resumed;
      rs_resumes = rs.resumes;
      const double *src = resumed ? a.Y_out : a.Y_init;
      x = cx.active ? __builtin_nontemporal_load(&src[(size_t)b * NK + lane]) : 0.0;
#ifdef GIK_DEV
      const long long dev_ts = (long long)__builtin_readcyclecounter();
#endif
      rtr_solve_one<K, THETA_ONE, true, Ctx, true>(cx, p, a.trace, a.has_trace, a.dbg, a.dbg_buf, b, x, ro, rs,
                                                   a.slice_its, &mig);
#ifdef GIK_DEV
      dev_t_solve += (long long)__builtin_readcyclecounter() - dev_ts;
#endif
      if (cx.active) a.Y_out[(size_t)b * NK + lane] = x;
      if (UNI(ro.paused)) {
        if (lane == 0) {
          SliceState st;
          st.Delta = ro.Delta;
          st.kiter = ro.iterations;
          st.inner_total = ro.inner_total;
          st.inner_exec = ro.inner_executed;
          st.n_accept = ro.n_accept;
          st.resumes = rs.resumes + 1;
          st.pad = 0;
          a.q_state[b] = st;
        }
        __threadfence();
#ifdef GIK_DEV
        if ((a.dbg & 4096) && a.dbg_buf && lane == 0) dev_log(a.dbg_buf, ro.paused == PAUSE_DONATE ? 3 : 2, b);
#endif
        if (lane == 0) {
          if (ro.paused == PAUSE_DONATE) {
            requeue_work(a.q_tail, a.q_ids, a.q_seq, a.B, b);      // a helper holds the ticket for it
          } else {                                                   // yield: behind everything that waits
            const unsigned int k = atomicAdd(a.y_tail, 1u);
            __hip_atomic_store(&a.y_ids[k], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.y_seq[k], k + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            owns_entry = true;
          }
          __hip_atomic_fetch_add(&mig.simd_run[mig.sid], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (ro.paused == PAUSE_DONATE) tail = true;
        continue;
      }
    } else {
      x = cx.active ? a.Y_init[(size_t)b * NK + lane] : 0.0;
      const RtrResume rs = {0.0, 0, 0, 0, 0, 0, 0};
      rtr_solve_one<K, THETA_ONE, false>(cx, p, a.trace, a.has_trace, a.dbg, a.dbg_buf, b, x, ro, rs, 0);
      if (cx.active) a.Y_out[(size_t)b * NK + lane] = x;
    }
    if (lane == 0) {
      gik_stats s;
      s.f = ro.f;
      s.gradnorm = ro.gradnorm;
      s.iterations = ro.iterations;
      s.inner_total = ro.inner_total;
      s.stop = ro.stop;
      s.n_accept = ro.n_accept;
      s.inner_executed = ro.inner_executed;
      s.flags = (MIG && resumed) ? (2 | (rs_resumes << 8)) : 0;
      s.stepsize = ro.Delta;
      a.stats[b] = s;
#ifdef GIK_DEV
      if (MIG && (a.dbg & 4096) && a.dbg_buf) dev_log(a.dbg_buf, 4, b);
#endif
      if constexpr (MIG) {
        __hip_atomic_fetch_add(&mig.simd_run[mig.sid], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.q_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
#ifdef GIK_DEV
  if (MIG && (a.dbg & 4096) && a.dbg_buf && lane == 0) {
    atomicAdd(&a.dbg_buf[4], (double)dev_t_solve);
    atomicAdd(&a.dbg_buf[5], (double)dev_t_claim);
    atomicAdd(&a.dbg_buf[6], (double)((long long)__builtin_readcyclecounter() - dev_t_start));
    atomicAdd(&a.dbg_buf[7], (double)dev_n_claim);
  }
#endif
}

// Riemannian conjugate gradients (the reference's alternative solver), same persistent scheme
template <int K, int MAXDEG>
__global__ void __launch_bounds__(WAVE, 2) rcg_wave_kernel(SolveArgs a) {
  using Ctx = WaveCtx<K, MAXDEG>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int NK = a.N * K;
  double *sh_tiles = smem;
  double *sh_tgt = smem + Ctx::NTILE * Ctx::TILE;
  uint32_t *sh_meta = reinterpret_cast<uint32_t *>(sh_tgt + ((a.T + 1) & ~1));
  stage_lds<Ctx>(sh_tiles, sh_meta, a.slot_meta, lane, MAXDEG, a.N);
  Ctx cx;
  cx.init(lane, a.N, sh_tiles, sh_tgt, sh_meta);
  int pass = 0;
  for (;;) {
    // Same claim as rtr_wave_kernel, INCLUDING the static block -> problem alternative (dbg & 1).
    // With only the atomic claim in the loop the compiler treats the loop exit as divergent and
    // wraps this loop and the solver loop inside it in exec masks, under which the wave-wide
    // reductions dead-lock (measured: a hang at maxiter = 3; an "+s" asm pin on b does not help).
    // Checked in the ISA: no s_andn2_b64 exec besides the two strided copy loops.
    int b = 0;
    if (a.dbg & 1) {
      b = (int)blockIdx.x + pass * (int)gridDim.x;
      ++pass;
    } else {
      if (lane == 0) b = (int)atomicAdd(a.work_counter, 1u);
      b = __builtin_amdgcn_readlane(b, 0);  // lane 0 explicitly, independent of exec
    }
    if (UNI(b >= a.B)) break;
    for (int t = lane; t < a.T; t += WAVE) sh_tgt[t] = a.targets[(size_t)b * a.T + t];
    __builtin_amdgcn_wave_barrier();
    cx.load_slot_records();
    double x = cx.active ? a.Y_init[(size_t)b * NK + lane] : 0.0;
    RtrOut ro;
    rcg_solve_one<K>(cx, a.cg, a.trace, a.has_trace, b, x, ro);
    if (cx.active) a.Y_out[(size_t)b * NK + lane] = x;
    if (lane == 0) {
      gik_stats s;
      s.f = ro.f;
      s.gradnorm = ro.gradnorm;
      s.iterations = ro.iterations;
      s.inner_total = ro.inner_total;
      s.stop = ro.stop;
      s.n_accept = ro.n_accept;
      s.inner_executed = ro.inner_executed;
      s.flags = 0;
      s.stepsize = ro.Delta;
      a.stats[b] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------
struct KatArgs {
  const uint32_t *slot_meta;
  const double *targets;  // [B][T] (unused for proj)
  const double *Y;        // [B][N*K]
  const double *W;        // [B][N*K] (hess, proj)
  double *out;            // cost: [B]; others: [B][N*K]
  double *out_f;          // mode 4: cost [B] next to the gradient in `out`
  int N, T, B, mode;      // 0 cost, 1 grad, 2 hess, 3 proj, 4 cost and grad (one pass)
  int planar_proj_exact;
  AnchArgs an;
  BlockTabs bt;
  NptTabs nt;
  double *npt_ctg_ws = nullptr;   // see SolveArgs
};

template <int K, int MAXDEG, bool ANCH = false, bool STRICT = false>
__global__ void __launch_bounds__(WAVE) kat_wave_kernel(KatArgs a) {
  using Ctx = std::conditional_t<STRICT, WaveCtxStrict<MAXDEG>, WaveCtx<K, MAXDEG, ANCH>>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int NK = a.N * K;
  double *sh_tiles = smem;
  double *sh_tgt = smem + Ctx::NTILE * Ctx::TILE;
  uint32_t *sh_meta = reinterpret_cast<uint32_t *>(sh_tgt + ((a.T + 1) & ~1));
  stage_lds<Ctx>(sh_tiles, sh_meta, a.slot_meta, lane, MAXDEG, a.N);
  for (int t = lane; t < a.T; t += WAVE)
    sh_tgt[t] = a.targets ? a.targets[ANCH ? (size_t)t : (size_t)b * a.T + t] : 0.0;
  __builtin_amdgcn_wave_barrier();
  Ctx cx;
  cx.init(lane, a.N, sh_tiles, sh_tgt, sh_meta);
  if constexpr (ANCH) {
    cx.init_anchored(a.an.obs_mask, a.an.obs, a.an.n_obs, true);
    for (int t = lane; t < 4 * ANCH_MAXA; t += WAVE) cx.sh_anch[t] = a.an.anch_const[t];
    __builtin_amdgcn_wave_barrier();
    if (lane < 3 * a.an.n_goal)
      cx.sh_anch[(a.an.goal_row0 + lane / 3) * 4 + lane % 3] =
          a.an.anchor_goal[(size_t)b * 3 * a.an.n_goal + lane];
    __builtin_amdgcn_wave_barrier();
    cx.load_pinned_records(a.an.pin_meta, a.an.pin_tgt);
  }
  cx.load_slot_records();
  const double y = cx.active ? a.Y[(size_t)b * NK + lane] : 0.0;
  const double w = (a.W && cx.active) ? a.W[(size_t)b * NK + lane] : 0.0;
  if (a.mode == 0) {
    const double f = cx.cost(y);
    if (lane == 0) a.out[b] = f;
    return;
  }
  double res = 0.0;
  if (a.mode == 1) {
    cx.put(y);
    res = cx.commit();
  } else if (a.mode == 4) {
    const double f = cx.cost(y);   // leaves the rows of y in the tiles for commit()
    res = cx.commit();
    if (lane == 0) a.out_f[b] = f;
  } else if (a.mode == 2) {
    cx.put(y);
    (void)cx.commit();
    res = cx.ehess(w);
  } else {
    cx.put(y);
    cx.proj_setup(a.planar_proj_exact);
    res = cx.proj(w);
  }
  if (cx.active) a.out[(size_t)b * NK + lane] = res;
}

// ------------------------------------------------------------------------------------------
// workgroup-per-problem variants (graphs with N*k > 64)
template <int K>
__device__ inline uint32_t *block_stage(BlockCtx<K> &cx, double *smem, const uint32_t *g_slots, int N,
                                        const BlockTabs &bt, int SL) {
  const int tid = threadIdx.x;
  const int T = bt.Tc;
  for (int t = tid; t < 3 * BLOCK_MAXN * BlockCtx<K>::RS; t += BLOCK_NT) smem[t] = 0.0;
  double *tg = smem + 3 * BLOCK_MAXN * BlockCtx<K>::RS;
  uint32_t *slots =
      reinterpret_cast<uint32_t *>(tg + ((T + 1) & ~1) + 2 * 8 * BLOCK_WAVES + CLQ_NMOM * BLOCK_WAVES + CLQ_NCQ +
                                   32 * BLOCK_WAVES);
  for (int s = 0; s < SL; ++s) slots[s * BLOCK_NT + tid] = g_slots[s * BLOCK_NT + tid];
  cx.init(N, SL, bt, smem, slots, T);
  if constexpr (K == 3) {   // pair ids of every thread's clique partners (launch invariant)
    unsigned short *pid = const_cast<unsigned short *>(cx.sh_pid);
    for (int m = 0; m < cx.M_clq; ++m) pid[m * BLOCK_NT + tid] = bt.clq_pid_t[m * BLOCK_NT + tid];
  }
  __syncthreads();
  return slots;
}

template <int K>
__global__ void __launch_bounds__(BLOCK_NT) rtr_block_kernel(SolveArgs a, int SL) {
  // 16-byte alignment matters: the static `sh_b` below would otherwise push the dynamic segment
  // to offset 8, and every ds_read_b128 of a point row would be misaligned (measured 5x slower)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ int sh_b, sh_resumed;
  const int tid = threadIdx.x;
  const int NK = a.N * K;
  BlockCtx<K> cx;
  block_stage<K>(cx, smem, a.slot_meta, a.N, a.bt, SL);
  double *sh_tgt = smem + 3 * BLOCK_MAXN * BlockCtx<K>::RS;
  for (;;) {
    if (tid == 0) {
      int res = 0;
      sh_b = claim_work(a.work_counter, a.q_seq, a.q_ids, a.q_done, a.B, a.slice_its > 0, res);
      sh_resumed = res;
    }
    __syncthreads();
    const int b = sh_b, resumed = sh_resumed;
    __syncthreads();
    if (UNI(b < 0)) break;
    cx.load_problem(a.targets + (size_t)b * a.T, a.bt, sh_tgt);
    RtrResume rs = {0.0, 0, 0, 0, 0, 0, 0};
    double x = 0.0;
    if (resumed) {
      rs = load_slice_state(&a.q_state[b]);
      if (cx.active) x = __builtin_nontemporal_load(&a.Y_out[(size_t)b * NK + cx.gnode * K + cx.part]);
    } else if (cx.active) {
      x = a.Y_init[(size_t)b * NK + cx.gnode * K + cx.part];
    }
    RtrOut ro;
#ifdef GIK_BLK_PROF
    cx.prof = ((a.dbg & 8) && b == 0) ? a.dbg_buf : nullptr;
#endif
    rtr_solve_one<K, false, true>(cx, a.p, a.trace, a.has_trace, a.dbg, a.dbg_buf, b, x, ro, rs, a.slice_its);
    if (cx.active) a.Y_out[(size_t)b * NK + cx.gnode * K + cx.part] = x;
    if (UNI(ro.paused)) {
      if (tid == 0) {
        SliceState st;
        st.Delta = ro.Delta;
        st.kiter = ro.iterations;
        st.inner_total = ro.inner_total;
        st.inner_exec = ro.inner_executed;
        st.n_accept = ro.n_accept;
        st.resumes = rs.resumes + 1;
        st.pad = 0;
        a.q_state[b] = st;
      }
      __threadfence();
      __syncthreads();
      if (tid == 0) requeue_work(a.q_tail, a.q_ids, a.q_seq, a.B, b);
      continue;
    }
    if (tid == 0) {
      gik_stats s;
      s.f = ro.f;
      s.gradnorm = ro.gradnorm;
      s.iterations = ro.iterations;
      s.inner_total = ro.inner_total;
      s.stop = ro.stop;
      s.n_accept = ro.n_accept;
      s.inner_executed = ro.inner_executed;
      s.flags = cx.lowrank ? 1 : 0;
      s.stepsize = ro.Delta;
      a.stats[b] = s;
      if (a.slice_its > 0) __hip_atomic_fetch_add(a.q_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int K>
__global__ void __launch_bounds__(BLOCK_NT) rcg_block_kernel(SolveArgs a, int SL) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ int sh_b;
  const int tid = threadIdx.x;
  const int NK = a.N * K;
  BlockCtx<K> cx;
  block_stage<K>(cx, smem, a.slot_meta, a.N, a.bt, SL);
  double *sh_tgt = smem + 3 * BLOCK_MAXN * BlockCtx<K>::RS;
  for (;;) {
    if (tid == 0) sh_b = (int)atomicAdd(a.work_counter, 1u);
    __syncthreads();
    const int b = sh_b;
    __syncthreads();
    if (UNI(b >= a.B)) break;
    cx.load_problem(a.targets + (size_t)b * a.T, a.bt, sh_tgt);
    double x = cx.active ? a.Y_init[(size_t)b * NK + cx.gnode * K + cx.part] : 0.0;
    RtrOut ro;
    rcg_solve_one<K>(cx, a.cg, a.trace, a.has_trace, b, x, ro);
    if (cx.active) a.Y_out[(size_t)b * NK + cx.gnode * K + cx.part] = x;
    if (tid == 0) {
      gik_stats s;
      s.f = ro.f;
      s.gradnorm = ro.gradnorm;
      s.iterations = ro.iterations;
      s.inner_total = ro.inner_total;
      s.stop = ro.stop;
      s.n_accept = ro.n_accept;
      s.inner_executed = ro.inner_executed;
      s.flags = cx.lowrank ? 1 : 0;
      s.stepsize = ro.Delta;
      a.stats[b] = s;
    }
  }
}

template <int K>
__global__ void __launch_bounds__(BLOCK_NT) kat_block_kernel(KatArgs a, int SL) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int NK = a.N * K;
  BlockCtx<K> cx;
  block_stage<K>(cx, smem, a.slot_meta, a.N, a.bt, SL);
  double *sh_tgt = smem + 3 * BLOCK_MAXN * BlockCtx<K>::RS;
  cx.load_problem(a.targets ? a.targets + (size_t)b * a.T : nullptr, a.bt, sh_tgt);
  const size_t at = (size_t)b * NK + cx.gnode * K + cx.part;
  const double y = cx.active ? a.Y[at] : 0.0;
  const double w = (a.W && cx.active) ? a.W[at] : 0.0;
  const double f = cx.cost(y);
  if (a.mode == 0) {
    if (tid == 0) a.out[b] = f;
    return;
  }
  if (a.mode == 4 && tid == 0) a.out_f[b] = f;
  double res = cx.commit();
  if (a.mode == 2) {
    res = cx.ehess(w);
  } else if (a.mode == 3) {
    cx.proj_setup(a.planar_proj_exact);
    res = cx.proj(w);
  }
  if (cx.active) a.out[at] = res;
}

// ------------------------------------------------------------------------------------------
// node-per-lane variants (graphs with N*k > 64, k = 3): a lane owns whole graph nodes with all their
// components (gik_npt.hip.h), driver over a small vector per thread (gik_rtrv.hip.h).  <TL, 1, 2>: one
// node per lane, two wavefronts (128 threads) per problem; <TL, 2, 1>: two nodes per lane, one
// wavefront per problem.  Persistent; problems are claimed from the same ticket counter / re-queue
// ring as the workgroup kernel's (FIFO time slicing: the long problems, unknown in advance, must
// not be the last to START).  LDS-bound at three problems per CU on the table scene.
template <int TL, int NS, int NW, bool CTG = false>
__global__ void __launch_bounds__(WAVE * NW, 1) rtr_npt_kernel(SolveArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ int sh_claim[2];
  using Ctx = NptCtx<TL, NS, NW, CTG>;
  const int tid = threadIdx.x;
  const int NK = a.N * 3;
  Ctx cx;
  cx.init(a.nt, smem, CTG ? a.npt_ctg_ws + (size_t)blockIdx.x * Ctx::ctg_doubles(a.nt.n_pairs) : nullptr);
  int pass = 0;
  for (;;) {
    int b = 0, resumed = 0;
    if (a.dbg & 1) {      // static block -> problem map (keeps the claim below a scalar-branch loop, see rcg_wave_kernel)
      b = (int)blockIdx.x + pass * (int)gridDim.x;
      ++pass;
      if (b >= a.B) b = -1;
    } else if constexpr (NW == 1) {
      if (tid == 0) b = claim_work(a.work_counter, a.q_seq, a.q_ids, a.q_done, a.B, a.slice_its > 0, resumed);
      b = __builtin_amdgcn_readlane(b, 0);
      resumed = __builtin_amdgcn_readlane(resumed, 0);
    } else {
      if (tid == 0) {
        int res = 0;
        sh_claim[0] = claim_work(a.work_counter, a.q_seq, a.q_ids, a.q_done, a.B, a.slice_its > 0, res);
        sh_claim[1] = res;
      }
      __syncthreads();
      b = __builtin_amdgcn_readfirstlane(sh_claim[0]);
      resumed = __builtin_amdgcn_readfirstlane(sh_claim[1]);
      __syncthreads();
    }
    if (UNI(b < 0)) break;
    cx.load_problem(a.targets + (size_t)b * a.T, a.nt);
    // (unconditional load of the resume state from a zeroed array: a conditional one makes the solver's
    // counters divergent -- NOTEBOOK 8.3)
    RtrResume rs = {0.0, 0, 0, 0, 0, 0, 0};
    if (a.slice_its > 0) {
      rs = load_slice_state(&a.q_state[b]);
      rs.resumed = resumed;
    }
    const double *src = resumed ? a.Y_out : a.Y_init;
    double x[Ctx::NE];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        x[3 * s + q] = cx.live[s] ? __builtin_nontemporal_load(&src[(size_t)b * NK + cx.gnode[s] * 3 + q]) : 0.0;
    RtrOut ro;
    rtr_solve_vec<true, true>(cx, a.p, a.trace, a.has_trace, a.dbg, a.dbg_buf, b, x, ro, rs, a.slice_its);
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (cx.live[s]) {
#pragma unroll
        for (int q = 0; q < 3; ++q) a.Y_out[(size_t)b * NK + cx.gnode[s] * 3 + q] = x[3 * s + q];
      }
    if (UNI(ro.paused)) {
      if (tid == 0) {
        SliceState st;
        st.Delta = ro.Delta;
        st.kiter = ro.iterations;
        st.inner_total = ro.inner_total;
        st.inner_exec = ro.inner_executed;
        st.n_accept = ro.n_accept;
        st.resumes = rs.resumes + 1;
        st.pad = 0;
        a.q_state[b] = st;
      }
      __threadfence();
      Ctx::block_sync();
      if (tid == 0) requeue_work(a.q_tail, a.q_ids, a.q_seq, a.B, b);
      continue;
    }
    if (tid == 0) {
      gik_stats s;
      s.f = ro.f;
      s.gradnorm = ro.gradnorm;
      s.iterations = ro.iterations;
      s.inner_total = ro.inner_total;
      s.stop = ro.stop;
      s.n_accept = ro.n_accept;
      s.inner_executed = ro.inner_executed;
      s.flags = cx.lowrank ? 1 : 0;
      s.stepsize = ro.Delta;
      a.stats[b] = s;
      if (a.slice_its > 0) __hip_atomic_fetch_add(a.q_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int TL, int NS, int NW, bool CTG = false>
__global__ void __launch_bounds__(WAVE * NW, 1) kat_npt_kernel(KatArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using Ctx = NptCtx<TL, NS, NW, CTG>;
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int NK = a.N * 3;
  Ctx cx;
  cx.init(a.nt, smem, CTG ? a.npt_ctg_ws + (size_t)blockIdx.x * Ctx::ctg_doubles(a.nt.n_pairs) : nullptr);
  cx.load_problem(a.targets ? a.targets + (size_t)b * a.T : nullptr, a.nt);
  double y[Ctx::NE], w[Ctx::NE], res[Ctx::NE];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const size_t at = (size_t)b * NK + (cx.live[s] ? cx.gnode[s] : 0) * 3 + q;
      y[3 * s + q] = cx.live[s] ? a.Y[at] : 0.0;
      w[3 * s + q] = (a.W && cx.live[s]) ? a.W[at] : 0.0;
      res[3 * s + q] = 0.0;
    }
  const double f = cx.cost(y);
  if (a.mode == 0) {
    if (tid == 0) a.out[b] = f;
    return;
  }
  if (a.mode == 4 && tid == 0) a.out_f[b] = f;
  cx.commit(y, res);
  if (a.mode == 2) {
    cx.ehess(w, res);
  } else if (a.mode == 3) {     // PSDFixedRank.proj: Z - Q Q^T Z
    cx.proj_setup(y);
    double v[3], u[3];
    cx.vert_dots(w, v[0], v[1], v[2]);
    cx.template sum_n<3>(v);
    cx.vert_coords(v, u);
    cx.vert_apply(u, w, res);
  }
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (cx.live[s]) {
#pragma unroll
      for (int q = 0; q < 3; ++q) a.out[(size_t)b * NK + cx.gnode[s] * 3 + q] = res[3 * s + q];
    }
}

// ------------------------------------------------------------------------------------------
// Four planar problems per wavefront (gik_quad.hip.h): TrustRegions.solve (trust_region.py:112-434)
// with _truncated_conjugate_gradient (:436-599), the k = 2 branch of rtr_solve_one with every
// solver scalar a per-lane value and every decision a lane mask.  One pass of the outer loop below is
//   refill : slots without a problem claim the next one of the batch (fresh: x = Y_init, no step yet)
//   tCG    : the slots that continue a problem run truncated CG together until the last has left it
//   step   : cost of the proposal x + eta (fresh: of x itself), the acceptance test (:248-382; a
//            fresh problem "accepts" its start point), gradient / Hessian constants / projector at
//            the accepted points, stopping rules (:414-416), results of the finished slots.
template <int DEG>
__global__ void __launch_bounds__(WAVE, 3) rtr_quad_kernel(SolveArgs a) {
  using Ctx = QuadCtx<DEG>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  double2 *sh_P = reinterpret_cast<double2 *>(smem);
  double2 *sh_W = sh_P + QUAD_ROWS;
  double *sh_tg = reinterpret_cast<double *>(sh_W + QUAD_ROWS);
  int *sh_claim = reinterpret_cast<int *>(sh_tg + DEG * WAVE);
  Ctx cx;
  cx.init(lane, a.N, sh_P, sh_W, sh_tg, a.slot_meta);
  const Params &p = a.p;
  const int NK = a.N * 2;
  const bool lead = cx.node == 0;
  const double Delta_bar = 10.0 + 2;            // typicaldist (fixed_rank_psd_sym.py:71-73), k = 2
  const double hm = cx.has_node ? 1.0 : 0.0;

  // per-slot state (equal in the 16 lanes of a slot)
  bool alive = false, fresh = false, more = true;
  int b = -1, kiter = 0, inner_total = 0, n_accept = 0;
  double x0 = 0.0, x1 = 0.0, g0 = 0.0, g1 = 0.0, fx = 0.0, Delta = 0.0, norm_grad = 0.0, rho0 = 0.0;

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
          cx.load_targets(a.targets + (size_t)b * a.T);
          const double2 xi = cx.has_node ? *reinterpret_cast<const double2 *>(a.Y_init + (size_t)b * NK + 2 * cx.node)
                                         : make_double2(0.0, 0.0);
          x0 = xi.x;
          x1 = xi.y;
          kiter = inner_total = n_accept = 0;
          Delta = Delta_bar / 8.0;                   // trust_region.py:134-135,164
        } else {
          more = false;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (!quad_any(alive)) break;

    // ---------------- _truncated_conjugate_gradient (:436-599) ----------------
    const bool run = alive && !fresh;
    double eta0 = 0.0, eta1 = 0.0, He0 = 0.0, He1 = 0.0;     // :444-445
    int stop_tCG = TCG_MAX_INNER_ITER;                       // :491
    int jx = p.maxinner - 1;                                 // Python leaves j at the last index
    bool bad = false;
    const double Delta2 = Delta * Delta;
    if (quad_any(run)) {
      double r0 = g0, r1 = g1;                               // :448
      double e_Pe = 0.0;
      double r_r = quad_sum(fma(r1, r1, r0 * r0));           // :455
      const double norm_r0 = sqrt(r_r);
      const double target = norm_r0 * fmin(norm_r0, p.kappa);   // rhs of :572 (theta = 1)
      const double target2 = target * target;
      const int stop_target = (p.kappa < norm_r0) ? TCG_REACHED_TARGET_LINEAR : TCG_REACHED_TARGET_SUPERLINEAR;
      double z_r = r_r, d_Pd = r_r;                          // :464-466 (precon = identity)
      double inv_z_r = frcp(z_r);
      double d0 = -r0, d1 = -r1;                             // :469
      double e_Pd = 0.0, model_value = 0.0;                  // :471,485
      double rho_pk = rho0, s_pk = -rho0;                    // <r, pk2>, <delta, pk2>
      bool act = run && p.maxinner > 0;
      int j = 0;
      while (quad_any(act)) {                                // :495
        double H0, H1;
        cx.ehess(d0, d1, H0, H1);                            // :497
        const double v0 = quad_sum(fma(cx.pk[1], H1, cx.pk[0] * H0));
        const double v1 = quad_sum(fma(d1, H1, d0 * H0));
        const double v2 = quad_sum(fma(cx.pk2[1], H1, cx.pk2[0] * H0));
        // rhess = proj(ehess) (:497-500): Omega = v0 (Pm = 1), see WaveCtx::proj_dot
        const double Hd0 = fma(-cx.pk2[0], v0, H0), Hd1 = fma(-cx.pk2[1], v0, H1);
        const double d_Hd = fma(-v0, s_pk, v1);              // :500
        const double hd_pk = fma(-v0, cx.G2, v2);
        const bool nan = act && !(d_Hd == d_Hd);
        bad = bad || nan;
        act = act && !nan;
        const double alpha = z_r * frcp(d_Hd);               // :503
        const double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd;   // :506
        const bool exb = act && (d_Hd <= 0.0 || e_Pe_new >= Delta2);                // :509
        if (exb) {
          const double tau = (-e_Pd + sqrt(e_Pd * e_Pd + d_Pd * (Delta2 - e_Pe))) / d_Pd;   // :514
          eta0 = eta0 + tau * d0;                            // :516
          eta1 = eta1 + tau * d1;
          He0 = He0 + tau * Hd0;                             // :521
          He1 = He1 + tau * Hd1;
          stop_tCG = (d_Hd <= 0.0) ? TCG_NEGATIVE_CURVATURE : TCG_EXCEEDED_TR;      // :531-534
          jx = j;
        }
        act = act && !exb;
        if (!quad_any(act)) break;
        const double ne0 = eta0 + alpha * d0, ne1 = eta1 + alpha * d1;              // :538
        const double nH0 = He0 + alpha * Hd0, nH1 = He1 + alpha * Hd1;              // :542
        const double nr0 = r0 + alpha * Hd0, nr1 = r1 + alpha * Hd1;                // :561
        const double m0 = quad_sum(fma(ne1, g1, ne0 * g0));
        const double m1 = quad_sum(fma(ne1, nH1, ne0 * nH0));
        const double m2 = quad_sum(fma(nr1, nr1, nr0 * nr0));
        const double new_model_value = m0 + 0.5 * m1;        // :551
        const bool exm = act && (new_model_value >= model_value);                   // :552
        if (exm) {
          stop_tCG = TCG_MODEL_INCREASED;
          jx = j;
        }
        act = act && !exm;
        if (act) {
          e_Pe = e_Pe_new;                                   // :537
          eta0 = ne0;                                        // :556-558
          eta1 = ne1;
          He0 = nH0;
          He1 = nH1;
          model_value = new_model_value;
          r0 = nr0;                                          // :561
          r1 = nr1;
          r_r = m2;                                          // :564
        }
        const bool ext = act && (j >= p.mininner && r_r <= target2);                // :572
        if (ext) {
          stop_tCG = stop_target;
          jx = j;
        }
        act = act && !ext;
        act = act && (j + 1 < p.maxinner);                   // :495 exhausted: stop stays MAX_INNER_ITER
        if (act) {
          z_r = r_r;                                         // :589
          const double beta = z_r * inv_z_r;                 // :592
          inv_z_r = frcp(z_r);
          d0 = -r0 + beta * d0;                              // :593
          d1 = -r1 + beta * d1;
          rho_pk = fma(alpha, hd_pk, rho_pk);
          s_pk = fma(beta, s_pk, -rho_pk);
          e_Pd = beta * (e_Pd + alpha * d_Pd);               // :596
          d_Pd = z_r + beta * beta * d_Pd;                   // :597
        }
        ++j;
      }
    }
    // (a NaN in tCG leaves the solve as rtr_solve_one does: point, counters and statistics as before it)
    const bool step = run && !bad;
    if (step) inner_total += jx + 1;

    // ---------------- outer iteration (:248-422) ----------------
    const bool tr = a.has_trace && step && lead && kiter < a.trace.cap;
    if (tr) {
      const size_t q = (size_t)b * a.trace.cap + kiter;
      a.trace.d_Delta[q] = Delta;
      a.trace.d_numit[q] = jx;
      a.trace.d_stop[q] = stop_tCG;
      a.trace.d_f_before[q] = fx;
    }
    const double xp0 = fresh ? x0 : x0 + eta0, xp1 = fresh ? x1 : x1 + eta1;        // :248 retr
    const double fx_prop = cx.cost(xp0, xp1);                // :251 (fresh: :159)
    const double gd0 = quad_sum(fma(g1, eta1, g0 * eta0)), gd1 = quad_sum(fma(eta1, He1, eta0 * He0));
    double rhonum = fx - fx_prop;                            // :255
    double rhoden = -gd0 - 0.5 * gd1;                        // :256
    const double rho_reg = fmax(1.0, fabs(fx)) * 2.220446049250313e-16 * p.rho_regularization;   // :287
    rhonum += rho_reg;                                       // :288
    rhoden += rho_reg;                                       // :289
    const bool model_decreased = rhoden >= 0.0;              // :311
    const double rho = rhonum / rhoden;                      // :317
    if (step) {
      if (rho < 0.25 || !model_decreased || !(rho == rho)) {                        // :336
        Delta = Delta / 4.0;                                 // :338
      } else if (rho > 0.75 && (stop_tCG == TCG_NEGATIVE_CURVATURE || stop_tCG == TCG_EXCEEDED_TR)) {
        Delta = fmin(2.0 * Delta, Delta_bar);                // :357-361
      }
    }
    const bool accept = step && model_decreased && rho > p.rho_prime;               // :382
    if (accept || fresh) {
      x0 = xp0;                                              // :385
      x1 = xp1;
      fx = fx_prop;                                          // :386
      cx.commit(g0, g1);                                     // :387 (fresh: :160)
    }
    if (accept) ++n_accept;
    // projector, ||grad|| and <grad, pk2> are functions of (x, grad): recomputed for every slot, the
    // same bits again where nothing was accepted
    cx.proj_setup(x0, x1, p.planar_proj_exact);
    norm_grad = sqrt(quad_sum(hm * fma(g1, g1, g0 * g0)));   // :388 (fresh: :161)
    rho0 = quad_sum(fma(g1, cx.pk2[1], g0 * cx.pk2[0]));
    if (tr) {
      const size_t q = (size_t)b * a.trace.cap + kiter;
      a.trace.d_gradnorm_after[q] = norm_grad;
      a.trace.d_accept[q] = accept ? 1 : 0;
    }
    if (step) ++kiter;                                       // :394
    // :414-416 stopping criterion (pymanopt 0.2.5 order: maxiter before gradnorm)
    const bool isnan = !(norm_grad == norm_grad) || !(fx == fx);
    int stop = -1;
    if (bad) stop = 2;
    else if (step && kiter >= p.maxiter) stop = 1;
    else if (step && norm_grad < p.mingradnorm) stop = 0;
    else if (alive && (isnan || (a.dbg & 2))) stop = 2;
    const bool fin = alive && stop >= 0;
    if (fin) {
      if (cx.has_node) *reinterpret_cast<double2 *>(a.Y_out + (size_t)b * NK + 2 * cx.node) = make_double2(x0, x1);
      if (lead) {
        gik_stats s;
        s.f = fx;
        s.gradnorm = norm_grad;
        s.iterations = kiter;
        s.inner_total = inner_total;
        s.stop = stop;
        s.n_accept = n_accept;
        s.inner_executed = inner_total;
        s.flags = 0;
        s.stepsize = Delta;
        a.stats[b] = s;
      }
      alive = false;
    }
    fresh = false;
  }
}

// The device functions of QuadCtx one call at a time (known-answer tests: gik_cost / gik_grad / gik_hess / gik_proj
// on a template created with debug_flags 16384), four problems per wavefront like the solve kernel
template <int DEG>
__global__ void __launch_bounds__(WAVE) kat_quad_kernel(KatArgs a) {
  using Ctx = QuadCtx<DEG>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  double2 *sh_P = reinterpret_cast<double2 *>(smem);
  double2 *sh_W = sh_P + QUAD_ROWS;
  double *sh_tg = reinterpret_cast<double *>(sh_W + QUAD_ROWS);
  Ctx cx;
  cx.init(lane, a.N, sh_P, sh_W, sh_tg, a.slot_meta);
  const int b_raw = (int)blockIdx.x * QUAD_SLOTS + cx.slot, NK = a.N * 2;
  const bool valid = b_raw < a.B;
  const int b = valid ? b_raw : a.B - 1;
  if (a.targets) cx.load_targets(a.targets + (size_t)b * a.T);
  const double2 y = cx.has_node ? *reinterpret_cast<const double2 *>(a.Y + (size_t)b * NK + 2 * cx.node) : make_double2(0.0, 0.0);
  const double2 w = (a.W && cx.has_node) ? *reinterpret_cast<const double2 *>(a.W + (size_t)b * NK + 2 * cx.node)
                                         : make_double2(0.0, 0.0);
  const double f = cx.cost(y.x, y.y);        // (also publishes the rows of y)
  if (a.mode == 0) {
    if (valid && cx.node == 0) a.out[b] = f;
    return;
  }
  double r0 = 0.0, r1 = 0.0;
  if (a.mode == 1 || a.mode == 4) {
    cx.commit(r0, r1);
    if (a.mode == 4 && valid && cx.node == 0) a.out_f[b] = f;
  } else if (a.mode == 2) {
    double g0, g1;
    cx.commit(g0, g1);
    cx.ehess(w.x, w.y, r0, r1);
  } else {
    cx.proj_setup(y.x, y.y, a.planar_proj_exact);
    const double o = quad_sum(fma(cx.pk[1], w.y, cx.pk[0] * w.x));     // Omega (Pm = 1)
    r0 = fma(-cx.pk2[0], o, w.x);
    r1 = fma(-cx.pk2[1], o, w.y);
  }
  if (valid && cx.has_node) *reinterpret_cast<double2 *>(a.out + (size_t)b * NK + 2 * cx.node) = make_double2(r0, r1);
}

// ------------------------------------------------------------------------------------------
// developer micro-benchmark: per-component cycle cost of one wavefront.  Only in the -DGIK_DEV
// build (graphik_amd/build.py --dev -> lib/exp/libgraphik_amd_dev.so); the shipped library has
// neither this kernel nor the gik_debug_* hooks.
#ifdef GIK_DEV
template <int K, int MAXDEG>
__global__ void __launch_bounds__(WAVE) parts_kernel(const uint32_t *slot_meta, int N, int T, int mode,
                                                     int iters, double *out) {
  using Ctx = WaveCtx<K, MAXDEG>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  double *sh_tiles = smem;
  double *sh_tgt = smem + Ctx::NTILE * Ctx::TILE;
  uint32_t *sh_meta = reinterpret_cast<uint32_t *>(sh_tgt + ((T + 1) & ~1));
  stage_lds<Ctx>(sh_tiles, sh_meta, slot_meta, lane, MAXDEG, N);
  for (int t = lane; t < T; t += WAVE) sh_tgt[t] = 1.0 + 0.01 * t;
  Ctx cx;
  cx.init(lane, N, sh_tiles, sh_tgt, sh_meta);
  cx.load_slot_records();
  double x = cx.active ? 0.37 * lane - 0.01 * lane * lane : 0.0;
  (void)cx.cost(x);
  double g = cx.commit();
  cx.proj_setup(0);
  double acc = g, sc = 1.0;
  const long long t0 = __builtin_readcyclecounter();
  if (mode == 0) {
    for (int it = 0; it < iters; ++it) {            // ehess only
      acc = cx.ehess(acc) * 1e-3 + g;
    }
  }
  else if (mode == 1) {
    for (int it = 0; it < iters; ++it) {     // 3-value reduction
      double v[3] = {acc, acc * 0.5, acc * 0.25};
      wave_sum_n<3>(v);
      acc = g + 1e-3 * (v[0] + v[1] + v[2]);
    }
  }
  else if (mode == 2) {
    for (int it = 0; it < iters; ++it) {     // 1-value reduction
      acc = g + 1e-3 * wave_sum(acc);
    }
  }
  else if (mode == 3) {
    for (int it = 0; it < iters; ++it) {     // fp64 division chain
      sc = 1.0 / (sc + 1.5);
      acc = acc + sc;
    }
  }
  else if (mode == 4) {
    for (int it = 0; it < iters; ++it) {     // proj(ehess)
      acc = cx.proj(cx.ehess(acc)) * 1e-3 + g;
    }
  }
  else if (mode == 5) {
    for (int it = 0; it < iters; ++it) {     // dependent fma chain (8 per iteration)
      for (int q = 0; q < 8; ++q) acc = fma(acc, 0.999, g);
    }
  }
  else if (mode == 6) {
    for (int it = 0; it < iters; ++it) {     // sqrt chain
      sc = sqrt(sc + 1.5);
      acc = acc + sc;
    }
  }
  else if (mode == 7) {
    for (int it = 0; it < iters; ++it) {     // 8 independent fma chains x 8 (64 fma / iteration)
      double c0 = acc, c1 = acc + 1, c2 = acc + 2, c3 = acc + 3, c4 = acc + 4, c5 = acc + 5,
             c6 = acc + 6, c7 = acc + 7;
      for (int q = 0; q < 8; ++q) {
        c0 = fma(c0, 0.999, g); c1 = fma(c1, 0.999, g); c2 = fma(c2, 0.999, g);
        c3 = fma(c3, 0.999, g); c4 = fma(c4, 0.999, g); c5 = fma(c5, 0.999, g);
        c6 = fma(c6, 0.999, g); c7 = fma(c7, 0.999, g);
      }
      acc = ((c0 + c1) + (c2 + c3)) + ((c4 + c5) + (c6 + c7));
    }
  }
  else if (mode == 8) {
    for (int it = 0; it < iters; ++it) {     // LDS write + dependent read round trip
      cx.put(acc);
      acc = cx.read_row(cx.own_off).v[1] + g;
    }
  }
  else if (mode == 9) {
    for (int it = 0; it < iters; ++it) {     // one DPP move + add (dependent)
      acc = acc + dpp_f64<0xB1>(acc) * 1e-3;
    }
  }
  else if (mode == 10) {
    for (int it = 0; it < iters; ++it) {    // readlane + add (dependent)
      acc = g + readlane_f64(acc, 17) * 1e-3;
    }
  }
  else if (mode == 11) {
    for (int it = 0; it < iters; ++it) {    // 8 dependent f32 fma
      float fa = (float)acc;
      for (int q = 0; q < 8; ++q) fa = fmaf(fa, 0.999f, 0.5f);
      acc = fa;
    }
  }
  else if (mode == 12 || mode == 13 || mode == 14) {
    for (int it = 0; it < iters; ++it) {  // 64 independent add / mul / fma(vvv)
      double c0 = acc, c1 = acc + 1, c2 = acc + 2, c3 = acc + 3, c4 = acc + 4, c5 = acc + 5,
             c6 = acc + 6, c7 = acc + 7;
      const double h = g * 0.5 + 1.0;
#define OP8(EXPR)                                                                    \
  for (int q = 0; q < 8; ++q) {                                                      \
    { double &c = c0; c = EXPR; } { double &c = c1; c = EXPR; } { double &c = c2; c = EXPR; } \
    { double &c = c3; c = EXPR; } { double &c = c4; c = EXPR; } { double &c = c5; c = EXPR; } \
    { double &c = c6; c = EXPR; } { double &c = c7; c = EXPR; }                        \
  }
      if (mode == 12) { OP8(c + g) } else if (mode == 13) { OP8(c * h) } else { OP8(fma(c, h, g)) }
      acc = ((c0 + c1) + (c2 + c3)) + ((c4 + c5) + (c6 + c7));
    }
  }
  else if (mode == 15) {
    for (int it = 0; it < iters; ++it) {    // 64 independent 32-bit ops (v_add_u32 / xor mix)
      int c[8];
      for (int q = 0; q < 8; ++q) c[q] = __double2loint(acc) + q;
      for (int q = 0; q < 8; ++q)
        for (int w = 0; w < 8; ++w) c[w] = (c[w] ^ (c[w] >> 3)) + lane;   // 3 ops each
      int z = 0;
      for (int q = 0; q < 8; ++q) z ^= c[q];
      acc = g + 1e-9 * z;
    }
  }
  else if (mode == 16) {
    for (int it = 0; it < iters; ++it) {    // 32 independent DPP f64 moves (64 v_mov_dpp) + adds
      double c[8];
      for (int q = 0; q < 8; ++q) c[q] = acc + q;
      for (int q = 0; q < 4; ++q)
        for (int w = 0; w < 8; ++w) c[w] = dpp_f64<0xB1>(c[w]);
      acc = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
    }
  }
  else if (mode == 17) {
    for (int it = 0; it < iters; ++it) {    // 32 selects on f64 (64 v_cndmask)
      double c[8];
      for (int q = 0; q < 8; ++q) c[q] = acc + q;
      for (int q = 0; q < 4; ++q)
        for (int w = 0; w < 8; ++w) c[w] = ((lane >> q) & 1) ? c[w] : c[(w + 1) & 7];
      acc = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
    }
  }
  else if (mode == 18) {
    for (int it = 0; it < iters; ++it) {    // 11 x (ds_read_b128 + ds_read_b64) of the gather, no math
      cx.put(acc);
      double z = 0.0;
      for (int s2 = 0; s2 < MAXDEG; ++s2) {
        const Row<K> r = cx.read_row(cx.rowoff(s2));
        z += r.v[0];
      }
      acc = g + z * 1e-3;
    }
  }
  else if (mode == 23) {
    for (int it = 0; it < iters; ++it) {     // 1-value reduction, pure DPP butterfly
      double t = acc;
      t += dpp_f64<0xB1>(t);
      t += dpp_f64<0x4E>(t);
      t += dpp_f64<0x141>(t);
      t += dpp_f64<0x140>(t);
      const double r0 = readlane_f64(t, 0), r1 = readlane_f64(t, 16), r2 = readlane_f64(t, 32), r3 = readlane_f64(t, 48);
      acc = g + 1e-3 * ((r0 + r1) + (r2 + r3));
    }
  }
  else if (mode == 19) {
    for (int it = 0; it < iters; ++it) acc = acc * 1e-9 + cx.cost(x + acc * 1e-12);
  } else if (mode == 20) {
    for (int it = 0; it < iters; ++it) acc = acc * 1e-9 + cx.commit();
  } else if (mode == 21) {
    for (int it = 0; it < iters; ++it) {
      cx.proj_setup(0);
      acc = acc * 1e-9 + cx.Q[0];
    }
  } else if (mode == 22) {
    for (int it = 0; it < iters; ++it) acc = acc * 1e-9 + sqrt(cx.sum1(acc * acc + g));
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0) {
    out[0] = (double)(t1 - t0) / iters;
    out[1] = acc + sc;
  }
}
#endif  // GIK_DEV

}  // namespace gik
