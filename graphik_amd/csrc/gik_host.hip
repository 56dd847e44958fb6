// graphik_amd/csrc/gik_host.hip -- host side of the library: template creation (slot tables, clique detection,
// node-per-lane tables), the scheduling slots of a batch call, pipeline attach and the C ABI of
// include/graphik_amd.h.  No device code is compiled here: the kernels live in gik_k_*.hip (gik_instances.h).

#include "gik_kernels.hip.h"
#include "gik_instances.h"

namespace gik {
GIK_ALL_KERNELS(GIK_EXTERN_TEMPLATE)

// the compiled node-per-lane variants
struct NptVariant {
  int TL, NW;
  void (*solve)(SolveArgs);
  void (*kat)(KatArgs);
  size_t (*lds)(int, int, int, int);
  size_t (*ctg)(int);      // doubles of global clique-target workspace per workgroup (0: the triangle sits in LDS)
};
template <int TL, int NS, int NW, bool CTG>
static size_t npt_lds_of(int n_pairs, int n_wrows, int n_rows, int n_terms) {
  return NptCtx<TL, NS, NW, CTG>::lds_bytes(n_pairs, n_wrows, n_rows, n_terms);
}
template <int TL, int NS, int NW, bool CTG>
static size_t npt_ctg_of(int n_pairs) {
  return NptCtx<TL, NS, NW, CTG>::ctg_doubles(n_pairs);
}
#define GIK_NPT_VARIANT(TL, NS, NW, CTG) \
  {TL, NW, rtr_npt_kernel<TL, NS, NW, CTG>, kat_npt_kernel<TL, NS, NW, CTG>, npt_lds_of<TL, NS, NW, CTG>, npt_ctg_of<TL, NS, NW, CTG>}
// (four wavefronts per problem: graphs of 129 .. 255 nodes, clique targets in global memory)
static const NptVariant kNptVariants[] = {GIK_NPT_VARIANT(1, 1, 2, false), GIK_NPT_VARIANT(4, 1, 2, false),
                                          GIK_NPT_VARIANT(1, 2, 1, false), GIK_NPT_VARIANT(4, 2, 1, false),
                                          GIK_NPT_VARIANT(1, 1, 4, true),  GIK_NPT_VARIANT(4, 1, 4, true)};

// ------------------------------------------------------------------------------------------
thread_local std::string g_err;
#ifdef GIK_DEV
static double *g_dbg_buf = nullptr;
#endif
static int fail(const std::string &m) {
  g_err = m;
  return -1;
}
#define HIP_OK(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(std::string(#expr) + ": " + hipGetErrorString(e_));                     \
  } while (0)

typedef void (*solve_fn)(SolveArgs);
typedef void (*kat_fn)(KatArgs);
typedef size_t (*lds_fn)(int);

template <int K, int D>
static size_t lds_bytes_of(int T) {
  return WaveCtx<K, D>::lds_bytes(T);
}

template <int K, int D>
static size_t lds_bytes_strict(int T) {
  if constexpr (K == 3) return WaveCtxStrict<D>::lds_bytes(T);
  return 0;
}

template <int K, int D>
static size_t lds_bytes_anch(int T) {
  return WaveCtx<K, D, true>::lds_bytes(T);
}
struct Variant {
  int K, maxdeg;
  solve_fn solve;        // theta == 1 (reference default)
  solve_fn solve_theta;  // any theta
  solve_fn solve_cg;     // ConjugateGradient
  kat_fn kat;
  lds_fn lds;
  solve_fn solve_anch;   // fixed-anchor formulation (k = 3, theta == 1), or null
  kat_fn kat_anch;
  lds_fn lds_anch;
  solve_fn solve_mig;    // theta == 1 with tail spreading (MigCtl), or null
  solve_fn solve_strict, solve_strict_mig;   // hessian_form = GIK_HESS_PER_EDGE (k = 3, theta == 1), or null
  kat_fn kat_strict;
  lds_fn lds_strict;
  solve_fn solve_strict_theta;               // ... any theta
};
#define GIK_VARIANT(K, D) \
  {K, D, rtr_wave_kernel<K, D, true>, rtr_wave_kernel<K, D, false>, rcg_wave_kernel<K, D>, kat_wave_kernel<K, D>, \
   lds_bytes_of<K, D>, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}
#define GIK_VARIANT_S(K, D) \
  {K, D, rtr_wave_kernel<K, D, true>, rtr_wave_kernel<K, D, false>, rcg_wave_kernel<K, D>, kat_wave_kernel<K, D>, \
   lds_bytes_of<K, D>, nullptr, nullptr, nullptr, nullptr, rtr_wave_kernel<K, D, true, false, false, true>, \
   rtr_wave_kernel<K, D, true, false, true, true>, kat_wave_kernel<K, D, false, true>, lds_bytes_strict<K, D>, \
   rtr_wave_kernel<K, D, false, false, false, true>}
#define GIK_VARIANT_A(K, D) \
  {K, D, rtr_wave_kernel<K, D, true>, rtr_wave_kernel<K, D, false>, rcg_wave_kernel<K, D>, kat_wave_kernel<K, D>, \
   lds_bytes_of<K, D>, rtr_wave_kernel<K, D, true, true>, kat_wave_kernel<K, D, true>, lds_bytes_anch<K, D>, nullptr, \
   nullptr, nullptr, nullptr, nullptr, nullptr}
#define GIK_VARIANT_AM(K, D) \
  {K, D, rtr_wave_kernel<K, D, true>, rtr_wave_kernel<K, D, false>, rcg_wave_kernel<K, D>, kat_wave_kernel<K, D>, \
   lds_bytes_of<K, D>, rtr_wave_kernel<K, D, true, true>, kat_wave_kernel<K, D, true>, lds_bytes_anch<K, D>, \
   rtr_wave_kernel<K, D, true, false, true>, rtr_wave_kernel<K, D, true, false, false, true>, \
   rtr_wave_kernel<K, D, true, false, true, true>, kat_wave_kernel<K, D, false, true>, lds_bytes_strict<K, D>, \
   rtr_wave_kernel<K, D, false, false, false, true>}
// anchored templates only: the free-free formulation with more than 10 terms at a node runs on the
// workgroup kernels (the 20-slot wavefront variant needed 796 B of scratch per lane: measured on the
// two-end-effector tree of tests/golden/tree5.npz, 13 terms, 144 k against 382 k solves/s;
// tools/gpu_variants.py.  <3,10> 48 B: -1.3 % against <3,9>; <2,16> 168 B: 5x faster than the
// workgroup kernels on the planar trees -- both stay)
#define GIK_VARIANT_ANCH_ONLY(K, D) \
  {K, D, nullptr, nullptr, nullptr, nullptr, lds_bytes_of<K, D>, rtr_wave_kernel<K, D, true, true>, \
   kat_wave_kernel<K, D, true>, lds_bytes_anch<K, D>, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}
static const Variant kVariants[] = {GIK_VARIANT_AM(3, 9), GIK_VARIANT_S(3, 10), GIK_VARIANT_ANCH_ONLY(3, 20),
                                    GIK_VARIANT(2, 6), GIK_VARIANT(2, 16), GIK_VARIANT(2, 31)};

}  // namespace gik

struct gik_template {
  int N, K, T, maxdeg;
  // fixed-anchor formulation (gik_template_create_anchored)
  bool anchored = false;
  hipEvent_t ev_solve0 = nullptr, ev_solve1 = nullptr;   // around the solve kernel of the last gik_anchored_ik_batch
  gik::AnchArgs an;                 // device pointers + counts (anchor_goal filled per call)
  double *d_targets_const = nullptr;   // [T] template-constant targets of the free-free terms
  int full_N = 0, n_anchor = 0;
  int *d_free_full = nullptr, *d_anchor_full = nullptr;   // node index in the full robot graph
  double axis_length = 1.0;
  int solver;
  gik::CgParams cg;
  gik::Params p;
  const gik::Variant *variant;
  uint32_t *d_slot_meta;
  // Ring of work-queue heads, one per in-flight solve call.  A slot is handed out again only
  // behind the event recorded after the launch that used it last (the new call's stream waits for
  // it), so a wrap of the ring can never reset the counter of a kernel that is still running --
  // whatever the number of calls in flight.
  unsigned int *d_counters;
  struct CounterSlot {
    hipEvent_t done = nullptr;
    bool pending = false;
    bool in_use = false;     // handed to a call that has not recorded `done` yet
  };
  std::vector<CounterSlot> counter_slot;   // [kCounterRing]
  unsigned next_counter = 0;
  int counter_ring = 256;   // slots in use (GIK_COUNTER_RING at creation: tests shrink it to force wraps)
  std::mutex call_mutex;    // counter ring + time-slicing pool: held for the slot hand-out and the hand-back only
                            // (a slot marked in_use belongs to its call: blocking work happens outside the lock)
  std::mutex ev_mutex;      // ev_solve0 / ev_solve1 (anchored templates)
  int clique_mode = 0;      // gik_template_desc::clique_closed_form as resolved at creation
  bool hess_per_edge = false;   // the wavefront kernel (k = 3) runs the per-edge product form (gik_template_desc::hessian_form)
  // time-slicing workspaces (re-queue ring + paused state), a small pool handed out round-robin;
  // a launch that gets a slot still in use by an earlier launch waits for it on its stream
  struct SliceWs {
    void *base = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;
    bool pending = false;
    bool in_use = false;     // handed to a call that has not recorded `done` yet (only its owner touches the slot)
  };
  static constexpr int kSlicePool = 32;
  SliceWs slice_ws[kSlicePool];
  unsigned next_slice = 0;
  int slice_pool = kSlicePool;   // slots in use (GIK_SLICE_POOL at creation: tests shrink it to force reuse)
  int device;
  int n_cu;
  // scheduling knobs, fixed at creation (descriptor fields, overridden once by the environment)
  int dbg;            // SolveArgs::dbg
  int wpc_override;   // persistent waves per CU, 0 = automatic
  int slice_its;      // time slice of the block kernel in outer iterations, 0 = off
  int npt_slice_its = 192;   // ... of the node-per-lane kernel
  int wave_slice_its; // round-robin slice of the wavefront kernel (large batches), 0 = off
  bool wave_slice_auto = true;   // ... scaled with the queue depth (gik_solve_batch)
  int wave_slice_cycles = 2000000;   // ... and its shortest duration (GIK_SLICE_CYCLES)
  int waves_per_cu;  // resident solve wavefronts per CU (from the occupancy query)
  size_t smem_bytes;
  bool is_block;  // workgroup-per-problem path
  int SL;         // slots per thread on the block path
  int SLE;        // ... of which the first SLE hold equality terms (or padding) only
  gik::BlockTabs bt = {nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0};
  // node-per-lane path (rtr_npt_kernel): trust-region solves and the known-answer entry points of
  // 3-D graphs beyond one wavefront's 64 unknowns; the workgroup tables above stay (ConjugateGradient)
  bool is_npt = false;
  gik::NptTabs nt = {};
  const gik::NptVariant *npt_variant = nullptr;
  size_t npt_smem = 0;
  int npt_waves_per_cu = 1;
  // four-problems-per-wavefront path (rtr_quad_kernel): trust-region solves of planar graphs with at most
  // 16 nodes and 6 terms per node; everything else of such a template stays on the wavefront kernels
  void (*quad_solve)(gik::SolveArgs) = nullptr;
  size_t quad_smem = 0;
  int quad_waves_per_cu = 8;
  int quad_min_batch = 0;      // smallest batch that runs it (12 problems per CU; GIK_QUAD_MIN_BATCH)
  // device pre/post-processing (gik_pipeline_attach)
  bool has_pipe;
  gik::PipeConst pc;
  std::vector<void *> pipe_allocs;
  size_t prep_smem;
  int prep_waves_per_cu = 8;   // resident prepare waves (workgroups on the block variant) per CU
  bool prep_block = false;
  bool prep_quad = false;      // four goals per wavefront (prep_quad_kernel: graphs of at most 16 nodes)
  size_t prep_quad_smem = 0;
  int prep_quad_waves_per_cu = 8;
  bool prep_a_lds = false;     // block variant: work matrix in LDS
  bool prep_big = false;       // block variant for graphs of 129 .. 255 nodes (work matrix in the slab, always compressed)
  bool prep_no_compress = false;   // block variant: full N x N Jacobi even where the Gram matrix is rank deficient
  double *prep_ws = nullptr;   // [n_cu * prep_waves_per_cu][5][N*N] (block variant)
  hipEvent_t prep_done = nullptr;   // block variant: launches share prep_ws, so each one waits
  std::mutex prep_mutex;            // for the previous one (whatever stream it ran on)
  bool prep_pending = false;
  int sweeps;
};
static constexpr int kCounterRing = 256;

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel, not of a launch: several
// templates share a kernel, so the allowance is only ever raised (a later, smaller template must
// not take it away from an earlier one).
static hipError_t raise_dynamic_lds(const void *fn, size_t bytes) {
  struct Grant { const void *fn; int device; size_t bytes; };
  static std::mutex mu;
  static std::vector<Grant> granted;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  for (Grant &g : granted)
    if (g.fn == fn && g.device == dev) {
      if (g.bytes >= bytes) return hipSuccess;
      const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e == hipSuccess) g.bytes = bytes;
      return e;
    }
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) granted.push_back({fn, dev, bytes});
  return e;
}

static bool capturing_stream(void *stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (!stream) return false;      // (the null stream cannot capture)
  if (hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return st != hipStreamCaptureStatusNone;
}

template <typename T>
static const T *upload(gik_template *t, const T *host, size_t count, bool &ok) {
  if (count == 0 || !host) return nullptr;
  void *d = nullptr;
  if (hipMalloc(&d, count * sizeof(T)) != hipSuccess ||
      hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
    ok = false;
    return nullptr;
  }
  t->pipe_allocs.push_back(d);
  return static_cast<const T *>(d);
}

extern "C" {

const char *gik_last_error(void) { return gik::g_err.c_str(); }
int gik_abi_version(void) { return GIK_ABI_VERSION; }

int gik_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void gik_default_params(gik_template_desc *d) {
  d->abi_version = GIK_ABI_VERSION;
  d->mingradnorm = 0.5 * 1e-9;  // riemannian_solver.py:45
  d->maxiter = 3000;            // :47
  d->maxinner = 10000;          // trust_region.py:118
  d->mininner = 1;              // trust_region.py:116
  d->theta = 1.0;               // riemannian_solver.py:48
  d->kappa = 0.1;               // :49
  d->rho_prime = 0.1;           // trust_region.py:90
  d->rho_regularization = 1e3;  // trust_region.py:92
  d->planar_proj_exact = 0;
  d->force_block_path = 0;
  d->waves_per_cu = 0;
  d->slice_outer_its = -1;
  d->debug_flags = 0;
  d->solver = GIK_SOLVER_TRUST_REGIONS;
  d->cg_minstepsize = 1e-10;    // riemannian_solver.py:56
  d->cg_orth_value = 10e10;     // :57
  d->cg_beta_type = 3;          // :58  BetaTypes[3] = HagerZhang
  d->clique_closed_form = GIK_CLIQUE_AUTO;
  d->hessian_form = GIK_HESS_AUTO;
}

void gik_default_cg_params(gik_template_desc *d) {
  gik_default_params(d);
  d->solver = GIK_SOLVER_CONJUGATE_GRADIENT;
  d->mingradnorm = 1e-9;        // riemannian_solver.py:53
  d->maxiter = 100000;          // :55  (10e4)
}

static int create_impl(const gik_template_desc *d, const gik_anchored_desc *ad, gik_template **out);

int gik_template_create(const gik_template_desc *d, gik_template **out) {
  return create_impl(d, nullptr, out);
}

int gik_template_create_anchored(const gik_template_desc *d, const gik_anchored_desc *ad, gik_template **out) {
  if (!ad) return gik::fail("null argument");
  return create_impl(d, ad, out);
}

static int create_impl(const gik_template_desc *d, const gik_anchored_desc *ad, gik_template **out) {
  using namespace gik;
  if (!d || !out) return fail("null argument");
  if (ad) {
    if (d->k != 3 || d->solver != GIK_SOLVER_TRUST_REGIONS || d->theta != 1.0 || d->force_block_path)
      return fail("anchored templates: k = 3, TrustRegions, theta = 1, wavefront path");
    if (ad->n_anchor < 1 || ad->n_anchor > ANCH_MAXA || ad->n_goal_anchor < 0 || ad->n_goal_anchor > ad->n_anchor)
      return fail("anchored templates: 1 <= n_anchor <= 16, goal anchors are the last rows");
    if (ad->n_obs > ANCH_MAXOBS) return fail("anchored templates: at most 128 obstacles");
    if (ad->n_obs < 0 || ad->n_pin < 0 || d->N > 63 || !ad->term_target || !ad->free_full_index ||
        !ad->anchor_full_index || !ad->anchor_pos)
      return fail("anchored templates: bad descriptor");
  }
  if (d->abi_version != GIK_ABI_VERSION) return fail("ABI version mismatch");
  if (d->k != 2 && d->k != 3) return fail("k must be 2 or 3");
  if (d->solver != GIK_SOLVER_TRUST_REGIONS && d->solver != GIK_SOLVER_CONJUGATE_GRADIENT)
    return fail("solver must be GIK_SOLVER_TRUST_REGIONS or GIK_SOLVER_CONJUGATE_GRADIENT");
  if (d->cg_beta_type < 0 || d->cg_beta_type > 3) return fail("cg_beta_type must be 0..3");
  if (d->clique_closed_form < GIK_CLIQUE_AUTO || d->clique_closed_form > GIK_CLIQUE_DENSE)
    return fail("clique_closed_form must be GIK_CLIQUE_AUTO, _OFF or _DENSE");
  if (d->hessian_form != GIK_HESS_COLUMN && d->hessian_form != GIK_HESS_PER_EDGE && d->hessian_form != GIK_HESS_AUTO)
    return fail("hessian_form must be GIK_HESS_AUTO, GIK_HESS_COLUMN or GIK_HESS_PER_EDGE");

  bool is_block = d->N * d->k > WAVE || d->N > 32 || d->force_block_path != 0;
  // 255 = what the node-per-lane kernel's 8-bit row fields take (four wavefronts per problem beyond 128 nodes);
  // every other kernel stops at 128 (checked below, where the kernel is chosen)
  if (d->N < 2 || d->N > 255) return fail("N must be in [2, 255]");
  const bool big = d->N > BLOCK_MAXN;
  if (big && (ad || d->k != 3 || d->solver != GIK_SOLVER_TRUST_REGIONS || d->theta != 1.0 || d->force_block_path == 1))
    return fail("graphs of more than 128 nodes run on the node-per-lane kernel only: k = 3, TrustRegions, theta = 1, "
                "not anchored, force_block_path != 1");
  if (d->n_terms < 1 || d->n_terms > 65535) return fail("n_terms out of range");
  const int N = d->N, T = d->n_terms;
  int dbg_eff = d->debug_flags;   // developer override, read once here (never inside a batch call)
  if (const char *e = getenv("GIK_DBG")) dbg_eff = atoi(e);
  // per-node slot lists, in (neighbour, kind) order == the order the reference's edge loop
  // (row-major upper-triangle index pairs) accumulates into each row
  std::vector<std::vector<uint32_t>> slots(N);
  struct Ent { int j, kind, term, owner; };
  std::vector<std::vector<Ent>> ents(N);
  for (int t = 0; t < T; ++t) {
    const int i = d->term_i[t], j = d->term_j[t], kind = d->term_kind[t];
    if (i < 0 || j < 0 || i >= N || j >= N || i == j) return fail("bad term indices");
    if (kind < GIK_TERM_EQ || kind > GIK_TERM_UPPER) return fail("bad term kind");
    ents[i].push_back({j, kind, t, i < j ? 1 : 0});
    ents[j].push_back({i, kind, t, j < i ? 1 : 0});
  }
  int maxdeg = 0;
  for (int i = 0; i < N; ++i) {
    std::stable_sort(ents[i].begin(), ents[i].end(), [](const Ent &a, const Ent &b) {
      return a.j != b.j ? a.j < b.j : a.kind < b.kind;
    });
    maxdeg = std::max(maxdeg, (int)ents[i].size());
  }
  const Variant *var = nullptr;
  int MD = 0, SL = 0, SLE = 0;
  std::vector<uint32_t> meta;
  if (!is_block) {   // smallest compiled slot count that holds the busiest node
    for (const Variant &v : kVariants)
      if (v.K == d->k && v.maxdeg >= maxdeg && (!var || v.maxdeg < var->maxdeg) && (ad ? v.solve_anch != nullptr : v.solve != nullptr))
        var = &v;
    if (!var) is_block = true;   // a node busier than any wave variant: workgroup-per-problem path
  }
  if (is_block && ad) return fail("anchored templates need N * k <= 64 free unknowns and at most 20 terms per node");
  // workgroup-per-problem tables (BlockTabs)
  int n_clq = 0, Tc = T;
  const int ROWCAP = big ? 256 : BLOCK_MAXN;       // rows of the host-side tables (the workgroup kernels' are 128)
  std::vector<int> nc_term, clq_term, clq_pair_term, node_of_row(ROWCAP, -1), wave_sl(2 * BLOCK_WAVES, 0);
  std::vector<unsigned short> clq_pid;   // [M][512] compact pair id per (thread, partner), 0xffff = none
  // node-per-lane path: 3-D graphs beyond one wavefront, trust-region solver, theta = 1.
  // force_block_path: 0 = automatic, 1 = the workgroup kernels, 2 = the node-per-lane kernel
  if (d->force_block_path == 2) is_block = true;
  const bool npt_wanted = is_block && d->k == 3 && d->solver == GIK_SOLVER_TRUST_REGIONS && d->theta == 1.0 && !ad &&
                          (d->force_block_path == 2 ||
                           (d->force_block_path == 0 && d->N * d->k > WAVE && !getenv("GIK_NO_NPT")));   // (developer A/B switch, read once)
  bool npt_ok = false;
  const bool npt_two_waves = big || !(dbg_eff & 2048);      // 2048: one wavefront per problem, two nodes per lane
  const int npt_NW = big ? 4 : (npt_two_waves ? 2 : 1);     // wavefronts per problem ("two_waves": one node per lane)
  const int NPT_ROWS = big ? 4 * WAVE : NPT_MAXN;
  int npt_TL = 1, npt_DEG0 = 0, npt_DEG1 = 0, npt_n_wrows = 0, npt_cbase = 0, npt_n_rows = 0, npt_n_terms = 0, npt_term_sync = 0;
  std::vector<int> npt_node_of_row, npt_term_tgt, npt_pair_term;
  std::vector<uint32_t> npt_term_rec;
  std::vector<unsigned short> npt_gather;
  std::vector<unsigned char> npt_wslot, npt_prow;
  int npt_n_helped = 0;
  if (is_block) {
    // A rigid clique -- a set of nodes every pair of which is tied by an equality term (the
    // anchors of a scene with many obstacles) -- is taken out of the slot tables and handled in
    // closed form (gik_block.hip.h).  Greedy by equality degree; rows 0..n_clq-1 of the LDS
    // arrays are the clique's nodes in ascending order, the other nodes follow.
    std::vector<int> eqterm((size_t)N * N, -1), deg(N, 0), order(N), row_of(N);
    for (int t = 0; t < T; ++t) {
      const int i = d->term_i[t], j = d->term_j[t];
      if (d->term_kind[t] == GIK_TERM_EQ && eqterm[(size_t)i * N + j] < 0) {
        eqterm[(size_t)i * N + j] = eqterm[(size_t)j * N + i] = t;
        ++deg[i];
        ++deg[j];
      }
    }
    std::vector<char> in_clq(N, 0);
    const int clique_min = (dbg_eff & 64) ? 4 : 16;
    if (d->k == 3 && !(dbg_eff & 128) && d->clique_closed_form != GIK_CLIQUE_OFF) {
      for (int i = 0; i < N; ++i) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return deg[a] > deg[b]; });
      std::vector<int> A;
      for (int v : order) {
        bool all = true;
        for (int a : A) all = all && eqterm[(size_t)v * N + a] >= 0;
        if (all) A.push_back(v);
      }
      if ((int)A.size() >= clique_min) {
        n_clq = (int)A.size();
        for (int a : A) in_clq[a] = 1;
      }
    }
    // with a clique the other nodes take the LAST rows (128 - F ...): their threads, the only
    // ones with more than a slot or two, then sit in wavefronts that have no clique work
    int r = 0;
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1 && n_clq) r = ROWCAP - (N - n_clq);
      for (int i = 0; i < N; ++i)
        if ((in_clq[i] != 0) == (pass == 0)) {
          row_of[i] = r;
          node_of_row[r++] = i;
        }
    }
    // slot entries in row numbering, clique pairs left out; `term` = index in the LDS target table
    ents.assign(ROWCAP, {});
    for (int t = 0; t < T; ++t) {
      const int i = d->term_i[t], j = d->term_j[t], kind = d->term_kind[t];
      if (n_clq && kind == GIK_TERM_EQ && in_clq[i] && in_clq[j] && eqterm[(size_t)i * N + j] == t) continue;
      const int ri = row_of[i], rj = row_of[j], tc = (int)nc_term.size();
      nc_term.push_back(t);
      ents[ri].push_back({rj, kind, tc, ri < rj ? 1 : 0});
      ents[rj].push_back({ri, kind, tc, rj < ri ? 1 : 0});
    }
    Tc = (int)nc_term.size();
    for (auto &e : ents)
      std::stable_sort(e.begin(), e.end(), [](const Ent &a, const Ent &b) {
        return a.j != b.j ? a.j < b.j : a.kind < b.kind;
      });
    if (!big) {      // tables of the 512-thread workgroup kernels (128 rows)
    const int M = (n_clq + 3) / 4;
    clq_term.assign((size_t)std::max(M, 1) * BLOCK_NT, -1);
    for (int tid = 0; tid < BLOCK_NT; ++tid) {
      const int row = tid >> 2, part = tid & 3;
      for (int m = 0; m < M && row < n_clq; ++m) {
        const int j = 4 * m + part;
        if (j < n_clq && j != row)
          clq_term[(size_t)m * BLOCK_NT + tid] = eqterm[(size_t)node_of_row[row] * N + node_of_row[j]];
      }
    }
    // each clique pair once (its target is staged in LDS per problem) + the pair id of every
    // (thread, partner): ids fit 16 bits (at most 128 * 127 / 2 pairs)
    {
      std::vector<int> pid_of_term((size_t)T, -1);
      clq_pid.assign((size_t)std::max(M, 1) * BLOCK_NT, (unsigned short)0xffff);
      for (size_t q = 0; q < clq_term.size(); ++q) {
        const int term = clq_term[q];
        if (term < 0) continue;
        if (pid_of_term[term] < 0) {
          pid_of_term[term] = (int)clq_pair_term.size();
          clq_pair_term.push_back(term);
        }
        clq_pid[q] = (unsigned short)pid_of_term[term];
      }
    }
    // four threads per node; a node's equality terms are dealt to them in turn, then its hinge
    // terms continuing the rotation, so that both kinds spread evenly (the padded slot count of a
    // wavefront is the largest equality count plus the largest hinge count among its threads:
    // 3 + 2 -> 2 + 1 for the free nodes of the table scene).  Within a thread the equality terms
    // come first (slots [0, SLE_w): no kind decoding in the kernels) and the hinge terms last
    // (slots [SLE_w, SL_w)), with the bounds of the thread's wavefront w; unused slots are inert
    // padding (own node, kind 0, not owner)
    std::vector<std::vector<Ent>> eqs(BLOCK_NT), hinges(BLOCK_NT);
    for (int node = 0; node < BLOCK_MAXN; ++node) {
      int turn = 0;
      for (int pass = 0; pass < 2; ++pass)
        for (const Ent &en : ents[node])
          if ((en.kind == GIK_TERM_EQ) == (pass == 0))
            (pass == 0 ? eqs : hinges)[4 * node + (turn++ & 3)].push_back(en);
    }
    for (int tid = 0; tid < BLOCK_NT; ++tid) {
      const int w = tid / WAVE;
      wave_sl[2 * w] = std::max(wave_sl[2 * w], (int)eqs[tid].size());
      wave_sl[2 * w + 1] = std::max(wave_sl[2 * w + 1], (int)hinges[tid].size());
    }
    // A wavefront with few slots runs them as ONE kind-decoding loop over each thread's equality
    // terms followed by its hinge terms ({0, T_w}: T_w = most terms of any of its threads) when that
    // is shorter than the padded split loops (table scene: 1 + 1 -> 1 for the base / goal nodes,
    // 3 + 1 -> 3 for the free nodes; an iteration is two dependent LDS round trips).
    std::vector<char> merged(BLOCK_WAVES, 0);
    for (int w = 0; w < BLOCK_WAVES; ++w) {
      int tot = 0;
      for (int tid = w * WAVE; tid < (w + 1) * WAVE; ++tid)
        tot = std::max(tot, (int)(eqs[tid].size() + hinges[tid].size()));
      if (tot <= 8 && tot < wave_sl[2 * w] + wave_sl[2 * w + 1]) {
        merged[w] = 1;
        wave_sl[2 * w] = 0;
        wave_sl[2 * w + 1] = tot;
      } else {
        wave_sl[2 * w + 1] += wave_sl[2 * w];   // {SLE_w, SL_w}
      }
      SLE = std::max(SLE, wave_sl[2 * w]);
      SL = std::max(SL, wave_sl[2 * w + 1]);
    }
    meta.assign((size_t)std::max(SL, 1) * BLOCK_NT, 0);
    for (int tid = 0; tid < BLOCK_NT; ++tid) {
      const int node = tid >> 2, w = tid / WAVE, sle = merged[w] ? (int)eqs[tid].size() : wave_sl[2 * w];
      for (int s = 0; s < SL; ++s) {
        uint32_t m = meta_pack(node_of_row[node] >= 0 ? node : 0, 0, 0, 0);
        const std::vector<Ent> &src = s < sle ? eqs[tid] : hinges[tid];
        const int e = s < sle ? s : s - sle;
        if (e < (int)src.size()) m = meta_pack(src[e].j, src[e].term, src[e].kind, src[e].owner);
        meta[(size_t)s * BLOCK_NT + tid] = m;
      }
    }

    }
    // ---- node-per-lane tables (NptTabs, gik_npt.hip.h) ----
    // Two layouts.  Two wavefronts per problem, one node per lane (default): the nodes outside the
    // clique take the first rows, then the clique's nodes, those that carry slot terms first -- every
    // end node of a slot term then sits in wavefront 0, which evaluates the terms, and the
    // direction / term tables need no barrier of their own.  One wavefront, two nodes per lane
    // (debug_flags 2048): the clique's nodes take rows 0..n_clq-1, the others follow; nodes that carry
    // slot terms go to EVEN rows where possible, so that a lane's second node has few or none (its
    // gather list is as long as the busiest second node's).
    if (npt_wanted) {
      const int NSn = npt_two_waves ? 1 : 2, NTn = npt_NW * WAVE;
      std::vector<int> sdeg(N, 0);
      for (int t : nc_term) {
        ++sdeg[d->term_i[t]];
        ++sdeg[d->term_j[t]];
      }
      std::vector<int> cl_busy, cl_idle, others;
      for (int i = 0; i < N; ++i) {
        if (in_clq[i]) (sdeg[i] ? cl_busy : cl_idle).push_back(i);
        else others.push_back(i);
      }
      auto by_deg = [&](int a, int b) { return sdeg[a] > sdeg[b]; };
      std::stable_sort(cl_busy.begin(), cl_busy.end(), by_deg);
      std::stable_sort(others.begin(), others.end(), by_deg);
      // nrow[v]: row of node v in the point table; npt_node_of_row[t]: node held by thread slot t
      npt_node_of_row.assign(NPT_ROWS, -1);
      npt_prow.assign(NPT_ROWS, 0);
      std::vector<int> nrow(N, -1), nslot(N, -1);
      int n_rows = 0;
      npt_n_helped = 0;
      if (npt_two_waves) {
        npt_cbase = (int)others.size();
        int r = 0;
        for (int v : others) nrow[v] = r++;
        for (int v : cl_busy) nrow[v] = r++;
        for (int v : cl_idle) nrow[v] = r++;
        n_rows = r;
        // thread slots: the nodes that carry slot terms on the even lanes 0, 2, ... of wavefront 0, each with a
        // clique node WITHOUT slot terms next to it (its helper in the gather); everything else behind
        std::vector<int> busy(others.begin(), others.end());
        busy.insert(busy.end(), cl_busy.begin(), cl_busy.end());
        busy.erase(std::remove_if(busy.begin(), busy.end(), [&](int v) { return sdeg[v] == 0; }), busy.end());
        std::stable_sort(busy.begin(), busy.end(), by_deg);
        std::vector<char> placed(N, 0);
        int slot = 0;
        size_t ih = 0;
        const bool can_help = 2 * busy.size() <= (size_t)WAVE && cl_idle.size() >= busy.size();
        for (int v : busy) {
          npt_node_of_row[slot] = v;
          nslot[v] = slot++;
          placed[v] = 1;
          if (can_help) {
            const int h = cl_idle[ih++];
            npt_node_of_row[slot] = h;
            nslot[h] = slot++;
            placed[h] = 1;
          }
        }
        npt_n_helped = can_help ? (int)busy.size() : 0;
        for (int pass = 0; pass < 3; ++pass)
          for (int v : (pass == 0 ? others : (pass == 1 ? cl_busy : cl_idle)))
            if (!placed[v]) {
              npt_node_of_row[slot] = v;
              nslot[v] = slot++;
              placed[v] = 1;
            }
      } else {
        npt_cbase = 0;
        size_t ib = 0, ii = 0;
        for (int r = 0; r < n_clq; ++r) {
          const bool want_busy = (r & 1) == 0;
          int v;
          if ((want_busy && ib < cl_busy.size()) || ii >= cl_idle.size()) v = cl_busy[ib++];
          else v = cl_idle[ii++];
          nrow[v] = r;
        }
        const int start = (n_clq + 1) & ~1;
        const bool even_only = others.empty() || start + 2 * ((int)others.size() - 1) < NPT_ROWS;
        n_rows = n_clq;
        for (size_t q = 0; q < others.size(); ++q) {
          const int r = even_only ? start + 2 * (int)q : n_clq + (int)q;
          nrow[others[q]] = r;
          n_rows = r + 1;
        }
        for (int v = 0; v < N; ++v) {      // thread slot = row
          npt_node_of_row[nrow[v]] = v;
          nslot[v] = nrow[v];
        }
      }
      for (int v = 0; v < N; ++v) npt_prow[nslot[v]] = (unsigned char)nrow[v];
      npt_n_rows = (n_rows + 1) & ~1;
      // compact direction table: one row per node that carries slot terms
      npt_wslot.assign(NPT_ROWS, 255);
      int n_wrows = 0;
      npt_term_sync = 0;
      std::vector<int> wslot_of_node(N, 255);
      for (int t = 0; t < NPT_ROWS; ++t)
        if (npt_node_of_row[t] >= 0 && sdeg[npt_node_of_row[t]]) {
          wslot_of_node[npt_node_of_row[t]] = n_wrows;
          npt_wslot[t] = (unsigned char)n_wrows++;
          if (npt_two_waves && t >= WAVE) npt_term_sync = 1;
        }
      const int Tn = (int)nc_term.size();
      const int TLn = Tn <= 64 ? 1 : 4;
      npt_ok = Tn <= 64 * 4 && n_wrows <= 127;
      bool npt_lists_fit = true;
      if (npt_ok) {
        npt_TL = TLn;
        npt_term_rec.assign((size_t)TLn * WAVE, 0u);
        npt_term_tgt.assign((size_t)TLn * WAVE, -1);
        for (size_t q = 0; q < npt_term_rec.size(); ++q)   // padding: rows 0 / 0, kind 0, the zero direction row
          npt_term_rec[q] = ((uint32_t)n_wrows << 18) | ((uint32_t)n_wrows << 25);
        struct GEnt { int other, kind, slot, neg; };
        std::vector<std::vector<GEnt>> glist(NPT_ROWS);
        for (int q = 0; q < Tn; ++q) {
          const int t = nc_term[q], i = d->term_i[t], j = d->term_j[t], kind = d->term_kind[t];
          const int ri = nrow[i], rj = nrow[j];
          npt_term_rec[q] = (uint32_t)ri | ((uint32_t)rj << 8) | ((uint32_t)kind << 16) |
                            ((uint32_t)wslot_of_node[i] << 18) | ((uint32_t)wslot_of_node[j] << 25);
          npt_term_tgt[q] = t;
          // term slot q = u * 64 + lane: the order of nc_term (= the reference's edge order)
          glist[nslot[i]].push_back({j, kind, q, 0});
          glist[nslot[j]].push_back({i, kind, q, 1});
        }
        int deg[2] = {0, 0};
        for (int r = 0; r < NPT_ROWS; ++r)
          std::stable_sort(glist[r].begin(), glist[r].end(), [](const GEnt &a, const GEnt &b) {
            return a.other != b.other ? a.other < b.other : a.kind < b.kind;
          });
        for (int i = 0; i < npt_n_helped; ++i) {      // the second half of a busy node's list moves to its helper
          std::vector<GEnt> &own = glist[2 * i], &hlp = glist[2 * i + 1];
          const size_t keep = (own.size() + 1) / 2;
          hlp.assign(own.begin() + keep, own.end());
          own.resize(keep);
        }
        for (int r = 0; r < NPT_ROWS; ++r) {
          const int sl = NSn == 1 ? 0 : (r & 1);
          deg[sl] = std::max(deg[sl], (int)glist[r].size());
        }
        npt_DEG0 = deg[0];
        npt_DEG1 = deg[1];
        npt_lists_fit = deg[0] + deg[1] <= 16;      // NptCtx::NG packed words
        const unsigned short pad = (unsigned short)(2 * Tn);
        npt_gather.assign((size_t)std::max(1, deg[0] + deg[1]) * NTn, pad);
        for (int r = 0; r < NPT_ROWS; ++r) {
          const int thr = r / NSn, sl = r % NSn;
          for (size_t e = 0; e < glist[r].size(); ++e)
            npt_gather[(size_t)(sl ? deg[0] + (int)e : (int)e) * NTn + thr] =
                (unsigned short)((glist[r][e].slot << 1) | glist[r][e].neg);
        }
        npt_pair_term.clear();
        std::vector<int> node_at_row(NPT_ROWS, 0);
        for (int v = 0; v < N; ++v) node_at_row[nrow[v]] = v;
        for (int a = 0; a < n_clq; ++a)
          for (int b = a + 1; b < n_clq; ++b)
            npt_pair_term.push_back(eqterm[(size_t)node_at_row[npt_cbase + a] * N + node_at_row[npt_cbase + b]]);
        npt_n_wrows = n_wrows;
        npt_n_terms = Tn;
      }
      npt_ok = npt_ok && npt_lists_fit;
    }
  } else {
  MD = var->maxdeg;
  meta.assign((size_t)MD * WAVE, 0);
  for (int lane = 0; lane < WAVE; ++lane) {
    const bool active = lane < N * d->k;
    const int node = active ? lane / d->k : 0;
    const int comp = active ? lane % d->k : 0;
    for (int s = 0; s < MD; ++s) {
      // padding slot: this lane's own row (idle lanes: the all-zero dump row), kind none
      uint32_t m = meta_pack(active ? node : TILE_ROWS - 1, 0, 0, 0);
      if (active && s < (int)ents[node].size()) {
        const Ent &e = ents[node][s];
        m = meta_pack(e.j, e.term, e.kind, (comp == 0 && e.owner) ? 1 : 0);
      }
      meta[(size_t)s * WAVE + lane] = m;
    }
  }
  }
  if (meta.empty()) meta.assign(1, 0u);      // (graphs beyond 128 nodes: no slot table of the 512-thread kernels)
  gik_template *t = new gik_template();
  t->is_block = is_block;
  t->SL = SL;
  t->SLE = SLE;
  t->N = N;
  t->K = d->k;
  t->T = T;
  t->maxdeg = is_block ? SL : MD;
  t->variant = var;
  // The product form concerns the one-unknown-per-lane kernel only (every other kernel forms s = y . w per edge
  // anyway).  There the per-edge form exists for k = 3, TrustRegions, free-free graphs and is what GIK_HESS_AUTO
  // selects; an explicit GIK_HESS_PER_EDGE without such a kernel is refused.
  if (!is_block && d->k == 3 && d->hessian_form != GIK_HESS_COLUMN) {
    const bool have = !ad && d->solver == GIK_SOLVER_TRUST_REGIONS && var->solve_strict && var->solve_strict_theta;
    if (!have && d->hessian_form == GIK_HESS_PER_EDGE) {
      delete t;
      return fail("hessian_form = GIK_HESS_PER_EDGE: wavefront kernel of 3-D free-free graphs, TrustRegions only");
    }
    t->hess_per_edge = have;
  }
  t->p.mingradnorm = d->mingradnorm;
  t->p.theta = d->theta;
  t->p.kappa = d->kappa;
  t->p.rho_prime = d->rho_prime;
  t->p.rho_regularization = d->rho_regularization;
  t->p.maxiter = d->maxiter;
  t->p.maxinner = d->maxinner;
  t->p.mininner = d->mininner;
  t->p.planar_proj_exact = d->planar_proj_exact;
  t->solver = d->solver;
  t->cg.mingradnorm = d->mingradnorm;
  t->cg.minstepsize = d->cg_minstepsize;
  t->cg.orth_value = d->cg_orth_value;
  t->cg.maxiter = d->maxiter;
  t->cg.beta_type = d->cg_beta_type;
  t->cg.planar_proj_exact = d->planar_proj_exact;
  t->dbg = dbg_eff;
  t->wpc_override = std::max(0, d->waves_per_cu);
  // workgroup kernel, table scene, 4096 goals (round 3): slice 96 / 160 / 256 -> 1430 / 1430 / 1409 solves/s and
  // 755 / 586 / 368 MB of HBM traffic per launch (every resumed slice re-reads the problem's 45 KB of
  // targets; 207 MB are the algorithmic bytes).  Without slicing: ~15 % slower (round 2: 795 vs 929).
  t->slice_its = d->slice_outer_its < 0 ? 256 : d->slice_outer_its;
  // node-per-lane kernel, table scene, 4096 goals (round 4): slice 0 / 48 / 96 / 256 / 600 -> 1689 / 1896 / 1894 / 1861 /
  // 1774 solves/s (two problems per CU: 512 slots, a third of the requeues of the workgroup kernel)
  // HBM traffic per launch (PMC): 603 MB at 128 = 2.9 x the algorithmic 207 MB (every resume re-reads the problem's 45 KB of
  // clique targets); 192 is the compromise
  t->npt_slice_its = d->slice_outer_its < 0 ? 192 : d->slice_outer_its;
  // developer overrides, read once here (never inside a batch call)
  if (const char *e = getenv("GIK_WAVES_PER_CU")) t->wpc_override = std::max(1, atoi(e));
  // wavefront kernel: 256 ... 32 iterations per slice give the same time (NOTEBOOK 8.3); the longest of
  // them moves the fewest problems through HBM (KUKA 65536: 118 k hand-overs of ~1.5 KB instead of 562 k at 64)
  t->wave_slice_its = d->slice_outer_its < 0 ? 256 : d->slice_outer_its;
  t->wave_slice_auto = d->slice_outer_its < 0 && !getenv("GIK_SLICE");   // (an explicit length is taken literally)
  if (const char *e = getenv("GIK_SLICE")) t->slice_its = t->npt_slice_its = t->wave_slice_its = std::max(0, atoi(e));
  if (const char *e = getenv("GIK_SLICE_CYCLES")) t->wave_slice_cycles = std::max(0, atoi(e));
  t->d_slot_meta = nullptr;
  t->d_counters = nullptr;
  t->next_counter = 0;
  t->counter_slot.resize(kCounterRing);
  t->counter_ring = kCounterRing;
  if (const char *e = getenv("GIK_COUNTER_RING")) t->counter_ring = std::min(kCounterRing, std::max(1, atoi(e)));
  if (const char *e = getenv("GIK_SLICE_POOL")) t->slice_pool = std::min(gik_template::kSlicePool, std::max(1, atoi(e)));
  t->has_pipe = false;
  t->smem_bytes = is_block ? (d->k == 3 ? BlockCtx<3>::lds_bytes(Tc, SL, (int)clq_pair_term.size(), n_clq)
                                        : BlockCtx<2>::lds_bytes(Tc, SL))
                           : (ad ? var->lds_anch(T) : (t->hess_per_edge ? var->lds_strict(T) : var->lds(T)));
  if (is_block && t->smem_bytes > 160 * 1024) {
    delete t;
    return fail("graph too large for the LDS-resident block path");
  }
  const bool cg = d->solver == GIK_SOLVER_CONJUGATE_GRADIENT;
  const void *solve_kernel =
      is_block ? (cg ? (d->k == 3 ? (const void *)rcg_block_kernel<3> : (const void *)rcg_block_kernel<2>)
                     : (d->k == 3 ? (const void *)rtr_block_kernel<3> : (const void *)rtr_block_kernel<2>))
               : (const void *)(ad ? var->solve_anch : (cg ? var->solve_cg : (t->hess_per_edge ? (d->theta == 1.0 ? (var->solve_strict_mig ? var->solve_strict_mig : var->solve_strict) : var->solve_strict_theta) : var->solve)));      // (occupancy: of the build large batches run -- the small-batch build trades registers for latency, gik_rtr.hip.h SPLIT)
  if (is_block && t->smem_bytes > 48 * 1024) {
    // more than the default dynamic-LDS allowance: opt in for exactly what this template needs
    const void *fns[2] = {solve_kernel, d->k == 3 ? (const void *)kat_block_kernel<3>
                                                  : (const void *)kat_block_kernel<2>};
    for (const void *fn : fns) {
      if (raise_dynamic_lds(fn, t->smem_bytes) != hipSuccess) {
        (void)hipGetLastError();
        const std::string msg =
            "cannot reserve " + std::to_string(t->smem_bytes) + " bytes of LDS per workgroup";
        delete t;
        return fail(msg);
      }
    }
  }
  hipDeviceProp_t prop;
  int occ = 0;
  if (hipGetDevice(&t->device) != hipSuccess ||
      hipGetDeviceProperties(&prop, t->device) != hipSuccess ||
      hipMalloc((void **)&t->d_slot_meta, meta.size() * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc((void **)&t->d_counters, kCounterRing * sizeof(unsigned int)) != hipSuccess ||
      hipMemcpy(t->d_slot_meta, meta.data(), meta.size() * sizeof(uint32_t),
                hipMemcpyHostToDevice) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, solve_kernel, is_block ? BLOCK_NT : WAVE,
                                                   t->smem_bytes) != hipSuccess) {
    if (t->d_slot_meta) (void)hipFree(t->d_slot_meta);
    if (t->d_counters) (void)hipFree(t->d_counters);
    delete t;
    return fail("HIP device setup failed (no GPU?)");
  }
  t->n_cu = prop.multiProcessorCount;
  t->waves_per_cu = std::max(1, std::min(occ, 32));
  if (!is_block && !ad && !cg && d->k == 2 && N <= QUAD_NODES && var->maxdeg == 6 && d->theta == 1.0 &&
      !(d->debug_flags & 8192) && !getenv("GIK_NO_QUAD")) {
    t->quad_solve = rtr_quad_kernel<6>;
    t->quad_smem = QuadCtx<6>::lds_bytes();
    int qocc = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&qocc, (const void *)t->quad_solve, WAVE, t->quad_smem) == hipSuccess)
      t->quad_waves_per_cu = std::max(1, std::min(qocc, 32));
    t->quad_min_batch = 12 * t->n_cu;
    if (const char *e = getenv("GIK_QUAD_MIN_BATCH")) t->quad_min_batch = std::max(0, atoi(e));
  }
  if (ad) {
    // ---- fixed-anchor data ----
    bool ok = true;
    std::vector<double> tab(4 * ANCH_MAXA, 0.0);
    for (int r = 0; r < ad->n_anchor; ++r)
      for (int c = 0; c < 3; ++c) tab[r * 4 + c] = ad->anchor_pos[r * 3 + c];
    std::vector<uint32_t> pm((size_t)ANCH_PMAX * WAVE, 0u);
    std::vector<double> pt((size_t)ANCH_PMAX * WAVE, 0.0);
    std::vector<int> cnt(N, 0);
    for (int q = 0; q < ad->n_pin; ++q) {
      const int i = ad->pin_node[q], r = ad->pin_anchor[q], kind = ad->pin_kind[q];
      if (i < 0 || i >= N || r < 0 || r >= ad->n_anchor || kind < GIK_TERM_EQ || kind > GIK_TERM_UPPER || cnt[i] >= ANCH_PMAX) {
        ok = false;
        break;
      }
      for (int c = 0; c < 3; ++c) {      // every lane of the node walks all of the node's pinned terms
        pm[(size_t)cnt[i] * WAVE + i * 3 + c] = (uint32_t)r | ((uint32_t)kind << 8);
        pt[(size_t)cnt[i] * WAVE + i * 3 + c] = ad->pin_target[q];
      }
      ++cnt[i];
    }
    unsigned long long mask = 0;
    for (int i = 0; i < N && ad->obs_node_mask; ++i)
      if (ad->obs_node_mask[i]) mask |= 1ull << i;
    AnchArgs an;
    an.anch_const = ok ? upload(t, tab.data(), tab.size(), ok) : nullptr;
    an.pin_meta = upload(t, pm.data(), pm.size(), ok);
    an.pin_tgt = upload(t, pt.data(), pt.size(), ok);
    an.obs = ad->n_obs ? upload(t, ad->obs, (size_t)ad->n_obs * 4, ok) : nullptr;
    an.anchor_goal = nullptr;
    an.obs_mask = mask;
    an.n_obs = ad->n_obs;
    an.n_goal = ad->n_goal_anchor;
    an.goal_row0 = ad->n_anchor - ad->n_goal_anchor;
    t->an = an;
    t->d_targets_const = const_cast<double *>(upload(t, ad->term_target, (size_t)T, ok));
    t->d_free_full = const_cast<int *>(upload(t, ad->free_full_index, (size_t)N, ok));
    t->d_anchor_full = const_cast<int *>(upload(t, ad->anchor_full_index, (size_t)ad->n_anchor, ok));
    t->full_N = ad->full_N;
    t->n_anchor = ad->n_anchor;
    t->axis_length = ad->axis_length;
    t->anchored = true;
    if (hipEventCreate(&t->ev_solve0) != hipSuccess || hipEventCreate(&t->ev_solve1) != hipSuccess) ok = false;
    if (!ok) {
      gik_template_destroy(t);
      return fail("anchored templates: bad pinned term (node / anchor / kind out of range, or more than 8 per node) "
                  "or device upload failed");
    }
  }
  if (is_block) {
    bool ok = true;
    t->bt.nc_term = upload(t, nc_term.data(), nc_term.size(), ok);
    t->bt.clq_term = upload(t, clq_term.data(), clq_term.size(), ok);
    t->bt.clq_pair_term = upload(t, clq_pair_term.data(), clq_pair_term.size(), ok);
    t->bt.clq_pid_t = upload(t, clq_pid.data(), clq_pid.size(), ok);
    t->bt.n_pairs = (int)clq_pair_term.size();
    t->bt.node_of_row = upload(t, node_of_row.data(), node_of_row.size(), ok);
    t->bt.wave_sl = upload(t, wave_sl.data(), wave_sl.size(), ok);
    t->bt.Tc = Tc;
    t->bt.n_clq = n_clq;
    t->bt.clq_euclid = (n_clq && !(dbg_eff & 256) && d->clique_closed_form != GIK_CLIQUE_DENSE) ? 1 : 0;   // 256: always the dense D w product
    t->clique_mode = !n_clq ? GIK_CLIQUE_OFF : (t->bt.clq_euclid ? GIK_CLIQUE_AUTO : GIK_CLIQUE_DENSE);
    if (!ok) {
      gik_template_destroy(t);
      return fail("device upload of the workgroup-path tables failed");
    }
  }
  if (d->force_block_path == 2 && !npt_ok) {
    gik_template_destroy(t);
    return fail("node-per-lane kernel: k = 3, TrustRegions, theta = 1, at most 256 terms outside the rigid clique, at most 16 per lane");
  }
  if (npt_ok) {
    bool ok = true;
    t->nt.node_of_row = upload(t, npt_node_of_row.data(), npt_node_of_row.size(), ok);
    t->nt.clq_pair_term = upload(t, npt_pair_term.data(), npt_pair_term.size(), ok);
    t->nt.term_rec = upload(t, npt_term_rec.data(), npt_term_rec.size(), ok);
    t->nt.term_tgt = upload(t, npt_term_tgt.data(), npt_term_tgt.size(), ok);
    t->nt.gather = upload(t, npt_gather.data(), npt_gather.size(), ok);
    t->nt.wslot_of_row = upload(t, npt_wslot.data(), npt_wslot.size(), ok);
    t->nt.prow_of_slot = upload(t, npt_prow.data(), npt_prow.size(), ok);
    t->nt.n_helped = npt_n_helped;
    t->nt.n_clq = n_clq;
    t->nt.n_pairs = (int)npt_pair_term.size();
    t->nt.DEG0 = npt_DEG0;
    t->nt.DEG1 = npt_DEG1;
    t->nt.n_wrows = npt_n_wrows;
    t->nt.TL = npt_TL;
    t->nt.clq_euclid = t->bt.clq_euclid;
    t->nt.cbase = npt_cbase;
    t->nt.n_rows = npt_n_rows;
    t->nt.n_terms = npt_n_terms;
    t->nt.term_sync = npt_term_sync;
    for (const NptVariant &v : kNptVariants)
      if (v.TL == npt_TL && v.NW == npt_NW) t->npt_variant = &v;
    t->npt_smem = t->npt_variant->lds(t->nt.n_pairs, npt_n_wrows, npt_n_rows, npt_n_terms);
    const void *fns[2] = {(const void *)t->npt_variant->solve, (const void *)t->npt_variant->kat};
    int occ_npt = 0;
    ok = ok && t->npt_smem <= 160 * 1024;
    if (ok && t->npt_smem > 48 * 1024)
      for (const void *fn : fns) ok = ok && raise_dynamic_lds(fn, t->npt_smem) == hipSuccess;
    ok = ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_npt, fns[0], WAVE * t->npt_variant->NW, t->npt_smem) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      if (d->force_block_path == 2) {
        gik_template_destroy(t);
        return fail("node-per-lane kernel: device setup failed (LDS)");
      }
    } else {
      t->is_npt = true;
      t->npt_waves_per_cu = std::max(1, std::min(occ_npt, 4));   // problems (workgroups) per CU
    }
  }
  if (big && !t->is_npt) {
    gik_template_destroy(t);
    return fail("graphs of more than 128 nodes need the node-per-lane kernel: at most 256 terms outside the rigid clique "
                "(16 per node), at most 127 nodes that carry such terms");
  }
  if ((t->dbg & 32) && t->is_npt)
    fprintf(stderr, "  node-per-lane kernel: %d wavefront(s) per problem, TL=%d, %d slot terms (sync %d), %d direction rows, gather lists %d + %d, "
            "clique rows from %d, lds=%zu B, %d problems per CU\n",
            t->npt_variant->NW, t->nt.TL, t->nt.n_terms, t->nt.term_sync, t->nt.n_wrows, t->nt.DEG0, t->nt.DEG1, t->nt.cbase,
            t->npt_smem, t->npt_waves_per_cu);
  if (t->dbg & 32)
      fprintf(stderr, "gik_template_create: N=%d k=%d T=%d %s maxdeg=%d lds=%zu B occupancy=%d per CU, %d CUs; "
              "clique %d, slot terms %d, slots %d\n",
              t->N, t->K, t->T, is_block ? "block" : "wave", is_block ? 0 : t->variant->maxdeg,
              t->smem_bytes, occ, t->n_cu, n_clq, Tc, SL);
  if ((t->dbg & 32) && is_block) {
    fprintf(stderr, "  slot loop bounds per wavefront {equalities, all}:");
    for (int w = 0; w < BLOCK_WAVES; ++w) fprintf(stderr, " {%d, %d}", wave_sl[2 * w], wave_sl[2 * w + 1]);
    fprintf(stderr, "\n");
  }
  *out = t;
  return 0;
}

void gik_template_destroy(gik_template *t) {
  if (!t) return;
  if (t->d_slot_meta) (void)hipFree(t->d_slot_meta);
  if (t->d_counters) (void)hipFree(t->d_counters);
  for (void *p : t->pipe_allocs) (void)hipFree(p);
  if (t->ev_solve0) (void)hipEventDestroy(t->ev_solve0);
  if (t->ev_solve1) (void)hipEventDestroy(t->ev_solve1);
  if (t->prep_done) (void)hipEventDestroy(t->prep_done);
  for (auto &c : t->counter_slot)
    if (c.done) (void)hipEventDestroy(c.done);
  for (auto &w : t->slice_ws) {
    if (w.base) (void)hipFree(w.base);
    if (w.done) (void)hipEventDestroy(w.done);
  }
  delete t;
}

int gik_pipeline_attach(gik_template *t, const gik_pipeline_desc *d) {
  using namespace gik;
  if (!t || !d) return fail("null argument");
  if (t->has_pipe) return fail("pipeline already attached");
  const int N = t->N, K = t->K, n = d->n_joints;
  if (n < 1 || n > 125) return fail("n_joints out of range (1 .. 125)");
  if (d->n_anchor < 1 || d->n_anchor > PREP_MAXA) return fail("n_anchor must be in [1, 256]");
  if (N > PREP_BIGN - 1 || (N > PREP_MAXN && K != 3))
    return fail("the device pipeline handles graphs of up to 128 nodes (3-D: 255)");
  if (!d->T0 || !d->p_index || !d->base_lower || !d->base_upper || !d->anchor_index ||
      !d->anchor_pos || !d->pair_i || !d->pair_j || !d->term_src || !d->term_static)
    return fail("null pipeline array");
  if (K == 3 && !d->q_index) return fail("q_index required for k=3");
  bool ok = true;
  PipeConst pc;
  const int DD = (K + 1) * (K + 1);
  pc.T0 = upload(t, d->T0, (size_t)(n + 1) * DD, ok);
  pc.p_idx = upload(t, d->p_index, n + 1, ok);
  pc.q_idx = (K == 3) ? upload(t, d->q_index, n + 1, ok) : pc.p_idx;
  pc.base_lower = upload(t, d->base_lower, (size_t)N * N, ok);
  pc.base_upper = upload(t, d->base_upper, (size_t)N * N, ok);
  pc.anchor_idx = upload(t, d->anchor_index, d->n_anchor, ok);
  pc.anchor_pos = upload(t, d->anchor_pos, (size_t)d->n_anchor * K, ok);
  pc.pair_i = upload(t, d->pair_i, d->n_pairs, ok);
  pc.pair_j = upload(t, d->pair_j, d->n_pairs, ok);
  pc.term_src = upload(t, d->term_src, t->T, ok);
  pc.term_static = upload(t, d->term_static, t->T, ok);
  if (!ok) return fail("device upload failed");
  pc.N = N;
  pc.K = K;
  pc.T = t->T;
  pc.n_anchor = d->n_anchor;
  pc.n_pairs = d->n_pairs;
  pc.n_joints = n;
  pc.goal0 = d->goal_node0;
  pc.goal1 = d->goal_node1;
  // end effectors: a chain is one path 0..n with goal nodes (goal_node0, goal_node1)
  const int n_ee = d->n_ee > 1 ? d->n_ee : 1;
  if (n_ee > PREP_MAX_EE) return fail("at most 8 end effectors");
  if (n_ee > 1 && (!d->ee_goal_nodes || !d->ee_path || d->n_goal_pairs < 0 || (K == 2 && !d->ee_goal_len) ||
                   (d->n_goal_pairs > 0 && (!d->goal_pair_a || !d->goal_pair_b))))
    return fail("several end effectors: ee_goal_nodes / ee_path / goal pairs (k = 2: ee_goal_len) required");
  pc.n_ee = n_ee;
  pc.n_gg = n_ee > 1 ? d->n_goal_pairs : 0;
  std::vector<int> path((size_t)n_ee * (n + 1), -1);
  bool inert_goal_slot = false;
  for (int e = 0; e < PREP_MAX_EE; ++e) pc.ee_len[e] = (n_ee > 1 && K == 2 && e < n_ee) ? d->ee_goal_len[e] : d->goal_len;
  if (n_ee > 1) {
    for (int g = 0; g < 2 * n_ee; ++g) {
      pc.goal_node[g] = d->ee_goal_nodes[g];
      // -1: a planar tree's parent node that an earlier end effector's pose pins already (odd slots only)
      if (pc.goal_node[g] < 0 && !(K == 2 && (g & 1))) return fail("bad ee_goal_nodes");
      if (pc.goal_node[g] >= N) return fail("bad ee_goal_nodes");
      inert_goal_slot = inert_goal_slot || pc.goal_node[g] < 0;
    }
    for (size_t t = 0; t < path.size(); ++t) path[t] = d->ee_path[t];
    for (int e = 0; e < n_ee; ++e)
      for (int k = 0; k <= n; ++k) {
        const int j = path[(size_t)e * (n + 1) + k];
        if (j < -1 || j > n || (k == 0 && j != 0)) return fail("bad ee_path");
      }
  } else {
    pc.goal_node[0] = d->goal_node0;
    pc.goal_node[1] = d->goal_node1;
    for (int k = 0; k <= n; ++k) path[k] = k;
  }
  pc.ee_path = upload(t, path.data(), path.size(), ok);

  pc.gg_a = pc.n_gg ? upload(t, d->goal_pair_a, pc.n_gg, ok) : nullptr;
  pc.gg_b = pc.n_gg ? upload(t, d->goal_pair_b, pc.n_gg, ok) : nullptr;
  if (2 * n_ee * d->n_anchor + pc.n_gg > 2 * PREP_MAXA + 16) return fail("too many anchor-goal pairs");
  if (!ok) return fail("device upload failed");
  pc.x_idx = d->x_index;
  pc.y_idx = d->y_index;
  pc.goal_len = d->goal_len;
  pc.axis_length = d->axis_length;
  pc.last_along_z = d->last_link_along_z;
  t->pc = pc;
  t->sweeps = d->jacobi_sweeps > 0 ? d->jacobi_sweeps : 10;
  t->prep_smem = sizeof(double) * ((size_t)5 * N * N + 2 * n_ee * d->n_anchor + pc.n_gg + 96) +
                 sizeof(int) * (48 + (size_t)((N + 1) / 2) * ((N | 1) + (N + 1) / 2 + 1));   // (+ jacobi_lds's pair tables)
  // graphs beyond one wavefront's LDS: workgroup-per-goal kernel with its matrices in a global slab
  t->prep_block = N > 32 || d->n_anchor > 32 || d->force_block_prepare != 0 ||
                  getenv("GIK_PREP_FORCE_BLOCK") != nullptr;
  t->prep_no_compress = getenv("GIK_PREP_NO_COMPRESS") != nullptr;   // (developer A/B switch, read once, here)
  t->prep_big = N > PREP_MAXN;      // graphs of 129 .. 255 nodes: prep_block_kernel<false, PREP_BIGN>, six matrices per slab
  if (t->prep_block) {
    int occ = 0;
    // work matrix in LDS when it fits next to the kernel's static arrays (N <= 123), one workgroup per CU
    t->prep_a_lds = false;
    const size_t a_bytes = sizeof(double) * (size_t)N * N;
    if (t->prep_big) {
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, prep_block_kernel<false, PREP_BIGN>, PREP_NT, 0) != hipSuccess)
        occ = 1;
    } else if (!getenv("GIK_PREP_A_GLOBAL") &&
        raise_dynamic_lds((const void *)prep_block_kernel<true>, a_bytes) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, prep_block_kernel<true>, PREP_NT, a_bytes) == hipSuccess &&
        occ >= 1)
      t->prep_a_lds = true;
    else
      (void)hipGetLastError();
    if (!t->prep_a_lds && !t->prep_big &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, prep_block_kernel<false>, PREP_NT, 0) != hipSuccess)
      occ = 1;
    occ = std::max(1, std::min(occ, 2));   // 5 N^2 doubles per workgroup: keep the slabs cache-resident
    if (const char *e = getenv("GIK_PREP_WAVES_PER_CU")) occ = std::max(1, atoi(e));
    t->prep_waves_per_cu = occ;
    const size_t bytes = sizeof(double) * (t->prep_big ? 6 : 5) * (size_t)N * N * (size_t)t->n_cu * occ;
    void *ws = nullptr;
    if (hipMalloc(&ws, bytes) != hipSuccess) return fail("cannot allocate the prepare workspace");
    t->pipe_allocs.push_back(ws);
    t->prep_ws = static_cast<double *>(ws);
    if (hipEventCreateWithFlags(&t->prep_done, hipEventDisableTiming) != hipSuccess)
      return fail("hipEventCreate failed");
  } else {
    // the prepare kernel is latency-bound (dependent Jacobi chains, LDS round trips): run as many
    // resident waves per CU as its registers and LDS allow (a fixed 8 per CU left half of them
    // unused: planar-10 prepare 2.3 -> see NOTEBOOK 4.2)
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, prep_wave_kernel, WAVE, t->prep_smem) != hipSuccess)
      occ = 8;
    if (const char *e = getenv("GIK_PREP_WAVES_PER_CU")) occ = atoi(e);   // developer override
    t->prep_waves_per_cu = std::max(1, std::min(occ, 32));
    if (N <= PREPQ_MAXN && !inert_goal_slot && !getenv("GIK_NO_PREP_QUAD")) {   // (inert slots: prep_wave_kernel skips them)
      t->prep_quad_smem = prep_quad_lds_bytes(N, 2 * n_ee * d->n_anchor + pc.n_gg);
      int qocc = 0;
      if (t->prep_quad_smem <= 40 * 1024 &&
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&qocc, N == 13 ? prep_quad_kernel<13> : prep_quad_kernel<0>, WAVE,
                                                       t->prep_quad_smem) == hipSuccess &&
          qocc >= 1) {
        t->prep_quad = true;
        if (const char *e = getenv("GIK_PREP_WAVES_PER_CU")) qocc = atoi(e);
        t->prep_quad_waves_per_cu = std::max(1, std::min(qocc, 32));
      } else {
        (void)hipGetLastError();
      }
    }
  }
  t->has_pipe = true;
  return 0;
}

int gik_prepare_batch(const gik_template *t, const double *d_T_goal, int B, double *d_targets,
                      double *d_Y_init, int32_t *d_K_out, void *stream) {
  return gik_prepare_batch_debug(t, d_T_goal, B, d_targets, d_Y_init, d_K_out, nullptr, stream);
}

int gik_prepare_batch_debug(const gik_template *t, const double *d_T_goal, int B, double *d_targets,
                            double *d_Y_init, int32_t *d_K_out, const gik_prepare_diag *diag,
                            void *stream) {
  using namespace gik;
  if (!t || B < 0) return fail("bad argument");
  if (!t->has_pipe) return fail("no pipeline attached (gik_pipeline_attach)");
  if (B == 0) return 0;
  if (!d_T_goal || !d_targets || !d_Y_init) return fail("null buffer");
  if (t->prep_block && capturing_stream(stream))      // (the workgroup variant chains its launches by an event)
    return fail("gik_prepare_batch: the stream is capturing (hipStreamBeginCapture); batch calls cannot be captured into a graph");
  PrepArgs a;
  a.pc = t->pc;
  a.T_goal = d_T_goal;
  a.targets = d_targets;
  a.Y_init = d_Y_init;
  a.K_out = d_K_out;
  a.dbg_lb = diag ? diag->d_lb : nullptr;
  a.dbg_ub = diag ? diag->d_ub : nullptr;
  a.dbg_eig = diag ? diag->d_eig : nullptr;
  if ((a.dbg_lb == nullptr) != (a.dbg_ub == nullptr)) return fail("d_lb and d_ub go together");
  a.B = B;
  a.sweeps = t->sweeps;
  a.stop_phase = 0;
  a.no_compress = t->prep_no_compress ? 1 : 0;
#ifdef GIK_DEV
  if (const char *e = getenv("GIK_PREP_STOP")) a.stop_phase = atoi(e);   // developer build: timing of the phases
#endif
  const int grid = std::min(B, t->n_cu * t->prep_waves_per_cu);
  if (t->prep_block) {
    gik_template *mt = const_cast<gik_template *>(t);   // the workspace hand-over is the mutable part
    std::lock_guard<std::mutex> lock(mt->prep_mutex);
    if (mt->prep_pending) HIP_OK(hipStreamWaitEvent((hipStream_t)stream, mt->prep_done, 0));
    if (t->prep_big)
      hipLaunchKernelGGL((prep_block_kernel<false, PREP_BIGN>), dim3(grid), dim3(PREP_NT), 0, (hipStream_t)stream, a,
                         t->prep_ws);
    else if (t->prep_a_lds)
      hipLaunchKernelGGL(prep_block_kernel<true>, dim3(grid), dim3(PREP_NT),
                         sizeof(double) * (size_t)t->N * t->N, (hipStream_t)stream, a, t->prep_ws);
    else
      hipLaunchKernelGGL(prep_block_kernel<false>, dim3(grid), dim3(PREP_NT), 0, (hipStream_t)stream, a,
                         t->prep_ws);
    HIP_OK(hipEventRecord(mt->prep_done, (hipStream_t)stream));
    mt->prep_pending = true;
  } else if (t->prep_quad && !a.dbg_lb && !a.dbg_eig) {
    // (the diagnostics -- bounds and spectra of gik_prepare_batch_debug -- come from the one-goal-per-wavefront kernel)
    const int qgrid = std::min((B + QUAD_SLOTS - 1) / QUAD_SLOTS, t->n_cu * t->prep_quad_waves_per_cu);
    hipLaunchKernelGGL(t->N == 13 ? prep_quad_kernel<13> : prep_quad_kernel<0>, dim3(qgrid), dim3(WAVE), t->prep_quad_smem,
                       (hipStream_t)stream, a);
  } else
    hipLaunchKernelGGL(prep_wave_kernel, dim3(grid), dim3(WAVE), t->prep_smem, (hipStream_t)stream,
                       a);
  HIP_OK(hipGetLastError());
  return 0;
}

int gik_recover_batch(const gik_template *t, const double *d_Y, const double *d_T_goal, int B,
                      double *d_q, double *d_pos_err, double *d_rot_err, void *stream) {
  using namespace gik;
  if (!t || B < 0) return fail("bad argument");
  if (!t->has_pipe) return fail("no pipeline attached (gik_pipeline_attach)");
  if (B == 0) return 0;
  if (!d_Y || !d_T_goal || !d_q || !d_pos_err || !d_rot_err) return fail("null buffer");
  RecoverArgs a;
  a.pc = t->pc;
  a.Y = d_Y;
  a.T_goal = d_T_goal;
  a.q = d_q;
  a.pos_err = d_pos_err;
  a.rot_err = d_rot_err;
  a.B = B;
  hipLaunchKernelGGL(recover_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, a);
  HIP_OK(hipGetLastError());
  return 0;
}

int gik_ik_batch(const gik_template *t, const double *d_T_goal, int B, double *d_targets,
                 double *d_Y, gik_stats *d_stats, double *d_q, double *d_pos_err,
                 double *d_rot_err, void *stream) {
  int rc = gik_prepare_batch(t, d_T_goal, B, d_targets, d_Y, nullptr, stream);
  if (rc) return rc;
  rc = gik_solve_batch(t, d_Y, d_targets, B, d_Y, d_stats, nullptr, stream);
  if (rc) return rc;
  return gik_recover_batch(t, d_Y, d_T_goal, B, d_q, d_pos_err, d_rot_err, stream);
}

size_t gik_anchored_ws_doubles(const gik_template *anch, const gik_template *base, int B) {
  if (!anch || !base || !anch->anchored || B < 0) return 0;
  return (size_t)B * ((size_t)base->T + (size_t)base->N * 3 + (size_t)anch->N * 3 + (size_t)anch->an.n_goal * 3);
}

int gik_anchored_ik_batch(const gik_template *anch, const gik_template *base, const double *d_T_goal, int B,
                          double *d_ws, double *d_Y_full, gik_stats *d_stats, double *d_q,
                          double *d_pos_err, double *d_rot_err, void *stream) {
  using namespace gik;
  if (!anch || !base || !anch->anchored || B < 0) return fail("bad argument");
  if (!base->has_pipe || base->K != 3 || base->N != anch->full_N)
    return fail("the base template must be the robot graph (full_N nodes) with its pipeline attached");
  if (B == 0) return 0;
  if (!d_T_goal || !d_ws || !d_Y_full || !d_stats || !d_q || !d_pos_err || !d_rot_err) return fail("null buffer");
  double *tg_base = d_ws;
  double *Y_full0 = tg_base + (size_t)B * base->T;
  double *Y_free = Y_full0 + (size_t)B * base->N * 3;
  double *goal = Y_free + (size_t)B * anch->N * 3;
  // initial point of the robot graph (bound smoothing + MDS, obstacles play no part in it) ...
  int rc = gik_prepare_batch(base, d_T_goal, B, tg_base, Y_full0, nullptr, stream);
  if (rc) return rc;
  AnchGlueArgs g;
  g.T_goal = d_T_goal;
  g.Y_full_in = Y_full0;
  g.Y_free = Y_free;
  g.anchor_goal = goal;
  g.Y_full_out = d_Y_full;
  g.anch_const = anch->an.anch_const;
  g.free_full = anch->d_free_full;
  g.anchor_full = anch->d_anchor_full;
  g.B = B;
  g.Nf = anch->N;
  g.full_N = anch->full_N;
  g.n_anchor = anch->n_anchor;
  g.n_goal = anch->an.n_goal;
  g.goal_row0 = anch->an.goal_row0;
  g.axis_length = anch->axis_length;
  // ... mapped onto the world frame by its anchors; free rows = anchored start point
  hipLaunchKernelGGL(anch_init_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, g);
  HIP_OK(hipGetLastError());
  // timing events around the solve (diagnostics only; created with the handle).  The pair is
  // recorded under a lock so that gik_anchored_last_solve_ms never reads a half-recorded pair;
  // with concurrent callers it reports whichever call recorded last.
  gik_template *ma = const_cast<gik_template *>(anch);
  {
    std::lock_guard<std::mutex> lock(ma->ev_mutex);
    (void)hipEventRecord(ma->ev_solve0, (hipStream_t)stream);
  }
  rc = gik_solve_batch(anch, Y_free, goal, B, Y_free, d_stats, nullptr, stream);     // (not under ev_mutex: other
  if (rc) return rc;                                                                // threads launch meanwhile)
  {
    std::lock_guard<std::mutex> lock(ma->ev_mutex);
    (void)hipEventRecord(ma->ev_solve1, (hipStream_t)stream);
  }
  hipLaunchKernelGGL(anch_gather_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, g);
  HIP_OK(hipGetLastError());
  return gik_recover_batch(base, d_Y_full, d_T_goal, B, d_q, d_pos_err, d_rot_err, stream);
}

double gik_anchored_last_solve_ms(const gik_template *anch) {
  if (!anch || !anch->ev_solve0) return -1.0;
  float ms = -1.0f;
  hipEvent_t e0, e1;
  {   // the handles under the lock (a record in another thread is not interleaved with this read), the wait outside it
    std::lock_guard<std::mutex> lock(const_cast<gik_template *>(anch)->ev_mutex);
    e0 = anch->ev_solve0;
    e1 = anch->ev_solve1;
  }
  if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
    (void)hipGetLastError();      // (a pair recorded by two different concurrent calls has no defined duration)
    return -1.0;
  }
  return (double)ms;
}

static int launch_kat(const gik_template *t, int mode, const double *d_Y, const double *d_W,
                      const double *d_targets, int B, double *d_out, void *stream,
                      double *d_out_f = nullptr) {
  using namespace gik;
  if (!t || B < 0) return fail("bad argument");
  if (B == 0) return 0;
  if (!d_Y || !d_out) return fail("null buffer");
  KatArgs a;
  a.out_f = d_out_f;
  a.slot_meta = t->d_slot_meta;
  a.bt = t->bt;
  a.targets = d_targets;
  a.Y = d_Y;
  a.W = d_W;
  a.out = d_out;
  a.N = t->N;
  a.T = t->T;
  a.B = B;
  a.mode = mode;
  a.planar_proj_exact = t->p.planar_proj_exact;
  if (t->anchored && mode == 3) {          // the anchors fix the gauge: proj is the identity
    HIP_OK(hipMemcpyAsync(d_out, d_W, sizeof(double) * (size_t)B * t->N * t->K, hipMemcpyDeviceToDevice,
                          (hipStream_t)stream));
    return 0;
  }
  if (t->anchored) {
    a.an = t->an;
    a.an.anchor_goal = d_targets;          // (anchored templates: the per-problem input is the goal anchors)
    a.targets = t->d_targets_const;
    hipLaunchKernelGGL(t->variant->kat_anch, dim3(B), dim3(WAVE), t->smem_bytes, (hipStream_t)stream, a);
    HIP_OK(hipGetLastError());
    return 0;
  }
  if (t->quad_solve && (t->dbg & 16384) && !(t->dbg & (1 | 8192))) {
    hipLaunchKernelGGL(kat_quad_kernel<6>, dim3((B + QUAD_SLOTS - 1) / QUAD_SLOTS), dim3(WAVE), t->quad_smem,
                       (hipStream_t)stream, a);
  } else if (t->is_npt) {
    a.nt = t->nt;
    // (graphs beyond 128 nodes: a stream-ordered scratch for the clique target triangles of this call's workgroups;
    //  these one-call-at-a-time entry points are the known-answer interface, not the batch path)
    const size_t ctg = t->npt_variant->ctg(t->nt.n_pairs) * sizeof(double) * (size_t)B;
    void *ws = nullptr;
    if (ctg) HIP_OK(hipMallocAsync(&ws, ctg, (hipStream_t)stream));
    a.npt_ctg_ws = static_cast<double *>(ws);
    hipLaunchKernelGGL(t->npt_variant->kat, dim3(B), dim3(WAVE * t->npt_variant->NW), t->npt_smem, (hipStream_t)stream, a);
    if (ws) HIP_OK(hipFreeAsync(ws, (hipStream_t)stream));
  } else if (t->is_block) {
    if (t->K == 3)
      hipLaunchKernelGGL(kat_block_kernel<3>, dim3(B), dim3(BLOCK_NT), t->smem_bytes,
                         (hipStream_t)stream, a, t->SL);
    else
      hipLaunchKernelGGL(kat_block_kernel<2>, dim3(B), dim3(BLOCK_NT), t->smem_bytes,
                         (hipStream_t)stream, a, t->SL);
  } else {
    hipLaunchKernelGGL(t->hess_per_edge ? t->variant->kat_strict : t->variant->kat, dim3(B), dim3(WAVE), t->smem_bytes,
                       (hipStream_t)stream, a);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

int gik_cost(const gik_template *t, const double *d_Y, const double *d_targets, int B,
             double *d_f, void *stream) {
  if (!d_targets) return gik::fail("targets required");
  return launch_kat(t, 0, d_Y, nullptr, d_targets, B, d_f, stream);
}
int gik_grad(const gik_template *t, const double *d_Y, const double *d_targets, int B,
             double *d_out, void *stream) {
  if (!d_targets) return gik::fail("targets required");
  return launch_kat(t, 1, d_Y, nullptr, d_targets, B, d_out, stream);
}
int gik_cost_and_grad(const gik_template *t, const double *d_Y, const double *d_targets, int B,
                      double *d_f, double *d_grad, void *stream) {
  if (!d_targets) return gik::fail("targets required");
  if (B > 0 && !d_f) return gik::fail("null buffer");
  return launch_kat(t, 4, d_Y, nullptr, d_targets, B, d_grad, stream, d_f);
}
int gik_hess(const gik_template *t, const double *d_Y, const double *d_W,
             const double *d_targets, int B, double *d_out, void *stream) {
  if (!d_targets || !d_W) return gik::fail("targets and W required");
  return launch_kat(t, 2, d_Y, d_W, d_targets, B, d_out, stream);
}
int gik_proj(const gik_template *t, const double *d_Y, const double *d_Z, int B, double *d_out,
             void *stream) {
  if (!d_Z) return gik::fail("Z required");
  return launch_kat(t, 3, d_Y, d_Z, nullptr, B, d_out, stream);
}

int gik_solve_batch(const gik_template *t, const double *d_Y_init, const double *d_targets,
                    int B, double *d_Y_out, gik_stats *d_stats, const gik_trace *trace,
                    void *stream) {
  using namespace gik;
  if (!t || B < 0) return fail("bad argument");
  if (B == 0) return 0;
  if (!d_Y_init || !d_targets || !d_Y_out || !d_stats) return fail("null buffer");
  SolveArgs a;
  a.slot_meta = t->d_slot_meta;
  a.bt = t->bt;
  a.targets = d_targets;
  a.Y_init = d_Y_init;
  a.Y_out = d_Y_out;
  a.stats = d_stats;
  a.has_trace = (trace && trace->cap > 0) ? 1 : 0;
  if (a.has_trace)
    a.trace = *trace;
  else
    std::memset(&a.trace, 0, sizeof(a.trace));
  a.N = t->N;
  a.T = t->T;
  a.B = B;
  a.p = t->p;
  a.cg = t->cg;
  if (t->anchored) {
    a.an = t->an;
    a.an.anchor_goal = d_targets;          // anchored templates: per-problem goal anchors [B][n_goal*3]
    a.targets = t->d_targets_const;
  }
  a.dbg = t->dbg;
  a.dbg_buf = nullptr;
#ifdef GIK_DEV
  if (a.dbg & (4 | 8 | 4096)) {
    static double *buf = nullptr;
    if (!buf) (void)hipMalloc((void **)&buf, (1 << 20) * sizeof(double));
    (void)hipMemset(buf, 0, (1 << 20) * sizeof(double));
    a.dbg_buf = buf;
    g_dbg_buf = buf;
  }
#endif
  // The handle's mutable parts: the ring of work-queue heads and the time-slicing workspaces.  Both
  // are handed out under call_mutex, held until the event that guards their reuse is recorded, so
  // concurrent calls on one handle (any number of host threads and streams) are safe.
  gik_template *mt = const_cast<gik_template *>(t);
  // Hand-out of a slot: the one used last if the launch that used it has completed (hipEventQuery) -- a
  // sequence of calls then keeps ONE workspace warm instead of growing all of the pool --, else the next one
  // that no other call holds.  The lock covers the hand-out only: waiting for a slot's previous user, growing
  // its workspace (hipEventSynchronize / hipFree / hipMalloc) and the launch happen outside it, on a slot
  // marked in_use.
  // Stream capture is refused: the slot protocol below queries, waits for and records events, grows workspaces
  // (hipMalloc / hipFree) and resets a counter the kernel consumes -- none of which may happen under capture, and a
  // replayed graph would reuse this call's counter slot and workspace behind the library's back.  A batch is ONE
  // persistent launch; there is no launch overhead for a graph to remove.
  if (capturing_stream(stream)) return fail("gik_solve_batch: the stream is capturing (hipStreamBeginCapture); batch calls cannot be captured into a graph");
  auto take = [&](auto &slots, unsigned &next, unsigned n) -> int {
    for (;;) {
      {
        std::lock_guard<std::mutex> lock(mt->call_mutex);
        const unsigned last = (next + n - 1) % n;
        auto &ls = slots[last];
        if (!ls.in_use && ls.done && (!ls.pending || hipEventQuery(ls.done) == hipSuccess)) {
          ls.pending = false;
          ls.in_use = true;
          return (int)last;
        }
        (void)hipGetLastError();      // (hipErrorNotReady of the query)
        for (unsigned k = 0; k < n; ++k) {
          const unsigned i = (next + k) % n;
          if (!slots[i].in_use) {
            slots[i].in_use = true;
            next = (i + 1) % n;
            return (int)i;
          }
        }
      }
      std::this_thread::yield();      // every slot is between hand-out and launch in some other thread
    }
  };
  auto give_back = [&](auto &slot) {
    std::lock_guard<std::mutex> lock(mt->call_mutex);
    slot.pending = true;
    slot.in_use = false;
  };
  gik_template::CounterSlot &cs = mt->counter_slot[take(mt->counter_slot, mt->next_counter, (unsigned)t->counter_ring)];
  struct Release {      // error paths hand the slots back too (no launch: nothing pending)
    gik_template *mt;
    gik_template::CounterSlot *cs;
    gik_template::SliceWs *sw = nullptr;
    bool launched = false;
    ~Release() {
      std::lock_guard<std::mutex> lock(mt->call_mutex);
      cs->in_use = false;
      if (launched) cs->pending = true;
      if (sw) {
        sw->in_use = false;
        if (launched) sw->pending = true;
      }
    }
  } release{mt, &cs};
  a.work_counter = t->d_counters + (&cs - mt->counter_slot.data());
  if (!cs.done && hipEventCreateWithFlags(&cs.done, hipEventDisableTiming) != hipSuccess)
    return fail("hipEventCreate failed");
  if (cs.pending) HIP_OK(hipStreamWaitEvent((hipStream_t)stream, cs.done, 0));   // ring wrapped: previous user first
  HIP_OK(hipMemsetAsync(a.work_counter, 0, sizeof(unsigned int), (hipStream_t)stream));
  // Persistent waves per CU: as many as fit (two per SIMD at 249 VGPRs).  Two waves share a SIMD's
  // fp64 pipe and each runs 20-50 % slower than alone, which used to cost small batches -- whose
  // time is that of their slowest problem -- more than the extra throughput returned; with the age
  // priority of rtr_solve_one the old problems keep a lone wave's speed next to a young neighbour
  // (kernel ms at 1 / 2 waves per SIMD without, and 2 per SIMD with priorities -- LWA4D B=4096:
  // 116.9 / 132.4 / 116.3, B=16384: 190.6 / 196.7 / 187.7; KUKA B=8192: 182.8 / 156.4 / 155.4,
  // B=65536: 755.7 / 545.5 / 544.2).
  // Small batches still get one wave per SIMD: their time is the run time of the few problems that
  // go to maxiter, and two of THOSE on one SIMD (equal priority) slow each other down -- at 4096
  // LWA4D goals a third of the launches drew such a pair (128 instead of 116 ms).
  int wpc = t->is_npt ? t->npt_waves_per_cu : t->waves_per_cu;
  if (!t->is_block && t->K == 3 && wpc > 4 && (long long)B <= 6LL * 4 * t->n_cu) wpc = 4;
  // THREE waves per SIMD (the per-edge form: 153 VGPRs, 8.3 KB of LDS) only for queues of 128 problems per CU and more:
  // measured round 6 on KUKA, 65536 goals 130.5 k -> 134.2 k solves/s (+2.9 %), but 8192 goals 55.4 k -> 51.5 k (-7 %) --
  // a mid-size batch is its stragglers, and a straggler with two co-resident waves runs slower than with one
  if (!t->is_block && t->K == 3 && wpc > 8 && (long long)B < 128LL * t->n_cu) wpc = 8;
  if (t->wpc_override > 0) wpc = t->wpc_override;
  const int grid = std::min(B, t->n_cu * wpc);
  // Time slicing (workgroup-per-problem kernel): only when there are more problems than resident
  // workgroups (otherwise everything starts at once anyway).  Slice length: the handle's
  // slice_outer_its, 0 disables.  Measured on UR10 + table, 4096 goals: 8.9 -> 7.7 s.
  int slice = t->is_npt ? t->npt_slice_its : t->slice_its;
  const bool cg = t->solver == GIK_SOLVER_CONJUGATE_GRADIENT;
  if (!t->is_block || cg || B <= grid || (a.dbg & 1) || slice <= 0 || t->p.maxiter <= slice) slice = 0;
  a.slice_its = slice;
  a.y_head = a.y_tail = a.y_seq = nullptr;
  a.y_ids = a.y_avail = nullptr;
  a.y_cap = 0;
  a.q_tail = a.q_done = a.q_head = nullptr;
  a.mig_credits = a.mig_simd_run = nullptr;
  a.q_ids = nullptr;
  a.q_seq = nullptr;
  a.q_state = nullptr;
  // Tail spreading (wavefront kernel): only where two waves share a SIMD and the batch outlasts
  // the queue -- more problems than resident waves -- and only on the tuned default variant
  // (trust-region solver, theta = 1, not anchored).  debug_flags 512 turns it off (tests compare).
  // (At one wave per SIMD -- batches up to 6 problems per SIMD -- round-robin slicing LOSES 5-8 %: a
  // straggler that happens to start at t = 0 is better off keeping its slot than sharing it for the
  // first ~20 ms; measured on 4096 LWA4D / KUKA / UR10 goals, four seeds each, tools/attic/dev_rr_midbatch.py.)
  const bool mig = !t->is_block && !cg && !t->anchored && t->variant->solve_mig &&
                   (!t->hess_per_edge || t->variant->solve_strict_mig) && t->p.theta == 1.0 &&
                   wpc > 4 && B > grid && !(a.dbg & (1 | 512));
  gik_template::SliceWs *sw = nullptr;
  // graphs beyond 128 nodes (node-per-lane kernel on four wavefronts): the clique's target triangle of every resident
  // workgroup lives in global memory -- a region of the same pooled workspace
  const size_t ctg_bytes = t->is_npt ? t->npt_variant->ctg(t->nt.n_pairs) * sizeof(double) * (size_t)grid : 0;
  if (slice > 0 || mig || ctg_bytes) {
    const size_t cap = mig ? (size_t)B + (size_t)grid + 64 : (slice > 0 ? (size_t)B * (size_t)(t->p.maxiter / slice + 1) : 0);
    // wavefront kernel: round-robin slicing (slice length: the handle's wave_slice_its)
    // Slice length grows with the queue: 256 iterations up to 8 problems per wave, 4 x that from 32 per wave on.
    // A hand-over moves ~4.5 KB through HBM (point, state, the targets re-read; PMC, round 3: 624 MB per 65536-goal
    // KUKA launch = 6.3 x the algorithmic bytes at 118 k hand-overs); measured round 4 (tools/slice_scan.py), KUKA
    // 65536: slice 256 / 512 / 1024 / 2048 -> 501.5 / 500.9 / 510.6 / 514.0 ms and 118 k / 51 k / 19 k / 7.5 k
    // hand-overs; KUKA 8192: 145.5 / 147.7 / 161.6 ms -- short queues want the short slice.
    int wslice = (mig && !(a.dbg & 1024)) ? t->wave_slice_its : 0;
    // (PMC, round 4, 65536 KUKA goals, tools/attic/c4_slice_traffic.sh: no slicing 202 MB per launch = 2.0 x the
    // algorithmic bytes -- the floor of this kernel's 432 / 600-byte rows -- at 555 ms; slice 1024: 287 MB, 519 ms;
    // 1536: 526 ms; 2048: 239 MB = 2.4 x, 537 ms.  Throughput decides: 1024.)
    if (wslice > 0 && t->wave_slice_auto && (long long)B > 8LL * grid)
      wslice = (int)std::min<long long>(4LL * wslice, (long long)wslice * B / (8LL * grid));
    // yield queue: a problem yields at most maxiter / slice + 1 times; the margin covers the waves that may be
    // between the capacity test and their push (mig_anyone_waiting)
    const size_t ycap = wslice > 0 ? std::min((size_t)B * (size_t)(t->p.maxiter / wslice + 2), (size_t)16 * B + 8192) +
                                         2 * (size_t)grid + 256
                                   : 0;      // (very short slices: the queue fills and the problems stop yielding)
    const size_t off_simd = 32, off_seq = off_simd + (mig ? sizeof(int) * MIG_SIMDS : 0), off_ids = off_seq + cap * 4,
                 off_state = (off_ids + cap * 4 + 15) & ~(size_t)15,
                 off_yseq = off_state + (((size_t)B * sizeof(SliceState) + 15) & ~(size_t)15), off_yids = off_yseq + ycap * 4;
    const size_t off_ctg = (off_yids + ycap * 4 + 63) & ~(size_t)63;
    const size_t bytes = off_ctg + ctg_bytes;
    sw = &mt->slice_ws[take(mt->slice_ws, mt->next_slice, (unsigned)t->slice_pool)];
    release.sw = sw;
    if (!sw->done && hipEventCreateWithFlags(&sw->done, hipEventDisableTiming) != hipSuccess)
      return fail("hipEventCreate failed");
    if (sw->bytes < bytes) {
      if (sw->pending) (void)hipEventSynchronize(sw->done);
      if (sw->base) (void)hipFree(sw->base);
      sw->base = nullptr;
      sw->bytes = 0;
      if (hipMalloc(&sw->base, bytes) != hipSuccess) return fail("cannot allocate the time-slicing workspace");
      sw->bytes = bytes;
      sw->pending = false;
    }
    if (sw->pending) HIP_OK(hipStreamWaitEvent((hipStream_t)stream, sw->done, 0));
    char *base = static_cast<char *>(sw->base);
    a.q_tail = reinterpret_cast<unsigned int *>(base);
    a.q_done = a.q_tail + 1;
    a.q_head = a.q_tail + 2;
    a.mig_credits = reinterpret_cast<int *>(a.q_tail + 3);
    a.mig_simd_run = reinterpret_cast<int *>(base + off_simd);
    a.q_seq = reinterpret_cast<unsigned int *>(base + off_seq);
    a.q_ids = reinterpret_cast<int *>(base + off_ids);
    a.q_state = reinterpret_cast<SliceState *>(base + off_state);
    a.y_head = a.q_tail + 4;
    a.y_tail = a.q_tail + 5;
    a.y_avail = reinterpret_cast<int *>(a.q_tail + 6);
    a.y_seq = reinterpret_cast<unsigned int *>(base + off_yseq);
    a.y_ids = reinterpret_cast<int *>(base + off_yids);
    a.y_cap = (unsigned int)ycap;
    a.npt_ctg_ws = ctg_bytes ? reinterpret_cast<double *>(base + off_ctg) : nullptr;
    if (mig) { a.slice_its = wslice; a.slice_cycles = t->wave_slice_cycles; }
    HIP_OK(hipMemsetAsync(base, 0, off_seq, (hipStream_t)stream));
    if (cap) HIP_OK(hipMemsetAsync(a.q_seq, 0xFF, cap * 4, (hipStream_t)stream));
    if (mig || t->is_npt) HIP_OK(hipMemsetAsync(a.q_state, 0, (size_t)B * sizeof(SliceState), (hipStream_t)stream));
    if (ycap) HIP_OK(hipMemsetAsync(a.y_seq, 0, ycap * 4, (hipStream_t)stream));
  }
  // four planar problems per wavefront: from 12 problems per CU on (measured, planar-10, events around the call:
  // 4..64 problems 158 against 95 us, 1024: 206 / 150, 4096: 259 / 282 -- below that every problem has a wavefront
  // of its own anyway and the lone problem is faster there); debug_flags 16384: at any batch size
  const bool quad = t->quad_solve && !(a.dbg & (1 | 8192)) && ((a.dbg & 16384) || B >= t->quad_min_batch);
  if (quad) {
    // a wavefront holds four problems: a quarter of the waves (at least one slot each), no slicing
    int qw = t->quad_waves_per_cu;
    if (t->wpc_override > 0) qw = t->wpc_override;
    const int qgrid = std::max(1, std::min((B + QUAD_SLOTS - 1) / QUAD_SLOTS, t->n_cu * qw));
    hipLaunchKernelGGL(t->quad_solve, dim3(qgrid), dim3(WAVE), t->quad_smem, (hipStream_t)stream, a);
  } else if (t->is_npt) {
    a.nt = t->nt;
    hipLaunchKernelGGL(t->npt_variant->solve, dim3(grid), dim3(WAVE * t->npt_variant->NW), t->npt_smem, (hipStream_t)stream, a);
  } else if (t->is_block) {
    void (*kern)(SolveArgs, int) =
        cg ? (t->K == 3 ? rcg_block_kernel<3> : rcg_block_kernel<2>)
           : (t->K == 3 ? rtr_block_kernel<3> : rtr_block_kernel<2>);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK_NT), t->smem_bytes, (hipStream_t)stream, a, t->SL);
  } else {
    hipLaunchKernelGGL(t->anchored ? t->variant->solve_anch
                       : cg        ? t->variant->solve_cg
                       : t->hess_per_edge ? (t->p.theta != 1.0 ? t->variant->solve_strict_theta
                                             : (mig ? t->variant->solve_strict_mig : t->variant->solve_strict))
                       : mig       ? t->variant->solve_mig
                                   : (t->p.theta == 1.0 ? t->variant->solve : t->variant->solve_theta),
                       dim3(grid), dim3(WAVE), t->smem_bytes, (hipStream_t)stream, a);
  }
  HIP_OK(hipGetLastError());
  // The slots go back "pending" only behind an event that really covers this launch.  If a record fails,
  // nothing guards the counter / workspace against the next caller: wait for the kernel here and hand the
  // slots back idle instead.
  const hipError_t e_cs = hipEventRecord(cs.done, (hipStream_t)stream);
  const hipError_t e_sw = sw ? hipEventRecord(sw->done, (hipStream_t)stream) : hipSuccess;
  if (e_cs != hipSuccess || e_sw != hipSuccess) {
    (void)hipStreamSynchronize((hipStream_t)stream);
    return fail(std::string("hipEventRecord failed after the launch: ") + hipGetErrorString(e_cs != hipSuccess ? e_cs : e_sw));
  }
  release.launched = true;
  return 0;
}

int gik_template_get_info(const gik_template *t, gik_template_info *info) {
  if (!t || !info) return gik::fail("null argument");
  std::memset(info, 0, sizeof(*info));
  info->is_block = t->is_block ? 1 : 0;
  info->max_terms_per_node = t->is_block ? 0 : t->variant->maxdeg;
  info->n_clique = t->bt.n_clq;
  info->n_slot_terms = t->is_block ? t->bt.Tc : t->T;
  info->slots_per_thread = t->SL;
  info->waves_per_cu = t->waves_per_cu;
  info->n_cu = t->n_cu;
  info->lds_bytes = (int32_t)t->smem_bytes;
  info->clique_closed_form = t->clique_mode;
  info->hessian_form = (t->hess_per_edge || t->is_block) ? GIK_HESS_PER_EDGE : GIK_HESS_COLUMN;
  info->anchored = t->anchored ? 1 : 0;
  info->has_pipeline = t->has_pipe ? 1 : 0;
  info->prepare_is_block = t->prep_block ? 1 : 0;
  info->node_per_lane = t->is_npt ? t->npt_variant->NW : 0;
  info->goals_per_wave = !t->has_pipe || t->prep_block ? 0 : (t->prep_quad ? gik::QUAD_SLOTS : 1);
  info->problems_per_wave = t->is_block ? 0 : ((t->quad_solve && !(t->dbg & (1 | 8192))) ? gik::QUAD_SLOTS : 1);
  if (t->is_npt) {
    info->waves_per_cu = t->npt_waves_per_cu;
    info->lds_bytes = (int32_t)t->npt_smem;
  }
  return 0;
}

#ifdef GIK_DEV
// developer hook (not part of the ABI header): cycles per iteration of one kernel component
double gik_debug_parts(const gik_template *t, int mode, int iters) {
  using namespace gik;
  double *d = nullptr, h[2] = {0, 0};
  if (hipMalloc((void **)&d, 2 * sizeof(double)) != hipSuccess) return -1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  if (t->K == 3 && t->maxdeg == 9)
    hipLaunchKernelGGL((parts_kernel<3, 9>), dim3(1), dim3(WAVE), t->smem_bytes, 0, t->d_slot_meta,
                       t->N, t->T, mode % 100, iters, d);
  else if (t->K == 3 && t->maxdeg == 10)
    hipLaunchKernelGGL((parts_kernel<3, 10>), dim3(1), dim3(WAVE), t->smem_bytes, 0, t->d_slot_meta,
                       t->N, t->T, mode % 100, iters, d);
  else if (t->K == 3)
    return -1;
  else
    hipLaunchKernelGGL((parts_kernel<2, 6>), dim3(1), dim3(WAVE), t->smem_bytes, 0, t->d_slot_meta,
                       t->N, t->T, mode % 100, iters, d);
  (void)hipEventRecord(e1, 0);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(d);
  if (mode >= 100) return ms * 1e6 / iters;  // ns per iteration (wall)
  return h[0];
}

// developer hook (not part of the ABI header): copy the GIK_DBG=4 dump to the host
int gik_debug_fetch(double *host, int n) {
  if (!gik::g_dbg_buf) return -1;
  return hipMemcpy(host, gik::g_dbg_buf, sizeof(double) * n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif  // GIK_DEV

}  // extern "C"
