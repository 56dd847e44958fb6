// graphik_amd/csrc/gik_wave.hip.h -- one-wavefront-per-IK-problem device code (gfx950 / CDNA4)
//
// Layout.  A problem is the N x k point matrix Y (N*k <= 64 unknowns).  Lane l owns the scalar
// unknown (node i = l / k, component c = l % k); every tangent vector of the trust-region
// solver (Y, grad, eta, Heta, r, delta, Hdelta) is ONE fp64 register per lane, so axpys are a
// single v_fma_f64 and Frobenius inner products are one multiply + a DPP wave reduction.
//
// The masked EDM cost couples node i to its graph neighbours.  Each lane walks its node's
// residual terms ("slots", sorted by neighbour index -- the accumulation order of the
// reference loops, costs.py:98-123,175-207).
//   * cost / gradient (once per outer iteration) gather whole neighbour rows from LDS.  To keep
//     that gather free of per-lane component selects the point matrix is published k times,
//     tile c holding every row rotated left by c, so that lane (i,c) always finds "its"
//     component first:  tile_c[j] = (v[j][c], v[j][c+1], v[j][c+2]).  Rows are 48 B (k=3).
//   * the Hessian-vector product (every truncated-CG iteration) is in column form: lane (i,c)
//     fetches only W_j[c] from the natural-order tile 0 and the k lanes of a node exchange their
//     partial sums with whole-wave DPP shifts (see WaveCtx::ehess).
// Per-slot constants that stay fixed during one truncated-CG solve (the row of the 3x3 Hessian
// block 2a y y^T + c I that belongs to the lane) live in registers; per-problem targets, the
// per-(slot, lane) records and the slot metadata live in LDS.  HBM is touched only at problem
// entry/exit.
//
// A single wavefront executes its DS instructions in order, so a ds_write followed by
// ds_reads of other lanes' data needs no barrier.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "graphik_amd.h"

namespace gik {

constexpr int WAVE = 64;

constexpr int TILE_ROWS = 33;  // 32 nodes max + one dump row for idle lanes

// slot metadata word: [7:0] neighbour node j, [23:8] term index, [25:24] kind, [26] owner
__host__ __device__ inline uint32_t meta_pack(int j, int term, int kind, int owner) {
  return (uint32_t)j | ((uint32_t)term << 8) | ((uint32_t)kind << 24) | ((uint32_t)owner << 26);
}
__host__ __device__ inline int meta_j(uint32_t m) { return (int)(m & 0xffu); }
__device__ inline int meta_term(uint32_t m) { return (int)((m >> 8) & 0xffffu); }
__device__ inline int meta_kind(uint32_t m) { return (int)((m >> 24) & 3u); }
__device__ inline int meta_owner(uint32_t m) { return (int)((m >> 26) & 1u); }

// ---- cross-lane primitives ---------------------------------------------------------------
template <int CTRL>
__device__ inline double dpp_f64(double v) {
  // row-local permutations with a valid source in every lane: no `old` operand, so neither a
  // zero-initialising v_mov nor a copy of the source is needed (in-place is safe within a row)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ inline double readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// whole-wavefront lane shifts (GFX9 DPP wave_shr:1 / wave_shl:1): lane l reads lane l-N / l+N
// The destination must not alias the source: the 64 lanes execute 16 per cycle, so an in-place
// cross-row shift would read lanes the previous pass already overwrote.  The empty asm keeps
// sources and results live at the same time, which forces distinct registers at no cost.
template <int CTRL>
__device__ inline double dpp_fresh_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int rlo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  const int rhi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  asm volatile("" ::"v"(lo), "v"(hi), "v"(rlo), "v"(rhi));
  return __hiloint2double(rhi, rlo);
}
// cond ? a : b as pure data flow: (m & a) | (~m & b) per 32-bit half
__device__ inline double bit_select(bool cond, double a, double b) {
  const unsigned m = cond ? 0xffffffffu : 0u;
  const unsigned lo = ((unsigned)__double2loint(a) & m) | ((unsigned)__double2loint(b) & ~m);
  const unsigned hi = ((unsigned)__double2hiint(a) & m) | ((unsigned)__double2hiint(b) & ~m);
  return __hiloint2double((int)hi, (int)lo);
}
template <int N>
__device__ inline double wave_shr(double v) {
#pragma unroll
  for (int i = 0; i < N; ++i) v = dpp_fresh_f64<0x138>(v);
  return v;
}
template <int N>
__device__ inline double wave_shl(double v) {
#pragma unroll
  for (int i = 0; i < N; ++i) v = dpp_fresh_f64<0x130>(v);
  return v;
}

template <int CTRL, int ROW_MASK>
__device__ inline double dpp_rows_f64(double v) {
  // rows not selected by ROW_MASK receive 0.0 (`old`), selected rows the permuted source
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Sum NV independent per-lane values over the wavefront; result is wave-uniform (SGPR-backed).
// Four DPP butterfly stages inside each row of 16 lanes (quad_perm xor1, xor2, half-mirror,
// mirror) leave every lane with its row total; row_bcast15 folds row0 into row1 and row2 into
// row3, row_bcast31 folds (row0+row1) into row3, whose lane 63 is read back with v_readlane.
// Fixed order => bit-reproducible run to run.
template <int NV>
__device__ inline void wave_sum_n(double (&v)[NV]) {
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] += dpp_f64<0xB1>(v[q]);  // quad_perm [1,0,3,2]
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] += dpp_f64<0x4E>(v[q]);  // quad_perm [2,3,0,1]
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] += dpp_f64<0x141>(v[q]); // row_half_mirror
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] += dpp_f64<0x140>(v[q]); // row_mirror
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const double r0 = readlane_f64(v[q], 0), r1 = readlane_f64(v[q], 16);
    const double r2 = readlane_f64(v[q], 32), r3 = readlane_f64(v[q], 48);
    v[q] = (r0 + r1) + (r2 + r3);
  }
}

// ---- wave reductions on the matrix core ----------------------------------------------------
// A lone wavefront issues an fp64 VALU op every 5-8 cycles and a DPP move every 4, so the classic
// 6-stage DPP butterfly (2 v_mov_dpp + 1 v_add_f64 per value and stage) costs ~100 cycles per
// reduced value plus the cross-row read-back.  v_mfma_f64_4x4x4 sums across lanes for
// free: with B = 1 it returns  D[i][j] = sum_k A[i][k],  where on gfx950 (probed, see
// tools/exp/mfma_layout.hip)  A[i][k] = lane (4b + i) + 16k  and  D[i][j] = lane (4b + j) + 16i
// for the four 4-lane blocks b of a 16-lane row.  One instruction therefore adds the four rows
// of a lane column, and feeding the result back in adds the four columns of a block: after two
// instructions every lane holds the total of its block's 16 lanes (4 lanes x 4 rows).
//   - one value : blocksum, then the four block totals are combined with three row_ror DPP moves.
//   - up to four values: two transposing exchange stages first (partner = lane^8, then lane^4):
//     a lane keeps one value of each pair and receives the partner's contribution to it, so
//     afterwards block b of every row carries value perm(b); blocksum finishes all four at once.
// Fixed data flow => bit-reproducible.  ~170 cycles for one value, ~300 for four.
template <int CTRL, int BANK>
__device__ inline double dpp_bank_f64(double old, double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(old), lo, CTRL, 0xf, BANK, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(old), hi, CTRL, 0xf, BANK, false);
  return __hiloint2double(hi, lo);
}
__device__ inline double mfma_blocksum(double v) {
  const double p = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
  return __builtin_amdgcn_mfma_f64_4x4x4f64(p, 1.0, 0.0, 0, 0, 0);
}
__device__ inline double lane_xor4(double x) {
  const double t = dpp_bank_f64<0x104, 0x5>(x, x);  // row_shl:4 -> banks 0,2 read lane+4
  return dpp_bank_f64<0x114, 0xA>(t, x);            // row_shr:4 -> banks 1,3 read lane-4
}
__device__ inline void wave_sum4(double &v0, double &v1, double &v2, double &v3) {
  const int lane = threadIdx.x;
  const bool h8 = lane & 8, h4 = lane & 4;
  const double u0 = (h8 ? v1 : v0) + dpp_f64<0x128>(h8 ? v0 : v1);  // row_ror:8 == lane^8
  const double u1 = (h8 ? v3 : v2) + dpp_f64<0x128>(h8 ? v2 : v3);
  const double w = (h4 ? u1 : u0) + lane_xor4(h4 ? u0 : u1);
  const double q = mfma_blocksum(w);
  v0 = readlane_f64(q, 0);
  v2 = readlane_f64(q, 4);
  v1 = readlane_f64(q, 8);
  v3 = readlane_f64(q, 12);
}
template <>
__device__ inline void wave_sum_n<1>(double (&v)[1]) {
  const double q = mfma_blocksum(v[0]);
  const double t = (q + dpp_f64<0x124>(q)) + (dpp_f64<0x128>(q) + dpp_f64<0x12C>(q));
  v[0] = readlane_f64(t, 0);
}
template <>
__device__ inline void wave_sum_n<2>(double (&v)[2]) {
  double z0 = 0.0, z1 = 0.0;
  wave_sum4(v[0], v[1], z0, z1);
}
template <>
__device__ inline void wave_sum_n<3>(double (&v)[3]) {
  double z = 0.0;
  wave_sum4(v[0], v[1], v[2], z);
}
template <>
__device__ inline void wave_sum_n<4>(double (&v)[4]) {
  wave_sum4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ inline void wave_sum_n<6>(double (&v)[6]) {
  double z0 = 0.0, z1 = 0.0;
  wave_sum4(v[0], v[1], v[2], v[3]);
  wave_sum4(v[4], v[5], z0, z1);
}
// Eight values: transposing exchanges on lane bits 5 and 4 with the gfx950 register-pair swaps
// (v_permlane32_swap / v_permlane16_swap exchange the upper half of one register with the lower
// half of another, so "keep one value of the pair, receive the partner's share of it" needs no
// selects), one select exchange on bit 3, then a plain butterfly over bits 0-2 of the single
// remaining register.  Value q ends up in the lanes with (bit5, bit4, bit3) = (q&1, q&2, q&4).
// Measured 310 cycles against 390 for two wave_sum4 (tools/exp/reduce8.hip).
__device__ inline void lane_swap32(double &a, double &b) {
  unsigned al = __double2loint(a), ah = __double2hiint(a), bl = __double2loint(b), bh = __double2hiint(b);
  const auto r = __builtin_amdgcn_permlane32_swap(al, bl, false, false);
  const auto s = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
  a = __hiloint2double((int)s[0], (int)r[0]);
  b = __hiloint2double((int)s[1], (int)r[1]);
}
__device__ inline void lane_swap16(double &a, double &b) {
  unsigned al = __double2loint(a), ah = __double2hiint(a), bl = __double2loint(b), bh = __double2hiint(b);
  const auto r = __builtin_amdgcn_permlane16_swap(al, bl, false, false);
  const auto s = __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
  a = __hiloint2double((int)s[0], (int)r[0]);
  b = __hiloint2double((int)s[1], (int)r[1]);
}
// the eight wave totals, still distributed: value q in the lanes with (bit5, bit4, bit3) = (q&1, q&2, q&4)
__device__ inline double wave_sum8_distributed(double (&v)[8]) {
#pragma unroll
  for (int q = 0; q < 8; q += 2) {
    lane_swap32(v[q], v[q + 1]);
    v[q] += v[q + 1];
  }
  lane_swap16(v[0], v[2]);
  v[0] += v[2];
  lane_swap16(v[4], v[6]);
  v[4] += v[6];
  const bool h8 = threadIdx.x & 8;
  double w = (h8 ? v[4] : v[0]) + dpp_f64<0x128>(h8 ? v[0] : v[4]);  // row_ror:8 == lane^8
  w += dpp_f64<0xB1>(w);   // quad_perm [1,0,3,2]
  w += dpp_f64<0x4E>(w);   // quad_perm [2,3,0,1]
  w += dpp_f64<0x141>(w);  // row_half_mirror
  return w;
}
template <>
__device__ inline void wave_sum_n<8>(double (&v)[8]) {
  const double w = wave_sum8_distributed(v);
#pragma unroll
  for (int q = 0; q < 8; ++q)
    v[q] = readlane_f64(w, ((q & 1) ? 32 : 0) + ((q & 2) ? 16 : 0) + ((q & 4) ? 8 : 0));
}

__device__ inline double wave_sum(double x) {
  double v[1] = {x};
  wave_sum_n<1>(v);
  return v[0];
}

// 1 / b to about one ulp with a short dependency chain: v_rcp_f64 seed (error e ~ 4.6e-8),
// r1 = r0 (1 + e) and e^2 evaluated side by side, r2 = r1 (1 + e^2).  IEEE division is a chain
// of ten dependent fp64 instructions (~108 cycles for a lone wavefront); this is five.
__device__ inline double frcp(double b) {
  const double r0 = __builtin_amdgcn_rcp(b);
  const double e = fma(-b, r0, 1.0);
  const double r1 = fma(e, r0, r0);
  return fma(r1, e * e, r1);
}

// one Newton step on the v_rcp_f64 seed, chain of 3.  Measured on gfx950 over 2^20 samples
// (tools/exp/rcp_accuracy.hip): seed 4.6e-8, frcp1 2.1e-15, frcp 2.2e-16, frsqrt 1.4e-16 max
// relative error -- frcp1 feeds alpha and beta of the tCG step, whose inner products of 54 terms
// carry errors of the same size
__device__ inline double frcp1(double b) {
  const double r0 = __builtin_amdgcn_rcp(b);
  return fma(fma(-b, r0, 1.0), r0, r0);
}

// 1 / sqrt(x): v_rsq_f64 seed + two Newton steps (y <- y + y (1 - x y^2) / 2), ~1 ulp; the
// sqrt-then-divide pair it replaces is ~260 cycles for one wavefront.
__device__ inline double frsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double e = fma(-x * y, y, 1.0);
    y = fma(0.5 * y, e, y);
  }
  return y;
}

// Orthonormal basis of the vertical space at Y (k = 3).  The vertical space is spanned by
// pk_m = Y E_m (E_m the three skew generators); their Gram matrix is
//   M = [[a, b, c], [b, d, e], [c, e, f]]  (see proj_setup).
// With M = L L^T,  Q = pk L^-T  has orthonormal columns, so the horizontal projector of
// fixed_rank_psd_sym.py:91-113 becomes  Z - sum_m Q_m <Q_m, Z>  and
// |proj Z|^2 = |Z|^2 - sum_m <Q_m, Z>^2 -- which lets one reduction per tCG iteration carry
// everything the iteration needs (see rtr_solve_one).
__device__ inline void vertical_basis(double a, double b, double c, double d, double e, double f,
                                      const double (&pk)[3], double (&Q)[3]) {
  const double i00 = frsqrt(a);
  const double l10 = b * i00, l20 = c * i00;
  const double i11 = frsqrt(fma(-l10, l10, d));
  const double l21 = fma(-l20, l10, e) * i11;
  const double i22 = frsqrt(fma(-l21, l21, fma(-l20, l20, f)));
  Q[0] = pk[0] * i00;
  Q[1] = fma(-l10, Q[0], pk[1]) * i11;
  Q[2] = fma(-l21, Q[1], fma(-l20, Q[0], pk[2])) * i22;
}

// ---- solver parameters handed to the kernels ----------------------------------------------
struct Params {
  double mingradnorm, theta, kappa, rho_prime, rho_regularization;
  int maxiter, maxinner, mininner, planar_proj_exact;
};

enum {
  TCG_NEGATIVE_CURVATURE = 0,
  TCG_EXCEEDED_TR = 1,
  TCG_REACHED_TARGET_LINEAR = 2,
  TCG_REACHED_TARGET_SUPERLINEAR = 3,
  TCG_MAX_INNER_ITER = 4,
  TCG_MODEL_INCREASED = 5
};  // trust_region.py:68-75

template <int K>
struct Row {
  double v[K];
};

// ---- per-wave problem context -------------------------------------------------------------
// ANCH: fixed-anchor formulation (opt-in): nodes with a known position (base frame,
// goal nodes, obstacles) are constants instead of rows of Y, and every free node may carry
//   * up to ANCH_PMAX "pinned" point-to-anchor terms (equality / hinge terms to base and goal
//     anchors, own target each; anchor positions in a small LDS table), and
//   * lower hinges against ALL n_obs spherical obstacles (graph_base.py:205-211 as intended:
//     |p_i - o_k| >= r_k), walked by every lane at once so that the obstacle data is wave-uniform
//     (scalar loads from global memory, no LDS).
// An anchor does not move, so these terms touch cost(), commit() and the diagonal blocks `bsum` of
// the Hessian only -- the Hessian-vector product costs what it costs without obstacles.  Anchors
// fix the gauge: the search space is Euclidean (no horizontal projection, Q = 0).
constexpr int ANCH_PMAX = 8;
constexpr int ANCH_MAXA = 16;   // rows of the pinned-anchor table
constexpr int ANCH_MAXOBS = 128; // obstacles (staged in LDS: 32 B each)

// SLIM: the layout of the per-edge product form (WaveCtxStrict, gik_wave_strict.hip.h): only the natural-order tile,
// and slot metadata / slot records of the slots a lane OWNS (node slots comp, comp + 3, ...) -- 8.3 instead of 18.9 KB
// of LDS per wavefront on a 7-DOF arm, which is what lets three wavefronts share a SIMD (153 VGPRs).
template <int K, int MAXDEG, bool ANCH = false, bool SLIM = false>
struct WaveCtx {
  static_assert(!SLIM || (K == 3 && !ANCH), "the slim layout belongs to the 3-D per-edge context");
  static constexpr int RS = (K == 3) ? 6 : 2;  // LDS row stride in doubles (48 B / 16 B)
  static constexpr int NC = (K == 3) ? 3 : 1;  // independent entries of the skew matrix
  static constexpr int TILE = TILE_ROWS * RS;  // doubles per rotated tile
  static constexpr bool SLIM_LAYOUT = SLIM;
  static constexpr int NTILE = SLIM ? 1 : K;   // tiles kept in LDS
  static constexpr int NSL = SLIM ? (MAXDEG + 2) / 3 : MAXDEG;   // slots per lane in the LDS tables
  // Per (slot, lane) record for cost()/commit(): the residual target of the slot's term and the
  // clamp bounds that encode its kind -- EQ (-inf, +inf), LOWER (0, +inf), UPPER (-inf, 0),
  // padding (0, 0) -- so that with u = target - d the residual is clamp(u, lo, hi) for every kind
  // and no metadata has to be decoded per evaluation (one ds_read_b128 per slot).
  struct SlotRec {
    double tg;
    float lo, hi;
  };
  // LDS carve: K tiles | targets[T] | slot meta (MAXDEG*64 u32) | slot records (MAXDEG*64 x 16 B)
  //            | tCG checkpoint (4 x 64 doubles, k = 3: see "Retrace" in rtr_solve_one)
  static constexpr bool HAS_CK = (K == 3);
  static constexpr bool AGE_PRIORITY = true;   // waves of different problems share a SIMD (rtr_solve_one)
  __host__ __device__ static constexpr size_t lds_bytes(int T) {
    return sizeof(double) * ((size_t)NTILE * TILE + (size_t)((T + 1) & ~1)) +
           sizeof(uint32_t) * (size_t)NSL * WAVE + sizeof(SlotRec) * (size_t)NSL * WAVE +
           (HAS_CK ? sizeof(double) * 4 * WAVE : 0) +
           (ANCH ? sizeof(double) * 4 * (ANCH_MAXA + ANCH_MAXOBS) + 16 * (size_t)ANCH_PMAX * WAVE +
                       48 * (size_t)WAVE : 0);
  }
  // ---- fixed-anchor data (ANCH) ----
  // Everything per lane lives in LDS records (the 9-slot kernel has no VGPR to spare: per-lane
  // offset tables in registers pushed it into scratch, measured 9 GB of spill traffic per launch).
  struct PinRec {
    double tg;        // squared target
    uint32_t meta;    // [7:0] anchor row, [9:8] kind (0 = padding, inert)
    uint32_t pad;
  };
  double *sh_anch;         // [ANCH_MAXA][4] pinned anchor positions
  PinRec *sh_prec;         // [ANCH_PMAX][64] pinned slot records
  // Obstacle centres + squared radii, staged in LDS once per wave and read at a wave-uniform
  // address (broadcast).  Read straight from global memory the two obstacle loops cost a
  // dependent ~400-cycle round trip per obstacle (the compiler emits vector loads for the uniform
  // address): 80 k cycles per outer iteration, more than its truncated-CG solve.
  double *sh_obs;          // [ANCH_MAXOBS][4]
  int n_obs;
  bool obs_lane;           // this lane's node carries the obstacle hinges
  // Near lists.  Walking all obstacles in cost() and commit() costs 29 k cycles per outer iteration
  // (100 spheres: 28 % of the solve), although a hinge can only be active next to the node.  Every
  // node therefore remembers, from its last full walk at position `ref`, the (at most 8) obstacles
  // within OBS_TAU of their surface and the smallest clearance `slack` among all OTHERS.  While
  // the node stays within `slack` of `ref` every other obstacle is provably inactive -- a sphere
  // the node was `c` away from cannot be reached by moving less than `c` -- and contributes exactly
  // zero, so only the near list is walked: bit-identical results (tests), late in a solve (steps of
  // 1e-3 and less) at almost no cost.
  struct ObsState {
    double ref[3];
    double slack;          // <= 0: walk everything (fresh problem, or more than 8 obstacles near)
    uint32_t idx[2];       // 8 obstacle indices, one byte each
    uint32_t cnt, pad;
  };
  static constexpr float OBS_TAU = 0.05f;
  ObsState *sh_ost;        // [64]
  bool obs_cull;
  __device__ inline void init_anchored(const uint64_t obs_mask, const double *obs, int n_obs_, bool cull) {
    sh_anch = sh_ck + 4 * WAVE;
    sh_obs = sh_anch + 4 * ANCH_MAXA;
    sh_prec = reinterpret_cast<PinRec *>(sh_obs + 4 * ANCH_MAXOBS);
    sh_ost = reinterpret_cast<ObsState *>(sh_prec + ANCH_PMAX * WAVE);
    n_obs = n_obs_;
    obs_cull = cull;
    obs_lane = active && ((obs_mask >> node) & 1ull);
    for (int t = lane; t < 4 * n_obs_; t += WAVE) sh_obs[t] = obs[t];
    obs_reset();
  }
  // per problem: forget the near lists
  __device__ inline void obs_reset() {
    ObsState st = {{0.0, 0.0, 0.0}, -1.0, {0u, 0u}, 0u, 0u};
    sh_ost[lane] = st;
    __builtin_amdgcn_wave_barrier();
  }
  // true (wave-uniform) when every node is within the slack of its last full walk
  __device__ inline bool obs_near_only(const Row<K> &nat) const {
    const ObsState &st = sh_ost[lane];
    const double m0 = nat.v[0] - st.ref[0], m1 = nat.v[1] - st.ref[1], m2 = nat.v[K - 1] - st.ref[2];
    const double moved2 = fma(m2, m2, fma(m1, m1, m0 * m0));
    const bool ok = !obs_lane || (st.slack > 0.0 && moved2 * (1.0 + 1e-9) < st.slack * st.slack);
    return obs_cull && __builtin_amdgcn_ballot_w64(!ok) == 0ull;
  }
  // one obstacle during a full walk: classify it for the near list (conservative float bound of
  // its clearance), returns the clamped residual max(r^2 - d, 0)
  __device__ inline double obs_visit(const double4 &o, double d, int k, double &slack, uint32_t (&idx)[2],
                                     uint32_t &cnt) const {
    const float clr = __builtin_sqrtf((float)d) * (1.0f - 1e-6f) - __builtin_sqrtf((float)o.w) * (1.0f + 1e-6f) - 1e-6f;
    if (clr < OBS_TAU) {
      if (cnt < 8u) idx[cnt >> 2] |= (uint32_t)k << (8u * (cnt & 3u));
      ++cnt;
    } else {
      slack = fmin(slack, (double)clr);
    }
    return obs_lane ? fmax(o.w - d, 0.0) : 0.0;
  }
  __device__ inline void obs_store(const Row<K> &nat, double slack, const uint32_t (&idx)[2], uint32_t cnt) {
    ObsState st;
    st.ref[0] = nat.v[0];
    st.ref[1] = nat.v[1];
    st.ref[2] = nat.v[K - 1];
    st.slack = cnt > 8u ? -1.0 : slack;
    st.idx[0] = idx[0];
    st.idx[1] = idx[1];
    st.cnt = cnt > 8u ? 0u : cnt;
    st.pad = 0u;
    sh_ost[lane] = st;
  }
  // pinned slot records: anchor terms have template-constant targets
  __device__ inline void load_pinned_records(const uint32_t *pin_meta, const double *pin_tgt) {
#pragma unroll 1
    for (int s = 0; s < ANCH_PMAX; ++s) {
      PinRec r;
      r.tg = pin_tgt[s * WAVE + lane];
      r.meta = pin_meta[s * WAVE + lane];
      r.pad = 0;
      sh_prec[s * WAVE + lane] = r;
    }
    __builtin_amdgcn_wave_barrier();
  }
  // clamped residual of a point-to-anchor term from its squared distance: clamp(tg - d, lo, hi) with
  // the kind's bounds (see SlotRec); `eq` returns whether the term is an equality
  __device__ static inline double pin_residual(double tg, int kind, double d, bool &eq) {
    const double u = tg - d;
    eq = kind == GIK_TERM_EQ;
    const double lo = (kind == GIK_TERM_EQ || kind == GIK_TERM_UPPER) ? -__builtin_inf() : 0.0;
    const double hi = (kind == GIK_TERM_EQ || kind == GIK_TERM_LOWER) ? __builtin_inf() : 0.0;
    return fmin(fmax(u, lo), hi);
  }
  // this lane's rotated view (y_c, y_c+1, y_c+2) of a natural-order 3-vector
  __device__ inline void rotate_nat(const double (&yn)[3], double (&y)[K]) const {
    static_assert(!ANCH || K == 3, "");
    y[0] = comp == 0 ? yn[0] : (comp == 1 ? yn[1] : yn[2]);
    y[1] = comp == 0 ? yn[1] : (comp == 1 ? yn[2] : yn[0]);
    y[K - 1] = comp == 0 ? yn[2] : (comp == 1 ? yn[0] : yn[1]);
  }
  __device__ static inline SlotRec *rec_base(uint32_t *meta) {
    return reinterpret_cast<SlotRec *>(meta + NSL * WAVE);
  }
  __device__ inline void ck_put(int i, double v) { sh_ck[i * WAVE + lane] = v; }
  __device__ inline double ck_get(int i) const { return sh_ck[i * WAVE + lane]; }

  int lane, node, comp;
  bool active;

  // reductions over the problem's unknowns (interface shared with the block context)
  template <int NV>
  __device__ inline void sum_n(double (&v)[NV]) {
    wave_sum_n<NV>(v);
  }
  __device__ inline double sum1(double x) { return wave_sum(x); }
  __device__ inline bool lead() const { return lane == 0; }

  double *sh_tile;         // K rotated tiles
  const double *sh_tgt;    // [T] per-problem residual targets
  const uint32_t *sh_meta; // [MAXDEG][64]
  SlotRec *sh_rec;         // [MAXDEG][64]
  double *sh_ck;           // [4][64] tCG checkpoint (own lane only: no barrier needed)
  int waddr[K];            // where this lane's value goes in tile 0..K-1 (double index)
  int own_off;             // this lane's node row in its own tile (double index)
  int nat_off;             // this lane's node row in tile 0 (natural component order)
  int coloff[MAXDEG];      // neighbour entry (j, comp) in tile 0 (double index)
  int tile_delta;          // row of node j in this lane's rotated tile = coloff + tile_delta
  __device__ inline int rowoff(int s) const { return coloff[s] + tile_delta; }
  // Row of the 3x3 (2x2) Hessian block of slot s that belongs to this lane's component, rotated
  // like the tiles:  bq[s][q] = 2 a y_c y_(c+q) + c_ij [q == 0].   bsum = sum_s bq[s].
  double bq[MAXDEG][K];
  double bsum[K];
  double pk[NC], pk2[NC], Pm[NC * NC];
  double G2[NC * NC];      // <pk2_q, pk2_m>, constant during one tCG solve
  double Q[NC];            // k = 3: this lane's entries of the orthonormal vertical basis

  __device__ inline Row<K> read_row(int off) const {
    Row<K> r;
    const double2 a = *reinterpret_cast<const double2 *>(sh_tile + off);
    r.v[0] = a.x;
    r.v[1] = a.y;
    if constexpr (K == 3) r.v[2] = sh_tile[off + 2];
    return r;
  }

  // publish a lane-distributed vector as rows of the K rotated LDS tiles
  __device__ inline void put(double v) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < NTILE; ++t) sh_tile[waddr[t]] = v;
    __builtin_amdgcn_wave_barrier();
  }

  // publish only the natural-order tile (all the column-form Hessian product reads)
  __device__ inline void put1(double v) {
    __builtin_amdgcn_wave_barrier();
    sh_tile[waddr[0]] = v;
    __builtin_amdgcn_wave_barrier();
  }

  // per problem, after the targets were staged: fill this lane's slot records
  __device__ inline void load_slot_records() {
    const float inf = __builtin_inff();
#pragma unroll 1
    for (int s = 0; s < NSL; ++s) {   // once per problem: rolled, to keep register pressure down
      const uint32_t m = sh_meta[s * WAVE + lane];
      const int kind = meta_kind(m);
      SlotRec r;
      r.tg = sh_tgt[meta_term(m)];
      r.lo = (kind == GIK_TERM_EQ || kind == GIK_TERM_UPPER) ? -inf : 0.0f;
      r.hi = (kind == GIK_TERM_EQ || kind == GIK_TERM_LOWER) ? inf : 0.0f;
      sh_rec[s * WAVE + lane] = r;
    }
    __builtin_amdgcn_wave_barrier();
  }

  __device__ inline void init(int lane_, int N, double *tiles, const double *tgt,
                              uint32_t *meta) {
    lane = lane_;
    sh_rec = rec_base(meta);
    sh_ck = reinterpret_cast<double *>(sh_rec + NSL * WAVE);
    active = lane < N * K;
    node = active ? lane / K : (TILE_ROWS - 1);
    comp = active ? lane - node * K : 0;
    sh_tile = tiles;
    sh_tgt = tgt;
    sh_meta = meta;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const int pos = (comp - t + K) % K;  // tile t stores component (t + pos) % K at pos
      waddr[t] = t * TILE + node * RS + pos;
    }
    own_off = comp * TILE + node * RS;
    tile_delta = comp * TILE - comp;
    nat_off = node * RS;
    if constexpr (!SLIM) {
#pragma unroll
      for (int s = 0; s < MAXDEG; ++s) {
        coloff[s] = meta_j(sh_meta[s * WAVE + lane]) * RS + comp;
#pragma unroll
        for (int q = 0; q < K; ++q) bq[s][q] = 0.0;
      }
    }
  }

  // f(Yv): lcost / jcost (costs.py:80-93, 8-16).  Leaves the rows of Yv in the LDS tiles.
  // With u = target - d every residual is clamp(u, lo, hi) (see SlotRec).  Every term sits in the
  // slot lists of both of its nodes; the component-0 lane of each node accumulates, so each term
  // is counted exactly twice and the total is halved (exact).
  __device__ inline double cost(double Yv) {
    put(Yv);
    const Row<K> own = read_row(own_off);
    double f = 0.0;
#pragma unroll
    for (int s = 0; s < MAXDEG; ++s) {
      const Row<K> r = read_row(rowoff(s));
      const SlotRec rc = sh_rec[s * WAVE + lane];
      double y = own.v[0] - r.v[0];
      double d = y * y;
#pragma unroll
      for (int q = 1; q < K; ++q) {
        y = own.v[q] - r.v[q];
        d = fma(y, y, d);
      }
      const double cl = fmin(fmax(rc.tg - d, (double)rc.lo), (double)rc.hi);
      f = fma(cl, cl, f);
      if (s % 3 == 2) __builtin_amdgcn_sched_barrier(0);  // bound the number of rows in flight
    }
    if constexpr (ANCH) {
      // point-to-anchor terms occur once (not from both ends): count them twice before the halving
      double fa = 0.0;
      const Row<K> nat = read_row(nat_off);
#pragma unroll 1
      for (int s = 0; s < ANCH_PMAX; ++s) {
        const PinRec rc = sh_prec[s * WAVE + lane];
        const double *a = sh_anch + (rc.meta & 0xffu) * 4;
        const double y0 = nat.v[0] - a[0], y1 = nat.v[1] - a[1], y2 = nat.v[K - 1] - a[2];
        const double d = fma(y2, y2, fma(y1, y1, y0 * y0));
        bool eq;
        const double cl = pin_residual(rc.tg, (int)((rc.meta >> 8) & 3u), d, eq);
        fa = fma(cl, cl, fa);
      }
      if (obs_near_only(nat)) {
        const ObsState &st = sh_ost[lane];
        const uint32_t cnt = obs_lane ? st.cnt : 0u;
        for (uint32_t q = 0; __builtin_amdgcn_ballot_w64(q < cnt) != 0ull; ++q) {
          const int k = q < cnt ? (int)((st.idx[q >> 2] >> (8u * (q & 3u))) & 0xffu) : 0;
          const double4 o = *reinterpret_cast<const double4 *>(sh_obs + 4 * k);
          const double y0 = nat.v[0] - o.x, y1 = nat.v[1] - o.y, y2 = nat.v[K - 1] - o.z;
          const double d = fma(y2, y2, fma(y1, y1, y0 * y0));
          const double cl = q < cnt ? fmax(o.w - d, 0.0) : 0.0;
          fa = fma(cl, cl, fa);
        }
      } else {
        double slack = 1e30;
        uint32_t idx[2] = {0u, 0u}, cnt = 0u;
#pragma unroll 2
        for (int k = 0; k < n_obs; ++k) {
          const double4 o = *reinterpret_cast<const double4 *>(sh_obs + 4 * k);   // wave-uniform
          const double y0 = nat.v[0] - o.x, y1 = nat.v[1] - o.y, y2 = nat.v[K - 1] - o.z;
          const double d = fma(y2, y2, fma(y1, y1, y0 * y0));
          const double cl = obs_visit(o, d, k, slack, idx, cnt);
          fa = fma(cl, cl, fa);
        }
        obs_store(nat, slack, idx, cnt);
      }
      f = fma(2.0, fa, f);
    }
    return 0.5 * wave_sum((active && comp == 0) ? f : 0.0);
  }

  // Refresh the per-slot constants at the point whose rows are in the LDS tiles and return this
  // lane's entry of egrad (lgrad / jgrad, costs.py:98-123, 20-35): G_i = 2 sum_j c_ij (Y_i-Y_j).
  __device__ inline double commit() {
    const Row<K> own = read_row(own_off);
    double G = 0.0;
#pragma unroll
    for (int s = 0; s < MAXDEG; ++s) {
      const Row<K> r = read_row(rowoff(s));
      double y[K];
      double d = 0.0;
#pragma unroll
      for (int q = 0; q < K; ++q) {
        y[q] = own.v[q] - r.v[q];
        d = fma(y[q], y[q], d);
      }
      const SlotRec rc = sh_rec[s * WAVE + lane];
      // c = d - target where the term is active, else 0: -clamp(target - d, lo, hi).  A hinge is
      // active iff its clamped residual is non-zero (psi_L - d > 0 / d - psi_U > 0), an equality
      // always: lo * hi = -inf only for EQ (0 * inf = NaN and 0 * 0 = 0 compare false).
      const double cl_own = fmin(fmax(rc.tg - d, (double)rc.lo), (double)rc.hi);
      // One residual per TERM: d above is summed in this lane's rotated component order, so the K
      // lanes of a node (and the two ends of an edge) would carry residuals that differ in the
      // last bit.  That makes the round-off of the gradient a vector that is no longer of the form
      // sum_e s_e (e_i - e_j) y_e^T -- the form is horizontal and translation-free exactly, and
      // the reference's G[i] += t, G[j] -= t has it -- and a vertical component of 1e-16 in the
      // start residual of tCG is enough to delay it (measured on UR10: +12 % Hessian products).
      // Every lane therefore uses the value of its node's component-0 lane, which sums in natural
      // order at both ends of the edge.
      double cl;
      if constexpr (K == 3)
        cl = bit_select(comp >= 1, bit_select(comp >= 2, wave_shr<2>(cl_own), wave_shr<1>(cl_own)),
                        cl_own);
      else
        cl = bit_select(comp >= 1, wave_shr<1>(cl_own), cl_own);
      const bool act = (rc.lo * rc.hi < 0.0f) || (cl != 0.0);
      const double c = -cl;
      const double a2 = act ? 2.0 * y[0] : 0.0;          // 2 a y_c, a in {0,1}
#pragma unroll
      for (int q = 0; q < K; ++q) bq[s][q] = a2 * y[q];
      bq[s][0] += c;
#pragma unroll
      for (int q = 0; q < K; ++q) bq[s][q] *= 2.0;      // stored as +2 B_ij ...
      G = fma(c, y[0], G);
      if (s % 3 == 2) __builtin_amdgcn_sched_barrier(0);  // bound the number of rows in flight
    }
    double ba[K];
#pragma unroll
    for (int q = 0; q < K; ++q) ba[q] = 0.0;
    if constexpr (ANCH) {
      // point-to-anchor terms: the anchor does not move, so the term's Hessian block 2 a y y^T + c I
      // goes to the node's diagonal block only (row c of it, rotated like bq)
      const Row<K> nat = read_row(nat_off);
#pragma unroll 1
      for (int s = 0; s < ANCH_PMAX; ++s) {
        const PinRec rc = sh_prec[s * WAVE + lane];
        const double *a = sh_anch + (rc.meta & 0xffu) * 4;
        const double yn[3] = {nat.v[0] - a[0], nat.v[1] - a[1], nat.v[K - 1] - a[2]};
        const double d = fma(yn[2], yn[2], fma(yn[1], yn[1], yn[0] * yn[0]));
        bool eq;
        const double cl = pin_residual(rc.tg, (int)((rc.meta >> 8) & 3u), d, eq);
        double y[K];
        rotate_nat(yn, y);
        const bool act = eq || (cl != 0.0);
        const double c = -cl;
        const double a2 = act ? 2.0 * y[0] : 0.0;
#pragma unroll
        for (int q = 0; q < K; ++q) ba[q] = fma(a2, y[q], ba[q]);
        ba[0] += c;
        G = fma(c, y[0], G);
      }
      // (commit() follows cost() at the same point, so the near lists are fresh; it never refreshes)
      const bool near_only = obs_near_only(nat);
      const ObsState &st = sh_ost[lane];
      const uint32_t ncnt = (near_only && obs_lane) ? st.cnt : 0u;
      const int walk = near_only ? 8 : n_obs;
      for (int k0 = 0; k0 < walk; ++k0) {
        if (near_only && __builtin_amdgcn_ballot_w64((uint32_t)k0 < ncnt) == 0ull) break;
        const bool mine = near_only ? (uint32_t)k0 < ncnt : obs_lane;
        const int k = near_only ? (mine ? (int)((st.idx[k0 >> 2] >> (8u * (k0 & 3u))) & 0xffu) : 0) : k0;
        const double4 o = *reinterpret_cast<const double4 *>(sh_obs + 4 * k);
        const double yn[3] = {nat.v[0] - o.x, nat.v[1] - o.y, nat.v[K - 1] - o.z};
        const double d = fma(yn[2], yn[2], fma(yn[1], yn[1], yn[0] * yn[0]));
        const double cl = mine ? fmax(o.w - d, 0.0) : 0.0;
        if (__builtin_amdgcn_ballot_w64(cl != 0.0) == 0ull) continue;   // nobody touches this obstacle
        double y[K];
        rotate_nat(yn, y);
        const double c = -cl;
        const double a2 = (cl != 0.0) ? 2.0 * y[0] : 0.0;
#pragma unroll
        for (int q = 0; q < K; ++q) ba[q] = fma(a2, y[q], ba[q]);
        ba[0] += c;
        G = fma(c, y[0], G);
      }
    }
#pragma unroll
    for (int q = 0; q < K; ++q) {
      double t = 0.0;
#pragma unroll
      for (int s = 0; s < MAXDEG; ++s) t += bq[s][q];
      bsum[q] = fma(2.0, ba[q], t);                    // ... and bsum = +2 sum_j B_ij (+ anchors)
#pragma unroll
      for (int s = 0; s < MAXDEG; ++s) bq[s][q] = -bq[s][q];
    }
    return 2.0 * G;
  }

  // slot s of the column-form product, with an explicit staged wait: gather s has landed once at
  // most MAXDEG-1-s DS operations are outstanding (DS returns in order)
  template <int S>
  __device__ inline void hv_slots(double (&p)[K], const double (&ww)[MAXDEG]) {
    if constexpr (S < MAXDEG) {
      __builtin_amdgcn_s_waitcnt(0xC07F | ((MAXDEG - 1 - S) << 8));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < K; ++t) p[t] = fma(bq[S][t], ww[S], p[t]);
      __builtin_amdgcn_sched_barrier(0);
      hv_slots<S + 1>(p, ww);
    }
  }

  // ehess(Y, W) (lhess / jhess, costs.py:175-207, 39-58) with Y = last commit():
  //   H_i = 2 sum_j [ 2 a (y.w) y + c w ],  y = Y_i - Y_j,  w = W_i - W_j
  __device__ inline double ehess(double W) {
    // Column form.  A lone wavefront issues a DS instruction only every ~10 (b64) / ~16 (b128)
    // cycles, so the row gathers (ds_read_b128 + ds_read_b64 per neighbour, three ds_write per
    // vector) bound the row-form product.  Here lane (i, c) fetches only W_j[c] -- one 8-byte read
    // per neighbour from the natural-order tile, one write -- and accumulates what its component
    // contributes to all K outputs of node i:
    //     p_t = sum_j B_ij[(c+t)%K][c] (W_i[c] - W_j[c]),   B symmetric => the same bq registers.
    // The K lanes of a node then exchange the K-1 foreign partial sums with whole-wave DPP shifts:
    //     H_(i,c) = p_0 + sum_t ( c >= t ? p_t[lane - t] : p_t[lane + K - t] ).
    put1(W);
    double ww[MAXDEG];
#pragma unroll
    for (int s = 0; s < MAXDEG; ++s) ww[s] = sh_tile[coloff[s]];
    __builtin_amdgcn_sched_barrier(0);
    double p[K];
#pragma unroll
    for (int t = 0; t < K; ++t) p[t] = bsum[t] * W;
    hv_slots<0>(p, ww);
    // bit-select (v_bfi_b32) instead of ?: -- the compiler turns a lane-dependent ?: over the two
    // shifts into divergent branches, and a DPP move under a partial EXEC reads disabled lanes
    double H = p[0];
    if constexpr (K == 3) {
      H += bit_select(comp >= 1, wave_shr<1>(p[1]), wave_shl<2>(p[1]));
      H += bit_select(comp >= 2, wave_shr<2>(p[2]), wave_shl<1>(p[2]));
    } else {
      H += bit_select(comp >= 1, wave_shr<1>(p[1]), wave_shl<1>(p[1]));
    }
    return H;
  }

  // Factor the horizontal-space projector at the point whose rows are in the LDS tiles
  // (PSDFixedRank.proj, fixed_rank_psd_sym.py:91-113).  X = Y^T Y and the k^2 x k^2 system
  // are constant during one tCG solve, so they are reduced/solved once per accepted step:
  //   k = 3: C = Y^T Z - Z^T Y is skew, hence so is Omega; the 9x9 system collapses to the 3x3
  //          SPD system M o = vee(C) (same solution as :97-105 up to round-off).
  //   k = 2: the literal 4x4 matrix of :107-110 (its [1][1] entry is X01 + X00) is solved for
  //          the right-hand side [0, 1, -1, 0]; Omega = c * u.  planar_proj_exact selects the
  //          mathematically intended matrix instead.
  __device__ inline void proj_setup(int planar_proj_exact) {
    if constexpr (ANCH) {   // the anchors fix the gauge: no vertical space, proj = identity
#pragma unroll
      for (int m = 0; m < NC; ++m) Q[m] = pk[m] = pk2[m] = 0.0;
      return;
    }
    const Row<K> own = read_row(nat_off);  // natural component order (tile 0)
    const bool lead = active && comp == 0;
    if constexpr (K == 3) {
      const double y0 = own.v[0], y1 = own.v[1], y2 = own.v[2];
      const double lm = lead ? 1.0 : 0.0;
      double x[6] = {lm * y0 * y0, lm * y0 * y1, lm * y0 * y2, lm * y1 * y1, lm * y1 * y2,
                     lm * y2 * y2};
      wave_sum_n<6>(x);
      const double X00 = x[0], X01 = x[1], X02 = x[2], X11 = x[3], X12 = x[4], X22 = x[5];
      // M = [[X00+X11, X12, -X02], [X12, X00+X22, X01], [-X02, X01, X11+X22]]
      const double a = X00 + X11, b = X12, c = -X02, d = X00 + X22, e = X01, f = X11 + X22;
      const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
      const double c11 = a * f - c * c, c12 = b * c - a * e, c22 = a * d - b * b;
      const double idet = 1.0 / (a * c00 + b * c01 + c * c02);
      Pm[0] = c00 * idet; Pm[1] = c01 * idet; Pm[2] = c02 * idet;
      Pm[3] = c01 * idet; Pm[4] = c11 * idet; Pm[5] = c12 * idet;
      Pm[6] = c02 * idet; Pm[7] = c12 * idet; Pm[8] = c22 * idet;
      // <pk_q, pk_m> summed over the lanes is M itself
      G2[0] = a; G2[1] = b; G2[2] = c; G2[3] = b; G2[4] = d; G2[5] = e;
      G2[6] = c; G2[7] = e; G2[8] = f;
      // vee(C) = sum_lanes pk * Z_lane ; (Y Omega)_lane = pk . o
      //   comp 0: (-y1, -y2, 0)   comp 1: (y0, 0, -y2)   comp 2: (0, y0, y1)
      const double am = active ? 1.0 : 0.0;
      const double e0 = comp == 0 ? 1.0 : 0.0, e1 = comp == 1 ? 1.0 : 0.0,
                   e2 = comp == 2 ? 1.0 : 0.0;
      pk[0] = pk2[0] = am * (e1 * y0 - e0 * y1);
      pk[1] = pk2[1] = am * (e2 * y0 - e0 * y2);
      pk[2] = pk2[2] = am * (e2 * y1 - e1 * y2);
      vertical_basis(a, b, c, d, e, f, pk, Q);
    } else {
      const double y0 = own.v[0], y1 = own.v[1];
      const double lm = lead ? 1.0 : 0.0;
      double x[3] = {lm * y0 * y0, lm * y0 * y1, lm * y1 * y1};
      wave_sum_n<3>(x);
      const double X00 = x[0], X01 = x[1], X11 = x[2];
      double u0, u1, u2, u3;
      if (planar_proj_exact) {
        const double it = 1.0 / (X00 + X11);
        u0 = 0.0; u1 = it; u2 = -it; u3 = 0.0;
      } else {
        // rows of fixed_rank_psd_sym.py:107-110, augmented with rhs vec(C)/c = [0, 1, -1, 0]
        double A[4][5] = {{X00 + X00, X01, X01, 0.0, 0.0},
                          {X01, X01 + X00, 0.0, X01, 1.0},
                          {X01, 0.0, X00 + X11, X01, -1.0},
                          {0.0, X01, X01, X11 + X11, 0.0}};
#pragma unroll
        for (int col = 0; col < 4; ++col) {
#pragma unroll
          for (int r = col + 1; r < 4; ++r) {  // partial pivoting by compare-and-swap
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
      // c = (Y^T Z)_01 - (Z^T Y)_01 ; (Y Omega)_{i,col} = c * (Y_i0 u[col] + Y_i1 u[2+col])
      const double am = active ? 1.0 : 0.0;
      const double e0 = comp == 0 ? 1.0 : 0.0, e1 = 1.0 - e0;
      pk[0] = am * (e1 * y0 - e0 * y1);
      pk2[0] = am * (e0 * (y0 * u0 + y1 * u2) + e1 * (y0 * u1 + y1 * u3));
      Pm[0] = 1.0;
      G2[0] = wave_sum(pk2[0] * pk2[0]);
      Q[0] = 0.0;
    }
  }

  // Z - Y Omega(Z)  (fixed_rank_psd_sym.py:111-113)
  __device__ inline double proj(double Z) const {
    if constexpr (ANCH) return Z;
    double v[NC];
    if constexpr (K == 3) {
#pragma unroll
      for (int m = 0; m < NC; ++m) v[m] = Q[m] * Z;
      wave_sum_n<NC>(v);
      return fma(-Q[2], v[2], fma(-Q[1], v[1], fma(-Q[0], v[0], Z)));
    }
#pragma unroll
    for (int m = 0; m < NC; ++m) v[m] = pk[m] * Z;
    wave_sum_n<NC>(v);
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

  // rhess(x, delta) = proj(ehess(delta)) and the curvature <delta, rhess> (trust_region.py
  // :497-500) from ONE wave reduction.  With H = ehess(delta), v = <pk, H>, o = Pm v:
  //     Hdelta = H - sum_m pk2_m o_m ,   <delta, Hdelta> = <delta, H> - sum_m o_m <delta, pk2_m>.
  // s_m = <delta, pk2_m> only changes by wave-uniform scalars along the CG recurrences
  // (delta' = -r' + beta delta, r' = r + alpha Hdelta), so the caller carries it:
  //     rho_m += alpha * hd_pk_m ,  s_m = -rho_m + beta * s_m ,  rho_m(0) = <g, pk2_m> = -s_m(0)
  // with hd_pk_m = <Hdelta, pk2_m> = <H, pk2_m> - sum_q o_q <pk2_q, pk2_m> returned here.
  // Same inner product as the literal <delta, proj(ehess(delta))>, different summation order.
  __device__ inline double hess_proj_dot(double delta, const double (&s_dpk)[NC], double &d_Hd,
                                         double (&hd_pk)[NC]) {
    return proj_dot(ehess(delta), delta, s_dpk, d_Hd, hd_pk);
  }
  __device__ inline double proj_dot(double H, double delta, const double (&s_dpk)[NC], double &d_Hd,
                                    double (&hd_pk)[NC]) {
    constexpr int NV = (K == 3) ? NC + 1 : NC + 2;  // k=2: pk2 != pk needs <pk2, H> as well
    double v[NV];
#pragma unroll
    for (int m = 0; m < NC; ++m) v[m] = pk[m] * H;
    v[NC] = delta * H;
    if constexpr (K == 2) v[NC + 1] = pk2[0] * H;
    wave_sum_n<NV>(v);
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
      // non-orthogonal (literal k=2) projector: <delta, pk2> is not small, carry it exactly
      hd_pk[0] = fma(-o[0], G2[0], v[NC + 1]);
    } else {
      // k=3: proj is the orthogonal projector onto the horizontal space and delta is horizontal
      // up to round-off, so <delta, pk_m> = O(eps |delta||pk|) and the correction
      // sum_m o_m <delta, pk_m> is below the rounding error of <delta, H> itself: it is dropped
      // (<Hdelta, pk_m> = (I - M M^-1) v is pure round-off as well).
#pragma unroll
      for (int m = 0; m < NC; ++m) hd_pk[m] = 0.0;
    }
    d_Hd = dot;
    return out;
  }
};

}  // namespace gik
