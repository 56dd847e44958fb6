// graphik_amd/csrc/gik_wave.hip.h -- one-wavefront-per-IK-problem device code (gfx950 / CDNA4)
//
// Layout.  A problem is the N x k point matrix Y (N*k <= 64 unknowns).  Lane l owns the scalar
// unknown (node i = l / k, component c = l % k); every tangent vector of the trust-region
// solver (Y, grad, eta, Heta, r, delta, Hdelta) is ONE fp64 register per lane, so axpys are a
// single v_fma_f64 and Frobenius inner products are one multiply + a DPP wave reduction.
//
// The masked EDM cost couples node i to its graph neighbours.  Each lane walks its node's
// residual terms ("slots", sorted by neighbour index -- the accumulation order of the
// reference loops, costs.py:98-123,175-207) and gathers the neighbour rows of the vector being
// differentiated from LDS.  To keep the gather free of per-lane component selects the vector
// is published k times, tile c holding every row rotated left by c, so that lane (i,c) always
// finds "its" component first:  tile_c[j] = (v[j][c], v[j][c+1], v[j][c+2]).  Rows are padded
// to 48 B (k=3) so the ds_read_b128 of a 16-lane group lands on 16 distinct 16-B slots.
// Per-slot constants that stay fixed during one truncated-CG solve (sqrt(2a)*(Y_i - Y_j),
// rotated the same way, and the residual c_ij) live in registers; per-problem targets and the
// slot metadata live in LDS.  HBM is touched only at problem entry/exit.
//
// A single wavefront executes its DS instructions in order, so a ds_write followed by
// ds_reads of other lanes' data needs no barrier and no s_waitcnt in between.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "graphik_amd.h"

namespace gik {

constexpr int WAVE = 64;
constexpr int TILE_ROWS = 33;  // 32 nodes max + one dump row for idle lanes

// slot metadata word: [7:0] neighbour node j, [19:8] term index, [21:20] kind, [22] owner
__host__ __device__ inline uint32_t meta_pack(int j, int term, int kind, int owner) {
  return (uint32_t)j | ((uint32_t)term << 8) | ((uint32_t)kind << 20) | ((uint32_t)owner << 22);
}
__host__ __device__ inline int meta_j(uint32_t m) { return (int)(m & 0xffu); }
__device__ inline int meta_term(uint32_t m) { return (int)((m >> 8) & 0xfffu); }
__device__ inline int meta_kind(uint32_t m) { return (int)((m >> 20) & 3u); }
__device__ inline int meta_owner(uint32_t m) { return (int)((m >> 22) & 1u); }

// ---- cross-lane primitives ---------------------------------------------------------------
template <int CTRL>
__device__ inline double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // every lane has a valid source for the permutations used here, so `old` is never kept;
  // passing the source itself avoids a zero-initialising v_mov per half
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ inline double readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Sum NV independent per-lane values over the wavefront; result is wave-uniform (SGPR-backed).
// Four DPP butterfly stages inside each row of 16 lanes (quad_perm xor1, xor2, half-mirror,
// mirror), then the four row totals are read with v_readlane and added.  Fixed order =>
// bit-reproducible run to run.
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
__device__ inline double wave_sum(double x) {
  double v[1] = {x};
  wave_sum_n<1>(v);
  return v[0];
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
template <int K, int MAXDEG>
struct WaveCtx {
  static constexpr int RS = (K == 3) ? 6 : 2;  // LDS row stride in doubles (48 B / 16 B)
  static constexpr int NC = (K == 3) ? 3 : 1;  // independent entries of the skew matrix
  static constexpr int TILE = TILE_ROWS * RS;  // doubles per rotated tile
  // LDS carve (in doubles): K tiles | targets[T] | slot meta (MAXDEG*64 u32)
  __host__ __device__ static constexpr size_t lds_bytes(int T) {
    return sizeof(double) * ((size_t)K * TILE + (size_t)((T + 1) & ~1)) +
           sizeof(uint32_t) * (size_t)MAXDEG * WAVE;
  }

  int lane, node, comp;
  bool active;
  double *sh_tile;         // K rotated tiles
  const double *sh_tgt;    // [T] per-problem residual targets
  const uint32_t *sh_meta; // [MAXDEG][64]
  int waddr[K];            // where this lane's value goes in tile 0..K-1 (double index)
  int own_off;             // this lane's node row in its own tile (double index)
  int nat_off;             // this lane's node row in tile 0 (natural component order)
  int rowoff[MAXDEG];      // neighbour rows in this lane's tile (double index)
  double ys[MAXDEG][K];    // sqrt(2 a_ij) * (Y_i - Y_j), rotated: [0] is this lane's component
  double cc[MAXDEG];       // c_ij = sum over active residuals of (d - target)
  double pk[NC], pk2[NC], Pm[NC * NC];

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
    for (int t = 0; t < K; ++t) sh_tile[waddr[t]] = v;
#ifdef GIK_LDS_WAIT
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
#endif
    __builtin_amdgcn_wave_barrier();
  }

  __device__ inline void init(int lane_, int N, double *tiles, const double *tgt,
                              const uint32_t *meta) {
    lane = lane_;
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
    nat_off = node * RS;
#pragma unroll
    for (int s = 0; s < MAXDEG; ++s) {
      rowoff[s] = comp * TILE + meta_j(sh_meta[s * WAVE + lane]) * RS;
      cc[s] = 0.0;
#pragma unroll
      for (int q = 0; q < K; ++q) ys[s][q] = 0.0;
    }
  }

  // f(Yv): lcost / jcost (costs.py:80-93, 8-16).  Leaves the rows of Yv in the LDS tiles.
  __device__ inline double cost(double Yv) {
    put(Yv);
    const Row<K> own = read_row(own_off);
    double f = 0.0;
#pragma unroll
    for (int s = 0; s < MAXDEG; ++s) {
      const Row<K> r = read_row(rowoff[s]);
      double d = 0.0;
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const double y = own.v[q] - r.v[q];
        d = fma(y, y, d);
      }
      const uint32_t m = sh_meta[s * WAVE + lane];
      const int kind = meta_kind(m);
      const double u = sh_tgt[meta_term(m)] - d;
      // EQ: u^2 ; LOWER: max(u,0)^2 ; UPPER: max(-u,0)^2 ; counted once per term (owner lane)
      const double wp = (meta_owner(m) && (kind == GIK_TERM_EQ || kind == GIK_TERM_LOWER)) ? 1.0 : 0.0;
      const double wn = (meta_owner(m) && (kind == GIK_TERM_EQ || kind == GIK_TERM_UPPER)) ? 1.0 : 0.0;
      const double p = fmax(u, 0.0), n = fmax(-u, 0.0);
      f = fma(wp * p, p, f);
      f = fma(wn * n, n, f);
    }
    return wave_sum(f);
  }

  // Refresh the per-slot constants at the point whose rows are in the LDS tiles and return this
  // lane's entry of egrad (lgrad / jgrad, costs.py:98-123, 20-35): G_i = 2 sum_j c_ij (Y_i-Y_j).
  __device__ inline double commit() {
    const Row<K> own = read_row(own_off);
    double G = 0.0;
#pragma unroll
    for (int s = 0; s < MAXDEG; ++s) {
      const Row<K> r = read_row(rowoff[s]);
      double y[K];
      double d = 0.0;
#pragma unroll
      for (int q = 0; q < K; ++q) {
        y[q] = own.v[q] - r.v[q];
        d = fma(y[q], y[q], d);
      }
      const uint32_t m = sh_meta[s * WAVE + lane];
      const int kind = meta_kind(m);
      const double c0 = d - sh_tgt[meta_term(m)];
      // hinge active iff psi_L - d > 0 (lower) / d - psi_U > 0 (upper); equality always
      const bool act = (kind == GIK_TERM_EQ) || (kind == GIK_TERM_LOWER && c0 < 0.0) ||
                       (kind == GIK_TERM_UPPER && c0 > 0.0);
      const double c = act ? c0 : 0.0;
      const double sc = act ? 1.4142135623730951 : 0.0;  // sqrt(2 a), a in {0,1}
      cc[s] = c;
#pragma unroll
      for (int q = 0; q < K; ++q) ys[s][q] = sc * y[q];
      G = fma(c, y[0], G);
    }
    return 2.0 * G;
  }

  // ehess(Y, W) (lhess / jhess, costs.py:175-207, 39-58) with Y = last commit():
  //   H_i = 2 sum_j [ 2 a (y.w) y + c w ],  y = Y_i - Y_j,  w = W_i - W_j
  __device__ inline double ehess(double W) {
    put(W);
    const Row<K> own = read_row(own_off);
    double H = 0.0;
#pragma unroll
    for (int s = 0; s < MAXDEG; ++s) {
      const Row<K> r = read_row(rowoff[s]);
      double w[K];
#pragma unroll
      for (int q = 0; q < K; ++q) w[q] = own.v[q] - r.v[q];
      double sd = ys[s][0] * w[0];
#pragma unroll
      for (int q = 1; q < K; ++q) sd = fma(ys[s][q], w[q], sd);
      H = fma(sd, ys[s][0], fma(cc[s], w[0], H));
    }
    return 2.0 * H;
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
      // vee(C) = sum_lanes pk * Z_lane ; (Y Omega)_lane = pk . o
      //   comp 0: (-y1, -y2, 0)   comp 1: (y0, 0, -y2)   comp 2: (0, y0, y1)
      const double am = active ? 1.0 : 0.0;
      const double e0 = comp == 0 ? 1.0 : 0.0, e1 = comp == 1 ? 1.0 : 0.0,
                   e2 = comp == 2 ? 1.0 : 0.0;
      pk[0] = pk2[0] = am * (e1 * y0 - e0 * y1);
      pk[1] = pk2[1] = am * (e2 * y0 - e0 * y2);
      pk[2] = pk2[2] = am * (e2 * y1 - e1 * y2);
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
    }
  }

  // Z - Y Omega(Z)  (fixed_rank_psd_sym.py:111-113)
  __device__ inline double proj(double Z) const {
    double v[NC];
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
  // :497-500), evaluated literally: project first, then reduce delta * Hdelta.
  __device__ inline double hess_proj_dot(double delta, double &d_Hd) {
    const double Hd = proj(ehess(delta));
    d_Hd = wave_sum(delta * Hd);
    return Hd;
  }
};

}  // namespace gik
