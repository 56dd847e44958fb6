// graphik_amd/csrc/gik_wave_strict.hip.h -- the wavefront kernel's Hessian product, term by term as costs.py forms it
//
// WaveCtx<3, MAXDEG>::ehess (gik_wave.hip.h) is the COLUMN form: per accepted point it caches the rows of the 3 x 3
// blocks B_ij = 2 a y y^T + c I and a product multiplies them by the neighbour's entries.  What it never forms is the
// scalar
//     s = (Y_i - Y_j) . (W_i - W_j)                                  (costs.py:186-203:  t = 2 s a y + c w,  +t / -t)
// and that is what the parity residual of rounds 1-4 hangs on (NOTEBOOK 9.3): with s formed ONCE per edge
// the round-off of the Gauss-Newton part is a multiple of y -- in range(J^T) -- and truncated CG's late iterations,
// which live along the flex directions, do not see it.
//
// WaveCtxStrict is the same layout -- lane l = unknown (node l / 3, component l % 3) -- with the product in the
// reference's form (gik_template_desc.hessian_form = GIK_HESS_PER_EDGE, the default of 3-D wavefront graphs since
// round 6).  The three lanes of a node SPLIT the node's slot list: lane c OWNS the slots c, c + 3, c + 6 and evaluates
// those terms completely -- from whole rows in natural component order it forms
//     u = W_i - W_j,   s = (2 a y) . u,   t = s (2 a y) + (2 c) u        (all three components of t)
// exactly as the reference's loop body does, the same products in the same order at both ends of an edge (t_ij and
// t_ji are the same bits with opposite sign, as hess[idx] += c / hess[jdx] += -c) -- and accumulates a 3-vector partial
// sum over its slots.  The node's result is the sum of its three lanes' partial vectors: one transposing exchange at
// the end (lane c needs component c of all three; six whole-wave DPP shifts, the same traffic as the column form's
// exchange of its partial sums).  Round 5's first rendering let every lane accumulate its own component over ALL the
// node's slots and exchanged the nine scalars s instead (24 DPP moves, nine more column reads, 33 per-slot constants):
// 220 instructions per tCG step against the column form's 188.  This one: 12 multiply-adds per owned slot (36, column
// form 30), 1 + 8 DS instructions (column form 1 + 9), 22 for the exchange (18), 12 per-slot constants per lane.
// cost() and commit() walk only the owned slots as well (three row gathers per lane instead of nine), the gradient's
// partial vectors take the same exchange.
#pragma once

#include "gik_wave.hip.h"

namespace gik {

template <int MAXDEG>
struct WaveCtxStrict : WaveCtx<3, MAXDEG, false, true> {
  using Base = WaveCtx<3, MAXDEG, false, true>;     // (the slim LDS layout: tile 0 and the owned slots' tables only)
  using SlotRec = typename Base::SlotRec;
  static constexpr int K = 3;
  static constexpr int RS = Base::RS;
  static constexpr int NSH = Base::NSL;          // slots a lane owns (node slots beyond MAXDEG: padding)
  static constexpr bool HAS_CK = Base::HAS_CK;
  static constexpr bool AGE_PRIORITY = Base::AGE_PRIORITY;

  int natoff[NSH];       // row of the neighbour of owned slot k (node slot comp + 3 k) in tile 0, double index
  double ysn[NSH][3];    // 2 a y, natural component order (zero while the term is inactive)
  double cc[NSH];        // 2 c

  __device__ inline void init(int lane_, int N, double *tiles, const double *tgt, uint32_t *meta) {
    Base::init(lane_, N, tiles, tgt, meta);
#pragma unroll
    for (int k = 0; k < NSH; ++k) {      // (the LDS tables hold this lane's owned slots: stage_lds, SLIM)
      natoff[k] = meta_j(this->sh_meta[k * WAVE + this->lane]) * RS;
      cc[k] = ysn[k][0] = ysn[k][1] = ysn[k][2] = 0.0;
    }
  }

  // slot record of owned slot k (a padding slot -- node slot comp + 3 k beyond the node's list -- points at the
  // lane's own row with kind none: clamp(., 0, 0) = 0, never active)
  __device__ inline SlotRec record(int k) const { return this->sh_rec[k * WAVE + this->lane]; }

  // Lane (i, c) <- component c of the sum of the three partial vectors held by the lanes of node i, added in the order
  // of the holders' components.  Six whole-wave shifts; what a lane reads across the border of its triple is
  // discarded by the select (a bit-select: the shifted values exist in every lane, no divergent branch).
  __device__ inline double triple_sum(const double (&p)[3]) const {
    const double a0 = wave_shl<1>(p[0]), b0 = wave_shl<1>(a0);   // p[0] of lane + 1, lane + 2   (for c = 0)
    const double a1 = wave_shl<1>(p[1]), c1 = wave_shr<1>(p[1]); // p[1] of lane + 1, lane - 1   (for c = 1)
    const double c2 = wave_shr<1>(p[2]), d2 = wave_shr<1>(c2);   // p[2] of lane - 1, lane - 2   (for c = 2)
    const double t0 = (p[0] + a0) + b0;
    const double t1 = (c1 + p[1]) + a1;
    const double t2 = (d2 + c2) + p[2];
    return bit_select(this->comp == 0, t0, bit_select(this->comp == 1, t1, t2));
  }

  // f(Yv): lcost (costs.py:80-93); leaves the rows of Yv in tile 0.  Every term sits in the slot lists of both of
  // its nodes and every slot has exactly one owner: each term is counted twice, the total halved (exact).
  __device__ inline double cost(double Yv) {
    this->put1(Yv);
    const Row<3> own = this->read_row(this->nat_off);
    double f = 0.0;
#pragma unroll
    for (int k = 0; k < NSH; ++k) {
      const Row<3> r = this->read_row(natoff[k]);
      const SlotRec rc = record(k);
      const double y0 = own.v[0] - r.v[0], y1 = own.v[1] - r.v[1], y2 = own.v[2] - r.v[2];
      const double d = fma(y2, y2, fma(y1, y1, y0 * y0));
      const double cl = fmin(fmax(rc.tg - d, (double)rc.lo), (double)rc.hi);
      f = fma(cl, cl, f);
    }
    return 0.5 * wave_sum(this->active ? f : 0.0);
  }

  // egrad at the point in tile 0 (lgrad, costs.py:98-123: grad[idx] += t, grad[jdx] -= t) + the per-slot constants
  // of the product.  d is summed in natural order at both ends of an edge: one residual per TERM, bit for bit (what
  // keeps the gradient's round-off horizontal, see WaveCtx::commit).
  __device__ inline double commit() {
    const Row<3> own = this->read_row(this->nat_off);
    double gp[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < NSH; ++k) {
      const Row<3> r = this->read_row(natoff[k]);
      const SlotRec rc = record(k);
      const double y0 = own.v[0] - r.v[0], y1 = own.v[1] - r.v[1], y2 = own.v[2] - r.v[2];
      const double d = fma(y2, y2, fma(y1, y1, y0 * y0));
      const double cl = fmin(fmax(rc.tg - d, (double)rc.lo), (double)rc.hi);
      const bool act = (rc.lo * rc.hi < 0.0f) || (cl != 0.0);  // (see WaveCtx::commit)
      const double c = -cl;
      ysn[k][0] = act ? y0 + y0 : 0.0;
      ysn[k][1] = act ? y1 + y1 : 0.0;
      ysn[k][2] = act ? y2 + y2 : 0.0;
      cc[k] = c + c;
      gp[0] = fma(c, y0, gp[0]);
      gp[1] = fma(c, y1, gp[1]);
      gp[2] = fma(c, y2, gp[2]);
    }
    return 2.0 * triple_sum(gp);
  }

  // ehess(Y, W) (lhess, costs.py:175-207) at the last commit(): H_i = sum_j [ (2 a y . w)(2 a y) + 2 c w ], w = W_i - W_j
  // In two halves, so that the tCG loop can put the LDS round trip of the NEXT product (one write, eight reads: ~100
  // cycles in which a lone wavefront has nothing else in flight) behind the bookkeeping of the current step:
  // ehess_begin publishes the direction and issues the gathers, ehess_end consumes them (rtr_solve_one, SPLIT_EHESS).
  static constexpr bool SPLIT_EHESS = true;
  Row<3> wn_, rw_[NSH];
  __device__ inline void ehess_begin(double W) {
    this->put1(W);
    wn_ = this->read_row(this->nat_off);
#pragma unroll
    for (int k = 0; k < NSH; ++k) rw_[k] = this->read_row(natoff[k]);
  }
  __device__ inline double ehess_end() {
    double p[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < NSH; ++k) {
      const double u0 = wn_.v[0] - rw_[k].v[0], u1 = wn_.v[1] - rw_[k].v[1], u2 = wn_.v[2] - rw_[k].v[2];
      const double s = fma(ysn[k][2], u2, fma(ysn[k][1], u1, ysn[k][0] * u0));
      p[0] = fma(s, ysn[k][0], fma(cc[k], u0, p[0]));
      p[1] = fma(s, ysn[k][1], fma(cc[k], u1, p[1]));
      p[2] = fma(s, ysn[k][2], fma(cc[k], u2, p[2]));
    }
    return triple_sum(p);
  }
  __device__ inline double ehess(double W) {
    ehess_begin(W);
    return ehess_end();
  }

  __device__ inline double hess_proj_dot(double delta, const double (&s_dpk)[3], double &d_Hd, double (&hd_pk)[3]) {
    return this->proj_dot(ehess(delta), delta, s_dpk, d_Hd, hd_pk);
  }
};

}  // namespace gik
