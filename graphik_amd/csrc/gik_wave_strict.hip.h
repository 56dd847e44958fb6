// graphik_amd/csrc/gik_wave_strict.hip.h -- the wavefront kernel's Hessian product, term by term as costs.py forms it
//
// WaveCtx<3, MAXDEG>::ehess (gik_wave.hip.h) is the COLUMN form: per accepted point it caches the rows of the 3 x 3
// blocks B_ij = 2 a y y^T + c I and a product multiplies them by the neighbour's entries -- 30 multiply-adds and one
// 8-byte gather per neighbour, the cheapest rendering for one unknown per lane.  What it never forms is the scalar
//     s = (Y_i - Y_j) . (W_i - W_j)                                             (costs.py:186-203:  t = 2 s a y + c w)
// and that is what the parity residual of rounds 1-4 hangs on (NOTEBOOK 9.3): with s formed ONCE per edge
// the round-off of the Gauss-Newton part is a multiple of y -- in range(J^T) -- and truncated CG's late iterations,
// which live along the flex directions, do not see it; the kernels that do so (workgroup, node-per-lane) need +2 %
// Hessian products against the oracle where the column form needs +7 %, and end KUKA goals 2.5e-3 rad from the
// reference instead of 8.4e-3.
//
// WaveCtxStrict is the same layout -- lane l = unknown (node l / 3, component l % 3) -- with the product in the
// reference's form (gik_template_desc.hessian_form = GIK_HESS_PER_EDGE).  The three lanes of a node SHARE the work
// of the node's slot list: lane c forms s for the node's slots c, c + 3, c + 6 from whole rows in natural component
// order (the same three products in the same order at both ends of an edge: s_ij and s_ji are the same bits), the
// three lanes exchange their scalars with whole-wave DPP shifts, and every lane then adds
//     H_(i,c) = sum_slots [ s (2 a y_c) + (2 c) (W_i[c] - W_j[c]) ].
// Each lane keeps its slots in a ROTATED order -- local slot (g, k) is the node's slot ((c + g) % 3) + 3 k -- so that
// "the scalar of the lane g places further on in the triple" lands in a register with a compile-time index.
// cost() and commit() work from natural-order rows as well: the squared distance of a term is then one value, bit
// for bit, in all six lanes that hold it (the column form needs a DPP exchange for that, WaveCtx::commit).
// Per tCG step: 220 instructions instead of 188 (18 DS instead of 10, 32 DPP moves instead of 20); measured price:
// c2 -8.0 %, c4 -8.9 %.
#pragma once

#include "gik_wave.hip.h"

namespace gik {

template <int MAXDEG>
struct WaveCtxStrict : WaveCtx<3, MAXDEG, false> {
  using Base = WaveCtx<3, MAXDEG, false>;
  using SlotRec = typename Base::SlotRec;
  static constexpr int K = 3;
  static constexpr int RS = Base::RS;
  static constexpr int NSH = (MAXDEG + 2) / 3;   // slots a lane forms the scalar of
  static constexpr int LS = 3 * NSH;             // local slots (node slots beyond MAXDEG: padding)
  static constexpr bool HAS_CK = Base::HAS_CK;
  static constexpr bool AGE_PRIORITY = Base::AGE_PRIORITY;

  int natoff[LS];        // row of the neighbour of local slot sigma in tile 0 (natural order), double index
  double ysc[NSH];       // 2 a y_c      (this lane's component of the term's difference vector), own slots (g = 0)
  // ... and of the other lanes' slots, one coefficient per DIRECTION the scalar can arrive from -- lane + 1 (P: c = 0,
  // 1), lane - 2 (S: c = 2) for g = 1; lane + 2 (R: c = 0), lane - 1 (Q: c = 1, 2) for g = 2 -- zero where the
  // direction is not this lane's: the product multiplies all four shifted copies instead of bit-selecting two
  double ysP[NSH], ysS[NSH], ysR[NSH], ysQ[NSH];
  double cc[LS];         // 2 c
  double ysn[NSH][3];    // 2 a y, natural order, of the slots whose scalar this lane forms (g = 0)

  __device__ inline void init(int lane_, int N, double *tiles, const double *tgt, uint32_t *meta) {
    Base::init(lane_, N, tiles, tgt, meta);
#pragma unroll
    for (int sg = 0; sg < LS; ++sg) {
      const int g = sg / NSH, k = sg % NSH;
      int h = this->comp + g;
      h = h >= 3 ? h - 3 : h;
      const int s = h + 3 * k;
      const bool real = s < MAXDEG;
      natoff[sg] = real ? meta_j(this->sh_meta[(real ? s : 0) * WAVE + this->lane]) * RS : this->nat_off;
      cc[sg] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < NSH; ++k) ysc[k] = ysP[k] = ysS[k] = ysR[k] = ysQ[k] = ysn[k][0] = ysn[k][1] = ysn[k][2] = 0.0;
  }

  // slot record of local slot sg (its index is recomputed from the component where it is used -- once per outer
  // iteration -- instead of held in a register per slot across the tCG loop)
  __device__ inline SlotRec record(int sg) const {
    int c = this->comp;
    asm volatile("" : "+v"(c));
    int h = c + sg / NSH;
    h = h >= 3 ? h - 3 : h;
    const int s = h + 3 * (sg % NSH);
    const bool real = s < MAXDEG;
    SlotRec r = this->sh_rec[(real ? s : 0) * WAVE + this->lane];
    if (!real) {               // padding: clamp(., 0, 0) = 0, never active
      r.tg = 0.0;
      r.lo = r.hi = 0.0f;
    }
    return r;
  }
  // component `comp` of a natural-order triple, as data flow (no divergent branch)
  __device__ inline double own_comp(double v0, double v1, double v2) const {
    return bit_select(this->comp == 0, v0, bit_select(this->comp == 1, v1, v2));
  }

  // f(Yv): lcost (costs.py:80-93); leaves the rows of Yv in tile 0
  __device__ inline double cost(double Yv) {
    this->put1(Yv);
    const Row<3> own = this->read_row(this->nat_off);
    double f = 0.0;
#pragma unroll
    for (int sg = 0; sg < LS; ++sg) {
      const Row<3> r = this->read_row(natoff[sg]);
      const SlotRec rc = record(sg);
      const double y0 = own.v[0] - r.v[0], y1 = own.v[1] - r.v[1], y2 = own.v[2] - r.v[2];
      const double d = fma(y2, y2, fma(y1, y1, y0 * y0));
      const double cl = fmin(fmax(rc.tg - d, (double)rc.lo), (double)rc.hi);
      f = fma(cl, cl, f);
      if (sg % 3 == 2) __builtin_amdgcn_sched_barrier(0);
    }
    return 0.5 * wave_sum((this->active && this->comp == 0) ? f : 0.0);
  }

  // egrad at the point in tile 0 (lgrad, costs.py:98-123) + the per-slot constants of the product
  __device__ inline double commit() {
    const Row<3> own = this->read_row(this->nat_off);
    double G = 0.0;
#pragma unroll
    for (int sg = 0; sg < LS; ++sg) {
      const Row<3> r = this->read_row(natoff[sg]);
      const SlotRec rc = record(sg);
      const double y0 = own.v[0] - r.v[0], y1 = own.v[1] - r.v[1], y2 = own.v[2] - r.v[2];
      const double d = fma(y2, y2, fma(y1, y1, y0 * y0));      // natural order: one value per term in every lane
      const double cl = fmin(fmax(rc.tg - d, (double)rc.lo), (double)rc.hi);
      const bool act = (rc.lo * rc.hi < 0.0f) || (cl != 0.0);  // (see WaveCtx::commit)
      const double c = -cl;
      const double yc = own_comp(y0, y1, y2);
      const double ys = act ? yc + yc : 0.0;
      if (sg < NSH) {
        ysc[sg] = ys;
      } else if (sg < 2 * NSH) {        // holder (c + 1) % 3: its scalar arrives from lane + 1 (c = 0, 1) or lane - 2 (c = 2)
        ysP[sg - NSH] = this->comp != 2 ? ys : 0.0;
        ysS[sg - NSH] = this->comp == 2 ? ys : 0.0;
      } else {                          // holder (c + 2) % 3: from lane + 2 (c = 0) or lane - 1 (c = 1, 2)
        ysR[sg - 2 * NSH] = this->comp == 0 ? ys : 0.0;
        ysQ[sg - 2 * NSH] = this->comp != 0 ? ys : 0.0;
      }
      cc[sg] = c + c;
      G = fma(c, yc, G);
      if (sg < NSH) {      // g = 0: this lane forms the slot's scalar
        ysn[sg][0] = act ? y0 + y0 : 0.0;
        ysn[sg][1] = act ? y1 + y1 : 0.0;
        ysn[sg][2] = act ? y2 + y2 : 0.0;
      }
      if (sg % 3 == 2) __builtin_amdgcn_sched_barrier(0);
    }
    return 2.0 * G;
  }

  // ehess(Y, W) (lhess, costs.py:175-207) at the last commit(): H_i = sum_j [ (2a)^2 (y.w) y + 2 c w ]
  __device__ inline double ehess(double W) {
    this->put1(W);
    const double *tile_c = this->sh_tile + this->comp;
    const Row<3> wn = this->read_row(this->nat_off);
    Row<3> rw[NSH];
#pragma unroll
    for (int k = 0; k < NSH; ++k) rw[k] = this->read_row(natoff[k]);
    // (the own component of the whole rows as well: one more 8-byte read per slot instead of a two-level bit-select
    //  on the row -- the vector ALU, not the LDS pipe, is what this kernel runs out of)
    double wj[LS];
#pragma unroll
    for (int sg = 0; sg < LS; ++sg) wj[sg] = tile_c[natoff[sg]];
    // the scalars of this lane's slots, natural order (both ends of an edge: the same bits)
    double sc[NSH], H = 0.0;
#pragma unroll
    for (int k = 0; k < NSH; ++k) {
      const double u0 = wn.v[0] - rw[k].v[0], u1 = wn.v[1] - rw[k].v[1], u2 = wn.v[2] - rw[k].v[2];
      sc[k] = fma(ysn[k][2], u2, fma(ysn[k][1], u1, ysn[k][0] * u0));
      H = fma(sc[k], ysc[k], fma(cc[k], W - wj[k], H));
    }
    // the other two lanes' scalars: lane c needs those of the lanes one and two places on in its triple
#pragma unroll
    for (int k = 0; k < NSH; ++k) {
      const double dn1 = wave_shl<1>(sc[k]), dn2 = wave_shl<1>(dn1);     // from lane + 1, + 2
      const double up1 = wave_shr<1>(sc[k]), up2 = wave_shr<1>(up1);     // from lane - 1, - 2
      // holder (c + 1) % 3: dn1 or up2; holder (c + 2) % 3: dn2 or up1 -- the coefficient of the wrong one is zero
      H = fma(dn1, ysP[k], fma(up2, ysS[k], fma(cc[NSH + k], W - wj[NSH + k], H)));
      H = fma(dn2, ysR[k], fma(up1, ysQ[k], fma(cc[2 * NSH + k], W - wj[2 * NSH + k], H)));
    }
    return H;
  }

  __device__ inline double hess_proj_dot(double delta, const double (&s_dpk)[3], double &d_Hd, double (&hd_pk)[3]) {
    return this->proj_dot(ehess(delta), delta, s_dpk, d_Hd, hd_pk);
  }
};

}  // namespace gik
