// tools/exp/gik_wave_strict_bisect.hip.h -- developer experiment (round 6, NOTEBOOK 11.2): round 5's rendering of the
// per-edge product (every lane accumulates ITS component over all the node's slots, the scalars s are exchanged) with the
// SUMMATION GROUPING of round 6's rendering switched in piece by piece:
//   GIK_STRICT_EXP bit 0: gradient  -- three partial sums by slot owner, added in owner order, instead of one chain
//                  bit 1: product   -- likewise
//                  bit 2: cost      -- every lane sums its owned slots instead of the component-0 lane summing all
//                  bit 3: gradient  -- three partial sums added with the lane's OWN group first (a tree in the chain's order)
//                  bit 4 / 5: gradient -- one chain over the nine rounded terms in owner order / in slot-list order
// Built by tools/exp/strict_bisect.sh into gpurun_out-independent libraries (GIK_LIB_PATH), never part of the product.
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

  // group g of a lane's local slots belongs to the owner of component (comp + g) % 3: add the three groups in the
  // order of the owners' components, as round 6's triple_sum does
  __device__ inline double by_owner(const double (&p)[3]) const {
    const double c0 = (p[0] + p[1]) + p[2];      // comp 0: owners 0, 1, 2 = groups 0, 1, 2
    const double c1 = (p[2] + p[0]) + p[1];      // comp 1: owners 0, 1, 2 = groups 2, 0, 1
    const double c2 = (p[1] + p[2]) + p[0];      // comp 2: owners 0, 1, 2 = groups 1, 2, 0
    return bit_select(this->comp == 0, c0, bit_select(this->comp == 1, c1, c2));
  }

  // f(Yv): lcost (costs.py:80-93); leaves the rows of Yv in tile 0
  __device__ inline double cost(double Yv) {
    this->put1(Yv);
    const Row<3> own = this->read_row(this->nat_off);
    double f = 0.0, fg[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int sg = 0; sg < LS; ++sg) {
      const Row<3> r = this->read_row(natoff[sg]);
      const SlotRec rc = record(sg);
      const double y0 = own.v[0] - r.v[0], y1 = own.v[1] - r.v[1], y2 = own.v[2] - r.v[2];
      const double d = fma(y2, y2, fma(y1, y1, y0 * y0));
      const double cl = fmin(fmax(rc.tg - d, (double)rc.lo), (double)rc.hi);
      f = fma(cl, cl, f);
      fg[sg / NSH] = fma(cl, cl, fg[sg / NSH]);
      if (sg % 3 == 2) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr ((GIK_STRICT_EXP & 4) != 0) return 0.5 * wave_sum(this->active ? fg[0] : 0.0);
    return 0.5 * wave_sum((this->active && this->comp == 0) ? f : 0.0);
  }

  // egrad at the point in tile 0 (lgrad, costs.py:98-123) + the per-slot constants of the product
  __device__ inline double commit() {
    const Row<3> own = this->read_row(this->nat_off);
    double G = 0.0, Gg[3] = {0.0, 0.0, 0.0}, gt[LS], gy[LS];
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
      Gg[sg / NSH] = fma(c, yc, Gg[sg / NSH]);
      gt[sg] = c;
      gy[sg] = yc;
      if (sg < NSH) {      // g = 0: this lane forms the slot's scalar
        ysn[sg][0] = act ? y0 + y0 : 0.0;
        ysn[sg][1] = act ? y1 + y1 : 0.0;
        ysn[sg][2] = act ? y2 + y2 : 0.0;
      }
      if (sg % 3 == 2) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr ((GIK_STRICT_EXP & 1) != 0) return 2.0 * by_owner(Gg);
    if constexpr ((GIK_STRICT_EXP & 8) != 0) return 2.0 * ((Gg[0] + Gg[1]) + Gg[2]);     // tree, own group first
    if constexpr ((GIK_STRICT_EXP & 48) != 0) {
      // one chain over the nine terms, in owner order (16) or in the order of the node's slot list (32: the oracle's)
      double Gc = 0.0;
#pragma unroll
      for (int q = 0; q < LS; ++q) {
        // owner order: owner o = q / NSH, its k-th slot; slot order: node slot q
        const int o = (GIK_STRICT_EXP & 16) ? q / NSH : q % 3, k = (GIK_STRICT_EXP & 16) ? q % NSH : q / 3;
        double tc = 0.0, ty = 0.0;      // the term of owner o, slot k = this lane's local slot ((o - comp) mod 3) * NSH + k
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          tc = ((this->comp + g) % 3 == o) ? gt[g * NSH + k] : tc;
          ty = ((this->comp + g) % 3 == o) ? gy[g * NSH + k] : ty;
        }
        Gc = fma(tc, ty, Gc);
      }
      return 2.0 * Gc;
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
    double sc[NSH], H = 0.0, Hg[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < NSH; ++k) {
      const double u0 = wn.v[0] - rw[k].v[0], u1 = wn.v[1] - rw[k].v[1], u2 = wn.v[2] - rw[k].v[2];
      sc[k] = fma(ysn[k][2], u2, fma(ysn[k][1], u1, ysn[k][0] * u0));
      H = fma(sc[k], ysc[k], fma(cc[k], W - wj[k], H));
      Hg[0] = fma(sc[k], ysc[k], fma(cc[k], W - wj[k], Hg[0]));
    }
    // the other two lanes' scalars: lane c needs those of the lanes one and two places on in its triple
#pragma unroll
    for (int k = 0; k < NSH; ++k) {
      const double dn1 = wave_shl<1>(sc[k]), dn2 = wave_shl<1>(dn1);     // from lane + 1, + 2
      const double up1 = wave_shr<1>(sc[k]), up2 = wave_shr<1>(up1);     // from lane - 1, - 2
      // holder (c + 1) % 3: dn1 or up2; holder (c + 2) % 3: dn2 or up1 -- the coefficient of the wrong one is zero
      H = fma(dn1, ysP[k], fma(up2, ysS[k], fma(cc[NSH + k], W - wj[NSH + k], H)));
      H = fma(dn2, ysR[k], fma(up1, ysQ[k], fma(cc[2 * NSH + k], W - wj[2 * NSH + k], H)));
      // (the scalar that is not this lane's arrives with a zero coefficient: adding it first leaves the other exact)
      Hg[1] = fma(dn1, ysP[k], fma(up2, ysS[k], fma(cc[NSH + k], W - wj[NSH + k], Hg[1])));
      Hg[2] = fma(dn2, ysR[k], fma(up1, ysQ[k], fma(cc[2 * NSH + k], W - wj[2 * NSH + k], Hg[2])));
    }
    if constexpr ((GIK_STRICT_EXP & 2) != 0) return by_owner(Hg);
    return H;
  }

  __device__ inline double hess_proj_dot(double delta, const double (&s_dpk)[3], double &d_Hd, double (&hd_pk)[3]) {
    return this->proj_dot(ehess(delta), delta, s_dpk, d_Hd, hd_pk);
  }
};

}  // namespace gik
