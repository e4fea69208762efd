// csw_kernel.h -- c_sw (model/sw_core.F90:79-488) as ONE fused tile kernel per (tile, k).
//
// Reference structure: d2a2c_vect (:3006-3345) -> divergence_corner (:1740-1845) -> time-scaled
// contravariant fluxes ut, vt (:159-176) -> first-order upwind transport of delp, pt, w
// (:182-286) -> upstream kinetic energy (:297-366) -> C-grid circulation / absolute vorticity
// (:372-403) -> upwind vorticity flux and the half-step update of uc, vc (:414-486).
// All ~12 loop nests run here out of LDS: HBM sees the 5 input fields once (+ tile halos, mostly
// served by L2) and each output once.  Branches: grid_type >= 3 (no cube edges/corners).
//
// Tiles cover the index box [is-1, ie+2] x [js-1, je+2]; a tile "owns" (writes) every staggered
// point whose (i,j) lies in its cell range, clipped to the validity range the reference gives
// that output.  E(n) below = owned range expanded by n.
#pragma once

#include "fv3_common.h"
#include "tp2d_tile.h"

namespace fv3 {

struct CswArgs {
  double *delpc, *ptc, *wc, *uc, *vc, *ua, *va, *ut, *vt, *divg_d;
  const double *delp, *pt, *u, *v, *w;
  int nord, hydrostatic;
  double dt2;
  // cubed-sphere hybrid (fv3_api.hip, csw_cubed): the marching kernel runs over a whole face with the interior formulas,
  // leaves the frame of width mask_w along the face edges to the passes of cubed_csw.h, and does not form divg_d (the
  // divergence of the cubed sphere is the non-orthogonal form, computed by a pass of its own)
  int mask_w = 0;
};

template <int TI, int TJ>
struct CswTile {
  Grid g;
  CswArgs a;

  static constexpr int nSU = (TI + 5) * (TJ + 5), nSV = (TI + 5) * (TJ + 5);
  static constexpr int nUT = (TI + 5) * (TJ + 2), nVT = (TI + 2) * (TJ + 5);
  static constexpr int nC = (TI + 2) * (TJ + 2), nA1 = (TI + 1) * (TJ + 1);
  static constexpr int lds_doubles = nSU + nSV + nUT + nVT + 2 * nC + 2 * nA1 + (TI + 1) * TJ + TI * (TJ + 1) +
                                     2 * nA1 + 3 * nC;

  static void grid_dims(const Grid &g, unsigned &nbx, unsigned &nby) {
    nbx = (unsigned)((g.nx + 4 + TI - 1) / TI);
    nby = (unsigned)((g.ny + 4 + TJ - 1) / TJ);
  }

  FV3_HD void operator()(int bx, int by, int bz, int tid, double *lds) const {
    constexpr double a1 = 0.5625, a2 = -0.0625;  // sw_core.F90:53-54
    const int is = g.is, ie = g.ie, js = g.js, je = g.je;
    const int i0 = is - 1 + bx * TI, j0 = js - 1 + by * TJ;
    const int k = bz;
    const double *u = a.u + (size_t)k * g.nU(), *v = a.v + (size_t)k * g.nV();
    const double *delp = a.delp + (size_t)k * g.nA(), *pt = a.pt + (size_t)k * g.nA();
    const double *w = a.hydrostatic ? nullptr : a.w + (size_t)k * g.nA();
    const size_t oA = (size_t)k * g.nA(), oU = (size_t)k * g.nU(), oV = (size_t)k * g.nV(), oB = (size_t)k * g.nB();
    const double dt2 = a.dt2;

    double *p = lds;
    const Tile su{p, i0 - 3, j0 - 2, TI + 5};   p += nSU;   // u  on [i0-3,i0+TI+1] x [j0-2,j0+TJ+2]
    const Tile sv{p, i0 - 2, j0 - 3, TI + 5};   p += nSV;   // v  on [i0-2,i0+TI+2] x [j0-3,j0+TJ+1]
    const Tile sutmp{p, i0 - 3, j0 - 1, TI + 5}; p += nUT;  // utmp [i0-3,i0+TI+1] x [j0-1,j0+TJ]
    const Tile svtmp{p, i0 - 1, j0 - 3, TI + 2}; p += nVT;  // vtmp [i0-1,i0+TI] x [j0-3,j0+TJ+1]
    const Tile suc{p, i0 - 1, j0 - 1, TI + 2};  p += nC;    // uc (interp) on E(1)
    const Tile svc{p, i0 - 1, j0 - 1, TI + 2};  p += nC;
    const Tile sua{p, i0 - 1, j0 - 1, TI + 1};  p += nA1;   // ua, va on [i0-1,i0+TI-1] x [j0-1,j0+TJ-1]
    const Tile sva{p, i0 - 1, j0 - 1, TI + 1};  p += nA1;
    const Tile sut{p, i0, j0, TI + 1};          p += (TI + 1) * TJ;  // scaled ut on [i0,i0+TI] x [j0,j0+TJ-1]
    const Tile svt{p, i0, j0, TI};              p += TI * (TJ + 1);  // scaled vt on [i0,i0+TI-1] x [j0,j0+TJ]
    const Tile ske{p, i0 - 1, j0 - 1, TI + 1};  p += nA1;   // ke on [i0-1,i0+TI-1] x [j0-1,j0+TJ-1]
    const Tile svo{p, i0, j0, TI + 1};          p += nA1;   // abs. vorticity at corners [i0,i0+TI] x [j0,j0+TJ]
    const Tile sdp{p, i0 - 1, j0 - 1, TI + 2};  p += nC;    // delp, pt, w on E(1)
    const Tile spt{p, i0 - 1, j0 - 1, TI + 2};  p += nC;
    const Tile sw{p, i0 - 1, j0 - 1, TI + 2};   p += nC;

    // ---- P0: stage inputs ---------------------------------------------------------------
    load_tile<TI + 5, TJ + 5>(su, u, g.nid, g.isd, g.ied, g.jsd, g.jed + 1, tid);
    load_tile<TI + 5, TJ + 5>(sv, v, g.nid + 1, g.isd, g.ied + 1, g.jsd, g.jed, tid);
    load_tile<TI + 2, TJ + 2>(sdp, delp, g.nid, g.isd, g.ied, g.jsd, g.jed, tid);
    load_tile<TI + 2, TJ + 2>(spt, pt, g.nid, g.isd, g.ied, g.jsd, g.jed, tid);
    if (w) load_tile<TI + 2, TJ + 2>(sw, w, g.nid, g.isd, g.ied, g.jsd, g.jed, tid);
    FV3_SYNC();

    // ---- P1: D -> A interpolation (d2a2c_vect :3099-3108), divergence at corners ----------
    FV3_TILE_FOR((TI + 5), (nUT) / (TI + 5), li_, lj_) {
      const int i = i0 - 3 + li_, j = j0 - 1 + lj_;
      double val = 0.;
      if (i >= g.isd && i <= g.ied && j >= js - 1 && j <= je + 1)
        val = a2 * (su(i, j - 1) + su(i, j + 2)) + a1 * (su(i, j) + su(i, j + 1));
      sutmp(i, j) = val;
    }
    FV3_TILE_FOR((TI + 2), (nVT) / (TI + 2), li_, lj_) {
      const int i = i0 - 1 + li_, j = j0 - 3 + lj_;
      double val = 0.;
      if (j >= g.jsd && j <= g.jed && i >= is - 1 && i <= ie + 1)
        val = a2 * (sv(i - 1, j) + sv(i + 2, j)) + a1 * (sv(i, j) + sv(i + 1, j));
      svtmp(i, j) = val;
    }
    if (a.nord > 0) {  // divergence_corner, grid_type > 3 branch (:1781-1796)
      FV3_TILE_FOR(TI, (TI * TJ) / TI, li_, lj_) {
        const int i = i0 + li_, j = j0 + lj_;
        if (i > ie + 2 || j > je + 2) continue;
        const double uf0 = su(i - 1, j) * g.dyc[g.iU(i - 1, j)], uf1 = su(i, j) * g.dyc[g.iU(i, j)];
        const double vf0 = sv(i, j - 1) * g.dxc[g.iV(i, j - 1)], vf1 = sv(i, j) * g.dxc[g.iV(i, j)];
        a.divg_d[oB + g.iB(i, j)] = g.rarea_c[g.iB(i, j)] * (vf0 - vf1 + uf0 - uf1);
      }
    }
    FV3_SYNC();

    // ---- P2: A-grid contravariant winds (:3152-3157); A -> C interpolation (:3197-3202,3337-3342)
    FV3_TILE_FOR((TI + 1), (nA1) / (TI + 1), li_, lj_) {
      const int i = i0 - 1 + li_, j = j0 - 1 + lj_;
      double uav = 0., vav = 0.;
      if (i >= is - 1 && i <= ie + 1 && j >= js - 1 && j <= je + 1) {
        const double cs = g.cosa_s[g.iA(i, j)], rs = g.rsin2[g.iA(i, j)];
        uav = (sutmp(i, j) - svtmp(i, j) * cs) * rs;
        vav = (svtmp(i, j) - sutmp(i, j) * cs) * rs;
        if (i >= i0 && j >= j0) {  // owned
          a.ua[oA + g.iA(i, j)] = uav;
          a.va[oA + g.iA(i, j)] = vav;
        }
      }
      sua(i, j) = uav;
      sva(i, j) = vav;
    }
    FV3_TILE_FOR((TI + 2), (nC) / (TI + 2), li_, lj_) {
      const int i = i0 - 1 + li_, j = j0 - 1 + lj_;
      double ucv = 0., vcv = 0.;
      if (i >= is - 1 && i <= ie + 2 && j >= js - 1 && j <= je + 1)
        ucv = a2 * (sutmp(i - 2, j) + sutmp(i + 1, j)) + a1 * (sutmp(i - 1, j) + sutmp(i, j));
      if (i >= is - 1 && i <= ie + 1 && j >= js - 1 && j <= je + 2)
        vcv = a2 * (svtmp(i, j - 2) + svtmp(i, j + 1)) + a1 * (svtmp(i, j - 1) + svtmp(i, j));
      suc(i, j) = ucv;
      svc(i, j) = vcv;
    }
    FV3_SYNC();

    // ---- P3: time-scaled fluxes ut, vt (:159-176); KE (:297-315,361-366); abs. vorticity (:372-403)
    FV3_TILE_FOR((TI + 1), ((TI + 1) * TJ) / (TI + 1), li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      double val = 0.;
      if (i <= ie + 2 && j <= je + 1) {
        // d2a2c: ut = (uc - v*cosa_u)*rsin_u (:3200)
        val = (suc(i, j) - sv(i, j) * g.cosa_u[g.iV(i, j)]) * g.rsin_u[g.iV(i, j)];
        if (val > 0.)
          val = dt2 * val * g.dy[g.iV(i, j)] * g.sinsg(i - 1, j, 3);
        else
          val = dt2 * val * g.dy[g.iV(i, j)] * g.sinsg(i, j, 1);
        if (i < i0 + TI) a.ut[oA + g.iA(i, j)] = val;  // owned
      }
      sut(i, j) = val;
    }
    FV3_TILE_FOR(TI, (TI * (TJ + 1)) / TI, li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      double val = 0.;
      if (i <= ie + 1 && j <= je + 2) {
        val = svc(i, j);  // grid_type >= 3: vt = vc (:3340)
        if (val > 0.)
          val = dt2 * val * g.dx[g.iU(i, j)] * g.sinsg(i, j - 1, 4);
        else
          val = dt2 * val * g.dx[g.iU(i, j)] * g.sinsg(i, j, 2);
        if (j < j0 + TJ) a.vt[oA + g.iA(i, j)] = val;
      }
      svt(i, j) = val;
    }
    {
      const double dt4 = 0.5 * dt2;
      FV3_TILE_FOR((TI + 1), (nA1) / (TI + 1), li_, lj_) {
        const int i = i0 - 1 + li_, j = j0 - 1 + lj_;
        double kev = 0.;
        if (i >= is - 1 && i <= ie + 1 && j >= js - 1 && j <= je + 1) {
          const double uav = sua(i, j), vav = sva(i, j);
          const double k1 = (uav > 0.) ? suc(i, j) : suc(i + 1, j);
          const double k2 = (vav > 0.) ? svc(i, j) : svc(i, j + 1);
          kev = dt4 * (uav * k1 + vav * k2);
        }
        ske(i, j) = kev;
      }
    }
    FV3_TILE_FOR((TI + 1), (nA1) / (TI + 1), li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      double vo = 0.;
      if (i >= is && i <= ie + 1 && j >= js && j <= je + 1) {
        const double fxm = suc(i, j - 1) * g.dxc[g.iV(i, j - 1)], fx0 = suc(i, j) * g.dxc[g.iV(i, j)];
        const double fym = svc(i - 1, j) * g.dyc[g.iU(i - 1, j)], fy0 = svc(i, j) * g.dyc[g.iU(i, j)];
        vo = fxm - fx0 - fym + fy0;
        vo = g.fC[g.iB(i, j)] + g.rarea_c[g.iB(i, j)] * vo;
      }
      svo(i, j) = vo;
    }
    FV3_SYNC();

    // ---- P4: owned cells: upwind transport (:182-286) and the C-grid wind update (:414-486) ----
    FV3_TILE_FOR(TI, (TI * TJ) / TI, li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      if (i > ie + 2 || j > je + 2) continue;
      if (i <= ie + 1 && j <= je + 1) {
        const double ut0 = sut(i, j), ut1 = sut(i + 1, j), vt0 = svt(i, j), vt1 = svt(i, j + 1);
        const double ra = g.rarea[g.iA(i, j)];
        // x faces
        const int iu0 = (ut0 > 0.) ? i - 1 : i, iu1 = (ut1 > 0.) ? i : i + 1;
        const int ju0 = (vt0 > 0.) ? j - 1 : j, ju1 = (vt1 > 0.) ? j : j + 1;
        const double fx1_0 = ut0 * sdp(iu0, j), fx1_1 = ut1 * sdp(iu1, j);
        const double fy1_0 = vt0 * sdp(i, ju0), fy1_1 = vt1 * sdp(i, ju1);
        const double fx_0 = fx1_0 * spt(iu0, j), fx_1 = fx1_1 * spt(iu1, j);
        const double fy_0 = fy1_0 * spt(i, ju0), fy_1 = fy1_1 * spt(i, ju1);
        const double dpc = sdp(i, j) + (fx1_0 - fx1_1 + fy1_0 - fy1_1) * ra;
        a.delpc[oA + g.iA(i, j)] = dpc;
        a.ptc[oA + g.iA(i, j)] = (spt(i, j) * sdp(i, j) + (fx_0 - fx_1 + fy_0 - fy_1) * ra) / dpc;
        if (w) {
          const double fx2_0 = fx1_0 * sw(iu0, j), fx2_1 = fx1_1 * sw(iu1, j);
          const double fy2_0 = fy1_0 * sw(i, ju0), fy2_1 = fy1_1 * sw(i, ju1);
          a.wc[oA + g.iA(i, j)] = (sw(i, j) * sdp(i, j) + (fx2_0 - fx2_1 + fy2_0 - fy2_1) * ra) / dpc;
        }
      }
      // uc: interpolated value on [is-1,ie+2] x [js-1,je+1], advanced on [is,ie+1] x [js,je]
      if (j <= je + 1) {
        double ucv = suc(i, j);
        if (i >= is && i <= ie + 1 && j >= js && j <= je) {
          const double fy1 = dt2 * (sv(i, j) - ucv * g.cosa_u[g.iV(i, j)]) / g.sina_u[g.iV(i, j)];
          const double fy = (fy1 > 0.) ? svo(i, j) : svo(i, j + 1);
          ucv = ucv + fy1 * fy + g.rdxc[g.iV(i, j)] * (ske(i - 1, j) - ske(i, j));
        }
        a.uc[oV + g.iV(i, j)] = ucv;
      }
      if (i <= ie + 1) {
        double vcv = svc(i, j);
        if (i >= is && i <= ie && j >= js && j <= je + 1) {
          const double fx1 = dt2 * (su(i, j) - vcv * g.cosa_v[g.iU(i, j)]) / g.sina_v[g.iU(i, j)];
          const double fx = (fx1 > 0.) ? svo(i, j) : svo(i + 1, j);
          vcv = vcv - fx1 * fx + g.rdyc[g.iU(i, j)] * (ske(i, j - 1) - ske(i, j));
        }
        a.vc[oU + g.iU(i, j)] = vcv;
      }
    }
  }
};

}  // namespace fv3
