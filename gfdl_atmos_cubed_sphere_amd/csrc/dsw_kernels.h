// dsw_kernels.h -- d_sw (model/sw_core.F90:494-1606) as three kernels per call (all levels):
//
//   DswCourant  : contravariant winds -> Courant numbers / area fluxes crx, xfx, cry, yfx and the
//                 accumulation of cx, cy (:850-902, :923-927, :933-936).  Pointwise.
//   DswTransport: fv_tp_2d of delp -> mass fluxes (-> mfx, mfy), then fv_tp_2d of w, q_con, pt with
//                 those mass fluxes, del-2n damping of w, and the flux-form update of delp, pt, w,
//                 q_con (:908-1066, :1249-1283).  One tile kernel; every intermediate of the 3-4
//                 Lin-Rood transports lives in LDS.
//   DswMomentum : B-grid KE fluxes via ytp_v/xtp_u (:1078-1198), relative vorticity (:1231-1247),
//                 divergence damping (:1290-1460), fv_tp_2d of absolute vorticity and the D-grid wind
//                 update (:1476-1509), vorticity damping and dissipative heating (:1513-1600).
//
// Branches: grid_type >= 3, inline_q = .false., do_f3d = .false.; nord_* <= 2.
// The in-place fields get *_out buffers (see include/fv3_mi355x.h): tiles of the same launch still
// read the old halo of their neighbours.
#pragma once

#include "fv3_common.h"
#include "tp2d_tile.h"

namespace fv3 {

// per-level coefficients (device arrays of length npz), dyn_core.F90:666-733
struct DswLevels {
  const int *nord_k, *nord_v, *nord_w, *nord_t;
  const double *d2_divg, *damp_vt, *damp_w, *damp_t, *d_con_k;
};

struct DswArgs {
  double dt;
  int hord_tr, hord_mt, hord_vt, hord_tm, hord_dp;
  double dddmp, d4_bg, kgb;
  int hydrostatic, use_cond;
  DswLevels lv;
  const double *delp, *pt, *u, *v, *w, *uc, *vc, *ua, *va, *divg_d, *q_con;
  double *mfx, *mfy, *cx, *cy, *crx, *cry, *xfx, *yfx;
  double *delp_out, *pt_out, *u_out, *v_out, *w_out, *q_con_out, *heat_s, *diss_e, *delpc;
  // cubed-sphere hybrid (fv3_api.hip, dsw_cubed): the marching kernels run over a whole face with the interior formulas and
  // keep their hands off the frame of width mask_w along the face edges (i <= mask_w, i >= npx - mask_w, likewise j), which
  // the pass kernels of cubed_dsw.h own.  rsina: (is:ie+1, js:je+1), for the B-grid winds of the kinetic energy.
  int mask_w = 0;
  const double *rsina = nullptr;
  // the caller passed heat_s / diss_e = NULL (nobody reads them) and heat_s / diss_e above are the context's scratch for the kernels that
  // need real arrays (the damped levels' passes): the branch-free marching kernel does not store its zeros
  int skip_heat = 0;
  // DswTransportFused<..., FLUXES = true> (the damped levels of a cubed-sphere face): the fluxes themselves are the result -- delp's
  // (with the del-2n fluxes dfx / dfy of delp added on the levels with dcoef > 1e-4, V / U layout: cubed_damp.h), w's and pt's
  // weighted with them -- and the fields are left to the pass that also applies the other damping terms (cubed_dsw.h D4)
  const double *dfx = nullptr, *dfy = nullptr, *dcoef = nullptr;
  double *ofx = nullptr, *ofy = nullptr, *ogxw = nullptr, *ogyw = nullptr, *ogx = nullptr, *ogy = nullptr;
};

// ------------------------------------------------------------------------------------------------
struct DswCourant {
  Grid g;
  DswArgs a;
  const int *klist;  // level of the bz-th slab, or null = identity
  static constexpr int CH = 1024;  // points per workgroup
  FV3_HD void operator()(int bx, int /*by*/, int bz, int tid, double * /*lds*/) const {
    const int k = klist ? klist[bz] : bz;
    const double dt = a.dt;
    const size_t oV = (size_t)k * g.nV(), oU = (size_t)k * g.nU();
    const size_t oCX = (size_t)k * g.nCX(), oCY = (size_t)k * g.nCY();
    const int ncx = (int)g.nCX(), ncy = (int)g.nCY();
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH; idx += kNT) {
      if (idx < ncx) {  // (is:ie+1, jsd:jed)
        const int i = g.is + idx % (g.nx + 1), j = g.jsd + idx / (g.nx + 1);
        const double ut = a.uc[oV + g.iV(i, j)];  // grid_type >= 3: ut = uc (:850-854)
        double x = dt * ut;                        // :865
        double cr;
        if (x > 0.) {  // :882-888
          cr = x * g.rdxa[g.iA(i - 1, j)];
          x = g.dy[g.iV(i, j)] * x * g.sinsg(i - 1, j, 3);
        } else {
          cr = x * g.rdxa[g.iA(i, j)];
          x = g.dy[g.iV(i, j)] * x * g.sinsg(i, j, 1);
        }
        a.crx[oCX + idx] = cr;
        a.xfx[oCX + idx] = x;
        a.cx[oCX + idx] = a.cx[oCX + idx] + cr;  // :923-927
      }
      if (idx < ncy) {  // (isd:ied, js:je+1)
        const int i = g.isd + idx % g.nid, j = g.js + idx / g.nid;
        const double vt = a.vc[oU + g.iU(i, j)];
        double y = dt * vt;
        double cr;
        if (y > 0.) {  // :894-900
          cr = y * g.rdya[g.iA(i, j - 1)];
          y = g.dx[g.iU(i, j)] * y * g.sinsg(i, j - 1, 4);
        } else {
          cr = y * g.rdya[g.iA(i, j)];
          y = g.dx[g.iU(i, j)] * y * g.sinsg(i, j, 2);
        }
        a.cry[oCY + idx] = cr;
        a.yfx[oCY + idx] = y;
        a.cy[oCY + idx] = a.cy[oCY + idx] + cr;  // :933-936
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------
template <int TI, int TJ>
struct DswTransport {
  Grid g;
  DswArgs a;
  const int *klist;  // level of the bz-th slab, or null = identity
  using TS = Tp2dScratch<TI, TJ>;
  using DS = DelnScratch<TI, TJ>;
  static constexpr int nQ = (TI + 6) * (TJ + 6);
  static constexpr int nFXt = (TI + 1) * TJ, nFYt = TI * (TJ + 1), nCell = TI * TJ;
  static constexpr int nScr = TS::total > DS::total ? TS::total : DS::total;
  // q tile, mass tile (delp, kept), scratch, face values, mass fluxes, 3 per-cell accumulators
  static constexpr int lds_doubles = 2 * nQ + nScr + 2 * (nFXt + nFYt) + 3 * nCell;

  static void grid_dims(const Grid &g, unsigned &nbx, unsigned &nby) {
    nbx = (unsigned)((g.nx + TI - 1) / TI);
    nby = (unsigned)((g.ny + TJ - 1) / TJ);
  }

  // add the del-n diffusive fluxes of field sq to (sfx, sfy): deln_flux, tp_core.F90:1301-1445
  FV3_HD void add_deln(const TileBox &b, int tid, const Tile &sq, const Tile *smass, int nord, double damp_c,
                       double *scr, const Tile &sfx, const Tile &sfy) const {
    const double damp = ipow(damp_c * g.da_min, nord + 1);  // tp_core.F90:203,229
    Tile fxd, fyd;
    deln_tile<TI, TJ>(g, b, tid, sq, nord, damp, smass == nullptr, scr, fxd, fyd);
    const double damp2 = 0.5 * damp;
    FV3_TILE_FOR((TI + 1), (nFXt) / (TI + 1), li_, lj_) {
      const int i = b.i0 + li_, j = b.j0 + lj_;
      if (i > b.ilast + 1 || j > b.jlast) continue;
      if (smass)
        sfx(i, j) = sfx(i, j) + damp2 * ((*smass)(i - 1, j) + (*smass)(i, j)) * fxd(i, j);
      else
        sfx(i, j) = sfx(i, j) + fxd(i, j);
    }
    FV3_TILE_FOR(TI, (nFYt) / TI, li_, lj_) {
      const int i = b.i0 + li_, j = b.j0 + lj_;
      if (i > b.ilast || j > b.jlast + 1) continue;
      if (smass)
        sfy(i, j) = sfy(i, j) + damp2 * ((*smass)(i, j - 1) + (*smass)(i, j)) * fyd(i, j);
      else
        sfy(i, j) = sfy(i, j) + fyd(i, j);
    }
    FV3_SYNC();
  }

  FV3_HD void operator()(int bx, int by, int bz, int tid, double *lds) const {
    const int k = klist ? klist[bz] : bz;
    const TileBox b = make_box<TI, TJ>(g, bx, by);
    const int i0 = b.i0, j0 = b.j0;
    const size_t oA = (size_t)k * g.nA(), oCX = (size_t)k * g.nCX(), oCY = (size_t)k * g.nCY();
    const size_t oFX = (size_t)k * g.nFX(), oFY = (size_t)k * g.nFY(), oCC = (size_t)k * g.nCC();
    const double *crx = a.crx + oCX, *xfx = a.xfx + oCX, *cry = a.cry + oCY, *yfx = a.yfx + oCY;
    const int nord_v = a.lv.nord_v[k], nord_w = a.lv.nord_w[k], nord_t = a.lv.nord_t[k];
    const double damp_v = a.lv.damp_vt[k], damp_w = a.lv.damp_w[k], damp_t = a.lv.damp_t[k];

    double *p = lds;
    const Tile sq{p, i0 - 3, j0 - 3, TI + 6};   p += nQ;    // field being transported
    const Tile sdp{p, i0 - 3, j0 - 3, TI + 6};  p += nQ;    // delp (mass), kept for the whole kernel
    double *scr = p;                            p += nScr;
    const Tile sfx{p, i0, j0, TI + 1};          p += nFXt;  // face values / fluxes of the current field
    const Tile sfy{p, i0, j0, TI};              p += nFYt;
    const Tile smx{p, i0, j0, TI + 1};          p += nFXt;  // delp mass fluxes fx, fy (:919)
    const Tile smy{p, i0, j0, TI};              p += nFYt;
    const Tile cw{p, i0, j0, TI};               p += nCell; // delp*w + div(gx,gy)*rarea
    const Tile cdw{p, i0, j0, TI};              p += nCell; // dw
    const Tile cq{p, i0, j0, TI};               p += nCell; // delp*q_con + div*rarea

    // ---- delp: fv_tp_2d(delp, ..., hord_dp, nord=nord_v, damp_c=damp_v)  (:919-920) ------------
    load_tile<TI + 6, TJ + 6>(sdp, a.delp + oA, g.nid, g.isd, g.ied, g.jsd, g.jed, tid);
    FV3_SYNC();
    tp2d_tile<TI, TJ>(g, b, tid, sdp, crx, cry, xfx, yfx, nullptr, nullptr, a.hord_dp, scr, smx, smy);
    FV3_TILE_FOR((TI + 1), (nFXt) / (TI + 1), li_, lj_) {  // tp_core.F90:217-226
      const int i = i0 + li_, j = j0 + lj_;
      if (i > b.ilast + 1 || j > b.jlast) continue;
      smx(i, j) = smx(i, j) * xfx[g.iCX(i, j)];
    }
    FV3_TILE_FOR(TI, (nFYt) / TI, li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      if (i > b.ilast || j > b.jlast + 1) continue;
      smy(i, j) = smy(i, j) * yfx[g.iCY(i, j)];
    }
    FV3_SYNC();
    if (damp_v > 1.E-4) add_deln(b, tid, sdp, nullptr, nord_v, damp_v, scr, smx, smy);
    // flux capacitors (:928-940); a face is accumulated by the tile that owns its cell, the last
    // face of the domain (ie+1 / je+1) by the last tile.
    FV3_TILE_FOR((TI + 1), (nFXt) / (TI + 1), li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      if (i > b.ilast + 1 || j > b.jlast) continue;
      if (i == i0 + TI && i <= g.ie) continue;  // owned by the next tile
      a.mfx[oFX + g.iFX(i, j)] = a.mfx[oFX + g.iFX(i, j)] + smx(i, j);
    }
    FV3_TILE_FOR(TI, (nFYt) / TI, li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      if (i > b.ilast || j > b.jlast + 1) continue;
      if (j == j0 + TJ && j <= g.je) continue;
      a.mfy[oFY + g.iFY(i, j)] = a.mfy[oFY + g.iFY(i, j)] + smy(i, j);
    }

    // heat_source = diss_est = 0 (:943-948)
    FV3_TILE_FOR(TI, (nCell) / TI, li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      if (i > b.ilast || j > b.jlast) continue;
      a.heat_s[oCC + g.iCC(i, j)] = 0.;
      a.diss_e[oCC + g.iCC(i, j)] = 0.;
      cdw(i, j) = 0.;
    }

    // ---- w (:950-990) ------------------------------------------------------------------------
    if (!a.hydrostatic) {
      load_tile<TI + 6, TJ + 6>(sq, a.w + oA, g.nid, g.isd, g.ied, g.jsd, g.jed, tid);
      FV3_SYNC();
      if (damp_w > 1.E-5) {
        const double dd8 = a.kgb * fabs(a.dt);
        const double damp4 = ipow(damp_w * g.da_min_c, nord_w + 1);
        Tile fxd, fyd;
        deln_tile<TI, TJ>(g, b, tid, sq, nord_w, damp4, true, scr, fxd, fyd);
        FV3_TILE_FOR(TI, (nCell) / TI, li_, lj_) {
          const int i = i0 + li_, j = j0 + lj_;
          if (i > b.ilast || j > b.jlast) continue;
          const double dw = (fxd(i, j) - fxd(i + 1, j) + fyd(i, j) - fyd(i, j + 1)) * g.rarea[g.iA(i, j)];
          cdw(i, j) = dw;
          const double tmp = dw * (sq(i, j) + 0.5 * dw);
          if (g.prevent_diss_cooling) {
            a.heat_s[oCC + g.iCC(i, j)] = dd8 - dmin(0., tmp);
            if (g.do_diss_est) a.diss_e[oCC + g.iCC(i, j)] = dd8 - tmp;
          } else {
            const double hs = dd8 - dw * (sq(i, j) + 0.5 * dw);
            a.heat_s[oCC + g.iCC(i, j)] = hs;
            if (g.do_diss_est) a.diss_e[oCC + g.iCC(i, j)] = hs;
          }
        }
        FV3_SYNC();
      }
      tp2d_tile<TI, TJ>(g, b, tid, sq, crx, cry, xfx, yfx, nullptr, nullptr, a.hord_vt, scr, sfx, sfy);
      FV3_TILE_FOR(TI, (nCell) / TI, li_, lj_) {  // :985-989 with tp_core.F90:191-200
        const int i = i0 + li_, j = j0 + lj_;
        if (i > b.ilast || j > b.jlast) continue;
        const double gx0 = sfx(i, j) * smx(i, j), gx1 = sfx(i + 1, j) * smx(i + 1, j);
        const double gy0 = sfy(i, j) * smy(i, j), gy1 = sfy(i, j + 1) * smy(i, j + 1);
        cw(i, j) = sdp(i, j) * sq(i, j) + (gx0 - gx1 + gy0 - gy1) * g.rarea[g.iA(i, j)];
      }
      FV3_SYNC();
    }

    // ---- q_con (:992-1000) ---------------------------------------------------------------------
    if (a.use_cond) {
      load_tile<TI + 6, TJ + 6>(sq, a.q_con + oA, g.nid, g.isd, g.ied, g.jsd, g.jed, tid);
      FV3_SYNC();
      tp2d_tile<TI, TJ>(g, b, tid, sq, crx, cry, xfx, yfx, nullptr, nullptr, a.hord_dp, scr, sfx, sfy);
      FV3_TILE_FOR((TI + 1), (nFXt) / (TI + 1), li_, lj_) {
        const int i = i0 + li_, j = j0 + lj_;
        if (i > b.ilast + 1 || j > b.jlast) continue;
        sfx(i, j) = sfx(i, j) * smx(i, j);
      }
      FV3_TILE_FOR(TI, (nFYt) / TI, li_, lj_) {
        const int i = i0 + li_, j = j0 + lj_;
        if (i > b.ilast || j > b.jlast + 1) continue;
        sfy(i, j) = sfy(i, j) * smy(i, j);
      }
      FV3_SYNC();
      if (damp_t > 1.e-4) add_deln(b, tid, sq, &sdp, nord_t, damp_t, scr, sfx, sfy);
      FV3_TILE_FOR(TI, (nCell) / TI, li_, lj_) {
        const int i = i0 + li_, j = j0 + lj_;
        if (i > b.ilast || j > b.jlast) continue;
        cq(i, j) = sdp(i, j) * sq(i, j) +
                   (sfx(i, j) - sfx(i + 1, j) + sfy(i, j) - sfy(i, j + 1)) * g.rarea[g.iA(i, j)];
      }
      FV3_SYNC();
    }

    // ---- pt (:1014-1016) and the flux-form update (:1053-1066, :1262-1283) -----------------------
    load_tile<TI + 6, TJ + 6>(sq, a.pt + oA, g.nid, g.isd, g.ied, g.jsd, g.jed, tid);
    FV3_SYNC();
    tp2d_tile<TI, TJ>(g, b, tid, sq, crx, cry, xfx, yfx, nullptr, nullptr, a.hord_tm, scr, sfx, sfy);
    FV3_TILE_FOR((TI + 1), (nFXt) / (TI + 1), li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      if (i > b.ilast + 1 || j > b.jlast) continue;
      sfx(i, j) = sfx(i, j) * smx(i, j);
    }
    FV3_TILE_FOR(TI, (nFYt) / TI, li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      if (i > b.ilast || j > b.jlast + 1) continue;
      sfy(i, j) = sfy(i, j) * smy(i, j);
    }
    FV3_SYNC();
    if (damp_t > 1.e-4) add_deln(b, tid, sq, &sdp, nord_t, damp_t, scr, sfx, sfy);
    FV3_TILE_FOR(TI, (nCell) / TI, li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      if (i > b.ilast || j > b.jlast) continue;
      const double ra = g.rarea[g.iA(i, j)];
      double ptn = sq(i, j) * sdp(i, j) + (sfx(i, j) - sfx(i + 1, j) + sfy(i, j) - sfy(i, j + 1)) * ra;
      const double dpn = sdp(i, j) + (smx(i, j) - smx(i + 1, j) + smy(i, j) - smy(i, j + 1)) * ra;
      ptn = ptn / dpn;
      a.pt_out[oA + g.iA(i, j)] = ptn;
      a.delp_out[oA + g.iA(i, j)] = dpn;
      if (!a.hydrostatic) {
        double wn = cw(i, j) / dpn;                 // :1264
        if (damp_w > 1.E-5) wn = wn + cdw(i, j);    // :1268-1274
        a.w_out[oA + g.iA(i, j)] = wn;
      }
      if (a.use_cond) a.q_con_out[oA + g.iA(i, j)] = cq(i, j) / dpn;  // :1280
    }
  }
};

// ------------------------------------------------------------------------------------------------
template <int TI, int TJ>
struct DswMomentum {
  Grid g;
  DswArgs a;
  const int *klist;  // level of the bz-th slab, or null = identity
  using TS = Tp2dScratch<TI, TJ>;
  using DS = DelnScratch<TI, TJ>;
  static constexpr int nSU = (TI + 6) * (TJ + 7), nSV = (TI + 7) * (TJ + 6), nQ = (TI + 6) * (TJ + 6);
  static constexpr int nScr = TS::total > DS::total ? TS::total : DS::total;
  static constexpr int nFXt = (TI + 1) * TJ, nFYt = TI * (TJ + 1);
  static constexpr int nKE = (TI + 2) * (TJ + 2);    // corners [i0, i0+TI+1] x [j0, j0+TJ+1]
  static constexpr int nDV = (TI + 8) * (TJ + 8);    // divg_d work copy on corners E(3)+1
  static constexpr int lds_doubles = nSU + nSV + 2 * nQ + nScr + nFXt + nFYt + 2 * nKE + 3 * nDV;

  static void grid_dims(const Grid &g, unsigned &nbx, unsigned &nby) {
    nbx = (unsigned)((g.nx + TI - 1) / TI);
    nby = (unsigned)((g.ny + TJ - 1) / TJ);
  }

  FV3_HD void operator()(int bx, int by, int bz, int tid, double *lds) const {
    constexpr double a1 = 0.5625, a2 = -0.0625, b1 = 7. / 12., b2 = -1. / 12.;  // a2b_edge.F90:34-40
    const int k = klist ? klist[bz] : bz;
    const TileBox b = make_box<TI, TJ>(g, bx, by);
    const int i0 = b.i0, j0 = b.j0, il = b.ilast, jl = b.jlast;
    const int is = g.is, ie = g.ie, js = g.js, je = g.je;
    const size_t oA = (size_t)k * g.nA(), oU = (size_t)k * g.nU(), oV = (size_t)k * g.nV(), oB = (size_t)k * g.nB();
    const size_t oCX = (size_t)k * g.nCX(), oCY = (size_t)k * g.nCY(), oCC = (size_t)k * g.nCC();
    const double *crx = a.crx + oCX, *xfx = a.xfx + oCX, *cry = a.cry + oCY, *yfx = a.yfx + oCY;
    const double *uc = a.uc + oV, *vc = a.vc + oU;
    const double dt = a.dt;
    const int nord = a.lv.nord_k[k], nord_v = a.lv.nord_v[k];
    const double d2_bg = a.lv.d2_divg[k], damp_v = a.lv.damp_vt[k], d_con = a.lv.d_con_k[k];
    const bool need_heat = (d_con > 1.e-5) || g.do_diss_est;

    double *p = lds;
    const Tile su{p, i0 - 3, j0 - 3, TI + 6};  p += nSU;   // u on [i0-3,i0+TI+2] x [j0-3,j0+TJ+3]
    const Tile sv{p, i0 - 3, j0 - 3, TI + 7};  p += nSV;   // v on [i0-3,i0+TI+3] x [j0-3,j0+TJ+2]
    const Tile swk{p, i0 - 3, j0 - 3, TI + 6}; p += nQ;    // relative vorticity wk on E(3)
    const Tile svo{p, i0 - 3, j0 - 3, TI + 6}; p += nQ;    // absolute vorticity on E(3)
    double *scr = p;                           p += nScr;
    const Tile sfx{p, i0, j0, TI + 1};         p += nFXt;
    const Tile sfy{p, i0, j0, TI};             p += nFYt;
    const Tile ske{p, i0, j0, TI + 2};         p += nKE;   // ke at corners [i0,i0+TI+1] x [j0,j0+TJ+1]
    const Tile sdm{p, i0, j0, TI + 2};         p += nKE;   // damping term "vort" at the same corners
    const Tile sdv{p, i0 - 3, j0 - 3, TI + 8}; p += nDV;   // divg_d work copy, corners [i0-3,i0+TI+4] x ..
    const Tile svc2{p, i0 - 3, j0 - 3, TI + 8}; p += nDV;  // Laplacian work arrays "vc", "uc" (:1394,:1401)
    const Tile suc2{p, i0 - 3, j0 - 3, TI + 8}; p += nDV;

    load_tile<TI + 6, TJ + 7>(su, a.u + oU, g.nid, g.isd, g.ied, g.jsd, g.jed + 1, tid);
    load_tile<TI + 7, TJ + 6>(sv, a.v + oV, g.nid + 1, g.isd, g.ied + 1, g.jsd, g.jed, tid);
    FV3_SYNC();

    // ---- KE fluxes at corners [i0, il+2] x [j0, jl+2] clipped to [is,ie+1] x [js,je+1] (:1078-1198)
    {
      const double dt5 = 0.5 * dt;
      FV3_TILE_FOR((TI + 2), (nKE) / (TI + 2), li_, lj_) {
        const int i = i0 + li_, j = j0 + lj_;
        double kev = 0.;
        if (i <= il + 1 && j <= jl + 1) {
          const double vb = dt5 * (vc[g.iU(i - 1, j)] + vc[g.iU(i, j)]);                       // :1129
          const double ub = ppm_face_sw(&sv(i, j), sv.pitch, vb, g.rdy[g.iV(i, j - 1)], g.rdy[g.iV(i, j)],
                                        a.hord_mt);                                             // ytp_v :1134
          kev = vb * ub;                                                                        // :1139
          const double ub2 = dt5 * (uc[g.iV(i, j - 1)] + uc[g.iV(i, j)]);                      // :1186
          const double vb2 = ppm_face_sw(&su(i, j), 1, ub2, g.rdx[g.iU(i - 1, j)], g.rdx[g.iU(i, j)],
                                         a.hord_mt);                                            // xtp_u :1191
          kev = 0.5 * (kev + ub2 * vb2);                                                        // :1196
        }
        ske(i, j) = kev;
        sdm(i, j) = 0.;
      }
    }
    // ---- relative vorticity wk on E(3) (:1231-1247) and absolute vorticity (:1476-1495) ----------
    FV3_TILE_FOR((TI + 6), (nQ) / (TI + 6), li_, lj_) {
      const int i = i0 - 3 + li_, j = j0 - 3 + lj_;
      double wkv = 0., vo = 0.;
      if (i <= il + 3 && j <= jl + 3) {
        const double vt0 = su(i, j) * g.dx[g.iU(i, j)], vt1 = su(i, j + 1) * g.dx[g.iU(i, j + 1)];
        const double ut0 = sv(i, j) * g.dy[g.iV(i, j)], ut1 = sv(i + 1, j) * g.dy[g.iV(i + 1, j)];
        wkv = g.rarea[g.iA(i, j)] * (vt0 - vt1 - ut0 + ut1);
        vo = wkv + g.f0[g.iA(i, j)];
      }
      swk(i, j) = wkv;
      svo(i, j) = vo;
    }
    FV3_SYNC();

    // ---- divergence damping -> sdm ("vort" of :1368/:1455) and ke += sdm ---------------------------
    // corners needed: [i0, il+1] x [j0, jl+1]
    const int ic1 = il + 1, jc1 = jl + 1;
    if (nord == 0) {  // :1290-1371 with the global-index edge rules of the non-nested branch
      const int npx = g.npx, npy = g.npy;
      FV3_TILE_FOR((TI + 2), (nKE) / (TI + 2), li_, lj_) {
        const int i = i0 + li_, j = j0 + lj_;
        if (i > ic1 || j > jc1) continue;
        // ptc(i-1,j), ptc(i,j), vort(i,j-1), vort(i,j)
        double ptc2[2], vor2[2];
        for (int s = 0; s < 2; s++) {
          const int ii = i - 1 + s;
          if (j == 1 || j == npy) {
            ptc2[s] = (vc[g.iU(ii, j)] > 0) ? su(ii, j) * g.dyc[g.iU(ii, j)] * g.sinsg(ii, j - 1, 4)
                                             : su(ii, j) * g.dyc[g.iU(ii, j)] * g.sinsg(ii, j, 2);
          } else {
            ptc2[s] = (su(ii, j) - 0.5 * (a.va[oA + g.iA(ii, j - 1)] + a.va[oA + g.iA(ii, j)]) * g.cosa_v[g.iU(ii, j)]) *
                      g.dyc[g.iU(ii, j)] * g.sina_v[g.iU(ii, j)];
          }
          const int jj = j - 1 + s;
          if (i == 1 && is == 1) {
            vor2[s] = (uc[g.iV(1, jj)] > 0) ? sv(1, jj) * g.dxc[g.iV(1, jj)] * g.sinsg(0, jj, 3)
                                            : sv(1, jj) * g.dxc[g.iV(1, jj)] * g.sinsg(1, jj, 1);
          } else if (i == npx && (ie + 1) == npx) {
            vor2[s] = (uc[g.iV(npx, jj)] > 0) ? sv(npx, jj) * g.dxc[g.iV(npx, jj)] * g.sinsg(npx - 1, jj, 3)
                                              : sv(npx, jj) * g.dxc[g.iV(npx, jj)] * g.sinsg(npx, jj, 1);
          } else {
            vor2[s] = (sv(i, jj) - 0.5 * (a.ua[oA + g.iA(i - 1, jj)] + a.ua[oA + g.iA(i, jj)]) * g.cosa_u[g.iV(i, jj)]) *
                      g.dxc[g.iV(i, jj)] * g.sina_u[g.iV(i, jj)];
          }
        }
        double dpc = vor2[0] - vor2[1] + ptc2[0] - ptc2[1];                        // :1354
        dpc = g.rarea_c[g.iB(i, j)] * dpc;                                          // :1366
        const double damp = g.da_min_c * dmax(d2_bg, dmin(0.20, a.dddmp * fabs(dpc * dt)));
        const double vd = damp * dpc;
        sdm(i, j) = vd;
        ske(i, j) = ske(i, j) + vd;
        if (a.delpc && i < i0 + TI + (il == ie ? 1 : 0) && j < j0 + TJ + (jl == je ? 1 : 0) && i <= il + 1 && j <= jl + 1)
          a.delpc[oA + g.iA(i, j)] = dpc;
      }
    } else {  // :1372-1460
      // work copy of divg_d on corners [i0-nt0, ic1+nt0], nt0 = nord-1
      FV3_TILE_FOR((TI + 8), (nDV) / (TI + 8), li_, lj_) {
        const int i = i0 - 3 + li_, j = j0 - 3 + lj_;
        double val = 0.;
        if (i >= g.isd && i <= g.ied + 1 && j >= g.jsd && j <= g.jed + 1) val = a.divg_d[oB + g.iB(i, j)];
        sdv(i, j) = val;
      }
      FV3_SYNC();
      for (int n = 1; n <= nord; n++) {
        const int nt = nord - n;
        // vc(i,j), j in [j0-nt, jc1+nt], i in [i0-1-nt, ic1+nt]  (:1392-1396)
        // uc(i,j), j in [j0-1-nt, jc1+nt], i in [i0-nt, ic1+nt]  (:1399-1403)
        FV3_TILE_FOR((TI + 8), (nDV) / (TI + 8), li_, lj_) {
          const int i = i0 - 3 + li_, j = j0 - 3 + lj_;
          if (j >= j0 - nt && j <= jc1 + nt && i >= i0 - 1 - nt && i <= ic1 + nt)
            svc2(i, j) = (sdv(i + 1, j) - sdv(i, j)) * g.divg_u[g.iU(i, j)];
          if (j >= j0 - 1 - nt && j <= jc1 + nt && i >= i0 - nt && i <= ic1 + nt)
            suc2(i, j) = (sdv(i, j + 1) - sdv(i, j)) * g.divg_v[g.iV(i, j)];
        }
        FV3_SYNC();
        FV3_TILE_FOR((TI + 8), (nDV) / (TI + 8), li_, lj_) {  // :1406-1424
          const int i = i0 - 3 + li_, j = j0 - 3 + lj_;
          if (j >= j0 - nt && j <= jc1 + nt && i >= i0 - nt && i <= ic1 + nt) {
            double d = suc2(i, j - 1) - suc2(i, j) + svc2(i - 1, j) - svc2(i, j);
            if (!g.stretched_grid) d = d * g.rarea_c[g.iB(i, j)];
            sdv(i, j) = d;
          }
        }
        FV3_SYNC();
      }
      // Smagorinsky-type coefficient (:1428-1443): smag_corner (:1937-2024) when dddmp >= 1e-5
      const bool smag = !(a.dddmp < 1.E-5);
      Tile ssh{scr, i0 - 3, j0 - 3, TI + 6};  // shear strain "wk" of smag_corner on E(2) (scratch is free here)
      if (smag) {
        FV3_TILE_FOR((TI + 6), (nQ) / (TI + 6), li_, lj_) {
          const int i = i0 - 3 + li_, j = j0 - 3 + lj_;
          double val = 0.;
          if (i >= i0 - 2 && i <= il + 3 && j >= j0 - 2 && j <= jl + 3 && i <= g.ied && j <= g.jed) {
            const double vt0 = su(i, j) * g.dx[g.iU(i, j)], vt1 = su(i, j + 1) * g.dx[g.iU(i, j + 1)];
            const double ut0 = sv(i, j) * g.dy[g.iV(i, j)], ut1 = sv(i + 1, j) * g.dy[g.iV(i + 1, j)];
            val = g.rarea[g.iA(i, j)] * (vt0 - vt1 + ut0 - ut1);  // :2014
          }
          ssh(i, j) = val;
        }
        FV3_SYNC();
      }
      const int n2 = nord + 1;
      const double dd8 = g.stretched_grid ? g.da_min * ipow(a.d4_bg, n2) : ipow(g.da_min_c * a.d4_bg, n2);
      FV3_TILE_FOR((TI + 2), (nKE) / (TI + 2), li_, lj_) {
        const int i = i0 + li_, j = j0 + lj_;
        if (i > ic1 || j > jc1) continue;
        const double dpc = a.divg_d[oB + g.iB(i, j)];  // delpc = saved divergence (:1376-1381)
        double vs = 0.;
        if (smag) {
          // tension strain at the corner (:1983-1997)
          const double utm = su(i - 1, j) * g.dyc[g.iU(i - 1, j)], ut0 = su(i, j) * g.dyc[g.iU(i, j)];
          const double vtm = sv(i, j - 1) * g.dxc[g.iV(i, j - 1)], vt0 = sv(i, j) * g.dxc[g.iV(i, j)];
          const double ten = g.rarea_c[g.iB(i, j)] * (vtm - vt0 - utm + ut0);
          // a2b_ord4 of the shear strain, doubly periodic branch (a2b_edge.F90:297-313)
          double qx[4], qy[4];
          for (int s = 0; s < 4; s++) {
            const int jj = j - 2 + s;  // qx(i, j-2..j+1)
            qx[s] = b1 * (ssh(i - 1, jj) + ssh(i, jj)) + b2 * (ssh(i - 2, jj) + ssh(i + 1, jj));
            const int ii = i - 2 + s;  // qy(i-2..i+1, j)
            qy[s] = b1 * (ssh(ii, j - 1) + ssh(ii, j)) + b2 * (ssh(ii, j - 2) + ssh(ii, j + 1));
          }
          const double sh = 0.5 * (a1 * (qx[1] + qx[2] + qy[1] + qy[2]) + a2 * (qx[0] + qx[3] + qy[0] + qy[3]));
          vs = fabs(dt) * sqrt(sh * sh + ten * ten);  // :2020
        }
        const double damp2 = g.da_min_c * dmax(d2_bg, dmin(0.20, a.dddmp * vs));  // :1454
        const double vd = damp2 * dpc + dd8 * sdv(i, j);
        sdm(i, j) = vd;
        ske(i, j) = ske(i, j) + vd;
        if (a.delpc && i < i0 + TI + (il == ie ? 1 : 0) && j < j0 + TJ + (jl == je ? 1 : 0) && i <= il + 1 && j <= jl + 1)
          a.delpc[oA + g.iA(i, j)] = dpc;
      }
    }
    FV3_SYNC();

    // ---- vorticity transport (:1498-1499) ----------------------------------------------------------
    tp2d_tile<TI, TJ>(g, b, tid, svo, crx, cry, xfx, yfx, nullptr, nullptr, a.hord_vt, scr, sfx, sfy);
    // fx = face value * xfx, fy = face value * yfx (tp_core.F90:217-226)

    // ---- vorticity damping fluxes (:1513-1519): ut := fx2, vt := fy2 of del6_vt_flux(wk) ------------
    Tile fxd{nullptr, 0, 0, 0}, fyd{nullptr, 0, 0, 0};
    const bool vdamp = damp_v > 1.E-5;
    if (vdamp) {
      const double damp4 = ipow(damp_v * g.da_min_c, nord_v + 1);
      deln_tile<TI, TJ>(g, b, tid, swk, nord_v, damp4, true, scr, fxd, fyd);
    }

    // ---- wind update (:1500-1509, :1589-1600) and heating (:1523-1586) for owned cells ---------------
    // A thread evaluates, for cell (i,j): u_new at (i,j) and (i,j+1), v_new at (i,j) and (i+1,j)
    // (the far ones feed the cell-mean heating term; only the owned ones are stored).
    FV3_TILE_FOR(TI, (TI * TJ) / TI, li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      if (i > il || j > jl) continue;
      double un[2], vn[2], ubn[2], vbn[2], fyh[2], fxh[2];
      for (int s = 0; s < 2; s++) {
        const int jj = j + s;
        // u(i,jj) = vt + ke(i,jj) - ke(i+1,jj) + fy(i,jj),  vt = u*dx (:1233,:1502)
        const double fyv = sfy(i, jj) * yfx[g.iCY(i, jj)];
        double uu = su(i, jj) * g.dx[g.iU(i, jj)] + ske(i, jj) - ske(i + 1, jj) + fyv;
        const double vtd = vdamp ? fyd(i, jj) : 0.;
        if (need_heat) {
          // work array vt at this point: del6 flux, or 0 (do_diss_est), or still u*dx of :1233 (:1513-1519)
          const double vth = vdamp ? vtd : (g.do_diss_est ? 0. : su(i, jj) * g.dx[g.iU(i, jj)]);
          const double ub = ((sdm(i, jj) - sdm(i + 1, jj)) + vth) * g.rdx[g.iU(i, jj)];  // :1465,:1526
          fyh[s] = uu * g.rdx[g.iU(i, jj)];                                              // :1527
          ubn[s] = ub;
        }
        if (vdamp) uu = uu + vtd;  // :1592
        un[s] = uu;
        const int ii = i + s;
        const double fxv = sfx(ii, j) * xfx[g.iCX(ii, j)];
        double vv = sv(ii, j) * g.dy[g.iV(ii, j)] + ske(ii, j) - ske(ii, j + 1) - fxv;   // :1507
        const double utd = vdamp ? fxd(ii, j) : 0.;
        if (need_heat) {
          const double uth = vdamp ? utd : (g.do_diss_est ? 0. : sv(ii, j) * g.dy[g.iV(ii, j)]);
          const double vb = ((sdm(ii, j) - sdm(ii, j + 1)) - uth) * g.rdy[g.iV(ii, j)];  // :1470,:1533
          fxh[s] = vv * g.rdy[g.iV(ii, j)];                                              // :1534
          vbn[s] = vb;
        }
        if (vdamp) vv = vv - utd;  // :1597
        vn[s] = vv;
      }
      a.u_out[oU + g.iU(i, j)] = un[0];
      if (j == je) a.u_out[oU + g.iU(i, j + 1)] = un[1];
      a.v_out[oV + g.iV(i, j)] = vn[0];
      if (i == ie) a.v_out[oV + g.iV(i + 1, j)] = vn[1];
      if (need_heat) {
        const double gy0 = fyh[0] * ubn[0], gy1 = fyh[1] * ubn[1], gx0 = fxh[0] * vbn[0], gx1 = fxh[1] * vbn[1];
        const double u2 = fyh[0] + fyh[1], du2 = ubn[0] + ubn[1], v2 = fxh[0] + fxh[1], dv2 = vbn[0] + vbn[1];
        const double damp = 0.25 * d_con;
        const double t2 = (ubn[0] * ubn[0] + ubn[1] * ubn[1] + vbn[0] * vbn[0] + vbn[1] * vbn[1]) +
                          2. * (gy0 + gy1 + gx0 + gx1) - g.cosa_s[g.iA(i, j)] * (u2 * dv2 + v2 * du2 + du2 * dv2);
        const double rs2 = g.rsin2[g.iA(i, j)];
        const double dpn = a.delp_out[oA + g.iA(i, j)];  // delp after the transport kernel (:1557)
        double hs = a.heat_s[oCC + g.iCC(i, j)];
        if (g.prevent_diss_cooling) {
          const double tmp = rs2 * t2;
          if (d_con > 1.e-5) a.heat_s[oCC + g.iCC(i, j)] = dpn * (hs - damp * dmin(0., tmp));
          if (g.do_diss_est) a.diss_e[oCC + g.iCC(i, j)] = a.diss_e[oCC + g.iCC(i, j)] - tmp;
        } else {
          a.heat_s[oCC + g.iCC(i, j)] = dpn * (hs - damp * rs2 * t2);
          if (g.do_diss_est) a.diss_e[oCC + g.iCC(i, j)] = a.diss_e[oCC + g.iCC(i, j)] - rs2 * t2;
        }
      }
    }
  }
};

}  // namespace fv3
