// cubed_dsw.h -- d_sw (model/sw_core.F90:494-1606) on a cubed-sphere face (grid_type < 3, not bounded), as passes
// (cubed_common.h): the contravariant winds with the face-edge forms and the 2x2 solves next to the corners (:652-846), the
// Courant numbers / area fluxes (:863-902), the Lin-Rood transports through cubed_tp.h, the B-grid winds for the kinetic
// energy with their edge extrapolations (:1099-1181), xtp_u / ytp_v in their cubed-sphere forms (:2154-2998), the corner
// kinetic energy (:1203-1228), divergence damping with the corner terms and fill_corners (:1290-1460), the vorticity flux
// and the D-grid wind update (:1476-1509).  Not built here (the entry point refuses them): del-2n damping of delp / w / pt /
// vorticity (deln_flux, del6_vt_flux), dissipative heating, Smagorinsky damping (dddmp > 0).
#pragma once

#include "cubed_tp.h"
#include "dsw_kernels.h"

namespace fv3 {

struct DswCubedState {
  Grid g;
  CubedGeom cg;
  DswArgs a;
  double *ut, *vt;         // contravariant winds: V / U layouts
  double *fx, *fy;         // mass fluxes of this call (FX / FY)
  double *gxw, *gyw, *gx, *gy;  // fluxes of w and of pt / q_con
  double *ke;              // B layout
  double *wk;              // A layout: relative vorticity, then absolute vorticity
  double *dd, *svc, *suc;  // divergence damping work arrays (B, U, V layouts)
  // del-2n damping and dissipative heating (cubed_damp.h; null when no level asks for them)
  double *wfx2 = nullptr, *wfy2 = nullptr;   // del6_vt_flux of w (V / U layouts), levels with nord_w > 0
  double *dfx2 = nullptr, *dfy2 = nullptr;   // del6_vt_flux of the relative vorticity: "ut", "vt" of sw_core.F90:1513-1515
  double *vortv = nullptr;                   // the damping term added to ke (B layout), kept for the heating (:1462-1473)
  double *smag = nullptr;                    // a2b_ord4 of the relative vorticity (B values on the A layout), dddmp > 0
  double *gxq = nullptr, *gyq = nullptr;     // fluxes of q_con (use_cond, :992-1000)
  // hybrid: the passes that write the outputs of d_sw only write the points of the frame of width own_w along the face edges
  // (0: every point); the marching kernels own the rest (DswArgs::mask_w)
  int own_w;
  FV3_HD bool own(int i, int j) const { return own_w == 0 || i <= own_w || i >= g.npx - own_w || j <= own_w || j >= g.npy - own_w; }
};

// D1a: contravariant winds, first layer: the interior form where it applies and the edge rows / columns that follow from
// uc, vc alone (:671-690, :695-701, :711-717, :731-737, :748-754); box (isd:ied+1, jsd:jed+1)
struct DswCubedD1a {
  DswCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const int npx = g.npx, npy = g.npy;
    const CA uc = cview_V(g, s.a.uc), vc = cview_U(g, s.a.vc);
    const double dt = s.a.dt;
    if (j <= g.jed && i >= g.is - 1 && i <= g.ie + 2) {
      const VA ut = view_V(g, s.ut);
      if (i == 1 || i == npx) {
        const double u0 = uc(i, j, k);
        ut(i, j, k) = (u0 * dt > 0.) ? u0 / g.sinsg(i - 1, j, 3) : u0 / g.sinsg(i, j, 1);
      } else if (j != 0 && j != 1 && j != npy - 1 && j != npy) {
        ut(i, j, k) = (uc(i, j, k) - 0.25 * g.cosa_u[g.iV(i, j)] * (vc(i - 1, j, k) + vc(i, j, k) + vc(i - 1, j + 1, k) + vc(i, j + 1, k))) *
                      g.rsin_u[g.iV(i, j)];
      }
    }
    if (i <= g.ied && j >= g.js - 1 && j <= g.je + 2) {
      const VA vt = view_U(g, s.vt);
      if (j == 1 || j == npy) {
        const double v0 = vc(i, j, k);
        vt(i, j, k) = (v0 * dt > 0.) ? v0 / g.sinsg(i, j - 1, 4) : v0 / g.sinsg(i, j, 2);
      } else {
        vt(i, j, k) = (vc(i, j, k) - 0.25 * g.cosa_v[g.iU(i, j)] * (uc(i, j - 1, k) + uc(i + 1, j - 1, k) + uc(i, j, k) + uc(i + 1, j, k))) *
                      g.rsin_v[g.iU(i, j)];
      }
    }
  }
};

// D1a as a tile kernel whose threads MARCH: 64 lanes along i, four groups of kRows consecutive rows; the rows of uc / vc a thread has read
// serve two rows of results (ut(j) reads vc(j), vc(j+1); vt(j) reads uc(j-1), uc(j)), so a point costs 4 field loads and 4 metric loads
// instead of 10 + 4.  What D1a waits for is its loads' way through the L1 (profiles/r06_pmc_pass.csv: 90 M cache-line accesses a launch,
// 18 per 8-byte-a-lane load instruction, for 0.76 GB of HBM traffic).  The statements and their conditions are D1a's: the same bits.
struct DswCubedD1aRows {
  DswCubedState s;
  static constexpr int kRows = 8;
  FV3_HD void operator()(int bx, int by, int k, int tid, double *) const {
    const Grid &g = s.g;
    const int npx = g.npx, npy = g.npy;
    const int i0 = g.isd, i1 = g.ied + 1, j0 = g.jsd, j1 = g.jed + 1;
    const CA uc = cview_V(g, s.a.uc), vc = cview_U(g, s.a.vc);
    const VA ut = view_V(g, s.ut), vt = view_U(g, s.vt);
    const double dt = s.a.dt;
    for (int t = tid; t < 256; t += kNT) {
      const int i = i0 + bx * 64 + (t & 63);
      const int jA = j0 + (by * 4 + (t >> 6)) * kRows;
      if (i > i1 || jA > j1) continue;
      const int jB = jA + kRows - 1 < j1 ? jA + kRows - 1 : j1;
      // clamped addresses: a clamped value is one no kept result reads (the conditions below are D1a's)
      const int iu1 = i + 1 <= g.ied + 1 ? i + 1 : g.ied + 1;            // uc(i+1, .)
      const int ivm = i - 1 >= g.isd ? i - 1 : g.isd;                    // vc(i-1, .)
      const int iv0 = i <= g.ied ? i : g.ied;                            // vc(i, .)
      auto ju = [&](int j) { return j < g.jsd ? g.jsd : (j > g.jed ? g.jed : j); };
      auto jv = [&](int j) { return j < g.jsd ? g.jsd : (j > g.jed + 1 ? g.jed + 1 : j); };
      double ucm0 = uc(i, ju(jA - 1), k), ucm1 = uc(iu1, ju(jA - 1), k);        // row j-1
      double vc0m = vc(ivm, jv(jA), k), vc00 = vc(iv0, jv(jA), k);              // row j
      for (int j = jA; j <= jB; j++) {
        const double uc00 = uc(i, ju(j), k), uc01 = uc(iu1, ju(j), k);          // row j
        const double vc1m = vc(ivm, jv(j + 1), k), vc10 = vc(iv0, jv(j + 1), k);  // row j+1
        if (j <= g.jed && i >= g.is - 1 && i <= g.ie + 2) {
          if (i == 1 || i == npx) {
            const double u0 = uc00;
            ut(i, j, k) = (u0 * dt > 0.) ? u0 / g.sinsg(i - 1, j, 3) : u0 / g.sinsg(i, j, 1);
          } else if (j != 0 && j != 1 && j != npy - 1 && j != npy) {
            ut(i, j, k) = (uc00 - 0.25 * g.cosa_u[g.iV(i, j)] * (vc0m + vc00 + vc1m + vc10)) * g.rsin_u[g.iV(i, j)];
          }
        }
        if (i <= g.ied && j >= g.js - 1 && j <= g.je + 2) {
          if (j == 1 || j == npy) {
            const double v0 = vc00;
            vt(i, j, k) = (v0 * dt > 0.) ? v0 / g.sinsg(i, j - 1, 4) : v0 / g.sinsg(i, j, 2);
          } else {
            vt(i, j, k) = (vc00 - 0.25 * g.cosa_v[g.iU(i, j)] * (ucm0 + ucm1 + uc00 + uc01)) * g.rsin_v[g.iU(i, j)];
          }
        }
        ucm0 = uc00; ucm1 = uc01; vc0m = vc1m; vc00 = vc10;
      }
    }
  }
};

// D1b: second layer: the winds parallel to an edge in the two rows / columns next to it (:702-708, :719-724, :739-745,
// :756-762); box (0:npx, 0:npy)
struct DswCubedD1b {
  DswCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const int npx = g.npx, npy = g.npy;
    const CA uc = cview_V(g, s.a.uc), vc = cview_U(g, s.a.vc);
    const VA ut = view_V(g, s.ut), vt = view_U(g, s.vt);
    if ((i == 0 || i == 1 || i == npx - 1 || i == npx) && j >= 3 && j <= npy - 2)
      vt(i, j, k) = vc(i, j, k) - 0.25 * g.cosa_v[g.iU(i, j)] * (ut(i, j - 1, k) + ut(i + 1, j - 1, k) + ut(i, j, k) + ut(i + 1, j, k));
    if ((j == 0 || j == 1 || j == npy - 1 || j == npy) && i >= 3 && i <= npx - 2)
      ut(i, j, k) = uc(i, j, k) - 0.25 * g.cosa_u[g.iV(i, j)] * (vt(i - 1, j, k) + vt(i, j, k) + vt(i - 1, j + 1, k) + vt(i, j + 1, k));
  }
};

// D1c: the 2x2 solves next to the four corners (:779-846); one thread per (corner, k)
struct DswCubedD1c {
  DswCubedState s;
  FV3_HD void operator()(int m, int, int k) const {
    const Grid &g = s.g;
    const int npx = g.npx, npy = g.npy;
    const CA uc_ = cview_V(g, s.a.uc), vc_ = cview_U(g, s.a.vc);
    const VA ut_ = view_V(g, s.ut), vt_ = view_U(g, s.vt);
    auto UC = [&](int i, int j) { return uc_(i, j, k); };
    auto VC = [&](int i, int j) { return vc_(i, j, k); };
    auto UT = [&](int i, int j) -> double & { return ut_(i, j, k); };
    auto VT = [&](int i, int j) -> double & { return vt_(i, j, k); };
    auto CU = [&](int i, int j) { return g.cosa_u[g.iV(i, j)]; };
    auto CV = [&](int i, int j) { return g.cosa_v[g.iU(i, j)]; };
    double damp;
    if (m == 0) {  // sw
      damp = 1. / (1. - 0.0625 * CU(2, 0) * CV(1, 0));
      const double a0 = (UC(2, 0) - 0.25 * CU(2, 0) * (VT(1, 1) + VT(2, 1) + VT(2, 0) + VC(1, 0) - 0.25 * CV(1, 0) * (UT(1, 0) + UT(1, -1) + UT(2, -1)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(0, 1) * CV(0, 2));
      const double a1 = (VC(0, 2) - 0.25 * CV(0, 2) * (UT(1, 1) + UT(1, 2) + UT(0, 2) + UC(0, 1) - 0.25 * CU(0, 1) * (VT(0, 1) + VT(-1, 1) + VT(-1, 2)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(2, 1) * CV(1, 2));
      const double a2 = (UC(2, 1) - 0.25 * CU(2, 1) * (VT(1, 1) + VT(2, 1) + VT(2, 2) + VC(1, 2) - 0.25 * CV(1, 2) * (UT(1, 1) + UT(1, 2) + UT(2, 2)))) * damp;
      const double a3 = (VC(1, 2) - 0.25 * CV(1, 2) * (UT(1, 1) + UT(1, 2) + UT(2, 2) + UC(2, 1) - 0.25 * CU(2, 1) * (VT(1, 1) + VT(2, 1) + VT(2, 2)))) * damp;
      UT(2, 0) = a0; VT(0, 2) = a1; UT(2, 1) = a2; VT(1, 2) = a3;
    } else if (m == 1) {  // se
      damp = 1. / (1. - 0.0625 * CU(npx - 1, 0) * CV(npx - 1, 0));
      const double a0 = (UC(npx - 1, 0) - 0.25 * CU(npx - 1, 0) * (VT(npx - 1, 1) + VT(npx - 2, 1) + VT(npx - 2, 0) + VC(npx - 1, 0) -
                         0.25 * CV(npx - 1, 0) * (UT(npx, 0) + UT(npx, -1) + UT(npx - 1, -1)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(npx + 1, 1) * CV(npx, 2));
      const double a1 = (VC(npx, 2) - 0.25 * CV(npx, 2) * (UT(npx, 1) + UT(npx, 2) + UT(npx + 1, 2) + UC(npx + 1, 1) -
                         0.25 * CU(npx + 1, 1) * (VT(npx, 1) + VT(npx + 1, 1) + VT(npx + 1, 2)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(npx - 1, 1) * CV(npx - 1, 2));
      const double a2 = (UC(npx - 1, 1) - 0.25 * CU(npx - 1, 1) * (VT(npx - 1, 1) + VT(npx - 2, 1) + VT(npx - 2, 2) + VC(npx - 1, 2) -
                         0.25 * CV(npx - 1, 2) * (UT(npx, 1) + UT(npx, 2) + UT(npx - 1, 2)))) * damp;
      const double a3 = (VC(npx - 1, 2) - 0.25 * CV(npx - 1, 2) * (UT(npx, 1) + UT(npx, 2) + UT(npx - 1, 2) + UC(npx - 1, 1) -
                         0.25 * CU(npx - 1, 1) * (VT(npx - 1, 1) + VT(npx - 2, 1) + VT(npx - 2, 2)))) * damp;
      UT(npx - 1, 0) = a0; VT(npx, 2) = a1; UT(npx - 1, 1) = a2; VT(npx - 1, 2) = a3;
    } else if (m == 2) {  // ne
      damp = 1. / (1. - 0.0625 * CU(npx - 1, npy) * CV(npx - 1, npy + 1));
      const double a0 = (UC(npx - 1, npy) - 0.25 * CU(npx - 1, npy) * (VT(npx - 1, npy) + VT(npx - 2, npy) + VT(npx - 2, npy + 1) + VC(npx - 1, npy + 1) -
                         0.25 * CV(npx - 1, npy + 1) * (UT(npx, npy) + UT(npx, npy + 1) + UT(npx - 1, npy + 1)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(npx + 1, npy - 1) * CV(npx, npy - 1));
      const double a1 = (VC(npx, npy - 1) - 0.25 * CV(npx, npy - 1) * (UT(npx, npy - 1) + UT(npx, npy - 2) + UT(npx + 1, npy - 2) + UC(npx + 1, npy - 1) -
                         0.25 * CU(npx + 1, npy - 1) * (VT(npx, npy) + VT(npx + 1, npy) + VT(npx + 1, npy - 1)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(npx - 1, npy - 1) * CV(npx - 1, npy - 1));
      const double a2 = (UC(npx - 1, npy - 1) - 0.25 * CU(npx - 1, npy - 1) * (VT(npx - 1, npy) + VT(npx - 2, npy) + VT(npx - 2, npy - 1) + VC(npx - 1, npy - 1) -
                         0.25 * CV(npx - 1, npy - 1) * (UT(npx, npy - 1) + UT(npx, npy - 2) + UT(npx - 1, npy - 2)))) * damp;
      const double a3 = (VC(npx - 1, npy - 1) - 0.25 * CV(npx - 1, npy - 1) * (UT(npx, npy - 1) + UT(npx, npy - 2) + UT(npx - 1, npy - 2) + UC(npx - 1, npy - 1) -
                         0.25 * CU(npx - 1, npy - 1) * (VT(npx - 1, npy) + VT(npx - 2, npy) + VT(npx - 2, npy - 1)))) * damp;
      UT(npx - 1, npy) = a0; VT(npx, npy - 1) = a1; UT(npx - 1, npy - 1) = a2; VT(npx - 1, npy - 1) = a3;
    } else {  // nw
      damp = 1. / (1. - 0.0625 * CU(2, npy) * CV(1, npy + 1));
      const double a0 = (UC(2, npy) - 0.25 * CU(2, npy) * (VT(1, npy) + VT(2, npy) + VT(2, npy + 1) + VC(1, npy + 1) -
                         0.25 * CV(1, npy + 1) * (UT(1, npy) + UT(1, npy + 1) + UT(2, npy + 1)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(0, npy - 1) * CV(0, npy - 1));
      const double a1 = (VC(0, npy - 1) - 0.25 * CV(0, npy - 1) * (UT(1, npy - 1) + UT(1, npy - 2) + UT(0, npy - 2) + UC(0, npy - 1) -
                         0.25 * CU(0, npy - 1) * (VT(0, npy) + VT(-1, npy) + VT(-1, npy - 1)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(2, npy - 1) * CV(1, npy - 1));
      const double a2 = (UC(2, npy - 1) - 0.25 * CU(2, npy - 1) * (VT(1, npy) + VT(2, npy) + VT(2, npy - 1) + VC(1, npy - 1) -
                         0.25 * CV(1, npy - 1) * (UT(1, npy - 1) + UT(1, npy - 2) + UT(2, npy - 2)))) * damp;
      const double a3 = (VC(1, npy - 1) - 0.25 * CV(1, npy - 1) * (UT(1, npy - 1) + UT(1, npy - 2) + UT(2, npy - 2) + UC(2, npy - 1) -
                         0.25 * CU(2, npy - 1) * (VT(1, npy) + VT(2, npy) + VT(2, npy - 1)))) * damp;
      UT(2, npy) = a0; VT(0, npy - 1) = a1; UT(2, npy - 1) = a2; VT(1, npy - 1) = a3;
    }
  }
};

// D2: Courant numbers and area fluxes (:863-902), the accumulation of cx, cy (:923-936); box (isd:ied, jsd:jed)
struct DswCubedD2 {
  DswCubedState s;
  int acc = 1;   // 0: no accumulation of cx, cy (the frame's own copies of the Courant numbers, fv3_api.hip dsw_cubed)
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const double dt = s.a.dt;
    if (i >= g.is && i <= g.ie + 1) {
      double x = dt * cview_V(g, s.ut)(i, j, k), cr;
      if (x > 0.) {
        cr = x * g.rdxa[g.iA(i - 1, j)];
        x = g.dy[g.iV(i, j)] * x * g.sinsg(i - 1, j, 3);
      } else {
        cr = x * g.rdxa[g.iA(i, j)];
        x = g.dy[g.iV(i, j)] * x * g.sinsg(i, j, 1);
      }
      view_CX(g, s.a.crx)(i, j, k) = cr;
      view_CX(g, s.a.xfx)(i, j, k) = x;
      if (acc) {
        double &cxv = view_CX(g, s.a.cx)(i, j, k);
        cxv = cxv + cr;
      }
    }
    if (j >= g.js && j <= g.je + 1) {
      double y = dt * cview_U(g, s.vt)(i, j, k), cr;
      if (y > 0.) {
        cr = y * g.rdya[g.iA(i, j - 1)];
        y = g.dx[g.iU(i, j)] * y * g.sinsg(i, j - 1, 4);
      } else {
        cr = y * g.rdya[g.iA(i, j)];
        y = g.dx[g.iU(i, j)] * y * g.sinsg(i, j, 2);
      }
      view_CY(g, s.a.cry)(i, j, k) = cr;
      view_CY(g, s.a.yfx)(i, j, k) = y;
      if (acc) {
        double &cyv = view_CY(g, s.a.cy)(i, j, k);
        cyv = cyv + cr;
      }
    }
  }
};

// D4: the flux capacitors of the mass fluxes (:928-940) and the flux-form update of delp, pt, w, q_con (:983-1066, :1249-1283);
// box (is:ie+1, js:je+1)
struct DswCubedD4 {
  DswCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const CA fx = cview_FX(g, s.fx), fy = cview_FY(g, s.fy);
    if (!s.own(i, j)) return;
    if (j <= g.je) {
      double &m = view_FX(g, s.a.mfx)(i, j, k);
      m = m + fx(i, j, k);
    }
    if (i <= g.ie) {
      double &m = view_FY(g, s.a.mfy)(i, j, k);
      m = m + fy(i, j, k);
    }
    if (i <= g.ie && j <= g.je) {
      const double ra = g.rarea[g.iA(i, j)];
      const double dp = cview_A(g, s.a.delp)(i, j, k);
      const CA gx = cview_FX(g, s.gx), gy = cview_FY(g, s.gy);
      double ptv = cview_A(g, s.a.pt)(i, j, k) * dp + (gx(i, j, k) - gx(i + 1, j, k) + gy(i, j, k) - gy(i, j + 1, k)) * ra;
      const double dpn = dp + (fx(i, j, k) - fx(i + 1, j, k) + fy(i, j, k) - fy(i, j + 1, k)) * ra;
      ptv = ptv / dpn;
      view_A(g, s.a.delp_out)(i, j, k) = dpn;
      view_A(g, s.a.pt_out)(i, j, k) = ptv;
      double heat = 0., diss = 0.;
      if (!s.a.hydrostatic) {
        const CA gxw = cview_FX(g, s.gxw), gyw = cview_FY(g, s.gyw), w = cview_A(g, s.a.w);
        const double w0 = w(i, j, k);
        const double wv = dp * w0 + (gxw(i, j, k) - gxw(i + 1, j, k) + gyw(i, j, k) - gyw(i, j + 1, k)) * ra;
        double wn = wv / dpn;
        const double damp_w = s.a.lv.damp_w[k];
        if (damp_w > 1.E-5) {  // del-2 damping of w on the sponge levels (nord_w = 0): :950-982, del6_vt_flux :1640-1672
          const double damp4 = ipow(damp_w * g.da_min_c, s.a.lv.nord_w[k] + 1), dd8 = s.a.kgb * fabs(s.a.dt);
          double dw;
          if (s.a.lv.nord_w[k] > 0) {  // the fluxes of the del-2n passes
            const CA fx2 = cview_V(g, s.wfx2), fy2 = cview_U(g, s.wfy2);
            dw = (fx2(i, j, k) - fx2(i + 1, j, k) + fy2(i, j, k) - fy2(i, j + 1, k)) * ra;
          } else {
            const double d0 = damp4 * w0;
            const double fx0 = g.del6_v[g.iV(i, j)] * (damp4 * w(i - 1, j, k) - d0), fx1 = g.del6_v[g.iV(i + 1, j)] * (d0 - damp4 * w(i + 1, j, k));
            const double fy0 = g.del6_u[g.iU(i, j)] * (damp4 * w(i, j - 1, k) - d0), fy1 = g.del6_u[g.iU(i, j + 1)] * (d0 - damp4 * w(i, j + 1, k));
            dw = (fx0 - fx1 + fy0 - fy1) * ra;
          }
          const double tmp = dw * (w0 + 0.5 * dw);
          heat = g.prevent_diss_cooling ? dd8 - dmin(0., tmp) : dd8 - tmp;
          if (g.do_diss_est) diss = g.prevent_diss_cooling ? dd8 - tmp : heat;   // :964-966, :976-978
          wn = wn + dw;
        }
        view_A(g, s.a.w_out)(i, j, k) = wn;
      }
      if (s.a.use_cond) {  // :992-1000, :1277-1283
        const CA gq = cview_FX(g, s.gxq), hq = cview_FY(g, s.gyq);
        const double qv = dp * cview_A(g, s.a.q_con)(i, j, k) + (gq(i, j, k) - gq(i + 1, j, k) + hq(i, j, k) - hq(i, j + 1, k)) * ra;
        view_A(g, s.a.q_con_out)(i, j, k) = qv / dpn;
      }
      view_CC(g, s.a.heat_s)(i, j, k) = heat;
      view_CC(g, s.a.diss_e)(i, j, k) = diss;
    }
  }
};

// ---- xtp_u / ytp_v on the cubed sphere, one corner value per call (sw_core.F90:2154-2998) ------------------------------------
// W(m), DX(m), RDX(m): the wind, its metric and the reciprocal along the line; i: the corner index along the line; c: the
// advective displacement there; edge_row: the line runs along a face edge.
template <class W, class DX, class RDX>
FV3_HD double tp_wind_face_cs(const W &w, const DX &dx, const RDX &rdx, int i, double c, int iord, int npx, bool edge_row) {
  constexpr double r3 = 1. / 3., near_zero = 1.E-9, p1 = 7. / 12., p2 = -1. / 12.;
  constexpr double c1 = -2. / 14., c2 = 11. / 14., c3 = 5. / 14., s11 = 11. / 14., s14 = 4. / 7., s15 = 3. / 14.;
  const int ic = (c > 0.) ? i - 1 : i;
  auto dm = [&](int m) { return ppm_dm(w(m - 1), w(m), w(m + 1)); };
  auto dq = [&](int m) { return w(m + 1) - w(m); };
  if (iord >= 8) {
    auto al = [&](int m) { return 0.5 * (w(m - 1) + w(m)) + r3 * (dm(m - 1) - dm(m)); };
    auto x_edge = [&](int e) {  // :2448-2452
      const double x0L = 0.5 * ((2. * dx(e - 1) + dx(e - 2)) * (w(e - 1)) - dx(e - 1) * (w(e - 2))) / (dx(e - 1) + dx(e - 2));
      const double x0R = 0.5 * ((2. * dx(e) + dx(e + 1)) * (w(e)) - dx(e) * (w(e + 1))) / (dx(e) + dx(e + 1));
      return x0L + x0R;
    };
    double bl, br;
    const double w0 = w(ic);
    if (ic >= 0 && ic <= 2) {
      if (ic == 2) {
        br = al(3) - w(2);
        bl = (s15 * w(1) + s11 * w(2) - s14 * dm(2)) - w(2);
        pert_ppm_std(bl, br);
      } else if (edge_row) {
        bl = 0.;
        br = 0.;
      } else if (ic == 1) {
        br = (s15 * w(1) + s11 * w(2) - s14 * dm(2)) - w(1);
        bl = x_edge(1) - w(1);
      } else {
        bl = s14 * dm(-1) - s11 * dq(-1);
        br = x_edge(1) - w(0);
      }
    } else if (ic >= npx - 2 && ic <= npx) {
      if (ic == npx - 2) {
        bl = al(npx - 2) - w(npx - 2);
        br = (s15 * w(npx - 1) + s11 * w(npx - 2) + s14 * dm(npx - 2)) - w(npx - 2);
        pert_ppm_std(bl, br);
      } else if (edge_row) {
        bl = 0.;
        br = 0.;
      } else if (ic == npx - 1) {
        bl = (s15 * w(npx - 1) + s11 * w(npx - 2) + s14 * dm(npx - 2)) - w(npx - 1);
        br = x_edge(npx) - w(npx - 1);
      } else {
        bl = x_edge(npx) - w(npx);
        br = s11 * dq(npx) - s14 * dm(npx + 1);
      }
    } else if (iord == 8) {
      const double xt = 2. * dm(ic);
      bl = -fsign(dmin(fabs(xt), fabs(al(ic) - w0)), xt);
      br = fsign(dmin(fabs(xt), fabs(al(ic + 1) - w0)), xt);
    } else if (iord == 9) {
      const double pmp_1 = -2. * dq(ic), lac_1 = pmp_1 + 1.5 * dq(ic + 1);
      bl = dmin(dmax3(0., pmp_1, lac_1), dmax(al(ic) - w0, dmin3(0., pmp_1, lac_1)));
      const double pmp_2 = 2. * dq(ic - 1), lac_2 = pmp_2 - 1.5 * dq(ic - 2);
      br = dmin(dmax3(0., pmp_2, lac_2), dmax(al(ic + 1) - w0, dmin3(0., pmp_2, lac_2)));
    } else if (iord == 10) {
      bl = al(ic) - w0;
      br = al(ic + 1) - w0;
      if (fabs(dm(ic)) < near_zero) {
        if (fabs(dm(ic - 1)) + fabs(dm(ic + 1)) < near_zero) {  // 2-delta-x structure detected within 3 cells
          bl = 0.;
          br = 0.;
        }
      } else if (fabs(3. * (bl + br)) > fabs(bl - br)) {
        const double pmp_1 = -2. * dq(ic), lac_1 = pmp_1 + 1.5 * dq(ic + 1);
        bl = dmin(dmax3(0., pmp_1, lac_1), dmax(bl, dmin3(0., pmp_1, lac_1)));
        const double pmp_2 = 2. * dq(ic - 1), lac_2 = pmp_2 - 1.5 * dq(ic - 2);
        br = dmin(dmax3(0., pmp_2, lac_2), dmax(br, dmin3(0., pmp_2, lac_2)));
      }
    } else {
      bl = al(ic) - w0;
      br = al(ic + 1) - w0;
    }
    if (c > 0.) {
      const double cfl = c * rdx(i - 1);
      return w0 + (1. - cfl) * (br - cfl * (bl + br));
    }
    const double cfl = c * rdx(i);
    return w0 + (1. + cfl) * (bl + cfl * (bl + br));
  }
  // iord = 5, 6, 7 (:2187-2377)
  auto al = [&](int m) { return p1 * (w(m - 1) + w(m)) + p2 * (w(m - 2) + w(m + 1)); };
  auto x_edge = [&](int e) {
    return 0.5 * (((2. * dx(e - 1) + dx(e - 2)) * (w(e - 1)) - dx(e - 1) * w(e - 2)) / (dx(e - 1) + dx(e - 2)) +
                  ((2. * dx(e) + dx(e + 1)) * (w(e)) - dx(e) * w(e + 1)) / (dx(e) + dx(e + 1)));
  };
  auto cell = [&](int m, double &bl, double &br) {
    if (m >= 0 && m <= 2) {
      const double xt = c3 * w(1) + c2 * w(2) + c1 * w(3);
      if (m == 2) {
        bl = xt - w(2);
        br = al(3) - w(2);
      } else if (edge_row) {
        bl = 0.;
        br = 0.;
      } else if (m == 1) {
        br = xt - w(1);
        bl = x_edge(1) - w(1);
      } else {
        bl = c1 * w(-2) + c2 * w(-1) + c3 * w(0) - w(0);
        br = x_edge(1) - w(0);
      }
    } else if (m >= npx - 2 && m <= npx) {
      const double xt = c1 * w(npx - 3) + c2 * w(npx - 2) + c3 * w(npx - 1);
      if (m == npx - 2) {
        bl = al(npx - 2) - w(npx - 2);
        br = xt - w(npx - 2);
      } else if (edge_row) {
        bl = 0.;
        br = 0.;
      } else if (m == npx - 1) {
        bl = xt - w(npx - 1);
        br = x_edge(npx) - w(npx - 1);
      } else {
        bl = x_edge(npx) - w(npx);
        br = c3 * w(npx) + c2 * w(npx + 1) + c1 * w(npx + 2) - w(npx);
      }
    } else {
      bl = al(m) - w(m);
      br = al(m + 1) - w(m);
    }
  };
  double blm, brm, bl0, br0;
  cell(i - 1, blm, brm);
  cell(i, bl0, br0);
  const double b0m = blm + brm, b00 = bl0 + br0;
  auto edge_cell = [&](int m) { return m == 0 || m == 1 || m == npx - 1 || m == npx; };
  bool sm, s0;
  if (iord == 5) {
    sm = blm * brm < 0.;
    s0 = bl0 * br0 < 0.;
  } else {
    sm = edge_cell(i - 1) ? (blm * brm < 0.) : (3. * fabs(b0m) < fabs(blm - brm));
    s0 = edge_cell(i) ? (bl0 * br0 < 0.) : (3. * fabs(b00) < fabs(bl0 - br0));
  }
  double fx0, flux;
  if (c > 0.) {
    const double cfl = c * rdx(i - 1);
    fx0 = (1. - cfl) * (brm - cfl * b0m);
    flux = w(i - 1);
  } else {
    const double cfl = c * rdx(i);
    fx0 = (1. + cfl) * (bl0 + cfl * b00);
    flux = w(i);
  }
  if (sm || s0) flux = flux + fx0;
  return flux;
}

// D5: kinetic energy at the cell corners: B-grid winds with the edge extrapolations (:1099-1181), the advected winds by
// ytp_v / xtp_u, the corner fix (:1203-1228); box (is:ie+1, js:je+1)
struct DswCubedD5 {
  DswCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const int npx = g.npx, npy = g.npy;
    const CA uc = cview_V(g, s.a.uc), vc = cview_U(g, s.a.vc), ut = cview_V(g, s.ut), vt = cview_U(g, s.vt);
    const CA u = cview_U(g, s.a.u), v = cview_V(g, s.a.v);
    const double dt = s.a.dt, dt5 = 0.5 * dt, dt4 = 0.25 * dt;
    const double cosa = g.cosa[g.iB(i, j)], rsina = s.cg.rsina[(size_t)(j - g.js) * (g.nx + 1) + (i - g.is)];
    double vb, ub;
    if (j == 1 || j == npy)
      vb = dt5 * (vt(i - 1, j, k) + vt(i, j, k));  // corner values are incorrect
    else if (i == 1 || i == npx)
      vb = dt4 * (-vt(i - 2, j, k) + 3. * (vt(i - 1, j, k) + vt(i, j, k)) - vt(i + 1, j, k));
    else
      vb = dt5 * (vc(i - 1, j, k) + vc(i, j, k) - (uc(i, j - 1, k) + uc(i, j, k)) * cosa) * rsina;
    if (i == 1 || i == npx)
      ub = dt5 * (ut(i, j - 1, k) + ut(i, j, k));
    else if (j == 1 || j == npy)
      ub = dt4 * (-ut(i, j - 2, k) + 3. * (ut(i, j - 1, k) + ut(i, j, k)) - ut(i, j + 1, k));
    else
      ub = dt5 * (uc(i, j - 1, k) + uc(i, j, k) - (vc(i - 1, j, k) + vc(i, j, k)) * cosa) * rsina;
    // ytp_v(vb): v along j at column i; xtp_u(ub): u along i at row j
    auto wv = [&](int m) { return v(i, m, k); };
    auto dyv = [&](int m) { return g.dy[g.iV(i, m)]; };
    auto rdyv = [&](int m) { return g.rdy[g.iV(i, m)]; };
    const double ubn = tp_wind_face_cs(wv, dyv, rdyv, j, vb, s.a.hord_mt, npy, i == 1 || i == npx);
    auto wu = [&](int m) { return u(m, j, k); };
    auto dxu = [&](int m) { return g.dx[g.iU(m, j)]; };
    auto rdxu = [&](int m) { return g.rdx[g.iU(m, j)]; };
    const double vbn = tp_wind_face_cs(wu, dxu, rdxu, i, ub, s.a.hord_mt, npx, j == 1 || j == npy);
    double ke = vb * ubn;
    ke = 0.5 * (ke + ub * vbn);
    const double dt6 = dt / 6.;
    auto UT = [&](int a, int b) { return ut(a, b, k); };
    auto VT = [&](int a, int b) { return vt(a, b, k); };
    if (i == 1 && j == 1)
      ke = dt6 * ((UT(1, 1) + UT(1, 0)) * u(1, 1, k) + (VT(1, 1) + VT(0, 1)) * v(1, 1, k) + (UT(1, 1) + VT(1, 1)) * u(0, 1, k));
    if (i == npx && j == 1)
      ke = dt6 * ((UT(i, 1) + UT(i, 0)) * u(i - 1, 1, k) + (VT(i, 1) + VT(i - 1, 1)) * v(i, 1, k) + (UT(i, 1) - VT(i - 1, 1)) * u(i, 1, k));
    if (i == npx && j == npy)
      ke = dt6 * ((UT(i, j) + UT(i, j - 1)) * u(i - 1, j, k) + (VT(i, j) + VT(i - 1, j)) * v(i, j - 1, k) + (UT(i, j - 1) + VT(i - 1, j)) * u(i, j, k));
    if (i == 1 && j == npy)
      ke = dt6 * ((UT(1, j) + UT(1, j - 1)) * u(1, j, k) + (VT(1, j) + VT(0, j)) * v(1, j - 1, k) + (UT(1, j - 1) - VT(1, j)) * u(0, j, k));
    view_B(g, s.ke)(i, j, k) = ke;
  }
};

// D6: relative vorticity (:1231-1247) on (isd:ied, jsd:jed) and the copy of divg_d into the damping work array
struct DswCubedD6 {
  DswCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const CA u = cview_U(g, s.a.u), v = cview_V(g, s.a.v);
    if (i <= g.ied && j <= g.jed) {
      const double vt0 = u(i, j, k) * g.dx[g.iU(i, j)], vt1 = u(i, j + 1, k) * g.dx[g.iU(i, j + 1)];
      const double ut0 = v(i, j, k) * g.dy[g.iV(i, j)], ut1 = v(i + 1, j, k) * g.dy[g.iV(i + 1, j)];
      view_A(g, s.wk)(i, j, k) = g.rarea[g.iA(i, j)] * (vt0 - vt1 - ut0 + ut1);
    }
    view_B(g, s.dd)(i, j, k) = cview_B(g, s.a.divg_d)(i, j, k);
  }
};

// divergence damping, nord_k > 0 (:1372-1460): one iteration n of the del-2n loop as three passes; fill_corners as tiny passes
struct DswCubedFillB {  // fill_corners(divg_d, FILL = dir, BGRID); one thread per (i, j) in 1..3 x 1..3
  DswCubedState s;
  int dir, n;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    if (!(s.a.lv.nord_k[k] > 0 && n <= s.a.lv.nord_k[k] && s.a.lv.nord_k[k] - n != 0)) return;
    const int npx = g.npx, npy = g.npy;
    const VA q = view_B(g, s.dd);
    if (dir == 1) {
      q(1 - i, 1 - j, k) = q(1 - j, i + 1, k);
      q(1 - i, npy + j, k) = q(1 - j, npy - i, k);
      q(npx + i, 1 - j, k) = q(npx + j, i + 1, k);
      q(npx + i, npy + j, k) = q(npx + j, npy - i, k);
    } else {
      q(1 - j, 1 - i, k) = q(i + 1, 1 - j, k);
      q(1 - j, npy + i, k) = q(i + 1, npy + j, k);
      q(npx + j, 1 - i, k) = q(npx - i, 1 - j, k);
      q(npx + j, npy + i, k) = q(npx - i, npy + j, k);
    }
  }
};
struct DswCubedFillD {  // fill_corners(vc, uc, VECTOR, DGRID): x = svc (U layout), y = suc (V layout); two steps
  DswCubedState s;
  int step, n;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    if (!(s.a.lv.nord_k[k] > 0 && n <= s.a.lv.nord_k[k] && s.a.lv.nord_k[k] - n != 0)) return;
    const int npx = g.npx, npy = g.npy;
    const VA x = view_U(g, s.svc), y = view_V(g, s.suc);
    const double sg = -1.;
    if (step == 0) {
      x(1 - i, 1 - j, k) = sg * y(1 - j, i, k);
      x(1 - i, npy + j, k) = y(1 - j, npy - i, k);
      x(npx - 1 + i, 1 - j, k) = y(npx + j, i, k);
      x(npx - 1 + i, npy + j, k) = sg * y(npx + j, npy - i, k);
    } else {
      y(1 - i, 1 - j, k) = sg * x(j, 1 - i, k);
      y(1 - i, npy - 1 + j, k) = x(j, npy + i, k);
      y(npx + i, 1 - j, k) = x(npx - j, 1 - i, k);
      y(npx + i, npy - 1 + j, k) = sg * x(npx - j, npy + i, k);
    }
  }
};
struct DswCubedDampVC {  // vc (then uc) of iteration n; box (is-3:ie+3, js-3:je+4)
  DswCubedState s;
  int n, which;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const int nord = s.a.lv.nord_k[k];
    if (nord <= 0 || n > nord) return;
    const int nt = nord - n;
    const CA dd = cview_B(g, s.dd);
    if (which == 0) {
      if (j >= g.js - nt && j <= g.je + 1 + nt && i >= g.is - 1 - nt && i <= g.ie + 1 + nt)
        view_U(g, s.svc)(i, j, k) = (dd(i + 1, j, k) - dd(i, j, k)) * g.divg_u[g.iU(i, j)];
    } else {
      if (j >= g.js - 1 - nt && j <= g.je + 1 + nt && i >= g.is - nt && i <= g.ie + 1 + nt)
        view_V(g, s.suc)(i, j, k) = (dd(i, j + 1, k) - dd(i, j, k)) * g.divg_v[g.iV(i, j)];
    }
  }
};
struct DswCubedDampDiv {  // the divergence of (uc, vc), corner terms, 1/area_c; box (is-2:ie+3, js-2:je+3)
  DswCubedState s;
  int n;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const int nord = s.a.lv.nord_k[k];
    if (nord <= 0 || n > nord) return;
    const int nt = nord - n, npx = g.npx, npy = g.npy;
    if (!(j >= g.js - nt && j <= g.je + 1 + nt && i >= g.is - nt && i <= g.ie + 1 + nt)) return;
    const CA uc = cview_V(g, s.suc), vc = cview_U(g, s.svc);
    double d = uc(i, j - 1, k) - uc(i, j, k) + vc(i - 1, j, k) - vc(i, j, k);
    if (i == 1 && j == 1) d = d - uc(1, 0, k);
    if (i == npx && j == 1) d = d - uc(npx, 0, k);
    if (i == npx && j == npy) d = d + uc(npx, npy, k);
    if (i == 1 && j == npy) d = d + uc(1, npy, k);
    if (!g.stretched_grid) d = d * g.rarea_c[g.iB(i, j)];
    view_B(g, s.dd)(i, j, k) = d;
  }
};

// The del-2n loop of the divergence (:1372-1426) in ONE launch, away from the face corners (see cubed_damp.h DelnFused for the
// scheme): a workgroup owns 32 x 16 B-grid points at one level, stages divg_d on them grown by nord_k, and alternates the gradients
// (DampVC) and the divergence (DampDiv) in LDS on squares that shrink by one per iteration.  fill_corners and the corner terms of the
// divergence only reach points within nord_k of BOTH edges of a face corner: those squares stay with the passes (CornerPass).
struct DswDampFused {
  static constexpr int TI = 32, TJ = 16, kMaxN = 3;
  static constexpr int PW = TI + 2 * kMaxN + 2, PH = TJ + 2 * kMaxN + 2;
  static constexpr int lds_doubles = 3 * PW * PH;
  DswCubedState s;
  int wo;
  const int *klist;
  FV3_D void operator()(int bx, int by, int bz, int tid, double *lds) const {
    const int k = klist ? klist[bz] : bz;
    const Grid &g = s.g;
    const int N = s.a.lv.nord_k[k], npx = g.npx, npy = g.npy;
    if (N <= 0) return;
    const int ia = g.is + bx * TI, ja = g.js + by * TJ;
    const int ib = ia + TI - 1 < g.ie + 1 ? ia + TI - 1 : g.ie + 1, jb = ja + TJ - 1 < g.je + 1 ? ja + TJ - 1 : g.je + 1;
    const int i0 = ia - kMaxN - 1, j0 = ja - kMaxN - 1;
    double *dd = lds, *vc = lds + PW * PH, *uc = lds + 2 * PW * PH;
#define TD(a, i, j) (a)[((j) - j0) * PW + ((i) - i0)]
    {
      const CA d0 = cview_B(g, s.a.divg_d);
      const int ca = ia - N, cb = ib + N, ra = ja - N, rb = jb + N, nc = cb - ca + 1, nr = rb - ra + 1;
      for (int idx = tid; idx < nc * nr; idx += kNT) {
        const int i = ca + idx % nc, j = ra + idx / nc;
        TD(dd, i, j) = d0(i, j, k);
      }
    }
    FV3_SYNC();
    for (int n = 1; n <= N; n++) {
      const int nt = N - n;
      const int ca = ia - nt, cb = ib + nt, ra = ja - nt, rb = jb + nt;   // the divergence of this iteration
      {
        // vc(i, j), i in [ca - 1, cb], j in [ra, rb]; uc(i, j), i in [ca, cb], j in [ra - 1, rb]
        const int nc = cb - ca + 2, nr = rb - ra + 1;
        for (int idx = tid; idx < nc * nr; idx += kNT) {
          const int i = ca - 1 + idx % nc, j = ra + idx / nc;
          TD(vc, i, j) = (TD(dd, i + 1, j) - TD(dd, i, j)) * g.divg_u[g.iU(i, j)];
        }
        const int nc2 = cb - ca + 1, nr2 = rb - ra + 2;
        for (int idx = tid; idx < nc2 * nr2; idx += kNT) {
          const int i = ca + idx % nc2, j = ra - 1 + idx / nc2;
          TD(uc, i, j) = (TD(dd, i, j + 1) - TD(dd, i, j)) * g.divg_v[g.iV(i, j)];
        }
      }
      FV3_SYNC();
      {
        const int nc = cb - ca + 1, nr = rb - ra + 1;
        for (int idx = tid; idx < nc * nr; idx += kNT) {
          const int i = ca + idx % nc, j = ra + idx / nc;
          double d = TD(uc, i, j - 1) - TD(uc, i, j) + TD(vc, i - 1, j) - TD(vc, i, j);
          if (!g.stretched_grid) d = d * g.rarea_c[g.iB(i, j)];
          TD(dd, i, j) = d;
        }
      }
      FV3_SYNC();
    }
    {
      const VA out = view_B(g, s.dd);
      const int nc = ib - ia + 1, nr = jb - ja + 1;
      for (int idx = tid; idx < nc * nr; idx += kNT) {
        const int i = ia + idx % nc, j = ja + idx / nc;
        if ((i <= wo || i >= npx + 1 - wo) && (j <= wo || j >= npy + 1 - wo)) continue;
        out(i, j, k) = TD(dd, i, j);
      }
    }
#undef TD
  }
};

// D7: divergence damping of the sponge levels (nord_k = 0, :1290-1371) and the final damping term added to ke (:1446-1458);
// box (is:ie+1, js:je+1)
struct DswCubedD7 {
  DswCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const int npx = g.npx, npy = g.npy, nord = s.a.lv.nord_k[k];
    const double dt = s.a.dt, d2_bg = s.a.lv.d2_divg[k];
    const CA u = cview_U(g, s.a.u), v = cview_V(g, s.a.v), ua = cview_A(g, s.a.ua), va = cview_A(g, s.a.va);
    const CA uc = cview_V(g, s.a.uc), vc = cview_U(g, s.a.vc);
    double vortv, delpc;
    if (nord == 0) {
      auto ptc = [&](int ii, int jj) {
        if (jj == 1 || jj == npy)
          return (vc(ii, jj, k) > 0) ? u(ii, jj, k) * g.dyc[g.iU(ii, jj)] * g.sinsg(ii, jj - 1, 4)
                                     : u(ii, jj, k) * g.dyc[g.iU(ii, jj)] * g.sinsg(ii, jj, 2);
        return (u(ii, jj, k) - 0.5 * (va(ii, jj - 1, k) + va(ii, jj, k)) * g.cosa_v[g.iU(ii, jj)]) * g.dyc[g.iU(ii, jj)] * g.sina_v[g.iU(ii, jj)];
      };
      auto vo = [&](int ii, int jj) {
        if (ii == 1 || ii == npx)
          return (uc(ii, jj, k) > 0) ? v(ii, jj, k) * g.dxc[g.iV(ii, jj)] * g.sinsg(ii - 1, jj, 3)
                                     : v(ii, jj, k) * g.dxc[g.iV(ii, jj)] * g.sinsg(ii, jj, 1);
        return (v(ii, jj, k) - 0.5 * (ua(ii - 1, jj, k) + ua(ii, jj, k)) * g.cosa_u[g.iV(ii, jj)]) * g.dxc[g.iV(ii, jj)] * g.sina_u[g.iV(ii, jj)];
      };
      delpc = vo(i, j - 1) - vo(i, j) + ptc(i - 1, j) - ptc(i, j);
      if (i == 1 && j == 1) delpc = delpc - vo(1, 0);
      if (i == npx && j == 1) delpc = delpc - vo(npx, 0);
      if (i == npx && j == npy) delpc = delpc + vo(npx, npy);
      if (i == 1 && j == npy) delpc = delpc + vo(1, npy);
      delpc = g.rarea_c[g.iB(i, j)] * delpc;
      const double damp = g.da_min_c * dmax(d2_bg, dmin(0.20, s.a.dddmp * fabs(delpc * dt)));
      vortv = damp * delpc;
    } else {
      delpc = cview_B(g, s.a.divg_d)(i, j, k);
      const int n2 = nord + 1;
      const double dd8 = g.stretched_grid ? g.da_min * ipow(s.a.d4_bg, n2) : ipow(g.da_min_c * s.a.d4_bg, n2);
      double vs = 0.;  // dddmp < 1e-5: vort = 0 (:1428-1429); else the Smagorinsky deformation (:1431-1440)
      if (!(s.a.dddmp < 1.E-5)) {
        const double vb = cview_A(g, s.smag)(i, j, k);
        vs = fabs(dt) * sqrt(delpc * delpc + vb * vb);
      }
      const double damp2 = g.da_min_c * dmax(d2_bg, dmin(0.20, s.a.dddmp * vs));
      vortv = damp2 * delpc + dd8 * cview_B(g, s.dd)(i, j, k);
    }
    if (s.a.delpc && s.own(i, j)) view_A(g, s.a.delpc)(i, j, k) = delpc;
    if (s.vortv) view_B(g, s.vortv)(i, j, k) = vortv;
    double &kev = view_B(g, s.ke)(i, j, k);
    kev = kev + vortv;
  }
};

// D8: absolute vorticity on (isd:ied, jsd:jed) (:1477-1481), in place over the relative vorticity
struct DswCubedD8 {
  DswCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    double &w = view_A(g, s.wk)(i, j, k);
    w = w + g.f0[g.iA(i, j)];
  }
};

// D9: the D-grid winds (:1500-1509); box (is:ie+1, js:je+1); fx, fy = the vorticity fluxes
struct DswCubedD9 {
  DswCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const CA ke = cview_B(g, s.ke), u = cview_U(g, s.a.u), v = cview_V(g, s.a.v);
    if (!s.own(i, j)) return;
    if (i <= g.ie)
      view_U(g, s.a.u_out)(i, j, k) = u(i, j, k) * g.dx[g.iU(i, j)] + ke(i, j, k) - ke(i + 1, j, k) + cview_FY(g, s.gy)(i, j, k);
    if (j <= g.je)
      view_V(g, s.a.v_out)(i, j, k) = v(i, j, k) * g.dy[g.iV(i, j)] + ke(i, j, k) - ke(i, j + 1, k) - cview_FX(g, s.gx)(i, j, k);
  }
};

// D10: dissipative heating (:1462-1473, :1523-1586) of the levels with d_con_k > 1e-5, after D9 (u_out, v_out before the
// vorticity damping fluxes are added); box (is:ie, js:je).  Where the level has no vorticity damping the reference's "ut", "vt"
// still hold v*dy, u*dx of the vorticity computation (:1231-1236) -- reproduced.
struct DswCubedD10 {
  DswCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const double d_con = s.a.lv.d_con_k[k];
    const bool est = g.do_diss_est;   // :1462, :1523: every level then; "ut", "vt" are zeroed where nothing damps the vorticity (:1516-1519)
    if (!(d_con > 1.E-5) && !est) return;
    const bool vdamp = s.a.lv.damp_vt[k] > 1.E-5;
    const CA vv = cview_B(g, s.vortv), un = cview_U(g, s.a.u_out), vn = cview_V(g, s.a.v_out);
    const CA u = cview_U(g, s.a.u), v = cview_V(g, s.a.v);
    auto vt_ = [&](int ii, int jj) { return vdamp ? cview_U(g, s.dfy2)(ii, jj, k) : (est ? 0. : u(ii, jj, k) * g.dx[g.iU(ii, jj)]); };
    auto ut_ = [&](int ii, int jj) { return vdamp ? cview_V(g, s.dfx2)(ii, jj, k) : (est ? 0. : v(ii, jj, k) * g.dy[g.iV(ii, jj)]); };
    auto ub = [&](int ii, int jj) { return ((vv(ii, jj, k) - vv(ii + 1, jj, k)) + vt_(ii, jj)) * g.rdx[g.iU(ii, jj)]; };
    auto vb = [&](int ii, int jj) { return ((vv(ii, jj, k) - vv(ii, jj + 1, k)) - ut_(ii, jj)) * g.rdy[g.iV(ii, jj)]; };
    auto fy = [&](int ii, int jj) { return un(ii, jj, k) * g.rdx[g.iU(ii, jj)]; };
    auto fx = [&](int ii, int jj) { return vn(ii, jj, k) * g.rdy[g.iV(ii, jj)]; };
    const double ub0 = ub(i, j), ub1 = ub(i, j + 1), vb0 = vb(i, j), vb1 = vb(i + 1, j);
    const double fy0 = fy(i, j), fy1 = fy(i, j + 1), fx0 = fx(i, j), fx1 = fx(i + 1, j);
    const double gy0 = fy0 * ub0, gy1 = fy1 * ub1, gx0 = fx0 * vb0, gx1 = fx1 * vb1;
    const double u2 = fy0 + fy1, du2 = ub0 + ub1, v2 = fx0 + fx1, dv2 = vb0 + vb1;
    const double damp = 0.25 * d_con;
    const double rs = g.rsin2[g.iA(i, j)], cs = g.cosa_s[g.iA(i, j)];
    const double dpn = cview_A(g, s.a.delp_out)(i, j, k);
    double &h = view_CC(g, s.a.heat_s)(i, j, k);
    double &de = view_CC(g, s.a.diss_e)(i, j, k);
    if (g.prevent_diss_cooling) {
      const double tmp = rs * ((ub0 * ub0 + ub1 * ub1 + vb0 * vb0 + vb1 * vb1) + 2. * (gy0 + gy1 + gx0 + gx1) - cs * (u2 * dv2 + v2 * du2 + du2 * dv2));
      if (d_con > 1.E-5) h = dpn * (h - damp * dmin(0., tmp));
      if (est) de = de - tmp;
    } else {
      const double t2 = (ub0 * ub0 + ub1 * ub1 + vb0 * vb0 + vb1 * vb1) + 2. * (gy0 + gy1 + gx0 + gx1) - cs * (u2 * dv2 + v2 * du2 + du2 * dv2);
      h = dpn * (h - damp * rs * t2);
      if (est) de = de - rs * t2;
    }
  }
};

// D11: the diffusive fluxes of the vorticity damping added to the winds (:1589-1600); box (is:ie+1, js:je+1)
struct DswCubedD11 {
  DswCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    if (!(s.a.lv.damp_vt[k] > 1.E-5)) return;
    if (i <= g.ie) {
      double &un = view_U(g, s.a.u_out)(i, j, k);
      un = un + cview_U(g, s.dfy2)(i, j, k);
    }
    if (j <= g.je) {
      double &vn = view_V(g, s.a.v_out)(i, j, k);
      vn = vn - cview_V(g, s.dfx2)(i, j, k);
    }
  }
};

}  // namespace fv3