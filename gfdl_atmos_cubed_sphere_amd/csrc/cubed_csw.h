// cubed_csw.h -- c_sw (model/sw_core.F90:79-488) on a cubed-sphere face (grid_type < 3, not bounded): d2a2c_vect with the face
// edges and corners (:3006-3345), divergence_corner in its non-orthogonal form (:1798-1843), fill_4corners (:3362-3555, as
// an index map on the reads), the edge forms of the upstream kinetic energy (:316-359), the corner terms of the
// circulation (:390-394) and the edge forms of the vorticity fluxes (:433-472).  Passes (cubed_common.h):
//   P1 utmp, vtmp                      (scratch)      P2 ua, va                 P2c the corner assignments
//   P3 uc, vc (interpolated), ut, vt (time-scaled), divg_d       P4 ke, absolute vorticity (scratch)
//   P5 delpc, ptc, wc, the half-step update of uc, vc
#pragma once

#include "cubed_common.h"
#include "csw_kernel.h"

namespace fv3 {

struct CswCubedState {
  Grid g;
  CswArgs a;
  double *utmp, *vtmp, *ke, *vort;  // A x npz scratch
  // Work copies of the outputs the passes read back (ua, va, ut, vt: A layout; uc: V; vc: U), or null: the outputs themselves (the
  // passes alone, or the hybrid on one stream).  The passes form their intermediates on a frame WIDER than the frame they own, the
  // outer part of it from incomplete stencils -- on one stream the marching kernel overwrites those points afterwards.  With the
  // work copies the passes write an output only where they own it (wr) and read their own copies, so they touch nothing the marching
  // kernel writes and the two can run side by side (fv3_api.hip csw_cubed, two lanes).
  double *ua_w = nullptr, *va_w = nullptr, *uc_w = nullptr, *vc_w = nullptr, *ut_w = nullptr, *vt_w = nullptr;
  FV3_HD const double *uaR() const { return ua_w ? ua_w : a.ua; }
  FV3_HD const double *vaR() const { return va_w ? va_w : a.va; }
  FV3_HD const double *ucR() const { return uc_w ? uc_w : a.uc; }
  FV3_HD const double *vcR() const { return vc_w ? vc_w : a.vc; }
  FV3_HD const double *utR() const { return ut_w ? ut_w : a.ut; }
  FV3_HD const double *vtR() const { return vt_w ? vt_w : a.vt; }
  FV3_HD bool wr(int i, int j) const { return !ua_w || own(i, j); }   // may a pass write the OUTPUT at (i, j)?
  // hybrid: P5 writes only the points of the frame of width own_w along the face edges (0: every point), the marching kernel
  // owns the rest (CswArgs::mask_w); P3 forms the winds (divg 0), the divergence (divg 2) or both (divg 1)
  int own_w = 0, divg = 1;
  FV3_HD bool own(int i, int j) const { return own_w == 0 || i <= own_w || i >= g.npx - own_w || j <= own_w || j >= g.npy - own_w; }
  // metric views
  CA cosa_s, rsin2, dxa, dya, rarea, cosa_u, rsin_u, sina_u, dy, dxc, rdxc, cosa_v, rsin_v, sina_v, dx, dyc, rdyc, rarea_c, fC;
};

inline CswCubedState make_csw_cubed(const Grid &g, const CswArgs &a, double *const scr[4]) {
  CswCubedState s;
  s.g = g;
  s.a = a;
  s.utmp = scr[0]; s.vtmp = scr[1]; s.ke = scr[2]; s.vort = scr[3];
  s.cosa_s = cview_A(g, g.cosa_s); s.rsin2 = cview_A(g, g.rsin2); s.dxa = cview_A(g, g.dxa); s.dya = cview_A(g, g.dya);
  s.rarea = cview_A(g, g.rarea);
  s.cosa_u = cview_V(g, g.cosa_u); s.rsin_u = cview_V(g, g.rsin_u); s.sina_u = cview_V(g, g.sina_u); s.dy = cview_V(g, g.dy);
  s.dxc = cview_V(g, g.dxc); s.rdxc = cview_V(g, g.rdxc);
  s.cosa_v = cview_U(g, g.cosa_v); s.rsin_v = cview_U(g, g.rsin_v); s.sina_v = cview_U(g, g.sina_v); s.dx = cview_U(g, g.dx);
  s.dyc = cview_U(g, g.dyc); s.rdyc = cview_U(g, g.rdyc);
  s.rarea_c = cview_B(g, g.rarea_c); s.fC = cview_B(g, g.fC);
  return s;
}

// P1: D -> A interpolation, (isd:ied, jsd:jed): 4th order inside [npt, npx-npt] x [npt, npy-npt] (npt = 4), two-point
// averages in the strips next to the face edges (:3099-3149); the points the reference leaves at big_number are not read
struct CswCubedP1 {
  CswCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    constexpr double a1 = 0.5625, a2 = -0.0625;
    const Grid &g = s.g;
    const CA u = cview_U(g, s.a.u), v = cview_V(g, s.a.v);
    const VA ut = view_A(g, s.utmp), vt = view_A(g, s.vtmp);
    const int npt = 4;
    const bool box = i >= npt && i <= g.npx - npt && j >= npt && j <= g.npy - npt;
    if (box) {
      ut(i, j, k) = a2 * (u(i, j - 1, k) + u(i, j + 2, k)) + a1 * (u(i, j, k) + u(i, j + 1, k));
      vt(i, j, k) = a2 * (v(i - 1, j, k) + v(i + 2, j, k)) + a1 * (v(i, j, k) + v(i + 1, j, k));
    } else {
      ut(i, j, k) = 0.5 * (u(i, j, k) + u(i, j + 1, k));
      vt(i, j, k) = 0.5 * (v(i, j, k) + v(i + 1, j, k));
    }
  }
};

// P2: contravariant winds at the cell centres, (is-2:ie+2, js-2:je+2) (:3152-3157)
struct CswCubedP2 {
  CswCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const CA ut = cview_A(g, s.utmp), vt = cview_A(g, s.vtmp);
    const double cs = FV3_M(s.cosa_s, i, j), rs = FV3_M(s.rsin2, i, j);
    const double uav = (ut(i, j, k) - vt(i, j, k) * cs) * rs, vav = (vt(i, j, k) - ut(i, j, k) * cs) * rs;
    if (s.ua_w) {
      view_A(g, s.ua_w)(i, j, k) = uav;
      view_A(g, s.va_w)(i, j, k) = vav;
    }
    if (s.wr(i, j)) {
      view_A(g, s.a.ua)(i, j, k) = uav;
      view_A(g, s.a.va)(i, j, k) = vav;
    }
  }
};

// P2c: the corner assignments of utmp, vtmp (:3166-3185, :3260-3279) and of ua, va (:3204-3220, :3280-3295): one thread per
// (m, k), m = 0..2
struct CswCubedP2c {
  CswCubedState s;
  FV3_HD void operator()(int m, int, int k) const {
    const Grid &g = s.g;
    const int npx = g.npx, npy = g.npy, ie = g.ie, je = g.je;
    const VA ut = view_A(g, s.utmp), vt = view_A(g, s.vtmp);
    // every source below is a point no assignment of this pass writes
    {  // Xdir: i = -2..0 resp. 0..2
      const int iw = m - 2, ip = m;
      ut(iw, 0, k) = -vt(0, 1 - iw, k);            // sw
      ut(npx + ip, 0, k) = vt(npx, ip + 1, k);     // se
      ut(npx + ip, npy, k) = -vt(npx, je - ip, k);  // ne
      ut(iw, npy, k) = vt(0, je + iw, k);          // nw
    }
    {  // Ydir: j = -2..0 resp. 0..2
      const int jw = m - 2, jp = m;
      vt(0, jw, k) = -ut(1 - jw, 0, k);             // sw
      vt(0, npy + jp, k) = ut(jp + 1, npy, k);      // nw
      vt(npx, jw, k) = ut(ie + jw, 0, k);           // se
      vt(npx, npy + jp, k) = -ut(ie - jp, npy, k);  // ne
    }
    // (every point below lies within two cells of a face edge: the passes own it, in the outputs and in their work copies alike)
    for (int fam = 0; fam < (s.ua_w ? 2 : 1); fam++) {
    const VA ua = view_A(g, fam ? s.ua_w : s.a.ua), va = view_A(g, fam ? s.va_w : s.a.va);
    if (m == 0) {
      ua(-1, 0, k) = -va(0, 2, k);  ua(0, 0, k) = -va(0, 1, k);                            // sw
      ua(npx, 0, k) = va(npx, 1, k);  ua(npx + 1, 0, k) = va(npx, 2, k);                   // se
      ua(npx, npy, k) = -va(npx, npy - 1, k);  ua(npx + 1, npy, k) = -va(npx, npy - 2, k);  // ne
      ua(-1, npy, k) = va(0, npy - 2, k);  ua(0, npy, k) = va(0, npy - 1, k);              // nw
    } else if (m == 1) {
      va(0, -1, k) = -ua(2, 0, k);  va(0, 0, k) = -ua(1, 0, k);                            // sw
      va(npx, 0, k) = ua(npx - 1, 0, k);  va(npx, -1, k) = ua(npx - 2, 0, k);              // se
      va(npx, npy, k) = -ua(npx - 1, npy, k);  va(npx, npy + 1, k) = -ua(npx - 2, npy, k);  // ne
      va(0, npy, k) = ua(1, npy, k);  va(0, npy + 1, k) = ua(2, npy, k);                   // nw
    }
    }
  }
};

// the corner assignments of ua / va read each other's untouched neighbours only if they run in two steps; P2c's m == 0 / 1
// branches touch disjoint cells: ua(-1..0, 0), ua(npx..npx+1, 0), ... vs va(0, -1..0), ...; the sources va(0, 1..2),
// ua(1..2, 0), ... are regular cells.  (ua(0,0) and va(0,0) are both written: by different m, from regular sources.)

// P3: A -> C interpolation with the edge forms, the time-scaled fluxes and the corner divergence, box
// (is-1:ie+2, js-1:je+2): uc, ut on (is-1:ie+2, js-1:je+1); vc, vt on (is-1:ie+1, js-1:je+2); divg_d on (is:ie+1, js:je+1)
struct CswCubedP3 {
  CswCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    constexpr double a1 = 0.5625, a2 = -0.0625, c1 = -2. / 14., c2 = 11. / 14., c3 = 5. / 14.;
    const Grid &g = s.g;
    const int is = g.is, ie = g.ie, js = g.js, je = g.je, npx = g.npx, npy = g.npy;
    const CA utmp = cview_A(g, s.utmp), vtmp = cview_A(g, s.vtmp), ua = cview_A(g, s.uaR()), va = cview_A(g, s.vaR());
    const CA u = cview_U(g, s.a.u), v = cview_V(g, s.a.v);
    const double dt2 = s.a.dt2;
    const bool wr = s.wr(i, j);
    if (s.divg != 2 && j <= je + 1) {  // uc, ut (:3187-3255)
      double ucv, utv;
      if (i == 1 || i == npx) {
        utv = edge_interpolate4(ua(i - 2, j, k), ua(i - 1, j, k), ua(i, j, k), ua(i + 1, j, k), FV3_M(s.dxa, i - 2, j),
                                FV3_M(s.dxa, i - 1, j), FV3_M(s.dxa, i, j), FV3_M(s.dxa, i + 1, j));
        ucv = (utv > 0.) ? utv * g.sinsg(i - 1, j, 3) : utv * g.sinsg(i, j, 1);  // the UPSTREAM value
      } else {
        if (i == 0 || i == npx - 1)
          ucv = c1 * utmp(i - 2, j, k) + c2 * utmp(i - 1, j, k) + c3 * utmp(i, j, k);
        else if (i == 2)
          ucv = c1 * utmp(3, j, k) + c2 * utmp(2, j, k) + c3 * utmp(1, j, k);
        else if (i == npx + 1)
          ucv = c3 * utmp(npx, j, k) + c2 * utmp(npx + 1, j, k) + c1 * utmp(npx + 2, j, k);
        else
          ucv = a2 * (utmp(i - 2, j, k) + utmp(i + 1, j, k)) + a1 * (utmp(i - 1, j, k) + utmp(i, j, k));
        utv = (ucv - v(i, j, k) * FV3_M(s.cosa_u, i, j)) * FV3_M(s.rsin_u, i, j);
      }
      if (s.uc_w) view_V(g, s.uc_w)(i, j, k) = ucv;
      if (wr) view_V(g, s.a.uc)(i, j, k) = ucv;
      // :159-167
      if (utv > 0.)
        utv = dt2 * utv * FV3_M(s.dy, i, j) * g.sinsg(i - 1, j, 3);
      else
        utv = dt2 * utv * FV3_M(s.dy, i, j) * g.sinsg(i, j, 1);
      if (s.ut_w) view_A(g, s.ut_w)(i, j, k) = utv;
      if (wr) view_A(g, s.a.ut)(i, j, k) = utv;
    }
    if (s.divg != 2 && i <= ie + 1) {  // vc, vt (:3298-3334)
      double vcv, vtv;
      if (j == 1 || j == npy) {
        vtv = edge_interpolate4(va(i, j - 2, k), va(i, j - 1, k), va(i, j, k), va(i, j + 1, k), FV3_M(s.dya, i, j - 2),
                                FV3_M(s.dya, i, j - 1), FV3_M(s.dya, i, j), FV3_M(s.dya, i, j + 1));
        vcv = (vtv > 0.) ? vtv * g.sinsg(i, j - 1, 4) : vtv * g.sinsg(i, j, 2);
      } else {
        if (j == 0 || j == npy - 1)
          vcv = c1 * vtmp(i, j - 2, k) + c2 * vtmp(i, j - 1, k) + c3 * vtmp(i, j, k);
        else if (j == 2 || j == npy + 1)
          vcv = c1 * vtmp(i, j + 1, k) + c2 * vtmp(i, j, k) + c3 * vtmp(i, j - 1, k);
        else
          vcv = a2 * (vtmp(i, j - 2, k) + vtmp(i, j + 1, k)) + a1 * (vtmp(i, j - 1, k) + vtmp(i, j, k));
        vtv = (vcv - u(i, j, k) * FV3_M(s.cosa_v, i, j)) * FV3_M(s.rsin_v, i, j);
      }
      if (s.vc_w) view_U(g, s.vc_w)(i, j, k) = vcv;
      if (wr) view_U(g, s.a.vc)(i, j, k) = vcv;
      // :168-176
      if (vtv > 0.)
        vtv = dt2 * vtv * FV3_M(s.dx, i, j) * g.sinsg(i, j - 1, 4);
      else
        vtv = dt2 * vtv * FV3_M(s.dx, i, j) * g.sinsg(i, j, 2);
      if (s.vt_w) view_A(g, s.vt_w)(i, j, k) = vtv;
      if (wr) view_A(g, s.a.vt)(i, j, k) = vtv;
    }
    if (s.divg != 0 && s.a.nord > 0 && i >= is && i <= ie + 1 && j >= js && j <= je + 1) {  // divergence_corner, :1798-1843
      auto uf = [&](int ii, int jj) {
        if (jj == 1 || jj == npy)
          return u(ii, jj, k) * FV3_M(s.dyc, ii, jj) * 0.5 * (g.sinsg(ii, jj - 1, 4) + g.sinsg(ii, jj, 2));
        return (u(ii, jj, k) - 0.25 * (va(ii, jj - 1, k) + va(ii, jj, k)) * (g.cossg(ii, jj - 1, 4) + g.cossg(ii, jj, 2))) *
               FV3_M(s.dyc, ii, jj) * 0.5 * (g.sinsg(ii, jj - 1, 4) + g.sinsg(ii, jj, 2));
      };
      auto vf = [&](int ii, int jj) {
        if (ii == 1 || ii == npx)
          return v(ii, jj, k) * FV3_M(s.dxc, ii, jj) * 0.5 * (g.sinsg(ii - 1, jj, 3) + g.sinsg(ii, jj, 1));
        return (v(ii, jj, k) - 0.25 * (ua(ii - 1, jj, k) + ua(ii, jj, k)) * (g.cossg(ii - 1, jj, 3) + g.cossg(ii, jj, 1))) *
               FV3_M(s.dxc, ii, jj) * 0.5 * (g.sinsg(ii - 1, jj, 3) + g.sinsg(ii, jj, 1));
      };
      double d = vf(i, j - 1) - vf(i, j) + uf(i - 1, j) - uf(i, j);
      // Remove the extra term at the corners
      if (i == 1 && j == 1) d = d - vf(1, 0);
      if (i == npx && j == 1) d = d - vf(npx, 0);
      if (i == npx && j == npy) d = d + vf(npx, npy);
      if (i == 1 && j == npy) d = d + vf(1, npy);
      view_B(g, s.a.divg_d)(i, j, k) = FV3_M(s.rarea_c, i, j) * d;
    }
  }
};

// P4: upstream kinetic energy on (is-1:ie+1, js-1:je+1) (:316-366) and absolute vorticity on (is:ie+1, js:je+1) (:372-403)
struct CswCubedP4 {
  CswCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const int is = g.is, js = g.js, npx = g.npx, npy = g.npy;
    const CA ua = cview_A(g, s.uaR()), va = cview_A(g, s.vaR()), u = cview_U(g, s.a.u), v = cview_V(g, s.a.v);
    const CA uc = cview_V(g, s.ucR()), vc = cview_U(g, s.vcR());
    const double dt4 = 0.5 * s.a.dt2;
    double kx, ky;
    if (ua(i, j, k) > 0.) {
      if (i == 1)
        kx = uc(1, j, k) * g.sinsg(1, j, 1) + v(1, j, k) * g.cossg(1, j, 1);
      else if (i == npx)
        kx = uc(npx, j, k) * g.sinsg(npx, j, 1) + v(npx, j, k) * g.cossg(npx, j, 1);
      else
        kx = uc(i, j, k);
    } else {
      if (i == 0)
        kx = uc(1, j, k) * g.sinsg(0, j, 3) + v(1, j, k) * g.cossg(0, j, 3);
      else if (i == npx - 1)
        kx = uc(npx, j, k) * g.sinsg(npx - 1, j, 3) + v(npx, j, k) * g.cossg(npx - 1, j, 3);
      else
        kx = uc(i + 1, j, k);
    }
    if (va(i, j, k) > 0.) {
      if (j == 1)
        ky = vc(i, 1, k) * g.sinsg(i, 1, 2) + u(i, 1, k) * g.cossg(i, 1, 2);
      else if (j == npy)
        ky = vc(i, npy, k) * g.sinsg(i, npy, 2) + u(i, npy, k) * g.cossg(i, npy, 2);
      else
        ky = vc(i, j, k);
    } else {
      if (j == 0)
        ky = vc(i, 1, k) * g.sinsg(i, 0, 4) + u(i, 1, k) * g.cossg(i, 0, 4);
      else if (j == npy - 1)
        ky = vc(i, npy, k) * g.sinsg(i, npy - 1, 4) + u(i, npy, k) * g.cossg(i, npy - 1, 4);
      else
        ky = vc(i, j + 1, k);
    }
    view_A(g, s.ke)(i, j, k) = dt4 * (ua(i, j, k) * kx + va(i, j, k) * ky);
    if (i >= is && j >= js) {
      auto fx = [&](int ii, int jj) { return uc(ii, jj, k) * FV3_M(s.dxc, ii, jj); };
      auto fy = [&](int ii, int jj) { return vc(ii, jj, k) * FV3_M(s.dyc, ii, jj); };
      double vo = fx(i, j - 1) - fx(i, j) - fy(i - 1, j) + fy(i, j);
      if (i == 1 && j == 1) vo = vo + fy(0, 1);
      if (i == npx && j == 1) vo = vo - fy(npx, 1);
      if (i == npx && j == npy) vo = vo - fy(npx, npy);
      if (i == 1 && j == npy) vo = vo + fy(0, npy);
      view_A(g, s.vort)(i, j, k) = FV3_M(s.fC, i, j) + FV3_M(s.rarea_c, i, j) * vo;
    }
  }
};

// P5: first-order upwind transport of delp, pt, w on (is-1:ie+1, js-1:je+1) with fill_4corners as an index map on the
// reads (:182-286); the C-grid winds advanced half a step (:433-486)
struct CswCubedP5 {
  CswCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const int is = g.is, ie = g.ie, js = g.js, je = g.je, npx = g.npx, npy = g.npy;
    const CA delp = cview_A(g, s.a.delp), pt = cview_A(g, s.a.pt), w = cview_A(g, s.a.w);
    const CA ut = cview_A(g, s.utR()), vt = cview_A(g, s.vtR()), u = cview_U(g, s.a.u), v = cview_V(g, s.a.v);
    const bool nh = !s.a.hydrostatic;
    const double dt2 = s.a.dt2;
    auto rd = [&](const CA &q, int dir, int ii, int jj) {
      fill4_src(dir, npx, npy, ii, jj);
      return q(ii, jj, k);
    };
    if (!s.own(i, j)) return;
    // ONE memory round trip: every value the point may read is requested before anything is selected or stored -- both upwind
    // candidates of a face (the reference's `if (ut > 0)` picks the cell: a load whose address waits for another load), both vorticity
    // candidates of uc and vc, and the inputs of the uc / vc updates before the stores of delpc / ptc / wc (the compiler cannot know that
    // those arrays are not ke or vort).  The pass is a chain of such round trips, not bandwidth (profiles/r06_pmc_pass.csv: 0.48 GB in
    // 224 us); the statements are the same.
    const double ut0 = ut(i, j, k), ut1 = ut(i + 1, j, k), vt0 = vt(i, j, k), vt1 = vt(i, j + 1, k);
    const double dxm = rd(delp, 1, i - 1, j), dx0 = rd(delp, 1, i, j), dxp = rd(delp, 1, i + 1, j);
    const double dym = rd(delp, 2, i, j - 1), dy0 = rd(delp, 2, i, j), dyp = rd(delp, 2, i, j + 1);
    const double pxm = rd(pt, 1, i - 1, j), px0 = rd(pt, 1, i, j), pxp = rd(pt, 1, i + 1, j);
    const double pym = rd(pt, 2, i, j - 1), py0 = rd(pt, 2, i, j), pyp = rd(pt, 2, i, j + 1);
    double wxm = 0., wx0 = 0., wxp = 0., wym = 0., wy0 = 0., wyp = 0.;
    if (nh) {
      wxm = rd(w, 1, i - 1, j); wx0 = rd(w, 1, i, j); wxp = rd(w, 1, i + 1, j);
      wym = rd(w, 2, i, j - 1); wy0 = rd(w, 2, i, j); wyp = rd(w, 2, i, j + 1);
    }
    const double ra = FV3_M(s.rarea, i, j);
    const CA ke = cview_A(g, s.ke), vort = cview_A(g, s.vort);
    const bool do_uc = i >= is && i <= ie + 1 && j >= js && j <= je;
    const bool do_vc = i >= is && i <= ie && j >= js && j <= je + 1;
    double ucv = 0., vu = 0., cu = 0., su = 1., vo00 = 0., vo01 = 0., vo10 = 0., rdx = 0., kem0 = 0., ke00 = 0., ke0m = 0.;
    double vcv = 0., uv = 0., cv = 0., sv = 1., rdy = 0.;
    if (do_uc || do_vc) { vo00 = vort(i, j, k); ke00 = ke(i, j, k); }
    if (do_uc) {
      ucv = cview_V(g, s.ucR())(i, j, k); vu = v(i, j, k); vo01 = vort(i, j + 1, k); rdx = FV3_M(s.rdxc, i, j); kem0 = ke(i - 1, j, k);
      if (!(i == 1 || i == npx)) { cu = FV3_M(s.cosa_u, i, j); su = FV3_M(s.sina_u, i, j); }
    }
    if (do_vc) {
      vcv = cview_U(g, s.vcR())(i, j, k); uv = u(i, j, k); vo10 = vort(i + 1, j, k); rdy = FV3_M(s.rdyc, i, j); ke0m = ke(i, j - 1, k);
      if (!(j == 1 || j == npy)) { cv = FV3_M(s.cosa_v, i, j); sv = FV3_M(s.sina_v, i, j); }
    }
    {
      // x faces i, i+1 and y faces j, j+1 of the cell
      double fx1[2], fxp[2], fxw[2], fy1[2], fyp[2], fyw[2];
      fx1[0] = ut0 * ((ut0 > 0.) ? dxm : dx0); fx1[1] = ut1 * ((ut1 > 0.) ? dx0 : dxp);
      fxp[0] = fx1[0] * ((ut0 > 0.) ? pxm : px0); fxp[1] = fx1[1] * ((ut1 > 0.) ? px0 : pxp);
      fxw[0] = nh ? fx1[0] * ((ut0 > 0.) ? wxm : wx0) : 0.; fxw[1] = nh ? fx1[1] * ((ut1 > 0.) ? wx0 : wxp) : 0.;
      fy1[0] = vt0 * ((vt0 > 0.) ? dym : dy0); fy1[1] = vt1 * ((vt1 > 0.) ? dy0 : dyp);
      fyp[0] = fy1[0] * ((vt0 > 0.) ? pym : py0); fyp[1] = fy1[1] * ((vt1 > 0.) ? py0 : pyp);
      fyw[0] = nh ? fy1[0] * ((vt0 > 0.) ? wym : wy0) : 0.; fyw[1] = nh ? fy1[1] * ((vt1 > 0.) ? wy0 : wyp) : 0.;
      // the four corner cells of the box hold what the second fill (dir = 2) left there when the reference updates them
      const double dp = dy0;
      const double dpc = dp + (fx1[0] - fx1[1] + fy1[0] - fy1[1]) * ra;
      view_A(g, s.a.delpc)(i, j, k) = dpc;
      view_A(g, s.a.ptc)(i, j, k) = (py0 * dp + (fxp[0] - fxp[1] + fyp[0] - fyp[1]) * ra) / dpc;
      if (nh) view_A(g, s.a.wc)(i, j, k) = (wy0 * dp + (fxw[0] - fxw[1] + fyw[0] - fyw[1]) * ra) / dpc;
    }
    if (do_uc) {
      double fy1;
      if (i == 1 || i == npx)
        fy1 = dt2 * vu;
      else
        fy1 = dt2 * (vu - ucv * cu) / su;
      const double fy = (fy1 > 0.) ? vo00 : vo01;
      view_V(g, s.a.uc)(i, j, k) = ucv + fy1 * fy + rdx * (kem0 - ke00);
    }
    if (do_vc) {
      double fx1;
      if (j == 1 || j == npy)
        fx1 = dt2 * uv;
      else
        fx1 = dt2 * (uv - vcv * cv) / sv;
      const double fx = (fx1 > 0.) ? vo00 : vo10;
      view_U(g, s.a.vc)(i, j, k) = vcv - fx1 * fx + rdy * (ke0m - ke00);
    }
  }
};

}  // namespace fv3
