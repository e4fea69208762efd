// cubed_damp.h -- the del-2n damping operators on a cubed-sphere face as passes (cubed_common.h):
//   deln_flux     (model/tp_core.F90:1267-1447): damping fluxes added to the fluxes of fv_tp_2d (delp with nord_v / damp_v,
//                 pt and q_con mass-weighted with nord_t / damp_t);
//   del6_vt_flux  (model/sw_core.F90:1610-1730): the same operator returning its fluxes (w with nord_w / damp_w, the relative
//                 vorticity with nord_v / damp_v).
// copy_corners (tp_core.F90:245-322) before every x / y difference is an index map on the reads of d2 (copyc_src).  The
// order and the damping coefficient are per level (DswLevels); a level is active when its coefficient exceeds `thresh`.
//   L1  d2 = damp * q  (or q itself when the fluxes are mass-weighted)              box (isd:ied, jsd:jed)
//   L2  fx2 = del6_v * (d2(i-1) - d2(i)), fy2 = del6_u * (d2(j-1) - d2(j))          box (isd:ied+1, jsd:jed+1)
//   n = 1 .. nord:  L3  d2 = div(fx2, fy2) * rarea on the box shrunk to nt = nord - n;  L4  fx2, fy2 with the sign of :1363
//   L5  fx += fx2 (mass-weighted: fx += 0.5 * damp * (mass(i-1) + mass(i)) * fx2), deln_flux only
#pragma once

#include "cubed_common.h"

namespace fv3 {

struct DelnCubedState {
  Grid g;
  const double *q;           // A x npz
  const double *mass;        // A x npz or null
  double *d2, *fx2, *fy2;    // work arrays: A, V (nid+1 wide), U layouts
  double *fx, *fy;           // FX / FY: the fluxes the damping is added to (L5)
  const int *nord;           // per level
  const double *coef;        // per level: damp_v / damp_t / damp_w
  double thresh;             // 1e-4 inside fv_tp_2d (tp_core.F90:230), 1e-5 for del6_vt_flux (sw_core.F90:951, :1513)
  int corner_area;           // 1: damp = (coef * da_min_c)^(nord+1) (sw_core.F90:953, :1514); 0: (coef * da_min)^(nord+1);
                             // 2: damp = coef as it is (update_dz_d hands its damp(k) to del6_vt_flux, nh_utils.F90:278)
  FV3_HD bool active(int k) const { return coef[k] > thresh; }
  FV3_HD double damp(int k) const {
    return corner_area == 2 ? coef[k] : ipow(coef[k] * (corner_area ? g.da_min_c : g.da_min), nord[k] + 1);
  }
  // d2 read for a difference along direction dir: through the copy_corners map when the level has nord > 0
  FV3_HD double D(int dir, int i, int j, int k) const {
    if (nord[k] > 0) copyc_src(dir, g.npx, g.npy, i, j);
    return cview_A(g, d2)(i, j, k);
  }
};

struct DelnCubedL1 {
  DelnCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    if (!s.active(k)) return;
    const Grid &g = s.g;
    const int n = s.nord[k];
    if (i < g.is - 1 - n || i > g.ie + 1 + n || j < g.js - 1 - n || j > g.je + 1 + n) return;
    const double qv = cview_A(g, s.q)(i, j, k);
    view_A(g, s.d2)(i, j, k) = s.mass ? qv : s.damp(k) * qv;
  }
};

// first = 1: the differences of L2 (d2(i-1) - d2(i)) on the box of order nord; first = 0: iteration n (d2(i) - d2(i-1)) on nt
struct DelnCubedL24 {
  DelnCubedState s;
  int first, n;
  FV3_HD void operator()(int i, int j, int k) const {
    if (!s.active(k)) return;
    const Grid &g = s.g;
    const int nord = s.nord[k];
    if (!first && n > nord) return;
    const int nt = first ? nord : nord - n;
    if (j >= g.js - nt && j <= g.je + nt && i >= g.is - nt && i <= g.ie + nt + 1) {
      const double a = s.D(1, i - 1, j, k), b = s.D(1, i, j, k);
      view_V(g, s.fx2)(i, j, k) = g.del6_v[g.iV(i, j)] * (first ? a - b : b - a);
    }
    if (j >= g.js - nt && j <= g.je + nt + 1 && i >= g.is - nt && i <= g.ie + nt) {
      const double a = s.D(2, i, j - 1, k), b = s.D(2, i, j, k);
      view_U(g, s.fy2)(i, j, k) = g.del6_u[g.iU(i, j)] * (first ? a - b : b - a);
    }
  }
};

struct DelnCubedL3 {
  DelnCubedState s;
  int n;
  FV3_HD void operator()(int i, int j, int k) const {
    if (!s.active(k)) return;
    const Grid &g = s.g;
    const int nord = s.nord[k];
    if (n > nord) return;
    const int nt = nord - n;
    if (i < g.is - nt - 1 || i > g.ie + nt + 1 || j < g.js - nt - 1 || j > g.je + nt + 1) return;
    const CA fx2 = cview_V(g, s.fx2), fy2 = cview_U(g, s.fy2);
    view_A(g, s.d2)(i, j, k) = (fx2(i, j, k) - fx2(i + 1, j, k) + fy2(i, j, k) - fy2(i, j + 1, k)) * g.rarea[g.iA(i, j)];
  }
};

struct DelnCubedL5 {
  DelnCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    if (!s.active(k)) return;
    const Grid &g = s.g;
    const CA fx2 = cview_V(g, s.fx2), fy2 = cview_U(g, s.fy2);
    if (s.mass) {
      const CA m = cview_A(g, s.mass);
      const double damp2 = 0.5 * s.damp(k);
      if (j <= g.je) {
        double &f = view_FX(g, s.fx)(i, j, k);
        f = f + damp2 * (m(i - 1, j, k) + m(i, j, k)) * fx2(i, j, k);
      }
      if (i <= g.ie) {
        double &f = view_FY(g, s.fy)(i, j, k);
        f = f + damp2 * (m(i, j - 1, k) + m(i, j, k)) * fy2(i, j, k);
      }
    } else {
      if (j <= g.je) {
        double &f = view_FX(g, s.fx)(i, j, k);
        f = f + fx2(i, j, k);
      }
      if (i <= g.ie) {
        double &f = view_FY(g, s.fy)(i, j, k);
        f = f + fy2(i, j, k);
      }
    }
  }
};

// The whole chain of one operator in ONE launch, away from the face corners.  The passes above are 7 launches for nord = 2, each
// of them two or three arrays through HBM (0.95 ms per operator on a C384 L127 face; a production namelist runs four of them per
// d_sw: 3.4 of its 14 ms).  Here a workgroup owns a 32 x 8 rectangle of flux points at one level, stages d2 = damp * q on the
// rectangle grown by nord + 1, and alternates fluxes and divergences in LDS on rectangles that shrink by one per iteration; every
// value is the expression of L1 .. L5 on the same operands.  What it does NOT do is copy_corners: the index maps on the reads of
// d2 only differ from the identity in the ghost corners of a face, and only flux points within nord + 1 of BOTH edges of a corner
// depend on those -- they are inside the four corner squares of side `wo`, which this kernel leaves to the passes (launched first,
// on those squares and the rim their intermediates need, CornerPass: the kernel then overwrites the rim).
struct DelnFused {
  static constexpr int TI = 32, TJ = 16, kMaxN = 2;
  static constexpr int PW = TI + 2 * kMaxN + 2, PH = TJ + 2 * kMaxN + 2;   // cells [ia-1-n, ib+n] (+1: the face ib+n+1)
  static constexpr int lds_doubles = 3 * PW * PH;
  DelnCubedState s;
  int wo;                 // flux points within wo of BOTH edges of a face corner are the passes'
  const int *klist;
  int raw;                // 1: fx2 / fy2 themselves to s.fx2 / s.fy2 (del6_vt_flux); 0: added to s.fx / s.fy (deln_flux, L5)
  FV3_D void operator()(int bx, int by, int bz, int tid, double *lds) const {
    const int k = klist ? klist[bz] : bz;
    if (!s.active(k)) return;
    const Grid &g = s.g;
    const int n = s.nord[k], npx = g.npx, npy = g.npy;
    const int ia = g.is + bx * TI, ja = g.js + by * TJ;
    const int ib = ia + TI - 1 < g.ie + 1 ? ia + TI - 1 : g.ie + 1, jb = ja + TJ - 1 < g.je + 1 ? ja + TJ - 1 : g.je + 1;
    const int i0 = ia - 1 - kMaxN, j0 = ja - 1 - kMaxN;
    double *d2 = lds, *fx2 = lds + PW * PH, *fy2 = lds + 2 * PW * PH;
#define TD(a, i, j) (a)[((j) - j0) * PW + ((i) - i0)]
    const double damp = s.damp(k);
    {   // L1 on the cells [ia-1-n, ib+n] x [ja-1-n, jb+n]
      const CA q = cview_A(g, s.q);
      const int ca = ia - 1 - n, cb = ib + n, ra = ja - 1 - n, rb = jb + n, nc = cb - ca + 1, nr = rb - ra + 1;
      for (int idx = tid; idx < nc * nr; idx += kNT) {
        const int i = ca + idx % nc, j = ra + idx / nc;
        const double qv = q(i, j, k);
        TD(d2, i, j) = s.mass ? qv : damp * qv;
      }
    }
    FV3_SYNC();
    for (int it = 0; it <= n; it++) {
      // cells of the NEXT d2 (it < n): [ia-1-m, ib+m] x [ja-1-m, jb+m], m = n - it - 1; their faces; it == n: the rectangle's own faces
      const int m = n - it - 1;
      const bool last = it == n;
      const int ca = last ? ia : ia - 1 - m, cb = last ? ib : ib + m, ra = last ? ja : ja - 1 - m, rb = last ? jb : jb + m;
      {   // fx2 on faces [ca, cb + 1] x rows [ra, rb]  (last: [ia, ib] x [ja, jb]); fy2 on columns [ca, cb] x faces [ra, rb + 1]
        const int xa = ca, xb = last ? cb : cb + 1, nxf = xb - xa + 1, nr = rb - ra + 1;
        for (int idx = tid; idx < nxf * nr; idx += kNT) {
          const int i = xa + idx % nxf, j = ra + idx / nxf;
          const double a = TD(d2, i - 1, j), b = TD(d2, i, j);
          TD(fx2, i, j) = g.del6_v[g.iV(i, j)] * (it == 0 ? a - b : b - a);
        }
        const int ya = ra, yb = last ? rb : rb + 1, nyf = yb - ya + 1, nc = cb - ca + 1;
        for (int idx = tid; idx < nc * nyf; idx += kNT) {
          const int i = ca + idx % nc, j = ya + idx / nc;
          const double a = TD(d2, i, j - 1), b = TD(d2, i, j);
          TD(fy2, i, j) = g.del6_u[g.iU(i, j)] * (it == 0 ? a - b : b - a);
        }
      }
      FV3_SYNC();
      if (!last) {   // L3
        const int nc = cb - ca + 1, nr = rb - ra + 1;
        for (int idx = tid; idx < nc * nr; idx += kNT) {
          const int i = ca + idx % nc, j = ra + idx / nc;
          TD(d2, i, j) = (TD(fx2, i, j) - TD(fx2, i + 1, j) + TD(fy2, i, j) - TD(fy2, i, j + 1)) * g.rarea[g.iA(i, j)];
        }
        FV3_SYNC();
      }
    }
    {   // L5 / the fluxes themselves, outside the passes' frame
      const int nc = ib - ia + 1, nr = jb - ja + 1;
      const double damp2 = 0.5 * damp;
      for (int idx = tid; idx < nc * nr; idx += kNT) {
        const int i = ia + idx % nc, j = ja + idx / nc;
        if ((i <= wo || i >= npx - wo) && (j <= wo || j >= npy - wo)) continue;
        if (j <= g.je) {
          const double f2 = TD(fx2, i, j);
          if (raw) {
            view_V(g, s.fx2)(i, j, k) = f2;
          } else {
            double &f = view_FX(g, s.fx)(i, j, k);
            if (s.mass) { const CA mm = cview_A(g, s.mass); f = f + damp2 * (mm(i - 1, j, k) + mm(i, j, k)) * f2; }
            else f = f + f2;
          }
        }
        if (i <= g.ie) {
          const double f2 = TD(fy2, i, j);
          if (raw) {
            view_U(g, s.fy2)(i, j, k) = f2;
          } else {
            double &f = view_FY(g, s.fy)(i, j, k);
            if (s.mass) { const CA mm = cview_A(g, s.mass); f = f + damp2 * (mm(i, j - 1, k) + mm(i, j, k)) * f2; }
            else f = f + f2;
          }
        }
      }
    }
#undef TD
  }
};

}  // namespace fv3
