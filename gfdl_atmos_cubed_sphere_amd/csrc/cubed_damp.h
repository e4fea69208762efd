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

}  // namespace fv3
