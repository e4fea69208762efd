// cubed_tp.h -- fv_tp_2d (model/tp_core.F90:85-241) on a cubed-sphere face: xppm / yppm with the face-edge forms
// (:349-397, :534-547, :643-681 and their yppm twins), copy_corners (:245-322) as an index map on the reads of q.
// One face value per thread: the (bl, br) of the upwind cell are formed from the reference's cell formulas, where the cells
// 0, 1, 2 and npx-2, npx-1, npx of a line that crosses a face edge take the edge forms.
#pragma once

#include "cubed_common.h"
#include "ppm.h"

namespace fv3 {

// standard PPM constraint of one cell: pert_ppm with iv /= 0 (tp_core.F90:1243-1262)
FV3_HD void pert_ppm_std(double &al, double &ar) {
  if (al * ar < 0.) {
    const double da1 = al - ar, da2 = da1 * da1, a6da = 3. * (al + ar) * da1;
    if (a6da < -da2)
      ar = -2. * al;
    else if (a6da > da2)
      al = -2. * ar;
  } else {
    al = 0.;
    ar = 0.;
  }
}

// value of the two-sided edge interpolation at a face edge (tp_core.F90:376-377): cells e-1 | e, Q(m) = line value, D(m) = width
template <class Q, class D>
FV3_HD double edge_value(const Q &q, const D &d, int e) {
  return 0.5 * (((2. * d(e - 1) + d(e - 2)) * q(e - 1) - d(e - 1) * q(e - 2)) / (d(e - 2) + d(e - 1)) +
                ((2. * d(e) + d(e + 1)) * q(e) - d(e) * q(e + 1)) / (d(e) + d(e + 1)));
}

// (bl, br) of cell ic for the monotone family iord >= 8 (tp_core.F90:570-681)
template <class Q, class D>
FV3_HD void ppm_cell_mono_cs(const Q &q, const D &d, int ic, int iord, int npx, double &bl, double &br) {
  constexpr double r3 = 1. / 3., near_zero = 1.E-25, r12 = 1. / 12.;
  constexpr double s11 = 11. / 14., s14 = 4. / 7., s15 = 3. / 14.;
  auto dm = [&](int m) { return ppm_dm(q(m - 1), q(m), q(m + 1)); };
  auto al = [&](int m) { return 0.5 * (q(m - 1) + q(m)) + r3 * (dm(m - 1) - dm(m)); };
  const double q0 = q(ic);
  if (ic >= 0 && ic <= 2) {  // west / south edge, :644-661
    auto xe = [&]() {
      double xt = edge_value(q, d, 1);
      xt = dmax(xt, dmin(dmin(q(-1), q(0)), dmin(q(1), q(2))));
      return dmin(xt, dmax(dmax(q(-1), q(0)), dmax(q(1), q(2))));
    };
    if (ic == 0) {
      bl = s14 * dm(-1) + s11 * (q(-1) - q(0));
      br = xe() - q(0);
    } else if (ic == 1) {
      bl = xe() - q(1);
      br = (s15 * q(1) + s11 * q(2) - s14 * dm(2)) - q(1);
    } else {
      bl = (s15 * q(1) + s11 * q(2) - s14 * dm(2)) - q(2);
      br = al(3) - q(2);
    }
    pert_ppm_std(bl, br);
    return;
  }
  if (ic >= npx - 2 && ic <= npx) {  // east / north edge, :662-680
    auto xe = [&]() {
      double xt = edge_value(q, d, npx);
      xt = dmax(xt, dmin(dmin(q(npx - 2), q(npx - 1)), dmin(q(npx), q(npx + 1))));
      return dmin(xt, dmax(dmax(q(npx - 2), q(npx - 1)), dmax(q(npx), q(npx + 1))));
    };
    if (ic == npx - 2) {
      bl = al(npx - 2) - q(npx - 2);
      br = (s15 * q(npx - 1) + s11 * q(npx - 2) + s14 * dm(npx - 2)) - q(npx - 2);
    } else if (ic == npx - 1) {
      bl = (s15 * q(npx - 1) + s11 * q(npx - 2) + s14 * dm(npx - 2)) - q(npx - 1);
      br = xe() - q(npx - 1);
    } else {
      bl = xe() - q(npx);
      br = s11 * (q(npx + 1) - q(npx)) - s14 * dm(npx + 1);
    }
    pert_ppm_std(bl, br);
    return;
  }
  const double dmm = dm(ic - 1), dm0 = dm(ic), dmp = dm(ic + 1);
  const double qm1 = q(ic - 1), qp1 = q(ic + 1);
  const double al0 = 0.5 * (qm1 + q0) + r3 * (dmm - dm0), al1 = 0.5 * (q0 + qp1) + r3 * (dm0 - dmp);
  if (iord == 8) {
    const double xt = 2. * dm0;
    bl = -fsign(dmin(fabs(xt), fabs(al0 - q0)), xt);
    br = fsign(dmin(fabs(xt), fabs(al1 - q0)), xt);
  } else if (iord == 11) {
    const double xt = 1.5 * dm0;
    bl = -fsign(dmin(fabs(xt), fabs(al0 - q0)), xt);
    br = fsign(dmin(fabs(xt), fabs(al1 - q0)), xt);
  } else if (iord == 12 || iord == 7 || iord == 9 || iord == 13) {
    bl = al0 - q0;
    br = al1 - q0;
    const bool pert = iord != 12 && iord != 7;
    if (pert && q0 <= 0.) {
      bl = 0.;
      br = 0.;
    } else {
      const double a4 = -3. * (bl + br), da1 = br - bl;
      if (fabs(da1) < -a4 && q0 + 0.25 / a4 * (da1 * da1) + a4 * r12 < 0.) {
        const bool both = pert ? (br > 0. && bl > 0.) : (br * bl > 0.);
        if (both) {
          br = 0.;
          bl = 0.;
        } else if (da1 > 0.) {
          br = -2. * bl;
        } else {
          bl = -2. * br;
        }
      }
    }
  } else {  // 10
    bl = al0 - q0;
    br = al1 - q0;
    if (fabs(dmm) + fabs(dm0) + fabs(dmp) < near_zero) {
      bl = 0.;
      br = 0.;
    } else if (fabs(3. * (bl + br)) > fabs(bl - br)) {
      const double pmp_2 = 2. * (q0 - qm1);
      const double lac_2 = pmp_2 - 0.75 * (2. * (qm1 - q(ic - 2)));
      br = dmin(dmax3(0., pmp_2, lac_2), dmax(br, dmin3(0., pmp_2, lac_2)));
      const double pmp_1 = -(2. * (qp1 - q0));
      const double lac_1 = pmp_1 + 0.75 * (2. * (q(ic + 2) - qp1));
      bl = dmin(dmax3(0., pmp_1, lac_1), dmax(bl, dmin3(0., pmp_1, lac_1)));
    }
  }
}

// face value at face i of a line (between cells i-1 and i) with Courant number c; Q / D: accessors of the line
template <class Q, class D>
FV3_HD double ppm_face_cs(const Q &q, const D &d, int i, double c, int iord, int npx) {
  constexpr double r12 = 1. / 12., p1 = 7. / 12., p2 = -1. / 12., c1 = -2. / 14., c2 = 11. / 14., c3 = 5. / 14.;
  if (iord == 7) {  // :685-699: both cells of the face, the flux form of the unlimited family
    double blm, brm, bl0, br0;
    ppm_cell_mono_cs(q, d, i - 1, iord, npx, blm, brm);
    ppm_cell_mono_cs(q, d, i, iord, npx, bl0, br0);
    const bool sm = blm * brm < 0., s0 = bl0 * br0 < 0.;
    double fx1, flux;
    if (c > 0.) {
      fx1 = (1. - c) * (brm - c * (blm + brm));
      flux = q(i - 1);
    } else {
      fx1 = (1. + c) * (bl0 + c * (bl0 + br0));
      flux = q(i);
    }
    if (sm || s0) flux = flux + fx1;
    return flux;
  }
  if (iord >= 8) {
    const int ic = (c > 0.) ? i - 1 : i;
    double bl, br;
    ppm_cell_mono_cs(q, d, ic, iord, npx, bl, br);
    const double q0 = q(ic);
    if (c > 0.) return q0 + (1. - c) * (br - c * (bl + br));
    return q0 + (1. + c) * (bl + c * (bl + br));
  }
  // iord = 5, -5, 6 (:365-560)
  auto al = [&](int m) {
    double a;
    if (m == 0 || m == npx - 1)
      a = c1 * q(m - 2) + c2 * q(m - 1) + c3 * q(m);
    else if (m == 1 || m == npx)
      a = edge_value(q, d, m);
    else if (m == 2 || m == npx + 1)
      a = c3 * q(m - 1) + c2 * q(m) + c1 * q(m + 1);
    else
      a = p1 * (q(m - 1) + q(m)) + p2 * (q(m - 2) + q(m + 1));
    return iord < 0 ? dmax(0., a) : a;
  };
  const double alm = al(i - 1), al0 = al(i), alp = al(i + 1);
  const double qm1 = q(i - 1), q0 = q(i);
  const double blm = alm - qm1, brm = al0 - qm1, b0m = blm + brm;
  const double bl0 = al0 - q0, br0 = alp - q0, b00 = bl0 + br0;
  auto edge_cell = [&](int m) { return m == 0 || m == 1 || m == npx - 1 || m == npx; };
  bool sm, s0;
  if (iord == 6) {
    sm = edge_cell(i - 1) ? (blm * brm < 0.) : (3. * fabs(b0m) < fabs(blm - brm));  // :534-546
    s0 = edge_cell(i) ? (bl0 * br0 < 0.) : (3. * fabs(b00) < fabs(bl0 - br0));
  } else {
    sm = blm * brm < 0.;
    s0 = bl0 * br0 < 0.;
  }
  double bl, br, b0, qu;
  bool su;
  if (c > 0.) {
    bl = blm; br = brm; b0 = b0m; qu = qm1; su = sm;
  } else {
    bl = bl0; br = br0; b0 = b00; qu = q0; su = s0;
  }
  if (iord == -5) {
    const double da1 = br - bl, a4 = -3. * b0;
    if (fabs(da1) < -a4) {
      if (qu + 0.25 / a4 * (da1 * da1) + a4 * r12 < 0.) {
        if (!su) {
          br = 0.; bl = 0.; b0 = 0.;
        } else if (da1 > 0.) {
          br = -2. * bl; b0 = -bl;
        } else {
          bl = -2. * br; b0 = -br;
        }
      }
    }
  }
  double fx1;
  if (c > 0.)
    fx1 = (1. - c) * (br - c * b0);
  else
    fx1 = (1. + c) * (bl + c * b0);
  double flux = qu;
  if (sm || s0) flux = flux + fx1;
  return flux;
}

struct Tp2dCubedState {
  Grid g;
  const double *q;                       // A x nk
  const double *crx, *cry, *xfx, *yfx;   // CX / CY x nk
  const double *ra_x, *ra_y;             // (is:ie, jsd:jed) / (isd:ied, js:je) x nk, or null: area + xfx(i) - xfx(i+1)
  const double *mfx, *mfy;               // FX / FY or null
  double *fx, *fy;                       // FX / FY
  double *fx2, *fy2, *q_i, *q_j;         // scratch on the A layout
  int hord;
};

// T1: inner sweeps on the field itself: fy2 on (isd:ied, js:je+1) with copy_corners(q, 2), fx2 on (is:ie+1, jsd:jed) with
// copy_corners(q, 1) (tp_core.F90:143-168); box (isd:ied, jsd:jed)
struct Tp2dCubedT1 {
  Tp2dCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const int npx = g.npx, npy = g.npy;
    const int ord_in = (s.hord == 10) ? 8 : s.hord;
    const CA q = cview_A(g, s.q), dxa = cview_A(g, g.dxa), dya = cview_A(g, g.dya);
    if (j >= g.js && j <= g.je + 1) {
      auto ql = [&](int m) { int ii = i, jj = m; copyc_src(2, npx, npy, ii, jj); return q(ii, jj, k); };
      auto dl = [&](int m) { return FV3_M(dya, i, m); };
      view_A(g, s.fy2)(i, j, k) = ppm_face_cs(ql, dl, j, cview_CY(g, s.cry)(i, j, k), ord_in, npy);
    }
    if (i >= g.is && i <= g.ie + 1) {
      auto ql = [&](int m) { int ii = m, jj = j; copyc_src(1, npx, npy, ii, jj); return q(ii, jj, k); };
      auto dl = [&](int m) { return FV3_M(dxa, m, j); };
      view_A(g, s.fx2)(i, j, k) = ppm_face_cs(ql, dl, i, cview_CX(g, s.crx)(i, j, k), ord_in, npx);
    }
  }
};

// T2: q_i on (isd:ied, js:je), q_j on (is:ie, jsd:jed) (:150-159, :171-178); box (isd:ied, jsd:jed)
struct Tp2dCubedT2 {
  Tp2dCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const CA q = cview_A(g, s.q), area = cview_A(g, g.area), fx2 = cview_A(g, s.fx2), fy2 = cview_A(g, s.fy2);
    const CA xfx = cview_CX(g, s.xfx), yfx = cview_CY(g, s.yfx);
    if (j >= g.js && j <= g.je) {
      const double y0 = yfx(i, j, k), y1 = yfx(i, j + 1, k), ar = FV3_M(area, i, j);
      const double fyy0 = y0 * fy2(i, j, k), fyy1 = y1 * fy2(i, j + 1, k);
      const double ray = s.ra_y ? s.ra_y[(size_t)k * g.nRY() + g.iRY(i, j)] : (ar + y0 - y1);
      view_A(g, s.q_i)(i, j, k) = (q(i, j, k) * ar + fyy0 - fyy1) / ray;
    }
    if (i >= g.is && i <= g.ie) {
      const double x0 = xfx(i, j, k), x1 = xfx(i + 1, j, k), ar = FV3_M(area, i, j);
      const double fx10 = x0 * fx2(i, j, k), fx11 = x1 * fx2(i + 1, j, k);
      const double rax = s.ra_x ? s.ra_x[(size_t)k * g.nRX() + g.iRX(i, j)] : (ar + x0 - x1);
      view_A(g, s.q_j)(i, j, k) = (q(i, j, k) * ar + fx10 - fx11) / rax;
    }
  }
};

// T3: outer sweeps and flux averaging (:161, :180, :187-224); box (is:ie+1, js:je+1)
struct Tp2dCubedT3 {
  Tp2dCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    const Grid &g = s.g;
    const CA qi = cview_A(g, s.q_i), qj = cview_A(g, s.q_j), dxa = cview_A(g, g.dxa), dya = cview_A(g, g.dya);
    const CA fx2 = cview_A(g, s.fx2), fy2 = cview_A(g, s.fy2);
    if (j <= g.je) {
      auto ql = [&](int m) { return qi(m, j, k); };
      auto dl = [&](int m) { return FV3_M(dxa, m, j); };
      const double f = ppm_face_cs(ql, dl, i, cview_CX(g, s.crx)(i, j, k), s.hord, g.npx);
      const double m = s.mfx ? cview_FX(g, s.mfx)(i, j, k) : cview_CX(g, s.xfx)(i, j, k);
      view_FX(g, s.fx)(i, j, k) = 0.5 * (f + fx2(i, j, k)) * m;
    }
    if (i <= g.ie) {
      auto ql = [&](int m) { return qj(i, m, k); };
      auto dl = [&](int m) { return FV3_M(dya, i, m); };
      const double f = ppm_face_cs(ql, dl, j, cview_CY(g, s.cry)(i, j, k), s.hord, g.npy);
      const double m = s.mfy ? cview_FY(g, s.mfy)(i, j, k) : cview_CY(g, s.yfx)(i, j, k);
      view_FY(g, s.fy)(i, j, k) = 0.5 * (f + fy2(i, j, k)) * m;
    }
  }
};

// update_dz_d (nh_utils.F90:261-306): the new interface height from the fluxes of fv_tp_2d
struct ZhCubedFinal {
  Grid g;
  const double *zh_in, *fx, *fy, *xfa, *yfa;
  double *zh_out;
  const double *damp = nullptr;                     // per level (npz + 1): del6_vt_flux damping of the levels with damp > 1e-5 ...
  const double *fx2 = nullptr, *fy2 = nullptr;      // ... its fluxes (V / U layouts), nh_utils.F90:278-284
  FV3_HD void operator()(int i, int j, int k) const {
    const CA z = cview_A(g, zh_in), fxv = cview_FX(g, fx), fyv = cview_FY(g, fy), xf = cview_CX(g, xfa), yf = cview_CY(g, yfa);
    const double ar = g.area[g.iA(i, j)];
    const double rax = ar + xf(i, j, k) - xf(i + 1, j, k), ray = ar + yf(i, j, k) - yf(i, j + 1, k);
    double v = (z(i, j, k) * ar + fxv(i, j, k) - fxv(i + 1, j, k) + fyv(i, j, k) - fyv(i, j + 1, k)) / (rax + ray - ar);
    if (damp && damp[k] > 1.E-5) {
      const CA f2 = cview_V(g, fx2), g2 = cview_U(g, fy2);
      v = v + (f2(i, j, k) - f2(i + 1, j, k) + g2(i, j, k) - g2(i, j + 1, k)) * g.rarea[g.iA(i, j)];
    }
    view_A(g, zh_out)(i, j, k) = v;
  }
};

// tracer_2d (fv_tracer2d.F90:497-533): one tracer's update from the fluxes of fv_tp_2d; levels that have finished their
// sub-cycles carry q (and dp1) over
struct TracerCubedFinal {
  Grid g;
  int it, nsplt, last;
  const int *ksplt;
  const double *q, *dp1, *fx, *fy, *mfx, *mfy;
  double *q_out, *dp1_out;
  FV3_HD void operator()(int i, int j, int k) const {
    const CA qv = cview_A(g, q), d1v = cview_A(g, dp1);
    if (it > ksplt[k]) {
      view_A(g, q_out)(i, j, k) = qv(i, j, k);
      if (last && it != nsplt) view_A(g, dp1_out)(i, j, k) = d1v(i, j, k);
      return;
    }
    const CA fxv = cview_FX(g, fx), fyv = cview_FY(g, fy), mx = cview_FX(g, mfx), my = cview_FY(g, mfy);
    const double ra = g.rarea[g.iA(i, j)], d1 = d1v(i, j, k);
    const double dp2 = d1 + (mx(i, j, k) - mx(i + 1, j, k) + my(i, j, k) - my(i, j + 1, k)) * ra;
    view_A(g, q_out)(i, j, k) = (qv(i, j, k) * d1 + (fxv(i, j, k) - fxv(i + 1, j, k) + fyv(i, j, k) - fyv(i, j + 1, k)) * ra) / dp2;
    if (last && it != nsplt) view_A(g, dp1_out)(i, j, k) = dp2;
  }
};

}  // namespace fv3
