// ppm.h -- 1-D PPM face values, one face per call (device functions).
//
// Reference: model/tp_core.F90 xppm :324-712 / yppm :715-1152 (scalar transport) and
// model/sw_core.F90 xtp_u :2154-2521 / ytp_v :2524-2998 (wind transport for the KE flux), in the
// branches taken for grid_type >= 3 (no cubed-sphere edge overrides).
//
// The reference builds al/bl/br for a whole row and then picks the upwind cell per face.  Here a
// thread evaluates one face directly: for the monotone schemes (hord >= 8) only the upwind
// cell's (bl, br) matter, so a face costs one 5-point stencil read centred on the upwind cell;
// for the unlimited schemes (hord 5, 6, -5) the smoothness flags of both neighbours are needed
// (6-point read).  Every expression is evaluated in the reference's order, so with FMA
// contraction disabled the result is bit-identical to the row-wise formulation.
#pragma once

#include "fv3_common.h"

namespace fv3 {

inline bool tp_ord_supported(int iord) { return iord == 5 || iord == -5 || iord == 6 || iord == 8 || iord == 10; }
// fv_tp_2d as a unit and tracer_2d also take the positive-definite / van Leer members of the monotone family:
// 9 == 13 (unlimited + pert_ppm), 11 (ppm_fac slopes), 12 (Lin & Rood 1996 positive definite) -- tp_core.F90:604-641
inline bool tp_ord_supported_tr(int iord) { return tp_ord_supported(iord) || iord == 7 || iord == 9 || iord == 11 || iord == 12 || iord == 13; }
inline bool sw_ord_supported(int iord) { return iord >= 5 && iord <= 11; }

// monotone slope, tp_core.F90:570-574 == sw_core.F90:2383-2387.  s[0] is the cell.
FV3_HD double ppm_dm(double qm, double q0, double qp) {
  const double xt = 0.25 * (qp - qm);
  return fsign(dmin3(fabs(xt), dmax3(qm, q0, qp) - q0, q0 - dmin3(qm, q0, qp)), xt);
}

// ---- tp_core flavour ---------------------------------------------------------------------
// s points at cell i of a line with element stride st; the face is the one between i-1 and i.
// c is the Courant number at that face.  Reads s[-3*st .. 2*st].
FV3_HD double ppm_face_tp(const double *s, int st, double c, int iord, double lim_fac) {
  (void)lim_fac;
  constexpr double r3 = 1. / 3., near_zero = 1.E-25, r12 = 1. / 12., p1 = 7. / 12., p2 = -1. / 12.;
  if (iord == 7) {  // the monotone family's edge values, the positive-definite cell of :611-633, the flux form of :685-699
    auto cell = [&](const double *u, double &bl, double &br) {
      const double qm2 = u[-2 * st], qm1 = u[-st], q0 = u[0], qp1 = u[st], qp2 = u[2 * st];
      const double dmm = ppm_dm(qm2, qm1, q0), dm0 = ppm_dm(qm1, q0, qp1), dmp = ppm_dm(q0, qp1, qp2);
      bl = (0.5 * (qm1 + q0) + r3 * (dmm - dm0)) - q0;
      br = (0.5 * (q0 + qp1) + r3 * (dm0 - dmp)) - q0;
      const double a4 = -3. * (bl + br), da1 = br - bl;
      if (fabs(da1) < -a4 && q0 + 0.25 / a4 * (da1 * da1) + a4 * r12 < 0.) {
        if (br * bl > 0.) {
          br = 0.;
          bl = 0.;
        } else if (da1 > 0.) {
          br = -2. * bl;
        } else {
          bl = -2. * br;
        }
      }
    };
    double blm, brm, bl0, br0;
    cell(s - st, blm, brm);
    cell(s, bl0, br0);
    const bool sm = blm * brm < 0., s0 = bl0 * br0 < 0.;
    double fx1, flux;
    if (c > 0.) {
      fx1 = (1. - c) * (brm - c * (blm + brm));
      flux = s[-st];
    } else {
      fx1 = (1. + c) * (bl0 + c * (bl0 + br0));
      flux = s[0];
    }
    if (sm || s0) flux = flux + fx1;
    return flux;
  }
  if (iord >= 8) {
    // upwind cell
    const double *u = (c > 0.) ? s - st : s;
    const double qm2 = u[-2 * st], qm1 = u[-st], q0 = u[0], qp1 = u[st], qp2 = u[2 * st];
    const double dmm = ppm_dm(qm2, qm1, q0), dm0 = ppm_dm(qm1, q0, qp1), dmp = ppm_dm(q0, qp1, qp2);
    const double al0 = 0.5 * (qm1 + q0) + r3 * (dmm - dm0);   // al(i)   tp_core.F90:576
    const double al1 = 0.5 * (q0 + qp1) + r3 * (dm0 - dmp);   // al(i+1)
    double bl, br;
    if (iord == 8) {  // :579-584
      const double xt = 2. * dm0;
      bl = -fsign(dmin(fabs(xt), fabs(al0 - q0)), xt);
      br = fsign(dmin(fabs(xt), fabs(al1 - q0)), xt);
    } else if (iord == 11) {  // :604-610, ppm_fac = 1.5
      const double xt = 1.5 * dm0;
      bl = -fsign(dmin(fabs(xt), fabs(al0 - q0)), xt);
      br = fsign(dmin(fabs(xt), fabs(al1 - q0)), xt);
    } else if (iord == 12 || iord == 9 || iord == 13) {  // :611-633 / :634-641 with pert_ppm(iv = 0), :1219-1242
      bl = al0 - q0;
      br = al1 - q0;
      const bool pert = iord != 12;
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
    } else {  // iord == 10, :585-603
      bl = al0 - q0;
      br = al1 - q0;
      if (fabs(dmm) + fabs(dm0) + fabs(dmp) < near_zero) {
        bl = 0.;
        br = 0.;
      } else if (fabs(3. * (bl + br)) > fabs(bl - br)) {
        const double pmp_2 = 2. * (q0 - qm1);                 // dq(i-1)
        const double lac_2 = pmp_2 - 0.75 * (2. * (qm1 - qm2)); // dq(i-2)
        br = dmin(dmax3(0., pmp_2, lac_2), dmax(br, dmin3(0., pmp_2, lac_2)));
        const double pmp_1 = -(2. * (qp1 - q0));              // -dq(i)
        const double lac_1 = pmp_1 + 0.75 * (2. * (qp2 - qp1)); // dq(i+1)
        bl = dmin(dmax3(0., pmp_1, lac_1), dmax(bl, dmin3(0., pmp_1, lac_1)));
      }
    }
    // :701-707
    if (c > 0.) return q0 + (1. - c) * (br - c * (bl + br));
    return q0 + (1. + c) * (bl + c * (bl + br));
  }
  // unlimited family iord = 5, -5, 6 (:365-560)
  const double qm3 = s[-3 * st], qm2 = s[-2 * st], qm1 = s[-st], q0 = s[0], qp1 = s[st], qp2 = s[2 * st];
  double alm = p1 * (qm2 + qm1) + p2 * (qm3 + q0);   // al(i-1)
  double al0 = p1 * (qm1 + q0) + p2 * (qm2 + qp1);   // al(i)
  double alp = p1 * (q0 + qp1) + p2 * (qm1 + qp2);   // al(i+1)
  if (iord < 0) {
    alm = dmax(0., alm);
    al0 = dmax(0., al0);
    alp = dmax(0., alp);
  }
  // cell i-1 and cell i
  const double blm = alm - qm1, brm = al0 - qm1, b0m = blm + brm;
  const double bl0 = al0 - q0, br0 = alp - q0, b00 = bl0 + br0;
  bool sm, s0;
  if (iord == 6) {
    sm = 3. * fabs(b0m) < fabs(blm - brm);
    s0 = 3. * fabs(b00) < fabs(bl0 - br0);
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
  if (iord == -5) {  // positive-definite adjustment of the upwind cell, :499-524
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
  double fx1, flux;  // :549-558
  if (c > 0.) {
    fx1 = (1. - c) * (br - c * b0);
    flux = qu;
  } else {
    fx1 = (1. + c) * (bl + c * b0);
    flux = qu;
  }
  if (sm || s0) flux = flux + fx1;
  return flux;
}

// ---- sw_core flavour (xtp_u / ytp_v) -----------------------------------------------------------
// s points at wind value i (u(i,j) for xtp_u, v(i,j) for ytp_v) with stride st along the sweep.
// c is the advective displacement at the corner; rdm/rd0 are 1/dx (1/dy) of cell i-1 and cell i.
FV3_HD double ppm_face_sw(const double *s, int st, double c, double rdm, double rd0, int iord) {
  constexpr double r3 = 1. / 3., p1 = 7. / 12., p2 = -1. / 12.;
  if (iord >= 8) {  // "Other grids" branch, sw_core.F90:2492-2516 / :2973-2996
    const double *u = (c > 0.) ? s - st : s;
    const double qm2 = u[-2 * st], qm1 = u[-st], q0 = u[0], qp1 = u[st], qp2 = u[2 * st];
    const double dmm = ppm_dm(qm2, qm1, q0), dm0 = ppm_dm(qm1, q0, qp1), dmp = ppm_dm(q0, qp1, qp2);
    const double al0 = 0.5 * (qm1 + q0) + r3 * (dmm - dm0);
    const double al1 = 0.5 * (q0 + qp1) + r3 * (dm0 - dmp);
    double pmp = -2. * (qp1 - q0);         // -2*dq(i)
    double lac = pmp + 1.5 * (qp2 - qp1);  // + 1.5*dq(i+1)
    const double bl = dmin(dmax3(0., pmp, lac), dmax(al0 - q0, dmin3(0., pmp, lac)));
    pmp = 2. * (q0 - qm1);                 // 2*dq(i-1)
    lac = pmp - 1.5 * (qm1 - qm2);         // - 1.5*dq(i-2)
    const double br = dmin(dmax3(0., pmp, lac), dmax(al1 - q0, dmin3(0., pmp, lac)));
    if (c > 0.) {
      const double cfl = c * rdm;
      return q0 + (1. - cfl) * (br - cfl * (bl + br));
    }
    const double cfl = c * rd0;
    return q0 + (1. + cfl) * (bl + cfl * (bl + br));
  }
  // iord = 5, 6, 7 (sw_core.F90:2190-2243, 2337-2374)
  const double qm3 = s[-3 * st], qm2 = s[-2 * st], qm1 = s[-st], q0 = s[0], qp1 = s[st], qp2 = s[2 * st];
  const double alm = p1 * (qm2 + qm1) + p2 * (qm3 + q0);
  const double al0 = p1 * (qm1 + q0) + p2 * (qm2 + qp1);
  const double alp = p1 * (q0 + qp1) + p2 * (qm1 + qp2);
  const double blm = alm - qm1, brm = al0 - qm1, b0m = blm + brm;
  const double bl0 = al0 - q0, br0 = alp - q0, b00 = bl0 + br0;
  bool sm, s0;
  if (iord == 5) {
    sm = blm * brm < 0.;
    s0 = bl0 * br0 < 0.;
  } else {
    sm = 3. * fabs(b0m) < fabs(blm - brm);
    s0 = 3. * fabs(b00) < fabs(bl0 - br0);
  }
  double fx0, flux;
  if (c > 0.) {
    const double cfl = c * rdm;
    fx0 = (1. - cfl) * (brm - cfl * b0m);
    flux = qm1;
  } else {
    const double cfl = c * rd0;
    fx0 = (1. + cfl) * (bl0 + cfl * b00);
    flux = q0;
  }
  if (sm || s0) flux = flux + fx0;
  return flux;
}

}  // namespace fv3
