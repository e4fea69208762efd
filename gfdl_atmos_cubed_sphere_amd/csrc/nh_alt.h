// nh_alt.h -- the other vertical solvers Riem_Solver3 / Riem_Solver_c dispatch on a_imp (model/nh_core.F90:169-185,
// model/nh_utils.F90:449-459):
//   a_imp < -0.999 (C grid: < -0.01)   SIM3p0_solver   nh_utils.F90:1134-1274
//   a_imp < -0.5                       SIM3_solver     nh_utils.F90:984-1132   (alpha = |a_imp|)
//   a_imp <= 0.5                       RIM_2D          nh_utils.F90:751-982    (ms = flagstruct%m_split sub-steps)
// SIM1 (a_imp > 0.999) and SIM are nh_kernels.h / nh_fast.h.  These three are off in every BASELINE configuration and rarely run:
// one thread per column with the column's work arrays in private memory, the statements in the reference's order -- correctness
// against the oracle is the bar here, not speed.  km <= kAltKm.
#pragma once

#include "nh_kernels.h"

namespace fv3 {

constexpr int kAltKm = 128;

// SIM3_solver (p0 = false) / SIM3p0_solver (p0 = true) of one column; arrays 1-based, pe2 / pem: km + 1 entries
FV3_HD void sim3_col(int km, double dt, double rgas, double gama, double kappa, double *pe2, const double *dm, const double *pem, double *w2,
                     double *dz2, const double *pt2, double ws, double alpha, double p_fac, double scale_m, double grav, bool p0) {
  constexpr double r3 = 1. / 3.;
  double aa[kAltKm + 2], bb[kAltKm + 2], dd[kAltKm + 2], w1[kAltKm + 2], wk[kAltKm + 2], g_rat[kAltKm + 2], gam[kAltKm + 2], pp[kAltKm + 3];
  const double beta = 1. - alpha, ra = 1. / alpha, t2 = beta / alpha;
  const double t1g = p0 ? 2. * gama * (dt * dt) : gama * 2. * ((alpha * dt) * (alpha * dt));
  const double rdt = 1. / dt, capa1 = kappa - 1., r2g = grav / 2., r6g = grav / 6.;
  for (int k = 1; k <= km; k++) {
    w1[k] = w2[k];
    wk[k] = 0.;
    aa[k] = dexp(gama * dlog(-dm[k] / dz2[k] * rgas * pt2[k]));      // full pressure at the centre
  }
  for (int k = 1; k <= km - 1; k++) {
    g_rat[k] = dm[k] / dm[k + 1];
    bb[k] = 2. * (1. + g_rat[k]);
    dd[k] = 3. * (aa[k] + g_rat[k] * aa[k + 1]);
  }
  double bet = bb[1];
  pe2[1] = pem[1];
  pe2[2] = (dd[1] - pem[1]) / bet;
  bb[km] = 2.;
  dd[km] = 3. * aa[km] + r2g * dm[km];
  for (int k = 2; k <= km; k++) {
    gam[k] = g_rat[k - 1] / bet;
    bet = bb[k] - gam[k];
    pe2[k + 1] = (dd[k] - pe2[k]) / bet;
  }
  for (int k = km; k >= 2; k--) pe2[k] = pe2[k] - gam[k] * pe2[k + 1];
  for (int k = 1; k <= km + 1; k++) pp[k] = pe2[k] - pem[k];        // perturbation pressure at the interfaces
  for (int k = 2; k <= km; k++) {
    if (p0) {
      aa[k] = t1g / (dz2[k - 1] + dz2[k]) * pe2[k] - scale_m * dm[1];
    } else {
      aa[k] = t1g / (dz2[k - 1] + dz2[k]) * pe2[k];
      wk[k] = t2 * aa[k] * (w1[k - 1] - w1[k]);
      aa[k] = aa[k] - scale_m * dm[1];
    }
  }
  bet = dm[1] - aa[2];
  w2[1] = p0 ? (dm[1] * w1[1] + dt * pp[2]) / bet : (dm[1] * w1[1] + dt * pp[2] + wk[2]) / bet;
  for (int k = 2; k <= km - 1; k++) {
    gam[k] = aa[k] / bet;
    bet = dm[k] - (aa[k] + aa[k + 1] + aa[k] * gam[k]);
    if (p0)
      w2[k] = (dm[k] * w1[k] + dt * (pp[k + 1] - pp[k]) - aa[k] * w2[k - 1]) / bet;
    else
      w2[k] = (dm[k] * w1[k] + dt * (pp[k + 1] - pp[k]) + wk[k + 1] - wk[k] - aa[k] * w2[k - 1]) / bet;
  }
  const double wk1 = t1g / dz2[km] * pe2[km + 1];
  gam[km] = aa[km] / bet;
  bet = dm[km] - (aa[km] + wk1 + aa[km] * gam[km]);
  if (p0)
    w2[km] = (dm[km] * w1[km] + dt * (pp[km + 1] - pp[km]) - wk1 * ws - aa[km] * w2[km - 1]) / bet;
  else
    w2[km] = (dm[km] * w1[km] + dt * (pp[km + 1] - pp[km]) - wk[km] + wk1 * (t2 * w1[km] - ra * ws) - aa[km] * w2[km - 1]) / bet;
  for (int k = km - 1; k >= 1; k--) w2[k] = w2[k] - gam[k + 1] * w2[k + 1];
  pe2[1] = 0.;
  for (int k = 1; k <= km; k++) {
    if (p0)
      pe2[k + 1] = pe2[k] + dm[k] * (w2[k] - w1[k]) * rdt;
    else
      pe2[k + 1] = pe2[k] + (dm[k] * (w2[k] - w1[k]) * rdt - beta * (pp[k + 1] - pp[k])) * ra;
  }
  pe2[1] = pem[1];
  for (int k = 2; k <= km + 1; k++) pe2[k] = dmax(p_fac * pem[k], pe2[k] + pem[k]);   // full nonhydrostatic pressure
  double p1 = (pe2[km] + 2. * pe2[km + 1]) * r3 - r6g * dm[km];
  dz2[km] = -dm[km] * rgas * pt2[km] * dexp(capa1 * dlog(p1));
  for (int k = km - 1; k >= 1; k--) {
    p1 = (pe2[k] + bb[k] * pe2[k + 1] + g_rat[k] * pe2[k + 2]) * r3 - g_rat[k] * p1;
    dz2[k] = -dm[k] * rgas * pt2[k] * dexp(capa1 * dlog(p1));
  }
  for (int k = 1; k <= km + 1; k++) {
    pe2[k] = pe2[k] - pem[k];
    if (!p0) pe2[k] = pe2[k] + beta * (pp[k] - pe2[k]);
  }
}

// RIM_2D of one column: the Riemann invariants w dm +- dts (p - pm) of every layer carried along the characteristics for ms sub-steps;
// what an interface collects from above (m_top, r_top) and from below, reflected at the surface (m_bot, r_bot), gives its velocity
// and pressure
FV3_HD void rim2d_col(int ms, double bdt, int km, double rgas, double gama, const double *gm2, double *pe2, const double *dm2,
                      const double *pm2, double *w2, double *dz2, const double *pt2, double ws, bool c_core) {
  double m_bot[kAltKm + 3], m_top[kAltKm + 3], r_bot[kAltKm + 3], r_top[kAltKm + 3], pe1[kAltKm + 3], pbar[kAltKm + 3], wbar[kAltKm + 3];
  double r_hi[kAltKm + 2], r_lo[kAltKm + 2], dz[kAltKm + 2], wm[kAltKm + 2], dts[kAltKm + 2], pf1[kAltKm + 2], wc[kAltKm + 2],
      cm[kAltKm + 2], pp[kAltKm + 2];
  const double grg = gama * rgas, rdt = 1. / bdt, dt = bdt / (double)ms, ws2 = 2. * ws;
  for (int k = 0; k <= km + 2; k++) m_bot[k] = m_top[k] = r_bot[k] = r_top[k] = pe1[k] = pbar[k] = wbar[k] = 0.;
  for (int k = 1; k <= km; k++) {
    dz[k] = dz2[k];
    wm[k] = w2[k] * dm2[k];
  }
  wbar[km + 1] = ws;
  int ks0 = 1;
  if (ms > 1 && ms < 8) {     // the layers from the top whose sound-crossing time exceeds bdt: one step (:795-851)
    ks0 = km;
    for (int k = 1; k <= km; k++) {
      const double rden = -rgas * dm2[k] / dz[k];
      pf1[k] = dexp(gm2[k] * dlog(rden * pt2[k]));
      dts[k] = -dz[k] / sqrt(grg * pf1[k] / rden);
      if (bdt > dts[k]) {
        ks0 = k - 1;
        break;
      }
    }
    if (ks0 < 1) ks0 = 1;     // (undefined in the reference -- unset locals, pbar(0): see oracle/nh_core.c)
    if (ks0 != 1) {
      for (int k = 1; k <= ks0; k++) {
        cm[k] = dm2[k] / dts[k];
        wc[k] = wm[k] / dts[k];
        pp[k] = pf1[k] - pm2[k];
      }
      wbar[1] = (wc[1] + pp[1]) / cm[1];
      for (int k = 2; k <= ks0; k++) {
        wbar[k] = (wc[k - 1] + wc[k] + pp[k] - pp[k - 1]) / (cm[k - 1] + cm[k]);
        pbar[k] = bdt * (cm[k - 1] * wbar[k] - wc[k - 1] + pp[k - 1]);
        pe1[k] = pbar[k];
      }
      if (ks0 == km) {
        pbar[km + 1] = bdt * (cm[km] * wbar[km + 1] - wc[km] + pp[km]);
        for (int k = 1; k <= km; k++) {
          dz2[k] = dz[k] + bdt * (wbar[k + 1] - wbar[k]);
          if (!c_core) w2[k] = (wm[k] + pbar[k + 1] - pbar[k]) / dm2[k];
        }
        pe2[1] = 0.;
        for (int k = 2; k <= km + 1; k++) pe2[k] = pbar[k] * rdt;
        return;
      }
      for (int k = 1; k <= ks0 - 1; k++) {
        dz2[k] = dz[k] + bdt * (wbar[k + 1] - wbar[k]);
        if (!c_core) w2[k] = (wm[k] + pbar[k + 1] - pbar[k]) / dm2[k];
      }
      pbar[ks0] = pbar[ks0] / (double)ms;
    }
  }
  const int ks1 = ks0;
  for (int n = 1; n <= ms; n++) {
    for (int k = ks1; k <= km; k++) {
      const double rden = -rgas * dm2[k] / dz[k];
      const double pf = dexp(gm2[k] * dlog(rden * pt2[k]));
      dts[k] = -dz[k] / sqrt(grg * pf / rden);
      const double ptmp1 = dts[k] * (pf - pm2[k]);
      r_lo[k] = wm[k] + ptmp1;
      r_hi[k] = wm[k] - ptmp1;
    }
    int ktop = km;
    for (int k = ks1; k <= km; k++)
      if (dt > dts[k]) {
        ktop = k - 1;
        break;
      }
    for (int k = ks1; k <= ktop; k++) {       // the sub-step stays inside the layer: both interfaces take their share directly
      const double z_frac = dt / dts[k];
      r_bot[k] = z_frac * r_lo[k];
      r_top[k + 1] = z_frac * r_hi[k];
      m_bot[k] = z_frac * dm2[k];
      m_top[k + 1] = m_bot[k];
    }
    if (!(ktop >= ks1 && ktop == km)) {
      for (int k = ktop + 2; k <= km + 1; k++) {
        m_top[k] = 0.;
        r_top[k] = 0.;
      }
      const int kt1 = ktop > 1 ? ktop : 1;
      for (int ke = km + 1; ke >= ktop + 2; ke--) {         // what reaches interface ke from above within dt
        double time_left = dt;
        for (int k = ke - 1; k >= kt1; k--) {
          if (time_left > dts[k]) {
            time_left = time_left - dts[k];
            m_top[ke] = m_top[ke] + dm2[k];
            r_top[ke] = r_top[ke] + r_hi[k];
          } else {
            const double z_frac = time_left / dts[k];
            m_top[ke] = m_top[ke] + z_frac * dm2[k];
            r_top[ke] = r_top[ke] + z_frac * r_hi[k];
            break;
          }
        }
      }
      for (int k = ktop + 1; k <= km; k++) {
        m_bot[k] = 0.;
        r_bot[k] = 0.;
      }
      for (int ke = ktop + 1; ke <= km; ke++) {             // ... from below, and reflected at the surface
        double time_left = dt;
        bool found = false;
        for (int k = ke; k <= km; k++) {
          if (time_left > dts[k]) {
            time_left = time_left - dts[k];
            m_bot[ke] = m_bot[ke] + dm2[k];
            r_bot[ke] = r_bot[ke] + r_lo[k];
          } else {
            const double z_frac = time_left / dts[k];
            m_bot[ke] = m_bot[ke] + z_frac * dm2[k];
            r_bot[ke] = r_bot[ke] + z_frac * r_lo[k];
            found = true;
            break;
          }
        }
        if (found) continue;
        const double m_surf = m_bot[ke];
        for (int k = km; k >= kt1; k--) {
          if (time_left > dts[k]) {
            time_left = time_left - dts[k];
            m_bot[ke] = m_bot[ke] + dm2[k];
            r_bot[ke] = r_bot[ke] - r_hi[k];
          } else {
            const double z_frac = time_left / dts[k];
            m_bot[ke] = m_bot[ke] + z_frac * dm2[k];
            r_bot[ke] = r_bot[ke] - z_frac * r_hi[k] + (m_bot[ke] - m_surf) * ws2;
            break;
          }
        }
      }
    }
    if (ks1 == 1) wbar[1] = r_bot[1] / m_bot[1];
    for (int k = ks1 + 1; k <= km; k++) wbar[k] = (r_bot[k] + r_top[k]) / (m_top[k] + m_bot[k]);
    for (int k = ks1 + 1; k <= km + 1; k++) {     // pbar is dt * pbar
      pbar[k] = m_top[k] * wbar[k] - r_top[k];
      pe1[k] = pe1[k] + pbar[k];
    }
    if (n == ms) {
      for (int k = ks1; k <= km; k++) {
        dz2[k] = dz[k] + dt * (wbar[k + 1] - wbar[k]);
        if (!c_core) w2[k] = (wm[k] + pbar[k + 1] - pbar[k]) / dm2[k];
      }
    } else {
      for (int k = ks1; k <= km; k++) {
        dz[k] = dz[k] + dt * (wbar[k + 1] - wbar[k]);
        wm[k] = wm[k] + pbar[k + 1] - pbar[k];
      }
    }
  }
  pe2[1] = 0.;
  for (int k = 2; k <= km + 1; k++) pe2[k] = pe1[k] * rdt;
}

// Riem_Solver3 (CG = false) / Riem_Solver_c (CG = true) around those solvers: one thread per column.  mode: 0 SIM3p0, 1 SIM3, 2 RIM_2D
template <bool CG>
struct RiemSolverAlt {
  Grid g;
  int km, mode, m_split;
  double dt;
  NhConsts cn;
  const double *zs, *pt, *delp, *ws;     // zs = hs on the C grid
  double *wq, *zl;                       // w / w3 (C grid: read only), zh / gz
  double *delz, *ppe, *pk3, *pe, *pk, *peln, *pef;
  int use_logp, last_call, fp_out;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int wdt = CG ? g.nx + 2 : g.nx, ncol = wdt * (CG ? g.ny + 2 : g.ny);
    const size_t nA = g.nA(), nCC = g.nCC();
    const double rgrav = 1. / cn.grav, gama = 1. / (1. - cn.akap);
    const double peln1 = dlog(cn.ptop), ptk = dexp(cn.akap * peln1);
    FV3_COL_FOR(c, ncol) {
      const int i = (CG ? g.is - 1 : g.is) + c % wdt, j = (CG ? g.js - 1 : g.js) + c / wdt;
      const int o = g.iA(i, j), occ = CG ? 0 : g.iCC(i, j);
      double dm[kAltKm + 2], dz2[kAltKm + 2], w2[kAltKm + 2], pm2[kAltKm + 2], gm2[kAltKm + 2], pem[kAltKm + 3], pe2[kAltKm + 3],
          pt2[kAltKm + 2], peln2[kAltKm + 3];
      pem[1] = cn.ptop;
      peln2[1] = peln1;
      for (int k = 2; k <= km + 1; k++) {
        pem[k] = pem[k - 1] + delp[(size_t)(k - 2) * nA + o];
        if (!CG) peln2[k] = dlog(pem[k]);
      }
      for (int k = 1; k <= km; k++) {
        const double d = delp[(size_t)(k - 1) * nA + o];
        pm2[k] = CG ? d / dlog(pem[k + 1] / pem[k]) : d / (peln2[k + 1] - peln2[k]);
        gm2[k] = gama;
        dm[k] = d * rgrav;
        dz2[k] = zl[(size_t)k * nA + o] - zl[(size_t)(k - 1) * nA + o];
        w2[k] = wq[(size_t)(k - 1) * nA + o];
        pt2[k] = pt[(size_t)(k - 1) * nA + o];
      }
      const double wsv = ws[CG ? o : occ];
      if (mode == 0)
        sim3_col(km, dt, cn.rdgas, gama, cn.akap, pe2, dm, pem, w2, dz2, pt2, wsv, 1.0, cn.p_fac, 0.0, cn.grav, true);
      else if (mode == 1)
        sim3_col(km, dt, cn.rdgas, gama, cn.akap, pe2, dm, pem, w2, dz2, pt2, wsv, fabs(cn.a_imp), cn.p_fac, 0.0, cn.grav, false);
      else
        rim2d_col(m_split, dt, km, cn.rdgas, gama, gm2, pe2, dm, pm2, w2, dz2, pt2, wsv, CG);
      if (CG) {           // pef = pe2 + pem (:461-465); gz = hs - sum dz2 grav (:468-476)
        pef[o] = cn.ptop;
        for (int k = 2; k <= km + 1; k++) pef[(size_t)(k - 1) * nA + o] = pe2[k] + pem[k];
        double zb = zs[o];
        zl[(size_t)km * nA + o] = zb;
        for (int k = km; k >= 1; k--) {
          zb = zb - dz2[k] * cn.grav;
          zl[(size_t)(k - 1) * nA + o] = zb;
        }
        continue;
      }
      for (int k = 1; k <= km; k++) {
        wq[(size_t)(k - 1) * nA + o] = w2[k];
        delz[(size_t)(k - 1) * nCC + occ] = dz2[k];
      }
      pk3[o] = ptk;
      for (int k = 1; k <= km + 1; k++) {
        const double pkv = k == 1 ? ptk : dexp(cn.akap * peln2[k]);
        if (k > 1) pk3[(size_t)(k - 1) * nA + o] = use_logp ? peln2[k] : pkv;
        if (last_call) {
          peln[(size_t)(j - g.js) * g.nx * (km + 1) + (size_t)(k - 1) * g.nx + (i - g.is)] = peln2[k];
          pk[(size_t)(k - 1) * nCC + occ] = pkv;
          pe[(size_t)(j - (g.js - 1)) * (g.nx + 2) * (km + 1) + (size_t)(k - 1) * (g.nx + 2) + (i - (g.is - 1))] = pem[k];
        }
        ppe[(size_t)(k - 1) * nA + o] = fp_out ? pe2[k] + pem[k] : pe2[k];
      }
      double zb = zs[o];
      zl[(size_t)km * nA + o] = zb;
      for (int k = km; k >= 1; k--) {
        zb = zb - dz2[k];
        zl[(size_t)(k - 1) * nA + o] = zb;
      }
    }
  }
};

}  // namespace fv3
