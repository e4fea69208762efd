// csw_march.h -- c_sw (model/sw_core.F90:79-488) as a wave-marching stencil (grid_type >= 3 branches).
//
// One wavefront owns 58 output columns (lanes 3..60 of a 64-column strip) of the box
// [is-1, ie+2] x [js-1, je+2] and marches along j.  Step t loads u(:, t), v(:, t-1), delp/pt/w(:, t-2) and
// finishes row R = t-2 of the interpolated winds and row Q = t-3 of every output:
//
//   utmp(R), vtmp(t-1)         D -> A interpolation               d2a2c_vect :3099-3108
//   ua, va (R)                 contravariant A-grid winds          :3152-3157
//   uc(R) (x shifts), vc(R)    A -> C interpolation                :3197-3202, :3337-3342
//   ut(R), vt(R)               time-scaled area fluxes             :159-176
//   vort(R)                    absolute vorticity at the corners   :372-403
//   ke(Q)                      upstream kinetic energy             :297-366
//   delpc, ptc, wc (Q)         first-order upwind transport        :182-286
//   uc, vc (Q)                 C-grid wind half-step update        :414-486
//   divg_d(Q)                  divergence_corner                   :1781-1796
//
// x-neighbours come from DPP wavefront shifts, y-neighbours from small register windows; no LDS.
#pragma once

#include "csw_kernel.h"
#include "tp2d_march.h"

namespace fv3 {

// owned columns per strip: lanes 3..60.  Three lanes are needed on either side: ke(L-1) reaches utmp(L-3) on the left;
// on the right the vc update at lane L takes vort(L+1) when the wind blows from there (:452-486), vort(L+1) needs
// vc(L+1), and the 4-point interpolation vtmp(L+1) reads v(L+3).
constexpr int kCswCols = 58;
constexpr int kCswLast = 60;  // last owned lane

inline MarchDims make_csw_dims(const Grid &g, int tj) {
  MarchDims d;
  d.tj = tj;
  d.klist = nullptr;
  d.nk = 0;
  d.k_fast = march_k_fast();
  d.nstrips = (g.nx + 4 + kCswCols - 1) / kCswCols;
  d.nsegs = (g.ny + 4 + tj - 1) / tj;
  d.alt_nk = d.alt_ng = d.alt_tj = 0;
  d.set_box(0, d.nstrips, 0, d.nsegs);
  return d;
}

// metric rows one step reads (shared by the KPW levels a wavefront carries)
struct CswMetrics {
  vd cs, rs, cosau, rsinu, dy, sg3, sg1, dx, sg4, sg2, dxc, dyc, rac, fc;  // row R
  vd ra, sinau, rdxc, cosav, sinav, rdyc;                                    // row Q
};
struct CswFields {
  vd u, v, dp, pt, w;
};

// register state of one level
struct CswLevel {
  vd u0, u1, u2, u3;          // u(t-3 .. t)
  vd v0, v1, v2, v3;          // v(t-4 .. t-1)
  vd vt0, vt1, vt2, vt3;      // vtmp(t-4 .. t-1)
  vd dp0, dpp, pt0, ptp, w0, wp;  // rows Q, Q+1
  vd ua_p, va_p, uc_p, vc_p, ut_p, vt_p;  // row Q = t-3 (previous step's row R)
  vd ucdx_p, vort_p, ke_p, vdxc_p;
  vd fy1_p, fyp_p, fyw_p;     // upwind fluxes through y-face Q (delp, pt, w)
  vd va_pp, vf_p;             // cubed-sphere divergence: va of row Q-1, vf of corner row Q-1
  FV3_D void init() {
    u0 = u1 = u2 = u3 = v0 = v1 = v2 = v3 = vt0 = vt1 = vt2 = vt3 = vd(0.);
    dp0 = dpp = pt0 = ptp = w0 = wp = vd(0.);
    ua_p = va_p = uc_p = vc_p = ut_p = vt_p = ucdx_p = vort_p = ke_p = vdxc_p = vd(0.);
    fy1_p = fyp_p = fyw_p = vd(0.);
    va_pp = vf_p = vd(0.);
  }
};

// KPW = levels per wavefront: the ~20 metric rows of a step are loaded once and used for KPW levels, which
// cuts the L2 traffic per cell (the kernel is bound by it, not by HBM or VALU) at the price of registers.
// GM = geometry mode of the gridstruct (Grid::geom): 0 = every metric row is read; 1 = orthogonal grid (the angle
// terms cosa_* = 0, sin* = rsin* = 1 are not read: x*1 and x - y*0 are exact); 2 = orthogonal and uniform (the
// length / area terms are wave-uniform scalars, only fC is read).
// CS: the interior of a cubed-sphere face (hybrid of fv3_api.hip csw_cubed: store mask, vt of d2a2c_vect's grid_type < 3 branch,
// the non-orthogonal divergence_corner) -- a compile-time switch: the doubly periodic instantiations keep their registers
template <int KPW, int GM = 0, bool CS = false>
struct CswMarch {
#ifdef FV3_CSW_TWO_WAVES
  static constexpr int kTwoWavesPerSimd = 1;
#endif
  Grid g;
  CswArgs a;
  MarchDims md;
  int nkg;  // level groups = ceil(npz / KPW)

  FV3_D void operator()(int gid) const {
    constexpr double a1 = 0.5625, a2 = -0.0625;  // sw_core.F90:53-54
    int strip, seg, kg, tjw;
    md.decode_tj(gid, strip, seg, kg, tjw);
    const int is = g.is, ie = g.ie, js = g.js, je = g.je;
    const int ilo = is - 1 + strip * kCswCols - 3;  // column of lane 0
    auto cl = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    const vl cA = make_lanes(cl(g.isd - ilo, 0, kW - 1), cl(g.ied - ilo, 0, kW - 1));      // nid-wide rows
    const vl cV = make_lanes(cl(g.isd - ilo, 0, kW - 1), cl(g.ied + 1 - ilo, 0, kW - 1));  // (nid+1)-wide rows
    int l2 = cl(ie + 2 - ilo, 0, kCswLast), l1 = cl(ie + 1 - ilo, 0, kCswLast);  // last owned lane: i <= ie+2 / ie+1
    // cubed-sphere hybrid: the lanes / rows of the outputs this kernel owns (CswArgs::mask_w)
    const int mw = CS ? a.mask_w : 0;
    const int l0 = mw ? (3 > mw + 1 - ilo ? 3 : mw + 1 - ilo) : 3;
    if (mw) {
      const int lm = g.npx - mw - 1 - ilo;
      l2 = l2 < lm ? l2 : lm;
      l1 = l1 < lm ? l1 : lm;
    }
    const int oJ0 = mw ? mw + 1 : g.jsd - 1, oJ1 = mw ? g.npy - mw - 1 : g.jed + 2;
    const int jA = js - 1 + seg * tjw;
    const int jB = (jA + tjw - 1 < je + 2) ? jA + tjw - 1 : je + 2;
    const bool nh = !a.hydrostatic;
    const double dt2 = a.dt2, dt4 = 0.5 * dt2;
    const long nAp = (long)g.nA();  // one sin_sg plane
    // row loaders (row index clamped into the array; clamped rows are never used for a kept value)
    auto LA = [&](const double *p, int j, long shift = 0) { return vload(p, (long)g.iA(ilo, cl(j, g.jsd, g.jed)) + shift, cA); };
    auto LU = [&](const double *p, int j) { return vload(p, (long)g.iU(ilo, cl(j, g.jsd, g.jed + 1)), cA); };
    auto LV = [&](const double *p, int j) { return vload(p, (long)g.iV(ilo, cl(j, g.jsd, g.jed)), cV); };
    auto LB = [&](const double *p, int j) { return vload(p, (long)g.iB(ilo, cl(j, g.jsd, g.jed + 1)), cV); };
    const vb m_uc = lane_mask(is - ilo, ie + 1 - ilo);  // columns where uc is advanced (:414-447)
    const vb m_vc = lane_mask(is - ilo, ie - ilo);      // columns where vc is advanced (:452-486)
    // the row step has no control flow (spmd.h "branch-free rows"): the row conditions go into the stores, the lane masks into their offsets
    const vm s1 = make_mask(l0, l1), s2 = make_mask(l0, l2);

    // the levels of this wavefront (the last group may be short: the surplus slot repeats the last level and
    // does not store)
    int kl[KPW];
    bool live[KPW];
    for (int m = 0; m < KPW; m++) {
      const int k = kg * KPW + m;
      live[m] = k < g.npz;
      kl[m] = live[m] ? k : g.npz - 1;
    }
    auto load_metrics = [&](int t) {
      const int R = t - 2, Q = t - 3;
      CswMetrics in;
      if constexpr (GM == 0) {
        in.cs = LA(g.cosa_s, R);  in.rs = LA(g.rsin2, R);
        in.cosau = LV(g.cosa_u, R);  in.rsinu = LV(g.rsin_u, R);
        in.sg3 = LA(g.sin_sg + 2 * nAp, R, -1);  in.sg1 = LA(g.sin_sg, R);      // sin_sg(i-1,j,3), sin_sg(i,j,1)
        in.sg4 = LA(g.sin_sg + 3 * nAp, R - 1);  in.sg2 = LA(g.sin_sg + nAp, R);  // sin_sg(i,j-1,4), sin_sg(i,j,2)
        in.sinau = LV(g.sina_u, Q);  in.cosav = LU(g.cosa_v, Q);  in.sinav = LU(g.sina_v, Q);
      }
      if constexpr (GM <= 1) {
        in.dy = LV(g.dy, R);  in.dx = LU(g.dx, R);
        in.dxc = LV(g.dxc, R);  in.dyc = LU(g.dyc, R);  in.rac = LB(g.rarea_c, R);
        in.ra = LA(g.rarea, Q);  in.rdxc = LV(g.rdxc, Q);  in.rdyc = LU(g.rdyc, Q);
      } else {
        in.dy = vd(g.c_dy);  in.dx = vd(g.c_dx);
        in.dxc = vd(g.c_dxc);  in.dyc = vd(g.c_dyc);  in.rac = vd(g.c_rarea_c);
        in.ra = vd(g.c_rarea);  in.rdxc = vd(g.c_rdxc);  in.rdyc = vd(g.c_rdyc);
      }
      in.fc = LB(g.fC, R);
      return in;
    };
    auto load_fields = [&](int t, int k) {
      const int R = t - 2;
      CswFields f;
      f.u = LU(a.u + (size_t)k * g.nU(), t);
      f.v = LV(a.v + (size_t)k * g.nV(), t - 1);
      f.dp = LA(a.delp + (size_t)k * g.nA(), R);
      f.pt = LA(a.pt + (size_t)k * g.nA(), R);
      f.w = LA((nh ? a.w : a.pt) + (size_t)k * g.nA(), R);   // hydrostatic: any valid row (no branch in the row step; wc is not stored)
      return f;
    };

    CswLevel st[KPW];
    for (int m = 0; m < KPW; m++) st[m].init();
    // metric rows that are needed again one step later (row R becomes row Q)
    vd cosau_p(0.), dxc_p(0.), dyc_p(0.), rac_p(0.);
    vd sg3_p(0.), sg1_p(0.), sg4_p(0.), sg2_p(0.);   // cubed-sphere divergence: the sin_sg rows of row R, used one step later

    // everything step t reads from memory is loaded one step ahead (software pipelining)
    CswMetrics mnxt = load_metrics(jA - 2);
    CswFields fnxt[KPW];
    for (int m = 0; m < KPW; m++) fnxt[m] = load_fields(jA - 2, kl[m]);
    vdrain_loads();
    for (int t = jA - 2; t <= jB + 3; t++) {
      const int R = t - 2, Q = t - 3;
      const int tn = t < jB + 3 ? t + 1 : t;
      const CswMetrics in = mnxt;
      mnxt = load_metrics(tn);
      vd cosav_r(0.), rsinv_r(1.);  // cubed-sphere face: vt = (vc - u*cosa_v)*rsin_v (:3340), not vt = vc (:3350)
      vd csu(0.), csv(0.);          // ... and the cos_sg sums of the non-orthogonal divergence_corner at row Q (:1798-1843)
      if constexpr (GM == 0 && CS)
        if (mw) {
          cosav_r = LU(g.cosa_v, R); rsinv_r = LU(g.rsin_v, R);
          if (a.nord > 0) {
            csu = LA(g.cos_sg + 3 * nAp, Q - 1) + LA(g.cos_sg + nAp, Q);        // cos_sg(i,j-1,4) + cos_sg(i,j,2)
            csv = LA(g.cos_sg + 2 * nAp, Q, -1) + LA(g.cos_sg, Q);              // cos_sg(i-1,j,3) + cos_sg(i,j,1)
          }
        }
      for (int m = 0; m < KPW; m++) {
        CswLevel &S = st[m];
        const int k = kl[m];
        const size_t oA = (size_t)k * g.nA(), oU = (size_t)k * g.nU(), oV = (size_t)k * g.nV(), oB = (size_t)k * g.nB();
        const CswFields f = fnxt[m];
        fnxt[m] = load_fields(tn, k);
        S.u0 = S.u1; S.u1 = S.u2; S.u2 = S.u3; S.u3 = f.u;
        S.v0 = S.v1; S.v1 = S.v2; S.v2 = S.v3; S.v3 = f.v;
        S.dp0 = S.dpp; S.dpp = f.dp;
        S.pt0 = S.ptp; S.ptp = f.pt;
        S.w0 = S.wp; S.wp = f.w;
        // ---- row R: interpolated winds, fluxes, vorticity ----------------------------------------------------
        const vd utmp = a2 * (S.u0 + S.u3) + a1 * (S.u1 + S.u2);                 // :3099-3103
        const vd v3p = shl1(S.v3);
        S.vt0 = S.vt1; S.vt1 = S.vt2; S.vt2 = S.vt3;
        S.vt3 = a2 * (shr1(S.v3) + shl1(v3p)) + a1 * (S.v3 + v3p);               // vtmp(t-1), :3104-3108
        vd ua = utmp, va = S.vt2;
        if constexpr (GM == 0) {
          ua = (utmp - S.vt2 * in.cs) * in.rs;  va = (S.vt2 - utmp * in.cs) * in.rs;  // :3152-3157
        }
        const vd um1 = shr1(utmp);
        const vd uc = a2 * (shr1(um1) + shl1(utmp)) + a1 * (um1 + utmp);         // :3197-3199
        const vd vc = a2 * (S.vt0 + S.vt3) + a1 * (S.vt1 + S.vt2);               // :3337-3339
        vd ut = uc, vt = vc;
        if constexpr (GM == 0) {
          ut = (uc - S.v2 * in.cosau) * in.rsinu;                                   // :3200
          ut = vsel(ut > 0., dt2 * ut * in.dy * in.sg3, dt2 * ut * in.dy * in.sg1);  // :159-167
          if constexpr (CS) { if (mw) vt = (vc - S.u1 * cosav_r) * rsinv_r; }
          vt = vsel(vt > 0., dt2 * vt * in.dx * in.sg4, dt2 * vt * in.dx * in.sg2);  // :168-176 (vt = vc, :3350)
        } else {
          ut = dt2 * ut * in.dy;
          vt = dt2 * vc * in.dx;
        }
        const vd ucdx = uc * in.dxc;
        const vd vcdy = vc * in.dyc;
        const vd vort = in.fc + in.rac * (S.ucdx_p - ucdx - shr1(vcdy) + vcdy);  // :372-403
        {
          const bool on = live[m] && R >= jA && R <= jB && R >= oJ0 && R <= oJ1, on1 = on && R <= je + 1;
          const long iAr = (long)g.iA(ilo, R);
          vstore_b(a.ua + oA, iAr, ua, s1, on1);
          vstore_b(a.va + oA, iAr, va, s1, on1);
          vstore_b(a.ut + oA, iAr, ut, s2, on1);
          vstore_b(a.vt + oA, iAr, vt, s1, on);
        }
        // ---- row Q: KE, transport, wind update, divergence -------------------------------------------------------
        const vd ke = dt4 * (S.ua_p * vsel(S.ua_p > 0., S.uc_p, shl1(S.uc_p)) + S.va_p * vsel(S.va_p > 0., S.vc_p, vc));  // :297-366
        // upwind fluxes through y-face Q+1 (= R) and x-face i (:182-286)
        const vb vpos = vt > 0.;
        const vd fy1_n = vt * vsel(vpos, S.dp0, S.dpp);
        const vd fyp_n = fy1_n * vsel(vpos, S.pt0, S.ptp);
        const vd fyw_n = fy1_n * vsel(vpos, S.w0, S.wp);
        {
          const bool on = live[m] && Q >= jA && Q <= jB && Q >= oJ0 && Q <= oJ1, on1 = on && Q <= je + 1;
          const long iAq = (long)g.iA(ilo, Q);
          const vb upos = S.ut_p > 0.;
          const vd fx1 = S.ut_p * vsel(upos, shr1(S.dp0), S.dp0);
          const vd fxp = fx1 * vsel(upos, shr1(S.pt0), S.pt0);
          const vd ra = in.ra;
          const vd dpc = S.dp0 + (fx1 - shl1(fx1) + S.fy1_p - fy1_n) * ra;
          vstore_b(a.delpc + oA, iAq, dpc, s1, on1);
          vstore_b(a.ptc + oA, iAq, (S.pt0 * S.dp0 + (fxp - shl1(fxp) + S.fyp_p - fyp_n) * ra) / dpc, s1, on1);
          {
            const vd fxw = fx1 * vsel(upos, shr1(S.w0), S.w0);
            vstore_b((nh ? a.wc : a.ptc) + oA, iAq, (S.w0 * S.dp0 + (fxw - shl1(fxw) + S.fyw_p - fyw_n) * ra) / dpc, s1, on1 && nh);
          }
          // uc: interpolated value, advanced on [is, ie+1] x [js, je] (:414-447)
          vd ucv = S.uc_p;
          {
            const bool adv = Q >= js && Q <= je;
            vd fy1 = dt2 * S.v1;
            if constexpr (GM == 0) fy1 = dt2 * (S.v1 - ucv * cosau_p) / in.sinau;
            const vd fy = vsel(fy1 > 0., S.vort_p, vort);
            ucv = vsel(m_uc && vball(adv), ucv + fy1 * fy + in.rdxc * (shr1(ke) - ke), ucv);
          }
          vstore_b(a.uc + oV, (long)g.iV(ilo, Q), ucv, s2, on1);
          // vc: interpolated value, advanced on [is, ie] x [js, je+1] (:452-486)
          vd vcv = S.vc_p;
          {
            const bool adv = Q >= js && Q <= je + 1;
            vd fx1v = dt2 * S.u0;
            if constexpr (GM == 0) fx1v = dt2 * (S.u0 - vcv * in.cosav) / in.sinav;
            const vd fx = vsel(fx1v > 0., S.vort_p, shl1(S.vort_p));
            vcv = vsel(m_vc && vball(adv), vcv - fx1v * fx + in.rdyc * (S.ke_p - ke), vcv);
          }
          vstore_b(a.vc + oU, (long)g.iU(ilo, Q), vcv, s1, on);
        }
        // divergence at the corners of row Q (:1781-1796); v0 = v(Q-1), v1 = v(Q), u0 = u(Q)
        const vd dxc_q = GM == 2 ? vd(g.c_dxc) : dxc_p, dyc_q = GM == 2 ? vd(g.c_dyc) : dyc_p;
        const vd rac_q = GM == 2 ? vd(g.c_rarea_c) : rac_p;
        const vd vdxc = S.v1 * dxc_q;
        {
          const bool on = live[m] && a.nord > 0 && !mw && Q >= jA && Q <= jB;
          const vd uf = S.u0 * dyc_q;
          vstore_b(a.divg_d + oB, (long)g.iB(ilo, Q), rac_q * (S.vdxc_p - vdxc + shr1(uf) - uf), s2, on);
        }
        if constexpr (GM == 0 && CS) {
          if (mw && a.nord > 0) {  // the interior of a cubed-sphere face: the non-orthogonal form with ua, va of rows Q-1, Q
            const vd uf = (S.u0 - 0.25 * (S.va_pp + S.va_p) * csu) * dyc_q * 0.5 * (sg4_p + sg2_p);
            const vd vf = (S.v1 - 0.25 * (shr1(S.ua_p) + S.ua_p) * csv) * dxc_q * 0.5 * (sg3_p + sg1_p);
            vstore_b(a.divg_d + oB, (long)g.iB(ilo, Q), rac_q * (S.vf_p - vf + shr1(uf) - uf), s2,
                     live[m] && Q >= jA && Q <= jB && Q >= oJ0 && Q <= oJ1);
            S.vf_p = vf;
            S.va_pp = S.va_p;
          }
        }
        // ---- rotate the row state ------------------------------------------------------------------------------------
        S.ua_p = ua; S.va_p = va; S.uc_p = uc; S.vc_p = vc; S.ut_p = ut; S.vt_p = vt;
        S.ucdx_p = ucdx; S.vort_p = vort; S.ke_p = ke; S.vdxc_p = vdxc;
        S.fy1_p = fy1_n; S.fyp_p = fyp_n; S.fyw_p = fyw_n;
      }
      if constexpr (GM == 0) {
        cosau_p = in.cosau;
        if constexpr (CS) { sg3_p = in.sg3; sg1_p = in.sg1; sg4_p = in.sg4; sg2_p = in.sg2; }
      }
      if constexpr (GM <= 1) { dxc_p = in.dxc; dyc_p = in.dyc; rac_p = in.rac; }
    }
  }
};

}  // namespace fv3
