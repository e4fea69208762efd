// dsw_march.h -- the flux-form transports of d_sw (model/sw_core.F90:908-1066, :1249-1283) on the
// wave-marching fv_tp_2d (tp2d_march.h).  Two kernels:
//
//   DswDelpMarch<HORD>   : fv_tp_2d(delp) -> mass fluxes fx, fy (kept in a scratch pair for the scalars,
//                          accumulated into mfx / mfy), delp_out = delp + div(fx, fy)*rarea.
//   DswScalarMarch<HORD> : fv_tp_2d(q, mfx=fx, mfy=fy) for q in {w, q_con, pt}:
//                          q_out = (q*delp + div(gx, gy)*rarea) / delp_out.
//
// They cover the configuration without del-2n damping of the scalars (damp_w, damp_t, damp_vt <=
// threshold on every level: the reference defaults); fv3_d_sw falls back to the LDS-tile kernel
// DswTransport otherwise.
#pragma once

#include "dsw_kernels.h"
#include "tp2d_march.h"

namespace fv3 {

template <int HORD>
struct DswDelpMarch {
  Grid g;
  DswArgs a;
  MarchDims md;
  double *fxs, *fys;  // mass-flux scratch: FX kind and FY kind, npz levels
  int zero_heat;

  struct Sink {
    const DswDelpMarch &K;
    const StripGeom &s;
    int k;
    int lFx1;  // faces owned by this strip: its cells' west faces, plus face ie+1 on the last strip
    vl Fx;
    struct In {
      vd xf, y0, y1, mx, my, dp, ra;
    };
    FV3_D In load(int j) const {
      const Grid &g = K.g;
      const int ilo = s.ilo;
      const size_t oCX = (size_t)k * g.nCX(), oCY = (size_t)k * g.nCY(), oA = (size_t)k * g.nA();
      const size_t oFX = (size_t)k * g.nFX(), oFY = (size_t)k * g.nFY();
      In in;
      in.xf = vload(K.a.xfx + oCX, (long)g.iCX(ilo, j), s.F);
      in.y0 = vload(K.a.yfx + oCY, (long)g.iCY(ilo, j), s.A);
      in.y1 = vload(K.a.yfx + oCY, (long)g.iCY(ilo, j + 1), s.A);
      in.mx = vload(K.a.mfx + oFX, (long)g.iFX(ilo, j), Fx);
      in.my = vload(K.a.mfy + oFY, (long)g.iFY(ilo, j), s.C);
      const long iA = (long)g.iA(ilo, j);
      in.dp = vload(K.a.delp + oA, iA, s.C);
      in.ra = vload(g.rarea, iA, s.C);
      return in;
    }
    FV3_D void row(int j, const In &in, const vd &fxv, const vd &fyv0, const vd &fyv1) const {
      const Grid &g = K.g;
      const int ilo = s.ilo;
      const size_t oA = (size_t)k * g.nA();
      const size_t oFX = (size_t)k * g.nFX(), oFY = (size_t)k * g.nFY(), oCC = (size_t)k * g.nCC();
      const vd fxm = fxv * in.xf;  // tp_core.F90:217-226
      const vd fym0 = fyv0 * in.y0, fym1 = fyv1 * in.y1;
      const long iFX = (long)g.iFX(ilo, j), iFY0 = (long)g.iFY(ilo, j);
      vstore(K.fxs + oFX, iFX, fxm, s.lC0, lFx1);
      vstore(K.a.mfx + oFX, iFX, in.mx + fxm, s.lC0, lFx1);  // sw_core.F90:928-940
      vstore(K.fys + oFY, iFY0, fym0, s.lC0, s.lC1);
      vstore(K.a.mfy + oFY, iFY0, in.my + fym0, s.lC0, s.lC1);
      if (j == g.je) {
        const long iFY1 = (long)g.iFY(ilo, j + 1);
        vstore(K.fys + oFY, iFY1, fym1, s.lC0, s.lC1);
        const vd my1 = vload(K.a.mfy + oFY, iFY1, s.C);
        vstore(K.a.mfy + oFY, iFY1, my1 + fym1, s.lC0, s.lC1);
      }
      const long iA = (long)g.iA(ilo, j);
      vstore(K.a.delp_out + oA, iA, in.dp + (fxm - shl1(fxm) + fym0 - fym1) * in.ra, s.lC0, s.lC1);
      if (K.zero_heat) {  // :943-948
        const long iCC = (long)g.iCC(ilo, j);
        vstore(K.a.heat_s + oCC, iCC, vd(0.), s.lC0, s.lC1);
        vstore(K.a.diss_e + oCC, iCC, vd(0.), s.lC0, s.lC1);
      }
    }
  };

  FV3_D void operator()(int gid) const {
    int strip, seg, kk;
    md.decode(gid, strip, seg, kk);
    const int k = md.klist ? md.klist[kk] : kk;
    const StripGeom s = make_strip(g, strip);
    const int jA = g.js + seg * md.tj;
    const int jB = (jA + md.tj - 1 < g.je) ? jA + md.tj - 1 : g.je;
    const int lFx1 = (s.ilo + s.lC1 == g.ie) ? s.lC1 + 1 : s.lC1;
    Sink sink{*this, s, k, lFx1, make_lanes(s.lC0, lFx1)};
    tp2d_march<HORD>(g, s, jA, jB, a.delp + (size_t)k * g.nA(), a.crx + (size_t)k * g.nCX(),
                     a.cry + (size_t)k * g.nCY(), a.xfx + (size_t)k * g.nCX(), a.yfx + (size_t)k * g.nCY(), sink);
  }
};

template <int HORD>
struct DswScalarMarch {
  Grid g;
  DswArgs a;
  MarchDims md;
  const double *fxs, *fys;
  const double *q;
  double *q_out;

  struct Sink {
    const DswScalarMarch &K;
    const StripGeom &s;
    int k;
    struct In {
      vd mx, my0, my1, qo, dp, dpn, ra;
    };
    FV3_D In load(int j) const {
      const Grid &g = K.g;
      const int ilo = s.ilo;
      const size_t oA = (size_t)k * g.nA(), oFX = (size_t)k * g.nFX(), oFY = (size_t)k * g.nFY();
      In in;
      in.mx = vload(K.fxs + oFX, (long)g.iFX(ilo, j), s.F);
      in.my0 = vload(K.fys + oFY, (long)g.iFY(ilo, j), s.C);
      in.my1 = vload(K.fys + oFY, (long)g.iFY(ilo, j + 1), s.C);
      const long iA = (long)g.iA(ilo, j);
      in.qo = vload(K.q + oA, iA, s.C);
      in.dp = vload(K.a.delp + oA, iA, s.C);
      in.dpn = vload(K.a.delp_out + oA, iA, s.C);
      in.ra = vload(g.rarea, iA, s.C);
      return in;
    }
    FV3_D void row(int j, const In &in, const vd &fxv, const vd &fyv0, const vd &fyv1) const {
      const Grid &g = K.g;
      const size_t oA = (size_t)k * g.nA();
      const vd gx = fxv * in.mx, gy0 = fyv0 * in.my0, gy1 = fyv1 * in.my1;  // tp_core.F90:191-200
      // sw_core.F90:985-989, 1053-1066, 1262-1283
      const vd qn = (in.qo * in.dp + (gx - shl1(gx) + gy0 - gy1) * in.ra) / in.dpn;
      vstore(K.q_out + oA, (long)g.iA(s.ilo, j), qn, s.lC0, s.lC1);
    }
  };

  FV3_D void operator()(int gid) const {
    int strip, seg, kk;
    md.decode(gid, strip, seg, kk);
    const int k = md.klist ? md.klist[kk] : kk;
    const StripGeom s = make_strip(g, strip);
    const int jA = g.js + seg * md.tj;
    const int jB = (jA + md.tj - 1 < g.je) ? jA + md.tj - 1 : g.je;
    Sink sink{*this, s, k};
    tp2d_march<HORD>(g, s, jA, jB, q + (size_t)k * g.nA(), a.crx + (size_t)k * g.nCX(), a.cry + (size_t)k * g.nCY(),
                     a.xfx + (size_t)k * g.nCX(), a.yfx + (size_t)k * g.nCY(), sink);
  }
};

}  // namespace fv3

// =====================================================================================================
// d_sw momentum on the marching stencils, for the levels with nord_k == 1, no vorticity damping, no
// Smagorinsky coefficient (dddmp < 1e-5) and no dissipative heating -- the reference defaults below the
// sponge.  Other levels run in the LDS-tile kernel DswMomentum.
//
//   DswKeMarch<SWC>   : kinetic-energy flux at the corners via ytp_v / xtp_u (sw_core.F90:1078-1198) plus the
//                       del-4 divergence damping term (:1372-1460, nord = 1) -> ke scratch (B kind), delpc.
//   DswVortMarch<HORD>: absolute vorticity (:1231-1247, :1476-1495) computed row by row from u, v, its
//                       fv_tp_2d (:1498) and the D-grid wind update (:1500-1509).
namespace fv3 {

template <int SWC>
struct DswKeMarch {
  Grid g;
  DswArgs a;
  MarchDims md;   // nsegs counts segments of corner rows js..je+1
  double *ke;     // B kind scratch, npz levels

  FV3_D void operator()(int gid) const {
    int strip, seg, kk;
    md.decode(gid, strip, seg, kk);
    const int k = md.klist ? md.klist[kk] : kk;
    const StripGeom s = make_strip(g, strip);
    const int ilo = s.ilo;
    const int jA = g.js + seg * md.tj;
    int jB = (jA + md.tj - 1 < g.je) ? jA + md.tj - 1 : g.je;
    if (jB == g.je) jB = g.je + 1;  // the last segment also owns corner row je+1
    const int lFx1 = (ilo + s.lC1 == g.ie) ? s.lC1 + 1 : s.lC1;
    const double *u = a.u + (size_t)k * g.nU(), *v = a.v + (size_t)k * g.nV();
    const double *uc = a.uc + (size_t)k * g.nV(), *vc = a.vc + (size_t)k * g.nU();
    const double *dv = a.divg_d + (size_t)k * g.nB();
    double *kek = ke + (size_t)k * g.nB();
    double *dpc = a.delpc ? a.delpc + (size_t)k * g.nA() : nullptr;
    const double dt5 = 0.5 * a.dt;
    const double d2_bg = a.lv.d2_divg[k];
    const double damp2 = g.da_min_c * dmax(d2_bg, dmin(0.20, a.dddmp * 0.));            // :1454 with vort = 0
    const double dd8 = g.stretched_grid ? g.da_min * ipow(a.d4_bg, 2) : ipow(g.da_min_c * a.d4_bg, 2);  // :1446-1450
    PpmYsw<SWC> yv;
    yv.init();
    for (int r = jA - 3; r <= jB + 2; r++) {
      yv.push(vload(v, (long)g.iV(ilo, r), s.A));
      const int jc = r - 2;
      if (jc < jA) continue;
      // ---- ytp_v: v advected by vb along y (:1129-1139) ---------------------------------------------------
      const long oU = (long)g.iU(ilo, jc), oV = (long)g.iV(ilo, jc), oVm = (long)g.iV(ilo, jc - 1);
      const vd vcr = vload(vc, oU, s.A);
      const vd vb = dt5 * (shr1(vcr) + vcr);
      const vd ub = yv.face(vb, vload(g.rdy, oVm, s.A), vload(g.rdy, oV, s.A));
      const vd kev = vb * ub;
      // ---- xtp_u: u advected by ub along x (:1186-1196) ------------------------------------------------------
      const vd ub2 = dt5 * (vload(uc, oVm, s.A) + vload(uc, oV, s.A));
      const vd vb2 = ppm_faces_x_sw<SWC>(vload(u, oU, s.A), ub2, vload(g.rdx, oU, s.A));
      vd kex = 0.5 * (kev + ub2 * vb2);
      // ---- divergence damping, nord = 1 (:1372-1460) -------------------------------------------------------------
      const long oB = (long)g.iB(ilo, jc);
      const vd d0 = vload(dv, oB, s.A), dm = vload(dv, (long)g.iB(ilo, jc - 1), s.A),
               dp = vload(dv, (long)g.iB(ilo, jc + 1), s.A);
      const vd vc2 = (shl1(d0) - d0) * vload(g.divg_u, oU, s.A);      // :1392-1396
      const vd uc2m = (d0 - dm) * vload(g.divg_v, oVm, s.A);          // :1399-1403
      const vd uc2 = (dp - d0) * vload(g.divg_v, oV, s.A);
      vd lap = uc2m - uc2 + shr1(vc2) - vc2;                          // :1406-1424
      if (!g.stretched_grid) lap = lap * vload(g.rarea_c, oB, s.A);
      const vd vdmp = damp2 * d0 + dd8 * lap;                         // :1455
      kex = kex + vdmp;
      vstore(kek, oB, kex, s.lC0, lFx1);
      if (dpc) vstore(dpc, (long)g.iA(ilo, jc), d0, s.lC0, lFx1);     // delpc = saved divergence (:1376-1381)
    }
  }
};

template <int HORD>
struct DswVortMarch {
  Grid g;
  DswArgs a;
  MarchDims md;
  const double *ke;  // B kind scratch written by DswKeMarch

  struct Src {
    const DswVortMarch &K;
    const StripGeom &s;
    int k;
    struct In {
      vd u0, u1, v0, v1, dx0, dx1, dy0, dy1, ra, f0;
    };
    FV3_D In load(int r) const {
      const Grid &g = K.g;
      const double *u = K.a.u + (size_t)k * g.nU(), *v = K.a.v + (size_t)k * g.nV();
      const long oU = (long)g.iU(s.ilo, r), oU1 = (long)g.iU(s.ilo, r + 1), oV = (long)g.iV(s.ilo, r);
      const long oA = (long)g.iA(s.ilo, r);
      In in;
      in.u0 = vload(u, oU, s.A);   in.dx0 = vload(g.dx, oU, s.A);
      in.u1 = vload(u, oU1, s.A);  in.dx1 = vload(g.dx, oU1, s.A);
      in.v0 = vload(v, oV, s.A);   in.dy0 = vload(g.dy, oV, s.A);
      in.v1 = vload(v, oV + 1, s.A);  in.dy1 = vload(g.dy, oV + 1, s.A);  // V kind has the extra column ied+1
      in.ra = vload(g.rarea, oA, s.A);
      in.f0 = vload(g.f0, oA, s.A);
      return in;
    }
    FV3_D vd value(const In &in) const {  // :1231-1247, :1476-1495
      const vd vt0 = in.u0 * in.dx0, vt1 = in.u1 * in.dx1, ut0 = in.v0 * in.dy0, ut1 = in.v1 * in.dy1;
      return in.ra * (vt0 - vt1 - ut0 + ut1) + in.f0;
    }
  };

  struct Sink {
    const DswVortMarch &K;
    const StripGeom &s;
    int k, lFx1;
    struct In {
      vd u, dx, v, dy, ke0, ke1, xf, yf;
    };
    FV3_D In load(int j) const {
      const Grid &g = K.g;
      const long oU = (long)g.iU(s.ilo, j), oV = (long)g.iV(s.ilo, j);
      In in;
      in.u = vload(K.a.u + (size_t)k * g.nU(), oU, s.A);
      in.dx = vload(g.dx, oU, s.A);
      in.v = vload(K.a.v + (size_t)k * g.nV(), oV, s.A);
      in.dy = vload(g.dy, oV, s.A);
      in.ke0 = vload(K.ke + (size_t)k * g.nB(), (long)g.iB(s.ilo, j), s.A);
      in.ke1 = vload(K.ke + (size_t)k * g.nB(), (long)g.iB(s.ilo, j + 1), s.A);
      in.xf = vload(K.a.xfx + (size_t)k * g.nCX(), (long)g.iCX(s.ilo, j), s.F);
      in.yf = vload(K.a.yfx + (size_t)k * g.nCY(), (long)g.iCY(s.ilo, j), s.A);
      return in;
    }
    FV3_D void row(int j, const In &in, const vd &fxv, const vd &fyv0, const vd &fyv1) const {
      const Grid &g = K.g;
      double *uo = K.a.u_out + (size_t)k * g.nU(), *vo = K.a.v_out + (size_t)k * g.nV();
      // u = vt + ke(i,j) - ke(i+1,j) + fy (:1500-1504);  v = ut + ke(i,j) - ke(i,j+1) - fx (:1505-1509)
      const vd un = in.u * in.dx + in.ke0 - shl1(in.ke0) + fyv0 * in.yf;
      const vd vn = in.v * in.dy + in.ke0 - in.ke1 - fxv * in.xf;
      vstore(uo, (long)g.iU(s.ilo, j), un, s.lC0, s.lC1);
      vstore(vo, (long)g.iV(s.ilo, j), vn, s.lC0, lFx1);
      if (j == g.je) {  // the north edge row of u
        const long oU1 = (long)g.iU(s.ilo, j + 1);
        const vd u1 = vload(K.a.u + (size_t)k * g.nU(), oU1, s.A), dx1 = vload(g.dx, oU1, s.A);
        const vd yf1 = vload(K.a.yfx + (size_t)k * g.nCY(), (long)g.iCY(s.ilo, j + 1), s.A);
        vstore(uo, oU1, u1 * dx1 + in.ke1 - shl1(in.ke1) + fyv1 * yf1, s.lC0, s.lC1);
      }
    }
  };

  FV3_D void operator()(int gid) const {
    int strip, seg, kk;
    md.decode(gid, strip, seg, kk);
    const int k = md.klist ? md.klist[kk] : kk;
    const StripGeom s = make_strip(g, strip);
    const int jA = g.js + seg * md.tj;
    const int jB = (jA + md.tj - 1 < g.je) ? jA + md.tj - 1 : g.je;
    const int lFx1 = (s.ilo + s.lC1 == g.ie) ? s.lC1 + 1 : s.lC1;
    const Src src{*this, s, k};
    Sink sink{*this, s, k, lFx1};
    tp2d_march_src<HORD>(g, s, jA, jB, src, a.crx + (size_t)k * g.nCX(), a.cry + (size_t)k * g.nCY(),
                         a.xfx + (size_t)k * g.nCX(), a.yfx + (size_t)k * g.nCY(), sink);
  }
};

}  // namespace fv3

// =====================================================================================================
// The other two fv_tp_2d users on the marching stencil (undamped levels / sub-steps; the LDS-tile kernels
// ZhTransport and TracerStep keep the del-2n damped cases).
namespace fv3 {

// update_dz_d, model/nh_utils.F90:256-301: fv_tp_2d of one interface height + flux-form update
#ifndef FV3_ZH_BF
#define FV3_ZH_BF 0   // 1: the row step without a branch around a load / store (tp2d_march.h sink_branch_free).  Measured at C384 L127
                      // (tools/tz_ab.py, same arrays): 0.299 - 0.314 ms either way -- this kernel is not held up by its waitcnt; off
#endif
template <int HORD>
struct ZhMarch {
  Grid g;
  MarchDims md;
  const double *zh, *crx, *cry, *xfx, *yfx;  // interface-level Courant numbers / fluxes (km+1 levels)
  double *zh_out;

  struct Sink {
    const ZhMarch &K;
    const StripGeom &s;
    int k;
    struct In {
      vd z, ar, x0, y0, y1;
    };
    FV3_D In load(int j) const {
      const Grid &g = K.g;
      const long iA = (long)g.iA(s.ilo, j);
      In in;
      in.z = vload(K.zh + (size_t)k * g.nA(), iA, s.A);
      in.ar = vload(g.area, iA, s.A);
      in.x0 = vload(K.xfx + (size_t)k * g.nCX(), (long)g.iCX(s.ilo, j), s.F);
      in.y0 = vload(K.yfx + (size_t)k * g.nCY(), (long)g.iCY(s.ilo, j), s.A);
      in.y1 = vload(K.yfx + (size_t)k * g.nCY(), (long)g.iCY(s.ilo, j + 1), s.A);
      return in;
    }
    FV3_D void row(int j, const In &in, const vd &fxv, const vd &fyv0, const vd &fyv1) const {
      const Grid &g = K.g;
      const vd x1 = shl1(in.x0);
      const vd fx0 = fxv * in.x0, fy0 = fyv0 * in.y0, fy1 = fyv1 * in.y1;
      const vd rax = in.ar + in.x0 - x1, ray = in.ar + in.y0 - in.y1;
      const vd zn = (in.z * in.ar + fx0 - shl1(fx0) + fy0 - fy1) / (rax + ray - in.ar);  // :283-299
      vstore(K.zh_out + (size_t)k * g.nA(), (long)g.iA(s.ilo, j), zn, s.lC0, s.lC1);
    }
#if FV3_ZH_BF
    static constexpr bool kBranchFree = true;
    FV3_D void row_bf(int j, const In &in, const vd &fxv, const vd &fyv0, const vd &fyv1, bool on) const {
      const Grid &g = K.g;
      const vd x1 = shl1(in.x0);
      const vd fx0 = fxv * in.x0, fy0 = fyv0 * in.y0, fy1 = fyv1 * in.y1;
      const vd rax = in.ar + in.x0 - x1, ray = in.ar + in.y0 - in.y1;
      const vd zn = (in.z * in.ar + fx0 - shl1(fx0) + fy0 - fy1) / (rax + ray - in.ar);  // :283-299
      vstore_b(K.zh_out + (size_t)k * g.nA(), (long)g.iA(s.ilo, j), zn, make_mask(s.lC0, s.lC1), on);
    }
#endif
  };

  FV3_D void operator()(int gid) const {
    int strip, seg, kk;
    md.decode(gid, strip, seg, kk);
    const int k = md.klist ? md.klist[kk] : kk;
    const StripGeom s = make_strip(g, strip);
    const int jA = g.js + seg * md.tj;
    const int jB = (jA + md.tj - 1 < g.je) ? jA + md.tj - 1 : g.je;
    Sink sink{*this, s, k};
    tp2d_march<HORD>(g, s, jA, jB, zh + (size_t)k * g.nA(), crx + (size_t)k * g.nCX(), cry + (size_t)k * g.nCY(),
                     xfx + (size_t)k * g.nCX(), yfx + (size_t)k * g.nCY(), sink);
  }
};

// fv_tp_2d of one field with the fluxes themselves as the result (tp_core.F90:187-224 without mass fluxes: fx = 0.5 (fx + fx2) xfx):
// the absolute vorticity of d_sw (sw_core.F90:1483-1485) on the levels the fused momentum kernel does not take.  On a cubed-sphere
// face the values within the frame along the edges are overwritten afterwards by the frame kernel (cubed_tpf.h).
template <int HORD>
struct FluxMarch {
  Grid g;
  MarchDims md;
  const double *q, *crx, *cry, *xfx, *yfx;
  double *fx, *fy;   // FX / FY

  struct Sink {
    const FluxMarch &K;
    const StripGeom &s;
    int k, lFx1;
    vl Fx;
    struct In {
      vd x0, y0, y1;
    };
    FV3_D In load(int j) const {
      const Grid &g = K.g;
      In in;
      in.x0 = vload(K.xfx + (size_t)k * g.nCX(), (long)g.iCX(s.ilo, j), s.F);
      in.y0 = vload(K.yfx + (size_t)k * g.nCY(), (long)g.iCY(s.ilo, j), s.A);
      in.y1 = vload(K.yfx + (size_t)k * g.nCY(), (long)g.iCY(s.ilo, j + 1), s.A);
      return in;
    }
    FV3_D void row(int j, const In &in, const vd &fxv, const vd &fyv0, const vd &fyv1) const {
      const Grid &g = K.g;
      const size_t oFX = (size_t)k * g.nFX(), oFY = (size_t)k * g.nFY();
      vstore(K.fx + oFX, (long)g.iFX(s.ilo, j), fxv * in.x0, s.lC0, lFx1);
      vstore(K.fy + oFY, (long)g.iFY(s.ilo, j), fyv0 * in.y0, s.lC0, s.lC1);
      if (j == g.je) vstore(K.fy + oFY, (long)g.iFY(s.ilo, j + 1), fyv1 * in.y1, s.lC0, s.lC1);
    }
  };

  FV3_D void operator()(int gid) const {
    int strip, seg, kk;
    md.decode(gid, strip, seg, kk);
    const int k = md.klist ? md.klist[kk] : kk;
    const StripGeom s = make_strip(g, strip);
    const int jA = g.js + seg * md.tj;
    const int jB = (jA + md.tj - 1 < g.je) ? jA + md.tj - 1 : g.je;
    const int lFx1 = (s.ilo + s.lC1 == g.ie) ? s.lC1 + 1 : s.lC1;
    Sink sink{*this, s, k, lFx1, make_lanes(s.lC0, lFx1)};
    tp2d_march<HORD>(g, s, jA, jB, q + (size_t)k * g.nA(), crx + (size_t)k * g.nCX(), cry + (size_t)k * g.nCY(),
                     xfx + (size_t)k * g.nCX(), yfx + (size_t)k * g.nCY(), sink);
  }
};

// one sub-cycle of tracer_2d for every (tracer, level), model/fv_tracer2d.F90:471-541, without the del-2n
// damping of the first sub-cycle (trdm <= 1e-4)
template <int HORD>
struct TracerMarch {
  Grid g;
  MarchDims md;
  int npz, nq, it, nsplt;
  const int *ksplt;  // device, npz
  const double *q, *dp1, *mfx, *mfy, *cx, *cy, *xfx, *yfx;
  double *q_out, *dp1_out;

  struct Sink {
    const TracerMarch &K;
    const StripGeom &s;
    int k, iq;
    struct In {
      vd q, d1, ra, mx, my0, my1;
    };
    FV3_D In load(int j) const {
      const Grid &g = K.g;
      const long iA = (long)g.iA(s.ilo, j);
      In in;
      in.q = vload(K.q + ((size_t)iq * K.npz + k) * g.nA(), iA, s.A);
      in.d1 = vload(K.dp1 + (size_t)k * g.nA(), iA, s.A);
      in.ra = vload(g.rarea, iA, s.A);
      in.mx = vload(K.mfx + (size_t)k * g.nFX(), (long)g.iFX(s.ilo, j), s.F);
      in.my0 = vload(K.mfy + (size_t)k * g.nFY(), (long)g.iFY(s.ilo, j), s.C);
      in.my1 = vload(K.mfy + (size_t)k * g.nFY(), (long)g.iFY(s.ilo, j + 1), s.C);
      return in;
    }
    FV3_D void row(int j, const In &in, const vd &fxv, const vd &fyv0, const vd &fyv1) const {
      const Grid &g = K.g;
      const long iA = (long)g.iA(s.ilo, j);
      const vd gx = fxv * in.mx, gy0 = fyv0 * in.my0, gy1 = fyv1 * in.my1;
      const vd dp2 = in.d1 + (in.mx - shl1(in.mx) + in.my0 - in.my1) * in.ra;                 // :517-522
      const vd qn = (in.q * in.d1 + (gx - shl1(gx) + gy0 - gy1) * in.ra) / dp2;              // :523-531
      vstore(K.q_out + ((size_t)iq * K.npz + k) * g.nA(), iA, qn, s.lC0, s.lC1);
      if (iq == K.nq - 1 && K.it != K.nsplt) vstore(K.dp1_out + (size_t)k * g.nA(), iA, dp2, s.lC0, s.lC1);
    }
  };

  FV3_D void operator()(int gid) const {
    int strip, seg, kq;
    md.decode(gid, strip, seg, kq);
    const int k = kq % npz, iq = kq / npz;
    const StripGeom s = make_strip(g, strip);
    const int jA = g.js + seg * md.tj;
    const int jB = (jA + md.tj - 1 < g.je) ? jA + md.tj - 1 : g.je;
    const size_t oq = ((size_t)iq * npz + k) * g.nA();
    if (it > ksplt[k]) {  // the level is finished: carry q (and dp1) over to the output buffers
      for (int j = jA; j <= jB; j++) {
        const long iA = (long)g.iA(s.ilo, j);
        vstore(q_out + oq, iA, vload(q + oq, iA, s.A), s.lC0, s.lC1);
        if (iq == nq - 1 && it != nsplt)
          vstore(dp1_out + (size_t)k * g.nA(), iA, vload(dp1 + (size_t)k * g.nA(), iA, s.A), s.lC0, s.lC1);
      }
      return;
    }
    Sink sink{*this, s, k, iq};
    tp2d_march<HORD>(g, s, jA, jB, q + oq, cx + (size_t)k * g.nCX(), cy + (size_t)k * g.nCY(),
                     xfx + (size_t)k * g.nCX(), yfx + (size_t)k * g.nCY(), sink);
  }
};

}  // namespace fv3
