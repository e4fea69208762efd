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

struct MarchDims {
  int nstrips, nsegs, tj;
  const int *klist;  // level of the n-th marching slab (device), or null = identity
  FV3_HD int nwaves(int npz) const { return nstrips * nsegs * npz; }
};
inline MarchDims make_march_dims(const Grid &g, int tj) {
  MarchDims d;
  d.tj = tj;
  d.klist = nullptr;
  d.nstrips = num_strips(g);
  d.nsegs = (g.ny + tj - 1) / tj;
  return d;
}

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
    const int strip = gid % md.nstrips, seg = (gid / md.nstrips) % md.nsegs, kk = gid / (md.nstrips * md.nsegs);
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
    const int strip = gid % md.nstrips, seg = (gid / md.nstrips) % md.nsegs, kk = gid / (md.nstrips * md.nsegs);
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
