// remap_kernels.h -- the vertical remap Lagrangian_to_Eulerian (model/fv_mapz.F90:56-845) and the
// profile / mapping operators it uses (model/fv_operators.F90: scalar_profile :546-916, cs_profile
// :919-1300, cs_limiters :1303-1378, map_scalar :40, map1_ppm :137, mapn_tracer :234, map1_q2 :352).
//
// One thread per column, consecutive threads = consecutive i.  The piecewise-parabolic coefficients
// a4(1:4,k), the interface values and the two pressure coordinates of a column live in context-owned
// scratch slabs laid out like the fields (level stride = one A slab), so every sweep over k is a
// coalesced access across the wavefront; the data-dependent search of the mapping loop only moves
// forward (k0 carry) and neighbouring columns move together.
// Branches: remap_te = .false., use_cond = moist_kappa = .false., consv = 0, fill = .false.;
// kord 8..15 (scalar_profile / cs_profile) and kord <= 7 (ppm_profile :1382-1639 + ppm_limiters :1642-1723, what the map
// routines call for "kord > 7" false, with the SIGNED kord); iv in {-2,-1,0,1} (iv=-3 is undefined behaviour in the reference).
#pragma once

#include "fv3_common.h"

namespace fv3 {

struct ColScr {
  double *a1, *q, *a2, *a3, *a4, *pe1, *pe2, *gam;  // slabs, (km+1) levels each
  size_t ls;                                         // level stride
  int o;                                             // column offset inside a slab
};
#define CS(p, k) (c.p)[(size_t)((k)-1) * c.ls + c.o]
// where column `col` of a kernel keeps its scratch levels.  blocked: the 64 columns of a wavefront are one contiguous
// (km+1) x 64 block per slab, so the k loop of a wavefront walks through 64 KB of consecutive addresses instead of
// touching one 512-byte piece in each of km+1 planes megabytes apart; otherwise the slabs have the layout of the fields.
FV3_HD void scr_col(ColScr &c, int col, int fo, int km, size_t nA, int blocked) {
  if (blocked) {
    c.ls = 64;
    c.o = (col >> 6) * 64 * (km + 1) + (col & 63);
  } else {
    c.ls = nA;
    c.o = fo;
  }
}

// what the map routines do with a kord: > 7 -> scalar_profile / cs_profile (built for 8..15), otherwise ppm_profile (any value)
FV3_HD bool kord_supported(int kord) { return kord <= 15; }
FV3_HD bool kord_is_ppm(int kord) { return kord <= 7; }

// cs_limiters for one cell (fv_operators.F90:1303-1378)
FV3_HD void cs_limit(bool extm, double a1, double &a2, double &a3, double &a4, int iv) {
  constexpr double r12 = 1. / 12.;
  if (iv == 0) {
    if (a1 <= 0.) {
      a2 = a1; a3 = a1; a4 = 0.;
    } else if (fabs(a3 - a2) < -a4) {
      if ((a1 + 0.25 * ((a3 - a2) * (a3 - a2)) / a4 + a4 * r12) < 0.) {
        if (a1 < a3 && a1 < a2) {
          a3 = a1; a2 = a1; a4 = 0.;
        } else if (a3 > a2) {
          a4 = 3. * (a2 - a1); a3 = a2 - a4;
        } else {
          a4 = 3. * (a3 - a1); a2 = a3 - a4;
        }
      }
    }
    return;
  }
  bool flat;
  if (iv == 1)
    flat = (a1 - a2) * (a1 - a3) >= 0.;
  else
    flat = extm;
  if (flat) {
    a2 = a1; a3 = a1; a4 = 0.;
  } else {
    const double da1 = a3 - a2, da2 = da1 * da1, a6da = a4 * da1;
    if (a6da < -da2) {
      a4 = 3. * (a2 - a1); a3 = a2 - a4;
    } else if (a6da > da2) {
      a4 = 3. * (a3 - a1); a2 = a3 - a4;
    }
  }
}

// the unfused back-substitution + constraint + limiter sweeps (|kord| = 11, 12: their limiters test the interface values
// of both neighbouring cells)
FV3_HD void profile_col_tail_unfused(const ColScr &c, int km, bool is_scalar, int iv, int ak, double qmin) {
  {  // back-substitution
    double qk = CS(q, km + 1);
    if (iv == -2) {
      qk = CS(q, km);
      for (int k = km - 1; k >= 1; k--) {
        qk = CS(q, k) - CS(gam, k + 1) * qk;
        CS(q, k) = qk;
      }
    } else {
      for (int k = km; k >= 1; k--) {
        qk = CS(q, k) - CS(gam, k) * qk;
        CS(q, k) = qk;
      }
    }
  }
  // ---- large-scale constraints on the interface values (:643-680 / :1037-1073) ----
  {
    const double a_1 = CS(a1, 1), a_2 = CS(a1, 2);
    double v = dmin(CS(q, 2), dmax(a_1, a_2));
    CS(q, 2) = dmax(v, dmin(a_1, a_2));
    for (int k = 3; k <= km - 1; k++) {
      const double am2 = CS(a1, k - 2), am1 = CS(a1, k - 1), a0 = CS(a1, k), ap1 = CS(a1, k + 1);
      const double gm = am1 - am2, gp = ap1 - a0;  // gam(k-1), gam(k+1)
      double qk = CS(q, k);
      if (ak >= 14 || gm * gp > 0.) {
        qk = dmin(qk, dmax(am1, a0));
        qk = dmax(qk, dmin(am1, a0));
      } else if (gm > 0.) {
        qk = dmax(qk, dmin(am1, a0));
      } else {
        qk = dmin(qk, dmax(am1, a0));
        if (iv == 0) qk = dmax(0., qk);
      }
      CS(q, k) = qk;
    }
    const double b1 = CS(a1, km - 1), b0 = CS(a1, km);
    v = dmin(CS(q, km), dmax(b1, b0));
    CS(q, km) = dmax(v, dmin(b1, b0));
  }
  // ---- subgrid constraints (:691-914 / :1082-1298) ----
  auto dq = [&](int k) { return CS(a1, k) - CS(a1, k - 1); };                     // gam(k) after :650
  auto extm = [&](int k) {
    if (k == 1 || k == km) return (CS(q, k) - CS(a1, k)) * (CS(q, k + 1) - CS(a1, k)) > 0.;
    return dq(k) * dq(k + 1) < 0.;
  };
  auto ext5 = [&](int k) {
    const double x0 = 2. * CS(a1, k) - (CS(q, k) + CS(q, k + 1));
    return fabs(x0) > fabs(CS(q, k) - CS(q, k + 1));
  };
  auto ext6 = [&](int k) {
    const double x0 = 2. * CS(a1, k) - (CS(q, k) + CS(q, k + 1));
    return fabs(3. * x0) > fabs(CS(q, k) - CS(q, k + 1));
  };
  for (int k = 1; k <= km; k++) {
    const double a1v = CS(a1, k);
    double a2v = CS(q, k), a3v = CS(q, k + 1), a4v;
    if (k == 1) {
      const bool e = extm(1);
      if (iv == 0) a2v = dmax(0., a2v);
      if (iv == -1 && a2v * a1v <= 0.) a2v = 0.;
      a4v = 3. * (2. * a1v - (a2v + a3v));
      cs_limit(e, a1v, a2v, a3v, a4v, 1);
    } else if (k == 2) {
      a4v = 3. * (2. * a1v - (a2v + a3v));
      cs_limit(extm(2), a1v, a2v, a3v, a4v, 2);
    } else if (k <= km - 2) {
      const double g_k = dq(k), g_p1 = dq(k + 1), g_p2 = dq(k + 2), g_m1 = dq(k - 1);
      auto huynh = [&]() {
        const double pmp_1 = a1v - 2. * g_p1, lac_1 = pmp_1 + 1.5 * g_p2;
        a2v = dmin(dmax(a2v, dmin3(a1v, pmp_1, lac_1)), dmax3(a1v, pmp_1, lac_1));
        const double pmp_2 = a1v + 2. * g_k, lac_2 = pmp_2 - 1.5 * g_m1;
        a3v = dmin(dmax(a3v, dmin3(a1v, pmp_2, lac_2)), dmax3(a1v, pmp_2, lac_2));
      };
      if (ak <= 8) {
        huynh();
        a4v = 3. * (2. * a1v - (a2v + a3v));
      } else if (ak == 9) {
        const bool e = extm(k);
        if ((e && extm(k - 1)) || (e && extm(k + 1)) || (is_scalar && e && a1v < qmin)) {
          a2v = a1v; a3v = a1v; a4v = 0.;
        } else {
          a4v = is_scalar ? 3. * (2. * a1v - (a2v + a3v)) : 6. * a1v - 3. * (a2v + a3v);
          if (fabs(a4v) > fabs(a2v - a3v)) {
            huynh();
            a4v = is_scalar ? 3. * (2. * a1v - (a2v + a3v)) : 6. * a1v - 3. * (a2v + a3v);
          }
        }
      } else if (ak == 10) {
        if (extm(k)) {
          if ((is_scalar && a1v < qmin) || extm(k - 1) || extm(k + 1)) {
            a2v = a1v; a3v = a1v; a4v = 0.;
          } else {
            a4v = 6. * a1v - 3. * (a2v + a3v);
          }
        } else {
          a4v = 6. * a1v - 3. * (a2v + a3v);
          if (fabs(a4v) > fabs(a2v - a3v)) {
            huynh();
            a4v = 6. * a1v - 3. * (a2v + a3v);
          }
        }
      } else if (ak == 11) {
        if (ext5(k) && (ext5(k - 1) || ext5(k + 1) || (is_scalar && a1v < qmin))) {
          a2v = a1v; a3v = a1v; a4v = 0.;
        } else {
          a4v = 3. * (2. * a1v - (a2v + a3v));
        }
      } else if (ak == 12) {  // post-AM4 case 10 (:835-866 / :1225-1256)
        if (ext5(k)) {
          if (ext5(k - 1) || ext5(k + 1)) {
            a2v = a1v; a3v = a1v;
          } else if (ext6(k - 1) || ext6(k + 1)) {
            huynh();
          }
        } else if (ext6(k)) {
          if (ext5(k - 1) || ext5(k + 1)) huynh();
        }
        a4v = 3. * (2. * a1v - (a2v + a3v));
      } else {  // 13
        a4v = 3. * (2. * a1v - (a2v + a3v));
      }
      if (iv == 0 && ak <= 13) cs_limit(false, a1v, a2v, a3v, a4v, 0);
    } else {
      if (k == km) {
        if (iv == 0) a3v = dmax(0., a3v);
        if (iv == -1 && a3v * a1v <= 0.) a3v = 0.;
      }
      a4v = 3. * (2. * a1v - (a2v + a3v));
      cs_limit(extm(k), a1v, a2v, a3v, a4v, k == km ? 1 : 2);
    }
    CS(a2, k) = a2v;
    CS(a3, k) = a3v;
    CS(a4, k) = a4v;
  }
}


// Subgrid constraints of one cell (scalar_profile :691-914 / cs_profile :1082-1298): cell k with the constrained
// interface values a2v = q(k), a3v = q(k+1) (in / out) and the layer means a1(k-2 .. k+2) (0 outside 1..km); a4o = a4(4,k).
struct ProfCfg {
  int km, iv, ak;
  bool is_scalar;
  double qmin;
  bool streamed;  // true: c.q holds the constrained interface values and the mapping loop calls cs_cell on demand;
                  // false (|kord| = 11): a2, a3, a4 are in their slabs
};
FV3_HD void cs_cell(const ProfCfg &pc, int k, double &a2v, double &a3v, double am2, double am1, double a1v, double ap1,
                    double ap2, double &a4o) {
  const int km = pc.km, iv = pc.iv, ak = pc.ak;
  const bool is_scalar = pc.is_scalar;
  const double qmin = pc.qmin;
    // cell k with interface values a2v = q(k), a3v = q(k+1) and layer means a1(k-2..k+2)
    double a4v;
    const double g_m1 = am1 - am2, g_k = a1v - am1, g_p1 = ap1 - a1v, g_p2 = ap2 - ap1;  // dq(k-1), dq(k), dq(k+1), dq(k+2)
    auto extm_q = [&]() { return (a2v - a1v) * (a3v - a1v) > 0.; };  // k == 1 or km (:691, :1082)
    if (k == 1) {
      const bool e = extm_q();
      if (iv == 0) a2v = dmax(0., a2v);
      if (iv == -1 && a2v * a1v <= 0.) a2v = 0.;
      a4v = 3. * (2. * a1v - (a2v + a3v));
      cs_limit(e, a1v, a2v, a3v, a4v, 1);
    } else if (k == 2) {
      a4v = 3. * (2. * a1v - (a2v + a3v));
      cs_limit(g_k * g_p1 < 0., a1v, a2v, a3v, a4v, 2);
    } else if (k <= km - 2) {
      auto huynh = [&]() {
        const double pmp_1 = a1v - 2. * g_p1, lac_1 = pmp_1 + 1.5 * g_p2;
        a2v = dmin(dmax(a2v, dmin3(a1v, pmp_1, lac_1)), dmax3(a1v, pmp_1, lac_1));
        const double pmp_2 = a1v + 2. * g_k, lac_2 = pmp_2 - 1.5 * g_m1;
        a3v = dmin(dmax(a3v, dmin3(a1v, pmp_2, lac_2)), dmax3(a1v, pmp_2, lac_2));
      };
      // extm(k-1), extm(k), extm(k+1) for 3 <= k <= km-2: k-1 >= 2 and k+1 <= km-1 are interior (a1-based) except
      // k+1 == km ... which cannot happen here (k <= km-2)
      const bool e_k = g_k * g_p1 < 0.;
      if (ak <= 8) {
        huynh();
        a4v = 3. * (2. * a1v - (a2v + a3v));
      } else if (ak == 9) {
        const bool e_m = g_m1 * g_k < 0., e_p = g_p1 * g_p2 < 0.;
        if ((e_k && e_m) || (e_k && e_p) || (is_scalar && e_k && a1v < qmin)) {
          a2v = a1v; a3v = a1v; a4v = 0.;
        } else {
          a4v = is_scalar ? 3. * (2. * a1v - (a2v + a3v)) : 6. * a1v - 3. * (a2v + a3v);
          if (fabs(a4v) > fabs(a2v - a3v)) {
            huynh();
            a4v = is_scalar ? 3. * (2. * a1v - (a2v + a3v)) : 6. * a1v - 3. * (a2v + a3v);
          }
        }
      } else if (ak == 10) {
        const bool e_m = g_m1 * g_k < 0., e_p = g_p1 * g_p2 < 0.;
        if (e_k) {
          if ((is_scalar && a1v < qmin) || e_m || e_p) {
            a2v = a1v; a3v = a1v; a4v = 0.;
          } else {
            a4v = 6. * a1v - 3. * (a2v + a3v);
          }
        } else {
          a4v = 6. * a1v - 3. * (a2v + a3v);
          if (fabs(a4v) > fabs(a2v - a3v)) {
            huynh();
            a4v = 6. * a1v - 3. * (a2v + a3v);
          }
        }
      } else if (ak == 14) {  // strict monotonicity constraint (:883-884 / :1267-1268)
        a4v = 3. * (2. * a1v - (a2v + a3v));
        cs_limit(e_k, a1v, a2v, a3v, a4v, 2);
      } else if (ak == 15) {  // :885-886 / :1269-1270
        a4v = 3. * (2. * a1v - (a2v + a3v));
        cs_limit(false, a1v, a2v, a3v, a4v, 1);
      } else {  // 13
        a4v = 3. * (2. * a1v - (a2v + a3v));
      }
      if (iv == 0 && ak <= 13) cs_limit(false, a1v, a2v, a3v, a4v, 0);
    } else {
      bool e;
      if (k == km) {
        e = extm_q();  // uses q(km), q(km+1) before the iv adjustments of a3 (:700 / :1091 come first)
        if (iv == 0) a3v = dmax(0., a3v);
        if (iv == -1 && a3v * a1v <= 0.) a3v = 0.;
      } else {
        e = g_k * g_p1 < 0.;
      }
      a4v = 3. * (2. * a1v - (a2v + a3v));
      cs_limit(e, a1v, a2v, a3v, a4v, k == km ? 1 : 2);
    }
    a4o = a4v;
}

// ppm_limiters for one cell (fv_operators.F90:1642-1723)
FV3_HD void ppm_limit(double dm, double a1, double &a2, double &a3, double &a6, int lmt) {
  constexpr double r12 = 1. / 12.;
  if (lmt == 3) return;
  if (lmt == 0) {
    if (dm == 0.) {
      a2 = a1; a3 = a1; a6 = 0.;
    } else {
      const double da1 = a3 - a2, da2 = da1 * da1, a6da = a6 * da1;
      if (a6da < -da2) {
        a6 = 3. * (a2 - a1); a3 = a2 - a6;
      } else if (a6da > da2) {
        a6 = 3. * (a3 - a1); a2 = a3 - a6;
      }
    }
  } else if (lmt == 1) {
    const double qmp = 2. * dm;
    a2 = a1 - copysign(dmin(fabs(qmp), fabs(a2 - a1)), qmp);
    a3 = a1 + copysign(dmin(fabs(qmp), fabs(a3 - a1)), qmp);
    a6 = 3. * (2. * a1 - (a2 + a3));
  } else if (lmt == 2) {
    if (fabs(a3 - a2) < -a6) {
      const double fm = a1 + 0.25 * ((a3 - a2) * (a3 - a2)) / a6 + a6 * r12;
      if (fm < 0.) {
        if (a1 < a3 && a1 < a2) {
          a3 = a1; a2 = a1; a6 = 0.;
        } else if (a3 > a2) {
          a6 = 3. * (a2 - a1); a3 = a2 - a6;
        } else {
          a6 = 3. * (a3 - a1); a2 = a3 - a6;
        }
      }
    }
  }
}

// ppm_profile of one column (fv_operators.F90:1382-1639; kord <= 7).  a1 from src, the source coordinate in c.pe1; writes the
// a2, a3, a4 slabs (the stored form of ProfCfg); dc lives in c.q, h2 in c.gam; delq, d4, df2 are formed from a1 / pe1 where
// they are used (same expressions, same roundings as the reference's arrays).  km >= 5.
template <class Src>
FV3_HD ProfCfg profile_col_ppm(const ColScr &c, int km, int iv, int kord, const Src &src) {
  ProfCfg pc{km, iv, 0, false, 0., false};
  const int km1 = km - 1;
  for (int k = 1; k <= km; k++) CS(a1, k) = src(k);
  auto DP = [&](int k) { return CS(pe1, k + 1) - CS(pe1, k); };
  auto DQ = [&](int k) { return CS(a1, k + 1) - CS(a1, k); };   // delq(k)
  auto D4 = [&](int k) { return DP(k - 1) + DP(k); };
  for (int k = 2; k <= km1; k++) {  // limited slopes dc(k)
    const double am = CS(a1, k - 1), a0 = CS(a1, k), ap = CS(a1, k + 1);
    const double c1 = (DP(k - 1) + 0.5 * DP(k)) / D4(k + 1);
    const double c2 = (DP(k + 1) + 0.5 * DP(k)) / D4(k);
    const double df2 = DP(k) * (c1 * DQ(k) + c2 * DQ(k - 1)) / (D4(k) + DP(k + 1));
    const double hi = dmax(dmax(am, a0), ap) - a0, lo = a0 - dmin(dmin(am, a0), ap);
    CS(q, k) = copysign(dmin(dmin(fabs(df2), hi), lo), df2);
  }
  for (int k = 3; k <= km1; k++) {  // 4th order interpolation of the provisional cell edge value
    const double c1 = DQ(k - 1) * DP(k - 1) / D4(k);
    const double a1_ = D4(k - 1) / (D4(k) + DP(k - 1));
    const double a2_ = D4(k + 1) / (D4(k) + DP(k));
    CS(a2, k) = CS(a1, k - 1) + c1 + 2. / (D4(k - 1) + D4(k + 1)) * (DP(k) * (c1 * (a1_ - a2_) + a2_ * CS(q, k - 1)) - DP(k - 1) * a1_ * CS(q, k));
  }
  {  // top: area preserving cubic with 2nd derivative = 0 at the boundary
    const double d1 = DP(1), d2 = DP(2), t1 = CS(a1, 1), t2 = CS(a1, 2);
    const double qm = (d2 * t1 + d1 * t2) / (d1 + d2);
    const double dq = 2. * (t2 - t1) / (d1 + d2);
    const double c1 = 4. * (CS(a2, 3) - qm - d2 * dq) / (d2 * (2. * d2 * d2 + d1 * (d2 + 3. * d1)));
    const double c3 = dq - 0.5 * c1 * (d2 * (5. * d1 + d2) - 3. * d1 * d1);
    double e2 = qm - 0.25 * c1 * d1 * d2 * (d2 + 3. * d1);
    double e1 = d1 * (2. * c1 * (d1 * d1) - c3) + e2;
    e2 = dmax(e2, dmin(t1, t2));
    e2 = dmin(e2, dmax(t1, t2));
    CS(q, 1) = 0.5 * (e2 - t1);
    if (iv == 0) {
      e1 = dmax(0., e1);
      e2 = dmax(0., e2);
    } else if (iv == -1) {
      if (e1 * t1 <= 0.) e1 = 0.;
    } else if (iv == 2 || iv == -2) {
      e1 = t1;
    }
    CS(a2, 1) = e1;
    CS(a2, 2) = e2;
  }
  {  // bottom
    const double d1 = DP(km), d2 = DP(km1), b0 = CS(a1, km), b1 = CS(a1, km1);
    const double qm = (d2 * b0 + d1 * b1) / (d1 + d2);
    const double dq = 2. * (b1 - b0) / (d1 + d2);
    const double c1 = (CS(a2, km1) - qm - d2 * dq) / (d2 * (2. * d2 * d2 + d1 * (d2 + 3. * d1)));
    const double c3 = dq - 2.0 * c1 * (d2 * (5. * d1 + d2) - 3. * d1 * d1);
    double e2 = qm - c1 * d1 * d2 * (d2 + 3. * d1);
    double e3 = d1 * (8. * c1 * (d1 * d1) - c3) + e2;
    e2 = dmax(e2, dmin(b0, b1));
    e2 = dmin(e2, dmax(b0, b1));
    CS(q, km) = 0.5 * (b0 - e2);
    if (iv == 0) {
      e2 = dmax(0., e2);
      e3 = dmax(0., e3);
    } else if (iv < 0) {
      if (b0 * e3 <= 0.) e3 = 0.;
    }
    CS(a2, km) = e2;
    CS(a3, km) = e3;
  }
  for (int k = 1; k <= km1; k++) CS(a3, k) = CS(a2, k + 1);   // (abs(iv) == 2 sets a4(3,1) = a4(1,1) first; this overwrites it, :1544)
  auto finish = [&](int k, int lmt, bool form_a6) {
    const double a1v = CS(a1, k);
    double a2v = CS(a2, k), a3v = CS(a3, k);
    double a6 = form_a6 ? 3. * (2. * a1v - (a2v + a3v)) : CS(a4, k);
    ppm_limit(CS(q, k), a1v, a2v, a3v, a6, lmt);
    CS(a2, k) = a2v; CS(a3, k) = a3v; CS(a4, k) = a6;
  };
  finish(1, 0, true);
  finish(2, 0, true);
  if (kord >= 7) {  // Huynh's 2nd constraint
    for (int k = 2; k <= km1; k++)
      CS(gam, k) = 2. * (CS(q, k + 1) / DP(k + 1) - CS(q, k - 1) / DP(k - 1)) / (DP(k) + 0.5 * (DP(k - 1) + DP(k + 1))) * (DP(k) * DP(k));
    const double fac = 1.5;
    for (int k = 3; k <= km - 2; k++) {
      const double a1v = CS(a1, k), dck = CS(q, k);
      const double pmp = 2. * dck;
      double qmp = a1v + pmp;
      double lac = a1v + fac * CS(gam, k - 1) + dck;
      double a3v = dmin(dmax(CS(a3, k), dmin(dmin(a1v, qmp), lac)), dmax(dmax(a1v, qmp), lac));
      qmp = a1v - pmp;
      lac = a1v + fac * CS(gam, k + 1) - dck;
      double a2v = dmin(dmax(CS(a2, k), dmin(dmin(a1v, qmp), lac)), dmax(dmax(a1v, qmp), lac));
      double a6 = 3. * (2. * a1v - (a2v + a3v));
      if (iv == 0 && kord >= 6) ppm_limit(dck, a1v, a2v, a3v, a6, 2);
      // a3(k) = a2(k+1) is read by nothing below: cell k + 1 takes its own left edge from the a2 slab
      CS(a2, k) = a2v; CS(a3, k) = a3v; CS(a4, k) = a6;
    }
  } else {
    int lmt = kord - 3;
    if (lmt < 0) lmt = 0;
    if (iv == 0 && lmt > 2) lmt = 2;
    for (int k = 3; k <= km - 2; k++) {
      if (kord != 6)
        finish(k, lmt, kord != 4);
      else
        CS(a4, k) = 3. * (2. * CS(a1, k) - (CS(a2, k) + CS(a3, k)));
    }
  }
  finish(km1, 0, true);
  finish(km, 0, true);
  return pc;
}

// scalar_profile (is_scalar) / cs_profile of one column.  src(k) yields the layer mean a4(1,k) of the field (it is
// called once per level, in order, and the value is kept in c.a1 for the mapping loop); the source coordinate is in
// c.pe1.  Writes c.a2, c.a3, c.a4 (and uses c.q, c.gam).
//
// Two sweeps over k instead of the reference's five loop nests: (1) forward elimination of the cubic-spline
// tridiagonal, fused with fetching the field; (2) ONE backward sweep that does the back-substitution, the large-scale
// constraints on the interface values (:643-680 / :1037-1073) and the subgrid limiters (:691-914 / :1082-1298) with a
// sliding 5-level register window of a1 -- every operation is the reference's, only the loop nests are merged, so the
// results are bit-identical while the scratch-slab traffic drops from ~18 to ~11 accesses per level.  The backward sweep
// leaves the CONSTRAINED interface values in c.q; the subgrid limiters (cs_cell) run in the mapping loop, so a2/a3/a4 are
// never stored (another 4 accesses per level).
// (|kord| = 11 tests the monotonicity of the NEXT-higher cell's interface values, which a descending sweep has not
// produced yet; it keeps the unfused sweeps below.)
template <class Src>
FV3_HD ProfCfg profile_col(const ColScr &c, int km, bool is_scalar, double qs, int iv, int kord, double qmin, const Src &src) {
  if (kord_is_ppm(kord)) return profile_col_ppm(c, km, iv, kord, src);   // "if (kord > 7) ... else call ppm_profile"
  const int ak = kord < 0 ? -kord : kord;
  ProfCfg pc{km, iv, ak, is_scalar, qmin, ak != 11 && ak != 12};
#define DP(k) (CS(pe1, (k) + 1) - CS(pe1, k))
  // ---- interface values: cubic spline tridiagonal, forward elimination ----
  if (iv == -2) {  // :572-595 / :941-964
    double a_prev = src(1);
    CS(a1, 1) = a_prev;
    double gam = 0.5, qk = 1.5 * a_prev;
    CS(q, 1) = qk;
    CS(gam, 2) = gam;
    double pe_a = CS(pe1, 1), pe_b = CS(pe1, 2);
    double dp_prev = pe_b - pe_a;
    for (int k = 2; k <= km - 1; k++) {
      const double a_k = src(k);
      CS(a1, k) = a_k;
      pe_a = pe_b;
      pe_b = CS(pe1, k + 1);
      const double dp_k = pe_b - pe_a;
      const double grat = dp_prev / dp_k;
      const double bet = 2. + grat + grat - gam;
      qk = (3. * (a_prev + a_k) - qk) / bet;
      gam = grat / bet;
      CS(q, k) = qk;
      CS(gam, k + 1) = gam;
      a_prev = a_k;
      dp_prev = dp_k;
    }
    const double a_km = src(km);
    CS(a1, km) = a_km;
    const double grat = dp_prev / (CS(pe1, km + 1) - pe_b);
    qk = (3. * (a_prev + a_km) - grat * qs - qk) / (2. + grat + grat - gam);
    CS(q, km) = qk;
    CS(q, km + 1) = qs;
  } else {  // :597-623 / :967-1016
    const double a_1 = src(1), a_2 = src(2);
    CS(a1, 1) = a_1;
    CS(a1, 2) = a_2;
    double pe_a = CS(pe1, 2), pe_b = CS(pe1, 3);
    double dp_prev = pe_a - CS(pe1, 1), dp_k = pe_b - pe_a;  // DP(1), DP(2)
    double grat = dp_k / dp_prev;
    double bet = grat * (grat + 0.5);
    double qk = ((grat + grat) * (grat + 1.) * a_1 + a_2) / bet;
    double gam = (1. + grat * (grat + 1.5)) / bet;
    CS(q, 1) = qk;
    CS(gam, 1) = gam;
    double d4 = 0.;
    double a_prev = a_1, a_k = a_2;
    for (int k = 2; k <= km; k++) {
      if (k > 2) {
        a_prev = a_k;
        a_k = src(k);
        CS(a1, k) = a_k;
        dp_prev = dp_k;
        pe_a = pe_b;
        pe_b = CS(pe1, k + 1);
        dp_k = pe_b - pe_a;
      }
      d4 = dp_prev / dp_k;
      bet = 2. + d4 + d4 - gam;
      qk = (3. * (a_prev + d4 * a_k) - qk) / bet;
      gam = d4 / bet;
      CS(q, k) = qk;
      CS(gam, k) = gam;
    }
    const double a_bot = 1. + d4 * (d4 + 1.5);
    qk = (2. * d4 * (d4 + 1.) * a_k + a_prev - a_bot * qk) / (d4 * (d4 + 0.5) - a_bot * gam);
    CS(q, km + 1) = qk;
  }
#undef DP
  if (ak == 11 || ak == 12) {
    profile_col_tail_unfused(c, km, is_scalar, iv, ak, qmin);
    return pc;
  }
  // ---- backward sweep: back-substitution + constraints + subgrid limiters -------------------------------------------
  // window of layer means: w_m2 = a1(k-2) .. w_p2 = a1(k+2) for the cell k being finished
  double qraw = CS(q, km + 1);            // unconstrained q(k+1) of the recurrence
  double w_p1 = 0., w_0 = CS(a1, km), w_m1 = CS(a1, km - 1), w_m2 = km >= 3 ? CS(a1, km - 2) : 0.;
  for (int k = km; k >= 1; k--) {
    // back-substitution (:590-595 / :1010-1016); gam index differs between the two eliminations
    double qk;
    if (iv == -2)
      qk = (k == km) ? CS(q, km) : CS(q, k) - CS(gam, k + 1) * qraw;
    else
      qk = CS(q, k) - CS(gam, k) * qraw;
    qraw = qk;
    // large-scale constraint on q(k) (:643-680 / :1037-1073); window: w_m2 = a1(k-2), w_m1 = a1(k-1), w_0 = a1(k), w_p1 = a1(k+1)
    double qc = qk;
    if (k == 2) {
      const double v = dmin(qc, dmax(w_m1, w_0));
      qc = dmax(v, dmin(w_m1, w_0));
    } else if (k == km && km >= 3) {
      const double v = dmin(qc, dmax(w_m1, w_0));
      qc = dmax(v, dmin(w_m1, w_0));
    } else if (k >= 3 && k <= km - 1) {
      const double gm = w_m1 - w_m2, gp = w_p1 - w_0;  // gam(k-1), gam(k+1)
      if (ak >= 14 || gm * gp > 0.) {
        qc = dmin(qc, dmax(w_m1, w_0));
        qc = dmax(qc, dmin(w_m1, w_0));
      } else if (gm > 0.) {
        qc = dmax(qc, dmin(w_m1, w_0));
      } else {
        qc = dmin(qc, dmax(w_m1, w_0));
        if (iv == 0) qc = dmax(0., qc);
      }
    }
    // the constrained interface value replaces the raw one (already consumed by the recurrence); the cell coefficients
    // a4(2:4,k) are formed from it by the mapping loop (cs_cell) instead of being stored and re-read
    CS(q, k) = qc;
    // slide the window down one level
    w_p1 = w_0; w_0 = w_m1; w_m1 = w_m2;
    w_m2 = (k - 3 >= 1) ? CS(a1, k - 3) : 0.;
  }
  return pc;
}

// the search-and-integrate loop (fv_operators.F90:93-132 == :188-227 == :402-441; tracer_form: :277-335)
template <class Out>
FV3_HD void map_col(const ColScr &c, int km, bool tracer_form, const ProfCfg &pc, const Out &out) {
  constexpr double r3 = 1. / 3., r23 = 2. / 3.;
  int k0 = 1;
  double qsum = 0.;
  // a4(2:4, l): from the slabs, or formed here from the constrained interface values and the layer means around l (the
  // cell a target layer ends in is the one the next target layer starts in: keep the last one)
  int c_l = 0;
  double c_2 = 0., c_3 = 0., c_4 = 0.;
  auto coef = [&](int l, double &b2, double &b3, double &b4) {
    if (!pc.streamed) {
      b2 = CS(a2, l); b3 = CS(a3, l); b4 = CS(a4, l);
      return;
    }
    if (l != c_l) {
      double a2v = CS(q, l), a3v = CS(q, l + 1), a4v;
      const double am2 = l - 2 >= 1 ? CS(a1, l - 2) : 0., am1 = l - 1 >= 1 ? CS(a1, l - 1) : 0.;
      const double ap1 = l + 1 <= km ? CS(a1, l + 1) : 0., ap2 = l + 2 <= km ? CS(a1, l + 2) : 0.;
      cs_cell(pc, l, a2v, a3v, am2, am1, CS(a1, l), ap1, ap2, a4v);
      c_l = l; c_2 = a2v; c_3 = a3v; c_4 = a4v;
    }
    b2 = c_2; b3 = c_3; b4 = c_4;
  };
  for (int k = 1; k <= km; k++) {
    const double p2t = CS(pe2, k), p2b = CS(pe2, k + 1);
    int done = 0;
    for (int l = k0; l <= km && !done; l++) {
      const double p1t = CS(pe1, l), p1b = CS(pe1, l + 1);
      if (p2t >= p1t && p2t <= p1b) {
        const double dp1 = p1b - p1t;
        const double pl = (p2t - p1t) / dp1;
        double b2, b3, b4;
        coef(l, b2, b3, b4);
        if (p2b <= p1b) {
          const double pr = (p2b - p1t) / dp1;
          double val;
          if (tracer_form) {
            double fac1 = pr + pl;
            const double fac2 = r3 * (pr * fac1 + pl * pl);
            fac1 = 0.5 * fac1;
            val = b2 + (b4 + b3 - b2) * fac1 - b4 * fac2;
          } else {
            val = b2 + 0.5 * (b4 + b3 - b2) * (pr + pl) - b4 * r3 * (pr * (pr + pl) + pl * pl);
          }
          out(k, val);
          k0 = l;
          done = 2;
        } else {
          if (tracer_form) {
            const double dp = p1b - p2t;
            double fac1 = 1. + pl;
            const double fac2 = r3 * (1. + pl * fac1);
            fac1 = 0.5 * fac1;
            qsum = dp * (b2 + (b4 + b3 - b2) * fac1 - b4 * fac2);
          } else {
            qsum = (p1b - p2t) * (b2 + 0.5 * (b4 + b3 - b2) * (1. + pl) - b4 * (r3 * (1. + pl * (1. + pl))));
          }
          for (int m = l + 1; m <= km; m++) {
            const double mt = CS(pe1, m), mb = CS(pe1, m + 1);
            if (p2b > mb) {
              qsum = qsum + (mb - mt) * CS(a1, m);
            } else {
              const double dp = p2b - mt;
              const double esl = dp / (mb - mt);
              double m2, m3, m4;
              coef(m, m2, m3, m4);
              if (tracer_form) {
                const double fac1 = 0.5 * esl, fac2 = 1. - r23 * esl;
                qsum = qsum + dp * (m2 + fac1 * (m3 - m2 + m4 * fac2));
              } else {
                qsum = qsum + dp * (m2 + 0.5 * esl * (m3 - m2 + m4 * (1. - r23 * esl)));
              }
              k0 = m;
              break;
            }
          }
          done = 1;
        }
      }
    }
    if (done != 2) out(k, qsum / (p2b - p2t));
  }
}

// ---- several tracers of one column side by side -------------------------------------------------------------------
// The elimination coefficients of the spline (grat, bet, gam), the search of the mapping loop and its fractional
// positions depend on the coordinates only: a group of NT tracers forms them once (one reciprocal per divisor, the
// quotients through it -- correctly rounded, see spmd.h vrecip / vdiv_r) and runs NT recurrences / limiters / integrals
// on them.  Same arithmetic per tracer as profile_col + map_col (iv = 0), streamed coefficients only (|kord| != 11).
FV3_HD double rcp_rn(double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  return __builtin_fma(y, e, y);
#else
  return 1. / b;
#endif
}
FV3_HD double div_rn(double a, double b, double y) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double q0 = a * y;
  const double r = __builtin_fma(-b, q0, a);
  return __builtin_fma(r, y, q0);
#else
  (void)y;
  return a / b;
#endif
}

template <int NT>
struct TrcGroup {
  double *a1[NT], *q[NT];  // profile slabs of each tracer: layer means, interface values
  double *f[NT];           // the tracer columns (element of level 1; level stride fs)
  int ak[NT];              // |kord_tr|
};

template <int NT>
FV3_HD void remap_tracers_col(const ColScr &c, const TrcGroup<NT> &t, size_t fs, int km, bool tracer_form) {
#define TA1(i, k) t.a1[i][(size_t)((k)-1) * c.ls + c.o]
#define TQ(i, k) t.q[i][(size_t)((k)-1) * c.ls + c.o]
#define TF(i, k) t.f[i][(size_t)((k)-1) * fs]
  // ---- forward elimination (fv_operators.F90:597-623 / :967-1016) ----
  {
    double a_prev[NT], a_k[NT], qk[NT];
    for (int i = 0; i < NT; i++) {
      a_prev[i] = TF(i, 1);
      a_k[i] = TF(i, 2);
      TA1(i, 1) = a_prev[i];
      TA1(i, 2) = a_k[i];
    }
    double pe_a = CS(pe1, 2), pe_b = CS(pe1, 3);
    double dp_prev = pe_a - CS(pe1, 1), dp_k = pe_b - pe_a;
    const double grat = dp_k / dp_prev;
    double bet = grat * (grat + 0.5);
    double rb = rcp_rn(bet);
    const double c1 = (grat + grat) * (grat + 1.);
    for (int i = 0; i < NT; i++) {
      qk[i] = div_rn(c1 * a_prev[i] + a_k[i], bet, rb);
      TQ(i, 1) = qk[i];
    }
    double gam = div_rn(1. + grat * (grat + 1.5), bet, rb);
    CS(gam, 1) = gam;
    double d4 = 0.;
    for (int k = 2; k <= km; k++) {
      if (k > 2) {
        for (int i = 0; i < NT; i++) {
          a_prev[i] = a_k[i];
          a_k[i] = TF(i, k);
          TA1(i, k) = a_k[i];
        }
        dp_prev = dp_k;
        pe_a = pe_b;
        pe_b = CS(pe1, k + 1);
        dp_k = pe_b - pe_a;
      }
      d4 = dp_prev / dp_k;
      bet = 2. + d4 + d4 - gam;
      rb = rcp_rn(bet);
      for (int i = 0; i < NT; i++) {
        qk[i] = div_rn(3. * (a_prev[i] + d4 * a_k[i]) - qk[i], bet, rb);
        TQ(i, k) = qk[i];
      }
      gam = div_rn(d4, bet, rb);
      CS(gam, k) = gam;
    }
    const double a_bot = 1. + d4 * (d4 + 1.5);
    const double den = d4 * (d4 + 0.5) - a_bot * gam, rden = rcp_rn(den);
    const double c2 = 2. * d4 * (d4 + 1.);
    for (int i = 0; i < NT; i++) TQ(i, km + 1) = div_rn(c2 * a_k[i] + a_prev[i] - a_bot * qk[i], den, rden);
  }
  // ---- backward sweep: back-substitution + large-scale constraints (:643-680 / :1037-1073), as profile_col ----
  {
    double qraw[NT], w_p1[NT], w_0[NT], w_m1[NT], w_m2[NT];
    for (int i = 0; i < NT; i++) {
      qraw[i] = TQ(i, km + 1);
      w_p1[i] = 0.;
      w_0[i] = TA1(i, km);
      w_m1[i] = TA1(i, km - 1);
      w_m2[i] = km >= 3 ? TA1(i, km - 2) : 0.;
    }
    for (int k = km; k >= 1; k--) {
      const double gam = CS(gam, k);
      for (int i = 0; i < NT; i++) {
        const double qk = TQ(i, k) - gam * qraw[i];
        qraw[i] = qk;
        double qc = qk;
        if (k == 2 || (k == km && km >= 3)) {
          const double v = dmin(qc, dmax(w_m1[i], w_0[i]));
          qc = dmax(v, dmin(w_m1[i], w_0[i]));
        } else if (k >= 3 && k <= km - 1) {
          const double gm = w_m1[i] - w_m2[i], gp = w_p1[i] - w_0[i];
          if (t.ak[i] >= 14 || gm * gp > 0.) {
            qc = dmin(qc, dmax(w_m1[i], w_0[i]));
            qc = dmax(qc, dmin(w_m1[i], w_0[i]));
          } else if (gm > 0.) {
            qc = dmax(qc, dmin(w_m1[i], w_0[i]));
          } else {
            qc = dmin(qc, dmax(w_m1[i], w_0[i]));
            qc = dmax(0., qc);  // iv = 0
          }
        }
        TQ(i, k) = qc;
        w_p1[i] = w_0[i]; w_0[i] = w_m1[i]; w_m1[i] = w_m2[i];
        w_m2[i] = (k - 3 >= 1) ? TA1(i, k - 3) : 0.;
      }
    }
  }
  // ---- search and integrate (fv_operators.F90:277-335 mapn_tracer / :402-441 map1_q2), as map_col ----
  {
    constexpr double r3 = 1. / 3., r23 = 2. / 3.;
    int k0 = 1, c_l = 0;
    double qsum[NT], c_2[NT], c_3[NT], c_4[NT];
    for (int i = 0; i < NT; i++) qsum[i] = c_2[i] = c_3[i] = c_4[i] = 0.;
    auto coef = [&](int l) {
      if (l == c_l) return;
      for (int i = 0; i < NT; i++) {
        const ProfCfg pc{km, 0, t.ak[i], true, 0., true};
        double a2v = TQ(i, l), a3v = TQ(i, l + 1), a4v;
        const double am2 = l - 2 >= 1 ? TA1(i, l - 2) : 0., am1 = l - 1 >= 1 ? TA1(i, l - 1) : 0.;
        const double ap1 = l + 1 <= km ? TA1(i, l + 1) : 0., ap2 = l + 2 <= km ? TA1(i, l + 2) : 0.;
        cs_cell(pc, l, a2v, a3v, am2, am1, TA1(i, l), ap1, ap2, a4v);
        c_2[i] = a2v; c_3[i] = a3v; c_4[i] = a4v;
      }
      c_l = l;
    };
    for (int k = 1; k <= km; k++) {
      const double p2t = CS(pe2, k), p2b = CS(pe2, k + 1);
      int done = 0;
      for (int l = k0; l <= km && !done; l++) {
        const double p1t = CS(pe1, l), p1b = CS(pe1, l + 1);
        if (p2t >= p1t && p2t <= p1b) {
          const double dp1 = p1b - p1t, rdp1 = rcp_rn(dp1);
          const double pl = div_rn(p2t - p1t, dp1, rdp1);
          coef(l);
          if (p2b <= p1b) {
            const double pr = div_rn(p2b - p1t, dp1, rdp1);
            if (tracer_form) {
              double fac1 = pr + pl;
              const double fac2 = r3 * (pr * fac1 + pl * pl);
              fac1 = 0.5 * fac1;
              for (int i = 0; i < NT; i++) TF(i, k) = c_2[i] + (c_4[i] + c_3[i] - c_2[i]) * fac1 - c_4[i] * fac2;
            } else {
              const double s1 = pr + pl, s2 = pr * (pr + pl) + pl * pl;
              for (int i = 0; i < NT; i++) TF(i, k) = c_2[i] + 0.5 * (c_4[i] + c_3[i] - c_2[i]) * s1 - c_4[i] * r3 * s2;
            }
            k0 = l;
            done = 2;
          } else {
            if (tracer_form) {
              const double dp = p1b - p2t;
              double fac1 = 1. + pl;
              const double fac2 = r3 * (1. + pl * fac1);
              fac1 = 0.5 * fac1;
              for (int i = 0; i < NT; i++) qsum[i] = dp * (c_2[i] + (c_4[i] + c_3[i] - c_2[i]) * fac1 - c_4[i] * fac2);
            } else {
              const double dp = p1b - p2t, s1 = 1. + pl, s2 = r3 * (1. + pl * (1. + pl));
              for (int i = 0; i < NT; i++) qsum[i] = dp * (c_2[i] + 0.5 * (c_4[i] + c_3[i] - c_2[i]) * s1 - c_4[i] * s2);
            }
            for (int m = l + 1; m <= km; m++) {
              const double mt = CS(pe1, m), mb = CS(pe1, m + 1);
              if (p2b > mb) {
                const double dm = mb - mt;
                for (int i = 0; i < NT; i++) qsum[i] = qsum[i] + dm * TA1(i, m);
              } else {
                const double dp = p2b - mt;
                const double esl = dp / (mb - mt);
                coef(m);
                if (tracer_form) {
                  const double fac1 = 0.5 * esl, fac2 = 1. - r23 * esl;
                  for (int i = 0; i < NT; i++) qsum[i] = qsum[i] + dp * (c_2[i] + fac1 * (c_3[i] - c_2[i] + c_4[i] * fac2));
                } else {
                  const double s1 = 0.5 * esl, s2 = 1. - r23 * esl;
                  for (int i = 0; i < NT; i++) qsum[i] = qsum[i] + dp * (c_2[i] + s1 * (c_3[i] - c_2[i] + c_4[i] * s2));
                }
                k0 = m;
                break;
              }
            }
            done = 1;
          }
        }
      }
      if (done != 2) {
        const double dp2 = p2b - p2t, rdp2 = rcp_rn(dp2);
        for (int i = 0; i < NT; i++) TF(i, k) = div_rn(qsum[i], dp2, rdp2);
      }
    }
  }
#undef TA1
#undef TQ
#undef TF
}

// ------------------------------------------------------------------------------------------------
struct RemapPar {
  int last_step, hydrostatic, adiabatic, nq, kord_mt, kord_wz, kord_tm, sphum;
  double akap, ptop, rdgas, grav, cv_air, r_vir, cp, t_min;
  // thermostruct%moist_kappa / use_cond (nonhydrostatic) and the inputs of moist_cv (fv3_set_moist)
  int moist_kappa, use_cond, nwat, liq_wat, rainwat, ice_wat, snowwat, graupel;
  double cv_vap, c_liq, c_ice;
  double *q_con, *cappa;  // A x km, written where the reference writes them (fv_mapz.F90:212-219, :463-478)
  int fill;               // flagstruct%fill: fillz on the remapped tracers
  int scr_blocked;        // layout of the scratch slabs (scr_col)
  // flagstruct%remap_te (fv_mapz.F90:232-286, :348-360, :576-619, :655-663; fv3_set_remap_te): total energy is remapped in the
  // place of T_v / theta_v.  hs: A; te: A x km work array (the reference's te argument); u_old: U x km copy of u before the remap
  int remap_te = 0;
  const double *hs = nullptr;
  double *te = nullptr;
  const double *u_old = nullptr;
};

// fillz of one tracer column (fv_fill.F90:34-137, default branch): q(k) at q[(k-1)*qs], dp2(k) = pe2(k+1) - pe2(k) through
// dp(k).  In place; the borrowing steps are sequential in k exactly as in the reference.
template <class Dp>
FV3_HD void fillz_col(int km, double *q, size_t qs, const Dp &dp) {
#define QK(k) q[(size_t)((k)-1) * qs]
  bool zfix = false;
  if (QK(1) < 0.) {
    QK(2) = QK(2) + QK(1) * dp(1) / dp(2);
    QK(1) = 0.;
  }
  for (int k = 2; k <= km - 1; k++) {
    if (QK(k) < 0.) {
      zfix = true;
      if (QK(k - 1) > 0.) {
        const double dq = dmin(QK(k - 1) * dp(k - 1), -QK(k) * dp(k));
        QK(k - 1) = QK(k - 1) - dq / dp(k - 1);
        QK(k) = QK(k) + dq / dp(k);
      }
      if (QK(k) < 0.0 && QK(k + 1) > 0.) {
        const double dq = dmin(QK(k + 1) * dp(k + 1), -QK(k) * dp(k));
        QK(k + 1) = QK(k + 1) - dq / dp(k + 1);
        QK(k) = QK(k) + dq / dp(k);
      }
    }
  }
  if (QK(km) < 0. && QK(km - 1) > 0.) {
    const double qup = QK(km - 1) * dp(km - 1), qly = -QK(km) * dp(km), dup = dmin(qly, qup);
    zfix = true;
    QK(km - 1) = QK(km - 1) - dup / dp(km - 1);
    QK(km) = QK(km) + dup / dp(km);
  }
  if (zfix) {
    double sum0 = 0., sum1 = 0.;
    for (int k = 2; k <= km; k++) sum0 = sum0 + QK(k) * dp(k);
    if (sum0 > 0.) {
      for (int k = 2; k <= km; k++) sum1 = sum1 + dmax(0., QK(k) * dp(k));
      const double fac = sum0 / sum1;
      for (int k = 2; k <= km; k++) QK(k) = dmax(0., fac * (QK(k) * dp(k)) / dp(k));
    }
  }
#undef QK
}

// moist_cv of one cell (fv_thermodynamics.F90:250-325 without the t1 special case): returns cvm, sets q_con.
// qk = &q(i,j,k,1), ns = stride between species
FV3_HD double moist_cv(const RemapPar &p, const double *qk, size_t ns, double &q_con) {
  auto Q = [&](int n) { return n > 0 ? qk[(size_t)(n - 1) * ns] : 0.; };
  double qv, ql, qs;
  switch (p.nwat) {
    case 2:
      qv = dmax(0., Q(p.sphum));
      qs = dmax(0., Q(p.liq_wat));
      q_con = qs;
      return (1. - qv) * p.cv_air + qv * p.cv_vap;
    case 3:
      qv = Q(p.sphum); ql = Q(p.liq_wat); qs = Q(p.ice_wat);
      q_con = ql + qs;
      return (1. - (qv + q_con)) * p.cv_air + qv * p.cv_vap + ql * p.c_liq + qs * p.c_ice;
    case 4:
      qv = Q(p.sphum);
      q_con = Q(p.liq_wat) + Q(p.rainwat);
      return (1. - (qv + q_con)) * p.cv_air + qv * p.cv_vap + q_con * p.c_liq;
    case 5:
      qv = Q(p.sphum); ql = Q(p.liq_wat) + Q(p.rainwat); qs = Q(p.ice_wat) + Q(p.snowwat);
      q_con = ql + qs;
      return (1. - (qv + q_con)) * p.cv_air + qv * p.cv_vap + ql * p.c_liq + qs * p.c_ice;
    case 6:
      qv = Q(p.sphum); ql = Q(p.liq_wat) + Q(p.rainwat); qs = Q(p.ice_wat) + Q(p.snowwat) + Q(p.graupel);
      q_con = ql + qs;
      return (1. - (qv + q_con)) * p.cv_air + qv * p.cv_vap + ql * p.c_liq + qs * p.c_ice;
    default:
      q_con = 0.;
      return p.cv_air;
  }
}

#define FV3_COL_FOR2(c, ncol) for (int c = bx * 256 + tid; c < (bx + 1) * 256 && c < (ncol); c += kNT)

// The remap of one time step is four launches:
//   RemapCoords    -- per column: ps and the source / target coordinates of the cell-centred fields (p, and log p for T_v)
//   RemapFields    -- one thread per (column, field): T_v / theta_v (+ omega), every tracer, w, u, v are remapped side by
//                     side, each with its own set of profile slabs.  A column kernel has only nx*ny threads and long
//                     serial k loops with data-dependent loads; running the fields concurrently instead of one after
//                     the other in the same thread multiplies the loads in flight by the number of fields.
//   RemapDelzFinal -- per column: delz (its source is -delz/delp of the OLD delp and it is overwritten in place, so it
//                     must follow the T_v task, which reads the old delz) and delp, pk, peln, pkz, pt (:426-503, :793-841)
//   RemapPe        -- pe(k) = ak + bk*ps

// source / target coordinates of the cell-centred fields (fv_mapz.F90:298-345, :363-374)
// map1_cubic with T_VAR = 1 (total energy in log p) and conserv = .true. (fv_operators.F90:1897-2096, call site fv_mapz.F90:353-355)
// of one column, in place in f (level stride fs).  Scratch slabs: a1 = log p of the old layer centres, q = of the new ones,
// a2 = their old differences, a3 = the interpolated values.  pe1 / pe2: the pressure coordinates of the column.
FV3_HD void map1_cubic_te_col(const ColScr &c, int km, double *f, size_t fs) {
  double vsum1 = 0., vsum2 = 0.;
  for (int k = 1; k <= km; k++) {
    CS(a1, k) = dlog(0.5 * (CS(pe1, k) + CS(pe1, k + 1)));
    CS(q, k) = dlog(0.5 * (CS(pe2, k) + CS(pe2, k + 1)));
  }
  for (int k = 1; k <= km - 1; k++) CS(a2, k) = CS(a1, k + 1) - CS(a1, k);
  for (int k = 1; k <= km; k++) vsum1 = vsum1 + f[(size_t)(k - 1) * fs] * (CS(pe1, k + 1) - CS(pe1, k));
  vsum1 = vsum1 / (CS(pe1, km + 1) - CS(pe1, 1));
  auto Q1 = [&](int k) { return f[(size_t)(k - 1) * fs]; };
  for (int k = 1; k <= km; k++) {
    const double P = CS(q, k);
    int lp0 = 1;
    while (lp0 <= km && CS(a1, lp0) < P) lp0 = lp0 + 1;
    const int lm1 = lp0 - 1 > 1 ? lp0 - 1 : 1;
    lp0 = lp0 < km ? lp0 : km;
    double r;
    if (lm1 == 1 && lp0 == 1)
      r = Q1(1) + (Q1(2) - Q1(1)) * (P - CS(a1, 1)) / (CS(a1, 2) - CS(a1, 1));
    else if (lm1 == km && lp0 == km)
      r = Q1(km) + (Q1(km) - Q1(km - 1)) * (P - CS(a1, km)) / (CS(a1, km) - CS(a1, km - 1));
    else if (lm1 == 1 || lp0 == km)
      r = Q1(lp0) + (Q1(lm1) - Q1(lp0)) * (P - CS(a1, lp0)) / (CS(a1, lm1) - CS(a1, lp0));
    else {
      const int lp1 = lp0 + 1, lm2 = lm1 - 1;
      const double plp1 = CS(a1, lp1), plp0 = CS(a1, lp0), plm1 = CS(a1, lm1), plm2 = CS(a1, lm2);
      const double dlp0 = CS(a2, lp0), dlm1 = CS(a2, lm1), dlm2 = CS(a2, lm2);
      const double ap1 = (P - plp0) * (P - plm1) * (P - plm2) / (dlp0 * (dlp0 + dlm1) * (dlp0 + dlm1 + dlm2));
      const double ap0 = (plp1 - P) * (P - plm1) * (P - plm2) / (dlp0 * dlm1 * (dlm1 + dlm2));
      const double am1 = (plp1 - P) * (plp0 - P) * (P - plm2) / (dlm1 * dlm2 * (dlp0 + dlm1));
      const double am2 = (plp1 - P) * (plp0 - P) * (plm1 - P) / (dlm2 * (dlm1 + dlm2) * (dlp0 + dlm1 + dlm2));
      r = ap1 * Q1(lp1) + ap0 * Q1(lp0) + am1 * Q1(lm1) + am2 * Q1(lm2);
    }
    CS(a3, k) = r;
  }
  for (int k = 1; k <= km; k++) vsum2 = vsum2 + CS(a3, k) * (CS(pe2, k + 1) - CS(pe2, k));
  vsum2 = vsum2 / (CS(pe2, km + 1) - CS(pe2, 1));
  for (int k = 1; k <= km; k++) f[(size_t)(k - 1) * fs] = CS(a3, k) + vsum1 - vsum2;
}

FV3_HD double te_wind_bracket(const Grid &g, const double *u, const double *v, int i, int j, int k);
FV3_HD double moist_cv(const RemapPar &p, const double *qk, size_t ns, double &q_con);

// remap_te, before the remap (fv_mapz.F90:232-286): te = cp T + KE + phis of every layer from the un-remapped state; pkz as the
// reference leaves it there (pkez :898-903 / :272, :282).  phiz: A x (km+1) scratch.
struct RemapTePre {
  Grid g;
  int km;
  RemapPar p;
  const double *u, *v, *w, *delz, *pt, *delp, *q, *pe, *pk;
  double *peln, *pkz, *phiz;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int ncol = g.nx * g.ny;
    const size_t nA = g.nA(), nCC = g.nCC();
    const double k1k = p.rdgas / p.cv_air, rrg = -p.rdgas / p.grav, akap = p.akap;
    const double rv = p.adiabatic ? 0. : p.r_vir;   // the reference's caller passes zvir = 0 for an adiabatic run
    FV3_COL_FOR2(col, ncol) {
      const int i = g.is + col % g.nx, j = g.js + col / g.nx;
      const int o = g.iA(i, j), occ = g.iCC(i, j);
      const double rs2 = g.rsin2[o];
      auto PE = [&](int k) { return pe[(size_t)(j - (g.js - 1)) * (g.nx + 2) * (km + 1) + (size_t)(k - 1) * (g.nx + 2) + (i - (g.is - 1))]; };
      auto PELN = [&](int k) -> double & { return peln[(size_t)(j - g.js) * g.nx * (km + 1) + (size_t)(k - 1) * g.nx + (i - g.is)]; };
      auto PK = [&](int k) { return pk[(size_t)(k - 1) * nCC + occ]; };
      double ph = p.hs[o];
      phiz[(size_t)km * nA + o] = ph;
      if (p.hydrostatic) {
        PELN(1) = dlog(p.ptop);
        for (int k = km; k >= 1; k--) {
          ph = ph + p.cp * pt[(size_t)(k - 1) * nA + o] * (PK(k + 1) - PK(k));
          phiz[(size_t)(k - 1) * nA + o] = ph;
        }
        for (int k = 1; k <= km + 1; k++) phiz[(size_t)(k - 1) * nA + o] = phiz[(size_t)(k - 1) * nA + o] * PE(k);
        for (int k = 1; k <= km; k++) {
          const size_t o3 = (size_t)(k - 1) * nA + o;
          const double pz = (PK(k + 1) - PK(k)) / (akap * (PELN(k + 1) - PELN(k)));
          pkz[(size_t)(k - 1) * nCC + occ] = pz;
          p.te[o3] = 0.25 * rs2 * te_wind_bracket(g, u, v, i, j, k) + p.cp * pt[o3] * pz + (phiz[o3 + nA] - phiz[o3]) / (PE(k + 1) - PE(k));
        }
      } else {
        for (int k = km; k >= 1; k--) {
          const size_t o3 = (size_t)(k - 1) * nA + o, c3 = (size_t)(k - 1) * nCC + occ;
          const double qv = p.sphum > 0 ? q[(size_t)(p.sphum - 1) * nA * km + o3] : 0.;
          const double ph1 = ph;
          ph = ph - p.grav * delz[c3];
          const double ww = w[o3], ke = 0.25 * rs2 * te_wind_bracket(g, u, v, i, j, k);
          if (p.moist_kappa) {
            double qc;
            const double cvm = moist_cv(p, q + o3, nA * km, qc);
            const double cap = p.rdgas / (p.rdgas + cvm / (1. + rv * qv));
            p.q_con[o3] = qc;
            p.cappa[o3] = cap;
            const double pz = dexp(cap / (1. - cap) * dlog(rrg * delp[o3] / delz[c3] * pt[o3]));
            pkz[c3] = pz;
            p.te[o3] = cvm * pt[o3] * pz / ((1. + rv * qv) * (1. - qc)) + 0.5 * (ww * ww) + ke + 0.5 * (ph1 + ph);
          } else {
            const double pz = dexp(k1k * dlog(rrg * delp[o3] / delz[c3] * pt[o3]));
            pkz[c3] = pz;
            p.te[o3] = p.cv_air * pt[o3] * pz / (1. + rv * qv) + 0.5 * (ww * ww) + ke + 0.5 * (ph1 + ph);
          }
        }
      }
    }
  }
};

// remap_te, after the remap of the winds (fv_mapz.F90:576-619): T_v and pkz of every layer from the remapped energy, then the
// conversion of pt the remap ends with (:793-841).  The reference's loop over j takes the kinetic energy of row j out of te with
// u(:, j) and v(:, j) remapped and u(:, j + 1) NOT yet (it is remapped in the next iteration): u_old supplies that row.
struct RemapTePost {
  Grid g;
  int km;
  RemapPar p;
  const double *ak, *bk;
  const double *u, *v, *w, *delz, *delp, *q, *pe, *pk, *peln;
  double *pt, *pkz;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int ncol = g.nx * g.ny;
    const size_t nA = g.nA(), nCC = g.nCC(), nU = g.nU(), nV = g.nV();
    const double rrg = -p.rdgas / p.grav, akap = p.akap;
    const double rv = p.adiabatic ? 0. : p.r_vir;
    FV3_COL_FOR2(col, ncol) {
      const int i = g.is + col % g.nx, j = g.js + col / g.nx;
      const int o = g.iA(i, j), occ = g.iCC(i, j);
      const double rs2 = g.rsin2[o], ca = g.cosa_s[o];
      const size_t peb = (size_t)(j - (g.js - 1)) * (g.nx + 2) * (km + 1) + (i - (g.is - 1));
      const double psfc = pe[peb + (size_t)km * (g.nx + 2)];
      auto PELN = [&](int k) { return peln[(size_t)(j - g.js) * g.nx * (km + 1) + (size_t)(k - 1) * g.nx + (i - g.is)]; };
      auto PK = [&](int k) { return pk[(size_t)(k - 1) * nCC + occ]; };
      double ph = p.hs[o];
      for (int k = km; k >= 1; k--) {
        const size_t o3 = (size_t)(k - 1) * nA + o, c3 = (size_t)(k - 1) * nCC + occ;
        const double u0 = u[(size_t)(k - 1) * nU + g.iU(i, j)], u1 = p.u_old[(size_t)(k - 1) * nU + g.iU(i, j + 1)];
        const double v0 = v[(size_t)(k - 1) * nV + g.iV(i, j)], v1 = v[(size_t)(k - 1) * nV + g.iV(i + 1, j)];
        const double ke = 0.25 * rs2 * (u0 * u0 + u1 * u1 + v0 * v0 + v1 * v1 - (u0 + u1) * (v0 + v1) * ca);
        double tv, pz;
        if (p.hydrostatic) {
          const double pe2k = (k == 1) ? p.ptop : ak[k - 1] + bk[k - 1] * psfc;
          const double dlnp = p.rdgas * (PELN(k + 1) - PELN(k));
          const double tpe = p.te[o3] - ph - ke;
          tv = tpe / (p.cp - pe2k * dlnp / delp[o3]);
          pz = (PK(k + 1) - PK(k)) / (akap * (PELN(k + 1) - PELN(k)));
          ph = ph + dlnp * tv;
        } else {
          const double qv = p.sphum > 0 ? q[(size_t)(p.sphum - 1) * nA * km + o3] : 0.;
          const double ph1 = ph, ww = w[o3];
          ph = ph1 - delz[c3] * p.grav;
          const double tpe = p.te[o3] - 0.5 * (ph + ph1) - 0.5 * (ww * ww) - ke;
          if (p.moist_kappa) {
            double qc;
            const double cvm = moist_cv(p, q + o3, nA * km, qc);
            const double cap = p.rdgas / (p.rdgas + cvm / (1. + rv * qv));
            p.q_con[o3] = qc;
            p.cappa[o3] = cap;
            tv = tpe / cvm * (1. + rv * qv) * (1. - qc);
            pz = dexp(cap * dlog(rrg * delp[o3] / delz[c3] * tv));
          } else {
            tv = tpe / p.cv_air * (1. + rv * qv);
            pz = dexp(akap * dlog(rrg * delp[o3] / delz[c3] * tv));
          }
        }
        pkz[c3] = pz;
        double tnew = tv;
        if (p.last_step == 2) {                 // the energy fixer follows: T_v / T_m stays, fv3_remap_finish converts (:793-821)
        } else if (p.last_step) {               // :793-821 (dtmp = 0)
          if (!p.hydrostatic && p.use_cond) {   // :806-811
            double qc;
            const double cvm = moist_cv(p, q + o3, nA * km, qc);
            tnew = (tnew + 0. / cvm * pz) / ((1. + p.r_vir * q[(size_t)(p.sphum - 1) * nA * km + o3]) * (1. - qc));
          } else if (!p.adiabatic) {
            const double qv = p.sphum > 0 ? q[(size_t)(p.sphum - 1) * nA * km + o3] : 0.;
            tnew = (tnew + 0. / (p.hydrostatic ? p.cp : p.cv_air) * pz) / (1. + p.r_vir * qv);
          }
        } else {
          tnew = tnew / pz;                     // :833-841
        }
        pt[o3] = tnew;
      }
    }
  }
};

struct RemapCoords {
  Grid g;
  int km;
  RemapPar p;
  const double *ak, *bk, *pe, *peln;
  double *ps;
  double *pe1p, *pe2p, *pe1l, *pe2l;  // slabs: pressure coordinates; log-pressure coordinates (kord_tm < 0)
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int ncol = g.nx * g.ny;
    const size_t nA = g.nA();
    FV3_COL_FOR2(col, ncol) {
      const int i = g.is + col % g.nx, j = g.js + col / g.nx;
      const int o = g.iA(i, j);
      ColScr c{};
      scr_col(c, col, o, km, nA, p.scr_blocked);
      const size_t peb = (size_t)(j - (g.js - 1)) * (g.nx + 2) * (km + 1) + (i - (g.is - 1));
      const size_t lnb = (size_t)(j - g.js) * g.nx * (km + 1) + (i - g.is);
      const double psfc = pe[peb + (size_t)km * (g.nx + 2)];
      ps[o] = psfc;  // :298-300
      for (int k = 1; k <= km + 1; k++) {
        const size_t so = (size_t)(k - 1) * c.ls + c.o;
        const double pe2k = (k == 1) ? p.ptop : (k == km + 1 ? psfc : ak[k - 1] + bk[k - 1] * psfc);
        pe1p[so] = pe[peb + (size_t)(k - 1) * (g.nx + 2)];
        pe2p[so] = pe2k;
        if (p.kord_tm < 0 || p.remap_te) {
          // (remap_te, hydrostatic: pkez sets peln(1) = log(ptop), :886-895 -- the value it holds already, dyn_core's geopk)
          const double pl = peln[lnb + (size_t)(k - 1) * g.nx];
          pe1l[so] = pl;
          pe2l[so] = (k == 1 || k == km + 1) ? pl : dlog(pe2k);
        }
      }
    }
  }
};

// one thread per (column, field task); tasks: 0 = T_v / theta_v (+ omega on the last step), w (nonhydrostatic), u, v,
// then the nq tracers.  Task t of this launch uses profile-slab set t - task0.  (The tracers come last so that, with
// moist_kappa, the T_v task -- whose source transform reads the un-remapped tracers through moist_cv -- can be launched
// before them.)
struct RemapFields {
  Grid g;
  int km;
  RemapPar p;
  const double *ak, *bk;
  const int *kord_tr;  // device, nq
  const double *delp, *pk, *delz, *peln, *pe, *ws;
  double *w, *pt, *q, *omga, *u, *v;
  double *pe1p, *pe2p, *pe1l, *pe2l, *pe1u, *pe2u, *pe1v, *pe2v;  // coordinate slabs
  double *sets;                                                     // kSetSlabs profile slabs per task of this launch
  size_t slab;                                                      // doubles per slab
  int task0, nblk;                                                  // first task of this launch; workgroups per task
  int ngrp;                                                         // tracer groups (tasks after T_v, w, u, v): the nq
                                                                    // tracers dealt evenly, at most kGroupMax per group
  static constexpr int kSetSlabs = 7, kGroupMax = 3;

  FV3_HD void operator()(int bxg, int, int, int tid, double *) const {
    const int task = task0 + bxg / nblk, bx = bxg % nblk;
    const size_t nA = g.nA(), nCC = g.nCC();
    double *base = sets + (size_t)(task - task0) * kSetSlabs * slab;
    ColScr c{base, base + slab, base + 2 * slab, base + 3 * slab, base + 4 * slab, pe1p, pe2p, base + 5 * slab, nA, 0};
    const int t_w = p.hydrostatic ? -1 : 1, t_u = p.hydrostatic ? 1 : 2, t_v = t_u + 1;
    const int akt = p.kord_tm < 0 ? -p.kord_tm : p.kord_tm;
    if (task == t_u || task == t_v) {  // D-grid winds: u on (is:ie, js:je+1), v on (is:ie+1, js:je)  (fv_mapz.F90:530-573)
      const int which = task == t_u ? 0 : 1;
      const int wdt = which == 0 ? g.nx : g.nx + 1, hgt = which == 0 ? g.ny + 1 : g.ny;
      const int ncol = wdt * hgt;
      c.pe1 = which == 0 ? pe1u : pe1v;
      c.pe2 = which == 0 ? pe2u : pe2v;
      FV3_COL_FOR2(col, ncol) {
        const int i = g.is + col % wdt, j = g.js + col / wdt;
        scr_col(c, col, g.iA(i, j), km, nA, p.scr_blocked);
        auto PE = [&](int ii, int k, int jj) {
          return pe[(size_t)(jj - (g.js - 1)) * (g.nx + 2) * (km + 1) + (size_t)(k - 1) * (g.nx + 2) + (ii - (g.is - 1))];
        };
        const int i2 = which == 0 ? i : i - 1, j2 = which == 0 ? j - 1 : j;  // the other cell sharing the face
        const double psum = PE(i2, km + 1, j2) + PE(i, km + 1, j);
        for (int k = 1; k <= km + 1; k++) {
          CS(pe1, k) = (k == 1) ? PE(i, 1, j) : 0.5 * (PE(i2, k, j2) + PE(i, k, j));
          const double bkh = 0.5 * bk[k - 1];
          CS(pe2, k) = (which == 1 && k == 1) ? ak[0] : ak[k - 1] + bkh * psum;
        }
        double *f = which == 0 ? u + g.iU(i, j) : v + g.iV(i, j);
        const size_t fs = which == 0 ? g.nU() : g.nV();
        const ProfCfg pc = profile_col(c, km, false, 0., -1, p.kord_mt, 0., [&](int k) { return f[(size_t)(k - 1) * fs]; });
        map_col(c, km, false, pc, [&](int k, double val) { f[(size_t)(k - 1) * fs] = val; });
      }
      return;
    }
    const int ncol = g.nx * g.ny;
    FV3_COL_FOR2(col, ncol) {
      const int i = g.is + col % g.nx, j = g.js + col / g.nx;
      const int fo = g.iA(i, j);
      scr_col(c, col, fo, km, nA, p.scr_blocked);
      const int occ = g.iCC(i, j);
      if (task == 0) {
        const double k1k = p.rdgas / p.cv_air, rrg = -p.rdgas / p.grav, akap = p.akap;
        const size_t lnb = (size_t)(j - g.js) * g.nx * (km + 1) + (i - g.is);
        auto PELN = [&](int k) { return peln[lnb + (size_t)(k - 1) * g.nx]; };
        // temperature transform (:200-229), level by level as the profile sweep fetches the field
        auto src_pt = [&](int k) {
          double t = pt[(size_t)(k - 1) * nA + fo];
          if (p.kord_tm < 0) {
            if (p.hydrostatic) {
              t = t * (pk[(size_t)k * nCC + occ] - pk[(size_t)(k - 1) * nCC + occ]) / (akap * (PELN(k + 1) - PELN(k)));
            } else {
              const double dpo = delp[(size_t)(k - 1) * nA + fo];
              if (p.moist_kappa) {  // :212-219
                const size_t o3 = (size_t)(k - 1) * nA + fo;
                double qc;
                const double cvm = moist_cv(p, q + o3, nA * km, qc);
                const double cap = p.rdgas / (p.rdgas + cvm / (1. + p.r_vir * q[(size_t)(p.sphum - 1) * nA * km + o3]));
                p.q_con[o3] = qc;
                p.cappa[o3] = cap;
                t = t * dexp(cap / (1. - cap) * dlog(rrg * dpo / delz[(size_t)(k - 1) * nCC + occ] * t));
              } else {
                t = t * dexp(k1k * dlog(rrg * dpo / delz[(size_t)(k - 1) * nCC + occ] * t));
              }
            }
          }
          return t;
        };
        // remap T_v (log-p coordinate, :363-368) or theta_v (:370-374) -- or the total energy (:348-360)
        ProfCfg pc;
        if (p.remap_te) {
          double *te = p.te + fo;
          if (p.kord_tm == 0) {
            map1_cubic_te_col(c, km, te, nA);
          } else {
            c.pe1 = pe1l;
            c.pe2 = pe2l;
            pc = profile_col(c, km, true, 0., 1, akt, p.cp * p.t_min, [&](int k) { return te[(size_t)(k - 1) * nA]; });
            map_col(c, km, false, pc, [&](int k, double v_) { te[(size_t)(k - 1) * nA] = v_; });
          }
        } else {
        if (p.kord_tm < 0) {
          c.pe1 = pe1l;
          c.pe2 = pe2l;
          pc = profile_col(c, km, true, 0., 1, akt, p.t_min, src_pt);
        } else {
          pc = profile_col(c, km, false, 0., 1, akt, 0., src_pt);
        }
        map_col(c, km, false, pc, [&](int k, double v_) { pt[(size_t)(k - 1) * nA + fo] = v_; });
        }
        // omega (:432-443, :506-526): interpolated in the old log-p coordinate
        if (p.last_step) {
          const size_t peb = (size_t)(j - (g.js - 1)) * (g.nx + 2) * (km + 1) + (i - (g.is - 1));
          const double psfc = pe[peb + (size_t)km * (g.nx + 2)];
          CS(gam, 1) = 0.;
          for (int k = 2; k <= km + 1; k++) CS(gam, k) = omga[(size_t)(k - 2) * nA + fo];  // pe3
          int k_next = 1;
          for (int n = 1; n <= km; n++) {
            const double pn_t = (n == 1) ? PELN(1) : dlog(ak[n - 1] + bk[n - 1] * psfc);
            const double pn_b = (n + 1 == km + 1) ? PELN(km + 1) : dlog(ak[n] + bk[n] * psfc);
            const double mid = 0.5 * (pn_t + pn_b);
            for (int k = k_next; k <= km; k++) {
              const double e0 = PELN(k), e1 = PELN(k + 1);
              if (mid <= e1 && mid >= e0) {
                omga[(size_t)(n - 1) * nA + fo] = CS(gam, k) + (CS(gam, k + 1) - CS(gam, k)) * (mid - e0) / (e1 - e0);
                k_next = k;
                break;
              }
            }
          }
        }
      } else if (task == t_w) {  // w (:400-411)
        const ProfCfg pc = profile_col(c, km, false, ws[occ], -2, p.kord_wz, 0., [&](int k) { return w[(size_t)(k - 1) * nA + fo]; });
        map_col(c, km, false, pc, [&](int k, double v_) { w[(size_t)(k - 1) * nA + fo] = v_; });
      } else {  // constituents (:380-397): the tracers of group grp
        const int grp = task - (t_v + 1);
        const int gb = p.nq / ngrp, gr = p.nq % ngrp;
        const int nl = gb + (grp < gr ? 1 : 0), iq0 = grp * gb + (grp < gr ? grp : gr);
        bool side_by_side = nl > 1;
        for (int n = 0; n < nl; n++) {
          const int a = kord_tr[iq0 + n] < 0 ? -kord_tr[iq0 + n] : kord_tr[iq0 + n];
          if (a == 11 || a == 12 || kord_is_ppm(kord_tr[iq0 + n])) side_by_side = false;  // these keep a2, a3, a4 in slabs
        }
        if (side_by_side) {
          ColScr cg = c;
          cg.gam = base + 6 * slab;
          if (nl == 2) {
            TrcGroup<2> t;
            for (int n = 0; n < 2; n++) {
              t.a1[n] = base + (size_t)(2 * n) * slab;
              t.q[n] = base + (size_t)(2 * n + 1) * slab;
              t.f[n] = q + (size_t)(iq0 + n) * nA * km + fo;
              t.ak[n] = kord_tr[iq0 + n] < 0 ? -kord_tr[iq0 + n] : kord_tr[iq0 + n];
            }
            remap_tracers_col<2>(cg, t, nA, km, p.nq > 5);
          } else {
            TrcGroup<3> t;
            for (int n = 0; n < 3; n++) {
              t.a1[n] = base + (size_t)(2 * n) * slab;
              t.q[n] = base + (size_t)(2 * n + 1) * slab;
              t.f[n] = q + (size_t)(iq0 + n) * nA * km + fo;
              t.ak[n] = kord_tr[iq0 + n] < 0 ? -kord_tr[iq0 + n] : kord_tr[iq0 + n];
            }
            remap_tracers_col<3>(cg, t, nA, km, p.nq > 5);
          }
        }
        for (int n = 0; n < nl; n++) {
          const int iq = iq0 + n;
          double *qq = q + (size_t)iq * nA * km;
          if (!side_by_side) {
            const ProfCfg pc = profile_col(c, km, true, 0., 0, kord_tr[iq], 0., [&](int k) { return qq[(size_t)(k - 1) * nA + fo]; });
            map_col(c, km, p.nq > 5, pc, [&](int k, double v_) { qq[(size_t)(k - 1) * nA + fo] = v_; });
          }
          if (p.fill)  // fv_operators.F90:337 / fv_mapz.F90:390
            fillz_col(km, qq + fo, nA, [&](int k) { return CS(pe2, k + 1) - CS(pe2, k); });
        }
      }
    }
  }
};

// delz (:292, :412-423), then delp, pk, peln, pkz, pt of the column (:318-322, :426-430, :445-503, :793-841)
struct RemapDelzFinal {
  Grid g;
  int km;
  RemapPar p;
  double *delp, *pkz, *pk, *delz, *pt, *peln;
  const double *q;
  ColScr s;  // profile slabs of set 0 with the pressure coordinates
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int ncol = g.nx * g.ny;
    const size_t nA = g.nA(), nCC = g.nCC();
    const double k1k = p.rdgas / p.cv_air, rrg = -p.rdgas / p.grav, akap = p.akap;
    const int akt = p.kord_tm < 0 ? -p.kord_tm : p.kord_tm;
    FV3_COL_FOR2(col, ncol) {
      const int i = g.is + col % g.nx, j = g.js + col / g.nx;
      ColScr c = s;
      const int fo = g.iA(i, j);
      scr_col(c, col, fo, km, nA, p.scr_blocked);
      const int occ = g.iCC(i, j);
      const size_t lnb = (size_t)(j - g.js) * g.nx * (km + 1) + (i - g.is);
      auto PELN = [&](int k) -> double & { return peln[lnb + (size_t)(k - 1) * g.nx]; };
      if (!p.hydrostatic) {
        const ProfCfg pc = profile_col(c, km, false, 0., 1, akt, 0., [&](int k) {
          return -delz[(size_t)(k - 1) * nCC + occ] / delp[(size_t)(k - 1) * nA + fo];  // :292
        });
        map_col(c, km, false, pc, [&](int k, double v_) {
          delz[(size_t)(k - 1) * nCC + occ] = -v_ * (CS(pe2, k + 1) - CS(pe2, k));
        });
      }
      double pn_prev = PELN(1), pk_prev = pk[occ];
      for (int k = 1; k <= km; k++) {
        const double dp2 = CS(pe2, k + 1) - CS(pe2, k);
        delp[(size_t)(k - 1) * nA + fo] = dp2;
        double pn_next, pk_next;
        if (k + 1 == km + 1) {
          pn_next = PELN(km + 1);
          pk_next = pk[(size_t)km * nCC + occ];
        } else {
          pn_next = dlog(CS(pe2, k + 1));
          pk_next = dexp(akap * pn_next);
          PELN(k + 1) = pn_next;
          pk[(size_t)k * nCC + occ] = pk_next;
        }
        if (p.remap_te) {   // T_v, pkz and the conversion of pt follow from the remapped energy (RemapTePost)
          pn_prev = pn_next;
          pk_prev = pk_next;
          continue;
        }
        double pkzv;
        const double tv = pt[(size_t)(k - 1) * nA + fo];
        if (p.hydrostatic)
          pkzv = (pk_next - pk_prev) / (akap * (pn_next - pn_prev));
        else if (p.moist_kappa) {  // :463-478: q holds the remapped tracers
          const size_t o3 = (size_t)(k - 1) * nA + fo;
          double qc;
          const double cvm = moist_cv(p, q + o3, nA * km, qc);
          const double cap = p.rdgas / (p.rdgas + cvm / (1. + p.r_vir * q[(size_t)(p.sphum - 1) * nA * km + o3]));
          p.q_con[o3] = qc;
          p.cappa[o3] = cap;
          pkzv = dexp((p.kord_tm < 0 ? cap : cap / (1. - cap)) * dlog(rrg * dp2 / delz[(size_t)(k - 1) * nCC + occ] * tv));
        } else if (p.kord_tm < 0)
          pkzv = dexp(akap * dlog(rrg * dp2 / delz[(size_t)(k - 1) * nCC + occ] * tv));
        else
          pkzv = dexp(k1k * dlog(rrg * dp2 / delz[(size_t)(k - 1) * nCC + occ] * tv));
        pkz[(size_t)(k - 1) * nCC + occ] = pkzv;
        double tnew = tv;
        if (p.kord_tm > 0) tnew = tnew * pkzv;  // :496-502
        if (p.last_step == 2) {                 // the energy fixer follows: T_v / T_m stays, fv3_remap_finish converts (:793-821)
        } else if (p.last_step) {               // :793-821 (dtmp = 0)
          if (!p.hydrostatic && p.use_cond) {   // :806-811
            const size_t o3 = (size_t)(k - 1) * nA + fo;
            double qc;
            const double cvm = moist_cv(p, q + o3, nA * km, qc);
            tnew = (tnew + 0. / cvm * pkzv) / ((1. + p.r_vir * q[(size_t)(p.sphum - 1) * nA * km + o3]) * (1. - qc));
          } else if (!p.adiabatic) {
            const double qv = p.sphum > 0 ? q[(size_t)(p.sphum - 1) * nA * km + (size_t)(k - 1) * nA + fo] : 0.;
            tnew = (tnew + 0. / (p.hydrostatic ? p.cp : p.cv_air) * pkzv) / (1. + p.r_vir * qv);
          }
        } else {
          tnew = tnew / pkzv;                   // :833-841
        }
        pt[(size_t)(k - 1) * nA + fo] = tnew;
        pn_prev = pn_next;
        pk_prev = pk_next;
      }
    }
  }
};

// pe(i,k,j) = pe2(i,k) for k = 2..km (fv_mapz.F90:624-641)
struct RemapPe {
  Grid g;
  int km;
  const double *ak, *bk;
  double *pe;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int ncol = g.nx * g.ny;
    FV3_COL_FOR2(col, ncol) {
      const int i = g.is + col % g.nx, j = g.js + col / g.nx;
      const size_t peb = (size_t)(j - (g.js - 1)) * (g.nx + 2) * (km + 1) + (i - (g.is - 1));
      const double psfc = pe[peb + (size_t)km * (g.nx + 2)];
      for (int k = 2; k <= km; k++) pe[peb + (size_t)(k - 1) * (g.nx + 2)] = ak[k - 1] + bk[k - 1] * psfc;
    }
  }
};

#undef CS
// ---- total energy and the energy fixer of the last remap (consv_te) ----------------------------------------------------------
// kinetic-energy bracket of a cell from the D-grid winds on its four edges (fv_thermodynamics.F90:152-155, fv_mapz.F90:679-681)
FV3_HD double te_wind_bracket(const Grid &g, const double *u, const double *v, int i, int j, int k) {
  const double u0 = u[(size_t)(k - 1) * g.nU() + g.iU(i, j)], u1 = u[(size_t)(k - 1) * g.nU() + g.iU(i, j + 1)];
  const double v0 = v[(size_t)(k - 1) * g.nV() + g.iV(i, j)], v1 = v[(size_t)(k - 1) * g.nV() + g.iV(i + 1, j)];
  return u0 * u0 + u1 * u1 + v0 * v0 + v1 * v1 - (u0 + u1) * (v0 + v1) * g.cosa_s[g.iA(i, j)];
}

// compute_total_energy (fv_thermodynamics.F90:90-225; called at fv_dynamics.F90:345 with pt = T, qc = zvir*q(sphum)): te_2d of
// every column.  moist_cvm: the moist_phys .and. moist_kappa branch (cvm from moist_cv, :186-193).  phiz (A x (km+1)): scratch.
struct TotalEnergy {
  Grid g;
  int km;
  RemapPar p;
  int moist_cvm;
  const double *u, *v, *w, *delz, *pt, *delp, *q, *qc, *pe, *peln, *hs;
  double *te_2d, *phiz;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int ncol = g.nx * g.ny;
    const size_t nA = g.nA(), nCC = g.nCC();
    FV3_COL_FOR2(col, ncol) {
      const int i = g.is + col % g.nx, j = g.js + col / g.nx;
      const int o = g.iA(i, j), occ = g.iCC(i, j);
      const double rs2 = g.rsin2[o];
      auto PE = [&](int k) { return pe[(size_t)(j - (g.js - 1)) * (g.nx + 2) * (km + 1) + (size_t)(k - 1) * (g.nx + 2) + (i - (g.is - 1))]; };
      auto PELN = [&](int k) { return peln[(size_t)(j - g.js) * g.nx * (km + 1) + (size_t)(k - 1) * g.nx + (i - g.is)]; };
      // qc = zvir * q(sphum) (fv_dynamics.F90:295-301): the caller's array, or formed here from the tracer
      auto QC = [&](size_t o3) { return qc ? qc[o3] : ((p.sphum > 0 && !p.adiabatic && q) ? p.r_vir * q[(size_t)(p.sphum - 1) * nA * km + o3] : 0.); };
      double te;
      if (p.hydrostatic) {
        double ph = hs[o];
        for (int k = km; k >= 1; k--) {
          const size_t o3 = (size_t)(k - 1) * nA + o;
          const double tv = pt[o3] * (1. + QC(o3));
          ph = ph + p.rdgas * tv * (PELN(k + 1) - PELN(k));
        }
        te = PE(km + 1) * hs[o] - PE(1) * ph;
        for (int k = 1; k <= km; k++) {
          const size_t o3 = (size_t)(k - 1) * nA + o;
          const double tv = pt[o3] * (1. + QC(o3));
          te = te + delp[o3] * (p.cp * tv + 0.25 * rs2 * te_wind_bracket(g, u, v, i, j, k));
        }
      } else {
        double ph = hs[o];
        phiz[(size_t)km * nA + o] = ph;
        for (int k = km; k >= 1; k--) {
          ph = ph - p.grav * delz[(size_t)(k - 1) * nCC + occ];
          phiz[(size_t)(k - 1) * nA + o] = ph;
        }
        te = 0.;
        for (int k = 1; k <= km; k++) {
          const size_t o3 = (size_t)(k - 1) * nA + o;
          double cv = p.cv_air;
          if (moist_cvm) {
            double qd;
            cv = moist_cv(p, q + o3, nA * km, qd);
          }
          const double ww = w[o3];
          te = te + delp[o3] * (cv * pt[o3] + 0.5 * (phiz[o3] + phiz[o3 + nA] + ww * ww + 0.5 * rs2 * te_wind_bracket(g, u, v, i, j, k)));
        }
      }
      te_2d[occ] = te;
    }
  }
};

// The energy fixer of the last remap, fv_mapz.F90:647-734 (consv > consv_min, remap_te = .false.): te_2d := te0_2d - (total energy
// of the remapped column), zsum1 = sum(pkz*delp), zsum0 = ptop*(pk(1) - pk(km+1)) + zsum1 (hydrostatic); pt is T_v / T_m here.
// With use_cond q_con is rewritten from moist_cv (:701).  only_sums: the consv < -consv_min branch (:745-763).
struct EnergyFixerSums {
  Grid g;
  int km;
  RemapPar p;
  int only_sums;
  const double *u, *v, *w, *delz, *pt, *delp, *q, *pe, *peln, *hs, *pkz, *pk, *te0_2d;
  double *te_2d, *zsum1, *zsum0, *phiz;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int ncol = g.nx * g.ny;
    const size_t nA = g.nA(), nCC = g.nCC();
    FV3_COL_FOR2(col, ncol) {
      const int i = g.is + col % g.nx, j = g.js + col / g.nx;
      const int o = g.iA(i, j), occ = g.iCC(i, j);
      const double rs2 = g.rsin2[o];
      auto PE = [&](int k) { return pe[(size_t)(j - (g.js - 1)) * (g.nx + 2) * (km + 1) + (size_t)(k - 1) * (g.nx + 2) + (i - (g.is - 1))]; };
      auto PELN = [&](int k) { return peln[(size_t)(j - g.js) * g.nx * (km + 1) + (size_t)(k - 1) * g.nx + (i - g.is)]; };
      if (!only_sums) {
        double te;
        if (p.remap_te) {   // :655-663
          te = p.te[o] * delp[o];
          for (int k = 2; k <= km; k++) te = te + p.te[(size_t)(k - 1) * nA + o] * delp[(size_t)(k - 1) * nA + o];
        } else if (p.hydrostatic) {
          double gz = hs[o];
          for (int k = 1; k <= km; k++) gz = gz + p.rdgas * pt[(size_t)(k - 1) * nA + o] * (PELN(k + 1) - PELN(k));
          te = PE(km + 1) * hs[o] - PE(1) * gz;
          for (int k = 1; k <= km; k++) {
            const size_t o3 = (size_t)(k - 1) * nA + o;
            te = te + delp[o3] * (p.cp * pt[o3] + 0.25 * rs2 * te_wind_bracket(g, u, v, i, j, k));
          }
        } else {
          double ph = hs[o];
          phiz[(size_t)km * nA + o] = ph;
          for (int k = km; k >= 1; k--) {
            ph = ph - p.grav * delz[(size_t)(k - 1) * nCC + occ];
            phiz[(size_t)(k - 1) * nA + o] = ph;
          }
          te = 0.;
          for (int k = 1; k <= km; k++) {
            const size_t o3 = (size_t)(k - 1) * nA + o;
            const double qv = (p.sphum > 0 && !p.adiabatic) ? q[(size_t)(p.sphum - 1) * nA * km + o3] : 0.;   // adiabatic: the caller's zvir = 0
            const double ww = w[o3];
            const double mech = 0.5 * (phiz[o3] + phiz[o3 + nA] + ww * ww + 0.5 * rs2 * te_wind_bracket(g, u, v, i, j, k));
            if (p.use_cond) {
              double qc;
              const double cvm = moist_cv(p, q + o3, nA * km, qc);
              p.q_con[o3] = qc;
              te = te + delp[o3] * (cvm * pt[o3] / ((1. + p.r_vir * qv) * (1. - qc)) + mech);
            } else {
              te = te + delp[o3] * (p.cv_air * pt[o3] / (1. + p.r_vir * qv) + mech);
            }
          }
        }
        te_2d[occ] = te0_2d[occ] - te;
      }
      double z1 = pkz[occ] * delp[o];
      for (int k = 2; k <= km; k++) z1 = z1 + pkz[(size_t)(k - 1) * nCC + occ] * delp[(size_t)(k - 1) * nA + o];
      zsum1[occ] = z1;
      if (p.hydrostatic) zsum0[occ] = p.ptop * (pk[occ] - pk[(size_t)km * nCC + occ]) + z1;
    }
  }
};

// step 9a of Lagrangian_to_Eulerian with the increment of the energy fixer (fv_mapz.F90:793-821): T_v / T_m -> T
struct RemapFinish {
  Grid g;
  int km;
  RemapPar p;
  double dtmp;
  const double *q, *pkz;
  double *pt;
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const int n = g.nx * g.ny;
    const size_t nA = g.nA(), nCC = g.nCC();
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH && idx < n; idx += kNT) {
      const int i = g.is + idx % g.nx, j = g.js + idx / g.nx;
      const size_t o3 = (size_t)bz * nA + g.iA(i, j), c3 = (size_t)bz * nCC + g.iCC(i, j);
      const double qv = (p.sphum > 0 && !p.adiabatic) ? q[(size_t)(p.sphum - 1) * nA * km + o3] : 0.;   // adiabatic: the caller's zvir = 0
      if (p.hydrostatic) {
        pt[o3] = (pt[o3] + dtmp / p.cp * pkz[c3]) / (1. + p.r_vir * qv);
      } else if (p.use_cond) {
        double qc;
        const double cvm = moist_cv(p, q + o3, nA * km, qc);
        pt[o3] = (pt[o3] + dtmp / cvm * pkz[c3]) / ((1. + p.r_vir * qv) * (1. - qc));
      } else if (!p.adiabatic) {
        pt[o3] = (pt[o3] + dtmp / p.cv_air * pkz[c3]) / (1. + p.r_vir * qv);
      }
    }
  }
};

}  // namespace fv3
