/*
 * oracle/dyn_pre.c -- CPU oracle (test infrastructure, see fvo.h) for the pieces of fv_dynamics around the
 * k_split loop that touch the prognostic state:
 *   cubed_to_latlon   tools-free part of model/fv_grid_utils.F90:2319-2561: c2l_ord2 (:2526-2558) and c2l_ord4 (:2384-2475),
 *                     the "simple Cartesian geometry" branches (grid_type = 4) and the cubed-sphere ones (one tile per face)
 *   Rayleigh_Friction model/fv_dynamics.F90:1126-1264 (the branch fv_dynamics takes for grid_type = 4,
 *                     :368-376), split at the halo update of u2f (:1207-1209) that the caller performs
 * Plain IEEE evaluation of the reference's expressions, no FMA contraction.
 */
#include "fvo.h"

#include <math.h>
#include <stdlib.h>

#define BOUNDS(g)                                                                         \
  const int is = (g)->is, ie = (g)->ie, js = (g)->js, je = (g)->je;                       \
  const int isd = (g)->isd, ied = (g)->ied, jsd = (g)->jsd, jed = (g)->jed;               \
  const int nid = ied - isd + 1, njd = jed - jsd + 1, nx = ie - is + 1, ny = je - js + 1; \
  (void)nx; (void)ny; (void)ied; (void)jed
#define A3(i, j, k) ((size_t)((k)-1) * nid * njd + (size_t)((j)-jsd) * nid + ((i)-isd))
#define U3(i, j, k) ((size_t)((k)-1) * nid * (njd + 1) + (size_t)((j)-jsd) * nid + ((i)-isd))
#define V3(i, j, k) ((size_t)((k)-1) * (nid + 1) * njd + (size_t)((j)-jsd) * (nid + 1) + ((i)-isd))
#define CC3(i, j, k) ((size_t)((k)-1) * nx * ny + (size_t)((j)-js) * nx + ((i)-is))
#define IA(i, j) ((size_t)((j)-jsd) * nid + ((i)-isd))
#define IU(i, j) ((size_t)((j)-jsd) * nid + ((i)-isd))
#define IV(i, j) ((size_t)((j)-jsd) * (nid + 1) + ((i)-isd))

/* cubed_to_latlon.  ord 2: c2l_ord2 (:2551-2558); ord 4: c2l_ord4 (:2468-2475), which needs the
 * halo of u, v up to date (the reference's mode > 0 update, :2372-2376, is the caller's).  u: U x km, v: V x km,
 * ua, va: A x km (written on is:ie, js:je). */
int fvo_c2l(const fvo_grid *g, int km, int ord, const double *u, const double *v, double *ua, double *va) {
  BOUNDS(g);
  const double a1 = 0.5625, a2 = -0.0625;
  int i, j, k;
  if (g->grid_type == 3) return FVO_ERR_UNSUPPORTED;
  if (g->grid_type < 3) { /* a face of the cubed sphere: fv_grid_utils.F90:2384-2466 (ord 4), :2526-2546 (ord 2) */
    const double c1 = 1.125, c2 = -0.125;
    const int npx = g->npx, npy = g->npy;
    if (!g->a11) return FVO_ERR_UNSUPPORTED;
    for (k = 1; k <= km; k++)
      for (j = js; j <= je; j++)
        for (i = is; i <= ie; i++) {
          double ut, vt;
          if (ord == 2 || i == 1 || i == npx - 1 || j == 1 || j == npy - 1) {
            ut = 2. * (u[U3(i, j, k)] * g->dx[IU(i, j)] + u[U3(i, j + 1, k)] * g->dx[IU(i, j + 1)]) / (g->dx[IU(i, j)] + g->dx[IU(i, j + 1)]);
            vt = 2. * (v[V3(i, j, k)] * g->dy[IV(i, j)] + v[V3(i + 1, j, k)] * g->dy[IV(i + 1, j)]) / (g->dy[IV(i, j)] + g->dy[IV(i + 1, j)]);
          } else {
            ut = c2 * (u[U3(i, j - 1, k)] + u[U3(i, j + 2, k)]) + c1 * (u[U3(i, j, k)] + u[U3(i, j + 1, k)]);
            vt = c2 * (v[V3(i - 1, j, k)] + v[V3(i + 2, j, k)]) + c1 * (v[V3(i, j, k)] + v[V3(i + 1, j, k)]);
          }
          ua[A3(i, j, k)] = g->a11[IA(i, j)] * ut + g->a12[IA(i, j)] * vt;
          va[A3(i, j, k)] = g->a21[IA(i, j)] * ut + g->a22[IA(i, j)] * vt;
        }
    return FVO_OK;
  }
  for (k = 1; k <= km; k++)
    for (j = js; j <= je; j++)
      for (i = is; i <= ie; i++) {
        if (ord == 2) {
          ua[A3(i, j, k)] = 0.5 * (u[U3(i, j, k)] + u[U3(i, j + 1, k)]);
          va[A3(i, j, k)] = 0.5 * (v[V3(i, j, k)] + v[V3(i + 1, j, k)]);
        } else {
          ua[A3(i, j, k)] = a2 * (u[U3(i, j - 1, k)] + u[U3(i, j + 2, k)]) + a1 * (u[U3(i, j, k)] + u[U3(i, j + 1, k)]);
          va[A3(i, j, k)] = a2 * (v[V3(i - 1, j, k)] + v[V3(i + 2, j, k)]) + a1 * (v[V3(i, j, k)] + v[V3(i + 1, j, k)]);
        }
      }
  return FVO_OK;
}

/* the damping profile rf(k) and kmax (fv_dynamics.F90:1169-1182); pm: layer-mean pressure (npz).  Returns kmax. */
int fvo_rayleigh_rf(int npz, double dt, double tau, double rf_cutoff, double ptop, const double *pm, double *rf) {
  const double sday = 86400., pi = 3.1415926535897931; /* constants_mod PI */
  int k, kmax = 0;
  for (k = 1; k <= npz; k++) {
    if (pm[k - 1] < rf_cutoff) {
      const double s = sin(0.5 * pi * log(rf_cutoff / pm[k - 1]) / log(rf_cutoff / ptop));
      rf[k - 1] = dt / (tau * sday) * (s * s);
      kmax = k;
    } else {
      break;
    }
  }
  return kmax;
}

/* :1186-1205: A-grid winds (c2l_ord2) and the squared wind speed u2f on is:ie, js:je for k <= kmax.  u2f: A x kmax. */
int fvo_rayleigh_u2f(const fvo_grid *g, int kmax, int hydrostatic, const double *u, const double *v, const double *w,
                     double *ua, double *va, double *u2f) {
  BOUNDS(g);
  int i, j, k, rc;
  if ((rc = fvo_c2l(g, kmax, 2, u, v, ua, va)) != FVO_OK) return rc;
  for (k = 1; k <= kmax; k++)
    for (j = js; j <= je; j++)
      for (i = is; i <= ie; i++) {
        const double a = ua[A3(i, j, k)], b = va[A3(i, j, k)];
        u2f[A3(i, j, k)] = hydrostatic ? a * a + b * b : a * a + b * b + w[A3(i, j, k)] * w[A3(i, j, k)];
      }
  return FVO_OK;
}

/* Rayleigh_Super, fv_dynamics.F90:953-1124, after cubed_to_latlon (:1040): u2f(:,:,k) = 1/(1+rf(k)) (its halo update moves
 * a constant), heating if conserve, scaling of u, v, w; u00 / v00 (is_ideal_case): relaxation towards the t = 0 winds. */
int fvo_rayleigh_super(const fvo_grid *g, int kmax, int conserve, int hydrostatic, double cp, double rg, double ptop,
                       const double *pm, const double *rf, const double *ua, const double *va, double *pt, double *u,
                       double *v, double *w, const double *u00, const double *v00) {
  BOUNDS(g);
  const double rcv = 1. / (cp - rg);
  int i, j, k;
  for (k = 1; k <= kmax; k++) {
    const double rfk = rf[k - 1], u2f = 1. / (1. + rfk);
    if (u00) { /* :1064-1081 */
      if (!hydrostatic)
        for (j = js; j <= je; j++)
          for (i = is; i <= ie; i++) w[A3(i, j, k)] = w[A3(i, j, k)] / (1. + rfk);
      for (j = js; j <= je + 1; j++)
        for (i = is; i <= ie; i++) u[U3(i, j, k)] = (u[U3(i, j, k)] + rfk * u00[U3(i, j, k)]) / (1. + rfk);
      for (j = js; j <= je; j++)
        for (i = is; i <= ie + 1; i++) v[V3(i, j, k)] = (v[V3(i, j, k)] + rfk * v00[V3(i, j, k)]) / (1. + rfk);
      continue;
    }
    if (conserve) { /* :1084-1098 */
      for (j = js; j <= je; j++)
        for (i = is; i <= ie; i++) {
          const double a = ua[A3(i, j, k)], b = va[A3(i, j, k)];
          if (hydrostatic)
            pt[A3(i, j, k)] = pt[A3(i, j, k)] + 0.5 * (a * a + b * b) * (1. - u2f * u2f) / (cp - rg * ptop / pm[k - 1]);
          else
            pt[A3(i, j, k)] = pt[A3(i, j, k)] + 0.5 * (a * a + b * b + w[A3(i, j, k)] * w[A3(i, j, k)]) * (1. - u2f * u2f) * rcv;
        }
    }
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++) u[U3(i, j, k)] = 0.5 * (u2f + u2f) * u[U3(i, j, k)];
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++) v[V3(i, j, k)] = 0.5 * (u2f + u2f) * v[V3(i, j, k)];
    if (!hydrostatic)
      for (j = js; j <= je; j++)
        for (i = is; i <= ie; i++) w[A3(i, j, k)] = u2f * w[A3(i, j, k)];
  }
  return FVO_OK;
}

/* :1211-1260 with the halo of u2f filled by the caller: frictional heating (conserve) and the implicit damping of
 * u, v, w.  u2f is overwritten with rf*sqrt(u2f/u000) on is-1:ie+1, js-1:je+1 as in the reference. */
int fvo_rayleigh_apply(const fvo_grid *g, int kmax, int conserve, int hydrostatic, double cp, double rg, double ptop,
                       const double *pm, const double *rf, double *u2f, double *pt, double *delz, double *u, double *v,
                       double *w) {
  BOUNDS(g);
  const double u000 = 4900., rcv = 1. / (cp - rg);
  int i, j, k;
  for (k = 1; k <= kmax; k++) {
    const double rfk = rf[k - 1];
    if (conserve) {
      for (j = js; j <= je; j++)
        for (i = is; i <= ie; i++) {
          const double x = u2f[A3(i, j, k)];
          const double d = 1. + rfk * sqrt(x / u000);
          if (hydrostatic) {
            pt[A3(i, j, k)] = pt[A3(i, j, k)] + 0.5 * x / (cp - rg * ptop / pm[k - 1]) * (1. - 1. / (d * d));
          } else {
            delz[CC3(i, j, k)] = delz[CC3(i, j, k)] / pt[A3(i, j, k)];
            pt[A3(i, j, k)] = pt[A3(i, j, k)] + 0.5 * x * rcv * (1. - 1. / (d * d));
            delz[CC3(i, j, k)] = delz[CC3(i, j, k)] * pt[A3(i, j, k)];
          }
        }
    }
    for (j = js - 1; j <= je + 1; j++)
      for (i = is - 1; i <= ie + 1; i++) u2f[A3(i, j, k)] = rfk * sqrt(u2f[A3(i, j, k)] / u000);
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++) u[U3(i, j, k)] = u[U3(i, j, k)] / (1. + 0.5 * (u2f[A3(i, j - 1, k)] + u2f[A3(i, j, k)]));
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++) v[V3(i, j, k)] = v[V3(i, j, k)] / (1. + 0.5 * (u2f[A3(i - 1, j, k)] + u2f[A3(i, j, k)]));
    if (!hydrostatic)
      for (j = js; j <= je; j++)
        for (i = is; i <= ie; i++) w[A3(i, j, k)] = w[A3(i, j, k)] / (1. + u2f[A3(i, j, k)]);
  }
  return FVO_OK;
}

/* The shared deterministic exp / log (include/fv3_math.h) over arrays: lets the Python side of the tests
 * form pressure powers with the same operation sequence as the kernels under test. */
int fvo_exp_n(const double *x, double *y, long n) {
  for (long i = 0; i < n; i++) y[i] = fv3_exp(x[i]);
  return FVO_OK;
}
int fvo_log_n(const double *x, double *y, long n) {
  for (long i = 0; i < n; i++) y[i] = fv3_log(x[i]);
  return FVO_OK;
}

/* Ray_fast, dyn_core.F90:2485-2601.  fvo_ray_fast_profile: what the routine keeps from its first call (:2519-2545) -- rf(1:npz)
 * (1 / (1 + rff) on the levels with pfull < rf_cutoff, 1 below), k_rf and dm = sum of dp(1:k_rf); returns kmax (the module's initial
 * value 1 when no level is above the cutoff).  dt: abs(dt) of the acoustic step; tau in days; ks, dp = dp_ref as at the call site :1059. */
int fvo_ray_fast_profile(int npz, int ks, double dt, double tau, double rf_cutoff, double ptop, const double *pfull, const double *dp,
                         double *rf, int *k_rf, double *dm_out) {
  const double sday = 86400., pi = 3.1415926535897931;
  const double tau0 = tau * sday;
  int k, kmax = 1;
  double dm = 0.;
  for (k = 0; k < npz; k++) rf[k] = 1.;
  for (k = 1; k <= npz; k++) {
    if (pfull[k - 1] < rf_cutoff) {
      const double s = sin(0.5 * pi * log(rf_cutoff / pfull[k - 1]) / log(rf_cutoff / ptop));
      const double rff = dt / tau0 * (s * s);
      kmax = k;
      rf[k - 1] = 1.0 / (1.0 + rff);
    } else {
      break;
    }
  }
  *k_rf = 0;
  for (k = 1; k <= ks; k++) {
    const double lim = 10. * ptop < 100. ? 10. * ptop : 100.;
    if (pfull[k - 1] < rf_cutoff + lim) {
      dm = dm + dp[k - 1];
      *k_rf = k;
    } else {
      break;
    }
  }
  *dm_out = dm;
  return kmax;
}

/* :2549-2597; u: U x npz, v: V x npz, w: A x npz (NULL when hydrostatic) */
int fvo_ray_fast(const fvo_grid *g, int npz, int kmax, int k_rf, const double *rf, const double *dp, int hydrostatic, double *u,
                 double *v, double *w) {
  BOUNDS(g);
  int i, j, k;
  const int km = npz;
  (void)km;
  for (j = js; j <= je + 1; j++) {
    double dm = 0.;
    double *dmu = (double *)calloc((size_t)(ie - is + 3), sizeof(double)), *dmv = (double *)calloc((size_t)(ie - is + 3), sizeof(double));
    for (k = 1; k <= k_rf; k++) dm = dm + dp[k - 1];
    for (k = 1; k <= kmax; k++) {
      for (i = is; i <= ie; i++) {
        dmu[i - is] = dmu[i - is] + (1. - rf[k - 1]) * dp[k - 1] * u[U3(i, j, k)];
        u[U3(i, j, k)] = rf[k - 1] * u[U3(i, j, k)];
      }
      if (j != je + 1) {
        for (i = is; i <= ie + 1; i++) {
          dmv[i - is] = dmv[i - is] + (1. - rf[k - 1]) * dp[k - 1] * v[V3(i, j, k)];
          v[V3(i, j, k)] = rf[k - 1] * v[V3(i, j, k)];
        }
        if (!hydrostatic)
          for (i = is; i <= ie; i++) w[A3(i, j, k)] = rf[k - 1] * w[A3(i, j, k)];
      }
    }
    for (i = is; i <= ie; i++) dmu[i - is] = dmu[i - is] / dm;
    if (j != je + 1)
      for (i = is; i <= ie + 1; i++) dmv[i - is] = dmv[i - is] / dm;
    for (k = 1; k <= k_rf; k++) {
      for (i = is; i <= ie; i++) u[U3(i, j, k)] = u[U3(i, j, k)] + dmu[i - is];
      if (j != je + 1)
        for (i = is; i <= ie + 1; i++) v[V3(i, j, k)] = v[V3(i, j, k)] + dmv[i - is];
    }
    free(dmu); free(dmv);
  }
  return FVO_OK;
}

/* compute_aam, fv_dynamics.F90:1266-1314, after the caller's cubed_to_latlon(ord 2) (:1287): the mass-integrated atmospheric angular
 * momentum of every column, m_fac = sum dm r^2 and ps.  coslat = cos(agrid(:,:,2)): A (2-D); ua, delp: A x npz; aam, m_fac: CC; ps: A. */
int fvo_compute_aam(const fvo_grid *g, int npz, double radius, double omega, double agrav, double ptop, const double *coslat,
                    const double *ua, const double *delp, double *aam, double *m_fac, double *ps) {
  BOUNDS(g);
  int i, j, k;
  for (j = js; j <= je; j++)
    for (i = is; i <= ie; i++) {
      const double r1 = radius * coslat[(size_t)(j - jsd) * nid + (i - isd)], r2 = r1 * r1;
      double a = 0., m = 0., p = ptop;
      for (k = 1; k <= npz; k++) {
        double dm = delp[A3(i, j, k)];
        p = p + dm;
        dm = dm * agrav;
        a = a + (r2 * omega + r1 * ua[A3(i, j, k)]) * dm;
        m = m + dm * r2;
      }
      aam[(size_t)(j - js) * nx + (i - is)] = a;
      m_fac[(size_t)(j - js) * nx + (i - is)] = m;
      ps[(size_t)(j - jsd) * nid + (i - isd)] = p;
    }
  return FVO_OK;
}

/* consv_am, fv_dynamics.F90:784-798: u += u00 l2c_u on (is:ie, js:je+1), v += u00 l2c_v on (is:ie+1, js:je); l2c_u: U (2-D), l2c_v: V (2-D) */
int fvo_consv_am_apply(const fvo_grid *g, int npz, double u00, const double *l2c_u, const double *l2c_v, double *u, double *v) {
  BOUNDS(g);
  int i, j, k;
  for (k = 1; k <= npz; k++) {
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++) u[U3(i, j, k)] = u[U3(i, j, k)] + u00 * l2c_u[(size_t)(j - jsd) * nid + (i - isd)];
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++) v[V3(i, j, k)] = v[V3(i, j, k)] + u00 * l2c_v[(size_t)(j - jsd) * (nid + 1) + (i - isd)];
  }
  return FVO_OK;
}

/* mix_dp, model/dyn_core.F90:2119-2200 with CG = .false. (the call of :820): loop for loop */
int fvo_mix_dp(const fvo_grid *g, int km, int hydrostatic, const double *ak, const double *bk, double *w, double *delp, double *pt) {
  BOUNDS(g);
  int i, j, k;
  for (j = js; j <= je; j++) {
    for (k = 1; k <= km - 1; k++) {
      const double dpmin = 0.01 * (ak[k] - ak[k - 1] + (bk[k] - bk[k - 1]) * 1.E5);
      for (i = is; i <= ie; i++) {
        if (!(delp[A3(i, j, k)] >= dpmin)) {
          const double dp = dpmin - delp[A3(i, j, k)];
          pt[A3(i, j, k)] = (pt[A3(i, j, k)] * delp[A3(i, j, k)] + pt[A3(i, j, k + 1)] * dp) / dpmin;
          if (!hydrostatic) w[A3(i, j, k)] = (w[A3(i, j, k)] * delp[A3(i, j, k)] + w[A3(i, j, k + 1)] * dp) / dpmin;
          delp[A3(i, j, k)] = dpmin;
          delp[A3(i, j, k + 1)] = delp[A3(i, j, k + 1)] - dp;
        }
      }
    }
    {
      const double dpmin = 0.01 * (ak[km] - ak[km - 1] + (bk[km] - bk[km - 1]) * 1.E5);
      for (i = is; i <= ie; i++) {
        if (!(delp[A3(i, j, km)] >= dpmin)) {
          const double dp = dpmin - delp[A3(i, j, km)];
          pt[A3(i, j, km)] = (pt[A3(i, j, km)] * delp[A3(i, j, km)] + pt[A3(i, j, km - 1)] * dp) / dpmin;
          if (!hydrostatic) w[A3(i, j, km)] = (w[A3(i, j, km)] * delp[A3(i, j, km)] + w[A3(i, j, km - 1)] * dp) / dpmin;
          delp[A3(i, j, km)] = dpmin;
          delp[A3(i, j, km - 1)] = delp[A3(i, j, km - 1)] - dp;
        }
      }
    }
  }
  return 0;
}
