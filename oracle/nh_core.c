/*
 * oracle/nh_core.c -- CPU oracle (test infrastructure, see fvo.h) for the nonhydrostatic column
 * path of the acoustic substep: model/nh_utils.F90 (update_dz_c, update_dz_d, edge_profile,
 * Riem_Solver_c, SIM1_solver, SIM_solver), model/nh_core.F90 (Riem_Solver3) and the pressure
 * gradient / halo-recompute helpers of model/dyn_core.F90 (p_grad_c, nh_p_grad, pk3_halo, pln_halo,
 * pe_halo, geopk).  Branches: use_cond / moist_kappa in the Riemann solvers only (q_con, cappa arguments; .false.
 * elsewhere), fast_tau_w_sec = 0, d2bg_zq = 0,
 * grid_type >= 3.  Physical constants (grav, rdgas, cp_air) come from FMS constants_mod, which is
 * not part of the reference tree; they are passed in by the caller (GFDL defaults: grav=9.80,
 * rdgas=287.04, kappa=2/7, cp_air=rdgas/kappa).
 */
#include "fvo.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static const double r3 = 1. / 3.;
static const double dz_min = 2.; /* nh_utils.F90:49 */

static inline double dmax(double a, double b) { return a > b ? a : b; }
static double *dalloc(size_t n) { return (double *)calloc(n, sizeof(double)); }

/* fast_tau_w_sec > 0: the module state of nh_utils.F90:53-55 (rff, k_rf, set once by Riem_Solver_c's first call, :356-367, and used
 * by SIM1_solver :1363-1371 and SIM_solver :1498-1506 of both Riemann solvers from then on).  fvo_fast_tau_w_rff evaluates the
 * profile (rff: km values; returns k_rf), fvo_set_fast_tau_w installs it (k_rf = 0: fast_tau_w_sec = 0, the default). */
static int g_k_rf = 0;
static double g_rff[512];
int fvo_fast_tau_w_rff(int km, double dt, double fast_tau_w_sec, double rf_cutoff, double ptop, const double *pfull, double *rff) {
  const double pi_8 = 3.14159265358979323846; /* constants_mod pi_8 */
  int k, k_rf = 0;
  if (!(fast_tau_w_sec > 1.e-5)) return 0;
  for (k = 1; k <= km; k++) {
    double s, rff_temp;
    if (pfull[k - 1] > rf_cutoff) break;
    k_rf = k;
    s = sin(0.5 * pi_8 * log(rf_cutoff / pfull[k - 1]) / log(rf_cutoff / ptop));
    rff_temp = dt / fast_tau_w_sec * (s * s);
    rff[k - 1] = 1.0 / (1.0 + rff_temp);
  }
  return k_rf;
}
int fvo_set_fast_tau_w(int k_rf, const double *rff) {
  int k;
  if (k_rf < 0 || k_rf > 512) return FVO_ERR_UNSUPPORTED;
  for (k = 0; k < k_rf; k++) g_rff[k] = rff[k];
  g_k_rf = k_rf;
  return FVO_OK;
}

#define BOUNDS(g)                                                                         \
  const int is = (g)->is, ie = (g)->ie, js = (g)->js, je = (g)->je;                       \
  const int isd = (g)->isd, ied = (g)->ied, jsd = (g)->jsd, jed = (g)->jed;               \
  const int nid = ied - isd + 1, njd = jed - jsd + 1, nx = ie - is + 1, ny = je - js + 1; \
  (void)nid; (void)njd; (void)nx; (void)ny; (void)is; (void)ie; (void)js; (void)je;       \
  (void)isd; (void)ied; (void)jsd; (void)jed
#define IA(i, j) ((size_t)((j)-jsd) * nid + ((i)-isd))
#define IU(i, j) ((size_t)((j)-jsd) * nid + ((i)-isd))
#define IV(i, j) ((size_t)((j)-jsd) * (nid + 1) + ((i)-isd))
#define IB(i, j) ((size_t)((j)-jsd) * (nid + 1) + ((i)-isd))
#define ICX(i, j) ((size_t)((j)-jsd) * (nx + 1) + ((i)-is))
#define ICY(i, j) ((size_t)((j)-js) * nid + ((i)-isd))
#define IFX(i, j) ((size_t)((j)-js) * (nx + 1) + ((i)-is))
#define IFY(i, j) ((size_t)((j)-js) * nx + ((i)-is))
#define ICC(i, j) ((size_t)((j)-js) * nx + ((i)-is))
#define A3(i, j, k) ((size_t)((k)-1) * nid * njd + IA(i, j))

/* update_dz_c, nh_utils.F90:59-201.  ut, vt: A x km; gz: A x (km+1); zs, ws: A. */
int fvo_update_dz_c(const fvo_grid *g, int km, double dt, const double *dp0, const double *zs,
                    const double *ut, const double *vt, double *gz, double *ws) {
  BOUNDS(g);
  int i, j, k;
  const double rdt = 1. / dt;
  const double top_ratio = dp0[0] / (dp0[0] + dp0[1]);
  const double bot_ratio = dp0[km - 1] / (dp0[km - 2] + dp0[km - 1]);
  const int is1 = is - 1, js1 = js - 1, ie1 = ie + 1, je1 = je + 1, ie2 = ie + 2, je2 = je + 2;
  const size_t nA = (size_t)nid * njd;
#pragma omp parallel for private(i, j) schedule(dynamic)
  for (k = 1; k <= km + 1; k++) {
    double *gz2 = dalloc(nA), *xfx = dalloc(nA), *yfx = dalloc(nA), *fx = dalloc(nA), *fy = dalloc(nA);
    if (k == 1) {
      for (j = js1; j <= je1; j++)
        for (i = is1; i <= ie2; i++) xfx[IA(i, j)] = ut[A3(i, j, 1)] + (ut[A3(i, j, 1)] - ut[A3(i, j, 2)]) * top_ratio;
      for (j = js1; j <= je2; j++)
        for (i = is1; i <= ie1; i++) yfx[IA(i, j)] = vt[A3(i, j, 1)] + (vt[A3(i, j, 1)] - vt[A3(i, j, 2)]) * top_ratio;
    } else if (k == km + 1) {
      for (j = js1; j <= je1; j++)
        for (i = is1; i <= ie2; i++)
          xfx[IA(i, j)] = ut[A3(i, j, km)] + (ut[A3(i, j, km)] - ut[A3(i, j, km - 1)]) * bot_ratio;
      for (j = js1; j <= je2; j++)
        for (i = is1; i <= ie1; i++)
          yfx[IA(i, j)] = vt[A3(i, j, km)] + (vt[A3(i, j, km)] - vt[A3(i, j, km - 1)]) * bot_ratio;
    } else {
      const double int_ratio = 1. / (dp0[k - 2] + dp0[k - 1]);
      for (j = js1; j <= je1; j++)
        for (i = is1; i <= ie2; i++)
          xfx[IA(i, j)] = (dp0[k - 1] * ut[A3(i, j, k - 1)] + dp0[k - 2] * ut[A3(i, j, k)]) * int_ratio;
      for (j = js1; j <= je2; j++)
        for (i = is1; i <= ie1; i++)
          yfx[IA(i, j)] = (dp0[k - 1] * vt[A3(i, j, k - 1)] + dp0[k - 2] * vt[A3(i, j, k)]) * int_ratio;
    }
    for (j = jsd; j <= jed; j++)
      for (i = isd; i <= ied; i++) gz2[IA(i, j)] = gz[A3(i, j, k)];
    if (g->grid_type < 3) fvo_fill_4corners(g, gz2, 1); /* :151 */
    for (j = js1; j <= je1; j++)
      for (i = is1; i <= ie2; i++) {
        if (xfx[IA(i, j)] > 0.)
          fx[IA(i, j)] = gz2[IA(i - 1, j)];
        else
          fx[IA(i, j)] = gz2[IA(i, j)];
        fx[IA(i, j)] = xfx[IA(i, j)] * fx[IA(i, j)];
      }
    if (g->grid_type < 3) fvo_fill_4corners(g, gz2, 2); /* :163 */
    for (j = js1; j <= je2; j++)
      for (i = is1; i <= ie1; i++) {
        if (yfx[IA(i, j)] > 0.)
          fy[IA(i, j)] = gz2[IA(i, j - 1)];
        else
          fy[IA(i, j)] = gz2[IA(i, j)];
        fy[IA(i, j)] = yfx[IA(i, j)] * fy[IA(i, j)];
      }
    for (j = js1; j <= je1; j++)
      for (i = is1; i <= ie1; i++)
        gz[A3(i, j, k)] = (gz2[IA(i, j)] * g->area[IA(i, j)] + fx[IA(i, j)] - fx[IA(i + 1, j)] + fy[IA(i, j)] - fy[IA(i, j + 1)]) /
                          (g->area[IA(i, j)] + xfx[IA(i, j)] - xfx[IA(i + 1, j)] + yfx[IA(i, j)] - yfx[IA(i, j + 1)]);
    free(gz2); free(xfx); free(yfx); free(fx); free(fy);
  }
  for (j = js1; j <= je1; j++) {
    for (i = is1; i <= ie1; i++) ws[IA(i, j)] = (zs[IA(i, j)] - gz[A3(i, j, km + 1)]) * rdt;
    for (k = km; k >= 1; k--)
      for (i = is1; i <= ie1; i++) gz[A3(i, j, k)] = dmax(gz[A3(i, j, k)], gz[A3(i, j, k + 1)] + dz_min);
  }
  return FVO_OK;
}

/* SIM1_solver, nh_utils.F90:1277-1394, one column (fast_tau_w_sec: fvo_set_fast_tau_w).  Arrays are 1-based
 * [1..km] / [1..km+1] (element 0 unused). */
static void sim1_column(int km, double dt, double rgas, const double *gm2, const double *cp2, double *pe,
                        const double *dm2, const double *pm2, const double *pem, double *w2, double *dz2,
                        const double *pt2, double ws, double p_fac) {
  int k;
  double *aa = dalloc(km + 2), *bb = dalloc(km + 2), *dd = dalloc(km + 2), *w1 = dalloc(km + 2),
         *g_rat = dalloc(km + 2), *gam = dalloc(km + 2), *pp = dalloc(km + 3);
  double p1, bet;
  const double t1g = 2. * dt * dt, rdt = 1. / dt;
  for (k = 1; k <= km; k++) {
    pe[k] = fv3_exp(gm2[k] * fv3_log(-dm2[k] / dz2[k] * rgas * pt2[k])) - pm2[k];
    w1[k] = w2[k];
  }
  for (k = 1; k <= km - 1; k++) {
    g_rat[k] = dm2[k] / dm2[k + 1];
    bb[k] = 2. * (1. + g_rat[k]);
    dd[k] = 3. * (pe[k] + g_rat[k] * pe[k + 1]);
  }
  bet = bb[1];
  pp[1] = 0.;
  pp[2] = dd[1] / bet;
  bb[km] = 2.;
  dd[km] = 3. * pe[km];
  for (k = 2; k <= km; k++) {
    gam[k] = g_rat[k - 1] / bet;
    bet = bb[k] - gam[k];
    pp[k + 1] = (dd[k] - pp[k]) / bet;
  }
  for (k = km; k >= 2; k--) pp[k] = pp[k] - gam[k] * pp[k + 1];
  for (k = 2; k <= km; k++) aa[k] = t1g * 0.5 * (gm2[k - 1] + gm2[k]) / (dz2[k - 1] + dz2[k]) * (pem[k]);
  bet = dm2[1] - aa[2];
  w2[1] = (dm2[1] * w1[1] + dt * pp[2]) / bet;
  for (k = 2; k <= km - 1; k++) {
    gam[k] = aa[k] / bet;
    bet = dm2[k] - (aa[k] + aa[k + 1] + aa[k] * gam[k]);
    w2[k] = (dm2[k] * w1[k] + dt * (pp[k + 1] - pp[k]) - aa[k] * w2[k - 1]) / bet;
  }
  p1 = t1g * gm2[km] / dz2[km] * (pem[km + 1]);
  gam[km] = aa[km] / bet;
  bet = dm2[km] - (aa[km] + p1 + aa[km] * gam[km]);
  w2[km] = (dm2[km] * w1[km] + dt * (pp[km + 1] - pp[km]) - p1 * ws - aa[km] * w2[km - 1]) / bet;
  for (k = km - 1; k >= 1; k--) w2[k] = w2[k] - gam[k + 1] * w2[k + 1];
  for (k = 1; k <= g_k_rf && k <= km; k++) w2[k] = w2[k] * g_rff[k - 1]; /* :1363-1371 */
  pe[1] = 0.;
  for (k = 1; k <= km; k++) pe[k + 1] = pe[k] + dm2[k] * (w2[k] - w1[k]) * rdt;
  p1 = (pe[km] + 2. * pe[km + 1]) * r3;
  dz2[km] = -dm2[km] * rgas * pt2[km] * fv3_exp((cp2[km] - 1.) * fv3_log(dmax(p_fac * pm2[km], p1 + pm2[km])));
  for (k = km - 1; k >= 1; k--) {
    p1 = (pe[k] + bb[k] * pe[k + 1] + g_rat[k] * pe[k + 2]) * r3 - g_rat[k] * p1;
    dz2[k] = -dm2[k] * rgas * pt2[k] * fv3_exp((cp2[k] - 1.) * fv3_log(dmax(p_fac * pm2[k], p1 + pm2[k])));
  }
  free(aa); free(bb); free(dd); free(w1); free(g_rat); free(gam); free(pp);
}

/* SIM_solver, nh_utils.F90:1396-1537, one column (scale_m = 0; fast_tau_w_sec: fvo_set_fast_tau_w). */
static void sim_column(int km, double dt, double rgas, const double *gm2, const double *cp2, double *pe2,
                       const double *dm2, const double *pm2, const double *pem, double *w2, double *dz2,
                       const double *pt2, double ws, double alpha, double p_fac, double scale_m) {
  int k;
  double *aa = dalloc(km + 2), *bb = dalloc(km + 2), *dd = dalloc(km + 2), *w1 = dalloc(km + 2),
         *wk = dalloc(km + 2), *g_rat = dalloc(km + 2), *gam = dalloc(km + 2), *pp = dalloc(km + 3);
  double p1, wk1, bet;
  const double beta = 1. - alpha, ra = 1. / alpha, t2 = beta / alpha, t1g = 2. * ((alpha * dt) * (alpha * dt)),
               rdt = 1. / dt;
  for (k = 1; k <= km; k++) {
    w1[k] = w2[k];
    pe2[k] = fv3_exp(gm2[k] * fv3_log(-dm2[k] / dz2[k] * rgas * pt2[k])) - pm2[k];
  }
  for (k = 1; k <= km - 1; k++) {
    g_rat[k] = dm2[k] / dm2[k + 1];
    bb[k] = 2. * (1. + g_rat[k]);
    dd[k] = 3. * (pe2[k] + g_rat[k] * pe2[k + 1]);
  }
  bet = bb[1];
  pp[1] = 0.;
  pp[2] = dd[1] / bet;
  bb[km] = 2.;
  dd[km] = 3. * pe2[km];
  for (k = 2; k <= km; k++) {
    gam[k] = g_rat[k - 1] / bet;
    bet = bb[k] - gam[k];
    pp[k + 1] = (dd[k] - pp[k]) / bet;
  }
  for (k = km; k >= 2; k--) pp[k] = pp[k] - gam[k] * pp[k + 1];
  for (k = 1; k <= km + 1; k++) pe2[k] = pem[k];
  for (k = 2; k <= km; k++) {
    aa[k] = t1g * 0.5 * (gm2[k - 1] + gm2[k]) / (dz2[k - 1] + dz2[k]) * pe2[k];
    wk[k] = t2 * aa[k] * (w1[k - 1] - w1[k]);
    aa[k] = aa[k] - scale_m * dm2[1];
  }
  bet = dm2[1] - aa[2];
  w2[1] = (dm2[1] * w1[1] + dt * pp[2] + wk[2]) / bet;
  for (k = 2; k <= km - 1; k++) {
    gam[k] = aa[k] / bet;
    bet = dm2[k] - (aa[k] + aa[k + 1] + aa[k] * gam[k]);
    w2[k] = (dm2[k] * w1[k] + dt * (pp[k + 1] - pp[k]) + wk[k + 1] - wk[k] - aa[k] * w2[k - 1]) / bet;
  }
  wk1 = t1g * gm2[km] / dz2[km] * pe2[km + 1];
  gam[km] = aa[km] / bet;
  bet = dm2[km] - (aa[km] + wk1 + aa[km] * gam[km]);
  w2[km] = (dm2[km] * w1[km] + dt * (pp[km + 1] - pp[km]) - wk[km] + wk1 * (t2 * w1[km] - ra * ws) - aa[km] * w2[km - 1]) / bet;
  for (k = km - 1; k >= 1; k--) w2[k] = w2[k] - gam[k + 1] * w2[k + 1];
  for (k = 1; k <= g_k_rf && k <= km; k++) w2[k] = w2[k] * g_rff[k - 1]; /* :1498-1506 */
  pe2[1] = 0.;
  for (k = 1; k <= km; k++) pe2[k + 1] = pe2[k] + (dm2[k] * (w2[k] - w1[k]) * rdt - beta * (pp[k + 1] - pp[k])) * ra;
  p1 = (pe2[km] + 2. * pe2[km + 1]) * r3;
  dz2[km] = -dm2[km] * rgas * pt2[km] * fv3_exp((cp2[km] - 1.) * fv3_log(dmax(p_fac * pm2[km], p1 + pm2[km])));
  for (k = km - 1; k >= 1; k--) {
    p1 = (pe2[k] + bb[k] * pe2[k + 1] + g_rat[k] * pe2[k + 2]) * r3 - g_rat[k] * p1;
    dz2[k] = -dm2[k] * rgas * pt2[k] * fv3_exp((cp2[k] - 1.) * fv3_log(dmax(p_fac * pm2[k], p1 + pm2[k])));
  }
  for (k = 1; k <= km + 1; k++) pe2[k] = pe2[k] + beta * (pp[k] - pe2[k]);
  free(aa); free(bb); free(dd); free(w1); free(wk); free(g_rat); free(gam); free(pp);
}

/* SIM3_solver, nh_utils.F90:984-1132 (alpha = |a_imp|, scale_m = 0) and SIM3p0_solver, :1134-1274 (beta = 0; p0 != 0), one column:
 * the full pressure at the layer centres reconstructed to the interfaces (top value pem(1), bottom with the weight of the half layer),
 * the w system with the FULL interface pressure in its coefficients, the new thickness from the full pressure. */
static void sim3_column(int km, double dt, double rgas, double gama, double kappa, double *pe2, const double *dm, const double *pem,
                        double *w2, double *dz2, const double *pt2, double ws, double alpha, double p_fac, double scale_m, double grav,
                        int p0) {
  int k;
  double *aa = dalloc(km + 2), *bb = dalloc(km + 2), *dd = dalloc(km + 2), *w1 = dalloc(km + 2), *wk = dalloc(km + 2),
         *g_rat = dalloc(km + 2), *gam = dalloc(km + 2), *pp = dalloc(km + 3);
  double p1, wk1, bet;
  const double beta = 1. - alpha, ra = 1. / alpha, t2 = beta / alpha;
  const double t1g = p0 ? 2. * gama * (dt * dt) : gama * 2. * ((alpha * dt) * (alpha * dt));
  const double rdt = 1. / dt, capa1 = kappa - 1., r2g = grav / 2., r6g = grav / 6.;
  for (k = 1; k <= km; k++) {
    w1[k] = w2[k];
    aa[k] = fv3_exp(gama * fv3_log(-dm[k] / dz2[k] * rgas * pt2[k]));
  }
  for (k = 1; k <= km - 1; k++) {
    g_rat[k] = dm[k] / dm[k + 1];
    bb[k] = 2. * (1. + g_rat[k]);
    dd[k] = 3. * (aa[k] + g_rat[k] * aa[k + 1]);
  }
  bet = bb[1];
  pe2[1] = pem[1];
  pe2[2] = (dd[1] - pem[1]) / bet;
  bb[km] = 2.;
  dd[km] = 3. * aa[km] + r2g * dm[km];
  for (k = 2; k <= km; k++) {
    gam[k] = g_rat[k - 1] / bet;
    bet = bb[k] - gam[k];
    pe2[k + 1] = (dd[k] - pe2[k]) / bet;
  }
  for (k = km; k >= 2; k--) pe2[k] = pe2[k] - gam[k] * pe2[k + 1];
  for (k = 1; k <= km + 1; k++) pp[k] = pe2[k] - pem[k];
  for (k = 2; k <= km; k++) {
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
  for (k = 2; k <= km - 1; k++) {
    gam[k] = aa[k] / bet;
    bet = dm[k] - (aa[k] + aa[k + 1] + aa[k] * gam[k]);
    if (p0)
      w2[k] = (dm[k] * w1[k] + dt * (pp[k + 1] - pp[k]) - aa[k] * w2[k - 1]) / bet;
    else
      w2[k] = (dm[k] * w1[k] + dt * (pp[k + 1] - pp[k]) + wk[k + 1] - wk[k] - aa[k] * w2[k - 1]) / bet;
  }
  wk1 = t1g / dz2[km] * pe2[km + 1];
  gam[km] = aa[km] / bet;
  bet = dm[km] - (aa[km] + wk1 + aa[km] * gam[km]);
  if (p0)
    w2[km] = (dm[km] * w1[km] + dt * (pp[km + 1] - pp[km]) - wk1 * ws - aa[km] * w2[km - 1]) / bet;
  else
    w2[km] = (dm[km] * w1[km] + dt * (pp[km + 1] - pp[km]) - wk[km] + wk1 * (t2 * w1[km] - ra * ws) - aa[km] * w2[km - 1]) / bet;
  for (k = km - 1; k >= 1; k--) w2[k] = w2[k] - gam[k + 1] * w2[k + 1];
  pe2[1] = 0.;
  for (k = 1; k <= km; k++) {
    if (p0)
      pe2[k + 1] = pe2[k] + dm[k] * (w2[k] - w1[k]) * rdt;
    else
      pe2[k + 1] = pe2[k] + (dm[k] * (w2[k] - w1[k]) * rdt - beta * (pp[k + 1] - pp[k])) * ra;
  }
  pe2[1] = pem[1];
  for (k = 2; k <= km + 1; k++) pe2[k] = dmax(p_fac * pem[k], pe2[k] + pem[k]);
  p1 = (pe2[km] + 2. * pe2[km + 1]) * r3 - r6g * dm[km];
  dz2[km] = -dm[km] * rgas * pt2[km] * fv3_exp(capa1 * fv3_log(p1));
  for (k = km - 1; k >= 1; k--) {
    p1 = (pe2[k] + bb[k] * pe2[k + 1] + g_rat[k] * pe2[k + 2]) * r3 - g_rat[k] * p1;
    dz2[k] = -dm[k] * rgas * pt2[k] * fv3_exp(capa1 * fv3_log(p1));
  }
  for (k = 1; k <= km + 1; k++) {
    pe2[k] = pe2[k] - pem[k];
    if (!p0) pe2[k] = pe2[k] + beta * (pp[k] - pe2[k]);
  }
  free(aa); free(bb); free(dd); free(w1); free(wk); free(g_rat); free(gam); free(pp);
}

/* RIM_2D, nh_utils.F90:751-982, one column: the Riemann invariants of every layer carried along the characteristics for ms sub-steps
 * of bdt / ms; the layers from the top down to ks0 whose sound-crossing time exceeds bdt take the one-step form (:795-851). */
static void rim_2d_column(int ms, double bdt, int km, double rgas, double gama, const double *gm2, double *pe2, const double *dm2,
                          const double *pm2, double *w2, double *dz2, const double *pt2, double ws, int c_core) {
  int k, n, ke, kt1, ktop, ks0, ks1;
  double *m_bot = dalloc(km + 3), *m_top = dalloc(km + 3), *r_bot = dalloc(km + 3), *r_top = dalloc(km + 3), *pe1 = dalloc(km + 3),
         *pbar = dalloc(km + 3), *wbar = dalloc(km + 3), *r_hi = dalloc(km + 2), *r_lo = dalloc(km + 2), *dz = dalloc(km + 2),
         *wm = dalloc(km + 2), *dm = dalloc(km + 2), *dts = dalloc(km + 2), *pf1 = dalloc(km + 2), *wc = dalloc(km + 2),
         *cm = dalloc(km + 2), *pp = dalloc(km + 2), *pt1 = dalloc(km + 2);
  const double grg = gama * rgas, rdt = 1. / bdt, dt = bdt / (double)ms, ws2 = 2. * ws;
  double z_frac, ptmp1, rden, pf, time_left, m_surf;
  int done = 0;
  for (k = 1; k <= km; k++) {
    dz[k] = dz2[k];
    dm[k] = dm2[k];
    wm[k] = w2[k] * dm[k];
    pt1[k] = pt2[k];
  }
  wbar[km + 1] = ws;
  ks0 = 1;
  if (ms > 1 && ms < 8) {
    ks0 = km;
    for (k = 1; k <= km; k++) {
      rden = -rgas * dm[k] / dz[k];
      pf1[k] = fv3_exp(gm2[k] * fv3_log(rden * pt1[k]));
      dts[k] = -dz[k] / sqrt(grg * pf1[k] / rden);
      if (bdt > dts[k]) {
        ks0 = k - 1;
        break;
      }
    }
    /* ks0 = 0 (the top layer itself is crossed within bdt) reads unset locals and writes pbar(0) in the reference: undefined there;
     * taken here as ks0 = 1, the form of every other case that leaves no layer to the one-step branch */
    if (ks0 < 1) ks0 = 1;
    if (ks0 != 1) {
      for (k = 1; k <= ks0; k++) {
        cm[k] = dm[k] / dts[k];
        wc[k] = wm[k] / dts[k];
        pp[k] = pf1[k] - pm2[k];
      }
      wbar[1] = (wc[1] + pp[1]) / cm[1];
      for (k = 2; k <= ks0; k++) {
        wbar[k] = (wc[k - 1] + wc[k] + pp[k] - pp[k - 1]) / (cm[k - 1] + cm[k]);
        pbar[k] = bdt * (cm[k - 1] * wbar[k] - wc[k - 1] + pp[k - 1]);
        pe1[k] = pbar[k];
      }
      if (ks0 == km) {
        pbar[km + 1] = bdt * (cm[km] * wbar[km + 1] - wc[km] + pp[km]);
        for (k = 1; k <= km; k++) {
          dz2[k] = dz[k] + bdt * (wbar[k + 1] - wbar[k]);
          if (!c_core) w2[k] = (wm[k] + pbar[k + 1] - pbar[k]) / dm[k];
        }
        pe2[1] = 0.;
        for (k = 2; k <= km + 1; k++) pe2[k] = pbar[k] * rdt;
        done = 1;
      } else {
        for (k = 1; k <= ks0 - 1; k++) {
          dz2[k] = dz[k] + bdt * (wbar[k + 1] - wbar[k]);
          if (!c_core) w2[k] = (wm[k] + pbar[k + 1] - pbar[k]) / dm[k];
        }
        pbar[ks0] = pbar[ks0] / (double)ms;
      }
    }
  }
  if (!done) {
    ks1 = ks0;
    for (n = 1; n <= ms; n++) {
      for (k = ks1; k <= km; k++) {
        rden = -rgas * dm[k] / dz[k];
        pf = fv3_exp(gm2[k] * fv3_log(rden * pt1[k]));
        dts[k] = -dz[k] / sqrt(grg * pf / rden);
        ptmp1 = dts[k] * (pf - pm2[k]);
        r_lo[k] = wm[k] + ptmp1;
        r_hi[k] = wm[k] - ptmp1;
      }
      ktop = km;
      for (k = ks1; k <= km; k++)
        if (dt > dts[k]) {
          ktop = k - 1;
          break;
        }
      if (ktop >= ks1)
        for (k = ks1; k <= ktop; k++) {
          z_frac = dt / dts[k];
          r_bot[k] = z_frac * r_lo[k];
          r_top[k + 1] = z_frac * r_hi[k];
          m_bot[k] = z_frac * dm[k];
          m_top[k + 1] = m_bot[k];
        }
      if (!(ktop >= ks1 && ktop == km)) {
        for (k = ktop + 2; k <= km + 1; k++) {
          m_top[k] = 0.;
          r_top[k] = 0.;
        }
        kt1 = ktop > 1 ? ktop : 1;
        for (ke = km + 1; ke >= ktop + 2; ke--) {
          time_left = dt;
          for (k = ke - 1; k >= kt1; k--) {
            if (time_left > dts[k]) {
              time_left = time_left - dts[k];
              m_top[ke] = m_top[ke] + dm[k];
              r_top[ke] = r_top[ke] + r_hi[k];
            } else {
              z_frac = time_left / dts[k];
              m_top[ke] = m_top[ke] + z_frac * dm[k];
              r_top[ke] = r_top[ke] + z_frac * r_hi[k];
              break;
            }
          }
        }
        for (k = ktop + 1; k <= km; k++) {
          m_bot[k] = 0.;
          r_bot[k] = 0.;
        }
        for (ke = ktop + 1; ke <= km; ke++) {
          int next = 0;
          time_left = dt;
          for (k = ke; k <= km; k++) {
            if (time_left > dts[k]) {
              time_left = time_left - dts[k];
              m_bot[ke] = m_bot[ke] + dm[k];
              r_bot[ke] = r_bot[ke] + r_lo[k];
            } else {
              z_frac = time_left / dts[k];
              m_bot[ke] = m_bot[ke] + z_frac * dm[k];
              r_bot[ke] = r_bot[ke] + z_frac * r_lo[k];
              next = 1;
              break;
            }
          }
          if (next) continue;
          m_surf = m_bot[ke];
          for (k = km; k >= kt1; k--) {
            if (time_left > dts[k]) {
              time_left = time_left - dts[k];
              m_bot[ke] = m_bot[ke] + dm[k];
              r_bot[ke] = r_bot[ke] - r_hi[k];
            } else {
              z_frac = time_left / dts[k];
              m_bot[ke] = m_bot[ke] + z_frac * dm[k];
              r_bot[ke] = r_bot[ke] - z_frac * r_hi[k] + (m_bot[ke] - m_surf) * ws2;
              break;
            }
          }
        }
      }
      if (ks1 == 1) wbar[1] = r_bot[1] / m_bot[1];
      for (k = ks1 + 1; k <= km; k++) wbar[k] = (r_bot[k] + r_top[k]) / (m_top[k] + m_bot[k]);
      for (k = ks1 + 1; k <= km + 1; k++) {
        pbar[k] = m_top[k] * wbar[k] - r_top[k];
        pe1[k] = pe1[k] + pbar[k];
      }
      if (n == ms) {
        for (k = ks1; k <= km; k++) {
          dz2[k] = dz[k] + dt * (wbar[k + 1] - wbar[k]);
          if (!c_core) w2[k] = (wm[k] + pbar[k + 1] - pbar[k]) / dm[k];
        }
      } else {
        for (k = ks1; k <= km; k++) {
          dz[k] = dz[k] + dt * (wbar[k + 1] - wbar[k]);
          wm[k] = wm[k] + pbar[k + 1] - pbar[k];
        }
      }
    }
    pe2[1] = 0.;
    for (k = 2; k <= km + 1; k++) pe2[k] = pe1[k] * rdt;
  }
  free(m_bot); free(m_top); free(r_bot); free(r_top); free(pe1); free(pbar); free(wbar); free(r_hi); free(r_lo); free(dz); free(wm);
  free(dm); free(dts); free(pf1); free(wc); free(cm); free(pp); free(pt1);
}

/* Riem_Solver_c, nh_utils.F90:323-480 (a_imp > 0.5 -> SIM1_solver).  q_con != NULL: use_cond = .true. (:383-396,
 * :413-438); cappa != NULL (with q_con): moist_kappa = .true. (:414-424).  hs, ws: A; w3, pt, delp, q_con, cappa: A x km;
 * gz, pef: A x (km+1). */
int fvo_riem_solver_c(const fvo_grid *g, int km, double dt, double akap, double ptop, const double *hs,
                      const double *w3, const double *pt, const double *delp, double *gz, double *pef,
                      const double *ws, double p_fac, double a_imp, double grav, double rdgas, const double *q_con,
                      const double *cappa) {
  return fvo_riem_solver_c_ms(g, km, dt, akap, ptop, hs, w3, pt, delp, gz, pef, ws, p_fac, a_imp, grav, rdgas, q_con, cappa, 1);
}
/* ... with m_split (the sub-steps of RIM_2D, a_imp <= 0.5) */
int fvo_riem_solver_c_ms(const fvo_grid *g, int km, double dt, double akap, double ptop, const double *hs,
                         const double *w3, const double *pt, const double *delp, double *gz, double *pef,
                         const double *ws, double p_fac, double a_imp, double grav, double rdgas, const double *q_con,
                         const double *cappa, int m_split) {
  BOUNDS(g);
  int j;
  const double rgrav = 1. / grav;
  const int is1 = is - 1, ie1 = ie + 1;
#pragma omp parallel for schedule(dynamic)
  for (j = js - 1; j <= je + 1; j++) {
    int i, k;
    double *dm = dalloc(km + 2), *dz2 = dalloc(km + 2), *w2 = dalloc(km + 2), *pm2 = dalloc(km + 2),
           *gm2 = dalloc(km + 2), *cp2 = dalloc(km + 2), *pem = dalloc(km + 3), *pe2 = dalloc(km + 3),
           *pt2 = dalloc(km + 2), *peg = dalloc(km + 3);
    for (i = is1; i <= ie1; i++) {
      for (k = 1; k <= km; k++) dm[k] = delp[A3(i, j, k)];
      pef[A3(i, j, 1)] = ptop;
      pem[1] = ptop;
      peg[1] = ptop;
      for (k = 2; k <= km + 1; k++) {
        pem[k] = pem[k - 1] + dm[k - 1];
        if (q_con) peg[k] = peg[k - 1] + dm[k - 1] * (1. - q_con[A3(i, j, k - 1)]); /* :394 */
      }
      for (k = 1; k <= km; k++) {
        dz2[k] = gz[A3(i, j, k + 1)] - gz[A3(i, j, k)];
        if (q_con)
          pm2[k] = (peg[k + 1] - peg[k]) / fv3_log(peg[k + 1] / peg[k]);
        else
          pm2[k] = dm[k] / fv3_log(pem[k + 1] / pem[k]);
        cp2[k] = (q_con && cappa) ? cappa[A3(i, j, k)] : akap;
        gm2[k] = 1. / (1. - cp2[k]);
        dm[k] = dm[k] * rgrav;
        w2[k] = w3[A3(i, j, k)];
        pt2[k] = pt[A3(i, j, k)];
      }
      if (a_imp < -0.01)        /* nh_utils.F90:449-459 */
        sim3_column(km, dt, rdgas, 1. / (1. - akap), akap, pe2, dm, pem, w2, dz2, pt2, ws[IA(i, j)], 1.0, p_fac, 0.0, grav, 1);
      else if (a_imp <= 0.5)
        rim_2d_column(m_split, dt, km, rdgas, 1. / (1. - akap), gm2, pe2, dm, pm2, w2, dz2, pt2, ws[IA(i, j)], 1);
      else
        sim1_column(km, dt, rdgas, gm2, cp2, pe2, dm, pm2, pem, w2, dz2, pt2, ws[IA(i, j)], p_fac);
      for (k = 2; k <= km + 1; k++) pef[A3(i, j, k)] = pe2[k] + pem[k];
      gz[A3(i, j, km + 1)] = hs[IA(i, j)];
      for (k = km; k >= 1; k--) gz[A3(i, j, k)] = gz[A3(i, j, k + 1)] - dz2[k] * grav;
    }
    free(dm); free(dz2); free(w2); free(pm2); free(gm2); free(cp2); free(pem); free(pe2); free(pt2); free(peg);
  }
  return FVO_OK;
}

/* Riem_Solver3, nh_core.F90:47-241 (d2bg_zq = 0).  q_con != NULL: use_cond = .true. (:113-131, :145-154); cappa != NULL:
 * moist_kappa = .true. (:96-102).  q_con, cappa: A x km.
 * zs: A; ws: CC; w, delp, pt: A x km; zh, ppe, pk3: A x (km+1); delz: CC x km; pk: CC x (km+1);
 * pe: (is-1:ie+1, km+1, js-1:je+1); peln: (is:ie, km+1, js:je). */
int fvo_riem_solver3(const fvo_grid *g, int km, double dt, double akap, double ptop, const double *zs, double *w,
                     double *delz, const double *pt, const double *delp, double *zh, double *pe, double *ppe,
                     double *pk3, double *pk, double *peln, const double *ws, double p_fac, double a_imp,
                     int use_logp, int last_call, int fp_out, double grav, double rdgas, const double *q_con,
                     const double *cappa) {
  return fvo_riem_solver3_ms(g, km, dt, akap, ptop, zs, w, delz, pt, delp, zh, pe, ppe, pk3, pk, peln, ws, p_fac, a_imp, use_logp,
                             last_call, fp_out, grav, rdgas, q_con, cappa, 1);
}
int fvo_riem_solver3_ms(const fvo_grid *g, int km, double dt, double akap, double ptop, const double *zs, double *w,
                        double *delz, const double *pt, const double *delp, double *zh, double *pe, double *ppe,
                        double *pk3, double *pk, double *peln, const double *ws, double p_fac, double a_imp,
                        int use_logp, int last_call, int fp_out, double grav, double rdgas, const double *q_con,
                        const double *cappa, int m_split) {
  BOUNDS(g);
  int j;
  const double rgrav = 1. / grav;
  const double peln1 = fv3_log(ptop);
  const double ptk = fv3_exp(akap * peln1);
#pragma omp parallel for schedule(dynamic)
  for (j = js; j <= je; j++) {
    int i, k;
    double *dm = dalloc(km + 2), *dz2 = dalloc(km + 2), *w2 = dalloc(km + 2), *pm2 = dalloc(km + 2),
           *gm2 = dalloc(km + 2), *cp2 = dalloc(km + 2), *pem = dalloc(km + 3), *pe2 = dalloc(km + 3),
           *peln2 = dalloc(km + 3), *pt2 = dalloc(km + 2), *peg = dalloc(km + 3), *pelng = dalloc(km + 3);
    for (i = is; i <= ie; i++) {
      for (k = 1; k <= km; k++) {
        dm[k] = delp[A3(i, j, k)];
        cp2[k] = cappa ? cappa[A3(i, j, k)] : akap;
      }
      pem[1] = ptop;
      peln2[1] = peln1;
      pk3[A3(i, j, 1)] = ptk;
      peg[1] = ptop;
      pelng[1] = peln1;
      for (k = 2; k <= km + 1; k++) {
        pem[k] = pem[k - 1] + dm[k - 1];
        peln2[k] = fv3_log(pem[k]);
        if (q_con) { /* excluding the contribution from condensates, :125-127 */
          peg[k] = peg[k - 1] + dm[k - 1] * (1. - q_con[A3(i, j, k - 1)]);
          pelng[k] = fv3_log(peg[k]);
        }
        pk3[A3(i, j, k)] = fv3_exp(akap * peln2[k]);
      }
      for (k = 1; k <= km; k++) {
        if (q_con)
          pm2[k] = (peg[k + 1] - peg[k]) / (pelng[k + 1] - pelng[k]);
        else
          pm2[k] = dm[k] / (peln2[k + 1] - peln2[k]);
        gm2[k] = 1. / (1. - cp2[k]);
        dm[k] = dm[k] * rgrav;
        dz2[k] = zh[A3(i, j, k + 1)] - zh[A3(i, j, k)];
        w2[k] = w[A3(i, j, k)];
        pt2[k] = pt[A3(i, j, k)];
      }
      if (a_imp < -0.999)       /* nh_core.F90:169-185 */
        sim3_column(km, dt, rdgas, 1. / (1. - akap), akap, pe2, dm, pem, w2, dz2, pt2, ws[ICC(i, j)], 1.0, p_fac, 0.0, grav, 1);
      else if (a_imp < -0.5)
        sim3_column(km, dt, rdgas, 1. / (1. - akap), akap, pe2, dm, pem, w2, dz2, pt2, ws[ICC(i, j)], fabs(a_imp), p_fac, 0.0, grav, 0);
      else if (a_imp <= 0.5)
        rim_2d_column(m_split, dt, km, rdgas, 1. / (1. - akap), gm2, pe2, dm, pm2, w2, dz2, pt2, ws[ICC(i, j)], 0);
      else if (a_imp > 0.999)
        sim1_column(km, dt, rdgas, gm2, cp2, pe2, dm, pm2, pem, w2, dz2, pt2, ws[ICC(i, j)], p_fac);
      else
        sim_column(km, dt, rdgas, gm2, cp2, pe2, dm, pm2, pem, w2, dz2, pt2, ws[ICC(i, j)], a_imp, p_fac, 0.0);
      for (k = 1; k <= km; k++) {
        w[A3(i, j, k)] = w2[k];
        delz[(size_t)(k - 1) * nx * ny + ICC(i, j)] = dz2[k];
      }
      if (last_call) {
        for (k = 1; k <= km + 1; k++) {
          peln[(size_t)(j - js) * nx * (km + 1) + (size_t)(k - 1) * nx + (i - is)] = peln2[k];
          pk[(size_t)(k - 1) * nx * ny + ICC(i, j)] = pk3[A3(i, j, k)];
          pe[(size_t)(j - (js - 1)) * (nx + 2) * (km + 1) + (size_t)(k - 1) * (nx + 2) + (i - (is - 1))] = pem[k];
        }
      }
      for (k = 1; k <= km + 1; k++) ppe[A3(i, j, k)] = fp_out ? pe2[k] + pem[k] : pe2[k];
      if (use_logp)
        for (k = 2; k <= km + 1; k++) pk3[A3(i, j, k)] = peln2[k];
      zh[A3(i, j, km + 1)] = zs[IA(i, j)];
      for (k = km; k >= 1; k--) zh[A3(i, j, k)] = zh[A3(i, j, k + 1)] - dz2[k];
    }
    free(dm); free(dz2); free(w2); free(pm2); free(gm2); free(cp2); free(pem); free(pe2); free(peln2); free(pt2);
    free(peg); free(pelng);
  }
  return FVO_OK;
}

/* edge_profile, nh_utils.F90:1590-1696, non-uniform branch, limiter = 0, one (i,j) column of two
 * fields.  q1, q2 point at level 1 with level stride ks; outputs have stride kse. */
static void edge_profile_col(int km, const double *dp0, const double *q1, const double *q2, size_t ks, double *q1e,
                             double *q2e, size_t kse) {
  int k;
  double *qe1 = dalloc(km + 2), *qe2 = dalloc(km + 2), *gam = dalloc(km + 2);
  double g0, gk = 0., xt1, xt2, a_bot, bet;
#define Q1(k) q1[(size_t)((k)-1) * ks]
#define Q2(k) q2[(size_t)((k)-1) * ks]
  g0 = dp0[1] / dp0[0];
  xt1 = 2. * g0 * (g0 + 1.);
  bet = g0 * (g0 + 0.5);
  qe1[1] = (xt1 * Q1(1) + Q1(2)) / bet;
  qe2[1] = (xt1 * Q2(1) + Q2(2)) / bet;
  gam[1] = (1. + g0 * (g0 + 1.5)) / bet;
  for (k = 2; k <= km; k++) {
    gk = dp0[k - 2] / dp0[k - 1];
    bet = 2. + 2. * gk - gam[k - 1];
    qe1[k] = (3. * (Q1(k - 1) + gk * Q1(k)) - qe1[k - 1]) / bet;
    qe2[k] = (3. * (Q2(k - 1) + gk * Q2(k)) - qe2[k - 1]) / bet;
    gam[k] = gk / bet;
  }
  a_bot = 1. + gk * (gk + 1.5);
  xt1 = 2. * gk * (gk + 1.);
  xt2 = gk * (gk + 0.5) - a_bot * gam[km];
  qe1[km + 1] = (xt1 * Q1(km) + Q1(km - 1) - a_bot * qe1[km]) / xt2;
  qe2[km + 1] = (xt1 * Q2(km) + Q2(km - 1) - a_bot * qe2[km]) / xt2;
  for (k = km; k >= 1; k--) {
    qe1[k] = qe1[k] - gam[k] * qe1[k + 1];
    qe2[k] = qe2[k] - gam[k] * qe2[k + 1];
  }
  for (k = 1; k <= km + 1; k++) {
    q1e[(size_t)(k - 1) * kse] = qe1[k];
    q2e[(size_t)(k - 1) * kse] = qe2[k];
  }
#undef Q1
#undef Q2
  free(qe1); free(qe2); free(gam);
}

int fvo_del6_vt_flux(const fvo_grid *g, int nord, double damp, const double *q, double *d2, double *fx2, double *fy2);

/* update_dz_d, nh_utils.F90:204-321.  ndif, damp: length km+1 (entry km+1 is set here, :240-241);
 * zs: A; zh: A x (km+1); crx, xfx: CX x km; cry, yfx: CY x km; ws: CC. */
int fvo_update_dz_d(const fvo_grid *g, int km, int *ndif, double *damp, int hord, const double *dp0, const double *zs,
                    double *zh, const double *crx, const double *cry, const double *xfx, const double *yfx, double *ws,
                    double rdt) {
  BOUNDS(g);
  int i, j, k;
  const size_t nCX = (size_t)(nx + 1) * njd, nCY = (size_t)nid * (ny + 1), nA = (size_t)nid * njd;
  double *crx_adv = dalloc(nCX * (km + 1)), *xfx_adv = dalloc(nCX * (km + 1));
  double *cry_adv = dalloc(nCY * (km + 1)), *yfx_adv = dalloc(nCY * (km + 1));
  damp[km] = damp[km - 1];
  ndif[km] = ndif[km - 1];
  for (j = jsd; j <= jed; j++) {
    for (i = is; i <= ie + 1; i++)
      edge_profile_col(km, dp0, crx + ICX(i, j), xfx + ICX(i, j), nCX, crx_adv + ICX(i, j), xfx_adv + ICX(i, j), nCX);
    if (j <= je + 1 && j >= js)
      for (i = isd; i <= ied; i++)
        edge_profile_col(km, dp0, cry + ICY(i, j), yfx + ICY(i, j), nCY, cry_adv + ICY(i, j), yfx_adv + ICY(i, j), nCY);
  }
#pragma omp parallel for private(i, j) schedule(dynamic)
  for (k = 1; k <= km + 1; k++) {
    double *ra_x = dalloc((size_t)nx * njd), *ra_y = dalloc((size_t)nid * ny);
    double *fx = dalloc((size_t)(nx + 1) * ny), *fy = dalloc((size_t)nx * (ny + 1));
    const double *cxa = crx_adv + nCX * (k - 1), *xfa = xfx_adv + nCX * (k - 1);
    const double *cya = cry_adv + nCY * (k - 1), *yfa = yfx_adv + nCY * (k - 1);
    double *zk = zh + nA * (k - 1);
    for (j = jsd; j <= jed; j++)
      for (i = is; i <= ie; i++)
        ra_x[(size_t)(j - jsd) * nx + (i - is)] = g->area[IA(i, j)] + xfa[ICX(i, j)] - xfa[ICX(i + 1, j)];
    for (j = js; j <= je; j++)
      for (i = isd; i <= ied; i++) ra_y[ICY(i, j)] = g->area[IA(i, j)] + yfa[ICY(i, j)] - yfa[ICY(i, j + 1)];
    if (damp[k - 1] > 1.E-5) {
      double *z2 = dalloc(nA), *wk2 = dalloc(nA), *fx2 = dalloc((size_t)(nid + 1) * njd), *fy2 = dalloc((size_t)nid * (njd + 1));
      memcpy(z2, zk, sizeof(double) * nA);
      fvo_fv_tp_2d(g, z2, cxa, cya, hord, fx, fy, xfa, yfa, ra_x, ra_y, NULL, NULL, NULL, -1, 0.);
      fvo_del6_vt_flux(g, ndif[k - 1], damp[k - 1], z2, wk2, fx2, fy2);
      for (j = js; j <= je; j++)
        for (i = is; i <= ie; i++)
          zk[IA(i, j)] = (z2[IA(i, j)] * g->area[IA(i, j)] + fx[IFX(i, j)] - fx[IFX(i + 1, j)] + fy[IFY(i, j)] - fy[IFY(i, j + 1)]) /
                             (ra_x[(size_t)(j - jsd) * nx + (i - is)] + ra_y[ICY(i, j)] - g->area[IA(i, j)]) +
                         (fx2[IV(i, j)] - fx2[IV(i + 1, j)] + fy2[IU(i, j)] - fy2[IU(i, j + 1)]) * g->rarea[IA(i, j)];
      free(z2); free(wk2); free(fx2); free(fy2);
    } else {
      fvo_fv_tp_2d(g, zk, cxa, cya, hord, fx, fy, xfa, yfa, ra_x, ra_y, NULL, NULL, NULL, -1, 0.);
      for (j = js; j <= je; j++)
        for (i = is; i <= ie; i++)
          zk[IA(i, j)] = (zk[IA(i, j)] * g->area[IA(i, j)] + fx[IFX(i, j)] - fx[IFX(i + 1, j)] + fy[IFY(i, j)] - fy[IFY(i, j + 1)]) /
                         (ra_x[(size_t)(j - jsd) * nx + (i - is)] + ra_y[ICY(i, j)] - g->area[IA(i, j)]);
    }
    free(ra_x); free(ra_y); free(fx); free(fy);
  }
  for (j = js; j <= je; j++) {
    for (i = is; i <= ie; i++) ws[ICC(i, j)] = (zs[IA(i, j)] - zh[A3(i, j, km + 1)]) * rdt;
    for (k = km; k >= 1; k--)
      for (i = is; i <= ie; i++) zh[A3(i, j, k)] = dmax(zh[A3(i, j, k)], zh[A3(i, j, k + 1)] + dz_min);
  }
  free(crx_adv); free(xfx_adv); free(cry_adv); free(yfx_adv);
  return FVO_OK;
}

/* p_grad_c, dyn_core.F90:1635-1694.  delpc: A x npz; pkc, gz: A x (npz+1); uc: V x npz; vc: U x npz. */
int fvo_p_grad_c(const fvo_grid *g, int npz, double dt2, const double *delpc, const double *pkc, const double *gz,
                 double *uc, double *vc, int hydrostatic) {
  BOUNDS(g);
  int k;
  const size_t nA = (size_t)nid * njd, nV = (size_t)(nid + 1) * njd, nU = (size_t)nid * (njd + 1);
#pragma omp parallel for schedule(dynamic)
  for (k = 1; k <= npz; k++) {
    int i, j;
    double *wk = dalloc(nA);
    for (j = js - 1; j <= je + 1; j++)
      for (i = is - 1; i <= ie + 1; i++)
        wk[IA(i, j)] = hydrostatic ? pkc[A3(i, j, k + 1)] - pkc[A3(i, j, k)] : delpc[A3(i, j, k)];
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++)
        uc[nV * (k - 1) + IV(i, j)] =
            uc[nV * (k - 1) + IV(i, j)] + dt2 * g->rdxc[IV(i, j)] / (wk[IA(i - 1, j)] + wk[IA(i, j)]) *
                                              ((gz[A3(i - 1, j, k + 1)] - gz[A3(i, j, k)]) * (pkc[A3(i, j, k + 1)] - pkc[A3(i - 1, j, k)]) +
                                               (gz[A3(i - 1, j, k)] - gz[A3(i, j, k + 1)]) * (pkc[A3(i - 1, j, k + 1)] - pkc[A3(i, j, k)]));
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++)
        vc[nU * (k - 1) + IU(i, j)] =
            vc[nU * (k - 1) + IU(i, j)] + dt2 * g->rdyc[IU(i, j)] / (wk[IA(i, j - 1)] + wk[IA(i, j)]) *
                                              ((gz[A3(i, j - 1, k + 1)] - gz[A3(i, j, k)]) * (pkc[A3(i, j, k + 1)] - pkc[A3(i, j - 1, k)]) +
                                               (gz[A3(i, j - 1, k)] - gz[A3(i, j, k + 1)]) * (pkc[A3(i, j - 1, k + 1)] - pkc[A3(i, j, k)]));
    free(wk);
  }
  return FVO_OK;
}

/* nh_p_grad, dyn_core.F90:1697-1792.  pp, pk, gz: A x (npz+1), converted to corner values in place
 * (a2b_ord4 with replace=.true.) exactly as the reference does; delp: A x npz; u: U x npz; v: V x npz. */
int fvo_nh_p_grad(const fvo_grid *g, int npz, double *u, double *v, double *pp, double *gz, double *delp, double *pk,
                  double dt, double top_value) {
  BOUNDS(g);
  int k;
  const size_t nA = (size_t)nid * njd, nV = (size_t)(nid + 1) * njd, nU = (size_t)nid * (njd + 1);
#pragma omp parallel for schedule(dynamic)
  for (k = 1; k <= npz + 1; k++) {
    int i, j;
    double *wk1 = dalloc(nA);
    if (k == 1) {
      for (j = js; j <= je + 1; j++)
        for (i = is; i <= ie + 1; i++) {
          pp[A3(i, j, 1)] = 0.;
          pk[A3(i, j, 1)] = top_value;
        }
    } else {
      fvo_a2b_ord4(g, pp + nA * (k - 1), wk1, 1);
      fvo_a2b_ord4(g, pk + nA * (k - 1), wk1, 1);
    }
    fvo_a2b_ord4(g, gz + nA * (k - 1), wk1, 1);
    free(wk1);
  }
#pragma omp parallel for schedule(dynamic)
  for (k = 1; k <= npz; k++) {
    int i, j;
    double *wk1 = dalloc(nA), *wk = dalloc(nA);
    double du1, dv1;
    fvo_a2b_ord4(g, delp + nA * (k - 1), wk1, 0);
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) wk[IA(i, j)] = pk[A3(i, j, k + 1)] - pk[A3(i, j, k)];
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++) {
        du1 = dt / (wk[IA(i, j)] + wk[IA(i + 1, j)]) *
              ((gz[A3(i, j, k + 1)] - gz[A3(i + 1, j, k)]) * (pk[A3(i + 1, j, k + 1)] - pk[A3(i, j, k)]) +
               (gz[A3(i, j, k)] - gz[A3(i + 1, j, k + 1)]) * (pk[A3(i, j, k + 1)] - pk[A3(i + 1, j, k)]));
        u[nU * (k - 1) + IU(i, j)] =
            (u[nU * (k - 1) + IU(i, j)] + du1 +
             dt / (wk1[IA(i, j)] + wk1[IA(i + 1, j)]) *
                 ((gz[A3(i, j, k + 1)] - gz[A3(i + 1, j, k)]) * (pp[A3(i + 1, j, k + 1)] - pp[A3(i, j, k)]) +
                  (gz[A3(i, j, k)] - gz[A3(i + 1, j, k + 1)]) * (pp[A3(i, j, k + 1)] - pp[A3(i + 1, j, k)]))) *
            g->rdx[IU(i, j)];
      }
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++) {
        dv1 = dt / (wk[IA(i, j)] + wk[IA(i, j + 1)]) *
              ((gz[A3(i, j, k + 1)] - gz[A3(i, j + 1, k)]) * (pk[A3(i, j + 1, k + 1)] - pk[A3(i, j, k)]) +
               (gz[A3(i, j, k)] - gz[A3(i, j + 1, k + 1)]) * (pk[A3(i, j, k + 1)] - pk[A3(i, j + 1, k)]));
        v[nV * (k - 1) + IV(i, j)] =
            (v[nV * (k - 1) + IV(i, j)] + dv1 +
             dt / (wk1[IA(i, j)] + wk1[IA(i, j + 1)]) *
                 ((gz[A3(i, j, k + 1)] - gz[A3(i, j + 1, k)]) * (pp[A3(i, j + 1, k + 1)] - pp[A3(i, j, k)]) +
                  (gz[A3(i, j, k)] - gz[A3(i, j + 1, k + 1)]) * (pp[A3(i, j, k + 1)] - pp[A3(i, j + 1, k)]))) *
            g->rdy[IV(i, j)];
      }
    free(wk1); free(wk);
  }
  return FVO_OK;
}

/* pk3_halo (dyn_core.F90:1395-1446) / pln_halo (:1449-1496): use_logp selects log(p). */
int fvo_pk3_halo(const fvo_grid *g, int npz, double ptop, double akap, double *pk3, const double *delp, int use_logp) {
  BOUNDS(g);
  int i, j, k, n;
  double pet;
  for (j = js; j <= je; j++) {
    const int ii[4] = {is - 2, is - 1, ie + 1, ie + 2};
    for (n = 0; n < 4; n++) {
      i = ii[n];
      pet = ptop;
      for (k = 1; k <= npz; k++) {
        pet = pet + delp[A3(i, j, k)];
        pk3[A3(i, j, k + 1)] = use_logp ? fv3_log(pet) : fv3_exp(akap * fv3_log(pet));
      }
    }
  }
  for (i = is - 2; i <= ie + 2; i++) {
    const int jj[4] = {js - 2, js - 1, je + 1, je + 2};
    for (n = 0; n < 4; n++) {
      j = jj[n];
      pet = ptop;
      for (k = 1; k <= npz; k++) {
        pet = pet + delp[A3(i, j, k)];
        pk3[A3(i, j, k + 1)] = use_logp ? fv3_log(pet) : fv3_exp(akap * fv3_log(pet));
      }
    }
  }
  return FVO_OK;
}

/* pe_halo, dyn_core.F90:1498-1526.  pe: (is-1:ie+1, npz+1, js-1:je+1). */
int fvo_pe_halo(const fvo_grid *g, int npz, double ptop, double *pe, const double *delp) {
  BOUNDS(g);
  int i, j, k;
#define PE(i, k, j) pe[(size_t)((j) - (js - 1)) * (nx + 2) * (npz + 1) + (size_t)((k)-1) * (nx + 2) + ((i) - (is - 1))]
  for (j = js; j <= je; j++) {
    PE(is - 1, 1, j) = ptop;
    PE(ie + 1, 1, j) = ptop;
    for (k = 1; k <= npz; k++) {
      PE(is - 1, k + 1, j) = PE(is - 1, k, j) + delp[A3(is - 1, j, k)];
      PE(ie + 1, k + 1, j) = PE(ie + 1, k, j) + delp[A3(ie + 1, j, k)];
    }
  }
  for (i = is - 1; i <= ie + 1; i++) {
    PE(i, 1, js - 1) = ptop;
    PE(i, 1, je + 1) = ptop;
    for (k = 1; k <= npz; k++) {
      PE(i, k + 1, js - 1) = PE(i, k, js - 1) + delp[A3(i, js - 1, k)];
      PE(i, k + 1, je + 1) = PE(i, k, je + 1) + delp[A3(i, je + 1, k)];
    }
  }
#undef PE
  return FVO_OK;
}

/* geopk, dyn_core.F90:2202-2353 (use_cond = .false., not bounded_domain).  hs: A; pt, delp: A x km;
 * gz, pk: A x (km+1); pe (is-1:ie+1, km+1, js-1:je+1); peln (is:ie, km+1, js:je); pkz CC x km. */
int fvo_geopk(const fvo_grid *g, int km, double ptop, double akap, double cp_air, double *pe, double *peln,
              const double *delp, double *pk, double *gz, const double *hs, const double *pt, double *pkz, int CG) {
  BOUNDS(g);
  int j;
  const double peln1 = fv3_log(ptop);
  const double ptk = pow(ptop, akap); /* dyn_core.F90:222 (the one place the reference uses **) */
  const int ifirst = CG ? is - 1 : is - 2, ilast = CG ? ie + 1 : ie + 2;
  const int jfirst = CG ? js - 1 : js - 2, jlast = CG ? je + 1 : je + 2;
#define PE(i, k, j) pe[(size_t)((j) - (js - 1)) * (nx + 2) * (km + 1) + (size_t)((k)-1) * (nx + 2) + ((i) - (is - 1))]
#define PELN(i, k, j) peln[(size_t)((j)-js) * nx * (km + 1) + (size_t)((k)-1) * nx + ((i)-is)]
#pragma omp parallel for schedule(dynamic)
  for (j = jfirst; j <= jlast; j++) {
    int i, k;
    for (i = ifirst; i <= ilast; i++) {
      double p1d = ptop, logp;
      pk[A3(i, j, 1)] = ptk;
      gz[A3(i, j, km + 1)] = hs[IA(i, j)];
      if (j >= js && j <= je && i >= is && i <= ie) PELN(i, 1, j) = peln1;
      if (j > js - 2 && j < je + 2 && i >= is - 1 && i <= ie + 1) PE(i, 1, j) = ptop;
      for (k = 2; k <= km + 1; k++) {
        p1d = p1d + delp[A3(i, j, k - 1)];
        logp = fv3_log(p1d);
        pk[A3(i, j, k)] = fv3_exp(akap * logp);
        if (j > js - 2 && j < je + 2) {
          if (i >= is - 1 && i <= ie + 1) PE(i, k, j) = p1d;
          if (j >= js && j <= je && i >= is && i <= ie) PELN(i, k, j) = logp;
        }
      }
      for (k = km; k >= 1; k--)
        gz[A3(i, j, k)] = gz[A3(i, j, k + 1)] + cp_air * pt[A3(i, j, k)] * (pk[A3(i, j, k + 1)] - pk[A3(i, j, k)]);
      if (!CG && j >= js && j <= je && i >= is && i <= ie)
        for (k = 1; k <= km; k++)
          pkz[(size_t)(k - 1) * nx * ny + ICC(i, j)] =
              (pk[A3(i, j, k + 1)] - pk[A3(i, j, k)]) / (akap * (PELN(i, k + 1, j) - PELN(i, k, j)));
    }
  }
#undef PE
#undef PELN
  return FVO_OK;
}

/* del2_cubed, model/dyn_core.F90:2356-2465 (grid_type >= 3: no cube corners, no copy_corners; the halo of q
 * must be up to date on entry -- the reference calls mpp_update_domains first, :2399).  q: A x km, in place. */
int fvo_del2_cubed(const fvo_grid *g, int km, double cd, int nmax, double *q) {
  BOUNDS(g);
  int k;
  const int ntimes = nmax < 3 ? nmax : 3;
  const int cubed = g->grid_type < 3, npx = g->npx, npy = g->npy;
  const double r3 = 1. / 3.;
#pragma omp parallel for schedule(dynamic)
  for (k = 1; k <= km; k++) {
    int i, j, n;
    double *fx = dalloc((size_t)(nid + 1) * njd), *fy = dalloc((size_t)nid * (njd + 1));
    for (n = 1; n <= ntimes; n++) {
      const int nt = ntimes - n;
      if (cubed) { /* :2409-2428: the three cells around a cube corner share their mean */
        if (g->sw_corner) {
          q[A3(1, 1, k)] = (q[A3(1, 1, k)] + q[A3(0, 1, k)] + q[A3(1, 0, k)]) * r3;
          q[A3(0, 1, k)] = q[A3(1, 1, k)];
          q[A3(1, 0, k)] = q[A3(1, 1, k)];
        }
        if (g->se_corner) {
          q[A3(ie, 1, k)] = (q[A3(ie, 1, k)] + q[A3(npx, 1, k)] + q[A3(ie, 0, k)]) * r3;
          q[A3(npx, 1, k)] = q[A3(ie, 1, k)];
          q[A3(ie, 0, k)] = q[A3(ie, 1, k)];
        }
        if (g->ne_corner) {
          q[A3(ie, je, k)] = (q[A3(ie, je, k)] + q[A3(npx, je, k)] + q[A3(ie, npy, k)]) * r3;
          q[A3(npx, je, k)] = q[A3(ie, je, k)];
          q[A3(ie, npy, k)] = q[A3(ie, je, k)];
        }
        if (g->nw_corner) {
          q[A3(1, je, k)] = (q[A3(1, je, k)] + q[A3(0, je, k)] + q[A3(1, npy, k)]) * r3;
          q[A3(0, je, k)] = q[A3(1, je, k)];
          q[A3(1, npy, k)] = q[A3(1, je, k)];
        }
        if (nt > 0) fvo_copy_corners(g, q + A3(isd, jsd, k), 1); /* :2430 */
      }
      for (j = js - nt; j <= je + nt; j++)
        for (i = is - nt; i <= ie + 1 + nt; i++)
          fx[IV(i, j)] = g->del6_v[IV(i, j)] * (q[A3(i - 1, j, k)] - q[A3(i, j, k)]); /* :2440 */
      if (cubed && nt > 0) fvo_copy_corners(g, q + A3(isd, jsd, k), 2); /* :2443 */
      for (j = js - nt; j <= je + 1 + nt; j++)
        for (i = is - nt; i <= ie + nt; i++)
          fy[IU(i, j)] = g->del6_u[IU(i, j)] * (q[A3(i, j - 1, k)] - q[A3(i, j, k)]); /* :2452 */
      for (j = js - nt; j <= je + nt; j++)
        for (i = is - nt; i <= ie + nt; i++)
          q[A3(i, j, k)] = q[A3(i, j, k)] +
                           cd * g->rarea[IA(i, j)] * (fx[IV(i, j)] - fx[IV(i + 1, j)] + fy[IU(i, j)] - fy[IU(i, j + 1)]);
    }
    free(fx);
    free(fy);
  }
  return FVO_OK;
}

/* Application of the dissipative heating after the substep loop, model/dyn_core.F90:1306-1355
 * cappa: A x npz (thermostruct%moist_kappa) or NULL.  pt, delp, heat_source: A x npz; delz, pkz: CC x npz.  heat_source is overwritten
 * with the applied rate where the reference does so. */
int fvo_apply_heat_source(const fvo_grid *g, int npz, int n_con, int hydrostatic, double bdt, double delt_max,
                          double cp_air, double cv_air, double rdgas, double grav, double *pt, double *heat_source,
                          const double *delp, const double *delz, double *pkz, const double *cappa) {
  BOUNDS(g);
  int i, j, k;
  const double rdg = -rdgas / grav, k1k = rdgas / cv_air;
  if (hydrostatic) {
    for (j = js; j <= je; j++)
      for (k = 1; k <= n_con; k++) {
        if (k < 3) {
          for (i = is; i <= ie; i++)
            pt[A3(i, j, k)] = pt[A3(i, j, k)] +
                              heat_source[A3(i, j, k)] / (cp_air * delp[A3(i, j, k)] * pkz[(size_t)(k - 1) * nx * ny + ICC(i, j)]);
        } else {
          for (i = is; i <= ie; i++) {
            const double dtmp = heat_source[A3(i, j, k)] / (cp_air * delp[A3(i, j, k)]);
            const double lim = fmin(fabs(bdt) * delt_max, fabs(dtmp));
            pt[A3(i, j, k)] = pt[A3(i, j, k)] + copysign(lim, dtmp) / pkz[(size_t)(k - 1) * nx * ny + ICC(i, j)];
            heat_source[A3(i, j, k)] = dtmp;
          }
        }
      }
  } else {
    for (k = 1; k <= n_con; k++) {
      double delt = fabs(bdt * delt_max);
      if (k == 1) delt = 0.1 * delt;
      if (k == 2) delt = 0.5 * delt;
      for (j = js; j <= je; j++)
        for (i = is; i <= ie; i++) {
          const size_t c = (size_t)(k - 1) * nx * ny + ICC(i, j);
          if (cappa) { /* thermostruct%moist_kappa (:1338-1340) */
            const double cap = cappa[A3(i, j, k)];
            pkz[c] = fv3_exp(cap / (1. - cap) * fv3_log(rdg * delp[A3(i, j, k)] / delz[c] * pt[A3(i, j, k)]));
          } else {
            pkz[c] = fv3_exp(k1k * fv3_log(rdg * delp[A3(i, j, k)] / delz[c] * pt[A3(i, j, k)]));
          }
          const double dtmp = heat_source[A3(i, j, k)] / (cv_air * delp[A3(i, j, k)]);
          pt[A3(i, j, k)] = pt[A3(i, j, k)] + copysign(fmin(delt, fabs(dtmp)), dtmp) / pkz[c];
          heat_source[A3(i, j, k)] = dtmp;
        }
    }
  }
  return FVO_OK;
}

/* External-mode divergence damping coefficient field divg2 (model/dyn_core.F90:745-747, :791-797, :828-848):
 * ptc(k) = a2b_ord2(delp(k)) (grid_type >= 3: 4-point mean, a2b_edge.F90:427-433) with the delp BEFORE d_sw,
 * divg2 = d_ext*da_min_c * sum_k ptc*vt / sum_k ptc at the corners is:ie+1 x js:je+1; vt = d_sw's delpc output.
 * divg2: A-kind 2-D array (corner indices).  d_ext <= 0: zeros. */
int fvo_divg2_ext(const fvo_grid *g, int npz, double d_ext, const double *delp, const double *vt, double *divg2) {
  BOUNDS(g);
  int i, j, k;
  for (j = jsd; j <= jed; j++)
    for (i = isd; i <= ied; i++) divg2[IA(i, j)] = 0.;
  if (!(d_ext > 0.)) return FVO_OK;
  const double d2_divg = d_ext * g->da_min_c;
  const int cubed = g->grid_type < 3, npx = g->npx, npy = g->npy;
  const double r3 = 1. / 3.;
#define DP(i, j) delp[A3(i, j, k)]
  for (j = js; j <= je + 1; j++)
    for (i = is; i <= ie + 1; i++) {
      double wk = 0., d2 = 0.;
      for (k = 1; k <= npz; k++) {
        double ptc; /* a2b_ord2 (a2b_edge.F90:329-450) of delp at corner (i, j) */
        if (!cubed || (i > 1 && i < npx && j > 1 && j < npy))
          ptc = 0.25 * (DP(i - 1, j - 1) + DP(i, j - 1) + DP(i - 1, j) + DP(i, j));
        else if (i == 1 && j == 1)
          ptc = r3 * (DP(1, 1) + DP(1, 0) + DP(0, 1)); /* :382-385 */
        else if (i == npx && j == 1)
          ptc = r3 * (DP(npx - 1, 1) + DP(npx - 1, 0) + DP(npx, 1));
        else if (i == npx && j == npy)
          ptc = r3 * (DP(npx - 1, npy - 1) + DP(npx, npy - 1) + DP(npx - 1, npy));
        else if (i == 1 && j == npy)
          ptc = r3 * (DP(1, npy - 1) + DP(0, npy - 1) + DP(1, npy));
        else if (i == 1 || i == npx) { /* west / east edge, :388-405 */
          const int ia = (i == 1) ? 0 : npx - 1;
          const double ew = (i == 1) ? g->edge_w[j - 1] : g->edge_e[j - 1]; /* 1-based arrays */
          const double qa = 0.5 * (DP(ia, j - 1) + DP(ia + 1, j - 1)), qb = 0.5 * (DP(ia, j) + DP(ia + 1, j));
          ptc = ew * qa + (1. - ew) * qb;
        } else { /* south / north edge, :408-425 */
          const int ja = (j == 1) ? 0 : npy - 1;
          const double es = (j == 1) ? g->edge_s[i - 1] : g->edge_n[i - 1];
          const double qa = 0.5 * (DP(i - 1, ja) + DP(i - 1, ja + 1)), qb = 0.5 * (DP(i, ja) + DP(i, ja + 1));
          ptc = es * qa + (1. - es) * qb;
        }
        if (k == 1) {
          wk = ptc;
          d2 = wk * vt[A3(i, j, 1)];
        } else {
          wk = wk + ptc;
          d2 = d2 + ptc * vt[A3(i, j, k)];
        }
      }
      divg2[IA(i, j)] = d2_divg * d2 / wk;
    }
  return FVO_OK;
}

/* one_grad_p, hydrostatic form (model/dyn_core.F90:1909-2030, call site :1021): pk = pe**kappa.  pk, gz are replaced
 * by their corner interpolants like in the reference; divg2 as produced by fvo_divg2_ext (zeros when d_ext <= 0). */
static int one_grad_p_any(const fvo_grid *g, int npz, double dt, double ptk, const double *divg2, double *u, double *v,
                          double *pk, double *gz, const double *delp);
int fvo_one_grad_p_hydro(const fvo_grid *g, int npz, double dt, double ptk, const double *divg2, double *u, double *v,
                         double *pk, double *gz) {
  return one_grad_p_any(g, npz, dt, ptk, divg2, u, v, pk, gz, NULL);
}
/* one_grad_p, nonhydrostatic form (hydrostatic = .false.: the call of the nonhydrostatic loop with beta < -0.1, dyn_core.F90:1029-1030,
 * after Riem_Solver3 left the FULL pressure in pkc, :939): pk(:,:,1) = ptop (:1950) and the layer weight wk is a2b_ord4 of delp
 * (:1997) instead of the difference of the corner pk. */
int fvo_one_grad_p_nh(const fvo_grid *g, int npz, double dt, double ptop, const double *divg2, double *u, double *v, double *pk,
                      double *gz, const double *delp) {
  return one_grad_p_any(g, npz, dt, ptop, divg2, u, v, pk, gz, delp);
}
static int one_grad_p_any(const fvo_grid *g, int npz, double dt, double ptk, const double *divg2, double *u, double *v,
                          double *pk, double *gz, const double *delp) {
  BOUNDS(g);
  int k;
  const size_t nA = (size_t)nid * njd, nV = (size_t)(nid + 1) * njd, nU = (size_t)nid * (njd + 1);
  {
    int i, j;
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) pk[A3(i, j, 1)] = ptk;
  }
#pragma omp parallel for schedule(dynamic)
  for (k = 1; k <= npz + 1; k++) {
    double *wk = dalloc(nA);
    if (k >= 2) fvo_a2b_ord4(g, pk + nA * (k - 1), wk, 1);
    fvo_a2b_ord4(g, gz + nA * (k - 1), wk, 1);
    free(wk);
  }
#pragma omp parallel for schedule(dynamic)
  for (k = 1; k <= npz; k++) {
    int i, j;
    double *wk = dalloc(nA);
    if (delp) {   /* :1996-1997: a2b_ord4(delp(k), wk) -- the layer's delp is left as it is */
      double *dcopy = dalloc(nA);
      memcpy(dcopy, delp + nA * (k - 1), nA * sizeof(double));
      fvo_a2b_ord4(g, dcopy, wk, 0);
      free(dcopy);
    } else {
      for (j = js; j <= je + 1; j++)
        for (i = is; i <= ie + 1; i++) wk[IA(i, j)] = pk[A3(i, j, k + 1)] - pk[A3(i, j, k)];
    }
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++) {
        const double wk2 = divg2[IA(i, j)] - divg2[IA(i + 1, j)];
        u[nU * (k - 1) + IU(i, j)] =
            g->rdx[IU(i, j)] *
            (wk2 + u[nU * (k - 1) + IU(i, j)] +
             dt / (wk[IA(i, j)] + wk[IA(i + 1, j)]) *
                 ((gz[A3(i, j, k + 1)] - gz[A3(i + 1, j, k)]) * (pk[A3(i + 1, j, k + 1)] - pk[A3(i, j, k)]) +
                  (gz[A3(i, j, k)] - gz[A3(i + 1, j, k + 1)]) * (pk[A3(i, j, k + 1)] - pk[A3(i + 1, j, k)])));
      }
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++) {
        const double wk1 = divg2[IA(i, j)] - divg2[IA(i, j + 1)];
        v[nV * (k - 1) + IV(i, j)] =
            g->rdy[IV(i, j)] *
            (wk1 + v[nV * (k - 1) + IV(i, j)] +
             dt / (wk[IA(i, j)] + wk[IA(i, j + 1)]) *
                 ((gz[A3(i, j, k + 1)] - gz[A3(i, j + 1, k)]) * (pk[A3(i, j + 1, k + 1)] - pk[A3(i, j, k)]) +
                  (gz[A3(i, j, k)] - gz[A3(i, j + 1, k + 1)]) * (pk[A3(i, j, k + 1)] - pk[A3(i, j + 1, k)])));
      }
    free(wk);
  }
  return FVO_OK;
}

/* adv_pe, model/dyn_core.F90:1529-1632 (called at :1195 on the last substep, use_old_omega): the advective term of omega on a
 * face of the cubed sphere.  pem (is-1:ie+1, npz+1, js-1:je+1) as dyn_core forms it at :409-421 from the delp the substep started
 * from; a2b_ord2 (a2b_edge.F90:329-425, grid_type < 3, not bounded) written out level by level as the reference calls it. */
int fvo_adv_pe(const fvo_grid *g, int km, double ptop, const double *ua, const double *va, const double *delp_before, double *om) {
  BOUNDS(g);
  if (!(g->grid_type < 3) || !g->ec1 || !g->ec2 || !g->en1 || !g->en2) return FVO_ERR_UNSUPPORTED;
  const int npx = g->npx, npy = g->npy;
  const size_t nA = (size_t)nid * njd, nFX = (size_t)(nx + 1) * ny, nFY = (size_t)nx * (ny + 1);
  const double r3 = 1. / 3.;
  int i, j, k, n;
  double *pin = (double *)calloc(nA, sizeof(double)), *pb = (double *)calloc(nA, sizeof(double));
  double *pemc = (double *)calloc(nA, sizeof(double)); /* running pem(:, k+1, :) on the ring (is-1:ie+1, js-1:je+1) */
  for (j = js - 1; j <= je + 1; j++)
    for (i = is - 1; i <= ie + 1; i++) pemc[IA(i, j)] = ptop;
  for (k = 1; k <= km; k++) {
    for (j = js - 1; j <= je + 1; j++)
      for (i = is - 1; i <= ie + 1; i++) {
        pemc[IA(i, j)] = pemc[IA(i, j)] + delp_before[A3(i, j, k)];
        pin[IA(i, j)] = pemc[IA(i, j)];
      }
    /* a2b_ord2(pin, pb) */
    {
      const int is1 = is - 1 > 1 ? is - 1 : 1, js1 = js - 1 > 1 ? js - 1 : 1, is2 = is > 2 ? is : 2, js2 = js > 2 ? js : 2;
      const int ie1 = ie + 1 < npx - 1 ? ie + 1 : npx - 1, je1 = je + 1 < npy - 1 ? je + 1 : npy - 1;
      (void)is1; (void)js1;
      for (j = js2; j <= je1; j++)
        for (i = is2; i <= ie1; i++) pb[IA(i, j)] = 0.25 * (pin[IA(i - 1, j - 1)] + pin[IA(i, j - 1)] + pin[IA(i - 1, j)] + pin[IA(i, j)]);
      if (is == 1 && js == 1) pb[IA(1, 1)] = r3 * (pin[IA(1, 1)] + pin[IA(1, 0)] + pin[IA(0, 1)]);
      if (ie + 1 == npx && js == 1) pb[IA(npx, 1)] = r3 * (pin[IA(npx - 1, 1)] + pin[IA(npx - 1, 0)] + pin[IA(npx, 1)]);
      if (ie + 1 == npx && je + 1 == npy) pb[IA(npx, npy)] = r3 * (pin[IA(npx - 1, npy - 1)] + pin[IA(npx, npy - 1)] + pin[IA(npx - 1, npy)]);
      if (is == 1 && je + 1 == npy) pb[IA(1, npy)] = r3 * (pin[IA(1, npy - 1)] + pin[IA(0, npy - 1)] + pin[IA(1, npy)]);
      if (is == 1)
        for (j = js2; j <= je1; j++)
          pb[IA(1, j)] = g->edge_w[j - 1] * (0.5 * (pin[IA(0, j - 1)] + pin[IA(1, j - 1)])) + (1. - g->edge_w[j - 1]) * (0.5 * (pin[IA(0, j)] + pin[IA(1, j)]));
      if (ie + 1 == npx)
        for (j = js2; j <= je1; j++)
          pb[IA(npx, j)] = g->edge_e[j - 1] * (0.5 * (pin[IA(npx - 1, j - 1)] + pin[IA(npx, j - 1)])) +
                           (1. - g->edge_e[j - 1]) * (0.5 * (pin[IA(npx - 1, j)] + pin[IA(npx, j)]));
      if (js == 1)
        for (i = is2; i <= ie1; i++)
          pb[IA(i, 1)] = g->edge_s[i - 1] * (0.5 * (pin[IA(i - 1, 0)] + pin[IA(i - 1, 1)])) + (1. - g->edge_s[i - 1]) * (0.5 * (pin[IA(i, 0)] + pin[IA(i, 1)]));
      if (je + 1 == npy)
        for (i = is2; i <= ie1; i++)
          pb[IA(i, npy)] = g->edge_n[i - 1] * (0.5 * (pin[IA(i - 1, npy - 1)] + pin[IA(i - 1, npy)])) +
                           (1. - g->edge_n[i - 1]) * (0.5 * (pin[IA(i, npy - 1)] + pin[IA(i, npy)]));
    }
    for (j = js; j <= je; j++)
      for (i = is; i <= ie; i++) {
        const double up = (k == km) ? ua[A3(i, j, km)] : 0.5 * (ua[A3(i, j, k)] + ua[A3(i, j, k + 1)]);
        const double vp = (k == km) ? va[A3(i, j, km)] : 0.5 * (va[A3(i, j, k)] + va[A3(i, j, k + 1)]);
        double v3[3], grad[3];
        for (n = 0; n < 3; n++) {
          v3[n] = up * g->ec1[(size_t)n * nA + IA(i, j)] + vp * g->ec2[(size_t)n * nA + IA(i, j)];
          const double pdx0 = (pb[IA(i, j)] + pb[IA(i + 1, j)]) * g->dx[IU(i, j)] * g->en1[(size_t)n * nFY + IFY(i, j)];
          const double pdx1 = (pb[IA(i, j + 1)] + pb[IA(i + 1, j + 1)]) * g->dx[IU(i, j + 1)] * g->en1[(size_t)n * nFY + IFY(i, j + 1)];
          const double pdy0 = (pb[IA(i, j)] + pb[IA(i, j + 1)]) * g->dy[IV(i, j)] * g->en2[(size_t)n * nFX + IFX(i, j)];
          const double pdy1 = (pb[IA(i + 1, j)] + pb[IA(i + 1, j + 1)]) * g->dy[IV(i + 1, j)] * g->en2[(size_t)n * nFX + IFX(i + 1, j)];
          grad[n] = pdx1 - pdx0 - pdy0 + pdy1;
        }
        om[A3(i, j, k)] = om[A3(i, j, k)] + 0.5 * g->rarea[IA(i, j)] * (v3[0] * grad[0] + v3[1] * grad[1] + v3[2] * grad[2]);
      }
  }
  free(pin); free(pb); free(pemc);
  return FVO_OK;
}


/* split_p_grad (model/dyn_core.F90:1795-1900): nh_p_grad with the hydrostatic part split between two substeps.  du (U x npz), dv
 * (V x npz): dyn_core's saved arrays (zero before the first call, :278-283) */
int fvo_split_p_grad(const fvo_grid *g, int npz, double *u, double *v, double *pp, double *gz, double *delp, double *pk, double beta,
                     double dt, double top_value, double *du, double *dv) {
  BOUNDS(g);
  int k;
  const size_t nA = (size_t)nid * njd, nV = (size_t)(nid + 1) * njd, nU = (size_t)nid * (njd + 1);
  const double alpha = 1. - beta;
#pragma omp parallel for schedule(dynamic)
  for (k = 1; k <= npz + 1; k++) {
    int i, j;
    double *wk1 = dalloc(nA);
    if (k == 1) {
      for (j = js; j <= je + 1; j++)
        for (i = is; i <= ie + 1; i++) {
          pp[A3(i, j, 1)] = 0.;
          pk[A3(i, j, 1)] = top_value;
        }
    } else {
      fvo_a2b_ord4(g, pp + nA * (k - 1), wk1, 1);
      fvo_a2b_ord4(g, pk + nA * (k - 1), wk1, 1);
    }
    fvo_a2b_ord4(g, gz + nA * (k - 1), wk1, 1);
    free(wk1);
  }
#pragma omp parallel for schedule(dynamic)
  for (k = 1; k <= npz; k++) {
    int i, j;
    double *wk1 = dalloc(nA), *wk = dalloc(nA);
    fvo_a2b_ord4(g, delp + nA * (k - 1), wk1, 0);
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) wk[IA(i, j)] = pk[A3(i, j, k + 1)] - pk[A3(i, j, k)];
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++) {
        double *uu = &u[nU * (k - 1) + IU(i, j)], *dd = &du[nU * (k - 1) + IU(i, j)];
        *uu = *uu + beta * *dd;
        *dd = dt / (wk[IA(i, j)] + wk[IA(i + 1, j)]) *
              ((gz[A3(i, j, k + 1)] - gz[A3(i + 1, j, k)]) * (pk[A3(i + 1, j, k + 1)] - pk[A3(i, j, k)]) +
               (gz[A3(i, j, k)] - gz[A3(i + 1, j, k + 1)]) * (pk[A3(i, j, k + 1)] - pk[A3(i + 1, j, k)]));
        *uu = (*uu + alpha * *dd +
               dt / (wk1[IA(i, j)] + wk1[IA(i + 1, j)]) *
                   ((gz[A3(i, j, k + 1)] - gz[A3(i + 1, j, k)]) * (pp[A3(i + 1, j, k + 1)] - pp[A3(i, j, k)]) +
                    (gz[A3(i, j, k)] - gz[A3(i + 1, j, k + 1)]) * (pp[A3(i, j, k + 1)] - pp[A3(i + 1, j, k)]))) *
              g->rdx[IU(i, j)];
      }
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++) {
        double *vv = &v[nV * (k - 1) + IV(i, j)], *dd = &dv[nV * (k - 1) + IV(i, j)];
        *vv = *vv + beta * *dd;
        *dd = dt / (wk[IA(i, j)] + wk[IA(i, j + 1)]) *
              ((gz[A3(i, j, k + 1)] - gz[A3(i, j + 1, k)]) * (pk[A3(i, j + 1, k + 1)] - pk[A3(i, j, k)]) +
               (gz[A3(i, j, k)] - gz[A3(i, j + 1, k + 1)]) * (pk[A3(i, j, k + 1)] - pk[A3(i, j + 1, k)]));
        *vv = (*vv + alpha * *dd +
               dt / (wk1[IA(i, j)] + wk1[IA(i, j + 1)]) *
                   ((gz[A3(i, j, k + 1)] - gz[A3(i, j + 1, k)]) * (pp[A3(i, j + 1, k + 1)] - pp[A3(i, j, k)]) +
                    (gz[A3(i, j, k)] - gz[A3(i, j + 1, k + 1)]) * (pp[A3(i, j, k + 1)] - pp[A3(i, j + 1, k)]))) *
              g->rdy[IV(i, j)];
      }
    free(wk1);
    free(wk);
  }
  return 0;
}

/* grad1_p_update (model/dyn_core.F90:2033-2116) */
int fvo_grad1_p_update(const fvo_grid *g, int npz, const double *divg2, double *u, double *v, double *pk, double *gz, double dt,
                       double ptk, double beta, double *du, double *dv) {
  BOUNDS(g);
  int k;
  const size_t nA = (size_t)nid * njd, nV = (size_t)(nid + 1) * njd, nU = (size_t)nid * (njd + 1);
  const double alpha = 1. - beta;
  {
    int i, j;
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) pk[A3(i, j, 1)] = ptk;
  }
#pragma omp parallel for schedule(dynamic)
  for (k = 1; k <= npz + 1; k++) {
    double *wk = dalloc(nA);
    if (k >= 2) fvo_a2b_ord4(g, pk + nA * (k - 1), wk, 1);
    fvo_a2b_ord4(g, gz + nA * (k - 1), wk, 1);
    free(wk);
  }
#pragma omp parallel for schedule(dynamic)
  for (k = 1; k <= npz; k++) {
    int i, j;
    double *wk = dalloc(nA);
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) wk[IA(i, j)] = pk[A3(i, j, k + 1)] - pk[A3(i, j, k)];
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++) {
        double *uu = &u[nU * (k - 1) + IU(i, j)], *dd = &du[nU * (k - 1) + IU(i, j)];
        *uu = *uu + beta * *dd;
        *dd = dt / (wk[IA(i, j)] + wk[IA(i + 1, j)]) *
              ((gz[A3(i, j, k + 1)] - gz[A3(i + 1, j, k)]) * (pk[A3(i + 1, j, k + 1)] - pk[A3(i, j, k)]) +
               (gz[A3(i, j, k)] - gz[A3(i + 1, j, k + 1)]) * (pk[A3(i, j, k + 1)] - pk[A3(i + 1, j, k)]));
        *uu = (*uu + divg2[IA(i, j)] - divg2[IA(i + 1, j)] + alpha * *dd) * g->rdx[IU(i, j)];
      }
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++) {
        double *vv = &v[nV * (k - 1) + IV(i, j)], *dd = &dv[nV * (k - 1) + IV(i, j)];
        *vv = *vv + beta * *dd;
        *dd = dt / (wk[IA(i, j)] + wk[IA(i, j + 1)]) *
              ((gz[A3(i, j, k + 1)] - gz[A3(i, j + 1, k)]) * (pk[A3(i, j + 1, k + 1)] - pk[A3(i, j, k)]) +
               (gz[A3(i, j, k)] - gz[A3(i, j + 1, k + 1)]) * (pk[A3(i, j, k + 1)] - pk[A3(i, j + 1, k)]));
        *vv = (*vv + divg2[IA(i, j)] - divg2[IA(i, j + 1)] + alpha * *dd) * g->rdy[IV(i, j)];
      }
    free(wk);
  }
  return 0;
}
