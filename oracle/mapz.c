/*
 * oracle/mapz.c -- CPU oracle (test infrastructure, see fvo.h) for the vertical remap:
 * model/fv_operators.F90 (scalar_profile :546-916, cs_profile :919-1300, cs_limiters :1303-1378,
 * map_scalar :40-134, map1_ppm :137-229, mapn_tracer :234-348, map1_q2 :352-443) and
 * model/fv_mapz.F90 Lagrangian_to_Eulerian :56-845.
 * Branches restated: remap_te both ways (map_scalar and map1_cubic, fv_operators.F90:1897-2096), moist_kappa / use_cond both ways (moist_cv: fv_thermodynamics.F90:250-325),
 * consv = 0 (no energy fixer),
 * fill = .false., do_intermediate_phys = .false.; kord 8..15 (scalar_profile / cs_profile) and kord <= 7 (ppm_profile); iv in {-2,-1,0,1}.
 * (iv = -3, i.e. kord_wz < 0, is not restated: the reference's back-substitution reads gam(i,km), which
 * that branch never sets -- fv_operators.F90:974-993,1012-1016.)
 */
#include "fvo.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static const double r3 = 1. / 3., r23 = 2. / 3., r12 = 1. / 12.;

static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin3(double a, double b, double c) { return dmin(dmin(a, b), c); }
static inline double dmax3(double a, double b, double c) { return dmax(dmax(a, b), c); }
static double *dalloc(size_t n) { return (double *)calloc(n, sizeof(double)); }

/* a4 is stored [n][k] with n = 1..4, k = 1..km for ONE column: A4(n,k) */
#define A4(n, k) a4[((n)-1) * (km + 2) + (k)]

/* cs_limiters for one cell, fv_operators.F90:1303-1378 */
static void cs_limiter_cell(int extm, double *a1, double *a2, double *a3, double *a4v, int iv) {
  double da1, da2, a6da;
  if (iv == 0) {
    if (*a1 <= 0.) {
      *a2 = *a1;
      *a3 = *a1;
      *a4v = 0.;
    } else {
      if (fabs(*a3 - *a2) < -*a4v) {
        if ((*a1 + 0.25 * ((*a3 - *a2) * (*a3 - *a2)) / *a4v + *a4v * r12) < 0.) {
          if (*a1 < *a3 && *a1 < *a2) {
            *a3 = *a1;
            *a2 = *a1;
            *a4v = 0.;
          } else if (*a3 > *a2) {
            *a4v = 3. * (*a2 - *a1);
            *a3 = *a2 - *a4v;
          } else {
            *a4v = 3. * (*a3 - *a1);
            *a2 = *a3 - *a4v;
          }
        }
      }
    }
  } else if (iv == 1) {
    if ((*a1 - *a2) * (*a1 - *a3) >= 0.) {
      *a2 = *a1;
      *a3 = *a1;
      *a4v = 0.;
    } else {
      da1 = *a3 - *a2;
      da2 = da1 * da1;
      a6da = *a4v * da1;
      if (a6da < -da2) {
        *a4v = 3. * (*a2 - *a1);
        *a3 = *a2 - *a4v;
      } else if (a6da > da2) {
        *a4v = 3. * (*a3 - *a1);
        *a2 = *a3 - *a4v;
      }
    }
  } else {
    if (extm) {
      *a2 = *a1;
      *a3 = *a1;
      *a4v = 0.;
    } else {
      da1 = *a3 - *a2;
      da2 = da1 * da1;
      a6da = *a4v * da1;
      if (a6da < -da2) {
        *a4v = 3. * (*a2 - *a1);
        *a3 = *a2 - *a4v;
      } else if (a6da > da2) {
        *a4v = 3. * (*a3 - *a1);
        *a2 = *a3 - *a4v;
      }
    }
  }
}
#define LIM(k, mode) cs_limiter_cell(extm[k], &A4(1, k), &A4(2, k), &A4(3, k), &A4(4, k), mode)

/* ppm_limiters for one cell, fv_operators.F90:1642-1723 (lmt 0: standard PPM, 1: Lin's full monotonicity, 2: positive
 * definite, 3: nothing) */
static void ppm_limiter_cell(double dm, double *a1, double *a2, double *a3, double *a6, int lmt) {
  if (lmt == 3) return;
  if (lmt == 0) {
    if (dm == 0.) {
      *a2 = *a1; *a3 = *a1; *a6 = 0.;
    } else {
      const double da1 = *a3 - *a2, da2 = da1 * da1, a6da = *a6 * da1;
      if (a6da < -da2) {
        *a6 = 3. * (*a2 - *a1); *a3 = *a2 - *a6;
      } else if (a6da > da2) {
        *a6 = 3. * (*a3 - *a1); *a2 = *a3 - *a6;
      }
    }
  } else if (lmt == 1) {
    const double qmp = 2. * dm;
    *a2 = *a1 - copysign(dmin(fabs(qmp), fabs(*a2 - *a1)), qmp);
    *a3 = *a1 + copysign(dmin(fabs(qmp), fabs(*a3 - *a1)), qmp);
    *a6 = 3. * (2. * *a1 - (*a2 + *a3));
  } else if (lmt == 2) {
    if (fabs(*a3 - *a2) < -*a6) {
      const double fmin_ = *a1 + 0.25 * ((*a3 - *a2) * (*a3 - *a2)) / *a6 + *a6 * r12;
      if (fmin_ < 0.) {
        if (*a1 < *a3 && *a1 < *a2) {
          *a3 = *a1; *a2 = *a1; *a6 = 0.;
        } else if (*a3 > *a2) {
          *a6 = 3. * (*a2 - *a1); *a3 = *a2 - *a6;
        } else {
          *a6 = 3. * (*a3 - *a1); *a2 = *a3 - *a6;
        }
      }
    }
  }
}
#define PLIM(k, lmt) ppm_limiter_cell(dc[k], &A4(1, k), &A4(2, k), &A4(3, k), &A4(4, k), lmt)

/* ppm_profile for one column, fv_operators.F90:1382-1639 (what the map routines call for kord <= 7, :87-91, :182-186, :395-399,
 * :480-484): 4th-order edge values from limited slopes, area-preserving cubics at the top and the bottom, Huynh's 2nd
 * constraint for kord >= 7, ppm_limiters otherwise.  Needs km >= 5 like the reference's loops. */
static int ppm_profile_column(double *a4, const double *delp, int km, int iv, int kord) {
  int k;
  const int km1 = km - 1;
  if (km < 5) return FVO_ERR_UNSUPPORTED;
  double *dc = dalloc(km + 3), *h2 = dalloc(km + 3), *delq = dalloc(km + 3), *df2 = dalloc(km + 3), *d4 = dalloc(km + 3);
  double c1, c2, c3, a1, a2, d1, d2, qm, dq, qmp, lac, pmp;
  for (k = 2; k <= km; k++) {
    delq[k - 1] = A4(1, k) - A4(1, k - 1);
    d4[k] = delp[k - 1] + delp[k];
  }
  for (k = 2; k <= km1; k++) {
    c1 = (delp[k - 1] + 0.5 * delp[k]) / d4[k + 1];
    c2 = (delp[k + 1] + 0.5 * delp[k]) / d4[k];
    df2[k] = delp[k] * (c1 * delq[k] + c2 * delq[k - 1]) / (d4[k] + delp[k + 1]);
    dc[k] = copysign(dmin3(fabs(df2[k]), dmax3(A4(1, k - 1), A4(1, k), A4(1, k + 1)) - A4(1, k),
                           A4(1, k) - dmin3(A4(1, k - 1), A4(1, k), A4(1, k + 1))), df2[k]);
  }
  for (k = 3; k <= km1; k++) { /* 4th order interpolation of the provisional cell edge value */
    c1 = delq[k - 1] * delp[k - 1] / d4[k];
    a1 = d4[k - 1] / (d4[k] + delp[k - 1]);
    a2 = d4[k + 1] / (d4[k] + delp[k]);
    A4(2, k) = A4(1, k - 1) + c1 + 2. / (d4[k - 1] + d4[k + 1]) * (delp[k] * (c1 * (a1 - a2) + a2 * dc[k - 1]) - delp[k - 1] * a1 * dc[k]);
  }
  /* top: area preserving cubic with 2nd deriv. = 0 at the boundary */
  d1 = delp[1];
  d2 = delp[2];
  qm = (d2 * A4(1, 1) + d1 * A4(1, 2)) / (d1 + d2);
  dq = 2. * (A4(1, 2) - A4(1, 1)) / (d1 + d2);
  c1 = 4. * (A4(2, 3) - qm - d2 * dq) / (d2 * (2. * d2 * d2 + d1 * (d2 + 3. * d1)));
  c3 = dq - 0.5 * c1 * (d2 * (5. * d1 + d2) - 3. * d1 * d1);
  A4(2, 2) = qm - 0.25 * c1 * d1 * d2 * (d2 + 3. * d1);
  A4(2, 1) = d1 * (2. * c1 * (d1 * d1) - c3) + A4(2, 2);
  A4(2, 2) = dmax(A4(2, 2), dmin(A4(1, 1), A4(1, 2)));
  A4(2, 2) = dmin(A4(2, 2), dmax(A4(1, 1), A4(1, 2)));
  dc[1] = 0.5 * (A4(2, 2) - A4(1, 1));
  if (iv == 0) {
    A4(2, 1) = dmax(0., A4(2, 1));
    A4(2, 2) = dmax(0., A4(2, 2));
  } else if (iv == -1) {
    if (A4(2, 1) * A4(1, 1) <= 0.) A4(2, 1) = 0.;
  } else if (abs(iv) == 2) {
    A4(2, 1) = A4(1, 1);
    A4(3, 1) = A4(1, 1);
  }
  /* bottom */
  d1 = delp[km];
  d2 = delp[km1];
  qm = (d2 * A4(1, km) + d1 * A4(1, km1)) / (d1 + d2);
  dq = 2. * (A4(1, km1) - A4(1, km)) / (d1 + d2);
  c1 = (A4(2, km1) - qm - d2 * dq) / (d2 * (2. * d2 * d2 + d1 * (d2 + 3. * d1)));
  c3 = dq - 2.0 * c1 * (d2 * (5. * d1 + d2) - 3. * d1 * d1);
  A4(2, km) = qm - c1 * d1 * d2 * (d2 + 3. * d1);
  A4(3, km) = d1 * (8. * c1 * (d1 * d1) - c3) + A4(2, km);
  A4(2, km) = dmax(A4(2, km), dmin(A4(1, km), A4(1, km1)));
  A4(2, km) = dmin(A4(2, km), dmax(A4(1, km), A4(1, km1)));
  dc[km] = 0.5 * (A4(1, km) - A4(2, km));
  if (iv == 0) {
    A4(2, km) = dmax(0., A4(2, km));
    A4(3, km) = dmax(0., A4(3, km));
  } else if (iv < 0) {
    if (A4(1, km) * A4(3, km) <= 0.) A4(3, km) = 0.;
  }
  for (k = 1; k <= km1; k++) A4(3, k) = A4(2, k + 1);
  /* top 2 and bottom 2 layers always use monotonic mapping */
  for (k = 1; k <= 2; k++) {
    A4(4, k) = 3. * (2. * A4(1, k) - (A4(2, k) + A4(3, k)));
    PLIM(k, 0);
  }
  if (kord >= 7) { /* Huynh's 2nd constraint */
    for (k = 2; k <= km1; k++)
      h2[k] = 2. * (dc[k + 1] / delp[k + 1] - dc[k - 1] / delp[k - 1]) / (delp[k] + 0.5 * (delp[k - 1] + delp[k + 1])) * (delp[k] * delp[k]);
    const double fac = 1.5;
    for (k = 3; k <= km - 2; k++) {
      pmp = 2. * dc[k];
      qmp = A4(1, k) + pmp;
      lac = A4(1, k) + fac * h2[k - 1] + dc[k];
      A4(3, k) = dmin(dmax(A4(3, k), dmin3(A4(1, k), qmp, lac)), dmax3(A4(1, k), qmp, lac));
      qmp = A4(1, k) - pmp;
      lac = A4(1, k) + fac * h2[k + 1] - dc[k];
      A4(2, k) = dmin(dmax(A4(2, k), dmin3(A4(1, k), qmp, lac)), dmax3(A4(1, k), qmp, lac));
      A4(4, k) = 3. * (2. * A4(1, k) - (A4(2, k) + A4(3, k)));
      if (iv == 0 && kord >= 6) PLIM(k, 2);
    }
  } else {
    int lmt = kord - 3;
    if (lmt < 0) lmt = 0;
    if (iv == 0 && lmt > 2) lmt = 2;
    for (k = 3; k <= km - 2; k++) {
      if (kord != 4) A4(4, k) = 3. * (2. * A4(1, k) - (A4(2, k) + A4(3, k)));
      if (kord != 6) PLIM(k, lmt);
    }
  }
  for (k = km1; k <= km; k++) {
    A4(4, k) = 3. * (2. * A4(1, k) - (A4(2, k) + A4(3, k)));
    PLIM(k, 0);
  }
  free(dc); free(h2); free(delq); free(df2); free(d4);
  return FVO_OK;
}

/* scalar_profile (is_scalar=1, :546-916) / cs_profile (is_scalar=0, :919-1300) for one column.
 * delp[1..km], a4 as above (A4(1,k) on entry). Returns 0 or FVO_ERR_UNSUPPORTED. */
int fvo_profile_column(int is_scalar, double qs, double *a4, const double *delp, int km, int iv, int kord, double qmin) {
  int k;
  const int ak = abs(kord);
  if (!(iv == -2 || iv == -1 || iv == 0 || iv == 1)) return FVO_ERR_UNSUPPORTED;
  if (kord <= 7) return ppm_profile_column(a4, delp, km, iv, kord); /* "if (kord > 7) ... else ppm_profile", the SIGNED kord */
  if (!(ak >= 8 && ak <= 15)) return FVO_ERR_UNSUPPORTED;
  double *gam = dalloc(km + 3), *q = dalloc(km + 3);
  unsigned char *extm = (unsigned char *)calloc(km + 3, 1), *ext5 = (unsigned char *)calloc(km + 3, 1),
                *ext6 = (unsigned char *)calloc(km + 3, 1);
  double bet, a_bot, grat, d4 = 0., pmp_1, lac_1, pmp_2, lac_2, x0, x1;
  if (iv == -2) { /* :572-595 / :941-964 */
    gam[2] = 0.5;
    q[1] = 1.5 * A4(1, 1);
    for (k = 2; k <= km - 1; k++) {
      grat = delp[k - 1] / delp[k];
      bet = 2. + grat + grat - gam[k];
      q[k] = (3. * (A4(1, k - 1) + A4(1, k)) - q[k - 1]) / bet;
      gam[k + 1] = grat / bet;
    }
    grat = delp[km - 1] / delp[km];
    q[km] = (3. * (A4(1, km - 1) + A4(1, km)) - grat * qs - q[km - 1]) / (2. + grat + grat - gam[km]);
    q[km + 1] = qs;
    for (k = km - 1; k >= 1; k--) q[k] = q[k] - gam[k + 1] * q[k + 1];
  } else { /* :597-623 / :967-1016 */
    grat = delp[2] / delp[1];
    bet = grat * (grat + 0.5);
    q[1] = ((grat + grat) * (grat + 1.) * A4(1, 1) + A4(1, 2)) / bet;
    gam[1] = (1. + grat * (grat + 1.5)) / bet;
    for (k = 2; k <= km; k++) {
      d4 = delp[k - 1] / delp[k];
      bet = 2. + d4 + d4 - gam[k - 1];
      q[k] = (3. * (A4(1, k - 1) + d4 * A4(1, k)) - q[k - 1]) / bet;
      gam[k] = d4 / bet;
    }
    a_bot = 1. + d4 * (d4 + 1.5);
    q[km + 1] = (2. * d4 * (d4 + 1.) * A4(1, km) + A4(1, km - 1) - a_bot * q[km]) / (d4 * (d4 + 0.5) - a_bot * gam[km]);
    for (k = km; k >= 1; k--) q[k] = q[k] - gam[k] * q[k + 1];
  }
  /* large-scale constraints, :643-680 / :1037-1073 */
  q[2] = dmin(q[2], dmax(A4(1, 1), A4(1, 2)));
  q[2] = dmax(q[2], dmin(A4(1, 1), A4(1, 2)));
  for (k = 2; k <= km; k++) gam[k] = A4(1, k) - A4(1, k - 1);
  for (k = 3; k <= km - 1; k++) {
    if (ak >= 14 || gam[k - 1] * gam[k + 1] > 0.) {
      q[k] = dmin(q[k], dmax(A4(1, k - 1), A4(1, k)));
      q[k] = dmax(q[k], dmin(A4(1, k - 1), A4(1, k)));
    } else {
      if (gam[k - 1] > 0.) {
        q[k] = dmax(q[k], dmin(A4(1, k - 1), A4(1, k)));
      } else {
        q[k] = dmin(q[k], dmax(A4(1, k - 1), A4(1, k)));
        if (iv == 0) q[k] = dmax(0., q[k]);
      }
    }
  }
  q[km] = dmin(q[km], dmax(A4(1, km - 1), A4(1, km)));
  q[km] = dmax(q[km], dmin(A4(1, km - 1), A4(1, km)));
  for (k = 1; k <= km; k++) {
    A4(2, k) = q[k];
    A4(3, k) = q[k + 1];
  }
  for (k = 1; k <= km; k++) { /* :693-712 / :1082-1101 */
    if (k == 1 || k == km)
      extm[k] = (A4(2, k) - A4(1, k)) * (A4(3, k) - A4(1, k)) > 0.;
    else
      extm[k] = gam[k] * gam[k + 1] < 0.;
    if (ak > 9) {
      x0 = 2. * A4(1, k) - (A4(2, k) + A4(3, k));
      x1 = fabs(A4(2, k) - A4(3, k));
      A4(4, k) = 3. * x0;
      ext5[k] = fabs(x0) > x1;
      ext6[k] = fabs(A4(4, k)) > x1;
    }
  }
  /* top, :718-747 / :1109-1137 */
  if (iv == 0) A4(2, 1) = dmax(0., A4(2, 1));
  if (iv == -1) {
    if (A4(2, 1) * A4(1, 1) <= 0.) A4(2, 1) = 0.;
  }
  A4(4, 1) = 3. * (2. * A4(1, 1) - (A4(2, 1) + A4(3, 1)));
  LIM(1, 1);
  A4(4, 2) = 3. * (2. * A4(1, 2) - (A4(2, 2) + A4(3, 2)));
  LIM(2, 2);
  /* interior, :752-892 / :1142-1276 */
  for (k = 3; k <= km - 2; k++) {
#define HUYNH()                                                                                          \
  do {                                                                                                   \
    pmp_1 = A4(1, k) - 2. * gam[k + 1];                                                                  \
    lac_1 = pmp_1 + 1.5 * gam[k + 2];                                                                    \
    A4(2, k) = dmin(dmax(A4(2, k), dmin3(A4(1, k), pmp_1, lac_1)), dmax3(A4(1, k), pmp_1, lac_1));       \
    pmp_2 = A4(1, k) + 2. * gam[k];                                                                      \
    lac_2 = pmp_2 - 1.5 * gam[k - 1];                                                                    \
    A4(3, k) = dmin(dmax(A4(3, k), dmin3(A4(1, k), pmp_2, lac_2)), dmax3(A4(1, k), pmp_2, lac_2));       \
  } while (0)
    if (ak <= 8) {
      HUYNH();
      A4(4, k) = 3. * (2. * A4(1, k) - (A4(2, k) + A4(3, k)));
    } else if (ak == 9) {
      if ((extm[k] && extm[k - 1]) || (extm[k] && extm[k + 1]) || (is_scalar && extm[k] && A4(1, k) < qmin)) {
        A4(2, k) = A4(1, k);
        A4(3, k) = A4(1, k);
        A4(4, k) = 0.;
      } else {
        /* scalar_profile: 3*(2a-(l+r)) (:789,800); cs_profile: 6a-3(l+r) (:1173,1184) */
        A4(4, k) = is_scalar ? 3. * (2. * A4(1, k) - (A4(2, k) + A4(3, k))) : 6. * A4(1, k) - 3. * (A4(2, k) + A4(3, k));
        if (fabs(A4(4, k)) > fabs(A4(2, k) - A4(3, k))) {
          HUYNH();
          A4(4, k) = is_scalar ? 3. * (2. * A4(1, k) - (A4(2, k) + A4(3, k))) : 6. * A4(1, k) - 3. * (A4(2, k) + A4(3, k));
        }
      }
    } else if (ak == 10) {
      if (extm[k]) {
        if ((is_scalar && A4(1, k) < qmin) || extm[k - 1] || extm[k + 1]) {
          A4(2, k) = A4(1, k);
          A4(3, k) = A4(1, k);
          A4(4, k) = 0.;
        } else {
          A4(4, k) = 6. * A4(1, k) - 3. * (A4(2, k) + A4(3, k));
        }
      } else {
        A4(4, k) = 6. * A4(1, k) - 3. * (A4(2, k) + A4(3, k));
        if (fabs(A4(4, k)) > fabs(A4(2, k) - A4(3, k))) {
          HUYNH();
          A4(4, k) = 6. * A4(1, k) - 3. * (A4(2, k) + A4(3, k));
        }
      }
    } else if (ak == 11) {
      if (ext5[k] && (ext5[k - 1] || ext5[k + 1] || (is_scalar && A4(1, k) < qmin))) {
        A4(2, k) = A4(1, k);
        A4(3, k) = A4(1, k);
        A4(4, k) = 0.;
      } else {
        A4(4, k) = 3. * (2. * A4(1, k) - (A4(2, k) + A4(3, k)));
      }
    } else if (ak == 12) { /* post-AM4 case 10, :835-866 / :1225-1256 */
      if (ext5[k]) {
        if (ext5[k - 1] || ext5[k + 1]) {
          A4(2, k) = A4(1, k);
          A4(3, k) = A4(1, k);
        } else if (ext6[k - 1] || ext6[k + 1]) {
          HUYNH();
        }
      } else if (ext6[k]) {
        if (ext5[k - 1] || ext5[k + 1]) HUYNH();
      }
      A4(4, k) = 3. * (2. * A4(1, k) - (A4(2, k) + A4(3, k)));
    } else if (ak == 14) { /* strict monotonicity constraint, :883-884 / :1267-1268 (a4(4) = 3*x0 from :703-711) */
      LIM(k, 2);
    } else if (ak == 15) { /* :885-886 / :1269-1270 */
      LIM(k, 1);
    } else { /* 13 */
      A4(4, k) = 3. * (2. * A4(1, k) - (A4(2, k) + A4(3, k)));
    }
    if (iv == 0 && ak <= 13) LIM(k, 0);
  }
  /* bottom, :897-914 / :1281-1298 */
  if (iv == 0) A4(3, km) = dmax(0., A4(3, km));
  if (iv == -1) {
    if (A4(3, km) * A4(1, km) <= 0.) A4(3, km) = 0.;
  }
  for (k = km - 1; k <= km; k++) {
    A4(4, k) = 3. * (2. * A4(1, k) - (A4(2, k) + A4(3, k)));
    if (k == km - 1) LIM(k, 2);
    if (k == km) LIM(k, 1);
  }
  free(gam); free(q); free(extm); free(ext5); free(ext6);
  return FVO_OK;
}

/* The search-and-integrate loop shared by map_scalar / map1_ppm / map1_q2 (:93-132, :188-227, :402-441) for one
 * column.  pe1[1..km+1], pe2[1..kn+1], dp1[1..km]; q2[1..kn] out; div: 0 -> /(pe2(k+1)-pe2(k)), else /dp2[k].
 * tracer_form: the factored expressions of mapn_tracer (:283-333). */
static void map_column(int km, int kn, const double *pe1, const double *pe2, const double *dp1, const double *a4,
                       double *q2, const double *dp2, int tracer_form) {
  int k, l, m, k0 = 1;
  double pl, pr, qsum = 0., dp, esl, fac1, fac2;
  for (k = 1; k <= kn; k++) {
    int done = 0;
    for (l = k0; l <= km && !done; l++) {
      if (pe2[k] >= pe1[l] && pe2[k] <= pe1[l + 1]) {
        pl = (pe2[k] - pe1[l]) / dp1[l];
        if (pe2[k + 1] <= pe1[l + 1]) {
          pr = (pe2[k + 1] - pe1[l]) / dp1[l];
          if (tracer_form) {
            fac1 = pr + pl;
            fac2 = r3 * (pr * fac1 + pl * pl);
            fac1 = 0.5 * fac1;
            q2[k] = A4(2, l) + (A4(4, l) + A4(3, l) - A4(2, l)) * fac1 - A4(4, l) * fac2;
          } else {
            q2[k] = A4(2, l) + 0.5 * (A4(4, l) + A4(3, l) - A4(2, l)) * (pr + pl) - A4(4, l) * r3 * (pr * (pr + pl) + pl * pl);
          }
          k0 = l;
          done = 2; /* goto 555 */
        } else {
          if (tracer_form) {
            dp = pe1[l + 1] - pe2[k];
            fac1 = 1. + pl;
            fac2 = r3 * (1. + pl * fac1);
            fac1 = 0.5 * fac1;
            qsum = dp * (A4(2, l) + (A4(4, l) + A4(3, l) - A4(2, l)) * fac1 - A4(4, l) * fac2);
          } else {
            qsum = (pe1[l + 1] - pe2[k]) *
                   (A4(2, l) + 0.5 * (A4(4, l) + A4(3, l) - A4(2, l)) * (1. + pl) - A4(4, l) * (r3 * (1. + pl * (1. + pl))));
          }
          for (m = l + 1; m <= km; m++) {
            if (pe2[k + 1] > pe1[m + 1]) {
              qsum = qsum + dp1[m] * A4(1, m);
            } else {
              dp = pe2[k + 1] - pe1[m];
              esl = dp / dp1[m];
              if (tracer_form) {
                fac1 = 0.5 * esl;
                fac2 = 1. - r23 * esl;
                qsum = qsum + dp * (A4(2, m) + fac1 * (A4(3, m) - A4(2, m) + A4(4, m) * fac2));
              } else {
                qsum = qsum + dp * (A4(2, m) + 0.5 * esl * (A4(3, m) - A4(2, m) + A4(4, m) * (1. - r23 * esl)));
              }
              k0 = m;
              break;
            }
          }
          done = 1; /* goto 123 */
        }
      }
    }
    if (done != 2) q2[k] = dp2 ? qsum / dp2[k] : qsum / (pe2[k + 1] - pe2[k]);
  }
}

/* Remap one column: which = 0 map_scalar (scalar_profile), 1 map1_ppm (cs_profile), 2 map1_q2 (scalar_profile,
 * /dp2), 3 mapn_tracer arithmetic.  q1[1..km] in, q2[1..kn] out (may alias q1: a4(1,:) is a copy). */
int fvo_remap_column(int which, int km, const double *pe1, const double *pe2, const double *q1, double *q2, double qs,
                     int iv, int kord, double qmin) {
  int k, rc;
  double *a4 = dalloc(4 * (size_t)(km + 2)), *dp1 = dalloc(km + 2), *dp2 = dalloc(km + 2);
  for (k = 1; k <= km; k++) {
    dp1[k] = pe1[k + 1] - pe1[k];
    dp2[k] = pe2[k + 1] - pe2[k];
    A4(1, k) = q1[k];
  }
  rc = fvo_profile_column(which != 1, qs, a4, dp1, km, iv, kord, qmin);
  if (!rc) map_column(km, km, pe1, pe2, dp1, a4, q2, which >= 2 ? dp2 : NULL, which == 3);
  free(a4); free(dp1); free(dp2);
  return rc;
}

/* ---------------------------------------------------------------------------------------------------
 * Lagrangian_to_Eulerian, fv_mapz.F90:56-845 (branches listed in the file header).
 * ps: A; pe (is-1:ie+1, km+1, js-1:je+1); delp, pt, w, omga: A x km; q: A x km x nq; u: U x km; v: V x km;
 * delz, pkz: CC x km; pk: CC x (km+1); peln (is:ie, km+1, js:je); ws: CC; ak, bk: km+1.
 * ------------------------------------------------------------------------------------------------- */


void fvo_fillz_column(int km, double *q, const double *dp) {
  int k, zfix = 0;
  double dq;
  if (q[1] < 0.) { /* top layer, :67-73 */
    q[2] = q[2] + q[1] * dp[1] / dp[2];
    q[1] = 0.;
  }
  for (k = 2; k <= km - 1; k++) { /* interior, :76-96 */
    if (q[k] < 0.) {
      zfix = 1;
      if (q[k - 1] > 0.) { /* borrow from above */
        dq = dmin(q[k - 1] * dp[k - 1], -q[k] * dp[k]);
        q[k - 1] = q[k - 1] - dq / dp[k - 1];
        q[k] = q[k] + dq / dp[k];
      }
      if (q[k] < 0.0 && q[k + 1] > 0.) { /* borrow from below */
        dq = dmin(q[k + 1] * dp[k + 1], -q[k] * dp[k]);
        q[k + 1] = q[k + 1] - dq / dp[k + 1];
        q[k] = q[k] + dq / dp[k];
      }
    }
  }
  k = km; /* bottom layer, :99-110 */
  if (q[k] < 0. && q[k - 1] > 0.) {
    const double qup = q[k - 1] * dp[k - 1], qly = -q[k] * dp[k], dup = dmin(qly, qup);
    zfix = 1;
    q[k - 1] = q[k - 1] - dup / dp[k - 1];
    q[k] = q[k] + dup / dp[k];
  }
  if (zfix) { /* final check and non-local fix, :113-133 */
    double sum0 = 0., sum1 = 0., fac;
    for (k = 2; k <= km; k++) sum0 = sum0 + q[k] * dp[k];
    if (sum0 > 0.) {
      for (k = 2; k <= km; k++) sum1 = sum1 + dmax(0., q[k] * dp[k]);
      fac = sum0 / sum1;
      for (k = 2; k <= km; k++) q[k] = dmax(0., fac * (q[k] * dp[k]) / dp[k]);
    }
  }
}

double fvo_moist_cv(const fvo_remap_par *p, const double *qk, size_t ns, double *q_con) {
#define Q(n) ((n) > 0 ? qk[(size_t)((n)-1) * ns] : 0.)
  double qv, ql, qs;
  switch (p->nwat) {
    case 2: /* :279-285 (no t1) */
      qv = fmax(0., Q(p->sphum));
      qs = fmax(0., Q(p->liq_wat));
      *q_con = qs;
      return (1. - qv) * p->cv_air + qv * p->cv_vap;
    case 3:
      qv = Q(p->sphum); ql = Q(p->liq_wat); qs = Q(p->ice_wat);
      *q_con = ql + qs;
      return (1. - (qv + *q_con)) * p->cv_air + qv * p->cv_vap + ql * p->c_liq + qs * p->c_ice;
    case 4:
      qv = Q(p->sphum);
      *q_con = Q(p->liq_wat) + Q(p->rainwat);
      return (1. - (qv + *q_con)) * p->cv_air + qv * p->cv_vap + *q_con * p->c_liq;
    case 5:
      qv = Q(p->sphum); ql = Q(p->liq_wat) + Q(p->rainwat); qs = Q(p->ice_wat) + Q(p->snowwat);
      *q_con = ql + qs;
      return (1. - (qv + *q_con)) * p->cv_air + qv * p->cv_vap + ql * p->c_liq + qs * p->c_ice;
    case 6:
      qv = Q(p->sphum); ql = Q(p->liq_wat) + Q(p->rainwat); qs = Q(p->ice_wat) + Q(p->snowwat) + Q(p->graupel);
      *q_con = ql + qs;
      return (1. - (qv + *q_con)) * p->cv_air + qv * p->cv_vap + ql * p->c_liq + qs * p->c_ice;
    default:
      *q_con = 0.;
      return p->cv_air;
  }
#undef Q
}

/* map1_cubic with T_VAR = 1 (total energy in log p) and conserv = .true. (fv_operators.F90:1897-2096, call site
 * fv_mapz.F90:353-355), one column; 1-based pe1, pe2 [1..km+1], q1, q2 [1..km] */
static void map1_cubic_te(int km, const double *pe1, const double *pe2, const double *q1, double *q2) {
  double *l1 = dalloc(km + 2), *l2 = dalloc(km + 2), *dl = dalloc(km + 2);
  double vsum1 = 0., vsum2 = 0.;
  int k;
  for (k = 1; k <= km; k++) {
    l1[k] = fv3_log(0.5 * (pe1[k] + pe1[k + 1]));
    l2[k] = fv3_log(0.5 * (pe2[k] + pe2[k + 1]));
  }
  for (k = 1; k <= km - 1; k++) dl[k] = l1[k + 1] - l1[k];
  for (k = 1; k <= km; k++) vsum1 = vsum1 + q1[k] * (pe1[k + 1] - pe1[k]);
  vsum1 = vsum1 / (pe1[km + 1] - pe1[1]);
  for (k = 1; k <= km; k++) {
    int lp0 = 1, lm1;
    while (lp0 <= km) {
      if (l1[lp0] < l2[k]) lp0 = lp0 + 1; else break;
    }
    lm1 = lp0 - 1 > 1 ? lp0 - 1 : 1;
    lp0 = lp0 < km ? lp0 : km;
    if (lm1 == 1 && lp0 == 1)
      q2[k] = q1[1] + (q1[2] - q1[1]) * (l2[k] - l1[1]) / (l1[2] - l1[1]);
    else if (lm1 == km && lp0 == km)
      q2[k] = q1[km] + (q1[km] - q1[km - 1]) * (l2[k] - l1[km]) / (l1[km] - l1[km - 1]);
    else if (lm1 == 1 || lp0 == km)
      q2[k] = q1[lp0] + (q1[lm1] - q1[lp0]) * (l2[k] - l1[lp0]) / (l1[lm1] - l1[lp0]);
    else {
      const int lp1 = lp0 + 1, lm2 = lm1 - 1;
      const double P = l2[k], plp1 = l1[lp1], plp0 = l1[lp0], plm1 = l1[lm1], plm2 = l1[lm2];
      const double dlp0 = dl[lp0], dlm1 = dl[lm1], dlm2 = dl[lm2];
      const double ap1 = (P - plp0) * (P - plm1) * (P - plm2) / (dlp0 * (dlp0 + dlm1) * (dlp0 + dlm1 + dlm2));
      const double ap0 = (plp1 - P) * (P - plm1) * (P - plm2) / (dlp0 * dlm1 * (dlm1 + dlm2));
      const double am1 = (plp1 - P) * (plp0 - P) * (P - plm2) / (dlm1 * dlm2 * (dlp0 + dlm1));
      const double am2 = (plp1 - P) * (plp0 - P) * (plm1 - P) / (dlm2 * (dlm1 + dlm2) * (dlp0 + dlm1 + dlm2));
      q2[k] = ap1 * q1[lp1] + ap0 * q1[lp0] + am1 * q1[lm1] + am2 * q1[lm2];
    }
  }
  for (k = 1; k <= km; k++) vsum2 = vsum2 + q2[k] * (pe2[k + 1] - pe2[k]);
  vsum2 = vsum2 / (pe2[km + 1] - pe2[1]);
  for (k = 1; k <= km; k++) q2[k] = q2[k] + vsum1 - vsum2;
  free(l1); free(l2); free(dl);
}

int fvo_lagrangian_to_eulerian(const fvo_grid *g, int km, const fvo_remap_par *p, double *ps, double *pe, double *delp,
                               double *pkz, double *pk, double *u, double *v, double *w, double *delz, double *pt,
                               double *q, double *peln, double *omga, const double *ws, const double *ak,
                               const double *bk, double *q_con, double *cappa) {
  const int is = g->is, ie = g->ie, js = g->js, je = g->je, isd = g->isd, ied = g->ied, jsd = g->jsd, jed = g->jed;
  const int nid = ied - isd + 1, njd = jed - jsd + 1, nx = ie - is + 1, ny = je - js + 1;
  const size_t nA = (size_t)nid * njd, nU = (size_t)nid * (njd + 1), nV = (size_t)(nid + 1) * njd, nCC = (size_t)nx * ny;
  int i, j, k, n, iq, rc = 0;
  const double k1k = p->rdgas / p->cv_air, rrg = -p->rdgas / p->grav, akap = p->akap;
  if (!p->hydrostatic && p->kord_wz < 0) return FVO_ERR_UNSUPPORTED; /* iv = -3: reads gam(i,km) unset (file header) */
  if ((p->moist_kappa || p->use_cond) && (p->hydrostatic || !q || (p->moist_kappa && (!q_con || !cappa))))
    return FVO_ERR_UNSUPPORTED;
  /* remap_te (:232-286, :348-360, :576-619): total energy is remapped in the place of T_v / theta_v and T_v, pkz follow from it
   * after the winds of the row were remapped.  te: A x km (the reference's te argument: fv_dynamics hands it dp1), hs: A.
   * The loop over j runs in order, as the reference's does without OpenMP: row j reads the winds of row j + 1 BEFORE they are
   * remapped, both in the energy (:248-252, :275-279) and in the kinetic energy taken back out of it (:587-589, :603-617) */
  const int remap_te = p->remap_te;
  const double *hs = p->hs;
  double *te = p->te;
  const double rv = p->adiabatic ? 0. : p->r_vir; /* the reference's caller passes zvir = 0 for an adiabatic run */
  if (remap_te && (!hs || !te)) return FVO_ERR_UNSUPPORTED;
#define IA3(i, j, k) ((size_t)((k)-1) * nA + (size_t)((j)-jsd) * nid + ((i)-isd))
#define IU3(i, j, k) ((size_t)((k)-1) * nU + (size_t)((j)-jsd) * nid + ((i)-isd))
#define IV3(i, j, k) ((size_t)((k)-1) * nV + (size_t)((j)-jsd) * (nid + 1) + ((i)-isd))
#define ICC3(i, j, k) ((size_t)((k)-1) * nCC + (size_t)((j)-js) * nx + ((i)-is))
#define PE(i, k, j) pe[(size_t)((j) - (js - 1)) * (nx + 2) * (km + 1) + (size_t)((k)-1) * (nx + 2) + ((i) - (is - 1))]
#define PELN(i, k, j) peln[(size_t)((j)-js) * nx * (km + 1) + (size_t)((k)-1) * nx + ((i)-is)]
#define KE_TE(i, j, k)                                                                                                     \
  (0.25 * g->rsin2[(size_t)((j)-jsd) * nid + ((i)-isd)] *                                                                  \
   (u[IU3(i, j, k)] * u[IU3(i, j, k)] + u[IU3(i, j + 1, k)] * u[IU3(i, j + 1, k)] + v[IV3(i, j, k)] * v[IV3(i, j, k)] +     \
    v[IV3(i + 1, j, k)] * v[IV3(i + 1, j, k)] -                                                                            \
    (u[IU3(i, j, k)] + u[IU3(i, j + 1, k)]) * (v[IV3(i, j, k)] + v[IV3(i + 1, j, k)]) * g->cosa_s[(size_t)((j)-jsd) * nid + ((i)-isd)]))
  double *phis = dalloc(km + 3);
  double *pe4 = dalloc(nA * km);
  double *c1 = dalloc(km + 3), *c2 = dalloc(km + 3), *pe1 = dalloc(km + 3), *pe2 = dalloc(km + 3), *pn1 = dalloc(km + 3),
         *pn2 = dalloc(km + 3), *pk2 = dalloc(km + 3), *dp2 = dalloc(km + 3), *pe0 = dalloc(km + 3), *pe3 = dalloc(km + 3);
  for (j = js; j <= je + 1; j++) {
    for (i = is; i <= ie + 1; i++) {
      if (i <= ie) {
        for (k = 1; k <= km + 1; k++) pe1[k] = PE(i, k, j);
        pe2[1] = p->ptop;
        pe2[km + 1] = PE(i, km + 1, j);
      }
      if (j != je + 1 && i <= ie) {
        if (remap_te) { /* :232-286: cp T + KE + phis */
          phis[km + 1] = hs[(size_t)(j - jsd) * nid + (i - isd)];
          if (p->hydrostatic) {
            PELN(i, 1, j) = fv3_log(p->ptop); /* pkez, :886-895 (ptop >= ptop_min) */
            for (k = 1; k <= km; k++)         /* pkez, :898-903 */
              pkz[ICC3(i, j, k)] = (pk[ICC3(i, j, k + 1)] - pk[ICC3(i, j, k)]) / (akap * (PELN(i, k + 1, j) - PELN(i, k, j)));
            for (k = km; k >= 1; k--)
              phis[k] = phis[k + 1] + p->cp * pt[IA3(i, j, k)] * (pk[ICC3(i, j, k + 1)] - pk[ICC3(i, j, k)]);
            for (k = 1; k <= km + 1; k++) phis[k] = phis[k] * pe1[k];
            for (k = 1; k <= km; k++)
              te[IA3(i, j, k)] = KE_TE(i, j, k) + p->cp * pt[IA3(i, j, k)] * pkz[ICC3(i, j, k)] +
                                 (phis[k + 1] - phis[k]) / (pe1[k + 1] - pe1[k]);
          } else {
            for (k = km; k >= 1; k--) {
              const double qv = p->sphum > 0 ? q[(size_t)(p->sphum - 1) * nA * km + IA3(i, j, k)] : 0.;
              phis[k] = phis[k + 1] - p->grav * delz[ICC3(i, j, k)];
              if (p->moist_kappa) {
                double qc;
                const double cvm = fvo_moist_cv(p, q + IA3(i, j, k), nA * km, &qc);
                const double cap = p->rdgas / (p->rdgas + cvm / (1. + rv * qv));
                q_con[IA3(i, j, k)] = qc;
                cappa[IA3(i, j, k)] = cap;
                pkz[ICC3(i, j, k)] =
                    fv3_exp(cap / (1. - cap) * fv3_log(rrg * delp[IA3(i, j, k)] / delz[ICC3(i, j, k)] * pt[IA3(i, j, k)]));
                te[IA3(i, j, k)] = cvm * pt[IA3(i, j, k)] * pkz[ICC3(i, j, k)] / ((1. + rv * qv) * (1. - qc)) +
                                   0.5 * (w[IA3(i, j, k)] * w[IA3(i, j, k)]) + KE_TE(i, j, k) + 0.5 * (phis[k + 1] + phis[k]);
              } else {
                pkz[ICC3(i, j, k)] = fv3_exp(k1k * fv3_log(rrg * delp[IA3(i, j, k)] / delz[ICC3(i, j, k)] * pt[IA3(i, j, k)]));
                te[IA3(i, j, k)] = p->cv_air * pt[IA3(i, j, k)] * pkz[ICC3(i, j, k)] / (1. + rv * qv) +
                                   0.5 * (w[IA3(i, j, k)] * w[IA3(i, j, k)]) + KE_TE(i, j, k) + 0.5 * (phis[k + 1] + phis[k]);
              }
            }
          }
        } else
        if (p->kord_tm < 0) { /* :200-229 */
          for (k = 1; k <= km; k++) {
            if (p->hydrostatic)
              pt[IA3(i, j, k)] = pt[IA3(i, j, k)] * (pk[ICC3(i, j, k + 1)] - pk[ICC3(i, j, k)]) /
                                 (akap * (PELN(i, k + 1, j) - PELN(i, k, j)));
            else if (p->moist_kappa) { /* :212-219 */
              double qc;
              const double cvm = fvo_moist_cv(p, q + IA3(i, j, k), nA * km, &qc);
              const double qv = q[(size_t)(p->sphum - 1) * nA * km + IA3(i, j, k)];
              q_con[IA3(i, j, k)] = qc;
              cappa[IA3(i, j, k)] = p->rdgas / (p->rdgas + cvm / (1. + p->r_vir * qv));
              pt[IA3(i, j, k)] = pt[IA3(i, j, k)] * fv3_exp(cappa[IA3(i, j, k)] / (1. - cappa[IA3(i, j, k)]) *
                                                      fv3_log(rrg * delp[IA3(i, j, k)] / delz[ICC3(i, j, k)] * pt[IA3(i, j, k)]));
            } else
              pt[IA3(i, j, k)] = pt[IA3(i, j, k)] *
                                 fv3_exp(k1k * fv3_log(rrg * delp[IA3(i, j, k)] / delz[ICC3(i, j, k)] * pt[IA3(i, j, k)]));
          }
        }
        if (!p->hydrostatic)
          for (k = 1; k <= km; k++) delz[ICC3(i, j, k)] = -delz[ICC3(i, j, k)] / delp[IA3(i, j, k)]; /* :292 */
        ps[(size_t)(j - jsd) * nid + (i - isd)] = pe1[km + 1];
        for (k = 2; k <= km; k++) pe2[k] = ak[k - 1] + bk[k - 1] * PE(i, km + 1, j);
        for (k = 1; k <= km; k++) dp2[k] = pe2[k + 1] - pe2[k];
        for (k = 1; k <= km; k++) delp[IA3(i, j, k)] = dp2[k];
        pn2[1] = PELN(i, 1, j);
        pn2[km + 1] = PELN(i, km + 1, j);
        pk2[1] = pk[ICC3(i, j, 1)];
        pk2[km + 1] = pk[ICC3(i, j, km + 1)];
        for (k = 2; k <= km; k++) {
          pn2[k] = fv3_log(pe2[k]);
          pk2[k] = fv3_exp(akap * pn2[k]);
        }
        /* 1) remap Tv / thetav, :362-376 -- or the total energy, :348-360 */
        if (remap_te) {
          for (k = 1; k <= km; k++) c1[k] = te[IA3(i, j, k)];
          if (p->kord_tm == 0) {
            map1_cubic_te(km, pe1, pe2, c1, c2);
          } else {
            for (k = 1; k <= km + 1; k++) pn1[k] = PELN(i, k, j);
            rc |= fvo_remap_column(0, km, pn1, pn2, c1, c2, 0., 1, abs(p->kord_tm), p->cp * p->t_min);
          }
          for (k = 1; k <= km; k++) te[IA3(i, j, k)] = c2[k];
        }
        for (k = 1; k <= km; k++) c1[k] = pt[IA3(i, j, k)];
        if (remap_te) {
        } else if (p->kord_tm < 0) {
          for (k = 1; k <= km + 1; k++) pn1[k] = PELN(i, k, j);
          rc |= fvo_remap_column(0, km, pn1, pn2, c1, c2, 0., 1, abs(p->kord_tm), p->t_min);
        } else {
          rc |= fvo_remap_column(1, km, pe1, pe2, c1, c2, 0., 1, abs(p->kord_tm), 0.);
        }
        if (!remap_te)
          for (k = 1; k <= km; k++) pt[IA3(i, j, k)] = c2[k];
        /* 2) constituents, :380-397 */
        for (iq = 1; iq <= p->nq; iq++) {
          double *qq = q + (size_t)(iq - 1) * nA * km;
          for (k = 1; k <= km; k++) c1[k] = qq[IA3(i, j, k)];
          rc |= fvo_remap_column(p->nq > 5 ? 3 : 2, km, pe1, pe2, c1, c2, 0., 0, p->kord_tr[iq - 1], 0.);
          if (p->fill) fvo_fillz_column(km, c2, dp2); /* fv_operators.F90:337 / fv_mapz.F90:390 */
          for (k = 1; k <= km; k++) qq[IA3(i, j, k)] = c2[k];
        }
        /* 3) w and delz, :400-423 */
        if (!p->hydrostatic) {
          for (k = 1; k <= km; k++) c1[k] = w[IA3(i, j, k)];
          rc |= fvo_remap_column(1, km, pe1, pe2, c1, c2, ws[(size_t)(j - js) * nx + (i - is)], -2, abs(p->kord_wz), 0.);
          for (k = 1; k <= km; k++) w[IA3(i, j, k)] = c2[k];
          for (k = 1; k <= km; k++) c1[k] = delz[ICC3(i, j, k)];
          rc |= fvo_remap_column(1, km, pe1, pe2, c1, c2, 0., 1, abs(p->kord_tm), 0.);
          for (k = 1; k <= km; k++) delz[ICC3(i, j, k)] = -c2[k] * dp2[k];
        }
        for (k = 1; k <= km + 1; k++) pk[ICC3(i, j, k)] = pk2[k]; /* :426-430 */
        if (p->last_step) {                                        /* :432-443 */
          pe3[1] = 0.;
          for (k = 2; k <= km + 1; k++) pe3[k] = omga[IA3(i, j, k - 1)];
        }
        for (k = 1; k <= km + 1; k++) { /* :445-450 */
          pe0[k] = PELN(i, k, j);
          PELN(i, k, j) = pn2[k];
        }
        /* 3.2) pkz, :453-503 */
        for (k = 1; k <= km && !remap_te; k++) {
          if (p->hydrostatic)
            pkz[ICC3(i, j, k)] = (pk2[k + 1] - pk2[k]) / (akap * (PELN(i, k + 1, j) - PELN(i, k, j)));
          else if (p->moist_kappa) { /* :463-478; q holds the remapped tracers here */
            double qc;
            const double cvm = fvo_moist_cv(p, q + IA3(i, j, k), nA * km, &qc);
            const double qv = q[(size_t)(p->sphum - 1) * nA * km + IA3(i, j, k)];
            const double cap = p->rdgas / (p->rdgas + cvm / (1. + p->r_vir * qv));
            q_con[IA3(i, j, k)] = qc;
            cappa[IA3(i, j, k)] = cap;
            pkz[ICC3(i, j, k)] = fv3_exp((p->kord_tm < 0 ? cap : cap / (1. - cap)) *
                                     fv3_log(rrg * delp[IA3(i, j, k)] / delz[ICC3(i, j, k)] * pt[IA3(i, j, k)]));
          } else if (p->kord_tm < 0)
            pkz[ICC3(i, j, k)] = fv3_exp(akap * fv3_log(rrg * delp[IA3(i, j, k)] / delz[ICC3(i, j, k)] * pt[IA3(i, j, k)]));
          else
            pkz[ICC3(i, j, k)] = fv3_exp(k1k * fv3_log(rrg * delp[IA3(i, j, k)] / delz[ICC3(i, j, k)] * pt[IA3(i, j, k)]));
        }
        if (p->kord_tm > 0 && !remap_te)
          for (k = 1; k <= km; k++) pt[IA3(i, j, k)] = pt[IA3(i, j, k)] * pkz[ICC3(i, j, k)];
        /* 3.3) omega, :506-526 */
        if (p->last_step) {
          int k_next = 1, kp;
          for (k = 1; k <= km; k++) dp2[k] = 0.5 * (PELN(i, k, j) + PELN(i, k + 1, j));
          for (n = 1; n <= km; n++) {
            kp = k_next;
            for (k = kp; k <= km; k++) {
              if (dp2[n] <= pe0[k + 1] && dp2[n] >= pe0[k]) {
                omga[IA3(i, j, n)] = pe3[k] + (pe3[k + 1] - pe3[k]) * (dp2[n] - pe0[k]) / (pe0[k + 1] - pe0[k]);
                k_next = k;
                break;
              }
            }
          }
        }
      }
      /* 4.1) u, :530-553 */
      if (i <= ie) {
        pe0[1] = PE(i, 1, j);
        for (k = 2; k <= km + 1; k++) pe0[k] = 0.5 * (PE(i, k, j - 1) + pe1[k]);
        for (k = 1; k <= km + 1; k++) {
          const double bkh = 0.5 * bk[k - 1];
          pe3[k] = ak[k - 1] + bkh * (PE(i, km + 1, j - 1) + pe1[km + 1]);
        }
        for (k = 1; k <= km; k++) c1[k] = u[IU3(i, j, k)];
        rc |= fvo_remap_column(1, km, pe0, pe3, c1, c2, 0., -1, p->kord_mt, 0.);
        for (k = 1; k <= km; k++) u[IU3(i, j, k)] = c2[k];
      }
      /* 4.2) v, :555-573 */
      if (j < je + 1) {
        pe0[1] = PE(i, 1, j);
        pe3[1] = ak[0];
        for (k = 2; k <= km + 1; k++) {
          const double bkh = 0.5 * bk[k - 1];
          pe0[k] = 0.5 * (PE(i - 1, k, j) + PE(i, k, j));
          pe3[k] = ak[k - 1] + bkh * (PE(i - 1, km + 1, j) + PE(i, km + 1, j));
        }
        for (k = 1; k <= km; k++) c1[k] = v[IV3(i, j, k)];
        rc |= fvo_remap_column(1, km, pe0, pe3, c1, c2, 0., -1, p->kord_mt, 0.);
        for (k = 1; k <= km; k++) v[IV3(i, j, k)] = c2[k];
      }
      if (i <= ie && j <= je)
        for (k = 1; k <= km; k++) pe4[IA3(i, j, k)] = pe2[k + 1]; /* :624-628 */
    }
    /* 4a) T_v and pkz from the remapped total energy, :576-619: u(:, j), v(:, j) are remapped, u(:, j + 1) is not yet */
    if (remap_te && j <= je)
      for (i = is; i <= ie; i++) {
        phis[km + 1] = hs[(size_t)(j - jsd) * nid + (i - isd)];
        pe2[1] = p->ptop;
        pe2[km + 1] = PE(i, km + 1, j);
        for (k = 2; k <= km; k++) pe2[k] = ak[k - 1] + bk[k - 1] * PE(i, km + 1, j);
        for (k = km; k >= 1; k--) {
          double tpe;
          if (p->hydrostatic) {
            const double dlnp = p->rdgas * (PELN(i, k + 1, j) - PELN(i, k, j));
            tpe = te[IA3(i, j, k)] - phis[k + 1] - KE_TE(i, j, k);
            pt[IA3(i, j, k)] = tpe / (p->cp - pe2[k] * dlnp / delp[IA3(i, j, k)]);
            pkz[ICC3(i, j, k)] = (pk[ICC3(i, j, k + 1)] - pk[ICC3(i, j, k)]) / (akap * (PELN(i, k + 1, j) - PELN(i, k, j)));
            phis[k] = phis[k + 1] + dlnp * pt[IA3(i, j, k)];
          } else {
            const double qv = p->sphum > 0 ? q[(size_t)(p->sphum - 1) * nA * km + IA3(i, j, k)] : 0.;
            phis[k] = phis[k + 1] - delz[ICC3(i, j, k)] * p->grav;
            tpe = te[IA3(i, j, k)] - 0.5 * (phis[k] + phis[k + 1]) - 0.5 * (w[IA3(i, j, k)] * w[IA3(i, j, k)]) - KE_TE(i, j, k);
            if (p->moist_kappa) {
              double qc;
              const double cvm = fvo_moist_cv(p, q + IA3(i, j, k), nA * km, &qc);
              q_con[IA3(i, j, k)] = qc;
              cappa[IA3(i, j, k)] = p->rdgas / (p->rdgas + cvm / (1. + rv * qv));
              pt[IA3(i, j, k)] = tpe / cvm * (1. + rv * qv) * (1. - qc);
              pkz[ICC3(i, j, k)] =
                  fv3_exp(cappa[IA3(i, j, k)] * fv3_log(rrg * delp[IA3(i, j, k)] / delz[ICC3(i, j, k)] * pt[IA3(i, j, k)]));
            } else {
              pt[IA3(i, j, k)] = tpe / p->cv_air * (1. + rv * qv);
              pkz[ICC3(i, j, k)] = fv3_exp(akap * fv3_log(rrg * delp[IA3(i, j, k)] / delz[ICC3(i, j, k)] * pt[IA3(i, j, k)]));
            }
          }
        }
      }
  }
  for (k = 2; k <= km; k++) /* :635-641 */
    for (j = js; j <= je; j++)
      for (i = is; i <= ie; i++) PE(i, k, j) = pe4[IA3(i, j, k - 1)];
  if (p->last_step == 2) { /* the energy fixer follows (fvo_energy_fixer_sums, fvo_remap_finish) */
  } else if (p->last_step) { /* :793-821 with dtmp = 0 */
    if (!p->hydrostatic && p->use_cond) { /* :806-811 */
      for (k = 1; k <= km; k++)
        for (j = js; j <= je; j++)
          for (i = is; i <= ie; i++) {
            double qc;
            const double cvm = fvo_moist_cv(p, q + IA3(i, j, k), nA * km, &qc);
            const double qv = q[(size_t)(p->sphum - 1) * nA * km + IA3(i, j, k)];
            pt[IA3(i, j, k)] = (pt[IA3(i, j, k)] + 0. / cvm * pkz[ICC3(i, j, k)]) / ((1. + p->r_vir * qv) * (1. - qc));
          }
    } else if (!p->adiabatic) {
      for (k = 1; k <= km; k++)
        for (j = js; j <= je; j++)
          for (i = is; i <= ie; i++) {
            const double qv = p->sphum > 0 ? q[(size_t)(p->sphum - 1) * nA * km + IA3(i, j, k)] : 0.;
            const double den = p->hydrostatic ? p->cp : p->cv_air;
            pt[IA3(i, j, k)] = (pt[IA3(i, j, k)] + 0. / den * pkz[ICC3(i, j, k)]) / (1. + p->r_vir * qv);
          }
    }
  } else { /* :833-841 */
    for (k = 1; k <= km; k++)
      for (j = js; j <= je; j++)
        for (i = is; i <= ie; i++) pt[IA3(i, j, k)] = pt[IA3(i, j, k)] / pkz[ICC3(i, j, k)];
  }
  free(phis);
  free(pe4); free(c1); free(c2); free(pe1); free(pe2); free(pn1); free(pn2); free(pk2); free(dp2); free(pe0); free(pe3);
  return rc;
}

/* ---------------------------------------------------------------------------------------------------
 * Total energy and the energy fixer (consv_te).  Layouts as in fvo_lagrangian_to_eulerian; te_2d, zsum*: CC.
 * ------------------------------------------------------------------------------------------------- */
#define EBOUNDS                                                                                           \
  const int is = g->is, ie = g->ie, js = g->js, je = g->je, ng = g->ng;                                     \
  const int isd = is - ng, ied = ie + ng, jsd = js - ng, jed = je + ng;                                     \
  const int nid = ied - isd + 1, njd = jed - jsd + 1, nx = ie - is + 1, ny = je - js + 1;                   \
  const size_t nA = (size_t)nid * njd, nU = (size_t)nid * (njd + 1), nV = (size_t)(nid + 1) * njd, nCC = (size_t)nx * ny; \
  int i, j, k;                                                                                            \
  (void)jed; (void)nU; (void)nV; (void)nCC
#define EA2(i, j) ((size_t)((j)-jsd) * nid + ((i)-isd))
#define ECC2(i, j) ((size_t)((j)-js) * nx + ((i)-is))
#define EPE(i, k, j) pe[(size_t)((j) - (js - 1)) * (nx + 2) * (km + 1) + (size_t)((k)-1) * (nx + 2) + ((i) - (is - 1))]
#define EPELN(i, k, j) peln[(size_t)((j)-js) * nx * (km + 1) + (size_t)((k)-1) * nx + ((i)-is)]
#define EWIND(i, j, k)                                                                                                    \
  (u[IU3(i, j, k)] * u[IU3(i, j, k)] + u[IU3(i, j + 1, k)] * u[IU3(i, j + 1, k)] + v[IV3(i, j, k)] * v[IV3(i, j, k)] +     \
   v[IV3(i + 1, j, k)] * v[IV3(i + 1, j, k)] -                                                                            \
   (u[IU3(i, j, k)] + u[IU3(i, j + 1, k)]) * (v[IV3(i, j, k)] + v[IV3(i + 1, j, k)]) * g->cosa_s[EA2(i, j)])

/* compute_total_energy, fv_thermodynamics.F90:90-225 (USE_COND not defined; teq not restated).  qc: A x km or NULL. */
int fvo_compute_total_energy(const fvo_grid *g, int km, const fvo_remap_par *p, int moist_phys, const double *u,
                             const double *v, const double *w, const double *delz, const double *pt, const double *delp,
                             const double *q, const double *qc, const double *pe, const double *peln, const double *hs,
                             double *te_2d) {
  EBOUNDS;
/* qc = zvir*q(sphum), fv_dynamics.F90:295-301: the caller's array, or formed from the tracer */
#define QCV(i, j, k) (qc ? qc[IA3(i, j, k)] : ((p->sphum > 0 && !p->adiabatic && q) ? p->r_vir * q[(size_t)(p->sphum - 1) * nA * km + IA3(i, j, k)] : 0.))
  double *phiz = dalloc(km + 2);
  for (j = js; j <= je; j++)
    for (i = is; i <= ie; i++) {
      double te;
      if (p->hydrostatic) {
        phiz[km + 1] = hs[EA2(i, j)];
        for (k = km; k >= 1; k--) {
          const double tv = pt[IA3(i, j, k)] * (1. + QCV(i, j, k));
          phiz[k] = phiz[k + 1] + p->rdgas * tv * (EPELN(i, k + 1, j) - EPELN(i, k, j));
        }
        te = EPE(i, km + 1, j) * phiz[km + 1] - EPE(i, 1, j) * phiz[1];
        for (k = 1; k <= km; k++) {
          const double tv = pt[IA3(i, j, k)] * (1. + QCV(i, j, k));
          te = te + delp[IA3(i, j, k)] * (p->cp * tv + 0.25 * g->rsin2[EA2(i, j)] * EWIND(i, j, k));
        }
      } else {
        phiz[km + 1] = hs[EA2(i, j)];
        for (k = km; k >= 1; k--) phiz[k] = phiz[k + 1] - p->grav * delz[ICC3(i, j, k)];
        te = 0.;
        for (k = 1; k <= km; k++) {
          double cv = p->cv_air, qd;
          if (moist_phys && p->moist_kappa) cv = fvo_moist_cv(p, q + IA3(i, j, k), nA * km, &qd); /* :186-193 */
          te = te + delp[IA3(i, j, k)] *
                        (cv * pt[IA3(i, j, k)] + 0.5 * (phiz[k] + phiz[k + 1] + w[IA3(i, j, k)] * w[IA3(i, j, k)] +
                                                        0.5 * g->rsin2[EA2(i, j)] * EWIND(i, j, k)));
        }
      }
      te_2d[ECC2(i, j)] = te;
    }
  free(phiz);
  return FVO_OK;
}

/* fv_mapz.F90:647-734 (consv > consv_min, remap_te = .false.) and :745-763 (only_sums) */
int fvo_energy_fixer_sums(const fvo_grid *g, int km, const fvo_remap_par *p, int only_sums, const double *u, const double *v,
                          const double *w, const double *delz, const double *pt, const double *delp, const double *q,
                          const double *pe, const double *peln, const double *hs, const double *pkz, const double *pk,
                          const double *te0_2d, double *te_2d, double *zsum1, double *zsum0, double *q_con) {
  EBOUNDS;
  double *phis = dalloc(km + 2);
  const double rv = p->adiabatic ? 0. : p->r_vir; /* the reference's caller passes zvir = 0 for an adiabatic run */
  for (j = js; j <= je; j++)
    for (i = is; i <= ie; i++) {
      if (!only_sums) {
        double te;
        if (p->remap_te) { /* :655-663 */
          te = p->te[IA3(i, j, 1)] * delp[IA3(i, j, 1)];
          for (k = 2; k <= km; k++) te = te + p->te[IA3(i, j, k)] * delp[IA3(i, j, k)];
        } else if (p->hydrostatic) {
          double gz = hs[EA2(i, j)];
          for (k = 1; k <= km; k++) gz = gz + p->rdgas * pt[IA3(i, j, k)] * (EPELN(i, k + 1, j) - EPELN(i, k, j));
          te = EPE(i, km + 1, j) * hs[EA2(i, j)] - EPE(i, 1, j) * gz;
          for (k = 1; k <= km; k++)
            te = te + delp[IA3(i, j, k)] * (p->cp * pt[IA3(i, j, k)] + 0.25 * g->rsin2[EA2(i, j)] * EWIND(i, j, k));
        } else {
          te = 0.;
          phis[km + 1] = hs[EA2(i, j)];
          for (k = km; k >= 1; k--) phis[k] = phis[k + 1] - p->grav * delz[ICC3(i, j, k)];
          for (k = 1; k <= km; k++) {
            const double qv = p->sphum > 0 ? q[(size_t)(p->sphum - 1) * nA * km + IA3(i, j, k)] : 0.;
            const double mech = 0.5 * (phis[k] + phis[k + 1] + w[IA3(i, j, k)] * w[IA3(i, j, k)] +
                                       0.5 * g->rsin2[EA2(i, j)] * EWIND(i, j, k));
            if (p->use_cond) {
              double qc;
              const double cvm = fvo_moist_cv(p, q + IA3(i, j, k), nA * km, &qc);
              q_con[IA3(i, j, k)] = qc;
              te = te + delp[IA3(i, j, k)] * (cvm * pt[IA3(i, j, k)] / ((1. + rv * qv) * (1. - qc)) + mech);
            } else {
              te = te + delp[IA3(i, j, k)] * (p->cv_air * pt[IA3(i, j, k)] / (1. + rv * qv) + mech);
            }
          }
        }
        te_2d[ECC2(i, j)] = te0_2d[ECC2(i, j)] - te;
      }
      zsum1[ECC2(i, j)] = pkz[ICC3(i, j, 1)] * delp[IA3(i, j, 1)];
      for (k = 2; k <= km; k++) zsum1[ECC2(i, j)] = zsum1[ECC2(i, j)] + pkz[ICC3(i, j, k)] * delp[IA3(i, j, k)];
      if (p->hydrostatic) zsum0[ECC2(i, j)] = p->ptop * (pk[ICC3(i, j, 1)] - pk[ICC3(i, j, km + 1)]) + zsum1[ECC2(i, j)];
    }
  free(phis);
  return FVO_OK;
}

/* fv_mapz.F90:793-821 with the fixer's dtmp */
int fvo_remap_finish(const fvo_grid *g, int km, const fvo_remap_par *p, double dtmp, double *pt, const double *pkz,
                     const double *q) {
  EBOUNDS;
  const double rv = p->adiabatic ? 0. : p->r_vir;
  for (k = 1; k <= km; k++)
    for (j = js; j <= je; j++)
      for (i = is; i <= ie; i++) {
        const double qv = p->sphum > 0 ? q[(size_t)(p->sphum - 1) * nA * km + IA3(i, j, k)] : 0.;
        if (p->hydrostatic) {
          pt[IA3(i, j, k)] = (pt[IA3(i, j, k)] + dtmp / p->cp * pkz[ICC3(i, j, k)]) / (1. + rv * qv);
        } else if (p->use_cond) {
          double qc;
          const double cvm = fvo_moist_cv(p, q + IA3(i, j, k), nA * km, &qc);
          pt[IA3(i, j, k)] = (pt[IA3(i, j, k)] + dtmp / cvm * pkz[ICC3(i, j, k)]) / ((1. + rv * qv) * (1. - qc));
        } else if (!p->adiabatic) {
          pt[IA3(i, j, k)] = (pt[IA3(i, j, k)] + dtmp / p->cv_air * pkz[ICC3(i, j, k)]) / (1. + rv * qv);
        }
      }
  return FVO_OK;
}
