/*
 * oracle/tp_core.c -- CPU oracle (test infrastructure, see fvo.h) for model/tp_core.F90:
 * xppm/yppm (1-D PPM face values), pert_ppm, deln_flux, fv_tp_2d (Lin-Rood 2-D transport).
 * Doubly-periodic / Cartesian branches only (grid_type >= 3, no cubed-sphere edges).
 */
#include "fvo.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* tp_core.F90:35-70 */
static const double ppm_fac = 1.5;
static const double r3 = 1. / 3.;
static const double near_zero = 1.E-25;
static const double r12 = 1. / 12.;
static const double p1 = 7. / 12.;
static const double p2 = -1. / 12.;
/* cubed-sphere edge formulas (tp_core.F90:54-62) */
static const double s11 = 11. / 14., s14 = 4. / 7., s15 = 3. / 14.;
static const double c1 = -2. / 14., c2 = 11. / 14., c3 = 5. / 14.;

static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin3(double a, double b, double c) { return dmin(dmin(a, b), c); }
static inline double dmax3(double a, double b, double c) { return dmax(dmax(a, b), c); }
/* Fortran sign(a,b) = |a| with the sign of b (b = +0 counts as positive) */
static inline double fsign(double a, double b) { return copysign(fabs(a), b); }
/* Fortran x**n with integer n: repeated multiplication */
static inline double ipow(double x, int n) {
  double r = x;
  int k;
  for (k = 1; k < n; k++) r = r * x;
  return r;
}

/* pert_ppm, tp_core.F90:1206-1264.  Arrays are 0-based here (Fortran 1-based). */
void fvo_pert_ppm(int im, const double *a0, double *al, double *ar, int iv) {
  double a4, da1, da2, a6da, fmin;
  int i;
  if (iv == 0) { /* positive definite constraint, :1219-1242 */
    for (i = 0; i < im; i++) {
      if (a0[i] <= 0.) {
        al[i] = 0.;
        ar[i] = 0.;
      } else {
        a4 = -3. * (ar[i] + al[i]);
        da1 = ar[i] - al[i];
        if (fabs(da1) < -a4) {
          fmin = a0[i] + 0.25 / a4 * (da1 * da1) + a4 * r12;
          if (fmin < 0.) {
            if (ar[i] > 0. && al[i] > 0.) {
              ar[i] = 0.;
              al[i] = 0.;
            } else if (da1 > 0.) {
              ar[i] = -2. * al[i];
            } else {
              al[i] = -2. * ar[i];
            }
          }
        }
      }
    }
  } else { /* standard PPM constraint, :1243-1262 */
    for (i = 0; i < im; i++) {
      if (al[i] * ar[i] < 0.) {
        da1 = al[i] - ar[i];
        da2 = da1 * da1;
        a6da = 3. * (al[i] + ar[i]) * da1;
        if (a6da < -da2) {
          ar[i] = -2. * al[i];
        } else if (a6da > da2) {
          al[i] = -2. * ar[i];
        }
      } else {
        al[i] = 0.;
        ar[i] = 0.;
      }
    }
  }
}

/*
 * One line of xppm (tp_core.F90:359-710) / yppm (:750-1150) in the branch
 * "bounded_domain .or. grid_type>=3": is1=is-1, ie3=ie+2, ie1=ie+1 (:352-355, :743-746).
 * q1[i] valid for i in [is-3, ie+3]; c[i], flux[i] for i in [is, ie+1].
 */
int fvo_ppm_line(const double *q1, const double *c, double *flux, int is, int ie, int iord,
                 double lim_fac) {
  return fvo_ppm_line_cs(q1, c, flux, is, ie, iord, lim_fac, NULL, 0);
}

/* The same with the cubed-sphere face edges (.not. bounded_domain .and. grid_type < 3): dxa[i] = the cell widths along
 * the line (dxa(:, j) for xppm, dya(i, :) for yppm), npx = the edge index of the direction (npx / npy); the west / south
 * edge is treated when is == 1, the east / north edge when ie + 1 == npx.  dxa == NULL: no edges (the branch above). */
int fvo_ppm_line_cs(const double *q1, const double *c, double *flux, int is, int ie, int iord,
                    double lim_fac, const double *dxa, int npx) {
  const int lo = is - 3, n = ie - is + 7;
  const int cubed = dxa != NULL;
  const int is1 = cubed ? (3 > is - 1 ? 3 : is - 1) : is - 1;               /* :349-356 */
  const int ie3 = cubed ? (npx - 2 < ie + 2 ? npx - 2 : ie + 2) : ie + 2;
  const int ie1 = cubed ? (npx - 3 < ie + 1 ? npx - 3 : ie + 1) : ie + 1;
  const int mord = abs(iord);
  int i;
  int rc = FVO_OK;
  double *buf = (double *)malloc(sizeof(double) * 8 * (size_t)n);
  unsigned char *lbuf = (unsigned char *)malloc(4 * (size_t)n);
  /* work arrays addressed with the Fortran index */
  double *bl = buf - lo, *br = buf + n - lo, *b0 = buf + 2 * n - lo, *al = buf + 3 * n - lo;
  double *dm = buf + 4 * n - lo, *dq = buf + 5 * n - lo, *a4 = buf + 6 * n - lo,
         *da1 = buf + 7 * n - lo;
  unsigned char *smt5 = lbuf - lo, *smt6 = lbuf + n - lo, *ext5 = lbuf + 2 * n - lo,
                *ext6 = lbuf + 3 * n - lo;
  double x0, xt, qtmp, pmp_1, lac_1, pmp_2, lac_2, fx1;

  if (iord < 7) {
    /* :369-371 */
    for (i = is1; i <= ie3; i++) al[i] = p1 * (q1[i - 1] + q1[i]) + p2 * (q1[i - 2] + q1[i + 1]);
    if (cubed) { /* :373-386 */
      if (is == 1) {
        al[0] = c1 * q1[-2] + c2 * q1[-1] + c3 * q1[0];
        al[1] = 0.5 * (((2. * dxa[0] + dxa[-1]) * q1[0] - dxa[0] * q1[-1]) / (dxa[-1] + dxa[0]) +
                       ((2. * dxa[1] + dxa[2]) * q1[1] - dxa[1] * q1[2]) / (dxa[1] + dxa[2]));
        al[2] = c3 * q1[1] + c2 * q1[2] + c1 * q1[3];
      }
      if ((ie + 1) == npx) {
        al[npx - 1] = c1 * q1[npx - 3] + c2 * q1[npx - 2] + c3 * q1[npx - 1];
        al[npx] = 0.5 * (((2. * dxa[npx - 1] + dxa[npx - 2]) * q1[npx - 1] - dxa[npx - 1] * q1[npx - 2]) / (dxa[npx - 2] + dxa[npx - 1]) +
                         ((2. * dxa[npx] + dxa[npx + 1]) * q1[npx] - dxa[npx] * q1[npx + 1]) / (dxa[npx] + dxa[npx + 1]));
        al[npx + 1] = c3 * q1[npx] + c2 * q1[npx + 1] + c1 * q1[npx + 2];
      }
    }
    if (iord < 0) { /* :388-392 */
      for (i = is - 1; i <= ie + 2; i++) al[i] = dmax(0., al[i]);
    }
    if (mord == 1) { /* :394-411 */
      for (i = is - 1; i <= ie + 1; i++) {
        bl[i] = al[i] - q1[i];
        br[i] = al[i + 1] - q1[i];
        b0[i] = bl[i] + br[i];
        smt5[i] = fabs(lim_fac * b0[i]) < fabs(bl[i] - br[i]);
      }
      for (i = is; i <= ie + 1; i++) {
        if (c[i] > 0.) {
          fx1 = (1. - c[i]) * (br[i - 1] - c[i] * b0[i - 1]);
          flux[i] = q1[i - 1];
        } else {
          fx1 = (1. + c[i]) * (bl[i] + c[i] * b0[i]);
          flux[i] = q1[i];
        }
        if (smt5[i - 1] || smt5[i]) flux[i] = flux[i] + fx1;
      }
    } else if (mord == 2) { /* :413-429 */
      for (i = is; i <= ie + 1; i++) {
        xt = c[i];
        if (xt > 0.) {
          qtmp = q1[i - 1];
          flux[i] = qtmp + (1. - xt) * (al[i] - qtmp - xt * (al[i - 1] + al[i] - (qtmp + qtmp)));
        } else {
          qtmp = q1[i];
          flux[i] = qtmp + (1. + xt) * (al[i] - qtmp + xt * (al[i] + al[i + 1] - (qtmp + qtmp)));
        }
      }
    } else if (mord == 3) { /* :431-457 */
      for (i = is - 1; i <= ie + 1; i++) {
        bl[i] = al[i] - q1[i];
        br[i] = al[i + 1] - q1[i];
        b0[i] = bl[i] + br[i];
        x0 = fabs(b0[i]);
        xt = fabs(bl[i] - br[i]);
        smt5[i] = x0 < xt;
        smt6[i] = 3. * x0 < xt;
      }
      for (i = is; i <= ie + 1; i++) {
        double xt1 = c[i];
        if (xt1 > 0.) {
          if (smt5[i - 1] || smt6[i])
            flux[i] = q1[i - 1] + (1. - xt1) * (br[i - 1] - xt1 * b0[i - 1]);
          else
            flux[i] = q1[i - 1];
        } else {
          if (smt6[i - 1] || smt5[i])
            flux[i] = q1[i] + (1. + xt1) * (bl[i] + xt1 * b0[i]);
          else
            flux[i] = q1[i];
        }
      }
    } else if (mord == 4) { /* :459-487 */
      for (i = is - 1; i <= ie + 1; i++) {
        bl[i] = al[i] - q1[i];
        br[i] = al[i + 1] - q1[i];
        b0[i] = bl[i] + br[i];
        x0 = fabs(b0[i]);
        xt = fabs(bl[i] - br[i]);
        smt5[i] = x0 < xt;
        smt6[i] = 3. * x0 < xt;
      }
      for (i = is; i <= ie + 1; i++) {
        double xt1 = c[i];
        int hi5 = smt5[i - 1] && smt5[i];
        int hi6 = smt6[i - 1] || smt6[i];
        hi5 = hi5 || hi6;
        if (xt1 > 0.) {
          fx1 = (1. - xt1) * (br[i - 1] - xt1 * b0[i - 1]);
          flux[i] = q1[i - 1];
        } else {
          fx1 = (1. + xt1) * (bl[i] + xt1 * b0[i]);
          flux[i] = q1[i];
        }
        if (hi5) flux[i] = flux[i] + fx1;
      }
    } else { /* mord 5, 6: :489-558 */
      if (iord == 5) {
        for (i = is - 1; i <= ie + 1; i++) {
          bl[i] = al[i] - q1[i];
          br[i] = al[i + 1] - q1[i];
          b0[i] = bl[i] + br[i];
          smt5[i] = bl[i] * br[i] < 0.;
        }
      } else if (iord == -5) {
        for (i = is - 1; i <= ie + 1; i++) {
          bl[i] = al[i] - q1[i];
          br[i] = al[i + 1] - q1[i];
          b0[i] = bl[i] + br[i];
          smt5[i] = bl[i] * br[i] < 0.;
          da1[i] = br[i] - bl[i];
          a4[i] = -3. * b0[i];
        }
        for (i = is - 1; i <= ie + 1; i++) {
          if (fabs(da1[i]) < -a4[i]) {
            if (q1[i] + 0.25 / a4[i] * (da1[i] * da1[i]) + a4[i] * r12 < 0.) {
              if (!smt5[i]) {
                br[i] = 0.;
                bl[i] = 0.;
                b0[i] = 0.;
              } else if (da1[i] > 0.) {
                br[i] = -2. * bl[i];
                b0[i] = -bl[i];
              } else {
                bl[i] = -2. * br[i];
                b0[i] = -br[i];
              }
            }
          }
        }
      } else {
        for (i = is - 1; i <= ie + 1; i++) {
          bl[i] = al[i] - q1[i];
          br[i] = al[i + 1] - q1[i];
          b0[i] = bl[i] + br[i];
          smt5[i] = 3. * fabs(b0[i]) < fabs(bl[i] - br[i]);
        }
      }
      if (cubed && iord != 5) { /* fix edge issues, :534-546 */
        if (is == 1) {
          smt5[0] = bl[0] * br[0] < 0.;
          smt5[1] = bl[1] * br[1] < 0.;
        }
        if ((ie + 1) == npx) {
          smt5[npx - 1] = bl[npx - 1] * br[npx - 1] < 0.;
          smt5[npx] = bl[npx] * br[npx] < 0.;
        }
      }
      for (i = is; i <= ie + 1; i++) { /* :549-558 */
        if (c[i] > 0.) {
          fx1 = (1. - c[i]) * (br[i - 1] - c[i] * b0[i - 1]);
          flux[i] = q1[i - 1];
        } else {
          fx1 = (1. + c[i]) * (bl[i] + c[i] * b0[i]);
          flux[i] = q1[i];
        }
        if (smt5[i - 1] || smt5[i]) flux[i] = flux[i] + fx1;
      }
    }
    goto done;
  }

  /* Monotonic constraints, iord >= 7: :570-708 */
  for (i = is - 2; i <= ie + 2; i++) {
    xt = 0.25 * (q1[i + 1] - q1[i - 1]);
    dm[i] = fsign(dmin3(fabs(xt), dmax3(q1[i - 1], q1[i], q1[i + 1]) - q1[i],
                        q1[i] - dmin3(q1[i - 1], q1[i], q1[i + 1])),
                  xt);
  }
  for (i = is1; i <= ie1 + 1; i++) al[i] = 0.5 * (q1[i - 1] + q1[i]) + r3 * (dm[i - 1] - dm[i]);

  if (iord == 8) { /* :579-584 */
    for (i = is1; i <= ie1; i++) {
      xt = 2. * dm[i];
      bl[i] = -fsign(dmin(fabs(xt), fabs(al[i] - q1[i])), xt);
      br[i] = fsign(dmin(fabs(xt), fabs(al[i + 1] - q1[i])), xt);
    }
  } else if (iord == 10) { /* :585-603 */
    for (i = is1 - 2; i <= ie1 + 1; i++) dq[i] = 2. * (q1[i + 1] - q1[i]);
    for (i = is1; i <= ie1; i++) {
      bl[i] = al[i] - q1[i];
      br[i] = al[i + 1] - q1[i];
      if (fabs(dm[i - 1]) + fabs(dm[i]) + fabs(dm[i + 1]) < near_zero) {
        bl[i] = 0.;
        br[i] = 0.;
      } else if (fabs(3. * (bl[i] + br[i])) > fabs(bl[i] - br[i])) {
        pmp_2 = dq[i - 1];
        lac_2 = pmp_2 - 0.75 * dq[i - 2];
        br[i] = dmin(dmax3(0., pmp_2, lac_2), dmax(br[i], dmin3(0., pmp_2, lac_2)));
        pmp_1 = -dq[i];
        lac_1 = pmp_1 + 0.75 * dq[i + 1];
        bl[i] = dmin(dmax3(0., pmp_1, lac_1), dmax(bl[i], dmin3(0., pmp_1, lac_1)));
      }
    }
  } else if (iord == 11) { /* :604-610 */
    for (i = is1; i <= ie1; i++) {
      xt = ppm_fac * dm[i];
      bl[i] = -fsign(dmin(fabs(xt), fabs(al[i] - q1[i])), xt);
      br[i] = fsign(dmin(fabs(xt), fabs(al[i + 1] - q1[i])), xt);
    }
  } else if (iord == 7 || iord == 12) { /* :611-633 */
    for (i = is1; i <= ie1; i++) {
      bl[i] = al[i] - q1[i];
      br[i] = al[i + 1] - q1[i];
      a4[i] = -3. * (bl[i] + br[i]);
      da1[i] = br[i] - bl[i];
      ext5[i] = br[i] * bl[i] > 0.;
      ext6[i] = fabs(da1[i]) < -a4[i];
    }
    for (i = is1; i <= ie1; i++) {
      if (ext6[i]) {
        if (q1[i] + 0.25 / a4[i] * (da1[i] * da1[i]) + a4[i] * r12 < 0.) {
          if (ext5[i]) {
            br[i] = 0.;
            bl[i] = 0.;
          } else if (da1[i] > 0.) {
            br[i] = -2. * bl[i];
          } else {
            bl[i] = -2. * br[i];
          }
        }
      }
    }
  } else { /* :634-639 */
    for (i = is1; i <= ie1; i++) {
      bl[i] = al[i] - q1[i];
      br[i] = al[i + 1] - q1[i];
    }
  }
  if (iord == 9 || iord == 13) fvo_pert_ppm(ie1 - is1 + 1, q1 + is1, bl + is1, br + is1, 0); /* :641 */

  if (cubed) { /* :643-681 */
    if (is == 1) {
      bl[0] = s14 * dm[-1] + s11 * (q1[-1] - q1[0]);
      xt = 0.5 * (((2. * dxa[0] + dxa[-1]) * q1[0] - dxa[0] * q1[-1]) / (dxa[-1] + dxa[0]) +
                  ((2. * dxa[1] + dxa[2]) * q1[1] - dxa[1] * q1[2]) / (dxa[1] + dxa[2]));
      xt = dmax(xt, dmin(dmin(q1[-1], q1[0]), dmin(q1[1], q1[2])));
      xt = dmin(xt, dmax(dmax(q1[-1], q1[0]), dmax(q1[1], q1[2])));
      br[0] = xt - q1[0];
      bl[1] = xt - q1[1];
      xt = s15 * q1[1] + s11 * q1[2] - s14 * dm[2];
      br[1] = xt - q1[1];
      bl[2] = xt - q1[2];
      br[2] = al[3] - q1[2];
      fvo_pert_ppm(3, q1 + 0, bl + 0, br + 0, 1);
    }
    if ((ie + 1) == npx) {
      bl[npx - 2] = al[npx - 2] - q1[npx - 2];
      xt = s15 * q1[npx - 1] + s11 * q1[npx - 2] + s14 * dm[npx - 2];
      br[npx - 2] = xt - q1[npx - 2];
      bl[npx - 1] = xt - q1[npx - 1];
      xt = 0.5 * (((2. * dxa[npx - 1] + dxa[npx - 2]) * q1[npx - 1] - dxa[npx - 1] * q1[npx - 2]) / (dxa[npx - 2] + dxa[npx - 1]) +
                  ((2. * dxa[npx] + dxa[npx + 1]) * q1[npx] - dxa[npx] * q1[npx + 1]) / (dxa[npx] + dxa[npx + 1]));
      xt = dmax(xt, dmin(dmin(q1[npx - 2], q1[npx - 1]), dmin(q1[npx], q1[npx + 1])));
      xt = dmin(xt, dmax(dmax(q1[npx - 2], q1[npx - 1]), dmax(q1[npx], q1[npx + 1])));
      br[npx - 1] = xt - q1[npx - 1];
      bl[npx] = xt - q1[npx];
      br[npx] = s11 * (q1[npx + 1] - q1[npx]) - s14 * dm[npx + 1];
      fvo_pert_ppm(3, q1 + npx - 2, bl + npx - 2, br + npx - 2, 1);
    }
  }

  if (iord == 7) { /* :685-699 */
    for (i = is - 1; i <= ie + 1; i++) {
      b0[i] = bl[i] + br[i];
      smt5[i] = bl[i] * br[i] < 0.;
    }
    for (i = is; i <= ie + 1; i++) {
      if (c[i] > 0.) {
        fx1 = (1. - c[i]) * (br[i - 1] - c[i] * b0[i - 1]);
        flux[i] = q1[i - 1];
      } else {
        fx1 = (1. + c[i]) * (bl[i] + c[i] * b0[i]);
        flux[i] = q1[i];
      }
      if (smt5[i - 1] || smt5[i]) flux[i] = flux[i] + fx1;
    }
  } else { /* :701-707 */
    for (i = is; i <= ie + 1; i++) {
      if (c[i] > 0.)
        flux[i] = q1[i - 1] + (1. - c[i]) * (br[i - 1] - c[i] * (bl[i - 1] + br[i - 1]));
      else
        flux[i] = q1[i] + (1. + c[i]) * (bl[i] + c[i] * (bl[i] + br[i]));
    }
  }

done:
  free(buf);
  free(lbuf);
  return rc;
}

/* ---- 2-D helpers -------------------------------------------------------------------------- */

/* copy_corners, tp_core.F90:245-322: the corner regions of the halo of q get the values the sweep in direction dir
 * (1 = x, 2 = y) should see there.  q on the A layout.  No-op unless a corner flag is set. */
void fvo_copy_corners(const fvo_grid *g, double *q, int dir) {
  const int isd = g->isd, ied = g->ied, jsd = g->jsd, ng = g->ng, npx = g->npx, npy = g->npy;
  const int nid = ied - isd + 1;
  int i, j;
  if (g->bounded_domain) return;
#define QC(i, j) q[(size_t)((j)-jsd) * nid + ((i)-isd)]
  if (dir == 1) {
    if (g->sw_corner)
      for (j = 1 - ng; j <= 0; j++)
        for (i = 1 - ng; i <= 0; i++) QC(i, j) = QC(j, 1 - i);
    if (g->se_corner)
      for (j = 1 - ng; j <= 0; j++)
        for (i = npx; i <= npx + ng - 1; i++) QC(i, j) = QC(npy - j, i - npx + 1);
    if (g->ne_corner)
      for (j = npy; j <= npy + ng - 1; j++)
        for (i = npx; i <= npx + ng - 1; i++) QC(i, j) = QC(j, 2 * npx - 1 - i);
    if (g->nw_corner)
      for (j = npy; j <= npy + ng - 1; j++)
        for (i = 1 - ng; i <= 0; i++) QC(i, j) = QC(npy - j, i - 1 + npx);
  } else if (dir == 2) {
    if (g->sw_corner)
      for (j = 1 - ng; j <= 0; j++)
        for (i = 1 - ng; i <= 0; i++) QC(i, j) = QC(1 - j, i);
    if (g->se_corner)
      for (j = 1 - ng; j <= 0; j++)
        for (i = npx; i <= npx + ng - 1; i++) QC(i, j) = QC(npy + j - 1, npx - i);
    if (g->ne_corner)
      for (j = npy; j <= npy + ng - 1; j++)
        for (i = npx; i <= npx + ng - 1; i++) QC(i, j) = QC(2 * npy - 1 - j, i);
    if (g->nw_corner)
      for (j = npy; j <= npy + ng - 1; j++)
        for (i = 1 - ng; i <= 0; i++) QC(i, j) = QC(j + 1 - npx, npy - i);
  }
#undef QC
}

/* xppm over rows jfirst..jlast.  q(isd:ied, jq0:...) with leading dimension ldq and the row
 * index origin jq0; c and flux (is:ie+1, jc0:...) with leading dimension ldc. */
static void xppm_2d(const fvo_grid *g, double *flux, const double *q, const double *c, int iord,
                    int jfirst, int jlast, int ldq, int jq0, int ldc, int jc0) {
  const int is = g->is, ie = g->ie, isd = g->isd;
  int j;
  for (j = jfirst; j <= jlast; j++) {
    const double *qrow = q + (size_t)(j - jq0) * ldq - isd; /* qrow[i] */
    const double *crow = c + (size_t)(j - jc0) * ldc - is;
    double *frow = flux + (size_t)(j - jc0) * ldc - is;
    if (g->grid_type < 3 && !g->bounded_domain) {
      const int nid = g->ied - isd + 1;
      fvo_ppm_line_cs(qrow, crow, frow, is, ie, iord, g->lim_fac, g->dxa + (size_t)(j - g->jsd) * nid - isd, g->npx);
    } else {
      fvo_ppm_line(qrow, crow, frow, is, ie, iord, g->lim_fac);
    }
  }
}

/* yppm over columns ifirst..ilast.  q(ifirst:ilast, jsd:jed) leading dim ldq (origin ifirst,jsd);
 * c(isd:ied, js:je+1) leading dim ldc (origin isd, js); flux(ifirst:ilast, js:je+1) leading dim
 * ldq (origin ifirst, js).  Gathers each column into a line buffer. */
static void yppm_2d(const fvo_grid *g, double *flux, const double *q, const double *c, int jord,
                    int ifirst, int ilast, int ldq, int ldc) {
  const int js = g->js, je = g->je, jsd = g->jsd, jed = g->jed, isd = g->isd;
  const int nj = jed - jsd + 1;
  double *line = (double *)malloc(sizeof(double) * (size_t)(4 * nj + 12));
  double *ql = line - jsd;                /* ql[j], j in jsd..jed */
  double *cl = line + nj + 2 - js;        /* cl[j], j in js..je+1 */
  double *fl = line + 2 * nj + 4 - js;    /* fl[j] */
  double *dl = line + 3 * nj + 8 - jsd;   /* dya(i, j), j in jsd..jed */
  const int cubed = g->grid_type < 3 && !g->bounded_domain;
  const int nid = g->ied - isd + 1;
  int i, j;
  for (i = ifirst; i <= ilast; i++) {
    for (j = jsd; j <= jed; j++) ql[j] = q[(size_t)(j - jsd) * ldq + (i - ifirst)];
    for (j = js; j <= je + 1; j++) cl[j] = c[(size_t)(j - js) * ldc + (i - isd)];
    if (cubed) {
      for (j = jsd; j <= jed; j++) dl[j] = g->dya[(size_t)(j - jsd) * nid + (i - isd)];
      fvo_ppm_line_cs(ql, cl, fl, js, je, jord, g->lim_fac, dl, g->npy);
    } else
      fvo_ppm_line(ql, cl, fl, js, je, jord, g->lim_fac);
    for (j = js; j <= je + 1; j++) flux[(size_t)(j - js) * ldq + (i - ifirst)] = fl[j];
  }
  free(line);
}

/* deln_flux, tp_core.F90:1267-1447 (damp_Km absent; copy_corners is a no-op on a doubly periodic tile, where no corner
 * flag is set, and fills the corner cells of d2 before each sweep on a cubed-sphere face). mass may be NULL. */
int fvo_deln_flux(const fvo_grid *g, int nord, double damp, const double *q, double *fx, double *fy,
                  const double *mass) {
  const int is = g->is, ie = g->ie, js = g->js, je = g->je;
  const int isd = g->isd, ied = g->ied, jsd = g->jsd, jed = g->jed;
  const int nid = ied - isd + 1, njd = jed - jsd + 1, nx = ie - is + 1;
  int i, j, n, nt;
  double damp2;
#define Q(i, j) q[(size_t)((j)-jsd) * nid + ((i)-isd)]
#define MASS(i, j) mass[(size_t)((j)-jsd) * nid + ((i)-isd)]
#define D2(i, j) d2[(size_t)((j)-jsd) * nid + ((i)-isd)]
#define FX2(i, j) fx2[(size_t)((j)-jsd) * (nid + 1) + ((i)-isd)]
#define FY2(i, j) fy2[(size_t)((j)-jsd) * nid + ((i)-isd)]
#define FX(i, j) fx[(size_t)((j)-js) * (nx + 1) + ((i)-is)]
#define FY(i, j) fy[(size_t)((j)-js) * nx + ((i)-is)]
#define DEL6_V(i, j) g->del6_v[(size_t)((j)-jsd) * (nid + 1) + ((i)-isd)]
#define DEL6_U(i, j) g->del6_u[(size_t)((j)-jsd) * nid + ((i)-isd)]
#define RAREA(i, j) g->rarea[(size_t)((j)-jsd) * nid + ((i)-isd)]
  double *d2 = (double *)calloc((size_t)nid * njd, sizeof(double));
  double *fx2 = (double *)calloc((size_t)(nid + 1) * njd, sizeof(double));
  double *fy2 = (double *)calloc((size_t)nid * (njd + 1), sizeof(double));
  const int i1 = is - 1 - nord, i2 = ie + 1 + nord, j1 = js - 1 - nord, j2 = je + 1 + nord;

  if (!mass) { /* :1304-1316 */
    for (j = j1; j <= j2; j++)
      for (i = i1; i <= i2; i++) D2(i, j) = damp * Q(i, j);
  } else {
    for (j = j1; j <= j2; j++)
      for (i = i1; i <= i2; i++) D2(i, j) = Q(i, j);
  }
  if (nord > 0) fvo_copy_corners(g, d2, 1); /* :1317 */
  for (j = js - nord; j <= je + nord; j++) /* :1321-1329 */
    for (i = is - nord; i <= ie + nord + 1; i++) FX2(i, j) = DEL6_V(i, j) * (D2(i - 1, j) - D2(i, j));
  if (nord > 0) fvo_copy_corners(g, d2, 2); /* :1331 */
  for (j = js - nord; j <= je + nord + 1; j++) /* :1333-1341 */
    for (i = is - nord; i <= ie + nord; i++) FY2(i, j) = DEL6_U(i, j) * (D2(i, j - 1) - D2(i, j));

  if (nord > 0) { /* :1343-1384 */
    for (n = 1; n <= nord; n++) {
      nt = nord - n;
      for (j = js - nt - 1; j <= je + nt + 1; j++)
        for (i = is - nt - 1; i <= ie + nt + 1; i++)
          D2(i, j) = (FX2(i, j) - FX2(i + 1, j) + FY2(i, j) - FY2(i, j + 1)) * RAREA(i, j);
      fvo_copy_corners(g, d2, 1); /* :1359 */
      for (j = js - nt; j <= je + nt; j++)
        for (i = is - nt; i <= ie + nt + 1; i++) FX2(i, j) = DEL6_V(i, j) * (D2(i, j) - D2(i - 1, j));
      fvo_copy_corners(g, d2, 2); /* :1371 */
      for (j = js - nt; j <= je + nt + 1; j++)
        for (i = is - nt; i <= ie + nt; i++) FY2(i, j) = DEL6_U(i, j) * (D2(i, j) - D2(i, j - 1));
    }
  }

  if (mass) { /* :1390-1417 */
    damp2 = 0.5 * damp;
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++)
        FX(i, j) = FX(i, j) + damp2 * (MASS(i - 1, j) + MASS(i, j)) * FX2(i, j);
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++)
        FY(i, j) = FY(i, j) + damp2 * (MASS(i, j - 1) + MASS(i, j)) * FY2(i, j);
  } else { /* :1432-1444 */
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++) FX(i, j) = FX(i, j) + FX2(i, j);
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++) FY(i, j) = FY(i, j) + FY2(i, j);
  }
  free(d2);
  free(fx2);
  free(fy2);
  return FVO_OK;
#undef Q
#undef MASS
#undef D2
#undef FX2
#undef FY2
#undef FX
#undef FY
#undef DEL6_V
#undef DEL6_U
#undef RAREA
}

/* fv_tp_2d, tp_core.F90:85-241 */
int fvo_fv_tp_2d(const fvo_grid *g, double *q, const double *crx, const double *cry, int hord,
                 double *fx, double *fy, const double *xfx, const double *yfx, const double *ra_x,
                 const double *ra_y, const double *mfx, const double *mfy, const double *mass,
                 int nord, double damp_c) {
  const int is = g->is, ie = g->ie, js = g->js, je = g->je;
  const int isd = g->isd, ied = g->ied, jsd = g->jsd, jed = g->jed;
  const int nid = ied - isd + 1, njd = jed - jsd + 1, nx = ie - is + 1, ny = je - js + 1;
  int i, j, ord_in, ord_ou;
  double damp;
  if (g->bounded_domain) return FVO_ERR_UNSUPPORTED;
#define Q(i, j) q[(size_t)((j)-jsd) * nid + ((i)-isd)]
#define AREA(i, j) g->area[(size_t)((j)-jsd) * nid + ((i)-isd)]
#define XFX(i, j) xfx[(size_t)((j)-jsd) * (nx + 1) + ((i)-is)]
#define YFX(i, j) yfx[(size_t)((j)-js) * nid + ((i)-isd)]
#define RA_X(i, j) ra_x[(size_t)((j)-jsd) * nx + ((i)-is)]
#define RA_Y(i, j) ra_y[(size_t)((j)-js) * nid + ((i)-isd)]
#define FX(i, j) fx[(size_t)((j)-js) * (nx + 1) + ((i)-is)]
#define FY(i, j) fy[(size_t)((j)-js) * nx + ((i)-is)]
#define MFX(i, j) mfx[(size_t)((j)-js) * (nx + 1) + ((i)-is)]
#define MFY(i, j) mfy[(size_t)((j)-js) * nx + ((i)-is)]
#define Q_I(i, j) q_i[(size_t)((j)-js) * nid + ((i)-isd)]
#define Q_J(i, j) q_j[(size_t)((j)-jsd) * nx + ((i)-is)]
#define FX2(i, j) fx2[(size_t)((j)-jsd) * (nx + 1) + ((i)-is)]
#define FY2(i, j) fy2[(size_t)((j)-js) * nid + ((i)-isd)]
#define FYY(i, j) fyy[(size_t)((j)-js) * nid + ((i)-isd)]
  double *q_i = (double *)malloc(sizeof(double) * (size_t)nid * ny);
  double *q_j = (double *)malloc(sizeof(double) * (size_t)nx * njd);
  double *fx2 = (double *)malloc(sizeof(double) * (size_t)(nx + 1) * njd);
  double *fy2 = (double *)malloc(sizeof(double) * (size_t)nid * (ny + 1));
  double *fyy = (double *)malloc(sizeof(double) * (size_t)nid * (ny + 1));
  double *fx1 = (double *)malloc(sizeof(double) * (size_t)(nx + 1));

  if (hord == 10) /* :136-141 */
    ord_in = 8;
  else
    ord_in = hord;
  ord_ou = hord;

  fvo_copy_corners(g, q, 2); /* :143-145 */
  /* :147 yppm(fy2, q, cry, ord_in, isd,ied, ...) */
  yppm_2d(g, fy2, q, cry, ord_in, isd, ied, nid, nid);
  for (j = js; j <= je + 1; j++) /* :150-154 */
    for (i = isd; i <= ied; i++) FYY(i, j) = YFX(i, j) * FY2(i, j);
  for (j = js; j <= je; j++) /* :155-159 */
    for (i = isd; i <= ied; i++)
      Q_I(i, j) = (Q(i, j) * AREA(i, j) + FYY(i, j) - FYY(i, j + 1)) / RA_Y(i, j);

  /* :161 xppm(fx, q_i, crx(is,js), ord_ou, ..., js,je) */
  xppm_2d(g, fx, q_i, &crx[(size_t)(js - jsd) * (nx + 1)], ord_ou, js, je, nid, js, nx + 1, js);
  fvo_copy_corners(g, q, 1); /* :164-166 */
  /* :168 xppm(fx2, q, crx, ord_in, ..., jsd,jed) */
  xppm_2d(g, fx2, q, crx, ord_in, jsd, jed, nid, jsd, nx + 1, jsd);

  for (j = jsd; j <= jed; j++) { /* :171-178 */
    for (i = is; i <= ie + 1; i++) fx1[i - is] = XFX(i, j) * FX2(i, j);
    for (i = is; i <= ie; i++)
      Q_J(i, j) = (Q(i, j) * AREA(i, j) + fx1[i - is] - fx1[i + 1 - is]) / RA_X(i, j);
  }
  /* :180 yppm(fy, q_j, cry, ord_ou, is,ie, ...) */
  yppm_2d(g, fy, q_j, cry, ord_ou, is, ie, nx, nid);

  if (mfx && mfy) { /* :187-212 */
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++) FX(i, j) = 0.5 * (FX(i, j) + FX2(i, j)) * MFX(i, j);
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++) FY(i, j) = 0.5 * (FY(i, j) + FY2(i, j)) * MFY(i, j);
    if (nord >= 0 && mass) {
      if (damp_c > 1.e-4) {
        damp = ipow(damp_c * g->da_min, nord + 1);
        fvo_deln_flux(g, nord, damp, q, fx, fy, mass);
      }
    }
  } else { /* :213-239 */
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++) FX(i, j) = 0.5 * (FX(i, j) + FX2(i, j)) * XFX(i, j);
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++) FY(i, j) = 0.5 * (FY(i, j) + FY2(i, j)) * YFX(i, j);
    if (nord >= 0) {
      if (damp_c > 1.E-4) {
        damp = ipow(damp_c * g->da_min, nord + 1);
        fvo_deln_flux(g, nord, damp, q, fx, fy, NULL);
      }
    }
  }
  free(q_i);
  free(q_j);
  free(fx2);
  free(fy2);
  free(fyy);
  free(fx1);
  return FVO_OK;
#undef Q
#undef AREA
#undef XFX
#undef YFX
#undef RA_X
#undef RA_Y
#undef FX
#undef FY
#undef MFX
#undef MFY
#undef Q_I
#undef Q_J
#undef FX2
#undef FY2
#undef FYY
}
