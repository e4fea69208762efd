/*
 * oracle/sw_core.c -- CPU oracle (test infrastructure, see fvo.h) for model/sw_core.F90
 * (c_sw, d_sw and their helpers) and model/a2b_edge.F90 (a2b_ord4), restated loop for loop in
 * the grid_type >= 3 (doubly periodic / Cartesian) branches with array-valued metric terms.
 */
#include "fvo.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* sw_core.F90:36-70 */
static const double r3 = 1. / 3.;
static const double near_zero = 1.E-9;
static const double big_number = 1.E30;
static const double p1 = 7. / 12.;
static const double p2 = -1. / 12.;
static const double a1 = 0.5625;
static const double a2 = -0.0625;
/* volume-conserving cubic with 2nd derivative = 0 at the end point (sw_core.F90:56-58) */
static const double c1 = -2. / 14.;
static const double c2 = 11. / 14.;
static const double c3 = 5. / 14.;
static const double s11 = 11. / 14., s14 = 4. / 7., s15 = 3. / 14.; /* :38 */

static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin3(double a, double b, double c) { return dmin(dmin(a, b), c); }
static inline double dmax3(double a, double b, double c) { return dmax(dmax(a, b), c); }
static inline double fsign(double a, double b) { return copysign(fabs(a), b); }
static inline double ipow(double x, int n) {
  double r = x;
  int k;
  for (k = 1; k < n; k++) r = r * x;
  return r;
}

/* Bounds boilerplate shared by every routine */
#define BOUNDS(g)                                                                         \
  const int is = (g)->is, ie = (g)->ie, js = (g)->js, je = (g)->je;                       \
  const int isd = (g)->isd, ied = (g)->ied, jsd = (g)->jsd, jed = (g)->jed;               \
  const int nid = ied - isd + 1, njd = jed - jsd + 1, nx = ie - is + 1, ny = je - js + 1; \
  (void)nid; (void)njd; (void)nx; (void)ny; (void)is; (void)ie; (void)js; (void)je;       \
  (void)isd; (void)ied; (void)jsd; (void)jed

/* index helpers: A = (isd:ied, jsd:jed); U = (isd:ied, jsd:jed+1); V = (isd:ied+1, jsd:jed);
 * B = (isd:ied+1, jsd:jed+1); CX = (is:ie+1, jsd:jed); CY = (isd:ied, js:je+1);
 * FXC = (is:ie+1, js:je); FYC = (is:ie, js:je+1); CC = (is:ie, js:je); BC = (is:ie+1, js:je+1) */
#define IA(i, j) ((size_t)((j)-jsd) * nid + ((i)-isd))
#define IU(i, j) ((size_t)((j)-jsd) * nid + ((i)-isd))
#define IV(i, j) ((size_t)((j)-jsd) * (nid + 1) + ((i)-isd))
#define IB(i, j) ((size_t)((j)-jsd) * (nid + 1) + ((i)-isd))
#define ICX(i, j) ((size_t)((j)-jsd) * (nx + 1) + ((i)-is))
#define ICY(i, j) ((size_t)((j)-js) * nid + ((i)-isd))
#define IFX(i, j) ((size_t)((j)-js) * (nx + 1) + ((i)-is))
#define IFY(i, j) ((size_t)((j)-js) * nx + ((i)-is))
#define ICC(i, j) ((size_t)((j)-js) * nx + ((i)-is))
#define IBC(i, j) ((size_t)((j)-js) * (nx + 1) + ((i)-is))
#define SIN_SG(i, j, n) g->sin_sg[(size_t)((n)-1) * nid * njd + IA(i, j)]
#define COS_SG(i, j, n) g->cos_sg[(size_t)((n)-1) * nid * njd + IA(i, j)]

static double *dalloc(size_t n) { return (double *)calloc(n, sizeof(double)); }

/* edge_interpolate4, sw_core.F90:3348-3359; ua[0..3], dxa[0..3] = the Fortran (1:4) */
static double edge_interpolate4(const double *ua, const double *dxa) {
  const double t1 = dxa[0] + dxa[1], t2 = dxa[2] + dxa[3];
  return 0.5 * (((t1 + dxa[1]) * ua[1] - dxa[1] * ua[0]) / t1 + ((t2 + dxa[2]) * ua[2] - dxa[2] * ua[3]) / t2);
}

/* fill_4corners, sw_core.F90:3506-3553 (one tile per face: every corner flag is tested) ; q on the A layout */
void fvo_fill_4corners(const fvo_grid *g, double *q, int dir) {
  BOUNDS(g);
  const int npx = g->npx, npy = g->npy;
  if (dir == 1) {
    if (g->sw_corner) { q[IA(-1, 0)] = q[IA(0, 2)]; q[IA(0, 0)] = q[IA(0, 1)]; }
    if (g->se_corner) { q[IA(npx + 1, 0)] = q[IA(npx, 2)]; q[IA(npx, 0)] = q[IA(npx, 1)]; }
    if (g->nw_corner) { q[IA(0, npy)] = q[IA(0, npy - 1)]; q[IA(-1, npy)] = q[IA(0, npy - 2)]; }
    if (g->ne_corner) { q[IA(npx, npy)] = q[IA(npx, npy - 1)]; q[IA(npx + 1, npy)] = q[IA(npx, npy - 2)]; }
  } else {
    if (g->sw_corner) { q[IA(0, 0)] = q[IA(1, 0)]; q[IA(0, -1)] = q[IA(2, 0)]; }
    if (g->se_corner) { q[IA(npx, 0)] = q[IA(npx - 1, 0)]; q[IA(npx, -1)] = q[IA(npx - 2, 0)]; }
    if (g->nw_corner) { q[IA(0, npy)] = q[IA(1, npy)]; q[IA(0, npy + 1)] = q[IA(2, npy)]; }
    if (g->ne_corner) { q[IA(npx, npy)] = q[IA(npx - 1, npy)]; q[IA(npx, npy + 1)] = q[IA(npx - 2, npy)]; }
  }
}

/* ------------------------------------------------------------------------------------------
 * d2a2c_vect, sw_core.F90:3006-3345: grid_type >= 3 (npt = -2, no edge handling) and the cubed sphere
 * (grid_type < 3, not bounded: npt = 4, face edges and corners)
 * ---------------------------------------------------------------------------------------- */
#define IMAX(a, b) ((a) > (b) ? (a) : (b))
#define IMIN(a, b) ((a) < (b) ? (a) : (b))
int fvo_d2a2c_vect(const fvo_grid *g, const double *u, const double *v, double *ua, double *va,
                   double *uc, double *vc, double *ut, double *vt, int dord4) {
  BOUNDS(g);
  const int npx = g->npx, npy = g->npy;
  const int cubed = g->grid_type < 3;
  int i, j, id, npt, ifirst, ilast;
  if (g->bounded_domain) return FVO_ERR_UNSUPPORTED;
  double *utmp = dalloc((size_t)nid * njd), *vtmp = dalloc((size_t)nid * njd);
  id = dord4 ? 1 : 0;
  npt = cubed ? 4 : -2; /* :3054-3058 */
  for (i = 0; i < nid * njd; i++) { /* :3061-3062 */
    utmp[i] = big_number;
    vtmp[i] = big_number;
  }
  /* Interior, :3099-3108 */
  for (j = IMAX(npt, js - 1); j <= IMIN(npy - npt, je + 1); j++)
    for (i = IMAX(npt, isd); i <= IMIN(npx - npt, ied); i++)
      utmp[IA(i, j)] = a2 * (u[IU(i, j - 1)] + u[IU(i, j + 2)]) + a1 * (u[IU(i, j)] + u[IU(i, j + 1)]);
  for (j = IMAX(npt, jsd); j <= IMIN(npy - npt, jed); j++)
    for (i = IMAX(npt, is - 1); i <= IMIN(npx - npt, ie + 1); i++)
      vtmp[IA(i, j)] = a2 * (v[IV(i - 1, j)] + v[IV(i + 2, j)]) + a1 * (v[IV(i, j)] + v[IV(i + 1, j)]);
  if (cubed) { /* edges, :3113-3149 */
    if (js == 1 || jsd < npt)
      for (j = jsd; j <= npt - 1; j++)
        for (i = isd; i <= ied; i++) {
          utmp[IA(i, j)] = 0.5 * (u[IU(i, j)] + u[IU(i, j + 1)]);
          vtmp[IA(i, j)] = 0.5 * (v[IV(i, j)] + v[IV(i + 1, j)]);
        }
    if ((je + 1) == npy || jed >= (npy - npt))
      for (j = npy - npt + 1; j <= jed; j++)
        for (i = isd; i <= ied; i++) {
          utmp[IA(i, j)] = 0.5 * (u[IU(i, j)] + u[IU(i, j + 1)]);
          vtmp[IA(i, j)] = 0.5 * (v[IV(i, j)] + v[IV(i + 1, j)]);
        }
    if (is == 1 || isd < npt)
      for (j = IMAX(npt, jsd); j <= IMIN(npy - npt, jed); j++)
        for (i = isd; i <= npt - 1; i++) {
          utmp[IA(i, j)] = 0.5 * (u[IU(i, j)] + u[IU(i, j + 1)]);
          vtmp[IA(i, j)] = 0.5 * (v[IV(i, j)] + v[IV(i + 1, j)]);
        }
    if ((ie + 1) == npx || ied >= (npx - npt))
      for (j = IMAX(npt, jsd); j <= IMIN(npy - npt, jed); j++)
        for (i = npx - npt + 1; i <= ied; i++) {
          utmp[IA(i, j)] = 0.5 * (u[IU(i, j)] + u[IU(i, j + 1)]);
          vtmp[IA(i, j)] = 0.5 * (v[IV(i, j)] + v[IV(i + 1, j)]);
        }
  }
  /* Contra-variant components at cell center, :3152-3157 */
  for (j = js - 1 - id; j <= je + 1 + id; j++)
    for (i = is - 1 - id; i <= ie + 1 + id; i++) {
      ua[IA(i, j)] = (utmp[IA(i, j)] - vtmp[IA(i, j)] * g->cosa_s[IA(i, j)]) * g->rsin2[IA(i, j)];
      va[IA(i, j)] = (vtmp[IA(i, j)] - utmp[IA(i, j)] * g->cosa_s[IA(i, j)]) * g->rsin2[IA(i, j)];
    }
  /* A -> C: fix the edges, Xdir :3166-3185 */
  if (g->sw_corner) for (i = -2; i <= 0; i++) utmp[IA(i, 0)] = -vtmp[IA(0, 1 - i)];
  if (g->se_corner) for (i = 0; i <= 2; i++) utmp[IA(npx + i, 0)] = vtmp[IA(npx, i + 1)];
  if (g->ne_corner) for (i = 0; i <= 2; i++) utmp[IA(npx + i, npy)] = -vtmp[IA(npx, je - i)];
  if (g->nw_corner) for (i = -2; i <= 0; i++) utmp[IA(i, npy)] = vtmp[IA(0, je + i)];
  if (cubed) { /* :3187-3193 */
    ifirst = IMAX(3, is - 1);
    ilast = IMIN(npx - 2, ie + 2);
  } else {
    ifirst = is - 1;
    ilast = ie + 2;
  }
  for (j = js - 1; j <= je + 1; j++) /* :3197-3202 */
    for (i = ifirst; i <= ilast; i++) {
      uc[IV(i, j)] = a2 * (utmp[IA(i - 2, j)] + utmp[IA(i + 1, j)]) + a1 * (utmp[IA(i - 1, j)] + utmp[IA(i, j)]);
      ut[IA(i, j)] = (uc[IV(i, j)] - v[IV(i, j)] * g->cosa_u[IV(i, j)]) * g->rsin_u[IV(i, j)];
    }
  if (cubed) { /* :3204-3255 */
    if (g->sw_corner) { ua[IA(-1, 0)] = -va[IA(0, 2)]; ua[IA(0, 0)] = -va[IA(0, 1)]; }
    if (g->se_corner) { ua[IA(npx, 0)] = va[IA(npx, 1)]; ua[IA(npx + 1, 0)] = va[IA(npx, 2)]; }
    if (g->ne_corner) { ua[IA(npx, npy)] = -va[IA(npx, npy - 1)]; ua[IA(npx + 1, npy)] = -va[IA(npx, npy - 2)]; }
    if (g->nw_corner) { ua[IA(-1, npy)] = va[IA(0, npy - 2)]; ua[IA(0, npy)] = va[IA(0, npy - 1)]; }
    if (is == 1)
      for (j = js - 1; j <= je + 1; j++) {
        double ua4[4], dx4[4];
        int m;
        uc[IV(0, j)] = c1 * utmp[IA(-2, j)] + c2 * utmp[IA(-1, j)] + c3 * utmp[IA(0, j)];
        for (m = 0; m < 4; m++) { ua4[m] = ua[IA(-1 + m, j)]; dx4[m] = g->dxa[IA(-1 + m, j)]; }
        ut[IA(1, j)] = edge_interpolate4(ua4, dx4);
        /* Want to use the UPSTREAM value */
        if (ut[IA(1, j)] > 0.)
          uc[IV(1, j)] = ut[IA(1, j)] * SIN_SG(0, j, 3);
        else
          uc[IV(1, j)] = ut[IA(1, j)] * SIN_SG(1, j, 1);
        uc[IV(2, j)] = c1 * utmp[IA(3, j)] + c2 * utmp[IA(2, j)] + c3 * utmp[IA(1, j)];
        ut[IA(0, j)] = (uc[IV(0, j)] - v[IV(0, j)] * g->cosa_u[IV(0, j)]) * g->rsin_u[IV(0, j)];
        ut[IA(2, j)] = (uc[IV(2, j)] - v[IV(2, j)] * g->cosa_u[IV(2, j)]) * g->rsin_u[IV(2, j)];
      }
    if ((ie + 1) == npx)
      for (j = js - 1; j <= je + 1; j++) {
        double ua4[4], dx4[4];
        int m;
        uc[IV(npx - 1, j)] = c1 * utmp[IA(npx - 3, j)] + c2 * utmp[IA(npx - 2, j)] + c3 * utmp[IA(npx - 1, j)];
        for (m = 0; m < 4; m++) { ua4[m] = ua[IA(npx - 2 + m, j)]; dx4[m] = g->dxa[IA(npx - 2 + m, j)]; }
        ut[IA(npx, j)] = edge_interpolate4(ua4, dx4);
        if (ut[IA(npx, j)] > 0.)
          uc[IV(npx, j)] = ut[IA(npx, j)] * SIN_SG(npx - 1, j, 3);
        else
          uc[IV(npx, j)] = ut[IA(npx, j)] * SIN_SG(npx, j, 1);
        uc[IV(npx + 1, j)] = c3 * utmp[IA(npx, j)] + c2 * utmp[IA(npx + 1, j)] + c1 * utmp[IA(npx + 2, j)];
        ut[IA(npx - 1, j)] = (uc[IV(npx - 1, j)] - v[IV(npx - 1, j)] * g->cosa_u[IV(npx - 1, j)]) * g->rsin_u[IV(npx - 1, j)];
        ut[IA(npx + 1, j)] = (uc[IV(npx + 1, j)] - v[IV(npx + 1, j)] * g->cosa_u[IV(npx + 1, j)]) * g->rsin_u[IV(npx + 1, j)];
      }
  }
  /* Ydir, :3260-3296 */
  if (g->sw_corner) for (j = -2; j <= 0; j++) vtmp[IA(0, j)] = -utmp[IA(1 - j, 0)];
  if (g->nw_corner) for (j = 0; j <= 2; j++) vtmp[IA(0, npy + j)] = utmp[IA(j + 1, npy)];
  if (g->se_corner) for (j = -2; j <= 0; j++) vtmp[IA(npx, j)] = utmp[IA(ie + j, 0)];
  if (g->ne_corner) for (j = 0; j <= 2; j++) vtmp[IA(npx, npy + j)] = -utmp[IA(ie - j, npy)];
  if (g->sw_corner) { va[IA(0, -1)] = -ua[IA(2, 0)]; va[IA(0, 0)] = -ua[IA(1, 0)]; }
  if (g->se_corner) { va[IA(npx, 0)] = ua[IA(npx - 1, 0)]; va[IA(npx, -1)] = ua[IA(npx - 2, 0)]; }
  if (g->ne_corner) { va[IA(npx, npy)] = -ua[IA(npx - 1, npy)]; va[IA(npx, npy + 1)] = -ua[IA(npx - 2, npy)]; }
  if (g->nw_corner) { va[IA(0, npy)] = ua[IA(1, npy)]; va[IA(0, npy + 1)] = ua[IA(2, npy)]; }
  if (cubed) { /* :3298-3334 */
    for (j = js - 1; j <= je + 2; j++) {
      if (j == 1 || j == npy) {
        for (i = is - 1; i <= ie + 1; i++) {
          double va4[4], dy4[4];
          int m;
          for (m = 0; m < 4; m++) { va4[m] = va[IA(i, j - 2 + m)]; dy4[m] = g->dya[IA(i, j - 2 + m)]; }
          vt[IA(i, j)] = edge_interpolate4(va4, dy4);
          if (vt[IA(i, j)] > 0.)
            vc[IU(i, j)] = vt[IA(i, j)] * SIN_SG(i, j - 1, 4);
          else
            vc[IU(i, j)] = vt[IA(i, j)] * SIN_SG(i, j, 2);
        }
      } else if (j == 0 || j == (npy - 1)) {
        for (i = is - 1; i <= ie + 1; i++) {
          vc[IU(i, j)] = c1 * vtmp[IA(i, j - 2)] + c2 * vtmp[IA(i, j - 1)] + c3 * vtmp[IA(i, j)];
          vt[IA(i, j)] = (vc[IU(i, j)] - u[IU(i, j)] * g->cosa_v[IU(i, j)]) * g->rsin_v[IU(i, j)];
        }
      } else if (j == 2 || j == (npy + 1)) {
        for (i = is - 1; i <= ie + 1; i++) {
          vc[IU(i, j)] = c1 * vtmp[IA(i, j + 1)] + c2 * vtmp[IA(i, j)] + c3 * vtmp[IA(i, j - 1)];
          vt[IA(i, j)] = (vc[IU(i, j)] - u[IU(i, j)] * g->cosa_v[IU(i, j)]) * g->rsin_v[IU(i, j)];
        }
      } else {
        for (i = is - 1; i <= ie + 1; i++) {
          vc[IU(i, j)] = a2 * (vtmp[IA(i, j - 2)] + vtmp[IA(i, j + 1)]) + a1 * (vtmp[IA(i, j - 1)] + vtmp[IA(i, j)]);
          vt[IA(i, j)] = (vc[IU(i, j)] - u[IU(i, j)] * g->cosa_v[IU(i, j)]) * g->rsin_v[IU(i, j)];
        }
      }
    }
  } else { /* :3336-3342 */
    for (j = js - 1; j <= je + 2; j++)
      for (i = is - 1; i <= ie + 1; i++) {
        vc[IU(i, j)] = a2 * (vtmp[IA(i, j - 2)] + vtmp[IA(i, j + 1)]) + a1 * (vtmp[IA(i, j - 1)] + vtmp[IA(i, j)]);
        vt[IA(i, j)] = vc[IU(i, j)];
      }
  }
  free(utmp);
  free(vtmp);
  return FVO_OK;
}

/* divergence_corner, sw_core.F90:1740-1845: grid_type > 3 (:1781-1796) and the non-orthogonal form (:1798-1843) */
int fvo_divergence_corner(const fvo_grid *g, const double *u, const double *v, const double *ua,
                          const double *va, double *divg_d) {
  BOUNDS(g);
  const int npx = g->npx, npy = g->npy;
  int i, j;
  if (g->bounded_domain) return FVO_ERR_UNSUPPORTED;
  double *uf = dalloc((size_t)nid * (njd + 1)), *vf = dalloc((size_t)(nid + 1) * njd);
  if (g->grid_type > 3) {
    for (j = js - 1; j <= je + 2; j++)
      for (i = is - 2; i <= ie + 2; i++) uf[IU(i, j)] = u[IU(i, j)] * g->dyc[IU(i, j)];
    for (j = js - 2; j <= je + 2; j++)
      for (i = is - 1; i <= ie + 2; i++) vf[IV(i, j)] = v[IV(i, j)] * g->dxc[IV(i, j)];
    for (j = js - 1; j <= je + 2; j++)
      for (i = is - 1; i <= ie + 2; i++)
        divg_d[IB(i, j)] = g->rarea_c[IB(i, j)] * (vf[IV(i, j - 1)] - vf[IV(i, j)] + uf[IU(i - 1, j)] - uf[IU(i, j)]);
  } else {
    const int is2 = IMAX(2, is), ie1 = IMIN(npx - 1, ie + 1);
    for (j = js; j <= je + 1; j++) {
      if (j == 1 || j == npy) {
        for (i = is - 1; i <= ie + 1; i++)
          uf[IU(i, j)] = u[IU(i, j)] * g->dyc[IU(i, j)] * 0.5 * (SIN_SG(i, j - 1, 4) + SIN_SG(i, j, 2));
      } else {
        for (i = is - 1; i <= ie + 1; i++)
          uf[IU(i, j)] = (u[IU(i, j)] - 0.25 * (va[IA(i, j - 1)] + va[IA(i, j)]) * (COS_SG(i, j - 1, 4) + COS_SG(i, j, 2))) *
                         g->dyc[IU(i, j)] * 0.5 * (SIN_SG(i, j - 1, 4) + SIN_SG(i, j, 2));
      }
    }
    for (j = js - 1; j <= je + 1; j++) {
      for (i = is2; i <= ie1; i++)
        vf[IV(i, j)] = (v[IV(i, j)] - 0.25 * (ua[IA(i - 1, j)] + ua[IA(i, j)]) * (COS_SG(i - 1, j, 3) + COS_SG(i, j, 1))) *
                       g->dxc[IV(i, j)] * 0.5 * (SIN_SG(i - 1, j, 3) + SIN_SG(i, j, 1));
      if (is == 1) vf[IV(1, j)] = v[IV(1, j)] * g->dxc[IV(1, j)] * 0.5 * (SIN_SG(0, j, 3) + SIN_SG(1, j, 1));
      if ((ie + 1) == npx)
        vf[IV(npx, j)] = v[IV(npx, j)] * g->dxc[IV(npx, j)] * 0.5 * (SIN_SG(npx - 1, j, 3) + SIN_SG(npx, j, 1));
    }
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++)
        divg_d[IB(i, j)] = vf[IV(i, j - 1)] - vf[IV(i, j)] + uf[IU(i - 1, j)] - uf[IU(i, j)];
    /* Remove the extra term at the corners */
    if (g->sw_corner) divg_d[IB(1, 1)] = divg_d[IB(1, 1)] - vf[IV(1, 0)];
    if (g->se_corner) divg_d[IB(npx, 1)] = divg_d[IB(npx, 1)] - vf[IV(npx, 0)];
    if (g->ne_corner) divg_d[IB(npx, npy)] = divg_d[IB(npx, npy)] + vf[IV(npx, npy)];
    if (g->nw_corner) divg_d[IB(1, npy)] = divg_d[IB(1, npy)] + vf[IV(1, npy)];
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) divg_d[IB(i, j)] = g->rarea_c[IB(i, j)] * divg_d[IB(i, j)];
  }
  free(uf);
  free(vf);
  return FVO_OK;
}

/* c_sw, sw_core.F90:79-488 */
int fvo_c_sw(const fvo_grid *g, double *delpc, double *delp, double *ptc, double *pt, double *u,
             double *v, double *w, double *uc, double *vc, double *ua, double *va, double *wc,
             double *ut, double *vt, double *divg_d, int nord, double dt2, int hydrostatic,
             int dord4) {
  BOUNDS(g);
  int i, j, rc;
  const int iep1 = ie + 1, jep1 = je + 1;
  double dt4;
  const int npx = g->npx, npy = g->npy;
  const int cubed = g->grid_type < 3;
  if (g->bounded_domain) return FVO_ERR_UNSUPPORTED;
  /* local (is-1:ie+2, js-1:je+2) work arrays; allocate on the A shape for simplicity */
  double *vort = dalloc((size_t)nid * njd), *ke = dalloc((size_t)nid * njd);
  double *fx = dalloc((size_t)nid * njd), *fx1 = dalloc((size_t)nid * njd), *fx2 = dalloc((size_t)nid * njd);
  double *fy = dalloc((size_t)nid * njd), *fy1 = dalloc((size_t)nid * njd), *fy2 = dalloc((size_t)nid * njd);

  rc = fvo_d2a2c_vect(g, u, v, ua, va, uc, vc, ut, vt, dord4); /* :148 */
  if (rc) goto out;
  if (nord > 0) { /* :151-157 */
    rc = fvo_divergence_corner(g, u, v, ua, va, divg_d);
    if (rc) goto out;
  }
  for (j = js - 1; j <= jep1; j++) /* :159-167 */
    for (i = is - 1; i <= iep1 + 1; i++) {
      if (ut[IA(i, j)] > 0.)
        ut[IA(i, j)] = dt2 * ut[IA(i, j)] * g->dy[IV(i, j)] * SIN_SG(i - 1, j, 3);
      else
        ut[IA(i, j)] = dt2 * ut[IA(i, j)] * g->dy[IV(i, j)] * SIN_SG(i, j, 1);
    }
  for (j = js - 1; j <= je + 2; j++) /* :168-176 */
    for (i = is - 1; i <= iep1; i++) {
      if (vt[IA(i, j)] > 0.)
        vt[IA(i, j)] = dt2 * vt[IA(i, j)] * g->dx[IU(i, j)] * SIN_SG(i, j - 1, 4);
      else
        vt[IA(i, j)] = dt2 * vt[IA(i, j)] * g->dx[IU(i, j)] * SIN_SG(i, j, 2);
    }

  /* Transport delp: Xdir */
  if (cubed) { /* :182 fill2_4corners(delp, pt, 1) */
    fvo_fill_4corners(g, delp, 1);
    fvo_fill_4corners(g, pt, 1);
    if (!hydrostatic) fvo_fill_4corners(g, w, 1); /* :212 */
  }
  if (hydrostatic) { /* :197-209 */
    for (j = js - 1; j <= jep1; j++)
      for (i = is - 1; i <= ie + 2; i++) {
        if (ut[IA(i, j)] > 0.) {
          fx1[IA(i, j)] = delp[IA(i - 1, j)];
          fx[IA(i, j)] = pt[IA(i - 1, j)];
        } else {
          fx1[IA(i, j)] = delp[IA(i, j)];
          fx[IA(i, j)] = pt[IA(i, j)];
        }
        fx1[IA(i, j)] = ut[IA(i, j)] * fx1[IA(i, j)];
        fx[IA(i, j)] = fx1[IA(i, j)] * fx[IA(i, j)];
      }
  } else { /* :214-229 */
    for (j = js - 1; j <= je + 1; j++)
      for (i = is - 1; i <= ie + 2; i++) {
        if (ut[IA(i, j)] > 0.) {
          fx1[IA(i, j)] = delp[IA(i - 1, j)];
          fx[IA(i, j)] = pt[IA(i - 1, j)];
          fx2[IA(i, j)] = w[IA(i - 1, j)];
        } else {
          fx1[IA(i, j)] = delp[IA(i, j)];
          fx[IA(i, j)] = pt[IA(i, j)];
          fx2[IA(i, j)] = w[IA(i, j)];
        }
        fx1[IA(i, j)] = ut[IA(i, j)] * fx1[IA(i, j)];
        fx[IA(i, j)] = fx1[IA(i, j)] * fx[IA(i, j)];
        fx2[IA(i, j)] = fx1[IA(i, j)] * fx2[IA(i, j)];
      }
  }
  /* Ydir */
  if (cubed) { /* :233, :260 */
    fvo_fill_4corners(g, delp, 2);
    fvo_fill_4corners(g, pt, 2);
    if (!hydrostatic) fvo_fill_4corners(g, w, 2);
  }
  if (hydrostatic) { /* :235-258 */
    for (j = js - 1; j <= jep1 + 1; j++)
      for (i = is - 1; i <= iep1; i++) {
        if (vt[IA(i, j)] > 0.) {
          fy1[IA(i, j)] = delp[IA(i, j - 1)];
          fy[IA(i, j)] = pt[IA(i, j - 1)];
        } else {
          fy1[IA(i, j)] = delp[IA(i, j)];
          fy[IA(i, j)] = pt[IA(i, j)];
        }
        fy1[IA(i, j)] = vt[IA(i, j)] * fy1[IA(i, j)];
        fy[IA(i, j)] = fy1[IA(i, j)] * fy[IA(i, j)];
      }
    for (j = js - 1; j <= jep1; j++)
      for (i = is - 1; i <= iep1; i++) {
        delpc[IA(i, j)] = delp[IA(i, j)] +
                          (fx1[IA(i, j)] - fx1[IA(i + 1, j)] + fy1[IA(i, j)] - fy1[IA(i, j + 1)]) * g->rarea[IA(i, j)];
        ptc[IA(i, j)] = (pt[IA(i, j)] * delp[IA(i, j)] +
                         (fx[IA(i, j)] - fx[IA(i + 1, j)] + fy[IA(i, j)] - fy[IA(i, j + 1)]) * g->rarea[IA(i, j)]) /
                        delpc[IA(i, j)];
      }
  } else { /* :261-285 */
    for (j = js - 1; j <= je + 2; j++)
      for (i = is - 1; i <= ie + 1; i++) {
        if (vt[IA(i, j)] > 0.) {
          fy1[IA(i, j)] = delp[IA(i, j - 1)];
          fy[IA(i, j)] = pt[IA(i, j - 1)];
          fy2[IA(i, j)] = w[IA(i, j - 1)];
        } else {
          fy1[IA(i, j)] = delp[IA(i, j)];
          fy[IA(i, j)] = pt[IA(i, j)];
          fy2[IA(i, j)] = w[IA(i, j)];
        }
        fy1[IA(i, j)] = vt[IA(i, j)] * fy1[IA(i, j)];
        fy[IA(i, j)] = fy1[IA(i, j)] * fy[IA(i, j)];
        fy2[IA(i, j)] = fy1[IA(i, j)] * fy2[IA(i, j)];
      }
    for (j = js - 1; j <= je + 1; j++)
      for (i = is - 1; i <= ie + 1; i++) {
        delpc[IA(i, j)] = delp[IA(i, j)] +
                          (fx1[IA(i, j)] - fx1[IA(i + 1, j)] + fy1[IA(i, j)] - fy1[IA(i, j + 1)]) * g->rarea[IA(i, j)];
        ptc[IA(i, j)] = (pt[IA(i, j)] * delp[IA(i, j)] +
                         (fx[IA(i, j)] - fx[IA(i + 1, j)] + fy[IA(i, j)] - fy[IA(i, j + 1)]) * g->rarea[IA(i, j)]) /
                        delpc[IA(i, j)];
        wc[IA(i, j)] = (w[IA(i, j)] * delp[IA(i, j)] +
                        (fx2[IA(i, j)] - fx2[IA(i + 1, j)] + fy2[IA(i, j)] - fy2[IA(i, j + 1)]) * g->rarea[IA(i, j)]) /
                       delpc[IA(i, j)];
      }
  }

  /* Compute KE, :297-315 (bounded_domain .or. grid_type>=3) / :316-359 (cubed sphere) */
  for (j = js - 1; j <= jep1; j++)
    for (i = is - 1; i <= iep1; i++) {
      if (ua[IA(i, j)] > 0.) {
        if (cubed && i == 1)
          ke[IA(1, j)] = uc[IV(1, j)] * SIN_SG(1, j, 1) + v[IV(1, j)] * COS_SG(1, j, 1);
        else if (cubed && i == npx)
          ke[IA(i, j)] = uc[IV(npx, j)] * SIN_SG(npx, j, 1) + v[IV(npx, j)] * COS_SG(npx, j, 1);
        else
          ke[IA(i, j)] = uc[IV(i, j)];
      } else {
        if (cubed && i == 0)
          ke[IA(0, j)] = uc[IV(1, j)] * SIN_SG(0, j, 3) + v[IV(1, j)] * COS_SG(0, j, 3);
        else if (cubed && i == (npx - 1))
          ke[IA(i, j)] = uc[IV(npx, j)] * SIN_SG(npx - 1, j, 3) + v[IV(npx, j)] * COS_SG(npx - 1, j, 3);
        else
          ke[IA(i, j)] = uc[IV(i + 1, j)];
      }
    }
  for (j = js - 1; j <= jep1; j++)
    for (i = is - 1; i <= iep1; i++) {
      if (va[IA(i, j)] > 0.) {
        if (cubed && j == 1)
          vort[IA(i, 1)] = vc[IU(i, 1)] * SIN_SG(i, 1, 2) + u[IU(i, 1)] * COS_SG(i, 1, 2);
        else if (cubed && j == npy)
          vort[IA(i, j)] = vc[IU(i, npy)] * SIN_SG(i, npy, 2) + u[IU(i, npy)] * COS_SG(i, npy, 2);
        else
          vort[IA(i, j)] = vc[IU(i, j)];
      } else {
        if (cubed && j == 0)
          vort[IA(i, 0)] = vc[IU(i, 1)] * SIN_SG(i, 0, 4) + u[IU(i, 1)] * COS_SG(i, 0, 4);
        else if (cubed && j == (npy - 1))
          vort[IA(i, j)] = vc[IU(i, npy)] * SIN_SG(i, npy - 1, 4) + u[IU(i, npy)] * COS_SG(i, npy - 1, 4);
        else
          vort[IA(i, j)] = vc[IU(i, j + 1)];
      }
    }
  dt4 = 0.5 * dt2; /* :361-366 */
  for (j = js - 1; j <= jep1; j++)
    for (i = is - 1; i <= iep1; i++)
      ke[IA(i, j)] = dt4 * (ua[IA(i, j)] * ke[IA(i, j)] + va[IA(i, j)] * vort[IA(i, j)]);

  /* circulation on C grid, :372-388 */
  for (j = js - 1; j <= je + 1; j++)
    for (i = is; i <= ie + 1; i++) fx[IA(i, j)] = uc[IV(i, j)] * g->dxc[IV(i, j)];
  for (j = js; j <= je + 1; j++)
    for (i = is - 1; i <= ie + 1; i++) fy[IA(i, j)] = vc[IU(i, j)] * g->dyc[IU(i, j)];
  for (j = js; j <= je + 1; j++)
    for (i = is; i <= ie + 1; i++)
      vort[IA(i, j)] = fx[IA(i, j - 1)] - fx[IA(i, j)] - fy[IA(i - 1, j)] + fy[IA(i, j)];
  /* Remove the extra term at the corners, :390-394 */
  if (g->sw_corner) vort[IA(1, 1)] = vort[IA(1, 1)] + fy[IA(0, 1)];
  if (g->se_corner) vort[IA(npx, 1)] = vort[IA(npx, 1)] - fy[IA(npx, 1)];
  if (g->ne_corner) vort[IA(npx, npy)] = vort[IA(npx, npy)] - fy[IA(npx, npy)];
  if (g->nw_corner) vort[IA(1, npy)] = vort[IA(1, npy)] + fy[IA(0, npy)];
  /* absolute vorticity, :399-403 */
  for (j = js; j <= je + 1; j++)
    for (i = is; i <= ie + 1; i++) vort[IA(i, j)] = g->fC[IB(i, j)] + g->rarea_c[IB(i, j)] * vort[IA(i, j)];

  /* transport absolute vorticity, :414-434 */
  for (j = js; j <= je; j++)
    for (i = is; i <= iep1; i++) {
      if (cubed && (i == 1 || i == npx)) /* :437-441 */
        fy1[IA(i, j)] = dt2 * v[IV(i, j)];
      else
        fy1[IA(i, j)] = dt2 * (v[IV(i, j)] - uc[IV(i, j)] * g->cosa_u[IV(i, j)]) / g->sina_u[IV(i, j)];
      if (fy1[IA(i, j)] > 0.)
        fy[IA(i, j)] = vort[IA(i, j)];
      else
        fy[IA(i, j)] = vort[IA(i, j + 1)];
    }
  for (j = js; j <= jep1; j++)
    for (i = is; i <= ie; i++) {
      if (cubed && (j == 1 || j == npy)) /* :450-459 */
        fx1[IA(i, j)] = dt2 * u[IU(i, j)];
      else
        fx1[IA(i, j)] = dt2 * (u[IU(i, j)] - vc[IU(i, j)] * g->cosa_v[IU(i, j)]) / g->sina_v[IU(i, j)];
      if (fx1[IA(i, j)] > 0.)
        fx[IA(i, j)] = vort[IA(i, j)];
      else
        fx[IA(i, j)] = vort[IA(i + 1, j)];
    }
  /* Update time-centered winds on the C-Grid, :477-486 */
  for (j = js; j <= je; j++)
    for (i = is; i <= iep1; i++)
      uc[IV(i, j)] = uc[IV(i, j)] + fy1[IA(i, j)] * fy[IA(i, j)] + g->rdxc[IV(i, j)] * (ke[IA(i - 1, j)] - ke[IA(i, j)]);
  for (j = js; j <= jep1; j++)
    for (i = is; i <= ie; i++)
      vc[IU(i, j)] = vc[IU(i, j)] - fx1[IA(i, j)] * fx[IA(i, j)] + g->rdyc[IU(i, j)] * (ke[IA(i, j - 1)] - ke[IA(i, j)]);
out:
  free(vort);
  free(ke);
  free(fx);
  free(fx1);
  free(fx2);
  free(fy);
  free(fy1);
  free(fy2);
  return rc;
}

/* del6_vt_flux, sw_core.F90:1608-1737 (damp_Km absent; copy_corners no-op) */
int fvo_del6_vt_flux(const fvo_grid *g, int nord, double damp, const double *q, double *d2,
                     double *fx2, double *fy2) {
  BOUNDS(g);
  int i, j, n, nt;
  const int i1 = is - 1 - nord, i2 = ie + 1 + nord, j1 = js - 1 - nord, j2 = je + 1 + nord;
  for (j = j1; j <= j2; j++)
    for (i = i1; i <= i2; i++) d2[IA(i, j)] = damp * q[IA(i, j)];
  if (nord > 0) fvo_copy_corners(g, d2, 1); /* :1653 (no-op without corner flags) */
  for (j = js - nord; j <= je + nord; j++)
    for (i = is - nord; i <= ie + nord + 1; i++)
      fx2[IV(i, j)] = g->del6_v[IV(i, j)] * (d2[IA(i - 1, j)] - d2[IA(i, j)]);
  if (nord > 0) fvo_copy_corners(g, d2, 2); /* :1666 */
  for (j = js - nord; j <= je + nord + 1; j++)
    for (i = is - nord; i <= ie + nord; i++)
      fy2[IU(i, j)] = g->del6_u[IU(i, j)] * (d2[IA(i, j - 1)] - d2[IA(i, j)]);
  if (nord > 0) {
    for (n = 1; n <= nord; n++) {
      nt = nord - n;
      for (j = js - nt - 1; j <= je + nt + 1; j++)
        for (i = is - nt - 1; i <= ie + nt + 1; i++)
          d2[IA(i, j)] = (fx2[IV(i, j)] - fx2[IV(i + 1, j)] + fy2[IU(i, j)] - fy2[IU(i, j + 1)]) * g->rarea[IA(i, j)];
      fvo_copy_corners(g, d2, 1); /* :1691 */
      for (j = js - nt; j <= je + nt; j++)
        for (i = is - nt; i <= ie + nt + 1; i++)
          fx2[IV(i, j)] = g->del6_v[IV(i, j)] * (d2[IA(i, j)] - d2[IA(i - 1, j)]);
      fvo_copy_corners(g, d2, 2); /* :1703 */
      for (j = js - nt; j <= je + nt + 1; j++)
        for (i = is - nt; i <= ie + nt; i++)
          fy2[IU(i, j)] = g->del6_u[IU(i, j)] * (d2[IA(i, j)] - d2[IA(i, j - 1)]);
    }
  }
  return FVO_OK;
}

/* One line of xtp_u (sw_core.F90:2154-2521) / ytp_v (:2524-2998) on the cubed sphere (grid_type < 3, not bounded):
 * w[i] = the wind along the line (u(:, j) resp. v(i, :)), dx[i] / rdx[i] = its metric (dx(:, j) resp. dy(i, :)),
 * c[i], flux[i] on [is, ie+1]; npx = the edge index of the line's direction; edge_row = the line itself runs along a
 * face edge (j == 1 or npy for xtp_u, i == 1 or npx for ytp_v: there the first / last two parabolas are flat). */
static void tp_wind_line_cs(const double *w, const double *dx, const double *rdx, const double *c, double *flux, int is,
                            int ie, int iord, int npx, int edge_row, double lim_fac) {
  const int lo = is - 3, n = ie - is + 8;
  const int is3 = IMAX(3, is - 1), ie3 = IMIN(npx - 3, ie + 1);
  double *buf = dalloc((size_t)6 * n);
  double *bl = buf - lo, *br = buf + n - lo, *b0 = buf + 2 * n - lo, *al = buf + 3 * n - lo, *dm = buf + 4 * n - lo,
         *dq = buf + 5 * n - lo;
  unsigned char *lb = (unsigned char *)calloc((size_t)2 * n, 1);
  unsigned char *smt5 = lb - lo, *smt6 = lb + n - lo;
  double cfl, fx0, x0, x1, xt, x0L, x0R;
  int i;
  if (iord < 8) {
    for (i = is3; i <= ie3 + 1; i++) al[i] = p1 * (w[i - 1] + w[i]) + p2 * (w[i - 2] + w[i + 1]);
    for (i = is3; i <= ie3; i++) {
      bl[i] = al[i] - w[i];
      br[i] = al[i + 1] - w[i];
    }
    if (is == 1) { /* :2200-2220 */
      xt = c3 * w[1] + c2 * w[2] + c1 * w[3];
      br[1] = xt - w[1];
      bl[2] = xt - w[2];
      br[2] = al[3] - w[2];
      if (edge_row) {
        bl[0] = 0.; br[0] = 0.; bl[1] = 0.; br[1] = 0.;
      } else {
        bl[0] = c1 * w[-2] + c2 * w[-1] + c3 * w[0] - w[0];
        xt = 0.5 * (((2. * dx[0] + dx[-1]) * (w[0]) - dx[0] * w[-1]) / (dx[0] + dx[-1]) +
                    ((2. * dx[1] + dx[2]) * (w[1]) - dx[1] * w[2]) / (dx[1] + dx[2]));
        br[0] = xt - w[0];
        bl[1] = xt - w[1];
      }
    }
    if ((ie + 1) == npx) { /* :2222-2242 */
      bl[npx - 2] = al[npx - 2] - w[npx - 2];
      xt = c1 * w[npx - 3] + c2 * w[npx - 2] + c3 * w[npx - 1];
      br[npx - 2] = xt - w[npx - 2];
      bl[npx - 1] = xt - w[npx - 1];
      if (edge_row) {
        bl[npx - 1] = 0.; br[npx - 1] = 0.; bl[npx] = 0.; br[npx] = 0.;
      } else {
        xt = 0.5 * (((2. * dx[npx - 1] + dx[npx - 2]) * w[npx - 1] - dx[npx - 1] * w[npx - 2]) / (dx[npx - 1] + dx[npx - 2]) +
                    ((2. * dx[npx] + dx[npx + 1]) * w[npx] - dx[npx] * w[npx + 1]) / (dx[npx] + dx[npx + 1]));
        br[npx - 1] = xt - w[npx - 1];
        bl[npx] = xt - w[npx];
        br[npx] = c3 * w[npx] + c2 * w[npx + 1] + c1 * w[npx + 2] - w[npx];
      }
    }
    for (i = is - 1; i <= ie + 1; i++) b0[i] = bl[i] + br[i];
    if (iord == 1 || iord == 4 || iord >= 5) {
      if (iord == 1) {
        for (i = is - 1; i <= ie + 1; i++) smt5[i] = fabs(lim_fac * b0[i]) < fabs(bl[i] - br[i]);
      } else if (iord == 4) {
        for (i = is - 1; i <= ie + 1; i++) {
          x0 = fabs(b0[i]);
          x1 = fabs(bl[i] - br[i]);
          smt5[i] = x0 < x1;
          smt6[i] = 3. * x0 < x1;
        }
      } else if (iord == 5) {
        for (i = is - 1; i <= ie + 1; i++) smt5[i] = bl[i] * br[i] < 0.;
      } else {
        for (i = is - 1; i <= ie + 1; i++) smt5[i] = 3. * fabs(b0[i]) < fabs(bl[i] - br[i]);
        if (is == 1) { /* fix edge issues, :2343-2352 */
          smt5[0] = bl[0] * br[0] < 0.;
          smt5[1] = bl[1] * br[1] < 0.;
        }
        if ((ie + 1) == npx) {
          smt5[npx - 1] = bl[npx - 1] * br[npx - 1] < 0.;
          smt5[npx] = bl[npx] * br[npx] < 0.;
        }
      }
      for (i = is; i <= ie + 1; i++) {
        int on;
        if (c[i] > 0.) {
          cfl = c[i] * rdx[i - 1];
          fx0 = (1. - cfl) * (br[i - 1] - cfl * b0[i - 1]);
          flux[i] = w[i - 1];
        } else {
          cfl = c[i] * rdx[i];
          fx0 = (1. + cfl) * (bl[i] + cfl * b0[i]);
          flux[i] = w[i];
        }
        if (iord == 4) {
          int hi5 = smt5[i - 1] && smt5[i];
          int hi6 = smt6[i - 1] || smt6[i];
          on = hi5 || hi6;
        } else {
          on = smt5[i - 1] || smt5[i];
        }
        if (on) flux[i] = flux[i] + fx0;
      }
    } else if (iord == 2) {
      for (i = is; i <= ie + 1; i++) {
        if (c[i] > 0.) {
          cfl = c[i] * rdx[i - 1];
          flux[i] = w[i - 1] + (1. - cfl) * (br[i - 1] - cfl * b0[i - 1]);
        } else {
          cfl = c[i] * rdx[i];
          flux[i] = w[i] + (1. + cfl) * (bl[i] + cfl * b0[i]);
        }
      }
    } else { /* iord == 3 */
      for (i = is - 1; i <= ie + 1; i++) {
        x0 = fabs(b0[i]);
        x1 = fabs(bl[i] - br[i]);
        smt5[i] = x0 < x1;
        smt6[i] = 3. * x0 < x1;
      }
      for (i = is; i <= ie + 1; i++) {
        int hi5 = smt5[i - 1] && smt5[i];
        int hi6 = smt6[i - 1] || smt6[i];
        fx0 = 0.;
        if (c[i] > 0.) {
          cfl = c[i] * rdx[i - 1];
          if (hi6)
            fx0 = br[i - 1] - cfl * b0[i - 1];
          else if (hi5)
            fx0 = fsign(dmin(fabs(bl[i - 1]), fabs(br[i - 1])), br[i - 1]);
          flux[i] = w[i - 1] + (1. - cfl) * fx0;
        } else {
          cfl = c[i] * rdx[i];
          if (hi6)
            fx0 = bl[i] + cfl * b0[i];
          else if (hi5)
            fx0 = fsign(dmin(fabs(bl[i]), fabs(br[i])), bl[i]);
          flux[i] = w[i] + (1. + cfl) * fx0;
        }
      }
    }
  } else { /* iord = 8, 9, 10, 11 on the cubed sphere, :2381-2490 */
    double pmp_1, lac_1, pmp_2, lac_2;
    for (i = is - 2; i <= ie + 2; i++) {
      xt = 0.25 * (w[i + 1] - w[i - 1]);
      dm[i] = fsign(dmin3(fabs(xt), dmax3(w[i - 1], w[i], w[i + 1]) - w[i], w[i] - dmin3(w[i - 1], w[i], w[i + 1])), xt);
    }
    for (i = is - 3; i <= ie + 2; i++) dq[i] = w[i + 1] - w[i];
    for (i = is3; i <= ie3 + 1; i++) al[i] = 0.5 * (w[i - 1] + w[i]) + r3 * (dm[i - 1] - dm[i]);
    if (iord == 8) {
      for (i = is3; i <= ie3; i++) {
        xt = 2. * dm[i];
        bl[i] = -fsign(dmin(fabs(xt), fabs(al[i] - w[i])), xt);
        br[i] = fsign(dmin(fabs(xt), fabs(al[i + 1] - w[i])), xt);
      }
    } else if (iord == 9) {
      for (i = is3; i <= ie3; i++) {
        pmp_1 = -2. * dq[i];
        lac_1 = pmp_1 + 1.5 * dq[i + 1];
        bl[i] = dmin(dmax3(0., pmp_1, lac_1), dmax(al[i] - w[i], dmin3(0., pmp_1, lac_1)));
        pmp_2 = 2. * dq[i - 1];
        lac_2 = pmp_2 - 1.5 * dq[i - 2];
        br[i] = dmin(dmax3(0., pmp_2, lac_2), dmax(al[i + 1] - w[i], dmin3(0., pmp_2, lac_2)));
      }
    } else if (iord == 10) {
      for (i = is3; i <= ie3; i++) {
        bl[i] = al[i] - w[i];
        br[i] = al[i + 1] - w[i];
        if (fabs(dm[i]) < near_zero) {
          if (fabs(dm[i - 1]) + fabs(dm[i + 1]) < near_zero) { /* 2-delta-x structure detected within 3 cells */
            bl[i] = 0.;
            br[i] = 0.;
          }
        } else if (fabs(3. * (bl[i] + br[i])) > fabs(bl[i] - br[i])) {
          pmp_1 = -2. * dq[i];
          lac_1 = pmp_1 + 1.5 * dq[i + 1];
          bl[i] = dmin(dmax3(0., pmp_1, lac_1), dmax(bl[i], dmin3(0., pmp_1, lac_1)));
          pmp_2 = 2. * dq[i - 1];
          lac_2 = pmp_2 - 1.5 * dq[i - 2];
          br[i] = dmin(dmax3(0., pmp_2, lac_2), dmax(br[i], dmin3(0., pmp_2, lac_2)));
        }
      }
    } else {
      for (i = is3; i <= ie3; i++) {
        bl[i] = al[i] - w[i];
        br[i] = al[i + 1] - w[i];
      }
    }
    if (is == 1) { /* fix the edges, :2437-2460 */
      br[2] = al[3] - w[2];
      xt = s15 * w[1] + s11 * w[2] - s14 * dm[2];
      bl[2] = xt - w[2];
      br[1] = xt - w[1];
      if (edge_row) {
        bl[0] = 0.; br[0] = 0.; bl[1] = 0.; br[1] = 0.;
      } else {
        bl[0] = s14 * dm[-1] - s11 * dq[-1];
        x0L = 0.5 * ((2. * dx[0] + dx[-1]) * (w[0]) - dx[0] * (w[-1])) / (dx[0] + dx[-1]);
        x0R = 0.5 * ((2. * dx[1] + dx[2]) * (w[1]) - dx[1] * (w[2])) / (dx[1] + dx[2]);
        xt = x0L + x0R;
        br[0] = xt - w[0];
        bl[1] = xt - w[1];
      }
      fvo_pert_ppm(1, w + 2, bl + 2, br + 2, -1);
    }
    if ((ie + 1) == npx) { /* :2461-2484 */
      bl[npx - 2] = al[npx - 2] - w[npx - 2];
      xt = s15 * w[npx - 1] + s11 * w[npx - 2] + s14 * dm[npx - 2];
      br[npx - 2] = xt - w[npx - 2];
      bl[npx - 1] = xt - w[npx - 1];
      if (edge_row) {
        bl[npx - 1] = 0.; br[npx - 1] = 0.; bl[npx] = 0.; br[npx] = 0.;
      } else {
        br[npx] = s11 * dq[npx] - s14 * dm[npx + 1];
        x0L = 0.5 * ((2. * dx[npx - 1] + dx[npx - 2]) * (w[npx - 1]) - dx[npx - 1] * (w[npx - 2])) / (dx[npx - 1] + dx[npx - 2]);
        x0R = 0.5 * ((2. * dx[npx] + dx[npx + 1]) * (w[npx]) - dx[npx] * (w[npx + 1])) / (dx[npx] + dx[npx + 1]);
        xt = x0L + x0R;
        br[npx - 1] = xt - w[npx - 1];
        bl[npx] = xt - w[npx];
      }
      fvo_pert_ppm(1, w + npx - 2, bl + npx - 2, br + npx - 2, -1);
    }
    for (i = is; i <= ie + 1; i++) {
      if (c[i] > 0.) {
        cfl = c[i] * rdx[i - 1];
        flux[i] = w[i - 1] + (1. - cfl) * (br[i - 1] - cfl * (bl[i - 1] + br[i - 1]));
      } else {
        cfl = c[i] * rdx[i];
        flux[i] = w[i] + (1. + cfl) * (bl[i] + cfl * (bl[i] + br[i]));
      }
    }
  }
  free(buf);
  free(lb);
}

/* xtp_u, sw_core.F90:2154-2521: branch "bounded_domain .or. grid_type>3" (is3=is-1, ie3=ie+1).
 * c, flux: (is:ie+1, js:je+1). */
int fvo_xtp_u(const fvo_grid *g, const double *c, const double *u, const double *v, double *flux,
              int iord) {
  BOUNDS(g);
  const double *dx = g->dx, *rdx = g->rdx; /* U shape */
  const double lim_fac = g->lim_fac;
  int i, j;
  (void)v;
  if (g->grid_type == 3) return FVO_ERR_UNSUPPORTED;
  if (g->grid_type < 3) { /* cubed sphere: one line per j */
    for (j = js; j <= je + 1; j++)
      tp_wind_line_cs(u + IU(0, j), dx + IU(0, j), rdx + IU(0, j), c + IBC(0, j), flux + IBC(0, j), is, ie, iord, g->npx,
                      j == 1 || j == g->npy, lim_fac);
    return FVO_OK;
  }
  const int is3 = is - 1, ie3 = ie + 1;
  const int lo = is - 3, n = nx + 8;
  double *buf = dalloc((size_t)6 * n);
  double *bl = buf - lo, *br = buf + n - lo, *b0 = buf + 2 * n - lo, *al = buf + 3 * n - lo,
         *dm = buf + 4 * n - lo, *dq = buf + 5 * n - lo;
  unsigned char *lb = (unsigned char *)calloc((size_t)2 * n, 1);
  unsigned char *smt5 = lb - lo, *smt6 = lb + n - lo;
  double cfl, fx0, x0, x1, xt, pmp, lac;

  if (iord < 8) { /* :2187-2377 */
    for (j = js; j <= je + 1; j++) {
      for (i = is3; i <= ie3 + 1; i++)
        al[i] = p1 * (u[IU(i - 1, j)] + u[IU(i, j)]) + p2 * (u[IU(i - 2, j)] + u[IU(i + 1, j)]);
      for (i = is3; i <= ie3; i++) {
        bl[i] = al[i] - u[IU(i, j)];
        br[i] = al[i + 1] - u[IU(i, j)];
      }
      for (i = is - 1; i <= ie + 1; i++) b0[i] = bl[i] + br[i];
      if (iord == 1) {
        for (i = is - 1; i <= ie + 1; i++) smt5[i] = fabs(lim_fac * b0[i]) < fabs(bl[i] - br[i]);
        for (i = is; i <= ie + 1; i++) {
          if (c[IBC(i, j)] > 0.) {
            cfl = c[IBC(i, j)] * rdx[IU(i - 1, j)];
            fx0 = (1. - cfl) * (br[i - 1] - cfl * b0[i - 1]);
            flux[IBC(i, j)] = u[IU(i - 1, j)];
          } else {
            cfl = c[IBC(i, j)] * rdx[IU(i, j)];
            fx0 = (1. + cfl) * (bl[i] + cfl * b0[i]);
            flux[IBC(i, j)] = u[IU(i, j)];
          }
          if (smt5[i - 1] || smt5[i]) flux[IBC(i, j)] = flux[IBC(i, j)] + fx0;
        }
      } else if (iord == 2) {
        for (i = is; i <= ie + 1; i++) {
          if (c[IBC(i, j)] > 0.) {
            cfl = c[IBC(i, j)] * rdx[IU(i - 1, j)];
            flux[IBC(i, j)] = u[IU(i - 1, j)] + (1. - cfl) * (br[i - 1] - cfl * b0[i - 1]);
          } else {
            cfl = c[IBC(i, j)] * rdx[IU(i, j)];
            flux[IBC(i, j)] = u[IU(i, j)] + (1. + cfl) * (bl[i] + cfl * b0[i]);
          }
        }
      } else if (iord == 3) {
        for (i = is - 1; i <= ie + 1; i++) {
          x0 = fabs(b0[i]);
          x1 = fabs(bl[i] - br[i]);
          smt5[i] = x0 < x1;
          smt6[i] = 3. * x0 < x1;
        }
        for (i = is; i <= ie + 1; i++) {
          int hi5 = smt5[i - 1] && smt5[i];
          int hi6 = smt6[i - 1] || smt6[i];
          fx0 = 0.;
          if (c[IBC(i, j)] > 0.) {
            cfl = c[IBC(i, j)] * rdx[IU(i - 1, j)];
            if (hi6)
              fx0 = br[i - 1] - cfl * b0[i - 1];
            else if (hi5)
              fx0 = fsign(dmin(fabs(bl[i - 1]), fabs(br[i - 1])), br[i - 1]);
            flux[IBC(i, j)] = u[IU(i - 1, j)] + (1. - cfl) * fx0;
          } else {
            cfl = c[IBC(i, j)] * rdx[IU(i, j)];
            if (hi6)
              fx0 = bl[i] + cfl * b0[i];
            else if (hi5)
              fx0 = fsign(dmin(fabs(bl[i]), fabs(br[i])), bl[i]);
            flux[IBC(i, j)] = u[IU(i, j)] + (1. + cfl) * fx0;
          }
        }
      } else if (iord == 4) {
        for (i = is - 1; i <= ie + 1; i++) {
          x0 = fabs(b0[i]);
          x1 = fabs(bl[i] - br[i]);
          smt5[i] = x0 < x1;
          smt6[i] = 3. * x0 < x1;
        }
        for (i = is; i <= ie + 1; i++) {
          int hi5 = smt5[i - 1] && smt5[i];
          int hi6 = smt6[i - 1] || smt6[i];
          hi5 = hi5 || hi6;
          if (c[IBC(i, j)] > 0.) {
            cfl = c[IBC(i, j)] * rdx[IU(i - 1, j)];
            fx0 = (1. - cfl) * (br[i - 1] - cfl * b0[i - 1]);
            flux[IBC(i, j)] = u[IU(i - 1, j)];
          } else {
            cfl = c[IBC(i, j)] * rdx[IU(i, j)];
            fx0 = (1. + cfl) * (bl[i] + cfl * b0[i]);
            flux[IBC(i, j)] = u[IU(i, j)];
          }
          if (hi5) flux[IBC(i, j)] = flux[IBC(i, j)] + fx0;
        }
      } else { /* iord = 5,6,7 */
        if (iord == 5) {
          for (i = is - 1; i <= ie + 1; i++) smt5[i] = bl[i] * br[i] < 0.;
        } else {
          for (i = is - 1; i <= ie + 1; i++) smt5[i] = 3. * fabs(b0[i]) < fabs(bl[i] - br[i]);
        }
        for (i = is; i <= ie + 1; i++) {
          if (c[IBC(i, j)] > 0.) {
            cfl = c[IBC(i, j)] * rdx[IU(i - 1, j)];
            fx0 = (1. - cfl) * (br[i - 1] - cfl * b0[i - 1]);
            flux[IBC(i, j)] = u[IU(i - 1, j)];
          } else {
            cfl = c[IBC(i, j)] * rdx[IU(i, j)];
            fx0 = (1. + cfl) * (bl[i] + cfl * b0[i]);
            flux[IBC(i, j)] = u[IU(i, j)];
          }
          if (smt5[i - 1] || smt5[i]) flux[IBC(i, j)] = flux[IBC(i, j)] + fx0;
        }
      }
    }
  } else { /* iord = 8..11, "Other grids" branch :2492-2506 */
    for (j = js; j <= je + 1; j++) {
      for (i = is - 2; i <= ie + 2; i++) {
        xt = 0.25 * (u[IU(i + 1, j)] - u[IU(i - 1, j)]);
        dm[i] = fsign(dmin3(fabs(xt), dmax3(u[IU(i - 1, j)], u[IU(i, j)], u[IU(i + 1, j)]) - u[IU(i, j)],
                            u[IU(i, j)] - dmin3(u[IU(i - 1, j)], u[IU(i, j)], u[IU(i + 1, j)])),
                      xt);
      }
      for (i = is - 3; i <= ie + 2; i++) dq[i] = u[IU(i + 1, j)] - u[IU(i, j)];
      for (i = is - 1; i <= ie + 2; i++) al[i] = 0.5 * (u[IU(i - 1, j)] + u[IU(i, j)]) + r3 * (dm[i - 1] - dm[i]);
      for (i = is - 1; i <= ie + 1; i++) {
        pmp = -2. * dq[i];
        lac = pmp + 1.5 * dq[i + 1];
        bl[i] = dmin(dmax3(0., pmp, lac), dmax(al[i] - u[IU(i, j)], dmin3(0., pmp, lac)));
        pmp = 2. * dq[i - 1];
        lac = pmp - 1.5 * dq[i - 2];
        br[i] = dmin(dmax3(0., pmp, lac), dmax(al[i + 1] - u[IU(i, j)], dmin3(0., pmp, lac)));
      }
      for (i = is; i <= ie + 1; i++) {
        if (c[IBC(i, j)] > 0.) {
          cfl = c[IBC(i, j)] * rdx[IU(i - 1, j)];
          flux[IBC(i, j)] = u[IU(i - 1, j)] + (1. - cfl) * (br[i - 1] - cfl * (bl[i - 1] + br[i - 1]));
        } else {
          cfl = c[IBC(i, j)] * rdx[IU(i, j)];
          flux[IBC(i, j)] = u[IU(i, j)] + (1. + cfl) * (bl[i] + cfl * (bl[i] + br[i]));
        }
      }
    }
  }
  free(buf);
  free(lb);
  return FVO_OK;
}

/* ytp_v, sw_core.F90:2524-2998, same branch as xtp_u (js3=js-1, je3=je+1). */
int fvo_ytp_v(const fvo_grid *g, const double *c, const double *u, const double *v, double *flux,
              int jord) {
  BOUNDS(g);
  const double *rdy = g->rdy; /* V shape */
  const double lim_fac = g->lim_fac;
  int i, j;
  (void)u;
  if (g->grid_type == 3) return FVO_ERR_UNSUPPORTED;
  if (g->grid_type < 3) { /* cubed sphere: one line per i (columns gathered into line buffers) */
    const int nl = njd + 2;
    double *line = dalloc((size_t)5 * nl);
    double *wl = line - jsd, *dl = line + nl - jsd, *rl = line + 2 * nl - jsd, *cl = line + 3 * nl - jsd,
           *fl = line + 4 * nl - jsd;
    for (i = is; i <= ie + 1; i++) {
      for (j = jsd; j <= jed; j++) {
        wl[j] = v[IV(i, j)];
        dl[j] = g->dy[IV(i, j)];
        rl[j] = rdy[IV(i, j)];
      }
      for (j = js; j <= je + 1; j++) cl[j] = c[IBC(i, j)];
      tp_wind_line_cs(wl, dl, rl, cl, fl, js, je, jord, g->npy, i == 1 || i == g->npx, lim_fac);
      for (j = js; j <= je + 1; j++) flux[IBC(i, j)] = fl[j];
    }
    free(line);
    return FVO_OK;
  }
  const int js3 = js - 1, je3 = je + 1;
  /* work arrays (is:ie+1, js-3:je+3) */
  const int nw = nx + 1, mh = ny + 8;
#define W(a, i, j) a[(size_t)((j) - (js - 3)) * nw + ((i)-is)]
  double *bl = dalloc((size_t)nw * mh), *br = dalloc((size_t)nw * mh), *b0 = dalloc((size_t)nw * mh);
  double *al = dalloc((size_t)nw * mh), *dm = dalloc((size_t)nw * mh), *dq = dalloc((size_t)nw * mh);
  unsigned char *smt5 = (unsigned char *)calloc((size_t)nw * mh, 1);
  unsigned char *smt6 = (unsigned char *)calloc((size_t)nw * mh, 1);
  double cfl, fx0, x0, x1, xt, pmp, lac;

  if (jord < 8) {
    for (j = js3; j <= je3 + 1; j++)
      for (i = is; i <= ie + 1; i++)
        W(al, i, j) = p1 * (v[IV(i, j - 1)] + v[IV(i, j)]) + p2 * (v[IV(i, j - 2)] + v[IV(i, j + 1)]);
    for (j = js3; j <= je3; j++)
      for (i = is; i <= ie + 1; i++) {
        W(bl, i, j) = W(al, i, j) - v[IV(i, j)];
        W(br, i, j) = W(al, i, j + 1) - v[IV(i, j)];
      }
    for (j = js - 1; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) W(b0, i, j) = W(bl, i, j) + W(br, i, j);

    if (jord == 1) {
      for (j = js - 1; j <= je + 1; j++)
        for (i = is; i <= ie + 1; i++)
          W(smt5, i, j) = fabs(lim_fac * W(b0, i, j)) < fabs(W(bl, i, j) - W(br, i, j));
      for (j = js; j <= je + 1; j++)
        for (i = is; i <= ie + 1; i++) {
          if (c[IBC(i, j)] > 0.) {
            cfl = c[IBC(i, j)] * rdy[IV(i, j - 1)];
            fx0 = (1. - cfl) * (W(br, i, j - 1) - cfl * W(b0, i, j - 1));
            flux[IBC(i, j)] = v[IV(i, j - 1)];
          } else {
            cfl = c[IBC(i, j)] * rdy[IV(i, j)];
            fx0 = (1. + cfl) * (W(bl, i, j) + cfl * W(b0, i, j));
            flux[IBC(i, j)] = v[IV(i, j)];
          }
          if (W(smt5, i, j - 1) || W(smt5, i, j)) flux[IBC(i, j)] = flux[IBC(i, j)] + fx0;
        }
    } else if (jord == 2) {
      for (j = js; j <= je + 1; j++)
        for (i = is; i <= ie + 1; i++) {
          if (c[IBC(i, j)] > 0.) {
            cfl = c[IBC(i, j)] * rdy[IV(i, j - 1)];
            flux[IBC(i, j)] = v[IV(i, j - 1)] + (1. - cfl) * (W(br, i, j - 1) - cfl * W(b0, i, j - 1));
          } else {
            cfl = c[IBC(i, j)] * rdy[IV(i, j)];
            flux[IBC(i, j)] = v[IV(i, j)] + (1. + cfl) * (W(bl, i, j) + cfl * W(b0, i, j));
          }
        }
    } else if (jord == 3) {
      for (j = js - 1; j <= je + 1; j++)
        for (i = is; i <= ie + 1; i++) {
          x0 = fabs(W(b0, i, j));
          x1 = fabs(W(bl, i, j) - W(br, i, j));
          W(smt5, i, j) = x0 < x1;
          W(smt6, i, j) = 3. * x0 < x1;
        }
      for (j = js; j <= je + 1; j++)
        for (i = is; i <= ie + 1; i++) {
          int hi5 = W(smt5, i, j - 1) && W(smt5, i, j);
          int hi6 = W(smt6, i, j - 1) || W(smt6, i, j);
          fx0 = 0.;
          if (c[IBC(i, j)] > 0.) {
            cfl = c[IBC(i, j)] * rdy[IV(i, j - 1)];
            if (hi6)
              fx0 = W(br, i, j - 1) - cfl * W(b0, i, j - 1);
            else if (hi5)
              fx0 = fsign(dmin(fabs(W(bl, i, j - 1)), fabs(W(br, i, j - 1))), W(br, i, j - 1));
            flux[IBC(i, j)] = v[IV(i, j - 1)] + (1. - cfl) * fx0;
          } else {
            cfl = c[IBC(i, j)] * rdy[IV(i, j)];
            if (hi6)
              fx0 = W(bl, i, j) + cfl * W(b0, i, j);
            else if (hi5)
              fx0 = fsign(dmin(fabs(W(bl, i, j)), fabs(W(br, i, j))), W(bl, i, j));
            flux[IBC(i, j)] = v[IV(i, j)] + (1. + cfl) * fx0;
          }
        }
    } else if (jord == 4) {
      for (j = js - 1; j <= je + 1; j++)
        for (i = is; i <= ie + 1; i++) {
          x0 = fabs(W(b0, i, j));
          x1 = fabs(W(bl, i, j) - W(br, i, j));
          W(smt5, i, j) = x0 < x1;
          W(smt6, i, j) = 3. * x0 < x1;
        }
      for (j = js; j <= je + 1; j++)
        for (i = is; i <= ie + 1; i++) {
          int hi5 = W(smt5, i, j - 1) && W(smt5, i, j);
          int hi6 = W(smt6, i, j - 1) || W(smt6, i, j);
          hi5 = hi5 || hi6;
          if (c[IBC(i, j)] > 0.) {
            cfl = c[IBC(i, j)] * rdy[IV(i, j - 1)];
            fx0 = (1. - cfl) * (W(br, i, j - 1) - cfl * W(b0, i, j - 1));
            flux[IBC(i, j)] = v[IV(i, j - 1)];
          } else {
            cfl = c[IBC(i, j)] * rdy[IV(i, j)];
            fx0 = (1. + cfl) * (W(bl, i, j) + cfl * W(b0, i, j));
            flux[IBC(i, j)] = v[IV(i, j)];
          }
          if (hi5) flux[IBC(i, j)] = flux[IBC(i, j)] + fx0;
        }
    } else { /* jord = 5,6,7 */
      if (jord == 5) {
        for (j = js - 1; j <= je + 1; j++)
          for (i = is; i <= ie + 1; i++) W(smt5, i, j) = W(bl, i, j) * W(br, i, j) < 0.;
      } else {
        for (j = js - 1; j <= je + 1; j++)
          for (i = is; i <= ie + 1; i++)
            W(smt5, i, j) = 3. * fabs(W(b0, i, j)) < fabs(W(bl, i, j) - W(br, i, j));
      }
      for (j = js; j <= je + 1; j++)
        for (i = is; i <= ie + 1; i++) {
          if (c[IBC(i, j)] > 0.) {
            cfl = c[IBC(i, j)] * rdy[IV(i, j - 1)];
            fx0 = (1. - cfl) * (W(br, i, j - 1) - cfl * W(b0, i, j - 1));
            flux[IBC(i, j)] = v[IV(i, j - 1)];
          } else {
            cfl = c[IBC(i, j)] * rdy[IV(i, j)];
            fx0 = (1. + cfl) * (W(bl, i, j) + cfl * W(b0, i, j));
            flux[IBC(i, j)] = v[IV(i, j)];
          }
          if (W(smt5, i, j - 1) || W(smt5, i, j)) flux[IBC(i, j)] = flux[IBC(i, j)] + fx0;
        }
    }
  } else { /* jord = 8..11 */
    for (j = js - 2; j <= je + 2; j++)
      for (i = is; i <= ie + 1; i++) {
        xt = 0.25 * (v[IV(i, j + 1)] - v[IV(i, j - 1)]);
        W(dm, i, j) = fsign(dmin3(fabs(xt), dmax3(v[IV(i, j - 1)], v[IV(i, j)], v[IV(i, j + 1)]) - v[IV(i, j)],
                                  v[IV(i, j)] - dmin3(v[IV(i, j - 1)], v[IV(i, j)], v[IV(i, j + 1)])),
                            xt);
      }
    for (j = js - 3; j <= je + 2; j++)
      for (i = is; i <= ie + 1; i++) W(dq, i, j) = v[IV(i, j + 1)] - v[IV(i, j)];
    /* "other grids" :2973-2990 */
    for (j = js - 1; j <= je + 2; j++)
      for (i = is; i <= ie + 1; i++)
        W(al, i, j) = 0.5 * (v[IV(i, j - 1)] + v[IV(i, j)]) + r3 * (W(dm, i, j - 1) - W(dm, i, j));
    for (j = js - 1; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) {
        pmp = 2. * W(dq, i, j - 1);
        lac = pmp - 1.5 * W(dq, i, j - 2);
        W(br, i, j) = dmin(dmax3(0., pmp, lac), dmax(W(al, i, j + 1) - v[IV(i, j)], dmin3(0., pmp, lac)));
        pmp = -2. * W(dq, i, j);
        lac = pmp + 1.5 * W(dq, i, j + 1);
        W(bl, i, j) = dmin(dmax3(0., pmp, lac), dmax(W(al, i, j) - v[IV(i, j)], dmin3(0., pmp, lac)));
      }
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) {
        if (c[IBC(i, j)] > 0.) {
          cfl = c[IBC(i, j)] * rdy[IV(i, j - 1)];
          flux[IBC(i, j)] = v[IV(i, j - 1)] + (1. - cfl) * (W(br, i, j - 1) - cfl * (W(bl, i, j - 1) + W(br, i, j - 1)));
        } else {
          cfl = c[IBC(i, j)] * rdy[IV(i, j)];
          flux[IBC(i, j)] = v[IV(i, j)] + (1. + cfl) * (W(bl, i, j) + cfl * (W(bl, i, j) + W(br, i, j)));
        }
      }
  }
#undef W
  free(bl);
  free(br);
  free(b0);
  free(al);
  free(dm);
  free(dq);
  free(smt5);
  free(smt6);
  return FVO_OK;
}

/* a2b_ord4, a2b_edge.F90:47-327, branch grid_type>=3 (:292-315). qin, qout on the A shape. */
/* a2b_ord4 on the cubed sphere, a2b_edge.F90:83-290 (one tile per face: all four edges and corners).  The corner values
 * use extrap_corner (:452-462): q1 + x1/(x2-x1) (q1-q2) with the great-circle distances x1, x2 of the two cell centres
 * from the corner; the factors x1/(x2-x1) are geometry (g->corner_f[corner][pair], corners sw, se, ne, nw). */
static int a2b_ord4_cubed(const fvo_grid *g, double *qin, double *qout, int replace) {
  BOUNDS(g);
  static const double b1 = 7. / 12., b2 = -1. / 12., cc1 = 2. / 3., cc2 = -1. / 6.;
  const int npx = g->npx, npy = g->npy;
  const double *dxa = g->dxa, *dya = g->dya;
  const double *edge_w = g->edge_w - 1, *edge_e = g->edge_e - 1, *edge_s = g->edge_s - 1, *edge_n = g->edge_n - 1; /* 1-based */
  const int is1 = IMAX(1, is - 1), js1 = IMAX(1, js - 1), is2 = IMAX(2, is), js2 = IMAX(2, js);
  const int ie1 = IMIN(npx - 1, ie + 1), je1 = IMIN(npy - 1, je + 1);
  int i, j;
  double g_in, g_ou;
  double *qx = dalloc((size_t)nid * njd), *qy = dalloc((size_t)nid * njd), *qxx = dalloc((size_t)nid * njd),
         *qyy = dalloc((size_t)nid * njd);
  double *q1 = dalloc((size_t)nid + 4) - isd + 1, *q2 = dalloc((size_t)njd + 4) - jsd + 1;
#define QI(i, j) qin[IA(i, j)]
#define EXTRAP(f, a, b) ((a) + (f) * ((a) - (b)))
  if (g->sw_corner)
    qout[IA(1, 1)] = (EXTRAP(g->corner_f[0], QI(1, 1), QI(2, 2)) + EXTRAP(g->corner_f[1], QI(0, 1), QI(-1, 2)) +
                      EXTRAP(g->corner_f[2], QI(1, 0), QI(2, -1))) * r3;
  if (g->se_corner)
    qout[IA(npx, 1)] = (EXTRAP(g->corner_f[3], QI(npx - 1, 1), QI(npx - 2, 2)) + EXTRAP(g->corner_f[4], QI(npx - 1, 0), QI(npx - 2, -1)) +
                        EXTRAP(g->corner_f[5], QI(npx, 1), QI(npx + 1, 2))) * r3;
  if (g->ne_corner)
    qout[IA(npx, npy)] = (EXTRAP(g->corner_f[6], QI(npx - 1, npy - 1), QI(npx - 2, npy - 2)) +
                          EXTRAP(g->corner_f[7], QI(npx, npy - 1), QI(npx + 1, npy - 2)) +
                          EXTRAP(g->corner_f[8], QI(npx - 1, npy), QI(npx - 2, npy + 1))) * r3;
  if (g->nw_corner)
    qout[IA(1, npy)] = (EXTRAP(g->corner_f[9], QI(1, npy - 1), QI(2, npy - 2)) + EXTRAP(g->corner_f[10], QI(0, npy - 1), QI(-1, npy - 2)) +
                        EXTRAP(g->corner_f[11], QI(1, npy), QI(2, npy + 1))) * r3;
  /* X-sweep, :133-173 */
  for (j = IMAX(1, js - 2); j <= IMIN(npy - 1, je + 2); j++)
    for (i = IMAX(3, is); i <= IMIN(npx - 2, ie + 1); i++)
      qx[IA(i, j)] = b2 * (QI(i - 2, j) + QI(i + 1, j)) + b1 * (QI(i - 1, j) + QI(i, j));
  if (is == 1) {
    for (j = js1; j <= je1; j++) q2[j] = (QI(0, j) * dxa[IA(1, j)] + QI(1, j) * dxa[IA(0, j)]) / (dxa[IA(0, j)] + dxa[IA(1, j)]);
    for (j = js2; j <= je1; j++) qout[IA(1, j)] = edge_w[j] * q2[j - 1] + (1. - edge_w[j]) * q2[j];
    for (j = IMAX(1, js - 2); j <= IMIN(npy - 1, je + 2); j++) {
      g_in = dxa[IA(2, j)] / dxa[IA(1, j)];
      g_ou = dxa[IA(-1, j)] / dxa[IA(0, j)];
      qx[IA(1, j)] = 0.5 * (((2. + g_in) * QI(1, j) - QI(2, j)) / (1. + g_in) + ((2. + g_ou) * QI(0, j) - QI(-1, j)) / (1. + g_ou));
      qx[IA(2, j)] = (3. * (g_in * QI(1, j) + QI(2, j)) - (g_in * qx[IA(1, j)] + qx[IA(3, j)])) / (2. + 2. * g_in);
    }
  }
  if ((ie + 1) == npx) {
    for (j = js1; j <= je1; j++)
      q2[j] = (QI(npx - 1, j) * dxa[IA(npx, j)] + QI(npx, j) * dxa[IA(npx - 1, j)]) / (dxa[IA(npx - 1, j)] + dxa[IA(npx, j)]);
    for (j = js2; j <= je1; j++) qout[IA(npx, j)] = edge_e[j] * q2[j - 1] + (1. - edge_e[j]) * q2[j];
    for (j = IMAX(1, js - 2); j <= IMIN(npy - 1, je + 2); j++) {
      g_in = dxa[IA(npx - 2, j)] / dxa[IA(npx - 1, j)];
      g_ou = dxa[IA(npx + 1, j)] / dxa[IA(npx, j)];
      qx[IA(npx, j)] = 0.5 * (((2. + g_in) * QI(npx - 1, j) - QI(npx - 2, j)) / (1. + g_in) +
                              ((2. + g_ou) * QI(npx, j) - QI(npx + 1, j)) / (1. + g_ou));
      qx[IA(npx - 1, j)] = (3. * (QI(npx - 2, j) + g_in * QI(npx - 1, j)) - (g_in * qx[IA(npx, j)] + qx[IA(npx - 2, j)])) / (2. + 2. * g_in);
    }
  }
  /* Y-sweep, :183-222 */
  for (j = IMAX(3, js); j <= IMIN(npy - 2, je + 1); j++)
    for (i = IMAX(1, is - 2); i <= IMIN(npx - 1, ie + 2); i++)
      qy[IA(i, j)] = b2 * (QI(i, j - 2) + QI(i, j + 1)) + b1 * (QI(i, j - 1) + QI(i, j));
  if (js == 1) {
    for (i = is1; i <= ie1; i++) q1[i] = (QI(i, 0) * dya[IA(i, 1)] + QI(i, 1) * dya[IA(i, 0)]) / (dya[IA(i, 0)] + dya[IA(i, 1)]);
    for (i = is2; i <= ie1; i++) qout[IA(i, 1)] = edge_s[i] * q1[i - 1] + (1. - edge_s[i]) * q1[i];
    for (i = IMAX(1, is - 2); i <= IMIN(npx - 1, ie + 2); i++) {
      g_in = dya[IA(i, 2)] / dya[IA(i, 1)];
      g_ou = dya[IA(i, -1)] / dya[IA(i, 0)];
      qy[IA(i, 1)] = 0.5 * (((2. + g_in) * QI(i, 1) - QI(i, 2)) / (1. + g_in) + ((2. + g_ou) * QI(i, 0) - QI(i, -1)) / (1. + g_ou));
      qy[IA(i, 2)] = (3. * (g_in * QI(i, 1) + QI(i, 2)) - (g_in * qy[IA(i, 1)] + qy[IA(i, 3)])) / (2. + 2. * g_in);
    }
  }
  if ((je + 1) == npy) {
    for (i = is1; i <= ie1; i++)
      q1[i] = (QI(i, npy - 1) * dya[IA(i, npy)] + QI(i, npy) * dya[IA(i, npy - 1)]) / (dya[IA(i, npy - 1)] + dya[IA(i, npy)]);
    for (i = is2; i <= ie1; i++) qout[IA(i, npy)] = edge_n[i] * q1[i - 1] + (1. - edge_n[i]) * q1[i];
    for (i = IMAX(1, is - 2); i <= IMIN(npx - 1, ie + 2); i++) {
      g_in = dya[IA(i, npy - 2)] / dya[IA(i, npy - 1)];
      g_ou = dya[IA(i, npy + 1)] / dya[IA(i, npy)];
      qy[IA(i, npy)] = 0.5 * (((2. + g_in) * QI(i, npy - 1) - QI(i, npy - 2)) / (1. + g_in) +
                              ((2. + g_ou) * QI(i, npy) - QI(i, npy + 1)) / (1. + g_ou));
      qy[IA(i, npy - 1)] = (3. * (QI(i, npy - 2) + g_in * QI(i, npy - 1)) - (g_in * qy[IA(i, npy)] + qy[IA(i, npy - 2)])) / (2. + 2. * g_in);
    }
  }
  /* :240-276 */
  for (j = IMAX(3, js); j <= IMIN(npy - 2, je + 1); j++)
    for (i = IMAX(2, is); i <= IMIN(npx - 1, ie + 1); i++)
      qxx[IA(i, j)] = a2 * (qx[IA(i, j - 2)] + qx[IA(i, j + 1)]) + a1 * (qx[IA(i, j - 1)] + qx[IA(i, j)]);
  if (js == 1)
    for (i = IMAX(2, is); i <= IMIN(npx - 1, ie + 1); i++)
      qxx[IA(i, 2)] = cc1 * (qx[IA(i, 1)] + qx[IA(i, 2)]) + cc2 * (qout[IA(i, 1)] + qxx[IA(i, 3)]);
  if ((je + 1) == npy)
    for (i = IMAX(2, is); i <= IMIN(npx - 1, ie + 1); i++)
      qxx[IA(i, npy - 1)] = cc1 * (qx[IA(i, npy - 2)] + qx[IA(i, npy - 1)]) + cc2 * (qout[IA(i, npy)] + qxx[IA(i, npy - 2)]);
  for (j = IMAX(2, js); j <= IMIN(npy - 1, je + 1); j++) {
    for (i = IMAX(3, is); i <= IMIN(npx - 2, ie + 1); i++)
      qyy[IA(i, j)] = a2 * (qy[IA(i - 2, j)] + qy[IA(i + 1, j)]) + a1 * (qy[IA(i - 1, j)] + qy[IA(i, j)]);
    if (is == 1) qyy[IA(2, j)] = cc1 * (qy[IA(1, j)] + qy[IA(2, j)]) + cc2 * (qout[IA(1, j)] + qyy[IA(3, j)]);
    if ((ie + 1) == npx)
      qyy[IA(npx - 1, j)] = cc1 * (qy[IA(npx - 2, j)] + qy[IA(npx - 1, j)]) + cc2 * (qout[IA(npx, j)] + qyy[IA(npx - 2, j)]);
    for (i = IMAX(2, is); i <= IMIN(npx - 1, ie + 1); i++) qout[IA(i, j)] = 0.5 * (qxx[IA(i, j)] + qyy[IA(i, j)]);
  }
#undef QI
#undef EXTRAP
  if (replace) {
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) qin[IA(i, j)] = qout[IA(i, j)];
  }
  free(qx);
  free(qy);
  free(qxx);
  free(qyy);
  free(q1 + isd - 1);
  free(q2 + jsd - 1);
  return FVO_OK;
}

int fvo_a2b_ord4(const fvo_grid *g, double *qin, double *qout, int replace) {
  BOUNDS(g);
  static const double b1 = 7. / 12., b2 = -1. / 12.;
  int i, j;
  if (g->grid_type < 3) return a2b_ord4_cubed(g, qin, qout, replace);
  double *qx = dalloc((size_t)nid * njd), *qy = dalloc((size_t)nid * njd);
  for (j = js - 2; j <= je + 2; j++)
    for (i = is; i <= ie + 1; i++)
      qx[IA(i, j)] = b1 * (qin[IA(i - 1, j)] + qin[IA(i, j)]) + b2 * (qin[IA(i - 2, j)] + qin[IA(i + 1, j)]);
  for (j = js; j <= je + 1; j++)
    for (i = is - 2; i <= ie + 2; i++)
      qy[IA(i, j)] = b1 * (qin[IA(i, j - 1)] + qin[IA(i, j)]) + b2 * (qin[IA(i, j - 2)] + qin[IA(i, j + 1)]);
  for (j = js; j <= je + 1; j++)
    for (i = is; i <= ie + 1; i++)
      qout[IA(i, j)] = 0.5 * (a1 * (qx[IA(i, j - 1)] + qx[IA(i, j)] + qy[IA(i - 1, j)] + qy[IA(i, j)]) +
                              a2 * (qx[IA(i, j - 2)] + qx[IA(i, j + 1)] + qy[IA(i - 2, j)] + qy[IA(i + 1, j)]));
  if (replace) {
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) qin[IA(i, j)] = qout[IA(i, j)];
  }
  free(qx);
  free(qy);
  return FVO_OK;
}

/* smag_corner, sw_core.F90:1937-2024 (grid_type==4 only). smag_c on the A shape. */
int fvo_smag_corner(const fvo_grid *g, double dt, const double *u, const double *v, double *smag_c) {
  BOUNDS(g);
  int i, j;
  if (g->grid_type < 3) return FVO_ERR_UNSUPPORTED;
  double *ut = dalloc((size_t)(nid + 1) * njd), *vt = dalloc((size_t)nid * (njd + 1));
  double *wk = dalloc((size_t)nid * njd), *sh = dalloc((size_t)nid * njd);
  for (j = js; j <= je + 1; j++)
    for (i = is - 1; i <= ie + 1; i++) ut[IV(i, j)] = u[IU(i, j)] * g->dyc[IU(i, j)];
  for (j = js - 1; j <= je + 1; j++)
    for (i = is; i <= ie + 1; i++) vt[IU(i, j)] = v[IV(i, j)] * g->dxc[IV(i, j)];
  for (j = js; j <= je + 1; j++)
    for (i = is; i <= ie + 1; i++)
      smag_c[IA(i, j)] = g->rarea_c[IB(i, j)] * (vt[IU(i, j - 1)] - vt[IU(i, j)] - ut[IV(i - 1, j)] + ut[IV(i, j)]);
  for (j = jsd; j <= jed + 1; j++)
    for (i = isd; i <= ied; i++) vt[IU(i, j)] = u[IU(i, j)] * g->dx[IU(i, j)];
  for (j = jsd; j <= jed; j++)
    for (i = isd; i <= ied + 1; i++) ut[IV(i, j)] = v[IV(i, j)] * g->dy[IV(i, j)];
  for (j = jsd; j <= jed; j++)
    for (i = isd; i <= ied; i++)
      wk[IA(i, j)] = g->rarea[IA(i, j)] * (vt[IU(i, j)] - vt[IU(i, j + 1)] + ut[IV(i, j)] - ut[IV(i + 1, j)]);
  fvo_a2b_ord4(g, wk, sh, 0);
  for (j = js; j <= je + 1; j++)
    for (i = is; i <= ie + 1; i++)
      smag_c[IA(i, j)] = dt * sqrt(sh[IA(i, j)] * sh[IA(i, j)] + smag_c[IA(i, j)] * smag_c[IA(i, j)]);
  free(ut);
  free(vt);
  free(wk);
  free(sh);
  return FVO_OK;
}

/* fill_corners (tools/fv_mp_mod.F90:944-1016), BGRID: q on the B layout; dir 1 = XDir, 2 = YDir */
void fvo_fill_corners_b(const fvo_grid *g, double *q, int dir) {
  BOUNDS(g);
  const int npx = g->npx, npy = g->npy, ng = g->ng;
  int i, j;
  for (j = 1; j <= ng; j++)
    for (i = 1; i <= ng; i++) {
      if (dir == 1) {
        if (g->sw_corner) q[IB(1 - i, 1 - j)] = q[IB(1 - j, i + 1)];
        if (g->nw_corner) q[IB(1 - i, npy + j)] = q[IB(1 - j, npy - i)];
        if (g->se_corner) q[IB(npx + i, 1 - j)] = q[IB(npx + j, i + 1)];
        if (g->ne_corner) q[IB(npx + i, npy + j)] = q[IB(npx + j, npy - i)];
      } else {
        if (g->sw_corner) q[IB(1 - j, 1 - i)] = q[IB(i + 1, 1 - j)];
        if (g->nw_corner) q[IB(1 - j, npy + i)] = q[IB(i + 1, npy + j)];
        if (g->se_corner) q[IB(npx + j, 1 - i)] = q[IB(npx - i, 1 - j)];
        if (g->ne_corner) q[IB(npx + j, npy + i)] = q[IB(npx - i, npy + j)];
      }
    }
}

/* fill_corners_dgrid (tools/fv_mp_mod.F90:1290-1322): x on the U layout (isd:ied, jsd:jed+1), y on the V layout */
void fvo_fill_corners_dgrid(const fvo_grid *g, double *x, double *y, double mySign) {
  BOUNDS(g);
  const int npx = g->npx, npy = g->npy, ng = g->ng;
  int i, j;
  for (j = 1; j <= ng; j++)
    for (i = 1; i <= ng; i++) {
      if (g->sw_corner) x[IU(1 - i, 1 - j)] = mySign * y[IV(1 - j, i)];
      if (g->nw_corner) x[IU(1 - i, npy + j)] = y[IV(1 - j, npy - i)];
      if (g->se_corner) x[IU(npx - 1 + i, 1 - j)] = y[IV(npx + j, i)];
      if (g->ne_corner) x[IU(npx - 1 + i, npy + j)] = mySign * y[IV(npx + j, npy - i)];
    }
  for (j = 1; j <= ng; j++)
    for (i = 1; i <= ng; i++) {
      if (g->sw_corner) y[IV(1 - i, 1 - j)] = mySign * x[IU(j, 1 - i)];
      if (g->nw_corner) y[IV(1 - i, npy - 1 + j)] = x[IU(j, npy + i)];
      if (g->se_corner) y[IV(npx + i, 1 - j)] = x[IU(npx - j, 1 - i)];
      if (g->ne_corner) y[IV(npx + i, npy - 1 + j)] = mySign * x[IU(npx - j, npy + i)];
    }
}

/* ------------------------------------------------------------------------------------------
 * d_sw, sw_core.F90:494-1606 (do_f3d=.false.)
 * ---------------------------------------------------------------------------------------- */
int fvo_d_sw(const fvo_grid *g, const fvo_dsw_par *p, double *delpc, double *delp, double *ptc,
             double *pt, double *u, double *v, double *w, double *uc, double *vc, double *ua,
             double *va, double *divg_d, double *xflux, double *yflux, double *cx, double *cy,
             double *crx_adv, double *cry_adv, double *xfx_adv, double *yfx_adv, double *q_con,
             double *heat_source, double *diss_est) {
  BOUNDS(g);
  const double dt = p->dt;
  const int nord = p->nord, nord_v = p->nord_v, nord_w = p->nord_w, nord_t = p->nord_t;
  const double dddmp = p->dddmp, d2_bg = p->d2_bg, d4_bg = p->d4_bg, damp_v = p->damp_v,
               damp_w = p->damp_w, damp_t = p->damp_t, d_con = p->d_con, kgb = p->kgb;
  const int hydrostatic = p->hydrostatic, use_cond = p->use_cond;
  int i, j, n, nt, n2;
  double damp, damp2, damp4, dd8, u2, v2, du2, dv2, tmp, dt5;
  (void)ua;
  (void)va;
  const int npx = g->npx, npy = g->npy;
  const int cubed = g->grid_type < 3;
  if (g->grid_type == 3 || g->bounded_domain || g->do_f3d) return FVO_ERR_UNSUPPORTED;
  const size_t nA = (size_t)nid * njd, nU = (size_t)nid * (njd + 1), nV = (size_t)(nid + 1) * njd,
               nB = (size_t)(nid + 1) * (njd + 1);
  double *ut = dalloc(nV), *vt = dalloc(nU), *fx2 = dalloc(nV), *fy2 = dalloc(nU);
  double *dw = dalloc((size_t)nx * ny), *ub = dalloc((size_t)(nx + 1) * (ny + 1)),
         *vb = dalloc((size_t)(nx + 1) * (ny + 1));
  double *wk = dalloc(nA), *ke = dalloc(nB), *vort = dalloc(nA);
  double *fx = dalloc((size_t)(nx + 1) * ny), *fy = dalloc((size_t)nx * (ny + 1));
  double *ra_x = dalloc((size_t)nx * njd), *ra_y = dalloc((size_t)nid * ny);
  double *gx = dalloc((size_t)(nx + 1) * ny), *gy = dalloc((size_t)nx * (ny + 1));

  if (cubed) { /* :652-846: contravariant winds on the cubed sphere, face edges and the 2x2 solves at the corners */
#define UT(i, j) ut[IV(i, j)]
#define VT(i, j) vt[IU(i, j)]
#define UC(i, j) uc[IV(i, j)]
#define VC(i, j) vc[IU(i, j)]
#define CU(i, j) g->cosa_u[IV(i, j)]
#define CV(i, j) g->cosa_v[IU(i, j)]
    for (j = jsd; j <= jed; j++)
      if (j != 0 && j != 1 && j != (npy - 1) && j != npy)
        for (i = is - 1; i <= ie + 2; i++)
          UT(i, j) = (UC(i, j) - 0.25 * CU(i, j) * (VC(i - 1, j) + VC(i, j) + VC(i - 1, j + 1) + VC(i, j + 1))) * g->rsin_u[IV(i, j)];
    for (j = js - 1; j <= je + 2; j++)
      if (j != 1 && j != npy)
        for (i = isd; i <= ied; i++)
          VT(i, j) = (VC(i, j) - 0.25 * CV(i, j) * (UC(i, j - 1) + UC(i + 1, j - 1) + UC(i, j) + UC(i + 1, j))) * g->rsin_v[IU(i, j)];
    if (is == 1) { /* West edge */
      for (j = jsd; j <= jed; j++) {
        if (UC(1, j) * dt > 0.)
          UT(1, j) = UC(1, j) / SIN_SG(0, j, 3);
        else
          UT(1, j) = UC(1, j) / SIN_SG(1, j, 1);
      }
      for (j = IMAX(3, js); j <= IMIN(npy - 2, je + 1); j++) {
        VT(0, j) = VC(0, j) - 0.25 * CV(0, j) * (UT(0, j - 1) + UT(1, j - 1) + UT(0, j) + UT(1, j));
        VT(1, j) = VC(1, j) - 0.25 * CV(1, j) * (UT(1, j - 1) + UT(2, j - 1) + UT(1, j) + UT(2, j));
      }
    }
    if ((ie + 1) == npx) { /* East edge */
      for (j = jsd; j <= jed; j++) {
        if (UC(npx, j) * dt > 0.)
          UT(npx, j) = UC(npx, j) / SIN_SG(npx - 1, j, 3);
        else
          UT(npx, j) = UC(npx, j) / SIN_SG(npx, j, 1);
      }
      for (j = IMAX(3, js); j <= IMIN(npy - 2, je + 1); j++) {
        VT(npx - 1, j) = VC(npx - 1, j) - 0.25 * CV(npx - 1, j) * (UT(npx - 1, j - 1) + UT(npx, j - 1) + UT(npx - 1, j) + UT(npx, j));
        VT(npx, j) = VC(npx, j) - 0.25 * CV(npx, j) * (UT(npx, j - 1) + UT(npx + 1, j - 1) + UT(npx, j) + UT(npx + 1, j));
      }
    }
    if (js == 1) { /* South edge */
      for (i = isd; i <= ied; i++) {
        if (VC(i, 1) * dt > 0.)
          VT(i, 1) = VC(i, 1) / SIN_SG(i, 0, 4);
        else
          VT(i, 1) = VC(i, 1) / SIN_SG(i, 1, 2);
      }
      for (i = IMAX(3, is); i <= IMIN(npx - 2, ie + 1); i++) {
        UT(i, 0) = UC(i, 0) - 0.25 * CU(i, 0) * (VT(i - 1, 0) + VT(i, 0) + VT(i - 1, 1) + VT(i, 1));
        UT(i, 1) = UC(i, 1) - 0.25 * CU(i, 1) * (VT(i - 1, 1) + VT(i, 1) + VT(i - 1, 2) + VT(i, 2));
      }
    }
    if ((je + 1) == npy) { /* North edge */
      for (i = isd; i <= ied; i++) {
        if (VC(i, npy) * dt > 0.)
          VT(i, npy) = VC(i, npy) / SIN_SG(i, npy - 1, 4);
        else
          VT(i, npy) = VC(i, npy) / SIN_SG(i, npy, 2);
      }
      for (i = IMAX(3, is); i <= IMIN(npx - 2, ie + 1); i++) {
        UT(i, npy - 1) = UC(i, npy - 1) - 0.25 * CU(i, npy - 1) * (VT(i - 1, npy - 1) + VT(i, npy - 1) + VT(i - 1, npy) + VT(i, npy));
        UT(i, npy) = UC(i, npy) - 0.25 * CU(i, npy) * (VT(i - 1, npy) + VT(i, npy) + VT(i - 1, npy + 1) + VT(i, npy + 1));
      }
    }
    /* 2x2 systems for the parallel-to-edge components next to the corners, :779-846 */
    if (g->sw_corner) {
      damp = 1. / (1. - 0.0625 * CU(2, 0) * CV(1, 0));
      UT(2, 0) = (UC(2, 0) - 0.25 * CU(2, 0) * (VT(1, 1) + VT(2, 1) + VT(2, 0) + VC(1, 0) - 0.25 * CV(1, 0) * (UT(1, 0) + UT(1, -1) + UT(2, -1)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(0, 1) * CV(0, 2));
      VT(0, 2) = (VC(0, 2) - 0.25 * CV(0, 2) * (UT(1, 1) + UT(1, 2) + UT(0, 2) + UC(0, 1) - 0.25 * CU(0, 1) * (VT(0, 1) + VT(-1, 1) + VT(-1, 2)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(2, 1) * CV(1, 2));
      UT(2, 1) = (UC(2, 1) - 0.25 * CU(2, 1) * (VT(1, 1) + VT(2, 1) + VT(2, 2) + VC(1, 2) - 0.25 * CV(1, 2) * (UT(1, 1) + UT(1, 2) + UT(2, 2)))) * damp;
      VT(1, 2) = (VC(1, 2) - 0.25 * CV(1, 2) * (UT(1, 1) + UT(1, 2) + UT(2, 2) + UC(2, 1) - 0.25 * CU(2, 1) * (VT(1, 1) + VT(2, 1) + VT(2, 2)))) * damp;
    }
    if (g->se_corner) {
      damp = 1. / (1. - 0.0625 * CU(npx - 1, 0) * CV(npx - 1, 0));
      UT(npx - 1, 0) = (UC(npx - 1, 0) - 0.25 * CU(npx - 1, 0) * (VT(npx - 1, 1) + VT(npx - 2, 1) + VT(npx - 2, 0) + VC(npx - 1, 0) -
                        0.25 * CV(npx - 1, 0) * (UT(npx, 0) + UT(npx, -1) + UT(npx - 1, -1)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(npx + 1, 1) * CV(npx, 2));
      VT(npx, 2) = (VC(npx, 2) - 0.25 * CV(npx, 2) * (UT(npx, 1) + UT(npx, 2) + UT(npx + 1, 2) + UC(npx + 1, 1) -
                    0.25 * CU(npx + 1, 1) * (VT(npx, 1) + VT(npx + 1, 1) + VT(npx + 1, 2)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(npx - 1, 1) * CV(npx - 1, 2));
      UT(npx - 1, 1) = (UC(npx - 1, 1) - 0.25 * CU(npx - 1, 1) * (VT(npx - 1, 1) + VT(npx - 2, 1) + VT(npx - 2, 2) + VC(npx - 1, 2) -
                        0.25 * CV(npx - 1, 2) * (UT(npx, 1) + UT(npx, 2) + UT(npx - 1, 2)))) * damp;
      VT(npx - 1, 2) = (VC(npx - 1, 2) - 0.25 * CV(npx - 1, 2) * (UT(npx, 1) + UT(npx, 2) + UT(npx - 1, 2) + UC(npx - 1, 1) -
                        0.25 * CU(npx - 1, 1) * (VT(npx - 1, 1) + VT(npx - 2, 1) + VT(npx - 2, 2)))) * damp;
    }
    if (g->ne_corner) {
      damp = 1. / (1. - 0.0625 * CU(npx - 1, npy) * CV(npx - 1, npy + 1));
      UT(npx - 1, npy) = (UC(npx - 1, npy) - 0.25 * CU(npx - 1, npy) * (VT(npx - 1, npy) + VT(npx - 2, npy) + VT(npx - 2, npy + 1) + VC(npx - 1, npy + 1) -
                          0.25 * CV(npx - 1, npy + 1) * (UT(npx, npy) + UT(npx, npy + 1) + UT(npx - 1, npy + 1)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(npx + 1, npy - 1) * CV(npx, npy - 1));
      VT(npx, npy - 1) = (VC(npx, npy - 1) - 0.25 * CV(npx, npy - 1) * (UT(npx, npy - 1) + UT(npx, npy - 2) + UT(npx + 1, npy - 2) + UC(npx + 1, npy - 1) -
                          0.25 * CU(npx + 1, npy - 1) * (VT(npx, npy) + VT(npx + 1, npy) + VT(npx + 1, npy - 1)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(npx - 1, npy - 1) * CV(npx - 1, npy - 1));
      UT(npx - 1, npy - 1) = (UC(npx - 1, npy - 1) - 0.25 * CU(npx - 1, npy - 1) * (VT(npx - 1, npy) + VT(npx - 2, npy) + VT(npx - 2, npy - 1) + VC(npx - 1, npy - 1) -
                              0.25 * CV(npx - 1, npy - 1) * (UT(npx, npy - 1) + UT(npx, npy - 2) + UT(npx - 1, npy - 2)))) * damp;
      VT(npx - 1, npy - 1) = (VC(npx - 1, npy - 1) - 0.25 * CV(npx - 1, npy - 1) * (UT(npx, npy - 1) + UT(npx, npy - 2) + UT(npx - 1, npy - 2) + UC(npx - 1, npy - 1) -
                              0.25 * CU(npx - 1, npy - 1) * (VT(npx - 1, npy) + VT(npx - 2, npy) + VT(npx - 2, npy - 1)))) * damp;
    }
    if (g->nw_corner) {
      damp = 1. / (1. - 0.0625 * CU(2, npy) * CV(1, npy + 1));
      UT(2, npy) = (UC(2, npy) - 0.25 * CU(2, npy) * (VT(1, npy) + VT(2, npy) + VT(2, npy + 1) + VC(1, npy + 1) -
                    0.25 * CV(1, npy + 1) * (UT(1, npy) + UT(1, npy + 1) + UT(2, npy + 1)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(0, npy - 1) * CV(0, npy - 1));
      VT(0, npy - 1) = (VC(0, npy - 1) - 0.25 * CV(0, npy - 1) * (UT(1, npy - 1) + UT(1, npy - 2) + UT(0, npy - 2) + UC(0, npy - 1) -
                        0.25 * CU(0, npy - 1) * (VT(0, npy) + VT(-1, npy) + VT(-1, npy - 1)))) * damp;
      damp = 1. / (1. - 0.0625 * CU(2, npy - 1) * CV(1, npy - 1));
      UT(2, npy - 1) = (UC(2, npy - 1) - 0.25 * CU(2, npy - 1) * (VT(1, npy) + VT(2, npy) + VT(2, npy - 1) + VC(1, npy - 1) -
                        0.25 * CV(1, npy - 1) * (UT(1, npy - 1) + UT(1, npy - 2) + UT(2, npy - 2)))) * damp;
      VT(1, npy - 1) = (VC(1, npy - 1) - 0.25 * CV(1, npy - 1) * (UT(1, npy - 1) + UT(1, npy - 2) + UT(2, npy - 2) + UC(2, npy - 1) -
                        0.25 * CU(2, npy - 1) * (VT(1, npy) + VT(2, npy) + VT(2, npy - 1)))) * damp;
    }
  } else { /* grid_type >= 3: :850-860 */
    for (j = jsd; j <= jed; j++)
      for (i = is; i <= ie + 1; i++) ut[IV(i, j)] = uc[IV(i, j)];
    for (j = js; j <= je + 1; j++)
      for (i = isd; i <= ied; i++) vt[IU(i, j)] = vc[IU(i, j)];
  }
  /* :863-873 */
  for (j = jsd; j <= jed; j++)
    for (i = is; i <= ie + 1; i++) xfx_adv[ICX(i, j)] = dt * ut[IV(i, j)];
  for (j = js; j <= je + 1; j++)
    for (i = isd; i <= ied; i++) yfx_adv[ICY(i, j)] = dt * vt[IU(i, j)];
  /* :879-902 */
  for (j = jsd; j <= jed; j++)
    for (i = is; i <= ie + 1; i++) {
      if (xfx_adv[ICX(i, j)] > 0.) {
        crx_adv[ICX(i, j)] = xfx_adv[ICX(i, j)] * g->rdxa[IA(i - 1, j)];
        xfx_adv[ICX(i, j)] = g->dy[IV(i, j)] * xfx_adv[ICX(i, j)] * SIN_SG(i - 1, j, 3);
      } else {
        crx_adv[ICX(i, j)] = xfx_adv[ICX(i, j)] * g->rdxa[IA(i, j)];
        xfx_adv[ICX(i, j)] = g->dy[IV(i, j)] * xfx_adv[ICX(i, j)] * SIN_SG(i, j, 1);
      }
    }
  for (j = js; j <= je + 1; j++)
    for (i = isd; i <= ied; i++) {
      if (yfx_adv[ICY(i, j)] > 0.) {
        cry_adv[ICY(i, j)] = yfx_adv[ICY(i, j)] * g->rdya[IA(i, j - 1)];
        yfx_adv[ICY(i, j)] = g->dx[IU(i, j)] * yfx_adv[ICY(i, j)] * SIN_SG(i, j - 1, 4);
      } else {
        cry_adv[ICY(i, j)] = yfx_adv[ICY(i, j)] * g->rdya[IA(i, j)];
        yfx_adv[ICY(i, j)] = g->dx[IU(i, j)] * yfx_adv[ICY(i, j)] * SIN_SG(i, j, 2);
      }
    }
  /* :908-917 */
  for (j = jsd; j <= jed; j++)
    for (i = is; i <= ie; i++)
      ra_x[(size_t)(j - jsd) * nx + (i - is)] = g->area[IA(i, j)] + xfx_adv[ICX(i, j)] - xfx_adv[ICX(i + 1, j)];
  for (j = js; j <= je; j++)
    for (i = isd; i <= ied; i++)
      ra_y[ICY(i, j)] = g->area[IA(i, j)] + yfx_adv[ICY(i, j)] - yfx_adv[ICY(i, j + 1)];

  /* :919-920 */
  fvo_fv_tp_2d(g, delp, crx_adv, cry_adv, p->hord_dp, fx, fy, xfx_adv, yfx_adv, ra_x, ra_y, NULL, NULL,
               NULL, nord_v, damp_v);
  /* flux capacitor, :923-940 */
  for (j = jsd; j <= jed; j++)
    for (i = is; i <= ie + 1; i++) cx[ICX(i, j)] = cx[ICX(i, j)] + crx_adv[ICX(i, j)];
  for (j = js; j <= je; j++)
    for (i = is; i <= ie + 1; i++) xflux[IFX(i, j)] = xflux[IFX(i, j)] + fx[IFX(i, j)];
  for (j = js; j <= je + 1; j++) {
    for (i = isd; i <= ied; i++) cy[ICY(i, j)] = cy[ICY(i, j)] + cry_adv[ICY(i, j)];
    for (i = is; i <= ie; i++) yflux[IFY(i, j)] = yflux[IFY(i, j)] + fy[IFY(i, j)];
  }
  /* :943-948 */
  for (j = js; j <= je; j++)
    for (i = is; i <= ie; i++) {
      heat_source[ICC(i, j)] = 0.;
      diss_est[ICC(i, j)] = 0.;
    }
  if (!hydrostatic) { /* :950-990 */
    if (damp_w > 1.E-5) {
      dd8 = kgb * fabs(dt);
      damp4 = ipow(damp_w * g->da_min_c, nord_w + 1);
      fvo_del6_vt_flux(g, nord_w, damp4, w, wk, fx2, fy2);
      if (g->prevent_diss_cooling) {
        for (j = js; j <= je; j++)
          for (i = is; i <= ie; i++) {
            dw[ICC(i, j)] = (fx2[IV(i, j)] - fx2[IV(i + 1, j)] + fy2[IU(i, j)] - fy2[IU(i, j + 1)]) * g->rarea[IA(i, j)];
            tmp = dw[ICC(i, j)] * (w[IA(i, j)] + 0.5 * dw[ICC(i, j)]);
            heat_source[ICC(i, j)] = dd8 - dmin(0., tmp);
            if (g->do_diss_est) diss_est[ICC(i, j)] = dd8 - tmp;
          }
      } else {
        for (j = js; j <= je; j++)
          for (i = is; i <= ie; i++) {
            dw[ICC(i, j)] = (fx2[IV(i, j)] - fx2[IV(i + 1, j)] + fy2[IU(i, j)] - fy2[IU(i, j + 1)]) * g->rarea[IA(i, j)];
            heat_source[ICC(i, j)] = dd8 - dw[ICC(i, j)] * (w[IA(i, j)] + 0.5 * dw[ICC(i, j)]);
            if (g->do_diss_est) diss_est[ICC(i, j)] = heat_source[ICC(i, j)];
          }
      }
    }
    fvo_fv_tp_2d(g, w, crx_adv, cry_adv, p->hord_vt, gx, gy, xfx_adv, yfx_adv, ra_x, ra_y, fx, fy, NULL, -1, 0.);
    for (j = js; j <= je; j++)
      for (i = is; i <= ie; i++)
        w[IA(i, j)] = delp[IA(i, j)] * w[IA(i, j)] +
                      (gx[IFX(i, j)] - gx[IFX(i + 1, j)] + gy[IFY(i, j)] - gy[IFY(i, j + 1)]) * g->rarea[IA(i, j)];
  }
  if (use_cond) { /* :992-1000 */
    fvo_fv_tp_2d(g, q_con, crx_adv, cry_adv, p->hord_dp, gx, gy, xfx_adv, yfx_adv, ra_x, ra_y, fx, fy, delp,
                 nord_t, damp_t);
    for (j = js; j <= je; j++)
      for (i = is; i <= ie; i++)
        q_con[IA(i, j)] = delp[IA(i, j)] * q_con[IA(i, j)] +
                          (gx[IFX(i, j)] - gx[IFX(i + 1, j)] + gy[IFY(i, j)] - gy[IFY(i, j + 1)]) * g->rarea[IA(i, j)];
  }
  /* :1014-1016 (not GFS_PHYS/DCMIP: nord_t, damp_t) */
  fvo_fv_tp_2d(g, pt, crx_adv, cry_adv, p->hord_tm, gx, gy, xfx_adv, yfx_adv, ra_x, ra_y, fx, fy, delp, nord_t,
               damp_t);
  if (p->inline_q) { /* :1020-1043 */
    double *wq = dalloc(nA);
    int iq;
    for (j = js; j <= je; j++)
      for (i = is; i <= ie; i++) {
        wq[IA(i, j)] = delp[IA(i, j)];
        delp[IA(i, j)] = wq[IA(i, j)] + (fx[IFX(i, j)] - fx[IFX(i + 1, j)] + fy[IFY(i, j)] - fy[IFY(i, j + 1)]) * g->rarea[IA(i, j)];
        pt[IA(i, j)] = (pt[IA(i, j)] * wq[IA(i, j)] +
                        (gx[IFX(i, j)] - gx[IFX(i + 1, j)] + gy[IFY(i, j)] - gy[IFY(i, j + 1)]) * g->rarea[IA(i, j)]) /
                       delp[IA(i, j)];
      }
    for (iq = 0; iq < p->nq; iq++) {
      double *q = p->q + (size_t)iq * p->q_stride;
      /* mass = delp: the compute domain already holds the new values, the halo the old ones */
      fvo_fv_tp_2d(g, q, crx_adv, cry_adv, p->hord_tr, gx, gy, xfx_adv, yfx_adv, ra_x, ra_y, fx, fy, delp, nord_t, damp_t);
      for (j = js; j <= je; j++)
        for (i = is; i <= ie; i++)
          q[IA(i, j)] = (q[IA(i, j)] * wq[IA(i, j)] +
                         (gx[IFX(i, j)] - gx[IFX(i + 1, j)] + gy[IFY(i, j)] - gy[IFY(i, j + 1)]) * g->rarea[IA(i, j)]) /
                        delp[IA(i, j)];
    }
    free(wq);
  } else
  /* :1053-1066 */
  for (j = js; j <= je; j++)
    for (i = is; i <= ie; i++) {
      pt[IA(i, j)] = pt[IA(i, j)] * delp[IA(i, j)] +
                     (gx[IFX(i, j)] - gx[IFX(i + 1, j)] + gy[IFY(i, j)] - gy[IFY(i, j + 1)]) * g->rarea[IA(i, j)];
      delp[IA(i, j)] = delp[IA(i, j)] +
                       (fx[IFX(i, j)] - fx[IFX(i + 1, j)] + fy[IFY(i, j)] - fy[IFY(i, j + 1)]) * g->rarea[IA(i, j)];
      pt[IA(i, j)] = pt[IA(i, j)] / delp[IA(i, j)];
    }

  /* Kinetic energy fluxes, :1078-1198 */
  dt5 = 0.5 * dt;
  {
    const double dt4 = 0.25 * dt;
    const int is2 = IMAX(2, is), ie1 = IMIN(npx - 1, ie + 1), js2 = IMAX(2, js), je1 = IMIN(npy - 1, je + 1);
    if (cubed) { /* :1099-1126 */
      if (js == 1)
        for (i = is; i <= ie + 1; i++) vb[IBC(i, 1)] = dt5 * (VT(i - 1, 1) + VT(i, 1)); /* corner values are incorrect */
      for (j = js2; j <= je1; j++) {
        for (i = is2; i <= ie1; i++)
          vb[IBC(i, j)] = dt5 * (VC(i - 1, j) + VC(i, j) - (UC(i, j - 1) + UC(i, j)) * g->cosa[IB(i, j)]) * g->rsina[IBC(i, j)];
        if (is == 1) vb[IBC(1, j)] = dt4 * (-VT(-1, j) + 3. * (VT(0, j) + VT(1, j)) - VT(2, j));
        if ((ie + 1) == npx) vb[IBC(npx, j)] = dt4 * (-VT(npx - 2, j) + 3. * (VT(npx - 1, j) + VT(npx, j)) - VT(npx + 1, j));
      }
      if ((je + 1) == npy)
        for (i = is; i <= ie + 1; i++) vb[IBC(i, npy)] = dt5 * (VT(i - 1, npy) + VT(i, npy));
    } else {
      for (j = js; j <= je + 1; j++)
        for (i = is; i <= ie + 1; i++) vb[IBC(i, j)] = dt5 * (vc[IU(i - 1, j)] + vc[IU(i, j)]);
    }
    fvo_ytp_v(g, vb, u, v, ub, p->hord_mt);
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) ke[IB(i, j)] = vb[IBC(i, j)] * ub[IBC(i, j)];
    if (cubed) { /* :1157-1181 */
      if (is == 1)
        for (j = js; j <= je + 1; j++) ub[IBC(1, j)] = dt5 * (UT(1, j - 1) + UT(1, j));
      for (j = js; j <= je + 1; j++) {
        if (j == 1 || j == npy) {
          for (i = is2; i <= ie1; i++) ub[IBC(i, j)] = dt4 * (-UT(i, j - 2) + 3. * (UT(i, j - 1) + UT(i, j)) - UT(i, j + 1));
        } else {
          for (i = is2; i <= ie1; i++)
            ub[IBC(i, j)] = dt5 * (UC(i, j - 1) + UC(i, j) - (VC(i - 1, j) + VC(i, j)) * g->cosa[IB(i, j)]) * g->rsina[IBC(i, j)];
        }
      }
      if ((ie + 1) == npx)
        for (j = js; j <= je + 1; j++) ub[IBC(npx, j)] = dt5 * (UT(npx, j - 1) + UT(npx, j));
    } else {
      for (j = js; j <= je + 1; j++)
        for (i = is; i <= ie + 1; i++) ub[IBC(i, j)] = dt5 * (uc[IV(i, j - 1)] + uc[IV(i, j)]);
    }
    fvo_xtp_u(g, ub, u, v, vb, p->hord_mt);
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) ke[IB(i, j)] = 0.5 * (ke[IB(i, j)] + ub[IBC(i, j)] * vb[IBC(i, j)]);
    /* Fix KE at the 4 corners of the face, :1203-1228 */
    {
      const double dt6 = dt / 6.;
      if (g->sw_corner)
        ke[IB(1, 1)] = dt6 * ((UT(1, 1) + UT(1, 0)) * u[IU(1, 1)] + (VT(1, 1) + VT(0, 1)) * v[IV(1, 1)] + (UT(1, 1) + VT(1, 1)) * u[IU(0, 1)]);
      if (g->se_corner) {
        i = npx;
        ke[IB(i, 1)] = dt6 * ((UT(i, 1) + UT(i, 0)) * u[IU(i - 1, 1)] + (VT(i, 1) + VT(i - 1, 1)) * v[IV(i, 1)] + (UT(i, 1) - VT(i - 1, 1)) * u[IU(i, 1)]);
      }
      if (g->ne_corner) {
        i = npx;
        j = npy;
        ke[IB(i, j)] = dt6 * ((UT(i, j) + UT(i, j - 1)) * u[IU(i - 1, j)] + (VT(i, j) + VT(i - 1, j)) * v[IV(i, j - 1)] + (UT(i, j - 1) + VT(i - 1, j)) * u[IU(i, j)]);
      }
      if (g->nw_corner) {
        j = npy;
        ke[IB(1, j)] = dt6 * ((UT(1, j) + UT(1, j - 1)) * u[IU(1, j)] + (VT(1, j) + VT(0, j)) * v[IV(1, j - 1)] + (UT(1, j - 1) - VT(1, j)) * u[IU(0, j)]);
      }
    }
  }

  /* vorticity, :1231-1247 */
  for (j = jsd; j <= jed + 1; j++)
    for (i = isd; i <= ied; i++) vt[IU(i, j)] = u[IU(i, j)] * g->dx[IU(i, j)];
  for (j = jsd; j <= jed; j++)
    for (i = isd; i <= ied + 1; i++) ut[IV(i, j)] = v[IV(i, j)] * g->dy[IV(i, j)];
  for (j = jsd; j <= jed; j++)
    for (i = isd; i <= ied; i++)
      wk[IA(i, j)] = g->rarea[IA(i, j)] * (vt[IU(i, j)] - vt[IU(i, j + 1)] - ut[IV(i, j)] + ut[IV(i + 1, j)]);

  if (!hydrostatic) { /* :1249-1276 */
    for (j = js; j <= je; j++)
      for (i = is; i <= ie; i++) w[IA(i, j)] = w[IA(i, j)] / delp[IA(i, j)];
    if (damp_w > 1.E-5) {
      for (j = js; j <= je; j++)
        for (i = is; i <= ie; i++) w[IA(i, j)] = w[IA(i, j)] + dw[ICC(i, j)];
    }
  }
  if (use_cond) { /* :1277-1283 */
    for (j = js; j <= je; j++)
      for (i = is; i <= ie; i++) q_con[IA(i, j)] = q_con[IA(i, j)] / delp[IA(i, j)];
  }

  /* divergence damping */
  if (nord == 0) { /* :1290-1371, non-bounded branch (:1309-1350) incl. its global-index edge rules */
    const int is2 = (2 > is ? 2 : is), ie1 = (npx - 1 < ie + 1 ? npx - 1 : ie + 1);
    for (j = js; j <= je + 1; j++) {
      if (j == 1 || j == npy) {
        for (i = is - 1; i <= ie + 1; i++) {
          if (vc[IU(i, j)] > 0)
            ptc[IA(i, j)] = u[IU(i, j)] * g->dyc[IU(i, j)] * SIN_SG(i, j - 1, 4);
          else
            ptc[IA(i, j)] = u[IU(i, j)] * g->dyc[IU(i, j)] * SIN_SG(i, j, 2);
        }
      } else {
        for (i = is - 1; i <= ie + 1; i++)
          ptc[IA(i, j)] = (u[IU(i, j)] - 0.5 * (va[IA(i, j - 1)] + va[IA(i, j)]) * g->cosa_v[IU(i, j)]) *
                          g->dyc[IU(i, j)] * g->sina_v[IU(i, j)];
      }
    }
    for (j = js - 1; j <= je + 1; j++) {
      for (i = is2; i <= ie1; i++)
        vort[IA(i, j)] = (v[IV(i, j)] - 0.5 * (ua[IA(i - 1, j)] + ua[IA(i, j)]) * g->cosa_u[IV(i, j)]) *
                         g->dxc[IV(i, j)] * g->sina_u[IV(i, j)];
      if (is == 1) {
        if (uc[IV(1, j)] > 0)
          vort[IA(1, j)] = v[IV(1, j)] * g->dxc[IV(1, j)] * SIN_SG(0, j, 3);
        else
          vort[IA(1, j)] = v[IV(1, j)] * g->dxc[IV(1, j)] * SIN_SG(1, j, 1);
      }
      if ((ie + 1) == npx) {
        if (uc[IV(npx, j)] > 0)
          vort[IA(npx, j)] = v[IV(npx, j)] * g->dxc[IV(npx, j)] * SIN_SG(npx - 1, j, 3);
        else
          vort[IA(npx, j)] = v[IV(npx, j)] * g->dxc[IV(npx, j)] * SIN_SG(npx, j, 1);
      }
    }
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++)
        delpc[IA(i, j)] = vort[IA(i, j - 1)] - vort[IA(i, j)] + ptc[IA(i - 1, j)] - ptc[IA(i, j)];
    /* Remove the extra term at the corners, :1357-1360 */
    if (g->sw_corner) delpc[IA(1, 1)] = delpc[IA(1, 1)] - vort[IA(1, 0)];
    if (g->se_corner) delpc[IA(npx, 1)] = delpc[IA(npx, 1)] - vort[IA(npx, 0)];
    if (g->ne_corner) delpc[IA(npx, npy)] = delpc[IA(npx, npy)] + vort[IA(npx, npy)];
    if (g->nw_corner) delpc[IA(1, npy)] = delpc[IA(1, npy)] + vort[IA(1, npy)];
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) {
        delpc[IA(i, j)] = g->rarea_c[IB(i, j)] * delpc[IA(i, j)];
        damp = g->da_min_c * dmax(d2_bg, dmin(0.20, dddmp * fabs(delpc[IA(i, j)] * dt)));
        vort[IA(i, j)] = damp * delpc[IA(i, j)];
        ke[IB(i, j)] = ke[IB(i, j)] + vort[IA(i, j)];
      }
  } else { /* :1372-1460 */
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) delpc[IA(i, j)] = divg_d[IB(i, j)];
    n2 = nord + 1;
    for (n = 1; n <= nord; n++) {
      nt = nord - n;
      const int fill_c = (nt != 0) && cubed && (g->sw_corner || g->se_corner || g->ne_corner || g->nw_corner); /* :1387-1389 */
      if (fill_c) fvo_fill_corners_b(g, divg_d, 1);
      for (j = js - nt; j <= je + 1 + nt; j++)
        for (i = is - 1 - nt; i <= ie + 1 + nt; i++)
          vc[IU(i, j)] = (divg_d[IB(i + 1, j)] - divg_d[IB(i, j)]) * g->divg_u[IU(i, j)];
      if (fill_c) fvo_fill_corners_b(g, divg_d, 2);
      for (j = js - 1 - nt; j <= je + 1 + nt; j++)
        for (i = is - nt; i <= ie + 1 + nt; i++)
          uc[IV(i, j)] = (divg_d[IB(i, j + 1)] - divg_d[IB(i, j)]) * g->divg_v[IV(i, j)];
      if (fill_c) fvo_fill_corners_dgrid(g, vc, uc, -1.);
      for (j = js - nt; j <= je + 1 + nt; j++)
        for (i = is - nt; i <= ie + 1 + nt; i++)
          divg_d[IB(i, j)] = uc[IV(i, j - 1)] - uc[IV(i, j)] + vc[IU(i - 1, j)] - vc[IU(i, j)];
      /* Remove the extra term at the corners, :1413-1416 */
      if (g->sw_corner) divg_d[IB(1, 1)] = divg_d[IB(1, 1)] - uc[IV(1, 0)];
      if (g->se_corner) divg_d[IB(npx, 1)] = divg_d[IB(npx, 1)] - uc[IV(npx, 0)];
      if (g->ne_corner) divg_d[IB(npx, npy)] = divg_d[IB(npx, npy)] + uc[IV(npx, npy)];
      if (g->nw_corner) divg_d[IB(1, npy)] = divg_d[IB(1, npy)] + uc[IV(1, npy)];
      if (!g->stretched_grid) {
        for (j = js - nt; j <= je + 1 + nt; j++)
          for (i = is - nt; i <= ie + 1 + nt; i++) divg_d[IB(i, j)] = divg_d[IB(i, j)] * g->rarea_c[IB(i, j)];
      }
    }
    if (dddmp < 1.E-5) {
      memset(vort, 0, sizeof(double) * nA);
    } else if (cubed) { /* :1431-1440: relative vorticity interpolated to the cell corners */
      fvo_a2b_ord4(g, wk, vort, 0);
      for (j = js; j <= je + 1; j++)
        for (i = is; i <= ie + 1; i++)
          vort[IA(i, j)] = fabs(dt) * sqrt(delpc[IA(i, j)] * delpc[IA(i, j)] + vort[IA(i, j)] * vort[IA(i, j)]);
    } else {
      fvo_smag_corner(g, fabs(dt), u, v, vort); /* grid_type>=3, :1441 */
    }
    if (g->stretched_grid)
      dd8 = g->da_min * ipow(d4_bg, n2);
    else
      dd8 = ipow(g->da_min_c * d4_bg, n2);
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie + 1; i++) {
        damp2 = g->da_min_c * dmax(d2_bg, dmin(0.20, dddmp * vort[IA(i, j)]));
        vort[IA(i, j)] = damp2 * delpc[IA(i, j)] + dd8 * divg_d[IB(i, j)];
        ke[IB(i, j)] = ke[IB(i, j)] + vort[IA(i, j)];
      }
  }

  if (d_con > 1.e-5 || g->do_diss_est) { /* :1462-1473 */
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++) ub[IBC(i, j)] = vort[IA(i, j)] - vort[IA(i + 1, j)];
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++) vb[IBC(i, j)] = vort[IA(i, j)] - vort[IA(i, j + 1)];
  }

  /* vorticity transport, :1476-1509 */
  for (j = jsd; j <= jed; j++)
    for (i = isd; i <= ied; i++) vort[IA(i, j)] = wk[IA(i, j)] + g->f0[IA(i, j)];
  fvo_fv_tp_2d(g, vort, crx_adv, cry_adv, p->hord_vt, fx, fy, xfx_adv, yfx_adv, ra_x, ra_y, NULL, NULL, NULL,
               -1, 0.);
  for (j = js; j <= je + 1; j++)
    for (i = is; i <= ie; i++) u[IU(i, j)] = vt[IU(i, j)] + ke[IB(i, j)] - ke[IB(i + 1, j)] + fy[IFY(i, j)];
  for (j = js; j <= je; j++)
    for (i = is; i <= ie + 1; i++) v[IV(i, j)] = ut[IV(i, j)] + ke[IB(i, j)] - ke[IB(i, j + 1)] - fx[IFX(i, j)];

  /* damping applied to relative vorticity, :1513-1519 */
  if (damp_v > 1.E-5) {
    damp4 = ipow(damp_v * g->da_min_c, nord_v + 1);
    fvo_del6_vt_flux(g, nord_v, damp4, wk, vort, ut, vt);
  } else if (g->do_diss_est) {
    memset(ut, 0, sizeof(double) * nV);
    memset(vt, 0, sizeof(double) * nU);
  }

  if (d_con > 1.e-5 || g->do_diss_est) { /* :1523-1586 */
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++) {
        ub[IBC(i, j)] = (ub[IBC(i, j)] + vt[IU(i, j)]) * g->rdx[IU(i, j)];
        fy[IFY(i, j)] = u[IU(i, j)] * g->rdx[IU(i, j)];
        gy[IFY(i, j)] = fy[IFY(i, j)] * ub[IBC(i, j)];
      }
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++) {
        vb[IBC(i, j)] = (vb[IBC(i, j)] - ut[IV(i, j)]) * g->rdy[IV(i, j)];
        fx[IFX(i, j)] = v[IV(i, j)] * g->rdy[IV(i, j)];
        gx[IFX(i, j)] = fx[IFX(i, j)] * vb[IBC(i, j)];
      }
    damp = 0.25 * d_con;
    for (j = js; j <= je; j++)
      for (i = is; i <= ie; i++) {
        u2 = fy[IFY(i, j)] + fy[IFY(i, j + 1)];
        du2 = ub[IBC(i, j)] + ub[IBC(i, j + 1)];
        v2 = fx[IFX(i, j)] + fx[IFX(i + 1, j)];
        dv2 = vb[IBC(i, j)] + vb[IBC(i + 1, j)];
        tmp = g->rsin2[IA(i, j)] *
              ((ub[IBC(i, j)] * ub[IBC(i, j)] + ub[IBC(i, j + 1)] * ub[IBC(i, j + 1)] +
                vb[IBC(i, j)] * vb[IBC(i, j)] + vb[IBC(i + 1, j)] * vb[IBC(i + 1, j)]) +
               2. * (gy[IFY(i, j)] + gy[IFY(i, j + 1)] + gx[IFX(i, j)] + gx[IFX(i + 1, j)]) -
               g->cosa_s[IA(i, j)] * (u2 * dv2 + v2 * du2 + du2 * dv2));
        if (g->prevent_diss_cooling) {
          if (d_con > 1.e-5) heat_source[ICC(i, j)] = delp[IA(i, j)] * (heat_source[ICC(i, j)] - damp * dmin(0., tmp));
          if (g->do_diss_est) diss_est[ICC(i, j)] = diss_est[ICC(i, j)] - tmp;
        } else {
          /* :1573-1582: same expression inlined; the factor order damp*rsin2*(...) differs */
          double t2 = (ub[IBC(i, j)] * ub[IBC(i, j)] + ub[IBC(i, j + 1)] * ub[IBC(i, j + 1)] +
                       vb[IBC(i, j)] * vb[IBC(i, j)] + vb[IBC(i + 1, j)] * vb[IBC(i + 1, j)]) +
                      2. * (gy[IFY(i, j)] + gy[IFY(i, j + 1)] + gx[IFX(i, j)] + gx[IFX(i + 1, j)]) -
                      g->cosa_s[IA(i, j)] * (u2 * dv2 + v2 * du2 + du2 * dv2);
          heat_source[ICC(i, j)] = delp[IA(i, j)] * (heat_source[ICC(i, j)] - damp * g->rsin2[IA(i, j)] * t2);
          if (g->do_diss_est) diss_est[ICC(i, j)] = diss_est[ICC(i, j)] - g->rsin2[IA(i, j)] * t2;
        }
      }
  }
  /* Add diffusive fluxes to the momentum equation, :1589-1600 */
  if (damp_v > 1.E-5) {
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++) u[IU(i, j)] = u[IU(i, j)] + vt[IU(i, j)];
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++) v[IV(i, j)] = v[IV(i, j)] - ut[IV(i, j)];
  }

  free(ut);
  free(vt);
  free(fx2);
  free(fy2);
  free(dw);
  free(ub);
  free(vb);
  free(wk);
  free(ke);
  free(vort);
  free(fx);
  free(fy);
  free(ra_x);
  free(ra_y);
  free(gx);
  free(gy);
  return FVO_OK;
}

/* ------------------------------------------------------------------------------------------
 * all-k drivers: the OpenMP k loops of dyn_core.F90:436-447 (c_sw) and :658-812 (d_sw).
 * ---------------------------------------------------------------------------------------- */
int fvo_c_sw_3d(const fvo_grid *g, int npz, double *delpc, double *delp, double *ptc, double *pt,
                double *u, double *v, double *w, double *uc, double *vc, double *ua, double *va,
                double *wc, double *ut, double *vt, double *divg_d, int nord, double dt2,
                int hydrostatic, int dord4) {
  BOUNDS(g);
  const size_t nA = (size_t)nid * njd, nU = (size_t)nid * (njd + 1), nV = (size_t)(nid + 1) * njd,
               nB = (size_t)(nid + 1) * (njd + 1);
  int k, rc = 0;
#pragma omp parallel for schedule(dynamic)
  for (k = 0; k < npz; k++) {
    int r = fvo_c_sw(g, delpc + k * nA, delp + k * nA, ptc + k * nA, pt + k * nA, u + k * nU, v + k * nV,
                     hydrostatic ? NULL : w + k * nA, uc + k * nV, vc + k * nU, ua + k * nA, va + k * nA,
                     hydrostatic ? NULL : wc + k * nA, ut + k * nA, vt + k * nA, divg_d + k * nB, nord, dt2,
                     hydrostatic, dord4);
    if (r) rc = r;
  }
  return rc;
}

int fvo_d_sw_3d(const fvo_grid *g, int npz, const fvo_dsw_par *p, const fvo_dsw_levels *lv,
                double *delpc, double *delp, double *ptc, double *pt, double *u, double *v,
                double *w, double *uc, double *vc, double *ua, double *va, double *divg_d,
                double *mfx, double *mfy, double *cx, double *cy, double *crx, double *cry,
                double *xfx, double *yfx, double *q_con, double *heat_source, double *diss_est) {
  BOUNDS(g);
  const size_t nA = (size_t)nid * njd, nU = (size_t)nid * (njd + 1), nV = (size_t)(nid + 1) * njd,
               nB = (size_t)(nid + 1) * (njd + 1);
  const size_t nCX = (size_t)(nx + 1) * njd, nCY = (size_t)nid * (ny + 1), nFX = (size_t)(nx + 1) * ny,
               nFY = (size_t)nx * (ny + 1), nCC = (size_t)nx * ny;
  int k, rc = 0;
#pragma omp parallel for schedule(dynamic)
  for (k = 0; k < npz; k++) {
    fvo_dsw_par pk = *p;
    pk.nord = lv->nord_k[k];
    pk.nord_v = lv->nord_v[k];
    pk.nord_w = lv->nord_w[k];
    pk.nord_t = lv->nord_t[k];
    pk.d2_bg = lv->d2_divg[k];
    pk.damp_v = lv->damp_vt[k];
    pk.damp_w = lv->damp_w[k];
    pk.damp_t = lv->damp_t[k];
    pk.d_con = lv->d_con_k[k];
    if (p->inline_q) pk.q = p->q + k * nA;
    int r = fvo_d_sw(g, &pk, delpc + k * nA, delp + k * nA, ptc + k * nA, pt + k * nA, u + k * nU, v + k * nV,
                     p->hydrostatic ? NULL : w + k * nA, uc + k * nV, vc + k * nU, ua + k * nA, va + k * nA,
                     divg_d + k * nB, mfx + k * nFX, mfy + k * nFY, cx + k * nCX, cy + k * nCY, crx + k * nCX,
                     cry + k * nCY, xfx + k * nCX, yfx + k * nCY, p->use_cond ? q_con + k * nA : NULL,
                     heat_source + k * nCC, diss_est + k * nCC);
    if (r) rc = r;
  }
  return rc;
}
