/*
 * oracle/tracer2d.c -- CPU oracle (test infrastructure, see fvo.h) for tracer_2d, model/fv_tracer2d.F90:297-557
 * (sub-cycled tracer transport with the accumulated mass fluxes / Courant numbers of the acoustic loop).
 * Single rank of a doubly periodic tile: mp_reduce_max (:405) is the identity and the q halo updates (:474,536)
 * are the periodic contacts of tools/fv_mp_mod.F90:473-483.  id_divg_mean = 0, GLOBAL_CFL not defined.
 */
#include "fvo.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static double *dalloc(size_t n) { return (double *)calloc(n, sizeof(double)); }

static void periodic_fill_A(const fvo_grid *g, double *a) {
  const int is = g->is, ie = g->ie, js = g->js, je = g->je, isd = g->isd, ied = g->ied, jsd = g->jsd, jed = g->jed;
  const int nid = ied - isd + 1, nx = ie - is + 1, ny = je - js + 1;
  int i, j;
  for (j = jsd; j <= jed; j++)
    for (i = isd; i <= ied; i++) {
      int si = i, sj = j;
      if (i >= is && i <= ie && j >= js && j <= je) continue;
      if (si < is) si += nx; else if (si > ie) si -= nx;
      if (sj < js) sj += ny; else if (sj > je) sj -= ny;
      a[(size_t)(j - jsd) * nid + (i - isd)] = a[(size_t)(sj - jsd) * nid + (si - isd)];
    }
}

#define DIMS                                                                                                            \
  const int is = g->is, ie = g->ie, js = g->js, je = g->je, isd = g->isd, ied = g->ied, jsd = g->jsd, jed = g->jed;       \
  const int nid = ied - isd + 1, njd = jed - jsd + 1, nx = ie - is + 1, ny = je - js + 1;                                  \
  const size_t nA = (size_t)nid * njd, nCX = (size_t)(nx + 1) * njd, nCY = (size_t)nid * (ny + 1),                         \
               nFX = (size_t)(nx + 1) * ny, nFY = (size_t)nx * (ny + 1);                                                   \
  (void)nA; (void)nCX; (void)nCY; (void)nFX; (void)nFY; (void)njd
#define IA(i, j) ((size_t)((j)-jsd) * nid + ((i)-isd))
#define IV(i, j) ((size_t)((j)-jsd) * (nid + 1) + ((i)-isd))
#define ICX(i, j) ((size_t)((j)-jsd) * (nx + 1) + ((i)-is))
#define ICY(i, j) ((size_t)((j)-js) * nid + ((i)-isd))
#define IFX(i, j) ((size_t)((j)-js) * (nx + 1) + ((i)-is))
#define IFY(i, j) ((size_t)((j)-js) * nx + ((i)-is))
#define SIN_SG(i, j, n) g->sin_sg[(size_t)((n)-1) * nA + IA(i, j)]

/* :362-400: the area fluxes xfx, yfx (CX / CY x npz) of the accumulated Courant numbers and, for q_split = 0, the largest
 * Courant number of every level on this domain (cmax[npz]; the caller reduces it over the domains, :405) */
void fvo_tracer_2d_prep(const fvo_grid *g, int npz, int q_split, const double *cx, const double *cy, double *xfx, double *yfx,
                        double *cmax) {
  DIMS;
  int i, j, k;
  for (k = 0; k < npz; k++) {
    const double *cxk = cx + nCX * k, *cyk = cy + nCY * k;
    double *xk = xfx + nCX * k, *yk = yfx + nCY * k;
    for (j = jsd; j <= jed; j++)
      for (i = is; i <= ie + 1; i++) {
        if (cxk[ICX(i, j)] > 0.)
          xk[ICX(i, j)] = cxk[ICX(i, j)] * g->dxa[IA(i - 1, j)] * g->dy[IV(i, j)] * SIN_SG(i - 1, j, 3);
        else
          xk[ICX(i, j)] = cxk[ICX(i, j)] * g->dxa[IA(i, j)] * g->dy[IV(i, j)] * SIN_SG(i, j, 1);
      }
    for (j = js; j <= je + 1; j++)
      for (i = isd; i <= ied; i++) {
        if (cyk[ICY(i, j)] > 0.)
          yk[ICY(i, j)] = cyk[ICY(i, j)] * g->dya[IA(i, j - 1)] * g->dx[IA(i, j)] * SIN_SG(i, j - 1, 4);
        else
          yk[ICY(i, j)] = cyk[ICY(i, j)] * g->dya[IA(i, j)] * g->dx[IA(i, j)] * SIN_SG(i, j, 2);
      }
    if (q_split == 0) {
      cmax[k] = 0.;
      if (k + 1 < npz / 6) {
        for (j = js; j <= je; j++)
          for (i = is; i <= ie; i++) cmax[k] = fmax(cmax[k], fmax(fabs(cxk[ICX(i, j)]), fabs(cyk[ICY(i, j)])));
      } else {
        for (j = js; j <= je; j++)
          for (i = is; i <= ie; i++)
            cmax[k] = fmax(cmax[k], fmax(fabs(cxk[ICX(i, j)]), fabs(cyk[ICY(i, j)])) + 1. - SIN_SG(i, j, 5));
      }
    }
  }
}

/* :421-461: the sub-cycling fractions of the levels applied to cx, xfx, mfx, cy, yfx, mfy */
void fvo_tracer_2d_scale(const fvo_grid *g, int npz, const double *frac, double *cx, double *xfx, double *mfx, double *cy,
                         double *yfx, double *mfy) {
  DIMS;
  int i, j, k;
  for (k = 0; k < npz; k++) {
    size_t n;
    for (j = jsd; j <= jed; j++)
      for (i = is; i <= ie + 1; i++) {
        cx[nCX * k + ICX(i, j)] = cx[nCX * k + ICX(i, j)] * frac[k];
        xfx[nCX * k + ICX(i, j)] = xfx[nCX * k + ICX(i, j)] * frac[k];
      }
    for (n = 0; n < nFX; n++) mfx[nFX * k + n] = mfx[nFX * k + n] * frac[k];
    for (j = js; j <= je + 1; j++)
      for (i = isd; i <= ied; i++) {
        cy[nCY * k + ICY(i, j)] = cy[nCY * k + ICY(i, j)] * frac[k];
        yfx[nCY * k + ICY(i, j)] = yfx[nCY * k + ICY(i, j)] * frac[k];
      }
    for (n = 0; n < nFY; n++) mfy[nFY * k + n] = mfy[nFY * k + n] * frac[k];
  }
}

/* :471-541, one sub-cycle `it` of nsplt with the halos of q (and of dp1 when trdm > 1e-4) already filled */
void fvo_tracer_2d_step(const fvo_grid *g, int npz, int nq, int it, int nsplt, const int *ksplt, double *q, double *dp1,
                        const double *mfx, const double *mfy, const double *cx, const double *cy, const double *xfx,
                        const double *yfx, int hord, int nord_tr, double trdm) {
  DIMS;
  int i, j, k, iq;
#pragma omp parallel for private(i, j, iq) schedule(dynamic)
  for (k = 0; k < npz; k++) {
    if (it <= ksplt[k]) {
      double *dp2 = dalloc((size_t)nx * ny), *fx = dalloc(nFX), *fy = dalloc(nFY);
      double *ra_x = dalloc((size_t)nx * njd), *ra_y = dalloc((size_t)nid * ny);
      const double *xk = xfx + nCX * k, *yk = yfx + nCY * k, *mx = mfx + nFX * k, *my = mfy + nFY * k;
      double *d1 = dp1 + nA * k;
      for (j = js; j <= je; j++)
        for (i = is; i <= ie; i++)
          dp2[(size_t)(j - js) * nx + (i - is)] =
              d1[IA(i, j)] + (mx[IFX(i, j)] - mx[IFX(i + 1, j)] + my[IFY(i, j)] - my[IFY(i, j + 1)]) * g->rarea[IA(i, j)];
      for (j = jsd; j <= jed; j++)
        for (i = is; i <= ie; i++) ra_x[(size_t)(j - jsd) * nx + (i - is)] = g->area[IA(i, j)] + xk[ICX(i, j)] - xk[ICX(i + 1, j)];
      for (j = js; j <= je; j++)
        for (i = isd; i <= ied; i++) ra_y[ICY(i, j)] = g->area[IA(i, j)] + yk[ICY(i, j)] - yk[ICY(i, j + 1)];
      for (iq = 0; iq < nq; iq++) {
        double *qk = q + ((size_t)iq * npz + k) * nA;
        if (it == 1 && trdm > 1.e-4)
          fvo_fv_tp_2d(g, qk, cx + nCX * k, cy + nCY * k, hord, fx, fy, xk, yk, ra_x, ra_y, mx, my, d1, nord_tr, trdm);
        else
          fvo_fv_tp_2d(g, qk, cx + nCX * k, cy + nCY * k, hord, fx, fy, xk, yk, ra_x, ra_y, mx, my, NULL, -1, 0.);
        for (j = js; j <= je; j++)
          for (i = is; i <= ie; i++)
            qk[IA(i, j)] = (qk[IA(i, j)] * d1[IA(i, j)] +
                            (fx[IFX(i, j)] - fx[IFX(i + 1, j)] + fy[IFY(i, j)] - fy[IFY(i, j + 1)]) * g->rarea[IA(i, j)]) /
                           dp2[(size_t)(j - js) * nx + (i - is)];
      }
      if (it != nsplt)
        for (j = js; j <= je; j++)
          for (i = is; i <= ie; i++) d1[IA(i, j)] = dp2[(size_t)(j - js) * nx + (i - is)];
      free(dp2); free(fx); free(fy); free(ra_x); free(ra_y);
    }
  }
}

/* tracer_2d on one doubly periodic domain.  q: A x npz x nq; dp1: A x npz; mfx: FX x npz; mfy: FY x npz; cx: CX x npz;
 * cy: CY x npz.  Returns nsplt (>0) or a negative error. */
int fvo_tracer_2d(const fvo_grid *g, int npz, int nq, double *q, double *dp1, double *mfx, double *mfy, double *cx,
                  double *cy, int hord, int q_split, int nord_tr, double trdm) {
  DIMS;
  int k, it, iq, nsplt;
  double *xfx = dalloc(nCX * npz), *yfx = dalloc(nCY * npz), *cmax = dalloc(npz), *frac = dalloc(npz);
  int *ksplt = (int *)calloc(npz, sizeof(int));
  if (g->grid_type < 3) return -1; /* the six faces are the caller's loop over prep / scale / step */
  fvo_tracer_2d_prep(g, npz, q_split, cx, cy, xfx, yfx, cmax);
  for (k = 0; k < npz; k++) ksplt[k] = 1;
  if (q_split == 0) { /* :404-417 */
    double c_global = cmax[0];
    if (npz != 1)
      for (k = 1; k < npz; k++) c_global = fmax(cmax[k], c_global);
    nsplt = (int)(1. + c_global);
  } else {
    nsplt = q_split;
  }
  if (nsplt != 1) {
    for (k = 0; k < npz; k++) {
      ksplt[k] = (int)(1. + cmax[k]);
      frac[k] = 1. / (double)ksplt[k];
    }
    fvo_tracer_2d_scale(g, npz, frac, cx, xfx, mfx, cy, yfx, mfy);
  }
  if (trdm > 1.e-4)
    for (k = 0; k < npz; k++) periodic_fill_A(g, dp1 + nA * k);
  for (it = 1; it <= nsplt; it++) {
    for (iq = 0; iq < nq; iq++)
      for (k = 0; k < npz; k++) periodic_fill_A(g, q + ((size_t)iq * npz + k) * nA);
    fvo_tracer_2d_step(g, npz, nq, it, nsplt, ksplt, q, dp1, mfx, mfy, cx, cy, xfx, yfx, hord, nord_tr, trdm);
  }
  free(xfx); free(yfx); free(cmax); free(frac); free(ksplt);
  return nsplt;
}


/* fill2D (model/fv_fill.F90:183-258), in the two halves around its mpp_update_domains (the caller's) */
void fvo_fill2d_mass(const fvo_grid *g, int km, const double *q, const double *delp, double *qt) { /* :228-235 */
  DIMS;
  int i, j, k;
  for (k = 1; k <= km; k++)
    for (j = js; j <= je; j++)
      for (i = is; i <= ie; i++) qt[nA * (k - 1) + IA(i, j)] = q[nA * (k - 1) + IA(i, j)] * delp[nA * (k - 1) + IA(i, j)] * g->area[IA(i, j)];
}
void fvo_fill2d_apply(const fvo_grid *g, int km, const double *qt, const double *delp, double *q) { /* :238-256 */
  DIMS;
  const double dif = 0.25;
  int i, j, k;
  double *fx = dalloc((size_t)(nx + 1) * ny), *fy = dalloc((size_t)nx * (ny + 1));
  for (k = 1; k <= km; k++) {
    for (j = js; j <= je; j++)
      for (i = is; i <= ie + 1; i++) {
        fx[IFX(i, j)] = 0.;
        if (qt[nA * (k - 1) + IA(i - 1, j)] * qt[nA * (k - 1) + IA(i, j)] < 0.) fx[IFX(i, j)] = qt[nA * (k - 1) + IA(i - 1, j)] - qt[nA * (k - 1) + IA(i, j)];
      }
    for (j = js; j <= je + 1; j++)
      for (i = is; i <= ie; i++) {
        fy[IFY(i, j)] = 0.;
        if (qt[nA * (k - 1) + IA(i, j - 1)] * qt[nA * (k - 1) + IA(i, j)] < 0.) fy[IFY(i, j)] = qt[nA * (k - 1) + IA(i, j - 1)] - qt[nA * (k - 1) + IA(i, j)];
      }
    for (j = js; j <= je; j++)
      for (i = is; i <= ie; i++)
        q[nA * (k - 1) + IA(i, j)] = q[nA * (k - 1) + IA(i, j)] + dif * (fx[IFX(i, j)] - fx[IFX(i + 1, j)] + fy[IFY(i, j)] - fy[IFY(i, j + 1)]) /
                                              (delp[nA * (k - 1) + IA(i, j)] * g->area[IA(i, j)]);
  }
  free(fx);
  free(fy);
}
