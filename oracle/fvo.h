/*
 * fvo.h -- CPU ORACLE for the FV3 dyn_core hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory is a plain-C, loop-for-loop restatement of the reference algorithm
 * (NOAA-GFDL/GFDL_atmos_cubed_sphere release 202411; citations are file:line under
 * /root/reference).  It exists so that the HIP kernels in gfdl_atmos_cubed_sphere_amd/csrc
 * can be checked against an independent CPU statement of the same arithmetic.
 *
 *   * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.
 *   * The product (libfv3_mi355x.so) never links, loads or falls back to anything here.
 *
 * PINNING STATUS ("parity unpinned" unless listed):
 *   pinned    : 1-D PPM flux operator xppm/yppm for hord 5, -5, 6, 8, 10 on a periodic
 *               line -- checked against vectors produced by executing the reference's own
 *               Python restatement docs/examples/tp_core.ipynb (tests/golden/ppm1d_*.npz,
 *               generator tests/golden/make_ppm1d_golden.py).  The same vectors also go straight through the
 *               LIBRARY's fv_tp_2d, without this oracle in between (tests/parity_common.py
 *               check_golden_ppm_through_fv_tp_2d; -m gpu and host-emulation tests, x and y sweeps, hord 5, -5, 6, 8).
 *               set_eta (L79, L127; restated in the package's test_cases.py) against the reference's own stand-alone
 *               fv_eta.F90 compiled here (oracle/Makefile target `ref` -> oracle/_ref/, tests/golden/set_eta_golden.npz).
 *   unpinned  : everything else (fv_tp_2d, c_sw, d_sw, column solvers, remap incl. ppm_profile, compute_total_energy and the
 *               energy fixer, Rayleigh_Super / _Friction, adv_pe, cubed_to_latlon).  The reference
 *               ships no unit tests / golden vectors, and its Fortran
 *               cannot be built here without writing stand-ins for the absent FMS library,
 *               which the build rules forbid.  Those operators are pinned only by the
 *               reference's conservation identities and other size-independent properties (tests/test_oracle_properties.py:
 *               column integrals / constants / linear profiles under every remap profile family; tests/parity_*.py: global
 *               mass on the six faces, energy closure of the fixer, the O(h^2) closure of adv_pe's uniform-pressure term).
 *
 * Scope of the restated branches: grid_type 4 (doubly periodic) and grid_type < 3 (the cubed sphere, one whole tile per
 * face: the edge / corner branches of c_sw, d_sw, fv_tp_2d, xppm / yppm, xtp_u / ytp_v, a2b_ord4, update_dz_c / _d), with
 * general (array-valued) metric terms, bounded_domain = .false., no nesting, no regional BCs.  Branches that are not
 * restated (remap_te, kord_wz < 0) return FVO_ERR_UNSUPPORTED.
 *
 * Array layout is the reference's (Fortran column-major, i fastest), with the exact
 * lower/upper bounds of model/fv_arrays.F90:1521-1563; see the accessor macros below.
 * Build: gcc -O2 -ffp-contract=off -fopenmp (no FMA contraction: results are the
 * straightforward IEEE evaluation of the reference's expressions, left to right).
 */
#ifndef FVO_H
#define FVO_H

#include <stddef.h>
/* exp / log: the deterministic pair shared with the HIP kernels (include/fv3_math.h explains why) */
#include "../include/fv3_math.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FVO_OK 0
#define FVO_ERR_UNSUPPORTED 2

/* fv_grid_bounds_type (model/fv_arrays.F90:1192-1200) + the gridstruct/flagstruct members
 * that the hot path reads (model/fv_arrays.F90:75-205, shapes :1749-1881). */
typedef struct fvo_grid {
  int is, ie, js, je, isd, ied, jsd, jed, ng;
  int npx, npy, grid_type;
  int bounded_domain, sw_corner, se_corner, ne_corner, nw_corner, stretched_grid;
  double da_min, da_min_c;
  /* (isd:ied, jsd:jed) */
  const double *area, *rarea, *dxa, *dya, *rdxa, *rdya, *cosa_s, *rsin2, *f0;
  /* (isd:ied, jsd:jed+1) */
  const double *dx, *rdx, *dyc, *rdyc, *cosa_v, *sina_v, *rsin_v, *divg_u, *del6_u;
  /* (isd:ied+1, jsd:jed) */
  const double *dy, *rdy, *dxc, *rdxc, *cosa_u, *sina_u, *rsin_u, *divg_v, *del6_v;
  /* (isd:ied+1, jsd:jed+1) */
  const double *rarea_c, *fC, *cosa, *sina;
  /* (is:ie+1, js:je+1) */
  const double *rsina;
  /* (isd:ied, jsd:jed, 9) */
  const double *sin_sg, *cos_sg;
  /* flagstruct members used by the kernels (sw_core.F90:126-127,590-591,624,964,1250) */
  double lim_fac;
  int do_diss_est, prevent_diss_cooling, do_f3d;
  /* cubed sphere only (grid_type < 3): A -> B interpolation weights on the face edges (edge_w/e(npy), edge_s/n(npx),
   * fv_grid_utils.F90:1121-1230) and the extrap_corner factors x1/(x2-x1) of a2b_ord4 (a2b_edge.F90:83-112, :452-462):
   * corners sw, se, ne, nw x the three (inner, outer) cell-centre pairs in the reference's order */
  const double *edge_w, *edge_e, *edge_s, *edge_n;
  double corner_f[12];
  const double *a11, *a12, *a21, *a22; /* cubed_to_latlon matrix (fv_grid_utils.F90:2255-2315), A layout; NULL = absent */
  const double *ec1, *ec2, *en1, *en2; /* adv_pe's unit vectors, component-last planes (A / (is:ie,js:je+1) / (is:ie+1,js:je)) */
} fvo_grid;

/* ---- tp_core (model/tp_core.F90) ------------------------------------------------------- */

/* 1-D PPM face values on one line.  q1 is indexable on [is-3, ie+3] (pointer to element 0 of a
 * virtual array), c and flux on [is, ie+1].  Doubly-periodic / bounded branch of
 * xppm (tp_core.F90:324-712) == yppm (:715-1152) applied to one line. */
int fvo_ppm_line(const double *q1, const double *c, double *flux, int is, int ie, int iord,
                 double lim_fac);

int fvo_ppm_line_cs(const double *q1, const double *c, double *flux, int is, int ie, int iord,
                    double lim_fac, const double *dxa, int npx);
void fvo_copy_corners(const fvo_grid *g, double *q, int dir);
/* pert_ppm (tp_core.F90:1206-1264) */
void fvo_pert_ppm(int im, const double *a0, double *al, double *ar, int iv);

/* fv_tp_2d (tp_core.F90:85-241).  Optional arguments are nullable pointers; nord<0 means
 * "nord/damp_c not present". */
int fvo_fv_tp_2d(const fvo_grid *g, double *q, const double *crx, const double *cry, int hord,
                 double *fx, double *fy, const double *xfx, const double *yfx, const double *ra_x,
                 const double *ra_y, const double *mfx, const double *mfy, const double *mass,
                 int nord, double damp_c);

/* deln_flux (tp_core.F90:1267-1447), damp_Km absent. */
int fvo_deln_flux(const fvo_grid *g, int nord, double damp, const double *q, double *fx, double *fy,
                  const double *mass);

/* ---- sw_core (model/sw_core.F90) ------------------------------------------------------- */

void fvo_fill_4corners(const fvo_grid *g, double *q, int dir);
void fvo_fill_corners_b(const fvo_grid *g, double *q, int dir);
void fvo_fill_corners_dgrid(const fvo_grid *g, double *x, double *y, double mySign);
int fvo_d2a2c_vect(const fvo_grid *g, const double *u, const double *v, double *ua, double *va,
                   double *uc, double *vc, double *ut, double *vt, int dord4);
int fvo_divergence_corner(const fvo_grid *g, const double *u, const double *v, const double *ua,
                          const double *va, double *divg_d);
int fvo_del6_vt_flux(const fvo_grid *g, int nord, double damp, const double *q, double *d2,
                     double *fx2, double *fy2);
int fvo_xtp_u(const fvo_grid *g, const double *c, const double *u, const double *v, double *flux,
              int iord);
int fvo_ytp_v(const fvo_grid *g, const double *c, const double *u, const double *v, double *flux,
              int jord);
int fvo_a2b_ord4(const fvo_grid *g, double *qin, double *qout, int replace);
int fvo_smag_corner(const fvo_grid *g, double dt, const double *u, const double *v, double *smag_c);

/* c_sw, one k-slab (sw_core.F90:79-488).  w/wc may be NULL when hydrostatic. */
int fvo_c_sw(const fvo_grid *g, double *delpc, double *delp, double *ptc, double *pt, double *u,
             double *v, double *w, double *uc, double *vc, double *ua, double *va, double *wc,
             double *ut, double *vt, double *divg_d, int nord, double dt2, int hydrostatic,
             int dord4);

/* d_sw, one k-slab (sw_core.F90:494-1606); use_cond optional (q_con nullable). */
typedef struct fvo_dsw_par {
  double dt;
  int hord_tr, hord_mt, hord_vt, hord_tm, hord_dp;
  int nord, nord_v, nord_w, nord_t;
  double dddmp, d2_bg, d4_bg, damp_v, damp_w, damp_t, d_con, kgb;
  int hydrostatic, use_cond;
  /* inline_q (sw_core.F90:1020-1043): nq tracers advected inside d_sw.  q: the slab of tracer 0 at this level (fvo_d_sw) / the
   * A x npz x nq array (fvo_d_sw_3d); q_stride: doubles between tracers */
  int inline_q, nq;
  double *q;
  size_t q_stride;
} fvo_dsw_par;

int fvo_d_sw(const fvo_grid *g, const fvo_dsw_par *p, double *delpc, double *delp, double *ptc,
             double *pt, double *u, double *v, double *w, double *uc, double *vc, double *ua,
             double *va, double *divg_d, double *xflux, double *yflux, double *cx, double *cy,
             double *crx_adv, double *cry_adv, double *xfx_adv, double *yfx_adv, double *q_con,
             double *heat_source, double *diss_est);

/* all-k drivers (the OpenMP k-loops of dyn_core.F90:436-447 and :658-812).  3-D arrays are
 * the 2-D slabs above stacked in k. Per-level coefficient arrays have length npz. */
int fvo_c_sw_3d(const fvo_grid *g, int npz, double *delpc, double *delp, double *ptc, double *pt,
                double *u, double *v, double *w, double *uc, double *vc, double *ua, double *va,
                double *wc, double *ut, double *vt, double *divg_d, int nord, double dt2,
                int hydrostatic, int dord4);

typedef struct fvo_dsw_levels {
  const int *nord_k, *nord_v, *nord_w, *nord_t;
  const double *d2_divg, *damp_vt, *damp_w, *damp_t, *d_con_k;
} fvo_dsw_levels;

int fvo_d_sw_3d(const fvo_grid *g, int npz, const fvo_dsw_par *p, const fvo_dsw_levels *lv,
                double *delpc, double *delp, double *ptc, double *pt, double *u, double *v,
                double *w, double *uc, double *vc, double *ua, double *va, double *divg_d,
                double *mfx, double *mfy, double *cx, double *cy, double *crx, double *cry,
                double *xfx, double *yfx, double *q_con, double *heat_source, double *diss_est);

/* ---- nonhydrostatic column path (oracle/nh_core.c) ------------------------------------------ */
int fvo_update_dz_c(const fvo_grid *g, int km, double dt, const double *dp0, const double *zs,
                    const double *ut, const double *vt, double *gz, double *ws);
int fvo_riem_solver_c(const fvo_grid *g, int km, double dt, double akap, double ptop, const double *hs,
                      const double *w3, const double *pt, const double *delp, double *gz, double *pef,
                      const double *ws, double p_fac, double a_imp, double grav, double rdgas, const double *q_con,
                      const double *cappa);
int fvo_riem_solver3(const fvo_grid *g, int km, double dt, double akap, double ptop, const double *zs, double *w,
                     double *delz, const double *pt, const double *delp, double *zh, double *pe, double *ppe,
                     double *pk3, double *pk, double *peln, const double *ws, double p_fac, double a_imp,
                     int use_logp, int last_call, int fp_out, double grav, double rdgas, const double *q_con,
                     const double *cappa);
/* the same with flagstruct%m_split (the sub-steps of RIM_2D, taken for a_imp <= 0.5; nh_utils.F90:751-982) */
int fvo_riem_solver_c_ms(const fvo_grid *g, int km, double dt, double akap, double ptop, const double *hs,
                         const double *w3, const double *pt, const double *delp, double *gz, double *pef,
                         const double *ws, double p_fac, double a_imp, double grav, double rdgas, const double *q_con,
                         const double *cappa, int m_split);
int fvo_riem_solver3_ms(const fvo_grid *g, int km, double dt, double akap, double ptop, const double *zs, double *w,
                        double *delz, const double *pt, const double *delp, double *zh, double *pe, double *ppe,
                        double *pk3, double *pk, double *peln, const double *ws, double p_fac, double a_imp,
                        int use_logp, int last_call, int fp_out, double grav, double rdgas, const double *q_con,
                        const double *cappa, int m_split);
int fvo_update_dz_d(const fvo_grid *g, int km, int *ndif, double *damp, int hord, const double *dp0, const double *zs,
                    double *zh, const double *crx, const double *cry, const double *xfx, const double *yfx, double *ws,
                    double rdt);
int fvo_p_grad_c(const fvo_grid *g, int npz, double dt2, const double *delpc, const double *pkc, const double *gz,
                 double *uc, double *vc, int hydrostatic);
int fvo_nh_p_grad(const fvo_grid *g, int npz, double *u, double *v, double *pp, double *gz, double *delp, double *pk,
                  double dt, double top_value);
int fvo_pk3_halo(const fvo_grid *g, int npz, double ptop, double akap, double *pk3, const double *delp, int use_logp);
int fvo_pe_halo(const fvo_grid *g, int npz, double ptop, double *pe, const double *delp);
int fvo_divg2_ext(const fvo_grid *g, int npz, double d_ext, const double *delp, const double *vt, double *divg2);
int fvo_one_grad_p_hydro(const fvo_grid *g, int npz, double dt, double ptk, const double *divg2, double *u, double *v,
                         double *pk, double *gz);
int fvo_one_grad_p_nh(const fvo_grid *g, int npz, double dt, double ptop, const double *divg2, double *u, double *v, double *pk,
                      double *gz, const double *delp);
int fvo_split_p_grad(const fvo_grid *g, int npz, double *u, double *v, double *pp, double *gz, double *delp, double *pk, double beta,
                     double dt, double top_value, double *du, double *dv);
int fvo_grad1_p_update(const fvo_grid *g, int npz, const double *divg2, double *u, double *v, double *pk, double *gz, double dt,
                       double ptk, double beta, double *du, double *dv);
int fvo_del2_cubed(const fvo_grid *g, int km, double cd, int nmax, double *q);
int fvo_apply_heat_source(const fvo_grid *g, int npz, int n_con, int hydrostatic, double bdt, double delt_max,
                          double cp_air, double cv_air, double rdgas, double grav, double *pt, double *heat_source,
                          const double *delp, const double *delz, double *pkz, const double *cappa);
int fvo_geopk(const fvo_grid *g, int km, double ptop, double akap, double cp_air, double *pe, double *peln,
              const double *delp, double *pk, double *gz, const double *hs, const double *pt, double *pkz, int CG);

/* ---- vertical remap (oracle/mapz.c) ------------------------------------------------------------- */
/* scalar_profile (is_scalar=1) / cs_profile (0) for one column; a4 is [4][km+2] (index (n-1)*(km+2)+k). */
int fvo_profile_column(int is_scalar, double qs, double *a4, const double *delp, int km, int iv, int kord, double qmin);
/* which: 0 map_scalar, 1 map1_ppm, 2 map1_q2, 3 mapn_tracer.  1-based columns (element 0 unused). */
int fvo_remap_column(int which, int km, const double *pe1, const double *pe2, const double *q1, double *q2, double qs,
                     int iv, int kord, double qmin);
typedef struct fvo_remap_par {
  int last_step, hydrostatic, adiabatic, nq, kord_mt, kord_wz, kord_tm;
  const int *kord_tr;
  double akap, ptop, rdgas, grav, cv_air, r_vir, cp, t_min;
  int sphum;
  /* thermostruct%moist_kappa / use_cond (nonhydrostatic only) and what moist_cv needs (fv_thermodynamics.F90:250-325):
   * nwat and the 1-based tracer indices of the water species (0 = absent), cv_vap = 3*rvgas, c_liq, c_ice (gfdl_mp) */
  int moist_kappa, use_cond, nwat, liq_wat, rainwat, ice_wat, snowwat, graupel;
  double cv_vap, c_liq, c_ice;
  int fill; /* flagstruct%fill: fillz on the remapped tracers (fv_operators.F90:337, fv_fill.F90:34-137) */
  /* flagstruct%remap_te (fv_mapz.F90:232-286, :348-360, :576-619, :655-663): hs = phis (A), te: A x km work array (the
   * reference's te argument; fv_dynamics hands it dp1) */
  int remap_te;
  const double *hs;
  double *te;
} fvo_remap_par;
/* fillz of one column of one tracer (fv_fill.F90:34-137, the default (not DEV_GFS_PHYS) branch); q, dp: 1-based [1..km] */
void fvo_fillz_column(int km, double *q, const double *dp);
/* q_con, cappa: A x km, written when moist_kappa (fv_mapz.F90:212-219, :463-478); may be NULL otherwise */
int fvo_lagrangian_to_eulerian(const fvo_grid *g, int km, const fvo_remap_par *p, double *ps, double *pe, double *delp,
                               double *pkz, double *pk, double *u, double *v, double *w, double *delz, double *pt,
                               double *q, double *peln, double *omga, const double *ws, const double *ak,
                               const double *bk, double *q_con, double *cappa);
/* consv_te: compute_total_energy (fv_thermodynamics.F90:90-225), the energy fixer sums (fv_mapz.F90:647-763) and step 9a with
 * the fixer's increment (:793-821); fvo_lagrangian_to_eulerian with last_step = 2 leaves that last conversion to fvo_remap_finish */
int fvo_compute_total_energy(const fvo_grid *g, int km, const fvo_remap_par *p, int moist_phys, const double *u,
                             const double *v, const double *w, const double *delz, const double *pt, const double *delp,
                             const double *q, const double *qc, const double *pe, const double *peln, const double *hs,
                             double *te_2d);
int fvo_energy_fixer_sums(const fvo_grid *g, int km, const fvo_remap_par *p, int only_sums, const double *u, const double *v,
                          const double *w, const double *delz, const double *pt, const double *delp, const double *q,
                          const double *pe, const double *peln, const double *hs, const double *pkz, const double *pk,
                          const double *te0_2d, double *te_2d, double *zsum1, double *zsum0, double *q_con);
int fvo_remap_finish(const fvo_grid *g, int km, const fvo_remap_par *p, double dtmp, double *pt, const double *pkz,
                     const double *q);
/* moist_cv for one cell (fv_thermodynamics.F90:250-325, without the t1 special case): returns cvm, sets *q_con.
 * qk points at q(i,j,k,1); species stride ns. */
double fvo_moist_cv(const fvo_remap_par *p, const double *qk, size_t ns, double *q_con);

/* ---- tracer_2d (oracle/tracer2d.c) ---------------------------------------------------------------- */
int fvo_tracer_2d(const fvo_grid *g, int npz, int nq, double *q, double *dp1, double *mfx, double *mfy, double *cx,
                  double *cy, int hord, int q_split, int nord_tr, double trdm);
void fvo_fill2d_mass(const fvo_grid *g, int km, const double *q, const double *delp, double *qt);
void fvo_fill2d_apply(const fvo_grid *g, int km, const double *qt, const double *delp, double *q);
/* the pieces of tracer_2d between its reductions / halo updates (the caller's, for the six faces of the cubed sphere) */
void fvo_tracer_2d_prep(const fvo_grid *g, int npz, int q_split, const double *cx, const double *cy, double *xfx, double *yfx,
                        double *cmax);
void fvo_tracer_2d_scale(const fvo_grid *g, int npz, const double *frac, double *cx, double *xfx, double *mfx, double *cy,
                         double *yfx, double *mfy);
void fvo_tracer_2d_step(const fvo_grid *g, int npz, int nq, int it, int nsplt, const int *ksplt, double *q, double *dp1,
                        const double *mfx, const double *mfy, const double *cx, const double *cy, const double *xfx,
                        const double *yfx, int hord, int nord_tr, double trdm);

#ifdef __cplusplus
}
#endif
/* ---- fv_dynamics around the k_split loop (oracle/dyn_pre.c) ------------------------------------- */
int fvo_c2l(const fvo_grid *g, int km, int ord, const double *u, const double *v, double *ua, double *va);
int fvo_rayleigh_rf(int npz, double dt, double tau, double rf_cutoff, double ptop, const double *pm, double *rf);
int fvo_rayleigh_u2f(const fvo_grid *g, int kmax, int hydrostatic, const double *u, const double *v, const double *w,
                     double *ua, double *va, double *u2f);
int fvo_rayleigh_apply(const fvo_grid *g, int kmax, int conserve, int hydrostatic, double cp, double rg, double ptop,
                       const double *pm, const double *rf, double *u2f, double *pt, double *delz, double *u, double *v,
                       double *w);
/* adv_pe, dyn_core.F90:1529-1632 (cubed sphere): om += 0.5*rarea*(V3 . grad pe); pem from delp_before on (is-1:ie+1, js-1:je+1) */
int fvo_adv_pe(const fvo_grid *g, int km, double ptop, const double *ua, const double *va, const double *delp_before, double *om);
/* consv_am: compute_aam (fv_dynamics.F90:1266-1314) and the wind correction (:784-798) */
int fvo_compute_aam(const fvo_grid *g, int npz, double radius, double omega, double agrav, double ptop, const double *coslat,
                    const double *ua, const double *delp, double *aam, double *m_fac, double *ps);
int fvo_consv_am_apply(const fvo_grid *g, int npz, double u00, const double *l2c_u, const double *l2c_v, double *u, double *v);
/* Ray_fast (dyn_core.F90:2485-2601) and fast_tau_w_sec (nh_utils.F90:356-367, :1363-1371, :1498-1506) */
int fvo_ray_fast_profile(int npz, int ks, double dt, double tau, double rf_cutoff, double ptop, const double *pfull, const double *dp,
                         double *rf, int *k_rf, double *dm_out);
int fvo_ray_fast(const fvo_grid *g, int npz, int kmax, int k_rf, const double *rf, const double *dp, int hydrostatic, double *u,
                 double *v, double *w);
int fvo_fast_tau_w_rff(int km, double dt, double fast_tau_w_sec, double rf_cutoff, double ptop, const double *pfull, double *rff);
/* mix_dp (dyn_core.F90:2119-2200, flagstruct%fill_dp; CG = .false.): delp, pt, w (A x km) in place */
int fvo_mix_dp(const fvo_grid *g, int km, int hydrostatic, const double *ak, const double *bk, double *w, double *delp, double *pt);
int fvo_set_fast_tau_w(int k_rf, const double *rff);
int fvo_rayleigh_super(const fvo_grid *g, int kmax, int conserve, int hydrostatic, double cp, double rg, double ptop,
                       const double *pm, const double *rf, const double *ua, const double *va, double *pt, double *u,
                       double *v, double *w, const double *u00, const double *v00);

#endif
